"""HipGraphRunner: the vectorised rollout of HipVecRunner as hipGraph replays (10 timesteps per replay by default).

Same data flow and stored batch as HipVecRunner / the reference loop (episode_runner.py:57-119); what changes is the
mechanics that make a timestep capturable and replayable for every t:
  * the episode storage is persistent and indexed by a device-side time counter; when the replay buffer's size is a small
    multiple of the env batch the storage IS the next slots of the replay buffer (ReplayBuffer.reserve), so inserting the
    finished batch moves no data.  Everything that holds pointers into a storage lives in a per-storage bundle;
  * the controller state (hidden states, previous action / reward / incentives) lives in static tensors; the t == 0 history
    features are zeros because the previous action is -1 (all-zero one-hot);
  * with FastPolicy's fused kernels (default) a timestep is 3 launches (pipelined): k_head<env> on the features the previous
    timestep's launch left, the fused env step + observe (writes obs[:, t + 1] of the storage), and k_inc_encode = k_head<inc> of
    t together with k_encode of slot t + 1 (they share no data; the input rows live in a buffer pair indexed by the parity of t);
    the heads file actions / pose / rewards in slot t, carry the runner state and hand the device-side counters over.  Several env
    groups or an odd number of timesteps per graph take the four standalone launches (k_encode, k_head<env>, env, k_head<inc>).
    Other configurations (obs_others_last_action, fast_policy=False) take the generic torch timestep, captured the same way;
  * epsilon is a device scalar; exploration uses the package's counter generator (no multinomial, no host sync).
The first episode runs eagerly (warm-up of hipBLASLt plans and the allocator); graphs are captured from the second on.
The returned EpisodeBatch is the persistent storage: consume it (buffer.insert_episode_batch) before the next run(), as the
training loop does (run.py:184-185).
"""
import ctypes as C

import torch as th
import torch.nn.functional as F

from .. import abi, ops

from ..components.episode_buffer import EpisodeBatch
from .hip_vec_runner import HipVecRunner


class HipGraphRunner(HipVecRunner):
    def setup(self, scheme, groups, preprocess, mac):
        super().setup(scheme, groups, preprocess, mac)
        self._scheme, self._groups, self._preprocess = scheme, groups, preprocess
        self._graph = None
        self._episodes = 0
        self._ready = False
        self._bundles, self._own_store, self._replay = {}, None, None
        self._graph_steps = 1

    def set_replay_buffer(self, buffer):
        """Write training episodes directly into `buffer` (ReplayBuffer.reserve) when its size is a small multiple (<= 8) of the env
        batch and it lives on the env's device; buffer.insert_episode_batch(batch) then moves no data."""
        N = self.batch_size
        ok = (buffer is not None and buffer.buffer_size % N == 0 and buffer.buffer_size // N <= 8
              and th.device(buffer.device) == th.device(self.args.device) and buffer.max_seq_length == self.episode_limit + 1)
        self._replay = buffer if ok else None
        return ok

    # ---- static state ---------------------------------------------------------------------------------------------
    def _allocate(self):
        a, N, n, T = self.args, self.batch_size, self.args.n_agents, self.episode_limit
        dev = self.env.device
        simplified = self.env.native.cfg.obs_color == abi.COLOR_SIMPLIFIED
        # the fused encoder reads u8 class codes: the env kernel emits them next to the f32 observation (format R storage) or the
        # storage itself holds them (format C)
        self._want_code = bool(getattr(a, "fast_policy", True) and getattr(a, "fused_policy", True) and simplified
                               and self.obs_fmt != abi.OBS_CODE and self.env.native.V in (15, 31))
        self._dense_cur = self.env.native.obs_buffers(self.obs_fmt, want_code=self._want_code)   # obs / pos / orient written by the env kernel
        self.cur = self._dense_cur
        self.t_dev = th.zeros(1, dtype=th.long, device=dev)
        self.rng_ctr = th.zeros(1, dtype=th.long, device=dev)          # never reset: exploration draws differ between episodes
        self.prev_actions = th.full((N, n), -1, dtype=th.long, device=dev)
        self.prev_reward = th.zeros(N, n, device=dev)
        self.prev_inc = th.zeros(N, n, n, dtype=th.long, device=dev)
        # the same incentive actions as receiver-major bytes [n(receiver), N, 16] (ssd_policy_head.recv_inc): what the fused env head reads
        self.recv_inc = th.zeros(n, N, 16, dtype=th.uint8, device=dev) if n <= 16 else None
        H = a.rnn_hidden_dim
        self.h_env = th.zeros(N, n, 1, H, device=dev)
        self.h_inc = th.zeros(N, n, 1, H, device=dev)
        self.eps = th.zeros((), device=dev)
        self.ep_return = th.zeros(N, n, device=dev)
        avail = self.env.avail_actions_batch[0, 0]
        self.avail_idx = th.nonzero(avail).squeeze(-1)                  # constant table of available env actions
        self.avail_mask = self.env.avail_actions_batch                  # [N, n, A]
        self.inc_mask = (1 - th.eye(n, device=dev, dtype=th.long)).reshape(1, n, n)
        self.fast = None
        self.direct_obs = self.fold_store = False
        from ..fast_policy import FastPolicy
        use_fused = bool(getattr(a, "fused_policy", True)) and simplified
        if getattr(a, "fast_policy", True) and FastPolicy.supports(self.mac, use_fused):   # else: the generic captured timestep
            # Optionally the policy work of a timestep is evaluated per env GROUP on separate streams (fork/join inside the
            # captured graph) around the single full-batch env launch.  Measured on MI355X / ROCm 7.2: no gain (the graph
            # runs the branches back to back), so the default is one group.
            G = int(getattr(a, "policy_groups", 1))
            if N % G or N // G < 1:
                G = 1
            self.groups, hsz = G, N // G
            self.actions_full = th.zeros(N, n, dtype=th.long, device=dev)
            self.actions_inc_full = th.zeros(N, n, n, dtype=th.long, device=dev)
            seed = int(self.env.native.cfg.seed) * 2654435761 + 12345
            prec = 1 if str(getattr(a, "qnet_dtype", "fp32")).lower() in ("bf16", "bfloat16") else 2
            base = int(self.env.native.cfg.env_id_base)
            self.fasts = []
            for g in range(G):
                sl = slice(g * hsz, (g + 1) * hsz)
                # one exploration seed for every group and rank: the draws are keyed by the GLOBAL env id (env_id_base + local env)
                self.fasts.append(FastPolicy(self.mac, hsz, avail, seed=seed, actions_out=self.actions_full[sl],
                                             actions_inc_out=self.actions_inc_full[sl],
                                             share_packs_from=self.fasts[0] if g else None,
                                             fused=use_fused, precision=prec,
                                             env_id_base=base + g * hsz))
            self.fast = self.fasts[0]
            self.gslices = [slice(g * hsz, (g + 1) * hsz) for g in range(G)]
            self.side_streams = [th.cuda.Stream(device=dev) for _ in range(G)] if G > 1 else []
            self._zeros_nn = th.zeros(N, n, device=dev)
            self.pos_t, self.orient_t = th.zeros(N, n, 2, device=dev), th.zeros(N, n, 2, device=dev)   # pose before the env step
            self.actions_i32 = th.zeros(N, n, dtype=th.int32, device=dev)
            self.t_store = th.zeros(1, dtype=th.long, device=dev)     # the encoder's copy of t_dev, read by the store-step launch
            # fused encoder: the env kernel writes obs[:, t + 1] of the storage itself (and the class codes the encoder reads)
            self.direct_obs = self.fast.fused_enc and self.obs_fmt in (abi.OBS_F32, abi.OBS_CODE)
            self.fold_store = self.direct_obs and G == 1 and bool(getattr(a, "fold_store", True))
            if self.obs_fmt == abi.OBS_CODE and not self.direct_obs:
                # class-code storage is consumed by the fused encoder only; other window sizes take the generic timestep
                # (the torch controller expands the codes itself)
                self.fast, self.fasts, self.fold_store = None, [], False
        # Pipelined timestep (3 launches): env head -> env step -> [inc head of t + encoder of t + 1] as one launch
        # (FastPolicy.act_inc_encode).  The encoder writes the OTHER buffer of FastPolicy.inputs_pair, so the buffer of a timestep is
        # its parity -- baked into the captured graph, hence an even number of timesteps per graph.
        K = max(1, int(getattr(a, "steps_per_graph", 10)))
        while self.episode_limit % K:
            K -= 1
        self._graph_steps_planned = K
        use_graph = bool(getattr(a, "rollout_graph", True))
        self.pipe = bool(self.fast is not None and self.fold_store and self.fast.fused and self.fast.fused_enc and self.groups == 1
                         and getattr(a, "pipeline_encode", True) and (K % 2 == 0 or not use_graph)
                         and (self.obs_fmt == abi.OBS_CODE or self._want_code))
        self.rng_copy = th.zeros(1, dtype=th.long, device=dev)          # pipelined: the env head's copy of rng_ctr for the inc head
        self._par = 0
        self._ready = True

    def _bind_store(self, store):
        """Select the episode storage of this episode.  Everything that holds pointers into a storage (output buffers of the env
        kernel, store-step arguments, the captured graph) lives in a per-storage bundle created on first use."""
        from types import SimpleNamespace
        st = store.data.transition_data
        key = st["obs"].data_ptr()
        b = self._bundles.get(key)
        self.store = store
        if b is None:
            st["filled"].fill_(1)                                       # fixed-length episodes: every slot is filled
            st["avail_actions"].copy_(self.env.avail_actions_batch.unsqueeze(1).expand(-1, self.episode_limit + 1, -1, -1))
            b = SimpleNamespace(graph=None, cur=self._dense_cur, ss=None, ss_last=None, file_env=None, file_inc=None, file_inc_last=None,
                                begin_graph=None, finish_graph=None)
            if self.fast is not None:
                if self.direct_obs:
                    b.cur = self.env.native.storage_obs_buffers(st["obs"], self.obs_fmt, want_code=self._want_code)
                b.ss, b.ss_last = self._make_store_args(True), self._make_store_args(False)
                if self.fold_store:     # the two heads file their results in the storage themselves: no store-step launch
                    out, slots = self.env.native.out, self.episode_limit + 1
                    common = dict(t_index=self.t_store.data_ptr(), t_slots=slots)
                    b.file_env = dict(common, dst_pos=st["agent_pos"].data_ptr(), dst_orient=st["agent_orientation"].data_ptr(),
                                      dst_actions=st["actions"].data_ptr(), dst_actions_onehot=st["actions_onehot"].data_ptr(),
                                      prev_actions_out=self.prev_actions.data_ptr())
                    b.file_inc_last = dict(common, dst_actions_inc=st["actions_inc"].data_ptr())
                    if self.recv_inc is not None and self.groups == 1:
                        b.file_env = dict(b.file_env, recv_inc=self.recv_inc.data_ptr())
                    b.file_inc = dict(b.file_inc_last, prev_actions_inc_out=self.prev_inc.data_ptr(), dst_reward=st["reward"].data_ptr(),
                                      dst_clean_num=st["clean_num"].data_ptr(), dst_apple_den=st["apple_den"].data_ptr(),
                                      dst_terminated=st["terminated"].data_ptr(), terminated=out["terminated"].data_ptr(),
                                      prev_reward_out=self.prev_reward.data_ptr(), ep_return=self.ep_return.data_ptr(),
                                      next_t_out=self.t_dev.data_ptr())
                    if self.recv_inc is not None and self.groups == 1:
                        b.file_inc = dict(b.file_inc, recv_inc_out=self.recv_inc.data_ptr())
                    if self.pipe:
                        # counter hand-over without a launch writing a scalar it reads: the env head reads the masters (t_dev, rng_ctr)
                        # and writes the copies (t_store, rng_copy); the inc head reads the copies and writes the masters' next values
                        b.file_env = dict(b.file_env, t_index=self.t_dev.data_ptr(), t_copy_out=self.t_store.data_ptr(),
                                          step_copy_out=self.rng_copy.data_ptr())
                        b.file_inc = dict(b.file_inc, next_step_out=self.rng_ctr.data_ptr())
                        b.file_inc_last = dict(b.file_inc_last, next_step_out=self.rng_ctr.data_ptr())
            self._bundles[key] = b
        self._bundle, self.cur, self._ss, self._ss_last, self._graph = b, b.cur, b.ss, b.ss_last, b.graph

    def _pick(self, q, avail_mask, idx_table, k):
        """epsilon-greedy (action_selectors.py:44-68) without host sync: random AVAILABLE action = table[floor(u * k)]."""
        if avail_mask is not None:
            q = q.masked_fill(avail_mask == 0, -float("inf"))
        greedy = q.argmax(dim=-1)
        u = th.rand(greedy.shape, device=q.device)
        r = th.clamp((th.rand(greedy.shape, device=q.device) * k).long(), max=k - 1)
        rand = idx_table[r] if idx_table is not None else r
        return th.where(u < self.eps, rand, greedy)

    def _fork(self, fn):
        """run fn(g) for every env group, each on its own stream, then join (capturable fork/join)."""
        if self.groups == 1:
            fn(0)
            return
        main = th.cuda.current_stream(self.env.device)
        for g, s in enumerate(self.side_streams):
            s.wait_stream(main)
            with th.cuda.stream(s):
                fn(g)
        for s in self.side_streams:
            main.wait_stream(s)

    def _fast_stages(self, store_env_step):
        """The launches of one timestep with the FastPolicy kernels, in order, as (kernel name, key, closure).  Fused path
        (default): k_encode reads obs[:, t] from the storage where the env kernel put it, the two head kernels file actions / pose /
        rewards into slot t and carry the previous-step inputs, return and counters themselves -- a timestep is 4 launches
        (encode, env head, env step+observe, inc head), or 3 when pipelined (self.pipe: env head, env step+observe, inc head of t +
        encoder of t + 1 as one launch).  Otherwise one store-step launch writes the nine small fields."""
        st = self.store.data.transition_data
        td = self.t_dev
        obs, pos, orient = self.cur["obs"], self.cur["pos"], self.cur["orient"]
        fused = self.fast.fused
        actions = self.actions_full
        pos_t, orient_t = self.pos_t, self.orient_t                     # forward_inc sees the PRE-step pose (controller :78-82)
        bundle = self._bundle

        if self.pipe:
            par = self._par                                             # buffer of this timestep = its parity
            codes = st["obs"] if self.obs_fmt == abi.OBS_CODE else self.cur["code"]

            def env_head_p():
                self.fast.head_env(self.prev_actions, self.prev_reward, self.prev_inc, pos, self.eps, self.rng_ctr, file=bundle.file_env,
                                   orient=orient, actions_i32=self.actions_i32, pos_copy=pos_t, orient_copy=orient_t, buf=par)

            def env_step_p():
                self.env.step_batch(self.actions_i32, observe=True, fmt=self.obs_fmt, out=self.cur)

            def inc_encode_p():
                out = self.env.native.out
                self.fast.act_inc_encode(actions, pos_t, orient_t, out["reward"], out["clean_num"], out["apple_den"], self.eps, self.rng_copy,
                                         codes, slot_t=self.t_store, slot_add=1, buf=par, file=bundle.file_inc)

            def inc_last_p():      # slot T: zeros for reward / clean_num / apple_den, nothing left to encode
                z = self._zeros_nn
                self.fast.act_inc(actions, pos_t, orient_t, z, z, z, self.eps, self.rng_copy, file=bundle.file_inc_last, buf=par)

            if store_env_step:
                return [("ssd::k_head<env>", "head_env", env_head_p), ("ssd::k_env<MODE_STEP_OBS>", "env", env_step_p),
                        ("ssd::k_inc_encode", "inc_encode", inc_encode_p)]
            return [("ssd::k_head<env>", "head_env", env_head_p), ("ssd::k_head<inc>", "head_inc", inc_last_p)]

        def encode(g):
            sl = self.gslices[g]
            if self.direct_obs:     # the observation is already in the storage; the encoder reads its class codes
                code_store = self.obs_fmt == abi.OBS_CODE
                self.fasts[g].encode(None, codes=(st["obs"] if code_store else self.cur["code"])[sl], slot_t=td,
                                     t_copy=self.t_store if g == 0 else None, counter_inc=self.rng_ctr if self.fold_store else None)
            else:
                self.fasts[g].encode(obs[sl], codes=self.cur["code"][sl] if "code" in self.cur else None, store_obs=st["obs"][sl], store_t=td)

        def env_head(g):
            sl = self.gslices[g]
            extra = dict(orient=orient[sl], actions_i32=self.actions_i32[sl], pos_copy=pos_t[sl], orient_copy=orient_t[sl]) if fused else {}
            self.fasts[g].head_env(self.prev_actions[sl], self.prev_reward[sl], self.prev_inc[sl], pos[sl], self.eps, self.rng_ctr,
                                   file=bundle.file_env, **extra)

        def env_step():
            if not fused:
                pos_t.copy_(pos); orient_t.copy_(orient)
                self.actions_i32.copy_(actions)
            if store_env_step:
                self.env.step_batch(self.actions_i32, observe=True, fmt=self.obs_fmt, out=self.cur if self.direct_obs else None)

        def inc_head(g):
            sl = self.gslices[g]
            if store_env_step:
                out = self.env.native.out
                reward, clean, den = out["reward"], out["clean_num"], out["apple_den"]
            else:
                reward = clean = den = self._zeros_nn
            self.fasts[g].act_inc(actions[sl], pos_t[sl], orient_t[sl], reward[sl], clean[sl], den[sl], self.eps, self.rng_ctr,
                                  file=bundle.file_inc if store_env_step else bundle.file_inc_last)

        return [("ssd::k_encode", "encode", lambda: self._fork(encode)),
                ("ssd::k_head<env>", "head_env", lambda: self._fork(env_head)),
                ("ssd::k_env<MODE_STEP_OBS>", "env", env_step),
                ("ssd::k_head<inc>", "head_inc", lambda: self._fork(inc_head))]

    def timestep_launches(self):
        """(kernel name, key, closure) of the launches of one rollout timestep on the live buffers -- what the rollout hipGraph was
        captured from; bench.py times them one by one with HIP events (a graph replay has no host call to bracket)."""
        if self.fast is None or not self.fast.fused or not self.fold_store:
            return []
        self._par = self.t & 1
        return self._fast_stages(True)

    def _select_fast(self, store_env_step):
        """_select with the FastPolicy kernels (see _fast_stages)."""
        td = self.t_dev
        for _, _, fn in self._fast_stages(store_env_step):
            fn()
        if self.fold_store:
            return
        ss = self._ss if store_env_step else self._ss_last
        # ONE launch: the nine small fields of slot t, the controller's "previous step" inputs, the episode return and the
        # time / exploration counters (incremented after every block has read t)
        abi.check(self.fast.lib, self.fast.lib.ssd_store_step_launch(C.byref(ss), th.cuda.current_stream(self.env.device).cuda_stream))
        if not self.direct_obs:     # without the encoder's t copy the counters are advanced by two small torch kernels
            self.rng_ctr += 1
            if store_env_step:
                td += 1

    def _make_store_args(self, with_step_outputs):
        st, out, fp = self.store.data.transition_data, self.env.native.out, self.fast
        ss = abi.SsdStoreStep()
        ss.t_index = self.t_dev.data_ptr()
        ss.n_env, ss.n_agents, ss.n_actions, ss.t_slots = self.batch_size, self.args.n_agents, self.args.n_actions, self.episode_limit + 1
        ss.actions, ss.actions_inc = self.actions_full.data_ptr(), self.actions_inc_full.data_ptr()
        ss.pos, ss.orient = self.pos_t.data_ptr(), self.orient_t.data_ptr()
        ss.dst_pos, ss.dst_orient = st["agent_pos"].data_ptr(), st["agent_orientation"].data_ptr()
        if self.direct_obs:
            ss.t_index, ss.counter_inc = self.t_store.data_ptr(), self.rng_ctr.data_ptr()
        ss.dst_actions, ss.dst_actions_onehot, ss.dst_actions_inc = st["actions"].data_ptr(), st["actions_onehot"].data_ptr(), st["actions_inc"].data_ptr()
        if with_step_outputs:
            ss.reward, ss.clean_num, ss.apple_den, ss.terminated = (out["reward"].data_ptr(), out["clean_num"].data_ptr(),
                                                                      out["apple_den"].data_ptr(), out["terminated"].data_ptr())
            ss.dst_reward, ss.dst_clean_num, ss.dst_apple_den, ss.dst_terminated = (st["reward"].data_ptr(), st["clean_num"].data_ptr(),
                                                                                    st["apple_den"].data_ptr(), st["terminated"].data_ptr())
            ss.prev_actions, ss.prev_reward, ss.prev_actions_inc = (self.prev_actions.data_ptr(), self.prev_reward.data_ptr(),
                                                                    self.prev_inc.data_ptr())
            ss.ep_return = self.ep_return.data_ptr()
            if self.direct_obs:
                ss.next_t_out = self.t_dev.data_ptr()
        return ss

    def _select(self, store_env_step):
        """One controller evaluation on the current observation; with store_env_step also the env transition."""
        if self.fast is not None:
            return self._select_fast(store_env_step)
        a, mac, st = self.args, self.mac, self.store.data.transition_data
        td = self.t_dev
        obs, pos, orient = self.cur["obs"], self.cur["pos"], self.cur["orient"]
        st["obs"].index_copy_(1, td, obs.unsqueeze(1))
        st["agent_pos"].index_copy_(1, td, pos.unsqueeze(1))
        st["agent_orientation"].index_copy_(1, td, orient.unsqueeze(1))
        feat = mac.encode_obs(obs)
        inputs = mac.assemble_inputs(feat, self.prev_actions, self.prev_reward, self.prev_inc, pos, False)
        q_env, h_env, _ = mac.agent.forward_env(inputs, self.h_env)
        self.h_env.copy_(h_env)
        actions = self._pick(q_env, self.avail_mask, self.avail_idx, self.avail_idx.numel())      # [N, n]
        pos_t, orient_t = pos.clone(), orient.clone()                   # forward_inc sees the PRE-step pose (controller :78-82)
        if store_env_step:
            out = self.env.step_batch((actions % a.n_actions).to(th.int32), observe=True, fmt=self.obs_fmt)
            reward, clean, den = out["reward"], out["clean_num"], out["apple_den"]
            st["reward"].index_copy_(1, td, reward.unsqueeze(1))
            st["terminated"].index_copy_(1, td, out["terminated"].reshape(-1, 1, 1))
            st["clean_num"].index_copy_(1, td, clean.unsqueeze(1))
            st["apple_den"].index_copy_(1, td, den.unsqueeze(1))
            self.ep_return += reward
        else:   # slot T: the batch holds zeros for reward / clean_num / apple_den (episode_runner.py:108-119)
            reward = clean = den = th.zeros_like(self.prev_reward)
        onehot = F.one_hot(actions, num_classes=a.n_actions)
        st["actions"].index_copy_(1, td, actions.reshape(actions.shape[0], 1, -1, 1))
        st["actions_onehot"].index_copy_(1, td, onehot.float().unsqueeze(1))
        q_inc, h_inc, _ = mac.agent.forward_inc(inputs, self.h_inc, onehot, pos_t / mac.pos_scale, orient_t, reward.unsqueeze(-1),
                                                clean.unsqueeze(-1), den.unsqueeze(-1))
        self.h_inc.copy_(h_inc)
        actions_inc = self._pick(q_inc, None, None, a.n_inc_actions) * self.inc_mask              # [N, n, n]
        st["actions_inc"].index_copy_(1, td, actions_inc.reshape(actions_inc.shape[0], 1, actions_inc.shape[1], -1, 1))
        if store_env_step:
            self.prev_actions.copy_(actions)
            self.prev_reward.copy_(reward)
            self.prev_inc.copy_(actions_inc)
            td += 1

    # ---- runner surface ---------------------------------------------------------------------------------------------
    def begin_episode(self, test_mode=False):
        if not self._ready:
            self._allocate()
        self._test_mode = test_mode
        store = self._replay.reserve(self.batch_size) if (self._replay is not None and not test_mode) else None
        if store is None:
            if self._own_store is None:
                self._own_store = EpisodeBatch(self._scheme, self._groups, self.batch_size, self.episode_limit + 1,
                                               preprocess=self._preprocess, device=self.args.device)
            store = self._own_store
        self._bind_store(store)
        self.batch = self.store
        sel = self.mac.action_selector
        sel.epsilon = 0.0 if test_mode else sel.schedule.eval(self.sched_t)
        zero_after = getattr(self.args, "epsilon_zero", None)
        if zero_after is not None and self.t_env > zero_after:
            sel.epsilon = 0.0
        self.t = 0
        self._episodes += 1
        b = self._bundle
        # The episode's opening launches (env reset + first observation, the runner state's fills, the weight packs, the encoder of
        # slot 0) and its closing ones (slot-T pass, statistics) are ~25 small launches the host issues one by one while the GPU waits
        # for them (0.4 of a 7.5 ms iteration); from the third episode of a storage on they are two more hipGraph replays.
        if b.begin_graph is not None:
            self.eps.fill_(sel.epsilon)
            b.begin_graph.replay()
            return
        self._begin_launches()
        self.eps.fill_(sel.epsilon)
        if self._graph is None and self._episodes >= 2 and getattr(self.args, "rollout_graph", True):
            th.cuda.synchronize()
            g = th.cuda.CUDAGraph()
            # K consecutive timesteps per graph (the device-side time index makes every step of the replay land in its own
            # slot); K divides the episode length, so an episode is episode_limit / K replays
            K = self._graph_steps_planned
            self._graph_steps = K
            with th.no_grad(), th.cuda.graph(g, capture_error_mode="thread_local"):
                for k in range(K):
                    self._par = k & 1        # a replay starts at a multiple of K (even when pipelined)
                    self._select(True)
            self._graph = self._bundle.graph = g
            # capture records but does not run: re-establish the episode start state
            self.t_dev.zero_()
        elif (self._graph is not None and self.fast is not None and getattr(self.fast, "_pack_graph", None) is not None
              and getattr(self.args, "rollout_graph", True) and getattr(self.args, "episode_edge_graphs", True)):
            # third episode of this storage: everything is warm (the pack graph exists) -- capture the opening launches; they were
            # just run eagerly for THIS episode, the capture itself runs nothing
            th.cuda.synchronize()
            g = th.cuda.CUDAGraph()
            with th.no_grad(), th.cuda.graph(g, capture_error_mode="thread_local"):
                self._begin_launches(in_capture=True)
            b.begin_graph = g

    def _begin_launches(self, in_capture=False):
        """the device work that opens an episode on the bound storage (everything but the epsilon scalar, which the host computes)"""
        self.env.reset_batch()
        self.env.observe_batch(self.obs_fmt, out=self.cur if getattr(self, "direct_obs", False) else None)   # fills self.cur (or obs[:, 0])
        # the runner state an episode opens with, as ONE launch: time index, previous actions (-1) / reward / incentive actions, the
        # episode returns, the hidden states (the generic timestep's, or FastPolicy's own)
        hidden = [self.h_env, self.h_inc] if self.fast is None else [h for fp in self.fasts for h in (fp.h_env, fp.h_inc)]
        ops.fill_blocks([(self.t_dev, 0), (self.prev_actions, 0xFFFFFFFF), (self.prev_reward, 0), (self.prev_inc, 0), (self.ep_return, 0)]
                        + ([(self.recv_inc, 0)] if self.recv_inc is not None else []) + [(h, 0) for h in hidden])
        if self.fast is not None:
            if in_capture:
                self.fast._pack_eager()   # the learner may have stepped the weights since the last episode (packs are shared)
            else:
                self.fast.pack()
        if self.pipe:       # the encoder runs one timestep ahead of the heads: the observation of slot 0 is encoded here
            with th.no_grad():
                codes = self.store.data.transition_data["obs"] if self.obs_fmt == abi.OBS_CODE else self.cur["code"]
                self.fast.encode(None, codes=codes, slot_t=self.t_dev, buf=0)

    @th.no_grad()
    def step_once(self):
        if self.t >= self.episode_limit:
            raise RuntimeError("step_once() past episode_limit: the episode storage has no slot left (finish_episode / begin_episode first)")
        if self._graph is not None:
            if self.t % self._graph_steps == 0:       # one replay advances _graph_steps timesteps
                self._graph.replay()
        else:
            self._par = self.t & 1
            self._select(True)
        self.t += 1
        return self.t >= self.episode_limit

    @th.no_grad()
    def finish_episode(self):
        self._par = self.t & 1
        b = self._bundle
        self._out = dict(collective_return=self.env.native.out["collective_return"], equality=self.env.native.out["equality"])
        self._ep_return = self.ep_return
        stats_on = bool(getattr(self.args, "runner_stats", True))
        if b.finish_graph is not None and not self._test_mode:
            b.finish_graph.replay()                                     # the slot-T pass + this rollout's statistics
            self._stats_on_device_done = stats_on
            return self._finish_stats()
        self._select(False)
        if (b.begin_graph is not None and b.finish_graph is None and not self._test_mode and self._par == 0
                and getattr(self.args, "episode_edge_graphs", True)):
            # the closing launches of a training episode on this storage as a graph (captured after they ran eagerly for this episode)
            if stats_on:
                self._stat_acc(False)                                   # the accumulator exists before the capture
            th.cuda.synchronize()
            g = th.cuda.CUDAGraph()
            with th.cuda.graph(g, capture_error_mode="thread_local"):
                self._select(False)
                if stats_on:
                    self._stats_device(self._out, self._ep_return, False)
            b.finish_graph = g
        return self._finish_stats()
