"""FastPolicy: rollout-time (no-grad) evaluation of the homophily controller for N vectorised envs.

Computes what HomophilyMAC.select_actions_env / select_actions_inc compute (homophily_controller.py:30-65 on top of
homophily_agent.py:154-208) with agent-major activations [n, N, 64] and the weights re-packed once per episode:
  * fused path (default; csrc/ssd_policy_mfma.hip): k_encode (conv + Linear encoder as banded GEMMs on the 16-bit matrix cores,
    reading the u8 class codes the env kernel emits) and ONE launch per head -- k_head<env> (input tail + fc1 + GRU + dueling +
    epsilon-greedy) and k_head<inc> (the same plus the per-pair term) -- which also file their results in the episode storage.
    `precision` 2 (default): every f32 product as three f16 MFMA products of two-term splits (f32-equivalent, the reference's dtype);
    1: single bf16 products (BASELINE.json's "bf16 Q-net" variant; rollout inference only, the learner stays fp32);
  * per-layer path (fused=False, or window sizes / palettes without a fused encoder): every per-agent layer as one batched GEMM
    (hipBLASLt) between the small kernels of csrc/ssd_policy.hip; the incentive head's pairwise layer [h_i | other_j] @ W is
    split into h_i @ W_h + other_j @ W_o, so the [n, N * n, H + E] concatenation is never materialised.
Both implement the shipped _build_inputs flag set (config/default.yaml:45-51).
Action RNG: the package's counter generator (not torch's Philox) keyed by the GLOBAL env id (env_id_base + local env), so env
shards draw what the unsharded job draws; exploration draws are not parity-pinned against torch.
"""
import ctypes as C

import torch as th
import torch.nn.functional as F

from . import abi


class FastPolicy:
    def __init__(self, mac, n_env, avail_mask_u8, seed=0, actions_out=None, actions_inc_out=None, share_packs_from=None, fused=True,
                 precision=2, env_id_base=0):
        self.mac, self.agent, self.a = mac, mac.agent, mac.args
        a = self.a
        assert a.rgb_input and a.conv_out == 6 and a.obs_dim_net == 32 and a.conv_kernel == 3 and a.conv_stride == 1
        assert precision in (1, 2)
        self.lib = abi.load_library()
        self.N, self.n, self.H, self.A = n_env, mac.n_agents, a.rnn_hidden_dim, a.n_actions
        self.dev = next(self.agent.parameters()).device
        self.inp = mac.input_shape
        self.V = int(a.obs_dims[0])
        self.precision, self.env_id_base = int(precision), int(env_id_base) & 0xFFFFFFFF
        n, N, H = self.n, self.N, self.H
        f32 = dict(dtype=th.float32, device=self.dev)
        # fused: one launch per head (csrc/ssd_policy_mfma.hip), inputs padded to 64 columns; otherwise the per-layer
        # composition below (batched hipBLASLt GEMMs + the small kernels of csrc/ssd_policy.hip)
        self.fused = bool(fused) and H == 64 and self.inp + self.A <= 64 and self.A + 7 <= 16 and mac.input_flags is not None
        # the per-layer composition assembles the shipped input layout only (ssd_build_inputs)
        assert self.fused or mac.shipped_flags, "FastPolicy: this _build_inputs flag set needs the fused heads (FastPolicy.supports)"
        # the fused encoder exists for 15 x 15 and 31 x 31 windows (view_size 7 / 15: the shipped configurations)
        self.fused_enc = self.fused and self.V in (15, 31) and tuple(a.obs_dims) == (self.V, self.V)
        self.bands = abi.encode_bands(self.V) if self.fused_enc else 1
        # encoder images: the class-LUT layout (conv as a table sum, no conv MFMAs: include/ssd_hip.h SSD_ENCODE_LAYOUT_LUT) unless
        # enc_layout / SSD_ENC_LAYOUT asks for round 3's Toeplitz fragments (kept as the cross-check and for the training forward)
        import os
        lay = getattr(a, "enc_layout", None) or os.environ.get("SSD_ENC_LAYOUT", "lut")
        self.enc_layout = abi.ENCODE_LAYOUT_TOEPLITZ if str(lay).lower() in ("toeplitz", "0") else abi.ENCODE_LAYOUT_LUT
        # bf16 MFMA products an f32-equivalent product costs (bench.py's roofline accounting); conv: the planes are exact, 2
        self.n_products = dict(encode_conv=2, encode_lin=3, head_env=3, head_inc=3) if precision == 2 else \
            dict(encode_conv=1, encode_lin=1, head_env=1, head_inc=1)
        if self.fused_enc and self.enc_layout == abi.ENCODE_LAYOUT_LUT:
            self.n_products["encode_conv"] = 0      # the conv is a table sum on the vector unit: no matrix-core products

        # [feat | tail (| 0)], agent-major; a PAIR of buffers: the pipelined rollout (act_inc_encode) encodes timestep t + 1 into
        # the other buffer while the inc head still reads the rows of t.  Everything else uses buffer 0 (`inputs`).
        self.inputs_pair = th.zeros(2, n, N, 64 if self.fused else self.inp, **f32)
        self.inputs = self.inputs_pair[0]
        self.feat_part = th.zeros(self.bands, n * N, 32, **f32) if self.bands > 1 else None
        self.h_env = th.zeros(n, N, H, **f32)
        self.h_inc = th.zeros(n, N, H, **f32)
        # outputs may be slices of a caller-owned full-batch buffer (env groups evaluated on separate streams)
        self.actions = th.zeros(N, n, dtype=th.long, device=self.dev) if actions_out is None else actions_out
        self.actions_inc = th.zeros(N, n, n, dtype=th.long, device=self.dev) if actions_inc_out is None else actions_inc_out
        assert self.actions.is_contiguous() and self.actions_inc.is_contiguous()
        self.avail = avail_mask_u8.to(device=self.dev, dtype=th.uint8).contiguous()
        self._avail_bits = 0x80000000 | sum(1 << k for k, v in enumerate(avail_mask_u8.detach().cpu().reshape(-1).tolist()[:31]) if v)   # ssd_policy_head.avail_bits
        self.seed = seed & 0xFFFFFFFF
        self.arange_n = th.arange(n, device=self.dev).unsqueeze(1)
        if share_packs_from is not None:
            self.p = share_packs_from.p          # same weights: one packed copy serves every group
        else:
            self.pack()

    def _stream(self):
        return th.cuda.current_stream(self.dev).cuda_stream

    @th.no_grad()
    def pack(self):
        """Snapshot the (possibly just trained) weights into kernel-ready packs; values are written in place so that a captured
        graph keeps seeing the same addresses.  Fused path: the fragment images of the encoder and one image per agent and head,
        built on the device by ssd_policy_pack_encoder / ssd_policy_pack_head (3 launches); per-layer path: GEMM-ready operands.
        The launches are themselves replayed as one hipGraph from the third call on (parameters and packs live at fixed addresses;
        the optimisers update in place)."""
        self._pack_calls = getattr(self, "_pack_calls", 0) + 1
        g = getattr(self, "_pack_graph", None)
        if g is not None:
            g.replay()
            return
        if self._pack_calls == 3 and self.dev.type == "cuda" and not th.cuda.is_current_stream_capturing():
            th.cuda.synchronize()
            g = th.cuda.CUDAGraph()
            with th.cuda.graph(g, capture_error_mode="thread_local"):
                self._pack_eager()
            self._pack_graph = g
            g.replay()
            return
        self._pack_eager()

    @th.no_grad()
    def _pack_eager(self):
        ag, H, A = self.agent, self.H, self.A
        w, b = ag._w, ag._b
        lin = ag.conv_to_fc[3].weight
        packs = dict(cw=ag.conv_to_fc[0].weight, cb=ag.conv_to_fc[0].bias, lb=ag.conv_to_fc[3].bias)
        if not self.fused_enc:
            packs["lw_t"] = lin.t()
        if not self.fused:
            packs.update(
                w1e=w("fc1_env_w"), b1e=b("fc1_env_b"),
                w2e=th.cat([w("fc2_env_w"), w("fc2_env_v_w")], dim=2), b2e=th.cat([b("fc2_env_b"), b("fc2_env_v_b")], dim=2),
                w1i_x=w("fc1_inc_w")[:, :self.inp], w1i_a=w("fc1_inc_w")[:, self.inp:], b1i=b("fc1_inc_b"))
            packs["wie"], packs["whe"], packs["bie"], packs["bhe"] = ag._gru_weights("env")
            packs["wii"], packs["whi"], packs["bii"], packs["bhi"] = ag._gru_weights("inc")
            w2i = th.cat([w("fc2_inc_w"), w("fc2_inc_v_w")], dim=2)                       # [n, H + E, 4]
            packs["w2i_h"] = w2i[:, :H]
            packs["w2i_o"] = w2i[:, H:].permute(1, 0, 2).reshape(w2i.shape[1] - H, -1)    # [E, n(i) * 4]
            packs["b2i"] = th.cat([b("fc2_inc_b"), b("fc2_inc_v_b")], dim=2).unsqueeze(2)  # [n, 1, 1, 4]
        if not hasattr(self, "p"):
            self.p = {k: v.detach().clone().contiguous() for k, v in packs.items()}
            u8 = dict(dtype=th.uint8, device=self.dev)
            if self.fused:
                for head in ("env", "inc"):
                    self.p["img_" + head] = th.zeros(self.n, abi.policy_image_bytes(self.precision), **u8)
            if self.fused_enc:
                cbytes, lbytes = abi.encode_frag_bytes(self.V, self.precision, self.enc_layout)
                self.p["conv_frags"], self.p["lin_frags"] = th.zeros(cbytes, **u8), th.zeros(lbytes, **u8)
        else:
            for k, v in packs.items():
                self.p[k].copy_(v)
        st = self._stream()
        if self.fused:
            for head in ("env", "inc"):
                hp = self._head_params(head)
                abi.check(self.lib, self.lib.ssd_policy_pack_head(C.byref(hp), self.precision, self.p["img_" + head].data_ptr(), st))
        if self.fused_enc:
            pack = self.lib.ssd_policy_pack_encoder_lut if self.enc_layout == abi.ENCODE_LAYOUT_LUT else self.lib.ssd_policy_pack_encoder
            abi.check(self.lib, pack(ag.conv_to_fc[0].weight.data_ptr(), ag.conv_to_fc[0].bias.data_ptr(), lin.data_ptr(), self.V, self.precision,
                                     self.p["conv_frags"].data_ptr(), self.p["lin_frags"].data_ptr(), st))

    @staticmethod
    def supports(mac, fused=True):
        """Whether the rollout kernels build this controller's input layout: the shipped flag set on either path, any other
        combination of the _build_inputs flags (homophily_controller.py:137-184) on the fused heads as long as the inputs (+ the inc
        head's one-hot action) fit the 64-column weight image -- obs_others_last_action never does."""
        a = mac.args
        if mac.shipped_flags:
            return True
        return bool(fused) and mac.input_flags is not None and a.rnn_hidden_dim == 64 and mac.input_shape + a.n_actions <= 64 \
            and a.n_actions + 7 <= 16

    def _head_params(self, head):
        """ssd_policy_head_params of one head: pointers to the reference-shaped parameters (homophily_agent.py:37-125)."""
        ag = self.agent
        g = lambda name: getattr(ag, name)
        for t in (g("fc1_%s_w" % head), g("fc2_%s_w" % head)):
            assert t.is_contiguous() and t.dtype == th.float32
        hp = abi.SsdPolicyHeadParams()
        hp.fc1_w, hp.fc1_b = g("fc1_%s_w" % head).data_ptr(), g("fc1_%s_b" % head).data_ptr()
        for k, gate in enumerate("rzn"):
            hp.w_i[k], hp.w_h[k] = g("rnn_%s_i%s_w" % (head, gate)).data_ptr(), g("rnn_%s_h%s_w" % (head, gate)).data_ptr()
            hp.b_i[k], hp.b_h[k] = g("rnn_%s_i%s_b" % (head, gate)).data_ptr(), g("rnn_%s_h%s_b" % (head, gate)).data_ptr()
        hp.fc2_w, hp.fc2_b = g("fc2_%s_w" % head).data_ptr(), g("fc2_%s_b" % head).data_ptr()
        hp.fc2_v_w, hp.fc2_v_b = g("fc2_%s_v_w" % head).data_ptr(), g("fc2_%s_v_b" % head).data_ptr()
        hp.n_agents, hp.fc1_in = self.n, g("fc1_%s_w" % head).shape[2]
        hp.fc2_in, hp.fc2_out = g("fc2_%s_w" % head).shape[2], g("fc2_%s_w" % head).shape[3]
        return hp

    def _head_args(self, inc, eps, step, q_out=None, buf=0):
        a = abi.SsdPolicyHead()
        a.n_env, a.n_agents, a.n_actions, a.input_shape = self.N, self.n, self.A, self.inp
        a.pos_scale = float(self.mac.pos_scale)
        a.seed = (self.seed ^ 0x5bd1e995) if inc else self.seed
        a.inputs = self.inputs_pair[buf].data_ptr()
        a.h = (self.h_inc if inc else self.h_env).data_ptr()
        a.weights = self.p["img_inc" if inc else "img_env"].data_ptr()
        a.epsilon, a.step = eps.data_ptr(), step.data_ptr()
        a.q_out = None if q_out is None else q_out.data_ptr()
        a.precision, a.env_id_base = self.precision, self.env_id_base
        a.input_flags = abi.INPUT_EXPLICIT | int(self.mac.input_flags)
        return a

    def reset(self):
        self.h_env.zero_(); self.h_inc.zero_()

    # ---- env head -----------------------------------------------------------------------------------------------
    @th.no_grad()
    def act_env(self, obs, prev_actions, prev_reward, prev_inc, pos, eps, step, codes=None, slot_t=None, store_obs=None, store_t=None,
                q_out=None, orient=None, actions_i32=None, pos_copy=None, orient_copy=None, t_copy=None, counter_inc=None, file=None,
                mask_alphabet=None):
        """encode() followed by head_env(); returns actions i64 [N, n] (static buffer).  See the two methods for the arguments."""
        self.encode(obs, codes=codes, slot_t=slot_t, store_obs=store_obs, store_t=store_t, t_copy=t_copy, counter_inc=counter_inc,
                    mask_alphabet=mask_alphabet)
        return self.head_env(prev_actions, prev_reward, prev_inc, pos, eps, step, q_out=q_out, orient=orient, actions_i32=actions_i32,
                             pos_copy=pos_copy, orient_copy=orient_copy, file=file)

    @staticmethod
    def codes_from_obs(obs):
        """u8 class codes [N, n, stride] (SSD_OBS_CODE alphabet; stride = V * V rounded up to 16) of a simplified-palette observation
        f32 [N, n, 3, V, V]: waste = R -> 2, apple = G -> 1, wall / agent = B -> 3 (cleanup.py:93-105).  For callers that hold no
        code buffer (tests); the runner takes the codes the env kernel emits (ssd_obs_out.obs_code)."""
        N, n, _, V, _ = obs.shape
        c = ((obs[:, :, 0] > 0) * 2 + (obs[:, :, 1] > 0) * 1 + (obs[:, :, 2] > 0) * 3).to(th.uint8).reshape(N, n, V * V)
        return F.pad(c, (0, abi.code_agent_stride(V) - V * V)).contiguous()

    def _encode_args(self, obs, codes, slot_t, mask_alphabet, buf=0, slot_add=0, t_copy=None, counter_inc=None):
        """ssd_policy_encode_args of the fused encoder (see encode) writing inputs_pair[buf] / the band sums; returns (args, codes)."""
        p, n, N, V = self.p, self.n, self.N, self.V
        if codes is None:
            codes, mask_alphabet = self.codes_from_obs(obs), False
        if mask_alphabet is None:
            mask_alphabet = codes.dim() == 3
        assert codes.dtype == th.uint8 and codes.shape[0] == N
        ea = abi.SsdPolicyEncodeArgs()
        ea.codes = codes.data_ptr()
        ea.code_bytes = codes.untyped_storage().nbytes() - codes.storage_offset()
        if codes.dim() == 5:        # an episode storage [N, T+1, n, V, V], time slot *slot_t (+ slot_add)
            assert slot_t is not None and codes[0, 0].is_contiguous()
            ea.env_stride, ea.slot_stride, ea.agent_stride = codes.stride(0), codes.stride(1), V * V
        else:                       # the dense side buffer [N, n, stride]
            assert codes.dim() == 3 and codes[0].is_contiguous()
            ea.env_stride, ea.slot_stride, ea.agent_stride = codes.stride(0), 0, codes.stride(1)
        ea.slot_t = None if slot_t is None else slot_t.data_ptr()
        ea.slot_add = int(slot_add) if slot_t is not None else 0
        ea.rows, ea.view_edge, ea.n_agents, ea.agent_major, ea.precision = N * n, V, n, 1, self.precision
        ea.alphabet = abi.CODE_CHANNEL_MASK if mask_alphabet else abi.CODE_CLASS
        ea.conv_frags, ea.lin_frags = p["conv_frags"].data_ptr(), p["lin_frags"].data_ptr()
        ea.conv_b, ea.lin_b = p["cb"].data_ptr(), p["lb"].data_ptr()
        ea.layout = self.enc_layout
        if self.bands > 1:
            ea.part = self.feat_part.data_ptr()
        else:
            ea.out, ea.out_stride = self.inputs_pair[buf].data_ptr(), self.inputs_pair.shape[-1]
        ea.slot_t_copy = None if t_copy is None else t_copy.data_ptr()
        ea.counter_inc = None if counter_inc is None else counter_inc.data_ptr()
        return ea, codes

    @th.no_grad()
    def encode(self, obs, codes=None, slot_t=None, store_obs=None, store_t=None, t_copy=None, counter_inc=None, mask_alphabet=None, buf=0):
        """rgb_preprocess (homophily_agent.py:20-27,213-214) of the current observation into columns 0..31 of `inputs` (31 x 31
        windows: into the per-band partial sums that head_env finishes).
        Fused encoder: reads one byte per window cell -- `codes` [N, n, stride] = the env's obs_code side buffer (channel masks;
        mask_alphabet defaults to True for this layout) or an episode storage of class codes u8 [N, T+1, n, V, V] read at the
        device time index slot_t (SSD_OBS_CODE classes); with codes = None, class codes are derived from obs f32 [N, n, 3, V, V].
        Per-layer path: obs f32.  store_obs / store_t: copy obs into store_obs[:, t] (callers whose env does not write the storage
        itself).  t_copy receives *slot_t; counter_inc is incremented (device scalars for the kernels that follow)."""
        p, lib, n, N = self.p, self.lib, self.n, self.N
        st = self._stream()
        if self.fused_enc:
            ea, codes = self._encode_args(obs, codes, slot_t, mask_alphabet, buf=buf, t_copy=t_copy, counter_inc=counter_inc)
            abi.check(lib, lib.ssd_policy_encode(C.byref(ea), st))
            self._keep_codes = codes
            if store_obs is not None and obs is not None:
                store_obs.index_copy_(1, store_t, obs.unsqueeze(1))
            return
        assert buf == 0
        V = obs.shape[-1]
        so = (None if store_obs is None else store_obs.data_ptr(), 0 if store_obs is None else store_obs.stride(0),
              None if store_t is None else store_t.data_ptr())
        K = 6 * (V - 2) * (V - 2)
        if getattr(self, "_conv", None) is None or self._conv.shape[1] != K:
            self._conv = th.empty(n * N, K, dtype=th.float32, device=self.dev)           # agent-major rows
        abi.check(lib, lib.ssd_conv_leaky(obs.data_ptr(), N * n, V, 6, p["cw"].data_ptr(), p["cb"].data_ptr(), self._conv.data_ptr(), n, 1,
                                          so[0], so[1], so[2], st))
        feat = self.inputs.view(n * N, self.inputs.shape[-1])[:, :32]                    # Linear + LeakyReLU straight into the input matrix
        th.addmm(p["lb"], self._conv, p["lw_t"], out=feat)
        F.leaky_relu_(feat)

    @th.no_grad()
    def head_env(self, prev_actions, prev_reward, prev_inc, pos, eps, step, q_out=None, orient=None, actions_i32=None, pos_copy=None,
                 orient_copy=None, file=None, buf=0):
        """input tail + fc1 + GRU + dueling + epsilon-greedy of the env head on the features encode() left in `inputs`."""
        p, lib, n, N, H = self.p, self.lib, self.n, self.N, self.H
        st = self._stream()
        if self.fused:
            ha = self._head_args(False, eps, step, q_out, buf)
            ha.avail = self.avail.data_ptr()
            ha.avail_bits = self._avail_bits       # the mask itself (a constant of the env class): the kernel issues no loads for it
            ha.prev_actions, ha.prev_reward, ha.prev_actions_inc, ha.pos = (prev_actions.data_ptr(), prev_reward.data_ptr(),
                                                                          prev_inc.data_ptr(), pos.data_ptr())
            ha.out_actions = self.actions.data_ptr()
            if actions_i32 is not None:
                ha.out_actions_i32 = actions_i32.data_ptr()
            if pos_copy is not None:
                ha.orient, ha.pos_copy, ha.orient_copy = orient.data_ptr(), pos_copy.data_ptr(), orient_copy.data_ptr()
            if self.fused_enc and self.bands > 1:      # the encoder left per-band partial sums: the head finishes the features
                ha.feat_part, ha.feat_bands, ha.lin_b = self.feat_part.data_ptr(), self.bands, p["lb"].data_ptr()
            for k, v in (file or {}).items():
                setattr(ha, k, v)
            abi.check(lib, lib.ssd_policy_head_env(C.byref(ha), st))
            return self.actions
        abi.check(lib, lib.ssd_build_inputs(N, n, self.A, 2, prev_actions.data_ptr(), prev_reward.data_ptr(), prev_inc.data_ptr(),
                                            pos.data_ptr(), float(self.mac.pos_scale), self.inputs.data_ptr(), self.inp, 32, st))
        x = F.leaky_relu(th.baddbmm(p["b1e"], self.inputs, p["w1e"]))
        gi = th.baddbmm(p["bie"], x, p["wie"])
        gh = th.baddbmm(p["bhe"], self.h_env, p["whe"])
        abi.check(lib, lib.ssd_gru_gates(gi.data_ptr(), gh.data_ptr(), self.h_env.data_ptr(), n * N, H, st))
        av = th.baddbmm(p["b2e"], self.h_env, p["w2e"])                                # [n, N, A + 1]
        abi.check(lib, lib.ssd_dueling_pick(av.data_ptr(), n * N, self.A, self.avail.data_ptr(), eps.data_ptr(), step.data_ptr(),
                                            self.seed, n, N, 0, self.actions.data_ptr(), None if q_out is None else q_out.data_ptr(),
                                            self.env_id_base, st))
        self._keep = (x, gi, gh, av)
        return self.actions

    # ---- incentive head ---------------------------------------------------------------------------------------------
    @th.no_grad()
    def _inc_args(self, actions, pos, orient, reward, clean_num, apple_den, eps, step, q_out=None, file=None, buf=0):
        ha = self._head_args(True, eps, step, q_out, buf)
        ha.actions, ha.pos_pre, ha.orient_pre = actions.data_ptr(), pos.data_ptr(), orient.data_ptr()
        ha.reward, ha.clean_num, ha.apple_den = reward.data_ptr(), clean_num.data_ptr(), apple_den.data_ptr()
        ha.out_actions = self.actions_inc.data_ptr()
        for k, v in (file or {}).items():
            setattr(ha, k, v)
        return ha

    @th.no_grad()
    def act_inc_encode(self, actions, pos, orient, reward, clean_num, apple_den, eps, step, codes, slot_t=None, slot_add=0, buf=0,
                       q_out=None, file=None, mask_alphabet=None):
        """act_inc of timestep t on inputs_pair[buf] AND encode of timestep t + 1 into inputs_pair[buf ^ 1] (31 x 31 windows: into
        the band sums) as ONE launch (ssd_policy_head_inc_encode): the pipelined rollout's third launch of a timestep.  `codes` /
        slot_t / slot_add as in encode(): the observation the env step of t just produced (storage slot *slot_t + slot_add)."""
        assert self.fused and self.fused_enc
        ha = self._inc_args(actions, pos, orient, reward, clean_num, apple_den, eps, step, q_out=q_out, file=file, buf=buf)
        ea, codes = self._encode_args(None, codes, slot_t, mask_alphabet, buf=buf ^ 1, slot_add=slot_add)
        abi.check(self.lib, self.lib.ssd_policy_head_inc_encode(C.byref(ha), C.byref(ea), self._stream()))
        self._keep_codes = codes
        return self.actions_inc

    @th.no_grad()
    def act_inc(self, actions, pos, orient, reward, clean_num, apple_den, eps, step, q_out=None, file=None, buf=0):
        """actions i64 [N, n] (the env actions just taken); pos / orient: the PRE-step pose [N, n, 2]; reward, clean_num,
        apple_den [N, n] of this step.  Returns actions_inc i64 [N, n, n] with a zero diagonal (static buffer)."""
        p, lib, n, N, H = self.p, self.lib, self.n, self.N, self.H
        st = self._stream()
        if self.fused:
            ha = self._inc_args(actions, pos, orient, reward, clean_num, apple_den, eps, step, q_out=q_out, file=file, buf=buf)
            abi.check(lib, lib.ssd_policy_head_inc(C.byref(ha), st))
            return self.actions_inc
        assert buf == 0
        x = th.baddbmm(p["b1i"], self.inputs, p["w1i_x"]) + p["w1i_a"][self.arange_n, actions.t()]    # one-hot(a) @ W_a = row gather
        x = F.leaky_relu(x)
        gi = th.baddbmm(p["bii"], x, p["wii"])
        gh = th.baddbmm(p["bhi"], self.h_inc, p["whi"])
        abi.check(lib, lib.ssd_gru_gates(gi.data_ptr(), gh.data_ptr(), self.h_inc.data_ptr(), n * N, H, st))
        hpart = th.bmm(self.h_inc, p["w2i_h"])                                          # [n(i), N, 4]
        other = th.cat([F.one_hot(actions, self.A).float(), pos / self.mac.pos_scale, orient, reward.unsqueeze(-1),
                        clean_num.unsqueeze(-1), apple_den.unsqueeze(-1)], dim=-1)       # [N, n(j), E]
        opart = (other.reshape(N * n, -1) @ p["w2i_o"]).reshape(N, n, n, 4).permute(2, 0, 1, 3)   # [n(i), N, n(j), 4]
        av = (hpart.unsqueeze(2) + opart + p["b2i"]).contiguous()                        # [n(i), N, n(j), 4]
        abi.check(lib, lib.ssd_dueling_pick(av.data_ptr(), n * N * n, self.a.n_inc_actions, None, eps.data_ptr(), step.data_ptr(),
                                            self.seed ^ 0x5bd1e995, n, N, 1, self.actions_inc.data_ptr(), None if q_out is None else q_out.data_ptr(),
                                            self.env_id_base, st))
        self._keep2 = (x, gi, gh, av)
        return self.actions_inc

    # ---- reference-shaped Q access (tests) ---------------------------------------------------------------------------
    @th.no_grad()
    def q_values(self, av):
        a = av[..., :-1]
        return av[..., -1:] + a - a.mean(dim=-1, keepdim=True)
