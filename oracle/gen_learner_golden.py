"""Generate the learner / controller golden fixture FROM THE IMPORTED REFERENCE (build container only).

Follows run.py:84-132 with the reference's own EpisodeRunner, ReplayBuffer, HomophilyMAC and HomophilyLearner on the
reference env (Cleanup default5, 5 agents, short episodes), then records:
  * the sampled batch (all scheme keys the learner reads; obs as u8 = obs * 256),
  * the initial network weights (agent state_dict; the target net starts equal),
  * _build_inputs outputs for t = 0 and t = 3, q_env / q_inc of the unrolled controller,
  * learning_logs of two consecutive cal_loss_and_step calls and checksums of every parameter after each step.
pyclustering is absent: its two entry points are stubbed with the documented exact-value rule
(cluster id = 2 * rewards_t + clean_num_t), so `loss_sim` is parity-pinned only relative to that rule.
Output: tests/golden/learner_cleanup5.npz, learner_harvest5.npz, learner_cleanup5_w4.npz (the same at 4 x the initial weights)
"""
import contextlib
import io
import os
import random
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch as th
import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ref_harness as RH  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def install_cluster_stub():
    xm = sys.modules["pyclustering.cluster.xmeans"]
    ci = sys.modules["pyclustering.cluster.center_initializer"]

    class kmeans_plusplus_initializer:
        def __init__(self, sample, k):
            pass

        def initialize(self):
            return None

    class xmeans:
        def __init__(self, sample, centers, kmax):
            self.ids = (2 * sample[:, 0] + sample[:, 1]).numpy()

        def process(self):
            return self

        def get_clusters(self):
            return [np.nonzero(self.ids == v)[0].tolist() for v in np.unique(self.ids)]

    xm.xmeans = xmeans
    ci.kmeans_plusplus_initializer = kmeans_plusplus_initializer


def merge(d, u):
    for k, v in u.items():
        d[k] = merge(d.get(k, {}), v) if isinstance(v, dict) else v
    return d


def main(env_name, env_over, out_name, seed, wscale=1.0):
    OUT = os.path.join(GOLDEN, out_name)
    RH.import_reference()
    install_cluster_stub()
    cfg = {}
    for f in ("default.yaml", "envs/%s.yaml" % env_name, "algs/homophily.yaml"):
        merge(cfg, yaml.safe_load(open(os.path.join(RH.REF_SRC, "config", f))))
    merge(cfg, dict(env_args=env_over, batch_size=4, buffer_size=8, use_cuda=False, use_tensorboard=False, save_model=False))
    np.random.seed(seed); random.seed(seed); th.manual_seed(seed)
    args = SimpleNamespace(**cfg)
    args.device = "cpu"
    logger = SimpleNamespace(log_stat=lambda *a, **k: None, console_logger=SimpleNamespace(info=lambda *a: None))
    with contextlib.redirect_stdout(io.StringIO()):
        from runners import REGISTRY as r_REGISTRY
        from controllers import REGISTRY as mac_REGISTRY
        from learners import REGISTRY as le_REGISTRY
        from components.episode_buffer import ReplayBuffer
        from components.transforms import OneHot
        runner = r_REGISTRY[args.runner](args=args, logger=logger)
    env_info = runner.get_env_info()
    args.n_agents, args.n_actions = env_info["n_agents"], env_info["n_actions"]
    args.state_shape, args.obs_shape = env_info["state_shape"], env_info["obs_shape"]
    args.state_dims, args.obs_dims = env_info["state_dims"], env_info["obs_dims"]
    scheme = {
        "state": {"vshape": env_info["state_shape"]}, "obs": {"vshape": env_info["obs_shape"], "group": "agents"},
        "actions": {"vshape": (1,), "group": "agents", "dtype": th.long},
        "avail_actions": {"vshape": (env_info["n_actions"],), "group": "agents", "dtype": th.int},
        "reward": {"vshape": (args.n_agents,)}, "terminated": {"vshape": (1,), "dtype": th.uint8},
        "clean_num": {"vshape": (args.n_agents,)}, "apple_den": {"vshape": (args.n_agents,)},
        "agent_pos": {"vshape": (args.n_agents, 2)}, "agent_orientation": {"vshape": (args.n_agents, 2)},
        "actions_inc": {"vshape": (args.n_agents, 1), "group": "agents", "dtype": th.long},
    }
    groups = {"agents": args.n_agents}
    preprocess = {"actions": ("actions_onehot", [OneHot(out_dim=args.n_actions)])}
    buffer = ReplayBuffer(scheme, groups, args.buffer_size, env_info["episode_limit"] + 1, preprocess=preprocess, device="cpu")
    mac = mac_REGISTRY[args.mac](buffer.scheme, groups, args)
    if wscale != 1.0:
        # "trained-magnitude" fixture: every parameter of the freshly initialised reference controller times wscale before anything
        # is computed (the soak runs' parameter norm grows 47 -> 155 over 8 000 rollouts: x 3.3), so that the reference's Q-values,
        # losses and steps are recorded at the magnitudes a trained network has
        with th.no_grad():
            for prm in mac.parameters():
                prm.mul_(wscale)
    runner.setup(scheme=scheme, groups=groups, preprocess=preprocess, mac=mac)
    learner = le_REGISTRY[args.learner](mac, buffer.scheme, logger, args)
    init_sd = {k: v.detach().clone().numpy() for k, v in mac.agent.state_dict().items()}
    # make the rollout interesting: agents start near the waste so that rewards / clean_num are not all zero
    for ep in range(args.batch_size):
        buffer.insert_episode_batch(runner.run(test_mode=False))
    batch = buffer.sample(args.batch_size)
    batch = batch[:, :batch.max_t_filled()]
    out = {}
    for k in ("obs", "actions", "actions_inc", "reward", "terminated", "clean_num", "apple_den", "agent_pos", "agent_orientation",
              "avail_actions", "filled"):
        v = batch[k].numpy()
        if k == "obs":
            v8 = np.round(v * 256)
            assert (v8 / 256 == v).all()
            v = v8.astype(np.uint8)
        out["batch_" + k] = v
    for k, v in init_sd.items():
        out["w_" + k] = v
    with th.no_grad():
        out["inputs_t0"] = mac._build_inputs(batch, 0).numpy()
        out["inputs_t3"] = mac._build_inputs(batch, 3).numpy()
        mac.init_hidden(batch.batch_size)
        qe, qi = [], []
        for t in range(batch.max_seq_length):
            a, b, _ = mac.forward(batch, t=t)
            qe.append(a.numpy().copy()); qi.append(b.numpy().copy())
        out["q_env"] = np.stack(qe, 1); out["q_inc"] = np.stack(qi, 1)
    for step in range(2):
        logs = learner.cal_loss_and_step(batch)
        for k, v in logs.items():
            out["step%d_%s" % (step, k)] = np.float64(v.item())
        names, sums, sqs, heads = [], [], [], []
        for k, v in mac.agent.state_dict().items():
            x = v.detach().double().reshape(-1)
            names.append(k); sums.append(x.sum().item()); sqs.append((x * x).sum().item()); heads.append(np.resize(x[:5].numpy(), 5))
        out["step%d_param_sum" % step] = np.array(sums); out["step%d_param_sq" % step] = np.array(sqs)
        out["step%d_param_head" % step] = np.stack(heads)
        out["param_names"] = np.array(names)
    out["meta_seq_len"] = np.int64(batch.max_seq_length)
    import json
    out["meta"] = np.frombuffer(json.dumps(dict(env=env_name, env_args=cfg["env_args"], wscale=wscale)).encode(), np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB;",
          {k[6:]: float(out[k]) for k in out if k.startswith("step0_loss") or k.startswith("step1_loss")},
          "reward sum", float(out["batch_reward"].sum()), "clean", float(out["batch_clean_num"].sum()))


if __name__ == "__main__":
    if "--w4-only" not in sys.argv:
        main("cleanup", dict(num_agents=5, map="default5", episode_limit=12), "learner_cleanup5.npz", 3)
        main("harvest", dict(num_agents=5, map="default10", episode_limit=12, view_size=7), "learner_harvest5.npz", 4)
    main("cleanup", dict(num_agents=5, map="default5", episode_limit=12), "learner_cleanup5_w4.npz", 5, wscale=4.0)
