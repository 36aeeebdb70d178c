"""Generate a REFERENCE-WRITTEN checkpoint fixture (build container only; imports /root/reference like gen_learner_golden.py).

The reference's own HomophilyLearner.save_models (homophily_learner.py:276-279 -> agent.th, opt_env.th, opt_inc.th via th.save)
is called after two optimisation steps on a small configuration (Cleanup default3, 3 agents, view_size 3, rnn_hidden_dim 16: the
files stay a few hundred KB), then a THIRD step is taken and its losses and parameter checksums are recorded.  The test
(tests/test_checkpoint.py) loads the three files with this package's load_models (weights_only=True: tensors and plain containers
only), replays the third step on the recorded batch and must land on the reference's numbers -- this pins the file names, the
state_dict keys, the optimiser-state layout (param_groups / state indexing, both Adam moments, step counters) and the resume
arithmetic against the reference itself.
Output: tests/golden/ckpt_cleanup3/{agent.th,opt_env.th,opt_inc.th,ckpt.npz}
"""
import contextlib
import io
import json
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import torch as th
import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ref_harness as RH  # noqa: E402
from oracle.gen_learner_golden import GOLDEN, install_cluster_stub, merge  # noqa: E402


def main(seed=5):
    out_dir = os.path.join(GOLDEN, "ckpt_cleanup3")
    os.makedirs(out_dir, exist_ok=True)
    RH.import_reference()
    install_cluster_stub()
    cfg = {}
    for f in ("default.yaml", "envs/cleanup.yaml", "algs/homophily.yaml"):
        merge(cfg, yaml.safe_load(open(os.path.join(RH.REF_SRC, "config", f))))
    env_over = dict(num_agents=3, map="default3", episode_limit=10, view_size=3)
    over = dict(env_args=env_over, batch_size=4, buffer_size=8, use_cuda=False, use_tensorboard=False, save_model=False, rnn_hidden_dim=16)
    merge(cfg, over)
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
    runner.setup(scheme=scheme, groups=groups, preprocess=preprocess, mac=mac)
    learner = le_REGISTRY[args.learner](mac, buffer.scheme, logger, args)
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
    for step in range(2):
        learner.cal_loss_and_step(batch)
    learner.save_models(out_dir)                  # the reference's own th.save calls: agent.th, opt_env.th, opt_inc.th
    # the target network of a resumed run is loaded from the same agent.th (homophily_learner.py:281-288): do the same here so
    # that the recorded third step is what a resume produces
    learner.load_models(out_dir)
    logs = learner.cal_loss_and_step(batch)
    for k, v in logs.items():
        out["step2_" + k] = np.float64(v.item())
    names, sums, sqs = [], [], []
    for k, v in mac.agent.state_dict().items():
        x = v.detach().double().reshape(-1)
        names.append(k); sums.append(x.sum().item()); sqs.append((x * x).sum().item())
    out["step2_param_sum"], out["step2_param_sq"], out["param_names"] = np.array(sums), np.array(sqs), np.array(names)
    out["meta"] = np.frombuffer(json.dumps(dict(env="cleanup", env_args=cfg["env_args"], overrides=dict(rnn_hidden_dim=16))).encode(), np.uint8)
    np.savez_compressed(os.path.join(out_dir, "ckpt.npz"), **out)
    print("wrote", out_dir, {f: os.path.getsize(os.path.join(out_dir, f)) for f in sorted(os.listdir(out_dir))},
          {k: float(v) for k, v in out.items() if k.startswith("step2_loss")})


if __name__ == "__main__":
    main()
