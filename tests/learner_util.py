"""Rebuild a batch / controller / learner of THIS package from the learner golden fixtures (tests/golden/learner_*.npz,
generated from the reference by oracle/gen_learner_golden.py)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch as th

from homophily_marl_amd.components.episode_buffer import EpisodeBatch
from homophily_marl_amd.components.transforms import OneHot
from homophily_marl_amd.controllers import REGISTRY as mac_REGISTRY
from homophily_marl_amd.learners import REGISTRY as le_REGISTRY
from homophily_marl_amd.run import load_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def build(z, meta, device="cpu", sl=slice(None), overrides=None, code_obs=False):
    """Returns (args, batch, mac, learner) with the fixture's initial weights (random-init when the fixture holds none); `sl`
    selects episodes (DP shards); `overrides`: extra config keys; code_obs: the batch holds the observations as u8 class codes
    [B, T, n, V, V] (obs_storage: code; simplified palette only) instead of f32 planes."""
    cfg = load_config(meta["env"], overrides=dict(dict(env_args=meta["env_args"], use_cuda=device != "cpu", batch_size=4),
                                                  **dict(meta.get("overrides", {}), **(overrides or {}))))
    args = SimpleNamespace(**cfg)
    args.device = device
    obs = z["batch_obs"][sl]
    B, T, n = obs.shape[:3]
    args.n_agents, args.n_actions = n, z["batch_avail_actions"].shape[-1]
    args.obs_shape = obs.shape[3:]
    args.obs_dims = obs.shape[4:]
    H, W = {"default3": (10, 10), "default5": (25, 18), "default10": (48, 18)}[meta["env_args"].get("map", "default5")] \
        if meta["env"] == "cleanup" else (9, 38)
    args.state_dims = (H, W)
    scheme = {
        "obs": {"vshape": tuple(obs.shape[4:]), "group": "agents", "dtype": th.uint8} if code_obs else {"vshape": tuple(obs.shape[3:]), "group": "agents"},
        "actions": {"vshape": (1,), "group": "agents", "dtype": th.long},
        "avail_actions": {"vshape": (args.n_actions,), "group": "agents", "dtype": th.int},
        "reward": {"vshape": (n,)}, "terminated": {"vshape": (1,), "dtype": th.uint8},
        "clean_num": {"vshape": (n,)}, "apple_den": {"vshape": (n,)},
        "agent_pos": {"vshape": (n, 2)}, "agent_orientation": {"vshape": (n, 2)},
        "actions_inc": {"vshape": (n, 1), "group": "agents", "dtype": th.long},
    }
    groups = {"agents": n}
    preprocess = {"actions": ("actions_onehot", [OneHot(out_dim=args.n_actions)])}
    batch = EpisodeBatch(scheme, groups, B, T, preprocess=preprocess, device=device)
    data = {k: th.as_tensor(z["batch_" + k][sl]) for k in ("actions", "actions_inc", "reward", "terminated", "clean_num", "apple_den",
                                                            "agent_pos", "agent_orientation", "avail_actions")}
    if code_obs:
        r, g, b = (obs[:, :, :, c] == 255 for c in range(3))
        assert (np.isin(obs, (0, 255))).all() and (r.astype(int) + g + b <= 1).all()          # one colour per cell, full intensity
        data["obs"] = th.as_tensor((2 * r + 1 * g + 3 * b).astype(np.uint8))              # include/ssd_hip.h SSD_OBS_CODE classes
    else:
        data["obs"] = th.as_tensor(z["batch_obs"][sl]).float() / 256
    batch.update(data)
    assert (batch["filled"].cpu().numpy() == z["batch_filled"][sl]).all()
    mac = mac_REGISTRY[args.mac](batch.scheme, groups, args)
    sd = {k[2:]: th.as_tensor(z[k]) for k in z.files if k.startswith("w_")}
    if sd:
        mac.agent.load_state_dict(sd)
    logger = SimpleNamespace(log_stat=lambda *a, **k: None, console_logger=None)
    if device != "cpu":
        mac.cuda()
    learner = le_REGISTRY[args.learner](mac, batch.scheme, logger, args)
    if device != "cpu":
        learner.cuda()
    return args, batch, mac, learner


def param_checksums(mac):
    sums, sqs, heads = [], [], []
    for k, v in mac.agent.state_dict().items():
        x = v.detach().double().reshape(-1).cpu()
        sums.append(x.sum().item()); sqs.append((x * x).sum().item()); heads.append(np.resize(x[:5].numpy(), 5))
    return np.array(sums), np.array(sqs), np.stack(heads)
