"""Build an `ssd_config` from the reference's env kwargs (config/envs/cleanup.yaml:3-15, harvest.yaml:3-15)."""
from . import maps
from .. import abi

DEFAULT_EXTRA_ARGS = dict(random_spawn_point=False, random_spawn_rotation=0, disable_rotation_action=True,
                          disable_fire_action=True, obs_color="simplified")


def make_config(env, map="default", num_agents=1, n_env=1, view_size=7, episode_limit=100, extra_args=None,
                rng_mode=abi.RNG_COUNTER, seed=0, device=0, env_id_base=0, ascii_map=None):
    """Returns (SsdConfig, MapSpec).  `env` is the registry key ("cleanup" / "harvest"); the remaining names are
    the reference constructor's (cleanup.py:29, harvest.py:18).  `ascii_map` overrides the layout of `map`
    (the reference's `ascii_map` kwarg is overridden by `map`, cleanup.py:31-54; here an explicit layout wins and
    keeps the rule constants of `map`)."""
    ea = dict(DEFAULT_EXTRA_ARGS)
    if extra_args:
        ea.update(extra_args)
    spec = maps.get_spec(env, map)
    if ascii_map is not None:
        spec = maps.MapSpec(spec.env, tuple(ascii_map), spec.threshold_depletion, spec.threshold_restoration,
                            spec.waste_spawn_prob, spec.apple_respawn_prob, spec.harvest_spawn_prob)
    if ea["obs_color"] not in ("simplified", "full"):
        raise ValueError("obs_color must be 'simplified' or 'full'")
    rot = ea["random_spawn_rotation"]
    cfg = abi.SsdConfig()
    cfg.env_kind = abi.ENV_CLEANUP if spec.env == "cleanup" else abi.ENV_HARVEST
    cfg.height, cfg.width = spec.height, spec.width
    cfg._ascii_keepalive = spec.ascii          # keep the bytes object alive as long as the struct
    cfg.ascii_map = cfg._ascii_keepalive
    cfg.n_agents, cfg.n_env, cfg.view_size, cfg.episode_limit = int(num_agents), int(n_env), int(view_size), int(episode_limit)
    cfg.random_spawn_point = 1 if ea["random_spawn_point"] else 0
    cfg.spawn_rotation = -1 if rot is None else int(rot)
    cfg.obs_color = abi.COLOR_SIMPLIFIED if ea["obs_color"] == "simplified" else abi.COLOR_FULL
    cfg.rng_mode = int(rng_mode)
    cfg.device = int(device)
    cfg.env_id_base = int(env_id_base)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.threshold_depletion = spec.threshold_depletion
    cfg.threshold_restoration = spec.threshold_restoration
    cfg.waste_spawn_prob = spec.waste_spawn_prob
    cfg.apple_respawn_prob = spec.apple_respawn_prob
    for i in range(4):
        cfg.harvest_spawn_prob[i] = spec.harvest_spawn_prob[i]
    return cfg, spec
