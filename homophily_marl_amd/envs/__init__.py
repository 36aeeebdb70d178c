"""Env registry with the reference's keys (src/envs/__init__.py:9-11): REGISTRY[args.env](**args.env_args)."""
from functools import partial


def _make(name, **kwargs):
    from .ssd_env import CleanupHipEnv, HarvestHipEnv     # imported lazily: needs torch + the HIP library
    return {"cleanup": CleanupHipEnv, "harvest": HarvestHipEnv}[name](**kwargs)


REGISTRY = {"cleanup": partial(_make, "cleanup"), "harvest": partial(_make, "harvest"),
            "cleanup_hip": partial(_make, "cleanup"), "harvest_hip": partial(_make, "harvest")}
