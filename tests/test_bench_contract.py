"""The bench lines committed under profiles/ carry every field of the driver contract (bench.py writes them; this guards the
contract against accidental edits -- it does not run the bench)."""
import json
import os

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("name", ["r01_bench_e2e.json", "r01_bench_env.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    d = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in REQUIRED:
        assert k in d, k
    assert d["metric"] == "agent_steps_per_sec" and d["unit"] == "agent-steps/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    assert abs(d["value"] - d["config"]["n_env_per_gpu"] * d["config"]["n_agents"] * d["n_gpus"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-3 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "agent-steps/s" and c["sample"]


def test_roofline_bytes_follow_the_survey_formula():
    import sys
    sys.path.insert(0, ROOT)
    from bench import algorithmic_bytes_per_env_step
    assert algorithmic_bytes_per_env_step(25, 18, 5, 15) == 14640       # Cleanup-5, SURVEY.md section 8(d)
    assert algorithmic_bytes_per_env_step(10, 10, 3, 15) == 8444        # Cleanup-3
    assert algorithmic_bytes_per_env_step(48, 18, 10, 15) == 29208      # Cleanup-10
    assert algorithmic_bytes_per_env_step(9, 38, 5, 31) == 58584        # Harvest-5
