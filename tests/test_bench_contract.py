"""The bench lines committed under profiles/ carry every field of the driver contract (bench.py writes them; this guards the
contract against accidental edits -- it does not run the bench)."""
import json
import os

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _check_roofline(r):
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == {"hbm": "GB/s", "mfma": "TFLOP/s"}[r["bound"]]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["kernel_avg_us"] > 0
    if r["bound"] == "hbm":
        assert r["peak"] == 8000.0
        assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    else:
        assert r["peak"] == 2500.0                 # the ISSUED 16-bit MFMA products (2-3 per f32-equivalent product) against the dense f16/bf16 peak
        assert r["algorithmic_flops_per_launch"] > 0
        a32 = r.get("algorithmic_f32")      # informational second pricing: the reference's f32 FLOPs against the f32-MFMA peak
        assert a32 is None or (a32["peak"] == 157.3 and abs(a32["frac"] - a32["achieved"] / a32["peak"]) < 1e-9
                               and a32["algorithmic_flops_per_launch"] <= r["algorithmic_flops_per_launch"])


@pytest.mark.parametrize("name", ["r02_bench_e2e_cleanup5.json", "r02_bench_e2e_harvest5.json", "r02_bench_e2e_cleanup10.json",
                                  "r02_bench_env_cleanup5.json", "r03_bench_e2e_cleanup5.json", "r03_bench_e2e_harvest5.json",
                                  "r03_bench_e2e_cleanup10.json", "r03_bench_env_cleanup5.json", "r03_bench_e2e_cleanup5_tspr8.json",
                                  "r04_bench_e2e_cleanup5.json", "r04_bench_e2e_harvest5.json", "r04_bench_e2e_cleanup10.json", "r04_bench_env_cleanup5.json",
                                  "r04_bench_e2e_cleanup5_tspr8.json", "r04_bench_e2e_cleanup5_learner_bf16_tspr8.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    d = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in REQUIRED:
        assert k in d or (k == "cpu_baseline" and ("r02_bench_env" in name or "tspr8" in name)), k      # lines taken with --no-cpu-baseline
    assert d["metric"] == "agent_steps_per_sec" and d["unit"] == "agent-steps/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    _check_roofline(d["roofline"])
    for k in d.get("kernels", []):
        _check_roofline(k)
    assert d["roofline"]["kernel_avg_us"] == max(k["kernel_avg_us"] for k in d.get("kernels", [d["roofline"]]))      # the DOMINANT kernel is the one reported
    cfg = d["config"]
    assert cfg["world_size"] == d["n_gpus"] and cfg["backend"]
    units = cfg["n_env_per_gpu"] * cfg["n_agents"] * d["n_gpus"] * d["steps"]
    if "e2e" in name:       # one e2e step = one whole iteration: a 100-timestep rollout of every env + the train steps
        units *= cfg["episode_limit"]
        assert cfg["timesteps_timed"] == d["steps"] * cfg["episode_limit"] and cfg["train_steps_timed"] >= d["steps"]
    assert abs(d["value"] - units / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-3 * d["value"]
    ew = d.get("env_workload")      # round 4: the format-R env workload measured after the timed region of the e2e line (1 GPU)
    if name.startswith("r04_bench_e2e") and d["n_gpus"] == 1:
        assert ew is not None and ew["unit"] == "GB/s" and ew["peak"] == 8000.0 and abs(ew["frac"] - ew["achieved"] / ew["peak"]) < 1e-9
        assert abs(ew["achieved"] - ew["algorithmic_bytes_per_launch"] / (ew["avg_us"] * 1e-6) / 1e9) < 1e-6 * ew["achieved"] and ew["transitions"] >= 100
    c = d.get("cpu_baseline")
    assert c is None or c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "agent-steps/s" and c["sample"]


def test_roofline_bytes_follow_the_survey_formula():
    import sys
    sys.path.insert(0, ROOT)
    from bench import algorithmic_bytes_per_env_step
    assert algorithmic_bytes_per_env_step(25, 18, 5, 15) == 14640       # Cleanup-5, SURVEY.md section 8(d)
    assert algorithmic_bytes_per_env_step(10, 10, 3, 15) == 8444        # Cleanup-3
    assert algorithmic_bytes_per_env_step(48, 18, 10, 15) == 29208      # Cleanup-10
    assert algorithmic_bytes_per_env_step(9, 38, 5, 31) == 58584        # Harvest-5


def test_bench_front_door_starts_its_own_ranks_for_n_gt_1():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command form) starts two ranks as a child
    torch.distributed.run job, the ranks count each other with an all-reduce, and exactly one JSON line comes back on stdout."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["world_size"] == 2 and d["config"]["self_launched"] is True


def test_bench_front_door_propagates_a_failing_rank():
    """a rank that dies makes the self-launched job exit non-zero and print no JSON line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SSD_DRY_RUN_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and not p.stdout.decode().strip()
