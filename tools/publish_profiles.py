"""Copy a tools/collect_profiles.sh result directory (gpurun_out/<name>) into profiles/ under this round's prefix and rebuild
profiles/traffic.json (per-launch HBM bytes of the PMC passes, keyed "<kernel>@<config>", read back by bench.py).
usage: python tools/publish_profiles.py gpurun_out/r02p r02"""
import json
import os
import shutil
import sys

src, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
sys.path.insert(0, root)
from bench import CONFIGS  # noqa: E402

for f in sorted(os.listdir(src)):
    p = os.path.join(src, f)
    if f.startswith("bench_") and f.endswith(".json"):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]      # gloo prints its connection banner to stdout
        json.loads(lines[-1])
        open(os.path.join(dst, "%s_%s" % (rnd, f)), "w").write(lines[-1] + "\n")
    elif f.endswith("_kernel_stats.csv"):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (rnd, f)))
names = {"k_inc_encode": "ssd::k_inc_encode", "k_encode": "ssd::k_encode", "k_head<env>": "ssd::k_head<env>", "k_head<inc>": "ssd::k_head<inc>", "k_env<STEP_OBS>": "ssd::k_env<MODE_STEP_OBS>"}
traffic = dict(note="HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/kprof.py (the rollout's "
                    "launches on live data); FETCH_SIZE x2 (gfx950 reports half of a 16 B/lane coalesced stream, MI355X_MICROARCH.md "
                    "HBM), WRITE_SIZE as counted.  Infinity-Cache hits are counted: kernels that re-read their weights show L2-miss "
                    "traffic, not DRAM traffic.", kernels={})
for f in sorted(os.listdir(src)):
    d = os.path.join(src, f)
    if not (f.startswith("kprof_") and os.path.isdir(d)):
        continue
    tag = f[len("kprof_"):]
    cfg = tag.split("_")[0]
    storage = "f32" if tag.endswith("f32storage") else "code"
    standalone = tag.endswith("standalone")
    for g in ("pmc.json", "kernel_stats.csv", "events.txt"):
        if os.path.exists(os.path.join(d, g)):
            shutil.copy(os.path.join(d, g), os.path.join(dst, "%s_kprof_%s_%s" % (rnd, tag, g)))
    pmc = json.load(open(os.path.join(d, "pmc.json")))
    for k, v in pmc.items():
        if standalone and k not in ("k_encode", "k_head<inc>"):
            continue                    # the standalone pass adds only the two kernels the pipelined timestep fuses
        if k in names and "hbm_bytes_per_launch" in v:
            key = "%s@%s" % (names[k], cfg) + ("" if storage == "code" else "@f32storage")
            traffic["kernels"][key] = dict(n_env=CONFIGS[cfg]["n_env"], obs_format=storage, hbm_bytes_per_launch=v["hbm_bytes_per_launch"],
                                           FETCH_SIZE_KB=v.get("FETCH_SIZE"), WRITE_SIZE_KB=v.get("WRITE_SIZE"), source="%s_kprof_%s_pmc.json" % (rnd, tag))
ew = os.path.join(src, "env_workload_traffic.json")
if os.path.exists(ew):
    v = json.load(open(ew))
    if v.get("hbm_bytes_per_launch"):
        shutil.copy(ew, os.path.join(dst, "%s_env_workload_traffic.json" % rnd))
        traffic["kernels"]["ssd::k_env<MODE_STEP_OBS>@cleanup5@env_workload"] = dict(
            n_env=CONFIGS["cleanup5"]["n_env"], obs_format="f32", hbm_bytes_per_launch=v["hbm_bytes_per_launch"], FETCH_SIZE_KB=v["FETCH_SIZE_KB"],
            WRITE_SIZE_KB=v["WRITE_SIZE_KB"], source="%s_env_workload_traffic.json" % rnd)
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1, sort_keys=True)
print("published", len(traffic["kernels"]), "traffic entries")
