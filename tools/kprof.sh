#!/bin/bash
# Diagnostic: per-kernel time (rocprofv3 --kernel-trace --stats) and SQ / TCC counters (separate --pmc passes, never combined with
# other trace domains) of the rollout-timestep kernels.  usage: tools/kprof.sh <outdir> [kprof.py args...]
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp_*
python3 "$ROOT/tools/kprof.py" "$@" > "$OUT/events.txt" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_trace -- python3 "$ROOT/tools/kprof.py" "$@" > /dev/null 2>&1
cp /tmp/kp_trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P3="SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/kp_p1 -- python3 "$ROOT/tools/kprof.py" "$@" > /dev/null 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/kp_p2 -- python3 "$ROOT/tools/kprof.py" "$@" > /dev/null 2>&1
rocprofv3 --pmc $P3 --kernel-trace --output-format csv -d /tmp/kp_p3 -- python3 "$ROOT/tools/kprof.py" "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/kp_f -- python3 "$ROOT/tools/kprof.py" "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/kp_w -- python3 "$ROOT/tools/kprof.py" "$@" > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
keys = {"k_inc_encode": "k_inc_encode", "k_encode": "k_encode", "k_head<0": "k_head<env>", "k_head<1": "k_head<inc>", "k_env<2": "k_env<STEP_OBS>", "k_env<(ssd::MODE)2": "k_env<STEP_OBS>"}
res = collections.defaultdict(dict)
for d in ("kp_p1", "kp_p2", "kp_p3", "kp_f", "kp_w"):
    fs = glob.glob("/tmp/%s/*/*counter_collection.csv" % d)
    if not fs:
        res["_missing"][d] = True
        continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        for pat, name in keys.items():
            if pat in r["Kernel_Name"]:
                vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
    for name, v in vals.items():
        for cname, x in v.items():
            res[name][cname] = sum(x) / len(x)          # per launch
for name, v in res.items():
    if name.startswith("_") or "SQ_WAVES" not in v:
        continue
    w = v["SQ_WAVES"]
    v["per_wave"] = {c: round(x / w, 1) for c, x in v.items() if c.startswith("SQ_") and c != "SQ_WAVES"}
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:           # KB; gfx950: FETCH_SIZE counts half of a wide streaming read (MI355X_MICROARCH.md)
        v["hbm_bytes_per_launch"] = 1024.0 * (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0))
json.dump(res, open(out + "/pmc.json", "w"), indent=1, sort_keys=True)
for name, v in sorted(res.items()):
    if "per_wave" in v:
        print(name, "waves", v["SQ_WAVES"], v["per_wave"], "hbm MB", round(v.get("hbm_bytes_per_launch", 0) / 1e6, 2))
PY
cat "$OUT/events.txt"
head -12 "$OUT/kernel_stats.csv"
