#!/bin/bash
# Diagnostic: dynamic instruction counts + time of the fused env kernel (rocprofv3 PMC pass + kbench).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_i
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --iters 40 --only step_observe_f32 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_i/*/*counter_collection.csv")[0]
vals = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_env<2" in r["Kernel_Name"]:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(vals["SQ_WAVES"]) / len(vals["SQ_WAVES"])
print("per wave:", {k: round(sum(v) / len(v) / w, 1) for k, v in vals.items() if k != "SQ_WAVES"})
PY
cd $GRAFT_REPO_ROOT && python3 tools/kbench.py --only step,step_observe_f32 2>&1 | grep -v amdgpu
