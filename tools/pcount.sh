#!/bin/bash
# Diagnostic: SQ counters per wave of the fused policy kernels (rocprofv3 PMC pass over tools/policy_prof.py).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_p
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_p -- python3 $GRAFT_REPO_ROOT/tools/policy_prof.py > /dev/null 2>&1
rm -rf /tmp/pmc_q
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_q -- python3 $GRAFT_REPO_ROOT/tools/policy_prof.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pmc_p", "/tmp/pmc_q"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print("no counters in", d); continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        for key in ("k_encode", "k_head<0>", "k_head<1>"):
            if key in k:
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, v in vals.items():
        w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
        print(key, "waves", w, {c: round(sum(x) / len(x) / w, 1) for c, x in v.items() if c != "SQ_WAVES"})
PY
