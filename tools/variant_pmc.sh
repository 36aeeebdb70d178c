#!/bin/bash
# Diagnostic: per-wave instruction counts (SQ PMC pass) + GPU-side duration of the rollout kernels for A/B builds of the library.
# usage: tools/variant_pmc.sh <kprof args...> -- lib1.so ...   ("product" = the in-tree library)
ARGS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/vq
  if [ "$lib" = "product" ]; then unset SSD_HIP_LIB_PATH; else export SSD_HIP_LIB_PATH=$ROOT/$lib; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/vq -- python3 $ROOT/tools/kprof.py "${ARGS[@]}" > /dev/null 2>&1
  echo "== $lib"
  python3 - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob("/tmp/vq/*/*counter_collection.csv")[0])):
    for pat in ("k_env<2", "k_encode", "k_head<0", "k_head<1"):
        if pat in r["Kernel_Name"]:
            vals[pat][r["Counter_Name"]].append(float(r["Counter_Value"]))
for pat, v in vals.items():
    w = sum(v["SQ_WAVES"]) / len(v["SQ_WAVES"])
    print("  ", pat, {k: round(sum(x) / len(x) / w, 1) for k, x in v.items() if k != "SQ_WAVES"})
PY
done
