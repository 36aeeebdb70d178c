#!/bin/bash
# Diagnostic: GPU-side duration (rocprofv3 --kernel-trace --stats) of the rollout kernels for A/B builds of the library.
# usage: tools/variant_prof.sh <kprof args...> -- lib1.so lib2.so ...   ("product" = the in-tree library)
ARGS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/vp
  if [ "$lib" = "product" ]; then unset SSD_HIP_LIB_PATH; else export SSD_HIP_LIB_PATH=$ROOT/$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -- python3 $ROOT/tools/kprof.py "${ARGS[@]}" > /dev/null 2>&1
  echo "== $lib"
  python3 - <<'PY'
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/vp/*/*kernel_stats.csv")[0])):
    if any(k in r["Name"] for k in ("k_head", "k_encode", "k_env<2")):
        print("   %-60s calls %5s avg %8.2f us  min %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
