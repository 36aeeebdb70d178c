#!/bin/bash
# Diagnostic: build libssd_hip variants with extra -D flags (CPU side, before the GPU call) and bench each on the GPU box.
#   local:  bash tools/variants.sh build name1 "-DX=1" name2 "-DX=2" ...
#   box:    bash tools/variants.sh run outdir name1 name2 ...   (runs `python bench.py --no-cpu-baseline $BENCH_ARGS` per variant)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $R/build/variants
  while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=14 $flags -o $R/build/variants/libssd_$name.so $R/homophily_marl_amd/csrc/*.hip 2>/dev/null && echo built $name ) &
  done
  wait
else
  out=$1; shift
  mkdir -p $R/$out
  for name in "$@"; do
    SSD_HIP_LIB_PATH=$R/build/variants/libssd_$name.so python3 $R/bench.py --no-cpu-baseline $BENCH_ARGS > $R/$out/bench_$name.json 2> $R/$out/bench_$name.err || echo "$name FAILED"
    python3 $R/tools/benchsum.py $R/$out/bench_$name.json
  done
fi
