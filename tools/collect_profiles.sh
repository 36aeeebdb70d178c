#!/bin/bash
# Collect the round's judged evidence on a GPU box into gpurun_out/$1/ (copy what is to be kept into profiles/ afterwards):
# bench lines of the three BASELINE configurations (+ variants), the kernel-trace summary of the default bench command, the
# per-kernel SQ / TCC counter passes (tools/kprof.sh), a 2-rank gloo rehearsal of the distributed bench path.
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
set -x
python3 bench.py > $OUT/bench_e2e_cleanup5.json 2> $OUT/bench_e2e_cleanup5.err || exit 1
python3 bench.py --config harvest5 --steps 10 --warmup 3 > $OUT/bench_e2e_harvest5.json 2> $OUT/bench_e2e_harvest5.err || exit 1
python3 bench.py --config cleanup10 --steps 10 --warmup 3 > $OUT/bench_e2e_cleanup10.json 2> $OUT/bench_e2e_cleanup10.err || exit 1
python3 bench.py --no-cpu-baseline --qnet-dtype bf16 > $OUT/bench_e2e_cleanup5_bf16.json 2> $OUT/bf16.err || exit 1
python3 bench.py --no-cpu-baseline --obs-storage f32 > $OUT/bench_e2e_cleanup5_f32storage.json 2> $OUT/f32.err || exit 1
python3 bench.py --workload env > $OUT/bench_env_cleanup5.json 2> $OUT/env.err || exit 1
python3 bench.py --no-cpu-baseline --train-steps-per-rollout 8 > $OUT/bench_e2e_cleanup5_tspr8.json 2> $OUT/tspr8.err || exit 1
# labelled variants: the bf16 learner (single bf16 MFMA products in the learner's GEMMs), alone and with the bf16 rollout; round 3's Toeplitz encoder
python3 bench.py --no-cpu-baseline --learner-dtype bf16 --train-steps-per-rollout 8 > $OUT/bench_e2e_cleanup5_learner_bf16_tspr8.json 2> $OUT/lbf16.err || exit 1
python3 bench.py --no-cpu-baseline --learner-dtype bf16 --qnet-dtype bf16 > $OUT/bench_e2e_cleanup5_all_bf16.json 2> $OUT/allbf16.err || exit 1
SSD_ENC_LAYOUT=toeplitz python3 bench.py --no-cpu-baseline > $OUT/bench_e2e_cleanup5_toeplitz_encoder.json 2> $OUT/toeplitz.err || exit 1
# the multi-rank front door (python bench.py --gpus N starts its own ranks): a 1-rank RCCL group and a 2-rank gloo group on this one GPU
SSD_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_e2e_1rank_rccl_rehearsal.json 2> $OUT/rccl1.err || exit 1
SSD_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --n-env 1024 --no-cpu-baseline > $OUT/bench_e2e_2rank_gloo_rehearsal.json 2> $OUT/gloo.err || exit 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$OUT/bench_e2e_under_rocprof.json 2> $R/$OUT/rocprof.err; cp /tmp/bp/*/*kernel_stats.csv $R/$OUT/e2e_kernel_stats.csv)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/be && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/be -- python3 $R/bench.py --workload env --no-cpu-baseline > $R/$OUT/bench_env_under_rocprof.json 2> $R/$OUT/rocprof_env.err; cp /tmp/be/*/*kernel_stats.csv $R/$OUT/env_kernel_stats.csv)
# HBM traffic of the env workload's kernel (separate --pmc passes, nothing else traced)
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ef /tmp/ew && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ef -- python3 $R/bench.py --workload env --no-cpu-baseline > /dev/null 2>&1; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ew -- python3 $R/bench.py --workload env --no-cpu-baseline > /dev/null 2>&1; python3 $R/tools/pmc_bytes.py /tmp/ef /tmp/ew "k_env<2" > $R/$OUT/env_workload_traffic.json)
for c in cleanup5 harvest5 cleanup10; do bash tools/kprof.sh $OUT/kprof_$c --config $c > $OUT/kprof_$c.log 2>&1; done
bash tools/kprof.sh $OUT/kprof_cleanup5_f32storage --config cleanup5 --obs-storage f32 > $OUT/kprof_cleanup5_f32storage.log 2>&1
# the four standalone launches (k_encode and k_head<inc> on their own: what the pipelined timestep fuses)
bash tools/kprof.sh $OUT/kprof_cleanup5_standalone --config cleanup5 --pipeline 0 > $OUT/kprof_cleanup5_standalone.log 2>&1
python3 tools/benchsum.py $OUT/bench_*.json
