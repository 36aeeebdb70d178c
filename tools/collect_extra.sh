#!/bin/bash
# The rest of a round's evidence (after tools/collect_profiles.sh): the GPU test log, smoke(), in-kernel stamps of the rollout kernels,
# the learner's kernel mix and ordered launch list, the micro-benchmarks of the learner kernels, a short soak.  -> gpurun_out/$1/
OUT=gpurun_out/${1:-extra}
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
set -x
python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 || exit 1
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || exit 1
python3 tools/pstamps.py > $OUT/pstamps.txt 2> $OUT/pstamps.err || exit 1
python3 tools/stamps.py --fmt code --dest storage --light > $OUT/env_stamps.txt 2> $OUT/env_stamps.err || exit 1
python3 tools/stamps.py --fmt f32 --light > $OUT/env_stamps_f32.txt 2>> $OUT/env_stamps.err || exit 1
python3 tools/gru_prof.py > $OUT/gru_prof.txt 2>&1 || exit 1
python3 tools/bmm_prof.py > $OUT/bmm_prof.txt 2>&1 || exit 1
python3 tools/train_prof.py > $OUT/train_prof.txt 2>&1 || exit 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -- python3 $R/tools/train_prof.py > $R/$OUT/train_prof_under_rocprof.txt 2>&1; cp /tmp/tt/*/*kernel_stats.csv $R/$OUT/train_kernel_stats.csv; python3 $R/tools/train_trace.py /tmp/tt > $R/$OUT/train_trace.txt 2>&1)
python3 tools/soak.py --iters ${SOAK_ITERS:-2500} --every 250 > $OUT/soak_cleanup5_tspr8_seed1.txt 2>&1 || exit 1
for f in pytest_gpu.log soak_cleanup5_tspr8_seed1.txt train_prof.txt; do tail -n 3 $OUT/$f; done
