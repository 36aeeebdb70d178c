#!/bin/bash
# Diagnostic: the e2e bench line under HIP runtime knobs that touch graph dispatch (launch-to-launch floor of dependent kernel nodes).
# bash tools/knobs.sh OUTDIR ; stops at the first run that had to be killed.
OUT=${1:-gpurun_out/knobs}
mkdir -p $OUT
run() {
    name=$1; shift
    env "$@" timeout -k 10 150 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/$name.json 2> $OUT/$name.err
    rc=$?
    echo "$name rc=$rc $(python3 tools/benchsum.py $OUT/$name.json 2>/dev/null | head -1)"
    if [ $rc -ge 124 ]; then echo "killed: stopping"; exit 1; fi
}
run base SSD_KNOB=none
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch1000 DEBUG_HIP_GRAPH_BATCH_SIZE=1000
# ROC_SYSTEM_SCOPE_SIGNAL=0 was tried once: the process hangs (killed by the timeout) -- do not set it
run graphq1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run base2 SSD_KNOB=none
