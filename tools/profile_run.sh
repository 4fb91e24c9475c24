#!/bin/bash
# One profiling round on the GPU box: kernel trace of the one-stream (default) and of the two-stream bench, then the
# PMC passes (FETCH_SIZE, WRITE_SIZE, matrix-pipe busy cycles) each in a run of its own.  usage: tools/profile_run.sh <tag>
set -u
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --streams 1 > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- $B --streams 2 > $OUT/trace2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --streams 1 --steps 5 > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --streams 1 --steps 5 > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $B --streams 1 --steps 5 > $OUT/mfma.log 2>&1 || exit 1
echo profile $TAG done
