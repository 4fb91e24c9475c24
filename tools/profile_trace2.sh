#!/bin/bash
# The two-stream kernel trace of tools/profile_run.sh alone.  usage: tools/profile_trace2.sh <tag>
set -u
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python3 /root/repo/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-secondary --streams 2 > $OUT/trace2.log 2>&1 || exit 1
echo trace2 $TAG done
