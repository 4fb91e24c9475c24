#!/bin/bash
# Developer tool: the default bench (two streams) of the library in the tree against ablation / variant builds, interleaved.
# usage: tools/ab2.sh NAME ...   (tools/ubench/libabl_NAME.so)
cd "$(dirname "$0")/.."
for i in 1 2 3; do for L in current "$@"; do
  LIB=""; [ "$L" != current ] && LIB=$PWD/tools/ubench/libabl_$L.so
  EQUSS_LIB=$LIB timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$L', d['value'], d['ms_per_step'])" || exit 1
done; done
