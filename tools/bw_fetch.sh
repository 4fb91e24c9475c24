#!/bin/bash
# Developer tool: time and HBM read bytes (FETCH_SIZE, one --pmc pass) of the block-tail kernel for builds of the library
# (tools/build_ablations.sh BW...; "current" = the library in the tree).  usage: tools/bw_fetch.sh current BWNTX ...
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
for L in "$@"; do
  LIB=""; [ "$L" != current ] && LIB=$ROOT/tools/ubench/libabl_$L.so
  EQUSS_LIB=$LIB timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 --streams 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$L', d['value'], d['ms_per_step'], {n: k[n]['avg_launch_us'] for n in ('block_tail', 'gemm_qkv', 'attention', 'head_gemm2')})" || exit 1
  OUT=$ROOT/gpurun_out/bwf_$L; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && EQUSS_LIB=$LIB timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $ROOT/tools/ab_bench.py --cpu-seconds 0 --streams 1 --steps 3 --warmup 1 > $OUT.log 2>&1) || exit 1
  python3 - $OUT <<'P'
import csv, glob, sys
tot = n = 0
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "block_ws_kernel" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print("   block tail HBM read per launch: %.1f MB over %d launches (FETCH_SIZE x 2)" % (2 * tot / max(n, 1) * 1024 / 1e6, n))
P
done
