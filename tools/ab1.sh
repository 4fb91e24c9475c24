# Developer tool: the one-stream bench of the library in the tree against builds under tools/ubench/libabl_NAME.so, interleaved.
# usage: tools/ab1.sh NAME ...
cd "$(dirname "$0")/.."
for i in 1 2; do for L in current "$@"; do
  LIB=""; [ "$L" != current ] && LIB=$PWD/tools/ubench/libabl_$L.so
  EQUSS_LIB=$LIB timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 --streams 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('$L', d['value'], d['ms_per_step'], {n: k[n]['avg_launch_us'] for n in ('block_tail', 'gemm_qkv', 'attention', 'head_gemm1', 'head_gemm2', 'patch')})" || exit 1
done; done
