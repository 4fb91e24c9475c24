# Developer tool: A/B of the current library against other builds (default tools/ubench/libhead.so, a build of the
# last commit) in one GPU call.  usage: tools/ab_run.sh [lib.so ...]
cd "$(dirname "$0")/.."
show() { grep "^{" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('%-12s %7.1f img/s  %6.3f ms | ' % (sys.argv[1][:12], d['value'], d['ms_per_step']) + '  '.join('%s %.1f' % (n, k[n]['avg_launch_us']) for n in ('block_tail', 'gemm_qkv', 'attention', 'pq_assign', 'head_gemm2') if n in k))" $1; }
LIBS="${@:-tools/ubench/libhead.so}"
for i in 1 2; do
for L in $LIBS; do
EQUSS_LIB=$PWD/$L timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 2>/dev/null | show $(basename $L .so) || exit 1
done
timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 2>/dev/null | show new || exit 1
done
