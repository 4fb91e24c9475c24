# Developer tool: A/B of the current library against tools/ubench/libhead.so (a build of the last commit) in one call
cd "$(dirname "$0")/.."
show() { grep "^{" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('%-8s %7.1f img/s  %6.3f ms | ' % (sys.argv[1], d['value'], d['ms_per_step']) + '  '.join('%s %.1f' % (n, k[n]['avg_launch_us']) for n in ('block_tail', 'gemm_qkv', 'attention', 'pq_assign', 'head_gemm2') if n in k))" $1; }
for i in 1 2; do
EQUSS_LIB=$PWD/tools/ubench/libhead.so timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 2>/dev/null | show head || exit 1
timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 2>/dev/null | show new || exit 1
done
