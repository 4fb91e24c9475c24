# Developer tool: PQ assign builds (tools/build_ablations.sh PQ...) against the current library, interleaved, in one GPU call
cd "$(dirname "$0")/.."
show() { grep "^{" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('%-14s %7.1f img/s  %6.3f ms | ' % (sys.argv[1][:14], d['value'], d['ms_per_step']) + '  '.join('%s %.1f' % (n, k[n]['avg_launch_us']) for n in ('pq_assign', 'head_gemm2', 'block_tail', 'gemm_qkv', 'attention') if n in k))" $1; }
for i in 1 2; do
timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 --streams 1 2>/dev/null | show current || exit 1
for L in "$@"; do
EQUSS_LIB=$PWD/tools/ubench/libabl_$L.so timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 --streams 1 2>/dev/null | show $L || exit 1
done
done
