# Developer tool: bench.py with and without the producer / consumer block-tail kernel, interleaved in one GPU call
cd "$(dirname "$0")/.."
show() { grep "^{" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d.get('kernels', {})
print('%-22s %7.1f img/s  %6.3f ms | ' % (sys.argv[1][:22], d['value'], d['ms_per_step']) + '  '.join('%s %.1f' % (n, k[n]['avg_launch_us']) for n in ('block_tail', 'gemm_qkv', 'attention', 'pq_assign') if n in k))" $1; }
for i in 1 2; do
timeout -k 10 200 python bench.py --cpu-seconds 0 --streams 1 2>/dev/null | show default_1stream || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --streams 1 --dev block_ws=1 2>/dev/null | show ws_1stream || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --streams 1 --dev block_ws=1 --dev mlp_tail_max=0 2>/dev/null | show ws_nosplit_1stream || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --streams 3 --dev block_ws=1 --dev mlp_tail_max=0 2>/dev/null | show ws_nosplit_3streams || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --streams 4 --dev block_ws=1 --dev mlp_tail_max=0 2>/dev/null | show ws_nosplit_4streams || exit 1
done
