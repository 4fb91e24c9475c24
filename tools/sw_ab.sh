# Developer tool: the stationary-weights GEMM (default) against the producer / consumer kernel (gemm_variant=6), one stream
cd "$(dirname "$0")/.."
for v in ${@:-6 3 6 3}; do L=""; case $v in SW*) L=$PWD/tools/ubench/libabl_$v.so ;; esac
EQUSS_LIB=$L timeout -k 10 200 python tools/ab_bench.py --cpu-seconds 0 --streams 1 $(case $v in SW*) ;; *) echo --dev gemm_variant=$v ;; esac) 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k = d['kernels']
print('variant', sys.argv[1], d['value'], d['ms_per_step'], {n: k[n]['avg_launch_us'] for n in ('gemm_qkv', 'head_gemm1', 'attention', 'block_tail')})" $v || exit 1; done
