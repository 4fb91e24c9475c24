# Developer tool: the producer / consumer GEMM (variant 3) against the one-role kernel (variant 4) and its ablation builds
cd "$(dirname "$0")/.."
timeout -k 10 120 python tools/gemm_bench.py 4 3 2>&1 | grep -E "qkv|fc1 " || exit 1
for n in WSNOEPI WSNOSTORE WSNOMFMA; do
  echo "-- $n"; EQUSS_LIB=$PWD/tools/ubench/libabl_$n.so timeout -k 10 120 python tools/gemm_bench.py 3 2>&1 | grep -E "qkv|fc1 " || exit 1
done
