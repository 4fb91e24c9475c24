# Developer tool: tools/tail_bench.py over the default library and the ablation builds named on the command line,
# three interleaved rounds in one GPU call
cd "$(dirname "$0")/.."
for round in 1 2 3; do
timeout -k 10 120 python tools/tail_bench.py || exit 1
for n in "$@"; do
  EQUSS_LIB=$PWD/tools/ubench/libabl_$n.so timeout -k 10 120 python tools/tail_bench.py || exit 1
done
done
