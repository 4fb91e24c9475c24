# Developer tool: tools/tail_bench.py over the default library and the ablation builds named on the command line
cd "$(dirname "$0")/.."
timeout -k 10 120 python tools/tail_bench.py || exit 1
for n in "$@"; do
  EQUSS_LIB=$PWD/tools/ubench/libabl_$n.so timeout -k 10 120 python tools/tail_bench.py || exit 1
done
timeout -k 10 120 python tools/tail_bench.py
