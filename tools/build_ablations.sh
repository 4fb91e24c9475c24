#!/bin/bash
# Developer tool: builds of libequss_hip.so with one cost of the fused block-tail kernel removed each
# (tools/ubench/libabl_<NAME>.so), for tools/tail_bench.py.  Results of such builds are wrong by design.
set -e
P=expand-and-quantize-for-unsupervised-semantic-segmentation_amd
cd "$(dirname "$0")/.."
for NAME in "$@"; do
  case $NAME in RS*) SRC=block_rs ;; WS*) SRC=gemm_ws ;; ATT*) SRC=attention ;; BW*) SRC=block_ws ;; PQ*) SRC=pq_assign ;; G8*) SRC=gemm8_bf16 ;; SW*) SRC=gemm_sw ;; *) SRC=mlp_fused ;; esac      # WS...: ablations of the producer / consumer GEMM
  OBJS=$(ls $P/build/*.o | grep -v $SRC.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=on $(for D in ${NAME//+/ }; do echo -n "-DEQUSS_ABL_$D "; done) -c $P/csrc/$SRC.hip -o /tmp/abl_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libabl_$NAME.so $OBJS /tmp/abl_$NAME.o
done
