#!/bin/bash
# Developer run: the attention kernel in the ablation builds named on the command line (DIAGS = knob values to time).
export ATTN_ONLY=1
for L in "$@"; do
  echo "== $L"
  EQUSS_LIB=$PWD/tools/ubench/libabl_$L.so timeout -k 10 100 python tools/attn_one.py ${DIAGS:-0 0} 2>&1 | grep "attention diag" || exit 1
done
