#!/bin/bash
# Developer tool: hardware counters of the block-tail kernel alone (tools/tail_bench.py), one pass per run.
set -u
OUT=/root/repo/gpurun_out/tail_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/tools/tail_bench.py"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- $B > $OUT/tcc.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -- $B > $OUT/tcp.log 2>&1 || exit 1
echo done
