#!/bin/bash
# Developer run: SQ counters of the attention kernel (one rocprofv3 --pmc pass per counter group).
set -u
OUT=/root/repo/gpurun_out/attn_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ATTN_ONLY=1
DIAG=${1:-0}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 /root/repo/tools/attn_one.py $DIAG > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p2 -- python3 /root/repo/tools/attn_one.py $DIAG > $OUT/p2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"/root/repo/gpurun_out/attn_pmc/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attention" not in r["Kernel_Name"]:
                continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(f"{p} {k:28s} {v / n:16.0f} per launch ({n} launches)")
PY
