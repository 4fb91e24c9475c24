#!/bin/bash
# Developer tool: the shader clock a kernel actually ran at = GRBM_GUI_ACTIVE cycles / kernel duration
# (one counter pass with the kernel trace), for the block-tail kernel and for the GEMM micro-benchmark.
set -u
OUT=/root/repo/gpurun_out/clock_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/tail -- python3 /root/repo/tools/tail_bench.py > $OUT/tail.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/gemm -- python3 /root/repo/tools/gemm_bench.py 3 > $OUT/gemm.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("tail", "gemm"):
    for f in glob.glob(f"/root/repo/gpurun_out/clock_pmc/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
                agg[r["Kernel_Name"][:70]].append((float(r["Counter_Value"]), dur))
        for k, v in agg.items():
            c = sum(x[0] for x in v) / len(v); t = sum(x[1] for x in v) / len(v)
            print(f"{d} {k:70s} n={len(v):4d} cycles={c:12.0f} dur_ns={t:10.0f} GHz={c / t if t else 0:.3f}")
PY
