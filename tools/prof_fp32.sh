cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_fp32
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 /root/repo/tools/fp32_mode_run.py > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob
f=max(glob.glob('/root/repo/gpurun_out/prof_fp32/t/*/*_kernel_stats.csv'))
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per forward', tot/1e6/5)
for r in rows[:16]:
    print(f"{float(r['TotalDurationNs'])/1e6/5:8.3f} ms/fwd {int(r['Calls'])/5:>6} {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
