cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_train
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 /root/repo/tools/train_step.py --batch 64 --res 320 --steps 5 --warmup 2 > $OUT/log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob
f=max(glob.glob('/root/repo/gpurun_out/prof_train/t/*/*_kernel_stats.csv'))
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/1e6/7)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6/7:8.3f} ms/step {r['Calls']:>6} {r['Name'][:100]}")
PY
tail -2 $OUT/log.txt
