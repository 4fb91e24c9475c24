"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (its *_results.db) as CSV:
name, calls, total_us, average_us, percent.   python tools/rocprof_top.py <results.db> [rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("name,calls,total_us,average_us,percent")
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (rows,)):
    print(f"\"{name[:160]}\",{calls},{total:.1f},{avg:.2f},{pct:.2f}")
