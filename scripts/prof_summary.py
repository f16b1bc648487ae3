"""Turns a rocprofv3 rocpd sqlite database (kernel trace) into a kernel-stats summary text."""
import re
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary ({db}); durations in us; {steps} profiled steps (+warmup); total kernel time {total/1e3:.1f} us")
print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
for name, n, tot, avg, mn, mx in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    short = re.sub(r"\s+", " ", name)[:150]
    print(f"{n:7d} {tot/1e3:12.1f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}  {short}")
