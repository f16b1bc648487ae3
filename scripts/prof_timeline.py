"""One training step of a rocprofv3 rocpd kernel trace as a timeline: python scripts/prof_timeline.py <prof_results.db> [step_index_from_end]
Prints, for the chosen step (steps are cut at clip_adam_kernel launches), every kernel as `start_us  dur_us  gap_us  name` plus phase
totals (backbone fwd / encoder fwd / decoder fwd / heads+loss / decoder bwd / encoder bwd / backbone bwd / optimiser) cut at marker kernels."""
import re
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
tcol = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else None)
if tcol is None:
    print("columns:", cols)
    raise SystemExit("no start column")
ecol = "end" if "end" in cols else "end_timestamp"
rows = con.execute(f"select name, {tcol}, {ecol} from kernels order by {tcol}").fetchall()
short = lambda n: re.sub(r"\s+", " ", re.sub(r"\(.*", "", n)).replace("void detr::", "").replace("detr::", "")[:70]
# steps end with the last clip_adam_kernel of a group of (up to) 3
ends = [i for i, r in enumerate(rows) if "clip_adam" in r[0] and (i + 1 == len(rows) or "clip_adam" not in rows[i + 1][0]) and
        not any("clip_adam" in rows[j][0] for j in range(i + 1, min(i + 4, len(rows))))]
if len(ends) < back + 1:
    raise SystemExit(f"only {len(ends)} steps found")
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = step[0][1]
print(f"# step of {len(step)} launches, wall {((step[-1][2] - t0) / 1e3):.1f} us, kernel sum {sum(r[2] - r[1] for r in step) / 1e3:.1f} us")
prev_end = t0
phase_marks = []
for name, s, e in step:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {short(name)}")
    prev_end = max(prev_end, e)
