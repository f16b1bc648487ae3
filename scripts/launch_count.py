"""Kernel launches per step from a rocprofv3 kernel trace (rocpd sqlite): launches of all kernels / number of steps run."""
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
con = sqlite3.connect(db)
n, tot = con.execute("select count(*), sum(duration) from kernels").fetchone()
print(f"{n} kernel launches over {steps} steps (warm-up included) = {n / steps:.0f} per step; kernel time {tot / 1e6 / steps:.2f} ms per step")
