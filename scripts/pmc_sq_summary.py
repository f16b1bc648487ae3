"""Per-kernel SQ counter sums from a rocprofv3 --pmc pass (rocpd sqlite): scripts/pmc_sq_summary.py <out dir> [pass dir name].

Normalisation.  rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs of an MI355X (each XCD has its own GRBM) and the SQ
counters summed over all CUs / SIMDs.  Per-SIMD utilisation therefore divides by (GRBM_GUI_ACTIVE / 8) * 1024 SIMDs =
GRBM_GUI_ACTIVE * 128.  (Round 1 divided by GRBM_GUI_ACTIVE * 1024 and printed 0.07 for kernels that run at 100 of
157 TFLOP/s fp32 -- impossible: a kernel doing 64 % of the MFMA peak must show >= 64 % MFMA-busy.  The corrected figure,
0.57-0.63 there, brackets the FLOP-derived one.)  SQ_VALU_MFMA_BUSY_CYCLES counts cycles; SQ_WAVE_CYCLES, SQ_WAIT_*,
SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md), so those are shown as fractions of SQ_WAVE_CYCLES."""
import glob
import re
import sqlite3
import sys

out = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "pmc_SQ"
dbs = glob.glob(f"{out}/{sub}/**/*.db", recursive=True)
con = sqlite3.connect(dbs[0])
cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
res = {}
for name, cn, n, v in con.execute(f"select {namecol}, counter_name, count(*), sum(value) from counters_collection group by {namecol}, counter_name"):
    res.setdefault(re.sub(r"\s+", " ", name)[:100], {})[cn] = (n, v)
ctrs = sorted({c for v in res.values() for c in v})
rows = sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", (0, 0))[1])
print("# sums over all launches of each kernel; see the module docstring for the normalisation")
print(f"{'kernel':100s} {'n':>5} " + " ".join(f"{c[:18]:>19}" for c in ctrs) + "   mfma_busy   frac_of_wave_cycles: " + " ".join(c[3:][:14] for c in ctrs if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE")))
for k, v in rows[:60]:
    vals = [v.get(c, (0, 0))[1] for c in ctrs]
    g = v.get("GRBM_GUI_ACTIVE", (0, 1))[1] or 1
    wc = v.get("SQ_WAVE_CYCLES", (0, 0))[1] or 1
    util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1] / (g * 128.0)
    fr = " ".join(f"{v[c][1] / wc:14.3f}" for c in ctrs if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE"))
    print(f"{k:100s} {v.get('GRBM_GUI_ACTIVE', (0,0))[0]:5d} " + " ".join(f"{x:19.5g}" for x in vals) + f"   {util:9.3f}   {fr}")
