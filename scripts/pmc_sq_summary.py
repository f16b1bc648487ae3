"""Per-kernel SQ counter sums from a rocprofv3 --pmc pass (rocpd sqlite)."""
import glob, re, sqlite3, sys
out = sys.argv[1]
dbs = glob.glob(f"{out}/pmc_SQ/**/*.db", recursive=True)
con = sqlite3.connect(dbs[0])
cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
res = {}
for name, cn, n, v in con.execute(f"select {namecol}, counter_name, count(*), sum(value) from counters_collection group by {namecol}, counter_name"):
    res.setdefault(re.sub(r"\s+", " ", name)[:100], {})[cn] = (n, v)
ctrs = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT"]
rows = sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", (0, 0))[1])
print("# sums over all launches of each kernel (bench.py --steps 2 --warmup 1); MFMA_BUSY counts cycles summed over SIMDs,")
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)")
print(f"{'kernel':100s} {'n':>5} " + " ".join(f"{c[3:][:14]:>15}" for c in ctrs) + "  mfma_util")
for k, v in rows[:25]:
    vals = [v.get(c, (0, 0))[1] for c in ctrs]
    g = v.get("GRBM_GUI_ACTIVE", (0, 1))[1] or 1
    util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1] / (g * 1024.0)
    print(f"{k:100s} {v.get('GRBM_GUI_ACTIVE', (0,0))[0]:5d} " + " ".join(f"{x:15.4g}" for x in vals) + f"  {util:8.3f}")
