#!/bin/bash
# rocprofv3 passes over scripts/experiments/attn2_prof.py (run on the GPU box from the repo root): kernel durations, then two SQ counter passes.
OUT=${1:-gpurun_out/attn2_prof}; shift
ARGS="$@"
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o kt -- python $R/scripts/experiments/attn2_prof.py $ARGS > $R/$OUT/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $R/$OUT/pmcA -o pmc -- python $R/scripts/experiments/attn2_prof.py $ARGS > $R/$OUT/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES -d $R/$OUT/pmcB -o pmc -- python $R/scripts/experiments/attn2_prof.py $ARGS > $R/$OUT/pmcB.log 2>&1
cd $R
python - <<PY
import glob, sqlite3, re, csv
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.reader(open(f)))[:8]:
        print(" | ".join(x[:70] for x in row[:6]))
for sub in ("pmcA", "pmcB"):
    dbs = glob.glob(f"$OUT/{sub}/**/*.db", recursive=True)
    if not dbs:
        print(sub, "no db"); continue
    con = sqlite3.connect(dbs[0])
    cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    res = {}
    for name, cn, n, v in con.execute(f"select {namecol}, counter_name, count(*), sum(value) from counters_collection group by {namecol}, counter_name"):
        res.setdefault(re.sub(r"\s+", " ", name)[:60], {})[cn] = (n, v)
    for k, v in res.items():
        if "attn2" not in k: continue
        n = max(x[0] for x in v.values())
        print(sub, k, "launches", n, {c: round(x[1] / n) for c, x in sorted(v.items())})
PY
rm -rf $OUT/kt $OUT/pmcA $OUT/pmcB
