#!/bin/bash
# SQ counter passes over scripts/experiments/x3_prof.py (one f32x3 GEMM shape); run on the GPU box from the repo root
OUT=${1:-gpurun_out/x3_prof}; shift
ARGS="$@"
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $R/$OUT/pmcA -o pmc -- python $R/scripts/experiments/x3_prof.py $ARGS > $R/$OUT/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES -d $R/$OUT/pmcB -o pmc -- python $R/scripts/experiments/x3_prof.py $ARGS > $R/$OUT/pmcB.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmcC -o pmc -- python $R/scripts/experiments/x3_prof.py $ARGS > $R/$OUT/pmcC.log 2>&1
cd $R
python - <<PY
import glob, sqlite3, re
for sub in ("pmcA", "pmcB", "pmcC"):
    dbs = glob.glob(f"$OUT/{sub}/**/*.db", recursive=True)
    if not dbs:
        print(sub, "no db"); continue
    con = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    t = [x for x in tabs if x.startswith("counters_collection")][0]
    cols = [d[1] for d in con.execute(f"pragma table_info({t})")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    res = {}
    for name, cn, n, v in con.execute(f"select {namecol}, counter_name, count(*), sum(value) from {t} group by {namecol}, counter_name"):
        res.setdefault(re.sub(r"\s+", " ", name)[:70], {})[cn] = (n, v)
    for k, v in res.items():
        if "x3" not in k: continue
        n = max(x[0] for x in v.values())
        print(sub, k, "launches", n, {c: round(x[1] / n) for c, x in sorted(v.items())})
PY
rm -rf $OUT/pmcA $OUT/pmcB $OUT/pmcC
