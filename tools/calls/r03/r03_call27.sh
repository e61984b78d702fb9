#!/bin/bash
set -u
OUT=gpurun_out/r03_c27
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  GYMRS_AQL_SYNC=1 timeout 300 rocprofv3 --pmc $ctr -d $GRAFT_REPO_ROOT/$OUT/pmc_$ctr -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --cpu-seconds 0 --no-probe --no-configs --repetitions 2 > $GRAFT_REPO_ROOT/$OUT/bench_$ctr.json 2> $GRAFT_REPO_ROOT/$OUT/bench_$ctr.err
  echo "rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, sqlite3, glob
for ctr in ("FETCH_SIZE","WRITE_SIZE"):
    try:
        d=json.load(open(f"gpurun_out/r03_c27/bench_{ctr}.json")); print(ctr, "submission:", d["config"]["submission"][:70], d["roofline"]["launch_us"])
    except Exception as e: print(ctr, "bench json:", e)
    for db in glob.glob(f"gpurun_out/r03_c27/pmc_{ctr}/**/*.db", recursive=True):
        c=sqlite3.connect(db)
        for r in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name order by 2 desc limit 6",(ctr,)): print("   ", r[0][:60], r[1], r[2])
PY
rm -rf $OUT/pmc_*
