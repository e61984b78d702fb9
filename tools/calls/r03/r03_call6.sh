#!/bin/bash
set -u
OUT=gpurun_out/r03_c6
mkdir -p $OUT
export TMPDIR=/tmp
for f in off a0 a1 00 01 11 22; do
  echo "== cartpole 2^20: $f" >> $OUT/timing.log
  if [ $f = off ]; then
    GYMRS_AQL=0 timeout 300 python tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 7 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/timing.log
  else
    f2=${f/a/1}
    GYMRS_AQL=1 GYMRS_AQL_FENCES=$f2 timeout 300 python tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 7 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/timing.log
  fi
done
cat $OUT/timing.log
cd /tmp && GYMRS_AQL=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 3 > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3,glob,statistics
for db in glob.glob("gpurun_out/r03_c6/prof/**/*.db", recursive=True):
    c=sqlite3.connect(db)
    for r in c.execute("select name, count(*), avg(end-start) from kernels group by name order by 2 desc limit 8"): print(r)
    ks=c.execute("select start,end from kernels where name like 'gymrs_aql_cartpole%' order by start").fetchall()
    if len(ks)>100:
        ks=ks[len(ks)//2:]
        d=[e-s for s,e in ks]; gaps=[ks[i+1][0]-ks[i][1] for i in range(len(ks)-1)]
        print("aql cartpole: n",len(ks),"median dur",statistics.median(d),"mean",statistics.mean(d),"median gap",statistics.median(gaps),"mean gap",statistics.mean(gaps), "period", (ks[-1][0]-ks[0][0])/(len(ks)-1))
PY
find $OUT/prof -name "*.db" -size +20M -delete
echo done6
