#!/bin/bash
set -u
OUT=gpurun_out/r03_c14
mkdir -p $OUT
export TMPDIR=/tmp
for env in 0 1 2; do
  for n in 1048576 2097152 4194304; do
    for lib in vplain vh1 vh8 vh9; do
      echo "== chain env $env n $n $lib" >> $OUT/hints_chain.log
      timeout 100 python tools/step_timer.py --lib _ab/lib$lib.so --env $env --n $n --steps 1000 --reps 5 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/hints_chain.log
    done
  done
done
cat $OUT/hints_chain.log
cd /tmp
for v in "GYMRS_AQL_FLUSH=1" "X=1"; do
  env $v timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_$v -o t -- python $GRAFT_REPO_ROOT/tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 3 > $GRAFT_REPO_ROOT/$OUT/rocprof_$v.log 2>&1
  echo "rocprof $v rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3,glob,statistics
for db in glob.glob("gpurun_out/r03_c14/prof_*/**/*.db", recursive=True):
    print(db)
    c=sqlite3.connect(db)
    for r in c.execute("select name, count(*), avg(end-start) from kernels group by name order by 2 desc limit 8"): print(r)
    ks=c.execute("select start,end from kernels where name like 'gymrs_aql_cartpole%' order by start").fetchall()
    if len(ks)>100:
        ks=ks[len(ks)//2:]
        d=[e-s for s,e in ks]; gaps=[ks[i+1][0]-ks[i][1] for i in range(len(ks)-1)]
        print("aql cartpole: n",len(ks),"median dur",statistics.median(d),"mean",statistics.mean(d),"median gap",statistics.median(gaps),"mean gap",statistics.mean(gaps), "period", (ks[-1][0]-ks[0][0])/(len(ks)-1))
PY
find $OUT -name "*.db" -size +20M -delete
echo done14
