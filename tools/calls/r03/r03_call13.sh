#!/bin/bash
set -u
OUT=gpurun_out/r03_c13
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_aql_chain.py -x -q -s 2>&1 | tail -12 > $OUT/pytest_aql.log
cat $OUT/pytest_aql.log
for lib in vplain vplain_nolog vh8 vh4 vh12 vh2 vh1; do
  for v in "GYMRS_AQL=0" "GYMRS_AQL_LAZY_WAIT=1"; do
    echo "== $lib $v" >> $OUT/ablate.log
    env $v timeout 100 python tools/step_timer.py --lib _ab/lib$lib.so --env 0 --n 1048576 --steps 2000 --reps 6 --wall 1 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/ablate.log
  done
done
cat $OUT/ablate.log
for env in 0 1 2; do
  for n in 1048576 2097152 4194304 16777216; do
    for v in "GYMRS_AQL=0" "GYMRS_AQL=1"; do
      for nt in 0 1 2; do
        if [ "$v" = "GYMRS_AQL=0" ] && [ $nt != 0 ]; then continue; fi
        echo "== main lib env $env n $n $v --nt $nt" >> $OUT/main.log
        env $v timeout 200 python tools/step_timer.py --env $env --n $n --steps 1000 --reps 6 --nt $nt 2>&1 | grep -v "amdgpu.ids\|ring at" >> $OUT/main.log
      done
    done
  done
done
cat $OUT/main.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/step_timer.py --env 0 --n 1048576 --steps 2000 --reps 3 > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1
echo "rocprof rc=$?"; tail -3 $GRAFT_REPO_ROOT/$OUT/rocprof.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3,glob,statistics
for db in glob.glob("gpurun_out/r03_c13/prof/**/*.db", recursive=True):
    c=sqlite3.connect(db)
    for r in c.execute("select name, count(*), avg(end-start) from kernels group by name order by 2 desc limit 8"): print(r)
    ks=c.execute("select start,end from kernels where name like 'gymrs_aql_cartpole%' order by start").fetchall()
    if len(ks)>100:
        ks=ks[len(ks)//2:]
        d=[e-s for s,e in ks]; gaps=[ks[i+1][0]-ks[i][1] for i in range(len(ks)-1)]
        print("aql cartpole: n",len(ks),"median dur",statistics.median(d),"mean",statistics.mean(d),"median gap",statistics.median(gaps),"mean gap",statistics.mean(gaps), "period", (ks[-1][0]-ks[0][0])/(len(ks)-1))
PY
find $OUT/prof -name "*.db" -size +20M -delete
echo done13
