#!/bin/bash
# fabric-side bytes per launch where every array streams from HBM (CartPole, 2^24 lanes: 637.5 MB algorithmic per step), both submission paths:
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (FETCH_SIZE doubled per the guide's gfx950 correction), counters only
set -u
OUT=gpurun_out/r03_c52; mkdir -p $OUT; REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for path in hip chain; do
  if [ $path = hip ]; then export GYMRS_AQL=0; unset GYMRS_AQL_SYNC; else export GYMRS_AQL=1 GYMRS_AQL_SYNC=1; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc24_${path}_$ctr
    timeout 600 rocprofv3 --pmc $ctr -d /tmp/pmc24_${path}_$ctr -o r -- python $REPO/bench.py --n-envs 16777216 --steps 50 --warmup 10 --cpu-seconds 0 --no-probe --no-configs --repetitions 2 > $REPO/$OUT/${path}_$ctr.json 2> $REPO/$OUT/${path}_$ctr.err
  done
done
python - <<'PY'
import sqlite3, glob
alg = 38 * (1 << 24)
print("# CartPole 2^24 lanes, algorithmic bytes per launch %.1f MB (38 B per lane)" % (alg / 1e6))
for path, pat in (("hip", "*step_kernel*"), ("chain", "gymrs_aql_cartpole_f[0-9]_t*")):
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        dbs = glob.glob(f"/tmp/pmc24_{path}_{ctr}/**/*_results.db", recursive=True)
        if not dbs:
            print(path, ctr, "no database"); continue
        c = sqlite3.connect(dbs[0])
        n, avg = c.execute("select count(*), avg(value) from counters_collection where kernel_name glob ? and counter_name = ?", (pat, ctr)).fetchone()
        vals[ctr] = (n, avg)
    if len(vals) == 2 and vals["FETCH_SIZE"][0] and vals["WRITE_SIZE"][0]:
        fetch = 2.0 * vals["FETCH_SIZE"][1] * 1024.0
        write = vals["WRITE_SIZE"][1] * 1024.0
        print("%-5s launches counted %d / %d   fetch %.1f MB (FETCH_SIZE x 2)   write %.1f MB   total %.1f MB = %.3f x algorithmic   (read 17 B, write 21 B per lane: %.1f / %.1f MB)"
              % (path, vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0], fetch / 1e6, write / 1e6, (fetch + write) / 1e6, (fetch + write) / alg, 17 * (1 << 24) / 1e6, 21 * (1 << 24) / 1e6))
    else:
        print(path, vals)
PY
