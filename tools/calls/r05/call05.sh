#!/bin/bash
# round 5, call 5: the ADVICE fixes on hardware (chain tests), then where the 0.4 us sit: rocprofv3 --kernel-trace of the same 2^20-lane CartPole launches
# through HIP (GYMRS_AQL=0) and through the engine's queue with HIP's header (GYMRS_AQL=2): kernel duration against start-to-start period
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_aql_chain.py tests/test_gpu_sharded_native.py tests/test_gpu_params_and_serde.py -x -q 2>&1 | tail -5 | tee gpurun_out/r05/pytest_call05.log
export TMPDIR=/tmp
for q in 0 2 1; do
  rm -rf /tmp/kt$q
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/kt$q -o r -- python $GRAFT_REPO_ROOT/tools/step_timer.py --env 0 --n $((1<<20)) --steps 3000 --reps 3 --aql $q --nbuf 32 > /tmp/kt$q.out 2>&1)
  db=$(find /tmp/kt$q -name "*_results.db" | head -1)
  echo "# GYMRS_AQL=$q  ($(grep 'us median' /tmp/kt$q.out | head -1))" | tee -a gpurun_out/r05/kernel_trace_by_submission.log
  python - "$db" <<'PY' | tee -a gpurun_out/r05/kernel_trace_by_submission.log
import sqlite3, statistics, sys
c = sqlite3.connect(sys.argv[1])
ks = c.execute("select name, start, end from kernels where (name like '%step_kernel%' or name like 'gymrs_aql_cartpole%') order by start").fetchall()
ks = ks[len(ks) // 3:]
durs = [e - s for _, s, e in ks]
per = [ks[i + 1][1] - ks[i][1] for i in range(len(ks) - 1)]
gap = [ks[i + 1][1] - ks[i][2] for i in range(len(ks) - 1)]
per = [p for p in per if p < 50000]
gap = [g for g in gap if g < 50000]
print(f"  {ks[0][0][:60]}: n={len(ks)} duration median {statistics.median(durs):.0f} ns (mean {statistics.mean(durs):.0f}), start-to-start median {statistics.median(per):.0f} ns, end-to-next-start median {statistics.median(gap):.0f} ns")
PY
done
