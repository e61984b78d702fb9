#!/bin/bash
set -u
OUT=gpurun_out/r03_c16
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES -d $GRAFT_REPO_ROOT/$OUT/pmc -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --cpu-seconds 0 --no-probe --no-configs --repetitions 2 > $GRAFT_REPO_ROOT/$OUT/bench_under_pmc.json 2> $GRAFT_REPO_ROOT/$OUT/bench_under_pmc.err
cd $GRAFT_REPO_ROOT
python -c "
import json; d=json.load(open('$OUT/bench_under_pmc.json')); print('under --pmc:', d['config']['submission'], d['roofline']['launch_us'])"
grep -i "aql\|gymrs" $OUT/bench_under_pmc.err | head -5
rm -rf $OUT/pmc
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_time_limit_elision.py tests/test_gpu_aql_chain.py -q -x 2>&1 | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-probe --no-configs > $OUT/bench_driver_form.json 2>/dev/null
python -c "
import json; d=json.load(open('$OUT/bench_driver_form.json')); t=d['timing']; print('driver form:', d['value'], t['event_us_per_step'], t['calibration_calls'], t['settle_ms'], t['passes_per_repetition'])"
