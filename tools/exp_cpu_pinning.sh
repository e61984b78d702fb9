#!/bin/bash
# Evidence for sharded.pin_rank_to_cpus: the driver's own bench command, alternately pinned (default) and not, 8 times each;
# then the step timer under taskset, CartPole and MountainCar.   bash tools/exp_cpu_pinning.sh > gpurun_out/<tag>_cpu_pinning.log
show='import json,sys; d=json.loads(sys.stdin.read()); print("%.4g env-steps/s  %.3f us/launch  cpus %s" % (d["value"], d["roofline"]["launch_us"], d["timing"]["cpu_affinity"]))'
for i in 1 2 3 4 5 6 7 8; do
    echo -n "pinned    "; python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-probe 2>/dev/null | python -c "$show"
    echo -n "unpinned  "; GYMRS_NO_CPU_PIN=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-probe 2>/dev/null | python -c "$show"
done
for i in 1 2 3 4; do
    for env in 0 1; do
        echo -n "env $env taskset 4-7  "; taskset -c 4-7 python tools/step_timer.py --env $env --reps 3 2>/dev/null | tail -1
        echo -n "env $env unpinned     "; python tools/step_timer.py --env $env --reps 3 2>/dev/null | tail -1
    done
done
