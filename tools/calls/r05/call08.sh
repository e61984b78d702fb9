#!/bin/bash
# round 5, call 8: VERDICT r4 "next" #5 (b) -- CartPole 2^21 lanes: 256-work-item groups (the 512 window ends below 2^21), plain loads + streamed stores (hints 10),
# state loads + outputs streamed (hints 9), and their combinations, developer builds of ONE tree timed alternately in one process against the product kernel
# (r05_base = the same sources with no extra flag); HIP launches and chains; 2^20 and 2^22 beside it so that nothing else regresses.  Then the new parity test.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
L=gpurun_out/r05/cartpole_2p21.log
: > $L
LIBS="--lib _ab/libr05_base.so --lib _ab/libr05_b256.so --lib _ab/libr05_h10.so --lib _ab/libr05_b256_h10.so --lib _ab/libr05_h9.so --lib _ab/libr05_b256_h9.so"
for lg in 21 20 22; do for q in 0 1; do
  echo "# 2^$lg CartPole lanes, GYMRS_AQL=$q ($([ $q = 0 ] && echo 'HIP launches: per-step visible' || echo chains)), 8 action buffers, 7 repetitions of 6000 steps" >> $L
  GYMRS_AQL=$q timeout 900 python tools/step_timer.py $LIBS --env 0 --n $((1<<lg)) --steps 6000 --reps 7 --nbuf 8 2>&1 | grep "us median" >> $L
done; done
echo "# 2^21 lanes again with bench.py's ring of 32 action buffers" >> $L
for q in 0 1; do
  echo "# GYMRS_AQL=$q" >> $L
  GYMRS_AQL=$q timeout 900 python tools/step_timer.py $LIBS --env 0 --n $((1<<21)) --steps 6000 --reps 7 --nbuf 32 2>&1 | grep "us median" >> $L
done
cat $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envs.py -x -q -k "golden or example or cpp" 2>&1 | tail -4 | tee gpurun_out/r05/pytest_call08.log
# the bisect of the 0.4 us (profiles/r05_visible_through_queue.log part 6): the CHAIN'S binary launched through the HIP runtime's queue (dev hook 8) against the
# library's own HIP kernel (hooks 0), both GYMRS_AQL=0; and against the queue submissions of the same binary
L2=gpurun_out/r05/visible_through_queue_bisect.log
: > $L2
python - >> $L2 2>&1 <<'PY'
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
gymrs = importlib.import_module("gym-rs_amd")
os.environ["GYMRS_AQL"] = "0"
lib = gymrs.load_library(); lib.gymrs_dev_set_hooks.argtypes = [C.c_void_p, C.c_uint32]
for kind, n in ((0, 1 << 20), (1, 100003), (2, 50001)):
    flags = 3 | (4 if kind == 2 else 0)
    a, b = gymrs.BatchedEngine(kind, n, flags=flags), gymrs.BatchedEngine(kind, n, flags=flags)
    assert lib.gymrs_dev_set_hooks(b._h, 8) == 0
    ring = torch.empty((4, n), dtype=torch.float32 if kind == 2 else torch.uint8, device="cuda:0")
    for e in (a, b):
        e.reset(seed=5)
    for t in range(4):
        a.fill_actions(ring[t].data_ptr(), 1, t)
    esz = 4 if kind == 2 else 1
    a.step_many(ring.data_ptr(), n * esz, 4, 300); b.step_many(ring.data_ptr(), n * esz, 4, 300)
    a.sync(); b.sync()
    same = np.array_equal(a.get_state().view(np.uint32), b.get_state().view(np.uint32)) and np.array_equal(a.stats(), b.stats())
    print(f"# kind {kind}: 300 steps through the library's HIP kernel and through the chain's binary launched by hipModuleLaunchKernel: {'bit-identical' if same else 'DIFFERENT'}")
    a.close(); b.close()
PY
for env in 0 1; do
  echo "# env $env 2^20 lanes, 32 action buffers: hooks=0 the library's own HIP kernel, hooks=8 the chain's binary through HIP's queue (both GYMRS_AQL=0); then GYMRS_AQL=2" >> $L2
  timeout 600 python tools/step_timer.py --env $env --n $((1<<20)) --steps 16000 --reps 7 --aql 0 --hooks 0,8 --nbuf 32 2>&1 | grep "us median" >> $L2
  timeout 600 python tools/step_timer.py --env $env --n $((1<<20)) --steps 16000 --reps 7 --aql 2 --nbuf 32 2>&1 | grep "us median" >> $L2
done
cat $L2
# per-step calls through the native sharder's mailbox (one command per block per step) against gymrs_step on the engine directly: 2^20 CartPole lanes, 20000 steps
python - 2>&1 <<'PY' | grep -v amdgpu.ids | tee gpurun_out/r05/sharded_per_step_calls.log
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
gymrs = importlib.import_module("gym-rs_amd")
lib = gymrs.load_library()
n, steps, flags = 1 << 20, 20000, 3
ring = torch.empty((8, n), dtype=torch.uint8, device="cuda:0")
one = gymrs.BatchedEngine(0, n, flags=flags); one.reset(seed=0)
for b in range(8):
    one.fill_actions(ring[b].data_ptr(), 1, b)
one.sync()
def loop(call, sync):
    for t in range(2000):
        call(t)
    sync()
    t0 = time.perf_counter()
    for t in range(steps):
        call(t)
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e6 / steps, (t2 - t0) * 1e6 / steps
ptrs = [C.c_void_p(ring[b].data_ptr()) for b in range(8)]
h = one._h
e, w = loop(lambda t: lib.gymrs_step(h, ptrs[t & 7]), one.sync)
print(f"gymrs_step on the engine, calling thread launches itself:       {e:6.2f} us per call enqueued, {w:6.2f} us per step incl. the final wait")
one.close()
for k in (1, 2, 4):
    sh = gymrs.ShardedEngine(0, n, [0] * k, flags=flags); sh.reset(seed=0)
    rows = [(C.c_void_p * k)(*[C.c_void_p(ring[b].data_ptr() + s.first_lane) for s in sh.shards]) for b in range(8)]
    hs = sh._h
    e, w = loop(lambda t: lib.gymrs_sharded_step(hs, rows[t & 7]), sh.sync)
    print(f"gymrs_sharded_step, {k} block(s) on one GPU, one worker thread each: {e:6.2f} us per call enqueued, {w:6.2f} us per step incl. the final wait")
    sh.close()
PY
echo "# MountainCar / Pendulum: 8 lanes per work-item against 4 (HIP launches)" | tee -a gpurun_out/r05/cartpole_2p21.log
for env in 1 2; do for vec in 4 8; do
  timeout 600 python tools/step_timer.py --env $env --n $((env == 2 ? 1<<22 : 1<<20)) --steps 8000 --reps 5 --aql 0 --vec $vec --nbuf 8 2>&1 | grep "us median" | sed "s/^/env $env vec $vec: /" | tee -a gpurun_out/r05/cartpole_2p21.log
done; done
