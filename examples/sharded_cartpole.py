#!/usr/bin/env python
"""The loop of the reference's examples/cartpole.rs:15-30 (random action, step, reset on done, sum the rewards) for ONE batch of envs cut over all the
GPUs of the node, driven from ONE process through the C ABI's native sharder (include/gymrs_amd.h, gymrs_sharded_*): one engine and one native host
thread per GPU, lanes keep their global ids, so the result does not depend on how many GPUs there are; the episode statistics of the whole batch come
back through one RCCL all-reduce (a host-side sum when the blocks share a GPU).

    python examples/sharded_cartpole.py [lanes_per_gpu] [blocks]     # needs an MI355X; blocks defaults to the number of GPUs
"""
import importlib
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402  (first, so the extension shares torch's HIP runtime; used for the action buffers only)

gymrs = importlib.import_module("gym-rs_amd")


def main(lanes_per_gpu: int = 1 << 20, blocks: int = 0, steps: int = 475, ring: int = 8):
    n_dev = torch.cuda.device_count()
    blocks = blocks or n_dev
    devices = [r % n_dev for r in range(blocks)]
    sh = gymrs.ShardedEngine(gymrs.CARTPOLE, lanes_per_gpu * blocks, devices, flags=gymrs.AUTO_RESET | gymrs.TRACK_STATS)
    sh.reset(seed=0)
    # a ring of pre-drawn random-policy action buffers per block, on the block's own device (`rng.gen_range(0..=1)`, examples/cartpole.rs:19)
    pitch = max(s.n_envs for s in sh.shards)
    rings = [torch.empty((ring, pitch), dtype=torch.uint8, device=f"cuda:{s.device}") for s in sh.shards]
    for b in range(ring):
        sh.fill_actions([r[b].data_ptr() for r in rings], seed=1, t=b)
    sh.step_many([r.data_ptr() for r in rings], pitch, ring, steps)  # asynchronous on every block's stream
    sh.sync()
    total = sh.stats()
    print(f"{blocks} block(s) on {n_dev} GPU(s), {lanes_per_gpu * blocks} envs x {steps} steps: {int(total[2])} episodes, mean return "
          f"{total[0] / max(total[2], 1):.2f}, statistics summed by: {sh.reduce_path}")
    for s in sh.shards:
        print(f"  block at lane {s.first_lane}: {s.n_envs} lanes on cuda:{s.device}")
    sh.close()
    return total


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a[:2])
