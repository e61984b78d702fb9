"""Developer experiment: chains of two engines in one process (the second one ran 6x slower: why?)."""
import importlib, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
gymrs = importlib.import_module("gym-rs_amd")
n, nbuf, steps = 1 << 20, 8, 1000


def make():
    e = gymrs.BatchedEngine(0, n, flags=3)
    e.reset(seed=1)
    ring = torch.empty((nbuf, n), dtype=torch.uint8, device="cuda:0")
    for b in range(nbuf):
        e.fill_actions(ring[b].data_ptr(), seed=2, t=b)
    e.sync()
    return e, ring


def run(e, ring, label):
    ts = []
    for _ in range(3):
        e.sync(); t0 = time.perf_counter()
        e.step_many(ring.data_ptr(), n, nbuf, steps)
        e.sync(); ts.append((time.perf_counter() - t0) * 1e6 / steps)
    print(f"{label:50s} " + " ".join(f"{t:7.3f}" for t in ts), flush=True)


order = sys.argv[1] if len(sys.argv) > 1 else "ab"
A, ra = make()
if order == "late":  # A steps while it is the only engine; B and C appear later: A has to look at its hand-over again
    run(A, ra, "A alone")
    B, rb = make()
    C, rc = make()
    run(B, rb, "B (created after A stepped)"); run(C, rc, "C"); run(A, ra, "A again, now one of three"); run(B, rb, "B again")
    C.close(); B.close()
    run(A, ra, "A alone again")
    import json
    print("A", json.loads(A.env_json(0))["gymrs"].get("aql_handover"))
    sys.exit(0)
B, rb = make()
if order == "ab":
    run(A, ra, "A (created first), first chains"); run(B, rb, "B (created second)"); run(A, ra, "A again"); run(B, rb, "B again")
    A.close(); run(B, rb, "B after A was destroyed")
elif order == "ba":
    run(B, rb, "B (created second) steps FIRST"); run(A, ra, "A (created first) steps second"); run(B, rb, "B again"); run(A, ra, "A again")
elif order == "three":
    C, rc = make()
    run(A, ra, "A"); run(B, rb, "B"); run(C, rc, "C"); run(A, ra, "A"); run(C, rc, "C"); run(B, rb, "B")
import json
for nm, e in (("A", A), ("B", B)):
    try:
        print(nm, json.loads(e.env_json(0))["gymrs"].get("aql_handover"))
    except Exception as exc:
        print(nm, "closed" if "NULL" in str(exc) or True else exc)
