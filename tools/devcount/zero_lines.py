#!/usr/bin/env python
"""Do the fabric counters see lines of zeros?  (profiles/r04_copy_content.log: zeros move 3-14 % faster than any other content.)

    ROCP_TOOL_LIBRARIES=$PWD/tools/devcount/libgymrs_devcount.so python tools/devcount/zero_lines.py --counters FETCH_SIZE

Device-wide FETCH_SIZE / WRITE_SIZE (one per process) around torch device-to-device copies of 1 GiB and of 64 MiB (fits the Infinity
Cache) filled with zeros / ones / random floats, in the order A B A.  Equal counts = whatever makes zeros faster sits BEHIND the point
where the L2s hand requests to the fabric (Infinity Cache, memory controllers); fewer = the L2s already treat them differently.
"""
import argparse
import ctypes as C
import json
from pathlib import Path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--counters", default="FETCH_SIZE")
    args = ap.parse_args()
    import torch

    dc = C.CDLL(str(Path(__file__).resolve().parent / "libgymrs_devcount.so"))
    dc.gymrs_devcount_error.restype = C.c_char_p
    names = [c for c in args.counters.split(",") if c]
    out = (C.c_double * len(names))()

    def sample():
        if dc.gymrs_devcount_sample(out, len(names)) < 0:
            raise SystemExit("sample: " + dc.gymrs_devcount_error().decode())
        return [out[i] for i in range(len(names))]

    torch.cuda.init()
    dev = torch.device("cuda", 0)
    rec = {"counters": names, "rows": []}
    for label, words in (("1 GiB", 1 << 28), ("64 MiB", 1 << 24)):
        src = torch.empty(words, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        torch.cuda.synchronize()
        if label == "1 GiB" and dc.gymrs_devcount_start(args.counters.encode()) != 0:
            raise SystemExit("start: " + dc.gymrs_devcount_error().decode())
        copies = 20 if words == 1 << 28 else 400
        for fill in ("zeros", "ones", "zeros", "random", "zeros"):
            if fill == "zeros":
                src.zero_()
            elif fill == "ones":
                src.fill_(1.0)
            else:
                src.uniform_(-2.4, 2.4)
            dst.zero_()
            for _ in range(3):
                dst.copy_(src)
            torch.cuda.synchronize()
            a = sample()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(copies):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            b = sample()
            rec["rows"].append({"buffer": label, "content": fill, "copies": copies, "us_per_copy": e0.elapsed_time(e1) * 1e3 / copies,
                                "counter_per_copy": [(y - x) / copies for x, y in zip(a, b)]})
    dc.gymrs_devcount_stop()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
