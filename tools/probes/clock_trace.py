#!/usr/bin/env python
"""Throughput of the headline raster step in consecutive 25-step chunks from a cold start (is the first second slow? does it sag later?)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
args = bench.parse([])
args.steps, args.warmup, args.prewarm_seconds = 1, 0, 0.0
dev = torch.device("cuda:0")
# reuse the bench's own step by monkeypatching timed_steps to hand the step function back
from styl3r_amd import dist_utils
orig = dist_utils.timed_steps
def grab(step, n, sync, dist, dev_):
    time.sleep(3.0)                       # idle: clocks down
    t0 = time.perf_counter(); out = []
    for chunk in range(80):
        torch.cuda.synchronize(); a = time.perf_counter()
        for _ in range(25): step()
        torch.cuda.synchronize(); b = time.perf_counter()
        out.append((round(a - t0, 2), round(40 * 25 / (b - a))))
    print(" ".join(f"{t}s:{v}" for t, v in out))
    return orig(step, n, sync, dist, dev_)
dist_utils.timed_steps = grab
bench.raster_leg(args, 0, 1, dev, None)
