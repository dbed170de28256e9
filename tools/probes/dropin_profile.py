#!/usr/bin/env python
"""Where a per-view `GaussianRasterizer` call spends its host time: the reference's call pattern of bench.py's drop-in leg
under cProfile, and the same loop with the rasterizer stubbed out (what the reference pattern costs by itself).
usage (GPU box): python tools/probes/dropin_profile.py [--steps 4]"""
import argparse, cProfile, pstats, sys, time, io
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=4); a = ap.parse_args()
args = bench.parse([])
dev = torch.device("cuda:0")
scenes, g, cams = bench.build_batch(args, 0, dev)
for t in (g.means, g.covariances, g.harmonics, g.opacities):
    t.requires_grad_(True)
target = torch.rand((args.scenes, args.views, 3, args.res, args.res), device=dev)
for stub in (False, True):
    r = bench.dropin_leg(args, dev, g, cams, target, steps=a.steps, warmup=2, stub=stub)
    print("stub" if stub else "module", r["ms_per_step"], "ms/step", r["ms_per_step"] / 40, "ms/call")
pr = cProfile.Profile()
pr.enable()
bench.dropin_leg(args, dev, g, cams, target, steps=a.steps, warmup=0)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
