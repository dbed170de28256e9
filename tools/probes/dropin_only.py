#!/usr/bin/env python
"""bench.py's drop-in leg alone (for rocprofv3 runs): python tools/probes/dropin_only.py [--steps 6] [--stub]"""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=6); ap.add_argument("--stub", action="store_true"); a = ap.parse_args()
args = bench.parse([])
dev = torch.device("cuda:0")
scenes, g, cams = bench.build_batch(args, 0, dev)
for t in (g.means, g.covariances, g.harmonics, g.opacities):
    t.requires_grad_(True)
target = torch.rand((args.scenes, args.views, 3, args.res, args.res), device=dev)
print(bench.dropin_leg(args, dev, g, cams, target, steps=a.steps, warmup=2, stub=a.stub))
