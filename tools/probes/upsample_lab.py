#!/usr/bin/env python
"""x2 bilinear resampling kernels at the DPT-head shapes of a C3 step: us per call and GB/s of the algorithmic traffic (in + out)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
dev = "cuda:0"
for (B, C, H) in ((20, 256, 8), (20, 256, 16), (20, 256, 32), (20, 256, 64), (20, 128, 128), (10, 128, 128)):
    x = torch.randn(B, C, H, H, device=dev, requires_grad=True)
    y = vit_ops.upsample2x(x); g = torch.randn_like(y)
    def t(fn, n=30):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    with torch.no_grad():
        f_us = t(lambda: vit_ops.upsample2x(x))
    b_us = t(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    bytes_ = 5 * x.numel() * 4
    print(f"B={B} C={C} {H}->{2*H}: fwd {f_us:7.1f} us ({bytes_ / f_us / 1e3:6.0f} GB/s)   bwd {b_us:7.1f} us ({bytes_ / b_us / 1e3:6.0f} GB/s)")
