"""3x3 convolutions of the DPT heads at the C3 shapes, forward (= the dX launch's kernel, too), per arithmetic mode: run once as is (the halo
kernel k_conv3h_x6 for W >= 32) and once with VIT_CONV3=taps (k_conv_x6: nine taps = nine K slabs) and compare.
    python tools/probes/conv3_halo.py [B ...]   -> JSON lines"""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from styl3r_amd import vit_ops
from styl3r_amd.vit_ops import Conv2dX6

dev = "cuda:0"
LAYERS = [("rn1 96>256 @64", 96, 256, 64), ("rn2 192>256 @32", 192, 256, 32), ("rcu 256>256 @32", 256, 256, 32), ("rcu 256>256 @64", 256, 256, 64),
          ("rcu 256>256 @128", 256, 256, 128), ("head0 256>128 @128", 256, 128, 128), ("head2 128>128 @256", 128, 128, 256), ("gs head0 256>256 @256", 256, 256, 256)]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for B in [int(a) for a in sys.argv[1:]] or [20]:
    for name, Ci, Co, H in LAYERS:
        conv = Conv2dX6(Ci, Co, 3, 1, 1).to(dev)
        x = torch.randn(B, Ci, H, H, device=dev)
        row = dict(layer=name, B=B, kernel=os.environ.get("VIT_CONV3", "halo"), GF=round(2e-9 * B * H * H * Co * Ci * 9, 1))
        with torch.no_grad():
            for mode in ("bf16x6", "bf16x3", "f16x3"):
                vit_ops.LINEAR_MODE = mode
                ms = timeit(lambda: conv.forward_fused(x))
                row[mode] = dict(us=round(1e3 * ms, 1), TF=round(row["GF"] / ms, 1))
        vit_ops.LINEAR_MODE = "bf16x6"; vit_ops._x6()
        print(json.dumps(row), flush=True)
