"""VGG19 conv1_1 (3 -> 64, channels zero-padded to 16) and conv1_2 (64 -> 64) at the C4 shapes (36 images 256 x 256), forward and input gradient
(the loss networks are frozen: no weight gradient): the split-arithmetic kernels with the row gate opened vs the library (MIOpen), per arithmetic mode.
    python tools/probes/vgg_first_layers.py   -> JSON lines"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from styl3r_amd import vit_ops
from styl3r_amd.vit_ops import Conv2dX6

dev = "cuda:0"


def timeit(fn, iters=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B, H = 36, 256
for name, Ci, Co in (("conv1_2 64>64", 64, 64), ("conv1_1 3>64 as 16>64", 16, 64)):
    conv = Conv2dX6(Ci, Co, 3, padding=1).to(dev).requires_grad_(False)
    x = torch.randn(B, Ci, H, H, device=dev, requires_grad=True)
    g = torch.randn(B, Co, H, H, device=dev)
    row = dict(layer=name, B=B)
    for tag, rows in (("lib", 10 ** 9), ("x6", 16)):
        vit_ops._CONV_X6_MIN_ROWS = rows
        for mode in (("bf16x6",) if tag == "lib" else ("bf16x6", "bf16x3", "f16x3")):
            vit_ops.LINEAR_MODE = mode
            f = timeit(lambda: conv(x))
            fb = timeit(lambda: torch.autograd.grad(conv(x), x, g))
            row[f"{tag}:{mode}"] = dict(fwd_us=round(1e3 * f, 1), dx_us=round(1e3 * (fb - f), 1))
    vit_ops._CONV_X6_MIN_ROWS = 96; vit_ops.LINEAR_MODE = "bf16x6"; vit_ops._x6()
    print(json.dumps(row), flush=True)
