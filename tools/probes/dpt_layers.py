"""Every convolution of the DPT heads at the C3 shapes (b = 10 scenes, v = 2: head calls of 10 and 20 images), forward and
forward + backward, on the hand-written bf16x6 kernels with their size gates forced open vs the library (MIOpen) path:
the data behind the gates in styl3r_amd/vit_ops.py (heads/dpt_block.py:79-218,350-419; dpt_gs_head.py:113-157).
    python tools/probes/dpt_layers.py [B ...]   -> JSON lines"""
import json
import sys

sys.path.insert(0, ".")
import torch

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


LAYERS = [  # name, Ci, Co, k, H(=W)
    ("rn1 3x3 96>256 @64", 96, 256, 3, 64), ("rn2 3x3 192>256 @32", 192, 256, 3, 32), ("rn3 3x3 384>256 @16", 384, 256, 3, 16),
    ("rn4 3x3 768>256 @8", 768, 256, 3, 8), ("rcu 3x3 256>256 @8", 256, 256, 3, 8), ("rcu 3x3 256>256 @16", 256, 256, 3, 16),
    ("rcu 3x3 256>256 @32", 256, 256, 3, 32), ("rcu 3x3 256>256 @64", 256, 256, 3, 64), ("out 1x1 256>256 @16", 256, 256, 1, 16),
    ("out 1x1 256>256 @32", 256, 256, 1, 32), ("out 1x1 256>256 @64", 256, 256, 1, 64), ("out 1x1 256>256 @128", 256, 256, 1, 128),
    ("head0 3x3 256>128 @128", 256, 128, 3, 128), ("head2 3x3 128>128 @256", 128, 128, 3, 256), ("gs head0 3x3 256>256 @256", 256, 256, 3, 256),
]
gates = ("_CONV_X6_MIN_TILES", "_CONV_X6_MIN_ROWS", "_CONV_X6_WGRAD_MIN_PIXELS")
keep = {g: getattr(vit_ops, g) for g in gates}
for B in [int(a) for a in sys.argv[1:]] or [10, 20]:
    for name, Ci, Co, k, H in LAYERS:
        conv = Conv2dX6(Ci, Co, k, 1, k // 2).to(dev)
        x = torch.randn(B, Ci, H, H, device=dev, requires_grad=True)
        g = torch.randn(B, Co, H, H, device=dev)
        row = dict(layer=name, B=B, GF=round(2e-9 * B * H * H * Co * Ci * k * k, 2), default_x6=bool(conv._x6_ok(x)))
        for tag, vals in (("x6", (0, 0, 0)), ("lib", (10 ** 9, 10 ** 9, 10 ** 12))):
            for gname, v in zip(gates, vals):
                setattr(vit_ops, gname, v)
            try:
                f = timeit(lambda: conv(x))
                fb = timeit(lambda: torch.autograd.grad(conv(x), (x, conv.weight, conv.bias), g))
                row[tag] = dict(fwd_us=round(1e3 * f, 1), bwd_us=round(1e3 * (fb - f), 1))
            except Exception as e:
                row[tag] = dict(error=str(e)[:120])
        for gname in gates:
            setattr(vit_ops, gname, keep[gname])
        print(json.dumps(row), flush=True)
