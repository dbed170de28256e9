#!/usr/bin/env python
"""Cross-check of the bf16x6 attention kernels against the exact-f32 ones on random shapes (forward and all three gradients; packed
qkv views and separate tensors; with and without fused RoPE; ragged query / key counts)."""
import random, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
dev = torch.device("cuda:0")
random.seed(7); torch.manual_seed(7)
worst = 0.0
for trial in range(120):
    B, H = random.choice([1, 2, 3, 5]), random.choice([1, 2, 3, 12, 16])
    Nq = random.choice([1, 5, 31, 32, 33, 64, 100, 128, 129, 130, 257, 260, 400, 514])
    Nk = random.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 129, 257, 300, 514, 771])
    rope = random.random() < 0.6
    packed = random.random() < 0.5 and Nq == Nk
    if packed:
        qkv = torch.randn(B, Nq, 3, H, 64, device=dev)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q = torch.randn(B, Nq, H, 64, device=dev); k = torch.randn(B, Nk, H, 64, device=dev); v = torch.randn(B, Nk, H, 64, device=dev)
    kw = {}
    if rope:
        qpos = torch.randint(0, 30, (B, Nq, 2), device=dev); kpos = torch.randint(0, 30, (B, Nk, 2), device=dev)
        kw = dict(qpos=qpos, kpos=kpos, max_pos=32)
    gw = torch.randn(B, Nq, H, 64, device=dev)
    res = {}
    for arith in ("f32", "bf16x6"):
        vit_ops.ATTENTION_ARITH = arith
        q_, k_, v_ = (t.detach().clone().requires_grad_(True) for t in (q, k, v)) if not packed else (None, None, None)
        if packed:
            base = qkv.detach().clone().requires_grad_(True)
            o = vit_ops.memory_efficient_attention(base[:, :, 0], base[:, :, 1], base[:, :, 2], 0.125, **kw)
            (o * gw).sum().backward()
            res[arith] = (o.detach(), base.grad[:, :, 0], base.grad[:, :, 1], base.grad[:, :, 2])
        else:
            o = vit_ops.memory_efficient_attention(q_, k_, v_, 0.125, **kw)
            (o * gw).sum().backward()
            res[arith] = (o.detach(), q_.grad, k_.grad, v_.grad)
    for name, a, b in zip(("out", "dq", "dk", "dv"), res["f32"], res["bf16x6"]):
        if Nk == 1 and name in ("dq", "dk"):      # one key: P = 1, dS = 0 exactly -- both kernels return rounding noise there
            continue
        # (a gradient that is pure cancellation noise -- dq with ONE key is exactly 0 -- is compared on the scale of its terms)
        e = float((a - b).abs().max() / torch.maximum(a.abs().max(), 0.05 * gw.abs().max()))
        worst = max(worst, e)
        if not (e <= 2e-5) or not bool(torch.isfinite(b).all()):
            print("MISMATCH", dict(B=B, H=H, Nq=Nq, Nk=Nk, rope=rope, packed=packed), name, e); sys.exit(1)
print("attention soak: 120 random shapes, worst relative difference f32 vs bf16x6 kernels", "%.2e" % worst)
