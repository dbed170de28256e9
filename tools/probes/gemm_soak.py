#!/usr/bin/env python
"""Race screen for the LDS-DMA ring kernels: every configuration, several shapes, hundreds of launches each under a busy device,
every output compared bit for bit with the default kernel's."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
bad = 0
side = torch.cuda.Stream()
noise = torch.randn(4096, 4096, device=dev)
for (M, N, K) in [(5140, 3072, 1024), (1300, 1024, 4096), (777, 200, 48), (4096, 4096, 512), (257, 768, 768)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    vit_ops.LINEAR_MODE = "bf16x6"
    with torch.no_grad():
        want = vit_ops.fused_linear(x, w, b)
        if M * N <= 514 * 1024 * 8:      # the default path may have taken its split-K branch (atomics): compare with tolerance there
            exact = False
        else:
            exact = True
    blk = vit_ops.split_weight_block(w)
    for cfg in (1, 2, 3):
        for rep in range(200):
            if rep % 20 == 0:
                with torch.cuda.stream(side):          # unrelated traffic on another stream: shifts DMA / LDS timing
                    noise @ noise
            got = vit_ops.linear_x6r(x, blk, N, bias=b, cfg=cfg)
            ok = torch.equal(got, want) if exact else bool(((got - want).abs().max() <= 1e-5 * want.abs().max()).item())
            if not ok:
                bad += 1
                print("MISMATCH", (M, N, K), cfg, rep, float((got - want).abs().max()))
                break
    print((M, N, K), "ok" if not bad else "BAD", flush=True)
print("soak result:", "clean" if bad == 0 else f"{bad} mismatching configurations")
