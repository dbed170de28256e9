import sys; sys.path.insert(0, "/root/repo")
import torch
from styl3r_amd import vit_ops
DEV="cuda:0"
def run(B,H,Nq,Nk,seed=0,scale=1.0):
    g = torch.Generator(DEV).manual_seed(seed)
    q = torch.randn(B, Nq, H, 64, device=DEV, generator=g)*scale; k = torch.randn(B, Nk, H, 64, device=DEV, generator=g)*scale
    v = torch.randn(B, Nk, H, 64, device=DEV, generator=g)*scale
    res = {}
    for mode in ("bf16x6", "f16x3"):
        vit_ops.ATTENTION_ARITH = mode
        res[mode] = vit_ops.memory_efficient_attention(q, k, v, scale=0.125)
    a, b = res["bf16x6"], res["f16x3"]
    print(B,H,Nq,Nk,seed,scale, "o max rel %.2e" % float((a-b).abs().max()/a.abs().max()), "amax q %.3f k %.3f v %.3f" % (float(q.abs().max()), float(k.abs().max()), float(v.abs().max())))
for nk in (8, 16, 24, 31, 32, 33, 40, 48, 63, 64, 65, 80, 95, 96, 97, 100, 127, 128, 129, 160, 192, 193, 224):
    run(1, 1, 64, nk)
