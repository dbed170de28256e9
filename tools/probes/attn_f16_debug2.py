import sys; sys.path.insert(0, "/root/repo")
import torch
from styl3r_amd import vit_ops
DEV="cuda:0"
g = torch.Generator(DEV).manual_seed(0)
Nq, Nk = 32, 32
q = torch.randn(1, Nq, 1, 64, device=DEV, generator=g); k = torch.randn(1, Nk, 1, 64, device=DEV, generator=g); v = torch.randn(1, Nk, 1, 64, device=DEV, generator=g)
res = {}
for mode in ("bf16x6", "f16x3"):
    vit_ops.ATTENTION_ARITH = mode
    res[mode] = vit_ops.memory_efficient_attention(q, k, v, scale=0.125)
a, b = res["bf16x6"][0, :, 0], res["f16x3"][0, :, 0]
print("ref  ", a[0, :6].tolist()); print("f16  ", b[0, :6].tolist())
att = torch.softmax((q[0, :, 0] @ k[0, :, 0].T) * 0.125, -1)
# which keys are mis-weighted: solve b = att' @ v  -> att' = b @ pinv(v)
attp = b @ torch.linalg.pinv(v[0, :, 0])
d = (attp - att)
print("per-key weight error (query 0):", [round(x, 3) for x in d[0].tolist()])
print("per-key weight error abs max over queries:", [round(x, 3) for x in d.abs().amax(0).tolist()])
# test with V = identity-like to read P directly: v one-hot over first 32 dims
v2 = torch.zeros_like(v); v2[0, torch.arange(32), 0, torch.arange(32)] = 1.0
vit_ops.ATTENTION_ARITH = "f16x3"
p = vit_ops.memory_efficient_attention(q, k, v2, scale=0.125)[0, :, 0, :32]
print("P err with one-hot V, max over queries per key:", [round(x, 4) for x in (p - att).abs().amax(0).tolist()])
