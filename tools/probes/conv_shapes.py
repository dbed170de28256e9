"""MIOpen fp32 conv timings at the DPT-head shapes (target numbers for a bf16x6 implicit-GEMM conv)."""
import torch, json, torch.nn.functional as F
dev = "cuda:0"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
res = {}
for name, (B, Ci, Co, H) in dict(rcu128=(16, 256, 256, 128), rcu64=(16, 256, 256, 64), rcu32=(16, 256, 256, 32), head0_256=(16, 256, 128, 256),
                                 head2_256=(16, 128, 32, 256), rn64=(16, 96, 256, 64)).items():
    x = torch.randn(B, Ci, H, H, device=dev, requires_grad=True); w = torch.randn(Co, Ci, 3, 3, device=dev, requires_grad=True) * 0.02
    fl = 2 * B * H * H * Co * Ci * 9
    f = timeit(lambda: F.conv2d(x, w, padding=1))
    y = F.conv2d(x, w, padding=1); g = torch.randn_like(y)
    fb = timeit(lambda: torch.autograd.grad(F.conv2d(x, w, padding=1), (x, w), g))
    res[name] = dict(fwd_ms=round(f, 3), fwd_TF=round(fl / f / 1e9, 1), fwd_bwd_ms=round(fb, 3), bwd_TF=round(2 * fl / (fb - f) / 1e9, 1))
    print(name, res[name], flush=True)
