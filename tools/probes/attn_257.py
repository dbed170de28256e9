import sys; sys.path.insert(0, ".")
import torch
from styl3r_amd import vit_ops
dev = "cuda:0"
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
for N in (256, 257, 288, 320):
    B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 16
    q = torch.randn(B, N, H, 64, device=dev, requires_grad=True); k = torch.randn(B, N, H, 64, device=dev, requires_grad=True); v = torch.randn(B, N, H, 64, device=dev, requires_grad=True)
    pos = torch.zeros(B, N, 2, dtype=torch.int64, device=dev)
    f = lambda: vit_ops.memory_efficient_attention(q, k, v, 0.125, qpos=pos, kpos=pos, max_pos=64)
    tf = timeit(f); g = torch.randn_like(f())
    tb = timeit(lambda: torch.autograd.grad(f(), (q, k, v), g)) - tf
    print(N, "fwd %.4f ms  bwd %.4f ms" % (tf, tb), flush=True)
