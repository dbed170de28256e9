import sys, json; sys.path.insert(0, ".")
import torch
from styl3r_amd import vit_ops
dev = "cuda:0"
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
for name, (M, N, K, gelu) in dict(qkv=(5140, 3072, 1024, False), fc1=(5140, 4096, 1024, True), fc2=(5140, 1024, 4096, False), proj=(5140, 1024, 1024, False),
                                  dfc1=(5140, 3072, 768, True), small=(1028, 3072, 1024, False)).items():
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    ms = timeit(lambda: vit_ops.fused_linear(x, w, b, gelu=gelu))
    ref = (lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b))) if gelu else (lambda: torch.nn.functional.linear(x, w, b))
    mt = timeit(ref)
    print(f"{name:6s} M{M} N{N} K{K} gelu={gelu}: mine {ms:.4f} ms {2*M*N*K/ms/1e9:6.1f} TF | torch {mt:.4f} ms {2*M*N*K/mt/1e9:6.1f} TF", flush=True)
