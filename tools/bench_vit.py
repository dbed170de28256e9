#!/usr/bin/env python
"""Micro-benchmarks of the ViT kernels at Styl3R's shapes (C3: 10 scenes x 2 views): achieved TFLOP/s vs the
157 TF fp32-MFMA peak, GB/s for RoPE.  Prints one JSON object.  (tools/, not the headline bench.)"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from styl3r_amd import vit_ops

dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
res = {}
# attention: encoder self-attn (20 views, 16 heads, 257 tokens), decoder cross (12 heads), stylizer self (514)
for name, (B, H, Nq, Nk) in dict(enc_self=(20, 16, 257, 257), dec_cross=(20, 12, 257, 257), sty_self=(10, 12, 514, 514),
                                 c5_self=(4, 16, 1025, 1025)).items():
    q = torch.randn(B, Nq, H, 64, device=dev, requires_grad=True); k = torch.randn(B, Nk, H, 64, device=dev, requires_grad=True)
    v = torch.randn(B, Nk, H, 64, device=dev, requires_grad=True)
    pos = torch.zeros(B, Nq, 2, dtype=torch.int64, device=dev); posk = torch.zeros(B, Nk, 2, dtype=torch.int64, device=dev)
    f = lambda: vit_ops.memory_efficient_attention(q, k, v, 0.125, qpos=pos, kpos=posk, max_pos=64)
    ms = timeit(lambda: f())
    fl = 4 * B * H * Nq * Nk * 64
    out = f(); g = torch.randn_like(out)
    msb = timeit(lambda: torch.autograd.grad(f(), (q, k, v), g)) - ms
    sd = lambda: torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    ms_t = timeit(lambda: sd())
    res["attn_" + name] = dict(fwd_ms=round(ms, 4), fwd_TF=round(fl / ms / 1e9, 1), bwd_ms=round(msb, 4),
                               bwd_TF=round(2.5 * fl / msb / 1e9, 1), torch_sdpa_fwd_ms=round(ms_t, 4))
for name, (M, N, K, gelu) in dict(enc_qkv=(5140, 3072, 1024, False), enc_fc1=(5140, 4096, 1024, True), enc_fc2=(5140, 1024, 4096, False),
                                  dec_fc1=(5140, 3072, 768, True), enc_proj=(5140, 1024, 1024, False),
                                  infer_qkv=(514, 3072, 1024, False), infer_fc2=(514, 1024, 4096, False)).items():
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    vit_ops.LINEAR_MODE = "f32"
    ms = timeit(lambda: vit_ops.fused_linear(x, w, b, gelu=gelu))
    vit_ops.LINEAR_MODE = "bf16x6"
    ms6 = timeit(lambda: vit_ops.fused_linear(x, w, b, gelu=gelu))
    vit_ops.LINEAR_MODE = "f32"
    ref = (lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b))) if gelu else (lambda: torch.nn.functional.linear(x, w, b))
    ms_t = timeit(ref)
    # weight (+ bias) gradient: bf16x6 split-M kernel vs the library GEMM dY^T X plus the column-sum kernel
    gy = torch.randn(M, N, device=dev); dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    import ctypes as C
    st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms_w6 = timeit(lambda: vit_ops.load().vit_linear_x6_wgrad(gy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, st_))
    ms_wt = timeit(lambda: (gy.t() @ x, gy.sum(0)))
    res["wgrad_" + name] = dict(bf16x6_ms=round(ms_w6, 4), bf16x6_TF=round(2 * M * N * K / ms_w6 / 1e9, 1), torch_ms=round(ms_wt, 4),
                                torch_TF=round(2 * M * N * K / ms_wt / 1e9, 1))
    res["linear_" + name] = dict(ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1), bf16x6_ms=round(ms6, 4),
                                 bf16x6_TF=round(2 * M * N * K / ms6 / 1e9, 1), torch_ms=round(ms_t, 4),
                                 torch_TF=round(2 * M * N * K / ms_t / 1e9, 1))
t = torch.randn(20, 16, 257, 64, device=dev); pos = torch.zeros(20, 257, 2, dtype=torch.int64, device=dev)
rope = vit_ops.RoPE2D(100.0, max_pos=16)
ms = timeit(lambda: rope(t, pos))
res["rope2d"] = dict(ms=round(ms, 4), GBps=round(2 * t.numel() * 4 / ms / 1e6, 1))
print(json.dumps(res))
