#!/usr/bin/env python
"""Upper bound of what "operands arrive already split"  can buy the f16x3 kernels: the same launches on a library built
with -DVIT_EXP_NOSPLIT (split8s = a bit copy: wrong results, identical loads / LDS traffic / barriers / MFMAs) against the product library.
One JSON line per (kernel, shape); run once per library:  VIT_LIB_NAME=libvit_nosplit.so python tools/probes/nosplit_lab.py"""
import json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
vit_ops.LINEAR_MODE = "f16x3"; vit_ops._x6()
lib = vit_ops.load(); dev = "cuda:0"; tag = os.environ.get("VIT_LIB_NAME", "product")
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


_w = torch.randn(4096, 4096, device=dev)
for _ in range(200): _w @ _w
M = 5140
for name, (N, K) in {"enc_qkv": (3072, 1024), "enc_fc1": (4096, 1024), "enc_fc2": (1024, 4096), "enc_proj": (1024, 1024), "dec_qkv": (2304, 768), "dec_fc1": (3072, 768)}.items():
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
    buf = torch.empty(N * K + N, device=dev)
    wa, wb = vit_ops._amax_word(dy), vit_ops._amax_word(x)

    def wgrad():
        vit_ops._announce(wa, wb)
        lib.vit_linear_x6_wgrad(dy.data_ptr(), x.data_ptr(), buf.data_ptr(), buf[N * K:].data_ptr(), M, N, K, s)
    us = timeit(wgrad)
    print(json.dumps({"lib": tag, "kernel": "wgrad", "shape": name, "us": round(us, 1), "TF3": round(3 * 2 * M * N * K / us / 1e6, 1)}), flush=True)
    wt = torch.nn.Parameter(w); xr = x.clone().requires_grad_(True)
    ring = vit_ops._ring_cfg(M, N, K)
    us = timeit(lambda: vit_ops._FusedLinear.apply(xr, wt, None, None, 0))
    print(json.dumps({"lib": tag, "kernel": f"linear_fwd(ring cfg {ring})", "shape": name, "us": round(us, 1), "TF3": round(3 * 2 * M * N * K / us / 1e6, 1)}), flush=True)
    with torch.no_grad():
        us = timeit(lambda: vit_ops.fused_linear(x, wt))
    print(json.dumps({"lib": tag, "kernel": "linear_fwd(k_linear_x6)", "shape": name, "us": round(us, 1), "TF3": round(3 * 2 * M * N * K / us / 1e6, 1)}), flush=True)
# halo convolution of the DPT heads: 20 images, 256 -> 256 channels at 128^2 and 10 at 256^2
for B, C, HW in ((20, 256, 128), (10, 256, 256)):
    x = torch.randn(B, C, HW, HW, device=dev); w = torch.nn.Parameter(torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5)
    with torch.no_grad():
        us = timeit(lambda: vit_ops.conv_x6_forward(x, w), iters=10, warm=3)
    print(json.dumps({"lib": tag, "kernel": "conv3x3_fwd", "shape": f"{B}x{C}x{HW}^2", "us": round(us, 1), "TF3": round(3 * 2 * B * HW * HW * C * C * 9 / us / 1e6, 1)}), flush=True)
    dy = torch.randn(B, C, HW, HW, device=dev); dw = torch.empty_like(w); db = torch.empty(C, device=dev)
    wa, wb = vit_ops._amax_word(dy), vit_ops._amax_word(x)

    def cw():
        vit_ops._announce(wa, wb)
        lib.vit_conv_x6_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), B, C, C, HW, HW, 3, 0, s)
    us = timeit(cw, iters=10, warm=3)
    print(json.dumps({"lib": tag, "kernel": "conv3x3_wgrad", "shape": f"{B}x{C}x{HW}^2", "us": round(us, 1), "TF3": round(3 * 2 * B * HW * HW * C * C * 9 / us / 1e6, 1)}), flush=True)
