#!/usr/bin/env python
"""Weight-gradient kernel (vit_linear_x6_wgrad) per shape and arithmetic mode: TFLOP/s of fp32-accurate products, warm clocks.
   PRODUCTS={3,6} python tools/probes/wgrad_lab.py"""
import json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
if os.environ.get("OLD_LIB"):
    vit_ops.LIB_PATH = Path(os.environ["OLD_LIB"])
dev = "cuda:0"
vit_ops.LINEAR_MODE = os.environ.get("MODE") or ("bf16x3" if os.environ.get("PRODUCTS", "3") == "3" else "bf16x6")
vit_ops._x6()
lib = vit_ops.load()
M = int(os.environ.get("ROWS", "5140"))
SHAPES = {"enc_qkv": (3072, 1024), "enc_proj": (1024, 1024), "enc_fc1": (4096, 1024), "enc_fc2": (1024, 4096),
          "dec_qkv": (2304, 768), "dec_proj": (768, 768), "dec_fc1": (3072, 768), "dec_fc2": (768, 3072)}
s = torch.cuda.current_stream().cuda_stream
ONLY = os.environ.get("ONLY")
for name, (N, K) in SHAPES.items():
    if ONLY and name != ONLY: continue
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev)
    buf = torch.empty(N * K + N, device=dev)
    if vit_ops.LINEAR_MODE == "f16x3":       # the tensors' |max| words, announced before every launch (consumed by it)
        wa, wb = vit_ops._amax_word(dy), vit_ops._amax_word(x)

        def run():
            vit_ops._announce(wa, wb)
            return lib.vit_linear_x6_wgrad(dy.data_ptr(), x.data_ptr(), buf.data_ptr(), buf[N * K:].data_ptr(), M, N, K, s)
    else:
        run = lambda: lib.vit_linear_x6_wgrad(dy.data_ptr(), x.data_ptr(), buf.data_ptr(), buf[N * K:].data_ptr(), M, N, K, s)
    for _ in range(20): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "mode": vit_ops.LINEAR_MODE, "kernel": "old" if os.environ.get("VIT_WGRAD_OLD") else "x6t", "us": round(us, 1), "TF": round(2 * M * N * K / us / 1e6, 1)}))
