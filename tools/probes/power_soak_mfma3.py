#!/usr/bin/env python
"""Are the three MFMA kernels with the barrier-per-slab structure (halo convolution, Linear weight gradient) limited by the power budget like
the ring Linear kernel (profiles/r03_power_trace.md)?  Same instruction stream on RANDOM and on ZERO operands (no switching activity in the
multipliers), bf16x3 (three products), TF per phase; run beside an SMI sampler (tools/probes/power_trace_mfma3.sh).
    python tools/probes/power_soak_mfma3.py [seconds-per-phase]"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
from styl3r_amd.vit_ops import Conv2dX6
dev = torch.device("cuda:0")
lib = vit_ops.load()
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
vit_ops.LINEAR_MODE = "bf16x3"; vit_ops._x6()
st = torch.cuda.current_stream().cuda_stream


def soak(name, run, flops):
    t_end = time.time() + SECS
    best = []
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best.append(flops * 20 / e0.elapsed_time(e1) / 1e9)
    print(json.dumps({"t": round(time.time(), 2), "kernel": name, "TF_first": round(best[0], 1), "TF_last": round(best[-1], 1), "TF_median": round(sorted(best)[len(best) // 2], 1)}), flush=True)


for operands in ("random", "zero", "random"):
    B, Ci, Co, H = 20, 256, 256, 256
    conv = Conv2dX6(Ci, Co, 3, 1, 1).to(dev)
    x = torch.randn(B, Ci, H, H, device=dev)
    if operands == "zero":
        x.zero_(); conv.weight.data.zero_(); conv.bias.data.zero_()
    with torch.no_grad():
        soak(f"k_conv3h_x6 gs-head0 256>256 @256 [{operands}]", lambda: conv.forward_fused(x), 2.0 * B * H * H * Co * Ci * 9)
    del x, conv
    M, N, K = 5140, 4096, 1024
    dy = torch.randn(M, N, device=dev); xx = torch.randn(M, K, device=dev)
    if operands == "zero":
        dy.zero_(); xx.zero_()
    dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    soak(f"k_wgrad_x6 fc1 5140x4096x1024 [{operands}]", lambda: lib.vit_linear_x6_wgrad(dy.data_ptr(), xx.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, 0, st), 2.0 * M * N * K)
    del dy, xx, dw, db
