#!/usr/bin/env python
"""Is the six-product Linear kernel limited by the power budget?  Same instruction stream on RANDOM and on ZERO operands (no switching
activity in the multipliers), each held for several seconds, TF per 0.5 s window -- run beside tools/probes/power_trace.sh, which samples
the SMI clock / power counters.   python tools/probes/power_soak.py [seconds-per-phase]   (prints JSON lines with wall-clock stamps)"""
import ctypes as C, json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
dev = torch.device("cuda:0")
lib = vit_ops.load()
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
M, N, K = 5140, 3072, 1024
st = torch.cuda.current_stream().cuda_stream
for products in (6, 3):
    assert lib.vit_x6_set_products(products) == 0
    for operands in ("random", "zero", "random"):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        if operands == "zero":
            x.zero_(); w.zero_()
        vit_ops.LINEAR_MODE = "bf16x6" if products == 6 else "bf16x3"
        blk = vit_ops.split_weight_block(w); out = torch.empty(M, N, device=dev)
        run = lambda: lib.vit_linear_x6r_fwd(x.data_ptr(), blk.data_ptr(), None, None, out.data_ptr(), None, M, N, K, 0, 3, st)
        t_end = time.time() + SECS
        while time.time() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.time(); n = 0
            e0.record()
            while time.time() - t0 < 0.5:
                for _ in range(50): run()
                n += 50
                torch.cuda.synchronize()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(json.dumps({"t": round(time.time(), 2), "products": products, "operands": operands, "launches": n,
                              "TF_fp32_products": round(2.0 * M * N * K * n / ms / 1e9, 1)}), flush=True)
