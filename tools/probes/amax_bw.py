"""vit_amax pass: achieved read bandwidth per tensor size (f16x3 mode's per-tensor |max|).  python tools/probes/amax_bw.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from styl3r_amd import vit_ops
dev = "cuda:0"
lib = vit_ops.load()
for mb in (1, 5, 21, 84, 335, 671, 1342):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev)
    w = torch.zeros(64 * 32, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.vit_amax(x.data_ptr(), n, w.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        lib.vit_amax(x.data_ptr(), n, w.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ok = int(w.max().item()) == int(x.abs().max().view(torch.int32).item())
    print(f"{mb:5d} MB  {us:8.1f} us  {mb * 1.048576 / us * 1e3:7.1f} GB/s  exact={ok}")
