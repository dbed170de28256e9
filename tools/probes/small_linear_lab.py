#!/usr/bin/env python
"""Small-M Linear (csrc/vit_gemm_sm.hip) against the 128-row-tile kernel it replaces at batch-1 row counts: time per launch on the C2 shapes,
every tile / wave configuration, the three arithmetic modes.  Weights rotate through enough copies (> 600 MB) that every launch reads its
weight image from HBM, as inside the model (2+ GB of weight images per inference, 256 MB of last-level cache).  One JSON line per (mode, shape).
usage: python tools/probes/small_linear_lab.py [modes=f16x3,bf16x6] [act=0]"""
import ctypes as C, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops as vo

dev = torch.device("cuda:0")
lib = vo.load()
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16x3", "bf16x6"]
ACT = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shapes = dict(enc_qkv=(514, 3072, 1024), enc_proj=(514, 1024, 1024), enc_fc1=(514, 4096, 1024), enc_fc2=(514, 1024, 4096),
              sty_qkv=(257, 3072, 1024), sty_fc1=(257, 4096, 1024), sty_fc2=(257, 1024, 4096), sty_proj=(257, 1024, 1024),
              dec_qkv=(257, 2304, 768), dec_proj=(257, 768, 768), dec_fc1=(257, 3072, 768), dec_fc2=(257, 768, 3072), dec_kv514=(514, 768, 768),
              enc2dec=(514, 768, 1024))
import os
if os.environ.get("BIG"):      # the train-step row counts: is the barrier-free kernel worth taking above 1 024 rows?  (ring kernels cfg 1 / 3 timed beside it)
    Mb = int(os.environ["BIG"])
    shapes = dict(enc_qkv=(Mb, 3072, 1024), enc_proj=(Mb, 1024, 1024), enc_fc1=(Mb, 4096, 1024), enc_fc2=(Mb, 1024, 4096),
                  dec_qkv=(Mb // 2, 2304, 768), dec_proj=(Mb // 2, 768, 768), dec_fc1=(Mb // 2, 3072, 768), dec_fc2=(Mb // 2, 768, 3072))
MAXR = 1 << 20 if os.environ.get("BIG") else 1024
if os.environ.get("SHAPES"):
    shapes = {k: v for k, v in shapes.items() if k in os.environ["SHAPES"].split(",")}
CONFIGS = ((0, 0),) if os.environ.get("RULE_ONLY") else ((1, 4), (1, 8), (2, 4), (2, 8), (0, 0))
_w = torch.randn(4096, 4096, device=dev)
for _ in range(200): _w @ _w
torch.cuda.synchronize()


def timeit(fns, iters=240, warm=24):
    """GPU time per launch: the launches are captured into one hipGraph and replayed (no host time between them)"""
    n = len(fns)
    for i in range(warm): fns[i % n]()
    torch.cuda.synchronize()
    if os.environ.get("LAB_EAGER"):                      # (counter collection: plain launches, few of them)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for i in range(24): fns[i % n]()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 24 * 1e3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): fns[i % n]()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e3


for mode in modes:
    vo.LINEAR_MODE = mode
    assert vo._x6()
    for name, (M, N, K) in shapes.items():
        torch.manual_seed(1)
        x = torch.randn(M, K, device=dev); b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev)
        copies = max(2, int(600e6 / (N * K * 6)) + 1)
        ws = [torch.randn(N, K, device=dev) / K ** 0.5 for _ in range(copies)]
        wps = [vo.split_weight(w) for w in ws]; wbs = [vo.split_weight_block(w) for w in ws]
        out = torch.empty(M, N, device=dev)
        ax = vo._amax_of(x) if mode == "f16x3" else None
        ref = x.double() @ ws[0].double().t() + b.double()
        if ACT: ref = torch.nn.functional.gelu(ref)
        ref = ref + res.double()

        def mk(wp, small):
            def f():
                if ax is not None: vo._announce(ax)
                if small: rc = lib.vit_linear_x6r_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr(), res.data_ptr(), out.data_ptr(), None, M, N, K, ACT, 5, vo._stream(dev))
                else: rc = lib.vit_linear_x6_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr(), res.data_ptr(), out.data_ptr(), None, M, N, K, ACT, vo._stream(dev))
                assert rc == 0, rc
            return f
        fns = [mk(wp, False) for wp in wps]; fsm = [mk(wp, True) for wp in wbs]
        row = dict(mode=mode, shape=name, M=M, N=N, K=K, act=ACT, weight_copies=copies)
        fns[0](); row["err_old"] = float((out.double() - ref).abs().max() / ref.abs().max())
        row["old_us"] = round(timeit(fns), 2)
        for tm, nw in CONFIGS:
            assert lib.vit_linear_sm_set(MAXR, tm, nw) == 0
            if not lib.vit_linear_sm_ok(M, N, K): continue
            fsm[0](); e = float((out.double() - ref).abs().max() / ref.abs().max())
            row[f"sm_{tm}x{nw}_us" if tm else "sm_rule_us"] = round(timeit(fsm), 2)
            row[f"err_{tm}x{nw}"] = e
        if os.environ.get("BIG"):
            for cfg in (1, 3):
                def mkr(wp, cfg=cfg):
                    def f():
                        if ax is not None: vo._announce(ax)
                        rc = lib.vit_linear_x6r_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr(), res.data_ptr(), out.data_ptr(), None, M, N, K, ACT, cfg, vo._stream(dev))
                        assert rc == 0, rc
                    return f
                row[f"ring{cfg}_us"] = round(timeit([mkr(wp) for wp in wbs], iters=60, warm=6), 2)
        prod = {"bf16x6": 6, "bf16x3": 3, "f16x3": 3}[mode]
        best = min(v for k, v in row.items() if k.startswith("sm_") and k.endswith("_us"))
        row["best_TF_mfma"] = round(2 * M * N * K * prod / best / 1e6, 1)
        row["weight_GBps_best"] = round(N * K * (6 if prod == 6 else 4) / best / 1e3, 1)
        print(json.dumps(row), flush=True)
        del ws, wps, wbs
assert lib.vit_linear_sm_set(1024, 0, 0) == 0
