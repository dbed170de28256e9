#!/usr/bin/env python
"""bf16x6 Linear: the LDS-DMA ring kernel (vit_linear_x6r_fwd, every tile configuration) against the register-staged kernel
(vit_linear_x6_fwd) -- error vs float64 and time per shape.  Prints one JSON line per (shape, kernel)."""
import ctypes as C, json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops

dev = torch.device("cuda:0")
lib = vit_ops.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3]
shapes = dict(enc_qkv=(5140, 3072, 1024), enc_fc1=(5140, 4096, 1024), enc_fc2=(5140, 1024, 4096), enc_proj=(5140, 1024, 1024),
              dec_qkv=(5120, 2304, 768), dec_fc1=(5120, 3072, 768), dec_fc2=(5120, 768, 3072), dec_proj=(5120, 768, 768), odd=(1000, 200, 48))
if len(sys.argv) > 2:
    shapes = {k: (shapes[k] if k in shapes else tuple(int(v) for v in k.split("x"))) for k in sys.argv[2].split(",")}
assert lib.vit_x6_set_products(int(os.environ.get("PRODUCTS", "6"))) == 0      # 6 (default) or 3: partial products per launch
ACT = int(os.environ.get("ACT", "0")); USE_RES = int(os.environ.get("RES", "1")); USE_PRE = int(os.environ.get("PRE", "0"))
torch.manual_seed(0)
_w = torch.randn(4096, 4096, device=dev)
for _ in range(200): _w @ _w          # clocks up before the first timing
torch.cuda.synchronize()
for name, (M, N, K) in shapes.items():
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    if int(os.environ.get("ZERO", "0")): x.zero_(); w.zero_()      # DVFS check: same instruction stream, no switching activity in the matrix pipes
    res = torch.randn(M, N, device=dev)
    ref = (x.double() @ w.double().t() + b.double())
    if ACT: ref = torch.nn.functional.gelu(ref)
    if USE_RES: ref = ref + res.double()
    pre = torch.empty(M, N, device=dev) if USE_PRE else None
    rp = res.data_ptr() if USE_RES else None; pp = pre.data_ptr() if USE_PRE else None
    scale = float(ref.abs().max())
    old = torch.empty(lib.vit_split_weight_bytes(N, K), dtype=torch.uint8, device=dev)
    assert lib.vit_split_weight(w.data_ptr(), old.data_ptr(), N, K, 0, st) == 0
    blk = torch.empty(lib.vit_split_weight_block_bytes(N, K, 0), dtype=torch.uint8, device=dev)
    assert lib.vit_split_weight_block(w.data_ptr(), blk.data_ptr(), N, K, 0, st) == 0
    out = torch.empty(M, N, device=dev)
    f_old = lambda: lib.vit_linear_x6_fwd(x.data_ptr(), old.data_ptr(), b.data_ptr(), rp, out.data_ptr(), pp, M, N, K, ACT, st)
    assert f_old() == 0
    err = float((out.double() - ref).abs().max()) / scale
    ms = timeit(f_old)
    print(json.dumps(dict(shape=name, kernel="x6", err=err, ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1))), flush=True)
    S = lib.vit_linear_x6c_choose_splits(M, N, K)
    if S > 0:
        wsb = lib.vit_linear_x6c_workspace_bytes(M, N, S)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        out.zero_()
        fa = lambda: lib.vit_linear_x6c_fwd(x.data_ptr(), blk.data_ptr(), b.data_ptr(), rp, out.data_ptr(), pp, M, N, K, ACT, S, ws.data_ptr(), wsb, st)
        assert fa() == 0
        torch.cuda.synchronize()
        errs = []
        for _ in range(5):
            fa(); errs.append(float((out.double() - ref).abs().max()) / scale)
        ms = timeit(fa)
        print(json.dumps(dict(shape=name, kernel=f"x6c auto (S={S})", err=max(errs), ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1))), flush=True)
    else:
        print(json.dumps(dict(shape=name, kernel="x6c auto (S=0: default kernel kept)")), flush=True)
    for cfg in cfgs:
        out.zero_()
        f = lambda: lib.vit_linear_x6r_fwd(x.data_ptr(), blk.data_ptr(), b.data_ptr(), rp, out.data_ptr(), pp, M, N, K, ACT, cfg, st)
        rc = f()
        torch.cuda.synchronize()
        if rc != 0:
            print(json.dumps(dict(shape=name, kernel=f"x6r cfg {cfg}", rc=rc, err=vit_ops.load().vit_last_error().decode()))); continue
        err = float((out.double() - ref).abs().max()) / scale
        ms = timeit(f)
        if cfg == 4 and pre is not None:
            print("phase cycles per slab (wave: barrier-wait, dma issue, lds phase, mfma phase):", [[round(v) for v in row] for row in pre.flatten()[:32].view(8, 4).tolist()])
        print(json.dumps(dict(shape=name, kernel=f"x6r cfg {cfg}", err=err, ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1))), flush=True)
