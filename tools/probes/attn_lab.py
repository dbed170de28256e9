#!/usr/bin/env python
"""Attention forward: exact-f32 MFMA kernel vs the bf16x6 kernel -- error vs float64 and time, with and without fused RoPE."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
dev = torch.device("cuda:0")


def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


torch.manual_seed(0)
for name, (B, H, Nq, Nk, rope) in dict(enc_self=(20, 16, 257, 257, True), dec_cross=(20, 12, 257, 257, False), sty_self=(10, 12, 514, 514, True),
                                       c5_self=(4, 16, 1025, 1025, True), small=(2, 3, 70, 45, True)).items():
    qkv = torch.randn(B, max(Nq, Nk), 3, H, 64, device=dev)
    q, k, v = qkv[:, :Nq, 0], qkv[:, :Nk, 1], qkv[:, :Nk, 2]
    pos = torch.stack(torch.meshgrid(torch.arange(40, device=dev), torch.arange(40, device=dev), indexing="ij"), -1).reshape(-1, 2)
    qpos = pos[:Nq].expand(B, Nq, 2).contiguous() if rope else None
    kpos = pos[:Nk].expand(B, Nk, 2).contiguous() if rope else None
    kw = dict(qpos=qpos, kpos=kpos, max_pos=64) if rope else {}
    if rope:
        def rot(x, p):               # float64 2-D RoPE: (d, d+16) of [0,32) by y, of [32,64) by x; table = the fp32 table the kernels read
            cos, sin = vit_ops.rope_tables(64, 65, 100.0, x.device)
            x = x.double(); o = x.clone()
            for base, ax in ((0, 0), (32, 1)):
                c = cos[p[..., ax]].double()[:, :, None, :16]; s_ = sin[p[..., ax]].double()[:, :, None, :16]
                u, w = x[..., base:base + 16], x[..., base + 16:base + 32]
                o[..., base:base + 16] = u * c - w * s_; o[..., base + 16:base + 32] = w * c + u * s_
            return o
        qd, kd = rot(q, qpos), rot(k, kpos)
    else:
        qd, kd = q.double(), k.double()
    s = torch.einsum("bqhd,bkhd->bhqk", qd, kd) * 0.125
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v.double())
    for arith in ("f32", "bf16x6"):
        vit_ops.ATTENTION_ARITH = arith
        with torch.no_grad():
            out = vit_ops.memory_efficient_attention(q, k, v, 0.125, **kw)
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            ms = timeit(lambda: vit_ops.memory_efficient_attention(q, k, v, 0.125, **kw))
        # backward: gradients of sum(out * w) against float64 autograd of the same expression
        if arith == "f32": gw = torch.randn_like(out)          # one cotangent per shape: the float64 gradients below are reused by the second kernel
        q_, k_, v_ = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        o = vit_ops.memory_efficient_attention(q_, k_, v_, 0.125, **kw)
        (o * gw).sum().backward()
        if arith == "f32":
            qd_, kd_, vd_ = (t.detach().double().requires_grad_(True) for t in (q, k, v))
            qr, kr = (rot(qd_, qpos), rot(kd_, kpos)) if rope else (qd_, kd_)
            s_ = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * 0.125
            (torch.einsum("bhqk,bkhd->bqhd", s_.softmax(-1), vd_) * gw.double()).sum().backward()
            gref = (qd_.grad, kd_.grad, vd_.grad)
        gerr = [float((a_.grad.double() - r_).abs().max() / r_.abs().max()) for a_, r_ in zip((q_, k_, v_), gref)]
        def fb():
            o = vit_ops.memory_efficient_attention(q_, k_, v_, 0.125, **kw)
            torch.autograd.grad(o, (q_, k_, v_), gw)
        msb = timeit(fb, iters=50) - ms
        print(json.dumps(dict(shape=name, arith=arith, err=err, ms=round(ms, 4), TF=round(4 * B * H * Nq * Nk * 64 / ms / 1e9, 1), bwd_ms=round(msb, 4),
                              grad_err=["%.1e" % e for e in gerr])), flush=True)
