"""CPU oracle (numpy) for the ViT kernels -- TEST INFRASTRUCTURE ONLY.

rope2d   restates rope_2d_cpu, src/model/encoder/backbone/croco/curope/curope.cpp:11-47
         (four quarters [u_Y, v_Y, u_X, v_X], inv_freq = fwd / base^(d/Q)); pinned by golden
         vectors produced from the reference's own RoPE2D (tests/golden/make_vit_fixtures.py).
attention restates memory_efficient_attention's contract softmax(q k^T scale) v on (B,N,H,D)
         (xformers 0.0.24, call sites blocks.py:129,195), evaluated in float64.
"""
import numpy as np


def rope2d(tokens_bnhd, positions, base=100.0, fwd=1.0, dtype=np.float64):
    t = np.array(tokens_bnhd, dtype=dtype, copy=True)
    B, N, H, D = t.shape
    Q = D // 4
    d = np.arange(Q, dtype=dtype)
    inv_freq = dtype(fwd) / np.power(dtype(base), d / dtype(Q))
    for axis, off in ((0, 0), (1, 2 * Q)):
        th = positions[:, :, axis].astype(dtype)[:, :, None, None] * inv_freq[None, None, None, :]
        c, s = np.cos(th), np.sin(th)
        u = t[..., off:off + Q].copy(); v = t[..., off + Q:off + 2 * Q].copy()
        t[..., off:off + Q] = u * c - v * s
        t[..., off + Q:off + 2 * Q] = v * c + u * s
    return t


def attention(q, k, v, scale):
    q, k, v = (np.asarray(x, dtype=np.float64) for x in (q, k, v))
    s = np.einsum("bqhd,bkhd->bhqk", q, k) * scale
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    lse_shift = np.log(p.sum(-1))
    p = p / p.sum(-1, keepdims=True)
    return np.einsum("bhqk,bkhd->bqhd", p, v), p


def attention_backward(q, k, v, scale, g):
    q, k, v, g = (np.asarray(x, dtype=np.float64) for x in (q, k, v, g))
    o, p = attention(q, k, v, scale)
    dv = np.einsum("bhqk,bqhd->bkhd", p, g)
    dp = np.einsum("bqhd,bkhd->bhqk", g, v)
    delta = (g * o).sum(-1).transpose(0, 2, 1)[..., None]      # (b,h,q,1)
    ds = p * (dp - delta) * scale
    dq = np.einsum("bhqk,bkhd->bqhd", ds, k)
    dk = np.einsum("bhqk,bqhd->bkhd", ds, q)
    return dq, dk, dv
