"""CPU: the numpy ViT oracle against golden vectors from the reference's own RoPE2D / Attention."""
from pathlib import Path

import numpy as np

from oracle import vit_oracle as vo

G = np.load(Path(__file__).resolve().parent / "golden" / "vit_blocks.npz")


def test_rope_oracle_matches_reference_forward_and_backward():
    tok = G["rope_tokens"]                       # (B,H,N,D)
    pos = G["rope_pos"]
    out = vo.rope2d(tok.transpose(0, 2, 1, 3), pos, base=100.0, fwd=1.0).transpose(0, 2, 1, 3)
    np.testing.assert_allclose(out, G["rope_out"], atol=2e-6)
    # backward = the same rotation with fwd = -1 applied to the incoming gradient (curope2d.py:23-29)
    gin = vo.rope2d(G["rope_gout"].transpose(0, 2, 1, 3), pos, base=100.0, fwd=-1.0).transpose(0, 2, 1, 3)
    np.testing.assert_allclose(gin, G["rope_gin"], atol=2e-6)
    # known answer: forward then inverse is the identity
    back = vo.rope2d(out.transpose(0, 2, 1, 3), pos, fwd=-1.0).transpose(0, 2, 1, 3)
    np.testing.assert_allclose(back, tok, atol=1e-12)


def test_attention_oracle_matches_reference_attention_module():
    """recompute the reference Attention module's output from its state_dict with the oracle pieces"""
    sd = {k[len("attn_sd_"):]: G[k] for k in G.files if k.startswith("attn_sd_")}
    x, xpos = G["attn_in_x"].astype(np.float64), G["attn_in_xpos"]
    B, N, Cdim = x.shape
    H = 2
    qkv = (x @ sd["qkv.weight"].T.astype(np.float64) + sd["qkv.bias"]).reshape(B, N, 3, H, Cdim // H)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]       # (B,N,H,D)
    q = vo.rope2d(q, xpos); k = vo.rope2d(k, xpos)
    o, _ = vo.attention(q, k, v, (Cdim // H) ** -0.5)
    y = o.reshape(B, N, Cdim) @ sd["proj.weight"].T.astype(np.float64) + sd["proj.bias"]
    np.testing.assert_allclose(y, G["attn_out"], atol=2e-5)


def test_attention_backward_oracle_fd():
    rng = np.random.default_rng(0)
    q, k, v = (rng.normal(size=(1, n, 2, 8)) for n in (5, 7, 7))
    g = rng.normal(size=(1, 5, 2, 8))
    dq, dk, dv = vo.attention_backward(q, k, v, 0.3, g)
    f = lambda q_, k_, v_: float((vo.attention(q_, k_, v_, 0.3)[0] * g).sum())
    eps = 1e-6
    for arr, grad, idx in ((q, dq, (0, 2, 1, 3)), (k, dk, (0, 4, 0, 5)), (v, dv, (0, 6, 1, 2))):
        a0 = arr[idx]
        arr[idx] = a0 + eps; fp = f(q, k, v)
        arr[idx] = a0 - eps; fm = f(q, k, v)
        arr[idx] = a0
        np.testing.assert_allclose(grad[idx], (fp - fm) / (2 * eps), rtol=1e-6, atol=1e-9)
