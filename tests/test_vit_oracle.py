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


def test_torch_cpu_attention_of_the_baseline_leg_matches_the_numpy_oracle():
    """oracle/encoder_cpu.py (the plain-torch attention bench.py's CPU baseline runs the encoder with) against oracle/vit_oracle.py,
    which is pinned to the reference's RoPE2D / attention goldens above: RoPE, forward, and the gradients through autograd"""
    import numpy as np
    import torch
    from oracle import vit_oracle
    from oracle.encoder_cpu import attention_torch, rope2d_torch
    g = torch.Generator().manual_seed(3)
    B, N, H, D = 2, 37, 3, 64
    q, k, v = (torch.randn(B, N, H, D, generator=g, dtype=torch.float64, requires_grad=True) for _ in range(3))
    pos = torch.stack([torch.randint(0, 16, (B, N), generator=g), torch.randint(0, 16, (B, N), generator=g)], -1)
    want_q = vit_oracle.rope2d(q.detach().numpy(), pos.numpy())
    assert np.abs(rope2d_torch(q.detach(), pos).numpy() - want_q).max() < 1e-12
    out = attention_torch(q, k, v, 0.125, qpos=pos, kpos=pos)
    qr, kr = vit_oracle.rope2d(q.detach().numpy(), pos.numpy()), vit_oracle.rope2d(k.detach().numpy(), pos.numpy())
    want, _ = vit_oracle.attention(qr, kr, v.detach().numpy(), 0.125)
    assert np.abs(out.detach().numpy() - want).max() < 1e-12
    gy = torch.randn(B, N, H, D, generator=g, dtype=torch.float64)
    (out * gy).sum().backward()
    dq, dk, dv = vit_oracle.attention_backward(qr, kr, v.detach().numpy(), 0.125, gy.numpy())
    assert np.abs(v.grad.numpy() - dv).max() < 1e-12
    # dq / dk of the oracle are gradients w.r.t. the ROTATED tensors: rotate back (the rotation is orthogonal: inverse = fwd -1)
    assert np.abs(q.grad.numpy() - vit_oracle.rope2d(dq, pos.numpy(), fwd=-1.0)).max() < 1e-12
    assert np.abs(k.grad.numpy() - vit_oracle.rope2d(dk, pos.numpy(), fwd=-1.0)).max() < 1e-12
