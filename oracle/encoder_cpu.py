"""CPU (torch, fp32) attention for running the encoder's module graph on host tensors -- TEST / BASELINE INFRASTRUCTURE ONLY.

styl3r_amd's encoder has no CPU path for its attention kernels (every other layer falls back to the framework's ops on host tensors);
`cpu_attention()` patches the two names styl3r_amd.vit resolves at call time with a plain-torch restatement, differentiable through
autograd, so that bench.py's `cpu_baseline` can time the full-size encoder on the box's host cores.  Only tests/ and
bench.py's cpu_baseline leg may import this module; the product path never does.

rope2d_torch restates rope_2d_cpu (src/model/encoder/backbone/croco/curope/curope.cpp:11-47: four quarters [u_Y, v_Y, u_X, v_X],
inv_freq = 1 / base^(d / Q)) exactly as oracle/vit_oracle.py::rope2d does in numpy (pinned there against the reference's RoPE2D);
attention = softmax(q k^T scale) v on (B, N, H, D) (xformers' memory_efficient_attention contract, blocks.py:129,195).
"""
from contextlib import contextmanager

import torch


def rope2d_torch(t: torch.Tensor, positions: torch.Tensor, base: float = 100.0) -> torch.Tensor:
    B, N, H, D = t.shape
    Q = D // 4
    inv_freq = 1.0 / torch.pow(torch.tensor(float(base), dtype=t.dtype), torch.arange(Q, dtype=t.dtype) / Q)
    out = []
    for axis, off in ((0, 0), (1, 2 * Q)):
        th = positions[:, :, axis].to(t.dtype)[:, :, None, None] * inv_freq[None, None, None, :]
        c, s = torch.cos(th), torch.sin(th)
        u, v = t[..., off:off + Q], t[..., off + Q:off + 2 * Q]
        out += [u * c - v * s, v * c + u * s]
    return torch.cat(out, -1)


def attention_torch(q, k, v, scale=None, p=0.0, qpos=None, kpos=None, rope_base=100.0, max_pos=64):
    assert p == 0.0
    if scale is None:
        scale = q.shape[-1] ** -0.5
    if qpos is not None:
        q, k = rope2d_torch(q, qpos, rope_base), rope2d_torch(k, kpos, rope_base)
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    return torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v)


@contextmanager
def cpu_attention():
    """inside: styl3r_amd.vit's attention calls on HOST tensors run `attention_torch`"""
    from styl3r_amd import vit
    keep = vit.memory_efficient_attention
    vit.memory_efficient_attention = attention_torch
    try:
        yield
    finally:
        vit.memory_efficient_attention = keep
