"""CroCo / MASt3R ViT building blocks on the gfx950 kernels, with the reference's parameter names.

Mirrors src/model/encoder/backbone/croco/blocks.py: `Mlp` (:61-82), `Attention` (:84-134),
`Block` (:136-152), `CrossAttention` (:154-200), `DecoderBlock` (:202-222).  state_dict keys are
identical (qkv / proj / projq / projk / projv / fc1 / fc2 / norm1..3 / norm_y), so reference
checkpoints load unchanged.  Differences that do not change results: q, k, v are strided views of
the fused qkv buffer (no rearrange copies), the 2-D RoPE is applied inside the attention kernel
(the reference rotates the qkv buffer in place with two extra read/write sweeps), dropout layers with
p = 0 are omitted (every Styl3R config uses drop = attn_drop = drop_path = 0).
Linear (+ bias / exact GELU / residual epilogues) and LayerNorm run on the hand-written kernels of styl3r_amd.vit_ops for
device tensors (fp32-accurate bf16x6 by default), on the framework's fp32 ops for CPU tensors.
"""
from __future__ import annotations

from functools import partial
from typing import Optional

import torch
from torch import Tensor, nn

from .vit_ops import CALLS, GeluLink, LayerNorm, attention_qkv, fused_linear, memory_efficient_attention

def _linear(layer: nn.Linear, x: Tensor, residual: Optional[Tensor] = None, gelu: bool = False, link=None, link_in=None,
            amax_out: bool = False, amax_dx: bool = False) -> Tensor:
    """Linear layers of the blocks on the fused kernels (bias / exact GELU / residual in the epilogue).  Device tensors ALWAYS take them:
    a contraction length the kernels cannot take (not a multiple of 16: no layer of the model) is an error, not a silent library GEMM.
    CPU tensors (host-side tests of the module logic) take the framework's ops and are counted in vit_ops.CALLS["framework_linear"]."""
    if x.is_cuda:
        if layer.in_features % 16 != 0:
            raise RuntimeError(f"fused Linear: in_features = {layer.in_features} is not a multiple of 16 (zero-pad the contraction as encoder._intrinsics_token does)")
        return fused_linear(x, layer.weight, layer.bias, residual=residual, gelu=gelu, link=link, link_in=link_in, amax_out=amax_out, amax_dx=amax_dx)
    CALLS["framework_linear"] += 1
    y = torch.nn.functional.linear(x, layer.weight, layer.bias)
    if gelu:
        y = torch.nn.functional.gelu(y)
    return y if residual is None else residual + y


class RopeCfg:
    """what the reference passes around as a `rope` module: RoPE2D(freq=100) (backbone_croco_multiview.py:29)."""

    def __init__(self, freq: float = 100.0, max_pos: int = 64):
        self.freq, self.max_pos = freq, max_pos


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.0):
        super().__init__()
        assert drop == 0.0
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        if isinstance(self.act, nn.GELU) and self.act.approximate == "none":
            link = GeluLink() if torch.is_grad_enabled() else None     # GELU' runs inside fc2's input-gradient GEMM (vit_ops.GeluLink)
            return _linear(self.fc2, _linear(self.fc1, x, gelu=True, link=link), residual=residual, link_in=link)
        y = self.fc2(self.act(self.fc1(x)))
        return y if residual is None else residual + y


class Attention(nn.Module):
    def __init__(self, dim, rope: Optional[RopeCfg] = None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert attn_drop == 0.0 and proj_drop == 0.0 and dim // num_heads == 64
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def forward(self, x: Tensor, xpos: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        B, N, C = x.shape
        qkv = _linear(self.qkv, x, amax_out=True).view(B, N, 3, self.num_heads, C // self.num_heads)     # (f16x3: the attention's operand scales)
        if qkv.is_cuda and qkv.dtype == torch.float32:
            # packed path: the kernels read the three planes in place and the backward writes ONE (B,N,3,H,64) gradient
            o = attention_qkv(qkv, self.scale, xpos if self.rope is not None else None,
                              self.rope.freq if self.rope is not None else 100.0, self.rope.max_pos if self.rope is not None else 64)
            return _linear(self.proj, o.reshape(B, N, C), residual=residual, amax_dx=True)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]                 # (B,N,H,64) views, no copies
        if self.rope is not None:
            o = memory_efficient_attention(q, k, v, scale=self.scale, qpos=xpos, kpos=xpos, rope_base=self.rope.freq,
                                           max_pos=self.rope.max_pos)
        else:
            o = memory_efficient_attention(q, k, v, scale=self.scale)
        return _linear(self.proj, o.reshape(B, N, C), residual=residual)


def _norm_skip(norm: nn.Module, x: Tensor):
    """(norm(x), x): for vit_ops.LayerNorm the pair comes from ONE autograd node, so the gradient of the residual branch and
    the LayerNorm input gradient are summed inside vit_layernorm_bwd instead of by a separate add over the (B, N, C) tensor."""
    if isinstance(norm, LayerNorm):
        return norm.forward_skip(x)
    return norm(x), x


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, rope=None):
        super().__init__()
        assert drop_path == 0.0
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, rope=rope, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x: Tensor, xpos: Tensor) -> Tensor:
        h, xs = _norm_skip(self.norm1, x)
        x = self.attn(h, xpos, residual=xs)                   # x + attn(...): residual add in the proj epilogue
        h, xs = _norm_skip(self.norm2, x)
        x = self.mlp(h, residual=xs)
        return x


class CrossAttention(nn.Module):
    def __init__(self, dim, rope: Optional[RopeCfg] = None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert attn_drop == 0.0 and proj_drop == 0.0 and dim // num_heads == 64
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.projq = nn.Linear(dim, dim, bias=qkv_bias)
        self.projk = nn.Linear(dim, dim, bias=qkv_bias)
        self.projv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def forward(self, query: Tensor, key: Tensor, value: Tensor, qpos: Tensor, kpos: Tensor,
                residual: Optional[Tensor] = None) -> Tensor:
        B, Nq, C = query.shape
        H = self.num_heads
        q = _linear(self.projq, query, amax_out=True).view(B, Nq, H, C // H)
        k = _linear(self.projk, key, amax_out=True).view(B, key.shape[1], H, C // H)
        v = _linear(self.projv, value, amax_out=True).view(B, value.shape[1], H, C // H)
        if self.rope is not None:
            o = memory_efficient_attention(q, k, v, scale=self.scale, qpos=qpos, kpos=kpos, rope_base=self.rope.freq,
                                           max_pos=self.rope.max_pos)
        else:
            o = memory_efficient_attention(q, k, v, scale=self.scale)
        return _linear(self.proj, o.reshape(B, Nq, C), residual=residual, amax_dx=True)


class DecoderBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, norm_mem=True, rope=None):
        super().__init__()
        assert drop_path == 0.0
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, rope=rope, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.cross_attn = CrossAttention(dim, rope=rope, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop,
                                         proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.norm3 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.norm_y = norm_layer(dim) if norm_mem else nn.Identity()

    def forward(self, x: Tensor, y: Tensor, xpos: Tensor, ypos: Tensor):
        h, xs = _norm_skip(self.norm1, x)
        x = self.attn(h, xpos, residual=xs)
        y_ = self.norm_y(y)
        h, xs = _norm_skip(self.norm2, x)
        x = self.cross_attn(h, y_, y_, xpos, ypos, residual=xs)
        h, xs = _norm_skip(self.norm3, x)
        x = self.mlp(h, residual=xs)
        return x, y


def decoder_blocks_pair_ok(blk1: "DecoderBlock", blk2: "DecoderBlock", x: Tensor) -> bool:
    """serving path: can layer i of BOTH decoders run as one sequence of two-problem launches (vit_ops.grouped_linear / grouped_layernorm)?"""
    from .vit_ops import grouped_ok, grouped_shape_ok
    if not (x.dim() == 4 and x.shape[0] == 2 and x.is_cuda and x.is_contiguous() and x.dtype == torch.float32):
        return False
    C, M, hid = x.shape[-1], x.shape[1] * x.shape[2], blk1.mlp.fc1.out_features
    norms = [blk1.norm1, blk1.norm2, blk1.norm3, blk2.norm1, blk2.norm2, blk2.norm3] + [n for n in (blk1.norm_y, blk2.norm_y) if not isinstance(n, nn.Identity)]
    mlp_ok = all(isinstance(b.mlp.act, nn.GELU) and b.mlp.act.approximate == "none" for b in (blk1, blk2)) and blk2.mlp.fc1.out_features == hid
    return (mlp_ok and isinstance(blk1.norm_y, nn.Identity) == isinstance(blk2.norm_y, nn.Identity)
            and all(isinstance(n, LayerNorm) and n._hip_ok(x) for n in norms) and (blk1.attn.rope is None) == (blk2.attn.rope is None)
            and blk1.attn.num_heads == blk2.attn.num_heads
            and all(grouped_ok(x, n) for n in (C, 3 * C, hid)) and grouped_shape_ok(M, C, hid))


@torch.no_grad()
def decoder_blocks_pair(blk1: "DecoderBlock", blk2: "DecoderBlock", x: Tensor, pos: Tensor, mpos: Tensor) -> Tensor:
    """Layer i of decoder 1 (group 0) and decoder 2 (group 1) in ONE sequence of 14 two-problem launches (serving, two context views: the two
    decoders run the same shapes with two weight sets -- backbone_croco_multiview.py:147-188, blocks.py:203-242).  x: (2, b, l, c) = the two
    decoders' features stacked; the memory of a decoder's cross-attention is the OTHER decoder's input features (x flipped).  pos: (2 b, l, 2),
    mpos: (2 b, l, 2) positions of the queries / of the memory.  The arithmetic of DecoderBlock.forward, operation for operation."""
    from .vit_ops import grouped_layernorm as gln, grouped_linear as glin
    G, b, l, C = x.shape
    a1, a2, c1, c2 = blk1.attn, blk2.attn, blk1.cross_attn, blk2.cross_attn
    H = a1.num_heads
    rope = a1.rope
    h = gln(x, (blk1.norm1, blk2.norm1))
    qkv = glin(h, (a1.qkv, a2.qkv)).view(2 * b, l, 3, H, C // H)
    o = attention_qkv(qkv, a1.scale, pos if rope is not None else None, rope.freq if rope is not None else 100.0, rope.max_pos if rope is not None else 64)
    x1 = glin(o.reshape(2, b, l, C), (a1.proj, a2.proj), residual=x)
    y_ = gln(x, (blk1.norm_y, blk2.norm_y), flip=True)
    h = gln(x1, (blk1.norm2, blk2.norm2))
    q = glin(h, (c1.projq, c2.projq)).view(2 * b, l, H, C // H)
    k = glin(y_, (c1.projk, c2.projk)).view(2 * b, l, H, C // H)
    v = glin(y_, (c1.projv, c2.projv)).view(2 * b, l, H, C // H)
    if c1.rope is not None:
        o = memory_efficient_attention(q, k, v, scale=c1.scale, qpos=pos, kpos=mpos, rope_base=c1.rope.freq, max_pos=c1.rope.max_pos)
    else:
        o = memory_efficient_attention(q, k, v, scale=c1.scale)
    x2 = glin(o.reshape(2, b, l, C), (c1.proj, c2.proj), residual=x1)
    h = gln(x2, (blk1.norm3, blk2.norm3))
    hid = glin(h, (blk1.mlp.fc1, blk2.mlp.fc1), gelu=True)
    return glin(hid, (blk1.mlp.fc2, blk2.mlp.fc2), residual=x2)


LayerNorm6 = partial(LayerNorm, eps=1e-6)   # croco.py:34 norm_layer=partial(nn.LayerNorm, eps=1e-6); vit_ops.LayerNorm: same parameters, HIP kernels
