"""Generates tests/golden/vit_blocks.npz by importing the REFERENCE's own ViT building blocks
(src/model/encoder/backbone/croco/{blocks,pos_embed}.py) in the build container.  xformers is not
installed: `memory_efficient_attention` is stubbed with its contract softmax(q k^T scale) v on
(B,N,H,D) tensors; RoPE2D resolves to the reference's torch fallback
(pos_embed.py:109-159) because the curope CUDA extension cannot be built here.
Fixture = data only: seeded inputs, the modules' state_dicts, outputs and gradients.
    python tests/golden/make_vit_fixtures.py
"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, REF)

xf = types.ModuleType("xformers"); xo = types.ModuleType("xformers.ops")
def mea(q, k, v, scale=None, p=0.0):
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    a = (q.permute(0, 2, 1, 3) @ k.permute(0, 2, 3, 1)) * scale
    return (a.softmax(-1) @ v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
xo.memory_efficient_attention = mea; xf.ops = xo
sys.modules["xformers"] = xf; sys.modules["xformers.ops"] = xo
for name in ("src", "src.model", "src.model.encoder", "src.model.encoder.backbone", "src.model.encoder.backbone.croco"):
    m = types.ModuleType(name); m.__path__ = [REF + "/" + name.replace(".", "/")]; sys.modules[name] = m
import importlib
pe = importlib.import_module("src.model.encoder.backbone.croco.pos_embed")
bl = importlib.import_module("src.model.encoder.backbone.croco.blocks")

torch.manual_seed(0)
out = {}
DIM, HEADS = 128, 2          # head_dim 64 as in every Styl3R model (1024/16, 768/12)
rope = pe.RoPE2D(freq=100.0)

def positions(b, n, with_intr=True):
    side = int(np.ceil(np.sqrt(n)))
    p = torch.cartesian_prod(torch.arange(side), torch.arange(side))[:n].clone()
    if with_intr:
        p[-1] = torch.tensor([16, 0])       # the intrinsics token's position (backbone_croco_multiview.py:133-135)
    return p[None].expand(b, -1, -1).clone()

# ---- RoPE forward / backward -----------------------------------------------------------
tok = torch.randn(2, 4, 21, 64, requires_grad=True)
pos = positions(2, 21)
r = rope(tok, pos)
w = torch.randn_like(r)
(r * w).sum().backward()
out.update(rope_tokens=tok.detach().numpy(), rope_pos=pos.numpy(), rope_out=r.detach().numpy(),
           rope_gout=w.numpy(), rope_gin=tok.grad.numpy())

def save_module(prefix, mod):
    for k, v in mod.state_dict().items():
        out[f"{prefix}_sd_{k}"] = v.numpy()

def run(prefix, mod, inputs, call):
    mod.eval()
    ins = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in inputs.items()}
    y = call(mod, ins)
    y0 = y[0] if isinstance(y, tuple) else y
    w = torch.randn_like(y0)
    (y0 * w).sum().backward()
    save_module(prefix, mod)
    for k, v in ins.items():
        out[f"{prefix}_in_{k}"] = v.detach().numpy()
        if v.is_floating_point():
            out[f"{prefix}_gin_{k}"] = v.grad.numpy()
    out[f"{prefix}_out"] = y0.detach().numpy(); out[f"{prefix}_gout"] = w.numpy()
    for k, p in mod.named_parameters():     # a representative subset keeps the fixture small
        if any(t in k for t in ("qkv.weight", "projk.weight", "proj.bias", "norm1.weight", "fc1.bias", "norm_y.bias")):
            out[f"{prefix}_gp_{k}"] = p.grad.numpy()

x = torch.randn(2, 21, DIM); y = torch.randn(2, 37, DIM)
xpos, ypos = positions(2, 21), positions(2, 37)
run("mlp", bl.Mlp(DIM, 2 * DIM), dict(x=x), lambda m, i: m(i["x"]))
run("attn", bl.Attention(DIM, rope=rope, num_heads=HEADS, qkv_bias=True), dict(x=x, xpos=xpos), lambda m, i: m(i["x"], i["xpos"]))
run("block", bl.Block(DIM, HEADS, 1.0, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6), rope=rope),
    dict(x=x, xpos=xpos), lambda m, i: m(i["x"], i["xpos"]))
run("xattn", bl.CrossAttention(DIM, rope=rope, num_heads=HEADS, qkv_bias=True), dict(q=x, kv=y, qpos=xpos, kpos=ypos),
    lambda m, i: m(i["q"], i["kv"], i["kv"], i["qpos"], i["kpos"]))
run("dec", bl.DecoderBlock(DIM, HEADS, 1.0, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6), rope=rope),
    dict(x=x, y=y, xpos=xpos, ypos=ypos), lambda m, i: m(i["x"], i["y"], i["xpos"], i["ypos"]))
np.savez_compressed(ROOT / "tests/golden/vit_blocks.npz", **out)
print("wrote", len(out), "arrays")
