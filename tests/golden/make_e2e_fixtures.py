"""Generates tests/golden/e2e_<tag>.npz: the REFERENCE's training chain end to end --
    gaussians = encoder(context, style, step); output = decoder.forward(gaussians, extr, intr, near, far, (h, w)); MSE
(src/model/model_wrapper_style.py:189-198, src/loss/loss_mse.py:22-31) -- evaluated in FLOAT64 on the CPU: the reference's own
EncoderNoPoSplatMultiTokenStyle, the reference's own DecoderSplattingCUDA / render_cuda, and in place of the absent third-party
CUDA rasterizer (requirements.txt:17) the float64 build of oracle/gsr_oracle.c behind the `diff_gaussian_rasterization` names
(forward AND backward, so the loss back-propagates through render_cuda into the encoder exactly as in training).

Three runs of the same chain:
  r64   float64 everywhere                                   -> the golden values
  r32   the reference as it ships (float32, f32 oracle)      -> `fp32noise:*`  = its distance to r64
  rtf   float64, but every Linear / Conv2d / ConvTranspose2d product takes operands rounded to TF32 (10-bit mantissa), forward and
        backward, as on the GPUs the reference was developed on (croco.py:13 `allow_tf32 = True`; cudnn's conv TF32 default)
                                                            -> `tf32noise:*`  = its distance to r64
Tags:  full = the STOCK model of every README command (ViTLarge_BaseDecoder: 24 ViT-L blocks in the backbone encoder and 24 in the style
              encoder, 12 + 12 + 12 decoder blocks; 1 049 635 033 parameters; backbone_croco_multiview.py:21-32, token_stylizer.py:36-48), 1 scene,
              2 context views 256 x 256, 2 target views;  the two tags below use the 2-block trunk:
       c3 = 1 scene, 2 context views 256 x 256, 2 target views (NVS-pretrain shapes);
       c4 = 1 scene, 4 context views 128 x 160, 2 target views (the 4-view style-stage structure: dec_blocks2 on 3 views, stylizer on 4).
Trunk: decoder width 768 / 12 heads, 2 ViT-L encoder blocks, 12 decoder blocks (as encoder_mid.npz).  Weights are regenerated on both
sides from the parameter names; only the five 1x1 output convolutions (`*.dpt.head.4`), re-centred so that the random-init model
emits a usable scene (depth 2..4 inside the frustum instead of expm1 of a heavy-tailed norm), are stored.
    python tests/golden/make_e2e_fixtures.py c3 ;  python tests/golden/make_e2e_fixtures.py c4 ;  python tests/golden/make_e2e_fixtures.py full
"""
import sys
import time
import types
from collections import namedtuple
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
F32 = torch.float32                                               # kept: torch.float32 itself is rebound for the float64 run below
sys.path.insert(0, str(ROOT))
from oracle.gsr_oracle import Oracle
from tests.golden.ref_stubs import install, style_encoder_cfg
from tests.helpers import (closed_form_image, deterministic_init_, e2e_cameras, e2e_fragile_mask, E2E_HEAD_TARGETS)

TAG = sys.argv[1] if len(sys.argv) > 1 else "c3"
SHAPES = dict(c3=dict(v=2, H=256, W=256), c4=dict(v=4, H=128, W=160), full=dict(v=2, H=256, W=256))[TAG]
MID = dict(enc_depth=2, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12,
           pos_embed="RoPE100", img_size=(512, 512))
NTHREADS = 8

# ---- `diff_gaussian_rasterization`, backed by the CPU oracle (both precisions), with autograd ------------------------------
Settings = namedtuple("GaussianRasterizationSettings", "image_height image_width tanfovx tanfovy bg scale_modifier "
                      "viewmatrix projmatrix projmatrix_raw sh_degree campos prefiltered debug")
STATES = []          # (FwdState, {means, proj}) of every forward, in call order


class _OracleRaster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, cov6, s):
        prec = "f64" if means3D.dtype == torch.float64 else "f32"
        orc = Oracle(prec)
        n = lambda t: t.detach().numpy()
        st, octx = orc.forward(n(means3D), n(cov6), n(opacities)[:, 0], shs=n(shs), H=s.image_height, W=s.image_width,
                               tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=n(s.bg), view=n(s.viewmatrix).reshape(-1),
                               proj=n(s.projmatrix).reshape(-1), proj_raw=n(s.projmatrix_raw).reshape(-1), campos=n(s.campos),
                               sh_degree=s.sh_degree, nthreads=NTHREADS)
        ctx.orc, ctx.st, ctx.octx = orc, st, octx
        STATES.append((st, dict(means=octx["means"].copy(), proj=n(s.projmatrix).reshape(4, 4).copy())))
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(means3D.dtype)
        return (T(st.image), torch.from_numpy(st.radii.copy()), T(st.out_depth)[None], T(st.out_opacity)[None],
                torch.from_numpy(st.n_touched.copy()))

    @staticmethod
    def backward(ctx, g_img, g_radii, g_depth, g_opac, g_nt):
        gr = ctx.orc.backward(ctx.st, ctx.octx, g_img.numpy(), g_depth[0].numpy(), nthreads=NTHREADS)
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(g_img.dtype)
        return T(gr["means3D"]), T(gr["means2D"]), T(gr["shs"]), T(gr["opacities"])[:, None], T(gr["cov6"]), None


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__(); self.s = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                theta=None, rho=None):
        assert shs is not None and cov3D_precomp is not None and theta is None and rho is None
        return _OracleRaster.apply(means3D, means2D, shs, opacities, cov3D_precomp, self.s)


dgr = types.ModuleType("diff_gaussian_rasterization")
dgr.GaussianRasterizationSettings = Settings
dgr.GaussianRasterizer = GaussianRasterizer
sys.modules["diff_gaussian_rasterization"] = dgr

mods = install()
if TAG != "full":                                                  # `full`: the stock ViTLarge_BaseDecoder (24 + 24 ViT-L blocks, 12 + 12 + 12 decoder blocks)
    mods.bm.croco_params["ViTLarge_BaseDecoder"] = dict(MID)
    mods.ts.croco_params["ViTLarge_BaseDecoder"] = dict(MID)
enc_mod, cfg = style_encoder_cfg(mods, sh_degree=0)
import importlib
dsc = importlib.import_module("src.model.decoder.decoder_splatting_cuda")
torch.manual_seed(0)
model = enc_mod.EncoderNoPoSplatMultiTokenStyle(cfg).eval()
deterministic_init_(model)
decoder = dsc.DecoderSplattingCUDA(dsc.DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True))

b, v, H, W = 1, SHAPES["v"], SHAPES["H"], SHAPES["W"]
g = torch.Generator().manual_seed(47 + v)
img8 = torch.randint(0, 256, (b, v, 3, H, W), generator=g, dtype=torch.uint8)
img = img8.float() / 127.5 - 1
K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(b, v, 1, 1) + 0.01 * torch.rand(b, v, 3, 3, generator=g)
style = torch.rand(b, 3, 128, 128, generator=g) * 2 - 1
cams = e2e_cameras(b)                                             # target cameras: closed form, shared with the test
target = closed_form_image((b, cams["extrinsics"].shape[1], 3, H, W))

# ---- re-centre the five output convolutions (one calibration forward in fp32; the result is stored) ----------------------
HEADS = [k for k in E2E_HEAD_TARGETS if hasattr(model, k)]
stats = {}
hooks = [getattr(model, k).dpt.head[4].register_forward_hook(
    lambda m, i, o, k=k: stats.setdefault(k, []).append(o.detach().transpose(0, 1).reshape(o.shape[1], -1))) for k in HEADS]
with torch.no_grad():
    model(dict(image=img, intrinsics=K), dict(image=style), global_step=0)
for h_ in hooks:
    h_.remove()
head_tensors = {}
with torch.no_grad():
    for k in HEADS:
        o = torch.cat(stats[k], 1).double()
        mean, std = o.mean(1), o.std(1)
        t_mean, t_std = (torch.tensor(x, dtype=torch.float64) for x in E2E_HEAD_TARGETS[k])
        conv = getattr(model, k).dpt.head[4]
        kk = (t_std / std)
        conv.weight.copy_((conv.weight.double() * kk[:, None, None, None]).float())
        conv.bias.copy_(((conv.bias.double() - mean) * kk + t_mean).float())
        head_tensors[f"head:{k}.dpt.head.4.weight"] = conv.weight.detach().numpy().copy()
        head_tensors[f"head:{k}.dpt.head.4.bias"] = conv.bias.detach().numpy().copy()

GRADS = (("backbone.enc_blocks.0.attn.qkv.weight", 64), ("backbone.enc_blocks.1.mlp.fc1.weight", 64),
         ("backbone.dec_blocks.5.cross_attn.projk.weight", 64), ("backbone.dec_blocks2.11.mlp.fc2.weight", 64),
         ("token_stylizer.dec_blocks.3.cross_attn.projk.weight", 64), ("token_stylizer.enc_blocks.1.norm1.weight", None),
         ("backbone.dec_norm.weight", None), ("downstream_head1.dpt.scratch.refinenet4.resConfUnit2.conv1.weight", 8),
         ("downstream_head2.dpt.scratch.layer1_rn.weight", 8), ("gaussian_param_head.dpt.head.0.weight", 8),
         ("gaussian_param_head.dpt.input_merger.0.weight", 16), ("gaussian_appearance_head.dpt.act_postprocess.0.1.weight", 8),
         ("backbone.patch_embed.proj.weight", 8), ("backbone.intrinsic_encoder.weight", None))


if TAG == "full":                                                  # the model every README command runs: gradient slices across the real depth
    GRADS = (("backbone.enc_blocks.0.attn.qkv.weight", 64), ("backbone.enc_blocks.12.mlp.fc1.weight", 64), ("backbone.enc_blocks.23.attn.proj.weight", 64),
             ("backbone.enc_blocks.23.mlp.fc2.weight", 64), ("token_stylizer.enc_blocks.0.attn.qkv.weight", 64), ("token_stylizer.enc_blocks.23.mlp.fc1.weight", 64),
             ("token_stylizer.enc_blocks.12.norm1.weight", None),
             ("backbone.dec_blocks.5.cross_attn.projk.weight", 64), ("backbone.dec_blocks2.11.mlp.fc2.weight", 64),
             ("token_stylizer.dec_blocks.3.cross_attn.projk.weight", 64), ("backbone.dec_norm.weight", None), ("backbone.enc_norm.weight", None),
             ("downstream_head1.dpt.scratch.refinenet4.resConfUnit2.conv1.weight", 8), ("downstream_head2.dpt.scratch.layer1_rn.weight", 8),
             ("gaussian_param_head.dpt.head.0.weight", 8), ("gaussian_param_head.dpt.input_merger.0.weight", 16),
             ("gaussian_appearance_head.dpt.act_postprocess.0.1.weight", 8), ("backbone.patch_embed.proj.weight", 8),
             ("backbone.intrinsic_encoder.weight", None))

OK_MASK = None          # (b, v_t, 1, H, W) bool: pixels that enter the loss (set after the preliminary float64 forward below)


def run(model, decoder, dtype, no_grad=False):
    STATES.clear()
    for p in model.parameters():
        p.grad = None
    x = img.detach().to(dtype).clone().requires_grad_(True)
    c = {k: t.to(dtype) for k, t in cams.items()}
    t0 = time.time()
    if no_grad:
        with torch.no_grad():
            gs = model(dict(image=x, intrinsics=K.to(dtype)), dict(image=style.to(dtype)), global_step=0)
            decoder.forward(gs, c["extrinsics"], c["intrinsics"], c["near"], c["far"], (H, W))
        return None, list(STATES)
    gs = model(dict(image=x, intrinsics=K.to(dtype)), dict(image=style.to(dtype)), global_step=0)
    for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities):
        t.retain_grad()
    out = decoder.forward(gs, c["extrinsics"], c["intrinsics"], c["near"], c["far"], (H, W))
    # LossMse (weight 1) over the pixels that are NOT discontinuity-adjacent: a flipped alpha >= 1/255 decision in a masked pixel (any
    # fp32 encoder flips some, run to run) must not leak into every gradient through the loss
    loss = (((out.color - target.to(dtype)) ** 2) * OK_MASK.to(dtype)).mean()
    loss.backward()
    print(f"  run {dtype}: {time.time() - t0:.0f} s, loss {float(loss):.6f}", flush=True)
    pn = dict(model.named_parameters())
    res = dict(means=gs.means.detach(), cov=gs.covariances.detach(), sh=gs.harmonics.detach(), opac=gs.opacities.detach(),
               color=out.color.detach(), depth=out.depth.detach(), gimage=x.grad.detach(), gmeans=gs.means.grad.detach(),
               gopac=gs.opacities.grad.detach(), gsh=gs.harmonics.grad.detach(), loss=float(loss))
    for name, rows in GRADS:
        gr = pn[name].grad
        if gr is None:
            print("no gradient reaches", name); continue
        res["g:" + name] = (gr if rows is None else gr[:rows]).detach().clone()
    return res, list(STATES)


# ---- TF32 operand rounding (round to nearest, ties away: cvt.rna.tf32.f32) for every Linear / Conv product, both directions ------
def tf32(t):
    x = t.detach().to(F32).contiguous()
    bits = x.view(torch.int32)
    r = ((bits + 0x1000) & ~0x1FFF).view(F32)
    r = torch.where(torch.isfinite(x), r, x)
    return r.to(t.dtype)


class _LinearTF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w); ctx.has_bias = bias is not None
        return _F_linear(tf32(x), tf32(w), bias)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gyr = tf32(gy)
        gx = gyr @ tf32(w)
        gw = gyr.reshape(-1, gy.shape[-1]).t() @ tf32(x).reshape(-1, x.shape[-1])
        return gx, gw, (gy.reshape(-1, gy.shape[-1]).sum(0) if ctx.has_bias else None)


class _ConvTF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, transposed):
        ctx.save_for_backward(x, w); ctx.cfg = (stride, padding, transposed, bias is not None)
        f = _F_convT if transposed else _F_conv
        return f(tf32(x), tf32(w), bias, stride, padding)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, transposed, has_bias = ctx.cfg
        xr, wr = tf32(x).requires_grad_(True), tf32(w).requires_grad_(True)
        with torch.enable_grad():
            y = (_F_convT if transposed else _F_conv)(xr, wr, None, stride, padding)
        gx, gw = torch.autograd.grad(y, (xr, wr), tf32(gy))
        return gx, gw, (gy.sum((0, 2, 3)) if has_bias else None), None, None, None


_F_linear, _F_conv, _F_convT = F.linear, F.conv2d, F.conv_transpose2d


def patch_tf32(on):
    if on:
        F.linear = lambda x, w, b=None: _LinearTF32.apply(x, w, b)
        F.conv2d = lambda x, w, b=None, stride=1, padding=0, dilation=1, groups=1: _ConvTF32.apply(x, w, b, stride, padding, False)
        F.conv_transpose2d = (lambda x, w, b=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1:
                              _ConvTF32.apply(x, w, b, stride, padding, True))
    else:
        F.linear, F.conv2d, F.conv_transpose2d = _F_linear, _F_conv, _F_convT


print(f"[{TAG}] b={b} v={v} {H}x{W}", flush=True)
# preliminary float64 forward (the model itself, converted and converted back -- exact for float32 weights; a deepcopy would keep calling
# the ORIGINAL heads through the closures of transpose_to_landscape; the global float64 rebinds are undone afterwards): its oracle states give
# the mask of discontinuity-adjacent pixels -- alpha-threshold within 0.5 %, depth near-ties among the visible contributors, termination
# within 2 %, tile-rectangle membership within 4e-3 px (calibrated so that no flip of the reference's own fp32 run survives the mask while
# > 90 % of the pixels stay in) -- which is excluded from the loss AND from the image comparison
_orig_float, _orig_f32 = torch.Tensor.float, torch.float32
model.double(); decoder.double()
torch.Tensor.float = lambda self, *a, **k: self.double()
torch.float32 = torch.float64
_, states0 = run(model, decoder, torch.float64, no_grad=True)
torch.Tensor.float, torch.float32 = _orig_float, _orig_f32
model.float(); decoder.float()
frag = np.stack([e2e_fragile_mask(st, H, W, **extra) for st, extra in states0]).reshape(b, -1, H, W)
OK_MASK = torch.from_numpy(~frag)[:, :, None]
print("fragile pixel fraction per view:", frag.reshape(frag.shape[1], -1).mean(1))

r32, _ = run(model, decoder, torch.float32)
model = model.double(); decoder = decoder.double()
torch.Tensor.float = lambda self, *a, **k: self.double()          # the reference's `.float()` casts must keep float64 (see make_encoder_mid_fixtures.py)
torch.float32 = torch.float64                                     # get_projection_matrix / get_fov allocate with dtype=torch.float32 (cuda_splatting.py:35, projection.py:251)
r64, states = run(model, decoder, torch.float64)
patch_tf32(True)
rtf, _ = run(model, decoder, torch.float64)
patch_tf32(False)

rel = lambda a, e: float((a.double() - e.double()).abs().max() / e.double().abs().max().clamp_min(1e-300))
if "--debug-dump" in sys.argv:                                    # mask calibration only (build_tmp/, not shipped)
    import pickle
    pickle.dump(dict(states=[st for st, _ in states], extra=[e for _, e in states], c32=r32["color"].numpy(), c64=r64["color"].numpy(), ctf=rtf["color"].numpy()),
                open(ROOT / f"build_tmp/e2e_{TAG}_dbg.pkl", "wb"))
ok = OK_MASK.expand_as(r64["color"])


def noise(r):
    d = {k: rel(r[k], r64[k]) for k in r64 if k not in ("loss", "color", "depth")}
    d["loss"] = abs(r["loss"] - r64["loss"]) / abs(r64["loss"])
    d["color"] = float(((r["color"].double() - r64["color"]).abs() * ok).max() / r64["color"].abs().max())
    d["color_all"] = rel(r["color"], r64["color"])
    d["depth"] = float(((r["depth"].double() - r64["depth"]).abs() * ok[:, :, 0]).max() / r64["depth"].abs().max())
    return d


n32, ntf = noise(r32), noise(rtf)
for k in n32:
    print(f"  {k:72s} fp32 {n32[k]:.2e}   tf32 {ntf[k]:.2e}")
G = r64["means"].shape[1]
idx = torch.randperm(G, generator=torch.Generator().manual_seed(5))[:4096].sort().values
out = dict(image_u8=img8.numpy(), intrinsics=K.numpy(), style=style.numpy(), idx=idx.numpy(),
           means=r64["means"][0, idx].numpy(), cov=r64["cov"][0, idx].numpy(), sh=r64["sh"][0, idx].numpy(), opac=r64["opac"][0, idx].numpy(),
           gmeans=r64["gmeans"][0, idx].numpy(), gopac=r64["gopac"][0, idx].numpy(), gsh=r64["gsh"][0, idx].numpy(),
           color=r64["color"].numpy(), depth=r64["depth"].numpy(), fragile=np.packbits(frag), gimage_s2=r64["gimage"][..., ::2, ::2].numpy(),
           loss=np.array(r64["loss"]), nparams=np.array(sum(p.numel() for p in model.parameters())), **head_tensors)
for k in r64:
    if k.startswith("g:"):
        out[k] = r64[k].numpy()
for k in n32:
    out["fp32noise:" + k] = np.array(n32[k]); out["tf32noise:" + k] = np.array(ntf[k])
out = {k: (v_.astype(np.float32) if v_.dtype == np.float64 and k != "loss" else v_) for k, v_ in out.items()}
dst = ROOT / f"tests/golden/e2e_{TAG}.npz"
np.savez_compressed(dst, **out)
print("wrote", dst, dst.stat().st_size, "bytes; G", G, "loss", r64["loss"], "visible radii>0:",
      [int((st.radii > 0).sum()) for st, _ in states], "R:", [st.R for st, _ in states])
