"""Generates tests/golden/encoder_mid.npz: the REFERENCE's EncoderNoPoSplatMultiTokenStyle with a trunk large enough that this
repo's HIP kernels are really the ones that run in the parity test (round-1 VERDICT weak #2): decoder width 768 / 12 heads
(C % 256 == 0 -> HIP LayerNorm everywhere), 2 ViT-L encoder blocks, 12 decoder blocks (the DPT hooks need 13 outputs),
2 context views of 256 x 256 = the C2 / C3 image size (the 3x3 / 1x1 stride-1 head convolutions produce >= 100 output tiles ->
vit_conv_x6_fwd / dX, and B*H*W >= 65 536 pixels -> vit_conv_x6_wgrad).
The reference runs in FLOAT64 on the CPU here, so the golden values carry no fp32 noise of their own: what the test
measures is the HIP path's fp32 error alone.  Weights are regenerated on both sides from the parameter names
(tests/helpers.deterministic_init_), the loss weights from a closed form (tests/helpers.closed_form_weights) -- neither
is stored; outputs are stored on a fixed subset of Gaussians to keep the file small.
    python tests/golden/make_encoder_mid_fixtures.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden.ref_stubs import install, style_encoder_cfg
from tests.helpers import closed_form_weights, deterministic_init_

MID = dict(enc_depth=2, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12,
           pos_embed="RoPE100", img_size=(512, 512))
mods = install()
mods.bm.croco_params["ViTLarge_BaseDecoder"] = dict(MID)
mods.ts.croco_params["ViTLarge_BaseDecoder"] = dict(MID)
enc_mod, cfg = style_encoder_cfg(mods, sh_degree=0)
torch.manual_seed(0)
model = enc_mod.EncoderNoPoSplatMultiTokenStyle(cfg).eval()
deterministic_init_(model)
g = torch.Generator().manual_seed(31)
b, v, H, W = 1, 2, 256, 256
img8 = torch.randint(0, 256, (b, v, 3, H, W), generator=g, dtype=torch.uint8)     # stored as bytes: exact on both sides
img = img8.float() / 127.5 - 1
K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(b, v, 1, 1) + 0.01 * torch.rand(b, v, 3, 3, generator=g)
style = (torch.rand(b, 3, 128, 128, generator=g) * 2 - 1)
GRADS = (("backbone.enc_blocks.0.attn.qkv.weight", 64), ("backbone.enc_blocks.1.mlp.fc1.weight", 64),
         ("backbone.dec_blocks.5.cross_attn.projk.weight", 64), ("backbone.dec_blocks2.11.mlp.fc2.weight", 64),
         ("token_stylizer.dec_blocks.3.cross_attn.projk.weight", 64), ("token_stylizer.enc_blocks.1.norm1.weight", None),
         ("backbone.dec_norm.weight", None), ("downstream_head1.dpt.scratch.refinenet4.resConfUnit2.conv1.weight", 8),
         ("downstream_head2.dpt.scratch.layer1_rn.weight", 8), ("gaussian_param_head.dpt.head.0.weight", 8),
         ("gaussian_param_head.dpt.input_merger.0.weight", 16), ("gaussian_appearance_head.dpt.act_postprocess.0.1.weight", 8),
         ("backbone.patch_embed.proj.weight", 8), ("backbone.intrinsic_encoder.weight", None))


def run(model, dtype):
    for p in model.parameters():
        p.grad = None
    x = img.detach().to(dtype).clone().requires_grad_(True)
    gs = model(dict(image=x, intrinsics=K.to(dtype)), dict(image=style.to(dtype)), global_step=0)
    w = [closed_form_weights(t.shape, k).to(dtype) for k, t in enumerate((gs.means, gs.covariances, gs.harmonics, gs.opacities))]
    loss = (gs.means * w[0]).sum() + 1e4 * (gs.covariances * w[1]).sum() + (gs.harmonics * w[2]).sum() + (gs.opacities * w[3]).sum()
    loss.backward()
    pn = dict(model.named_parameters())
    res = dict(means=gs.means.detach(), cov=gs.covariances.detach(), sh=gs.harmonics.detach(), opac=gs.opacities.detach(),
               gimage=x.grad.detach(), loss=float(loss))
    for name, rows in GRADS:
        gr = pn[name].grad
        if gr is None:
            print("no gradient reaches", name); continue
        res["g:" + name] = (gr if rows is None else gr[:rows]).detach().clone()
    return res


# (1) the reference exactly as it ships: float32 on the CPU.  Only its DISTANCE to the float64 run is stored: the yardstick of
#     what fp32 round-off does to each quantity under this loss (the weighted sums over 131 072 Gaussians cancel heavily, so
#     deep gradients carry 1e-3-level noise in ANY fp32 evaluation, the reference's own included)
r32 = run(model, torch.float32)
# (2) float64: the golden values.  The reference casts the decoder tokens (and the RoPE frequencies, the patch tokens) with
#     `.float()` on the way (encoder_noposplat_multi_token_style.py:154-174, croco/patch_embed.py:63, pos_embed.py:122): for this
#     run those casts must keep the precision, so Tensor.float is rebound to Tensor.double for the rest of the script
model = model.double()
torch.Tensor.float = lambda self, *a, **k: self.double()
r64 = run(model, torch.float64)
rel = lambda a, e: float((a.double() - e).abs().max() / e.abs().max().clamp_min(1e-300))
noise = {k: rel(r32[k], r64[k]) for k in r64 if k != "loss"}
noise["loss"] = abs(r32["loss"] - r64["loss"]) / abs(r64["loss"])
print("fp32 reference vs fp64 reference:", {k: f"{v:.1e}" for k, v in noise.items()})
gs_means = r64["means"]
G = gs_means.shape[1]
idx = torch.randperm(G, generator=torch.Generator().manual_seed(5))[:4096].sort().values
out = dict(image_u8=img8.numpy(), intrinsics=K.numpy(), style=style.numpy(), idx=idx.numpy(),
           means=r64["means"][0, idx].numpy(), cov=r64["cov"][0, idx].numpy(), sh=r64["sh"][0, idx].numpy(),
           opac=r64["opac"][0, idx].numpy(), gimage_s2=r64["gimage"][..., ::2, ::2].numpy(),   # every 2nd pixel
           loss=np.array(r64["loss"]), nparams=np.array(sum(p.numel() for p in model.parameters())))
for k in r64:
    if k.startswith("g:"):
        out[k] = r64[k].numpy()
for k, v in noise.items():
    out["fp32noise:" + k] = np.array(v)
# golden values are float64 results rounded once to float32 (6e-8 relative: far below the 1e-4 bars) to keep the file small
out = {k: (v.astype(np.float32) if v.dtype == np.float64 and k not in ("loss",) else v) for k, v in out.items()}
np.savez_compressed(ROOT / "tests/golden/encoder_mid.npz", **out)
print("params", int(out["nparams"]), "G", G, "loss", r64["loss"], "mean|means|", float(gs_means.abs().mean()),
      "bytes", (ROOT / "tests/golden/encoder_mid.npz").stat().st_size)
