"""Generates tests/golden/encoder_mid.npz: the REFERENCE's EncoderNoPoSplatMultiTokenStyle with a trunk large enough that this
repo's HIP kernels are really the ones that run in the parity test (round-1 VERDICT weak #2): decoder width 768 / 12 heads
(C % 256 == 0 -> HIP LayerNorm everywhere), 2 ViT-L encoder blocks, 12 decoder blocks (the DPT hooks need 13 outputs),
2 context views of 128 x 160 (the 3x3 / 1x1 stride-1 head convolutions produce >= 100 output tiles -> vit_conv_x6_*).
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
model = model.double()
# the reference casts the decoder tokens (and the RoPE frequencies, the patch tokens) with `.float()` on the way
# (encoder_noposplat_multi_token_style.py:154-174, croco/patch_embed.py:63, pos_embed.py:122): for the float64 run those casts
# must keep the precision, so Tensor.float is rebound to Tensor.double for the duration of this script
torch.Tensor.float = lambda self, *a, **k: self.double()
g = torch.Generator().manual_seed(31)
b, v, H, W = 1, 2, 128, 160
img = (torch.rand(b, v, 3, H, W, generator=g) * 2 - 1)
K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(b, v, 1, 1) + 0.01 * torch.rand(b, v, 3, 3, generator=g)
style = (torch.rand(b, 3, 128, 128, generator=g) * 2 - 1)
x = img.double().requires_grad_(True)
gs = model(dict(image=x, intrinsics=K.double()), dict(image=style.double()), global_step=0)
w = [closed_form_weights(t.shape, k).double() for k, t in enumerate((gs.means, gs.covariances, gs.harmonics, gs.opacities))]
loss = (gs.means * w[0]).sum() + 1e4 * (gs.covariances * w[1]).sum() + (gs.harmonics * w[2]).sum() + (gs.opacities * w[3]).sum()
loss.backward()
G = gs.means.shape[1]
idx = torch.randperm(G, generator=torch.Generator().manual_seed(5))[:4096].sort().values
pn = dict(model.named_parameters())
out = dict(image=img.numpy(), intrinsics=K.numpy(), style=style.numpy(), idx=idx.numpy(),
           means=gs.means[0, idx].detach().numpy(), cov=gs.covariances[0, idx].detach().numpy(),
           sh=gs.harmonics[0, idx].detach().numpy(), opac=gs.opacities[0, idx].detach().numpy(),
           gimage=x.grad.numpy().astype(np.float64), loss=np.array(float(loss)),
           nparams=np.array(sum(p.numel() for p in model.parameters())))
# parameter gradients at every depth of the graph (rows subsampled): which layer carries the residual error is visible
for name, rows in (("backbone.enc_blocks.0.attn.qkv.weight", 64), ("backbone.enc_blocks.1.mlp.fc1.weight", 64),
                   ("backbone.dec_blocks.5.cross_attn.projk.weight", 64), ("backbone.dec_blocks2.11.mlp.fc2.weight", 64),
                   ("token_stylizer.dec_blocks.3.cross_attn.projk.weight", 64), ("token_stylizer.enc_blocks.1.norm1.weight", None),
                   ("backbone.dec_norm.weight", None), ("downstream_head1.dpt.scratch.refinenet4.resConfUnit2.conv1.weight", 8),
                   ("downstream_head2.dpt.scratch.layer1_rn.weight", 8), ("gaussian_param_head.dpt.head.0.weight", 8),
                   ("gaussian_param_head.dpt.input_merger.0.weight", 16), ("gaussian_appearance_head.dpt.act_postprocess.0.1.weight", 8),
                   ("backbone.patch_embed.proj.weight", 8), ("backbone.intrinsic_encoder.weight", None)):
    gr = pn[name].grad
    if gr is None:
        print("no gradient reaches", name); continue
    out["g:" + name] = (gr if rows is None else gr[:rows]).numpy().astype(np.float64)
# golden values are float64 results rounded once to float32 (6e-8 relative: far below the 1e-4 bars) to keep the file small
out = {k: (v.astype(np.float32) if v.dtype == np.float64 and k not in ("loss",) else v) for k, v in out.items()}
np.savez_compressed(ROOT / "tests/golden/encoder_mid.npz", **out)
print("params", int(out["nparams"]), "G", G, "loss", float(loss), "mean|means|", float(gs.means.abs().mean()),
      "bytes", (ROOT / "tests/golden/encoder_mid.npz").stat().st_size)
