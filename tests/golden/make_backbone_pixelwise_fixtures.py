"""Generates tests/golden/backbone_pixelwise.npz from the REFERENCE's backbones with the PIXELWISE intrinsics embedding on a small trunk:
  * `get_intrinsic_embedding` itself (src/geometry/camera_emb.py:7-31) at degree 0, 4 and 8, full resolution and downsample 16 / merge_hw;
  * `AsymmetricCroCoMulti.forward` (backbone_croco_multiview.py:59-67,83-85,190-227) with intrinsics_embed_loc='encoder',
    intrinsics_embed_type='pixelwise', degree 4 (a 3 + 25 channel patch embed), 3 views;
  * `AsymmetricCroCo.forward` (backbone_croco.py:69-101,236-263) with intrinsics_embed_loc='decoder', type='pixelwise', degree 4
    (25 extra feature channels in front of decoder_embed).
Weights: tests/helpers.deterministic_init_ on both sides (keyed by state-dict name).
    python tests/golden/make_backbone_pixelwise_fixtures.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden.ref_stubs import install
from tests.helpers import deterministic_init_

TINY = dict(enc_depth=2, dec_depth=3, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512))
mods = install()
for tab in (mods.bc.croco_params, mods.bm.croco_params):
    tab["ViTLarge_BaseDecoder"] = dict(TINY)
bc, bm = mods.bc, mods.bm
from src.geometry.camera_emb import get_intrinsic_embedding

out = {}
g = torch.Generator().manual_seed(33)
H, W = 32, 48


def cfg(name, loc, typ):
    return bc.BackboneCrocoCfg(name=name, model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R", asymmetry_decoder=True,
                               intrinsics_embed_loc=loc, intrinsics_embed_degree=4, intrinsics_embed_type=typ)


def inputs(v):
    img = (torch.rand(2, v, 3, H, W, generator=g) * 2 - 1).requires_grad_(True)
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(2, v, 1, 1) + 0.02 * torch.rand(2, v, 3, 3, generator=g)
    return img, K


# ---- the embedding itself ----
img, K = inputs(3)
ctx = dict(image=img.detach(), intrinsics=K)
out.update(emb_image_shape=np.array(img.shape), emb_K=K.numpy())
for deg in (0, 4, 8):
    out[f"emb_full_{deg}"] = get_intrinsic_embedding(ctx, degree=deg).numpy()[:1, :2]       # (one scene, two views: the file stays small)
    out[f"emb_tok_{deg}"] = get_intrinsic_embedding(ctx, degree=deg, downsample=16, merge_hw=True).numpy()

# ---- multi-view trunk, encoder-side pixelwise embedding ----
m = bm.AsymmetricCroCoMulti(cfg("croco_multi", "encoder", "pixelwise"), 3).eval()
deterministic_init_(m)
img, K = inputs(3)
feat, pos, dec_feat, shape, images = m(dict(image=img, intrinsics=K))
dec_feat = list(dec_feat)
wa, wb = torch.randn(dec_feat[-1].shape, generator=g), torch.randn(dec_feat[1].shape, generator=g)
((dec_feat[-1] * wa).sum() + (dec_feat[1] * wb).sum()).backward()
out.update(pixe_image=img.detach().numpy(), pixe_K=K.numpy(), pixe_wa=wa.numpy(), pixe_wb=wb.numpy(), pixe_gimage=img.grad.numpy(),
           pixe_feat=feat.detach().numpy(), pixe_pos=pos.numpy(), pixe_d_last=dec_feat[-1].detach().numpy(), pixe_d_1=dec_feat[1].detach().numpy(),
           pixe_d_0=dec_feat[0].detach().numpy(), pixe_images=images.detach().numpy()[:1, :1], pixe_g_patch=m.patch_embed.proj.weight.grad.numpy()[::64],     # every 64th output channel
          
           pixe_keys=np.array(sorted(m.state_dict().keys())), pixe_patch_shape=np.array(m.patch_embed.proj.weight.shape))

# ---- pairwise backbone, decoder-side pixelwise embedding ----
m = bc.AsymmetricCroCo(cfg("croco", "decoder", "pixelwise"), 3).eval()
deterministic_init_(m)
img, K = inputs(2)
dec1, dec2, s1, s2 = m(dict(image=img, intrinsics=K))
dec1, dec2 = list(dec1), list(dec2)
w1, w2 = torch.randn(dec1[-1].shape, generator=g), torch.randn(dec2[1].shape, generator=g)
((dec1[-1] * w1).sum() + (dec2[1] * w2).sum()).backward()
out.update(pixd_image=img.detach().numpy(), pixd_K=K.numpy(), pixd_w1=w1.numpy(), pixd_w2=w2.numpy(), pixd_gimage=img.grad.numpy(),
           pixd_n=np.array(len(dec1)), pixd_g_embed=m.decoder_embed.weight.grad.numpy(), pixd_keys=np.array(sorted(m.state_dict().keys())),
           pixd_embed_shape=np.array(m.decoder_embed.weight.shape))
for i in range(len(dec1)):
    out[f"pixd_d1_{i}"] = dec1[i].detach().numpy(); out[f"pixd_d2_{i}"] = dec2[i].detach().numpy()
np.savez_compressed(Path(__file__).resolve().parent / "backbone_pixelwise.npz", **out)
print({k: getattr(v, "shape", None) for k, v in out.items() if not k.endswith("keys")})
