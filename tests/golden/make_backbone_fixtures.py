"""Generates tests/golden/backbone_variants.npz from the REFERENCE's two CroCo backbones on a small trunk:
  * `AsymmetricCroCo.forward` (src/model/encoder/backbone/backbone_croco.py:220-286), the PAIRWISE `croco` backbone with the 'token'
    intrinsics embedding: per-view decoder outputs dec1 / dec2 (13 tensors each, token stripped) and gradients;
  * `AsymmetricCroCoMulti.forward` (backbone_croco_multiview.py:190-227) with intrinsics_embed_type='linear' (3 views) and with
    intrinsics_embed_loc='none' (2 views): decoder features and gradients.
Weights: tests/helpers.deterministic_init_ on both sides (keyed by state-dict name).
    python tests/golden/make_backbone_fixtures.py
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
out = {}
g = torch.Generator().manual_seed(21)
H, W = 32, 48


def cfg(name, loc, typ):
    return bc.BackboneCrocoCfg(name=name, model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R", asymmetry_decoder=True,
                               intrinsics_embed_loc=loc, intrinsics_embed_degree=4, intrinsics_embed_type=typ)


def inputs(v):
    img = (torch.rand(2, v, 3, H, W, generator=g) * 2 - 1).requires_grad_(True)
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(2, v, 1, 1) + 0.02 * torch.rand(2, v, 3, 3, generator=g)
    return img, K


# ---- pairwise croco, token ----
m = bc.AsymmetricCroCo(cfg("croco", "encoder", "token"), 3).eval()
deterministic_init_(m)
img, K = inputs(2)
dec1, dec2, s1, s2 = m(dict(image=img, intrinsics=K))
dec1, dec2 = list(dec1), list(dec2)
w1, w2, w3 = (torch.randn(t.shape, generator=g) for t in (dec1[-1], dec2[-1], dec2[1]))
((dec1[-1] * w1).sum() + (dec2[-1] * w2).sum() + (dec2[1] * w3).sum()).backward()
out.update(pair_image=img.detach().numpy(), pair_K=K.numpy(), pair_w1=w1.numpy(), pair_w2=w2.numpy(), pair_w3=w3.numpy(),
           pair_gimage=img.grad.numpy(), pair_n=np.array(len(dec1)), pair_shape=s1.numpy(),
           pair_g_dec2_qkv=m.dec_blocks2[1].attn.qkv.weight.grad.numpy(), pair_keys=np.array(sorted(m.state_dict().keys())))
for i in range(len(dec1)):
    out[f"pair_d1_{i}"] = dec1[i].detach().numpy(); out[f"pair_d2_{i}"] = dec2[i].detach().numpy()

# ---- multi-view, 'linear' embedding (3 views) and no embedding (2 views) ----
for tag, loc, typ, v in (("lin", "encoder", "linear", 3), ("none", "none", "token", 2)):
    m = bm.AsymmetricCroCoMulti(cfg("croco_multi", loc, typ), 3).eval()
    deterministic_init_(m)
    img, K = inputs(v)
    feat, pos, dec_feat, shape, images = m(dict(image=img, intrinsics=K))
    dec_feat = list(dec_feat)
    wa, wb = torch.randn(dec_feat[-1].shape, generator=g), torch.randn(dec_feat[1].shape, generator=g)
    ((dec_feat[-1] * wa).sum() + (dec_feat[1] * wb).sum()).backward()
    out.update({f"{tag}_image": img.detach().numpy(), f"{tag}_K": K.numpy(), f"{tag}_wa": wa.numpy(), f"{tag}_wb": wb.numpy(),
                f"{tag}_gimage": img.grad.numpy(), f"{tag}_feat": feat.detach().numpy(), f"{tag}_pos": pos.numpy(),
                f"{tag}_d_last": dec_feat[-1].detach().numpy(), f"{tag}_d_1": dec_feat[1].detach().numpy(), f"{tag}_d_0": dec_feat[0].detach().numpy(),
                f"{tag}_keys": np.array(sorted(m.state_dict().keys()))})
np.savez_compressed(Path(__file__).resolve().parent / "backbone_variants.npz", **out)
print({k: getattr(v, "shape", None) for k, v in out.items() if not k.endswith("keys")})
