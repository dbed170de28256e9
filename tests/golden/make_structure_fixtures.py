"""Generates tests/golden/structure_builder.npz from the REFERENCE: outputs / gradients of `StructureBuilder.forward`
(src/model/encoder/token_stylizer/structure_builder.py:128-141) on a small trunk, and the state-dict key set + parameter
count of `EncoderNoPoSplatTokenStyle.__init__` (src/model/encoder/encoder_noposplat_token_style.py:73-116; its forward cannot
run against the reference's current backbone / token stylizer -- see styl3r_amd/encoder.py).
    python tests/golden/make_structure_fixtures.py
"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden.ref_stubs import install, style_encoder_cfg
from tests.helpers import deterministic_init_

TINY = dict(enc_depth=1, dec_depth=12, enc_embed_dim=128, dec_embed_dim=128, enc_num_heads=2, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512))
mods = install()
for tab in (mods.bc.croco_params, mods.bm.croco_params, mods.ts.croco_params, mods.sb.croco_params):
    tab["ViTLarge_BaseDecoder"] = dict(TINY)
sbm = mods.sb.StructureBuilder(mods.sb.StructureBuilderCfg("ViTLarge_BaseDecoder")).eval()
deterministic_init_(sbm)
g = torch.Generator().manual_seed(3)
B, L = 2, 7                                  # 6 patch tokens (2 x 3 grid) + the intrinsics token per view
f1 = torch.randn(B, L, 128, generator=g).requires_grad_(True)
f2 = torch.randn(B, L, 128, generator=g).requires_grad_(True)
yx = torch.stack(torch.meshgrid(torch.arange(2), torch.arange(3), indexing="ij"), -1).reshape(1, 6, 2)
pos = torch.cat((yx, torch.tensor([[[2, 0]]])), 1).expand(B, L, 2).contiguous()        # token at (rows, 0)
d1, d2 = sbm(f1, pos, f2, pos)
w = [torch.randn(t.shape, generator=g) for t in (d1[-1], d2[-1], d1[6])]
((d1[-1] * w[0]).sum() + (d2[-1] * w[1]).sum() + (d1[6] * w[2]).sum()).backward()
out = dict(f1=f1.detach().numpy(), f2=f2.detach().numpy(), pos=pos.numpy(), w0=w[0].numpy(), w1=w[1].numpy(), w2=w[2].numpy(),
           gf1=f1.grad.numpy(), gf2=f2.grad.numpy(), g_qkv5=sbm.dec_blocks[5].attn.qkv.weight.grad.numpy(), n_out=np.array(len(d1)))
for i in (0, 1, 6, 12):
    out[f"d1_{i}"] = d1[i].detach().numpy(); out[f"d2_{i}"] = d2[i].detach().numpy()
# ---- the 2-view style encoder's constructor: key set and parameter count ----
enc_mod, cfg = style_encoder_cfg(mods, sh_degree=1)
ets = importlib.import_module("src.model.encoder.encoder_noposplat_token_style")
cfg.name = "noposplat_token_style"
cfg.backbone = mods.bc.BackboneCrocoCfg(name="croco", model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R",
                                        asymmetry_decoder=True, intrinsics_embed_loc="encoder", intrinsics_embed_degree=4,
                                        intrinsics_embed_type="token")
m = ets.EncoderNoPoSplatTokenStyle(cfg)
out["enc_keys"] = np.array(sorted(m.state_dict().keys()))
out["enc_nparams"] = np.array(sum(p.numel() for p in m.parameters()))
np.savez_compressed(ROOT / "tests/golden/structure_builder.npz", **out)
print("structure builder outputs", len(d1), d1[-1].shape, "encoder keys", len(out["enc_keys"]), "params", int(out["enc_nparams"]),
      "bytes", (ROOT / "tests/golden/structure_builder.npz").stat().st_size)
