"""Generates tests/golden/losses_ref.npz and tests/golden/checkpoint_ref.npz from the REFERENCE's own modules:
  * `LossStyle`          src/loss/loss_style.py:25-79
  * `IdentityLoss`       src/loss/loss_identity.py:13-52
  * `VGGEncoder`         src/test/vgg_model.py:79-98
  * `checkpoint_filter_fn`, `resample_patch_embed`   src/misc/weight_modify.py:13-81,144-197
torchvision is not installed here: the stub below supplies `torchvision.models.vgg19` (the stock `features` layout, weights
then overwritten by tests/helpers.deterministic_vgg_, keyed by the torchvision layer index so both sides agree) and
`torchvision.transforms.Normalize` ((x - mean) / std per channel, its documented definition).  The losses run in float64.
    python tests/golden/make_loss_fixtures.py
"""
import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch
from torch import nn

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden.ref_stubs import REF, install
from tests.helpers import deterministic_init_, deterministic_vgg_

install()
# ---- torchvision stub -------------------------------------------------------------------------------------------
tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models"); tvt = types.ModuleType("torchvision.transforms")
_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg19(pretrained=False, **kw):
    layers, c = [], 3
    for v in _CFG_E:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]; c = v
    return types.SimpleNamespace(features=nn.Sequential(*layers))


class Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.mean, self.std = list(mean), list(std)

    def forward(self, x):
        m = torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        s = torch.tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - m) / s


tvm.vgg19 = vgg19; tvt.Normalize = Normalize; tv.models = tvm; tv.transforms = tvt
sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt})
# ---- package shims for the loss modules' imports -------------------------------------------------------------------
sys.modules["src.dataset"].DatasetCfg = None
sys.modules["src.dataset.types"] = types.ModuleType("src.dataset.types"); sys.modules["src.dataset.types"].BatchedExample = dict
mt = importlib.import_module("src.model.types")
sys.modules["diff_gaussian_rasterization"] = types.ModuleType("diff_gaussian_rasterization")
dec = types.ModuleType("src.model.decoder.decoder")


class DecoderOutput:
    def __init__(self, color, depth=None):
        self.color, self.depth = color, depth


dec.DecoderOutput = DecoderOutput
sys.modules["src.model.decoder.decoder"] = dec
loss_base = importlib.import_module("src.loss.loss")
sys.modules["src.loss"].Loss = loss_base.Loss
ls = importlib.import_module("src.loss.loss_style")
li = importlib.import_module("src.loss.loss_identity")

g = torch.Generator().manual_seed(41)
b, v, H = 2, 2, 64
pred = torch.rand(b, v, 3, H, H, generator=g, dtype=torch.float64)
tgt = torch.rand(b, v, 3, H, H, generator=g, dtype=torch.float64)
style = torch.rand(b, 3, 80, 80, generator=g, dtype=torch.float64)
batch = {"target": {"image": tgt}, "style": {"image": style}}
out = dict(pred=pred.numpy().astype(np.float32), target=tgt.numpy().astype(np.float32), style=style.numpy().astype(np.float32))
pred32 = torch.tensor(out["pred"]).double(); tgt32 = torch.tensor(out["target"]).double(); sty32 = torch.tensor(out["style"]).double()
batch = {"target": {"image": tgt32}, "style": {"image": sty32}}

style_loss = ls.LossStyle(ls.LossStyleCfgWrapper(ls.LossStyleCfg(style_weight=10.0)))
deterministic_vgg_(style_loss.vgg)
style_loss = style_loss.double()
p = pred32.clone().requires_grad_(True)
l = style_loss(DecoderOutput(p), batch, None, 0)
l.backward()
out.update(style_value=np.array(float(l)), style_grad=p.grad.numpy().astype(np.float32), style_weight=np.array(10.0))
feats = style_loss.vgg(Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])(pred32.reshape(b * v, 3, H, H)))
for k, f in enumerate(feats):
    out[f"vgg_h{k + 1}_mean"] = f.mean(dim=(2, 3)).numpy()           # per-image, per-channel feature means (small)
    out[f"vgg_h{k + 1}_absmax"] = np.array(float(f.abs().max()))

ident = li.IdentityLoss(70, 1)
deterministic_vgg_(ident.vgg)
ident = ident.double()
p = pred32.clone().requires_grad_(True)
l = ident(DecoderOutput(p), batch, None, 0)
l.backward()
out.update(identity_value=np.array(float(l)), identity_grad=p.grad.numpy().astype(np.float32))
np.savez_compressed(ROOT / "tests/golden/losses_ref.npz", **out)
print("LossStyle", float(out["style_value"]), "IdentityLoss", float(out["identity_value"]),
      "bytes", (ROOT / "tests/golden/losses_ref.npz").stat().st_size)

# ---- checkpoint_filter_fn on a synthetic MASt3R 'model' dict + resample_patch_embed --------------------------------
wm = importlib.import_module("src.misc.weight_modify")
sys.path.insert(0, str(ROOT))
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
TINY = dict(enc_depth=1, dec_depth=12, enc_embed_dim=128, dec_embed_dim=64, enc_num_heads=2, dec_num_heads=1, pos_embed="RoPE100",
            img_size=(512, 512))
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=TINY)   # only its SHAPES are read by the filter
src = {}
gg = torch.Generator().manual_seed(9)
for k, t in enc.backbone.state_dict().items():
    if k.startswith(("dec_blocks2", "intrinsic_encoder")):
        continue
    src[k] = torch.randn(t.shape, generator=gg)
src["patch_embed.proj.weight"] = torch.randn(128, 3, 8, 8, generator=gg)         # 8 x 8 patches -> resampled to 16 x 16
src["decoder_embed.weight"] = torch.randn(64, 128, generator=gg)
for h in ("downstream_head1", "downstream_head2"):
    for k, t in getattr(enc, h).state_dict().items():
        src[f"{h}.{k}"] = torch.randn((4, *t.shape[1:]) if k.startswith("dpt.head.4.") else t.shape, generator=gg)
res = wm.checkpoint_filter_fn({k: v.clone() for k, v in src.items()}, enc)
ck = {"src:" + k: v.numpy() for k, v in src.items() if k in ("patch_embed.proj.weight", "decoder_embed.weight",
                                                               "downstream_head1.dpt.head.4.weight", "downstream_head2.dpt.head.4.bias",
                                                               "enc_blocks.0.attn.qkv.weight")}
ck["seed"] = np.array(9)
ck["out_keys"] = np.array(sorted(res.keys()))
for k in ("backbone.patch_embed.proj.weight", "backbone.decoder_embed.weight", "downstream_head1.dpt.head.4.weight",
          "downstream_head2.dpt.head.4.bias", "backbone.enc_blocks.0.attn.qkv.weight"):
    ck["out:" + k] = res[k].numpy()
# resample_patch_embed alone, non-square target
pe = torch.randn(6, 3, 8, 8, generator=gg)
ck["rpe_in"] = pe.numpy(); ck["rpe_out_12x20"] = wm.resample_patch_embed(pe, (12, 20)).numpy()
# adapt_input_conv / adapt_linear branches
ck["aic_in"] = torch.randn(5, 3, 4, 4, generator=gg).numpy()
ck["aic_out_1"] = wm.adapt_input_conv(1, torch.tensor(ck["aic_in"])).numpy()
ck["aic_out_7"] = wm.adapt_input_conv(7, torch.tensor(ck["aic_in"])).numpy()
ck["al_in"] = torch.randn(4, 162, generator=gg).numpy()
ck["al_out"] = wm.adapt_linear(torch.tensor(ck["al_in"])).numpy()
np.savez_compressed(ROOT / "tests/golden/checkpoint_ref.npz", **ck)
print("checkpoint fixture:", len(res), "keys; patch_embed", tuple(res["backbone.patch_embed.proj.weight"].shape),
      "bytes", (ROOT / "tests/golden/checkpoint_ref.npz").stat().st_size)
