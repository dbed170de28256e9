"""Generates tests/golden/encoder_tiny.npz by importing the REFERENCE's full style-token encoder
(src/model/encoder/encoder_noposplat_multi_token_style.py) in the build container, following
SURVEY.md Appendix B (stubs for jaxtyping / xformers / e3nn, package shims).  The transformer trunk is
shrunk through the reference's own `croco_params` table (1 ViT-L encoder block, 12 two-head decoder
blocks) so the run is CPU-sized; every code path of forward() is exercised.  Weights are not stored:
both sides fill them with tests/helpers.deterministic_init_ (keyed by state_dict name).
    python tests/golden/make_encoder_fixtures.py
"""
import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, REF); sys.path.insert(0, str(ROOT))
from tests.helpers import deterministic_init_

jt = types.ModuleType("jaxtyping")
class _Sub:
    def __class_getitem__(cls, item): return cls
for n in ("Float", "Int64", "Bool", "UInt8", "Shaped", "Int"):
    setattr(jt, n, type(n, (_Sub,), {}))
sys.modules["jaxtyping"] = jt
xf = types.ModuleType("xformers"); xo = types.ModuleType("xformers.ops")
def mea(q, k, v, scale=None, p=0.0):
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    a = (q.permute(0, 2, 1, 3) @ k.permute(0, 2, 3, 1)) * scale
    return (a.softmax(-1) @ v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
xo.memory_efficient_attention = mea; xf.ops = xo
sys.modules["xformers"] = xf; sys.modules["xformers.ops"] = xo
e3 = types.ModuleType("e3nn"); o3 = types.ModuleType("e3nn.o3")
o3.matrix_to_angles = o3.wigner_D = lambda *a, **k: None; e3.o3 = o3
sys.modules["e3nn"] = e3; sys.modules["e3nn.o3"] = o3

for name in ("src", "src.model", "src.model.encoder", "src.model.encoder.backbone", "src.model.encoder.backbone.croco",
             "src.model.encoder.common", "src.model.encoder.token_stylizer", "src.model.encoder.visualization",
             "src.model.decoder", "src.geometry", "src.misc", "src.dataset", "src.dataset.shims"):
    m = types.ModuleType(name); m.__path__ = [REF + "/" + name.replace(".", "/")]; sys.modules[name] = m
bb = sys.modules["src.model.encoder.backbone"]
bb.Backbone = importlib.import_module("src.model.encoder.backbone.backbone").Backbone
bc = importlib.import_module("src.model.encoder.backbone.backbone_croco")
bm = importlib.import_module("src.model.encoder.backbone.backbone_croco_multiview")
bb.BackboneCfg = bc.BackboneCrocoCfg
bb.get_backbone = lambda cfg, d_in=3: {"croco": bc.AsymmetricCroCo, "croco_multi": bm.AsymmetricCroCoMulti}[cfg.name](cfg, d_in)
sys.modules["src.dataset"].DatasetCfg = None
ts = importlib.import_module("src.model.encoder.token_stylizer.token_stylizer")
sb = importlib.import_module("src.model.encoder.token_stylizer.structure_builder")

TINY = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512))
bm.croco_params["ViTLarge_BaseDecoder"] = dict(TINY)
ts.croco_params["ViTLarge_BaseDecoder"] = dict(TINY)

enc_mod = importlib.import_module("src.model.encoder.encoder_noposplat_multi_token_style")
ets = importlib.import_module("src.model.encoder.encoder_noposplat_token_style")
ga = importlib.import_module("src.model.encoder.common.gaussian_adapter")
viz = importlib.import_module("src.model.encoder.visualization.encoder_visualizer_epipolar_cfg")

out = {}
for tag, sh_degree in (("sh0", 0), ("sh1", 1)):
    cfg = ets.EncoderNoPoSplatTokenStyleCfg(
        name="noposplat_multi_token_style", d_feature=128, num_monocular_samples=32,
        backbone=bc.BackboneCrocoCfg(name="croco_multi", model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R",
                                     asymmetry_decoder=True, intrinsics_embed_loc="encoder", intrinsics_embed_degree=4,
                                     intrinsics_embed_type="token"),
        token_stylizer=ts.TokenStylizerCfg("ViTLarge_BaseDecoder"), structure_builder=sb.StructureBuilderCfg("ViTLarge_BaseDecoder"),
        visualizer=viz.EncoderVisualizerEpipolarCfg(8, 256, False), gaussian_adapter=ga.GaussianAdapterCfg(0.5, 15.0, sh_degree),
        apply_bounds_shim=True, opacity_mapping=enc_mod.OpacityMappingCfg(0.0, 0.0, 1), gaussians_per_pixel=1, num_surfaces=1,
        gs_params_head_type="dpt_gs", gs_sh_head_type="dpt", stylized=True)
    torch.manual_seed(0)
    model = enc_mod.EncoderNoPoSplatMultiTokenStyle(cfg).eval()
    deterministic_init_(model)
    g = torch.Generator().manual_seed(11)
    b, v, H, W = 1, 3, 32, 48
    img = (torch.rand(b, v, 3, H, W, generator=g) * 2 - 1).requires_grad_(True)
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(b, v, 1, 1) + 0.01 * torch.rand(b, v, 3, 3, generator=g)
    style = (torch.rand(b, 3, 32, 32, generator=g) * 2 - 1)
    ctx = dict(image=img, intrinsics=K)
    dump = {}
    gs = model(ctx, dict(image=style), global_step=0, visualization_dump=dump)
    w = [torch.randn(t.shape, generator=g) for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities)]
    loss = (gs.means * w[0]).sum() + 1e4 * (gs.covariances * w[1]).sum() + (gs.harmonics * w[2]).sum() + (gs.opacities * w[3]).sum()
    loss.backward()
    out.update({f"{tag}_image": img.detach().numpy(), f"{tag}_intrinsics": K.numpy(), f"{tag}_style": style.numpy(),
                f"{tag}_means": gs.means.detach().numpy(), f"{tag}_cov": gs.covariances.detach().numpy(),
                f"{tag}_sh": gs.harmonics.detach().numpy(), f"{tag}_opac": gs.opacities.detach().numpy(),
                f"{tag}_gimage": img.grad.numpy(), f"{tag}_dump_scales": dump["scales"].detach().numpy(),
                f"{tag}_w0": w[0].numpy(), f"{tag}_w1": w[1].numpy(), f"{tag}_w2": w[2].numpy(), f"{tag}_w3": w[3].numpy(),
                f"{tag}_nparams": np.array(sum(p.numel() for p in model.parameters())),
                f"{tag}_keys": np.array(sorted(model.state_dict().keys()))})
    # one parameter gradient deep in the stylizer (reaches it only through the SH head)
    out[f"{tag}_g_sty_projk"] = model.token_stylizer.dec_blocks[3].cross_attn.projk.weight.grad.numpy()
    print(tag, "params", int(out[f"{tag}_nparams"]), "means", gs.means.shape, float(gs.means.abs().mean()))
# ---- non-style variant: EncoderNoPoSplat on the 2-view AsymmetricCroCo backbone (config 5's encoder; the reference's
# EncoderNoPoSplatMulti no longer runs against its own backbone, which now returns five values), sh_degree 1 ----------
enp = importlib.import_module("src.model.encoder.encoder_noposplat")
bc.croco_params["ViTLarge_BaseDecoder"] = dict(TINY)
cfg = enp.EncoderNoPoSplatCfg(
    name="noposplat", d_feature=128, num_monocular_samples=32,
    backbone=bc.BackboneCrocoCfg(name="croco", model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R",
                                 asymmetry_decoder=True, intrinsics_embed_loc="encoder", intrinsics_embed_degree=4,
                                 intrinsics_embed_type="token"),
    visualizer=viz.EncoderVisualizerEpipolarCfg(8, 256, False), gaussian_adapter=ga.GaussianAdapterCfg(0.5, 15.0, 1),
    apply_bounds_shim=True, opacity_mapping=enp.OpacityMappingCfg(0.0, 0.0, 1), gaussians_per_pixel=1, num_surfaces=1,
    gs_params_head_type="dpt_gs")
model = enp.EncoderNoPoSplat(cfg).eval()
deterministic_init_(model)
g = torch.Generator().manual_seed(23)
img = (torch.rand(1, 2, 3, 48, 32, generator=g) * 2 - 1).requires_grad_(True)
K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(1, 2, 1, 1) + 0.01 * torch.rand(1, 2, 3, 3, generator=g)
gs = model(dict(image=img, intrinsics=K), global_step=0)
w = [torch.randn(t.shape, generator=g) for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities)]
((gs.means * w[0]).sum() + 1e4 * (gs.covariances * w[1]).sum() + (gs.harmonics * w[2]).sum() + (gs.opacities * w[3]).sum()).backward()
out.update(np_image=img.detach().numpy(), np_intrinsics=K.numpy(), np_means=gs.means.detach().numpy(), np_cov=gs.covariances.detach().numpy(),
           np_sh=gs.harmonics.detach().numpy(), np_opac=gs.opacities.detach().numpy(), np_gimage=img.grad.numpy(),
           np_w0=w[0].numpy(), np_w1=w[1].numpy(), np_w2=w[2].numpy(), np_w3=w[3].numpy(),
           np_keys=np.array(sorted(model.state_dict().keys())), np_nparams=np.array(sum(p.numel() for p in model.parameters())))
print("noposplat params", int(out["np_nparams"]), gs.means.shape)
np.savez_compressed(ROOT / "tests/golden/encoder_tiny.npz", **out)
print("wrote", len(out))
