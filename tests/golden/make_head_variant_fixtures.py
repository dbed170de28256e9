"""Generates tests/golden/head_variants.npz by importing the REFERENCE's 2-view encoder (src/model/encoder/encoder_noposplat.py) with the two
Gaussian-parameter head types its forward runs besides 'dpt_gs' -- 'linear' (ReLU + Linear per token + pixel shuffle, :155-157) and 'dpt'
(the plain DPT regression head, :158-162) -- and `LinearPts3d` (heads/linear_head.py) through the head factory; tiny trunk, CPU, weights
regenerated on both sides from the parameter names (tests/helpers.deterministic_init_).  Also the constructor key sets of the multi-view
style encoder for the three head types (its forward raises for all but 'dpt_gs' in the reference, too).
    python tests/golden/make_head_variant_fixtures.py
"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
# the import recipe (stubs + package shims) lives in make_encoder_fixtures.py; reuse its module state without re-running its body
src = (ROOT / "tests/golden/make_encoder_fixtures.py").read_text()
ns = {"__name__": "fixture_prelude", "__file__": str(ROOT / "tests/golden/make_encoder_fixtures.py")}
exec(compile(src[:src.index("out = {}")], "make_encoder_fixtures.py (prelude)", "exec"), ns)
bc, ga, viz, ts, sb, enc_mod, ets, TINY = (ns[k] for k in ("bc", "ga", "viz", "ts", "sb", "enc_mod", "ets", "TINY"))
from tests.helpers import deterministic_init_
enp = importlib.import_module("src.model.encoder.encoder_noposplat")
heads = importlib.import_module("src.model.encoder.heads")
bc.croco_params["ViTLarge_BaseDecoder"] = dict(TINY)

out = {}
for ht in ("linear", "dpt"):
    cfg = enp.EncoderNoPoSplatCfg(
        name="noposplat", d_feature=128, num_monocular_samples=32,
        backbone=bc.BackboneCrocoCfg(name="croco", model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R", asymmetry_decoder=True,
                                     intrinsics_embed_loc="encoder", intrinsics_embed_degree=4, intrinsics_embed_type="token"),
        visualizer=viz.EncoderVisualizerEpipolarCfg(8, 256, False), gaussian_adapter=ga.GaussianAdapterCfg(0.5, 15.0, 1),
        apply_bounds_shim=True, opacity_mapping=enp.OpacityMappingCfg(0.0, 0.0, 1), gaussians_per_pixel=1, num_surfaces=1,
        gs_params_head_type=ht)
    model = enp.EncoderNoPoSplat(cfg).eval()
    deterministic_init_(model)
    g = torch.Generator().manual_seed(31)
    img = (torch.rand(1, 2, 3, 32, 48, generator=g) * 2 - 1).requires_grad_(True)
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(1, 2, 1, 1) + 0.01 * torch.rand(1, 2, 3, 3, generator=g)
    gs = model(dict(image=img, intrinsics=K), global_step=0)
    w = [torch.randn(t.shape, generator=g) for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities)]
    ((gs.means * w[0]).sum() + 1e4 * (gs.covariances * w[1]).sum() + (gs.harmonics * w[2]).sum() + (gs.opacities * w[3]).sum()).backward()
    out.update({f"{ht}_image": img.detach().numpy(), f"{ht}_intrinsics": K.numpy(), f"{ht}_means": gs.means.detach().numpy(),
                f"{ht}_cov": gs.covariances.detach().numpy(), f"{ht}_sh": gs.harmonics.detach().numpy(), f"{ht}_opac": gs.opacities.detach().numpy(),
                f"{ht}_gimage": img.grad.numpy(), f"{ht}_w0": w[0].numpy(), f"{ht}_w1": w[1].numpy(), f"{ht}_w2": w[2].numpy(), f"{ht}_w3": w[3].numpy(),
                f"{ht}_ghead": model.gaussian_param_head2[1].weight.grad.numpy() if ht == "linear" else model.gaussian_param_head2.dpt.head[4].weight.grad.numpy(),
                f"{ht}_keys": np.array(sorted(model.state_dict().keys())), f"{ht}_nparams": np.array(sum(p.numel() for p in model.parameters()))})
    print(ht, int(out[f"{ht}_nparams"]), gs.means.shape, float(gs.means.abs().mean()))
    if ht == "linear":      # LinearPts3d through the reference's factory, on this model's backbone and decoder tokens
        net = model.backbone
        net.depth_mode, net.conf_mode = ("exp", -float("inf"), float("inf")), None
        lp = heads.head_factory("linear", "pts3d", net, has_conf=False)
        deterministic_init_(lp)
        tok = torch.randn(2, 6, net.dec_embed_dim, generator=g)
        res = lp([tok], (32, 48))
        out.update(lp_tokens=tok.numpy(), lp_pts3d=res["pts3d"].detach().numpy(), lp_keys=np.array(sorted(lp.state_dict().keys())))
# constructor key sets of the multi-view style encoder for the three head types
for ht in ("linear", "dpt", "dpt_gs"):
    cfg = ets.EncoderNoPoSplatTokenStyleCfg(
        name="noposplat_multi_token_style", d_feature=128, num_monocular_samples=32,
        backbone=bc.BackboneCrocoCfg(name="croco_multi", model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R", asymmetry_decoder=True,
                                     intrinsics_embed_loc="encoder", intrinsics_embed_degree=4, intrinsics_embed_type="token"),
        token_stylizer=ts.TokenStylizerCfg("ViTLarge_BaseDecoder"), structure_builder=sb.StructureBuilderCfg("ViTLarge_BaseDecoder"),
        visualizer=viz.EncoderVisualizerEpipolarCfg(8, 256, False), gaussian_adapter=ga.GaussianAdapterCfg(0.5, 15.0, 0),
        apply_bounds_shim=True, opacity_mapping=enc_mod.OpacityMappingCfg(0.0, 0.0, 1), gaussians_per_pixel=1, num_surfaces=1,
        gs_params_head_type=ht, gs_sh_head_type="dpt", stylized=True)
    with torch.device("meta"):
        m = enc_mod.EncoderNoPoSplatMultiTokenStyle(cfg)
    out[f"style_{ht}_keys"] = np.array(sorted(m.state_dict().keys()))
    out[f"style_{ht}_nparams"] = np.array(sum(p.numel() for p in m.parameters()))
np.savez_compressed(ROOT / "tests/golden/head_variants.npz", **out)
print("wrote", len(out))
