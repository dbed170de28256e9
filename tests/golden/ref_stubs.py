"""Import recipe for the REFERENCE's Python modules in the build container: stub modules for
the dependencies the image lacks (jaxtyping, xformers, e3nn, torchvision, timm, lpips, einops is present) and package shims
so that the heavy `__init__` files of `src.*` never run.  Used only by the fixture generators in this directory; nothing
here travels to the GPU box as a dependency of the tests (they read the generated .npz files only).
    from tests.golden.ref_stubs import install;  mods = install()
"""
import importlib
import sys
import types

REF = "/root/reference"


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    jt = types.ModuleType("jaxtyping")

    class _Sub:
        def __class_getitem__(cls, item):
            return cls
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
                 "src.model.decoder", "src.geometry", "src.misc", "src.dataset", "src.dataset.shims", "src.loss", "src.test"):
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
    return types.SimpleNamespace(bc=bc, bm=bm, ts=ts, sb=sb)


def style_encoder_cfg(mods, sh_degree=0, stylized=True):
    """EncoderNoPoSplatTokenStyleCfg with the values of config/model/encoder/noposplat_token_style.yaml (Appendix B.4)."""
    enc_mod = importlib.import_module("src.model.encoder.encoder_noposplat_multi_token_style")
    ets = importlib.import_module("src.model.encoder.encoder_noposplat_token_style")
    ga = importlib.import_module("src.model.encoder.common.gaussian_adapter")
    viz = importlib.import_module("src.model.encoder.visualization.encoder_visualizer_epipolar_cfg")
    bc, ts, sb = mods.bc, mods.ts, mods.sb
    cfg = ets.EncoderNoPoSplatTokenStyleCfg(
        name="noposplat_multi_token_style", d_feature=128, num_monocular_samples=32,
        backbone=bc.BackboneCrocoCfg(name="croco_multi", model="ViTLarge_BaseDecoder", patch_embed_cls="PatchEmbedDust3R",
                                     asymmetry_decoder=True, intrinsics_embed_loc="encoder", intrinsics_embed_degree=4,
                                     intrinsics_embed_type="token"),
        token_stylizer=ts.TokenStylizerCfg("ViTLarge_BaseDecoder"), structure_builder=sb.StructureBuilderCfg("ViTLarge_BaseDecoder"),
        visualizer=viz.EncoderVisualizerEpipolarCfg(8, 256, False), gaussian_adapter=ga.GaussianAdapterCfg(0.5, 15.0, sh_degree),
        apply_bounds_shim=True, opacity_mapping=enc_mod.OpacityMappingCfg(0.0, 0.0, 1), gaussians_per_pixel=1, num_surfaces=1,
        gs_params_head_type="dpt_gs", gs_sh_head_type="dpt", stylized=stylized)
    return enc_mod, cfg
