"""Encoder against golden vectors captured from the reference's own
EncoderNoPoSplatMultiTokenStyle (tests/golden/make_encoder_fixtures.py)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import deterministic_init_

G = np.load(Path(__file__).resolve().parent / "golden" / "encoder_tiny.npz")
TINY = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512))


def _build(sh_degree):
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg, GaussianAdapterCfg
    cfg = EncoderNoPoSplatTokenStyleCfg(gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, sh_degree))
    return EncoderNoPoSplatMultiTokenStyle(cfg, trunk_params=TINY).eval()


@pytest.mark.parametrize("tag,sh_degree", [("sh0", 0), ("sh1", 1)])
def test_state_dict_keys_and_parameter_count_match_reference(tag, sh_degree):
    """drop-in checkpoint compatibility: identical key set (incl. the duplicated scratch.layer_rn aliases)"""
    m = _build(sh_degree)
    assert sorted(m.state_dict().keys()) == list(G[f"{tag}_keys"])
    assert sum(p.numel() for p in m.parameters()) == int(G[f"{tag}_nparams"])


def test_full_size_key_layout():
    """the real ViT-L / ViT-B layout of SURVEY 8b without allocating it"""
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    with torch.device("meta"):
        m = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg())
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 1_049_635_033
    assert sd["backbone.patch_embed.proj.weight"].shape == (1024, 3, 16, 16)
    assert sd["backbone.enc_blocks.23.attn.qkv.weight"].shape == (3072, 1024)
    assert sd["backbone.dec_blocks2.11.cross_attn.projk.weight"].shape == (768, 768)
    assert sd["backbone.intrinsic_encoder.weight"].shape == (1024, 9)
    assert sd["gaussian_param_head.dpt.input_merger.0.weight"].shape == (256, 3, 7, 7)
    assert sd["gaussian_param_head.dpt.head.4.weight"].shape[0] == 8
    assert "downstream_head1.dpt.scratch.layer_rn.0.weight" in sd and "downstream_head1.dpt.scratch.layer1_rn.weight" in sd
    assert "token_stylizer.dec_blocks.0.norm_y.weight" in sd and not any(k.startswith("token_stylizer.dec_blocks2") for k in sd)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,sh_degree", [("sh0", 0), ("sh1", 1)])
def test_encoder_forward_backward_matches_reference(tag, sh_degree):
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(_build(sh_degree)).to(dev)
    T = lambda k: torch.tensor(G[f"{tag}_{k}"], device=dev)
    img = T("image").requires_grad_(True)
    dump = {}
    gs = m(dict(image=img, intrinsics=T("intrinsics")), dict(image=T("style")), global_step=0, visualization_dump=dump)
    assert gs.means.shape == (1, 3 * 32 * 48, 3) and gs.harmonics.shape[-1] == (sh_degree + 1) ** 2
    assert_close_rel(gs.means.detach().cpu().numpy(), G[f"{tag}_means"], 1e-4, "means")
    assert_close_rel(gs.covariances.detach().cpu().numpy(), G[f"{tag}_cov"], 1e-4, "covariances")
    assert_close_rel(gs.harmonics.detach().cpu().numpy(), G[f"{tag}_sh"], 1e-4, "harmonics")
    assert_close_rel(gs.opacities.detach().cpu().numpy(), G[f"{tag}_opac"], 1e-4, "opacities")
    assert_close_rel(dump["scales"].detach().cpu().numpy(), G[f"{tag}_dump_scales"], 1e-4, "visualization_dump scales")
    loss = (gs.means * T("w0")).sum() + 1e4 * (gs.covariances * T("w1")).sum() + (gs.harmonics * T("w2")).sum() + \
        (gs.opacities * T("w3")).sum()
    loss.backward()
    # deep gradients (through the DPT heads and 12 decoder blocks).  TYPICAL distances are 3.3e-5 (d image) and 7.5e-6 (d projk)
    # (profiles/r05_parity_encoder_observed.txt), and round 5 first set the bars at 1e-4 -- the 3 x stress run of the whole
    # suite (tools/suite_stress.sh) then saw d projk at 1.05e-3 ONCE: the tiny fixture's head ReLUs / opacity clamps flip single pixels from run
    # to run (split-K atomics order the fp32 sums differently) and one flipped pixel moves this gradient by that much.  So the bar stays at the
    # 2e-3 of rounds 1 - 4, for that reason and not for "library noise" (no library kernel is left on the path)
    assert_close_rel(img.grad.cpu().numpy(), G[f"{tag}_gimage"], 2e-3, "d image")
    got = m.token_stylizer.dec_blocks[3].cross_attn.projk.weight.grad
    assert_close_rel(got.cpu().numpy(), G[f"{tag}_g_sty_projk"], 2e-3, "d token_stylizer projk")


def _build_noposplat():
    from styl3r_amd.encoder import EncoderNoPoSplatCfg, EncoderNoPoSplatMulti, GaussianAdapterCfg
    return EncoderNoPoSplatMulti(EncoderNoPoSplatCfg(gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, 1)), trunk_params=TINY).eval()


@pytest.mark.gpu
def test_encoder_forward_in_bf16x3_mode_is_an_order_looser_than_the_default_modes(monkeypatch):
    """VIT_LINEAR_MODE=bf16x3 (three partial products per GEMM launch) is opt-in and is NOT offered as meeting north_star's 1e-4:
    on this fixture its covariances land between 4.8e-5 and 1.07e-4 from one process to the next (nine runs on two trees,
    profiles/r06_run_to_run_spread.txt: the split-K summation order moves an activation across a bf16 rounding boundary and the
    dropped middle x middle product moves with it), against 6e-6 .. 7e-6 in the bf16x6 and f16x3 modes that the parity tests and the
    bench run in.  The bars here say what the mode delivers: 1e-4 for means / harmonics / opacities (observed <= 7e-5), 2e-4 for the
    covariances.  Deep in the network it also gives up gradient accuracy (5e-3 on the stylizer's projk weight against 1e-4 .. 6e-4)."""
    from styl3r_amd import vit_ops
    from tests.gpu_utils import assert_close_rel
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "bf16x3")
    dev, tag = "cuda:0", "sh1"
    m = deterministic_init_(_build(1)).to(dev)
    T = lambda k: torch.tensor(G[f"{tag}_{k}"], device=dev)
    with torch.no_grad():
        gs = m(dict(image=T("image"), intrinsics=T("intrinsics")), dict(image=T("style")), global_step=0)
    assert vit_ops.load().vit_x6_products() == 3
    assert_close_rel(gs.means.cpu().numpy(), G[f"{tag}_means"], 1e-4, "means")
    assert_close_rel(gs.covariances.cpu().numpy(), G[f"{tag}_cov"], 2e-4, "covariances")
    assert_close_rel(gs.harmonics.cpu().numpy(), G[f"{tag}_sh"], 1e-4, "harmonics")
    assert_close_rel(gs.opacities.cpu().numpy(), G[f"{tag}_opac"], 1e-4, "opacities")
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "bf16x6")
    assert vit_ops._x6() and vit_ops.load().vit_x6_products() == 6


def test_noposplat_variant_keys_match_reference():
    m = _build_noposplat()
    assert sorted(m.state_dict().keys()) == list(G["np_keys"])
    assert sum(p.numel() for p in m.parameters()) == int(G["np_nparams"])
    from styl3r_amd.encoder import ENCODERS
    assert ENCODERS["noposplat"] is ENCODERS["noposplat_multi"]


@pytest.mark.gpu
def test_noposplat_variant_matches_reference():
    """the non-style encoder of BASELINE config 5 (`forward(context, global_step, visualization_dump)`, 83-channel gs head)"""
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(_build_noposplat()).to(dev)
    T = lambda k: torch.tensor(G[f"np_{k}"], device=dev)
    img = T("image").requires_grad_(True)
    gs = m(dict(image=img, intrinsics=T("intrinsics")), global_step=0)
    assert gs.harmonics.shape == (1, 2 * 48 * 32, 3, 4)
    for name, t in (("means", gs.means), ("cov", gs.covariances), ("sh", gs.harmonics), ("opac", gs.opacities)):
        assert_close_rel(t.detach().cpu().numpy(), G[f"np_{name}"], 1e-4, name)
    ((gs.means * T("w0")).sum() + 1e4 * (gs.covariances * T("w1")).sum() + (gs.harmonics * T("w2")).sum() +
     (gs.opacities * T("w3")).sum()).backward()
    assert_close_rel(img.grad.cpu().numpy(), G["np_gimage"], 2e-3, "d image")      # typical 2.0e-4 (r05); single-pixel flips as above


@pytest.mark.gpu
def test_hipgraph_replay_of_the_encoder_matches_eager():
    """styl3r_amd.graphs.GraphedEncoder: the HIP kernels are captured into a hipGraph like torch's own; replay with new
    inputs equals the eager forward"""
    from styl3r_amd.graphs import GraphedEncoder
    dev = "cuda:0"
    m = deterministic_init_(_build(0)).to(dev)
    T = lambda k: torch.tensor(G[f"sh0_{k}"], device=dev)
    ctx, style = dict(image=T("image"), intrinsics=T("intrinsics")), dict(image=T("style"))
    genc = GraphedEncoder(m, ctx, style)
    ctx2 = dict(image=(ctx["image"] * 0.7 + 0.1).contiguous(), intrinsics=ctx["intrinsics"])
    for _ in range(3):                  # (several replays: a non-idempotent node -- r03: hipMemsetAsync's graph node -- only shows from the second on)
        got = genc(ctx2, style)
    with torch.no_grad():
        want = m(ctx2, style, 0)
    from tests.gpu_utils import assert_close_rel
    with torch.no_grad():
        old = m(ctx, style, 0)
    for name in ("means", "covariances", "harmonics", "opacities"):
        a, b = getattr(got, name).cpu().numpy(), getattr(want, name).cpu().numpy()
        assert_close_rel(a, b, 2e-5, name)                      # (library convolutions may pick another algorithm: fp32 noise)
        assert abs(a - getattr(old, name).cpu().numpy()).max() > 1e-2 * abs(b).max()   # and it really used the new inputs


@pytest.mark.gpu
def test_per_stream_hipgraph_replay_of_the_serving_forward_matches_eager():
    """styl3r_amd.graphs.StreamGraphedEncoder: one hipGraph per stream segment (backbone encoder | style encoder | stylizer decoder | each layer of
    decoder 1 / decoder 2 | five head calls | adapter) replayed on the serving streams with the eager path's fork / join waits; new inputs, repeated
    replays, results equal the eager forward"""
    from styl3r_amd.graphs import StreamGraphedEncoder
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(_build(0)).to(dev)
    T = lambda k: torch.tensor(G[f"sh0_{k}"], device=dev)
    ctx, style = dict(image=T("image"), intrinsics=T("intrinsics")), dict(image=T("style"))
    genc = StreamGraphedEncoder(m, ctx, style)
    assert len(genc.g_d1) == len(m.backbone.dec_blocks) and len(genc.g_heads) == 5
    ctx2 = dict(image=(ctx["image"] * 0.7 + 0.1).contiguous(), intrinsics=ctx["intrinsics"])
    with torch.no_grad():
        want, old = m(ctx2, style, 0), m(ctx, style, 0)
    for rep in range(3):
        got = genc(ctx2, style)
        torch.cuda.synchronize()
        for name in ("means", "covariances", "harmonics", "opacities"):
            a, b = getattr(got, name).cpu().numpy(), getattr(want, name).cpu().numpy()
            assert_close_rel(a, b, 1e-4, name)                      # (split-contraction atomics: the summation order is not fixed)
            assert abs(a - getattr(old, name).cpu().numpy()).max() > 1e-2 * abs(b).max()   # and it really used the new inputs
    got = genc(ctx, style)                                            # back to the first inputs
    torch.cuda.synchronize()
    assert_close_rel(got.means.cpu().numpy(), old.means.cpu().numpy(), 1e-4, "means, first inputs again")
    # two scenes per call: the per-view-group slices of the images are copies there and must be taken inside the captured pieces
    cat = lambda a, b_: torch.cat((a, b_), dim=0).contiguous()
    ctx_b2 = dict(image=cat(ctx["image"], ctx2["image"]), intrinsics=cat(ctx["intrinsics"], ctx["intrinsics"]))
    style_b2 = dict(image=cat(style["image"], style["image"] * 0.5))
    genc2 = StreamGraphedEncoder(m, ctx_b2, style_b2)
    ctx_b2n = dict(image=cat(ctx2["image"], ctx["image"] * 0.9), intrinsics=ctx_b2["intrinsics"])
    with torch.no_grad():
        want2 = m(ctx_b2n, style_b2, 0)
    for rep in range(2):
        got2 = genc2(ctx_b2n, style_b2)
        torch.cuda.synchronize()
        for name in ("means", "covariances", "harmonics", "opacities"):
            assert_close_rel(getattr(got2, name).cpu().numpy(), getattr(want2, name).cpu().numpy(), 1e-4, "b = 2: " + name)


@pytest.mark.gpu
def test_bf16x6_and_f32_paths_agree_through_the_whole_encoder():
    """Large enough images for the bf16x6 convolution kernels (Conv2dX6, >= 200 output tiles) and split-K Linear paths to
    run inside the real graph: Gaussians and input gradients must match the exact-f32 MFMA Linear + MIOpen convolution
    path (VIT_LINEAR_MODE=f32) to fp32 round-off"""
    from styl3r_amd import vit_ops
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(_build(0)).to(dev)
    g = torch.Generator(dev).manual_seed(5)
    H, W = 128, 160
    img = (torch.rand(1, 2, 3, H, W, device=dev, generator=g) * 2 - 1)
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]], device=dev).expand(1, 2, 3, 3).contiguous()
    style = torch.rand(1, 3, H, W, device=dev, generator=g) * 2 - 1
    res = {}
    old = vit_ops.LINEAR_MODE
    try:
        for mode in ("f32", "bf16x6"):
            vit_ops.LINEAR_MODE = mode
            x = img.clone().requires_grad_(True)
            gs = m(dict(image=x, intrinsics=K), dict(image=style), global_step=0)
            loss = (gs.means * 0.01).sum() + gs.covariances.sum() * 1e3 + gs.harmonics.sum() * 0.1 + gs.opacities.sum() * 0.1
            loss.backward()
            res[mode] = [t.detach().cpu().numpy() for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities, x.grad)]
            if mode == "f32":
                before = dict(vit_ops.CALLS)
    finally:
        vit_ops.LINEAR_MODE = old
    assert vit_ops.CALLS["conv_x6_fwd"] > before["conv_x6_fwd"] and vit_ops.CALLS["conv_x6_dx"] > before["conv_x6_dx"]   # really taken
    for name, a, b in zip(("means", "covariances", "harmonics", "opacities", "d image"), res["bf16x6"], res["f32"]):
        assert_close_rel(a, b, 2e-3 if name == "d image" else 1e-4, name)   # (d image typically 3.3e-6, r05; single-pixel flips as above)


@pytest.mark.gpu
def test_head_streams_option_gives_the_same_gaussians():
    """serving option `head_streams`: the five head calls on their own HIP streams (fork / join on the caller's stream)
    return the same Gaussians as the in-order launch, repeatedly (no allocator hazards across streams)"""
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(_build(0)).to(dev)
    T = lambda k: torch.tensor(G[f"sh0_{k}"], device=dev)
    ctx, style = dict(image=T("image"), intrinsics=T("intrinsics")), dict(image=T("style"))
    with torch.no_grad():
        ref = m(ctx, style, 0)
        m.head_streams = True
        for _ in range(3):
            got = m(ctx, style, 0)
            torch.cuda.synchronize()
            # (max-norm: the small convolutions go through split-K atomics / the library, whose summation order is not fixed)
            for name, a, b in (("means", got.means, ref.means), ("cov", got.covariances, ref.covariances),
                               ("sh", got.harmonics, ref.harmonics), ("opacity", got.opacities, ref.opacities)):
                assert_close_rel(a.cpu().numpy(), b.cpu().numpy(), 1e-4, name)
        assert len(m._head_stream_pool) == 5
    m.head_streams = False


@pytest.mark.gpu
def test_f16x3_maxima_words_across_streams_and_graph_replays(monkeypatch):
    """the f16x3 |max| word arena is keyed by (device, stream, capture state) and every captured graph zero-fills its own
    arena on replay.  (i) f16x3 with `head_streams` (words produced and consumed on side streams) equals the single-stream f16x3 result,
    repeatedly; (ii) the per-stream-segment graphs in f16x3 follow new inputs over several replays, including inputs 8x SMALLER than the
    captured ones right after larger ones -- a word that is not re-zeroed keeps the larger maximum and the fp16 pieces of the small
    tensors lose up to three bits."""
    from styl3r_amd import vit_ops
    from styl3r_amd.graphs import StreamGraphedEncoder
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "f16x3")
    vit_ops._x6()
    try:
        m = deterministic_init_(_build(0)).to(dev)
        T = lambda k: torch.tensor(G[f"sh0_{k}"], device=dev)
        ctx, style = dict(image=T("image"), intrinsics=T("intrinsics")), dict(image=T("style"))
        names = ("means", "covariances", "harmonics", "opacities")
        with torch.no_grad():
            ref = m(ctx, style, 0)
            m.head_streams = True
            for _ in range(3):
                got = m(ctx, style, 0)
                torch.cuda.synchronize()
                for n in names:
                    assert_close_rel(getattr(got, n).cpu().numpy(), getattr(ref, n).cpu().numpy(), 1e-4, "head_streams f16x3: " + n)
            m.head_streams = False
        genc = StreamGraphedEncoder(m, ctx, style)
        small = dict(image=(ctx["image"] * 0.125).contiguous(), intrinsics=ctx["intrinsics"])
        big = dict(image=(ctx["image"] * 0.9 + 0.05).contiguous(), intrinsics=ctx["intrinsics"])
        for inp in (big, small, big, small):
            with torch.no_grad():
                want = m(inp, style, 0)
            got = genc(inp, style)
            torch.cuda.synchronize()
            for n in names:
                assert_close_rel(getattr(got, n).cpu().numpy(), getattr(want, n).cpu().numpy(), 1e-4, "graph replay f16x3: " + n)
    finally:
        vit_ops.LINEAR_MODE = "bf16x6"
        vit_ops._x6()


# ---- StructureBuilder + the 2-view `noposplat_token_style` registry entry ------------------------------------------------
SB = np.load(Path(__file__).resolve().parent / "golden" / "structure_builder.npz")
SB_TINY = dict(enc_depth=1, dec_depth=12, enc_embed_dim=128, dec_embed_dim=128, enc_num_heads=2, dec_num_heads=2,
               pos_embed="RoPE100", img_size=(512, 512))


def test_noposplat_token_style_keys_match_reference_constructor():
    """state-dict key set and parameter count of the reference's EncoderNoPoSplatTokenStyle.__init__
    (encoder_noposplat_token_style.py:73-116), and the registry entry (src/model/encoder/__init__.py:10-15)"""
    from styl3r_amd.encoder import ENCODERS, EncoderNoPoSplatTokenStyle, EncoderNoPoSplatTokenStyleCfg, GaussianAdapterCfg, get_encoder
    cfg = EncoderNoPoSplatTokenStyleCfg(name="noposplat_token_style", gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, 1))
    m = EncoderNoPoSplatTokenStyle(cfg, trunk_params=SB_TINY)
    assert sorted(m.state_dict().keys()) == list(SB["enc_keys"])
    assert sum(p.numel() for p in m.parameters()) == int(SB["enc_nparams"])
    assert ENCODERS["noposplat_token_style"] is EncoderNoPoSplatTokenStyle
    assert set(ENCODERS) == {"noposplat", "noposplat_multi", "noposplat_token_style", "noposplat_multi_token_style"}


@pytest.mark.gpu
def test_structure_builder_matches_reference():
    """StructureBuilder.forward (structure_builder.py:128-141) against the reference module's outputs and gradients"""
    from styl3r_amd.encoder import StructureBuilder
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(StructureBuilder(SB_TINY).eval()).to(dev)
    T = lambda k: torch.tensor(SB[k], device=dev)
    f1, f2 = T("f1").requires_grad_(True), T("f2").requires_grad_(True)
    d1, d2 = m(f1, T("pos"), f2, T("pos"))
    assert len(d1) == len(d2) == int(SB["n_out"])
    for i in (0, 1, 6, 12):
        assert_close_rel(d1[i].detach().cpu().numpy(), SB[f"d1_{i}"], 2e-5, f"view 1 output {i}")
        assert_close_rel(d2[i].detach().cpu().numpy(), SB[f"d2_{i}"], 2e-5, f"view 2 output {i}")
    ((d1[-1] * T("w0")).sum() + (d2[-1] * T("w1")).sum() + (d1[6] * T("w2")).sum()).backward()
    assert_close_rel(f1.grad.cpu().numpy(), SB["gf1"], 1e-4, "d feat1")
    assert_close_rel(f2.grad.cpu().numpy(), SB["gf2"], 1e-4, "d feat2")
    assert_close_rel(m.dec_blocks[5].attn.qkv.weight.grad.cpu().numpy(), SB["g_qkv5"], 1e-4, "d dec_blocks.5 qkv")


@pytest.mark.gpu
def test_noposplat_token_style_forward_backward_runs_and_fused_adapter_agrees():
    """the 2-view style encoder end to end on the GPU (the reference's forward cannot run against its own current modules, see
    the class docstring): shapes, finiteness, gradients reach both trunks, fused adapter == element-wise path"""
    from styl3r_amd.encoder import EncoderNoPoSplatTokenStyle, EncoderNoPoSplatTokenStyleCfg, GaussianAdapterCfg
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    cfg = EncoderNoPoSplatTokenStyleCfg(name="noposplat_token_style", gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, 1))
    m = deterministic_init_(EncoderNoPoSplatTokenStyle(cfg, trunk_params=TINY).eval()).to(dev)   # (the intrinsics token is 1024 wide)
    g = torch.Generator(dev).manual_seed(4)
    img = torch.rand(2, 2, 3, 32, 48, device=dev, generator=g) * 2 - 1
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]], device=dev).expand(2, 2, 3, 3).contiguous()
    style = torch.rand(2, 3, 32, 32, device=dev, generator=g) * 2 - 1
    res = {}
    for fused in (True, False):
        m.fused_adapter = fused
        gs = m(dict(image=img, intrinsics=K), dict(image=style), 0)
        assert gs.means.shape == (2, 2 * 32 * 48, 3) and gs.harmonics.shape == (2, 2 * 32 * 48, 3, 4)
        res[fused] = gs
    for name in ("means", "covariances", "harmonics", "opacities"):
        a, e = getattr(res[True], name), getattr(res[False], name)
        assert torch.isfinite(a).all()
        assert_close_rel(a.detach().cpu().numpy(), e.detach().cpu().numpy(), 1e-5, name)
    m.fused_adapter = True
    (res[True].means.sum() * 1e-3 + res[True].harmonics.sum() + res[True].opacities.sum()).backward()
    assert m.structure_builder.dec_blocks[0].attn.qkv.weight.grad.abs().sum() > 0
    assert m.token_stylizer.dec_blocks[0].cross_attn.projk.weight.grad.abs().sum() > 0
    assert m.backbone.enc_blocks[0].mlp.fc1.weight.grad.abs().sum() > 0
