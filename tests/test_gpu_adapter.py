"""-m gpu: the fused head-tail + Gaussian-adapter kernels (include/vit_ops.h vit_adapter_fwd / vit_adapter_bwd) against the
element-wise expression of the same reference math -- reg_dense_depth(mode='exp') (postprocess.py:22-60), sigmoid +
map_pdf_to_opacity (encoder_noposplat_multi_token_style.py:115-128), UnifiedGaussianAdapter.forward
(gaussian_adapter.py:122-153), build_covariance (gaussians.py:8-44) -- evaluated by the framework in float64.  (That
expression itself is pinned to the reference by the encoder golden vectors, tests/test_encoder.py.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(pts0, ptsr, par0, parr, app, sh_mask, exponent, v):
    """float64 framework expression, same layout conventions as EncoderNoPoSplatMultiTokenStyle.forward"""
    from styl3r_amd.encoder import build_covariance, reg_dense_depth_exp
    import torch.nn.functional as F
    b, _, H, W = pts0.shape
    HW = H * W
    per_view = lambda first, rest: first.unsqueeze(1) if rest is None else torch.cat((first.unsqueeze(1), rest.reshape(b, v - 1, *rest.shape[1:])), 1)
    pts = per_view(pts0, ptsr).permute(0, 1, 3, 4, 2).reshape(b, v, HW, 3)
    par = per_view(par0, parr).flatten(3).transpose(2, 3)                     # (b, v, HW, C)
    means = reg_dense_depth_exp(pts)
    p = par[..., 0].sigmoid()
    opac = p if exponent == 1 else 0.5 * (1 - (1 - p) ** exponent + p ** (1 / exponent))
    scales = (0.001 * F.softplus(par[..., 1:4])).clamp_max(0.3)
    rot = par[..., 4:8] / (par[..., 4:8].norm(dim=-1, keepdim=True) + 1e-8)
    d_sh = sh_mask.numel()
    shraw = app.reshape(b, v, 3 * d_sh, HW).transpose(2, 3) if app is not None else par[..., 8:]
    sh = shraw.reshape(b, v, HW, 3, d_sh) * sh_mask
    cov = build_covariance(scales, rot)
    return (means.reshape(b, v * HW, 3), cov.reshape(b, v * HW, 3, 3), sh.reshape(b, v * HW, 3, d_sh), opac.reshape(b, v * HW),
            scales.reshape(b, v * HW, 3), rot.reshape(b, v * HW, 4))


@pytest.mark.parametrize("b,v,H,W,sh_degree,exponent,separate_app", [(2, 3, 16, 24, 0, 1.0, True), (1, 1, 8, 8, 1, 1.0, True),
                                                                    (2, 2, 12, 20, 2, 1.7, True), (1, 2, 16, 16, 1, 1.0, False),
                                                                    (1, 4, 32, 32, 4, 0.6, True)])
def test_adapter_kernels_match_float64_expression(b, v, H, W, sh_degree, exponent, separate_app):
    from styl3r_amd.vit_ops import gaussian_adapter_hip
    dev = torch.device("cuda:0")
    g = torch.Generator(dev).manual_seed(17 + v)
    d_sh = (sh_degree + 1) ** 2
    C = 8 if separate_app else 8 + 3 * d_sh
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    pts0, par0 = R(b, 3, H, W) * 0.8, R(b, C, H, W) * 2
    ptsr, parr = (R(b * (v - 1), 3, H, W) * 0.8, R(b * (v - 1), C, H, W) * 2) if v > 1 else (None, None)
    app = R(b * v, 3 * d_sh, H, W) if separate_app else None
    par0[0, 1:4, 0, :4] = torch.tensor([25.0, 7.0, -30.0], device=dev)[:, None]     # softplus threshold, clamp at 0.3, tiny scale
    mask = torch.ones(d_sh, device=dev)
    for deg in range(1, sh_degree + 1):
        mask[deg ** 2:(deg + 1) ** 2] = 0.1 * 0.25 ** deg
    ins = [t.requires_grad_(True) if t is not None else None for t in (pts0, ptsr, par0, parr, app)]
    out = gaussian_adapter_hip(*ins, mask, exponent, v, True)
    ins64 = [t.detach().double().requires_grad_(True) if t is not None else None for t in ins]
    ref = _reference(*ins64, mask.double(), exponent, v)
    names = ("means", "cov", "sh", "opac", "scales", "rot")
    for n, a, e in zip(names, out, ref):
        err = float((a.double() - e).abs().max() / e.abs().max().clamp_min(1e-30))
        assert err < 2e-6, (n, err)
    ws = [torch.randn(t.shape, device=dev, generator=g) for t in out[:4]]
    sum((a * w).sum() for a, w in zip(out[:4], ws)).backward()
    sum((e * w.double()).sum() for e, w in zip(ref[:4], ws)).backward()
    for n, a, e in zip(("pts0", "ptsr", "par0", "parr", "app"), ins, ins64):
        if a is None:
            continue
        err = float((a.grad.double() - e.grad).abs().max() / e.grad.abs().max().clamp_min(1e-30))
        assert err < 2e-5, (n, err)


def test_encoder_takes_the_fused_adapter_and_matches_the_elementwise_path():
    """the style encoder on a GPU routes E10-E12 through vit_adapter_*; Gaussians, visualization_dump and input gradients equal
    the element-wise path's (fused_adapter = False)"""
    from styl3r_amd import vit_ops
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg, GaussianAdapterCfg
    from tests.helpers import deterministic_init_
    from tests.gpu_utils import assert_close_rel
    tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
                pos_embed="RoPE100", img_size=(512, 512))
    dev = "cuda:0"
    m = deterministic_init_(EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, 1)),
                                                            trunk_params=tiny).eval()).to(dev)
    g = torch.Generator(dev).manual_seed(2)
    img = torch.rand(2, 3, 3, 32, 48, device=dev, generator=g) * 2 - 1
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]], device=dev).expand(2, 3, 3, 3).contiguous()
    style = torch.rand(2, 3, 32, 32, device=dev, generator=g) * 2 - 1
    res = {}
    for fused in (True, False):
        m.fused_adapter = fused
        before = vit_ops.CALLS["adapter_hip"]
        x = img.clone().requires_grad_(True)
        dump = {}
        gs = m(dict(image=x, intrinsics=K), dict(image=style), 0, dump)
        assert (vit_ops.CALLS["adapter_hip"] - before == 1) == fused
        (gs.means.sum() * 0.01 + gs.covariances.sum() * 1e3 + gs.harmonics.sum() + gs.opacities.sum()).backward()
        res[fused] = (gs, dump, x.grad)
    m.fused_adapter = True
    for name in ("means", "covariances", "harmonics", "opacities"):
        a, e = getattr(res[True][0], name), getattr(res[False][0], name)
        assert a.shape == e.shape
        assert_close_rel(a.detach().cpu().numpy(), e.detach().cpu().numpy(), 1e-5, name)
    for k in ("depth", "scales", "rotations", "means", "opacities"):
        a, e = res[True][1][k], res[False][1][k]
        assert a.shape == e.shape, k
        assert_close_rel(a.detach().cpu().numpy(), e.detach().cpu().numpy(), 1e-5, "dump " + k)
    assert_close_rel(res[True][2].cpu().numpy(), res[False][2].cpu().numpy(), 1e-3, "d image")   # deep fp32 gradient, two summation orders
