"""Loss arithmetic (CPU): LossMse / LossStyle / IdentityLoss against a direct numpy restatement of
loss_mse.py:22-31, loss_style.py:35-79, loss_identity.py:26-52 on the same (random-weight) VGG features."""
import numpy as np
import torch

from styl3r_amd.decoder import DecoderOutput
from styl3r_amd.losses import IdentityLoss, LossMse, LossStyle, LossStyleCfg, VGGEncoder, calc_mean_std, compute_psnr


def _batch(b=2, v=3, h=32, w=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    tgt = torch.rand(b, v, 3, h, w, generator=g)
    pred = torch.rand(b, v, 3, h, w, generator=g)
    style = torch.rand(b, 3, 40, 40, generator=g)
    return DecoderOutput(pred, None), {"target": {"image": tgt}, "style": {"image": style}}


def test_vgg_layout_matches_torchvision_vgg19_features():
    vgg = VGGEncoder()
    keys = sorted(vgg.features.state_dict().keys(), key=lambda k: (int(k.split(".")[0]), k))
    assert [k for k in keys if k.endswith("weight")] == [f"{i}.weight" for i in (0, 2, 5, 7, 10, 12, 14, 16, 19)]
    assert vgg.features[19].weight.shape == (512, 256, 3, 3)
    h = vgg(torch.rand(1, 3, 64, 64))
    assert [t.shape[1:] for t in h] == [(64, 64, 64), (128, 32, 32), (256, 16, 16), (512, 8, 8)]
    sd = {"features." + k: torch.randn_like(v) for k, v in vgg.features.state_dict().items()}
    sd["features.21.weight"] = torch.zeros(1); sd["classifier.0.weight"] = torch.zeros(1)
    vgg.load_vgg19_features({k: v for k, v in sd.items() if k.startswith("features.")})
    assert torch.equal(vgg.features[0].weight, sd["features.0.weight"])


def test_mse_and_psnr():
    pred, batch = _batch()
    got = LossMse()(pred, batch)
    want = ((pred.color.numpy() - batch["target"]["image"].numpy()) ** 2).mean()
    np.testing.assert_allclose(float(got), want, rtol=1e-6)
    psnr = compute_psnr(batch["target"]["image"][0], pred.color[0])
    assert psnr.shape == (3,) and torch.all(psnr > 0)


def test_style_and_identity_losses_match_numpy_restatement():
    torch.manual_seed(1)
    vgg = VGGEncoder()
    pred, batch = _batch(seed=3)
    mean = np.array([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1); std = np.array([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1)
    norm = lambda a: torch.tensor((a - mean) / std, dtype=torch.float32)
    P = pred.color.reshape(6, 3, 32, 32).numpy(); T = batch["target"]["image"].reshape(6, 3, 32, 32).numpy()
    S = np.repeat(batch["style"]["image"].numpy()[:, None], 3, 1).reshape(6, 3, 40, 40)
    fp, ft, fs = (tuple(t.numpy().astype(np.float64) for t in vgg(norm(x))) for x in (P, T, S))
    mse = lambda a, b: ((a - b) ** 2).mean()
    ms = lambda f: (f.reshape(*f.shape[:2], -1).mean(-1), f.reshape(*f.shape[:2], -1).std(-1, ddof=1) + 1e-8)
    content = mse(fp[-2], ft[-2]) + mse(fp[-1], ft[-1])
    style = sum(mse(ms(a)[0], ms(s)[0]) + mse(ms(a)[1], ms(s)[1]) for a, s in zip(fp, fs))
    got = LossStyle(LossStyleCfg(10.0), vgg)(pred, batch)
    np.testing.assert_allclose(float(got), content + 10.0 * style, rtol=2e-5)
    ident = 70 * mse(P, T) + sum(mse(a, t) for a, t in zip(fp, ft))
    np.testing.assert_allclose(float(IdentityLoss(70, 1, vgg)(pred, batch)), ident, rtol=2e-5)
    m, s = calc_mean_std(torch.tensor(fp[0], dtype=torch.float32))
    assert m.shape == (6, 64, 1) and s.shape == (6, 64, 1)


def test_lpips_structure_and_gating():
    """LPIPS-VGG layout with the lpips package's parameter names; LossLpips gating and zero at identity"""
    import torch
    from types import SimpleNamespace
    from styl3r_amd.losses import LPIPS, LossLpips, LossLpipsCfg
    torch.manual_seed(0)
    m = LPIPS()
    keys = set(m.state_dict().keys())
    assert {"net.slice1.0.weight", "net.slice1.2.bias", "net.slice2.5.weight", "net.slice3.10.weight", "net.slice4.17.weight",
            "net.slice5.24.weight", "lin0.model.1.weight", "lin4.model.1.weight"} <= keys
    assert m.lin2.model[1].weight.shape == (1, 256, 1, 1) and sum(p.numel() for p in m.net.parameters()) == 14_714_688
    loss = LossLpips(LossLpipsCfg(weight=0.05, apply_after_step=10), m)
    img = torch.rand(1, 2, 3, 32, 32)
    pred = SimpleNamespace(color=img.clone().requires_grad_(True))
    batch = {"target": {"image": img}}
    assert float(loss(pred, batch, None, 5)) == 0.0                          # before apply_after_step
    assert abs(float(loss(pred, batch, None, 10))) < 1e-12                   # identical images
    other = SimpleNamespace(color=(img * 0.5).requires_grad_(True))
    l = loss(other, batch, None, 10)
    l.backward()
    assert torch.isfinite(l) and other.color.grad.abs().sum() > 0


def _ref_case():
    from pathlib import Path
    from tests.helpers import deterministic_vgg_
    G = np.load(Path(__file__).resolve().parent / "golden" / "losses_ref.npz")
    return G, deterministic_vgg_(VGGEncoder())


def test_style_and_identity_losses_match_the_reference_modules():
    """LossStyle / IdentityLoss against the REFERENCE's own modules (src/loss/loss_style.py:35-79, loss_identity.py:26-49,
    src/test/vgg_model.py:79-98) evaluated in float64 on deterministic VGG19 weights (tests/golden/make_loss_fixtures.py).
    (a) this repo's modules in float64 reproduce value and d/d(prediction) to the fixture's storage precision: the arithmetic
        is pinned;
    (b) the fp32 path gives the value to 2e-5; its gradient agrees element-wise except where a ReLU / max-pool decision sits
        within fp32 round-off of a tie and flips (a flip at relu4_1 moves a 44 x 44-pixel patch of the input gradient; the
        reference's own fp32 run has the same property), so the bar is: >= 97 % of the elements within 1e-4 of max|g| and a
        2 % relative L2 error."""
    G, vgg = _ref_case()
    T = lambda k: torch.tensor(G[k])
    from styl3r_amd.losses import _imagenet_normalize
    feats = vgg(_imagenet_normalize(T("pred").reshape(4, 3, 64, 64)))
    for k, f in enumerate(feats):
        want = G[f"vgg_h{k + 1}_mean"]
        assert np.abs(f.mean(dim=(2, 3)).numpy() - want).max() <= 2e-5 * np.abs(want).max(), k
    for dt in (torch.float64, torch.float32):
        net = vgg.double() if dt == torch.float64 else vgg.float()
        batch = {"target": {"image": T("target").to(dt)}, "style": {"image": T("style").to(dt)}}
        for name, mod in (("style", LossStyle(LossStyleCfg(float(G["style_weight"])), net)), ("identity", IdentityLoss(70, 1, net))):
            p = T("pred").to(dt).clone().requires_grad_(True)
            val = mod(DecoderOutput(p, None), batch, None, 0)
            val.backward()
            want, gwant = float(G[f"{name}_value"]), G[f"{name}_grad"].astype(np.float64)
            err = np.abs(p.grad.numpy() - gwant)
            scale = np.abs(gwant).max()
            if dt == torch.float64:
                assert abs(float(val.detach()) - want) <= 1e-12 * abs(want), (name, float(val.detach()), want)
                assert err.max() <= 2e-7 * scale, (name, err.max() / scale)       # the fixture stores float32
            else:
                assert abs(float(val.detach()) - want) <= 2e-5 * abs(want), (name, float(val.detach()), want)
                assert (err <= 1e-4 * scale).mean() >= 0.97, (name, (err <= 1e-4 * scale).mean())
                assert np.linalg.norm(err) <= 2e-2 * np.linalg.norm(gwant), name
