"""-m gpu: the fused MSE kernels (include/gsr.h gsr_mse_forward/backward) against the torch expression of
`LossMse.forward` (src/loss/loss_mse.py:27-31) evaluated in fp64."""
import pytest
import torch

from styl3r_amd.losses import mse_loss

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 3, 3, 16, 16), (40, 3, 256, 256), (1, 3, 7, 11), (5,)])
def test_mse_forward_backward_match_torch(shape):
    dev = torch.device("cuda:0")
    g = torch.Generator(dev).manual_seed(3)
    pred = torch.rand(shape, device=dev, generator=g).requires_grad_(True)
    target = torch.rand(shape, device=dev, generator=g)
    for rep in range(2):                                   # second launch exercises the re-armed ticket
        pred.grad = None
        loss = mse_loss(pred, target, 0.7)
        (loss * 1.5).backward()
        p64 = pred.detach().double().requires_grad_(True)
        want = 0.7 * ((p64 - target.double()) ** 2).mean()
        (want * 1.5).backward()
        assert abs(loss.item() - want.item()) <= 2e-6 * abs(want.item())
        assert torch.allclose(pred.grad.double(), p64.grad, rtol=1e-6, atol=1e-12)
    # bitwise run-to-run determinism of the ticketed reduction
    assert float(mse_loss(pred.detach(), target)) == float(mse_loss(pred.detach(), target))


def test_mse_noncontiguous_and_cpu_dispatch():
    dev = torch.device("cuda:0")
    a = torch.rand(4, 6, 8, device=dev).transpose(0, 2)
    b = torch.rand(8, 6, 4, device=dev)
    assert torch.allclose(mse_loss(a, b), ((a - b) ** 2).mean(), rtol=1e-6)
    assert torch.allclose(mse_loss(a.cpu(), b.cpu()), ((a - b) ** 2).mean().cpu(), rtol=1e-6)   # CPU tensors: torch expression


def test_vgg_encoder_runs_on_the_bf16x6_convolutions_and_matches_fp64(monkeypatch):
    """VGGEncoder (relu1_1 .. relu4_1, src/test/vgg_model.py:79-98 in the reference) on the GPU: the layers with >= 96
    output channels take vit_conv_x6_fwd (forward and input gradient); features and d(sum of features)/d(image) agree
    with the same network evaluated in fp64 by the framework's convolution."""
    from styl3r_amd import vit_ops
    from styl3r_amd.losses import VGGEncoder
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = VGGEncoder().to(dev)
    ref = VGGEncoder().double().to(dev)
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    x = torch.rand(8, 3, 128, 128, device=dev).requires_grad_(True)
    x64 = x.detach().double().requires_grad_(True)
    before = dict(vit_ops.CALLS)
    feats = net(x)
    sum(f.square().mean() for f in feats).backward()
    assert vit_ops.CALLS["conv_x6_fwd"] - before["conv_x6_fwd"] >= 6      # conv2_1 .. conv4_1 (conv1_x: 64 channels -> MIOpen)
    assert vit_ops.CALLS["conv_x6_dx"] - before["conv_x6_dx"] >= 4        # those with >= 96 input channels and >= 100 tiles
    feats64 = ref(x64)                                                       # fp64 tensors never take the x6 path
    sum(f.square().mean() for f in feats64).backward()
    for f, f64 in zip(feats, feats64):
        assert float((f.double() - f64).abs().max() / f64.abs().max()) < 2e-5
    # the input gradient crosses seven ReLUs: an activation within fp32 round-off of zero flips its mask between fp32
    # and fp64, so the yardstick is the framework's own fp32 convolution against the same fp64 reference
    err_x6 = float((x.grad.double() - x64.grad).abs().max() / x64.grad.abs().max())
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "f32")                       # Conv2dX6 -> nn.Conv2d (MIOpen)
    x32 = x.detach().clone().requires_grad_(True)
    sum(f.square().mean() for f in net(x32)).backward()
    err_lib = float((x32.grad.double() - x64.grad).abs().max() / x64.grad.abs().max())
    assert err_x6 < max(1e-4, 3 * err_lib), (err_x6, err_lib)


def test_style_and_identity_losses_match_the_reference_modules_on_the_gpu():
    """the same reference-derived fixture as tests/test_losses.py, through the GPU path, value within 1e-4 of the float64 reference, d/d(prediction) to the fp32 bar of tests/test_losses.py"""
    import numpy as np
    from pathlib import Path
    from styl3r_amd import vit_ops
    from styl3r_amd.decoder import DecoderOutput
    from styl3r_amd.losses import IdentityLoss, LossStyle, LossStyleCfg, VGGEncoder
    from tests.helpers import deterministic_vgg_
    G = np.load(Path(__file__).resolve().parent / "golden" / "losses_ref.npz")
    dev = torch.device("cuda:0")
    vgg = deterministic_vgg_(VGGEncoder()).to(dev)
    T = lambda k: torch.tensor(G[k], device=dev)
    batch = {"target": {"image": T("target")}, "style": {"image": T("style")}}
    for name, mod in (("style", LossStyle(LossStyleCfg(float(G["style_weight"])), vgg)), ("identity", IdentityLoss(70, 1, vgg))):
        p = T("pred").clone().requires_grad_(True)
        val = mod(DecoderOutput(p, None), batch, None, 0)
        val.backward()
        want, gwant = float(G[f"{name}_value"]), G[f"{name}_grad"]
        assert abs(float(val.detach()) - want) <= 1e-4 * abs(want), (name, float(val.detach()), want)
        # gradient: element-wise 1e-4 except where a ReLU / max-pool tie flips in fp32 (see tests/test_losses.py)
        err = np.abs(p.grad.cpu().numpy() - gwant)
        scale = np.abs(gwant).max()
        assert (err <= 1e-4 * scale).mean() >= 0.85, (name, (err <= 1e-4 * scale).mean())
        assert np.linalg.norm(err) <= 3e-2 * np.linalg.norm(gwant), name
    # (at this fixture's 64 x 64 x 4 images no VGG layer reaches the >= 100-tile gate of the bf16x6 convolutions: the layers run on
    #  MIOpen here; the x6 VGG path has its own test above at 128 x 128 x 8)
