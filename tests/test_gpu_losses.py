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
