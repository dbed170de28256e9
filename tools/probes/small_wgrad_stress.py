"""Repeatability of the small-spatial 3x3 weight gradient (Conv2dX6 backward, 'linear' path) at the DPT shapes of a 128 x 160 image:
every repetition must reproduce the first result (up to atomics reordering) and match the library's."""
import sys; sys.path.insert(0, ".")
import torch
from styl3r_amd import vit_ops
from styl3r_amd.vit_ops import Conv2dX6
dev = "cuda:0"
torch.manual_seed(0)
shapes = [(1, 256, 256, 128, 160), (3, 256, 256, 128, 160), (1, 96, 256, 32, 40), (3, 192, 256, 16, 20), (3, 768, 256, 4, 5), (1, 384, 256, 8, 10),
          (4, 256, 256, 64, 80), (3, 256, 128, 64, 80), (1, 128, 128, 128, 160)]
for (B, Ci, Co, H, W) in shapes:
    conv = Conv2dX6(Ci, Co, 3, 1, 1).to(dev)
    x = torch.randn(B, Ci, H, W, device=dev, requires_grad=True)
    g = torch.randn(B, Co, H, W, device=dev)
    def grads(mode):
        vit_ops.SMALL_CONV_WGRAD = mode
        return torch.autograd.grad(conv.forward_fused(x) if Ci == Co else conv(x), (x, conv.weight, conv.bias), g)
    lib = grads("library")
    worst = 0.0; first = None
    for rep in range(12):
        # churn the allocator between repetitions like a real backward does
        junk = [torch.randn(1 << 20, device=dev) for _ in range(3)]
        got = grads("linear")
        del junk
        if first is None:
            first = got
        for a, b in zip(got, first):
            worst = max(worst, float((a - b).abs().max() / b.abs().max()))
    vs_lib = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(first, lib)]
    print(f"B={B} Ci={Ci} Co={Co} {H}x{W}: worst deviation between repetitions {worst:.2e}; vs library dx {vs_lib[0]:.1e} dW {vs_lib[1]:.1e} db {vs_lib[2]:.1e}", flush=True)
