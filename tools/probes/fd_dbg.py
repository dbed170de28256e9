import sys; sys.path.insert(0, ".")
import torch
from tests.test_gpu_properties import _scene, DEV
from styl3r_amd.decoder import Gaussians
sc, g, dec, args = _scene(1, 256, 3, seed=7)
gen = torch.Generator(DEV).manual_seed(1)
w = torch.rand(1, 3, 3, 256, 256, device=DEV, generator=gen).double()
m = g.means.clone().requires_grad_(True)
def loss(mm): return (dec.forward(Gaussians(mm, g.covariances, g.harmonics, g.opacities), *args).color.double() * w).sum()
L0 = loss(m); L0.backward()
d = torch.randn(m.shape, device=DEV, generator=gen)
an = (m.grad.double() * d.double()).sum().item()
for eps in (4e-3, 2e-3, 1e-3, 5e-4, 2e-4, 1e-4, 5e-5):
    fd = ((loss(m.detach() + eps * d) - loss(m.detach() - eps * d)) / (2 * eps)).item()
    print(f"eps {eps:g}: fd {fd:.2f}  analytic {an:.2f}  L0 {L0.item():.3f}")
