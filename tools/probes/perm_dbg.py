import sys; sys.path.insert(0, ".")
import torch
from tests.test_gpu_properties import _scene, DEV
from styl3r_amd.decoder import Gaussians
sc, g, dec, args = _scene(2, 256, 4, seed=5)
base = dec.forward(g, *args).color
perm = torch.randperm(g.means.shape[1], device=DEV, generator=torch.Generator(DEV).manual_seed(0))
gp = Gaussians(g.means[:, perm], g.covariances[:, perm], g.harmonics[:, perm], g.opacities[:, perm])
sh = dec.forward(gp, *args).color
d = (sh - base).abs()
print("max", d.max().item(), "n>1e-6", int((d > 1e-6).sum()), "of", d.numel(), "n>1e-4", int((d > 1e-4).sum()))
base2 = dec.forward(g, *args).color
print("rerun same", (base2 - base).abs().max().item())
