"""Shared test helpers: tiny hand-built scenes and numpy views of scenes."""
import numpy as np
import torch

from styl3r_amd.camera import build_view_setup


def simple_camera(H=32, W=32, fx=0.86, near=1.0, far=100.0, c2w=None):
    """One pinhole camera; returns dict of numpy float64 arrays in rasterizer convention."""
    ext = torch.eye(4)[None] if c2w is None else torch.as_tensor(c2w, dtype=torch.float32)[None]
    K = torch.tensor([[[fx, 0, 0.5], [0, fx, 0.5], [0, 0, 1.0]]])
    vs = build_view_setup(ext, K, torch.tensor([near]), torch.tensor([far]))
    return dict(H=H, W=W, tanfovx=float(vs.tanfovx[0]), tanfovy=float(vs.tanfovy[0]),
                view=vs.viewmatrix[0].numpy().astype(np.float64), proj=vs.projmatrix[0].numpy().astype(np.float64),
                proj_raw=vs.projmatrix_raw[0].numpy().astype(np.float64), campos=vs.campos[0].numpy().astype(np.float64))


def iso_cov6(s):
    """isotropic covariance s^2 I as the 6-vector xx,xy,xz,yy,yz,zz"""
    return np.array([s * s, 0, 0, s * s, 0, s * s], dtype=np.float64)


def random_scene(G, seed=0, z_range=(2.0, 6.0), spread=1.2, scale=(0.02, 0.12), sh_degree=0, op_range=(0.2, 0.95)):
    rng = np.random.default_rng(seed)
    z = rng.uniform(*z_range, G)
    xy = rng.uniform(-spread, spread, (G, 2)) * z[:, None] * 0.4
    means = np.concatenate([xy, z[:, None]], 1)
    A = rng.normal(size=(G, 3, 3))
    Q, _ = np.linalg.qr(A)
    s = rng.uniform(*scale, (G, 3)) * z[:, None] * 0.3
    cov = Q @ (s[:, :, None] ** 2 * np.eye(3)) @ Q.transpose(0, 2, 1)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
    opac = rng.uniform(*op_range, G)
    M = (sh_degree + 1) ** 2
    shs = rng.normal(size=(G, M, 3)) * (0.5 / np.sqrt(np.arange(M) + 1.0))[None, :, None]
    return means, cov6, opac, shs


def deterministic_init_(module, scale=1.0):
    """Name-keyed deterministic parameters, shared by the fixture generator (applied to the REFERENCE
    model) and the tests (applied to this repo's model): identical state_dict keys => identical weights,
    so a 100M-parameter state_dict never has to be stored."""
    import zlib
    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd.keys()):
            t = sd[name]
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if t.dim() >= 2:
                fan_in = t[0].numel()
                val = torch.randn(t.shape, generator=g) * (scale / fan_in ** 0.5)
            elif name.endswith("weight"):          # LayerNorm gains
                val = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            else:
                val = 0.02 * torch.randn(t.shape, generator=g)
            t.copy_(val.to(t.device))
    return module


def closed_form_weights(shape, k):
    """Loss weights for the large fixtures, as a closed form of the flat index (bounded, sign-changing, non-periodic over
    the tensor): identical on the generator and the test side without storing them or relying on an RNG stream."""
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.float64)
    w = torch.cos(0.7390851 * i + 1.3 * (k + 1)) + 0.5 * torch.sin(0.0113 * i * (k + 2) + 0.4)
    return w.reshape(shape).float()


def deterministic_vgg_(module):
    """VGG19 `features` weights keyed by the torchvision layer index (the trailing `<N>.weight` / `<N>.bias` of the state-dict
    key), He-scaled so activations keep their magnitude through the nine convolutions: the reference's VGGEncoder
    (`slice2.5.weight`, buffers after convert_to_buffer) and this repo's (`features.5.weight`) receive identical values."""
    import zlib
    sd = dict(module.named_parameters())
    sd.update(dict(module.named_buffers()))
    with torch.no_grad():
        for name, t in sd.items():
            if not t.is_floating_point():
                continue
            tail = ".".join(name.split(".")[-2:])
            g = torch.Generator().manual_seed(zlib.crc32(("vgg19.features." + tail).encode()))
            if t.dim() == 4:
                val = torch.randn(t.shape, generator=g) * (2.0 / t[0].numel()) ** 0.5
            else:
                val = 0.05 * torch.randn(t.shape, generator=g)
            t.copy_(val.to(t.device, t.dtype))
    return module
