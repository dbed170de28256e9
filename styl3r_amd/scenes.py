"""Seeded synthetic scenes with the statistics of the encoder's output
(SURVEY.md section 8d): one Gaussian per context pixel, RE10K-like cameras.

Used by bench.py, smoke() and the tests; pure CPU torch so that every rank and
the CPU oracle see identical bytes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

FX = 0.86  # normalised focal length (fov ~ 60 deg, tan(fov/2) = 0.5814)


@dataclass
class Scene:
    means: Tensor         # (G,3)
    covariances: Tensor   # (G,3,3)
    harmonics: Tensor     # (G,3,d_sh)
    opacities: Tensor     # (G,)
    extrinsics: Tensor    # (V,4,4) target cameras, camera-to-world
    intrinsics: Tensor    # (V,3,3) normalised
    near: Tensor          # (V,)
    far: Tensor           # (V,)
    image_shape: tuple

    def to(self, device):
        kw = {k: (v.to(device) if isinstance(v, Tensor) else v) for k, v in self.__dict__.items()}
        return Scene(**kw)


def _quat_to_rot(q: Tensor) -> Tensor:
    """xyzw unit quaternions (n,4) -> (n,3,3) (src/model/encoder/common/gaussians.py:8-32 convention)."""
    x, y, z, w = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def sh_mask(d_sh: int) -> Tensor:
    """per-coefficient damping 1, 0.025, 0.00625, ... (gaussian_adapter.py:42-48)."""
    deg = int(math.isqrt(d_sh)) - 1
    m = torch.ones(d_sh)
    for l in range(1, deg + 1):
        m[l * l:(l + 1) * (l + 1)] = 0.1 * 0.25 ** l
    return m


def make_scene(n_ctx: int = 1, grid_hw: tuple = (256, 256), n_views: int = 3, image_hw: tuple = (256, 256),
               sh_degree: int = 0, seed: int = 1234, near: float = 0.1, far: float = 100.0) -> Scene:
    """G = n_ctx * grid_h * grid_w Gaussians seen from ``n_views`` target cameras."""
    g = torch.Generator().manual_seed(seed)
    gh, gw = grid_hw
    d_sh = (sh_degree + 1) ** 2
    means, covs, ops, shs = [], [], [], []
    for k in range(n_ctx):
        # context camera k: identity ... translated 1.0 along +x (baseline-1 normalisation)
        tx = 0.0 if n_ctx == 1 else k / (n_ctx - 1)
        ys, xs = torch.meshgrid((torch.arange(gh) + 0.5) / gh, (torch.arange(gw) + 0.5) / gw, indexing="ij")
        # low-frequency depth field in [1,5] + small noise
        fy, fx_ = torch.rand(2, generator=g) * 2 + 0.5
        ph = torch.rand(2, generator=g) * 6.28
        depth = 3.0 + 1.4 * torch.sin(6.28 * fx_ * xs + ph[0]) * torch.cos(6.28 * fy * ys + ph[1])
        depth = (depth + 0.02 * torch.randn(gh, gw, generator=g)).clamp(1.0, 5.0)
        dirs = torch.stack([(xs - 0.5) / FX, (ys - 0.5) / FX, torch.ones_like(xs)], -1)
        pts = dirs * depth[..., None]
        pts[..., 0] += tx
        n = gh * gw
        foot = depth.reshape(n) / (FX * max(gh, gw))
        scales = foot[:, None] * (0.5 + 1.5 * torch.rand(n, 3, generator=g))
        q = torch.randn(n, 4, generator=g)
        q = q / q.norm(dim=-1, keepdim=True)
        R = _quat_to_rot(q)
        cov = R @ torch.diag_embed(scales ** 2) @ R.transpose(1, 2)
        means.append(pts.reshape(n, 3))
        covs.append(cov)
        ops.append(torch.sigmoid(1.5 * torch.randn(n, generator=g)))
        shs.append(torch.randn(n, 3, d_sh, generator=g) * sh_mask(d_sh))
    # target cameras interpolated between the first and last context camera (+ a little jitter)
    ext = torch.eye(4).repeat(n_views, 1, 1)
    for v in range(n_views):
        t = (v + 0.5) / n_views
        ext[v, 0, 3] = t * (1.0 if n_ctx > 1 else 0.3)
        ext[v, 1, 3] = 0.05 * math.sin(3.0 * t)
        ang = 0.05 * (t - 0.5)
        ext[v, 0, 0] = math.cos(ang); ext[v, 0, 2] = math.sin(ang)
        ext[v, 2, 0] = -math.sin(ang); ext[v, 2, 2] = math.cos(ang)
    K = torch.tensor([[FX, 0, 0.5], [0, FX, 0.5], [0, 0, 1.0]]).repeat(n_views, 1, 1)
    return Scene(torch.cat(means).float(), torch.cat(covs).float(), torch.cat(shs).float(), torch.cat(ops).float(),
                 ext, K, torch.full((n_views,), near), torch.full((n_views,), far), tuple(image_hw))


# Target statistics (mean, std per output channel) of the five 1x1 output convolutions of the encoder's DPT heads after re-centring:
# xyz of the point heads (depth expm1(|xyz|) ~ 2..4, inside a tan(fov/2) = 0.58 frustum), (opacity logit, 3 log-scales, 4 quaternion)
# of the gs heads, SH DC of the appearance head.  A RANDOM-INIT encoder emits expm1 of a heavy-tailed norm (|means| up to 1e4, nearly
# everything outside every frustum): benchmarks and fixtures that must render a real scene call `recentre_output_heads_` once.
HEAD_TARGETS = {
    "downstream_head1": ([0.0, 0.0, 1.25], [0.30, 0.30, 0.12]),
    "downstream_head2": ([0.0, 0.0, 1.25], [0.30, 0.30, 0.12]),
    "gaussian_param_head": ([0.5, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0, 0.0], [1.2, 1.5, 1.5, 1.5, 1.0, 1.0, 1.0, 1.0]),
    "gaussian_param_head2": ([0.5, 2.0, 2.0, 2.0, 0.0, 0.0, 0.0, 0.0], [1.2, 1.5, 1.5, 1.5, 1.0, 1.0, 1.0, 1.0]),
    "gaussian_appearance_head": ([0.0, 0.0, 0.0], [1.2, 1.2, 1.2]),
}


@torch.no_grad()
def recentre_output_heads_(encoder, context: dict, style: dict, targets: dict = None) -> dict:
    """One calibration forward (no gradients), then an affine re-parametrisation of each head's last 1x1 convolution so that its output
    channels have the target mean / std on this batch: set-up work for benchmarks on random-init weights (checkpoints are absent here),
    the same calibration tests/golden/make_e2e_fixtures.py applies to the reference model.  Returns {head: (mean, std) before}."""
    targets = targets or HEAD_TARGETS
    heads = [k for k in targets if hasattr(encoder, k)]
    stats, hooks = {}, []
    for k in heads:
        hooks.append(getattr(encoder, k).dpt.register_forward_hook(
            lambda m, i, o, k=k: stats.setdefault(k, []).append(o.detach().transpose(0, 1).reshape(o.shape[1], -1).double())))
    was_training = encoder.training
    encoder.eval()
    try:
        encoder(context, style, 0)
    finally:
        for h in hooks:
            h.remove()
        encoder.train(was_training)
    before = {}
    for k in heads:
        o = torch.cat(stats[k], 1)
        mean, std = o.mean(1), o.std(1)
        t_mean, t_std = (torch.tensor(x, dtype=torch.float64, device=o.device) for x in targets[k])
        if t_mean.numel() != mean.numel():     # (appearance head at sh_degree > 0: 3 x d_sh channels; keep the total colour energy of the degree-0 target)
            t_std = torch.full_like(mean, float(t_std.mean()) * (t_std.numel() / mean.numel()) ** 0.5)
            t_mean = torch.zeros_like(mean)
        conv = getattr(encoder, k).dpt.head[4]
        gain = t_std / std
        conv.weight.copy_((conv.weight.double() * gain[:, None, None, None]).float())
        conv.bias.copy_(((conv.bias.double() - mean) * gain + t_mean).float())
        torch.autograd.graph.increment_version([conv.weight, conv.bias])
        before[k] = (mean.tolist(), std.tolist())
    return before
