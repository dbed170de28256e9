"""Decoder boundary: the reference's `Decoder` API on top of the batched HIP
rasterizer.

Mirrors
  * `DecoderSplattingCUDA.forward`  src/model/decoder/decoder_splatting_cuda.py:37-68
  * `render_cuda`                   src/model/decoder/cuda_splatting.py:46-133
  * `DecoderOutput`                 src/model/decoder/decoder.py:18-24
with two MI355X-first changes that keep the results identical: the Gaussian
tensors are NOT replicated per view (the kernels index scene = view // v), and
all b*v views go through ONE launch sequence instead of a Python loop with two
host syncs per view; the scale-invariance rescale (cuda_splatting.py:65-72) is
folded into the kernels through `GsrView.scale`.
"""
from __future__ import annotations

from dataclasses import dataclass
from math import isqrt
from typing import Literal, Optional

import torch
from torch import Tensor, nn

from .camera import get_fov, get_projection_matrix
from .rasterizer import RasterMse, pack_views, rasterize_views

DepthRenderingMode = Literal["depth", "disparity", "relative_disparity", "log"]


@dataclass
class Gaussians:
    """src/model/types.py:8-12"""
    means: Tensor        # (b,G,3)
    covariances: Tensor  # (b,G,3,3)
    harmonics: Tensor    # (b,G,3,d_sh)
    opacities: Tensor    # (b,G)


@dataclass
class DecoderOutput:
    color: Tensor            # (b,v,3,h,w)
    depth: Optional[Tensor]  # (b,v,h,w)
    # not in the reference's DecoderOutput: LossMse of `color` against the `mse_target` handed to forward(), computed inside the
    # composite kernels (None without a target)
    loss_mse: Optional[Tensor] = None


@dataclass
class DecoderSplattingCUDACfg:
    name: Literal["splatting_cuda"]
    background_color: list
    make_scale_invariant: bool


_TRIU = ((0, 0, 0, 1, 1, 2), (0, 1, 2, 1, 2, 2))


def prepare_views(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                  scale_invariant: bool) -> Tensor:
    """(n,4,4) c2w, (n,3,3), (n,), (n,), (n,3) -> packed (n,64) GsrView rows; pure device-side torch ops in the
    order of cuda_splatting.py:65-88."""
    scale = None
    if scale_invariant:
        scale = 1 / near
        extrinsics = extrinsics.clone()
        extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
        near = near * scale
        far = far * scale
    fov_x, fov_y = get_fov(intrinsics).unbind(dim=-1)
    tan_x = (0.5 * fov_x).tan()
    tan_y = (0.5 * fov_y).tan()
    proj_raw = get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)
    view = extrinsics.inverse().transpose(1, 2)
    full = view @ proj_raw
    return pack_views(view, full, proj_raw, extrinsics[:, :3, 3], tan_x, tan_y, background, scale)


def build_views_hip(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                    scale_invariant: bool) -> Tensor:
    """Same result as `prepare_views` (to a few ulp: different 4x4 inverse / no torch intermediates) from ONE
    kernel (`gsr_build_views`) instead of ~100 tiny device ops; the per-step default of the decoder."""
    import ctypes as C
    from . import _lib
    n = extrinsics.shape[0]
    f = lambda t: t.detach().contiguous().float()
    e, k, nr, fr, bg = f(extrinsics), f(intrinsics), f(near), f(far), f(background)
    out = torch.empty((n, _lib.GSR_VIEW_FLOATS), dtype=torch.float32, device=e.device)
    rc = _lib.load().gsr_build_views(e.data_ptr(), k.data_ptr(), nr.data_ptr(), fr.data_ptr(), bg.data_ptr(), n,
                                     1 if scale_invariant else 0, out.data_ptr(),
                                     C.c_void_p(torch.cuda.current_stream(e.device).cuda_stream))
    _lib.check(rc, "gsr_build_views")
    return out


def render_hip(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape, background_color: Tensor,
               gaussians: Gaussians, views_per_scene: int, scale_invariant: bool = True, use_sh: bool = True,
               cam_rot_delta: Optional[Tensor] = None, cam_trans_delta: Optional[Tensor] = None,
               torch_view_setup: bool = False, mse: Optional[RasterMse] = None):
    """Batched counterpart of `render_cuda`: (b*v) cameras, b un-replicated Gaussian sets.
    Returns (color (b*v,3,h,w), depth (b*v,h,w)), plus the fused LossMse scalar when `mse` is given."""
    n = gaussians.harmonics.shape[-1]
    degree = isqrt(n) - 1
    shs = gaussians.harmonics.permute(0, 1, 3, 2).contiguous()            # (b,g,n,3)  cuda_splatting.py:76
    # cuda_splatting.py:118,126 passes covariances[:, triu]; the kernels read that upper triangle straight from
    # the (b,g,3,3) tensor (GSR_FLAG_COV9) and write its gradient there, so no gather / index_put kernels run.
    cov6 = gaussians.covariances
    # cameras carry no gradient in the reference either (pose gradients flow through theta / rho)
    if torch_view_setup or not extrinsics.is_cuda:
        views = prepare_views(extrinsics, intrinsics, near, far, background_color, scale_invariant)
    else:
        views = build_views_hip(extrinsics, intrinsics, near, far, background_color, scale_invariant)
    colors = shs if use_sh else shs[:, :, 0, :].contiguous()
    out = rasterize_views(gaussians.means, cov6, gaussians.opacities, colors, views, image_shape, views_per_scene,
                          sh_degree=degree, use_sh=use_sh, theta=cam_rot_delta, rho=cam_trans_delta, mse=mse)
    return (out.image, out.depth) if mse is None else (out.image, out.depth, out.loss_mse)


def render_hip_orthographic(extrinsics: Tensor, width: Tensor, height: Tensor, near: Tensor, far: Tensor, image_shape,
                            background_color: Tensor, gaussians: Gaussians, views_per_scene: int,
                            fov_degrees: float = 0.1, use_sh: bool = True, dump: Optional[dict] = None) -> Tensor:
    """`render_cuda_orthographic` (cuda_splatting.py:136-227, validation visualisations): a fake orthographic
    projection = camera moved back along -z with a tiny field of view.  (n,4,4) c2w, (n,) width/height/near/far in
    world units -> colour (n,3,h,w)."""
    n_coef = gaussians.harmonics.shape[-1]
    degree = isqrt(n_coef) - 1
    shs = gaussians.harmonics.permute(0, 1, 3, 2).contiguous()
    dev = extrinsics.device
    fov_x = torch.tensor(fov_degrees, device=dev).deg2rad()
    tan_fov_x = (0.5 * fov_x).tan()
    distance_to_near = (0.5 * width) / tan_fov_x
    tan_fov_y = 0.5 * height / distance_to_near
    fov_y = (2 * tan_fov_y).atan()
    near = near + distance_to_near
    far = far + distance_to_near
    move_back = torch.eye(4, dtype=torch.float32, device=dev)[None].repeat(extrinsics.shape[0], 1, 1)
    move_back[:, 2, 3] = -distance_to_near
    extrinsics = extrinsics @ move_back
    if dump is not None:
        dump.update(extrinsics=extrinsics, fov_x=fov_x, fov_y=fov_y, near=near, far=far)
    b = extrinsics.shape[0]
    proj_raw = get_projection_matrix(near, far, fov_x.expand(b), fov_y).transpose(1, 2)
    view = extrinsics.inverse().transpose(1, 2)
    views = pack_views(view, view @ proj_raw, proj_raw, extrinsics[:, :3, 3], tan_fov_x.expand(b), tan_fov_y, background_color)
    colors = shs if use_sh else shs[:, :, 0, :].contiguous()
    out = rasterize_views(gaussians.means, gaussians.covariances, gaussians.opacities, colors, views, image_shape,
                          views_per_scene, sh_degree=degree, use_sh=use_sh)
    return out.image


class DecoderSplattingHIP(nn.Module):
    """Same constructor cfg / forward signature / DecoderOutput as DecoderSplattingCUDA."""

    def __init__(self, cfg: DecoderSplattingCUDACfg) -> None:
        super().__init__()
        self.cfg = cfg
        self.make_scale_invariant = cfg.make_scale_invariant
        # True: build the cameras with the reference's torch op sequence (prepare_views, bit-identical to
        # render_cuda on the same device) instead of the single gsr_build_views kernel (a few ulp apart)
        self.torch_view_setup = False
        self.register_buffer("background_color", torch.tensor(cfg.background_color, dtype=torch.float32),
                             persistent=False)
        self._bg_rows = None   # (key, (n,3) contiguous copy of the background): one expand-copy kernel per shape instead of one per call

    def _background_rows(self, n: int) -> Tensor:
        bg = self.background_color
        key = (n, bg.device, bg.data_ptr(), bg._version)
        if self._bg_rows is None or self._bg_rows[0] != key:
            self._bg_rows = (key, bg.detach()[None].expand(n, 3).contiguous())
        return self._bg_rows[1]

    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape, depth_mode: Optional[DepthRenderingMode] = None,
                cam_rot_delta: Optional[Tensor] = None, cam_trans_delta: Optional[Tensor] = None,
                mse_target: Optional[Tensor] = None, mse_weight: float = 1.0) -> DecoderOutput:
        """`mse_target` (b,v,3,h,w), not in the reference's signature: the ground-truth images of `LossMse` (loss_mse.py:22-31).  The loss
        is then computed inside the compositing kernels and returned as `DecoderOutput.loss_mse` -- equal to
        `mse_weight * ((color - mse_target) ** 2).mean()`, its backward formed in the composite backward's prologue -- instead of by two more
        kernels and a round trip of dL/dcolor through HBM."""
        b, v = extrinsics.shape[:2]
        flat = lambda t: t.reshape(b * v, *t.shape[2:])
        h, w = image_shape
        mse = None if mse_target is None else RasterMse(mse_target.reshape(b * v, 3, h, w), mse_weight)
        out = render_hip(
            flat(extrinsics), flat(intrinsics), flat(near), flat(far), image_shape,
            self._background_rows(b * v), gaussians, v,
            scale_invariant=self.make_scale_invariant, torch_view_setup=self.torch_view_setup,
            cam_rot_delta=flat(cam_rot_delta) if cam_rot_delta is not None else None,
            cam_trans_delta=flat(cam_trans_delta) if cam_trans_delta is not None else None, mse=mse)
        return DecoderOutput(out[0].reshape(b, v, 3, h, w), out[1].reshape(b, v, h, w), out[2] if mse is not None else None)


def get_decoder(cfg: DecoderSplattingCUDACfg) -> DecoderSplattingHIP:
    """src/model/decoder/__init__.py registry: {"splatting_cuda": ...}."""
    if cfg.name != "splatting_cuda":
        raise KeyError(cfg.name)
    return DecoderSplattingHIP(cfg)
