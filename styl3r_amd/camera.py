"""Camera set-up on the decoder boundary (host side, torch).

Restates, with the same operation order so that results agree bit-for-bit on
identical inputs, the matrix conventions of the reference wrapper:
  * ``get_fov``               -- src/geometry/projection.py:247-261
  * ``get_projection_matrix`` -- src/model/decoder/cuda_splatting.py:16-43
  * view / full-projection    -- src/model/decoder/cuda_splatting.py:81-88
All matrices leave here in the rasterizer's row-vector ("transposed") form.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
from torch import Tensor


def get_fov(intrinsics: Tensor) -> Tensor:
    """(b,3,3) normalised intrinsics -> (b,2) field of view (x, y) in radians."""
    inv = intrinsics.inverse()

    def ray(v):
        v = torch.tensor(v, dtype=torch.float32, device=intrinsics.device)
        d = torch.einsum("bij,j->bi", inv, v)
        return d / d.norm(dim=-1, keepdim=True)

    fov_x = (ray([0, 0.5, 1]) * ray([1, 0.5, 1])).sum(dim=-1).acos()
    fov_y = (ray([0.5, 0, 1]) * ray([0.5, 1, 1])).sum(dim=-1).acos()
    return torch.stack((fov_x, fov_y), dim=-1)


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """(b,) each -> (b,4,4); x,y to (-1,1), z to (0,1), z flipped (column-vector form)."""
    tx = (0.5 * fov_x).tan()
    ty = (0.5 * fov_y).tan()
    top = ty * near
    bottom = -top
    right = tx * near
    left = -right
    P = torch.zeros((near.shape[0], 4, 4), dtype=torch.float32, device=near.device)
    P[:, 0, 0] = 2 * near / (right - left)
    P[:, 1, 1] = 2 * near / (top - bottom)
    P[:, 0, 2] = (right + left) / (right - left)
    P[:, 1, 2] = (top + bottom) / (top - bottom)
    P[:, 3, 2] = 1
    P[:, 2, 2] = far / (far - near)
    P[:, 2, 3] = -(far * near) / (far - near)
    return P


class ViewSetup(NamedTuple):
    viewmatrix: Tensor      # (b,4,4) inverse(c2w)^T
    projmatrix: Tensor      # (b,4,4) viewmatrix @ projmatrix_raw
    projmatrix_raw: Tensor  # (b,4,4) P^T
    tanfovx: Tensor         # (b,)
    tanfovy: Tensor         # (b,)
    campos: Tensor          # (b,3) c2w translation


def build_view_setup(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor) -> ViewSetup:
    """extrinsics (b,4,4) camera-to-world; intrinsics (b,3,3) normalised."""
    fov_x, fov_y = get_fov(intrinsics).unbind(dim=-1)
    tan_x = (0.5 * fov_x).tan()
    tan_y = (0.5 * fov_y).tan()
    proj_raw = get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)
    view = extrinsics.inverse().transpose(1, 2)
    full = view @ proj_raw
    return ViewSetup(view.contiguous(), full.contiguous(), proj_raw.contiguous(), tan_x, tan_y,
                     extrinsics[:, :3, 3].contiguous())
