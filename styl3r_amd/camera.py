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


# --------------------------------------------------------------------------- pixel-wise intrinsics embedding
def local_ray_directions(intrinsics: Tensor, h: int, w: int) -> Tensor:
    """(..., 3, 3) normalised intrinsics -> (..., h, w, 3) unit ray directions through the pixel centres in the camera frame
    (src/geometry/projection.py:117-151: `sample_image_grid` puts pixel (row i, column j) at ((j + 0.5) / w, (i + 0.5) / h),
    `get_local_rays` applies the inverse intrinsics to (x, y, 1) and normalises)."""
    dev, dt = intrinsics.device, intrinsics.dtype
    xs = (torch.arange(w, device=dev, dtype=dt) + 0.5) / w
    ys = (torch.arange(h, device=dev, dtype=dt) + 0.5) / h
    grid = torch.stack((xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w), torch.ones((h, w), device=dev, dtype=dt)), dim=-1)
    d = torch.einsum("...ij,hwj->...hwi", intrinsics.inverse(), grid)
    return d / d.norm(dim=-1, keepdim=True)


def real_sh(xyz: Tensor, degree: int) -> Tensor:
    """All real spherical harmonics up to `degree` of unit vectors (..., 3) -> (..., (degree + 1)^2), Y_n^m at index n (n + 1) + m with the
    Condon-Shortley phase -- the convention of the reference's generated tables (src/misc/sht.py:11-34, `rsh_cart_k`), evaluated here by
    the standard recurrences instead of per-degree polynomials:
        Y_n^0 = K_n^0 P_n(z);  Y_n^{+m} = sqrt 2 K_n^m Pbar_n^m(z) Re (x + i y)^m;  Y_n^{-m} = sqrt 2 K_n^m Pbar_n^m(z) Im (x + i y)^m
    with Pbar_n^m = P_n^m / sin^m(theta) (a polynomial in z): Pbar_m^m = (-1)^m (2m - 1)!!, Pbar_{m+1}^m = (2m + 1) z Pbar_m^m,
    (n - m) Pbar_n^m = (2n - 1) z Pbar_{n-1}^m - (n + m - 1) Pbar_{n-2}^m, and K_n^m = sqrt((2n + 1) / (4 pi) (n - m)! / (n + m)!)."""
    import math
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    out = [None] * ((degree + 1) ** 2)
    c, s = torch.ones_like(x), torch.zeros_like(x)          # Re / Im of (x + i y)^m
    pmm = torch.ones_like(x)                                 # Pbar_m^m
    for m in range(degree + 1):
        if m > 0:
            c, s = c * x - s * y, c * y + s * x
            pmm = pmm * (-(2 * m - 1))
        p_prev2, p_prev = None, pmm
        for n in range(m, degree + 1):
            if n == m:
                p = pmm
            elif n == m + 1:
                p = (2 * m + 1) * z * pmm
            else:
                p = ((2 * n - 1) * z * p_prev - (n + m - 1) * p_prev2) / (n - m)
            if n > m:
                p_prev2, p_prev = p_prev, p
            k = math.sqrt((2 * n + 1) / (4 * math.pi) * math.factorial(n - m) / math.factorial(n + m))
            if m == 0:
                out[n * (n + 1)] = k * p
            else:
                out[n * (n + 1) + m] = (math.sqrt(2.0) * k) * p * c
                out[n * (n + 1) - m] = (math.sqrt(2.0) * k) * p * s
    return torch.stack(out, dim=-1)


def intrinsic_embedding(context: dict, degree: int = 0, downsample: int = 1, merge_hw: bool = False) -> Tensor:
    """`get_intrinsic_embedding` (src/geometry/camera_emb.py:7-31): per-pixel camera-frame ray directions of every view (degree 0: 3 channels) or
    their real-SH expansion (degree 2 / 4 / 8: (degree + 1)^2 channels), (b, v, d, h, w) -- or (b, v, h w, d) with `merge_hw`."""
    assert degree in (0, 2, 4, 8)
    b, v, _, h, w = context["image"].shape
    d = local_ray_directions(context["intrinsics"], h // downsample, w // downsample)        # (b, v, h, w, 3)
    if degree > 0:
        d = real_sh(d, degree)
    return d.flatten(2, 3) if merge_hw else d.permute(0, 1, 4, 2, 3)
