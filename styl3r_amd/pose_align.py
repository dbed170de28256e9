"""Test-time pose alignment through the rasterizer's pose gradient.

Mirrors `ModelWrapperStyle.test_step_align` (src/model/model_wrapper_style.py:391-447) and
`update_pose` / `SE3_exp` / `SO3_exp` / `V` (src/misc/cam_utils.py:67-137): per target view two
3-vectors `cam_rot_delta`, `cam_trans_delta` stay at zero, receive dL/d(theta, rho) from the decoder
(`theta` / `rho` of the rasterizer), take an Adam step, and are folded into the camera as a LEFT
multiplication of the world->camera matrix, T_w2c' = exp(tau) T_w2c, then reset to zero.
Losses: the reference sums its configured losses (MSE + LPIPS); LPIPS weights are not available
offline, so the default here is MSE and any callable loss can be passed.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor


def _skew(x: Tensor) -> Tensor:
    z = torch.zeros((), dtype=x.dtype, device=x.device)
    return torch.stack([torch.stack([z, -x[2], x[1]]), torch.stack([x[2], z, -x[0]]), torch.stack([-x[1], x[0], z])])


def SO3_exp(theta: Tensor) -> Tensor:
    W = _skew(theta); W2 = W @ W
    angle = torch.norm(theta)
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    if angle < 1e-5:
        return I + W + 0.5 * W2
    return I + (torch.sin(angle) / angle) * W + ((1 - torch.cos(angle)) / (angle ** 2)) * W2


def V_mat(theta: Tensor) -> Tensor:
    W = _skew(theta); W2 = W @ W
    angle = torch.norm(theta)
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    if angle < 1e-5:
        return I + 0.5 * W + (1.0 / 6.0) * W2
    return I + W * ((1.0 - torch.cos(angle)) / (angle ** 2)) + W2 * ((angle - torch.sin(angle)) / (angle ** 3))


def SE3_exp(tau: Tensor) -> Tensor:
    rho, theta = tau[:3], tau[3:]
    T = torch.eye(4, device=tau.device, dtype=tau.dtype)
    T[:3, :3] = SO3_exp(theta)
    T[:3, 3] = V_mat(theta) @ rho
    return T


def update_pose(cam_trans_delta: Tensor, cam_rot_delta: Tensor, extrinsics: Tensor) -> Tensor:
    """(n,3), (n,3), c2w (n,4,4) -> new c2w  (cam_utils.py:118-137)."""
    tau = torch.cat([cam_trans_delta, cam_rot_delta], dim=-1)
    w2c = extrinsics.inverse()
    new = torch.stack([SE3_exp(tau[i]) @ w2c[i] for i in range(tau.shape[0])], dim=0)
    return new.inverse()


def align_poses(decoder, gaussians, target_image: Tensor, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                steps: int = 100, rot_lr: float = 0.005, trans_lr: float = 0.005,
                loss_fn: Optional[Callable[[Tensor, Tensor], Tensor]] = None):
    """extrinsics (b,v,4,4) initial target poses; returns (aligned extrinsics, list of per-step losses)."""
    b, v = extrinsics.shape[:2]
    h, w = target_image.shape[-2:]
    dev = extrinsics.device
    loss_fn = loss_fn or (lambda pred, tgt: ((pred - tgt) ** 2).mean())
    rot = torch.nn.Parameter(torch.zeros((b, v, 3), device=dev))
    trans = torch.nn.Parameter(torch.zeros((b, v, 3), device=dev))
    opt = torch.optim.Adam([{"params": [rot], "lr": rot_lr}, {"params": [trans], "lr": trans_lr}])
    extrinsics = extrinsics.clone()
    history = []
    for _ in range(steps):
        opt.zero_grad()
        out = decoder.forward(gaussians, extrinsics, intrinsics, near, far, (h, w), cam_rot_delta=rot, cam_trans_delta=trans)
        loss = loss_fn(out.color, target_image)
        loss.backward()
        history.append(float(loss.detach()))
        with torch.no_grad():
            opt.step()
            new = update_pose(trans.reshape(b * v, 3), rot.reshape(b * v, 3), extrinsics.reshape(b * v, 4, 4))
            rot.data.fill_(0); trans.data.fill_(0)
            extrinsics = new.reshape(b, v, 4, 4)
    return extrinsics, history
