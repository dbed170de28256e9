"""Host side of the HIP rasterizer: autograd binding over the C ABI and the
drop-in `diff_gaussian_rasterization` interface.

Mirrors the third-party module the reference imports at
src/model/decoder/cuda_splatting.py:5-8 and calls at :101-129:
`GaussianRasterizationSettings` (13 fields) and `GaussianRasterizer(settings)(
means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
cov3D_precomp, theta, rho)` -> `(image, radii, depth, opacity, n_touched)`.

PyTorch is plumbing here (device memory, the current HIP stream, autograd);
all rasterization runs in libgsr_hip.so.  There is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import NamedTuple, Optional

import torch
from torch import Tensor, nn

from . import _lib

# grow-only hint for the (tile, Gaussian) pair capacity, keyed by problem shape
_CAP_HINT: dict = {}
# longest per-tile list seen per problem shape (x1.25): picks the LDS budget of the per-tile sort
_MAX_TILE_HINT: dict = {}
# (pinned int32[GSR_STATUS_WORDS], event) for the status read-back of the forward, per host thread (threading.local: the
# buffers die with the thread and a recycled thread id can never pick up another thread's buffer) and per (device, stream)
# inside it, as a small LRU -- two forwards on different streams or threads of one device must not share a buffer (they would
# read each other's pair count / overflow flag), and short-lived side streams must not pin host memory for the life of the
# process.
_STATUS_LOCAL = threading.local()
_STATUS_LRU_MAX = 8


def _status_host(dev):
    from collections import OrderedDict
    cache = getattr(_STATUS_LOCAL, "cache", None)
    if cache is None:
        cache = _STATUS_LOCAL.cache = OrderedDict()
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    host = cache.get(key)
    if host is None:
        host = cache[key] = (torch.empty(_lib.GSR_STATUS_WORDS, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        while len(cache) > _STATUS_LRU_MAX:
            cache.popitem(last=False)
    else:
        cache.move_to_end(key)
    return host
# parity tests set KEEP_DEBUG to inspect the workspace (sorted lists, ranges, n_contrib) of the last forward
KEEP_DEBUG = False
LAST_DEBUG: dict = {}
LAST_STATS: dict = {}      # pairs (tile, Gaussian) R, views V and Gaussians per scene G of the most recent forward on this process
# bench.py sets PROFILE to a _lib.StageProfile to time every stage with hipEvents on the launch stream
PROFILE = None


def _ptr(t: Optional[Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def pack_views(viewmatrix: Tensor, projmatrix: Tensor, projmatrix_raw: Tensor, campos: Tensor, tanfovx: Tensor,
               tanfovy: Tensor, bg: Tensor, scale: Optional[Tensor] = None) -> Tensor:
    """Assemble the (V, 64) float32 GsrView array (include/gsr.h) on the inputs' device.
    All arguments carry a leading V dimension; nothing touches the host."""
    V = viewmatrix.shape[0]
    dev = viewmatrix.device
    out = torch.zeros((V, _lib.GSR_VIEW_FLOATS), dtype=torch.float32, device=dev)
    out[:, 0:16] = viewmatrix.reshape(V, 16)
    out[:, 16:32] = projmatrix.reshape(V, 16)
    out[:, 32:48] = projmatrix_raw.reshape(V, 16)
    out[:, 48:51] = campos.reshape(V, 3)
    out[:, 51] = tanfovx.reshape(V)
    out[:, 52] = tanfovy.reshape(V)
    out[:, 53:56] = bg.reshape(V, 3)
    out[:, 56] = 1.0 if scale is None else scale.reshape(V)
    return out


class RasterOutput(NamedTuple):
    image: Tensor      # (V,3,H,W)
    radii: Tensor      # (V,G) int32
    depth: Tensor      # (V,H,W)
    opacity: Tensor    # (V,H,W)
    n_touched: Tensor  # (V,G) int32 (an unwritten (1,1) placeholder unless requested)
    loss_mse: Optional[Tensor] = None   # scalar, only with `mse=`: weight * mean((image - target)^2), computed by the composite kernel


# persistent per-tile counters (GsrFused.tile_count), per (device, stream): zeroed once here, re-armed by every tile scan
_COUNTERS: dict = {}
# flags OR-ed into every call (tests / A-B tools: _lib.GSR_FLAG_BIN_BALLOT)
EXTRA_FLAGS = 0
# True: the tile scan stores the status words straight into pinned host memory (no copy kernel between the scan and the scatter);
# False: device status words + an asynchronous copy (the round-1..5 form, kept for A/B runs)
STATUS_DIRECT = True


def _tile_counters(dev, stream_handle: int, n: int) -> Tensor:
    key = (dev.index, stream_handle)
    buf = _COUNTERS.get(key)
    if buf is None or buf.numel() < n:
        buf = _COUNTERS[key] = torch.zeros(max(n, 1 << 14), dtype=torch.int32, device=dev)
    return buf


class RasterMse(NamedTuple):
    """LossMse (src/loss/loss_mse.py:22-31) fused into the composite kernels: target (V,3,H,W) ground truth, weight."""
    target: Tensor
    weight: float


class _Rasterize(torch.autograd.Function):
    """V = B*Vt views of B scenes in one launch sequence (include/gsr.h)."""

    @staticmethod
    def forward(ctx, means, cov6, opac, colors, views, means2D, theta, rho, H, W, Vt, sh_degree, use_sh, want_ntouched,
                mse_target, mse_weight):
        if not means.is_cuda:
            raise RuntimeError("styl3r_amd rasterizer needs tensors on an MI355X (HIP) device; there is no CPU path")
        lib = _lib.load()
        means = means.contiguous().float(); cov6 = cov6.contiguous().float(); opac = opac.contiguous().float()
        colors = colors.contiguous().float(); views = views.contiguous().float()
        B, G = means.shape[0], means.shape[1]
        V = views.shape[0]
        assert V == B * Vt, (V, B, Vt)
        cov9 = cov6.dim() == 4   # full (B,G,3,3) matrices: GSR_FLAG_COV9, no triu gather / scatter kernels
        assert (cov6.shape == (B, G, 3, 3) if cov9 else cov6.shape == (B, G, 6)) and opac.shape[:2] == (B, G)
        M = colors.shape[2] if use_sh else 0
        key = (B, Vt, G, H, W)
        # LDS budget of the per-tile sort from the longest list seen for this shape (+25 %): 1024 / 2048 / 4096 keys
        mt = _MAX_TILE_HINT.get(key, 4096)
        sort_sel = 1 if mt <= 1024 else (2 if mt <= 2048 else 0)
        # a backward will follow: the composite kernel zeroes the gradient accumulators on the side (GSR_FLAG_PREZERO_GRADS)
        prezero = any(ctx.needs_input_grad)
        flags = (_lib.GSR_FLAG_NTOUCHED if want_ntouched else 0) | (_lib.GSR_FLAG_COV9 if cov9 else 0) | \
                (sort_sel << _lib.GSR_FLAG_SORT_KEYS_SHIFT) | (_lib.GSR_FLAG_PREZERO_GRADS if prezero else 0) | EXTRA_FLAGS
        dims = _lib.GsrDims(B, Vt, G, H, W, M, sh_degree if use_sh else 0, flags,
                            PROFILE.handle if PROFILE is not None else None)
        dev = means.device
        image = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((V, H, W), dtype=torch.float32, device=dev)
        opacity = torch.empty((V, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((V, G), dtype=torch.int32, device=dev)
        # (not requested: an unwritten placeholder, no fill kernel; requested: the library zeroes it)
        n_touched = torch.empty((V, G) if want_ntouched else (1, 1), dtype=torch.int32, device=dev)
        stream_handle = torch.cuda.current_stream(dev).cuda_stream
        stream = C.c_void_p(stream_handle)
        cap = _CAP_HINT.get(key, max(4 * V * G, 1 << 16))
        T = ((H + 15) // 16) * ((W + 15) // 16)
        fx = _lib.GsrFused()
        fx.tile_count = _tile_counters(dev, stream_handle, V * T).data_ptr()
        loss = None
        if mse_target is not None:
            mse_target = mse_target.contiguous().float()
            assert mse_target.numel() == V * 3 * H * W and not mse_target.requires_grad, "fused MSE: a ground-truth target of the image's size"
            loss = torch.empty((), dtype=torch.float32, device=dev)
            fx.mse_target, fx.mse_weight, fx.mse_loss = mse_target.data_ptr(), float(mse_weight), loss.data_ptr()
        st, ev = _status_host(dev)
        status = st if STATUS_DIRECT else torch.empty(_lib.GSR_STATUS_WORDS, dtype=torch.int32, device=dev)   # fully written by the tile scan
        # Two-phase forward: preprocess + tile scan first; the pair count they produce is the only thing the host has
        # to see.  The scan stores it in pinned host memory and the render phase (scatter / sort / composite) is enqueued
        # OPTIMISTICALLY behind it, so the GPU never waits for the host; the host then spins on an event recorded behind the
        # scan (ready ~0.1 ms after launch, while the render phase is still running).  On overflow the render kernels have
        # exited early (they test the flag) and everything is re-issued with a larger capacity.
        def run(phase):
            dims.flags = flags | phase
            rc = lib.gsr_forward_fused(C.byref(dims), _ptr(views), _ptr(means), _ptr(cov6), _ptr(opac), _ptr(colors),
                                       cap, _ptr(ws), L.total, _ptr(image), _ptr(depth), _ptr(opacity), _ptr(radii),
                                       _ptr(n_touched) if want_ntouched else None, _ptr(status), C.byref(fx), stream)
            dims.flags = flags
            _lib.check(rc, "gsr_forward")

        try:
            while True:
                L = _lib.workspace_layout(dims, cap)
                ws = torch.empty(L.total, dtype=torch.uint8, device=dev)
                run(_lib.GSR_FLAG_PHASE_BIN)
                if not STATUS_DIRECT:
                    st.copy_(status, non_blocking=True)
                ev.record(torch.cuda.current_stream(dev))
                run(_lib.GSR_FLAG_PHASE_RENDER)
                while not ev.query():
                    pass
                R = (int(st[3]) << 32) | (int(st[0]) & 0xFFFFFFFF)
                if int(st[1]) == 0:
                    break
                if R > 0xFFFFFFFF:
                    raise RuntimeError(f"gsr_forward: {R} (tile, Gaussian) pairs exceed the 2^32 list limit")
                cap = int(R * 1.25) + 1024
        except BaseException:
            _COUNTERS.pop((dev.index, stream_handle), None)     # the counters may be left half-counted: never reuse them
            raise
        st = st.clone()
        _CAP_HINT[key] = max(_CAP_HINT.get(key, 0), min(int(R * 1.25) + 1024, 0xFFFFFFFF), 1 << 16)
        _MAX_TILE_HINT[key] = max(_MAX_TILE_HINT.get(key, 0), int(int(st[2]) * 1.25))
        ctx.dims, ctx.cap, ctx.ws_bytes = dims, cap, L.total
        ctx.want_tau = theta is not None or rho is not None
        ctx.want_m2d = means2D is not None and means2D.requires_grad
        ctx.has = (theta is not None, rho is not None, means2D is not None)
        ctx.mse_weight = float(mse_weight) if mse_target is not None else None
        ctx.save_for_backward(means, cov6, colors, views, ws)
        ctx.mse_target = mse_target       # (a ground-truth tensor without a graph: kept as the backward's "fused" switch)
        ctx.mark_non_differentiable(radii, n_touched)
        ctx.set_materialize_grads(False)   # unused outputs (depth, opacity, the image under a fused loss) arrive as None instead of zero-filled tensors
        ctx.num_pairs = R
        LAST_STATS.update(pairs=R, views=V, gaussians_per_scene=G, longest_tile_list=int(st[2]))     # what the last forward rendered (benchmarks assert on it)
        if KEEP_DEBUG:
            LAST_DEBUG.update(ws=ws, layout=L, dims=dims, cap=cap, num_pairs=R, status=st)
        return image, radii, depth, opacity, n_touched, loss

    @staticmethod
    def backward(ctx, g_image, g_radii, g_depth, g_opacity, g_ntouched, g_loss):
        lib = _lib.load()
        fused = ctx.mse_weight is not None
        means, cov6, colors, views, ws = ctx.saved_tensors
        dims = ctx.dims
        B, G, V = dims.B, dims.G, dims.B * dims.Vt
        dev = means.device
        fx = _lib.GsrFused()
        if fused and g_loss is not None:       # (a loss nobody differentiated: the plain backward)
            g_loss = g_loss.contiguous().float()
            fx.mse_target, fx.mse_weight, fx.mse_grad_loss = ctx.mse_target.data_ptr(), ctx.mse_weight, g_loss.data_ptr()
        else:
            fused = False
        if g_image is not None:
            g_image = g_image.contiguous().float()
        elif not fused:
            g_image = torch.zeros((V, 3, dims.H, dims.W), dtype=torch.float32, device=dev)
        g_depth = g_depth.contiguous().float() if g_depth is not None else None
        d_means = torch.empty_like(means); d_cov6 = torch.empty_like(cov6)
        d_opac = torch.empty((B, G), dtype=torch.float32, device=dev)
        d_colors = torch.empty_like(colors)
        d_m2d = torch.empty((V, G, 3), dtype=torch.float32, device=dev) if ctx.want_m2d else None
        d_tau = torch.empty((V, 6), dtype=torch.float32, device=dev) if ctx.want_tau else None
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.gsr_backward_fused(C.byref(dims), _ptr(views), _ptr(means), _ptr(cov6), _ptr(colors), ctx.cap, _ptr(ws),
                                    ctx.ws_bytes, _ptr(g_image), _ptr(g_depth), _ptr(d_means), _ptr(d_cov6), _ptr(d_opac),
                                    _ptr(d_colors), _ptr(d_m2d), _ptr(d_tau), C.byref(fx), stream)
        _lib.check(rc, "gsr_backward")
        dims.flags &= ~_lib.GSR_FLAG_PREZERO_GRADS     # the accumulators are dirty now: a second backward (retain_graph) zeroes them itself
        has_theta, has_rho, has_m2d = ctx.has
        g_theta = d_tau[:, 3:6] if (ctx.want_tau and has_theta) else None
        g_rho = d_tau[:, 0:3] if (ctx.want_tau and has_rho) else None
        return (d_means, d_cov6, d_opac, d_colors, None, d_m2d if has_m2d else None, g_theta, g_rho,
                None, None, None, None, None, None, None, None)


def rasterize_views(means: Tensor, cov6: Tensor, opacities: Tensor, colors: Tensor, views: Tensor, image_hw,
                    views_per_scene: int, sh_degree: int = 0, use_sh: bool = True, means2D: Optional[Tensor] = None,
                    theta: Optional[Tensor] = None, rho: Optional[Tensor] = None,
                    want_n_touched: bool = False, mse: Optional[RasterMse] = None) -> RasterOutput:
    """Batched entry point: means (B,G,3), cov6 (B,G,6) or full covariances (B,G,3,3), opacities (B,G), colors = SH (B,G,M,3) or RGB (B,G,3),
    views (B*Vt, 64) packed with `pack_views`; theta/rho (B*Vt, 3) receive the pose gradient.
    `mse`: also return LossMse of the image against a ground-truth target, computed inside the composite kernels (forward: in the epilogue
    of the compositing pass; backward: dL/dimage formed in the prologue) -- same value and gradients as `losses.mse_loss(out.image, target)`."""
    H, W = image_hw
    out = _Rasterize.apply(means, cov6, opacities, colors, views, means2D, theta, rho, int(H), int(W),
                           int(views_per_scene), int(sh_degree), bool(use_sh), bool(want_n_touched),
                           None if mse is None else mse.target, 1.0 if mse is None else float(mse.weight))
    return RasterOutput(*out)


# ---------------------------------------------------------------------------
# drop-in interface of `diff_gaussian_rasterization`
# ---------------------------------------------------------------------------
class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    projmatrix_raw: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


def _cov6_from_scale_rot(scales: Tensor, rotations: Tensor, scale_modifier: float) -> Tensor:
    """3DGS convention: quaternion (r,x,y,z), Sigma = R S S^T R^T, returned as xx,xy,xz,yy,yz,zz."""
    q = rotations / rotations.norm(dim=-1, keepdim=True)
    r, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Mm = R * (scale_modifier * scales)[:, None, :]
    S = Mm @ Mm.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


class GaussianRasterizer(nn.Module):
    """One view per call, exactly like the module the reference instantiates at cuda_splatting.py:116."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        s = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide exactly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        dev = means3D.device
        if cov3D_precomp is None:
            cov3D_precomp = _cov6_from_scale_rot(scales, rotations, float(s.scale_modifier))
        f32 = dict(dtype=torch.float32, device=dev)
        if isinstance(s.tanfovx, (int, float)) and isinstance(s.tanfovy, (int, float)):
            # the reference's call pattern (cuda_splatting.py:101-114): Python floats for the two tangents, device tensors for the rest.  One host
            # row [tanx, tany, 1 (scale), 0 x 7 (padding)] -> ONE host-to-device copy, and the 64-float view row is ONE concatenation
            # (round 5; before: two copies and pack_views' zero fill + eight slice assignments, ~10 tiny launches per view of a per-view loop)
            tail = torch.tensor([float(s.tanfovx), float(s.tanfovy), 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32).to(dev)
            views = torch.cat((s.viewmatrix.to(**f32).reshape(16), s.projmatrix.to(**f32).reshape(16), s.projmatrix_raw.to(**f32).reshape(16),
                               s.campos.to(**f32).reshape(3), tail[0:2], s.bg.to(**f32).reshape(3), tail[2:10]))[None]
        else:
            tanx = torch.as_tensor(s.tanfovx, **f32).reshape(1)
            tany = torch.as_tensor(s.tanfovy, **f32).reshape(1)
            views = pack_views(s.viewmatrix.to(**f32)[None], s.projmatrix.to(**f32)[None],
                               s.projmatrix_raw.to(**f32)[None], s.campos.to(**f32)[None], tanx, tany,
                               s.bg.to(**f32)[None])
        use_sh = shs is not None
        colors = shs if use_sh else colors_precomp
        th = theta.reshape(1, 3) if theta is not None else None
        rh = rho.reshape(1, 3) if rho is not None else None
        m2d = means2D[None] if means2D is not None else None
        out = rasterize_views(means3D[None], cov3D_precomp[None], opacities.reshape(1, -1), colors[None], views,
                              (int(s.image_height), int(s.image_width)), 1, int(s.sh_degree), use_sh, m2d, th, rh,
                              want_n_touched=True)
        return out.image[0], out.radii[0], out.depth, out.opacity, out.n_touched[0]
