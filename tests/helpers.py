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


# ---- end-to-end (encoder -> decoder -> MSE) fixtures: tests/golden/make_e2e_fixtures.py and tests/test_e2e_parity.py ----------------
# Target statistics (mean, std per output channel) of the five 1x1 output convolutions after re-centring: xyz of the point heads
# (depth expm1(|xyz|) ~ 2..4, inside a tan(fov/2) = 0.58 frustum), (opacity logit, 3 log-scales, 4 quaternion) of the gs heads,
# SH DC of the appearance head.
from styl3r_amd.scenes import HEAD_TARGETS as E2E_HEAD_TARGETS      # (one source: the benchmarks re-centre random-init heads the same way)


def e2e_cameras(b=1):
    """Two target cameras per scene in the frame of context view 0 (the encoder is pose-free): the identity and a 4-degree
    turn about y with a small translation; RE10K-like normalised intrinsics; near 0.5 so that make_scale_invariant rescales by 2."""
    import math
    a = math.radians(4.0)
    c2w1 = torch.tensor([[math.cos(a), 0, math.sin(a), 0.25], [0, 1, 0, -0.08], [-math.sin(a), 0, math.cos(a), -0.15], [0, 0, 0, 1.0]])
    ext = torch.stack([torch.eye(4), c2w1])[None].repeat(b, 1, 1, 1)
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]])
    K2 = torch.tensor([[0.90, 0, 0.49], [0, 0.88, 0.51], [0, 0, 1.0]])
    intr = torch.stack([K, K2])[None].repeat(b, 1, 1, 1)
    return dict(extrinsics=ext, intrinsics=intr, near=torch.full((b, 2), 0.5), far=torch.full((b, 2), 100.0))


def closed_form_image(shape):
    """smooth target image in [0.1, 0.9] as a closed form of (view, channel, y, x): identical on the generator and the test side"""
    b, v, c, H, W = shape
    y = torch.arange(H, dtype=torch.float64)[:, None]
    x = torch.arange(W, dtype=torch.float64)[None, :]
    out = torch.empty(shape, dtype=torch.float64)
    for i in range(b):
        for j in range(v):
            for k in range(c):
                out[i, j, k] = 0.5 + 0.4 * torch.cos(0.045 * x * (1 + 0.3 * k) + 0.031 * y * (1 + 0.5 * j) + 0.9 * k + 0.4 * i)
    return out.float()


def e2e_fragile_mask(st, H, W, eps_alpha=5e-3, tie_rel=5e-6, term_rel=0.02, w_min=2e-5, eps_rect_px=4e-3, means=None, proj=None):
    """Pixels of an oracle render (oracle.gsr_oracle.FwdState) whose value is discontinuity-adjacent when the INPUT Gaussians move
    by what an fp32 evaluation of the encoder moves them (1e-6 .. 1e-5 relative): (i) a contributor whose alpha is within eps_alpha of the
    1/255 cut, (ii) two consecutive visible contributors whose depths differ by less than tie_rel (their order may swap), (iii) the
    termination test T < 1e-4 decided within term_rel -- each only where the flip could move the pixel by more than w_min; (iv) the
    tile rectangle: tile t is listed for a Gaussian iff 16 t + 1 <= p + r (A1.7: `(int)((p + r + 15) / 16)`), so a splat whose p + r
    sits within eps_rect_px of 16 t + 1 gains or loses the first pixel columns / rows of tile t, where its alpha can be far above 1/255
    -- including splats the golden run SKIPPED because their rectangle was empty (centre 2 .. 3 px outside the left / top image edge:
    `means` (G,3) and the row-vector `proj` (4,4) of the view let the mask find those).
    Used by the fixture generator only; the mask is stored with the fixture."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    frag = np.zeros((H, W), bool)
    thr = 1.0 / 255.0
    xy = st.xy.astype(np.float64); co = st.conic_opacity.astype(np.float64); depth = st.depth.astype(np.float64)
    for tile in range(gx * gy):
        s, e = int(st.ranges[tile][0]), int(st.ranges[tile][1])
        if e <= s:
            continue
        ids = st.point_list[s:e]
        tx, ty = tile % gx, tile // gx
        xs = np.arange(tx * 16, min(tx * 16 + 16, W)); ys = np.arange(ty * 16, min(ty * 16 + 16, H))
        px, py = np.meshgrid(xs, ys)
        px = px.reshape(-1).astype(np.float64); py = py.reshape(-1).astype(np.float64)
        dx = xy[ids, 0][:, None] - px[None]; dy = xy[ids, 1][:, None] - py[None]
        A, B, C, op = (co[ids, k][:, None] for k in range(4))
        power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
        alpha = np.minimum(0.99, op * np.exp(np.minimum(power, 0.0)))
        valid = (power <= 0) & (alpha >= thr)
        om = np.where(valid, 1.0 - alpha, 1.0)
        T_after = np.cumprod(om, 0)
        T_before = T_after / om
        stop = valid & (T_after < 1e-4)
        dead = np.cumsum(stop, 0) > 0                              # the terminating entry and everything behind it
        live = ~dead
        near_thr = (power <= 0) & (np.abs(alpha - thr) < eps_alpha * thr) & live & (T_before * thr > w_min)
        first_stop = stop & (np.cumsum(stop, 0) == 1)
        near_term = (valid & live & (np.abs(T_after - 1e-4) < term_rel * 1e-4)) | (first_stop & (np.abs(T_after - 1e-4) < term_rel * 1e-4))
        contrib = valid & live
        n = len(ids)
        d = depth[ids][:, None]
        idx = np.where(contrib, np.arange(n)[:, None], -1)
        prev = np.maximum.accumulate(idx, 0)
        prev = np.vstack([np.full((1, prev.shape[1]), -1), prev[:-1]])
        has_prev = prev >= 0
        pc = np.clip(prev, 0, None)
        d_prev = np.take_along_axis(np.broadcast_to(d, alpha.shape), pc, 0)
        a_prev = np.take_along_axis(alpha, pc, 0)
        T_prev = np.take_along_axis(T_before, pc, 0)
        tie = contrib & has_prev & ((d - d_prev) < tie_rel * d) & (T_prev * a_prev * alpha > w_min)
        bad = (near_thr | near_term | tie).any(0)
        frag[py.astype(int), px.astype(int)] |= bad
    # (iv) tile-rectangle membership
    vis = np.nonzero(st.radii > 0)[0]
    r = st.radii[vis].astype(np.float64)
    for axis, n_px, n_tiles in ((0, W, gx), (1, H, gy)):
        p = xy[vis, axis]
        hi = (p + r + 15.0) / 16.0
        k = np.rint(hi)
        near = (np.abs(hi - k) * 16.0 < eps_rect_px) & (k >= 1) & (k <= n_tiles)
        for g, kk in zip(vis[near], k[near].astype(int)):
            rr = float(st.radii[g])
            a0 = 16 * (kk - 1)                                     # first column (row) of the tile that toggles
            strip = np.arange(a0, min(a0 + 3, n_px))
            o = 1 - axis
            other = np.arange(max(int(np.floor(xy[g, o] - rr)), 0), min(int(np.ceil(xy[g, o] + rr)) + 1, H if o == 1 else W))
            if len(strip) == 0 or len(other) == 0:
                continue
            ss, oo = np.meshgrid(strip, other)
            dxx = xy[g, 0] - (ss if axis == 0 else oo); dyy = xy[g, 1] - (oo if axis == 0 else ss)
            pw = -0.5 * (co[g, 0] * dxx * dxx + co[g, 2] * dyy * dyy) - co[g, 1] * dxx * dyy
            hit = (pw <= 0) & (co[g, 3] * np.exp(np.minimum(pw, 0)) >= 0.5 * thr)
            xs_, ys_ = (ss, oo) if axis == 0 else (oo, ss)
            frag[ys_[hit], xs_[hit]] = True
    if means is not None:
        # splats without a rectangle in the golden run (radii == 0): off the left / top edge by about their radius.  Their conic is not
        # in the state; every splat of these scenes has radius 2 .. 4, so all three are tried and a (2 r + 1)-pixel run is excluded
        m = np.asarray(means, np.float64); P = np.asarray(proj, np.float64).reshape(4, 4)
        hom = np.concatenate([m, np.ones((len(m), 1))], 1) @ P
        pw = 1.0 / (hom[:, 3] + 1e-7)
        pp = np.stack([((hom[:, 0] * pw + 1.0) * W - 1.0) / 2.0, ((hom[:, 1] * pw + 1.0) * H - 1.0) / 2.0], 1)
        skipped = np.nonzero((st.radii == 0) & (hom[:, 3] > 0.2))[0]
        for axis, n_px in ((0, W), (1, H)):
            o = 1 - axis
            n_o = H if o == 1 else W
            for rr in (2.0, 3.0, 4.0):
                near = np.abs(pp[skipped, axis] + rr - 1.0) < eps_rect_px
                for g in skipped[near]:
                    c = pp[g, o]
                    if c < -rr or c > n_o - 1 + rr:
                        continue
                    other = np.arange(max(int(np.floor(c - rr)), 0), min(int(np.ceil(c + rr)) + 1, n_o))
                    for a in range(0, min(3, n_px)):
                        if axis == 0:
                            frag[other, a] = True
                        else:
                            frag[a, other] = True
    return frag
