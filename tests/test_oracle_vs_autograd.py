"""Second, independent pin of the (reference-unpinned) rasterizer oracle: a dense pure-PyTorch fp64 restatement of the
published 3DGS forward (cull, EWA covariance + 0.3, conic, ceil(3 sigma) radius and 16x16 tile
rectangles, depth order with id ties, front-to-back compositing with the alpha < 1/255 / power > 0 skips and the
T < 1e-4 termination, SH colours with the +0.5 clamp) whose gradients come from AUTOGRAD, not from hand-derived formulas.
The oracle's analytic backward (A4/A5) must agree with it for every input; so must its images and integer state."""
import numpy as np
import pytest
import torch

from oracle.gsr_oracle import Oracle
from tests.helpers import random_scene, simple_camera

C0 = 0.28209479177387814
C1 = 0.4886025119029199


def _sh_color(deg, shs, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * shs[:, 0]
    if deg >= 1:
        res = res - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    return res


def torch_render(means, cov6, opac, colors_or_shs, cam, bg, sh_degree, use_sh, tau=None):
    H, W = cam["H"], cam["W"]
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    view, proj, campos = T(cam["view"]), T(cam["proj"]), T(cam["campos"])
    if tau is not None:
        # left se(3) perturbation of the world-to-camera transform, T_cw' = exp(tau^) T_cw with tau = (rho, theta)
        # (src/misc/cam_utils.py:118-137); matrices are stored row-vector / transposed (cuda_splatting.py:86-88)
        rho, th = tau[:3], tau[3:]
        z = torch.zeros((), dtype=torch.float64)
        xi = torch.stack([torch.stack([z, -th[2], th[1], rho[0]]), torch.stack([th[2], z, -th[0], rho[1]]),
                          torch.stack([-th[1], th[0], z, rho[2]]), torch.stack([z, z, z, z])])
        Tcw = torch.matrix_exp(xi) @ view.T
        view = Tcw.T
        proj = view @ T(cam["proj_raw"])
        # `campos` is a separate, non-differentiated setting of the rasterizer (cuda_splatting.py:112): the pose gradient
        # holds it fixed, i.e. the SH view direction does not move with tau (oracle/gsr_oracle.c: "campos held fixed")
    G = means.shape[0]
    hom = torch.cat([means, torch.ones(G, 1, dtype=torch.float64)], 1)
    pv = hom @ view                      # row-vector convention (cuda_splatting.py:86-88)
    ph = hom @ proj
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :3] * pw[:, None]
    tz = pv[:, 2]
    fx, fy = W / (2 * cam["tanfovx"]), H / (2 * cam["tanfovy"])
    limx, limy = 1.3 * cam["tanfovx"], 1.3 * cam["tanfovy"]
    tx = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    z0 = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, z0, -fx * tx / tz ** 2], 1), torch.stack([z0, fy / tz, -fy * ty / tz ** 2], 1)], 1)   # (G,2,3)
    Wm = view[:3, :3].T                                                     # world -> view rotation acting on column vectors
    S = torch.stack([torch.stack([cov6[:, 0], cov6[:, 1], cov6[:, 2]], 1), torch.stack([cov6[:, 1], cov6[:, 3], cov6[:, 4]], 1),
                     torch.stack([cov6[:, 2], cov6[:, 4], cov6[:, 5]], 1)], 1)
    Tm = J @ Wm
    c2 = Tm @ S @ Tm.transpose(1, 2)
    a, b, c = c2[:, 0, 0] + 0.3, c2[:, 0, 1], c2[:, 1, 1] + 0.3
    det = a * c - b * b
    cA, cB, cC = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tmin_x = torch.clamp(((px - radius) / 16).detach().to(torch.int64), 0, gx); tmax_x = torch.clamp(((px + radius + 15) / 16).detach().to(torch.int64), 0, gx)
    tmin_y = torch.clamp(((py - radius) / 16).detach().to(torch.int64), 0, gy); tmax_y = torch.clamp(((py + radius + 15) / 16).detach().to(torch.int64), 0, gy)
    visible = (tz > 0.2) & (det != 0) & ((tmax_x - tmin_x) * (tmax_y - tmin_y) > 0)
    if use_sh:
        d = means - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp(_sh_color(sh_degree, colors_or_shs, d) + 0.5, min=0.0)
    else:
        rgb = colors_or_shs
    order = torch.argsort(tz.detach(), stable=True)                         # depth, ties by id
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    tile_x, tile_y = (xs / 16).long(), (ys / 16).long()
    Tacc = torch.ones(H, W, dtype=torch.float64); done = torch.zeros(H, W, dtype=torch.bool)
    Cimg = torch.zeros(3, H, W, dtype=torch.float64); Dimg = torch.zeros(H, W, dtype=torch.float64)
    ncontrib = torch.zeros(H, W, dtype=torch.int64); cnt = torch.zeros(H, W, dtype=torch.int64)
    for i in order.tolist():
        if not bool(visible[i]):
            continue
        in_tile = (tile_x >= tmin_x[i]) & (tile_x < tmax_x[i]) & (tile_y >= tmin_y[i]) & (tile_y < tmax_y[i])
        dx, dy = px[i] - xs, py[i] - ys
        power = -0.5 * (cA[i] * dx * dx + cC[i] * dy * dy) - cB[i] * dx * dy
        alpha = torch.clamp(opac[i] * torch.exp(power), max=0.99)
        cnt = cnt + (in_tile & ~done).long()                                 # position in the tile's list (contributor counter)
        act = in_tile & ~done & (power <= 0) & (alpha >= 1.0 / 255.0)
        testT = Tacc * (1 - alpha)
        stop = act & (testT < 1e-4)
        done = done | stop
        act = act & ~stop
        w = torch.where(act, alpha * Tacc, torch.zeros_like(alpha))
        Cimg = Cimg + rgb[i][:, None, None] * w[None]
        Dimg = Dimg + tz[i] * w
        Tacc = torch.where(act, testT, Tacc)
        ncontrib = torch.where(act, cnt, ncontrib)
    img = Cimg + Tacc[None] * torch.tensor(bg, dtype=torch.float64)[:, None, None]
    return img, Dimg, radius * visible, ncontrib


@pytest.mark.parametrize("use_sh,sh_degree,seed", [(False, 0, 3), (True, 1, 4), (True, 0, 5)])
def test_oracle_forward_and_analytic_gradients_match_autograd(use_sh, sh_degree, seed):
    cam = simple_camera(32, 48, c2w=np.array([[0.995, 0, 0.0998, 0.1], [0, 1, 0, -0.05], [-0.0998, 0, 0.995, 0.2], [0, 0, 0, 1]]))
    # full projection and camera centre recomputed in fp64 from (view, proj_raw), so that the tau = 0 perturbed camera of the
    # autograd path is bit-identical to the oracle's inputs (the helper's matrices are fp32 products)
    cam = dict(cam, proj=cam["view"] @ cam["proj_raw"], campos=np.linalg.inv(cam["view"].T)[:3, 3])
    G = 40
    means, cov6, opac, shs = random_scene(G, seed=seed, sh_degree=sh_degree, scale=(0.05, 0.25))
    rng = np.random.default_rng(seed)
    colors = rng.uniform(0, 1, (G, 3))
    wI = rng.normal(size=(3, cam["H"], cam["W"])); wD = rng.normal(size=(cam["H"], cam["W"])) * 0.3
    bg = (0.3, 0.5, 0.2)
    orc = Oracle("f64")
    st, ctx = orc.forward(means, cov6, opac, shs=shs if use_sh else None, colors=None if use_sh else colors, H=cam["H"], W=cam["W"],
                          tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, view=cam["view"], proj=cam["proj"],
                          proj_raw=cam["proj_raw"], campos=cam["campos"], sh_degree=sh_degree)
    g = orc.backward(st, ctx, wI, wD, want_tau=True)
    tm, tc, to = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (means, cov6, opac))
    tcol = torch.tensor(shs if use_sh else colors, dtype=torch.float64, requires_grad=True)
    tau = torch.zeros(6, dtype=torch.float64, requires_grad=True)
    img, dep, radii, ncontrib = torch_render(tm, tc, to, tcol, cam, bg, sh_degree, use_sh, tau=tau)
    # forward: images, depth, radii, n_contrib
    np.testing.assert_allclose(st.image, img.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(st.out_depth, dep.detach().numpy(), rtol=1e-9, atol=1e-11)
    assert np.array_equal(st.radii, radii.detach().numpy().astype(st.radii.dtype))
    assert np.array_equal(st.n_contrib, ncontrib.numpy().astype(st.n_contrib.dtype))
    ((img * torch.tensor(wI)).sum() + (dep * torch.tensor(wD)).sum()).backward()
    for name, ours, ref in (("means3D", g["means3D"], tm.grad), ("cov6", g["cov6"], tc.grad), ("opacities", g["opacities"], to.grad),
                            ("shs" if use_sh else "colors", g["shs"] if use_sh else g.get("colors", g["shs"]), tcol.grad)):
        ref = ref.numpy().reshape(np.asarray(ours).shape)
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(np.asarray(ours) - ref).max() <= 1e-7 * scale, (name, np.abs(np.asarray(ours) - ref).max(), scale)
    # pose gradient (MonoGS "-w-pose" extension): tau = (rho, theta)
    got_tau = np.concatenate([np.asarray(g["rho"]).reshape(3), np.asarray(g["theta"]).reshape(3)])
    ref_tau = tau.grad.numpy()
    assert np.abs(got_tau - ref_tau).max() <= 1e-7 * np.abs(ref_tau).max(), (got_tau, ref_tau)
