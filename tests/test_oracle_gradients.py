"""fp64 finite-difference checks of the oracle's analytic backward (A4/A5) and
the SH tables.  The fp64 oracle is the authority for gradients."""
import numpy as np
import pytest

from oracle.gsr_oracle import Oracle
from tests.helpers import random_scene, simple_camera


def _loss_and_grads(orc, means, cov6, opac, shs, cam, wI, wD, sh_degree, bg, want_tau=False, colors=None):
    st, ctx = orc.forward(means, cov6, opac, shs=shs, colors=colors, H=cam["H"], W=cam["W"], tanfovx=cam["tanfovx"],
                          tanfovy=cam["tanfovy"], bg=bg, view=cam["view"], proj=cam["proj"], proj_raw=cam["proj_raw"],
                          campos=cam["campos"], sh_degree=sh_degree)
    loss = float((st.image * wI).sum() + (st.out_depth * wD).sum())
    return loss, st, ctx


def _fd(f, x, idxs, eps):
    out = []
    for idx in idxs:
        xp = x.copy(); xp[idx] += eps
        xm = x.copy(); xm[idx] -= eps
        out.append((f(xp) - f(xm)) / (2 * eps))
    return np.array(out)


@pytest.mark.parametrize("sh_degree", [0, 1, 3, 4])
def test_fd_all_inputs(sh_degree):
    orc = Oracle("f64")
    cam = simple_camera(32, 48, c2w=np.array([[0.995, 0, 0.0998, 0.1], [0, 1, 0, -0.05], [-0.0998, 0, 0.995, 0.2], [0, 0, 0, 1]]))
    G = 24
    means, cov6, opac, shs = random_scene(G, seed=11 + sh_degree, sh_degree=sh_degree, scale=(0.05, 0.2))
    rng = np.random.default_rng(1)
    wI = rng.normal(size=(3, cam["H"], cam["W"]))
    wD = rng.normal(size=(cam["H"], cam["W"])) * 0.3
    bg = (0.3, 0.5, 0.2)
    loss, st, ctx = _loss_and_grads(orc, means, cov6, opac, shs, cam, wI, wD, sh_degree, bg)
    g = orc.backward(st, ctx, wI, wD)
    vis = np.nonzero(st.radii > 0)[0]
    assert len(vis) >= G // 2
    pick = vis[:8]

    def L(**kw):
        a = dict(means=means, cov6=cov6, opac=opac, shs=shs); a.update(kw)
        return _loss_and_grads(orc, a["means"], a["cov6"], a["opac"], a["shs"], cam, wI, wD, sh_degree, bg)[0]

    idx = [(i, k) for i in pick for k in range(3)]
    fd = _fd(lambda x: L(means=x), means, idx, 1e-6)
    an = np.array([g["means3D"][i] for i in idx])
    np.testing.assert_allclose(an, fd, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(fd).max()))

    idx = [(i, k) for i in pick for k in range(6)]
    fd = _fd(lambda x: L(cov6=x), cov6, idx, 1e-7)
    an = np.array([g["cov6"][i] for i in idx])
    np.testing.assert_allclose(an, fd, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(fd).max()))

    idx = [(i,) for i in pick]
    fd = _fd(lambda x: L(opac=x), opac, idx, 1e-6)
    an = np.array([g["opacities"][i] for i in idx])
    np.testing.assert_allclose(an, fd, rtol=2e-4, atol=1e-7 * max(1.0, np.abs(fd).max()))

    M = shs.shape[1]
    idx = [(i, k, c) for i in pick[:4] for k in range(M) for c in range(3)]
    fd = _fd(lambda x: L(shs=x), shs, idx, 1e-6)
    an = np.array([g["shs"][i] for i in idx])
    np.testing.assert_allclose(an, fd, rtol=2e-4, atol=1e-7 * max(1.0, np.abs(fd).max()))


def test_fd_precomputed_colors_and_clamp():
    orc = Oracle("f64")
    cam = simple_camera(32, 32)
    G = 12
    means, cov6, opac, shs = random_scene(G, seed=2, scale=(0.05, 0.2))
    cols = np.abs(shs[:, 0, :])
    rng = np.random.default_rng(3)
    wI = rng.normal(size=(3, 32, 32)); wD = np.zeros((32, 32))
    st, ctx = orc.forward(means, cov6, opac, colors=cols, H=32, W=32, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                          bg=(0, 0, 0), view=cam["view"], proj=cam["proj"], proj_raw=cam["proj_raw"], campos=cam["campos"])
    g = orc.backward(st, ctx, wI, wD)
    vis = np.nonzero(st.radii > 0)[0][:4]

    def L(x):
        s, _ = orc.forward(means, cov6, opac, colors=x, H=32, W=32, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                           bg=(0, 0, 0), view=cam["view"], proj=cam["proj"], proj_raw=cam["proj_raw"], campos=cam["campos"])
        return float((s.image * wI).sum())
    idx = [(i, c) for i in vis for c in range(3)]
    fd = _fd(L, cols, idx, 1e-6)
    np.testing.assert_allclose(np.array([g["shs"][i] for i in idx]), fd, rtol=1e-5, atol=1e-9)
    # SH colour clamped at 0 (rgb+0.5 < 0): zero gradient to that channel's coefficients
    shs2 = shs.copy(); shs2[:, 0, 0] = -5.0
    st, ctx = orc.forward(means, cov6, opac, shs=shs2, H=32, W=32, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                          bg=(0, 0, 0), view=cam["view"], proj=cam["proj"], proj_raw=cam["proj_raw"], campos=cam["campos"])
    assert st.clamped[vis, 0].all() and not st.clamped[vis, 1].any()
    g = orc.backward(st, ctx, wI, wD)
    assert np.all(g["shs"][:, :, 0] == 0) and np.any(g["shs"][:, :, 1] != 0)


def _so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def test_fd_pose_tau():
    """dL/d(rho,theta): left perturbation T_cw' = exp(tau) T_cw (cam_utils.py:118-137), campos fixed."""
    orc = Oracle("f64")
    c2w = np.array([[0.98, 0.02, 0.198, 0.2], [-0.0198, 0.9998, -0.003, 0.1], [-0.198, -0.001, 0.98, -0.1], [0, 0, 0, 1.0]])
    u, _, vt = np.linalg.svd(c2w[:3, :3]); c2w[:3, :3] = u @ vt
    cam = simple_camera(32, 48, c2w=c2w)
    # rebuild the matrices in float64 (simple_camera goes through fp32 torch)
    T_cw = np.linalg.inv(c2w)
    proj_raw = cam["proj_raw"]
    G = 20
    means, cov6, opac, shs = random_scene(G, seed=22, scale=(0.05, 0.2))
    rng = np.random.default_rng(5)
    wI = rng.normal(size=(3, 32, 48)); wD = rng.normal(size=(32, 48)) * 0.2

    def run(tau):
        Tm = np.eye(4); Tm[:3, :3] = _so3_exp(tau[3:]); Tm[:3, 3] = tau[:3]   # first order in rho is enough for FD
        Tn = Tm @ T_cw
        view = Tn.T
        proj = view @ proj_raw
        st, ctx = orc.forward(means, cov6, opac, shs=shs, H=32, W=48, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                              bg=(0.1, 0.2, 0.3), view=view, proj=proj, proj_raw=proj_raw, campos=cam["campos"])
        return float((st.image * wI).sum() + (st.out_depth * wD).sum()), st, ctx

    _, st, ctx = run(np.zeros(6))
    g = orc.backward(st, ctx, wI, wD, want_tau=True)
    an = np.concatenate([g["rho"], g["theta"]])
    fd = _fd(lambda t: run(t)[0], np.zeros(6), [(k,) for k in range(6)], 1e-7)
    np.testing.assert_allclose(an, fd, rtol=5e-4, atol=1e-6 * np.abs(fd).max())


def test_sh_tables_orthonormal():
    """All 25 real-SH basis functions (bands 0-4) are orthonormal on the sphere."""
    orc = Oracle("f64")
    n_t, n_p = 64, 128
    xg, wg = np.polynomial.legendre.leggauss(n_t)
    phi = (np.arange(n_p) + 0.5) * 2 * np.pi / n_p
    ct, ph = np.meshgrid(xg, phi, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    dirs = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    w = (wg[:, None] * np.ones(n_p)[None] * 2 * np.pi / n_p).reshape(-1)
    B = orc.sh_basis(4, dirs)
    gram = (B * w[:, None]).T @ B
    np.testing.assert_allclose(gram, np.eye(25), atol=1e-10)


def test_sh_basis_gradient_tables():
    orc = Oracle("f64")
    d = np.array([0.3, -0.5, 0.8])
    gx, gy, gz = orc.sh_basis_grad(4, d)
    eps = 1e-6
    for k, ga in enumerate((gx, gy, gz)):
        dp = d.copy(); dp[k] += eps
        dm = d.copy(); dm[k] -= eps
        fd = (orc.sh_basis(4, dp[None])[0] - orc.sh_basis(4, dm[None])[0]) / (2 * eps)
        np.testing.assert_allclose(ga, fd, rtol=1e-6, atol=1e-8)
