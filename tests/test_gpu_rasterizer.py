"""-m gpu parity tests: the HIP rasterizer (through the C ABI / drop-in module)
against the CPU oracle on the same seeded inputs.
Bar (BASELINE.json north_star): bit-exact tile / sort indices (radii, tile
ranges, per-tile sorted id lists, n_contrib away from fp-fragile pixels),
<= 1e-4 rel on rendered RGB / depth / opacity and on every gradient."""
import numpy as np
import pytest
import torch

from styl3r_amd import rasterizer as rz
from tests.gpu_utils import assert_close_rel, hip_single_view, oracle_single_view, ws_view
from tests.helpers import random_scene, simple_camera

pytestmark = pytest.mark.gpu

CAM_C2W = np.array([[0.995, 0, 0.0998, 0.1], [0, 1, 0, -0.05], [-0.0998, 0, 0.995, 0.2], [0, 0, 0, 1]])


@pytest.fixture(autouse=True)
def _debug_on():
    rz.KEEP_DEBUG = True
    yield
    rz.KEEP_DEBUG = False
    rz.LAST_DEBUG.clear()


def _check_forward(means, cov6, opac, cam, shs=None, colors=None, bg=(0, 0, 0), sh_degree=0, min_ok=0.97):
    out = hip_single_view(means, cov6, opac, cam, shs=shs, colors=colors, bg=bg, sh_degree=sh_degree)
    orc, st, ctx = oracle_single_view("f32", means, cov6, opac, cam, shs=shs, colors=colors, bg=bg, sh_degree=sh_degree)
    H, W = cam["H"], cam["W"]
    G = len(opac)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    # ---- integer state: exact ----
    assert np.array_equal(out["radii"].cpu().numpy(), st.radii), "radii"
    assert rz.LAST_DEBUG["num_pairs"] == st.R, "R"
    off = ws_view("tile_offset", np.uint32, T + 1)
    nonempty = st.ranges[:, 1] > st.ranges[:, 0]
    assert np.array_equal(off[:-1][nonempty], st.ranges[nonempty, 0].astype(np.uint32)), "range starts"
    assert np.array_equal(np.diff(off.astype(np.int64)), st.ranges[:, 1] - st.ranges[:, 0]), "range lengths"
    pl = ws_view("point_list", np.uint32, max(st.R, 1))[:st.R]
    # bits 28..31 of a list word are the composite forward's quadrant mask for the backward (include/gsr.h GSR_ID_MASK)
    assert np.array_equal(pl & np.uint32(0x0FFFFFFF), st.point_list.astype(np.uint32)), "sorted (tile, id) list"
    recs = ws_view("records", np.float32, G * 12).reshape(G, 12)
    vis = st.radii > 0
    assert np.array_equal(recs[vis, 0:2], st.xy[vis]) and np.array_equal(recs[vis, 2], st.depth[vis]), "xy / depth bits"
    assert np.array_equal(recs[vis, 4:8], st.conic_opacity[vis]), "conic bits"
    assert np.array_equal(recs[vis, 8:11], st.rgb[vis]), "rgb bits"
    ok = st.fragile == 0
    nc = ws_view("n_contrib", np.uint32, H * W).reshape(H, W)
    assert np.array_equal(nc[ok], st.n_contrib[ok].astype(np.uint32)), "n_contrib"
    assert ok.mean() > min_ok   # the fp-fragile mask must not hide the comparison
    # ---- float outputs: <= 1e-4 rel ----
    assert_close_rel(out["image"].cpu().numpy()[:, ok], st.image[:, ok], 1e-4, "image")
    assert_close_rel(out["depth"].cpu().numpy()[0][ok], st.out_depth[ok], 1e-4, "depth")
    assert_close_rel(out["opacity"].cpu().numpy()[0][ok], st.out_opacity[ok], 1e-4, "opacity")
    fT = ws_view("final_T", np.float32, H * W).reshape(H, W)
    assert_close_rel(fT[ok], st.final_T[ok], 1e-4, "final_T")
    # n_touched: exact except Gaussians that touch a fragile pixel
    nt = out["n_touched"].cpu().numpy()
    diff = np.nonzero(nt != st.n_touched)[0]
    assert len(diff) <= max(2, int(0.002 * G)), f"n_touched differs for {len(diff)} Gaussians"
    return out, st


@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3, 4])
def test_forward_parity_sh_degrees(sh_degree):
    cam = simple_camera(96, 80, c2w=CAM_C2W)
    means, cov6, opac, shs = random_scene(3000, seed=40 + sh_degree, sh_degree=sh_degree)
    _check_forward(means, cov6, opac, cam, shs=shs, sh_degree=sh_degree, bg=(0.1, 0.3, 0.2))


def test_forward_parity_precomputed_colors_ragged_image():
    cam = simple_camera(50, 70)   # not multiples of 16: ragged edge tiles
    means, cov6, opac, shs = random_scene(1500, seed=7)
    _check_forward(means, cov6, opac, cam, colors=np.abs(shs[:, 0, :]))


def test_forward_empty_and_all_culled():
    cam = simple_camera(32, 32)
    means = np.array([[0, 0, -1.0], [0, 0, 0.1], [50.0, 0, 1.0]])   # behind, too near, far off-screen
    cov6 = np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4]), (3, 1))
    out, st = _check_forward(means, cov6, np.full(3, 0.5), cam, colors=np.ones((3, 3)), bg=(0.2, 0.4, 0.6))
    assert st.R == 0
    img = out["image"].cpu().numpy()
    np.testing.assert_allclose(img[0], 0.2); np.testing.assert_allclose(img[2], 0.6)
    assert torch.all(out["radii"] == 0)


def test_forward_known_answers_on_gpu():
    """the oracle's analytic pins, evaluated by the HIP path itself"""
    cam = simple_camera(32, 32)
    fx = 32 / (2 * cam["tanfovx"]); z = 4.0
    mean = np.array([[0.5 * z / fx, 0.5 * z / fx, z]])
    rgb = np.array([[0.8, 0.4, 0.2]]); bg = (0.1, 0.2, 0.3)
    out = hip_single_view(mean, np.array([[0.04, 0, 0, 0.04, 0, 0.04]]), np.array([0.7]), cam, colors=rgb, bg=bg)
    px = out["image"][:, 16, 16].cpu().numpy()
    np.testing.assert_allclose(px, rgb[0] * 0.7 + 0.3 * np.array(bg), rtol=2e-5)
    np.testing.assert_allclose(out["depth"][0, 16, 16].item(), 0.7 * z, rtol=2e-5)
    np.testing.assert_allclose(out["opacity"][0, 16, 16].item(), 0.7, rtol=2e-5)


def test_forward_oversize_tile_list_uses_global_sort():
    """> 4096 Gaussians in one tile: the LDS sort falls back to the in-place global network"""
    cam = simple_camera(32, 32)
    rng = np.random.default_rng(0)
    G = 6000
    z = rng.uniform(2, 9, G)
    means = np.stack([rng.uniform(-0.05, 0.05, G) * z, rng.uniform(-0.05, 0.05, G) * z, z], 1)
    cov6 = np.tile(np.array([4e-4, 0, 0, 4e-4, 0, 4e-4]), (G, 1)) * (z[:, None] ** 2)
    z[100:200] = z[100]            # equal depths: ties must resolve by id
    means[100:200, 2] = z[100]
    out, st = _check_forward(means, cov6, rng.uniform(0.01, 0.1, G), cam, colors=rng.uniform(0, 1, (G, 3)))
    assert (st.ranges[:, 1] - st.ranges[:, 0]).max() > 4096
    # a too-small LDS budget hint (GSR_FLAG_SORT_KEYS_*: 1024 keys) must still give the same lists: every tile between
    # 1025 and 6000 entries now takes the global-memory path
    old = rz._MAX_TILE_HINT.copy()
    try:
        for k in list(rz._MAX_TILE_HINT):
            rz._MAX_TILE_HINT[k] = 100
        rz._MAX_TILE_HINT[(1, 1, G, 32, 32)] = 100
        _check_forward(means, cov6, rng.uniform(0.01, 0.1, G), cam, colors=rng.uniform(0, 1, (G, 3)))
    finally:
        rz._MAX_TILE_HINT.clear(); rz._MAX_TILE_HINT.update(old)


@pytest.mark.parametrize("case", ["plane_plus_outliers", "all_equal_depth", "two_planes"])
def test_tile_sort_clustered_depths(case):
    """the per-tile sort buckets by the varying depth bits of the tile: clustered depths put most keys into a few buckets --
    buckets above RANK_MAX take the in-LDS network, a bucket above the LDS budget the global one; lists stay bit-identical to
    the oracle's stable radix order (ties by ascending id)"""
    cam = simple_camera(32, 32)
    rng = np.random.default_rng(7)
    G = {"plane_plus_outliers": 5200, "all_equal_depth": 4500, "two_planes": 3000}[case]
    if case == "plane_plus_outliers":      # a far plane with float-level jitter and a few near outliers stretching the range
        z = 6.0 + rng.integers(0, 40, G) * 4.8e-7
        z[:20] = rng.uniform(0.5, 1.0, 20)
    elif case == "all_equal_depth":        # range == 0: one bucket, order = id
        z = np.full(G, 4.0)
    else:                                   # two thin planes: two big buckets that fit LDS, plus duplicates inside them
        z = np.where(rng.uniform(size=G) < 0.5, 3.0, 7.0) + rng.integers(0, 8, G) * 4.8e-7
    z = z.astype(np.float32).astype(np.float64)
    means = np.stack([rng.uniform(-0.04, 0.04, G) * z, rng.uniform(-0.04, 0.04, G) * z, z], 1)
    cov6 = np.tile(np.array([4e-4, 0, 0, 4e-4, 0, 4e-4]), (G, 1)) * (z[:, None] ** 2)
    out, st = _check_forward(means, cov6, rng.uniform(0.01, 0.05, G), cam, colors=rng.uniform(0, 1, (G, 3)), min_ok=0.5)
    assert (st.ranges[:, 1] - st.ranges[:, 0]).max() > (4096 if case != "two_planes" else 1024)


def test_capacity_overflow_retry():
    cam = simple_camera(64, 64)
    means, cov6, opac, shs = random_scene(4000, seed=9, scale=(0.1, 0.3))
    key = (1, 1, 4000, 64, 64)
    rz._CAP_HINT[key] = 1 << 16
    # shrink the first-try capacity far below R to force the overflow path
    old = rz._CAP_HINT.copy()
    try:
        rz._CAP_HINT[key] = 64
        out, st = _check_forward(means, cov6, opac, cam, shs=shs)
        assert st.R > 64 and rz._CAP_HINT[key] >= st.R
    finally:
        rz._CAP_HINT.clear(); rz._CAP_HINT.update(old)


def _check_backward(means, cov6, opac, cam, shs=None, colors=None, bg=(0, 0, 0), sh_degree=0, pose=False, seed=0, f64_rel=2e-4):
    dev = torch.device("cuda:0")
    theta = torch.zeros(3, device=dev, requires_grad=True) if pose else None
    rho = torch.zeros(3, device=dev, requires_grad=True) if pose else None
    out = hip_single_view(means, cov6, opac, cam, shs=shs, colors=colors, bg=bg, sh_degree=sh_degree, theta=theta,
                          rho=rho, requires_grad=True)
    H, W = cam["H"], cam["W"]
    rng = np.random.default_rng(seed)
    wI = rng.normal(size=(3, H, W)).astype(np.float32); wD = (rng.normal(size=(H, W)) * 0.3).astype(np.float32)
    loss = (out["image"] * torch.tensor(wI, device=dev)).sum() + (out["depth"][0] * torch.tensor(wD, device=dev)).sum()
    loss.backward()
    res = {}
    for prec in ("f32", "f64"):
        orc, st, ctx = oracle_single_view(prec, means, cov6, opac, cam, shs=shs, colors=colors, bg=bg, sh_degree=sh_degree)
        res[prec] = orc.backward(st, ctx, wI, wD, want_tau=pose)
    t = out["inputs"]
    got = dict(means3D=t["means"].grad, cov6=t["cov6"].grad, opacities=t["opac"].grad.reshape(-1),
               shs=t["colors"].grad, means2D=out["means2D"].grad)
    for k, v in got.items():
        v = v.cpu().numpy()
        assert np.isfinite(v).all(), k
        # the fp32 oracle restates the GPU arithmetic; the fp64 oracle is the gradient authority
        assert_close_rel(v, res["f32"][k], 1e-4, f"d{k} vs f32 oracle")
        assert_close_rel(v, res["f64"][k], f64_rel, f"d{k} vs f64 oracle")
    if pose:
        # pose gradients are sums over all Gaussians (order differs: atomics): 5e-4 against the fp32 oracle
        assert_close_rel(rho.grad.cpu().numpy(), res["f32"]["rho"], 5e-4, "d rho vs f32 oracle")
        assert_close_rel(theta.grad.cpu().numpy(), res["f32"]["theta"], 5e-4, "d theta vs f32 oracle")
        assert_close_rel(rho.grad.cpu().numpy(), res["f64"]["rho"], max(2e-4, f64_rel), "d rho")
        assert_close_rel(theta.grad.cpu().numpy(), res["f64"]["theta"], max(2e-4, f64_rel), "d theta")


@pytest.mark.parametrize("sh_degree", [0, 2, 4])
def test_backward_parity(sh_degree):
    cam = simple_camera(64, 80, c2w=CAM_C2W)
    means, cov6, opac, shs = random_scene(1200, seed=60 + sh_degree, sh_degree=sh_degree, scale=(0.03, 0.15))
    _check_backward(means, cov6, opac, cam, shs=shs, sh_degree=sh_degree, bg=(0.3, 0.5, 0.2))


def test_backward_parity_colors_precomp_and_pose():
    cam = simple_camera(48, 48, c2w=CAM_C2W)
    means, cov6, opac, shs = random_scene(600, seed=77, scale=(0.03, 0.15))
    _check_backward(means, cov6, opac, cam, colors=np.abs(shs[:, 0, :]), bg=(0.1, 0.1, 0.4), pose=True)
    _check_backward(means, cov6, opac, cam, shs=shs, pose=True)


def test_rejects_cpu_tensors_and_bad_args():
    cam = simple_camera(32, 32)
    s = rz.GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4),
                                         0, torch.zeros(3), False, False)
    r = rz.GaussianRasterizer(s)
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), colors_precomp=torch.ones(4, 3), cov3D_precomp=torch.ones(4, 6))
    with pytest.raises(Exception):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), cov3D_precomp=torch.ones(4, 6))


def test_batched_decoder_matches_per_view_dropin_and_oracle():
    """DecoderSplattingHIP (b=2 scenes x v=3 views, one launch sequence, scale-invariant folding) ==
    the reference's per-view loop semantics (cuda_splatting.py:93-132) run through the drop-in module."""
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder, prepare_views
    from styl3r_amd.scenes import make_scene
    dev = torch.device("cuda:0")
    scs = [make_scene(n_ctx=2, grid_hw=(48, 48), n_views=3, image_hw=(64, 64), sh_degree=1, seed=100 + i) for i in range(2)]
    st = lambda n: torch.stack([getattr(s, n) for s in scs]).to(dev)
    g = Gaussians(st("means").requires_grad_(True), st("covariances").requires_grad_(True),
                  st("harmonics").requires_grad_(True), st("opacities").requires_grad_(True))
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.2, 0.1, 0.3], True)).to(dev)
    dec.torch_view_setup = True     # same cameras, bit for bit, as the per-view path below
    delta_r = torch.zeros(2, 3, 3, device=dev, requires_grad=True)
    delta_t = torch.zeros(2, 3, 3, device=dev, requires_grad=True)
    out = dec.forward(g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (64, 64),
                      cam_rot_delta=delta_r, cam_trans_delta=delta_t)
    assert out.color.shape == (2, 3, 3, 64, 64) and out.depth.shape == (2, 3, 64, 64)
    w = torch.rand(2, 3, 3, 64, 64, device=dev, generator=torch.Generator(dev).manual_seed(1))
    (out.color * w).sum().backward()
    grads = [t.grad.clone() for t in (g.means, g.covariances, g.harmonics, g.opacities, delta_r, delta_t)]

    # per-view path, replicating render_cuda's pre-scaling on the host side
    g2 = [t.detach().clone().requires_grad_(True) for t in (g.means, g.covariances, g.harmonics, g.opacities)]
    dr2 = torch.zeros(2, 3, 3, device=dev, requires_grad=True); dt2 = torch.zeros(2, 3, 3, device=dev, requires_grad=True)
    imgs = []
    for b in range(2):
        views = prepare_views(scs[b].extrinsics.to(dev), scs[b].intrinsics.to(dev), scs[b].near.to(dev), scs[b].far.to(dev),
                              torch.tensor([[0.2, 0.1, 0.3]], device=dev).expand(3, 3), True)
        for v in range(3):
            row = views[v]
            s = row[56]
            settings = rz.GaussianRasterizationSettings(64, 64, float(row[51]), float(row[52]), row[53:56], 1.0,
                                                        row[0:16].reshape(4, 4), row[16:32].reshape(4, 4),
                                                        row[32:48].reshape(4, 4), 1, row[48:51], False, False)
            cov = g2[1][b] * (s ** 2)
            r_i, c_i = torch.triu_indices(3, 3)
            image, radii, depth, opacity, n_touched = rz.GaussianRasterizer(settings)(
                means3D=g2[0][b] * s, means2D=torch.zeros_like(g2[0][b]), shs=g2[2][b].permute(0, 2, 1).contiguous(),
                opacities=g2[3][b][:, None], cov3D_precomp=cov[:, r_i, c_i], theta=dr2[b, v], rho=dt2[b, v])
            imgs.append(image)
    ref = torch.stack(imgs).reshape(2, 3, 3, 64, 64)
    assert_close_rel(out.color.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-6, "batched vs per-view image")
    (ref * w).sum().backward()
    for a, b_, name in zip(grads, [t.grad for t in g2] + [dr2.grad, dt2.grad],
                           ["means", "cov", "sh", "opac", "theta", "rho"]):
        assert_close_rel(a.cpu().numpy(), b_.cpu().numpy(), 2e-5, f"batched vs per-view d{name}")
    # default camera path (one gsr_build_views kernel): same image to the path's fp32 noise
    dec.torch_view_setup = False
    out2 = dec.forward(g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (64, 64))
    assert_close_rel(out2.color.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-4, "hip view setup vs torch view setup")


def test_full_size_workload_parity():
    """BASELINE config at full size (G = 65 536, 256x256): one view against the oracle."""
    from styl3r_amd.decoder import prepare_views
    from styl3r_amd.scenes import make_scene
    sc = make_scene(n_ctx=1, grid_hw=(256, 256), n_views=2, image_hw=(256, 256), sh_degree=0, seed=1234)
    views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(2, 3), True).numpy()
    row = views[1]; s = np.float32(row[56])
    cov = sc.covariances.numpy()
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1) * (s * s)
    cam = dict(H=256, W=256, tanfovx=row[51], tanfovy=row[52], view=row[0:16].reshape(4, 4), proj=row[16:32].reshape(4, 4),
               proj_raw=row[32:48].reshape(4, 4), campos=row[48:51])
    _check_forward(sc.means.numpy() * s, cov6, sc.opacities.numpy(), cam, shs=sc.harmonics.numpy().transpose(0, 2, 1))


def _scene_view_cam(sc, views, i):
    row = views[i]; s = np.float32(row[56])
    cov = sc.covariances.numpy()
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1) * (s * s)
    H, W = sc.image_shape
    cam = dict(H=H, W=W, tanfovx=row[51], tanfovy=row[52], view=row[0:16].reshape(4, 4), proj=row[16:32].reshape(4, 4),
               proj_raw=row[32:48].reshape(4, 4), campos=row[48:51])
    return s, cov6, cam


def test_full_size_workload_backward_parity_single_view():
    """the BACKWARD at the headline size (G = 65 536, 256 x 256; lists of ~630 entries, multi-batch
    back-to-front walk) against the f32 and f64 oracles, every gradient, depth gradient included (k_composite_bwd<true>)."""
    from styl3r_amd.decoder import prepare_views
    from styl3r_amd.scenes import make_scene
    sc = make_scene(n_ctx=1, grid_hw=(256, 256), n_views=2, image_hw=(256, 256), sh_degree=0, seed=1234)
    views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(2, 3), True).numpy()
    s, cov6, cam = _scene_view_cam(sc, views, 1)
    _check_backward(sc.means.numpy() * s, cov6, sc.opacities.numpy(), cam, shs=sc.harmonics.numpy().transpose(0, 2, 1), seed=5,
                    f64_rel=1e-3)   # parity bar = 1e-4 vs the f32 oracle; fp32 arithmetic itself (oracle and GPU alike) sits ~5e-4 from fp64 here:
    # single alpha >= 1/255 decisions differ between the two precisions on 630-entry lists


def test_c4_size_view_backward_parity():
    """one view of the C4 workload: 4 context views x 256^2 = 262 144 Gaussians (lists of ~2 250, up to ~4 700 entries: beyond the
    tile sort's LDS budget), forward integer state + images and every gradient against the oracles"""
    from styl3r_amd.decoder import prepare_views
    from styl3r_amd.scenes import make_scene
    sc = make_scene(n_ctx=4, grid_hw=(256, 256), n_views=2, image_hw=(256, 256), sh_degree=0, seed=4321)
    views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(2, 3), True).numpy()
    s, cov6, cam = _scene_view_cam(sc, views, 0)
    means, opac, shs = sc.means.numpy() * s, sc.opacities.numpy(), sc.harmonics.numpy().transpose(0, 2, 1)
    _check_forward(means, cov6, opac, cam, shs=shs, min_ok=0.9)
    _check_backward(means, cov6, opac, cam, shs=shs, seed=6, f64_rel=1e-2)     # 1e-4 vs the f32 oracle is the bar; fp64 is a sanity bound on 2 250-entry lists


def test_headline_workload_backward_parity_through_the_decoder():
    """The bench's own path at the bench's own size: 2 scenes x 4 target views, G = 65 536, 256 x 256, through DecoderSplattingHIP
    -- k_composite_fwd<false> / k_composite_bwd<false> (no n_touched, no depth gradient), LPT tile order, accumulators pre-zeroed
    by the forward, Gaussians shared by the 4 views of a scene -- image-only loss; the gradients of scene 1 (the second scene: batch
    indexing) against the oracle's, summed over its 4 views (decoder_splatting_cuda.py:37-68, cuda_splatting.py:46-133)."""
    from oracle.gsr_oracle import Oracle
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder, prepare_views
    from styl3r_amd.scenes import make_scene
    dev = torch.device("cuda:0")
    scs = [make_scene(n_ctx=1, grid_hw=(256, 256), n_views=4, image_hw=(256, 256), sh_degree=0, seed=1234 + i) for i in range(2)]
    st = lambda n: torch.stack([getattr(sc, n) for sc in scs]).to(dev)
    g = Gaussians(*(st(n).requires_grad_(True) for n in ("means", "covariances", "harmonics", "opacities")))
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    dec.torch_view_setup = True      # cameras bit-identical to prepare_views below (the one-kernel set-up is a few ulp apart: its own test)
    out = dec.forward(g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (256, 256))
    rng = np.random.default_rng(9)
    wI = rng.normal(size=(2, 4, 3, 256, 256)).astype(np.float32)
    (out.color * torch.tensor(wI, device=dev)).sum().backward()
    sc = scs[1]
    # the cameras exactly as the decoder built them: the same torch op sequence ON THE SAME DEVICE (a CPU `inverse()` is a few ulp away,
    # enough to flip single alpha >= 1/255 decisions on 630-entry lists, each worth up to ~3e-3 of the largest gradient)
    views = prepare_views(sc.extrinsics.to(dev), sc.intrinsics.to(dev), sc.near.to(dev), sc.far.to(dev), torch.zeros(4, 3, device=dev), True).cpu().numpy()
    G = sc.means.shape[0]
    for prec, bar in (("f32", 1e-4), ("f64", 1e-2)):      # parity bar: the f32 oracle; fp64 only bounds the fp32 arithmetic itself (threshold flips)
        orc = Oracle(prec)
        acc = dict(means=np.zeros((G, 3)), cov=np.zeros((G, 3, 3)), sh=np.zeros((G, 3, 1)), opac=np.zeros(G))
        for v in range(4):
            s, cov6, cam = _scene_view_cam(sc, views, v)
            stt, ctx = orc.forward(np.float32(sc.means.numpy() * s), np.float32(cov6), sc.opacities.numpy(), shs=sc.harmonics.numpy().transpose(0, 2, 1),
                                   H=256, W=256, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=(0, 0, 0), view=cam["view"], proj=cam["proj"],
                                   proj_raw=cam["proj_raw"], campos=cam["campos"], nthreads=8)
            if prec == "f32":
                ok = stt.fragile == 0
                assert_close_rel(out.color[1, v].detach().cpu().numpy()[:, ok], stt.image[:, ok], 1e-4, f"view {v} image")
            gr = orc.backward(stt, ctx, wI[1, v], None, nthreads=8)
            acc["means"] += gr["means3D"] * s
            r, c = np.triu_indices(3)
            acc["cov"][:, r, c] += gr["cov6"] * (s * s)                     # cov3D_precomp = covariances[:, row, col]: upper triangle only
            acc["sh"] += gr["shs"].transpose(0, 2, 1)
            acc["opac"] += gr["opacities"]
        for name, t in (("means", g.means), ("cov", g.covariances), ("sh", g.harmonics), ("opac", g.opacities)):
            assert_close_rel(t.grad[1].cpu().numpy(), acc[name], bar, f"decoder path d{name} vs {prec} oracle (4 views summed)")


def test_build_views_kernel_matches_torch_view_setup():
    """gsr_build_views (one kernel) == prepare_views (the reference's torch op sequence) to a few ulp"""
    from styl3r_amd.decoder import build_views_hip, prepare_views
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n = 37
    A = torch.randn(n, 3, 3, generator=g); Q, _ = torch.linalg.qr(A)
    ext = torch.eye(4).repeat(n, 1, 1); ext[:, :3, :3] = Q; ext[:, :3, 3] = torch.randn(n, 3, generator=g) * 2
    K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).repeat(n, 1, 1) + 0.05 * torch.rand(n, 3, 3, generator=g)
    K[:, 2] = torch.tensor([0, 0, 1.0])
    near = 0.05 + torch.rand(n, generator=g); far = 20 + 100 * torch.rand(n, generator=g)
    bg = torch.rand(n, 3, generator=g)
    for si in (True, False):
        ref = prepare_views(ext, K, near, far, bg, si).numpy()
        got = build_views_hip(ext.to(dev), K.to(dev), near.to(dev), far.to(dev), bg.to(dev), si).cpu().numpy()
        np.testing.assert_allclose(got[:, :57], ref[:, :57], rtol=2e-6, atol=2e-6)


def test_c5_stress_shapes_parity():
    """BASELINE config 5 shapes: 512x512 (1024 tiles), sh_degree 4 (d_sh = 25), 4 context views' worth of Gaussians
    subsampled to 4 x 128 x 128 = 65 536 so the oracle finishes in seconds; one view, forward + backward."""
    from styl3r_amd.decoder import prepare_views
    from styl3r_amd.scenes import make_scene
    sc = make_scene(n_ctx=4, grid_hw=(128, 128), n_views=2, image_hw=(512, 512), sh_degree=4, seed=77)
    views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(2, 3), True).numpy()
    row = views[0]; s = np.float32(row[56])
    cov = sc.covariances.numpy()
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1) * (s * s)
    cam = dict(H=512, W=512, tanfovx=row[51], tanfovy=row[52], view=row[0:16].reshape(4, 4), proj=row[16:32].reshape(4, 4),
               proj_raw=row[32:48].reshape(4, 4), campos=row[48:51])
    means = sc.means.numpy() * s
    shs = sc.harmonics.numpy().transpose(0, 2, 1)
    _check_forward(means, cov6, sc.opacities.numpy(), cam, shs=shs, sh_degree=4)
    # fp32 itself (GPU and the fp32 oracle alike) sits ~5e-4 from fp64 on this 512^2 / degree-4 case
    _check_backward(means[::8], cov6[::8], sc.opacities.numpy()[::8], cam, shs=shs[::8], sh_degree=4, f64_rel=2e-3)


def test_orthographic_renderer_matches_oracle():
    """render_cuda_orthographic's camera construction (cuda_splatting.py:136-196) through the HIP path vs the oracle"""
    from oracle.gsr_oracle import Oracle
    from styl3r_amd.camera import get_projection_matrix
    from styl3r_amd.decoder import Gaussians, render_hip_orthographic
    from styl3r_amd.scenes import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(n_ctx=1, grid_hw=(64, 64), n_views=1, image_hw=(64, 64), seed=9)
    g = Gaussians(sc.means[None].to(dev), sc.covariances[None].to(dev), sc.harmonics[None].to(dev), sc.opacities[None].to(dev))
    ext = torch.eye(4)[None]; ext[0, 2, 3] = -1.0
    width = torch.tensor([6.0]); height = torch.tensor([6.0]); near = torch.tensor([0.5]); far = torch.tensor([20.0])
    dump = {}
    img = render_hip_orthographic(ext.to(dev), width.to(dev), height.to(dev), near.to(dev), far.to(dev), (64, 64),
                                  torch.zeros(1, 3, device=dev), g, 1, dump=dump)
    assert img.shape == (1, 3, 64, 64) and img.abs().sum() > 0
    # oracle with the dumped camera
    e = dump["extrinsics"][0].cpu(); n2, f2 = dump["near"].cpu(), dump["far"].cpu()
    fx, fy = dump["fov_x"].cpu(), dump["fov_y"].cpu()
    P = get_projection_matrix(n2, f2, fx.expand(1), fy)[0].T
    V = e.inverse().T
    cov = sc.covariances.numpy()
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
    st, _ = Oracle("f32").forward(sc.means.numpy(), cov6, sc.opacities.numpy(), shs=sc.harmonics.numpy().transpose(0, 2, 1), H=64, W=64,
                                  tanfovx=float((0.5 * fx).tan()), tanfovy=float((0.5 * fy).tan()[0]), bg=(0, 0, 0), view=V.numpy(),
                                  proj=(V @ P).numpy(), proj_raw=P.numpy(), campos=e[:3, 3].numpy())
    ok = st.fragile == 0
    assert_close_rel(img[0].cpu().numpy()[:, ok], st.image[:, ok], 2e-4, "orthographic image")


def test_single_call_and_two_phase_forward_are_identical():
    """include/gsr.h: gsr_forward without phase flags == GSR_FLAG_PHASE_BIN followed by GSR_FLAG_PHASE_RENDER (what
    rasterizer.py issues), bit for bit, outputs and workspace; mixing both flags is rejected."""
    import ctypes as C
    from styl3r_amd import _lib
    from styl3r_amd.scenes import make_scene
    lib = _lib.load()
    dev = torch.device("cuda:0")
    sc = make_scene(n_ctx=1, grid_hw=(64, 64), n_views=2, image_hw=(80, 112), seed=5)
    from styl3r_amd.decoder import build_views_hip
    means, cov, har, op = (t.to(dev)[None].contiguous() for t in (sc.means, sc.covariances, sc.harmonics, sc.opacities))
    views = build_views_hip(sc.extrinsics.to(dev), sc.intrinsics.to(dev), sc.near.to(dev), sc.far.to(dev),
                            torch.zeros(3, device=dev), False)
    B, G, V, H, W = 1, means.shape[1], 2, 80, 112
    shs = har.permute(0, 1, 3, 2).contiguous()
    flags = _lib.GSR_FLAG_COV9
    cap = 1 << 20
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    results = []
    for phases in ((0,), (_lib.GSR_FLAG_PHASE_BIN, _lib.GSR_FLAG_PHASE_RENDER)):
        dims = _lib.GsrDims(B, V, G, H, W, shs.shape[2], 0, flags, None)
        L = _lib.workspace_layout(dims, cap)
        ws = torch.zeros(L.total, dtype=torch.uint8, device=dev)
        img = torch.zeros((V, 3, H, W), device=dev); dep = torch.zeros((V, H, W), device=dev); opa = torch.zeros((V, H, W), device=dev)
        radii = torch.zeros((V, G), dtype=torch.int32, device=dev); status = torch.zeros(8, dtype=torch.int32, device=dev)
        for ph in phases:
            dims.flags = flags | ph
            rc = lib.gsr_forward(C.byref(dims), views.data_ptr(), means.data_ptr(), cov.data_ptr(), op.data_ptr(), shs.data_ptr(),
                                 cap, ws.data_ptr(), L.total, img.data_ptr(), dep.data_ptr(), opa.data_ptr(), radii.data_ptr(),
                                 None, status.data_ptr(), stream)
            assert rc == 0
            if ph == _lib.GSR_FLAG_PHASE_BIN:
                st = status.cpu()
                assert int(st[1]) == 0 and int(st[0]) > 0          # status is final after the bin phase
                assert float(img.abs().sum()) == 0.0               # nothing rendered yet
        torch.cuda.synchronize()
        results.append((img, dep, opa, radii, status.cpu(), ws[L.final_T:L.final_T + 4 * V * H * W].clone(),
                        ws[L.n_contrib:L.n_contrib + 4 * V * H * W].clone(), ws[L.point_list:L.point_list + 4 * int(status[0])].clone()))
    for a, b in zip(*results):
        assert torch.equal(a, b)
    assert float(results[0][0].abs().sum()) > 0
    dims.flags = flags | _lib.GSR_FLAG_PHASE_BIN | _lib.GSR_FLAG_PHASE_RENDER
    assert lib.gsr_forward(C.byref(dims), views.data_ptr(), means.data_ptr(), cov.data_ptr(), op.data_ptr(), shs.data_ptr(),
                           cap, ws.data_ptr(), L.total, img.data_ptr(), dep.data_ptr(), opa.data_ptr(), radii.data_ptr(),
                           None, status.data_ptr(), stream) == -1


@pytest.mark.parametrize("seed", list(range(40)))
def test_randomized_sweep_forward_backward_parity(seed):
    """seeded sweep over image shapes (ragged, tiny, wide), Gaussian counts, SH degrees, splat sizes (sub-pixel to
    screen-filling), opacities (near 1/255 to 0.99), backgrounds, off-centre / rotated cameras, with and without pose
    gradients: forward integer state + images and every gradient against the oracle"""
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.choice([16, 17, 33, 48, 64, 95])), int(rng.choice([16, 31, 40, 64, 80, 130]))
    G = int(rng.choice([1, 7, 64, 500, 3000]))
    deg = int(rng.choice([0, 0, 1, 2, 3, 4]))
    use_sh = bool(rng.integers(0, 2)) or deg > 0
    scale = [(0.001, 0.01), (0.02, 0.12), (0.1, 0.6), (0.5, 3.0)][int(rng.integers(0, 4))]
    op_range = [(0.004, 0.02), (0.2, 0.95), (0.9, 1.0)][int(rng.integers(0, 3))]
    ang = rng.uniform(-0.3, 0.3)
    c2w = np.eye(4)
    c2w[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    c2w[:3, 3] = rng.uniform(-0.3, 0.3, 3)
    cam = simple_camera(H, W, c2w=c2w)
    means, cov6, opac, shs = random_scene(G, seed=seed, z_range=(0.1, 7.0), spread=1.5, scale=scale, sh_degree=deg, op_range=op_range)
    bg = tuple(rng.uniform(0, 1, 3)) if rng.integers(0, 2) else (0, 0, 0)
    kw = dict(shs=shs, sh_degree=deg) if use_sh else dict(colors=rng.uniform(0, 1, (G, 3)))
    # (hundreds of screen-filling splats per pixel: more pixels have SOME entry within 1e-4 of a threshold)
    _check_forward(means, cov6, opac, cam, bg=bg, min_ok=0.8, **kw)
    _check_backward(means, cov6, opac, cam, bg=bg, pose=bool(rng.integers(0, 2)), seed=seed, f64_rel=1e-2, **kw)   # fp64 check is a sanity bound here:
    # in these extreme scenes single alpha >= 1/255 decisions differ between fp32 and fp64 arithmetic; the parity bar is the 1e-4 match with the fp32 oracle


def test_prezeroed_gradient_accumulators_equal_the_memset_path_and_survive_a_second_backward(monkeypatch):
    """GSR_FLAG_PREZERO_GRADS (include/gsr.h): the composite-forward kernel zeroes the backward's per-(view, Gaussian)
    accumulators as a side job and gsr_backward skips its memset.  Same gradients as the memset path; a second backward
    through the same graph (retain_graph) finds dirty accumulators, zeroes them itself and returns the same gradients."""
    from styl3r_amd import _lib
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder
    from styl3r_amd.scenes import make_scene
    dev = torch.device("cuda:0")
    scs = [make_scene(n_ctx=1, grid_hw=(64, 64), n_views=2, image_hw=(80, 96), sh_degree=1, seed=300 + i) for i in range(2)]
    st = lambda n: torch.stack([getattr(s, n) for s in scs]).to(dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.1, 0.2, 0.3], True)).to(dev)
    w = torch.rand(2, 2, 3, 80, 96, device=dev, generator=torch.Generator(dev).manual_seed(2))

    def grads(twice):
        g = Gaussians(*(st(n).requires_grad_(True) for n in ("means", "covariances", "harmonics", "opacities")))
        out = dec.forward(g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (80, 96))
        loss = (out.color * w).sum()
        leaves = (g.means, g.covariances, g.harmonics, g.opacities)
        first = torch.autograd.grad(loss, leaves, retain_graph=twice)
        return first, (torch.autograd.grad(loss, leaves) if twice else None)

    seen = []
    real = _lib.load().gsr_backward_fused
    import ctypes as C
    class Spy:                                       # records the flag word every gsr_backward_fused call receives
        def __call__(self, dims, *a):
            seen.append(C.cast(dims, C.POINTER(_lib.GsrDims)).contents.flags & _lib.GSR_FLAG_PREZERO_GRADS)
            return real(dims, *a)
    monkeypatch.setattr(_lib.load(), "gsr_backward_fused", Spy(), raising=False)
    a, a2 = grads(True)
    assert seen == [_lib.GSR_FLAG_PREZERO_GRADS, 0], seen      # first backward trusts the forward, the second one zeroes
    monkeypatch.setattr(_lib, "GSR_FLAG_PREZERO_GRADS", 0)     # memset path
    b, _ = grads(False)
    assert seen[-1] == 0
    for x, y, z, name in zip(a, a2, b, ("means", "cov", "sh", "opac")):
        assert_close_rel(x.cpu().numpy(), z.cpu().numpy(), 2e-5, f"prezero vs memset d{name}")
        assert_close_rel(y.cpu().numpy(), z.cpu().numpy(), 2e-5, f"second backward d{name}")


def test_lds_histogram_binning_equals_the_wave_aggregated_one(monkeypatch):
    """K1 / K3 bin through LDS histograms when a view has at most LDS_TILES_MAX tiles (round 5); GSR_FLAG_BIN_BALLOT selects the wave-aggregated
    global atomics of rounds 1 - 4, which larger images still take.  Both must leave the same sorted lists: images, radii, depth, n_contrib and
    are compared bit for bit (the composite forward is deterministic given the lists); the gradients at 2e-5 (the backward's per-tile atomics
    arrive in whatever order the tiles finish)."""
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder
    from styl3r_amd.scenes import make_scene
    dev = torch.device("cuda:0")
    scs = [make_scene(n_ctx=1, grid_hw=(96, 96), n_views=3, image_hw=(112, 144), sh_degree=1, seed=500 + i) for i in range(2)]
    st = lambda n: torch.stack([getattr(s, n) for s in scs]).to(dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.2, 0.1, 0.3], True)).to(dev)
    w = torch.rand(2, 3, 3, 112, 144, device=dev, generator=torch.Generator(dev).manual_seed(5))

    def run():
        g = Gaussians(*(st(n).requires_grad_(True) for n in ("means", "covariances", "harmonics", "opacities")))
        out = dec.forward(g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (112, 144))
        grads = torch.autograd.grad((out.color * w).sum() + out.depth.sum(), (g.means, g.covariances, g.harmonics, g.opacities))
        nc = ws_view("n_contrib", np.int32, 2 * 3 * 112 * 144).copy()
        return out.color.detach().clone(), out.depth.detach().clone(), nc, int(rz.LAST_DEBUG["num_pairs"]), [x.clone() for x in grads]

    from styl3r_amd import _lib
    a = run()
    monkeypatch.setattr(rz, "EXTRA_FLAGS", _lib.GSR_FLAG_BIN_BALLOT)
    b = run()
    assert a[3] == b[3] and a[3] > 0
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for x, y, name in zip(a[4], b[4], ("means", "cov", "sh", "opac")):
        assert_close_rel(x.cpu().numpy(), y.cpu().numpy(), 2e-5, f"lds vs ballot binning d{name}")      # (the bar of the prezero / memset test below: atomics arrive in any order)


def _mse_step(fused, dev, seed=0, weight=0.7, res=(112, 144), grid=(96, 96), status_direct=True):
    """2 scenes x 3 views through the decoder + LossMse, fused into the composite kernels or as the stand-alone pair of kernels"""
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder
    from styl3r_amd.losses import mse_loss
    from styl3r_amd.scenes import make_scene
    scs = [make_scene(n_ctx=1, grid_hw=grid, n_views=3, image_hw=res, sh_degree=1, seed=700 + seed + i) for i in range(2)]
    st = lambda n: torch.stack([getattr(s, n) for s in scs]).to(dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.2, 0.1, 0.3], True)).to(dev)
    target = torch.rand(2, 3, 3, *res, device=dev, generator=torch.Generator(dev).manual_seed(11 + seed))
    g = Gaussians(*(st(n).requires_grad_(True) for n in ("means", "covariances", "harmonics", "opacities")))
    args = (g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), res)
    if fused:
        out = dec.forward(*args, mse_target=target, mse_weight=weight)
        loss = out.loss_mse
    else:
        out = dec.forward(*args)
        loss = mse_loss(out.color, target, weight)
    grads = torch.autograd.grad(loss, (g.means, g.covariances, g.harmonics, g.opacities))
    return out.color.detach(), target, loss.detach(), [x.detach() for x in grads]


def test_fused_mse_equals_the_stand_alone_loss_kernels_and_float64():
    """LossMse inside the composite kernels (GsrFused.mse_target): the forward value against the float64 expression on the rendered image
    (fixed-order fp32 partial sums: 1e-6) and bit-for-bit across repeats (deterministic); the image itself unchanged; the gradients against
    the stand-alone gsr_mse_forward / gsr_mse_backward path -- the same dL/dimage bits enter the same composite backward, the atomics'
    arrival order is the only difference (the 2e-5 bar of the prezero / memset test)."""
    dev = torch.device("cuda:0")
    img_f, target, loss_f, grads_f = _mse_step(True, dev)
    img_u, _, loss_u, grads_u = _mse_step(False, dev)
    assert torch.equal(img_f, img_u)
    ref = 0.7 * ((img_f.double() - target.double()) ** 2).mean()
    assert abs(loss_f.item() - ref.item()) <= 1e-6 * abs(ref.item()), (loss_f.item(), ref.item())
    assert abs(loss_u.item() - ref.item()) <= 1e-6 * abs(ref.item())
    for _ in range(3):                                   # deterministic: the tickets fix nothing but WHO adds, never the order
        assert _mse_step(True, dev)[2].item() == loss_f.item()
    for x, y, name in zip(grads_f, grads_u, ("means", "cov", "sh", "opac")):
        assert_close_rel(x.cpu().numpy(), y.cpu().numpy(), 2e-5, f"fused vs stand-alone MSE d{name}")


def test_fused_mse_adds_to_an_image_gradient_from_elsewhere_and_survives_an_unused_loss():
    """loss_mse + another consumer of the colour: the composite backward adds the two image gradients in its prologue; a fused forward whose
    loss nobody differentiates takes the plain backward."""
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder
    from styl3r_amd.scenes import make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(n_ctx=1, grid_hw=(64, 64), n_views=2, image_hw=(64, 80), sh_degree=0, seed=41)
    st = lambda n: getattr(sc, n)[None].to(dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    target = torch.rand(1, 2, 3, 64, 80, device=dev, generator=torch.Generator(dev).manual_seed(3))
    w = torch.rand(1, 2, 3, 64, 80, device=dev, generator=torch.Generator(dev).manual_seed(4))

    def run(mode):
        g = Gaussians(*(st(n).requires_grad_(True) for n in ("means", "covariances", "harmonics", "opacities")))
        args = (g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (64, 80))
        if mode == "fused+other":
            out = dec.forward(*args, mse_target=target, mse_weight=2.0)
            loss = out.loss_mse + (out.color * w).sum()
        elif mode == "fused, loss unused":
            out = dec.forward(*args, mse_target=target, mse_weight=2.0)
            loss = (out.color * w).sum()
        elif mode == "plain other":
            loss = (dec.forward(*args).color * w).sum()
        else:
            out = dec.forward(*args)
            loss = 2.0 * ((out.color - target) ** 2).mean() + (out.color * w).sum()
        return [x.detach().cpu().numpy() for x in torch.autograd.grad(loss, (g.means, g.covariances, g.harmonics, g.opacities))]

    a, b = run("fused+other"), run("torch")
    for x, y, name in zip(a, b, ("means", "cov", "sh", "opac")):
        assert_close_rel(x, y, 2e-5, f"fused MSE + other consumer d{name}")
    a, b = run("fused, loss unused"), run("plain other")
    for x, y, name in zip(a, b, ("means", "cov", "sh", "opac")):
        assert_close_rel(x, y, 2e-5, f"unused fused loss d{name}")


def test_status_words_in_pinned_host_memory_equal_the_copied_ones(monkeypatch):
    """the tile scan stores the status words straight into pinned host memory (no copy kernel in the step); the round-1..5 form -- device
    words + an asynchronous copy -- must read the same pair count and give the same image; persistent tile counters are re-armed by the
    scan: a second forward on them sees the same lists."""
    dev = torch.device("cuda:0")
    a = _mse_step(True, dev, seed=5)
    Ra = rz.LAST_STATS["pairs"]
    a2 = _mse_step(True, dev, seed=5)
    assert rz.LAST_STATS["pairs"] == Ra and torch.equal(a[0], a2[0])
    monkeypatch.setattr(rz, "STATUS_DIRECT", False)
    b = _mse_step(True, dev, seed=5)
    assert rz.LAST_STATS["pairs"] == Ra and Ra > 0
    assert torch.equal(a[0], b[0]) and a[2].item() == b[2].item()
