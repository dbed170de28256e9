import sys; sys.path.insert(0, ".")
import numpy as np, torch
from styl3r_amd.decoder import prepare_views
from styl3r_amd.scenes import make_scene
from tests.gpu_utils import hip_single_view, oracle_single_view
sc = make_scene(n_ctx=2, grid_hw=(256, 256), n_views=4, image_hw=(256, 256), sh_degree=0, seed=5)
views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(4, 3), True).numpy()
for v in range(4):
    row = views[v]; s = np.float32(row[56])
    cov = sc.covariances.numpy()
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1) * (s * s)
    cam = dict(H=256, W=256, tanfovx=row[51], tanfovy=row[52], view=row[0:16].reshape(4, 4), proj=row[16:32].reshape(4, 4),
               proj_raw=row[32:48].reshape(4, 4), campos=row[48:51])
    means = sc.means.numpy() * s; op = sc.opacities.numpy(); shs = sc.harmonics.numpy().transpose(0, 2, 1)
    perm = np.random.default_rng(0).permutation(len(op))
    a = hip_single_view(means, cov6, op, cam, shs=shs)["image"].cpu().numpy()
    b = hip_single_view(means[perm], cov6[perm], op[perm], cam, shs=shs[perm])["image"].cpu().numpy()
    _, sa, _ = oracle_single_view("f32", means, cov6, op, cam, shs=shs)
    _, sb, _ = oracle_single_view("f32", means[perm], cov6[perm], op[perm], cam, shs=shs[perm])
    d_gpu = np.abs(a - b).max(0); d_orc = np.abs(sa.image - sb.image).max(0)
    print(f"view {v}: gpu perm diff max {d_gpu.max():.4f} n {int((d_gpu>1e-4).sum())} | oracle perm diff max {d_orc.max():.4f} n {int((d_orc>1e-4).sum())} | gpu-vs-oracle {np.abs(a-sa.image).max():.2e} {np.abs(b-sb.image).max():.2e}")
    if d_orc.max() > 1e-3:
        y, x = np.unravel_index(d_orc.argmax(), d_orc.shape)
        t = (y // 16) * 16 + x // 16
        s0, e0 = sa.ranges[t]; ids = sa.point_list[s0:e0]; dep = sa.depth[ids]
        ties = int((np.diff(dep) == 0).sum())
        print("   worst pixel", y, x, "tile", t, "list len", e0 - s0, "exact depth ties in list", ties, "n_contrib", sa.n_contrib[y, x], sb.n_contrib[y, x])
