"""Analysis script, not a test (kept under tests/ because it drives the oracle, which only tests / smoke / the bench's CPU leg may use).
CPU statistic (no GPU): for the headline scene, the distribution of the number of pixels of a 16x16 tile that pass the
alpha >= 1/255 test per (tile, Gaussian) pair -- what the wave reduction of the composite backward is amortised over."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from oracle.gsr_oracle import Oracle
from styl3r_amd.decoder import prepare_views
from styl3r_amd.scenes import make_scene
sc = make_scene(1, (256, 256), 4, (256, 256), seed=1234)
orc = Oracle("f32")
views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(4, 3), True).numpy()
row = views[1]; s = np.float32(row[56])
cov = sc.covariances.numpy()
cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
st, ctx = orc.forward(sc.means.numpy() * s, cov6 * (s * s), sc.opacities.numpy(), shs=sc.harmonics.numpy().transpose(0, 2, 1), H=256, W=256,
                      tanfovx=row[51], tanfovy=row[52], bg=(0, 0, 0), view=row[0:16], proj=row[16:32], proj_raw=row[32:48],
                      campos=row[48:51], sh_degree=0, nthreads=8)
R = st.R
pl = st.point_list.astype(np.int64)
tile = np.repeat(np.arange(256), st.ranges[:, 1] - st.ranges[:, 0])
ox = (tile % 16) * 16; oy = (tile // 16) * 16
x = torch.tensor(st.xy[pl, 0]); y = torch.tensor(st.xy[pl, 1])
A = torch.tensor(st.conic_opacity[pl, 0]); B = torch.tensor(st.conic_opacity[pl, 1]); Cc = torch.tensor(st.conic_opacity[pl, 2]); op = torch.tensor(st.conic_opacity[pl, 3])
px = torch.arange(16).float()
cnt = np.zeros(R, np.int64); qcnt = np.zeros((R, 4), np.int64)
for a in range(0, R, 1 << 16):
    e = slice(a, min(R, a + (1 << 16)))
    dx = x[e, None, None] - (torch.tensor(ox[e])[:, None, None] + px[None, None, :])
    dy = y[e, None, None] - (torch.tensor(oy[e])[:, None, None] + px[None, :, None])
    power = -0.5 * (A[e, None, None] * dx * dx + Cc[e, None, None] * dy * dy) - B[e, None, None] * dx * dy
    alpha = torch.clamp(op[e, None, None] * torch.exp(power), max=0.99)
    ok = (power <= 0) & (alpha >= 1 / 255)
    cnt[e] = ok.sum(dim=(1, 2)).numpy()
    qcnt[e] = ok.view(-1, 2, 8, 2, 8).sum(dim=(2, 4)).view(-1, 4).numpy()
print("pairs", R, "mean valid px per pair", cnt.mean(), "zero:", (cnt == 0).mean())
nz = cnt[cnt > 0]
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 256)):
    m = (nz >= lo) & (nz <= hi)
    print(f"  valid px {lo:3d}..{hi:3d}: {m.mean() * 100:5.1f} % of contributing pairs, {nz[m].sum() / nz.sum() * 100:5.1f} % of the useful evaluations")
print("quadrants touched per contributing pair:", ((qcnt > 0).sum(1)[cnt > 0]).mean())
print("cumulative: pairs with <= k valid px:", {k: round(float((nz <= k).mean()), 3) for k in (1, 2, 3, 4, 6, 8, 12, 16)})

# ---- what a finer evaluation granularity would buy (round-3 estimate) --------------------------------------------------------
# Today a wave evaluates a pair once per touched 8x8 quadrant (64 lanes = 64 pixels).  Alternative: the two half-waves run as two
# independent 32-pixel machines -- lanes 0..31 own rows 0..3 of every quadrant, lanes 32..63 rows 4..7 -- each walking the list at
# its own pace over the 8x4 half-quadrants a splat touches; a 64-entry batch then costs max(top items, bottom items) passes.
hq = np.zeros((R, 8), bool)                      # [pair][quadrant * 2 + half]
for a in range(0, R, 1 << 16):
    e = slice(a, min(R, a + (1 << 16)))
    dx = x[e, None, None] - (torch.tensor(ox[e])[:, None, None] + px[None, None, :])
    dy = y[e, None, None] - (torch.tensor(oy[e])[:, None, None] + px[None, :, None])
    power = -0.5 * (A[e, None, None] * dx * dx + Cc[e, None, None] * dy * dy) - B[e, None, None] * dx * dy
    alpha = torch.clamp(op[e, None, None] * torch.exp(power), max=0.99)
    ok = (power <= 0) & (alpha >= 1 / 255)
    # (pair, qy, half, 4 rows, qx, 8 cols) -> any over rows / cols
    h = ok.view(-1, 2, 2, 4, 2, 8).any(dim=5).any(dim=3)          # (pair, qy, half, qx)
    hq[e] = h.permute(0, 1, 3, 2).reshape(-1, 8).numpy()
quad_passes = (qcnt > 0).sum()
starts = st.ranges[:, 0].astype(np.int64); ends = st.ranges[:, 1].astype(np.int64)
two_stream = 0
for s0, e0 in zip(starts, ends):
    for b0 in range(s0, e0, 64):
        blk = hq[b0:min(b0 + 64, e0)]
        top = blk[:, 0::2].sum(); bot = blk[:, 1::2].sum()
        two_stream += max(top, bot)
print(f"evaluation passes over view 1: per 8x8 quadrant (today, exact-valid masks) {quad_passes}; two independent half-waves over 8x4 "
      f"half-quadrants, per 64-entry batch {two_stream}  ->  {two_stream / quad_passes:.2f} x the passes, each over 32 instead of 64 pixels")
