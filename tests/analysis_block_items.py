"""Analysis script, not a test (drives the oracle).  CPU statistic for the round-5 composite backward (row-packed 4x4 blocks):
for the headline scene, per (tile, list entry) pair that actually composited (entry < n_contrib of the pixel, alpha >= 1/255):
valid pixels, 8x8 quadrants and 4x4 blocks touched; and the step count of the row-packed schedule (wave = tile quadrant, 16-lane
row = one 4x4 block, each row walks the entries that touch its block; a batch costs max over rows + drain)."""
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
starts = st.ranges[:, 0].astype(np.int64); ends = st.ranges[:, 1].astype(np.int64)
tile = np.repeat(np.arange(256), ends - starts)
pos = np.arange(R) - starts[tile]                                   # 0-based position in the tile's list
ncon = st.n_contrib.reshape(16, 16, 16, 16).transpose(0, 2, 1, 3).reshape(256, 16, 16)   # [tile][y][x]
ox = (tile % 16) * 16; oy = (tile // 16) * 16
x = torch.tensor(st.xy[pl, 0]); y = torch.tensor(st.xy[pl, 1])
A = torch.tensor(st.conic_opacity[pl, 0]); B = torch.tensor(st.conic_opacity[pl, 1]); Cc = torch.tensor(st.conic_opacity[pl, 2]); op = torch.tensor(st.conic_opacity[pl, 3])
px = torch.arange(16).float()
cnt = np.zeros(R, np.int64); blk = np.zeros((R, 16), bool); quad = np.zeros((R, 4), bool)
nct = torch.tensor(ncon.astype(np.int64))
for a in range(0, R, 1 << 15):
    e = slice(a, min(R, a + (1 << 15)))
    dx = x[e, None, None] - (torch.tensor(ox[e])[:, None, None] + px[None, None, :])
    dy = y[e, None, None] - (torch.tensor(oy[e])[:, None, None] + px[None, :, None])
    power = -0.5 * (A[e, None, None] * dx * dx + Cc[e, None, None] * dy * dy) - B[e, None, None] * dx * dy
    alpha = torch.clamp(op[e, None, None] * torch.exp(power), max=0.99)
    ok = (power <= 0) & (alpha >= 1 / 255) & (torch.tensor(pos[e])[:, None, None] < nct[torch.tensor(tile[e])])
    cnt[e] = ok.sum(dim=(1, 2)).numpy()
    blk[e] = ok.view(-1, 4, 4, 4, 4).any(dim=4).any(dim=2).reshape(-1, 16).numpy()      # [by*4+bx]
    quad[e] = ok.view(-1, 2, 8, 2, 8).any(dim=4).any(dim=2).reshape(-1, 4).numpy()
maxlast = ncon.reshape(256, -1).max(1)
live = pos < maxlast[tile]
print(f"pairs R {R}; R_eff (entry < tile's deepest contributor) {live.sum()}; composited pairs {(cnt > 0).sum()}")
c = cnt > 0
print(f"per composited pair: valid px {cnt[c].mean():.1f}, quadrants {quad[c].sum(1).mean():.2f}, 4x4 blocks {blk[c].sum(1).mean():.2f}")
print(f"per R_eff pair:      valid px {cnt[live].mean():.1f}, quadrants {quad[live].sum(1).mean():.2f}, 4x4 blocks {blk[live].sum(1).mean():.2f}")
print("lane density: today", cnt[live].sum() / (quad[live].sum() * 64), " row-packed blocks", cnt[live].sum() / (blk[live].sum() * 16))
# schedule simulation: wave = (tile, quadrant); its four rows = the quadrant's 2x2 blocks; batch = NB compacted entries (those that touch the quadrant)
for NB, DRAIN in ((64, 15), (128, 15), (10 ** 9, 15), (64, 0)):
    steps = 0; items = 0
    for t in range(256):
        b = blk[starts[t]:ends[t]][::-1]                    # back to front
        for q in range(4):
            qy, qx = q >> 1, q & 1
            cols = [(2 * qy + i) * 4 + 2 * qx + j for i in range(2) for j in range(2)]
            bq = b[:, cols]
            bq = bq[bq.any(1)]
            for a in range(0, len(bq), NB):
                n_r = bq[a:a + NB].sum(0)
                steps += int(n_r.max()) + DRAIN; items += int(n_r.sum())
    print(f"batch {NB if NB < 10**8 else 'inf'} drain {DRAIN}: steps {steps} for {items} items -> {items / (4 * steps):.2f} row occupancy; steps per R_eff pair {steps / live.sum():.3f}")

# ---- the ring schedule of k_composite_bwd_rows: chunks of CH compacted entries, K chunks resident, rows pop one item per step, a chunk slot is
# re-staged once every row's head and every lagging lane (15 steps behind) left it ----
def ring_steps(bq, CH, K, LAG=15):
    n = len(bq)
    nch = (n + CH - 1) // CH
    items = [[np.flatnonzero(bq[c * CH:(c + 1) * CH, r]).tolist() for r in range(4)] for c in range(nch)]
    staged = 0; cc = [-1] * 4; rem = [[] for _ in range(4)]
    hist = []                                     # per step: chunk of each row's popped item (or -1)
    steps = 0
    while True:
        # staging
        if staged < nch:
            free = staged < K
            if not free:
                victim = staged - K
                inflight = any(ch == victim for past in hist[-LAG:] for ch in past)
                free = all(c > victim for c in cc) and not inflight
            if free: staged += 1
        popped = []
        for r in range(4):
            if not rem[r] and cc[r] + 1 < staged:
                cc[r] += 1; rem[r] = list(items[cc[r]][r])
            if rem[r]:
                rem[r].pop(0); popped.append(cc[r])
            else: popped.append(-1)
        hist.append(popped); steps += 1
        if staged == nch and all((not rem[r]) and cc[r] + 1 >= staged for r in range(4)): break
    return steps + LAG
for CH, K in ((32, 4), (32, 8), (64, 2), (16, 8)):
    steps = 0
    for t in range(256):
        b = blk[starts[t]:ends[t]][::-1]
        for q in range(4):
            qy, qx = q >> 1, q & 1
            cols = [(2 * qy + i) * 4 + 2 * qx + j for i in range(2) for j in range(2)]
            bq = b[:, cols]; bq = bq[bq.any(1)]
            if len(bq): steps += ring_steps(bq, CH, K)
    print(f"ring CH {CH} x K {K}: steps per R_eff pair {steps / live.sum():.3f}")
