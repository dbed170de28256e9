"""How many evaluation passes would the composite kernels need if 16-lane groups (4x4-pixel sub-blocks of an 8x8
quadrant) each walked their OWN list inside a 64-entry batch, vs one pass per (entry, quadrant) today?"""
import sys; sys.path.insert(0, ".")
import torch
from styl3r_amd import rasterizer as rz
from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder
from styl3r_amd.scenes import make_scene
dev = "cuda:0"
scs = [make_scene(1, (256, 256), 4, (256, 256), seed=1234 + i) for i in range(2)]
st = lambda n: torch.stack([getattr(s, n) for s in scs]).to(dev)
g = Gaussians(st("means"), st("covariances"), st("harmonics"), st("opacities"))
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0, 0, 0], True)).to(dev)
rz.KEEP_DEBUG = True
dec.forward(g, st("extrinsics"), st("intrinsics"), st("near"), st("far"), (256, 256))
d = rz.LAST_DEBUG; L = d["layout"]; R = d["num_pairs"]; ws = d["ws"]
V, T = 8, 256
q = ws[L.queue:L.queue + R * 48].view(torch.float32).view(R, 12)
quad = q[:, 11].view(torch.int32)
off = ws[L.tile_offset:L.tile_offset + 4 * (V * T + 1)].view(torch.int32).long()
tile_of = torch.repeat_interleave(torch.arange(V * T, device=dev), off[1:] - off[:-1])
pos = torch.arange(R, device=dev) - off[tile_of]              # position in the tile list
tl = tile_of % T
ox = (tl % 16) * 16; oy = (tl // 16) * 16
px = torch.arange(16, device=dev)
# per-pixel truth: alpha >= 1/255 and power <= 0
sub = torch.zeros((R, 16), dtype=torch.bool, device=dev)      # [entry, sub-block sy*4+sx]
useful = 0
for s in range(0, R, 1 << 18):
    e = slice(s, min(R, s + (1 << 18)))
    dx = q[e, 0, None, None] - (ox[e, None, None] + px[None, None, :]).float()          # (n,1,16)
    dy = q[e, 1, None, None] - (oy[e, None, None] + px[None, :, None]).float()          # (n,16,1)
    power = -0.5 * (q[e, 2, None, None] * dx * dx + q[e, 4, None, None] * dy * dy) - q[e, 3, None, None] * dx * dy
    alpha = torch.clamp(q[e, 5, None, None] * torch.exp(power), max=0.99)
    ok = (power <= 0) & (alpha >= 1 / 255)                     # (n,16y,16x)
    useful += int(ok.sum())
    sub[e] = ok.view(-1, 4, 4, 4, 4).any(dim=4).any(dim=2).view(-1, 16)
now = sum(((quad >> k) & 1) for k in range(4)).sum().item()
print("pairs", R, "passes now (entry,quadrant)", now, "useful pixel evals", useful, "per pass", useful / now / 64)
print("truth quadrants", int(sub.view(R, 2, 2, 2, 2).any(dim=4).any(dim=2).sum()), "sub-blocks (entry,4x4)", int(sub.sum()))
batch = tile_of * 4096 + pos // 64                              # batch id (tile, 64-entry batch)
nb = int(batch.max()) + 1
for name, perm in (("4x4 sub-blocks as groups", None),):
    total = 0
    for Qy in range(2):
        for Qx in range(2):
            cnts = []
            for gy in range(2):
                for gx in range(2):
                    bit = sub[:, (Qy * 2 + gy) * 4 + Qx * 2 + gx]
                    cnts.append(torch.zeros(nb, dtype=torch.int32, device=dev).index_add_(0, batch, bit.int()))
            total += int(torch.stack(cnts).max(dim=0).values.sum())
    print(name, "passes", total, "ratio vs now", total / now)
# also 8 groups of 8 lanes (4x2 px): passes
total = 0
subs8 = None
