import sys; sys.path.insert(0, ".")
import torch, numpy as np
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
d = rz.LAST_DEBUG; L = d["layout"]
cur = d["ws"][L.tile_cursor:L.tile_cursor + 16].view(torch.int32).cpu().numpy()
R = d["num_pairs"]
print("pairs R", R, "pixel evals past power test", cur[0], "alpha >= 1/255", cur[1], "useful frac", cur[1] / max(cur[0], 1), "evals per pair", cur[0] / R, "useful per pair", cur[1] / R)
print("quadrant passes reaching the alpha test", cur[2], "with no lane >= 1/255", cur[3], "frac", cur[3] / max(cur[2], 1))
