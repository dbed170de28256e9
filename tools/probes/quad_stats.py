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
d = rz.LAST_DEBUG; L = d["layout"]; R = d["num_pairs"]
q = d["ws"][L.queue:L.queue + R * 48].view(torch.int32).view(R, 12)[:, 11]
pc = sum(((q >> k) & 1) for k in range(4))
print("pairs", R, "zero-mask frac", float((q == 0).float().mean()), "mean quadrants per pair", float(pc.float().mean()),
      "hist", [int((pc == i).sum()) for i in range(5)])
