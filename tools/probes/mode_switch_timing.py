"""Per-step time of the C3 train step across arithmetic-mode switches inside one process (bench.py times both modes in one run)."""
import sys, time; sys.path.insert(0, ".")
import torch
from styl3r_amd import vit_ops
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene
from styl3r_amd.train import TrainStep
dev = torch.device("cuda:0"); torch.manual_seed(0)
b = 10
with torch.device(dev):
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False))
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
step = TrainStep(enc, dec)
H = 256
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=4, image_hw=(H, H), seed=1)
ex = lambda t: t.to(dev)[None].expand(b, *t.shape).contiguous()
batch = dict(context=dict(image=torch.rand(b, 2, 3, H, H, device=dev) * 2 - 1, intrinsics=ex(sc.intrinsics[:1].expand(2, 3, 3))),
             target=dict(image=torch.rand(b, 4, 3, H, H, device=dev), extrinsics=ex(sc.extrinsics), intrinsics=ex(sc.intrinsics), near=ex(sc.near), far=ex(sc.far)))
for mode in ("bf16x3", "bf16x6", "bf16x3", "bf16x6"):
    vit_ops.LINEAR_MODE = mode
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(batch); torch.cuda.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 1))
    print(mode, ts, "mem GB", round(torch.cuda.memory_allocated() / 2 ** 30, 1), "reserved", round(torch.cuda.memory_reserved() / 2 ** 30, 1), "split cache entries", len(vit_ops._SPLIT_CACHE), flush=True)
