#!/usr/bin/env python
"""Host-side cost of a C3 train step: cProfile over 3 steps (the step has ~6 800 launches; the lazy weight re-split experiment of round 5
showed that stretches of it are host-bound).  GPU box: python tools/probes/train_host_profile.py [scenes]"""
import cProfile, pstats, sys, io, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene, recentre_output_heads_
from styl3r_amd.train import TrainStep

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False)).to(dev)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
g = torch.Generator(dev).manual_seed(1234)
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=4, image_hw=(256, 256), seed=1234)
ex = lambda t, *shape: t.to(dev)[None].expand(b, *shape).contiguous()
batch = dict(context=dict(image=torch.rand(b, 2, 3, 256, 256, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(b, 2, 3, 3).contiguous()),
             target=dict(image=torch.rand(b, 4, 3, 256, 256, device=dev, generator=g), extrinsics=ex(sc.extrinsics, -1, -1, -1),
                         intrinsics=ex(sc.intrinsics, -1, -1, -1), near=ex(sc.near, -1), far=ex(sc.far, -1)))
recentre_output_heads_(enc, batch["context"], dict(image=batch["context"]["image"][:, 0]))
step = TrainStep(enc, dec, warm_up_steps=2000)
vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = "f16x3"
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
# host time of a step when the GPU is NOT the limiter: enqueue only, then wait
t0 = time.perf_counter(); step(batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"one step: host returns after {1e3 * (t1 - t0):.1f} ms, GPU done after {1e3 * (t2 - t0):.1f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
