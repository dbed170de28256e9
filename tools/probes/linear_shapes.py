#!/usr/bin/env python
"""Which Linear shapes does one C3 train step launch, and which of them go to the ring kernels?  (forward and input-gradient GEMMs;
   counts per (M, N, K), FLOP share)   python tools/probes/linear_shapes.py [bf16x3|bf16x6]"""
import collections, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene
from styl3r_amd.train import TrainStep
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = mode
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False)).to(dev)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
step = TrainStep(enc, dec)
b, v_ctx, v_tgt, H = 10, 2, 4, 256
g = torch.Generator(dev).manual_seed(1234)
sc = make_scene(n_ctx=v_ctx, grid_hw=(8, 8), n_views=v_tgt, image_hw=(H, H), seed=1234)
ex = lambda t, *shape: t.to(dev)[None].expand(b, *shape).contiguous()
batch = dict(context=dict(image=torch.rand(b, v_ctx, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(b, v_ctx, 3, 3).contiguous()),
             target=dict(image=torch.rand(b, v_tgt, 3, H, H, device=dev, generator=g), extrinsics=ex(sc.extrinsics, -1, -1, -1),
                         intrinsics=ex(sc.intrinsics, -1, -1, -1), near=ex(sc.near, -1), far=ex(sc.far, -1)))
step(batch)
seen = collections.Counter()
orig = vit_ops._ring_cfg
def spy(M, N, K):
    r = orig(M, N, K)
    seen[(M, N, K, r)] += 1
    return r
vit_ops._ring_cfg = spy
step(batch); torch.cuda.synchronize()
tot = sum(2.0 * M * N * K * c for (M, N, K, r), c in seen.items())
print(f"| M | N | K | ring cfg | launches | GFLOP each | share of the Linear fwd+dX FLOPs |\n|---|---|---|---|---|---|---|")
for (M, N, K, r), c in sorted(seen.items(), key=lambda kv: -2.0 * kv[0][0] * kv[0][1] * kv[0][2] * kv[1]):
    print(f"| {M} | {N} | {K} | {r} | {c} | {2e-9 * M * N * K:.2f} | {100 * 2.0 * M * N * K * c / tot:.1f} % |")
