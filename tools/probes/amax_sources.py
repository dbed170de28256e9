#!/usr/bin/env python
"""Which tensors of a C3 train step still need their own vit_amax pass in f16x3 mode (667 launches / 6 ms per step in r04): the call sites
of vit_ops._amax_word, grouped by the two frames above it and the tensor shape.  GPU box:  python tools/probes/amax_sources.py [scenes]"""
import collections, sys, traceback
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene, recentre_output_heads_
from styl3r_amd.train import TrainStep

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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
for _ in range(2):
    step(batch)
seen = collections.Counter()
orig = vit_ops._amax_word


def spy(t):
    fr = traceback.extract_stack(limit=6)[:-1]
    where = " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(fr[-4:]))
    seen[(where, tuple(t.shape))] += 1
    return orig(t)


vit_ops._amax_word = spy
before = dict(vit_ops.CALLS)
step(batch)
torch.cuda.synchronize()
print("amax passes in one step:", vit_ops.CALLS["amax_pass"] - before["amax_pass"], " published:", vit_ops.CALLS["amax_published"] - before["amax_published"])
for (where, shape), n in seen.most_common(40):
    print(f"{n:5d} x {str(shape):28s} {where}")
