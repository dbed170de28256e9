#!/usr/bin/env python
"""Which Python lines launch the framework's copy / fill / add kernels inside a C3 train step (543 direct_copy launches)?
One step of the tiny-trunk configuration (same code paths, fewer blocks) under torch.profiler with stacks; prints the call sites of
aten::copy_ / aten::contiguous / aten::clone / aten::fill_ / aten::zero_ / aten::add / aten::add_ aggregated by the innermost repo frame.
usage (GPU box): python tools/probes/copy_sources.py [--full] [--mode f16x3]"""
import argparse, collections, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from torch.profiler import profile, ProfilerActivity
ap = argparse.ArgumentParser(); ap.add_argument("--full", action="store_true"); ap.add_argument("--mode", default="f16x3"); ap.add_argument("--scenes", type=int, default=2)
a = ap.parse_args()
from styl3r_amd import vit_ops
vit_ops.LINEAR_MODE = a.mode; vit_ops.ATTENTION_ARITH = a.mode
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene, recentre_output_heads_
from styl3r_amd.train import TrainStep
dev = torch.device("cuda:0"); torch.manual_seed(0)
tiny = None if a.full else dict(enc_depth=2, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12, pos_embed="RoPE100", img_size=(512, 512))
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False), trunk_params=tiny).to(dev)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
b, H = a.scenes, 256
g = torch.Generator(dev).manual_seed(1234)
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=4, image_hw=(H, H), seed=1234)
K = sc.intrinsics[:1].to(dev)
batch = dict(context=dict(image=torch.rand(b, 2, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=K.expand(b, 2, 3, 3).contiguous()),
             target=dict(image=torch.rand(b, 4, 3, H, H, device=dev, generator=g), extrinsics=sc.extrinsics.to(dev)[None].expand(b, -1, -1, -1).contiguous(),
                         intrinsics=sc.intrinsics.to(dev)[None].expand(b, -1, -1, -1).contiguous(), near=sc.near.to(dev)[None].expand(b, -1).contiguous(),
                         far=sc.far.to(dev)[None].expand(b, -1).contiguous()))
recentre_output_heads_(enc, batch["context"], dict(image=batch["context"]["image"][:, 0]))
step = TrainStep(enc, dec, dist=None, warm_up_steps=2000)
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(batch)
    torch.cuda.synchronize()
WANT = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::cat", "aten::flip", "aten::gelu_backward", "aten::index_put_", "aten::_foreach", "aten::clone")
agg = collections.Counter()
for ev in prof.events():
    if not any(ev.name == w or ev.name.startswith("aten::_foreach") for w in WANT) or ev.device_time_total <= 0 and not ev.name.startswith("aten::"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name in WANT:
        continue        # count the outermost of nested ops
    frames = [f for f in (ev.stack or []) if "/root/repo" in f or "styl3r_amd" in f or "graft" in f]
    site = frames[0].strip() if frames else (ev.stack[0].strip() if ev.stack else "")
    par, q = [], ev.cpu_parent
    while q is not None and len(par) < 3:
        par.append(q.name.replace("autograd::engine::evaluate_function: ", "")[:40]); q = q.cpu_parent
    agg[(ev.name, (" < ".join(par) + " " + str(ev.input_shapes)[:70] + " " + site)[-150:])] += 1
for (name, site), n in agg.most_common(60):
    print(f"{n:5d}  {name:28s} {site}")
