"""Which lines of this repo launch the framework's element-wise kernels (add / copy / fill / reduce / gelu_backward ...) in a C3 train step:
torch.profiler with Python stacks, grouped by (op, innermost styl3r_amd frame).   python tools/probes/train_elementwise.py [b]"""
import collections
import sys; sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene
from styl3r_amd.train import TrainStep
dev = torch.device("cuda:0"); torch.manual_seed(0)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with torch.device(dev):
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False))
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
from styl3r_amd import vit_ops
import os
vit_ops.LINEAR_MODE = os.environ.get("VIT_LINEAR_MODE", "f16x3")
H = 256
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=4, image_hw=(H, H), seed=1)
ex = lambda t: t.to(dev)[None].expand(b, *t.shape).contiguous()
batch = dict(context=dict(image=torch.rand(b, 2, 3, H, H, device=dev) * 2 - 1, intrinsics=ex(sc.intrinsics[:1].expand(2, 3, 3))),
             target=dict(image=torch.rand(b, 4, 3, H, H, device=dev), extrinsics=ex(sc.extrinsics), intrinsics=ex(sc.intrinsics), near=ex(sc.near), far=ex(sc.far)))
from styl3r_amd.scenes import recentre_output_heads_
recentre_output_heads_(enc, batch["context"], dict(image=batch["context"]["image"][:, 0]))
step = TrainStep(enc, dec, warm_up_steps=2000)
for _ in range(2): step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(batch); torch.cuda.synchronize()
WATCH = ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zero_", "aten::mul", "aten::mul_", "aten::sum", "aten::clamp_min", "aten::clamp_min_",
         "aten::relu", "aten::relu_", "aten::gelu_backward", "aten::cat", "aten::div", "aten::sub", "aten::where", "aten::threshold_backward", "aten::index_put_",
         "aten::_foreach_copy_", "aten::_foreach_zero_", "aten::_foreach_mul_", "aten::contiguous", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::_foreach_norm", "aten::linalg_vector_norm", "aten::stack", "aten::neg", "aten::exp", "aten::permute", "aten::pixel_shuffle")
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name not in WATCH or e.device_time_total <= 0:
        continue
    # an op nested inside another watched op (copy_ inside contiguous ...) is counted once, at the innermost level that owns kernels
    if e.cpu_children and any(c.name in WATCH and c.device_time_total > 0 for c in e.cpu_children):
        continue
    where = str(e.input_shapes)[:100]
    for fr in (e.stack or []):
        if "styl3r_amd" in fr:
            where = fr.split("styl3r_amd/")[-1].strip()
            break
    k = (e.name, where)
    agg[k][0] += 1; agg[k][1] += e.device_time_total
tot = sum(v[1] for v in agg.values())
print(f"framework element-wise total {tot / 1e3:.2f} ms in one step (b = {b})")
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t / 1e3:8.3f} ms  {n:5d} x  {name:28s} {where}")
