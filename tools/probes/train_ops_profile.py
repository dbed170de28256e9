"""torch.profiler view of one train step (b=2): which aten ops (with input shapes) own the GPU time."""
import sys; sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene
from styl3r_amd.train import TrainStep
dev = torch.device("cuda:0"); torch.manual_seed(0)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg()).to(dev)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
step = TrainStep(enc, dec)
H = 256
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=4, image_hw=(H, H), seed=1)
ex = lambda t: t.to(dev)[None].expand(b, *t.shape).contiguous()
batch = dict(context=dict(image=torch.rand(b, 2, 3, H, H, device=dev) * 2 - 1, intrinsics=ex(sc.intrinsics[:1].expand(2, 3, 3))),
             target=dict(image=torch.rand(b, 4, 3, H, H, device=dev), extrinsics=ex(sc.extrinsics), intrinsics=ex(sc.intrinsics), near=ex(sc.near), far=ex(sc.far)))
for _ in range(2): step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(batch); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=45, max_shapes_column_width=70))
