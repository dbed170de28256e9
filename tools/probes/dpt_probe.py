"""Which convolutions of a DPT head end up on naive_conv kernels?  torch profiler with shapes."""
import sys; sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from styl3r_amd.encoder import CrocoTrunk, head_factory
dev = "cuda:0"
net = CrocoTrunk(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12,
                 pos_embed="RoPE100", img_size=(512, 512))
head = head_factory("dpt_gs", "gs_params", net, out_nchan=8).to(dev)
b = 2
toks = [torch.randn(b, 256, 1024 if i == 0 else 768, device=dev, requires_grad=True) for i in range(13)]
img = torch.randn(b, 3, 256, 256, device=dev)
for _ in range(2):
    out = head(toks, (256, 256), img); out.flatten(2).transpose(1, 2).sum().backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    out = head(toks, (256, 256), img); out.flatten(2).transpose(1, 2).sum().backward()
    torch.cuda.synchronize()
evs = prof.events()
# print every kernel with 'naive' and the closest preceding cpu op with shapes
naive = [e for e in evs if "naive" in e.name]
print("naive kernel launches:", len(naive))
for e in prof.key_averages(group_by_input_shape=True):
    if "conv" in e.key.lower() and e.device_time_total > 0 and ("aten::" in e.key or "Backward" in e.key):
        print(f"{e.key:45s} {str(e.input_shapes)[:110]:110s} dev_us={e.device_time_total:10.1f} n={e.count}")
