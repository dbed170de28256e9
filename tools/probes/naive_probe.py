import sys; sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
dev = "cuda:0"
tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12, pos_embed="RoPE100", img_size=(512, 512))
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=tiny).to(dev)
b, v, H = 2, 2, 256
ctx = dict(image=torch.rand(b, v, 3, H, H, device=dev) * 2 - 1, intrinsics=torch.eye(3, device=dev).expand(b, v, 3, 3).contiguous())
def run():
    g = enc(ctx, dict(image=ctx["image"][:, 0]), 0)
    (g.means.sum() + g.covariances.sum() + g.harmonics.sum() + g.opacities.sum()).backward()
run(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    run(); torch.cuda.synchronize()
seen = {}
for e in prof.events():
    ks = getattr(e, "kernels", [])
    if any("naive" in k.name for k in ks) and ("miopen" in e.name or "convolution_backward" == e.name.split("::")[-1]):
        key = (e.name, str(e.input_shapes))
        seen[key] = seen.get(key, 0) + sum(k.duration for k in ks)
for (n, s), d in sorted(seen.items(), key=lambda kv: -kv[1])[:20]:
    print(f"{d/1e3:9.2f} ms  {n:34s} {s[:150]}")
