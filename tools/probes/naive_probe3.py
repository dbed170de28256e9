import sys; sys.path.insert(0, ".")
import torch
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "sum"
tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12, pos_embed="RoPE100", img_size=(512, 512))
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=tiny).to(dev)
b, v, H = 2, 2, 256
ctx = dict(image=torch.rand(b, v, 3, H, H, device=dev) * 2 - 1, intrinsics=torch.eye(3, device=dev).expand(b, v, 3, 3).contiguous())
for _ in range(2):
    g = enc(ctx, dict(image=ctx["image"][:, 0]), 0)
    (g.means.sum() + g.covariances.sum() + g.harmonics.sum() + g.opacities.sum()).backward()
torch.cuda.synchronize()
