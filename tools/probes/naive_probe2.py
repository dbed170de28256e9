import sys, time; sys.path.insert(0, ".")
import torch, torch.nn as nn
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
dev = "cuda:0"
tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16, dec_num_heads=12, pos_embed="RoPE100", img_size=(512, 512))
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=tiny).to(dev)
b, v, H = 2, 2, 256
ctx = dict(image=torch.rand(b, v, 3, H, H, device=dev) * 2 - 1, intrinsics=torch.eye(3, device=dev).expand(b, v, 3, 3).contiguous())
LOG = []
def wrap(cls):
    orig = cls.forward
    def fwd(self, x, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = orig(self, x, *a, **k)
        torch.cuda.synchronize(); LOG.append(((time.perf_counter() - t0) * 1e3, cls.__name__, tuple(x.shape), tuple(x.stride()), tuple(self.weight.shape), self.stride))
        return y
    cls.forward = fwd
wrap(nn.Conv2d); wrap(nn.ConvTranspose2d)
def run():
    g = enc(ctx, dict(image=ctx["image"][:, 0]), 0)
    return g
run(); LOG.clear(); run()
for ms, n, sh, st, w, s in sorted(LOG, reverse=True)[:14]:
    print(f"{ms:9.3f} ms {n:16s} x{sh} strides{st} w{w} s{s}")
print("total conv fwd ms", sum(l[0] for l in LOG), "calls", len(LOG))
