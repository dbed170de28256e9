import sys; sys.path.insert(0, ".")
import torch
from styl3r_amd.ddp import BucketedGradReducer
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
dev = torch.device("cuda:0"); torch.manual_seed(0)
tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2, pos_embed="RoPE100", img_size=(512, 512))
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=True), trunk_params=tiny).to(dev).eval()
g = torch.Generator(dev).manual_seed(2); H = 64
ctx = dict(image=torch.rand(1, 2, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]], device=dev).expand(1, 2, 3, 3).contiguous())
s1 = dict(image=torch.rand(1, 3, H, H, device=dev, generator=g) * 2 - 1); s2 = dict(image=ctx["image"][:, 0])
def run(inplace):
    params = [p for p in enc.parameters() if p.requires_grad]
    red = BucketedGradReducer(params, None, bucket_bytes=8 << 20, inplace_grads=inplace)
    red.prepare(); tot = 0
    for st in (s1, s2):
        gs = enc(ctx, st, 0)
        tot = tot + (gs.means * 0.01).sum() + gs.harmonics.sum() * 0.1 + gs.opacities.sum() * 0.1 + gs.covariances.sum() * 1e3
    tot.backward(); red.finish()
    out = {n: p.grad.detach().clone() for n, p in enc.named_parameters() if p.grad is not None}
    red.close(); return out
runs = {"c1": run(False), "c2": run(False), "i1": run(True), "i2": run(True), "c3": run(False)}
def cmp(a, b):
    worst = sorted(((float((runs[a][n] - runs[b][n]).abs().max() / (runs[a][n].abs().max() + 1e-30)), n) for n in runs[a]), reverse=True)[:3]
    print(a, b, [(f"{r:.1e}", n[-45:]) for r, n in worst])
for a, b in (("c1", "c2"), ("i1", "i2"), ("c1", "i1"), ("c2", "i2"), ("c1", "c3"), ("i1", "c3")): cmp(a, b)
