"""Host issue time vs GPU time of the batch-1 encoder forward (is C2 launch-bound?), and a cProfile of the host side."""
import sys, time, cProfile, pstats; sys.path.insert(0, ".")
import torch
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene
dev = torch.device("cuda:0"); torch.manual_seed(0)
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg()).to(dev).eval()
enc.head_streams = "--streams" in sys.argv
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=3, image_hw=(256, 256), seed=1234)
g = torch.Generator(dev).manual_seed(1234)
ctx = dict(image=torch.rand(1, 2, 3, 256, 256, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(1, 2, 3, 3).contiguous())
style = dict(image=ctx["image"][:, 0])
with torch.no_grad():
    for _ in range(5): enc(ctx, style, 0)
    torch.cuda.synchronize()
    host, tot = [], []
    for _ in range(10):
        t0 = time.perf_counter(); enc(ctx, style, 0); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print(f"host issue {1e3 * sum(host) / 10:.2f} ms, until GPU done {1e3 * sum(tot) / 10:.2f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): enc(ctx, style, 0)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
