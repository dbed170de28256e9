"""Potential of running the two halves of the scene batch on two HIP streams (binning of one half under the
composite kernels of the other).  Prints ms/step for 1 stream and 2 streams."""
import sys, time; sys.path.insert(0, ".")
import torch
from styl3r_amd.decoder import DecoderSplattingCUDACfg, Gaussians, get_decoder
from styl3r_amd.losses import mse_loss
from styl3r_amd.scenes import make_scene
dev = torch.device("cuda:0")
B, Vt, H = 10, 4, 256
scs = [make_scene(1, (256, 256), Vt, (H, H), seed=1234 + i) for i in range(B)]
st = lambda n: torch.stack([getattr(s, n) for s in scs]).to(dev)
g = Gaussians(st("means").requires_grad_(), st("covariances").requires_grad_(), st("harmonics").requires_grad_(), st("opacities").requires_grad_())
cams = {k: st(k) for k in ("extrinsics", "intrinsics", "near", "far")}
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
target = torch.rand((B, Vt, 3, H, H), device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(2)]

def step1():
    for t in (g.means, g.covariances, g.harmonics, g.opacities): t.grad = None
    out = dec.forward(g, cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, H))
    mse_loss(out.color, target).backward()

def step2():
    for t in (g.means, g.covariances, g.harmonics, g.opacities): t.grad = None
    main = torch.cuda.current_stream(dev)
    losses = []
    for c, s in enumerate(streams):
        sl = slice(c * B // 2, (c + 1) * B // 2)
        s.wait_stream(main)
        with torch.cuda.stream(s):
            gc = Gaussians(g.means[sl], g.covariances[sl], g.harmonics[sl], g.opacities[sl])
            out = dec.forward(gc, cams["extrinsics"][sl], cams["intrinsics"][sl], cams["near"][sl], cams["far"][sl], (H, H))
            losses.append(mse_loss(out.color, target[sl].contiguous()) * 0.5)
    for l, s in zip(losses, streams):
        with torch.cuda.stream(s):
            pass
    torch.autograd.backward(losses)
    for s in streams: main.wait_stream(s)

for name, fn in (("1 stream", step1), ("2 streams", step2), ("1 stream", step1), ("2 streams", step2)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 30 * 1e3, 3), "ms/step")

