"""Which MIOpen fp32 path do the DPT-head convolutions get on gfx950?  times fwd+bwd of the heavy configs."""
import sys, time, torch, torch.nn as nn
dev = "cuda:0"
def run(name, conv, x, iters=5):
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for cl in (False, True):
            c = conv.to(dev)
            xi = x.clone().to(dev)
            if cl:
                c = c.to(memory_format=torch.channels_last); xi = xi.contiguous(memory_format=torch.channels_last)
            xi.requires_grad_(True)
            for _ in range(2):
                y = c(xi); y.sum().backward()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(iters):
                y = c(xi); y.sum().backward()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / iters * 1e3
            fl = 3 * 2 * y.numel() * conv.in_channels * conv.kernel_size[0] * conv.kernel_size[1] / (1 if not hasattr(conv, "output_padding") else 1)
            print(f"{name:28s} benchmark={bench!s:5s} channels_last={cl!s:5s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF", flush=True)
B = 2
run("3x3 256->256 @256 (gs head)", nn.Conv2d(256, 256, 3, 1, 1, bias=False), torch.randn(B, 256, 256, 256))
run("3x3 256->256 @128 (refine1)", nn.Conv2d(256, 256, 3, 1, 1), torch.randn(B, 256, 128, 128))
run("7x7 3->256 @256 (merger)", nn.Conv2d(3, 256, 7, 1, 3), torch.randn(B, 3, 256, 256))
run("3x3 128->128 @256 (pts head)", nn.Conv2d(128, 128, 3, 1, 1), torch.randn(B, 128, 256, 256))
run("1x1 256->8 @256", nn.Conv2d(256, 8, 1), torch.randn(B, 256, 256, 256))
run("1x1 1024->96 @16", nn.Conv2d(1024, 96, 1), torch.randn(B, 1024, 16, 16))
