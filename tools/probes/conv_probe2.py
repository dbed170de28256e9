import time, torch, torch.nn as nn
dev = "cuda:0"
def run(name, conv, x, iters=5):
    c = conv.to(dev); xi = x.to(dev).requires_grad_(True)
    for _ in range(2):
        y = c(xi); y.sum().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        y = c(xi); y.sum().backward()
    torch.cuda.synchronize(); print(f"{name:34s} {(time.perf_counter() - t0) / iters * 1e3:9.3f} ms", flush=True)
B = 2
run("ConvT 96->96 k4 s4 @16", nn.ConvTranspose2d(96, 96, 4, 4), torch.randn(B, 96, 16, 16))
run("ConvT 192->192 k2 s2 @16", nn.ConvTranspose2d(192, 192, 2, 2), torch.randn(B, 192, 16, 16))
run("Conv 768->768 3x3 s2 @16", nn.Conv2d(768, 768, 3, 2, 1), torch.randn(B, 768, 16, 16))
run("Conv 768->768 1x1 @16", nn.Conv2d(768, 768, 1), torch.randn(B, 768, 16, 16))
run("Conv 96->256 3x3 nobias @64", nn.Conv2d(96, 256, 3, 1, 1, bias=False), torch.randn(B, 96, 64, 64))
run("Conv 768->256 3x3 nobias @8", nn.Conv2d(768, 256, 3, 1, 1, bias=False), torch.randn(B, 768, 8, 8))
run("Conv 256->256 3x3 @8", nn.Conv2d(256, 256, 3, 1, 1), torch.randn(B, 256, 8, 8))
run("Conv 256->256 1x1 @16", nn.Conv2d(256, 256, 1), torch.randn(B, 256, 16, 16))
run("Conv 3->1024 k16 s16 @256 (patch)", nn.Conv2d(3, 1024, 16, 16), torch.randn(2 * B, 3, 256, 256))
run("Conv 256->128 3x3 @128", nn.Conv2d(256, 128, 3, 1, 1), torch.randn(B, 256, 128, 128))
run("Conv 128->3 1x1 @256", nn.Conv2d(128, 3, 1), torch.randn(B, 128, 256, 256))
