import sys; sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from styl3r_amd import vit_ops
dev = "cuda:0"
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
for (B, Ci, Co, H) in [(2, 256, 256, 128), (2, 256, 256, 64), (2, 256, 256, 32), (2, 256, 256, 16), (1, 256, 256, 128), (1, 256, 256, 64), (8, 256, 256, 32), (8, 256, 256, 16), (16, 256, 256, 16), (2, 256, 128, 256), (1, 256, 128, 256)]:
    x = torch.randn(B, Ci, H, H, device=dev); w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.02
    wp = vit_ops.split_conv_weight(w)
    t6 = timeit(lambda: vit_ops.conv_x6_forward(x, w, packed=wp)); tm = timeit(lambda: F.conv2d(x, w, padding=1))
    print((B, Ci, Co, H), "pixels", B * H * H, "tiles", ((Co + 127) // 128) * ((B * H * H + 127) // 128), "x6 %.3f ms  miopen %.3f ms  ratio %.2f" % (t6, tm, tm / t6), flush=True)
