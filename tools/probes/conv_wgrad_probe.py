import sys; sys.path.insert(0, ".")
import torch, torch.nn.functional as F, ctypes as C
from styl3r_amd import vit_ops
dev = "cuda:0"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
lib = vit_ops.load(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def wgrad(gy, x, Co, Ci, k):
    dw = torch.empty(Co, Ci, k, k, device=dev); db = torch.empty(Co, device=dev)
    B, _, H, W = x.shape
    rc = lib.vit_conv_x6_wgrad(gy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), B, Ci, Co, H, W, k, 0, st); assert rc == 0, rc
    return dw, db
for (B, Ci, Co, H, W, k) in [(2, 32, 40, 8, 16, 3), (1, 16, 130, 16, 8, 3), (3, 48, 128, 8, 8, 1), (1, 20, 70, 4, 8, 3)]:
    x = torch.randn(B, Ci, H, W, device=dev); gy = torch.randn(B, Co, H, W, device=dev)
    dw, db = wgrad(gy, x, Co, Ci, k)
    xd = x.double(); wd = torch.zeros(Co, Ci, k, k, device=dev, dtype=torch.float64, requires_grad=True); bd = torch.zeros(Co, device=dev, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xd, wd, bd, padding=k // 2) * gy.double()).sum().backward()
    print((B, Ci, Co, H, W, k), "dw err", float((dw.double() - wd.grad).abs().max() / wd.grad.abs().max()), "db err", float((db.double() - bd.grad).abs().max() / bd.grad.abs().max()))
for name, (B, Ci, Co, H) in dict(rcu128=(16, 256, 256, 128), rcu64=(16, 256, 256, 64), rcu32=(16, 256, 256, 32), head0_256=(16, 256, 128, 256), rn64=(16, 96, 256, 64)).items():
    x = torch.randn(B, Ci, H, H, device=dev); w = torch.randn(Co, Ci, 3, 3, device=dev); gy = torch.randn(B, Co, H, H, device=dev)
    fl = 2 * B * H * H * Co * Ci * 9
    t6 = timeit(lambda: wgrad(gy, x, Co, Ci, 3))
    tm = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, [Co], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True]))
    print(name, "x6", round(t6, 3), "ms", round(fl / t6 / 1e9, 1), "TF | miopen", round(tm, 3), "ms", round(fl / tm / 1e9, 1), "TF", flush=True)
