import sys; sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from styl3r_amd import vit_ops
dev = "cuda:0"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
# correctness (small, ragged) vs fp64
for (B, Ci, Co, H, W, k, relu) in [(2, 32, 40, 9, 13, 3, False), (1, 16, 130, 17, 5, 3, True), (3, 48, 128, 8, 8, 1, False)]:
    x = torch.randn(B, Ci, H, W, device=dev); w = torch.randn(Co, Ci, k, k, device=dev) * 0.1; b = torch.randn(Co, device=dev); r = torch.randn(B, Co, H, W, device=dev)
    got = vit_ops.conv_x6_forward(x, w, b, r, relu)
    xi = F.relu(x) if relu else x
    ref = F.conv2d(xi.double(), w.double(), b.double(), padding=k // 2) + r.double()
    m = F.conv2d(xi, w, b, padding=k // 2) + r
    print("shape", (B, Ci, Co, H, W, k, relu), "x6 err", float((got.double() - ref).abs().max() / ref.abs().max()), "miopen err", float((m.double() - ref).abs().max() / ref.abs().max()))
for name, (B, Ci, Co, H) in dict(rcu128=(16, 256, 256, 128), rcu64=(16, 256, 256, 64), rcu32=(16, 256, 256, 32), head0_256=(16, 256, 128, 256), rn64=(16, 96, 256, 64)).items():
    x = torch.randn(B, Ci, H, H, device=dev); w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.02
    fl = 2 * B * H * H * Co * Ci * 9
    wp = vit_ops.split_conv_weight(w)
    t6 = timeit(lambda: vit_ops.conv_x6_forward(x, w, packed=wp)); tm = timeit(lambda: F.conv2d(x, w, padding=1))
    got = vit_ops.conv_x6_forward(x, w, packed=wp); ref = F.conv2d(x, w, padding=1)
    print(name, "x6", round(t6, 3), "ms", round(fl / t6 / 1e9, 1), "TF | miopen", round(tm, 3), "ms", round(fl / tm / 1e9, 1), "TF | max diff", float((got - ref).abs().max() / ref.abs().max()), flush=True)
