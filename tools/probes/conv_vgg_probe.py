"""VGG19 (relu1_1..relu4_1) convolution shapes: MIOpen (Winograd) vs vit_conv_x6_fwd, forward and input gradient.
usage: python tools/probes/conv_vgg_probe.py [batch]"""
import sys; sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from styl3r_amd import vit_ops
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
def timeit(fn, iters=10, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
_w = torch.randn(4096, 4096, device=dev)
for _ in range(600): _w @ _w
for name, (Ci, Co, H) in dict(c1_2=(64, 64, 256), c2_1=(64, 128, 128), c2_2=(128, 128, 128), c3_1=(128, 256, 64), c3_2=(256, 256, 64), c4_1=(256, 512, 32)).items():
    x = torch.randn(B, Ci, H, H, device=dev); w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.02; g = torch.randn(B, Co, H, H, device=dev)
    fl = 2 * B * H * H * Co * Ci * 9
    wp = vit_ops.split_conv_weight(w); wpt = vit_ops.split_conv_weight(w, True)
    dx = torch.empty_like(x)
    lib = vit_ops.load(); st = vit_ops._stream(x.device)
    t6 = timeit(lambda: vit_ops.conv_x6_forward(x, w, packed=wp)); tm = timeit(lambda: F.conv2d(x, w, padding=1))
    t6b = timeit(lambda: lib.vit_conv_x6_fwd(g.data_ptr(), wpt.data_ptr(), None, None, dx.data_ptr(), B, Co, Ci, H, H, 3, 0, st))
    tmb = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    print(name, f"fwd x6 {t6:.3f} ms {fl / t6 / 1e9:.0f} TF | miopen {tm:.3f} ms {fl / tm / 1e9:.0f} TF || dX x6 {t6b:.3f} ms {fl / t6b / 1e9:.0f} TF | miopen {tmb:.3f} ms {fl / tmb / 1e9:.0f} TF", flush=True)
