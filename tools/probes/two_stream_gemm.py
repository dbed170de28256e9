"""Do the dX and dW GEMMs of one Linear layer run faster concurrently on two streams (fill each other's idle slots)?"""
import sys; sys.path.insert(0, ".")
import ctypes as C, torch
from styl3r_amd import vit_ops
dev = torch.device("cuda:0"); lib = vit_ops.load()
side = torch.cuda.Stream(dev)
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
for name, (M, N, K) in dict(proj=(4112, 1024, 1024), fc2=(4112, 1024, 4096), qkv=(4112, 3072, 1024), dec=(4112, 768, 768)).items():
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; gy = torch.randn(M, N, device=dev)
    wT = vit_ops.split_weight(w, True)
    dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    def seq():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.vit_linear_x6_fwd(gy.data_ptr(), wT.data_ptr(), None, None, dx.data_ptr(), None, M, K, N, 0, st)
        lib.vit_linear_x6_wgrad(gy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, st)
    def par():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        lib.vit_linear_x6_wgrad(gy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, C.c_void_p(side.cuda_stream))
        lib.vit_linear_x6_fwd(gy.data_ptr(), wT.data_ptr(), None, None, dx.data_ptr(), None, M, K, N, 0, C.c_void_p(cur.cuda_stream))
        cur.wait_stream(side)
    print(name, "sequential %.4f ms   two streams %.4f ms" % (timeit(seq), timeit(par)), flush=True)
