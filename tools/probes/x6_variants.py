"""Time vit_linear_x6_fwd of several builds of libvit (kernel experiments compiled with -DX6_EXP=n).
usage: python tools/probes/x6_variants.py lib1.so lib2.so ...   (libs built by tools/probes/x6_build.sh)"""
import ctypes as C, sys, json
import torch
dev = torch.device("cuda:0")
shapes = dict(qkv=(5140, 3072, 1024), fc1=(5140, 4096, 1024), fc2=(5140, 1024, 4096), proj=(5140, 1024, 1024), dec=(5140, 768, 768),
              inf_qkv=(514, 3072, 1024))
vp = C.c_void_p
# warm the clocks first: the first ~100 ms after idle run at a lower frequency (the first library listed used to look 10-15 % slow)
_w = torch.randn(4096, 4096, device=dev)
for _ in range(1500): _w @ _w
torch.cuda.synchronize()
for path in sys.argv[1:]:
    lib = C.CDLL(path)
    lib.vit_split_weight_bytes.restype = C.c_size_t
    lib.vit_split_weight.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.vit_linear_x6_fwd.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    res = {}
    for name, (M, N, K) in shapes.items():
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        wp = torch.empty(lib.vit_split_weight_bytes(N, K), dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.vit_split_weight(x.data_ptr() * 0 + w.data_ptr(), wp.data_ptr(), N, K, 0, st) == 0
        f = lambda: lib.vit_linear_x6_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, M, N, K, 0, st)
        for _ in range(30): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        res[name] = (round(2 * M * N * K / ms / 1e9, 1), f"{err:.1e}")
    print(path.split("/")[-1], json.dumps(res))
