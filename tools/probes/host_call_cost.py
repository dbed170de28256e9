import torch, time
torch.zeros(1, device="cuda")
def t(f, n=20000):
    t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e6
print("is_current_stream_capturing us", t(torch.cuda.is_current_stream_capturing))
print("raw stream us", t(lambda: torch._C._cuda_getCurrentRawStream(0)))
print("current_stream us", t(lambda: torch.cuda.current_stream()))
