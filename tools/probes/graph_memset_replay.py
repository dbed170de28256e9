import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd import vit_ops
dev = "cuda:0"
torch.manual_seed(0)
for (M, N, K, bias) in ((256, 1024, 768, True), (256, 1024, 768, False), (514, 1024, 1024, True), (514, 1024, 1024, False), (256, 1024, 1024, False), (512, 1024, 768, False)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev) if bias else None
    with torch.no_grad():
        for _ in range(2): ref = vit_ops.fused_linear(x, w, b)
        torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = vit_ops.fused_linear(x, w, b)
        g.replay(); torch.cuda.synchronize(); e1 = float((y - ref).abs().max())
        g.replay(); torch.cuda.synchronize()
        d = (y - ref)
        bad = d.abs() > 1e-3
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        ratio = (y[bad] / ref[bad]).median().item() if bad.any() else None
        print((M, N, K, bias), "replay1 err", e1, "| replay2: bad frac", float(bad.float().mean()), "rows", (int(rows.min()), int(rows.max())) if len(rows) else None,
              "cols", (int(cols.min()), int(cols.max())) if len(cols) else None, "median y/ref", ratio, "splits", vit_ops.load().vit_x6_products())
