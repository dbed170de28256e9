"""Run-to-run spread of the tiny encoder forward per Linear mode: the same inputs, N forwards in one process, max |difference| of each
output against the first run, relative to the output's max |value| (the yardstick tests/gpu_utils.assert_close_rel uses)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from styl3r_amd import vit_ops
from tests.test_encoder import _build, G
from tests.helpers import deterministic_init_

dev, tag = "cuda:0", "sh1"
m = deterministic_init_(_build(1)).to(dev)
T = lambda k: torch.tensor(G[f"{tag}_{k}"], device=dev)
ctx, sty = dict(image=T("image"), intrinsics=T("intrinsics")), dict(image=T("style"))
for mode in ("bf16x6", "bf16x3", "f16x3"):
    vit_ops.LINEAR_MODE = mode
    runs = []
    with torch.no_grad():
        for i in range(8):
            gs = m(ctx, sty, global_step=0)
            runs.append({k: getattr(gs, k).double().cpu().numpy() for k in ("means", "covariances", "harmonics", "opacities")})
    ref = dict(means=G[f"{tag}_means"], covariances=G[f"{tag}_cov"], harmonics=G[f"{tag}_sh"], opacities=G[f"{tag}_opac"])
    for k in runs[0]:
        sc = np.abs(ref[k]).max()
        spread = max(np.abs(r[k] - runs[0][k]).max() for r in runs[1:]) / sc
        errs = [np.abs(r[k] - ref[k]).max() / sc for r in runs]
        print(f"{mode:7s} {k:12s} run-to-run {spread:.3e}   vs fixture min {min(errs):.3e} max {max(errs):.3e}")
