"""Helpers for the -m gpu parity tests: run the HIP path through the C ABI and
the CPU oracle on identical bytes."""
import os

import numpy as np
import torch

from oracle.gsr_oracle import Oracle
from styl3r_amd import rasterizer as rz


def ws_view(name, dtype, count):
    """typed view into the workspace of the last forward (rasterizer.KEEP_DEBUG must be on)."""
    dbg = rz.LAST_DEBUG
    off = getattr(dbg["layout"], name)
    nbytes = count * np.dtype(dtype).itemsize
    raw = dbg["ws"][off:off + nbytes].cpu().numpy()
    return raw.view(dtype)


def hip_single_view(means, cov6, opac, cam, shs=None, colors=None, bg=(0, 0, 0), sh_degree=0, theta=None, rho=None,
                    requires_grad=False):
    """One view through the drop-in GaussianRasterizer; inputs are numpy, returns dict of tensors."""
    dev = torch.device("cuda:0")
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    t = dict(means=f(means), cov6=f(cov6), opac=f(opac).reshape(-1, 1))
    t["colors"] = f(shs if shs is not None else colors)
    means2D = torch.zeros_like(t["means"])
    if requires_grad:
        for k in t:
            t[k].requires_grad_(True)
        means2D.requires_grad_(True)
    settings = rz.GaussianRasterizationSettings(
        image_height=cam["H"], image_width=cam["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=f(bg),
        scale_modifier=1.0, viewmatrix=f(cam["view"]), projmatrix=f(cam["proj"]), projmatrix_raw=f(cam["proj_raw"]),
        sh_degree=sh_degree, campos=f(cam["campos"]), prefiltered=False, debug=False)
    rast = rz.GaussianRasterizer(settings)
    image, radii, depth, opacity, n_touched = rast(
        means3D=t["means"], means2D=means2D, shs=t["colors"] if shs is not None else None,
        colors_precomp=None if shs is not None else t["colors"], opacities=t["opac"], cov3D_precomp=t["cov6"],
        theta=theta, rho=rho)
    return dict(image=image, radii=radii, depth=depth, opacity=opacity, n_touched=n_touched, inputs=t, means2D=means2D)


def oracle_single_view(precision, means, cov6, opac, cam, shs=None, colors=None, bg=(0, 0, 0), sh_degree=0, nthreads=8):
    orc = Oracle(precision)
    # feed the oracle the exact fp32 values the GPU sees
    r = lambda a: None if a is None else np.asarray(a, dtype=np.float32)
    st, ctx = orc.forward(r(means), r(cov6), r(opac), shs=r(shs), colors=r(colors), H=cam["H"], W=cam["W"],
                          tanfovx=np.float32(cam["tanfovx"]), tanfovy=np.float32(cam["tanfovy"]), bg=bg,
                          view=r(cam["view"]), proj=r(cam["proj"]), proj_raw=r(cam["proj_raw"]), campos=r(cam["campos"]),
                          sh_degree=sh_degree, nthreads=nthreads)
    return orc, st, ctx


def assert_close_rel(actual, expected, rel=1e-4, what="", atol=0.0):
    """max |a-e| <= rel * max|e| (+ tiny absolute floor): the 1e-4-rel bar of BASELINE.json's north_star."""
    a = np.asarray(actual, dtype=np.float64); e = np.asarray(expected, dtype=np.float64)
    scale = max(np.abs(e).max(), 1e-12)
    err = np.abs(a - e).max()
    if os.environ.get("PARITY_VERBOSE"):      # bar calibration runs: `PARITY_VERBOSE=1 pytest -s` lists every observed distance beside its bar
        print(f"    [parity] {what}: rel {err / scale:.3e} (bar {rel:.1e})")
    assert err <= rel * scale + atol, f"{what}: max abs err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e})"
