"""Generates tests/golden/decoder_boundary.npz by importing the REFERENCE's own
Python decoder wrapper (src/model/decoder/cuda_splatting.py render_cuda) in the
build container, with a recording stub in place of the absent third-party
`diff_gaussian_rasterization` module.  The fixture is data only: the inputs
handed to render_cuda and the exact per-view Settings / tensor arguments it
passes across the rasterizer boundary.  Run from the repo root:
    python tests/golden/make_decoder_fixtures.py
/root/reference is read here only; nothing of it travels to the GPU box.
"""
import sys
import types
from collections import namedtuple
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

# --- stubs -------------------------------------------------
jt = types.ModuleType("jaxtyping")
class _Sub:
    def __class_getitem__(cls, item):
        return cls
for n in ("Float", "Int64", "Bool", "UInt8", "Shaped", "Int"):
    setattr(jt, n, type(n, (_Sub,), {}))
sys.modules["jaxtyping"] = jt

CALLS = []
dgr = types.ModuleType("diff_gaussian_rasterization")
Settings = namedtuple("GaussianRasterizationSettings", "image_height image_width tanfovx tanfovy bg scale_modifier "
                      "viewmatrix projmatrix projmatrix_raw sh_degree campos prefiltered debug")
class Recorder(torch.nn.Module):
    def __init__(self, s):
        super().__init__(); self.s = s
    def forward(self, **kw):
        CALLS.append((self.s, {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in kw.items()}))
        h, w = self.s.image_height, self.s.image_width
        g = kw["means3D"].shape[0]
        return (torch.zeros(3, h, w), torch.zeros(g, dtype=torch.int32), torch.zeros(1, h, w), torch.zeros(1, h, w),
                torch.zeros(g, dtype=torch.int32))
dgr.GaussianRasterizationSettings = Settings
dgr.GaussianRasterizer = Recorder
sys.modules["diff_gaussian_rasterization"] = dgr

for name in ("src", "src.model", "src.model.decoder", "src.geometry", "src.dataset"):
    m = types.ModuleType(name); m.__path__ = [REF + "/" + name.replace(".", "/")]
    sys.modules[name] = m
import importlib
cs = importlib.import_module("src.model.decoder.cuda_splatting")

from styl3r_amd.scenes import make_scene

out = {}
for tag, scale_inv in (("si", True), ("raw", False)):
    CALLS.clear()
    sc = make_scene(n_ctx=1, grid_hw=(25, 40), n_views=3, image_hw=(64, 80), sh_degree=1, seed=5)
    # non-trivial intrinsics / near / far per view
    K = sc.intrinsics.clone(); K[1, 0, 0] = 0.9; K[2, 1, 1] = 0.8; K[2, 0, 2] = 0.52
    near = torch.tensor([0.1, 0.25, 0.5]); far = torch.tensor([100.0, 50.0, 80.0])
    bg = torch.tensor([[0.1, 0.2, 0.3]]).repeat(3, 1)
    rep = lambda t: t[None].repeat(3, *([1] * t.dim()))
    cs.render_cuda(sc.extrinsics, K, near, far, (64, 80), bg, rep(sc.means), rep(sc.covariances), rep(sc.harmonics),
                   rep(sc.opacities), scale_invariant=scale_inv)
    out[f"{tag}_in_extrinsics"] = sc.extrinsics.numpy(); out[f"{tag}_in_intrinsics"] = K.numpy()
    out[f"{tag}_in_near"] = near.numpy(); out[f"{tag}_in_far"] = far.numpy(); out[f"{tag}_in_bg"] = bg.numpy()
    out[f"{tag}_in_means"] = sc.means.numpy(); out[f"{tag}_in_cov"] = sc.covariances.numpy()
    out[f"{tag}_in_sh"] = sc.harmonics.numpy(); out[f"{tag}_in_opac"] = sc.opacities.numpy()
    for i, (s, kw) in enumerate(CALLS):
        p = f"{tag}_v{i}_"
        out[p + "tanfov"] = np.array([s.tanfovx, s.tanfovy], np.float64)
        for f in ("bg", "viewmatrix", "projmatrix", "projmatrix_raw", "campos"):
            out[p + f] = getattr(s, f).numpy()
        out[p + "sh_degree"] = np.array(s.sh_degree)
        for f in ("means3D", "shs", "opacities", "cov3D_precomp"):
            out[p + f] = kw[f].numpy()
np.savez_compressed(ROOT / "tests/golden/decoder_boundary.npz", **out)
print("wrote", len(out), "arrays")
