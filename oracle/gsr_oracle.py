"""ctypes/numpy front-end of the CPU oracle (oracle/gsr_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gsr_oracle.c.  Imported by
tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg only.
PARITY UNPINNED (third-party rasterizer absent from /root/reference; anchored
on src/model/decoder/cuda_splatting.py:101-129).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"


def build(force: bool = False) -> None:
    """Compile both precisions of the oracle with gcc (a few seconds)."""
    outs = [_BUILD / "libgsr_oracle_f32.so", _BUILD / "libgsr_oracle_f64.so"]
    src = _HERE / "gsr_oracle.c"
    if not force and all(o.exists() and o.stat().st_mtime >= src.stat().st_mtime for o in outs):
        return
    subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True,
                   stdout=subprocess.DEVNULL)


def _params_struct(real):
    class Params(C.Structure):
        _fields_ = [
            ("G", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("sh_degree", C.c_int32),
            ("M", C.c_int32), ("want_tau", C.c_int32),
            ("tanfovx", real), ("tanfovy", real), ("bg", real * 3),
            ("view", real * 16), ("proj", real * 16), ("proj_raw", real * 16), ("campos", real * 3),
        ]
    return Params


@dataclass
class FwdState:
    """Everything the forward produced (the oracle's geometry/binning/image state)."""
    depth: np.ndarray
    xy: np.ndarray
    conic_opacity: np.ndarray
    rgb: np.ndarray
    radii: np.ndarray
    tiles_touched: np.ndarray
    rect: np.ndarray
    clamped: np.ndarray
    R: int
    point_list: np.ndarray
    ranges: np.ndarray
    image: np.ndarray
    out_depth: np.ndarray
    out_opacity: np.ndarray
    final_T: np.ndarray
    n_contrib: np.ndarray
    n_touched: np.ndarray
    fragile: np.ndarray


class Oracle:
    def __init__(self, precision: str = "f32"):
        build()
        assert precision in ("f32", "f64")
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.creal = C.c_float if precision == "f32" else C.c_double
        self.lib = C.CDLL(str(_BUILD / f"libgsr_oracle_{precision}.so"))
        self.Params = _params_struct(self.creal)
        assert self.lib.gso_sizeof_real() == np.dtype(self.dtype).itemsize
        assert self.lib.gso_sizeof_params() == C.sizeof(self.Params)
        self.lib.gso_bin_sort.restype = C.c_int64

    # -- helpers ---------------------------------------------------------
    def _arr(self, a, dtype=None):
        return np.ascontiguousarray(a, dtype=dtype or self.dtype)

    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def make_params(self, G, H, W, tanfovx, tanfovy, bg, view, proj, proj_raw, campos, sh_degree, M, want_tau=False):
        p = self.Params()
        p.G, p.H, p.W, p.sh_degree, p.M, p.want_tau = int(G), int(H), int(W), int(sh_degree), int(M), int(want_tau)
        p.tanfovx, p.tanfovy = float(tanfovx), float(tanfovy)
        # round through the oracle's real type so f32 runs see the exact fp32 inputs
        for name, val, n in (("bg", bg, 3), ("view", view, 16), ("proj", proj, 16), ("proj_raw", proj_raw, 16),
                             ("campos", campos, 3)):
            flat = np.asarray(val, dtype=self.dtype).reshape(-1)
            assert flat.size == n, name
            setattr(p, name, (self.creal * n)(*[float(x) for x in flat]))
        return p

    # -- stages ----------------------------------------------------------
    def forward(self, means, cov6, opac, shs=None, colors=None, *, H, W, tanfovx, tanfovy, bg, view, proj,
                proj_raw=None, campos=(0, 0, 0), sh_degree=0, nthreads=1) -> tuple["FwdState", object]:
        means = self._arr(means).reshape(-1, 3)
        G = means.shape[0]
        cov6 = self._arr(cov6).reshape(G, 6)
        opac = self._arr(opac).reshape(G)
        if shs is not None:
            shs = self._arr(shs).reshape(G, -1, 3)
            M = shs.shape[1]
            col_in = shs
        else:
            M = 0
            col_in = self._arr(colors).reshape(G, 3)
        if proj_raw is None:
            proj_raw = np.eye(4)
        p = self.make_params(G, H, W, tanfovx, tanfovy, bg, view, proj, proj_raw, campos, sh_degree, M)
        self.lib.gso_set_threads(C.c_int(int(nthreads)))     # per-Gaussian stages (results do not depend on the count)
        dt = self.dtype
        depth = np.zeros(G, dt); xy = np.zeros((G, 2), dt); co = np.zeros((G, 4), dt); rgb = np.zeros((G, 3), dt)
        radii = np.zeros(G, np.int32); tt = np.zeros(G, np.int32); rect = np.zeros((G, 4), np.int32)
        clamped = np.zeros((G, 3), np.uint8)
        P = self._ptr
        self.lib.gso_preprocess(C.byref(p), P(means), P(cov6), P(opac), P(col_in), P(depth), P(xy), P(co), P(rgb),
                                P(radii), P(tt), P(rect), P(clamped))
        T = ((W + 15) // 16) * ((H + 15) // 16)
        R = int(tt.astype(np.int64).sum())
        point_list = np.zeros(max(R, 1), np.int32)
        ranges = np.zeros((T, 2), np.int32)
        r = self.lib.gso_bin_sort(C.byref(p), P(depth), P(tt), P(rect), C.c_int64(R), P(point_list), P(ranges))
        assert r == R, (r, R)
        image = np.zeros((3, H, W), dt); od = np.zeros((H, W), dt); oo = np.zeros((H, W), dt)
        fT = np.zeros((H, W), dt); nc = np.zeros((H, W), np.int32); nt = np.zeros(G, np.int32)
        frag = np.zeros((H, W), np.uint8)
        self.lib.gso_render_fwd(C.byref(p), P(point_list), P(ranges), P(xy), P(co), P(rgb), P(depth), P(image),
                                P(od), P(oo), P(fT), P(nc), P(nt), P(frag), C.c_int(nthreads))
        st = FwdState(depth, xy, co, rgb, radii, tt, rect, clamped, R, point_list[:R], ranges, image, od, oo, fT,
                      nc, nt, frag)
        ctx = dict(p=p, means=means, cov6=cov6, opac=opac, col_in=col_in, M=M)
        return st, ctx

    def backward(self, st: FwdState, ctx, dL_dimage, dL_ddepth_img=None, want_tau=False, nthreads=1):
        p = ctx["p"]
        p.want_tau = int(want_tau)
        self.lib.gso_set_threads(C.c_int(int(nthreads)))
        G, M = p.G, ctx["M"]
        dt = self.dtype
        P = self._ptr
        dL_dimage = self._arr(dL_dimage).reshape(3, p.H, p.W)
        dLd = self._arr(dL_ddepth_img).reshape(p.H, p.W) if dL_ddepth_img is not None else None
        d_mean2D = np.zeros((G, 2), dt); d_conic = np.zeros((G, 3), dt); d_opac = np.zeros(G, dt)
        d_rgb = np.zeros((G, 3), dt); d_depth = np.zeros(G, dt)
        pl = np.ascontiguousarray(st.point_list) if st.R > 0 else np.zeros(1, np.int32)
        self.lib.gso_render_bwd(C.byref(p), P(pl), P(st.ranges), P(st.xy), P(st.conic_opacity), P(st.rgb),
                                P(st.depth), P(st.final_T), P(st.n_contrib), P(dL_dimage), P(dLd), P(d_mean2D),
                                P(d_conic), P(d_opac), P(d_rgb), P(d_depth), C.c_int(nthreads))
        d_means = np.zeros((G, 3), dt); d_cov6 = np.zeros((G, 6), dt)
        d_shs = np.zeros((G, M, 3), dt) if M > 0 else np.zeros((G, 3), dt)
        d_tau = np.zeros((G, 6), dt)
        self.lib.gso_preprocess_bwd(C.byref(p), P(ctx["means"]), P(ctx["cov6"]), P(ctx["col_in"]), P(st.radii),
                                    P(st.clamped), P(d_mean2D), P(d_conic), P(d_rgb), P(d_depth), P(d_means),
                                    P(d_cov6), P(d_shs), P(d_tau))
        out = dict(means3D=d_means, means2D=np.concatenate([d_mean2D, np.zeros((G, 1), dt)], 1), cov6=d_cov6,
                   shs=d_shs, opacities=d_opac, conic=d_conic, rgb=d_rgb, depth=d_depth)
        if want_tau:
            out["tau_per_gaussian"] = d_tau
            out["rho"] = d_tau[:, :3].sum(0)
            out["theta"] = d_tau[:, 3:].sum(0)
        return out

    def sh_basis(self, deg, dirs):
        dirs = self._arr(dirs).reshape(-1, 3)
        n = (deg + 1) ** 2
        out = np.zeros((dirs.shape[0], 25), self.dtype)
        for i, d in enumerate(dirs):
            self.lib.gso_sh_basis(C.c_int(deg), self.creal(d[0]), self.creal(d[1]), self.creal(d[2]),
                                  self._ptr(out[i]))
        return out[:, :n]

    def sh_basis_grad(self, deg, d):
        n = (deg + 1) ** 2
        gx = np.zeros(25, self.dtype); gy = np.zeros(25, self.dtype); gz = np.zeros(25, self.dtype)
        self.lib.gso_sh_basis_grad(C.c_int(deg), self.creal(d[0]), self.creal(d[1]), self.creal(d[2]),
                                   self._ptr(gx), self._ptr(gy), self._ptr(gz))
        return gx[:n], gy[:n], gz[:n]
