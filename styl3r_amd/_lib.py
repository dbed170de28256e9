"""Loader / builder of the HIP C-ABI library (include/gsr.h).

The library is plain C ABI (no torch types) and is bound here with ctypes --
the same stub a maintainer of the reference would add.
There is NO fallback: if the library is missing the import of the product
path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_CSRC = _PKG / "csrc"
_LIBDIR = _PKG / "lib"
# GSR_LIB_NAME / GSR_HIPCC_EXTRA: kernel-experiment builds (tools/ only); the product is libgsr_hip.so
LIB_PATH = _LIBDIR / os.environ.get("GSR_LIB_NAME", "libgsr_hip.so")

_SOURCES = ["gsr_forward.hip", "gsr_backward.hip", "gsr_api.hip", "gsr_loss.hip"]
_HEADERS = ["gsr_common.h", "../../include/gsr.h"]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden",
    # hipcc's SLP vectoriser packs independent fp32 ops of the composite loops into v_pk_*_f32 plus the
    # v_mov shuffles that feed them: measured -22 % on k_composite_bwd without it (2.11 -> 1.65 ms)
    "-fno-slp-vectorize",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


def sources_digest(paths, cmd) -> str:
    """sha256 over the command line and the bytes of every source / header: the build stamp (a stale shipped .so whose
    sources changed -- even with a newer mtime -- is rebuilt) and the key that ties a PMC profile to the build it measured."""
    import hashlib
    h = hashlib.sha256(" ".join(map(str, cmd)).encode())
    for q in sorted(map(str, paths)):
        h.update(Path(q).name.encode())
        h.update(Path(q).read_bytes())
    return h.hexdigest()[:16]


def _build_cmd():
    srcs = [_CSRC / s for s in _SOURCES]
    deps = srcs + [(_CSRC / h).resolve() for h in _HEADERS]
    # the command is recorded with repo-relative paths so that the digest is the same here and on the GPU box
    cmd = ["hipcc", *HIPCC_FLAGS, *os.environ.get("GSR_HIPCC_EXTRA", "").split(), *_SOURCES, "-o", LIB_PATH.name]
    return srcs, deps, cmd


def build_digest() -> str:
    """digest of the sources + flags the CURRENT tree would build libgsr_hip.so from"""
    _, deps, cmd = _build_cmd()
    return sources_digest(deps, cmd)


def built_digest() -> str:
    """digest recorded when the shipped libgsr_hip.so was built ('' if unknown)"""
    stamp = LIB_PATH.with_suffix(".stamp")
    return stamp.read_text().strip() if stamp.exists() and LIB_PATH.exists() else ""


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 into styl3r_amd/lib/libgsr_hip.so (cross-compiles without a GPU).  The library is
    rebuilt whenever the digest of (sources, headers, command line) differs from the one stamped next to it."""
    srcs, deps, cmd = _build_cmd()
    _LIBDIR.mkdir(exist_ok=True)
    want = sources_digest(deps, cmd)
    if not force and built_digest() == want:
        return LIB_PATH
    real = [_hipcc(), *cmd[1:-2], "-o", str(LIB_PATH)]
    if verbose:
        print(" ".join(real))
    subprocess.run(real, check=True, cwd=str(_CSRC))
    LIB_PATH.with_suffix(".stamp").write_text(want)
    return LIB_PATH


class GsrDims(C.Structure):
    _fields_ = [("B", C.c_int32), ("Vt", C.c_int32), ("G", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("M", C.c_int32), ("sh_degree", C.c_int32), ("flags", C.c_int32), ("profile", C.c_void_p)]


class GsrLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("records", "tile_count", "tile_offset", "tile_cursor", "pairs", "point_list",
                                          "final_T", "n_contrib", "grad_rec", "status", "tile_order", "pairs_alt", "loss_partial", "loss_ticket", "loss_diff", "total")]


class GsrFused(C.Structure):
    """include/gsr.h GsrFused: persistent tile counters + LossMse fused into the composite kernels"""
    _fields_ = [("tile_count", C.c_void_p), ("mse_target", C.c_void_p), ("mse_weight", C.c_float), ("mse_loss", C.c_void_p),
                ("mse_grad_loss", C.c_void_p)]


GSR_FLAG_NTOUCHED = 1
GSR_FLAG_COV9 = 2
GSR_FLAG_PHASE_BIN = 4
GSR_FLAG_PHASE_RENDER = 8
GSR_FLAG_PREZERO_GRADS = 16
GSR_FLAG_BIN_BALLOT = 32
GSR_ID_MASK = 0x0FFFFFFF
GSR_QUAD_SHIFT = 28
GSR_FLAG_SORT_KEYS_SHIFT = 8
GSR_STATUS_WORDS = 8
GSR_VIEW_FLOATS = 64
GSR_N_STAGES = 7
STAGE_NAMES = ("preprocess", "scan_tiles", "scatter", "tile_sort", "composite_fwd", "composite_bwd", "preprocess_bwd")
EXPORTS = ("gsr_workspace_layout", "gsr_forward", "gsr_backward", "gsr_forward_fused", "gsr_backward_fused", "gsr_version", "gsr_profile_create",
           "gsr_profile_destroy", "gsr_profile_read", "gsr_profile_set_stages", "gsr_last_error", "gsr_build_views", "gsr_mse_scratch_bytes",
           "gsr_mse_forward", "gsr_mse_backward")
ERRORS = {-1: "GSR_EINVAL (bad dimension / null pointer / unsupported degree)",
          -2: "GSR_ENOSPACE (workspace too small)", -3: "GSR_ELAUNCH (kernel launch failed)"}

_lib = None


def load() -> C.CDLL:
    """dlopen the library and declare the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP rasterizer has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
    # The library must share ONE HIP runtime with the process that owns the device pointers and
    # streams it is handed.  torch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's):
    # load torch's first so the loader resolves our NEEDED entry to it instead of a second copy
    # (two runtimes => hipErrorNoDevice inside the library).
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    vp, i64, sz = C.c_void_p, C.c_int64, C.c_size_t
    lib.gsr_workspace_layout.argtypes = [C.POINTER(GsrDims), i64, C.POINTER(GsrLayout)]
    lib.gsr_workspace_layout.restype = C.c_int
    lib.gsr_forward.argtypes = [C.POINTER(GsrDims), vp, vp, vp, vp, vp, i64, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_forward.restype = C.c_int
    lib.gsr_backward.argtypes = [C.POINTER(GsrDims), vp, vp, vp, vp, i64, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_backward.restype = C.c_int
    lib.gsr_forward_fused.argtypes = [C.POINTER(GsrDims), vp, vp, vp, vp, vp, i64, vp, sz, vp, vp, vp, vp, vp, vp, C.POINTER(GsrFused), vp]
    lib.gsr_forward_fused.restype = C.c_int
    lib.gsr_backward_fused.argtypes = [C.POINTER(GsrDims), vp, vp, vp, vp, i64, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(GsrFused), vp]
    lib.gsr_backward_fused.restype = C.c_int
    lib.gsr_version.restype = C.c_char_p
    lib.gsr_build_views.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    lib.gsr_build_views.restype = C.c_int
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_mse_scratch_bytes.argtypes = []
    lib.gsr_mse_scratch_bytes.restype = C.c_size_t
    lib.gsr_mse_forward.argtypes = [vp, vp, i64, C.c_float, vp, vp, vp]
    lib.gsr_mse_forward.restype = C.c_int
    lib.gsr_mse_backward.argtypes = [vp, vp, vp, i64, C.c_float, vp, vp]
    lib.gsr_mse_backward.restype = C.c_int
    lib.gsr_profile_create.argtypes = [C.c_int]
    lib.gsr_profile_create.restype = C.c_void_p
    lib.gsr_profile_destroy.argtypes = [C.c_void_p]
    lib.gsr_profile_destroy.restype = None
    lib.gsr_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.gsr_profile_set_stages.argtypes = [C.c_void_p, C.c_uint32]
    lib.gsr_profile_set_stages.restype = C.c_int
    lib.gsr_profile_read.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        detail = load().gsr_last_error().decode() if rc == -3 else ""
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)} {detail}")


def workspace_layout(dims: GsrDims, capacity: int) -> GsrLayout:
    L = GsrLayout()
    check(load().gsr_workspace_layout(C.byref(dims), int(capacity), C.byref(L)), "gsr_workspace_layout")
    return L


class StageProfile:
    """hipEvent stage timer of the library (include/gsr.h GsrProfile); pass `.handle` via rasterizer.PROFILE."""

    def __init__(self, max_calls: int):
        self.handle = load().gsr_profile_create(int(max_calls))
        if not self.handle:
            raise RuntimeError("gsr_profile_create failed")

    def set_stages(self, names=None):
        """time only the named stages (None = all): every timed stage costs two event records between otherwise back-to-back kernels"""
        mask = (1 << GSR_N_STAGES) - 1 if names is None else sum(1 << STAGE_NAMES.index(n) for n in names)
        check(load().gsr_profile_set_stages(self.handle, mask), "gsr_profile_set_stages")

    def read(self) -> dict:
        ms = (C.c_float * GSR_N_STAGES)()
        cnt = (C.c_int32 * GSR_N_STAGES)()
        check(load().gsr_profile_read(self.handle, ms, cnt), "gsr_profile_read")
        return {STAGE_NAMES[i]: (float(ms[i]), int(cnt[i])) for i in range(GSR_N_STAGES)}

    def close(self):
        if self.handle:
            load().gsr_profile_destroy(self.handle)
            self.handle = None
