"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol
include/gsr.h declares, the workspace layout is sane, and the host-side decoder
boundary reproduces the golden vectors captured from the reference's own
render_cuda (tests/golden/make_decoder_fixtures.py)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from styl3r_amd import _lib
from styl3r_amd.decoder import _TRIU, prepare_views

ROOT = Path(__file__).resolve().parents[1]
GOLD = np.load(ROOT / "tests/golden/decoder_boundary.npz")


@pytest.fixture(scope="module")
def lib():
    _lib.build_library()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = (ROOT / "include/gsr.h").read_text()
    declared = set(re.findall(r"\b(gsr_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gsr_version().decode().startswith("gsr-hip gfx950")


def test_struct_sizes_match_header():
    assert C.sizeof(_lib.GsrDims) == 40          # 8 x int32 + pointer
    header = (ROOT / "include/gsr.h").read_text()
    body = header[header.index("typedef struct GsrLayout {"):header.index("} GsrLayout;")]
    fields = re.findall(r"^\s*size_t\s+(\w+);", body, flags=re.M)
    assert fields == [n for n, _ in _lib.GsrLayout._fields_]          # same names, same order as include/gsr.h
    assert C.sizeof(_lib.GsrLayout) == len(fields) * C.sizeof(C.c_size_t)
    assert _lib.GSR_VIEW_FLOATS * 4 == 256
    body = header[header.index("typedef struct GsrFused {"):header.index("} GsrFused;")]
    fields = re.findall(r"(\w+);", body)
    assert fields == [n for n, _ in _lib.GsrFused._fields_]           # same names, same order as include/gsr.h
    assert C.sizeof(_lib.GsrFused) == 40                               # 2 pointers, float + padding, 2 pointers
    assert f"#define GSR_ID_MASK 0x{_lib.GSR_ID_MASK:08x}u" in header and f"#define GSR_QUAD_SHIFT {_lib.GSR_QUAD_SHIFT}" in header
    assert f"#define GSR_FLAG_BIN_BALLOT {_lib.GSR_FLAG_BIN_BALLOT} " in header


def test_workspace_layout_and_argument_checks(lib):
    d = _lib.GsrDims(10, 4, 65536, 256, 256, 1, 0, 0, None)
    L = _lib.workspace_layout(d, 1 << 22)
    offs = [getattr(L, n) for n, _ in L._fields_]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert L.tile_count - L.records >= 40 * 65536 * 48
    assert L.point_list - L.pairs >= (1 << 22) * 8
    for bad in (_lib.GsrDims(0, 4, 10, 16, 16, 1, 0, 0, None), _lib.GsrDims(1, 1, 10, 16, 16, 1, 1, 0, None),
                _lib.GsrDims(1, 1, 10, 16, 16, 0, 2, 0, None), _lib.GsrDims(1, 1, 10, 16, 16, 25, 5, 0, None)):
        out = _lib.GsrLayout()
        assert lib.gsr_workspace_layout(C.byref(bad), 100, C.byref(out)) == -1
    out = _lib.GsrLayout()
    assert lib.gsr_workspace_layout(C.byref(d), 0, C.byref(out)) == -1
    # null pointers are rejected before anything is launched (no GPU needed)
    assert lib.gsr_forward(C.byref(d), *([None] * 5), 1 << 20, None, 0, *([None] * 7)) == -1
    assert lib.gsr_backward(C.byref(d), *([None] * 4), 1 << 20, None, 0, *([None] * 9)) == -1
    assert lib.gsr_forward_fused(C.byref(d), *([None] * 5), 1 << 20, None, 0, *([None] * 8)) == -1
    assert lib.gsr_backward_fused(C.byref(d), *([None] * 4), 1 << 20, None, 0, *([None] * 10)) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_path_never_imports_the_oracle():
    for py in (ROOT / "styl3r_amd").rglob("*.py"):
        assert "oracle" not in py.read_text().replace("the oracle", "").replace("oracle's", "").replace("fp32 oracle", "") \
            or py.name == "scenes.py", py
    for py in (ROOT / "diff_gaussian_rasterization").rglob("*.py"):
        assert "oracle" not in py.read_text()


@pytest.mark.parametrize("tag,scale_inv", [("si", True), ("raw", False)])
def test_view_setup_matches_reference_render_cuda(tag, scale_inv):
    """prepare_views == the matrices / tanfov / campos the reference hands to the rasterizer
    (cuda_splatting.py:65-115), bit for bit on the same CPU torch ops."""
    g = lambda k: torch.from_numpy(GOLD[f"{tag}_in_{k}"])
    views = prepare_views(g("extrinsics"), g("intrinsics"), g("near"), g("far"), g("bg"), scale_inv).numpy()
    for v in range(3):
        p = f"{tag}_v{v}_"
        row = views[v]
        assert np.array_equal(row[0:16].reshape(4, 4), GOLD[p + "viewmatrix"])
        assert np.array_equal(row[16:32].reshape(4, 4), GOLD[p + "projmatrix"])
        assert np.array_equal(row[32:48].reshape(4, 4), GOLD[p + "projmatrix_raw"])
        assert np.array_equal(row[48:51], GOLD[p + "campos"])
        # the reference passes tanfov through .item() (fp32 -> python float): same fp32 value
        assert np.array_equal(row[51:53], GOLD[p + "tanfov"].astype(np.float32))
        assert np.array_equal(row[53:56], GOLD[p + "bg"])
        s = np.float32(row[56])
        # Gaussian-side arguments: the kernel folds the scale in; the products it forms are the
        # reference's pre-scaled tensors bit for bit (one fp32 multiply each)
        assert np.array_equal(GOLD[f"{tag}_in_means"] * s, GOLD[p + "means3D"])
        cov6 = GOLD[f"{tag}_in_cov"][:, _TRIU[0], _TRIU[1]]
        assert np.array_equal(cov6 * np.float32(s * s), GOLD[p + "cov3D_precomp"])
        assert np.array_equal(GOLD[f"{tag}_in_sh"].transpose(0, 2, 1), GOLD[p + "shs"])
        assert np.array_equal(GOLD[f"{tag}_in_opac"][:, None], GOLD[p + "opacities"])
        assert int(GOLD[p + "sh_degree"]) == 1


def test_vit_library_exports_every_declared_symbol():
    from styl3r_amd import vit_ops
    vit_ops.build_library()
    lib = vit_ops.load()
    header = (ROOT / "include/vit_ops.h").read_text()
    declared = set(re.findall(r"\b(vit_[a-z_0-9]+)\s*\(", header))
    assert declared == set(vit_ops.EXPORTS), declared ^ set(vit_ops.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vit_version().decode().startswith("vit-hip gfx950")
    # argument validation happens before any launch (no GPU needed)
    assert lib.vit_rope2d(None, None, None, None, 1, 1, 1, 64, 4, 0, 0, 0, 1.0, None) == -1
    assert lib.vit_linear_fwd(None, None, None, None, None, None, 1, 1, 16, 0, None) == -1
    a = vit_ops.VitAttnArgs(); a.B = a.H = a.Nq = a.Nk = 1
    assert lib.vit_attention_fwd(C.byref(a), None, None, None, None, None, None) == -1


def test_vit_ops_reject_cpu_tensors():
    from styl3r_amd import vit_ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        vit_ops.memory_efficient_attention(torch.zeros(1, 2, 1, 64), torch.zeros(1, 2, 1, 64), torch.zeros(1, 2, 1, 64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        vit_ops.RoPE2D()(torch.zeros(1, 1, 2, 64), torch.zeros(1, 2, 2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        vit_ops.fused_linear(torch.zeros(2, 16), torch.zeros(4, 16))


def test_f16x3_switches_and_struct_layouts_without_a_gpu():
    """the round-4 additions to the C ABI, as far as they can be exercised without a device: the products switch accepts 2 ("f16x3") and nothing
    else new; f16x3 launches without announced |max| words are refused BEFORE any launch; VitAdamChunk as optim.py packs it is the header's
    struct (8 + 8 + 8 + 8 + 8 + 4 + 4 + 8 bytes: the trailing `amax` pointer)"""
    import numpy as np
    from styl3r_amd import optim, vit_ops
    vit_ops.build_library()
    lib = vit_ops.load()
    assert lib.vit_x6_set_products(2) == 0 and lib.vit_x6_products() == 2
    assert lib.vit_x6_set_products(4) == -1 and lib.vit_x6_products() == 2
    one = C.c_void_p(16)                                           # (any non-null value: validation only, nothing is dereferenced)
    assert lib.vit_linear_x6_fwd(one, one, None, None, one, None, 16, 16, 16, 0, None) == -1          # no |max| word announced
    assert lib.vit_linear_x6_wgrad(one, one, one, None, 16, 16, 16, None) == -1
    assert lib.vit_conv_x6_fwd(one, one, None, None, one, 1, 16, 16, 8, 8, 3, 0, None) == -1
    assert lib.vit_x6_set_products(6) == 0
    assert lib.vit_amax(None, 4, one, None) == -1 and lib.vit_amax(one, 0, one, None) == -1
    assert lib.vit_split_weight_bytes(64, 32) == 64 * 32 * 6 + 8192          # pieces + the |max| word (64 slots, one per cache line)
    header = (ROOT / "include/vit_ops.h").read_text()
    body = re.search(r"typedef struct VitAdamChunk \{(.*?)\} VitAdamChunk;", header, re.S).group(1)
    fields = re.findall(r"\b(\w+);", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert fields == list(optim._CHUNK_DTYPE.names) == ["p", "g", "m", "v", "step", "n", "vec", "amax"]
    assert optim._CHUNK_DTYPE.itemsize == 56 and optim._CHUNK_DTYPE.fields["amax"][1] == 48
    assert optim._AMAX_WORDS == vit_ops._AmaxArena.LINE == 64 * 32


def test_published_maxima_registry_never_outlives_its_tensor():
    """vit_ops._PUBLISHED (f16x3): a word published for a tensor is found through any view of the same memory, and never after an in-place
    write (version counter) or after the producing tensor died (its memory may have been handed to another tensor)"""
    import gc
    from styl3r_amd import vit_ops
    vit_ops._PUBLISHED.clear()
    t = torch.randn(4, 6)
    word = torch.zeros(8, dtype=torch.int32)
    vit_ops._publish(t, word)
    assert vit_ops._known_amax(t) is word and vit_ops._known_amax(t.reshape(24)) is word and vit_ops._known_amax(t.view(6, 4)) is word
    assert vit_ops._known_amax(t[1:]) is None and vit_ops._known_amax(torch.randn(4, 6)) is None
    t.add_(1.0)
    assert vit_ops._known_amax(t) is None                                        # modified in place: the published maximum is stale
    u = torch.randn(3, 3)
    vit_ops._publish(u, word)
    key = (u.data_ptr(), u.numel())
    assert key in vit_ops._PUBLISHED
    del u
    gc.collect()
    assert key not in vit_ops._PUBLISHED                                          # the entry went with the tensor
    keep, vit_ops.PUBLISH_AMAX = vit_ops.PUBLISH_AMAX, False
    try:
        v = torch.randn(2, 2); vit_ops._publish(v, word)
        assert vit_ops._known_amax(v) is None                                     # the A/B switch
    finally:
        vit_ops.PUBLISH_AMAX = keep


def test_swap_reductions_add_both_results(tmp_path):
    """The composite backward's wave reductions use __builtin_amdgcn_permlane{32,16}_swap; a hipcc build was once seen to fold the builtin's
    result pair r[0] + r[1] into r[0] + r[0] (gsr_common.h).  This compiles gsr_backward.hip to gfx950 assembly with the product's flags
    (no GPU needed) and checks that every swap's two registers feed one v_add_f32, and that there are as many swaps as the two tile-kernel
    instantiations need (ten-value: 5 + 3, nine-value: 4 + 2)."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if Path("/opt/rocm/bin/hipcc").exists() else None)
    if hipcc is None:
        pytest.skip("hipcc not available")
    src = Path(_lib.__file__).resolve().parent / "csrc" / "gsr_backward.hip"
    out = tmp_path / "gsr_backward.s"
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", str(src), "-o", str(out)], check=True, capture_output=True)
    lines = out.read_text().splitlines()
    swaps = 0
    for i, line in enumerate(lines):
        m = re.search(r"v_permlane(?:32|16)_swap_b32\S* (v\d+), (v\d+)", line)
        if not m:
            continue
        swaps += 1
        regs = {m.group(1), m.group(2)}
        assert any((mm := re.search(r"v_add_f32\S* v\d+, (v\d+), (v\d+)", lines[k])) and {mm.group(1), mm.group(2)} == regs
                   for k in range(i + 1, min(i + 20, len(lines)))), f"swap at line {i}: no v_add_f32 of both results: {line.strip()}"
    assert swaps == 14, swaps


def test_small_m_linear_main_loop_keeps_its_loads_in_flight(tmp_path):
    """csrc/vit_gemm_sm.hip lives on its software pipeline: the first version was correct and 2 x slower than the kernel it replaced because hipcc
    hoisted the (pure) MFMAs over the scheduling barriers and sank every global load to its first use (DESIGN R6.7).  This compiles the file to
    gfx950 assembly with the product's flags (no GPU needed) and checks, in the main loop of the encoder-shape instantiation (64-row tiles,
    4 waves, f16x3): no workgroup barrier, no scratch, 48 MFMAs and 32 global loads per two-stage body, and a counted `s_waitcnt vmcnt(N)` with
    N >= 16 in front of the first MFMA -- at least two of the three load groups behind it are still in flight."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if Path("/opt/rocm/bin/hipcc").exists() else None)
    if hipcc is None:
        pytest.skip("hipcc not available")
    src = Path(_lib.__file__).resolve().parent / "csrc" / "vit_gemm_sm.hip"
    out = tmp_path / "vit_gemm_sm.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fvisibility=hidden", "-S", "--cuda-device-only", str(src), "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    start = text.index("_ZN3vit2sm11k_linear_smILi2ELi4ELi2EEEvNS0_6SmArgsE:")
    body = text[start:text.index("s_endpgm", start)]
    assert "scratch_" not in body
    loop = body[body.index("Inner Loop Header"):]
    loop = loop[:re.search(r"s_cbranch_\w+ \.LBB\d+_\d+", loop).end()]
    assert "s_barrier" not in loop
    assert len(re.findall(r"v_mfma_f32_32x32x16_f16", loop)) == 48
    assert len(re.findall(r"global_load_dwordx4", loop)) == 32
    first_mfma = loop.index("v_mfma_f32_32x32x16_f16")
    waits = [int(n) for n in re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop[:first_mfma])]
    assert waits and min(waits) >= 16, waits


def test_derived_weight_cache_follows_the_parameter_version_and_stays_in_the_graph_when_training():
    """vit_ops.derived_weight (serving: reshaped / permuted / padded weights are built once per parameter version so that their split images stay
    cached; training: the derivation is part of the autograd graph)"""
    import torch
    from styl3r_amd.vit_ops import derived_weight
    w = torch.nn.Parameter(torch.arange(12.0).reshape(3, 4))
    fn = lambda t: t.t().contiguous()
    with torch.no_grad():
        a = derived_weight(w, "t", fn)
        assert derived_weight(w, "t", fn) is a and torch.equal(a, w.detach().t())
        assert derived_weight(w, "other", lambda t: t.reshape(4, 3)) is not a                      # another tag, another tensor
        w.add_(1.0)                                                                                 # an optimizer step: the version moves
        b = derived_weight(w, "t", fn)
        assert b is not a and torch.equal(b, w.detach().t())
    c = derived_weight(w, "t", fn)                                                                  # grad mode, trainable: recomputed, differentiable
    assert c.requires_grad and c is not b
    c.sum().backward()
    assert torch.equal(w.grad, torch.ones_like(w))
    w.requires_grad_(False)                                                                         # frozen (C4 style stage): the cache serves grad mode too
    assert derived_weight(w, "t", fn) is b
    assert derived_weight(None, "t", fn) is None


def test_edge_pixel_rule_of_the_e2e_image_comparison():
    """tests/test_e2e_parity.py::_edge_pixels: a pixel above the bar is set aside only if it touches the generator's fragile mask; more than four,
    or one that touches nothing, or one off by more than 10 % of the image scale is an error"""
    import numpy as np
    from tests.test_e2e_parity import _edge_pixels
    H = W = 16
    ref = np.ones((1, 2, 3, H, W)); ref[0, 0, 0, 0, 0] = 2.0
    ok = np.ones((1, 2, 1, H, W), bool); ok[0, 0, 0, 5, 5] = False                                  # one fragile pixel
    got = ref.copy(); got[0, 0, 1, 4, 6] += 0.01                                                    # its diagonal neighbour is off by 5e-3 of the scale
    rest, aside = _edge_pixels(got, ref, ok, bar=1e-3)
    assert rest == 0.0 and len(aside) == 1 and aside[0][:3] == (0, 4, 6) and abs(aside[0][3] - 0.005) < 1e-12 and aside[0][4] is True
    got[0, 0, 0, 5, 5] += 1.0                                                                        # the masked pixel itself never counts
    assert _edge_pixels(got, ref, ok, bar=1e-3)[0] == 0.0
    far = ref.copy(); far[0, 1, 0, 10, 10] += 0.01                                                   # touches no fragile pixel
    with pytest.raises(AssertionError):
        _edge_pixels(far, ref, ok, bar=1e-3)
    big = ref.copy(); big[0, 0, 1, 4, 6] += 0.5                                                      # 25 % of the scale
    with pytest.raises(AssertionError):
        _edge_pixels(big, ref, ok, bar=1e-3)
    many = ref.copy()
    for dy, dx in ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1)):
        many[0, 0, 2, 5 + dy, 5 + dx] += 0.01
    with pytest.raises(AssertionError):
        _edge_pixels(many, ref, ok, bar=1e-3)
