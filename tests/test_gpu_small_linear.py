"""-m gpu: the small-M Linear kernel (csrc/vit_gemm_sm.hip; batch-1 serving, C2: 257 / 514 token rows) through the C-ABI entry point
vit_linear_x6r_fwd (cfg 5) -- against float64, against the 128-row-tile kernel it replaces at these row counts (vit_linear_sm_set(0, ..)), every
tile / wave configuration, every activation code, the three arithmetic modes, ragged row counts, the published |max| word, run-to-run bits.
Shapes: croco/blocks.py:76-82, :97-134, :171-200 at the row counts of infer_model_re10k.py:262-560."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# error bars against float64, relative to the output's max-norm: fp32 round-off class for the six-product and the fp16-piece modes (the bars of
# test_fused_linear_vs_fp64), the three bf16 products' 2^-16 class for bf16x3
BAR = {"bf16x6": 4e-6, "f16x3": 4e-6, "bf16x3": 4e-5}


def _launch(vo, x, w, b, res, act, want_pre=False, want_word=False, small=None):
    """one Linear launch on the current stream, operands announced the way fused_linear does: the small-M kernel on the block image where
    it serves the shape (small=None: ask the library), else vit_linear_x6_fwd on the row image"""
    lib = vo.load()
    assert vo._x6()
    M, K = x.shape
    N = w.shape[0]
    if small is None:
        small = bool(lib.vit_linear_sm_ok(M, N, K))
    out = torch.empty(M, N, device=x.device)
    pre = torch.empty_like(out) if want_pre else None
    wp = vo.split_weight_block(w) if small else vo.split_weight(w)
    word = None
    if vo._f16():
        vo._announce(vo._amax_of(x))
    if want_word:
        word = vo._AMAX.word(x.device)
        vo._check(lib.vit_x6_set_output_amax(word.data_ptr()), "set_output_amax")
    args = (x.data_ptr(), wp.data_ptr(), b.data_ptr() if b is not None else None, res.data_ptr() if res is not None else None,
            out.data_ptr(), pre.data_ptr() if pre is not None else None, M, N, K, act)
    if small:
        vo._check(lib.vit_linear_x6r_fwd(*args, 5, vo._stream(x.device)), "vit_linear_x6r_fwd cfg 5")
    else:
        vo._check(lib.vit_linear_x6_fwd(*args, vo._stream(x.device)), "vit_linear_x6_fwd")
    return out, pre, word


def _ref64(x, w, b, res, act):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    pre = y
    if act == 2:
        return y * _gelu_grad64(res.double()), pre
    if act == 1:
        y = torch.nn.functional.gelu(y)
    if res is not None:
        y = y + res.double()
    return y, pre


def _gelu_grad64(p):
    return 0.5 * (1 + torch.erf(p / 2 ** 0.5)) + p * torch.exp(-0.5 * p * p) / (2 * np.pi) ** 0.5


def _rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.fixture
def vo(monkeypatch):
    from styl3r_amd import vit_ops
    yield vit_ops
    vit_ops.load().vit_linear_sm_set(vit_ops.SMALL_M_ROWS, 0, 0)
    vit_ops._SMALL_M_SET.clear()


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "bf16x3"])
@pytest.mark.parametrize("M,N,K", [(514, 3072, 1024), (257, 768, 3072), (514, 1024, 4096), (1, 64, 256), (33, 128, 512), (63, 192, 1024), (1000, 1024, 768), (257, 2304, 768)])
def test_small_m_kernel_vs_float64_and_vs_the_128_row_kernel_every_configuration(M, N, K, mode, vo, monkeypatch):
    monkeypatch.setattr(vo, "LINEAR_MODE", mode)
    g = torch.Generator(DEV).manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g)
    lib = vo.load()
    for act, use_res, use_b in ((0, True, True), (1, False, True), (1, True, True), (0, False, False), (2, True, False)):
        r = res if use_res else None
        bb = b if use_b else None
        ref, ref_pre = _ref64(x, w, bb, r, act)
        old, _, _ = _launch(vo, x, w, bb, r, act, small=False)               # the kernel this one replaces at small M
        served = 0
        for tm, nw in ((0, 0), (1, 4), (1, 8), (2, 4), (2, 8)):
            vo._check(lib.vit_linear_sm_set(1024, tm, nw), "sm_set")
            if not lib.vit_linear_sm_ok(M, N, K):                             # (K is not a multiple of 64 x waves)
                assert K % (64 * (nw or 4)) != 0 or (mode == "bf16x6" and (tm, nw) == (2, 8)), (M, N, K, tm, nw)   # (the one shape that is not built: LDS)
                continue
            served += 1
            got, pre, word = _launch(vo, x, w, bb, r, act, want_pre=(act == 1), want_word=True, small=True)
            e, e_old = _rel(got, ref), _rel(old, ref)
            assert e <= BAR[mode], (act, use_res, tm, nw, e)
            assert e <= 1.5 * e_old + 3e-7, (act, tm, nw, e, e_old)           # never worse than the kernel it replaces
            assert _rel(got, old.double()) <= 2 * BAR[mode]
            if pre is not None:
                assert _rel(pre, ref_pre) <= BAR[mode]
            # the published word: the exact |max| of the stored values (an integer max over bit patterns)
            assert int(word.max()) == int(got.abs().max().view(torch.int32)), (act, tm, nw)
            again, _, _ = _launch(vo, x, w, bb, r, act, small=True)
            assert torch.equal(again, got)                                     # fixed reduction order: run-to-run bits
            assert torch.isfinite(got).all()
        assert served >= 2


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
def test_serving_path_publishes_every_small_m_output_and_needs_no_amax_pass_behind_it(mode, vo, monkeypatch):
    """fused_linear under no_grad at C2's row counts: fc1 (GELU) -> fc2 (+ residual) -> qkv; in f16x3 the second and third layer find their input's
    |max| published by the layer in front (no vit_amax launch), and the chain equals the float64 chain"""
    monkeypatch.setattr(vo, "LINEAR_MODE", mode)
    g = torch.Generator(DEV).manual_seed(5)
    M, C = 514, 1024
    x = torch.randn(M, C, device=DEV, generator=g)
    w1 = torch.randn(4 * C, C, device=DEV, generator=g) / C ** 0.5; b1 = torch.randn(4 * C, device=DEV, generator=g)
    w2 = torch.randn(C, 4 * C, device=DEV, generator=g) / (4 * C) ** 0.5; b2 = torch.randn(C, device=DEV, generator=g)
    w3 = torch.randn(3 * C, C, device=DEV, generator=g) / C ** 0.5; b3 = torch.randn(3 * C, device=DEV, generator=g)
    with torch.no_grad():
        vo.fused_linear(x, w1, b1, gelu=True)                                 # (weight images cached)
        before = dict(vo.CALLS)
        h = vo.fused_linear(x, w1, b1, gelu=True)
        y = vo.fused_linear(h, w2, b2, residual=x)
        q = vo.fused_linear(y, w3, b3, amax_out=True)
    passes = vo.CALLS["amax_pass"] - before.get("amax_pass", 0)
    if mode == "f16x3":
        assert passes == 1, passes                                            # x only: h and y carry published words
        assert vo._known_amax(q) is not None
        assert int(vo._known_amax(q).max()) == int(q.abs().max().view(torch.int32))
    hd = torch.nn.functional.gelu(x.double() @ w1.double().t() + b1.double())
    yd = hd @ w2.double().t() + b2.double() + x.double()
    qd = yd @ w3.double().t() + b3.double()
    assert _rel(q, qd) <= 8e-6


def test_small_m_switch_off_restores_the_split_contraction_path(vo, monkeypatch):
    """VIT_SMALL_M_ROWS = 0 (A/B switch): the 128-row kernel with its zero fill runs again, results agree to fp32 round-off"""
    monkeypatch.setattr(vo, "LINEAR_MODE", "bf16x6")
    g = torch.Generator(DEV).manual_seed(9)
    x = torch.randn(257, 768, device=DEV, generator=g); w = torch.randn(768, 768, device=DEV, generator=g) / 27.7
    with torch.no_grad():
        a = vo.fused_linear(x, w)
        monkeypatch.setattr(vo, "SMALL_M_ROWS", 0)
        b = vo.fused_linear(x, w)
    assert not vo.small_m_kernel(257, 768, 768)
    monkeypatch.setattr(vo, "SMALL_M_ROWS", 1024)
    assert vo.small_m_kernel(257, 768, 768) and not vo.small_m_kernel(257, 768, 80) and not vo.small_m_kernel(2000, 768, 768)
    assert _rel(a, b.double()) <= 4e-6


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
def test_two_problem_launches_equal_the_single_launches_bit_for_bit(mode, vo, monkeypatch):
    """vit_linear_sm_grouped / vit_layernorm_fwd_grouped: group g of a two-problem launch = the one-problem launch on the same operands (same
    kernel, same tile walk; in f16x3 the shared |max| word of the stacked input is what the single launches are given too)"""
    from torch import nn
    monkeypatch.setattr(vo, "LINEAR_MODE", mode)
    g = torch.Generator(DEV).manual_seed(11)
    M, K, N = 257, 768, 2304
    x = torch.randn(2, M, K, device=DEV, generator=g)
    res = torch.randn(2, M, N, device=DEV, generator=g)
    layers = [nn.Linear(K, N).to(DEV), nn.Linear(K, N).to(DEV)]
    norms = [vo.LayerNorm(K, eps=1e-6).to(DEV), vo.LayerNorm(K, eps=1e-6).to(DEV)]
    for n in norms:
        n.weight.data.uniform_(0.5, 1.5, generator=g); n.bias.data.normal_(generator=g)
    lib = vo.load()
    with torch.no_grad():
        assert vo.grouped_ok(x, N)
        for flip in (False, True):
            for gelu, use_res in ((False, True), (True, False)):
                got = vo.grouped_linear(x, layers, residual=res if use_res else None, gelu=gelu, flip=flip)
                for gi in range(2):
                    xin = x[gi ^ int(flip)]
                    # the single launch with the STACKED tensor's |max| word (f16x3: the scale is part of the arithmetic)
                    out = torch.empty(M, N, device=DEV)
                    wp = vo.split_weight_block(layers[gi].weight)
                    if vo._f16():
                        vo._announce(vo._amax_of(x))
                    r = res[gi] if use_res else None
                    vo._check(lib.vit_linear_x6r_fwd(xin.data_ptr(), wp.data_ptr(), layers[gi].bias.data_ptr(), r.data_ptr() if r is not None else None,
                                                     out.data_ptr(), None, M, N, K, 1 if gelu else 0, 5, vo._stream(x.device)), "single")
                    assert torch.equal(got[gi], out), (flip, gelu, gi)
                if vo._f16():
                    assert int(vo._known_amax(got).max()) == int(got.abs().max().view(torch.int32))
            y = vo.grouped_layernorm(x, norms, flip=flip)
            for gi in range(2):
                ref = norms[gi](x[gi ^ int(flip)].contiguous())
                assert float((y[gi] - ref).abs().max()) <= 4e-7 * float(ref.abs().max()), (flip, gi)       # (same operations; the compiler contracts the affine step of the two kernels differently: <= 1 ulp)


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
def test_paired_decoder_layers_equal_the_two_decoder_path(mode, vo, monkeypatch):
    """encoder backbone, serving, two context views: layer i of both decoders as 14 two-problem launches (vit.decoder_blocks_pair) against the
    two DecoderBlock.forward calls -- same operations in the same order; in f16x3 the operand scales differ (one |max| word for the stacked
    tensor instead of one per decoder), so the comparison is to fp32 round-off, not to the bit"""
    from styl3r_amd import vit
    monkeypatch.setattr(vo, "LINEAR_MODE", mode)
    monkeypatch.setattr(vo, "ATTENTION_ARITH", "bf16x6" if mode == "bf16x6" else mode)
    torch.manual_seed(3)
    C, H, b, l = 256, 4, 1, 257
    rope = vit.RopeCfg(100.0, 64)
    blks = [vit.DecoderBlock(C, H, 4.0, qkv_bias=True, norm_layer=vit.LayerNorm6, rope=rope).to(DEV).eval() for _ in range(2)]
    for blk in blks:
        for n in (blk.norm1, blk.norm2, blk.norm3, blk.norm_y):
            n.weight.data.uniform_(0.5, 1.5); n.bias.data.normal_(0, 0.3)
    g = torch.Generator(DEV).manual_seed(4)
    x = torch.randn(2, b, l, C, device=DEV, generator=g)
    pos = torch.randint(0, 16, (2 * b, l, 2), device=DEV, generator=g)
    with torch.no_grad():
        assert vit.decoder_blocks_pair_ok(blks[0], blks[1], x)
        got = vit.decoder_blocks_pair(blks[0], blks[1], x, pos, torch.cat((pos[b:], pos[:b])))
        r0 = blks[0](x[0], x[1], pos[:b], pos[b:])[0]
        r1 = blks[1](x[1], x[0], pos[b:], pos[:b])[0]
    for gi, r in enumerate((r0, r1)):
        e = _rel(got[gi], r.double())
        assert e <= (1e-6 if mode == "bf16x6" else 3e-6), (gi, e)         # (bf16x6: the 1-ulp LayerNorm difference carried through the block)


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "bf16x3"])
@pytest.mark.parametrize("M,N,K,gelu_behind", [(2570, 768, 768, False), (2570, 768, 3072, True), (5140, 768, 1024, False), (2570, 3072, 768, False)])
def test_narrow_outputs_take_the_barrier_free_kernel_at_train_step_row_counts(M, N, K, gelu_behind, mode, vo, monkeypatch):
    """vit_ops.narrow_n_kernel: forward (N <= 768) and input-gradient (its N is the layer's K) launches of the decoders' 768-wide layers at
    M = 2 570 / 5 140 run csrc/vit_gemm_sm.hip; forward, dX (incl. the GELU' epilogue of an fc2 behind a GELU), dW, db against float64"""
    monkeypatch.setattr(vo, "LINEAR_MODE", mode)
    g = torch.Generator(DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g, requires_grad=True)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device=DEV, generator=g, requires_grad=True)
    gy = torch.randn(M, N, device=DEV, generator=g)
    pre = torch.randn(M, K, device=DEV, generator=g)
    link_in = None
    if gelu_behind:
        link_in = vo.GeluLink(); link_in.pre = pre
    before = vo.CALLS["linear_x6r"]
    y = vo.fused_linear(x, w, b, link_in=link_in)
    (y * gy).sum().backward()
    took = vo.CALLS["linear_x6r"] - before
    assert took >= (1 if N <= 768 else 0) + (1 if K <= 768 else 0), took        # (forward: N narrow; dX: its output width is K; the count includes the ring kernels of the wide layers)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.linear(xd, wd, bd)
    (ref * gy.double()).sum().backward()
    dx_ref = xd.grad * _gelu_grad64(pre.double()) if gelu_behind else xd.grad
    bar = BAR[mode]
    assert _rel(y.detach(), ref.detach()) <= bar
    assert _rel(x.grad, dx_ref) <= 3 * bar and _rel(w.grad, wd.grad) <= 3 * bar and _rel(b.grad, bd.grad) <= 3 * bar
