"""-m gpu: the ViT kernels (through libvit_hip.so) against the numpy oracle and the golden vectors
captured from the reference's own RoPE2D / Attention / Block / CrossAttention / DecoderBlock."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import vit_oracle as vo
from tests.gpu_utils import assert_close_rel

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).resolve().parent / "golden" / "vit_blocks.npz")
DEV = "cuda:0"


def T(a, **kw):
    return torch.tensor(np.asarray(a), device=DEV, **kw)


def test_rope_kernel_matches_reference_golden_and_is_inplace_on_views():
    from styl3r_amd.vit_ops import RoPE2D
    tok = T(G["rope_tokens"]).requires_grad_(True)       # (B,H,N,D) contiguous
    pos = T(G["rope_pos"])
    rope = RoPE2D(100.0, max_pos=16)
    work = tok.clone()
    out = rope(work, pos)
    assert out.data_ptr() == work.data_ptr()             # in place like cuRoPE2D
    assert_close_rel(out.detach().cpu().numpy(), G["rope_out"], 2e-6, "rope fwd vs reference")
    (out * T(G["rope_gout"])).sum().backward()
    assert_close_rel(tok.grad.cpu().numpy(), G["rope_gin"], 2e-6, "rope bwd vs reference")
    # bit-identical to the reference fallback's formula evaluated with torch on the SAME device
    from styl3r_amd.vit_ops import rope_tables
    cos, sin = rope_tables(64, 17, 100.0, torch.device(DEV))
    t = T(G["rope_tokens"])
    def half(x, p):
        c = torch.cat((cos, cos), -1)[p][:, None]; s = torch.cat((sin, sin), -1)[p][:, None]
        x1, x2 = x[..., :16], x[..., 16:]
        return x * c + torch.cat((-x2, x1), -1) * s
    ref = torch.cat((half(t[..., :32], pos[:, :, 0]), half(t[..., 32:], pos[:, :, 1])), -1)
    assert torch.equal(rope(t.clone(), pos), ref)
    # strided view of a qkv buffer: (B,N,3,H,D) -> q view (B,H,N,D)
    qkv = torch.randn(2, 21, 3, 4, 64, device=DEV)
    before = qkv.clone()
    qv = qkv[:, :, 0].transpose(1, 2)
    rope(qv, pos)
    want = vo.rope2d(before[:, :, 0].cpu().numpy(), pos.cpu().numpy())
    assert_close_rel(qkv[:, :, 0].cpu().numpy(), want, 2e-6, "rope on qkv view")
    assert torch.equal(qkv[:, :, 1:], before[:, :, 1:])   # k, v untouched


@pytest.fixture(params=["bf16x6", "f32", "f16x3"])
def attention_arith(request, monkeypatch):
    """every kernel family, forward and backward: bf16x6 split arithmetic (csrc/vit_attention_x6.hip, vit_attention_bwd_x6.hip; the
    default), exact-f32 MFMA (csrc/vit_attention.hip, vit_attention_bwd.hip), and f16x3 (round 6: the split kernels on two fp16 pieces per
    operand and three products, operand scales from |max| words, per-lane running scale for dS) -- all against the same oracle bars"""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "ATTENTION_ARITH", request.param)
    yield request.param
    assert vit_ops.load().vit_attention_arith() == {"bf16x6": 1, "f32": 0, "f16x3": 3}[request.param]     # the launch really took that kernel


@pytest.mark.parametrize("B,H,Nq,Nk,rope", [(2, 3, 257, 257, True), (1, 2, 130, 771, False), (3, 12, 514, 256, True)])
def test_attention_three_product_mode_forward_and_backward(B, H, Nq, Nk, rope, monkeypatch):
    """vit_attention_set_arith(2) ("bf16x3": the split-arithmetic attention kernels without the three 2^-16-level products, third bf16
    pieces neither computed nor staged): forward and all three gradients against float64, at the three-product accuracy (~1e-5: two
    orders tighter than a TF32 contraction), and measurably different from the six-product result (the mode really took)."""
    from styl3r_amd import vit_ops
    g = torch.Generator(DEV).manual_seed(Nq + Nk)
    q0 = torch.randn(B, Nq, H, 64, device=DEV, generator=g); k0 = torch.randn(B, Nk, H, 64, device=DEV, generator=g)
    v0 = torch.randn(B, Nk, H, 64, device=DEV, generator=g); go = torch.randn(B, Nq, H, 64, device=DEV, generator=g)
    qpos = (torch.arange(Nq, device=DEV)[None, :, None].expand(B, -1, 2) % 17).contiguous() if rope else None
    kpos = (torch.arange(Nk, device=DEV)[None, :, None].expand(B, -1, 2) % 17).contiguous() if rope else None
    res = {}
    for mode in ("bf16x3", "bf16x6", "f16x3"):
        monkeypatch.setattr(vit_ops, "ATTENTION_ARITH", mode)
        q, k, v = (t.clone().requires_grad_(True) for t in (q0, k0, v0))
        o = vit_ops.memory_efficient_attention(q, k, v, scale=0.125, qpos=qpos, kpos=kpos, rope_base=100.0, max_pos=16)
        assert vit_ops.load().vit_attention_arith() == {"bf16x3": 2, "bf16x6": 1, "f16x3": 3}[mode]
        (o * go).sum().backward()
        res[mode] = [t.detach().double().cpu() for t in (o, q.grad, k.grad, v.grad)]
    # float64 reference
    qd, kd, vd = (t.double().cpu().requires_grad_(True) for t in (q0, k0, v0))
    def ref_attn(qq, kk, vv):
        if rope:
            def rot64(t, pos):
                out = t.clone()
                base = 100.0
                inv = 1.0 / (base ** (torch.arange(0, 16, dtype=torch.float64) / 16))
                for half, p in ((0, pos[..., 0]), (1, pos[..., 1])):
                    ang = p.double().cpu()[..., None] * inv                   # (B, N, 16)
                    c, s_ = ang.cos()[:, :, None, :], ang.sin()[:, :, None, :]
                    u, w = t[..., 32 * half: 32 * half + 16], t[..., 32 * half + 16: 32 * half + 32]
                    out = torch.cat([out[..., :32 * half], u * c - w * s_, w * c + u * s_, out[..., 32 * half + 32:]], -1)
                return out
            qq, kk = rot64(qq, qpos), rot64(kk, kpos)
        att = torch.softmax(torch.einsum("bnhd,bmhd->bhnm", qq, kk) * 0.125, -1)
        return torch.einsum("bhnm,bmhd->bnhd", att, vv)
    od = ref_attn(qd, kd, vd)
    (od * go.double().cpu()).sum().backward()
    want = [od.detach(), qd.grad, kd.grad, vd.grad]
    errs = {}
    for name, a3, a6, ah, w in zip(("out", "dq", "dk", "dv"), res["bf16x3"], res["bf16x6"], res["f16x3"], want):
        e3 = float((a3 - w).abs().max() / w.abs().max()); e6 = float((a6 - w).abs().max() / w.abs().max())
        eh = float((ah - w).abs().max() / w.abs().max())
        errs[name] = (e3, e6, eh)
        assert e6 <= 5e-6, (name, e6)
        assert e3 <= 5e-5, (name, e3)
        assert eh <= 5e-6, (name, eh)          # f16x3 (three products on fp16 pieces): the six-product bar
    print("attention error vs float64 (bf16x3, bf16x6, f16x3):", errs)
    assert not torch.equal(res["bf16x3"][0], res["bf16x6"][0]) and not torch.equal(res["f16x3"][0], res["bf16x6"][0])


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 3, 257, 257), (1, 2, 130, 771), (2, 12, 64, 64), (1, 1, 5, 1), (11, 16, 257, 257), (11, 16, 260, 129), (1, 1, 1025, 1025)])
def test_attention_forward_vs_oracle(B, H, Nq, Nk, attention_arith):
    from styl3r_amd.vit_ops import memory_efficient_attention
    g = torch.Generator(DEV).manual_seed(Nq * 7 + Nk)
    q = torch.randn(B, Nq, H, 64, device=DEV, generator=g)
    k = torch.randn(B, Nk, H, 64, device=DEV, generator=g)
    v = torch.randn(B, Nk, H, 64, device=DEV, generator=g)
    out = memory_efficient_attention(q, k, v, scale=0.125)
    ref, _ = vo.attention(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), 0.125)
    assert_close_rel(out.cpu().numpy(), ref, 1e-5, "attention fwd")


def test_attention_forward_fused_rope_on_qkv_views(attention_arith):
    from styl3r_amd.vit_ops import memory_efficient_attention
    B, N, H = 2, 257, 4
    g = torch.Generator(DEV).manual_seed(3)
    qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g)
    keep = qkv.clone()
    pos = torch.cartesian_prod(torch.arange(17), torch.arange(17))[:N].to(DEV)[None].expand(B, -1, -1).contiguous()
    out = memory_efficient_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], scale=0.125, qpos=pos, kpos=pos, max_pos=16)
    assert torch.equal(qkv, keep)                          # the qkv buffer is not rewritten
    qn, kn, vn = (keep[:, :, i].cpu().numpy() for i in range(3))
    p = pos.cpu().numpy()
    ref, _ = vo.attention(vo.rope2d(qn, p), vo.rope2d(kn, p), vn, 0.125)
    assert_close_rel(out.cpu().numpy(), ref, 1e-5, "attention fwd + fused rope")


@pytest.mark.parametrize("B,H,Nq,Nk,rope", [(2, 3, 257, 257, False), (1, 2, 130, 771, True), (1, 1, 5, 1, False), (2, 2, 33, 160, True),
                                             (11, 16, 257, 257, True), (11, 16, 259, 514, True), (1, 2, 256, 130, True)])   # 11 x 16 heads: the tail-row kernels kick in (they need > 512 workgroups)
def test_attention_backward_vs_oracle(B, H, Nq, Nk, rope, attention_arith):
    from styl3r_amd.vit_ops import memory_efficient_attention
    g = torch.Generator(DEV).manual_seed(Nq * 3 + Nk)
    q = torch.randn(B, Nq, H, 64, device=DEV, generator=g, requires_grad=True)
    k = torch.randn(B, Nk, H, 64, device=DEV, generator=g, requires_grad=True)
    v = torch.randn(B, Nk, H, 64, device=DEV, generator=g, requires_grad=True)
    w = torch.randn(B, Nq, H, 64, device=DEV, generator=g)
    kw = {}
    qn, kn = q.detach().cpu().numpy(), k.detach().cpu().numpy()
    if rope:
        side = 28
        grid = torch.cartesian_prod(torch.arange(side), torch.arange(side)).to(DEV)
        qpos = grid[:Nq][None].expand(B, -1, -1).contiguous(); kpos = grid[5:5 + Nk][None].expand(B, -1, -1).contiguous()
        kw = dict(qpos=qpos, kpos=kpos, max_pos=side)
    out = memory_efficient_attention(q, k, v, scale=0.125, **kw)
    (out * w).sum().backward()
    if rope:
        qr, kr = vo.rope2d(qn, qpos.cpu().numpy()), vo.rope2d(kn, kpos.cpu().numpy())
    else:
        qr, kr = qn, kn
    dq, dk, dv = vo.attention_backward(qr, kr, v.detach().cpu().numpy(), 0.125, w.cpu().numpy())
    if rope:   # gradient w.r.t. the unrotated inputs = inverse rotation of the gradient of the rotated ones
        dq, dk = vo.rope2d(dq, qpos.cpu().numpy(), fwd=-1.0), vo.rope2d(dk, kpos.cpu().numpy(), fwd=-1.0)
    # atol: with a single key the softmax is constant and dq = dk = 0 exactly; fp32 leaves ~1e-6
    assert_close_rel(q.grad.cpu().numpy(), dq, 2e-5, "dq", atol=1e-5)
    assert_close_rel(k.grad.cpu().numpy(), dk, 2e-5, "dk", atol=1e-5)
    assert_close_rel(v.grad.cpu().numpy(), dv, 2e-5, "dv", atol=1e-5)


def _load(mod, prefix):
    sd = {k[len(prefix) + 4:]: T(G[k]) for k in G.files if k.startswith(prefix + "_sd_")}
    missing = mod.load_state_dict(sd, strict=True)
    return mod.to(DEV).eval()


def _check_module(prefix, mod, call, in_names, tol=2e-5):
    mod = _load(mod, prefix)
    ins = {}
    for n in in_names:
        a = T(G[f"{prefix}_in_{n}"])
        ins[n] = a.requires_grad_(True) if a.is_floating_point() else a
    y = call(mod, ins)
    y = y[0] if isinstance(y, tuple) else y
    assert_close_rel(y.detach().cpu().numpy(), G[f"{prefix}_out"], tol, f"{prefix} output vs reference")
    (y * T(G[f"{prefix}_gout"])).sum().backward()
    for n in in_names:
        if ins[n].is_floating_point():
            assert_close_rel(ins[n].grad.cpu().numpy(), G[f"{prefix}_gin_{n}"], 5e-5, f"{prefix} d{n} vs reference")
    for k in G.files:
        if k.startswith(prefix + "_gp_"):
            name = k[len(prefix) + 4:]
            got = dict(mod.named_parameters())[name].grad
            assert_close_rel(got.cpu().numpy(), G[k], 5e-5, f"{prefix} d{name} vs reference")


def test_blocks_match_reference_modules():
    from styl3r_amd import vit
    rope = vit.RopeCfg(100.0, max_pos=16)
    _check_module("mlp", vit.Mlp(128, 256), lambda m, i: m(i["x"]), ["x"])
    _check_module("attn", vit.Attention(128, rope=rope, num_heads=2, qkv_bias=True), lambda m, i: m(i["x"], i["xpos"]), ["x", "xpos"])
    _check_module("block", vit.Block(128, 2, 1.0, qkv_bias=True, norm_layer=vit.LayerNorm6, rope=rope),
                  lambda m, i: m(i["x"], i["xpos"]), ["x", "xpos"])
    _check_module("xattn", vit.CrossAttention(128, rope=rope, num_heads=2, qkv_bias=True),
                  lambda m, i: m(i["q"], i["kv"], i["kv"], i["qpos"], i["kpos"]), ["q", "kv", "qpos", "kpos"])
    _check_module("dec", vit.DecoderBlock(128, 2, 1.0, qkv_bias=True, norm_layer=vit.LayerNorm6, rope=rope),
                  lambda m, i: m(i["x"], i["y"], i["xpos"], i["ypos"]), ["x", "y", "xpos", "ypos"])


@pytest.mark.parametrize("M,N,K,gelu,res", [(300, 200, 64, True, False), (514, 3072, 1024, False, False),
                                             (257, 768, 3072, False, True), (1, 9, 16, True, True), (5140, 1024, 1024, True, True)])
def test_fused_linear_vs_fp64(M, N, K, gelu, res):
    from styl3r_amd.vit_ops import fused_linear
    g = torch.Generator(DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g, requires_grad=True)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device=DEV, generator=g, requires_grad=True)
    r = torch.randn(M, N, device=DEV, generator=g, requires_grad=True) if res else None
    y = fused_linear(x, w, b, residual=r, gelu=gelu)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    rd = r.detach().double().requires_grad_(True) if res else None
    ref = torch.nn.functional.linear(xd, wd, bd)
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + rd
    assert_close_rel(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), 4e-6, "fused linear fwd")   # fp32 round-off at K up to 4096 (split-K: atomic order varies)
    gy = torch.randn(M, N, device=DEV, generator=g)
    (y * gy).sum().backward(); (ref * gy.double()).sum().backward()
    assert_close_rel(x.grad.cpu().numpy(), xd.grad.cpu().numpy(), 1e-5, "dx")
    assert_close_rel(w.grad.cpu().numpy(), wd.grad.cpu().numpy(), 1e-5, "dw")
    assert_close_rel(b.grad.cpu().numpy(), bd.grad.cpu().numpy(), 1e-5, "db")
    if res:
        assert_close_rel(r.grad.cpu().numpy(), rd.grad.cpu().numpy(), 1e-6, "dres")


# ---------------------------------------------------------------------------
# bf16x6 split arithmetic (csrc/vit_gemm_x6.hip): fp32 accuracy on the bf16 matrix cores
# ---------------------------------------------------------------------------
def _unpack_split(packed, rows, kcols):
    """packed [rows][k/8][piece][8] bf16 -> three fp64 (rows, k) planes"""
    raw = packed[: rows * kcols * 6].view(torch.int16).view(rows, kcols // 8, 3, 8).to(torch.int32)     # (8 KiB of trailing bytes: the f16x3 mode's |max| word)
    f = (raw << 16).view(torch.float32)
    return [f[:, :, p, :].reshape(rows, kcols).double() for p in range(3)]


@pytest.mark.parametrize("transposed", [False, True])
def test_split_weight_is_exact_to_2pow26(transposed):
    from styl3r_amd.vit_ops import split_weight
    g = torch.Generator(DEV).manual_seed(11)
    w = torch.randn(96, 80, device=DEV, generator=g) * torch.logspace(-6, 3, 80, device=DEV)[None]
    packed = split_weight(w, transposed)
    src = w.t().contiguous() if transposed else w
    p0, p1, p2 = _unpack_split(packed, *src.shape)
    err = (p0 + p1 + p2 - src.double()).abs()
    assert float((err / src.double().abs().clamp_min(1e-300)).max()) <= 2.0 ** -25
    assert float(((p0 - src.double()).abs() / src.double().abs()).max()) <= 2.0 ** -8     # piece 0 = RNE bf16
    # cache: same tensor -> same buffer; in-place update -> re-split
    assert split_weight(w, transposed).data_ptr() == packed.data_ptr()
    w.mul_(2.0)
    p0b = _unpack_split(split_weight(w, transposed), *src.shape)[0]
    assert torch.equal(p0b, 2.0 * p0)


@pytest.mark.parametrize("M,N,K,gelu", [(514, 1024, 1024, False), (300, 192, 4096, True), (1028, 3072, 1024, False), (77, 40, 64, False)])
def test_bf16x6_linear_matches_fp64_like_the_f32_mfma_path(M, N, K, gelu, monkeypatch):
    """error of the bf16x6 path vs an fp64 reference is of the order of the exact-f32 MFMA path's (both are fp32
    roundoff: <= 2e-6 of the output scale), forward and input gradient"""
    from styl3r_amd import vit_ops
    g = torch.Generator(DEV).manual_seed(M * 7 + N)
    x0 = torch.randn(M, K, device=DEV, generator=g)
    w0 = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b0 = torch.randn(N, device=DEV, generator=g)
    gy = torch.randn(M, N, device=DEV, generator=g)
    xd, wd = x0.double().requires_grad_(True), w0.double()
    ref = torch.nn.functional.linear(xd, wd, b0.double())
    ref = torch.nn.functional.gelu(ref) if gelu else ref
    (ref * gy.double()).sum().backward()
    errs = {}
    for mode in ("f32", "bf16x6"):
        monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
        x = x0.clone().requires_grad_(True); w = w0.clone().requires_grad_(True)
        y = vit_ops.fused_linear(x, w, b0, gelu=gelu)
        (y * gy).sum().backward()
        ef = float((y.detach().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        eb = float((x.grad.double() - xd.grad).abs().max() / xd.grad.abs().max())
        errs[mode] = (ef, eb)
    print("max-norm errors vs fp64 (fwd, dX):", errs)
    for k in (0, 1):
        assert errs["bf16x6"][k] <= 4e-6, errs
        assert errs["bf16x6"][k] <= 3.0 * errs["f32"][k] + 2e-7, errs


@pytest.mark.parametrize("M,N,K", [(5140, 1024, 1024), (1028, 3072, 768), (77, 40, 64), (16, 128, 128), (9, 8, 264), (2570, 200, 4096), (40000, 24, 2304)])
@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3"])
def test_linear_weight_gradient_kernel_ragged_shapes_vs_fp64(M, N, K, mode, monkeypatch):
    """vit_linear_x6_wgrad (buffer-load row walk, rows past M read as zero in hardware, M split across workgroups): dW and db against
    float64 on ragged M / N / K (not multiples of the 16-row slab or the 128 x 128 tile), with and without the bias gradient, and in
    accumulate mode on top of a running gradient"""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
    vit_ops._x6()                                                    # products per launch follow LINEAR_MODE
    lib = vit_ops.load()
    g = torch.Generator(DEV).manual_seed(M + 3 * N + K)
    x = torch.randn(M, K, device=DEV, generator=g); dy = torch.randn(M, N, device=DEV, generator=g)
    want_w = dy.double().t() @ x.double(); want_b = dy.double().sum(0)
    bar = 4e-6 if mode == "bf16x6" else 2e-5
    s = torch.cuda.current_stream().cuda_stream
    for with_bias in (True, False):
        buf = torch.full((N * K + N,), float("nan"), device=DEV)
        dw, db = buf[:N * K].view(N, K), buf[N * K:]
        rc = lib.vit_linear_x6_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if with_bias else None, M, N, K, s)
        assert rc == 0
        assert float((dw.double() - want_w).abs().max() / want_w.abs().max()) <= bar
        if with_bias:
            assert float((db.double() - want_b).abs().max() / want_b.abs().max()) <= 4e-6
    run_w = torch.randn(N, K, device=DEV, generator=g); run_b = torch.randn(N, device=DEV, generator=g)
    buf = torch.cat((run_w.flatten(), run_b))
    dw, db = buf[:N * K].view(N, K), buf[N * K:]
    assert lib.vit_linear_x6_wgrad_acc(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, s) == 0
    assert float((dw.double() - (want_w + run_w.double())).abs().max() / want_w.abs().max()) <= bar
    assert float((db.double() - (want_b + run_b.double())).abs().max() / want_b.abs().max()) <= 4e-6


def test_bf16x3_mode_is_two_orders_tighter_than_tf32_on_every_split_arithmetic_kernel(monkeypatch):
    """VIT_LINEAR_MODE=bf16x3 (three partial products per launch instead of six: include/vit_ops.h vit_x6_set_products): Linear
    forward / dX / dW / db and the 3x3 convolution forward / dX / dW against float64.  Bars: 2e-5 of the output scale (measured
    3.3e-6 .. 3.7e-6 per GEMM); TF32, which the reference enables for these products (croco.py:13), is ~3e-4.  The mode must not
    leak: afterwards the default mode gives the six-product result again."""
    from styl3r_amd import vit_ops
    g = torch.Generator(DEV).manual_seed(5)
    M, N, K = 1028, 768, 1024
    x0 = torch.randn(M, K, device=DEV, generator=g); w0 = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b0 = torch.randn(N, device=DEV, generator=g); gy = torch.randn(M, N, device=DEV, generator=g)
    xd, wd, bd = x0.double().requires_grad_(True), w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    (torch.nn.functional.linear(xd, wd, bd) * gy.double()).sum().backward()
    ref = torch.nn.functional.linear(xd, wd, bd).detach()
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    errs = {}
    for mode in ("bf16x3", "bf16x6"):
        monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
        x = x0.clone().requires_grad_(True); w = w0.clone().requires_grad_(True); b = b0.clone().requires_grad_(True)
        y = vit_ops.fused_linear(x, w, b)
        (y * gy).sum().backward()
        errs[mode] = (rel(y.detach(), ref), rel(x.grad, xd.grad), rel(w.grad, wd.grad), rel(b.grad, bd.grad))
        assert vit_ops.load().vit_x6_products() == (3 if mode == "bf16x3" else 6)
    print("Linear (fwd, dX, dW, db) max-norm error vs fp64:", errs)
    assert max(errs["bf16x3"]) <= 2e-5 and max(errs["bf16x6"]) <= 4e-6
    assert errs["bf16x3"][0] > errs["bf16x6"][0]                      # the mode really changed the arithmetic
    # convolution (the DPT heads' 3x3, large enough for the implicit-GEMM kernels)
    B, Ci, Co, H, W = 4, 128, 128, 64, 64
    cx0 = torch.randn(B, Ci, H, W, device=DEV, generator=g); conv = vit_ops.Conv2dX6(Ci, Co, 3, padding=1).to(DEV)
    cg = torch.randn(B, Co, H, W, device=DEV, generator=g)
    cxd = cx0.double().requires_grad_(True); cwd = conv.weight.detach().double().requires_grad_(True)
    cref = torch.nn.functional.conv2d(cxd, cwd, conv.bias.detach().double(), padding=1)
    (cref * cg.double()).sum().backward()
    cerrs = {}
    for mode in ("bf16x3", "bf16x6"):
        monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
        conv.zero_grad()
        cx = cx0.clone().requires_grad_(True)
        before = vit_ops.CALLS["conv_x6_fwd"]
        y = conv(cx)
        assert vit_ops.CALLS["conv_x6_fwd"] == before + 1
        (y * cg).sum().backward()
        cerrs[mode] = (rel(y.detach(), cref.detach()), rel(cx.grad, cxd.grad), rel(conv.weight.grad, cwd.grad))
    print("conv 3x3 (fwd, dX, dW) max-norm error vs fp64:", cerrs)
    assert max(cerrs["bf16x3"]) <= 2e-5 and max(cerrs["bf16x6"]) <= 4e-6


@pytest.mark.parametrize("cfg", [1, 2, 3])
@pytest.mark.parametrize("M,N,K,gelu,res", [(1028, 3072, 1024, False, False), (600, 200, 48, True, False), (2000, 1024, 2048, False, True), (77, 40, 16, False, True)])
def test_ring_dma_linear_kernels_equal_the_default_bf16x6_kernel(cfg, M, N, K, gelu, res):
    """csrc/vit_gemm_x6r.hip (LDS-DMA ring, block-layout weight, inline-asm fragment reads with counted waits, ping-pong wave
    pairs): same six partial products in the same order as vit_linear_x6_fwd, so the outputs agree to the last bit -- any race
    between a DMA, a conversion and a fragment read shows up as a wrong tile; ragged M / N (row clamp, zero-padded weight
    block), one- and many-slab K; repeated to catch an intermittent race.  (Shapes chosen so that the default path does not
    take its split-K branch, whose atomics reorder the sum.)"""
    from styl3r_amd import vit_ops
    g = torch.Generator(DEV).manual_seed(M + 3 * N + cfg)
    x = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    r = torch.randn(M, N, device=DEV, generator=g) if res else None
    vit_ops.LINEAR_MODE, keep = "bf16x6", vit_ops.LINEAR_MODE
    try:
        want = vit_ops.fused_linear(x, w, b, gelu=gelu, residual=r) if r is not None else vit_ops.fused_linear(x, w, b, gelu=gelu)
    finally:
        vit_ops.LINEAR_MODE = keep
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = torch.nn.functional.gelu(ref) if gelu else ref
    ref = ref + r.double() if r is not None else ref
    blk = vit_ops.split_weight_block(w)
    for rep in range(5):
        got = vit_ops.linear_x6r(x, blk, N, bias=b, residual=r, gelu=gelu, cfg=cfg)
        assert float((got.double() - ref).abs().max() / ref.abs().max()) <= 4e-6
        assert torch.equal(got, want), (rep, float((got - want).abs().max()))
    if cfg == 3 and K >= 48:      # the ping-pong kernel with a 3-way K split: partial tiles through a workspace, last arriver reduces + epilogue
        import ctypes as C
        lib = vit_ops.load()
        nb = lib.vit_linear_x6c_workspace_bytes(M, N, 3)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        got3 = torch.empty(M, N, device=DEV); pre3 = torch.empty(M, N, device=DEV)
        for rep in range(5):
            rc = lib.vit_linear_x6c_fwd(x.data_ptr(), blk.data_ptr(), b.data_ptr(), r.data_ptr() if r is not None else None, got3.data_ptr(),
                                        pre3.data_ptr(), M, N, K, 1 if gelu else 0, 3, ws.data_ptr(), nb, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
            assert float((got3.double() - ref).abs().max() / ref.abs().max()) <= 4e-6, rep
        lin = torch.nn.functional.linear(x.double(), w.double(), b.double())
        assert float((pre3.double() - lin).abs().max() / lin.abs().max()) <= 4e-6           # the pre-activation store of the reducer
        assert lib.vit_linear_x6c_choose_splits(5140, 1024, 4096) == 3 and lib.vit_linear_x6c_choose_splits(514, 1024, 1024) == 0
    # dX product through the transposed block packing: dX = dY . W
    gy = torch.randn(M, N, device=DEV, generator=g)
    if N % 16 == 0:
        dx = vit_ops.linear_x6r(gy, vit_ops.split_weight_block(w, transposed=True), K, cfg=cfg)
        dref = gy.double() @ w.double()
        assert float((dx.double() - dref).abs().max() / dref.abs().max()) <= 4e-6


def test_relu_dropout_kernel_is_relu_then_an_unbiased_dropout_and_its_backward_gates_by_the_output():
    """vit_relu_dropout_fwd / _bwd (the 'gs_params' heads' ReLU(True) -> Dropout(0.1), dpt_block.py:332-340): every output is 0 or
    max(x, 0) / (1 - p); the kept fraction of the positive inputs is 1 - p within 4 sigma and independent of the position; the same
    seed repeats the mask, another seed does not; backward: g / (1 - p) exactly where the output is positive; eval mode / p = 0 is
    plain ReLU."""
    from styl3r_amd import vit_ops
    torch.manual_seed(11)
    x0 = torch.randn(3, 32, 64, 64, device=DEV)
    p = 0.1
    conv = lambda t: t * 1.0                                   # (a non-leaf, like the convolution output the heads hand over)
    torch.manual_seed(5); vit_ops.reseed_dropout(); x = conv(x0.clone().requires_grad_(True)); y = vit_ops.relu_dropout(x, p, True)
    pos = x0 > 0
    kept = y > 0
    assert not bool((kept & ~pos).any())
    assert torch.equal(y[kept], (x0[kept] / (1 - p)).float()) or float((y[kept] - x0[kept] / (1 - p)).abs().max()) <= 1e-6 * float(x0.abs().max())
    n = int(pos.sum()); frac = float(kept.sum()) / n
    assert abs(frac - (1 - p)) <= 4 * (p * (1 - p) / n) ** 0.5, frac
    halves = [float((kept & pos)[..., :32].sum()) / float(pos[..., :32].sum()), float((kept & pos)[..., 32:].sum()) / float(pos[..., 32:].sum())]
    assert abs(halves[0] - halves[1]) <= 0.01
    state = torch.random.get_rng_state()
    torch.manual_seed(5); vit_ops.reseed_dropout(); y2 = vit_ops.relu_dropout(conv(x0.clone().requires_grad_(True)), p, True)
    assert torch.equal(y, y2)
    torch.manual_seed(5)
    assert torch.equal(torch.random.get_rng_state(), state) or True   # (the default CPU generator is never consumed: ADVICE r2)
    before = torch.random.get_rng_state(); vit_ops.relu_dropout(conv(x0.clone().requires_grad_(True)), p, True)
    assert torch.equal(before, torch.random.get_rng_state())
    assert torch.equal(y, y2)
    y3 = vit_ops.relu_dropout(conv(x0.clone().requires_grad_(True)), p, True)
    assert not torch.equal(y3 > 0, kept)
    # backward
    leaf = x0.clone().requires_grad_(True)
    torch.manual_seed(5); vit_ops.reseed_dropout(); out = vit_ops.relu_dropout(conv(leaf), p, True)
    g = torch.randn_like(out)
    out.backward(g)
    assert torch.equal(leaf.grad, torch.where(out.detach() > 0, g / (1 - p), torch.zeros_like(g)).float()) or \
        float((leaf.grad - torch.where(out.detach() > 0, g / (1 - p), torch.zeros_like(g))).abs().max()) <= 1e-6 * float(g.abs().max())
    # eval / p = 0
    assert torch.equal(vit_ops.relu_dropout(conv(x0.clone().requires_grad_(True)), p, False), torch.relu(x0))
    assert torch.equal(vit_ops.relu_dropout(conv(x0.clone().requires_grad_(True)), 0.0, True), torch.relu(x0))


def test_split_cache_never_serves_a_dead_tensors_entry():
    """a new weight that reuses a freed one's Python id / device address (version 0 again) must be split afresh"""
    from styl3r_amd.vit_ops import split_weight
    for i in range(6):
        w = torch.full((32, 64), float(i + 1), device=DEV)
        p0 = _unpack_split(split_weight(w), 32, 64)[0]
        assert float(p0[0, 0]) == float(i + 1)
        del w, p0


@pytest.mark.parametrize("B,Ci,Co,H,W,k,bias", [(2, 32, 64, 9, 13, 3, True), (1, 16, 130, 17, 5, 3, False), (3, 48, 128, 8, 8, 1, True),
                                                 (2, 256, 256, 32, 32, 3, True), (1, 96, 256, 16, 24, 3, False),
                                                 # W >= 32: the halo kernel (k_conv3h_x6): ragged patches (H % 4, W % 32 != 0), one patch row, Co not a multiple of 128
                                                 (2, 32, 64, 37, 45, 3, True), (1, 16, 130, 6, 70, 3, False), (3, 64, 96, 4, 32, 3, True), (1, 512, 128, 33, 33, 3, True)])
def test_conv_x6_matches_fp64_forward_and_backward(B, Ci, Co, H, W, k, bias, monkeypatch):
    """Conv2dX6 (bf16x6 implicit GEMM: forward + input gradient; library weight gradient) vs an fp64 convolution"""
    from styl3r_amd.vit_ops import Conv2dX6
    torch.manual_seed(B * 100 + Ci)
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "_CONV_X6_MIN_TILES", 0)          # force the bf16x6 kernels at these small test sizes
    monkeypatch.setattr(vit_ops, "_CONV_X6_MIN_ROWS", 16)
    m = Conv2dX6(Ci, Co, k, 1, k // 2, bias=bias).to(DEV)
    assert m._x6_ok(torch.empty(1, Ci, 4, 4, device=DEV))
    x = torch.randn(B, Ci, H, W, device=DEV, requires_grad=True)
    y = m(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    xd = x.detach().double().requires_grad_(True)
    wd = m.weight.detach().double().requires_grad_(True)
    bd = m.bias.detach().double().requires_grad_(True) if bias else None
    ref = torch.nn.functional.conv2d(xd, wd, bd, padding=k // 2)
    (ref * gy.double()).sum().backward()
    assert_close_rel(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), 5e-6, "conv fwd")
    assert_close_rel(x.grad.cpu().numpy(), xd.grad.cpu().numpy(), 5e-6, "conv dx")
    assert_close_rel(m.weight.grad.cpu().numpy(), wd.grad.cpu().numpy(), 2e-5, "conv dw")
    if bias:
        assert_close_rel(m.bias.grad.cpu().numpy(), bd.grad.cpu().numpy(), 2e-5, "conv db")


@pytest.mark.parametrize("B,Ci,Co,H,W,k", [(16, 32, 64, 64, 64, 3), (4, 48, 130, 128, 128, 1), (17, 16, 64, 64, 64, 3)])
def test_conv_x6_weight_gradient_kernel_matches_fp64(B, Ci, Co, H, W, k):
    """enough pixels (>= 65 536) for Conv2dX6's backward to take vit_conv_x6_wgrad: dW and db vs fp64"""
    from styl3r_amd.vit_ops import Conv2dX6
    torch.manual_seed(B + Ci)
    m = Conv2dX6(Ci, Co, k, 1, k // 2, bias=True).to(DEV)
    x = torch.randn(B, Ci, H, W, device=DEV)
    y = m(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    wd = m.weight.detach().double().requires_grad_(True); bd = m.bias.detach().double().requires_grad_(True)
    (torch.nn.functional.conv2d(x.double(), wd, bd, padding=k // 2) * gy.double()).sum().backward()
    assert_close_rel(m.weight.grad.cpu().numpy(), wd.grad.cpu().numpy(), 5e-6, "conv dw (x6)")
    assert_close_rel(m.bias.grad.cpu().numpy(), bd.grad.cpu().numpy(), 5e-6, "conv db (x6)")


@pytest.mark.parametrize("B,C,H,W,force", [(2, 32, 9, 13, True), (1, 144, 17, 8, True), (16, 32, 64, 64, True), (2, 32, 8, 8, False), (2, 48, 19, 41, True)])
def test_residual_conv_unit_fused_relu_and_skip_match_fp64(B, C, H, W, force, monkeypatch):
    """_ResidualConvUnit (dpt_block.py:79-118: conv2(relu(conv1(relu(x)))) + x) with both ReLUs and the skip add inside the
    convolution kernels (relu_in staging, residual epilogue; backward: sign-gate epilogue of the dX launch, relu_in in the
    weight-gradient kernel, skip gradient passed through) against the plain sequence in fp64.  Inputs keep a margin from
    zero so that no ReLU mask depends on fp32 round-off.  force=False: the small-problem fallback (library convolutions)."""
    from styl3r_amd import vit_ops
    from styl3r_amd.encoder import _ResidualConvUnit
    torch.manual_seed(B * 7 + C)
    if force:
        monkeypatch.setattr(vit_ops, "_CONV_X6_MIN_TILES", 0)
        monkeypatch.setattr(vit_ops, "_CONV_X6_MIN_ROWS", 16)
    m = _ResidualConvUnit(C).to(DEV)
    x = torch.randn(B, C, H, W, device=DEV)
    x = (x + 0.05 * torch.sign(x)).requires_grad_(True)
    before = dict(vit_ops.CALLS)
    y = m(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    took = {k: vit_ops.CALLS[k] - before[k] for k in before}
    assert took["conv_x6_fwd"] == (2 if force else 0) and took["conv_x6_dx"] == (2 if force else 0), took
    assert took["conv_x6_wgrad"] == (2 if force and B * H * W >= 65536 else 0), took
    md = _ResidualConvUnit(C).double().to(DEV)
    md.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    mid = torch.nn.functional.conv2d(torch.relu(xd), md.conv1.weight, md.conv1.bias, padding=1)
    ref = torch.nn.functional.conv2d(torch.relu(mid), md.conv2.weight, md.conv2.bias, padding=1) + xd
    (ref * gy.double()).sum().backward()
    # (the intermediate activation can sit within round-off of zero: compare where its ReLU mask is unambiguous)
    safe = bool((mid.detach().abs() > 1e-5).all())
    assert_close_rel(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "rcu fwd")
    tol = 2e-5 if safe else 2e-3
    assert_close_rel(x.grad.cpu().numpy(), xd.grad.cpu().numpy(), tol, "rcu dx")
    for name in ("conv1", "conv2"):
        assert_close_rel(getattr(m, name).weight.grad.cpu().numpy(), getattr(md, name).weight.grad.cpu().numpy(), max(tol, 3e-5), name + " dw")
        assert_close_rel(getattr(m, name).bias.grad.cpu().numpy(), getattr(md, name).bias.grad.cpu().numpy(), max(tol, 3e-5), name + " db")


@pytest.mark.parametrize("shape", [(2, 5, 8, 8), (1, 3, 16, 24), (3, 4, 1, 2), (2, 16, 64, 64)])
def test_upsample2x_matches_framework_bilinear_align_corners(shape):
    from styl3r_amd.vit_ops import upsample2x
    x = torch.randn(shape, device=DEV, requires_grad=True)
    y = upsample2x(x)
    ref = torch.nn.functional.interpolate(x.detach().double(), scale_factor=2, mode="bilinear", align_corners=True)
    assert y.shape == ref.shape
    assert_close_rel(y.detach().cpu().numpy(), ref.cpu().numpy(), 1e-5, "upsample2x vs fp64")   # fp32 source-index rounding
    same = torch.nn.functional.interpolate(x.detach(), scale_factor=2, mode="bilinear", align_corners=True)
    assert_close_rel(y.detach().cpu().numpy(), same.cpu().numpy(), 2e-7, "upsample2x vs the framework's fp32 kernel")
    g = torch.randn_like(y)
    y.backward(g)
    x2 = x.detach().clone().requires_grad_(True)
    torch.nn.functional.interpolate(x2, scale_factor=2, mode="bilinear", align_corners=True).backward(g)
    assert torch.allclose(x.grad, x2.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape,bias", [((4, 257, 1024), True), ((2, 33, 768), True), ((5, 256), True), ((1, 1025, 1024), False),
                                         ((3000, 512), True)])
def test_layernorm_kernels_match_fp64_with_and_without_skip(shape, bias):
    """vit_layernorm_fwd / vit_layernorm_bwd (nn.LayerNorm(C, eps=1e-6) semantics) vs fp64: output, dx, dgamma, dbeta;
    forward_skip: the residual-branch gradient is added inside the backward kernel (dx = dskip + LN'(dy))."""
    from styl3r_amd.vit_ops import LayerNorm
    torch.manual_seed(shape[0])
    C = shape[-1]
    m = LayerNorm(C, eps=1e-6, bias=bias).to(DEV)
    with torch.no_grad():
        m.weight.copy_(1 + 0.3 * torch.randn(C, device=DEV))
        if bias:
            m.bias.copy_(0.2 * torch.randn(C, device=DEV))
    x = (3 * torch.randn(shape, device=DEV) + 0.7).requires_grad_(True)
    gy, gs = torch.randn(shape, device=DEV), torch.randn(shape, device=DEV)
    xd = x.detach().double().requires_grad_(True)
    wd = m.weight.detach().double().requires_grad_(True)
    bd = m.bias.detach().double().requires_grad_(True) if bias else None
    for skip in (False, True):
        x.grad = m.weight.grad = None
        if bias:
            m.bias.grad = None
        xd.grad = wd.grad = None
        if bias:
            bd.grad = None
        ref = torch.nn.functional.layer_norm(xd, (C,), wd, bd, 1e-6)
        if skip:
            y, xs = m.forward_skip(x)
            assert xs.data_ptr() == x.data_ptr()
            ((y * gy).sum() + (xs * gs).sum()).backward()
            ((ref * gy.double()).sum() + (xd * gs.double()).sum()).backward()
        else:
            y = m(x)
            (y * gy).sum().backward()
            (ref * gy.double()).sum().backward()
        assert_close_rel(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), 2e-6, "ln fwd")
        assert_close_rel(x.grad.cpu().numpy(), xd.grad.cpu().numpy(), 5e-6, f"ln dx skip={skip}")
        assert_close_rel(m.weight.grad.cpu().numpy(), wd.grad.cpu().numpy(), 1e-5, "ln dgamma")
        if bias:
            assert_close_rel(m.bias.grad.cpu().numpy(), bd.grad.cpu().numpy(), 1e-5, "ln dbeta")
    # only the skip output used: the gradient passes through untouched
    x.grad = None
    y, xs = m.forward_skip(x)
    (xs * gs).sum().backward()
    assert torch.equal(x.grad, gs)
    # bitwise run-to-run determinism of the parameter gradients (fixed-order partial sums, no atomics)
    def grads():
        m.weight.grad = None
        (m(x.detach()) * gy).sum().backward()
        return m.weight.grad.clone()
    assert torch.equal(grads(), grads())


def test_no_grad_fast_paths_equal_the_autograd_paths(monkeypatch):
    """serving: under no_grad fused_linear and LayerNorm skip the autograd node and go straight to the kernels; with the same kernel behind
    both paths (round 6: at up to 1 024 rows the no-grad Linear takes csrc/vit_gemm_sm.hip, the autograd path keeps the 128-row kernel --
    vit_ops._ring_cfg) the results are the same bits as the autograd path's forward; with the small-M kernel they agree to fp32 round-off"""
    from styl3r_amd import vit_ops
    from styl3r_amd.vit_ops import LayerNorm, fused_linear
    torch.manual_seed(11)
    x = torch.randn(3, 257, 1024, device=DEV)
    w = torch.randn(768, 1024, device=DEV) / 32; b = torch.randn(768, device=DEV); r = torch.randn(3, 257, 768, device=DEV)
    ln = LayerNorm(1024, eps=1e-6).to(DEV)
    xg = x.clone().requires_grad_(True)
    want_lin, want_ln = fused_linear(xg, w, b, r, gelu=True), ln(xg)
    with torch.no_grad():
        got_lin, got_ln = fused_linear(x, w, b, r, gelu=True), ln(x)
        got_t = fused_linear(x.transpose(0, 1), w, b)              # non-contiguous input
    assert torch.equal(got_ln, want_ln.detach())
    assert float((got_lin - want_lin.detach()).abs().max()) <= 2e-6 * float(want_lin.abs().max())         # (two kernels: another summation order)
    monkeypatch.setattr(vit_ops, "SMALL_M_ROWS", 0)
    with torch.no_grad():
        assert torch.equal(fused_linear(x, w, b, r, gelu=True), want_lin.detach())                          # the same kernel behind both paths: the same bits
    monkeypatch.setattr(vit_ops, "SMALL_M_ROWS", 1024)
    # (this small GEMM splits K across workgroups with fp32 atomics: equal up to the summation order)
    assert torch.allclose(got_t, fused_linear(xg.transpose(0, 1), w, b).detach(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C,CO,p", [(128, 3, 0.0), (256, 8, 0.0), (256, 8, 0.1), (256, 3, 0.1)])
def test_head_tail_kernel_matches_fp64_and_the_separate_relu_dropout_kernel(C, CO, p):
    """vit_head_tail_fwd / _bwd = ReLU [-> Dropout(p)] -> Conv2d(C, CO, 1) of the DPT head tails (dpt_block.py:319-320, 337-339) in one
    pass each way: output, dh, dW, db against float64 of the same expression; with dropout the keep decisions must equal
    vit_relu_dropout_fwd's for the same seed (same generator, same element indexing), so that kernel provides the mask."""
    from styl3r_amd import vit_ops
    torch.manual_seed(7 + C + CO)
    B, H, W = 3, 40, 36
    conv = torch.nn.Conv2d(C, CO, 1).to(DEV)
    h = torch.randn(B, C, H, W, device=DEV, requires_grad=True)
    seed = 123456789
    before = vit_ops.CALLS["head_tail"]
    y = vit_ops._HeadTail.apply(h, conv.weight, conv.bias, p, seed)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    assert y.shape == (B, CO, H, W)
    if p > 0:
        a_mask = vit_ops._ReluDropout.apply(h.detach().clone() * 1.0, p, seed)          # keep(i) max(h, 0) / (1 - p) from the stand-alone kernel
        fac = (a_mask > 0).double() / (1 - p)
        frac = float((a_mask > 0).sum()) / float((h > 0).sum())
        assert abs(frac - (1 - p)) < 0.01
    else:
        fac = (h.detach() > 0).double()
    hd = h.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True); bd = conv.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(hd * fac, wd, bd)
    (ref * g.double()).sum().backward()
    assert_close_rel(y.detach().cpu().numpy(), ref.detach().cpu().numpy(), 2e-6, "head tail fwd")
    assert_close_rel(h.grad.cpu().numpy(), hd.grad.cpu().numpy(), 2e-6, "head tail dh")
    assert_close_rel(conv.weight.grad.cpu().numpy(), wd.grad.cpu().numpy(), 1e-5, "head tail dW")
    assert_close_rel(conv.bias.grad.cpu().numpy(), bd.grad.cpu().numpy(), 1e-5, "head tail db")
    # the module-level entry: qualifies here, refuses other channel counts (the caller then keeps the separate kernels)
    assert vit_ops.head_tail(h.detach(), conv, p, True) is not None and vit_ops.CALLS["head_tail"] >= before + 2
    assert vit_ops.head_tail(h.detach(), torch.nn.Conv2d(C, 11, 1).to(DEV), p, True) is None


def test_input_merger_path_matches_fp64():
    """`feat_up(path_1) + ReLU(Conv2d(3, 256, 7, 1, 3)(imgs))` (dpt_gs_head.py:113-118,146-148) on vit_im2col7 + the bf16x6 1x1 convolution
    kernels + the fused up-sample / ReLU / add pass: output and the gradients of path_1, the 7x7 weight and its bias vs float64"""
    from styl3r_amd import vit_ops
    torch.manual_seed(3)
    B, H, W = 4, 128, 160                                # B H W >= 65 536 pixels: the weight gradient takes vit_conv_x6_wgrad
    conv7 = vit_ops.Conv2dX6(3, 256, 7, 1, 3).to(DEV)
    imgs = torch.rand(B, 3, H, W, device=DEV) * 2 - 1
    p1 = torch.randn(B, 256, H // 2, W // 2, device=DEV, requires_grad=True)
    before = dict(vit_ops.CALLS)
    out = vit_ops.input_merger_upsample_add(p1, imgs, conv7)
    assert out is not None and vit_ops.CALLS["input_merger_x6"] == before["input_merger_x6"] + 1
    g = torch.randn_like(out)
    (out * g).sum().backward()
    assert vit_ops.CALLS["conv_x6_wgrad"] == before["conv_x6_wgrad"] + 1
    pd = p1.detach().double().requires_grad_(True)
    wd = conv7.weight.detach().double().requires_grad_(True); bd = conv7.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.interpolate(pd, scale_factor=2, mode="bilinear", align_corners=True) + \
        torch.relu(torch.nn.functional.conv2d(imgs.double(), wd, bd, padding=3))
    (ref * g.double()).sum().backward()
    assert_close_rel(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "input merger fwd")
    assert_close_rel(p1.grad.cpu().numpy(), pd.grad.cpu().numpy(), 1e-5, "input merger d path_1")
    assert_close_rel(conv7.weight.grad.cpu().numpy(), wd.grad.cpu().numpy(), 1e-5, "input merger dW")
    assert_close_rel(conv7.bias.grad.cpu().numpy(), bd.grad.cpu().numpy(), 1e-5, "input merger db")
    # an image that needs a gradient (parity fixtures only) keeps the framework path
    assert vit_ops.input_merger_upsample_add(p1.detach(), imgs.clone().requires_grad_(True), conv7) is None


def test_dpt_reassemble_as_linear_layers_equals_the_convolution_stack():
    """DPTAdapter._reassemble (1x1 conv / ConvTranspose k == stride / 3x3 stride-2 conv as Linear layers on the token grid) vs the
    nn.Sequential it replaces (dpt_block.py:350-419), outputs and every gradient, on the device"""
    from styl3r_amd.encoder import DPTAdapter
    torch.manual_seed(0)
    m = DPTAdapter(3, [1024, 768, 768, 768], [0, 6, 9, 12], "pts3d").to(DEV)
    B, nh, nw = 4, 16, 16
    for i, c in enumerate((1024, 768, 768, 768)):
        tok = torch.randn(B, nh * nw, c, device=DEV, requires_grad=True)
        new = m._reassemble(i, tok, nh, nw)
        g = torch.randn_like(new)
        params = list(m.act_postprocess[i].parameters())
        ga = torch.autograd.grad(new, [tok] + params, g)
        td = tok.detach().double().requires_grad_(True)
        seq = m.act_postprocess[i]
        import copy
        seq64 = copy.deepcopy(seq).double()
        old = seq64(td.transpose(1, 2).reshape(B, c, nh, nw))
        gb = torch.autograd.grad(old, [td] + list(seq64.parameters()), g.double())
        assert_close_rel(new.detach().cpu().numpy(), old.detach().cpu().numpy(), 3e-6, f"reassemble {i}")
        for a, b_, nm in zip(ga, gb, ["tok"] + [n for n, _ in seq.named_parameters()]):
            assert_close_rel(a.cpu().numpy(), b_.cpu().numpy(), 1e-5, f"reassemble {i} d{nm}")


@pytest.mark.parametrize("arith", ["bf16x6", "f32"])
@pytest.mark.parametrize("B,N,H", [(3, 257, 12), (2, 514, 16), (2, 256, 4)])
def test_packed_qkv_attention_equals_the_three_tensor_path_and_writes_one_gradient(B, N, H, arith, monkeypatch):
    """attention_qkv (self-attention on the packed (B,N,3,H,64) projection; the backward writes dq / dk / dv into the planes of ONE
    gradient through VitAttnArgs.dq_sn / dkv_sn) == memory_efficient_attention on the three strided views (whose gradient autograd
    assembles with select_backward fills, copies and adds): same kernels, so outputs and gradients must agree to the bit, and both
    match float64 (blocks.py:97-134).  N = 257 runs the tail kernels, too."""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "ATTENTION_ARITH", arith)
    torch.manual_seed(B + N)
    qkv0 = torch.randn(B, N, 3, H, 64, device=DEV)
    pos = torch.stack(torch.meshgrid(torch.arange(B), torch.arange(N), indexing="ij"), -1)[..., 1:].repeat(1, 1, 2).to(DEV) % 17
    pos = pos.to(torch.int64).contiguous()
    g = torch.randn(B, N, H, 64, device=DEV)
    a = qkv0.clone().requires_grad_(True)
    oa = vit_ops.attention_qkv(a, 0.125, pos, 100.0, 16)
    (oa * g).sum().backward()
    b = qkv0.clone().requires_grad_(True)
    ob = vit_ops.memory_efficient_attention(b[:, :, 0], b[:, :, 1], b[:, :, 2], scale=0.125, qpos=pos, kpos=pos, rope_base=100.0, max_pos=16)
    (ob * g).sum().backward()
    assert torch.equal(oa, ob)
    assert torch.equal(a.grad, b.grad)
    # float64 of the same expression (RoPE through the module's own rotation of a copy)
    qd = qkv0.double()
    rope = vit_ops.RoPE2D(100.0, max_pos=16)
    qr = rope(qkv0[:, :, 0].transpose(1, 2).contiguous().clone(), pos).transpose(1, 2).double()
    kr = rope(qkv0[:, :, 1].transpose(1, 2).contiguous().clone(), pos).transpose(1, 2).double()
    att = torch.softmax(torch.einsum("bnhd,bmhd->bhnm", qr, kr) * 0.125, -1)
    ref = torch.einsum("bhnm,bmhd->bnhd", att, qd[:, :, 2])
    assert_close_rel(oa.detach().cpu().numpy(), ref.cpu().numpy(), 5e-6, "packed attention fwd vs fp64")


def test_gelu_backward_inside_the_fc2_input_gradient_gemm():
    """vit.Mlp with GeluLink (GELU' applied in the epilogue of fc2's dX GEMM, vit_linear_x6_fwd act = 2; no GeluBackward pass) vs the
    same two Linear layers without the link and vs float64 (blocks.py:76-82)."""
    from styl3r_amd import vit, vit_ops
    torch.manual_seed(5)
    m = vit.Mlp(768, 3072).to(DEV)
    x0 = torch.randn(1300, 768, device=DEV)
    res = torch.randn(1300, 768, device=DEV)
    g = torch.randn(1300, 768, device=DEV)

    def run(linked):
        for p in m.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        if linked:
            y = m(x, residual=res)
        else:
            y = vit_ops.fused_linear(vit_ops.fused_linear(x, m.fc1.weight, m.fc1.bias, gelu=True), m.fc2.weight, m.fc2.bias, residual=res)
        (y * g).sum().backward()
        return y.detach(), x.grad, [p.grad.clone() for p in m.parameters()]
    ya, xa, pa = run(True)
    yb, xb, pb = run(False)
    assert_close_rel(ya.cpu().numpy(), yb.cpu().numpy(), 1e-6, "forward linked vs unlinked (split-K atomics reorder fp32 sums)")
    xd = x0.double().requires_grad_(True)
    md = [p.detach().double().requires_grad_(True) for p in m.parameters()]
    yd = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xd, md[0], md[1])), md[2], md[3]) + res.double()
    (yd * g.double()).sum().backward()
    assert_close_rel(xa.cpu().numpy(), xd.grad.cpu().numpy(), 5e-6, "dx (GELU' in the epilogue) vs fp64")
    assert_close_rel(xa.cpu().numpy(), xb.cpu().numpy(), 2e-6, "dx linked vs unlinked")
    for a, b_, d, n in zip(pa, pb, md, ("fc1.w", "fc1.b", "fc2.w", "fc2.b")):
        assert_close_rel(a.cpu().numpy(), d.grad.cpu().numpy(), 1e-5, f"d{n} vs fp64")
        assert_close_rel(a.cpu().numpy(), b_.cpu().numpy(), 5e-6, f"d{n} linked vs unlinked")


def test_patch_embed_as_linear_equals_the_strided_convolution():
    """PatchEmbedDust3R (croco/patch_embed.py:19-29: Conv2d(3, D, 16, 16)) as a Linear over patch rows vs the convolution in float64"""
    from styl3r_amd.encoder import PatchEmbedDust3R
    torch.manual_seed(1)
    pe = PatchEmbedDust3R((512, 512), 16, 3, 1024).to(DEV)
    x = (torch.rand(3, 3, 64, 96, device=DEV) * 2 - 1).requires_grad_(True)
    tok, pos = pe(x)
    g = torch.randn_like(tok)
    (tok * g).sum().backward()
    xd = x.detach().double().requires_grad_(True)
    wd = pe.proj.weight.detach().double().requires_grad_(True); bd = pe.proj.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, wd, bd, stride=16).flatten(2).transpose(1, 2)
    (ref * g.double()).sum().backward()
    assert tok.shape == (3, 24, 1024) and pos.shape == (3, 24, 2) and pos[0, 7].tolist() == [1, 1]
    assert_close_rel(tok.detach().cpu().numpy(), ref.detach().cpu().numpy(), 3e-6, "patch embed fwd")
    assert_close_rel(x.grad.cpu().numpy(), xd.grad.cpu().numpy(), 5e-6, "patch embed dx")
    assert_close_rel(pe.proj.weight.grad.cpu().numpy(), wd.grad.cpu().numpy(), 1e-5, "patch embed dW")
    assert_close_rel(pe.proj.bias.grad.cpu().numpy(), bd.grad.cpu().numpy(), 1e-5, "patch embed db")


@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3", "f16x3"])
@pytest.mark.parametrize("M", [4100, 2562])             # ragged (not a multiple of any tile height): the large-M and the mid-M table
@pytest.mark.parametrize("N,K,gelu", [(3072, 1024, False), (4096, 1024, True), (1024, 4096, False), (2304, 768, False), (768, 3072, False)])
def test_ring_kernel_dispatch_is_bit_identical_forward_and_input_gradient(N, K, gelu, M, mode, monkeypatch):
    """vit_ops._RING_SHAPES sends the large-M Linear shapes (forward and the input-gradient GEMM, incl. the GELU' epilogue of an fc2 behind
    a GELU) to the LDS-DMA ring kernels (csrc/vit_gemm_x6r.hip): same split arithmetic, same accumulation order per output element, so
    outputs and dX must equal the default kernel's bit for bit, in six- and in three-product mode."""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
    monkeypatch.setattr(vit_ops, "NARROW_N", 0)        # (the 768-wide layers take the barrier-free kernel of vit_gemm_sm.hip by default: another summation order; tests/test_gpu_small_linear.py)
    torch.manual_seed(N + K)
    x0 = torch.randn(M, K, device=DEV); w = (torch.randn(N, K, device=DEV) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device=DEV, requires_grad=True); g = torch.randn(M, N, device=DEV)
    pre_for_gelu_grad = torch.randn(M, K, device=DEV)

    def run(ring):
        monkeypatch.setattr(vit_ops, "RING_DISPATCH", ring)
        before = vit_ops.CALLS["linear_x6r"]
        x = x0.clone().requires_grad_(True)
        link_in = None
        if not gelu and (K, N) in ((4096, 1024), (3072, 768)):      # this layer plays fc2: its dX runs GELU'(pre) in the epilogue
            link_in = vit_ops.GeluLink(); link_in.pre = pre_for_gelu_grad
        y = vit_ops.fused_linear(x, w, b, gelu=gelu, link_in=link_in)
        (gx,) = torch.autograd.grad(y, x, g)
        return y.detach(), gx, vit_ops.CALLS["linear_x6r"] - before
    ya, xa, na = run(True)
    yb, xb, nb = run(False)
    assert nb == 0
    table = (vit_ops._RING_SHAPES if M >= 4096 else vit_ops._RING_SHAPES_MID).get(mode, {})
    want = (1 if table.get((N, K)) else 0) + (1 if (mode in ("bf16x3", "f16x3") and table.get((K, N))) else 0)
    assert na == want and (want >= 1 or mode == "bf16x6" or M < 4096), (na, want)
    assert torch.equal(ya, yb) and torch.equal(xa, xb)


# --------------------------------------------------------------------------- f16x3: two fp16 pieces + tensor scales, three products
def _f16x3_table(monkeypatch, x0, w0, b0, gy, gelu=False):
    """(fwd, dX, dW, db) max-norm errors vs float64 of the fused Linear in every split-arithmetic mode"""
    from styl3r_amd import vit_ops
    xd, wd, bd = x0.double().requires_grad_(True), w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    ref = torch.nn.functional.linear(xd, wd, bd)
    ref = torch.nn.functional.gelu(ref) if gelu else ref
    (ref * gy.double()).sum().backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    errs = {}
    for mode in ("bf16x3", "f16x3", "bf16x6"):
        monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
        x = x0.clone().requires_grad_(True); w = w0.clone().requires_grad_(True); b = b0.clone().requires_grad_(True)
        y = vit_ops.fused_linear(x, w, b, gelu=gelu)
        (y * gy).sum().backward()
        errs[mode] = (rel(y.detach(), ref.detach()), rel(x.grad, xd.grad), rel(w.grad, wd.grad), rel(b.grad, bd.grad))
        assert vit_ops.load().vit_x6_products() == {"bf16x3": 3, "f16x3": 2, "bf16x6": 6}[mode]
    return errs


@pytest.mark.parametrize("M,N,K,gelu", [(514, 1024, 1024, False), (300, 192, 4096, True), (1028, 3072, 1024, False), (77, 40, 64, False), (5140, 768, 3072, False)])
def test_f16x3_linear_has_fp32_class_accuracy_at_the_three_product_price(M, N, K, gelu, monkeypatch):
    """the fp16 two-term split (value * 2^k = h + l, three products on the f16 MFMA) against float64, forward, dX, dW,
    db, beside bf16x3 (same MFMA count) and bf16x6 (twice as many): <= 2 x the six-product kernels' error (+ fp32 rounding of the
    output; measured: equal or below it on every shape -- at long contractions both sit on the fp32 accumulation floor) and well
    below bf16x3's 3e-6 .. 5e-6."""
    g = torch.Generator(DEV).manual_seed(M * 7 + N)
    x0 = torch.randn(M, K, device=DEV, generator=g)
    w0 = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b0 = torch.randn(N, device=DEV, generator=g)
    gy = torch.randn(M, N, device=DEV, generator=g)
    errs = _f16x3_table(monkeypatch, x0, w0, b0, gy, gelu)
    print(f"  Linear {M}x{N}x{K} (fwd, dX, dW, db) vs fp64:", {k: tuple(f"{e:.1e}" for e in v) for k, v in errs.items()})
    for i in range(3):
        assert errs["f16x3"][i] <= 2.0 * errs["bf16x6"][i] + 3e-7, (i, errs)
        assert errs["f16x3"][i] * 2.5 <= errs["bf16x3"][i] or errs["f16x3"][i] <= 5e-7, (i, errs)      # (the fp32 accumulation floor is ~1e-6 at K >= 3072)
    assert errs["f16x3"][3] <= 4e-6


@pytest.mark.parametrize("xs,ws,gs", [(1.0, 1.0, 1e-8), (1e-3, 30.0, 1e-2), (400.0, 1e-4, 3e-6), (1.0, 1.0, 1.0)])
def test_f16x3_linear_range_handling_gradient_magnitude_operands(xs, ws, gs, monkeypatch):
    """fp16 has 5 exponent bits; the mode lives on per-tensor power-of-two scales from vit_amax.  Operands at the magnitudes a train step
    really sees -- gradients of 1e-8, activations of 400 with one 30 x outlier row and a block of 1e-6-sized rows, weights of 1e-4 --
    must keep the accuracy of the unit-scale case (the error is measured against float64, relative to the output's max-norm), and a tensor
    with a wide spread (outliers 2^15 above the bulk) must not lose the bulk."""
    g = torch.Generator(DEV).manual_seed(17)
    M, N, K = 1028, 768, 1024
    x0 = torch.randn(M, K, device=DEV, generator=g) * xs
    x0[5] *= 30.0                                             # an outlier row sets the tensor's scale ...
    x0[100:140] *= 2.0 ** -15                                 # ... and a block of rows sits 2^20 below it
    w0 = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5 * ws
    b0 = torch.randn(N, device=DEV, generator=g) * xs * ws
    gy = torch.randn(M, N, device=DEV, generator=g) * gs
    gy[:, 7] *= 100.0
    errs = _f16x3_table(monkeypatch, x0, w0, b0, gy)
    print(f"  scales x {xs:g} w {ws:g} dY {gs:g} (fwd, dX, dW, db):", {k: tuple(f"{e:.1e}" for e in v) for k, v in errs.items()})
    for i in range(3):
        assert errs["f16x3"][i] <= 2.0 * errs["bf16x6"][i] + 3e-7, (i, errs)
    # the rows 2^20 below the tensor's maximum, judged on THEIR OWN scale: still far better than bf16x3 would be at full scale
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "f16x3")
    y = vit_ops.fused_linear(x0, w0, b0 * 0)
    ref = x0[100:140].double() @ w0.double().t()
    small = float((y[100:140].double() - ref).abs().max() / ref.abs().max())
    print(f"  rows 2^-20 below the tensor maximum, error on their own scale: {small:.1e}")
    assert small <= 2e-4


def test_f16x3_refuses_a_launch_without_operand_maxima(monkeypatch):
    """no guessed scale, no fallback: in f16x3 mode an x6 launch that was not given the |max| words of its activations is an error"""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "f16x3")
    vit_ops._x6()
    lib = vit_ops.load()
    x = torch.randn(64, 64, device=DEV); w = torch.randn(64, 64, device=DEV); out = torch.empty(64, 64, device=DEV)
    wp = vit_ops.split_weight(w)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.vit_linear_x6_fwd(x.data_ptr(), wp.data_ptr(), None, None, out.data_ptr(), None, 64, 64, 64, 0, s) == -1
    assert lib.vit_linear_x6_wgrad(x.data_ptr(), x.data_ptr(), out.data_ptr(), None, 64, 64, 64, s) == -1
    word = vit_ops._amax_word(x)
    torch.cuda.synchronize()
    assert word.numel() == 64 * 32 and int(word.max().item()) == int(x.abs().max().view(torch.int32).item())      # (a 64-word line) the exact bit pattern of the maximum
    vit_ops._announce(word)
    assert lib.vit_linear_x6_fwd(x.data_ptr(), wp.data_ptr(), None, None, out.data_ptr(), None, 64, 64, 64, 0, s) == 0
    assert lib.vit_linear_x6_fwd(x.data_ptr(), wp.data_ptr(), None, None, out.data_ptr(), None, 64, 64, 64, 0, s) == -1   # consumed
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "bf16x6")
    vit_ops._x6()


def test_f16x3_convolutions_forward_input_and_weight_gradient(monkeypatch):
    """3x3 (ReLU-fused residual unit included) and 1x1 convolutions of the DPT heads in f16x3: forward, dX, dW against float64,
    beside the other two modes; small-pixel layers go through vit_im2col3_rows + the Linear weight-gradient kernel"""
    from styl3r_amd import vit_ops
    g = torch.Generator(DEV).manual_seed(23)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    for (B, Ci, Co, H, W, k) in [(4, 128, 128, 64, 64, 3), (2, 256, 256, 32, 32, 3), (2, 160, 256, 64, 64, 1), (20, 128, 96, 128, 128, 3)]:
        cx0 = torch.randn(B, Ci, H, W, device=DEV, generator=g) * 0.3
        conv = vit_ops.Conv2dX6(Ci, Co, k, padding=k // 2).to(DEV)
        cg = torch.randn(B, Co, H, W, device=DEV, generator=g) * 1e-5
        cxd = cx0.double().requires_grad_(True); cwd = conv.weight.detach().double().requires_grad_(True)
        cref = torch.nn.functional.conv2d(torch.relu(cxd), cwd, conv.bias.detach().double(), padding=k // 2) + cxd[:, :1].expand(-1, Co, -1, -1) * 0
        (cref * cg.double()).sum().backward()
        cerrs = {}
        for mode in ("bf16x3", "f16x3", "bf16x6"):
            monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
            conv.zero_grad()
            cx = cx0.clone().requires_grad_(True)
            before = dict(vit_ops.CALLS)
            y = vit_ops._ConvX6.apply(cx, conv.weight, conv.bias, None, True)
            (y * cg).sum().backward()
            assert vit_ops.CALLS["conv_x6_fwd"] == before["conv_x6_fwd"] + 1 and vit_ops.CALLS["conv_x6_dx"] == before["conv_x6_dx"] + 1
            cerrs[mode] = (rel(y.detach(), cref.detach()), rel(cx.grad, cxd.grad), rel(conv.weight.grad, cwd.grad))
        print(f"  conv {k}x{k} B{B} {Ci}->{Co} {H}x{W} (fwd, dX, dW):", {m: tuple(f"{e:.1e}" for e in v) for m, v in cerrs.items()})
        for i in range(3):
            assert cerrs["f16x3"][i] <= 2.0 * cerrs["bf16x6"][i] + 3e-7, (i, cerrs)
            assert cerrs["f16x3"][i] * 2.5 <= cerrs["bf16x3"][i] or cerrs["f16x3"][i] <= 5e-7, (i, cerrs)
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "bf16x6")
    vit_ops._x6()


def test_f16x3_operand_maxima_published_by_the_producing_kernels(monkeypatch):
    """f16x3 needs the |max| of every activation operand.  LayerNorm (forward: y, backward: dx), the GEMM epilogues (GeluLink) and the 3x3 halo
    convolution publish the |max| of what they store (vit_x6_set_output_amax), and the consumer finds the word through vit_ops._PUBLISHED instead
    of running a vit_amax pass.  A transformer block and a residual convolution unit, forward + backward: (i) words really are reused,
    (ii) a published word IS the exact |max| (LayerNorm: bit-identical results with publishing switched off), (iii) results stay at the
    f16x3 accuracy against float64."""
    from styl3r_amd import vit_ops
    from styl3r_amd.encoder import _ResidualConvUnit
    from styl3r_amd.vit import Block, LayerNorm6, RopeCfg
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "f16x3")
    monkeypatch.setattr(vit_ops, "ATTENTION_ARITH", "bf16x6")
    torch.manual_seed(4)
    blk = Block(1024, 16, 4.0, qkv_bias=True, norm_layer=LayerNorm6, rope=RopeCfg(100.0)).to(DEV)
    x0 = torch.randn(2, 257, 1024, device=DEV)
    pos = torch.stack(torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij"), -1).reshape(1, 256, 2)
    pos = torch.cat([torch.tensor([[[16, 0]]]), pos], 1).expand(2, -1, -1).contiguous().to(DEV)
    gy = torch.randn(2, 257, 1024, device=DEV) * 1e-4

    def run_block(publish):
        monkeypatch.setattr(vit_ops, "PUBLISH_AMAX", publish)
        blk.zero_grad()
        before = dict(vit_ops.CALLS)
        x = x0.clone().requires_grad_(True)
        y = blk(x, pos)
        (y * gy).sum().backward()
        return y.detach(), x.grad, blk.mlp.fc1.weight.grad.clone(), {k: vit_ops.CALLS[k] - before[k] for k in ("amax_pass", "amax_published")}
    ya, xa, wa, ca = run_block(True)
    yb, xb, wb, cb = run_block(False)
    print(f"  block: with publishing {ca}, without {cb}")
    # published: LN1 -> qkv, LN2 -> fc1 (forward), LN2's dx -> proj's dY (backward); fc1 <-> fc2 go through the GeluLink words, which are not counted;
    # left to a pass: the attention output (-> proj), the attention backward's dqkv (-> qkv) and the incoming gradient of this stand-alone block
    assert cb["amax_published"] == 0 and ca["amax_published"] >= 3 and ca["amax_pass"] <= cb["amax_pass"] - 3, (ca, cb)
    # a published word is the exact |max| of the tensor: same scales either way (what is left is the split-K / split-M kernels' fp32 atomics
    # reordering their partial sums from run to run at this small M)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(ya, yb) <= 2e-6 and rel(xa, xb) <= 2e-6 and rel(wa, wb) <= 2e-6

    rcu = _ResidualConvUnit(128).to(DEV)
    cx0 = torch.randn(2, 128, 64, 64, device=DEV) * 0.5
    cg = torch.randn(2, 128, 64, 64, device=DEV) * 1e-3
    xd = cx0.double().requires_grad_(True)
    w1, b1, w2, b2 = (t.detach().double() for t in (rcu.conv1.weight, rcu.conv1.bias, rcu.conv2.weight, rcu.conv2.bias))
    F = torch.nn.functional
    ref = F.conv2d(torch.relu(F.conv2d(torch.relu(xd), w1, b1, padding=1)), w2, b2, padding=1) + xd
    (ref * cg.double()).sum().backward()
    monkeypatch.setattr(vit_ops, "PUBLISH_AMAX", True)
    before = dict(vit_ops.CALLS)
    cx = cx0.clone().requires_grad_(True)
    out = rcu(cx)
    (out * cg).sum().backward()
    took = {k: vit_ops.CALLS[k] - before[k] for k in ("amax_pass", "amax_published")}
    print(f"  residual conv unit: {took}; fwd {rel(out.detach().double(), ref.detach()):.1e}, dX {rel(cx.grad.double(), xd.grad):.1e}")
    assert took["amax_published"] >= 2, took                      # conv1's output -> conv2's input; conv2's dX -> conv1's dY
    assert rel(out.detach().double(), ref.detach()) <= 2e-6 and rel(cx.grad.double(), xd.grad) <= 2e-6
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "bf16x6")
    vit_ops._x6()


def test_halo_convolution_at_the_largest_head_shapes_is_linear_and_mode_consistent(monkeypatch):
    """k_conv3h_x6 at the sizes the oracle-style fp64 comparison above cannot reach in seconds -- the C5 stress shapes' gs head (256 -> 256 at
    512 x 512) and the C3 heads at 20 images -- through size-independent properties: (i) the three arithmetic modes (bf16x6 / f16x3: fp32 round-off;
    bf16x3: 2^-16 class) agree with each other at their own accuracy, (ii) linearity in the input, conv(a x1 + x2) - bias == a (conv(x1) - bias) +
    (conv(x2) - bias), (iii) a one-hot input reproduces the flipped weight taps exactly where the patch borders, the halo rows and the image
    borders meet."""
    from styl3r_amd import vit_ops
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    g = torch.Generator(DEV).manual_seed(31)
    for (B, Ci, Co, H, W) in [(1, 256, 256, 512, 512), (20, 256, 128, 128, 128)]:
        conv = vit_ops.Conv2dX6(Ci, Co, 3, padding=1).to(DEV)
        with torch.no_grad():                                         # weights from the seeded generator too (the global RNG made the bars order-dependent)
            bound = 1.0 / (Ci * 9) ** 0.5
            conv.weight.copy_((torch.rand(conv.weight.shape, device=DEV, generator=g) * 2 - 1) * bound)
            conv.bias.copy_((torch.rand(conv.bias.shape, device=DEV, generator=g) * 2 - 1) * bound)
        x1 = torch.randn(B, Ci, H, W, device=DEV, generator=g); x2 = torch.randn(B, Ci, H, W, device=DEV, generator=g)
        out = {}
        with torch.no_grad():
            for mode in ("bf16x6", "f16x3", "bf16x3"):
                monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
                out[mode] = conv(x1)
            # bars from tools/probes/halo_bar_calibration.py (6 seeds x both shapes, gpurun_out/r05a_halo_bars.jsonl -> profiles/r05_halo_bars.jsonl):
            # worst f16x3 3.3e-6, bf16x3 5.3e-6, linearity 2.2e-6; the bars keep >= 2.4x margin over the worst observation
            assert rel(out["f16x3"], out["bf16x6"]) <= 8e-6 and rel(out["bf16x3"], out["bf16x6"]) <= 3e-5, (rel(out["f16x3"], out["bf16x6"]), rel(out["bf16x3"], out["bf16x6"]))
            monkeypatch.setattr(vit_ops, "LINEAR_MODE", "f16x3")
            bias = conv.bias.view(1, -1, 1, 1)
            lhs = conv(0.5 * x1 + x2) - bias
            rhs = 0.5 * (out["f16x3"] - bias) + (conv(x2) - bias)
            assert rel(lhs, rhs) <= 6e-6, rel(lhs, rhs)
            del x2, lhs, rhs
            # one-hot probes at patch corners (4 x 32 patches), the image corner and an interior pixel
            for (y, x) in [(0, 0), (3, 31), (4, 32), (H - 1, W - 1), (H // 2 + 1, W // 2 - 3)]:
                x1.zero_(); x1[0, 5, y, x] = 1.0
                o = conv(x1) - bias
                for ky in range(3):
                    for kx in range(3):
                        yy, xx = y + 1 - ky, x + 1 - kx            # the output pixel that sees the probe through tap (ky, kx)
                        if 0 <= yy < H and 0 <= xx < W:
                            want = conv.weight[:, 5, ky, kx]
                            assert float((o[0, :, yy, xx] - want).abs().max()) <= 2e-6 * float(conv.weight.abs().max()) + 1e-9, (y, x, ky, kx)
                assert float(o[0].abs().sum((0,))[max(0, y - 1):y + 2, max(0, x - 1):x + 2].sum()) > 0
                o[0, :, max(0, y - 1):y + 2, max(0, x - 1):x + 2] = 0
                assert float(o.abs().max()) == 0.0                  # nothing leaks outside the 3 x 3 footprint
        del x1, out
        torch.cuda.empty_cache()
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "bf16x6")
    vit_ops._x6()


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "bf16x3"])
def test_batched_weight_resplit_equals_the_single_weight_images(mode, monkeypatch):
    """vit_ops.refresh_split_cache / vit_split_weights_many (one launch for every cached image of every updated weight, what TrainStep calls
    after the optimizer step) writes the bytes the single-weight calls write: forward and transposed, MFMA-order and block layout, ragged
    row counts, in every arithmetic mode; the cache entries carry the new version afterwards and untouched weights are left alone."""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
    vit_ops._x6()
    g = torch.Generator(DEV).manual_seed(7)
    shapes = [(3072, 1024), (200, 72), (64, 136), (768, 768), (1000, 64)]
    ws = [torch.nn.Parameter(torch.randn(n, k, device=DEV, generator=g) * (0.02 + 0.3 * i)) for i, (n, k) in enumerate(shapes)]
    images = lambda w: [vit_ops.split_weight(w, False), vit_ops.split_weight(w, True), vit_ops.split_weight_block(w, False), vit_ops.split_weight_block(w, True)]
    try:
        first = [images(w) for w in ws]                              # populate the cache
        ptrs = [[t.data_ptr() for t in im] for im in first]
        with torch.no_grad():
            for i, w in enumerate(ws[:-1]):                          # an "optimizer step" on all but the last weight
                w.mul_(1.0 + 0.01 * (i + 1)).add_(0.001)
        if mode == "f16x3":
            for w in ws[:-1]:
                vit_ops._weight_amax_word(w, w.detach())             # (AdamWHIP publishes these from its own kernel)
        before = vit_ops.CALLS["split_many_images"]
        n = vit_ops.refresh_split_cache(ws)
        assert n == 4 * (len(ws) - 1) and vit_ops.CALLS["split_many_images"] - before == n
        assert vit_ops.refresh_split_cache(ws) == 0                  # everything is current now
        torch.cuda.synchronize()
        for w, (N, K), pp in zip(ws, shapes, ptrs):
            got = images(w)                                          # cache hits: no launch, the refreshed buffers
            assert [t.data_ptr() for t in got] == pp
            ref = images(torch.nn.Parameter(w.detach().clone()))     # the single-weight kernels on the same values
            for li, (a, b) in enumerate(zip(got, ref)):
                transposed, block = li & 1, li >= 2
                R, Kc = (K, N) if transposed else (N, K)
                Rp = (R + 63) // 64 * 64 if block else R
                body = Rp * Kc * 6
                pa, pb = a[:body].view(-1, 16), b[:body].view(-1, 16)
                if block:                                            # [row block][k group][piece][64 rows][16 B]
                    pa, pb = pa.view(-1, 3, 64, 16), pb.view(-1, 3, 64, 16)
                else:                                                # [row][k group][piece][16 B]
                    pa, pb = pa.view(-1, 3, 16), pb.view(-1, 3, 16)
                pieces = 2 if mode == "f16x3" else 3                 # (f16x3 never writes or reads the third slot)
                assert torch.equal(pa[:, :pieces], pb[:, :pieces]), (mode, (N, K), li)
                if mode == "f16x3":                                  # the |max| line the readers take their inverse scale from
                    ta, tb = a[body:body + 8192].view(torch.int32)[::32], b[body:body + 8192].view(torch.int32)[::32]
                    assert torch.equal(ta, tb), (mode, (N, K), li, "tail")
    finally:
        vit_ops.LINEAR_MODE = "bf16x6"
        vit_ops._x6()


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "bf16x3"])
def test_pair_split_equals_the_single_image_kernels_and_feeds_the_backward(mode, monkeypatch):
    """vit_split_weight_pair (round 6: the forward AND the transposed image of a Linear weight from one read of the weight, what a
    trainable layer's forward launches when the optimizer has stepped) writes the bytes of the single-image kernels in all four layout
    combinations, ragged row counts included; a fused Linear whose images came from it gives the outputs and gradients of the
    one-launch-per-image path bit for bit, and its backward launches no split kernel of its own."""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
    vit_ops._x6()
    g = torch.Generator(DEV).manual_seed(11)
    try:
        for (N, K) in [(3072, 1024), (200, 72), (64, 136), (768, 1024), (1000, 64)]:
            for block_f in (False, True):
                for block_t in (False, True):
                    w = torch.nn.Parameter(torch.randn(N, K, device=DEV, generator=g) * 0.3)
                    vit_ops.split_weight_pair(w, block_f, block_t)
                    got = [(vit_ops.split_weight_block if block_f else vit_ops.split_weight)(w, False),
                           (vit_ops.split_weight_block if block_t else vit_ops.split_weight)(w, True)]          # cache hits
                    w2 = torch.nn.Parameter(w.detach().clone())
                    monkeypatch.setattr(vit_ops, "PAIR_SPLIT", False)
                    ref = [(vit_ops.split_weight_block if block_f else vit_ops.split_weight)(w2, False),
                           (vit_ops.split_weight_block if block_t else vit_ops.split_weight)(w2, True)]
                    monkeypatch.setattr(vit_ops, "PAIR_SPLIT", True)
                    torch.cuda.synchronize()
                    for li, (a, b, block) in enumerate(zip(got, ref, (block_f, block_t))):
                        R, Kc = (K, N) if li else (N, K)
                        body = ((R + 63) // 64 * 64 if block else R) * Kc * 6
                        pa, pb = a[:body].view(-1, 16), b[:body].view(-1, 16)
                        pa, pb = (pa.view(-1, 3, 64, 16), pb.view(-1, 3, 64, 16)) if block else (pa.view(-1, 3, 16), pb.view(-1, 3, 16))
                        pieces = 2 if mode == "f16x3" else 3
                        assert torch.equal(pa[:, :pieces], pb[:, :pieces]), (mode, (N, K), block_f, block_t, li)
                        if mode == "f16x3":
                            assert torch.equal(a[body:body + 8192].view(torch.int32)[::32], b[body:body + 8192].view(torch.int32)[::32]), "tail"
        # through the layer: M = 5 140 rows takes the ring kernels (block images) where the shape is in the table
        M, N, K = 5140, 3072, 1024
        x0 = torch.randn(M, K, device=DEV, generator=g)
        w0 = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
        outs = []
        for pair in (True, False):
            monkeypatch.setattr(vit_ops, "PAIR_SPLIT", pair)
            x = x0.clone().requires_grad_(True); w = torch.nn.Parameter(w0.clone())
            n0 = vit_ops.CALLS["split_pair"]
            y = vit_ops.fused_linear(x, w)
            assert (vit_ops.CALLS["split_pair"] - n0) == (1 if pair else 0)
            keys = len(vit_ops._SPLIT_CACHE)
            y.square().sum().backward()
            if pair:
                assert len(vit_ops._SPLIT_CACHE) == keys            # the backward found its image: no new entry, no split launch
            outs.append((y.detach(), x.grad, w.grad))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # same images -> same bits
        assert float((outs[0][2] - outs[1][2]).abs().max()) <= 1e-5 * float(outs[1][2].abs().max())  # (dW: split-M atomics arrive in any order)
    finally:
        vit_ops.LINEAR_MODE = "bf16x6"
        vit_ops._x6()


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
def test_conv_weight_pair_split_equals_the_rearranged_copy_path(mode, monkeypatch):
    """vit_split_conv_weight_pair (round 6) builds the forward and the input-gradient image of a 1x1 / 3x3 convolution weight straight from the
    (Co, Ci, k, k) parameter; the round-2..5 path made a permuted copy (and a flipped, channel-transposed one) and split those.  Same bytes,
    channel counts that are not multiples of the kernel's 16 x 16 tile included; and a ReLU-fused convolution run on either path gives the same
    outputs and input gradients."""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
    vit_ops._x6()
    g = torch.Generator(DEV).manual_seed(23)
    try:
        for (Co, Ci, k) in [(256, 256, 3), (128, 96, 3), (24, 40, 3), (256, 256, 1), (32, 136, 1), (8, 8, 3)]:
            w = torch.nn.Parameter(torch.randn(Co, Ci, k, k, device=DEV, generator=g) * 0.1)
            monkeypatch.setattr(vit_ops, "PAIR_SPLIT", True)
            n0 = vit_ops.CALLS["split_pair"]
            a_f = vit_ops.split_conv_weight(w, False, want_dx=True)
            a_d = vit_ops.split_conv_weight(w, True)
            assert vit_ops.CALLS["split_pair"] - n0 == 1                   # one launch made both
            w2 = torch.nn.Parameter(w.detach().clone())
            monkeypatch.setattr(vit_ops, "PAIR_SPLIT", False)
            b_f, b_d = vit_ops.split_conv_weight(w2, False), vit_ops.split_conv_weight(w2, True)
            torch.cuda.synchronize()
            pieces = 2 if mode == "f16x3" else 3
            for a, b, (R, Kc) in ((a_f, b_f, (Co, k * k * Ci)), (a_d, b_d, (Ci, k * k * Co))):
                body = R * Kc * 6
                assert torch.equal(a[:body].view(-1, 3, 16)[:, :pieces], b[:body].view(-1, 3, 16)[:, :pieces]), (mode, Co, Ci, k)
                if mode == "f16x3":     # (the |max| pass ran over differently ordered copies: the 64 slots differ, their maximum -- the scale -- does not)
                    assert int(a[body:body + 8192].view(torch.int32)[::32].max()) == int(b[body:body + 8192].view(torch.int32)[::32].max())
        x0 = torch.randn(2, 256, 32, 32, device=DEV, generator=g)
        w0 = torch.randn(256, 256, 3, 3, device=DEV, generator=g) * 0.02
        res = []
        for pair in (True, False):
            monkeypatch.setattr(vit_ops, "PAIR_SPLIT", pair)
            x = x0.clone().requires_grad_(True); w = torch.nn.Parameter(w0.clone())
            y = vit_ops._ConvX6.apply(x, w, None, None, True)
            y.square().sum().backward()
            res.append((y.detach(), x.grad))
        # (2 x 32 x 32 pixels = 32 output tiles: the kernel splits K across workgroups and adds with atomics -- equal up to their arrival order)
        for a, b in zip(res[0], res[1]):
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    finally:
        vit_ops.LINEAR_MODE = "bf16x6"
        vit_ops._x6()


def test_attention_kernels_publish_the_absolute_maximum_of_what_they_store(monkeypatch):
    """f16x3 (round 6): the attention forward fills a |max| word with the largest |out| it stores and the backward one for the packed qkv
    gradient (or one each for dq / dk / dv of a cross-attention), tail rows (the 257th token) included, so the proj / qkv / projq / projk /
    projv layers around it need no vit_amax pass.  The words must equal the maxima of the tensors exactly (a max is order-independent) and the
    consumers must find them (vit_ops.CALLS: no amax pass for these tensors)."""
    from styl3r_amd import vit_ops
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", "f16x3")
    monkeypatch.setattr(vit_ops, "ATTENTION_ARITH", "f16x3")
    vit_ops._x6()
    g = torch.Generator(DEV).manual_seed(3)
    word_max = lambda w: w.view(torch.int32)[::32].max().item()
    bits_max = lambda t: t.detach().abs().max().view(torch.int32).item()
    try:
        B, N, H = 2, 257, 4
        qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).requires_grad_(True)
        pos = torch.stack(torch.meshgrid(torch.arange(17, device=DEV), torch.arange(16, device=DEV), indexing="ij"), -1).reshape(-1, 2)[:N][None].expand(B, -1, -1).contiguous()
        o = vit_ops.attention_qkv(qkv, 0.125, pos)
        w = vit_ops._known_amax(o)
        assert w is not None and word_max(w) == bits_max(o)
        w2 = vit_ops._known_amax(o.reshape(B, N, H * 64))            # the view the proj layer sees
        assert w2 is not None and w2.data_ptr() == w.data_ptr()
        go = torch.randn(o.shape, device=DEV, generator=g)
        (dqkv,) = torch.autograd.grad(o, qkv, go)
        wg = vit_ops._known_amax(dqkv)
        assert wg is not None and word_max(wg) == bits_max(dqkv)
        # cross-attention: three tensors, three words (Nk = 514: two tail-free blocks + rows in the vector kernels)
        q = torch.randn(B, 257, H, 64, device=DEV, generator=g).requires_grad_(True)
        k = torch.randn(B, 514, H, 64, device=DEV, generator=g).requires_grad_(True)
        v = torch.randn(B, 514, H, 64, device=DEV, generator=g).requires_grad_(True)
        o2 = vit_ops.memory_efficient_attention(q, k, v, scale=0.125)
        assert word_max(vit_ops._known_amax(o2)) == bits_max(o2)
        dq, dk, dv = torch.autograd.grad(o2, (q, k, v), torch.randn(o2.shape, device=DEV, generator=g))
        for t in (dq, dk, dv):
            wt = vit_ops._known_amax(t)
            assert wt is not None and word_max(wt) == bits_max(t)
        # the exact-f32 kernels (mode f32) fill the words by a pass over the result
        monkeypatch.setattr(vit_ops, "ATTENTION_ARITH", "f32")
        o3 = vit_ops.attention_qkv(qkv.detach(), 0.125, pos)
        assert word_max(vit_ops._known_amax(o3)) == bits_max(o3)
        # a block's proj consumes the published word: no amax pass
        monkeypatch.setattr(vit_ops, "ATTENTION_ARITH", "f16x3")
        wt_ = torch.nn.Parameter(torch.randn(256, 256, device=DEV, generator=g) * 0.05)
        o4 = vit_ops.attention_qkv(qkv.detach(), 0.125, pos)
        n0 = vit_ops.CALLS["amax_pass"]
        with torch.no_grad():
            vit_ops.fused_linear(o4.reshape(B, N, H * 64), wt_)
        assert vit_ops.CALLS["amax_pass"] == n0
    finally:
        vit_ops.LINEAR_MODE = "bf16x6"; vit_ops.ATTENTION_ARITH = "bf16x6"
        vit_ops._x6()
