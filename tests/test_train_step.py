"""Train-step host logic: parameter selection / optimizer groups / LR schedule of `configure_optimizers`
(src/model/model_wrapper_style.py:843-916), and (GPU) the two-pass style-stage step of `training_step` (:118-232)."""
import pytest
import torch

from styl3r_amd.train import make_lr_scheduler, make_optimizer, select_trainable


def _meta_encoder(stylized):
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    with torch.device("meta"):
        return EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=stylized))


def test_style_stage_selection_matches_reference_rules():
    m = _meta_encoder(True)
    new, pre, frozen = select_trainable(m)
    names = {id(p): n for n, p in m.named_parameters()}
    assert all("stylizer.dec" in names[id(p)] or "gaussian_appearance_head" in names[id(p)] for p in new)
    assert all(any(k in names[id(p)] for k in ("stylizer.enc", "stylizer.mask_token", "stylizer.patch_embed")) for p in pre)
    assert any(n.startswith("backbone.") for n in frozen) and any(n.startswith("downstream_head1") for n in frozen)
    assert all(not p.requires_grad for n, p in m.named_parameters() if n in set(frozen))
    # the style stage trains the token stylizer + appearance head only: 1.75 GB of fp32 gradients
    grad_bytes = 4 * sum(p.numel() for p in new + pre)
    assert 1.70e9 < grad_bytes < 1.80e9, grad_bytes
    assert len(new) + len(pre) + len(frozen) == sum(1 for _ in m.parameters())


def test_nvs_stage_selection_trains_everything():
    m = _meta_encoder(False)
    new, pre, frozen = select_trainable(m)
    assert not frozen and 4 * sum(p.numel() for p in new + pre) == 4 * 1_049_635_033
    names = {id(p): n for n, p in m.named_parameters()}
    assert any("intrinsic_encoder" in names[id(p)] for p in new) and any("gaussian_param_head" in names[id(p)] for p in new)
    assert all("backbone.enc_blocks" not in names[id(p)] for p in new)


def test_optimizer_groups_and_schedule():
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
    opt = make_optimizer([a], [b], lr=2e-4, backbone_lr_multiplier=0.1)
    assert [g["lr"] for g in opt.param_groups] == [2e-4, 2e-5]
    assert all(g["weight_decay"] == 0.05 and tuple(g["betas"]) == (0.9, 0.95) for g in opt.param_groups)
    sched = make_lr_scheduler(opt, warm_up_steps=10, max_steps=100, lr=2e-4)
    lrs = []
    for _ in range(60):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sched.step()
    assert abs(lrs[0] - 2e-5) < 1e-12 and abs(lrs[10] - 2e-4) < 1e-12       # linear warm-up from lr/warm_up_steps
    assert all(x < y for x, y in zip(lrs[:10], lrs[1:11])) and all(x > y for x, y in zip(lrs[11:59], lrs[12:60]))
    # as in the reference, eta_min = 0.1 * lr is shared by both groups: the backbone group stays at 2e-5
    assert abs(opt.param_groups[1]["lr"] - 2e-5) < 1e-12


@pytest.mark.gpu
def test_style_stage_step_two_passes_on_gpu():
    """C4 shape of the step: stylized encoder, VGG style loss + identity pass, frozen backbone (random-init VGG:
    the torchvision weights are not available here)."""
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    from styl3r_amd.losses import IdentityLoss, LossStyle, VGGEncoder
    from styl3r_amd.scenes import make_scene
    from styl3r_amd.train import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
                pos_embed="RoPE100", img_size=(512, 512))
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=True), trunk_params=tiny).to(dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    vgg = VGGEncoder().to(dev)
    step = TrainStep(enc, dec, losses=[LossStyle(vgg=vgg)], identity_loss=IdentityLoss(vgg=vgg), warm_up_steps=5, max_steps=50)
    assert step.frozen_names and not enc.backbone.enc_blocks[0].attn.qkv.weight.requires_grad
    b, v, vt, H = 1, 2, 2, 64
    sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=vt, image_hw=(H, H), seed=3)
    ex = lambda t: t.to(dev)[None].expand(b, *t.shape).contiguous()
    batch = dict(context=dict(image=torch.rand(b, v, 3, H, H, device=dev) * 2 - 1, intrinsics=ex(sc.intrinsics[:1].expand(v, 3, 3))),
                 target=dict(image=torch.rand(b, vt, 3, H, H, device=dev), extrinsics=ex(sc.extrinsics), intrinsics=ex(sc.intrinsics),
                             near=ex(sc.near), far=ex(sc.far)),
                 style=dict(image=torch.rand(b, 3, H, H, device=dev)))
    frozen_before = enc.backbone.enc_blocks[0].attn.qkv.weight.detach().clone()
    train_before = enc.token_stylizer.dec_blocks[0].mlp.fc1.weight.detach().clone()
    l0 = step(batch); l1 = step(batch)
    assert torch.isfinite(l0) and torch.isfinite(l1)
    assert torch.equal(enc.backbone.enc_blocks[0].attn.qkv.weight, frozen_before)
    assert not torch.equal(enc.token_stylizer.dec_blocks[0].mlp.fc1.weight, train_before)
    assert enc.backbone.enc_blocks[0].attn.qkv.weight.grad is None
    assert step.global_step == 2


@pytest.mark.gpu
def test_nvs_training_reduces_the_loss_end_to_end():
    """encoder -> batched rasterizer -> MSE -> backward -> clip -> AdamW on a fixed batch: the loss goes down (gradients
    of every stage are consistent enough to optimise through)"""
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    from styl3r_amd.scenes import make_scene
    from styl3r_amd.train import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
                pos_embed="RoPE100", img_size=(512, 512))
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False), trunk_params=tiny).to(dev).eval()   # eval: no head dropout
    with torch.no_grad():      # a random-init point head puts every Gaussian behind / beside the cameras: start them in view
        for h in (enc.downstream_head1, enc.downstream_head2):
            h.dpt.head[4].bias.copy_(torch.tensor([0.0, 0.0, 1.2], device=dev))
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    step = TrainStep(enc, dec, lr=5e-4, clip=0.5)
    b, v, vt, H = 2, 2, 2, 64
    sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=vt, image_hw=(H, H), seed=3)
    ex = lambda t: t.to(dev)[None].expand(b, *t.shape).contiguous()
    g = torch.Generator(dev).manual_seed(1)
    batch = dict(context=dict(image=torch.rand(b, v, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=ex(sc.intrinsics[:1].expand(v, 3, 3))),
                 target=dict(image=torch.rand(b, vt, 3, 8, 8, device=dev, generator=g).repeat_interleave(8, -1).repeat_interleave(8, -2) * 0.5 + 0.25,
                             extrinsics=ex(sc.extrinsics), intrinsics=ex(sc.intrinsics), near=ex(sc.near), far=ex(sc.far)))
    losses = [float(step(batch)) for _ in range(40)]
    assert all(torch.isfinite(torch.tensor(losses)))
    first, last = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    assert last < 0.9 * first, (first, last, losses[::5])


@pytest.mark.gpu
def test_inplace_bucket_gradients_match_the_copy_path():
    """BucketedGradReducer(inplace_grads=True): the bf16x6 Linear accumulates dW / db straight into the bucket slices (incl.
    weights used by TWO encoder passes in one backward); every parameter gradient must equal the copy path's"""
    from styl3r_amd.ddp import BucketedGradReducer
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
                pos_embed="RoPE100", img_size=(512, 512))
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=True), trunk_params=tiny).to(dev).eval()
    g = torch.Generator(dev).manual_seed(2)
    H = 64
    ctx = dict(image=torch.rand(1, 2, 3, H, H, device=dev, generator=g) * 2 - 1,
               intrinsics=torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]], device=dev).expand(1, 2, 3, 3).contiguous())
    s1 = dict(image=torch.rand(1, 3, H, H, device=dev, generator=g) * 2 - 1)
    s2 = dict(image=ctx["image"][:, 0])

    def run(inplace):
        params = [p for p in enc.parameters() if p.requires_grad]
        red = BucketedGradReducer(params, None, bucket_bytes=8 << 20, inplace_grads=inplace)
        red.prepare()
        tot = 0
        for st in (s1, s2):                                  # two passes through the same weights (style + identity)
            gs = enc(ctx, st, 0)
            tot = tot + (gs.means * 0.01).sum() + gs.harmonics.sum() * 0.1 + gs.opacities.sum() * 0.1 + gs.covariances.sum() * 1e3
        tot.backward()
        red.finish()
        grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in enc.named_parameters()}
        adopted = sum(1 for b in red.buckets for p, v in zip(b["params"], b["views"]) if p.grad is not None and p.grad.data_ptr() == v.data_ptr())
        red.close()
        return grads, adopted

    ref, _ = run(False)
    ref2, _ = run(False)                                      # run-to-run noise of the copy path itself (atomics, library convs)
    got, adopted = run(True)
    assert adopted > 20                                       # the Linear weights really live in the buckets
    for n in ref:
        if ref[n] is None:
            assert got[n] is None or float(got[n].abs().max()) == 0.0, n
            continue
        scale = float(ref[n].abs().max())
        noise = float((ref2[n] - ref[n]).abs().max())
        # DPT convolution weights: the library's weight-gradient kernels are not run-to-run reproducible at these tiny sizes
        # (1e-4 .. 3e-3 relative between two identical copy-path runs: tools/probes/inplace_probe.py); they are not touched
        # by the in-place path, so they only get a sanity bound.  Everything else (the in-place Linear layers included): tight.
        tol = 1e-2 * scale if ".dpt." in n else 5e-4 * scale + 4 * noise + 1e-7
        assert float((got[n] - ref[n]).abs().max()) <= tol + 1e-12, (n, float((got[n] - ref[n]).abs().max()), noise, scale)


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_every_collective_and_matches_the_local_path():
    """the RCCL branch of the train step had never executed on a GPU (every GPU run was one rank without a
    process group).  A ONE-rank "nccl" (= RCCL) group with `force_collective=True` issues the module-state broadcast, every 64 MiB
    bucket all-reduce on RCCL's stream while the backward keeps writing dW / db in place into later buckets, and the used-map
    exchange; the sum over one rank is the identity, so after two full-size-trunk C3 steps the clipped gradients must equal the
    no-collective path's to within that path's own run-to-run noise (the weight-gradient kernels' fp32 atomics).  A missing stream
    dependency between the bucket writers, the pack copy and RCCL's stream would show up here as garbage in a bucket."""
    import os
    import torch.distributed as dist
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    from styl3r_amd import rasterizer
    from styl3r_amd.scenes import make_scene, recentre_output_heads_
    from styl3r_amd.train import TrainStep
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        b, v, H = 2, 2, 256
        sc = make_scene(n_ctx=v, grid_hw=(8, 8), n_views=4, image_hw=(H, H), seed=11)
        g = torch.Generator(dev).manual_seed(3)
        ex = lambda t, *shape: t.to(dev)[None].expand(b, *shape).contiguous()
        batch = dict(context=dict(image=torch.rand(b, v, 3, H, H, device=dev, generator=g) * 2 - 1,
                                  intrinsics=sc.intrinsics[:1].to(dev).expand(b, v, 3, 3).contiguous()),
                     target=dict(image=torch.rand(b, 4, 3, H, H, device=dev, generator=g), extrinsics=ex(sc.extrinsics, -1, -1, -1),
                                 intrinsics=ex(sc.intrinsics, -1, -1, -1), near=ex(sc.near, -1), far=ex(sc.far, -1)))
        dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)

        def run(group, force, dp_mode="all_reduce"):
            torch.manual_seed(0); torch.cuda.manual_seed(0)
            with torch.device(dev):
                enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False)).eval()   # eval: no dropout RNG in the comparison
            # a random-init point head puts (nearly) every Gaussian outside every frustum: the render is the background and EVERY gradient is
            # exactly zero (r03 ran this test on zeros).  Re-centre the five output convolutions so that the step renders a real scene
            recentre_output_heads_(enc, batch["context"], dict(image=batch["context"]["image"][:, 0]))
            step = TrainStep(enc, dec, dist=group, force_collective=force, dp_mode=dp_mode, warm_up_steps=2000)   # config/main.yaml:37: LinearLR from lr / 2000
            assert step.reducer.collective == force and step.reducer.mode == dp_mode
            losses = [float(step(batch)) for _ in range(2)]
            assert rasterizer.LAST_STATS["pairs"] > rasterizer.LAST_STATS["gaussians_per_scene"], rasterizer.LAST_STATS
            step.reducer.wait_params()
            flats = [bk["flat"][:bk["n"]].detach().clone() for bk in step.reducer.buckets]
            info = dict(synced=step.synced_bytes, buckets=len(flats), unused=len(step.reducer._unused), losses=losses,
                        probe=enc.backbone.enc_blocks[3].mlp.fc1.weight.detach()[:8].clone(), flat_params=all("pflat" in bk for bk in step.reducer.buckets))
            step.reducer.close()
            del step, enc
            torch.cuda.empty_cache()
            return flats, info
        a, ia = run(None, False)
        a2, _ = run(None, False)
        c, ic = run(dist, True)
        assert ia["synced"] == 0 and ic["synced"] >= 4 * 1_049_635_033          # the broadcast really went through RCCL
        assert ic["buckets"] == ia["buckets"] >= 60 and ic["unused"] == ia["unused"] >= 1     # mask_token: unused on "every" rank
        assert all(abs(x - y) <= 1e-5 * abs(x) for x, y in zip(ia["losses"], ic["losses"])), (ia, ic)
        worst = 0.0
        assert min(float(x.abs().max()) for x in a) > 0.0, "a bucket of all-zero gradients: the step rendered nothing"
        for i, (x, x2, y) in enumerate(zip(a, a2, c)):
            assert torch.isfinite(y).all(), f"bucket {i}"
            scale = float(x.abs().max())
            noise = float((x - x2).abs().max())
            err = float((x - y).abs().max())
            worst = max(worst, err / max(scale, 1e-30))
            # (with a real scene the buckets hold real gradients: both deviations are the weight-gradient kernels' fp32 atomics reordering their
            # split-M partial sums -- one noise sample per bucket is itself only good to a factor of a few)
            assert err <= 10 * noise + 5e-4 * scale, (i, err, noise, scale)
        print(f"  one-rank RCCL vs local: {len(a)} buckets, worst bucket deviation {worst:.2e} of the bucket scale")
        # SURVEY 8e's collective behind the switch: reduce-scatter per bucket + owned-range AdamW + parameter all-gather, all of them
        # issued on RCCL with one rank (every rank-r shard is the whole bucket): same losses, same gradients, same updated weights
        r, ir = run(dist, True, "rs_ag")
        assert ir["flat_params"] and not ic["flat_params"]
        assert all(abs(x - y) <= 1e-5 * abs(x) for x, y in zip(ia["losses"], ir["losses"])), (ia, ir)
        # (bucket boundaries differ -- optimizer groups never share a bucket in this mode -- so compare the concatenation)
        ca, ca2, cr = (torch.cat([t.reshape(-1) for t in fl]) for fl in (a, a2, r))
        assert ca.numel() == cr.numel()
        scale, noise = float(ca.abs().max()), float((ca - ca2).abs().max())
        assert float((ca - cr).abs().max()) <= 10 * noise + 5e-4 * scale
        assert float((ir["probe"] - ic["probe"]).abs().max()) <= 1e-6 * float(ic["probe"].abs().max()) + 4e-7
        print(f"  rs_ag mode on one RCCL rank: gradients within {float((ca - cr).abs().max()) / scale:.2e} of the local path")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_clip_coefficient_deferred_into_the_fused_adamw_equals_the_explicit_scale():
    """BucketedGradReducer.clip_grad_norm_(defer_to=optimizer): the fused AdamW divides by `grad_scale` while it reads the gradients
    (Trainer(gradient_clip_val=0.5), main_style.py:110, + AdamW, model_wrapper_style.py:885-895) -- same parameters after the step as
    scaling the buckets first."""
    from torch import nn
    from styl3r_amd.ddp import BucketedGradReducer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(64, 128, device=dev)

    def run(defer):
        torch.manual_seed(1)
        m = nn.Sequential(nn.Linear(128, 256), nn.GELU(), nn.Linear(256, 32)).to(dev)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=0.05, betas=(0.9, 0.95), fused=True)
        red = BucketedGradReducer(m.parameters(), None, bucket_bytes=64 * 1024)
        for _ in range(3):
            red.prepare(); (m(x) * 7).pow(2).sum().backward(); red.finish()
            total = red.clip_grad_norm_(0.5, defer_to=opt if defer else None)
            assert float(total) > 0.5                       # the coefficient really is below one
            opt.step()
        return [p.detach().clone() for p in m.parameters()]
    a, b = run(True), run(False)
    for p, q in zip(a, b):
        assert float((p - q).abs().max()) <= 2e-5 * float(q.abs().max())        # g / s vs g * (1 / s): one rounding apart, through three Adam steps


@pytest.mark.gpu
def test_hip_adamw_pass_equals_the_frameworks_fused_adamw_and_shares_its_state_layout():
    """optim.AdamWHIP (csrc/vit_optim.hip, one launch per parameter group) against torch.optim.AdamW(fused=True) on the reference's
    configuration (two groups, lr / 0.1 lr, weight_decay 0.05, betas (0.9, 0.95), model_wrapper_style.py:885-895): parameters and both
    moments after 12 steps with a changing learning rate, a deferred clip coefficient (`grad_scale`) on some steps, tensors whose size is
    not a multiple of four, a misaligned gradient view, a parameter without gradient; then the state dicts are swapped between the two
    optimizers and both keep producing the same numbers."""
    from styl3r_amd.optim import AdamWHIP
    dev = "cuda:0"
    g = torch.Generator(dev).manual_seed(11)
    shapes = [(1024, 1024), (3, 7, 7, 5), (40000,), (1,), (16385,), (257, 129), (5,)]
    def make():
        ps = [torch.nn.Parameter(torch.randn(s, device=dev, generator=torch.Generator(dev).manual_seed(i))) for i, s in enumerate(shapes)]
        return ps, [{"params": ps[:4], "lr": 2e-4}, {"params": ps[4:], "lr": 2e-5}]
    pa, ga = make(); pb, gb = make()
    oa = torch.optim.AdamW(ga, lr=2e-4, weight_decay=0.05, betas=(0.9, 0.95), fused=True)
    ob = AdamWHIP(gb, lr=2e-4, weight_decay=0.05, betas=(0.9, 0.95))
    flat = torch.empty(sum(p.numel() for p in pb) + 8, device=dev)      # gradients of `b` are views of one buffer, like the all-reduce buckets
    def step(k, swap=False):
        off = 1                                                          # every view starts 4 bytes off a 16-byte boundary
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 6 and k % 2 == 0:
                a.grad = None; b.grad = None                              # a parameter that was not used in this step
                continue
            gr = torch.randn(a.shape, device=dev, generator=g) * (10.0 if k == 3 else 1.0)
            a.grad = gr.clone()
            view = flat[off:off + a.numel()].view(a.shape); view.copy_(gr); b.grad = view
            off += a.numel()
        for o in (oa, ob):
            for gi, grp in enumerate(o.param_groups):
                grp["lr"] = (2e-4 if gi == 0 else 2e-5) * (0.5 + 0.1 * k)
            if k % 3 == 1:
                o.grad_scale = torch.tensor(1.7 + k, device=dev)          # == gradients multiplied by 1 / (1.7 + k)
            elif hasattr(o, "grad_scale"):
                del o.grad_scale
            o.step()
    def compare(tag):
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) <= 2e-6, (tag, i, "param")
            sa, sb = oa.state.get(a, {}), ob.state.get(b, {})
            assert (len(sa) == 0) == (len(sb) == 0)
            if sa:
                assert float(sa["step"]) == float(sb["step"]) and sb["step"].is_cuda and sb["step"].dtype == torch.float32
                for k_ in ("exp_avg", "exp_avg_sq"):
                    assert float((sa[k_] - sb[k_]).abs().max() / sa[k_].abs().max().clamp_min(1e-30)) <= 2e-6, (tag, i, k_)
    for k in range(12):
        step(k)
    compare("12 steps")
    assert float(oa.state[pa[6]]["step"]) == 6.0                          # per-parameter step counters, like the framework's
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    assert sd_a["param_groups"][0].keys() == sd_b["param_groups"][0].keys()
    oa.load_state_dict(sd_b); ob.load_state_dict(sd_a)                    # interchangeable checkpoints
    for k in range(12, 16):
        step(k)
    compare("after swapping the state dicts")


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["hip", "torch"])
@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3", "f16x3"])     # (f16x3 + hip: the split takes its scale from the |max| word the AdamW kernel folded)
def test_forward_sees_the_weights_the_optimizer_just_wrote(impl, mode, monkeypatch):
    """the kernels read PRE-SPLIT bf16 images of every Linear / convolution weight, cached on `Tensor._version`
    (vit_ops._SPLIT_CACHE).  An optimizer that writes parameters through raw pointers (optim.AdamWHIP) -- or the framework's fused AdamW,
    which does not bump the counter either -- must invalidate them, or the model keeps computing with its initial weights.  After every
    step the fused Linear (default kernel, ring kernel, input-gradient image) and the bf16x6 convolution must match the plain fp64
    product with the UPDATED fp32 weights, and differ from the product with the old ones."""
    from styl3r_amd import train, vit_ops
    dev = "cuda:0"
    monkeypatch.setattr(vit_ops, "LINEAR_MODE", mode)
    monkeypatch.setattr(train, "OPTIMIZER_IMPL", impl)
    g = torch.Generator(dev).manual_seed(5)
    lin_small = torch.nn.Parameter(torch.randn(256, 192, device=dev, generator=g) * 0.05)          # default kernel
    lin_ring = torch.nn.Parameter(torch.randn(3072, 1024, device=dev, generator=g) * 0.03)          # LDS-DMA ring kernel at M >= 4096
    conv = vit_ops.Conv2dX6(128, 128, 3, padding=1).to(dev)
    xs = torch.randn(300, 192, device=dev, generator=g)
    xr = torch.randn(4224, 1024, device=dev, generator=g)
    xc = torch.randn(2, 128, 64, 64, device=dev, generator=g)
    opt = train.make_optimizer([lin_small, lin_ring], list(conv.parameters()), lr=5e-2)
    tol = 2e-4 if mode == "bf16x3" else 2e-5
    rel = lambda a, e: float((a.double() - e).abs().max() / e.abs().max())

    def products():
        xs_ = xs.clone().requires_grad_(True)
        ys = vit_ops.fused_linear(xs_, lin_small)
        yr = vit_ops.fused_linear(xr, lin_ring)
        yc = conv(xc)
        (ys.sum() + yr.mean() + yc.mean()).backward()
        return ys.detach(), yr.detach(), yc.detach(), xs_.grad.detach()

    def exact():
        return (xs.double() @ lin_small.detach().double().t(), xr.double() @ lin_ring.detach().double().t(),
                torch.nn.functional.conv2d(xc.double(), conv.weight.detach().double(), conv.bias.detach().double(), padding=1),
                torch.ones(300, 256, device=dev, dtype=torch.float64) @ lin_small.detach().double())

    before = dict(vit_ops.CALLS)
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        got, want = products(), exact()
        for name, a, e in zip(("linear", "ring linear", "conv", "linear dX"), got, want):
            assert rel(a, e) <= tol, (impl, mode, step, name, rel(a, e))
        old = want
        opt.step()
        new = exact()
        assert all(rel(n.float(), o) > 1e-2 for n, o in zip(new, old)), "the step must move the weights for this test to mean anything"
    assert vit_ops.CALLS["conv_x6_fwd"] > before["conv_x6_fwd"]
    # and the last update is visible, too
    for name, a, e in zip(("linear", "ring linear", "conv", "linear dX"), products(), exact()):
        assert rel(a, e) <= tol, (impl, mode, "final", name, rel(a, e))


@pytest.mark.gpu
def test_full_size_style_stage_step_trains_only_the_stylizer_and_accumulates_both_passes_in_place():
    """VERDICT r03 missing #3: ONE C4 step on the FULL-SIZE stylized encoder (b = 1, 4 context / 6 target views 256 x 256, VGG style
    loss + identity pass: two encoder + decoder passes share every weight, model_wrapper_style.py:189,211-231) with the in-place bucket
    slots, against the same step with `inplace_grads=False` (autograd accumulates the two passes itself, the reducer copies):
      * only `token_stylizer.*` and `gaussian_appearance_head.*` receive gradients; everything else is frozen (`:854-868`);
      * the in-place slots hold the SUM of both passes (equal to the copy path's gradients to within the weight-gradient kernels'
        atomics noise), i.e. the second pass accumulated into the first one's slot instead of overwriting it;
      * the loss is finite and the step renders a real scene."""
    from styl3r_amd import rasterizer, vit_ops
    from styl3r_amd.ddp import BucketedGradReducer
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    from styl3r_amd.losses import IdentityLoss, LossStyle, VGGEncoder
    from styl3r_amd.scenes import make_scene, recentre_output_heads_
    from styl3r_amd.train import TrainStep
    dev = torch.device("cuda:0")
    b, v_ctx, v_tgt, H = 1, 4, 6, 256
    sc = make_scene(n_ctx=v_ctx, grid_hw=(8, 8), n_views=v_tgt, image_hw=(H, H), seed=21)
    g = torch.Generator(dev).manual_seed(9)
    ex = lambda t, *shape: t.to(dev)[None].expand(b, *shape).contiguous()
    batch = dict(context=dict(image=torch.rand(b, v_ctx, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(b, v_ctx, 3, 3).contiguous()),
                 target=dict(image=torch.rand(b, v_tgt, 3, H, H, device=dev, generator=g), extrinsics=ex(sc.extrinsics, -1, -1, -1),
                             intrinsics=ex(sc.intrinsics, -1, -1, -1), near=ex(sc.near, -1), far=ex(sc.far, -1)),
                 style=dict(image=torch.rand(b, 3, H, H, device=dev, generator=g)))
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)

    def run(inplace):
        torch.manual_seed(0); torch.cuda.manual_seed(0)
        with torch.device(dev):
            enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=True)).eval()      # eval: no dropout RNG in the comparison
            vgg = VGGEncoder()
        recentre_output_heads_(enc, batch["context"], dict(image=(batch["style"]["image"] - 0.5) / 0.5))
        step = TrainStep(enc, dec, losses=[LossStyle(vgg=vgg)], identity_loss=IdentityLoss(vgg=vgg), warm_up_steps=2000)
        if not inplace:
            step.reducer.close()
            step.reducer = BucketedGradReducer([p for p in enc.parameters() if p.requires_grad], None, 64 << 20, inplace_grads=False)
        before = dict(vit_ops.CALLS)
        loss = float(step(batch))
        grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in enc.named_parameters()}
        adopted = sum(1 for bk in step.reducer.buckets for p, v in zip(bk["params"], bk["views"]) if p.grad is not None and p.grad.data_ptr() == v.data_ptr())
        pairs = dict(rasterizer.LAST_STATS)
        step.reducer.close()
        del step, enc, vgg
        torch.cuda.empty_cache()
        return loss, grads, adopted, pairs, {k: vit_ops.CALLS[k] - before[k] for k in before}

    l_in, g_in, adopted, pairs, took = run(True)
    l_cp, g_cp, _, _, _ = run(False)
    l_cp2, g_cp2, _, _, _ = run(False)                        # the copy path's own run-to-run noise (fp32 atomics)
    assert all(map(lambda x: x == x and abs(x) < 1e30, (l_in, l_cp))) and abs(l_in - l_cp) <= 1e-4 * abs(l_cp), (l_in, l_cp)
    assert pairs["pairs"] > pairs["gaussians_per_scene"], pairs
    assert adopted > 100, adopted                             # the Linear weights' gradients live in the bucket slots
    assert took["layernorm_framework"] == 0 and took["conv_x6_fwd"] > 0, took
    with_grad = {n for n, t in g_in.items() if t is not None}
    assert with_grad and all(n.startswith("token_stylizer.") or n.startswith("gaussian_appearance_head.") for n in with_grad), sorted(with_grad)[:5]
    assert any(n.startswith("token_stylizer.enc_blocks.23.") for n in with_grad) and any(n.startswith("gaussian_appearance_head.") for n in with_grad)
    assert {n for n, t in g_cp.items() if t is not None} == with_grad
    worst = 0.0
    for n in sorted(with_grad):
        scale = float(g_cp[n].abs().max())
        noise = float((g_cp2[n] - g_cp[n]).abs().max())
        err = float((g_in[n] - g_cp[n]).abs().max())
        worst = max(worst, err / max(scale, 1e-30))
        # a slot that kept only ONE of the two passes would be off by O(scale)
        assert err <= 5e-4 * scale + 4 * noise + 1e-9, (n, err, noise, scale)
    print(f"  full-size C4 step: {len(with_grad)} trained tensors, in-place vs copy path worst deviation {worst:.2e} of the tensor scale; loss {l_in:.5f}; pairs {pairs}")
