#!/usr/bin/env python
"""M2: full train-step rendered-views/sec -- encoder fwd+bwd (full-size ViT-L/ViT-B style-token
encoder, random init: checkpoints are absent) + batched rasterizer fwd+bwd + MSE + bucketed DP all-reduce +
clip + AdamW, C3 shapes (2 context views 256x256 -> 131 072 Gaussians per scene, 4 target views).
Secondary benchmark (the headline bench.py is the raster-only M1).  One JSON line on rank 0.
  python tools/bench_train.py --scenes 4 --steps 3            (1 GPU)
  python -m torch.distributed.run --nproc-per-node N tools/bench_train.py ...   (DP over RCCL)
"""
import argparse, json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from styl3r_amd import dist_utils
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene
from styl3r_amd.train import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=4); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1); ap.add_argument("--tiny", action="store_true")
ap.add_argument("--linear-mode", choices=["bf16x6", "bf16x3", "f16x3", "f32"], default=None, help="arithmetic of the Linear / convolution kernels (default: VIT_LINEAR_MODE or bf16x6)")
ap.add_argument("--config", choices=["c3", "c4", "c5"], default="c3",
                help="c3: NVS-pretrain, 2 ctx / 4 tgt views, MSE, everything trains.  c4: style stage, 4 ctx / 6 tgt views, "
                     "VGG style loss + identity pass (two encoder/decoder passes), backbone frozen (random-init VGG: no weights here).  "
                     "c5: stress shapes, 4 ctx views 512x512 -> 1 048 576 Gaussians/scene, sh_degree 4, 4 tgt views 512x512, MSE, all train")
args = ap.parse_args()
if args.linear_mode:
    from styl3r_amd import vit_ops as _vo
    _vo.LINEAR_MODE = args.linear_mode
    if args.linear_mode in ("bf16x3", "bf16x6", "f16x3"):
        _vo.ATTENTION_ARITH = args.linear_mode          # one arithmetic mode for every GEMM-shaped kernel of the step
rank, local_rank, world = dist_utils.env_world()
torch.cuda.set_device(local_rank); dev = torch.device("cuda", local_rank)
dist = dist_utils.init_distributed("nccl", dev)
torch.manual_seed(0)
tiny = dict(enc_depth=2, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512)) if args.tiny else None
c4, c5 = args.config == "c4", args.config == "c5"
from styl3r_amd.encoder import GaussianAdapterCfg
cfg = EncoderNoPoSplatTokenStyleCfg(stylized=c4)
if c5:
    cfg.gaussian_adapter = GaussianAdapterCfg(cfg.gaussian_adapter.gaussian_scale_min, cfg.gaussian_adapter.gaussian_scale_max, 4)
enc = EncoderNoPoSplatMultiTokenStyle(cfg, trunk_params=tiny).to(dev)
# the reference's xavier init gives scales ~1e-3 softplus(0): keep default torch init (random weights, data=synthetic)
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
b, v_ctx, v_tgt, H = args.scenes, (4 if (c4 or c5) else 2), (6 if c4 else 4), (512 if c5 else 256)
g = torch.Generator(dev).manual_seed(1234 + rank)
sc = make_scene(n_ctx=v_ctx, grid_hw=(8, 8), n_views=v_tgt, image_hw=(H, H), seed=1234 + rank)
K = sc.intrinsics[:1].to(dev)
batch = dict(
    context=dict(image=torch.rand(b, v_ctx, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=K.expand(b, v_ctx, 3, 3).contiguous()),
    target=dict(image=torch.rand(b, v_tgt, 3, H, H, device=dev, generator=g), extrinsics=sc.extrinsics.to(dev)[None].expand(b, -1, -1, -1).contiguous(),
                intrinsics=sc.intrinsics.to(dev)[None].expand(b, -1, -1, -1).contiguous(), near=sc.near.to(dev)[None].expand(b, -1).contiguous(),
                far=sc.far.to(dev)[None].expand(b, -1).contiguous()))
if c4:
    batch["style"] = dict(image=torch.rand(b, 3, H, H, device=dev, generator=g))
# set-up (untimed): a random-init point head throws the Gaussians outside every frustum; re-centre the five output convolutions so the step renders a real scene
from styl3r_amd import rasterizer
from styl3r_amd.scenes import recentre_output_heads_
recentre_output_heads_(enc, batch["context"], dict(image=(batch["style"]["image"] - 0.5) / 0.5) if c4 else dict(image=batch["context"]["image"][:, 0]))
if c4:
    from styl3r_amd.losses import IdentityLoss, LossStyle, VGGEncoder
    vgg = VGGEncoder().to(dev)
    step = TrainStep(enc, dec, dist=dist, losses=[LossStyle(vgg=vgg)], identity_loss=IdentityLoss(vgg=vgg), warm_up_steps=2000)
else:
    step = TrainStep(enc, dec, dist=dist, warm_up_steps=2000)      # config/main.yaml:37
for _ in range(args.warmup):
    step(batch)
assert rasterizer.LAST_STATS["pairs"] > rasterizer.LAST_STATS["gaussians_per_scene"], rasterizer.LAST_STATS
dt = dist_utils.timed_steps(lambda: step(batch), args.steps, lambda: torch.cuda.synchronize(dev), dist, dev)
if rank == 0:
    nparam = sum(p.numel() for p in enc.parameters())
    print(json.dumps({"metric": ("512x512" if c5 else "256x256") + " rendered views/sec, full train step (encoder+rasterizer fwd+bwd, AdamW, DP all-reduce)",
                      "config": args.config + (" style stage: 4 ctx / 6 tgt views, VGG style + identity pass, backbone frozen" if c4 else
                                               " stress: 4 ctx views 512x512 (1 048 576 Gaussians/scene), sh_degree 4, 4 tgt views 512x512" if c5 else
                                               " NVS-pretrain: 2 ctx / 4 tgt views, MSE, all parameters train"),
                      "value": round(dist_utils.aggregate_throughput(b * v_tgt, args.steps, world, dt), 3), "unit": "views/s",
                      "n_gpus": world, "ms_per_step": round(1e3 * dt / args.steps, 2), "scenes_per_gpu": b, "params": nparam,
                      "grad_bytes": 4 * sum(p.numel() for p in enc.parameters() if p.requires_grad),
                      "buckets": len(step.reducer.buckets), "peak_mem_GB": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
                      "dtype": "f32", "linear_arithmetic": __import__("styl3r_amd.vit_ops", fromlist=["x"]).LINEAR_MODE
, "data": "synthetic, random-init weights"}))
if dist is not None:
    dist.barrier(); dist.destroy_process_group()
