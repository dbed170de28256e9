#!/usr/bin/env python
"""bench.py -- headline benchmark of the Styl3R hot path on MI355X.

Metric (BASELINE.json): 256x256 stylized views/sec, forward + backward, at a
fixed ~65k Gaussians per scene.  One "step" = one pass of the decoder hot path
(DecoderSplattingHIP.forward -> MSE -> backward) over one batch of synthetic
scenes: B scenes x Vt target views of 256x256, G = 65 536 Gaussians per scene,
inputs resident in HBM before the timed region.  Multi-GPU: one process per GPU,
scenes sharded on the batch axis (weak scaling, no data-path collective -- the
raster path has no exchange step).

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline":     live hipEvent timing of the dominant kernel vs the HBM peak
  "cpu_baseline": the CPU oracle (a port of the published algorithm; the
                  reference has no CPU rasterizer) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes", type=int, default=10, help="scenes per GPU per step (C3 batch)")
    ap.add_argument("--views", type=int, default=4, help="target views per scene")
    ap.add_argument("--ctx", type=int, default=1, help="context views per scene (grid x grid Gaussians each)")
    ap.add_argument("--grid", type=int, default=256, help="Gaussians per context view = grid^2 (256 -> 65 536; sweep: 128, 512)")
    ap.add_argument("--sh-degree", type=int, default=0)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=2, help="views in the CPU-baseline sample")
    return ap.parse_args()


def build_batch(args, rank, dev):
    from styl3r_amd.decoder import Gaussians
    from styl3r_amd.scenes import make_scene
    from styl3r_amd.dist_utils import scene_seeds
    scenes = [make_scene(n_ctx=args.ctx, grid_hw=(args.grid, args.grid), n_views=args.views, image_hw=(args.res, args.res),
                         sh_degree=args.sh_degree, seed=sd) for sd in scene_seeds(rank, args.scenes)]
    st = lambda name: torch.stack([getattr(s, name) for s in scenes]).to(dev)
    g = Gaussians(st("means"), st("covariances"), st("harmonics"), st("opacities"))
    cams = dict(extrinsics=st("extrinsics"), intrinsics=st("intrinsics"), near=st("near"), far=st("far"))
    return scenes, g, cams


def algorithmic_bytes(stage, V, B, G, P, n_sh, R, R_eff):
    """SURVEY.md section 8d per-view figures x the views one launch processes."""
    return {
        "preprocess": B * G * (40 + 12 * n_sh) + V * G * (48 + 4),
        "scan_tiles": 0,
        "scatter": V * G * 16 + 8 * R,
        "tile_sort": 12 * R,
        "composite_fwd": 44 * R_eff + 28 * V * P,
        "composite_bwd": 44 * R_eff + 24 * V * P + 44 * V * G,
        "preprocess_bwd": B * G * (40 + 12 * n_sh) + V * G * (48 + 16) + B * G * (40 + 12 * n_sh),
    }[stage]


def cpu_baseline(args, scenes):
    """Oracle (kind 'port') on `cpu_views` views of scene 0, fwd + bwd, all host cores in the OpenMP loops."""
    from oracle.gsr_oracle import Oracle
    from styl3r_amd.decoder import prepare_views
    sc = scenes[0]
    orc = Oracle("f32")
    cores = os.cpu_count() or 1
    nv = min(args.cpu_views, args.views)
    views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(args.views, 3), True).numpy()
    cov = sc.covariances.numpy()
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
    shs = sc.harmonics.numpy().transpose(0, 2, 1)
    H = W = args.res
    gI = np.ones((3, H, W), np.float32) / (3 * H * W)
    t0 = time.perf_counter()
    for v in range(nv):
        row = views[v]
        s = np.float32(row[56])
        st, ctx = orc.forward(sc.means.numpy() * s, cov6 * (s * s), sc.opacities.numpy(), shs=shs, H=H, W=W,
                              tanfovx=row[51], tanfovy=row[52], bg=(0, 0, 0), view=row[0:16], proj=row[16:32],
                              proj_raw=row[32:48], campos=row[48:51], sh_degree=args.sh_degree, nthreads=cores)
        orc.backward(st, ctx, gI, None, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(nv / dt, 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{nv} of the {args.views} views of scene 0 (G={sc.means.shape[0]}, {H}x{W}), fwd+bwd, "
                      f"oracle/gsr_oracle.c f32, OpenMP over tiles"}


def main():
    args = parse()
    from styl3r_amd import dist_utils
    rank, local_rank, world = dist_utils.env_world()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = dist_utils.init_distributed("nccl", dev)   # "nccl" is RCCL on ROCm

    from styl3r_amd import _lib, rasterizer as rz
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.losses import mse_loss
    _lib.load()  # fail loudly if the HIP library is missing

    scenes, g, cams = build_batch(args, rank, dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    H = W = args.res
    B, Vt = args.scenes, args.views
    V, G = B * Vt, g.means.shape[1]
    target = torch.rand((B, Vt, 3, H, W), device=dev, generator=torch.Generator(dev).manual_seed(7 + rank))
    for t in (g.means, g.covariances, g.harmonics, g.opacities):
        t.requires_grad_(True)

    def step():
        for t in (g.means, g.covariances, g.harmonics, g.opacities):
            t.grad = None
        out = dec.forward(g, cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W))
        loss = mse_loss(out.color, target)              # LossMse (src/loss/loss_mse.py:22-31) on gsr_mse_forward/backward
        loss.backward()
        return loss

    for _ in range(args.warmup):
        step()
    prof = _lib.StageProfile(args.steps + 1)
    rz.PROFILE = prof
    dt = dist_utils.timed_steps(step, args.steps, lambda: torch.cuda.synchronize(dev), dist, dev)
    rz.PROFILE = None
    stage_ms = prof.read()
    prof.close()

    # ---- per-launch algorithmic bytes of every stage (one extra un-timed forward to read R / n_contrib) ----
    rz.KEEP_DEBUG = True
    with torch.no_grad():
        dec.forward(g, cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W))
    dbg = rz.LAST_DEBUG
    R = int(dbg["num_pairs"])
    off = dbg["layout"].n_contrib
    nc = dbg["ws"][off:off + V * H * W * 4].view(torch.int32).reshape(V, H // 16, 16, W // 16, 16)
    R_eff = int(nc.amax(dim=(2, 4)).sum().item())
    rz.KEEP_DEBUG = False
    rz.LAST_DEBUG.clear()
    n_sh = (args.sh_degree + 1) ** 2
    stages = {}
    for name, (ms, cnt) in stage_ms.items():
        if cnt == 0:
            continue
        avg = ms / cnt
        by = algorithmic_bytes(name, V, B, G, H * W, n_sh, R, R_eff)
        stages[name] = {"avg_ms": round(avg, 4), "alg_bytes": int(by),
                        "GBps": round(by / (avg * 1e-3) / 1e9, 1) if avg > 0 else None}
    dominant = max(stages, key=lambda k: stages[k]["avg_ms"])
    dk = stages[dominant]
    # HBM traffic / VALU occupancy of the same kernel from the committed rocprofv3 --pmc passes of this workload
    # (profiles/pmc_latest.json, produced by tools/pmc_run.sh + tools/pmc_summary.py; FETCH_SIZE doubled as the
    # MI355X guide prescribes for gfx950).  null when the file does not cover the kernel.
    traffic, valu_busy = None, None
    pmc = ROOT / "profiles" / "pmc_latest.json"
    if pmc.exists() and (B, Vt, args.ctx, args.grid, args.res, args.sh_degree) == (10, 4, 1, 256, 256, 0):
        try:
            rec = json.loads(pmc.read_text()).get(dominant, {})
            traffic, valu_busy = rec.get("hbm_bytes_per_launch"), rec.get("valu_busy_frac")
        except Exception:
            pass
    roofline = {"kernel": "k_" + dominant, "bound": "hbm", "achieved": dk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dk["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "valu_busy_frac_pmc": valu_busy,
                "alg_bytes_per_launch": dk["alg_bytes"], "avg_launch_ms": dk["avg_ms"],
                "pairs_R": R, "R_eff": R_eff, "stages": stages}

    if rank == 0:
        res = {
            "metric": "256x256 stylized views/sec (fwd+bwd) @ ~65k Gaussians",
            "value": round(dist_utils.aggregate_throughput(V, args.steps, world, dt), 2), "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"rasterizer fwd+bwd (decoder API + MSE): {B} scenes x {Vt} target views/GPU/step, "
                                   f"{H}x{W}, G={G} Gaussians/scene ({args.ctx} ctx view x {args.grid}x{args.grid}), sh_degree="
                                   f"{args.sh_degree}, make_scale_invariant, all views in one batched launch",
                       "views_per_step_per_gpu": V, "gaussians_per_scene": G, "parallelism": f"dp{world} (scenes sharded)"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, scenes)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
