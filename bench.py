#!/usr/bin/env python
"""bench.py -- headline benchmark of the Styl3R hot path on MI355X.

Metric (BASELINE.json): 256x256 stylized views/sec, forward + backward, at a
fixed ~65k Gaussians per scene.  One "step" = one pass of the decoder hot path
(DecoderSplattingHIP.forward -> MSE -> backward) over one batch of synthetic
scenes: B scenes x Vt target views of 256x256, G = 65 536 Gaussians per scene,
inputs resident in HBM before the timed region.

Multi-GPU (`--gpus N`): one process per GPU.  Under `torch.distributed.run` the ranks come from
RANK / LOCAL_RANK / WORLD_SIZE; a bare `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the
environment) re-executes itself under `torch.distributed.run` with N ranks on 127.0.0.1, the launch the
reference gets from Lightning's DDP strategy (src/main_style.py:98-118: devices="auto", per-rank seed).
The raster leg shards scenes on the batch axis with no data-path collective (weak scaling);
the `train_step` leg is the C3 optimisation step (full-size encoder fwd+bwd + rasterizer fwd+bwd + MSE +
bucketed gradient all-reduce over RCCL overlapped with the backward + clip + AdamW) -- the one collective the
path has, so the 1 -> N curve contains it.

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline":     live hipEvent timing of the dominant kernel vs the HBM peak, what actually bounds it
  "cpu_baseline": the CPU oracle (a port of the published algorithm; the reference has no CPU rasterizer)
                  on a bounded sample
  "train_step":   M2: C3 ms/step, rendered views/s, bytes all-reduced, RCCL ranks
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-seconds", type=float, default=1.5, help="keep the device busy this long (untimed) before the warm-up steps so the clocks are up")
    ap.add_argument("--scenes", type=int, default=10, help="scenes per GPU per step (C3 batch)")
    ap.add_argument("--views", type=int, default=4, help="target views per scene")
    ap.add_argument("--ctx", type=int, default=1, help="context views per scene (grid x grid Gaussians each)")
    ap.add_argument("--grid", type=int, default=256, help="Gaussians per context view = grid^2 (256 -> 65 536; sweep: 128, 512)")
    ap.add_argument("--sh-degree", type=int, default=0)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=2, help="minimum number of views in the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline sample: keep taking views until this much wall time is spent")
    ap.add_argument("--no-infer-leg", action="store_true", help="skip the C2 inference leg")
    ap.add_argument("--no-dropin-leg", action="store_true", help="skip the per-view drop-in leg (the reference's own call pattern)")
    ap.add_argument("--dp-ab", action="store_true", help="train leg at N > 1: also time the step in the other data-parallel mode (all_reduce <-> rs_ag); automatic at N = 1 under torch.distributed.run")
    ap.add_argument("--no-comm-report", action="store_true", help="train leg: skip the exchange diagnosis / data-parallel mode A/B (runs only with a process group)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the C3 train-step leg (M2)")
    ap.add_argument("--no-stage-legs", action="store_true", help="skip the C4 style-stage and C5 stress train steps")
    ap.add_argument("--train-scenes", type=int, default=10, help="scenes per GPU per train step (C3: 10)")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--train-warmup", type=int, default=3)
    ap.add_argument("--train-mode", choices=["bf16x3", "bf16x6", "f16x3"], default="f16x3", help="arithmetic of the GEMM-shaped kernels in the headline train step")
    ap.add_argument("--train-tiny", action="store_true", help="small trunk for the train leg (smoke tests only; flagged in the line)")
    ap.add_argument("--dry-cpu", action="store_true",
                    help="launch-path test mode: gloo on CPU, no kernels, no oracle -- checks spawning / sharding / the JSON line")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ launch
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_torchrun(args) -> int:
    """`python bench.py --gpus N` without a torch.distributed.run environment: start N ranks of this script, one per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", STYL3R_BENCH_SPAWNED="1")
    return subprocess.run(cmd, env=env).returncode


ARITH_NOTES = {
    "f16x3": ("two fp16 pieces per operand x per-tensor power-of-two scale, three products on the f16 MFMA with fp32 accumulation (attention: six bf16 products): "
              "fp32 round-off accuracy -- Gaussians <= 1e-5 of the float64 reference at the model's full depth (tests/test_e2e_parity.py[full-f16x3], "
              "tests/golden/e2e_full.npz) -- at the MFMA count of bf16x3"),
    "bf16x6": "bf16x6 split products (six bf16 MFMAs per fp32 product): fp32 round-off accuracy, the library default",
    "bf16x3": ("three bf16 products per fp32 product: inside the reference's own TF32 distance on every quantity (worst ratio 0.13 .. 0.21) but covariances at "
               "1.3e-4 of float64 at full depth -- above north_star's 1e-4, hence not the headline mode"),
}


# ------------------------------------------------------------------ raster leg
def build_batch(args, rank, dev):
    import torch
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
    """SURVEY.md section 8d per-view figures x the views one launch processes.
    tile_sort: what this design's K4 moves per (tile, Gaussian) pair -- 8 B key read + 4 B sorted id written (12R).  The
    composite stages stay priced with SURVEY 8d's 44 B per staged entry (they now gather the 48-B record themselves)."""
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
    """Oracle (kind 'port') on a bounded sample of the SAME workload: views of the step's own scenes, fwd + bwd, all host cores in the
    OpenMP loops, until `--cpu-seconds` (default 10 s) of CPU work or the whole step's views are done."""
    import numpy as np
    import torch
    from oracle.gsr_oracle import Oracle
    from styl3r_amd.decoder import prepare_views
    orc = Oracle("f32")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)   # cores this process may use
    H = W = args.res
    gI = np.ones((3, H, W), np.float32) / (3 * H * W)
    done, t0 = 0, time.perf_counter()
    G = scenes[0].means.shape[0]
    for sc in scenes:
        views = prepare_views(sc.extrinsics, sc.intrinsics, sc.near, sc.far, torch.zeros(args.views, 3), True).numpy()
        cov = sc.covariances.numpy()
        cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
        shs = sc.harmonics.numpy().transpose(0, 2, 1)
        for v in range(args.views):
            row = views[v]
            s = np.float32(row[56])
            st, ctx = orc.forward(sc.means.numpy() * s, cov6 * (s * s), sc.opacities.numpy(), shs=shs, H=H, W=W,
                                  tanfovx=row[51], tanfovy=row[52], bg=(0, 0, 0), view=row[0:16], proj=row[16:32],
                                  proj_raw=row[32:48], campos=row[48:51], sh_degree=args.sh_degree, nthreads=cores)
            orc.backward(st, ctx, gI, None, nthreads=cores)
            done += 1
            if done >= max(args.cpu_views, 1) and time.perf_counter() - t0 >= args.cpu_seconds:
                break
        else:
            continue
        break
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"{done} of the step's {len(scenes) * args.views} views (scene order, G={G}, {H}x{W}), fwd+bwd, {dt:.1f} s of wall time, "
                      f"oracle/gsr_oracle.c f32, OpenMP: per-Gaussian stages (preprocess, key emission + radix sort, preprocess-backward) on "
                      f"min(cores, 32) threads, per-tile stages (composite forward / backward) on all cores"}


def cpu_encoder_baseline(args):
    """SURVEY 8d / BASELINE.md 3: the ENCODER's CPU number beside the GPU train step.  The reference has no separate CPU implementation; what
    runs here is this repo's restatement of its modules (styl3r_amd.encoder: CPU tensors take the framework's fp32 ops -- the same arithmetic the
    reference's modules run on a CPU; the attention, which has no CPU path in the product, comes from oracle/encoder_cpu.py) at FULL size (1 049 635 033 parameters), one scene of 2 context views 256 x 256, forward + backward of a
    scalar loss on the Gaussians, on the box's host cores.  Two calls, the second (warm) one is the number, the cold one is reported beside it (~2 x 10 - 30 s of CPU work); rank 0 only."""
    import torch
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(cores, 128)
    keep = torch.get_num_threads()
    try:
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False)).eval()
        g = torch.Generator().manual_seed(1234)
        K = torch.tensor([[0.86, 0, 0.5], [0, 0.86, 0.5], [0, 0, 1.0]]).expand(1, 2, 3, 3).contiguous()
        img = torch.rand(1, 2, 3, 256, 256, generator=g) * 2 - 1
        from oracle.encoder_cpu import cpu_attention      # (the encoder's attention kernels have no CPU path: plain-torch restatement, baseline leg only)
        def one():
            t0 = time.perf_counter()
            gs = enc(dict(image=img, intrinsics=K), dict(image=img[:, 0]), 0)
            t1 = time.perf_counter()
            loss = gs.means.tanh().sum() * 1e-3 + gs.opacities.sum() * 1e-3 + (gs.harmonics ** 2).sum() * 1e-3 + gs.covariances.sum()
            loss.backward()
            t2 = time.perf_counter()
            enc.zero_grad(set_to_none=True)
            return t1 - t0, t2 - t1
        with cpu_attention():
            cold = one()          # thread-pool start-up, first-touch allocation of 1.05 B parameters' gradients, oneDNN primitive creation
            warm = one()          # (the cold call alone was a pessimistic baseline next to warmed GPU numbers)
        tot = warm[0] + warm[1]
        return {"seconds_per_scene_fwd_bwd": round(tot, 2), "forward_s": round(warm[0], 2), "backward_s": round(warm[1], 2),
                "cold_seconds_per_scene_fwd_bwd": round(cold[0] + cold[1], 2), "threads": threads, "cores": cores,
                "equivalent_views_per_s": round(4 / tot, 3), "kind": "port",
                "sample": "1 scene (2 context views 256x256 -> 131 072 Gaussians), forward + backward of the full-size encoder (1.05 B parameters, fp32, framework CPU ops): "
                          "the SECOND of two calls (warm); the first, cold call is reported beside it; "
                          "equivalent_views_per_s = the C3 step's 4 target views per scene / the warm time (the rasterizer's CPU time, `value` above, comes on top)"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        torch.set_num_threads(keep)


def pmc_record(kernel, headline_workload):
    """Counter-derived figures of `kernel` from profiles/pmc_latest.json (rocprofv3 --pmc passes of this very command,
    tools/pmc_run.sh -> tools/pmc_summary.py -> tools/pmc_latest.py).  They are reported only when the file was measured
    on the SAME BUILD (digest of the kernel sources + flags, stamped into the file and next to libgsr_hip.so) and the run is
    the headline workload; otherwise null -- never a stale number."""
    from styl3r_amd import _lib
    pmc = ROOT / "profiles" / "pmc_latest.json"
    out = {"traffic": None, "pmc_source": None, "valu": None}
    if not (pmc.exists() and headline_workload):
        return out
    try:
        doc = json.loads(pmc.read_text())
    except Exception:
        return out
    if doc.get("build_digest") != _lib.built_digest() or not _lib.built_digest():
        out["pmc_source"] = f"stale: pmc_latest.json was measured on build {doc.get('build_digest')}, this is {_lib.built_digest()}"
        return out
    rec = doc.get("kernels", {}).get(kernel, {})
    out["traffic"] = rec.get("hbm_bytes_per_launch")
    out["pmc_source"] = doc.get("source")
    out["valu"] = {k: rec.get(k) for k in ("valu_insts_per_launch", "valu_insts_per_pair", "valu_active_frac", "salu_insts_per_pair",
                                           "lds_insts_per_pair") if k in rec}
    return out


def raster_leg(args, rank, world, dev, dist):
    import torch
    from styl3r_amd import _lib, dist_utils, rasterizer as rz
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
        # LossMse (src/loss/loss_mse.py:22-31) computed inside the composite kernels (DecoderOutput.loss_mse, round 6)
        out = dec.forward(g, cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W), mse_target=target)
        out.loss_mse.backward()
        return out.loss_mse

    def step_unfused():       # the round-5 step: decoder, then LossMse as the stand-alone gsr_mse_forward / gsr_mse_backward kernels
        for t in (g.means, g.covariances, g.harmonics, g.opacities):
            t.grad = None
        out = dec.forward(g, cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W))
        loss = mse_loss(out.color, target)
        loss.backward()
        return loss

    # Everything that idles the device (profile buffers, a collector pass) happens BEFORE the clocks are brought up; from then on the
    # device is kept busy without a pause: pre-warm (a fresh, idle MI355X runs its first ~0.1 s at less than half speed and a short timed
    # region right behind a pause still sees the ramp: 21.2 k instead of 21.9 k views/s at --steps 20), the W warm-up steps, the K timed
    # ones.  The host is never more than one step ahead of the device (the forward reads its 32-byte status back), so a collector pause
    # inside the timed region would be device idle time: the collector stays off until the region is over.
    import gc
    prof = _lib.StageProfile(args.steps + 1)
    gc.collect(); gc.disable()
    try:
        t_end = time.perf_counter() + max(args.prewarm_seconds, 0.0)
        while time.perf_counter() < t_end:
            for _ in range(25):
                step()
            torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            step()
        # Inside the wall-clock-timed region only the two composite kernels (the roofline kernels) are bracketed by hipEvents: every timed stage
        # puts two event records between kernels that otherwise follow each other back to back (~5 us per boundary, ~25 us of a 1.45 ms step with
        # all seven on).  The other five stages are timed right behind it, same workload, in a short pass of their own.
        prof.set_stages(("composite_fwd", "composite_bwd"))
        rz.PROFILE = prof
        dt = dist_utils.timed_steps(step, args.steps, lambda: torch.cuda.synchronize(dev), dist, dev)
        stage_ms = prof.read()
        prof.set_stages(None)
        for _ in range(min(args.steps, 50)):
            step()
        torch.cuda.synchronize(dev)
        for name, val in prof.read().items():
            if not name.startswith("composite"):
                stage_ms[name] = val
        rz.PROFILE = None
        # A/B in the same run: the same workload with the loss as two kernels of its own (what `value` was measured on up to round 5)
        n_ab = min(args.steps, 100)
        for _ in range(5):
            step_unfused()
        torch.cuda.synchronize(dev)
        t_ab = time.perf_counter()
        for _ in range(n_ab):
            step_unfused()
        torch.cuda.synchronize(dev)
        unfused = {"ms_per_step": round(1e3 * (time.perf_counter() - t_ab) / n_ab, 4), "steps": n_ab,
                   "what": "decoder forward, then LossMse as the stand-alone gsr_mse_forward / gsr_mse_backward kernels (the step of rounds 1-5)"}
        loss_fused, loss_unfused = float(step().item()), float(step_unfused().item())
    finally:
        gc.enable()
    rz.PROFILE = None
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
    headline = (B, Vt, args.ctx, args.grid, args.res, args.sh_degree) == (10, 4, 1, 256, 256, 0)
    pm = pmc_record(dominant, headline)
    # `bound`: the class SURVEY 8d assigns the stage (the fraction below is against THAT roof, as north_star asks);
    # `limited_by`: what the counters say limits it on this workload
    roofline = {"kernel": "k_" + dominant, "bound": "hbm", "achieved": dk["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dk["GBps"] / HBM_PEAK_GBS, 4), "traffic": pm["traffic"],
                "limited_by": "valu-issue" if dominant.startswith("composite") else "hbm",
                "valu": pm["valu"], "pmc_source": pm["pmc_source"],
                # the roof this kernel actually sits under (VERDICT r04 #7b): VALU-busy / wall from the same counter file
                # (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles), tools/pmc_latest.py); the HBM fraction stays the north-star figure
                "valu_frac": (pm["valu"] or {}).get("valu_active_frac"),
                "alg_bytes_per_launch": dk["alg_bytes"], "avg_launch_ms": dk["avg_ms"],
                "ns_per_pair_per_simd": round(dk["avg_ms"] * 1e6 * 1024 / max(R, 1), 2),
                "pairs_R": R, "R_eff": R_eff, "stages": stages}
    # the pass north_star's ">= 0.5 x HBM roofline" names is the alpha-composite FORWARD: reported beside the dominant kernel's block
    cf = stages.get("composite_fwd")
    pf = pmc_record("composite_fwd", headline)
    roofline_cf = None if cf is None else {
        "kernel": "k_composite_fwd", "bound": "hbm", "achieved": cf["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(cf["GBps"] / HBM_PEAK_GBS, 4), "traffic": pf["traffic"], "limited_by": "valu-issue", "valu": pf["valu"],
        "valu_frac": (pf["valu"] or {}).get("valu_active_frac"),
        "alg_bytes_per_launch": cf["alg_bytes"], "avg_launch_ms": cf["avg_ms"], "target_frac": 0.5,
        "note": "SURVEY 8d byte model (44 B per staged entry + 28 B per pixel); the kernel is VALU-issue-bound, DESIGN.md section 6"}
    res = {
        "metric": "256x256 stylized views/sec (fwd+bwd) @ ~65k Gaussians",
        "value": round(dist_utils.aggregate_throughput(V, args.steps, world, dt), 2), "unit": "views/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"rasterizer fwd+bwd (decoder API + MSE, the loss computed inside the composite kernels): {B} scenes x {Vt} target views/GPU/step, "
                               f"{H}x{W}, G={G} Gaussians/scene ({args.ctx} ctx view x {args.grid}x{args.grid}), sh_degree="
                               f"{args.sh_degree}, make_scale_invariant, all views in one batched launch; outputs the decoder discards are not computed "
                               f"(decoder_splatting_cuda.py:37-68 drops radii / opacity / n_touched, MSE sends no depth gradient: the composite kernels run "
                               f"their n_touched-free / depth-gradient-free instantiations -- the consumed results are identical)",
                   "views_per_step_per_gpu": V, "gaussians_per_scene": G, "parallelism": f"dp{world} (scenes sharded)"},
        "roofline": roofline, "roofline_composite_fwd": roofline_cf,
        "mse_stand_alone_kernels": dict(unfused, views_per_s=round(V / (unfused["ms_per_step"] * 1e-3), 1), loss=loss_unfused, loss_fused=loss_fused),
    }
    if rank == 0 and not getattr(args, "no_dropin_leg", False):
        try:
            res["batched_full_outputs"] = full_outputs_leg(args, dev, g, cams, target)
        except Exception as e:
            res["batched_full_outputs"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        try:
            res["dropin_per_view"] = dropin_leg(args, dev, g, cams, target)
            # the same loop around a rasterizer that does nothing: what the reference's call pattern costs by itself (the split)
            ref_only = dropin_leg(args, dev, g, cams, target, stub=True)
            d = res["dropin_per_view"]
            d["reference_pattern_alone_ms_per_step"] = ref_only["ms_per_step"]
            d["module_share_ms_per_call"] = round((d["ms_per_step"] - ref_only["ms_per_step"]) / d["rasterizer_calls_per_step"], 4)
            d["split"] = ("reference_pattern_alone = the identical loop with a do-nothing rasterizer (per-view settings, 2 .item() syncs, repeat / index / select "
                          "autograd kernels, stack, loss); module_share = the rest per call: one-view launches of the composite kernels fill a quarter of the chip "
                          "with one wave per SIMD (profiles/r06_dropin_kernel_stats.md), the module's host time is ~0.18 ms of it and hidden behind them")
        except Exception as e:
            res["dropin_per_view"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del g, cams, target, dec
    torch.cuda.empty_cache()
    return res, scenes


# ------------------------------------------------------------------ batched launch with EVERY output of the 5-tuple live
def full_outputs_leg(args, dev, g, cams, target, steps=50, warmup=5):
    """the headline runs the n_touched-free / depth-gradient-free instantiations of the composite kernels (the Decoder
    API discards those outputs).  The same batched launch with everything SURVEY 8b's 5-tuple carries switched on -- n_touched counted
    (k_composite_fwd<true>), a loss on colour AND depth (k_composite_bwd<true>), screen-space mean gradients written -- through
    `rasterize_views`, same scenes / cameras / target."""
    import torch
    from math import isqrt
    from styl3r_amd.decoder import build_views_hip
    from styl3r_amd.losses import mse_loss
    from styl3r_amd.rasterizer import rasterize_views
    b, v = cams["extrinsics"].shape[:2]
    h = w = args.res
    flat = lambda t: t.reshape(b * v, *t.shape[2:])
    views = build_views_hip(flat(cams["extrinsics"]), flat(cams["intrinsics"]), flat(cams["near"]), flat(cams["far"]), torch.zeros(b * v, 3, device=dev), True)
    degree = isqrt(g.harmonics.shape[-1]) - 1
    G = g.means.shape[1]
    leaves = (g.means, g.covariances, g.harmonics, g.opacities)

    def step():
        for t in leaves:
            t.grad = None
        m2d = torch.zeros(b * v, G, 3, device=dev, requires_grad=True)
        out = rasterize_views(g.means, g.covariances, g.opacities, g.harmonics.permute(0, 1, 3, 2).contiguous(), views, (h, w), v,
                              sh_degree=degree, use_sh=True, means2D=m2d, want_n_touched=True)
        loss = mse_loss(out.image.view(b, v, 3, h, w), target) + 1e-3 * out.depth.mean()
        loss.backward()
        return out

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    return {"views_per_s": round(b * v / dt, 1), "ms_per_step": round(dt * 1e3, 4), "steps": steps, "n_touched_sum": int(out.n_touched.sum().item()),
            "what": "one batched launch sequence with n_touched, the depth gradient and the screen-space mean gradients live (the whole 5-tuple of the drop-in boundary)"}


# ------------------------------------------------------------------ drop-in leg (the reference's own call pattern)
def dropin_leg(args, dev, g, cams, target, steps=8, warmup=2, stub=False):
    """INTEGRATION.md section 1 ("zero source changes"): what `DecoderSplattingCUDA.forward` + `render_cuda` would drive 40 times per step
    through the drop-in module `diff_gaussian_rasterization` -- the Gaussian tensors replicated per target view
    (decoder_splatting_cuda.py:51-63), the 1/near rescale as tensor ops (cuda_splatting.py:65-72), and per view: two `.item()` host
    syncs, a fresh `mean_gradients` tensor, the 13-field settings tuple, the `[:, row, col]` gather of the covariances, ONE
    `GaussianRasterizer` call with n_touched counted (the 5-tuple of SURVEY 8b), images stacked at the end (cuda_splatting.py:93-132);
    MSE on the stacked colours, backward through all of it.  Same scenes, cameras and target as the headline leg."""
    import torch
    from einops import rearrange, repeat
    from math import isqrt
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from styl3r_amd.camera import get_fov, get_projection_matrix
    if stub:
        # the reference's call pattern ALONE: the same loop around a rasterizer that does nothing but hand back differentiable
        # placeholders -- what the per-view settings / .item() syncs / gathers / stack / loss cost without any rasterization
        class GaussianRasterizer(torch.nn.Module):      # noqa: F811
            def __init__(self, st):
                super().__init__(); self.st = st
            def forward(self, means3D, means2D, shs, colors_precomp, opacities, cov3D_precomp, theta, rho):
                z = (means3D.sum() + shs.sum() + opacities.sum() + cov3D_precomp.sum()) * 0
                img = z.expand(3, self.st.image_height, self.st.image_width)
                return img, None, None, None, None
    b, v = cams["extrinsics"].shape[:2]
    h = w = args.res
    bg = torch.zeros(b * v, 3, device=dev)
    leaves = (g.means, g.covariances, g.harmonics, g.opacities)

    def step():
        for t in leaves:
            t.grad = None
        ext = rearrange(cams["extrinsics"], "b v i j -> (b v) i j"); K = rearrange(cams["intrinsics"], "b v i j -> (b v) i j")
        near = rearrange(cams["near"], "b v -> (b v)"); far = rearrange(cams["far"], "b v -> (b v)")
        means = repeat(g.means, "b g xyz -> (b v) g xyz", v=v); covs = repeat(g.covariances, "b g i j -> (b v) g i j", v=v)
        sh = repeat(g.harmonics, "b g c d -> (b v) g c d", v=v); op = repeat(g.opacities, "b g -> (b v) g", v=v)
        scale = 1 / near
        ext = ext.clone(); ext[..., :3, 3] = ext[..., :3, 3] * scale[:, None]
        covs = covs * (scale[:, None, None, None] ** 2); means = means * scale[:, None, None]
        near, far = near * scale, far * scale
        degree = isqrt(sh.shape[-1]) - 1
        shs = rearrange(sh, "b g xyz n -> b g n xyz").contiguous()
        fov_x, fov_y = get_fov(K).unbind(dim=-1)
        tx, ty = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
        proj = get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)
        view = ext.inverse().transpose(1, 2)
        full = view @ proj
        images = []
        row, col = torch.triu_indices(3, 3)
        for i in range(b * v):
            m2d = torch.zeros_like(means[i], requires_grad=True)
            st = GaussianRasterizationSettings(image_height=h, image_width=w, tanfovx=tx[i].item(), tanfovy=ty[i].item(), bg=bg[i],
                                               scale_modifier=1.0, viewmatrix=view[i], projmatrix=full[i], projmatrix_raw=proj[i],
                                               sh_degree=degree, campos=ext[i, :3, 3], prefiltered=False, debug=False)
            image, radii, depth, opacity, n_touched = GaussianRasterizer(st)(
                means3D=means[i], means2D=m2d, shs=shs[i], colors_precomp=None, opacities=op[i, ..., None],
                cov3D_precomp=covs[i, :, row, col], theta=None, rho=None)
            images.append(image)
        color = rearrange(torch.stack(images), "(b v) c h w -> b v c h w", b=b, v=v)
        loss = ((color - target) ** 2).mean()
        loss.backward()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    return {"views_per_s": round(b * v / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": steps, "rasterizer_calls_per_step": b * v,
            "what": "reference call pattern through the drop-in module: Gaussians replicated per view, per-view GaussianRasterizer calls (5-tuple, n_touched "
                    "counted, depth gradient path live), 2 .item() syncs and a [:, row, col] covariance gather per view, torch MSE; vs `value` = the batched decoder API"}


# ------------------------------------------------------------------ train leg (M2, C3)
def train_leg(args, rank, world, dev, dist):
    """C3 NVS-pretrain step (2 ctx / 4 tgt views, MSE, every parameter trains) on the full-size encoder,
    random init (checkpoints absent), per-rank data (seed 1234 + rank, main_style.py:118), gradients all-reduced in
    64 MiB buckets over RCCL while the backward runs (styl3r_amd/ddp.py)."""
    import torch
    from styl3r_amd import dist_utils
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    from styl3r_amd.scenes import make_scene
    from styl3r_amd.train import TrainStep
    cpu = dev.type == "cpu"
    if not cpu:
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats(dev)      # peak_mem_GB below is this leg's, not the process's
    torch.manual_seed(0)   # identical replicas on every rank
    tiny = dict(enc_depth=1, dec_depth=2, enc_embed_dim=64, dec_embed_dim=32, enc_num_heads=2, dec_num_heads=2,
                pos_embed="RoPE100", img_size=(512, 512)) if args.train_tiny else None
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(stylized=False), trunk_params=tiny).to(dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    # a one-rank group (torch.distributed.run --nproc-per-node 1) still issues every collective: the RCCL path runs on one GPU
    forced = dist is not None and world == 1
    b, v_ctx, v_tgt, H = args.train_scenes, 2, 4, (32 if args.train_tiny else 256)
    g = torch.Generator(dev).manual_seed(1234 + rank)
    sc = make_scene(n_ctx=v_ctx, grid_hw=(8, 8), n_views=v_tgt, image_hw=(H, H), seed=1234 + rank)
    K = sc.intrinsics[:1].to(dev)
    ex = lambda t, *shape: t.to(dev)[None].expand(b, *shape).contiguous()
    batch = dict(
        context=dict(image=torch.rand(b, v_ctx, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=K.expand(b, v_ctx, 3, 3).contiguous()),
        target=dict(image=torch.rand(b, v_tgt, 3, H, H, device=dev, generator=g), extrinsics=ex(sc.extrinsics, -1, -1, -1),
                    intrinsics=ex(sc.intrinsics, -1, -1, -1), near=ex(sc.near, -1), far=ex(sc.far, -1)))
    sync = (lambda: None) if cpu else (lambda: torch.cuda.synchronize(dev))
    from styl3r_amd import rasterizer, vit_ops
    from styl3r_amd.scenes import recentre_output_heads_
    # set-up, outside the timed region: a random-init point head throws (nearly) every Gaussian outside every frustum -- the step would render
    # empty images and every gradient would be zero.  One calibration forward re-centres the five output convolutions
    # (depth 2..4 in front of context view 0, as tests/golden/make_e2e_fixtures.py does for the reference model); rank 0's result is broadcast
    if not cpu:
        recentre_output_heads_(enc, batch["context"], dict(image=batch["context"]["image"][:, 0]))
    step = TrainStep(enc, dec, dist=dist, force_collective=forced, warm_up_steps=2000)    # config/main.yaml:37 (LinearLR from lr / 2000): the first steps of a run do not throw the scene away
    # Arithmetic of the GEMM-shaped kernels in the headline train step (ARITH_NOTES): "f16x3" -- the mode that meets north_star's 1e-4 on the
    # Gaussians at the model's FULL depth (tests/test_e2e_parity.py[full-*], tests/golden/e2e_full.npz: 24 + 24 ViT-L blocks) at the MFMA
    # count of bf16x3.  bf16x3 sits inside the reference's own TF32 distance (croco.py:13 allow_tf32) on every quantity but puts the
    # covariances at 1.3e-4 of float64 at full depth, so it no longer carries the headline; it and bf16x6 are timed beside it.
    keep_mode, keep_attn = vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH
    head_mode = "bf16x6" if (cpu or args.train_tiny) else args.train_mode
    try:
        vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = head_mode        # Linear / conv / attention contractions: one mode
        for _ in range(args.train_warmup):
            step(batch)
        dt = dist_utils.timed_steps(lambda: step(batch), args.train_steps, sync, dist, dev)
    finally:
        vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH = keep_mode, keep_attn
    rendered = dict(rasterizer.LAST_STATS)
    if not cpu:      # the step must render a real scene: more (tile, Gaussian) pairs than Gaussians
        assert rendered["pairs"] > rendered["gaussians_per_scene"], f"train leg rendered (nearly) nothing: {rendered}"
    grad_bytes = sum(step.reducer.bucket_sizes_bytes())
    out = {"metric": "256x256 rendered views/sec, full C3 train step (encoder + rasterizer fwd+bwd, MSE, DP all-reduce, clip, AdamW)",
           "value": round(dist_utils.aggregate_throughput(b * v_tgt, args.train_steps, world, dt), 3), "unit": "views/s",
           "ms_per_step": round(1e3 * dt / args.train_steps, 2), "steps": args.train_steps, "warmup": args.train_warmup,
           "scenes_per_gpu": b, "ctx_views": v_ctx, "tgt_views": v_tgt, "gaussians_per_scene": v_ctx * H * H,
           "pairs_R": rendered.get("pairs"), "longest_tile_list": rendered.get("longest_tile_list"),
           "params": sum(p.numel() for p in enc.parameters()), "grad_bytes_all_reduced_per_step": grad_bytes if step.reducer.collective else 0,
           "grad_bytes": grad_bytes, "buckets": len(step.reducer.buckets), "bucket_MiB": 64,
           "collective": ("all_reduce(SUM) per bucket on the backend's stream, overlapped with the backward"
                          + (" (ONE-rank group: collectives issued, identity result)" if forced else "") if step.reducer.collective else "none (1 rank, no process group)"),
           "linear_arithmetic": head_mode, "dtype": "f32",
           "arithmetic_note": ARITH_NOTES[head_mode],
           "data": "synthetic, random-init weights",
           "encoder": "tiny test trunk" if args.train_tiny else "full size (ViT-L encoder x2, 2x12 ViT-B decoder blocks, 5 DPT heads)"}
    if not cpu:
        out["peak_mem_GB"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)      # (of THIS leg: the counter is reset where the leg starts)
        if not args.train_tiny:
            for other in [m for m in ("f16x3", "bf16x6", "bf16x3") if m != head_mode]:
                try:
                    vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = other
                    n_other = args.train_steps
                    for _ in range(3):          # (the first steps after a switch build the other mode's packed-weight buffers)
                        step(batch)
                    dt3 = dist_utils.timed_steps(lambda: step(batch), n_other, sync, dist, dev)
                    out[other] = {"ms_per_step": round(1e3 * dt3 / n_other, 2), "value": round(dist_utils.aggregate_throughput(b * v_tgt, n_other, world, dt3), 3),
                                  "unit": "views/s", "steps": n_other, "products_per_launch": vit_ops.load().vit_x6_products(),
                                  "note": "the same step in another arithmetic mode: " + ARITH_NOTES[other]}
                finally:
                    vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH = keep_mode, keep_attn
                    vit_ops._x6()
            out["linear_mfma"] = linear_roofline(dev, b * v_ctx * 257)
    # ---- the exchange, diagnosed in the same line: per-bucket collective time, overlap fraction, and the other
    #      data-parallel mode (all_reduce <-> rs_ag) as an automatic A/B.  Only when a process group exists (N > 1, or a one-rank launch
    #      under torch.distributed.run, whose RCCL group still issues every collective).
    if step.reducer.collective and not getattr(args, "no_comm_report", False):
        from styl3r_amd.ddp import broadcast_module_state, comm_report
        try:
            vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = head_mode
            run = lambda: step(batch)
            out["comm"] = comm_report(step.reducer, run, sync, steps=max(2, min(4, args.train_steps)), destructive=True)
            modes = {step.dp_mode: {"ms_per_step": out["ms_per_step"], "value": out["value"]}}
            # the other data-parallel mode as an A/B: automatic in a one-rank group (that path has run on hardware); at N > 1 only with --dp-ab --
            # "rs_ag" has never run on more than one GPU (gloo world-2 and one-rank RCCL only), and a collective mismatch there would not fail, it
            # would hang the whole scaling run this line is meant to explain
            if world > 1 and not getattr(args, "dp_ab", False):
                out["dp_modes"] = modes
                out["dp_mode"] = step.dp_mode
                out["dp_ab_note"] = "rerun with --dp-ab for the all_reduce / rs_ag A/B at N > 1 (opt-in: untested on more than one GPU)"
                broadcast_module_state(enc, dist, force_collective=forced)       # the no-collective steps of the report let the replicas drift
            else:
                other_dp = "rs_ag" if step.dp_mode == "all_reduce" else "all_reduce"
                step.reducer.close()
                del step
                if not cpu:
                    torch.cuda.empty_cache()
                broadcast_module_state(enc, dist, force_collective=forced)       # the no-collective steps let the replicas drift apart
                step = TrainStep(enc, dec, dist=dist, force_collective=forced, warm_up_steps=2000, dp_mode=other_dp)
                for _ in range(max(2, args.train_warmup)):
                    step(batch)
                n2 = args.train_steps
                dt2 = dist_utils.timed_steps(lambda: step(batch), n2, sync, dist, dev)
                modes[other_dp] = {"ms_per_step": round(1e3 * dt2 / n2, 2), "value": round(dist_utils.aggregate_throughput(b * v_tgt, n2, world, dt2), 3),
                                   "comm": comm_report(step.reducer, lambda: step(batch), sync, steps=max(2, min(4, args.train_steps)), destructive=True)}
                step.reducer.wait_params()
                out["dp_modes"] = modes
                out["dp_mode"] = [m for m in modes if m != other_dp][0]
        except Exception as e:
            out["comm_error"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH = keep_mode, keep_attn
            if not cpu:
                vit_ops._x6()
    return out


def stage_leg(args, rank, world, dev, dist, config):
    """The other two train configurations of SURVEY 8 on the full-size encoder, bf16x3 like the C3 headline (tools/bench_train.py runs the same steps):
    "c4" = style stage at the reference's per-GPU batch: 6 scenes, 4 ctx / 6 tgt views 256 x 256, VGG style loss + identity pass (two encoder + decoder
    passes per step), backbone frozen (model_wrapper_style.py:118-232; random-init VGG: no weights here);  "c5" = stress shapes: 1 scene, 4 ctx views
    512 x 512 -> 1 048 576 Gaussians, sh_degree 4, 4 tgt views 512 x 512, MSE, every parameter trains."""
    import torch
    from styl3r_amd import dist_utils, vit_ops
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg, GaussianAdapterCfg
    from styl3r_amd.scenes import make_scene
    from styl3r_amd.train import TrainStep
    c4 = config == "c4"
    if dev.type != "cpu":
        torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats(dev)      # per-leg peak memory
    torch.manual_seed(0)
    cfg = EncoderNoPoSplatTokenStyleCfg(stylized=c4)
    if not c4:
        cfg.gaussian_adapter = GaussianAdapterCfg(cfg.gaussian_adapter.gaussian_scale_min, cfg.gaussian_adapter.gaussian_scale_max, 4)
    enc = EncoderNoPoSplatMultiTokenStyle(cfg).to(dev)
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    forced = dist is not None and world == 1
    b, v_ctx, v_tgt, H = (6, 4, 6, 256) if c4 else (1, 4, 4, 512)
    g = torch.Generator(dev).manual_seed(1234 + rank)
    sc = make_scene(n_ctx=v_ctx, grid_hw=(8, 8), n_views=v_tgt, image_hw=(H, H), seed=1234 + rank)
    ex = lambda t, *shape: t.to(dev)[None].expand(b, *shape).contiguous()
    batch = dict(
        context=dict(image=torch.rand(b, v_ctx, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(b, v_ctx, 3, 3).contiguous()),
        target=dict(image=torch.rand(b, v_tgt, 3, H, H, device=dev, generator=g), extrinsics=ex(sc.extrinsics, -1, -1, -1),
                    intrinsics=ex(sc.intrinsics, -1, -1, -1), near=ex(sc.near, -1), far=ex(sc.far, -1)))
    if c4:
        batch["style"] = dict(image=torch.rand(b, 3, H, H, device=dev, generator=g))
    from styl3r_amd import rasterizer
    from styl3r_amd.scenes import recentre_output_heads_
    # set-up (untimed): re-centre the random-init output heads so that the step renders a real scene (see train_leg)
    recentre_output_heads_(enc, batch["context"], dict(image=(batch["style"]["image"] - 0.5) / 0.5) if c4 else dict(image=batch["context"]["image"][:, 0]))
    if c4:
        from styl3r_amd.losses import IdentityLoss, LossStyle, VGGEncoder
        vgg = VGGEncoder().to(dev)
        step = TrainStep(enc, dec, dist=dist, losses=[LossStyle(vgg=vgg)], identity_loss=IdentityLoss(vgg=vgg), force_collective=forced, warm_up_steps=2000)
    else:
        step = TrainStep(enc, dec, dist=dist, force_collective=forced, warm_up_steps=2000)
    keep = vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH
    steps = 10
    try:
        vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = args.train_mode
        for _ in range(3):
            step(batch)
        dt = dist_utils.timed_steps(lambda: step(batch), steps, lambda: torch.cuda.synchronize(dev), dist, dev)
    finally:
        vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH = keep
        vit_ops._x6()
    rendered = dict(rasterizer.LAST_STATS)
    assert rendered["pairs"] > rendered["gaussians_per_scene"], f"{config} leg rendered (nearly) nothing: {rendered}"
    out = {"metric": f"{H}x{H} rendered views/sec, full train step, " + ("C4 style stage (VGG style loss + identity pass, backbone frozen)" if c4 else
                                                                       "C5 stress shapes (1 048 576 Gaussians/scene, sh_degree 4)"),
           "value": round(dist_utils.aggregate_throughput(b * v_tgt, steps, world, dt), 3), "unit": "views/s", "ms_per_step": round(1e3 * dt / steps, 2),
           "steps": steps, "warmup": 3, "scenes_per_gpu": b, "ctx_views": v_ctx, "tgt_views": v_tgt, "gaussians_per_scene": v_ctx * H * H,
           "pairs_R": rendered.get("pairs"), "longest_tile_list": rendered.get("longest_tile_list"),
           "trainable_params": sum(p.numel() for p in enc.parameters() if p.requires_grad), "linear_arithmetic": args.train_mode, "dtype": "f32",
           "peak_mem_GB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1), "n_gpus": world, "data": "synthetic, random-init weights (VGG included)"}
    del step, enc, dec, batch
    torch.cuda.empty_cache()
    return out


def infer_leg(args, dev):
    """C2 (BASELINE.json configs[1]): inference latency, 2 context views 256 x 256 -> 131 072 Gaussians on the full-size encoder
    (random init: re10k_2v.ckpt is absent), 3 target views rendered forward only, `no_grad`, bf16x6 arithmetic (the mode of the 1e-4 RGB
    statement), heads and style branch on side streams (infer_model_re10k.py:262-560 call order)."""
    import torch
    from styl3r_amd import vit_ops
    from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
    from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
    from styl3r_amd.scenes import make_scene
    torch.manual_seed(0)
    with torch.device(dev):
        enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg()).eval()
    enc.head_streams = True
    dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
    H, v_ctx, v_tgt = 256, 2, 3
    sc = make_scene(n_ctx=v_ctx, grid_hw=(8, 8), n_views=v_tgt, image_hw=(H, H), seed=1234)
    g = torch.Generator(dev).manual_seed(1234)
    ctx = dict(image=torch.rand(1, v_ctx, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(1, v_ctx, 3, 3).contiguous())
    style = dict(image=ctx["image"][:, 0])
    ex = lambda t: t.to(dev)[None].contiguous()
    cams = [ex(sc.extrinsics), ex(sc.intrinsics), ex(sc.near), ex(sc.far)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    steps, warm = 20, 4

    def timed(mode):
        keep = vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH
        vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = mode
        t_enc = t_ras = 0.0
        try:
            with torch.no_grad():
                for i in range(warm + steps):
                    ev[0].record()
                    gs = enc(ctx, style, 0)
                    ev[1].record()
                    dec.forward(gs, *cams, (H, H))
                    ev[2].record()
                    torch.cuda.synchronize(dev)
                    if i >= warm:
                        t_enc += ev[0].elapsed_time(ev[1]); t_ras += ev[1].elapsed_time(ev[2])
        finally:
            vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH = keep
        rec = {"encoder_ms": round(t_enc / steps, 3), "rasterizer_ms": round(t_ras / steps, 3), "eager_total_ms": round((t_enc + t_ras) / steps, 3),
               "gaussians": int(gs.means.shape[1])}
        # the same forward as one hipGraph per stream segment (styl3r_amd.graphs.StreamGraphedEncoder): no host launch cost, the GPU-side limit
        try:
            keep = vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH
            vit_ops.LINEAR_MODE = vit_ops.ATTENTION_ARITH = mode
            from styl3r_amd.graphs import StreamGraphedEncoder
            genc = StreamGraphedEncoder(enc, ctx, style)
            t_g = 0.0
            with torch.no_grad():
                for i in range(warm + steps):
                    ev[0].record()
                    gs = genc(ctx, style)
                    dec.forward(gs, *cams, (H, H))
                    ev[2].record()
                    torch.cuda.synchronize(dev)
                    if i >= warm:
                        t_g += ev[0].elapsed_time(ev[2])
            rec["stream_graphs_total_ms"] = round(t_g / steps, 3)
            del genc
        except Exception as e:
            rec["stream_graphs_total_ms"] = f"{type(e).__name__}: {e}"[:200]
        finally:
            vit_ops.LINEAR_MODE, vit_ops.ATTENTION_ARITH = keep
            vit_ops._x6()
        # total_ms: the serving form (stream graphs) when it captured, else the eager launch path
        sg = rec["stream_graphs_total_ms"]
        rec["total_ms"] = sg if isinstance(sg, float) else rec["eager_total_ms"]
        rec["serving_form"] = "stream graphs" if isinstance(sg, float) else "eager"
        rec["views_per_s"] = round(v_tgt * 1e3 / rec["total_ms"], 2)
        return rec

    # top level: f16x3, the arithmetic of the train leg's headline (fp32-class accuracy: tests/test_gpu_vit.py measures it at or below bf16x6's error
    # against float64) ; bf16x6 = six products, the mode rounds 1 - 5 quoted here; bf16x3 = the TF32-class mode (tests/test_e2e_parity.py bounds all three)
    out = {"metric": "C2 inference latency, 2 ctx + 3 tgt views 256x256, forward only, batch 1", **timed("f16x3"), "steps": steps,
           "linear_arithmetic": "f16x3", "bf16x6": timed("bf16x6"), "bf16x3": timed("bf16x3"),
           "encoder_launch": "total_ms / stream_graphs_total_ms: one hipGraph per stream segment (style branch and the five heads on side streams, the dual decoders "
                             "as two-problem launches on the main stream); eager_total_ms: the same forward launched kernel by kernel (host-bound: ~1 400 launches)",
           "dtype": "f32", "data": "synthetic, random-init weights"}
    # `test.align_pose` (config/main.yaml:57-60, model_wrapper_style.py:391-447): before the evaluation render the reference optimises the 3 target
    # poses for pose_align_steps = 100 Adam steps, each one a rasterizer forward + backward with pose gradients (theta / rho) on the 3 views.
    # Timed on a scene the heads were re-centred for (a random-init encoder renders nothing), from poses perturbed by ~1 degree / 1 % of the baseline.
    try:
        from styl3r_amd.pose_align import align_poses
        from styl3r_amd.scenes import recentre_output_heads_
        recentre_output_heads_(enc, ctx, style)
        with torch.no_grad():
            gs = enc(ctx, style, 0)
            ref_img = dec.forward(gs, *cams, (H, H)).color
        gq = torch.Generator(dev).manual_seed(7)
        pert = cams[0].clone()
        pert[..., :3, 3] += 0.01 * torch.randn(1, v_tgt, 3, device=dev, generator=gq)
        align_poses(dec, gs, ref_img, pert, cams[1], cams[2], cams[3], steps=5)            # warm-up
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        _, hist = align_poses(dec, gs, ref_img, pert, cams[1], cams[2], cams[3], steps=100)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out["align_pose"] = {"steps": 100, "views": v_tgt, "total_ms": round(1e3 * dt, 1), "ms_per_step": round(10.0 * dt, 3),
                             "loss_first": hist[0], "loss_last": hist[-1],
                             "note": "100 Adam steps x (rasterizer forward + backward with theta / rho gradients on 3 views, 131 072 Gaussians); host loop as in the reference (one loss read-back per step)"}
    except Exception as e:
        out["align_pose"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del enc, dec
    torch.cuda.empty_cache()
    return out


def linear_roofline(dev, M):
    """MFMA roofline of the step's dominant GEMM-shaped kernel, measured live: the encoder's qkv Linear (M tokens x 3072 x 1024) on the
    kernels the step runs, in both arithmetic modes.  Nominal peak-equivalent = dense bf16 MFMA peak (2.5 PF) / MFMAs per fp32 product."""
    import torch
    from styl3r_amd import vit_ops
    N, K = 3072, 1024
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; bias = torch.randn(N, device=dev)
    res = {}
    keep = vit_ops.LINEAR_MODE
    try:
        for mode, nprod in (("bf16x6", 6), ("bf16x3", 3), ("f16x3", 3)):
            vit_ops.LINEAR_MODE = mode
            before = vit_ops.CALLS["linear_x6r"]
            with torch.no_grad():
                xg = x.requires_grad_(False)
                for _ in range(10):
                    vit_ops._FusedLinear.apply(xg, w, bias, None, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    vit_ops._FusedLinear.apply(xg, w, bias, None, 0)
                e1.record(); torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / 50
            tf = 2.0 * M * N * K / ms / 1e9
            ring = vit_ops.CALLS["linear_x6r"] > before
            res[mode] = {"kernel": ("vit::x6r::k_linear_x6c (256 x 256 tiles, ping-pong wave pairs, LDS-DMA ring)" if ring else "vit::x6::k_linear_x6" + (" + vit::x6::k_amax" if mode == "f16x3" else "")),
                         "ms": round(ms, 4), "achieved": round(tf, 1), "unit": "TFLOP/s (fp32-accurate product rate)", "bound": "mfma",
                         "peak": round(2500.0 / nprod, 1), "frac": round(tf / (2500.0 / nprod), 3), "mfma_TFLOPs_bf16": round(nprod * tf, 1)}
    finally:
        vit_ops.LINEAR_MODE = keep
        vit_ops._x6()
    head = res["bf16x6"]
    return {"kernel": head["kernel"] + " (encoder qkv Linear)", "shape_MNK": [M, N, K], "ms": head["ms"], "achieved": head["achieved"], "unit": head["unit"],
            "bound": "mfma", "peak": head["peak"], "frac": head["frac"], "mfma_TFLOPs_bf16": head["mfma_TFLOPs_bf16"], "bf16x3": res["bf16x3"], "f16x3": res["f16x3"],
            "note": "power-limited on random operands: DESIGN.md 9.2"}


# ------------------------------------------------------------------ dry run (CPU launch-path test)
def dry_leg(args, rank, world, dist):
    """--dry-cpu: no kernels, no oracle.  Exercises the launch path only: rank -> shard mapping, barrier-bracketed timing,
    MAX over ranks, whole-job aggregation, one JSON line from rank 0."""
    import torch
    from styl3r_amd import dist_utils
    seeds = dist_utils.scene_seeds(rank, args.scenes)
    calls = []
    dt = dist_utils.timed_steps(lambda: calls.append(1), args.steps, lambda: None, dist)
    # all ranks' first seed, gathered: shows the shards are disjoint
    mine = torch.tensor([seeds[0]], dtype=torch.int64)
    allv = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    if dist is not None:
        dist.all_gather(allv, mine)
    else:
        allv = [mine]
    V = args.scenes * args.views
    return {"metric": "256x256 stylized views/sec (fwd+bwd) @ ~65k Gaussians", "value": round(dist_utils.aggregate_throughput(V, args.steps, world, max(dt, 1e-9)), 2),
            "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "DRY RUN on CPU: launch-path test, no kernels executed -- not a measurement", "dry_cpu": True,
            "config": {"workload": "none (dry run)", "views_per_step_per_gpu": V, "parallelism": f"dp{world} (scenes sharded)",
                       "first_scene_seed_per_rank": [int(t.item()) for t in allv], "timed_calls_rank0": len(calls)}}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    import torch
    from styl3r_amd import dist_utils
    rank, local_rank, world = dist_utils.env_world()
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: using the launcher's world size", file=sys.stderr)
    if args.dry_cpu:
        dev, backend = torch.device("cpu"), "gloo"
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (use --dry-cpu for the CPU launch-path test)"
        assert local_rank < torch.cuda.device_count(), f"rank {rank}: local rank {local_rank} but {torch.cuda.device_count()} GPUs visible"
        torch.cuda.set_device(local_rank)
        dev, backend = torch.device("cuda", local_rank), "nccl"   # "nccl" is RCCL on ROCm
    # under torch.distributed.run a one-rank launch still forms its (one-rank) RCCL group, so `--gpus 1` exercises the collective path
    dist = dist_utils.init_distributed(backend, dev if backend == "nccl" else None,
                                       single_rank_group=dist_utils.launched_by_torchrun() and not args.dry_cpu)

    if args.dry_cpu:
        res, scenes = dry_leg(args, rank, world, dist), None
    else:
        res, scenes = raster_leg(args, rank, world, dev, dist)
    res["launch"] = {"backend": ("rccl (torch 'nccl')" if backend == "nccl" else backend) if dist is not None else "none (single process)",
                     "ranks": dist.get_world_size() if dist is not None else 1,
                     "spawned_by": "bench.py self-spawn -> torch.distributed.run" if os.environ.get("STYL3R_BENCH_SPAWNED") else
                                   ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "direct"),
                     "device": (torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu")}
    if not args.no_train_leg and not args.dry_cpu:
        try:
            res["train_step"] = train_leg(args, rank, world, dev, dist)
            res["train_step"]["n_gpus"] = world
        except Exception as e:   # the headline line must survive a failure of the secondary leg; it is reported, not hidden
            res["train_step"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if not args.no_stage_legs and not args.no_train_leg and not args.dry_cpu and not args.train_tiny:
        import gc
        for key, config in (("style_stage_step", "c4"), ("stress_512_step", "c5")):
            gc.collect(); torch.cuda.empty_cache()
            try:
                res[key] = stage_leg(args, rank, world, dev, dist, config)
            except Exception as e:
                res[key] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if rank == 0 and not args.no_infer_leg and not args.dry_cpu:
        try:
            res["infer"] = infer_leg(args, dev)
        except Exception as e:
            res["infer"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if rank == 0:
        if not args.no_cpu_baseline and not args.dry_cpu:
            res["cpu_baseline"] = cpu_baseline(args, scenes)
            if not args.no_train_leg:
                res["cpu_baseline"]["encoder"] = cpu_encoder_baseline(args)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
