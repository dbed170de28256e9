"""world_size-2 gloo test (CPU) of the N>1 path of bench.py: disjoint scene shards per rank,
barrier-bracketed timing, MAX over ranks, whole-job throughput."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    from styl3r_amd import dist_utils
    from styl3r_amd.scenes import make_scene
    rank, local_rank, world = dist_utils.env_world()
    dist = dist_utils.init_distributed("gloo")
    assert dist is not None and dist.get_world_size() == 2
    seeds = dist_utils.scene_seeds(rank, 3)
    scenes = [make_scene(1, (8, 8), 2, (16, 16), seed=s) for s in seeds]
    # every rank gets different scenes: exchange a checksum of the first scene
    chk = torch.tensor([float(scenes[0].means.sum())], dtype=torch.float64)
    allc = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allc, chk)
    calls = []
    def step():
        calls.append(1)
        time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
    dt = dist_utils.timed_steps(step, 4, lambda: None, dist)
    thr = dist_utils.aggregate_throughput(6, 4, world, dt)
    print(json.dumps(dict(rank=rank, seeds=seeds, n=len(calls), dt=dt, thr=thr, chk=[float(c) for c in allc])), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % str(ROOT)


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    a, b = sorted(outs, key=lambda d: d["rank"])
    assert a["seeds"] == [1234, 1235, 1236] and b["seeds"] == [2234, 2235, 2236]
    assert set(a["seeds"]).isdisjoint(b["seeds"])
    assert a["chk"] == b["chk"] and a["chk"][0] != a["chk"][1]       # different scenes on the two ranks
    assert a["n"] == b["n"] == 4                                       # exactly K timed steps each
    assert abs(a["dt"] - b["dt"]) < 1e-9 and a["dt"] >= 4 * 0.1 - 1e-3  # MAX over ranks = the slow rank
    assert abs(a["thr"] - 6 * 2 * 4 / a["dt"]) < 1e-9                 # whole-job aggregate


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no torch.distributed.run environment must start 2 ranks itself (the driver's command
    form; round-1 VERDICT: the flag was parsed and ignored).  --dry-cpu = launch path only: gloo, no kernels, no oracle."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-cpu", "--steps", "4", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                   # ONE line, from rank 0
    d = __import__("json").loads(lines[0])
    assert d["n_gpus"] == 2 and d["launch"]["ranks"] == 2 and d["launch"]["backend"] == "gloo"
    assert d["launch"]["spawned_by"].startswith("bench.py self-spawn")
    assert d["config"]["first_scene_seed_per_rank"] == [1234, 2234]    # disjoint scene shards
    assert d["config"]["timed_calls_rank0"] == 4 and d["steps"] == 4
    assert d["dry_cpu"] is True and "DRY RUN" in d["data"]            # can never be mistaken for a measurement
    assert d["scaling"] == "weak" and d["config"]["parallelism"].startswith("dp2")


def test_bench_eight_rank_launch_path():
    """the launch the driver uses on the 8-GPU node -- `python -m torch.distributed.run --nproc-per-node 8 bench.py
    --gpus 8` -- in --dry-cpu mode (gloo, no kernels): eight ranks rendezvous on 127.0.0.1, shard the scenes disjointly, barrier,
    and rank 0 alone prints the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", "29571", str(ROOT / "bench.py"), "--gpus", "8", "--dry-cpu", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = __import__("json").loads(lines[0])
    assert d["n_gpus"] == 8 and d["launch"]["ranks"] == 8 and d["launch"]["backend"] == "gloo"
    assert d["config"]["first_scene_seed_per_rank"] == [1234 + 1000 * r for r in range(8)]
    assert d["config"]["parallelism"].startswith("dp8") and d["dry_cpu"] is True and d["steps"] == 3


def test_bench_single_process_line_unchanged_keys():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--dry-cpu", "--steps", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert p.returncode == 0, p.stderr[-2000:]
    d = __import__("json").loads(p.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d
    assert d["n_gpus"] == 1 and d["launch"]["ranks"] == 1
