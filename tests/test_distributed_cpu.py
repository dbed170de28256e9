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
