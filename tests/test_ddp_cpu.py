"""world_size-2 gloo test of the bucketed gradient reducer (the N>1 exchange step of the train path)."""
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import torch
    from torch import nn
    from styl3r_amd import dist_utils
    from styl3r_amd.ddp import BucketedGradReducer
    rank, _, world = dist_utils.env_world()
    dist = dist_utils.init_distributed("gloo")
    torch.manual_seed(0)                                   # identical replicas
    model = nn.Sequential(nn.Linear(16, 64), nn.GELU(), nn.Linear(64, 64), nn.GELU(), nn.Linear(64, 8))
    unused = nn.Parameter(torch.zeros(5))                  # like mask_token: never receives a gradient
    params = list(model.parameters()) + [unused]
    red = BucketedGradReducer(params, dist, bucket_bytes=8 * 1024)      # several small buckets
    assert len(red.buckets) >= 3
    torch.manual_seed(100 + rank)                          # different data per rank
    ok = True
    for step in range(2):
        x = torch.randn(32, 16)
        # local gradient without the reducer, for the check
        for p in params: p.grad = None
        model(x).pow(2).mean().backward()
        local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
        gathered = [[torch.zeros_like(t) for _ in range(world)] for t in local]
        for t, gl in zip(local, gathered): dist.all_gather(gl, t)
        want = [sum(gl) / world for gl in gathered]
        red.prepare()
        model(x).pow(2).mean().backward()
        red.finish()
        for p, w in zip(params[:-1], want[:-1]):
            ok &= bool(torch.allclose(p.grad, w, atol=1e-7))
        ok &= all(p.grad.data_ptr() != 0 for p in params[:-1])
        ok &= unused.grad is None              # unused on every rank -> None, as DDP(find_unused_parameters=True)
    print(json.dumps(dict(rank=rank, ok=ok, buckets=red.bucket_sizes_bytes())), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % str(ROOT)


def test_bucketed_reducer_two_ranks(tmp_path):
    script = tmp_path / "w.py"; script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        d = json.loads(o.strip().splitlines()[-1])
        assert d["ok"], d
        assert sum(d["buckets"]) == 4 * (16 * 64 + 64 + 64 * 64 + 64 + 64 * 8 + 8 + 5)


SYNC_WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import torch
    from torch import nn
    from styl3r_amd import dist_utils
    from styl3r_amd.ddp import broadcast_module_state
    rank, _, world = dist_utils.env_world()
    dist = dist_utils.init_distributed("gloo")
    torch.manual_seed(1000 + rank)                         # DIFFERENT replicas, as after a per-rank seed
    model = nn.Sequential(nn.Linear(16, 300), nn.BatchNorm1d(300), nn.Linear(300, 7))
    model[1].running_mean.fill_(float(rank)); model[1].num_batches_tracked.fill_(5 + rank)   # buffers too, incl. an int64 one
    before = float(model[0].weight.sum())
    sent = broadcast_module_state(model, dist, src=0, chunk_bytes=4096)       # several chunks
    flat = torch.cat([t.double().reshape(-1) for t in list(model.parameters()) + list(model.buffers())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    print(json.dumps(dict(rank=rank, same=same, changed=(float(model[0].weight.sum()) != before), sent=sent,
                          tracked=int(model[1].num_batches_tracked))), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % str(ROOT)


def test_initial_parameter_broadcast_makes_the_replicas_identical(tmp_path):
    """DDP's module-state sync (rank 0 wins, parameters and buffers of every dtype) -- TrainStep does it before building the optimizer"""
    script = tmp_path / "s.py"; script.write_text(SYNC_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29546", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    assert all(d["same"] for d in outs), outs
    assert [d["changed"] for d in sorted(outs, key=lambda d: d["rank"])] == [False, True]       # rank 0 keeps its state, rank 1 adopts it
    assert all(d["tracked"] == 5 for d in outs) and outs[0]["sent"] == outs[1]["sent"] > 0
    from styl3r_amd.ddp import broadcast_module_state
    import torch
    assert broadcast_module_state(torch.nn.Linear(2, 2), None) == 0                          # single process: nothing to do


def test_reducer_single_process_is_identity():
    import torch
    from torch import nn
    from styl3r_amd.ddp import BucketedGradReducer
    m = nn.Linear(4, 3)
    red = BucketedGradReducer(m.parameters(), None)
    red.prepare(); m(torch.ones(2, 4)).sum().backward(); red.finish()
    assert torch.allclose(m.weight.grad, torch.full((3, 4), 2.0))


def test_flat_clip_matches_torch_clip():
    import torch
    from torch import nn
    from styl3r_amd.ddp import BucketedGradReducer
    torch.manual_seed(0)
    m = nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 4))
    x = torch.randn(5, 8)
    m(x).pow(2).sum().backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
    want = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters():
        p.grad = None
    red = BucketedGradReducer(m.parameters(), None, bucket_bytes=256)
    red.prepare(); m(x).pow(2).sum().backward(); red.finish()
    total = red.clip_grad_norm_(0.5)
    assert total > 0.5
    for p, w in zip(m.parameters(), want):
        assert torch.allclose(p.grad, w, rtol=1e-6, atol=1e-8)


def test_second_backward_inside_prepare_finish_is_refused():
    import pytest
    import torch
    from torch import nn
    from styl3r_amd.ddp import BucketedGradReducer
    m = nn.Linear(4, 3)
    red = BucketedGradReducer(m.parameters(), None)
    red.prepare()
    m(torch.ones(2, 4)).sum().backward()
    with pytest.raises(RuntimeError, match="ONE backward"):
        m(torch.ones(2, 4)).sum().backward()
    red.finish()


def test_unused_parameter_keeps_grad_none_and_adamw_skips_it():
    import torch
    from torch import nn
    from styl3r_amd.ddp import BucketedGradReducer
    m = nn.Linear(4, 3)
    dead = nn.Parameter(torch.ones(5))
    opt = torch.optim.AdamW([*m.parameters(), dead], lr=0.1, weight_decay=0.5)
    red = BucketedGradReducer([*m.parameters(), dead], None)
    for _ in range(2):
        red.prepare(); m(torch.ones(2, 4)).sum().backward(); red.finish()
        assert dead.grad is None and m.weight.grad is not None
        opt.step()
    assert torch.equal(dead.detach(), torch.ones(5))       # no weight decay applied to a parameter without gradient


CHANGING_WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import torch
    from torch import nn
    from styl3r_amd import dist_utils
    from styl3r_amd.ddp import BucketedGradReducer
    rank, _, world = dist_utils.env_world()
    dist = dist_utils.init_distributed("gloo")
    torch.manual_seed(0)
    trunk, branch = nn.Linear(8, 8), nn.Linear(8, 8)       # `branch` only gets a gradient where the batch takes it
    params = list(trunk.parameters()) + list(branch.parameters())
    red = BucketedGradReducer(params, dist, bucket_bytes=64)
    x = torch.ones(4, 8)
    seen = []
    # step 0: no rank uses the branch; step 1: ONLY rank 1 does (its local arrival set changes, rank 0's does not);
    # step 2: nobody again.  The r02 reducer entered its map all-reduce on rank 1 alone at step 1 (ADVICE r2: hang / mispaired collective)
    for step in range(3):
        red.prepare()
        y = trunk(x)
        if step == 1 and rank == 1:
            y = y + branch(x)
        y.sum().backward()
        red.finish()
        seen.append(branch.weight.grad is not None)
        if step == 1:
            want = torch.full((8, 8), 4.0) / world           # rank 1's gradient (sum over 4 rows of ones), averaged over the ranks
            ok = bool(torch.allclose(branch.weight.grad, want))
    print(json.dumps(dict(rank=rank, seen=seen, ok=ok)), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % str(ROOT)


def test_used_map_exchange_is_symmetric_when_one_ranks_arrival_set_changes(tmp_path):
    """a collective must not be gated on rank-local state"""
    script = tmp_path / "c.py"; script.write_text(CHANGING_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29548", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        d = json.loads(o.strip().splitlines()[-1])
        assert d["seen"] == [False, True, False] and d["ok"], d       # used on ANY rank -> a gradient on EVERY rank, that step only


def test_never_used_parameter_does_not_hold_back_the_bucket_launches():
    """`mask_token` never receives a gradient and sits, in reverse registration order, in a bucket AHEAD of the heads
    and the whole backbone.  With in-order launches a bucket that waits for it -- and every later one -- would only be reduced in finish(),
    with no overlap.  From the second step on (once the agreed map knows it) every bucket must already be out when the backward ends;
    and when the parameter does get a gradient later, it arrives through the straggler path with the right value."""
    import torch
    from torch import nn
    from styl3r_amd.ddp import BucketedGradReducer
    torch.manual_seed(0)
    a, b = nn.Linear(6, 6), nn.Linear(6, 6)
    dead = nn.Parameter(torch.ones(6))
    params = [*a.parameters(), dead, *b.parameters()]               # reverse order: b | dead | a  -> `dead` is ahead of `a`
    red = BucketedGradReducer(params, None, bucket_bytes=32)
    x = torch.randn(3, 6)
    launched_before_finish = []
    for step in range(4):
        red.prepare()
        y = b(a(x))
        if step == 3:
            y = y + dead                                            # a data-dependent branch wakes the parameter up
        y.sum().backward()
        launched_before_finish.append(sum(bk["launched"] for bk in red.buckets) / len(red.buckets))
        red.finish()
        ref = torch.autograd.grad(b(a(x)).sum(), list(a.parameters()) + list(b.parameters()))
        for p, r in zip(list(a.parameters()) + list(b.parameters()), ref):
            assert torch.allclose(p.grad, r, atol=1e-6)
        if step == 3:
            assert torch.allclose(dead.grad, torch.full((6,), 3.0))
        else:
            assert dead.grad is None
    # step 0: nothing is known yet, the bucket of `dead` (and everything behind it) waits for finish(); from step 1 on only the
    # bucket holding nothing but `dead` is still pending at the end of the backward (nobody arrives to trigger it)
    assert launched_before_finish[0] < 0.5 and all(f >= (len(red.buckets) - 1) / len(red.buckets) for f in launched_before_finish[1:3]), launched_before_finish


RSAG_WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import torch
    from torch import nn
    from styl3r_amd import dist_utils
    from styl3r_amd.ddp import BucketedGradReducer
    from styl3r_amd.train import make_optimizer
    rank, _, world = dist_utils.env_world()
    dist = dist_utils.init_distributed("gloo")

    def run(mode):
        torch.manual_seed(0)                                   # identical replicas
        model = nn.Sequential(nn.Linear(16, 67), nn.GELU(), nn.Linear(67, 64), nn.GELU(), nn.Linear(64, 9))
        unused = nn.Parameter(torch.ones(5))                   # never receives a gradient: no weight decay, no moments, on any rank
        new, pre = list(model[4].parameters()) + [unused], list(model[0].parameters()) + list(model[2].parameters())
        params = list(model.parameters()) + [unused]
        red = BucketedGradReducer(params, dist, bucket_bytes=6 * 1024, mode=mode, groups=[new, pre])
        opt = make_optimizer(new, pre, lr=1e-2, owner=red if mode == "rs_ag" else None)
        torch.manual_seed(100 + rank)                          # different data per rank
        norms = []
        for step in range(4):
            x = torch.randn(32, 16)
            red.wait_params()
            red.prepare()
            model(x).pow(2).mean().backward()
            red.finish()
            norms.append(float(red.clip_grad_norm_(0.05, defer_to=None if mode == "all_reduce" else opt)))
            opt.step()
            red.gather_params()
        # ADVICE r04: (i) a reader between step() and the fence: the guarded module's forward waits for the all-gather and the
        # versions move when it lands; (ii) the moments of the other rank's ranges are gathered before a checkpoint
        v0 = [p._version for p in params]
        red.guard_readers(model)
        pending = len(red._gathers)
        model(torch.randn(2, 16))
        fenced = len(red._gathers) == 0 and (pending == 0 or all(p._version > v for p, v in zip(params[:-1], v0[:-1])))
        red.wait_params()
        refused = False
        if mode == "rs_ag":
            try:
                opt.state_dict()
            except RuntimeError:
                refused = True
            red.consolidate_optimizer_state(opt)
        moments = [opt.state[p][k].detach().clone() for p in params if opt.state.get(p) for k in ("exp_avg", "exp_avg_sq")]
        opt.state_dict()
        return [p.detach().clone() for p in params], norms, red, opt, dict(fenced=fenced, refused=refused, pending=pending), moments

    pa, na, ra, oa, xa, ma = run("all_reduce")
    pb, nb, rb, ob, xb, mb = run("rs_ag")
    mom_err = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(ma, mb))
    err = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(pa, pb))
    flat = torch.cat([p.reshape(-1) for p in pb]).double()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    owned = sum((hi - lo) for lo, hi in rb._range.values())
    total = sum(b["n"] for b in rb.buckets)
    print(json.dumps(dict(rank=rank, err=err, norm_err=max(abs(x - y) / x for x, y in zip(na, nb)), same=all(torch.equal(gathered[0], g) for g in gathered),
                          owned=owned, total=total, unused_untouched=bool(torch.equal(pb[-1], torch.ones(5))), nbuckets=len(rb.buckets),
                          kinds=[type(oa).__name__, type(ob).__name__], mom_err=mom_err, n_moments=len(mb), guard=xb,
                          bucket_n=[b["n"] for b in rb.buckets])), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % str(ROOT)


def test_reduce_scatter_all_gather_mode_trains_like_the_all_reduce_mode(tmp_path):
    """SURVEY 8e / VERDICT r03 #8: mode "rs_ag" (reduce-scatter per bucket, clip norm from the owned shards + one scalar all-reduce,
    AdamW on the owned element ranges only, all-gather of the updated parameters) must leave every rank with the parameters the
    all-reduce mode produces (<= 1e-6 after 4 clipped steps, two learning-rate groups, an unused parameter, odd-sized tensors so the
    buckets need padding), identical on both ranks, each rank owning about half of the elements."""
    script = tmp_path / "rsag.py"; script.write_text(RSAG_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29552", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-3000:]
        d = json.loads(o.strip().splitlines()[-1])
        assert d["err"] <= 1e-6 and d["norm_err"] <= 1e-6 and d["same"] and d["unused_untouched"], d
        assert d["nbuckets"] >= 3 and abs(d["owned"] - d["total"] / 2) <= d["nbuckets"], d
        assert d["kinds"][1] == "ShardedAdamWTorch", d
        # the consolidated moments equal the all-reduce mode's on every rank; state_dict() refused before the consolidation;
        # the guarded forward fenced a gather that was really in flight
        assert d["mom_err"] <= 1e-6 and d["n_moments"] >= 10 and d["guard"]["refused"] and d["guard"]["fenced"] and d["guard"]["pending"] > 0, d


def test_reduce_scatter_all_gather_mode_at_world_4_with_buckets_that_need_padding(tmp_path):
    """the same worker on FOUR ranks.  Its tensors have 67 x 16, 67, 64 x 67, 64, 9 x 64, 9 and 5 elements, so no bucket
    length is a multiple of 4: every bucket is padded to the world size, shards straddle parameter boundaries, and a rank may own
    nothing of a small tensor.  Same bars as the two-rank test; every rank owns about a quarter of the elements."""
    script = tmp_path / "rsag4.py"; script.write_text(RSAG_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29556", WORLD_SIZE="4", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(4)]
    owned = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-3000:]
        d = json.loads(o.strip().splitlines()[-1])
        assert d["err"] <= 1e-6 and d["norm_err"] <= 1e-6 and d["same"] and d["unused_untouched"], d
        assert d["nbuckets"] >= 3 and abs(d["owned"] - d["total"] / 4) <= d["nbuckets"], d
        assert any(b % 4 for b in d["bucket_n"]), d            # the point of the test: lengths that are not multiples of the world size
        assert d["mom_err"] <= 1e-6 and d["guard"]["refused"] and d["guard"]["fenced"], d
        owned.append(d["owned"])
    assert sum(owned) == d["total"]                            # the shards partition the elements


COMM_WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import torch
    from torch import nn
    from styl3r_amd import dist_utils
    from styl3r_amd.ddp import BucketedGradReducer, broadcast_module_state, comm_report
    from styl3r_amd.train import make_optimizer
    rank, _, world = dist_utils.env_world()
    dist = dist_utils.init_distributed("gloo")
    res = {}
    for mode in ("all_reduce", "rs_ag"):
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(64, 300), nn.GELU(), nn.Linear(300, 300), nn.GELU(), nn.Linear(300, 8))
        new, pre = list(model[4].parameters()), list(model[0].parameters()) + list(model[2].parameters())
        red = BucketedGradReducer(list(model.parameters()), dist, bucket_bytes=128 * 1024, mode=mode, groups=[new, pre])
        opt = make_optimizer(new, pre, lr=1e-3, owner=red if mode == "rs_ag" else None)
        torch.manual_seed(50 + rank)
        x = torch.randn(256, 64)

        def step():
            red.wait_params(); red.prepare()
            model(x).pow(2).mean().backward()
            red.finish(); red.clip_grad_norm_(0.5, defer_to=None if mode == "all_reduce" else opt)
            opt.step(); red.gather_params()
        step()
        rep = comm_report(red, step, lambda: None, steps=2, destructive=True)
        red.wait_params()
        broadcast_module_state(model, dist)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        rep["replicas_identical_after_rebroadcast"] = bool(torch.equal(both[0], both[1]))
        rep["collective_restored"] = bool(red.collective)
        res[mode] = rep
        red.close()
    print(json.dumps(res), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % str(ROOT)


def test_comm_report_of_the_train_leg_runs_in_both_data_parallel_modes(tmp_path):
    """the exchange diagnosis bench.py's train leg prints under a process group (styl3r_amd.ddp.comm_report: per-bucket
    collective time, the step with and without collectives, the overlap fraction) on a world-2 gloo group, in both modes; the
    no-collective steps let the replicas drift, the re-broadcast makes them identical again, and the reducer's collectives are back on."""
    script = tmp_path / "comm.py"; script.write_text(COMM_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-3000:]
        d = json.loads(o.strip().splitlines()[-1])
        for mode in ("all_reduce", "rs_ag"):
            r = d[mode]
            assert r["mode"] == mode and r["world"] == 2 and r["buckets"] >= 2 and len(r["per_bucket_ms"]) == r["buckets"], r
            assert all(t > 0 for t in r["per_bucket_ms"]) and abs(r["comm_alone_ms"] - sum(r["per_bucket_ms"])) < 1e-2, r
            assert r["step_ms"] > 0 and r["step_ms_no_collectives"] > 0 and 0.0 <= r["overlap_frac"] <= 1.0, r
            assert r["replicas_identical_after_rebroadcast"] and r["collective_restored"], r

