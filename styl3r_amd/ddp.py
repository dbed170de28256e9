"""Data-parallel gradient exchange for the hot path: one process per GPU, bucketed gradient
all-reduce over RCCL/xGMI, overlapped with the backward.

The reference gets this from Lightning's `strategy="ddp_find_unused_parameters_true"`
(src/main_style.py:104-108): torch DDP, 25 MB buckets, a per-step unused-parameter graph scan.
Here the graph is static, so the bucket map is built once (no scan): parameters are packed in
REVERSE registration order (= autograd readiness: DPT heads -> decoders -> encoders) into
flat fp32 buckets; when the last gradient of a bucket lands, ONE multi-tensor copy packs the bucket (instead of
one accumulate kernel per parameter: ~1 800 tiny launches per step on the full encoder), p.grad is re-pointed
at the bucket slices (the optimizer then reads the reduced values in place, no unpack) and the bucket is
all-reduced on RCCL's own stream while the backward keeps running.  xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU):
large buckets (default 64 MiB) keep every link busy and amortise the launch latency of the collective.
Parameters that never receive a gradient on ANY rank (mask_token; head2 when v == 1) keep `grad = None` after
`finish()`, as under the reference's `find_unused_parameters=True`: AdamW then skips them (no weight decay, no moment
update).  With more than one rank the used / unused map is exchanged EVERY step -- one small int32 MAX all-reduce issued
asynchronously behind the buckets, unconditionally and identically on every rank (a collective must never be gated on
rank-local state: a rank whose arrival set changed alone would enter it alone and hang or pair with another rank's next bucket);
a single rank recomputes its map only when its own arrival set changes.

Contract: exactly ONE backward between `prepare()` and `finish()`.  Buckets are launched strictly in index order (a complete
bucket waits for its predecessors; whatever is left goes out in `finish()`), so the collective sequence is the same on every rank
whatever the local gradient-arrival order or set.  "Complete" = every parameter the previous step's AGREED map calls used has its
gradient: the never-used ones (`mask_token` sits ahead of the heads and the whole backbone in reverse registration order) are not waited
for, so the overlap with the backward survives them; if one of them does receive a gradient (a data-dependent branch), it travels in one
extra "straggler" collective that every rank enters together, decided from the MAX-reduced map of this step.  A second backward after a bucket was launched would race with the collective and
is refused with a RuntimeError.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
from torch import nn


def broadcast_module_state(module: nn.Module, dist, src: int = 0, chunk_bytes: int = 64 << 20, force_collective: bool = False) -> int:
    """Every rank starts from rank `src`'s parameters and buffers -- what torch DDP does when it wraps a module
    (`_sync_module_states`; the reference gets it from Lightning's DDP strategy, src/main_style.py:104-108, whose per-rank seed only
    differs for the data).  Tensors are packed per dtype into flat chunks of <= chunk_bytes, one broadcast each (a handful of large
    collectives instead of ~1 000 small ones).  Returns the number of bytes sent; a no-op without a process group / at world size 1
    (unless `force_collective`: a one-rank group still issues every collective -- the hardware smoke test of this code path)."""
    if dist is None or (dist.get_world_size() == 1 and not force_collective):
        return 0
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    groups: dict = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    sent = 0
    for (dtype, dev), ts in groups.items():
        chunk, nbytes = [], 0
        def flush():
            nonlocal chunk, nbytes, sent
            if not chunk:
                return
            flat = torch.cat([t.reshape(-1) for t in chunk])
            dist.broadcast(flat, src=src)
            off = 0
            for t in chunk:
                t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
            sent += flat.numel() * flat.element_size()
            chunk, nbytes = [], 0
        for t in ts:
            nb = t.numel() * t.element_size()
            if chunk and nbytes + nb > chunk_bytes:
                flush()
            chunk.append(t); nbytes += nb
        flush()
    return sent


class BucketedGradReducer:
    """mode "all_reduce" (default): one all-reduce per bucket, every rank clips and steps the whole model.
    mode "rs_ag" ("bucketed direct reduce-scatter + all-gather using all 7 links"): one REDUCE-SCATTER per bucket -- rank r
    ends up with the summed gradient of elements [r n/w, (r+1) n/w) of every bucket (buckets are padded to a multiple of the world
    size) -- the clip norm is evaluated on the owned shards (+ one scalar all-reduce), the optimizer updates the owned element ranges
    only (`owned_range`; optim.AdamWHIP(owner=reducer): 1/world of the 28 B x 1.05 G optimizer traffic), and `gather_params()` all-gathers
    the updated parameters bucket by bucket, asynchronously, `wait_params()` fencing the next forward.  For that the parameters of a
    bucket live in ONE flat buffer (`p.data` is re-pointed at construction), so both collectives run in place on whole buckets; over the
    xGMI mesh each of them moves (w-1)/w of a bucket per rank over all 7 links at once, where a ring all-reduce is bound by one link.
    After finish() in this mode `p.grad` is meaningful on the owned range only.  `groups`: parameter lists that must not share a
    bucket (optimizer groups: a shard then never straddles two learning rates)."""

    def __init__(self, params: Iterable[nn.Parameter], dist=None, bucket_bytes: int = 64 << 20, inplace_grads: bool = True,
                 force_collective: bool = False, mode: str = "all_reduce", groups: Optional[List[List[nn.Parameter]]] = None):
        if mode not in ("all_reduce", "rs_ag"):
            raise ValueError(f"BucketedGradReducer: mode {mode!r}: expected 'all_reduce' or 'rs_ag'")
        self.dist = dist
        self.mode = mode
        self.inplace_grads = inplace_grads
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        # `force_collective`: issue every collective even in a one-rank group (sum over one rank = identity), so that the stream
        # ordering between the in-place bucket writers, the pack copy and RCCL's stream is exercised on a single GPU
        self.collective = dist is not None and (self.world > 1 or force_collective)
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))
        group_of = {id(p): gi for gi, g in enumerate(groups or []) for p in g}
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in order:
            nb = p.numel() * 4
            if cur and (cur_bytes + nb > bucket_bytes or group_of.get(id(p)) != group_of.get(id(cur[-1]))):
                self.buckets.append(self._make_bucket(cur)); cur, cur_bytes = [], 0
            cur.append(p); cur_bytes += nb
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self._range: dict = {}            # id(p) -> (lo, hi) element range of p this rank owns ("rs_ag"), absent = none
        self._gathers: list = []
        if mode == "rs_ag":
            for b in self.buckets:
                lo, hi = b["shard"]
                off = 0
                for p in b["params"]:
                    a, z = max(lo - off, 0), min(hi - off, p.numel())
                    if a < z:
                        self._range[id(p)] = (a, z)
                    off += p.numel()
        self._handles: list = []
        self._armed = False
        # The per-step used / unused map is HOST data (which hooks fired): it is exchanged on a host-side (gloo) group, never through the
        # device.  Round 5 sent it as one more RCCL all-reduce and read it back with .cpu(): a full drain of the GPU queue in the middle of
        # every step (the read-back waits for the whole backward and every bucket), after which the clip, the optimizer pass and the next
        # forward were enqueued onto an idle device -- the +11 ms of a one-rank RCCL step over the same step without a group
        # (profiles/r05j_bench_torchrun1.json), which every rank of an N-GPU job paid as well.
        self._host_group = None
        if self.collective and dist.get_backend() != "gloo":
            self._host_group = dist.new_group(backend="gloo")      # (collective: every rank constructs its reducer at the same point)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._arrived: set = set()
        self._arrived_key = None        # the local arrival set the cached map was computed for
        self._next = 0                  # index of the next bucket to launch
        self._unused: List[nn.Parameter] = []
        self._skipped: list = []
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _make_bucket(self, params):
        dev, n = params[0].device, sum(p.numel() for p in params)
        padded = n if self.mode == "all_reduce" else -(-n // self.world) * self.world
        flat = torch.zeros(padded, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p)); off += p.numel()
        b = dict(params=list(params), flat=flat, views=views, pending=len(params), launched=False, expect={id(p) for p in params}, n=n)
        if self.mode == "rs_ag":
            per = padded // self.world
            b["shard"] = (self.rank * per, (self.rank + 1) * per)
            # the parameters of the bucket move into one flat buffer (values kept), so the all-gather of the updated shards runs in place
            pflat = torch.zeros(padded, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in params:
                    if p.dtype != torch.float32:
                        raise NotImplementedError("BucketedGradReducer(mode='rs_ag'): fp32 parameters only")
                    dst = pflat[off:off + p.numel()].view_as(p)
                    dst.copy_(p.data); p.data = dst; off += p.numel()
            b["pflat"] = pflat
        return b

    def owned_range(self, p) -> Optional[tuple]:
        """(lo, hi): the flat element range of `p` whose reduced gradient this rank holds and whose update it owns; None if it owns
        nothing of `p`.  Mode "all_reduce": everything."""
        if self.mode != "rs_ag":
            return (0, p.numel())
        return self._range.get(id(p))

    def gather_params(self):
        """after the optimizer step ("rs_ag"): all-gather the updated shards, bucket by bucket, asynchronously on the collective's own
        stream; `wait_params()` before anything reads the parameters."""
        if self.mode != "rs_ag" or not self.collective:
            return
        for b in self.buckets:
            lo, hi = b["shard"]
            self._gathers.append(self.dist.all_gather_into_tensor(b["pflat"], b["pflat"][lo:hi], async_op=True))

    def wait_params(self):
        """fence of the asynchronous parameter all-gather ("rs_ag").  The optimizer bumped `Tensor._version` when IT wrote its owned ranges;
        the other ranks' ranges only change here, so the versions are bumped again once the gather has landed (a forward
        between step() and this fence had cached split weight images of the stale shards under the new version for good)."""
        if not self._gathers:
            return
        for h in self._gathers:
            h.wait()
        self._gathers.clear()
        torch.autograd.graph.increment_version([p for b in self.buckets for p in b["params"]])

    def guard_readers(self, module: nn.Module):
        """every reader of the parameters outside the train step -- a validation / inference forward of `module`, `state_dict()` for a
        checkpoint -- first waits for the parameter all-gather in flight (forward pre-hook + state-dict pre-hook; no-ops otherwise)."""
        self._hooks.append(module.register_forward_pre_hook(lambda m, args: self.wait_params()))
        self._hooks.append(module.register_state_dict_pre_hook(lambda m, prefix, keep_vars: self.wait_params()))

    @torch.no_grad()
    def consolidate_optimizer_state(self, optimizer, keys=("exp_avg", "exp_avg_sq")):
        """COLLECTIVE ("rs_ag"): all-gather the moment shards so that `optimizer.state_dict()` holds the full moments on every rank (each
        rank only ever updates the moments of its owned element ranges; a checkpoint written from one rank alone would store zeros for
        the rest).  Call on every rank before `state_dict()`; the sharded optimizers refuse `state_dict()` otherwise."""
        if self.mode == "rs_ag":
            self.wait_params()
            for b in self.buckets:
                lo, hi = b["shard"]
                for key in keys:
                    flat = torch.zeros_like(b["pflat"])
                    off = 0
                    for p in b["params"]:
                        st, r = optimizer.state.get(p), self._range.get(id(p))
                        if st and key in st and r is not None:
                            flat[off + r[0]:off + r[1]] = st[key].reshape(-1)[r[0]:r[1]]
                        off += p.numel()
                    if self.collective:
                        self.dist.all_gather_into_tensor(flat, flat[lo:hi].clone())
                    off = 0
                    for p in b["params"]:
                        st = optimizer.state.get(p)
                        if st and key in st:
                            st[key].copy_(flat[off:off + p.numel()].view_as(p))
                        off += p.numel()
        optimizer._shards_consolidated = True

    def _make_hook(self, bi):
        def hook(p):
            if not self._armed:                 # a backward outside prepare()/finish() is left alone
                return
            b = self.buckets[bi]
            if b["launched"] and id(p) in b["expect"]:
                raise RuntimeError("BucketedGradReducer: a gradient arrived for a bucket that was already all-reduced -- "
                                   "only ONE backward is allowed between prepare() and finish() (no gradient accumulation / "
                                   "retain_graph re-runs through the reducer)")
            self._arrived.add(self._index[id(p)])
            if id(p) in b["expect"]:            # (a parameter the agreed map calls unused may still show up: counted as arrived, never waited for)
                b["pending"] -= 1
            # buckets are launched strictly in index order, so that every rank issues the same collective sequence even
            # when a gradient arrives in a different order (or not at all) on some rank
            while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
                self._launch(self.buckets[self._next]); self._next += 1
        return hook

    def _launch(self, b):
        b["launched"] = True
        # (gradients the kernels already produced in place -- vit_ops._FusedLinear with `_grad_slot` -- need no copy).  Parameters of the
        # agreed unused map are NOT part of the bucket's payload (their slice stays zero): see `finish()`
        live = [(p, v) for p, v in zip(b["params"], b["views"]) if id(p) in b["expect"]]
        have = [(p, v) for p, v in live if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
        for p, v in live:
            p.grad = v
        if self.collective:
            if self.mode == "rs_ag":
                lo, hi = b["shard"]
                self._handles.append(self.dist.reduce_scatter_tensor(b["flat"][lo:hi], b["flat"], op=self.dist.ReduceOp.SUM, async_op=True))
            else:
                self._handles.append(self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, async_op=True))

    def prepare(self):
        """call before backward: zero the buckets (parameters without a gradient reduce as zeros) and drop old grads."""
        self._handles.clear()
        self._armed = True
        self._arrived = set()
        self._next = 0
        torch._foreach_zero_([b["flat"] for b in self.buckets])
        # a bucket is complete when every parameter that is EXPECTED to receive a gradient has one.  Parameters of the agreed unused
        # map of the previous step (MAX-reduced over the ranks in finish(): identical everywhere; e.g. `mask_token`, which sits in reverse
        # registration order AHEAD of the heads and the whole backbone) are not waited for -- otherwise their bucket, and with the
        # in-order launch rule every later one, would only go out in finish(), with no overlap with the backward.  Should such a
        # parameter receive a gradient after its bucket went out, the hook refuses it (RuntimeError) as any late arrival.
        skip = {id(p) for p in self._unused}
        self._skipped = []                  # (index, parameter, bucket view) of the parameters nobody waits for, in index order
        for b in self.buckets:
            b["expect"] = {id(p) for p in b["params"]} - skip
            b["pending"], b["launched"] = len(b["expect"]), False
            for p, v in zip(b["params"], b["views"]):
                p.grad = None
                if id(p) in skip:
                    # should it receive a gradient after all (a data-dependent branch), the gradient must not land in a slice that may
                    # already be inside a collective: it stays a tensor of its own and goes out with the stragglers in finish()
                    p._grad_slot = None
                    self._skipped.append((self._index[id(p)], p, v))
                    continue
                # in-place gradient slot: layers that can (the bf16x6 Linear) accumulate dW / db straight into this zeroed
                # slice of the bucket and hand it to autograd as the gradient (no per-layer memset, no pack copy)
                p._grad_slot = {"view": v, "used": False} if self.inplace_grads else None
        self._skipped.sort(key=lambda t: t[0])

    def finish(self):
        """call after backward: reduce the buckets whose gradients never all arrived, wait, average."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        # the used / unused exchange: unconditional and symmetric (same collective sequence on every rank, every step)
        used = None
        if self.collective:
            # host tensors on the host-side group: no device work, no device sync (a gloo default group -- the CPU tests -- serves itself)
            used = torch.zeros(len(self.params), dtype=torch.int32)
            if self._arrived:
                used[torch.tensor(sorted(self._arrived))] = 1
            self.dist.all_reduce(used, op=self.dist.ReduceOp.MAX, **({"group": self._host_group} if self._host_group is not None else {}))
        for h in self._handles:
            h.wait()
        self._handles.clear()
        self._armed = False
        for b in self.buckets:
            for p in b["params"]:
                p._grad_slot = None            # a backward outside prepare()/finish() must not write into the buckets
        # parameters no rank produced a gradient for: grad = None (the optimizer skips them), as DDP(find_unused_parameters)
        if used is not None:
            flags = used.tolist()                  # (host memory already)
            used_now = {i for i, u in enumerate(flags) if u}
        else:
            used_now = self._arrived               # one rank: nothing to exchange
        # stragglers: parameters the previous step's map called unused that some rank DID use in this one.  The set comes from the
        # MAX-reduced map, so every rank takes this branch together and issues the same (single) extra collective
        late = [(i, p, v) for i, p, v in self._skipped if i in used_now]
        if late:
            parts = [(p.grad if (i in self._arrived and p.grad is not None) else torch.zeros_like(v)).reshape(-1).float() for i, p, v in late]
            flat = torch.cat(parts)
            if self.collective:
                self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM)
            off = 0
            for _, p, v in late:
                v.copy_(flat[off:off + v.numel()].view_as(v)); off += v.numel()
                p.grad = v
        if self.world > 1:
            torch._foreach_mul_(self._reduced_flats(), 1.0 / self.world)
        if used is not None:
            self._unused = [p for i, p in enumerate(self.params) if i not in used_now]
        else:
            key = frozenset(self._arrived)         # the map changes only with the local set
            if key != self._arrived_key:
                self._unused = [p for i, p in enumerate(self.params) if i not in key]
                self._arrived_key = key
        for p in self._unused:
            p.grad = None

    def clip_grad_norm_(self, max_norm: float, defer_to: Optional[torch.optim.Optimizer] = None) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ (L2, eps 1e-6, coefficient clamped to 1) evaluated on the flat buckets:
        one norm + one scale per bucket instead of one per parameter.  Call after finish().
        `defer_to`: a FUSED Adam / AdamW whose next step() applies the coefficient itself -- its kernel divides every gradient by
        `optimizer.grad_scale` while it reads it (the hook torch.amp's GradScaler uses), so the separate read-modify-write pass over
        the 4.2 GB of gradients disappears; the stored gradients are then scaled by that step, not by this call."""
        flats = self._reduced_flats()
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(flats)))
        if self.mode == "rs_ag" and self.collective:        # the shards partition the gradient: ||g||^2 = sum over ranks of the local squares
            sq = total * total
            self.dist.all_reduce(sq, op=self.dist.ReduceOp.SUM)
            total = sq.sqrt()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        if defer_to is not None and defer_to.defaults.get("fused"):
            # 0-dim; g / grad_scale == g * coef.  A non-finite norm must stay visible: g / inf would turn an overflowed step into a silent
            # zero update, so the scale becomes NaN and the step poisons the parameters as the in-place g * coef path does.  The optimizer
            # drops the attribute after its step (one coefficient per step); the STORED gradients stay unclipped on this path.
            defer_to.grad_scale = torch.where(torch.isfinite(total), 1.0 / coef, torch.full_like(total, float("nan"))).float()
        else:
            torch._foreach_mul_(flats, coef)
        return total

    def _reduced_flats(self) -> List[torch.Tensor]:
        """the part of every bucket that holds reduced gradients on this rank"""
        if self.mode == "rs_ag" and self.collective:
            return [b["flat"][b["shard"][0]:b["shard"][1]] for b in self.buckets]
        return [b["flat"] for b in self.buckets]

    def bucket_sizes_bytes(self) -> List[int]:
        return [b["n"] * 4 for b in self.buckets]

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()


def comm_report(reducer: "BucketedGradReducer", run_step, sync, steps: int = 3, destructive: bool = False) -> dict:
    """One-shot diagnosis of the gradient exchange of a data-parallel step (bench.py prints it when a process group exists, so the first
    multi-GPU run explains its own scaling number).  Backend-agnostic: nothing here looks inside RCCL.
      per_bucket_ms   each bucket's collective(s) ALONE on its real buffer, nothing else queued (sync-bracketed, MAX over ranks):
                      all_reduce, or reduce_scatter + the parameter all_gather of "rs_ag"
      comm_alone_ms   their sum: the exchange with no compute to hide under
      step_ms         `run_step()` as it is (collectives overlapped with the backward)
      step_ms_no_collectives   the same step with the collectives switched off (every rank steps on its local gradients: the
                      replicas DIVERGE -- the caller re-broadcasts the module state afterwards)
      overlap_frac    (step_ms_no_collectives + comm_alone_ms - step_ms) / comm_alone_ms clamped to [0, 1]: the share of the exchange
                      that the compute hides; 1 = free, 0 = fully exposed.
    DESTRUCTIVE: the no-collective steps are real optimizer steps on rank-local gradients -- parameters, moments and step
    counters diverge across the ranks, and under "rs_ag" the moment shards keep those updates even after the caller re-broadcasts the
    module.  The caller has to say so (`destructive=True`) and restore / discard the training state afterwards; a benchmark does, a
    training loop must not call this."""
    import time
    if not destructive:
        raise RuntimeError("comm_report runs optimizer steps without the gradient exchange (the replicas diverge): pass destructive=True "
                           "from a benchmark that re-broadcasts or discards the training state afterwards")
    dist = reducer.dist

    def timed(fn, n):
        sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        dt = (time.perf_counter() - t0) / n
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=reducer.buckets[0]["flat"].device if reducer.buckets else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt * 1e3

    out = {"mode": reducer.mode, "world": reducer.world, "buckets": len(reducer.buckets), "bucket_MiB": [round(b["n"] * 4 / 2 ** 20, 1) for b in reducer.buckets]}
    if not reducer.collective:
        out["note"] = "no process group: nothing to exchange"
        return out
    reducer.wait_params()
    per = []
    for b in reducer.buckets:
        scratch = torch.zeros_like(b["flat"])           # (the real gradients / parameters are not touched)
        if reducer.mode == "rs_ag":
            lo, hi = b["shard"]
            pscr = torch.zeros_like(b["pflat"])

            def one(scratch=scratch, pscr=pscr, lo=lo, hi=hi):
                dist.reduce_scatter_tensor(scratch[lo:hi], scratch, op=dist.ReduceOp.SUM)
                dist.all_gather_into_tensor(pscr, pscr[lo:hi])
        else:
            def one(scratch=scratch):
                dist.all_reduce(scratch, op=dist.ReduceOp.SUM)
        one()                                            # (first use of a buffer size: communicator warm-up)
        per.append(round(timed(one, steps), 3))
        del scratch
    out["per_bucket_ms"] = per
    out["comm_alone_ms"] = round(sum(per), 3)
    out["step_ms"] = round(timed(run_step, steps), 3)
    keep = reducer.collective
    try:
        reducer.wait_params()
        reducer.collective = False
        run_step()
        out["step_ms_no_collectives"] = round(timed(run_step, steps), 3)
    finally:
        reducer.collective = keep
    hidden = out["step_ms_no_collectives"] + out["comm_alone_ms"] - out["step_ms"]
    out["overlap_frac"] = round(min(1.0, max(0.0, hidden / out["comm_alone_ms"])), 3) if out["comm_alone_ms"] > 0 else None
    out["exchanged_GB_per_step"] = round(sum(b["n"] for b in reducer.buckets) * 4 / 1e9, 3)
    out["busbw_GBps_alone"] = round(out["exchanged_GB_per_step"] * 2 * (reducer.world - 1) / max(reducer.world, 1) / (out["comm_alone_ms"] * 1e-3), 1) \
        if out["comm_alone_ms"] > 0 else None
    return out

