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
whatever the local gradient-arrival order or set.  A second backward after a bucket was launched would race with the collective and
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
    def __init__(self, params: Iterable[nn.Parameter], dist=None, bucket_bytes: int = 64 << 20, inplace_grads: bool = True,
                 force_collective: bool = False):
        self.dist = dist
        self.inplace_grads = inplace_grads
        self.world = dist.get_world_size() if dist is not None else 1
        # `force_collective`: issue every collective even in a one-rank group (sum over one rank = identity), so that the stream
        # ordering between the in-place bucket writers, the pack copy and RCCL's stream is exercised on a single GPU
        self.collective = dist is not None and (self.world > 1 or force_collective)
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in order:
            nb = p.numel() * 4
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(self._make_bucket(cur)); cur, cur_bytes = [], 0
            cur.append(p); cur_bytes += nb
        if cur:
            self.buckets.append(self._make_bucket(cur))
        self._handles: list = []
        self._armed = False
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._arrived: set = set()
        self._arrived_key = None        # the local arrival set the cached map was computed for
        self._next = 0                  # index of the next bucket to launch
        self._unused: List[nn.Parameter] = []
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    @staticmethod
    def _make_bucket(params):
        dev, n = params[0].device, sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p)); off += p.numel()
        return dict(params=list(params), flat=flat, views=views, pending=len(params), launched=False)

    def _make_hook(self, bi):
        def hook(p):
            if not self._armed:                 # a backward outside prepare()/finish() is left alone
                return
            b = self.buckets[bi]
            if b["launched"]:
                raise RuntimeError("BucketedGradReducer: a gradient arrived for a bucket that was already all-reduced -- "
                                   "only ONE backward is allowed between prepare() and finish() (no gradient accumulation / "
                                   "retain_graph re-runs through the reducer)")
            self._arrived.add(self._index[id(p)])
            b["pending"] -= 1
            # buckets are launched strictly in index order, so that every rank issues the same collective sequence even
            # when a gradient arrives in a different order (or not at all) on some rank
            while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
                self._launch(self.buckets[self._next]); self._next += 1
        return hook

    def _launch(self, b):
        b["launched"] = True
        # (gradients the kernels already produced in place -- vit_ops._FusedLinear with `_grad_slot` -- need no copy)
        have = [(p, v) for p, v in zip(b["params"], b["views"]) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
        for p, v in zip(b["params"], b["views"]):
            p.grad = v
        if self.collective:
            self._handles.append(self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, async_op=True))

    def prepare(self):
        """call before backward: zero the buckets (parameters without a gradient reduce as zeros) and drop old grads."""
        self._handles.clear()
        self._armed = True
        self._arrived = set()
        self._next = 0
        torch._foreach_zero_([b["flat"] for b in self.buckets])
        for b in self.buckets:
            b["pending"], b["launched"] = len(b["params"]), False
            for p, v in zip(b["params"], b["views"]):
                p.grad = None
                # in-place gradient slot: layers that can (the bf16x6 Linear) accumulate dW / db straight into this zeroed
                # slice of the bucket and hand it to autograd as the gradient (no per-layer memset, no pack copy)
                p._grad_slot = {"view": v, "used": False} if self.inplace_grads else None

    def finish(self):
        """call after backward: reduce the buckets whose gradients never all arrived, wait, average."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        # the used / unused exchange: unconditional and symmetric (same collective sequence on every rank, every step)
        used = None
        if self.collective:
            used = torch.zeros(len(self.params), dtype=torch.int32, device=self.params[0].device if self.params else "cpu")
            if self._arrived:
                used[torch.tensor(sorted(self._arrived), device=used.device)] = 1
            self._handles.append(self.dist.all_reduce(used, op=self.dist.ReduceOp.MAX, async_op=True))
        for h in self._handles:
            h.wait()
        self._handles.clear()
        self._armed = False
        for b in self.buckets:
            for p in b["params"]:
                p._grad_slot = None            # a backward outside prepare()/finish() must not write into the buckets
        if self.world > 1:
            torch._foreach_mul_([b["flat"] for b in self.buckets], 1.0 / self.world)
        # parameters no rank produced a gradient for: grad = None (the optimizer skips them), as DDP(find_unused_parameters)
        if used is not None:
            flags = used.cpu().tolist()            # 4 B per parameter; the optimizer step that follows needs the host anyway
            self._unused = [p for p, u in zip(self.params, flags) if not u]
        else:
            key = frozenset(self._arrived)         # one rank: nothing to exchange, the map changes only with the local set
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
        flats = [b["flat"] for b in self.buckets]
        total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(flats)))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        if defer_to is not None and defer_to.defaults.get("fused"):
            defer_to.grad_scale = (1.0 / coef).float()                     # 0-dim; g / grad_scale == g * coef
        else:
            torch._foreach_mul_(flats, coef)
        return total

    def bucket_sizes_bytes(self) -> List[int]:
        return [b["flat"].numel() * 4 for b in self.buckets]

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
