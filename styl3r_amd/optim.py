"""AdamW of the train step (src/model/model_wrapper_style.py:885-895: AdamW(param_dicts, lr, weight_decay=0.05, betas=(0.9, 0.95)))
on the hand-written optimizer pass csrc/vit_optim.hip: ONE launch per parameter group over a device table of <= 16 384-element chunks
instead of the framework's 64 multi-tensor launches (8.9 -> ~5 ms per step at full size: 1.05 G parameters x 28 bytes).

`AdamWHIP` IS a torch.optim.AdamW (fused flavour): same constructor defaults, same state layout (`step` = 0-dim float32 device tensor per
parameter, `exp_avg`, `exp_avg_sq`), so LR schedulers, `state_dict()` / `load_state_dict()` and checkpoints are interchangeable with the
framework's optimizer; only `step()` is replaced.  `optimizer.grad_scale` (0-dim device tensor) keeps the meaning it has for the
framework's fused kernel -- every gradient is divided by it while it is read (ddp.BucketedGradReducer.clip_grad_norm_(defer_to=...)) --
except that the stored gradients are not rewritten (4.2 GB of stores nobody reads)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import vit_ops

_CHUNK = 16384
_CHUNK_DTYPE = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("step", "u8"), ("n", "i4"), ("vec", "i4"), ("amax", "u8")])   # VitAdamChunk
_AMAX_WORDS = 64 * 32     # one |max| word (vit_ops._AmaxArena.LINE): 64 slots, one per cache line


def _lib():
    lib = vit_ops.load()
    lib.vit_adamw_step.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.vit_adamw_step.restype = C.c_int
    return lib


class AdamWHIP(torch.optim.AdamW):
    """`owner`: optional object with `owned_range(p) -> (lo, hi) | None` (ddp.BucketedGradReducer in "rs_ag" mode): only that element range
    of every parameter is updated by this rank (the rest arrives with the parameter all-gather); None = everything."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, owner=None):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self._tables, self._step_flat = {}, {}
        self._owner = owner
        self._amax = {}          # group index -> (flat int32 tensor of per-parameter |max| words, [parameters in table order])
        _lib()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables, self._step_flat = {}, {}                # the moments and counters were replaced

    def _advance_steps(self, gi, params):
        """step += 1 for every parameter with a gradient.  The counters of a group are 0-dim views of ONE flat tensor (packed here on
        first use and after load_state_dict), so the usual case -- every parameter of the group has a gradient -- is one launch; the
        framework's _foreach_add_ over ~600 scalar tensors is 75 launches / 0.6 ms."""
        everyone = [p for p in self.param_groups[gi]["params"] if self.state.get(p)]       # .get: indexing the defaultdict would plant {} entries that state_dict() then carries
        flat = self._step_flat.get(gi)
        if flat is None or flat.numel() != len(everyone) or any(self.state[p]["step"].data_ptr() != flat.data_ptr() + 4 * i for i, p in enumerate(everyone)):
            flat = torch.stack([self.state[p]["step"].to(params[0].device, torch.float32).reshape(()) for p in everyone])
            for i, p in enumerate(everyone):
                self.state[p]["step"] = flat[i]
            self._step_flat[gi] = flat
            self._tables.pop(gi, None)                        # the table points at the counters
        if len(params) == len(everyone):
            flat += 1
        else:
            torch._foreach_add_([self.state[p]["step"] for p in params], 1)

    def _table(self, gi, params):
        key = tuple((id(p), p.data_ptr(), p.grad.data_ptr()) for p in params)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        rows = []
        n_weights = sum(1 for p in params if p.dim() >= 2)
        amax_flat = torch.zeros(max(n_weights, 1) * _AMAX_WORDS, dtype=torch.int32, device=params[0].device) if n_weights else None
        amax_params = []
        self._amax[gi] = (amax_flat, amax_params)
        for p in params:
            st = self.state[p]
            g, m, v = p.grad, st["exp_avg"], st["exp_avg_sq"]
            for t, what in ((p, "parameter"), (g, "gradient"), (m, "exp_avg"), (v, "exp_avg_sq")):
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == p.device and not t.is_sparse):
                    raise NotImplementedError(f"AdamWHIP: {what} of shape {tuple(t.shape)} must be a contiguous fp32 tensor on the parameter's GPU")
            lo, hi = 0, p.numel()
            if self._owner is not None:
                rng = self._owner.owned_range(p)
                if rng is None:
                    continue
                lo, hi = rng
            ptrs = [t.data_ptr() + 4 * lo for t in (p, g, m, v)]
            vec = int(all(a % 16 == 0 for a in ptrs))
            n = hi - lo
            # the |max| of the updated values, for the f16x3 weight scale: only weights (>= 2-D) that are updated as a whole by this rank
            word = 0
            if amax_flat is not None and p.dim() >= 2 and lo == 0 and hi == p.numel():
                word = amax_flat.data_ptr() + 4 * _AMAX_WORDS * len(amax_params)
                amax_params.append(p)
            for off in range(0, n, _CHUNK):
                rows.append((ptrs[0] + 4 * off, ptrs[1] + 4 * off, ptrs[2] + 4 * off, ptrs[3] + 4 * off, st["step"].data_ptr(), min(_CHUNK, n - off), vec, word))
        host = np.array(rows, dtype=_CHUNK_DTYPE) if rows else np.zeros(0, dtype=_CHUNK_DTYPE)
        if not rows:
            self._tables[gi] = (key, torch.empty(0, dtype=torch.uint8, device=params[0].device), 0)
            return self._tables[gi][1], 0
        # (rebuilt only when a parameter / gradient pointer changes: with the reducer's stable bucket views that is once per run)
        dev = torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).pin_memory().to(params[0].device, non_blocking=True)
        self._tables[gi] = (key, dev, len(rows))
        return dev, len(rows)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib()
        gs = getattr(self, "grad_scale", None)
        if getattr(self, "found_inf", None) is not None:
            raise NotImplementedError("AdamWHIP: found_inf (torch.amp.GradScaler) is not supported; the hot path trains in fp32")
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
                raise NotImplementedError("AdamWHIP: amsgrad / maximize / capturable / differentiable are not part of the reference's configuration")
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                st = self.state[p]                             # (a parameter WITH a gradient gets its state here, as in torch)
                if len(st) == 0:                               # torch.optim.AdamW._init_group, fused flavour
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            dev = params[0].device
            if any(p.device != dev for p in params):
                raise NotImplementedError("AdamWHIP: one device per parameter group")
            self._advance_steps(gi, params)
            table, n_chunks = self._table(gi, params)
            if gs is not None and not (gs.is_cuda and gs.device == dev and gs.dtype == torch.float32 and gs.numel() == 1):
                raise ValueError("AdamWHIP: grad_scale must be a one-element fp32 tensor on the parameters' device")
            b1, b2 = group["betas"]
            amax_flat, amax_params = self._amax.get(gi, (None, []))
            if amax_params:
                amax_flat.zero_()
            if n_chunks:
                rc = lib.vit_adamw_step(table.data_ptr(), n_chunks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                        float(group["weight_decay"]), gs.data_ptr() if gs is not None else None,
                                        torch.cuda.current_stream(dev).cuda_stream)
                vit_ops._check(rc, "vit_adamw_step")
            # the kernel wrote the parameters through raw pointers: tell autograd (and every cache keyed on `_version` -- the pre-split
            # bf16 weight images of vit_ops._SPLIT_CACHE) that they changed, exactly as an in-place torch op would
            torch.autograd.graph.increment_version(params)
            # hand the words to the split cache: the next f16x3 image of each weight takes its scale from here (no vit_amax pass)
            for i, p in enumerate(amax_params):
                vit_ops.register_weight_amax(p, amax_flat[i * _AMAX_WORDS:(i + 1) * _AMAX_WORDS])
        # the coefficient belongs to THIS step's gradients (ddp.BucketedGradReducer.clip_grad_norm_(defer_to=...)); a later step without a
        # fresh clip must not reuse it
        self.grad_scale = None
        self._shards_consolidated = False
        return loss

    def state_dict(self):
        _refuse_sharded_state_dict(self)
        return super().state_dict()


def _refuse_sharded_state_dict(opt):
    """owned-range optimizers ("rs_ag"): the moments of the element ranges other ranks own are zeros here until
    `reducer.consolidate_optimizer_state(optimizer)` (a collective) has gathered them"""
    owner = getattr(opt, "_owner", None)
    if owner is not None and getattr(owner, "mode", "") == "rs_ag" and owner.world > 1 and not getattr(opt, "_shards_consolidated", True):
        raise RuntimeError("state_dict() of a sharded (rs_ag) optimizer: call reducer.consolidate_optimizer_state(optimizer) on EVERY rank "
                           "first -- each rank holds the moments of its owned element ranges only")


class ShardedAdamWTorch(torch.optim.AdamW):
    """The same owned-range AdamW in plain torch ops, for parameters that do not live on a GPU (the world-size-2 gloo tests of the
    "rs_ag" exchange): identical arithmetic and state layout (fused flavour: `step` is a float32 tensor), one Python loop over the owned
    slices.  Not a product path -- on the GPU `AdamWHIP(owner=reducer)` runs the owned ranges through csrc/vit_optim.hip."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, owner=None):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=False)
        self._owner = owner

    @torch.no_grad()
    def step(self, closure=None):
        gs = getattr(self, "grad_scale", None)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p); st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1                                  # on every rank, owner or not: the counters stay identical
                rng = (0, p.numel()) if self._owner is None else self._owner.owned_range(p)
                if rng is None:
                    continue
                lo, hi = rng
                t = float(st["step"])
                P, G = p.data.reshape(-1)[lo:hi], p.grad.reshape(-1)[lo:hi]
                M, V = st["exp_avg"].reshape(-1)[lo:hi], st["exp_avg_sq"].reshape(-1)[lo:hi]
                if gs is not None:
                    G = G / gs
                P.mul_(1 - lr * wd)
                M.lerp_(G, 1 - b1)
                V.mul_(b2).addcmul_(G, G, value=1 - b2)
                denom = (V.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
                P.addcdiv_(M, denom, value=-lr / (1 - b1 ** t))
            touched = [p for p in group["params"] if p.grad is not None]
            if touched:
                torch.autograd.graph.increment_version(touched)
        self.grad_scale = None
        self._shards_consolidated = False
        return None

    def state_dict(self):
        _refuse_sharded_state_dict(self)
        return super().state_dict()
