"""Train step over the hot path (encoder -> decoder -> losses -> backward -> DP all-reduce -> clip -> AdamW), the
part of `ModelWrapperStyle.training_step` / `configure_optimizers` (src/model/model_wrapper_style.py:118-232,
843-916) that the benchmarks and the multi-GPU path need.  Everything else of the LightningModule (logging,
video, validation, distillation) is out of scope."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os

import torch
from torch import nn

from . import vit_ops
from .ddp import BucketedGradReducer, broadcast_module_state
from .losses import mse_loss


def select_trainable(encoder: nn.Module) -> Tuple[List[nn.Parameter], List[nn.Parameter], List[str]]:
    """Parameter selection of `configure_optimizers` (model_wrapper_style.py:845-883), same substring rules on the same
    parameter names.  Returns (new_params [lr], pretrained_params [lr * backbone_lr_multiplier], frozen names);
    frozen parameters get `requires_grad = False` so nothing is tracked for them (`:866-868`).
      stylized     : train `*stylizer.dec*` + `gaussian_appearance_head`; fine-tune the style encoder
                     (`stylizer.enc*`, `stylizer.mask_token`, `stylizer.patch_embed`); freeze the rest.
      not stylized : new = stylizer decoder, both Gaussian heads, intrinsic_encoder; everything else pretrained."""
    new, pre, frozen = [], [], []
    stylized = bool(getattr(encoder, "stylized", False))
    for name, p in encoder.named_parameters():
        if not p.requires_grad:
            continue
        if stylized:
            if "stylizer.dec" in name or "gaussian_appearance_head" in name:
                new.append(p)
            elif "stylizer.enc" in name or "stylizer.mask_token" in name or "stylizer.patch_embed" in name:
                pre.append(p)
            else:
                p.requires_grad = False
                frozen.append(name)
        else:
            if ("stylizer.dec" in name or "gaussian_appearance_head" in name or "gaussian_param_head" in name
                    or "intrinsic_encoder" in name):
                new.append(p)
            else:
                pre.append(p)
    return new, pre, frozen


OPTIMIZER_IMPL = os.environ.get("STYL3R_OPTIMIZER", "hip")      # "hip": styl3r_amd.optim.AdamWHIP (one launch per group) | "torch": torch.optim.AdamW(fused=True); CPU parameters always take torch's


# 1 = all cached weight images are rebuilt by ONE launch right after the optimizer step (vit_ops.refresh_split_cache).  Measured (DESIGN R5.3): 4.7 ms
# less kernel time per C3 step, 2 ms MORE wall time -- the ~1 270 lazy launches sit in host-bound stretches of the forward / backward where the GPU
# would idle anyway, the batched kernel is 5.3 ms of serial GPU time at the step's end.  Default: off (lazy, as in rounds 1 - 4).
BATCHED_RESPLIT = os.environ.get("STYL3R_RESPLIT", "0") == "1"
DP_MODE = os.environ.get("STYL3R_DP_MODE", "all_reduce")       # "all_reduce" | "rs_ag" (reduce-scatter + sharded AdamW + parameter all-gather, ddp.py)


def make_optimizer(new: Sequence[nn.Parameter], pre: Sequence[nn.Parameter], lr: float = 2e-4,
                   backbone_lr_multiplier: float = 0.1, owner=None) -> torch.optim.Optimizer:
    """AdamW(param_dicts, lr, weight_decay=0.05, betas=(0.9, 0.95)) of `:885-895`; on a GPU the single-pass optimizer kernel
    (csrc/vit_optim.hip through optim.AdamWHIP, state-compatible with the framework's fused AdamW; the foreach implementation makes
    ~10 passes over the 4.2 GB of parameter / moment state per step)."""
    groups = [g for g in ({"params": list(new), "lr": lr}, {"params": list(pre), "lr": lr * backbone_lr_multiplier})
              if g["params"]]
    on_gpu = all(p.is_cuda for g in groups for p in g["params"])
    if owner is not None and getattr(owner, "mode", "all_reduce") == "rs_ag":
        # sharded step: this rank updates only the element ranges whose reduced gradient it holds (`owner.owned_range`)
        from .optim import AdamWHIP, ShardedAdamWTorch
        cls = AdamWHIP if on_gpu else ShardedAdamWTorch
        return cls(groups, lr=lr, weight_decay=0.05, betas=(0.9, 0.95), owner=owner)
    if on_gpu and OPTIMIZER_IMPL == "hip" and all(p.dtype == torch.float32 for g in groups for p in g["params"]):
        from .optim import AdamWHIP
        return AdamWHIP(groups, lr=lr, weight_decay=0.05, betas=(0.9, 0.95))
    fused = any(p.is_cuda for g in groups for p in g["params"])
    opt = torch.optim.AdamW(groups, lr=lr, weight_decay=0.05, betas=(0.9, 0.95), fused=fused)
    opt.register_step_post_hook(_after_step)
    return opt


def _after_step(optimizer, args, kwargs):
    """The framework's FUSED AdamW does not always bump `Tensor._version` of the parameters it rewrites (torch 2.10: stays 0), and
    everything keyed on the version counter -- vit_ops._SPLIT_CACHE's pre-split weight images -- would keep serving the old weights.
    Bump it for every parameter the step touched, and drop the one-step clip coefficient."""
    touched = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    if touched:
        torch.autograd.graph.increment_version(touched)
    if getattr(optimizer, "grad_scale", None) is not None:
        optimizer.grad_scale = None


def make_lr_scheduler(optimizer, warm_up_steps: int, max_steps: int, lr: float):
    """LinearLR(1/warm_up .. 1) for `warm_up_steps`, then CosineAnnealingLR(T_max=max_steps, eta_min=0.1*lr) (`:896-906`)."""
    warm = torch.optim.lr_scheduler.LinearLR(optimizer, 1 / warm_up_steps, 1, total_iters=warm_up_steps)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=max_steps, eta_min=lr * 0.1)
    return torch.optim.lr_scheduler.SequentialLR(optimizer, schedulers=[warm, cos], milestones=[warm_up_steps])


class TrainStep:
    """One optimisation step.  `losses`: modules called as loss(prediction, batch, gaussians, global_step) and summed
    (`:205-209`); default = MSE.  `identity_loss`: if given, a second encoder + decoder pass with style := context view 0
    (`:211-229`).  `batch["style"]["image"]` is in [0,1] and mapped to [-1,1] for a stylized encoder (`:151-155`)."""

    def __init__(self, encoder: nn.Module, decoder: nn.Module, lr: float = 2e-4, dist=None, bucket_bytes: int = 64 << 20,
                 clip: Optional[float] = 0.5, losses: Optional[Sequence[nn.Module]] = None,
                 identity_loss: Optional[nn.Module] = None, backbone_lr_multiplier: float = 0.1,
                 warm_up_steps: Optional[int] = None, max_steps: int = 100_000, force_collective: bool = False,
                 dp_mode: Optional[str] = None):
        self.encoder, self.decoder, self.clip = encoder, decoder, clip
        self.losses, self.identity_loss = (list(losses) if losses is not None else None), identity_loss
        # identical replicas before anything else looks at the parameters (DDP semantics: rank 0's state wins)
        self.synced_bytes = broadcast_module_state(encoder, dist, force_collective=force_collective)
        new, pre, self.frozen_names = select_trainable(encoder)
        # bucket order = reverse registration order of the trainable parameters (autograd readiness)
        trainable = {id(p) for p in list(new) + list(pre)}
        self.dp_mode = dp_mode or DP_MODE
        self.reducer = BucketedGradReducer([p for p in encoder.parameters() if id(p) in trainable], dist, bucket_bytes,
                                           force_collective=force_collective, mode=self.dp_mode, groups=[list(new), list(pre)])
        self.optimizer = make_optimizer(new, pre, lr, backbone_lr_multiplier, owner=self.reducer if self.dp_mode == "rs_ag" else None)
        if self.dp_mode == "rs_ag":
            self.reducer.guard_readers(encoder)   # validation forwards / checkpoints between two steps wait for the parameter all-gather
        self.scheduler = make_lr_scheduler(self.optimizer, warm_up_steps, max_steps, lr) if warm_up_steps else None
        self._weights = [p for p in encoder.parameters() if id(p) in trainable and p.dim() == 2]
        self.global_step = 0

    def _render(self, ctx, style, tgt, mse_target=None):
        g = self.encoder(ctx, style, self.global_step)
        h, w = tgt["image"].shape[-2:]
        kw = {} if mse_target is None else {"mse_target": mse_target}     # LossMse inside the composite kernels (DecoderOutput.loss_mse)
        return g, self.decoder.forward(g, tgt["extrinsics"], tgt["intrinsics"], tgt["near"], tgt["far"], (h, w), **kw)

    def __call__(self, batch: dict) -> torch.Tensor:
        ctx, tgt = batch["context"], batch["target"]
        if getattr(self.encoder, "stylized", False) and "style" in batch:
            style = {"image": (batch["style"]["image"] - 0.5) / 0.5}     # (0,1) -> (-1,1), `:151-155`
        else:
            style = {"image": ctx["image"][:, 0]}                         # stylized=False: style := context view 0 (`:149-150`)
        self.reducer.wait_params()                                        # "rs_ag": the previous step's parameter all-gather must have landed
        self.reducer.prepare()
        fuse = self.losses is None and tgt["image"].is_cuda and not tgt["image"].requires_grad
        g, out = self._render(ctx, style, tgt, tgt["image"] if fuse else None)
        if self.losses is None:
            total = out.loss_mse if fuse else mse_loss(out.color, tgt["image"])     # LossMse
        else:
            total = sum(fn(out, batch, g, self.global_step) for fn in self.losses)
        if self.identity_loss is not None:
            ig, iout = self._render(ctx, {"image": ctx["image"][:, 0]}, tgt)
            total = total + self.identity_loss(iout, batch, ig, self.global_step)
        total.backward()
        self.reducer.finish()
        if self.clip is not None:                                         # Trainer(gradient_clip_val=0.5), main_style.py:110
            self.reducer.clip_grad_norm_(self.clip, defer_to=self.optimizer)    # the fused AdamW applies the coefficient while it reads the gradients
        self.optimizer.step()
        self.reducer.gather_params()                                      # "rs_ag": asynchronous; fenced at the top of the next step
        if BATCHED_RESPLIT and self.dp_mode != "rs_ag" and self._weights and self._weights[0].is_cuda:
            # every cached split image of every updated weight in ONE launch (vit_ops.refresh_split_cache) instead of ~1 270 lazy ones
            # during the next forward / backward.  ("rs_ag": the weights are complete only when the all-gather has landed.)
            vit_ops.refresh_split_cache([p for p in self._weights if p.grad is not None])
        if self.scheduler is not None:
            self.scheduler.step()
        self.global_step += 1
        return total.detach()
