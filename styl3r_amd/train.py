"""Minimal train step over the hot path (encoder -> decoder -> MSE -> backward -> DP all-reduce ->
clip -> AdamW), the part of `ModelWrapperStyle.training_step` / `configure_optimizers`
(src/model/model_wrapper_style.py:118-315, 843-916) that the benchmark and the multi-GPU path need.
Everything else of the LightningModule (logging, video, validation) is out of scope."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .ddp import BucketedGradReducer
from .losses import mse_loss


class TrainStep:
    def __init__(self, encoder: nn.Module, decoder: nn.Module, lr: float = 2e-4, dist=None, bucket_bytes: int = 64 << 20,
                 clip: Optional[float] = 0.5):
        self.encoder, self.decoder, self.clip = encoder, decoder, clip
        params = [p for p in encoder.parameters() if p.requires_grad]
        # model_wrapper_style.py:898 (AdamW, default betas/eps/weight_decay); on a GPU the single-pass fused
        # implementation: the foreach one makes ~10 passes over the 4.2 GB of parameter/moment state per step
        self.optimizer = torch.optim.AdamW(params, lr=lr, fused=bool(params) and params[0].is_cuda)
        self.reducer = BucketedGradReducer(params, dist, bucket_bytes)

    def __call__(self, batch: dict) -> torch.Tensor:
        ctx, tgt = batch["context"], batch["target"]
        style = batch.get("style") or {"image": ctx["image"][:, 0]}      # stylized=False: style := context view 0 (:149-150)
        self.reducer.prepare()
        g = self.encoder(ctx, style, 0)
        h, w = tgt["image"].shape[-2:]
        out = self.decoder.forward(g, tgt["extrinsics"], tgt["intrinsics"], tgt["near"], tgt["far"], (h, w))
        loss = mse_loss(out.color, tgt["image"])                          # LossMse
        loss.backward()
        self.reducer.finish()
        if self.clip is not None:                                         # Trainer(gradient_clip_val=0.5), main_style.py:110
            self.reducer.clip_grad_norm_(self.clip)
        self.optimizer.step()
        return loss.detach()
