"""hipGraph replay of the inference encoder (config C2: one scene, forward only).

At batch 1 the encoder is ~1 900 small launches (40 us of GPU work each on average): the host cannot enqueue them as
fast as the GPU retires them.  `GraphedEncoder` captures one forward pass of `encoder(context, style)` into a hipGraph
(torch.cuda.CUDAGraph: the HIP kernels of libvit_hip.so / libgsr_hip.so are launched on torch's current stream, so
they are captured like torch's own kernels) and replays it with the inputs copied into static buffers.  Shapes are
fixed at capture time; the weights are read in place (the bf16x6 split cache must be warm: one eager pass first)."""
from __future__ import annotations

import torch
from torch import nn

from .decoder import Gaussians


class GraphedEncoder:
    def __init__(self, encoder: nn.Module, context: dict, style: dict, warmup: int = 2):
        self.encoder = encoder.eval()
        self.ctx = {k: v.clone() for k, v in context.items() if torch.is_tensor(v)}
        self.style = {k: v.clone() for k, v in style.items() if torch.is_tensor(v)}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                      # allocator / split-weight cache / MIOpen solver warm-up
                self.encoder(self.ctx, self.style, 0)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self.encoder(self.ctx, self.style, 0)

    @torch.no_grad()
    def __call__(self, context: dict, style: dict) -> Gaussians:
        for k, v in self.ctx.items():
            v.copy_(context[k])
        for k, v in self.style.items():
            v.copy_(style[k])
        self.graph.replay()
        return self.out
