"""hipGraph replay of the inference encoder (config C2: one scene, forward only).

At batch 1 the encoder is ~1 900 small launches (40 us of GPU work each on average): the host cannot enqueue them as
fast as the GPU retires them.  `GraphedEncoder` captures one forward pass of `encoder(context, style)` into a hipGraph
(torch.cuda.CUDAGraph: the HIP kernels of libvit_hip.so / libgsr_hip.so are launched on torch's current stream, so
they are captured like torch's own kernels) and replays it with the inputs copied into static buffers.  Shapes are
fixed at capture time; the weights are read in place (the bf16x6 split cache must be warm: one eager pass first)."""
from __future__ import annotations

import torch
from torch import nn

from . import vit_ops
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
        try:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.out = self.encoder(self.ctx, self.style, 0)
        finally:
            # f16x3 |max| words: this graph's arena (zero fill captured with it) is not handed to anyone else -- also when the capture raised
            # (a failed capture left its arena registered, and the next capture on the stream drew words from a dead pool)
            vit_ops._AMAX.end_capture()

    @torch.no_grad()
    def __call__(self, context: dict, style: dict) -> Gaussians:
        for k, v in self.ctx.items():
            v.copy_(context[k])
        for k, v in self.style.items():
            v.copy_(style[k])
        self.graph.replay()
        return self.out


class StreamGraphedEncoder:
    """Serving replay of the style-token encoder as ONE hipGraph PER STREAM SEGMENT.

    At batch 1 the eager forward with `head_streams` is host-bound: Python needs 17 - 19 ms to enqueue ~1 700 launches that the GPU, with the
    independent parts on their own streams, finishes ~3 ms later (tools/probes/infer_host_time.py).  A single captured graph removes the host
    cost but replays in order (29.8 ms), and a single graph with internal fork / join replays slower still (34 ms, r02).  Here every piece that
    runs on one stream between two cross-stream dependencies is its own graph -- backbone encoder | style encoder | stylizer decoder |
    decoder prologue | decoder 1 layer i | decoder 2 layer i | decoder epilogue | five head calls | adapter -- and `__call__` replays them on
    the streams of the eager serving path with the same fork / join waits: ~45 graph launches instead of ~1 700 kernel launches.
    Shapes and the arithmetic mode are fixed at capture time; the weights are read in place (one eager pass first warms the split caches)."""

    HEAD_ORDER = (2, 1, 4, 0, 3)      # launch order of the head jobs: the appearance head (two views per call) and the Gaussian-parameter heads first
    early_heads = True                # the heads' front-end branches start as soon as their decoder layer is done (A/B switch)
    early_appearance = False          # ... the appearance head's too (it hooks into the stylizer's decoder): built, measured, off -- same box, f16x3:
    #                                   none 13.32 / 13.32 ms, decoder-fed heads only 12.97 / 13.09, all five 13.32 / 13.53 (tools/probes/infer_phases.py)

    def __init__(self, encoder: nn.Module, context: dict, style: dict, global_step: int = 0, warmup: int = 2):
        enc = self.encoder = encoder.eval()
        dev = context["image"].device
        self.ctx = {k: v.clone() for k, v in context.items() if torch.is_tensor(v)}
        self.style = {k: v.clone() for k, v in style.items() if torch.is_tensor(v)}
        keep = enc.head_streams
        enc.head_streams = False                              # the pieces are captured in order; the streams are this class's business
        enc.backbone.branch_streams = False
        warm = torch.cuda.Stream(dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm), torch.no_grad():
            for _ in range(warmup):
                enc(self.ctx, self.style, global_step)
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        self.s_style, self.s_dec2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        self.s_heads = [torch.cuda.Stream(dev) for _ in range(5)]
        cap = torch.cuda.Stream(dev)                          # every piece is captured on this stream, replayed wherever it belongs
        self._keep = []

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            box = {}
            try:
                with torch.no_grad(), torch.cuda.graph(g, stream=cap):
                    box["out"] = fn()
            finally:
                vit_ops._AMAX.end_capture()  # every graph zero-fills its own |max| arena on replay; in a finally: a failed capture must not leave its arena registered
            self._keep.append(box)
            return g, box["out"]

        bb, ts = enc.backbone, enc.token_stylizer
        images = self.ctx["image"]
        self.g_be, (enc_feat, enc_pos) = capture(lambda: bb.encode(self.ctx))
        self.g_se, encoded = capture(lambda: ts.encode_style(self.style))
        # the stylizer's decoder, cut where the appearance head hooks into it (as the dual decoders below): the head's front-end branches start
        # on the head's stream as soon as their layer is done
        Ls = len(ts.dec_blocks)
        self.g_sd_begin, sst = capture(lambda: ts.decode_begin(self.style, enc_feat, enc_pos, encoded=encoded))
        app_on = self.early_heads and self.early_appearance
        s_hooks = sorted({h_ for h_ in (Ls * 2 // 4, Ls * 3 // 4) if 0 < h_ < Ls}) if app_on else []
        s_cuts = [0] + s_hooks + [Ls]
        self.g_sd, self.g_app_early, app_pre = [], [], []

        def app_early(hook_index, layer):
            job = enc._head_early_job_appearance(images, hook_index, sst.outs[layer])
            if job is None:
                return None, None
            with torch.autocast("cuda", enabled=False):
                return capture(job)
        if app_on:
            g, r = app_early(0, 0); self.g_app_early.append(g); app_pre.append(r)
        for lo, hi in zip(s_cuts[:-1], s_cuts[1:]):
            g, _ = capture(lambda: ts.decode_layers(sst, lo, hi))
            self.g_sd.append(g)
            if hi < Ls and app_on:
                g, r = app_early(len(app_pre), hi); self.g_app_early.append(g); app_pre.append(r)
        self.g_sd_end, sty_feat = capture(lambda: ts.decode_end(sst))
        self._app_pre = (app_pre + [None] * (4 - len(app_pre))) if (app_pre and all(r is not None for r in app_pre)) else None
        self.g_dpre, st = capture(lambda: bb._decoder_begin(enc_feat, enc_pos))
        self.g_d1, self.g_d2, self.g_dpair, self.g_early = [], [], None, []
        with torch.no_grad():
            pair = bb._decoder_pair_ok(st)
        if pair:
            # two context views: every layer of the two decoders is one sequence of two-problem launches on the main stream, no fork / join per
            # layer.  The decoder is cut where the DPT heads hook into it (dpt_block.py hooks = [0, L/2, 3L/4, L]): branch i of a head's front
            # end (reassemble + layer_rn[i]) needs only hook i, so it is captured on its own and replayed on the head's stream as soon as that
            # layer is done -- the decoder phase is a chain of small launches that leaves most of the chip idle, the heads are what fills it
            L = len(bb.dec_blocks)
            hooks = sorted({h_ for h_ in (L * 2 // 4, L * 3 // 4) if 0 < h_ < L}) if self.early_heads else []
            cuts = [0] + hooks + [L]
            self.g_dpair, self.g_early = [], []

            def early(hook_index, layer):
                a, r = st.outs[layer]
                jobs = enc._head_early_jobs(images, hook_index, a[:, :-1], r[:, :-1])
                row = []
                for job in jobs:
                    if job is None:
                        row.append((None, None))
                    else:
                        with torch.autocast("cuda", enabled=False):
                            row.append(capture(job))
                return row
            pre_rows = [early(0, 0)] if self.early_heads else []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                g, _ = capture(lambda: [bb._decoder_layer_pair(st, i) for i in range(lo, hi)] and None)
                self.g_dpair.append(g)
                if hi < L and self.early_heads:
                    pre_rows.append(early(len(pre_rows), hi))
            self.g_early = [[g for g, _ in row] for row in pre_rows]               # [hook][head job] -> graph or None
            n_jobs = len(pre_rows[0]) if pre_rows else 0
            self._pre = [[pre_rows[i][j][1] for i in range(len(pre_rows))] + [None] * (4 - len(pre_rows)) if any(pre_rows[i][j][0] is not None for i in range(len(pre_rows))) else None
                         for j in range(n_jobs)] if pre_rows else None
        else:
            for i in range(len(bb.dec_blocks)):
                g1, n1 = capture(lambda: bb._decoder_layer(st, i, 1))
                g2, n2 = capture(lambda: bb._decoder_layer(st, i, 2))
                bb._decoder_advance(st, n1, n2)
                self.g_d1.append(g1); self.g_d2.append(g2)
        self.g_dpost, dec_feat = capture(lambda: [(a[:, :-1], r[:, :-1]) for a, r in bb._decoder_end(st)])
        pre = getattr(self, "_pre", None)
        if self._app_pre is not None:
            pre = list(pre) if pre is not None else [None] * 5
            pre[2] = self._app_pre
        jobs = enc._head_jobs(images, dec_feat, sty_feat, pre=pre)
        assert len(jobs) <= len(self.s_heads)
        self.g_heads, res = [], []
        for job in jobs:
            with torch.autocast("cuda", enabled=False):
                g, r = capture(job)
            self.g_heads.append(g); res.append(r)
        self.g_fin, self.out = capture(lambda: enc._adapter(images, res, global_step, None))
        enc.head_streams = keep
        torch.cuda.synchronize(dev)

    @torch.no_grad()
    def __call__(self, context: dict, style: dict) -> Gaussians:
        for k, v in self.ctx.items():
            v.copy_(context[k])
        for k, v in self.style.items():
            v.copy_(style[k])
        main = torch.cuda.current_stream(self.ctx["image"].device)
        self.s_style.wait_stream(main)
        with torch.cuda.stream(self.s_style):
            self.g_se.replay()
        self.g_be.replay()

        def app_early(i, after):                              # front-end branch i of the appearance head (job 2), on its stream
            if i < len(self.g_app_early) and self.g_app_early[i] is not None:
                self.s_heads[2].wait_stream(after)
                with torch.cuda.stream(self.s_heads[2]):
                    self.g_app_early[i].replay()
        app_early(0, main)
        self.s_style.wait_stream(main)                        # the stylizer's decoder reads the backbone's encoder features
        with torch.cuda.stream(self.s_style):
            self.g_sd_begin.replay()
        for k, g in enumerate(self.g_sd):
            with torch.cuda.stream(self.s_style):
                g.replay()
            if k + 1 < len(self.g_sd):
                app_early(k + 1, self.s_style)
        with torch.cuda.stream(self.s_style):
            self.g_sd_end.replay()
        self.g_dpre.replay()
        if self.g_dpair is not None:
            def early(i):                                     # front-end branch i of every head that has one, on the head's own stream
                if i < len(self.g_early):
                    for j in self.HEAD_ORDER:
                        if j < len(self.g_early[i]) and self.g_early[i][j] is not None:
                            self.s_heads[j].wait_stream(main)
                            with torch.cuda.stream(self.s_heads[j]):
                                self.g_early[i][j].replay()
            early(0)
            for k, g in enumerate(self.g_dpair):
                g.replay()
                if k + 1 < len(self.g_dpair):
                    early(k + 1)
        for g1, g2 in zip(self.g_d1, self.g_d2):
            self.s_dec2.wait_stream(main)                     # decoder 2, layer i reads decoder 1's layer i-1 (and its own)
            with torch.cuda.stream(self.s_dec2):
                g2.replay()
            g1.replay()
            main.wait_stream(self.s_dec2)                     # decoder 1, layer i+1 reads decoder 2's layer i
        self.g_dpost.replay()
        main.wait_stream(self.s_style)
        # (the five heads share the chip: the heavy ones -- the Gaussian-parameter heads with their input merger, jobs 1 and 4 -- go first, so the
        #  last stream to finish is not the one that started last with the most work)
        order = [i for i in self.HEAD_ORDER if i < len(self.g_heads)] + [i for i in range(len(self.g_heads)) if i not in self.HEAD_ORDER]
        for i in order:
            s, g = self.s_heads[i], self.g_heads[i]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                g.replay()
        for s in self.s_heads[:len(self.g_heads)]:
            main.wait_stream(s)
        self.g_fin.replay()
        return self.out
