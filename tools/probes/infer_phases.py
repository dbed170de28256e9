#!/usr/bin/env python
"""C2 inference as stream graphs (styl3r_amd.graphs.StreamGraphedEncoder): where the wall time goes, phase by phase -- hipEvents at the segment
boundaries of the replay (the profiler serialises the streams, so a kernel trace cannot show this).
usage: VIT_LINEAR_MODE=f16x3 VIT_ATTENTION=f16x3 python tools/probes/infer_phases.py [steps=20]"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.graphs import StreamGraphedEncoder
from styl3r_amd.scenes import make_scene
from styl3r_amd import vit_ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0"); torch.manual_seed(0)
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg()).to(dev).eval()
H = 256
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=3, image_hw=(H, H), seed=1234)
g = torch.Generator(dev).manual_seed(1234)
ctx = dict(image=torch.rand(1, 2, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(1, 2, 3, 3).contiguous())
style = dict(image=ctx["image"][:, 0])
with torch.no_grad():
    enc(ctx, style, 0)
import os
StreamGraphedEncoder.early_heads = os.environ.get("EARLY_HEADS", "1") == "1"
StreamGraphedEncoder.early_appearance = os.environ.get("EARLY_APP", "1") == "1"
if os.environ.get("PAIR") == "0":
    type(enc.backbone).pair_launches = False
if os.environ.get("HEAD_ORDER"):
    StreamGraphedEncoder.HEAD_ORDER = tuple(int(c) for c in os.environ["HEAD_ORDER"])
ge = StreamGraphedEncoder(enc, ctx, style)
E = lambda: torch.cuda.Event(enable_timing=True)


def run(self, marks):
    main = torch.cuda.current_stream(dev)
    ev = {}
    def mark(name, stream=None):
        e = E(); e.record(stream or main); ev[name] = e
    mark("t0")
    self.s_style.wait_stream(main)
    with torch.cuda.stream(self.s_style):
        self.g_se.replay(); mark("style_encode_end", self.s_style)
    self.g_be.replay(); mark("backbone_encode_end")

    def app_early(i, after):
        if i < len(self.g_app_early) and self.g_app_early[i] is not None:
            self.s_heads[2].wait_stream(after)
            with torch.cuda.stream(self.s_heads[2]):
                self.g_app_early[i].replay()
    app_early(0, main)
    self.s_style.wait_stream(main)
    with torch.cuda.stream(self.s_style):
        self.g_sd_begin.replay()
    for k, g in enumerate(self.g_sd):
        with torch.cuda.stream(self.s_style):
            g.replay()
        if k + 1 < len(self.g_sd):
            app_early(k + 1, self.s_style)
    with torch.cuda.stream(self.s_style):
        self.g_sd_end.replay(); mark("stylizer_end", self.s_style)
    self.g_dpre.replay()
    if getattr(self, "g_dpair", None) is not None:
        def early(i):
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
        self.s_dec2.wait_stream(main)
        with torch.cuda.stream(self.s_dec2):
            g2.replay()
        g1.replay()
        main.wait_stream(self.s_dec2)
    self.g_dpost.replay(); mark("decoders_end")
    main.wait_stream(self.s_style); mark("join_style")
    order = [i for i in self.HEAD_ORDER if i < len(self.g_heads)]
    for i in order:
        s, gg = self.s_heads[i], self.g_heads[i]
        s.wait_stream(main)
        with torch.cuda.stream(s):
            gg.replay(); mark(f"head{i}_end", s)
    for s in self.s_heads[:len(self.g_heads)]:
        main.wait_stream(s)
    mark("heads_joined")
    self.g_fin.replay(); mark("end")
    marks.append(ev)


acc = {}
with torch.no_grad():
    for i in range(steps + 3):
        m = []
        run(ge, m); torch.cuda.synchronize()
        if i >= 3:
            ev = m[0]
            for k, e in ev.items():
                acc[k] = acc.get(k, 0.0) + ev["t0"].elapsed_time(e)
print(json.dumps({"mode": vit_ops.LINEAR_MODE, "small_m_rows": vit_ops.SMALL_M_ROWS, "pair_path": ge.g_dpair is not None, "early_heads": StreamGraphedEncoder.early_heads, "early_app": StreamGraphedEncoder.early_appearance, "order": StreamGraphedEncoder.HEAD_ORDER, "ms_since_start": {k: round(v / steps, 3) for k, v in acc.items()}}))
