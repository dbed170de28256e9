#!/usr/bin/env python
"""GPU time of each stage of the serving forward in isolation (graph replays: no host cost): where is the critical path of C2?"""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.graphs import StreamGraphedEncoder
from styl3r_amd.scenes import make_scene
dev = torch.device("cuda:0"); torch.manual_seed(0)
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg()).to(dev).eval()
sc = make_scene(n_ctx=2, grid_hw=(8, 8), n_views=3, image_hw=(256, 256), seed=1234)
g = torch.Generator(dev).manual_seed(1234)
ctx = dict(image=torch.rand(1, 2, 3, 256, 256, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(1, 2, 3, 3).contiguous())
style = dict(image=ctx["image"][:, 0])
ge = StreamGraphedEncoder(enc, ctx, style)
for _ in range(3): ge(ctx, style)
torch.cuda.synchronize()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
main = torch.cuda.current_stream()
def dec_two_streams():
    for g1, g2 in zip(ge.g_d1, ge.g_d2):
        ge.s_dec2.wait_stream(main)
        with torch.cuda.stream(ge.s_dec2): g2.replay()
        g1.replay()
        main.wait_stream(ge.s_dec2)
def dec_in_order():
    for g1, g2 in zip(ge.g_d1, ge.g_d2):
        g1.replay(); g2.replay()
def heads_par():
    for s, g in zip(ge.s_heads, ge.g_heads):
        s.wait_stream(main)
        with torch.cuda.stream(s): g.replay()
    for s in ge.s_heads: main.wait_stream(s)
def heads_seq():
    for g in ge.g_heads: g.replay()
print("backbone encoder (24 blocks, 2 views)   %.2f ms" % t(ge.g_be.replay))
print("style encoder (24 blocks, 1 image)      %.2f ms" % t(ge.g_se.replay))
print("stylizer decoder (12 blocks)            %.2f ms" % t(ge.g_sd.replay))
print("decoder prologue + epilogue             %.2f ms" % (t(ge.g_dpre.replay) + t(ge.g_dpost.replay)))
print("dual decoders, two streams              %.2f ms" % t(dec_two_streams))
print("dual decoders, in order                 %.2f ms" % t(dec_in_order))
print("decoder 1 only                          %.2f ms" % t(lambda: [g.replay() for g in ge.g_d1]))
print("five heads, five streams                %.2f ms" % t(heads_par))
print("five heads, in order                    %.2f ms" % t(heads_seq))
for i, g_ in enumerate(ge.g_heads): print("   head job %d                          %.2f ms" % (i, t(g_.replay)))
print("adapter                                 %.2f ms" % t(ge.g_fin.replay))
print("whole forward (stream graphs)           %.2f ms" % t(lambda: ge(ctx, style)))
