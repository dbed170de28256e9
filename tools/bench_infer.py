#!/usr/bin/env python
"""C2 (BASELINE.json configs[1]): inference latency -- 2 context views 256x256 -> 131 072 Gaussians
(full-size encoder, random init: re10k_2v.ckpt is absent), 3 target views rendered forward only, `no_grad`.
Reports encoder and rasterizer latency separately (hipEvent timing on the current stream).
  python tools/bench_infer.py [--steps 20] [--tiny]"""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from styl3r_amd.decoder import DecoderSplattingCUDACfg, get_decoder
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg
from styl3r_amd.scenes import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--tiny", action="store_true")
ap.add_argument("--streams", action="store_true", help="run the five head calls on their own streams (encoder.head_streams)")
ap.add_argument("--stream-graphs", action="store_true", help="one hipGraph per stream segment, replayed on the serving streams (styl3r_amd.graphs.StreamGraphedEncoder)")
ap.add_argument("--graph", action="store_true", help="replay the encoder forward as one hipGraph (styl3r_amd.graphs.GraphedEncoder)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
tiny = dict(enc_depth=2, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512)) if args.tiny else None
enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=tiny).to(dev).eval()
enc.head_streams = args.streams
dec = get_decoder(DecoderSplattingCUDACfg("splatting_cuda", [0.0, 0.0, 0.0], True)).to(dev)
H, v_ctx, v_tgt = 256, 2, 3
sc = make_scene(n_ctx=v_ctx, grid_hw=(8, 8), n_views=v_tgt, image_hw=(H, H), seed=1234)
g = torch.Generator(dev).manual_seed(1234)
ctx = dict(image=torch.rand(1, v_ctx, 3, H, H, device=dev, generator=g) * 2 - 1, intrinsics=sc.intrinsics[:1].to(dev).expand(1, v_ctx, 3, 3).contiguous())
style = dict(image=ctx["image"][:, 0])
ex = lambda t: t.to(dev)[None].contiguous()
if args.stream_graphs:
    from styl3r_amd.graphs import StreamGraphedEncoder
    with torch.no_grad():
        ref = enc(ctx, style, 0)
    genc = StreamGraphedEncoder(enc, ctx, style)
    got = genc(ctx, style)
    torch.cuda.synchronize()
    assert torch.allclose(got.means, ref.means, rtol=1e-4, atol=1e-5) and torch.allclose(got.covariances, ref.covariances, rtol=1e-3, atol=1e-8), \
        (float((got.means - ref.means).abs().max()), float((got.covariances - ref.covariances).abs().max()))
    run_enc = lambda: genc(ctx, style)
elif args.graph:
    from styl3r_amd.graphs import GraphedEncoder
    with torch.no_grad():
        ref = enc(ctx, style, 0)
    genc = GraphedEncoder(enc, ctx, style)
    got = genc(ctx, style)
    assert torch.allclose(got.means, ref.means, rtol=1e-5, atol=1e-6) and torch.allclose(got.covariances, ref.covariances, rtol=1e-4, atol=1e-9)
    run_enc = lambda: genc(ctx, style)
else:
    run_enc = lambda: enc(ctx, style, 0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t_enc = t_ras = 0.0
with torch.no_grad():
    for i in range(args.warmup + args.steps):
        ev[0].record()
        gs = run_enc()
        ev[1].record()
        out = dec.forward(gs, ex(sc.extrinsics), ex(sc.intrinsics), ex(sc.near), ex(sc.far), (H, H))
        ev[2].record()
        torch.cuda.synchronize()
        if i >= args.warmup:
            t_enc += ev[0].elapsed_time(ev[1]); t_ras += ev[1].elapsed_time(ev[2])
n = args.steps
print(json.dumps({"metric": "C2 inference latency, 2 ctx + 3 tgt views 256x256, forward only", "encoder_ms": round(t_enc / n, 3),
                  "rasterizer_ms": round(t_ras / n, 3), "total_ms": round((t_enc + t_ras) / n, 3),
                  "views_per_s": round(v_tgt * 1e3 * n / (t_enc + t_ras), 2), "gaussians": int(gs.means.shape[1]),
                  "encoder_fwd_TFLOPs_per_s": round(1.3146 / (t_enc / n) * 1e3 / 1e0, 1) if not args.tiny else None,
                  "encoder_launch": "hipGraph per stream segment" if args.stream_graphs else "hipGraph replay" if args.graph else ("eager, heads on 5 streams" if args.streams else "eager"), "dtype": "f32", "data": "synthetic, random-init weights"}))
