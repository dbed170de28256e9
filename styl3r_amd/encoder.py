"""Style-token encoder: the reference's `src/model/encoder` API on the gfx950 ViT kernels.

Own restatement (same parameter names, so reference checkpoints load unchanged) of
  * `AsymmetricCroCoMulti`          src/model/encoder/backbone/backbone_croco_multiview.py:50-227
  * `CroCoNet` trunk                src/model/encoder/backbone/croco/croco.py:21-128
  * `PatchEmbedDust3R`              src/model/encoder/backbone/croco/patch_embed.py:19-29, blocks.py:226-238
  * `TokenStylizer`                 src/model/encoder/token_stylizer/token_stylizer.py:36-154
  * DPT heads (pts3d / gs / sh)     src/model/encoder/heads/{dpt_block,dpt_head,dpt_gs_head,dpt_gs_sh_head}.py
  * `reg_dense_depth('exp')`        src/model/encoder/heads/postprocess.py:22-60
  * `UnifiedGaussianAdapter`        src/model/encoder/common/gaussian_adapter.py:122-153, gaussians.py:8-44
  * `EncoderNoPoSplatMultiTokenStyle.forward`  src/model/encoder/encoder_noposplat_multi_token_style.py:136-251
  * `StructureBuilder`, `EncoderNoPoSplatTokenStyle`   src/model/encoder/token_stylizer/structure_builder.py:36-150,
                                    src/model/encoder/encoder_noposplat_token_style.py:69-295
  * `get_encoder`                   src/model/encoder/__init__.py:20-25
Everything heavy runs on hand-written gfx950 kernels, all fp32-accurate like the reference (heads under
autocast(enabled=False), encoder_noposplat_multi_token_style.py:150): transformer blocks on styl3r_amd.vit (flash attention with
fused RoPE in bf16x6 split arithmetic on the bf16 MFMA -- exact-f32-MFMA kernels behind VIT_ATTENTION=f32 -- bf16x6 Linear layers with bias / GELU / residual epilogues, HIP LayerNorm), the DPT heads entirely on own kernels since
round 3: reassemble stage (1x1 / transposed / strided convolutions on the token grid) and patch embedding as Linear layers, every 3x3 / 1x1
stride-1 convolution on vit_conv_x6_* at every size, x2 resampling on vit_upsample2x_*, the 7x7 input merger via vit_im2col7, the head tails
(ReLU [-> Dropout] -> 1x1 convolution) on vit_head_tail_*, then the Gaussian adapter on vit_adapter_*.  No library convolution or GEMM is left
on the float32 device path (CPU tensors -- the fixture generators' side -- take the framework ops).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Literal, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .decoder import Gaussians
from .vit import Block, DecoderBlock, LayerNorm6, RopeCfg, _linear
from . import vit_ops
from .vit_ops import Conv2dX6, derived_weight as _derived, fused_linear, head_tail, input_merger_upsample_add, relu_dropout, upsample2x

inf = float("inf")

CROCO_PARAMS = {
    # backbone_croco_multiview.py:21-32 / token_stylizer.py croco_params
    "ViTLarge_BaseDecoder": dict(enc_depth=24, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=768, enc_num_heads=16,
                                 dec_num_heads=12, pos_embed="RoPE100", img_size=(512, 512)),
}


# --------------------------------------------------------------------------- config (field names = the YAML spec)
@dataclass
class BackboneCrocoCfg:
    name: Literal["croco", "croco_multi"] = "croco_multi"
    model: str = "ViTLarge_BaseDecoder"
    patch_embed_cls: str = "PatchEmbedDust3R"
    asymmetry_decoder: bool = True
    intrinsics_embed_loc: Literal["encoder", "decoder", "none"] = "encoder"
    intrinsics_embed_degree: int = 4
    intrinsics_embed_type: Literal["pixelwise", "linear", "token"] = "token"


@dataclass
class TokenStylizerCfg:
    model: str = "ViTLarge_BaseDecoder"
    patch_embed_cls: str = "PatchEmbedDust3R"
    pretrained_weights: str = ""


@dataclass
class GaussianAdapterCfg:
    gaussian_scale_min: float = 0.5
    gaussian_scale_max: float = 15.0
    sh_degree: int = 0


@dataclass
class OpacityMappingCfg:
    initial: float = 0.0
    final: float = 0.0
    warm_up: int = 1


@dataclass
class EncoderNoPoSplatTokenStyleCfg:
    name: str = "noposplat_multi_token_style"
    d_feature: int = 128
    num_monocular_samples: int = 32
    backbone: BackboneCrocoCfg = field(default_factory=BackboneCrocoCfg)
    token_stylizer: TokenStylizerCfg = field(default_factory=TokenStylizerCfg)
    gaussian_adapter: GaussianAdapterCfg = field(default_factory=GaussianAdapterCfg)
    opacity_mapping: OpacityMappingCfg = field(default_factory=OpacityMappingCfg)
    apply_bounds_shim: bool = True
    gaussians_per_pixel: int = 1
    num_surfaces: int = 1
    gs_params_head_type: str = "dpt_gs"
    gs_sh_head_type: str = "dpt"
    input_mean: tuple = (0.5, 0.5, 0.5)
    input_std: tuple = (0.5, 0.5, 0.5)
    pretrained_weights: str = ""
    pose_free: bool = True
    stylized: bool = True


class _PackGrad(torch.autograd.Function):
    """Identity whose backward hands a PACKED gradient to the producing convolution.  The heads' outputs are
    consumed through transposed views ("b d h w -> b (h w) d"), so their gradients arrive as non-contiguous
    NHWC-like views; MIOpen sends such tensors to naive_conv_* kernels (hundreds of ms per call on gfx950,
    profiles/r01d_train_step_kernel_stats.md) instead of its MFMA solvers."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


# --------------------------------------------------------------------------- trunk
class PatchEmbedDust3R(nn.Module):
    def __init__(self, img_size, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x: Tensor):
        B, C, H, W = x.shape
        assert H % self.patch_size[0] == 0 and W % self.patch_size[1] == 0, "image size must be a multiple of the patch size"
        ph, pw = self.patch_size
        h, w = H // ph, W // pw
        pos = torch.cartesian_prod(torch.arange(h, device=x.device), torch.arange(w, device=x.device))
        pos = pos.view(1, h * w, 2).expand(B, -1, 2).clone()
        if x.is_cuda and x.dtype == torch.float32 and (C * ph * pw) % 16 == 0:
            # kernel == stride: the convolution is a Linear over the (B h w, C ph pw) patch rows (row order = the weight's (c, i, j)
            # order), on the bf16x6 kernels forward and backward; no library convolution, no NCHW -> token transpose behind it
            rows = x.reshape(B, C, h, ph, w, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * h * w, C * ph * pw)
            tok = _tok_linear(rows, _derived(self.proj.weight, "patch", lambda w_: w_.reshape(self.proj.out_channels, C * ph * pw)), self.proj.bias)
            return tok.reshape(B, h * w, self.proj.out_channels), pos
        x = _PackGrad.apply(self.proj(x))
        return x.flatten(2).transpose(1, 2), pos


class CrocoTrunk(nn.Module):
    """Parameter layout of CroCoNet with RoPE positional embedding (croco.py:21-99)."""

    def __init__(self, enc_depth, dec_depth, enc_embed_dim, dec_embed_dim, enc_num_heads, dec_num_heads, pos_embed,
                 img_size, mlp_ratio=4, norm_im2_in_dec=True, max_pos=64):
        super().__init__()
        assert pos_embed.startswith("RoPE")
        self.rope = RopeCfg(float(pos_embed[len("RoPE"):]), max_pos=max_pos)
        self.patch_embed = PatchEmbedDust3R(img_size, 16, 3, enc_embed_dim)
        self.enc_depth, self.enc_embed_dim = enc_depth, enc_embed_dim
        self.enc_blocks = nn.ModuleList([Block(enc_embed_dim, enc_num_heads, mlp_ratio, qkv_bias=True, norm_layer=LayerNorm6,
                                               rope=self.rope) for _ in range(enc_depth)])
        self.enc_norm = LayerNorm6(enc_embed_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, dec_embed_dim))
        self.dec_depth, self.dec_embed_dim = dec_depth, dec_embed_dim
        self.decoder_embed = nn.Linear(enc_embed_dim, dec_embed_dim, bias=True)
        self.dec_blocks = nn.ModuleList([DecoderBlock(dec_embed_dim, dec_num_heads, mlp_ratio=mlp_ratio, qkv_bias=True,
                                                      norm_layer=LayerNorm6, norm_mem=norm_im2_in_dec, rope=self.rope)
                                         for _ in range(dec_depth)])
        self.dec_norm = LayerNorm6(dec_embed_dim)
        self.depth_mode, self.conf_mode = ("exp", -inf, inf), None

    @property
    def patch_size(self) -> int:
        return 16

    @property
    def d_out(self) -> int:
        return 1024


class AsymmetricCroCoMulti(CrocoTrunk):
    supports_decoder_embedding = False

    def __init__(self, cfg: BackboneCrocoCfg, d_in: int = 3, params: Optional[dict] = None):
        super().__init__(**(params or CROCO_PARAMS[cfg.model]))
        # intrinsics embedding (backbone_croco_multiview.py:59-78,129-135,199-206; backbone_croco.py:69-101,236-263): 'token' (every documented
        # run; the style encoders require it), 'linear' (the embedding added to every patch token), 'pixelwise' (camera-frame ray directions,
        # or their real-SH expansion, as extra IMAGE channels in front of a 3 + d channel patch embed -- loc 'encoder' -- or as extra FEATURE
        # channels in front of decoder_embed -- loc 'decoder', pairwise backbone only), or none at all.
        pr = params or CROCO_PARAMS[cfg.model]
        pix = (cfg.intrinsics_embed_degree + 1) ** 2 if cfg.intrinsics_embed_degree > 0 else 3
        self.intrinsics_embed_degree = cfg.intrinsics_embed_degree
        self.pix_enc_dim = pix if (cfg.intrinsics_embed_loc == "encoder" and cfg.intrinsics_embed_type == "pixelwise") else 0
        self.pix_dec_dim = pix if (cfg.intrinsics_embed_loc == "decoder" and cfg.intrinsics_embed_type == "pixelwise") else 0
        if cfg.intrinsics_embed_loc == "decoder" and not (self.pix_dec_dim and self.supports_decoder_embedding):
            # (the reference builds these modules, but its forward cannot run them: the multi-view trunk never hands the embedding to its
            #  decoder, backbone_croco_multiview.py:217, and a non-pixelwise decoder-side embedding does not fit decoder_embed, backbone_croco.py:99-101)
            raise NotImplementedError("intrinsics_embed_loc='decoder' exists for the pairwise `croco` backbone with intrinsics_embed_type='pixelwise' only")
        if self.pix_enc_dim:
            self.patch_embed = PatchEmbedDust3R(pr["img_size"], 16, 3 + self.pix_enc_dim, pr["enc_embed_dim"])
        if self.pix_dec_dim:
            self.decoder_embed = nn.Linear(pr["enc_embed_dim"] + self.pix_dec_dim, pr["dec_embed_dim"], bias=True)
        self.intrinsics_embed_type = cfg.intrinsics_embed_type if (cfg.intrinsics_embed_loc == "encoder" and cfg.intrinsics_embed_type != "pixelwise") else "none"
        if cfg.asymmetry_decoder:
            self.dec_blocks2 = copy.deepcopy(self.dec_blocks)
        if cfg.intrinsics_embed_type in ("linear", "token"):          # (the reference creates it whatever the location, :77-78)
            self.intrinsic_encoder = nn.Linear(9, 1024)

    def load_state_dict(self, ckpt, **kw):
        ckpt = dict(ckpt)
        if not any(k.startswith("dec_blocks2") for k in ckpt):      # DUSt3R/MASt3R checkpoints (:99-106)
            for k, v in list(ckpt.items()):
                if k.startswith("dec_blocks"):
                    ckpt[k.replace("dec_blocks", "dec_blocks2")] = v
        return super().load_state_dict(ckpt, **kw)

    def _encode_image(self, image: Tensor, intrinsics_token: Optional[Tensor]):
        x, pos = self.patch_embed(image)
        if intrinsics_token is not None and self.intrinsics_embed_type == "linear":
            x = x + intrinsics_token                                     # (:128-129)
        elif intrinsics_token is not None:
            x = torch.cat((x, intrinsics_token), dim=1)
            extra = pos[:, 0:1, :].clone()
            extra[:, :, 0] += pos[:, -1, 0].unsqueeze(-1) + 1            # the token sits at (rows, 0)  (:131-135)
            pos = torch.cat((pos, extra), dim=1)
        for blk in self.enc_blocks:
            x = blk(x, pos)
        return self.enc_norm(x), pos

    @staticmethod
    def _other_views(x: Tensor) -> Tensor:
        """(b,v,l,c) -> (b,v,(v-1)*l,c): for each view the tokens of all OTHER views, in view order (:159-165)."""
        b, v, l, c = x.shape
        out = []
        for i in range(v):
            out.append(torch.cat([x[:, j] for j in range(v) if j != i], dim=1))
        return torch.stack(out, dim=1)

    branch_streams = False   # inference option (set by the encoder's `head_streams`): decoder 2 on its own HIP stream

    def _decoder_split(self, feat: Tensor, pos: Tensor, extra: Optional[Tensor] = None):
        """The dual decoders (:147-188) with view 0 and views 1.. kept as SEPARATE tensors from start to end:
        returns a list of 13 pairs (first (b,l,c), rest (b*(v-1),l,c)).  The reference -- and round 1 of this build --
        re-assembles a (b,v,l,c) tensor after every block and re-slices it for the next one (the "ctx concat" copies
        SURVEY 8a E7 flags: two cats, a stack and four strided-slice copies per layer, forward and backward).  The memory of
        view 0's cross-attention, "all other views in view order", IS the rest tensor viewed as (b, (v-1) l, c); at v = 2
        the memory of the other decoder is the first tensor itself, so the C2 / C3 path copies nothing at all.  For
        v > 2 the memory of view i >= 1 (view 0 followed by the other rest views) is gathered once per layer."""
        st = self._decoder_begin(feat, pos, extra)
        # Serving (`branch_streams`, no-grad, device tensors): within a layer the two decoders only read each other's PREVIOUS outputs, and
        # at batch 1 neither fills the chip -- decoder 2 runs on its own stream, forked and joined once per layer (the critical path of a
        # C2 forward drops from 24 encoder + 24 decoder block-times to 24 + 12).
        if self._decoder_pair_ok(st):
            # Serving at two context views: the two decoders run the same shapes with two weight sets -- each layer is ONE sequence of 14
            # two-problem launches (vit.decoder_blocks_pair) instead of 2 x 14 launches on two streams with a fork and a join per layer
            for i in range(len(self.dec_blocks)):
                self._decoder_layer_pair(st, i)
            return self._decoder_end(st)
        side = None
        if self.branch_streams and st.f1.is_cuda and not torch.is_grad_enabled():
            main = torch.cuda.current_stream(st.f1.device)
            side = self.__dict__.setdefault("_dec2_stream", torch.cuda.Stream(st.f1.device))
        for i in range(len(self.dec_blocks)):
            if side is not None:
                side.wait_stream(main)                                 # both inputs of this layer are complete on `main`
                with torch.cuda.stream(side):
                    n2 = self._decoder_layer(st, i, 2)
                n1 = self._decoder_layer(st, i, 1)
                main.wait_stream(side)
                n2.record_stream(main)                                 # allocated on the side stream, consumed on `main` from here on
            else:
                n1 = self._decoder_layer(st, i, 1)
                n2 = self._decoder_layer(st, i, 2)
            self._decoder_advance(st, n1, n2)
        return self._decoder_end(st)

    # The pieces of `_decoder_split`, also driven one by one by graphs.StreamGraphedEncoder (one hipGraph per piece and stream)
    def _decoder_begin(self, feat: Tensor, pos: Tensor, extra: Optional[Tensor] = None):
        from types import SimpleNamespace
        b, v, l, c = feat.shape
        # (`extra`: the decoder-side pixelwise embedding (b, v, l, d), concatenated in front of decoder_embed only: output 0 stays the bare features)
        if extra is None:
            cur = _linear(self.decoder_embed, feat)
        else:
            x = torch.cat((feat, extra.to(feat.dtype)), dim=-1)
            if x.is_cuda and x.dtype == torch.float32:      # 1024 + d columns: the contraction is zero-padded to a multiple of 16 (as _intrinsics_token)
                pad = (0, (-x.shape[-1]) % 16)
                cur = fused_linear(torch.nn.functional.pad(x, pad), torch.nn.functional.pad(self.decoder_embed.weight, pad), self.decoder_embed.bias)
            else:
                cur = self.decoder_embed(x)
        st = SimpleNamespace(b=b, v=v, l=l, f1=cur[:, 0].contiguous(), f2=cur[:, 1:].reshape(b * (v - 1), l, -1),
                             # (contiguous ONCE here: a strided position view was copied by every attention call of the 24 decoder
                             #  blocks, forward and backward -- ~100 of the step's framework copy launches)
                             p1=pos[:, 0].contiguous(), p2=pos[:, 1:].reshape(b * (v - 1), l, 2).contiguous(),
                             pm1=pos[:, 1:].reshape(b, (v - 1) * l, 2).contiguous(),       # memory positions of view 0
                             outs=[(feat[:, 0], feat[:, 1:].reshape(b * (v - 1), l, c))])
        st.pm2 = self._mem_of_rest(st, st.p1, st.p2, 2).contiguous()
        return st

    @staticmethod
    def _mem_of_rest(st, first, rest, width):                          # (b(v-1), (v-1) l, width): for view i, view 0 then the others
        b, v, l = st.b, st.v, st.l
        if v == 2:
            return first
        r = rest.view(b, v - 1, l, width)
        parts = [torch.cat([first.unsqueeze(1)] + [r[:, j:j + 1] for j in range(v - 1) if j != i - 1], dim=1).reshape(b, (v - 1) * l, width)
                 for i in range(1, v)]
        return torch.stack(parts, dim=1).reshape(b * (v - 1), (v - 1) * l, width)

    def _decoder_layer(self, st, i: int, which: int) -> Tensor:
        """layer i of decoder 1 (view 0) or decoder 2 (views 1..): reads the previous layer's st.f1 / st.f2, returns the new features"""
        if which == 1:
            return self.dec_blocks[i](st.f1, st.f2.view(st.b, (st.v - 1) * st.l, -1), st.p1, st.pm1)[0]
        return self.dec_blocks2[i](st.f2, self._mem_of_rest(st, st.f1, st.f2, st.f1.shape[-1]), st.p2, st.pm2)[0]

    pair_launches = True     # serving option: layer i of both decoders as two-problem launches where the shapes allow it (A/B switch)

    def _decoder_pair_ok(self, st) -> bool:
        """two context views, batch rows equal, serving (no grad), every layer shape served by the small-M kernel"""
        from .vit import decoder_blocks_pair_ok
        if not self.pair_launches or torch.is_grad_enabled() or st.v != 2 or not st.f1.is_cuda or st.f1.shape != st.f2.shape:
            return False
        hit = getattr(st, "pair_ok", None)
        if hit is None:
            x = torch.stack((st.f1, st.f2))
            hit = st.pair_ok = all(decoder_blocks_pair_ok(b1, b2, x) for b1, b2 in zip(self.dec_blocks, self.dec_blocks2))
        return hit

    def _decoder_layer_pair(self, st, i: int):
        """layer i of BOTH decoders (serving, v == 2): st.x (2, b, l, c) holds the stacked features; st.f1 / st.f2 are views of it"""
        from .vit import decoder_blocks_pair
        if getattr(st, "x", None) is None:
            st.x = torch.stack((st.f1, st.f2))
            st.pos2 = torch.cat((st.p1, st.p2), dim=0).contiguous()
            st.mpos2 = torch.cat((st.pm1, st.pm2), dim=0).contiguous()
        st.x = decoder_blocks_pair(self.dec_blocks[i], self.dec_blocks2[i], st.x, st.pos2, st.mpos2)
        self._decoder_advance(st, st.x[0], st.x[1])

    @staticmethod
    def _decoder_advance(st, n1: Tensor, n2: Tensor):
        st.f1, st.f2 = n1, n2
        st.outs.append((n1, n2))

    def _decoder_end(self, st):
        if getattr(st, "x", None) is not None:               # (pair path: the stacked features, one launch)
            y = self.dec_norm(st.x)
            st.outs[-1] = (y[0], y[1])
        else:
            st.outs[-1] = (self.dec_norm(st.f1), self.dec_norm(st.f2))
        return st.outs

    def _decoder(self, feat: Tensor, pos: Tensor):
        """list[13] of (b, v, l, c): the reference's return layout, assembled from the split form"""
        b, v, l, _ = feat.shape
        return [torch.cat((a.unsqueeze(1), r.view(b, v - 1, l, -1)), dim=1) for a, r in self._decoder_split(feat, pos)]

    def _input_images(self, context: dict) -> Tensor:
        """the images the patch embed sees: with the encoder-side pixelwise embedding, 3 + d channels (:199-201)"""
        if not self.pix_enc_dim:
            return context["image"]
        from .camera import intrinsic_embedding
        return torch.cat((context["image"], intrinsic_embedding(context, self.intrinsics_embed_degree).to(context["image"].dtype)), dim=2)

    def encode(self, context: dict):
        """first half of forward(): the 24 encoder blocks over all views -> (feat (b,v,l,c), pos (b,v,l,2))"""
        b, v, _, h, w = context["image"].shape
        images = self._input_images(context).reshape(b * v, -1, h, w)
        token = None
        if self.intrinsics_embed_type != "none":
            token = _intrinsics_token(self.intrinsic_encoder, context["intrinsics"]).reshape(b * v, 1, -1)
        feat, pos = self._encode_image(images, token)
        return feat.view(b, v, feat.shape[1], -1), pos.view(b, v, pos.shape[1], 2)

    def decode(self, feat: Tensor, pos: Tensor):
        """second half: the dual decoders; strips the intrinsics token where there is one (:222-225)"""
        outs = self._decoder(feat, pos)
        return [t[:, :, :-1] for t in outs] if self.intrinsics_embed_type == "token" else outs

    def decode_split(self, feat: Tensor, pos: Tensor, extra: Optional[Tensor] = None):
        """the same as pairs (view 0 (b,l-1,c), views 1.. (b*(v-1),l-1,c)) -- what the per-view-group heads consume, no re-assembly"""
        outs = self._decoder_split(feat, pos, extra)
        return [(a[:, :-1], r[:, :-1]) for a, r in outs] if self.intrinsics_embed_type == "token" else outs

    def forward(self, context: dict):
        b, v, _, h, w = context["image"].shape
        feat, pos = self.encode(context)
        dec_feat = self.decode(feat, pos)
        shape = torch.tensor([h, w]).repeat(b, v, 1)
        return feat, pos, dec_feat, shape, self._input_images(context)     # (the reference returns the images WITH the pixelwise channels, :218)


class AsymmetricCroCo(AsymmetricCroCoMulti):
    """The pairwise backbone `croco` (src/model/encoder/backbone/backbone_croco.py:61-286): two views, view 1 decoded by `dec_blocks`
    against view 2's tokens and view 2 by `dec_blocks2` against view 1's (:196-218) -- the v = 2 case of the multi-view trunk, same
    parameters and state-dict keys -- behind that class's own return convention `(dec1, dec2, shape1, shape2)`: two lists of 13 per-view
    tensors (b, l, c), the intrinsics token stripped (:259-263).  Pinned by tests/golden/backbone_variants.npz (the reference's
    AsymmetricCroCo.forward on the same weights)."""

    supports_decoder_embedding = True

    def forward(self, context: dict, return_views: bool = False):
        b, v, _, h, w = context["image"].shape
        assert v == 2, "the `croco` backbone is the 2-view (pairwise) trunk; `croco_multi` takes any number of views"
        feat, pos = self.encode(context)
        extra = None
        if self.pix_dec_dim:       # one embedding row per token: the ray through the centre of its 16 x 16 patch (:260-263, downsample 16)
            from .camera import intrinsic_embedding
            extra = intrinsic_embedding(context, self.intrinsics_embed_degree, downsample=16, merge_hw=True)
        outs = self.decode_split(feat, pos, extra)
        dec1, dec2 = [a for a, _ in outs], [r for _, r in outs]
        shape = torch.tensor([h, w]).repeat(b, 1)
        if return_views:
            img = self._input_images(context)
            return dec1, dec2, shape, shape, {"img": img[:, 0]}, {"img": img[:, 1]}
        return dec1, dec2, shape, shape


BACKBONES = {"croco": AsymmetricCroCo, "croco_multi": AsymmetricCroCoMulti}


def get_backbone(cfg: BackboneCrocoCfg, d_in: int = 3, params: Optional[dict] = None):
    """src/model/encoder/backbone/__init__.py:13-20 registry"""
    return BACKBONES[cfg.name](cfg, d_in, params)


class TokenStylizer(CrocoTrunk):
    def __init__(self, cfg: TokenStylizerCfg, params: Optional[dict] = None):
        super().__init__(**(params or CROCO_PARAMS[cfg.model]))

    def encode_style(self, style: dict):
        """style image -> (style tokens in decoder width, positions); independent of the content views"""
        x, spos = self.patch_embed(style["image"])
        for blk in self.enc_blocks:
            x = blk(x, spos)
        return _linear(self.decoder_embed, self.enc_norm(x)), spos

    def forward(self, style: dict, content_feat: Tensor, content_pos: Tensor, encoded=None):
        st = self.decode_begin(style, content_feat, content_pos, encoded)
        self.decode_layers(st, 0, len(self.dec_blocks))
        return self.decode_end(st)

    # the pieces of `forward`, also driven one by one by graphs.StreamGraphedEncoder (the appearance head hooks into layers 0, L/2, 3L/4, L)
    def decode_begin(self, style: dict, content_feat: Tensor, content_pos: Tensor, encoded=None):
        from types import SimpleNamespace
        style_feat, spos = encoded if encoded is not None else self.encode_style(style)
        b, v, l, c = content_feat.shape
        cf = _linear(self.decoder_embed, content_feat.reshape(b, v * l, c))
        return SimpleNamespace(b=b, v=v, l=l, style_feat=style_feat, spos=spos.contiguous(), cf=cf,
                               cpos=content_pos.reshape(b, v * l, 2).contiguous(), outs=[content_feat])

    def decode_layers(self, st, lo: int, hi: int):
        for blk in self.dec_blocks[lo:hi]:
            st.cf, _ = blk(st.cf, st.style_feat, st.cpos, st.spos)
            st.outs.append(st.cf.view(st.b, st.v, st.l, -1))

    def decode_end(self, st):
        st.outs[-1] = self.dec_norm(st.cf).view(st.b, st.v, st.l, -1)
        return [t[:, :, :-1] for t in st.outs]                         # drop the last (intrinsics) token per view (:151-152)


class StructureBuilder(nn.Module):
    """`StructureBuilder` (token_stylizer/structure_builder.py:36-150): decoder_embed, then `dec_depth` SELF-attention
    blocks over the concatenated tokens of the two views, dec_norm; returns per view the 13 hooked outputs without the
    trailing intrinsics token.  Parameter names equal the reference's (decoder_embed, dec_blocks.N.*, dec_norm)."""

    def __init__(self, params: Optional[dict] = None, model: str = "ViTLarge_BaseDecoder"):
        super().__init__()
        pr = params or CROCO_PARAMS[model]
        assert pr["pos_embed"].startswith("RoPE")
        self.rope = RopeCfg(float(pr["pos_embed"][len("RoPE"):]), max_pos=64)
        self.enc_embed_dim, self.dec_embed_dim, self.dec_depth = pr["enc_embed_dim"], pr["dec_embed_dim"], pr["dec_depth"]
        self.decoder_embed = nn.Linear(self.enc_embed_dim, self.dec_embed_dim, bias=True)
        self.dec_blocks = nn.ModuleList([Block(self.dec_embed_dim, pr["dec_num_heads"], 4, qkv_bias=True, norm_layer=LayerNorm6,
                                               rope=self.rope) for _ in range(self.dec_depth)])
        self.dec_norm = LayerNorm6(self.dec_embed_dim)
        self.depth_mode, self.conf_mode = ("exp", -inf, inf), None

    def forward(self, feat1: Tensor, pos1: Tensor, feat2: Tensor, pos2: Tensor):
        outs = [(feat1, feat2)]
        x = torch.cat((_linear(self.decoder_embed, feat1), _linear(self.decoder_embed, feat2)), dim=1)
        pos = torch.cat((pos1, pos2), dim=1)
        for blk in self.dec_blocks:
            x = blk(x, pos)
            outs.append(tuple(x.chunk(2, dim=1)))
        outs[-1] = tuple(self.dec_norm(x).chunk(2, dim=1))
        d1, d2 = zip(*outs)
        return [t[:, :-1] for t in d1], [t[:, :-1] for t in d2]

    @property
    def patch_size(self) -> int:
        return 16

    @property
    def d_out(self) -> int:
        return 1024


# --------------------------------------------------------------------------- DPT heads
class _ResidualConvUnit(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.conv1 = Conv2dX6(features, features, 3, 1, 1, bias=True)
        self.conv2 = Conv2dX6(features, features, 3, 1, 1, bias=True)

    def forward(self, x):
        # conv2(relu(conv1(relu(x)))) + x with both ReLUs and the skip add inside the convolution kernels
        return self.conv2.forward_fused(self.conv1.forward_fused(x), x)


class _FusionBlock(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.out_conv = Conv2dX6(features, features, 1, bias=True)
        self.resConfUnit1 = _ResidualConvUnit(features)
        self.resConfUnit2 = _ResidualConvUnit(features)

    def forward(self, x, skip=None):
        if skip is not None:
            x = x + self.resConfUnit1(skip)
        x = self.resConfUnit2(x)
        x = upsample2x(x)
        return self.out_conv(x)


class _Up2(nn.Module):
    def forward(self, x):
        return upsample2x(x)


def _intrinsics_token(layer: nn.Linear, K: Tensor) -> Tensor:
    """`intrinsic_encoder` = Linear(9, 1024) on the flattened intrinsics (backbone_croco_multiview.py:77-78,204-206).  On the device the
    contraction is zero-padded from 9 to 16 so that this layer, too, runs on the fused bf16x6 Linear (forward, dX, dW) -- it was the one
    Linear of the step left on the GEMM library."""
    x = K.flatten(2)
    if x.is_cuda and x.dtype == torch.float32:
        pad = (0, 16 - x.shape[-1])
        return fused_linear(torch.nn.functional.pad(x, pad).reshape(-1, 16), _derived(layer.weight, "pad16", lambda w_: torch.nn.functional.pad(w_, pad)), layer.bias).reshape(*x.shape[:-1], -1)
    return layer(x)


_TAP_INDEX: dict = {}


def _stride2_tap_index(nh: int, nw: int, device) -> Tensor:
    """flat source index into the (nh * nw + 1)-row token grid (last row = zero padding) of tap (di, dj) of output pixel (oy, ox) of a
    3x3 / stride 2 / padding 1 convolution, laid out [(oy, ox), (di, dj)]"""
    key = (nh, nw, str(device))
    if key not in _TAP_INDEX:
        oh, ow = (nh - 1) // 2 + 1, (nw - 1) // 2 + 1
        oy, ox, di, dj = torch.meshgrid(torch.arange(oh), torch.arange(ow), torch.arange(3), torch.arange(3), indexing="ij")
        sy, sx = 2 * oy + di - 1, 2 * ox + dj - 1
        inside = (sy >= 0) & (sy < nh) & (sx >= 0) & (sx < nw)
        _TAP_INDEX[key] = torch.where(inside, sy * nw + sx, torch.full_like(sy, nh * nw)).reshape(-1).to(device)
    return _TAP_INDEX[key]


def _tok_linear(x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    """x . w^T + bias on the fused bf16x6 Linear (device fp32, contraction a multiple of 16), the framework's otherwise"""
    if x.is_cuda and x.dtype == torch.float32 and w.shape[1] % 16 == 0:
        return fused_linear(x, w, bias)
    return torch.nn.functional.linear(x, w, bias)


class DPTAdapter(nn.Module):
    """DPTOutputAdapter_fix of the three head files; `kind` selects the head variant."""

    def __init__(self, num_channels, dim_tokens, hooks, kind: Literal["pts3d", "gs", "sh"], feature_dim=256, last_dim=128,
                 layer_dims=(96, 192, 384, 768)):
        super().__init__()
        self.hooks, self.kind = list(hooks), kind
        sc = nn.Module()
        sc.layer1_rn = Conv2dX6(layer_dims[0], feature_dim, 3, 1, 1, bias=False)
        sc.layer2_rn = Conv2dX6(layer_dims[1], feature_dim, 3, 1, 1, bias=False)
        sc.layer3_rn = Conv2dX6(layer_dims[2], feature_dim, 3, 1, 1, bias=False)
        sc.layer4_rn = Conv2dX6(layer_dims[3], feature_dim, 3, 1, 1, bias=False)
        sc.layer_rn = nn.ModuleList([sc.layer1_rn, sc.layer2_rn, sc.layer3_rn, sc.layer4_rn])   # same tensors, both key sets
        sc.refinenet1, sc.refinenet2 = _FusionBlock(feature_dim), _FusionBlock(feature_dim)
        sc.refinenet3, sc.refinenet4 = _FusionBlock(feature_dim), _FusionBlock(feature_dim)
        self.scratch = sc
        if kind == "pts3d":     # 'regression' head (dpt_block.py:313-321)
            self.head = nn.Sequential(Conv2dX6(feature_dim, feature_dim // 2, 3, 1, 1), _Up2(),
                                      Conv2dX6(feature_dim // 2, last_dim, 3, 1, 1), nn.ReLU(True),
                                      Conv2dX6(last_dim, num_channels, 1))
        else:                   # 'gs_params' head (:332-340)
            self.head = nn.Sequential(Conv2dX6(feature_dim, feature_dim, 3, padding=1, bias=False), nn.Identity(),
                                      nn.ReLU(True), nn.Dropout(0.1, False), Conv2dX6(feature_dim, num_channels, 1))
        d = list(dim_tokens)
        self.act_postprocess = nn.ModuleList([
            nn.Sequential(Conv2dX6(d[0], layer_dims[0], 1), nn.ConvTranspose2d(layer_dims[0], layer_dims[0], 4, 4)),
            nn.Sequential(Conv2dX6(d[1], layer_dims[1], 1), nn.ConvTranspose2d(layer_dims[1], layer_dims[1], 2, 2)),
            nn.Sequential(Conv2dX6(d[2], layer_dims[2], 1)),
            nn.Sequential(Conv2dX6(d[3], layer_dims[3], 1), Conv2dX6(layer_dims[3], layer_dims[3], 3, 2, 1)),
        ])
        if kind == "gs":
            self.input_merger = nn.Sequential(Conv2dX6(3, 256, 7, 1, 3), nn.ReLU())

    def _reassemble(self, i: int, tok: Tensor, nh: int, nw: int) -> Tensor:
        """`act_postprocess[i]` (dpt_block.py:350-419) applied to the hooked tokens (B, nh*nw, C) -> NCHW feature map.  The tokens ARE the
        pixel-major operand of a GEMM, so every layer here is a Linear on the bf16x6 kernels (forward, dX, dW all hand-written) instead
        of a library convolution on a transposed copy: the 1x1 convolution is x . W^T; ConvTranspose2d with kernel == stride (4x4 s4,
        2x2 s2) is x . W'^T with W' (Co s^2, Ci) followed by a pixel shuffle; the 3x3 stride-2 convolution gathers its nine taps from the
        (B, nh, nw, C) token grid (a 9C-wide row per output pixel) and is one more Linear."""
        seq = self.act_postprocess[i]
        B, N, C = tok.shape
        c1 = seq[0]
        y = _tok_linear(tok.reshape(B * N, C), _derived(c1.weight, "1x1", lambda w_: w_.reshape(c1.out_channels, C)), c1.bias)                    # (B N, C1)
        C1 = c1.out_channels
        if i in (0, 1):
            ct = seq[1]
            s_ = ct.kernel_size[0]
            Co = ct.out_channels
            w = _derived(ct.weight, "ct", lambda w_: w_.permute(1, 2, 3, 0).reshape(Co * s_ * s_, C1))       # [(co, di, dj), ci] = W[ci, co, di, dj]
            bias = _derived(ct.bias, "ct_bias", lambda b_: b_.repeat_interleave(s_ * s_))
            z = _tok_linear(y, w, bias)                                                   # (B N, Co s s)
            return z.reshape(B, nh, nw, Co, s_, s_).permute(0, 3, 1, 4, 2, 5).reshape(B, Co, nh * s_, nw * s_)
        if i == 2:
            return y.reshape(B, nh, nw, C1).permute(0, 3, 1, 2).contiguous()
        cv = seq[1]                                                                       # 3x3, stride 2, padding 1
        oh, ow = (nh - 1) // 2 + 1, (nw - 1) // 2 + 1
        # the nine taps of every output pixel with ONE gather (row nh * nw of the padded grid is the zero padding): forward = one index
        # kernel, backward = one index_add -- not nine slices with a fill + copy + add each
        idx = _stride2_tap_index(nh, nw, y.device)                                        # (oh * ow * 9,)
        grid = torch.cat([y.reshape(B, nh * nw, C1), y.new_zeros(B, 1, C1)], dim=1)
        cols = grid[:, idx].reshape(B * oh * ow, 9 * C1)
        w = _derived(cv.weight, "s2", lambda w_: w_.permute(0, 2, 3, 1).reshape(cv.out_channels, 9 * C1))               # [co, (tap, ci)]
        z = _tok_linear(cols, w, cv.bias)
        return z.reshape(B, oh, ow, cv.out_channels).permute(0, 3, 1, 2).contiguous()

    def early(self, i: int, tok: Tensor, image_size) -> Tensor:
        """branch i of the head's front end -- reassemble + layer_rn[i] -- needs only the tokens of hook i: a serving loop can run it as
        soon as that trunk layer is done (graphs.StreamGraphedEncoder), long before the last decoder layer the rest of the head waits for"""
        H, W = image_size
        return self.scratch.layer_rn[i](self._reassemble(i, tok, H // 16, W // 16))

    def forward(self, tokens: list, image_size, imgs: Optional[Tensor] = None, layers: Optional[list] = None) -> Tensor:
        """layers: per hook, the result of `early(i, ...)` computed ahead (None entries are computed here)"""
        H, W = image_size
        layers = [layers[i] if (layers is not None and layers[i] is not None) else self.early(i, tokens[h], image_size)
                  for i, h in enumerate(self.hooks)]
        # (.contiguous(): MIOpen sends non-packed views -- this crop, the per-view image slice -- to naive_conv_* kernels)
        p4 = self.scratch.refinenet4(layers[3])[:, :, :layers[2].shape[2], :layers[2].shape[3]].contiguous()
        p3 = self.scratch.refinenet3(p4, layers[2])
        p2 = self.scratch.refinenet2(p3, layers[1])
        p1 = self.scratch.refinenet1(p2, layers[0])
        if self.kind == "gs":
            merged = input_merger_upsample_add(p1, imgs, self.input_merger[0])          # im2col planes + bf16x6 1x1 conv + fused ReLU / add
            if merged is None:       # (the image itself needs a gradient -- parity tests only -- or a host tensor: the framework's 7x7 convolution)
                vit_ops.CALLS["input_merger_library"] += 1
                merged = upsample2x(p1) + self.input_merger(imgs.contiguous())
            p1 = merged
        elif self.kind == "sh":
            p1 = upsample2x(p1)
        # the head tails -- ReLU [-> Dropout] -> 1x1 convolution to 3 / 8 channels -- are one pass over the activation each way
        # (vit_ops.head_tail); the modules stay in `self.head` for the state_dict keys (head.0 / head.2 / head.4)
        if self.kind == "pts3d":
            h = self.head[2](self.head[1](self.head[0](p1)))
            y = head_tail(h, self.head[4], 0.0, False)
            return _PackGrad.apply(y if y is not None else self.head[4](torch.relu(h)))
        h = self.head[0](p1)
        y = head_tail(h, self.head[4], self.head[3].p, self.training)
        if y is None:   # conv 3x3, ReLU(True) -> Dropout(0.1) in one pass each way (vit_ops.relu_dropout), conv 1x1
            y = self.head[4](relu_dropout(h, self.head[3].p, self.training))
        return _PackGrad.apply(y)


def reg_dense_depth_exp(xyz: Tensor) -> Tensor:
    """mode ('exp', -inf, inf): unit direction x expm1(norm)  (postprocess.py:44-57)."""
    d = xyz.norm(dim=-1, keepdim=True)
    return xyz / d.clip(min=1e-8) * d.expm1()


class PixelwiseTaskWithDPT(nn.Module):
    def __init__(self, num_channels, net: CrocoTrunk, kind):
        super().__init__()
        l2 = net.dec_depth
        assert l2 > 9
        ed, dd = net.enc_embed_dim, net.dec_embed_dim
        self.kind = kind
        self.dpt = DPTAdapter(num_channels, [ed, dd, dd, dd], [0, l2 * 2 // 4, l2 * 3 // 4, l2], "pts3d" if kind == "pts3d_raw" else kind)

    def forward(self, tokens, image_size, imgs=None, raw: bool = False, layers=None):
        """raw=True: the DPT output (B, C, H, W) as it leaves the last convolution -- the fused adapter kernel
        (vit_adapter_fwd) applies reg_dense_depth itself.  layers: DPTAdapter.forward"""
        out = self.dpt(tokens, image_size, imgs, layers=layers)
        if self.kind == "pts3d" and not raw:
            return {"pts3d": reg_dense_depth_exp(out.permute(0, 2, 3, 1))}
        return out


def landscape_mean_head(head, tokens, h: int, w: int) -> Tensor:
    """`transpose_to_landscape(head, activate=True)` (croco/misc.py:71-111) as it behaves with PatchEmbedDust3R:
    a landscape / square batch goes straight through; for a portrait batch the wrapper calls the head with the true
    (h, w) and then TRANSPOSES the result, so the points are flattened in transposed pixel order (the wrapper was
    written for ManyAR_PatchEmbed, which feeds portrait images transposed).  Styl3R's data is square or landscape;
    the portrait branch is mirrored only so that results stay identical to the reference on any input."""
    pts = head(tokens, (h, w))["pts3d"]
    return pts if w >= h else pts.swapaxes(1, 2)


class LinearPts3d(nn.Module):
    """`LinearPts3d` (heads/linear_head.py:12-43): every decoder token emits its 16 x 16 patch of 3-D points through ONE Linear layer
    (on the fused kernels), a pixel shuffle puts them on the image grid, reg_dense_depth('exp') finishes.  has_conf = False as in every config."""

    def __init__(self, net, has_conf: bool = False):
        super().__init__()
        assert not has_conf
        self.patch_size = 16
        self.proj = nn.Linear(net.dec_embed_dim, 3 * self.patch_size ** 2)

    def forward(self, decout, img_shape, imgs=None, raw: bool = False):
        H, W = img_shape
        tokens = decout[-1]
        B = tokens.shape[0]
        feat = _linear(self.proj, tokens)                                             # (B, S, 3 p^2)
        feat = feat.transpose(-1, -2).reshape(B, -1, H // self.patch_size, W // self.patch_size)
        feat = torch.nn.functional.pixel_shuffle(feat, self.patch_size)                # (B, 3, H, W)
        return feat if raw else {"pts3d": reg_dense_depth_exp(feat.permute(0, 2, 3, 1))}


def rearrange_head(feat: Tensor, patch_size: int, H: int, W: int) -> Tensor:
    """encoder_noposplat.py:54-59: per-token (B, S, C p^2) outputs of a linear Gaussian-parameter head -> (B, H W, C)"""
    B = feat.shape[0]
    feat = feat.transpose(-1, -2).reshape(B, -1, H // patch_size, W // patch_size)
    return torch.nn.functional.pixel_shuffle(feat, patch_size).flatten(2).transpose(1, 2)


class _LinearGsHead(nn.Sequential):
    """the 'linear' Gaussian-parameter head of the encoders' set_gs_params_head (encoder_noposplat.py:98-106): nn.Sequential(ReLU, Linear) --
    same state-dict keys (`1.weight`, `1.bias`); device tensors take the fused Linear kernels"""

    def __init__(self, dim: int, out: int):
        super().__init__(nn.ReLU(), nn.Linear(dim, out))

    def forward(self, tokens: Tensor) -> Tensor:
        return _linear(self[1], torch.relu(tokens))


def head_factory(head_type, output_mode, net, has_conf=False, out_nchan=3):
    """heads/__init__.py:13-27, all five branches."""
    assert not has_conf
    if head_type == "linear" and output_mode == "pts3d":
        return LinearPts3d(net, has_conf)
    if head_type == "dpt" and output_mode == "gs_params":
        # create_dpt_head(out_nchan=..., postprocess_func=None) (dpt_head.py:101-119): the 'regression' DPT head, raw channels out
        return PixelwiseTaskWithDPT(out_nchan, net, "pts3d_raw")
    if head_type == "dpt" and output_mode == "pts3d":
        return PixelwiseTaskWithDPT(3, net, "pts3d")
    if head_type == "dpt_gs" and output_mode == "gs_params":
        return PixelwiseTaskWithDPT(out_nchan, net, "gs")
    if head_type == "dpt_gs_sh" and output_mode == "gs_params":
        return PixelwiseTaskWithDPT(out_nchan, net, "sh")
    raise NotImplementedError(f"unexpected {head_type=} and {output_mode=}")


# --------------------------------------------------------------------------- Gaussian adapter
def quaternion_to_matrix(q: Tensor, eps: float = 1e-8) -> Tensor:
    i, j, k, r = torch.unbind(q, dim=-1)                     # xyzw order (gaussians.py:13)
    two_s = 2 / ((q * q).sum(dim=-1) + eps)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(*q.shape[:-1], 3, 3)


def build_covariance(scale: Tensor, rotation_xyzw: Tensor) -> Tensor:
    """Sigma = R diag(s)^2 R^T (gaussians.py:8-44: `R @ S @ S^T @ R^T`), written element-wise: M = R diag(s),
    Sigma_ij = sum_k M_ik M_jk.  The reference's chain of batched 3x3 matmuls lowers to `bmm` on 2.6e5..1e6 tiny
    matrices, which the GEMM library runs as ONE 16x16 tile per matrix: 9 launches = 80 ms of a 460 ms train step at
    8 scenes (profiles/r01h); the broadcast form is a handful of bandwidth-bound element-wise kernels (< 1 ms)."""
    R = quaternion_to_matrix(rotation_xyzw)
    M = R * scale.unsqueeze(-2)                                   # M_ik = R_ik s_k
    return (M.unsqueeze(-2) * M.unsqueeze(-3)).sum(-1)            # (.., i, 1, k) * (.., 1, j, k) -> sum_k


@dataclass
class AdapterGaussians:
    means: Tensor
    covariances: Tensor
    scales: Tensor
    rotations: Tensor
    harmonics: Tensor
    opacities: Tensor


class UnifiedGaussianAdapter(nn.Module):
    def __init__(self, cfg: GaussianAdapterCfg):
        super().__init__()
        self.cfg = cfg
        mask = torch.ones((self.d_sh,), dtype=torch.float32)
        for degree in range(1, cfg.sh_degree + 1):
            mask[degree ** 2:(degree + 1) ** 2] = 0.1 * 0.25 ** degree
        self.register_buffer("sh_mask", mask, persistent=False)

    @property
    def d_sh(self) -> int:
        return (self.cfg.sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_sh

    def forward(self, means, depths, opacities, raw_gaussians, eps: float = 1e-8) -> AdapterGaussians:
        scales, rotations, sh = raw_gaussians.split((3, 4, 3 * self.d_sh), dim=-1)
        scales = (0.001 * F.softplus(scales)).clamp_max(0.3)
        rotations = rotations / (rotations.norm(dim=-1, keepdim=True) + eps)
        sh = sh.reshape(*sh.shape[:-1], 3, self.d_sh)
        sh = sh.broadcast_to((*opacities.shape, 3, self.d_sh)) * self.sh_mask
        return AdapterGaussians(means, build_covariance(scales, rotations), scales,
                                rotations.broadcast_to((*scales.shape[:-1], 4)), sh, opacities)


# --------------------------------------------------------------------------- encoder
class EncoderNoPoSplatMultiTokenStyle(nn.Module):
    def __init__(self, cfg: EncoderNoPoSplatTokenStyleCfg, trunk_params: Optional[dict] = None):
        super().__init__()
        self.cfg = cfg
        assert cfg.pose_free and cfg.num_surfaces == 1
        self.gs_params_head_type = cfg.gs_params_head_type
        self.backbone = AsymmetricCroCoMulti(cfg.backbone, 3, trunk_params)
        assert self.backbone.intrinsics_embed_type == "token", "the Gaussian heads of this encoder take the 'token' intrinsics embedding (every shipped config)"
        self.gaussian_adapter = UnifiedGaussianAdapter(cfg.gaussian_adapter)
        self.patch_size = 16
        self.raw_gs_dim = 1 + self.gaussian_adapter.d_in
        d_sh3 = 3 * self.gaussian_adapter.d_sh
        self.downstream_head1 = head_factory("dpt", "pts3d", self.backbone)
        self.downstream_head2 = head_factory("dpt", "pts3d", self.backbone)
        # set_gs_params_head (encoder_noposplat_multi_token_style.py:91-113): the constructor accepts 'linear' | 'dpt' | 'dpt_gs' (same
        # modules / state-dict keys here); the reference's FORWARD runs 'dpt_gs' only and raises for the others (:161-170) -- so does this one
        if cfg.gs_params_head_type == "linear":
            self.gaussian_param_head = _LinearGsHead(self.backbone.dec_embed_dim, cfg.num_surfaces * self.patch_size ** 2 * self.raw_gs_dim)
            self.gaussian_param_head2 = _LinearGsHead(self.backbone.dec_embed_dim, cfg.num_surfaces * self.patch_size ** 2 * self.raw_gs_dim)
        elif cfg.gs_params_head_type == "dpt":
            self.gaussian_param_head = head_factory("dpt", "gs_params", self.backbone, out_nchan=self.raw_gs_dim)
            self.gaussian_param_head2 = head_factory("dpt", "gs_params", self.backbone, out_nchan=self.raw_gs_dim)
        elif cfg.gs_params_head_type == "dpt_gs":
            self.gaussian_param_head = head_factory("dpt_gs", "gs_params", self.backbone, out_nchan=self.raw_gs_dim - d_sh3)
            self.gaussian_param_head2 = head_factory("dpt_gs", "gs_params", self.backbone, out_nchan=self.raw_gs_dim - d_sh3)
        else:
            raise NotImplementedError(f"unexpected head_type={cfg.gs_params_head_type!r}")
        self.stylized = cfg.stylized
        self.token_stylizer = TokenStylizer(cfg.token_stylizer, trunk_params)
        self.gaussian_appearance_head = head_factory("dpt_gs_sh", "gs_params", self.token_stylizer, out_nchan=d_sh3)

    head_streams = False     # inference option: the style branch next to the backbone and the five head calls, each on its own HIP stream
    fused_adapter = True     # E10-E12 on vit_adapter_fwd / vit_adapter_bwd (device tensors, landscape / square images);
                             # False = the element-wise framework expression of the same math (the kernel's test reference)

    def _run_heads(self, jobs, like: Tensor):
        """The head calls only depend on the trunk outputs.  At serving batch sizes (one scene) none of their kernels fills
        256 CUs, so with `head_streams` (no-grad, device tensors) each call is issued on its own stream between two
        fork / join waits on the caller's stream; otherwise they run in order on the current stream."""
        if not (self.head_streams and like.is_cuda and not torch.is_grad_enabled()):
            return [fn() for fn in jobs]
        cur = torch.cuda.current_stream(like.device)
        pool = self.__dict__.setdefault("_head_stream_pool", [])
        while len(pool) < len(jobs):
            pool.append(torch.cuda.Stream(like.device))
        outs = []
        for fn, st in zip(jobs, pool):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn())
        for o, st in zip(outs, pool):
            cur.wait_stream(st)
            o.record_stream(cur)         # allocated on the side stream, consumed on the caller's
        return outs

    def map_pdf_to_opacity(self, pdf: Tensor, global_step: int) -> Tensor:
        cfg = self.cfg.opacity_mapping
        x = cfg.initial + min(global_step / cfg.warm_up, 1) * (cfg.final - cfg.initial)
        exponent = 2 ** x
        return 0.5 * (1 - (1 - pdf) ** exponent + pdf ** (1 / exponent))

    def forward(self, context: dict, style: dict, global_step: int = 0,
                visualization_dump: Optional[dict] = None) -> Gaussians:
        if self.gs_params_head_type != "dpt_gs":       # encoder_noposplat_multi_token_style.py:161-170
            raise NotImplementedError(f"unexpected self.gs_params_head_type={self.gs_params_head_type!r}")
        b, v, _, h, w = context["image"].shape
        images = context["image"]
        self.backbone.branch_streams = bool(self.head_streams)
        if self.head_streams and images.is_cuda and not torch.is_grad_enabled():
            # Serving: the style image's 24 encoder blocks do not depend on the content views, and the stylizer's decoder only
            # needs the backbone's ENCODER features -- at batch 1 none of these kernels fills the chip, so the style branch
            # runs on a side stream next to the backbone (fork at entry, join before the heads).
            cur = torch.cuda.current_stream(images.device)
            side = self.__dict__.setdefault("_style_stream", torch.cuda.Stream(images.device))
            side.wait_stream(cur)
            # (the host needs ~17 ms to enqueue a forward the GPU finishes ~3 ms later; enqueueing the backbone before the style encoder
            # instead was measured: no difference beyond the run-to-run noise)
            with torch.cuda.stream(side):
                encoded = self.token_stylizer.encode_style(style)
            enc_feat, enc_pos = self.backbone.encode(context)
            side.wait_stream(cur)                                   # the encoder features are ready on `cur` from here on
            with torch.cuda.stream(side):
                sty_feat = self.token_stylizer(style, enc_feat, enc_pos, encoded=encoded)
            dec_feat = self.backbone.decode_split(enc_feat, enc_pos)
            cur.wait_stream(side)
            for t in sty_feat:
                t.record_stream(cur)
        else:
            enc_feat, enc_pos = self.backbone.encode(context)
            dec_feat = self.backbone.decode_split(enc_feat, enc_pos)
            sty_feat = self.token_stylizer(style, enc_feat, enc_pos)

        return self._heads_and_adapter(images, dec_feat, sty_feat, global_step, visualization_dump, self._run_heads)

    def _opacity_exponent(self, global_step: int) -> float:
        x_op = self.cfg.opacity_mapping
        return 2 ** (x_op.initial + min(global_step / x_op.warm_up, 1) * (x_op.final - x_op.initial))

    def _head_early_jobs(self, images: Tensor, hook_index: int, first: Tensor, rest: Optional[Tensor]):
        """Serving: branch `hook_index` (0 .. 2) of the front end of the four heads that read the decoders' hooked features, as closures in the
        order of `_head_jobs` (entries 0, 1, 3, 4; None where a head is not a DPT head).  first / rest: the hooked features of view 0 / views 1..
        WITHOUT the trailing intrinsics token, as `_head_jobs` gets them."""
        b, v, _, h, w = images.shape
        def job(head, tok):
            dpt = getattr(head, "dpt", None)
            return (lambda: dpt.early(hook_index, tok.float(), (h, w))) if dpt is not None else None
        jobs = [job(self.downstream_head1, first), job(self.gaussian_param_head, first), None]
        if v > 1:
            jobs += [job(self.downstream_head2, rest), job(self.gaussian_param_head2, rest)]
        return jobs

    def _head_early_job_appearance(self, images: Tensor, hook_index: int, tok: Tensor):
        """the same for the appearance head (job 2 of `_head_jobs`): tok = the stylizer's hooked features (b, v, l, c) WITH the intrinsics token"""
        b, v, _, h, w = images.shape
        dpt = getattr(self.gaussian_appearance_head, "dpt", None)
        return (lambda: dpt.early(hook_index, tok[:, :, :-1].flatten(0, 1).float(), (h, w))) if dpt is not None else None

    def _head_jobs(self, images: Tensor, dec_feat, sty_feat, pre=None):
        """The head calls as a list of closures (they only depend on the trunk outputs).  The reference calls a head once per view
        (encoder_noposplat_multi_token_style.py:152-177: head1 for view 0, head2 for every other view, the appearance head for each view).
        The heads act on every sample independently, so the views that share a head go through it as ONE batch of b * (#views) samples:
        same results, a third of the launches at v = 4, and small-resolution layers that fill more of the chip."""
        b, v, _, h, w = images.shape
        fused = self.fused_adapter and images.is_cuda and w >= h
        # pre[j]: per head j, the front-end branches a serving loop computed ahead (`_head_early_jobs`), or None
        lay = lambda j: ({"layers": pre[j]} if (pre is not None and pre[j] is not None) else {})
        if fused:
            mean_head = lambda head, toks, j=None: head(toks, (h, w), raw=True, **(lay(j) if j is not None else {}))       # (B, 3, h, w), reg_dense_depth in the kernel
        else:
            mean_head = lambda head, toks, j=None: landscape_mean_head(head, toks, h, w)
        # (evaluated inside the job: at b > 1 this slice is a copy, and a job may be captured once and replayed on new images)
        rest_images = lambda: images[:, 1:].reshape(b * (v - 1), *images.shape[2:])
        jobs = [lambda: mean_head(self.downstream_head1, [a.float() for a, _ in dec_feat], 0),
                lambda: self.gaussian_param_head([a.float() for a, _ in dec_feat], (h, w), images[:, 0, :3], **lay(1)),
                lambda: self.gaussian_appearance_head([t.flatten(0, 1).float() for t in sty_feat], (h, w))]
        if v > 1:
            jobs += [lambda: mean_head(self.downstream_head2, [r.float() for _, r in dec_feat], 3),
                     lambda: self.gaussian_param_head2([r.float() for _, r in dec_feat], (h, w), rest_images()[:, :3], **lay(4))]
        return jobs

    def _heads_and_adapter(self, images: Tensor, dec_feat, sty_feat, global_step: int, visualization_dump, run_heads) -> Gaussians:
        b, v, _, h, w = images.shape
        with torch.autocast("cuda", enabled=False):
            res = run_heads(self._head_jobs(images, dec_feat, sty_feat), images)
            return self._adapter(images, res, global_step, visualization_dump)

    def _adapter(self, images: Tensor, res, global_step: int, visualization_dump) -> Gaussians:
        """E10-E12: head outputs -> Gaussians"""
        b, v, _, h, w = images.shape
        exponent = self._opacity_exponent(global_step)
        fused = self.fused_adapter and images.is_cuda and w >= h

        def per_view(first, others):                   # -> (b, v, ...)
            if others is None:
                return first.unsqueeze(1)
            return torch.cat((first.unsqueeze(1), others.reshape(b, v - 1, *others.shape[1:])), dim=1)

        with torch.autocast("cuda", enabled=False):
            pts_0, par_0, app = res[:3]
            if fused:
                # E10-E12 in one kernel each way: head outputs (NCHW) -> Gaussians in the rasterizer's layout
                from .vit_ops import gaussian_adapter_hip
                out = gaussian_adapter_hip(pts_0, res[3] if v > 1 else None, par_0, res[4] if v > 1 else None, app,
                                           self.gaussian_adapter.sh_mask, exponent, v, visualization_dump is not None)
                means, cov, sh, opac = out[:4]
                if visualization_dump is not None:
                    visualization_dump["depth"] = means[..., 2].reshape(b, v, h, w, 1, 1)
                    visualization_dump["scales"] = out[4]
                    visualization_dump["rotations"] = out[5]
                    visualization_dump["means"] = means.reshape(b, v, h, w, 1, 3)
                    visualization_dump["opacities"] = opac.reshape(b, v, h, w, 1, 1)
                return Gaussians(means, cov, sh, opac)
            pts_r, par_r = (res[3], res[4].flatten(2).transpose(1, 2)) if v > 1 else (None, None)
            pts = per_view(pts_0, pts_r)
            params = per_view(par_0.flatten(2).transpose(1, 2), par_r)
            appearance = app.flatten(2).transpose(1, 2).reshape(b, v, h * w, -1)

        pts_all = pts.reshape(b, v, h * w, 1, 3)                                  # (b v r srf xyz)
        depths = pts_all[..., -1].unsqueeze(-1)
        d_sh3 = 3 * self.gaussian_adapter.d_sh
        raw = torch.cat((params[..., :self.raw_gs_dim - d_sh3], appearance), dim=-1)
        raw = raw.reshape(b, v, h * w, 1, -1)                                     # (b v r srf c)
        densities = raw[..., 0].sigmoid().unsqueeze(-1)
        g = self.gaussian_adapter(pts_all.unsqueeze(-2), depths, self.map_pdf_to_opacity(densities, global_step),
                                  raw[..., 1:].unsqueeze(-2))
        if visualization_dump is not None:
            visualization_dump["depth"] = depths.reshape(b, v, h, w, 1, 1)
            visualization_dump["scales"] = g.scales.reshape(b, -1, 3)
            visualization_dump["rotations"] = g.rotations.reshape(b, -1, 4)
            visualization_dump["means"] = g.means.reshape(b, v, h, w, 1, 3)
            visualization_dump["opacities"] = g.opacities.reshape(b, v, h, w, 1, 1)
        return Gaussians(g.means.reshape(b, -1, 3), g.covariances.reshape(b, -1, 3, 3),
                         g.harmonics.reshape(b, -1, 3, self.gaussian_adapter.d_sh), g.opacities.reshape(b, -1))

    def get_data_shim(self):
        mean = torch.tensor(self.cfg.input_mean).view(1, 1, 3, 1, 1)
        std = torch.tensor(self.cfg.input_std).view(1, 1, 3, 1, 1)

        def data_shim(batch):      # apply_normalize_shim (src/dataset/shims/normalize_shim.py:21-27)
            for key in ("context", "target"):
                if key in batch and "image" in batch[key]:
                    img = batch[key]["image"]
                    batch[key] = {**batch[key], "image": (img - mean.to(img)) / std.to(img)}
            return batch
        return data_shim


@dataclass
class EncoderNoPoSplatCfg:
    """encoder_noposplat.py:41-57 (the non-style variant of config/main.yaml and re10k_dl3dv_512x512.yaml)."""
    name: str = "noposplat_multi"
    d_feature: int = 128
    num_monocular_samples: int = 32
    backbone: BackboneCrocoCfg = field(default_factory=BackboneCrocoCfg)
    gaussian_adapter: GaussianAdapterCfg = field(default_factory=GaussianAdapterCfg)
    opacity_mapping: OpacityMappingCfg = field(default_factory=OpacityMappingCfg)
    apply_bounds_shim: bool = True
    gaussians_per_pixel: int = 1
    num_surfaces: int = 1
    gs_params_head_type: str = "dpt_gs"
    input_mean: tuple = (0.5, 0.5, 0.5)
    input_std: tuple = (0.5, 0.5, 0.5)
    pretrained_weights: str = ""
    pose_free: bool = True


class EncoderNoPoSplatMulti(EncoderNoPoSplatMultiTokenStyle):
    """`EncoderNoPoSplatMulti` / `EncoderNoPoSplat` (encoder_noposplat_multi.py:125-215, encoder_noposplat.py:139-235):
    no style branch, ONE dpt_gs head per view group emits all 1 + 7 + 3 d_sh raw channels, and the call signature
    has no `style` argument (driven by src/main.py / ModelWrapper).  The 2-view `noposplat` encoder is the v = 2 case
    of the same computation (its AsymmetricCroCo backbone has the identical parameter layout)."""

    def __init__(self, cfg: EncoderNoPoSplatCfg, trunk_params: Optional[dict] = None):
        nn.Module.__init__(self)
        self.cfg = cfg
        assert cfg.pose_free and cfg.num_surfaces == 1
        self.gs_params_head_type = cfg.gs_params_head_type
        self.backbone = AsymmetricCroCoMulti(cfg.backbone, 3, trunk_params)
        assert self.backbone.intrinsics_embed_type == "token", "the Gaussian heads of this encoder take the 'token' intrinsics embedding (every shipped config)"
        self.gaussian_adapter = UnifiedGaussianAdapter(cfg.gaussian_adapter)
        self.patch_size = 16
        self.raw_gs_dim = 1 + self.gaussian_adapter.d_in
        self.downstream_head1 = head_factory("dpt", "pts3d", self.backbone)
        self.downstream_head2 = head_factory("dpt", "pts3d", self.backbone)
        # set_gs_params_head (encoder_noposplat.py:97-116): 'linear' | 'dpt' | 'dpt_gs'.  (The reference's multi-view forward only runs
        # 'dpt_gs', its 2-view `noposplat` forward all three, :155-167; this class serves both registry names and runs all three at any v.)
        if cfg.gs_params_head_type == "linear":
            self.gaussian_param_head = _LinearGsHead(self.backbone.dec_embed_dim, cfg.num_surfaces * self.patch_size ** 2 * self.raw_gs_dim)
            self.gaussian_param_head2 = _LinearGsHead(self.backbone.dec_embed_dim, cfg.num_surfaces * self.patch_size ** 2 * self.raw_gs_dim)
        elif cfg.gs_params_head_type in ("dpt", "dpt_gs"):
            self.gaussian_param_head = head_factory(cfg.gs_params_head_type, "gs_params", self.backbone, out_nchan=self.raw_gs_dim)
            self.gaussian_param_head2 = head_factory(cfg.gs_params_head_type, "gs_params", self.backbone, out_nchan=self.raw_gs_dim)
        else:
            raise NotImplementedError(f"unexpected head_type={cfg.gs_params_head_type!r}")

    def forward(self, context: dict, global_step: int = 0, visualization_dump: Optional[dict] = None) -> Gaussians:
        b, v, _, h, w = context["image"].shape
        _, _, dec_feat, _, images = self.backbone(context)
        with torch.autocast("cuda", enabled=False):
            pts, params = [], []
            for i in range(v):
                head = self.downstream_head1 if i == 0 else self.downstream_head2
                pts.append(landscape_mean_head(head, [t[:, i].float() for t in dec_feat], h, w))
            for i in range(v):
                head = self.gaussian_param_head if i == 0 else self.gaussian_param_head2
                toks = [t[:, i].float() for t in dec_feat]
                if self.gs_params_head_type == "linear":        # per-token Linear + pixel shuffle (encoder_noposplat.py:155-157)
                    params.append(rearrange_head(head(toks[-1]), self.patch_size, h, w))
                elif self.gs_params_head_type == "dpt":         # plain DPT regression head, no image / point input (:158-162)
                    params.append(head(toks, (h, w)).flatten(2).transpose(1, 2))
                else:
                    params.append(head(toks, (h, w), images[:, i, :3]).flatten(2).transpose(1, 2))
        pts_all = torch.stack(pts, dim=1).reshape(b, v, h * w, 1, 3)
        depths = pts_all[..., -1].unsqueeze(-1)
        raw = torch.stack(params, dim=1).reshape(b, v, h * w, 1, -1)
        densities = raw[..., 0].sigmoid().unsqueeze(-1)
        g = self.gaussian_adapter(pts_all.unsqueeze(-2), depths, self.map_pdf_to_opacity(densities, global_step),
                                  raw[..., 1:].unsqueeze(-2))
        if visualization_dump is not None:
            visualization_dump["depth"] = depths.reshape(b, v, h, w, 1, 1)
            visualization_dump["scales"] = g.scales.reshape(b, -1, 3)
            visualization_dump["rotations"] = g.rotations.reshape(b, -1, 4)
            visualization_dump["means"] = g.means.reshape(b, v, h, w, 1, 3)
            visualization_dump["opacities"] = g.opacities.reshape(b, v, h, w, 1, 1)
        return Gaussians(g.means.reshape(b, -1, 3), g.covariances.reshape(b, -1, 3, 3),
                         g.harmonics.reshape(b, -1, 3, self.gaussian_adapter.d_sh), g.opacities.reshape(b, -1))


class EncoderNoPoSplatTokenStyle(EncoderNoPoSplatMultiTokenStyle):
    """`EncoderNoPoSplatTokenStyle` (encoder_noposplat_token_style.py:69-295), the 2-view style encoder of the registry entry
    `noposplat_token_style`: the CroCo encoder trunk, a `StructureBuilder` (self-attention over both views) feeding ONE mean
    head and ONE `gaussian_structure_head` shared by the two views, and the `TokenStylizer` feeding the appearance head.
    Same modules / state-dict keys as the reference's constructor (pinned: tests/test_encoder.py).  The reference's forward
    is stale against its own current modules -- it unpacks eight values from a backbone that returns four and calls the
    token stylizer with a five-argument signature that no longer exists (`:163,188`), and every documented run overrides
    `model.encoder.name=noposplat_multi_token_style` -- so the forward here is that method restated on the CURRENT
    interfaces: encoder features (with the intrinsics token) -> structure tokens / stylized tokens -> heads -> adapter."""

    def __init__(self, cfg: EncoderNoPoSplatTokenStyleCfg, trunk_params: Optional[dict] = None):
        nn.Module.__init__(self)
        self.cfg = cfg
        assert cfg.pose_free and cfg.gs_params_head_type == "dpt_gs" and cfg.num_surfaces == 1 and cfg.gs_sh_head_type == "dpt"
        self.backbone = AsymmetricCroCoMulti(cfg.backbone, 3, trunk_params)     # `croco`: same parameter layout, v = 2
        assert self.backbone.intrinsics_embed_type == "token", "the Gaussian heads of this encoder take the 'token' intrinsics embedding (every shipped config)"
        self.gaussian_adapter = UnifiedGaussianAdapter(cfg.gaussian_adapter)
        self.patch_size = 16
        self.raw_gs_dim = 1 + self.gaussian_adapter.d_in
        d_sh3 = 3 * self.gaussian_adapter.d_sh
        self.stylized = cfg.stylized
        self.structure_builder = StructureBuilder(trunk_params)
        self.token_stylizer = TokenStylizer(cfg.token_stylizer, trunk_params)
        self.downstream_head1 = head_factory("dpt", "pts3d", self.structure_builder)
        self.gaussian_structure_head = head_factory("dpt_gs_sh", "gs_params", self.structure_builder, out_nchan=self.raw_gs_dim - d_sh3)
        self.gaussian_appearance_head = head_factory("dpt_gs_sh", "gs_params", self.token_stylizer, out_nchan=d_sh3)

    def forward(self, context: dict, style: dict, global_step: int = 0, visualization_dump: Optional[dict] = None) -> Gaussians:
        b, v, _, h, w = context["image"].shape
        assert v == 2, "noposplat_token_style is the 2-view encoder"
        bb = self.backbone
        images = context["image"].reshape(b * v, -1, h, w)
        token = _intrinsics_token(bb.intrinsic_encoder, context["intrinsics"]).reshape(b * v, 1, -1)
        feat, pos = bb._encode_image(images, token)
        feat, pos = feat.view(b, v, feat.shape[1], -1), pos.view(b, v, pos.shape[1], 2)
        st1, st2 = self.structure_builder(feat[:, 0], pos[:, 0], feat[:, 1], pos[:, 1])
        sty = self.token_stylizer(style, feat, pos)
        x_op = self.cfg.opacity_mapping
        exponent = 2 ** (x_op.initial + min(global_step / x_op.warm_up, 1) * (x_op.final - x_op.initial))
        with torch.autocast("cuda", enabled=False):
            both = [torch.cat((a, c), dim=0).float() for a, c in zip(st1, st2)]       # the shared heads see both views as one batch
            fused = self.fused_adapter and images.is_cuda and w >= h
            pts = self.downstream_head1(both, (h, w), raw=True) if fused else landscape_mean_head(self.downstream_head1, both, h, w)
            par = self.gaussian_structure_head(both, (h, w))
            app = self.gaussian_appearance_head([t.flatten(0, 1).float() for t in sty], (h, w))       # (b*v, 3 d_sh, h, w), b-major
            if fused:
                from .vit_ops import gaussian_adapter_hip
                out = gaussian_adapter_hip(pts[:b], pts[b:], par[:b], par[b:], app, self.gaussian_adapter.sh_mask, exponent, v,
                                           visualization_dump is not None)
                means, cov, sh, opac = out[:4]
                if visualization_dump is not None:
                    visualization_dump.update(depth=means[..., 2].reshape(b, v, h, w, 1, 1), scales=out[4], rotations=out[5],
                                              means=means.reshape(b, v, h, w, 1, 3), opacities=opac.reshape(b, v, h, w, 1, 1))
                return Gaussians(means, cov, sh, opac)
            pts_all = torch.stack((pts[:b], pts[b:]), dim=1).reshape(b, v, h * w, 1, 3)
            params = torch.stack((par[:b], par[b:]), dim=1).flatten(3).transpose(2, 3)
            appearance = app.flatten(2).transpose(1, 2).reshape(b, v, h * w, -1)
        depths = pts_all[..., -1].unsqueeze(-1)
        raw = torch.cat((params, appearance), dim=-1).reshape(b, v, h * w, 1, -1)
        densities = raw[..., 0].sigmoid().unsqueeze(-1)
        g = self.gaussian_adapter(pts_all.unsqueeze(-2), depths, self.map_pdf_to_opacity(densities, global_step), raw[..., 1:].unsqueeze(-2))
        if visualization_dump is not None:
            visualization_dump.update(depth=depths.reshape(b, v, h, w, 1, 1), scales=g.scales.reshape(b, -1, 3),
                                      rotations=g.rotations.reshape(b, -1, 4), means=g.means.reshape(b, v, h, w, 1, 3),
                                      opacities=g.opacities.reshape(b, v, h, w, 1, 1))
        return Gaussians(g.means.reshape(b, -1, 3), g.covariances.reshape(b, -1, 3, 3),
                         g.harmonics.reshape(b, -1, 3, self.gaussian_adapter.d_sh), g.opacities.reshape(b, -1))


ENCODERS = {"noposplat_multi_token_style": EncoderNoPoSplatMultiTokenStyle, "noposplat_multi": EncoderNoPoSplatMulti,
            "noposplat": EncoderNoPoSplatMulti, "noposplat_token_style": EncoderNoPoSplatTokenStyle}


def get_encoder(cfg: EncoderNoPoSplatTokenStyleCfg):
    """src/model/encoder/__init__.py:20-25: returns (encoder, visualizer=None)."""
    return ENCODERS[cfg.name](cfg), None
