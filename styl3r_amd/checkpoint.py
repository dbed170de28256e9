"""Checkpoint ingestion: MASt3R `.pth` / NoPoSplat / Styl3R wrapper `.ckpt` -> encoder.

Mirrors
  * `checkpoint_filter_fn`  src/misc/weight_modify.py:144-197  (MASt3R 'model' dict -> encoder keys)
  * the loading branches    src/main_style.py:128-168          ('model' | 'state_dict', token-stylizer init)
  * wrapper checkpoints     infer_model_re10k.py:300-306       (strip the `encoder.` prefix)
The blobs themselves are absent (.MISSING_LARGE_BLOBS); tests/test_checkpoint.py drives these functions with
synthetic state dicts of the real key layout.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import Tensor, nn


def _adapt_input_conv(in_chans: int, w: Tensor) -> Tensor:
    """first-conv adaptation for a different number of input channels (weight_modify.py:85-114)."""
    dt = w.dtype
    w = w.float()
    O, I, J, K = w.shape
    if in_chans == 1:
        w = w.sum(dim=1, keepdim=True)
    elif in_chans != 3:
        if I != 3:
            raise NotImplementedError("Weight format not supported by conversion.")
        rep = -(-in_chans // 3)
        w = w.repeat(1, rep, 1, 1)[:, :in_chans] * (3 / float(in_chans))
    return w.to(dt)


def _adapt_linear(w: Tensor) -> Tensor:
    """decoder_embed with extra per-token embedding inputs (weight_modify.py:130-141)."""
    dt = w.dtype
    w = w.float()
    extra = torch.cat([c.mean(dim=1, keepdim=True) for c in torch.tensor_split(w, 81, dim=1)], dim=1)
    return torch.cat([w * 0.5, extra * 0.5], dim=1).to(dt)


def convert_mast3r_state_dict(state_dict: Dict[str, Tensor], encoder: nn.Module) -> Dict[str, Tensor]:
    """`checkpoint_filter_fn`: backbone keys get the `backbone.` prefix, the DPT mean heads keep their names and
    lose the confidence channel (head.4 -> first 3 outputs)."""
    out = {}
    for k, v in state_dict.items():
        if "patch_embed.proj.weight" in k:
            O, I, H, W = encoder.backbone.patch_embed.proj.weight.shape
            if v.dim() < 4:
                v = v.reshape(O, -1, H, W)
            if v.shape[-1] != W or v.shape[-2] != H:
                raise NotImplementedError("patch-embed resampling (timm resample_patch_embed) is not built: same 16x16 patches only")
            if v.shape[1] != I:
                v = _adapt_input_conv(I, v)
        elif "decoder_embed.weight" in k:
            O, I = encoder.backbone.decoder_embed.weight.shape
            if v.shape[1] != I:
                v = _adapt_linear(v)
        out[k] = v
    # DUSt3R-style checkpoints without a second decoder: duplicate dec_blocks (what AsymmetricCroCoMulti.load_state_dict,
    # backbone_croco_multiview.py:99-106, does when called directly; loading through the encoder bypasses it)
    if not any(k.startswith("dec_blocks2") for k in out):
        for k, v in list(out.items()):
            if k.startswith("dec_blocks"):
                out[k.replace("dec_blocks", "dec_blocks2")] = v
    out = {(k if "downstream_head" in k else "backbone." + k): v for k, v in out.items()}
    for h in ("downstream_head1", "downstream_head2"):
        for p in ("weight", "bias"):
            key = f"{h}.dpt.head.4.{p}"
            if key in out:
                out[key] = out[key][0:3]
    return out


def load_pretrained_encoder(encoder: nn.Module, ckpt: dict) -> Tuple[list, list]:
    """main_style.py:128-154: `{'model': ...}` (MASt3R) or `{'state_dict': ...}` (NoPoSplat / Styl3R wrapper)."""
    if "model" in ckpt:
        return tuple(encoder.load_state_dict(convert_mast3r_state_dict(ckpt["model"], encoder), strict=False))
    if "state_dict" in ckpt:
        sd = {k[len("encoder."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("encoder.")}
        # NoPoSplat's single gs head carries opacity + scale + rotation AND 3*d_sh SH channels in head.4; here the structure
        # rows stay in gaussian_param_head{,2} ([:-3 d_sh], main_style.py:144-146) and the trailing 3*d_sh rows seed the
        # appearance head (main_style.py:148-150).  The rows are split BEFORE load_state_dict: strict=False does not
        # forgive a size mismatch.
        d3 = 3 * encoder.gaussian_adapter.d_sh
        gs = {k[len("gaussian_param_head."):]: v for k, v in sd.items() if k.startswith("gaussian_param_head.")}
        app = None
        for head in ("gaussian_param_head", "gaussian_param_head2"):
            mod = getattr(encoder, head, None)
            wk = f"{head}.dpt.head.4.weight"
            if mod is None or wk not in sd:
                continue
            rows = mod.dpt.head[4].weight.shape[0]
            if sd[wk].shape[0] == rows + d3:
                if head == "gaussian_param_head":
                    app = dict(gs)
                    app["dpt.head.4.bias"], app["dpt.head.4.weight"] = gs["dpt.head.4.bias"][-d3:], gs["dpt.head.4.weight"][-d3:]
                for pn in ("weight", "bias"):
                    sd[f"{head}.dpt.head.4.{pn}"] = sd[f"{head}.dpt.head.4.{pn}"][:rows]
        missing, unexpected = encoder.load_state_dict(sd, strict=False)
        if app is not None:
            own = encoder.gaussian_appearance_head.state_dict()
            encoder.gaussian_appearance_head.load_state_dict(
                {k: v for k, v in app.items() if k in own and own[k].shape == v.shape}, strict=False)
        return list(missing), list(unexpected)
    raise ValueError("Invalid checkpoint format: expected a 'model' or a 'state_dict' entry")


def init_token_stylizer(encoder: nn.Module, ckpt: dict) -> Tuple[list, list]:
    """main_style.py:156-168: seed the stylizer's ViT from MASt3R ('model') or from a wrapper ckpt's backbone."""
    if "model" in ckpt:
        sd = {k: v for k, v in ckpt["model"].items() if k.startswith(("enc", "mask_token", "patch_embed", "dec"))}
    elif "state_dict" in ckpt:
        pre = "encoder.backbone."
        sd = {k[len(pre):]: v for k, v in ckpt["state_dict"].items()
              if k.startswith((pre + "enc", pre + "mask_token", pre + "patch_embed"))}
    else:
        raise ValueError("Invalid token_stylizer checkpoint format")
    return tuple(encoder.token_stylizer.load_state_dict(sd, strict=False))


def load_wrapper_checkpoint(encoder: nn.Module, ckpt: dict, strict: bool = True):
    """infer_model_re10k.py:300-306: Lightning wrapper checkpoint -> encoder (keys prefixed with `encoder.`)."""
    sd = {k[len("encoder."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("encoder.")}
    return encoder.load_state_dict(sd, strict=strict)
