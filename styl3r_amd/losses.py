"""Loss consumers of the rendered RGB: same interface `loss(prediction, batch, gaussians, step)`.

Mirrors
  * `LossMse`        src/loss/loss_mse.py:22-31
  * `LossStyle`      src/loss/loss_style.py:25-79   (VGG19 relu1_1 / 2_1 / 3_1 / 4_1 mean-std style + content)
  * `IdentityLoss`   src/loss/loss_identity.py:13-52
  * `VGGEncoder`, `calc_mean_std`   src/test/vgg_model.py:19-28,79-98
torchvision is not installed and the ImageNet VGG19 weights cannot be downloaded: `VGGEncoder` rebuilds the
`vgg19().features[:21]` stack with torchvision's parameter names (`N.weight`, N in 0,2,5,7,10,12,14,16,19), so
`load_vgg19_features(state_dict)` accepts the stock `vgg19-dcbb9e9d.pth` when it is available; until then the
weights are random and only the arithmetic is testable.  LPIPS (loss_lpips.py) needs its own learned weights too
and is not built.  The convolutions run on MIOpen in fp32.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .vit_ops import Conv2dX6   # nn.Conv2d on the CPU; bf16x6 implicit-GEMM kernels for eligible layers on the GPU

_VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512]   # features[:21] ends after relu4_1
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def calc_mean_std(x: Tensor, eps: float = 1e-8):
    """channel-wise instance mean / (unbiased) std over the flattened spatial dims -> (N, C, 1) each."""
    f = x.flatten(2)
    return f.mean(dim=-1, keepdim=True), f.std(dim=-1, keepdim=True) + eps


class VGGEncoder(nn.Module):
    """h1..h4 = relu1_1, relu2_1, relu3_1, relu4_1 of VGG19 (slices [:2], [2:7], [7:12], [12:21])."""

    def __init__(self):
        super().__init__()
        layers, c_in = [], 3
        for v in _VGG19_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [Conv2dX6(c_in, v, 3, padding=1), nn.ReLU(inplace=False)]
                c_in = v
        self.features = nn.Sequential(*layers)
        assert len(self.features) == 21
        self.requires_grad_(False)

    def load_vgg19_features(self, state_dict: dict):
        """accepts torchvision's vgg19 state dict (`features.N.*`) or the bare `features` dict (`N.*`)."""
        sd = {k[len("features."):] if k.startswith("features.") else k: v for k, v in state_dict.items()}
        return self.features.load_state_dict({k: v for k, v in sd.items() if int(k.split(".")[0]) < 21}, strict=True)

    def forward(self, images: Tensor, output_last_feature: bool = False):
        h1 = self.features[:2](images)
        h2 = self.features[2:7](h1)
        h3 = self.features[7:12](h2)
        h4 = self.features[12:21](h3)
        return h4 if output_last_feature else (h1, h2, h3, h4)


def _imagenet_normalize(x: Tensor) -> Tensor:
    mean = torch.tensor(IMAGENET_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


@dataclass
class LossMseCfg:
    weight: float = 1.0


_MSE_SCRATCH: dict = {}   # (device, stream) -> zero-initialised ticket/partials buffer of gsr_mse_forward


class _MseHip(torch.autograd.Function):
    """weight * mean((pred - target)^2) on libgsr_hip.so (include/gsr.h gsr_mse_forward/backward): 2 launches
    instead of the 7 of the torch expression; the target needs no gradient (it is ground truth)."""

    @staticmethod
    def forward(ctx, pred, target, weight):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        pred, target = pred.contiguous(), target.contiguous()
        dev = pred.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        key = (dev.index, stream)
        if key not in _MSE_SCRATCH:
            _MSE_SCRATCH[key] = torch.zeros(lib.gsr_mse_scratch_bytes(), dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.check(lib.gsr_mse_forward(pred.data_ptr(), target.data_ptr(), pred.numel(), float(weight),
                                       _MSE_SCRATCH[key].data_ptr(), loss.data_ptr(), C.c_void_p(stream)), "gsr_mse_forward")
        ctx.save_for_backward(pred, target)
        ctx.weight = float(weight)
        return loss

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from . import _lib
        pred, target = ctx.saved_tensors
        grad = torch.empty_like(pred)
        g = g.contiguous().float()
        _lib.check(_lib.load().gsr_mse_backward(pred.data_ptr(), target.data_ptr(), g.data_ptr(), pred.numel(), ctx.weight,
                                                grad.data_ptr(), C.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)),
                   "gsr_mse_backward")
        return grad, None, None


def mse_loss(pred: Tensor, target: Tensor, weight: float = 1.0) -> Tensor:
    """`weight * ((pred - target) ** 2).mean()` (loss_mse.py:27-31).  Device fp32 tensors with a ground-truth target go
    through the fused HIP kernels (and raise if libgsr_hip.so is missing); anything else is the plain torch expression."""
    if pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32 and not target.requires_grad \
            and pred.shape == target.shape and pred.numel() > 0:
        return _MseHip.apply(pred, target, weight)
    return weight * ((pred - target) ** 2).mean()


class LossMse(nn.Module):
    def __init__(self, cfg: LossMseCfg = LossMseCfg()):
        super().__init__()
        self.cfg = cfg

    def forward(self, prediction, batch, gaussians=None, global_step: int = 0) -> Tensor:
        return mse_loss(prediction.color, batch["target"]["image"], self.cfg.weight)


@dataclass
class LossStyleCfg:
    style_weight: float = 10.0


class LossStyle(nn.Module):
    def __init__(self, cfg: LossStyleCfg = LossStyleCfg(), vgg: VGGEncoder | None = None):
        super().__init__()
        self.cfg = cfg
        self.vgg = vgg or VGGEncoder()

    def forward(self, prediction, batch, gaussians=None, global_step: int = 0) -> Tensor:
        b, v = batch["target"]["image"].shape[:2]
        flat = lambda t: t.reshape(b * v, *t.shape[2:])
        target = _imagenet_normalize(flat(batch["target"]["image"]))
        pred = _imagenet_normalize(flat(prediction.color))
        style = _imagenet_normalize(batch["style"]["image"])
        # loss_style.py:53-60 repeats the style image v times and runs the VGG on the b*v copies; the feature statistics of
        # identical images are identical, so the VGG sees each style image ONCE and its (mean, std) are repeated instead
        fp, ft, fs = self.vgg(pred), self.vgg(target), self.vgg(style)
        content = F.mse_loss(fp[-2], ft[-2]) + F.mse_loss(fp[-1], ft[-1])
        style_loss = 0
        rep = lambda t: t[:, None].expand(b, v, *t.shape[1:]).reshape(b * v, *t.shape[1:])
        for a, s in zip(fp, fs):
            am, astd = calc_mean_std(a)
            sm, sstd = calc_mean_std(s)
            style_loss = style_loss + F.mse_loss(am, rep(sm)) + F.mse_loss(astd, rep(sstd))
        return content + self.cfg.style_weight * style_loss


class IdentityLoss(nn.Module):
    def __init__(self, weight_1: float = 70, weight_2: float = 1, vgg: VGGEncoder | None = None):
        super().__init__()
        self.weight_1, self.weight_2 = weight_1, weight_2
        self.vgg = vgg or VGGEncoder()

    def forward(self, prediction, batch, gaussians=None, global_step: int = 0) -> Tensor:
        b, v = batch["target"]["image"].shape[:2]
        target = batch["target"]["image"].reshape(b * v, *batch["target"]["image"].shape[2:])
        pred = prediction.color.reshape(b * v, *prediction.color.shape[2:])
        l1 = F.mse_loss(pred, target)
        fp, ft = self.vgg(_imagenet_normalize(pred)), self.vgg(_imagenet_normalize(target))
        l2 = sum(F.mse_loss(a, t) for a, t in zip(fp, ft))
        return l1 * self.weight_1 + l2 * self.weight_2


def compute_psnr(ground_truth: Tensor, predicted: Tensor) -> Tensor:
    """src/evaluation/metrics.py:11-20 (per image, inputs in [0,1])."""
    mse = ((ground_truth.clip(0, 1) - predicted.clip(0, 1)) ** 2).flatten(1).mean(dim=1)
    return -10 * mse.log10()


# ---------------------------------------------------------------------------
# LPIPS (src/loss/loss_lpips.py:27-54 uses `lpips.LPIPS(net="vgg")`, a third-party package that is not installed here and
# whose weights cannot be fetched).  This is the published LPIPS-VGG architecture with the package's parameter names, so
# its `state_dict` (vgg16 features + the five 1x1 "lin" layers) loads unchanged; without weights it is random-init.
# ---------------------------------------------------------------------------
class _LpipsLin(nn.Module):
    def __init__(self, c_in: int):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(c_in, 1, 1, bias=False))      # keys: linK.model.1.weight

    def forward(self, x):
        return self.model(x)


class _LpipsVgg16(nn.Module):
    """torchvision vgg16().features split at relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (keys: net.sliceK.<idx>.weight)"""
    _CFG = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))

    def __init__(self):
        super().__init__()
        idx, c_in = 0, 3
        for s, widths in enumerate(self._CFG, start=1):
            block = nn.Sequential()
            if s > 1:
                block.add_module(str(idx), nn.MaxPool2d(2, 2)); idx += 1
            for w in widths:
                block.add_module(str(idx), Conv2dX6(c_in, w, 3, padding=1)); idx += 1
                block.add_module(str(idx), nn.ReLU(inplace=False)); idx += 1
                c_in = w
            setattr(self, f"slice{s}", block)

    def forward(self, x):
        feats = []
        for s in range(1, 6):
            x = getattr(self, f"slice{s}")(x)
            feats.append(x)
        return feats


class LPIPS(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = _LpipsVgg16()
        for k, c in enumerate((64, 128, 256, 512, 512)):
            setattr(self, f"lin{k}", _LpipsLin(c))
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("scale", torch.tensor([.458, .448, .450]).view(1, 3, 1, 1), persistent=False)

    def forward(self, a: Tensor, b: Tensor, normalize: bool = False) -> Tensor:
        if normalize:                                                       # [0,1] -> [-1,1]
            a, b = 2 * a - 1, 2 * b - 1
        fa, fb = self.net((a - self.shift) / self.scale), self.net((b - self.shift) / self.scale)
        total = 0
        for k, (x, y) in enumerate(zip(fa, fb)):
            x = x / (x.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            y = y / (y.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            total = total + getattr(self, f"lin{k}")((x - y) ** 2).mean(dim=(2, 3), keepdim=True)
        return total                                                        # (N,1,1,1)


@dataclass
class LossLpipsCfg:
    weight: float = 0.05
    apply_after_step: int = 0


class LossLpips(nn.Module):
    def __init__(self, cfg: LossLpipsCfg = LossLpipsCfg(), lpips: LPIPS | None = None):
        super().__init__()
        self.cfg = cfg
        self.lpips = (lpips or LPIPS()).eval()
        for p in self.lpips.parameters():                                   # convert_to_buffer(..., persistent=False)
            p.requires_grad_(False)

    def forward(self, prediction, batch, gaussians=None, global_step: int = 0) -> Tensor:
        image = batch["target"]["image"]
        if global_step < self.cfg.apply_after_step:
            return torch.tensor(0, dtype=torch.float32, device=image.device)
        b, v = image.shape[:2]
        loss = self.lpips(prediction.color.reshape(b * v, *image.shape[2:]), image.reshape(b * v, *image.shape[2:]), normalize=True)
        return self.cfg.weight * loss.mean()
