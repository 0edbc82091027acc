"""include/loss_utils.h:28-124: L1, PSNR, SSIM (11x11 Gaussian window, sigma 1.5, grouped conv2d)."""
import math

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def psnr(img1, img2):
    mse = torch.pow(img1 - img2, 2).mean()
    return 10.0 * torch.log10(1.0 / mse)


def gaussian(window_size, sigma, device):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / (2.0 * sigma * sigma)) for x in range(window_size)],
                     dtype=torch.float32, device=device)
    return g / g.sum()


_window_cache = {}


def create_window(window_size, channel, device):
    key = (window_size, channel, str(device))
    if key not in _window_cache:
        w1 = gaussian(window_size, 1.5, device).unsqueeze(1)
        w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
        _window_cache[key] = w2.expand(channel, 1, window_size, window_size).contiguous()
    return _window_cache[key]


def _ssim(img1, img2, window, window_size, channel, size_average=True):
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 * 0.01, 0.03 * 0.03
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = create_window(window_size, channel, img1.device).type_as(img1)
    return _ssim(img1, img2, window, window_size, channel, size_average)


# ---------------------------------------------------------------------------------------------
# Fused HIP version of the train-step loss (csrc/train_ops.hip, gsr_l1_ssim_loss):
#   loss = (1 - lambda) * l1_loss(rendered * mask, gt) + lambda * (1 - ssim(rendered * mask, gt))
# (src/gaussian_mapper.cpp:692-698) with its gradient w.r.t. `rendered`, in two LDS-tiled kernels.
import ctypes as _C

from . import rasterize_points as _rp


class _FusedL1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rendered, gt, mask, lambda_dssim, is_root=False):
        ctx.is_root = bool(is_root)
        lib = _rp._lib()
        _rp._check_device(lib, rendered, gt)
        r = rendered.contiguous().float()
        g = gt.contiguous().float()
        m = None if mask is None else mask.contiguous().float()
        _, H, W = r.shape
        grad = torch.empty_like(r)
        loss = torch.empty(1, dtype=torch.float32, device=r.device)
        scratch = torch.empty(int(lib.gsr_loss_scratch_bytes(W, H)), dtype=torch.uint8, device=r.device)
        st = lib.gsr_l1_ssim_loss(r.data_ptr(), g.data_ptr(), None if m is None else m.data_ptr(), W, H,
                                  float(lambda_dssim), grad.data_ptr(), loss.data_ptr(), scratch.data_ptr(),
                                  _rp._stream_ptr(r))
        from . import capi
        capi.check(lib, st, "gsr_l1_ssim_loss")
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        if ctx.is_root:
            return grad, None, None, None, None
        return grad * grad_out, None, None, None, None


def fused_l1_ssim_loss(rendered, gt, mask, lambda_dssim, is_root=False):
    """is_root: the caller promises to call .backward() on this very value (upstream gradient exactly 1, as the train step
    does): backward then hands the stored gradient on without the [3,H,W] multiply by one."""
    return _FusedL1SSIM.apply(rendered, gt, mask, lambda_dssim, is_root)
