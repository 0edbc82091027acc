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
