"""Fused loss / Adam kernels (csrc/train_ops.hip) against their PyTorch definitions
(include/loss_utils.h restated in photo_slam_amd.loss_utils; torch.optim.Adam)."""
import numpy as np
import pytest
import torch

from photo_slam_amd import loss_utils
from photo_slam_amd import rasterize_points as rp
from photo_slam_amd.gaussian_model import FusedAdam


@pytest.fixture()
def emu(emu_lib_path):
    rp._LIB_OVERRIDE = emu_lib_path
    yield emu_lib_path
    rp._LIB_OVERRIDE = None


def reference_loss(rendered, gt, mask, lam):
    x = rendered * mask if mask is not None else rendered
    return (1.0 - lam) * loss_utils.l1_loss(x, gt) + lam * (1.0 - loss_utils.ssim(x.unsqueeze(0), gt.unsqueeze(0)))


def check_loss(dev, H, W, use_mask, seed):
    g = torch.Generator().manual_seed(seed)
    rendered = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    mask = (torch.rand(3, H, W, generator=g) > 0.2).float().to(dev) if use_mask else None
    ref = reference_loss(rendered, gt, mask, 0.2)
    (gref,) = torch.autograd.grad(ref, rendered)
    r2 = rendered.detach().clone().requires_grad_(True)
    out = loss_utils.fused_l1_ssim_loss(r2, gt, mask, 0.2)
    (gout,) = torch.autograd.grad(out * 3.0, r2)   # also checks grad_output scaling
    assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    err = (gout / 3.0 - gref).abs().max().item()
    assert err <= 1e-5 * gref.abs().max().item() + 1e-10, (err, gref.abs().max().item())


@pytest.mark.parametrize("H,W,use_mask", [(32, 32, False), (45, 70, True), (9, 11, True), (40, 72, True), (70, 36, False)])
def test_fused_loss_matches_autograd(emu, H, W, use_mask):
    check_loss(torch.device("cpu"), H, W, use_mask, H * W)


def check_adam(dev):
    torch.manual_seed(0)
    p_ref = [torch.randn(1000, 3, device=dev, requires_grad=True), torch.randn(257, 16, 3, device=dev, requires_grad=True)]
    p_new = [t.detach().clone().requires_grad_(True) for t in p_ref]
    # reference for the split tensor: separate leaves for dc / rest
    dc = p_ref[1].detach()[:, :1].clone().requires_grad_(True)
    rest = p_ref[1].detach()[:, 1:].clone().requires_grad_(True)
    ref = torch.optim.Adam([dict(params=[p_ref[0]], lr=1e-2), dict(params=[dc], lr=2e-3), dict(params=[rest], lr=1e-4)], eps=1e-15)
    new = FusedAdam([dict(params=[p_new[0]], lr=1e-2), dict(params=[p_new[1]], lr=2e-3, period=48, split=3, lr_tail=1e-4)], eps=1e-15)
    for it in range(5):
        g0, g1 = torch.randn_like(p_ref[0]), torch.randn_like(p_ref[1])
        p_ref[0].grad, dc.grad, rest.grad = g0.clone(), g1[:, :1].clone(), g1[:, 1:].clone()
        p_new[0].grad, p_new[1].grad = g0.clone(), g1.clone()
        ref.step()
        new.step()
    assert torch.allclose(p_new[0], p_ref[0], rtol=1e-5, atol=1e-7)
    assert torch.allclose(p_new[1][:, :1], dc, rtol=1e-5, atol=1e-7)
    assert torch.allclose(p_new[1][:, 1:], rest, rtol=1e-5, atol=1e-7)


def test_fused_adam_matches_torch(emu):
    check_adam(torch.device("cpu"))


@pytest.mark.gpu
def test_fused_loss_and_adam_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    dev = torch.device("cuda:0")
    check_loss(dev, 680, 1200, True, 1)
    check_loss(dev, 97, 333, False, 2)
    check_adam(dev)
