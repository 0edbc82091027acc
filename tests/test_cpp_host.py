"""The LibTorch C++ host layer (photo-slam_amd/host: RasterizeGaussiansCUDA, GaussianRasterizerFunction,
GaussianRasterizer, GaussianRenderer::render, TrainStep) driven through torch.ops -- the code a C++ caller
such as gaussian_mapper links against.  On the host it is linked with the emulator build of the kernels;
the gpu-marked twin uses libphotoslam_host.so + libgsr_hip.so."""
import os
import sys

import numpy as np
import pytest
import torch

from photo_slam_amd import rasterize_points as rp
from photo_slam_amd import scene
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams, GaussianRenderer
from photo_slam_amd.trainer import TrainStep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_loaded = {}


def load_host(variant):
    if variant not in _loaded:
        sys.path.insert(0, os.path.join(ROOT, "photo-slam_amd", "host"))
        import build_host
        torch.ops.load_library(build_host.build(variant))
        _loaded[variant] = True
    return torch.ops.photoslam_amd


def _scene(dev, P=300, W=48, H=32, n_views=1):
    cl = scene.make_cloud(P, W, H, 40.0, 40.0, seed=3, scale_k=0.35, n_views=n_views)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return cl, t


def run_rasterize_checks(ops, dev, lib_path):
    cl, t = _scene(dev)
    cam = cl.cameras[0]
    g = GaussianModel.from_cloud(cl, device=dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    means2D = torch.zeros_like(g.getXYZ(), requires_grad=True)
    e = torch.empty(0, device=dev)
    color, radii = ops.rasterize_gaussians(g.getXYZ(), means2D, g.getFeatures(), e, g.getOpacityActivation(),
                                           g.getScalingActivation(), g.getRotationActivation(), e, bg, 1.0,
                                           t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, cam.H, cam.W, 3,
                                           t(cam.campos), False)
    dpix = torch.from_numpy(np.random.default_rng(0).standard_normal((3, cam.H, cam.W)).astype(np.float32)).to(dev)
    (color * dpix).sum().backward()
    cpp = [p.grad.clone() for p in g.params()] + [means2D.grad.clone()]
    # Python mirror over the same C-ABI
    rp._LIB_OVERRIDE = lib_path
    try:
        g2 = GaussianModel.from_cloud(cl, device=dev)
        kf = GaussianKeyframe.from_camera(cam, dev)
        img2, vsp, vis, radii2 = GaussianRenderer.render(kf, cam.H, cam.W, g2, GaussianPipelineParams(), bg)
        (img2 * dpix).sum().backward()
        py = [p.grad.clone() for p in g2.params()] + [vsp.grad.clone()]
    finally:
        rp._LIB_OVERRIDE = None
    assert torch.equal(radii, radii2) and torch.allclose(color, img2, atol=1e-6)
    for a, b in zip(cpp, py):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-7)
    # reference error behaviour
    with pytest.raises(RuntimeError, match="excatly one of either SHs or precomputed colors"):
        ops.rasterize_gaussians(g.getXYZ(), means2D, e, e, g.getOpacityActivation(), g.getScalingActivation(),
                                g.getRotationActivation(), e, bg, 1.0, t(cam.viewmatrix), t(cam.projmatrix), 1.0, 1.0, 8, 8,
                                0, t(cam.campos), False)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        ops.rasterize_gaussians(torch.zeros(4, 2, device=dev), means2D, g.getFeatures(), e, g.getOpacityActivation(),
                                g.getScalingActivation(), g.getRotationActivation(), e, bg, 1.0, t(cam.viewmatrix),
                                t(cam.projmatrix), 1.0, 1.0, 8, 8, 0, t(cam.campos), False)
    assert ops.mark_visible(g.getXYZ().detach(), t(cam.viewmatrix), t(cam.projmatrix)).dtype == torch.bool
    d = ops.dist_cuda2(g.getXYZ().detach())
    assert d.shape == (300,) and (d > 0).all()


def run_trainer_checks(ops, dev, lib_path):
    """C++ TrainStep == Python TrainStep (both on the fused loss / Adam kernels) over 3 iterations."""
    cl, t = _scene(dev)
    cam = cl.cameras[0]
    torch.manual_seed(0)
    gt = torch.rand(3, cam.H, cam.W).to(dev)
    mask = torch.ones(3, cam.H, cam.W, device=dev)
    bg = torch.zeros(3, device=dev)
    g = GaussianModel.from_cloud(cl, device=dev)
    h = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                           g.rotation_.detach(), 3, float(cl.extent), bg)
    import math
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    losses_cpp = []
    for _ in range(3):
        losses_cpp.append(float(ops.trainer_render_and_backward(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx,
                                                                fovy, cam.H, cam.W, gt, mask)))
        ops.trainer_finish(h)
    # the same three steps through the pieces of the data-parallel step with the view-factored exchange (a batch of one
    # view: the rebuilt SH gradient is this view's own), per-group Adam in the order bench.py uses
    h2 = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                            g.rotation_.detach(), 3, float(cl.extent), bg)
    ops.trainer_set_factored_exchange(h2, True)
    for it in range(3):
        l2 = float(ops.trainer_render_and_backward(h2, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx, fovy,
                                                   cam.H, cam.W, gt, mask))
        assert np.isclose(l2, losses_cpp[it], rtol=1e-6)
        grads = ops.trainer_grads(h2)
        assert grads[1].numel() == 0 and all(grads[i].numel() for i in (0, 2, 3, 4))
        view = ops.trainer_sh_grad_view(h2)
        assert view.shape == (300, 3)
        ops.trainer_finish_begin(h2)
        ops.trainer_features_grad_from_views(h2, t(cam.campos).reshape(1, 3), view.unsqueeze(0))
        for i in (1, 4, 0, 3, 2):
            ops.trainer_adam_group(h2, i)
        ops.trainer_finish_end(h2)
    for a, b in zip(ops.trainer_params(h2), ops.trainer_params(h)):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
    ops.trainer_destroy(h2)
    rp._LIB_OVERRIDE = lib_path
    try:
        g.trainingSetup(GaussianOptimizationParams())
        ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), bg)
        kf = GaussianKeyframe.from_camera(cam, dev)
        losses_py = [float(ts.trainForOneIteration(kf, gt, mask)) for _ in range(3)]
    finally:
        rp._LIB_OVERRIDE = None
    assert np.allclose(losses_cpp, losses_py, rtol=1e-5), (losses_cpp, losses_py)
    for a, b in zip(ops.trainer_params(h), g.params()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    acc, den, maxr = ops.trainer_stats(h)
    assert torch.allclose(acc, g.xyz_gradient_accum_, rtol=1e-4, atol=1e-9) and torch.equal(den, g.denom_)
    assert torch.equal(maxr, g.max_radii2D_)
    ops.trainer_destroy(h)


def test_cpp_rasterizer_matches_python_mirror(emu_lib_path):
    run_rasterize_checks(load_host("emu"), torch.device("cpu"), emu_lib_path)


def test_cpp_train_step_matches_python(emu_lib_path):
    run_trainer_checks(load_host("emu"), torch.device("cpu"), emu_lib_path)


@pytest.mark.gpu
def test_cpp_host_layer_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    ops = load_host("hip")
    run_rasterize_checks(ops, torch.device("cuda:0"), None)
    run_trainer_checks(ops, torch.device("cuda:0"), None)


def test_cpp_point_operators_match_python_mirror(emu_lib_path, oracle):
    ops = load_host("emu")
    from photo_slam_amd import operate_points as op
    rng = np.random.default_rng(0)
    P = 500
    pts = torch.from_numpy(rng.standard_normal((P, 3)).astype(np.float32))
    rots = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((P, 4)).astype(np.float32)))
    M = torch.eye(4)
    M[3, :3] = torch.tensor([0.3, -0.2, 0.5])
    cl = scene.make_cloud(P, 64, 48, 50.0, 50.0, seed=5)
    cam = cl.cameras[0]
    view, proj = torch.from_numpy(cam.viewmatrix), torch.from_numpy(cam.projmatrix)
    assert torch.equal(ops.transform_points(pts.clone(), M), torch.from_numpy(oracle.transform_points(pts.numpy(), M.numpy())))
    a = torch.from_numpy(rng.random(P) < 0.7)
    b = torch.from_numpy(rng.random(P) < 0.8)
    xyz = torch.from_numpy(cl.xyz)
    p1, r1, m1, n1 = ops.scale_transform_mark_visible(xyz.clone(), rots.clone(), a.clone(), b, M, view, proj, 3, 1.25)
    rp._LIB_OVERRIDE = emu_lib_path
    try:
        p2, r2, m2 = xyz.clone(), rots.clone(), a.clone()
        n2 = op.scaleAndTransformThenMarkVisiblePoints(p2, r2, m2, b, M, view, proj, 3, scale=1.25)
    finally:
        rp._LIB_OVERRIDE = None
    assert n1 == n2 and torch.equal(p1, p2) and torch.equal(r1, r2) and torch.equal(m1, m2)
    d = torch.from_numpy((rng.random(40 * 30) * 4).astype(np.float32))
    mk = torch.from_numpy(rng.random(40 * 30) < 0.5)
    assert torch.equal(ops.reproject_depth_pinhole(d, mk, [50.0, 52.0, 19.5, 14.5], 40),
                       torch.from_numpy(oracle.reproject_depth_pinhole(d.numpy(), mk.numpy(), [50.0, 52.0, 19.5, 14.5], 40)))
