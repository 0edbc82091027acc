"""Lazy Adam steps for the SH rows of culled Gaussians (gsr_sh_adam_lazy, include/gsr.h): bit-identical to the eager fused
update (tests/parity.py: check_lazy_sh_adam), on the wave64 emulator here and on the GPU."""
import copy
import math

import numpy as np
import pytest
import torch

from photo_slam_amd import scene
from tests import parity


def _scene(P, W, H, seed):
    cl = scene.make_cloud(P, W, H, 40.0, 40.0, seed=seed, scale_k=0.35)
    # three keyframes that look in different directions: the culled sets differ a lot
    cams = [scene.make_camera(W, H, 40.0, 40.0, scene.look_rotation(yaw, 0.1 * k), np.array([0.3 * k, 0.0, -0.2 * k]))
            for k, yaw in enumerate((0.3, 2.4, 4.5))]
    return cl, cams


@pytest.mark.parametrize("window,steps", [(4, 9), (32, 7)])
def test_lazy_sh_adam_equals_the_eager_update(emu_lib_path, window, steps):
    cl, cams = _scene(320, 48, 32, seed=11)
    parity.check_lazy_sh_adam(emu_lib_path, torch.device("cpu"), cl, cams, np.array([0.1, 0.2, 0.3], np.float32), steps=steps,
                              window=window)


def test_lazy_sh_adam_at_a_lower_sh_degree(emu_lib_path):
    cl, cams = _scene(200, 48, 32, seed=12)
    parity.check_lazy_sh_adam(emu_lib_path, torch.device("cpu"), cl, cams, np.zeros(3, np.float32), steps=6, window=3, sh_degree=1)


@pytest.mark.gpu
def test_lazy_sh_adam_equals_the_eager_update_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    cl, cams = _scene(200_000, 640, 480, seed=13)
    dev = torch.device("cuda:0")
    parity.check_lazy_sh_adam(None, dev, cl, cams, np.array([0.1, 0.2, 0.3], np.float32), steps=13, window=4, zero_gradient=True)
    parity.check_lazy_sh_adam(None, dev, cl, cams, np.zeros(3, np.float32), steps=40, window=32, seed=3, zero_gradient=True)
    parity.check_lazy_sh_adam(None, dev, cl, cams, np.array([0.1, 0.2, 0.3], np.float32), steps=13, window=4, exact=False)


def run_host_lazy_checks(ops, dev, lib_path, P=300, iterations=7):
    """Both hosts' train step with the lazy rows (window 3: the shortest schedule that lets rows fall two steps behind) against
    the same train step with every row stepping eagerly: keyframes that look in different directions, a densification and an
    opacity reset in the sequence, everything read back through the accessors that bring the rows up to date.  On the
    emulator the two runs are bit-identical; on the GPU two runs of the SAME program differ in the last bits (the order of the
    four quad-waves' LDS adds in the backward blend), so the comparison there is to 1e-5 of the value range."""
    from photo_slam_amd import rasterize_points as rp
    from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
    from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams
    from photo_slam_amd.trainer import TrainStep
    cl, cams = _scene(P, 48, 32, seed=21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    torch.manual_seed(0)
    gts = [torch.rand(3, c.H, c.W).to(dev) for c in cams]
    mask = torch.ones(3, cams[0].H, cams[0].W, device=dev)
    bg = torch.zeros(3, device=dev)
    options = {"densify": 1.0, "cameras_extent": float(cl.extent), "seed": 7.0, "densify_from_iter": 1.0, "densification_interval": 4.0,
               "opacity_reset_interval": 6.0, "densify_grad_threshold": 2e-5}

    def same(a, b, what):
        if dev.type == "cpu":
            assert torch.equal(a, b), what
        else:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max())), what

    # C++ host
    results = {}
    for window in (3, 0):
        g0 = GaussianModel.from_cloud(copy.deepcopy(cl), device=dev)   # (on the host from_cloud aliases the cloud's arrays)
        h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(),
                               g0.rotation_.detach(), 3, float(cl.extent), bg)
        ops.trainer_set_options(h, dict(options, lazy_sh_adam_window=float(window)))
        losses = []
        for it in range(iterations):
            c = cams[it % 3]
            losses.append(float(ops.trainer_render_and_backward(h, t(c.viewmatrix), t(c.projmatrix), t(c.campos), 2 * math.atan(c.tanfovx),
                                                                2 * math.atan(c.tanfovy), c.H, c.W, gts[it % 3], mask)))
            ops.trainer_finish(h)
        results[window] = (losses, [x.detach().clone() for x in ops.trainer_params(h)], [x.clone() for x in ops.trainer_moments(h)])
        ops.trainer_destroy(h)
    assert results[3][1][0].shape[0] != P   # the model was rebuilt in between
    assert results[3][0] == results[0][0] if dev.type == "cpu" else np.allclose(results[3][0], results[0][0], rtol=1e-5)
    for k, (a, b) in enumerate(zip(results[3][1] + results[3][2], results[0][1] + results[0][2])):
        same(a, b, ("cpp", k))
    # Python host
    rp._LIB_OVERRIDE = lib_path
    try:
        py = {}
        for window in (3, 0):
            g = GaussianModel.from_cloud(copy.deepcopy(cl), device=dev)
            opt = GaussianOptimizationParams()
            opt.densify_from_iter_, opt.densification_interval_, opt.opacity_reset_interval_, opt.densify_grad_threshold_ = 1, 4, 6, 2e-5
            g.trainingSetup(opt)
            ts = TrainStep(g, opt, GaussianPipelineParams(), bg, cameras_extent=float(cl.extent), densify=True, seed=7,
                           lazy_sh_adam_window=window)
            kfs = [GaussianKeyframe.from_camera(c, dev) for c in cams]
            losses = [float(ts.trainForOneIteration(kfs[it % 3], gts[it % 3], mask)) for it in range(iterations)]
            if window:
                assert g.optimizer_.is_lazy(g._features)   # rows ARE behind at this point ...
            feats = g.features_                             # ... and this read brings them up to date
            assert not g.optimizer_.is_lazy(g._features)
            py[window] = (losses, [x.detach().clone() for x in g.params()], [m.clone() for p in g.params() for m in g.optimizer_.moments(p)])
        assert py[3][0] == py[0][0] if dev.type == "cpu" else np.allclose(py[3][0], py[0][0], rtol=1e-5)
        for k, (a, b) in enumerate(zip(py[3][1] + py[3][2], py[0][1] + py[0][2])):
            same(a, b, ("py", k))
    finally:
        rp._LIB_OVERRIDE = None
    # the two hosts agree with each other as usual
    for a, b in zip(results[3][1], py[3][1]):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5)


def test_both_hosts_train_the_same_with_lazy_rows(emu_lib_path):
    from tests.test_cpp_host import load_host
    run_host_lazy_checks(load_host("emu"), torch.device("cpu"), emu_lib_path)


@pytest.mark.gpu
def test_both_hosts_train_the_same_with_lazy_rows_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from tests.test_cpp_host import load_host
    run_host_lazy_checks(load_host("hip"), torch.device("cuda:0"), None, P=20000, iterations=15)
