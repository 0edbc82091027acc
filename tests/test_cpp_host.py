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
    # ... and with the separate Adam pass on the SH tensor instead of the step fused into backward (the default)
    h3 = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                            g.rotation_.detach(), 3, float(cl.extent), bg)
    ops.trainer_set_options(h3, {"fused_sh_adam": 0.0})
    for it in range(3):
        ops.trainer_render_and_backward(h3, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx, fovy, cam.H, cam.W, gt,
                                        mask)
        assert ops.trainer_grads(h3)[1].numel() == 300 * 48 and ops.trainer_grads(h)[1].numel() == 0
        ops.trainer_finish(h3)
    for a, b in zip(ops.trainer_params(h3), ops.trainer_params(h)):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
    # (the fused steps form the update term with v_rcp / v_sqrt: the parameters differ in the last bit after the first step,
    # the later gradients -- and with them the moments -- at 1e-6 of their range; on the GPU two runs also differ by the order
    # of the blend's LDS adds)
    for a, b in zip(ops.trainer_moments(h3), ops.trainer_moments(h)):
        assert torch.allclose(a, b, rtol=1e-5 if dev.type == "cpu" else 1e-3, atol=1e-6 * float(b.abs().max()))
    ops.trainer_destroy(h3)
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
        from photo_slam_amd.trainer import _one_buffer
        flat = _one_buffer([grads[i] for i in (0, 2, 3, 4)])   # one collective for the four in a data-parallel step
        assert flat is not None and flat.numel() == 11 * 300
        view = ops.trainer_sh_grad_view(h2)
        assert view.shape == (300, 3)
        assert not ops.trainer_densify_due(h2)
        ops.trainer_finish_begin(h2)
        if it == 1:   # the gradient tensor + the separate pass
            ops.trainer_features_grad_from_views(h2, t(cam.campos).reshape(1, 3), view.unsqueeze(0))
            ops.trainer_adam_group(h2, 1)
        elif it == 0:   # rebuild + Adam in one pass over two row ranges, lazy rows, then the slice (what bench.py does)
            ops.trainer_features_step_from_views(h2, t(cam.campos).reshape(1, 3), view[:148].unsqueeze(0), 0, True)
            ops.trainer_features_step_from_views(h2, t(cam.campos).reshape(1, 3), view[148:].unsqueeze(0), 148, False)
            ops.trainer_features_finish_from_views(h2)
        else:
            ops.trainer_features_step_from_views(h2, t(cam.campos).reshape(1, 3), view.unsqueeze(0), 0, True)
        if it == 0:   # the four small tensors in one Adam launch (what bench.py does)
            ops.trainer_geom_adam(h2, 1.0)
        else:
            for i in (4, 0, 3, 2):
                ops.trainer_adam_group(h2, i)
        ops.trainer_finish_end(h2)
    assert list(ops.trainer_steps(h2)) == [3] * 5
    for a, b in zip(ops.trainer_params(h2), ops.trainer_params(h)):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
    for a, b in zip(ops.trainer_stats(h2), ops.trainer_stats(h)):
        assert torch.allclose(a, b, rtol=1e-6, atol=0) and a.abs().sum() > 0
    ops.trainer_destroy(h2)
    rp._LIB_OVERRIDE = lib_path
    try:
        g.trainingSetup(GaussianOptimizationParams())
        ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), bg)
        kf = GaussianKeyframe.from_camera(cam, dev)
        losses_py = [float(ts.trainForOneIteration(kf, gt, mask)) for _ in range(3)]
        g.sync_features()   # lazy SH Adam: the rows that are behind catch up while the emulator library is still selected
    finally:
        rp._LIB_OVERRIDE = None
    assert np.allclose(losses_cpp, losses_py, rtol=1e-5), (losses_cpp, losses_py)
    for a, b in zip(ops.trainer_params(h), g.params()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    acc, den, maxr = ops.trainer_stats(h)
    assert torch.allclose(acc, g.xyz_gradient_accum_, rtol=1e-4, atol=1e-9) and torch.equal(den, g.denom_)
    assert torch.equal(maxr, g.max_radii2D_)
    ops.trainer_destroy(h)


def run_map_maintenance_checks(ops, dev, lib_path):
    """C++ GaussianModel::densifyAndPrune / resetOpacity / prunePoints / createFromPcd (host/src/gaussian_model_densify.cpp)
    == the Python mirror (gaussian_model.py), from identical states and identically seeded generators: same selection,
    same order, same children, same Adam moments."""
    cl, t = _scene(dev, P=400)
    cam = cl.cameras[0]
    torch.manual_seed(0)
    gt = torch.rand(3, cam.H, cam.W).to(dev)
    mask = torch.ones(3, cam.H, cam.W, device=dev)
    bg = torch.zeros(3, device=dev)
    names = ("xyz_", "features_", "opacity_", "scaling_", "rotation_")
    rp._LIB_OVERRIDE = lib_path
    try:
        g = GaussianModel.from_cloud(cl, device=dev)
        g.trainingSetup(GaussianOptimizationParams())
        ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), bg)
        kf = GaussianKeyframe.from_camera(cam, dev)
        for _ in range(3):
            ts.trainForOneIteration(kf, gt, mask)
        # a C++ model in exactly this state: parameters through the constructor, statistics and moments copied in place
        h = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                               g.rotation_.detach(), 3, float(cl.extent), bg)
        for dst, src in zip(ops.trainer_stats(h), (g.xyz_gradient_accum_, g.denom_, g.max_radii2D_)):
            dst.copy_(src)
        mom = ops.trainer_moments(h)
        for i, n in enumerate(names):
            m, v = g.optimizer_.moments(getattr(g, n))
            mom[i].copy_(m)
            mom[5 + i].copy_(v)
        grads = (g.xyz_gradient_accum_ / g.denom_).nan_to_num(0.0).squeeze(-1)
        thr = float(grads[grads > 0].median())
        gen = torch.Generator(device=dev).manual_seed(5)
        extent = float(torch.exp(g.scaling_.detach()).max(dim=1).values.median()) / 0.01   # half clone, half split
        info = g.densifyAndPrune(thr, 0.005, extent, 20, generator=gen)
        got = ops.trainer_densify_and_prune(h, thr, 0.005, extent, 20, 5)
        assert list(got) == [info["cloned"], info["split"], info["pruned"], info["points"]] and info["split"] > 0 and info["cloned"] > 0

        def same_state():
            for a, n in zip(ops.trainer_params(h), names):
                assert torch.equal(a, getattr(g, n).detach()), n
            mom = ops.trainer_moments(h)
            for i, n in enumerate(names):
                m, v = g.optimizer_.moments(getattr(g, n))
                assert torch.equal(mom[i], m) and torch.equal(mom[5 + i], v), n
            for a, b in zip(ops.trainer_stats(h), (g.xyz_gradient_accum_, g.denom_, g.max_radii2D_)):
                assert torch.equal(a, b)
        same_state()
        # the rebuilt leaves train on: one more iteration on both sides
        l_py = float(ts.trainForOneIteration(kf, gt, mask))
        import math
        l_cpp = float(ops.trainer_render_and_backward(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos),
                                                      2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy), cam.H, cam.W,
                                                      gt, mask))
        ops.trainer_finish(h)
        assert np.isclose(l_py, l_cpp, rtol=1e-5)
        # (the C++ trainer is three iterations younger: its Adam step counters and learning-rate schedule differ, so the
        # updated values are compared by the schedule test below, not here)
        for a, n in zip(ops.trainer_params(h), names):
            assert a.shape == getattr(g, n).shape and torch.isfinite(a).all(), n
            a.detach().copy_(getattr(g, n).detach())
        # resetOpacity / prunePoints
        g.resetOpacity()
        ops.trainer_reset_opacity(h)
        assert torch.allclose(ops.trainer_params(h)[2], g.opacity_.detach(), rtol=1e-4, atol=1e-6)
        assert not ops.trainer_moments(h)[2].any()   # (the values survive: tests/test_densify_reference.py pins the semantics)
        pm = torch.zeros(g.xyz_.shape[0], dtype=torch.bool, device=dev)
        pm[::3] = True
        g.prunePoints(pm)
        ops.trainer_prune_points(h, pm)
        for a, n in zip(ops.trainer_params(h), names):
            assert a.shape == getattr(g, n).shape and torch.allclose(a, getattr(g, n).detach(), rtol=1e-4, atol=1e-6), n
        assert ops.trainer_stats(h)[0].shape == g.xyz_gradient_accum_.shape
        ops.trainer_destroy(h)
        # createFromPcd (kNN scales)
        pts = torch.from_numpy(cl.xyz).to(dev)
        cols = torch.rand(pts.shape[0], 3, generator=torch.Generator().manual_seed(1)).to(dev)
        g2 = GaussianModel(3, device=dev)
        g2.createFromPcd(pts, cols, 2.5)
        h2 = ops.trainer_create_from_pcd(pts, cols, 3, 2.5, bg)
        for a, n in zip(ops.trainer_params(h2), names):
            assert torch.equal(a, getattr(g2, n).detach()), n
        ops.trainer_destroy(h2)
    finally:
        rp._LIB_OVERRIDE = None


def run_densify_schedule_checks(ops, dev, lib_path):
    """TrainStep's densification schedule (gaussian_mapper.cpp:711-735) in C++ == the Python trainer over 7 iterations
    with densification every 2nd and an opacity reset at the 6th."""
    cl, t = _scene(dev, P=250)
    cam = cl.cameras[0]
    torch.manual_seed(0)
    gt = torch.rand(3, cam.H, cam.W).to(dev)
    mask = torch.ones(3, cam.H, cam.W, device=dev)
    bg = torch.zeros(3, device=dev)
    g = GaussianModel.from_cloud(cl, device=dev)
    h = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                           g.rotation_.detach(), 3, float(cl.extent), bg)
    ops.trainer_set_options(h, {"densify": 1.0, "cameras_extent": float(cl.extent), "seed": 7.0, "densify_from_iter": 1.0,
                                "densification_interval": 2.0, "opacity_reset_interval": 6.0, "densify_grad_threshold": 2e-5,
                                "prune_big_point_after_iter": 3.0})
    with pytest.raises(RuntimeError, match="unknown trainer option"):
        ops.trainer_set_options(h, {"no_such_option": 1.0})
    import math
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    rp._LIB_OVERRIDE = lib_path
    try:
        opt = GaussianOptimizationParams()
        opt.densify_from_iter_, opt.densification_interval_, opt.opacity_reset_interval_ = 1, 2, 6
        opt.densify_grad_threshold_ = 2e-5
        g.trainingSetup(opt)
        ts = TrainStep(g, opt, GaussianPipelineParams(), bg, cameras_extent=float(cl.extent), densify=True,
                       prune_big_point_after_iter=3, seed=7)
        kf = GaussianKeyframe.from_camera(cam, dev)
        sizes = []
        for it in range(1, 8):
            l_py = float(ts.trainForOneIteration(kf, gt, mask))
            l_cpp = float(ops.trainer_render_and_backward(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx, fovy,
                                                          cam.H, cam.W, gt, mask))
            ops.trainer_finish(h)
            assert np.isclose(l_py, l_cpp, rtol=2e-4), (it, l_py, l_cpp)
            assert ops.trainer_params(h)[0].shape == g.xyz_.shape, it
            sizes.append(g.xyz_.shape[0])
            if it % 2 == 0:
                d = ts.last_densify_
                assert list(ops.trainer_last_densify(h)) == [d["cloned"], d["split"], d["pruned"], d["points"]], it
        assert len(set(sizes)) > 2, sizes   # the model did change size
        for a, b in zip(ops.trainer_params(h), g.params()):
            assert torch.allclose(a, b.detach(), rtol=1e-3, atol=1e-5)
    finally:
        rp._LIB_OVERRIDE = None
    ops.trainer_destroy(h)


def run_pipeline_flag_checks(ops, dev, lib_path):
    """GaussianPipelineParams::convert_SHs_ / compute_cov3D_ (src/gaussian_renderer.cpp:78-113) in both hosts: the image and
    the gradients equal the default data flow (SH and covariance evaluated inside the rasterizer, whose arithmetic is pinned
    to the reference kernels) -- the SH colour to rounding, the gradients to 1e-4."""
    import math
    cl, t = _scene(dev, P=500)
    cam = cl.cameras[0]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    dpix = torch.from_numpy(np.random.default_rng(1).standard_normal((3, cam.H, cam.W)).astype(np.float32)).to(dev)
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    kf = GaussianKeyframe.from_camera(cam, dev)
    rp._LIB_OVERRIDE = lib_path
    try:
        results = {}
        for flags in ((False, False), (True, False), (False, True), (True, True)):
            g = GaussianModel.from_cloud(cl, device=dev)
            g.active_sh_degree_ = 3
            pipe = GaussianPipelineParams(convert_SHs_=flags[0], compute_cov3D_=flags[1])
            img, vsp, vis, radii = GaussianRenderer.render(kf, cam.H, cam.W, g, pipe, bg, fuse_activations=False)
            (img * dpix).sum().backward()
            py = (img.detach(), radii, [p.grad.clone() for p in g.params()])
            h = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                                   g.rotation_.detach(), 3, float(cl.extent), bg)
            img_c, radii_c = ops.trainer_render(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx, fovy, cam.H, cam.W,
                                                flags[0], flags[1], False)
            (img_c * dpix).sum().backward()
            cpp = [x.clone() for x in ops.trainer_grads(h)]
            assert torch.equal(radii_c, radii) and torch.allclose(img_c.detach(), py[0], atol=1e-6), flags
            for a, b in zip(cpp, py[2]):
                assert float((a - b).abs().sum() / (b.abs().sum() + 1e-30)) < 1e-5, flags
            ops.trainer_destroy(h)
            results[flags] = py
        base = results[(False, False)]
        for flags, (img, radii, grads) in results.items():
            assert torch.equal(radii, base[1]), flags          # the covariance built in torch gives the same radii here
            assert torch.allclose(img, base[0], atol=2e-6), flags
            for a, b in zip(grads, base[2]):
                rel = float((a - b).abs().sum() / (b.abs().sum() + 1e-30))
                assert rel < 1e-4, (flags, rel)
    finally:
        rp._LIB_OVERRIDE = None


def run_pipeline_flag_train_checks(ops, dev, lib_path):
    """A TRAIN step with convert_SHs_ / compute_cov3D_ set: render() then keeps the SH tensor resp. the raw scaling /
    rotation leaves out of the rasterizer, so the fused optimizer paths must not be armed -- every group takes exactly ONE Adam
    step per iteration, in both hosts, and the result equals the default data flow's (same maths) to a small fraction of a
    learning-rate step."""
    import math
    cl, t = _scene(dev, P=240 if dev.type == "cpu" else 400)
    cam = cl.cameras[0]
    torch.manual_seed(0)
    gt = torch.rand(3, cam.H, cam.W).to(dev)
    mask = torch.ones(3, cam.H, cam.W, device=dev)
    mask[:, :6, :] = 0.0   # a real mask (the train steps recognise an all-ones mask and skip it: this one takes the masked path)
    bg = torch.zeros(3, device=dev)
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    kf = GaussianKeyframe.from_camera(cam, dev)
    lrs = [0.00016 * cl.extent, 0.0025, 0.05, 0.005, 0.001]
    n_it = 3 if dev.type == "cpu" else 4
    rp._LIB_OVERRIDE = lib_path
    try:
        final = {}
        for flags in ((False, False), (True, False), (False, True), (True, True)):
            g = GaussianModel.from_cloud(cl, device=dev)
            opt = GaussianOptimizationParams()
            g.trainingSetup(opt)
            h = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                                   g.rotation_.detach(), 3, float(cl.extent), bg)
            ops.trainer_set_options(h, {"convert_SHs": float(flags[0]), "compute_cov3D": float(flags[1])})
            ts = TrainStep(g, opt, GaussianPipelineParams(convert_SHs_=flags[0], compute_cov3D_=flags[1]), bg)
            for it in range(n_it):
                l_py = float(ts.trainForOneIteration(kf, gt, mask))
                l_cpp = float(ops.trainer_render_and_backward(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx, fovy,
                                                              cam.H, cam.W, gt, mask))
                ops.trainer_finish(h)
                assert np.isclose(l_py, l_cpp, rtol=2e-5), (flags, it, l_py, l_cpp)
            # one Adam step per group and iteration: nothing stepped twice, no phantom lazy steps
            assert list(ops.trainer_steps(h)) == [n_it] * 5, (flags, list(ops.trainer_steps(h)))
            assert [g.optimizer_.state[id(p)]["step"] for p in g.params()] == [n_it] * 5, flags
            got = [p.detach().clone() for p in g.params()]
            for a, b, lr in zip(ops.trainer_params(h), got, lrs):
                err = (a - b).abs() / lr
                assert float((err > 1e-2).float().mean()) < 2e-3, (flags, float(err.max()))
            final[flags] = got
            ops.trainer_destroy(h)
        for flags, got in final.items():
            for a, b, lr in zip(got, final[(False, False)], lrs):
                err = (a - b).abs() / lr
                assert float((err > 2e-2).float().mean()) < 5e-3, (flags, float(err.max()), float((err > 2e-2).float().mean()))
    finally:
        rp._LIB_OVERRIDE = None


def test_train_step_with_pipeline_flags_steps_every_group_once(emu_lib_path):
    run_pipeline_flag_train_checks(load_host("emu"), torch.device("cpu"), emu_lib_path)


def test_cpp_pipeline_flags_match_python_and_the_default_flow(emu_lib_path):
    run_pipeline_flag_checks(load_host("emu"), torch.device("cpu"), emu_lib_path)


def test_cpp_map_maintenance_matches_python(emu_lib_path):
    run_map_maintenance_checks(load_host("emu"), torch.device("cpu"), emu_lib_path)


def test_cpp_densify_schedule_matches_python(emu_lib_path):
    run_densify_schedule_checks(load_host("emu"), torch.device("cpu"), emu_lib_path)


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
    run_map_maintenance_checks(ops, torch.device("cuda:0"), None)
    run_pipeline_flag_checks(ops, torch.device("cuda:0"), None)
    run_pipeline_flag_train_checks(ops, torch.device("cuda:0"), None)


@pytest.mark.gpu
def test_cpp_data_parallel_step_over_rccl_with_one_rank_on_gpu():
    """The C++ host's data-parallel step on the MI355X with a ONE-rank RCCL process group (what `GSR_BENCH_FORCE_DP=1 bench.py`
    times): render + backward in the factored mode with the lazy rows' catch-up inside, the all-gather issued on the side stream
    that waits only for the colour gradients, the all-reduce (SUM), the SH step for the lit rows, the four small tensors in one
    Adam launch -- against the fused single-GPU program from the same start, and against the plain all-reduce exchange."""
    import math
    import torch.distributed as dist
    ops = load_host("hip")
    dev = torch.device("cuda:0")
    cl, t = _scene(dev, P=20000, W=320, H=240)
    cam = cl.cameras[0]
    torch.manual_seed(0)
    gt = torch.rand(3, cam.H, cam.W).to(dev)
    mask = torch.ones(3, cam.H, cam.W, device=dev)
    bg = torch.zeros(3, device=dev)
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    g = GaussianModel.from_cloud(cl, device=dev)
    make = lambda: ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                                      g.rotation_.detach(), 3, float(cl.extent), bg)
    step = lambda h: float(ops.trainer_train_one_iteration(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx, fovy, cam.H,
                                                           cam.W, gt, mask))
    n_it = 6
    h_fused = make()
    ops.trainer_set_options(h_fused, {"lazy_sh_adam_window": 4.0})
    l_fused = [step(h_fused) for _ in range(n_it)]
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", rank=0, world_size=1, device_id=dev)
    try:
        results = {}
        # ("packed": the view-factored exchange sending only the rows the view sees -- the message packed on the gather stream behind
        # the backward pass, the decoder inside the SH step; include/gsr.h gsr_pack_color_view.  "packed_in_backward": the backward
        # pass writes the message itself, gsr_backward_args.packed_view)
        for factored in (True, False, "packed", "packed_in_backward"):
            h = make()
            ops.trainer_set_options(h, {"lazy_sh_adam_window": 4.0})   # (window 4: rows fall behind and catch up within six steps)
            ops.trainer_set_process_group(h, dist.group.WORLD.group_name, bool(factored))
            if factored in ("packed", "packed_in_backward"):
                ops.trainer_set_options(h, {"packed_exchange": 1.0, "pack_in_backward": 1.0 if factored == "packed_in_backward" else 0.0})
            losses = [step(h) for _ in range(n_it)]
            results[factored] = (losses, [p.detach().clone() for p in ops.trainer_params(h)], list(ops.trainer_steps(h)))
            ops.trainer_destroy(h)
    finally:
        if own_group:
            dist.destroy_process_group()
    want = ops.trainer_params(h_fused)
    lrs = [0.00016 * cl.extent, 0.0025 / 20.0, 0.05, 0.005, 0.001]
    for factored, (losses, params, steps) in results.items():
        assert steps == [n_it] * 5, (factored, steps)
        assert np.allclose(losses, l_fused, rtol=2e-5), (factored, losses, l_fused)
        for a, b, lr in zip(params, want, lrs):
            # (the fused program forms its update terms with v_rcp / v_sqrt, the separate Adam launches with exact division: in units
            # of a learning-rate step, as everywhere the two are compared)
            err = (a - b).abs() / lr
            assert float((err > 1e-2).float().mean()) < 2e-3, (factored, float(err.max()))
    # the packed form against the dense one: the same rows in the same order -- differences only where two runs of the same
    # backward pass differ (the blend's LDS add order)
    for form in ("packed", "packed_in_backward"):
        for a, b, lr in zip(results[form][1], results[True][1], lrs):
            assert float((((a - b).abs() / lr) > 1e-2).float().mean()) < 2e-3, form
    ops.trainer_destroy(h_fused)


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


def _reference_loss_ops():
    from oracle import build_ref
    path = build_ref.build_loss()
    if not path or not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_loss.so was never built (no reference tree, no prebuilt library)")
    torch.ops.load_library(path)
    ops = torch.ops.photoslam_reference
    if not hasattr(ops, "ssim_ex"):
        pytest.skip("a libref_loss.so from before round 6 (no reference tree here to rebuild it)")
    return ops


def _check_loss_header_functions(ops, ref, dev):
    """host/include/loss_utils.h keeps the reference's names and signatures (include/loss_utils.h:24-126) and computes the functions
    the fused kernels do not cover its own way (separable windows, one grouped convolution pair for the five statistics): the VALUES
    are the reference's -- its own header compiled (oracle/ref_loss.cpp) -- for every call shape the fallback takes."""
    g = torch.Generator().manual_seed(5)
    rnd = lambda *shape: torch.rand(*shape, generator=g).to(dev)
    cpu = lambda t: t.detach().cpu()
    for shape, ws, avg in (((1, 3, 40, 56), 11, False),      # no size average: per-image values
                           ((2, 3, 33, 47), 11, True),       # a batch
                           ((2, 3, 33, 47), 11, False),
                           ((1, 3, 40, 56), 7, True),        # another window
                           ((1, 1, 24, 31), 5, True),        # one channel
                           ((3, 40, 56), 9, True)):          # unbatched [3,H,W]
        a, b = rnd(*shape), rnd(*shape)
        got, want = cpu(ops.loss_ssim(a, b, ws, avg)), ref.ssim_ex(cpu(a), cpu(b), ws, avg)
        assert got.shape == want.shape and torch.allclose(got, want, rtol=0, atol=1e-6), (shape, ws, avg, float((got - want).abs().max()))
    a, b = rnd(1, 3, 40, 56), rnd(1, 3, 40, 56)
    # a target that requires a gradient: the fallback, differentiable in both arguments
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ops.loss_ssim(a1, b1, 11, True).backward()
    a2, b2 = cpu(a).clone().requires_grad_(True), cpu(b).clone().requires_grad_(True)
    ref.ssim_ex(a2, b2, 11, True).backward()
    # (gradients of ~1e-4 per pixel through two summation orders: 1e-4 relative + a few float ulps of the largest absolute)
    assert torch.allclose(cpu(a1.grad), a2.grad, rtol=1e-4, atol=2e-9) and torch.allclose(cpu(b1.grad), b2.grad, rtol=1e-4, atol=2e-9), \
        (float((cpu(a1.grad) - a2.grad).abs().max()), float((cpu(b1.grad) - b2.grad).abs().max()), float(a2.grad.abs().max()))
    # the window itself, and _ssim with a caller's window: a separable one and one that is not (the plain 2-D path)
    for ws, ch in ((11, 3), (7, 1)):
        assert torch.allclose(cpu(ops.loss_create_window(ws, ch, a)), ref.create_window(ws, ch, cpu(a)), rtol=0, atol=1e-7)
    w = ref.create_window(11, 3, cpu(a)).to(dev)
    assert torch.allclose(cpu(ops.loss_ssim_with_window(a, b, w, 11, True)), ref.ssim_ex(cpu(a), cpu(b), 11, True), rtol=0, atol=1e-6)
    lumpy = torch.rand(11, 11, generator=g)
    lumpy = (lumpy / lumpy.sum()).expand(3, 1, 11, 11).contiguous()
    import torch.nn.functional as F
    from photo_slam_amd import loss_utils as mirror
    want = mirror._ssim(cpu(a), cpu(b), lumpy, 11, 3, True)
    assert torch.allclose(cpu(ops.loss_ssim_with_window(a, b, lumpy.to(dev), 11, True)), want, rtol=0, atol=1e-6)
    assert torch.allclose(cpu(ops.loss_psnr(a, b)), ref.psnr(cpu(a), cpu(b)), rtol=1e-6, atol=1e-5)
    x, y = rnd(4, 3, 20, 30), rnd(4, 3, 20, 30)
    assert torch.allclose(cpu(ops.loss_psnr_gaussian_splatting(x, y)), ref.psnr_gaussian_splatting(cpu(x), cpu(y)), rtol=1e-6, atol=1e-5)
    # host tensors of another dtype / layout take the fallback too (the fused kernels are float32 [3,H,W] only)
    ad, bd = cpu(a).double(), cpu(b).double()
    if dev.type == "cpu":
        # (the window is float32 on both sides; the reference multiplies its 2-D product, this library its two 1-D factors)
        assert torch.allclose(ops.loss_ssim(ad, bd, 11, True), ref.ssim_ex(ad, bd, 11, True), rtol=0, atol=1e-7)
        assert torch.equal(ops.loss_l1(ad, bd), ref.l1_loss(ad, bd))


def test_loss_header_functions_equal_the_reference_header_emu():
    _check_loss_header_functions(load_host("emu"), _reference_loss_ops(), torch.device("cpu"))


@pytest.mark.gpu
def test_loss_header_functions_equal_the_reference_header_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    _check_loss_header_functions(load_host("hip"), _reference_loss_ops(), torch.device("cuda:0"))
