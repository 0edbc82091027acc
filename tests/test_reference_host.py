"""The reference's HOST code, literally unchanged, on this repository's kernels.

oracle/_ref/libref_host_{emu,hip}.so (oracle/build_ref.py: build_host_tree) holds the reference's src/gaussian_rasterizer.cpp,
src/gaussian_renderer.cpp, src/gaussian_trainer.cpp and src/gaussian_parameters.cpp compiled VERBATIM, the member functions of
GaussianModel extracted verbatim from src/gaussian_model.cpp, and nothing of this repository above the link-level boundary:
its undefined symbols RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible / distCUDA2 /
scaleAndTransformThenMarkVisiblePoints resolve in libcuda_rasterizer.so + libsimple_knn.so, the libraries the repository's
CMakeLists.txt builds under the names Photo-SLAM's gaussian_mapper links.

  * GaussianRenderer::render of the reference (ATen activations, cat(dc.clone(), rest.clone()), GaussianRasterizer through
    torch autograd) against the CPU oracle: radii exact, image to 1e-6;
  * GaussianTrainer::trainingOnce of the reference -- the whole loop: random keyframe by std::rand, the reference's loss
    header through autograd, torch::optim::Adam, addDensificationStats, the reference's own densifyAndPrune / resetOpacity --
    against oracle/cpu_trainer.train_sequence (the same loop on the CPU oracle) with the bars of
    tests/test_train_sequence_reference.py.

On the host: the emulator build of the kernels at toy size.  On the GPU (-m gpu): BASELINE config C1 (50 k Gaussians @ 640x480)
on the HIP kernels."""
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from photo_slam_amd import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHEDULE = dict(densification_interval=5, densify_from_iter=1, opacity_reset_interval=7)   # densify at 5, reset at 7
ITERATIONS = 10     # trainingOnce does not step the optimizer on its last iteration (:129): nine stepped ones
SEED = 21
RAND_SEED = 7


def _ops(kind):
    from oracle import build_ref
    path = build_ref.build_host_tree().get(kind)
    if path is None or not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_host_*.so was never built (no reference tree, no prebuilt library)")
    torch.ops.load_library(path)
    return getattr(torch.ops, build_ref.HOST_OPS[kind]), path


def _keyframe_order(draws, n_keyframes):
    """src/gaussian_trainer.cpp:59: std::rand() / ((RAND_MAX + 1u) / size); `draws` = the values the loop's std::rand() returns
    (oracle/ref_host.cpp: the library's own generator behind -Wl,--wrap=rand -- libc's is advanced by other libraries too)"""
    rand_max = 2147483647
    return [int(d) // ((rand_max + 1) // n_keyframes) for d in draws]


def _session(ops, cl, dev, gts):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    h = ops.create([t(cl.xyz), t(cl.features_dc), t(cl.features_rest), t(cl.opacity), t(cl.scaling), t(cl.rotation)], 3, 3,
                   float(cl.extent), float(cl.extent))
    cams = []
    for k, cam in enumerate(cl.cameras):
        fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
        ops.add_keyframe(h, k, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), fovx, fovy, cam.H, cam.W,
                         t(gts[k]) if gts is not None else torch.zeros(3, cam.H, cam.W, device=dev))
        # the oracle's camera carries the floats the reference's renderer forms from FoVx_ / FoVy_ (:51-52)
        tx, ty = ops.tanfov(h, k)
        cams.append(scene.Camera(cam.W, cam.H, float(tx), float(ty), cam.viewmatrix, cam.projmatrix, cam.campos))
    return h, cams


def _ground_truth(oracle, cl, cams):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_train_sequence_reference import _ground_truth
    cl2 = scene.Cloud(cl.xyz, cl.features_dc, cl.features_rest, cl.scaling, cl.rotation, cl.opacity, cams, cl.extent)
    return _ground_truth(oracle, cl2, len(cams))


def _losses_from_log(log):
    """trainingReport prints the exponential average 0.4 loss + 0.6 ema with 8 decimals (:92, :147-153): undo it"""
    ema = [float(m) for m in re.findall(r"ema_loss:([-\d.]+)", log)]
    out, prev = [], 0.0
    for e in ema:
        out.append((e - 0.6 * prev) / 0.4)
        prev = e
    return out


def check_render(ops, dev, cl):
    from oracle import oracle
    h, cams = _session(ops, cl, dev, None)
    try:
        bg_np = np.array([0.1, 0.2, 0.3], np.float32)
        bg = torch.from_numpy(bg_np).to(dev)
        for k, cam in enumerate(cams):
            image, viewspace, visible, radii = ops.render(h, k, bg, False, False)
            res, ocolor, oradii = oracle.forward(bg_np, cl.xyz, cl.get_opacity(), cam.viewmatrix, cam.projmatrix, cam.campos, cam.tanfovx,
                                                 cam.tanfovy, cam.H, cam.W, shs=cl.get_features(), sh_degree=3, scales=cl.get_scaling(),
                                                 rotations=cl.get_rotation())
            res.free()
            r = radii.cpu().numpy()
            # ATen's sigmoid / exp / normalize against numpy's: a radius may move by one where the 3-sigma extent sits on an integer
            assert (r != oradii).mean() < 2e-3 and np.abs(r.astype(np.int64) - oradii).max() <= 1, (r != oradii).sum()
            assert np.array_equal(visible.cpu().numpy(), r > 0)
            assert float(np.abs(image.detach().cpu().numpy() - ocolor).mean()) < 1e-5
            assert viewspace.requires_grad and tuple(viewspace.shape) == cl.xyz.shape
            # convert_SHs_ / compute_cov3D_ (:82-121): the reference's Python-side SH evaluation and covariance, same image
            image2 = ops.render(h, k, bg, True, True)[0]
            assert float((image2 - image).detach().abs().mean()) < 2e-5
    finally:
        ops.destroy(h)


def run_training_once(ops, dev, cl, kind, note=""):
    from oracle import cpu_trainer, oracle
    threads = min(os.cpu_count() or 1, 32)
    oracle.set_threads(threads)
    n_views = len(cl.cameras)
    # keyframes and ground truth (the oracle's render of the initial model + smooth noise)
    h, cams = _session(ops, cl, dev, None)
    ops.destroy(h)
    gts = _ground_truth(oracle, cl, cams)
    order = _keyframe_order(ops.rand_preview(RAND_SEED, ITERATIONS), n_views)
    assert len(set(order)) > 1, "the random keyframe sequence visits one keyframe only: pick another RAND_SEED"
    cl_ref = scene.Cloud(cl.xyz, cl.features_dc, cl.features_rest, cl.scaling, cl.rotation, cl.opacity, cams, cl.extent)
    # the threshold that clones / splits a few per cent of the Gaussians at the fifth iteration of THIS scene
    probe = cpu_trainer.train_sequence(cl_ref, cams, gts, SCHEDULE["densification_interval"], seed=SEED, kind=kind, threads=threads,
                                       keyframe_order=order)
    pm = probe["model"]
    g = (pm.xyz_gradient_accum / pm.denom).nan_to_num(0.0).squeeze(1)
    thr = float(np.float32(torch.quantile(g[g > 0], 0.93)))
    ref = cpu_trainer.train_sequence(cl_ref, cams, gts, ITERATIONS, densify_grad_threshold=thr, seed=SEED, kind=kind, threads=threads,
                                     keyframe_order=order, step_on_last_iteration=False, **SCHEDULE)
    assert ref["densified_at"] == [5, 10] and ref["reset_at"] == [7]
    assert ref["points"][4] != ref["points"][3], "the densification changed nothing: the sequence would not test it"

    h, _ = _session(ops, cl, dev, gts)
    try:
        (torch.cuda.manual_seed if dev.type == "cuda" else torch.manual_seed)(SEED)   # at::normal of densifyAndSplit (:734)
        opts = {"iterations": float(ITERATIONS), "densify_grad_threshold": thr, "densify_until_iter": 15000.0}
        opts.update({k: float(v) for k, v in SCHEDULE.items()})
        seconds = ops.training_once(h, opts, RAND_SEED)
        log = ops.log(h)
        dump = ops.dump(h)
        assert _keyframe_order(ops.rand_trace(), n_views) == order, "the loop drew other keyframes than the preview said"
    finally:
        ops.destroy(h)
    losses = _losses_from_log(log)
    points = [int(m) for m in re.findall(r"num_points:(\d+)", log)]
    assert len(losses) == ITERATIONS, log
    d = ref["densified_at"][0]
    # (8 printed decimals of the average: 2.5e-8 / 0.4 absolute on losses of ~0.1)
    assert np.allclose(losses[:d], ref["losses"][:d], rtol=2e-5, atol=2e-7), (losses, ref["losses"])
    assert np.allclose(losses[d:], ref["losses"][d:], rtol=1e-4, atol=2e-7), (losses, ref["losses"])
    # trainingReport runs BEFORE the densification of its iteration (:95-107): it prints the count the iteration started with
    assert points == [cl.xyz.shape[0]] + ref["points"][:-1], (points, ref["points"])
    m = ref["model"]
    params, exp_avg, exp_avg_sq = dump[0:6], dump[6:12], dump[12:18]
    accum, denom, max_radii, exist, steps = dump[18], dump[19], dump[20], dump[21], dump[22]
    assert params[0].shape[0] == ref["points"][-1]
    want = dict(xyz=m.xyz, features_dc=m.features_dc, features_rest=m.features_rest, opacity=m.opacity, scaling=m.scaling,
                rotation=m.rotation)
    lrs = dict(xyz=0.00016 * cl.extent, features_dc=0.0025, features_rest=0.0025 / 20, opacity=0.05, scaling=0.005, rotation=0.001)
    for (k, w), got in zip(want.items(), params):
        got = got.detach().cpu()
        assert got.shape == w.shape, k
        finite = torch.isfinite(w) & torch.isfinite(got)
        assert torch.equal(torch.isfinite(w), torch.isfinite(got)), k
        err = ((got - w.detach()).abs() / lrs[k])[finite]
        bad = float((err > 1e-2).float().mean())
        assert bad < 5e-3, (k, float(err.max()), bad)
    # the second densification (iteration 10) reset the statistics: compare what it left
    assert torch.equal(denom.cpu(), m.denom) and torch.equal(max_radii.cpu(), m.max_radii2D)
    assert torch.equal(exist.cpu(), m.exist_since_iter)
    ref_steps = [int(m.optimizer.state[p]["step"]) if p in m.optimizer.state else -1
                 for p in (m.xyz, m.features_dc, m.features_rest, m.opacity, m.scaling, m.rotation)]
    assert steps.tolist() == ref_steps, (steps.tolist(), ref_steps)
    print(f"[reference host, {kind}{note}] {ITERATIONS} iterations of GaussianTrainer::trainingOnce in {seconds:.2f} s: losses "
          f"{losses[0]:.6f} -> {losses[-1]:.6f} (cpu_trainer {ref['losses'][0]:.6f} -> {ref['losses'][-1]:.6f}), "
          f"points {points[0]} -> {params[0].shape[0]}, keyframes {order}")


def test_library_holds_the_reference_symbols_and_links_only_the_named_libraries():
    from oracle import build_ref
    path = build_ref.build_host_tree().get("hip")
    if path is None:
        pytest.skip("oracle/_ref/libref_host_hip.so was never built")
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[(.+?)\]", subprocess.check_output(["readelf", "-d", path], text=True))
    ours = [n for n in needed if "torch" not in n and "c10" not in n and not n.startswith(("libstdc++", "libm.", "libgcc", "libc."))]
    assert sorted(ours) == ["libcuda_rasterizer.so", "libsimple_knn.so"], needed
    defined = subprocess.check_output(["nm", "-D", "--defined-only", "-C", path], text=True)
    for want in ("GaussianRenderer::render(std::shared_ptr<GaussianKeyframe>, int, int, std::shared_ptr<GaussianModel>, "
                 "GaussianPipelineParams&, at::Tensor&, at::Tensor&, float, bool)",
                 "GaussianRasterizer::forward(at::Tensor, at::Tensor, at::Tensor, bool, bool, bool, bool, bool, at::Tensor, at::Tensor, "
                 "at::Tensor, at::Tensor, at::Tensor)", "GaussianRasterizerFunction::backward(", "GaussianTrainer::trainingOnce(",
                 "GaussianModel::densifyAndPrune(float, float, float, int)"):
        assert want in defined, want
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", "-C", path], text=True)
    for want in ("RasterizeGaussiansCUDA(", "RasterizeGaussiansBackwardCUDA(", "markVisible(at::Tensor&, at::Tensor&, at::Tensor&)",
                 "distCUDA2(at::Tensor const&)", "scaleAndTransformThenMarkVisiblePoints("):
        assert want in undefined, want
    sys.path.insert(0, os.path.join(ROOT, "photo-slam_amd", "host"))
    import build_host
    libs = build_host.outputs("hip")
    exported = subprocess.check_output(["nm", "-D", "--defined-only", "-C", libs["cuda_rasterizer"]], text=True) + \
        subprocess.check_output(["nm", "-D", "--defined-only", "-C", libs["simple_knn"]], text=True)
    for line in undefined.splitlines():
        sym = line.split(" U ", 1)[-1].strip()
        if sym.startswith(("RasterizeGaussians", "markVisible(", "distCUDA2(", "scaleAndTransform", "transformPoints(")):
            assert sym in exported, sym


def test_reference_renderer_on_the_emulated_kernels():
    ops, _ = _ops("emu")
    cl = scene.make_cloud(600, 64, 48, 50.0, 50.0, seed=5, scale_k=0.35, n_views=2)
    check_render(ops, torch.device("cpu"), cl)


@pytest.mark.parametrize("flavour", ["emu", "emu_fused_loss"])
def test_reference_training_loop_on_the_emulated_kernels(flavour):
    """emu_fused_loss: the same unchanged sources with include/loss_utils.h resolving to this repository's
    host/include/loss_utils.h (l1_loss / ssim on the fused HIP kernels): the header swap of INTEGRATION.md section 5"""
    from oracle import ref_model
    if ref_model.load("cpu") is None:
        pytest.skip("oracle/_ref/libref_densify.so was never built")
    ops, _ = _ops(flavour)
    cl = scene.make_cloud(320, 48, 32, 40.0, 40.0, seed=3, scale_k=0.35, n_views=3)
    run_training_once(ops, torch.device("cpu"), cl, "cpu", " " + flavour)


@pytest.mark.gpu
def test_reference_renderer_on_the_gpu():
    ops, _ = _ops("hip")
    cl = scene.make_config("C1", seed=0, n_views=2)
    check_render(ops, torch.device("cuda:0"), cl)


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["hip", "hip_fused_loss"])
def test_reference_training_loop_at_C1_on_the_gpu(flavour):
    """BASELINE config C1: the reference's own trainingOnce, unchanged, on the MI355X kernels against the same loop on the CPU
    oracle; hip_fused_loss: with this repository's loss_utils.h in place of the reference's (the header swap)."""
    from oracle import ref_model
    if ref_model.load("cuda") is None:
        pytest.skip("oracle/_ref/libref_densify_cuda.so was never built")
    ops, _ = _ops(flavour)
    cl = scene.make_config("C1", seed=0, n_views=3)
    run_training_once(ops, torch.device("cuda:0"), cl, "cuda", f" @C1 {flavour}")
