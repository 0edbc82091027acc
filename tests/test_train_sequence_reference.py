"""a21 end to end, the TIMED program: the fused train step of both hosts (raw_params activations, fused loss, SH Adam and
geometry Adam inside the backward kernels, lazy SH rows with window 32, fused statistics) against the reference's loop as it
composes it (oracle/cpu_trainer.train_sequence: src/gaussian_trainer.cpp:45-133 -- CPU oracle rasterizer behind the autograd
Function, ATen activations / cat(dc, rest), the reference's loss_utils.h, torch.optim.Adam with six groups, and the REFERENCE'S
OWN densifyAndPrune / resetOpacity compiled from src/gaussian_model.cpp) over a sequence of >= 9 iterations that cycles through
three keyframes and contains one densification and one opacity reset.

On the GPU (-m gpu): the C1 cloud of BASELINE.json (50 k Gaussians, 640x480), real kernels (the v_rcp / v_sqrt update terms of
the fused optimizer steps only exist there).  On the host: the same driver on the emulator at toy size.

Bars: losses to 2e-5 relative (1e-4 after the densification, where the split children's positions agree to 1e-6 only);
the number of Gaussians after the densification exactly; parameters in units of the learning rate (Adam normalises every
gradient to a step of about lr, so a gradient whose SIGN is rounding noise flips a whole step: the overwhelming majority must
agree to 1e-2 of ONE step after all iterations); the three statistics."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from photo_slam_amd import rasterize_points as rp
from photo_slam_amd import scene
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams
from photo_slam_amd.trainer import TrainStep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHEDULE = dict(densification_interval=5, densify_from_iter=1, opacity_reset_interval=7)   # densify at 5, reset at 7
ITERATIONS = 9
SEED = 21


def _host(variant):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    return load_host(variant)


def _ground_truth(oracle, cl, n_views):
    """Per keyframe: the oracle's render of the initial model + smooth noise (a converged map under refinement)."""
    gts = []
    for k in range(n_views):
        cam = cl.cameras[k]
        res, color, _ = oracle.forward(np.zeros(3, np.float32), cl.xyz, cl.get_opacity(), cam.viewmatrix, cam.projmatrix, cam.campos,
                                       cam.tanfovx, cam.tanfovy, cam.H, cam.W, shs=cl.get_features(), sh_degree=3,
                                       scales=cl.get_scaling(), rotations=cl.get_rotation())
        res.free()
        rng = np.random.default_rng(77 + k)
        noise = rng.random((3, cam.H // 4 + 1, cam.W // 4 + 1), dtype=np.float32).repeat(4, 1).repeat(4, 2)[:, :cam.H, :cam.W]
        gts.append(np.clip(color + 0.2 * (noise - 0.5), 0.0, 1.0).astype(np.float32))
    return gts


def _compare(name, losses, points, params, stats, ref, cl, grad_threshold):
    m = ref["model"]
    d = ref["densified_at"][0]
    assert np.allclose(losses[:d], ref["losses"][:d], rtol=2e-5), (name, losses, ref["losses"])
    assert np.allclose(losses[d:], ref["losses"][d:], rtol=1e-4), (name, losses, ref["losses"])
    assert points == ref["points"], (name, points, ref["points"])
    want = dict(xyz=m.xyz, features=torch.cat([m.features_dc, m.features_rest], 1), opacity=m.opacity, scaling=m.scaling,
                rotation=m.rotation)
    lrs = dict(xyz=0.00016 * cl.extent, features=0.0025, opacity=0.05, scaling=0.005, rotation=0.001)
    for (k, w), got in zip(want.items(), params):
        got = got.detach().cpu()
        assert got.shape == w.shape, (name, k)
        # the SH tail (coefficients 1..15) steps with lr / 20
        lr = torch.full_like(w, lrs[k])
        if k == "features":
            lr[:, 1:] = lrs[k] / 20.0
        finite = torch.isfinite(w) & torch.isfinite(got)           # (the shipped opacity reset maps saturated sigmoids to +-inf)
        assert torch.equal(torch.isfinite(w), torch.isfinite(got)), (name, k)
        err = ((got - w.detach()).abs() / lr)[finite]
        bad = float((err > 1e-2).float().mean())
        assert bad < 5e-3, (name, k, float(err.max()), bad)
    accum, denom, max_radii = [t.detach().cpu() for t in stats]
    assert torch.equal(denom, m.denom), name
    assert torch.equal(max_radii, m.max_radii2D), name
    # (four iterations after the densification: the split children sit 1e-6 off the reference's, and the parameters a rounding
    # of an Adam step apart -- the gradient norms follow to a few 1e-4)
    rel = float((accum - m.xyz_gradient_accum).abs().sum() / m.xyz_gradient_accum.abs().sum())
    assert rel < 1e-4, (name, rel)
    # element by element: a few Gaussians whose whole gradient is a handful of barely-touched pixels move by more
    off = (accum - m.xyz_gradient_accum).abs() > 5e-3 * m.xyz_gradient_accum.abs() + 1e-3 * float(m.xyz_gradient_accum.median())
    assert float(off.float().mean()) < 1e-3, (name, rel, float(off.float().mean()))
    print(f"[{name}] losses {losses[0]:.6f} -> {losses[-1]:.6f} (reference {ref['losses'][0]:.6f} -> {ref['losses'][-1]:.6f}), "
          f"points {points[0]} -> {points[-1]}, statistics rel. L1 {rel:.1e}")


def run_sequence(dev, lib_path, host_variant, cl, kind, P_note=""):
    from oracle import cpu_trainer, oracle
    n_views = 3
    threads = min(os.cpu_count() or 1, 32)   # (a 50 k-Gaussian step does not feed 256 hardware threads)
    oracle.set_threads(threads)
    gts = _ground_truth(oracle, cl, n_views)
    # the threshold that clones / splits a few per cent of the Gaussians at the fifth iteration of THIS scene
    probe = cpu_trainer.train_sequence(cl, cl.cameras[:n_views], gts, SCHEDULE["densification_interval"], seed=SEED, kind=kind,
                                       threads=threads)
    pm = probe["model"]
    g = (pm.xyz_gradient_accum / pm.denom).nan_to_num(0.0).squeeze(1)
    thr = float(torch.quantile(g[g > 0], 0.93))
    ref = cpu_trainer.train_sequence(cl, cl.cameras[:n_views], gts, ITERATIONS, densify_grad_threshold=thr, seed=SEED, kind=kind,
                                     threads=threads, **SCHEDULE)
    assert ref["densified_at"] == [5] and ref["reset_at"] == [7]
    assert ref["points"][4] != ref["points"][3], "the densification changed nothing: the sequence would not test it"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    gt_t = [t(g_) for g_ in gts]
    cams = cl.cameras[:n_views]
    mask = torch.ones(3, cams[0].H, cams[0].W, device=dev)
    bg = torch.zeros(3, device=dev)
    rp._LIB_OVERRIDE = lib_path
    try:
        # ---- the Python host
        gm = GaussianModel.from_cloud(cl, device=dev)
        opt = GaussianOptimizationParams()
        opt.densification_interval_, opt.densify_from_iter_ = SCHEDULE["densification_interval"], SCHEDULE["densify_from_iter"]
        opt.opacity_reset_interval_, opt.densify_grad_threshold_ = SCHEDULE["opacity_reset_interval"], thr
        gm.trainingSetup(opt)
        ts = TrainStep(gm, opt, GaussianPipelineParams(), bg, cameras_extent=float(cl.extent), densify=True, seed=SEED,
                       lazy_sh_adam_window=32, fused_sh_adam=True, fused_geom_adam=True)
        kfs = [GaussianKeyframe.from_camera(c, dev) for c in cams]
        losses, points = [], []
        for it in range(1, ITERATIONS + 1):
            k = (it - 1) % n_views
            losses.append(float(ts.trainForOneIteration(kfs[k], gt_t[k], mask).detach()))
            points.append(int(gm.xyz_.shape[0]))
        _compare("python host" + P_note, losses, points, gm.params(), (gm.xyz_gradient_accum_, gm.denom_, gm.max_radii2D_), ref,
                 cl, thr)
        # ---- the C++ host (what bench.py times)
        ops = _host(host_variant)
        g0 = GaussianModel.from_cloud(cl, device=dev)
        h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(),
                               g0.rotation_.detach(), 3, float(cl.extent), bg)
        ops.trainer_set_options(h, {"densify": 1.0, "cameras_extent": float(cl.extent), "seed": float(SEED),
                                    "densify_from_iter": float(SCHEDULE["densify_from_iter"]),
                                    "densification_interval": float(SCHEDULE["densification_interval"]),
                                    "opacity_reset_interval": float(SCHEDULE["opacity_reset_interval"]),
                                    "densify_grad_threshold": thr, "lazy_sh_adam_window": 32.0, "fused_sh_adam": 1.0,
                                    "fused_geom_adam": 1.0})
        losses, points = [], []
        for it in range(1, ITERATIONS + 1):
            k = (it - 1) % n_views
            c = cams[k]
            loss = ops.trainer_render_and_backward(h, t(c.viewmatrix), t(c.projmatrix), t(c.campos), 2 * math.atan(c.tanfovx),
                                                   2 * math.atan(c.tanfovy), c.H, c.W, gt_t[k], mask)
            ops.trainer_finish(h)
            losses.append(float(loss))
            points.append(int(ops.trainer_params(h)[0].shape[0]))
        _compare("c++ host" + P_note, losses, points, ops.trainer_params(h), ops.trainer_stats(h), ref, cl, thr)
        ops.trainer_destroy(h)
    finally:
        rp._LIB_OVERRIDE = None
    return ref


def _need_reference_ops(kind):
    from oracle import ref_model
    if ref_model.load(kind) is None:
        pytest.skip("oracle/_ref/libref_densify*.so was never built (no reference tree, no prebuilt library)")


def test_fused_train_sequence_equals_the_reference_loop_on_the_emulator(emu_lib_path):
    _need_reference_ops("cpu")
    cl = scene.make_cloud(320, 48, 32, 40.0, 40.0, seed=3, scale_k=0.35, n_views=3)
    run_sequence(torch.device("cpu"), emu_lib_path, "emu", cl, "cpu")


@pytest.mark.gpu
def test_fused_train_sequence_equals_the_reference_loop_at_C1_on_gpu():
    """BASELINE config C1 (50 k Gaussians @ 640x480, the reference's CPU-runnable case): the program bench.py times, on the
    MI355X, against the reference's loop on the host cores."""
    _need_reference_ops("cuda")
    cl = scene.make_config("C1", seed=0, n_views=3)
    run_sequence(torch.device("cuda:0"), None, "hip", cl, "cuda", " @C1")
