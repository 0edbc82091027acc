"""ONE TrainStep driven through alternating image sizes: the SLAM flavour of the train step.

GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:631-699) renders a keyframe at the size of its CURRENT
Gaussian-pyramid level -- gaus_pyramid_height_/width_[level] = size * 0.5^(levels - level) (:155-160, :302-306), the level's image
and undistortion mask -- multiplies the rendering by the mask (:692) and sets the position learning rate from the keyframe's
use count (:663-671), so consecutive iterations of one session differ in H x W, mask and learning rate.  The fused step keeps
state across iterations (the loss kernels' scratch, the all-ones-mask cache, gsr_forward's growable scratch buffers, the lazy SH
rows' learning-rate history, the exchange's persistent gather buffers): this file drives both hosts through three keyframes in
rotation over three pyramid levels (full, 1/2, 1/4; staggered, so that every iteration changes the size) for 42 iterations with
lazy SH rows, a non-trivial mask per level, per-keyframe position learning rates and a densification inside, against the
reference's loop (oracle/cpu_trainer.train_sequence with a plan) -- on the emulator at toy size and on the MI355X at C1 -- and
gives two gloo ranks different H x W ("mixed pyramid levels within a batch are allowed"), dense and packed
exchange, replicas bit-identical and equal to one process that accumulates both views."""
import dataclasses
import math
import os
import subprocess
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
ITERATIONS = 42
DENSIFY_AT = 36   # (late: what follows a densification amplifies the 1e-6 the split children's positions differ by)
USES_PER_LEVEL = 4
SEED = 23
FACTORS = (0.25, 0.5, 1.0)   # kf_gaus_pyramid_factors_ of two sub-levels + the full image (:302-306, :631-647)


def _host(variant):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    return load_host(variant)


def level_camera(cam, level):
    f = FACTORS[level]
    return dataclasses.replace(cam, W=int(cam.W * f), H=int(cam.H * f))   # (:159-160: int truncation; same pose, same field of view)


def level_mask(H, W, seed):
    """an undistortion mask with black edges and one blind spot: [3,H,W] of 0 / 1"""
    m = np.ones((H, W), np.float32)
    b = max(1, min(H, W) // 16)
    m[:b] = m[-b:] = 0.0
    m[:, :b] = m[:, -b:] = 0.0
    rng = np.random.default_rng(seed)
    cy, cx, r = rng.integers(H // 4, 3 * H // 4), rng.integers(W // 4, 3 * W // 4), max(1, min(H, W) // 8)
    yy, xx = np.mgrid[:H, :W]
    m[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = 0.0
    return np.repeat(m[None], 3, 0).copy()


def make_plan(oracle, cl, n_views, iterations):
    """Per iteration dict(k, level, cam, gt, mask, lr_step): keyframe k = (it - 1) % n_views at the level its use count says
    (USES_PER_LEVEL uses per sub-level, then the full image: GaussianKeyframe::getCurrentGausPyramidLevel,
    src/gaussian_keyframe.cpp:206-216), keyframe k starting k levels up so that consecutive iterations differ in size."""
    cams = [[level_camera(cl.cameras[k], lv) for lv in range(3)] for k in range(n_views)]
    gts, masks = {}, {}
    for k in range(n_views):
        for lv in range(3):
            cam = cams[k][lv]
            res, color, _ = oracle.forward(np.zeros(3, np.float32), cl.xyz, cl.get_opacity(), cam.viewmatrix, cam.projmatrix, cam.campos,
                                           cam.tanfovx, cam.tanfovy, cam.H, cam.W, shs=cl.get_features(), sh_degree=3,
                                           scales=cl.get_scaling(), rotations=cl.get_rotation())
            res.free()
            rng = np.random.default_rng(900 + 10 * k + lv)
            noise = rng.random((3, cam.H // 4 + 1, cam.W // 4 + 1), dtype=np.float32).repeat(4, 1).repeat(4, 2)[:, :cam.H, :cam.W]
            masks[k, lv] = level_mask(cam.H, cam.W, 50 + 10 * k + lv)
            gts[k, lv] = (np.clip(color + 0.2 * (noise - 0.5), 0.0, 1.0) * masks[k, lv]).astype(np.float32)   # (black where the mask is)
    used = [0] * n_views
    plan = []
    for it in range(1, iterations + 1):
        k = (it - 1) % n_views
        lv = min((used[k] + k * USES_PER_LEVEL) // USES_PER_LEVEL, 2)
        used[k] += 1
        # kfs_used_times_ (:663): in thousands here, so that the schedule moves the learning rate visibly within 42 iterations
        plan.append(dict(k=k, level=lv, cam=cams[k][lv], gt=gts[k, lv], mask=masks[k, lv], lr_step=1000 * used[k]))
    return plan


def _compare(name, losses, points, params, stats, ref, cl):
    m = ref["model"]
    d = ref["densified_at"][0]
    assert np.allclose(losses[:d], ref["losses"][:d], rtol=5e-5), (name, np.abs(np.array(losses[:d]) / np.array(ref["losses"][:d]) - 1).max())
    assert np.allclose(losses[d:], ref["losses"][d:], rtol=3e-4), (name, np.abs(np.array(losses[d:]) / np.array(ref["losses"][d:]) - 1).max())
    assert points == ref["points"], (name, points, ref["points"])
    want = dict(xyz=m.xyz, features=torch.cat([m.features_dc, m.features_rest], 1), opacity=m.opacity, scaling=m.scaling, rotation=m.rotation)
    lrs = dict(xyz=0.00016 * cl.extent, features=0.0025, opacity=0.05, scaling=0.005, rotation=0.001)
    worst = {}
    for (k, w), got in zip(want.items(), params):
        got = got.detach().cpu()
        assert got.shape == w.shape, (name, k)
        lr = torch.full_like(w, lrs[k])
        if k == "features":
            lr[:, 1:] = lrs[k] / 20.0
        err = (got - w.detach()).abs() / lr
        # Adam turns a gradient whose SIGN is rounding noise into a whole step, 42 times here: the overwhelming majority of the
        # elements must still agree to a few per cent of ONE step after all iterations
        worst[k] = (float((err > 5e-2).float().mean()), float(err.max()))
        assert worst[k][0] < 2e-2, (name, k, worst[k])
    accum, denom, max_radii = [t.detach().cpu() for t in stats]
    # (the radius is ceil(3 sqrt(lambda)) of activations the fused step evaluates in-kernel -- v_exp_f32 / v_rcp_f32 on the hardware -- and
    # the reference in ATen: over 42 views at three sizes a Gaussian whose 3 sqrt(lambda) sits within an ulp of an integer flips by
    # one pixel now and then, and one at the edge of visibility is counted once more or less)
    assert float((denom != m.denom).float().mean()) < 5e-4 and float((denom - m.denom).abs().max()) <= 1, name
    assert float((max_radii != m.max_radii2D).float().mean()) < 5e-4 and float((max_radii - m.max_radii2D).abs().max()) <= 1, name
    rel = float((accum - m.xyz_gradient_accum).abs().sum() / m.xyz_gradient_accum.abs().sum())
    assert rel < 5e-4, (name, rel)
    print(f"[{name}] losses {losses[0]:.6f} -> {losses[-1]:.6f} (reference {ref['losses'][0]:.6f} -> {ref['losses'][-1]:.6f}), points "
          f"{points[0]} -> {points[-1]}, statistics rel. L1 {rel:.1e}, share of elements off by > 5 % of a step / worst (steps): {worst}")


def run_mixed_sequence(dev, lib_path, host_variant, cl, kind, note=""):
    from oracle import cpu_trainer, oracle
    n_views = 3
    threads = min(os.cpu_count() or 1, 32)
    oracle.set_threads(threads)
    plan = make_plan(oracle, cl, n_views, ITERATIONS)
    sizes = sorted({(s["cam"].H, s["cam"].W) for s in plan})
    assert len(sizes) == 3 and all(plan[i]["cam"].H != plan[i + 1]["cam"].H for i in range(8)), "the size must change from step to step"
    schedule = dict(densification_interval=DENSIFY_AT, densify_from_iter=1)
    probe = cpu_trainer.train_sequence(cl, None, None, DENSIFY_AT, seed=SEED, kind=kind, threads=threads, plan=plan)
    pm = probe["model"]
    g = (pm.xyz_gradient_accum / pm.denom).nan_to_num(0.0).squeeze(1)
    thr = float(torch.quantile(g[g > 0], 0.93))
    ref = cpu_trainer.train_sequence(cl, None, None, ITERATIONS, densify_grad_threshold=thr, seed=SEED, kind=kind, threads=threads,
                                     plan=plan, **schedule)
    assert ref["densified_at"] == [DENSIFY_AT] and ref["points"][DENSIFY_AT - 1] != ref["points"][DENSIFY_AT - 2]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    gt_t = {(s["k"], s["level"]): t(s["gt"]) for s in plan}
    mask_t = {(s["k"], s["level"]): t(s["mask"]) for s in plan}      # (one tensor object per keyframe and level, as a session holds them)
    bg = torch.zeros(3, device=dev)
    rp._LIB_OVERRIDE = lib_path
    try:
        gm = GaussianModel.from_cloud(cl, device=dev)
        opt = GaussianOptimizationParams()
        opt.densification_interval_, opt.densify_from_iter_, opt.densify_grad_threshold_ = DENSIFY_AT, 1, thr
        opt.opacity_reset_interval_ = 0
        gm.trainingSetup(opt)
        ts = TrainStep(gm, opt, GaussianPipelineParams(), bg, cameras_extent=float(cl.extent), densify=True, seed=SEED,
                       lazy_sh_adam_window=32, fused_sh_adam=True, fused_geom_adam=True)
        kfs = {(s["k"], s["level"]): GaussianKeyframe.from_camera(s["cam"], dev) for s in plan}
        losses, points = [], []
        for s in plan:
            key = (s["k"], s["level"])
            losses.append(float(ts.trainForOneIteration(kfs[key], gt_t[key], mask_t[key], position_lr_step=s["lr_step"]).detach()))
            points.append(int(gm.xyz_.shape[0]))
        _compare("python host" + note, losses, points, gm.params(), (gm.xyz_gradient_accum_, gm.denom_, gm.max_radii2D_), ref, cl)
        ops = _host(host_variant)
        g0 = GaussianModel.from_cloud(cl, device=dev)
        h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(), g0.rotation_.detach(), 3,
                               float(cl.extent), bg)
        ops.trainer_set_options(h, {"densify": 1.0, "cameras_extent": float(cl.extent), "seed": float(SEED), "densify_from_iter": 1.0,
                                    "densification_interval": float(DENSIFY_AT), "opacity_reset_interval": 0.0,
                                    "densify_grad_threshold": thr, "lazy_sh_adam_window": 32.0, "fused_sh_adam": 1.0, "fused_geom_adam": 1.0})
        losses, points = [], []
        for s in plan:
            key, c = (s["k"], s["level"]), s["cam"]
            ops.trainer_set_options(h, {"position_lr_step": float(s["lr_step"])})
            loss = ops.trainer_render_and_backward(h, t(c.viewmatrix), t(c.projmatrix), t(c.campos), 2 * math.atan(c.tanfovx),
                                                   2 * math.atan(c.tanfovy), c.H, c.W, gt_t[key], mask_t[key])
            ops.trainer_finish(h)
            losses.append(float(loss))
            points.append(int(ops.trainer_params(h)[0].shape[0]))
        _compare("c++ host" + note, losses, points, ops.trainer_params(h), ops.trainer_stats(h), ref, cl)
        ops.trainer_destroy(h)
    finally:
        rp._LIB_OVERRIDE = None


def _need_reference_ops(kind):
    from oracle import ref_model
    if ref_model.load(kind) is None:
        pytest.skip("oracle/_ref/libref_densify*.so was never built (no reference tree, no prebuilt library)")


def test_one_train_step_through_alternating_resolutions_on_the_emulator(emu_lib_path):
    _need_reference_ops("cpu")
    cl = scene.make_cloud(240, 64, 48, 55.0, 55.0, seed=3, scale_k=0.35, n_views=3)   # levels: 64x48, 32x24, 16x12
    run_mixed_sequence(torch.device("cpu"), emu_lib_path, "emu", cl, "cpu")


@pytest.mark.gpu
def test_one_train_step_through_alternating_resolutions_at_C1_on_gpu():
    """BASELINE config C1's cloud (50 k Gaussians), keyframes at 640x480 / 320x240 / 160x120 in rotation, on the MI355X."""
    _need_reference_ops("cuda")
    cl = scene.make_config("C1", seed=0, n_views=3)
    run_mixed_sequence(torch.device("cuda:0"), None, "hip", cl, "cuda", " @C1")


# ---- two ranks, two sizes
WORKER_MIXED = r'''
import dataclasses, math, os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
entry.load_package()
from photo_slam_amd import scene
from photo_slam_amd.gaussian_model import GaussianModel
sys.path.insert(0, os.path.join(sys.argv[1], "photo-slam_amd", "host"))
import build_host
torch.ops.load_library(build_host.build("emu"))
ops = torch.ops.photoslam_amd
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
cl = scene.make_cloud(300, 96, 64, 80.0, 80.0, seed=3, scale_k=0.35, n_views=ws)
g0 = GaussianModel.from_cloud(cl, device="cpu")
h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(), g0.rotation_.detach(), 3,
                       float(cl.extent), torch.zeros(3))
ops.trainer_set_options(h, {"cameras_extent": float(cl.extent), "seed": 7.0})
ops.trainer_set_process_group(h, dist.group.WORLD.group_name, True)
if sys.argv[4] == "packed":
    ops.trainer_set_options(h, {"packed_exchange": 1.0, "pack_in_backward": 1.0})
    count_group = dist.new_group(backend="gloo")
    ops.trainer_set_count_group(h, count_group.group_name)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
for it in range(3):
    # rank r renders ITS keyframe at the pyramid level (r + it) % 2: full size or half size, the two ranks never the same
    f = (1.0, 0.5)[(rank + it) % 2]
    cam = dataclasses.replace(cl.cameras[rank], W=int(96 * f), H=int(64 * f))
    torch.manual_seed(100 + 10 * rank + it)
    gt = torch.rand(3, cam.H, cam.W)
    mask = torch.ones(3, cam.H, cam.W); mask[:, :2] = 0.0; mask[:, :, -3:] = 0.0
    ops.trainer_train_one_iteration(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), 2 * math.atan(cam.tanfovx),
                                    2 * math.atan(cam.tanfovy), cam.H, cam.W, gt * mask, mask)
out = {n: p.detach().numpy() for n, p in zip(["xyz", "features", "opacity", "scaling", "rotation"], ops.trainer_params(h))}
acc, den, maxr = ops.trainer_stats(h)
out["accum"] = acc.numpy(); out["denom"] = den.numpy(); out["maxr"] = maxr.numpy()
np.savez(os.path.join(sys.argv[3], f"rank{rank}.npz"), **out)
dist.barrier()
'''


def _launch_mixed(tmp_path, emu, port, form):
    script = tmp_path / "worker_mixed.py"
    script.write_text(WORKER_MIXED)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), str(script), ROOT, emu, str(tmp_path), form], env=dict(os.environ, MASTER_ADDR="127.0.0.1"),
                          timeout=900)
    return [{k: v.copy() for k, v in np.load(tmp_path / f"rank{r}.npz").items()} for r in range(2)]


def test_two_ranks_render_different_sizes_gloo(emu_lib_path, tmp_path):
    """Data-parallel keyframe batch whose two keyframes sit on different pyramid levels (96x64 and 48x32, swapping every iteration):
    the C++ host's exchange on 2 gloo ranks, dense and packed -- replicas bit-identical, both forms bit-identical, and equal to ONE
    process that renders both views and applies the mean gradient."""
    import dataclasses as dc
    from photo_slam_amd import loss_utils
    from photo_slam_amd.gaussian_renderer import GaussianRenderer
    dense = _launch_mixed(tmp_path, emu_lib_path, 29561, "dense")
    packed = _launch_mixed(tmp_path, emu_lib_path, 29563, "packed")
    names = ("xyz", "features", "opacity", "scaling", "rotation")
    for k in names:
        assert np.array_equal(dense[0][k], dense[1][k]), f"replicas diverged on {k}"
    for r in range(2):
        for k in names + ("accum", "denom", "maxr"):
            assert np.array_equal(dense[r][k], packed[r][k]), (r, k)
    assert not np.array_equal(dense[0]["denom"], dense[1]["denom"])   # (two views, two visibility sets)
    # one process, both views per iteration, mean gradient
    rp._LIB_OVERRIDE = emu_lib_path
    try:
        cl = scene.make_cloud(300, 96, 64, 80.0, 80.0, seed=3, scale_k=0.35, n_views=2)
        g = GaussianModel.from_cloud(cl, device="cpu")
        g.trainingSetup(GaussianOptimizationParams())
        for it in range(3):
            g.updateLearningRate(it + 1)
            grads = None
            for rank in range(2):
                f = (1.0, 0.5)[(rank + it) % 2]
                cam = dc.replace(cl.cameras[rank], W=int(96 * f), H=int(64 * f))
                torch.manual_seed(100 + 10 * rank + it)
                gt = torch.rand(3, cam.H, cam.W)
                mask = torch.ones(3, cam.H, cam.W)
                mask[:, :2] = 0.0
                mask[:, :, -3:] = 0.0
                img, _, _, _ = GaussianRenderer.render(GaussianKeyframe.from_camera(cam, "cpu"), cam.H, cam.W, g, GaussianPipelineParams(), torch.zeros(3))
                loss_utils.fused_l1_ssim_loss(img, gt * mask, mask, 0.2).backward()
                cur = [p.grad.clone() for p in g.params()]
                for p in g.params():
                    p.grad = None
                grads = cur if grads is None else [a + b for a, b in zip(grads, cur)]
            with torch.no_grad():
                for p, gr in zip(g.params(), grads):
                    p.grad = gr * 0.5
                g.optimizer_.step()
                g.optimizer_.zero_grad(set_to_none=True)
        for n, p in zip(names, g.params()):
            lr = dict(xyz=0.00016 * 4.5, features=0.0025 / 20.0, opacity=0.05, scaling=0.005, rotation=0.001)[n]
            assert np.abs(dense[0][n] - p.detach().numpy()).max() < 2e-3 * lr * 3, (n, np.abs(dense[0][n] - p.detach().numpy()).max() / lr)
    finally:
        rp._LIB_OVERRIDE = None
