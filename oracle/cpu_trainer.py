"""The reference's train step on the HOST cores: "the reference CPU LibTorch path" of BASELINE.json configs[0].

TEST / BENCHMARK INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests import it; the product never does).

The reference has no CPU rasterizer (cuda_rasterizer/ is CUDA only), so the step is assembled as SURVEY.md 8(d) prescribes:
the CPU oracle (oracle/gsr_oracle.c, pinned bit for bit to the reference's kernel sources compiled for the host) behind the
same autograd Function shape as GaussianRasterizerFunction (src/gaussian_rasterizer.cpp:28-180), and around it LibTorch-CPU
ops exactly as the reference composes them:

  GaussianTrainer::trainingOnce, src/gaussian_trainer.cpp:45-133 ( = GaussianMapper::trainForOneIteration,
  src/gaussian_mapper.cpp:614-774 without the SLAM keyframe scheduling):
    updateLearningRate -> GaussianRenderer::render (activations sigmoid / exp / normalize, getFeatures = cat(dc.clone(),
    rest.clone()), src/gaussian_renderer.cpp:23-149, src/gaussian_model.cpp:48-71) -> (1 - lambda) L1 + lambda (1 - SSIM)
    (include/loss_utils.h; the reference's OWN header compiled into torch ops when oracle/_ref/libref_loss.so exists, its
    pinned torch mirror otherwise) -> backward -> max_radii2D / addDensificationStats (:109-117) -> Adam with six groups,
    eps 1e-15 (src/gaussian_model.cpp:477-510; torch.optim.Adam runs the same ATen kernels torch::optim::Adam does) ->
    zero_grad.

torch.set_num_threads(host cores) for the ATen part, OpenMP over Gaussians / tiles inside the oracle.
"""
import os
import time

import numpy as np
import torch

from . import oracle


def _loss_ops():
    """(l1_loss, ssim, kind): the reference's include/loss_utils.h behind torch ops if the library was built (it travels to
    the GPU boxes prebuilt), else the torch mirror pinned against it by tests/test_reference_pinning.py."""
    from . import build_ref
    path = build_ref.build_loss()
    if path and os.path.exists(path):
        try:
            torch.ops.load_library(path)
            ops = torch.ops.photoslam_reference
            return ops.l1_loss, ops.ssim, "reference header include/loss_utils.h"
        except Exception:
            pass
    import importlib
    lu = importlib.import_module("photo_slam_amd.loss_utils")
    return lu.l1_loss, lu.ssim, "torch mirror of include/loss_utils.h"


class OracleRasterizerFunction(torch.autograd.Function):
    """GaussianRasterizerFunction (src/gaussian_rasterizer.cpp:28-180) on the CPU oracle: same inputs, same gradient slots
    (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, cam, bg, sh_degree):
        n = lambda t: np.ascontiguousarray(t.detach().numpy())
        res, color, radii = oracle.forward(bg, n(means3D), n(opacities), cam.viewmatrix, cam.projmatrix, cam.campos, cam.tanfovx,
                                           cam.tanfovy, cam.H, cam.W, shs=n(sh), sh_degree=sh_degree, scales=n(scales),
                                           rotations=n(rotations))
        ctx.res = res
        ctx.mark_non_differentiable(r := torch.from_numpy(radii))
        return torch.from_numpy(color), r

    @staticmethod
    def backward(ctx, grad_color, _grad_radii):
        g = oracle.backward(ctx.res, np.ascontiguousarray(grad_color.numpy()))
        ctx.res.free()
        t = torch.from_numpy
        return (t(g["dL_dmeans3D"]), t(g["dL_dmeans2D"]), t(g["dL_dsh"]), t(g["dL_dopacity"]), t(g["dL_dscales"]),
                t(g["dL_drotations"]), None, None, None)


class CpuModel:
    """The six leaves of GaussianModel (include/gaussian_model.h:146-155) + Adam as trainingSetup builds it."""

    def __init__(self, cloud, spatial_lr_scale):
        leaf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).clone().requires_grad_(True)
        self.xyz, self.features_dc, self.features_rest = leaf(cloud.xyz), leaf(cloud.features_dc), leaf(cloud.features_rest)
        self.opacity, self.scaling, self.rotation = leaf(cloud.opacity), leaf(cloud.scaling), leaf(cloud.rotation)
        P = self.xyz.shape[0]
        self.max_radii2D = torch.zeros(P)
        self.xyz_gradient_accum = torch.zeros(P, 1)
        self.denom = torch.zeros(P, 1)
        self.spatial_lr_scale = spatial_lr_scale
        f32 = lambda x: float(np.float32(x))
        # GaussianOptimizationParams defaults (include/gaussian_parameters.h:61-96), src/gaussian_model.cpp:477-510
        self.lr_init, self.lr_final, self.max_steps = f32(0.00016) * spatial_lr_scale, f32(0.0000016) * spatial_lr_scale, 30000
        self.optimizer = torch.optim.Adam([
            dict(params=[self.xyz], lr=self.lr_init), dict(params=[self.features_dc], lr=f32(0.0025)),
            dict(params=[self.features_rest], lr=f32(0.0025) / 20.0), dict(params=[self.opacity], lr=f32(0.05)),
            dict(params=[self.scaling], lr=f32(0.005)), dict(params=[self.rotation], lr=f32(0.001))], lr=0.0, eps=1e-15)

    def update_learning_rate(self, step):
        """exponLrFunc, src/gaussian_model.cpp:1118-1131"""
        t = min(max(step / self.max_steps, 0.0), 1.0)
        lr = float(np.exp(np.log(self.lr_init) * (1 - t) + np.log(self.lr_final) * t))
        self.optimizer.param_groups[0]["lr"] = lr
        return lr


def render(model, cam, bg, sh_degree=3):
    """GaussianRenderer::render, src/gaussian_renderer.cpp:23-149 (compute_cov3D_ / convert_SHs_ off, as every shipped config)"""
    means2D = torch.zeros_like(model.xyz, requires_grad=True)     # screenspace_points, :41-48
    features = torch.cat([model.features_dc.clone(), model.features_rest.clone()], 1)   # getFeatures, gaussian_model.cpp:63-66
    color, radii = OracleRasterizerFunction.apply(model.xyz, means2D, features, torch.sigmoid(model.opacity),
                                                  torch.exp(model.scaling), torch.nn.functional.normalize(model.rotation), cam,
                                                  bg, sh_degree)
    return color, means2D, radii > 0, radii


def train(cloud, cam, gt_image, iterations, warmup=0, lambda_dssim=0.2, threads=None, time_budget_s=None, min_iterations=3):
    """Runs warmup + iterations steps (fewer when time_budget_s runs out, but at least min_iterations measured ones);
    returns dict(seconds per iteration list, loss list, ...).  gt_image: [3,H,W] numpy."""
    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    t_start = time.perf_counter()
    oracle.build()
    l1_loss, ssim, loss_kind = _loss_ops()
    model = CpuModel(cloud, cloud.extent)
    gt = torch.from_numpy(np.ascontiguousarray(gt_image, np.float32))
    bg = np.zeros(3, np.float32)
    times, losses = [], []
    raster_times = []
    phases = []   # (render forward, loss forward, backward (loss + rasterizer), statistics + Adam) seconds
    for it in range(1, warmup + iterations + 1):
        t0 = time.perf_counter()
        model.update_learning_rate(it)
        image, viewspace, visibility, radii = render(model, cam, bg)
        t_r = time.perf_counter()
        Ll1 = l1_loss(image, gt)
        loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim(image, gt))
        t_l = time.perf_counter()
        loss.backward()
        t_b = time.perf_counter()
        with torch.no_grad():
            losses.append(float(loss))      # ema_loss_for_log: the per-iteration host read, :92
            model.max_radii2D[visibility] = torch.max(model.max_radii2D[visibility], radii[visibility].float())
            model.xyz_gradient_accum[visibility] += torch.norm(viewspace.grad[visibility][:, :2], dim=-1, keepdim=True)
            model.denom[visibility] += 1
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        dt = time.perf_counter() - t0
        if it > warmup:
            times.append(dt)
            raster_times.append(t_r - t0)
            phases.append((t_r - t0, t_l - t_r, t_b - t_l, t0 + dt - t_b))
            if time_budget_s is not None and len(times) >= min_iterations and time.perf_counter() - t_start > time_budget_s:
                break
    return dict(seconds=times, forward_seconds=raster_times, losses=losses, loss_ops=loss_kind, threads=threads,
                oracle_threads=oracle.get_threads(), model=model,
                phase_seconds_median=[round(float(np.median([p[k] for p in phases])), 4) for k in range(4)])


# ---------------------------------------------------------------------------------------------------------------------------
# The same step over a SEQUENCE of keyframes with the map maintenance of the loop: densifyAndPrune every
# densification_interval iterations after densify_from_iter, resetOpacity every opacity_reset_interval
# (src/gaussian_trainer.cpp:108-127 = src/gaussian_mapper.cpp:711-735).  The two maintenance calls are the REFERENCE's OWN
# member functions (oracle/_ref/libref_densify{,_cuda}.so: src/gaussian_model.cpp:556-565, 716-815 extracted verbatim,
# oracle/ref_densify.cpp) on this model's tensors and Adam state -- nothing of them is restated here.


def _adam_state(model):
    """(exp_avg[6], exp_avg_sq[6], steps[6]) of the six leaves in the reference's group order; a leaf that never stepped
    has zero moments and step 0."""
    m, v, steps = [], [], []
    for p in (model.xyz, model.features_dc, model.features_rest, model.opacity, model.scaling, model.rotation):
        st = model.optimizer.state.get(p)
        if st:
            m.append(st["exp_avg"]), v.append(st["exp_avg_sq"]), steps.append(int(st["step"]))
        else:
            m.append(torch.zeros_like(p)), v.append(torch.zeros_like(p)), steps.append(0)
    return m, v, steps


def _reference_state(model, ref_model, dev):
    m, v, steps = _adam_state(model)
    to = lambda t: t.detach().to(dev)
    leaves = (model.xyz, model.features_dc, model.features_rest, model.opacity, model.scaling, model.rotation)
    P = model.xyz.shape[0]
    exist = getattr(model, "exist_since_iter", None)
    if exist is None:
        exist = torch.zeros(P, dtype=torch.int32)
    return ref_model.State([to(p) for p in leaves], [to(t) for t in m], [to(t) for t in v], steps, to(model.xyz_gradient_accum),
                           to(model.denom), to(model.max_radii2D), to(exist))


def _install_leaf(model, index, name, value, exp_avg, exp_avg_sq, step):
    """Replace one leaf (and its Adam state) the way replaceTensorToOptimizer / densificationPostfix do: a fresh leaf without a
    gradient, the given moments, the old step counter."""
    old = getattr(model, name)
    model.optimizer.state.pop(old, None)
    leaf = value.detach().cpu().clone().requires_grad_(True)
    setattr(model, name, leaf)
    model.optimizer.param_groups[index]["params"][0] = leaf
    if step >= 0:
        model.optimizer.state[leaf] = dict(step=torch.tensor(float(step)), exp_avg=exp_avg.detach().cpu().clone(),
                                           exp_avg_sq=exp_avg_sq.detach().cpu().clone())


_LEAVES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")


def reference_densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size, kind="cpu", percent_dense=0.01):
    """GaussianModel::densifyAndPrune of the reference (src/gaussian_model.cpp:716-815) on this model.  kind "cuda": the HIP
    build of the same code on the GPU box (at::normal then draws from the device generator the GPU hosts seed identically)."""
    from . import ref_model
    ops = ref_model.load(kind)
    if ops is None:
        raise RuntimeError("oracle/_ref/libref_densify*.so was never built")
    dev = torch.device("cuda:0" if kind == "cuda" else "cpu")
    new = ref_model.densify_and_prune(ops, _reference_state(model, ref_model, dev), percent_dense, max_grad, min_opacity, extent,
                                      max_screen_size)
    for i, name in enumerate(_LEAVES):
        _install_leaf(model, i, name, new.params[i], new.exp_avg[i], new.exp_avg_sq[i], new.steps[i])
    model.xyz_gradient_accum, model.denom = new.accum.cpu().clone(), new.denom.cpu().clone()
    model.max_radii2D, model.exist_since_iter = new.max_radii2D.cpu().clone(), new.exist_since_iter.cpu().clone()
    return model.xyz.shape[0]


def reference_reset_opacity(model, kind="cpu"):
    """GaussianModel::resetOpacity of the reference (:556-565): only the opacity leaf is replaced (no gradient: the
    optimizer step that follows skips it, the other five groups step)."""
    from . import ref_model
    ops = ref_model.load(kind)
    if ops is None:
        raise RuntimeError("oracle/_ref/libref_densify*.so was never built")
    dev = torch.device("cuda:0" if kind == "cuda" else "cpu")
    new = ref_model.reset_opacity(ops, _reference_state(model, ref_model, dev))
    _install_leaf(model, 3, "opacity", new.params[3], new.exp_avg[3], new.exp_avg_sq[3], new.steps[3])


def train_sequence(cloud, cams, gt_images, iterations, densification_interval=0, densify_from_iter=0, opacity_reset_interval=0,
                   densify_until_iter=15000, densify_grad_threshold=0.0002, min_opacity=0.005, lambda_dssim=0.2, seed=0,
                   kind="cpu", threads=None, on_iteration=None, keyframe_order=None, step_on_last_iteration=True, plan=None):
    """trainingOnce's loop (src/gaussian_trainer.cpp:45-133) over the keyframes cams[(it - 1) % len(cams)], it = 1..iterations,
    with the reference's own densifyAndPrune / resetOpacity on the schedule of :108-127.  Returns dict(losses, points per
    iteration, densified_at, reset_at, model).  The split samples come from the default generator of `kind`'s device,
    seeded once with `seed` (the hosts under test seed a generator of their own identically).
    keyframe_order: the keyframe index of every iteration instead of the cycle (trainingOnce draws it with std::rand, :59);
    step_on_last_iteration=False: trainingOnce's `if (iteration < opt.iterations_)` around the optimizer step (:129).
    plan (instead of cams / gt_images): per iteration a dict(cam, gt [3,H,W], mask [3,H,W] or None, lr_step or None) -- the SLAM
    flavour of the step, GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:631-699): the keyframe rendered at the size of
    its current Gaussian-pyramid level (cam.H, cam.W) against that level's image, `rendered * mask` (:692), and the position
    learning rate at the keyframe's use count (:663-671) instead of the iteration."""
    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    oracle.build()
    l1_loss, ssim, loss_kind = _loss_ops()
    model = CpuModel(cloud, cloud.extent)
    model.exist_since_iter = torch.zeros(model.xyz.shape[0], dtype=torch.int32)
    gts = [torch.from_numpy(np.ascontiguousarray(g, np.float32)) for g in gt_images] if plan is None else None
    bg = np.zeros(3, np.float32)
    (torch.cuda.manual_seed if kind == "cuda" else torch.manual_seed)(seed)
    losses, points, densified_at, reset_at = [], [], [], []
    for it in range(1, iterations + 1):
        if plan is not None:
            step = plan[it - 1]
            model.update_learning_rate(it if step.get("lr_step") is None else min(int(step["lr_step"]), model.max_steps))
            image, viewspace, visibility, radii = render(model, step["cam"], bg)
            if step.get("mask") is not None:
                image = image * torch.from_numpy(np.ascontiguousarray(step["mask"], np.float32))       # :692
            gt = torch.from_numpy(np.ascontiguousarray(step["gt"], np.float32))
        else:
            model.update_learning_rate(it)
            k = keyframe_order[it - 1] if keyframe_order is not None else (it - 1) % len(cams)
            image, viewspace, visibility, radii = render(model, cams[k], bg)
            gt = gts[k]
        loss = (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))
        loss.backward()
        with torch.no_grad():
            losses.append(float(loss))
            if it < densify_until_iter:
                model.max_radii2D[visibility] = torch.max(model.max_radii2D[visibility], radii[visibility].float())
                model.xyz_gradient_accum[visibility] += torch.norm(viewspace.grad[visibility][:, :2], dim=-1, keepdim=True)
                model.denom[visibility] += 1
                if densification_interval and it > densify_from_iter and it % densification_interval == 0:
                    size_threshold = 20 if (opacity_reset_interval and it > opacity_reset_interval) else 0     # :120
                    reference_densify_and_prune(model, densify_grad_threshold, min_opacity, float(cloud.extent), size_threshold, kind)
                    densified_at.append(it)
                if opacity_reset_interval and it % opacity_reset_interval == 0:
                    reference_reset_opacity(model, kind)
                    reset_at.append(it)
            if step_on_last_iteration or it < iterations:
                model.optimizer.step()      # leaves without a gradient (fresh ones) are skipped, their step counters rest
                model.optimizer.zero_grad(set_to_none=True)
            points.append(int(model.xyz.shape[0]))
        if on_iteration is not None:
            on_iteration(it, model, losses[-1])
    return dict(losses=losses, points=points, densified_at=densified_at, reset_at=reset_at, model=model, loss_ops=loss_kind)
