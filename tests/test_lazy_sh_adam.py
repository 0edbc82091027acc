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
    # lower active degrees: the fused row kernel's other instantiations (whole parameter rows staged, basis values beyond the
    # active coefficients zero: shrows.h, wave_adam_rows_rank1)
    for degree in (0, 1, 2):
        parity.check_lazy_sh_adam(None, dev, cl, cams, np.zeros(3, np.float32), steps=7, window=3, sh_degree=degree, zero_gradient=True)
    parity.check_lazy_sh_adam(None, dev, cl, cams, np.array([0.1, 0.2, 0.3], np.float32), steps=7, window=3, sh_degree=1, exact=False)


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
    # new map points (GaussianModel::increasePcd) behind iterations 2, 5 and 6: the first before any rebuild (the tensors are
    # re-seated: every lazy row is flushed), the others appended IN PLACE into the arena the densification of iteration 4 left
    # (the lazy rows stay behind, the new rows join up to date)
    rng_ins = np.random.default_rng(3)
    new_pts = t(rng_ins.uniform([-3, -1.5, -3], [3, 1.5, 3], (40, 3)).astype(np.float32))
    new_cols = t(rng_ins.random((40, 3)).astype(np.float32))
    insert_after = (2, 5, 6)

    def same(a, b, what):
        if dev.type == "cpu":
            assert torch.equal(a, b), what
        else:
            # (two runs of the same program differ in the last bits of a gradient there; Adam turns a gradient whose SIGN is
            # rounding noise -- the freshly inserted points of increasePcd, seen by a handful of pixels -- into a whole step of
            # the learning rate: all but a few elements must agree)
            off = (a - b).abs() > 1e-4 * b.abs() + 1e-5 * float(b.abs().max())
            assert float(off.float().mean()) < 2e-3, (what, float(off.float().mean()))
            # ... and the few may be off by Adam steps of flipped sign only -- at most two learning rates per iteration (the
            # largest is the opacity's 0.05), nothing like a row left steps behind with a large gradient or a wrong row_step;
            # a moment cannot move further than a hundredth of the tensor's range on a noise-sized gradient
            bound = 2 * 0.05 * iterations if what[1] < 5 else 1e-2 * float(b.abs().max())
            assert float((a - b).abs().max()) <= bound, (what, float((a - b).abs().max()), bound)

    # C++ host
    results = {}
    in_place_appends = [0]
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
            if it + 1 in insert_after:
                rs0 = ops.trainer_features_row_step(h)
                ops.trainer_increase_pcd(h, new_pts, new_cols, it + 1, False)
                rs1 = ops.trainer_features_row_step(h)
                if window and rs0.numel() and rs1.numel():
                    # an append IN PLACE (the lazy state survived it): the rows that were behind are exactly as far behind as
                    # before, the new rows have taken every step so far
                    assert rs1.numel() == rs0.numel() + new_pts.shape[0]
                    assert torch.equal(rs1[:rs0.numel()], rs0), "an in-place increasePcd moved the lag of existing rows"
                    assert bool((rs1[rs0.numel():] == ops.trainer_steps(h)[1]).all()), "new rows must join at the current step"
                    in_place_appends[0] += 1
        results[window] = (losses, [x.detach().clone() for x in ops.trainer_params(h)], [x.clone() for x in ops.trainer_moments(h)])
        ops.trainer_destroy(h)
    assert results[3][1][0].shape[0] != P   # the model was rebuilt in between
    assert in_place_appends[0] >= 1          # ... and at least one insertion kept the lazy rows (checked above)
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
            losses = []
            for it in range(iterations):
                losses.append(float(ts.trainForOneIteration(kfs[it % 3], gts[it % 3], mask)))
                if it + 1 in insert_after:
                    g.increasePcd(new_pts, new_cols, it + 1)
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
        if dev.type == "cpu":
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-5)
        else:   # (see same(): a few elements of the freshly inserted points may sit a whole Adam step apart)
            off = (a - b).abs() > 1e-3 * b.abs() + 1e-5
            assert float(off.float().mean()) < 2e-3, float(off.float().mean())


def test_both_hosts_train_the_same_with_lazy_rows(emu_lib_path):
    from tests.test_cpp_host import load_host
    run_host_lazy_checks(load_host("emu"), torch.device("cpu"), emu_lib_path, P=200)


@pytest.mark.gpu
def test_both_hosts_train_the_same_with_lazy_rows_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from tests.test_cpp_host import load_host
    run_host_lazy_checks(load_host("hip"), torch.device("cuda:0"), None, P=20000, iterations=15)


def run_lazy_from_views_checks(dev, lib_path, P=700, n_views=3, steps=11, window=4, ahead=False):
    """gsr_sh_adam_from_views with lazy rows (the data-parallel step): rows no gathered view lights are left alone and catch up
    later (in a later from-views call that lights them, in their slice, or in the flush) -- bit-identical to the dense update
    that steps every row at every step.  Row ranges (two parts per step), learning rates that change per step, Gaussians that
    are never lit, rows lit after a long pause."""
    from photo_slam_amd import rasterize_points as rp
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g)
    xyz = (3.0 * r(P, 3)).to(dev)
    sh0, m0, v0 = (0.3 * r(P, 16, 3)).to(dev), (0.01 * r(P, 16, 3)).to(dev), (1e-4 * torch.rand(P, 16, 3, generator=g)).to(dev)
    never = torch.rand(P, generator=g) < 0.2
    base = dict(beta1=0.9, beta2=0.999, eps=1e-15)
    rp._LIB_OVERRIDE = lib_path
    try:
        eager = [t.clone() for t in (sh0, m0, v0)]
        lazy = [t.clone() for t in (sh0, m0, v0)]
        row_step = torch.full((P,), 5, dtype=torch.int32, device=dev)    # the tensor has taken 5 steps so far
        hist = []
        half = (P // 2 // 4) * 4
        for k in range(steps):
            step = 6 + k
            lr, lr_tail = 0.0025 * (1.0 + 0.1 * k), 0.000125 * (1.0 + 0.05 * k)
            lit = (torch.rand(n_views, P, generator=g) < (0.05 if 3 <= k < 8 else 0.35)) & ~never     # (a long pause for most rows)
            # (+0.0 where a view does not see the Gaussian: a negative zero is the "visible, zero gradient" marker of the
            # factored backward and would count as lit)
            views = torch.where(lit.unsqueeze(-1), r(n_views, P, 3), torch.zeros(())).to(dev)
            if k % 3 == 1:
                views[0, ::7, 0] = torch.where(never[::7].to(dev), views[0, ::7, 0], torch.full((), -0.0, device=dev))   # seen, zero gradient
                views[0, ::7, 1:] = torch.where(never[::7].to(dev).unsqueeze(-1), views[0, ::7, 1:], torch.zeros((), device=dev))
            centres = (2.0 * r(n_views, 3)).to(dev)
            d = dict(base, lr=lr, lr_tail=lr_tail, step=step)
            rp.shAdamFromViews(xyz, centres, views, 3, 1.0 / n_views, eager[0], dict(d, exp_avg=eager[1], exp_avg_sq=eager[2]))
            dl = dict(d, window=window, lr_past=[a for a, _ in hist], lr_tail_past=[b for _, b in hist])
            if ahead:   # the rotating catch-up AHEAD of the step's from-views calls (what gsr_backward does in the factored mode)
                rp.shAdamLazySlice(lazy[0], dict(dl, exp_avg=lazy[1], exp_avg_sq=lazy[2], row_step=row_step), ahead=True)
            for a, b in ((0, half), (half, P)):
                rp.shAdamFromViews(xyz[a:b], centres, views[:, a:b].contiguous(), 3, 1.0 / n_views, lazy[0][a:b],
                                   dict(dl, exp_avg=lazy[1][a:b], exp_avg_sq=lazy[2][a:b], row_step=row_step[a:b]))
            if not ahead:
                rp.shAdamLazySlice(lazy[0], dict(dl, exp_avg=lazy[1], exp_avg_sq=lazy[2], row_step=row_step))
            hist.insert(0, (lr, lr_tail))
            del hist[window:]
            assert int(row_step.max()) == step and int(row_step.min()) >= step - window + (0 if ahead else 1)
            if k == 4:
                assert int((row_step < step).sum()) > P // 10, "no row is behind: the case under test does not occur"
        rp.shAdamFlush(lazy[0], dict(base, lr=hist[0][0], lr_tail=hist[0][1], step=step, exp_avg=lazy[1], exp_avg_sq=lazy[2],
                                     row_step=row_step, window=window, lr_past=[a for a, _ in hist[1:]],
                                     lr_tail_past=[b for _, b in hist[1:]]))
        assert int(row_step.min()) == step
        for a, b, name in zip(lazy, eager, ("sh", "exp_avg", "exp_avg_sq")):
            assert torch.equal(a, b), name
        assert not torch.equal(eager[0], sh0)
    finally:
        rp._LIB_OVERRIDE = None


def run_adam_multi_checks(dev, lib_path, P=1237):
    """gsr_adam_step_multi == one gsr_adam_step per tensor (the four small tensors of a data-parallel step in one launch)."""
    from photo_slam_amd import capi, rasterize_points as rp
    g = torch.Generator().manual_seed(9)
    shapes = [(P, 4), (P, 3), (P, 3), (P, 1)]
    mk = lambda s, k: (k * torch.randn(*s, generator=g)).to(dev)
    params = [mk(s, 1.0) for s in shapes]
    grads = torch.cat([mk(s, 1e-3).reshape(-1) for s in shapes])          # slices of one buffer, as the backward pass leaves them
    gs, off = [], 0
    for s in shapes:
        n = s[0] * s[1]
        gs.append(grads[off:off + n].view(s))
        off += n
    m = [mk(s, 1e-2) for s in shapes]
    v = [(1e-4 * torch.rand(*s, generator=g)).to(dev) for s in shapes]
    lrs, steps = [0.001, 0.00016 * 4.5, 0.005, 0.05], [7, 7, 9, 3]
    rp._LIB_OVERRIDE = lib_path
    try:
        lib = rp._lib()
        one = [[t.clone() for t in ts] for ts in (params, m, v)]
        for k in range(4):
            capi.check(lib, lib.gsr_adam_step(one[0][k].data_ptr(), gs[k].data_ptr(), one[1][k].data_ptr(), one[2][k].data_ptr(),
                                              one[0][k].numel(), lrs[k], 0.9, 0.999, 1e-15, steps[k], 0, 0, lrs[k], None), "gsr_adam_step")
        multi = [[t.clone() for t in ts] for ts in (params, m, v)]
        rp.adamStepMulti([(multi[0][k], gs[k], multi[1][k], multi[2][k], lrs[k], steps[k]) for k in range(4)], 0.9, 0.999, 1e-15)
        for a, b in zip(one[0] + one[1] + one[2], multi[0] + multi[1] + multi[2]):
            assert torch.equal(a, b)
        assert not torch.equal(multi[0][0], params[0])
        # grad_scale: the gradient multiplied as it is read == a pre-scaled gradient
        scaled = [[t.clone() for t in ts] for ts in (params, m, v)]
        rp.adamStepMulti([(scaled[0][k], gs[k], scaled[1][k], scaled[2][k], lrs[k], steps[k]) for k in range(4)], 0.9, 0.999, 1e-15,
                         grad_scale=0.125)
        pre = [[t.clone() for t in ts] for ts in (params, m, v)]
        gpre = grads * 0.125
        off = 0
        ent = []
        for k, sh in enumerate(shapes):
            n = sh[0] * sh[1]
            ent.append((pre[0][k], gpre[off:off + n].view(sh), pre[1][k], pre[2][k], lrs[k], steps[k]))
            off += n
        rp.adamStepMulti(ent, 0.9, 0.999, 1e-15)
        for a, b in zip(scaled[0] + scaled[1] + scaled[2], pre[0] + pre[1] + pre[2]):
            assert torch.equal(a, b)
    finally:
        rp._LIB_OVERRIDE = None


def test_lazy_rows_in_the_from_views_step_equal_the_dense_update(emu_lib_path):
    run_lazy_from_views_checks(torch.device("cpu"), emu_lib_path)
    run_lazy_from_views_checks(torch.device("cpu"), emu_lib_path, P=333, n_views=1, steps=9, window=32)
    run_lazy_from_views_checks(torch.device("cpu"), emu_lib_path, steps=14, window=4, ahead=True)
    run_lazy_from_views_checks(torch.device("cpu"), emu_lib_path, P=450, n_views=2, steps=9, window=3, ahead=True)


def test_adam_step_multi_equals_separate_steps(emu_lib_path):
    run_adam_multi_checks(torch.device("cpu"), emu_lib_path)


@pytest.mark.gpu
def test_lazy_from_views_and_adam_multi_on_gpu():
    dev = torch.device("cuda:0")
    run_lazy_from_views_checks(dev, None, P=200_003, n_views=4, steps=13, window=4)
    run_lazy_from_views_checks(dev, None, P=50_000, n_views=8, steps=40, window=32)
    run_lazy_from_views_checks(dev, None, P=120_000, n_views=4, steps=21, window=5, ahead=True)
    run_adam_multi_checks(dev, None, P=300_001)
