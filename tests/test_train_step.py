"""Host logic of the measured train step (GaussianRenderer.render -> loss -> backward -> Adam) and of
the keyframe-batch data parallelism, on the host: CPU tensors, kernels through the wave64 emulator,
collectives through gloo (world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from photo_slam_amd import rasterize_points as rp
from photo_slam_amd import scene
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
from photo_slam_amd.gaussian_rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams, GaussianRenderer
from photo_slam_amd.trainer import TrainStep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def emu(emu_lib_path):
    rp._LIB_OVERRIDE = emu_lib_path
    yield emu_lib_path
    rp._LIB_OVERRIDE = None


def _setup(P=300, W=48, H=32, seed=3, n_views=1):
    cl = scene.make_cloud(P, W, H, 40.0, 40.0, seed=seed, scale_k=0.35, n_views=n_views)
    g = GaussianModel.from_cloud(cl, device="cpu")
    g.trainingSetup(GaussianOptimizationParams())
    kfs = [GaussianKeyframe.from_camera(c, "cpu") for c in cl.cameras]
    return cl, g, kfs


def test_render_returns_reference_tuple_and_grads_flow(emu):
    cl, g, kfs = _setup()
    bg = torch.zeros(3)
    img, viewspace, vis, radii = GaussianRenderer.render(kfs[0], 32, 48, g, GaussianPipelineParams(), bg)
    assert img.shape == (3, 32, 48) and viewspace.shape == (300, 3) and vis.dtype == torch.bool
    assert torch.equal(vis, radii > 0) and radii.dtype == torch.int32
    img.sum().backward()
    for p in g.params():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    assert viewspace.grad is not None and not viewspace.grad[:, 2].any()
    assert not g.xyz_.grad[~vis].any()


def test_rasterizer_argument_validation(emu):
    cl, g, kfs = _setup()
    s = GaussianRasterizationSettings(32, 48, 1.0, 1.0, torch.zeros(3), 1.0, kfs[0].world_view_transform_,
                                      kfs[0].full_proj_transform_, 3, kfs[0].camera_center_, False)
    r = GaussianRasterizer(s)
    with pytest.raises(RuntimeError, match="excatly one of either SHs or precomputed colors"):
        r(g.getXYZ(), g.getXYZ(), g.getOpacityActivation(), False, False, True, True, False)
    with pytest.raises(RuntimeError, match="exactly one of either scale/rotation pair"):
        r(g.getXYZ(), g.getXYZ(), g.getOpacityActivation(), True, False, True, False, False, shs=g.getFeatures())
    assert r.markVisibleGaussians(g.getXYZ()).dtype == torch.bool


def test_train_step_reduces_loss(emu):
    cl, g, kfs = _setup()
    torch.manual_seed(0)
    gt = torch.rand(3, 32, 48)
    mask = torch.ones(3, 32, 48)
    ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3))
    losses = [float(ts.trainForOneIteration(kfs[0], gt, mask)) for _ in range(8)]
    assert losses[-1] < losses[0], losses
    assert g.denom_.sum() > 0 and g.max_radii2D_.max() > 0


def test_fused_sh_adam_step_equals_separate_pass(emu):
    """The Adam step of the SH tensor inside the rasterizer's backward (default) == the separate gsr_adam_step pass."""
    out = []
    for fused in (True, False):
        cl, g, kfs = _setup()
        torch.manual_seed(0)
        gt = torch.rand(3, 32, 48)
        ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3), fused_sh_adam=fused)
        losses = [float(ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48))) for _ in range(4)]
        out.append((losses, [p.detach().clone() for p in g.params()], g.optimizer_.moments(g.features_),
                    g.optimizer_.state[id(g.features_)]["step"]))
    (l1, p1, m1, s1), (l2, p2, m2, s2) = out
    assert s1 == s2 == 4
    assert np.allclose(l1, l2, rtol=1e-6)
    for a, b in zip(p1, p2):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)
    for a, b in zip(m1, m2):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-12) and a.abs().sum() > 0


def test_small_gradients_share_one_buffer(emu):
    """The rasterizer's backward hands the gradients of rotation, position, scaling and opacity out as slices of one buffer
    and autograd adopts them as the leaves' .grad: a data-parallel trainer reduces the four with one collective."""
    from photo_slam_amd.trainer import _one_buffer
    cl, g, kfs = _setup()
    img, _, _, _ = GaussianRenderer.render(kfs[0], 32, 48, g, GaussianPipelineParams(), torch.zeros(3))
    img.sum().backward()
    small = [g.rotation_.grad, g.xyz_.grad, g.scaling_.grad, g.opacity_.grad]
    flat = _one_buffer(small)
    assert flat is not None and flat.numel() == 11 * 300
    assert flat.data_ptr() == g.rotation_.grad.data_ptr() and torch.equal(flat[4 * 300:7 * 300].view(300, 3), g.xyz_.grad)
    assert _one_buffer(small + [g.features_.grad]) is None and _one_buffer(small[:3]) is None
    assert _one_buffer([torch.zeros(4), torch.zeros(4)]) is None


WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
entry.load_package()
from photo_slam_amd import rasterize_points as rp, scene
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams
from photo_slam_amd.trainer import TrainStep
rp._LIB_OVERRIDE = sys.argv[2]
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
cl = scene.make_cloud(300, 48, 32, 40.0, 40.0, seed=3, scale_k=0.35, n_views=ws)
opt = GaussianOptimizationParams()
mode = sys.argv[5] if len(sys.argv) > 5 else ""
densify = mode in ("densify", "late")
if densify:
    opt.densify_from_iter_, opt.densification_interval_, opt.densify_grad_threshold_ = (1 if mode == "densify" else 10), 2, 2e-5
g = GaussianModel.from_cloud(cl, device="cpu"); g.trainingSetup(opt)
kf = GaussianKeyframe.from_camera(cl.cameras[rank], "cpu")
torch.manual_seed(100 + rank); gt = torch.rand(3, 32, 48)
ts = TrainStep(g, opt, GaussianPipelineParams(), torch.zeros(3), world_size=ws,
               factored_exchange=sys.argv[4] == "factored", densify=densify, cameras_extent=float(cl.extent), seed=7)
for _ in range(int(os.environ.get('GSR_TEST_ITERS', 3 if densify else 2))): ts.trainForOneIteration(kf, gt, torch.ones(3, 32, 48))
out = {n: p.detach().numpy() for n, p in zip(["xyz","features","opacity","scaling","rotation"], g.params())}
out["accum"] = g.xyz_gradient_accum_.numpy(); out["denom"] = g.denom_.numpy(); out["maxr"] = g.max_radii2D_.numpy()
np.savez(os.path.join(sys.argv[3], f"rank{rank}.npz"), **out)
dist.barrier()
'''


def _launch(tmp_path, emu, n_ranks, port, *worker_args, extra=""):
    script = tmp_path / "worker.py"
    script.write_text(WORKER + extra)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, emu, str(tmp_path)] +
                          list(worker_args), env=env, timeout=900)
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(n_ranks)]


def _single_process_batch(n_views, iterations=2):
    """One process that accumulates the mean gradient of all keyframes of the batch, same Adam: what the ranks must equal."""
    cl, g, kfs = _setup(n_views=n_views)
    mask = torch.ones(3, 32, 48)
    gts = []
    for rank in range(n_views):
        torch.manual_seed(100 + rank)
        gts.append(torch.rand(3, 32, 48))
    from photo_slam_amd import loss_utils
    per_view = None
    for it in range(1, iterations + 1):
        g.updateLearningRate(it)
        grads = None
        stats = []
        for kf, gt in zip(kfs, gts):
            img, vsp, vis, radii = GaussianRenderer.render(kf, 32, 48, g, GaussianPipelineParams(), torch.zeros(3))
            loss = loss_utils.fused_l1_ssim_loss(img, gt, mask, 0.2)
            loss.backward()
            cur = [p.grad.clone() for p in g.params()]
            for p in g.params():
                p.grad = None
            grads = cur if grads is None else [a + b for a, b in zip(grads, cur)]
            gn = torch.zeros_like(g.xyz_gradient_accum_)
            gn[vis] = torch.norm(vsp.grad[vis][:, :2], dim=-1, keepdim=True)
            stats.append((gn, vis.float().unsqueeze(1), torch.where(vis, radii.float(), torch.zeros(300))))
        with torch.no_grad():
            for p, gr in zip(g.params(), grads):
                p.grad = gr * (1.0 / n_views)
            for gn, cnt, rad in stats:
                g.xyz_gradient_accum_ += gn
                g.denom_ += cnt
                g.max_radii2D_ = torch.max(g.max_radii2D_, rad)
            g.optimizer_.step()
            g.optimizer_.zero_grad(set_to_none=True)
    return g


def _check_batch(ranks, g, lr_units=None):
    """lr_units: the comparison with the single process in units of the learning rate (the C++ host rebuilds tan(fov/2) from the
    keyframe's FoV as the reference's GaussianKeyframe does -- one ulp off the camera's own value, so its gradients differ from
    the Python mirror's by ~6e-7 relative, and Adam's second step turns that into up to ~1e-4 of a step on single elements)."""
    r0 = ranks[0]
    for r in ranks[1:]:
        for k in ("xyz", "features", "opacity", "scaling", "rotation"):
            assert np.array_equal(r0[k], r[k]), f"replicas diverged on {k}"
    names = ["xyz", "features", "opacity", "scaling", "rotation"]
    for n, p in zip(names, g.params()):
        if lr_units is None:
            assert np.allclose(r0[n], p.detach().numpy(), rtol=1e-5, atol=1e-7), n
        else:
            lr = dict(xyz=0.00016 * 4.5, features=0.0025 / 20.0, opacity=0.05, scaling=0.005, rotation=0.001)[n]
            assert np.abs(r0[n] - p.detach().numpy()).max() < lr_units * lr, (n, np.abs(r0[n] - p.detach().numpy()).max() / lr)
    # the statistics accumulate per rank (reduced only when densification consumes them): their SUM / MAX is the batch's
    assert np.allclose(sum(r["accum"] for r in ranks), g.xyz_gradient_accum_.numpy(), rtol=1e-5, atol=1e-9)
    assert np.array_equal(sum(r["denom"] for r in ranks), g.denom_.numpy())
    assert np.array_equal(np.maximum.reduce([r["maxr"] for r in ranks]), g.max_radii2D_.numpy())
    assert not np.array_equal(ranks[0]["denom"], ranks[1]["denom"])


@pytest.mark.parametrize("exchange", ["factored", "allreduce"])
def test_keyframe_batch_data_parallel_gloo(emu, tmp_path, exchange):
    """2 ranks x 1 keyframe each == 1 process accumulating both keyframes' gradients (mean), with the view-factored
    exchange (all-gathers of the colour gradients + local SH rebuild, the default) and with the plain all-reduce."""
    ranks = _launch(tmp_path, emu, 2, 29511 if exchange == "factored" else 29513, exchange)
    _check_batch(ranks, _single_process_batch(2))


def test_eight_keyframe_batch_gloo(emu, tmp_path):
    """The shape of BASELINE config C4 -- a batch of EIGHT keyframes, one per rank (here 8 gloo ranks on the host, a scaled
    cloud) -- against a single process that accumulates the eight views: replicas bit-identical, parameters equal."""
    ranks = _launch(tmp_path, emu, 8, 29521, "factored")
    _check_batch(ranks, _single_process_batch(8))


def test_factored_exchange_steps_sh_on_a_non_densifying_interval_iteration(emu, tmp_path):
    """densify on, densify_from_iter_ (10) beyond the iterations run, interval 2: iteration 2 is a multiple of the interval
    but does NOT densify -- the factored exchange must still rebuild and apply the SH gradient there (it used to
    drop the gathered views and skip the update).  Equal to the plain all-reduce run of the same schedule."""
    a = _launch(tmp_path, emu, 2, 29523, "factored", "late")
    fa = {k: a[0][k].copy() for k in a[0].files}
    b = _launch(tmp_path, emu, 2, 29525, "allreduce", "late")
    for k in ("xyz", "features", "opacity", "scaling", "rotation"):
        assert fa[k].shape == b[0][k].shape and np.allclose(fa[k], b[0][k], rtol=1e-5, atol=1e-7), k
    g3 = _single_process_batch(2, iterations=3)
    assert np.allclose(fa["features"], g3.features_.detach().numpy(), rtol=1e-5, atol=1e-7)


def test_keyframe_batch_densification_gloo(emu, tmp_path):
    """Two ranks that densify at the second iteration: the per-rank statistics are reduced right before, every rank takes the
    same decisions with the same samples, and the replicas (new size, Adam moments carried) stay bit-identical; the cloned /
    split / pruned counts equal a single process that saw both views."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER + """
np.savez(os.path.join(sys.argv[3], f"densify{rank}.npz"), info=np.array([ts.last_densify_[k] for k in ("cloned", "split", "pruned", "points")]))
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29515", str(script), ROOT, emu, str(tmp_path),
                           "factored", "densify"], env=env, timeout=600)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ("xyz", "features", "opacity", "scaling", "rotation"):
        assert r0[k].shape == r1[k].shape and np.array_equal(r0[k], r1[k]), f"replicas diverged on {k}"
    d0, d1 = np.load(tmp_path / "densify0.npz")["info"], np.load(tmp_path / "densify1.npz")["info"]
    assert np.array_equal(d0, d1) and d0[0] + d0[1] > 0 and d0[3] == r0["xyz"].shape[0] != 300
    # a single process with both views: same statistics at the second iteration, same densification decisions
    cl, g, kfs = _setup(n_views=2)
    opt = GaussianOptimizationParams()
    opt.densify_from_iter_, opt.densification_interval_, opt.densify_grad_threshold_ = 1, 2, 2e-5
    g.trainingSetup(opt)
    mask = torch.ones(3, 32, 48)
    gts = []
    for rank in range(2):
        torch.manual_seed(100 + rank)
        gts.append(torch.rand(3, 32, 48))
    from photo_slam_amd import loss_utils
    gen = torch.Generator().manual_seed(7)
    for it in range(1, 3):
        g.updateLearningRate(it)
        grads = None
        for kf, gt in zip(kfs, gts):
            img, vsp, vis, radii = GaussianRenderer.render(kf, 32, 48, g, GaussianPipelineParams(), torch.zeros(3),
                                                           view_stats=(g.xyz_gradient_accum_, g.denom_, g.max_radii2D_))
            loss_utils.fused_l1_ssim_loss(img, gt, mask, 0.2).backward()
            cur = [p.grad.clone() for p in g.params()]
            for p in g.params():
                p.grad = None
            grads = cur if grads is None else [a + b for a, b in zip(grads, cur)]
        with torch.no_grad():
            if it == 2:
                info = g.densifyAndPrune(opt.densify_grad_threshold_, 0.005, float(cl.extent), 0, generator=gen)
                assert [info[k] for k in ("cloned", "split", "pruned", "points")] == list(d0)
            else:
                for p, gr in zip(g.params(), grads):
                    p.grad = gr * 0.5
                g.optimizer_.step()
                g.optimizer_.zero_grad(set_to_none=True)


WORKER_CPP = r'''
import math, os, sys, numpy as np, torch, torch.distributed as dist
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
cl = scene.make_cloud(300, 48, 32, 40.0, 40.0, seed=3, scale_k=0.35, n_views=ws)
mode = sys.argv[5] if len(sys.argv) > 5 else ""
g0 = GaussianModel.from_cloud(cl, device="cpu")
bg = torch.zeros(3)
h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(), g0.rotation_.detach(), 3,
                       float(cl.extent), bg)
opts = {"cameras_extent": float(cl.extent), "seed": 7.0}
if mode in ("densify", "densify_morton"):
    opts.update({"densify": 1.0, "densify_from_iter": 1.0, "densification_interval": 2.0, "densify_grad_threshold": 2e-5,
                 "morton_reindex": 1.0 if mode == "densify_morton" else 0.0})
ops.trainer_set_options(h, opts)
# every collective of the step is issued by the C++ host from here on (host/src/keyframe_batch_exchange.cpp)
ops.trainer_set_process_group(h, dist.group.WORLD.group_name, sys.argv[4] in ("factored", "packed", "packed_late"))
if sys.argv[4] in ("packed", "packed_late"):   # the view-factored exchange in its packed form: only the rows a view sees travel
    # packed: the backward pass writes the message itself; packed_late: it is packed from the dense view behind the pass
    ops.trainer_set_options(h, {"packed_exchange": 1.0, "pack_in_backward": 1.0 if sys.argv[4] == "packed" else 0.0})
    if sys.argv[4] == "packed":   # the visible counts over a host group of their own (what bench.py does next to RCCL); packed_late: the main group
        count_group = dist.new_group(backend="gloo")
        ops.trainer_set_count_group(h, count_group.group_name)
cam = cl.cameras[rank]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
torch.manual_seed(100 + rank); gt = torch.rand(3, 32, 48)
for _ in range(int(os.environ.get("GSR_TEST_ITERS", 3 if mode in ("densify", "densify_morton") else 2))):
    ops.trainer_train_one_iteration(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), 2 * math.atan(cam.tanfovx),
                                    2 * math.atan(cam.tanfovy), cam.H, cam.W, gt, torch.ones(3, 32, 48))
out = {n: p.detach().numpy() for n, p in zip(["xyz","features","opacity","scaling","rotation"], ops.trainer_params(h))}
acc, den, maxr = ops.trainer_stats(h)
out["accum"] = acc.numpy(); out["denom"] = den.numpy(); out["maxr"] = maxr.numpy()
out["densify"] = np.array([int(x) for x in ops.trainer_last_densify(h)])
np.savez(os.path.join(sys.argv[3], f"rank{rank}.npz"), **out)
dist.barrier()
'''


def _launch_cpp(tmp_path, emu, n_ranks, port, *worker_args):
    script = tmp_path / "worker_cpp.py"
    script.write_text(WORKER_CPP)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, emu, str(tmp_path)] +
                          list(worker_args), env=env, timeout=900)
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(n_ranks)]


@pytest.mark.parametrize("exchange", ["factored", "allreduce", "packed"])
def test_cpp_host_drives_the_exchange_itself_gloo(emu, tmp_path, exchange):
    """The C++ host's data-parallel step (TrainStep::trainForOneIterationDataParallel on c10d::ProcessGroup,
    host/src/keyframe_batch_exchange.cpp: no collective is issued from Python) on 2 gloo ranks: replicas bit-identical, equal to
    one process that accumulates both keyframes, per-rank statistics as in the Python host's run."""
    ranks = _launch_cpp(tmp_path, emu, 2, {"factored": 29531, "allreduce": 29533, "packed": 29539}[exchange], exchange)
    _check_batch(ranks, _single_process_batch(2), lr_units=2e-3)


def test_packed_exchange_equals_the_dense_one_bit_for_bit_gloo(emu, tmp_path):
    """The packed form of the view-factored exchange (only the rows a view sees travel; include/gsr.h: gsr_pack_color_view)
    against the dense one, C++-driven: the same parameters and statistics BIT FOR BIT after the same iterations -- at 2 ranks,
    at the EIGHT ranks of BASELINE config C4's batch, and with a densification in the sequence."""
    dense_at = {}
    for n, port in ((2, 29541), (8, 29545)):
        dense = dense_at[n] = [{k: r[k].copy() for k in r.files} for r in _launch_cpp(tmp_path, emu, n, port, "factored")]
        packed = _launch_cpp(tmp_path, emu, n, port + 2, "packed")
        for r in range(n):
            for k in ("xyz", "features", "opacity", "scaling", "rotation", "accum", "denom", "maxr"):
                assert np.array_equal(dense[r][k], packed[r][k]), (n, r, k)
        for k in ("xyz", "features", "opacity", "scaling", "rotation"):
            assert np.array_equal(packed[0][k], packed[n - 1][k]), f"replicas diverged on {k}"
    late = _launch_cpp(tmp_path, emu, 2, 29553, "packed_late")          # the message packed behind the backward pass (gsr_pack_color_view)
    dense2 = dense_at[2]
    for r in range(2):
        for k in ("xyz", "features", "opacity", "scaling", "rotation", "accum", "denom", "maxr"):
            assert np.array_equal(late[r][k], dense2[r][k]), (r, k)
    dense = [{k: r[k].copy() for k in r.files} for r in _launch_cpp(tmp_path, emu, 2, 29549, "factored", "densify")]
    packed = _launch_cpp(tmp_path, emu, 2, 29551, "packed", "densify")
    assert packed[0]["xyz"].shape[0] != 300
    for r in range(2):
        for k in ("xyz", "features", "opacity", "scaling", "rotation", "densify"):
            assert np.array_equal(dense[r][k], packed[r][k]), (r, k)


def test_cpp_host_data_parallel_densification_gloo(emu, tmp_path):
    """... and with a densification in the sequence: the C++ host reduces the per-rank statistics itself right before, every
    rank takes the same decisions with the same samples; same counts as the Python host's data-parallel run."""
    ranks = _launch_cpp(tmp_path, emu, 2, 29535, "factored", "densify")
    for k in ("xyz", "features", "opacity", "scaling", "rotation"):
        assert ranks[0][k].shape == ranks[1][k].shape and np.array_equal(ranks[0][k], ranks[1][k]), f"replicas diverged on {k}"
    assert np.array_equal(ranks[0]["densify"], ranks[1]["densify"]) and ranks[0]["densify"][3] == ranks[0]["xyz"].shape[0] != 300
    py = tmp_path / "py"
    py.mkdir()
    script = tmp_path / "worker.py"
    script.write_text(WORKER + """
np.savez(os.path.join(sys.argv[3], f"densify{rank}.npz"), info=np.array([ts.last_densify_[k] for k in ("cloned", "split", "pruned", "points")]))
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29537", str(script), ROOT, emu, str(py),
                           "factored", "densify"], env=env, timeout=600)
    assert np.array_equal(np.load(py / "densify0.npz")["info"], ranks[0]["densify"])
    r0 = np.load(py / "rank0.npz")
    for k in ("xyz", "features", "opacity", "scaling", "rotation"):
        assert np.allclose(r0[k], ranks[0][k], rtol=1e-4, atol=1e-5), k
    # ... and with the Z-order layout (morton_reindex): every rank permutes alike -- identical replicas, the same decisions; the iteration
    # behind the densification then sums in another row order, so the rows are compared as sorted sets with a float tolerance
    ranks = [{k: r[k].copy() for k in r.files} for r in ranks]   # (the next launch writes the same files)
    zr = _launch_cpp(tmp_path, emu, 2, 29557, "factored", "densify_morton")
    for k in ("xyz", "features", "opacity", "scaling", "rotation"):
        assert np.array_equal(zr[0][k], zr[1][k]), f"replicas diverged on {k} (Z-order layout)"
    assert np.array_equal(zr[0]["densify"], ranks[0]["densify"])
    key = lambda r: np.lexsort(np.round(r["xyz"].astype(np.float64), 4).T[::-1])
    a, b = ranks[0], zr[0]
    for k in ("xyz", "opacity", "scaling", "rotation", "features"):
        assert np.allclose(a[k][key(a)], b[k][key(b)], rtol=1e-4, atol=1e-5), k


def test_densify_and_prune_keeps_model_consistent(emu):
    """densifyAndPrune (src/gaussian_model.cpp:716-815): selection rules, resulting order, Adam-state surgery."""
    cl, g, kfs = _setup(P=400)
    torch.manual_seed(0)
    gt = torch.rand(3, 32, 48)
    ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3))
    for _ in range(3):
        ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48))
    P0 = g.xyz_.shape[0]
    xyz0, op0, sc0 = g.xyz_.detach().clone(), g.opacity_.detach().clone(), g.scaling_.detach().clone()
    m0 = g.optimizer_.moments(g.xyz_)[0].clone()
    grads = (g.xyz_gradient_accum_ / g.denom_).nan_to_num(0.0).squeeze(-1)
    thr = float(grads[grads > 0].median())
    extent = 1.0
    smax = torch.exp(sc0).max(dim=1).values
    big = smax > g.percent_dense_ * extent
    clone_mask, split_mask = (grads >= thr) & ~big, (grads >= thr) & big
    gen = torch.Generator().manual_seed(5)
    info = g.densifyAndPrune(thr, 0.005, extent, 0, generator=gen)
    assert info["cloned"] == int(clone_mask.sum()) and info["split"] == int(split_mask.sum()) and info["split"] > 0
    n_keep = P0 - info["split"]
    # nothing is pruned by opacity here? compute expected survivors in reference order
    order = torch.cat([torch.arange(P0)[~split_mask], torch.arange(P0)[clone_mask], torch.arange(P0)[split_mask].repeat(2)])
    survive = torch.sigmoid(op0[order]).squeeze(-1) >= 0.005
    assert g.xyz_.shape[0] == int(survive.sum()) == info["points"]
    src = order[survive]
    is_child = (torch.arange(order.shape[0]) >= n_keep + info["cloned"])[survive]
    # originals and clones copy their source; children are displaced and shrunk by 1/(0.8*2)
    assert torch.equal(g.xyz_.detach()[~is_child], xyz0[src][~is_child])
    assert torch.allclose(g.scaling_.detach()[is_child], torch.log(torch.exp(sc0[src][is_child]) / 1.6), atol=1e-6)
    assert not torch.equal(g.xyz_.detach()[is_child], xyz0[src][is_child])
    # Adam moments: kept for originals, zero for every new Gaussian; statistics reset
    is_orig = (torch.arange(order.shape[0]) < n_keep)[survive]
    m1 = g.optimizer_.moments(g.xyz_)[0]
    assert torch.equal(m1[is_orig], m0[src][is_orig]) and not m1[~is_orig].any()
    for p in g.params():
        assert p.shape[0] == g.xyz_.shape[0] and p.requires_grad and p.is_leaf
    assert not g.denom_.any() and g.denom_.shape[0] == g.xyz_.shape[0]
    # training continues on the new set; the opacity reset as the reference ships it keeps the values and zeroes the
    # opacity moments (tests/test_densify_reference.py); clamp_to=0.01 is the reset 3DGS intended, on request only
    loss = ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48))
    assert torch.isfinite(loss)
    before = g.opacity_.detach().clone()
    g.resetOpacity()
    assert torch.allclose(g.opacity_.detach(), before, rtol=1e-4, atol=1e-5) and not g.optimizer_.moments(g.opacity_)[0].any()
    g.resetOpacity(clamp_to=0.01)
    assert float(torch.sigmoid(g.opacity_).max()) <= 0.01 + 1e-6
    assert torch.isfinite(ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48)))


def test_reference_cpu_train_step_equals_the_hip_train_step(emu):
    """a21 end to end: oracle/cpu_trainer.py -- the reference's step as it composes it (src/gaussian_trainer.cpp:45-133): CPU
    oracle rasterizer behind the autograd Function, activations / cat(dc, rest) in ATen, the reference's loss_utils.h (or its
    pinned mirror) through autograd, torch.optim.Adam with the six groups -- against TrainStep on the HIP kernels (emulated
    here): fused activations, fused loss + gradient, SH Adam inside backward, fused Adam.  Four iterations, same parameters."""
    from oracle import cpu_trainer
    cl, g, kfs = _setup(P=400)
    cam = cl.cameras[0]
    torch.manual_seed(5)
    gt = torch.rand(3, 32, 48)
    r = cpu_trainer.train(cl, cam, gt.numpy(), 4, threads=2)
    ref = r["model"]
    ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3))
    losses = [float(ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48)).detach()) for _ in range(4)]
    assert np.allclose(losses, r["losses"], rtol=2e-5), (losses, r["losses"])
    want = dict(xyz_=ref.xyz, features_=torch.cat([ref.features_dc, ref.features_rest], 1), opacity_=ref.opacity,
                scaling_=ref.scaling, rotation_=ref.rotation)
    for name, w in want.items():
        got = getattr(g, name).detach()
        # Adam normalises every gradient to a step of about lr: a gradient whose SIGN is rounding noise flips the step, so
        # compare in units of the learning rate (4 steps): the overwhelming majority must agree to 1e-3 of a step
        lr = {"xyz_": 0.00016 * cl.extent, "features_": 0.0025, "opacity_": 0.05, "scaling_": 0.005, "rotation_": 0.001}[name]
        err = (got - w.detach()).abs() / lr
        assert float((err > 1e-2).float().mean()) < 2e-3, (name, float(err.max()), float((err > 1e-2).float().mean()))
    assert torch.allclose(g.denom_, ref.denom) and torch.equal(g.max_radii2D_, ref.max_radii2D)
    assert torch.allclose(g.xyz_gradient_accum_, ref.xyz_gradient_accum, rtol=1e-4, atol=1e-9)


def test_fused_activations_match_the_reference_data_flow(emu):
    """raw_params extension: sigmoid/exp/normalize applied inside the rasterizer == activating with torch first."""
    cl, g, kfs = _setup(P=500)
    bg = torch.tensor([0.2, 0.1, 0.3])
    torch.manual_seed(1)
    dpix = torch.randn(3, 32, 48)
    outs = []
    for fuse in (False, True):
        for p in g.params():
            p.grad = None
        img, vsp, vis, radii = GaussianRenderer.render(kfs[0], 32, 48, g, GaussianPipelineParams(), bg, fuse_activations=fuse)
        (img * dpix).sum().backward()
        outs.append((img.detach().clone(), radii.clone(), [p.grad.clone() for p in g.params()], vsp.grad.clone()))
    (img0, r0, g0, v0), (img1, r1, g1, v1) = outs
    assert (r0 != r1).float().mean() < 1e-3          # exp() ulp differences may flip a ceil() very rarely
    assert (img0 - img1).abs().mean() < 1e-5
    for a, b in zip(g0 + [v0], g1 + [v1]):
        rel = (a - b).abs().sum() / (a.abs().sum() + 1e-30)
        assert rel < 1e-4, rel


def test_morton_reindex_is_a_permutation_of_the_reference_order(emu):
    """GaussianModel::morton_reindex_ (include/gsr.h: gsr_densify_gather_args.morton_scratch): densifyAndPrune lays the new set out
    along a Z-order curve of the Gaussians' positions.  The SAME rows -- parameters, the split children's sampled positions, both Adam
    moments, exist_since_iter -- as the reference's order gives, permuted; neighbouring rows are neighbours in space; training
    continues.  Python host and C++ host."""
    import math
    cl, _, kfs = _setup(P=2500)

    def rows(params, moments, exist):
        n = params[0].shape[0]
        cols = [p.detach().reshape(n, -1) for p in params] + [m.reshape(n, -1) for m in moments] + [exist.reshape(n, 1).float()]
        a = torch.cat(cols, 1).numpy()
        return a[np.lexsort(a.T[::-1])]

    def py_run(morton):
        g = GaussianModel.from_cloud(cl, device="cpu")
        g.trainingSetup(GaussianOptimizationParams())
        ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3))
        torch.manual_seed(0)
        gt = torch.rand(3, 32, 48)
        for _ in range(2):
            ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48))
        g.exist_since_iter_ = torch.arange(g.xyz_.shape[0], dtype=torch.int32)
        grads = (g.xyz_gradient_accum_ / g.denom_).nan_to_num(0.0)
        thr = float(grads[grads > 0].median())
        g.morton_reindex_ = morton
        dense = float(torch.exp(g.scaling_.detach()).max(1).values.median())   # half clone, half split
        info = g.densifyAndPrune(thr, 0.005, dense / g.percent_dense_, 0, generator=torch.Generator().manual_seed(5))
        moments = [t for p in g.params() for t in g.optimizer_.moments(p)]
        out = rows(g.params(), moments, g.exist_since_iter_), g.xyz_.detach().clone(), info
        assert torch.isfinite(ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48)))
        return out
    (r0, x0, i0), (r1, x1, i1) = py_run(False), py_run(True)
    assert i0 == i1 and i0["split"] > 0 and i0["cloned"] > 0
    assert np.array_equal(r0, r1) and not torch.equal(x0, x1)
    step = lambda x: float((x[1:] - x[:-1]).norm(dim=1).mean())
    assert step(x1) < 0.4 * step(x0), (step(x0), step(x1))
    # the C++ host: the same option through trainer_set_options
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    ops = load_host("emu")
    outs = []
    for morton in (0.0, 1.0):
        g0 = GaussianModel.from_cloud(cl, device="cpu")
        h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(), g0.rotation_.detach(), 3,
                               float(cl.extent), torch.zeros(3))
        ops.trainer_set_options(h, {"morton_reindex": morton})
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        cam = cl.cameras[0]
        torch.manual_seed(0)
        gt = torch.rand(3, 32, 48)
        for _ in range(2):
            ops.trainer_render_and_backward(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), 2 * math.atan(cam.tanfovx),
                                            2 * math.atan(cam.tanfovy), cam.H, cam.W, gt, torch.ones(3, 32, 48))
            ops.trainer_finish(h)
        acc, den, _ = ops.trainer_stats(h)
        grads = (acc / den).nan_to_num(0.0)
        thr = float(grads[grads > 0].median())
        ops.trainer_set_exist_since_iter(h, torch.arange(acc.shape[0], dtype=torch.int32))
        dense = float(torch.exp(ops.trainer_params(h)[3].detach()).max(1).values.median())
        info = [int(x) for x in ops.trainer_densify_and_prune(h, thr, 0.005, dense / 0.01, 0, 5)]
        outs.append((rows(ops.trainer_params(h), ops.trainer_moments(h), ops.trainer_exist_since_iter(h)), ops.trainer_params(h)[0].detach().clone(), info))
        ops.trainer_destroy(h)
    assert outs[0][2] == outs[1][2] and np.array_equal(outs[0][0], outs[1][0]) and step(outs[1][1]) < 0.4 * step(outs[0][1])


def test_persistent_workspace_changes_no_result_and_is_reused(emu):
    """TrainStep keeps the rasterizer's three scratch buffers across iterations (RasterWorkspace: grown with 50 % headroom, never
    shrunk) instead of allocating them per call as the reference's resizeFunctional does (src/rasterize_points.cu:28-34,71-76):
    the same parameters bit for bit, the same buffers from iteration to iteration, both hosts; views of different sizes share them."""
    import math
    cl, _, kfs = _setup(P=400, n_views=2)
    torch.manual_seed(1)
    gt = torch.rand(3, 32, 48)

    def py_run(persistent):
        g = GaussianModel.from_cloud(cl, device="cpu")
        g.trainingSetup(GaussianOptimizationParams())
        ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3))
        ts.persistent_workspace_ = persistent
        ptrs = []
        for i in range(4):
            ts.trainForOneIteration(kfs[i % 2], gt, torch.ones(3, 32, 48))
            ptrs.append([None if b is None else (b.data_ptr(), b.numel()) for b in ts.workspace_.bufs])
        return [p.detach().clone() for p in g.params()], ptrs
    (p1, ptr1), (p0, ptr0) = py_run(True), py_run(False)
    for a, b in zip(p1, p0):
        assert torch.equal(a, b)
    assert ptr0[-1] == [None, None, None]
    assert all(x is not None for x in ptr1[0])
    # the geometry and image buffers never move (same P, same H x W); the binning buffer moves only when it has to grow
    assert [p[0] for p in ptr1] == [ptr1[0][0]] * 4 and [p[2] for p in ptr1] == [ptr1[0][2]] * 4
    assert all(b[1][1] >= a[1][1] for a, b in zip(ptr1, ptr1[1:])) and len({p[1] for p in ptr1}) <= 2
    # a larger view grows the buffers once, a smaller one fits them
    g = GaussianModel.from_cloud(cl, device="cpu")
    g.trainingSetup(GaussianOptimizationParams())
    ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3))
    big = GaussianKeyframe.from_camera(scene.make_cloud(400, 96, 64, 40.0, 40.0, seed=3, scale_k=0.35).cameras[0], "cpu")
    ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48))
    small_img = ts.workspace_.bufs[2].numel()
    ts.trainForOneIteration(big, torch.rand(3, 64, 96), torch.ones(3, 64, 96))
    grown = [b.numel() for b in ts.workspace_.bufs]
    assert grown[2] > small_img
    ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48))
    assert [b.numel() for b in ts.workspace_.bufs] == grown

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    ops = load_host("emu")
    outs = []
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for persistent in (1.0, 0.0):
        g0 = GaussianModel.from_cloud(cl, device="cpu")
        h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(), g0.rotation_.detach(), 3,
                               float(cl.extent), torch.zeros(3))
        ops.trainer_set_options(h, {"persistent_workspace": persistent})
        for i in range(4):
            cam = cl.cameras[i % 2]
            ops.trainer_train_one_iteration(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), 2 * math.atan(cam.tanfovx),
                                            2 * math.atan(cam.tanfovy), cam.H, cam.W, gt, torch.ones(3, 32, 48))
        outs.append([p.detach().clone() for p in ops.trainer_params(h)])
        ops.trainer_destroy(h)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], p1):   # ... and the two hosts agree as they do without it
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_reorder_along_z_curve_permutes_the_whole_model(emu):
    """GaussianModel::reorderAlongZCurve (both hosts): parameters, both Adam moments, the three statistics arrays and
    exist_since_iter_ are the old rows at perm -- bit for bit --, neighbouring rows are neighbours in space, training continues,
    and the image of a view is the same up to the order of equal depths."""
    import math
    cl, _, kfs = _setup(P=1500)
    torch.manual_seed(0)
    gt = torch.rand(3, 32, 48)
    g = GaussianModel.from_cloud(cl, device="cpu")
    g.trainingSetup(GaussianOptimizationParams())
    ts = TrainStep(g, GaussianOptimizationParams(), GaussianPipelineParams(), torch.zeros(3))
    for _ in range(3):
        ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48))
    g.exist_since_iter_ = torch.arange(1500, dtype=torch.int32) * 3 + 1
    before = dict(params=[p.detach().clone() for p in g.params()], mom=[[t.clone() for t in g.optimizer_.moments(p)] for p in g.params()],
                  stats=[g.xyz_gradient_accum_.clone(), g.denom_.clone(), g.max_radii2D_.clone()], exist=g.exist_since_iter_.clone())
    img0 = GaussianRenderer.render(kfs[0], 32, 48, g, GaussianPipelineParams(), torch.zeros(3))[0].detach().clone()
    perm = g.reorderAlongZCurve()
    assert perm.dtype == torch.long and torch.equal(torch.sort(perm).values, torch.arange(1500))
    for p, old in zip(g.params(), before["params"]):
        assert torch.equal(p.detach(), old[perm])
    for p, (m0, v0) in zip(g.params(), before["mom"]):
        m, v = g.optimizer_.moments(p)
        assert torch.equal(m, m0[perm]) and torch.equal(v, v0[perm])
    for a, b in zip((g.xyz_gradient_accum_, g.denom_, g.max_radii2D_), before["stats"]):
        assert torch.equal(a, b[perm])
    assert torch.equal(g.exist_since_iter_, before["exist"][perm]) and g.denom_.abs().sum() > 0
    step = lambda x: float((x[1:] - x[:-1]).norm(dim=1).mean())
    assert step(g.xyz_.detach()) < 0.4 * step(before["params"][0])
    img1 = GaussianRenderer.render(kfs[0], 32, 48, g, GaussianPipelineParams(), torch.zeros(3))[0].detach()
    assert (img1 - img0).abs().max() < 1e-5
    assert torch.isfinite(ts.trainForOneIteration(kfs[0], gt, torch.ones(3, 32, 48)))

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    ops = load_host("emu")
    g0 = GaussianModel.from_cloud(cl, device="cpu")
    h = ops.trainer_create(g0.xyz_.detach(), g0.features_.detach(), g0.opacity_.detach(), g0.scaling_.detach(), g0.rotation_.detach(), 3,
                           float(cl.extent), torch.zeros(3))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    cam = cl.cameras[0]
    for _ in range(3):
        ops.trainer_train_one_iteration(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), 2 * math.atan(cam.tanfovx),
                                        2 * math.atan(cam.tanfovy), cam.H, cam.W, gt, torch.ones(3, 32, 48))
    ops.trainer_set_exist_since_iter(h, torch.arange(1500, dtype=torch.int32) * 3 + 1)
    p0 = [p.detach().clone() for p in ops.trainer_params(h)]
    m0 = [m.clone() for m in ops.trainer_moments(h)]
    s0 = [s.clone() for s in ops.trainer_stats(h)]
    perm_c = ops.trainer_reorder_along_z_curve(h)
    assert torch.equal(perm_c, perm)   # (the same positions after the same three iterations would be too strict: the hosts agree to 1e-5)  # noqa
    for a, b in zip(ops.trainer_params(h), p0):
        assert torch.equal(a.detach(), b[perm_c])
    for a, b in zip(ops.trainer_moments(h), m0):
        assert torch.equal(a, b[perm_c])
    for a, b in zip(ops.trainer_stats(h), s0):
        assert torch.equal(a, b[perm_c])
    assert torch.equal(ops.trainer_exist_since_iter(h), (torch.arange(1500, dtype=torch.int32) * 3 + 1)[perm_c])
    ops.trainer_train_one_iteration(h, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), 2 * math.atan(cam.tanfovx),
                                    2 * math.atan(cam.tanfovy), cam.H, cam.W, gt, torch.ones(3, 32, 48))
    ops.trainer_destroy(h)
