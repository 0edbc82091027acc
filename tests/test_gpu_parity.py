"""Parity tests proper: the hand-written HIP path (libgsr_hip.so, through the C-ABI) against
the CPU oracle on the same seeded inputs, on a real MI355X.  Run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import parity
from photo_slam_amd import capi, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    lib = capi.load()  # raises when the HIP extension is missing: no fallback
    assert lib.gsr_backend() == b"hip-gfx950"
    return torch.device("cuda:0")


BINNINGS = (("depth-first", 32), ("tile-first", 64))   # GSR_BINNING_DEPTH_FIRST / GSR_BINNING_TILE_FIRST (include/gsr.h)


def _check(oracle, dev, cl, cam, bg, seed=0, **kw):
    """Every stage of the HIP path against the oracle -- once per binning arrangement (the Gaussians sorted by depth in front of
    the emission / every tile's list sorted by depth behind the tile sort): both must give the reference's lists bit for bit."""
    rng = np.random.default_rng(seed)
    dpix = rng.standard_normal((3, cam.H, cam.W)).astype(np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, dL_dpix=dpix, **kw)
    for name, flag in BINNINGS:
        r = parity.run_backend(None, dev, cl, cam, bg, dL_dpix=dpix, flags=flag, **kw)
        rep = parity.compare(r, ores, ocolor, oradii, ograds, cam,
                             use_colors_precomp=kw.get("use_colors_precomp", False),
                             use_cov3D_precomp=kw.get("use_cov3D_precomp", False))
        print(name, dict(P=ores.P, V=int((oradii > 0).sum()), R=ores.R), rep)
    return r, ores


@pytest.mark.parametrize("P,W,H,fx,seed,k", [
    (600, 64, 48, 50.0, 1, 0.35),          # the emulator-sized case
    (20000, 320, 240, 300.0, 3, 0.12),
    (50000, 640, 480, 535.4, 0, 0.2),      # BASELINE config C1 shape
    (200000, 333, 207, 260.0, 5, 0.15),    # ragged image size (partial tiles on both edges)
])
def test_forward_backward_matches_oracle(oracle, dev, P, W, H, fx, seed, k):
    cl = scene.make_cloud(P, W, H, fx, fx, seed=seed, scale_k=k)
    _check(oracle, dev, cl, cl.cameras[0], np.array([0.2, 0.5, 0.1], np.float32), seed)


def test_config_C2_full_size(oracle, dev):
    """BASELINE config C2: 500k Gaussians at 1200x680, full size, all stages against the oracle."""
    cl = scene.make_config("C2", seed=0)
    r, ores = _check(oracle, dev, cl, cl.cameras[0], np.zeros(3, np.float32))
    assert ores.R > 1_000_000


def test_config_C3_full_size(oracle, dev):
    """The headline config (BASELINE.json configs[2], the one bench.py times): 2 M Gaussians at 1920x1080, every stage
    against the oracle at full size (the oracle needs a few seconds on the GPU box's host cores)."""
    cl = scene.make_config("C3", seed=0)
    r, ores = _check(oracle, dev, cl, cl.cameras[0], np.zeros(3, np.float32))
    assert ores.R > 4_000_000 and ores.T == 120 * 68


def test_config_C4_view_full_size(oracle, dev):
    """One keyframe of BASELINE config C4 (2 M Gaussians at 640x480, TUM intrinsics): the per-rank work of the 8-GPU batch."""
    cl = scene.make_config("C4", seed=0, n_views=8)
    for v in (0, 7):
        r, ores = _check(oracle, dev, cl, cl.cameras[v], np.zeros(3, np.float32), seed=v)
        assert ores.T == 40 * 30


def test_config_C5_view_full_size(oracle, dev):
    """One keyframe of BASELINE config C5 (4 M Gaussians at 752x480, EuRoC intrinsics, SH degree 3)."""
    cl = scene.make_config("C5", seed=0)
    r, ores = _check(oracle, dev, cl, cl.cameras[0], np.array([0.1, 0.1, 0.1], np.float32))
    assert ores.P == 4_000_000 and ores.T == 47 * 30


def test_no_gaussians(dev):
    """P == 0 (src/rasterize_points.cu:68,81): a valid no-op -- the image is all ZERO (not the background: out_color is
    created as zeros and the launch is skipped), zero instances, empty radii and empty gradients."""
    from photo_slam_amd import rasterize_points as rp
    e = torch.empty(0, device=dev)
    z3 = torch.empty((0, 3), device=dev)
    bg = torch.tensor([0.3, 0.6, 0.9], device=dev)
    cam = scene.make_cloud(1, 96, 64, 80.0, 80.0, seed=0).cameras[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    R, color, radii, geom, binning, img = rp.RasterizeGaussiansCUDA(
        bg, z3, e, torch.empty((0, 1), device=dev), z3, torch.empty((0, 4), device=dev), 1.0, e, t(cam.viewmatrix),
        t(cam.projmatrix), cam.tanfovx, cam.tanfovy, cam.H, cam.W, torch.empty((0, 16, 3), device=dev), 3, t(cam.campos), False)
    assert R == 0 and radii.numel() == 0 and tuple(color.shape) == (3, cam.H, cam.W)
    assert not color.any()
    g = rp.RasterizeGaussiansBackwardCUDA(bg, z3, radii, e, z3, torch.empty((0, 4), device=dev), 1.0, e, t(cam.viewmatrix),
                                          t(cam.projmatrix), cam.tanfovx, cam.tanfovy, torch.ones_like(color),
                                          torch.empty((0, 16, 3), device=dev), 3, t(cam.campos), geom, R, binning, img)
    assert all(x is None or x.numel() == 0 for x in g)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        rp.RasterizeGaussiansCUDA(bg, torch.empty((4,), device=dev), e, e, e, e, 1.0, e, t(cam.viewmatrix), t(cam.projmatrix),
                                  cam.tanfovx, cam.tanfovy, cam.H, cam.W, e, 3, t(cam.campos), False)


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_lower_sh_degrees(oracle, dev, degree):
    cl = scene.make_cloud(30000, 256, 192, 200.0, 200.0, seed=8, scale_k=0.15)
    _check(oracle, dev, cl, cl.cameras[0], np.ones(3, np.float32), sh_degree=degree)


@pytest.mark.gpu
@pytest.mark.parametrize("degree,coeffs", [(0, 1), (1, 4), (2, 9)])
def test_compact_sh_layouts(oracle, dev, degree, coeffs):
    # SH tensors that are not [P,16,3]: per-lane row access in preprocess fwd / bwd instead of the LDS row movers
    cl = scene.make_cloud(20000, 256, 192, 200.0, 200.0, seed=10, scale_k=0.15)
    _check(oracle, dev, cl, cl.cameras[0], np.array([0.2, 0.4, 0.6], np.float32), sh_degree=degree, sh_coeffs=coeffs)


@pytest.mark.gpu
@pytest.mark.parametrize("degree,coeffs,n_views", [(3, None, 4), (2, None, 2), (0, None, 2), (1, 4, 3)])
def test_view_factored_sh_gradient(dev, degree, coeffs, n_views):
    # keyframe-batch data parallelism: dL_dcolor_view + gsr_sh_grad_from_views == mean of the per-view dL_dsh, and the
    # factored backward leaves every other gradient bit-identical (parity.check_view_factored)
    cl = scene.make_cloud(40000, 320, 240, 250.0, 250.0, seed=12, scale_k=0.15, n_views=n_views)
    parity.check_view_factored(None, dev, cl, np.array([0.1, 0.2, 0.3], np.float32), sh_degree=degree, sh_coeffs=coeffs)


@pytest.mark.gpu
def test_fused_sh_adam(dev):
    # optimizer-in-backward for the SH tensor == backward + gsr_adam_step (parity.check_fused_sh_adam); the culled Gaussians'
    # rows are updated on the library's second stream, the visible ones by the row kernel
    cl = scene.make_cloud(50000, 320, 240, 250.0, 250.0, seed=14, scale_k=0.15)
    parity.check_fused_sh_adam(None, dev, cl, np.array([0.1, 0.2, 0.3], np.float32))


@pytest.mark.gpu
def test_backward_may_follow_one_forward_more_than_once(dev):
    cl = scene.make_cloud(80_000, 640, 480, 400.0, 400.0, seed=16)
    parity.check_backward_twice(None, dev, cl, np.array([0.1, 0.2, 0.3], np.float32))


@pytest.mark.gpu
def test_fused_geom_adam(dev):
    cl = scene.make_cloud(60_000, 320, 240, 250.0, 250.0, seed=12)
    parity.check_fused_geom_adam(None, dev, cl, np.array([0.1, 0.2, 0.3], np.float32))


@pytest.mark.gpu
def test_fused_view_stats(dev):
    cl = scene.make_cloud(50000, 320, 240, 250.0, 250.0, seed=15, scale_k=0.15)
    parity.check_fused_view_stats(None, dev, cl, np.array([0.1, 0.2, 0.3], np.float32))


@pytest.mark.gpu
def test_full_size_view_factored_exchange(dev):
    # the same property at BASELINE.json's size: 2 M Gaussians @1080p, a batch of two keyframes
    cl = scene.make_config("C3", seed=0, n_views=2)
    rel = parity.check_view_factored(None, dev, cl, np.zeros(3, np.float32))
    assert rel < 1e-6, rel


def test_precomputed_colors_and_cov3D(oracle, dev):
    cl = scene.make_cloud(30000, 256, 192, 200.0, 200.0, seed=9, scale_k=0.15)
    cam = cl.cameras[0]
    bg = np.zeros(3, np.float32)
    rng = np.random.default_rng(0)
    colors = rng.random((cl.xyz.shape[0], 3)).astype(np.float32)
    o0, _, _, _ = parity.run_oracle(oracle, cl, cam, bg, do_backward=False)
    cov3D = o0.cov3D.copy()
    cov3D[o0.radii <= 0] = np.array([1, 0, 0, 1, 0, 1], np.float32) * 1e-3
    _check(oracle, dev, cl, cam, bg, use_colors_precomp=True, use_cov3D_precomp=True, colors=colors, cov3D=cov3D)


def test_edge_cases(oracle, dev):
    # all culled; P == 1; huge splat + opaque wall (early termination)
    cl = scene.make_cloud(5000, 128, 96, 100.0, 100.0, seed=11, scale_k=0.3)
    cam = cl.cameras[0]
    bg = np.array([0.3, 0.6, 0.9], np.float32)
    fwd = cam.viewmatrix[:3, 2]
    behind = scene.make_cloud(5000, 128, 96, 100.0, 100.0, seed=11, scale_k=0.3)
    behind.xyz[:] = cam.campos - 5.0 * fwd
    r, ores = _check(oracle, dev, behind, cam, bg)
    assert ores.R == 0 and np.allclose(r.out_color, bg[:, None, None])
    one = scene.make_cloud(1, 128, 96, 100.0, 100.0, seed=12, scale_k=0.3)
    one.xyz[0] = cam.campos + 2.0 * fwd
    _check(oracle, dev, one, cam, bg)
    center = cam.campos + 2.0 * fwd
    cl.xyz[0] = center
    cl.scaling[0] = np.log(5.0)
    cl.xyz[1:400] = center + 0.02 * np.random.default_rng(1).standard_normal((399, 3)).astype(np.float32) - 0.5 * fwd
    cl.scaling[1:400] = np.log(1.0)
    cl.opacity[0:400] = 8.0
    r, ores = _check(oracle, dev, cl, cam, np.zeros(3, np.float32))
    assert ores.tiles_touched[0] == ores.T


def test_forward_is_deterministic_and_backward_is_stable(dev):
    cl = scene.make_cloud(100000, 640, 480, 535.4, 535.4, seed=4, scale_k=0.2)
    cam = cl.cameras[0]
    bg = np.zeros(3, np.float32)
    a = parity.run_backend(None, dev, cl, cam, bg)
    b = parity.run_backend(None, dev, cl, cam, bg)
    for name in ("out_color", "radii", "point_list", "ranges", "n_contrib", "final_T"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    for name, g in a.grads.items():
        assert parity.rel_l1(g, b.grads[name]) <= 1e-5, name   # only the inter-tile atomic order varies


def test_full_size_C3_properties(dev):
    """BASELINE headline size (2M Gaussians, 1920x1080): size-independent properties next to the full oracle comparison
    of test_config_C3_full_size: sortedness of the instance list, ranges partition it, the instance count
    equals sum(tiles_touched), rectangles are consistent, blend weights are a sub-convex
    combination, and background linearity C(bg) = C(0) + T*bg."""
    cl = scene.make_config("C3", seed=0)
    cam = cl.cameras[0]
    r0 = parity.run_backend(None, dev, cl, cam, np.zeros(3, np.float32))
    P = cl.xyz.shape[0]
    assert r0.R == int(r0.tiles_touched.astype(np.int64).sum())
    vis = r0.radii > 0
    rect = r0.rect.astype(np.int64)
    assert np.array_equal(((rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1]))[vis], r0.tiles_touched[vis])
    # (tile, depth bits, id) lexicographic order of the whole list
    depth = r0.depth_key[r0.point_list].astype(np.uint64)
    key = (r0.tile_keys.astype(np.uint64) << np.uint64(32)) | depth
    assert np.all(key[1:] >= key[:-1])
    ties = key[1:] == key[:-1]
    assert np.all(r0.point_list[1:][ties] > r0.point_list[:-1][ties]), "equal keys must keep ascending Gaussian id"
    # ranges partition [0, R) in tile order
    counts = np.bincount(r0.tile_keys, minlength=r0.ranges.shape[0])
    nz = counts > 0
    assert np.array_equal((r0.ranges[:, 1] - r0.ranges[:, 0])[nz], counts[nz])
    assert np.array_equal(r0.ranges[nz][:, 1], np.cumsum(counts)[nz])
    # every instance's tile lies inside its Gaussian's rectangle
    gx = (cam.W + 15) // 16
    ty, tx = r0.tile_keys // gx, r0.tile_keys % gx
    rr = rect[r0.point_list]
    assert np.all((tx >= rr[:, 0]) & (tx < rr[:, 2]) & (ty >= rr[:, 1]) & (ty < rr[:, 3]))
    # blending: 0 <= T <= 1, n_contrib <= list length, colour bounded by max colour
    assert r0.final_T.min() >= 0 and r0.final_T.max() <= 1
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    r1 = parity.run_backend(None, dev, cl, cam, bg, do_backward=False)
    lin = r0.out_color + r1.final_T[None] * bg[:, None, None]
    assert np.abs(r1.out_color - lin).max() <= 1e-6
    assert np.array_equal(r0.n_contrib, r1.n_contrib)
    # gradients exist, are finite and vanish on culled Gaussians
    for name, g in r0.grads.items():
        assert np.isfinite(g).all(), name
        assert not np.any(g.reshape(P, -1)[~vis]), name


def test_training_recovers_a_perturbed_scene(dev):
    """End-to-end on the GPU: render a target from a cloud, perturb the cloud, optimise with the measured train
    step (HIP rasterizer fwd/bwd + fused loss + fused Adam): the loss must fall and PSNR rise."""
    from photo_slam_amd import loss_utils
    from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
    from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams, GaussianRenderer
    from photo_slam_amd.trainer import TrainStep
    cl = scene.make_cloud(20000, 320, 240, 300.0, 300.0, seed=3, scale_k=0.12)
    kf = GaussianKeyframe.from_camera(cl.cameras[0], dev)
    bg = torch.zeros(3, device=dev)
    target_model = GaussianModel.from_cloud(cl, device=dev)
    with torch.no_grad():
        gt = GaussianRenderer.render(kf, 240, 320, target_model, GaussianPipelineParams(), bg)[0].clone()
    rng = np.random.default_rng(0)
    cl.features_dc += 0.3 * rng.standard_normal(cl.features_dc.shape).astype(np.float32)
    cl.opacity += 0.5 * rng.standard_normal(cl.opacity.shape).astype(np.float32)
    cl.scaling += 0.1 * rng.standard_normal(cl.scaling.shape).astype(np.float32)
    g = GaussianModel.from_cloud(cl, device=dev)
    opt = GaussianOptimizationParams()
    g.trainingSetup(opt)
    ts = TrainStep(g, opt, GaussianPipelineParams(), bg)
    mask = torch.ones(3, 240, 320, device=dev)
    with torch.no_grad():
        psnr0 = float(loss_utils.psnr(GaussianRenderer.render(kf, 240, 320, g, GaussianPipelineParams(), bg)[0], gt))
    losses = [float(ts.trainForOneIteration(kf, gt, mask).detach()) for _ in range(150)]
    with torch.no_grad():
        psnr1 = float(loss_utils.psnr(GaussianRenderer.render(kf, 240, 320, g, GaussianPipelineParams(), bg)[0], gt))
    print(dict(loss0=losses[0], loss_end=losses[-1], psnr0=psnr0, psnr1=psnr1))
    assert losses[-1] < 0.6 * losses[0] and psnr1 > psnr0 + 2.0


def test_knn_large(oracle, dev):
    """simple-knn at 1 M points: result equals the oracle's; prints the device time."""
    import time
    import stages
    rng = np.random.default_rng(7)
    pts = (rng.random((1_000_000, 3)) * [6, 3, 6] - [3, 1.5, 3]).astype(np.float32)
    t = torch.from_numpy(pts).to(dev)
    from photo_slam_amd import rasterize_points as rp
    rp.distCUDA2(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = rp.distCUDA2(t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    want = oracle.knn(pts)
    print(f"distCUDA2(1M points): {dt * 1e3:.2f} ms")
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("cfg", ["C2", "C3", "C5"])
def test_cull_empty_tiles_at_full_size(dev, cfg):
    """GSR_CULL_EMPTY_TILES at the BASELINE shapes: the same image bit for bit and the same gradients (to the order of the
    four quad-waves' LDS adds) from a shorter instance list."""
    cl = scene.make_config(cfg, seed=0)
    kept, listed = parity.check_cull_empty_tiles(None, dev, cl, cl.cameras[0], np.array([0.1, 0.2, 0.3], np.float32), exact=False)
    print(f"{cfg}: instances listed {listed} -> {kept} ({kept / listed:.1%})")
    assert kept < 0.8 * listed   # measured: C2 62 %, C5 74 % of the rectangles' instances survive the tile-level bound


@pytest.mark.parametrize("switches", [{"GSR_XCD_CHUNK": "0", "GSR_EMIT_HIST": "0", "GSR_BINNING": "0"},
                                      {"GSR_XCD_CHUNK": "1", "GSR_BINNING": "1"},
                                      {"GSR_EMIT_SEEDS": "0", "GSR_DEPTH_SORT_9BIT": "0"}])
def test_library_switches_on_gpu(dev, switches, tmp_path):
    """The A/B handles on the hardware: one band of the image per XCD / single tiles; depth-first or tile-first binning; the
    emission without seeds, the plain four-pass depth sort.  Every stage against the oracle at C2, in a child process (the
    switches are read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import conftest, parity
from photo_slam_amd import scene
from oracle import oracle
oracle.build()
cl = scene.make_config("C2", seed=0)
cam = cl.cameras[0]
bg = np.zeros(3, np.float32)
dpix = np.random.default_rng(0).standard_normal((3, cam.H, cam.W)).astype(np.float32)
ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, dL_dpix=dpix)
r = parity.run_backend(None, torch.device("cuda:0"), cl, cam, bg, dL_dpix=dpix)
print(parity.compare(r, ores, ocolor, oradii, ograds, cam))
assert ores.R > 1_000_000
"""
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, PYTEST_CURRENT_TEST="switches", **switches), timeout=900)


def test_forward_from_two_host_threads_and_devices(oracle, dev):
    """The library's per-call state (the mapped count words, the event, the second stream) belongs to the calling THREAD and the
    CURRENT DEVICE (csrc/gsr_api.hip: host_sync): two host threads render different views at the same time, each on a stream of its
    own, ten times over -- every result equals the single-threaded one bit for bit (forward) / to the order of the LDS adds
    (backward); where the box has a second device, one thread then renders on both in turn."""
    import threading
    cl = scene.make_cloud(60000, 400, 300, 350.0, 350.0, seed=21, scale_k=0.2, n_views=2)
    bg = np.array([0.1, 0.3, 0.2], np.float32)
    dpix = np.random.default_rng(4).standard_normal((3, 300, 400)).astype(np.float32)
    want = [parity.run_backend(None, dev, cl, cl.cameras[v], bg, dL_dpix=dpix) for v in range(2)]
    errors, got = [], {0: [], 1: []}

    def work(v):
        try:
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                for _ in range(10):
                    got[v].append(parity.run_backend(None, dev, cl, cl.cameras[v], bg, dL_dpix=dpix))
                torch.cuda.current_stream(dev).synchronize()
        except Exception as e:   # (an assertion in a thread would otherwise vanish)
            errors.append((v, repr(e)))

    threads = [threading.Thread(target=work, args=(v,)) for v in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for v in range(2):
        assert len(got[v]) == 10
        for r in got[v]:
            assert r.R == want[v].R and np.array_equal(r.point_list, want[v].point_list) and np.array_equal(r.ranges, want[v].ranges)
            assert np.array_equal(r.out_color, want[v].out_color) and np.array_equal(r.n_contrib, want[v].n_contrib)
            for k, g in r.grads.items():
                assert parity.rel_l1(g, want[v].grads[k]) <= 1e-5, (v, k)
    if torch.cuda.device_count() > 1:
        other = torch.device("cuda", 1)
        for d in (other, dev, other):
            with torch.cuda.device(d):   # (the caller keeps the device of its stream current, as the reference's single-device code does)
                r = parity.run_backend(None, d, cl, cl.cameras[0], bg, dL_dpix=dpix)
            assert r.R == want[0].R and np.array_equal(r.out_color, want[0].out_color)
    else:
        print("one device on this box: the two-device half of the test did not run")
