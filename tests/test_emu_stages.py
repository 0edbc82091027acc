"""Building blocks and edge cases of the C-ABI, run on the host through the wave64 emulator
build of the kernel sources (GPU twins: test_gpu_stages.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity
import stages
from photo_slam_amd import capi, scene
from photo_slam_amd import rasterize_points as rp

CPU = torch.device("cpu")


@pytest.mark.parametrize("n", [1, 63, 64, 255, 256, 257, 2047, 2048, 2049, 10000])
@pytest.mark.parametrize("inclusive", [0, 1])
def test_scan(emu_lib_path, n, inclusive):
    rng = np.random.default_rng(n)
    v = rng.integers(0, 50, n).astype(np.uint32)
    got = stages.scan_u32(emu_lib_path, CPU, v, inclusive)
    want = np.cumsum(v, dtype=np.uint64).astype(np.uint32)
    if not inclusive:
        want = want - v
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,begin,end", [(1, 0, 8), (100, 0, 8), (4096, 0, 16), (4097, 0, 13), (9000, 0, 32),
                                         (5000, 3, 9), (12345, 0, 5),
                                         # > 32 blocks of 2048: several block GROUPS of the two-level prefix (sort.hip), the last one
                                         # partial, one pass with fewer than 8 bits
                                         (150_001, 0, 13)])
def test_radix_sort_stable(emu_lib_path, n, begin, end):
    rng = np.random.default_rng(n + end)
    keys = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if n > 50:
        keys[rng.integers(0, n, n // 2)] = keys[0]  # many duplicates: stability matters
    vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    k, v = stages.radix_sort_pairs(emu_lib_path, CPU, keys, vals, begin, end)
    wk, wv = stages.reference_sort(keys, vals, begin, end)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)
    # values_in == NULL -> iota
    k, v = stages.radix_sort_pairs(emu_lib_path, CPU, keys, None, begin, end)
    wk, wv = stages.reference_sort(keys, None, begin, end)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)


def test_radix_sort_skewed_digits(emu_lib_path):
    # all keys share the high bytes (like depth exponents): every lane of a wave hits one bin
    n = 6000
    keys = (np.uint32(0x3F800000) + (np.arange(n) % 7).astype(np.uint32)).astype(np.uint32)
    k, v = stages.radix_sort_pairs(emu_lib_path, CPU, keys, None, 0, 32)
    wk, wv = stages.reference_sort(keys, None, 0, 32)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)


@pytest.mark.parametrize("P", [1, 2, 3, 4, 50, 1024, 1025, 3000])
def test_knn_matches_oracle_and_bruteforce(emu_lib_path, oracle, P):
    rng = np.random.default_rng(P)
    pts = (rng.standard_normal((P, 3)) * [3, 1, 2] + [0.5, -0.2, 1.0]).astype(np.float32)
    if P > 10:
        pts[5] = pts[6]  # duplicate point: distance 0
    got = stages.knn(emu_lib_path, CPU, pts)
    want = oracle.knn(pts)
    brute = oracle.knn(pts, bruteforce=True)
    assert np.array_equal(want, brute), "oracle's Morton/box algorithm disagrees with brute force"
    assert np.array_equal(got, want)


def test_knn_box_hierarchy_across_super_boxes(emu_lib_path, oracle):
    """40 000 points = 40 boxes of 1024 in 2 super boxes of the three-level hierarchy (csrc/knn.hip), with a dense clump whose
    distances are 1/20 of the rest and duplicates: bit-identical to the oracle's scan of whole 1024-boxes (simple_knn.cu:147-183)."""
    rng = np.random.default_rng(3)
    P = 40000
    pts = (rng.random((P, 3), dtype=np.float32) * np.array([6, 3, 6], np.float32)).astype(np.float32)
    pts[:5000] *= 0.05
    pts[7000:7010] = pts[6999]
    got = stages.knn(emu_lib_path, CPU, pts)
    assert np.array_equal(got, oracle.knn(pts))


def test_mark_visible(emu_lib_path, oracle):
    cl = scene.make_cloud(3000, 64, 48, 50.0, 50.0, seed=5)
    cam = cl.cameras[0]
    got = stages.mark_visible(emu_lib_path, CPU, cl.xyz, cam.viewmatrix, cam.projmatrix)
    want = oracle.mark_visible(cl.xyz, cam.viewmatrix, cam.projmatrix)
    assert np.array_equal(got, want) and 0 < want.sum() < 3000


def _scene(P=400, W=48, H=32, seed=7, **kw):
    return scene.make_cloud(P, W, H, 40.0, 40.0, seed=seed, scale_k=0.35, **kw)


def test_colors_precomp_and_cov3D_precomp_paths(emu_lib_path, oracle):
    cl = _scene()
    cam = cl.cameras[0]
    bg = np.zeros(3, np.float32)
    rng = np.random.default_rng(0)
    colors = rng.random((cl.xyz.shape[0], 3)).astype(np.float32)
    # a valid cov3D: take the oracle's own cov3D of the scale/rot path
    o0, _, _, _ = parity.run_oracle(oracle, cl, cam, bg, do_backward=False)
    cov3D = o0.cov3D.copy()
    # culled Gaussians have no cov3D in the oracle state; give them something finite
    cov3D[o0.radii <= 0] = np.array([1, 0, 0, 1, 0, 1], np.float32) * 1e-3
    kw = dict(use_colors_precomp=True, use_cov3D_precomp=True, colors=colors, cov3D=cov3D)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, **kw)
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, **kw)
    parity.compare(r, ores, ocolor, oradii, ograds, cam, use_colors_precomp=True, use_cov3D_precomp=True)
    assert r.grads["dL_dsh"].size == 0 and not r.grads["dL_dscales"].any()


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_lower_sh_degrees(emu_lib_path, oracle, degree):
    cl = _scene(seed=8)
    cam = cl.cameras[0]
    bg = np.array([1, 1, 1], np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, sh_degree=degree)
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, sh_degree=degree)
    parity.compare(r, ores, ocolor, oradii, ograds, cam)
    k = (degree + 1) ** 2
    assert not r.grads["dL_dsh"][:, k:, :].any(), "coefficients above the active degree must get zero gradient"


@pytest.mark.parametrize("degree,coeffs", [(0, 1), (1, 4), (2, 9), (1, 9)])
def test_compact_sh_layouts(emu_lib_path, oracle, degree, coeffs):
    # SH tensors that are not [P,16,3] (rows of 3, 12 or 27 floats): the kernels' per-lane row access instead of the LDS movers
    cl = _scene(seed=9)
    cam = cl.cameras[0]
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, sh_degree=degree, sh_coeffs=coeffs)
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, sh_degree=degree, sh_coeffs=coeffs)
    parity.compare(r, ores, ocolor, oradii, ograds, cam)
    assert r.grads["dL_dsh"].shape[1] == coeffs
    k = (degree + 1) ** 2
    assert not r.grads["dL_dsh"][:, k:, :].any()


@pytest.mark.parametrize("degree,coeffs,n_views", [(3, None, 3), (1, None, 2), (1, 4, 2)])
def test_view_factored_sh_gradient(emu_lib_path, degree, coeffs, n_views):
    cl = _scene(P=300, seed=21, n_views=n_views)
    parity.check_view_factored(emu_lib_path, CPU, cl, np.array([0.1, 0.2, 0.3], np.float32), sh_degree=degree,
                               sh_coeffs=coeffs)


@pytest.mark.parametrize("degree", [3, 1])
def test_fused_sh_adam(emu_lib_path, degree):
    parity.check_fused_sh_adam(emu_lib_path, CPU, _scene(P=330, seed=23), np.array([0.1, 0.2, 0.3], np.float32),
                               sh_degree=degree)


def test_backward_may_follow_one_forward_more_than_once(emu_lib_path):
    parity.check_backward_twice(emu_lib_path, CPU, _scene(P=400, seed=26), np.array([0.1, 0.2, 0.3], np.float32))


def test_fused_geom_adam(emu_lib_path):
    parity.check_fused_geom_adam(emu_lib_path, CPU, _scene(P=330, seed=25), np.array([0.1, 0.2, 0.3], np.float32))


def test_fused_view_stats(emu_lib_path):
    parity.check_fused_view_stats(emu_lib_path, CPU, _scene(P=330, seed=24), np.array([0.1, 0.2, 0.3], np.float32))


def test_empty_and_tiny_inputs(emu_lib_path, oracle):
    rp._LIB_OVERRIDE = emu_lib_path
    try:
        e = torch.empty(0)
        cam = _scene().cameras[0]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        # P == 0: valid no-op, all-zero image (not bg) -- src/rasterize_points.cu:68,81
        R, color, radii, g, b, i = rp.RasterizeGaussiansCUDA(
            torch.ones(3), torch.zeros((0, 3)), e, torch.zeros((0, 1)), torch.zeros((0, 3)), torch.zeros((0, 4)), 1.0, e,
            t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, cam.H, cam.W, torch.zeros((0, 16, 3)), 3,
            t(cam.campos), False)
        assert R == 0 and color.shape == (3, cam.H, cam.W) and not color.any() and radii.numel() == 0
        # bad means3D shape -> the reference's AT_ERROR message
        with pytest.raises(RuntimeError, match="means3D must have dimensions"):
            rp.RasterizeGaussiansCUDA(torch.ones(3), torch.zeros((4, 2)), e, e, e, e, 1.0, e, t(cam.viewmatrix),
                                      t(cam.projmatrix), 1.0, 1.0, 8, 8, e, 0, t(cam.campos), False)
    finally:
        rp._LIB_OVERRIDE = None
    # P == 1 and all-culled (camera looking away)
    for P, flip in ((1, False), (200, True)):
        cl = _scene(P=P, seed=11)
        cam = cl.cameras[0]
        if flip:
            cl.xyz[:] = cl.xyz * 0 + cam.campos - 5.0 * cam.viewmatrix[:3, 2]  # behind the camera
        bg = np.array([0.3, 0.6, 0.9], np.float32)
        ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg)
        r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg)
        parity.compare(r, ores, ocolor, oradii, ograds, cam)
        if flip:
            assert ores.R == 0 and np.allclose(r.out_color, bg[:, None, None])


def test_single_huge_splat_and_opaque_wall(emu_lib_path, oracle):
    # one Gaussian covering every tile + an opaque stack that triggers early termination (T < 1e-4)
    cl = _scene(P=300, W=64, H=48, seed=12)
    cam = cl.cameras[0]
    fwd = cam.viewmatrix[:3, 2]
    center = cam.campos + 2.0 * fwd
    cl.xyz[0] = center
    cl.scaling[0] = np.log(5.0)
    cl.xyz[1:40] = center + 0.02 * np.random.default_rng(1).standard_normal((39, 3)).astype(np.float32) - 0.5 * fwd
    cl.scaling[1:40] = np.log(1.0)
    cl.opacity[0:40] = 8.0
    bg = np.zeros(3, np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg)
    assert ores.tiles_touched[0] == ores.T and float(np.median(ores.final_T)) < 5e-4  # saturated: early termination
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg)
    parity.compare(r, ores, ocolor, oradii, ograds, cam)


def test_run_of_more_than_4096_instance_slots(emu_lib_path, oracle):
    """A splat that covers every tile of a 72 x 64-tile image: a run of 4 608 instance slots, i.e. more than 64 slots per lane of
    the wave that sums it (partials.h: wave_sum_long_run takes a lane's consecutive slots in groups of 64)."""
    W, H = 72 * 16, 64 * 16
    cl = scene.make_cloud(8, W, H, 0.9 * W, 0.9 * W, seed=5, scale_k=0.3)      # (few Gaussians: the emulator walks 1.2 M pixels)
    cam = cl.cameras[0]
    fwd = cam.viewmatrix[:3, 2]
    cl.xyz[0] = cam.campos + 2.0 * fwd
    cl.scaling[0] = np.log(6.0)
    cl.opacity[0] = -1.0          # translucent: every pixel of every tile blends it, everything behind it too
    cl.xyz[1:4] = cam.campos + 2.5 * fwd + 0.3 * np.random.default_rng(2).standard_normal((3, 3)).astype(np.float32)
    cl.scaling[1:4] = np.log(3.0)
    cl.opacity[1:4] = -1.5
    bg = np.array([0.2, 0.1, 0.3], np.float32)
    dpix = np.random.default_rng(7).standard_normal((3, H, W)).astype(np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, dL_dpix=dpix)
    assert ores.tiles_touched[0] == 72 * 64 and (ores.tiles_touched[1:4] > 4096).any()
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, dL_dpix=dpix)
    parity.compare(r, ores, ocolor, oradii, ograds, cam)


def _long_run_scene():
    W, H = 176, 160   # 11 x 10 = 110 tiles
    cl = scene.make_cloud(200, W, H, 0.8 * W, 0.8 * W, seed=31, scale_k=0.35)
    cam = cl.cameras[0]
    fwd = cam.viewmatrix[:3, 2]
    rng = np.random.default_rng(3)
    big = np.arange(130, 200)
    cl.xyz[big] = cam.campos + (1.5 + rng.random((70, 1)).astype(np.float32)) * fwd + 0.3 * rng.standard_normal((70, 3)).astype(np.float32)
    cl.scaling[big] = np.log(0.25 + 0.6 * rng.random((70, 3))).astype(np.float32)
    cl.opacity[big] = -2.0 + rng.standard_normal((70, 1)).astype(np.float32)   # translucent: every layer contributes
    return cl, cam, big, rng


def test_backward_twice_with_long_runs(emu_lib_path):
    """A second and a third backward pass on the state of one forward pass give the gradients of a first one, long runs included
    (long_run_sums_kernel overwrites a run's first slot with the run's total: the backward blend rewrites every touched slot)."""
    cl, cam, big, rng = _long_run_scene()
    parity.check_backward_twice(emu_lib_path, CPU, cl, np.array([0.1, 0.3, 0.2], np.float32))


def test_long_runs_of_instance_slots(emu_lib_path, oracle):
    """Gaussians that touch more than 64 tiles (state.h LONG_RUN): their per-instance gradient slots are summed by
    long_run_sums_kernel, one wave per run, from the list the offset scan leaves -- here 70 of them CONSECUTIVE in index order
    at the end of the arrays (what densification produces: the children of split Gaussians), more than one wave of the
    backward preprocess holds, next to small ones; runs of exactly 64 and 65 tiles sit on the threshold."""
    cl, cam, big, rng = _long_run_scene()
    W, H = cam.W, cam.H
    bg = np.array([0.1, 0.3, 0.2], np.float32)
    dpix = rng.standard_normal((3, H, W)).astype(np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, dL_dpix=dpix)
    tt = ores.tiles_touched
    assert (tt[big] > 64).sum() >= 40 and (tt > 64).sum() < 200 and ((tt > 0) & (tt <= 64)).sum() > 20, np.sort(tt)[-80:]
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, dL_dpix=dpix)
    parity.compare(r, ores, ocolor, oradii, ograds, cam)


def test_invalid_argument_combinations(emu_lib_path):
    lib = capi.load(emu_lib_path)
    a = capi.ForwardArgs()
    n = C.c_int(0)
    cb = capi.ALLOC_FN(lambda ctx, nbytes: 0)
    assert lib.gsr_forward(None, cb, None, cb, None, cb, None, None, C.byref(n)) == -1
    a.P, a.width, a.height = 5, 16, 16
    # neither SHs nor colours (gaussian_rasterizer.cpp:201-203)
    assert lib.gsr_forward(C.byref(a), cb, None, cb, None, cb, None, None, C.byref(n)) == -1
    assert lib.gsr_strerror(-1) == b"invalid argument"
    assert lib.gsr_backend() == b"emu-wave64"
    # allocation failure is reported, not dereferenced
    buf = np.zeros(64, np.float32)
    p = buf.ctypes.data
    a.shs = None
    a.colors_precomp = p; a.cov3D_precomp = p; a.means3D = p; a.opacities = p; a.background = p
    a.viewmatrix = p; a.projmatrix = p; a.cam_pos = p; a.out_color = p
    assert lib.gsr_forward(C.byref(a), cb, None, cb, None, cb, None, None, C.byref(n)) == -2


def test_backward_extension_errors(emu_lib_path):
    """Error behaviour of the gsr_backward extensions: the fused SH Adam step needs [P,16,3] rows, excludes the factored
    mode, and the three statistics pointers come together."""
    cl = _scene(P=120, seed=31)
    cam = cl.cameras[0]
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    hyper = dict(lr=0.0025, lr_tail=0.0025 / 20, beta1=0.9, beta2=0.999, eps=1e-15, step=1)
    # compact SH layout (4 coefficients): the fused step is refused by the library
    moments = dict(exp_avg=torch.zeros(120, 4, 3), exp_avg_sq=torch.zeros(120, 4, 3))
    with pytest.raises(capi.GsrError, match="unsupported"):
        parity.run_backend(emu_lib_path, CPU, cl, cam, bg, sh_degree=1, sh_coeffs=4, sh_adam=dict(**moments, **hyper))
    # moments of another shape, step 0, both SH extensions at once: refused by the wrapper / the library
    with pytest.raises(RuntimeError, match="shaped like sh"):
        parity.run_backend(emu_lib_path, CPU, cl, cam, bg, sh_adam=dict(**moments, **hyper))
    full = dict(exp_avg=torch.zeros(120, 16, 3), exp_avg_sq=torch.zeros(120, 16, 3))
    with pytest.raises(capi.GsrError, match="invalid argument"):
        parity.run_backend(emu_lib_path, CPU, cl, cam, bg, sh_adam=dict(**full, **dict(hyper, step=0)))
    with pytest.raises(RuntimeError, match="mutually exclusive"):
        parity.run_backend(emu_lib_path, CPU, cl, cam, bg, factored=True, sh_adam=dict(**full, **hyper))
    with pytest.raises(RuntimeError, match="num_points elements"):
        parity.run_backend(emu_lib_path, CPU, cl, cam, bg, view_stats=[torch.zeros(120), torch.zeros(120), torch.zeros(7)])
    # strided views are fine, strided inner dimensions are not
    views = torch.zeros(2, 121, 3)
    with pytest.raises(RuntimeError, match="only the view dimension may be strided"):
        rp._LIB_OVERRIDE = emu_lib_path
        try:
            rp.shGradFromViews(torch.zeros(120, 3), views[:, 120, :], torch.zeros(2, 120, 6)[:, :, ::2], 3, 16, 0.5)
        finally:
            rp._LIB_OVERRIDE = None


def tile_depth_sort_case(lib_path, dev, seed=0, big=10_000):
    """csrc/tile_depth_sort.hip at the sizes where its paths change: lists of 0, 1, 2, 63 ... 65, 2 047 ... 2 049 (registers + LDS up to
    2 048 entries, chunks of 2 048 beyond), 4 095 ... 4 097 and one long list; keys that are all equal, two-valued, differ only in the
    high or only in the low bits, already sorted, reversed, and random -- against a stable argsort of every list."""
    rng = np.random.default_rng(seed)
    lengths = [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, big, 0, 3]
    P = 50_000
    kinds = ("random", "equal", "two", "high", "low", "sorted", "reversed")
    for kind in kinds:
        key = {"random": lambda: rng.integers(0, 2**32, P, dtype=np.uint64),
               "equal": lambda: np.full(P, 0x3F800000, np.uint64),
               "two": lambda: np.where(rng.random(P) < 0.5, 0x40000000, 0x3F000000).astype(np.uint64),
               "high": lambda: rng.integers(0, 256, P, dtype=np.uint64) << np.uint64(24),
               "low": lambda: np.uint64(0x3F800000) + rng.integers(0, 300, P, dtype=np.uint64),
               "sorted": lambda: np.arange(P, dtype=np.uint64) * np.uint64(7919),
               "reversed": lambda: np.uint64(0xFFFFFFF0) - np.arange(P, dtype=np.uint64) * np.uint64(7919)}[kind]().astype(np.uint32)
        # ids in ascending order inside a list, as the stable tile sort delivers them (any ids: the kernel only gathers their keys)
        lists = [np.sort(rng.choice(P, n, replace=n > P)) for n in lengths]
        pl = np.concatenate(lists).astype(np.uint32)
        got = stages.tile_depth_sort(lib_path, dev, lengths, key, pl)
        at = 0
        for n, ids in zip(lengths, lists):
            want = ids[np.argsort(key[ids], kind="stable")]
            assert np.array_equal(got[at:at + n], want), (kind, n)
            at += n


def test_tile_depth_sort_at_its_path_boundaries(emu_lib_path):
    tile_depth_sort_case(emu_lib_path, torch.device("cpu"), big=5000)
