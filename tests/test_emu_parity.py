"""Kernel-logic parity on the host: the product kernel sources, compiled against the wave64
emulator (tests/emu), must reproduce the oracle stage by stage on small scenes.  (The same
comparison runs on the real GPU build in test_gpu_parity.py, marked gpu.)"""
import numpy as np
import pytest
import torch

import parity
from photo_slam_amd import scene

CPU = torch.device("cpu")


def small_scene(P, W, H, seed, scale_k=0.35):
    return scene.make_cloud(P, W, H, 0.8 * W, 0.8 * W, seed=seed, scale_k=scale_k)


@pytest.mark.parametrize("flags", [32, 64])   # GSR_BINNING_DEPTH_FIRST / GSR_BINNING_TILE_FIRST (models this small default to tile-first)
@pytest.mark.parametrize("P,W,H,seed", [(600, 64, 48, 1), (1500, 80, 70, 2)])
def test_forward_backward_matches_oracle(emu_lib_path, oracle, P, W, H, seed, flags):
    cl = small_scene(P, W, H, seed)
    cam = cl.cameras[0]
    bg = np.array([0.2, 0.5, 0.1], np.float32)
    rng = np.random.default_rng(seed)
    dpix = rng.standard_normal((3, H, W)).astype(np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, dL_dpix=dpix)
    assert ores.R > 0
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, dL_dpix=dpix, flags=flags)
    rep = parity.compare(r, ores, ocolor, oradii, ograds, cam)
    print(rep)


@pytest.mark.parametrize("P,W,H,seed,scale_k,flags", [
    (600, 64, 48, 1, 0.35, 64), (1500, 80, 70, 2, 0.35, 64),
    (300, 96, 64, 5, 1.5, 64),       # large splats: long runs of instance slots
    (40, 16, 16, 7, 0.35, 64),       # a one-tile image: a tile sort of zero passes
    (6000, 64, 64, 3, 0.5, 64),      # lists of ~1 000 entries: several passes in LDS
    (30000, 32, 32, 4, 0.3, 64),     # lists of more than 2 048 entries: the chunked passes through the tile's own segments
    (150000, 96, 64, 6, 0.5, 64),    # more than 128 k Gaussians: the counts in two levels
    (1500, 80, 70, 2, 0.35, 64 | 8),  # with GSR_CULL_EMPTY_TILES: compared with the culled depth-first run below
])
def test_tile_first_binning_matches_oracle(emu_lib_path, oracle, P, W, H, seed, scale_k, flags):
    """GSR_BINNING_TILE_FIRST (include/gsr.h): no depth sort of the Gaussians -- the visible ones compacted in id order, every tile's
    list sorted by depth on its own behind the tile sort (tile_depth_sort.hip).  Every stage equals the oracle exactly as with the
    depth-first arrangement: the final (tile, depth bits, id) order is the reference's."""
    cl = small_scene(P, W, H, seed, scale_k=scale_k)
    cam = cl.cameras[0]
    bg = np.array([0.2, 0.5, 0.1], np.float32)
    dpix = np.random.default_rng(seed).standard_normal((3, H, W)).astype(np.float32)
    if flags & 8:
        a = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, dL_dpix=dpix, flags=32 | 8)
        b = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, dL_dpix=dpix, flags=flags)
        kept = int(a.ranges[:, 1].max())   # (behind the instances the tile sort kept the list's content is undefined)
        assert 0 < kept < a.R and np.array_equal(a.point_list[:kept], b.point_list[:kept])
        for k in ("ranges", "out_color", "n_contrib"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
        for k, g in a.grads.items():
            assert np.array_equal(g, b.grads[k]), k
        return
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, dL_dpix=dpix)
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, dL_dpix=dpix, flags=flags)
    parity.compare(r, ores, ocolor, oradii, ograds, cam)
    lens = ores.ranges[:, 1] - ores.ranges[:, 0]
    print("R", r.R, "longest list", int(lens.max()))
    if P == 30000:
        assert lens.max() > 4096     # (at least three chunks per pass)


@pytest.mark.parametrize("P,W,H,seed,scale_k", [(600, 64, 48, 1, 0.35), (1500, 80, 70, 2, 0.35), (300, 96, 64, 5, 1.5), (40, 16, 16, 7, 0.35)])
def test_cull_empty_tiles_keeps_image_and_gradients(emu_lib_path, P, W, H, seed, scale_k):
    """GSR_CULL_EMPTY_TILES: shorter instance lists, the same image and the same gradients bit for bit (the last case is a
    one-tile image: a tile sort of zero passes, where the flag must be ignored)."""
    cl = small_scene(P, W, H, seed, scale_k=scale_k)
    kept, listed = parity.check_cull_empty_tiles(emu_lib_path, CPU, cl, cl.cameras[0], np.array([0.2, 0.5, 0.1], np.float32), seed=seed)
    print(f"instances listed {listed} -> {kept}")
    assert kept < listed or W * H <= 256


def test_emission_without_seeds_gives_the_same_instance_list(emu_lib_path, tmp_path):
    """GSR_EMIT_SEEDS=0: emit_instances searches the offsets for every 256-slot window (the path windows beyond the seed table's
    capacity take) instead of starting from the seeds the offset scan leaves -- the same instance list, ranges and image.  The
    switch is read once per process: the run happens in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import conftest, parity
from photo_slam_amd import scene
cl = scene.make_cloud(1500, 80, 70, 64.0, 64.0, seed=2, scale_k=0.35)
r = parity.run_backend({emu_lib_path!r}, torch.device('cpu'), cl, cl.cameras[0], np.array([0.2, 0.5, 0.1], np.float32), do_backward=False)
np.savez(sys.argv[1], point_list=r.point_list, tile_keys=r.tile_keys, ranges=r.ranges, color=r.out_color, R=r.R)
"""
    outs = []
    for seeds, binning in (("1", "0"), ("0", "0"), ("0", "1")):
        out = str(tmp_path / f"emit_{seeds}_{binning}.npz")
        env = dict(os.environ, GSR_EMIT_SEEDS=seeds, GSR_BINNING=binning, PYTEST_CURRENT_TEST="emit")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
        outs.append(np.load(out))
    assert int(outs[0]["R"]) > 256 * 4      # (several windows)
    for o in outs[1:]:
        for k in ("point_list", "tile_keys", "ranges", "color"):
            assert np.array_equal(outs[0][k], o[k]), k


def test_library_switches_change_no_result(emu_lib_path, tmp_path):
    """The A/B handles of round 5 select HOW the work is dealt, counted and summed -- never what comes out:
    the blend kernels' deal of tiles to the XCDs (one band per XCD / row-major chunks / squares with the backward blend taking the
    heaviest squares first), the tile sort's first histogram counted by the emission or by its own launch, the three forms of the
    wave that sums a long run, the touched slots per trip.  Image and lists are those of the default setting bit for bit, the
    gradients to the order of a tile's four LDS adds.  Each switch is read once per process: children."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import conftest, parity
from photo_slam_amd import scene
cl = scene.make_cloud(900, 208, 176, 170.0, 170.0, seed=3, scale_k=0.5)      # 13 x 11 tiles: squares of 2 and 4 with ragged edges
cam = cl.cameras[0]
dpix = np.random.default_rng(1).standard_normal((3, cam.H, cam.W)).astype(np.float32)
r = parity.run_backend({emu_lib_path!r}, torch.device('cpu'), cl, cam, np.array([0.2, 0.5, 0.1], np.float32), dL_dpix=dpix)
np.savez(sys.argv[1], point_list=r.point_list, ranges=r.ranges, color=r.out_color, n_contrib=r.n_contrib, R=r.R,
         **{{'g_' + k: v for k, v in r.grads.items()}})
"""
    # (independent switches share a child: the suite stays short)
    settings = [{}, {"GSR_XCD_CHUNK": "0"}, {"GSR_XCD_CHUNK": "5", "GSR_EMIT_HIST": "0"},
                {"GSR_BINNING": "0", "GSR_XCD_CHUNK": "1"}, {"GSR_BINNING": "1"}, {"GSR_BINNING": "0"}]
    procs = []
    for i, extra in enumerate(settings):   # (the children run side by side)
        out = str(tmp_path / f"switch_{i}.npz")
        procs.append((subprocess.Popen([sys.executable, "-c", code, out], env=dict(os.environ, PYTEST_CURRENT_TEST="switches", OMP_NUM_THREADS="2", **extra)), out))
    outs = []
    for proc, out in procs:
        assert proc.wait(timeout=600) == 0
        outs.append(np.load(out))
    assert int(outs[0]["R"]) > 2048 and any(k.startswith("g_") for k in outs[0].files)
    for i, o in enumerate(outs[1:], 1):
        for k in outs[0].files:
            if k.startswith("g_"):
                # (another deal of the tiles: the four quad-waves of a tile meet the emulator's scheduler in another order, and so do
                # their LDS adds -- the last bits, as on the hardware)
                d = np.abs(outs[0][k].astype(np.float64) - o[k]).sum() / max(np.abs(outs[0][k]).sum(), 1e-30)
                assert d < 2e-6, (settings[i], k, d)
            else:
                assert np.array_equal(outs[0][k], o[k], equal_nan=True), (settings[i], k)


def test_depth_sort_second_path(emu_lib_path, oracle, tmp_path):
    """The depth sort runs three 9-bit passes over key - bits(0.2f) (27 bits: z < 13 107) and sorts AGAIN with the plain four passes
    when the host finds a larger key in the view (include/gsr.h: gsr_depth_resort_count).  (a) A Gaussian at z = 20 000 and one
    whose depth is Inf-like large next to an ordinary scene: the second path runs, every stage equals the oracle.  (b) The range
    narrowed to 9 bits (GSR_DEPTH_SORT_BITS, read once per process: a child): an ordinary scene takes the second path and gives
    the lists and the image of the default setting bit for bit; so does the plain sort (GSR_DEPTH_SORT_9BIT=0)."""
    import ctypes as C
    import os
    import subprocess
    import sys
    from photo_slam_amd import capi
    lib = capi.load(emu_lib_path)
    lib.gsr_depth_resort_count.restype = C.c_longlong
    cl = small_scene(800, 96, 64, 4)
    cam = cl.cameras[0]
    fwd = cam.viewmatrix[:3, 2]
    cl.xyz[0] = cam.campos + 20000.0 * fwd          # far beyond the three-pass range, still in front of the camera
    cl.scaling[0] = np.log(3000.0)                   # (large enough to cover pixels from there)
    cl.opacity[0] = 2.0
    bg = np.array([0.2, 0.5, 0.1], np.float32)
    dpix = np.random.default_rng(0).standard_normal((3, cam.H, cam.W)).astype(np.float32)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, cl, cam, bg, dL_dpix=dpix)
    assert oradii[0] > 0, "the far Gaussian must be visible for this test to mean anything"
    before = lib.gsr_depth_resort_count()
    r = parity.run_backend(emu_lib_path, CPU, cl, cam, bg, dL_dpix=dpix, flags=32)   # (depth-first: a model this small defaults to tile-first)
    assert lib.gsr_depth_resort_count() == before + 1
    parity.compare(r, ores, ocolor, oradii, ograds, cam)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import conftest, parity
from photo_slam_amd import scene, capi
cl = scene.make_cloud(1500, 80, 70, 64.0, 64.0, seed=2, scale_k=0.35)
r = parity.run_backend({emu_lib_path!r}, torch.device('cpu'), cl, cl.cameras[0], np.array([0.2, 0.5, 0.1], np.float32), do_backward=False)
lib = capi.load({emu_lib_path!r}); lib.gsr_depth_resort_count.restype = C.c_longlong
np.savez(sys.argv[1], point_list=r.point_list, tile_keys=r.tile_keys, ranges=r.ranges, color=r.out_color, R=r.R, resorts=lib.gsr_depth_resort_count())
"""
    outs = []
    for extra in ({}, {"GSR_DEPTH_SORT_BITS": "9"}, {"GSR_DEPTH_SORT_9BIT": "0"}):
        out = str(tmp_path / f"depth_{len(outs)}.npz")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, PYTEST_CURRENT_TEST="depth", GSR_BINNING="0", **extra), timeout=600)
        outs.append(np.load(out))
    assert int(outs[0]["resorts"]) == 0 and int(outs[1]["resorts"]) == 1 and int(outs[2]["resorts"]) == 0
    for o in outs[1:]:
        for k in ("point_list", "tile_keys", "ranges", "color"):
            assert np.array_equal(outs[0][k], o[k]), k
