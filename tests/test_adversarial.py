"""Pathological inputs through the HIP kernels against the CPU oracle: zero quaternions, scales of 1e4 /
1e-12 / negative, NaN and Inf positions, 200 coincident points, opacity exactly 0 and 1, SH coefficients x 1e6 (both signs), a
Gaussian exactly on the camera centre.  The reference guards none of these (cuda_rasterizer/forward.cu:155-256 has no input
validation); what it DOES with them is defined by its arithmetic, which the oracle restates line by line
(oracle/gsr_oracle.c) -- the HIP kernels must take the same decisions (radii, tile counts, sort order, tile ranges exactly)
and produce the same image and gradients, non-finite values in the same places.

On the host: the kernel sources on the wave64 emulator.  On the GPU (-m gpu): the real instruction stream -- v_cvt_i32_f32,
v_exp_f32, v_rcp_f32 with NaN / Inf / denormal operands exist only there."""
import numpy as np
import pytest
import torch

import parity
from photo_slam_amd import scene

CASES = ("zero_quaternion", "scale_1e4", "scale_1e-12", "negative_scale", "nan_xyz", "inf_xyz", "coincident_200", "opacity_0_and_1",
         "sh_times_1e6", "sh_times_minus_1e6", "point_on_the_camera")


class _Activated:
    """activated parameters, as RasterizeGaussiansCUDA receives them (src/gaussian_renderer.cpp:66-121)"""

    def __init__(self, cl):
        self.xyz, self.cameras = cl.xyz.copy(), cl.cameras
        self.opacity, self.features = cl.get_opacity().copy(), cl.get_features().copy()
        self.scaling, self.rotation = cl.get_scaling().copy(), cl.get_rotation().copy()

    def get_opacity(self): return self.opacity
    def get_features(self): return self.features
    def get_scaling(self): return self.scaling
    def get_rotation(self): return self.rotation


def make_case(name, P, W, H, f):
    cl = scene.make_cloud(P, W, H, f, f, seed=41, scale_k=0.3)
    a = _Activated(cl)
    cam = cl.cameras[0]
    rng = np.random.default_rng(5)
    idx = rng.choice(P, max(P // 20, 8), replace=False)        # 5 % of the cloud, spread over depth and screen
    fwd = cam.viewmatrix[:3, 2].copy()                         # (transposed 4x4: the forward axis in world coordinates)
    if name == "zero_quaternion":
        a.rotation[idx] = 0.0
    elif name == "scale_1e4":
        a.scaling[idx] = 1e4
    elif name == "scale_1e-12":
        a.scaling[idx] = 1e-12
    elif name == "negative_scale":
        a.scaling[idx] *= -1.0
    elif name == "nan_xyz":
        a.xyz[idx[::2], 0] = np.nan
        a.xyz[idx[1::2]] = np.nan
    elif name == "inf_xyz":
        a.xyz[idx[::3], 2] = np.inf
        a.xyz[idx[1::3], 1] = -np.inf
        a.xyz[idx[2::3]] = np.inf
    elif name == "coincident_200":
        n = min(200, P // 2)
        a.xyz[:n] = cam.campos + 2.5 * fwd
        a.scaling[:n] = a.scaling[0]
        a.rotation[:n] = a.rotation[0]
    elif name == "opacity_0_and_1":
        a.opacity[idx[::2]] = 0.0
        a.opacity[idx[1::2]] = 1.0
    elif name == "sh_times_1e6":
        a.features[idx] *= 1e6
    elif name == "sh_times_minus_1e6":
        a.features[idx] *= -1e6
    elif name == "point_on_the_camera":
        a.xyz[idx[0]] = cam.campos
        a.xyz[idx[1]] = cam.campos + 0.2 * fwd                 # exactly on Photo-SLAM's near plane (auxiliary.h:154)
        a.xyz[idx[2]] = cam.campos + np.float32(0.2000001) * fwd
    return a, cam


def check_case(lib_path, dev, oracle, name, P, W, H, f):
    """both binning arrangements (include/gsr.h: GSR_BINNING_DEPTH_FIRST with its second depth-sort path for out-of-range and NaN
    depths, GSR_BINNING_TILE_FIRST with the per-tile sorts on the raw key bits); returns the worst gradient deviation"""
    return max(_check_case(lib_path, dev, oracle, name, P, W, H, f, flags) for flags in (32, 64))


def _check_case(lib_path, dev, oracle, name, P, W, H, f, flags):
    a, cam = make_case(name, P, W, H, f)
    bg = np.array([0.2, 0.1, 0.4], np.float32)
    dpix = np.random.default_rng(9).standard_normal((3, H, W)).astype(np.float32)
    r = parity.run_backend(lib_path, dev, a, cam, bg, sh_degree=3, dL_dpix=dpix, flags=flags)
    ores, ocolor, oradii, ograds = parity.run_oracle(oracle, a, cam, bg, sh_degree=3, dL_dpix=dpix)
    vis = oradii > 0
    # ---- decisions: exact
    assert np.array_equal(r.radii, oradii), (name, int((r.radii != oradii).sum()))
    assert np.array_equal(r.tiles_touched, ores.tiles_touched), name
    assert np.array_equal(r.depth_key, np.where(vis, ores.depths.view(np.uint32), np.uint32(0xFFFFFFFF))), name
    assert r.R == ores.R and np.array_equal(r.point_list, ores.point_list), name
    assert np.array_equal(r.ranges, ores.ranges), name
    # ---- image: non-finite pixels in the same places, the others to 1e-4 mean abs (a pixel that a 1e6-fold colour reaches
    # carries a value of that magnitude: relative to the image's own scale)
    fin_o, fin_r = np.isfinite(ocolor), np.isfinite(r.out_color)
    assert np.array_equal(fin_o, fin_r), (name, int((fin_o != fin_r).sum()))
    scale = max(1.0, float(np.abs(ocolor[fin_o]).max(initial=0.0)))
    assert np.abs(r.out_color[fin_o] - ocolor[fin_o]).mean() <= 1e-4 * scale, name
    solid = (ores.fragile == 0) & np.isfinite(ores.final_T)
    assert (r.n_contrib != ores.n_contrib)[solid].sum() == 0, name
    # ---- gradients: non-finite entries in the same places, the finite ones to 1e-4 relative L1; exact zeros where culled
    worst = 0.0
    for k, g in r.grads.items():
        if g.size == 0:
            continue
        o = ograds[k]
        fo, fg = np.isfinite(o), np.isfinite(g)
        assert np.array_equal(fo, fg), (name, k, int((fo != fg).sum()))
        rel = float(np.abs(g[fo].astype(np.float64) - o[fo].astype(np.float64)).sum() / (np.abs(o[fo].astype(np.float64)).sum() + 1e-30))
        worst = max(worst, rel)
        assert rel <= 1e-4, (name, k, rel)
        assert not np.any(g.reshape(g.shape[0], -1)[~vis]), (name, k)
    return worst


@pytest.mark.parametrize("name", CASES)
def test_pathological_inputs_on_the_emulated_kernels(emu_lib_path, oracle, name):
    check_case(emu_lib_path, torch.device("cpu"), oracle, name, 600, 64, 48, 50.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_pathological_inputs_on_the_gpu(oracle, name):
    worst = check_case(None, torch.device("cuda", 0), oracle, name, 20000, 320, 240, 300.0)
    print(f"{name}: worst gradient deviation {worst:.1e}")
