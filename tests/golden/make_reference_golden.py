"""Regenerates tests/golden/reference_small.npz: outputs of the REFERENCE's own rasterizer sources
(/root/reference/cuda_rasterizer/*.cu compiled for the host by oracle/build_ref.py) on small seeded scenes.

These are reference outputs -- modulo the host shims for CUDA blocks, the two CUB calls and the glm operators
(oracle/ref_shim/) and modulo nvcc's FMA contraction / the GPU's atomic order, which no host build reproduces.
The fixture lets the oracle AND the HIP kernels be checked against the reference's code on machines where
/root/reference does not exist (the GPU boxes).

    python tests/golden/make_reference_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from photo_slam_amd import scene  # noqa: E402

CASES = (
    # name, P, W, H, fx, seed, scale_k, sh_degree, precomputed colours + covariances
    ("a", 600, 64, 48, 50.0, 1, 0.35, 3, False),
    ("b", 1500, 80, 70, 70.0, 2, 0.30, 1, False),     # ragged tile grid (80x70), lower SH degree
    ("c", 400, 48, 32, 40.0, 3, 0.40, 3, True),
)
FIELDS = ("out_color", "radii", "depths", "clamped", "means2D", "cov3D", "conic_opacity", "rgb", "tiles_touched", "point_offsets",
          "keys_sorted", "point_list", "ranges", "n_contrib", "final_T")


def inputs(case):
    name, P, W, H, fx, seed, k, deg, precomp = case
    cl = scene.make_cloud(P, W, H, fx, fx, seed=seed, scale_k=k)
    cam = cl.cameras[0]
    rng = np.random.default_rng(100 + seed)
    d = dict(bg=np.array([0.1, 0.2, 0.3], np.float32), xyz=cl.xyz, opacity=cl.get_opacity(), features=cl.get_features(),
             scaling=cl.get_scaling(), rotation=cl.get_rotation(), viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
             campos=cam.campos, tanfov=np.array([cam.tanfovx, cam.tanfovy], np.float32), size=np.array([W, H, deg], np.int32),
             dpix=rng.standard_normal((3, H, W)).astype(np.float32))
    if precomp:
        d["colors"] = rng.random((P, 3)).astype(np.float32)
    return d


def run_reference(d, precomp_cov3D=None):
    from oracle import ref
    W, H, deg = (int(v) for v in d["size"])
    kw = dict(shs=d["features"], sh_degree=deg, scales=d["scaling"], rotations=d["rotation"])
    if "colors" in d:
        kw = dict(colors_precomp=d["colors"], sh_degree=deg, cov3D_precomp=precomp_cov3D) if precomp_cov3D is not None else \
            dict(colors_precomp=d["colors"], sh_degree=deg, scales=d["scaling"], rotations=d["rotation"])
    return ref.forward_backward(d["bg"], d["xyz"], d["opacity"], d["viewmatrix"], d["projmatrix"], d["campos"],
                                float(d["tanfov"][0]), float(d["tanfov"][1]), H, W, dL_dpix=d["dpix"], **kw)


def compute():
    out = {}
    for case in CASES:
        name = case[0]
        d = inputs(case)
        if case[-1]:
            # precomputed covariances = the reference's own cov3D of a first pass (harmless values where culled)
            first = run_reference(d)
            cov = first.cov3D.copy()
            cov[first.radii <= 0] = np.array([1, 0, 0, 1, 0, 1], np.float32) * 1e-3
            d["cov3D_precomp"] = cov
            r = run_reference(d, precomp_cov3D=cov)
        else:
            r = run_reference(d)
        for k, v in d.items():
            out[f"{name}_in_{k}"] = v
        for k in FIELDS:
            out[f"{name}_{k}"] = getattr(r, k)
        for k, v in r.grads.items():
            out[f"{name}_{k}"] = v
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_small.npz")
    np.savez_compressed(path, **compute())
    print("written", os.path.getsize(path), "bytes")
