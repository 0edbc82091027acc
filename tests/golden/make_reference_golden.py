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


# ---------------------------------------------------------------------------------------------------------------------------
# tests/golden/reference_C1.npz: the reference's sources on BASELINE config C1 (50 k Gaussians @ 640x480, seed 0) -- the size at
# which the paths the small cases cannot reach meet the reference ITSELF: tile lists longer than one 256-entry batch (303),
# 258 Gaussians with more than 64 tiles (the long-run sums of the backward pass), 212 730 instances = 104 radix-sort blocks
# in four block groups.  The inputs are NOT stored (12 MB): they are regenerated from the seed and checked against a digest.
# Stored: every integer output exactly, the image and final T at every second row (+ per-row sums of all rows), the gradients
# at the visible Gaussians (the others must be exactly zero), the SH gradient for every fourth visible Gaussian in full and as
# per-row L1 norms for all of them.


def c1_inputs():
    cl = scene.make_config("C1", seed=0)
    cam = cl.cameras[0]
    rng = np.random.default_rng(100)
    return dict(bg=np.array([0.1, 0.2, 0.3], np.float32), xyz=cl.xyz, opacity=cl.get_opacity(), features=cl.get_features(),
                scaling=cl.get_scaling(), rotation=cl.get_rotation(), viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix,
                campos=cam.campos, tanfov=np.array([cam.tanfovx, cam.tanfovy], np.float32),
                size=np.array([cam.W, cam.H, 3], np.int32), dpix=rng.standard_normal((3, cam.H, cam.W)).astype(np.float32))


def inputs_digest(d):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(d):
        h.update(k.encode())
        h.update(np.ascontiguousarray(d[k]).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


C1_GRADS = ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")


def reduce_c1(r_fields, grads):
    """The stored form of one result (a dict of FIELDS + the gradient dict): see the header of this section."""
    out = {}
    vis = r_fields["radii"] > 0
    for k in ("radii", "tiles_touched", "point_list", "n_contrib"):
        out[k] = r_fields[k]
    out["ranges"] = r_fields["ranges"].reshape(-1, 2)
    out["tile_ids"] = (r_fields["keys_sorted"] >> np.uint64(32)).astype(np.uint32)
    out["depth_bits"] = r_fields["depths"].view(np.uint32)[vis]
    out["clamped"] = np.packbits(r_fields["clamped"].reshape(-1, 3)[vis].astype(bool))
    for k in ("means2D", "conic_opacity", "rgb", "cov3D"):
        out[k] = r_fields[k][vis]
    for k in ("out_color", "final_T"):
        a = r_fields[k]
        img = a.reshape(-1, a.shape[-2], a.shape[-1])
        out[k + "_even_rows"] = img[:, ::2]
        out[k + "_row_sums"] = img.astype(np.float64).sum(-1)
    for k in C1_GRADS:
        g = grads[k].reshape(grads[k].shape[0], -1)
        assert not g[~vis].any(), k          # culled Gaussians: exactly zero
        out[k] = g[vis]
    sh = grads["dL_dsh"].reshape(vis.shape[0], -1)
    assert not sh[~vis].any()
    out["dL_dsh_every_4th"] = sh[vis][::4]
    out["dL_dsh_row_l1"] = np.abs(sh[vis].astype(np.float64)).sum(1)
    return out


def compute_c1():
    d = c1_inputs()
    r = run_reference(d)
    out = reduce_c1({k: getattr(r, k) for k in FIELDS}, r.grads)
    out["inputs_sha256"] = inputs_digest(d)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "reference_small.npz")
    np.savez_compressed(path, **compute())
    print("written", path, os.path.getsize(path), "bytes")
    path = os.path.join(here, "reference_C1.npz")
    np.savez_compressed(path, **compute_c1())
    print("written", path, os.path.getsize(path), "bytes")
