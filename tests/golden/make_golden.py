"""Regenerates tests/golden/oracle_small.npz: outputs of the CPU oracle on a small seeded scene.

These vectors are oracle outputs (reference outputs are in reference_small.npz, see make_reference_golden.py); they
freeze the oracle together with the scene generator, so that later edits to either cannot silently change what the GPU
path is compared against.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from oracle import oracle  # noqa: E402
from photo_slam_amd import scene  # noqa: E402


def compute():
    cl = scene.make_cloud(200, 48, 32, 40.0, 40.0, seed=21, scale_k=0.4)
    cam = cl.cameras[0]
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    oracle.set_threads(1)
    res, img, radii = oracle.forward(bg, cl.xyz, cl.get_opacity(), cam.viewmatrix, cam.projmatrix, cam.campos, cam.tanfovx,
                                     cam.tanfovy, cam.H, cam.W, shs=cl.get_features(), sh_degree=3, scales=cl.get_scaling(),
                                     rotations=cl.get_rotation())
    dpix = np.random.default_rng(21).standard_normal((3, cam.H, cam.W)).astype(np.float32)
    g = oracle.backward(res, dpix)
    oracle.set_threads(0)
    out = dict(xyz=cl.xyz, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, out_color=img, radii=radii,
               tiles_touched=res.tiles_touched, point_list=res.point_list, ranges=res.ranges, n_contrib=res.n_contrib,
               final_T=res.final_T, means2D=res.means2D, conic_opacity=res.conic_opacity, rgb=res.rgb, dpix=dpix)
    out.update({k: v for k, v in g.items()})
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz"), **compute())
    print("written")
