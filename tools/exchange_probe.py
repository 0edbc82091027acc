"""Cost of the view-factored gradient exchange on one GPU: the factored backward against the plain one, and
gsr_sh_grad_from_views on the colour gradients of N real views of the bench scene (N = 2, 4, 8).
  python tools/exchange_probe.py [--config C3]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from photo_slam_amd import capi, scene  # noqa: E402
from photo_slam_amd import rasterize_points as rp  # noqa: E402
from photo_slam_amd.gaussian_model import GaussianModel  # noqa: E402
from photo_slam_amd.gaussian_renderer import GaussianKeyframe  # noqa: E402


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = capi.load()
    cl = scene.make_config(args.config, seed=0, n_views=8)
    g = GaussianModel.from_cloud(cl, device=dev)
    P = cl.xyz.shape[0]
    bg = torch.zeros(3, device=dev)
    e = torch.empty(0, device=dev)
    out = {"config": args.config, "P": P}
    views, full = [], []
    with torch.no_grad():
        xyz, op, sc, rot, sh = g.xyz_, g.opacity_, g.scaling_, g.rotation_, g.features_
        for vi, cam in enumerate(cl.cameras):
            kf = GaussianKeyframe.from_camera(cam, dev)
            fwd = rp.RasterizeGaussiansCUDA(bg, xyz, e, op, sc, rot, 1.0, e, kf.world_view_transform_, kf.full_proj_transform_,
                                            kf.tanfovx_, kf.tanfovy_, cam.H, cam.W, sh, 3, kf.camera_center_, False, 7)
            R, color, radii, geom, binning, img = fwd
            dpix = torch.randn(3, cam.H, cam.W, device=dev)
            view = torch.empty(P, 3, device=dev)

            def bwd(v):
                return rp.RasterizeGaussiansBackwardCUDA(bg, xyz, radii, e, sc, rot, 1.0, e, kf.world_view_transform_,
                                                         kf.full_proj_transform_, kf.tanfovx_, kf.tanfovy_, dpix, sh, 3,
                                                         kf.camera_center_, geom, R, binning, img, 7, v)
            if vi == 0:
                capi.profile_enable(lib, 1)
                for name, v in (("plain", None), ("factored", view)):
                    ms = []
                    for _ in range(6):
                        bwd(v)
                        ms.append(capi.profile_read(lib)["preprocess_bwd"])
                    out[f"preprocess_bwd_ms_{name}"] = round(float(np.mean(ms[1:])), 4)
                capi.profile_enable(lib, 0)
            gr = bwd(None)
            bwd(view)
            views.append(view)
            if vi < 2:
                full.append(gr[5])
            out.setdefault("visible_fraction", []).append(round(float((radii > 0).float().mean()), 3))
        centres = torch.stack([torch.from_numpy(c.campos) for c in cl.cameras]).float().to(dev)
        stack = torch.stack(views)
        for n in (1, 2, 4, 8):
            ms = timed(lambda: rp.shGradFromViews(xyz, centres[:n], stack[:n], 3, 16, 1.0 / n))
            out[f"sh_grad_from_views_ms_n{n}"] = round(ms, 4)
            out[f"sh_grad_from_views_GBps_n{n}"] = round(P * (12 + 12 * n + 192) / (ms * 1e-3) / 1e9, 1)
        got = rp.shGradFromViews(xyz, centres[:2], stack[:2], 3, 16, 0.5)
        want = (full[0].double() + full[1].double()) * 0.5
        out["rel_l1_vs_mean_of_rows_n2"] = float((got.double() - want).abs().sum() / want.abs().sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
