"""Cost of the fused SH Adam step (gsr_backward_args.sh_adam) at C3 on one GPU: backward preprocess stage + the separate
Adam pass on the [P,16,3] tensor, against the fused backward.   python tools/fused_adam_probe.py"""
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


def main():
    dev = torch.device("cuda:0")
    lib = capi.load()
    cl = scene.make_config("C3", seed=0, n_views=1)
    g = GaussianModel.from_cloud(cl, device=dev)
    cam = cl.cameras[0]
    kf = GaussianKeyframe.from_camera(cam, dev)
    P = cl.xyz.shape[0]
    bg = torch.zeros(3, device=dev)
    e = torch.empty(0, device=dev)
    out = {"P": P}
    with torch.no_grad():
        xyz, op, sc, rot, sh = g.xyz_, g.opacity_, g.scaling_, g.rotation_, g.features_.detach().clone()
        R, color, radii, geom, binning, img = rp.RasterizeGaussiansCUDA(
            bg, xyz, e, op, sc, rot, 1.0, e, kf.world_view_transform_, kf.full_proj_transform_, kf.tanfovx_, kf.tanfovy_,
            cam.H, cam.W, sh, 3, kf.camera_center_, False, 7)
        dpix = torch.randn(3, cam.H, cam.W, device=dev)
        m, v = torch.zeros_like(sh), torch.zeros_like(sh)
        hyper = dict(lr=0.0025, lr_tail=0.0025 / 20, beta1=0.9, beta2=0.999, eps=1e-15, step=1)

        def bwd(adam):
            return rp.RasterizeGaussiansBackwardCUDA(bg, xyz, radii, e, sc, rot, 1.0, e, kf.world_view_transform_,
                                                     kf.full_proj_transform_, kf.tanfovx_, kf.tanfovy_, dpix, sh, 3,
                                                     kf.camera_center_, geom, R, binning, img, 7, None, adam)
        capi.profile_enable(lib, 1)
        for name, adam in (("plain", None), ("fused", dict(exp_avg=m, exp_avg_sq=v, **hyper))):
            ms = []
            for _ in range(8):
                bwd(adam)
                ms.append(capi.profile_read(lib)["preprocess_bwd"])
            out[f"preprocess_bwd_ms_{name}"] = round(float(np.mean(ms[2:])), 4)
        capi.profile_enable(lib, 0)
        grad = bwd(None)[5]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def adam_pass():
            capi.check(lib, lib.gsr_adam_step(sh.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), sh.numel(), 0.0025, 0.9,
                                              0.999, 1e-15, 1, 48, 3, 0.0025 / 20, rp._stream_ptr(sh)), "adam")
        adam_pass()
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            adam_pass()
        b.record()
        torch.cuda.synchronize()
        out["adam_features_ms"] = round(a.elapsed_time(b) / 10, 4)
        out["separate_total_ms"] = round(out["preprocess_bwd_ms_plain"] + out["adam_features_ms"], 4)
        out["saved_ms"] = round(out["separate_total_ms"] - out["preprocess_bwd_ms_fused"], 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
