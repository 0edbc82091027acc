"""Host mirror of include/operate_points.h and include/stereo_vision.h of the reference:

  transformPoints                                     src/operate_points.cu:73-93
  scaleAndTransformThenMarkVisiblePoints              src/operate_points.cu:95-143
  reprojectDepthPinhole                               src/stereo_vision.cu:138-167
  monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints   src/stereo_vision.cu:172-215

Same names, argument meaning and results; compute happens in libgsr_hip.so (csrc/points.hip)."""
import torch

from . import capi
from . import rasterize_points as rp


def _p(t):
    return t.data_ptr()


def transformPoints(points, transformmatrix):
    """Returns the transformed points (the reference re-binds its `points` argument to them)."""
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    lib = rp._lib()
    P = points.size(0)
    if P == 0:
        return points
    pts, m = points.contiguous().float(), transformmatrix.contiguous().float()
    out = torch.zeros_like(pts)
    capi.check(lib, lib.gsr_transform_points(P, _p(pts), _p(m), _p(out), rp._stream_ptr(pts)), "transformPoints")
    return out


def scaleAndTransformThenMarkVisiblePoints(points, rots, point_not_transformed_mask, point_unstable_mask, transformmatrix,
                                           viewmatrix, projmatrix, num_transformed, scale=1.0, reference_rot_layout=True):
    """In-place on points / rots / point_not_transformed_mask like the reference; returns the new num_transformed.
    reference_rot_layout=True reproduces insert_rot_to_rots as shipped (see include/gsr.h)."""
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    lib = rp._lib()
    present = rp.markVisible(points, viewmatrix, projmatrix)
    n = present.size(0)
    if point_not_transformed_mask.size(0) != n or point_unstable_mask.size(0) != n:
        raise RuntimeError("points_mask must have dimensions (num_points)")
    final_mask = point_not_transformed_mask & point_unstable_mask & present
    num_transformed += int(final_mask.sum().item())
    P = points.size(0)
    if P != 0:
        pts, r, m = points.contiguous().float(), rots.contiguous().float(), transformmatrix.contiguous().float()
        tp, tr = torch.zeros_like(pts), torch.zeros_like(r)
        mk = final_mask.to(torch.uint8).contiguous()
        capi.check(lib, lib.gsr_scale_transform_points(P, float(scale), _p(pts), _p(r), _p(m), _p(mk), _p(tp), _p(tr),
                                                       int(reference_rot_layout), rp._stream_ptr(pts)),
                   "scaleAndTransformThenMarkVisiblePoints")
        points[final_mask] = tp[final_mask]
        rots[final_mask] = tr[final_mask]
        point_not_transformed_mask[final_mask] = False
    return num_transformed


def reprojectDepthPinhole(depth, mask, intr, width):
    if depth.dim() != 1:
        raise RuntimeError("points must have dimensions (num_points)")
    lib = rp._lib()
    P = depth.size(0)
    if P == 0:
        return torch.Tensor()
    d, mk = depth.contiguous().float(), mask.to(torch.uint8).contiguous()
    points = torch.zeros((P, 3), dtype=torch.float32, device=depth.device)
    capi.check(lib, lib.gsr_reproject_depth_pinhole(P, int(width), *[float(x) for x in intr[:4]], _p(d), _p(mk), _p(points),
                                                    rp._stream_ptr(d)), "reprojectDepthPinhole")
    return points


def monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(kps_pixel, kps_has3D, kps_point_local, colors,
                                                                       max_pixel_dist, intr, width):
    if kps_pixel.dim() != 2 or kps_pixel.size(1) != 2:
        raise RuntimeError("kps_pixel must have dimensions (num_points, 2)")
    if kps_has3D.dim() != 1:
        raise RuntimeError("kps_has3D must have dimensions (num_points)")
    if kps_point_local.dim() != 2 or kps_point_local.size(1) != 3:
        raise RuntimeError("kps_point_local must have dimensions (num_points, 3)")
    lib = rp._lib()
    N = kps_pixel.size(0)
    if N == 0:
        return torch.Tensor(), torch.Tensor()
    px, has, p3, col = (kps_pixel.contiguous().float(), kps_has3D.to(torch.uint8).contiguous(),
                        kps_point_local.contiguous().float(), colors.contiguous().float())
    rp_, rc = torch.zeros_like(p3), torch.zeros_like(p3)
    capi.check(lib, lib.gsr_neighborhood_depth_pinhole(N, int(width), *[float(x) for x in intr[:4]], float(max_pixel_dist),
                                                       _p(px), _p(has), _p(p3), _p(col), _p(rp_), _p(rc),
                                                       rp._stream_ptr(px)),
               "monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints")
    valid = rp_[:, 2] > 0.0
    return rp_[valid], rc[valid]
