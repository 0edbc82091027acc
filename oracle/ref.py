"""ctypes binding of oracle/_ref/libref_rasterizer.so: the REFERENCE's own rasterizer sources compiled for the host
(oracle/build_ref.py).  TEST INFRASTRUCTURE ONLY -- used to pin the CPU oracle and to make tests/golden fixtures."""
import ctypes as C
import os

import numpy as np

from . import build_ref

_LIB = None
_fp = C.POINTER(C.c_float)


class _State(C.Structure):
    _fields_ = [("P", C.c_int), ("W", C.c_int), ("H", C.c_int), ("R", C.c_int), ("buf", C.c_void_p),
                ("depths", _fp), ("clamped", C.POINTER(C.c_uint8)), ("radii", C.POINTER(C.c_int)), ("means2D", _fp),
                ("cov3D", _fp), ("conic_opacity", _fp), ("rgb", _fp), ("point_offsets", C.POINTER(C.c_uint32)),
                ("tiles_touched", C.POINTER(C.c_uint32)), ("keys_unsorted", C.POINTER(C.c_uint64)),
                ("keys_sorted", C.POINTER(C.c_uint64)), ("vals_unsorted", C.POINTER(C.c_uint32)),
                ("point_list", C.POINTER(C.c_uint32)), ("ranges", C.POINTER(C.c_uint32)),
                ("n_contrib", C.POINTER(C.c_uint32)), ("accum_alpha", _fp)]


def available():
    return build_ref.available() or os.path.exists(build_ref.OUT)


def lib():
    global _LIB
    if _LIB is None:
        path = build_ref.build()
        if path is None:
            raise RuntimeError("reference rasterizer library unavailable (no /root/reference and no prebuilt oracle/_ref)")
        L = C.CDLL(path)
        L.ref_forward.restype = C.POINTER(_State)
        L.ref_forward.argtypes = [C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_float, _fp, _fp,
                                  _fp, _fp, _fp, C.c_float, C.c_float, C.c_int, _fp, C.POINTER(C.c_int)]
        L.ref_backward.restype = None
        L.ref_backward.argtypes = [C.POINTER(_State), C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_float, _fp, _fp, _fp, _fp, _fp,
                                   C.c_float, C.c_float, C.POINTER(C.c_int)] + [_fp] * 10
        L.ref_free.argtypes = [C.POINTER(_State)]
        u8p = C.POINTER(C.c_uint8)
        L.ref_transform_points.argtypes = [C.c_int, _fp, _fp, _fp]
        L.ref_scale_transform_points.argtypes = [C.c_int, C.c_float, _fp, _fp, _fp, u8p, _fp, _fp]
        L.ref_reproject_depth_pinhole.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _fp, u8p, _fp]
        L.ref_neighborhood_depth_pinhole.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _fp, u8p,
                                                     _fp, _fp, _fp, _fp]
        for fn in (L.ref_transform_points, L.ref_scale_transform_points, L.ref_reproject_depth_pinhole,
                   L.ref_neighborhood_depth_pinhole):
            fn.restype = None
        L.ref_knn.restype = None
        L.ref_knn.argtypes = [C.c_int, _fp, _fp]
        _LIB = L
    return _LIB


def _f(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(_fp)


class RefResult:
    """Copies of everything the reference keeps in its Geometry/Binning/Image states (rasterizer_impl.h:32-62)."""


def forward_backward(bg, means3D, opacity, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, shs=None, sh_degree=3,
                     colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, dL_dpix=None):
    L = lib()
    P = int(means3D.shape[0])
    M = int(shs.shape[1]) if shs is not None else 0
    keep = [_f(x) for x in (bg, means3D, shs, colors_precomp, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                            campos)]
    (k_bg, p_bg), (k_m, p_m), (k_sh, p_sh), (k_c, p_c), (k_o, p_o), (k_s, p_s), (k_r, p_r), (k_cov, p_cov), (k_v, p_v), \
        (k_p, p_p), (k_cam, p_cam) = keep
    out_color = np.zeros((3, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    st = L.ref_forward(P, sh_degree, M, p_bg, W, H, p_m, p_sh, p_c, p_o, p_s, scale_modifier, p_r, p_cov, p_v, p_p, p_cam,
                       tanfovx, tanfovy, 0, out_color.ctypes.data_as(_fp), radii.ctypes.data_as(C.POINTER(C.c_int)))
    s = st.contents
    r = RefResult()
    r.P, r.W, r.H, r.R = P, W, H, int(s.R)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)
    r.out_color, r.radii = out_color, radii
    r.depths = arr(s.depths, P, np.float32)
    r.clamped = arr(s.clamped, 3 * P, np.uint8).reshape(P, 3)
    r.means2D = arr(s.means2D, 2 * P, np.float32).reshape(P, 2)
    r.cov3D = arr(s.cov3D, 6 * P, np.float32).reshape(P, 6)
    r.conic_opacity = arr(s.conic_opacity, 4 * P, np.float32).reshape(P, 4)
    r.rgb = arr(s.rgb, 3 * P, np.float32).reshape(P, 3)
    r.point_offsets = arr(s.point_offsets, P, np.uint32)
    r.tiles_touched = arr(s.tiles_touched, P, np.uint32)
    r.keys_sorted = arr(s.keys_sorted, r.R, np.uint64)
    r.point_list = arr(s.point_list, r.R, np.uint32)
    r.ranges = arr(s.ranges, 2 * T, np.uint32).reshape(T, 2)
    r.n_contrib = arr(s.n_contrib, W * H, np.uint32).reshape(H, W)
    r.final_T = arr(s.accum_alpha, W * H, np.float32).reshape(H, W)
    r.grads = None
    if dL_dpix is not None:
        k_d, p_d = _f(dL_dpix)
        # torch::zeros in the reference wrapper (src/rasterize_points.cu:149-157)
        g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
                 dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
                 dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
                 dL_dsh=np.zeros((P, max(M, 1), 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
                 dL_drotations=np.zeros((P, 4), np.float32))
        ptr = lambda k: g[k].ctypes.data_as(_fp)
        L.ref_backward(st, sh_degree, M, p_bg, p_m, p_sh, p_c, p_s, scale_modifier, p_r, p_cov, p_v, p_p, p_cam, tanfovx, tanfovy,
                       radii.ctypes.data_as(C.POINTER(C.c_int)), p_d, ptr("dL_dmeans2D"), ptr("dL_dconic"), ptr("dL_dopacity"),
                       ptr("dL_dcolors"), ptr("dL_dmeans3D"), ptr("dL_dcov3D"), ptr("dL_dsh"), ptr("dL_dscales"),
                       ptr("dL_drotations"))
        if M == 0:
            g["dL_dsh"] = np.zeros((P, 0, 3), np.float32)
        r.grads = g
    L.ref_free(st)
    return r


def knn(points):
    """SimpleKNN::knn of the reference (third_party/simple-knn/simple_knn.cu:185-221): mean squared distance to the 3 nearest."""
    k, p = _f(points)
    out = np.zeros(k.shape[0], np.float32)
    lib().ref_knn(k.shape[0], p, out.ctypes.data_as(_fp))
    return out


def _u8(a):
    a = np.ascontiguousarray(a, np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


# Photo-SLAM's point kernels (src/operate_points.cu:38-71, src/stereo_vision.cu:39-136), same signatures as oracle.py
def transform_points(points, m):
    k1, p1 = _f(points); k2, p2 = _f(m)
    out = np.zeros_like(k1)
    lib().ref_transform_points(k1.shape[0], p1, p2, out.ctypes.data_as(_fp))
    return out


def scale_transform_points(scale, points, rots, m, mask):
    k1, p1 = _f(points); k2, p2 = _f(rots); k3, p3 = _f(m); k4, p4 = _u8(mask)
    op, orot = np.zeros_like(k1), np.zeros_like(k2)
    lib().ref_scale_transform_points(k1.shape[0], scale, p1, p2, p3, p4, op.ctypes.data_as(_fp), orot.ctypes.data_as(_fp))
    return op, orot


def reproject_depth_pinhole(depth, mask, intr, width):
    k1, p1 = _f(depth); k2, p2 = _u8(mask)
    out = np.zeros((k1.shape[0], 3), np.float32)
    lib().ref_reproject_depth_pinhole(k1.shape[0], width, *[float(x) for x in intr], p1, p2, out.ctypes.data_as(_fp))
    return out


def neighborhood_depth_pinhole(pixels, has3D, p3d, colors, max_pixel_dist, intr, width):
    k1, p1 = _f(pixels); k2, p2 = _u8(has3D); k3, p3 = _f(p3d); k4, p4 = _f(colors)
    op, oc = np.zeros_like(k3), np.zeros_like(k3)
    lib().ref_neighborhood_depth_pinhole(k1.shape[0], width, *[float(x) for x in intr], float(max_pixel_dist), p1, p2, p3, p4,
                                         op.ctypes.data_as(_fp), oc.ctypes.data_as(_fp))
    return op, oc
