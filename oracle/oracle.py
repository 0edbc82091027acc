"""ctypes binding of the CPU oracle (oracle/gsr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (photo-slam_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsr_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("gsr_oracle.c", "gsr_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class _State(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("P", "D", "M", "W", "H", "grid_x", "grid_y", "R", "sort_bits")] + [
        ("depths", C.POINTER(C.c_float)), ("clamped", C.POINTER(C.c_uint8)),
        ("radii", C.POINTER(C.c_int)), ("means2D", C.POINTER(C.c_float)),
        ("cov3D", C.POINTER(C.c_float)), ("conic_opacity", C.POINTER(C.c_float)),
        ("rgb", C.POINTER(C.c_float)), ("tiles_touched", C.POINTER(C.c_uint32)),
        ("point_offsets", C.POINTER(C.c_uint32)),
        ("keys_unsorted", C.POINTER(C.c_uint64)), ("vals_unsorted", C.POINTER(C.c_uint32)),
        ("keys_sorted", C.POINTER(C.c_uint64)), ("point_list", C.POINTER(C.c_uint32)),
        ("ranges", C.POINTER(C.c_uint32)), ("final_T", C.POINTER(C.c_float)),
        ("n_contrib", C.POINTER(C.c_uint32)), ("fragile", C.POINTER(C.c_uint8))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.gsro_forward.restype = C.POINTER(_State)
        L.gsro_forward.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp,
                                   C.c_float, fp, fp, fp, fp, fp, C.c_float, C.c_float, C.c_int, fp, ip]
        L.gsro_backward.restype = None
        L.gsro_backward.argtypes = [C.POINTER(_State), fp, fp, fp, fp, fp, C.c_float, fp, fp, fp, fp, fp,
                                    C.c_float, C.c_float, fp] + [fp] * 9
        L.gsro_free.argtypes = [C.POINTER(_State)]
        L.gsro_mark_visible.argtypes = [C.c_int, fp, fp, fp, C.POINTER(C.c_uint8)]
        L.gsro_knn.argtypes = [C.c_int, fp, fp]
        L.gsro_knn_bruteforce.argtypes = [C.c_int, fp, fp]
        L.gsro_higher_msb.restype = C.c_uint32
        L.gsro_higher_msb.argtypes = [C.c_uint32]
        u8 = C.POINTER(C.c_uint8)
        L.gsro_transform_points.argtypes = [C.c_int, fp, fp, fp]
        L.gsro_scale_transform_points.argtypes = [C.c_int, C.c_float, fp, fp, fp, u8, fp, fp, C.c_int]
        L.gsro_reproject_depth_pinhole.argtypes = [C.c_int, C.c_int] + [C.c_float] * 4 + [fp, u8, fp]
        L.gsro_neighborhood_depth_pinhole.argtypes = [C.c_int, C.c_int] + [C.c_float] * 5 + [fp, u8, fp, fp, fp, fp]
        L.gsro_cull_stats.argtypes = [C.POINTER(_State), C.POINTER(C.c_double)]
        L.gsro_set_threads.argtypes = [C.c_int]
        L.gsro_get_threads.restype = C.c_int
        _lib = L
    return _lib


def _f(a):
    """float32 C-contiguous array or None -> (keepalive, pointer)"""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.size == 0:
        return None, None
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def set_threads(n):
    lib().gsro_set_threads(int(n))


def get_threads():
    return int(lib().gsro_get_threads())


class ForwardResult:
    """Outputs + every intermediate of the oracle forward, as numpy arrays (copies)."""

    def __init__(self, st, out_color, radii, inputs):
        s = st.contents
        self._st = st
        self.inputs = inputs
        P, W, H, R = s.P, s.W, s.H, s.R
        T = s.grid_x * s.grid_y
        self.P, self.W, self.H, self.R, self.T = P, W, H, R, T
        self.grid = (s.grid_x, s.grid_y)
        self.sort_bits = s.sort_bits
        self.out_color = out_color
        self.radii = radii

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

        self.depths = arr(s.depths, P, np.float32)
        self.clamped = arr(s.clamped, 3 * P, np.uint8).reshape(P, 3)
        self.means2D = arr(s.means2D, 2 * P, np.float32).reshape(P, 2)
        self.cov3D = arr(s.cov3D, 6 * P, np.float32).reshape(P, 6)
        self.conic_opacity = arr(s.conic_opacity, 4 * P, np.float32).reshape(P, 4)
        self.rgb = arr(s.rgb, 3 * P, np.float32).reshape(P, 3)
        self.tiles_touched = arr(s.tiles_touched, P, np.uint32)
        self.point_offsets = arr(s.point_offsets, P, np.uint32)
        self.keys_unsorted = arr(s.keys_unsorted, R, np.uint64)
        self.vals_unsorted = arr(s.vals_unsorted, R, np.uint32)
        self.keys_sorted = arr(s.keys_sorted, R, np.uint64)
        self.point_list = arr(s.point_list, R, np.uint32)
        self.ranges = arr(s.ranges, 2 * T, np.uint32).reshape(T, 2)
        self.final_T = arr(s.final_T, W * H, np.float32).reshape(H, W)
        self.n_contrib = arr(s.n_contrib, W * H, np.uint32).reshape(H, W)
        self.fragile = arr(s.fragile, W * H, np.uint8).reshape(H, W)

    def free(self):
        if self._st is not None:
            lib().gsro_free(self._st)
            self._st = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def forward(background, means3D, opacities, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, H, W,
            shs=None, sh_degree=0, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            scale_modifier=1.0, prefiltered=False):
    """Mirrors RasterizeGaussiansCUDA (src/rasterize_points.cu:36-114) on numpy arrays."""
    L = lib()
    means3D = np.ascontiguousarray(means3D, np.float32)
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")
    P = means3D.shape[0]
    M = 0 if shs is None or np.size(shs) == 0 else np.asarray(shs).shape[1]
    out_color = np.zeros((3, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    keep = {}
    ptr = {}
    for name, a in dict(background=background, means3D=means3D, shs=shs, colors_precomp=colors_precomp,
                        opacities=opacities, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                        viewmatrix=viewmatrix, projmatrix=projmatrix, cam_pos=cam_pos).items():
        keep[name], ptr[name] = _f(a)
    if P == 0:
        return None, out_color, radii
    st = L.gsro_forward(P, sh_degree, M, ptr["background"], W, H, ptr["means3D"], ptr["shs"],
                        ptr["colors_precomp"], ptr["opacities"], ptr["scales"], scale_modifier,
                        ptr["rotations"], ptr["cov3D_precomp"], ptr["viewmatrix"], ptr["projmatrix"],
                        ptr["cam_pos"], tan_fovx, tan_fovy, int(prefiltered),
                        out_color.ctypes.data_as(C.POINTER(C.c_float)), radii.ctypes.data_as(C.POINTER(C.c_int)))
    keep.update(scale_modifier=scale_modifier, tan_fovx=tan_fovx, tan_fovy=tan_fovy, sh_degree=sh_degree, M=M)
    res = ForwardResult(st, out_color, radii, (keep, ptr))
    return res, out_color, radii


def backward(res, dL_dout_color):
    """Mirrors RasterizeGaussiansBackwardCUDA (src/rasterize_points.cu:116-193).
    Returns dict of numpy gradients (incl. the internal dL_dconic)."""
    L = lib()
    keep, ptr = res.inputs
    P, M = res.P, keep["M"]
    g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
             dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
             dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
             dL_drotations=np.zeros((P, 4), np.float32))
    dpix = np.ascontiguousarray(dL_dout_color, np.float32)
    fp = C.POINTER(C.c_float)

    def p(a):
        return a.ctypes.data_as(fp)

    L.gsro_backward(res._st, ptr["background"], ptr["means3D"], ptr["shs"], ptr["colors_precomp"], ptr["scales"],
                    keep["scale_modifier"], ptr["rotations"], ptr["cov3D_precomp"], ptr["viewmatrix"],
                    ptr["projmatrix"], ptr["cam_pos"], keep["tan_fovx"], keep["tan_fovy"], p(dpix),
                    p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dopacity"]), p(g["dL_dcolors"]),
                    p(g["dL_dmeans3D"]), p(g["dL_dcov3D"]), p(g["dL_dsh"]), p(g["dL_dscales"]),
                    p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    k1, p1 = _f(means3D)
    k2, p2 = _f(viewmatrix)
    k3, p3 = _f(projmatrix)
    P = 0 if k1 is None else k1.shape[0]
    out = np.zeros(P, np.uint8)
    if P:
        lib().gsro_mark_visible(P, p1, p2, p3, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)


def knn(points, bruteforce=False):
    k, p = _f(points)
    P = 0 if k is None else k.shape[0]
    out = np.zeros(P, np.float32)
    if P:
        fn = lib().gsro_knn_bruteforce if bruteforce else lib().gsro_knn
        fn(P, p, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def higher_msb(n):
    return int(lib().gsro_higher_msb(int(n)))


def cull_stats(res):
    out = (C.c_double * 16)()
    lib().gsro_cull_stats(res._st, out)
    names = ("list_entries", "bwd_staged_entries", "fwd_quad_visits", "bwd_quad_visits", "blended_pairs",
             "reference_fwd_pair_evals", "wrongly_rejected_pairs", "fwd_quad_visits_without_rejection",
             "quad_visits_with_a_blending_pixel", "tile_instances_with_a_blending_pixel",
             "bwd_visits_16x8_units", "bwd_visits_8x16_units", "block4x4_visits_with_a_blending_pixel",
             "longest_4x4_block_walk_per_quad", "longest_8x4_half_walk_per_quad", "half8x4_visits_with_a_blending_pixel")
    return dict(zip(names, [float(v) for v in out]))


def _u8(a):
    a = np.ascontiguousarray(a, np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def transform_points(points, m):
    k1, p1 = _f(points); k2, p2 = _f(m)
    out = np.zeros_like(k1)
    lib().gsro_transform_points(k1.shape[0], p1, p2, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def scale_transform_points(scale, points, rots, m, mask, reference_rot_layout=True):
    k1, p1 = _f(points); k2, p2 = _f(rots); k3, p3 = _f(m); k4, p4 = _u8(mask)
    op, orot = np.zeros_like(k1), np.zeros_like(k2)
    fp = C.POINTER(C.c_float)
    lib().gsro_scale_transform_points(k1.shape[0], scale, p1, p2, p3, p4, op.ctypes.data_as(fp), orot.ctypes.data_as(fp),
                                      int(reference_rot_layout))
    return op, orot


def reproject_depth_pinhole(depth, mask, intr, width):
    k1, p1 = _f(depth); k2, p2 = _u8(mask)
    out = np.zeros((k1.shape[0], 3), np.float32)
    lib().gsro_reproject_depth_pinhole(k1.shape[0], width, *[float(x) for x in intr], p1, p2, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def neighborhood_depth_pinhole(pixels, has3D, p3d, colors, max_pixel_dist, intr, width):
    k1, p1 = _f(pixels); k2, p2 = _u8(has3D); k3, p3 = _f(p3d); k4, p4 = _f(colors)
    op, oc = np.zeros_like(k3), np.zeros_like(k3)
    fp = C.POINTER(C.c_float)
    lib().gsro_neighborhood_depth_pinhole(k1.shape[0], width, *[float(x) for x in intr], float(max_pixel_dist), p1, p2, p3, p4,
                                          op.ctypes.data_as(fp), oc.ctypes.data_as(fp))
    return op, oc
