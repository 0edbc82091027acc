"""ctypes binding of the C-ABI (include/gsr.h).

The product library is photo-slam_amd/libgsr_hip.so (hand-written HIP for gfx950).  There is
NO CPU fallback: if the library is missing, `load()` raises.  (The test-suite may pass the
path of tests/emu/libgsr_emu.so explicitly to exercise kernel logic without a GPU; nothing
in this package does.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "libgsr_hip.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

GSR_OK = 0


class GsrError(RuntimeError):
    def __init__(self, lib, status, where):
        msg = lib.gsr_strerror(status).decode()
        hip = lib.gsr_last_hip_error_string().decode()
        super().__init__(f"{where}: {msg} (status {status})" + (f" [{hip}]" if status == -3 and hip else ""))
        self.status = status


class ForwardArgs(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("background", C.c_void_p),
                ("width", C.c_int), ("height", C.c_int), ("means3D", C.c_void_p), ("shs", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("scale_modifier", C.c_float), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("cam_pos", C.c_void_p),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("prefiltered", C.c_int),
                ("out_color", C.c_void_p), ("radii", C.c_void_p), ("raw_params", C.c_int), ("sh_adam", C.c_void_p)]


SH_LAZY_WINDOW = 32   # GSR_SH_LAZY_WINDOW


class ShAdamLazy(C.Structure):
    _fields_ = [("row_step", C.c_void_p), ("window", C.c_int), ("lr_past", C.c_double * SH_LAZY_WINDOW),
                ("lr_tail_past", C.c_double * SH_LAZY_WINDOW)]


class ShAdam(C.Structure):
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("lr", C.c_double), ("lr_tail", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("step", C.c_int), ("lazy", C.POINTER(ShAdamLazy)),
                # scheduling switches, 0 = the measured-best arrangement (include/gsr.h)
                ("no_side_stream", C.c_int), ("lazy_slice_late", C.c_int), ("side_blocks", C.c_int)]


def make_sh_adam(sh, d):
    """gsr_sh_adam from the dict the Python host passes around: exp_avg, exp_avg_sq, lr, lr_tail, beta1, beta2, eps, step and --
    lazy mode (gsr_sh_adam_lazy) -- row_step (int32 [P]), window, lr_past / lr_tail_past (most recent step first).  Returns
    (struct, objects to keep alive)."""
    adam = ShAdam(sh.data_ptr(), d["exp_avg"].data_ptr(), d["exp_avg_sq"].data_ptr(), float(d["lr"]), float(d["lr_tail"]),
                  float(d["beta1"]), float(d["beta2"]), float(d["eps"]), int(d["step"]), None,
                  int(d.get("no_side_stream", 0)), int(d.get("lazy_slice_late", 0)), int(d.get("side_blocks", 0)))
    keep = [adam]
    if d.get("row_step") is not None:
        z = ShAdamLazy()
        z.row_step = d["row_step"].data_ptr()
        z.window = int(d["window"])
        for k, (a, b) in enumerate(zip(d.get("lr_past", ()), d.get("lr_tail_past", ()))):
            if k < SH_LAZY_WINDOW:
                z.lr_past[k], z.lr_tail_past[k] = float(a), float(b)
        adam.lazy = C.pointer(z)
        keep.append(z)
    return adam, keep


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("lr", C.c_double), ("step", C.c_int)]


class AdamMultiTensor(C.Structure):
    """gsr_adam_multi_tensor"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_longlong), ("lr", C.c_double), ("step", C.c_int), ("grad_scale", C.c_float)]


class GeomAdam(C.Structure):
    """gsr_geom_adam: xyz, opacity, scaling, rotation"""
    _fields_ = [("xyz", AdamTensor), ("opacity", AdamTensor), ("scaling", AdamTensor), ("rotation", AdamTensor),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


def make_geom_adam(d):
    """gsr_geom_adam from dict(tensors=[(param, exp_avg, exp_avg_sq, lr, step) x 4 in the order xyz, opacity, scaling, rotation],
    beta1, beta2, eps)."""
    g = GeomAdam()
    for name, (p, m, v, lr, step) in zip(("xyz", "opacity", "scaling", "rotation"), d["tensors"]):
        setattr(g, name, AdamTensor(p.data_ptr(), m.data_ptr(), v.data_ptr(), float(lr), int(step)))
    g.beta1, g.beta2, g.eps = float(d["beta1"]), float(d["beta2"]), float(d["eps"])
    return g


class BackwardArgs(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("R", C.c_int), ("background", C.c_void_p),
                ("width", C.c_int), ("height", C.c_int), ("means3D", C.c_void_p), ("shs", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("scales", C.c_void_p), ("scale_modifier", C.c_float),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("tan_fovx", C.c_float),
                ("tan_fovy", C.c_float), ("radii", C.c_void_p), ("geom_buffer", C.c_void_p),
                ("binning_buffer", C.c_void_p), ("image_buffer", C.c_void_p), ("dL_dpix", C.c_void_p),
                ("dL_dmean2D", C.c_void_p), ("dL_dconic", C.c_void_p), ("dL_dopacity", C.c_void_p),
                ("dL_dcolor", C.c_void_p), ("dL_dmean3D", C.c_void_p), ("dL_dcov3D", C.c_void_p),
                ("dL_dsh", C.c_void_p), ("dL_dscale", C.c_void_p), ("dL_drot", C.c_void_p), ("raw_params", C.c_int),
                ("dL_dcolor_view", C.c_void_p), ("sh_adam", C.POINTER(ShAdam)),
                ("stat_grad_accum", C.c_void_p), ("stat_denom", C.c_void_p), ("stat_max_radii", C.c_void_p),
                ("geom_adam", C.POINTER(GeomAdam)), ("color_view_ready_stream", C.c_void_p),
                ("packed_view", C.c_void_p), ("packed_capacity_rows", C.c_int)]

class DensifySelectArgs(C.Structure):
    _fields_ = [("P", C.c_int), ("xyz_gradient_accum", C.c_void_p), ("denom", C.c_void_p), ("scaling", C.c_void_p),
                ("opacity", C.c_void_p), ("percent_dense", C.c_float), ("max_grad", C.c_float), ("min_opacity", C.c_float),
                ("extent", C.c_float), ("max_screen_size", C.c_int), ("prune_mask", C.c_void_p)]


class DensifyGatherArgs(C.Structure):
    _fields_ = [("P", C.c_int), ("n_new", C.c_int), ("n_keep", C.c_int), ("n_clone", C.c_int), ("n_child", C.c_int),
                ("n_split", C.c_int), ("features_row_floats", C.c_int),
                ("param_in", C.c_void_p * 5), ("exp_avg_in", C.c_void_p * 5), ("exp_avg_sq_in", C.c_void_p * 5),
                ("param_out", C.c_void_p * 5), ("exp_avg_out", C.c_void_p * 5), ("exp_avg_sq_out", C.c_void_p * 5),
                ("samples", C.c_void_p), ("stats_out", C.c_void_p * 3), ("exist_since_iter_in", C.c_void_p),
                ("exist_since_iter_out", C.c_void_p), ("morton_scratch", C.c_void_p)]


RAW_OPACITY, RAW_SCALING, RAW_ROTATION = 1, 2, 4   # GSR_RAW_* of include/gsr.h


# every symbol include/gsr.h declares
EXPORTED_SYMBOLS = [
    "gsr_forward", "gsr_backward", "gsr_mark_visible", "gsr_knn_mean_dist2", "gsr_geometry_bytes", "gsr_binning_bytes",
    "gsr_image_bytes", "gsr_knn_scratch_bytes", "gsr_sh_grad_from_views", "gsr_sh_adam_from_views", "gsr_sh_adam_flush", "gsr_sh_adam_lazy_slice", "gsr_adam_step_multi", "gsr_strerror", "gsr_last_hip_error", "gsr_last_hip_error_string",
    "gsr_backend", "gsr_profile_enable", "gsr_profile_stage_count", "gsr_profile_stage_name", "gsr_profile_read",
    "gsr_loss_scratch_bytes", "gsr_l1_ssim_loss", "gsr_adam_step", "gsr_densify_stats", "gsr_densify_scratch_bytes",
    "gsr_densify_select", "gsr_densify_gather", "gsr_transform_points", "gsr_scale_transform_points", "gsr_reproject_depth_pinhole",
    "gsr_neighborhood_depth_pinhole", "gsr_packed_view_words", "gsr_pack_scratch_bytes", "gsr_pack_color_view", "gsr_pack_view_plan",
    "gsr_sh_grad_from_packed_views", "gsr_sh_adam_from_packed_views", "gsr_last_visible_count", "gsr_check_packed_views", "gsr_depth_resort_count",
    "gsr_host_wait_stats", "gsr_binning_tile_first", "gsr_densify_morton_scratch_bytes",
]

_libs = {}


def load(path=None):
    """Load the C-ABI library.  Default: the in-tree HIP build; raises if it does not exist."""
    path = os.path.abspath(path or HIP_LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build the HIP extension first (python photo-slam_amd/build.py or "
            f"__graft_entry__.build()); there is no CPU fallback.")
    L = C.CDLL(path)
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    L.gsr_forward.restype = i32
    L.gsr_forward.argtypes = [C.POINTER(ForwardArgs), ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, vp, C.POINTER(i32)]
    L.gsr_backward.restype = i32
    L.gsr_backward.argtypes = [C.POINTER(BackwardArgs), vp]
    L.gsr_sh_grad_from_views.restype = i32
    L.gsr_sh_grad_from_views.argtypes = [i32, i32, i32, i32, vp, vp, C.c_longlong, vp, C.c_longlong, f32, vp, vp]
    L.gsr_sh_adam_from_views.restype = i32
    L.gsr_sh_adam_from_views.argtypes = [i32, i32, i32, i32, vp, vp, C.c_longlong, vp, C.c_longlong, f32, vp, C.POINTER(ShAdam), vp]
    L.gsr_sh_adam_flush.restype = i32
    L.gsr_sh_adam_flush.argtypes = [i32, C.POINTER(ShAdam), vp]
    L.gsr_sh_adam_lazy_slice.restype = i32
    L.gsr_sh_adam_lazy_slice.argtypes = [i32, C.POINTER(ShAdam), i32, vp]
    L.gsr_adam_step_multi.restype = i32
    L.gsr_adam_step_multi.argtypes = [i32, C.POINTER(AdamMultiTensor), C.c_double, C.c_double, C.c_double, vp]
    L.gsr_mark_visible.restype = i32
    L.gsr_mark_visible.argtypes = [i32, vp, vp, vp, vp, vp]
    L.gsr_knn_mean_dist2.restype = i32
    L.gsr_knn_mean_dist2.argtypes = [i32, vp, vp, ALLOC_FN, vp, vp]
    for n in ("gsr_geometry_bytes", "gsr_binning_bytes", "gsr_knn_scratch_bytes"):
        getattr(L, n).restype = sz
        getattr(L, n).argtypes = [i32]
    L.gsr_image_bytes.restype = sz
    L.gsr_image_bytes.argtypes = [i32, i32]
    L.gsr_strerror.restype = C.c_char_p
    L.gsr_strerror.argtypes = [i32]
    L.gsr_last_hip_error.restype = i32
    L.gsr_last_hip_error_string.restype = C.c_char_p
    L.gsr_backend.restype = C.c_char_p
    L.gsr_profile_enable.restype = i32
    L.gsr_profile_enable.argtypes = [i32]
    L.gsr_profile_stage_count.restype = i32
    L.gsr_profile_stage_name.restype = C.c_char_p
    L.gsr_profile_stage_name.argtypes = [i32]
    L.gsr_profile_read.restype = i32
    L.gsr_profile_read.argtypes = [C.POINTER(f32), i32]
    L.gsr_loss_scratch_bytes.restype = sz
    L.gsr_loss_scratch_bytes.argtypes = [i32, i32]
    L.gsr_l1_ssim_loss.restype = i32
    L.gsr_l1_ssim_loss.argtypes = [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]
    L.gsr_adam_step.restype = i32
    f64 = C.c_double
    L.gsr_adam_step.argtypes = [vp, vp, vp, vp, C.c_longlong, f64, f64, f64, f64, i32, i32, i32, f64, vp]
    L.gsr_densify_stats.restype = i32
    L.gsr_densify_stats.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    L.gsr_densify_scratch_bytes.restype = sz
    L.gsr_densify_scratch_bytes.argtypes = [i32]
    L.gsr_densify_select.restype = i32
    L.gsr_densify_select.argtypes = [C.POINTER(DensifySelectArgs), vp, vp, vp]
    L.gsr_densify_morton_scratch_bytes.restype = sz
    L.gsr_densify_morton_scratch_bytes.argtypes = [i32]
    L.gsr_densify_gather.restype = i32
    L.gsr_densify_gather.argtypes = [C.POINTER(DensifyGatherArgs), vp, vp]
    L.gsr_transform_points.restype = i32
    L.gsr_transform_points.argtypes = [i32, vp, vp, vp, vp]
    L.gsr_scale_transform_points.restype = i32
    L.gsr_scale_transform_points.argtypes = [i32, f32, vp, vp, vp, vp, vp, vp, i32, vp]
    L.gsr_reproject_depth_pinhole.restype = i32
    L.gsr_reproject_depth_pinhole.argtypes = [i32, i32, f32, f32, f32, f32, vp, vp, vp, vp]
    L.gsr_neighborhood_depth_pinhole.restype = i32
    L.gsr_neighborhood_depth_pinhole.argtypes = [i32, i32, f32, f32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp]
    _libs[path] = L
    return L


def check(lib, status, where):
    if status != GSR_OK:
        raise GsrError(lib, status, where)


def profile_enable(lib, on=True):
    """True / 1: every stage; 2: only the backward blend (cheap enough for a timed region); False / 0: off"""
    check(lib, lib.gsr_profile_enable(int(on)), "gsr_profile_enable")


def profile_read(lib):
    """dict stage name -> milliseconds of the last forward/backward on this thread (-1 = did not run)"""
    n = lib.gsr_profile_stage_count()
    ms = (C.c_float * n)()
    check(lib, lib.gsr_profile_read(ms, n), "gsr_profile_read")
    return {lib.gsr_profile_stage_name(i).decode(): float(ms[i]) for i in range(n)}


def host_wait_stats(lib, reset=True):
    """gsr_host_wait_stats: (microseconds the calling thread was blocked in gsr_forward's host synchronisation, number of waits)"""
    us, n = C.c_double(0.0), C.c_longlong(0)
    lib.gsr_host_wait_stats(C.byref(us), C.byref(n), 1 if reset else 0)
    return float(us.value), int(n.value)
