"""ctypes access to the TEST-ONLY introspection library (tests/dev/gsr_dev.h): tests/dev/libgsr_dev.so next to the product
library on a GPU box, or the emulator build (which contains the same functions) on the host."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libgsr_dev.so")
PKG = os.path.join(ROOT, "photo-slam_amd")


class GeometryView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("depth_key", "tiles_touched", "radii", "rect", "rec", "cov3D", "clamped",
                                          "order", "offsets")]


class BinningView(C.Structure):
    _fields_ = [("point_list", C.c_void_p), ("tile_keys", C.c_void_p)]


class ImageView(C.Structure):
    _fields_ = [("final_T", C.c_void_p), ("n_contrib", C.c_void_p), ("ranges", C.c_void_p)]


def build(force=False):
    """tests/dev/libgsr_dev.so: gsr_dev.cpp (host code) against the HIP headers, linked to photo-slam_amd/libgsr_hip.so."""
    src = os.path.join(HERE, "gsr_dev.cpp")
    hip = os.path.join(PKG, "libgsr_hip.so")
    deps = [src, os.path.join(HERE, "gsr_dev.h"), hip, os.path.join(PKG, "csrc", "state.h"), os.path.join(PKG, "csrc", "rt.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-I/opt/rocm/include",
                           "-I" + os.path.join(PKG, "csrc"), "-I" + HERE, src, "-o", OUT, "-L" + PKG, "-lgsr_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../../photo-slam_amd"])
    return OUT


_libs = {}


def load(lib_path=None):
    """lib_path None: the HIP build (libgsr_dev.so, built on demand); else the emulator library, which contains the hooks."""
    path = os.path.abspath(lib_path) if lib_path else build()
    if path in _libs:
        return _libs[path]
    L = C.CDLL(path)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.gsr_stage_tile_depth_sort.restype = i32
    L.gsr_stage_tile_depth_sort.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.gsr_view_geometry.restype = i32
    L.gsr_view_geometry.argtypes = [vp, i32, C.POINTER(GeometryView)]
    L.gsr_view_binning.restype = i32
    L.gsr_view_binning.argtypes = [vp, i32, i32, i32, C.POINTER(BinningView)]
    L.gsr_view_image.restype = i32
    L.gsr_view_image.argtypes = [vp, i32, i32, C.POINTER(ImageView)]
    L.gsr_stage_scan_u32.restype = i32
    L.gsr_stage_scan_u32.argtypes = [vp, vp, i32, i32, vp, vp]
    L.gsr_stage_radix_sort_pairs.restype = i32
    L.gsr_stage_radix_sort_pairs.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    for n in ("gsr_scan_scratch_bytes", "gsr_sort_scratch_bytes"):
        getattr(L, n).restype = sz
        getattr(L, n).argtypes = [i32]
    _libs[path] = L
    return L


if __name__ == "__main__":
    print(build(force=True))
