"""Builds the LibTorch C++ host layer (photo-slam_amd/host) in-tree.

  variant "hip" (product; HIP stream from c10::hip) -- the targets of the repository's CMakeLists.txt:
      photo-slam_amd/lib/libcuda_rasterizer.so   RasterizeGaussiansCUDA / ...BackwardCUDA / markVisible + the point kernels'
                                                 wrappers: the library Photo-SLAM's gaussian_mapper links by that name
      photo-slam_amd/lib/libsimple_knn.so        distCUDA2, likewise
      photo-slam_amd/host/libphotoslam_host.so   this repository's own host layer on top of the two (returned)
  variant "emu" (test-suite only; -DGSR_HOST_NO_HIP, CPU tensors, links tests/emu/libgsr_emu.so), g++ directly:
      tests/emu/libcuda_rasterizer_emu.so        the same split: the reference's link-level boundary ...
      tests/emu/libphotoslam_host_emu.so         ... and the layer above it (returned)

g++ against the LibTorch headers of the installed torch wheel (no hipify, no nvcc)."""
import importlib.util
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
# the link-level boundary of the reference (its `cuda_rasterizer` + `simple_knn` libraries) and what this repository adds on top
BOUNDARY_SRCS = ["rasterize_points.cpp", "operate_points.cpp", "spatial.cpp", "loss_utils.cpp"]
HOST_SRCS = ["gaussian_rasterizer.cpp", "train_step.cpp", "gaussian_model_densify.cpp", "ply_io.cpp", "keyframe_batch_exchange.cpp",
             "keyframe_scheduler.cpp", "ops_register.cpp"]
HIP_OUT = {"cuda_rasterizer": os.path.join(PKG, "lib", "libcuda_rasterizer.so"), "simple_knn": os.path.join(PKG, "lib", "libsimple_knn.so"),
           "photoslam_host": os.path.join(HERE, "libphotoslam_host.so")}
EMU_OUT = {"cuda_rasterizer": os.path.join(ROOT, "tests", "emu", "libcuda_rasterizer_emu.so"),
           "photoslam_host": os.path.join(ROOT, "tests", "emu", "libphotoslam_host_emu.so")}


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _torch_paths():
    import torch
    base = os.path.dirname(torch.__file__)
    inc = [os.path.join(base, "include"), os.path.join(base, "include", "torch", "csrc", "api", "include")]
    return inc, os.path.join(base, "lib"), torch._C._GLIBCXX_USE_CXX11_ABI


def _host_deps():
    return [os.path.join(HERE, "src", s) for s in BOUNDARY_SRCS + HOST_SRCS] + \
        [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(HERE, "include")) for f in fs] + [os.path.join(ROOT, "include", "gsr.h")]


def outputs(variant="hip"):
    """{target name: path} of the variant's libraries (built or not)."""
    return dict(HIP_OUT if variant == "hip" else EMU_OUT)


def build(variant="hip", force=False):
    gsr_build = _load("gsr_build", os.path.join(PKG, "build.py"))
    if variant == "hip":
        gsr = gsr_build.build()
        deps = _host_deps() + [gsr, gsr_build.CMAKE_LISTS]
        if not force and gsr_build.up_to_date(HIP_OUT.values(), deps):
            return HIP_OUT["photoslam_host"]
        gsr_build.cmake_build(list(HIP_OUT), extra_flags=" ".join(os.environ.get("GSR_EXTRA_FLAGS", "").split()), force=force)
        for o in HIP_OUT.values():
            os.utime(o)
        return HIP_OUT["photoslam_host"]
    inc, libdir, cxx11 = _torch_paths()
    gsr_dir = os.path.join(ROOT, "tests", "emu")
    # a fresh checkout has no emulator library yet (it is git-ignored): build it before linking against it
    _load("gsr_build_emu", os.path.join(gsr_dir, "build_emu.py")).build()
    deps = _host_deps() + [os.path.join(gsr_dir, "libgsr_emu.so"), os.path.abspath(__file__)]
    if not force and gsr_build.up_to_date(EMU_OUT.values(), deps):
        return EMU_OUT["photoslam_host"]
    bdir = os.path.join(HERE, "build_emu")
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], {}
    for s in BOUNDARY_SRCS + HOST_SRCS:
        o = os.path.join(bdir, s + ".o")
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", f"-D_GLIBCXX_USE_CXX11_ABI={int(cxx11)}", "-Wno-deprecated-declarations",
               "-DGSR_HOST_NO_HIP=1", "-I" + os.path.join(HERE, "include")] + ["-I" + i for i in inc] + \
            ["-c", os.path.join(HERE, "src", s), "-o", o]
        procs.append((subprocess.Popen(cmd), cmd))
        objs[s] = o
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("host build failed: " + " ".join(cmd))
    common = ["-L" + libdir, "-L" + gsr_dir, "-lgsr_emu", "-ltorch", "-ltorch_cpu", "-lc10",
              "-Wl,-rpath," + libdir, "-Wl,-rpath," + gsr_dir, "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(["g++", "-shared", "-o", EMU_OUT["cuda_rasterizer"]] + [objs[s] for s in BOUNDARY_SRCS] + common)
    subprocess.check_call(["g++", "-shared", "-o", EMU_OUT["photoslam_host"]] + [objs[s] for s in HOST_SRCS] +
                          ["-L" + gsr_dir, "-lcuda_rasterizer_emu"] + common)
    return EMU_OUT["photoslam_host"]


if __name__ == "__main__":
    print(build("emu" if "--emu" in sys.argv else "hip", force="--force" in sys.argv))
