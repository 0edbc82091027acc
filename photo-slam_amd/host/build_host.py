"""Builds the LibTorch C++ host layer (photo-slam_amd/host) in-tree:

  libphotoslam_host.so       links libgsr_hip.so   (product; HIP stream from c10::hip)
  libphotoslam_host_emu.so   links tests/emu/libgsr_emu.so, -DGSR_HOST_NO_HIP (CPU tensors; test-suite only)

g++ against the LibTorch headers of the installed torch wheel (no hipify, no nvcc)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SRCS = ["rasterize_points.cpp", "gaussian_rasterizer.cpp", "train_step.cpp", "gaussian_model_densify.cpp", "ply_io.cpp", "operate_points.cpp",
        "keyframe_batch_exchange.cpp", "ops_register.cpp"]


def _torch_paths():
    import torch
    base = os.path.dirname(torch.__file__)
    inc = [os.path.join(base, "include"), os.path.join(base, "include", "torch", "csrc", "api", "include")]
    return inc, os.path.join(base, "lib"), torch._C._GLIBCXX_USE_CXX11_ABI


def build(variant="hip", force=False):
    inc, libdir, cxx11 = _torch_paths()
    if variant == "hip":
        out = os.path.join(HERE, "libphotoslam_host.so")
        gsr_dir, gsr_name = PKG, "gsr_hip"
        defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"]
        extra_inc = ["-I/opt/rocm/include"]
        libs = ["-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip"]
    else:
        out = os.path.join(ROOT, "tests", "emu", "libphotoslam_host_emu.so")
        gsr_dir, gsr_name = os.path.join(ROOT, "tests", "emu"), "gsr_emu"
        # a fresh checkout has no emulator library yet (it is git-ignored): build it before linking against it
        import importlib.util
        spec = importlib.util.spec_from_file_location("gsr_build_emu", os.path.join(gsr_dir, "build_emu.py"))
        emu = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(emu)
        emu.build()
        defs = ["-DGSR_HOST_NO_HIP=1"]
        extra_inc = []
        libs = ["-ltorch", "-ltorch_cpu", "-lc10"]
    srcs = [os.path.join(HERE, "src", s) for s in SRCS]
    deps = srcs + [os.path.join(HERE, "include", f) for f in os.listdir(os.path.join(HERE, "include"))] + \
        [os.path.join(ROOT, "include", "gsr.h"), os.path.join(gsr_dir, f"lib{gsr_name}.so")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    bdir = os.path.join(HERE, "build_" + variant)
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(bdir, os.path.basename(s) + ".o")
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", f"-D_GLIBCXX_USE_CXX11_ABI={int(cxx11)}", "-Wno-deprecated-declarations"] + \
            defs + ["-I" + os.path.join(HERE, "include")] + ["-I" + i for i in inc] + extra_inc + ["-c", s, "-o", o]
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("host build failed: " + " ".join(cmd))
    subprocess.check_call(["g++", "-shared", "-o", out] + objs + ["-L" + libdir, "-L" + gsr_dir, "-l" + gsr_name] + libs +
                          ["-Wl,-rpath," + libdir, "-Wl,-rpath," + gsr_dir, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."])
    return out


if __name__ == "__main__":
    print(build("emu" if "--emu" in sys.argv else "hip", force="--force" in sys.argv))
