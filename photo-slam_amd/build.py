"""Builds libgsr_hip.so (the C-ABI product library) for gfx950, in-tree, by driving the repository's CMakeLists.txt -- the ONE
description of sources and flags (HIP language of CMake = clang++ -x hip --offload-arch=gfx950, what hipcc runs).

  python photo-slam_amd/build.py [--force] [-v]

hipcc / clang cross-compile without a GPU.  Translation units whose results must be bit-comparable with the CPU oracle (tile
rectangles, kNN distances) are compiled with -ffp-contract=off; the blend kernels keep the default fast contraction (FMA).
-munsafe-fp-atomics selects the hardware global_atomic_add_f32 / ds_add_f32 instead of CAS loops; -fno-slp-vectorize: see
CMakeLists.txt.  Experiment switches: GSR_EXTRA_FLAGS="-DGSR_EXP_..." (also "-save-temps=obj" for the ISA: the .s files land
next to the objects under build/cmake/CMakeFiles/gsr_hip.dir/).

On a box where the libraries arrive prebuilt (the GPU boxes: the snapshot carries the in-tree .so files) nothing is
configured or compiled: an up-to-date output is returned as it is.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgsr_hip.so")
BUILD_DIR = os.path.join(ROOT, "build", "cmake")
CMAKE_LISTS = os.path.join(ROOT, "CMakeLists.txt")


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d += [os.path.join(ROOT, "include", "gsr.h"), CMAKE_LISTS]   # the flags live in CMakeLists.txt
    return d


def up_to_date(outputs, deps):
    return all(os.path.exists(o) for o in outputs) and \
        all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps for o in outputs)


def cmake_build(targets, verbose=False, extra_flags="", force=False):
    """Configure (once, or when the experiment flags change) and build `targets` of the root CMakeLists.txt with Ninja.
    `force`: the targets' objects are dropped first (ninja decides by mtime, like up_to_date() above)."""
    import shutil
    import torch
    if force:
        for t in targets:
            shutil.rmtree(os.path.join(BUILD_DIR, "CMakeFiles", t + ".dir"), ignore_errors=True)
    prefix = torch.utils.cmake_prefix_path + ";/opt/rocm"
    cache = os.path.join(BUILD_DIR, "CMakeCache.txt")
    configured_flags = None
    if os.path.exists(cache):
        for line in open(cache):
            if line.startswith("GSR_EXTRA_FLAGS:"):
                configured_flags = line.split("=", 1)[1].rstrip("\n")
    quiet = None if verbose else subprocess.DEVNULL
    if configured_flags != extra_flags or not os.path.exists(os.path.join(BUILD_DIR, "build.ninja")) or \
            os.path.getmtime(CMAKE_LISTS) > os.path.getmtime(os.path.join(BUILD_DIR, "build.ninja")):
        subprocess.check_call(["cmake", "-S", ROOT, "-B", BUILD_DIR, "-G", "Ninja", "-DCMAKE_PREFIX_PATH=" + prefix,
                               "-DGSR_EXTRA_FLAGS=" + extra_flags], stdout=quiet, stderr=quiet)
    cmd = ["cmake", "--build", BUILD_DIR, "--target"] + list(targets)
    if verbose:
        cmd += ["--verbose"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("cmake --build failed for " + " ".join(targets))


def build(force=False, verbose=False):
    extra = " ".join(os.environ.get("GSR_EXTRA_FLAGS", "").split())   # experiment switches (-DGSR_EXP_...)
    stamp = OUT + ".flags"   # a library built with experiment switches must never be mistaken for the product build
    built_with = open(stamp).read() if os.path.exists(stamp) else ""
    if extra != built_with:
        force = True
    if not force and up_to_date([OUT], _deps()):
        return OUT
    cmake_build(["gsr_hip"], verbose=verbose, extra_flags=extra, force=force)
    os.utime(OUT)   # ninja leaves an up-to-date library alone; the mtime check above must see this build
    with open(stamp, "w") as f:
        f.write(extra)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
