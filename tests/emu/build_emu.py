"""Builds tests/emu/libgsr_emu.so: the product kernel sources compiled for the host against the
wave64 emulator (hip_emu.h).  Test infrastructure only -- never loaded by the product package."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "photo-slam_amd", "csrc")
OUT = os.path.join(HERE, "libgsr_emu.so")
SOURCES = ["gsr_api.hip", "preprocess.hip", "sort.hip", "binning.hip", "tile_depth_sort.hip", "blend_fwd.hip", "blend_bwd.hip",
           "preprocess_bwd.hip", "knn.hip", "train_ops.hip", "points.hip", "densify.hip"]


def build(force=False):
    dev = os.path.join(ROOT, "tests", "dev")   # the test-only introspection hooks are part of the emulator library
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "hip_emu.cpp"), os.path.join(dev, "gsr_dev.cpp")]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [
        os.path.join(HERE, "hip_emu.h"), os.path.join(ROOT, "include", "gsr.h"), os.path.join(dev, "gsr_dev.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        os.makedirs(os.path.dirname(o), exist_ok=True)
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-DGSR_EMU", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
               "-Wno-unused-variable", "-Wno-unknown-pragmas", "-Wno-sign-compare", "-I", HERE, "-I", CSRC, "-I", dev, "-x", "c++", "-c", s, "-o", o]
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("emu build failed: " + " ".join(cmd))
    subprocess.check_call(["g++", "-shared", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
