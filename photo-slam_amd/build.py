"""Builds libgsr_hip.so (the C-ABI product library) for gfx950 with hipcc, in-tree.

  python photo-slam_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Translation units whose results must be bit-comparable
with the CPU oracle (tile rectangles, kNN distances) are compiled with -ffp-contract=off; the
blend kernels keep the default fast contraction (FMA).  -munsafe-fp-atomics selects the
hardware global_atomic_add_f32 / ds_add_f32 instead of CAS loops.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgsr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# (source, extra flags)
SOURCES = [
    ("gsr_api.hip", []),
    ("preprocess.hip", ["-ffp-contract=off"]),
    ("preprocess_bwd.hip", ["-ffp-contract=off"]),
    ("knn.hip", ["-ffp-contract=off"]),
    ("points.hip", ["-ffp-contract=off"]),
    ("densify.hip", ["-ffp-contract=off"]),
    ("sort.hip", []),
    ("binning.hip", []),
    ("blend_fwd.hip", []),
    ("blend_bwd.hip", []),
    ("train_ops.hip", []),
]
# -fno-slp-vectorize everywhere: on gfx950 the SLP vectoriser's v_pk_*_f32 pairings cost more issue cycles than the two plain
# instructions they replace (pk_fma 5.6 against 2 x 2.5 for v_fmac, profiles/r02_a_valu_rate.json) AND need v_mov_b32s to
# pair their operands (loss_fwd: 298 pk_fma + 254 mov instead of 625 v_fmac); measured per kernel at C3: blend_fwd 221 -> 210 us,
# blend_bwd 592 -> 578 us, loss 134 -> 126 us (tools/gpu_r2m.sh).  The packed adds of the backward blend's butterfly are
# written by hand (blend.h) and stay.
COMMON = ["-std=c++17", "-O3", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-fno-slp-vectorize",
          "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d += [os.path.join(HERE, "..", "include", "gsr.h"), os.path.abspath(__file__)]   # the flags live in this file
    return d


def build(force=False, verbose=False, save_temps=False):
    extra_env = os.environ.get("GSR_EXTRA_FLAGS", "").split()   # experiment switches (-DGSR_EXP_...)
    stamp = OUT + ".flags"   # a library built with experiment switches must never be mistaken for the product build
    built_with = open(stamp).read() if os.path.exists(stamp) else ""
    if " ".join(extra_env) != built_with:
        force = True
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in _deps()):
        return OUT
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    for src, extra in SOURCES:
        o = os.path.join(bdir, src + ".o")
        cmd = [HIPCC] + COMMON + extra + extra_env + ["-c", os.path.join(CSRC, src), "-o", o]
        if save_temps:
            cmd += ["-save-temps=obj"]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd, cwd=bdir), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    subprocess.check_call([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs)
    with open(stamp, "w") as f:
        f.write(" ".join(extra_env))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, save_temps="--save-temps" in sys.argv))
