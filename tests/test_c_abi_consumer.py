"""include/gsr.h is a C header: a strict-C99 program (tests/c_abi/consumer.c: malloc'd buffers, C callbacks, no C++ and no
torch anywhere) compiles against it, links the C-ABI library and gets the oracle's image."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CC = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
      os.path.join(ROOT, "tests", "c_abi", "consumer.c")]


def test_plain_c_consumer_matches_the_oracle(emu_lib_path, oracle, tmp_path):
    exe = str(tmp_path / "consumer")
    emu_dir = os.path.dirname(emu_lib_path)
    subprocess.check_call(CC + ["-o", exe, "-L" + emu_dir, "-lgsr_emu", "-lm", "-Wl,-rpath," + emu_dir])
    _check_against_oracle(subprocess.check_output([exe], text=True), oracle, "emu-wave64")


@pytest.mark.gpu
def test_plain_c_consumer_on_the_gpu(oracle, tmp_path):
    """The same C99 program against the product library: hipMalloc'd buffers through HIP's C runtime API, device pointers
    into gsr_forward / gsr_backward, no C++ and no torch in the process."""
    from photo_slam_amd import capi
    capi.load()   # raises when libgsr_hip.so is missing
    exe = str(tmp_path / "consumer_hip")
    lib_dir = os.path.dirname(capi.HIP_LIB_PATH)
    # HIP's own headers are not -pedantic clean (anonymous structs etc.): -isystem keeps -Werror for OUR header and source
    subprocess.check_call(CC + ["-DGSR_CONSUMER_HIP", "-D__HIP_PLATFORM_AMD__", "-isystem", "/opt/rocm/include", "-o", exe,
                                "-L" + lib_dir, "-lgsr_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                                "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    _check_against_oracle(subprocess.check_output([exe], text=True), oracle, "hip-gfx950")


def _check_against_oracle(out, oracle, backend):
    m = re.search(r"backend=(\S+) R=(\d+) visible=(\d+) image_sum=([-\d.]+) grad_sum=([-\d.]+)", out)
    assert m, out
    # the same five Gaussians through the oracle
    P, W, H, M = 5, 40, 24, 16
    i = np.arange(P, dtype=np.float32)
    means = np.stack([-0.6 + 0.3 * i, 0.1 * (i - 2), 2.0 + 0.25 * i], 1).astype(np.float32)
    opac = (0.5 + 0.08 * i).astype(np.float32)[:, None]
    scales = np.stack([np.full(P, 0.12), 0.08 + 0.01 * i, np.full(P, 0.1)], 1).astype(np.float32)
    rots = np.stack([np.ones(P), 0.1 * i, np.zeros(P), np.full(P, 0.05)], 1).astype(np.float32)
    sh = np.zeros((P, M, 3), np.float32)
    for k in range(P):
        for c in range(3):
            sh[k, 0, c] = 0.3 + 0.2 * ((k + c) % 3)
    tanfov, zn, zf = 0.6, 0.01, 100.0
    view = np.eye(4, dtype=np.float32)
    proj = np.zeros((4, 4), np.float32)
    proj[0, 0] = proj[1, 1] = 1.0 / tanfov
    proj[2, 2], proj[2, 3], proj[3, 2] = zf / (zf - zn), 1.0, -(zf * zn) / (zf - zn)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    res, color, radii = oracle.forward(bg, means, opac, view, proj, np.zeros(3, np.float32), tanfov, tanfov, H, W, shs=sh,
                                       sh_degree=0, scales=scales, rotations=rots)
    grads = oracle.backward(res, np.ones((3, H, W), np.float32))
    assert m.group(1) == backend
    assert int(m.group(2)) == int(res.tiles_touched.sum()) and int(m.group(3)) == int((radii > 0).sum()) == P
    assert np.isclose(float(m.group(4)), float(color.astype(np.float64).sum()), rtol=1e-5)
    gsum = sum(float(np.abs(grads[k][:, j] if grads[k].ndim == 2 else grads[k].reshape(P, -1)[:, j]).sum())
               for k, j in (("dL_dopacity", 0), ("dL_dmeans3D", 0), ("dL_dscales", 0), ("dL_drotations", 1), ("dL_dsh", 0)))
    assert np.isclose(float(m.group(5)), gsum, rtol=1e-4)
