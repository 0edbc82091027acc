"""Link-level drop-in check of the LibTorch boundary: tests/ref_link/consumer.cpp is compiled against the
REFERENCE's own declarations (include/rasterize_points.h, include/gaussian_rasterizer.h -- oracle/build_ref.py:
build_link_consumer) and linked against this repository's host library.  It constructs GaussianRasterizationSettings and
GaussianRasterizer with the reference's layout, calls the free functions with the reference's mangled names, and must get
the oracle's numbers."""
import os
import re
import subprocess

import pytest

from test_c_abi_consumer import _check_against_oracle


def _consumer(kind):
    from oracle import build_ref
    exe = build_ref.build_link_consumer().get(kind)
    if exe is None:
        pytest.skip("oracle/_ref/ref_link_consumer_* was never built (no reference tree, no prebuilt binary)")
    return exe


def _run(exe, *args):
    out = subprocess.check_output([exe] + list(args), text=True, env=dict(os.environ, PYTEST_CURRENT_TEST="ref_link"))
    m = re.search(r"class_vs_free=([-\d.e+]+) present=(\d+) vis=(\d+)", out)
    assert m, out
    # GaussianRasterizer / autograd == the free functions (two backward passes: on the GPU they differ by the order of the
    # blend's LDS adds, one ulp of the compared sums)
    assert float(m.group(1)) < 1e-5, out
    assert int(m.group(2)) == int(m.group(3)) == 5, out
    assert "reference_exception_text=1" in out, out
    return out


def test_reference_declarations_link_and_run_on_the_emulator_host(emu_lib_path, oracle):
    _check_against_oracle(_run(_consumer("emu")), oracle, "emu-wave64")


@pytest.mark.gpu
def test_reference_declarations_link_and_run_on_the_gpu(oracle):
    _check_against_oracle(_run(_consumer("hip"), "cuda"), oracle, "hip-gfx950")


def test_hip_consumer_loads_one_rocm_runtime_and_fails_loudly_without_a_device():
    """The executable linked against lib/libcuda_rasterizer.so + lib/libsimple_knn.so + libphotoslam_host.so must come up with ONE
    ROCm stack (round 4: CMake's ${TORCH_LIBRARIES} pulled the system's MIOpen / hipBLAS next to the wheel's bundled copies and the
    process died inside library initialisation) and, on a box without a GPU, stop at the first HIP call with the library's own
    message -- no CPU fallback, no segmentation fault."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the gpu-marked twin runs the consumer for real")
    p = subprocess.run([_consumer("hip")], capture_output=True, text=True)
    assert p.returncode == -6, (p.returncode, p.stderr[-400:])      # std::terminate on the uncaught std::runtime_error
    assert "HIP runtime error" in p.stderr and "no ROCm-capable device" in p.stderr, p.stderr[-400:]


def test_library_exports_the_reference_symbols():
    """the mangled names an object compiled against the reference headers asks for"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "photo-slam_amd", "host"))
    import build_host
    build_host.build("emu")
    libs = build_host.outputs("emu")
    boundary = subprocess.check_output(["nm", "-D", "--defined-only", "-C", libs["cuda_rasterizer"]], text=True)
    syms = boundary + subprocess.check_output(["nm", "-D", "--defined-only", "-C", libs["photoslam_host"]], text=True)
    ref_fwd = ("RasterizeGaussiansCUDA(at::Tensor const&, at::Tensor const&, at::Tensor const&, at::Tensor const&, at::Tensor const&, "
               "at::Tensor const&, float, at::Tensor const&, at::Tensor const&, at::Tensor const&, float, float, int, int, "
               "at::Tensor const&, int, at::Tensor const&, bool)")
    ref_bwd = ("RasterizeGaussiansBackwardCUDA(at::Tensor const&, at::Tensor const&, at::Tensor const&, at::Tensor const&, "
               "at::Tensor const&, at::Tensor const&, float, at::Tensor const&, at::Tensor const&, at::Tensor const&, float, float, "
               "at::Tensor const&, at::Tensor const&, int, at::Tensor const&, at::Tensor const&, int, at::Tensor const&, "
               "at::Tensor const&)")
    for want in (ref_fwd, ref_bwd, "markVisible(at::Tensor&, at::Tensor&, at::Tensor&)", "distCUDA2(at::Tensor const&)",
                 "GaussianRasterizer::markVisibleGaussians(at::Tensor&)", "GaussianRasterizerFunction::backward(",
                 "GaussianRasterizerFunction::forward(torch::autograd::AutogradContext*, at::Tensor, at::Tensor, at::Tensor, "
                 "at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, GaussianRasterizationSettings)"):
        assert want in syms, want
    # the reference's `cuda_rasterizer` / `simple_knn` libraries export the free functions; the classes above them are the
    # reference's own src/gaussian_rasterizer.cpp (or this repository's, in the layer on top)
    for want in (ref_fwd, ref_bwd, "markVisible(at::Tensor&, at::Tensor&, at::Tensor&)", "distCUDA2(at::Tensor const&)"):
        assert want in boundary, want
    assert "GaussianRasterizer::" not in boundary
