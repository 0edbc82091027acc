"""The C-ABI library loads on a machine without a GPU and exports every function that
include/gsr.h declares (no compute calls here); the test-only introspection hooks live in tests/dev, NOT in the product."""
import ctypes
import os
import re

import pytest

from photo_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in ("gsr.h",):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", src))
    return sorted(names - {"gsr_alloc_fn"})


def test_header_functions_are_listed_in_the_binding():
    assert set(declared_functions()) == set(capi.EXPORTED_SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(capi.HIP_LIB_PATH):
        pytest.skip("libgsr_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(capi.HIP_LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    lib.gsr_backend.restype = ctypes.c_char_p
    assert lib.gsr_backend() == b"hip-gfx950"
    lib.gsr_strerror.restype = ctypes.c_char_p
    assert lib.gsr_strerror(-3) == b"HIP runtime error"


def test_product_library_exports_no_test_hooks():
    if not os.path.exists(capi.HIP_LIB_PATH):
        pytest.skip("libgsr_hip.so not built")
    lib = ctypes.CDLL(capi.HIP_LIB_PATH)
    for n in ("gsr_view_geometry", "gsr_view_binning", "gsr_view_image", "gsr_stage_scan_u32", "gsr_stage_radix_sort_pairs"):
        assert not hasattr(lib, n), n


def test_product_loader_has_no_fallback(tmp_path):
    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        capi.load(str(tmp_path / "libgsr_hip.so"))


def test_scratch_size_queries():
    if not os.path.exists(capi.HIP_LIB_PATH):
        pytest.skip("libgsr_hip.so not built")
    lib = capi.load()
    assert lib.gsr_geometry_bytes(0) > 0 and lib.gsr_geometry_bytes(1000) > lib.gsr_geometry_bytes(10)
    assert lib.gsr_binning_bytes(1000) >= 1000 * 16 and lib.gsr_image_bytes(64, 48) >= 64 * 48 * 8
    assert lib.gsr_image_bytes(0, 10) == 0 and lib.gsr_loss_scratch_bytes(64, 48) >= 9 * 64 * 48 * 4
