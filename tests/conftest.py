import os
import sys

# the end-to-end tests alternate between two OpenMP consumers (the CPU oracle and ATen): with spinning worker threads each
# starves the other on a many-core host (the GPU boxes have 256 hardware threads: minutes instead of seconds) -- sleep instead
# (must be set before libgomp loads; bench.py does the same for its CPU baseline leg)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

entry.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device); run with -m gpu")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests are skipped where no HIP device exists (a plain `pytest tests/` on a CPU box used to reach
    load_host('hip') after the emulator host: a second TORCH_LIBRARY registration aborts the whole process)."""
    if not any("gpu" in item.keywords for item in items):
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def emu_lib_path():
    """tests/emu/libgsr_emu.so: the kernel sources compiled for the host wave64 emulator."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    return build_emu.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
