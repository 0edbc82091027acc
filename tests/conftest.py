import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

entry.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device); run with -m gpu")


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
