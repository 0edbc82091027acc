"""host/include/compat/optimizer_key.h (last row): the Adam state key the reference's map-maintenance code needs
on a current LibTorch -- a consumer that builds torch::optim::Adam, steps it and finds the parameter's state through
optim_key(), compiled against the installed LibTorch."""
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_optim_key_finds_the_adam_state(tmp_path):
    base = os.path.dirname(torch.__file__)
    exe = str(tmp_path / "optimizer_key_consumer")
    subprocess.check_call(["g++", "-std=c++17", "-O0", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w",
                           "-I" + os.path.join(ROOT, "photo-slam_amd", "host", "include"), "-I" + os.path.join(base, "include"),
                           "-I" + os.path.join(base, "include", "torch", "csrc", "api", "include"),
                           os.path.join(ROOT, "tests", "compat", "optimizer_key_consumer.cpp"), "-o", exe,
                           "-L" + os.path.join(base, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-Wl,-rpath," + os.path.join(base, "lib")])
    assert subprocess.call([exe]) == 0
