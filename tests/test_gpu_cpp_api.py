"""Builds and runs the C++ restatement of the reference's step tests (tests/cpp) against the
HIP engine through include/crowdsim.hpp."""
import os
import subprocess

import pytest

from rmf_crowdsim_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_reference_api")


def build_cpp_test():
    lib = _native.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_reference_api.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                        "-L", os.path.dirname(lib), "-lcrowdstep_hip",
                        "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def test_cpp_mirror_compiles():
    build_cpp_test()


@pytest.mark.gpu
def test_reference_step_tests_in_cpp():
    exe = build_cpp_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "6 passed" in out.stdout
