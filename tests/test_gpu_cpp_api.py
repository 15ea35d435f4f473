"""Builds and runs the C++ restatement of the reference's step tests (tests/cpp) against the
HIP engine through include/crowdsim.hpp."""
import os
import subprocess

import pytest

from rmf_crowdsim_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_reference_api")


def build_cpp_test(name="test_reference_api"):
    lib = _native.build()
    exe = os.path.join(os.path.dirname(EXE), name)
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                        "-L", os.path.dirname(lib), "-lcrowdstep_hip",
                        "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_cpp_mirror_compiles():
    build_cpp_test()
    build_cpp_test("test_mesh_api")


@pytest.mark.gpu
def test_mesh_api_in_cpp():
    """A 2 x 2 in-process mesh through cs_mesh_* from C++ = one engine, bit for bit (tests/cpp/test_mesh_api.cpp)."""
    out = subprocess.run([build_cpp_test("test_mesh_api")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "mesh api: passed" in out.stdout


@pytest.mark.gpu
def test_reference_step_tests_in_cpp():
    exe = build_cpp_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "8 passed" in out.stdout
