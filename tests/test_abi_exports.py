"""The C-ABI library loads on a CPU-only box and exports every symbol include/crowdstep.h
declares (no compute calls: the engine has no CPU path)."""
import ctypes
import os
import re

import pytest

from rmf_crowdsim_amd import _abi, _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "crowdstep.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z_0-9]+)\s*\(", text)) - {"cs_hlp_velocity_fn"})


def test_bindings_cover_the_header():
    declared = [s for s in _declared_symbols() if not s.endswith("_fn")]
    assert sorted(_abi.SYMBOLS) == declared


def test_hip_library_builds_and_exports_every_symbol():
    path = _native.build()
    lib = ctypes.CDLL(path)
    for name in _abi.SYMBOLS:
        assert hasattr(lib, name), name
    assert lib.cs_abi_version() == _abi.CS_ABI_VERSION


def test_oracle_exports_the_same_abi(oracle_lib):
    for name in _abi.SYMBOLS:
        assert hasattr(oracle_lib, name), name


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rmf_crowdsim_amd import CrowdSimError, LocationHash2D, Simulation
    with pytest.raises(CrowdSimError, match="no HIP device visible; the engine has no CPU fallback"):
        Simulation(LocationHash2D(10.0, 10.0, 1.0, (0.0, 0.0)))
