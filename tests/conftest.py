import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The f64 CPU oracle (test infrastructure), built on demand with g++."""
    from oracle_sim import load_oracle
    return load_oracle("f64")


@pytest.fixture(scope="session")
def hip_available():
    import torch
    return torch.cuda.is_available()
