"""Locate, build and load the HIP engine (libcrowdstep_hip.so).

The product has exactly one compute backend: the HIP shared library built from
csrc/.  There is no CPU fallback; a missing or unloadable library is an error.
"""
import ctypes
import os
import shutil
import subprocess

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcrowdstep_hip.so")
SOURCES = ["crowdstep_hip.hip"]
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # IEEE behaviour the Zanlungo NaN/inf semantics rely on (DESIGN.md "Numerics")
    "-ffp-contract=off", "-fno-fast-math",
    # measured on the step kernel (3 runs each, 1M agents): 168.9 -> 164.4 us; scheduling only,
    # no effect on the arithmetic
    "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-fno-unroll-loops",
    # no SLP vectorisation: it packs adjacent f32 operations into v_pk_mul / v_pk_fma / v_pk_add, which issue at
    # half the rate of the plain forms on MI355X (tools/valu_ceiling.hip: 2 FMAs per 4.7 clocks against 2.3 each)
    # and cost registers: the step kernel 145.9 -> 141.1 us, 128 -> 119 VGPRs, scratch 32 -> 0 B (round 3)
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function", "-Wno-unused-value",
]

_lib = None


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP engine cannot be built")


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(REPO_ROOT, "include", "crowdstep.h"))
    return any(os.path.getmtime(d) > built for d in deps if os.path.isfile(d))


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into lib/libcrowdstep_hip.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = os.environ.get("CS_HIPCC_EXTRA", "").split()
    cmd = [_hipcc()] + HIPCC_FLAGS + extra + ["-I", os.path.join(REPO_ROOT, "include"), "-o", LIB_PATH]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    if verbose and proc.stderr:
        print(proc.stderr)
    return LIB_PATH


def load():
    """Load and bind the HIP engine.  Raises if it is missing: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    # development only (tools/variants_bench.sh): time another build of the same engine
    lib_path = os.environ.get("CS_LIB_PATH") or LIB_PATH
    if not os.path.exists(lib_path):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). rmf_crowdsim_amd has no CPU fallback.")
    # One process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64 (same
    # SONAME as /opt/rocm's); whichever loads first serves both.  torch only works on its own
    # copy, so when torch is installed it goes first (bench.py and the tile transports use torch
    # streams / torch.distributed next to the engine).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = _abi.bind(ctypes.CDLL(lib_path))
    return _lib
