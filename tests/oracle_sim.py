"""Test-only binding of the CPU oracle (oracle/crowdstep_oracle.cpp) behind the same
Python surface as the product, so a scenario can be replayed on both."""
import ctypes
import os
import subprocess

from rmf_crowdsim_amd import _abi
from rmf_crowdsim_amd.simulation import Simulation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_libs = {}


def load_oracle(kind="f64"):
    if kind in _libs:
        return _libs[kind]
    name = "libcrowdstep_oracle.so" if kind == "f64" else "libcrowdstep_oracle_f32.so"
    target = []
    if kind == "f64" and os.environ.get("CS_ORACLE_SANITIZED") == "1":
        # the ASan + UBSan build (make -C oracle asan): this process must have libasan preloaded
        name, target = "libcrowdstep_oracle_asan.so", ["asan"]
    path = os.path.join(ORACLE_DIR, "_build", name)
    src = os.path.join(ORACLE_DIR, "crowdstep_oracle.cpp")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR] + target, check=True, capture_output=True)
    lib = _abi.bind(ctypes.CDLL(path))
    lib.oracle_time_to_collision.restype = ctypes.c_double
    lib.oracle_time_to_collision.argtypes = [ctypes.c_double] * 5
    lib.oracle_index_add_or_update.restype = ctypes.c_int
    lib.oracle_index_add_or_update.argtypes = [ctypes.c_void_p, ctypes.c_uint64,
                                               ctypes.c_double, ctypes.c_double]
    lib.oracle_index_remove.restype = None
    lib.oracle_index_remove.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    lib.oracle_count_shell_crossings.restype = None
    lib.oracle_count_shell_crossings.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.oracle_shell_crossings.restype = ctypes.c_uint64
    lib.oracle_shell_crossings.argtypes = [ctypes.c_void_p]
    lib.oracle_fast_steps_ex.restype = ctypes.c_double
    lib.oracle_fast_steps_ex.argtypes = ([ctypes.c_uint64] + [ctypes.POINTER(ctypes.c_double)] * 3 +
                                         [ctypes.c_double] * 11 + [ctypes.c_uint32, ctypes.c_int,
                                                                    ctypes.POINTER(ctypes.c_uint8), ctypes.c_int])
    lib.oracle_degenerate_flips.restype = ctypes.c_uint64
    lib.oracle_degenerate_flips.argtypes = [ctypes.c_void_p]
    lib.oracle_spurious_victims.restype = ctypes.c_size_t
    lib.oracle_spurious_victims.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_zanlungo_pair_force.restype = None
    lib.oracle_zanlungo_pair_force.argtypes = [dp, dp, dp, ctypes.c_double, dp]
    lib.oracle_zanlungo_desired_velocity.restype = ctypes.c_double
    lib.oracle_zanlungo_desired_velocity.argtypes = [dp, dp, dp, ctypes.c_uint64, ctypes.c_double,
                                                     ctypes.c_double, dp]
    _libs[kind] = lib
    return lib


class OracleSimulation(Simulation):
    """`Simulation` whose engine is the f64 CPU oracle.  Tests only."""
    _kind = "f64"

    def _load_library(self):
        return load_oracle(self._kind)

    def spurious_victims(self):
        """Ids (a set) of the agents whose t_i came out 0 through a pair whose |rel_vel|^2 underflowed: the
        reference's f64 flaw of DESIGN.md section 5.  Recorded from the first call on: call it once before stepping."""
        import numpy as np
        n = self._lib.oracle_spurious_victims(self._engine, None, 0)
        buf = np.zeros(max(n, 1), dtype=np.uint64)
        self._lib.oracle_spurious_victims(self._engine, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n)
        return set(int(i) for i in buf[:n])

    def degenerate_flips(self):
        """Force terms so far whose sideways direction hung on the sign of a dot product that is zero to rounding (a
        neighbour straight ahead or behind the agent: walkers in file), DESIGN.md section 5.  Counted from the first
        call on: call it once before stepping.  Planners registered later are picked up by the next call."""
        return int(self._lib.oracle_degenerate_flips(self._engine))

    def count_shell_crossings(self, on=True):
        """from now on count, per step, the ordered neighbour pairs whose
        membership would depend on the reference's visiting order."""
        self._lib.oracle_count_shell_crossings(self._engine, 1 if on else 0)

    @property
    def shell_crossings(self):
        return int(self._lib.oracle_shell_crossings(self._engine))


class OracleSimulationF32(OracleSimulation):
    _kind = "f32"


def fast_steps(pts, pref, zanlungo, eyesight, grid, dt, steps, threads=0, vel=None, spurious=None, kind="f64",
               guarded=False, cell_relative=False):
    """The OpenMP / cell-sorted CPU baseline (oracle_fast_steps): same arithmetic as the oracle,
    not the reference's data structures.  Returns (positions, velocities, seconds; negative = an agent left the grid).
    `spurious`: a uint8 array of len(pts) that gets a 1 for every agent whose t_i came out 0 through an underflowed pair
    in some step.  kind = "f32": the f32 build of the same code (state and arithmetic in f32, global coordinates).
    guarded: with Zanlungo::guard_underflow (a pair outside the collision distance with |rel_vel|^2 < 1e-30 never
    collides): the only way a plain f32 reading of the reference survives a long scene with real forces.
    cell_relative: positions kept in f64, every agent's update computed in `kind`'s type on positions relative to its own
    cell (the precision class of the HIP engine's layout, implemented independently: oracle_fast_steps_ex, flag 2)."""
    import numpy as np
    lib = load_oracle(kind)
    xy = np.ascontiguousarray(pts, dtype=np.float64).copy()
    v = np.zeros_like(xy) if vel is None else np.ascontiguousarray(vel, dtype=np.float64).copy()
    pv = np.ascontiguousarray(pref, dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    A, _, _, D, m, R = zanlungo
    sec = lib.oracle_fast_steps_ex(len(xy), xy.ctypes.data_as(dp), v.ctypes.data_as(dp), pv.ctypes.data_as(dp),
                                   A, D, m, R, eyesight, grid["width"], grid["height"], grid["cell_size"],
                                   grid["offset"][0], grid["offset"][1], dt, steps, threads,
                                   None if spurious is None else spurious.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                   (1 if guarded else 0) | (2 if cell_relative else 0))
    return xy, v, sec
