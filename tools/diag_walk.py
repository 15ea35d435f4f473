import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["CS_HIPCC_EXTRA"] = "-DCS_PHASE_CLOCKS"
from rmf_crowdsim_amd import _native
_native.build(force=True)
try:
    import numpy as np
    from rmf_crowdsim_amd import scenes, _abi
    from rmf_crowdsim_amd.simulation import Simulation, Zanlungo, LocationHash2D
    n = 125000
    room = scenes.WALK_SPEED * 0.05 * 120 + 4.0
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, room=room)
    sim = Simulation(LocationHash2D(**grid))
    scenes.add_walking_crowd(sim, pts, group, Zanlungo(*scenes.METRIC_ZANLUNGO), 2.0)
    fn = sim._lib.cs_debug_phase_cycles
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]; fn.restype = None
    out = (C.c_ulonglong * 12)()
    for _ in range(5): sim.step(0.05, report=False)
    fn(sim._engine, out, 1)
    sim.profile_reset(); sim.profile_stride(1); sim.profile_enable(1 << _abi.CS_K_NEIGHBOUR_FORCE)
    for k in range(70):
        sim.profile_reset()
        sim.step(0.05, report=False)
        sim.synchronize()
        prof = sim.profile_read()
        fn(sim._engine, out, 1)
        print(k, "k4 us", [round(1e3 * v["total_ms"] / v["launches"], 1) for v in prof.values() if v["launches"]], "offpath/chunk", out[11], "total cyc", sum(out[:11]))
finally:
    del os.environ["CS_HIPCC_EXTRA"]
    _native.build(force=True)
