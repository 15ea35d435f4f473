import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_tiles as T
from oracle_sim import OracleSimulation
from rmf_crowdsim_amd import CrowdSimError, LocationHash2D, Simulation
seed = int(sys.argv[1]); stop = int(sys.argv[2])
grid = dict(width=80.0, height=80.0, cell_size=float([1.0, 2.0, 2.5][seed % 3]), offset=(0.0, 0.0))
sims = [Simulation(LocationHash2D(**grid)), OracleSimulation(LocationHash2D(**grid))]
for s in sims: T._random_sink_scene(s, 900 + seed)
np.set_printoptions(precision=17, linewidth=200)
for k in range(stop + 1):
    outs = []
    for s in sims:
        try:
            s.step(0.1); outs.append((s.read_agents(), dict(s.last_report)))
        except CrowdSimError as e:
            outs.append((None, str(e)))
    (a, ra), (b, rb) = outs
    if a is None or b is None or ra["n_tti_zero"] != rb["n_tti_zero"] or ra["n_nonfinite"] != rb["n_nonfinite"] or k >= stop - 2:
        print("step", k, "engine", ra if a is None else {q: ra[q] for q in ("n_agents","n_tti_zero","n_nonfinite","n_clamped")}, "oracle", rb if b is None else {q: rb[q] for q in ("n_agents","n_tti_zero","n_nonfinite","n_clamped")})
        if a is not None and b is not None:
            d = np.hypot(a["x"] - b["x"], a["y"] - b["y"]); dv = np.hypot(a["vx"] - b["vx"], a["vy"] - b["vy"])
            j = np.argsort(-dv)[:4]
            print("  largest |dv|:", [(int(a["id"][i]), float(dv[i]), float(d[i]), (float(a["x"][i]), float(a["y"][i])), (float(a["vx"][i]), float(a["vy"][i])), (float(b["vx"][i]), float(b["vy"][i]))) for i in j])
        if a is None or b is None: break
