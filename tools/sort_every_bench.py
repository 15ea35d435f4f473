"""Round 5, the one structural K4 experiment: what would a sort kept for k steps save at most?

CS_DEBUG_SORT_EVERY=k (measurement only, cs_engine::rebuild) skips the scan and the scatter + window builder on k - 1 of k
steps.  That is correct only while no agent changes its cell: the creep scene slowed to 1e-9 m/s (nobody of a million
comes within the 1.5e-8 m it travels of a cell boundary; at the bench's 2.5e-4 m/s thousands do, and the lever then
steps garbage); the kernels' cost does not depend on the speed scale.  The bits must equal the plain run's, or this
script fails.  The time saved is the UPPER BOUND of the sort's share of what a
Verlet-style scheme (sort and candidate lists with a skin, reused for k steps) could gain; the filter's share is bounded
by the ablation of profiles/K4_LEVERS.md (the candidates' part of the kernel).  Run on the GPU box:
    python tools/sort_every_bench.py > gpurun_out/sort_every.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(k, agents, steps, workload="creep"):
    import torch
    import bench
    from rmf_crowdsim_amd import Simulation, _abi, scenes
    if k:
        os.environ["CS_DEBUG_SORT_EVERY"] = str(k)
    else:
        os.environ.pop("CS_DEBUG_SORT_EVERY", None)
    speed = 1e-9
    sim, grid, extent = bench.build_crowd(Simulation, agents, 2.0, 2.0, speed, workload=workload, steps=steps + 100,
                                          capacity=agents + 1024)
    for _ in range(80):
        sim.step(0.05, report=False)
    sim.synchronize()
    sim.profile_reset()
    sim.profile_stride(4)
    sim.profile_enable((1 << _abi.CS_K_NEIGHBOUR_FORCE) | (1 << _abi.CS_K_SCAN) | (1 << _abi.CS_K_SCATTER))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step(0.05, report=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sim.profile_enable(0)
    prof = sim.profile_read()
    sim.step(0.05)
    assert sim.last_report["n_agents"] == agents and sim.last_report["n_nonfinite"] == 0
    out = sim.read_agents()
    us = {name: (1e3 * prof[name]["total_ms"] / prof[name]["launches"] if prof[name]["launches"] else 0.0)
          for name in ("neighbour_force", "scan", "scatter")}
    return out, el / steps * 1e6, us


def main():
    agents = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    steps = 200
    base, base_us, base_k = run(0, agents, steps)
    print(f"{agents} agents, creep scene, {steps} timed steps")
    print(f"  sort every step : {base_us:7.1f} us/step   K4 {base_k['neighbour_force']:.1f}  scan {base_k['scan']:.1f}  scatter+builder {base_k['scatter']:.1f}")
    for k in (2, 4, 8):
        out, us, ks = run(k, agents, steps)
        same = out.tobytes() == base.tobytes()
        print(f"  sort every {k}th   : {us:7.1f} us/step   K4 {ks['neighbour_force']:.1f}   saved {base_us - us:5.1f} us = "
              f"{100 * (base_us - us) / base_us:4.1f} %   bits equal: {same}")
        assert same, "CS_DEBUG_SORT_EVERY changed the result: an agent changed its cell, or the lever is wrong"


if __name__ == "__main__":
    main()
