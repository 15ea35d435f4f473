"""What CS_CFG_TILE_OVERLAP costs or saves on ONE GPU at configs[2]'s tile size (round 5): a middle tile of the 4 x 2
decomposition's population (125,000 walkers) whose XLO / XHI peers over RCCL are the rank itself (a communicator of one:
real ncclSend / ncclRecv launches of the fixed-capacity halo buffers, no wire), stepped through cs_tile_step_rccl with and
without the flag.  The scene of tests/test_gpu_tiles.py (`middle_tile_125k`).  Not a multi-GPU number: the wire and the
peers' skew are missing; it says whether the split launch + second stream pays for itself before any of that.
    python tools/overlap_ab_one_gpu.py > gpurun_out/overlap_ab_one_gpu.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from rmf_crowdsim_amd import LocationHash2D, Simulation, StubHighLevelPlan, Zanlungo, _abi, scenes
    from rmf_crowdsim_amd.tiles import RECORD, XHI, XLO
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    own = os.environ.get("CS_TOOL_OWN_STREAM") == "1"  # the engine creates its own stream (CS_RESERVE_CUS applies to that one)
    steps = 400
    print("(the phase lines come before the summary line of each pair: plain first, overlapped second)")

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
    # the tile: x rows [10, rows_x) of a grid whose x extent is its HEIGHT (location_hash_2d.rs:59), all of y; the crowd
    # keeps 6 m clear of the XLO / XHI edges (their bands are 4 m deep)
    columns = 560 if n == 125_000 else int(np.ceil(np.sqrt(n)))
    rows = int(np.ceil(n / columns))
    x_extent, y_extent = columns * 0.6325, rows * 0.6325
    tile_x1 = int(np.ceil((26.0 + x_extent + 6.0) / 2.0))
    grid_h = 2.0 * (tile_x1 + 10)
    grid_w = 2.0 * int(np.ceil((10.0 + y_extent + 80.0) / 2.0))

    def run(flags):
        with torch.cuda.stream(side):
            big = dict(width=max(grid_w, 240.0), height=max(grid_h, 420.0), cell_size=2.0, offset=(0.0, 0.0))
            sim = Simulation(LocationHash2D(**big), device=0, stream=None if own else side.cuda_stream,
                             tile=(10, tile_x1, 0, int(big["width"] / 2.0)), halo_cells=1, flags=flags,
                             capacity_hint=n + n // 8)
            cap = max(8192, int(2 * 4.0 * y_extent * 2.5 * 1.5))
            keep = {d: (torch.zeros((cap + 1) * RECORD, dtype=torch.uint8, device="cuda"),
                        torch.zeros((cap + 1) * RECORD, dtype=torch.uint8, device="cuda")) for d in (XLO, XHI)}
            torch.cuda.synchronize()  # (the buffers were zeroed on torch's stream; the engine may have its own)
            for d, (s_, r_) in keep.items():
                sim.halo_set_buffers(d, s_.data_ptr(), r_.data_ptr(), cap)
            sim.rccl_comm_init(1, 0, sim.rccl_unique_id())
            sim.halo_set_peers([0, 0, -1, -1, -1, -1, -1, -1])
            lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
            pts = scenes.jittered_lattice(n, 0.6325, (26.0, 10.0), 0.2, 11, columns=columns)
            k = np.arange(len(pts))
            group = ((k % columns) + (k // columns)) % 2
            for g, vx in ((0, 2.5e-4), (1, -2.5e-4)):
                sim.add_agents(pts[group == g], StubHighLevelPlan((vx, scenes.WALK_SPEED * 0.2)), lp, 2.0)
            for _ in range(100):
                sim.tile_step_rccl(0.05)
            (sim.synchronize() if own else side.synchronize())
            t0 = time.perf_counter()
            for _ in range(steps):
                sim.tile_step_rccl(0.05)
            (sim.synchronize() if own else side.synchronize())
            el = time.perf_counter() - t0
            # the phases, in a pass of their own (event pairs cost the streams a few us each)
            sim.profile_reset()
            sim.profile_stride(2)
            sim.profile_enable((1 << _abi.CS_K_COUNT) - 1)
            for _ in range(40):
                sim.tile_step_rccl(0.05)
            (sim.synchronize() if own else side.synchronize())
            sim.profile_enable(0)
            prof = sim.profile_read()
            print("   phases (us):", {k: round(1e3 * v["total_ms"] / v["launches"], 1) for k, v in prof.items() if v["launches"]}, flush=True)
            for _ in range(steps - 40):
                sim.tile_step_rccl(0.05)
            (sim.synchronize() if own else side.synchronize())
            out = sim.read_agents()
            stats = (sim.kernel_stat(_abi.CS_STAT_EXCHANGES_AHEAD), sim.kernel_stat(_abi.CS_STAT_EXCHANGES_AHEAD_USED))
            del sim
        return out, el / steps * 1e6, stats

    for rep in range(2):
        plain, us_plain, _ = run(0)
        ahead, us_ahead, stats = run(_abi.CS_CFG_TILE_OVERLAP)
        print(f"{n} agents on a middle tile, 2 self-peers over RCCL, {steps} steps: plain {us_plain:.1f} us/step, "
              f"overlapped {us_ahead:.1f} us/step (exchanges ahead / used {stats}), same bits {plain.tobytes() == ahead.tobytes()}",
              flush=True)


if __name__ == "__main__":
    main()
