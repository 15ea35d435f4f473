#!/usr/bin/env python3
"""What one halo exchange costs on ONE GPU: a tile in the middle of the grid sends its buffers to
itself through cs_halo_exchange_rccl (a single-rank communicator).  No wire, so this is the fixed
part of an exchange (the RCCL launch and its send / receive pairs), per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rmf_crowdsim_amd import LocationHash2D, Simulation, StubHighLevelPlan, Zanlungo, scenes  # noqa: E402
from rmf_crowdsim_amd.tiles import RECORD  # noqa: E402

side = torch.cuda.Stream()
grid = dict(width=600.0, height=600.0, cell_size=2.0, offset=(0.0, 0.0))
for ndir, cap in ((2, 18000), (4, 18000), (8, 9000)):
    with torch.cuda.stream(side):
        sim = Simulation(LocationHash2D(**grid), device=0, stream=side.cuda_stream, tile=(100, 200, 100, 200), halo_cells=1)
        bufs = {d: (torch.zeros((cap + 1) * RECORD, dtype=torch.uint8, device="cuda"),
                    torch.zeros((cap + 1) * RECORD, dtype=torch.uint8, device="cuda")) for d in range(ndir)}
        for d, (s_, r_) in bufs.items():
            sim.halo_set_buffers(d, s_.data_ptr(), r_.data_ptr(), cap)
        sim.rccl_comm_init(1, 0, sim.rccl_unique_id())
        sim.halo_set_peers([0 if d < ndir else -1 for d in range(8)])
        for _ in range(20):
            sim.halo_exchange_rccl(-1)
        side.synchronize()
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            sim.halo_exchange_rccl(-1)
        t1 = time.perf_counter()
        side.synchronize()
        t2 = time.perf_counter()
        mb = ndir * (cap + 1) * RECORD / 1e6
        print(f"{ndir} directions, {mb:.2f} MB per exchange: host issue {1e6 * (t1 - t0) / n:.1f} us, total {1e6 * (t2 - t0) / n:.1f} us per exchange")
        del sim
