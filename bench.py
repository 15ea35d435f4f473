#!/usr/bin/env python3
"""bench.py — agent-steps/sec of the MI355X crowd-step engine (BASELINE.json metric).

A "step" is one Simulation::step (lib.rs:195-383 of the reference) over the whole synthetic
crowd: cell re-sort (scan + scatter) + the Zanlungo neighbour kernel + integration.  State is
resident in HBM before the timed region; nothing is read back inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--agents A] [--eyesight E] [--cell C]

N > 1: one rank per GPU.  Either under a launcher (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`, what the driver does) or as plain `python bench.py --gpus N`, which starts that
launcher itself as a child process before anything touches the GPU.  The default is STRONG scaling:
--agents (1M) agents in all, one spatial tile per rank (BASELINE.json configs[2]); the weak-scaled figure
(N x --agents) is timed after it and reported under `weak_scaled`.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# Algorithmic HBM bytes of the neighbour kernel per launch, as SURVEY.md section 8(d) counts them: read
# 32 B/agent (pos 8 + vel 8 + own pref 8 + id 4 + eyesight/group 4) + 8 B per cell (start, count);
# write 16 B/agent (new pos 8 + new vel 8).  (What the fused kernel really moves, without
# neighbour re-reads, is 60 B/agent + 4 B/cell: it also carries id / meta / cell / rank through
# for the re-sort; `roofline.fused_kernel_bytes_per_launch` reports that figure.)
K4_READ_BYTES = 32
K4_WRITE_BYTES = 16
K4_CELL_BYTES = 8
K4_FUSED_BYTES_PER_AGENT, K4_FUSED_BYTES_PER_CELL = 60, 4
N_SIMDS, SIMD_CLOCK_HZ = 1024, 2.4e9
VALU_ISSUE_PEAK = N_SIMDS * SIMD_CLOCK_HZ / 2.0  # wave64 VALU instructions per second: 1024 SIMDs, one per 2 clocks


def profile_key(workload_desc):
    """Identifies what a cached PMC summary (profiles/rNN/k4_traffic.json) was measured on: the
    kernel sources (device code and the engine that configures and launches it; not the orchestration above the
    engine: the mesh and the RCCL binding), the compiler flags and the workload.  A summary with another key is stale."""
    import hashlib
    from rmf_crowdsim_amd import _native
    h = hashlib.sha256()
    for name in sorted(os.listdir(_native.CSRC)):
        if name in ("cs_mesh.hip.inc", "cs_rccl.hip.inc"):
            continue
        with open(os.path.join(_native.CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    h.update(" ".join(_native.HIPCC_FLAGS).encode())
    h.update(workload_desc.encode())
    return h.hexdigest()[:16]


def walk_room(steps):
    """Free metres on the high-x side for a crowd that walks +x at 1.3 m/s for `steps` steps."""
    from rmf_crowdsim_amd import scenes
    return scenes.WALK_SPEED * 0.05 * (steps + 8) + 4.0


def populate(target, workload, pts, group, speed, lp, eyesight):
    from rmf_crowdsim_amd import scenes
    if workload == "walk":
        scenes.add_walking_crowd(target, pts, group, lp, eyesight, creep=speed)
    else:
        scenes.add_counterflow(target, pts, group, speed, lp, eyesight)


def build_crowd(sim_cls, n, cell, eyesight, speed, workload="walk", steps=200, device=0, stream=None, capacity=0):
    from rmf_crowdsim_amd import LocationHash2D, Zanlungo, scenes
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=cell,
                                                    room=walk_room(steps) if workload == "walk" else 0.0)
    kwargs = {}
    if sim_cls.__name__ == "Simulation":
        kwargs = dict(device=device, stream=stream, capacity_hint=capacity)
    sim = sim_cls(LocationHash2D(**grid), **kwargs)
    populate(sim, workload, pts, group, speed, Zanlungo(*scenes.METRIC_ZANLUNGO), eyesight)
    return sim, grid, extent


def cpu_baseline(agents, cell, eyesight, speed, workload="walk", budget_s=15.0):
    """Times the CPU oracle (the reference-shaped single-thread port) on a bounded sample of the
    same workload: same density / parameters, fewer agents, a few steps."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_sim import OracleSimulation
    n = min(agents, 1_000_000)  # (round 5: at the metric's own 1M agents; until then a 100k sample was extrapolated)
    sim, _, _ = build_crowd(OracleSimulation, n, cell, eyesight, speed, workload=workload, steps=110)
    sim.step(0.05)  # first step: all velocities 0 -> no forces; not representative
    steps, t0 = 0, time.perf_counter()
    while True:
        sim.step(0.05)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 100:
            break
    return {
        "value": n * steps / el, "unit": "agent-steps/s", "cores": 1, "kind": "port",
        "sample": f"{n} agents x {steps} steps of the same scene (density, planner, eyesight, dt) after one untimed step, "
                  f"oracle/crowdstep_oracle.cpp (the reference's data structures: hash maps, per-cell sets) f64 single "
                  f"thread, {el:.1f} s",
    }


def cpu_baseline_openmp(agents, cell, eyesight, speed, workload="walk", budget_s=8.0):
    """What a good CPU does with the same arithmetic (not the reference's shape): the oracle's
    Zanlungo on cell-sorted arrays, the agent loop spread over the host cores with OpenMP
    (oracle_fast_steps; bit-identical to the oracle, tests/test_oracle_reference_kats.py)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_sim import fast_steps
    from rmf_crowdsim_amd import scenes
    n = min(agents, 1_000_000)
    threads = min(16, os.cpu_count() or 1)  # the GPU box's CPU share
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=cell,
                                                    room=walk_room(60) if workload == "walk" else 0.0)
    pref = np.zeros((n, 2))
    pref[:, 1] = np.where(group == 0, speed, -speed)
    if workload == "walk":
        pref[:, 0] = scenes.WALK_SPEED
    xy, vel, _ = fast_steps(pts, pref, scenes.METRIC_ZANLUNGO, eyesight, grid, 0.05, 1, threads=threads)
    xy, vel, probe = fast_steps(xy, pref, scenes.METRIC_ZANLUNGO, eyesight, grid, 0.05, 1, threads=threads, vel=vel)
    steps = int(max(1, min(50, budget_s / max(probe, 1e-3))))
    xy, vel, sec = fast_steps(xy, pref, scenes.METRIC_ZANLUNGO, eyesight, grid, 0.05, steps, threads=threads, vel=vel)
    return {
        "value": n * steps / sec, "unit": "agent-steps/s", "cores": threads, "kind": "port, cell-sorted + OpenMP",
        "sample": f"{n} agents x {steps} steps of the same scene, the oracle's f64 arithmetic on cell-sorted "
                  f"arrays, {threads} threads, {sec:.1f} s",
    }


def cpu_baselines(args, per_gpu, speed):
    """The two CPU figures of a line: the reference-shaped single-thread port (`cpu_baseline`) and, for the uniform
    crowds, the same arithmetic on cell-sorted arrays over the host's cores (`cpu_baseline_openmp`)."""
    uniform_kind = args.workload in ("walk", "creep")
    wl = args.workload if uniform_kind else "creep"
    out = {"cpu_baseline": cpu_baseline(per_gpu, args.cell, args.eyesight, speed, workload=wl)}
    if uniform_kind:
        # a second, stronger CPU number (not the reference's shape), for orientation
        out["cpu_baseline_openmp"] = cpu_baseline_openmp(per_gpu, args.cell, args.eyesight, speed, workload=wl)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--agents", type=int, default=1_000_000,
                    help="agents in all (strong scaling, the default) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default): --agents agents IN ALL, cut into one tile per rank (the metric's "
                         "configuration, BASELINE.json configs[2]: 1M agents, 4 x 2 tiles on 8 GPUs); weak: every rank "
                         "adds --agents agents to one crowd.  With N > 1 the other mode is timed as well and reported "
                         "under `weak_scaled` / `strong_scaled`")
    ap.add_argument("--watchdog", type=float, default=420.0,
                    help="N > 1: seconds a rank may spend in one phase before it reports where it is stuck and exits "
                         "(0 = off)")
    ap.add_argument("--no-second-scaling-leg", action="store_true",
                    help="N > 1: skip the second run in the other scaling mode")
    ap.add_argument("--eyesight", type=float, default=2.0)
    ap.add_argument("--cell", type=float, default=2.0)
    ap.add_argument("--speed", type=float, default=None, help="default: scenes.CREEP_SPEED")
    ap.add_argument("--kernel", choices=["auto", "tiled", "gather"], default="auto")
    ap.add_argument("--workload", choices=["walk", "creep", "uniform", "stream", "hotspots", "random"], default="walk",
                    help="walk (default): the uniform crowd of BASELINE configs[1..2] walking at 1.3 m/s with the "
                         "creeping counter-flow on top (agents change cells at the real rate); creep (= uniform): "
                         "the same crowd standing but for the counter-flow (non-zero forces); stream: configs[3], "
                         "agents fed by source-sinks; hotspots: configs[4]; random: a thinned lattice")
    ap.add_argument("--no-creep-leg", action="store_true",
                    help="skip the short run of the creep scene whose kernel time is reported beside the default one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clock-warmup", type=int, default=60,
                    help="the device gets at least this many untimed steps of the same scene before the timed region "
                         "(the W warm-up steps count towards it): after a handful of steps the clocks have not settled "
                         "and the first timed steps run 7-10 %% slow (20 timed steps after 5: 5.3e9, after 100: 5.8e9); "
                         "reported as config.clock_warmup_steps")
    ap.add_argument("--profile-stride", type=int, default=4,
                    help="hipEvent pair around the neighbour kernel at every N-th timed step (a pair costs the stream "
                         "a few us: at small crowds use a larger stride)")
    ap.add_argument("--overlap", action="store_true",
                    help="--gpus > 1: the headline leg steps WITH CS_CFG_TILE_OVERLAP (the next step's halo exchange runs "
                         "behind the border windows' launch on a second stream while the interior windows are stepped).  "
                         "Not the default: on one GPU with RCCL self-peers the split launch costs more than the exchange it "
                         "hides (round 5, profiles/r05/overlap_ab_one_gpu.txt); whichever setting is the headline, the OTHER "
                         "one is timed beside it for 20 steps (config.overlap_ab), so one multi-GPU run measures both")
    ap.add_argument("--no-overlap", action="store_true", help="(the default; kept for old commands)")
    ap.add_argument("--no-overlap-ab", action="store_true", help="--gpus > 1: skip the short run with the other overlap setting")
    ap.add_argument("--mesh", choices=["native", "python"], default="native",
                    help="--gpus > 1: what steps the tiles.  native (default): the C ABI's own mesh, cs_mesh_* "
                         "(csrc/cs_mesh.hip.inc; what a Rust or C++ host binds: one cs_mesh_step per step, halo records "
                         "over RCCL from the engine; under CS_BENCH_BACKEND=gloo over a host transport).  python: "
                         "tiles.DistributedTiles, the Python orchestration of the same tile engines, for comparison")
    ap.add_argument("--verify", action="store_true",
                    help="--gpus > 1 with --mesh native: after the timed region rank 0 steps a single engine through the "
                         "same scene and the whole crowd of the mesh must equal it bit for bit (exit code 6 otherwise)")
    ap.add_argument("--debug", type=int, default=0, help="kernel ablation bits (profiling only)")
    ap.add_argument("--planner", choices=["stub", "route"], default="stub",
                    help="stream workload: constant-velocity stub planners (the reference tests' kind) or "
                         "device route followers (CS_HLP_ROUTE, straight two-point routes)")
    ap.add_argument("--readback", action="store_true",
                    help="stream a snapshot of all agents to pinned host memory every step (the "
                         "PCIe-inclusive rate; not the headline value)")
    args = ap.parse_args()
    launch_ranks_if_needed(args)  # --gpus N without a launcher: start the N ranks as a child and exit with its code

    import torch
    import torch.distributed as dist
    from rmf_crowdsim_amd import _abi

    if args.workload == "uniform":
        args.workload = "creep"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("CS_BENCH_BACKEND", "nccl")  # "gloo": ranks sharing one GPU (tests)
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU, launched as "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` "
                         f"(or plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev and not os.environ.get("CS_BENCH_SHARE_DEVICE"):  # (the variable: a test of the
        # loud-failure path: RCCL itself then refuses the second rank of a device)
        raise SystemExit(f"bench: {world} ranks but {n_dev} visible GPU(s): RCCL needs one device per rank "
                         f"(CS_BENCH_BACKEND=gloo lets ranks share a device, for functional tests only)")
    device = local_rank % n_dev if (backend != "nccl" or os.environ.get("CS_BENCH_SHARE_DEVICE")) else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device))
            else:
                dist.init_process_group(backend)
        except Exception as err:  # noqa: BLE001
            print(f"bench: rank {rank} FAILED in phase 'process group init ({backend})': {err}", file=sys.stderr, flush=True)
            os._exit(5)
    if args.scaling is None:
        # the metric is quoted at 1M agents IN ALL (BASELINE.json configs[2]: 4 x 2 tiles of one 1M crowd)
        args.scaling = "strong"
    ctx = dict(torch=torch, dist=dist, rank=rank, world=world, device=device, backend=backend)

    progress = {"phase": "start", "line": None}

    def fail(phase, err):
        """A rank that cannot go on says in which phase and leaves at once with a non-zero code (no unwinding through
        collectives its peers are not in; the launcher ends the other ranks).  If a complete line already stands (the
        short leg without the overlap, measured before the overlapped headline leg), rank 0 prints it first: code 4."""
        print(f"bench: rank {rank} FAILED in phase '{phase}': {err}", file=sys.stderr, flush=True)
        if progress["line"] is not None:
            if rank == 0:
                print(progress["line"], flush=True)
            os._exit(4)
        os._exit(5)
    ctx["fail"] = fail
    if world > 1 and rank == 0:
        # the launcher ends the other ranks with SIGTERM when one of them fails: rank 0 still says what it has
        import signal

        def on_term(_sig, _frame):
            if progress["line"] is not None:
                print(progress["line"], flush=True)
                os._exit(4)
            os._exit(5)
        signal.signal(signal.SIGTERM, on_term)

    # N > 1: the first run of this code over RCCL with real peers may be the driver's.  A rank that sits in one
    # phase for --watchdog seconds says where and leaves; if the headline leg is already measured (the stall is in
    # the optional second leg), rank 0 prints its line first, so the measurement is not lost with the extra.
    progress["phase"] = "headline leg (" + args.scaling + " scaling)"
    watchdog = None
    if world > 1 and args.watchdog > 0:
        import threading

        def expired():
            print(f"bench: rank {rank} made no progress for {args.watchdog:.0f} s in: {progress['phase']}", file=sys.stderr,
                  flush=True)
            if rank == 0 and progress["line"] is not None:
                print(progress["line"], flush=True)
            # 3: nothing measured; 4: the headline line stands (printed above or before), an extra leg or the shutdown hung
            os._exit(4 if (progress["line"] is not None or progress.get("printed")) else 3)

        def arm(phase, seconds=None):
            nonlocal watchdog
            if watchdog is not None:
                watchdog.cancel()
            progress["phase"] = phase
            watchdog = threading.Timer(min(args.watchdog, seconds or args.watchdog), expired)
            watchdog.daemon = True
            watchdog.start()
        arm(progress["phase"])
    else:
        def arm(phase, seconds=None):
            progress["phase"] = phase
    ctx["arm"] = arm

    args.overlap = world > 1 and args.overlap and not args.no_overlap
    ab = None

    def overlap_ab_leg(headline):
        arm("A/B leg (the other overlap setting)", 180.0)
        return run_leg(args, ctx, args.scaling, min(args.steps, 20), min(args.warmup, 10), 30, headline=headline,
                       overlap=not args.overlap)
    if world > 1 and not args.no_overlap_ab and args.overlap:
        # The overlapped schedule has never run with real peers: when it is asked for as the headline, the short leg
        # WITHOUT it runs first and its complete line stands as the fallback (a headline leg that fails or hangs then costs
        # the A/B, not the measurement: exit code 4, "fallback" in the line).
        ab = overlap_ab_leg(True)
        line = dict(ab["line"]) if rank == 0 else {}
        line["fallback"] = "the overlapped headline leg did not finish: this is the short leg WITHOUT the overlap that ran before it"
        progress["line"] = json.dumps(line)
        arm("headline leg (" + args.scaling + " scaling)")
    main_leg = run_leg(args, ctx, args.scaling, args.steps, args.warmup, args.clock_warmup, headline=True)
    other_leg = None
    if world > 1 and not args.no_second_scaling_leg:
        line = dict(main_leg["line"]) if rank == 0 else {}
        line["second_scaling_leg"] = "did not finish"
        progress["line"] = json.dumps(line)  # (on every rank: they all leave with 0 once the headline stands)
        arm("second leg (the other scaling mode)")
        # the other scaling mode beside the headline (weak: N x --agents agents in one crowd)
        other = "weak" if args.scaling == "strong" else "strong"
        leg = run_leg(args, ctx, other, min(args.steps, 100), min(args.warmup, 10), 30, headline=False)
        other_leg = {"scaling": other, "value": leg["value"], "ms_per_step": leg["ms_per_step"],
                     "agents_total": leg["total_agents"], "agents_per_gpu": leg["per_gpu"], "steps": leg["steps"],
                     "kernel_ms": leg["k4_ms"], **leg["tile_report"]}

    if world > 1 and not args.no_overlap_ab and not args.overlap:
        # the default: the headline stands; the same crowd for 20 steps under the overlapped schedule beside it
        # (LAST of the legs, under a shorter watchdog: it is the one schedule that has never met real peers)
        line = dict(main_leg["line"]) if rank == 0 else {}
        if other_leg:
            line[other_leg["scaling"] + "_scaled"] = other_leg
        line["overlap_ab"] = "did not finish"
        progress["line"] = json.dumps(line)  # (on every rank: they all leave with 0 once the headline stands)
        ab = overlap_ab_leg(False)
    if ab is not None and rank == 0:
        mine = {"ms_per_step": main_leg["ms_per_step"], "value": main_leg["value"], "steps": main_leg["steps"],
                "phase_us_max_over_ranks": main_leg["tile_report"].get("phase_us_max_over_ranks"),
                "exchanges_ahead_used": main_leg["tile_report"].get("exchanges_ahead_used")}
        theirs = {"ms_per_step": ab["ms_per_step"], "value": ab["value"], "steps": ab["steps"],
                  "phase_us_max_over_ranks": ab["tile_report"].get("phase_us_max_over_ranks"),
                  "exchanges_ahead_used": ab["tile_report"].get("exchanges_ahead_used")}
        main_leg["line"]["config"]["overlap_ab"] = {
            "overlap": mine if args.overlap else theirs, "no_overlap": theirs if args.overlap else mine,
            "headline_is": "overlap" if args.overlap else "no_overlap",
            "note": "the headline leg's figures beside a short run (its own mesh, same crowd and scaling) with the other "
                    "setting" + (", which ran first" if args.overlap else ", which ran after it")}
    if rank == 0:
        out = main_leg["line"]
        if other_leg:
            out[other_leg["scaling"] + "_scaled"] = other_leg
        if world > 1 and not args.no_cpu_baseline:
            # N > 1 lines carry the CPU baseline too (review of round 4): rank 0 times the same bounded samples as at
            # N = 1 while its peers wait at the barrier below (no GPU work is pending; a failure here loses the baseline,
            # never the line)
            progress["line"] = json.dumps(out)
            arm("CPU baseline on rank 0 (the measurement is complete)")
            try:
                out.update(cpu_baselines(args, main_leg["total_agents"], main_leg["speed"]))  # (the whole crowd: the metric's 1M agents)
            except Exception as err:  # noqa: BLE001
                out["cpu_baseline"] = {"error": str(err)}
        print(json.dumps(out), flush=True)
    progress["line"], progress["printed"] = None, True
    arm("shutdown")
    verdicts = [main_leg["tile_report"].get("verify")] + ([other_leg.get("verify")] if other_leg else [])
    bad_verify = any(isinstance(v, dict) and not v.get("mesh_equals_single_engine", True) for v in verdicts)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if bad_verify:
        raise SystemExit(6)


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: the N ranks are started as ONE
    child (`python -m torch.distributed.run`, one rank per GPU) before this process has imported torch
    or touched the GPU, and this process exits with the child's code.  Under a launcher (WORLD_SIZE set)
    this is a no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


class TileCounts:
    """Agents per tile of a set of positions under the cuts a mesh of this shape gets (the library cuts the same way:
    tiles.TileLayout mirrors cs_mesh_create's mesh_even_edges / mesh_weighted_edges; tests/test_native_mesh.py)."""

    def __init__(self, spatial_index, tiling, weights, halo):
        from rmf_crowdsim_amd.tiles import TileLayout
        self.index = spatial_index
        self.layout = TileLayout(spatial_index, *tiling, weights=weights, min_cells=2 * int(halo))

    def of(self, pts):
        return self.layout.tile_counts(pts, self.index)


def make_mesh(args, ctx, spatial_index, tiling, halo, density, capacity, flags, weights):
    """N > 1: this rank's tile of the crowd.  Returns (what is stepped, the tile's engine for profiling, report)."""
    torch, dist = ctx["torch"], ctx["dist"]
    rank, world, device, backend = ctx["rank"], ctx["world"], ctx["device"], ctx["backend"]
    if args.mesh == "python":
        from rmf_crowdsim_amd.tiles import DistributedTiles
        stepper = DistributedTiles(spatial_index, tiling, halo, device, density_per_cell=density, capacity_hint=capacity,
                                   flags=flags, weights=weights)
        return stepper, stepper.sim, {"mesh": "python", "transport": stepper.transport, "entry_point": "tiles.DistributedTiles.step"}
    import ctypes as C
    from rmf_crowdsim_amd import _abi, _native
    from rmf_crowdsim_amd.tiles import NativeTileMesh, TorchHostTransport
    if os.environ.get("CS_BENCH_BREAK_NATIVE_MESH"):  # (test hook: the fallback when the ranks cannot set their tiles up)
        raise RuntimeError("native mesh creation failed on this rank (CS_BENCH_BREAK_NATIVE_MESH)")
    if backend == "nccl":
        # rank 0 makes the communicator's id, the launcher's process group hands it round; from then on every byte of
        # the step travels over RCCL from the engine (what a Rust host does with MPI or a file in torch's place)
        box = [None]
        if rank == 0:
            raw = (C.c_uint8 * _abi.CS_RCCL_UNIQUE_ID_BYTES)()
            if _native.load().cs_rccl_unique_id(raw) != 0:
                box = [RuntimeError("cs_rccl_unique_id failed: librccl could not be bound")]
            else:
                box = [bytes(raw)]
        dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], Exception):
            raise box[0]
        stepper = NativeTileMesh(spatial_index, tiling, halo, device=device, density_per_cell=density, flags=flags,
                                 weights=weights, capacity_hint=capacity, rank=rank, n_ranks=world, rccl_unique_id=box[0])
        transport = "rccl (ncclSend / ncclRecv issued by the engine on its stream)"
    else:
        # functional double (ranks may share a GPU): the same cs_mesh_* calls over a host transport on gloo
        stepper = NativeTileMesh(spatial_index, tiling, halo, device=device, density_per_cell=density, flags=flags,
                                 weights=weights, capacity_hint=capacity, rank=rank, n_ranks=world,
                                 host_transport=TorchHostTransport(dist))
        transport = f"cs_mesh_host_transport over torch.distributed ({backend}), staged through pinned host memory"
    return stepper, stepper.tile(0), {"mesh": "native", "transport": transport,
                                      "entry_point": "cs_mesh_step (include/crowdstep.h; lib.rs:195)"}


def verify_mesh(args, ctx, stepper, mesh_kind, steps_made, fill, single_engine, grid, n_total, flags):
    """The whole crowd of the mesh (collective: every rank gets it) against a single engine that rank 0 steps through
    the same scene for the same number of steps: bit for bit."""
    from rmf_crowdsim_amd import _abi
    torch, dist, rank = ctx["torch"], ctx["dist"], ctx["rank"]
    if mesh_kind != "native":
        return {"mesh_equals_single_engine": None, "note": "--verify needs --mesh native (cs_mesh_read_agents gathers the crowd)"}
    crowd = stepper.read_agents()
    verdict = [None]
    if rank == 0:
        ref = single_engine(grid, n_total + 4096, flags & ~_abi.CS_CFG_TILE_OVERLAP)
        made = fill(ref)
        for _ in range(steps_made - made):
            ref.step(0.05, report=False)
        want = ref.read_agents()
        same = len(want) == len(crowd) and want.tobytes() == crowd.tobytes()
        verdict = [{"mesh_equals_single_engine": bool(same), "steps_compared": steps_made, "agents": int(len(want)),
                    "agents_on_the_mesh": int(len(crowd))}]
        del ref
    dist.broadcast_object_list(verdict, src=0)
    return verdict[0]


def run_leg(args, ctx, scaling, steps, warmup, clock_warmup_min, headline, overlap=None):
    """One timed run of `steps` steps (after `warmup` + clock warm-up untimed ones) under `scaling`.
    headline=True also builds the JSON line (roofline, creep leg, CPU baselines).  overlap: CS_CFG_TILE_OVERLAP for this
    leg's mesh (default: args.overlap)."""
    overlap = args.overlap if overlap is None else overlap
    torch, dist = ctx["torch"], ctx["dist"]
    rank, world, device, backend = ctx["rank"], ctx["world"], ctx["device"], ctx["backend"]
    from rmf_crowdsim_amd import Simulation, scenes, _abi

    per_gpu = args.agents if scaling == "weak" else max(1, args.agents // world)
    n_total = per_gpu * world
    # The counter-flow closes the lattice gaps at 2 * speed; once the first pair (of a million)
    # gets within the model's collision distance its t_i -> 0, the force clamps at 1e15 and the
    # step fails with "Index out of bounds", on the reference's f64 path as well (DESIGN.md
    # section 5).  The kernel's cost does not depend on the speed scale (measured: 1e-3, 1e-4 and 1e-5 m/s
    # give the same time), so long runs creep slower: at most 2.5 cm of closing over the run.
    clock_warmup = max(0, clock_warmup_min - warmup)  # extra untimed steps before the W warm-up steps
    if args.speed is None:
        speed = min(scenes.CREEP_SPEED, 0.25 / (steps + warmup + clock_warmup + 2))
    else:
        speed = args.speed
    flags = {"auto": 0, "gather": 1, "tiled": 2}[args.kernel] | (args.debug << 8)
    if args.workload == "hotspots":
        flags |= _abi.CS_CFG_DENSE  # more than 64 neighbours in sight in the cores
    if overlap and world > 1:
        flags |= _abi.CS_CFG_TILE_OVERLAP

    from rmf_crowdsim_amd import LocationHash2D, Zanlungo
    from rmf_crowdsim_amd.tiles import default_tiling
    arm, fail = ctx["arm"], ctx["fail"]
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    n_sinks = 0
    tile_report = {}
    uniform_kind = args.workload in ("walk", "creep")
    crowd = {"hotspots": scenes.hotspot_crowd, "random": scenes.random_crowd}.get(args.workload)
    total_steps = steps + warmup + clock_warmup + 2
    halo = int(np.ceil(args.eyesight / args.cell - 1e-9))
    hot = args.workload == "hotspots"
    tiling = (1, 1) if world == 1 else default_tiling(world)
    leg_name = f"{scaling}-scaling leg"

    def single_engine(grid, capacity, engine_flags=flags):
        return Simulation(LocationHash2D(**grid), device=device, flags=engine_flags,
                          stream=torch.cuda.current_stream().cuda_stream, capacity_hint=capacity)

    if args.workload == "stream":
        # BASELINE.json configs[3]: the population is spawned and despawned by source-sinks; every
        # step runs the spawn kernel, the sink test and the compaction in the re-sort
        from rmf_crowdsim_amd import MonotonicCrowd, SourceSink, StubHighLevelPlan
        lanes, grid, fill_steps = scenes.stream_lanes(n_total, lane_length=16.0, cell_size=args.cell)
        extent = grid["width"]
        pts = group = None
        density, capacity, weights = 1.5 * scenes.METRIC_DENSITY * args.cell ** 2, int(per_gpu * 1.3) + 4096, None
        speed = scenes.WALK_SPEED
    else:
        if uniform_kind:
            pts, grid, extent, group = scenes.uniform_crowd(
                n_total, seed=7, cell_size=args.cell, room=walk_room(total_steps) if args.workload == "walk" else 0.0)
        else:
            pts, grid, extent, group = crowd(n_total, seed=7, cell_size=args.cell)
        density = (3.0 if hot else 1.5) * scenes.METRIC_DENSITY * args.cell ** 2
        capacity = int(per_gpu * (1.5 if hot else 1.15)) + 4096
        # a clustered crowd gets cuts at the quantiles of its row / column histograms; so does the
        # walking crowd, which stands at the low-x end of its grid
        weights = pts if (hot or args.workload == "walk") else None

    def fill(target):
        """The scene into a single engine or a mesh; returns the steps made while filling (stream workload)."""
        if args.workload != "stream":
            populate(target, args.workload, pts, group, speed, lp, args.eyesight)
            return 0
        plans = {}
        if args.planner == "route":
            from rmf_crowdsim_amd import RouteFollower
            route_hlp = RouteFollower(lambda start, goal: [start, goal], scale=0.25, speed=scenes.WALK_SPEED)
        for src, dst, vel in lanes:
            hlp = route_hlp if args.planner == "route" else plans.setdefault(vel, StubHighLevelPlan(vel))
            target.add_source_sink(SourceSink(src, 0.5, MonotonicCrowd(1000.0), hlp, lp, [dst], False, args.eyesight))
        for _ in range(fill_steps):
            target.step(0.05, report=False)
        return fill_steps

    steps_made = 0
    if world == 1:
        sim = single_engine(grid, (int(per_gpu * 1.2) if args.workload == "stream" else per_gpu) + 4096)
        stepper = sim
        mesh_kind = None
    else:
        # one crowd of n_total agents, cut into spatial tiles, one tile per rank; every rank sees the global
        # add_agents call and keeps the agents of its own cells.  A rank that cannot set its tile up says where.
        mesh_kind = args.mesh
        arm(f"{leg_name}: mesh creation ({mesh_kind}; RCCL communicator init on the nccl backend)")
        failure = None
        try:
            stepper, sim, how = make_mesh(args, ctx, LocationHash2D(**grid), tiling, halo, density, capacity, flags, weights)
        except Exception as err:  # noqa: BLE001
            failure = err
        # Did every rank get its tile?  (Over the launcher's process group, which does not depend on the engine's own
        # communicator.)  If the NATIVE mesh could not be set up, e.g. the engine could not bind librccl or
        # ncclCommInitRank was refused (failures every rank meets alike, before its first collective), every rank drops
        # it and the run goes on with the Python orchestration over torch.distributed's own point-to-point calls, saying
        # so in the line: a scaling curve with a note beats none.  (A failure on SOME ranks only leaves the others inside
        # the creation's collectives: that is the watchdog's case.)  Anything else ends the rank with the phase named.
        try:
            bad = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        except Exception as err:  # noqa: BLE001 (the launcher's own process group does not work: nothing to fall back to)
            fail(f"{leg_name}: mesh creation ({mesh_kind}): {failure}; and the process group's all-reduce", err)
        if int(bad.item()):
            if mesh_kind != "native" or os.environ.get("CS_BENCH_NO_FALLBACK"):
                fail(f"{leg_name}: mesh creation ({mesh_kind})", failure or "another rank failed")
            print(f"bench: rank {rank}: the native mesh could not be created ({failure or 'on another rank'}); falling back to "
                  f"--mesh python over torch.distributed", file=sys.stderr, flush=True)
            stepper = sim = None
            os.environ["CS_TILES_TRANSPORT"] = "torch"
            arm(f"{leg_name}: mesh creation (python, fallback)")
            try:
                fb = argparse.Namespace(**{**vars(args), "mesh": "python"})
                stepper, sim, how = make_mesh(fb, ctx, LocationHash2D(**grid), tiling, halo, density, capacity, flags, weights)
            except Exception as err:  # noqa: BLE001
                fail(f"{leg_name}: mesh creation (python, fallback)", err)
            mesh_kind = "python"
            how["mesh"] = "python"
            how["native_mesh_failed"] = str(failure or "on another rank")
        tile_report.update(how)
        if pts is not None:
            counts = TileCounts(LocationHash2D(**grid), tiling, weights, halo).of(pts)
            tile_report.update({"agents_per_tile": counts.reshape(-1).tolist(),
                                "imbalance_max_over_mean": float(counts.max() / counts.mean())})
    n_sinks = len(lanes) if args.workload == "stream" else 0
    arm(f"{leg_name}: populating the crowd")
    try:
        steps_made += fill(stepper)
    except Exception as err:  # noqa: BLE001
        fail(f"{leg_name}: populating the crowd", err)
    if args.workload != "stream" and not args.verify:
        del pts, group
    if world > 1:
        tile_report["ranks_in_comm"] = dist.get_world_size()
        tile_report["devices_visible"] = torch.cuda.device_count()
        # the first exchange with real peers, waited for: a transport that does not work fails HERE, by name
        arm(f"{leg_name}: first halo exchange + step (waited for)")
        try:
            stepper.step(0.05, report=True)
            steps_made += 1
        except Exception as err:  # noqa: BLE001
            fail(f"{leg_name}: first halo exchange + step", err)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def drain():
        """Waits for the device and surfaces what fire-and-forget steps left behind (a mesh: agreed over the ranks)."""
        (stepper if mesh_kind == "native" else sim).synchronize()

    arm(f"{leg_name}: warm-up steps")
    try:
        for _ in range(clock_warmup + warmup):
            stepper.step(0.05, report=False)
        steps_made += clock_warmup + warmup
        drain()
    except Exception as err:  # noqa: BLE001
        fail(f"{leg_name}: warm-up steps", err)
    sim.profile_reset()
    # hipEvents around K4 on the engine's stream; every 4th launch of the timed region, since an
    # event pair costs the stream ~6 us per step.  ONLY K4 inside the region the headline is taken from, at any N (round
    # 4 timed all ten phases of a tile's step there: ten event pairs per fourth step, some spanning two streams, inside
    # the wall clock that made the multi-GPU value); the phases get a short pass of their own after it.
    sim.profile_stride(max(1, args.profile_stride))
    sim.profile_enable(1 << _abi.CS_K_NEIGHBOUR_FORCE)
    arm(f"{leg_name}: timed region")
    sync_all()
    t0 = time.perf_counter()
    try:
        for _ in range(steps):
            stepper.step(0.05, report=False)
            if args.readback:  # frame k is fetched while step k + 1 runs
                sim.snapshot(wait=True)
                sim.request_snapshot()
        if args.readback:
            sim.snapshot(wait=True)
    except Exception as err:  # noqa: BLE001
        fail(f"{leg_name}: timed region", err)
    sync_all()
    elapsed = time.perf_counter() - t0
    steps_made += steps
    sim.profile_enable(0)
    prof = sim.profile_read()
    arm(f"{leg_name}: after the timed region (error check, report step)")
    try:
        if not args.debug:
            drain()  # surfaces "Index out of bounds" if any step left the grid
    except Exception as err:  # noqa: BLE001
        fail(f"{leg_name}: a step of the timed region failed", err)

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # the scene must still be the scene: everyone alive and finite
    if args.debug:
        rep = {"n_tti_zero": -1, "n_nonfinite": -1, "n_agents": per_gpu}
    else:
        stepper.step(0.05, report=True)
        steps_made += 1
        rep = stepper.last_report if mesh_kind == "native" else sim.last_report
    total_agents = n_total
    t_n = torch.tensor([rep["n_agents"], rep["n_tti_zero"], rep["n_nonfinite"]], dtype=torch.int64,
                       device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t_n)
    alive_all = int(t_n[0].item())
    if args.workload == "stream":
        total_agents = alive_all  # what the sinks actually sustain, over all tiles
    elif not args.debug and alive_all != n_total:
        raise SystemExit(f"bench: {alive_all} of {n_total} agents alive after the run")
    k4 = prof["neighbour_force"]
    k4_ms = k4["total_ms"] / max(k4["launches"], 1)
    if world > 1:
        # every phase of a tile's step (pack, exchange, unpack, border / interior launch, scan, scatter), device time
        # between hipEvents on the stream the phase ran on, in a pass of its own OUTSIDE the headline's clock: rank 0's and
        # the slowest rank's (the exchange includes the wait for the slowest peer: that is what the step pays)
        arm(f"{leg_name}: per-phase pass (after the timed region)")
        try:
            sim.profile_reset()
            sim.profile_stride(2)
            sim.profile_enable((1 << _abi.CS_K_COUNT) - 1)
            phase_steps = 24
            for _ in range(phase_steps):
                stepper.step(0.05, report=False)
            steps_made += phase_steps
            drain()
            sim.profile_enable(0)
            prof = sim.profile_read()
        except Exception as err:  # noqa: BLE001
            fail(f"{leg_name}: per-phase pass", err)
        names = list(_abi.KERNEL_NAMES)
        us = torch.tensor([1e3 * prof[k]["total_ms"] / prof[k]["launches"] if prof[k]["launches"] else 0.0 for k in names],
                          dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        us_max = us.clone()
        dist.all_reduce(us_max, op=dist.ReduceOp.MAX)
        tile_report["phase_us_rank0"] = {k: round(float(v), 2) for k, v in zip(names, us.tolist()) if v > 0}
        tile_report["phase_us_max_over_ranks"] = {k: round(float(v), 2) for k, v in zip(names, us_max.tolist()) if v > 0}
        tile_report["phase_us_note"] = ("24 steps after the timed region, an event pair at every second launch of each kind; "
                                        "neighbour_force spans step_border + step_interior when the launch is split "
                                        "(overlap); halo_pack is absent when the step kernel packed; with the overlap the "
                                        "exchange runs on the second stream, beside step_interior")
        tile_report["overlap"] = bool(overlap)
        if mesh_kind == "native":
            tile_report["exchange_bytes_per_step_rank0"] = stepper.exchange_bytes
            tile_report["exchanges_ahead"] = sim.kernel_stat(_abi.CS_STAT_EXCHANGES_AHEAD)
            tile_report["exchanges_ahead_used"] = sim.kernel_stat(_abi.CS_STAT_EXCHANGES_AHEAD_USED)
    if args.verify and world > 1:
        arm(f"{leg_name}: --verify (mesh == single engine)")
        tile_report["verify"] = verify_mesh(args, ctx, stepper, mesh_kind, steps_made, fill, single_engine, grid,
                                            n_total, flags)
    leg = {"value": total_agents * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "total_agents": total_agents,
           "per_gpu": per_gpu, "steps": steps, "k4_ms": k4_ms, "tile_report": tile_report, "speed": speed}
    if not headline:
        del stepper, sim
        return leg
    ncells = int(round(grid["width"] / grid["cell_size"])) ** 2 // world  # per tile
    agents_here = rep["n_agents"] if args.workload == "stream" else per_gpu
    alg_bytes = agents_here * (K4_READ_BYTES + K4_WRITE_BYTES) + K4_CELL_BYTES * ncells
    fused_bytes = agents_here * K4_FUSED_BYTES_PER_AGENT + K4_FUSED_BYTES_PER_CELL * ncells
    achieved = alg_bytes / (k4_ms * 1e-3) / 1e9 if k4_ms > 0 else 0.0

    desc = {
        "walk": (f"{per_gpu} agents/GPU uniform {scenes.METRIC_DENSITY}/m^2 jittered lattice, everyone walking +x at "
                 f"{scenes.WALK_SPEED} m/s (6.5 cm per step: ~3 % change cell every step) with a counter-flow of "
                 f"+-{speed} m/s in y on top, Zanlungo(A=1,D=0.4,m=2,R=0.2), eyesight {args.eyesight} m, "
                 f"LocationHash2D cell {args.cell} m, dt 0.05 s"),
        "creep": (f"{per_gpu} agents/GPU uniform {scenes.METRIC_DENSITY}/m^2 jittered lattice, counter-flow {speed} m/s, "
                  f"Zanlungo(A=1,D=0.4,m=2,R=0.2), eyesight {args.eyesight} m, LocationHash2D cell {args.cell} m, "
                  f"dt 0.05 s"),
        "hotspots": (f"{per_gpu} agents/GPU, half uniform background, half in Gaussian hotspots (sigma 5 m, 800 agents "
                     f"each), counter-flow {speed} m/s, Zanlungo(A=1,D=0.4,m=2,R=0.2), eyesight {args.eyesight} m, cell "
                     f"{args.cell} m, dt 0.05 s"),
        "random": (f"{per_gpu} agents/GPU, {scenes.METRIC_DENSITY}/m^2 on a randomly thinned 0.45 m lattice (neighbour "
                   f"counts scatter like a real crowd's), counter-flow {speed} m/s, Zanlungo(A=1,D=0.4,m=2,R=0.2), "
                   f"eyesight {args.eyesight} m, cell {args.cell} m, dt 0.05 s"),
        "stream": (f"~{per_gpu} agents/GPU sustained by {n_sinks} source-sinks (MonotonicCrowd, lanes 1 m apart, "
                   f"alternating direction, 1.3 m/s), Zanlungo(A=1,D=0.4,m=2,R=0.2), eyesight {args.eyesight} m, cell "
                   f"{args.cell} m, dt 0.05 s"),
    }[args.workload]

    # HBM traffic and instruction counts of the kernel from the PMC counters (separate rocprofv3
    # passes, FETCH_SIZE x2 on gfx950; tools/rocprof_passes.sh) are measured offline and cached under
    # profiles/.  The cache is keyed by the kernel sources, the compiler flags and the workload: a
    # summary taken on anything else is stale and reads as null here.
    key = profile_key(f"{args.workload} agents {per_gpu} cell {args.cell} eyesight {args.eyesight} kernel {args.kernel} "
                      f"world {world}")
    traffic = valu_insts = valu_active = None
    pmc_source = None
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "k4_traffic.json")) as f:
                tr = json.load(f)
            if tr.get("profile_key") == key and not args.debug:
                traffic, valu_insts = tr["traffic_bytes_per_launch"], tr.get("valu_wave_insts_per_launch")
                valu_active = tr.get("valu_active_quad_cycles_per_launch")
                pmc_source = f"profiles/{rnd}/k4_traffic.json"
                break
        except (OSError, KeyError, ValueError):
            continue

    # Two more scenes timed beside the walking one, in the same run: the creep scene (same crowd, standing, NON-ZERO
    # forces: in the walking scene the force term is computed in full and underflows to exactly 0)
    # and the scattered crowd (a randomly thinned lattice: neighbour counts scatter like a real crowd's, forces of
    # every size), the dearest member of the family at this density.
    side_legs = {}
    if rank == 0 and world == 1 and args.workload == "walk" and not args.no_creep_leg and not args.debug:
        del stepper, sim
        for name in ("creep", "random"):
            if name == "creep":
                c_sim, _, _ = build_crowd(Simulation, per_gpu, args.cell, args.eyesight, speed, workload="creep",
                                          device=device, stream=torch.cuda.current_stream().cuda_stream, capacity=per_gpu + 1024)
            else:
                r_pts, r_grid, _, r_group = scenes.random_crowd(per_gpu, seed=7, cell_size=args.cell)
                c_sim = Simulation(LocationHash2D(**r_grid), device=device, flags=flags,
                                   stream=torch.cuda.current_stream().cuda_stream, capacity_hint=per_gpu + 1024)
                populate(c_sim, "random", r_pts, r_group, speed, lp, args.eyesight)
                del r_pts, r_group
            for _ in range(max(10, warmup + clock_warmup)):  # the same untimed steps as the headline leg (the clocks
                c_sim.step(0.05, report=False)                # have dropped while the host built this crowd)
            c_sim.synchronize()
            c_sim.profile_reset()
            c_sim.profile_stride(4)
            c_sim.profile_enable(1 << _abi.CS_K_NEIGHBOUR_FORCE)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for _ in range(40):
                c_sim.step(0.05, report=False)
            torch.cuda.synchronize()
            c_el = time.perf_counter() - c0
            c_sim.profile_enable(0)
            cp = c_sim.profile_read()["neighbour_force"]
            c_sim.step(0.05, report=True)
            side_legs[name] = {"workload": name, "kernel_ms": cp["total_ms"] / max(cp["launches"], 1),
                               "ms_per_step": c_el / 40 * 1e3, "value": per_gpu * 40 / c_el, "steps": 40,
                               "n_tti_zero": c_sim.last_report["n_tti_zero"], "n_nonfinite": c_sim.last_report["n_nonfinite"],
                               "scene_stats": scene_stats(name, per_gpu, args)}
            del c_sim
    creep_leg = side_legs.get("creep")

    if rank == 0:
        ceiling = valu_ceiling()
        valu_peak = ceiling["fma_wave_insts_per_s"] if ceiling else VALU_ISSUE_PEAK
        valu_frac = (valu_insts / (k4_ms * 1e-3) / valu_peak) if (valu_insts and k4_ms > 0) else None
        out = {
            "metric": "agent-steps/sec at 1M agents, dt=0.05 s; % HBM roofline on Zanlungo kernel",
            "value": total_agents * steps / elapsed,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "workload_name": args.workload,
                "readback": "every step, 32 B/agent to pinned host memory" if args.readback else "none",
                "planner": args.planner,
                "n_spawned_last_step": rep.get("n_spawned"), "n_destroyed_last_step": rep.get("n_destroyed"),
                "agents_per_gpu": per_gpu, "agents_total": total_agents, "eyesight": args.eyesight, "cell": args.cell,
                "speed": speed, "kernel": args.kernel,
                "parallelism": "1 GPU" if world == 1 else
                f"{tiling[0]}x{tiling[1]} spatial tiles, one per GPU, one halo exchange with the (up to) 8 neighbours over "
                f"{backend} send/recv" + (", exchange overlapped with the interior windows" if (overlap and world > 1) else ""),
                "n_tti_zero": int(t_n[1].item()), "n_nonfinite": int(t_n[2].item()),
                "n_agents_alive": alive_all,
                "clock_warmup_steps": clock_warmup,
                "profile_key": key,
                **tile_report,
            },
            "roofline": {
                # the nominal bound of a neighbour gather, with SURVEY.md section 8(d)'s algorithmic bytes
                # (48 B per agent + 8 B per cell)
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                # the figure north_star's target is stated in: the kernel's algorithmic READ bytes alone (32 B per
                # agent: position, velocity, own preferred velocity, id, group) against the HBM peak
                "hbm_read_frac": (agents_here * K4_READ_BYTES / (k4_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if k4_ms > 0 else 0.0,
                "hbm_read_target_frac": 0.40,
                # the whole step (K1-K5 without spawn) against the same peak: SURVEY.md section 8(d)'s ~120 B per
                # agent-step over the step's wall time
                "whole_step_bytes_per_agent": 120, "whole_step_frac": agents_here * 120 / (elapsed / steps) / 1e9 / HBM_PEAK_GBPS,
                "kernel": "k_step_tiled" if args.kernel != "gather" else "k_step_gather",
                "kernel_ms": k4_ms, "kernel_launches_timed": k4["launches"],
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes": "SURVEY.md 8(d): 32 B read + 16 B written per agent, 8 B per cell",
                "fused_kernel_bytes_per_launch": fused_bytes,
                # what actually bounds it: wave64 VALU instructions per launch (PMC) against the chip's measured
                # issue rate of independent v_fma_f32 (tools/valu_ceiling.hip -> profiles/rNN/valu_ceiling.json;
                # the nominal one per 2 clocks per SIMD at 2.4 GHz while no measurement is cached)
                "bound_actual": "valu_issue",
                "valu_issue_frac": valu_frac,
                # what the counters say about the pipe itself: SQ_ACTIVE_INST_VALU (in units of four SIMD cycles) against
                # the kernel's duration on all 1,024 SIMDs at the 2.4 GHz peak clock: the share of cycles in which a SIMD
                # was executing a vector instruction of ANY rate class (valu_issue_frac prices the count as if all were
                # full rate; the gap between the two is the half- and quarter-rate instructions)
                "valu_busy_frac": (valu_active * 4.0 / (N_SIMDS * k4_ms * 1e-3 * SIMD_CLOCK_HZ)) if (valu_active and k4_ms > 0) else None,
                "valu_wave_insts_per_launch": valu_insts,
                "valu_issue_peak_per_s": valu_peak,
                "valu_ceiling": ceiling,
                "pmc_source": pmc_source,
                "note": "HBM is the nominal bound of a neighbour gather; at ~31 neighbours per agent the kernel is "
                        "bound by instruction issue (DESIGN.md section 4, measured issue costs); traffic / valu_* are null unless a "
                        "PMC summary taken on exactly these sources and this workload is cached under profiles/",
            },
        }
        out["scene_stats"] = scene_stats(args.workload, per_gpu, args)
        if creep_leg:
            out["creep_scene"] = creep_leg
            # the same crowd with forces that do not underflow: the figure to quote when the force path must count
            out["value_full_force"] = creep_leg["value"]
        if "random" in side_legs:
            out["scattered_scene"] = side_legs["random"]
        if not args.no_cpu_baseline and world == 1:  # (N > 1: main() adds them once every leg is measured)
            out.update(cpu_baselines(args, per_gpu, speed))
        leg["line"] = out
    return leg


def scene_stats(workload, agents, args):
    """What the neighbour pass of a scene consists of, measured offline with the kernel's trip counters (a
    -DCS_TILE_TRIPS build, tools/trip_counts.py) and cached under profiles/rNN/scene_stats.json: mean neighbours in
    sight, mean neighbours an agent yields to (the force terms that are not identically 0), the share of agents with a
    finite time to collision (whose force pass runs at all), loop trips per wave.  Keyed like k4_traffic.json: a
    summary taken on other sources or another scene reads as null."""
    key = profile_key(f"scene {workload} agents {agents} cell {args.cell} eyesight {args.eyesight}")
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "scene_stats.json")) as f:
                table = json.load(f)
            if key in table:
                return {**table[key], "source": f"profiles/{rnd}/scene_stats.json", "profile_key": key}
        except (OSError, ValueError):
            continue
    return {"profile_key": key, "note": "no cached trip-counter summary for these sources and this scene"}


def valu_ceiling():
    """Measured VALU issue rates of the leased chip (tools/valu_ceiling.hip, run by tools/rocprof_passes.sh):
    the newest profiles/rNN/valu_ceiling.json, or None."""
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "valu_ceiling.json")) as f:
                c = json.load(f)
            pc = c.get("per_class", {})
            at4 = lambda k: pc.get(k, {}).get("waves_per_simd_4")  # noqa: E731
            # (the line carries the three classes' rates, not the table of fifty)
            return {"source": f"profiles/{rnd}/valu_ceiling.json", "device": c.get("device"),
                    "unit": "wave64 instructions per second, whole chip, 4 waves per SIMD",
                    "fma_wave_insts_per_s": c["fma_wave_insts_per_s"],
                    "half_rate_class_per_s": at4("v_cvt_f32_i32"), "quarter_rate_class_per_s": at4("v_sqrt_f32"),
                    "one_wave_per_simd_fma_per_s": pc.get("v_fma_f32", {}).get("waves_per_simd_1")}
        except (OSError, ValueError, KeyError):
            continue
    return None


if __name__ == "__main__":
    main()
