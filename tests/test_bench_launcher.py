"""bench.py --gpus N: the rank launcher and its guards ("Next" item 1).

CPU tests: the launcher starts `torch.distributed.run` as a child before torch is imported, and a
rank refuses a WORLD_SIZE that differs from --gpus.  GPU test: `python bench.py --gpus 2` as a plain
command (ranks sharing the one GPU of the test box over gloo) prints "n_gpus": 2.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_gpus_n_without_a_launcher_starts_n_ranks(tmp_path):
    """--gpus 3 with no WORLD_SIZE: three ranks run (each fails here for want of a GPU, with the
    engine's message, which proves they were started and parsed --gpus) and the exit code is theirs."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "1", "--warmup", "0", "--agents", "1000",
                        "--no-cpu-baseline"], env=_env(CS_BENCH_BACKEND="gloo"), capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    err = p.stderr + p.stdout
    # every rank got as far as the device check: world size matched --gpus
    assert err.count("bench.py needs an MI355X") >= 1, err[-2000:]
    assert "WORLD_SIZE=" not in err, err[-2000:]


def test_a_rank_refuses_a_world_size_that_is_not_gpus():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1"],
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 1 but WORLD_SIZE=2" in p.stderr


def test_launcher_is_a_no_op_under_a_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")

    class A:
        gpus = 4
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.launch_ranks_if_needed(A()) is None  # returns instead of spawning


def _bench_line(p):
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_bench_gpus_2_steps_the_native_mesh_and_it_equals_the_single_engine():
    """The driver's command shape without its launcher: two ranks (ranks sharing the one GPU, so the cs_mesh_* calls
    go over the gloo host transport), strong scaling = 200k agents in all, the weak-scaled leg beside it.  What is
    stepped is the C ABI's mesh (one cs_mesh_step per step), and after the timed region of either leg the whole crowd
    of the mesh equals a single engine stepped through the same scene, bit for bit (--verify)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--agents", "200000", "--steps", "10", "--warmup", "3",
                        "--clock-warmup", "5", "--verify"], env=_env(CS_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=900)
    line = _bench_line(p)
    assert line["n_gpus"] == 2
    assert line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["mesh"] == "native" and cfg["entry_point"].startswith("cs_mesh_step")
    assert cfg["agents_total"] == 200000 and cfg["agents_per_gpu"] == 100000
    assert cfg["ranks_in_comm"] == 2
    assert cfg["verify"]["mesh_equals_single_engine"] is True and cfg["verify"]["agents"] == 200000
    # first exchange (1), clock warm-up up to 5 untimed steps in all (2), warm-up (3), timed (10), the report step (1),
    # the per-phase pass outside the headline's clock (24)
    assert cfg["verify"]["steps_compared"] == 1 + 2 + 3 + 10 + 1 + 24
    assert cfg["exchange_bytes_per_step_rank0"] > 0
    phases = cfg["phase_us_rank0"]
    assert phases["neighbour_force"] > 0 and phases["halo_unpack"] > 0 and phases["scan"] > 0 and phases["scatter"] > 0
    assert line["weak_scaled"]["agents_total"] == 400000
    assert line["weak_scaled"]["verify"]["mesh_equals_single_engine"] is True
    assert line["value"] > 0 and line["weak_scaled"]["value"] > 0
    # round 5: the overlapped schedule is timed beside the headline for 20 steps (last of the legs); N > 1 lines carry both
    # CPU figures
    ab = cfg["overlap_ab"]
    assert cfg["overlap"] is False and ab["headline_is"] == "no_overlap"
    assert ab["no_overlap"]["ms_per_step"] == line["ms_per_step"] and ab["overlap"]["ms_per_step"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["kind"] == "port"
    assert line["cpu_baseline_openmp"]["value"] > line["cpu_baseline"]["value"]


@pytest.mark.gpu
def test_bench_gpus_2_with_the_overlap_as_the_headline_runs_the_plain_leg_first():
    """--overlap: the schedule that has never met real peers is the headline only behind a complete fallback line (the
    short leg without it runs first); when it finishes, the line is its own and carries both."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--agents", "100000", "--steps", "5", "--warmup", "2",
                        "--clock-warmup", "3", "--no-cpu-baseline", "--no-second-scaling-leg", "--overlap"],
                       env=_env(CS_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    line = _bench_line(p)
    ab = line["config"]["overlap_ab"]
    assert "fallback" not in line and line["config"]["overlap"] is True and ab["headline_is"] == "overlap"
    assert ab["overlap"]["ms_per_step"] == line["ms_per_step"] and ab["no_overlap"]["ms_per_step"] > 0
    assert "ran first" in ab["note"]


@pytest.mark.gpu
def test_bench_gpus_2_split_launches_at_configs2_size_equal_the_single_engine():
    """The decomposition the overlapped schedule steps on (border windows as a launch of their own, CS_TILE_SPLIT=1) with
    TWO RANKS at configs[2]'s full million agents, over the gloo host transport: after warm-up, timed region, report step
    and per-phase pass the whole crowd of the mesh == one engine, bit for bit (--verify)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--agents", "1000000", "--steps", "10", "--warmup", "3",
                        "--clock-warmup", "5", "--no-cpu-baseline", "--no-second-scaling-leg", "--no-overlap-ab", "--verify"],
                       env=_env(CS_BENCH_BACKEND="gloo", CS_TILE_SPLIT="1"), capture_output=True, text=True, timeout=900)
    line = _bench_line(p)
    cfg = line["config"]
    assert cfg["agents_total"] == 1_000_000 and cfg["verify"]["mesh_equals_single_engine"] is True
    assert cfg["verify"]["agents"] == 1_000_000 and cfg["verify"]["steps_compared"] == 1 + 2 + 3 + 10 + 1 + 24
    assert cfg["phase_us_rank0"].get("step_border", 0) > 0 and cfg["phase_us_rank0"].get("step_interior", 0) > 0


@pytest.mark.gpu
def test_bench_gpus_2_python_mesh_for_comparison():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--agents", "100000", "--steps", "5", "--warmup", "2",
                        "--clock-warmup", "3", "--no-cpu-baseline", "--mesh", "python", "--no-second-scaling-leg", "--no-overlap-ab"],
                       env=_env(CS_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    line = _bench_line(p)
    assert line["config"]["mesh"] == "python" and line["n_gpus"] == 2 and line["value"] > 0


@pytest.mark.gpu
def test_a_native_mesh_that_cannot_be_created_falls_back_to_the_python_mesh_on_every_rank():
    """cs_mesh_create fails on the ranks (here: a test hook; on a real node: librccl not bound, ncclCommInitRank
    refused): they agree over the launcher's process group, all drop the native mesh, and the run is measured on
    the Python orchestration, saying so; with CS_BENCH_NO_FALLBACK the same failure ends the run with the phase named."""
    common = [sys.executable, BENCH, "--gpus", "2", "--agents", "60000", "--steps", "5", "--warmup", "2", "--clock-warmup", "3",
              "--no-cpu-baseline", "--no-second-scaling-leg", "--no-overlap-ab"]
    p = subprocess.run(common, env=_env(CS_BENCH_BACKEND="gloo", CS_BENCH_BREAK_NATIVE_MESH="1"), capture_output=True, text=True,
                       timeout=900)
    line = _bench_line(p)
    assert line["config"]["mesh"] == "python" and "CS_BENCH_BREAK_NATIVE_MESH" in line["config"]["native_mesh_failed"]
    assert line["value"] > 0 and "falling back" in p.stderr
    p = subprocess.run(common, env=_env(CS_BENCH_BACKEND="gloo", CS_BENCH_BREAK_NATIVE_MESH="1", CS_BENCH_NO_FALLBACK="1"),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode != 0 and "FAILED in phase" in p.stderr and "mesh creation (native)" in p.stderr


@pytest.mark.gpu
def test_a_rank_that_cannot_set_up_its_transport_says_where_and_exits_non_zero():
    """RCCL refuses two ranks on one device; with the device check bypassed the native mesh's communicator init (or
    the first exchange) fails: every rank must leave with a non-zero code and the phase named, not hang."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--agents", "20000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--watchdog", "120"], env=_env(CS_BENCH_SHARE_DEVICE="1", CS_BENCH_NO_FALLBACK="1"),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode != 0
    err = p.stderr + p.stdout
    assert "FAILED in phase" in err or "made no progress" in err, err[-3000:]
    assert "process group init" in err or "mesh creation" in err or "first halo exchange" in err, err[-3000:]


@pytest.mark.gpu
def test_nccl_ranks_may_not_share_a_device():
    """On the nccl backend two ranks on a one-GPU box must fail loudly, not land on one device."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--agents", "20000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "visible GPU" in (p.stderr + p.stdout)
