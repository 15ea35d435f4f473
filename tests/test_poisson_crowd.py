"""PoissonCrowd (source_sink.rs:63-82) in the Python mirror: unseeded like the reference's thread_rng, so only its
statistics can be pinned (the reference has no test of it).  On the oracle (CPU) and on the engine (GPU)."""
import datetime

import numpy as np
import pytest

from oracle_sim import OracleSimulation
from rmf_crowdsim_amd import (LocationHash2D, NoLocalPlan, PoissonCrowd, Simulation, SourceSink, StubHighLevelPlan)


def test_poisson_crowd_draws_poisson_numbers():
    g = PoissonCrowd(40.0)
    draws = np.array([g.get_number_to_spawn(datetime.timedelta(seconds=0.05)) for _ in range(20000)])
    assert abs(draws.mean() - 2.0) < 0.06 and abs(draws.var() - 2.0) < 0.15  # mean = variance = dt * rate
    assert abs(np.mean(draws == 0) - np.exp(-2.0)) < 0.015
    other = PoissonCrowd(40.0)
    again = np.array([other.get_number_to_spawn(datetime.timedelta(seconds=0.05)) for _ in range(200)])
    assert (again != draws[:200]).any()  # unseeded: two generators do not repeat each other


def _spawned_in_300_steps(cls):
    """dt * rate = 0.5 and a source that is always free (the walker leaves 1 m per step, occupancy radius 0.4:
    lib.rs:212-217): a step spawns with probability 1 - e^-0.5, one agent at most (lib.rs:207-219)."""
    sim = cls(LocationHash2D(1000.0, 1000.0, 20.0, (-500.0, -500.0)))
    sim.add_source_sink(SourceSink((0.0, 0.0), 1.0, PoissonCrowd(0.5), StubHighLevelPlan((1.0, 0.0)), NoLocalPlan(),
                                   [(400.0, 0.0)], False, 5.0))
    total = 0
    for _ in range(300):
        sim.step(1.0)
        assert sim.last_report["n_spawned"] in (0, 1)
        total += sim.last_report["n_spawned"]
    assert total == len(sim)
    return total


def test_poisson_crowd_through_the_oracle():
    p = 1.0 - np.exp(-0.5)
    assert abs(_spawned_in_300_steps(OracleSimulation) - 300 * p) < 5 * np.sqrt(300 * p * (1 - p))


@pytest.mark.gpu
def test_poisson_crowd_through_the_engine():
    p = 1.0 - np.exp(-0.5)
    assert abs(_spawned_in_300_steps(Simulation) - 300 * p) < 5 * np.sqrt(300 * p * (1 - p))
