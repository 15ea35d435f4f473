"""rmf_crowdsim_amd — MI355X-native engine for the `Simulation::step` hot path of rmf_crowdsim.

Host-side mirror of the reference trait surface over the C ABI of the HIP engine
(include/crowdstep.h, csrc/crowdstep_hip.hip).  There is no CPU fallback.
"""
from ._abi import (CS_CFG_DEFAULT, CS_CFG_DENSE, CS_CFG_FORCE_GATHER, CS_CFG_FORCE_TILED)
from .simulation import (Agent, CrowdGenerator, CrowdSimError, EventListener, HighLevelPlanner,
                         IdParityHighLevelPlan, LocalPlanner, LocationHash2D, MonotonicCrowd,
                         NoHighLevelPlan, NoLocalPlan, PoissonCrowd, RouteFollower, SeededPoissonCrowd, Simulation,
                         SourceSink, SpatialIndex,
                         StubHighLevelPlan, Zanlungo)

__all__ = [
    "Agent", "CrowdGenerator", "CrowdSimError", "EventListener", "HighLevelPlanner",
    "IdParityHighLevelPlan", "LocalPlanner", "LocationHash2D", "MonotonicCrowd",
    "NoHighLevelPlan", "NoLocalPlan", "PoissonCrowd", "RouteFollower", "SeededPoissonCrowd", "Simulation",
    "SourceSink", "SpatialIndex",
    "StubHighLevelPlan", "Zanlungo", "CS_CFG_DEFAULT", "CS_CFG_DENSE", "CS_CFG_FORCE_GATHER",
    "CS_CFG_FORCE_TILED",
]
