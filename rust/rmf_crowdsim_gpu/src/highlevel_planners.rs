// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).
//
// Attribution: the trait and struct declarations in this file restate, signature for signature, those of
// open-rmf/rmf_crowdsim (Copyright (C) 2022 Open Source Robotics Foundation, licensed under the Apache License,
// Version 2.0: http://www.apache.org/licenses/LICENSE-2.0), file rmf_crowdsim/src/highlevel_planners/highlevel_planners.rs; they are kept
// identical on purpose, so that this crate drops in for that one.  Everything else in the file is original.
use crate::{Agent, AgentId, Vec2f};

/// How a planner runs on the device.  The default is the slow path: a batched host callback.
pub enum DeviceHighLevelPlan {
    /// `get_desired_velocity` returns None for everybody (lib.rs:263-273 leaves vel = 0)
    None,
    /// Some(v) for everybody: the reference tests' StubHighLevelPlan (lib.rs:391-420)
    Constant(Vec2f),
    /// even ids -> Some(-v), odd ids -> Some(v): the visualiser's stub (rmf_crowdsim_viz/src/main.rs:20-30)
    IdParity(Vec2f),
    /// RMFPlanner's follower (rmf/mod.rs:195-242) on the device; the planner's `plan_route` is
    /// called once per new (start, goal) SpatialHash pair (rmf/mod.rs:217-236)
    Route { scale: f64, arrive: f64, speed: f64 },
    /// anything else: `get_desired_velocity` / `set_target` / `remove_agent_id` are called on the
    /// host, batched per step
    HostCallback,
}

/// highlevel_planners/highlevel_planners.rs:8-16, plus the two provided methods the GPU backend
/// needs (`device_form`, `plan_route`); existing implementations compile unchanged.
pub trait HighLevelPlanner {
    fn get_desired_velocity(&mut self, agent: &Agent, time: std::time::Duration) -> Option<Vec2f>;

    /// Set the target position for a given agent
    fn set_target(&mut self, agent: &Agent, point: Vec2f, tolerance: Vec2f);

    /// Remove an agent
    fn remove_agent_id(&mut self, _agent: AgentId) {}

    fn device_form(&self) -> DeviceHighLevelPlan {
        DeviceHighLevelPlan::HostCallback
    }

    /// `RMFPlanner::plan_route` (rmf/mod.rs:160-192) for `DeviceHighLevelPlan::Route` planners:
    /// the waypoints of a route from `start` to `goal`, the goal last; None = no contiguous path.
    fn plan_route(&mut self, _start: Vec2f, _goal: Vec2f) -> Option<Vec<Vec2f>> {
        None
    }
}

/// The reference tests' stub planner (lib.rs:391-420, tests/event_listeners_test.rs:6-35) as a
/// ready-made device planner.
pub struct StubHighLevelPlan {
    pub default_vel: Vec2f,
}

impl StubHighLevelPlan {
    pub fn new(default_vel: Vec2f) -> Self {
        StubHighLevelPlan { default_vel }
    }
}

impl HighLevelPlanner for StubHighLevelPlan {
    fn get_desired_velocity(&mut self, _agent: &Agent, _time: std::time::Duration) -> Option<Vec2f> {
        Some(self.default_vel)
    }
    fn set_target(&mut self, _agent: &Agent, _point: Vec2f, _tolerance: Vec2f) {}
    fn device_form(&self) -> DeviceHighLevelPlan {
        DeviceHighLevelPlan::Constant(self.default_vel)
    }
}
