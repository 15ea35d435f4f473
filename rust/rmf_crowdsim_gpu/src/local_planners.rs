// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).
//
// Attribution: the trait and struct declarations in this file restate, signature for signature, those of
// open-rmf/rmf_crowdsim (Copyright (C) 2022 Open Source Robotics Foundation, licensed under the Apache License,
// Version 2.0: http://www.apache.org/licenses/LICENSE-2.0), file rmf_crowdsim/src/local_planners/{local_planner,zanlungo,no_local_plan}.rs; they are kept
// identical on purpose, so that this crate drops in for that one.  Everything else in the file is original.
use crate::ffi;
use crate::{Agent, AgentId, Vec2f};

/// How a local planner runs on the device.
pub enum DeviceLocalPlan {
    NoLocalPlan,
    Zanlungo(ffi::cs_zanlungo_params),
    /// a user-defined planner (the default): host code, evaluated every step through the engine's batched
    /// callback with each agent's neighbours (cs_register_lp_callback: the slow path)
    HostCallback,
}

/// local_planners/local_planner.rs:7-18, plus `device_form`.
pub trait LocalPlanner {
    fn get_desired_velocity(&self, agent: &Agent, nearby_agents: &Vec<Agent>, recommended_velocity: Vec2f) -> Vec2f;

    fn add_agent(&mut self, _id: AgentId) {}

    fn remove_agent(&mut self, _id: AgentId) {}

    fn device_form(&self) -> DeviceLocalPlan {
        DeviceLocalPlan::HostCallback
    }
}

/// local_planners/no_local_plan.rs:7-18
pub struct NoLocalPlan {}

impl LocalPlanner for NoLocalPlan {
    fn get_desired_velocity(&self, _agent: &Agent, _nearby_agents: &Vec<Agent>, recommended_velocity: Vec2f) -> Vec2f {
        recommended_velocity
    }
    fn device_form(&self) -> DeviceLocalPlan {
        DeviceLocalPlan::NoLocalPlan
    }
}

/// local_planners/zanlungo.rs:9-48.  The arithmetic (:49-217) runs in the engine's neighbour kernel
/// (rmf_crowdsim_amd/csrc/cs_kernels_step.hip.inc); the host-side `get_desired_velocity` is not
/// used by `Simulation::step` and is not restated here.
pub struct Zanlungo {
    agent_scale: f64,
    obstacle_scale: f64,
    reaction_time: f64,
    force_distance: f64,
    agent_mass: f64,
    agent_radius: f64,
}

impl Zanlungo {
    pub fn new(agent_scale: f64, obstacle_scale: f64, reaction_time: f64, force_distance: f64, agent_mass: f64,
               agent_radius: f64) -> Self {
        Zanlungo { agent_scale, obstacle_scale, reaction_time, force_distance, agent_mass, agent_radius }
    }
}

impl LocalPlanner for Zanlungo {
    fn get_desired_velocity(&self, _agent: &Agent, _nearby_agents: &Vec<Agent>, _recommended_velocity: Vec2f) -> Vec2f {
        unimplemented!("Zanlungo runs on the device (Simulation::step); there is no host evaluation in this crate")
    }
    fn device_form(&self) -> DeviceLocalPlan {
        DeviceLocalPlan::Zanlungo(ffi::cs_zanlungo_params {
            agent_scale: self.agent_scale,
            obstacle_scale: self.obstacle_scale,
            reaction_time: self.reaction_time,
            force_distance: self.force_distance,
            agent_mass: self.agent_mass,
            agent_radius: self.agent_radius,
        })
    }
}
