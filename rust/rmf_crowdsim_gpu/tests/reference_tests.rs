// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).  Needs an MI355X to run.
//
// The reference's two step tests, against this crate instead of rmf_crowdsim:
//   test_step_integration                   rmf_crowdsim/src/lib.rs:423-453
//   test_event_listener_source_sink_api     rmf_crowdsim/tests/event_listeners_test.rs:65-111
// Only the `use` lines differ from the originals' bodies (rmf_crowdsim -> rmf_crowdsim_gpu, and the
// stub planner says it is a constant so that it runs on the device).  The same two tests run for
// real through the Python and C++ mirrors (tests/test_gpu_parity.py, tests/cpp/test_reference_api.cpp).
use rmf_crowdsim_gpu::highlevel_planners::{DeviceHighLevelPlan, HighLevelPlanner};
use rmf_crowdsim_gpu::local_planners::NoLocalPlan;
use rmf_crowdsim_gpu::source_sink::{MonotonicCrowd, SourceSink};
use rmf_crowdsim_gpu::spatial_index::LocationHash2D;
use rmf_crowdsim_gpu::*;
use std::sync::{Arc, Mutex};
use std::time::Duration;

struct StubHighLevelPlan {
    default_vel: Vec2f,
}

impl StubHighLevelPlan {
    fn new(default_vel: Vec2f) -> Self {
        StubHighLevelPlan { default_vel }
    }
}

impl HighLevelPlanner for StubHighLevelPlan {
    fn get_desired_velocity(&mut self, _agent: &Agent, _time: Duration) -> Option<Vec2f> {
        Some(self.default_vel)
    }
    fn set_target(&mut self, _agent: &Agent, _point: Vec2f, _tolerance: Vec2f) {}
    fn remove_agent_id(&mut self, _agent: AgentId) {}
    fn device_form(&self) -> DeviceHighLevelPlan {
        DeviceHighLevelPlan::Constant(self.default_vel)
    }
}

struct MockEventListener {
    pub added: Vec<AgentId>,
    pub removed: Vec<AgentId>,
}

impl EventListener for MockEventListener {
    fn agent_spawned(&mut self, _position: Vec2f, agent: AgentId) {
        self.added.push(agent);
    }
    fn agent_destroyed(&mut self, agent: AgentId) {
        self.removed.push(agent);
    }
}

#[test]
fn test_step_integration() {
    let velocity = Vec2f::new(1.0f64, 0.0f64);
    let step_size = Duration::new(1, 0);
    let stub_spatial = LocationHash2D::new(1000f64, 1000f64, 20f64, Point::new(-500f64, -500f64));
    let mut crowd_simulation = Simulation::<LocationHash2D>::new(stub_spatial);
    let agent_start_positions = vec![Point::new(0f64, 0f64)];
    let high_level_planner = Arc::new(Mutex::new(StubHighLevelPlan::new(velocity)));
    let local_planner = Arc::new(Mutex::new(NoLocalPlan {}));

    assert_eq!(crowd_simulation.agents.len(), 0usize);
    let res = crowd_simulation.add_agents(&agent_start_positions, high_level_planner, local_planner, 100f64);
    assert_eq!(res.unwrap(), vec![0usize]);
    assert_eq!(crowd_simulation.agents.len(), 1usize);

    crowd_simulation.step(step_size).unwrap();
    assert_eq!(crowd_simulation.agents.len(), 1usize);
    let position = crowd_simulation.agents[&0usize].position;
    assert!((position - Point::new(1f64, 0f64)).norm() < 1e-5f64);
}

#[test]
fn test_event_listener_source_sink_api() {
    let stub_spatial = LocationHash2D::new(1000f64, 1000f64, 20f64, Point::new(-500f64, -500f64));
    let mut crowd_simulation = Simulation::<LocationHash2D>::new(stub_spatial);
    let velocity = Vec2f::new(1.0f64, 0.0f64);
    let step_size = Duration::new(1, 0);
    let high_level_planner = Arc::new(Mutex::new(StubHighLevelPlan::new(velocity)));
    let local_planner = Arc::new(Mutex::new(NoLocalPlan {}));
    let crowd_generator = Arc::new(MonotonicCrowd::new(1f64));

    let source_sink = Arc::new(SourceSink {
        source: Vec2f::new(0f64, 0f64),
        radius_sink: 1f64,
        crowd_generator,
        high_level_planner,
        local_planner,
        waypoints: vec![Vec2f::new(20f64, 0f64)],
        loop_forever: false,
        agent_eyesight_range: 5f64,
    });
    let event_listener = Arc::new(Mutex::new(MockEventListener { added: vec![], removed: vec![] }));

    crowd_simulation.add_event_listener(event_listener.clone());
    crowd_simulation.add_source_sink(source_sink);

    for i in 0usize..20usize {
        assert_eq!(crowd_simulation.agents.len(), i);
        assert_eq!(event_listener.lock().unwrap().added.len(), i);
        crowd_simulation.step(step_size).unwrap();
    }
    for i in 20usize..40usize {
        assert_eq!(crowd_simulation.agents.len(), 20usize);
        assert_eq!(event_listener.lock().unwrap().added.len(), i);
        assert_eq!(event_listener.lock().unwrap().removed.len(), i - 20usize);
        crowd_simulation.step(step_size).unwrap();
    }
}
