// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).  Needs an MI355X to run.
//
// Known answers of the step path as tables, driven through this crate's trait surface.  The numbers are
// the ones the reference's own tests pin (rmf_crowdsim/src/lib.rs:423-453: one step of 1 s at (1, 0) m/s
// moves (0,0) to (1,0); rmf_crowdsim/tests/event_listeners_test.rs:97-110: a source 20 m from its sink,
// one agent per step at 1 m/s); the checks are written against the closed forms those numbers come from.
// The same answers are checked for real through the C ABI by tests/test_gpu_parity.py and
// tests/cpp/test_reference_api.cpp.
use rmf_crowdsim_gpu::highlevel_planners::{DeviceHighLevelPlan, HighLevelPlanner};
use rmf_crowdsim_gpu::local_planners::NoLocalPlan;
use rmf_crowdsim_gpu::source_sink::{MonotonicCrowd, SourceSink};
use rmf_crowdsim_gpu::spatial_index::LocationHash2D;
use rmf_crowdsim_gpu::*;
use std::sync::{Arc, Mutex};
use std::time::Duration;

/// A planner that always answers with one velocity and says so, which lets it run on the device.
struct Cruise(Vec2f);

impl HighLevelPlanner for Cruise {
    fn get_desired_velocity(&mut self, _: &Agent, _: Duration) -> Option<Vec2f> {
        Some(self.0)
    }
    fn set_target(&mut self, _: &Agent, _: Vec2f, _: Vec2f) {}
    fn device_form(&self) -> DeviceHighLevelPlan {
        DeviceHighLevelPlan::Constant(self.0)
    }
}

#[derive(Default)]
struct Ledger {
    born: Vec<AgentId>,
    gone: Vec<AgentId>,
}

impl EventListener for Ledger {
    fn agent_spawned(&mut self, _: Vec2f, id: AgentId) {
        self.born.push(id);
    }
    fn agent_destroyed(&mut self, id: AgentId) {
        self.gone.push(id);
    }
}

fn arena() -> Simulation<LocationHash2D> {
    Simulation::new(LocationHash2D::new(1000.0, 1000.0, 20.0, Point::new(-500.0, -500.0)))
}

/// (velocity, seconds per step, steps): where an agent that starts at the origin must end up.
const DRIFT_CASES: &[((f64, f64), u64, usize)] = &[((1.0, 0.0), 1, 1), ((0.0, -2.0), 1, 3), ((0.5, 0.5), 2, 4)];

#[test]
fn an_agent_without_a_local_plan_drifts_at_its_preferred_velocity() {
    for &((vx, vy), secs, steps) in DRIFT_CASES {
        let mut sim = arena();
        let ids = sim
            .add_agents(
                &vec![Point::new(0.0, 0.0)],
                Arc::new(Mutex::new(Cruise(Vec2f::new(vx, vy)))),
                Arc::new(Mutex::new(NoLocalPlan {})),
                100.0,
            )
            .unwrap();
        assert_eq!(ids, vec![0usize]);
        for _ in 0..steps {
            sim.step(Duration::new(secs, 0)).unwrap();
        }
        let t = (secs as f64) * (steps as f64);
        let p = sim.agents[&0usize].position;
        assert!((p - Point::new(vx * t, vy * t)).norm() < 1e-5, "{:?} after {} s", p, t);
    }
}

/// One agent leaves the source every step and walks `gap` metres to the sink at 1 m/s.  Before step k the
/// crowd holds min(k, gap) agents; once the stream is full, k - gap of them have been destroyed, oldest first.
#[test]
fn a_source_feeds_its_sink_at_the_walking_speed() {
    let gap = 20usize;
    let mut sim = arena();
    let ledger = Arc::new(Mutex::new(Ledger::default()));
    sim.add_event_listener(ledger.clone());
    sim.add_source_sink(Arc::new(SourceSink {
        source: Vec2f::new(0.0, 0.0),
        waypoints: vec![Vec2f::new(gap as f64, 0.0)],
        radius_sink: 1.0,
        crowd_generator: Arc::new(MonotonicCrowd::new(1.0)),
        high_level_planner: Arc::new(Mutex::new(Cruise(Vec2f::new(1.0, 0.0)))),
        local_planner: Arc::new(Mutex::new(NoLocalPlan {})),
        agent_eyesight_range: 5.0,
        loop_forever: false,
    }));
    for k in 0..(2 * gap) {
        assert_eq!(sim.agents.len(), k.min(gap), "population before step {}", k);
        {
            let seen = ledger.lock().unwrap();
            assert_eq!(seen.born.len(), k);
            assert_eq!(seen.gone.len(), k.saturating_sub(gap));
            assert!(seen.gone.iter().enumerate().all(|(n, &id)| id == n), "oldest first");
        }
        sim.step(Duration::new(1, 0)).unwrap();
    }
}
