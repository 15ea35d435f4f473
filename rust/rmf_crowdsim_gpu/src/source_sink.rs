// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).
//
// Attribution: the trait and struct declarations in this file restate, signature for signature, those of
// open-rmf/rmf_crowdsim (Copyright (C) 2022 Open Source Robotics Foundation, licensed under the Apache License,
// Version 2.0: http://www.apache.org/licenses/LICENSE-2.0), file rmf_crowdsim/src/source_sink/source_sink.rs (the two generator bodies, :76-81 and :97-100, included); they are kept
// identical on purpose, so that this crate drops in for that one.  Everything else in the file is original.
use crate::{HighLevelPlanner, LocalPlanner, Vec2f};
use std::sync::{Arc, Mutex};
use std::time::Duration;

use rand::distributions::Distribution;
use statrs::distribution::Poisson;

/// How a generator runs in the engine.
pub enum DeviceGenerator {
    /// round(dt * rate), source_sink.rs:96-100
    Monotonic(f64),
    /// Poisson(dt * rate) from the engine's counter-based generator keyed by (seed, step)
    SeededPoisson(f64, u64),
    /// asked on the host every step
    HostCallback,
}

/// source_sink/source_sink.rs:30-33, plus `device_form`.
pub trait CrowdGenerator {
    /// Gets the number of pedestrians to spawn at a given time.
    fn get_number_to_spawn(&self, time_elapsed: Duration) -> usize;

    fn device_form(&self) -> DeviceGenerator {
        DeviceGenerator::HostCallback
    }
}

/// source_sink/source_sink.rs:36-60
pub struct SourceSink {
    pub source: Vec2f,
    pub radius_sink: f64,
    pub crowd_generator: Arc<dyn CrowdGenerator>,
    pub high_level_planner: Arc<Mutex<dyn HighLevelPlanner>>,
    pub local_planner: Arc<Mutex<dyn LocalPlanner>>,
    pub waypoints: Vec<Vec2f>,
    pub loop_forever: bool,
    pub agent_eyesight_range: f64,
}

/// source_sink/source_sink.rs:63-82 (unseeded thread_rng, as in the reference: asked on the host)
pub struct PoissonCrowd {
    pub rate: f64,
}

impl PoissonCrowd {
    pub fn new(rate: f64) -> Self {
        PoissonCrowd { rate }
    }
}

impl CrowdGenerator for PoissonCrowd {
    fn get_number_to_spawn(&self, time_elapsed: Duration) -> usize {
        let rt = time_elapsed.as_secs_f64() * self.rate;
        let mut rng = rand::thread_rng();
        let n = Poisson::new(rt).unwrap();
        n.sample(&mut rng) as usize
    }
}

/// The seedable stand-in (include/crowdstep.h CS_GEN_POISSON_SEEDED): reproducible streams.
pub struct SeededPoissonCrowd {
    pub rate: f64,
    pub seed: u64,
}

impl CrowdGenerator for SeededPoissonCrowd {
    fn get_number_to_spawn(&self, _time_elapsed: Duration) -> usize {
        unimplemented!("evaluated by the engine (counter-based generator keyed by seed and step)")
    }
    fn device_form(&self) -> DeviceGenerator {
        DeviceGenerator::SeededPoisson(self.rate, self.seed)
    }
}

/// source_sink/source_sink.rs:85-101
pub struct MonotonicCrowd {
    pub rate: f64,
}

impl MonotonicCrowd {
    pub fn new(rate: f64) -> Self {
        MonotonicCrowd { rate }
    }
}

impl CrowdGenerator for MonotonicCrowd {
    fn get_number_to_spawn(&self, time_elapsed: Duration) -> usize {
        let num_spawned = time_elapsed.as_secs_f64() * self.rate;
        num_spawned.round() as usize
    }
    fn device_form(&self) -> DeviceGenerator {
        DeviceGenerator::Monotonic(self.rate)
    }
}
