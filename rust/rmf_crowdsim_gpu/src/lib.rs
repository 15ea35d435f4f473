// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).  The FFI declarations in
// ffi.rs are checked mechanically against include/crowdstep.h (tools/check_ffi_layout.py); this
// file is the host side a maintainer builds where `cargo` exists.
//
//
// Attribution: the trait and struct declarations in this file restate, signature for signature, those of
// open-rmf/rmf_crowdsim (Copyright (C) 2022 Open Source Robotics Foundation, licensed under the Apache License,
// Version 2.0: http://www.apache.org/licenses/LICENSE-2.0), file rmf_crowdsim/src/lib.rs (EventListener, Agent, the type aliases and the public methods of Simulation); they are kept
// identical on purpose, so that this crate drops in for that one.  Everything else in the file is original.
//! `rmf_crowdsim`'s public surface over the MI355X crowd-step engine.
//!
//! Same items, names, argument order and error behaviour as the reference crate
//! (paths relative to rmf_crowdsim/src of open-rmf/rmf_crowdsim):
//!
//! | here | reference |
//! |---|---|
//! | `Simulation<T: SpatialIndex>`: `new`, `add_agents`, `add_source_sink`, `remove_source_sink`, `add_event_listener`, `remove_agents`, `step`, `pub agents` | lib.rs:69-383 |
//! | `EventListener` | lib.rs:22-33 |
//! | `Agent`, `AgentId`, `Point`, `Vec2f` | lib.rs:36-65 |
//! | `HighLevelPlanner` | highlevel_planners/highlevel_planners.rs:8-16 |
//! | `LocalPlanner`, `Zanlungo`, `NoLocalPlan` | local_planners/*.rs |
//! | `SpatialIndex`, `LocationHash2D` | spatial_index/*.rs |
//! | `SourceSink`, `CrowdGenerator`, `MonotonicCrowd`, `PoissonCrowd` | source_sink/source_sink.rs |
//!
//! What differs, and why: trait objects cannot run on a GPU, so each trait gains ONE provided
//! method (`device_form`) through which the planners the reference ships describe themselves as
//! data; every other implementation keeps working through the documented slow paths
//! (`HighLevelPlanner`: a batched host callback per step; `LocalPlanner`: a batched host callback too,
//! `cs_register_lp_callback`: the engine hands over every agent of the planner with its neighbours in
//! canonical order, from the device's batch radius query, and takes the velocities back before it
//! integrates).  The step is the canonical
//! (Jacobi) member of the reference's order-dependent family: DESIGN.md section 2.

pub extern crate nalgebra as na;
use na::Vector2;

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::sync::{Arc, Mutex};

pub mod ffi;
pub mod highlevel_planners;
pub mod local_planners;
pub mod source_sink;
pub mod spatial_index;
pub mod tiled;

pub use crate::highlevel_planners::{DeviceHighLevelPlan, HighLevelPlanner};
pub use crate::local_planners::{DeviceLocalPlan, LocalPlanner, NoLocalPlan, Zanlungo};
pub use crate::source_sink::{CrowdGenerator, MonotonicCrowd, PoissonCrowd, SeededPoissonCrowd, SourceSink};
pub use crate::spatial_index::{LocationHash2D, SpatialIndex};
pub use crate::tiled::{Placement, TiledSimulation};

/// lib.rs:22-33
pub trait EventListener {
    fn agent_spawned(&mut self, position: Vec2f, agent: AgentId);
    fn agent_destroyed(&mut self, agent: AgentId);
    /// Declared by the reference, never called (lib.rs:32).
    fn waypoint_reached(&mut self, _position: Vec2f, _agent: AgentId) {}
}

/// lib.rs:36
pub type AgentId = usize;
/// lib.rs:40
pub type Point = Vector2<f64>;
/// lib.rs:43
pub type Vec2f = Vector2<f64>;

/// lib.rs:46-65.  `preferred_vel` is private in the reference and never committed (SURVEY.md
/// section 8a row a1), so it does not exist here.
#[derive(Clone, Copy, Debug)]
pub struct Agent {
    pub agent_id: AgentId,
    pub position: Point,
    pub orientation: f64,
    pub velocity: Vector2<f64>,
    pub angular_vel: f64,
    pub next_waypoint: usize,
    pub eyesight_range: f64,
}

/// What one step did (the reference prints or drops these).
pub type StepReport = ffi::cs_step_report;

/// lib.rs:69-91.  The state lives in HBM; `agents` is the host view, refreshed after every
/// `step` / `add_agents` / `remove_agents` unless `set_agents_view(false)` turned that off (a
/// renderer then uses `request_snapshot` / `snapshot`).
pub struct Simulation<T: SpatialIndex> {
    pub agents: HashMap<AgentId, Agent>,
    engine: *mut ffi::cs_engine,
    spatial_index: T,
    event_listeners: Vec<(usize, Arc<Mutex<dyn EventListener>>)>,
    next_listener: usize,
    // planners registered with the engine, by the address of their Arc (one handle per object)
    hlp_handles: HashMap<usize, u32>,
    lp_handles: HashMap<usize, u32>,
    // everything the engine holds raw pointers into stays alive as long as the engine
    keep_hlps: Vec<Box<Arc<Mutex<dyn HighLevelPlanner>>>>,
    keep_lps: Vec<Box<Arc<Mutex<dyn LocalPlanner>>>>,
    keep_generators: Vec<Box<Arc<dyn CrowdGenerator>>>,
    keep_sinks: Vec<Arc<SourceSink>>,
    agents_view: bool,
    pub last_report: StepReport,
}

// trampolines: the engine calls these on the caller's thread, inside cs_step / cs_add_agents
unsafe extern "C" fn hlp_velocity_trampoline(user: *mut c_void, n: usize, ids: *const u64, pos_xy: *const f64,
                                             vel_xy: *const f64, time_s: f64, out_vel_xy: *mut f64, out_some: *mut u8) {
    let planner = &*(user as *const Arc<Mutex<dyn HighLevelPlanner>>);
    let mut p = planner.lock().unwrap();
    for i in 0..n {
        let agent = Agent {
            agent_id: *ids.add(i) as usize,
            position: Point::new(*pos_xy.add(2 * i), *pos_xy.add(2 * i + 1)),
            orientation: 0f64,
            velocity: Vector2::new(*vel_xy.add(2 * i), *vel_xy.add(2 * i + 1)),
            angular_vel: 0f64,
            next_waypoint: 0,
            eyesight_range: 0f64,
        };
        match p.get_desired_velocity(&agent, std::time::Duration::from_secs_f64(time_s)) {
            Some(v) => {
                *out_some.add(i) = 1;
                *out_vel_xy.add(2 * i) = v.x;
                *out_vel_xy.add(2 * i + 1) = v.y;
            }
            None => *out_some.add(i) = 0,
        }
    }
}

/// cs_lp_batch_fn: `LocalPlanner::get_desired_velocity` for every agent of one host planner (lib.rs:276-291)
pub(crate) unsafe extern "C" fn lp_batch_trampoline(user: *mut c_void, n_agents: usize, agents: *const ffi::cs_lp_agent,
                                         recommended_xy: *const f64, nb_begin: *const u64,
                                         neighbours: *const ffi::cs_lp_agent, out_velocity_xy: *mut f64) -> std::os::raw::c_int {
    let planner = &*(user as *const Arc<Mutex<dyn LocalPlanner>>);
    // (a panic must not unwind through the C frame: a poisoned lock fails the step instead, with Err from `step`)
    let p = match planner.lock() {
        Ok(p) => p,
        Err(_) => return 1,
    };
    let view = |r: &ffi::cs_lp_agent| Agent {
        agent_id: r.agent_id as usize,
        position: Point::new(r.x, r.y),
        orientation: 0f64,
        velocity: Vector2::new(r.vx, r.vy),
        angular_vel: 0f64,
        next_waypoint: r.next_waypoint as usize,
        eyesight_range: r.eyesight_range,
    };
    // A panic inside the user's planner must not unwind across `extern "C"` either (an abort on current toolchains,
    // undefined behaviour on older ones): it is caught here and fails the step ("a host LocalPlanner failed"), as the
    // C++ (catch ...) and Python (try / except) thunks do.
    let outcome = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
        for k in 0..n_agents {
            let (b, e) = (*nb_begin.add(k) as usize, *nb_begin.add(k + 1) as usize);
            let nearby: Vec<Agent> = (b..e).map(|q| view(&*neighbours.add(q))).collect();
            let recommended = Vector2::new(*recommended_xy.add(2 * k), *recommended_xy.add(2 * k + 1));
            let v = p.get_desired_velocity(&view(&*agents.add(k)), &nearby, recommended);
            *out_velocity_xy.add(2 * k) = v.x;
            *out_velocity_xy.add(2 * k + 1) = v.y;
        }
    }));
    match outcome {
        Ok(()) => 0,
        Err(_) => 1,
    }
}

unsafe extern "C" fn hlp_set_target_trampoline(user: *mut c_void, id: u64, pos_x: f64, pos_y: f64, point_x: f64,
                                               point_y: f64, tol_x: f64, tol_y: f64) {
    let planner = &*(user as *const Arc<Mutex<dyn HighLevelPlanner>>);
    let agent = Agent {
        agent_id: id as usize,
        position: Point::new(pos_x, pos_y),
        orientation: 0f64,
        velocity: Vector2::new(0f64, 0f64),
        angular_vel: 0f64,
        next_waypoint: 0,
        eyesight_range: 0f64,
    };
    planner.lock().unwrap().set_target(&agent, Vec2f::new(point_x, point_y), Vec2f::new(tol_x, tol_y));
}

unsafe extern "C" fn hlp_remove_trampoline(user: *mut c_void, id: u64) {
    let planner = &*(user as *const Arc<Mutex<dyn HighLevelPlanner>>);
    planner.lock().unwrap().remove_agent_id(id as usize);
}

unsafe extern "C" fn route_plan_trampoline(user: *mut c_void, start_x: f64, start_y: f64, goal_x: f64, goal_y: f64,
                                           out_xy: *mut f64, cap: usize) -> usize {
    let planner = &*(user as *const Arc<Mutex<dyn HighLevelPlanner>>);
    let route = planner.lock().unwrap().plan_route(Vec2f::new(start_x, start_y), Vec2f::new(goal_x, goal_y));
    match route {
        Some(points) => {
            let n = points.len().min(cap);
            for (k, p) in points.iter().take(n).enumerate() {
                *out_xy.add(2 * k) = p.x;
                *out_xy.add(2 * k + 1) = p.y;
            }
            n
        }
        None => 0,
    }
}

unsafe extern "C" fn generator_trampoline(user: *mut c_void, dt_seconds: f64) -> usize {
    let generator = &*(user as *const Arc<dyn CrowdGenerator>);
    generator.get_number_to_spawn(std::time::Duration::from_secs_f64(dt_seconds))
}

impl<T: SpatialIndex> Simulation<T> {
    /// lib.rs:103.  Panics when the index is not device-evaluable or no MI355X is visible: the
    /// engine has no CPU path (`cs_last_error(null)` says why).
    pub fn new(spatial_index: T) -> Self {
        let grid = spatial_index
            .device_form()
            .expect("the GPU backend needs a LocationHash2D (SpatialIndex::device_form)");
        let engine = unsafe { ffi::cs_create(&grid, std::ptr::null()) };
        if engine.is_null() {
            let why = unsafe { CStr::from_ptr(ffi::cs_last_error(std::ptr::null())) };
            panic!("cs_create failed: {}", why.to_string_lossy());
        }
        assert_eq!(unsafe { ffi::cs_abi_version() }, ffi::CS_ABI_VERSION);
        unsafe { ffi::cs_event_recording(engine, 0) }; // no listeners yet (lib.rs:88)
        Self {
            agents: HashMap::new(),
            engine,
            spatial_index,
            event_listeners: Vec::new(),
            next_listener: 0,
            hlp_handles: HashMap::new(),
            lp_handles: HashMap::new(),
            keep_hlps: Vec::new(),
            keep_lps: Vec::new(),
            keep_generators: Vec::new(),
            keep_sinks: Vec::new(),
            agents_view: true,
            last_report: StepReport::default(),
        }
    }

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(ffi::cs_last_error(self.engine)) }.to_string_lossy().into_owned()
    }

    fn hlp_handle(&mut self, planner: &Arc<Mutex<dyn HighLevelPlanner>>) -> Result<u32, String> {
        let key = Arc::as_ptr(planner) as *const () as usize;
        if let Some(h) = self.hlp_handles.get(&key) {
            return Ok(*h);
        }
        let form = planner.lock().unwrap().device_form();
        let mut desc = ffi::cs_hlp_desc {
            kind: ffi::CS_HLP_CALLBACK,
            vx: 0f64,
            vy: 0f64,
            velocity: None,
            set_target: None,
            remove_agent: None,
            user: std::ptr::null_mut(),
            route_plan: None,
            route_scale: 0f64,
            route_arrive: 0f64,
            route_speed: 0f64,
        };
        // the engine keeps `user`: a heap cell holding a clone of the Arc, alive as long as `self`
        let cell = Box::new(planner.clone());
        let user = &*cell as *const Arc<Mutex<dyn HighLevelPlanner>> as *mut c_void;
        match form {
            DeviceHighLevelPlan::None => desc.kind = ffi::CS_HLP_NONE,
            DeviceHighLevelPlan::Constant(v) => {
                desc.kind = ffi::CS_HLP_CONSTANT;
                desc.vx = v.x;
                desc.vy = v.y;
            }
            DeviceHighLevelPlan::IdParity(v) => {
                desc.kind = ffi::CS_HLP_ID_PARITY;
                desc.vx = v.x;
                desc.vy = v.y;
            }
            DeviceHighLevelPlan::Route { scale, arrive, speed } => {
                desc.kind = ffi::CS_HLP_ROUTE;
                desc.user = user;
                desc.route_plan = Some(route_plan_trampoline);
                desc.route_scale = scale;
                desc.route_arrive = arrive;
                desc.route_speed = speed;
            }
            DeviceHighLevelPlan::HostCallback => {
                desc.user = user;
                desc.velocity = Some(hlp_velocity_trampoline);
                desc.set_target = Some(hlp_set_target_trampoline);
                desc.remove_agent = Some(hlp_remove_trampoline);
            }
        }
        let handle = unsafe { ffi::cs_register_hlp(self.engine, &desc) };
        if handle == u32::MAX {
            return Err(self.last_error());
        }
        self.keep_hlps.push(cell);
        self.hlp_handles.insert(key, handle);
        Ok(handle)
    }

    fn lp_handle(&mut self, planner: &Arc<Mutex<dyn LocalPlanner>>) -> Result<u32, String> {
        let key = Arc::as_ptr(planner) as *const () as usize;
        if let Some(h) = self.lp_handles.get(&key) {
            return Ok(*h);
        }
        let handle = match planner.lock().unwrap().device_form() {
            DeviceLocalPlan::NoLocalPlan => unsafe { ffi::cs_register_no_local_plan(self.engine) },
            DeviceLocalPlan::Zanlungo(params) => unsafe { ffi::cs_register_zanlungo(self.engine, &params) },
            DeviceLocalPlan::HostCallback => {
                // the engine keeps `user`: a heap cell holding a clone of the Arc, alive as long as `self`
                let cell = Box::new(planner.clone());
                let user = &*cell as *const Arc<Mutex<dyn LocalPlanner>> as *mut c_void;
                self.keep_lps.push(cell);
                unsafe { ffi::cs_register_lp_callback(self.engine, Some(lp_batch_trampoline), user) }
            }
        };
        if handle == u32::MAX {
            return Err(self.last_error());
        }
        self.lp_handles.insert(key, handle);
        Ok(handle)
    }

    /// agent_spawned / agent_destroyed in the order the reference would have called them
    /// (spawns in sink order during Phase A, removals in ascending id after the commit).
    fn dispatch_events(&mut self) {
        let mut buf = vec![ffi::cs_event { kind: 0, source_sink: 0, id: 0, x: 0f64, y: 0f64 }; 4096];
        loop {
            let n = unsafe { ffi::cs_drain_events(self.engine, buf.as_mut_ptr(), buf.len()) };
            for ev in &buf[..n] {
                for (_, listener) in &self.event_listeners {
                    let mut l = listener.lock().unwrap();
                    if ev.kind == ffi::CS_EVENT_SPAWNED {
                        l.agent_spawned(Vec2f::new(ev.x, ev.y), ev.id as usize);
                    } else if ev.kind == ffi::CS_EVENT_DESTROYED {
                        l.agent_destroyed(ev.id as usize);
                    }
                }
            }
            if n < buf.len() {
                break;
            }
        }
    }

    /// `pub agents` (lib.rs:71) from the device state.
    fn refresh_agents(&mut self) {
        if !self.agents_view {
            return;
        }
        let n = unsafe { ffi::cs_agent_count(self.engine) };
        let mut buf = vec![
            ffi::cs_agent_view { id: 0, x: 0f64, y: 0f64, vx: 0f64, vy: 0f64, next_waypoint: 0, eyesight_range: 0f64 };
            n.max(1)
        ];
        let got = unsafe { ffi::cs_read_agents(self.engine, buf.as_mut_ptr(), n) };
        self.agents.clear();
        for v in &buf[..got] {
            self.agents.insert(
                v.id as usize,
                Agent {
                    agent_id: v.id as usize,
                    position: Point::new(v.x, v.y),
                    orientation: 0f64, // never written after creation (lib.rs:138)
                    velocity: Vector2::new(v.vx, v.vy),
                    angular_vel: 0f64, // never written after creation (lib.rs:141)
                    next_waypoint: v.next_waypoint as usize,
                    eyesight_range: v.eyesight_range,
                },
            );
        }
    }

    /// Keep (default) or stop keeping `agents` in sync after every call; with 1M agents the
    /// read-back is what a frame costs, and a renderer wants `snapshot`.
    pub fn set_agents_view(&mut self, on: bool) {
        self.agents_view = on;
        if on {
            self.refresh_agents();
        }
    }

    /// lib.rs:119-156
    pub fn add_agents(
        &mut self,
        spawn_positions: &Vec<Point>,
        high_level_planner: Arc<Mutex<dyn HighLevelPlanner>>,
        local_planner: Arc<Mutex<dyn LocalPlanner>>,
        agent_eyesight_range: f64,
    ) -> Result<Vec<AgentId>, String> {
        let hlp = self.hlp_handle(&high_level_planner)?;
        let lp = self.lp_handle(&local_planner)?;
        let xy: Vec<f64> = spawn_positions.iter().flat_map(|p| [p.x, p.y]).collect();
        let mut ids = vec![0u64; spawn_positions.len()];
        let rc = unsafe {
            ffi::cs_add_agents(self.engine, xy.as_ptr(), ids.len(), hlp, lp, agent_eyesight_range, ids.as_mut_ptr())
        };
        self.dispatch_events();
        self.refresh_agents();
        if rc != 0 {
            return Err(self.last_error()); // "Index out of bounds"
        }
        Ok(ids.into_iter().map(|i| i as usize).collect())
    }

    /// lib.rs:159-161
    pub fn add_source_sink(&mut self, source_sink: Arc<SourceSink>) -> usize {
        let hlp = self.hlp_handle(&source_sink.high_level_planner).expect("high-level planner refused");
        let lp = self.lp_handle(&source_sink.local_planner).expect("local planner refused");
        let waypoints: Vec<f64> = source_sink.waypoints.iter().flat_map(|p| [p.x, p.y]).collect();
        let mut desc = ffi::cs_source_sink_desc {
            source_x: source_sink.source.x,
            source_y: source_sink.source.y,
            radius_sink: source_sink.radius_sink,
            generator_kind: ffi::CS_GEN_CALLBACK,
            rate: 0f64,
            seed: 0,
            generator: None,
            generator_user: std::ptr::null_mut(),
            hlp,
            lp,
            waypoints_xy: waypoints.as_ptr(), // copied by the engine during the call
            n_waypoints: source_sink.waypoints.len(),
            loop_forever: if source_sink.loop_forever { 1 } else { 0 },
            agent_eyesight_range: source_sink.agent_eyesight_range,
        };
        match source_sink.crowd_generator.device_form() {
            source_sink::DeviceGenerator::Monotonic(rate) => {
                desc.generator_kind = ffi::CS_GEN_MONOTONIC;
                desc.rate = rate;
            }
            source_sink::DeviceGenerator::SeededPoisson(rate, seed) => {
                desc.generator_kind = ffi::CS_GEN_POISSON_SEEDED;
                desc.rate = rate;
                desc.seed = seed;
            }
            source_sink::DeviceGenerator::HostCallback => {
                let cell = Box::new(source_sink.crowd_generator.clone());
                desc.generator_user = &*cell as *const Arc<dyn CrowdGenerator> as *mut c_void;
                desc.generator = Some(generator_trampoline);
                self.keep_generators.push(cell);
            }
        }
        let handle = unsafe { ffi::cs_add_source_sink(self.engine, &desc) };
        assert!(handle != u32::MAX, "{}", self.last_error());
        self.keep_sinks.push(source_sink);
        handle as usize
    }

    /// lib.rs:164-168
    pub fn remove_source_sink(&mut self, id: &usize) {
        unsafe { ffi::cs_remove_source_sink(self.engine, *id as u32) };
    }

    /// lib.rs:171-173
    pub fn add_event_listener(&mut self, event_listener: Arc<Mutex<dyn EventListener>>) -> usize {
        let id = self.next_listener;
        self.next_listener += 1;
        self.event_listeners.push((id, event_listener));
        unsafe { ffi::cs_event_recording(self.engine, 1) };
        id
    }

    /// lib.rs:176-192 (the reference panics on an unknown id: so does this)
    pub fn remove_agents(&mut self, agent: AgentId) {
        let rc = unsafe { ffi::cs_remove_agent(self.engine, agent as u64) };
        assert!(rc == 0, "{}", self.last_error());
        self.dispatch_events();
        self.refresh_agents();
    }

    /// lib.rs:195-383
    pub fn step(&mut self, dur: std::time::Duration) -> Result<(), String> {
        let mut report = StepReport::default();
        // a report makes the call wait for the device, so Err("Index out of bounds") belongs to
        // THIS step, as in the reference
        let rc = unsafe { ffi::cs_step(self.engine, dur.as_secs_f64(), &mut report) };
        self.last_report = report;
        self.dispatch_events();
        self.refresh_agents();
        if rc != 0 {
            Err(self.last_error())
        } else {
            Ok(())
        }
    }

    /// Fire-and-forget form for hosts that read no state between steps (no listeners, no host
    /// planners): returns before the device has finished; the Err of such a step comes back from
    /// `synchronize` or from the next `step`.
    pub fn step_unsynced(&mut self, dur: std::time::Duration) -> Result<(), String> {
        let rc = unsafe { ffi::cs_step(self.engine, dur.as_secs_f64(), std::ptr::null_mut()) };
        if rc != 0 {
            Err(self.last_error())
        } else {
            Ok(())
        }
    }

    pub fn synchronize(&mut self) -> Result<(), String> {
        if unsafe { ffi::cs_synchronize(self.engine) } != 0 {
            return Err(self.last_error());
        }
        self.refresh_agents();
        Ok(())
    }

    /// SpatialIndex::get_neighbours_in_radius on the device index (location_hash_2d.rs:240-258)
    pub fn get_neighbours_in_radius(&mut self, radius: f64, position: Point) -> Vec<AgentId> {
        let mut cap = 256usize;
        loop {
            let mut out = vec![0u64; cap];
            let n = unsafe { ffi::cs_query_radius(self.engine, radius, position.x, position.y, out.as_mut_ptr(), cap) };
            if n <= cap {
                out.truncate(n);
                return out.into_iter().map(|i| i as usize).collect();
            }
            cap = n;
        }
    }

    /// SpatialIndex::get_nearest_neighbours on the device index (location_hash_2d.rs:151-238; exact)
    pub fn get_nearest_neighbours(&mut self, n: usize, position: Point) -> Vec<AgentId> {
        let mut out = vec![0u64; n.max(1)];
        let got = unsafe { ffi::cs_query_knn(self.engine, n, position.x, position.y, out.as_mut_ptr()) };
        out.truncate(got);
        out.into_iter().map(|i| i as usize).collect()
    }

    /// Streaming `agents` view for a renderer (rmf_crowdsim_viz/src/main.rs:112-128): queue a frame
    /// behind the steps queued so far; the next step overlaps the transfer.
    pub fn request_snapshot(&mut self) -> Result<(), String> {
        if unsafe { ffi::cs_snapshot_request(self.engine) } != 0 {
            return Err(self.last_error());
        }
        Ok(())
    }

    /// The most recently requested frame (valid until the second `request_snapshot` from now) and
    /// the number of steps it was taken after; None while it is still in flight (`wait == false`).
    pub fn snapshot(&mut self, wait: bool) -> Option<(&[ffi::cs_snapshot_record], u64)> {
        let mut out: *const ffi::cs_snapshot_record = std::ptr::null();
        let (mut n, mut step) = (0usize, 0u64);
        let rc = unsafe { ffi::cs_snapshot_acquire(self.engine, if wait { 1 } else { 0 }, &mut out, &mut n, &mut step) };
        if rc != 0 || out.is_null() {
            return None;
        }
        Some((unsafe { std::slice::from_raw_parts(out, n) }, step))
    }

    pub fn spatial_index(&self) -> &T {
        &self.spatial_index
    }
}

impl<T: SpatialIndex> Drop for Simulation<T> {
    fn drop(&mut self) {
        unsafe { ffi::cs_destroy(self.engine) };
    }
}
