// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).
//
// `Simulation` cut into spatial tiles (one per GPU of a node): a thin wrapper over the C ABI's mesh handle
// (cs_mesh_*, include/crowdstep.h), where layout, halo exchange, spawn flags, route-cache misses, re-cuts and
// merged queries live.  The planners a tile mesh evaluates are the ones that describe themselves as data
// (`device_form`): every tile evaluates them for the agents it owns.  Host-callback planners stay with
// `Simulation` (one engine).
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::sync::{Arc, Mutex};

use crate::ffi;
use crate::highlevel_planners::{DeviceHighLevelPlan, HighLevelPlanner};
use crate::local_planners::{DeviceLocalPlan, LocalPlanner};
use crate::source_sink::{DeviceGenerator, SourceSink};
use crate::spatial_index::{LocationHash2D, SpatialIndex};
use crate::{Agent, AgentId, EventListener, Point, Vec2f};
use nalgebra::Vector2;

/// Where the tiles of a mesh live.
pub enum Placement {
    /// every tile in this process on one device (the single-GPU double of the multi-GPU logic)
    InProcess { device: i32 },
    /// one tile per rank and GPU; `unique_id` comes from `ffi::cs_rccl_unique_id` on one rank and is passed
    /// around by the host (MPI, a file, a socket)
    Distributed { device: i32, rank: i32, n_ranks: i32, unique_id: [u8; ffi::CS_RCCL_UNIQUE_ID_BYTES] },
    /// one tile per rank over a transport the host brings (`ffi::cs_mesh_host_transport`: three `extern "C"`
    /// functions over MPI, sockets, ...; host memory throughout), instead of RCCL.  The struct must outlive the mesh.
    HostTransport { device: i32, rank: i32, n_ranks: i32, transport: &'static ffi::cs_mesh_host_transport },
}

pub struct TiledSimulation {
    /// lib.rs:71.  Distributed: the whole crowd on every rank (the refresh is collective).
    pub agents: HashMap<AgentId, Agent>,
    mesh: *mut ffi::cs_mesh,
    hlp_handles: HashMap<usize, u32>,
    lp_handles: HashMap<usize, u32>,
    keep_lps: Vec<Box<Arc<Mutex<dyn LocalPlanner>>>>,  // what the mesh's `user` pointers of host planners point at
    listeners: Vec<Arc<Mutex<dyn EventListener>>>,
    keep_sinks: Vec<Arc<SourceSink>>,
}

impl TiledSimulation {
    pub fn new(index: &LocationHash2D, tiles_x: u32, tiles_y: u32, halo_cells: u32, placement: Placement) -> Result<Self, String> {
        let grid = index.device_form().expect("LocationHash2D describes itself as data");
        let mut desc = ffi::cs_mesh_desc {
            tiles_x,
            tiles_y,
            halo_cells,
            flags: 0,
            device_ordinal: 0,
            rank: 0,
            n_ranks: 1,
            density_per_cell: 16f64,
            capacity_hint: 0,
            weights_xy: std::ptr::null(),
            n_weights: 0,
            rccl_unique_id: std::ptr::null(),
            host_transport: std::ptr::null(),
        };
        let id_cell; // keeps the unique id alive across the call
        match placement {
            Placement::InProcess { device } => desc.device_ordinal = device,
            Placement::Distributed { device, rank, n_ranks, unique_id } => {
                desc.device_ordinal = device;
                desc.rank = rank;
                desc.n_ranks = n_ranks;
                id_cell = unique_id;
                desc.rccl_unique_id = id_cell.as_ptr();
            }
            Placement::HostTransport { device, rank, n_ranks, transport } => {
                desc.device_ordinal = device;
                desc.rank = rank;
                desc.n_ranks = n_ranks;
                desc.host_transport = transport as *const ffi::cs_mesh_host_transport;
            }
        }
        let mesh = unsafe { ffi::cs_mesh_create(&grid, &desc) };
        if mesh.is_null() {
            let why = unsafe { CStr::from_ptr(ffi::cs_mesh_last_error(std::ptr::null())) };
            return Err(why.to_string_lossy().into_owned());
        }
        Ok(TiledSimulation {
            agents: HashMap::new(),
            mesh,
            hlp_handles: HashMap::new(),
            lp_handles: HashMap::new(),
            keep_lps: Vec::new(),
            listeners: Vec::new(),
            keep_sinks: Vec::new(),
        })
    }

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(ffi::cs_mesh_last_error(self.mesh)) }.to_string_lossy().into_owned()
    }

    fn hlp_handle(&mut self, planner: &Arc<Mutex<dyn HighLevelPlanner>>) -> Result<u32, String> {
        let key = Arc::as_ptr(planner) as *const () as usize;
        if let Some(h) = self.hlp_handles.get(&key) {
            return Ok(*h);
        }
        let mut desc = ffi::cs_hlp_desc {
            kind: ffi::CS_HLP_NONE,
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
        match planner.lock().unwrap().device_form() {
            DeviceHighLevelPlan::None => {}
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
            _ => return Err("a tile mesh evaluates planners that describe themselves as data (device_form); host planners run on a single engine".to_string()),
        }
        let handle = unsafe { ffi::cs_mesh_register_hlp(self.mesh, &desc) };
        if handle == u32::MAX {
            return Err(self.last_error());
        }
        self.hlp_handles.insert(key, handle);
        Ok(handle)
    }

    fn lp_handle(&mut self, planner: &Arc<Mutex<dyn LocalPlanner>>) -> Result<u32, String> {
        let key = Arc::as_ptr(planner) as *const () as usize;
        if let Some(h) = self.lp_handles.get(&key) {
            return Ok(*h);
        }
        let handle = match planner.lock().unwrap().device_form() {
            DeviceLocalPlan::NoLocalPlan => unsafe { ffi::cs_mesh_register_no_local_plan(self.mesh) },
            DeviceLocalPlan::Zanlungo(params) => unsafe { ffi::cs_mesh_register_zanlungo(self.mesh, &params) },
            DeviceLocalPlan::HostCallback => {
                // every tile asks the planner for the agents it owns (cs_mesh_register_lp_callback); the mesh keeps
                // `user`: a heap cell holding a clone of the Arc, alive as long as `self`
                let cell = Box::new(planner.clone());
                let user = &*cell as *const Arc<Mutex<dyn LocalPlanner>> as *mut c_void;
                self.keep_lps.push(cell);
                unsafe { ffi::cs_mesh_register_lp_callback(self.mesh, Some(crate::lp_batch_trampoline), user) }
            }
        };
        if handle == u32::MAX {
            return Err(self.last_error());
        }
        self.lp_handles.insert(key, handle);
        Ok(handle)
    }

    fn after_mutation(&mut self) {
        let mut buf = vec![ffi::cs_event { kind: 0, source_sink: 0, id: 0, x: 0f64, y: 0f64 }; 4096];
        loop {
            let n = unsafe { ffi::cs_mesh_drain_events(self.mesh, buf.as_mut_ptr(), buf.len()) };
            for ev in &buf[..n] {
                for listener in &self.listeners {
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
        let n = unsafe { ffi::cs_mesh_agent_count(self.mesh) };
        let mut view = vec![ffi::cs_agent_view { id: 0, x: 0f64, y: 0f64, vx: 0f64, vy: 0f64, next_waypoint: 0, eyesight_range: 0f64 }; n.max(1)];
        let got = unsafe { ffi::cs_mesh_read_agents(self.mesh, view.as_mut_ptr(), n) };
        self.agents.clear();
        if got == usize::MAX {
            return;
        }
        for v in &view[..got] {
            self.agents.insert(
                v.id as usize,
                Agent {
                    agent_id: v.id as usize,
                    position: Point::new(v.x, v.y),
                    orientation: 0f64,
                    velocity: Vector2::new(v.vx, v.vy),
                    angular_vel: 0f64,
                    next_waypoint: v.next_waypoint as usize,
                    eyesight_range: v.eyesight_range,
                },
            );
        }
    }

    /// lib.rs:119-156 (every rank of a distributed mesh makes the same call)
    pub fn add_agents(&mut self, spawn_positions: &Vec<Point>, high_level_planner: Arc<Mutex<dyn HighLevelPlanner>>,
                      local_planner: Arc<Mutex<dyn LocalPlanner>>, agent_eyesight_range: f64) -> Result<Vec<AgentId>, String> {
        let hlp = self.hlp_handle(&high_level_planner)?;
        let lp = self.lp_handle(&local_planner)?;
        let xy: Vec<f64> = spawn_positions.iter().flat_map(|p| [p.x, p.y]).collect();
        let mut ids = vec![0u64; spawn_positions.len()];
        let rc = unsafe { ffi::cs_mesh_add_agents(self.mesh, xy.as_ptr(), ids.len(), hlp, lp, agent_eyesight_range, ids.as_mut_ptr()) };
        self.after_mutation();
        if rc != 0 {
            return Err(self.last_error());
        }
        Ok(ids.into_iter().map(|i| i as usize).collect())
    }

    /// lib.rs:159-161.  Generators must be deterministic across tiles (MonotonicCrowd, SeededPoissonCrowd).
    pub fn add_source_sink(&mut self, source_sink: Arc<SourceSink>) -> Result<usize, String> {
        let hlp = self.hlp_handle(&source_sink.high_level_planner)?;
        let lp = self.lp_handle(&source_sink.local_planner)?;
        let waypoints: Vec<f64> = source_sink.waypoints.iter().flat_map(|p| [p.x, p.y]).collect();
        let (generator_kind, rate, seed) = match source_sink.crowd_generator.device_form() {
            DeviceGenerator::Monotonic(rate) => (ffi::CS_GEN_MONOTONIC, rate, 0u64),
            DeviceGenerator::SeededPoisson(rate, seed) => (ffi::CS_GEN_POISSON_SEEDED, rate, seed),
            DeviceGenerator::HostCallback => return Err("a tile mesh needs generators that are the same on every tile (MonotonicCrowd, SeededPoissonCrowd)".to_string()),
        };
        let desc = ffi::cs_source_sink_desc {
            source_x: source_sink.source.x,
            source_y: source_sink.source.y,
            radius_sink: source_sink.radius_sink,
            generator_kind,
            rate,
            seed,
            generator: None,
            generator_user: std::ptr::null_mut(),
            hlp,
            lp,
            waypoints_xy: waypoints.as_ptr(),
            n_waypoints: source_sink.waypoints.len(),
            loop_forever: if source_sink.loop_forever { 1 } else { 0 },
            agent_eyesight_range: source_sink.agent_eyesight_range,
        };
        let handle = unsafe { ffi::cs_mesh_add_source_sink(self.mesh, &desc) };
        if handle == u32::MAX {
            return Err(self.last_error());
        }
        self.keep_sinks.push(source_sink);
        Ok(handle as usize)
    }

    /// lib.rs:171-173
    pub fn add_event_listener(&mut self, event_listener: Arc<Mutex<dyn EventListener>>) -> usize {
        self.listeners.push(event_listener);
        unsafe { ffi::cs_mesh_event_recording(self.mesh, 1) };
        self.listeners.len() - 1
    }

    /// lib.rs:176-192 (Err for an id nobody holds, where the reference panics)
    pub fn remove_agents(&mut self, agent: AgentId) -> Result<(), String> {
        let rc = unsafe { ffi::cs_mesh_remove_agent(self.mesh, agent as u64) };
        self.after_mutation();
        if rc != 0 { Err(self.last_error()) } else { Ok(()) }
    }

    /// lib.rs:195-383 on every tile; collective on a distributed mesh
    pub fn step(&mut self, dur: std::time::Duration) -> Result<(), String> {
        let mut report = ffi::cs_step_report { n_agents: 0, n_spawned: 0, n_destroyed: 0, n_waypoint_hits: 0, n_tti_zero: 0, n_nonfinite: 0, n_clamped: 0 };
        let rc = unsafe { ffi::cs_mesh_step(self.mesh, dur.as_secs_f64(), &mut report) };
        self.after_mutation();
        if rc != 0 { Err(self.last_error()) } else { Ok(()) }
    }

    /// Cuts to the quantiles of where the crowd stands now (in-process meshes).
    pub fn recut(&mut self) -> Result<(), String> {
        if unsafe { ffi::cs_mesh_recut(self.mesh) } != 0 { Err(self.last_error()) } else { Ok(()) }
    }
}

impl Drop for TiledSimulation {
    fn drop(&mut self) {
        unsafe { ffi::cs_mesh_destroy(self.mesh) };
    }
}
