// NEVER COMPILED (no Rust toolchain in the build image; see Cargo.toml).
//
// Attribution: the trait and struct declarations in this file restate, signature for signature, those of
// open-rmf/rmf_crowdsim (Copyright (C) 2022 Open Source Robotics Foundation, licensed under the Apache License,
// Version 2.0: http://www.apache.org/licenses/LICENSE-2.0), file rmf_crowdsim/src/spatial_index/{spatial_index,location_hash_2d}.rs; they are kept
// identical on purpose, so that this crate drops in for that one.  Everything else in the file is original.
use crate::ffi;
use crate::{AgentId, Point};

/// spatial_index/spatial_index.rs:4-14, plus `device_form`.  The index itself lives on the device
/// (cell-sorted arrays rebuilt every step); `Simulation::get_neighbours_in_radius` /
/// `get_nearest_neighbours` query it there.
pub trait SpatialIndex {
    fn add_or_update(&mut self, index: AgentId, position: Point) -> Result<(), String>;

    fn get_nearest_neighbours(&self, n: usize, position: Point) -> Vec<AgentId>;

    fn get_neighbours_in_radius(&self, radius: f64, position: Point) -> Vec<AgentId>;

    fn remove_agent(&mut self, _agent: AgentId) {
        // Do Nothing
    }

    /// The grid the engine should build, for index types it can run (LocationHash2D).
    fn device_form(&self) -> Option<ffi::cs_grid_desc> {
        None
    }
}

/// spatial_index/location_hash_2d.rs:14-51: the constructor arguments, handed to the engine.
pub struct LocationHash2D {
    width: f64,
    height: f64,
    resolution: f64,
    offset: Point,
}

impl LocationHash2D {
    pub fn new(width: f64, height: f64, cell_size: f64, offset: Point) -> Self {
        Self { width, height, resolution: cell_size, offset }
    }
}

impl SpatialIndex for LocationHash2D {
    fn add_or_update(&mut self, _index: AgentId, _position: Point) -> Result<(), String> {
        Err("the index lives on the device: agents enter it through Simulation::add_agents".to_string())
    }
    fn get_nearest_neighbours(&self, _n: usize, _position: Point) -> Vec<AgentId> {
        unimplemented!("ask the Simulation: the index lives on the device")
    }
    fn get_neighbours_in_radius(&self, _radius: f64, _position: Point) -> Vec<AgentId> {
        unimplemented!("ask the Simulation: the index lives on the device")
    }
    fn device_form(&self) -> Option<ffi::cs_grid_desc> {
        Some(ffi::cs_grid_desc {
            width: self.width,
            height: self.height,
            cell_size: self.resolution,
            offset_x: self.offset.x,
            offset_y: self.offset.y,
        })
    }
}
