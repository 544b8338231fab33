//! FFI declarations for include/wgsparkl_hip.h — SOURCE ONLY: this repository's build image has no
//! Rust toolchain, so this file is not compiled or tested here (the ctypes twin wgsparkl_amd/_ffi.py is,
//! by tests/test_capi_abi.py). See INTEGRATION.md §3 for the safe wrapper that keeps wgsparkl's
//! `MpmPipeline` / `MpmData` surface.
//! build.rs: println!("cargo:rustc-link-lib=dylib=wgsparkl3d_hip");   (feature dim3; wgsparkl2d_hip for dim2)
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const DIM: usize = if cfg!(feature = "dim2") { 2 } else { 3 };
pub type wgs_status = i32;
pub const WGS_OK: wgs_status = 0;
pub const WGS_MAX_COLLIDERS: usize = 16;
pub const WGS_NUM_PASSES: usize = 10;
pub enum wgs_pipeline {}
pub enum wgs_data {}

#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_sim_params {
    pub gravity: [f32; DIM],
    pub dt: f32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_cdf {
    pub normal: [f32; DIM],
    pub rigid_vel: [f32; DIM],
    pub signed_distance: f32,
    pub affinity: u32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_particle_dynamics {
    pub velocity: [f32; DIM],
    pub def_grad: [f32; DIM * DIM],
    pub affine: [f32; DIM * DIM],
    pub cdf: wgs_cdf,
    pub init_volume: f32,
    pub init_radius: f32,
    pub mass: f32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_particle {
    pub position: [f32; DIM],
    pub dynamics: wgs_particle_dynamics,
    pub model: [f32; 2], // ElasticCoefficients { lambda, mu }
    pub has_plasticity: u32,
    pub plasticity: [f32; 6], // DruckerPrager { h0, h1, h2, h3, lambda, mu }
    pub has_phase: u32,
    pub phase: [f32; 2], // ParticlePhase { phase, max_stretch }
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_plastic_state {
    pub plastic_deformation_gradient_det: f32,
    pub plastic_hardening: f32,
    pub log_vol_gain: f32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_pose {
    pub rotation: [f32; 4], // 3D: unit quaternion (i, j, k, w); 2D: (cos, sin, 0, 0)
    pub translation: [f32; 3],
    pub scale: f32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_velocity {
    pub linear: [f32; 3],
    pub angular: [f32; 3], // 2D: angular[0]
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_collider {
    pub shape_type: u32, // 0 ball, 1 cuboid, 2 capsule
    pub shape: [f32; 4],
    pub pose: wgs_pose,
    pub velocity: wgs_velocity,
    pub com: [f32; 3], // world-space centre of mass
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_mass_properties {
    pub inv_mass: [f32; 3],
    pub inv_inertia_local: [f32; 9], // 3D: column-major, body frame; 2D: [0] = 1 / I
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_sample_ids {
    // = GpuSampleIds (src/solver/particle3d.rs:62-66); 2D: vertex[0..2] is the segment
    pub vertex: [u32; 3],
    pub collider: u32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_instance {
    // = src_testbed/instancing3d.rs InstanceData
    pub deformation: [[f32; 4]; 3],
    pub position: [f32; 4],
    pub base_color: [f32; 4],
    pub color: [f32; 4],
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_node_record {
    pub cell: [i32; DIM],
    pub velocity: [f32; DIM],
    pub mass: f32,
    pub cdf_distance: f32,
    pub cdf_affinities: u32,
    pub cdf_closest_id: u32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_block_record {
    pub virtual_id: [i32; DIM],
    pub first_particle: u32,
    pub num_particles: u32,
}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_stats {
    pub num_particles: u32,
    pub num_active_blocks: u32,
    pub grid_capacity: u32,
    pub overflow: u32,
    pub substeps_done: u64,
    pub device_bytes: u64,
    pub num_near_collider_blocks: u32,
    pub grid_growths: u32,
    pub cell_changers: u64,
    pub table_rebuilds: u64,
    pub block_ids: u32,
    pub block_ids_free: u32,
    pub table_marks: u32,
    pub table_refreshes: u32,
}

/// Optional interop view of the particle state on the device (`wgs_get_device_ptrs`).
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_device_ptrs {
    pub position_quads: *const f32,
    pub particle_ids: *const u32,
    pub count: u32,
    pub capacity: u32,
    pub dim: u32,
    pub reserved: u32,
    pub hip_stream: *mut c_void,
}

#[repr(C)]
pub struct wgs_comm {
    _private: [u8; 0],
}
pub const WGS_COMM_ID_BYTES: usize = 128;
pub const WGS_COMM_SELF_NEIGHBOURS: i32 = 1;

/// include/wgsparkl_hip.h WGS_ABI_VERSION as this file mirrors it
pub const ABI_VERSION: u32 = 6;

extern "C" {
    pub fn wgs_last_error() -> *const c_char;
    pub fn wgs_dim() -> i32;
    pub fn wgs_pipeline_create(hip_device: i32, out: *mut *mut wgs_pipeline) -> wgs_status;
    pub fn wgs_pipeline_destroy(p: *mut wgs_pipeline);
    pub fn wgs_data_create(
        p: *mut wgs_pipeline, params: *const wgs_sim_params, particles: *const wgs_particle, n: usize,
        colliders: *const wgs_collider, nc: usize, cell_width: f32, grid_capacity: u32, out: *mut *mut wgs_data,
    ) -> wgs_status;
    pub fn wgs_data_destroy(d: *mut wgs_data);
    pub fn wgs_set_constitutive_model(d: *mut wgs_data, model: i32) -> wgs_status;
    pub fn wgs_step(p: *mut wgs_pipeline, d: *mut wgs_data, num_substeps: u32, timestamps: i32) -> wgs_status;
    pub fn wgs_sync(d: *mut wgs_data) -> wgs_status;
    pub fn wgs_set_sim_params(d: *mut wgs_data, params: *const wgs_sim_params) -> wgs_status;
    pub fn wgs_set_collider_poses(d: *mut wgs_data, poses: *const wgs_pose, coms: *const f32, n: usize) -> wgs_status;
    pub fn wgs_set_body_velocities(d: *mut wgs_data, vels: *const wgs_velocity, n: usize) -> wgs_status;
    pub fn wgs_set_body_mass_properties(d: *mut wgs_data, mprops: *const wgs_mass_properties, n: usize) -> wgs_status;
    pub fn wgs_read_body_poses(
        d: *mut wgs_data, poses: *mut wgs_pose, vels: *mut wgs_velocity, coms: *mut f32, n: usize,
    ) -> wgs_status;
    pub fn wgs_read_positions(d: *mut wgs_data, out: *mut f32) -> wgs_status;
    pub fn wgs_get_device_ptrs(d: *mut wgs_data, out: *mut wgs_device_ptrs) -> wgs_status;
    pub fn wgs_read_particles(d: *mut wgs_data, out: *mut wgs_particle, plastic: *mut wgs_plastic_state) -> wgs_status;
    pub fn wgs_prep_vertex_buffer(d: *mut wgs_data, mode: u32, instances: *mut wgs_instance) -> wgs_status;
    pub fn wgs_prep_vertex_buffer_device(d: *mut wgs_data, mode: u32, device_instances: *mut wgs_instance) -> wgs_status;
    pub fn wgs_set_plastic_state(d: *mut wgs_data, states: *const wgs_plastic_state) -> wgs_status;
    pub fn wgs_set_rigid_particles(
        d: *mut wgs_data, local_points: *const f32, ids: *const wgs_sample_ids, n: usize, local_vertices: *const f32,
        vertex_collider_ids: *const u32, nv: usize,
    ) -> wgs_status;
    pub fn wgs_read_grid(d: *mut wgs_data, out: *mut wgs_node_record, capacity: usize, count: *mut usize) -> wgs_status;
    pub fn wgs_read_blocks(
        d: *mut wgs_data, out: *mut wgs_block_record, capacity: usize, count: *mut usize, sorted_ids: *mut u32,
    ) -> wgs_status;
    pub fn wgs_read_timings(d: *mut wgs_data, ms: *mut f32) -> wgs_status;
    pub fn wgs_get_stats(d: *mut wgs_data, out: *mut wgs_stats) -> wgs_status;
    pub fn wgs_read_timing_overhead(d: *mut wgs_data, ms_per_mark: *mut f32) -> wgs_status;
    pub fn wgs_build_info() -> *const c_char;
    /// WGS_ABI_VERSION of the header the library was built from; a binding compares it with `ABI_VERSION` once after loading
    pub fn wgs_abi_version() -> u32;
    pub fn wgs_set_grid_growth(d: *mut wgs_data, enabled: i32) -> wgs_status;
    pub fn wgs_set_uniform_material(d: *mut wgs_data, mass: f32, init_volume: f32, lambda: f32, mu: f32) -> wgs_status;
    /// test hook: the device scan on caller data (prefix_sum.rs:183-229 vectors)
    pub fn wgs_debug_scan(p: *mut wgs_pipeline, values: *const u32, n: u32, out: *mut u32, total: *mut u32) -> wgs_status;

    // ---- multi-GPU (x-slab decomposition): include/wgsparkl_hip.h "Multi-GPU" sections ----
    pub fn wgs_data_create_sharded(
        p: *mut wgs_pipeline, params: *const wgs_sim_params, particles: *const wgs_particle, n: usize, global_ids: *const u32,
        colliders: *const wgs_collider, nc: usize, cell_width: f32, grid_capacity: u32, particle_capacity: u32, block_lo: i32,
        block_hi: i32, force_plastic: i32, out: *mut *mut wgs_data,
    ) -> wgs_status;
    pub fn wgs_shard_halo_record_bytes() -> u32;
    pub fn wgs_shard_particle_record_bytes() -> u32;
    pub fn wgs_shard_buffer_header_bytes() -> u32;
    pub fn wgs_set_stream(d: *mut wgs_data, hip_stream: *mut c_void) -> wgs_status;
    // one call per frame: the whole substep protocol inside the library, RCCL point-to-point as transport
    pub fn wgs_comm_get_unique_id(id: *mut u8) -> wgs_status; // WGS_COMM_ID_BYTES = 128
    pub fn wgs_comm_create(
        p: *mut wgs_pipeline, id: *const u8, rank: i32, world: i32, flags: i32, out: *mut *mut wgs_comm,
    ) -> wgs_status;
    pub fn wgs_comm_destroy(c: *mut wgs_comm);
    pub fn wgs_shard_attach(
        d: *mut wgs_data, comm: *mut wgs_comm, has_lower: i32, has_upper: i32, halo_capacity_records: u32, migrant_capacity: u32,
    ) -> wgs_status;
    pub fn wgs_sharded_step(p: *mut wgs_pipeline, d: *mut wgs_data, num_substeps: u32) -> wgs_status;
    pub fn wgs_sharded_step_lockstep(p: *mut wgs_pipeline, slabs: *mut *mut wgs_data, num_slabs: u32, num_substeps: u32) -> wgs_status;
    pub fn wgs_shard_export(d: *mut wgs_data, device_buf: *mut c_void, capacity_records: u32, count: *mut u32) -> wgs_status;
}
