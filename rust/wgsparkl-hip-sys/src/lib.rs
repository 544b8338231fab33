//! FFI declarations for include/wgsparkl_hip.h — SOURCE ONLY: this repository's build image has no
//! Rust toolchain, so this file is not compiled or tested here. See INTEGRATION.md §2 for the full
//! listing and §3 for the safe wrapper that keeps wgsparkl's `MpmPipeline` / `MpmData` surface.
#![allow(non_camel_case_types)]
pub const DIM: usize = if cfg!(feature = "dim2") { 2 } else { 3 };
pub type wgs_status = i32;
pub enum wgs_pipeline {}
pub enum wgs_data {}
#[repr(C)]
#[derive(Copy, Clone)]
pub struct wgs_sim_params {
    pub gravity: [f32; DIM],
    pub dt: f32,
}
extern "C" {
    pub fn wgs_pipeline_create(hip_device: i32, out: *mut *mut wgs_pipeline) -> wgs_status;
    pub fn wgs_pipeline_destroy(p: *mut wgs_pipeline);
    pub fn wgs_step(p: *mut wgs_pipeline, d: *mut wgs_data, num_substeps: u32, timestamps: i32) -> wgs_status;
    pub fn wgs_sync(d: *mut wgs_data) -> wgs_status;
    pub fn wgs_data_destroy(d: *mut wgs_data);
    pub fn wgs_set_sim_params(d: *mut wgs_data, params: *const wgs_sim_params) -> wgs_status;
    pub fn wgs_read_positions(d: *mut wgs_data, out: *mut f32) -> wgs_status;
    // remaining entry points and structs: INTEGRATION.md §2
}
