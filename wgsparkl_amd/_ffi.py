"""ctypes binding of the C ABI (include/wgsparkl_hip.h) — one CDLL per dimension.

The library is the product; this module is the thinnest possible host glue.
It fails loudly when the HIP extension is missing: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

WGS_OK = 0
ABI_VERSION = 6   # include/wgsparkl_hip.h WGS_ABI_VERSION
WGS_NUM_PASSES = 10
PASS_NAMES = ("update rigid particles", "grid sort", "grid_update_cdf", "p2g_cdf", "g2p_cdf", "p2g",
              "grid_update", "g2p", "particles_update", "integrate_bodies")  # src/pipeline.rs:201-271
EXPORTS = (
    "wgs_last_error", "wgs_dim", "wgs_pipeline_create", "wgs_pipeline_destroy", "wgs_data_create",
    "wgs_data_destroy", "wgs_set_constitutive_model", "wgs_step", "wgs_sync", "wgs_set_sim_params",
    "wgs_set_collider_poses", "wgs_set_body_velocities", "wgs_set_body_mass_properties", "wgs_read_body_poses", "wgs_set_plastic_state", "wgs_read_timing_overhead", "wgs_set_rigid_particles", "wgs_prep_vertex_buffer", "wgs_prep_vertex_buffer_device", "wgs_read_positions", "wgs_get_device_ptrs", "wgs_read_particles",
    "wgs_read_grid", "wgs_read_blocks", "wgs_read_timings", "wgs_get_stats",
    # multi-GPU (x-slab decomposition; new design, no reference counterpart)
    "wgs_data_create_sharded", "wgs_shard_halo_record_bytes", "wgs_shard_particle_record_bytes",
    "wgs_shard_buffer_header_bytes", "wgs_set_stream", "wgs_shard_export",
    # one call per frame on sharded data (RCCL inside the library) + build identification
    "wgs_comm_get_unique_id", "wgs_comm_create", "wgs_comm_destroy", "wgs_shard_attach", "wgs_sharded_step",
    "wgs_sharded_step_lockstep", "wgs_build_info", "wgs_abi_version", "wgs_debug_scan", "wgs_set_grid_growth", "wgs_set_uniform_material",
)


class WgsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"wgsparkl_hip error {code}: {msg}")
        self.code = code


def lib_path(dim: int) -> str:
    return os.path.join(_CSRC, f"libwgsparkl{dim}d_hip.so")


def build(force: bool = False) -> None:
    """hipcc --offload-arch=gfx950 build of both libraries (csrc/build.sh)."""
    cmd = ["bash", os.path.join(_CSRC, "build.sh")] + (["force"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the HIP extension failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])


def make_types(D: int):
    f = C.c_float
    u = C.c_uint32

    class SimParams(C.Structure):
        _fields_ = [("gravity", f * D), ("dt", f)]

    class Elastic(C.Structure):
        _fields_ = [("lambda_", f), ("mu", f)]

    class DruckerPrager(C.Structure):
        _fields_ = [("h0", f), ("h1", f), ("h2", f), ("h3", f), ("lambda_", f), ("mu", f)]

    class PlasticState(C.Structure):
        _fields_ = [("plastic_deformation_gradient_det", f), ("plastic_hardening", f), ("log_vol_gain", f)]

    class Phase(C.Structure):
        _fields_ = [("phase", f), ("max_stretch", f)]

    class Cdf(C.Structure):
        _fields_ = [("normal", f * D), ("rigid_vel", f * D), ("signed_distance", f), ("affinity", u)]

    class Dynamics(C.Structure):
        _fields_ = [("velocity", f * D), ("def_grad", f * (D * D)), ("affine", f * (D * D)), ("cdf", Cdf),
                    ("init_volume", f), ("init_radius", f), ("mass", f)]

    class Particle(C.Structure):
        _fields_ = [("position", f * D), ("dynamics", Dynamics), ("model", Elastic), ("has_plasticity", u),
                    ("plasticity", DruckerPrager), ("has_phase", u), ("phase", Phase)]

    class Pose(C.Structure):
        _fields_ = [("rotation", f * 4), ("translation", f * 3), ("scale", f)]

    class Velocity(C.Structure):
        _fields_ = [("linear", f * 3), ("angular", f * 3)]

    class Collider(C.Structure):
        _fields_ = [("shape_type", u), ("shape", f * 4), ("pose", Pose), ("velocity", Velocity), ("com", f * 3)]

    class MassProperties(C.Structure):
        _fields_ = [("inv_mass", f * 3), ("inv_inertia_local", f * 9)]

    class NodeRecord(C.Structure):
        _fields_ = [("cell", C.c_int32 * D), ("velocity", f * D), ("mass", f), ("cdf_distance", f),
                    ("cdf_affinities", u), ("cdf_closest_id", u)]

    class BlockRecord(C.Structure):
        _fields_ = [("virtual_id", C.c_int32 * D), ("first_particle", u), ("num_particles", u)]

    class Stats(C.Structure):
        _fields_ = [("num_particles", u), ("num_active_blocks", u), ("grid_capacity", u), ("overflow", u),
                    ("substeps_done", C.c_uint64), ("device_bytes", C.c_uint64), ("num_near_collider_blocks", u), ("grid_growths", u),
                    ("cell_changers", C.c_uint64), ("table_rebuilds", C.c_uint64),
                    ("block_ids", u), ("block_ids_free", u), ("table_marks", u), ("table_refreshes", u)]

    ns = dict(SimParams=SimParams, Elastic=Elastic, DruckerPrager=DruckerPrager, PlasticState=PlasticState,
              Phase=Phase, Cdf=Cdf, Dynamics=Dynamics, Particle=Particle, Pose=Pose, Velocity=Velocity,
              Collider=Collider, MassProperties=MassProperties, NodeRecord=NodeRecord, BlockRecord=BlockRecord, Stats=Stats)
    return type("Types", (), ns)


class DevicePtrs(C.Structure):
    """wgs_device_ptrs (include/wgsparkl_hip.h): the optional interop view of the particle state on the device."""
    _fields_ = [("position_quads", C.c_void_p), ("particle_ids", C.c_void_p), ("count", C.c_uint32), ("capacity", C.c_uint32),
                ("dim", C.c_uint32), ("reserved", C.c_uint32), ("hip_stream", C.c_void_p)]


_LIBS = {}


def load(dim: int):
    """Load libwgsparkl{dim}d_hip.so; raises if it has not been built."""
    if dim in _LIBS:
        return _LIBS[dim]
    path = lib_path(dim)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc, gfx950). There is no CPU fallback for the MPM step.")
    # PyTorch-ROCm ships its own libamdhip64; if this library pulled in /opt/rocm's copy first, a later
    # torch.cuda init would find a second HIP runtime and report "No HIP GPUs are available". Loading torch
    # first makes both resolve to one runtime (torch is only plumbing here: streams, torch.distributed).
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the single-GPU path
        pass
    lib = C.CDLL(path)
    T = make_types(dim)
    vp = C.c_void_p
    lib.wgs_last_error.restype = C.c_char_p
    lib.wgs_dim.restype = C.c_int32
    lib.wgs_pipeline_create.argtypes = [C.c_int32, C.POINTER(vp)]
    lib.wgs_pipeline_destroy.argtypes = [vp]
    lib.wgs_pipeline_destroy.restype = None
    lib.wgs_data_create.argtypes = [vp, C.POINTER(T.SimParams), C.POINTER(T.Particle), C.c_size_t,
                                    C.POINTER(T.Collider), C.c_size_t, C.c_float, C.c_uint32, C.POINTER(vp)]
    lib.wgs_data_destroy.argtypes = [vp]
    lib.wgs_data_destroy.restype = None
    lib.wgs_set_constitutive_model.argtypes = [vp, C.c_int32]
    lib.wgs_step.argtypes = [vp, vp, C.c_uint32, C.c_int32]
    lib.wgs_sync.argtypes = [vp]
    lib.wgs_set_sim_params.argtypes = [vp, C.POINTER(T.SimParams)]
    lib.wgs_set_collider_poses.argtypes = [vp, C.POINTER(T.Pose), C.POINTER(C.c_float), C.c_size_t]
    lib.wgs_set_body_velocities.argtypes = [vp, C.POINTER(T.Velocity), C.c_size_t]
    lib.wgs_set_body_mass_properties.argtypes = [vp, C.POINTER(T.MassProperties), C.c_size_t]
    lib.wgs_read_body_poses.argtypes = [vp, C.POINTER(T.Pose), C.POINTER(T.Velocity), C.POINTER(C.c_float), C.c_size_t]
    lib.wgs_read_positions.argtypes = [vp, C.POINTER(C.c_float)]
    lib.wgs_get_device_ptrs.argtypes = [vp, C.POINTER(DevicePtrs)]
    lib.wgs_read_particles.argtypes = [vp, C.POINTER(T.Particle), C.POINTER(T.PlasticState)]
    lib.wgs_set_plastic_state.argtypes = [vp, C.POINTER(T.PlasticState)]
    lib.wgs_set_rigid_particles.argtypes = [vp, C.POINTER(C.c_float), vp, C.c_size_t, C.POINTER(C.c_float),
                                            C.POINTER(C.c_uint32), C.c_size_t]
    lib.wgs_prep_vertex_buffer.argtypes = [vp, C.c_uint32, vp]
    lib.wgs_prep_vertex_buffer_device.argtypes = [vp, C.c_uint32, vp]
    lib.wgs_read_grid.argtypes = [vp, C.POINTER(T.NodeRecord), C.c_size_t, C.POINTER(C.c_size_t)]
    lib.wgs_read_blocks.argtypes = [vp, C.POINTER(T.BlockRecord), C.c_size_t, C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_uint32)]
    lib.wgs_read_timings.argtypes = [vp, C.POINTER(C.c_float)]
    lib.wgs_read_timing_overhead.argtypes = [vp, C.POINTER(C.c_float)]
    lib.wgs_get_stats.argtypes = [vp, C.POINTER(T.Stats)]
    u32p = C.POINTER(C.c_uint32)
    lib.wgs_data_create_sharded.argtypes = [vp, C.POINTER(T.SimParams), C.POINTER(T.Particle), C.c_size_t, u32p,
                                            C.POINTER(T.Collider), C.c_size_t, C.c_float, C.c_uint32, C.c_uint32,
                                            C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.wgs_shard_halo_record_bytes.restype = C.c_uint32
    lib.wgs_shard_particle_record_bytes.restype = C.c_uint32
    lib.wgs_shard_buffer_header_bytes.restype = C.c_uint32
    lib.wgs_set_stream.argtypes = [vp, vp]
    lib.wgs_shard_export.argtypes = [vp, vp, C.c_uint32, u32p]
    lib.wgs_build_info.restype = C.c_char_p
    lib.wgs_abi_version.restype = C.c_uint32
    if lib.wgs_abi_version() != ABI_VERSION:   # (the structs carry no size field: include/wgsparkl_hip.h WGS_ABI_VERSION)
        raise RuntimeError(f"{path}: ABI version {lib.wgs_abi_version()}, this binding mirrors {ABI_VERSION} — rebuild the library (csrc/build.sh)")
    lib.wgs_set_grid_growth.argtypes = [vp, C.c_int32]
    lib.wgs_set_uniform_material.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.wgs_debug_scan.argtypes = [vp, u32p, C.c_uint32, u32p, u32p]
    lib.wgs_comm_get_unique_id.argtypes = [C.c_char_p]
    lib.wgs_comm_create.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.wgs_comm_destroy.argtypes = [vp]
    lib.wgs_comm_destroy.restype = None
    lib.wgs_shard_attach.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32]
    lib.wgs_sharded_step.argtypes = [vp, vp, C.c_uint32]
    lib.wgs_sharded_step_lockstep.argtypes = [vp, C.POINTER(vp), C.c_uint32, C.c_uint32]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:  # default restype -> wgs_status
            fn.restype = C.c_int32
    assert lib.wgs_dim() == dim
    _LIBS[dim] = (lib, T)
    return _LIBS[dim]


def check(lib, status):
    if status != WGS_OK:
        msg = lib.wgs_last_error()
        raise WgsError(status, msg.decode() if msg else "")
