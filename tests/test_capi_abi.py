"""-m "not gpu": the C-ABI libraries load and export every symbol include/wgsparkl_hip.h declares;
struct layouts seen by ctypes match the header; creating a pipeline without a GPU fails loudly
(no CPU fallback). No compute call is made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "wgsparkl_hip.h")).read()
# function declarations: "<ret> wgs_name(" at the start of a statement (skips the "wgs_status (0 = ok)" prose)
DECLARED = sorted(set(re.findall(r"^(?:const char \*|int32_t |uint32_t |void |wgs_status )(wgs_[a-z_]+)\(", HEADER, re.M)))


def test_header_declares_the_expected_surface():
    must = {"wgs_pipeline_create", "wgs_data_create", "wgs_step", "wgs_sync", "wgs_set_sim_params",
            "wgs_set_collider_poses", "wgs_set_body_velocities", "wgs_read_positions", "wgs_read_particles",
            "wgs_read_grid", "wgs_read_blocks", "wgs_read_timings", "wgs_get_stats", "wgs_data_destroy",
            "wgs_pipeline_destroy", "wgs_last_error", "wgs_dim", "wgs_set_constitutive_model"}
    assert must <= set(DECLARED)
    # every entry point cites the reference interface it replaces
    assert HEADER.count("src/pipeline.rs") >= 5 and "src_testbed/step.rs" in HEADER


@pytest.mark.parametrize("dim", [2, 3])
def test_library_exports_every_declared_symbol(hip_libs, dim):
    lib, T = hip_libs.load(dim)
    for name in DECLARED:
        assert hasattr(lib, name), f"libwgsparkl{dim}d_hip.so does not export {name}"
    assert set(hip_libs.EXPORTS) == set(DECLARED)
    assert lib.wgs_dim() == dim


@pytest.mark.parametrize("dim,size", [(3, 188), (2, 132)])
def test_struct_layouts(hip_libs, dim, size):
    """wgs_particle is the repr(C) image of Particle (particle3d.rs:53-60): all 4-byte members, no padding."""
    _, T = hip_libs.load(dim)
    assert C.sizeof(T.Particle) == size
    assert T.Particle.dynamics.offset == 4 * dim
    assert C.sizeof(T.Cdf) == 4 * (2 * dim + 2)
    assert C.sizeof(T.SimParams) == 4 * (dim + 1)
    assert C.sizeof(T.Collider) == 4 * (1 + 4 + 8 + 6 + 3)
    assert C.sizeof(T.NodeRecord) == 4 * (2 * dim + 4)
    assert C.sizeof(T.BlockRecord) == 4 * (dim + 2)


@pytest.mark.parametrize("dim", [2, 3])
def test_ctypes_structs_match_the_c_header(hip_libs, dim, tmp_path):
    """sizeof / offsetof of every struct of include/wgsparkl_hip.h as gcc sees it against the ctypes twin."""
    import subprocess
    _, T = hip_libs.load(dim)
    fields = {
        "wgs_particle": ("Particle", ["position", "dynamics", "model", "has_plasticity", "plasticity", "has_phase", "phase"]),
        "wgs_collider": ("Collider", ["shape_type", "shape", "pose", "velocity", "com"]),
        "wgs_pose": ("Pose", ["rotation", "translation", "scale"]),
        "wgs_velocity": ("Velocity", ["linear", "angular"]),
        "wgs_mass_properties": ("MassProperties", ["inv_mass", "inv_inertia_local"]),
        "wgs_sim_params": ("SimParams", ["gravity", "dt"]),
        "wgs_node_record": ("NodeRecord", ["cell", "velocity", "mass", "cdf_distance", "cdf_affinities", "cdf_closest_id"]),
        "wgs_block_record": ("BlockRecord", ["virtual_id", "first_particle", "num_particles"]),
        "wgs_stats": ("Stats", ["num_particles", "num_active_blocks", "grid_capacity", "overflow", "substeps_done", "device_bytes", "num_near_collider_blocks", "grid_growths", "cell_changers", "table_rebuilds", "block_ids", "block_ids_free", "table_marks", "table_refreshes"]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#define WGS_DIM {dim}', '#include "wgsparkl_hip.h"', 'int main(void) {']
    for cname, (_, fs) in fields.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for f in fs:
            lines.append(f'  printf(" %zu", offsetof({cname}, {f}));')
        lines.append('  printf("\\n");')
    lines += ['  printf("wgs_instance %zu\\n", sizeof(wgs_instance));', '  printf("wgs_sample_ids %zu\\n", sizeof(wgs_sample_ids));',
              '  printf("wgs_plastic_state %zu\\n", sizeof(wgs_plastic_state));', '  return 0; }']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c11", f"-I{inc}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    seen = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in out}
    for cname, (tname, fs) in fields.items():
        ct = getattr(T, tname)
        assert seen[cname][0] == C.sizeof(ct), cname
        offs = [getattr(ct, {"lambda": "lambda_"}.get(f, f)).offset for f in fs]
        assert seen[cname][1:] == offs, (cname, seen[cname][1:], offs)
    assert seen["wgs_instance"] == [96] and seen["wgs_sample_ids"] == [16]
    assert seen["wgs_plastic_state"] == [C.sizeof(T.PlasticState)]


def test_no_cpu_fallback(hip_libs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib, _ = hip_libs.load(3)
    h = C.c_void_p()
    st = lib.wgs_pipeline_create(0, C.byref(h))
    assert st == 2 and not h.value                      # WGS_ERR_NO_DEVICE
    assert b"no HIP device" in lib.wgs_last_error()
    from wgsparkl_amd import MpmPipeline
    from wgsparkl_amd._ffi import WgsError
    with pytest.raises(WgsError):
        MpmPipeline(0, 3)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under wgsparkl_amd/ or include/ may reference it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "wgsparkl_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".sh")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle\b|oracle/|mpm_oracle|liborc", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_rust_binding_declares_every_header_symbol():
    """rust/wgsparkl-hip-sys/src/lib.rs cannot be compiled here (no cargo / rustc in the image), so at least its
    declarations are kept in step with the header: every function of include/wgsparkl_hip.h appears in the extern block."""
    rs = open(os.path.join(ROOT, "rust", "wgsparkl-hip-sys", "src", "lib.rs")).read()
    declared_rs = set(re.findall(r"pub fn (wgs_[a-z_]+)\(", rs))
    assert set(DECLARED) <= declared_rs, sorted(set(DECLARED) - declared_rs)
    assert declared_rs <= set(DECLARED), sorted(declared_rs - set(DECLARED))
    assert "grid_growths" in rs and "WGS_COMM_ID_BYTES" in rs
