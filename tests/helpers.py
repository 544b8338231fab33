"""Shared helpers of the parity tests: run the same scene through the CPU oracle
(fp32 restatement + fp64 "truth") and through the HIP path (C ABI), then compare
by VIRTUAL ids (world cell coordinates), never by physical node index."""
import numpy as np

from oracle.orc import Oracle
from wgsparkl_amd import MpmData, MpmPipeline

_ORACLES = {}


def oracle(dim, dtype):
    key = (dim, np.dtype(dtype).name)
    if key not in _ORACLES:
        _ORACLES[key] = Oracle(dim, dtype)
    return _ORACLES[key]


def run_oracle(scene, nsteps, dtype=np.float32):
    ps = scene["particles"]
    st = oracle(ps.dim, dtype).new_state(ps, scene["params"], scene["colliders"], scene["cell_width"],
                                         scene["grid_capacity"], scene.get("model", 0))
    st.step(nsteps)
    assert not st.overflow
    return st


_PIPES = {}


def pipeline(dim):
    if dim not in _PIPES:
        _PIPES[dim] = MpmPipeline(0, dim)
    return _PIPES[dim]


def run_gpu(scene, nsteps, timestamps=False):
    ps = scene["particles"]
    pipe = pipeline(ps.dim)
    data = MpmData.new(pipe, scene["params"], ps, scene["colliders"], scene["cell_width"],
                       scene["grid_capacity"], scene.get("model", 0))
    pipe.step(data, nsteps, timestamps)
    data.sync()
    return data


def rel_rms(a, b):
    """RMS of (a - b) relative to RMS of b (b = truth)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.sqrt(np.mean(b * b))
    num = np.sqrt(np.mean((a - b) ** 2))
    return num / den if den > 0 else num


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if len(a) else 0.0


def grid_of(state):
    """Oracle grid restricted to what matters for parity: (cells, velocity|mass)."""
    cells, mv, dist, aff, closest = state.grid_records()
    return cells, np.asarray(mv, np.float64), dist, aff, closest


def compare_grids(gpu_grid, orc_grid):
    """Both sorted lexicographically by cell; the GPU path may hold the same set of active
    blocks only (bit-exact requirement on cell / block indices)."""
    gc, gv = gpu_grid[0], gpu_grid[1]
    oc, ov = orc_grid[0], orc_grid[1]
    assert gc.shape == oc.shape, f"active node sets differ: {gc.shape} vs {oc.shape}"
    assert np.array_equal(gc, oc), "active node cells differ"
    return gv, ov


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


_MARGINS = []


def report_margin(name, value, bound, **extra):
    """Parity margins are REPORTED, not only bounded: every floating-point comparison appends (measured, bound) to
    $WGS_MARGINS_FILE (JSON lines; tools/gpu_margins.sh collects them into profiles/rNN_parity_margins.json)."""
    import json
    import os
    rec = dict(test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], name=name, value=float(value), bound=float(bound), **extra)
    _MARGINS.append(rec)
    path = os.environ.get("WGS_MARGINS_FILE")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


def assert_close_to_truth(name, got, ref32, truth, rel_tol, k32=None):
    """fp32 tolerance rule used by every floating-point parity check:
    the HIP result must be within `rel_tol` (relative RMS) of the fp64 oracle. One field has an escape: `affine`
    (C' = APIC matrix minus the stress term) is round-off times E wherever the body moves rigidly — its fp64 value is
    ~0 and a relative bound means nothing —, so there the HIP result may instead be within `k32` = 4 times the error
    the fp32 restatement of the reference's own arithmetic makes against the same fp64 truth (measured: 1.0-2.6 x,
    profiles/r02_parity_margins.json). No other field needs it, and none gets it."""
    if k32 is None:
        k32 = 4.0 if name.startswith("affine") else 0.0
    got = np.asarray(got, np.float64)
    truth = np.asarray(truth, np.float64)
    e_gpu = rms(got - truth)
    e_32 = rms(np.asarray(ref32, np.float64) - truth)
    scale = rms(truth)
    bound = max(rel_tol * scale, k32 * e_32)
    escape = bool(k32 * e_32 > rel_tol * scale)
    if escape and e_gpu > rel_tol * scale:
        # the relative bound means nothing here (truth ~ 0): what is checked, and reported under a name of its own, is the HIP
        # error in multiples of the error the fp32 restatement of the reference's arithmetic makes against the same truth
        report_margin(name + " [escape used: multiples of the fp32 restatement's own error]", e_gpu / e_32, k32, escape_used=True)
    else:
        report_margin(name, e_gpu / scale if scale > 0 else e_gpu, rel_tol, fp32_oracle_err=e_32 / scale if scale > 0 else e_32,
                      escape_available=escape, escape_used=False)
    assert np.isfinite(e_gpu) and e_gpu <= bound, (
        f"{name}: rms err {e_gpu:.3e} > bound {bound:.3e} (scale {scale:.3e}, fp32-oracle err {e_32:.3e})")
    return e_gpu / scale if scale > 0 else e_gpu


def compare_cpic(data, st32, st64, dim, grid_tol, part_tol, min_same=0.995, fields=("pos", "vel", "def_grad", "affine"), h=1.0,
                 truth=None):
    """Parity of a collider (CPIC) scene. `truth` = (dict of fp64 particle arrays, fp64 grid velocity|mass array) may
    replace the fp64 oracle state (golden vectors). Integer results first — active cells, node affinity / sign bits, closest
    collider ids: exact. CPIC has discrete per-particle decisions sitting on fp32 thresholds (sign votes, det M > 1e-8),
    so a few particles may carry other affinity bits than the oracle's; they, and the grid nodes inside their 3^D
    stencil (a node inside a collider can hold 1e-6 of a particle's mass: one such particle moves its velocity by
    O(1)), are left out of the floating-point comparison, and their NUMBER is reported and bounded. Everything else is
    compared against the fp64 oracle with `grid_tol` / `part_tol` (relative RMS)."""
    cells, vm, dist, aff, closest = data.read_grid()
    oc, omv, odist, oaff, oclosest = st32.grid_records()
    assert np.array_equal(cells, oc), "active node cells differ"
    assert np.array_equal(aff, oaff), "node affinity / sign bits differ"
    assert np.array_equal(closest, oclosest), "closest collider ids differ"
    got = data.read_particles()
    same = got.cdf_affinity == st32.arr["cdf_affinity"]
    report_margin("particle affinity mismatch fraction", 1.0 - float(same.mean()), 1.0 - min_same, count=int((~same).sum()))
    assert same.mean() >= min_same, f"affinity bits agree for only {same.mean():.5f} of the particles"
    keep = np.ones(len(cells), bool)
    if not same.all():
        h = np.float32(h)
        bad = []
        for pos in (st32.arr["pos"][~same], got.pos[~same]):          # stencils at either position (they agree to 1e-6)
            base = (np.rint(pos.astype(np.float32) / h) - 1).astype(np.int64)
            for off in np.ndindex(*([3] * dim)):
                bad.append(base + np.array(off))
        bad = np.unique(np.concatenate(bad), axis=0)
        key = lambda c: (c[:, 0] * 4099 + c[:, 1]) * 4099 + (c[:, 2] if dim == 3 else 0)
        keep = ~np.isin(key(cells.astype(np.int64)), key(bad))
        report_margin("grid nodes left out (stencils of mismatching particles)", float((~keep).mean()), 0.02, count=int((~keep).sum()))
        assert (~keep).mean() < 0.02
    t_arr, o64 = truth if truth is not None else (st64.arr, grid_of(st64)[1])
    assert_close_to_truth("grid velocity (CPIC)", vm[keep][:, :dim], omv[keep][:, :dim], o64[keep][:, :dim], grid_tol)
    assert_close_to_truth("grid mass (CPIC)", vm[keep][:, dim], omv[keep][:, dim], o64[keep][:, dim], grid_tol)
    for f in fields:
        assert_close_to_truth(f + " (CPIC)", getattr(got, f)[same], st32.arr[f][same], t_arr[f][same], part_tol)
    return got, same
