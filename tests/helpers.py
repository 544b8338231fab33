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
