"""The C oracle (gather over per-node lists, hash grid, Jacobi SVD) and the numpy twin
(vectorised scatter, dense grid, LAPACK SVD) are two independent restatements of the same
WGSL; they must agree to fp64 round-off. This is the guard against transcription errors."""
import numpy as np
import pytest

from oracle.np_oracle import NpState
from wgsparkl_amd import scenes
from wgsparkl_amd.models import MODEL_COROTATED, MODEL_NEO_HOOKEAN, DruckerPrager, ParticlePhase
from wgsparkl_amd.solver import SimulationParams


def run_pair(oracle_libs, ps, params, h, model, k, cap=4096):
    orc = oracle_libs.Oracle(ps.dim, np.float64)
    st = orc.new_state(ps, params, [], h, cap, model)
    tw = NpState(ps, params, h, model)
    st.step(k)
    tw.step(k)
    return st, tw


def close(a, b, tol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.sqrt(np.mean(b * b)), 1e-300)
    err = np.sqrt(np.mean((a - b) ** 2)) / scale
    assert err < tol, err


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("model", [MODEL_COROTATED, MODEL_NEO_HOOKEAN])
def test_elastic_cloud(oracle_libs, dim, model):
    ps = scenes.random_cloud(3000, dim=dim, seed=5, phase=ParticlePhase(1.0, -1.0))
    params = SimulationParams(gravity=(0.3, -9.81, 0.2)[:dim], dt=1e-3)
    st, tw = run_pair(oracle_libs, ps, params, 1.0, model, 3)
    close(st.arr["pos"], tw.pos, 1e-13)
    close(st.arr["vel"], tw.vel, 1e-11)
    close(st.arr["def_grad"], tw.F, 1e-12)
    close(st.arr["affine"], tw.C, 1e-9)
    cells, mv = st.grid_records()[:2]
    gv, gm = tw.grid_at(cells)
    close(mv[:, :dim], gv, 1e-11)
    close(mv[:, dim], gm, 1e-12)


@pytest.mark.parametrize("dim", [2, 3])
def test_drucker_prager_and_fracture(oracle_libs, dim):
    """phase None -> DP always on (quirk B1 with lambda = mu = -1 when plasticity is None too);
    a second population with phase 1 and a finite max_stretch exercises the fracture switch."""
    ps = scenes.random_cloud(2000, dim=dim, seed=9, young=1e6, plasticity=DruckerPrager.new(1e6, 0.25),
                             phase=None, perturb_F=0.08)
    ps.phase[::3] = (1.0, 1.05)      # may fracture: max singular value > 1.05
    ps.phase[1::3] = (1.0, -1.0)     # never fractures, never plastic
    ps.dp[2::7, 4:] = -1.0           # plasticity None -> lambda = mu = -1
    params = SimulationParams(gravity=(0.0, -9.81, 0.0)[:dim], dt=5e-4)
    st, tw = run_pair(oracle_libs, ps, params, 1.0, MODEL_COROTATED, 2)
    assert np.array_equal(st.arr["phase"][:, 0], tw.phase[:, 0])
    assert (st.arr["phase"][::3, 0] == 0).any() and (st.arr["phase"][::3, 0] == 1).any()
    close(st.arr["def_grad"], tw.F, 1e-10)
    close(st.arr["dp_state"], tw.dp_state, 1e-10)
    close(st.arr["affine"], tw.C, 1e-8)
    close(st.arr["vel"], tw.vel, 1e-10)
