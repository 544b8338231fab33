"""Physical invariants of one substep, checked on the fp64 C oracle (SURVEY.md section 8c lists
them as the only other way to pin a restatement when the reference cannot be executed)."""
import numpy as np
import pytest

from wgsparkl_amd import scenes
from wgsparkl_amd.models import MODEL_COROTATED, MODEL_NEO_HOOKEAN, DruckerPrager, ElasticCoefficients, ParticlePhase
from wgsparkl_amd.solver import Collider, ParticleSet, SimulationParams


def state(oracle_libs, ps, params, colliders=(), h=1.0, model=0, cap=4096, dtype=np.float64):
    return oracle_libs.Oracle(ps.dim, dtype).new_state(ps, params, list(colliders), h, cap, model)


@pytest.mark.parametrize("dim", [2, 3])
def test_p2g_conserves_mass_and_momentum(oracle_libs, dim):
    ps = scenes.random_cloud(4000, dim=dim, seed=1, phase=ParticlePhase(1.0, -1.0))
    ps.affine[:] = 0.0   # with C' = 0: sum node momentum = sum m v (weights sum to 1)
    st = state(oracle_libs, ps, SimulationParams(gravity=(0.0,) * dim, dt=1e-3))
    st.sort(); st.grid_update_cdf(); st.g2p_cdf(); st.p2g()
    mv = st.grid_records()[1]
    assert abs(mv[:, dim].sum() - ps.mass.astype(np.float64).sum()) < 1e-9 * ps.mass.sum()
    want = (ps.mass[:, None].astype(np.float64) * ps.vel).sum(0)
    assert np.allclose(mv[:, :dim].sum(0), want, rtol=0, atol=1e-9 * np.abs(ps.mass[:, None] * ps.vel).sum())


def test_affine_term_carries_no_net_momentum(oracle_libs):
    """sum_i w_ip (x_i - x_p) = 0, so C' changes the momentum distribution, not its total."""
    ps = scenes.random_cloud(3000, dim=3, seed=2, phase=ParticlePhase(1.0, -1.0))
    st = state(oracle_libs, ps, SimulationParams(gravity=(0.0,) * 3, dt=1e-3))
    st.sort(); st.grid_update_cdf(); st.g2p_cdf(); st.p2g()
    mv = st.grid_records()[1]
    want = (ps.mass[:, None].astype(np.float64) * ps.vel).sum(0)
    assert np.allclose(mv[:, :3].sum(0), want, atol=1e-8 * np.abs(ps.affine).sum())


@pytest.mark.parametrize("model", [MODEL_COROTATED, MODEL_NEO_HOOKEAN])
def test_uniform_translation_and_free_fall(oracle_libs, model):
    """A cloud with one velocity keeps it (plus k g dt), F stays I, stress stays 0 (tau(I) = 0)."""
    ps = scenes.neo_hookean_cube(n_side=12)["particles"]
    v0 = np.array([0.7, -0.2, 0.4])
    ps.vel[:] = v0
    g = np.array([0.0, -9.81, 0.0])
    dt, k = 1.0 / 1200.0, 7
    st = state(oracle_libs, ps, SimulationParams(gravity=tuple(g), dt=dt), model=model)
    st.step(k)
    assert np.allclose(st.arr["vel"], v0 + k * g * dt, atol=1e-12)
    assert np.allclose(st.arr["def_grad"], np.eye(3).reshape(-1), atol=1e-12)
    # C' = grad * m - tau * ...: both vanish
    assert np.abs(st.arr["affine"]).max() < 1e-6
    x_want = ps.pos.astype(np.float64) + sum((v0 + (i + 1) * g * dt) * dt for i in range(k))
    assert np.allclose(st.arr["pos"], x_want, atol=1e-12)


@pytest.mark.parametrize("model", [MODEL_COROTATED, MODEL_NEO_HOOKEAN])
def test_stress_objectivity(oracle_libs, model):
    """tau(R F) = R tau(F) R^T for a rotation R; tau(I) = 0."""
    orc = oracle_libs.Oracle(3, np.float64)
    rng = np.random.default_rng(4)
    lam, mu = 1.3e5, 0.7e5
    assert np.abs(orc.kirchoff_stress(model, lam, mu, np.eye(3).reshape(-1))).max() < 1e-9
    for _ in range(20):
        F = np.eye(3) + rng.normal(0, 0.2, (3, 3))
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        R = q * np.sign(np.linalg.det(q))
        cm = lambda m: m.T.reshape(-1)          # column-major flatten
        tau = orc.kirchoff_stress(model, lam, mu, cm(F)).reshape(3, 3).T
        tau_r = orc.kirchoff_stress(model, lam, mu, cm(R @ F)).reshape(3, 3).T
        assert np.allclose(tau_r, R @ tau @ R.T, atol=1e-7 * np.abs(tau).max())
        assert np.allclose(tau, tau.T, atol=1e-7 * np.abs(tau).max())   # Kirchhoff stress is symmetric


def test_svd_reconstructs(oracle_libs):
    orc = oracle_libs.Oracle(3, np.float64)
    rng = np.random.default_rng(5)
    mats = [np.eye(3), np.zeros((3, 3)), np.diag([2.0, 0.0, 0.0]), np.diag([1.0, 1.0, -1.0])]
    mats += [rng.normal(size=(3, 3)) for _ in range(30)]
    for F in mats:
        u, s, vt = orc.svd(F.T.reshape(-1))
        U, Vt = u.reshape(3, 3).T, vt.reshape(3, 3).T
        assert np.allclose(U @ np.diag(s) @ Vt, F, atol=1e-12)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12) and np.allclose(Vt @ Vt.T, np.eye(3), atol=1e-12)
        assert np.linalg.det(U) > 0 and np.linalg.det(Vt) > 0            # proper rotations
        assert np.allclose(np.sort(np.abs(s)), np.sort(np.linalg.svd(F, compute_uv=False)), atol=1e-12)


def test_drucker_prager_cases(oracle_libs):
    """drucker_prager.wgsl:112-158: disabled when lambda == 0; expansion (tr > 0) projects to U V^T;
    gamma <= 0 (inside the yield surface) leaves everything unchanged."""
    orc = oracle_libs.Oracle(3, np.float64)
    dp = DruckerPrager.new(1.0e6, 0.2).as_array().astype(np.float64)
    st0 = np.array([1.0, 1.0, 0.0])
    F = (np.eye(3) * 1.1).T.reshape(-1)
    off = dp.copy(); off[4] = 0.0
    changed, st, Fo = orc.drucker_prager_project(off, st0, F)
    assert not changed and np.array_equal(Fo, F) and np.array_equal(st, st0)
    changed, st, Fo = orc.drucker_prager_project(dp, st0, F)          # pure expansion
    assert changed and np.allclose(Fo.reshape(3, 3), np.eye(3), atol=1e-12)
    assert np.isclose(st[2], np.log(1.1 ** 3)) and np.isclose(st[0], 1.1 ** 3)
    # sheared + slightly compressed: gamma > 0 -> return mapping onto the yield surface with tr unchanged
    Fy = np.diag([1.25, 0.78, 1.0]).T.reshape(-1)
    changed, st, Fo = orc.drucker_prager_project(dp, st0, Fy)
    assert changed
    eps = np.log(np.abs(np.diag(Fo.reshape(3, 3))))
    tr, dev = eps.sum(), eps - eps.sum() / 3
    angle = dp[0] + (dp[1] * 1.0 - dp[3]) * np.exp(-dp[2] * 1.0)
    alpha = np.sqrt(2 / 3) * 2 * np.sin(angle) / (3 - np.sin(angle))
    assert np.isclose(tr, np.log(1.25 * 0.78))
    assert np.isclose(np.linalg.norm(dev) + (3 * dp[4] + 2 * dp[5]) / (2 * dp[5]) * tr * alpha, 0.0, atol=1e-12)
    assert st[1] > 1.0 and np.isclose(st[0], 1.0) and abs(st[2]) < 1e-12   # volume preserved by the projection
    Fs = np.diag([0.90, 0.905, 0.91]).T.reshape(-1)                    # strongly compressed, tiny shear: gamma <= 0
    changed, st, Fo = orc.drucker_prager_project(dp, st0, Fs)
    assert not changed and np.array_equal(Fo, Fs)


def test_first_substep_transfers_no_stress(oracle_libs):
    """Quirk B12: affine = 0 initially, so stress enters C' only at the end of substep 1."""
    ps = scenes.random_cloud(500, seed=3, phase=ParticlePhase(1.0, -1.0), perturb_F=0.2, perturb_C=0.0, vel_scale=0.0)
    ps.affine[:] = 0.0
    st = state(oracle_libs, ps, SimulationParams(gravity=(0.0, 0.0, 0.0), dt=1e-3))
    st.step(1)
    assert np.abs(st.arr["vel"]).max() == 0.0            # no force reached the grid yet
    assert np.abs(st.arr["affine"]).max() > 0.0          # but the stress is now stored in C'
    st.step(1)
    assert np.abs(st.arr["vel"]).max() > 0.0


def test_node_cdf_of_a_floor(oracle_libs):
    """collide.wgsl:23-56 with one cuboid floor: nodes within 1.5 h of the top face get affinity bit 0,
    nodes inside also get the sign bit; distance = |y - top|."""
    sc = scenes.neo_hookean_cube(n_side=8, with_floor=True)
    sc["particles"].pos[:, 1] -= 6.5       # bring the cube down to the floor (top face at y = 2)
    st = state(oracle_libs, sc["particles"], sc["params"], sc["colliders"], model=sc["model"])
    st.sort(); st.grid_update_cdf()
    cells, _, dist, aff, closest = st.grid_records()
    y = cells[:, 1].astype(np.float64)
    near = np.abs(y - 2.0) <= 1.5
    inside = y <= 2.0
    assert np.array_equal((aff & 1) == 1, near | inside)
    assert np.array_equal(((aff >> 16) & 1) == 1, inside)
    assert np.allclose(dist[(aff & 1) == 1], np.abs(y - 2.0)[(aff & 1) == 1])
    assert np.all(closest[(aff & 1) == 1] == 0) and np.all(closest[(aff & 1) == 0] == 0xFFFFFFFF)
    assert np.all(dist[(aff & 1) == 0] == 1e10)


def test_particles_rest_on_floor(oracle_libs):
    """CPIC end to end on the oracle: a block dropped on a floor cuboid stays above it."""
    sc = scenes.neo_hookean_cube(n_side=8, with_floor=True)
    ps = sc["particles"]
    ps.pos[:, 1] -= 5.9
    ps.vel[:, 1] = -2.0
    st = state(oracle_libs, ps, sc["params"], sc["colliders"], model=sc["model"], dtype=np.float32)
    st.step(300)
    assert np.isfinite(st.arr["pos"]).all()
    assert st.arr["pos"][:, 1].min() > 1.5           # floor top at y = 2, penalty keeps particles near/above it
    assert (st.arr["cdf_affinity"] & 1).any()


def test_fluid_is_the_pressure_only_neo_hookean(oracle_libs):
    """configs[4] (C5): the "weakly-compressible fluid" is src/models/neo_hookean_elasticity.wgsl:14-25 with mu = 0,
    i.e. tau = lambda * ln(max(det F, 1e-10)) * I whatever the shear in F (fp64 restatement)."""
    from oracle.orc import Oracle
    orc = Oracle(3, np.float64)
    rng = np.random.default_rng(17)
    lam = 1.0e7
    for _ in range(20):
        F = np.eye(3) + rng.normal(0, 0.2, (3, 3))
        if np.linalg.det(F) <= 0:
            continue
        tau = orc.kirchoff_stress(1, lam, 0.0, F.T.reshape(-1))          # column-major
        want = lam * np.log(np.linalg.det(F)) * np.eye(3)
        assert np.allclose(np.asarray(tau).reshape(3, 3), want, rtol=1e-12, atol=1e-9 * lam)


def test_fluid_block_scene_is_config_5():
    """BASELINE.json configs[4] / SURVEY 8d C5: 256 x 250 x 250 = 16 M particles, mu = 0, every decomposition generates
    the same particles (checked at a reduced y/z extent: the generator is separable)."""
    from wgsparkl_amd import scenes
    full = scenes.fluid_block(256, 6, 6)
    assert full["global_particles"] == 256 * 36 and full["particles"].n == 256 * 36
    assert 256 * 250 * 250 == 16_000_000
    ps = full["particles"]
    assert np.all(ps.mu == 0.0) and np.all(ps.lambda_ == np.float32(1.0e7)) and full["model"] == 1
    for world in (2, 4, 8):
        seen, counts = [], []
        for r in range(world):
            sc = scenes.fluid_block(256, 6, 6, world=world, rank=r)
            assert np.array_equal(ps.pos[sc["global_ids"]], sc["particles"].pos)
            seen.append(sc["global_ids"])
            counts.append(sc["particles"].n)
        seen = np.concatenate(seen)
        assert len(seen) == ps.n and len(np.unique(seen)) == ps.n
        assert max(counts) == min(counts)                                # slabs balanced by particle count
