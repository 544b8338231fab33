"""-m "not gpu": host-side mirror of the reference's types (models / solver / scenes) and the
AoS packing that crosses the C ABI."""
import ctypes as C

import numpy as np
import pytest

from wgsparkl_amd import scenes
from wgsparkl_amd.models import DruckerPrager, ElasticCoefficients, ParticlePhase, lame_lambda_mu
from wgsparkl_amd.solver import Particle, ParticleDynamics, ParticleSet


def test_lame_parameters():
    """models/mod.rs:52-63"""
    lam, mu = lame_lambda_mu(1.0e5, 0.33)
    assert np.isclose(lam, 1e5 * 0.33 / (1.33 * 0.34), rtol=1e-6)
    assert np.isclose(mu, 1e5 / 2.66, rtol=1e-6)
    e = ElasticCoefficients.from_young_modulus(1.0e7, 0.2)
    assert np.isclose(e.lambda_, 1e7 * 0.2 / (1.2 * 0.6), rtol=1e-6) and np.isclose(e.mu, 1e7 / 2.4, rtol=1e-6)


def test_drucker_prager_defaults():
    """drucker_prager.rs:18-33: angles 35/9/10 degrees, h2 = 0.2; E <= 0 -> lambda = mu = -1."""
    d = DruckerPrager.new(2.0e9, 0.2)
    assert np.isclose(d.h0, np.deg2rad(35)) and np.isclose(d.h1, np.deg2rad(9)) and np.isclose(d.h3, np.deg2rad(10))
    assert d.h2 == 0.2 and d.lambda_ > 0
    off = DruckerPrager.new(-1.0, -1.0)
    assert off.lambda_ == -1.0 and off.mu == -1.0


@pytest.mark.parametrize("dim", [2, 3])
def test_with_density(dim):
    """particle3d.rs:28-42: V0 = (2r)^dim, m = rho V0, F = I, affine = 0."""
    dyn = ParticleDynamics.with_density(0.25, 2700.0, dim)
    assert np.isclose(dyn.init_volume, 0.5 ** dim) and np.isclose(dyn.mass, 2700.0 * 0.5 ** dim)
    assert np.array_equal(dyn.def_grad, np.eye(dim, dtype=np.float32).reshape(-1)) and not dyn.affine.any()


def test_particle_defaults_quirk_b1():
    """models/mod.rs:24,33-36: plasticity None -> DruckerPrager::new(-1,-1); phase None -> {0, -1}."""
    p = Particle(position=np.zeros(3, np.float32), dynamics=ParticleDynamics.with_density(0.25, 1.0),
                 model=ElasticCoefficients.from_young_modulus(1e5, 0.33))
    ps = ParticleSet.from_particles([p])
    assert ps.dp[0, 4] == -1.0 and ps.dp[0, 5] == -1.0
    assert ps.phase[0].tolist() == [0.0, -1.0]
    assert not ps.has_plasticity[0] and not ps.has_phase[0]
    ps2 = ParticleSet.uniform(np.zeros((1, 3), np.float32), 0.25, 1.0, ElasticCoefficients(1.0, 1.0),
                              phase=ParticlePhase(1.0, 3.0e38))
    assert ps2.phase[0].tolist() == [1.0, np.float32(3.0e38)]


@pytest.mark.parametrize("dim", [2, 3])
def test_pack_unpack_roundtrip(hip_libs, dim):
    from wgsparkl_amd.pipeline import _pack_particles, _unpack_particles
    _, T = hip_libs.load(dim)
    ps = scenes.random_cloud(50, dim=dim, seed=3, plasticity=DruckerPrager.new(1e6, 0.3), phase=ParticlePhase(1.0, 2.0))
    ps.cdf_affinity[:] = np.arange(50, dtype=np.uint32) * 0x01010101
    ps.cdf_dist[:] = np.linspace(-1, 1, 50)
    raw = _pack_particles(T, ps)
    assert raw.shape == (50, C.sizeof(T.Particle) // 4)
    back = _unpack_particles(T, raw, dim, ps.dp_state.copy())
    for name in ("pos", "vel", "def_grad", "affine", "cdf_dist", "cdf_affinity", "init_volume", "init_radius",
                 "mass", "lambda_", "mu", "dp", "phase"):
        assert np.array_equal(getattr(back, name), getattr(ps, name)), name
    # spot-check against the ctypes view of the same bytes
    arr = (T.Particle * 50).from_buffer(raw)
    assert arr[7].dynamics.mass == ps.mass[7] and arr[7].model.mu == ps.mu[7]
    assert arr[7].dynamics.cdf.affinity == ps.cdf_affinity[7]
    assert list(arr[7].position) == ps.pos[7].tolist()


def test_scenes_match_the_named_configs():
    c2 = scenes.neo_hookean_cube(n_side=20)
    assert c2["particles"].n == 8000 and c2["model"] == 1 and np.isclose(c2["params"].dt, 1 / 1200)
    # 8 particles per cell: spacing h/2
    assert np.isclose(np.ptp(c2["particles"].pos[:, 0]), 9.5, atol=0.11)
    sm = scenes.reference_smoke_scene()
    assert sm["particles"].n == 1000 and sm["grid_capacity"] == 100_000 and np.isclose(sm["params"].dt, 1 / 600)
    c1 = scenes.elastic_block_2d()
    assert c1["particles"].n == 10_000 and c1["particles"].dim == 2
    c3 = scenes.sand_column(nx=10, ny=12, nz=10)
    assert c3["particles"].has_plasticity.all() and (c3["particles"].phase[:, 0] == 0).all()
