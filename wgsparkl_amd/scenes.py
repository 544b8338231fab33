"""Synthetic scenes (the workloads of BASELINE.json / SURVEY.md §8d).

Conventions follow the reference's example scenes: lattice spacing h/2
(8 particles per cell in 3D, 4 in 2D), radius h/4, `with_density`
(crates/wgsparkl3d/examples/sand3.rs:28-49, crates/wgsparkl2d/examples/elasticity2.rs:33-55),
plus a reproducible jitter so cells are not degenerate.
"""
from __future__ import annotations

import numpy as np

from .models import (MODEL_COROTATED, MODEL_NEO_HOOKEAN, DruckerPrager,
                     ElasticCoefficients, ParticlePhase)
from .solver import Collider, ParticleSet, SimulationParams

F32 = np.float32
FLT_MAX = float(np.finfo(np.float32).max)


def lattice(counts, origin, cell_width, jitter=0.05, seed=1234):
    """Particles at spacing h/2 starting at `origin` (+h/4), uniform jitter ±jitter*h."""
    dim = len(counts)
    axes = [np.arange(c, dtype=np.float64) for c in counts]
    grid = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, dim)
    pos = (grid + 0.5) * (cell_width / 2.0) + np.asarray(origin, np.float64)
    if jitter:
        rng = np.random.default_rng(seed)
        pos = pos + rng.uniform(-jitter * cell_width, jitter * cell_width, size=pos.shape)
    return pos.astype(F32)


def reference_smoke_scene():
    """The scene of the reference's own smoke tests (src/pipeline.rs:302-331,
    src/grid/grid.rs:355-370): 10^3 particles at i/2, r = h/4, rho = 1,
    E = 1e5, nu = 0.33, plasticity None, phase None (quirk B1), g = (0,-9.81,0),
    dt = 1/600, h = 1, no colliders, capacity 100_000."""
    h = 1.0
    idx = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(10), indexing="ij"), -1).reshape(-1, 3)
    pos = (idx.astype(F32) / F32(h)) / F32(2.0)
    ps = ParticleSet.uniform(pos, h / 4.0, 1.0, ElasticCoefficients.from_young_modulus(100_000.0, 0.33))
    params = SimulationParams(gravity=(0.0, -9.81, 0.0), dt=float(F32(1.0 / 60.0) / F32(10.0)))
    return dict(particles=ps, params=params, colliders=[], cell_width=h, grid_capacity=100_000,
                model=MODEL_COROTATED)


def elastic_block_2d(nx=100, ny=100, with_floor=True, jitter=0.05):
    """C1: wgsparkl2d elastic block, 10k particles, 64x64 grid (8x8 blocks), corotated."""
    h = 1.0
    pos = lattice((nx, ny), (7.0, 7.0), h, jitter)
    ps = ParticleSet.uniform(pos, h / 4.0, 1000.0, ElasticCoefficients.from_young_modulus(5.0e6, 0.2),
                             phase=ParticlePhase(1.0, FLT_MAX))
    colliders = [Collider.cuboid((1000.0, 1.0), (0.0, 1.0), rotation=(0.0,))] if with_floor else []
    params = SimulationParams(gravity=(0.0, -9.81), dt=1.0 / 900.0)
    return dict(particles=ps, params=params, colliders=colliders, cell_width=h, grid_capacity=256,
                model=MODEL_COROTATED)


def neo_hookean_cube(n_side=100, with_floor=False, jitter=0.05, cell_width=1.0, grid_capacity=None):
    """C2: n_side^3 particles (n_side/2)^3 cells inside a 128^3-cell domain, neo-Hookean,
    E = 1e7, nu = 0.2, rho = 2700, phase = 1 (never fractures), dt = 1/1200."""
    h = cell_width
    origin = (20.0 * h, 8.0 * h, 20.0 * h)
    pos = lattice((n_side,) * 3, origin, h, jitter)
    ps = ParticleSet.uniform(pos, h / 4.0, 2700.0, ElasticCoefficients.from_young_modulus(1.0e7, 0.2),
                             phase=ParticlePhase(1.0, FLT_MAX))
    colliders = [Collider.cuboid((1000.0 * h, 2.0 * h, 1000.0 * h), (0.0, 0.0, 0.0))] if with_floor else []
    params = SimulationParams(gravity=(0.0, -9.81, 0.0), dt=1.0 / 1200.0)
    if grid_capacity is None:
        nb = (n_side // 8 + 3) ** 3
        grid_capacity = max(1024, 1 << int(np.ceil(np.log2(nb * 1.5))))
    return dict(particles=ps, params=params, colliders=colliders, cell_width=h,
                grid_capacity=grid_capacity, model=MODEL_NEO_HOOKEAN)


def sand_column(nx=100, ny=400, nz=100, with_floor=False, jitter=0.05, grid_capacity=None):
    """C3: Drucker-Prager sand column (sand3.rs:45-46 material), corotated stress, phase None."""
    h = 1.0
    origin = (20.0, 8.0, 20.0)
    pos = lattice((nx, ny, nz), origin, h, jitter)
    ps = ParticleSet.uniform(pos, h / 4.0, 2700.0, ElasticCoefficients.from_young_modulus(2.0e9, 0.2),
                             plasticity=DruckerPrager.new(2.0e9, 0.2), phase=None)
    colliders = [Collider.cuboid((1000.0, 2.0, 1000.0), (0.0, 0.0, 0.0))] if with_floor else []
    params = SimulationParams(gravity=(0.0, -9.81, 0.0), dt=1.0 / 1200.0)
    if grid_capacity is None:
        nb = (nx // 8 + 3) * (ny // 8 + 3) * (nz // 8 + 3)
        grid_capacity = max(1024, 1 << int(np.ceil(np.log2(nb * 1.5))))
    return dict(particles=ps, params=params, colliders=colliders, cell_width=h,
                grid_capacity=grid_capacity, model=MODEL_COROTATED)


def random_cloud(n, dim=3, extent=12.0, cell_width=1.0, seed=7, vel_scale=1.0,
                 young=1.0e5, nu=0.3, density=10.0, plasticity=None, phase=None,
                 perturb_F=0.05, perturb_C=0.5):
    """Unstructured test cloud with random velocities, F and APIC matrices
    (exercises every term of P2G / G2P; ragged cells, empty cells)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(1.5 * cell_width, extent * cell_width, size=(n, dim)).astype(F32)
    ps = ParticleSet.uniform(pos, cell_width / 4.0, density,
                             ElasticCoefficients.from_young_modulus(young, nu),
                             plasticity=plasticity, phase=phase)
    ps.vel[:] = rng.normal(0.0, vel_scale, size=(n, dim)).astype(F32)
    eye = np.eye(dim, dtype=F32).reshape(-1)
    ps.def_grad[:] = eye + rng.normal(0.0, perturb_F, size=(n, dim * dim)).astype(F32)
    ps.affine[:] = (rng.normal(0.0, perturb_C, size=(n, dim * dim)) * ps.mass[:, None]).astype(F32)
    ps.mass[:] = (ps.mass * rng.uniform(0.5, 1.5, size=n)).astype(F32)
    return ps
