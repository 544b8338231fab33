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


def corotated_cube_with_paddle(n_side=200, jitter=0.05, cell_width=1.0, paddle_speed=0.8):
    """C4 (single-GPU form): n_side^3 corotated elastic particles on the floor cuboid, hit by one kinematic rotating
    cuboid (the `sand3.rs:95-103` pattern: a body whose angular velocity is set by the host and whose pose the
    device integrates every substep)."""
    sc = neo_hookean_cube(n_side=n_side, with_floor=True, jitter=jitter, cell_width=cell_width)
    h = cell_width
    sc["model"] = MODEL_COROTATED
    side = n_side * h / 2.0
    sc["particles"].pos[:, 1] -= 5.6 * h                       # resting on the floor (top face at y = 2 h)
    centre = (20.0 * h + side + 2.2 * h, 2.4 * h + side / 2.0, 20.0 * h + side / 2.0)
    sc["colliders"].append(Collider.cuboid((2.0 * h, side / 2.0, side / 3.0), centre, angvel=(0.0, paddle_speed, 0.0),
                                           linvel=(-2.0, 0.0, 0.0)))
    return sc


def sand_column(nx=100, ny=400, nz=100, with_floor=False, jitter=0.05, grid_capacity=None, with_walls=False):
    """C3: Drucker-Prager sand column (sand3.rs:45-46 material), corotated stress, phase None. `with_walls`: the floor
    and four walls of SURVEY 8d C3 (five cuboid colliders; the walls stand 1.5 cells off the column's faces)."""
    h = 1.0
    origin = (20.0, 8.0, 20.0)
    pos = lattice((nx, ny, nz), origin, h, jitter)
    ps = ParticleSet.uniform(pos, h / 4.0, 2700.0, ElasticCoefficients.from_young_modulus(2.0e9, 0.2),
                             plasticity=DruckerPrager.new(2.0e9, 0.2), phase=None)
    colliders = [Collider.cuboid((1000.0, 2.0, 1000.0), (0.0, 0.0, 0.0))] if (with_floor or with_walls) else []
    if with_walls:
        x0, x1 = origin[0] - 1.5 * h, origin[0] + nx * h / 2.0 + 1.5 * h
        z0, z1 = origin[2] - 1.5 * h, origin[2] + nz * h / 2.0 + 1.5 * h
        t = 2.0 * h                                                # wall half thickness
        colliders += [Collider.cuboid((t, 1000.0, 1000.0), (x0 - t, 0.0, 0.0)), Collider.cuboid((t, 1000.0, 1000.0), (x1 + t, 0.0, 0.0)),
                      Collider.cuboid((1000.0, 1000.0, t), (0.0, 0.0, z0 - t)), Collider.cuboid((1000.0, 1000.0, t), (0.0, 0.0, z1 + t))]
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


def _hash_jitter(ids: np.ndarray, dim: int, amplitude: float) -> np.ndarray:
    """Deterministic per-particle jitter from the GLOBAL particle id (so that every rank of a sharded run
    generates exactly the particles a single-domain run would): splitmix64 -> uniform in [-a, a)."""
    out = np.empty((len(ids), dim), np.float64)
    for k in range(dim):
        z = (ids.astype(np.uint64) * np.uint64(dim) + np.uint64(k) + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        out[:, k] = (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return (out * 2.0 - 1.0) * amplitude


def neo_hookean_bar(n_side=100, world=1, rank=None, jitter=0.05, cell_width=1.0):
    """Weak-scaling workload: `world` C2 cubes side by side along x = one elastic bar of
    (n_side * world) x n_side x n_side particles. With `rank` given, only the particles of that rank's
    x-slab are generated (global ids = index in the full lattice); returns the slab partition too."""
    from .sharded import SlabPartition, associated_block_x
    h = cell_width
    ox, oy, oz = 20.0 * h, 8.0 * h, 20.0 * h
    nx = n_side * world
    # slab cuts in blocks (4 cells): the block containing the first lattice plane of each rank
    cuts = [int(np.floor((ox / h - 1.0) / 4.0))]
    for r in range(1, world):
        cuts.append(int(np.round((ox / h + n_side * r / 2.0 - 1.0) / 4.0)))
    cuts.append(int(np.floor((ox / h + nx / 2.0) / 4.0)) + 2)
    part = SlabPartition(cuts)
    if rank is None:
        i0, i1 = 0, nx
    else:
        i0, i1 = max(0, n_side * rank - 12), min(nx, n_side * (rank + 1) + 12)
    i, j, k = np.meshgrid(np.arange(i0, i1), np.arange(n_side), np.arange(n_side), indexing="ij")
    idx = np.stack([i.ravel(), j.ravel(), k.ravel()], 1)
    gid = ((idx[:, 0].astype(np.int64) * n_side + idx[:, 1]) * n_side + idx[:, 2])
    pos = (idx + 0.5) * (h / 2.0) + np.array([ox, oy, oz])
    if jitter:
        pos = pos + _hash_jitter(gid, 3, jitter * h)
    pos = pos.astype(F32)
    if rank is not None:
        own = part.owner_of_blocks(associated_block_x(pos, h, 3)) == rank
        pos, gid = pos[own], gid[own]
    ps = ParticleSet.uniform(pos, h / 4.0, 2700.0, ElasticCoefficients.from_young_modulus(1.0e7, 0.2),
                             phase=ParticlePhase(1.0, FLT_MAX))
    nb = (n_side // 8 + 4) ** 2 * (n_side // 8 + 8)
    return dict(particles=ps, global_ids=gid.astype(np.uint32), partition=part,
                params=SimulationParams(gravity=(0.0, -9.81, 0.0), dt=1.0 / 1200.0),
                colliders=[Collider.cuboid((100000.0 * h, 2.0 * h, 1000.0 * h), (0.0, 0.0, 0.0))],
                cell_width=h, grid_capacity=max(1024, 1 << int(np.ceil(np.log2(nb * 1.5)))), model=MODEL_NEO_HOOKEAN,
                global_particles=nx * n_side * n_side)
