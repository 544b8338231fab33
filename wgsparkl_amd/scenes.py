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
from .solver import SHAPE_CAPSULE, Collider, ParticleSet, SimulationParams

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


def reference_sand3():
    """The reference's shipping 3D scene as written (crates/wgsparkl3d/examples/sand3.rs:28-113): 45 x 100 x 45 = 202 500
    Drucker-Prager sand particles (E = 2e9, nu = 0.2, phase None) at spacing h/2 above a floor cuboid, four wall cuboids
    and one kinematic cuboid (tilted -0.5 rad about z, spinning at -1 rad/s about y); h = 1, dt = 1/1200, capacity 60 000."""
    import math
    h, nxz = 1.0, 45
    i, j, k = np.meshgrid(np.arange(nxz), np.arange(100), np.arange(nxz), indexing="ij")
    pos = (np.stack([i.ravel() + 0.5 - nxz / 2.0, j.ravel() + 0.5 + 10.0, k.ravel() + 0.5 - nxz / 2.0], 1) * (h / 2.0)).astype(F32)
    ps = ParticleSet.uniform(pos, h / 4.0, 2700.0, ElasticCoefficients.from_young_modulus(2.0e9, 0.2),
                             plasticity=DruckerPrager.new(2.0e9, 0.2), phase=None)
    colliders = [Collider.cuboid((100.0, 4.0, 100.0), (0.0, -4.0, 0.0)),
                 Collider.cuboid((35.0, 5.0, 0.5), (0.0, 5.0, -35.0)), Collider.cuboid((35.0, 5.0, 0.5), (0.0, 5.0, 35.0)),
                 Collider.cuboid((0.5, 5.0, 35.0), (-35.0, 5.0, 0.0)), Collider.cuboid((0.5, 5.0, 35.0), (35.0, 5.0, 0.0)),
                 Collider.cuboid((0.5, 2.0, 30.0), (0.0, 2.0, 0.0), rotation=(0.0, 0.0, math.sin(-0.25), math.cos(-0.25)),
                                 angvel=(0.0, -1.0, 0.0))]
    return dict(particles=ps, params=SimulationParams(gravity=(0.0, -9.81, 0.0), dt=(1.0 / 60.0) / 20.0), colliders=colliders,
                cell_width=h, grid_capacity=60_000, model=MODEL_COROTATED,
                name="the reference's sand3 example as shipped: 45x100x45 Drucker-Prager sand, floor + 4 walls + kinematic rotating cuboid",
                bytes_per_particle=216.0)


def reference_sand2():
    """The reference's largest shipping scene as written (crates/wgsparkl2d/examples/sand2.rs:21-175): 700 x 700 = 490 000
    Drucker-Prager sand particles (E = 1e7, nu = 0.2, rho = 1000, phase None) at spacing h/2 with h = 0.2, lifted by 46; a floor
    and two tilted walls (fixed cuboids), three spinning cuboids, a spinning ball and a spinning capsule (kinematic), eight
    dynamic cuboids of density 10 + 100 k dropped from y = 120; dt = 1/600, capacity 60 000. That is 16 coupled colliders,
    the reference's limit (quirk B11). rapier's body-body contacts are not part of the MPM path: the dynamic cuboids move
    under gravity and the sand's impulses."""
    h = 0.2
    i, j = np.meshgrid(np.arange(700), np.arange(700), indexing="ij")
    pos = (np.stack([i.ravel() + 0.5, j.ravel() + 0.5], 1) * (h / 2.0)).astype(F32)
    pos[:, 1] += F32(46.0)
    ps = ParticleSet.uniform(pos, h / 4.0, 1000.0, ElasticCoefficients.from_young_modulus(1.0e7, 0.2),
                             plasticity=DruckerPrager.new(1.0e7, 0.2), phase=None)
    w = 1.0
    colliders = [Collider.cuboid((42.0, 1.0), (35.0, -1.0), rotation=(0.0,)),
                 Collider.cuboid((1.0, 52.0), (-25.0, 45.0), rotation=(0.5,)), Collider.cuboid((1.0, 52.0), (95.0, 45.0), rotation=(-0.5,)),
                 Collider.cuboid((1.0, 10.0), (5.0, 35.0), rotation=(0.0,), angvel=(w,)),
                 Collider.cuboid((10.0, 1.0), (35.0, 35.0), rotation=(0.0,), angvel=(-w,)),
                 Collider.cuboid((1.0, 10.0), (65.0, 35.0), rotation=(0.0,), angvel=(w,)),
                 Collider.ball(5.0, (20.0, 20.0), rotation=(0.0,), angvel=(-w,)),
                 Collider(SHAPE_CAPSULE, (5.0, 3.0), (50.0, 20.0), rotation=(0.0,), angvel=(-w,))]
    for k in range(8):
        colliders.append(Collider.cuboid((5.0, 1.0), (35.0 + 3.0 * k, 120.0), rotation=(0.0,)).with_density(10.0 + 100.0 * k, 2))
    return dict(particles=ps, params=SimulationParams(gravity=(0.0, -9.81), dt=(1.0 / 60.0) / 10.0), colliders=colliders,
                cell_width=h, grid_capacity=60_000, model=MODEL_COROTATED,
                name="the reference's sand2 example as shipped (2D): 700x700 Drucker-Prager sand, h = 0.2, 16 coupled colliders "
                     "(3 fixed, 5 kinematic spinning, 8 dynamic cuboids)",
                bytes_per_particle=144.0)


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


def _slab_block(counts, origin, h, jitter, world, rank, cuts=None):
    """Lattice block of counts[0] x counts[1] x counts[2] particles (spacing h/2, hashed jitter by GLOBAL id) cut into
    `world` x-slabs of (nearly) equal particle count, cuts on block boundaries. With `rank` given only that slab is
    generated. Returns (pos, global ids, SlabPartition)."""
    from .sharded import SlabPartition, associated_block_x
    nx, ny, nz = counts
    ox = origin[0]
    if cuts is None:
        # the block column of lattice plane i is floor((round((ox + (i + .5) h/2) / h) - 1) / 4); cut r sits at the
        # block boundary closest to the plane that splits the particle count evenly
        cuts = [int(np.floor((ox / h - 1.0) / 4.0))]
        for r in range(1, world):
            cuts.append(int(np.round((ox / h + (nx * r / world) / 2.0 - 1.0) / 4.0)))
        cuts.append(int(np.floor((ox / h + nx / 2.0) / 4.0)) + 2)
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1] + 1)
    part = SlabPartition(cuts)
    if rank is None:
        i0, i1 = 0, nx
    else:
        # lattice planes that can fall into this rank's block range (+ a margin for the jitter)
        lo_b, hi_b = part.cuts[rank], part.cuts[rank + 1]
        i0 = 0 if rank == 0 else max(0, int(np.floor(((lo_b * 4 + 0.5) * h - ox) * 2.0 / h)) - 3)
        i1 = nx if rank == world - 1 else min(nx, int(np.ceil(((hi_b * 4 + 1.5) * h - ox) * 2.0 / h)) + 3)
    i, j, k = np.meshgrid(np.arange(i0, i1, dtype=np.int32), np.arange(ny, dtype=np.int32), np.arange(nz, dtype=np.int32), indexing="ij")
    gid = (i.ravel().astype(np.int64) * ny + j.ravel()) * nz + k.ravel()
    pos = np.stack([i.ravel(), j.ravel(), k.ravel()], 1).astype(np.float64)
    del i, j, k
    pos = (pos + 0.5) * (h / 2.0) + np.asarray(origin, np.float64)
    if jitter:
        pos += _hash_jitter(gid, 3, jitter * h)
    pos = pos.astype(F32)
    if rank is not None:
        own = part.owner_of_blocks(associated_block_x(pos, h, 3)) == rank
        pos, gid = pos[own], gid[own]
    return pos, gid.astype(np.uint32), part


def fluid_block(nx=256, ny=250, nz=250, world=1, rank=None, jitter=0.05, cell_width=1.0, with_floor=True,
                density=1000.0, bulk_modulus=1.0e7, grid_capacity=None):
    """C5 (BASELINE.json configs[4]): weakly-compressible "fluid" = pressure-only neo-Hookean
    (mu = 0 in src/models/neo_hookean_elasticity.wgsl:14-25 => tau = lambda * ln(J) * I; the reference has no fluid
    model of its own), nx x ny x nz = 16 M particles at 8 per cell inside a 512^3-cell domain, over the floor cuboid.
    lambda = bulk modulus 1e7 (sound speed 100 cells/s against h/dt = 1200), rho = 1000, dt = 1/1200, phase = 1.
    `world` / `rank`: the x-slab of one rank of a strong-scaling run (global ids = index in the full lattice, jitter
    hashed from the id, so every decomposition generates the very same particles)."""
    h = cell_width
    # x origin 16.5 h: lattice planes 2i, 2i + 1 sit at (16.75 + i) h and (17.25 + i) h, both inside cell 16 + i, so
    # 8 planes fill a block column exactly and equal block ranges are equal particle counts
    origin = (16.5 * h, 8.0 * h, 20.0 * h)
    pos, gid, part = _slab_block((nx, ny, nz), origin, h, jitter, world, rank)
    ps = ParticleSet.uniform(pos, h / 4.0, density, ElasticCoefficients(bulk_modulus, 0.0),
                             phase=ParticlePhase(1.0, FLT_MAX))
    colliders = [Collider.cuboid((100000.0 * h, 2.0 * h, 100000.0 * h), (0.0, 0.0, 0.0))] if with_floor else []
    if grid_capacity is None:
        slab_nx = nx if rank is None else -(-nx // world) + 16
        nb = (slab_nx // 8 + 4) * (ny // 8 + 4) * (nz // 8 + 4)
        grid_capacity = max(1024, 1 << int(np.ceil(np.log2(nb * 1.5))))
    return dict(particles=ps, global_ids=gid, partition=part,
                params=SimulationParams(gravity=(0.0, -9.81, 0.0), dt=1.0 / 1200.0), colliders=colliders,
                cell_width=h, grid_capacity=grid_capacity, model=MODEL_NEO_HOOKEAN, global_particles=nx * ny * nz)


def config_scene(config="c2", world=1, rank=None, scaling="weak", n_side=None, jitter=0.05):
    """The BASELINE.json configs as (possibly sharded) scenes for bench.py: `c2` neo-Hookean cube (1 M), `c3`
    Drucker-Prager sand column standing between the floor and four walls (4 M), `c4` corotated cube + kinematic rotating
    cuboid (8 M), `c5` pressure-only neo-Hookean fluid block (16 M). world > 1: "strong" cuts the named size into `world` x-slabs, "weak" puts `world` copies side by
    side along x (fixed work per GPU). With `rank` given only that rank's slab is generated. Particles are identified
    by their index in the global lattice and jittered by a hash of it, so every decomposition simulates the same scene."""
    h = 1.0
    mult = world if scaling == "weak" else 1
    if config == "c5":
        nx, ny, nz = (256, 250, 250) if n_side is None else (n_side, n_side, n_side)
        sc = fluid_block(nx * mult, ny, nz, world=world, rank=rank, jitter=jitter)
        sc["name"] = f"wgsparkl3d weakly-compressible fluid (pressure-only neo-Hookean, mu = 0), {nx * mult}x{ny}x{nz} particles, 512^3-cell domain, floor cuboid"
        sc["bytes_per_particle"] = 160.0
        return sc
    if config == "c2":
        n = 100 if n_side is None else n_side
        counts, origin = (n * mult, n, n), (20.0 * h, 8.0 * h, 20.0 * h)
        model, elastic = MODEL_NEO_HOOKEAN, ElasticCoefficients.from_young_modulus(1.0e7, 0.2)
        plast, phase = None, ParticlePhase(1.0, FLT_MAX)
        colliders = [Collider.cuboid((100000.0 * h, 2.0 * h, 1000.0 * h), (0.0, 0.0, 0.0))]
        name = f"wgsparkl3d neo-Hookean elastic cube, {n * mult}x{n}x{n} particles, 128^3-cell domain, floor cuboid"
        bpp = 160.0
    elif config == "c3":
        nx, ny, nz = (100, 400, 100) if n_side is None else (n_side, 4 * n_side, n_side)
        counts, origin = (nx * mult, ny, nz), (20.0 * h, 2.2 * h, 20.0 * h)      # standing on the floor (top face y = 2)
        model, elastic = MODEL_COROTATED, ElasticCoefficients.from_young_modulus(2.0e9, 0.2)
        plast, phase = DruckerPrager.new(2.0e9, 0.2), None
        x0, x1 = origin[0] - 1.5 * h, origin[0] + counts[0] * h / 2.0 + 1.5 * h
        z0, z1 = origin[2] - 1.5 * h, origin[2] + nz * h / 2.0 + 1.5 * h
        t = 2.0 * h
        colliders = [Collider.cuboid((100000.0, 2.0, 1000.0), (0.0, 0.0, 0.0)),
                     Collider.cuboid((t, 1000.0, 1000.0), (x0 - t, 0.0, 0.0)), Collider.cuboid((t, 1000.0, 1000.0), (x1 + t, 0.0, 0.0)),
                     Collider.cuboid((100000.0, 1000.0, t), (0.0, 0.0, z0 - t)), Collider.cuboid((100000.0, 1000.0, t), (0.0, 0.0, z1 + t))]
        name = f"wgsparkl3d Drucker-Prager sand column, {nx * mult}x{ny}x{nz} particles, 256^3-cell domain, standing between the floor and four walls"
        bpp = 216.0
    elif config == "c4":
        # BASELINE.json configs[3]: corotated elastic cube resting on the floor, hit by one kinematic rotating cuboid — a body
        # whose velocity the host sets and whose pose the device integrates every substep (the pattern of
        # crates/wgsparkl3d/examples/sand3.rs:95-103); same geometry as corotated_cube_with_paddle, by global lattice
        # index so that it can be cut into x-slabs (every rank holds every collider).
        n = 200 if n_side is None else n_side
        counts, origin = (n * mult, n, n), (20.0 * h, 2.4 * h, 20.0 * h)
        model, elastic = MODEL_COROTATED, ElasticCoefficients.from_young_modulus(1.0e7, 0.2)
        plast, phase = None, ParticlePhase(1.0, FLT_MAX)
        sx, side = counts[0] * h / 2.0, n * h / 2.0
        colliders = [Collider.cuboid((100000.0 * h, 2.0 * h, 1000.0 * h), (0.0, 0.0, 0.0)),
                     Collider.cuboid((2.0 * h, side / 2.0, side / 3.0), (origin[0] + sx + 2.2 * h, origin[1] + side / 2.0, origin[2] + side / 2.0),
                                     angvel=(0.0, 0.8, 0.0), linvel=(-2.0, 0.0, 0.0))]
        name = (f"wgsparkl3d corotated elastic cube on the floor + kinematic rotating cuboid (rigid-body collision), {n * mult}x{n}x{n} particles, "
                "256^3-cell domain")
        bpp = 160.0
    else:
        raise ValueError(f"unknown config {config!r}")
    pos, gid, part = _slab_block(counts, origin, h, jitter, world, rank)
    ps = ParticleSet.uniform(pos, h / 4.0, 2700.0, elastic, plasticity=plast, phase=phase)
    slab_nx = counts[0] if rank is None else -(-counts[0] // world) + 16
    nb = (slab_nx // 8 + 4) * (counts[1] // 8 + 4) * (counts[2] // 8 + 4)
    return dict(particles=ps, global_ids=gid, partition=part, params=SimulationParams(gravity=(0.0, -9.81, 0.0), dt=1.0 / 1200.0),
                colliders=colliders, cell_width=h, grid_capacity=max(1024, 1 << int(np.ceil(np.log2(nb * 1.5)))), model=model,
                global_particles=counts[0] * counts[1] * counts[2], name=name, bytes_per_particle=bpp)
