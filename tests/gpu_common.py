"""Shared by the -m gpu test files: tolerances, seeded scenes, the oracle comparisons of block sets / particle fields / grid."""
import numpy as np
import pytest

from wgsparkl_amd import scenes
from wgsparkl_amd.models import (MODEL_COROTATED, MODEL_NEO_HOOKEAN, DruckerPrager, ElasticCoefficients, ParticlePhase)
from wgsparkl_amd.solver import Collider, ParticleSet, SimulationParams

from helpers import assert_close_to_truth, compare_cpic, compare_grids, grid_of, max_abs, rel_rms, report_margin, run_gpu, run_oracle

GRID_V_TOL = 1e-5      # relative RMS of grid velocity vs the fp64 oracle (north_star target)
PART_TOL = 2e-5        # relative RMS of particle x, v, F, C' vs the fp64 oracle
# Collider (CPIC) scenes: discrete decisions (affinity / sign bits, det M > 1e-8, closest collider) sit on fp32
# thresholds, so a handful of particles may land on the other side; the comparison runs over the particles whose
# affinity bits agree, and the measured margins are reported (helpers.report_margin -> profiles/rNN_parity_margins.json)
CPIC_GRID_V_TOL = 5e-5
CPIC_PART_TOL = 5e-5
# fuzz scenes (random colliders of every kind, some dynamic, 12 substeps, against the fp32 oracle)
FUZZ_NODE_MISMATCH = 0.001     # measured: 0 in all 32 fuzz cases (profiles/r02_parity_margins.json)
FUZZ_PART_MISMATCH = 0.002     # measured: 0
FUZZ_VEL_TOL = 1e-4            # measured worst: 2.5e-5 (12 substeps of fp32 round-off growth through contact)
FUZZ_BODY_ATOL = 1e-4          # measured worst: 9.3e-6 (fixed-point impulses: 1e-5 resolution)


def cloud_scene(n=20000, dim=3, model=MODEL_COROTATED, seed=7, **kw):
    ps = scenes.random_cloud(n, dim=dim, seed=seed, phase=ParticlePhase(1.0, -1.0), **kw)
    g = (0.0, -9.81, 0.0)[:dim]
    return dict(particles=ps, params=SimulationParams(gravity=g, dt=1.0e-3), colliders=[], cell_width=1.0,
                grid_capacity=4096, model=model)


def check_blocks(data, st):
    vid, first, num, ids = data.read_blocks()
    ovid, ofirst, onum = st.blocks()
    assert np.array_equal(vid, ovid), "active block sets differ"
    assert np.array_equal(num, onum), "per-block particle counts differ"
    # sorted ids: same particles in each block (order inside a block is free in the reference)
    osorted = st.g["sorted_ids"][:st.n]
    of = st.g["first_particle"][:st.n_blocks]
    on = st.g["num_particles"][:st.n_blocks]
    ov = st.g["block_vid"][:st.n_blocks]
    oracle_sets = {tuple(ov[b]): frozenset(osorted[of[b]:of[b] + on[b]].tolist()) for b in range(st.n_blocks)}
    for b in range(len(vid)):
        got = frozenset(ids[first[b]:first[b] + num[b]].tolist())
        assert got == oracle_sets[tuple(vid[b])]
    assert sorted(ids.tolist()) == list(range(st.n))


def check_fields(data, st32, st64, tol=PART_TOL):
    got = data.read_particles()
    for name in ("pos", "vel", "def_grad", "affine"):
        assert_close_to_truth(name, getattr(got, name), st32.arr[name], st64.arr[name], tol)
    return got


def check_grid(data, st32, st64, dim=3):
    gv, ov = compare_grids(data.read_grid(), grid_of(st64))
    o32 = grid_of(st32)[1]
    assert_close_to_truth("grid velocity", gv[:, :dim], o32[:, :dim], ov[:, :dim], GRID_V_TOL)
    assert_close_to_truth("grid mass", gv[:, dim], o32[:, dim], ov[:, dim], GRID_V_TOL)


def _exploding_cube():
    sc = scenes.neo_hookean_cube(n_side=8)
    ps = sc["particles"]
    c = ps.pos.mean(0)
    ps.vel[:] = ((ps.pos - c) * 25.0).astype(np.float32)       # radial: the cube flies apart
    ps.lambda_[:] = 1.0                                         # (next to no stiffness: nothing holds it together)
    ps.mu[:] = 1.0
    sc["params"] = SimulationParams(gravity=(0.0, 0.0, 0.0), dt=sc["params"].dt)
    return sc


def _random_scene(seed):
    """Seeded random configuration: dimension, material / plasticity, 0-3 colliders of random kind (ball, cuboid,
    capsule, mesh), pose and motion, some of them dynamic."""
    rng = np.random.default_rng(1000 + seed)
    dim = 3 if seed % 2 == 0 else 2
    plastic = DruckerPrager.new(1e6, 0.25) if rng.random() < 0.4 else None
    phase = None if (plastic is not None and rng.random() < 0.5) else ParticlePhase(1.0, -1.0)
    ps = scenes.random_cloud(1200, dim=dim, seed=100 + seed, extent=9.0, young=1e6, plasticity=plastic, phase=phase,
                             vel_scale=1.5, perturb_F=0.02, perturb_C=0.2)
    cols = []
    for _ in range(int(rng.integers(0, 4))):
        kind = int(rng.integers(0, 4))
        pos = tuple(float(x) for x in rng.uniform(1.0, 9.0, dim))
        vel = tuple(float(x) for x in rng.uniform(-1.0, 1.0, 3))
        if dim == 3:
            axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
            ang = float(rng.uniform(0, 1.5))
            rot = tuple(float(x) for x in np.append(axis * np.sin(ang / 2), np.cos(ang / 2)))
            angvel = tuple(float(x) for x in rng.uniform(-0.8, 0.8, 3))
        else:
            rot = (float(rng.uniform(0, 1.5)),)
            angvel = (float(rng.uniform(-0.8, 0.8)),)
        kw = dict(rotation=rot, linvel=vel, angvel=angvel)
        if kind == 0:
            c = Collider.ball(float(rng.uniform(0.8, 2.0)), pos, **kw)
        elif kind == 1:
            c = Collider.cuboid(tuple(float(x) for x in rng.uniform(0.6, 2.5, dim)), pos, **kw)
        elif kind == 2:
            c = Collider(2, (float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.4, 1.0))), pos, **kw)   # capsule
        elif dim == 3:
            v = np.array([[-2.3, 0.1, -2.1], [-2.2, 0.0, 2.4], [2.1, 0.3, -2.2], [2.4, -0.2, 2.3]], np.float32)
            c = Collider.trimesh(v, np.array([[0, 1, 2], [2, 1, 3]]), pos, **kw)
        else:
            v = np.array([[-3.1, 0.2], [-0.4, -0.3], [2.9, 0.4]], np.float32)
            c = Collider.polyline(v, np.array([[0, 1], [1, 2]]), pos, **kw)
        if kind in (0, 1) and rng.random() < 0.5:
            c = c.with_density(float(rng.uniform(5.0, 50.0)), dim)
        cols.append(c)
    g = (0.0, -9.81, 0.0)[:dim]
    return dict(particles=ps, params=SimulationParams(gravity=g, dt=8e-4), colliders=cols, cell_width=1.0,
                grid_capacity=2048, model=int(rng.integers(0, 2)))


def _native_slabs(sc, world, pipe, **kw):
    """The scene cut into `world` x-slabs balanced by particle count, each a NativeShard of a lockstep group."""
    from wgsparkl_amd.sharded import NativeShard, SlabPartition, associated_block_x, split_scene, uniform_material_of
    ps = sc["particles"]
    part = SlabPartition.balanced(associated_block_x(ps.pos, sc["cell_width"], ps.dim), world)
    shards = []
    for r, (sub, gids) in enumerate(split_scene(ps, part, sc["cell_width"])):
        lo, hi = part.block_range(r)
        shards.append(NativeShard(pipe, sc["params"], sub, gids, sc["colliders"], sc["cell_width"], sc["grid_capacity"],
                                  lo, hi, r > 0, r < world - 1, particle_capacity=ps.n, model=sc["model"],
                                  uniform_material=uniform_material_of(ps), **kw))
    return shards, part

