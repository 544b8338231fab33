"""-m gpu: one scene cut into x-slabs (kernels_shard.h, capi_sharded.inc) on one GPU — lockstep groups with device copies as the transport, one
rank over a real RCCL communicator as its own neighbours — against the single-domain run of the same scene."""
import numpy as np
import pytest

from wgsparkl_amd import scenes
from wgsparkl_amd.models import (MODEL_COROTATED, MODEL_NEO_HOOKEAN, DruckerPrager, ElasticCoefficients, ParticlePhase)
from wgsparkl_amd.solver import Collider, ParticleSet, SimulationParams

from helpers import assert_close_to_truth, compare_cpic, compare_grids, grid_of, max_abs, rel_rms, report_margin, run_gpu, run_oracle
from gpu_common import (CPIC_GRID_V_TOL, CPIC_PART_TOL, FUZZ_BODY_ATOL, FUZZ_NODE_MISMATCH, FUZZ_PART_MISMATCH, FUZZ_VEL_TOL, GRID_V_TOL, PART_TOL,
                        _exploding_cube, _native_slabs, _random_scene, check_blocks, check_fields, check_grid, cloud_scene)
import os as _os

from golden_cases import CASES as _CASES

pytestmark = pytest.mark.gpu


def test_bench_decomposition_eight_ranks_on_one_gpu(hip_libs):
    """The N = 8 workload of bench.py (one elastic bar cut into 8 x-slabs, every rank generating only its own slab,
    the floor collider, bench.py's buffer capacities) advanced as a lockstep group on one GPU — wgs_sharded_step_lockstep:
    the per-phase code of wgs_sharded_step with device-to-device copies as the transport —: same particles as the
    single-domain run of the whole bar, none lost."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    from wgsparkl_amd.sharded import NativeShard, native_lockstep, uniform_material_of
    world, n_side, k = 8, 24, 40
    pipe = pipeline(3)
    full = scenes.neo_hookean_bar(n_side=n_side, world=world, rank=None)
    vx = lambda gid: (8.0 + 3.0 * np.sin(0.37 * gid.astype(np.float64))).astype(np.float32)   # particles cross the faces
    full["particles"].vel[:, 0] = vx(full["global_ids"])
    ref_data = MpmData.new(pipe, full["params"], full["particles"], full["colliders"], full["cell_width"],
                           full["grid_capacity"] * 4, full["model"])
    pipe.step(ref_data, k)
    ref = ref_data.read_particles()
    shards, total = [], 0
    for rank in range(world):
        sc = scenes.neo_hookean_bar(n_side=n_side, world=world, rank=rank)
        ps = sc["particles"]
        ps.vel[:, 0] = vx(sc["global_ids"])
        total += ps.n
        lo, hi = sc["partition"].block_range(rank)
        shards.append(NativeShard(pipe, sc["params"], ps, sc["global_ids"], sc["colliders"], sc["cell_width"],
                                  sc["grid_capacity"], lo, hi, rank > 0, rank < world - 1,
                                  particle_capacity=int(ps.n * 1.25) + 4096, model=sc["model"], uniform_material=uniform_material_of(ps),
                                  halo_capacity_records=2 * ((n_side // 8 + 3) ** 2 + 32), migrant_capacity=512))
    assert total == full["global_particles"] == full["particles"].n
    n0 = [s.num_particles() for s in shards]
    native_lockstep(pipe, shards, k)
    for s in shards:
        s.sync()                                             # would report a message / capacity overflow
    outs = [s.export() for s in shards]
    assert [len(o["ids"]) for o in outs] != n0, "particles must have crossed the faces"
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.sort(full["global_ids"]))
    order = np.argsort(ids)
    ref_order = np.argsort(full["global_ids"])
    for f in ("pos", "vel"):
        got = np.concatenate([o[f] for o in outs])[order]
        err = rel_rms(got, getattr(ref, f)[ref_order])
        report_margin(f"bench decomposition, 8 slabs, {f}", err, 1e-5)
        assert err < 1e-5, f


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_sharded_match_single_domain(hip_libs, seed):
    """Fuzz-style check of the decomposition: random clouds (stretched along x so that every slab is a few blocks wide)
    with kinematic analytic colliders, cut into 2-4 slabs, advanced as a lockstep group (wgs_sharded_step_lockstep),
    against the single-domain run on the same GPU."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import native_lockstep
    rng = np.random.default_rng(500 + seed)
    dim = 3 if seed % 2 == 0 else 2
    world = int(rng.integers(2, 5))
    stretch = 3.0 if dim == 3 else 5.0
    ps = scenes.random_cloud(4000, dim=dim, seed=300 + seed, extent=22.0, young=1e6, phase=ParticlePhase(1.0, -1.0),
                             vel_scale=2.5, perturb_F=0.02, perturb_C=0.2)
    ps.pos[:, 0] *= np.float32(stretch)
    ps.vel[:, 0] += np.float32(rng.uniform(-6.0, 6.0))      # a drift: particles cross the cuts
    cols = []
    for _ in range(int(rng.integers(0, 3))):
        pos = [float(x) for x in rng.uniform(2.0, 20.0, dim)]
        pos[0] *= stretch
        kw = dict(linvel=tuple(float(x) for x in rng.uniform(-1.0, 1.0, 3)),
                  angvel=tuple(float(x) for x in rng.uniform(-0.5, 0.5, 3 if dim == 3 else 1)))
        cols.append(Collider.ball(float(rng.uniform(1.0, 3.0)), tuple(pos), **kw) if rng.random() < 0.5 else
                    Collider.cuboid(tuple(float(x) for x in rng.uniform(1.0, 4.0, dim)), tuple(pos), **kw))
    g = (0.0, -9.81, 0.0)[:dim]
    sc = dict(particles=ps, params=SimulationParams(gravity=g, dt=8e-4), colliders=cols, cell_width=1.0,
              grid_capacity=4096, model=int(rng.integers(0, 2)))
    k = 30
    ref = run_gpu(sc, k).read_particles()
    pipe = pipeline(dim)
    shards, part = _native_slabs(sc, world, pipe)
    assert part.min_interior_width() >= 3
    n0 = [s.num_particles() for s in shards]
    native_lockstep(pipe, shards, k)
    for s in shards:
        s.sync()
    outs = [s.export() for s in shards]
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.arange(ps.n, dtype=np.uint32))
    order = np.argsort(ids)
    for f in ("pos", "vel"):
        got = np.concatenate([o[f] for o in outs])[order]
        tol = 1e-5 if f == "pos" else 2e-4
        err = rel_rms(got, getattr(ref, f))
        report_margin(f"sharded fuzz {f}", err, tol, migrated=bool([len(o["ids"]) for o in outs] != n0))
        assert err < tol, f


def test_sharded_run_with_kinematic_collider(hip_libs):
    """configs[3]'s decomposition on one GPU at a small size: the corotated bar on the floor, cut into 4 slabs, and the
    kinematic rotating cuboid at its end, which every rank integrates identically; particles, CPIC state and the body
    pose match the single-domain run (the bar slides towards the cuboid, so particles cross the cuts)."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import NativeShard, native_lockstep, uniform_material_of
    k, world, n = 30, 4, 24

    def c4(rank):
        sc = scenes.config_scene("c4", world, rank, "weak", n_side=n)
        sc["particles"].vel[:, 0] = (16.0 + 2.0 * np.sin(0.37 * sc["global_ids"].astype(np.float64))).astype(np.float32)
        return sc
    full = c4(None)
    single = run_gpu(full, k)
    ref, ref_body = single.read_particles(), single.read_body_poses()
    assert ((ref.cdf_affinity & 2) != 0).sum() > 100       # the paddle does touch the bar
    pipe = pipeline(3)
    shards = []
    for r in range(world):
        sc = c4(r)
        lo, hi = sc["partition"].block_range(r)
        shards.append(NativeShard(pipe, sc["params"], sc["particles"], sc["global_ids"], sc["colliders"], sc["cell_width"], sc["grid_capacity"],
                                  lo, hi, r > 0, r < world - 1, particle_capacity=full["particles"].n, model=sc["model"],
                                  uniform_material=uniform_material_of(sc["particles"]), halo_capacity_records=512, migrant_capacity=2048))
    n0 = [s.num_particles() for s in shards]
    native_lockstep(pipe, shards, k)
    for s in shards:
        s.sync()
    outs = [s.export() for s in shards]
    assert [len(o["ids"]) for o in outs] != n0
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.sort(full["global_ids"]))
    order, ref_order = np.argsort(ids), np.argsort(full["global_ids"])
    for f in ("pos", "vel"):
        got = np.concatenate([o[f] for o in outs])[order]
        err = rel_rms(got, getattr(ref, f)[ref_order])
        report_margin(f"sharded c4 {f}", err, 1e-5)
        assert err < 1e-5, f
    for s in shards:                                        # every rank holds the same body state
        b = s.read_body_poses()
        assert np.allclose(b[1]["rotation"], ref_body[1]["rotation"], atol=1e-6)
        assert np.allclose(b[1]["translation"], ref_body[1]["translation"], atol=1e-5)


@pytest.mark.parametrize("name", ["dynamic_ball3d", "dynamic_ball2d"])
def test_dynamic_bodies_on_sharded_data(hip_libs, name):
    """Two-way coupling across slabs (rigid_impulses.wgsl:94-137, p2g.wgsl:142-155): every slab accumulates the
    fixed-point impulses of its own particles, the sums are reduced over the slabs before integrate_bodies. Against
    the single-domain run of the golden scene: the bodies to the fixed-point resolution (1e-5 per node and substep:
    a node's impulse is truncated per slab here, once in a single domain), the particles to fp32 round-off."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import native_lockstep
    make, k = _CASES[name]
    sc = make()
    dim = sc["particles"].dim
    ref = run_gpu(sc, k)
    ref_p, ref_b = ref.read_particles(), ref.read_body_poses()
    pipe = pipeline(dim)
    shards, _ = _native_slabs(sc, 2, pipe)
    assert min(s.num_particles() for s in shards) > 0
    native_lockstep(pipe, shards, k)
    for s in shards:
        s.sync()
    bodies = [s.read_body_poses() for s in shards]
    for key in ("rotation", "translation", "linvel", "angvel"):
        a = np.stack([b[key] for b in bodies[0]])
        assert np.array_equal(a, np.stack([b[key] for b in bodies[1]])), "every slab integrates the same bodies"
        want = np.stack([b[key] for b in ref_b])
        err = float(np.abs(a - want).max())
        report_margin(f"sharded body {key} abs err", err, 3e-4)
        assert err < 3e-4, (key, a, want)
    assert np.abs(np.stack([b["linvel"] for b in bodies[0]])[0] - np.asarray(sc["colliders"][0].linvel)[:dim]).max() > 1e-3, \
        "the dynamic body must have been pushed"
    outs = [s.export() for s in shards]
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.arange(sc["particles"].n, dtype=np.uint32))
    order = np.argsort(ids)
    for f, tol in (("pos", 1e-5), ("vel", 5e-5)):
        got = np.concatenate([o[f] for o in outs])[order]
        err = rel_rms(got, getattr(ref_p, f))
        report_margin(f"sharded two-way {f}", err, tol)
        assert err < tol, (f, err)


@pytest.mark.parametrize("dim", [3, 2])
def test_sharded_substep_with_pack_and_interior_grid_update_inside_the_p2g_launch_is_bit_identical(hip_libs, dim, monkeypatch):
    """Inside wgs_sharded_step the waves that pack the outgoing messages and the grid update of the interior blocks ride in
    the P2G launch (GU = 3: slabs handed over word by word); the interface layers are updated after the
    exchange. WGS_DEBUG = 262144 brings the k_pack_face launch and the one grid update back: the same bits on every slab
    (3 slabs in lockstep, a floor, particles migrating, a table rebuild inside the run)."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import native_lockstep
    monkeypatch.setenv("WGS_REHASH_PERIOD", "64")
    world = 3

    def run():
        if dim == 3:
            sc = scenes.config_scene("c2", world, None, "weak", n_side=24)
            sc["particles"].pos[:, 1] -= 5.6                  # in contact with the floor: both P2G bodies run
        else:
            sc = scenes.elastic_block_2d(nx=48 * world, ny=40, with_floor=True)
            sc["particles"].pos[:, 1] -= 4.6
        ps = sc["particles"]
        rng = np.random.default_rng(8)
        ps.vel[:] = rng.normal(0.0, 2.0, ps.vel.shape).astype(np.float32)
        ps.vel[:, 0] += 8.0
        pipe = pipeline(dim)
        shards, part = _native_slabs(sc, world, pipe)
        native_lockstep(pipe, shards, 40)
        for sh in shards:
            sh.sync()                                          # (the near-collider lists are seen here: paired launches from now on)
        native_lockstep(pipe, shards, 40)
        for sh in shards:
            sh.sync()
        return [sh.export() for sh in shards]
    a = run()
    monkeypatch.setenv("WGS_DEBUG", "262144")
    b = run()
    monkeypatch.delenv("WGS_DEBUG")
    # wgs_sharded_step with neighbours splits P2G: the two block layers at each cut first (their slabs are what the messages
    # are gathered from: they run beside the exchange on a stream of their own), every other block and the interior's grid
    # update in a second launch. WGS_DEBUG = 4194304 splits the lockstep slabs the same way (on their one stream): same bits.
    monkeypatch.setenv("WGS_DEBUG", "4194304")
    c = run()
    monkeypatch.delenv("WGS_DEBUG")
    # A slab's fused G2P bins its residents for the next substep (the guests it drops leave their block's total) and
    # k_g2p_arrivals the particles that arrive (Dev::bin_next); WGS_DEBUG = 1048576 brings launch 1 of the sort, k_rebin, back:
    # the same bits, the same storage order.
    monkeypatch.setenv("WGS_DEBUG", "1048576")
    e = run()
    monkeypatch.delenv("WGS_DEBUG")
    for other in (b, c, e):
        for x, y in zip(a, other):
            ox, oy = np.argsort(x["ids"]), np.argsort(y["ids"])    # (the storage order of a slab follows the arrival order of its guests)
            assert np.array_equal(x["ids"][ox], y["ids"][oy])
            for f in ("pos", "vel", "def_grad", "affine"):
                assert np.array_equal(x[f][ox], y[f][oy]), f


@pytest.mark.parametrize("world,dim", [(2, 3), (3, 3), (4, 3), (2, 2)])
def test_native_lockstep_matches_single_domain(hip_libs, world, dim, monkeypatch):
    """wgs_sharded_step_lockstep — the C++ driver of the substep protocol that wgs_sharded_step runs per rank over
    RCCL — reproduces the single-domain run (80 substeps: crosses a table rebuild; particles migrate)."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import associated_block_x, native_lockstep
    monkeypatch.setenv("WGS_REHASH_PERIOD", "64")             # table rebuilds inside the run (developer override, same results)
    # a bar along x, a few blocks per slab (a slab between two neighbours must be at least 3 blocks wide)
    if dim == 3:
        sc = scenes.config_scene("c2", world, None, "weak", n_side=24)      # 24 * world x 24 x 24 particles
        sc["colliders"] = []
    else:
        sc = scenes.elastic_block_2d(nx=48 * world, ny=40, with_floor=False)
    ps = sc["particles"]
    rng = np.random.default_rng(8)
    ps.vel[:] = rng.normal(0.0, 3.0, ps.vel.shape).astype(np.float32)
    ps.vel[:, 0] += 8.0
    k = 80
    ref = run_gpu(sc, k).read_particles()
    pipe = pipeline(dim)
    shards, part = _native_slabs(sc, world, pipe)
    assert part.min_interior_width() >= 3
    n0 = [s.num_particles() for s in shards]
    native_lockstep(pipe, shards, 30)
    native_lockstep(pipe, shards, k - 30)                  # two calls: state carried across frames
    for s in shards:
        s.sync()
    outs = [s.export() for s in shards]
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.arange(ps.n, dtype=np.uint32))
    assert [len(o["ids"]) for o in outs] != n0, "the test scene must make particles migrate"
    order = np.argsort(ids)
    for f, tol in (("pos", 1e-5), ("vel", 1e-5), ("def_grad", 1e-5), ("affine", 2e-4)):
        err = rel_rms(np.concatenate([o[f] for o in outs])[order], getattr(ref, f))
        report_margin(f"native lockstep {f}", err, tol)
        assert err < tol, (f, err)
    for r, o in enumerate(outs):                               # a rank holds its core range + the particles that left it in
        lo, hi = part.block_range(r)                           # the last substep (handed over with the next message)
        bx = associated_block_x(o["pos"], sc["cell_width"], dim)
        assert ((bx >= lo - 1) & (bx <= hi)).all()


@pytest.mark.parametrize("with_floor", [False, True])
def test_particles_enter_an_empty_slab_and_leave_theirs_empty(hip_libs, with_floor):
    """The edge of the one-exchange protocol: a cube flies from slab 0 into slab 1, which holds NO particle at the start —
    the first arrivals find none of their blocks active on their new rank and read their nodes from the message (the old
    owner's partial sums are then the totals; with the floor also the node cdfs, evaluated on the spot) — and keeps going
    until slab 0 is empty. Both against the single-domain run."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import NativeShard, SlabPartition, associated_block_x, native_lockstep, split_scene, uniform_material_of
    sc = scenes.neo_hookean_cube(n_side=16, with_floor=with_floor)
    ps = sc["particles"]
    if with_floor:
        ps.pos[:, 1] -= 5.6                                   # sliding just above the floor: CPIC state travels with the particles
    ps.vel[:, 0] = 400.0                                      # a third of a cell per substep
    k = 45                                                    # 15 cells: the whole cube (8 cells wide) crosses the cut
    ref = run_gpu(sc, k).read_particles()
    bx = associated_block_x(ps.pos, sc["cell_width"], 3)
    part = SlabPartition([int(bx.min()), int(bx.max()) + 1, int(bx.max()) + 12])
    pipe = pipeline(3)
    shards = []
    for r, (sub, gids) in enumerate(split_scene(ps, part, sc["cell_width"])):
        lo, hi = part.block_range(r)
        shards.append(NativeShard(pipe, sc["params"], sub, gids, sc["colliders"], sc["cell_width"], sc["grid_capacity"], lo, hi, r > 0, r < 1,
                                  particle_capacity=ps.n, model=sc["model"], uniform_material=uniform_material_of(ps),
                                  halo_capacity_records=256, migrant_capacity=2048))
    assert [s.num_particles() for s in shards] == [ps.n, 0]
    native_lockstep(pipe, shards, 20)
    mid = [s.num_particles() for s in shards]
    assert 0 < mid[0] < ps.n and sum(mid) == ps.n, mid         # on its way
    native_lockstep(pipe, shards, k - 20)
    for s in shards:
        s.sync()
    assert [s.num_particles() for s in shards] == [0, ps.n]
    outs = [s.export() for s in shards]
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.arange(ps.n, dtype=np.uint32))
    order = np.argsort(ids)
    for f, tol in (("pos", 1e-5), ("vel", 1e-5), ("def_grad", 1e-5)):
        err = rel_rms(np.concatenate([o[f] for o in outs])[order], getattr(ref, f))
        report_margin(f"cube into an empty slab ({'floor' if with_floor else 'free'}) {f}", err, tol)
        assert err < tol, (f, err)


def test_slabs_evict_the_blocks_a_body_leaves_behind(hip_libs):
    """Round 6: slabs of a decomposition evict their long-inactive blocks like single-domain data (kernels_sort.h regroup_block; until now a slab
    rebuilt its table every 1 024 substeps and whenever a moving body had used up the ids). A 16^3 cube flies through three slabs at a third of
    a cell per substep, with a block capacity that 150 substeps of its trail would exhaust: every slab builds its table once, ids come back
    on the free lists, nothing overflows, and the particles are those of the single-domain run."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import NativeShard, SlabPartition, associated_block_x, native_lockstep, split_scene, uniform_material_of
    sc = scenes.neo_hookean_cube(n_side=16, with_floor=False)
    ps = sc["particles"]
    ps.vel[:, 0] = 400.0
    ps.vel[:, 1] = 0.0
    sc["params"] = SimulationParams((0.0, 0.0, 0.0), sc["params"].dt)
    sc["grid_capacity"] = 256                                  # the cube holds ~64 blocks at a time and touches ~40 new ones every dozen substeps
    k = 150                                                    # 50 cells
    ref_data = run_gpu(sc, k)
    ref = ref_data.read_particles()
    assert ref_data.stats()["overflow"] == 0
    bx = associated_block_x(ps.pos, sc["cell_width"], 3)
    b0 = int(bx.min())
    part = SlabPartition([b0, b0 + 5, b0 + 10, b0 + 18])
    pipe = pipeline(3)
    shards = []
    for r, (sub, gids) in enumerate(split_scene(ps, part, sc["cell_width"])):
        lo, hi = part.block_range(r)
        shards.append(NativeShard(pipe, sc["params"], sub, gids, sc["colliders"], sc["cell_width"], sc["grid_capacity"], lo, hi, r > 0, r < 2,
                                  particle_capacity=ps.n, model=sc["model"], uniform_material=uniform_material_of(ps),
                                  halo_capacity_records=512, migrant_capacity=4096))
    for _ in range(k // 10):
        native_lockstep(pipe, shards, 10)
        for s in shards:
            s.sync()
    st = [s.stats() for s in shards]
    assert all(x["overflow"] == 0 for x in st), st
    assert all(x["table_rebuilds"] <= 1 for x in st), [x["table_rebuilds"] for x in st]
    assert sum(x["block_ids_free"] + x["table_marks"] for x in st) > 0, "no slab evicted a block"
    assert [s.num_particles() for s in shards][0] == 0
    outs = [s.export() for s in shards]
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.arange(ps.n, dtype=np.uint32))
    order = np.argsort(ids)
    for f, tol in (("pos", 1e-5), ("vel", 1e-5), ("def_grad", 1e-5)):
        err = rel_rms(np.concatenate([o[f] for o in outs])[order], getattr(ref, f))
        report_margin(f"slabs evicting a body's trail {f}", err, tol)
        assert err < tol, (f, err)


@pytest.mark.parametrize("name", ["mesh_floor3d", "polyline2d"])
def test_mesh_colliders_on_sharded_data(hip_libs, name):
    """Mesh colliders (rigid-particle samples) on slabs: every slab holds every sample, the node cdfs of the
    nodes two slabs share are computed by both from the same inputs. Two slabs in lockstep against the single-domain run of
    the golden scene: the same particles on the same side of the mesh, fields to fp32 round-off."""
    from helpers import pipeline
    from wgsparkl_amd.sharded import native_lockstep
    make, k = _CASES[name]
    sc = make()
    dim = sc["particles"].dim
    ref = run_gpu(sc, k).read_particles()
    pipe = pipeline(dim)
    shards, _ = _native_slabs(sc, 2, pipe)
    assert min(s.num_particles() for s in shards) > 0
    native_lockstep(pipe, shards, k)
    for s in shards:
        s.sync()
    outs = [s.export() for s in shards]
    ids = np.concatenate([o["ids"] for o in outs])
    assert np.array_equal(np.sort(ids), np.arange(sc["particles"].n, dtype=np.uint32))
    order = np.argsort(ids)
    assert (ref.cdf_affinity != 0).sum() > 10, "the scene must feel its mesh"
    for f, tol in (("pos", 1e-5), ("vel", 5e-5), ("def_grad", 1e-5)):
        err = rel_rms(np.concatenate([o[f] for o in outs])[order], getattr(ref, f))
        report_margin(f"sharded mesh {f}", err, tol)
        assert err < tol, (f, err)


def test_native_sharded_step_over_rccl_one_rank(hip_libs):
    """wgs_comm_* + wgs_shard_attach + wgs_sharded_step with a real RCCL communicator. A second rank on the same GPU
    is refused by RCCL, so: (a) world = 1 without neighbours must reproduce wgs_step on single-domain data;
    (b) WGS_COMM_SELF_NEIGHBOURS: the rank is its own lower and upper neighbour, so every ncclSend / ncclRecv group
    of an interior rank is issued and matched (its physics is meaningless: the slab adds its own halo to itself) —
    the run must complete, keep its particles and report no error."""
    import ctypes as C
    from helpers import pipeline
    from wgsparkl_amd import _ffi
    from wgsparkl_amd.sharded import INT_MAX, INT_MIN, NativeComm, NativeShard
    sc = scenes.neo_hookean_cube(n_side=24, with_floor=True)
    ps = sc["particles"]
    ps.vel[:, 0] = 2.0
    pipe = pipeline(3)
    k = 20
    ref = run_gpu(sc, k).read_particles()
    comm = NativeComm(pipe, None, 0, 1)
    gids = np.arange(ps.n, dtype=np.uint32)
    sh = NativeShard(pipe, sc["params"], ps, gids, sc["colliders"], sc["cell_width"], sc["grid_capacity"], INT_MIN, INT_MAX,
                     False, False, ps.n, sc["model"], comm=comm, halo_capacity_records=64, migrant_capacity=64)
    sh.step(k)
    sh.sync()
    out = sh.export()
    order = np.argsort(out["ids"])
    assert rel_rms(out["pos"][order], ref.pos) < 1e-6 and rel_rms(out["vel"][order], ref.vel) < 1e-5
    sh.close(); comm.close()
    # (b) self-neighbour proxy
    comm = NativeComm(pipe, None, 0, 1, flags=1)
    from wgsparkl_amd.sharded import associated_block_x
    bx = associated_block_x(ps.pos, sc["cell_width"], 3)
    sh = NativeShard(pipe, sc["params"], ps, gids, sc["colliders"], sc["cell_width"], sc["grid_capacity"], int(bx.min()), int(bx.max()) + 2,
                     True, True, int(ps.n * 1.5), sc["model"], comm=comm, halo_capacity_records=512, migrant_capacity=1024)
    ps.vel[:, 0] = 0.0
    sh2 = sh
    sh2.step(k)
    sh2.sync()
    assert sh2.num_particles() == ps.n
    sh2.close(); comm.close()

