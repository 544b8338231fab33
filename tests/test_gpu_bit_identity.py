"""-m gpu: results that must be THE SAME BITS — run to run, across launch shapes (WGS_DEBUG switches of the shipped library), with the sort's
binning inside the fused G2P or as a launch of its own, across a checkpoint / restart, with other data running on the device at the same time."""
import numpy as np
import pytest

from wgsparkl_amd import scenes
from wgsparkl_amd.models import (MODEL_COROTATED, MODEL_NEO_HOOKEAN, DruckerPrager, ElasticCoefficients, ParticlePhase)
from wgsparkl_amd.solver import Collider, ParticleSet, SimulationParams

from helpers import assert_close_to_truth, compare_cpic, compare_grids, grid_of, max_abs, rel_rms, report_margin, run_gpu, run_oracle
from gpu_common import (CPIC_GRID_V_TOL, CPIC_PART_TOL, FUZZ_BODY_ATOL, FUZZ_NODE_MISMATCH, FUZZ_PART_MISMATCH, FUZZ_VEL_TOL, GRID_V_TOL, PART_TOL,
                        _exploding_cube, _native_slabs, _random_scene, check_blocks, check_fields, check_grid, cloud_scene)

pytestmark = pytest.mark.gpu


def test_determinism(hip_libs):
    sc = cloud_scene(n=30000, seed=11)
    a = run_gpu(sc, 5).read_particles()
    b = run_gpu(sc, 5).read_particles()
    for name in ("pos", "vel", "def_grad", "affine"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_uniform_material_mode_is_bit_identical(hip_libs, monkeypatch):
    """One material for all particles: the four per-particle constants (mass, V0, lambda, mu) become kernel arguments
    and F[8] rides in their place (layout.h, Dev::uniform) — 32 bytes per particle and substep less through HBM. Same
    arithmetic on the same values: bit-identical to the general layout (WGS_DEBUG = 65536 keeps that one), incl. the
    CPIC passes and Drucker-Prager, and the read-back shows the caller's constants."""
    for make in (lambda: scenes.neo_hookean_cube(n_side=20, with_floor=True), lambda: scenes.sand_column(nx=12, ny=20, nz=12, with_floor=True)):
        def run():
            sc = make()
            sc["particles"].pos[:, 1] -= 5.6
            sc["particles"].vel[:, 1] = -2.0
            data = run_gpu(sc, 30)
            return sc, data.read_particles(), data.read_grid()
        sc, a, ga = run()
        monkeypatch.setenv("WGS_DEBUG", "65536")
        _, b, gb = run()
        monkeypatch.delenv("WGS_DEBUG")
        for f in ("pos", "vel", "def_grad", "affine", "mass", "init_volume", "lambda_", "mu", "cdf_affinity", "dp_state"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert np.array_equal(a.mass, sc["particles"].mass) and np.array_equal(a.mu, sc["particles"].mu)
        assert np.array_equal(ga[0], gb[0]) and np.array_equal(ga[1], gb[1])


@pytest.mark.parametrize("dim", [2, 3])
def test_uniform_plasticity_parameters_give_the_bits_of_the_general_layout(hip_libs, monkeypatch, dim):
    """Round 6 (layout.h Dev::uni_dp): when every particle carries the same Drucker-Prager h0..h3 the plastic fused G2P takes them
    as kernel arguments and leaves the DP0 quad alone (mode 1); when lambda, mu of the plasticity and max_stretch are shared as well,
    the per-particle plastic state is ONE quad (mode 2). Decided at creation, bitwise. Four clouds — everything shared with a
    breakable phase (mode 2), every other particle with its own plastic lambda / mu (mode 1), with its own max_stretch (mode 1),
    with its own h0 (general layout) — against WGS_DEBUG = 65536 (never a uniform mode): same bits, and the read-back shows the
    caller's parameters."""
    def cloud(kind):
        ps = scenes.random_cloud(2500, dim=dim, seed=11 + kind, extent=9.0, young=1e6, plasticity=DruckerPrager.new(1e6, 0.25),
                                 phase=ParticlePhase(1.0, 1.04) if kind != 3 else None)
        if kind == 1: ps.dp[::2, 4:6] *= np.float32(1.5)
        if kind == 2: ps.phase[::2, 1] = np.float32(1.08)
        if kind == 3: ps.dp[::2, 0] *= np.float32(0.9)
        return ps
    cols = [Collider.cuboid((50.0,) * dim, (5.0, -49.0) + ((5.0,) if dim == 3 else ()), **({} if dim == 3 else {"rotation": (0.0,)}))]
    for kind in range(4):
        def run():
            ps = cloud(kind)
            sc = dict(particles=ps, params=SimulationParams((0.0, -9.81, 0.0)[:dim], 5e-4), colliders=cols, cell_width=1.0, grid_capacity=4096, model=MODEL_COROTATED)
            return ps, run_gpu(sc, 30).read_particles()
        ps, a = run()
        monkeypatch.setenv("WGS_DEBUG", "65536")
        _, b = run()
        monkeypatch.delenv("WGS_DEBUG")
        for f in ("pos", "vel", "def_grad", "affine", "dp", "dp_state", "phase"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (kind, f)
        assert np.array_equal(a.dp, ps.dp) and np.array_equal(a.phase[:, 1], ps.phase[:, 1]), kind
        assert (a.dp_state != np.array([1.0, 1.0, 0.0], np.float32)).any() and (kind == 3 or (a.phase[:, 0] == 0.0).any()), "the scene should have yielded and broken by now"


def test_g2p_launch_shapes_are_bit_identical(hip_libs, monkeypatch):
    """The fused G2P advances one chunk of 64 sorted particles per wave, or — from 1.5 M particles on, where the launch is
    bound by latency x occupancy — two, with both chunks' particle state requested up front (kernels_transfer.h). The
    large-scene shape forced on small scenes (WGS_DEBUG = 131072) must give the same bits: elastic with the floor
    (both bodies of the paired launch), plastic, 2D."""
    makes = (lambda: scenes.neo_hookean_cube(n_side=24, with_floor=True), lambda: scenes.sand_column(nx=12, ny=20, nz=12, with_floor=True),
             lambda: scenes.elastic_block_2d(nx=50, ny=40))
    for make in makes:
        def run():
            sc = make()
            sc["particles"].pos[:, 1] -= 5.6 if sc["particles"].dim == 3 else 4.6
            sc["particles"].vel[:, 0] = 1.5
            return run_gpu(sc, 25).read_particles()
        a = run()
        monkeypatch.setenv("WGS_DEBUG", "131072")
        b = run()
        monkeypatch.delenv("WGS_DEBUG")
        for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "dp_state"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("seed", [1, 4, 9, 12])
def test_binning_inside_the_fused_g2p_is_bit_identical_to_the_rebin_launch(hip_libs, seed, monkeypatch):
    """Single-domain data: the fused G2P bins its own output for the next substep (new cell ids, block activation and totals,
    mover lists: g2p_body.inc, Dev::bin_next), and launch 1 of that substep's sort (k_rebin) is not launched. WGS_DEBUG =
    1048576 brings k_rebin back. The sort is only a permutation with a canonical order inside a cell, so 150 substeps — random
    colliders, particles flying through blocks, two table rebuilds, the calls cut at odd places with a wgs_sync between them —
    must end bit-identical, particles, grid, block set and counts; and both must have counted the same cell-changers."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    monkeypatch.setenv("WGS_REHASH_PERIOD", "64")             # (developer override, same results; the default is 1024)

    def run():
        sc = _random_scene(seed)
        pipe = pipeline(sc["particles"].dim)
        data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        for k in (1, 7, 63, 2, 77):
            pipe.step(data, k)
            data.sync()
        return data.read_particles(), data.read_grid(), data.read_blocks(), data.stats()
    a, ga, ka, sa = run()
    # (... and launch 2 of the sort puts the members of a dirty block's cells in order by ranking the newcomers among the stayers;
    # WGS_DEBUG = 16777216 keeps the insertion sort that covers the cases the ranking does not: the same order)
    for switch in ("1048576", "16777216"):
        monkeypatch.setenv("WGS_DEBUG", switch)
        b, gb, kb, sb = run()
        for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "dp_state"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (switch, f)
        for x, y in zip(ga, gb):
            assert np.array_equal(x, y)
        assert np.array_equal(ka[0], kb[0]) and np.array_equal(ka[2], kb[2])   # (block set and counts; where a block sits in memory is up to the atomics)
        assert sa["cell_changers"] == sb["cell_changers"] and sa["cell_changers"] > 0
        assert sa["overflow"] == 0 and sb["overflow"] == 0


@pytest.mark.parametrize("which", ["dynamic_ball_and_polyline_2d", "cube_on_floor_3d", "sand_between_walls_3d"])
def test_data_stepped_concurrently_on_their_own_streams_stay_bit_identical(hip_libs, which):
    """Several wgs_data of one pipeline may run at the same time, each on its own stream (threading). The grid
    update waits INSIDE the P2G launch for slabs of other workgroups (kernels_transfer.h gu_waves) — a wait that must make
    progress, and hand over complete data, also while kernels of other data occupy the device. Four copies of a scene are
    stepped interleaved, no synchronisation between the calls (their kernels overlap), and must end with the same bits as a
    copy that ran alone; nobody may report a hand-over time-out."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData

    def make():
        if which == "dynamic_ball_and_polyline_2d":
            sc = _random_scene(1)          # a dynamic ball pushed to its velocity cap, a kinematic cuboid, a moving polyline
        elif which == "cube_on_floor_3d":
            sc = scenes.neo_hookean_cube(n_side=40, with_floor=True)
            sc["particles"].pos[:, 1] -= 5.6
            sc["particles"].vel[:, 0] = 1.5
        else:
            sc = scenes.sand_column(nx=12, ny=20, nz=12, with_floor=True)
        pipe = pipeline(sc["particles"].dim)
        return pipe, MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
    pipe, alone = make()
    for k in (3, 17, 20):
        pipe.step(alone, k)
    alone.sync()
    ref = alone.read_particles()
    copies = [make()[1] for _ in range(4)]
    for k in (3, 17, 20):
        for _ in range(k):
            for c in copies:
                pipe.step(c, 1)
    for c in copies:
        c.sync()                 # (raises on ERRBIT_HANDOVER)
        got = c.read_particles()
        for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "dp_state"):
            assert np.array_equal(getattr(got, f), getattr(ref, f)), f
        assert c.stats()["overflow"] == 0


def test_grid_update_inside_the_p2g_launch_is_bit_identical_to_its_own_launch(hip_libs, monkeypatch):
    """Single-domain simulations run the grid update as waves of the (last) P2G launch: P2G hands its slabs over
    inside the launch (write-through stores, one word per block), the waves gather past their XCD's L2
    (kernels_transfer.h gu_waves). WGS_DEBUG = 262144 brings the launch of its own back: same sums in the same order, so
    the same bits — no colliders (one P2G launch), a floor in contact (two launches, then the paired one after the
    host has seen the list), plastic between walls, 2D; particles AND the grid (nodes, slabs' velocities feed the G2P)."""
    makes = (lambda: scenes.neo_hookean_cube(n_side=24), lambda: scenes.neo_hookean_cube(n_side=40, with_floor=True),
             lambda: scenes.sand_column(nx=12, ny=20, nz=12, with_floor=True), lambda: scenes.elastic_block_2d(nx=50, ny=40),
             lambda: scenes.corotated_cube_with_paddle(n_side=32))   # (two-way coupling: node impulses gathered by the same waves)
    for make in makes:
        def run():
            sc = make()
            if sc["colliders"] and len(sc["colliders"]) == 1:
                sc["particles"].pos[:, 1] -= 5.6 if sc["particles"].dim == 3 else 4.6
            sc["particles"].vel[:, 0] = 1.5
            from helpers import pipeline
            from wgsparkl_amd import MpmData
            pipe = pipeline(sc["particles"].dim)
            data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
            pipe.step(data, 12)
            data.sync()          # (a long near-collider list seen here switches P2G to its paired launch)
            pipe.step(data, 13)
            return data.read_particles(), data.read_grid(), data.read_body_poses()
        a, ga, ba = run()
        monkeypatch.setenv("WGS_DEBUG", "262144")
        b, gb, bb = run()
        monkeypatch.delenv("WGS_DEBUG")
        for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "dp_state"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        for x, y in zip(ga, gb):
            assert np.array_equal(x, y)
        for x, y in zip(ba, bb):
            for key in ("translation", "rotation", "linvel", "angvel"):
                assert np.array_equal(x[key], y[key]), key
        if len(ba) > 1:      # moving bodies: integrate_bodies rides in the next substep's first sort launch (524288: a launch of its own)
            monkeypatch.setenv("WGS_DEBUG", "524288")
            c, _, bc = run()
            monkeypatch.delenv("WGS_DEBUG")
            for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity"):
                assert np.array_equal(getattr(a, f), getattr(c, f)), f
            for x, y in zip(ba, bc):
                for key in ("translation", "rotation", "linvel", "angvel"):
                    assert np.array_equal(x[key], y[key]), key


@pytest.mark.parametrize("seed", [0, 3, 8])
def test_steady_state_rebinning_is_bit_identical_to_full_binning(hip_libs, seed, monkeypatch):
    """k_rebin (re-binning relative to the previous substep's blocks) against the general k_bin forced on every
    substep (WGS_DEBUG=128, read when the data is created): the sort is only a permutation, so 150 substeps —
    across two table rebuilds — must end bit-identical."""
    sc = _random_scene(seed)
    k = 150
    monkeypatch.setenv("WGS_REHASH_PERIOD", "64")             # (developer override, same results; the default is 1024)
    a = run_gpu(sc, k).read_particles()
    monkeypatch.setenv("WGS_DEBUG", "128")
    b = run_gpu(sc, k).read_particles()
    for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "dp_state"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_plastic_pair_register_budgets_are_bit_identical(hip_libs, monkeypatch):
    """Drucker-Prager sand between a floor and four walls, half of the blocks near a collider: after the first wgs_sync
    the fused G2P runs the variant compiled for 2 waves per SIMD (no spills in the CPIC body). Same source, another
    register budget: the results must be the bits of the 3-waves variant (WGS_DEBUG = 16384 keeps that one), because
    which of the two runs depends on when the host synchronised."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    sc = scenes.sand_column(nx=40, ny=60, nz=40, with_walls=True)
    sc["particles"].pos[:, 1] -= 5.8

    def run():
        pipe = pipeline(3)
        data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        pipe.step(data, 4)
        data.sync()
        st = data.stats()
        assert st["num_near_collider_blocks"] * 2 >= st["num_active_blocks"]     # the switch condition of capi.hip
        pipe.step(data, 8)
        data.sync()
        return data.read_particles()
    a = run()
    monkeypatch.setenv("WGS_DEBUG", "16384")
    b = run()
    for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "dp_state"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.isfinite(a.pos).all() and len(sc["colliders"]) == 5
    # the one-way P2G pair has two register budgets too (chosen from the particle count and the list length): force the
    # small one by making the scene "large" is not possible at this size, so compare the large budget (this scene's
    # choice) with the separate launches, and the small budget at a size that selects it below
    monkeypatch.setenv("WGS_DEBUG", "8192")
    c = run()
    for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "dp_state"):
        assert np.array_equal(getattr(a, f), getattr(c, f)), f
    for c in range(5):
        assert ((a.cdf_affinity >> c) & 1).sum() > 100, c


def test_large_one_way_scenes_do_not_depend_on_when_the_host_synchronised(hip_libs, monkeypatch):
    """From 600 k particles on, one-way collider simulations always run the paired P2G launch with the CPIC body cut to
    168 VGPRs — a budget that differs from the unconstrained one in the last bit here and there, so it must not follow
    the near-collider list the host last saw. 640 k neo-Hookean particles lying on the floor: eight substeps in one call
    and the same eight with a wgs_sync after the third end bit-identical; the unconstrained budget (WGS_DEBUG = 32768)
    agrees to round-off."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    sc = scenes.neo_hookean_cube(n_side=86, with_floor=True)
    sc["particles"].pos[:, 1] -= 5.7
    assert sc["particles"].n >= 600_000

    def run(chunks):
        pipe = pipeline(3)
        data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        for k in chunks:
            pipe.step(data, k)
            data.sync()
        assert data.stats()["num_near_collider_blocks"] >= 8
        return data.read_particles()
    a, b = run((8,)), run((3, 5))
    for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    monkeypatch.setenv("WGS_DEBUG", "32768")
    c = run((3, 5))
    assert np.array_equal(a.cdf_affinity, c.cdf_affinity)
    for f in ("pos", "vel", "def_grad"):
        assert rel_rms(getattr(c, f), getattr(a, f)) < 1e-6, f


@pytest.mark.parametrize("seed", [1, 2, 6, 8])
def test_checkpoint_restart_random_scenes(hip_libs, seed):
    """Bit-exact restart on the fuzz scenes: dynamic and kinematic bodies, mesh colliders, plasticity."""
    import dataclasses
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    sc = _random_scene(seed)
    dim = sc["particles"].dim
    pipe = pipeline(dim)
    args = (sc["cell_width"], sc["grid_capacity"], sc["model"])
    full = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], *args)
    pipe.step(full, 20)
    part = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], *args)
    pipe.step(part, 9)
    snap, bodies = part.read_particles(), part.read_body_poses()
    def restored(c, b):
        rot = tuple(b["rotation"]) if dim == 3 else (float(np.arctan2(b["rotation"][1], b["rotation"][0])),)
        return dataclasses.replace(c, translation=tuple(b["translation"]), rotation=rot, linvel=tuple(b["linvel"]) + (0.0,) * (3 - dim),
                                   angvel=tuple(b["angvel"]), com=tuple(b["com"]))
    cols2 = [restored(c, b) for c, b in zip(sc["colliders"], bodies)]
    rest = MpmData.new(pipe, sc["params"], snap, cols2, *args)
    rest.set_plastic_state(snap.dp_state)
    pipe.step(rest, 11)
    a, b = full.read_particles(), rest.read_particles()
    exact = dim == 3     # 2D poses are handed over as an angle: cos / sin round-trip costs an ulp
    for f in ("pos", "vel", "def_grad", "affine", "dp_state", "cdf_affinity"):
        if exact or not sc["colliders"]:
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        elif f != "cdf_affinity":
            assert rel_rms(getattr(b, f), getattr(a, f)) < 1e-4, f


def test_checkpoint_restart_is_bit_exact(hip_libs):
    """read_particles (+ plastic state, + body poses) -> MpmData.new -> set_plastic_state continues the
    run bit-for-bit (every reduction is in canonical particle order, whatever the storage order)."""
    import dataclasses
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    ps = scenes.random_cloud(3000, dim=3, seed=5, extent=10.0, young=1e6, plasticity=DruckerPrager.new(1e6, 0.25), phase=None)
    ps.pos[:, 1] += 3.0
    cols = [Collider.cuboid((50.0, 1.0, 50.0), (8.0, 1.0, 8.0)),
            Collider.ball(1.5, (8.0, 12.0, 8.0), linvel=(0.0, -1.0, 0.0), angvel=(0.0, 0.0, 0.5))]
    params = SimulationParams((0.0, -9.81, 0.0), 5e-4)
    pipe = pipeline(3)
    args = (1.0, 4096, MODEL_COROTATED)
    full = MpmData.new(pipe, params, ps, cols, *args)
    pipe.step(full, 24)
    part = MpmData.new(pipe, params, ps, cols, *args)
    pipe.step(part, 12)
    snap, bodies = part.read_particles(), part.read_body_poses()
    assert (snap.dp_state != np.array([1.0, 1.0, 0.0], np.float32)).any(), "scene should have yielded by now"
    cols2 = [dataclasses.replace(c, translation=tuple(b["translation"]), rotation=tuple(b["rotation"]),
                                 linvel=tuple(b["linvel"]), angvel=tuple(b["angvel"]), com=tuple(b["com"]))
             for c, b in zip(cols, bodies)]
    rest = MpmData.new(pipe, params, snap, cols2, *args)
    rest.set_plastic_state(snap.dp_state)
    pipe.step(rest, 12)
    a, b = full.read_particles(), rest.read_particles()
    for f in ("pos", "vel", "def_grad", "affine", "dp_state", "phase", "cdf_affinity", "cdf_normal", "cdf_dist"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    pa, pb = full.read_body_poses(), rest.read_body_poses()
    for x, y in zip(pa, pb):
        for key in x:
            assert np.array_equal(x[key], y[key]), key



def test_checkpoint_restart_with_a_rotated_fixed_collider_next_to_a_moving_one(hip_libs):
    """A fixed collider is BIT-static whatever else moves (kernels_bodies.h bodies_integrate_one skips the identity
    integration of a body at rest): node cdfs cached out of reach of the moving colliders (Dev::cdf_moving) stay those a
    restarted run computes, and the pose a run reads back after any number of substeps is the one it was given."""
    import dataclasses
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    ps = scenes.random_cloud(4000, dim=3, seed=11, extent=10.0, young=1e6)
    ps.pos[:, 1] += 4.0
    q = np.array([0.13, -0.31, 0.22, 0.91])
    q = tuple((q / np.linalg.norm(q)).astype(np.float32).astype(float))
    cols = [Collider.cuboid((50.0, 1.0, 50.0), (8.0, 0.5, 8.0), rotation=q),
            Collider.ball(1.5, (8.0, 13.0, 8.0), linvel=(0.0, -2.0, 0.0), angvel=(0.3, 0.0, 0.5))]
    params = SimulationParams((0.0, -9.81, 0.0), 5e-4)
    pipe = pipeline(3)
    args = (1.0, 4096, MODEL_NEO_HOOKEAN)
    full = MpmData.new(pipe, params, ps, cols, *args)
    pose0 = full.read_body_poses()[0]
    pipe.step(full, 30)
    part = MpmData.new(pipe, params, ps, cols, *args)
    pipe.step(part, 13)
    snap, bodies = part.read_particles(), part.read_body_poses()
    for key in ("translation", "rotation"):
        assert np.array_equal(bodies[0][key], pose0[key]), f"the fixed collider's {key} changed bits"
    cols2 = [dataclasses.replace(c, translation=tuple(b["translation"]), rotation=tuple(b["rotation"]),
                                 linvel=tuple(b["linvel"]), angvel=tuple(b["angvel"]), com=tuple(b["com"]))
             for c, b in zip(cols, bodies)]
    rest = MpmData.new(pipe, params, snap, cols2, *args)
    rest.set_plastic_state(snap.dp_state)   # (phase None: the reference's default Drucker-Prager, quirk B1 — the scene is plastic)
    pipe.step(rest, 17)
    a, b = full.read_particles(), rest.read_particles()
    for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity", "cdf_normal", "cdf_dist"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    ga, gb = full.read_grid(), rest.read_grid()
    for x, y in zip(ga, gb):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("plastic", [False, True])
def test_a_body_crossing_the_grid_evicts_blocks_and_stays_bit_identical_to_the_rebuild_path(hip_libs, plastic, monkeypatch):
    """A cube flies through the grid and spins for 2 000 substeps: it leaves a trail of blocks nobody activates any more. Launch 2 of
    the sort evicts them (their table slots are marked, their ids reused: kernels_sort.h regroup_block) — no table rebuild but the
    periodic ones; with WGS_DEBUG=1024 nothing is evicted and the table is rebuilt whenever three quarters of the ids are handed out.
    The sort is only a permutation: the same bits either way, and the evicting run rebuilds less often."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    sc = scenes.neo_hookean_cube(n_side=16, with_floor=True, grid_capacity=512)   # (~125-200 blocks at a time: the grid never grows; 384 ids run out every few hundred substeps)
    ps = sc["particles"]
    if plastic:   # (the variants with the plasticity / fracture branch bin in launch 1 of the sort, k_rebin, not in the fused G2P; a
        # breakable phase that never breaks selects them and leaves the cube in one piece)
        ps = ParticleSet.uniform(sc["particles"].pos, 0.25, 2700.0, ElasticCoefficients.from_young_modulus(1.0e7, 0.2),
                                 phase=ParticlePhase(1.0, 1.0e6))
    rel = ps.pos - ps.pos.mean(0)
    ps.vel[:, 0] = 40.0 + 1.5 * rel[:, 2]
    ps.vel[:, 1] = 25.0
    ps.vel[:, 2] = 40.0 - 1.5 * rel[:, 0]
    sc["params"] = SimulationParams((0.0, 0.0, 0.0), 1.0 / 300.0)   # (270 cells along x and z, 170 along y in 2 000 substeps)
    monkeypatch.setenv("WGS_REHASH_PERIOD", "100000")   # (developer override, same results: no periodic rebuild inside the run)
    pipe = pipeline(3)
    def run():
        data = MpmData.new(pipe, sc["params"], ps, sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        for _ in range(20):
            pipe.step(data, 100)
            data.sync()
        return data.read_particles(), data.stats()
    a, sa = run()
    monkeypatch.setenv("WGS_DEBUG", "1024")
    b, sb = run()
    for f in ("pos", "vel", "def_grad", "affine", "dp_state"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert sa["overflow"] == 0 and sb["overflow"] == 0
    assert sa["table_rebuilds"] < sb["table_rebuilds"], (sa["table_rebuilds"], sb["table_rebuilds"])
    assert sa["table_rebuilds"] <= 1 + sa["grid_growths"]     # the first substep (+ one per growth of the grid: none expected)


@pytest.mark.parametrize("scene", ["sand3", "paddle"])
def test_node_cdf_summaries_shared_between_blocks_give_the_bits_of_whole_tile_evaluation(hip_libs, scene, monkeypatch):
    """Where a collider moves, launch 2 of the sort evaluates every block's OWN nodes against the colliders and lets the block's
    neighbours know which of them have an affinity (Dev::block_cdf_summ, kernels_sort.h) instead of evaluating the (BW+2)^3 tile of
    every block — each node up to eight times. The class of a block (listed for the CPIC bodies or not) and everything downstream must
    be what the whole-tile evaluation gives (WGS_DEBUG=2048), also when no word is ever waited for (WGS_DEBUG=2: every neighbour whose
    word is late is evaluated locally — the path taken when a neighbour's wave is not resident)."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    if scene == "sand3":
        sc = scenes.reference_sand3()
        steps = 150
    else:   # a box swept through a block of elastic material resting on the floor: the set of blocks in reach changes every substep
        sc = scenes.neo_hookean_cube(n_side=40, with_floor=True)
        sc["particles"].pos[:, 1] -= sc["particles"].pos[:, 1].min() - 2.5 * sc["cell_width"]
        lo, hi = sc["particles"].pos.min(0), sc["particles"].pos.max(0)
        paddle = Collider.cuboid((0.6, 2.0, 6.0), translation=(float(lo[0]) - 1.0, float(0.5 * (lo[1] + hi[1])), float(0.5 * (lo[2] + hi[2]))),
                                 linvel=(30.0, 0.0, 0.0))
        sc["colliders"] = list(sc["colliders"]) + [paddle]
        steps = 200
    pipe = pipeline(3)
    def run():
        data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        pipe.step(data, steps)
        data.sync()
        return data.read_particles(), data.stats()
    a, sa = run()
    assert sa["overflow"] == 0
    assert np.any(a.cdf_affinity != 0)   # (somebody is within reach of a collider)
    for dbg in ("2048", "2"):
        monkeypatch.setenv("WGS_DEBUG", dbg)
        b, sb = run()
        for f in ("pos", "vel", "def_grad", "affine", "dp_state", "cdf_affinity", "cdf_normal", "cdf_dist"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (dbg, f)


@pytest.mark.parametrize("scene", ["sand3", "paddle", "sand2"])
def test_particle_cdf_by_prologue_waves_of_the_p2g_launch_gives_the_bits_of_the_in_workgroup_prologue(hip_libs, scene, monkeypatch):
    """With a short near-collider list (as of the host's last look) the paired P2G launch computes the particle cdf of the listed
    blocks in prologue WAVES, one per visit-list entry, ahead of its workgroups, and hands the quads over inside the launch
    (written through, counted per block, fetched past the L2: kernels_transfer.h pcdf_waves) instead of in three to six rounds inside
    each block's CPIC workgroup. Same particles, same arithmetic: the same bits as with WGS_DEBUG=4 (never prologue waves), substep
    after substep across host looks."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    dim = 3
    if scene == "sand3":
        sc = scenes.reference_sand3()
    elif scene == "sand2":
        sc = scenes.reference_sand2()
        dim = 2
    else:
        sc = scenes.corotated_cube_with_paddle(n_side=32)
    pipe = pipeline(dim)
    def run():
        data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        for _ in range(12):
            pipe.step(data, 10)
            data.sync()      # (the host looks at the lists here: the next call sizes its launches from what it saw)
        st = data.stats()
        return data.read_particles(), st, (data.read_body_poses() if sc["colliders"] else None)
    a, sa, ba = run()
    assert sa["overflow"] == 0 and sa["num_near_collider_blocks"] > 0
    # 4: never prologue waves; 8: prologue workgroups sized for an empty list whatever the host saw — the launch then decides from the
    # lists of the substep itself (too long for so few waves: the blocks' workgroups do the work; short enough: the waves do)
    for dbg in ("4", "8"):
        monkeypatch.setenv("WGS_DEBUG", dbg)
        b, sb, bb = run()
        for f in ("pos", "vel", "def_grad", "affine", "dp_state", "cdf_affinity", "cdf_normal", "cdf_dist"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (dbg, f)
        if ba is not None:
            for x, y in zip(ba, bb):
                for key in x:
                    assert np.array_equal(np.asarray(x[key]), np.asarray(y[key])), (dbg, key)


@pytest.mark.parametrize("scene", ["at_rest", "landed", "flying"])
def test_direct_runs_of_unchanged_blocks_give_the_bits_of_the_gather_through_the_permutation(hip_libs, scene, monkeypatch):
    """A block whose run is its previous run member for member (nobody moved, nobody arrived) is handed to the plain body of P2G with
    its cells' runs in the buffer's own coordinates: the particles are read where they are, without the gather through `perm`
    (layout.h CELL_DIRECT; one dependent round trip less per block). The same particles in the same order: the same bits as with
    WGS_DEBUG = 33554432 (every block through the permutation) — a cube in free fall (every block direct), the cube landing on the floor
    (direct and gathered blocks side by side, listed blocks beside them) and the cube crossing the grid and spinning (hardly any direct
    block; blocks becoming direct and dirty again), with a host look in between."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    sc = scenes.neo_hookean_cube(n_side=86, with_floor=True)
    ps = sc["particles"]
    nsub = 12
    if scene == "landed":
        ps.pos[:, 1] -= 5.7
        ps.vel[:, 1] = -3.0
        nsub = 24
    elif scene == "flying":
        r = ps.pos - ps.pos.mean(0)
        ps.vel[:] = (np.array([4.0, 2.5, 3.0]) + np.cross(np.array([0.0, 0.3, 0.1]), r)).astype(np.float32)
        nsub = 60

    def run():
        pipe = pipeline(3)
        data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        pipe.step(data, nsub // 2)
        data.sync()
        pipe.step(data, nsub - nsub // 2)
        data.sync()
        st = data.stats()
        assert st["overflow"] == 0
        return data.read_particles(), data.read_grid(), st
    a, ga, sta = run()
    monkeypatch.setenv("WGS_DEBUG", "33554432")
    b, gb, stb = run()
    for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for x, y in zip(ga, gb):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    if scene == "flying":
        assert sta["cell_changers"] > 0   # (the scene does move: blocks are dirty)


def test_large_two_way_scenes_with_the_near_collider_launch_first_give_the_bits_of_the_other_order(hip_libs, monkeypatch):
    """Large simulations with a body that moves (>= 600 k particles: BASELINE.json configs[3]) run P2G as two launches; since round 6 the
    near-collider launch goes FIRST and the grid update rides behind the plain launch (at the plain body's occupancy instead of the two-way
    body's). Same slabs, same gather: the bits of the plain-launch-first order (WGS_DEBUG = 67108864), also across a host look."""
    from helpers import pipeline
    from wgsparkl_amd import MpmData
    sc = scenes.config_scene("c4", n_side=86)
    assert sc["particles"].n >= 600_000

    def run():
        pipe = pipeline(3)
        data = MpmData.new(pipe, sc["params"], sc["particles"], sc["colliders"], sc["cell_width"], sc["grid_capacity"], sc["model"])
        pipe.step(data, 30)
        data.sync()
        pipe.step(data, 30)
        data.sync()
        st = data.stats()
        assert st["overflow"] == 0 and st["num_near_collider_blocks"] > 0
        return data.read_particles(), data.read_grid(), data.read_body_poses()
    a, ga, pa = run()
    monkeypatch.setenv("WGS_DEBUG", "67108864")
    b, gb, pb = run()
    for f in ("pos", "vel", "def_grad", "affine", "cdf_affinity"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for x, y in zip(ga, gb):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    for x, y in zip(pa, pb):
        for key in ("rotation", "translation", "linvel", "angvel"):
            assert np.array_equal(np.asarray(x[key]), np.asarray(y[key])), key
