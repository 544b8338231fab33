"""-m "not gpu": mesh colliders — the product's host-side sampler (wgsparkl_amd/sampling.py, vectorised) against a
line-by-line restatement of the reference's loops (oracle/ref_sampling.py: src/solver/particle3d.rs:250-428,
particle2d.rs:206-234) — same points, same order, bit for bit — and the oracle's rigid-particle passes
(sort.wgsl:38-86,139-161, p2g_cdf.wgsl:52-190)."""
import numpy as np
import pytest

from oracle import ref_sampling
from wgsparkl_amd import sampling, scenes
from wgsparkl_amd.solver import Collider


def _bary(p, a, b, c):
    v0, v1, v2 = b - a, c - a, p - a
    d00, d01, d11, d20, d21 = v0 @ v0, v0 @ v1, v1 @ v1, v2 @ v0, v2 @ v1
    den = d00 * d11 - d01 * d01
    v = (d11 * d20 - d01 * d21) / den
    w = (d00 * d21 - d01 * d20) / den
    return 1 - v - w, v, w


def _same(got, want):
    assert got[0].shape == want[0].shape, (got[0].shape, want[0].shape)
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), "sampled points differ from the reference's loops"
    assert np.array_equal(got[1], want[1])


def test_vectorised_mesh_sampler_reproduces_the_reference_loops():
    """Random soups (shared edges, every longest-edge case), a heightfield and degenerate triangles: identical output."""
    rng = np.random.default_rng(11)
    for trial in range(12):
        nv = int(rng.integers(4, 40))
        v = rng.uniform(-4.0, 4.0, (nv, 3)).astype(np.float32)
        idx = np.stack([rng.choice(nv, 3, replace=False) for _ in range(int(rng.integers(1, 60)))]).astype(np.uint32)
        spacing = float(rng.uniform(0.3, 1.5))
        _same(sampling.sample_mesh(v, idx, spacing), ref_sampling.sample_mesh(v, idx, spacing))
    ii, jj = np.meshgrid(np.arange(9), np.arange(7), indexing="ij")
    hv, hi = sampling.heightfield_to_trimesh((0.3 * np.sin(0.9 * ii) * np.cos(0.7 * jj)).astype(np.float32), (18.0, 1.5, 14.0))
    rv, ri = ref_sampling.heightfield_to_trimesh((0.3 * np.sin(0.9 * ii) * np.cos(0.7 * jj)).astype(np.float32), (18.0, 1.5, 14.0))
    assert np.array_equal(hv, rv) and np.array_equal(hi, ri)
    _same(sampling.sample_mesh(hv, hi, 1.0), ref_sampling.sample_mesh(rv, ri, 1.0))
    # degenerate: a zero-area sliver, a repeated vertex, a triangle smaller than the spacing
    v = np.array([[0, 0, 0], [3, 0, 0], [1.5, 0, 0], [0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [2, 0, 0], [1, 1.7, 0], [0, 3, 0], [3, 3, 0]], np.float32)
    idx = np.array([[0, 1, 2], [0, 3, 1], [3, 4, 5], [0, 6, 7], [0, 1, 8], [1, 9, 8]], np.uint32)
    _same(sampling.sample_mesh(v, idx, 0.5), ref_sampling.sample_mesh(v, idx, 0.5))
    assert sampling.sample_mesh(v, np.zeros((0, 3), np.uint32), 0.5)[0].shape == (0, 3)


def test_longest_edge_ties_go_to_bc_then_ca_then_ab():
    """particle3d.rs:348-362 tests `max == bc` first, then `max == ca`, and keeps ab only otherwise. Triangles whose longest
    edge is tied (exactly, in fp32) in every combination, an equilateral face, an octahedron and a square pyramid."""
    tied = {
        "ab=bc": [(0, 0, 0), (3, 4, 0), (-1, 1, 0)],
        "ab=bc (advisor)": [(0, 0, 0), (1, 3, 0), (-1, 3, 0)],
        "ab=ca": [(0, 0, 0), (3, 4, 0), (5, 0, 0)],
        "bc=ca": [(-3, 0, 0), (3, 0, 0), (0, 0, 8)],
        "ab=bc=ca": [(1, 0, 0), (0, 1, 0), (0, 0, 1)],
    }
    for name, tri in tied.items():
        v = np.array(tri, np.float32)
        d = [np.float32(np.sqrt(np.sum((v[(k + 1) % 3] - v[k]) ** 2, dtype=np.float32))) for k in range(3)]
        assert sum(x == max(d) for x in d) >= 2, (name, d)                      # the tie is exact in fp32
        for spacing in (0.5, 0.31, 1.0):
            for rolled in range(3):                                             # every rotation of the vertex order
                idx = np.array([np.roll(np.arange(3), rolled)], np.uint32)
                got = sampling.sample_mesh(v, idx, spacing)
                assert got[0].shape[0] > 0
                _same(got, ref_sampling.sample_mesh(v, idx, spacing))
    octa_v = np.array([[2, 0, 0], [-2, 0, 0], [0, 2, 0], [0, -2, 0], [0, 0, 2], [0, 0, -2]], np.float32)
    octa_i = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.uint32)
    pyr_v = np.array([[-2, 0, -2], [2, 0, -2], [2, 0, 2], [-2, 0, 2], [0, 3, 0]], np.float32)
    pyr_i = np.array([[0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4], [0, 2, 1], [0, 3, 2]], np.uint32)
    fan_v = np.array([[0, 0, 0]] + [[3 * np.cos(k * np.pi / 3), 0, 3 * np.sin(k * np.pi / 3)] for k in range(6)], np.float32)
    fan_i = np.array([[0, 1 + k, 1 + (k + 1) % 6] for k in range(6)], np.uint32)
    for v, idx in ((octa_v, octa_i), (pyr_v, pyr_i), (fan_v, fan_i)):
        for spacing in (0.4, 0.75):
            _same(sampling.sample_mesh(v, idx, spacing), ref_sampling.sample_mesh(v, idx, spacing))


def test_vectorised_polyline_sampler_reproduces_the_reference_loops():
    rng = np.random.default_rng(12)
    for trial in range(10):
        nv = int(rng.integers(2, 20))
        v = rng.uniform(-6.0, 6.0, (nv, 2)).astype(np.float32)
        seg = np.stack([np.arange(nv - 1), np.arange(1, nv)], 1).astype(np.uint32)
        step = float(rng.uniform(0.2, 2.0))
        _same(sampling.sample_polyline(v, seg, step), ref_sampling.sample_polyline(v, seg, step))
    v = np.array([[0, 0], [2, 0], [2, 0], [2, 1.0]], np.float32)              # a degenerate segment; a length that is a multiple of the step
    seg = np.array([[0, 1], [1, 2], [2, 3]], np.uint32)
    _same(sampling.sample_polyline(v, seg, 1.0), ref_sampling.sample_polyline(v, seg, 1.0))


def test_build_rigid_particles_matches_the_reference_for_the_golden_scenes():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_cases import mesh_floor3d, polyline2d
    for make, dim in ((mesh_floor3d, 3), (polyline2d, 2)):
        sc = make()
        got = sampling.build_rigid_particles(sc["colliders"], dim, sc["cell_width"])
        want = ref_sampling.build_rigid_particles(sc["colliders"], dim, sc["cell_width"])
        assert got.keys() == want.keys()
        for k in got:
            assert np.array_equal(got[k], want[k]), k


def test_sampled_points_stay_inside_their_triangle_and_off_its_vertices():
    rng = np.random.default_rng(0)
    for _ in range(20):
        v = rng.uniform(-3, 3, (3, 3)).astype(np.float32)
        pts, tri = sampling.sample_mesh(v, np.array([[0, 1, 2]], np.uint32), 0.7)
        a, b, c = (v[k].astype(np.float64) for k in range(3))
        for p in pts:
            u, vv, w = _bary(p.astype(np.float64), a, b, c)
            assert min(u, vv, w) > -1e-4, (u, vv, w)
        if len(pts) > 1:   # spacing / sqrt(2) along the base and the height: never farther apart than one cell
            nearest = np.sort(np.linalg.norm(pts[:, None] - pts[None], axis=-1), axis=1)[:, 1]
            assert nearest.max() < 0.7 + 1e-4
        assert not any(np.array_equal(p, q) for p in pts for q in v)


def test_sample_mesh_samples_shared_edges_once():
    v = np.array([[0, 0, 0], [4, 0, 0], [0, 0, 3], [4, 0, 3]], np.float32)
    idx = np.array([[0, 1, 2], [1, 3, 2]], np.uint32)
    pts, tri = sampling.sample_mesh(v, idx, 1.0)
    assert len(pts) == len(tri) and set(tri.tolist()) == {0, 1}
    # the shared edge (1, 2) belongs to the first triangle only: sampling both triangles separately finds more points
    sep = sum(len(sampling.sample_mesh(v, idx[i:i + 1], 1.0)[0]) for i in range(2))
    step = np.float32(1.0) / np.sqrt(np.float32(2.0))
    on_edge = int(np.ceil(np.linalg.norm(v[1] - v[2]) / step)) - 1
    assert sep - len(pts) == on_edge > 0


def test_sample_polyline_walks_a_then_steps_then_b():
    v = np.array([[0, 0], [2.5, 0]], np.float32)
    pts, seg = sampling.sample_polyline(v, np.array([[0, 1]]), 1.0)
    # a, then a + 0, a + 1, a + 2 (k = 0 repeats a), then b
    assert np.allclose(pts[:, 0], [0, 0, 1, 2, 2.5]) and np.all(seg == 0)


def _drop_scene(kind):
    sc = scenes.neo_hookean_cube(n_side=8, with_floor=False)
    sc["particles"].pos[:, 1] -= 0.3
    sc["particles"].vel[:, 1] = -3.0
    if kind == "mesh":
        v = np.array([[-14, 0, -14], [-14, 0, 14], [14, 0, -14], [14, 0, 14]], np.float32)
        sc["colliders"] = [Collider.trimesh(v, np.array([[0, 1, 2], [2, 1, 3]]), (22.13, 7.3, 21.81))]
    else:
        sc["colliders"] = [Collider.cuboid((14.0, 1.0, 14.0), (22.13, 6.3, 21.81))]
    return sc


def test_flat_trimesh_acts_like_the_cuboid_it_covers(oracle_libs):
    """A horizontal two-triangle mesh at the height of a cuboid's top face gives the particles near it the same
    signed distances (p2g_cdf.wgsl:160-186 vs collide.wgsl) and therefore the same motion; the rigid samples
    add blocks of their own (sort.wgsl:38-86)."""
    res = {}
    for kind in ("mesh", "cuboid"):
        sc = _drop_scene(kind)
        ps = sc["particles"]
        st = oracle_libs.Oracle(3, np.float64).new_state(ps, sc["params"], sc["colliders"], sc["cell_width"],
                                                         sc["grid_capacity"], sc.get("model", 0))
        st.step(120)
        res[kind] = st
    m, c = res["mesh"], res["cuboid"]
    assert (m.arr["cdf_affinity"] != 0).sum() > 50
    assert np.allclose(m.arr["pos"], c.arr["pos"], atol=1e-9) and np.allclose(m.arr["vel"], c.arr["vel"], atol=1e-8)
    assert m.n_blocks > c.n_blocks and m.rig["needs_block"].sum() > 0
    # nodes under the sheet carry the sign bit of collider 0, nodes above it do not
    cells, _, dist, aff, closest = m.grid_records()
    touched = aff != 0
    below = cells[:, 1] * sc["cell_width"] < 7.3
    assert np.all((aff[touched & below] >> 16) & 1) and not np.any((aff[touched & ~below] >> 16) & 1)
    assert np.allclose(dist[touched], np.abs(cells[touched, 1] * sc["cell_width"] - 7.3), atol=1e-6)
    assert np.all(closest[touched] == 0)


def test_samples_without_a_neighbouring_block_are_ignored(oracle_libs):
    """sort.wgsl:80-84,149-151: a sample far from every particle neither creates a block nor enters a list."""
    sc = _drop_scene("mesh")
    ps = sc["particles"]
    st = oracle_libs.Oracle(3, np.float64).new_state(ps, sc["params"], sc["colliders"], sc["cell_width"],
                                                     sc["grid_capacity"], sc.get("model", 0))
    st.step(1)
    far = np.linalg.norm(st.rig["world_pts"][:, [0, 2]] - np.array([22.0, 22.0]), axis=1) > 13.0
    assert far.any() and not st.rig["needs_block"][far].any()
    assert np.all(st.rig["next"][far] == 0xFFFFFFFF)
    assert st.rig["node_len"].sum() < st.R.n
