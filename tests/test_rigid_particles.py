"""-m "not gpu": mesh colliders — host-side sampling (src/solver/particle3d.rs:250-428, particle2d.rs:206-234
restated in wgsparkl_amd/sampling.py) and the oracle's rigid-particle passes (sort.wgsl:38-86,139-161,
p2g_cdf.wgsl:52-190)."""
import numpy as np
import pytest

from wgsparkl_amd import sampling, scenes
from wgsparkl_amd.solver import Collider


def _bary(p, a, b, c):
    v0, v1, v2 = b - a, c - a, p - a
    d00, d01, d11, d20, d21 = v0 @ v0, v0 @ v1, v1 @ v1, v2 @ v0, v2 @ v1
    den = d00 * d11 - d01 * d01
    v = (d11 * d20 - d01 * d21) / den
    w = (d00 * d21 - d01 * d20) / den
    return 1 - v - w, v, w


def test_sample_triangle_stays_inside_and_off_the_base():
    rng = np.random.default_rng(0)
    for _ in range(20):
        a, b, c = (rng.uniform(-3, 3, 3).astype(np.float32) for _ in range(3))
        out = []
        sampling.sample_triangle(a, b, c, 0.7, out)
        for p in out:
            u, v, w = _bary(p.astype(np.float64), a.astype(np.float64), b.astype(np.float64), c.astype(np.float64))
            assert min(u, v, w) > -1e-4, (u, v, w)
        # spacing / sqrt(2) between neighbours along the base direction: never farther apart than one cell diagonal
        if len(out) > 1:
            pts = np.stack(out)
            nearest = np.sort(np.linalg.norm(pts[:, None] - pts[None], axis=-1), axis=1)[:, 1]
            assert nearest.max() < 0.7 + 1e-4


def test_sample_edge_excludes_a_and_is_evenly_spaced():
    out = []
    a, b = np.array([0, 0, 0], np.float32), np.array([3, 0, 0], np.float32)
    sampling.sample_edge(a, b, 1.0, out)
    xs = np.array([p[0] for p in out])
    step = np.float32(1.0) / np.sqrt(np.float32(2.0))
    assert np.allclose(xs, step * np.arange(1, len(xs) + 1)) and xs.min() > 0 and xs.max() < 3
    assert len(xs) == int(np.ceil(3 / step)) - 1


def test_sample_mesh_samples_shared_edges_once():
    v = np.array([[0, 0, 0], [4, 0, 0], [0, 0, 3], [4, 0, 3]], np.float32)
    idx = np.array([[0, 1, 2], [1, 3, 2]], np.uint32)
    pts, tri = sampling.sample_mesh(v, idx, 1.0)
    assert len(pts) == len(tri) and set(tri.tolist()) == {0, 1}
    # the shared edge (1, 2) belongs to the first triangle only: sampling both triangles separately finds more points
    sep = sum(len(sampling.sample_mesh(v, idx[i:i + 1], 1.0)[0]) for i in range(2))
    edge = []
    sampling.sample_edge(v[1], v[2], 1.0, edge)
    assert sep - len(pts) == len(edge) > 0
    assert not any(np.array_equal(p, q) for p in pts for q in v)          # never the vertices


def test_sample_polyline_matches_the_reference_loop():
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
