"""Rigid-particle sampling of mesh colliders on the host, vectorised over all primitives at once.

What it must produce is fixed by the reference: `GpuRigidParticles::from_rapier` samples every trimesh / heightfield (3D) or
polyline (2D) collider once, on the CPU, in the collider's local frame (src/solver/particle3d.rs:100-150,250-428,
src/solver/particle2d.rs:75-125,206-234), and a drop-in host has to hand the device the same points in the same order (a
Rust caller keeps using the reference's own function and passes its buffers to `wgs_set_rigid_particles`). The reference
walks the primitives one at a time with nested loops; here every stage is one array expression over ALL primitives —
per-triangle frames, then all (triangle, base step) rows, then all (row, height step) points — and the ragged results are
laid out with prefix sums. Each point goes through the same float32 operations in the same order as in the reference, so
the output is identical bit for bit (tests/test_rigid_particles.py checks that against a line-by-line restatement kept
with the oracle, incl. degenerate triangles and shared edges).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

F32 = np.float32
EPS = F32(1.0e-5)          # particle3d.rs:243
_INV_SQRT2_DEN = F32(np.sqrt(F32(2.0)))


def _len(v: np.ndarray) -> np.ndarray:
    """Euclidean length of the rows of v, float32, components added left to right."""
    sq = (v * v).astype(F32)
    acc = sq[..., 0]
    for k in range(1, v.shape[-1]):
        acc = (acc + sq[..., k]).astype(F32)
    return np.sqrt(acc, dtype=F32)


def _dot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    pr = (a * b).astype(F32)
    acc = pr[..., 0]
    for k in range(1, a.shape[-1]):
        acc = (acc + pr[..., k]).astype(F32)
    return acc


def _ragged(counts: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """For row r with counts[r] items: (row index, 1-based item index) of every item, rows in order."""
    counts = np.maximum(counts.astype(np.int64), 0)
    total = int(counts.sum())
    row = np.repeat(np.arange(len(counts)), counts)
    first = np.cumsum(counts) - counts
    item = np.arange(total) - np.repeat(first, counts) + 1
    return row, item


def _steps(length: np.ndarray, spacing: F32) -> np.ndarray:
    """ceil(length / spacing) as an integer; 0 where the quotient is not finite."""
    with np.errstate(invalid="ignore", divide="ignore"):
        q = np.ceil((length / spacing).astype(F32))
    return np.where(np.isfinite(q), q, 0).astype(np.int64)


def _edge_points(a: np.ndarray, b: np.ndarray, spacing_xy: float):
    """Interior points of the edges a[e] -> b[e], one every spacing / sqrt(2) (particle3d.rs:301-322): (points, edge of each)."""
    spacing = F32(F32(spacing_xy) / _INV_SQRT2_DEN)
    ab = (b - a).astype(F32)
    length = _len(ab)
    ok = length > EPS
    with np.errstate(invalid="ignore", divide="ignore"):
        d = (ab / length[:, None]).astype(F32)
    n = np.where(ok, _steps(length, spacing) - 1, 0)
    e, i = _ragged(n)
    pts = (a[e] + d[e] * (spacing * i.astype(F32)).astype(F32)[:, None]).astype(F32)
    return pts, e


def _triangle_points(a: np.ndarray, b: np.ndarray, c: np.ndarray, spacing_xy: float):
    """Interior grid of every triangle along its longest edge and its height (particle3d.rs:338-428): (points, triangle of
    each), triangles in order, base steps outer, height steps inner."""
    spacing = F32(F32(spacing_xy) / _INV_SQRT2_DEN)
    dab, dbc, dca = _len(b - a), _len(c - b), _len(a - c)
    mx = np.maximum(np.maximum(dab, dbc), dca)
    # the longest edge becomes the base a -> b; a tie goes to bc, then ca, and only otherwise ab (particle3d.rs:348-362)
    rot = np.where(mx == dbc, 1, np.where(mx == dca, 2, 0))
    tri = np.stack([a, b, c], 1)                                    # [T, 3, 3]
    pick = lambda k: np.take_along_axis(tri, ((rot + k) % 3)[:, None, None].repeat(3, 2), 1)[:, 0, :]
    a, b, c = pick(0), pick(1), pick(2)
    ac, base = (c - a).astype(F32), (b - a).astype(F32)
    base_length = _len(base)
    with np.errstate(invalid="ignore", divide="ignore"):
        base_dir = (base / base_length[:, None]).astype(F32)
        ac_off = _dot(ac, base_dir)
        bc_off = (base_length - ac_off).astype(F32)
        live = (base_length > 0) & ~((ac_off < EPS) | (bc_off < EPS) | (base_length < EPS))
        height = (ac - base_dir * ac_off[:, None]).astype(F32)
        height_length = _len(height)
        hstep = ((height / height_length[:, None]).astype(F32) * spacing).astype(F32)
        tan_alpha = (height_length / ac_off).astype(F32)
        tan_beta = (height_length / bc_off).astype(F32)
    base_step = (base_dir * spacing).astype(F32)
    t, i = _ragged(np.where(live, _steps(base_length, spacing) - 1, 0))        # rows = (triangle, base step)
    base_pos = (a[t] + i.astype(F32)[:, None] * base_step[t]).astype(F32)
    with np.errstate(invalid="ignore"):
        hl = np.minimum((tan_alpha[t] * _len(base_pos - a[t])).astype(F32), (tan_beta[t] * _len(base_pos - b[t])).astype(F32))
    r, j = _ragged(_steps(hl, spacing) - 1)                                     # points = (row, height step)
    pts = (base_pos[r] + j.astype(F32)[:, None] * hstep[t[r]]).astype(F32)
    keep = np.all(np.isfinite(pts), axis=1)
    return pts[keep], t[r][keep]


def sample_mesh(vertices: np.ndarray, indices: np.ndarray, spacing_xy: float) -> Tuple[np.ndarray, np.ndarray]:
    """(points [m, 3] float32, triangle id per point [m] uint32) in the reference's order (particle3d.rs:250-299): for each
    triangle its interior, then those of its edges (0-1, 1-2, 2-0) that no earlier triangle brought; never the vertices."""
    v = np.asarray(vertices, F32).reshape(-1, 3)
    idx = np.asarray(indices, np.int64).reshape(-1, 3)
    nt = len(idx)
    if nt == 0:
        return np.zeros((0, 3), F32), np.zeros(0, np.uint32)
    ip, it = _triangle_points(v[idx[:, 0]], v[idx[:, 1]], v[idx[:, 2]], spacing_xy)
    # edges in visiting order (triangle-major); an undirected edge is sampled at its FIRST visit only
    ea, eb = idx[:, [0, 1, 2]].reshape(-1), idx[:, [1, 2, 0]].reshape(-1)
    key = np.maximum(ea, eb) * (int(idx.max()) + 1) + np.minimum(ea, eb)
    first = np.zeros(len(key), bool)
    first[np.unique(key, return_index=True)[1]] = True
    visit = np.nonzero(first)[0]
    ep, ee = _edge_points(v[ea[visit]], v[eb[visit]], spacing_xy)
    et = visit[ee] // 3
    # a triangle's block = interior points, then its edges' points: stable sort of (triangle, interior before edges)
    pts = np.concatenate([ip, ep])
    tri = np.concatenate([it, et])
    order = np.argsort(tri * 2 + np.concatenate([np.zeros(len(it), np.int64), np.ones(len(et), np.int64)]), kind="stable")
    return pts[order].astype(F32), tri[order].astype(np.uint32)


def sample_polyline(vertices: np.ndarray, indices: np.ndarray, sampling_step: float) -> Tuple[np.ndarray, np.ndarray]:
    """(points [m, 2], segment id per point) like particle2d.rs:206-234: per segment a, then a + k step for k = 0, 1, ...
    while k step <= length (k = 0 repeats a, as in the reference), then b; a degenerate segment contributes a alone."""
    v = np.asarray(vertices, F32).reshape(-1, 2)
    idx = np.asarray(indices, np.int64).reshape(-1, 2)
    if len(idx) == 0:
        return np.zeros((0, 2), F32), np.zeros(0, np.uint32)
    step = F32(sampling_step)
    a, b = v[idx[:, 0]], v[idx[:, 1]]
    ab = (b - a).astype(F32)
    length = _len(ab)
    ok = length > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        d = (ab / length[:, None]).astype(F32)
        most = np.where(ok, np.floor(length / step) + 2, 0).astype(np.int64)   # candidates k = 0 .. most - 1, filtered exactly below
    s, k1 = _ragged(most)
    shift = ((k1 - 1).astype(F32) * step).astype(F32)
    inside = shift <= length[s]
    s, shift = s[inside], shift[inside]
    walk = (a[s] + d[s] * shift[:, None]).astype(F32)
    # per segment: [a] + walk + [b if not degenerate]
    seg = np.concatenate([np.arange(len(idx)), s, np.nonzero(ok)[0]])
    pts = np.concatenate([a, walk, b[ok]])
    rank = np.concatenate([np.zeros(len(idx), np.int64), np.ones(len(s), np.int64), np.full(int(ok.sum()), 2, np.int64)])
    order = np.argsort(seg * 3 + rank, kind="stable")
    return pts[order].astype(F32), seg[order].astype(np.uint32)


def heightfield_to_trimesh(heights: np.ndarray, scale: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """parry's HeightField::to_trimesh (third party): heights[i, j] on a regular grid spanning
    [-scale.x / 2, scale.x / 2] x [-scale.z / 2, scale.z / 2], two triangles per cell, row-major vertices."""
    hts = np.asarray(heights, F32)
    nr, nc = hts.shape
    xs = ((np.arange(nc, dtype=F32) / F32(nc - 1) - F32(0.5)) * F32(scale[0])).astype(F32)
    zs = ((np.arange(nr, dtype=F32) / F32(nr - 1) - F32(0.5)) * F32(scale[2])).astype(F32)
    vtx = np.stack([np.broadcast_to(xs[None, :], (nr, nc)), (hts * F32(scale[1])).astype(F32), np.broadcast_to(zs[:, None], (nr, nc))], -1)
    i, j = np.meshgrid(np.arange(nr - 1), np.arange(nc - 1), indexing="ij")
    p00 = (i * nc + j).reshape(-1)
    p01, p10, p11 = p00 + 1, p00 + nc, p00 + nc + 1
    idx = np.stack([np.stack([p00, p10, p01], 1), np.stack([p10, p11, p01], 1)], 1).reshape(-1, 3)
    return vtx.reshape(-1, 3).astype(F32), idx.astype(np.uint32)


def build_rigid_particles(colliders, dim: int, sampling_step: float):
    """All mesh colliders of a scene -> the buffers `wgs_set_rigid_particles` takes: dict(local_pts [n, D] f32, ids [n, 4]
    u32 (the primitive's vertex ids offset by the collider's first vertex, collider id last), local_vtx [nv, D] f32,
    vtx_collider [nv] u32), or None without mesh colliders. Sampling step = cell width (src/pipeline.rs:144)."""
    pts, ids, vtx, vcol = [], [], [], []
    base = 0
    for cid, c in enumerate(colliders):
        if getattr(c, "vertices", None) is None:
            continue
        v = np.asarray(c.vertices, F32).reshape(-1, dim)
        ind = np.asarray(c.indices, np.uint32).reshape(-1, dim)
        p, prim = sample_mesh(v, ind, sampling_step) if dim == 3 else sample_polyline(v, ind, sampling_step)
        rec = np.zeros((len(p), 4), np.uint32)
        rec[:, :dim] = ind[prim] + base
        rec[:, 3] = cid
        pts.append(p); ids.append(rec); vtx.append(v); vcol.append(np.full(len(v), cid, np.uint32))
        base += len(v)
    if not pts:
        return None
    return dict(local_pts=np.concatenate(pts).astype(F32), ids=np.concatenate(ids).astype(np.uint32),
                local_vtx=np.concatenate(vtx).astype(F32), vtx_collider=np.concatenate(vcol).astype(np.uint32))
