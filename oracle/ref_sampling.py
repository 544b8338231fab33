"""TEST INFRASTRUCTURE — a line-by-line RESTATEMENT of the reference's host-side rigid-particle sampling, kept next to the
oracle as the checker of the product's own sampler (wgsparkl_amd/sampling.py, which is written differently — vectorised
over all primitives — and must reproduce these points exactly, in this order): only tests/ and oracle/ import it.
(`GpuRigidParticles::from_rapier`, src/solver/particle3d.rs:100-150; 2D src/solver/particle2d.rs:75-125).

The reference samples every trimesh / heightfield (3D) or polyline (2D) collider once, on the CPU, in the
collider's local frame; the device only transforms the samples by the body pose every substep
(src/solver/rigid_particle_update.wgsl). All arithmetic is float32 like the Rust code.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

F32 = np.float32
EPS = F32(1.0e-5)          # particle3d.rs:243


def _dot(a, b):
    """nalgebra's dot of a small static vector: a0 b0 + a1 b1 (+ a2 b2), left to right, float32 (no BLAS, no fma)."""
    acc = F32(a[0]) * F32(b[0])
    for k in range(1, len(a)):
        acc = F32(acc + F32(a[k]) * F32(b[k]))
    return F32(acc)


def _norm(v):
    v = np.asarray(v, F32)
    return F32(np.sqrt(_dot(v, v), dtype=F32))


def sample_edge(a, b, spacing_xy, out: List[np.ndarray]):
    """particle3d.rs:301-322: points strictly after `a`, every spacing / sqrt(2) along the edge."""
    ab = (b - a).astype(F32)
    length = _norm(ab)
    if length > EPS:
        d = (ab / length).astype(F32)
        spacing = F32(spacing_xy) / F32(np.sqrt(F32(2.0)))
        nsteps = int(np.ceil(length / spacing))
        for i in range(1, nsteps):
            out.append((a + d * (spacing * F32(i))).astype(F32))


def sample_triangle(a, b, c, spacing_xy, out: List[np.ndarray]):
    """particle3d.rs:338-428: grid along the longest edge (base) and the height, interior only."""
    dab, dbc, dca = _norm(b - a), _norm(c - b), _norm(a - c)
    mx = max(dab, dbc, dca)
    if mx == dbc:
        a, b, c = b, c, a
    elif mx == dca:
        a, b, c = c, a, b
    ac = (c - a).astype(F32)
    base = (b - a).astype(F32)
    base_length = _norm(base)
    if not base_length > 0:
        return
    base_dir = (base / base_length).astype(F32)
    spacing = F32(spacing_xy) / F32(np.sqrt(F32(2.0)))
    base_step_count = np.ceil(base_length / spacing)
    base_step = (base_dir * spacing).astype(F32)
    ac_off = _dot(ac, base_dir)
    bc_off = F32(base_length - ac_off)
    if ac_off < EPS or bc_off < EPS or base_length < EPS:
        return
    height = (ac - base_dir * ac_off).astype(F32)
    height_length = _norm(height)
    height_dir = (height / height_length).astype(F32)
    tan_alpha = F32(height_length / ac_off)
    tan_beta = F32(height_length / bc_off)
    for i in range(1, int(base_step_count)):
        base_pos = (a + F32(i) * base_step).astype(F32)
        h_ac = tan_alpha * _norm(base_pos - a)
        h_bc = tan_beta * _norm(base_pos - b)
        hl = min(h_ac, h_bc)
        steps = np.ceil(hl / spacing)
        hstep = (height_dir * spacing).astype(F32)
        for j in range(1, int(steps)):
            pt = (base_pos + F32(j) * hstep).astype(F32)
            if np.all(np.isfinite(pt)):
                out.append(pt)


def sample_mesh(vertices: np.ndarray, indices: np.ndarray, spacing_xy: float) -> Tuple[np.ndarray, np.ndarray]:
    """particle3d.rs:250-299 -> (points [m, 3] float32, triangle id per point [m] uint32): triangle interiors,
    then each edge once (the first triangle that brings it), never the vertices."""
    v = np.asarray(vertices, F32)
    pts: List[np.ndarray] = []
    tri: List[int] = []
    visited = set()

    def needs(ia, ib):
        key = (max(ia, ib), min(ia, ib))
        if key in visited:
            return False
        visited.add(key)
        return True

    for t, idx in enumerate(np.asarray(indices, np.int64)):
        before = len(pts)
        sample_triangle(v[idx[0]], v[idx[1]], v[idx[2]], spacing_xy, pts)
        for ia, ib in ((idx[0], idx[1]), (idx[1], idx[2]), (idx[2], idx[0])):
            if needs(int(ia), int(ib)):
                sample_edge(v[ia], v[ib], spacing_xy, pts)
        tri += [t] * (len(pts) - before)
    if not pts:
        return np.zeros((0, 3), F32), np.zeros(0, np.uint32)
    return np.stack(pts).astype(F32), np.asarray(tri, np.uint32)


def sample_polyline(vertices: np.ndarray, indices: np.ndarray, sampling_step: float) -> Tuple[np.ndarray, np.ndarray]:
    """particle2d.rs:206-234 -> (points [m, 2], segment id per point): a, then a + k step while k step <= length
    (k = 0 repeats a, as in the reference), then b."""
    v = np.asarray(vertices, F32)
    pts: List[np.ndarray] = []
    seg: List[int] = []
    step = F32(sampling_step)
    for s, idx in enumerate(np.asarray(indices, np.int64)):
        a, b = v[idx[0]], v[idx[1]]
        pts.append(a.copy()); seg.append(s)
        ab = (b - a).astype(F32)
        length = _norm(ab)
        if length > 0:        # parry Segment::direction(): None for a degenerate segment
            d = (ab / length).astype(F32)
            i = 0
            while True:
                shift = F32(i) * step
                if shift > length:
                    break
                pts.append((a + d * shift).astype(F32)); seg.append(s)
                i += 1
            pts.append(b.copy()); seg.append(s)
    if not pts:
        return np.zeros((0, 2), F32), np.zeros(0, np.uint32)
    return np.stack(pts).astype(F32), np.asarray(seg, np.uint32)


def heightfield_to_trimesh(heights: np.ndarray, scale: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """parry HeightField::to_trimesh (third party, restated): heights[i, j] on a regular grid spanning
    [-scale.x/2, scale.x/2] x [-scale.z/2, scale.z/2], two triangles per cell."""
    hts = np.asarray(heights, F32)
    nr, nc = hts.shape
    xs = (np.arange(nc, dtype=F32) / F32(nc - 1) - F32(0.5)) * F32(scale[0])
    zs = (np.arange(nr, dtype=F32) / F32(nr - 1) - F32(0.5)) * F32(scale[2])
    vtx = np.zeros((nr * nc, 3), F32)
    for i in range(nr):
        for j in range(nc):
            vtx[i * nc + j] = (xs[j], hts[i, j] * F32(scale[1]), zs[i])
    idx = []
    for i in range(nr - 1):
        for j in range(nc - 1):
            p00, p01, p10, p11 = i * nc + j, i * nc + j + 1, (i + 1) * nc + j, (i + 1) * nc + j + 1
            idx += [[p00, p10, p01], [p10, p11, p01]]
    return vtx, np.asarray(idx, np.uint32)


def build_rigid_particles(colliders, dim: int, sampling_step: float):
    """All mesh colliders of a scene -> the buffers of GpuRigidParticles + the shape vertex buffers of wgrapier's
    GpuBodySet: dict(local_pts [n, D] f32, ids [n, 4] u32 (primitive vertex ids with the collider's base vertex id
    added, collider id last), local_vtx [nv, D] f32, vtx_collider [nv] u32). Sampling step = cell width
    (src/pipeline.rs:144)."""
    pts, ids, vtx, vcol = [], [], [], []
    base = 0
    for cid, c in enumerate(colliders):
        if getattr(c, "vertices", None) is None:
            continue
        v = np.asarray(c.vertices, F32).reshape(-1, dim)
        ind = np.asarray(c.indices, np.uint32).reshape(-1, dim)
        if dim == 3:
            p, prim = sample_mesh(v, ind, sampling_step)
        else:
            p, prim = sample_polyline(v, ind, sampling_step)
        rec = np.zeros((len(p), 4), np.uint32)
        rec[:, :dim] = ind[prim] + base
        rec[:, 3] = cid
        pts.append(p); ids.append(rec); vtx.append(v); vcol.append(np.full(len(v), cid, np.uint32))
        base += len(v)
    if not pts:
        return None
    return dict(local_pts=np.concatenate(pts).astype(F32), ids=np.concatenate(ids).astype(np.uint32),
                local_vtx=np.concatenate(vtx).astype(F32), vtx_collider=np.concatenate(vcol).astype(np.uint32))
