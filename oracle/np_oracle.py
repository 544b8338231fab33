"""np_oracle.py — second, independent CPU restatement of the MLS-MPM substep (numpy, fp64).

TEST INFRASTRUCTURE ONLY (imported by tests/ only). Parity unpinned, like
mpm_oracle.c: the reference cannot be run here, so this twin exists to catch
TRANSCRIPTION errors — it is written from Appendix A of SURVEY.md and the WGSL
(cited per function, paths relative to /root/reference/src), with a different
algorithmic structure than the C oracle (vectorised per-particle SCATTER with
np.add.at instead of the per-node gather over linked lists), dense dictionary
grid instead of the hash map, numpy's LAPACK SVD instead of the Jacobi SVD.
tests/test_oracle_twin.py requires both to agree to fp64 round-off.

No colliders (CPIC) here: those paths are cross-checked by invariants instead.
"""
from __future__ import annotations

import numpy as np

# grid/kernel.wgsl:22-50 (3D) / :7-17 (2D): the neighbourhood is {0,1,2}^D
MODEL_COROTATED, MODEL_NEO_HOOKEAN = 0, 1


def assoc_cell(pos_f32: np.ndarray, h: float) -> np.ndarray:
    """particle3d.wgsl:41-49: round(x / h) - 1, fp32 division, ties to even (np.rint)."""
    q = pos_f32.astype(np.float32) / np.float32(h)
    return (np.rint(q) - np.float32(1.0)).astype(np.int64)


def eval_all(x: np.ndarray) -> np.ndarray:
    """kernel.wgsl:60-66 -> [..., 3]"""
    return np.stack([0.5 * (1.5 - x) ** 2, 0.75 - (x - 1.0) ** 2, 0.5 * (x - 0.5) ** 2], axis=-1)


def _mat(a: np.ndarray, d: int) -> np.ndarray:
    """column-major [n, d*d] -> [n, d(row), d(col)]"""
    return a.reshape(-1, d, d).transpose(0, 2, 1)


def _unmat(m: np.ndarray) -> np.ndarray:
    n, d, _ = m.shape
    return m.transpose(0, 2, 1).reshape(n, d * d)


def kirchoff_stress(model: int, lam, mu, F):
    """linear_elasticity.wgsl:14-41 (corotated) / neo_hookean_elasticity.wgsl:12-25. F: [n,d,d]."""
    n, d, _ = F.shape
    eye = np.eye(d)
    if model == MODEL_NEO_HOOKEAN:
        j = np.maximum(np.linalg.det(F), 1.0e-10)
        diag = lam * np.log(j) - mu
        return mu[:, None, None] * (F @ F.transpose(0, 2, 1)) + diag[:, None, None] * eye
    U, S, Vt = np.linalg.svd(F)
    # proper rotations, sign on the smallest singular value (same convention as the C oracle)
    neg = np.linalg.det(U) * np.linalg.det(Vt) < 0
    S = S.copy()
    U = U.copy()
    S[neg, -1] *= -1.0
    U[neg, :, -1] *= -1.0
    j = np.prod(S, axis=1)
    R = U @ Vt
    diag = lam * (j - 1.0) * j
    return 2.0 * mu[:, None, None] * ((F - R) @ F.transpose(0, 2, 1)) + diag[:, None, None] * eye


def drucker_prager_project(dp, state, F):
    """drucker_prager.wgsl:112-158 (3D) / :42-101 (2D). dp [n,6], state [n,3], F [n,d,d]; returns (F', state')."""
    n, d, _ = F.shape
    F = F.copy()
    state = state.copy()
    on = dp[:, 4] != 0.0
    if not on.any():
        return F, state
    U, S, Vt = np.linalg.svd(F)
    neg = np.linalg.det(U) * np.linalg.det(Vt) < 0
    S = S.copy(); U = U.copy()
    S[neg, -1] *= -1.0
    U[neg, :, -1] *= -1.0
    q = state[:, 1]
    angle = dp[:, 0] + (dp[:, 1] * q - dp[:, 3]) * np.exp(-dp[:, 2] * q)
    sa = np.sin(angle)
    alpha = np.sqrt(2.0 / 3.0) * (2.0 * sa) / (3.0 - sa)
    with np.errstate(all="ignore"):
        strain = np.log(S) + (state[:, 2] / d)[:, None]
        tr = strain.sum(1)
        dev = strain - (tr / d)[:, None]
        dev_norm = np.linalg.norm(dev, axis=1)
        case_a = (tr > 0) | np.all(dev == 0.0, axis=1)
        gamma = dev_norm + (d * dp[:, 4] + 2.0 * dp[:, 5]) / (2.0 * dp[:, 5]) * tr * alpha
        valid = on & (case_a | (gamma > 0))
        new_S = np.where(case_a[:, None], 1.0, np.exp(strain - dev * (gamma / dev_norm)[:, None]))
        hard = np.where(case_a, np.linalg.norm(strain, axis=1), gamma)
        prev_det = np.prod(S, axis=1)
        new_det = np.prod(new_S, axis=1)
        st0 = state[:, 0] * prev_det / new_det
        st2 = state[:, 2] + np.log(prev_det) - np.log(new_det)
        st1 = state[:, 1] + hard
    newF = (U * new_S[:, None, :]) @ Vt
    F[valid] = newF[valid]
    state[valid, 0] = st0[valid]
    state[valid, 1] = st1[valid]
    state[valid, 2] = st2[valid]
    return F, state


class NpState:
    """fp64 particle state + one substep (no colliders)."""

    def __init__(self, particles, params, cell_width, model=0):
        self.d = particles.dim
        f = lambda a: np.asarray(a, np.float64).copy()
        self.pos, self.vel = f(particles.pos), f(particles.vel)
        self.F, self.C = f(particles.def_grad), f(particles.affine)
        self.mass, self.vol = f(particles.mass), f(particles.init_volume)
        self.lam, self.mu = f(particles.lambda_), f(particles.mu)
        self.dp, self.dp_state, self.phase = f(particles.dp), f(particles.dp_state), f(particles.phase)
        self.g = np.asarray(list(params.gravity), np.float64)
        self.dt = float(params.dt)
        self.h = float(cell_width)
        self.model = int(model)
        self.grid = {}

    def step(self, n=1):
        for _ in range(n):
            self._substep()

    def _substep(self):
        d, h, dt = self.d, self.h, self.dt
        n = self.pos.shape[0]
        cell = assoc_cell(self.pos.astype(np.float32), h)               # [n, d] associated cell (bit-exact rule)
        ref = cell * h - self.pos                                         # dir_to_associated_grid_node
        w = eval_all(-ref / h)                                            # [n, d, 3]
        shifts = np.stack(np.meshgrid(*([np.arange(3)] * d), indexing="ij"), -1).reshape(-1, d)  # [3^d, d]
        # weights and offsets of the 3^d nodes of every particle
        wn = np.ones((n, len(shifts)))
        for k in range(d):
            wn *= w[:, k, shifts[:, k]]
        dpt = ref[:, None, :] + shifts[None, :, :] * h                    # node - particle  [n, S, d]
        node = cell[:, None, :] + shifts[None, :, :]                      # [n, S, d] world node coordinates
        # ---- P2G (p2g.wgsl:176-236): node += w * (C' dpt + m v, m)
        Cm = _mat(self.C, d)
        mom = np.einsum("nrc,nsc->nsr", Cm, dpt) + (self.mass[:, None] * self.vel)[:, None, :]
        lo = node.reshape(-1, d).min(0)
        ext = node.reshape(-1, d).max(0) - lo + 1
        flat = np.ravel_multi_index(tuple((node - lo).reshape(-1, d).T), tuple(ext)).reshape(n, -1)
        gm = np.zeros((int(np.prod(ext)), d))
        gmass = np.zeros(int(np.prod(ext)))
        np.add.at(gm, flat.reshape(-1), (mom * wn[:, :, None]).reshape(-1, d))
        np.add.at(gmass, flat.reshape(-1), (self.mass[:, None] * wn).reshape(-1))
        # ---- grid update (grid_update.wgsl:55-64)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.where(gmass > 0, 1.0 / gmass, 0.0)
        gv = (gm + gmass[:, None] * self.g[None, :] * dt) * inv[:, None]
        gv = np.clip(gv, -h / dt, h / dt)
        # keep the grid for comparisons (only nodes of active blocks matter; compared by coordinate)
        self.grid = dict(lo=lo, ext=ext, vel=gv, mass=gmass)
        # ---- G2P (g2p.wgsl:150-218)
        nv = gv[flat]                                                     # [n, S, d]
        vel = np.einsum("ns,nsr->nr", wn, nv)
        invd = 4.0 / (h * h)
        grad = invd * np.einsum("ns,nsr,nsc->nrc", wn, nv, dpt)          # [n, r, c]
        # ---- particle update (particle_update.wgsl:66-132), no colliders
        speed = np.linalg.norm(vel, axis=1)
        too_fast = speed > h / dt
        vel[too_fast] = vel[too_fast] / speed[too_fast, None] * h / dt
        self.pos = self.pos + vel * dt
        Fm = _mat(self.F, d)
        Fm = Fm + (grad * dt) @ Fm
        phase = self.phase[:, 0].copy()
        max_stretch = self.phase[:, 1]
        chk = (phase > 0) & (max_stretch > 0)
        if chk.any():
            S = np.linalg.svd(Fm[chk], compute_uv=False)
            broken = (S > max_stretch[chk, None]).any(1)
            idx = np.nonzero(chk)[0][broken]
            phase[idx] = 0.0
        plastic = phase == 0.0
        if plastic.any():
            Fp, st = drucker_prager_project(self.dp[plastic], self.dp_state[plastic], Fm[plastic])
            Fm[plastic] = Fp
            self.dp_state[plastic] = st
        self.phase[:, 0] = phase
        tau = kirchoff_stress(self.model, self.lam, self.mu, Fm)
        Cn = grad * self.mass[:, None, None] - tau * (self.vol * invd * dt)[:, None, None]
        self.vel = vel
        self.F = _unmat(Fm)
        self.C = _unmat(Cn)

    def grid_at(self, cells: np.ndarray):
        """(velocity, mass) of the given world node coordinates (zeros outside the touched box)."""
        g = self.grid
        rel = cells - g["lo"]
        ok = np.all((rel >= 0) & (rel < g["ext"]), axis=1)
        vel = np.zeros((len(cells), self.d))
        mass = np.zeros(len(cells))
        flat = np.ravel_multi_index(tuple(rel[ok].T), tuple(g["ext"]))
        vel[ok] = g["vel"][flat]
        mass[ok] = g["mass"][flat]
        return vel, mass


# ---------------------------------------------------------------------------------------------
# Render hand-off: src_testbed/prep_vertex_buffer{2,3}d.wgsl `main`, restated on arrays (tests only)
# ---------------------------------------------------------------------------------------------
RENDER_DEFAULT, RENDER_VOLUME, RENDER_VELOCITY, RENDER_CDF_NORMALS, RENDER_CDF_DISTANCES, RENDER_CDF_SIGNS = range(6)


def prep_instances(pos, vel, def_grad, cdf_normal, cdf_dist, cdf_affinity, mode, cell_width, dt, base_color):
    """-> [n, 24] float64: deformation (3 padded columns), position (vec4), base_color, color.
    prep_vertex_buffer3d.wgsl:45-95 / prep_vertex_buffer2d.wgsl:40-90. Singular values in descending order."""
    n, d = pos.shape
    out = np.zeros((n, 24))
    m = np.tile(np.eye(3), (n, 1, 1))                       # [n, row, col]
    m[:, :d, :d] = _mat(np.asarray(def_grad, np.float64), d)
    for c in range(3):
        out[:, 4 * c:4 * c + 3] = m[:, :, c]
    out[:, 12:12 + d] = pos
    out[:, 16:20] = base_color
    col = np.array(base_color, np.float64).copy().reshape(n, 4)
    if mode == RENDER_VELOCITY:
        col[:, :d] = np.abs(vel) * dt * 100.0 + 0.2
    elif mode == RENDER_VOLUME:
        S = np.linalg.svd(_mat(np.asarray(def_grad, np.float64), d), compute_uv=False)   # descending
        col[:, :d] = (1.0 - S) / 0.005 + 0.2
    elif mode == RENDER_CDF_NORMALS:
        zero = np.all(cdf_normal == 0.0, axis=1)
        col[:, :3] = 0.0
        col[~zero, :d] = (cdf_normal[~zero] + 1.0) / 2.0
    elif mode == RENDER_CDF_DISTANCES:
        dd = cdf_dist / (cell_width * 1.5)
        col[:, 0] = np.where(dd > 0, 0.0, np.abs(dd))
        col[:, 1] = np.where(dd > 0, np.abs(dd), 0.0)
        col[:, 2] = 0.0
    elif mode == RENDER_CDF_SIGNS:
        aff = np.asarray(cdf_affinity, np.uint32)
        a = (aff >> 16) & (aff & 0xFFFF)
        col[:, 0] = ((aff != 0) & (a != 0)).astype(np.float64)
        col[:, 1] = ((aff != 0) & (a == 0)).astype(np.float64)
        col[:, 2] = 0.0
    out[:, 20:24] = col
    return out
