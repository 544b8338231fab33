"""ctypes front-end of the CPU oracle (oracle/mpm_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package (wgsparkl_amd/) never imports
this module. Parity status: see the header of mpm_oracle.h ("parity unpinned"
except the prefix sum).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.environ.get("WGS_ORACLE_BUILD_DIR") or os.path.join(_HERE, "_build")   # (override: sanitizer builds)
NONE = 0xFFFFFFFF


def build(force: bool = False) -> None:
    """Compile the four oracle flavours with gcc (see oracle/Makefile)."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def _lib_path(dim: int, dtype, omp: bool = False) -> str:
    suffix = "f32" if np.dtype(dtype) == np.float32 else "f64"
    return os.path.join(_BUILD, f"liborc{dim}d_{suffix}{'_omp' if omp else ''}.so")


class Oracle:
    """One (dim, precision) flavour of the oracle."""

    def __init__(self, dim: int = 3, dtype=np.float32, omp: bool = False):
        """`omp`: the OpenMP build of the same source (3D fp32 only; bench.py's multi-core CPU baseline)."""
        path = _lib_path(dim, dtype, omp)
        if not os.path.exists(path):
            build()
        self.dim = dim
        self.dtype = np.dtype(dtype)
        self.real = C.c_float if self.dtype == np.float32 else C.c_double
        self.lib = C.CDLL(path)
        assert self.lib.orc_dim() == dim and self.lib.orc_real_size() == self.dtype.itemsize
        self.num_threads = int(self.lib.orc_num_threads())
        R = self.real
        RP = C.POINTER(R)
        U32P = C.POINTER(C.c_uint32)
        I32P = C.POINTER(C.c_int32)

        class Particles(C.Structure):
            _fields_ = [("n", C.c_int32)] + [(k, RP) for k in ("pos", "vel", "def_grad", "affine", "cdf_normal",
                                                                 "cdf_rigid_vel", "cdf_dist")] + \
                       [("cdf_affinity", U32P)] + \
                       [(k, RP) for k in ("init_volume", "mass", "lambda_", "mu", "dp", "dp_state", "phase")]

        class Collider(C.Structure):
            _fields_ = [("shape_type", C.c_int32), ("shape", R * 4), ("rot", R * 4), ("trans", R * 3),
                        ("scale", R), ("linvel", R * 3), ("angvel", R * 3), ("com", R * 3)]

        class Body(C.Structure):
            _fields_ = [("inv_mass", R * 3), ("inv_inertia_local", R * 9), ("local_com", R * 3),
                        ("inv_inertia_world", R * 9)]

        class Rigid(C.Structure):
            _fields_ = [("n", C.c_int32), ("local_pts", RP), ("world_pts", RP), ("ids", U32P), ("nv", C.c_int32),
                        ("local_vtx", RP), ("world_vtx", RP), ("vtx_collider", U32P), ("needs_block", U32P),
                        ("node_head", U32P), ("node_len", U32P), ("next", U32P)]

        class Params(C.Structure):
            _fields_ = [("gravity", R * 3), ("dt", R), ("cell_width", R), ("model", C.c_int32),
                        ("n_colliders", C.c_int32), ("colliders", C.POINTER(Collider))]

        class Grid(C.Structure):
            _fields_ = [("cap_blocks", C.c_int32), ("hmap_capacity", C.c_int32), ("n_blocks", C.c_int32),
                        ("overflow", C.c_int32), ("hmap_state", U32P), ("hmap_value", U32P),
                        ("block_vid", I32P), ("first_particle", U32P), ("num_particles", U32P),
                        ("sorted_ids", U32P), ("node_head", U32P), ("node_len", U32P),
                        ("particle_next", U32P), ("node_mv", RP), ("node_cdf_dist", RP),
                        ("node_cdf_aff", U32P), ("node_cdf_closest", U32P), ("impulses", I32P)]

        self.Particles, self.Collider, self.Params, self.Grid, self.Body, self.Rigid = Particles, Collider, Params, Grid, Body, Rigid
        L = self.lib
        L.orc_pack_key.restype = C.c_uint32
        L.orc_pack_key.argtypes = [I32P]
        L.orc_hash.restype = C.c_uint32
        L.orc_hash.argtypes = [C.c_uint32]
        L.orc_assoc_cell.argtypes = [C.POINTER(C.c_float), C.c_float, I32P]
        L.orc_block_and_local.argtypes = [C.POINTER(C.c_float), C.c_float, I32P, U32P]
        L.orc_prefix_sum_eval_cpu.argtypes = [U32P, C.c_int32]
        L.orc_prefix_sum_gpu_algorithm.argtypes = [U32P, C.c_int32]
        L.orc_eval_all.argtypes = [R, RP]
        L.orc_svd.argtypes = [RP, RP, RP, RP]
        L.orc_kirchoff_stress.argtypes = [C.c_int, R, R, RP, RP]
        L.orc_drucker_prager_project.argtypes = [RP, RP, RP]
        L.orc_drucker_prager_project.restype = C.c_int
        for name in ("orc_sort", "orc_p2g"):
            getattr(L, name).argtypes = [C.POINTER(Particles), C.POINTER(Params), C.POINTER(Grid)]
        for name in ("orc_g2p_cdf", "orc_g2p"):
            getattr(L, name).argtypes = [C.POINTER(Particles), C.POINTER(Params), C.POINTER(Grid)]
        for name in ("orc_grid_update_cdf", "orc_grid_update"):
            getattr(L, name).argtypes = [C.POINTER(Params), C.POINTER(Grid)]
        L.orc_particle_update.argtypes = [C.POINTER(Particles), C.POINTER(Params)]
        L.orc_step.argtypes = [C.POINTER(Particles), C.POINTER(Params), C.POINTER(Grid), C.c_int]

    # ---- small helpers -------------------------------------------------
    def _p(self, a, ctype=None):
        return a.ctypes.data_as(C.POINTER(ctype or self.real))

    def pack_key(self, block) -> int:
        b = np.ascontiguousarray(block, np.int32)
        return int(self.lib.orc_pack_key(b.ctypes.data_as(C.POINTER(C.c_int32))))

    def hash(self, key: int) -> int:
        return int(self.lib.orc_hash(C.c_uint32(key)))

    def block_and_local(self, pt, h):
        p = np.ascontiguousarray(pt, np.float32)
        b = np.zeros(3, np.int32)
        l = np.zeros(3, np.uint32)
        self.lib.orc_block_and_local(p.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(h),
                                     b.ctypes.data_as(C.POINTER(C.c_int32)), l.ctypes.data_as(C.POINTER(C.c_uint32)))
        return b[:self.dim].copy(), l[:self.dim].copy()

    def assoc_cells(self, pos, h):
        pos = np.ascontiguousarray(pos, np.float32)
        out = np.zeros((pos.shape[0], self.dim), np.int32)
        tmp = np.zeros(3, np.int32)
        for i in range(pos.shape[0]):
            self.lib.orc_assoc_cell(pos[i].ctypes.data_as(C.POINTER(C.c_float)), C.c_float(h),
                                    tmp.ctypes.data_as(C.POINTER(C.c_int32)))
            out[i] = tmp[:self.dim]
        return out

    def prefix_sum_eval_cpu(self, v):
        a = np.ascontiguousarray(v, np.uint32).copy()
        self.lib.orc_prefix_sum_eval_cpu(self._p(a, C.c_uint32), len(a))
        return a

    def prefix_sum_gpu_algorithm(self, v):
        a = np.ascontiguousarray(v, np.uint32).copy()
        self.lib.orc_prefix_sum_gpu_algorithm(self._p(a, C.c_uint32), len(a))
        return a

    def eval_all(self, x):
        w = np.zeros(3, self.dtype)
        self.lib.orc_eval_all(self.real(x), self._p(w))
        return w

    def svd(self, m):
        m = np.ascontiguousarray(m, self.dtype).reshape(-1)
        d = self.dim
        u, s, vt = np.zeros(d * d, self.dtype), np.zeros(d, self.dtype), np.zeros(d * d, self.dtype)
        self.lib.orc_svd(self._p(m), self._p(u), self._p(s), self._p(vt))
        return u, s, vt

    def kirchoff_stress(self, model, lam, mu, F):
        F = np.ascontiguousarray(F, self.dtype).reshape(-1)
        tau = np.zeros_like(F)
        self.lib.orc_kirchoff_stress(int(model), self.real(lam), self.real(mu), self._p(F), self._p(tau))
        return tau

    def drucker_prager_project(self, dp6, state3, F):
        dp6 = np.ascontiguousarray(dp6, self.dtype)
        st = np.ascontiguousarray(state3, self.dtype).copy()
        Fm = np.ascontiguousarray(F, self.dtype).reshape(-1).copy()
        changed = self.lib.orc_drucker_prager_project(self._p(dp6), self._p(st), self._p(Fm))
        return bool(changed), st, Fm

    # ---- simulation state ----------------------------------------------
    def new_state(self, particles, params, colliders, cell_width, grid_capacity, model=0):
        return OracleState(self, particles, params, colliders, cell_width, grid_capacity, model)


def _is_moving(c) -> bool:
    return any(float(v) != 0.0 for k in ("linvel", "angvel", "inv_mass", "inv_inertia_local")
               for v in getattr(c, k, ()))


def _quat_rotate(q, v):
    u, w = q[:3], q[3]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


class OracleState:
    """Particles + grid in the oracle's precision; `step()` advances it."""

    FIELDS = ("pos", "vel", "def_grad", "affine", "cdf_normal", "cdf_rigid_vel", "cdf_dist",
              "init_volume", "mass", "lambda_", "mu", "dp", "dp_state", "phase")

    def __init__(self, orc: Oracle, particles, params, colliders, cell_width, grid_capacity, model):
        self.orc = orc
        dt = orc.dtype
        D = orc.dim
        assert particles.dim == D
        self.n = particles.n
        self.arr = {k: np.ascontiguousarray(getattr(particles, k), dt).copy() for k in self.FIELDS}
        self.arr["cdf_affinity"] = particles.cdf_affinity.astype(np.uint32).copy()
        self.init_radius = particles.init_radius.copy()
        P = orc.Particles()
        P.n = self.n
        for k in self.FIELDS:
            setattr(P, k, orc._p(self.arr[k]))
        P.cdf_affinity = orc._p(self.arr["cdf_affinity"], C.c_uint32)
        self.P = P

        self.cols = (orc.Collider * max(1, len(colliders)))()
        for i, c in enumerate(colliders):
            self._fill_collider(self.cols[i], c)
        self.bodies = (orc.Body * max(1, len(colliders)))()
        for i, c in enumerate(colliders):
            self._fill_body(self.bodies[i], self.cols[i], c)
        self.n_colliders = len(colliders)
        # like the HIP host side: bodies that cannot move skip the (identity) integration pass
        self.moving = any(_is_moving(c) for c in colliders)
        prm = orc.Params()
        g = list(params.gravity) + [0.0] * (3 - len(params.gravity))
        prm.gravity = (orc.real * 3)(*g)
        prm.dt = params.dt
        prm.cell_width = cell_width
        prm.model = int(model)
        prm.n_colliders = len(colliders)
        prm.colliders = C.cast(self.cols, C.POINTER(orc.Collider))
        self.prm = prm

        cap = int(grid_capacity)
        hcap = 1 << int(np.ceil(np.log2(max(cap, 2))))
        self.cap, self.hcap = cap, hcap
        nn = cap * 64
        self.g = {
            "hmap_state": np.zeros(hcap, np.uint32), "hmap_value": np.zeros(hcap, np.uint32),
            "block_vid": np.zeros((cap, D), np.int32), "first_particle": np.zeros(cap, np.uint32),
            "num_particles": np.zeros(cap, np.uint32), "sorted_ids": np.zeros(max(self.n, 1), np.uint32),
            "node_head": np.zeros(nn, np.uint32), "node_len": np.zeros(nn, np.uint32),
            "particle_next": np.zeros(max(self.n, 1), np.uint32), "node_mv": np.zeros((nn, D + 1), dt),
            "node_cdf_dist": np.zeros(nn, dt), "node_cdf_aff": np.zeros(nn, np.uint32),
            "node_cdf_closest": np.zeros(nn, np.uint32), "impulses": np.zeros(16 * (D + (1 if D == 2 else 3)), np.int32),
        }
        # rigid particles of the mesh colliders (sampled on the host like the reference does)
        from oracle.ref_sampling import build_rigid_particles
        rb = build_rigid_particles(colliders, D, float(cell_width))
        self.R = None
        if rb is not None:
            n_r, n_v = len(rb["local_pts"]), len(rb["local_vtx"])
            self.rig = {"local_pts": rb["local_pts"].astype(dt), "world_pts": np.zeros((n_r, D), dt),
                        "ids": np.ascontiguousarray(rb["ids"], np.uint32), "local_vtx": rb["local_vtx"].astype(dt),
                        "world_vtx": np.zeros((n_v, D), dt), "vtx_collider": rb["vtx_collider"],
                        "needs_block": np.zeros(n_r, np.uint32), "node_head": np.zeros(nn, np.uint32),
                        "node_len": np.zeros(nn, np.uint32), "next": np.zeros(n_r, np.uint32)}
            Rg = orc.Rigid()
            Rg.n, Rg.nv = n_r, n_v
            for k, a in self.rig.items():
                setattr(Rg, k, orc._p(a, C.c_uint32 if a.dtype == np.uint32 else orc.real))
            self.R = Rg
        G = orc.Grid()
        G.cap_blocks, G.hmap_capacity = cap, hcap
        for k, a in self.g.items():
            ct = {np.dtype(np.uint32): C.c_uint32, np.dtype(np.int32): C.c_int32}.get(a.dtype, orc.real)
            setattr(G, k, orc._p(a, ct))
        self.G = G

    def _fill_collider(self, dst, c):
        orc, D = self.orc, self.orc.dim
        R = orc.real
        dst.shape_type = int(c.shape_type)
        dst.shape = (R * 4)(*(list(c.shape) + [0.0] * (4 - len(c.shape))))
        if D == 2:
            ang = float(c.rotation[0])
            dst.rot = (R * 4)(float(np.cos(np.float32(ang))), float(np.sin(np.float32(ang))), 0.0, 0.0)
        else:
            dst.rot = (R * 4)(*c.rotation)
        t = list(c.translation) + [0.0] * (3 - len(c.translation))
        dst.trans = (R * 3)(*t)
        dst.scale = c.scale
        dst.linvel = (R * 3)(*(list(c.linvel)[:D] + [0.0] * (3 - D)))
        dst.angvel = (R * 3)(*(list(c.angvel) + [0.0] * (3 - len(c.angvel)))[:3])
        com = list(c.com) if c.com is not None else list(c.translation)
        dst.com = (R * 3)(*(com + [0.0] * (3 - len(com))))

    def _fill_body(self, dst, col, c):
        """Mass properties; the local centre of mass is the world one mapped through the inverse pose."""
        orc, D = self.orc, self.orc.dim
        R = orc.real
        im = list(getattr(c, "inv_mass", (0.0,) * 3))
        dst.inv_mass = (R * 3)(*(im + [0.0] * (3 - len(im)))[:3])
        ii = list(getattr(c, "inv_inertia_local", (0.0,) * 9))
        dst.inv_inertia_local = (R * 9)(*(ii + [0.0] * (9 - len(ii)))[:9])
        d = np.array([col.com[k] - col.trans[k] for k in range(3)], np.float64)
        if D == 2:
            cs, sn = col.rot[0], col.rot[1]
            loc = np.array([cs * d[0] + sn * d[1], -sn * d[0] + cs * d[1], 0.0]) / col.scale
        else:
            loc = _quat_rotate(np.array([-col.rot[0], -col.rot[1], -col.rot[2], col.rot[3]]), d) / col.scale
        dst.local_com = (R * 3)(*[float(x) for x in loc])

    def set_colliders(self, colliders):
        for i, c in enumerate(colliders):
            self._fill_collider(self.cols[i], c)
            self._fill_body(self.bodies[i], self.cols[i], c)
        self.moving = any(_is_moving(c) for c in colliders)

    def collider_states(self):
        """Current (rotation, translation, linvel, angvel, com) of every collider, as float64 arrays."""
        D = self.orc.dim
        out = []
        for i in range(self.n_colliders):
            c = self.cols[i]
            out.append(dict(rotation=np.array(list(c.rot)[:(2 if D == 2 else 4)], np.float64),
                            translation=np.array(list(c.trans)[:D], np.float64),
                            linvel=np.array(list(c.linvel)[:D], np.float64),
                            angvel=np.array(list(c.angvel)[:(1 if D == 2 else 3)], np.float64),
                            com=np.array(list(c.com)[:D], np.float64)))
        return out

    def impulses(self):
        return self.g["impulses"].copy()

    def set_params(self, params):
        g = list(params.gravity) + [0.0] * (3 - len(params.gravity))
        self.prm.gravity = (self.orc.real * 3)(*g)
        self.prm.dt = params.dt

    # passes
    def sort(self): self.orc.lib.orc_sort(C.byref(self.P), C.byref(self.prm), C.byref(self.G))
    def grid_update_cdf(self): self.orc.lib.orc_grid_update_cdf(C.byref(self.prm), C.byref(self.G))
    def g2p_cdf(self): self.orc.lib.orc_g2p_cdf(C.byref(self.P), C.byref(self.prm), C.byref(self.G))
    def p2g(self): self.orc.lib.orc_p2g(C.byref(self.P), C.byref(self.prm), C.byref(self.G))
    def grid_update(self): self.orc.lib.orc_grid_update(C.byref(self.prm), C.byref(self.G))
    def g2p(self): self.orc.lib.orc_g2p(C.byref(self.P), C.byref(self.prm), C.byref(self.G))
    def particle_update(self): self.orc.lib.orc_particle_update(C.byref(self.P), C.byref(self.prm))

    def update_world_mass_properties(self):
        self.orc.lib.orc_update_world_mass_properties(self.cols, self.bodies, self.n_colliders)

    def integrate_bodies(self):
        self.orc.lib.orc_integrate_bodies(C.byref(self.prm), self.cols, self.bodies, self.n_colliders, self.G.impulses)

    def step(self, n_substeps=1):
        if self.R is not None:  # mesh colliders: the complete pipeline.rs:201-280 order
            self.orc.lib.orc_step_full(C.byref(self.P), C.byref(self.prm), C.byref(self.G), self.cols, self.bodies,
                                       C.byref(self.R), 1 if self.moving else 0, int(n_substeps))
        elif self.moving:  # pipeline.rs:201-280 with the rigid-body passes (rigid_impulses.wgsl)
            self.orc.lib.orc_step_bodies(C.byref(self.P), C.byref(self.prm), C.byref(self.G), self.cols, self.bodies,
                                         int(n_substeps))
        else:
            self.orc.lib.orc_step(C.byref(self.P), C.byref(self.prm), C.byref(self.G), int(n_substeps))

    @property
    def n_blocks(self): return int(self.G.n_blocks)

    @property
    def overflow(self): return bool(self.G.overflow)

    def grid_records(self):
        """Active nodes keyed by world cell coordinate, sorted lexicographically:
        (cells[int32 M x D], mv[M x (D+1)], cdf_dist, cdf_aff, cdf_closest)."""
        D = self.orc.dim
        nb = self.n_blocks
        bw = 8 if D == 2 else 4
        vid = self.g["block_vid"][:nb]
        t = np.arange(64)
        if D == 2:
            loc = np.stack([t % 8, t // 8], -1)
        else:
            loc = np.stack([t % 4, (t // 4) % 4, t // 16], -1)
        cells = (vid[:, None, :] * bw + loc[None, :, :]).reshape(-1, D).astype(np.int32)
        order = np.lexsort(cells.T[::-1])
        nn = nb * 64
        return (cells[order], self.g["node_mv"][:nn][order], self.g["node_cdf_dist"][:nn][order],
                self.g["node_cdf_aff"][:nn][order], self.g["node_cdf_closest"][:nn][order])

    def blocks(self):
        """(vid, first_particle, num_particles) sorted by vid."""
        nb = self.n_blocks
        vid = self.g["block_vid"][:nb]
        order = np.lexsort(vid.T[::-1])
        return vid[order], self.g["first_particle"][:nb][order], self.g["num_particles"][:nb][order]
