"""Multi-GPU host glue: x-slab domain decomposition of the MLS-MPM substep, one process per GPU.

NEW DESIGN — the reference is single-GPU (one wgpu::Device, src/pipeline.rs:176-193).
The substep protocol itself lives in the library (include/wgsparkl_hip.h "Multi-GPU": `wgs_comm_*`, `wgs_shard_attach`,
`wgs_sharded_step`; wgsparkl_amd/csrc/kernels_shard.h, capi_sharded.inc) — ONE implementation, in C++, RCCL
point-to-point as its transport. This module only holds what a host needs around it: the partition of block space into
slabs, thin ctypes wrappers of a communicator and a slab, and the record layout of `wgs_shard_export`.

Why x slabs and what crosses a face: a particle with associated cell c touches nodes c..c+2 only, so a rank's particles
reach the first two node layers of the block layer owned by the NEXT rank and nothing on the lower side. Per substep ONE
message per neighbour: the partial (momentum, mass) sums of the node layers the two ranks share (both add what they
receive: a + b == b + a bitwise, so both hold identical totals and update them redundantly) and, in the same message,
the records of the particles that changed owner in the previous substep — the old owner still transfers them to the
grid, the new owner advances them (kernels_shard.h). A slab with two neighbours must be at least 3 blocks wide.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .models import MODEL_COROTATED
from .pipeline import MpmPipeline, _fill_collider, _pack_particles
from .solver import ParticleSet, SimulationParams

INT_MIN, INT_MAX = -(2 ** 31), 2 ** 31 - 1
MIN_INTERIOR_WIDTH = 3      # blocks; wgs_shard_attach refuses narrower slabs between two neighbours

@dataclass
class SlabPartition:
    """Contiguous block ranges along x: rank r owns bx in [cuts[r], cuts[r+1]); the two ends are open."""
    cuts: Sequence[int]          # world + 1 entries, in BLOCK units (4 cells in 3D, 8 in 2D)

    @property
    def world(self) -> int:
        return len(self.cuts) - 1

    def block_range(self, rank: int):
        lo = INT_MIN if rank == 0 else int(self.cuts[rank])
        hi = INT_MAX if rank == self.world - 1 else int(self.cuts[rank + 1])
        return lo, hi

    def owner_of_blocks(self, bx: np.ndarray) -> np.ndarray:
        inner = np.asarray(self.cuts[1:-1], np.int64)
        return np.searchsorted(inner, np.asarray(bx, np.int64), side="right")

    @staticmethod
    def balanced(block_x: np.ndarray, world: int) -> "SlabPartition":
        """Cuts at particle-count quantiles of the associated block x coordinate (balance by particle
        count, not by volume)."""
        bx = np.sort(np.asarray(block_x, np.int64))
        cuts = [int(bx[0])]
        for r in range(1, world):
            cuts.append(int(bx[(len(bx) * r) // world]))
        cuts.append(int(bx[-1]) + 1)
        for i in range(1, len(cuts)):          # strictly increasing; interior slabs wide enough for the protocol
            cuts[i] = max(cuts[i], cuts[i - 1] + (MIN_INTERIOR_WIDTH if 1 < i < len(cuts) - 1 else 1))
        return SlabPartition(cuts)

    def min_interior_width(self) -> int:
        """Narrowest slab that has two neighbours (blocks); a large number when there is none."""
        w = [self.cuts[r + 1] - self.cuts[r] for r in range(1, self.world - 1)]
        return int(min(w)) if w else 1 << 30


def associated_block_x(pos: np.ndarray, cell_width: float, dim: int) -> np.ndarray:
    """floor((round(x / h) - 1) / BW) with the reference's fp32 rule (particle3d.wgsl:41-49, grid.wgsl:284-292)."""
    bw = 4 if dim == 3 else 8
    c = np.rint(pos[:, 0].astype(np.float32) / np.float32(cell_width)) - np.float32(1.0)
    return np.floor(c / np.float32(bw)).astype(np.int64)



# ------------------------------------------------------------------------------------------------
# The substep driven from inside the library (include/wgsparkl_hip.h: wgs_comm_*, wgs_shard_attach,
# wgs_sharded_step): what bench.py --gpus N and a Rust caller use.
# ------------------------------------------------------------------------------------------------
class NativeComm:
    """RCCL communicator owned by the library (`wgs_comm_create`). The 128-byte unique id is made by rank 0 and
    handed around with whatever the host has — here torch.distributed's broadcast."""

    def __init__(self, pipeline: MpmPipeline, dist, rank: int, world: int, flags: int = 0):
        import torch
        self.lib, self.rank, self.world = pipeline.lib, rank, world
        uid = C.create_string_buffer(128)
        failure = None
        if rank == 0:
            try:
                _ffi.check(self.lib, self.lib.wgs_comm_get_unique_id(uid))
            except Exception as e:  # noqa: BLE001 — the other ranks are waiting in the broadcast below: send them zeros
                failure = e
                uid = C.create_string_buffer(128)
        if world > 1:
            t = torch.tensor(list(uid.raw), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", pipeline.device))
            dist.broadcast(t, 0)
            uid = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
        if failure is not None or not any(uid.raw):
            raise RuntimeError(f"no RCCL unique id from rank 0 ({failure})")   # on every rank alike
        h = C.c_void_p()
        _ffi.check(self.lib, self.lib.wgs_comm_create(pipeline._h, uid, rank, world, int(flags), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wgs_comm_destroy(self._h)
            self._h = None


class NativeShard:
    """One slab as a Rust caller would hold it: a sharded `wgs_data` whose message buffers and substep protocol live
    in the library. `comm` = NativeComm (one process per GPU) or None (a slab of a lockstep group / no neighbours)."""

    def __init__(self, pipeline: MpmPipeline, params: SimulationParams, particles: ParticleSet, global_ids: np.ndarray,
                 colliders, cell_width: float, grid_capacity: int, block_lo: int, block_hi: int, has_lower: bool,
                 has_upper: bool, particle_capacity: int, model: int = MODEL_COROTATED, force_plastic: bool = False,
                 halo_capacity_records: int = 4096, migrant_capacity: int = 4096, comm: Optional[NativeComm] = None,
                 uniform_material=None):
        """`uniform_material` = (mass, init_volume, lambda, mu) shared by EVERY particle of EVERY rank (the caller's
        promise: a rank only sees its own), or None: the constants then travel with each particle."""
        self.pipeline, self.lib, self.T = pipeline, pipeline.lib, pipeline.T
        T, D = self.T, pipeline.dim
        self.dim = D
        self.uniform_material = uniform_material
        self.n_colliders = len(colliders)
        sp = T.SimParams()
        sp.gravity = (C.c_float * D)(*params.gravity)
        sp.dt = params.dt
        raw = _pack_particles(T, particles)
        gids = np.ascontiguousarray(global_ids, np.uint32)
        cols = (T.Collider * max(1, len(colliders)))()
        for i, c in enumerate(colliders):
            _fill_collider(T, cols[i], c, D)
        h = C.c_void_p()
        cap = max(int(particle_capacity), particles.n)
        _ffi.check(self.lib, self.lib.wgs_data_create_sharded(
            pipeline._h, C.byref(sp), raw.ctypes.data_as(C.POINTER(T.Particle)), particles.n,
            gids.ctypes.data_as(C.POINTER(C.c_uint32)), cols, len(colliders), float(cell_width), int(grid_capacity),
            cap, max(int(block_lo), INT_MIN), min(int(block_hi), INT_MAX), 1 if force_plastic else 0, C.byref(h)))
        self._h = h
        self.capacity = cap
        if model != MODEL_COROTATED:
            _ffi.check(self.lib, self.lib.wgs_set_constitutive_model(self._h, int(model)))
        if uniform_material is not None:
            _ffi.check(self.lib, self.lib.wgs_set_uniform_material(self._h, *[float(x) for x in uniform_material]))
            _ffi.check(self.lib, self.lib.wgs_sync(self._h))       # (a particle with other constants is reported here)
        if any(any(c.inv_mass) or any(c.inv_inertia_local) for c in colliders):   # dynamic bodies: two-way coupling
            arr = (T.MassProperties * len(colliders))()
            for i, c in enumerate(colliders):
                arr[i].inv_mass = tuple(c.inv_mass)
                arr[i].inv_inertia_local = tuple(c.inv_inertia_local)
            _ffi.check(self.lib, self.lib.wgs_set_body_mass_properties(self._h, arr, len(colliders)))
        _ffi.check(self.lib, self.lib.wgs_shard_attach(self._h, comm._h if comm is not None else None, 1 if has_lower else 0,
                                                         1 if has_upper else 0, int(halo_capacity_records), int(migrant_capacity)))
        # mesh colliders: sampled on the host like GpuRigidParticles::from_rapier; every rank holds every sample
        from .sampling import build_rigid_particles
        rb = build_rigid_particles(colliders, D, float(cell_width))
        if rb is not None:
            F32 = np.float32
            pts, ids = np.ascontiguousarray(rb["local_pts"], F32), np.ascontiguousarray(rb["ids"], np.uint32)
            vtx, vcol = np.ascontiguousarray(rb["local_vtx"], F32), np.ascontiguousarray(rb["vtx_collider"], np.uint32)
            fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
            _ffi.check(self.lib, self.lib.wgs_set_rigid_particles(self._h, pts.ctypes.data_as(fp), ids.ctypes.data_as(C.c_void_p), len(pts),
                                                                  vtx.ctypes.data_as(fp), vcol.ctypes.data_as(up), len(vtx)))
        self.part_rec = self.lib.wgs_shard_particle_record_bytes() // 4
        self.hdr = self.lib.wgs_shard_buffer_header_bytes() // 4

    def step(self, num_substeps: int):
        """`wgs_sharded_step`: whole substeps incl. the neighbour exchange, asynchronous."""
        _ffi.check(self.lib, self.lib.wgs_sharded_step(self.pipeline._h, self._h, int(num_substeps)))

    def sync(self):
        _ffi.check(self.lib, self.lib.wgs_sync(self._h))

    def stats(self):
        s = self.T.Stats()
        _ffi.check(self.lib, self.lib.wgs_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in s._fields_}

    def num_particles(self) -> int:
        return self.stats()["num_particles"]

    def read_body_poses(self):
        """Like MpmData.read_body_poses: every rank integrates the same bodies, any rank can be asked."""
        n, D = self.n_colliders, self.dim
        poses, vels = (self.T.Pose * max(1, n))(), (self.T.Velocity * max(1, n))()
        coms = (C.c_float * (3 * max(1, n)))()
        _ffi.check(self.lib, self.lib.wgs_read_body_poses(self._h, poses, vels, coms, n))
        return [dict(rotation=np.array(list(poses[i].rotation)[:(2 if D == 2 else 4)], np.float64),
                     translation=np.array(list(poses[i].translation)[:D], np.float64),
                     linvel=np.array(list(vels[i].linear)[:D], np.float64),
                     angvel=np.array(list(vels[i].angular)[:(1 if D == 2 else 3)], np.float64)) for i in range(n)]

    def export(self):
        """(global ids, pos, vel, def_grad, affine, mass) of the particles this rank owns now (blocking)."""
        import torch
        buf = torch.zeros(self.hdr + self.capacity * self.part_rec, dtype=torch.float32, device=torch.device("cuda", self.pipeline.device))
        cnt = C.c_uint32(0)
        _ffi.check(self.lib, self.lib.wgs_shard_export(self._h, C.c_void_p(buf.data_ptr()), self.capacity, C.byref(cnt)))
        rec = buf[self.hdr: self.hdr + cnt.value * self.part_rec].cpu().numpy().reshape(cnt.value, self.part_rec)
        return unpack_records(rec, self.dim, self.uniform_material if self.dim == 3 else None)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wgs_data_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def native_lockstep(pipeline: MpmPipeline, shards: List[NativeShard], num_substeps: int):
    """`wgs_sharded_step_lockstep`: all slabs of one process on one device, in x order."""
    arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
    _ffi.check(pipeline.lib, pipeline.lib.wgs_sharded_step_lockstep(pipeline._h, arr, len(shards), int(num_substeps)))


def uniform_material_of(particles: ParticleSet):
    """(mass, init_volume, lambda, mu) if all particles of the set share them bitwise, else None."""
    if particles.n == 0:
        return None
    cols = (particles.mass, particles.init_volume, particles.lambda_, particles.mu)
    if all(bool(np.all(c.view(np.uint32) == c.view(np.uint32)[0])) for c in cols):
        return tuple(float(c[0]) for c in cols)
    return None


def unpack_records(rec: np.ndarray, dim: int, uniform_material=None):
    """Particle records (quad layout of csrc/layout.h) -> dict of arrays. In uniform-material mode (3D) the record's
    XM.w slot holds F[8] and the mass is the shared one."""
    ids = rec[:, -2].copy().view(np.uint32)     # [..quads.., pid, cdf epoch]
    if dim == 3:
        q = lambda k: rec[:, 4 * k:4 * k + 4]
        pos, mass = q(0)[:, :3], q(0)[:, 3]
        C_ = np.concatenate([q(1), q(2), q(3)[:, :1]], 1)
        vel = q(3)[:, 1:4]
        F = np.concatenate([q(4), q(5), q(6)[:, :1]], 1)
        if uniform_material is not None:
            F = np.concatenate([q(4), q(5), q(0)[:, 3:4]], 1)
            mass = np.full(len(rec), np.float32(uniform_material[0]), np.float32)
    else:
        q = lambda k: rec[:, 4 * k:4 * k + 4]
        pos, mass = q(0)[:, :2], q(0)[:, 2]
        C_, vel, F = q(1), q(2)[:, :2], q(3)
    return dict(ids=ids, pos=pos.copy(), vel=vel.copy(), def_grad=F.copy(), affine=C_.copy(), mass=mass.copy())


def split_scene(particles: ParticleSet, partition: SlabPartition, cell_width: float):
    """Assign every particle to the rank owning its associated block; returns per-rank (ParticleSet, global ids)."""
    bx = associated_block_x(particles.pos, cell_width, particles.dim)
    owner = partition.owner_of_blocks(bx)
    out = []
    for r in range(partition.world):
        idx = np.nonzero(owner == r)[0]
        sub = ParticleSet(**{k: (v[idx] if isinstance(v, np.ndarray) else v) for k, v in particles.__dict__.items()})
        out.append((sub, idx.astype(np.uint32)))
    return out
