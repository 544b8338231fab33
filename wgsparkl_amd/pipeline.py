"""`MpmPipeline` / `MpmData` — host-side mirror of `wgsparkl::pipeline`
(reference: src/pipeline.rs:24-281) over the C ABI of include/wgsparkl_hip.h.

The reference records ~25 dispatches into a `KernelInvocationQueue` once
(`queue_step`) and the caller replays that queue `num_substeps` times per frame
(src_testbed/step.rs:122-128). The same call shape is kept here:

    pipeline = MpmPipeline(device=0, dim=3)
    data = MpmData.new(pipeline, params, particles, colliders, cell_width, grid_capacity)
    queue = KernelInvocationQueue()
    pipeline.queue_step(data, queue, add_timestamps=False)
    for _ in range(num_substeps):
        queue.encode()          # enqueue one substep on the data's HIP stream
    data.sync()                 # device.poll(Maintain::Wait)
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from .models import MODEL_COROTATED
from .solver import Collider, ParticleSet, SimulationParams

F32 = np.float32


class KernelInvocationQueue:
    """Stand-in for wgcore's KernelInvocationQueue: holds the recorded step so it can
    be replayed with `encode()` (src_testbed/step.rs:126-128)."""

    def __init__(self):
        self._recorded = []

    def _push(self, fn):
        self._recorded.append(fn)

    def clear(self):
        self._recorded.clear()

    def encode(self, num_replays: int = 1):
        for fn in self._recorded:
            fn(num_replays)


class MpmPipeline:
    """src/pipeline.rs:24-39,176-193. `MpmPipeline.new(device)` fails when the HIP
    extension or a HIP device is missing (reference: ComposerError)."""

    def __init__(self, device: int = 0, dim: int = 3):
        self.dim = dim
        self.lib, self.T = _ffi.load(dim)
        h = C.c_void_p()
        _ffi.check(self.lib, self.lib.wgs_pipeline_create(int(device), C.byref(h)))
        self._h = h
        self.device = device

    new = classmethod(lambda cls, device=0, dim=3: cls(device, dim))

    def queue_step(self, data: "MpmData", queue: KernelInvocationQueue, add_timestamps: bool = False):
        """src/pipeline.rs:195-281 — records one substep; nothing runs until `queue.encode()`."""
        def run(n):
            _ffi.check(self.lib, self.lib.wgs_step(self._h, data._h, int(n), 1 if add_timestamps else 0))
        queue._push(run)

    def step(self, data: "MpmData", num_substeps: int = 1, timestamps: bool = False):
        """queue_step + encode x num_substeps in one call."""
        _ffi.check(self.lib, self.lib.wgs_step(self._h, data._h, int(num_substeps), 1 if timestamps else 0))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wgs_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _pack_particles(T, ps: ParticleSet):
    """ParticleSet -> contiguous array of wgs_particle (the repr(C) image of `Particle`)."""
    n, D = ps.n, ps.dim
    words = C.sizeof(T.Particle) // 4
    raw = np.zeros((n, words), np.float32)
    u = raw.view(np.uint32)
    off = lambda path: _offset_words(T.Particle, path)
    raw[:, off("position"):off("position") + D] = ps.pos
    o = off("dynamics.velocity"); raw[:, o:o + D] = ps.vel
    o = off("dynamics.def_grad"); raw[:, o:o + D * D] = ps.def_grad
    o = off("dynamics.affine"); raw[:, o:o + D * D] = ps.affine
    o = off("dynamics.cdf.normal"); raw[:, o:o + D] = ps.cdf_normal
    o = off("dynamics.cdf.rigid_vel"); raw[:, o:o + D] = ps.cdf_rigid_vel
    raw[:, off("dynamics.cdf.signed_distance")] = ps.cdf_dist
    u[:, off("dynamics.cdf.affinity")] = ps.cdf_affinity
    raw[:, off("dynamics.init_volume")] = ps.init_volume
    raw[:, off("dynamics.init_radius")] = ps.init_radius
    raw[:, off("dynamics.mass")] = ps.mass
    raw[:, off("model.lambda_")] = ps.lambda_
    raw[:, off("model.mu")] = ps.mu
    hp = ps.has_plasticity if ps.has_plasticity is not None else np.ones(n, bool)
    hph = ps.has_phase if ps.has_phase is not None else np.ones(n, bool)
    u[:, off("has_plasticity")] = hp.astype(np.uint32)
    o = off("plasticity"); raw[:, o:o + 6] = ps.dp
    u[:, off("has_phase")] = hph.astype(np.uint32)
    o = off("phase"); raw[:, o:o + 2] = ps.phase
    return raw


def _offset_words(struct, path: str) -> int:
    off = 0
    cur = struct
    for name in path.split("."):
        fld = getattr(cur, name)
        off += fld.offset
        cur = dict((f[0], f[1]) for f in cur._fields_)[name]
    return off // 4


def _unpack_particles(T, raw: np.ndarray, D: int, plastic: Optional[np.ndarray]) -> ParticleSet:
    u = raw.view(np.uint32)
    off = lambda path: _offset_words(T.Particle, path)
    sl = lambda path, k: raw[:, off(path):off(path) + k].copy()
    n = raw.shape[0]
    return ParticleSet(
        dim=D, pos=sl("position", D), vel=sl("dynamics.velocity", D), def_grad=sl("dynamics.def_grad", D * D),
        affine=sl("dynamics.affine", D * D), cdf_normal=sl("dynamics.cdf.normal", D),
        cdf_rigid_vel=sl("dynamics.cdf.rigid_vel", D), cdf_dist=raw[:, off("dynamics.cdf.signed_distance")].copy(),
        cdf_affinity=u[:, off("dynamics.cdf.affinity")].copy(), init_volume=raw[:, off("dynamics.init_volume")].copy(),
        init_radius=raw[:, off("dynamics.init_radius")].copy(), mass=raw[:, off("dynamics.mass")].copy(),
        lambda_=raw[:, off("model.lambda_")].copy(), mu=raw[:, off("model.mu")].copy(), dp=sl("plasticity", 6),
        dp_state=plastic if plastic is not None else np.tile(np.array([1, 1, 0], F32), (n, 1)),
        phase=sl("phase", 2), has_plasticity=u[:, off("has_plasticity")] != 0, has_phase=u[:, off("has_phase")] != 0)


def _is_dynamic(c: Collider) -> bool:
    return any(float(v) != 0.0 for v in list(c.inv_mass) + list(c.inv_inertia_local))


def _fill_collider(T, dst, c: Collider, D: int):
    dst.shape_type = int(c.shape_type)
    sh = list(c.shape) + [0.0] * (4 - len(c.shape))
    dst.shape = (C.c_float * 4)(*sh)
    if D == 2:
        ang = F32(c.rotation[0])
        dst.pose.rotation = (C.c_float * 4)(float(np.cos(ang)), float(np.sin(ang)), 0.0, 0.0)
    else:
        dst.pose.rotation = (C.c_float * 4)(*c.rotation)
    t = list(c.translation) + [0.0] * (3 - len(c.translation))
    dst.pose.translation = (C.c_float * 3)(*t)
    dst.pose.scale = c.scale
    dst.velocity.linear = (C.c_float * 3)(*(list(c.linvel)[:D] + [0.0] * (3 - D)))
    dst.velocity.angular = (C.c_float * 3)(*((list(c.angvel) + [0.0, 0.0, 0.0])[:3]))
    com = list(c.com) if c.com is not None else list(c.translation)
    dst.com = (C.c_float * 3)(*(com + [0.0] * (3 - len(com))))


class MpmData:
    """src/pipeline.rs:84-173. Owns every device buffer of one simulation."""

    def __init__(self, pipeline: MpmPipeline, params: SimulationParams, particles: ParticleSet,
                 colliders: Sequence[Collider], cell_width: float, grid_capacity: int,
                 model: int = MODEL_COROTATED):
        self.pipeline = pipeline
        self.lib, self.T = pipeline.lib, pipeline.T
        T, D = self.T, pipeline.dim
        if particles.dim != D:
            raise ValueError("particle dimension does not match the pipeline")
        self.dim = D
        self.n = particles.n
        sp = T.SimParams()
        sp.gravity = (C.c_float * D)(*params.gravity)
        sp.dt = params.dt
        raw = _pack_particles(T, particles)
        cols = (T.Collider * max(1, len(colliders)))()
        for i, c in enumerate(colliders):
            _fill_collider(T, cols[i], c, D)
        h = C.c_void_p()
        _ffi.check(self.lib, self.lib.wgs_data_create(
            pipeline._h, C.byref(sp), raw.ctypes.data_as(C.POINTER(T.Particle)), self.n,
            cols, len(colliders), float(cell_width), int(grid_capacity), C.byref(h)))
        self._h = h
        self.n_colliders = len(colliders)
        if model != MODEL_COROTATED:
            self.set_constitutive_model(model)
        if any(_is_dynamic(c) for c in colliders):
            self.set_body_mass_properties(colliders)
        # mesh colliders: sampled on the host like GpuRigidParticles::from_rapier (sampling step = cell width)
        from .sampling import build_rigid_particles
        rb = build_rigid_particles(colliders, D, float(cell_width))
        if rb is not None:
            self.set_rigid_particles(rb)

    @classmethod
    def new(cls, pipeline, params, particles, colliders, cell_width, grid_capacity, model=MODEL_COROTATED):
        """MpmData::new(device, params, &particles, &bodies, &colliders, cell_width, grid_capacity)
        (src/pipeline.rs:98-128). `particles` may be a ParticleSet or a list of Particle."""
        if not isinstance(particles, ParticleSet):
            particles = ParticleSet.from_particles(particles)
        return cls(pipeline, params, particles, colliders, cell_width, grid_capacity, model)

    # -- host -> device writes the caller performs every frame (src_testbed/step.rs:79-119, ui.rs:91-104)
    def set_constitutive_model(self, model: int):
        _ffi.check(self.lib, self.lib.wgs_set_constitutive_model(self._h, int(model)))

    def set_sim_params(self, params: SimulationParams):
        sp = self.T.SimParams()
        sp.gravity = (C.c_float * self.dim)(*params.gravity)
        sp.dt = params.dt
        _ffi.check(self.lib, self.lib.wgs_set_sim_params(self._h, C.byref(sp)))

    def set_colliders(self, colliders: Sequence[Collider]):
        """Refresh poses + velocities of the coupled colliders."""
        n = len(colliders)
        tmp = (self.T.Collider * max(1, n))()
        poses = (self.T.Pose * max(1, n))()
        vels = (self.T.Velocity * max(1, n))()
        coms = (C.c_float * (3 * max(1, n)))()
        for i, c in enumerate(colliders):
            _fill_collider(self.T, tmp[i], c, self.dim)
            poses[i] = tmp[i].pose
            vels[i] = tmp[i].velocity
            for k in range(3):
                coms[3 * i + k] = tmp[i].com[k]
        _ffi.check(self.lib, self.lib.wgs_set_collider_poses(self._h, poses, coms, n))
        _ffi.check(self.lib, self.lib.wgs_set_body_velocities(self._h, vels, n))
        self.set_body_mass_properties(colliders)

    def set_rigid_particles(self, rb: dict):
        """`rb` = sampling.build_rigid_particles(...): local sample points, (vertex ids, collider) per sample, local
        mesh vertices and their collider (GpuRigidParticles + the shape vertex buffers)."""
        pts = np.ascontiguousarray(rb["local_pts"], F32)
        ids = np.ascontiguousarray(rb["ids"], np.uint32)
        vtx = np.ascontiguousarray(rb["local_vtx"], F32)
        vcol = np.ascontiguousarray(rb["vtx_collider"], np.uint32)
        fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        _ffi.check(self.lib, self.lib.wgs_set_rigid_particles(
            self._h, pts.ctypes.data_as(fp), ids.ctypes.data_as(C.c_void_p), len(pts), vtx.ctypes.data_as(fp),
            vcol.ctypes.data_as(up), len(vtx)))

    def set_body_mass_properties(self, colliders: Sequence[Collider]):
        """GpuBodySet::from_rapier's local mass properties: all zero = kinematic, else two-way coupling."""
        n = len(colliders)
        mp = (self.T.MassProperties * max(1, n))()
        for i, c in enumerate(colliders):
            im = (list(c.inv_mass) + [0.0] * 3)[:3]
            ii = (list(c.inv_inertia_local) + [0.0] * 9)[:9]
            mp[i].inv_mass = (C.c_float * 3)(*im)
            mp[i].inv_inertia_local = (C.c_float * 9)(*ii)
        _ffi.check(self.lib, self.lib.wgs_set_body_mass_properties(self._h, mp, n))

    def read_body_poses(self):
        """poses_staging read-back (src_testbed/step.rs:129-132): one dict per collider with rotation
        (3D quaternion (i,j,k,w); 2D (cos, sin)), translation, linvel, angvel, com as float64 arrays."""
        n, D = self.n_colliders, self.dim
        poses = (self.T.Pose * max(1, n))()
        vels = (self.T.Velocity * max(1, n))()
        coms = (C.c_float * (3 * max(1, n)))()
        _ffi.check(self.lib, self.lib.wgs_read_body_poses(self._h, poses, vels, coms, n))
        out = []
        for i in range(n):
            out.append(dict(rotation=np.array(list(poses[i].rotation)[:(2 if D == 2 else 4)], np.float64),
                            translation=np.array(list(poses[i].translation)[:D], np.float64),
                            linvel=np.array(list(vels[i].linear)[:D], np.float64),
                            angvel=np.array(list(vels[i].angular)[:(1 if D == 2 else 3)], np.float64),
                            com=np.array([coms[3 * i + k] for k in range(D)], np.float64)))
        return out

    # -- device -> host
    def sync(self):
        _ffi.check(self.lib, self.lib.wgs_sync(self._h))

    def read_positions(self) -> np.ndarray:
        out = np.zeros((self.n, self.dim), F32)
        _ffi.check(self.lib, self.lib.wgs_read_positions(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def device_ptrs(self):
        """`wgs_get_device_ptrs`: where the current particle state lives on the device (position quads + particle ids in
        sorted order, and the HIP stream that produces them) — the optional interop view, for readers on the same device."""
        v = _ffi.DevicePtrs()
        _ffi.check(self.lib, self.lib.wgs_get_device_ptrs(self._h, C.byref(v)))
        return v

    def read_particles(self) -> ParticleSet:
        T = self.T
        words = C.sizeof(T.Particle) // 4
        raw = np.zeros((self.n, words), F32)
        plastic = np.zeros((self.n, 3), F32)
        _ffi.check(self.lib, self.lib.wgs_read_particles(
            self._h, raw.ctypes.data_as(C.POINTER(T.Particle)), plastic.ctypes.data_as(C.POINTER(T.PlasticState))))
        return _unpack_particles(T, raw, self.dim, plastic)

    def prep_vertex_buffer(self, mode: int = 0, base_color=None) -> np.ndarray:
        """src_testbed/prep_vertex_buffer{2,3}d.wgsl: n x 24 floats (deformation 3 x vec4, position vec4, base_color,
        color), particle i of the caller's order in row i. `base_color`: n x 4 (default opaque white)."""
        inst = np.zeros((self.n, 24), F32)
        inst[:, 16:20] = 1.0 if base_color is None else np.asarray(base_color, F32).reshape(self.n, 4)
        _ffi.check(self.lib, self.lib.wgs_prep_vertex_buffer(self._h, int(mode), inst.ctypes.data_as(C.c_void_p)))
        return inst

    def set_plastic_state(self, dp_state: np.ndarray):
        """Checkpoint restore of the Drucker-Prager plastic state (n x 3: plastic det, hardening, log_vol_gain)
        in the caller's particle order; together with `MpmData.new(read_particles())` it continues a run
        bit-exactly."""
        st = np.ascontiguousarray(dp_state, F32).reshape(self.n, 3)
        _ffi.check(self.lib, self.lib.wgs_set_plastic_state(self._h, st.ctypes.data_as(C.POINTER(self.T.PlasticState))))

    def read_grid(self):
        """(cells[int32 M x D], vel_mass[M x (D+1)], cdf_dist, cdf_aff, cdf_closest), lexicographically
        sorted by cell coordinate (physical node ids are not comparable between runs)."""
        T, D = self.T, self.dim
        cnt = C.c_size_t(0)
        _ffi.check(self.lib, self.lib.wgs_read_grid(self._h, None, 0, C.byref(cnt)))
        m = cnt.value
        words = C.sizeof(T.NodeRecord) // 4
        raw = np.zeros((max(m, 1), words), np.int32)
        if m:
            _ffi.check(self.lib, self.lib.wgs_read_grid(self._h, raw.ctypes.data_as(C.POINTER(T.NodeRecord)), m, C.byref(cnt)))
        raw = raw[:m]
        cells = raw[:, :D].copy()
        fl = raw.view(F32)
        vm = fl[:, D:2 * D + 1].copy()
        dist = fl[:, 2 * D + 1].copy()
        aff = raw.view(np.uint32)[:, 2 * D + 2].copy()
        closest = raw.view(np.uint32)[:, 2 * D + 3].copy()
        order = np.lexsort(cells.T[::-1]) if m else np.zeros(0, np.int64)
        return cells[order], vm[order], dist[order], aff[order], closest[order]

    def read_blocks(self, with_sorted_ids: bool = True):
        """(vid, first_particle, num_particles) sorted by vid, and sorted particle ids."""
        T, D = self.T, self.dim
        cnt = C.c_size_t(0)
        _ffi.check(self.lib, self.lib.wgs_read_blocks(self._h, None, 0, C.byref(cnt), None))
        m = cnt.value
        raw = np.zeros((max(m, 1), D + 2), np.int32)
        ids = np.zeros(max(self.n, 1), np.uint32)
        _ffi.check(self.lib, self.lib.wgs_read_blocks(
            self._h, raw.ctypes.data_as(C.POINTER(T.BlockRecord)), m, C.byref(cnt),
            ids.ctypes.data_as(C.POINTER(C.c_uint32)) if with_sorted_ids else None))
        raw = raw[:m]
        vid = raw[:, :D].copy()
        order = np.lexsort(vid.T[::-1]) if m else np.zeros(0, np.int64)
        u = raw.view(np.uint32)
        return vid[order], u[:, D][order].copy(), u[:, D + 1][order].copy(), ids[:self.n]

    def read_timings(self):
        ms = (C.c_float * _ffi.WGS_NUM_PASSES)()
        _ffi.check(self.lib, self.lib.wgs_read_timings(self._h, ms))
        return dict(zip(_ffi.PASS_NAMES, [float(x) for x in ms]))

    def stats(self):
        s = self.T.Stats()
        _ffi.check(self.lib, self.lib.wgs_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in s._fields_}

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wgs_data_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
