"""Particle / parameter types — host-side mirror of `wgsparkl::solver`.

Reference: src/solver/particle3d.rs:16-60, src/solver/particle2d.rs:14-58,
src/solver/params.rs:6-16, src/models/mod.rs:20-49 (defaults applied when
`plasticity` / `phase` are None).

`ParticleSet` is the vectorised (structure-of-arrays, numpy f32) form of
`&[Particle]`; `Particle` is kept for API parity with the reference's scenes.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from .models import (DRUCKER_PRAGER_DEFAULT_STATE, DruckerPrager,
                     ElasticCoefficients, ParticlePhase)

F32 = np.float32


@dataclass
class SimulationParams:
    """src/solver/params.rs:6-16"""
    gravity: Sequence[float]
    dt: float


@dataclass
class ParticleDynamics:
    """src/solver/particle3d.rs:16-42"""
    velocity: np.ndarray
    def_grad: np.ndarray   # column-major D*D
    affine: np.ndarray
    init_volume: float
    init_radius: float
    mass: float

    @staticmethod
    def with_density(radius: float, density: float, dim: int = 3) -> "ParticleDynamics":
        r = F32(radius)
        vol = F32(1.0)
        for _ in range(dim):       # (radius * 2).powi(dim)
            vol = vol * (r * F32(2.0))
        return ParticleDynamics(
            velocity=np.zeros(dim, F32),
            def_grad=np.eye(dim, dtype=F32).reshape(-1),
            affine=np.zeros(dim * dim, F32),
            init_volume=float(vol),
            init_radius=float(r),
            mass=float(vol * F32(density)),
        )


@dataclass
class Particle:
    """src/solver/particle3d.rs:53-60"""
    position: np.ndarray
    dynamics: ParticleDynamics
    model: ElasticCoefficients
    plasticity: Optional[DruckerPrager] = None
    phase: Optional[ParticlePhase] = None


@dataclass
class ParticleSet:
    """SoA image of a `[Particle]` slice with the reference's defaults applied."""
    dim: int
    pos: np.ndarray            # [n, D]
    vel: np.ndarray            # [n, D]
    def_grad: np.ndarray       # [n, D*D] column-major
    affine: np.ndarray         # [n, D*D]
    cdf_normal: np.ndarray     # [n, D]
    cdf_rigid_vel: np.ndarray  # [n, D]
    cdf_dist: np.ndarray       # [n]
    cdf_affinity: np.ndarray   # [n] u32
    init_volume: np.ndarray    # [n]
    init_radius: np.ndarray    # [n]
    mass: np.ndarray           # [n]
    lambda_: np.ndarray        # [n]
    mu: np.ndarray             # [n]
    dp: np.ndarray             # [n, 6]
    dp_state: np.ndarray       # [n, 3]
    phase: np.ndarray          # [n, 2]
    has_plasticity: np.ndarray = field(default=None)  # [n] bool (bookkeeping only)
    has_phase: np.ndarray = field(default=None)

    @property
    def n(self) -> int:
        return int(self.pos.shape[0])

    def copy(self) -> "ParticleSet":
        kw = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.__dict__.items()}
        return ParticleSet(**kw)

    @staticmethod
    def uniform(pos: np.ndarray, radius: float, density: float,
                model: ElasticCoefficients,
                plasticity: Optional[DruckerPrager] = None,
                phase: Optional[ParticlePhase] = None,
                vel: Optional[np.ndarray] = None) -> "ParticleSet":
        """All particles share one material (what every reference scene does)."""
        pos = np.ascontiguousarray(pos, dtype=F32)
        n, dim = pos.shape
        dyn = ParticleDynamics.with_density(radius, density, dim)
        # src/models/mod.rs:24: None -> DruckerPrager::new(-1, -1)  (lambda = mu = -1, quirk B1)
        dp = (plasticity or DruckerPrager.new(-1.0, -1.0)).as_array()
        # src/models/mod.rs:33-36: None -> {phase: 0, max_stretch: -1}
        ph = phase or ParticlePhase(0.0, -1.0)
        eye = np.eye(dim, dtype=F32).reshape(-1)
        return ParticleSet(
            dim=dim,
            pos=pos,
            vel=np.zeros((n, dim), F32) if vel is None else np.ascontiguousarray(vel, F32),
            def_grad=np.tile(eye, (n, 1)),
            affine=np.zeros((n, dim * dim), F32),
            cdf_normal=np.zeros((n, dim), F32),
            cdf_rigid_vel=np.zeros((n, dim), F32),
            cdf_dist=np.zeros(n, F32),
            cdf_affinity=np.zeros(n, np.uint32),
            init_volume=np.full(n, dyn.init_volume, F32),
            init_radius=np.full(n, dyn.init_radius, F32),
            mass=np.full(n, dyn.mass, F32),
            lambda_=np.full(n, model.lambda_, F32),
            mu=np.full(n, model.mu, F32),
            dp=np.tile(dp, (n, 1)),
            dp_state=np.tile(DRUCKER_PRAGER_DEFAULT_STATE, (n, 1)),
            phase=np.tile(np.array([ph.phase, ph.max_stretch], F32), (n, 1)),
            has_plasticity=np.full(n, plasticity is not None),
            has_phase=np.full(n, phase is not None),
        )

    @staticmethod
    def from_particles(particles: Sequence[Particle]) -> "ParticleSet":
        """GpuParticles::from_particles + GpuModels::from_particles
        (src/solver/particle3d.rs:192-210, src/models/mod.rs:20-49)."""
        n = len(particles)
        dim = int(len(particles[0].position)) if n else 3
        ps = ParticleSet.uniform(np.zeros((n, dim), F32), 1.0, 1.0, ElasticCoefficients(0.0, 0.0))
        default_dp = DruckerPrager.new(-1.0, -1.0).as_array()
        for i, p in enumerate(particles):
            ps.pos[i] = p.position
            ps.vel[i] = p.dynamics.velocity
            ps.def_grad[i] = p.dynamics.def_grad
            ps.affine[i] = p.dynamics.affine
            ps.init_volume[i] = p.dynamics.init_volume
            ps.init_radius[i] = p.dynamics.init_radius
            ps.mass[i] = p.dynamics.mass
            ps.lambda_[i] = p.model.lambda_
            ps.mu[i] = p.model.mu
            ps.dp[i] = p.plasticity.as_array() if p.plasticity else default_dp
            ps.has_plasticity[i] = p.plasticity is not None
            ps.phase[i] = (p.phase.phase, p.phase.max_stretch) if p.phase else (0.0, -1.0)
            ps.has_phase[i] = p.phase is not None
        return ps


# Collider shape ids shared with include/wgsparkl_hip.h (WGS_SHAPE_*)
SHAPE_BALL = 0
SHAPE_CUBOID = 1
SHAPE_CAPSULE = 2
SHAPE_MESH = 3       # trimesh / heightfield (3D), polyline (2D): coupled through rigid particles (sampling.py)


@dataclass
class Collider:
    """One coupled collider, flattened from rapier's (RigidBody, Collider) pair the
    way `MpmData::new` couples them (src/pipeline.rs:107-117) and the testbed
    refreshes them each frame (src_testbed/step.rs:79-119)."""
    shape_type: int
    shape: Sequence[float]                 # ball: (r,), cuboid: half extents, capsule: (half_height, r)
    translation: Sequence[float]
    rotation: Sequence[float] = (0.0, 0.0, 0.0, 1.0)  # 3D quaternion (i,j,k,w); 2D: (angle,)
    scale: float = 1.0
    linvel: Sequence[float] = (0.0, 0.0, 0.0)
    angvel: Sequence[float] = (0.0, 0.0, 0.0)         # 2D: (w,)
    com: Optional[Sequence[float]] = None             # world-space; defaults to translation
    # Mass properties of the parent body (wgrapier local_mprops, rigid_impulses.wgsl:81-84). All zero =
    # kinematic / fixed: the body follows its velocity only. Non-zero = dynamic: two-way coupling.
    inv_mass: Sequence[float] = (0.0, 0.0, 0.0)
    inv_inertia_local: Sequence[float] = (0.0,) * 9   # 3D: column-major, body frame; 2D: (1/I,)
    # mesh colliders (shape_type == SHAPE_MESH): local-frame vertices [nv, D] and triangles [nt, 3] / segments [ns, 2]
    vertices: Optional[np.ndarray] = None
    indices: Optional[np.ndarray] = None

    def with_density(self, density: float, dim: int) -> "Collider":
        """Dynamic body of uniform density (rapier's MassProperties::from_ball / from_cuboid / from_capsule
        formulas): fills inv_mass and inv_inertia_local; the centre of mass is the shape origin."""
        import math
        s = [float(x) * self.scale for x in self.shape]
        if self.shape_type == SHAPE_BALL:
            r = s[0]
            if dim == 2:
                m = density * math.pi * r * r
                inertia = [m * r * r / 2.0]
            else:
                m = density * 4.0 / 3.0 * math.pi * r ** 3
                inertia = [2.0 / 5.0 * m * r * r] * 3
        elif self.shape_type == SHAPE_CUBOID:
            if dim == 2:
                m = density * 4.0 * s[0] * s[1]
                inertia = [m * (s[0] ** 2 + s[1] ** 2) / 3.0]
            else:
                m = density * 8.0 * s[0] * s[1] * s[2]
                inertia = [m * (s[1] ** 2 + s[2] ** 2) / 3.0, m * (s[0] ** 2 + s[2] ** 2) / 3.0,
                           m * (s[0] ** 2 + s[1] ** 2) / 3.0]
        else:
            raise NotImplementedError("with_density: ball and cuboid only")
        inv = [1.0 / x for x in inertia]
        ii = [inv[0]] + [0.0] * 8 if dim == 2 else [inv[0], 0, 0, 0, inv[1], 0, 0, 0, inv[2]]
        import dataclasses
        return dataclasses.replace(self, inv_mass=(1.0 / m,) * 3, inv_inertia_local=tuple(ii))

    @staticmethod
    def cuboid(half_extents, translation, **kw) -> "Collider":
        return Collider(SHAPE_CUBOID, tuple(half_extents), tuple(translation), **kw)

    @staticmethod
    def trimesh(vertices, indices, translation, **kw) -> "Collider":
        """TriMesh collider (src/solver/particle3d.rs:118-127)."""
        return Collider(SHAPE_MESH, (0.0,), tuple(translation), vertices=np.asarray(vertices, np.float32),
                        indices=np.asarray(indices, np.uint32), **kw)

    @staticmethod
    def polyline(vertices, indices, translation, **kw) -> "Collider":
        """2D Polyline collider (src/solver/particle2d.rs:92-102)."""
        return Collider(SHAPE_MESH, (0.0,), tuple(translation), vertices=np.asarray(vertices, np.float32),
                        indices=np.asarray(indices, np.uint32), **kw)

    @staticmethod
    def heightfield(heights, scale, translation, **kw) -> "Collider":
        """HeightField collider, converted to a trimesh like the reference does (src/solver/particle3d.rs:128-138)."""
        from .sampling import heightfield_to_trimesh
        v, i = heightfield_to_trimesh(heights, scale)
        return Collider.trimesh(v, i, translation, **kw)

    @staticmethod
    def ball(radius, translation, **kw) -> "Collider":
        return Collider(SHAPE_BALL, (radius,), tuple(translation), **kw)
