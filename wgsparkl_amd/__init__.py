"""wgsparkl_amd — MI355X-native MLS-MPM substep behind wgsparkl's pipeline API.

Product code: `csrc/` (HIP kernels + C ABI, include/wgsparkl_hip.h) and the thin
host mirror of the reference's `pipeline` / `solver` / `models` modules.
The CPU oracle under /oracle is test infrastructure and is never imported here.
"""
from . import models, scenes, solver  # noqa: F401
from .models import (MODEL_COROTATED, MODEL_NEO_HOOKEAN, DruckerPrager,  # noqa: F401
                     ElasticCoefficients, ParticlePhase)
from .pipeline import KernelInvocationQueue, MpmData, MpmPipeline  # noqa: F401
from .solver import Collider, Particle, ParticleDynamics, ParticleSet, SimulationParams  # noqa: F401

__all__ = ["MpmPipeline", "MpmData", "KernelInvocationQueue", "Particle", "ParticleDynamics", "ParticleSet",
           "SimulationParams", "Collider", "ElasticCoefficients", "DruckerPrager", "ParticlePhase",
           "MODEL_COROTATED", "MODEL_NEO_HOOKEAN", "models", "solver", "scenes"]
