"""Material models — host-side mirror of the reference's `wgsparkl::models`.

Reference: src/models/mod.rs:52-75 (ElasticCoefficients, lame_lambda_mu),
src/models/drucker_prager.rs:6-53 (DruckerPrager, DruckerPragerPlasticState).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

F32 = np.float32

# Constitutive model ids (the reference selects at shader-compile time,
# src/solver/particle_update.wgsl:7-8; here it is a runtime per-MpmData choice).
MODEL_COROTATED = 0      # models/linear_elasticity.wgsl (reference default)
MODEL_NEO_HOOKEAN = 1    # models/neo_hookean_elasticity.wgsl


def lame_lambda_mu(young_modulus: float, poisson_ratio: float):
    """src/models/mod.rs:52-63, evaluated in f32 like the Rust code."""
    e = F32(young_modulus)
    nu = F32(poisson_ratio)
    one = F32(1.0)
    two = F32(2.0)
    lam = e * nu / ((one + nu) * (one - two * nu))
    mu = e / (two * (one + nu))
    return F32(lam), F32(mu)


@dataclass(frozen=True)
class ElasticCoefficients:
    """src/models/mod.rs:65-75"""
    lambda_: float
    mu: float

    @staticmethod
    def from_young_modulus(young_modulus: float, poisson_ratio: float) -> "ElasticCoefficients":
        lam, mu = lame_lambda_mu(young_modulus, poisson_ratio)
        return ElasticCoefficients(float(lam), float(mu))


@dataclass(frozen=True)
class DruckerPrager:
    """src/models/drucker_prager.rs:6-34"""
    h0: float
    h1: float
    h2: float
    h3: float
    lambda_: float
    mu: float

    @staticmethod
    def new(young_modulus: float, poisson_ratio: float) -> "DruckerPrager":
        if young_modulus > 0.0:
            lam, mu = lame_lambda_mu(young_modulus, poisson_ratio)
        else:
            lam, mu = F32(-1.0), F32(-1.0)
        rad = lambda deg: float(F32(deg) * F32(math.pi) / F32(180.0))
        return DruckerPrager(rad(35.0), rad(9.0), 0.2, rad(10.0), float(lam), float(mu))

    def as_array(self):
        return np.array([self.h0, self.h1, self.h2, self.h3, self.lambda_, self.mu], dtype=F32)


# src/models/drucker_prager.rs:44-53
DRUCKER_PRAGER_DEFAULT_STATE = np.array([1.0, 1.0, 0.0], dtype=F32)


@dataclass(frozen=True)
class ParticlePhase:
    """src/solver/particle_update.rs:35-40"""
    phase: float
    max_stretch: float
