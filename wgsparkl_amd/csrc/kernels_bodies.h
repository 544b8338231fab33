// kernels_bodies.h — rigid bodies coupled to the particles (solver/rigid_impulses.wgsl).
//   k_bodies_refresh   = update_world_mass_properties (rigid_impulses.wgsl:138-149)
//   k_bodies_integrate = update (rigid_impulses.wgsl:95-136) followed by the refresh of the next substep
// Body::applyImpulse / integrateVelocity / updateMprops are third party in the reference (wgrapier, not on
// disk); restated from rapier's published algorithms (RigidBodyVelocity::integrate, world-space mass
// properties), identically to the oracle. At most 16 bodies (CPIC affinity mask, grid.wgsl:230-240):
// one 16-thread workgroup, like the reference.
#pragma once
#include "kernels_cdf.h"

namespace wgs {

// rigid_impulses.wgsl:50-58. WGSL's i32(f32) truncates toward zero and saturates (NaN -> 0).
__device__ inline int32_t flt2int(float f) {
    const float s = f * 1.0e5f;
    if (!(s == s)) return 0;
    if (s >= 2147483648.0f) return 2147483647;
    if (s <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)s;
}
__device__ inline float int2flt(int32_t i) { return (float)i / 1.0e5f; }

// World centre of mass and world inverse inertia from the current pose. `com_given` bit i: the caller wrote
// the WORLD centre of mass of body i (wgs_data_create, wgs_set_collider_poses with coms): derive the local
// one from it instead.
template <int D> __device__ inline void body_refresh(const Dev &d, uint32_t i, bool com_given) {
    ColliderDev &c = d.colliders[i];
    BodyDev &b = d.bodies[i];
    if (com_given) {
        float loc[3] = {0.f, 0.f, 0.f};
        pose_to_local<D>(c, c.com, loc);
        for (int k = 0; k < 3; k++) b.local_com[k] = loc[k];
    } else {
        float com[3] = {0.f, 0.f, 0.f};
        pose_to_world<D>(c, b.local_com, com);
        for (int k = 0; k < D; k++) c.com[k] = com[k];
    }
    if constexpr (D == 2) {
        b.inv_inertia_world[0] = b.inv_inertia_local[0];
    } else {
        float rm[9], tmp[9];
#pragma unroll
        for (int col = 0; col < 3; col++) {
            float e[3] = {col == 0 ? 1.f : 0.f, col == 1 ? 1.f : 0.f, col == 2 ? 1.f : 0.f}, o[3];
            quat_rotate<3>(c.rot, e, o);
            for (int r = 0; r < 3; r++) rm[col * 3 + r] = o[r];
        }
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int r = 0; r < 3; r++) {
                float s = 0.f;
                for (int k = 0; k < 3; k++) s += rm[k * 3 + r] * b.inv_inertia_local[cc * 3 + k];
                tmp[cc * 3 + r] = s;
            }
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int r = 0; r < 3; r++) {
                float s = 0.f;
                for (int k = 0; k < 3; k++) s += tmp[k * 3 + r] * rm[k * 3 + cc];
                b.inv_inertia_world[cc * 3 + r] = s;
            }
    }
}

template <int D> __global__ __launch_bounds__(16) void k_bodies_refresh(Dev d, uint32_t com_given) {
    const uint32_t i = threadIdx.x;
    if (i < d.n_colliders) body_refresh<D>(d, i, (com_given >> i) & 1u);
}

// integrate_bodies for body i (one thread per body). A function of its own: single-domain simulations without mesh colliders run
// it at the head of the NEXT substep's first sort launch (k_rebin / k_bin, workgroup 0) instead of as a 16-thread launch
// of its own at the tail of every substep — nothing in between reads a pose, a velocity or an impulse (capi.hip).
template <int D> __device__ inline void bodies_integrate_one(const Dev &d, uint32_t i) {
    constexpr int ANG = D == 3 ? 3 : 1;
    if (i >= d.n_colliders) return;
    ColliderDev &c = d.colliders[i];
    const BodyDev &b = d.bodies[i];
    int32_t *acc = d.impulses + i * 8;
    float lin[3] = {0.f, 0.f, 0.f}, ang[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < D; k++) lin[k] = int2flt(acc[k]);
    for (int k = 0; k < ANG; k++) ang[k] = int2flt(acc[D + k]);
    for (int k = 0; k < 8; k++) acc[k] = 0;  // reset for the next substep (rigid_impulses.wgsl:104-109)
    // Body::applyImpulse
    float nl[3] = {0.f, 0.f, 0.f}, na[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < D; k++) nl[k] = c.linvel[k] + b.inv_mass[k] * lin[k];
    if constexpr (D == 2) {
        na[0] = c.angvel[0] + b.inv_inertia_world[0] * ang[0];
    } else {
        for (int r = 0; r < 3; r++) {
            float s = 0.f;
            for (int k = 0; k < 3; k++) s += b.inv_inertia_world[k * 3 + r] * ang[k];
            na[r] = c.angvel[r] + s;
        }
    }
    float ln2 = 0.f, an2 = 0.f, il2 = 0.f, ia2 = 0.f;
    for (int k = 0; k < D; k++) { ln2 += nl[k] * nl[k]; il2 += lin[k] * lin[k]; }
    for (int k = 0; k < ANG; k++) { an2 += na[k] * na[k]; ia2 += ang[k] * ang[k]; }
    const float lnorm = sqrtf(ln2), anorm = sqrtf(an2);
    const float dt = d.sp->dt;
    const float lin_limit = 0.1f * d.h / dt, ang_limit = 1.0f;
    if (sqrtf(il2) != 0.f || sqrtf(ia2) != 0.f) {
        if (lnorm > lin_limit)
            for (int k = 0; k < D; k++) nl[k] = nl[k] * (lin_limit / lnorm);
        if (anorm > ang_limit)
            for (int k = 0; k < ANG; k++) na[k] = na[k] * (ang_limit / anorm);
    }
    // A body at rest that nothing pushes — no velocity, no impulse, no gravity on a dynamic axis — keeps the very bits of its
    // pose: the integration below is the identity in exact arithmetic, but in fp32 it renormalises the quaternion and
    // recomposes the translation about the centre of mass, which rewrites the pose of a ROTATED fixed collider by an ulp for
    // some substeps (a fifth of the rotations on the first, a few never settle). The node cdfs of blocks out of reach of the
    // colliders that move are cached (Dev::cdf_moving, kernels_sort.h): a fixed collider must be bit-static for that cache —
    // and for a bit-exact restart — whatever else moves in the scene.
    {
        bool rest = true;
        for (int k = 0; k < D; k++) rest = rest && nl[k] == 0.f && (b.inv_mass[k] == 0.f || d.sp->gravity[k] == 0.f);
        for (int k = 0; k < ANG; k++) rest = rest && na[k] == 0.f;
        if (rest) {
            // (the velocity all the same: an impulse that cancels a body's velocity exactly leaves nl = 0 with c.linvel != 0 — the reference
            // writes the zero, rigid_impulses.wgsl:128-131; storing 0 over 0 keeps a fixed collider bit-static)
            for (int k = 0; k < D; k++) c.linvel[k] = nl[k];
            for (int k = 0; k < ANG; k++) c.angvel[k] = na[k];
            return;
        }
    }
    // Body::integrateVelocity: rotate about the world centre of mass by exp(angvel dt), translate by linvel dt
    float comw[3] = {0.f, 0.f, 0.f}, arm[3] = {0.f, 0.f, 0.f}, rarm[3] = {0.f, 0.f, 0.f};
    pose_to_world<D>(c, b.local_com, comw);
    for (int k = 0; k < D; k++) arm[k] = c.trans[k] - comw[k];
    if constexpr (D == 2) {
        const float a = na[0] * dt;
        const float ca = cosf(a), sa = sinf(a);
        rarm[0] = ca * arm[0] - sa * arm[1];
        rarm[1] = sa * arm[0] + ca * arm[1];
        const float nc = ca * c.rot[0] - sa * c.rot[1], ns = sa * c.rot[0] + ca * c.rot[1];
        const float nn = sqrtf(nc * nc + ns * ns);
        c.rot[0] = nc / nn;
        c.rot[1] = ns / nn;
    } else {
        const float ax[3] = {na[0] * dt, na[1] * dt, na[2] * dt};
        const float angle = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        float dq[4] = {0.f, 0.f, 0.f, 1.f};
        if (angle != 0.f) {
            const float s = sinf(angle * 0.5f) / angle;
            dq[0] = ax[0] * s; dq[1] = ax[1] * s; dq[2] = ax[2] * s;
            dq[3] = cosf(angle * 0.5f);
        }
        quat_rotate<3>(dq, arm, rarm);
        const float q[4] = {c.rot[0], c.rot[1], c.rot[2], c.rot[3]};
        float nq[4];
        nq[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
        nq[1] = dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0];
        nq[2] = dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3];
        nq[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
        const float nn = sqrtf(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
        for (int k = 0; k < 4; k++) c.rot[k] = nq[k] / nn;
    }
    for (int k = 0; k < D; k++) c.trans[k] = comw[k] + rarm[k] + nl[k] * dt;
    // gravity on the dynamic axes only (rigid_impulses.wgsl:130-131)
    for (int k = 0; k < D; k++) c.linvel[k] = nl[k] + (b.inv_mass[k] != 0.f ? d.sp->gravity[k] * dt : 0.f);
    for (int k = 0; k < ANG; k++) c.angvel[k] = na[k];
    // update_world_mass_properties of the next substep (pipeline.rs:204-205)
    body_refresh<D>(d, i, false);
}

template <int D> __global__ __launch_bounds__(16) void k_bodies_integrate(Dev d) { bodies_integrate_one<D>(d, threadIdx.x); }
}  // namespace wgs
