// kernels_rigid.h — rigid particles of mesh colliders (trimesh / heightfield / polyline): the CDF of shapes that
// have no analytic projection comes from samples of their surface.
//   k_rigid_transform = solver/rigid_particle_update.wgsl:26-49 (samples and mesh vertices to world space)
//   k_rigid_mark / k_rigid_touch = grid/sort.wgsl:38-86 (blocks that must exist because a sample reaches them)
//   k_p2g_cdf = solver/p2g_cdf.wgsl:52-190
// The reference bins the samples into per-node linked lists (sort.wgsl:139-161) and lets every node gather the 27
// cells below it. The result per node is an OR of bits and a minimum, so it is computed here as a SCATTER with
// integer atomics (atomicOr on the affinity / sign bits, one 64-bit atomicMin on (distance bits, collider id)):
// no sort of the rigid particles, same result whatever the order (equal distances: the smaller collider id).
#pragma once
#include "kernels_cdf.h"

namespace wgs {

template <int D> __global__ __launch_bounds__(256) void k_rigid_transform(Dev d) {
    const uint32_t total = d.n_rigid + d.n_rvtx;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const bool is_vtx = t >= d.n_rigid;
        const uint32_t i = is_vtx ? t - d.n_rigid : t;
        const uint32_t col = is_vtx ? d.rv_collider[i] : d.rp_ids[i].w;
        const float *src = (is_vtx ? d.rv_local : d.rp_local) + (size_t)i * D;
        float *dst = (is_vtx ? d.rv_world : d.rp_world) + (size_t)i * D;
        float l[3] = {0.f, 0.f, 0.f}, w[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D; k++) l[k] = src[k];
        pose_to_world<D>(d.colliders[min(col, 15u)], l, w);
#pragma unroll
        for (int k = 0; k < D; k++) dst[k] = w[k];
    }
}

template <int D> __device__ inline void rigid_cell(const Dev &d, uint32_t i, int *cell) {
    const float *p = d.rp_world + (size_t)i * D;
#pragma unroll
    for (int k = 0; k < D; k++) cell[k] = assoc_cell(p[k], d.h);
}

// sort.wgsl:55-86: the sample's own block is missing although one of its "+1" neighbours exists. Read-only
// with respect to the block set (all marks are taken before any block is added, as in the reference's two
// dispatches).
template <int D> __global__ __launch_bounds__(256) void k_rigid_mark(Dev d, uint32_t epoch) {
    constexpr int BS = Dim<D>::BSHIFT, NN = Dim<D>::NNBR;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < d.n_rigid; i += gridDim.x * 256) {
        int c[D], b[3] = {0, 0, 0};
        rigid_cell<D>(d, i, c);
#pragma unroll
        for (int k = 0; k < D; k++) b[k] = c[k] >> BS;
        bool own = false, other = false, in_range = true;
        uint32_t keys[NN], ids[NN];
        bool wanted[NN];
#pragma unroll
        for (int o = 0; o < NN; o++) {
            int nb[3] = {b[0] + (o & 1), b[1] + ((o >> 1) & 1), b[2] + ((o >> 2) & 1)};
            wanted[o] = block_in_key_range<D>(nb);
            in_range = in_range && wanted[o];
            keys[o] = wanted[o] ? pack_key<D>(nb) : 0u;
        }
        hmap_find_many<NN>(d, keys, wanted, epoch, ids);
#pragma unroll
        for (int o = 0; o < NN; o++) {
            const bool ex = ids[o] != NONE;
            if (o == 0) own = ex; else other = other || ex;
        }
        d.rp_needs[i] = (!own && other && in_range) ? 1u : 0u;
    }
}

// sort.wgsl:38-52: only the sample's own block is added
template <int D> __global__ __launch_bounds__(256) void k_rigid_touch(Dev d, uint32_t epoch) {
    constexpr int BS = Dim<D>::BSHIFT;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < d.n_rigid; i += gridDim.x * 256) {
        if (d.rp_needs[i] == 0u) continue;
        int c[D], b[3] = {0, 0, 0};
        rigid_cell<D>(d, i, c);
#pragma unroll
        for (int k = 0; k < D; k++) b[k] = c[k] >> BS;
        activate_block(d, pack_key<D>(b), epoch);
    }
}

// p2g_cdf.wgsl:123-190 for one (sample, node) pair: is the node's projection on the sample's primitive valid?
template <int D> __device__ inline bool project_on_primitive(const Dev &d, const uint4 ids, const float *cell, float &dist, bool &sign) {
    const float *va = d.rv_world + (size_t)ids.x * D, *vb = d.rv_world + (size_t)ids.y * D;
    if constexpr (D == 2) {
        // wgparry Segment::projectLocalPoint (third party): clamped orthogonal projection
        const float ab[2] = {vb[0] - va[0], vb[1] - va[1]}, ap[2] = {cell[0] - va[0], cell[1] - va[1]};
        const float den = ab[0] * ab[0] + ab[1] * ab[1];
        const float tt = den > 0.f ? (ap[0] * ab[0] + ap[1] * ab[1]) / den : 0.f;
        float proj[2];
        if (tt <= 0.f) { proj[0] = va[0]; proj[1] = va[1]; }
        else if (tt >= 1.f) { proj[0] = vb[0]; proj[1] = vb[1]; }
        else { proj[0] = va[0] + ab[0] * tt; proj[1] = va[1] + ab[1] * tt; }
        if ((proj[0] != va[0] || proj[1] != va[1]) && (proj[0] != vb[0] || proj[1] != vb[1])) {
            const float dp[2] = {cell[0] - proj[0], cell[1] - proj[1]};
            dist = sqrtf(dp[0] * dp[0] + dp[1] * dp[1]);
            sign = (dp[0] * (-ab[1]) + dp[1] * ab[0]) < 0.f;
            return true;
        }
        return false;
    } else {
        const float *vc = d.rv_world + (size_t)ids.z * D;
        float ap[3], bp[3], cp[3], ab[3], ac[3], bc[3], n[3], t[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            ap[k] = cell[k] - va[k]; bp[k] = cell[k] - vb[k]; cp[k] = cell[k] - vc[k];
            ab[k] = vb[k] - va[k]; ac[k] = vc[k] - va[k]; bc[k] = vc[k] - vb[k];
        }
        auto cross = [](float *o, const float *x, const float *y) {
            o[0] = x[1] * y[2] - x[2] * y[1]; o[1] = x[2] * y[0] - x[0] * y[2]; o[2] = x[0] * y[1] - x[1] * y[0];
        };
        auto dot = [](const float *x, const float *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
        cross(n, ab, ac);
        const float nlen = sqrtf(dot(n, n));
        if (nlen == 0.f) return false;
        cross(t, ab, n); const float d1 = dot(t, ap);
        cross(t, bc, n); const float d2 = dot(t, bp);
        cross(t, ac, n); const float d3 = dot(t, cp);
        if (d1 <= 0.f && d2 <= 0.f && d3 >= 0.f) {  // projection inside the face
            const float sd = dot(n, ap) / nlen;
            sign = sd < 0.f;
            dist = fabsf(sd);
            return true;
        }
        return false;
    }
}

// One thread per (sample, node) pair — 32 lanes per sample, 3^D of them busy —, so a sample's nodes are projected and
// merged side by side instead of one after the other (each costs a few dependent memory round trips).
template <int D> __global__ __launch_bounds__(256) void k_p2g_cdf(Dev d, uint32_t epoch) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, NBH = Dim<D>::NBH;
    const uint32_t total = d.n_rigid * 32u;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const uint32_t i = t >> 5;
        const int s = (int)(t & 31u);
        if (s >= NBH) continue;
        int c[D], b[3] = {0, 0, 0};
        rigid_cell<D>(d, i, c);
#pragma unroll
        for (int k = 0; k < D; k++) b[k] = c[k] >> BS;
        const uint4 ids = d.rp_ids[i];
        if (ids.w >= 16u) continue;
        int nc[3] = {c[0] + s % 3, c[1] + (s / 3) % 3, D == 3 ? c[D - 1] + s / 9 : 0};
        int nb[3] = {nc[0] >> BS, nc[1] >> BS, D == 3 ? nc[2] >> BS : 0};
        // a sample whose own block does not exist is in no node list (sort.wgsl:149-151): ignored; nodes of blocks that
        // do not exist do not exist
        uint32_t keys[2], blk[2];
        const bool wanted[2] = {block_in_key_range<D>(b), block_in_key_range<D>(nb)};
        keys[0] = wanted[0] ? pack_key<D>(b) : 0u;
        keys[1] = wanted[1] ? pack_key<D>(nb) : 0u;
        hmap_find_many<2>(d, keys, wanted, epoch, blk);
        if (blk[0] == NONE || blk[1] == NONE) continue;
        float cell[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D; k++) cell[k] = (float)nc[k] * d.h;
        float dist;
        bool sign;
        if (!project_on_primitive<D>(d, ids, cell, dist, sign)) continue;
        const uint32_t ln = (uint32_t)(nc[0] & (BW - 1)) + ((uint32_t)(nc[1] & (BW - 1)) << BS) +
                            (D == 3 ? ((uint32_t)(nc[2] & (BW - 1)) << (2 * BS)) : 0u);
        const size_t node = (size_t)blk[1] * NPB + ln;
        // Hundreds of samples reach the same node and device-scope atomics execute at the memory side, so look first
        // (coherent loads): the bits only ever get set and the minimum only ever shrinks within a substep, a stale
        // view costs a redundant atomic, never a wrong result.
        const uint32_t bits = (1u << ids.w) | ((sign ? 1u : 0u) << (ids.w + 16u));
        const unsigned long long cand = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned long long)ids.w;
        if ((__hip_atomic_load(&d.mesh_aff[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(&d.mesh_aff[node], bits);
        if (cand < __hip_atomic_load(&d.mesh_min[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&d.mesh_min[node], cand);
    }
}

}  // namespace wgs
