// device_math.h — index math, B-spline weights, 2x2/3x3 linear algebra and the
// constitutive models as gfx950 device functions. Citations are relative to
// /root/reference/src/.
#pragma once
#include "layout.h"

namespace wgs {

// ---------------------------------------------------------------- index math
// grid/grid.wgsl:82-95 pack_key. Bit-exact.
template <int D> __host__ __device__ inline uint32_t pack_key(const int *b) {
    if constexpr (D == 2) {
        return ((uint32_t)(b[0] + 0x00007fff) & 0x0000ffffu) | (((uint32_t)(b[1] + 0x00007fff) & 0x0000ffffu) << 16);
    } else {
        return ((uint32_t)(b[0] + 0x000003ff) & 0x000007ffu) | (((uint32_t)(b[1] + 0x000001ff) & 0x000003ffu) << 11) |
               (((uint32_t)(b[2] + 0x000003ff) & 0x000007ffu) << 21);
    }
}

// Inverse of pack_key for blocks inside the representable range.
template <int D> __host__ __device__ inline void unpack_key(uint32_t key, int *b) {
    if constexpr (D == 2) {
        b[0] = (int)(key & 0xffffu) - 0x7fff;
        b[1] = (int)(key >> 16) - 0x7fff;
    } else {
        b[0] = (int)(key & 0x7ffu) - 0x3ff;
        b[1] = (int)((key >> 11) & 0x3ffu) - 0x1ff;
        b[2] = (int)(key >> 21) - 0x3ff;
    }
}

template <int D> __host__ __device__ inline bool block_in_key_range(const int *b) {
    if constexpr (D == 2) {
        // (the corner that packs to NONE, quirk B5 — and the block next to it along x, whose key is KEY_TOMB, the mark of an evicted table slot)
        return b[0] >= -0x7fff && b[0] <= 0x8000 && b[1] >= -0x7fff && b[1] <= 0x8000 && !(b[0] >= 0x7fff && b[1] == 0x8000);
    } else {
        bool in = b[0] >= -0x3ff && b[0] <= 0x400 && b[1] >= -0x1ff && b[1] <= 0x200 && b[2] >= -0x3ff && b[2] <= 0x400;
        return in && !(b[0] >= 0x3ff && b[1] == 0x200 && b[2] == 0x400);  // that corner packs to NONE (quirk B5); its x-neighbour to KEY_TOMB
    }
}

// grid/grid.wgsl:98-105 murmur3 scramble. Bit-exact.
__host__ __device__ inline uint32_t hash_key(uint32_t key) {
    key *= 0xcc9e2d51u;
    key = (key << 15) | (key >> 17);
    key *= 0x1b873593u;
    return key;
}

// solver/particle3d.wgsl:41-49, grid/grid.wgsl:284-292: assoc_cell = round(x / h) - 1 with WGSL
// round() = ties-to-even and a true fp32 division. v_rndne_f32 + IEEE division (no fast-math).
__device__ inline int assoc_cell(float x, float h) { return (int)(__builtin_rintf(x / h) - 1.0f); }
// Same value without the ~12-instruction IEEE division when h is a power of two (x * (1/h) is then exact).
__device__ inline int assoc_cell(float x, float h, float inv_h, bool h_pow2) {
    return h_pow2 ? (int)(__builtin_rintf(x * inv_h) - 1.0f) : assoc_cell(x, h);
}

// The hash map is PERSISTENT across substeps (the reference rebuilds it every substep,
// grid.wgsl:186-203 reset_hmap): blocks keep their slot and their physical id, and a
// per-block epoch stamp says whether the block is active in the current substep. Almost
// every touch is then a plain L2-served lookup + an idempotent plain store; device-scope
// atomics (memory-side on MI355X, ~1-2 us each and bandwidth-limited on a small table)
// are only issued for blocks never seen before. The table is cleared every REHASH_PERIOD
// substeps to drop blocks that stopped being active.
// (long: a rebuild substep costs ~2 ordinary ones; the host also triggers one as soon as three quarters of the ids are
// handed out, capi.hip maintain_grid, which is what bounds the table in practice)
constexpr uint32_t REHASH_PERIOD = 1024;
constexpr uint32_t ID_OVERFLOW = 0xfffffffeu;
// The key of a table slot whose block was EVICTED (kernels_sort.h regroup_block): look-ups walk past it; an insertion whose key is not
// in the table takes the first marked slot on its probe sequence (activate_block). No block packs to it (block_in_key_range).
constexpr uint32_t KEY_TOMB = 0xfffffffeu;
// substeps a block must have been inactive before launch 2 of the sort may evict it (its id goes on the free list, its table slot is marked);
// also the period of the substeps that evict (a power of two: kernels_sort.h k_regroup)
constexpr uint32_t EVICT_AGE = 8;
static_assert((EVICT_AGE & (EVICT_AGE - 1u)) == 0u, "EVICT_AGE is used as a mask");

// grid/grid.wgsl:167-184 find_block_header_id (active blocks only)
__device__ inline uint32_t hmap_find(const Dev &d, uint32_t key, uint32_t epoch) {
    uint32_t slot = hash_key(key) & d.hmask;
    for (uint32_t probe = 0; probe <= d.hmask; ++probe) {
        const uint32_t st = d.hkeys[slot];
        if (st == key) {
            const uint32_t id = d.hvals[slot];
            return (id < d.cap && d.block_stamp[id] == epoch) ? id : NONE;
        }
        if (st == NONE) return NONE;
        slot = (slot + 1u) & d.hmask;
    }
    return NONE;
}

// ... for a reader BEHIND the fused G2P of the substep: that launch stamps the blocks of the next substep while it bins its output
// (Dev::bin_next), so "active in this substep" is read from the note launch 2 of this substep's sort left (links_epoch)
__device__ inline uint32_t hmap_find_sorted(const Dev &d, uint32_t key, uint32_t epoch) {
    uint32_t slot = hash_key(key) & d.hmask;
    for (uint32_t probe = 0; probe <= d.hmask; ++probe) {
        const uint32_t st = d.hkeys[slot];
        if (st == key) {
            const uint32_t id = d.hvals[slot];
            return (id < d.cap && d.links_epoch[id] == epoch) ? id : NONE;
        }
        if (st == NONE) return NONE;
        slot = (slot + 1u) & d.hmask;
    }
    return NONE;
}

// the block's physical id whether it is active or not (NONE: not in the table)
// One visit-list entry per chunk of 64 sorted particles that holds particles of the listed block `id` (its run in the
// sorted order is [start, start + count), count > 0). Called by one whole wave. Consecutive groups of g2p_npass chunks
// (what one wave of the fused G2P advances in a row) go to the eight lists in turn: every XCD gets an even share of
// the visits, and a group stays together so that its later chunks find the block's node tile staged. One atomic per
// group, all groups of the block at once (one lane each). The order inside a list does not matter: every particle is
// advanced exactly once, by the visit of its chunk for its block.
__device__ inline void append_visits(const Dev &d, uint32_t id, uint32_t start, uint32_t count, int lane, uint32_t epoch) {
    const uint32_t np = d.g2p_npass;
    const uint32_t c0 = start >> 6, c1 = (start + count - 1u) >> 6, g0 = c0 / np, ng = c1 / np - g0 + 1u;
    for (uint32_t t = (uint32_t)lane; t < ng; t += 64u) {
        const uint32_t g = g0 + t, k = g & 7u;
        const uint32_t cs = max(c0, g * np), ce = min(c1, g * np + np - 1u), n = ce - cs + 1u;
        const uint32_t slot = atomicAdd(&d.counters[ctr_nvisit(k, epoch)], n);
        for (uint32_t e = 0; e < n; e++)
            if (slot + e < d.visit_cap) d.visit_list[(size_t)k * d.visit_cap + slot + e] = make_uint2(id, cs + e);
            else atomicOr(&d.counters[CTR_ERRORS], ERRBIT_OVERFLOW);  // (cannot happen within the capacity the lists are sized for; never silently)
    }
}

// Do the prologue workgroups of this P2G launch take the particle cdf of the listed blocks (kernels_transfer.h pcdf_waves)? The host sized
// them from the lists it saw last; the lists of THIS substep decide: no wave gets more than two visits, or the blocks' own workgroups
// do the work as before (a list that grew a hundredfold since the host's last look would otherwise be walked by a few waves, visit
// after visit, with every listed block waiting). Every workgroup of the launch reads the same eight counters: the same answer.
__device__ inline bool pcdf_waves_on(const Dev &d, uint32_t epoch, uint32_t waves_per_workgroup) {
    if (d.pcdf_waves == 0u) return false;
    uint32_t longest = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) longest = max(longest, d.counters[ctr_nvisit(k, epoch)]);
    return longest <= 2u * (d.pcdf_waves >> 3) * waves_per_workgroup;
}

__device__ inline uint32_t hmap_lookup(const Dev &d, uint32_t key) {
    uint32_t slot = hash_key(key) & d.hmask;
    for (uint32_t probe = 0; probe <= d.hmask; ++probe) {
        const uint32_t st = d.hkeys[slot];
        if (st == key) {
            const uint32_t id = d.hvals[slot];
            return id < d.cap ? id : NONE;
        }
        if (st == NONE) return NONE;
        slot = (slot + 1u) & d.hmask;
    }
    return NONE;
}

// K lookups at once, probing in lockstep: every round issues the loads of all unresolved keys together, then the values
// of the hits, then their stamps — a few dependent round trips for the lot instead of up to three per key, one key
// after the other.
template <int K> __device__ inline void hmap_find_many(const Dev &d, const uint32_t *keys, const bool *wanted, uint32_t epoch, uint32_t *out) {
    uint32_t slot[K], id[K];
    bool pend[K], hit[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        slot[k] = hash_key(keys[k]) & d.hmask;
        pend[k] = wanted[k];
        hit[k] = false;
    }
    for (uint32_t probe = 0; probe <= d.hmask; ++probe) {
        uint32_t st[K];
#pragma unroll
        for (int k = 0; k < K; k++) st[k] = pend[k] ? d.hkeys[slot[k]] : NONE;
        bool more = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (!pend[k]) continue;
            if (st[k] == keys[k]) { hit[k] = true; pend[k] = false; }
            else if (st[k] == NONE) pend[k] = false;
            else { slot[k] = (slot[k] + 1u) & d.hmask; more = true; }
        }
        if (__ballot(more) == 0ull) break;
    }
#pragma unroll
    for (int k = 0; k < K; k++) id[k] = hit[k] ? d.hvals[slot[k]] : NONE;
#pragma unroll
    for (int k = 0; k < K; k++) {
        out[k] = NONE;
        if (id[k] < d.cap && d.block_stamp[id[k]] == epoch) out[k] = id[k];
    }
}

// grid/grid.wgsl:121-164 insertion_index + :323-334 mark_block_as_active: make sure `key`
// is in the table, stamp its block active for `epoch` and return the block's physical id.
__device__ inline uint32_t activate_block(const Dev &d, uint32_t key, uint32_t epoch) {
    const uint32_t home = hash_key(key) & d.hmask;
    uint32_t slot = home;
    uint32_t first_mark = NONE;   // the first KEY_TOMB slot on the key's probe sequence: where the key goes if it is not in the table
    uint32_t result = NONE;
    bool done = false;
    bool coherent = false;   // after a lost claim: read the table past this CU's cache (a stale mark would be tried again and again)
    for (uint32_t probe = 0; probe <= d.hmask && !done; ++probe) {
        // plain load: a stale NONE only costs one extra CAS below
        uint32_t cur = coherent ? __hip_atomic_load(&d.hkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : d.hkeys[slot];
        bool won = false;
        if (cur == KEY_TOMB && first_mark == NONE) first_mark = slot;
        if (cur == NONE) {
            // the key is not in the table (its sequence ends here): it takes the first marked slot it passed, else this empty one. Every
            // wave that inserts the same key walks the same sequence and tries the same slot; if ANOTHER key took it meanwhile, start over.
            const uint32_t target = first_mark != NONE ? first_mark : slot;
            const uint32_t expect = first_mark != NONE ? KEY_TOMB : NONE;
            const uint32_t old = atomicCAS(&d.hkeys[target], expect, key);
            if (old == expect || old == key) {
                won = old == expect;
                slot = target;
                cur = key;
            } else {
                slot = home;
                first_mark = NONE;
                coherent = true;
                continue;
            }
        }
        // Winners first, in program order and WITHOUT leaving the divergent region: lanes of the
        // same wave that lost the race for this very slot wait below for hvals, and would spin
        // forever if the winner's branch were scheduled after their loop.
        if (won) {  // slot claimed: hand out a physical id (rare: new block)
            // an id from the free list of evicted blocks (pushed by launch 2 of the sort, never while anybody inserts), else a new one
            uint32_t id = NONE;
            if (d.free_ids != nullptr && __hip_atomic_load(&d.counters[CTR_NFREE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u < d.cap) {
                const uint32_t nf = atomicSub(&d.counters[CTR_NFREE], 1u);
                if (nf - 1u < d.cap) id = d.free_ids[nf - 1u];
                else atomicAdd(&d.counters[CTR_NFREE], 1u);   // (somebody else took the last one)
            }
            if (id == NONE) id = atomicAdd(&d.counters[CTR_NPHYS], 1u);
            atomicAdd(&d.counters[CTR_NINSERT], 1u);   // (what "no block was inserted since" is read from: ids are reused)
            if (first_mark != NONE && slot == first_mark) atomicSub(&d.counters[CTR_NTOMB], 1u);   // (a marked slot is a key's again)
            if (id < d.cap) {
                d.block_key[id] = key;
                if (d.block_slot != nullptr) d.block_slot[id] = slot;
            } else {
                atomicOr(&d.counters[CTR_ERRORS], ERRBIT_OVERFLOW);
                id = ID_OVERFLOW;
            }
            __hip_atomic_store(&d.hvals[slot], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cur == key) {
            uint32_t id = d.hvals[slot];
            // a thread of ANOTHER wave may still be between its CAS and its hvals store: bounded wait
            // (the value of a marked slot is NONE, like an empty slot's: the eviction cleared it)
            for (int spin = 0; id == NONE && spin < (1 << 16); spin++) {
                __builtin_amdgcn_s_sleep(1);
                id = __hip_atomic_load(&d.hvals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (id < d.cap) {
                d.block_stamp[id] = epoch;  // every writer stores the same value
                result = id;
            } else if (id == NONE) {
                // the bounded wait expired with the insert still in flight: the caller's particles would vanish from
                // the sort without a trace. Never seen; reported like every other loss path.
                atomicOr(&d.counters[CTR_ERRORS], ERRBIT_OVERFLOW);
            }
            done = true;
        }
        slot = (slot + 1u) & d.hmask;
    }
    if (!done) atomicOr(&d.counters[CTR_ERRORS], ERRBIT_OVERFLOW);
    return result;
}

// ------------------------------------------------------- quadratic B-spline
// grid/kernel.wgsl:60-66 eval_all
__device__ inline void eval_all(float x, float *w) {
    w[0] = 0.5f * (1.5f - x) * (1.5f - x);
    w[1] = 0.75f - (x - 1.0f) * (x - 1.0f);
    w[2] = 0.5f * (x - 0.5f) * (x - 0.5f);
}

// ------------------------------------------------------------ small matrices
// Column-major like WGSL / nalgebra: element (r, c) at [c * D + r].
template <int D> __device__ inline void mat_mul(const float *a, const float *b, float *out) {
#pragma unroll
    for (int c = 0; c < D; c++)
#pragma unroll
        for (int r = 0; r < D; r++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < D; k++) s += a[k * D + r] * b[c * D + k];
            out[c * D + r] = s;
        }
}

// out = a * b^T
template <int D> __device__ inline void mat_mul_bt(const float *a, const float *b, float *out) {
#pragma unroll
    for (int c = 0; c < D; c++)
#pragma unroll
        for (int r = 0; r < D; r++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < D; k++) s += a[k * D + r] * b[k * D + c];
            out[c * D + r] = s;
        }
}

template <int D> __device__ inline float mat_det(const float *m) {
    if constexpr (D == 2) {
        return m[0] * m[3] - m[2] * m[1];
    } else {
        return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2]) + m[6] * (m[1] * m[5] - m[4] * m[2]);
    }
}

// One-sided (Hestenes) Jacobi SVD, fp32: F = U diag(S) V^T, written for registers
// (fully unrolled, no indexing by run-time values). Replaces wgebra::svd2/svd3
// (third party, not on disk; call sites linear_elasticity.wgsl:15,29,
// drucker_prager.wgsl:80,139, particle_update.wgsl:103,109).
// Convention: U, V proper rotations; for det F < 0 the sign sits on the singular
// value of smallest magnitude (every consumer is invariant to ordering).
template <int D> struct Svd {
    float u[D * D], s[D], v[D * D];  // v = V (not transposed), column-major
};

// One rotation of the column pair (p, q). `live` = false leaves the lane untouched (it has converged). Returns whether
// the rotation was more than round-off (|cos| of the column pair above 3e-7, ~5 ulp).
template <int D> __device__ inline bool jacobi_rotate(float *a, float *v, int p, int q, bool live) {
    float alpha = 0.f, beta = 0.f, gamma = 0.f;
#pragma unroll
    for (int r = 0; r < D; r++) {
        alpha += a[p * D + r] * a[p * D + r];
        beta += a[q * D + r] * a[q * D + r];
        gamma += a[p * D + r] * a[q * D + r];
    }
    // rotation angle zeroing the (p,q) inner product; nothing to do when already orthogonal
    // The rotation angle only steers the iteration (any angle close to the exact one converges the same way, and the
    // singular values and vectors are read off the rotated columns afterwards), so it is computed with the hardware's
    // 1-ulp reciprocal / square root / reciprocal square root instead of the correctly rounded sequences: a rotation
    // drops from ~120 to ~70 VALU instructions, and the Drucker-Prager G2P is bound by exactly this dependent chain.
    // What must stay tight is c^2 + s^2 = 1: c is one rsq of 1 + t^2 and s = c t, so the pair is off unit length by
    // ~1 ulp per rotation, the same order as the rounding of c and s themselves.
#ifndef WGS_SVD_IEEE
    float zeta = (beta - alpha) * __builtin_amdgcn_rcpf(2.0f * gamma);
    float t = copysignf(__builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.0f))), zeta);
#else
    float zeta = (beta - alpha) / (2.0f * gamma);
    float t = copysignf(1.0f, zeta) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
#endif
    const bool skip = !live || !(fabsf(gamma) > 1.0e-30f) || !(gamma * gamma > 1.0e-15f * alpha * beta) || !(t == t);
    if (!skip) {
#ifndef WGS_SVD_IEEE
        float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.0f));
#else
        float c = 1.0f / sqrtf(1.0f + t * t);
#endif
        float s = c * t;
#pragma unroll
        for (int r = 0; r < D; r++) {
            float ap = a[p * D + r], aq = a[q * D + r];
            a[p * D + r] = c * ap - s * aq;
            a[q * D + r] = s * ap + c * aq;
            float vp = v[p * D + r], vq = v[q * D + r];
            v[p * D + r] = c * vp - s * vq;
            v[q * D + r] = s * vp + c * vq;
        }
    }
    return !skip && gamma * gamma > 1.0e-13f * alpha * beta;
}

template <int D> __device__ inline void svd(const float *F, Svd<D> &out) {
    float a[D * D];
#pragma unroll
    for (int i = 0; i < D * D; i++) {
        a[i] = F[i];
        out.v[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < D; i++) out.v[i * D + i] = 1.f;
    if constexpr (D == 2) {
        jacobi_rotate<2>(a, out.v, 0, 1, true);  // one rotation is exact in 2D
        jacobi_rotate<2>(a, out.v, 0, 1, true);  // second pass mops up fp32 residue
    } else {
        // Cyclic sweeps, at most 5 (fp32 converges quadratically). A lane whose sweep rotated by no more than
        // round-off has converged and is frozen — what is left is below 1e-13 in the singular values and 3e-7 in
        // the vectors, later sweeps would chase fp32 noise —, so its result depends on its own matrix only; the
        // wave leaves the loop when all its lanes are frozen: 3 sweeps for the deformations met in practice.
        bool live = true;
#pragma unroll 1
        for (int sweep = 0; sweep < 5; sweep++) {
            bool rotated = jacobi_rotate<3>(a, out.v, 0, 1, live);
            rotated = jacobi_rotate<3>(a, out.v, 0, 2, live) || rotated;
            rotated = jacobi_rotate<3>(a, out.v, 1, 2, live) || rotated;
            live = live && rotated;
            if (__ballot(live) == 0ull) break;
        }
    }
    float smax = 0.f;
#pragma unroll
    for (int c = 0; c < D; c++) {
        float n2 = 0.f;
#pragma unroll
        for (int r = 0; r < D; r++) n2 += a[c * D + r] * a[c * D + r];
        out.s[c] = sqrtf(n2);
        smax = fmaxf(smax, out.s[c]);
    }
    bool ok[D];
#pragma unroll
    for (int c = 0; c < D; c++) {
        ok[c] = out.s[c] > 1.0e-30f && out.s[c] > 1.0e-7f * smax;
        float inv = ok[c] ? 1.0f / out.s[c] : 0.f;
#pragma unroll
        for (int r = 0; r < D; r++) out.u[c * D + r] = a[c * D + r] * inv;
        if (!ok[c]) out.s[c] = 0.f;
    }
    // Rebuild columns of U that belong to vanishing singular values.
    if constexpr (D == 2) {
        if (!ok[0] && !ok[1]) {
            out.u[0] = 1.f; out.u[1] = 0.f; out.u[2] = 0.f; out.u[3] = 1.f;
        } else if (!ok[0]) {
            out.u[0] = out.u[3]; out.u[1] = -out.u[2];
        } else if (!ok[1]) {
            out.u[2] = -out.u[1]; out.u[3] = out.u[0];
        }
    } else {
        int nbad = (!ok[0]) + (!ok[1]) + (!ok[2]);
        if (nbad == 3) {
#pragma unroll
            for (int i = 0; i < 9; i++) out.u[i] = (i % 4 == 0) ? 1.f : 0.f;
        } else if (nbad == 2) {
            float e[3];
#pragma unroll
            for (int r = 0; r < 3; r++) e[r] = ok[0] ? out.u[r] : (ok[1] ? out.u[3 + r] : out.u[6 + r]);
            // axis least aligned with e
            float t[3] = {0.f, 0.f, 0.f};
            float a0 = fabsf(e[0]), a1 = fabsf(e[1]), a2 = fabsf(e[2]);
            if (a0 <= a1 && a0 <= a2) t[0] = 1.f; else if (a1 <= a2) t[1] = 1.f; else t[2] = 1.f;
            float dt = e[0] * t[0] + e[1] * t[1] + e[2] * t[2];
            float b1[3] = {t[0] - dt * e[0], t[1] - dt * e[1], t[2] - dt * e[2]};
            float n1 = 1.0f / sqrtf(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
            b1[0] *= n1; b1[1] *= n1; b1[2] *= n1;
            float b2[3] = {e[1] * b1[2] - e[2] * b1[1], e[2] * b1[0] - e[0] * b1[2], e[0] * b1[1] - e[1] * b1[0]};
            // columns (g+1)%3 and (g+2)%3 receive b1, b2 (g = the good column). Written as selects on STATIC elements: with the
            // assignments inside `if (ok[0]) .. else if (ok[1]) ..` hipcc indexed out.u by a run-time offset, which put the array into
            // scratch memory — ten scratch stores and loads per particle of every Drucker-Prager kernel, for a path no healthy run takes.
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const float c0 = out.u[r], c1 = out.u[3 + r], c2 = out.u[6 + r];
                out.u[r] = ok[0] ? c0 : (ok[1] ? b2[r] : b1[r]);
                out.u[3 + r] = ok[0] ? b1[r] : (ok[1] ? c1 : b2[r]);
                out.u[6 + r] = ok[0] ? b2[r] : (ok[1] ? b1[r] : c2);
            }
        } else if (nbad == 1) {
            // missing column = cross product of the next two (cyclic)
            float x[3], y[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                x[r] = !ok[0] ? out.u[3 + r] : (!ok[1] ? out.u[6 + r] : out.u[r]);
                y[r] = !ok[0] ? out.u[6 + r] : (!ok[1] ? out.u[r] : out.u[3 + r]);
            }
            float z[3] = {x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]};
#pragma unroll
            for (int r = 0; r < 3; r++) {
                out.u[r] = !ok[0] ? z[r] : out.u[r];
                out.u[3 + r] = (ok[0] && !ok[1]) ? z[r] : out.u[3 + r];
                out.u[6 + r] = (ok[0] && ok[1]) ? z[r] : out.u[6 + r];
            }
        }
    }
    // Proper rotations: flip the column of the smallest singular value.
    int kmin = 0;
#pragma unroll
    for (int c = 1; c < D; c++) kmin = out.s[c] < out.s[kmin] ? c : kmin;
    bool flip_v = mat_det<D>(out.v) < 0.f;
    bool flip_u = mat_det<D>(out.u) < 0.f;
#pragma unroll
    for (int c = 0; c < D; c++) {
        bool m = c == kmin;
        float sv = (m && flip_v) ? -1.f : 1.f, su = (m && flip_u) ? -1.f : 1.f;
#pragma unroll
        for (int r = 0; r < D; r++) {
            out.v[c * D + r] *= sv;
            out.u[c * D + r] *= su;
        }
        out.s[c] *= sv * su;
    }
}

// U * diag(s) * V^T
template <int D> __device__ inline void svd_recompose(const Svd<D> &d, const float *s, float *out) {
    float us[D * D];
#pragma unroll
    for (int c = 0; c < D; c++)
#pragma unroll
        for (int r = 0; r < D; r++) us[c * D + r] = d.u[c * D + r] * s[c];
    mat_mul_bt<D>(us, d.v, out);
}

// --------------------------------------------------------------- models
// models/neo_hookean_elasticity.wgsl:12-25
template <int D> __device__ inline void kirchoff_neo_hookean(float lambda, float mu, const float *F, float *tau) {
    float j = fmaxf(mat_det<D>(F), 1.0e-10f);
    float diag = lambda * logf(j) - mu;
    float fft[D * D];
    mat_mul_bt<D>(F, F, fft);
#pragma unroll
    for (int i = 0; i < D * D; i++) tau[i] = mu * fft[i];
#pragma unroll
    for (int k = 0; k < D; k++) tau[k * D + k] += diag;
}

// models/linear_elasticity.wgsl:14-41 (corotated), given the SVD of F.
template <int D> __device__ inline void kirchoff_corotated(float lambda, float mu, const float *F, const Svd<D> &d, float *tau) {
    float j = d.s[0];
#pragma unroll
    for (int k = 1; k < D; k++) j = j * d.s[k];
    float sm1[D];
#pragma unroll
    for (int k = 0; k < D; k++) sm1[k] = d.s[k] - 1.0f;
    float diag = lambda * (j - 1.0f) * j;
    float rec[D * D], prod[D * D];
    svd_recompose<D>(d, sm1, rec);  // F - R
    mat_mul_bt<D>(rec, F, prod);
#pragma unroll
    for (int i = 0; i < D * D; i++) tau[i] = prod[i] * (2.0f * mu);
#pragma unroll
    for (int k = 0; k < D; k++) tau[k * D + k] += diag;
}

// models/drucker_prager.wgsl:25-29
__device__ inline float dp_alpha(const float *dp, float q) {
    float angle = dp[0] + (dp[1] * q - dp[3]) * expf(-dp[2] * q);
    float s_angle = sinf(angle);
    return sqrtf(2.0f / 3.0f) * (2.0f * s_angle) / (3.0f - s_angle);
}

// models/drucker_prager.wgsl:42-101 (2D) / :112-158 (3D). Updates `d.s`, `state`
// and F in place when the projection is valid; returns whether it changed anything.
template <int D> __device__ inline bool drucker_prager_project(const float *dp, float *state, float *F, Svd<D> &d) {
    const float dd = (float)D;
    float alpha = dp_alpha(dp, state[1]);
    float strain[D], dev[D], nsv[D];
    float trace = 0.f;
#pragma unroll
    for (int k = 0; k < D; k++) {
        strain[k] = logf(d.s[k]) + state[2] / dd;
        trace += strain[k];
    }
    bool all_zero = true;
#pragma unroll
    for (int k = 0; k < D; k++) {
        dev[k] = strain[k] - trace / dd;
        all_zero = all_zero && (dev[k] == 0.f);
    }
    float hardening;
    if (trace > 0.f || all_zero) {
        float n2 = 0.f;
#pragma unroll
        for (int k = 0; k < D; k++) {
            nsv[k] = 1.f;
            n2 += strain[k] * strain[k];
        }
        hardening = sqrtf(n2);
    } else {
        float n2 = 0.f;
#pragma unroll
        for (int k = 0; k < D; k++) n2 += dev[k] * dev[k];
        float dev_norm = sqrtf(n2);
        float gamma = dev_norm + (dd * dp[4] + 2.0f * dp[5]) / (2.0f * dp[5]) * trace * alpha;
        if (gamma <= 0.f) return false;
#pragma unroll
        for (int k = 0; k < D; k++) nsv[k] = expf(strain[k] - dev[k] * (gamma / dev_norm));
        hardening = gamma;
    }
    float prev_det = d.s[0], new_det = nsv[0];
#pragma unroll
    for (int k = 1; k < D; k++) {
        prev_det = prev_det * d.s[k];
        new_det = new_det * nsv[k];
    }
    state[0] = state[0] * prev_det / new_det;
    state[2] = state[2] + logf(prev_det) - logf(new_det);
    state[1] = state[1] + hardening;
#pragma unroll
    for (int k = 0; k < D; k++) d.s[k] = nsv[k];
    svd_recompose<D>(d, d.s, F);
    return true;
}

// grid/grid.wgsl:390-404 project_velocity (friction 20)
template <int D> __device__ inline void project_velocity(const float *vel, const float *n, float *out) {
    float nv = 0.f;
#pragma unroll
    for (int k = 0; k < D; k++) nv += vel[k] * n[k];
    if (nv < 0.f) {
        float t[D], l2 = 0.f;
#pragma unroll
        for (int k = 0; k < D; k++) {
            t[k] = vel[k] - n[k] * nv;
            l2 += t[k] * t[k];
        }
        float len = sqrtf(l2);
        float scale = fmaxf(0.f, len + 20.0f * nv);
#pragma unroll
        for (int k = 0; k < D; k++) out[k] = (len > 1.0e-8f ? t[k] / len : 0.f) * scale;
    } else {
#pragma unroll
        for (int k = 0; k < D; k++) out[k] = vel[k];
    }
}

// grid/grid.wgsl:250-255
__device__ inline bool affinities_are_compatible(uint32_t a1, uint32_t a2) {
    uint32_t common = a1 & a2 & 0x0000ffffu;
    return ((a1 >> 16) & common) == ((a2 >> 16) & common);
}

// wgrapier Body::velocity_at_point (call sites p2g.wgsl:208, g2p.wgsl:191,224)
// (C: ColliderDev, or the copy of its motion a wave keeps in LDS — ColliderMotion, layout.h)
template <int D, class C> __device__ inline void velocity_at_point(const C &c, const float *pt, float *out) {
    if constexpr (D == 2) {
        float dx = pt[0] - c.com[0], dy = pt[1] - c.com[1];
        out[0] = c.linvel[0] - c.angvel[0] * dy;
        out[1] = c.linvel[1] + c.angvel[0] * dx;
    } else {
        float dx = pt[0] - c.com[0], dy = pt[1] - c.com[1], dz = pt[2] - c.com[2];
        out[0] = c.linvel[0] + (c.angvel[1] * dz - c.angvel[2] * dy);
        out[1] = c.linvel[1] + (c.angvel[2] * dx - c.angvel[0] * dz);
        out[2] = c.linvel[2] + (c.angvel[0] * dy - c.angvel[1] * dx);
    }
}

}  // namespace wgs
