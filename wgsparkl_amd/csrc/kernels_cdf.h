// kernels_cdf.h — CPIC colour-distance-field passes for analytic colliders (k_block_prep).
//   node cdf     = solver/grid_update_cdf.wgsl:16-39 + collision/collide.wgsl:23-56
//   particle cdf = solver/g2p_cdf.wgsl:39-250
// The shape projections / pose maths are third party in the reference (wgparry
// Shape::projectPointOnBoundary, wgebra Sim2/Sim3 — not on disk); they are
// restated from parry's published algorithms, identically to the oracle.
#pragma once
#include "device_math.h"

namespace wgs {

template <int D> __device__ inline void quat_rotate(const float *q, const float *v, float *out) {
    float ux = q[0], uy = q[1], uz = q[2], w = q[3];
    float tx = 2.0f * (uy * v[2] - uz * v[1]);
    float ty = 2.0f * (uz * v[0] - ux * v[2]);
    float tz = 2.0f * (ux * v[1] - uy * v[0]);
    out[0] = v[0] + w * tx + (uy * tz - uz * ty);
    out[1] = v[1] + w * ty + (uz * tx - ux * tz);
    out[2] = v[2] + w * tz + (ux * ty - uy * tx);
}

template <int D> __device__ inline void pose_to_local(const ColliderDev &c, const float *pw, float *pl) {
    float dlt[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D; k++) dlt[k] = pw[k] - c.trans[k];
    if constexpr (D == 2) {
        float cs = c.rot[0], sn = c.rot[1];
        pl[0] = (cs * dlt[0] + sn * dlt[1]) / c.scale;
        pl[1] = (-sn * dlt[0] + cs * dlt[1]) / c.scale;
    } else {
        float qi[4] = {-c.rot[0], -c.rot[1], -c.rot[2], c.rot[3]};
        float t[3];
        quat_rotate<3>(qi, dlt, t);
#pragma unroll
        for (int k = 0; k < 3; k++) pl[k] = t[k] / c.scale;
    }
}

template <int D> __device__ inline void pose_to_world(const ColliderDev &c, const float *pl, float *pw) {
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D; k++) s[k] = pl[k] * c.scale;
    if constexpr (D == 2) {
        float cs = c.rot[0], sn = c.rot[1];
        pw[0] = cs * s[0] - sn * s[1] + c.trans[0];
        pw[1] = sn * s[0] + cs * s[1] + c.trans[1];
    } else {
        float t[3];
        quat_rotate<3>(c.rot, s, t);
#pragma unroll
        for (int k = 0; k < 3; k++) pw[k] = t[k] + c.trans[k];
    }
}

// Local-space projection on the shape BOUNDARY; returns is_inside.
template <int D> __device__ inline bool project_local_on_boundary(const ColliderDev &c, const float *pt, float *proj) {
    if (c.shape_type == 0u) {  // ball
        float r = c.shape[0], n2 = 0.f;
#pragma unroll
        for (int k = 0; k < D; k++) n2 += pt[k] * pt[k];
        float n = sqrtf(n2);
        if (n == 0.f) {
#pragma unroll
            for (int k = 0; k < D; k++) proj[k] = 0.f;
            proj[1] = r;
        } else {
#pragma unroll
            for (int k = 0; k < D; k++) proj[k] = pt[k] * (r / n);
        }
        return n2 <= r * r;
    }
    if (c.shape_type == 2u) {  // capsule along local y
        float hh = c.shape[0], r = c.shape[1];
        float seg[3] = {0.f, fmaxf(-hh, fminf(hh, pt[1])), 0.f};
        float dl[D], n2 = 0.f;
#pragma unroll
        for (int k = 0; k < D; k++) {
            dl[k] = pt[k] - seg[k];
            n2 += dl[k] * dl[k];
        }
        float n = sqrtf(n2);
        if (n == 0.f) {
#pragma unroll
            for (int k = 0; k < D; k++) proj[k] = seg[k];
            proj[0] += r;
        } else {
#pragma unroll
            for (int k = 0; k < D; k++) proj[k] = seg[k] + dl[k] * (r / n);
        }
        return n2 <= r * r;
    }
    // cuboid: parry Aabb::do_project_local_point(solid = false)
    float mins_pt[D], pt_maxs[D], shift[D];
    bool inside = true;
#pragma unroll
    for (int k = 0; k < D; k++) {
        float he = c.shape[k];
        mins_pt[k] = -he - pt[k];
        pt_maxs[k] = pt[k] - he;
        shift[k] = fmaxf(mins_pt[k], 0.f) - fmaxf(pt_maxs[k], 0.f);
        inside = inside && shift[k] == 0.f;
    }
    if (!inside) {
#pragma unroll
        for (int k = 0; k < D; k++) proj[k] = pt[k] + shift[k];
        return false;
    }
    float best = -3.402823466e+38f;
    bool is_mins = false;
    int best_id = 0;
#pragma unroll
    for (int k = 0; k < D; k++) {
        if (mins_pt[k] < pt_maxs[k]) {
            if (pt_maxs[k] > best) { best_id = k; is_mins = false; best = pt_maxs[k]; }
        } else if (mins_pt[k] > best) {
            best_id = k; is_mins = true; best = mins_pt[k];
        }
    }
#pragma unroll
    for (int k = 0; k < D; k++) proj[k] = pt[k] + (k == best_id ? (is_mins ? best : -best) : 0.f);
    return true;
}

// grid_update_cdf.wgsl:16-39 + collide.wgsl:23-56 for one node at world position pt: a pure function of
// the node position and the collider poses.
// `which`: bit i clear = collider i is known to be out of reach of this node (its vote would be empty): skipped.
template <int D> __device__ inline NodeCdf node_cdf_eval(const ColliderDev *colliders, uint32_t n_colliders, float h, const float *pt, uint32_t which) {
    const float cap = h * 1.5f;
    NodeCdf cdf = {1.0e10f, 0u, NONE, 0u};
    for (uint32_t i = 0; i < n_colliders && i < 16u; i++) {
        if (!((which >> i) & 1u)) continue;  // (wave-uniform in k_setup_scatter)
        const ColliderDev &c = colliders[i];
        if (c.shape_type >= 3u) continue;  // mesh shapes have no analytic projection (collide.wgsl:36-38)
        float pl[D], projl[D], proj[D];
        pose_to_local<D>(c, pt, pl);
        bool inside = project_local_on_boundary<D>(c, pl, projl);
        pose_to_world<D>(c, projl, proj);
        float n2 = 0.f;
        bool within = true;
#pragma unroll
        for (int k = 0; k < D; k++) {
            float dl = proj[k] - pt[k];
            n2 += dl * dl;
            within = within && (fabsf(dl) <= cap);
        }
        if (inside || within) {
            float dist = sqrtf(n2);
            if (dist < cdf.distance) cdf.closest_id = i;
            cdf.distance = fminf(cdf.distance, dist);
            cdf.affinities |= (inside ? 0x00010001u : 0x00000001u) << i;
        }
    }
    return cdf;
}

template <int D> __device__ inline NodeCdf node_cdf_eval(const Dev &d, const float *pt, uint32_t which = 0xffffu) {
    return node_cdf_eval<D>(d.colliders, d.n_colliders, d.h, pt, which);
}
// Solve the symmetric (N x N) system M x = r by LDL^T without pivoting (M is a
// weighted Gram matrix, positive definite whenever its determinant passes the
// reference's 1e-8 test). The reference uses wgebra Inv::inv3/inv4 (g2p_cdf.wgsl:236,242).
template <int N> __device__ inline void solve_spd(float *m, float *r) {
#pragma unroll
    for (int c = 0; c < N; c++) {
        float inv = 1.0f / m[c * N + c];
#pragma unroll
        for (int i = c + 1; i < N; i++) {
            float f = m[c * N + i] * inv;
#pragma unroll
            for (int j = c + 1; j < N; j++) m[j * N + i] -= f * m[j * N + c];
            r[i] -= f * r[c];
        }
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        float s = r[i];
#pragma unroll
        for (int j = i + 1; j < N; j++) s -= m[j * N + i] * r[j];
        r[i] = s / m[i * N + i];
    }
}

template <int N> __device__ inline float det_small(const float *m) {
    if constexpr (N == 3) {
        return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2]) + m[6] * (m[1] * m[5] - m[4] * m[2]);
    } else {
        float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[9] - m[8] * m[1], s2 = m[0] * m[13] - m[12] * m[1];
        float s3 = m[4] * m[9] - m[8] * m[5], s4 = m[4] * m[13] - m[12] * m[5], s5 = m[8] * m[13] - m[12] * m[9];
        float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[6] * m[15] - m[14] * m[7], c3 = m[6] * m[11] - m[10] * m[7];
        float c2 = m[2] * m[15] - m[14] * m[3], c1 = m[2] * m[11] - m[10] * m[3], c0 = m[2] * m[7] - m[6] * m[3];
        return s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    }
}

// Node cdfs of block b's (BW+2)^D tile -> LDS (s_cdf[TILE]); NT = threads of the calling workgroup (whole waves, all
// lanes active). The tile's nodes live in b and its 7 "+" neighbours: lanes 0..7 of every wave fetch the links once,
// then ALL node loads of the thread are issued before the first LDS store — two dependent round trips per tile instead
// of two per 64 / NT nodes.
// (`link`: lanes 0..7 of every wave hold nbr_plus[b * 8 + lane] — a caller that knows the block a round trip before it stages the tile asks for the links
// then: the list walk of the fused G2P, whose visit entry names the block while the sort entries of its particles are still on their way)
template <int D, int NT> __device__ __forceinline__ void stage_node_cdf_tile_links(const Dev &d, uint32_t link, NodeCdf *s_cdf, int tid) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE;
    constexpr int K = (TILE + NT - 1) / NT;
    NodeCdf c[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int n = tid + k * NT;
        const int tt[3] = {n % TW, (n / TW) % TW, D == 3 ? n / (TW * TW) : 0};
        const int o = (tt[0] >= BW ? 1 : 0) | (tt[1] >= BW ? 2 : 0) | (tt[2] >= BW ? 4 : 0);
        const int ln = (tt[0] & (BW - 1)) + ((tt[1] & (BW - 1)) << BS) + (D == 3 ? ((tt[2] & (BW - 1)) << (2 * BS)) : 0);
        const uint32_t nb = __shfl(link, o & 7);
        c[k] = NodeCdf{0.f, 0u, NONE, 0u};
        // (the lane's part of the address is pinned at the access: hipcc otherwise keeps node_cdf + 16 ln, one 64-bit pair per k, alive
        // for the whole kernel as loop invariants — and, in the kernels that have no registers to spare, in scratch memory)
        uint32_t lnp = (uint32_t)ln;
        asm volatile("" : "+v"(lnp));
        if (n < TILE && nb != NONE) c[k] = d.node_cdf[(size_t)nb * NPB + lnp];
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int n = tid + k * NT;
        if (n < TILE) s_cdf[n] = c[k];
    }
}

template <int D, int NT> __device__ __forceinline__ void stage_node_cdf_tile(const Dev &d, uint32_t b, NodeCdf *s_cdf, int tid) {
    const int lane = tid & 63;
    uint32_t link = NONE;
    if (lane < 8) link = d.nbr_plus[b * 8u + (uint32_t)lane];
    stage_node_cdf_tile_links<D, NT>(d, link, s_cdf, tid);
}

// g2p_cdf.wgsl:39-250 for one particle (`src` = its slot in the buffer): affinity / sign bits, distance and normal
// from the node cdfs of its block's tile (LDS image s_cdf, tile origin = block coordinates bc). Writes the
// particle's cdf quads and stamps them with the substep.
// AGENT: the quads are written through (agent scope) — a P2G launch whose pack waves copy a guest's record in the same launch
// What particle_cdf_update needs of the particle, fetched apart from it: a caller with several particles per thread requests them
// all before it computes the first (the prologue of the CPIC P2G: one pair of dependent round trips per batch instead of one per particle)
struct ParticleCdfIn {
    float4 xm;
    uint32_t prev_aff, stamp;
};
template <int D> __device__ __forceinline__ ParticleCdfIn particle_cdf_fetch(const Dev &d, const float *buf, uint32_t src) {
    using P = Pl<D>;
    ParticleCdfIn in;
    in.xm = ldq(buf, d.npad, P::XM, src);
    in.prev_aff = __float_as_uint(ldq(buf, d.npad, D == 3 ? (int)P::CDF1 : (int)P::CDF0, src).w);
    in.stamp = ldstamp<D>(buf, d.npad, src);
    return in;
}
template <int D, bool AGENT = false> __device__ inline void particle_cdf_update(const Dev &d, float *buf, uint32_t src, const ParticleCdfIn &pin, const NodeCdf *s_cdf,
                                                            const int *bc, uint32_t epoch) {
    constexpr int BW = Dim<D>::BW, TW = Dim<D>::TW;
    constexpr int N = D + 1;
    using P = Pl<D>;
    const uint32_t npad = d.npad;
    const float h = d.h, inv_h = d.inv_h;
    const bool any = true;
    float nrm[D], dist = 0.f;
    uint32_t aff = 0u;
#pragma unroll
    for (int k = 0; k < D; k++) nrm[k] = 0.f;
    if (any) {
        const float4 xm = pin.xm;
        // previous affinity (sign persistence, g2p_cdf.wgsl:183-190): only if computed last substep
        const uint32_t prev = pin.stamp == epoch - 1u ? pin.prev_aff : 0u;
        float x[D], ref[D], w[D][3];
        x[0] = xm.x; x[1] = xm.y;
        if constexpr (D == 3) x[2] = xm.z;
        int tbase = 0, stride = 1;
#pragma unroll
        for (int k = 0; k < D; k++) {
            int c = assoc_cell(x[k], h);
            ref[k] = (float)c * h - x[k];
            eval_all(-ref[k] * inv_h, w[k]);
            tbase += (c - bc[k] * BW) * stride;
            stride *= TW;
        }
        // pass 1 (g2p_cdf.wgsl:150-181): union of affinities, sign vote per collider
        constexpr int SZN = D == 3 ? 3 : 1;
#pragma unroll
        for (int sz = 0; sz < SZN; sz++)
#pragma unroll
            for (int sy = 0; sy < 3; sy++)
#pragma unroll
                for (int sx = 0; sx < 3; sx++) aff |= s_cdf[tbase + sx + TW * sy + (D == 3 ? TW * TW * sz : 0)].affinities & 0xffffu;
        // The vote of collider c is sum over the nodes of compatible * w * sign * distance (one accumulator per
        // collider in the reference). Only colliders some lane of the wave has an affinity with are visited — scenes
        // have one or two in reach, not 16 —, and a node that is not compatible with c adds an exact zero, so it is
        // skipped: the sums are bit-identical to the 16-accumulator form.
        uint32_t voters = 0u;
#pragma unroll
        for (int c = 0; c < 16; c++) voters |= __ballot((aff >> c) & 1u) != 0ull ? (1u << c) : 0u;
        for (uint32_t vm = voters; vm != 0u; vm &= vm - 1u) {  // wave-uniform
            const int c = __ffs((int)vm) - 1;
            float vote = 0.f;
#pragma unroll
            for (int sz = 0; sz < SZN; sz++)
#pragma unroll
                for (int sy = 0; sy < 3; sy++)
#pragma unroll
                    for (int sx = 0; sx < 3; sx++) {
                        const NodeCdf nc = s_cdf[tbase + sx + TW * sy + (D == 3 ? TW * TW * sz : 0)];
                        float wgt = w[0][sx] * w[1][sy];
                        if constexpr (D == 3) wgt *= w[2][sz];
                        if ((nc.affinities >> c) & 1u) vote += (((nc.affinities >> (16 + c)) & 1u) ? -wgt : wgt) * nc.distance;
                    }
            const uint32_t mask = 1u << (c + 16);
            if ((prev & (1u << c)) == 0u) aff |= vote < 0.f ? mask : 0u;
            else aff |= prev & mask;
        }
        // (colliders nobody voted on: vote = 0, so only a persisting sign bit of the previous substep can be set)
        aff |= prev & (prev << 16) & ~(voters << 16) & 0xffff0000u;
        // pass 2 (g2p_cdf.wgsl:192-231): weighted least squares for (grad d, d)
        float qtq[N * N], qtu[N];
#pragma unroll
        for (int k = 0; k < N * N; k++) qtq[k] = 0.f;
#pragma unroll
        for (int k = 0; k < N; k++) qtu[k] = 0.f;
#pragma unroll
        for (int sz = 0; sz < SZN; sz++)
#pragma unroll
            for (int sy = 0; sy < 3; sy++)
#pragma unroll
                for (int sx = 0; sx < 3; sx++) {
                    NodeCdf nc = s_cdf[tbase + sx + TW * sy + (D == 3 ? TW * TW * sz : 0)];
                    uint32_t combined = nc.affinities & aff & 0xffffu;
                    if (combined == 0u) continue;
                    uint32_t sdiff = ((nc.affinities >> 16) ^ (aff >> 16)) & combined;
                    float wgt = w[0][sx] * w[1][sy];
                    float pv[N];
                    pv[0] = ref[0] + (float)sx * h;
                    pv[1] = ref[1] + (float)sy * h;
                    if constexpr (D == 3) {
                        wgt *= w[2][sz];
                        pv[2] = ref[2] + (float)sz * h;
                    }
                    pv[D] = 1.f;
                    float dd = sdiff == 0u ? nc.distance : -nc.distance;
#pragma unroll
                    for (int c = 0; c < N; c++)
#pragma unroll
                        for (int r = 0; r <= c; r++) qtq[c * N + r] += (pv[r] * pv[c]) * wgt;  // (symmetric: mirrored below)
#pragma unroll
                    for (int r = 0; r < N; r++) qtu[r] += pv[r] * wgt * dd;
                }
#pragma unroll
        for (int c = 0; c < N; c++)
#pragma unroll
            for (int r = c + 1; r < N; r++) qtq[c * N + r] = qtq[r * N + c];  // same products, same order: bit-identical
        if (det_small<N>(qtq) > 1.0e-8f) {
            solve_spd<N>(qtq, qtu);
            float n2 = 0.f;
#pragma unroll
            for (int k = 0; k < D; k++) n2 += qtu[k] * qtu[k];
            float len = sqrtf(n2);
#pragma unroll
            for (int k = 0; k < D; k++) nrm[k] = (D == 2 && !(len > 1.0e-6f)) ? 0.f : qtu[k] / len;
            dist = qtu[D];
        } else {
            aff = 0u;  // default_cdf()
        }
    }
    const float4 q0 = D == 3 ? make_float4(nrm[0], nrm[1], nrm[2], dist) : make_float4(nrm[0], nrm[1], dist, __uint_as_float(aff));
    const float4 q1 = D == 3 ? make_float4(0.f, 0.f, 0.f, __uint_as_float(aff)) : make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (AGENT) {
        const __amdgpu_buffer_rsrc_t pr = particle_rsrc<D>(buf, npad);
        st_agent(pr, quad_off(npad, P::CDF0, src), q0);
        st_agent(pr, quad_off(npad, P::CDF1, src), q1);
        __hip_atomic_store(stamp_ptr<D>(buf, npad, src), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        stq(buf, npad, P::CDF0, src, q0);
        stq(buf, npad, P::CDF1, src, q1);
        ststamp<D>(buf, npad, src, epoch);
    }
}

// The three CDF passes of a substep in ONE launch, one single-wave workgroup per active block (every active
// block is in flight at once; the pass costs one dependent-load chain):
//   1. node cdf of the block's (BW+2)^D tile — its own nodes and the +1 rim. The rim belongs to the
//      neighbour blocks; being a pure function of position it is recomputed here (3.4x redundant, a few
//      hundred flops per node) instead of waiting for the neighbours behind a grid-wide dependency. The
//      block's own nodes are written to node_cdf (P2G, G2P and wgs_read_grid read them).
//   2. class of the block: does any node of the tile carry a collider affinity? Blocks that do not are
//      plain MLS-MPM this substep: every particle cdf is default_cdf() (g2p_cdf.wgsl:246-249), nothing reads
//      it, P2G / G2P take the plain path. The reference runs the full machinery for every block (PERF
//      note at grid_update_cdf.wgsl:34-36).
//   3. particle cdf (g2p_cdf.wgsl:39-250) of the particles of the blocks near a collider, from the tile in LDS.
constexpr int CDF_THREADS = 64;
template <int D> __global__ __launch_bounds__(CDF_THREADS) void k_cdf(Dev d, int side, uint32_t epoch) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    constexpr int N = D + 1;
    __shared__ NodeCdf s_cdf[TILE];
    __shared__ uint32_t s_any[2];
    float *buf = d.buf[side];
    const uint32_t npad = d.npad;
    const float h = d.h, inv_h = d.inv_h;
    const int tid = threadIdx.x;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    uint32_t it = 0;
    for (uint32_t a = blockIdx.x; a < B; a += gridDim.x, it ^= 1u) {
        const uint32_t b = d.active[a];
        const uint32_t cnt = d.block_count[b];
        const uint32_t start = d.block_start[b];
        int bc[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[b], bc);
        if (tid == 0) s_any[it] = 0u;
        __syncthreads();  // previous block's tile fully consumed
        uint32_t mine = 0u;
        for (int n = tid; n < TILE; n += CDF_THREADS) {
            int t[3] = {n % TW, (n / TW) % TW, D == 3 ? n / (TW * TW) : 0};
            int o = (t[0] >= BW ? 1 : 0) | (t[1] >= BW ? 2 : 0) | (t[2] >= BW ? 4 : 0);
            NodeCdf c = {0.f, 0u, NONE, 0u};
            if (d.nbr_plus[b * 8u + o] != NONE) {  // nodes of blocks that are not active do not exist
                float pt[D];
#pragma unroll
                for (int k = 0; k < D; k++) pt[k] = (float)(bc[k] * BW + t[k]) * h;
                c = node_cdf_eval<D>(d, pt);
                if (d.n_rigid != 0u) {  // mesh colliders: merge what k_p2g_cdf scattered (p2g_cdf.wgsl:103-117)
                    int ln = (t[0] & (BW - 1)) + ((t[1] & (BW - 1)) << BS) + (D == 3 ? ((t[2] & (BW - 1)) << (2 * BS)) : 0);
                    const size_t mn = (size_t)d.nbr_plus[b * 8u + o] * NPB + ln;
                    const unsigned long long mm = d.mesh_min[mn];
                    if (mm != ~0ull) {
                        const float md = __uint_as_float((uint32_t)(mm >> 32));
                        const uint32_t mid = (uint32_t)mm;
                        c.affinities |= d.mesh_aff[mn];
                        if (md < c.distance || (md == c.distance && mid < c.closest_id)) {
                            c.distance = md;
                            c.closest_id = mid;
                        }
                    }
                }
                if (o == 0) {
                    int ln = t[0] + (t[1] << BS) + (D == 3 ? (t[2] << (2 * BS)) : 0);
                    d.node_cdf[(size_t)b * NPB + ln] = c;
                }
            }
            s_cdf[n] = c;
            mine |= c.affinities;
        }
        if (mine != 0u) s_any[it] = 1u;  // benign race: every writer stores 1
        __syncthreads();
        const bool any = s_any[it] != 0u;
        if (tid == 0) {
            d.block_cpic[b] = any ? 1u : 0u;
            if (any && cnt > 0) d.cpic_list[(size_t)(b & 7u) * d.cap + atomicAdd(&d.counters[ctr_ncpic(b & 7u, epoch)], 1u)] = b;
        }
        if (any && cnt > 0 && tid < 64) append_visits(d, b, d.block_start[b], cnt, tid, epoch);  // (the first wave of the workgroup)
        if (any && cnt > 0 && tid == 0) d.pcdf_done[b] = 0u;
        // (3: the particle cdf of the listed blocks runs in the prologue of the CPIC P2G launch, three waves per block)
    }
}

}  // namespace wgs
