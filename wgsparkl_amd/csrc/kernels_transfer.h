// kernels_transfer.h — P2G, grid update and the fused G2P + particle update.
//
// P2G (reference: solver/p2g.wgsl:69-236, a per-node gather over linked lists of
// 8 blocks with ~3.4x redundant particle fetches) is re-expressed as a per-block
// SCATTER with no atomics and a fixed summation order:
//   1. a workgroup owns one block; its particles (contiguous, cell-sorted) are
//      staged once into LDS as 64-byte records (coalesced HBM reads);
//   2. thread (cell, sy, sz) walks the particles of its cell in order and keeps
//      the three sx contributions in registers;
//   3. the 64 x 27 partials are reduced per tile node in a fixed order and the
//      (BW+2)^D tile is written with plain coalesced stores to the block's slab.
// The grid update then gathers, for every node, the (at most 2^D) slabs that
// cover it, in a fixed order, and applies grid_update.wgsl:55-64 in the same
// pass (the reference keeps P2G and grid update apart only because of WebGPU's
// binding limit, p2g.wgsl:129-133).
#pragma once
#include "device_math.h"

namespace wgs {

// ------------------------------------------------------------------- P2G
template <int D> struct P2GCfg;
template <> struct P2GCfg<3> {
    static constexpr int TPC = 9;            // threads per cell: (sy, sz)
    static constexpr int THREADS = 64 * 9;   // 576 = 9 waves
    static constexpr int REC4 = 4;           // float4 per staged particle: x,m | mv,c0 | c1..c4 | c5..c8
    static constexpr int CHUNK = 512;        // particles staged per pass (32 KiB)
};
template <> struct P2GCfg<2> {
    static constexpr int TPC = 3;            // (sy)
    static constexpr int THREADS = 64 * 3;   // 192 = 3 waves
    static constexpr int REC4 = 3;           // x,y,m,0 | mvx,mvy,c0,c1 | c2,c3,0,0
    static constexpr int CHUNK = 512;
};

template <int D, bool CPIC>
__global__ __launch_bounds__(P2GCfg<D>::THREADS) void k_p2g(Dev d, int side) {
    using Cfg = P2GCfg<D>;
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NBH = Dim<D>::NBH;
    constexpr int DD = D * D;
    __shared__ float4 s_rec[Cfg::CHUNK * Cfg::REC4];
    __shared__ uint32_t s_aff[CPIC ? Cfg::CHUNK : 1];
    __shared__ float4 s_partial[NPB * NBH];

    const float *in = d.buf[side];
    const uint32_t npad = d.npad;
    const float h = d.h, inv_h = d.inv_h;
    const int tid = threadIdx.x;
    const int cell = tid / Cfg::TPC, sub = tid % Cfg::TPC;
    const int sy = sub % 3, sz = sub / 3;
    int lc[3];
    lc[0] = cell & (BW - 1);
    lc[1] = (cell >> BS) & (BW - 1);
    lc[2] = D == 3 ? (cell >> (2 * BS)) : 0;

    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    for (uint32_t b = blockIdx.x; b < B; b += gridDim.x) {
        const uint32_t cnt = d.block_count[b];
        if (cnt == 0) continue;  // no particles: its slab is never read
        const uint32_t start = d.block_start[b];
        int bc[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[b], bc);
        // world position of this thread's cell (= associated grid node of its particles)
        float cpos[D];
#pragma unroll
        for (int k = 0; k < D; k++) cpos[k] = (float)(bc[k] * BW + lc[k]) * h;
        const uint32_t cs = d.cell_start[b * NPB + cell];
        const uint32_t ce = cell < NPB - 1 ? d.cell_start[b * NPB + cell + 1] : start + cnt;
        // CPIC: affinities of this thread's three target nodes (p2g.wgsl:100-103)
        uint32_t naff[3] = {0u, 0u, 0u};
        if constexpr (CPIC) {
#pragma unroll
            for (int s = 0; s < 3; s++) {
                int t[3] = {lc[0] + s, lc[1] + sy, lc[2] + sz};
                int o = (t[0] >= BW ? 1 : 0) | (t[1] >= BW ? 2 : 0) | (t[2] >= BW ? 4 : 0);
                int ln = (t[0] & (BW - 1)) + ((t[1] & (BW - 1)) << BS) + (D == 3 ? ((t[2] & (BW - 1)) << (2 * BS)) : 0);
                uint32_t nb = d.nbr_plus[b * 8u + o];
                if (nb != NONE) naff[s] = d.node_cdf[(size_t)nb * NPB + ln].affinities;
            }
        }

        float acc[3][D + 1];
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
            for (int k = 0; k <= D; k++) acc[s][k] = 0.f;

        for (uint32_t base = start; base < start + cnt; base += Cfg::CHUNK) {
            const uint32_t m = min((uint32_t)Cfg::CHUNK, start + cnt - base);
            __syncthreads();
            for (uint32_t j = tid; j < m; j += Cfg::THREADS) {
                const uint32_t src = d.perm[base + j];
                if constexpr (D == 3) {
                    const float4 xm = ldq(in, npad, Pl<3>::XM, src);
                    const float4 c0 = ldq(in, npad, Pl<3>::CV0, src);
                    const float4 c1 = ldq(in, npad, Pl<3>::CV1, src);
                    const float4 c2 = ldq(in, npad, Pl<3>::CV2, src);
                    s_rec[j * 4 + 0] = xm;
                    s_rec[j * 4 + 1] = make_float4(c2.y * xm.w, c2.z * xm.w, c2.w * xm.w, c0.x);  // m v, c0
                    s_rec[j * 4 + 2] = make_float4(c0.y, c0.z, c0.w, c1.x);
                    s_rec[j * 4 + 3] = make_float4(c1.y, c1.z, c1.w, c2.x);
                } else {
                    const float4 xm = ldq(in, npad, Pl<2>::XM, src);   // x, y, m, V0
                    const float4 c0 = ldq(in, npad, Pl<2>::CV0, src);
                    const float4 vl = ldq(in, npad, Pl<2>::CV2, src);  // vx, vy, lambda, mu
                    s_rec[j * 3 + 0] = make_float4(xm.x, xm.y, xm.z, 0.f);
                    s_rec[j * 3 + 1] = make_float4(vl.x * xm.z, vl.y * xm.z, c0.x, c0.y);
                    s_rec[j * 3 + 2] = make_float4(c0.z, c0.w, 0.f, 0.f);
                }
                if constexpr (CPIC) {
                    const float4 cd = ldq(in, npad, D == 3 ? (int)Pl<D>::CDF1 : (int)Pl<D>::CDF0, src);
                    s_aff[j] = __float_as_uint(cd.w);
                }
            }
            __syncthreads();
            const uint32_t lo = max(cs, base), hi = min(ce, base + m);
            for (uint32_t j = lo; j < hi; j++) {
                const uint32_t r = j - base;
                float x[D], mv[D], c[DD], mass;
                if constexpr (D == 3) {
                    float4 r0 = s_rec[r * 4 + 0], r1 = s_rec[r * 4 + 1], r2 = s_rec[r * 4 + 2], r3 = s_rec[r * 4 + 3];
                    x[0] = r0.x; x[1] = r0.y; x[2] = r0.z; mass = r0.w;
                    mv[0] = r1.x; mv[1] = r1.y; mv[2] = r1.z;
                    c[0] = r1.w; c[1] = r2.x; c[2] = r2.y; c[3] = r2.z; c[4] = r2.w;
                    c[5] = r3.x; c[6] = r3.y; c[7] = r3.z; c[8] = r3.w;
                } else {
                    float4 r0 = s_rec[r * 3 + 0], r1 = s_rec[r * 3 + 1], r2 = s_rec[r * 3 + 2];
                    x[0] = r0.x; x[1] = r0.y; mass = r0.z;
                    mv[0] = r1.x; mv[1] = r1.y;
                    c[0] = r1.z; c[1] = r1.w; c[2] = r2.x; c[3] = r2.y;
                }
                // p2g.wgsl:176-198: ref = assoc_node - x ; w = eval_all(-ref / h) ; dpt = ref + shift * h
                float ref[D], wx[3], wy[3], wz[3];
#pragma unroll
                for (int k = 0; k < D; k++) ref[k] = cpos[k] - x[k];
                eval_all(-ref[0] * inv_h, wx);
                eval_all(-ref[1] * inv_h, wy);
                float wyz = wy[sy];
                float dy = ref[1] + (float)sy * h;
                float part[D];  // C[:,1..] * dpt[1..] + m v
                if constexpr (D == 3) {
                    eval_all(-ref[2] * inv_h, wz);
                    wyz *= wz[sz];
                    float dz = ref[2] + (float)sz * h;
#pragma unroll
                    for (int k = 0; k < 3; k++) part[k] = c[3 + k] * dy + c[6 + k] * dz + mv[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 2; k++) part[k] = c[2 + k] * dy + mv[k];
                }
                uint32_t paff = 0u;
                if constexpr (CPIC) paff = s_aff[r];
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    if constexpr (CPIC) {
                        // p2g.wgsl:200-228: incompatible pairs transfer nothing to the grid (their
                        // momentum goes to the rigid body as an impulse: two-way coupling)
                        if (!affinities_are_compatible(naff[s], paff)) continue;
                    }
                    const float w = wx[s] * wyz;
                    const float dx = ref[0] + (float)s * h;
#pragma unroll
                    for (int k = 0; k < D; k++) acc[s][k] += (c[k] * dx + part[k]) * w;
                    acc[s][D] += mass * w;
                }
            }
        }
        // partials: [cell][sx + 3 sy + 9 sz]
#pragma unroll
        for (int s = 0; s < 3; s++) {
            float4 o;
            if constexpr (D == 3) o = make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
            else o = make_float4(acc[s][0], acc[s][1], acc[s][2], 0.f);
            s_partial[cell * NBH + s + 3 * sub] = o;
        }
        __syncthreads();
        // ordered reduction per tile node, then one coalesced slab store
        for (int n = tid; n < TILE; n += Cfg::THREADS) {
            int t[3];
            t[0] = n % TW;
            t[1] = (n / TW) % TW;
            t[2] = D == 3 ? n / (TW * TW) : 0;
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            constexpr int SZN = D == 3 ? 3 : 1;
            for (int z = 0; z < SZN; z++) {
                int cz = t[2] - z;
                if (D == 3 && (cz < 0 || cz >= BW)) continue;
                for (int y = 0; y < 3; y++) {
                    int cy = t[1] - y;
                    if (cy < 0 || cy >= BW) continue;
                    for (int xx = 0; xx < 3; xx++) {
                        int cx = t[0] - xx;
                        if (cx < 0 || cx >= BW) continue;
                        int c = cx + (cy << BS) + (D == 3 ? (cz << (2 * BS)) : 0);
                        float4 p = s_partial[c * NBH + xx + 3 * y + 9 * z];
                        sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
                    }
                }
            }
            d.slab[(size_t)b * TILE + n] = sum;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ grid update
// Gather of the slabs covering each node + solver/grid_update.wgsl:55-64.
template <int D> __global__ __launch_bounds__(256) void k_grid_update(Dev d) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const uint32_t total = B * NPB;
    const float dt = d.sp->dt;
    const float lim = d.h / dt;
    float g[3] = {d.sp->gravity[0], d.sp->gravity[1], d.sp->gravity[2]};
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const uint32_t b = t >> 6, ln = t & 63u;
        int l[3];
        l[0] = ln & (BW - 1);
        l[1] = (ln >> BS) & (BW - 1);
        l[2] = D == 3 ? (ln >> (2 * BS)) : 0;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < NN; o++) {
            int tt[3] = {l[0] + BW * (o & 1), l[1] + BW * ((o >> 1) & 1), l[2] + BW * ((o >> 2) & 1)};
            bool in_tile = tt[0] < TW && tt[1] < TW && (D == 2 || tt[2] < TW);
            if (!in_tile) continue;
            uint32_t src = d.nbr_minus[b * 8u + o];
            if (src == NONE || d.block_count[src] == 0) continue;
            int ti = tt[0] + TW * tt[1] + (D == 3 ? TW * TW * tt[2] : 0);
            float4 p = d.slab[(size_t)src * TILE + ti];
            sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
        }
        float mass = D == 3 ? sum.w : sum.z;
        float inv_mass = mass > 0.f ? 1.0f / mass : 0.f;
        float mom[3] = {sum.x, sum.y, sum.z};
        float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D; k++) {
            float vel = (mom[k] + mass * g[k] * dt) * inv_mass;
            v[k] = fminf(fmaxf(vel, -lim), lim);
        }
        d.nodes[t] = D == 3 ? make_float4(v[0], v[1], v[2], mass) : make_float4(v[0], v[1], mass, 0.f);
    }
}

// ------------------------------------------------- fused G2P + particle update
// solver/g2p.wgsl:134-238 + solver/particle_update.wgsl:45-141 in one launch:
// the velocity gradient never leaves registers (the reference round-trips it
// through the `affine` buffer, g2p.wgsl:230-232), the SVD is computed once
// (reference: up to three times, quirk B8) and the result is written straight
// into the other ping-pong buffer in sorted order.
constexpr int G2P_THREADS = 256;
#ifndef G2P_WAVES_PER_EU
#define G2P_WAVES_PER_EU 3
#endif

template <int D, int MODEL, bool PLASTIC, bool CPIC>
__global__ __launch_bounds__(G2P_THREADS, G2P_WAVES_PER_EU) void k_g2p_update(Dev d, int side) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    constexpr int DD = D * D;
    using P = Pl<D>;
    __shared__ float4 s_node[TILE];
    __shared__ NodeCdf s_cdf[CPIC ? TILE : 1];

    const float *in = d.buf[side];
    float *out = d.buf[side ^ 1];
    const uint32_t npad = d.npad;
    const float h = d.h, inv_h = d.inv_h;
    const float dt = d.sp->dt;
    const float invd = 4.0f / (h * h);  // kernel.wgsl:56-58
    const int tid = threadIdx.x;

    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    for (uint32_t b = blockIdx.x; b < B; b += gridDim.x) {
        const uint32_t cnt = d.block_count[b];
        if (cnt == 0) continue;
        const uint32_t start = d.block_start[b];
        int bc[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[b], bc);
        __syncthreads();
        // g2p.wgsl:72-132: nodes of the block and of its +1 neighbours -> LDS tile
        for (int n = tid; n < TILE; n += G2P_THREADS) {
            int t[3];
            t[0] = n % TW;
            t[1] = (n / TW) % TW;
            t[2] = D == 3 ? n / (TW * TW) : 0;
            int o = (t[0] >= BW ? 1 : 0) | (t[1] >= BW ? 2 : 0) | (t[2] >= BW ? 4 : 0);
            int ln = (t[0] & (BW - 1)) + ((t[1] & (BW - 1)) << BS) + (D == 3 ? ((t[2] & (BW - 1)) << (2 * BS)) : 0);
            uint32_t nb = d.nbr_plus[b * 8u + o];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            NodeCdf cdf = {0.f, 0u, NONE, 0u};
            if (nb != NONE) {
                v = d.nodes[(size_t)nb * NPB + ln];
                if constexpr (CPIC) cdf = d.node_cdf[(size_t)nb * NPB + ln];
            }
            s_node[n] = v;
            if constexpr (CPIC) s_cdf[n] = cdf;
        }
        __syncthreads();

        for (uint32_t j = start + tid; j < start + cnt; j += G2P_THREADS) {
            const uint32_t src = d.perm[j];
            float x[D], Fm[DD], mass, vol0, lambda, mu;
            if constexpr (D == 3) {
                const float4 xm = ldq(in, npad, P::XM, src);
                const float4 f0 = ldq(in, npad, P::F0, src), f1 = ldq(in, npad, P::F1, src), f2 = ldq(in, npad, P::F2, src);
                x[0] = xm.x; x[1] = xm.y; x[2] = xm.z; mass = xm.w;
                Fm[0] = f0.x; Fm[1] = f0.y; Fm[2] = f0.z; Fm[3] = f0.w;
                Fm[4] = f1.x; Fm[5] = f1.y; Fm[6] = f1.z; Fm[7] = f1.w;
                Fm[8] = f2.x; vol0 = f2.y; lambda = f2.z; mu = f2.w;
            } else {
                const float4 xm = ldq(in, npad, P::XM, src);
                const float4 f0 = ldq(in, npad, P::F0, src);
                const float4 vl = ldq(in, npad, P::CV2, src);
                x[0] = xm.x; x[1] = xm.y; mass = xm.z; vol0 = xm.w;
                Fm[0] = f0.x; Fm[1] = f0.y; Fm[2] = f0.z; Fm[3] = f0.w;
                lambda = vl.z; mu = vl.w;
            }
            const uint32_t pid = ldpid<D>(in, npad, src);

            float pvel[D], nrm[D], sdist = 0.f;
            uint32_t paff = 0;
            if constexpr (CPIC) {
                const float4 c0 = ldq(in, npad, P::CDF0, src);
                nrm[0] = c0.x; nrm[1] = c0.y;
                if constexpr (D == 3) {
                    const float4 cv = ldq(in, npad, P::CV2, src);
                    const float4 c1 = ldq(in, npad, P::CDF1, src);
                    nrm[2] = c0.z; sdist = c0.w; paff = __float_as_uint(c1.w);
                    pvel[0] = cv.y; pvel[1] = cv.z; pvel[2] = cv.w;
                } else {
                    const float4 vl = ldq(in, npad, P::CV2, src);
                    sdist = c0.z; paff = __float_as_uint(c0.w);
                    pvel[0] = vl.x; pvel[1] = vl.y;
                }
            }

            // ---- G2P (g2p.wgsl:150-218)
            int lcell[D];
            float ref[D], w[D][3];
            int tbase = 0, stride = 1;
#pragma unroll
            for (int k = 0; k < D; k++) {
                int c = assoc_cell(x[k], h);
                lcell[k] = c - bc[k] * BW;  // in [0, BW)
                ref[k] = (float)c * h - x[k];
                eval_all(-ref[k] * inv_h, w[k]);
                tbase += lcell[k] * stride;
                stride *= TW;
            }
            float vel[D], grad[DD];
#pragma unroll
            for (int k = 0; k < D; k++) vel[k] = 0.f;
#pragma unroll
            for (int k = 0; k < DD; k++) grad[k] = 0.f;
            constexpr int SZN = D == 3 ? 3 : 1;
            // The z loop is kept rolled on purpose: fully unrolled, hipcc issues all 27
            // ds_read_b128 up front (108 VGPRs of tile values) and the kernel drops to 2 waves/SIMD.
#pragma unroll 1
            for (int sz = 0; sz < SZN; sz++)
#pragma unroll
                for (int sy = 0; sy < 3; sy++)
#pragma unroll
                    for (int sx = 0; sx < 3; sx++) {
                        const int idx = tbase + sx + TW * sy + (D == 3 ? TW * TW * sz : 0);
                        float4 nd = s_node[idx];
                        float nv[D];
                        nv[0] = nd.x; nv[1] = nd.y;
                        if constexpr (D == 3) nv[2] = nd.z;
                        float dpt[D];
                        dpt[0] = ref[0] + (float)sx * h;
                        dpt[1] = ref[1] + (float)sy * h;
                        if constexpr (D == 3) dpt[2] = ref[2] + (float)sz * h;
                        float wgt = w[0][sx] * w[1][sy];
                        if constexpr (D == 3) wgt *= (sz == 0 ? w[2][0] : (sz == 1 ? w[2][1] : w[2][2]));
                        if constexpr (CPIC) {
                            NodeCdf nc = s_cdf[idx];
                            if (!affinities_are_compatible(paff, nc.affinities)) {
                                if (nc.closest_id != NONE && nc.closest_id < d.n_colliders) {
                                    const ColliderDev &col = d.colliders[nc.closest_id];
                                    float cc[D], bv[D], rel[D], pr[D];
#pragma unroll
                                    for (int k = 0; k < D; k++) cc[k] = dpt[k] + x[k];
                                    velocity_at_point<D>(col, cc, bv);
#pragma unroll
                                    for (int k = 0; k < D; k++) rel[k] = pvel[k] - bv[k];
                                    project_velocity<D>(rel, nrm, pr);
#pragma unroll
                                    for (int k = 0; k < D; k++) nv[k] = bv[k] + pr[k];
                                } else {
#pragma unroll
                                    for (int k = 0; k < D; k++) nv[k] = pvel[k];
                                }
                            }
                        }
                        const float wi = wgt * invd;
#pragma unroll
                        for (int k = 0; k < D; k++) vel[k] += nv[k] * wgt;
#pragma unroll
                        for (int c = 0; c < D; c++)
#pragma unroll
                            for (int r = 0; r < D; r++) grad[c * D + r] += wi * (nv[r] * dpt[c]);
                    }

            float rvel[D];
#pragma unroll
            for (int k = 0; k < D; k++) rvel[k] = 0.f;
            if constexpr (CPIC) {  // g2p.wgsl:220-226 (bounded by the real collider count, quirk B9)
                for (uint32_t c = 0; c < d.n_colliders && c < 16u; c++)
                    if (paff & (1u << c)) {
                        float bv[D];
                        velocity_at_point<D>(d.colliders[c], x, bv);
#pragma unroll
                        for (int k = 0; k < D; k++) rvel[k] += bv[k];
                    }
            }

            // ---- particle update (particle_update.wgsl:58-132)
            if constexpr (CPIC) {
                if (sdist < -0.05f * h) {
                    float rel[D], pr[D];
#pragma unroll
                    for (int k = 0; k < D; k++) rel[k] = vel[k] - rvel[k];
                    project_velocity<D>(rel, nrm, pr);
#pragma unroll
                    for (int k = 0; k < D; k++) vel[k] = rvel[k] + pr[k];
                }
            }
            float l2 = 0.f;
#pragma unroll
            for (int k = 0; k < D; k++) l2 += vel[k] * vel[k];
            const float len = sqrtf(l2);
            if (len > h / dt) {
#pragma unroll
                for (int k = 0; k < D; k++) vel[k] = vel[k] / len * h / dt;
            }
            float xn[D];
#pragma unroll
            for (int k = 0; k < D; k++) xn[k] = x[k] + vel[k] * dt;
            if constexpr (CPIC) {
                if (sdist < -0.05f * h) {
                    const float corrected = fmaxf(sdist, -0.3f * h);
                    const float imp = dt * -corrected * 1.0e3f;
#pragma unroll
                    for (int k = 0; k < D; k++) vel[k] += imp * nrm[k];
                }
            }
            // F <- F + (grad * dt) * F
            float gdt[DD], prod[DD];
#pragma unroll
            for (int k = 0; k < DD; k++) gdt[k] = grad[k] * dt;
            mat_mul<D>(gdt, Fm, prod);
#pragma unroll
            for (int k = 0; k < DD; k++) Fm[k] += prod[k];

            float tau[DD];
            bool have_svd = false;
            Svd<D> sv;
            if constexpr (PLASTIC) {
                float dp[6], st[3], phase, max_stretch;
                {
                    const float4 d0 = ldq(in, npad, P::DP0, src), d1 = ldq(in, npad, P::DP1, src), d2 = ldq(in, npad, P::DP2, src);
                    dp[0] = d0.x; dp[1] = d0.y; dp[2] = d0.z; dp[3] = d0.w; dp[4] = d1.x; dp[5] = d1.y;
                    st[0] = d1.z; st[1] = d1.w; st[2] = d2.x; phase = d2.y; max_stretch = d2.z;
                }
                if (phase > 0.f && max_stretch > 0.f) {  // particle_update.wgsl:98-116
                    svd<D>(Fm, sv);
                    have_svd = true;
                    bool broken = false;
#pragma unroll
                    for (int k = 0; k < D; k++) broken = broken || sv.s[k] > max_stretch;
                    if (broken) phase = 0.f;
                }
                if (phase == 0.f && dp[4] != 0.f) {  // particle_update.wgsl:118-122, drucker_prager.wgsl:134
                    if (!have_svd) svd<D>(Fm, sv);
                    have_svd = true;
                    drucker_prager_project<D>(dp, st, Fm, sv);  // sv.s follows the projected F
                }
                stq(out, npad, P::DP0, j, make_float4(dp[0], dp[1], dp[2], dp[3]));
                stq(out, npad, P::DP1, j, make_float4(dp[4], dp[5], st[0], st[1]));
                stq(out, npad, P::DP2, j, make_float4(st[2], phase, max_stretch, 0.f));
            }
            if constexpr (MODEL == 1) {
                kirchoff_neo_hookean<D>(lambda, mu, Fm, tau);
            } else {
                if (!have_svd) svd<D>(Fm, sv);
                kirchoff_corotated<D>(lambda, mu, Fm, sv, tau);
            }
            // particle_update.wgsl:129-132: C' = grad * m - tau * (V0 * inv_d * dt)
            const float coeff = vol0 * invd * dt;
            float Cn[DD];
#pragma unroll
            for (int k = 0; k < DD; k++) Cn[k] = grad[k] * mass - tau[k] * coeff;
            if constexpr (D == 3) {
                stq(out, npad, P::XM, j, make_float4(xn[0], xn[1], xn[2], mass));
                stq(out, npad, P::CV0, j, make_float4(Cn[0], Cn[1], Cn[2], Cn[3]));
                stq(out, npad, P::CV1, j, make_float4(Cn[4], Cn[5], Cn[6], Cn[7]));
                stq(out, npad, P::CV2, j, make_float4(Cn[8], vel[0], vel[1], vel[2]));
                stq(out, npad, P::F0, j, make_float4(Fm[0], Fm[1], Fm[2], Fm[3]));
                stq(out, npad, P::F1, j, make_float4(Fm[4], Fm[5], Fm[6], Fm[7]));
                stq(out, npad, P::F2, j, make_float4(Fm[8], vol0, lambda, mu));
            } else {
                stq(out, npad, P::XM, j, make_float4(xn[0], xn[1], mass, vol0));
                stq(out, npad, P::CV0, j, make_float4(Cn[0], Cn[1], Cn[2], Cn[3]));
                stq(out, npad, P::CV2, j, make_float4(vel[0], vel[1], lambda, mu));
                stq(out, npad, P::F0, j, make_float4(Fm[0], Fm[1], Fm[2], Fm[3]));
            }
            stpid<D>(out, npad, j, pid);
            if constexpr (CPIC) {
                if constexpr (D == 3) {
                    stq(out, npad, P::CDF0, j, make_float4(nrm[0], nrm[1], nrm[2], sdist));
                    stq(out, npad, P::CDF1, j, make_float4(rvel[0], rvel[1], rvel[2], __uint_as_float(paff)));
                } else {
                    stq(out, npad, P::CDF0, j, make_float4(nrm[0], nrm[1], sdist, __uint_as_float(paff)));
                    stq(out, npad, P::CDF1, j, make_float4(rvel[0], rvel[1], 0.f, 0.f));
                }
            }
        }
    }
}

}  // namespace wgs
