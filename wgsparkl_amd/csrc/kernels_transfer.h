// kernels_transfer.h — P2G, grid update and the fused G2P + particle update.
//
// P2G (reference: solver/p2g.wgsl:69-236, a per-node gather over linked lists of
// 8 blocks with ~3.4x redundant particle fetches) is re-expressed as a per-block
// SCATTER with no atomics and a fixed summation order: a workgroup owns one block,
// every thread accumulates in registers over the (cell-sorted) particles of its own
// cell, and the (BW+2)^D tile (block + "+1" rim) is written with plain coalesced
// stores to the block's slab.
// The grid update then gathers, for every node, the (at most 2^D) slabs that
// cover it, in a fixed order, and applies grid_update.wgsl:55-64 in the same
// pass (the reference keeps P2G and grid update apart only because of WebGPU's
// binding limit, p2g.wgsl:129-133).
#pragma once
#include "kernels_bodies.h"

namespace wgs {

// ------------------------------------------------------------------- P2G
// Work decomposition (3D): workgroup = one block, 3 waves; lane = cell of the block (64),
// wave = z-offset sz of the target node. A thread walks the particles of ITS cell in
// canonical order and accumulates the 9 (sx, sy) contributions for its sz in registers:
// no cross-lane traffic, no atomics, a fixed summation order.
// Particles are staged through LDS in rounds of P2G_J ranks per cell, transposed to
// [rank][cell] so that the 64 lanes of a wave read 64 consecutive float4 (conflict-free
// ds_read_b128) while the global side reads whole 64-byte runs of the cell-sorted arrays.
constexpr int P2G_J = 4;
template <int D> struct P2GCfg;
template <> struct P2GCfg<3> {
    static constexpr int NW = 3;        // waves per workgroup = sz values
    static constexpr int NSXY = 9;      // (sx, sy) pairs per thread
    static constexpr int NQ = 4;        // staged quads per particle: XM, CV0, CV1, CV2
};
template <> struct P2GCfg<2> {
    static constexpr int NW = 1;
    static constexpr int NSXY = 9;      // (sx, sy): the whole 3x3 stencil in one wave
    static constexpr int NQ = 3;        // XM, CV0, CV2
};

// `filter`: 0 = every block; in collider simulations the pass is launched twice, CPIC = false with
// filter 1 (blocks whose tile sees no collider: every affinity is 0, plain MLS-MPM) and CPIC = true
// with filter 2 (blocks near a collider).
// TWOWAY (with CPIC): the momentum an incompatible (particle, node) pair does not transfer is accumulated
// as an impulse on the node's closest body (p2g.wgsl:200-228), per node, in a second LDS tile; the
// per-block partial node sums go to imp_slab and are gathered, converted to fixed point and added to the
// bodies by the grid update (p2g.wgsl:142-155).
// PCDF (with CPIC): the particle cdf of the block's particles (g2p_cdf.wgsl) is computed in the prologue, from the
// node cdfs k_block_setup<CDF> left in node_cdf — the third step of k_cdf without a launch of its own.
template <int D, bool CPIC, bool TWOWAY = false, bool PCDF = false>
__global__ __launch_bounds__(P2GCfg<D>::NW * 64) void k_p2g(Dev d, int side, int filter, uint32_t epoch) {
    using Cfg = P2GCfg<D>;
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE;
    constexpr int NT = Cfg::NW * 64;
    constexpr int SLOTS = P2G_J * NPB;
    constexpr int ROW = NPB + 4;                 // padded [rank] row: the J float4 a (cell) group writes land on distinct banks
    constexpr int KS = (SLOTS + NT - 1) / NT;    // staging slots per thread and round
    constexpr int NQ = Cfg::NQ;
    __shared__ float4 s_q[NQ][P2G_J * ROW];
    __shared__ uint32_t s_aff[CPIC ? P2G_J * ROW : 1];
    __shared__ float4 s_tile[Cfg::NW][TILE];
    __shared__ uint32_t s_cs[NPB], s_cn[NPB];
    constexpr int IMPQ = D == 3 ? 2 : 1;  // impulse quads per node: (lin, 0), (ang, 0) | (lin.xy, ang, 0)
    __shared__ float4 s_nrm[TWOWAY ? P2G_J * ROW : 1];
    __shared__ NodeCdf s_ncdf[PCDF ? TILE : 1];
    __shared__ float4 s_imp[TWOWAY ? Cfg::NW : 1][TWOWAY ? IMPQ : 1][TWOWAY ? TILE : 1];

    const float *in = d.buf[side];
    const uint32_t npad = d.npad;
    const float h = d.h, inv_h = d.inv_h;
    const int tid = threadIdx.x;
    const int cell = tid & 63;
    const int sz = D == 3 ? (tid >> 6) : 0;  // wave-uniform
    int lc[3];
    lc[0] = cell & (BW - 1);
    lc[1] = (cell >> BS) & (BW - 1);
    lc[2] = D == 3 ? (cell >> (2 * BS)) : 0;
    const int tnode0 = lc[0] + TW * lc[1] + (D == 3 ? TW * TW * (lc[2] + sz) : 0);  // tile node of (sx,sy) = (0,0)

    // filter 2 walks the (short) list of blocks near a collider, the others the active list
    const uint32_t B = min(d.counters[filter == 2 ? CTR_NCPIC : CTR_NBLOCKS], d.cap);
    for (uint32_t a = blockIdx.x; a < B; a += gridDim.x) {
        const uint32_t b = filter == 2 ? d.cpic_list[a] : d.active[a];
        const uint32_t cnt = d.block_count[b];
        if (cnt == 0) continue;  // no particles: its slab is never read
        if (filter == 1 && d.block_cpic[b] != 0u) continue;
        int bc[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[b], bc);
        const uint32_t cs = d.cell_start[b * NPB + cell];
        const uint32_t cn = d.cell_cursor[b * NPB + cell] - cs;
        __syncthreads();  // previous block fully consumed
        if constexpr (PCDF) {
            stage_node_cdf_tile<D, NT>(d, b, s_ncdf, tid);
            __syncthreads();
            const uint32_t bstart = d.block_start[b];
            for (uint32_t j = bstart + tid; j < bstart + cnt; j += NT)
                particle_cdf_update<D>(d, d.buf[side], d.perm[j], s_ncdf, bc, epoch);
            __threadfence_block();
            __syncthreads();  // the affinities written above are fetched below by other threads of the workgroup
        }
        if (tid < NPB) {
            s_cs[cell] = cs;
            s_cn[cell] = cn;
        }
        for (int n = cell; n < TILE; n += 64) s_tile[sz][n] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (TWOWAY) {
            for (int n = cell; n < TILE; n += 64)
                for (int q = 0; q < IMPQ; q++) s_imp[sz][q][n] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        uint32_t maxc = cn;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, off));
        float cpos[D];
#pragma unroll
        for (int k = 0; k < D; k++) cpos[k] = (float)(bc[k] * BW + lc[k]) * h;
        uint32_t naff[Cfg::NSXY];
        unsigned long long ncol = 0ull;  // closest collider of the nine nodes: 4 bits each + valid bit at 36 + s
        if constexpr (CPIC) {  // affinities of this thread's nine target nodes (p2g.wgsl:100-103)
#pragma unroll
            for (int s = 0; s < 9; s++) {
                int t[3] = {lc[0] + s % 3, lc[1] + s / 3, lc[2] + sz};
                int o = (t[0] >= BW ? 1 : 0) | (t[1] >= BW ? 2 : 0) | (t[2] >= BW ? 4 : 0);
                int ln = (t[0] & (BW - 1)) + ((t[1] & (BW - 1)) << BS) + (D == 3 ? ((t[2] & (BW - 1)) << (2 * BS)) : 0);
                uint32_t nb = d.nbr_plus[b * 8u + o];
                naff[s] = nb != NONE ? d.node_cdf[(size_t)nb * NPB + ln].affinities : 0u;
                if constexpr (TWOWAY) {
                    const uint32_t cl = nb != NONE ? d.node_cdf[(size_t)nb * NPB + ln].closest_id : NONE;
                    if (cl < 16u) ncol |= ((unsigned long long)cl << (4 * s)) | (1ull << (36 + s));
                }
            }
        }
        float acc[Cfg::NSXY][D + 1];
#pragma unroll
        for (int s = 0; s < 9; s++)
#pragma unroll
            for (int k = 0; k <= D; k++) acc[s][k] = 0.f;

        // Software-pipelined staging: the particle quads of round r+1 are fetched into registers while
        // round r is being accumulated from LDS, so the HBM latency of a round hides behind the
        // previous round's arithmetic instead of being paid at every barrier.
        // Staging slot sl -> (cell c = sl / J, rank j = sl % J): a cell's J particles are one contiguous
        // 64-byte run of each quad in HBM; in LDS they go to [j][c] (row padded to ROW).
        float4 pre[KS][NQ];
        float4 pre_nrm[TWOWAY ? KS : 1];
        uint32_t pre_aff[KS];
        bool pre_ok[KS];
        __syncthreads();  // s_cs / s_cn visible
        auto fetch_round = [&](uint32_t r0) {
#pragma unroll
            for (int k = 0; k < KS; k++) {
                const int sl = tid + k * NT;
                pre_ok[k] = false;
                if (sl < SLOTS) {
                    const int c = sl / P2G_J;
                    const uint32_t rank = r0 + (uint32_t)(sl % P2G_J);
                    if (rank < s_cn[c] && !(d.dbg & 512u)) {
                        const uint32_t src = d.perm[s_cs[c] + rank];
                        pre_ok[k] = true;
                        if constexpr (D == 3) {
                            pre[k][0] = ldq(in, npad, Pl<3>::XM, src);
                            pre[k][1] = ldq(in, npad, Pl<3>::CV0, src);
                            pre[k][2] = ldq(in, npad, Pl<3>::CV1, src);
                            pre[k][3] = ldq(in, npad, Pl<3>::CV2, src);
                        } else {
                            pre[k][0] = ldq(in, npad, Pl<2>::XM, src);   // x, y, m, V0
                            pre[k][1] = ldq(in, npad, Pl<2>::CV0, src);
                            pre[k][2] = ldq(in, npad, Pl<2>::CV2, src);  // vx, vy, lambda, mu
                        }
                        if constexpr (CPIC) {
                            const float4 cd = ldq(in, npad, D == 3 ? (int)Pl<D>::CDF1 : (int)Pl<D>::CDF0, src);
                            pre_aff[k] = __float_as_uint(cd.w);
                            if constexpr (TWOWAY) pre_nrm[k] = D == 3 ? ldq(in, npad, Pl<D>::CDF0, src) : cd;  // cdf normal
                        }
                    }
                }
            }
        };
        fetch_round(0);
        for (uint32_t r0 = 0; r0 < maxc; r0 += P2G_J) {
            __syncthreads();  // previous round consumed
#pragma unroll
            for (int k = 0; k < KS; k++) {
                if (pre_ok[k]) {
                    const int sl = tid + k * NT;
                    const int dst = (sl % P2G_J) * ROW + sl / P2G_J;
                    if constexpr (D == 3) {
                        float4 c2 = pre[k][3];
                        const float m = TWOWAY ? 1.f : pre[k][0].w;  // (the two-way path needs the raw velocity too)
                        c2.y *= m; c2.z *= m; c2.w *= m;  // momentum m v
                        s_q[0][dst] = pre[k][0]; s_q[1][dst] = pre[k][1]; s_q[2][dst] = pre[k][2]; s_q[3][dst] = c2;
                    } else {
                        float4 vl = pre[k][2];
                        const float m = TWOWAY ? 1.f : pre[k][0].z;
                        vl.x *= m; vl.y *= m;
                        s_q[0][dst] = pre[k][0]; s_q[1][dst] = pre[k][1]; s_q[2][dst] = vl;
                    }
                    if constexpr (CPIC) s_aff[dst] = pre_aff[k];
                    if constexpr (TWOWAY) s_nrm[dst] = pre_nrm[k];
                }
            }
            __syncthreads();
            if (r0 + P2G_J < maxc) fetch_round(r0 + P2G_J);  // in flight during the accumulation below
            const uint32_t jn = (cn > r0 && !(d.dbg & 256u)) ? min((uint32_t)P2G_J, cn - r0) : 0u;
            for (uint32_t j = 0; j < jn; j++) {
                const int sl = j * ROW + cell;
                float x[D], mv[D], c[D * D], mass;
                if constexpr (D == 3) {
                    const float4 xm = s_q[0][sl], c0 = s_q[1][sl], c1 = s_q[2][sl], c2 = s_q[3][sl];
                    x[0] = xm.x; x[1] = xm.y; x[2] = xm.z; mass = xm.w;
                    c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w;
                    c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w; c[8] = c2.x;
                    mv[0] = c2.y; mv[1] = c2.z; mv[2] = c2.w;
                } else {
                    const float4 xm = s_q[0][sl], c0 = s_q[1][sl], vl = s_q[2][sl];
                    x[0] = xm.x; x[1] = xm.y; mass = xm.z;
                    c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w;
                    mv[0] = vl.x; mv[1] = vl.y;
                }
                uint32_t paff = 0u;
                if constexpr (CPIC) paff = s_aff[sl];
                float pv[D], pn[D];  // raw particle velocity and cdf normal (two-way coupling)
                if constexpr (TWOWAY) {
                    const float4 nq = s_nrm[sl];
                    pn[0] = nq.x; pn[1] = nq.y;
                    if constexpr (D == 3) pn[2] = nq.z;
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        pv[k] = mv[k];
                        mv[k] = mv[k] * mass;
                    }
                }
                // p2g.wgsl:176-198: ref = assoc_node - x ; w = eval_all(-ref / h) ; dpt = ref + shift * h
                float ref[D], wx[3], wy[3];
#pragma unroll
                for (int k = 0; k < D; k++) ref[k] = cpos[k] - x[k];
                eval_all(-ref[0] * inv_h, wx);
                eval_all(-ref[1] * inv_h, wy);
                float wzs = 1.f;
                float base[D];  // C[:, 2] * dz + m v
                if constexpr (D == 3) {
                    float wz[3];
                    eval_all(-ref[2] * inv_h, wz);
                    wzs = sz == 0 ? wz[0] : (sz == 1 ? wz[1] : wz[2]);
                    const float dz = ref[2] + (float)sz * h;
#pragma unroll
                    for (int k = 0; k < 3; k++) base[k] = c[6 + k] * dz + mv[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 2; k++) base[k] = mv[k];
                }
#pragma unroll
                for (int sy = 0; sy < 3; sy++) {
                    const float dy = ref[1] + (float)sy * h;
                    const float wyz = wy[sy] * wzs;
                    float py[D];
#pragma unroll
                    for (int k = 0; k < D; k++) py[k] = c[D + k] * dy + base[k];
#pragma unroll
                    for (int sx = 0; sx < 3; sx++) {
                        const int s = sx + 3 * sy;
                        if constexpr (CPIC) {
                            // p2g.wgsl:200-228: incompatible pairs transfer nothing to the grid (their momentum
                            // becomes an impulse on the rigid body: two-way coupling)
                            if (!affinities_are_compatible(naff[s], paff)) {
                                if constexpr (TWOWAY) {
                                    if ((ncol >> (36 + s)) & 1ull) {  // p2g.wgsl:203-224
                                        const ColliderDev &cl = d.colliders[(uint32_t)(ncol >> (4 * s)) & 15u];
                                        const float wgt = wx[sx] * wyz;
                                        float dpt[3] = {ref[0] + (float)sx * h, dy, 0.f}, cc[3] = {0.f, 0.f, 0.f};
                                        if constexpr (D == 3) dpt[2] = ref[2] + (float)sz * h;
#pragma unroll
                                        for (int k = 0; k < D; k++) cc[k] = dpt[k] + x[k];  // cell_center
                                        float bv[D], rel[D], prj[D], dimp[3] = {0.f, 0.f, 0.f}, lever[3] = {0.f, 0.f, 0.f};
                                        velocity_at_point<D>(cl, cc, bv);
#pragma unroll
                                        for (int k = 0; k < D; k++) rel[k] = pv[k] - bv[k];
                                        project_velocity<D>(rel, pn, prj);
#pragma unroll
                                        for (int k = 0; k < D; k++) {
                                            const float ghost = bv[k] + prj[k];
                                            dimp[k] = (pv[k] - ghost) * (wgt * mass);
                                            lever[k] = cl.com[k] - cc[k];
                                        }
                                        const int node = tnode0 + sx + TW * sy;
                                        if constexpr (D == 3) {
                                            float4 a = s_imp[sz][0][node], bq = s_imp[sz][1][node];
                                            a.x += dimp[0]; a.y += dimp[1]; a.z += dimp[2];
                                            bq.x += dimp[1] * lever[2] - dimp[2] * lever[1];
                                            bq.y += dimp[2] * lever[0] - dimp[0] * lever[2];
                                            bq.z += dimp[0] * lever[1] - dimp[1] * lever[0];
                                            s_imp[sz][0][node] = a;
                                            s_imp[sz][1][node] = bq;
                                        } else {
                                            float4 a = s_imp[sz][0][node];
                                            a.x += dimp[0]; a.y += dimp[1];
                                            a.z += dimp[0] * lever[1] + dimp[1] * (-lever[0]);
                                            s_imp[sz][0][node] = a;
                                        }
                                        // lanes own distinct nodes within one (sx, sy) phase; keep the phases in order
                                        asm volatile("" ::: "memory");
                                    }
                                }
                                continue;
                            }
                        }
                        const float dx = ref[0] + (float)sx * h;
                        const float w = wx[sx] * wyz;
#pragma unroll
                        for (int k = 0; k < D; k++) acc[s][k] += (c[k] * dx + py[k]) * w;
                        acc[s][D] += mass * w;
                    }
                }
            }
        }
        // Per-wave tile: in phase (sx, sy) every lane (cell) owns a distinct node, so a plain float4
        // read-add-write is race-free inside the wave and the LDS executes a wave's accesses in order.
        // The asm memory clobbers stop hipcc from hoisting a later phase's load above an earlier phase's
        // store (legal for one thread, wrong across lanes).
        {
#pragma unroll
            for (int s = 0; s < 9; s++) {
                float4 *p = &s_tile[sz][tnode0 + (s % 3) + TW * (s / 3)];
                float4 v = *p;
                v.x += acc[s][0]; v.y += acc[s][1]; v.z += acc[s][2];
                if constexpr (D == 3) v.w += acc[s][3];
                *p = v;
                asm volatile("" ::: "memory");
            }
        }
        __syncthreads();
        // combine the sz tiles in a fixed order; one coalesced slab store per node
        for (int n = tid; n < TILE; n += NT) {
            float4 sum = s_tile[0][n];
#pragma unroll
            for (int w = 1; w < Cfg::NW; w++) {
                const float4 p = s_tile[w][n];
                sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
            }
            d.slab[(size_t)b * TILE + n] = sum;
            if constexpr (TWOWAY) {
#pragma unroll
                for (int q = 0; q < IMPQ; q++) {
                    float4 is = s_imp[0][q][n];
#pragma unroll
                    for (int w = 1; w < Cfg::NW; w++) {
                        const float4 p = s_imp[w][q][n];
                        is.x += p.x; is.y += p.y; is.z += p.z;
                    }
                    d.imp_slab[((size_t)b * TILE + n) * IMPQ + q] = is;
                }
            }
        }
    }
}

// ------------------------------------------------------------ grid update
// Gather of the slabs covering each node + solver/grid_update.wgsl:55-64.
// PHASE 0: gather + update in one pass (single GPU). Sharded runs: PHASE 3 = the same single pass, except that the
// interface node layers come from nodes[] (gathered and exchanged by k_pack_halos / k_add_halo). PHASE 1 = gather only
// (partial momentum / mass sums into nodes[]) and PHASE 2 = update from nodes[] remain for callers that pack the two
// faces separately (wgs_shard_pack_halo).

// Partial (momentum, mass) sum of one node from the (at most 2^D) slabs that cover it, in the fixed order of the
// grid update.
template <int D> __device__ inline float4 gather_slabs(const Dev &d, uint32_t b, uint32_t ln) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    const int l[3] = {(int)(ln & (BW - 1)), (int)((ln >> BS) & (BW - 1)), D == 3 ? (int)(ln >> (2 * BS)) : 0};
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int o = 0; o < NN; o++) {
        const int tt[3] = {l[0] + BW * (o & 1), l[1] + BW * ((o >> 1) & 1), l[2] + BW * ((o >> 2) & 1)};
        const bool in_tile = tt[0] < TW && tt[1] < TW && (D == 2 || tt[2] < TW);
        if (!in_tile) continue;
        const uint32_t src = d.nbr_minus[b * 8u + o];
        if (src == NONE || d.block_count[src] == 0) continue;
        const float4 p = d.slab[(size_t)src * TILE + tt[0] + TW * tt[1] + (D == 3 ? TW * TW * tt[2] : 0)];
        sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
    }
    return sum;
}

template <int D, int PHASE, bool TWOWAY = false> __global__ __launch_bounds__(256) void k_grid_update(Dev d) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const uint32_t total = B * NPB;
    const float dt = d.sp->dt;
    const float lim = d.h / dt;
    float g[3] = {d.sp->gravity[0], d.sp->gravity[1], d.sp->gravity[2]};
    // Two-way coupling: the node impulses are summed per body in LDS first (integers: any order gives the same sum) and
    // leave the workgroup as at most 16 x 6 global atomics. One atomic per node and component instead serialises at the
    // memory side: 40 k of them on a dozen addresses took 290 us in a scene whose cube rests on the floor.
    __shared__ int32_t s_body_imp[TWOWAY ? 128 : 1];
    if constexpr (TWOWAY) {
        if (threadIdx.x < 128) s_body_imp[threadIdx.x] = 0;
        __syncthreads();
    }
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const uint32_t b = d.active[t >> 6], ln = t & 63u;
        const uint32_t node = b * NPB + ln;
        int l[3];
        l[0] = ln & (BW - 1);
        l[1] = (ln >> BS) & (BW - 1);
        l[2] = D == 3 ? (ln >> (2 * BS)) : 0;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        // PHASE 3 (sharded runs): the two interface node layers of the blocks of layer shard_lo / shard_hi were
        // gathered by k_pack_halos and completed with the neighbour's partial sums (k_add_halo): take them from nodes[]
        bool from_nodes = false;
        if constexpr (PHASE == 3) {
            int bc[3] = {0, 0, 0};
            unpack_key<D>(d.block_key[b], bc);
            from_nodes = l[0] < 2 && (bc[0] == d.shard_lo || bc[0] == d.shard_hi);
        }
        float isum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // node impulse (two-way coupling): linear, angular
        uint32_t srcs[NN];
        int tis[NN];
#pragma unroll
        for (int o = 0; o < NN; o++) {
            srcs[o] = NONE;
            int tt[3] = {l[0] + BW * (o & 1), l[1] + BW * ((o >> 1) & 1), l[2] + BW * ((o >> 2) & 1)};
            bool in_tile = tt[0] < TW && tt[1] < TW && (D == 2 || tt[2] < TW);
            if (!in_tile) continue;
            uint32_t src = d.nbr_minus[b * 8u + o];
            if (src == NONE || d.block_count[src] == 0) continue;
            int ti = tt[0] + TW * tt[1] + (D == 3 ? TW * TW * tt[2] : 0);
            srcs[o] = src;
            tis[o] = ti;
            if (PHASE != 2 && !(PHASE == 3 && from_nodes)) {
                float4 p = d.slab[(size_t)src * TILE + ti];
                sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
                if constexpr (TWOWAY) {
                    if (d.block_cpic[src] != 0u) {  // only the CPIC launch of P2G writes impulse partials
                        constexpr int IMPQ = D == 3 ? 2 : 1;
                        const float4 a = d.imp_slab[((size_t)src * TILE + ti) * IMPQ];
                        isum[0] += a.x; isum[1] += a.y; isum[2] += a.z;
                        if constexpr (D == 3) {
                            const float4 bq = d.imp_slab[((size_t)src * TILE + ti) * IMPQ + 1];
                            isum[3] += bq.x; isum[4] += bq.y; isum[5] += bq.z;
                        }
                    }
                }
            }
        }
        if constexpr (TWOWAY) {
            // p2g.wgsl:142-155: the node's total impulse goes to its closest body, in fixed point (integer
            // atomics: order-independent, like the reference)
            const uint32_t cl = d.node_cdf[node].closest_id;
            if (cl < 16u) {
                constexpr int NI = D == 3 ? 6 : 3;
#pragma unroll
                for (int k = 0; k < NI; k++) {
                    const int32_t v = flt2int(isum[k]);
                    if (v != 0) atomicAdd(&s_body_imp[cl * 8u + k], v);
                }
            }
        }
        if constexpr (PHASE == 1) {
            d.nodes[node] = sum;
            continue;
        }
        d.cell_count[node] = 0;  // the sort's per-cell accumulators are zero at rest (last read: k_setup_scatter)
        if constexpr (PHASE == 2) sum = d.nodes[node];
        if constexpr (PHASE == 3) {
            if (from_nodes) sum = d.nodes[node];
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
        const float4 nv = D == 3 ? make_float4(v[0], v[1], v[2], mass) : make_float4(v[0], v[1], mass, 0.f);
        // Write the node back into every slab entry it was gathered from: each (block, tile
        // index) pair is read and written by exactly this thread, so the slabs turn in place
        // from per-block momentum tiles into per-block VELOCITY tiles, which is what the fused
        // G2P kernel stages (one contiguous 3.4 KiB read per block, no neighbour indirection).
#pragma unroll
        for (int o = 0; o < NN; o++)
            if (srcs[o] != NONE) d.slab[(size_t)srcs[o] * TILE + tis[o]] = nv;
        d.nodes[node] = nv;
    }
    if constexpr (TWOWAY) {
        __syncthreads();
        if (threadIdx.x < 128) {
            const int32_t v = s_body_imp[threadIdx.x];
            if (v != 0) atomicAdd(&d.impulses[threadIdx.x], v);
        }
    }
}

// ------------------------------------------------- fused G2P + particle update
// solver/g2p.wgsl:134-238 + solver/particle_update.wgsl:45-141 in one launch:
// the velocity gradient never leaves registers (the reference round-trips it
// through the `affine` buffer, g2p.wgsl:230-232), the SVD is computed once
// (reference: up to three times, quirk B8) and the result is written straight
// into the other ping-pong buffer in sorted order.
// Launch shape: one 64-lane workgroup per 64 consecutive SORTED particles (no per-block
// serial loop: ~16 k independent waves at 1 M particles, so the hardware overlaps the
// perm -> particle-load -> node-tile -> compute -> store chains of many waves). The wave
// stages the node tile of its particles' block in LDS itself (3.4 KiB); the 1-in-8 waves
// that straddle a block boundary do it once per distinct block.
constexpr int G2P_THREADS = 64;
#ifndef G2P_WAVES_PER_EU
#define G2P_WAVES_PER_EU 3
#endif

// CMODE: 0 = simulation without colliders; 1 = collider simulation, this launch handles the blocks
// whose tile sees no collider (plain MLS-MPM maths, writes default_cdf()); 2 = collider simulation,
// blocks near a collider (full CPIC). Modes 1 and 2 are launched back to back and partition the blocks.
// CMODE 0 / 1: one workgroup (one wave) per chunk = 64 consecutive particles of the sorted order.
// CMODE 2 walks the (short) list of blocks near a collider instead: blockIdx.y strides the list, blockIdx.x the
// chunks a block spans, and only that block's particles are processed; the launch is nearly free while no
// particle is near a collider.
#define G2P_DONE                                   \
    {                                              \
        if constexpr (G2P_CMODE == 2) continue;    \
        else return;                               \
    }
template <int D, int MODEL, bool PLASTIC, int CMODE>
__global__ __launch_bounds__(G2P_THREADS, G2P_WAVES_PER_EU) void k_g2p_update(Dev d, int side, uint32_t epoch) {
    __shared__ float4 s_node[Dim<D>::TILE];
    __shared__ NodeCdf s_cdf[CMODE == 2 ? Dim<D>::TILE : 1];
#define G2P_CMODE CMODE
#define G2P_BX blockIdx.x
#define G2P_BY blockIdx.y
#define G2P_GX gridDim.x
#define G2P_GY gridDim.y
#include "g2p_body.inc"
#undef G2P_CMODE
#undef G2P_BX
#undef G2P_BY
#undef G2P_GX
#undef G2P_GY
}

// Collider simulations: the body of CMODE 2 (walk of the near-collider block list) and the body of CMODE 1 (blocks
// away from colliders, one workgroup per 64 sorted particles) in ONE launch. The first 8 x `nlist` workgroups walk
// the list — they start at once and run beside the others —, the `nmain` workgroups after them run the main body;
// 8 * nlist and nmain are multiples of 8, so the XCD-aware chunk mapping of the main body is unchanged. Both bodies
// are capped at the same register budget, so the main path keeps its occupancy; the launch saves the ~4.5 us a
// dependent kernel boundary costs even when the list is empty. `nlist` follows the list length the host last saw
// (wgs_sync), so an empty list costs a few hundred workgroups that exit at once.
template <int D, int MODEL, bool PLASTIC>
__global__ __launch_bounds__(G2P_THREADS, G2P_WAVES_PER_EU) void k_g2p_pair(Dev d, int side, uint32_t epoch, uint32_t nmain, uint32_t nlist) {
    __shared__ float4 s_node[Dim<D>::TILE];
    __shared__ NodeCdf s_cdf[Dim<D>::TILE];
    if (blockIdx.x >= 8u * nlist) {
        const uint32_t widx = blockIdx.x - 8u * nlist;
#define G2P_CMODE 1
#define G2P_BX widx
#define G2P_BY 0u
#define G2P_GX nmain
#define G2P_GY 1u
#include "g2p_body.inc"
#undef G2P_CMODE
#undef G2P_BX
#undef G2P_BY
#undef G2P_GX
#undef G2P_GY
    } else {
#define G2P_CMODE 2
#define G2P_BX (blockIdx.x & 7u)
#define G2P_BY (blockIdx.x >> 3)
#define G2P_GX 8u
#define G2P_GY nlist
#include "g2p_body.inc"
#undef G2P_CMODE
#undef G2P_BX
#undef G2P_BY
#undef G2P_GX
#undef G2P_GY
    }
}
#undef G2P_DONE

}  // namespace wgs
