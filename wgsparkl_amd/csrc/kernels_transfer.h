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
#include "kernels_shard.h"

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
#ifndef WGS_P2G_ROW_PAD
#define WGS_P2G_ROW_PAD 2   // (see ROW below; 4 = rounds 1-3)
#endif
#ifdef WGS_ABLATE
// stage clocks of P2G (timing experiments, tools/gpu_p2g_prof.py): one row per active-list index of the plain body
constexpr int WGS_P2G_ROWS = 8192;
__device__ unsigned long long g_p2g_prof[WGS_P2G_ROWS][8];
// rows [0, R/2): plain body, by list position; [R/2, R): CPIC body over the near-collider lists (8 x R/16 rows)
#define P2G_PROF(k)                                                                                               \
    if (threadIdx.x == 0) {                                                                                       \
        if (filter != 2 && a < WGS_P2G_ROWS / 2) g_p2g_prof[a][k] = wall_clock64();                               \
        if (filter == 2 && a < WGS_P2G_ROWS / 16) g_p2g_prof[WGS_P2G_ROWS / 2 + (P2G_BLK & 7u) * (WGS_P2G_ROWS / 16) + a][k] = wall_clock64(); \
    }
#else
#define P2G_PROF(k)
#endif
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

// solver/grid_update.wgsl:55-64 for one node: (momentum, mass) sum -> (velocity, mass). One definition for the grid-update
// launch and for the grid-update waves inside the P2G launch: the same expressions, the same roundings.
template <int D> __device__ __forceinline__ float4 node_velocity(const float4 sum, const float *g, float dt, float lim) {
    const float mass = D == 3 ? sum.w : sum.z;
    const float inv_mass = mass > 0.f ? 1.0f / mass : 0.f;
    const float mom[3] = {sum.x, sum.y, sum.z};
    float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D; k++) {
        const float vel = (mom[k] + mass * g[k] * dt) * inv_mass;
        v[k] = fminf(fmaxf(vel, -lim), lim);
    }
    return D == 3 ? make_float4(v[0], v[1], v[2], mass) : make_float4(v[0], v[1], mass, 0.f);
}

// The grid update as waves INSIDE the P2G launch (k_p2g / k_p2g_pair with GU = 2; single-domain simulations):
// one wave per active block and turn, lane = node. The wave waits for the words of the (at most 2^D) slabs its nodes
// are gathered from — P2G publishes a block's word once the slab's write-through stores have been acknowledged —,
// gathers with agent-scope loads (past this XCD's L2, which may never have seen the slab) and from there on is the
// loop body of k_grid_update<D, 0>: the same sums in the same order, the velocity written back into the slabs in place
// with plain stores (their readers, the fused G2P, are another launch). These workgroups follow every P2G workgroup in
// dispatch order and wait only for P2G workgroups: nothing they wait for can be waiting for a slot of theirs.
// TWOWAY: the node impulses P2G's CPIC body left in imp_slab are gathered the same way, converted to fixed point and summed
// per body — in LDS first, then at most 16 x 6 global atomics per workgroup, as k_grid_update<D, PHASE, true> does.
// INTERIOR (one slab of a decomposition): only the blocks whose layer neither receives a neighbour's sums nor travels to
// one — the others wait for the exchange and are updated by k_grid_update<D, 3> with IFACE_ONLY set.
template <int D, bool TWOWAY = false, bool INTERIOR = false> __device__ __forceinline__ void gu_waves(const Dev &d, uint32_t epoch, uint32_t wave, uint32_t nwaves, int lane) {
    __shared__ int32_t s_body_imp[TWOWAY ? 128 : 1];
    if constexpr (TWOWAY) {
        for (uint32_t i = threadIdx.x; i < 128u; i += blockDim.x) s_body_imp[i] = 0;
        __syncthreads();
    }
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const float dt = d.sp->dt;
    const float lim = d.h / dt;
    const float g[3] = {d.sp->gravity[0], d.sp->gravity[1], d.sp->gravity[2]};
    const int l[3] = {lane & (BW - 1), (lane >> BS) & (BW - 1), D == 3 ? (lane >> (2 * BS)) : 0};
    for (uint32_t a = wave; a < B; a += nwaves) {
        const uint32_t b = d.active[a];
        if constexpr (INTERIOR) {
            int bc[3] = {0, 0, 0};
            unpack_key<D>(d.act_info[a].y, bc);
            const IfaceMasks m = iface_masks<D>(d, bc[0]);
            if ((m.recv | m.send_lo | m.send_hi) != 0u) continue;   // (wave-uniform)
        }
        const uint32_t mysrc = lane < NN ? d.act_src[a * 8u + (uint32_t)lane] : NONE;   // (k_regroup: "-" neighbour with particles, else NONE)
        if (mysrc != NONE) {
            // (bounded: ~a second. Every block the sort counted particles for is visited by a P2G workgroup of this launch,
            // which publishes its word; should the two ever disagree the wave reports it and gathers what is there — the
            // behaviour of the launch of its own — instead of hanging the device)
            uint32_t spins = 0u;
            while (__hip_atomic_load(&d.slab_epoch[mysrc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins == (1u << 20)) {
                    atomicOr(&d.counters[CTR_ERRORS], ERRBIT_HANDOVER);
                    break;
                }
            }
        }
        asm volatile("" ::: "memory");   // (no load below may be scheduled above the loop)
        // Branch-free gather: one descriptor per source slab (wave-uniform, scalar registers; NO source: zero records), a lane
        // whose node the slab does not cover asks for an offset past the end — the buffer unit returns zeros for both
        // without touching memory, and drops such stores. All 2^D loads are in flight together.
        __amdgpu_buffer_rsrc_t rs[NN];
        uint32_t off[NN];
        float4 part[NN];
#pragma unroll
        for (int o = 0; o < NN; o++) {
            const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)mysrc, o);
            const int tt[3] = {l[0] + BW * (o & 1), l[1] + BW * ((o >> 1) & 1), l[2] + BW * ((o >> 2) & 1)};
            const bool in_tile = tt[0] < TW && tt[1] < TW && (D == 2 || tt[2] < TW);
            rs[o] = slab_rsrc(&d.slab[(size_t)(src != NONE ? src : 0u) * TILE], src != NONE ? TILE * 16u : 0u);
            off[o] = in_tile ? slab_pos<D>(o, l) * 16u : 0x7ffffff0u;
            part[o] = ld_agent(rs[o], off[o]);
        }
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < NN; o++) {   // (the fixed order of k_grid_update; an absent term is +0: x + 0 == x, and no partial sum is -0)
            sum.x += part[o].x; sum.y += part[o].y; sum.z += part[o].z; sum.w += part[o].w;
        }
        if constexpr (TWOWAY) {
            constexpr int IMPQ = D == 3 ? 2 : 1;
            float isum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // node impulse: linear, angular
#pragma unroll
            for (int o = 0; o < NN; o++) {
                const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)mysrc, o);
                if (src == NONE || d.block_cpic[src] == 0u) continue;   // (wave-uniform) only the CPIC body of P2G writes impulse partials
                const __amdgpu_buffer_rsrc_t ri = slab_rsrc(&d.imp_slab[(size_t)src * TILE * IMPQ], TILE * IMPQ * 16u);
                const uint32_t io = off[o] < TILE * 16u ? off[o] * IMPQ : 0x7ffffff0u;
                const float4 a = ld_agent(ri, io);
                isum[0] += a.x; isum[1] += a.y; isum[2] += a.z;
                if constexpr (D == 3) {
                    const float4 bq = ld_agent(ri, io + 16u);
                    isum[3] += bq.x; isum[4] += bq.y; isum[5] += bq.z;
                }
            }
            // p2g.wgsl:142-155: the node's total impulse goes to its closest body, in fixed point (integer atomics: order-independent)
            const uint32_t cl = d.node_cdf[(size_t)b * NPB + (uint32_t)lane].closest_id;
            if (cl < 16u) {
                constexpr int NI = D == 3 ? 6 : 3;
#pragma unroll
                for (int k = 0; k < NI; k++) {
                    const int32_t v = flt2int(isum[k]);
                    if (v != 0) atomicAdd(&s_body_imp[cl * 8u + k], v);
                }
            }
        }
        const float4 nv = node_velocity<D>(sum, g, dt, lim);
#pragma unroll
        for (int o = 0; o < NN; o++) st_plain(rs[o], off[o], nv);
        d.nodes[b * NPB + (uint32_t)lane] = nv;
        // the sort's per-block accumulator is zero at rest (last read: k_regroup) — unless the fused G2P of this substep keeps it up
        // to date for the next one (Dev::bin_next: only the particles that change block move a unit)
        if (lane == 0 && !d.bin_next) d.block_acc[b] = 0u;
    }
    if constexpr (TWOWAY) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < 128u; i += blockDim.x) {
            const int32_t v = s_body_imp[i];
            if (v != 0) atomicAdd(&d.impulses[i], v);
        }
    }
}

// `filter`: 0 = every block; in collider simulations the pass is launched twice, CPIC = false with
// filter 1 (blocks whose tile sees no collider: every affinity is 0, plain MLS-MPM) and CPIC = true
// with filter 2 (blocks near a collider).
// TWOWAY (with CPIC): the momentum an incompatible (particle, node) pair does not transfer is accumulated
// as an impulse on the node's closest body (p2g.wgsl:200-228), per node, in a second LDS tile; the
// per-block partial node sums go to imp_slab and are gathered, converted to fixed point and added to the
// bodies by the grid update (p2g.wgsl:142-155).
// PCDF (with CPIC): the particle cdf of the block's particles (g2p_cdf.wgsl) is computed in the prologue, from the
// node cdfs k_block_setup<CDF> left in node_cdf — the third step of k_cdf without a launch of its own.
// GU (single-domain simulations): 1 = the slabs are handed over inside the launch (write-through stores + the
// block's word in slab_epoch); 2 = also, the workgroups from index `nblk` on are not P2G workgroups but run the grid
// update (gu_waves below) — they are dispatched after every P2G workgroup and gather a node as soon as the slabs that
// cover it are complete: no grid-update launch.
// `layer_sel` (one slab of a decomposition; 0 = every block): 1 = only the blocks of the BOUNDARY layers — the two block layers at each
// cut whose slabs hold what travels to a neighbour (kernels_shard.h shard_boundary_layer) —, 2 = all the others. wgs_sharded_step
// launches the boundary first, on a stream of its own together with the exchange, and the interior meanwhile (capi_sharded.inc).
// GU = 3 (one slab of a decomposition, inside wgs_sharded_step): behind the P2G workgroups first `npack` workgroups whose
// waves pack the outgoing messages (kernels_shard.h pack_face_body; `npack_blk` of their waves walk the interface-block
// list, the others copy the guests), then the grid update of the INTERIOR blocks; the interface layers are updated after
// the exchange (k_grid_update<D, 3> with iface_only).
// One wave of the prologue workgroups of a P2G launch: the particle cdf of the visit-list entries it strides over (described at k_p2g_pair).
template <int D> __device__ __forceinline__ void pcdf_waves(const Dev &d, int side, uint32_t epoch, uint32_t wave_of_list, uint32_t waves_per_list, uint32_t k, int lane, NodeCdf *tile) {
    if (!pcdf_waves_on(d, epoch, (uint32_t)P2GCfg<D>::NW)) return;   // (the lists outgrew what the host sized these waves for: the blocks' workgroups do it)
    const uint32_t nvis = min(d.counters[ctr_nvisit(k, epoch)], d.visit_cap);
    const uint2 *vl = d.visit_list + (size_t)k * d.visit_cap;
    const uint32_t nsorted = min(d.nv, d.counters[CTR_NSORTED]);
    float *buf = d.buf[side];
    uint32_t staged = NONE;
    for (uint32_t v = wave_of_list; v < nvis; v += waves_per_list) {
        const uint2 e = vl[v];
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
        // (the particle's chain — sort entry, state — is requested before the tile's — links, node cdfs: two pairs of round trips side by side.
        // A lane whose slot belongs to another block fetches a particle it will not touch.)
        const uint32_t j = e.y * 64u + (uint32_t)lane;
        uint32_t cid = NONE, src = 0u;
        if (j < nsorted) {
            cid = d.perm_cell[j];
            src = d.perm[j];
        }
        const uint32_t bkey = d.block_key[b];
        const ParticleCdfIn in = particle_cdf_fetch<D>(d, buf, src);
        if (b != staged) {
            // (single wave: its LDS accesses are served in order; the fences keep the compiler from moving the reads of the previous
            // tile below, or the reads of this one above, the stores)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            stage_node_cdf_tile<D, 64>(d, b, tile, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            staged = b;
        }
        int bc[3] = {0, 0, 0};
        unpack_key<D>(bkey, bc);
        const bool mine = cid != NONE && ((cid & ~CELL_LISTED) >> 6) == b;
        if (mine) particle_cdf_update<D, true>(d, buf, src, in, tile, bc, epoch);
        const uint32_t done = (uint32_t)__popcll(__ballot(mine));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the quads are in memory before the count says so
        if (lane == 0 && done != 0u) __hip_atomic_fetch_add(&d.pcdf_done[b], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// GUTW: the grid-update waves that ride in this launch gather the bodies' impulses (two-way coupling) although the launch's own body does not
// accumulate any — the plain launch of a large two-way simulation, run BEHIND the near-collider launch (capi.hip).
template <int D, bool CPIC, bool TWOWAY = false, bool PCDF = false, int GU = 0, bool GUTW = TWOWAY>
__global__ __launch_bounds__(P2GCfg<D>::NW * 64) void k_p2g(Dev d, int side, int filter, uint32_t epoch, uint32_t nblk, uint32_t npack, uint32_t npack_blk, uint32_t layer_sel, uint32_t npro) {
    using Cfg = P2GCfg<D>;
    constexpr int TILE = Dim<D>::TILE;
    __shared__ float4 s_tile[Cfg::NW][TILE];
    if constexpr (PCDF) {   // (prologue waves in front of the CPIC launch of the near-collider list: as in k_p2g_pair, below)
        if (blockIdx.x < npro) {
            const uint32_t w = threadIdx.x >> 6;
            pcdf_waves<D>(d, side, epoch, (blockIdx.x >> 3) * (uint32_t)Cfg::NW + w, (npro >> 3) * (uint32_t)Cfg::NW, blockIdx.x & 7u, (int)(threadIdx.x & 63u),
                          reinterpret_cast<NodeCdf *>(s_tile[w]));
            return;
        }
    }
    const uint32_t wg = blockIdx.x - (PCDF ? npro : 0u), nwg = gridDim.x - (PCDF ? npro : 0u);
    if constexpr (GU == 2) {
        if (wg >= nblk) {
            gu_waves<D, GUTW>(d, epoch, (wg - nblk) * Cfg::NW + (threadIdx.x >> 6), (nwg - nblk) * Cfg::NW, (int)(threadIdx.x & 63u));
            return;
        }
    }
    if constexpr (GU == 3) {
        if (wg >= nblk) {
            const uint32_t t = wg - nblk, w = threadIdx.x >> 6;
            const int lane = (int)(threadIdx.x & 63u);
            if (t < npack) pack_face_body<D, true>(d, side, epoch, t * Cfg::NW + w, npack_blk, npack * Cfg::NW, lane);
            else gu_waves<D, GUTW, true>(d, epoch, (t - npack) * Cfg::NW + w, (nwg - nblk - npack) * Cfg::NW, lane);
            return;
        }
    }
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW;
    constexpr int NT = Cfg::NW * 64;
    constexpr int SLOTS = P2G_J * NPB;
    constexpr int ROW = NPB + WGS_P2G_ROW_PAD;   // padded [rank] row: ds_write_b128 serves 8 contiguous lanes per cycle over 8 slots of 16 bytes (bank = dword mod 32);
                                                 // the 8 lanes hold 4 ranks of 2 cells, and 2 rank + cell takes 8 values for ROW = 2 mod 8 (+ 4: ranks 0 / 2 and 1 / 3 collided)
    constexpr int KS = (SLOTS + NT - 1) / NT;    // staging slots per thread and round
    constexpr int NQ = Cfg::NQ;
    __shared__ float4 s_q[NQ][P2G_J * ROW];
    __shared__ uint32_t s_aff[CPIC ? P2G_J * ROW : 1];
    __shared__ uint32_t s_cs[NPB], s_cn[NPB];
    constexpr int IMPQ = D == 3 ? 2 : 1;  // impulse quads per node: (lin, 0), (ang, 0) | (lin.xy, ang, 0)
    __shared__ float4 s_nrm[TWOWAY ? P2G_J * ROW : 1];
    __shared__ NodeCdf s_ncdf[PCDF ? TILE : 1];
    __shared__ float4 s_imp[TWOWAY ? Cfg::NW : 1][TWOWAY ? IMPQ : 1][TWOWAY ? TILE : 1];
    __shared__ ColliderMotion s_colm[TWOWAY ? 16 : 1];   // (p2g_body.inc: the colliders' motion for the two-way impulses)

#define P2G_CPIC CPIC
#define P2G_TWOWAY TWOWAY
#define P2G_PCDF PCDF
#define P2G_HANDOVER (GU != 0)
#define P2G_GUESTS_INLAUNCH (GU == 3)
#define P2G_BLK wg
#define P2G_NBLK nblk
#include "p2g_body.inc"
#undef P2G_CPIC
#undef P2G_TWOWAY
#undef P2G_PCDF
#undef P2G_BLK
#undef P2G_NBLK
#undef P2G_HANDOVER
#undef P2G_GUESTS_INLAUNCH
}

// Scenes with MANY blocks near colliders: the plain body (filter 1) and the CPIC body over the near-collider list
// (filter 2, particle cdf prologue included) in ONE launch; the first half of the grid runs one, the second half the
// other. The kernel takes the CPIC body's registers, so the plain body loses a third of its occupancy (36 -> 48 us at
// C2): a loss while the list is short — the two launches then simply follow each other —, a gain once the CPIC launch
// is the longer of the two, because a near-collider block costs ~3x a plain one in latency and the launches no longer
// add up. capi.hip switches on the list length the host last saw.
// WPE = waves per SIMD the register budget is cut for. 1 = whatever the CPIC body wants (203 VGPRs one-way: fastest
// per near-collider block, but the plain body then runs at 2 waves per SIMD); 3 = 168 VGPRs: the CPIC body spills
// 116 B and each of its blocks takes ~20 % longer, the plain body keeps its occupancy. The second wins when the plain
// half is the long one or the list is long enough to be a throughput problem (1 M sand on the floor: P2G 54 -> 44 us,
// 4 M sand between walls 259 -> 207 us), the first on small scenes with a short list (262 k cube on a heightfield:
// 26 vs 31 us). The two budgets differ in the last bit here and there (another instruction selection), so the small
// one is not chosen from the moment's list length: capi.hip runs it ALWAYS for one-way simulations from 600 k particles
// on — with the plain body at full occupancy the pair costs nothing while the list is empty —, and smaller ones keep
// the unconstrained body, whose paired and separate forms are bit-identical. The two-way body (256 VGPRs) is not
// offered the small budget.
// (GU: as for k_p2g; the grid is `npro` prologue + `half` CPIC + `half` plain workgroups, then the grid-update workgroups)
//
// PROLOGUE WAVES (`npro` > 0, Dev::pcdf_waves set): the particle cdf of the listed blocks (g2p_cdf.wgsl) is 3-6 rounds of ~4 us in the
// CPIC workgroup of a block — 192 threads for 512 particles — and that workgroup's accumulation cannot start before the last of them;
// with few listed blocks (the reference's sand3: a dozen of 450) the launch waited for those chains while most of the chip was idle.
// The first `npro` workgroups take the cdf instead, one WAVE per visit-list entry (a listed block's share of a chunk of 64 sorted
// particles — the list the fused G2P walks): node-cdf tile of the block into the wave's LDS tile, the particles' quads written through
// (agent scope), the count of finished particles added to Dev::pcdf_done[block] once the stores are acknowledged. The block's CPIC
// workgroup — a higher workgroup index: dispatched after every prologue workgroup, so the wait cannot deadlock — polls that word
// until it equals the block's particle total and fetches the affinities with agent-scope loads (p2g_body.inc). No fence anywhere.
template <int D, bool TWOWAY, int WPE = 1, int GU = 0>
__global__ __launch_bounds__(P2GCfg<D>::NW * 64, WPE) void k_p2g_pair(Dev d, int side, uint32_t epoch, uint32_t half, uint32_t npack, uint32_t npack_blk, uint32_t layer_sel, uint32_t npro) {
    using Cfg = P2GCfg<D>;
    constexpr int TILE = Dim<D>::TILE;
    __shared__ float4 s_tile[Cfg::NW][TILE];
    if (blockIdx.x < npro) {   // (npro is a multiple of 8: list k = the workgroup's XCD)
        static_assert(sizeof(NodeCdf) == sizeof(float4), "a wave's accumulation tile doubles as its node-cdf tile");
        const uint32_t w = threadIdx.x >> 6;
        pcdf_waves<D>(d, side, epoch, (blockIdx.x >> 3) * (uint32_t)Cfg::NW + w, (npro >> 3) * (uint32_t)Cfg::NW, blockIdx.x & 7u, (int)(threadIdx.x & 63u),
                      reinterpret_cast<NodeCdf *>(s_tile[w]));
        return;
    }
    const uint32_t wg = blockIdx.x - npro, nwg = gridDim.x - npro;   // (everything below counts from the first CPIC workgroup)
    if constexpr (GU == 2) {
        if (wg >= 2u * half) {
            gu_waves<D, TWOWAY>(d, epoch, (wg - 2u * half) * Cfg::NW + (threadIdx.x >> 6), (nwg - 2u * half) * Cfg::NW, (int)(threadIdx.x & 63u));
            return;
        }
    }
    if constexpr (GU == 3) {
        if (wg >= 2u * half) {
            const uint32_t t = wg - 2u * half, w = threadIdx.x >> 6;
            const int lane = (int)(threadIdx.x & 63u);
            if (t < npack) pack_face_body<D, true>(d, side, epoch, t * Cfg::NW + w, npack_blk, npack * Cfg::NW, lane);
            else gu_waves<D, TWOWAY, true>(d, epoch, (t - npack) * Cfg::NW + w, (nwg - 2u * half - npack) * Cfg::NW, lane);
            return;
        }
    }
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW;
    constexpr int NT = Cfg::NW * 64;
    constexpr int SLOTS = P2G_J * NPB;
    constexpr int ROW = NPB + WGS_P2G_ROW_PAD;
    constexpr int KS = (SLOTS + NT - 1) / NT;
    constexpr int NQ = Cfg::NQ;
    __shared__ float4 s_q[NQ][P2G_J * ROW];
    __shared__ uint32_t s_aff[P2G_J * ROW];
    __shared__ uint32_t s_cs[NPB], s_cn[NPB];
    constexpr int IMPQ = D == 3 ? 2 : 1;
    __shared__ float4 s_nrm[TWOWAY ? P2G_J * ROW : 1];
    __shared__ NodeCdf s_ncdf[TILE];
    __shared__ float4 s_imp[TWOWAY ? Cfg::NW : 1][TWOWAY ? IMPQ : 1][TWOWAY ? TILE : 1];
    __shared__ ColliderMotion s_colm[TWOWAY ? 16 : 1];   // (p2g_body.inc: the colliders' motion for the two-way impulses)
#define P2G_HANDOVER (GU != 0)
#define P2G_GUESTS_INLAUNCH (GU == 3)
#define P2G_NBLK half
    if (wg >= half) {
        const int filter = 1;
#define P2G_CPIC false
#define P2G_TWOWAY false
#define P2G_PCDF false
#define P2G_BLK (wg - half)
#include "p2g_body.inc"
#undef P2G_CPIC
#undef P2G_TWOWAY
#undef P2G_PCDF
#undef P2G_BLK
    } else {  // the long bodies start first
        const int filter = 2;
#define P2G_CPIC true
#define P2G_TWOWAY TWOWAY
#define P2G_PCDF true
#define P2G_BLK wg
#include "p2g_body.inc"
#undef P2G_CPIC
#undef P2G_TWOWAY
#undef P2G_PCDF
#undef P2G_BLK
    }
#undef P2G_NBLK
#undef P2G_HANDOVER
#undef P2G_GUESTS_INLAUNCH
}

// ------------------------------------------------------------ grid update
// Gather of the slabs covering each node + solver/grid_update.wgsl:55-64.
// PHASE 0: gather + update in one pass (single GPU). Sharded runs: PHASE 3 = the same single pass, which also adds the
// neighbours' partial sums of the interface layers, straight from the inbound messages (kernels_shard.h).

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
        const float4 p = d.slab[(size_t)src * TILE + slab_pos<D>(o, l)];
        sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
    }
    return sum;
}

template <int D, int PHASE, bool TWOWAY = false> __global__ __launch_bounds__(256) void k_grid_update(Dev d, uint32_t epoch, uint32_t iface_only) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const uint32_t total = B * NPB;
    const float dt = d.sp->dt;
    const float lim = d.h / dt;
    float g[3] = {d.sp->gravity[0], d.sp->gravity[1], d.sp->gravity[2]};
    if (d.sharded && blockIdx.x == 0 && threadIdx.x == 0) d.counters[CTR_NLEAVE] = 0;  // list of the coming G2P launch
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
        if constexpr (PHASE == 3) {
            if (iface_only) {   // the interior blocks were updated by waves of the P2G launch (gu_waves<.., INTERIOR>)
                int kc[3] = {0, 0, 0};
                unpack_key<D>(d.block_key[b], kc);
                const IfaceMasks im = iface_masks<D>(d, kc[0]);
                if ((im.recv | im.send_lo | im.send_hi) == 0u) continue;   // (wave-uniform: a wave is one block)
            }
        }
        if (ln == 0u && !d.bin_next) d.block_acc[b] = 0u;  // the sort's per-block accumulator is zero at rest (last read: k_regroup; see gu_waves)
        int l[3];
        l[0] = ln & (BW - 1);
        l[1] = (ln >> BS) & (BW - 1);
        l[2] = D == 3 ? (ln >> (2 * BS)) : 0;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        // PHASE 3 (sharded runs): pairs of the interface layers that a neighbour's particles reach get the neighbour's partial
        // sum added to this rank's own gather (a + b == b + a bitwise: both ranks hold the same total), found in the inbound
        // message's record table (kernels_shard.h rec_find)
        float4 recv = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PHASE == 3) {
            int bc[3] = {0, 0, 0};
            unpack_key<D>(d.block_key[b], bc);
            const uint32_t rmask = iface_masks<D>(d, bc[0]).recv;  // wave-uniform
            if (rmask != 0u) {
                int tag, q;
                halo_slot<D>(ln, tag, q);
                const int face = iface_recv_face(d, bc[0]);
                uint32_t mine = NONE;
#pragma unroll
                for (int tt = 0; tt < HaloCfg<D>::NTAG; tt++) {
                    if (!((rmask >> tt) & 1u)) continue;  // wave-uniform: the wave is one block
                    const uint32_t slot = rec_find_wave<D>(d, face, d.block_key[b], (uint32_t)tt, epoch, (int)(threadIdx.x & 63u));
                    if (tt == tag) mine = slot;
                }
                if (mine != NONE) recv = msg_halo<D>(d.msg.in[face])[(size_t)mine * HaloCfg<D>::REC_F4 + 1 + q];
            }
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
            const uint32_t src = d.act_src[(t >> 6) * 8u + o];  // (k_regroup: "-" neighbour with particles, else NONE)
            if (src == NONE) continue;
            int ti = (int)slab_pos<D>(o, l);   // (position in the source slab: region o, layout.h)
            srcs[o] = src;
            tis[o] = ti;
            {
                float4 p = d.slab[(size_t)src * TILE + ti];
                sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
            }
            if constexpr (TWOWAY) {
                // (sharded runs: also for the interface nodes — the impulses of a rank's OWN particles; the ranks'
                // fixed-point sums are reduced before integrate_bodies)
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
        if constexpr (PHASE == 3) {
            sum.x += recv.x; sum.y += recv.y; sum.z += recv.z; sum.w += recv.w;
        }
        const float4 nv = node_velocity<D>(sum, g, dt, lim);
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
#ifdef WGS_ABLATE
// stage clocks of the fused G2P main body (tools/gpu_g2p_prof.py): one row per chunk of 64 sorted particles
constexpr int WGS_G2P_ROWS = 16384;
__device__ unsigned long long g_g2p_prof[WGS_G2P_ROWS][8];
// rows [0, R/2): main body, by chunk; [R/2, 3R/4): visits of the list walk (CMODE 2), by position in the visit list;
// [3R/4, R): one row per list wave — [0] start, [5] end, [1] chunks visited
#define G2P_PROF(k)                                                                                                   \
    if (threadIdx.x == 0) {                                                                                           \
        if (G2P_CMODE != 2 && chunk < WGS_G2P_ROWS / 2) g_g2p_prof[chunk][k] = wall_clock64();                        \
        if (G2P_CMODE == 2 && run * npass + (uint32_t)pass < WGS_G2P_ROWS / 4)                                        \
            g_g2p_prof[WGS_G2P_ROWS / 2 + run * npass + (uint32_t)pass][k] = wall_clock64();                          \
    }
#define G2P_PROF_LIST(k, v) if (threadIdx.x == 0 && G2P_CMODE == 2 && blockIdx.x < WGS_G2P_ROWS / 4) g_g2p_prof[WGS_G2P_ROWS * 3 / 4 + blockIdx.x][k] = (v);
#else
#define G2P_PROF(k)
#define G2P_PROF_LIST(k, v)
#endif
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
#define G2P_DONE continue;
#define G2P_BIN BIN
// Chunks of 64 sorted particles per wave of the main body (template parameter NPASS, g2p_body.inc): 1 while the two particle
// buffers fit the 256 MB Infinity Cache (the launch then runs at HBM-roofline speed for its real traffic and a longer
// wave life only costs), 2 beyond — there the kernel is bound by latency x occupancy and twice the bytes in flight per
// wave buy 12-15 % (4.1 M particles: 184 -> 161 us, 16 M: 761 -> 647 us; 1 M: 39.0 -> 41.7 us). Same results either way.
#ifndef WGS_G2P_TWO_PASS_MIN
#define WGS_G2P_TWO_PASS_MIN 1500000
#endif
constexpr uint32_t G2P_TWO_PASS_MIN_PARTICLES = WGS_G2P_TWO_PASS_MIN;
#ifndef WGS_G2P_LIST_PASSES
#define WGS_G2P_LIST_PASSES 1
#endif
#ifndef WGS_G2P_MANY_PASSES
#define WGS_G2P_MANY_PASSES 4
#endif
#ifndef WGS_G2P_MANY_PASS_MIN
#define WGS_G2P_MANY_PASS_MIN 2500000
#endif
constexpr int G2P_MANY_PASSES = WGS_G2P_MANY_PASSES;
constexpr uint32_t G2P_MANY_PASS_MIN_PARTICLES = WGS_G2P_MANY_PASS_MIN;
// BIN: the launch can also bin its output for the next substep (g2p_body.inc, Dev::bin_next; on a slab the arrivals are binned by k_g2p_arrivals). A template parameter:
// the binning is some 350 instructions per body, and the plastic paired kernel — 8 000 instructions, beyond the instruction cache —
// runs a third slower with them compiled in whether they execute or not (C3: fused G2P 320 -> 425 us); those variants leave the
// binning to launch 1 of the sort (capi.hip).
template <int D, int MODEL, bool PLASTIC, int CMODE, int NPASS = 1, bool SHARD = false, bool BIN = false>
__global__ __launch_bounds__(G2P_THREADS, G2P_WAVES_PER_EU) void k_g2p_update(Dev d, int side, uint32_t epoch) {
    constexpr uint32_t npass = NPASS;  // (a template parameter: as a kernel argument the second pass's registers spilled in the one-pass launch)
    __shared__ float4 s_node[Dim<D>::TILE];
    __shared__ uint32_t s_nbr[16];   // Dev::bin_next: ids of the staged block's 2^D "+" and 2^D "-" neighbours (nbr_known)
    __shared__ NodeCdf s_cdf[CMODE == 2 ? Dim<D>::TILE : 1];
    __shared__ ColliderMotion s_col[CMODE == 2 ? 16 : 1];   // (CPIC body: the colliders' velocities and centres of mass)
#define G2P_CMODE CMODE
#define G2P_BX blockIdx.x
#define G2P_GX gridDim.x
#include "g2p_body.inc"
#undef G2P_CMODE
#undef G2P_BX
#undef G2P_GX
}

// Collider simulations: the body of CMODE 2 (walk of the near-collider block list) and the body of CMODE 1 (blocks
// away from colliders, one workgroup per 64 sorted particles) in ONE launch. The first 8 x `nlist` workgroups walk
// the list — they start at once and run beside the others —, the `nmain` workgroups after them run the main body;
// 8 * nlist and nmain are multiples of 8, so the XCD-aware chunk mapping of the main body is unchanged. Both bodies
// are capped at the same register budget, so the main path keeps its occupancy; the launch saves the ~4.5 us a
// dependent kernel boundary costs even when the list is empty. `nlist` follows the list length the host last saw
// (wgs_sync), so an empty list costs a few hundred workgroups that exit at once.
// WPE (waves per SIMD the register budget is cut for): 3 everywhere (168 VGPRs; the corotated Drucker-Prager pair keeps 20 B of
// scratch per lane under that cap since round 5 — it was 312 B, and every reload from scratch waits for the acknowledgement of every
// store in flight: NOTES.md —, the neo-Hookean + plastic pair 92-116 B); with 2 (256 VGPRs, nothing spilled) the list half runs faster
// and the main half slower (lower occupancy). capi.hip picks 2 when half of the active blocks or more are listed, as of the host's
// last look.
template <int D, int MODEL, bool PLASTIC, int WPE = G2P_WAVES_PER_EU, int NPASS = 1, bool SHARD = false, bool BIN = false>
__global__ __launch_bounds__(G2P_THREADS, WPE) void k_g2p_pair(Dev d, int side, uint32_t epoch, uint32_t nmain, uint32_t nlist) {
    constexpr uint32_t npass = NPASS;
    __shared__ float4 s_node[Dim<D>::TILE];
    __shared__ uint32_t s_nbr[16];   // Dev::bin_next: ids of the staged block's 2^D "+" and 2^D "-" neighbours (nbr_known)
    __shared__ NodeCdf s_cdf[Dim<D>::TILE];
    __shared__ ColliderMotion s_col[16];   // (CPIC body: the colliders' velocities and centres of mass)
    if (blockIdx.x >= 8u * nlist) {
        const uint32_t widx = blockIdx.x - 8u * nlist;
#define G2P_CMODE 1
#define G2P_BX widx
#define G2P_GX nmain
#include "g2p_body.inc"
#undef G2P_CMODE
#undef G2P_BX
#undef G2P_GX
    } else {  // the 8 * nlist waves of the visit list
        // (one visit at a time, WGS_G2P_LIST_PASSES: the CPIC body has no registers left for the main body's prefetch — with two entries
        // per run the next visit's quads went through scratch, 12-20 dwords per visit; with four the non-plastic pair lost 5 % at 16 M)
        constexpr uint32_t npass = NPASS < WGS_G2P_LIST_PASSES ? NPASS : WGS_G2P_LIST_PASSES;
#define G2P_CMODE 2
#define G2P_BX blockIdx.x
#define G2P_GX (8u * nlist)
#include "g2p_body.inc"
#undef G2P_CMODE
#undef G2P_BX
#undef G2P_GX
    }
}
#undef G2P_DONE
#undef G2P_BIN

}  // namespace wgs
