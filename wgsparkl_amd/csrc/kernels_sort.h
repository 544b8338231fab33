// kernels_sort.h — "grid sort" pass: sparse-grid activation and the counting
// sort of particles by (block, cell, particle id).
//
// Replaces the reference's 14 dispatches of WgGrid::queue_sort
// (grid/grid.rs:30-207; kernels grid.wgsl:186-203,355-379, sort.wgsl:26-36,89-137,
// prefix_sum.wgsl:11-93). Same observable result — the set of active blocks (a
// particle's block and its "+1" neighbours), per-block first_particle /
// num_particles, particle ids grouped by block — with a different mechanism:
//   * the hash map and the block ids PERSIST across substeps; a per-block epoch stamp
//     says which blocks are active now. Touching an already-known block is a plain
//     L2-served lookup plus an idempotent store; device-scope atomics (memory-side on
//     MI355X) are only issued for blocks never seen before;
//   * one pass over the particles does both the activation and the counting
//     (reference: touch_particle_blocks + update_block_particle_count + finalize);
//   * counting uses LDS histograms per (wave, block) and one coalesced returning atomic
//     per (wave, block) instead of one global atomic per particle;
//   * no per-node linked lists: cells become contiguous ranges of `perm`, and the order
//     inside a cell is canonical (ascending persistent particle id), so every downstream
//     fp32 sum is reproducible run to run (reference: atomic race order, sort.wgsl:126,133).
#pragma once
#include "kernels_cdf.h"

namespace wgs {

constexpr int SORT_THREADS = 256;
constexpr int TOUCH_SET = 32;  // distinct blocks a workgroup can de-duplicate in LDS

template <int D> __device__ inline void load_cell(const float *in, uint32_t npad, uint32_t i, float h, int *cell) {
    const float4 xm = ldq(in, npad, Pl<D>::XM, i);
    cell[0] = assoc_cell(xm.x, h);
    cell[1] = assoc_cell(xm.y, h);
    if constexpr (D == 3) cell[2] = assoc_cell(xm.z, h);
}

// Per-cell counting of one wave's particles (sort.wgsl:89-99 extended to cells). Scattered device-scope
// atomics run at the memory side on MI355X (~20 G/s when every lane hits its own line), so: LDS histogram
// per (wave, block), lanes get their rank inside the wave's group from an LDS atomic, and ONE coalesced
// returning global atomic per (wave, block) reserves the group's range inside each cell. The arrival
// order of those atomics (and the LDS arbitration order) leaks into `rank`; k_canonical_order sorts each
// cell by particle id afterwards. Wave-uniform control flow: call with all 64 lanes.
__device__ inline void count_cells(const Dev &d, uint32_t *hist, int lane, uint32_t myid, uint32_t local, uint32_t &cid, uint32_t &rank) {
    unsigned long long todo = __ballot(myid != NONE);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t id0 = __shfl(myid, leader);
        const bool mine = myid == id0;
        const unsigned long long same = __ballot(mine);
        todo &= ~same;
        if (lane == leader) atomicAdd(&d.block_acc[id0], (uint32_t)__popcll(same));
        // Cross-lane traffic through LDS inside one wave uses (relaxed, wavefront-scope) atomic
        // accesses so the compiler may not forward this lane's own stores to its loads.
        __hip_atomic_store(&hist[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        uint32_t r = 0;
        if (mine) r = __hip_atomic_fetch_add(&hist[local], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t cnt = __hip_atomic_load(&hist[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        uint32_t base = 0;
        if (cnt) base = atomicAdd(&d.cell_count[id0 * NPB + lane], cnt);
        __hip_atomic_store(&hist[lane], base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (mine) {
            cid = id0 * NPB + local;
            rank = __hip_atomic_load(&hist[local], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + r;
        }
    }
}

// sort.wgsl:26-36 touch_particle_blocks + sort.wgsl:89-99 update_block_particle_count, fused.
// `tail`: sharded steady state — only the particles that arrived from the neighbours, slots [NPREV, N); the
// residents go through k_rebin. `tail` = 2 also APPENDS them first: thread r copies record r of the two inbound
// migration messages into slot NPREV + r and does the bookkeeping of the migration round (one launch instead of
// append + bin).
struct MigIn {
    const float *in_lo, *in_hi, *out_lo, *out_hi;
    uint32_t cap;
};
template <int D, int TAIL> __global__ __launch_bounds__(SORT_THREADS) void k_bin(Dev d, int side, uint32_t epoch, MigIn mig) {
    constexpr int tail = TAIL;  // 0 = every slot, 1 = the arrivals (already appended), 2 = append + bin the arrivals
    constexpr int BS = Dim<D>::BSHIFT, BW = Dim<D>::BW, NN = Dim<D>::NNBR;
    __shared__ uint32_t s_keys[TOUCH_SET], s_ids[TOUCH_SET];
    __shared__ uint32_t s_hist[SORT_THREADS / 64][NPB];
    const float *in = d.buf[side];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < TOUCH_SET) s_keys[tid] = NONE;
    __syncthreads();
    const uint32_t first = tail ? d.counters[CTR_NPREV] : 0u;
    const uint32_t i = first + blockIdx.x * SORT_THREADS + tid;
    uint32_t slots_end = num_slots(d);
    if constexpr (TAIL == 2) {
        constexpr int NQ = Pl<D>::NQ, RF = Pl<D>::NQ * 4 + 2;  // record = quads, pid, cdf epoch (kernels_shard.h)
        auto cnt = [&](const float *b) { return b ? min(reinterpret_cast<const uint32_t *>(b)[0], mig.cap) : 0u; };
        const uint32_t n_lo = cnt(mig.in_lo), n_hi = cnt(mig.in_hi);
        const uint32_t arrivals = min(n_lo + n_hi, d.n - first);  // d.n = allocated capacity in sharded mode
        if (n_lo + n_hi > arrivals && blockIdx.x == 0 && tid == 0) atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);
        slots_end = first + arrivals;
        if (blockIdx.x == 0 && tid == 0) {  // nothing else in this launch reads these two
            d.counters[CTR_N] = slots_end;
            d.counters[CTR_NV] = first - cnt(mig.out_lo) - cnt(mig.out_hi) + arrivals;
        }
        const uint32_t r = i - first;
        if (r < arrivals) {
            const float *rec = r < n_lo ? mig.in_lo + 4 + (size_t)r * RF : mig.in_hi + 4 + (size_t)(r - n_lo) * RF;
            float *buf = d.buf[side];
#pragma unroll
            for (int q = 0; q < NQ; q++) stq(buf, d.npad, q, i, make_float4(rec[q * 4], rec[q * 4 + 1], rec[q * 4 + 2], rec[q * 4 + 3]));
            stpid<D>(buf, d.npad, i, __float_as_uint(rec[NQ * 4]));
            ststamp<D>(buf, d.npad, i, __float_as_uint(rec[NQ * 4 + 1]));
        }
    }
    bool valid = i < slots_end;
    if (d.sharded && valid) valid = ldpid<D>(in, d.npad, i) != 0xffffffffu;  // slot vacated by a migrated particle
    int b[3] = {0, 0, 0};
    uint32_t key = NONE, local = 0;
    if (valid) {
        int c[D];
        load_cell<D>(in, d.npad, i, d.h, c);
        uint32_t shift = 0;
#pragma unroll
        for (int k = 0; k < D; k++) {
            b[k] = c[k] >> BS;                               // floor(cell / BW), grid.wgsl:284-292
            local |= (uint32_t)(c[k] & (BW - 1)) << shift;   // grid.wgsl:346-348 node_id
            shift += BS;
        }
        int hi[D];  // the block and its +1 neighbours must all be representable (grid.wgsl:88-95)
#pragma unroll
        for (int k = 0; k < D; k++) hi[k] = b[k] + 1;
        if (!block_in_key_range<D>(b) || !block_in_key_range<D>(hi)) {
            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_KEYRANGE);
            valid = false;
        } else {
            key = pack_key<D>(b);
        }
    }
    // ---- 1. distinct blocks of the workgroup -> LDS set (1-3 entries on block-sorted input)
    uint32_t myslot = NONE, mydirect = NONE;
    unsigned long long todo = __ballot(valid);
    while (todo) {  // wave-uniform: one iteration per distinct block in the wave
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = __shfl(key, leader);
        const bool mine = valid && key == k0;
        todo &= ~__ballot(mine);
        uint32_t slot = NONE, direct = NONE;
        if (lane == leader) {
            uint32_t s = hash_key(k0) & (TOUCH_SET - 1);
            for (int probe = 0; probe < TOUCH_SET; probe++) {
                const uint32_t old = atomicCAS(&s_keys[s], NONE, k0);
                if (old == NONE || old == k0) { slot = s; break; }
                s = (s + 1) & (TOUCH_SET - 1);
            }
            if (slot == NONE) {  // set overflow (unsorted input): activate directly
                for (int o = 0; o < NN; o++) {
                    int nb[3] = {b[0] + (o & 1), b[1] + ((o >> 1) & 1), b[2] + ((o >> 2) & 1)};
                    const uint32_t id = activate_block(d, pack_key<D>(nb), epoch);
                    if (o == 0) direct = id;
                }
            }
        }
        slot = __shfl(slot, leader);
        direct = __shfl(direct, leader);
        if (mine) { myslot = slot; mydirect = direct; }
    }
    __syncthreads();
    // ---- 2. grid.wgsl:300-320: the 2^D blocks {b, b+1} per axis of every distinct block
    {
        const int slot = tid >> 3, o = tid & 7;
        const uint32_t k0 = s_keys[slot];
        if (k0 != NONE && o < NN) {
            int lb[3] = {0, 0, 0};
            unpack_key<D>(k0, lb);
            int nb[3] = {lb[0] + (o & 1), lb[1] + ((o >> 1) & 1), lb[2] + ((o >> 2) & 1)};
            const uint32_t id = activate_block(d, pack_key<D>(nb), epoch);
            if (o == 0) s_ids[slot] = id;
        }
    }
    __syncthreads();
    const uint32_t myid = !valid ? NONE : (myslot != NONE ? s_ids[myslot] : mydirect);
    // ---- 3. count per cell
    uint32_t cid = NONE, rank = 0;
    count_cells(d, s_hist[wave], lane, myid, local, cid, rank);
    if (i < slots_end) {
        d.cellid[i] = cid;
        d.rank[i] = rank;
    }
}

// Steady-state binning (sort.wgsl:26-36,89-99 for a buffer that is the sorted output of the previous
// substep): slot i held cell perm_cell[i] of block b = perm_cell[i] >> 6 one substep ago, particles move
// less than a cell per substep, so almost every particle is still in block b: its new cell id is
// b * 64 + new local cell and the blocks to activate are b's neighbour links of the previous substep —
// no hash lookup, no LDS set. Only particles that changed block go through the hash map.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_rebin(Dev d, int side, uint32_t epoch) {
    constexpr int BS = Dim<D>::BSHIFT, BW = Dim<D>::BW, NN = Dim<D>::NNBR;
    __shared__ uint32_t s_hist[SORT_THREADS / 64][NPB];
    const float *in = d.buf[side];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i = blockIdx.x * SORT_THREADS + tid;
    // sharded runs: the residents only (arrivals have no previous cell: k_bin's tail pass), minus the slots
    // vacated by particles that migrated away
    const bool in_range = i < (d.sharded ? min(d.counters[CTR_NPREV], d.counters[CTR_N]) : num_slots(d));
    bool valid = in_range;
    if (d.sharded && valid) valid = ldpid<D>(in, d.npad, i) != 0xffffffffu;
    uint32_t myid = NONE, local = 0;
    if (valid) {
        const uint32_t old = d.perm_cell[i];  // NONE only after a grid overflow: take the hash path then
        const uint32_t ob = old >> 6;
        const uint32_t okey = old != NONE ? d.block_key[ob] : 0u;
        int c[D], nb[3] = {0, 0, 0};
        load_cell<D>(in, d.npad, i, d.h, c);
        uint32_t shift = 0;
#pragma unroll
        for (int k = 0; k < D; k++) {
            nb[k] = c[k] >> BS;
            local |= (uint32_t)(c[k] & (BW - 1)) << shift;
            shift += BS;
        }
        int hi[3] = {nb[0] + 1, nb[1] + 1, nb[2] + 1};
        if (!block_in_key_range<D>(nb) || !block_in_key_range<D>(hi)) {
            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_KEYRANGE);
        } else {
            const uint32_t key = pack_key<D>(nb);
            myid = (old != NONE && key == okey) ? ob : activate_block(d, key, epoch);  // few particles change block
        }
    }
    // activate every distinct block of the wave and its +1 neighbours (grid.wgsl:300-320)
    unsigned long long todo = __ballot(myid != NONE);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t b1 = __shfl(myid, leader);
        todo &= ~__ballot(myid == b1);
        if (lane < NN) {
            // links of the previous substep when they exist: a plain idempotent store. A block created just now,
            // re-activated after a pause, or whose neighbour was not active (it held no particle) goes
            // through the hash map.
            const uint32_t le = d.links_epoch[b1], link = d.nbr_plus[b1 * 8u + lane];  // independent loads
            const uint32_t t1 = le == epoch - 1u ? link : NONE;
            if (t1 != NONE) {
                d.block_stamp[t1] = epoch;
            } else {
                int kb[3] = {0, 0, 0};
                unpack_key<D>(d.block_key[b1], kb);
                int nb[3] = {kb[0] + (lane & 1), kb[1] + ((lane >> 1) & 1), kb[2] + ((lane >> 2) & 1)};
                if (block_in_key_range<D>(nb)) activate_block(d, pack_key<D>(nb), epoch);
            }
        }
    }
    uint32_t cid = NONE, rank = 0;
    count_cells(d, s_hist[wave], lane, myid, local, cid, rank);
    if (in_range) {  // (a vacated slot gets NONE: k_scatter skips it)
        d.cellid[i] = cid;
        d.rank[i] = rank;
    }
}

// Active list + first_particle: one pass over the known blocks (physical ids).
//   active[a] = id of the a-th block stamped with the current epoch   (grid.wgsl:323-334's
//               active_blocks list, in physical-id order)
//   block_start[id] = exclusive scan of the particle counts           (sort.wgsl:101-115 + prefix_sum.wgsl)
// One workgroup; the number of known blocks is at most a few hundred thousand.
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 4;
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_active(Dev d, uint32_t epoch) {
    __shared__ unsigned long long wave_sums[SCAN_THREADS / 64];
    __shared__ unsigned long long carry_s;
    const uint32_t nphys = min(d.counters[CTR_NPHYS], d.cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < nphys; base += SCAN_THREADS * SCAN_ITEMS) {
        // packed scan: high word = number of active blocks, low word = particles
        unsigned long long v[SCAN_ITEMS];
        unsigned long long sum = 0ull;
        const uint32_t first = base + (uint32_t)tid * SCAN_ITEMS;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            const uint32_t id = first + k;
            const uint32_t idc = min(id, nphys - 1u);  // both loads unconditional: one round trip, not two
            const uint32_t stamp = d.block_stamp[idc], acc = d.block_acc[idc];
            const bool act = id < nphys && stamp == epoch;
            v[k] = act ? ((1ull << 32) | (unsigned long long)acc) : 0ull;
            sum += v[k];
        }
        unsigned long long inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        unsigned long long wave_off = 0ull;
        for (int w = 0; w < wave; w++) wave_off += wave_sums[w];
        unsigned long long run = carry_s + wave_off + inc - sum;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (v[k]) {
                const uint32_t id = first + k;
                d.active[(uint32_t)(run >> 32)] = id;
                d.block_start[id] = (uint32_t)run;
            }
            run += v[k];
        }
        __syncthreads();
        if (tid == SCAN_THREADS - 1) carry_s = run;
        __syncthreads();
    }
    if (tid < 4 && d.hdr_clear[tid]) d.hdr_clear[tid][0] = 0u;  // outgoing halo / migrant message counts of this substep
    if (tid == 0) {
        d.counters[CTR_NBLOCKS] = (uint32_t)(carry_s >> 32);
        d.counters[CTR_NCPIC] = 0;
    }
}

// Per active block, one wave: neighbour links (replaces the per-thread hash lookups of
// p2g.wgsl:238-275 / g2p.wgsl:72-132), per-cell offsets, and reset of the accumulators.
// CDF (collider simulations without mesh colliders): also the node cdf of the block's (BW+2)^D tile and the class of
// the block (see k_cdf, whose steps 1 and 2 these are; step 3 then runs in the prologue of the CPIC P2G launch), so
// that a collider simulation needs no CDF launch of its own.
template <int D, bool CDF> __device__ __forceinline__ void block_setup_body(const Dev &d, uint32_t epoch, uint32_t wg, uint32_t nwg) {
    constexpr int NN = Dim<D>::NNBR;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (wg * SORT_THREADS + threadIdx.x) >> 6;
    const uint32_t nwaves = (nwg * SORT_THREADS) >> 6;
    for (uint32_t a = wave; a < B; a += nwaves) {
        const uint32_t id = d.active[a];
        uint32_t res = NONE;
        int b[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[id], b);
        if (lane < 16) {
            const uint32_t o = lane & 7u;
            const bool minus = lane >= 8;
            if ((int)o < NN) {
                const int sgn = minus ? -1 : 1;
                int nb[3] = {b[0] + sgn * (int)(o & 1u), b[1] + sgn * (int)((o >> 1) & 1u), b[2] + sgn * (int)((o >> 2) & 1u)};
                if (block_in_key_range<D>(nb)) res = hmap_find(d, pack_key<D>(nb), epoch);
            }
            (minus ? d.nbr_minus : d.nbr_plus)[id * 8u + o] = res;
        }
        const uint32_t idx = id * NPB + lane;
        const uint32_t cnt = d.cell_count[idx];
        uint32_t inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        const uint32_t start = d.block_start[id] + inc - cnt;
        d.cell_start[idx] = start;
        d.cell_cursor[idx] = start + cnt;  // cell end (cell_count itself is cleared by k_grid_update: the scatter
                                           // half of this launch still reads it)
        if (d.n_rigid != 0u) {             // mesh-collider cdf accumulators of this substep (k_p2g_cdf)
            d.mesh_min[idx] = ~0ull;
            d.mesh_aff[idx] = 0u;
        }
        if (lane == 63) {
            d.block_count[id] = inc;       // snapshot used by P2G / grid update / G2P
            d.links_epoch[id] = epoch;     // the neighbour links written above are those of this substep
            d.block_acc[id] = 0;
            d.block_cdf_flag[id] = 0;
        }
        if constexpr (CDF) {
            constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE;
            uint32_t mine = 0u;
            // Quick reject, wave-uniform: a collider whose boundary is farther from the tile's centre than the tile's
            // half diagonal plus the affinity reach (1.5 h per axis) touches none of its nodes, and a centre outside
            // the shape then means every node is outside. One projection per collider instead of (BW+2)^D.
            uint32_t near = 0u;  // bit i: collider i can reach a node of this tile
            {
                float ctr[D];
#pragma unroll
                for (int k = 0; k < D; k++) ctr[k] = ((float)(b[k] * BW) + 0.5f * (float)(TW - 1)) * d.h;
                const float reach = (0.5f * (float)(TW - 1) + 1.5f) * d.h * (D == 3 ? 1.7320508f : 1.4142136f) * 1.001f;
                for (uint32_t i = 0; i < d.n_colliders && i < 16u; i++) {
                    const ColliderDev &c = d.colliders[i];
                    if (c.shape_type >= 3u) continue;
                    float pl[D], projl[D], proj[D];
                    pose_to_local<D>(c, ctr, pl);
                    const bool inside = project_local_on_boundary<D>(c, pl, projl);
                    pose_to_world<D>(c, projl, proj);
                    float n2 = 0.f;
#pragma unroll
                    for (int k = 0; k < D; k++) n2 += (proj[k] - ctr[k]) * (proj[k] - ctr[k]);
                    near |= (inside || !(n2 > reach * reach)) ? (1u << i) : 0u;
                }
            }
            for (int n = lane; n < ((TILE + 63) / 64) * 64; n += 64) {
                int t[3] = {n % TW, (n / TW) % TW, D == 3 ? n / (TW * TW) : 0};
                const int o = (t[0] >= BW ? 1 : 0) | (t[1] >= BW ? 2 : 0) | (t[2] >= BW ? 4 : 0);
                const uint32_t nb = __shfl(res, o & 7);        // "+" links live in lanes 0..7
                if (n < TILE && nb != NONE) {                  // nodes of blocks that are not active do not exist
                    float pt[D];
#pragma unroll
                    for (int k = 0; k < D; k++) pt[k] = (float)(b[k] * BW + t[k]) * d.h;
                    const NodeCdf far_cdf = {1.0e10f, 0u, NONE, 0u};
                    const NodeCdf c = near ? node_cdf_eval<D>(d, pt, near) : far_cdf;
                    if (o == 0) d.node_cdf[(size_t)id * NPB + (t[0] + (t[1] << BS) + (D == 3 ? (t[2] << (2 * BS)) : 0))] = c;
                    mine |= c.affinities;
                }
            }
            const bool any = __ballot(mine != 0u) != 0ull;
            const uint32_t total = __shfl(inc, 63);
            if (lane == 0) {
                d.block_cpic[id] = any ? 1u : 0u;
                if (any && total > 0u) d.cpic_list[atomicAdd(&d.counters[CTR_NCPIC], 1u)] = id;  // few blocks
            }
        }
    }
}

// sort.wgsl:117-127 finalize_particles_sort (the sorted-ids half). Also writes the particle
// ids in sorted order (= the reference's sorted_particle_ids) so that the canonical-order
// pass reads them contiguously instead of gathering through `perm`.
// The cell offsets are recomputed here from the cell counts (one coalesced 256-byte read + a wave scan per distinct
// block of the wave: one to three blocks in a sorted buffer) instead of read from cell_start, so that this pass does
// not depend on the block setup and shares its launch. Wave-uniform control flow: call with all 64 lanes.
template <int D> __device__ __forceinline__ void scatter_body(const Dev &d, int side, uint32_t i, uint32_t *offs) {
    const int lane = threadIdx.x & 63;
    uint32_t cid = NONE;
    if (i < num_slots(d)) cid = d.cellid[i];
    const uint32_t myid = cid == NONE ? NONE : cid / NPB;
    uint32_t start = 0;
    unsigned long long todo = __ballot(myid != NONE);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t id0 = __shfl(myid, leader);
        const bool mine = myid == id0;
        todo &= ~__ballot(mine);
        const uint32_t cnt = d.cell_count[id0 * NPB + lane];
        uint32_t inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(inc, off);
            if (lane >= off) inc += v;
        }
        __hip_atomic_store(&offs[lane], d.block_start[id0] + inc - cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (mine) start = __hip_atomic_load(&offs[cid % NPB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    if (cid == NONE) return;
    const uint32_t r = start + d.rank[i];
    d.perm[r] = i;
    d.perm_cell[r] = cid;
    d.perm_pid[r] = ldpid<D>(d.buf[side], d.npad, i);
}

// Block setup and scatter in ONE launch: both only need the scan (active list, block_start) and the cell counts. The
// first `nsetup` workgroups run the (longer) per-block setup, the others scatter 256 particles each.
template <int D, bool CDF> __global__ __launch_bounds__(SORT_THREADS) void k_setup_scatter(Dev d, int side, uint32_t epoch, uint32_t nsetup) {
    __shared__ uint32_t s_offs[SORT_THREADS / 64][64];
    if (blockIdx.x < nsetup) block_setup_body<D, CDF>(d, epoch, blockIdx.x, nsetup);
    else scatter_body<D>(d, side, (blockIdx.x - nsetup) * SORT_THREADS + threadIdx.x, s_offs[threadIdx.x >> 6]);
}

// Canonical order inside each cell: ascending persistent particle id. One thread per cell.
// Cells whose membership did not change since the last substep are already sorted (the buffer
// is written in canonical order), so the common case is one contiguous read and an early exit.
constexpr int CANON_REG = 12;
__global__ __launch_bounds__(SORT_THREADS) void k_canonical_order(Dev d) {
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const uint32_t total = B * NPB;
    for (uint32_t t = blockIdx.x * SORT_THREADS + threadIdx.x; t < total; t += gridDim.x * SORT_THREADS) {
        const uint32_t c = d.active[t >> 6] * NPB + (t & 63u);
        const uint32_t s = d.cell_start[c], e = d.cell_cursor[c];
        const uint32_t m = e - s;
        if (m < 2) continue;
        if (m <= CANON_REG) {
            uint32_t k[CANON_REG];
#pragma unroll
            for (int q = 0; q < CANON_REG; q++) k[q] = (uint32_t)q < m ? d.perm_pid[s + q] : NONE;  // NONE sorts last
            bool sorted = true;
#pragma unroll
            for (int q = 1; q < CANON_REG; q++) sorted = sorted && k[q - 1] <= k[q];
            if (sorted) continue;
            // (pid << 32 | perm) packed so that one 64-bit compare-exchange moves both;
            // odd-even transposition sort: fixed network, no run-time register indexing
            unsigned long long kp[CANON_REG];
#pragma unroll
            for (int q = 0; q < CANON_REG; q++)
                kp[q] = ((unsigned long long)k[q] << 32) | ((uint32_t)q < m ? d.perm[s + q] : NONE);
#pragma unroll
            for (int pass = 0; pass < CANON_REG; pass++) {
#pragma unroll
                for (int q = pass & 1; q + 1 < CANON_REG; q += 2) {
                    const unsigned long long lo = kp[q] < kp[q + 1] ? kp[q] : kp[q + 1];
                    const unsigned long long hi = kp[q] < kp[q + 1] ? kp[q + 1] : kp[q];
                    kp[q] = lo;
                    kp[q + 1] = hi;
                }
            }
#pragma unroll
            for (int q = 0; q < CANON_REG; q++)
                if ((uint32_t)q < m) {
                    d.perm[s + q] = (uint32_t)kp[q];
                    d.perm_pid[s + q] = (uint32_t)(kp[q] >> 32);
                }
        } else {
            for (uint32_t a = s + 1; a < e; a++) {
                const uint32_t pa = d.perm[a], ka = d.perm_pid[a];
                uint32_t j = a;
                while (j > s) {
                    const uint32_t kj = d.perm_pid[j - 1];
                    if (kj <= ka) break;
                    d.perm[j] = d.perm[j - 1];
                    d.perm_pid[j] = kj;
                    j--;
                }
                d.perm[j] = pa;
                d.perm_pid[j] = ka;
            }
        }
    }
}

}  // namespace wgs
