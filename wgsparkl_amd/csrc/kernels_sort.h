// kernels_sort.h — "grid sort" pass: sparse-grid activation and the regrouping of the particles by
// (block, cell, particle id). TWO launches per substep.
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
//   * launch 1 (k_rebin / k_bin), one thread per particle: new cell, block activation, per-block
//     particle totals (one atomic per (wave, block)). The buffer is the sorted output of the
//     previous substep and particles move less than a cell per substep, so almost every particle
//     is a STAYER — same cell as before, nothing else to do for it. The few MOVERS are pushed on
//     a per-destination-cell list (one atomic exchange each; the reference pushes EVERY particle
//     on such a list every substep, sort.wgsl:129-137);
//   * launch 2 (k_regroup): the first workgroups scan the per-block totals in chunks (active list,
//     first_particle: prefix_sum.wgsl), the others — one wave per active block, lane = cell — set
//     the block up (neighbour links, node cdf tile) and build the new cell runs by merging each
//     cell's stayers (already in id order) with its arrivals (selected in id order from the list):
//     the order inside a cell is canonical (ascending persistent particle id) by construction, so
//     every downstream fp32 sum is reproducible run to run and independent of the storage order
//     (reference: atomic race order, sort.wgsl:126,133). The waves only wait for the scan (a flag
//     per chunk) when they need their block's first_particle, after all their independent work;
//   * no per-particle rank, no per-cell counters, no sorting pass.
#pragma once
#include "kernels_shard.h"

namespace wgs {

constexpr int SORT_THREADS = 256;
constexpr int TOUCH_SET = 32;  // distinct blocks a workgroup can de-duplicate in LDS

template <int D> __device__ inline void load_cell(const Dev &d, const float *in, uint32_t i, int *cell) {
    const float4 xm = ldq(in, d.npad, Pl<D>::XM, i);
    const bool p2 = d.h_pow2 != 0u;  // x * (1 / h) is then x / h bit for bit, without the IEEE division sequence
    cell[0] = assoc_cell(xm.x, d.h, d.inv_h, p2);
    cell[1] = assoc_cell(xm.y, d.h, d.inv_h, p2);
    if constexpr (D == 3) cell[2] = assoc_cell(xm.z, d.h, d.inv_h, p2);
}

// sort.wgsl:129-137 (per-cell linked list), for movers only: slot i becomes the head of its destination
// cell's list. cell_head holds slot + 1 (0 = empty: the array is zero at rest, k_regroup resets what it consumed).
__device__ inline void push_mover(const Dev &d, uint32_t cid, uint32_t i) { d.mv_next[i] = atomicExch(&d.cell_head[cid], i + 1u); }
// ... for a particle that came from another block (steady state): the first BLK_ARR arrivals of a block are recorded in the block's
// array — the wave that regroups the block fetches them with one coalesced load —, the others go on their cell's list.
__device__ inline void push_arrival(const Dev &d, uint32_t cid, uint32_t i) {
    const uint32_t pos = atomicAdd(&d.blk_narr[cid >> 6], 1u);
    if (pos < BLK_ARR) d.blk_arr[(size_t)(cid >> 6) * BLK_ARR + pos] = i;
    else push_mover(d, cid, i);
}

// sort.wgsl:89-99 update_block_particle_count, aggregated: one atomic per (wave, block). Wave-uniform control flow.
__device__ inline void count_blocks(const Dev &d, int lane, uint32_t myid) {
    unsigned long long todo = __ballot(myid != NONE);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t id0 = __shfl(myid, leader);
        const unsigned long long same = __ballot(myid == id0);
        todo &= ~same;
        if (lane == leader) atomicAdd(&d.block_acc[id0], (uint32_t)__popcll(same));
    }
}

// sort.wgsl:26-36 touch_particle_blocks + sort.wgsl:89-99 update_block_particle_count, fused: the general form of
// launch 1 (first substep, table-rebuild substeps). Every particle it bins is a mover (it has no previous cell).
template <int D> __device__ __forceinline__ void bin_body(const Dev &d, int side, uint32_t epoch, uint32_t bid) {
    constexpr int BS = Dim<D>::BSHIFT, BW = Dim<D>::BW, NN = Dim<D>::NNBR;
    __shared__ uint32_t s_keys[TOUCH_SET], s_ids[TOUCH_SET];
    const float *in = d.buf[side];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < TOUCH_SET) s_keys[tid] = NONE;
    __syncthreads();
    const uint32_t i = bid * SORT_THREADS + tid;
    uint32_t slots_end = num_slots(d);
    bool valid = i < slots_end;
    if (d.sharded && valid) valid = ldpid<D>(in, d.npad, i) != 0xffffffffu;  // slot vacated by a migrated particle
    int b[3] = {0, 0, 0};
    uint32_t key = NONE, local = 0;
    if (valid) {
        int c[D];
        load_cell<D>(d, in, i, c);
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
    uint32_t myid = !valid ? NONE : (myslot != NONE ? s_ids[myslot] : mydirect);
    if (myid >= d.cap) myid = NONE;  // grid overflow (reported): the particle is left out of this substep
    // ---- 3. per-block totals; every particle is a mover into its cell
    count_blocks(d, lane, myid);
    if (i < slots_end) {
        const uint32_t cid = myid == NONE ? NONE : myid * NPB + local;
        d.cellid[i] = cid;
        if (cid != NONE) push_mover(d, cid, i);
    }
}
// do_bodies: the previous substep left integrate_bodies to this launch (kernels_bodies.h bodies_integrate_one)
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_bin(Dev d, int side, uint32_t epoch, uint32_t do_bodies) {
    if (do_bodies && blockIdx.x == 0 && threadIdx.x < 16) bodies_integrate_one<D>(d, threadIdx.x);
    bin_body<D>(d, side, epoch, blockIdx.x);
}

// Steady-state launch 1 (sort.wgsl:26-36,89-99 for a buffer that is the sorted output of the previous
// substep): slot i held cell perm_cell[i] of block b = perm_cell[i] >> 6 one substep ago, particles move
// less than a cell per substep, so almost every particle is still in block b: its new cell id is
// b * 64 + new local cell and the blocks to activate are b's neighbour links of the previous substep —
// no hash lookup, no LDS set. Only particles that changed block go through the hash map, and only particles
// that changed CELL are pushed on a list.
// REBIN_K particles per thread (slices of SORT_THREADS consecutive slots, so every slice is coalesced and a wave's 64 lanes
// are 64 consecutive sorted particles — one or two blocks): the kernel's life is one dependent chain per wave (sort entry ->
// block key -> links -> stamps / atomics, ~4.5 us at full occupancy), so K slices with all their first loads in flight
// together cost about one chain. Measured (sort pass, A/B on one box): K = 2: C5 256 -> 220 us, C4 154 -> ~140 us, C2 25.0 -> 24.1 us;
// K = 4: 224 / 25.1; K = 8: 228 / 29.2 — the atomics and the per-slice stamping do not shrink with K.
#ifndef WGS_REBIN_K
#define WGS_REBIN_K 2
#endif
constexpr int REBIN_K = WGS_REBIN_K;
template <int D> __device__ __forceinline__ void rebin_body(const Dev &d, int side, uint32_t epoch, uint32_t bid) {
    constexpr int BS = Dim<D>::BSHIFT, BW = Dim<D>::BW, NN = Dim<D>::NNBR, K = REBIN_K;
    const float *in = d.buf[side];
    const int tid = threadIdx.x, lane = tid & 63;
    // sharded runs: the residents minus the slots vacated by particles that now live on a neighbour, and behind them,
    // in [NPREV, N), the particles that arrived in the last substep (kernels_arrivals.h): they have no previous cell and
    // take the hash path below like a particle that changed block
    const uint32_t nslots = num_slots(d);
    const uint32_t nprev = d.sharded ? ctr_cur(d, CTR_NPREV) : 0u;
    uint32_t idx[K], old[K];
    bool in_range[K], valid[K];
    float4 xm[K];
    // ---- round 1: everything that depends on the slot alone
#pragma unroll
    for (int k = 0; k < K; k++) {
        idx[k] = (bid * (uint32_t)K + (uint32_t)k) * SORT_THREADS + (uint32_t)tid;
        in_range[k] = idx[k] < nslots;
        valid[k] = in_range[k];
        old[k] = NONE;
        xm[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in_range[k]) {
            if (!d.sharded || idx[k] < nprev) old[k] = d.perm_cell[idx[k]];  // NONE only after a grid overflow: take the hash path then
            xm[k] = ldq(in, d.npad, Pl<D>::XM, idx[k]);
            if (d.sharded) valid[k] = ldpid<D>(in, d.npad, idx[k]) != 0xffffffffu;
        }
    }
    // ---- round 2: the key of the previous block
    uint32_t okey[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (old[k] != NONE) old[k] &= ~CELL_LISTED;
        okey[k] = (valid[k] && old[k] != NONE) ? d.block_key[old[k] >> 6] : 0u;
    }
    const bool p2 = d.h_pow2 != 0u;
    uint32_t last_b1 = NONE;   // (wave-uniform) the block whose neighbours this wave stamped last: consecutive slices mostly share it
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint32_t myid = NONE, local = 0u, mykey = 0u;
        if (valid[k]) {
            const uint32_t ob = old[k] >> 6;
            int c[3] = {assoc_cell(xm[k].x, d.h, d.inv_h, p2), assoc_cell(xm[k].y, d.h, d.inv_h, p2), D == 3 ? assoc_cell(xm[k].z, d.h, d.inv_h, p2) : 0};
            int nb[3] = {0, 0, 0};
            uint32_t shift = 0;
#pragma unroll
            for (int a = 0; a < D; a++) {
                nb[a] = c[a] >> BS;
                local |= (uint32_t)(c[a] & (BW - 1)) << shift;
                shift += BS;
            }
            int hi[3] = {nb[0] + 1, nb[1] + 1, nb[2] + 1};
            if (!block_in_key_range<D>(nb) || !block_in_key_range<D>(hi)) {
                atomicOr(&d.counters[CTR_ERRORS], ERRBIT_KEYRANGE);
            } else {
                const uint32_t key = pack_key<D>(nb);
                mykey = key;
                myid = (old[k] != NONE && key == okey[k]) ? ob : activate_block(d, key, epoch);  // few particles change block
                if (myid >= d.cap) myid = NONE;
            }
        }
        // activate every distinct block of the wave and its +1 neighbours (grid.wgsl:300-320), count its particles
        unsigned long long todo = __ballot(myid != NONE);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t b1 = __shfl(myid, leader);
            // (the block's key from the leader's own particle, not from block_key[b1]: a block handed out in THIS launch by a
            // wave of another XCD has its key in that XCD's L2 only)
            const uint32_t k1 = __shfl(mykey, leader);
            const unsigned long long same = __ballot(myid == b1);
            todo &= ~same;
            if (lane == leader) atomicAdd(&d.block_acc[b1], (uint32_t)__popcll(same));
            if (b1 == last_b1) continue;   // its neighbours carry this substep's stamp already (this wave wrote it a slice ago)
            last_b1 = b1;
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
                    unpack_key<D>(k1, kb);
                    int nb[3] = {kb[0] + (lane & 1), kb[1] + ((lane >> 1) & 1), kb[2] + ((lane >> 2) & 1)};
                    if (block_in_key_range<D>(nb)) activate_block(d, pack_key<D>(nb), epoch);
                }
            }
        }
        if (in_range[k]) {  // (a vacated slot gets NONE: k_regroup skips it)
            const uint32_t cid = myid == NONE ? NONE : myid * NPB + local;
            d.cellid[idx[k]] = (cid != NONE && old[k] != NONE && cid != old[k]) ? (cid | CELL_MOVED) : cid;
            if (old[k] != NONE && cid != old[k]) d.block_dirty[old[k] >> 6] = epoch;   // its previous block's run changes
            // (on a list only when it came from another block — or from a neighbouring rank: no previous cell;
            // a particle that changed cell inside its block is met by the wave that regroups the block)
            if (cid != NONE && (old[k] == NONE || (cid >> 6) != (old[k] >> 6))) push_arrival(d, cid, idx[k]);
        }
    }
}
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_rebin(Dev d, int side, uint32_t epoch, uint32_t do_bodies) {
    if (do_bodies && blockIdx.x == 0 && threadIdx.x < 16) bodies_integrate_one<D>(d, threadIdx.x);
    rebin_body<D>(d, side, epoch, blockIdx.x);
}
// ---------------------------------------------------------------------------------------------------------------
// Launch 2. Chunked scan of the known blocks (physical ids): chunk k = ids [k * SCAN_CHUNK, (k + 1) * SCAN_CHUNK).
//   active[a] = id of the a-th block stamped with the current epoch   (grid.wgsl:323-334's active_blocks list,
//               in physical-id order)
//   block_start[id] = exclusive scan of the particle counts           (sort.wgsl:101-115 + prefix_sum.wgsl)
// in two levels: the scan workgroups produce the totals before every group of 16 blocks, the wave that regroups a block
// adds the blocks of its own group (block_prefix) and writes the block's two entries.
// A chunk's workgroup publishes its total, adds the totals of the chunks before it and publishes the group totals
// (wait_tagged / publish_tagged: epoch-tagged words, no fences).
#ifdef WGS_ABLATE
// Stage clocks of launch 2 (timing experiments, tools/gpu_sort_prof.py; never compiled into the product): one row of
// absolute wall_clock64 readings (100 MHz) per regrouped block id: [0] start, [1..7] stages, and row WGS_PROF_ROWS - 1
// [0] = when the scan published its last word. Plain stores, no atomics.
constexpr int WGS_PROF_ROWS = 8192;
__device__ unsigned long long g_prof[WGS_PROF_ROWS][8];
#define WGS_PROF_START() if (lane == 0 && id < WGS_PROF_ROWS - 1) g_prof[id][0] = wall_clock64();
#define WGS_PROF(k) if (lane == 0 && id < WGS_PROF_ROWS - 1) g_prof[id][1 + (k)] = wall_clock64();
#define WGS_PROF_END()
#else
#define WGS_PROF_START()
#define WGS_PROF(k)
#define WGS_PROF_END()
#endif
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_CHUNK = SORT_THREADS * SCAN_ITEMS;  // 4096 blocks per scan workgroup
constexpr int RUNCAP = 768;   // particles of one block a wave stages in LDS (more: same code on global memory)

// Cross-workgroup hand-over inside launch 2 WITHOUT fences: on a multi-XCD part an agent-scope release / acquire
// fence writes back / invalidates a whole L2, and four thousand waves doing that made this launch 100 us long. Every
// value that crosses workgroups is instead one 64-bit word = (epoch << 32 | value), stored and polled with relaxed
// agent-scope atomics (coherent accesses, no cache maintenance): a reader that sees the current epoch in the high word
// has the value in the low word of the very same access. Epochs only grow, so nothing is reset between substeps.
__device__ inline uint32_t wait_tagged(const unsigned long long *word, uint32_t epoch) {
    unsigned long long v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((uint32_t)(v >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(8);
        v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return (uint32_t)v;
}
__device__ inline void publish_tagged(unsigned long long *word, uint32_t epoch, uint32_t value) {
    __hip_atomic_store(word, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool SHARD = false>
__device__ __forceinline__ void scan_chunk(const Dev &d, uint32_t epoch, uint32_t k, uint32_t nchunks, unsigned long long *s_wave, unsigned long long *s_bcast) {
    const uint32_t nphys = min(d.counters[CTR_NPHYS], d.cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef WGS_ABLATE
    if (tid == 0 && k == 0) g_prof[WGS_PROF_ROWS - 1][1] = wall_clock64();
#endif
    // packed scan: high word = number of active blocks, low word = particles
    unsigned long long sum = 0ull;
    const uint32_t first = k * SCAN_CHUNK + (uint32_t)tid * SCAN_ITEMS;
    uint32_t stamp[SCAN_ITEMS], acc[SCAN_ITEMS];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {  // all loads unconditional and independent: one round trip
        const uint32_t idc = min(first + j, nphys ? nphys - 1u : 0u);
        stamp[j] = d.block_stamp[idc];
        acc[j] = d.block_acc[idc];
    }
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const bool act = first + j < nphys && stamp[j] == epoch;
        sum += act ? ((1ull << 32) | (unsigned long long)acc[j]) : 0ull;
    }
    // (the two words scanned separately — the particle total of a launch stays below 2^32 — by DPP: layout.h wave_scan_incl_u32)
    const unsigned long long inc = ((unsigned long long)wave_scan_incl_u32((uint32_t)(sum >> 32)) << 32) | (unsigned long long)wave_scan_incl_u32((uint32_t)sum);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned long long wave_off = 0ull, total = 0ull;
#pragma unroll
    for (int w = 0; w < SORT_THREADS / 64; w++) {
        const unsigned long long t = s_wave[w];
        if (w < wave) wave_off += t;
        total += t;
    }
#ifdef WGS_ABLATE
    if (tid == 0 && k == 0) g_prof[WGS_PROF_ROWS - 1][2] = wall_clock64();
#endif
    if (tid == 0) {
        publish_tagged(&d.chunk_a[k], epoch, (uint32_t)(total >> 32));
        publish_tagged(&d.chunk_b[k], epoch, (uint32_t)total);
    }
    // totals of the chunks before this one (at most cap / 4096 of them), fetched in parallel
    unsigned long long part = 0ull;
    for (uint32_t p = (uint32_t)tid; p < k; p += SORT_THREADS)
        part += ((unsigned long long)wait_tagged(&d.chunk_a[p], epoch) << 32) + wait_tagged(&d.chunk_b[p], epoch);
    part = ((unsigned long long)wave_sum_u32((uint32_t)(part >> 32)) << 32) + (unsigned long long)wave_sum_u32((uint32_t)part);   // (each word below 2^32)
    __syncthreads();  // s_wave consumed above
    if (lane == 0) s_wave[wave] = part;
    __syncthreads();
    if (tid == 0) {
        unsigned long long b = 0ull;
        for (int w = 0; w < SORT_THREADS / 64; w++) b += s_wave[w];
        *s_bcast = b;
    }
    __syncthreads();
    const unsigned long long base = *s_bcast;
    // what the regrouping waves of this launch need: the totals before each group of SCAN_ITEMS blocks (one coalesced
    // pair of words per thread); a wave adds the few blocks of its own group itself (block_prefix)
    const unsigned long long run = base + wave_off + inc - sum;
    publish_tagged(&d.group_a[k * SORT_THREADS + tid], epoch, (uint32_t)(run >> 32));
    publish_tagged(&d.group_b[k * SORT_THREADS + tid], epoch, (uint32_t)run);
    if (tid == 0 && k + 1 == nchunks) {
        d.counters[CTR_NBLOCKS] = (uint32_t)((base + total) >> 32);
        d.counters[CTR_NSORTED] = (uint32_t)(base + total);
    }
#ifdef WGS_ABLATE
    if (tid == 0) g_prof[WGS_PROF_ROWS - 1][0] = wall_clock64();
#endif
    if (k == 0 && tid == 0) d.counters[CTR_NPHYS_SEEN + (epoch & 1u)] = d.counters[CTR_NINSERT];
    // the list counters of the NEXT substep (the other set: nothing of this substep reads or appends to it; layout.h)
    if (k == 0 && tid < 16) d.counters[tid < 8 ? ctr_ncpic((uint32_t)tid, epoch + 1u) : ctr_nvisit((uint32_t)tid - 8u, epoch + 1u)] = 0u;
    if (k == 0 && tid == 16) d.counters[ctr_nhalo(epoch + 1u)] = 0u;
    if constexpr (SHARD) {
        if (k == 0 && tid < 4 && d.msg.out[tid >> 1]) reinterpret_cast<uint32_t *>(d.msg.out[tid >> 1])[tid & 1] = 0u;  // record counts of this substep's outgoing messages
    }
}

// Second level of the scan, by the wave that owns block `id`: loads of the group's 16 (stamp, count) pairs — issued
// with the wave's other first-round loads —, then the group's tagged totals. Returns first_particle and the block's
// index in the active list.
struct GroupLoads {
    uint32_t stamp, acc;
};
__device__ inline GroupLoads block_prefix_loads(const Dev &d, uint32_t id, int lane) {
    GroupLoads g = {0u, 0u};
    if (lane < SCAN_ITEMS) {
        const uint32_t j = min((id & ~(uint32_t)(SCAN_ITEMS - 1)) + (uint32_t)lane, d.cap - 1u);
        g.stamp = d.block_stamp[j];
        g.acc = d.block_acc[j];
    }
    return g;
}
__device__ inline void block_prefix(const Dev &d, uint32_t epoch, uint32_t id, int lane, const GroupLoads &g, uint32_t nphys, uint32_t &start, uint32_t &aidx) {
    const uint32_t gbase = id & ~(uint32_t)(SCAN_ITEMS - 1), j = id & (uint32_t)(SCAN_ITEMS - 1);
    const bool before = lane < (int)j && gbase + (uint32_t)lane < nphys && g.stamp == epoch;
    const uint32_t cnt = (uint32_t)__popcll(__ballot(before));
    uint32_t psum = before ? g.acc : 0u;
    psum = wave_sum_u32(psum);   // (lanes 0..15 hold the terms, the others zero)
    const uint32_t grp = id / SCAN_ITEMS;
    aidx = wait_tagged(&d.group_a[grp], epoch) + cnt;
    start = wait_tagged(&d.group_b[grp], epoch) + psum;
}

// Per active block, one wave, lane = cell: neighbour links (replaces the per-thread hash lookups of
// p2g.wgsl:238-275 / g2p.wgsl:72-132), the new cell runs (sort.wgsl:117-127 finalize_particles_sort, in canonical
// order), reset of the accumulators.
// CDF (collider simulations without mesh colliders): also the node cdf of the block's (BW+2)^D tile and the class of
// the block (see k_cdf, whose steps 1 and 2 these are; step 3 then runs in the prologue of the CPIC P2G launch), so
// that a collider simulation needs no CDF launch of its own.
// `have_old`: the buffer is the sorted output of the previous substep (cell_start / cell_cursor of a block whose
// links are one epoch old describe its previous runs: that is where the stayers are); otherwise every particle is
// on a list.
// Returns the number of particles of the block's new run that changed cell in the last step (statistics, wave-uniform).
template <int D, bool CDF, bool SHARD, bool SUMM>
__device__ __forceinline__ uint32_t regroup_block(const Dev &d, int side, uint32_t epoch_of_launch, uint32_t id, uint32_t nphys, bool have_old, bool no_new_blocks, bool check_keys,
                                              uint32_t *s_in, uint32_t *s_out, uint32_t *s_pid, ColliderDev *cols, bool &cols_staged) {
    constexpr int NN = Dim<D>::NNBR;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    // (and the substep number: what is stored of it — stamps, notes — would otherwise sit in vector registers, or scratch, for the whole launch)
    uint32_t epoch = epoch_of_launch;
    if constexpr (SUMM) asm volatile("" : "+s"(epoch));
    const float *in = d.buf[side];
    // (the lane's part of every per-block index is pinned once per block: hipcc otherwise hoists array + 4 lane, one 64-bit pair per
    // array, out of the loop over the blocks, keeps the pairs in scratch for want of registers and reloads them — a scratch load and a
    // wait — in front of the block's first loads: the sort is a chain of dependent round trips, and that was one more)
    const uint32_t lane_p = (uint32_t)lane;
    const uint32_t idx = id * NPB + lane_p;
    WGS_PROF_START()
    // ---- first round of loads, all independent: activity stamp, previous runs and links of this block (read before
    // anything of it is overwritten), list heads
    const uint32_t stamp = d.block_stamp[id];
    const uint32_t le = d.links_epoch[id];
    uint32_t cs_old = d.cell_start[idx], ce_old = d.cell_cursor[idx];
    const uint32_t head = d.cell_head[idx];
    const uint32_t bkey = d.block_key[id];
    const uint32_t dirty_at = d.block_dirty[id];
    const uint32_t bstart_old = d.block_start[id], ident = d.block_ident[id];
    const uint32_t narr = d.blk_narr[id];                                 // arrivals from other blocks ...
    uint32_t a_ent = NONE;                                                // ... lane's entry of their array (fetched when there are any)
    uint32_t cdf_seen = 0u, cdf_class = 0u, cdf_summ = 0u;
    if constexpr (CDF) {
        cdf_seen = d.block_cdf_gen[id];
        cdf_class = d.block_cpic[id];
        if constexpr (SUMM) cdf_summ = d.block_cdf_summ[id];
    }
    uint32_t link = NONE;
    if (lane < 16) link = d.nbr_known[id * 16u + lane_p];
    const GroupLoads grp = block_prefix_loads(d, id, lane);
    if (stamp != epoch) {  // wave-uniform: not active in this substep
        // EVICTION: a block nobody activated for EVICT_AGE substeps leaves the table — its slot is marked KEY_TOMB,
        // its id goes on the free list the next insertion takes from. The table and the ids then live as long as the simulation moves
        // slowly enough for the marks not to crowd the table (the host watches CTR_NTOMB), instead of being rebuilt from every particle —
        // a k_bin launch and a regrouping of everything, ~0.65 ms at 1 M particles — whenever three quarters of the ids were handed out
        // (a body crossing the grid: every few hundred substeps). Whoever still holds the id as "a neighbour that is in the table"
        // (nbr_known) compares the key before trusting it: block_key of an evicted id is NONE until the id is handed out again.
        {
            // (bounded on the device: while the marks hold a quarter of the table's 2 x cap slots nobody is evicted — live blocks + marks
            // then stay below 1.5 x cap whatever the host's look is late by, and a probe sequence always ends: the host clears the marks,
            // capi.hip maintain_grid -> k_table_refresh, and eviction goes on)
            if (d.free_ids != nullptr && lane == 0 && (epoch & (EVICT_AGE - 1u)) == 0u && stamp != 0u && epoch - stamp > EVICT_AGE && bkey != NONE &&
                d.counters[CTR_NTOMB] <= d.cap / 2u) {
                const uint32_t hs = d.block_slot[id];
                d.hkeys[hs] = KEY_TOMB;
                d.hvals[hs] = NONE;   // (like an empty slot's: an insertion that takes the slot publishes its id here)
                d.block_key[id] = NONE;
                if constexpr (CDF) d.block_cdf_gen[id] = 0u;
                d.free_ids[atomicAdd(&d.counters[CTR_NFREE], 1u)] = id;
                atomicAdd(&d.counters[CTR_NTOMB], 1u);
            }
        }
        return 0u;
    }
    const bool old_ok = have_old && le == epoch - 1u;
    if (!old_ok) {
        cs_old = ce_old = 0u;
        link = NONE;
    }
    // a neighbour that was in the table one substep ago keeps its physical id: it only has to be active now (one round
    // trip instead of the three of a hash lookup)
    uint32_t link_stamp = 0u, link_cnt = 0u, link_key = NONE;
    if (link != NONE) {
        link_stamp = d.block_stamp[link];
        link_cnt = d.block_acc[link];  // particles of that neighbour in this substep (launch 1's total)
        // (an id that was evicted since — and maybe handed out again, to another block — is not this neighbour; compared in the launch
        // behind one that may evict: k_regroup)
        if (check_keys) link_key = d.block_key[link];
    }
    const bool links_valid = old_ok;
    // ---- stage the new cell ids of the block's previous run (contiguous: cells are consecutive runs); with movers
    // into this block also the particle ids of the run (the merge compares them)
    const uint32_t run0 = lane_value(cs_old, 0), run1 = lane_value(ce_old, 63);
    const uint32_t runlen = run1 - run0;
    const bool in_lds = runlen <= (uint32_t)RUNCAP;
    const uint32_t narr_in = min(narr, BLK_ARR);          // (wave-uniform)
    const bool any_arr = __ballot(head != 0u) != 0ull || narr != 0u;   // particles from other blocks
    uint32_t a_cell = NONE, a_epid = 0u;                  // the array arrival of this lane: its new cell and its id
    if ((uint32_t)lane < narr_in) {
        a_ent = d.blk_arr[(size_t)id * BLK_ARR + (uint32_t)lane];
        a_cell = cell_of(d.cellid[a_ent]);
        a_epid = ldpid<D>(in, d.npad, a_ent);
        if ((a_cell >> 6) != id) a_cell = NONE;           // (never: the particle named this block)
    } else {
        a_ent = NONE;
    }
    // Every particle of the previous run is still in its cell (nobody flagged the block for this substep; on a slab a slot
    // vacated by a migrated particle flags its block like a particle that changed cell): the runs only move, nothing
    // of the run has to be looked at — unless particles arrive, whose ids are merged with the stayers'.
    const bool clean = dirty_at != epoch;
    const bool need_ids = !clean || any_arr;
    // The block's new run is built by the whole wave at once (below: a counting sort by new cell, then every cell orders its own
    // few particles by id) when something changed, the previous run and what arrives fit the LDS stages, and nobody is on a
    // cell's list (lists: table-rebuild substeps, and arrivals beyond a block's array); lane by lane, cell by cell, otherwise.
    uint32_t n_moved = 0u;   // (per lane, summed at the end) statistics: particles of this block's new run that changed cell
    const bool par = old_ok && need_ids && in_lds && __ballot(head != 0u) == 0ull && runlen + narr_in <= (uint32_t)RUNCAP;   // (wave-uniform)
    // (two batches of PAR_R / 2 rounds, every load of a batch in flight before the first LDS store: a round at a time — one wait per
    // round — the staging of a full block was six to eight dependent round trips, 5 us of a dirty block's 16)
    constexpr int PAR_R = (RUNCAP + 63) / 64;
    if (in_lds && need_ids) {
        const bool want_cells = !clean || par;   // (wave-uniform)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half == 1 && runlen <= 64u * (uint32_t)(PAR_R / 2)) break;
            uint32_t ce[PAR_R / 2], pe[PAR_R / 2];
#pragma unroll
            for (int r = 0; r < PAR_R / 2; r++) {
                const uint32_t t = (uint32_t)lane + 64u * (uint32_t)(half * (PAR_R / 2) + r);
                ce[r] = NONE;
                pe[r] = 0u;
                if (t < runlen) {
                    if (want_cells) ce[r] = d.cellid[run0 + t];
                    pe[r] = ldpid<D>(in, d.npad, run0 + t);
                }
            }
#pragma unroll
            for (int r = 0; r < PAR_R / 2; r++) {
                const uint32_t t = (uint32_t)lane + 64u * (uint32_t)(half * (PAR_R / 2) + r);
                if (t < runlen) {
                    if (want_cells) {
                        s_in[t] = ce[r];   // (the raw entry: bit 31 tells the wave-parallel form below who changed cell)
                        n_moved += (ce[r] != NONE && (ce[r] & CELL_MOVED) != 0u && (cell_of(ce[r]) >> 6) == id) ? 1u : 0u;   // (came from another cell of this block)
                    }
                    s_pid[t] = pe[r];
                }
            }
        }
    }
    // ---- neighbour links (replaces the per-thread hash lookups of p2g.wgsl:238-275 / g2p.wgsl:72-132)
    uint32_t res = NONE;
    int b[3] = {0, 0, 0};
    unpack_key<D>(bkey, b);
    if (lane < 16) {
        const uint32_t o = lane & 7u;
        const bool minus = lane >= 8;
        uint32_t known = NONE;
        if ((int)o < NN) {
            const int sgn = minus ? -1 : 1;
            int nb[3] = {b[0] + sgn * (int)(o & 1u), b[1] + sgn * (int)((o >> 1) & 1u), b[2] + sgn * (int)((o >> 2) & 1u)};
            if (check_keys && link != NONE && link_key != pack_key<D>(nb)) link = NONE;   // (evicted: no longer in the table under that id)
            if (link != NONE) {
                known = link;
                res = link_stamp == epoch ? link : NONE;
            } else if (!(links_valid && no_new_blocks)) {  // unknown, or the table has grown: look it up
                if (block_in_key_range<D>(nb)) known = hmap_lookup(d, pack_key<D>(nb));
                if (known != NONE && d.block_stamp[known] == epoch) res = known;
                if (res != NONE) link_cnt = d.block_acc[res];
            }
        }
        if (res == NONE) link_cnt = 0u;
        (minus ? d.nbr_minus : d.nbr_plus)[id * 8u + o] = res;
        d.nbr_known[id * 16u + lane] = known;
    }
    WGS_PROF(0)
    // (single wave: LDS accesses of a wave execute in order, the relaxed wavefront-scope atomics below keep the
    // compiler from reordering or forwarding across lanes)
    auto new_cell_of = [&](uint32_t i) {
        return cell_of(in_lds ? __hip_atomic_load(&s_in[i - run0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : d.cellid[i]);
    };
    // (a particle of this cell's previous run that is still in the cell)
    auto stays_here = [&](uint32_t i) { return clean || new_cell_of(i) == idx; };
    auto pid_of_old = [&](uint32_t i) {
        return in_lds ? __hip_atomic_load(&s_pid[i - run0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : ldpid<D>(in, d.npad, i);
    };
    WGS_PROF(1)
    // (the block's particle total is known before its runs are: launch 1's count, or the count kept up to date by the fused G2P)
    const uint32_t bcount = lane_value(grp.acc, (int)(id & (uint32_t)(SCAN_ITEMS - 1)));
    uint32_t pc_flag = 0u;  // CELL_LISTED for the perm_cell entries of a block near a collider
    bool listed = false;  // near a collider and holding particles: on the lists of the CPIC bodies of P2G / G2P
    // ---- node cdf tile + block class (independent of the scan: placed before the wait for it)
    // A block keeps its physical id, i.e. its place in space, so the node cdfs and the class computed for it one substep ago
    // still hold — nothing below needs to run again — for a block that holds particles (then all its "+" neighbours are active,
    // which is what the class depends on besides position), was computed under the current generation (bumped by every table
    // rebuild, growth and pose upload) with no MOVING collider in reach of its tile, and has none in reach now (the quick
    // reject below, against the colliders of Dev::cdf_moving only: the reference's sand3 has one among six).
    // Quick reject, wave-uniform: a collider whose boundary is farther from the tile's centre than the tile's half diagonal
    // plus the affinity reach (1.5 h per axis) touches none of its nodes, and a centre outside the shape then means every node
    // is outside. One projection per collider instead of (BW+2)^D.
    auto reach_mask = [&](uint32_t which) {   // bit i: collider i (of `which`) can reach a node of this tile
        constexpr int BW = Dim<D>::BW, TW = Dim<D>::TW;
        if (SUMM && !cols_staged) {   // (wave-uniform; this wave reads what it wrote itself — another wave's copy holds the same words)
            const uint32_t nw = min(d.n_colliders, 16u) * (uint32_t)(sizeof(ColliderDev) / 4u);
            for (uint32_t i = (uint32_t)lane; i < nw; i += 64u)
                __hip_atomic_store(&reinterpret_cast<uint32_t *>(cols)[i], reinterpret_cast<const uint32_t *>(d.colliders)[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            // (orders the lanes' stores before the reads below for the compiler; in hardware a wave's LDS accesses are served in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            cols_staged = true;
        }
        uint32_t near = 0u;
        float ctr[D];
#pragma unroll
        for (int k = 0; k < D; k++) ctr[k] = ((float)(b[k] * BW) + 0.5f * (float)(TW - 1)) * d.h;
        const float reach = (0.5f * (float)(TW - 1) + 1.5f) * d.h * (D == 3 ? 1.7320508f : 1.4142136f) * 1.001f;
        auto reaches = [&](const ColliderDev &c) {
            float pl[D], projl[D], proj[D];
            pose_to_local<D>(c, ctr, pl);
            const bool inside = project_local_on_boundary<D>(c, pl, projl);
            pose_to_world<D>(c, projl, proj);
            float n2 = 0.f;
#pragma unroll
            for (int k = 0; k < D; k++) n2 += (proj[k] - ctr[k]) * (proj[k] - ctr[k]);
            return inside || !(n2 > reach * reach);
        };
        if constexpr (SUMM) {   // lane = collider: one projection's time for all of them (the same arithmetic per collider)
            const uint32_t i = (uint32_t)lane & 15u;
            bool r = false;
            if (lane < 16 && i < d.n_colliders && ((which >> i) & 1u) && cols[i].shape_type < 3u) r = reaches(cols[i]);
            near = (uint32_t)__ballot(r) & 0xffffu;
        } else {
            for (uint32_t i = 0; i < d.n_colliders && i < 16u; i++) {
                if (!((which >> i) & 1u)) continue;
                const ColliderDev &c = d.colliders[i];
                if (c.shape_type >= 3u) continue;
                near |= reaches(c) ? (1u << i) : 0u;
            }
        }
        return near;
    };
    // A block WITHOUT particles (the rim of a body: active as somebody's "+" neighbour) is asked for its own 64 nodes only — they
    // are the rim of its "-" neighbours' tiles; nobody asks for its class, which would depend on which of ITS "+" neighbours
    // happen to be active —, and those too keep (the note then lacks CDF_FULL: the class is computed when particles arrive).
    // (The rim blocks within reach of the floor were the slowest waves of this launch at C2: 10 us, every substep.)
    constexpr uint32_t CDF_FULL = 0x80000000u;
    uint32_t near_moving = 0u;
    bool cdf_cached = CDF && d.cdf_gen != 0u && bcount > 0u && cdf_seen == (d.cdf_gen | CDF_FULL);
    bool own_cached = CDF && d.cdf_gen != 0u && bcount == 0u && (cdf_seen & ~CDF_FULL) == d.cdf_gen;
    if ((cdf_cached || own_cached) && d.cdf_moving != 0u) {
        near_moving = reach_mask(d.cdf_moving);
        cdf_cached = cdf_cached && near_moving == 0u;
        own_cached = own_cached && near_moving == 0u;
    }
    if constexpr (!SUMM) {
        if (cdf_cached) {
            const bool any = cdf_class != 0u;
            if (lane == 0 && any) d.cpic_list[(size_t)(id & 7u) * d.cap + atomicAdd(&d.counters[ctr_ncpic(id & 7u, epoch)], 1u)] = id;
            listed = any;
            pc_flag = any ? CELL_LISTED : 0u;
        } else if (own_cached) {
            // (nothing to do: its nodes are what they were, it holds no particle)
        } else if (CDF WGS_ABLATE_AND(!(d.dbg & (1u << 21)))) {
            constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE;
            uint32_t mine = 0u;
            const uint32_t near = reach_mask(0xffffu);  // bit i: collider i can reach a node of this tile
            if (near == 0u) {
                // no collider in reach of the tile (nearly every block): its own 64 nodes get the "far" cdf — lane = node —
                // and the rim, which belongs to the neighbours, is theirs to write
                d.node_cdf[(size_t)id * NPB + (uint32_t)lane] = NodeCdf{1.0e10f, 0u, NONE, 0u};
            } else if (bcount == 0u) {   // (no particles: its own nodes, lane = node)
                float pt[D];
                int t[3] = {lane & (BW - 1), (lane >> BS) & (BW - 1), D == 3 ? (lane >> (2 * BS)) : 0};
#pragma unroll
                for (int k = 0; k < D; k++) pt[k] = (float)(b[k] * BW + t[k]) * d.h;
                const NodeCdf c = node_cdf_eval<D>(d, pt, near);
                d.node_cdf[(size_t)id * NPB + (uint32_t)lane] = c;
                mine |= c.affinities;
            } else
            for (int n = lane; n < ((TILE + 63) / 64) * 64; n += 64) {
                int t[3] = {n % TW, (n / TW) % TW, D == 3 ? n / (TW * TW) : 0};
                const int o = (t[0] >= BW ? 1 : 0) | (t[1] >= BW ? 2 : 0) | (t[2] >= BW ? 4 : 0);
                const uint32_t nb = __shfl(res, o & 7);        // "+" links live in lanes 0..7
                if (n < TILE && nb != NONE) {                  // nodes of blocks that are not active do not exist
                    float pt[D];
#pragma unroll
                    for (int k = 0; k < D; k++) pt[k] = (float)(b[k] * BW + t[k]) * d.h;
                    const NodeCdf c = node_cdf_eval<D>(d, pt, near);
                    if (o == 0) d.node_cdf[(size_t)id * NPB + (t[0] + (t[1] << BS) + (D == 3 ? (t[2] << (2 * BS)) : 0))] = c;
                    mine |= c.affinities;
                }
            }
            const bool any = __ballot(mine != 0u) != 0ull;
            if (lane == 0) {
                d.block_cpic[id] = any ? 1u : 0u;
                // (kept for the coming substeps only when no collider that moves is in reach)
                if (d.cdf_gen != 0u) d.block_cdf_gen[id] = (near & d.cdf_moving) == 0u ? (d.cdf_gen | (bcount > 0u ? CDF_FULL : 0u)) : 0u;
                if (any && bcount > 0u) d.cpic_list[(size_t)(id & 7u) * d.cap + atomicAdd(&d.counters[ctr_ncpic(id & 7u, epoch)], 1u)] = id;
            }
            listed = any && bcount > 0u;
            pc_flag = any ? CELL_LISTED : 0u;
        }
    } else {
        // The (BW+2)^D tile of a block = its own BW^D nodes + the first two layers of its "+" neighbours' — the same node is in the tiles of
        // up to 2^D blocks, and a wave that evaluated its whole tile (six rounds of 64 nodes against every collider in reach — 10 us, the
        // slowest waves of this launch wherever a collider MOVES: the reference's sand3) did 5.4 times the work there is. Every wave
        // evaluates its OWN nodes, one round, and publishes what its "-" neighbours want to know of them — Dev::block_cdf_summ: per offset o,
        // "some own node in the first BW+2-BW layers along the axes of o has an affinity" — under this substep's number; a particle-bearing
        // block then reads its "+" neighbours' words. A word that does not come within a few microseconds (the neighbour's wave is not
        // resident: more blocks than wave slots) is not waited for any longer: the wave evaluates that neighbour's part of its tile itself,
        // as before — same loop, same code. The class is the same either way: an affinity bit of a node does not depend on whose reach mask
        // it was evaluated under (a collider outside the mask gives no node of that tile an affinity).
        // SUMM is chosen by the host where blocks ARE evaluated substep after substep — a collider moves, or nothing keeps (Dev::cdf_gen 0).
        // With colliders at rest a block is evaluated once in its life and the words would serve nobody; that instantiation is the code
        // above, untouched: compiled into one kernel with it, this stage cost the launch 4 us of 57 at 16 M particles (registers).
        constexpr uint32_t SUMM_TAG = 0xffffffu;
        auto publish_summ = [&](uint32_t bits) {
            if (lane == 0) __hip_atomic_store(&d.block_cdf_summ[id], ((epoch & SUMM_TAG) << 8) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (cdf_cached) {
            const bool any = cdf_class != 0u;
            if (lane == 0 && any) d.cpic_list[(size_t)(id & 7u) * d.cap + atomicAdd(&d.counters[ctr_ncpic(id & 7u, epoch)], 1u)] = id;
            listed = any;
            pc_flag = any ? CELL_LISTED : 0u;
            publish_summ(cdf_summ & 0xffu);   // (its nodes are what they were)
        } else if (own_cached) {
            publish_summ(cdf_summ & 0xffu);   // (nothing else to do: its nodes are what they were, it holds no particle)
        } else if (CDF WGS_ABLATE_AND(!(d.dbg & (1u << 21)))) {
            constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE;
            constexpr int ROUNDS = (TILE + 63) / 64;
            const uint32_t near = reach_mask(0xffffu);  // bit i: collider i can reach a node of this tile
            bool any = false;
            if (near == 0u) {
                // no collider in reach of the tile (nearly every block): its own 64 nodes get the "far" cdf — lane = node —
                // and the rim, which belongs to the neighbours, is theirs to write
                d.node_cdf[(size_t)id * NPB + (uint32_t)lane] = NodeCdf{1.0e10f, 0u, NONE, 0u};
                publish_summ(0u);
            } else {
                uint32_t parts = 1u;   // bit o: the nodes the tile holds of the "+o" neighbour are to be evaluated here (0: the block's own)
                uint32_t mine = 0u;
#pragma unroll 1
                for (int it = 0; it <= ROUNDS; it++) {   // it = 0: the own nodes, lane = node; it >= 1: round it - 1 over the tile (fallback)
                    if (it == 1) {   // (wave-uniform)
                        uint32_t bits = 0u;
                        const int t0[3] = {lane & (BW - 1), (lane >> BS) & (BW - 1), D == 3 ? (lane >> (2 * BS)) : 0};
#pragma unroll
                        for (int o = 0; o < NN; o++) {
                            bool in = mine != 0u;
#pragma unroll
                            for (int k = 0; k < D; k++)
                                if ((o >> k) & 1) in = in && t0[k] < TW - BW;
                            bits |= __ballot(in) != 0ull ? (1u << o) : 0u;
                        }
                        publish_summ(bits);
                        any = (bits & 1u) != 0u;
                        parts = 0u;
                        if (!any && bcount > 0u) {   // (a block without particles: nobody asks for its class)
                            const uint32_t nb = lane >= 1 && lane < NN ? res : NONE;   // "+" links live in lanes 0..7
                            const unsigned long long t_start = wall_clock64();
                            for (;;) {
                                uint32_t w = 0u;
                                if (nb != NONE) w = __hip_atomic_load(&d.block_cdf_summ[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const bool have = nb != NONE && (w >> 8) == (epoch & SUMM_TAG);
                                any = __ballot(have && ((w >> lane) & 1u) != 0u) != 0ull;
                                parts = (uint32_t)__ballot(nb != NONE && !have);
                                if (any || parts == 0u || (d.dbg & 2u) || wall_clock64() - t_start > 400ull) break;   // (100 MHz: 4 us; WGS_DEBUG bit 1: no wait at all — tests)
                                __builtin_amdgcn_s_sleep(8);
                            }
                            if (any) parts = 0u;
                        }
                        if (parts == 0u) break;
                        mine = 0u;
                    }
                    int t[3], o = 0;
                    bool active = true;
                    if (it == 0) {
                        t[0] = lane & (BW - 1); t[1] = (lane >> BS) & (BW - 1); t[2] = D == 3 ? (lane >> (2 * BS)) : 0;
                    } else {
                        const int n = lane + 64 * (it - 1);
                        t[0] = n % TW; t[1] = (n / TW) % TW; t[2] = D == 3 ? n / (TW * TW) : 0;
                        o = (t[0] >= BW ? 1 : 0) | (t[1] >= BW ? 2 : 0) | (t[2] >= BW ? 4 : 0);
                        active = n < TILE && ((parts >> o) & 1u) != 0u;
                    }
                    if (active) {
                        float pt[D];
#pragma unroll
                        for (int k = 0; k < D; k++) pt[k] = (float)(b[k] * BW + t[k]) * d.h;
                        const NodeCdf c = node_cdf_eval<D>(cols, d.n_colliders, d.h, pt, near);
                        if (it == 0) d.node_cdf[(size_t)id * NPB + (uint32_t)lane] = c;
                        mine |= c.affinities;
                    }
                }
                if (parts != 0u) any = __ballot(mine != 0u) != 0ull;   // (the fallback ran)
            }
            if (lane == 0) {
                d.block_cpic[id] = any ? 1u : 0u;
                // (kept for the coming substeps only when no collider that moves is in reach)
                if (d.cdf_gen != 0u) d.block_cdf_gen[id] = (near & d.cdf_moving) == 0u ? (d.cdf_gen | (bcount > 0u ? CDF_FULL : 0u)) : 0u;
                if (any && bcount > 0u) d.cpic_list[(size_t)(id & 7u) * d.cap + atomicAdd(&d.counters[ctr_ncpic(id & 7u, epoch)], 1u)] = id;
            }
            listed = any && bcount > 0u;
            pc_flag = any ? CELL_LISTED : 0u;
        }
    }
    WGS_PROF(3)
    // ---- pass 1: members of the new run = stayers of the previous run + arrivals. Arrivals from OTHER blocks are on the
    // cell's list (sort.wgsl:129-137, for them only); a particle that changed cell INSIDE the block is on no list: the lane of
    // its old cell meets it here, in its own range of the previous run, and hands it to its new cell through LDS (a
    // counter and ARRC slots per cell; LDS atomics: the order of the slots is arbitrary, they are sorted by id below).
    // The first ARRC arrivals of a cell are kept in registers with their ids.
    constexpr int ARRC = 4;
    uint32_t a_slot[ARRC], a_pid[ARRC];
#pragma unroll
    for (int k = 0; k < ARRC; k++) { a_slot[k] = NONE; a_pid[k] = NONE; }
    uint32_t n_stay = ce_old - cs_old, n_arr = 0, n_in = 0;
    // -- the wave-parallel form: lane = entry of the previous run (strided). Every particle that names a cell of this block takes a
    // place in that cell (an LDS counter per cell: the order of arrival is arbitrary, the ids are put in order further down).
    static_assert(PAR_R % 2 == 0 && RUNCAP <= 1023, "two 16-bit (cell | place << 6) per register; 0xffff = none");
    uint32_t ckp[PAR_R / 2];  // per entry t = lane + 64 r: new cell | place in it << 6 in half r & 1 of word r / 2, 0xffff = not this block's
    uint32_t a_ck = NONE;     // ... of the lane's entry of the array of arrivals from other blocks
    // (two counters per cell: the particles that did NOT change cell — they come in the order of the cell's previous run, ascending ids —
    // take the first places, the others the places behind them: the ordering pass below then only has the newcomers to move)
    uint32_t *s_cnt = s_out + RUNCAP - 64, *s_cnt2 = s_out + RUNCAP - 128;   // (the output stage is not written before the counters are consumed)
    uint32_t par_new = 0u;    // bit r: entry lane + 64 r changed cell (its place counts from behind the stayers of its new cell)
    uint32_t par_nst = 0u;    // lane = cell: members of the new run that stayed
    if (par) {
        __hip_atomic_store(&s_cnt[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&s_cnt2[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
        for (int r = 0; r < PAR_R; r++) {
            const uint32_t t = (uint32_t)lane + 64u * (uint32_t)r;
            uint32_t ck = 0xffffu;
            if (t < runlen) {
                const uint32_t e = __hip_atomic_load(&s_in[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const uint32_t c = cell_of(e);
                if ((c >> 6) == id) {   // (NONE, a particle left out of the sort, is nobody's)
                    const bool came = (e & CELL_MOVED) != 0u;
                    ck = (c & 63u) | (__hip_atomic_fetch_add(came ? &s_cnt2[c & 63u] : &s_cnt[c & 63u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) << 6);
                    par_new |= came ? (1u << r) : 0u;
                }
            }
            if (r & 1) ckp[r / 2] |= ck << 16;
            else ckp[r / 2] = ck;
        }
        if (a_cell != NONE) a_ck = (a_cell & 63u) | (__hip_atomic_fetch_add(&s_cnt2[a_cell & 63u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) << 6);
        par_nst = __hip_atomic_load(&s_cnt[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        n_stay = par_nst + __hip_atomic_load(&s_cnt2[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // (all members of the cell's new run)
        n_arr = 0;
    } else {
        static_assert(RUNCAP >= 64 + 2 * 64 * ARRC, "the hand-over arrays alias the output stage");
        uint32_t *s_acnt = s_out, *s_aslot = s_out + 64, *s_apid = s_out + 64 + 64 * ARRC;   // (aliases: read into registers before pass 2 writes s_out)
        auto hand_over = [&](uint32_t c, uint32_t slot, uint32_t pid) {   // to cell c of this block
            const uint32_t k = __hip_atomic_fetch_add(&s_acnt[c & 63u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (k < (uint32_t)ARRC) {
                __hip_atomic_store(&s_aslot[(c & 63u) * ARRC + k], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_store(&s_apid[(c & 63u) * ARRC + k], pid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        };
        if (!clean || narr_in != 0u) {
            __hip_atomic_store(&s_acnt[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (!clean) {
                n_stay = 0;
                for (uint32_t i = cs_old; i < ce_old; i++) {
                    const uint32_t c = new_cell_of(i);
                    if (c == idx) n_stay++;
                    else if ((c >> 6) == id) hand_over(c, i, pid_of_old(i));   // (NONE, a particle left out of the sort, is nobody's)
                }
            }
            if (a_cell != NONE) hand_over(a_cell, a_ent, a_epid);   // the arrivals from other blocks, from the block's array
            n_in = __hip_atomic_load(&s_acnt[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // (the wave ran all of the above in every lane)
#pragma unroll
            for (int k = 0; k < ARRC; k++)
                if ((uint32_t)k < n_in) {
                    a_slot[k] = __hip_atomic_load(&s_aslot[lane * ARRC + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    a_pid[k] = __hip_atomic_load(&s_apid[lane * ARRC + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            n_arr = n_in;
        }
        for (uint32_t a = head; a != 0u;) {
            const uint32_t nxt = d.mv_next[a - 1u], p = ldpid<D>(in, d.npad, a - 1u);
#pragma unroll
            for (int k = 0; k < ARRC; k++)
                if (n_arr == (uint32_t)k) { a_slot[k] = a - 1u; a_pid[k] = p; }
            n_arr++;
            a = nxt;
        }
    }   // (!par)
    if (n_arr > 1u && n_arr <= (uint32_t)ARRC) {  // ascending id (empty entries hold NONE = the largest value)
#pragma unroll
        for (int pass = 0; pass < ARRC; pass++)
#pragma unroll
            for (int k = pass & 1; k + 1 < ARRC; k += 2)
                if (a_pid[k] > a_pid[k + 1]) {
                    const uint32_t tp = a_pid[k], ts = a_slot[k];
                    a_pid[k] = a_pid[k + 1]; a_slot[k] = a_slot[k + 1];
                    a_pid[k + 1] = tp; a_slot[k + 1] = ts;
                }
    }
    const uint32_t total = n_stay + n_arr;
    const uint32_t inc = wave_scan_incl_u32(total);
    const uint32_t btotal = lane_value(inc, 63);
    const uint32_t lstart = inc - total;  // start of this cell's run inside the block
    WGS_PROF(2)
    // ---- pass 2: merge in ascending particle id. Stayers are in id order already; the next arrival is selected
    // from the (short) list: smallest id above the last one taken.
    const bool out_lds = btotal <= (uint32_t)RUNCAP;
    // the scan's result for this block (first_particle) is only needed where global memory is written: a block that
    // fits the LDS stage asks for it after the merge
    uint32_t bstart = 0, aidx = 0;
    auto fetch_bstart = [&]() { block_prefix(d, epoch, id, lane, grp, nphys, bstart, aidx); };
    if (!out_lds) fetch_bstart();
    // Nothing moved, nothing arrived and the block's run starts where it started one substep ago, when the same was found: the
    // block's part of perm is the identity and its part of perm_cell what it was — both are left alone (the first such substep
    // writes them, the following ones only renew the note).
    const bool fast_clean = old_ok && clean && !any_arr;
    bool keep_perm = false;
    if (fast_clean) {
        if (out_lds) fetch_bstart();
        keep_perm = bstart == bstart_old && ident == (((epoch - 1u) << 1) | (pc_flag != 0u ? 1u : 0u));
    }
    uint32_t *s_lst = s_in, *s_av = s_in + 64, *s_new = s_in + 128;   // (wave-parallel form: the staged cell ids are consumed by now)
    if (par) {
        // (places) the cell's start + the place taken in it — behind the cell's stayers for a newcomer; an entry of the output stage =
        // cell << 26 | index v into the ids: v < runlen: entry v of the previous run, else entry v - runlen of the array of arrivals
        __hip_atomic_store(&s_lst[lane], lstart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&s_new[lane], lstart + par_nst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (a_ck != NONE) {
            __hip_atomic_store(&s_av[lane], a_ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_store(&s_pid[runlen + (uint32_t)lane], a_epid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
#pragma unroll
        for (int r = 0; r < PAR_R; r++) {
            const uint32_t ck = (ckp[r / 2] >> (16 * (r & 1))) & 0xffffu;
            if (ck != 0xffffu) {
                const uint32_t pos = __hip_atomic_load(((par_new >> r) & 1u) ? &s_new[ck & 63u] : &s_lst[ck & 63u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + (ck >> 6);
                __hip_atomic_store(&s_out[pos], ((ck & 63u) << 26) | ((uint32_t)lane + 64u * (uint32_t)r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        if (a_ck != NONE) {
            const uint32_t pos = __hip_atomic_load(&s_new[a_ck & 63u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + (a_ck >> 6);
            __hip_atomic_store(&s_out[pos], ((a_ck & 63u) << 26) | (runlen + (uint32_t)lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        // (order) lane = cell: its members by ascending (id, v). The stayers hold the first places in the order of the previous run —
        // ascending ids, checked in one pipelined pass —, the newcomers the places behind them. With up to NEWC newcomers (nearly
        // every cell) those are sorted in registers, each finds its rank among the stayers by bisection (the searches of a lane
        // advance together: their loads are independent), the stayers above the lowest rank move up by the number of newcomers
        // below them, and the newcomers drop into the gaps: about a dozen dependent LDS accesses per cell, where inserting
        // newcomer after newcomer cost the wave the longest insertion of its 64 cells at every step (the stirred cube: 10 of a dirty
        // block's 26 us). Any other case — more newcomers, stayers not in order — takes the insertion sort below, which is correct
        // for any order of the places.
        constexpr int NEWC = 4;
        constexpr uint32_t VM = 0x03ffffffu;
        // (plain LDS accesses in this pass: a lane touches the range of its own cell only — what other lanes wrote there, they wrote with
        // the atomic stores above, which a load of a possibly aliasing address is not moved across —, so the loads of a batch can
        // be issued together)
        auto lds = [&](const uint32_t *ptr) { return *ptr; };
        bool ordered = true;
        {
            uint32_t pk = 0u, pv = 0u;
#pragma unroll 4
            for (uint32_t m = 0; m < par_nst; m++) {
                const uint32_t v = lds(&s_out[lstart + m]) & VM, k = lds(&s_pid[v]);
                ordered = ordered && (m == 0u || pk < k || (pk == k && pv < v));
                pk = k;
                pv = v;
            }
        }
        const bool quick = ordered && !(d.dbg & 16777216u);   // (WGS_DEBUG bit 24: the insertion sort below for every cell — same order, tested)
        // (batches of NEWC newcomers: the members merged so far are the sorted prefix of the next batch)
        for (uint32_t base = par_nst; quick && base < total; base += (uint32_t)NEWC) {
            const uint32_t par_nst = base, n_new = min((uint32_t)NEWC, total - base);   // (this batch: `par_nst` sorted members, `n_new` newcomers)
            uint32_t nk[NEWC], nv[NEWC], rk[NEWC];
#pragma unroll
            for (int k = 0; k < NEWC; k++) {
                nk[k] = nv[k] = NONE;   // (empty entries sort last)
                if ((uint32_t)k < n_new) {
                    nv[k] = lds(&s_out[lstart + par_nst + (uint32_t)k]) & VM;
                    nk[k] = lds(&s_pid[nv[k]]);
                }
            }
#pragma unroll
            for (int pass = 0; pass < NEWC; pass++)
#pragma unroll
                for (int k = pass & 1; k + 1 < NEWC; k += 2)
                    if (nk[k] > nk[k + 1] || (nk[k] == nk[k + 1] && nv[k] > nv[k + 1])) {
                        const uint32_t tk = nk[k], tv = nv[k];
                        nk[k] = nk[k + 1]; nv[k] = nv[k + 1];
                        nk[k + 1] = tk; nv[k + 1] = tv;
                    }
            // rank of each newcomer = number of stayers with a smaller key
            uint32_t lo[NEWC], hi[NEWC];
#pragma unroll
            for (int k = 0; k < NEWC; k++) { lo[k] = 0u; hi[k] = (uint32_t)k < n_new ? par_nst : 0u; }
            for (;;) {
                bool more = false;
                uint32_t mid[NEWC], mv[NEWC], mk[NEWC];
#pragma unroll
                for (int k = 0; k < NEWC; k++) {
                    mid[k] = (lo[k] + hi[k]) >> 1;
                    mv[k] = lo[k] < hi[k] ? (lds(&s_out[lstart + mid[k]]) & VM) : 0u;
                }
#pragma unroll
                for (int k = 0; k < NEWC; k++) mk[k] = lo[k] < hi[k] ? lds(&s_pid[mv[k]]) : 0u;
#pragma unroll
                for (int k = 0; k < NEWC; k++)
                    if (lo[k] < hi[k]) {
                        if (mk[k] < nk[k] || (mk[k] == nk[k] && mv[k] < nv[k])) lo[k] = mid[k] + 1u;
                        else hi[k] = mid[k];
                        more = more || lo[k] < hi[k];
                    }
                if (!more) break;
            }
#pragma unroll
            for (int k = 0; k < NEWC; k++) rk[k] = lo[k];
            // the stayers from the top down to the lowest rank move up by the number of newcomers that rank at or below them
            for (uint32_t t = par_nst; t > rk[0]; t--) {
                const uint32_t at = t - 1u;
                uint32_t up = 0u;
#pragma unroll
                for (int k = 0; k < NEWC; k++) up += ((uint32_t)k < n_new && rk[k] <= at) ? 1u : 0u;
                s_out[lstart + at + up] = lds(&s_out[lstart + at]);
            }
#pragma unroll
            for (int k = 0; k < NEWC; k++)
                if ((uint32_t)k < n_new)
                    s_out[lstart + rk[k] + (uint32_t)k] = ((uint32_t)lane << 26) | nv[k];
        }
        uint32_t kprev = 0u, vprev = 0u;
        if (!quick && total > 1u) {
            vprev = __hip_atomic_load(&s_out[lstart], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) & 0x03ffffffu;
            kprev = __hip_atomic_load(&s_pid[vprev], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        for (uint32_t m = 1; m < (quick ? 0u : total); m++) {   // (the insertion sort: cells the quick form above does not cover)
            const uint32_t e = __hip_atomic_load(&s_out[lstart + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const uint32_t ve = e & 0x03ffffffu, ke = __hip_atomic_load(&s_pid[ve], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (kprev < ke || (kprev == ke && vprev < ve)) {   // in order behind the member before it (which stays the last one so far)
                kprev = ke;
                vprev = ve;
                continue;
            }
            // (moved down: the member before it ends up at place m, so the key kept in the registers is still that of the last place)
            uint32_t j = m;
            while (j > 0u) {
                const uint32_t q = __hip_atomic_load(&s_out[lstart + j - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const uint32_t vq = q & 0x03ffffffu, kq = __hip_atomic_load(&s_pid[vq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (kq < ke || (kq == ke && vq < ve)) break;
                __hip_atomic_store(&s_out[lstart + j], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                j--;
            }
            if (j != m) __hip_atomic_store(&s_out[lstart + j], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    uint32_t out = lstart;
    auto emit = [&](uint32_t src) {
        if (out_lds) {
            __hip_atomic_store(&s_out[out], ((uint32_t)lane << 26) | src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
            d.perm[bstart + out] = src;
            d.perm_cell[bstart + out] = idx | pc_flag;
        }
        out++;
    };
    uint32_t arr_slot = NONE, arr_pid = 0, arr_k = 0;
    bool have_last = false;
    uint32_t last_pid = 0, last_slot = 0;
    const bool cached = n_arr <= (uint32_t)ARRC;
    // A cell with more arrivals than fit the registers (rare in a steady flow) is sorted by the WHOLE wave further down when the
    // block fits the LDS stages and the cell holds at most 64 particles; its lane sits the merge out.
    const bool coop_ok = in_lds && out_lds;   // (wave-uniform)
    const bool coop_me = !par && !cached && coop_ok && total <= 64u;
    auto next_arrival = [&]() {
        arr_slot = NONE;
        if (cached) {  // k-th entry of the sorted register copy
#pragma unroll
            for (int k = 0; k < ARRC; k++)
                if (arr_k == (uint32_t)k && (uint32_t)k < n_arr) { arr_slot = a_slot[k]; arr_pid = a_pid[k]; }
            arr_k++;
            return;
        }
        // more than ARRC arrivals (rare): the smallest (id, slot) above the last one taken, among the list (other blocks)
        // and the particles of the block's previous run that name this cell as their new one (other cells of this block).
        // Ids are unique in a healthy run; after a reported grid overflow the buffer can hold the same particle twice, and
        // the merge must still place every arrival exactly once (slots are unique).
        unsigned long long best = ~0ull;
        const unsigned long long last_key = ((unsigned long long)last_pid << 32) | last_slot;
        for (uint32_t a = head; a != 0u; a = d.mv_next[a - 1u]) {
            const unsigned long long key = ((unsigned long long)ldpid<D>(in, d.npad, a - 1u) << 32) | (a - 1u);
            if (have_last && key <= last_key) continue;
            if (key < best) best = key;
        }
        if (n_in != 0u) {
            if (!clean)
                for (uint32_t i = run0; i < run1; i++) {
                    if ((i >= cs_old && i < ce_old) || new_cell_of(i) != idx) continue;
                    const unsigned long long key = ((unsigned long long)pid_of_old(i) << 32) | i;
                    if (have_last && key <= last_key) continue;
                    if (key < best) best = key;
                }
            for (uint32_t t = 0; t < narr_in; t++) {   // (the block's array of arrivals from other blocks)
                const uint32_t sl = d.blk_arr[(size_t)id * BLK_ARR + t];
                if (cell_of(d.cellid[sl]) != idx) continue;
                const unsigned long long key = ((unsigned long long)ldpid<D>(in, d.npad, sl) << 32) | sl;
                if (have_last && key <= last_key) continue;
                if (key < best) best = key;
            }
        }
        if (best != ~0ull) {
            arr_slot = (uint32_t)best;
            arr_pid = (uint32_t)(best >> 32);
        }
    };
    if (n_arr && !coop_me) next_arrival();
    for (uint32_t i = cs_old; i < ((coop_me || par || keep_perm) ? cs_old : ce_old); i++) {
        if (!stays_here(i)) continue;
        if (arr_slot != NONE) {
            const uint32_t ps = pid_of_old(i);
            while (arr_slot != NONE && arr_pid < ps) {
                emit(arr_slot);
                have_last = true;
                last_pid = arr_pid;
                last_slot = arr_slot;
                next_arrival();
            }
        }
        emit(i);
    }
    while (arr_slot != NONE) {
        emit(arr_slot);
        have_last = true;
        last_pid = arr_pid;
        last_slot = arr_slot;
        next_arrival();
    }
    // ---- cells with many arrivals, one at a time, by the whole wave: (1) the members of the new run — the particles of the
    // block's previous run that name the cell (stayers and arrivals from other cells alike: one coalesced look at the staged
    // cell ids) and the list of arrivals from other blocks — go to the cell's range of the output stage in any order;
    // (2) lane m holds member m with its id and counts the members with a smaller (id, slot): its place in canonical order.
    for (unsigned long long slow = __ballot(coop_me); slow != 0ull; slow &= slow - 1ull) {
        const int c = __ffsll((long long)slow) - 1;
        const uint32_t c_idx = id * NPB + (uint32_t)c;
        const uint32_t c_cs = __shfl(cs_old, c), c_ce = __shfl(ce_old, c), c_head = __shfl(head, c), c_lstart = __shfl(lstart, c), c_total = __shfl(total, c);
        uint32_t cnt = 0u;
        for (uint32_t t0 = 0; t0 < runlen; t0 += 64u) {
            const uint32_t t = t0 + (uint32_t)lane, i = run0 + t;
            const bool m = t < runlen && (clean ? (i >= c_cs && i < c_ce) : new_cell_of(i) == c_idx);
            const unsigned long long mm = __ballot(m);
            if (m) {
                const uint32_t at = c_lstart + cnt + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                if (at < c_lstart + c_total) __hip_atomic_store(&s_out[at], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            cnt += (uint32_t)__popcll(mm);
        }
        {   // the arrivals from other blocks that name the cell: from the block's array (lane = entry) ...
            const bool m = a_cell == c_idx;
            const unsigned long long mm = __ballot(m);
            if (m) {
                const uint32_t at = c_lstart + cnt + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                if (at < c_lstart + c_total) __hip_atomic_store(&s_out[at], a_ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            cnt += (uint32_t)__popcll(mm);
        }
        if (lane == 0)   // ... and from the cell's list
            for (uint32_t a = c_head; a != 0u && cnt < c_total; a = d.mv_next[a - 1u]) {
                __hip_atomic_store(&s_out[c_lstart + cnt], a - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                cnt++;
            }
        // (c_total = stayers + arrivals as counted in pass 1: the same particles)
        unsigned long long key = ~0ull;
        if ((uint32_t)lane < c_total) {
            const uint32_t sl = __hip_atomic_load(&s_out[c_lstart + (uint32_t)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const uint32_t pd = (sl >= run0 && sl < run1) ? pid_of_old(sl) : ldpid<D>(in, d.npad, sl);
            key = ((unsigned long long)pd << 32) | sl;
        }
        uint32_t rank = 0u;
        for (uint32_t m = 0; m < c_total; m++) rank += __shfl(key, (int)m) < key ? 1u : 0u;
        if ((uint32_t)lane < c_total)
            __hip_atomic_store(&s_out[c_lstart + rank], ((uint32_t)c << 26) | (uint32_t)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    WGS_PROF(4)
    if (out_lds && !fast_clean) fetch_bstart();
    WGS_PROF(5)
    if (out_lds && !keep_perm)
        for (uint32_t t = lane; t < btotal; t += 64) {
            const uint32_t v = __hip_atomic_load(&s_out[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            uint32_t src = v & 0x03ffffffu;
            if (par) src = src < runlen ? run0 + src : __hip_atomic_load(&s_av[src - runlen], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            d.perm[bstart + t] = src;
            d.perm_cell[bstart + t] = (id * NPB + (v >> 26)) | pc_flag;
        }
    // ---- new runs, reset of what this substep consumed
    d.cell_start[idx] = bstart + lstart;
    d.cell_cursor[idx] = bstart + lstart + total;
    // (the same, where P2G finds it without the block id — or, for a run that is the block's previous run member for member, where the
    // particles ARE: [cs_old, ce_old) of the buffer, which P2G then reads without the gather through perm: layout.h CELL_DIRECT.
    // WGS_DEBUG bit 25: never — same particles in the same order either way, tested)
    const bool direct = fast_clean && !(d.dbg & 33554432u);
    d.act_cells[(size_t)aidx * NPB + lane] = direct ? make_uint2(cs_old, ce_old) : make_uint2(bstart + lstart, bstart + lstart + total);
    if (head != 0u) d.cell_head[idx] = 0u;
    if (lane == 0 && narr != 0u) d.blk_narr[id] = 0u;
    if (d.n_rigid != 0u) {             // mesh-collider cdf accumulators of this substep (k_p2g_cdf)
        d.mesh_min[idx] = ~0ull;
        d.mesh_aff[idx] = 0u;
    }
    // the slabs the grid update gathers this block's nodes from: its "-" neighbours that hold particles
    if (lane >= 8 && lane < 16) d.act_src[aidx * 8u + (uint32_t)(lane & 7)] = link_cnt > 0u ? res : NONE;
    if (lane == 63) {
        d.active[aidx] = id;           // grid.wgsl:323-334: the active list, in physical-id order
        d.act_info[aidx] = make_uint4(id, bkey, btotal, pc_flag | (direct ? CELL_DIRECT : 0u));   // (.w: the block class, where this launch computes it; direct runs)
        d.block_start[id] = bstart;    // first_particle
        d.block_count[id] = btotal;    // snapshot used by P2G / grid update / G2P
        d.links_epoch[id] = epoch;     // the neighbour links and cell runs written above are those of this substep
        d.block_ident[id] = (fast_clean && bstart == bstart_old) ? ((epoch << 1) | (pc_flag != 0u ? 1u : 0u)) : 0u;
        // (block_acc is cleared by the grid update: the waves of this group read it)
    }
    if (listed) {
        append_visits(d, id, bstart, btotal, lane, epoch);
        if (lane == 0) d.pcdf_done[id] = 0u;   // (counted up by the prologue waves of this substep's P2G launch, if it has any)
    }
    if constexpr (SHARD) {  // block layers that travel to a neighbour: an entry k_pack_face can work from without another look-up —
        // [id, key, the 2^D slabs the block's nodes are gathered from (its "-" neighbours that hold particles: lanes 8..15)]
        const IfaceMasks im = iface_masks<D>(d, b[0]);
        if ((im.send_lo | im.send_hi) != 0u) {  // wave-uniform
            uint32_t e = 0u;
            if (lane == 0) e = atomicAdd(&d.counters[ctr_nhalo(epoch)], 1u);
            e = __shfl(e, 0);
            if (e < d.cap) {
                uint32_t *ent = d.halo_list + (size_t)e * HALO_ENT;
                if (lane == 0) { ent[0] = id; ent[1] = bkey; }
                if (lane >= 8 && lane < 16) ent[2 + (lane & 7)] = link_cnt > 0u ? res : NONE;
            }
        }
    }
    WGS_PROF(6)
    WGS_PROF_END()
    uint32_t movers = 0u;
    if (have_old) {   // (every particle is an arrival on a table-rebuild substep)
        if (par) {
            movers = wave_sum_u32(n_moved) + narr_in;
        } else {
            movers = wave_sum_u32(n_arr);
        }
    }
    return movers;
}

// Launch 2: the first `nscan` workgroups scan, the others regroup (one wave per active block, strided over the
// physical ids). The scan workgroups have the lowest indices, so they are resident before any wave can wait for them.
// (4 waves per SIMD = 128 VGPRs: with its 36 KB of LDS that is the 4 resident workgroups per CU the launch is sized for)
#ifndef WGS_REGROUP_WPE
#define WGS_REGROUP_WPE 4
#endif
// SHARD: the data is one slab of a decomposition (a template parameter: the single-domain kernel carries no register for it)
template <int D, bool CDF, bool SHARD = false, bool SUMM = false> __global__ __launch_bounds__(SORT_THREADS, WGS_REGROUP_WPE) void k_regroup(Dev d, int side, uint32_t epoch, uint32_t nscan, int have_old) {
    __shared__ unsigned long long s_wave[SORT_THREADS / 64];
    __shared__ unsigned long long s_bcast;
    __shared__ uint32_t s_in[SORT_THREADS / 64][RUNCAP], s_out[SORT_THREADS / 64][RUNCAP], s_pid[SORT_THREADS / 64][RUNCAP];
    if (blockIdx.x < nscan) {
        __builtin_amdgcn_s_setprio(3);  // every regrouping wave ends up waiting for these few workgroups
        scan_chunk<SHARD>(d, epoch, blockIdx.x, nscan, s_wave, &s_bcast);
        return;
    }
    // (the colliders' poses and shapes, for the node cdf: a copy per workgroup — from global memory every collider is a round trip of its
    // own in front of the first node, six of them in the reference's sand3)
    // Filled by the first wave of the workgroup that needs it and by every other that does (regroup_block: the same words, no barrier —
    // most launches need none: the blocks' cdfs keep from substep to substep unless a collider moves).
    __shared__ ColliderDev s_cols[SUMM ? 16 : 1];
    bool cols_staged = false;
    const uint32_t nphys = min(d.counters[CTR_NPHYS], d.cap);
    const uint32_t wave = ((blockIdx.x - nscan) * SORT_THREADS + threadIdx.x) >> 6;
    const uint32_t nwaves = ((gridDim.x - nscan) * SORT_THREADS) >> 6;
    const int w = SUMM ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);   // (uniform: the wave's LDS stages are scalar bases, not per-lane addresses held for the whole launch)
    // no block id was handed out since launch 2 of the previous substep? Then a neighbour that was not in the table is
    // still not in it (rim of the active region: the lookups with the longest probe sequences, every substep)
    const bool no_new_blocks = d.counters[CTR_NINSERT] == d.counters[CTR_NPHYS_SEEN + ((epoch - 1u) & 1u)];
    // Evictions happen in the substeps whose number is a multiple of EVICT_AGE only (regroup_block): an id noted as "in the table" during
    // such a launch may have been evicted by another wave of the same launch, so the launch AFTER it compares the keys of the ids it
    // inherits; every other launch inherits ids that were compared or looked up one substep ago and cannot have been evicted since.
    // (A counter of evictions sampled inside the launches does not say the same: the wave that samples it is not the first to run.)
    const bool check_keys = ((epoch - 1u) & (EVICT_AGE - 1u)) == 0u;
    uint32_t movers = 0u;
    for (uint32_t id = wave; id < nphys; id += nwaves)
        movers += regroup_block<D, CDF, SHARD, SUMM>(d, side, epoch, id, nphys, have_old != 0, no_new_blocks, check_keys, s_in[w], s_out[w], s_pid[w], s_cols, cols_staged);
    // statistics (wgs_stats.cell_changers): one add per workgroup, the partial counts in cache lines of their own
    if ((threadIdx.x & 63u) == 0u) s_wave[w] = movers;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long m = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (m != 0ull) atomicAdd(&d.counters[CTR_MOVERS + 32u * (blockIdx.x & 15u)], (uint32_t)m);
    }
}

// The table without its marks (capi.hip, when the marks of evicted blocks crowd it): the two arrays were cleared, every block that is
// still in the table — block_key is NONE for an evicted id — is inserted again UNDER ITS OWN ID. Nothing else changes: ids, cell ids,
// links and the particles' order stay what they are, the steady-state sort goes on. (A full rebuild bins every particle again through
// the table and regroups everything: ~0.65 ms at 1 M particles against a few microseconds here.) Alone on the stream: nobody else
// looks at the table meanwhile.
__global__ __launch_bounds__(256) void k_table_refresh(Dev d) {
    const uint32_t nphys = min(d.counters[CTR_NPHYS], d.cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) d.counters[CTR_NTOMB] = 0u;
    for (uint32_t id = blockIdx.x * 256u + threadIdx.x; id < nphys; id += gridDim.x * 256u) {
        const uint32_t key = d.block_key[id];
        if (key == NONE) continue;   // (on the free list)
        uint32_t slot = hash_key(key) & d.hmask;
        for (uint32_t probe = 0; probe <= d.hmask; ++probe) {
            if (atomicCAS(&d.hkeys[slot], NONE, key) == NONE) {
                d.hvals[slot] = id;
                d.block_slot[id] = slot;
                break;
            }
            slot = (slot + 1u) & d.hmask;
        }
    }
}

// Test hook (wgs_debug_scan): the scan workgroups, and workgroups that finish it per block exactly like the
// regrouping waves do (block_prefix), writing block_start only.
__global__ __launch_bounds__(SORT_THREADS) void k_scan_only(Dev d, uint32_t epoch, uint32_t nscan) {
    __shared__ unsigned long long s_wave[SORT_THREADS / 64];
    __shared__ unsigned long long s_bcast;
    if (blockIdx.x < nscan) {
        scan_chunk(d, epoch, blockIdx.x, nscan, s_wave, &s_bcast);
        return;
    }
    const uint32_t nphys = min(d.counters[CTR_NPHYS], d.cap);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = ((blockIdx.x - nscan) * SORT_THREADS + threadIdx.x) >> 6;
    const uint32_t nwaves = ((gridDim.x - nscan) * SORT_THREADS) >> 6;
    for (uint32_t id = wave; id < nphys; id += nwaves) {
        const GroupLoads g = block_prefix_loads(d, id, lane);
        uint32_t start = 0, aidx = 0;
        block_prefix(d, epoch, id, lane, g, nphys, start, aidx);
        if (lane == 0) {
            d.block_start[id] = start;
            d.active[aidx] = id;
        }
    }
}

}  // namespace wgs
