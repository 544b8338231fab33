// kernels_sort.h — "grid sort" pass: sparse-grid activation and the counting
// sort of particles by (block, cell, particle id).
//
// Replaces the reference's 14 dispatches of WgGrid::queue_sort
// (grid/grid.rs:30-207; kernels grid.wgsl:186-203,355-379, sort.wgsl:26-36,89-137,
// prefix_sum.wgsl:11-93). Same observable result — hash map of active blocks
// (a particle's block and its "+1" neighbours), per-block first_particle /
// num_particles, particle ids grouped by block — but:
//   * no per-node linked lists: cells become contiguous ranges of `perm`;
//   * the order inside a cell is canonical (ascending persistent particle id),
//     so every downstream fp32 sum is reproducible run to run (the reference's
//     order is decided by atomic races, sort.wgsl:126,133);
//   * wave64-level de-duplication: one hash probe per distinct block per wave
//     instead of 8 CAS loops per particle.
#pragma once
#include "device_math.h"

namespace wgs {

constexpr int SORT_THREADS = 256;

template <int D> __device__ inline void load_cell(const float *in, uint32_t npad, uint32_t i, float h, int *cell) {
    const float4 xm = ldq(in, npad, Pl<D>::XM, i);
    cell[0] = assoc_cell(xm.x, h);
    cell[1] = assoc_cell(xm.y, h);
    if constexpr (D == 3) cell[2] = assoc_cell(xm.z, h);
}

// sort.wgsl:26-36 touch_particle_blocks + grid.wgsl:323-334 mark_block_as_active.
// Workgroup-level de-duplication: the distinct blocks of the 256 particles are collected
// in a small LDS set (particles arrive block-sorted from the previous substep, so there
// are 1-3 of them), then 8 lanes per distinct block probe the global hash map. Global
// atomics happen only for genuinely new blocks.
constexpr int TOUCH_SET = 32;
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_touch_blocks(Dev d, int side, uint32_t epoch) {
    constexpr int BS = Dim<D>::BSHIFT;
    constexpr int NN = Dim<D>::NNBR;
    __shared__ uint32_t s_keys[TOUCH_SET];
    const float *in = d.buf[side];
    const int tid = threadIdx.x;
    if (tid < TOUCH_SET) s_keys[tid] = NONE;
    __syncthreads();
    uint32_t i = blockIdx.x * SORT_THREADS + tid;
    bool valid = i < d.n;
    int b[3] = {0, 0, 0};
    uint32_t key = NONE;
    if (valid) {
        int c[D];
        load_cell<D>(in, d.npad, i, d.h, c);
#pragma unroll
        for (int k = 0; k < D; k++) b[k] = c[k] >> BS;  // floor(cell / BW)
        // the block and its +1 neighbours must all be representable
        int hi[D];
#pragma unroll
        for (int k = 0; k < D; k++) hi[k] = b[k] + 1;
        if (!block_in_key_range<D>(b) || !block_in_key_range<D>(hi)) {
            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_KEYRANGE);
            valid = false;
        } else {
            key = pack_key<D>(b);
        }
    }
    const int lane = tid & 63;
    if (d.dbg & 2u) { if (valid) d.cellid[i] = key; return; }  // ablation: load + cell math only
    unsigned long long todo = __ballot(valid);
    while (todo) {  // wave-uniform: one iteration per distinct block in the wave
        int leader = __ffsll((long long)todo) - 1;
        uint32_t k0 = __shfl(key, leader);
        todo &= ~__ballot(valid && key == k0);
        if (lane == leader && !(d.dbg & 8u)) {
            // insert into the workgroup's LDS set; on overflow fall back to direct activation
            uint32_t slot = hash_key(k0) & (TOUCH_SET - 1);
            bool placed = false;
            for (int probe = 0; probe < TOUCH_SET; probe++) {
                uint32_t old = atomicCAS(&s_keys[slot], NONE, k0);
                if (old == NONE || old == k0) { placed = true; break; }
                slot = (slot + 1) & (TOUCH_SET - 1);
            }
            if (!placed) {
                for (int o = 0; o < NN; o++) {
                    int nb[3] = {b[0] + (o & 1), b[1] + ((o >> 1) & 1), b[2] + ((o >> 2) & 1)};
                    activate_block(d, pack_key<D>(nb), epoch);
                }
            }
        }
    }
    __syncthreads();
    if (d.dbg & 1u) return;  // ablation: LDS part only
    // grid.wgsl:300-320: the 2^D blocks {b, b+1} per axis of every distinct block, one per thread
    {
        const int slot = tid >> 3, o = tid & 7;
        const uint32_t k0 = s_keys[slot];
        if (k0 != NONE && o < NN) {
            int lb[3] = {0, 0, 0};
            unpack_key<D>(k0, lb);
            int nb[3] = {lb[0] + (o & 1), lb[1] + ((o >> 1) & 1), lb[2] + ((o >> 2) & 1)};
            activate_block(d, pack_key<D>(nb), epoch);
        }
    }
}

// grid.wgsl:323-334 mark_block_as_active, second half: number the occupied hash slots.
// Exclusive prefix sum of "slot occupied" over the table -> dense block id (in slot order,
// hence reproducible), hvals[slot] = id, block_key[id] = key, num_active_blocks = total.
constexpr int ASSIGN_THREADS = 1024;
constexpr int ASSIGN_ITEMS = 8;
__global__ __launch_bounds__(ASSIGN_THREADS) void k_assign_block_ids(Dev d, uint32_t epoch) {
    __shared__ uint32_t wave_sums[ASSIGN_THREADS / 64];
    __shared__ uint32_t carry_s;
    const uint32_t hcap = d.hmask + 1u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < hcap; base += ASSIGN_THREADS * ASSIGN_ITEMS) {
        uint32_t keys[ASSIGN_ITEMS];
        uint32_t sum = 0;
        const uint32_t first = base + (uint32_t)tid * ASSIGN_ITEMS;
#pragma unroll
        for (int k = 0; k < ASSIGN_ITEMS; k++) {
            keys[k] = (first + k < hcap && d.hstamp[first + k] == epoch) ? d.hkeys[first + k] : NONE;
            sum += keys[k] != NONE;
        }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += wave_sums[w];
        uint32_t run = carry_s + wave_off + inc - sum;
#pragma unroll
        for (int k = 0; k < ASSIGN_ITEMS; k++) {
            if (keys[k] != NONE) {
                if (run < d.cap) {
                    d.hvals[first + k] = run;
                    d.block_key[run] = keys[k];
                } else {
                    d.hvals[first + k] = NONE;
                }
                run++;
            }
        }
        __syncthreads();
        if (tid == ASSIGN_THREADS - 1) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) {
        d.counters[CTR_NBLOCKS] = carry_s;
        if (carry_s > d.cap) atomicOr(&d.counters[CTR_ERRORS], ERRBIT_OVERFLOW);
    }
}

// Neighbour links of every active block (replaces the per-thread hash lookups of
// p2g.wgsl:238-275, g2p.wgsl:72-132) and reset of the per-block particle counter.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_block_links(Dev d, uint32_t epoch) {
    constexpr int NN = Dim<D>::NNBR;
    uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    uint32_t total = B * 16u;
    for (uint32_t t = blockIdx.x * SORT_THREADS + threadIdx.x; t < total; t += gridDim.x * SORT_THREADS) {
        uint32_t id = t >> 4, j = t & 15u;
        uint32_t o = j & 7u;
        bool minus = j >= 8u;
        int b[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[id], b);
        uint32_t res = NONE;
        if ((int)o < NN) {
            int sgn = minus ? -1 : 1;
            int nb[3] = {b[0] + sgn * (int)(o & 1u), b[1] + sgn * (int)((o >> 1) & 1u), b[2] + sgn * (int)((o >> 2) & 1u)};
            if (block_in_key_range<D>(nb)) res = hmap_find(d, pack_key<D>(nb), epoch);
        }
        (minus ? d.nbr_minus : d.nbr_plus)[id * 8u + o] = res;
        if (j == 0) {
            d.block_count[id] = 0;
            d.block_cdf_flag[id] = 0;
        }
    }
}

// sort.wgsl:89-99 update_block_particle_count, extended to per-cell counts.
// Scattered device-scope atomics run at the memory side on MI355X (~20 G/s when every lane
// hits its own line), so counting is done in LDS per (wave, block): lanes get their rank
// inside the wave's group from an LDS atomic, and one coalesced returning global atomic per
// (wave, block) reserves the group's range inside each cell. rank_in_cell depends on the
// arrival order of those atomics; k_canonical_order removes that dependence.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_count(Dev d, int side, uint32_t epoch) {
    constexpr int BS = Dim<D>::BSHIFT, BW = Dim<D>::BW;
    __shared__ uint32_t s_hist[SORT_THREADS / 64][NPB];
    const float *in = d.buf[side];
    uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x;
    bool valid = i < d.n;
    uint32_t key = NONE, local = 0;
    if (valid) {
        int c[D], b[3] = {0, 0, 0};
        load_cell<D>(in, d.npad, i, d.h, c);
        uint32_t shift = 0;
#pragma unroll
        for (int k = 0; k < D; k++) {
            b[k] = c[k] >> BS;
            local |= (uint32_t)(c[k] & (BW - 1)) << shift;  // grid.wgsl:346-348 node_id: x + BW*y (+ BW^2*z)
            shift += BS;
        }
        valid = block_in_key_range<D>(b);
        key = valid ? pack_key<D>(b) : NONE;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *hist = s_hist[wave];
    uint32_t cid = NONE, rank = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        uint32_t k0 = __shfl(key, leader);
        const bool mine = valid && key == k0;
        unsigned long long same = __ballot(mine);
        todo &= ~same;
        uint32_t found = 0;
        if (lane == leader) {
            found = hmap_find(d, k0, epoch);
            if (found != NONE) atomicAdd(&d.block_count[found], (uint32_t)__popcll(same));
        }
        found = __shfl(found, leader);
        if (found == NONE) continue;  // block missing (grid overflow): particle is dropped, error already flagged
        // Cross-lane traffic through LDS inside one wave: use (relaxed, wavefront-scope) atomic
        // accesses so the compiler may not forward this lane's own stores to its loads.
        __hip_atomic_store(&hist[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        uint32_t r = 0;
        if (mine) r = __hip_atomic_fetch_add(&hist[local], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        uint32_t cnt = __hip_atomic_load(&hist[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        uint32_t base = 0;
        if (cnt) base = atomicAdd(&d.cell_count[found * NPB + lane], cnt);  // one coalesced atomic per (wave, block)
        __hip_atomic_store(&hist[lane], base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (mine) {
            cid = found * NPB + local;
            rank = __hip_atomic_load(&hist[local], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + r;
        }
    }
    if (i < d.n) {
        d.cellid[i] = cid;
        d.rank[i] = rank;
    }
}

// Exclusive scan of block_count -> block_start (prefix_sum.wgsl + sort.wgsl:101-115).
// One workgroup; B is at most a few hundred thousand.
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 8;
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_blocks(Dev d) {
    __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
    __shared__ uint32_t carry_s;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < B; base += SCAN_THREADS * SCAN_ITEMS) {
        uint32_t v[SCAN_ITEMS];
        uint32_t sum = 0;
        uint32_t first = base + (uint32_t)tid * SCAN_ITEMS;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            v[k] = first + k < B ? d.block_count[first + k] : 0u;
            sum += v[k];
        }
        // inclusive wave scan of the per-thread sums
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_sums[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += wave_sums[w];
        uint32_t carry = carry_s;
        uint32_t run = carry + wave_off + inc - sum;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (first + k < B) d.block_start[first + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == SCAN_THREADS - 1) carry_s = run;
        __syncthreads();
    }
}

// Per-cell offsets inside each block: one wave per block (64 cells = 64 lanes).
__global__ __launch_bounds__(SORT_THREADS) void k_cell_offsets(Dev d) {
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * SORT_THREADS + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * SORT_THREADS) >> 6;
    for (uint32_t b = wave; b < B; b += nwaves) {
        uint32_t idx = b * NPB + lane;
        uint32_t cnt = d.cell_count[idx];
        uint32_t inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        uint32_t start = d.block_start[b] + inc - cnt;
        d.cell_start[idx] = start;
        d.cell_cursor[idx] = start + cnt;  // cell end
        d.cell_count[idx] = 0;  // ready for the next substep
    }
}

// sort.wgsl:117-127 finalize_particles_sort (the sorted-ids half).
__global__ __launch_bounds__(SORT_THREADS) void k_scatter(Dev d) {
    uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x;
    if (i >= d.n) return;
    uint32_t cid = d.cellid[i];
    if (cid == NONE) return;
    d.perm[d.cell_start[cid] + d.rank[i]] = i;
}

// Canonical order inside each cell: ascending persistent particle id.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_canonical_order(Dev d, int side) {
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const uint32_t total = B * NPB;
    const uint32_t *pid = reinterpret_cast<const uint32_t *>(d.buf[side] + (size_t)Pl<D>::NQ * 4 * d.npad);
    for (uint32_t c = blockIdx.x * SORT_THREADS + threadIdx.x; c < total; c += gridDim.x * SORT_THREADS) {
        uint32_t s = d.cell_start[c], e = d.cell_cursor[c];
        for (uint32_t a = s + 1; a < e; a++) {  // insertion sort, ~8 elements
            uint32_t pa = d.perm[a], ka = pid[pa];
            uint32_t j = a;
            while (j > s) {
                uint32_t pj = d.perm[j - 1];
                if (pid[pj] <= ka) break;
                d.perm[j] = pj;
                j--;
            }
            d.perm[j] = pa;
        }
    }
}

}  // namespace wgs
