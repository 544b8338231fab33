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
#pragma unroll
    for (int k = 0; k < D; k++) cell[k] = assoc_cell(in[(size_t)(Pl<D>::POS + k) * npad + i], h);
}

// sort.wgsl:26-36 touch_particle_blocks + grid.wgsl:323-334 mark_block_as_active.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_touch_blocks(Dev d, int side) {
    constexpr int BS = Dim<D>::BSHIFT;
    constexpr int NN = Dim<D>::NNBR;
    const float *in = d.buf[side];
    uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x;
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
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    while (todo) {  // wave-uniform: one iteration per distinct block in the wave
        int leader = __ffsll((long long)todo) - 1;
        uint32_t k0 = __shfl(key, leader);
        int lb[3];
        lb[0] = __shfl(b[0], leader);
        lb[1] = __shfl(b[1], leader);
        lb[2] = __shfl(b[2], leader);
        todo &= ~__ballot(valid && key == k0);
        if (lane < NN) {  // grid.wgsl:300-320: the 2^D blocks {b, b+1} per axis, one per lane
            int nb[3];
            nb[0] = lb[0] + (lane & 1);
            nb[1] = lb[1] + ((lane >> 1) & 1);
            nb[2] = lb[2] + ((lane >> 2) & 1);
            activate_block(d, pack_key<D>(nb));
        }
    }
}

// Neighbour links of every active block (replaces the per-thread hash lookups of
// p2g.wgsl:238-275, g2p.wgsl:72-132) and reset of the per-block particle counter.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_block_links(Dev d) {
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
            if (block_in_key_range<D>(nb)) res = hmap_find(d, pack_key<D>(nb));
        }
        (minus ? d.nbr_minus : d.nbr_plus)[id * 8u + o] = res;
        if (j == 0) {
            d.block_count[id] = 0;
            d.block_cdf_flag[id] = 0;
        }
    }
}

// sort.wgsl:89-99 update_block_particle_count, extended to per-cell counts.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_count(Dev d, int side) {
    constexpr int BS = Dim<D>::BSHIFT, BW = Dim<D>::BW;
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
    const int lane = threadIdx.x & 63;
    uint32_t id = NONE;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        uint32_t k0 = __shfl(key, leader);
        unsigned long long same = __ballot(valid && key == k0);
        todo &= ~same;
        uint32_t found = 0;
        if (lane == leader) {
            found = hmap_find(d, k0);
            if (found != NONE) atomicAdd(&d.block_count[found], (uint32_t)__popcll(same));
        }
        found = __shfl(found, leader);
        if (valid && key == k0) id = found;
    }
    if (i < d.n) {
        uint32_t cid = id == NONE ? NONE : id * NPB + local;
        d.cellid[i] = cid;
        if (cid != NONE) atomicAdd(&d.cell_count[cid], 1u);
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
        d.cell_cursor[idx] = start;
        d.cell_count[idx] = 0;  // ready for the next substep
    }
}

// sort.wgsl:117-127 finalize_particles_sort (the sorted-ids half).
__global__ __launch_bounds__(SORT_THREADS) void k_scatter(Dev d) {
    uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x;
    if (i >= d.n) return;
    uint32_t cid = d.cellid[i];
    if (cid == NONE) return;
    uint32_t r = atomicAdd(&d.cell_cursor[cid], 1u);
    d.perm[r] = i;
}

// Canonical order inside each cell: ascending persistent particle id.
template <int D> __global__ __launch_bounds__(SORT_THREADS) void k_canonical_order(Dev d, int side) {
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const uint32_t total = B * NPB;
    const uint32_t *pid = reinterpret_cast<const uint32_t *>(d.buf[side] + (size_t)Pl<D>::PID * d.npad);
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
