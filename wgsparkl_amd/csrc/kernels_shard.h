// kernels_shard.h — x-slab domain decomposition across GPUs (new design: the reference is
// single-GPU). One process per GPU owns the particles whose associated
// block has bx in [shard_lo, shard_hi). Because the stencil of a particle only reaches
// nodes c .. c+2 of its associated cell c, a rank's particles write into, and read from, the
// first two node layers (lx in {0,1}) of the block layer bx = shard_hi owned by the next
// rank, and into nothing on the lower side. Per substep two small neighbour exchanges:
//   1. after the P2G gather: partial (momentum, mass) sums of the interface node layers,
//      both ways; each side adds what it received (a + b == b + a bitwise), so both ranks
//      hold identical totals and run the grid update redundantly on those nodes;
//   2. after the particle update: particles whose associated block left the rank's range
//      (at most one block per substep because of the h/dt velocity clamps) move, full state.
// Everything is stream-ordered: messages are fixed-capacity buffers whose first 16 bytes hold
// the record count, particle counts live in device counters, so a substep needs no host
// synchronisation (overflow of a buffer sets ERRBIT_SHARD, seen at the next wgs_sync).
#pragma once
#include "device_math.h"

namespace wgs {

constexpr uint32_t PID_DEAD = 0xffffffffu;  // slot vacated by a migrated particle

template <int D> struct HaloCfg {
    static constexpr int BW = Dim<D>::BW;
    static constexpr int NODES = D == 3 ? 2 * BW * BW : 2 * BW;  // two x-layers of a block
    static constexpr int REC_F4 = NODES + 1;                     // [key,0,0,0] + node partial sums
};

// local index of interface node q of a block: lx = q & 1, then y (, z)
template <int D> __device__ inline uint32_t halo_node(int q) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT;
    const int lx = q & 1, ly = (q >> 1) & (BW - 1), lz = D == 3 ? (q >> (1 + BS)) : 0;
    return (uint32_t)(lx + (ly << BS) + (D == 3 ? (lz << (2 * BS)) : 0));
}

__global__ void k_clear_headers(uint32_t *a, uint32_t *b) {
    if (threadIdx.x == 0) {
        if (a) a[0] = 0;
        if (b) b[0] = 0;
    }
}

// Pack the partial sums of the interface layer `layer_bx` (active blocks only).
// buf = [count, -, -, -] + cap records of REC_F4 float4.
template <int D> __global__ __launch_bounds__(64) void k_pack_halo(Dev d, int layer_bx, float4 *buf, uint32_t cap) {
    using H = HaloCfg<D>;
    uint32_t *count = reinterpret_cast<uint32_t *>(buf);
    float4 *out = buf + 1;
    const uint32_t B = min(d.counters[CTR_NBLOCKS], d.cap);
    const int lane = threadIdx.x;
    for (uint32_t a = blockIdx.x; a < B; a += gridDim.x) {
        const uint32_t b = d.active[a];
        int bc[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[b], bc);
        if (bc[0] != layer_bx) continue;  // wave-uniform
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(count, 1u);
        slot = __shfl(slot, 0);
        if (slot >= cap) {
            if (lane == 0) atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);
            continue;
        }
        float4 *rec = out + (size_t)slot * H::REC_F4;
        if (lane == 0) rec[0] = make_float4(__uint_as_float(d.block_key[b]), 0.f, 0.f, 0.f);
        if (lane < H::NODES) rec[1 + lane] = d.nodes[(size_t)b * NPB + halo_node<D>(lane)];
    }
}

// Both interface layers in one launch (a launch costs ~4.5 us whatever it does): layer_lo -> buf_lo, layer_hi -> buf_hi;
// a null buffer = no neighbour on that side.
// `gather`: the node sums are not in nodes[] yet (no separate gather pass ran): compute them here from the slabs and
// leave them in nodes[] for k_add_halo / the PHASE 3 grid update.
template <int D> __device__ inline float4 gather_slabs(const Dev &d, uint32_t b, uint32_t ln);  // kernels_transfer.h
template <int D> __global__ __launch_bounds__(64) void k_pack_halos(Dev d, float4 *buf_lo, float4 *buf_hi, uint32_t cap, int gather) {
    using H = HaloCfg<D>;
    const int lane = threadIdx.x;
    // the active blocks of the two layers were listed by launch 2 of the sort (one wave per entry here: even workgroups
    // walk layer_lo's list, odd ones layer_hi's) — scanning the whole active list for them took 11 us at 4000 blocks
    const uint32_t sd = blockIdx.x & 1u;
    const uint32_t nl = min(d.counters[CTR_NHALO + 32u * sd], d.cap);
    for (uint32_t a = blockIdx.x >> 1; a < nl; a += gridDim.x >> 1) {
        const uint32_t b = d.halo_list[(size_t)sd * d.cap + a];
        float4 *buf = sd == 0u ? buf_lo : buf_hi;  // wave-uniform; null: no neighbour on that side (the sums are still gathered)
        if (gather && lane < H::NODES) {
            const uint32_t ln = halo_node<D>(lane);
            d.nodes[(size_t)b * NPB + ln] = gather_slabs<D>(d, b, ln);  // (read back below by the same lane)
        }
        if (!buf) continue;
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(reinterpret_cast<uint32_t *>(buf), 1u);
        slot = __shfl(slot, 0);
        if (slot >= cap) {
            if (lane == 0) atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);
            continue;
        }
        float4 *rec = buf + 1 + (size_t)slot * H::REC_F4;
        if (lane == 0) rec[0] = make_float4(__uint_as_float(d.block_key[b]), 0.f, 0.f, 0.f);
        if (lane < H::NODES) rec[1 + lane] = d.nodes[(size_t)b * NPB + halo_node<D>(lane)];
    }
}

// Add a neighbour's partial sums to the blocks this rank has active.
// blockIdx.y selects the message (buf, or buf2 of the other neighbour when given): the two touch different block layers.
template <int D> __global__ __launch_bounds__(64) void k_add_halo(Dev d, const float4 *buf, const float4 *buf2, uint32_t cap, uint32_t epoch) {
    using H = HaloCfg<D>;
    if (blockIdx.y == 1) buf = buf2;
    if (!buf) return;
    const uint32_t n_rec = min(reinterpret_cast<const uint32_t *>(buf)[0], cap);
    const float4 *in = buf + 1;
    const int lane = threadIdx.x;
    for (uint32_t r = blockIdx.x; r < n_rec; r += gridDim.x) {
        const float4 *rec = in + (size_t)r * H::REC_F4;
        const uint32_t key = __float_as_uint(rec[0].x);
        const uint32_t b = hmap_find(d, key, epoch);
        if (b == NONE) continue;  // not active here: nobody on this rank reads those nodes
        if (lane < H::NODES) {
            const size_t node = (size_t)b * NPB + halo_node<D>(lane);
            const float4 a = d.nodes[node], p = rec[1 + lane];
            d.nodes[node] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        }
    }
}

// Full-state record of one particle: NQ quads + pid. A particle buffer = 4 header floats
// ([count, -, -, -]) + cap records.
template <int D> constexpr int particle_record_floats() { return Pl<D>::NQ * 4 + 2; }  // quads, pid, cdf epoch

// Particles whose associated block left [shard_lo, shard_hi): copy them to the outbox of the face
// they crossed and vacate their slot. mode 1 = export every valid particle instead (read-back).
template <int D> __global__ __launch_bounds__(256) void k_pack_migrants(Dev d, int side, int mode, float *buf_lo, float *buf_hi, uint32_t cap) {
    constexpr int BS = Dim<D>::BSHIFT, NQ = Pl<D>::NQ, RF = particle_record_floats<D>();
    float *buf = d.buf[side];
    const uint32_t npad = d.npad;
    // right after a substep (mode 0) the buffer holds the valid particles only; a read-back (mode 1) may come
    // after a migration round, when vacated slots and appended particles coexist
    const uint32_t n = mode == 1 ? num_slots(d) : num_valid(d);
    if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        // after the fused G2P kernel this buffer holds exactly the valid particles, in sorted order (nothing in this
        // launch reads CTR_N / CTR_NPREV)
        d.counters[CTR_N] = n;
        d.counters[CTR_NPREV] = n;  // residents of the next substep (arrivals are appended behind)
    }
    // mode 0: the fused G2P launch listed the slots of the particles that left the slab (Dev::leavers); mode 1 (read-back):
    // every slot
    const uint32_t nl = mode == 0 ? min(d.counters[CTR_NLEAVE], d.leavers_cap) : n;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < nl; t += gridDim.x * 256) {
        const uint32_t i = mode == 0 ? d.leavers[t] : t;
        if (i >= n) continue;
        const uint32_t pid = ldpid<D>(buf, npad, i);
        if (pid == PID_DEAD) continue;
        int face = -1;
        if (mode == 1) {
            face = 0;
        } else {
            const float4 xm = ldq(buf, npad, Pl<D>::XM, i);
            const int bx = assoc_cell(xm.x, d.h, d.inv_h, d.h_pow2 != 0u) >> BS;
            if (bx < d.shard_lo) face = 0;
            else if (bx >= d.shard_hi) face = 1;
        }
        if (face < 0) continue;
        float *ob = face ? buf_hi : buf_lo;
        const uint32_t slot = atomicAdd(reinterpret_cast<uint32_t *>(ob), 1u);  // a handful of particles per substep
        if (slot < cap) {
            float *rec = ob + 4 + (size_t)slot * RF;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const float4 v = ldq(buf, npad, q, i);
                rec[q * 4 + 0] = v.x; rec[q * 4 + 1] = v.y; rec[q * 4 + 2] = v.z; rec[q * 4 + 3] = v.w;
            }
            rec[NQ * 4] = __uint_as_float(pid);
            rec[NQ * 4 + 1] = __uint_as_float(ldstamp<D>(buf, npad, i));
            if (mode == 0) stpid<D>(buf, npad, i, PID_DEAD);
        } else {
            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);  // the particle stays here: wrong physics, reported
        }
    }
}

// Append the particles received from both neighbours after the current ones.
template <int D> __global__ __launch_bounds__(256) void k_append_migrants(Dev d, int side, const float *in_lo, const float *in_hi, const float *out_lo,
                                                                          const float *out_hi, uint32_t cap) {
    constexpr int NQ = Pl<D>::NQ, RF = particle_record_floats<D>();
    float *buf = d.buf[side];
    const uint32_t n_lo = in_lo ? min(reinterpret_cast<const uint32_t *>(in_lo)[0], cap) : 0u;
    const uint32_t n_hi = in_hi ? min(reinterpret_cast<const uint32_t *>(in_hi)[0], cap) : 0u;
    const uint32_t first = d.counters[CTR_NPREV];  // the valid particles of the last substep occupy [0, NPREV)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // bookkeeping of the migration round (nothing in this launch reads these two): slots = residents + arrivals,
        // valid = residents - departures + arrivals
        auto cnt = [&](const float *b) { return b ? min(reinterpret_cast<const uint32_t *>(b)[0], cap) : 0u; };
        const uint32_t arrivals = min(n_lo + n_hi, d.n - first);
        d.counters[CTR_N] = first + arrivals;
        d.counters[CTR_NV] = first - cnt(out_lo) - cnt(out_hi) + arrivals;
    }
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n_lo + n_hi; r += gridDim.x * 256) {
        const float *rec = r < n_lo ? in_lo + 4 + (size_t)r * RF : in_hi + 4 + (size_t)(r - n_lo) * RF;
        const uint32_t i = first + r;
        if (i >= d.n) {  // d.n = allocated capacity in sharded mode
            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);
            continue;
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) stq(buf, d.npad, q, i, make_float4(rec[q * 4], rec[q * 4 + 1], rec[q * 4 + 2], rec[q * 4 + 3]));
        stpid<D>(buf, d.npad, i, __float_as_uint(rec[NQ * 4]));
        ststamp<D>(buf, d.npad, i, __float_as_uint(rec[NQ * 4 + 1]));
    }
}

// After the fused G2P kernel the other buffer holds exactly the valid particles, in sorted order.
__global__ void k_shard_compacted(Dev d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        d.counters[CTR_N] = d.counters[CTR_NV];
        d.counters[CTR_NPREV] = d.counters[CTR_NV];  // residents of the next substep (arrivals are appended behind)
    }
}

}  // namespace wgs
