// kernels_shard.h — x-slab domain decomposition across GPUs (new design: the reference is
// single-GPU). One process per GPU owns the particles whose associated
// block has bx in [shard_lo, shard_hi) — its CORE range. ONE neighbour exchange per substep:
//
//   A particle with associated cell c touches the nodes c .. c+2 only, so a rank's core particles reach the first
//   two node layers of block layer shard_hi and nothing below shard_lo. A particle that LEAVES the core range in the
//   fused G2P of substep n is not sent at once (that would be a second exchange, between the G2P and the next sort):
//   it stays with its old owner as a GUEST for the sort and the P2G of substep n + 1, its record (state after
//   substep n, which nothing changes before the next G2P) travels in the SAME message as the node sums of substep
//   n + 1, and the new owner runs the G2P + particle update of substep n + 1 on it (k_g2p_arrivals) and keeps it.
//   The old owner's fused G2P drops it. A guest sits at bx = shard_hi or shard_lo - 1 (a particle moves less than
//   a cell per substep: grid_update.wgsl:60-62, particle_update.wgsl:70-72), so the old owner's P2G also writes
//   block layer shard_hi completely and two node layers of shard_hi + 1 (and shard_lo - 1, shard_lo below).
//
//   Message to the UPPER neighbour: partial (momentum, mass) sums of block layer hi (all x-layer pairs) and of
//   hi + 1 (first pair) + the records of the guests at bx = hi.  To the LOWER neighbour: layers lo - 1 (all pairs)
//   and lo (first pair) + the guests at bx = lo - 1. A record = one x-layer PAIR of one block (key, pair index,
//   2 * BW^(D-1) float4); pairs only guests can have touched are sent when they are non-zero. Both sides ADD what
//   they receive (a + b == b + a bitwise: both hold identical totals of the nodes they share) and update redundantly.
//   Nothing is added by a launch of its own: the grid update looks its interface blocks up in the inbound message (a wave
//   scans the ~300 record headers of a message in a handful of coalesced loads) and adds the neighbour's pair to its own
//   gather. A received pair whose block is not active here (the first particles to enter an empty region) can only
//   matter to an arriving particle: its G2P reads such nodes straight from the message — nobody here contributes to
//   them, the neighbour's partial sum IS the total.
//
// Slabs with two neighbours must be at least 3 blocks wide (the layers the two messages touch must not overlap).
// Everything is stream-ordered: messages are fixed-capacity buffers whose header holds the record counts,
// particle counts live in device counters, so a substep needs no host synchronisation (overflow of a buffer
// sets ERRBIT_SHARD, seen at the next wgs_sync).
#pragma once
#include "kernels_cdf.h"

namespace wgs {

template <int D> struct HaloCfg {
    static constexpr int BW = Dim<D>::BW;
    static constexpr int NODES = D == 3 ? 2 * BW * BW : 2 * BW;  // one x-layer pair of a block
    static constexpr int NTAG = BW / 2;                          // pairs per block (3D: 2, 2D: 4)
    static constexpr uint32_t ALL = (1u << NTAG) - 1u;
    static constexpr int REC_F4 = NODES + 1;                     // [key, pair, -, -] + node partial sums
};

// local index of node q of x-layer pair `tag` of a block: lx = 2 tag + (q & 1), then y (, z)
template <int D> __device__ inline uint32_t halo_node(int tag, int q) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT;
    const int lx = 2 * tag + (q & 1), ly = (q >> 1) & (BW - 1), lz = D == 3 ? (q >> (1 + BS)) : 0;
    return (uint32_t)(lx + (ly << BS) + (D == 3 ? (lz << (2 * BS)) : 0));
}
// inverse: (pair, q) of a local node index
template <int D> __device__ inline void halo_slot(uint32_t ln, int &tag, int &q) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT;
    const int lx = ln & (BW - 1), ly = (ln >> BS) & (BW - 1), lz = D == 3 ? (int)(ln >> (2 * BS)) : 0;
    tag = lx >> 1;
    q = (lx & 1) + 2 * (ly + (D == 3 ? BW * lz : 0));
}

// What block layer bx is to this slab (bit t = x-layer pair t):
//   recv       — pairs a neighbour's particles can reach: the grid update adds what the neighbour sent to its own gather
//   send_lo/hi — pairs that travel to the lower / upper neighbour (k_pack_face gathers them from the slabs)
struct IfaceMasks {
    uint32_t recv, send_lo, send_hi;
};
template <int D> __device__ inline IfaceMasks iface_masks(const Dev &d, int bx) {
    constexpr uint32_t ALL = HaloCfg<D>::ALL;
    IfaceMasks m = {0u, 0u, 0u};
    if (d.shard_has_lo) {
        const int lo = d.shard_lo;
        if (bx == lo - 1) m.send_lo |= ALL;
        if (bx == lo)     { m.send_lo |= 1u; m.recv |= ALL; }
        if (bx == lo + 1) m.recv |= 1u;
    }
    if (d.shard_has_hi) {
        const int hi = d.shard_hi;
        if (bx == hi - 1) m.recv |= ALL;
        if (bx == hi)     { m.send_hi |= ALL; m.recv |= 1u; }
        if (bx == hi + 1) m.send_hi |= 1u;
    }
    return m;
}
// The block layers whose P2G slabs feed an outgoing message: what travels to the upper neighbour is gathered from the slabs of
// layers hi - 1 (its rim) and hi (the guests), what travels to the lower one from lo - 1 (guests) and lo.
__device__ inline bool shard_boundary_layer(const Dev &d, int bx) {
    return (d.shard_has_lo && (bx == d.shard_lo - 1 || bx == d.shard_lo)) || (d.shard_has_hi && (bx == d.shard_hi - 1 || bx == d.shard_hi));
}
// the inbound message that can hold pairs of layer bx (layers lo, lo + 1 come from below, hi - 1, hi from above: a slab
// with two neighbours is at least 3 blocks wide)
__device__ inline int iface_recv_face(const Dev &d, int bx) { return (d.shard_has_hi && bx >= d.shard_hi - 1) ? 1 : 0; }

// Full-state record of one particle: NQ quads + pid + cdf epoch.
template <int D> constexpr int particle_record_floats() { return Pl<D>::NQ * 4 + 2; }

// One message = header (4 words: -, particle records, flags, -) + halo_cap halo record slots (a hash table, below)
// + mig_cap particle records.
constexpr uint32_t MSG_FLAG_UNIFORM = 1u;   // particle records are in the uniform-material layout (layout.h)
template <int D> __host__ __device__ inline size_t msg_floats(uint32_t halo_cap, uint32_t mig_cap) {
    return 4 + (size_t)halo_cap * HaloCfg<D>::REC_F4 * 4 + (size_t)mig_cap * particle_record_floats<D>();
}
template <int D> __device__ inline const float4 *msg_halo(const float *msg) { return reinterpret_cast<const float4 *>(msg) + 1; }
template <int D> __device__ inline float4 *msg_halo(float *msg) { return reinterpret_cast<float4 *>(msg) + 1; }
template <int D> __device__ inline const float *msg_particles(const float *msg, uint32_t halo_cap) {
    return msg + 4 + (size_t)halo_cap * HaloCfg<D>::REC_F4 * 4;
}
template <int D> __device__ inline float *msg_particles(float *msg, uint32_t halo_cap) {
    return msg + 4 + (size_t)halo_cap * HaloCfg<D>::REC_F4 * 4;
}

template <int D> __device__ inline float4 gather_slabs(const Dev &d, uint32_t b, uint32_t ln);  // kernels_transfer.h

// After P2G: (a) one wave per block of the layers that travel (the sort listed them): their pairs are gathered from the
// slabs and appended to the outgoing messages; (b) the guests — the particles the last
// G2P launch found outside the core range (Dev::leavers) — are copied into the message of the face they crossed. They
// are NOT vacated here: this rank's fused G2P drops them (their block lies outside the core range).
// One body for the launch of its own (k_pack_face: worker = workgroup of one wave) and for the pack waves that ride in the
// P2G launch (INLAUNCH: worker = wave behind the P2G workgroups; the slabs are then waited for word by word and gathered
// with agent-scope loads, like gu_waves in kernels_transfer.h): workers [0, nblk_wk) walk the interface-block list, the
// others copy the guests.
template <int D> __device__ inline uint32_t rec_claim(const Dev &d, float *msg, uint32_t key, uint32_t tag, uint32_t epoch, int lane);   // (below)
template <int D, bool INLAUNCH> __device__ __forceinline__ void pack_face_body(const Dev &d, int side, uint32_t epoch, uint32_t wk, uint32_t nblk_wk, uint32_t nwk, int lane) {
    using H = HaloCfg<D>;
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, TW = Dim<D>::TW, TILE = Dim<D>::TILE, NN = Dim<D>::NNBR;
    constexpr int NQ = Pl<D>::NQ, RF = particle_record_floats<D>();
    if (wk == 0 && lane < 2 && d.msg.out[lane])   // layout of this rank's particle records (checked by the receiver, k_g2p_arrivals)
        reinterpret_cast<uint32_t *>(d.msg.out[lane])[2] = d.uniform ? MSG_FLAG_UNIFORM : 0u;
    if (wk < nblk_wk) {
        const uint32_t nl = min(d.counters[ctr_nhalo(epoch)], d.cap);
        for (uint32_t a = wk; a < nl; a += nblk_wk) {
            // the sort left everything needed in the list entry: block id, key and the slabs its nodes are gathered from
            // (its "-" neighbours that hold particles)
            uint32_t ew = NONE;
            if (lane < 10) ew = d.halo_list[(size_t)a * HALO_ENT + (uint32_t)lane];
            const uint32_t bkey = __shfl(ew, 1);
            const uint32_t src = __shfl(ew, 2 + (lane & 7));   // lanes 0 .. 7 (and their images): source slab o = lane & 7
            int bc[3] = {0, 0, 0};
            unpack_key<D>(bkey, bc);
            const IfaceMasks m = iface_masks<D>(d, bc[0]);
            int tag, q;
            halo_slot<D>((uint32_t)lane, tag, q);   // lane = node of the block
            // all slab loads of the node issued together, summed in the fixed order of the grid update (gather_slabs)
            const int l[3] = {lane & (BW - 1), (lane >> BS) & (BW - 1), D == 3 ? (lane >> (2 * BS)) : 0};
            float4 part[NN];
            if constexpr (INLAUNCH) {   // the source slabs are being written by P2G workgroups of this very launch
                if (lane < NN && src != NONE) {
                    uint32_t spins = 0u;
                    while (__hip_atomic_load(&d.slab_epoch[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins == (1u << 20)) {
                            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_HANDOVER);
                            break;
                        }
                    }
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int o = 0; o < NN; o++) {
                const int tt[3] = {l[0] + BW * (o & 1), l[1] + BW * ((o >> 1) & 1), l[2] + BW * ((o >> 2) & 1)};
                const bool in_tile = tt[0] < TW && tt[1] < TW && (D == 2 || tt[2] < TW);
                const bool wanted = in_tile && (((m.send_lo | m.send_hi) >> tag) & 1u);
                if constexpr (INLAUNCH) {
                    const uint32_t so = (uint32_t)__builtin_amdgcn_readlane((int)src, o);   // (wave-uniform: the slab's descriptor lives in scalar registers)
                    part[o] = ld_agent(slab_rsrc(&d.slab[(size_t)(so != NONE ? so : 0u) * TILE], so != NONE ? TILE * 16u : 0u),
                                       wanted ? slab_pos<D>(o, l) * 16u : 0x7ffffff0u);   // (no source / not wanted: zeros, no access)
                } else {
                    const uint32_t so = __shfl(src, o);
                    part[o] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (wanted && so != NONE) part[o] = d.slab[(size_t)so * TILE + slab_pos<D>(o, l)];
                }
            }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < NN; o++) {  // (a slab that is skipped adds nothing in gather_slabs either: + 0 changes no bit of a sum that started at + 0)
                v.x += part[o].x; v.y += part[o].y; v.z += part[o].z; v.w += part[o].w;
            }
            const bool nz = v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const uint32_t send = f == 0 ? m.send_lo : m.send_hi;
                float *msg = d.msg.out[f];
                if (send == 0u || !msg) continue;  // wave-uniform
                for (int t = 0; t < H::NTAG; t++) {
                    if (!((send >> t) & 1u)) continue;
                    // the pair every core particle near the face writes travels always; pairs only guests reach, when non-zero
                    const bool regular = t == 0 && bc[0] == (f == 0 ? d.shard_lo : d.shard_hi);
                    const bool any = __ballot(tag == t && nz) != 0ull;
                    if (!regular && !any) continue;
                    const uint32_t slot = rec_claim<D>(d, msg, bkey, (uint32_t)t, epoch, lane);   // (no record counter: a thousand adds on one address serialise)
                    if (slot == NONE) continue;
                    float4 *rec = msg_halo<D>(msg) + (size_t)slot * H::REC_F4;
                    if (tag == t) rec[1 + q] = v;
                }
            }
        }
        return;
    }
    // (b) guests
    const float *buf = d.buf[side];
    const uint32_t npad = d.npad;
    const uint32_t n = num_slots(d);
    const uint32_t nl = min(d.counters[CTR_NLEAVE], d.leavers_cap);
    const uint32_t nl64 = (nl + 63u) & ~63u;   // (whole waves: the slots of a wave's guests are handed out with one atomic per face)
    for (uint32_t t = (wk - nblk_wk) * 64u + (uint32_t)lane; t < nl64; t += (nwk - nblk_wk) * 64u) {
        int face = -1;
        uint32_t i = 0, pid = PID_DEAD;
        if (t < nl) {
            i = d.leavers[t];
            if (i < n) pid = ldpid<D>(buf, npad, i);
            if (pid != PID_DEAD) {
                const float4 xm = ldq(buf, npad, Pl<D>::XM, i);
                const int bx = assoc_cell(xm.x, d.h, d.inv_h, d.h_pow2 != 0u) >> BS;
                face = bx < d.shard_lo ? 0 : (bx >= d.shard_hi ? 1 : -1);
                if constexpr (INLAUNCH) {
                    // A guest of a block near a collider gets its cdf quads rewritten by the CPIC prologue of this very launch
                    // (p2g_body.inc): its record is copied once its block's slab word says the block is done.
                    if (face >= 0 && d.n_colliders != 0u) {
                        int gb[3] = {bx, assoc_cell(xm.y, d.h, d.inv_h, d.h_pow2 != 0u) >> BS, 0};
                        if constexpr (D == 3) gb[2] = assoc_cell(xm.z, d.h, d.inv_h, d.h_pow2 != 0u) >> BS;
                        const uint32_t gid = hmap_find(d, pack_key<D>(gb), epoch);
                        if (gid != NONE) {
                            uint32_t spins = 0u;
                            while (__hip_atomic_load(&d.slab_epoch[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                                __builtin_amdgcn_s_sleep(2);
                                if (++spins == (1u << 20)) {
                                    atomicOr(&d.counters[CTR_ERRORS], ERRBIT_HANDOVER);
                                    break;
                                }
                            }
                        }
                    }
                }
            }
        }
        uint32_t slot = NONE;
#pragma unroll
        for (int f = 0; f < 2; f++) {
            const unsigned long long mine = __ballot(face == f);
            if (mine == 0ull) continue;
            float *msgf = d.msg.out[f];
            uint32_t base = 0;
            if (msgf && lane == __ffsll((long long)mine) - 1) base = atomicAdd(reinterpret_cast<uint32_t *>(msgf) + 1, (uint32_t)__popcll(mine));
            base = __shfl(base, __ffsll((long long)mine) - 1);
            if (face == f) slot = msgf ? base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull)) : NONE;
        }
        if (face < 0) continue;
        float *msg = d.msg.out[face];
        if (!msg) {  // no neighbour on that side: the caller's decomposition does not cover the scene
            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);
            continue;
        }
        if (slot >= d.msg.mig_cap) {
            atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);  // the particle is lost to the neighbour: wrong physics, reported
            continue;
        }
        float *rec = msg_particles<D>(msg, d.msg.halo_cap) + (size_t)slot * RF;
#pragma unroll
        for (int qd = 0; qd < NQ; qd++) {
            // (INLAUNCH: the cdf quads may have been written through by a P2G workgroup of another XCD in this launch)
            const float4 v = INLAUNCH ? ld_agent(particle_rsrc<D>(const_cast<float *>(buf), npad), quad_off(npad, qd, i)) : ldq(buf, npad, qd, i);
            rec[qd * 4 + 0] = v.x; rec[qd * 4 + 1] = v.y; rec[qd * 4 + 2] = v.z; rec[qd * 4 + 3] = v.w;
        }
        rec[NQ * 4] = __uint_as_float(pid);
        rec[NQ * 4 + 1] = __uint_as_float(INLAUNCH ? __hip_atomic_load(stamp_ptr<D>(const_cast<float *>(buf), npad, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                   : ldstamp<D>(buf, npad, i));
    }
}
template <int D> __global__ __launch_bounds__(64) void k_pack_face(Dev d, int side, uint32_t epoch, uint32_t nblk_wgs) {
    pack_face_body<D, false>(d, side, epoch, blockIdx.x, nblk_wgs, gridDim.x, (int)threadIdx.x);
}

// The halo area of a message is an OPEN-ADDRESSING TABLE of records keyed by (block key, pair): the sender claims the slot
// hash(key, pair) (linear probing, one 64-bit compare-and-swap on the record's header), the receiver finds a block's pair
// in one or two probes instead of scanning a thousand headers. A header = (key, substep << 4 | pair): slots of earlier
// substeps read as free (substep numbers only grow, and both ranks count the same substeps), so nothing is ever cleared.
__device__ inline uint32_t rec_hash(uint32_t key, uint32_t tag, uint32_t cap) { return hash_key(key ^ (tag * 0x9e3779b9u)) % cap; }
__device__ inline unsigned long long rec_header(uint32_t key, uint32_t tag, uint32_t epoch) {
    return (unsigned long long)key | ((unsigned long long)((epoch << 4) | tag) << 32);
}
// sender, a whole wave (same arguments in all lanes): slot claimed for (key, pair) in `msg`, or NONE (table full:
// reported). The 64 lanes look at 64 consecutive slots of the probe sequence at once — a few records of a face sit at the
// end of probe chains 20 slots long, and walked one slot per round trip they alone made this launch twice as long —
// and the first free one is claimed with one compare-and-swap.
template <int D> __device__ inline uint32_t rec_claim(const Dev &d, float *msg, uint32_t key, uint32_t tag, uint32_t epoch, int lane) {
    const uint32_t cap = d.msg.halo_cap;
    const unsigned long long want = rec_header(key, tag, epoch);
    const uint32_t h = rec_hash(key, tag, cap);
    auto header = [&](uint32_t slot) { return reinterpret_cast<unsigned long long *>(msg_halo<D>(msg) + (size_t)slot * HaloCfg<D>::REC_F4); };
    {   // steady state: this very record held its first slot one substep ago — one atomic, no look before
        int ok = 0;
        if (lane == 0) ok = atomicCAS(header(h), rec_header(key, tag, epoch - 1u), want) == rec_header(key, tag, epoch - 1u) ? 1 : 0;
        if (__shfl(ok, 0)) return h;
    }
    for (uint32_t base = 0; base < cap; base += 64u) {
        const uint32_t off = base + (uint32_t)lane;
        const uint32_t slot = (h + off) % cap;
        unsigned long long cur = want;  // (lanes past the end of the table: "taken")
        if (off < cap) cur = __hip_atomic_load(header(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long free = __ballot(off < cap && (uint32_t)(cur >> 36) != (epoch & 0x0fffffffu));
        while (free != 0ull) {  // first free slot in probe order; lost to another wave: the next one
            const int first = __ffsll((long long)free) - 1;
            int ok = 0;
            if (lane == first) ok = atomicCAS(header(slot), cur, want) == cur ? 1 : 0;
            if (__shfl(ok, first)) return __shfl(slot, first);
            free &= free - 1ull;
        }
    }
    if (lane == 0) atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);
    return NONE;
}
// receiver, a whole wave (same arguments in all lanes): the slot of (key, pair) in the inbound message of `face`, or NONE
template <int D> __device__ inline uint32_t rec_find_wave(const Dev &d, int face, uint32_t key, uint32_t tag, uint32_t epoch, int lane) {
    const float *msg = d.msg.in[face];
    if (!msg) return NONE;
    const uint32_t cap = d.msg.halo_cap;
    const unsigned long long want = rec_header(key, tag, epoch);
    const uint32_t h = rec_hash(key, tag, cap);
    for (uint32_t base = 0; base < cap; base += 64u) {
        const uint32_t off = base + (uint32_t)lane;
        const uint32_t slot = (h + off) % cap;
        unsigned long long cur = 0ull;
        if (off < cap) cur = *reinterpret_cast<const unsigned long long *>(msg_halo<D>(msg) + (size_t)slot * HaloCfg<D>::REC_F4);
        const unsigned long long hit = __ballot(off < cap && cur == want);
        const unsigned long long free = __ballot(off < cap && (uint32_t)(cur >> 36) != (epoch & 0x0fffffffu));
        // a free slot ends the probe sequence: a record behind it is not this substep's
        if (hit != 0ull && (free == 0ull || __ffsll((long long)hit) < __ffsll((long long)free))) return __shfl(slot, __ffsll((long long)hit) - 1);
        if (free != 0ull) return NONE;
    }
    return NONE;
}
// receiver, one lane: the record of (key, pair) in the inbound message of `face`, or null
template <int D> __device__ inline const float4 *rec_find(const Dev &d, int face, uint32_t key, uint32_t tag, uint32_t epoch) {
    const float *msg = d.msg.in[face];
    if (!msg) return nullptr;
    const uint32_t cap = d.msg.halo_cap;
    const unsigned long long want = rec_header(key, tag, epoch);
    uint32_t slot = rec_hash(key, tag, cap);
    for (uint32_t probe = 0; probe < cap; probe++) {
        const float4 *rec = msg_halo<D>(msg) + (size_t)slot * HaloCfg<D>::REC_F4;
        const unsigned long long cur = *reinterpret_cast<const unsigned long long *>(rec);
        if (cur == want) return rec;
        if ((uint32_t)(cur >> 36) != (epoch & 0x0fffffffu)) return nullptr;  // a free slot ends the probe sequence
        slot = slot + 1u == cap ? 0u : slot + 1u;
    }
    return nullptr;
}

// Read-back of a slab: full records of every live particle (guests included: they are this rank's until sent).
template <int D> __global__ __launch_bounds__(256) void k_export_records(Dev d, int side, float *buf_out, uint32_t cap) {
    constexpr int NQ = Pl<D>::NQ, RF = particle_record_floats<D>();
    const float *buf = d.buf[side];
    const uint32_t npad = d.npad;
    const uint32_t n = num_slots(d);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t pid = ldpid<D>(buf, npad, i);
        if (pid == PID_DEAD) continue;
        const uint32_t slot = atomicAdd(reinterpret_cast<uint32_t *>(buf_out), 1u);
        if (slot >= cap) continue;  // (the count says so)
        float *rec = buf_out + 4 + (size_t)slot * RF;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float4 v = ldq(buf, npad, q, i);
            rec[q * 4 + 0] = v.x; rec[q * 4 + 1] = v.y; rec[q * 4 + 2] = v.z; rec[q * 4 + 3] = v.w;
        }
        rec[NQ * 4] = __uint_as_float(pid);
        rec[NQ * 4 + 1] = __uint_as_float(ldstamp<D>(buf, npad, i));
    }
}

__global__ void k_clear_headers(uint32_t *a, uint32_t *b) {
    if (threadIdx.x < 4) {
        if (a) a[threadIdx.x] = 0;
        if (b) b[threadIdx.x] = 0;
    }
}

// Sharded data stepped WITHOUT its neighbours (wgs_step on a slab, or a slab that is not attached): the buffer the fused
// G2P wrote holds exactly the valid particles, in sorted order.
__global__ void k_shard_compacted(Dev d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t nv = ctr_next(d, CTR_NV);   // (launched at the head of the NEXT substep: the last one read the other set)
        ctr_cur(d, CTR_NV) = nv;
        ctr_cur(d, CTR_N) = nv;
        ctr_cur(d, CTR_NPREV) = nv;
    }
}

}  // namespace wgs
