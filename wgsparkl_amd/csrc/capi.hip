// capi.hip — host side of the C ABI declared in include/wgsparkl_hip.h.
//
// Host-language note: the reference's host code is Rust (src/pipeline.rs); this
// image has no Rust toolchain, so the host side above the C ABI is C++ here and
// the Rust shim a maintainer would add is shown in INTEGRATION.md / rust/.
//
// One wgs_data owns one HIP stream and every device buffer of a simulation
// (MpmData owns every wgpu buffer, src/pipeline.rs:84-95). wgs_step only
// enqueues; nothing on the step path synchronises with the host.
#include "../../include/wgsparkl_hip.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels_bodies.h"
#include "kernels_cdf.h"
#include "kernels_rigid.h"
#include "kernels_shard.h"
#include "kernels_sort.h"
#include "kernels_transfer.h"
#include "kernels_arrivals.h"

using namespace wgs;

namespace {

constexpr int D = WGS_DIM;
constexpr int DD = D * D;
using P = Pl<D>;

#define WGS_STR2(x) #x
#define WGS_STR(x) WGS_STR2(x)

thread_local std::string g_last_error;

wgs_status fail(wgs_status code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(WGS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                \
    } while (0)

struct Events {
    static constexpr int MAX_SUBSTEPS = 64;
    static constexpr int MARKS = 11;  // + 2 calibration marks (9, 10) recorded back to back: the cost of a mark itself
    static constexpr int PASS_MARKS = 9;  // boundaries: start, sort, cdf_nodes, cdf_particles, p2g, grid, g2p, g2p near colliders, bodies(end)
    hipEvent_t ev[MAX_SUBSTEPS][MARKS];
    int used = 0;
    bool created = false;
};

}  // namespace

struct wgs_pipeline {
    int device = 0;
    int num_cus = 256;
    hipDeviceProp_t props;
};

// Multi-GPU (capi_sharded.inc): one RCCL communicator of the x-slab chain, and the message buffers of one slab.
struct wgs_comm {
    void *comm = nullptr;
    int rank = 0, world = 1;
    int lower = -1, upper = -1;   // peer ranks, -1 = none
    int device = 0;
};
struct ShardLink {                // device memory owned by the wgs_data: one message per face and direction (kernels_shard.h)
    bool attached = false;
    wgs_comm *comm = nullptr;     // null: lockstep transport (device-to-device copies inside one process)
    bool has_lower = false, has_upper = false;
    uint32_t halo_cap = 0, mig_cap = 0;
    size_t msg_floats = 0;
    float *msg_out[2] = {nullptr, nullptr}, *msg_in[2] = {nullptr, nullptr};   // [lower, upper]
};

struct wgs_data {
    wgs_pipeline *pipeline = nullptr;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    hipStream_t stream2 = nullptr;   // wgs_sharded_step: the boundary layers' P2G and the exchange run here, beside the interior's P2G (capi_sharded.inc)
    hipEvent_t ev_sorted = nullptr, ev_exchanged = nullptr;
    Dev dev{};
    int side = 0;
    bool plastic = false;
    bool cpic = false;
    bool prev_sorted = false;   // the current buffer is the sorted output of the previous substep (perm_cell, links valid)
    bool needs_compact = false; // sharded: the last substep ran without its neighbours (wgs_step): the counters of its buffer are still to be set
    bool in_sharded_step = false;  // the substep being enqueued belongs to wgs_sharded_step[_lockstep]: guests are dropped, arrivals advanced
    uint64_t substeps = 0;
    uint64_t device_bytes = 0;
    uint32_t sticky_errors = 0;
    uint32_t last_nblocks = 0;
    uint32_t nv_hint = 0;              // sharded data: particles this slab holds as the host last saw them (the launch bound is the capacity); picks the G2P chunk count per wave
    uint32_t seen_nblocks = 0;         // active blocks as last seen by the host, wgs_sync or the pinned watch (0: not yet): sizes the P2G grid
    uint32_t last_ncpic = UINT32_MAX;  // near-collider list length at the last wgs_sync (picks the P2G launch shape and G2P's register budget)
    uint32_t last_nvisit = UINT32_MAX; // visit-list length at the last wgs_sync (sizes the list half of k_g2p_pair)
    uint32_t last_movers = 0;          // CTR_MOVERS at the last wgs_sync (cumulative, modulo 2^32)
    uint32_t last_nphys = 0, last_nfree = 0, last_ntomb = 0;   // id high-water mark, free list, table marks at the last wgs_sync (wgs_stats)
    uint64_t movers_total = 0;         // the same, accumulated in 64 bits over the host's looks
    uint64_t table_rebuilds = 0;       // substeps that rebuilt the table of block ids (wgs_stats)
    bool prebinned = false;            // the last fused G2P binned its output for the coming substep (Dev::bin_next): no k_rebin launch then
    uint32_t capacity = 0;      // particle slots allocated
    uint32_t *shard_counts = nullptr;  // device scratch for pack kernels
    std::vector<void *> allocs;
    std::vector<size_t> alloc_bytes;  // parallel to allocs
    // by-pid static tables (never reordered)
    float *static_radius = nullptr;
    float *static_dp = nullptr;     // n*6
    float *static_phase = nullptr;  // n*2
    uint32_t *static_flags = nullptr;  // bit0 has_plasticity, bit1 has_phase
    SimParamsDev *sp = nullptr;
    ColliderDev *colliders = nullptr;
    std::vector<ColliderDev> host_colliders;  // what the host last wrote (poses / velocities move on the device)
    std::vector<BodyDev> host_bodies;
    bool bodies_move = false;   // some body has a velocity or a mass: integrate_bodies runs every substep
    uint32_t moving_mask = 0;   // ... which ones (bit per collider; sticky like bodies_move): the blocks out of their reach keep their node cdfs
    bool two_way = false;       // P2G accumulates the bodies' impulses: whenever a body can move (a kinematic body uses
                                // them too: the velocity caps of rigid_impulses.wgsl:112-125 apply once it is pushed)
    SimParamsDev host_sp{};
    Events events;
    float timings[WGS_NUM_PASSES] = {0};
    float mark_overhead_ms = 0.f;   // average distance of two adjacent timing marks in the last timestamped step
    bool timings_pending = false;
    uint32_t *watch = nullptr;          // pinned host copy of the device counters as of the end of the last wgs_step call
    hipEvent_t watch_event = nullptr;
    bool watch_pending = false, force_rehash = false, auto_grow = true;
    bool force_refresh = false;    // the marks of evicted blocks crowd the table: the next substep re-inserts the live blocks into a cleared table (k_table_refresh)
    uint64_t table_refreshes = 0;
    bool bodies_pending = false;   // integrate_bodies of the last substep has not run yet (it rides in the next sort launch)
    bool gu_fused = false;   // this substep's grid update rode in its P2G launch
    bool shard_fused = false; // sharded substep: the pack waves and the interior blocks' grid update rode in the P2G launch
    uint32_t grid_grown = 0;            // times the block capacity was doubled
    uint32_t watch_skips = 0;
    uint32_t cdf_generation = 1;        // bumped whenever cached node cdfs / block classes become invalid (kernels_sort.h regroup_block)
    uint32_t rehash_period = REHASH_PERIOD;  // substeps between unconditional table rebuilds (developer override: WGS_REHASH_PERIOD); 0 = none
                                             // but the first substep's: data whose long-inactive blocks are evicted (wgs_data_create decides)
    ShardLink *link = nullptr;          // wgs_shard_attach
    int reduce_impulses = 0;            // sharded two-way coupling: 1 = ncclAllReduce of the body impulses before
                                        // integrate_bodies, 2 = the caller sums them and integrates (lockstep group)
    int32_t **lockstep_imp_ptrs = nullptr;
    uint32_t lockstep_imp_n = 0;
};

namespace {

template <typename T> wgs_status dev_alloc(wgs_data *d, T **out, size_t count, bool zero = true) {
    void *p = nullptr;
    size_t bytes = sizeof(T) * (count ? count : 1);
    HIP_TRY(hipMalloc(&p, bytes));
    if (zero) HIP_TRY(hipMemsetAsync(p, 0, bytes, d->stream));
    d->allocs.push_back(p);
    d->alloc_bytes.push_back(bytes);
    d->device_bytes += bytes;
    *out = static_cast<T *>(p);
    return WGS_OK;
}

// Bodies that move need the impulse accumulation of P2G (rigid_impulses.wgsl reads it every substep). On sharded data
// every rank accumulates the impulses of its own particles and the fixed-point sums are reduced over the ranks before
// integrate_bodies (wgs_sharded_step: ncclAllReduce of 16 x 8 int32; integers, so the order does not matter).
wgs_status enable_impulses(wgs_data *d) {
    if (d->two_way) return WGS_OK;
    if (!d->dev.imp_slab) {  // per-block partial node impulses
        const size_t count = (size_t)d->dev.cap * Dim<D>::TILE * (D == 3 ? 2 : 1);
        wgs_status st = dev_alloc(d, &d->dev.imp_slab, count);
        if (st != WGS_OK) return st;
    }
    d->two_way = true;
    return WGS_OK;
}

// Every array sized by the block capacity (dev.cap, dev.hmask set by the caller). All of them are rebuilt by the
// sort of a table-rebuild substep, so a fresh zeroed set is a valid state (see grow_grid).
wgs_status alloc_grid(wgs_data *d) {
    Dev &dev = d->dev;
    wgs_status st = WGS_OK;
    const size_t hcap = (size_t)dev.hmask + 1, cap = dev.cap, nchunk = (cap + SCAN_CHUNK - 1) / SCAN_CHUNK;
#define GRID_ALLOC(ptr, count)                              \
    if ((st = dev_alloc(d, ptr, (size_t)(count))) != WGS_OK) return st
    GRID_ALLOC(&dev.hkeys, hcap);
    GRID_ALLOC(&dev.hvals, hcap);
    GRID_ALLOC(&dev.block_key, cap);
    if (!(dev.dbg & 1024u)) {   // (eviction of blocks long inactive — slabs of a decomposition too since round 6; WGS_DEBUG bit 10 = never, the table is rebuilt instead)
        GRID_ALLOC(&dev.block_slot, cap);
        GRID_ALLOC(&dev.free_ids, cap);
    }
    GRID_ALLOC(&dev.block_count, cap);
    GRID_ALLOC(&dev.block_stamp, cap);
    GRID_ALLOC(&dev.links_epoch, cap);
    GRID_ALLOC(&dev.block_acc, cap);
    GRID_ALLOC(&dev.block_dirty, cap);
    GRID_ALLOC(&dev.blk_narr, cap);
    GRID_ALLOC(&dev.block_ident, cap);
    GRID_ALLOC(&dev.blk_arr, cap * BLK_ARR);
    GRID_ALLOC(&dev.active, cap);
    GRID_ALLOC(&dev.block_start, cap);
    GRID_ALLOC(&dev.act_info, cap);
    GRID_ALLOC(&dev.act_cells, cap * NPB);
    GRID_ALLOC(&dev.nbr_plus, cap * 8);
    GRID_ALLOC(&dev.nbr_minus, cap * 8);
    GRID_ALLOC(&dev.nbr_known, cap * 16);
    GRID_ALLOC(&dev.act_src, cap * 8);
    GRID_ALLOC(&dev.cell_head, cap * NPB);
    GRID_ALLOC(&dev.chunk_a, nchunk);
    GRID_ALLOC(&dev.chunk_b, nchunk);
    GRID_ALLOC(&dev.group_a, nchunk * SORT_THREADS);
    GRID_ALLOC(&dev.group_b, nchunk * SORT_THREADS);
    GRID_ALLOC(&dev.cell_start, cap * NPB);
    GRID_ALLOC(&dev.cell_cursor, cap * NPB);
    GRID_ALLOC(&dev.nodes, cap * NPB);
    GRID_ALLOC(&dev.node_cdf, cap * NPB);
    GRID_ALLOC(&dev.slab, cap * Dim<D>::TILE);
    GRID_ALLOC(&dev.slab_epoch, cap);
    GRID_ALLOC(&dev.block_cdf_gen, cap);
    GRID_ALLOC(&dev.block_cpic, cap);
    GRID_ALLOC(&dev.block_cdf_summ, cap);
    GRID_ALLOC(&dev.pcdf_done, cap);
    GRID_ALLOC(&dev.cpic_list, (size_t)cap * 8);
    dev.visit_cap = dev.npad / 512u + 2u * cap + 16u;
    GRID_ALLOC(&dev.visit_list, (size_t)dev.visit_cap * 8);
    if (dev.sharded) GRID_ALLOC(&dev.halo_list, (size_t)cap * HALO_ENT);
    if (d->two_way) GRID_ALLOC(&dev.imp_slab, cap * Dim<D>::TILE * (D == 3 ? 2 : 1));
    if (dev.mesh_min) {
        GRID_ALLOC(&dev.mesh_min, cap * NPB);
        GRID_ALLOC(&dev.mesh_aff, cap * NPB);
    }
#undef GRID_ALLOC
    return WGS_OK;
}

void release_alloc(wgs_data *d, void *p) {
    if (!p) return;
    for (size_t i = 0; i < d->allocs.size(); i++)
        if (d->allocs[i] == p) {
            d->device_bytes -= d->alloc_bytes[i];
            d->allocs.erase(d->allocs.begin() + (long)i);
            d->alloc_bytes.erase(d->alloc_bytes.begin() + (long)i);
            break;
        }
    hipFree(p);
}

// SURVEY 8f4, second half — the reference's resize loop is a stub (src/grid/grid.rs:43-45,116-117: "TODO: resize the
// hashmap and retry"). Here the block capacity doubles BEFORE the table fills: a new zeroed set of grid arrays replaces
// the old one and the next substep rebuilds the table from the particles (the same full pass a table rebuild runs).
// Particle state is untouched, so nothing is lost; the stream is drained once (rare).
wgs_status grow_grid(wgs_data *d, uint32_t new_cap) {
    Dev &dev = d->dev;
    HIP_TRY(hipStreamSynchronize(d->stream));
    // The new set is allocated BEFORE the old one is released: if the device cannot hold both, the old table stays in
    // place, growth is switched off for this wgs_data and the run continues (an overflow is then reported as such).
    const Dev old = dev;
    const size_t first_new = d->allocs.size();
    dev.cap = new_cap;
    dev.hmask = new_cap * 2u - 1u;
    const wgs_status st = alloc_grid(d);  // (re-creates the optional arrays that are in use: two_way / mesh_min / sharded say so)
    if (st != WGS_OK) {
        while (d->allocs.size() > first_new) release_alloc(d, d->allocs.back());
        dev = old;
        d->auto_grow = false;
        hipGetLastError();  // (the failed hipMalloc is not this call's error)
        return WGS_OK;
    }
    void *old_ptrs[] = {old.hkeys, old.hvals, old.block_key, old.block_slot, old.free_ids, old.block_count, old.block_stamp, old.links_epoch, old.block_acc, old.block_dirty, old.blk_narr, old.blk_arr, old.block_ident, old.active,
                        old.block_start, old.act_info, old.act_cells, old.nbr_plus, old.nbr_minus, old.nbr_known, old.act_src, old.cell_head, old.chunk_a, old.chunk_b, old.group_a, old.group_b,
                        old.cell_start, old.cell_cursor, old.nodes, old.node_cdf, old.slab, old.slab_epoch, old.block_cdf_gen, old.block_cpic, old.block_cdf_summ, old.pcdf_done, old.cpic_list, old.visit_list, old.halo_list,
                        old.imp_slab, old.mesh_min, old.mesh_aff};
    for (void *p : old_ptrs) release_alloc(d, p);
    HIP_TRY(hipMemsetAsync(dev.hkeys, 0xff, sizeof(uint32_t) * ((size_t)dev.hmask + 1), d->stream));
    HIP_TRY(hipMemsetAsync(dev.hvals, 0xff, sizeof(uint32_t) * ((size_t)dev.hmask + 1), d->stream));
    HIP_TRY(hipMemsetAsync(dev.counters + CTR_NPHYS, 0, sizeof(uint32_t), d->stream));
    HIP_TRY(hipMemsetAsync(dev.counters + CTR_NFREE, 0, 3 * sizeof(uint32_t), d->stream));   // (free list, insertion count, marks)
    d->prev_sorted = false;      // block ids start over: the next substep bins every particle through the hash map
    d->prebinned = false;        // (what the last G2P binned went with the old arrays)
    d->cdf_generation++;
    d->last_ncpic = UINT32_MAX;
    d->last_nvisit = UINT32_MAX;
    d->grid_grown++;
    return WGS_OK;
}

// Looks at the counters the LAST wgs_step call left in pinned host memory (no synchronisation: skipped while that copy
// is still in flight) and keeps the table comfortable: more than half of the capacity active -> double it; more than
// three quarters of the ids handed out (blocks that were active at some point since the last rebuild) -> rebuild at
// the next substep instead of waiting for the 64-substep period.
wgs_status maintain_grid(wgs_data *d) {
    if (!d->watch || !d->watch_pending) return WGS_OK;
    if (hipEventQuery(d->watch_event) != hipSuccess) {
        // the host runs ahead of the device: let it, for two calls; then wait for the copy (a bounded run-ahead keeps
        // the observation fresh enough to act before the table fills)
        if (++d->watch_skips < 2u) return WGS_OK;
        HIP_TRY(hipEventSynchronize(d->watch_event));
    }
    d->watch_skips = 0;
    d->watch_pending = false;
    const uint32_t nblocks = d->watch[CTR_NBLOCKS], nphys = d->watch[CTR_NPHYS], cap = d->dev.cap;
    if (d->dev.sharded) d->nv_hint = std::max(d->watch[CTR_NV], d->watch[CTR_NV + CTR_SET]);
    // The observation is up to three calls old (two skips + the call that made it): a scene that is growing is judged by
    // where it will be by then at the rate of its last two observations, not by where it was.
    const uint32_t rate = nblocks > d->seen_nblocks && d->seen_nblocks != 0u ? nblocks - d->seen_nblocks : 0u;
    d->seen_nblocks = std::min(nblocks, cap);
    // (half full: grow, as before; or on course to be three quarters full by the time the next look can act)
    const uint64_t ahead = (uint64_t)nblocks + 3ull * rate;
    if (d->auto_grow && (nblocks > cap / 2u || ahead > cap / 4u * 3u) && cap < (1u << 24)) {
        uint32_t new_cap = cap * 2u;
        while (ahead > new_cap / 4u * 3u && new_cap < (1u << 24)) new_cap *= 2u;
        return grow_grid(d, new_cap);
    }
    if (nphys > cap / 4u * 3u) d->force_rehash = true;
    if (d->watch[CTR_NTOMB] > cap / 2u) d->force_refresh = true;   // (marks of evicted blocks: a quarter of the 2 x cap slots)
    return WGS_OK;
}

uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

void fill_collider(ColliderDev &c, const wgs_collider &in) {
    c.shape_type = in.shape_type;
    for (int k = 0; k < 4; k++) c.shape[k] = in.shape[k];
    for (int k = 0; k < 4; k++) c.rot[k] = in.pose.rotation[k];
    for (int k = 0; k < 3; k++) c.trans[k] = in.pose.translation[k];
    c.scale = in.pose.scale;
    for (int k = 0; k < 3; k++) c.linvel[k] = in.velocity.linear[k];
    for (int k = 0; k < 3; k++) c.angvel[k] = in.velocity.angular[k];
    for (int k = 0; k < 3; k++) c.com[k] = in.com[k];
}

constexpr uint32_t WGS_LAUNCH_SHAPE_SWITCHES = 2u | 4u | 8u | 128u | 1024u | 2048u | 4096u | 8192u | 16384u | 32768u | 65536u | 131072u | 262144u | 524288u | 1048576u | 4194304u | 8388608u | 16777216u | 33554432u | 67108864u;  // WGS_DEBUG bits the shipped library honours
#ifndef WGS_PCDF_WAVES_MAX_VISITS
#define WGS_PCDF_WAVES_MAX_VISITS 256
#endif
constexpr uint32_t PCDF_WAVES_MAX_VISITS = WGS_PCDF_WAVES_MAX_VISITS;   // per XCD list: above, the prologue waves would be a round of work in front of the launch, not a use of idle CUs
constexpr uint32_t P2G_SMALL_BUDGET_MIN_PARTICLES = 600000;  // one-way CPIC P2G body at 168 VGPRs from this size on
#ifndef WGS_REGROUP_ROUNDS
#define WGS_REGROUP_ROUNDS 4u
#endif
#ifndef WGS_GU_WG_PER_CU
#define WGS_GU_WG_PER_CU 8
#endif
constexpr uint32_t P2G_PAIR_MIN_BLOCKS = 8;  // near-collider blocks from which P2G runs both bodies in one launch
int grid_for(const wgs_data *d, int blocks_per_cu) { return d->pipeline->num_cus * blocks_per_cu; }

// ---- read-back kernels ---------------------------------------------------
struct ParticleOffsets {  // word offsets inside wgs_particle
    uint32_t stride, pos, vel, F, C, nrm, rvel, dist, aff, vol, rad, mass, lam, mu, has_pl, dp, has_ph, phase;
};

// Unpacked view of one particle slot (quad layout of layout.h).
struct Unpacked {
    float x[D], v[D], F[DD], C[DD], mass, vol, lam, mu;
    float nrm[D], rvel[D], dist;
    uint32_t aff;
    float dp[6], st[3], phase[2];
};

// index into a 3x3 matrix; the 3D branch below is parsed (never run) in the 2D library too
[[maybe_unused]] constexpr int m9(int k) { return DD == 9 ? k : 0; }

template <int DIM> __device__ inline void unpack_slot(const float *in, uint32_t npad, uint32_t j, bool plastic, bool cpic_in, uint32_t cdf_epoch, Unpacked &u) {
    // cdf quads are valid only if stamped with the epoch of the last substep (0 = echo the input)
    const bool cpic = cpic_in && (cdf_epoch == 0u || ldstamp<DIM>(in, npad, j) == cdf_epoch);
    using P = Pl<DIM>;
    if constexpr (DIM == 3) {
        const float4 xm = ldq(in, npad, P::XM, j), c0 = ldq(in, npad, P::CV0, j), c1 = ldq(in, npad, P::CV1, j),
                     c2 = ldq(in, npad, P::CV2, j), f0 = ldq(in, npad, P::F0, j), f1 = ldq(in, npad, P::F1, j),
                     f2 = ldq(in, npad, P::F2, j);
        u.x[0] = xm.x; u.x[1] = xm.y; u.x[D - 1] = xm.z; u.mass = xm.w;  // (uniform-material mode: fixed up by the caller)
        u.C[0] = c0.x; u.C[1] = c0.y; u.C[2] = c0.z; u.C[3] = c0.w;
        u.C[m9(4)] = c1.x; u.C[m9(5)] = c1.y; u.C[m9(6)] = c1.z; u.C[m9(7)] = c1.w; u.C[m9(8)] = c2.x;
        u.v[0] = c2.y; u.v[1] = c2.z; u.v[D - 1] = c2.w;
        u.F[0] = f0.x; u.F[1] = f0.y; u.F[2] = f0.z; u.F[3] = f0.w;
        u.F[m9(4)] = f1.x; u.F[m9(5)] = f1.y; u.F[m9(6)] = f1.z; u.F[m9(7)] = f1.w; u.F[m9(8)] = f2.x;
        u.vol = f2.y; u.lam = f2.z; u.mu = f2.w;
    } else {
        const float4 xm = ldq(in, npad, P::XM, j), c0 = ldq(in, npad, P::CV0, j), vl = ldq(in, npad, P::CV2, j),
                     f0 = ldq(in, npad, P::F0, j);
        u.x[0] = xm.x; u.x[1] = xm.y; u.mass = xm.z; u.vol = xm.w;
        u.C[0] = c0.x; u.C[1] = c0.y; u.C[2] = c0.z; u.C[3] = c0.w;
        u.v[0] = vl.x; u.v[1] = vl.y; u.lam = vl.z; u.mu = vl.w;
        u.F[0] = f0.x; u.F[1] = f0.y; u.F[2] = f0.z; u.F[3] = f0.w;
    }
    for (int k = 0; k < D; k++) { u.nrm[k] = 0.f; u.rvel[k] = 0.f; }
    u.dist = 0.f;
    u.aff = 0u;
    if (cpic) {
        const float4 a = ldq(in, npad, P::CDF0, j), b = ldq(in, npad, P::CDF1, j);
        u.nrm[0] = a.x; u.nrm[1] = a.y; u.rvel[0] = b.x; u.rvel[1] = b.y;
        if constexpr (DIM == 3) { u.nrm[D - 1] = a.z; u.dist = a.w; u.rvel[D - 1] = b.z; u.aff = __float_as_uint(b.w); }
        else { u.dist = a.z; u.aff = __float_as_uint(a.w); }
    }
    if (plastic) {
        const float4 d0 = ldq(in, npad, P::DP0, j), d1 = ldq(in, npad, P::DP1, j), d2 = ldq(in, npad, P::DP2, j);
        u.dp[0] = d0.x; u.dp[1] = d0.y; u.dp[2] = d0.z; u.dp[3] = d0.w; u.dp[4] = d1.x; u.dp[5] = d1.y;
        u.st[0] = d1.z; u.st[1] = d1.w; u.st[2] = d2.x; u.phase[0] = d2.y; u.phase[1] = d2.z;
    }
}

// uniform-material mode (layout.h): XM.w holds F[8], the four constants are kernel arguments
template <int DIM> __device__ inline void fix_uniform(const Dev &d, Unpacked &u) {
    if constexpr (DIM == 3) {
        if (d.uniform) {
            u.F[m9(8)] = u.mass;
            u.mass = d.uni_mass; u.vol = d.uni_vol; u.lam = d.uni_lambda; u.mu = d.uni_mu;
        }
    }
    // uniform plasticity parameters (layout.h Dev::uni_dp): in mode 2 DP1 holds (st0, st1, st2, phase) — unpack_slot read it as
    // (dp4, dp5, st0, st1) — and DP2 is not kept up to date
    if (d.uni_dp == 2u) {
        const float s0 = u.dp[4], s1 = u.dp[5], s2 = u.st[0], ph = u.st[1];
        u.st[0] = s0; u.st[1] = s1; u.st[2] = s2;
        u.phase[0] = ph; u.phase[1] = d.uni_max_stretch;
    }
    if (d.uni_dp != 0u)
        for (int k = 0; k < (d.uni_dp == 2u ? 6 : 4); k++) u.dp[k] = d.uni_dpv[k];
}

// general layout -> uniform-material layout: F[8] takes the place of the mass in XM.w
// `check`: the caller ASSERTED the constants (wgs_set_uniform_material on sharded data): a particle that carries other
// values would silently lose them, so every particle is compared bit for bit first and a mismatch is reported
// (ERRBIT_MATERIAL -> the next wgs_sync).
__global__ void k_to_uniform(Dev d, int side, int check) {
    if constexpr (D == 3) {
        float *buf = d.buf[side];
        const uint32_t n = num_slots(d);
        for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
            float4 xm = ldq(buf, d.npad, Pl<3>::XM, j);
            const float4 f2 = ldq(buf, d.npad, Pl<3>::F2, j);
            if (check && (__float_as_uint(xm.w) != __float_as_uint(d.uni_mass) || __float_as_uint(f2.y) != __float_as_uint(d.uni_vol) ||
                          __float_as_uint(f2.z) != __float_as_uint(d.uni_lambda) || __float_as_uint(f2.w) != __float_as_uint(d.uni_mu)))
                atomicOr(&d.counters[CTR_ERRORS], ERRBIT_MATERIAL);
            xm.w = f2.x;
            stq(buf, d.npad, Pl<3>::XM, j, xm);
        }
    }
}

__global__ void k_export_particles(Dev d, int side, ParticleOffsets o, bool plastic, bool cpic, uint32_t cdf_epoch, const float *s_radius,
                                   const float *s_dp, const float *s_phase, const uint32_t *s_flags, float *out,
                                   float *plastic_out) {
    const float *in = d.buf[side];
    const uint32_t npad = d.npad;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < d.n; j += gridDim.x * blockDim.x) {
        const uint32_t pid = ldpid<D>(in, npad, j);
        Unpacked u;
        unpack_slot<D>(in, npad, j, plastic, cpic, cdf_epoch, u);
        fix_uniform<D>(d, u);
        float *r = out + (size_t)pid * o.stride;
        for (int k = 0; k < D; k++) {
            r[o.pos + k] = u.x[k];
            r[o.vel + k] = u.v[k];
            r[o.nrm + k] = u.nrm[k];
            r[o.rvel + k] = u.rvel[k];
        }
        for (int k = 0; k < DD; k++) {
            r[o.F + k] = u.F[k];
            r[o.C + k] = u.C[k];
        }
        r[o.dist] = u.dist;
        r[o.aff] = __uint_as_float(u.aff);
        r[o.vol] = u.vol;
        r[o.rad] = s_radius[pid];
        r[o.mass] = u.mass;
        r[o.lam] = u.lam;
        r[o.mu] = u.mu;
        const uint32_t fl = s_flags[pid];
        r[o.has_pl] = __uint_as_float(fl & 1u);
        r[o.has_ph] = __uint_as_float((fl >> 1) & 1u);
        for (int k = 0; k < 6; k++) r[o.dp + k] = s_dp[(size_t)pid * 6 + k];
        r[o.phase] = plastic ? u.phase[0] : s_phase[(size_t)pid * 2];
        r[o.phase + 1] = plastic ? u.phase[1] : s_phase[(size_t)pid * 2 + 1];
        if (plastic_out)
            for (int k = 0; k < 3; k++) plastic_out[(size_t)pid * 3 + k] = plastic ? u.st[k] : (k < 2 ? 1.f : 0.f);
    }
}

// checkpoint restore: Drucker-Prager plastic state by persistent particle id (models/drucker_prager.wgsl:18-23)
__global__ void k_import_plastic_state(Dev d, int side, const float *states) {
    using P = Pl<D>;
    float *buf = d.buf[side];
    const uint32_t npad = d.npad;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < num_slots(d); j += gridDim.x * blockDim.x) {
        const uint32_t pid = ldpid<D>(buf, npad, j);
        if (pid == 0xffffffffu) continue;  // vacated slot of a sharded run
        const float *st = states + (size_t)pid * 3;
        float4 q1 = ldq(buf, npad, P::DP1, j), q2 = ldq(buf, npad, P::DP2, j);
        if (d.uni_dp == 2u) {   // (the state is one quad: layout.h)
            q1.x = st[0];
            q1.y = st[1];
            q1.z = st[2];
        } else {
            q1.z = st[0];
            q1.w = st[1];
            q2.x = st[2];
        }
        stq(buf, npad, P::DP1, j, q1);
        stq(buf, npad, P::DP2, j, q2);
    }
}

// Render hand-off: src_testbed/prep_vertex_buffer{2,3}d.wgsl `main`. Instance i = particle i of the
// caller's order; base_color is read from the instance record, everything else is written.
__global__ void k_prep_instances(Dev d, int side, uint32_t mode, bool cpic, uint32_t cdf_epoch, float *inst) {
    const float *in = d.buf[side];
    const uint32_t npad = d.npad;
    const float h = d.h, dt = d.sp->dt;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < d.n; j += gridDim.x * blockDim.x) {
        const uint32_t pid = ldpid<D>(in, npad, j);
        Unpacked u;
        unpack_slot<D>(in, npad, j, false, cpic, cdf_epoch, u);
        fix_uniform<D>(d, u);
        float *r = inst + (size_t)pid * 24;
        // deformation: mat3x3 as three padded columns (instancing3d.rs:66-74); 2D embeds F in the xy block
        float m[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        for (int col = 0; col < D; col++)
            for (int row = 0; row < D; row++) m[col * 3 + row] = u.F[col * D + row];
        for (int col = 0; col < 3; col++) {
            for (int row = 0; row < 3; row++) r[col * 4 + row] = m[col * 3 + row];
            r[col * 4 + 3] = 0.f;
        }
        r[12] = u.x[0]; r[13] = u.x[1]; r[14] = D == 3 ? u.x[D - 1] : 0.f; r[15] = 0.f;
        const float base[4] = {r[16], r[17], r[18], r[19]};
        float col[4] = {base[0], base[1], base[2], base[3]};
        if (mode == WGS_RENDER_VELOCITY) {
            for (int k = 0; k < D; k++) col[k] = fabsf(u.v[k]) * dt * 100.0f + 0.2f;
        } else if (mode == WGS_RENDER_VOLUME) {
            Svd<D> sv;
            svd<D>(u.F, sv);
            float s[3] = {sv.s[0], sv.s[1], D == 3 ? sv.s[D - 1] : 0.f};
            // descending order, like the reference's SVD (wgebra Svd2/Svd3, third party)
            if (s[0] < s[1]) { float t = s[0]; s[0] = s[1]; s[1] = t; }
            if (D == 3) {
                if (s[1] < s[2]) { float t = s[1]; s[1] = s[2]; s[2] = t; }
                if (s[0] < s[1]) { float t = s[0]; s[0] = s[1]; s[1] = t; }
            }
            for (int k = 0; k < D; k++) col[k] = (1.0f - s[k]) / 0.005f + 0.2f;
        } else if (mode == WGS_RENDER_CDF_NORMALS) {
            bool zero = true;
            for (int k = 0; k < D; k++) zero = zero && u.nrm[k] == 0.f;
            col[0] = col[1] = col[2] = 0.f;
            if (!zero)
                for (int k = 0; k < D; k++) col[k] = (u.nrm[k] + 1.0f) / 2.0f;
        } else if (mode == WGS_RENDER_CDF_DISTANCES) {
            const float dd = u.dist / (h * 1.5f);
            col[0] = dd > 0.f ? 0.f : fabsf(dd);
            col[1] = dd > 0.f ? fabsf(dd) : 0.f;
            col[2] = 0.f;
        } else if (mode == WGS_RENDER_CDF_SIGNS) {
            const uint32_t a = (u.aff >> 16) & (u.aff & 0xffffu);
            col[0] = (u.aff != 0u && a != 0u) ? 1.f : 0.f;
            col[1] = (u.aff != 0u && a == 0u) ? 1.f : 0.f;
            col[2] = 0.f;
        }
        r[20] = col[0]; r[21] = col[1]; r[22] = col[2]; r[23] = col[3];
    }
}

__global__ void k_export_positions(Dev d, int side, float *out) {
    const float *in = d.buf[side];
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < d.n; j += gridDim.x * blockDim.x) {
        const uint32_t pid = ldpid<D>(in, d.npad, j);
        const float4 xm = ldq(in, d.npad, P::XM, j);
        out[(size_t)pid * D + 0] = xm.x;
        out[(size_t)pid * D + 1] = xm.y;
        if (D == 3) out[(size_t)pid * D + D - 1] = xm.z;
    }
}

__global__ void k_export_grid(Dev d, uint32_t nblocks, bool cpic, wgs_node_record *out) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT;
    const uint32_t total = nblocks * NPB;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const uint32_t b = d.active[t >> 6], ln = t & 63u;
        const uint32_t node = b * NPB + ln;
        int bc[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[b], bc);
        int l[3] = {(int)(ln & (BW - 1)), (int)((ln >> BS) & (BW - 1)), D == 3 ? (int)(ln >> (2 * BS)) : 0};
        wgs_node_record r;
        for (int k = 0; k < D; k++) r.cell[k] = bc[k] * BW + l[k];
        float4 v = d.nodes[node];
        r.velocity[0] = v.x;
        r.velocity[1] = v.y;
        if (D == 3) { r.velocity[D - 1] = v.z; r.mass = v.w; } else { r.mass = v.z; }
        NodeCdf c = {0.f, 0u, NONE, 0u};
        if (cpic) c = d.node_cdf[node];
        r.cdf_distance = c.distance;
        r.cdf_affinities = c.affinities;
        r.cdf_closest_id = c.closest_id;
        out[t] = r;
    }
}

__global__ void k_export_blocks(Dev d, uint32_t nblocks, wgs_block_record *out) {
    for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < nblocks; a += gridDim.x * blockDim.x) {
        const uint32_t b = d.active[a];
        int bc[3] = {0, 0, 0};
        unpack_key<D>(d.block_key[b], bc);
        wgs_block_record r;
        for (int k = 0; k < D; k++) r.virtual_id[k] = bc[k];
        r.first_particle = d.block_start[b];
        r.num_particles = d.block_count[b];
        out[a] = r;
    }
}

wgs_status allreduce_impulses(wgs_data *d);  // capi_sharded.inc

// leaves a copy of the device counters in pinned host memory for the next call's maintain_grid (asynchronous)
wgs_status watch_counters(wgs_data *d) {
    if (!d->watch) {
        HIP_TRY(hipHostMalloc((void **)&d->watch, sizeof(uint32_t) * CTR_COUNT, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&d->watch_event, hipEventDisableTiming));
    }
    HIP_TRY(hipMemcpyAsync(d->watch, d->dev.counters, sizeof(uint32_t) * CTR_COUNT, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipEventRecord(d->watch_event, d->stream));
    d->watch_pending = true;
    return WGS_OK;
}

wgs_status fetch_counters(wgs_data *d) {
    uint32_t host[CTR_COUNT];
    HIP_TRY(hipMemcpyAsync(host, d->dev.counters, sizeof(host), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->last_nblocks = host[CTR_NBLOCKS] < d->dev.cap ? host[CTR_NBLOCKS] : d->dev.cap;
    d->seen_nblocks = d->last_nblocks;
    // (the list counters of the last substep: the set of its parity, layout.h; epoch of substep number s = s, counted from 1)
    const uint32_t last_epoch = (uint32_t)d->substeps;
    d->last_ncpic = 0;  // the eight lists together
    for (uint32_t k = 0; k < 8; k++) d->last_ncpic += std::min(host[ctr_ncpic(k, last_epoch)], d->dev.cap);
    d->last_ncpic = std::min(d->last_ncpic, d->dev.cap);
    d->last_nvisit = 0;  // the longest of the eight lists
    for (uint32_t k = 0; k < 8; k++) d->last_nvisit = std::max(d->last_nvisit, std::min(host[ctr_nvisit(k, last_epoch)], d->dev.visit_cap));
    uint32_t movers = 0u;   // (16 partial counts, each modulo 2^32: so is their sum)
    for (int k = 0; k < 16; k++) movers += host[CTR_MOVERS + 32 * k];
    d->movers_total += (uint32_t)(movers - d->last_movers);
    d->last_movers = movers;
    d->last_nphys = host[CTR_NPHYS];
    d->last_nfree = host[CTR_NFREE];
    d->last_ntomb = host[CTR_NTOMB];
    d->sticky_errors |= host[CTR_ERRORS];
    if (host[CTR_NBLOCKS] > d->dev.cap) d->sticky_errors |= ERRBIT_OVERFLOW;
    if (host[CTR_NPHYS] > d->dev.cap / 4u * 3u) d->force_rehash = true;
    if (host[CTR_NTOMB] > d->dev.cap / 2u) d->force_refresh = true;
    return WGS_OK;
}

wgs_status sticky_status(wgs_data *d) {
    if (d->sticky_errors & ERRBIT_OVERFLOW)
        return fail(WGS_ERR_GRID_OVERFLOW, "sparse grid overflow: more active blocks than grid_capacity");
    if (d->sticky_errors & ERRBIT_SHARD)
        return fail(WGS_ERR_INVALID_ARGUMENT, "sharded run: a message buffer or the particle capacity overflowed, a particle left the decomposition, or the ranks disagree on the uniform-material mode");
    if (d->sticky_errors & ERRBIT_KEYRANGE)
        return fail(WGS_ERR_KEY_RANGE, "a particle left the packed block-key range (grid.wgsl:88-95)");
    if (d->sticky_errors & ERRBIT_HANDOVER)
        return fail(WGS_ERR_HIP, "internal: a grid-update wave gave up waiting for a block's P2G slab (the sort's block totals and cell runs disagree)");
    if (d->sticky_errors & ERRBIT_PCDF)
        return fail(WGS_ERR_HIP, "internal: a near-collider workgroup of P2G gave up waiting for the prologue waves of its launch and computed the particle cdf itself (results intact; the launch lost 0.2 s)");
    if (d->sticky_errors & ERRBIT_MATERIAL)
        return fail(WGS_ERR_INVALID_ARGUMENT, "wgs_set_uniform_material: a particle of this wgs_data carries other constants (mass, init_volume, lambda, mu)");
    return WGS_OK;
}

void resolve_timings(wgs_data *d) {
    if (!d->timings_pending) return;
    hipStreamSynchronize(d->stream);
    for (int p = 0; p < WGS_NUM_PASSES; p++) d->timings[p] = 0.f;
    // marks: 0 start | 1 after sort | 2 after node cdf | 3 after particle cdf | 4 after p2g | 5 after grid update |
    //        6 after the fused g2p launch | 7 after its near-collider launch | 8 after integrate_bodies
    const int pass_of_mark[8] = {WGS_PASS_GRID_SORT,   WGS_PASS_GRID_UPDATE_CDF, WGS_PASS_G2P_CDF,          WGS_PASS_P2G,
                                 WGS_PASS_GRID_UPDATE, WGS_PASS_G2P,             WGS_PASS_PARTICLES_UPDATE, WGS_PASS_INTEGRATE_BODIES};
    d->mark_overhead_ms = 0.f;
    for (int s = 0; s < d->events.used; s++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, d->events.ev[s][9], d->events.ev[s][10]) == hipSuccess) d->mark_overhead_ms += ms;
    }
    if (d->events.used > 0) d->mark_overhead_ms /= (float)d->events.used;
    for (int s = 0; s < d->events.used; s++)
        for (int m = 0; m < 8; m++) {
            if (m == 6 && !d->cpic) continue;  // no second G2P launch: the two marks are adjacent
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, d->events.ev[s][m], d->events.ev[s][m + 1]) == hipSuccess)
                d->timings[pass_of_mark[m]] += ms;
        }
    d->timings_pending = false;
}

// One substep = pipeline.rs:201-280 (MPM passes), enqueued on the data's stream.
// part 0 = the whole substep (single GPU, or a slab stepped without its neighbours); the sharded step splits it around
// its one neighbour exchange: part 1 = sort .. P2G, part 2 = grid update + fused G2P (+ the arrivals' G2P) + bodies.
// `p2g_sel` splits part 1 further (wgs_sharded_step with neighbours): 0 = all of it; 1 = the sort only; 2 = P2G of the boundary
// layers with the pack waves behind it, on the data's SECOND stream; 3 = P2G of all other blocks with the interior's grid update.
template <bool TS> wgs_status enqueue_substep(wgs_data *d, int ts_slot, int part, int p2g_sel = 0) {
    Dev &dev = d->dev;
    hipStream_t s = (p2g_sel == 2 && d->stream2) ? d->stream2 : d->stream;
    const bool first = part != 2 && p2g_sel <= 1;   // the first call of this substep
    const int side = d->side;
    const uint32_t n = dev.n;
    const int pgrid = (int)((n + SORT_THREADS - 1) / SORT_THREADS);
    static const bool trace = getenv("WGS_TRACE") != nullptr;   // developer aid: drain the stream at every pass boundary and say so
    auto mark = [&](int m) {
        if (TS) hipEventRecord(d->events.ev[ts_slot][m], s);
        if (trace) {
            const hipError_t te = hipStreamSynchronize(s);
            fprintf(stderr, "[wgs trace] substep %llu part %d mark %d: %s\n", (unsigned long long)d->substeps, part, m, hipGetErrorString(te));
        }
    };
    const uint32_t epoch = (uint32_t)(d->substeps + 1);
    d->gu_fused = false;
    if (first) d->shard_fused = false;   // (part 2 of a sharded substep consumes what its part 1 decided)
    dev.ctr_set = (uint32_t)(d->substeps & 1u);  // sharded runs: the set of particle counters this substep reads (layout.h)
    // chunks of 64 sorted particles per wave of the fused G2P (kernels_transfer.h); the sort files the visit list by it
    // (2D: the body keeps no state of the chunk after the next one — at most two chunks per wave)
    const uint32_t nv_now = dev.sharded && d->nv_hint != 0u ? std::min(d->nv_hint, dev.nv) : dev.nv;   // (a slab launches for its capacity)
    dev.g2p_npass = (D == 3 && nv_now >= G2P_MANY_PASS_MIN_PARTICLES) ? (uint32_t)G2P_MANY_PASSES
                    : (nv_now >= G2P_TWO_PASS_MIN_PARTICLES || (dev.dbg & 131072u)) ? 2u : 1u;
    // Steady state: the buffer is in the sorted order of the previous substep, whose block ids, cell ids
    // (perm_cell) and neighbour links are still valid, so the particles are re-binned RELATIVE to their old
    // block (k_rebin: no hash lookups except for the few particles that changed block). The full k_bin runs
    // on the first substep, on table-rebuild substeps and in sharded runs (particles arrive from neighbours).
    const bool rehash = d->substeps == 0 || (d->rehash_period != 0u && d->substeps % d->rehash_period == 0) || (d->force_rehash && first);
    if (rehash && first) {
        d->table_rebuilds++;
        d->force_rehash = false;
        d->cdf_generation++;   // block ids are handed out anew
    }
    // node cdfs / block classes are reused from one substep to the next while no collider can move
    dev.cdf_gen = d->cpic ? d->cdf_generation : 0u;
    dev.cdf_moving = d->moving_mask;
    const bool fused_cdf = d->cpic && dev.n_rigid == 0;  // (mesh cdfs are only complete after k_p2g_cdf)
    if (first) dev.listed_in_perm = fused_cdf ? 1u : 0u;  // (part 2 of a sharded substep consumes what its part 1 wrote)
    const bool use_rebin = d->prev_sorted && !rehash && !(dev.dbg & 128u);
    // The fused G2P of this substep also bins its output for the next one (g2p_body.inc, Dev::bin_next; slabs too), unless
    // that substep rebuilds the table anyway (dbg bit 20 brings launch 1 of the sort, k_rebin, back: same results, tested).
    // `prebinned`: the previous substep's G2P did so for this one.
    const bool binned = use_rebin && d->prebinned;
    if (first && d->prebinned && !binned) {
        // (a table rebuild nobody could foresee — ids three quarters handed out, seen by the host in between: what the G2P
        // accumulated for the old ids is dropped; the stamps it left mean nothing once the ids are handed out anew)
        HIP_TRY(hipMemsetAsync(dev.block_acc, 0, sizeof(uint32_t) * (size_t)dev.cap, s));
        HIP_TRY(hipMemsetAsync(dev.cell_head, 0, sizeof(uint32_t) * (size_t)dev.cap * NPB, s));
        HIP_TRY(hipMemsetAsync(dev.blk_narr, 0, sizeof(uint32_t) * (size_t)dev.cap, s));
    }
    if (first) d->prebinned = false;
    // (not the plastic variants: their fused G2P is compiled without the binning — kernels_transfer.h: the code alone, beyond the
    // instruction cache, cost a third of the launch — and launch 1 of the sort, k_rebin, stays)
    // (a slab: its fused G2P bins the residents — the guests it drops leave their block's total —, k_g2p_arrivals the particles that
    // arrive; both parts of a sharded substep see the same value)
    dev.bin_next = (!d->plastic && !(dev.dbg & (128u | 1048576u)) && (d->rehash_period == 0u || (d->substeps + 1) % d->rehash_period != 0)) ? 1u : 0u;
    // the fused G2P drops the guests only inside the sharded step (kernels_shard.h); wgs_step on a slab advances what it holds
    dev.skip_guests = (d->in_sharded_step && dev.sharded) ? 1u : 0u;
    if (dev.sharded && d->needs_compact && first) {
        hipLaunchKernelGGL(k_shard_compacted, dim3(1), dim3(64), 0, s, dev);
        d->needs_compact = false;
    }
    if (first) {
        if (TS) {  // two adjacent marks: their distance is what every interval below pays for its closing mark
            mark(9);
            mark(10);
        }
        mark(0);
        // ---- "grid sort" (grid.rs:30-207)
        if (d->force_refresh && !rehash && dev.free_ids != nullptr) {
            // the marks of evicted blocks crowd the table (the host's last look): clear it and insert the live blocks again under
            // their own ids — no particle is touched, the steady-state sort goes on (kernels_sort.h k_table_refresh)
            HIP_TRY(hipMemsetAsync(dev.hkeys, 0xff, sizeof(uint32_t) * ((size_t)dev.hmask + 1), s));
            HIP_TRY(hipMemsetAsync(dev.hvals, 0xff, sizeof(uint32_t) * ((size_t)dev.hmask + 1), s));
            hipLaunchKernelGGL(k_table_refresh, dim3(std::max(1u, std::min((dev.cap + 255u) / 256u, (uint32_t)grid_for(d, 4)))), dim3(256), 0, s, dev);
            d->table_refreshes++;
        }
        if (first) d->force_refresh = false;
        if (rehash) {  // reset_hmap, amortised (device_math.h)
            HIP_TRY(hipMemsetAsync(dev.hkeys, 0xff, sizeof(uint32_t) * ((size_t)dev.hmask + 1), s));
            HIP_TRY(hipMemsetAsync(dev.hvals, 0xff, sizeof(uint32_t) * ((size_t)dev.hmask + 1), s));
            HIP_TRY(hipMemsetAsync(dev.counters + CTR_NPHYS, 0, sizeof(uint32_t), s));
            HIP_TRY(hipMemsetAsync(dev.counters + CTR_NFREE, 0, 3 * sizeof(uint32_t), s));   // (free list, insertion count, marks: layout.h)
        }
        // ---- "update rigid particles" (rigid_particle_update.wgsl): samples and vertices of the mesh colliders
        if (dev.n_rigid > 0)
            hipLaunchKernelGGL(k_rigid_transform<D>, dim3(grid_for(d, 1)), dim3(256), 0, s, dev);
        if (n > 0) {
            // (sharded runs: k_rebin also bins the particles that arrived in the last substep, behind the residents)
            // (a pending integrate_bodies of the previous substep rides in workgroup 0 of this launch)
            const uint32_t do_bodies = d->bodies_pending ? 1u : 0u;
            d->bodies_pending = false;
            if (binned) {   // launch 1 ran inside the previous substep's fused G2P
                if (do_bodies) hipLaunchKernelGGL(k_bodies_integrate<D>, dim3(1), dim3(16), 0, s, dev);
            } else if (use_rebin) hipLaunchKernelGGL(k_rebin<D>, dim3((pgrid + REBIN_K - 1) / REBIN_K), dim3(SORT_THREADS), 0, s, dev, side, epoch, do_bodies);
            else hipLaunchKernelGGL(k_bin<D>, dim3(pgrid), dim3(SORT_THREADS), 0, s, dev, side, epoch, do_bodies);
            if (dev.n_rigid > 0) {  // blocks a mesh sample reaches must exist (sort.wgsl:38-86)
                hipLaunchKernelGGL(k_rigid_mark<D>, dim3(grid_for(d, 1)), dim3(256), 0, s, dev, epoch);
                hipLaunchKernelGGL(k_rigid_touch<D>, dim3(grid_for(d, 1)), dim3(256), 0, s, dev, epoch);
            }
            // launch 2: chunked scan (active list, first_particle) + per-block setup and regrouping in canonical order.
            // Collider simulations without mesh colliders: node cdf + block classes ride in this launch, the particle
            // cdf in the CPIC P2G launch (no CDF launch at all)
            {
                const uint32_t nscan = (dev.cap + SCAN_CHUNK - 1) / SCAN_CHUNK;
                // one resident round: 4 workgroups per CU (127 VGPRs, 36 KB of LDS), the scan workgroups among them
                const uint32_t nreg = std::max(1u, std::min((dev.cap + 3u) / 4u, WGS_REGROUP_ROUNDS * ((uint32_t)grid_for(d, 4) - std::min(nscan, (uint32_t)grid_for(d, 2)))));
                const dim3 g(nscan + nreg);
                const int have_old = use_rebin ? 1 : 0;
                // (summ: every block within reach of a collider is evaluated substep after substep — each evaluates its own nodes and
                // tells its neighbours, kernels_sort.h block_cdf_summ; with colliders at rest: the instantiation without)
                const bool summ = (dev.cdf_moving != 0u || dev.cdf_gen == 0u) && !(dev.dbg & 2048u);
                if (fused_cdf && dev.sharded && summ) hipLaunchKernelGGL((k_regroup<D, true, true, true>), g, dim3(SORT_THREADS), 0, s, dev, side, epoch, nscan, have_old);
                else if (fused_cdf && dev.sharded) hipLaunchKernelGGL((k_regroup<D, true, true>), g, dim3(SORT_THREADS), 0, s, dev, side, epoch, nscan, have_old);
                else if (fused_cdf && summ) hipLaunchKernelGGL((k_regroup<D, true, false, true>), g, dim3(SORT_THREADS), 0, s, dev, side, epoch, nscan, have_old);
                else if (fused_cdf) hipLaunchKernelGGL((k_regroup<D, true, false>), g, dim3(SORT_THREADS), 0, s, dev, side, epoch, nscan, have_old);
                else if (dev.sharded) hipLaunchKernelGGL((k_regroup<D, false, true>), g, dim3(SORT_THREADS), 0, s, dev, side, epoch, nscan, have_old);
                else hipLaunchKernelGGL((k_regroup<D, false, false>), g, dim3(SORT_THREADS), 0, s, dev, side, epoch, nscan, have_old);
            }
        } else {
            HIP_TRY(hipMemsetAsync(dev.counters + CTR_NBLOCKS, 0, sizeof(uint32_t), s));
        }
        mark(1);
        // ---- "grid_update_cdf" + "g2p_cdf" (collide.wgsl, grid_update_cdf.wgsl, g2p_cdf.wgsl): one launch
        // (kernels_cdf.h); the reference's two pass names share its time in wgs_read_timings
        if (dev.n_rigid > 0 && n > 0)  // "p2g_cdf": mesh primitives -> node cdf accumulators
            hipLaunchKernelGGL(k_p2g_cdf<D>, dim3(std::min((dev.n_rigid * 32u + 255u) / 256u, (uint32_t)grid_for(d, 32))), dim3(256), 0, s, dev, epoch);
        if (d->cpic && n > 0 && !fused_cdf)
            hipLaunchKernelGGL(k_cdf<D>, dim3(grid_for(d, 16)), dim3(CDF_THREADS), 0, s, dev, side, epoch);
        mark(2);
        mark(3);
    }
    if (part != 2 && p2g_sel != 1) {
        const uint32_t layer_sel = p2g_sel == 2 ? 1u : p2g_sel == 3 ? 2u : 0u;   // (kernels_transfer.h: boundary layers / the others)
        if (n > 0) {
            // ---- "p2g"
            // Workgroups per body: about one per two entries of the block list (as the host last saw it), between 8 and
            // 32 per CU. A workgroup strides over the list, and the dispatcher balances better than a fixed stride does:
            // blocks differ in cost, and with 5 per CU — one resident round and a quarter — the quarter started when the
            // first workgroups retired (C5, 16 M particles: P2G 472 -> 346 us; C2: 35.6 -> 31.8 us). Same results for
            // any grid: a block's slab is the work of one workgroup.
            const uint32_t p2g_wgs = std::min((uint32_t)grid_for(d, 32), std::max((uint32_t)grid_for(d, 8), (d->seen_nblocks / 2u + 255u) & ~255u));
            const dim3 p2g_grid(p2g_wgs), p2g_block(P2GCfg<D>::NW * 64);
            // Large one-way collider simulations ALWAYS run the paired launch, with the CPIC body cut to 168 VGPRs: the
            // plain body then keeps its occupancy, so the pair costs nothing while the list is empty, and the choice
            // does not follow the host's syncs (the two budgets differ in the last bit here and there).
            const bool big_one_way = !d->two_way && n >= P2G_SMALL_BUDGET_MIN_PARTICLES && !(dev.dbg & 32768u);
            // Single-domain simulations: the grid update rides in the (last) P2G launch as workgroups of its own
            // behind the P2G workgroups (kernels_transfer.h gu_waves; GU = 2), one wave per active block as the host last saw
            // them; a P2G launch before it hands its slabs over the same way (GU = 1). Same results as the launch of its own
            // (dbg bit 18 brings that back): the same sums in the same order.
            const bool fuse_gu = part == 0 && !dev.sharded && !(dev.dbg & 262144u);
            // Inside wgs_sharded_step (part 1 of a slab's substep): behind the P2G workgroups ride the waves that pack the
            // outgoing messages (no k_pack_face launch) and the grid update of the INTERIOR blocks — everything that does
            // not wait for the exchange; the interface layers are updated after it (GU = 3).
            const bool fuse_shard = part == 1 && d->in_sharded_step && d->link && d->link->attached && !(dev.dbg & 262144u);
            const int gum = fuse_gu ? 2 : fuse_shard ? 3 : 0;   // what rides in the LAST P2G launch of this substep
            const uint32_t NW = (uint32_t)P2GCfg<D>::NW;
            // (8, 16, 32 or 64 workgroups per CU at most: the same times at C2 / C3 / C5)
            const uint32_t gu_wgs = (gum == 0 || p2g_sel == 2) ? 0u : std::min((uint32_t)grid_for(d, 8), std::max((uint32_t)grid_for(d, 1), ((d->seen_nblocks + NW - 1u) / NW + 7u) & ~7u));
            // pack waves: one per interface block as the host last saw the grid (a face holds a fraction of the active
            // blocks), plus a few for the guests
            uint32_t npack = 0u, npack_blk = 0u;
            if (fuse_shard && (d->link->has_lower || d->link->has_upper) && p2g_sel != 3) {   // (they ride behind the boundary layers' P2G)
                npack_blk = std::max(64u, std::min(2048u, d->seen_nblocks ? d->seen_nblocks : 2048u));
                const uint32_t nmig = std::max(1u, std::min(64u, (2u * d->link->mig_cap + 63u) / 64u));
                npack = (npack_blk + nmig + NW - 1u) / NW;
            }
            d->gu_fused = fuse_gu;
            d->shard_fused = fuse_shard;
            const uint32_t ride = npack + gu_wgs;   // workgroups behind the P2G workgroups
            // Large TWO-WAY simulations never pair: the kernel would take the two-way CPIC body's 225 registers and the plain
            // body — nearly every block — would run at two thirds of its occupancy (C4, 8 M particles: P2G 416 -> 347 us
            // with the two launches). Bit-identical either way (the same body text under -ffp-contract=on).
            const bool big_two_way = d->two_way && n >= P2G_SMALL_BUDGET_MIN_PARTICLES;
            // Prologue waves (kernels_transfer.h pcdf_waves): the particle cdf of the listed blocks by one wave per visit-list entry in front
            // of the paired launch, while the lists are short enough for the idle part of the chip to take them at once (as of the
            // host's last look: the waves stride over whatever the lists hold now). Single-domain data only: a slab's pack waves read
            // the guests' quads inside the launch. WGS_DEBUG bit 2 (value 4) = never.
            uint32_t npro = 0u;
            if (d->cpic && !dev.sharded && d->last_nvisit != UINT32_MAX && d->last_nvisit != 0u && d->last_nvisit <= PCDF_WAVES_MAX_VISITS && !(dev.dbg & 4u))
                npro = 8u * ((std::min(d->last_nvisit + 8u, dev.visit_cap) + NW - 1u) / NW);
            if (d->cpic && !dev.sharded && (dev.dbg & 8u)) npro = 8u;   // (tests: waves sized for a list the host never saw grow — the launch itself then decides, device_math.h pcdf_waves_on)
            dev.pcdf_waves = 0u;
#define WGS_P2G_PAIR(TW, WPE)                                                                                                          \
    do {                                                                                                                               \
        dev.pcdf_waves = npro;                                                                                                                 \
        const dim3 pg(npro + 2u * p2g_wgs + ride);                                                                                     \
        if (gum == 2) hipLaunchKernelGGL((k_p2g_pair<D, TW, WPE, 2>), pg, p2g_block, 0, s, dev, side, epoch, p2g_wgs, npack, npack_blk, layer_sel, npro);      \
        else if (gum == 3) hipLaunchKernelGGL((k_p2g_pair<D, TW, WPE, 3>), pg, p2g_block, 0, s, dev, side, epoch, p2g_wgs, npack, npack_blk, layer_sel, npro); \
        else hipLaunchKernelGGL((k_p2g_pair<D, TW, WPE, 0>), pg, p2g_block, 0, s, dev, side, epoch, p2g_wgs, npack, npack_blk, layer_sel, npro);               \
        dev.pcdf_waves = 0u;                                                                                                           \
    } while (0)
#define WGS_P2G_LAST(CP, TW, PC, FILTER)                                                                                                        \
    do {                                                                                                                                        \
        const uint32_t np_ = (PC) ? npro : 0u;                                                                                                  \
        dev.pcdf_waves = np_;                                                                                                                    \
        const dim3 lg(np_ + p2g_wgs + ride);                                                                                                    \
        if (gum == 2) hipLaunchKernelGGL((k_p2g<D, CP, TW, PC, 2>), lg, p2g_block, 0, s, dev, side, FILTER, epoch, p2g_wgs, npack, npack_blk, layer_sel, np_);        \
        else if (gum == 3) hipLaunchKernelGGL((k_p2g<D, CP, TW, PC, 3>), lg, p2g_block, 0, s, dev, side, FILTER, epoch, p2g_wgs, npack, npack_blk, layer_sel, np_);   \
        else hipLaunchKernelGGL((k_p2g<D, CP, TW, PC, 0>), lg, p2g_block, 0, s, dev, side, FILTER, epoch, p2g_wgs, npack, npack_blk, layer_sel, np_);                 \
        dev.pcdf_waves = 0u;                                                                                                                    \
    } while (0)
            // (small two-way scenes pair whatever the list length: they fill less than one round of workgroups, so the plain body's lost
            // occupancy costs nothing and a launch goes — the reference's sand2, 490 k particles, 2D: 76-79 -> 67-68 us per substep;
            // the one-way 262 k cube: P2G 20.4 + a boundary -> 18.3 us, not taken: its fused G2P then ran 27 us every other run against 21-22)
            const bool small_two_way = d->two_way && n < P2G_SMALL_BUDGET_MIN_PARTICLES;
            if (d->cpic && !big_two_way && (big_one_way || small_two_way || (d->last_ncpic != UINT32_MAX && d->last_ncpic >= P2G_PAIR_MIN_BLOCKS)) && !(dev.dbg & 8192u)) {
                // many blocks near colliders (as of the last wgs_sync): both bodies in one launch (k_p2g_pair)
                if (d->two_way) WGS_P2G_PAIR(true, 1);
                else if (big_one_way) WGS_P2G_PAIR(false, 3);
                else WGS_P2G_PAIR(false, 1);
            } else if (d->cpic && big_two_way && gum == 2 && !(dev.dbg & 67108864u)) {
                // Large two-way simulations on a single domain: the near-collider launch FIRST, the plain launch behind it with the grid
                // update riding in IT. The grid-update waves take the registers of the launch they ride in: behind the two-way body (209
                // registers, two waves per SIMD) the update of every block of the scene ran at two thirds of the occupancy it has behind
                // the plain body (160), and started only when the last near-collider workgroup — a 30 us chain each — had a slot. Same
                // sums in the same order (WGS_DEBUG bit 26 = the plain launch first, as before: tested bit-identical).
                {
                    dev.pcdf_waves = npro;
                    const dim3 lg(npro + p2g_wgs);
                    hipLaunchKernelGGL((k_p2g<D, true, true, true, 1>), lg, p2g_block, 0, s, dev, side, 2, epoch, p2g_wgs, 0u, 0u, layer_sel, npro);
                    dev.pcdf_waves = 0u;
                }
                hipLaunchKernelGGL((k_p2g<D, false, false, false, 2, true>), dim3(p2g_wgs + ride), p2g_block, 0, s, dev, side, 1, epoch, p2g_wgs, npack, npack_blk, layer_sel, 0u);
            } else if (d->cpic) {
                // (the first of the two launches hands its slabs over like the last one when anything rides in that one)
                if (gum != 0) hipLaunchKernelGGL((k_p2g<D, false, false, false, 1>), p2g_grid, p2g_block, 0, s, dev, side, 1, epoch, p2g_wgs, 0u, 0u, layer_sel, 0u);
                else hipLaunchKernelGGL((k_p2g<D, false>), p2g_grid, p2g_block, 0, s, dev, side, 1, epoch, p2g_wgs, 0u, 0u, layer_sel, 0u);
                // near-collider list: particle cdf in the prologue (the node cdfs are complete: k_setup_scatter<CDF>, or
                // k_cdf after k_p2g_cdf with mesh colliders), then the CPIC transfer
                if (d->two_way) WGS_P2G_LAST(true, true, true, 2);
                else WGS_P2G_LAST(true, false, true, 2);
            } else {
                WGS_P2G_LAST(false, false, false, 0);
            }
#undef WGS_P2G_PAIR
#undef WGS_P2G_LAST
        }
        mark(4);
    }
    if (part != 1) {
        if (n > 0 && !(part == 0 && d->gu_fused)) {
            // ---- "grid_update" (single-domain simulations: done by waves of the P2G launch above)
            if (part == 0 && d->two_way)
                hipLaunchKernelGGL((k_grid_update<D, 0, true>), dim3(grid_for(d, WGS_GU_WG_PER_CU)), dim3(256), 0, s, dev, epoch, 0u);
            else if (part == 0) hipLaunchKernelGGL((k_grid_update<D, 0>), dim3(grid_for(d, WGS_GU_WG_PER_CU)), dim3(256), 0, s, dev, epoch, 0u);
            else if (d->two_way) hipLaunchKernelGGL((k_grid_update<D, 3, true>), dim3(grid_for(d, WGS_GU_WG_PER_CU)), dim3(256), 0, s, dev, epoch, d->shard_fused ? 1u : 0u);
            else hipLaunchKernelGGL((k_grid_update<D, 3>), dim3(grid_for(d, WGS_GU_WG_PER_CU)), dim3(256), 0, s, dev, epoch, d->shard_fused ? 1u : 0u);
        }
        mark(5);
        // sharded step: the particles that arrived with this substep's messages are advanced too (kernels_arrivals.h), by a
        // launch of their own behind the fused G2P. (As extra workgroups INSIDE that launch — first or last in its grid — they
        // made it 7-10 us longer at a 1 M slab for the 5 us launch they saved: measured twice in round 3, not kept.)
        const bool arrivals = part == 2 && d->in_sharded_step && d->link && d->link->attached;
        const uint32_t arr_most = arrivals ? ((d->link->has_lower ? 1u : 0u) + (d->link->has_upper ? 1u : 0u)) * d->link->mig_cap : 0u;
        if (dev.nv > 0) {
            // ---- "g2p" + "particles_update", fused
            // one single-wave workgroup per `npass` chunks of 64 sorted particles; multiple of 8: XCD-aware mapping
            const uint32_t npass = dev.g2p_npass;
            const int g = (int)(((dev.nv + G2P_THREADS * npass - 1) / (G2P_THREADS * npass) + 7) / 8) * 8;
#ifndef WGS_PLASTIC_WPE_DENSE
#define WGS_PLASTIC_WPE_DENSE 2
#endif
#ifndef WGS_PLASTIC_WPE
#define WGS_PLASTIC_WPE G2P_WAVES_PER_EU
#endif
            // (the decomposition is a template parameter of the fused G2P: kernels_transfer.h; a slab always takes the paired /
            // single-body launch shapes, the two-launch debug shape exists for single-domain data only)
#define WGS_LAUNCH_G2P(MODEL, PL, CM, NP, SH)                                                                          \
    hipLaunchKernelGGL((k_g2p_update<D, MODEL, PL, CM, NP, SH, !(PL)>), (CM) == 2 ? dim3(8 * (grid_for(d, 1) * 3 / 2)) : dim3(g), \
                       dim3(G2P_THREADS), 0, s, dev, side, epoch)
#define WGS_LAUNCH_G2P_PAIR(MODEL, PL, WPE, NP, SH)                                                                        \
    hipLaunchKernelGGL((k_g2p_pair<D, MODEL, PL, WPE, NP, SH, !(PL)>), dim3((uint32_t)g + 8u * nlist), dim3(G2P_THREADS), 0, s, \
                       dev, side, epoch, (uint32_t)g, nlist)
#define WGS_LAUNCH_G2P_NP(MODEL, PL, NP)    \
    do {                                    \
        if (d->cpic && (shard || !(dev.dbg & 4096u))) {                                                    \
            /* both bodies in one launch (k_g2p_pair) */                                                          \
            /* list waves (8 x nlist; the waves of an XCD stride over the runs of its visit list): 2 x the runs of the */ \
            /* longest list as the host last saw it */                                                             \
            /* (unknown: a wave and a half per SIMD) */                                                           \
            const uint32_t full = (uint32_t)grid_for(d, 1) * 3u / 2u;                                             \
            const uint32_t nlist = d->last_nvisit == UINT32_MAX ? full : std::min(full, std::max(8u, 2u * ((d->last_nvisit + std::min<uint32_t>(NP, WGS_G2P_LIST_PASSES) - 1u) / std::min<uint32_t>(NP, WGS_G2P_LIST_PASSES)))); \
            /* plastic scenes with a large share of listed blocks: the spill-free variant (kernels_transfer.h) */  \
            const bool dense = PL && d->last_ncpic != UINT32_MAX && d->last_ncpic * 2u >= std::max(1u, d->last_nblocks) && !(dev.dbg & 16384u); \
            if (dense && shard) WGS_LAUNCH_G2P_PAIR(MODEL, PL, (PL) ? WGS_PLASTIC_WPE_DENSE : G2P_WAVES_PER_EU, NP, true);      \
            else if (dense) WGS_LAUNCH_G2P_PAIR(MODEL, PL, (PL) ? WGS_PLASTIC_WPE_DENSE : G2P_WAVES_PER_EU, NP, false);         \
            else if (shard) WGS_LAUNCH_G2P_PAIR(MODEL, PL, (PL) ? WGS_PLASTIC_WPE : G2P_WAVES_PER_EU, NP, true);                 \
            else WGS_LAUNCH_G2P_PAIR(MODEL, PL, (PL) ? WGS_PLASTIC_WPE : G2P_WAVES_PER_EU, NP, false);                           \
            mark(6);                                                                                              \
        } else if (d->cpic) {               \
            WGS_LAUNCH_G2P(MODEL, PL, 1, NP, false);   \
            mark(6);                        \
            WGS_LAUNCH_G2P(MODEL, PL, 2, 1, false);    \
        } else if (shard) {                 \
            WGS_LAUNCH_G2P(MODEL, PL, 0, NP, true);    \
        } else {                            \
            WGS_LAUNCH_G2P(MODEL, PL, 0, NP, false);   \
        }                                   \
    } while (0)
#define WGS_LAUNCH_G2P_MP(MODEL, PL)                    \
    do {                                                \
        if (npass == (uint32_t)G2P_MANY_PASSES) WGS_LAUNCH_G2P_NP(MODEL, PL, G2P_MANY_PASSES); \
        else if (npass == 2u) WGS_LAUNCH_G2P_NP(MODEL, PL, 2); \
        else WGS_LAUNCH_G2P_NP(MODEL, PL, 1);           \
    } while (0)
            const bool shard = dev.sharded != 0u;
            const int sel = (dev.model == WGS_MODEL_NEO_HOOKEAN ? 2 : 0) | (d->plastic ? 1 : 0);
            switch (sel) {
                case 0: WGS_LAUNCH_G2P_MP(0, false); break;
                case 1: WGS_LAUNCH_G2P_MP(0, true); break;
                case 2: WGS_LAUNCH_G2P_MP(1, false); break;
                default: WGS_LAUNCH_G2P_MP(1, true); break;
            }
#undef WGS_LAUNCH_G2P_MP
#undef WGS_LAUNCH_G2P_NP
#undef WGS_LAUNCH_G2P_PAIR
#undef WGS_LAUNCH_G2P
        }
        if (!(d->cpic && dev.nv > 0)) mark(6);  // (collider simulations: recorded between the two G2P launches)
        // (the arrivals' body also does the bookkeeping of the migration round, so it runs even when nobody can arrive)
        if (arrivals) {
            const dim3 ag(std::max(1u, std::min((arr_most + ARR_PER_WG - 1u) / ARR_PER_WG, 1024u)));
            const int asel = (dev.model == WGS_MODEL_NEO_HOOKEAN ? 2 : 0) | (d->plastic ? 1 : 0);
            switch (asel) {
                case 0: hipLaunchKernelGGL((k_g2p_arrivals<D, 0, false>), ag, dim3(256), 0, s, dev, side, epoch); break;
                case 1: hipLaunchKernelGGL((k_g2p_arrivals<D, 0, true>), ag, dim3(256), 0, s, dev, side, epoch); break;
                case 2: hipLaunchKernelGGL((k_g2p_arrivals<D, 1, false>), ag, dim3(256), 0, s, dev, side, epoch); break;
                default: hipLaunchKernelGGL((k_g2p_arrivals<D, 1, true>), ag, dim3(256), 0, s, dev, side, epoch); break;
            }
        }
        mark(7);
        // ---- "integrate_bodies" (rigid_impulses.wgsl:95-136) + the world mass properties of the next substep
        // (pipeline.rs:204-205). Skipped while no body has a velocity or a mass: it would be the identity.
        if (d->bodies_move && dev.n_colliders > 0 && !(part == 2 && d->reduce_impulses == 2)) {
            if (part == 2 && d->reduce_impulses == 1) {
                wgs_status rst = allreduce_impulses(d);
                if (rst != WGS_OK) return rst;
            }
            // Single-domain simulations without mesh colliders: left to the first launch of the next substep (or to the end of
            // this wgs_step call, flush_bodies) — a 16-thread launch of its own costs a dependent launch, ~5 us, per substep.
            // (not when this substep's G2P binned for the next one: that substep has no launch in front of the node cdfs of its sort)
            if (part == 0 && !dev.sharded && dev.n_rigid == 0 && n > 0 && !(dev.dbg & 524288u) && !dev.bin_next) d->bodies_pending = true;
            else hipLaunchKernelGGL(k_bodies_integrate<D>, dim3(1), dim3(16), 0, s, dev);
        }
        mark(8);
        d->side ^= 1;
        d->substeps++;
        d->prev_sorted = true;
        d->prebinned = dev.bin_next != 0u && dev.nv > 0;
        dev.n = dev.nv;  // the buffer just written holds the valid particles only, in sorted order
        // sharded: the counters of the new buffer (CTR_N / CTR_NPREV / CTR_NV) are set by k_g2p_arrivals; a slab stepped
        // without its neighbours (wgs_step) sets them at the head of its next substep (k_shard_compacted)
        if (dev.sharded && !arrivals) d->needs_compact = true;
    }
    HIP_TRY(hipGetLastError());
    return WGS_OK;
}

// integrate_bodies of a lockstep group (the group summed the impulses of its slabs after every slab's grid update)
wgs_status enqueue_bodies(wgs_data *d) {
    if (d->bodies_move && d->dev.n_colliders > 0) hipLaunchKernelGGL(k_bodies_integrate<D>, dim3(1), dim3(16), 0, d->stream, d->dev);
    HIP_TRY(hipGetLastError());
    return WGS_OK;
}

}  // namespace

extern "C" {

const char *wgs_last_error(void) { return g_last_error.c_str(); }
int32_t wgs_dim(void) { return D; }
uint32_t wgs_abi_version(void) { return WGS_ABI_VERSION; }
const char *wgs_build_info(void) {
    return "wgsparkl_hip dim=" WGS_STR(WGS_DIM) " arch=gfx950"
#ifdef WGS_ABLATE
           " WGS_ABLATE"
#endif
        ;
}

wgs_status wgs_pipeline_create(int32_t hip_device, wgs_pipeline **out) {
    if (!out) return fail(WGS_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(WGS_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (hip_device < 0 || hip_device >= count) return fail(WGS_ERR_INVALID_ARGUMENT, "hip_device out of range");
    HIP_TRY(hipSetDevice(hip_device));
    wgs_pipeline *p = new wgs_pipeline();
    p->device = hip_device;
    {
        const hipError_t pe = hipGetDeviceProperties(&p->props, hip_device);
        if (pe != hipSuccess) {
            delete p;
            return fail(WGS_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(pe));
        }
    }
    p->num_cus = p->props.multiProcessorCount > 0 ? p->props.multiProcessorCount : 256;
    *out = p;
    return WGS_OK;
}

void wgs_pipeline_destroy(wgs_pipeline *pipeline) { delete pipeline; }

static wgs_status create_impl(wgs_pipeline *pipeline, const wgs_sim_params *params, const wgs_particle *particles,
                              size_t num_particles, const uint32_t *global_ids, const wgs_collider *colliders,
                              size_t num_colliders, float cell_width, uint32_t grid_capacity, size_t particle_capacity,
                              bool sharded, int32_t block_lo, int32_t block_hi, int32_t force_plastic, wgs_data **out) {
    if (!pipeline || !params || !out) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (num_particles && !particles) return fail(WGS_ERR_INVALID_ARGUMENT, "particles is NULL");
    if (num_colliders && !colliders) return fail(WGS_ERR_INVALID_ARGUMENT, "colliders is NULL");
    if (num_colliders > WGS_MAX_COLLIDERS)
        return fail(WGS_ERR_UNSUPPORTED, "at most 16 coupled colliders (grid.wgsl:230-240)");
    if (!(cell_width > 0.f)) return fail(WGS_ERR_INVALID_ARGUMENT, "cell_width must be > 0");
    if (grid_capacity == 0 || grid_capacity > (1u << 25)) return fail(WGS_ERR_INVALID_ARGUMENT, "grid_capacity out of range");
    // 32-bit byte offsets inside one ping-pong buffer (layout.h ldp/stp)
    if (particle_capacity < num_particles) particle_capacity = num_particles;
    if (buffer_floats<D>((uint32_t)particle_capacity + 64) * 4 >= (1ull << 32))
        return fail(WGS_ERR_UNSUPPORTED, "more than ~21M particles per wgs_data: shard across GPUs");
    *out = nullptr;
    HIP_TRY(hipSetDevice(pipeline->device));
    wgs_data *d = new wgs_data();
    d->pipeline = pipeline;
    wgs_status st = WGS_OK;
    auto bail = [&](wgs_status code) {
        wgs_data_destroy(d);
        return code;
    };
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(WGS_ERR_HIP, "hipStreamCreate failed"));
    Dev &dev = d->dev;
    const uint32_t n = (uint32_t)num_particles;
    dev.n = n;
    dev.nv = n;
    dev.sharded = sharded ? 1u : 0u;
    dev.shard_lo = sharded ? block_lo : INT32_MIN;
    dev.shard_hi = sharded ? block_hi : INT32_MAX;
    d->capacity = (uint32_t)particle_capacity;
    dev.npad = (((uint32_t)particle_capacity + 63u) / 64u) * 64u;
    if (dev.npad == 0) dev.npad = 64;
    dev.cap = next_pow2(grid_capacity);  // grid.rs:283
    dev.hmask = dev.cap * 2u - 1u;  // half-full table (reference: exactly cap slots, quirk B4)
    dev.h = cell_width;
    dev.inv_h = 1.0f / cell_width;
    {
        int e = 0;
        dev.h_pow2 = (frexpf(cell_width, &e) == 0.5f) ? 1u : 0u;
    }
    dev.model = WGS_MODEL_COROTATED;
    // Developer switches (read once, here; 0 in production): A/B of launch shapes, SAME results — 128 = full k_bin on
    // every substep (no k_rebin), 1024 = no eviction of long-inactive blocks from the table (it is rebuilt when the ids run out instead),
    // 2048 = launch 2 of the sort never shares node-cdf summaries between neighbouring blocks (every wave evaluates its whole tile),
    // 2 = ... shares them but never waits for one (a neighbour's word that is not there at the first look is evaluated locally),
    // 4 = the particle cdf of the listed blocks always inside their CPIC workgroups of P2G (no prologue waves), 8 = prologue waves sized for an
    // empty list whatever the host saw (the launch then finds the lists too long for them and leaves the work to the workgroups),
    // 4096 = the two G2P bodies as two launches, 8192 = the two P2G bodies always as two launches, 16384 = never the
    // spill-free variant of the plastic G2P pair, 32768 = never the small register budget of the one-way P2G pair,
    // 65536 = never the uniform-material mode (the per-particle constants always travel with the particle), 131072 = the
    // fused G2P always with two chunks per wave (the large-scene launch shape), 262144 = the grid update as a launch of its
    // own also where it could ride in the P2G launch, 524288 = integrate_bodies as a launch of its own at the tail of every substep,
    // 1048576 = launch 1 of the sort (k_rebin) every substep instead of the binning inside the fused G2P, 4194304 = P2G of a
    // lockstep slab split into its boundary layers and the rest (the shape wgs_sharded_step uses when it forks), 8388608 =
    // wgs_sharded_step forks the exchange onto a second stream beside the interior's P2G (measured slower here: capi_sharded.inc),
    // 16777216 = launch 2 of the sort orders the cells of a dirty block by insertion instead of by ranks (kernels_sort.h),
    // 33554432 = P2G gathers every block through the sort permutation (no direct runs for unchanged blocks: layout.h CELL_DIRECT),
    // 67108864 = large two-way simulations run the plain P2G launch in front of the near-collider one (round 5's order).
    // The ablations that change the RESULTS (64 = G2P moves bytes only, 256 = P2G without its accumulation loop,
    // 512 = P2G without its particle loads) exist only in builds with -DWGS_ABLATE; the shipped library ignores them.
    dev.dbg = getenv("WGS_DEBUG") ? (uint32_t)strtoul(getenv("WGS_DEBUG"), nullptr, 0) : 0u;
    if (getenv("WGS_REHASH_PERIOD")) d->rehash_period = std::max(1u, (uint32_t)strtoul(getenv("WGS_REHASH_PERIOD"), nullptr, 0));  // same results
#ifndef WGS_ABLATE
    dev.dbg &= WGS_LAUNCH_SHAPE_SWITCHES;
#endif
    dev.n_colliders = (uint32_t)num_colliders;
    d->cpic = num_colliders > 0;

#define TRY_ALLOC(...)                      \
    do {                                    \
        st = dev_alloc(d, __VA_ARGS__);     \
        if (st != WGS_OK) return bail(st);  \
    } while (0)
    const size_t plane_floats = buffer_floats<D>(dev.npad);
    TRY_ALLOC(&dev.buf[0], plane_floats);
    TRY_ALLOC(&dev.buf[1], plane_floats);
    TRY_ALLOC(&dev.perm, (size_t)dev.npad);
    TRY_ALLOC(&dev.perm_cell, (size_t)dev.npad);
    TRY_ALLOC(&dev.cellid, (size_t)dev.npad);
    TRY_ALLOC(&dev.mv_next, (size_t)dev.npad);
    st = alloc_grid(d);
    if (st != WGS_OK) return bail(st);
    // data that evicts its long-inactive blocks needs no periodic table rebuild (the marks the evictions leave are cleared by
    // k_table_refresh, without touching a particle) — slabs of a decomposition included since round 6
    if (dev.free_ids != nullptr && !getenv("WGS_REHASH_PERIOD")) d->rehash_period = 0u;
    TRY_ALLOC(&dev.counters, (size_t)CTR_COUNT);
    TRY_ALLOC(&d->sp, (size_t)1);
    TRY_ALLOC(&d->colliders, (size_t)WGS_MAX_COLLIDERS);
    TRY_ALLOC(&dev.bodies, (size_t)WGS_MAX_COLLIDERS);
    TRY_ALLOC(&dev.impulses, (size_t)WGS_MAX_COLLIDERS * 8);
    TRY_ALLOC(&d->static_radius, (size_t)dev.npad);
    TRY_ALLOC(&d->static_dp, (size_t)dev.npad * 6);
    TRY_ALLOC(&d->static_phase, (size_t)dev.npad * 2);
    TRY_ALLOC(&d->static_flags, (size_t)dev.npad);
    TRY_ALLOC(&d->shard_counts, (size_t)4);
    if (sharded) {
        dev.leavers_cap = std::max<uint32_t>(4096u, (uint32_t)(particle_capacity / 16));
        TRY_ALLOC(&dev.leavers, (size_t)dev.leavers_cap);
    }
#undef TRY_ALLOC
    dev.sp = d->sp;
    dev.colliders = d->colliders;

    // AoS -> SoA staging (GpuParticles::from_particles + GpuModels::from_particles,
    // particle3d.rs:192-210, models/mod.rs:20-49).
    std::vector<float> soa(plane_floats, 0.f);
    std::vector<float> s_radius(dev.npad, 0.f), s_dp((size_t)dev.npad * 6, 0.f), s_phase((size_t)dev.npad * 2, 0.f);
    std::vector<uint32_t> s_flags(dev.npad, 0u);
    const float deg = 3.14159265358979323846f / 180.0f;
    const float default_dp[6] = {35.0f * deg, 9.0f * deg, 0.2f, 10.0f * deg, -1.0f, -1.0f};  // DruckerPrager::new(-1, -1)
    bool plastic = false;
    auto quad = [&](int qd, uint32_t i) { return soa.data() + ((size_t)qd * dev.npad + i) * 4; };
    uint32_t *pid_plane = reinterpret_cast<uint32_t *>(soa.data() + (size_t)P::NQ * 4 * dev.npad);
    for (uint32_t i = 0; i < n; i++) {
        const wgs_particle &q = particles[i];
        const wgs_particle_dynamics &dy = q.dynamics;
        float aff_bits;
        memcpy(&aff_bits, &dy.cdf.affinity, 4);
        if constexpr (D == 3) {
            using P3 = Pl<3>;
            float *p;
            p = quad(P3::XM, i); p[0] = q.position[0]; p[1] = q.position[1]; p[2] = q.position[D - 1]; p[3] = dy.mass;
            p = quad(P3::CV0, i); p[0] = dy.affine[0]; p[1] = dy.affine[1]; p[2] = dy.affine[2]; p[3] = dy.affine[3];
            p = quad(P3::CV0 + 1, i); p[0] = dy.affine[DD - 5]; p[1] = dy.affine[DD - 4]; p[2] = dy.affine[DD - 3]; p[3] = dy.affine[DD - 2];
            p = quad(P3::CV2, i); p[0] = dy.affine[DD - 1]; p[1] = dy.velocity[0]; p[2] = dy.velocity[1]; p[3] = dy.velocity[D - 1];
            p = quad(P3::F0, i); p[0] = dy.def_grad[0]; p[1] = dy.def_grad[1]; p[2] = dy.def_grad[2]; p[3] = dy.def_grad[3];
            p = quad(P3::F0 + 1, i); p[0] = dy.def_grad[DD - 5]; p[1] = dy.def_grad[DD - 4]; p[2] = dy.def_grad[DD - 3]; p[3] = dy.def_grad[DD - 2];
            p = quad(P3::F0 + 2, i); p[0] = dy.def_grad[DD - 1]; p[1] = dy.init_volume; p[2] = q.model.lambda; p[3] = q.model.mu;
            p = quad(P3::CDF0, i); p[0] = dy.cdf.normal[0]; p[1] = dy.cdf.normal[1]; p[2] = dy.cdf.normal[D - 1]; p[3] = dy.cdf.signed_distance;
            p = quad(P3::CDF1, i); p[0] = dy.cdf.rigid_vel[0]; p[1] = dy.cdf.rigid_vel[1]; p[2] = dy.cdf.rigid_vel[D - 1]; p[3] = aff_bits;
        } else {
            using P2 = Pl<2>;
            float *p;
            p = quad(P2::XM, i); p[0] = q.position[0]; p[1] = q.position[1]; p[2] = dy.mass; p[3] = dy.init_volume;
            p = quad(P2::CV0, i); p[0] = dy.affine[0]; p[1] = dy.affine[1]; p[2] = dy.affine[2]; p[3] = dy.affine[3];
            p = quad(P2::CV2, i); p[0] = dy.velocity[0]; p[1] = dy.velocity[1]; p[2] = q.model.lambda; p[3] = q.model.mu;
            p = quad(P2::F0, i); p[0] = dy.def_grad[0]; p[1] = dy.def_grad[1]; p[2] = dy.def_grad[2]; p[3] = dy.def_grad[3];
            p = quad(P2::CDF0, i); p[0] = dy.cdf.normal[0]; p[1] = dy.cdf.normal[1]; p[2] = dy.cdf.signed_distance; p[3] = aff_bits;
            p = quad(P2::CDF1, i); p[0] = dy.cdf.rigid_vel[0]; p[1] = dy.cdf.rigid_vel[1]; p[2] = 0.f; p[3] = 0.f;
        }
        pid_plane[i] = global_ids ? global_ids[i] : i;
        const float *dp = q.has_plasticity ? &q.plasticity.h0 : default_dp;
        const float phase = q.has_phase ? q.phase.phase : 0.0f;            // models/mod.rs:33-36
        const float max_stretch = q.has_phase ? q.phase.max_stretch : -1.0f;
        {
            float *p;
            p = quad(P::DP0, i); p[0] = dp[0]; p[1] = dp[1]; p[2] = dp[2]; p[3] = dp[3];
            // DruckerPragerPlasticState::default() = {1, 1, 0}, drucker_prager.rs:44-53
            p = quad(P::DP1, i); p[0] = dp[4]; p[1] = dp[5]; p[2] = 1.0f; p[3] = 1.0f;
            p = quad(P::DP2, i); p[0] = 0.0f; p[1] = phase; p[2] = max_stretch; p[3] = 0.f;
        }
        for (int k = 0; k < 6; k++) s_dp[(size_t)i * 6 + k] = dp[k];
        s_phase[(size_t)i * 2] = phase;
        s_phase[(size_t)i * 2 + 1] = max_stretch;
        s_radius[i] = q.dynamics.init_radius;
        s_flags[i] = (q.has_plasticity ? 1u : 0u) | (q.has_phase ? 2u : 0u);
        // Does the plasticity / fracture branch ever run for this particle?
        // (particle_update.wgsl:98-122; max_stretch >= FLT_MAX can never be exceeded by a finite F)
        if ((phase == 0.0f && dp[4] != 0.0f) || (phase > 0.0f && max_stretch > 0.0f && max_stretch < FLT_MAX)) plastic = true;
    }
    d->plastic = plastic || force_plastic != 0;
    // uniform plasticity parameters (bitwise; single-domain data, like the automatic uniform-material mode; layout.h Dev::uni_dp):
    // 1 = one set of h0..h3, 2 = all six and max_stretch — the per-particle state is then packed into DP1 before the upload
    if (d->plastic && n > 0 && !sharded && !(dev.dbg & 65536u)) {
        bool u4 = true, u6 = true;
        for (uint32_t i = 1; i < n && u4; i++) {
            u4 = memcmp(&s_dp[(size_t)i * 6], &s_dp[0], 4 * sizeof(float)) == 0;
            u6 = u6 && memcmp(&s_dp[(size_t)i * 6 + 4], &s_dp[4], 2 * sizeof(float)) == 0 && memcmp(&s_phase[(size_t)i * 2 + 1], &s_phase[1], sizeof(float)) == 0;
        }
        if (u4) {
            dev.uni_dp = u6 ? 2u : 1u;
            for (int k = 0; k < 6; k++) dev.uni_dpv[k] = s_dp[k];
            dev.uni_max_stretch = s_phase[1];
            if (u6)
                for (uint32_t i = 0; i < n; i++) {
                    float *q1 = quad(P::DP1, i);
                    const float *q2 = quad(P::DP2, i);
                    q1[0] = q1[2]; q1[1] = q1[3]; q1[2] = q2[0]; q1[3] = q2[1];   // (st0, st1, st2, phase)
                }
        }
    }
    // one material for all particles (bitwise)? -> uniform-material mode (layout.h). Sharded data: the caller says so
    // (wgs_set_uniform_material), a rank cannot know the other ranks' particles.
    bool uniform = D == 3 && n > 0 && !sharded && !(dev.dbg & 65536u);
    for (uint32_t i = 1; i < n && uniform; i++)
        uniform = memcmp(&particles[i].dynamics.mass, &particles[0].dynamics.mass, 4) == 0 &&
                  memcmp(&particles[i].dynamics.init_volume, &particles[0].dynamics.init_volume, 4) == 0 &&
                  memcmp(&particles[i].model, &particles[0].model, sizeof(wgs_elastic_coefficients)) == 0;
#define H2D(dst, src, bytes)                                                               \
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->stream) != hipSuccess)   \
        return bail(fail(WGS_ERR_HIP, "hipMemcpy H2D failed"));
    H2D(dev.buf[0], soa.data(), plane_floats * sizeof(float));
    H2D(d->static_radius, s_radius.data(), s_radius.size() * sizeof(float));
    H2D(d->static_dp, s_dp.data(), s_dp.size() * sizeof(float));
    H2D(d->static_phase, s_phase.data(), s_phase.size() * sizeof(float));
    H2D(d->static_flags, s_flags.data(), s_flags.size() * sizeof(uint32_t));
    d->host_sp = SimParamsDev{};
    for (int k = 0; k < D; k++) d->host_sp.gravity[k] = params->gravity[k];
    d->host_sp.dt = params->dt;
    H2D(d->sp, &d->host_sp, sizeof(SimParamsDev));
    if (sharded) {
        // counts live on the device; the host-side n / nv become the launch bound (allocated capacity)
        uint32_t cnt[2] = {n, n};
        H2D(dev.counters + CTR_N, cnt, sizeof(cnt));
        H2D(dev.counters + CTR_N + CTR_SET, cnt, sizeof(cnt));   // (both sets: layout.h ctr_cur / ctr_next)
        dev.n = dev.nv = (uint32_t)particle_capacity;
        d->nv_hint = n;
    }
    d->host_colliders.resize(WGS_MAX_COLLIDERS);
    memset(d->host_colliders.data(), 0, sizeof(ColliderDev) * WGS_MAX_COLLIDERS);
    for (size_t i = 0; i < num_colliders; i++) fill_collider(d->host_colliders[i], colliders[i]);
    H2D(d->colliders, d->host_colliders.data(), sizeof(ColliderDev) * WGS_MAX_COLLIDERS);
    d->host_bodies.assign(WGS_MAX_COLLIDERS, BodyDev{});
    d->bodies_move = false;
    for (size_t i = 0; i < num_colliders; i++)
        for (int k = 0; k < 3; k++)
            if (colliders[i].velocity.linear[k] != 0.f || colliders[i].velocity.angular[k] != 0.f) {
                d->bodies_move = true;
                d->moving_mask |= 1u << i;
            }
    if (num_colliders)  // local centres of mass from the world ones (update_world_mass_properties' inverse)
        hipLaunchKernelGGL(k_bodies_refresh<D>, dim3(1), dim3(16), 0, d->stream, dev, 0xffffu);
    if (d->bodies_move && enable_impulses(d) != WGS_OK) return bail(fail(WGS_ERR_HIP, "out of device memory for the impulse accumulators"));
#undef H2D
    if (uniform) {
        dev.uniform = 1u;
        dev.uni_mass = particles[0].dynamics.mass;
        dev.uni_vol = particles[0].dynamics.init_volume;
        dev.uni_lambda = particles[0].model.lambda;
        dev.uni_mu = particles[0].model.mu;
        hipLaunchKernelGGL(k_to_uniform, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, dev, 0, 0);
    }
    if (dev.uni_dp != 0u) {   // (the other ping-pong buffer's copy of the quads the step leaves alone: layout.h Dev::uni_dp)
        for (int qd : {(int)Pl<D>::DP0, (int)Pl<D>::DP2}) {
            const size_t plane = (size_t)qd * dev.npad * 4;   // (floats: quad q of slot i sits at float (q * npad + i) * 4)
            if (hipMemcpyAsync(dev.buf[1] + plane, dev.buf[0] + plane, (size_t)dev.npad * 16, hipMemcpyDeviceToDevice, d->stream) != hipSuccess)
                return bail(fail(WGS_ERR_HIP, "hipMemcpy D2D failed"));
        }
    }
    if (hipStreamSynchronize(d->stream) != hipSuccess) return bail(fail(WGS_ERR_HIP, "initial upload failed"));
    *out = d;
    return WGS_OK;
}

wgs_status wgs_data_create(wgs_pipeline *pipeline, const wgs_sim_params *params, const wgs_particle *particles,
                           size_t num_particles, const wgs_collider *colliders, size_t num_colliders, float cell_width,
                           uint32_t grid_capacity, wgs_data **out) {
    return create_impl(pipeline, params, particles, num_particles, nullptr, colliders, num_colliders, cell_width,
                       grid_capacity, num_particles, false, 0, 0, 0, out);
}

wgs_status wgs_data_create_sharded(wgs_pipeline *pipeline, const wgs_sim_params *params, const wgs_particle *particles,
                                   size_t num_particles, const uint32_t *global_ids, const wgs_collider *colliders,
                                   size_t num_colliders, float cell_width, uint32_t grid_capacity,
                                   uint32_t particle_capacity, int32_t block_lo, int32_t block_hi, int32_t force_plastic,
                                   wgs_data **out) {
    if (block_lo >= block_hi) return fail(WGS_ERR_INVALID_ARGUMENT, "empty shard range");
    return create_impl(pipeline, params, particles, num_particles, global_ids, colliders, num_colliders, cell_width,
                       grid_capacity, particle_capacity, true, block_lo, block_hi, force_plastic, out);
}

uint32_t wgs_shard_halo_record_bytes(void) { return (uint32_t)(HaloCfg<D>::REC_F4 * sizeof(float4)); }
uint32_t wgs_shard_particle_record_bytes(void) { return (uint32_t)(particle_record_floats<D>() * sizeof(float)); }
uint32_t wgs_shard_buffer_header_bytes(void) { return 16u; }

wgs_status wgs_set_stream(wgs_data *d, void *hip_stream) {
    if (!d) return fail(WGS_ERR_INVALID_ARGUMENT, "data is NULL");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (d->owns_stream && d->stream) HIP_TRY(hipStreamDestroy(d->stream));
    d->stream = static_cast<hipStream_t>(hip_stream);
    d->owns_stream = false;
    return WGS_OK;
}

wgs_status wgs_shard_export(wgs_data *d, void *device_buf, uint32_t capacity_records, uint32_t *count) {
    if (!d || !device_buf || !count) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!d->dev.sharded) return fail(WGS_ERR_INVALID_ARGUMENT, "not a sharded wgs_data");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    d->dev.ctr_set = (uint32_t)(d->substeps & 1u);
    if (d->needs_compact) {
        hipLaunchKernelGGL(k_shard_compacted, dim3(1), dim3(64), 0, d->stream, d->dev);
        d->needs_compact = false;
    }
    hipLaunchKernelGGL(k_clear_headers, dim3(1), dim3(64), 0, d->stream, static_cast<uint32_t *>(device_buf), (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_export_records<D>, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, d->dev, d->side, static_cast<float *>(device_buf), capacity_records);
    HIP_TRY(hipMemcpyAsync(count, device_buf, sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (*count > capacity_records) return fail(WGS_ERR_INVALID_ARGUMENT, "export buffer too small");
    return WGS_OK;
}

void wgs_data_destroy(wgs_data *d) {
    if (!d) return;
    if (d->stream) hipStreamSynchronize(d->stream);
    if (d->stream2) {
        hipStreamSynchronize(d->stream2);
        hipStreamDestroy(d->stream2);
        if (d->ev_sorted) hipEventDestroy(d->ev_sorted);
        if (d->ev_exchanged) hipEventDestroy(d->ev_exchanged);
    }
    if (d->events.created)
        for (int s = 0; s < Events::MAX_SUBSTEPS; s++)
            for (int m = 0; m < Events::MARKS; m++) hipEventDestroy(d->events.ev[s][m]);
    for (void *p : d->allocs) hipFree(p);
    if (d->stream && d->owns_stream) hipStreamDestroy(d->stream);
    if (d->watch) hipHostFree(d->watch);
    if (d->watch_event) hipEventDestroy(d->watch_event);
    delete d->link;
    delete d;
}

wgs_status wgs_set_uniform_material(wgs_data *d, float mass, float init_volume, float lambda, float mu) {
    if (!d) return fail(WGS_ERR_INVALID_ARGUMENT, "data is NULL");
    if (D != 3) return WGS_OK;  // the 2D layout has no separate constants quad: nothing to gain
    if (d->substeps != 0) return fail(WGS_ERR_INVALID_ARGUMENT, "wgs_set_uniform_material: call before the first step");
    if (d->dev.uniform) return WGS_OK;
    HIP_TRY(hipSetDevice(d->pipeline->device));
    d->dev.uniform = 1u;
    d->dev.uni_mass = mass;
    d->dev.uni_vol = init_volume;
    d->dev.uni_lambda = lambda;
    d->dev.uni_mu = mu;
    if (d->dev.n) hipLaunchKernelGGL(k_to_uniform, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, d->dev, d->side, 1);
    HIP_TRY(hipGetLastError());
    return WGS_OK;
}

wgs_status wgs_set_grid_growth(wgs_data *d, int32_t enabled) {
    if (!d) return fail(WGS_ERR_INVALID_ARGUMENT, "data is NULL");
    d->auto_grow = enabled != 0;
    return WGS_OK;
}

wgs_status wgs_set_constitutive_model(wgs_data *d, int32_t model) {
    if (!d) return fail(WGS_ERR_INVALID_ARGUMENT, "data is NULL");
    if (model != WGS_MODEL_COROTATED && model != WGS_MODEL_NEO_HOOKEAN) return fail(WGS_ERR_INVALID_ARGUMENT, "unknown model");
    d->dev.model = model;
    return WGS_OK;
}

wgs_status wgs_step(wgs_pipeline *pipeline, wgs_data *d, uint32_t num_substeps, int32_t timestamps) {
    if (!pipeline || !d) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(pipeline->device));
    if (timestamps) {
        if (!d->events.created) {
            for (int s = 0; s < Events::MAX_SUBSTEPS; s++)
                for (int m = 0; m < Events::MARKS; m++) HIP_TRY(hipEventCreateWithFlags(&d->events.ev[s][m], hipEventDisableSystemFence));  // timing only: no cache writeback per mark
            d->events.created = true;
        }
        d->events.used = 0;
    }
    {
        wgs_status mst = maintain_grid(d);
        if (mst != WGS_OK) return mst;
    }
    auto flush_bodies = [&]() {   // the last substep's integrate_bodies: every other entry point finds the bodies integrated
        if (d->bodies_pending) {
            hipLaunchKernelGGL(k_bodies_integrate<D>, dim3(1), dim3(16), 0, d->stream, d->dev);
            d->bodies_pending = false;
        }
    };
    for (uint32_t i = 0; i < num_substeps; i++) {
        wgs_status st = WGS_OK;
        if (i > 0 && i % 64u == 0u) {  // long calls: keep an eye on the table inside the call too (bounded run-ahead)
            if ((st = watch_counters(d)) == WGS_OK) st = maintain_grid(d);
        }
        if (st == WGS_OK) {
            if (timestamps && d->events.used < Events::MAX_SUBSTEPS) {
                st = enqueue_substep<true>(d, d->events.used, 0);
                d->events.used++;
            } else {
                st = enqueue_substep<false>(d, 0, 0);
            }
        }
        if (st != WGS_OK) {   // (the substeps enqueued so far stand: a pose read-back after a failed call sees their bodies integrated)
            flush_bodies();
            return st;
        }
    }
    flush_bodies();
    if (timestamps) d->timings_pending = true;
    return watch_counters(d);
}

wgs_status wgs_sync(wgs_data *d) {
    if (!d) return fail(WGS_ERR_INVALID_ARGUMENT, "data is NULL");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    wgs_status st = fetch_counters(d);
    if (st != WGS_OK) return st;
    return sticky_status(d);
}

wgs_status wgs_set_sim_params(wgs_data *d, const wgs_sim_params *params) {
    if (!d || !params) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    for (int k = 0; k < D; k++) d->host_sp.gravity[k] = params->gravity[k];
    d->host_sp.dt = params->dt;
    // pageable memcpyAsync returns after staging, so host_sp may be reused at once
    HIP_TRY(hipMemcpyAsync(d->sp, &d->host_sp, sizeof(SimParamsDev), hipMemcpyHostToDevice, d->stream));
    return WGS_OK;
}

// The setters write single fields of the device-side ColliderDev records (strided copies): poses and
// velocities are integrated on the device, so a whole-record upload would roll them back.
namespace {
wgs_status upload_collider_field(wgs_data *d, size_t field_offset, size_t field_bytes, size_t n) {
    if (n == 0) return WGS_OK;
    HIP_TRY(hipMemcpy2DAsync(reinterpret_cast<char *>(d->colliders) + field_offset, sizeof(ColliderDev),
                             reinterpret_cast<const char *>(d->host_colliders.data()) + field_offset, sizeof(ColliderDev),
                             field_bytes, n, hipMemcpyHostToDevice, d->stream));
    return WGS_OK;
}
}  // namespace

wgs_status wgs_set_collider_poses(wgs_data *d, const wgs_pose *poses, const float *coms, size_t n) {
    if (!d || (!poses && n)) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > d->dev.n_colliders) return fail(WGS_ERR_INVALID_ARGUMENT, "more poses than colliders");
    for (size_t i = 0; i < n; i++) {
        ColliderDev &c = d->host_colliders[i];
        for (int k = 0; k < 4; k++) c.rot[k] = poses[i].rotation[k];
        for (int k = 0; k < 3; k++) c.trans[k] = poses[i].translation[k];
        c.scale = poses[i].scale;
        if (coms) for (int k = 0; k < 3; k++) c.com[k] = coms[i * 3 + k];
    }
    d->cdf_generation++;   // cached node cdfs / block classes are those of the old poses
    static_assert(offsetof(ColliderDev, scale) + sizeof(float) - offsetof(ColliderDev, rot) == 32, "rot|trans|scale contiguous");
    wgs_status st = upload_collider_field(d, offsetof(ColliderDev, rot), 32, n);
    if (st != WGS_OK) return st;
    if (coms && (st = upload_collider_field(d, offsetof(ColliderDev, com), sizeof(float) * 3, n)) != WGS_OK) return st;
    // update_world_mass_properties (rigid_impulses.wgsl:138-149) for the new poses; with explicit world
    // centres of mass the local ones are re-derived instead
    if (n) hipLaunchKernelGGL(k_bodies_refresh<D>, dim3(1), dim3(16), 0, d->stream, d->dev, coms ? (uint32_t)((1u << n) - 1u) : 0u);
    HIP_TRY(hipGetLastError());
    return WGS_OK;
}

wgs_status wgs_set_body_velocities(wgs_data *d, const wgs_velocity *vels, size_t n) {
    if (!d || (!vels && n)) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > d->dev.n_colliders) return fail(WGS_ERR_INVALID_ARGUMENT, "more velocities than colliders");
    const uint32_t moving_before = d->moving_mask;
    for (size_t i = 0; i < n; i++) {
        ColliderDev &c = d->host_colliders[i];
        for (int k = 0; k < 3; k++) c.linvel[k] = vels[i].linear[k];
        for (int k = 0; k < 3; k++) c.angvel[k] = vels[i].angular[k];
        for (int k = 0; k < 3; k++)
            if (c.linvel[k] != 0.f || c.angvel[k] != 0.f) {
                d->bodies_move = true;
                d->moving_mask |= 1u << i;
            }
    }
    if (d->moving_mask != moving_before) d->cdf_generation++;   // (what keeps of a block's node cdfs depends on which colliders move)
    static_assert(offsetof(ColliderDev, angvel) - offsetof(ColliderDev, linvel) == 12, "linvel|angvel contiguous");
    if (d->bodies_move) {
        wgs_status st = enable_impulses(d);
        if (st != WGS_OK) return st;
    }
    return upload_collider_field(d, offsetof(ColliderDev, linvel), sizeof(float) * 6, n);
}


wgs_status wgs_set_body_mass_properties(wgs_data *d, const wgs_mass_properties *mp, size_t n) {
    if (!d || (!mp && n)) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > d->dev.n_colliders) return fail(WGS_ERR_INVALID_ARGUMENT, "more mass properties than colliders");
    bool dynamic = false;
    const uint32_t moving_before = d->moving_mask;
    for (size_t i = 0; i < n; i++) {
        BodyDev &b = d->host_bodies[i];
        for (int k = 0; k < 3; k++) b.inv_mass[k] = mp[i].inv_mass[k];
        for (int k = 0; k < 9; k++) b.inv_inertia_local[k] = mp[i].inv_inertia_local[k];
    }
    for (size_t i = 0; i < d->dev.n_colliders; i++) {
        const BodyDev &b = d->host_bodies[i];
        bool dyn = false;
        for (int k = 0; k < 3; k++) dyn = dyn || b.inv_mass[k] != 0.f;
        for (int k = 0; k < 9; k++) dyn = dyn || b.inv_inertia_local[k] != 0.f;
        if (dyn) d->moving_mask |= 1u << i;
        dynamic = dynamic || dyn;
    }
    if (d->moving_mask != moving_before) d->cdf_generation++;   // (what keeps of a block's node cdfs depends on which colliders move)
    d->bodies_move = d->bodies_move || dynamic;
    if (d->bodies_move) {
        wgs_status st = enable_impulses(d);
        if (st != WGS_OK) return st;
    }
    // inv_mass | inv_inertia_local are the first 12 floats of BodyDev; local_com / world inertia stay device-owned
    static_assert(offsetof(BodyDev, local_com) == sizeof(float) * 12, "BodyDev layout");
    if (n)
        HIP_TRY(hipMemcpy2DAsync(d->dev.bodies, sizeof(BodyDev), d->host_bodies.data(), sizeof(BodyDev), sizeof(float) * 12, n,
                                 hipMemcpyHostToDevice, d->stream));
    if (n) hipLaunchKernelGGL(k_bodies_refresh<D>, dim3(1), dim3(16), 0, d->stream, d->dev, 0u);
    HIP_TRY(hipGetLastError());
    return WGS_OK;
}

wgs_status wgs_set_rigid_particles(wgs_data *d, const float *local_points, const wgs_sample_ids *ids, size_t n,
                                   const float *local_vertices, const uint32_t *vertex_collider_ids, size_t nv) {
    if (!d) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n && (!local_points || !ids || !local_vertices || !vertex_collider_ids || !nv))
        return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    // (sharded data: every rank holds every sample — the node cdfs are a function of position and colliders, both ranks of a
    // face compute the same values for the nodes they share, nothing about them is exchanged)
    if (n > 0xffffffffull || nv > 0xffffffffull) return fail(WGS_ERR_INVALID_ARGUMENT, "too many samples");
    for (size_t i = 0; i < n; i++) {
        if (ids[i].collider >= d->dev.n_colliders) return fail(WGS_ERR_INVALID_ARGUMENT, "sample of an unknown collider");
        for (int k = 0; k < D; k++)
            if (ids[i].vertex[k] >= nv) return fail(WGS_ERR_INVALID_ARGUMENT, "sample refers to a vertex out of range");
    }
    for (size_t i = 0; i < nv; i++)
        if (vertex_collider_ids[i] >= d->dev.n_colliders) return fail(WGS_ERR_INVALID_ARGUMENT, "vertex of an unknown collider");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    HIP_TRY(hipStreamSynchronize(d->stream));
    Dev &dev = d->dev;
    dev.n_rigid = 0;
    // buffers of an earlier call are released (the mesh accumulators, sized by the grid capacity, are kept)
    float *old_f[4] = {dev.rp_local, dev.rp_world, dev.rv_local, dev.rv_world};
    void *old[7] = {old_f[0], old_f[1], old_f[2], old_f[3], dev.rp_ids, dev.rv_collider, dev.rp_needs};
    for (void *p : old) {
        if (!p) continue;
        for (size_t i = 0; i < d->allocs.size(); i++)
            if (d->allocs[i] == p) {
                d->device_bytes -= d->alloc_bytes[i];
                d->allocs.erase(d->allocs.begin() + (long)i);
                d->alloc_bytes.erase(d->alloc_bytes.begin() + (long)i);
                break;
            }
        hipFree(p);
    }
    dev.rp_local = dev.rp_world = dev.rv_local = dev.rv_world = nullptr;
    dev.rp_ids = nullptr;
    dev.rv_collider = dev.rp_needs = nullptr;
    if (n == 0) return WGS_OK;
    wgs_status st;
#define RP_ALLOC(ptr, count) \
    if ((st = dev_alloc(d, ptr, (size_t)(count))) != WGS_OK) return st
    RP_ALLOC(&dev.rp_local, n * D);
    RP_ALLOC(&dev.rp_world, n * D);
    RP_ALLOC(&dev.rp_ids, n);
    RP_ALLOC(&dev.rv_local, nv * D);
    RP_ALLOC(&dev.rv_world, nv * D);
    RP_ALLOC(&dev.rv_collider, nv);
    RP_ALLOC(&dev.rp_needs, n);
    if (!dev.mesh_min) {
        RP_ALLOC(&dev.mesh_min, (size_t)dev.cap * NPB);
        RP_ALLOC(&dev.mesh_aff, (size_t)dev.cap * NPB);
    }
#undef RP_ALLOC
    std::vector<uint4> packed(n);
    for (size_t i = 0; i < n; i++)
        packed[i] = make_uint4(ids[i].vertex[0], ids[i].vertex[1], D == 3 ? ids[i].vertex[2] : 0u, ids[i].collider);
    HIP_TRY(hipMemcpyAsync(dev.rp_local, local_points, sizeof(float) * n * D, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(dev.rp_ids, packed.data(), sizeof(uint4) * n, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(dev.rv_local, local_vertices, sizeof(float) * nv * D, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipMemcpyAsync(dev.rv_collider, vertex_collider_ids, sizeof(uint32_t) * nv, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    dev.n_rigid = (uint32_t)n;
    dev.n_rvtx = (uint32_t)nv;
    return WGS_OK;
}

wgs_status wgs_read_body_poses(wgs_data *d, wgs_pose *poses, wgs_velocity *vels, float *coms, size_t n) {
    if (!d || (!poses && n)) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > d->dev.n_colliders) return fail(WGS_ERR_INVALID_ARGUMENT, "more poses than colliders");
    std::vector<ColliderDev> tmp(WGS_MAX_COLLIDERS);
    HIP_TRY(hipMemcpyAsync(tmp.data(), d->colliders, sizeof(ColliderDev) * WGS_MAX_COLLIDERS, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    for (size_t i = 0; i < n; i++) {
        const ColliderDev &c = tmp[i];
        for (int k = 0; k < 4; k++) poses[i].rotation[k] = c.rot[k];
        for (int k = 0; k < 3; k++) poses[i].translation[k] = c.trans[k];
        poses[i].scale = c.scale;
        if (vels) {
            for (int k = 0; k < 3; k++) vels[i].linear[k] = c.linvel[k];
            for (int k = 0; k < 3; k++) vels[i].angular[k] = c.angvel[k];
        }
        if (coms) for (int k = 0; k < 3; k++) coms[i * 3 + k] = c.com[k];
    }
    return WGS_OK;
}

wgs_status wgs_read_positions(wgs_data *d, float *out) {
    if (!d || !out) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->dev.sharded) return fail(WGS_ERR_UNSUPPORTED, "sharded wgs_data: use wgs_shard_export");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    if (d->dev.n == 0) return WGS_OK;
    float *tmp = nullptr;
    const size_t bytes = sizeof(float) * D * (size_t)d->dev.n;
    HIP_TRY(hipMalloc((void **)&tmp, bytes));
    hipLaunchKernelGGL(k_export_positions, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, d->dev, d->side, tmp);
    hipError_t e = hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(WGS_ERR_HIP, hipGetErrorString(e));
    return WGS_OK;
}

wgs_status wgs_get_device_ptrs(wgs_data *d, wgs_device_ptrs *out) {
    if (!d || !out) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->dev.sharded) return fail(WGS_ERR_UNSUPPORTED, "sharded wgs_data: use wgs_shard_export");
    const float *buf = d->dev.buf[d->side];
    out->position_quads = buf + (size_t)Pl<D>::XM * 4 * d->dev.npad;
    out->particle_ids = reinterpret_cast<const uint32_t *>(buf) + (size_t)Pl<D>::NQ * 4 * d->dev.npad;  // (layout.h ldpid)
    out->count = d->dev.n;
    out->capacity = d->dev.npad;
    out->dim = D;
    out->reserved = 0;
    out->hip_stream = d->stream;
    return WGS_OK;
}

wgs_status wgs_read_particles(wgs_data *d, wgs_particle *out, wgs_plastic_state *plastic_out) {
    if (!d || !out) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->dev.sharded) return fail(WGS_ERR_UNSUPPORTED, "sharded wgs_data: use wgs_shard_export");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    const uint32_t n = d->dev.n;
    if (n == 0) return WGS_OK;
    static_assert(sizeof(wgs_particle) % 4 == 0, "wgs_particle must be word-sized");
    ParticleOffsets o;
#define OFF(f) (uint32_t)(offsetof(wgs_particle, f) / 4)
    o.stride = sizeof(wgs_particle) / 4;
    o.pos = OFF(position); o.vel = OFF(dynamics.velocity); o.F = OFF(dynamics.def_grad); o.C = OFF(dynamics.affine);
    o.nrm = OFF(dynamics.cdf.normal); o.rvel = OFF(dynamics.cdf.rigid_vel); o.dist = OFF(dynamics.cdf.signed_distance);
    o.aff = OFF(dynamics.cdf.affinity); o.vol = OFF(dynamics.init_volume); o.rad = OFF(dynamics.init_radius);
    o.mass = OFF(dynamics.mass); o.lam = OFF(model.lambda); o.mu = OFF(model.mu); o.has_pl = OFF(has_plasticity);
    o.dp = OFF(plasticity); o.has_ph = OFF(has_phase); o.phase = OFF(phase);
#undef OFF
    float *tmp = nullptr, *ptmp = nullptr;
    const size_t bytes = sizeof(wgs_particle) * (size_t)n;
    HIP_TRY(hipMalloc((void **)&tmp, bytes));
    if (plastic_out) {
        hipError_t e = hipMalloc((void **)&ptmp, sizeof(float) * 3 * (size_t)n);
        if (e != hipSuccess) { hipFree(tmp); return fail(WGS_ERR_HIP, hipGetErrorString(e)); }
    }
    // After a step with zero colliders every particle cdf is default_cdf()
    // (g2p_cdf.wgsl:246-249 runs unconditionally); before any step the input is echoed.
    const bool cdf_live = d->cpic || d->substeps == 0;
    hipLaunchKernelGGL(k_export_particles, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, d->dev, d->side, o, d->plastic,
                       cdf_live, (uint32_t)d->substeps, d->static_radius, d->static_dp, d->static_phase, d->static_flags, tmp, ptmp);
    hipError_t e = hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess && plastic_out)
        e = hipMemcpyAsync(plastic_out, ptmp, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    hipFree(tmp);
    if (ptmp) hipFree(ptmp);
    if (e != hipSuccess) return fail(WGS_ERR_HIP, hipGetErrorString(e));
    return WGS_OK;
}

wgs_status wgs_prep_vertex_buffer_device(wgs_data *d, uint32_t mode, wgs_instance *device_instances) {
    if (!d || !device_instances) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (mode > WGS_RENDER_CDF_SIGNS) return fail(WGS_ERR_INVALID_ARGUMENT, "unknown render mode");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    const bool cdf_live = d->cpic || d->substeps == 0;
    if (d->dev.n)
        hipLaunchKernelGGL(k_prep_instances, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, d->dev, d->side, mode, cdf_live,
                           (uint32_t)d->substeps, reinterpret_cast<float *>(device_instances));
    HIP_TRY(hipGetLastError());
    return WGS_OK;
}

wgs_status wgs_prep_vertex_buffer(wgs_data *d, uint32_t mode, wgs_instance *instances) {
    if (!d || !instances) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (mode > WGS_RENDER_CDF_SIGNS) return fail(WGS_ERR_INVALID_ARGUMENT, "unknown render mode");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    const size_t bytes = sizeof(wgs_instance) * (size_t)d->dev.n;
    if (bytes == 0) return WGS_OK;
    wgs_instance *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, bytes));
    hipError_t e = hipMemcpyAsync(tmp, instances, bytes, hipMemcpyHostToDevice, d->stream);  // base colours
    wgs_status st = e == hipSuccess ? wgs_prep_vertex_buffer_device(d, mode, tmp) : WGS_ERR_HIP;
    if (st == WGS_OK) {
        e = hipMemcpyAsync(instances, tmp, bytes, hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    }
    hipFree(tmp);
    if (st != WGS_OK && e == hipSuccess) return st;
    if (e != hipSuccess) return fail(WGS_ERR_HIP, hipGetErrorString(e));
    return WGS_OK;
}

wgs_status wgs_set_plastic_state(wgs_data *d, const wgs_plastic_state *states) {
    if (!d || !states) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->dev.sharded) return fail(WGS_ERR_UNSUPPORTED, "plastic-state restore addresses particles by local index: single-domain data only");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    const size_t n = d->dev.n;
    if (n == 0 || !d->plastic) return WGS_OK;  // no particle carries plasticity: nothing reads the state
    float *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, sizeof(float) * 3 * n));
    hipError_t e = hipMemcpyAsync(tmp, states, sizeof(float) * 3 * n, hipMemcpyHostToDevice, d->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_import_plastic_state, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, d->dev, d->side, tmp);
        e = hipStreamSynchronize(d->stream);
    }
    hipFree(tmp);
    if (e != hipSuccess) return fail(WGS_ERR_HIP, hipGetErrorString(e));
    return WGS_OK;
}

wgs_status wgs_read_grid(wgs_data *d, wgs_node_record *out, size_t capacity, size_t *count) {
    if (!d || !count) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    wgs_status st = fetch_counters(d);
    if (st != WGS_OK) return st;
    const size_t total = (size_t)d->last_nblocks * NPB;
    *count = total;
    if (!out || total == 0) return WGS_OK;
    if (capacity < total) return fail(WGS_ERR_INVALID_ARGUMENT, "capacity too small; *count holds the required size");
    wgs_node_record *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, sizeof(wgs_node_record) * total));
    hipLaunchKernelGGL(k_export_grid, dim3(grid_for(d, 4)), dim3(256), 0, d->stream, d->dev, d->last_nblocks, d->cpic, tmp);
    hipError_t e = hipMemcpyAsync(out, tmp, sizeof(wgs_node_record) * total, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(WGS_ERR_HIP, hipGetErrorString(e));
    return WGS_OK;
}

wgs_status wgs_read_blocks(wgs_data *d, wgs_block_record *out, size_t capacity, size_t *count, uint32_t *sorted_ids) {
    if (!d || !count) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    wgs_status st = fetch_counters(d);
    if (st != WGS_OK) return st;
    const size_t total = d->last_nblocks;
    *count = total;
    if (out && total) {
        if (capacity < total) return fail(WGS_ERR_INVALID_ARGUMENT, "capacity too small; *count holds the required size");
        wgs_block_record *tmp = nullptr;
        HIP_TRY(hipMalloc((void **)&tmp, sizeof(wgs_block_record) * total));
        hipLaunchKernelGGL(k_export_blocks, dim3(grid_for(d, 1)), dim3(256), 0, d->stream, d->dev, d->last_nblocks, tmp);
        hipError_t e = hipMemcpyAsync(out, tmp, sizeof(wgs_block_record) * total, hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        hipFree(tmp);
        if (e != hipSuccess) return fail(WGS_ERR_HIP, hipGetErrorString(e));
    }
    if (sorted_ids && d->dev.n) {
        // The buffer written by the last substep is in sorted order: its pid plane IS sorted_ids.
        const float *pidp = d->dev.buf[d->side] + (size_t)P::NQ * 4 * d->dev.npad;
        HIP_TRY(hipMemcpyAsync(sorted_ids, pidp, sizeof(uint32_t) * (size_t)d->dev.n, hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
    }
    return WGS_OK;
}

// Test hook: the exclusive scan of launch 2 (kernels_sort.h scan_chunk: what replaces prefix_sum.wgsl) on caller
// data — values[i] plays the particle count of block i, every block active. The reference's own scan test vectors
// (src/grid/prefix_sum.rs:183-229) go through the HIP scan this way.
wgs_status wgs_debug_scan(wgs_pipeline *pipeline, const uint32_t *values, uint32_t n, uint32_t *out, uint32_t *total) {
    if (!pipeline || (!values && n) || (!out && n)) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > (1u << 25)) return fail(WGS_ERR_INVALID_ARGUMENT, "n out of range");
    HIP_TRY(hipSetDevice(pipeline->device));
    Dev dev{};
    dev.cap = std::max(1u, n);
    const uint32_t nscan = (dev.cap + SCAN_CHUNK - 1) / SCAN_CHUNK, epoch = 1u;
    std::vector<void *> tmp;
    auto alloc = [&](size_t bytes, int fill) -> void * {
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
        hipMemset(p, fill, bytes ? bytes : 4);
        tmp.push_back(p);
        return p;
    };
    auto cleanup = [&]() { for (void *p : tmp) hipFree(p); };
    dev.counters = (uint32_t *)alloc(sizeof(uint32_t) * CTR_COUNT, 0);
    dev.block_stamp = (uint32_t *)alloc(sizeof(uint32_t) * dev.cap, 0);
    dev.block_acc = (uint32_t *)alloc(sizeof(uint32_t) * dev.cap, 0);
    dev.active = (uint32_t *)alloc(sizeof(uint32_t) * dev.cap, 0);
    dev.block_start = (uint32_t *)alloc(sizeof(uint32_t) * dev.cap, 0);
    dev.chunk_a = (unsigned long long *)alloc(sizeof(unsigned long long) * nscan, 0);
    dev.chunk_b = (unsigned long long *)alloc(sizeof(unsigned long long) * nscan, 0);
    dev.group_a = (unsigned long long *)alloc(sizeof(unsigned long long) * nscan * SORT_THREADS, 0);
    dev.group_b = (unsigned long long *)alloc(sizeof(unsigned long long) * nscan * SORT_THREADS, 0);
    if (!dev.counters || !dev.block_stamp || !dev.block_acc || !dev.active || !dev.block_start || !dev.chunk_a || !dev.chunk_b || !dev.group_a || !dev.group_b) {
        cleanup();
        return fail(WGS_ERR_HIP, "out of device memory");
    }
    std::vector<uint32_t> ones(dev.cap, epoch);
    uint32_t ctr[CTR_COUNT] = {0};
    ctr[CTR_NPHYS] = n;
    hipError_t e = hipMemcpy(dev.counters, ctr, sizeof(ctr), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dev.block_stamp, ones.data(), sizeof(uint32_t) * dev.cap, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) e = hipMemcpy(dev.block_acc, values, sizeof(uint32_t) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_scan_only, dim3(nscan + std::min((n + 3u) / 4u + 1u, 2048u)), dim3(SORT_THREADS), 0, 0, dev, epoch, nscan);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess && n) e = hipMemcpy(out, dev.block_start, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && total) {
        unsigned long long t = 0;  // sum of the chunk totals (low words)
        std::vector<unsigned long long> ct(nscan);
        e = hipMemcpy(ct.data(), dev.chunk_b, sizeof(unsigned long long) * nscan, hipMemcpyDeviceToHost);
        for (auto v : ct) t += v & 0xffffffffull;
        *total = (uint32_t)t;
    }
    cleanup();
    if (e != hipSuccess) return fail(WGS_ERR_HIP, hipGetErrorString(e));
    return WGS_OK;
}

#ifdef WGS_ABLATE
wgs_status wgs_debug_g2p_prof(unsigned long long *out /* WGS_G2P_ROWS * 8 */) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_g2p_prof), sizeof(unsigned long long) * WGS_G2P_ROWS * 8));
    std::vector<unsigned long long> zero((size_t)WGS_G2P_ROWS * 8, 0ull);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_g2p_prof), zero.data(), sizeof(unsigned long long) * WGS_G2P_ROWS * 8));
    return WGS_OK;
}
wgs_status wgs_debug_p2g_prof(unsigned long long *out /* WGS_P2G_ROWS * 8 */) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2g_prof), sizeof(unsigned long long) * WGS_P2G_ROWS * 8));
    std::vector<unsigned long long> zero((size_t)WGS_P2G_ROWS * 8, 0ull);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_p2g_prof), zero.data(), sizeof(unsigned long long) * WGS_P2G_ROWS * 8));
    return WGS_OK;
}
// stage clocks of launch 2 (kernels_sort.h g_prof): read and reset. Experiment builds only, not in the header.
wgs_status wgs_debug_prof(unsigned long long *out /* WGS_PROF_ROWS * 8 */) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * WGS_PROF_ROWS * 8));
    std::vector<unsigned long long> zero((size_t)WGS_PROF_ROWS * 8, 0ull);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero.data(), sizeof(unsigned long long) * WGS_PROF_ROWS * 8));
    return WGS_OK;
}
#endif

wgs_status wgs_read_timing_overhead(wgs_data *d, float *ms_per_mark) {
    if (!d || !ms_per_mark) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    resolve_timings(d);
    *ms_per_mark = d->mark_overhead_ms;
    return WGS_OK;
}

wgs_status wgs_read_timings(wgs_data *d, float ms[WGS_NUM_PASSES]) {
    if (!d || !ms) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    resolve_timings(d);
    for (int p = 0; p < WGS_NUM_PASSES; p++) ms[p] = d->timings[p];
    return WGS_OK;
}

wgs_status wgs_get_stats(wgs_data *d, wgs_stats *out) {
    if (!d || !out) return fail(WGS_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(d->pipeline->device));
    wgs_status st = fetch_counters(d);
    if (st != WGS_OK) return st;
    out->num_particles = d->dev.n;
    if (d->dev.sharded) {
        HIP_TRY(hipMemcpyAsync(&out->num_particles, d->dev.counters + CTR_NV + CTR_SET * (d->needs_compact ? ((d->substeps & 1) ^ 1) : (d->substeps & 1)),
                               sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
    }
    out->num_active_blocks = d->last_nblocks;
    out->grid_capacity = d->dev.cap;
    out->overflow = d->sticky_errors;
    out->substeps_done = d->substeps;
    out->device_bytes = d->device_bytes;
    out->num_near_collider_blocks = d->cpic && d->last_ncpic != UINT32_MAX ? d->last_ncpic : 0u;
    out->grid_growths = d->grid_grown;
    out->cell_changers = d->movers_total;
    out->table_rebuilds = d->table_rebuilds;
    out->block_ids = d->last_nphys;
    out->block_ids_free = d->last_nfree;
    out->table_marks = d->last_ntomb;
    out->table_refreshes = (uint32_t)d->table_refreshes;
    return WGS_OK;
}

}  // extern "C"

#include "capi_sharded.inc"
