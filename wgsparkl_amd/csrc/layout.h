// layout.h — HBM data layout of one wgs_data.
//
// Particles live in two ping-pong structure-of-arrays buffers of fp32 planes
// (plane p of a buffer starts at base + p * npad). Each substep the fused
// G2P + particle-update kernel reads the current buffer through the sort
// permutation and writes the other buffer in (block, cell, particle-id) order,
// so the next substep's reads are coalesced and `perm` is near-identity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef WGS_DIM
#define WGS_DIM 3
#endif

// Ablation switches (kernels that skip their maths or loads on request) are compiled in only with -DWGS_ABLATE.
#ifdef WGS_ABLATE
#define WGS_ABLATE_AND(cond) &&(cond)
#else
#define WGS_ABLATE_AND(cond)
#endif

namespace wgs {

constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t PID_DEAD = 0xffffffffu;  // sharded runs: slot of a particle that now lives on a neighbouring rank
constexpr int NPB = 64;  // nodes (= cells) per block: 4^3 or 8^2, grid.wgsl:43
constexpr uint32_t BLK_ARR = 64;  // arrivals from other blocks a block records as an array (Dev::blk_arr)

template <int D> struct Dim;
template <> struct Dim<3> {
    static constexpr int BW = 4, BSHIFT = 2, TW = 6, TILE = 216, NBH = 27, DD = 9, NNBR = 8, ANG = 3;
};
template <> struct Dim<2> {
    static constexpr int BW = 8, BSHIFT = 3, TW = 10, TILE = 100, NBH = 9, DD = 4, NNBR = 4, ANG = 1;
};

// Particle buffer = structure of float4 arrays ("quads") + one u32 plane.
// Quad q of a buffer starts at base + q * npad float4; every access is a 16-byte-per-lane
// global_load/store_dwordx4 (coalescing sweet spot, and whole 16-byte sectors are used
// when the read goes through the sort permutation). Quads are grouped by consumer:
//   P2G reads XM + CV*            (64 B/particle in 3D = the algorithmic minimum)
//   fused G2P+update reads XM + F* + pid, writes everything.
template <int D> struct Pl;
template <> struct Pl<3> {
    static constexpr int XM = 0;    // x, y, z, mass   (uniform-material mode: x, y, z, F[8] — see Dev::uniform)
    static constexpr int CV0 = 1;   // C'[0..3]           (APIC matrix, column-major; particle3d.wgsl:12)
    static constexpr int CV1 = 2;   // C'[4..7]
    static constexpr int CV2 = 3;   // C'[8], vx, vy, vz
    static constexpr int F0 = 4;    // F[0..3]
    static constexpr int F1 = 5;    // F[4..7]
    static constexpr int F2 = 6;    // F[8], init_volume, lambda, mu   (uniform-material mode: not touched by the step)
    static constexpr int NBASE = 7;
    static constexpr int DP0 = 7;   // h0, h1, h2, h3                    (models/drucker_prager.wgsl:8-16)
    static constexpr int DP1 = 8;   // dp.lambda, dp.mu, plastic det, plastic hardening
    static constexpr int DP2 = 9;   // log_vol_gain, phase, max_stretch, -
    static constexpr int CDF0 = 10; // normal xyz, signed_distance      (particle3d.wgsl:17-25)
    static constexpr int CDF1 = 11; // rigid_vel xyz, affinity (u32 bits)
    static constexpr int NQ = 12;
};
template <> struct Pl<2> {
    static constexpr int XM = 0;    // x, y, mass, init_volume
    static constexpr int CV0 = 1;   // C'[0..3]
    static constexpr int CV2 = 2;   // vx, vy, lambda, mu
    static constexpr int F0 = 3;    // F[0..3]
    static constexpr int NBASE = 4;
    static constexpr int DP0 = 4;
    static constexpr int DP1 = 5;
    static constexpr int DP2 = 6;
    static constexpr int CDF0 = 7;  // normal xy, signed_distance, affinity
    static constexpr int CDF1 = 8;  // rigid_vel xy, -, -
    static constexpr int NQ = 9;
};
// floats per particle slot of one buffer (NQ quads + the pid plane + the cdf-epoch plane)
template <int D> constexpr size_t buffer_floats(uint32_t npad) { return ((size_t)Pl<D>::NQ * 4 + 2) * npad; }

struct SimParamsDev {   // solver/params.wgsl:3-10 + grid.cell_width; lives in HBM so graph replays see updates
    float gravity[3];
    float dt;
};

struct ColliderDev {    // wgs_collider, device copy
    uint32_t shape_type;
    float shape[4];
    float rot[4];
    float trans[3];
    float scale;
    float linvel[3];
    float angvel[3];
    float com[3];
};

// What velocity_at_point needs of a collider (the three fields are consecutive in ColliderDev): the CPIC body of the fused G2P keeps
// the colliders' copies in LDS (g2p_body.inc).
struct ColliderMotion {
    float linvel[3];
    float angvel[3];
    float com[3];
};

struct BodyDev {        // mass properties of the body behind a collider (rigid_impulses.wgsl:81-84)
    float inv_mass[3];
    float inv_inertia_local[9];   // 3D: column-major, body frame; 2D: [0]
    float local_com[3];
    float inv_inertia_world[9];   // refreshed from the pose every substep
};

struct NodeCdf {        // grid.wgsl:233-240
    float distance;
    uint32_t affinities;
    uint32_t closest_id;
    uint32_t pad;
};

// Counter slots in Dev::counters
// (the list counters sit in cache lines of their own: thousands of waves add to them in launch 2 of a collider-heavy scene while
// every wave of that launch reads the counters of the first line; sharing a line made those reads queue behind the atomics)
enum { CTR_NBLOCKS = 0, CTR_ERRORS = 1, CTR_NPHYS = 2,
       CTR_NSORTED = 3,  // particles in this substep's sorted order (the scan's grand total): all of them, unless a grid overflow or the key range left some out
       CTR_N = 4, CTR_NV = 5,
       CTR_NPREV = 6,  // sharded runs: slots [0, NPREV) are the sorted output of the last substep, [NPREV, N) arrivals
       CTR_NLEAVE = 7,  // sharded runs: particles the last fused G2P launches found outside the core range (Dev::leavers): next substep's guests
       CTR_TICKET = 10,  // sharded runs: workgroups of k_g2p_arrivals that are done (the last one does the bookkeeping)
       CTR_NFREE = 11,   // ids on the free list (Dev::free_ids): blocks launch 2 of the sort evicted from the table
       CTR_NINSERT = 12, // insertions into the table since its last rebuild (never decreases: ids are reused, so the id counter does not say)
       CTR_NTOMB = 13,   // table slots marked KEY_TOMB and not reused yet (the host clears the marks before they crowd the table)
       CTR_NPHYS_SEEN = 8,  // [8], [9]: the insertion counter (CTR_NINSERT) as launch 2 of an even / odd substep saw it (kernels_sort.h regroup_block)
       CTR_NVISIT = 64,  // [64 + 32 k], k = 0..7: length of the visit list of XCD k (Dev::visit_list), one cache line each
       CTR_NCPIC = 320,  // [320 + 32 k], k = 0..7: length of list k of the near-collider blocks (Dev::cpic_list), one cache line each
       CTR_NHALO = 576,  // sharded runs: length of the list of active blocks in the interface layers (Dev::halo_list)
       CTR_MOVERS = 640,  // [640 + 32 k], k = 0..15: particles that changed their associated cell since creation, 16 partial counts in cache lines of
                          // their own (modulo 2^32 each; one add per workgroup of launch 2 of the sort; wgs_get_stats sums them)
       CTR_COUNT = 1152 };
// The three list counters exist TWICE, 16 words apart (same cache line of their own): launch 2 of the sort of substep `epoch`
// appends to set epoch & 1 — which P2G, the fused G2P and the pack waves of that substep read — and its scan workgroup
// zeroes the other set for the next substep. No launch has to run in front of the sort just to reset them (the binning of a
// steady-state substep is done by the fused G2P of the substep before it: g2p_body.inc).
constexpr uint32_t CTR_PARITY = 16;
__host__ __device__ inline uint32_t ctr_ncpic(uint32_t k, uint32_t epoch) { return (uint32_t)CTR_NCPIC + 32u * k + CTR_PARITY * (epoch & 1u); }
__host__ __device__ inline uint32_t ctr_nvisit(uint32_t k, uint32_t epoch) { return (uint32_t)CTR_NVISIT + 32u * k + CTR_PARITY * (epoch & 1u); }
__host__ __device__ inline uint32_t ctr_nhalo(uint32_t epoch) { return (uint32_t)CTR_NHALO + CTR_PARITY * (epoch & 1u); }
enum { ERRBIT_OVERFLOW = 1u, ERRBIT_KEYRANGE = 2u, ERRBIT_SHARD = 4u, ERRBIT_MATERIAL = 8u, ERRBIT_HANDOVER = 16u,
       ERRBIT_PCDF = 32u };   // a CPIC workgroup of P2G gave up waiting for the prologue waves' particle cdf and computed it itself (results intact)
// Bit 31 of a perm_cell entry (block ids stay below 2^24): the particle's block is near a collider. Written by launch 2 of the
// sort when it computes the block classes itself; the fused G2P then knows which body a particle belongs to from the sort
// entry it loads anyway, instead of a dependent block_cpic lookup in the middle of every chunk.
constexpr uint32_t CELL_LISTED = 0x80000000u;
// Bit 30 of an act_info record's flags (round 6): the block's run is its PREVIOUS run, member for member (nobody moved, nobody arrived:
// the sort's fast_clean) — its act_cells entries then hold the cells' runs in the CURRENT BUFFER's coordinates (the previous sorted
// order, which is the storage order), and the plain body of P2G reads the particles there without the gather through `perm`: one
// dependent round trip less per block (stage clocks at the headline: 1.35 of a block's 8 us). The CPIC body walks cell_start /
// cell_cursor (sorted coordinates) and is not concerned.
constexpr uint32_t CELL_DIRECT = 0x40000000u;
// Bit 31 of a Dev::cellid entry: the particle changed cell in the step that wrote the entry (NONE stays NONE). Launch 2 of the sort tells
// the stayers of a run from the particles that came from another cell of the block by it, and counts the cell-changers.
constexpr uint32_t CELL_MOVED = 0x80000000u;
__device__ inline uint32_t cell_of(uint32_t entry) { return entry == NONE ? NONE : (entry & ~CELL_MOVED); }
constexpr uint32_t HALO_ENT = 12;  // words per Dev::halo_list entry: block id, key, 8 source slabs, 2 spare

// Message buffers of a slab (kernels_shard.h): [0] lower, [1] upper neighbour; null = no neighbour on that side.
struct ShardMsg {
    float *out[2];
    const float *in[2];
    uint32_t halo_cap, mig_cap;  // records per message
};

// Everything a kernel needs, passed by value.
struct Dev {
    // Particle counts. Single GPU: fixed, passed by value. Sharded: they change every substep with the
    // migrating particles and live in counters[CTR_N / CTR_NV] so that nothing on the step path needs the
    // host; n / nv then hold the allocated capacity (launch bound). Use num_slots() / num_valid().
    uint32_t n;          // particle slots in the current buffer (valid + vacated)
    uint32_t nv;         // valid particles = sorted slots (== n unless sharded and particles migrated)
    uint32_t sharded;    // 1: x-slab decomposition (kernels_shard.h)
    int shard_lo, shard_hi;  // core block range along x
    uint32_t shard_has_lo, shard_has_hi;  // a neighbour below / above (wgs_shard_attach)
    uint32_t ctr_set;        // which set of the particle counters this substep reads (substep number & 1; below)
    uint32_t skip_guests;    // the fused G2P drops the particles whose block lies outside the core range: their new owner
                             // advances them (set by the sharded step; wgs_step on a slab advances everything it holds)
    ShardMsg msg;            // message buffers of the slab
    uint32_t npad;       // plane stride (floats)
    float *buf[2];       // ping-pong particle buffers
    uint32_t *perm;      // sorted slot -> index in the current buffer
    uint32_t *perm_cell; // sorted slot -> physical block id * 64 + cell in block
    uint32_t *cellid;    // per particle (current-buffer index): NEW dense cell id = physical block id * 64 + cell in block
    uint32_t *mv_next;   // per particle: next mover (slot + 1, 0 = end) on the list of the same destination cell
    // sparse block grid (grid.wgsl:82-184): open-addressing hash of packed block keys
    uint32_t *hkeys;     // hcap: packed key or NONE
    uint32_t *hvals;     // hcap: physical block id of the slot's key (NONE while the insert is in flight)
    uint32_t hmask;      // hcap - 1
    uint32_t cap;        // block capacity
    // per physical block id (ids persist while the block stays in the hash map)
    uint32_t *block_key;   // cap: packed virtual id (NONE: the id is on the free list — the block was evicted)
    uint32_t *block_slot;  // cap: the table slot that holds the block's key (what an eviction marks); null: no eviction on this data (WGS_DEBUG bit 10)
    uint32_t *free_ids;    // cap: ids of evicted blocks, a stack of counters[CTR_NFREE] entries (pushed by launch 2 of the sort, popped by insertions)
    uint32_t *block_stamp; // cap: epoch of the last substep in which the block was active
    uint32_t *links_epoch; // cap: epoch at which nbr_plus / nbr_minus of the block were last written
    uint32_t *block_acc;   // cap: particle counter being accumulated by k_bin / k_rebin (zero at rest)
    uint32_t *block_ident; // cap: (epoch << 1 | near a collider) of the last substep that found the block's part of perm / perm_cell to be the
                           // identity (nobody moved, nobody arrived, the block's run starts where it started): a substep that finds the
                           // same again, right after, leaves the two arrays alone
    uint32_t *blk_narr;    // cap: particles that came into the block from OTHER blocks in this substep (zero outside the sort) ...
    uint32_t *blk_arr;     // cap*BLK_ARR: ... and the slots of the first BLK_ARR of them, in arrival order (arbitrary); the others are on
                           // their cell's list (cell_head). One coalesced load for the wave that regroups the block, no pointer chasing
    uint32_t *block_dirty; // cap: epoch of the last substep for which some particle of the block's previous run is no longer in its cell
                           // (it moved to another cell, left the block or was dropped); a block that is not dirty and has no arrivals keeps its runs
    uint32_t *block_count; // cap: particles whose associated cell is in the block (num_particles)
    uint32_t *block_start; // cap: exclusive scan of block_count over the active list (first_particle)
    uint32_t *active;      // cap: physical ids of the blocks active in this substep, [0, num_active_blocks)
    uint4 *act_info;       // cap, by ACTIVE-LIST index: {block id, key, particles, CELL_LISTED if near a collider | CELL_DIRECT} — what P2G needs of a block in one load
    uint2 *act_cells;      // cap*64, by ACTIVE-LIST index: {start, end} of each cell's run in perm (cell_start / cell_cursor are by block id) — in the current buffer itself where the record says CELL_DIRECT
    uint32_t *nbr_plus;    // cap*8: physical ids of b + {0,1}^D (always active)
    uint32_t *nbr_minus;   // cap*8: physical ids of b - {0,1}^D, NONE when inactive
    uint32_t *act_src;     // cap*8, by ACTIVE-LIST index: the b - {0,1}^D neighbours that hold particles (the slabs a node of b is
                           // gathered from), NONE otherwise — what the grid update reads instead of links + counts
    uint32_t *nbr_known;   // cap*16: the same 16 neighbours' ids if they are in the table at all (active or not), NONE if absent
    uint32_t *cell_head;   // cap*64: head of the cell's list of arrivals that did not fit blk_arr, and of every particle on a table-rebuild
                           // substep (slot + 1; zero outside the sort). A particle that changed cell inside its block is on no list: the wave that regroups the block meets
                           // it in the block's previous run and hands it to its new cell through LDS (kernels_sort.h regroup_block)
    uint32_t *cell_start;  // cap*64
    uint32_t *cell_cursor; // cap*64: end of the cell's range in perm
    unsigned long long *chunk_a, *chunk_b;  // cap / 4096: (epoch << 32 | active blocks), (epoch << 32 | particles) of a scan chunk
    unsigned long long *group_a, *group_b;  // cap / 16: the same two totals before every group of 16 blocks (second level: block_prefix)
    float4 *nodes;         // cap*64: velocity|momentum xyz, mass (2D: vx, vy, mass, 0)
    NodeCdf *node_cdf;     // cap*64
    float4 *slab;          // cap*TILE: per-block tile (block + its "+1" rim): momentum after P2G, velocity after the grid update
    uint32_t *slab_epoch;  // cap: substep (epoch) of the block's last COMPLETE momentum slab — P2G's hand-over to the grid-update waves that ride in
                           // the same launch (kernels_transfer.h): written after the slab's write-through stores have been acknowledged
    uint32_t *block_cdf_gen;  // cap: generation (Dev::cdf_gen) under which block_cpic / node_cdf of the block were last computed
    uint32_t *block_cpic;     // cap: some node of the block's (BW+2)^D tile has non-zero affinity
    uint32_t *block_cdf_summ; // cap: (epoch & 0xffffff) << 8 | bit o: some OWN node of the block with coordinate <= BW+2-BW-1 on every axis set in o has
                              // non-zero affinity — what the tile of the block's "-o" neighbour holds of this block. Published (agent scope) by the
                              // block's wave of launch 2 of the sort as soon as its 64 nodes are evaluated; a neighbour that finds the word of this
                              // substep does not evaluate those nodes again (kernels_sort.h)
    uint32_t *cpic_list;      // 8 x cap: particle-bearing blocks with block_cpic set; list k = (block id & 7) at [k * cap, + counters[CTR_NCPIC + 32 k]):
                              // eight lists because thousands of returning atomics on ONE counter serialise in the fabric
    uint2 *visit_list;        // 8 x visit_cap: (listed block, chunk of 64 sorted particles that holds some of its particles); list k, at
                              // [k * visit_cap, + counters[CTR_NVISIT + 32 k]), is the one the CPIC body of the fused G2P advances on
                              // XCD k (g2p_body.inc); device_math.h append_visits deals the chunks to the lists
    uint32_t *halo_list;      // sharded runs, cap x HALO_ENT words: the active blocks of the layers that travel (what k_pack_face gathers and packs)
    uint32_t *pcdf_done;      // cap: particles of a listed block whose cdf this substep's prologue WAVES of the P2G launch have written (reset by
                              // the sort when it lists the block); the block's CPIC workgroup waits for its particle total (kernels_transfer.h pcdf_waves)
    uint32_t pcdf_waves;      // prologue workgroups at the front of this P2G launch (set by the host per launch; 0: none). Whether they WORK is decided
                              // inside the launch from this substep's list lengths (device_math.h pcdf_waves_on): the same answer in every workgroup
    uint32_t visit_cap;       // per list (an eighth of the chunks + 2 per block would do; a block is visited once per chunk it spans)
    uint32_t cdf_gen;         // node cdfs and block classes computed under this generation stay valid for a block (it keeps its id and
                              // therefore its place) as long as no collider that can MOVE reaches its tile; 0: no colliders
    uint32_t cdf_moving;      // bit i: collider i has (or had) a velocity or a mass: its pose changes from substep to substep
    uint32_t listed_in_perm;  // this substep's perm_cell entries carry CELL_LISTED (launch 2 of the sort computed the block classes)
    uint32_t g2p_npass;       // chunks per wave of the fused G2P of this substep (defines the eighths; set by the host per substep)
    uint32_t bin_next;        // the fused G2P of this substep also BINS its output for the next substep (new cell ids, block activation and
                              // totals, mover lists — launch 1 of the next sort, k_rebin, is then not launched); on a slab k_g2p_arrivals
                              // does the same for the particles that arrive. Not the plastic variants (kernels_transfer.h BIN)
    uint32_t *counters;    // CTR_COUNT
    const SimParamsDev *sp;
    ColliderDev *colliders;  // poses / velocities are integrated on the device (kernels_bodies.h)
    uint32_t n_colliders;
    BodyDev *bodies;         // 16: mass properties
    int32_t *impulses;       // 16 * 8: fixed-point (x 1e5) linear[D] + angular impulses accumulated by P2G
    float4 *imp_slab;        // cap*TILE*IMPQ per-block partial node impulses (two-way coupling only), or null
    uint32_t *leavers;       // sharded: output slots of the particles that left [shard_lo, shard_hi) in the last substep (filled by the
                             // fused G2P and k_g2p_arrivals, consumed by the next substep's k_pack_face: its guests)
    uint32_t leavers_cap;
    // rigid particles of mesh colliders (kernels_rigid.h); n_rigid == 0 when there is none
    uint32_t n_rigid, n_rvtx;
    float *rp_local, *rp_world;          // n_rigid * D: sample points, body frame / world
    uint4 *rp_ids;                       // n_rigid: vertex ids of the sample's triangle / segment, collider id in .w
    float *rv_local, *rv_world;          // n_rvtx * D: mesh vertices
    uint32_t *rv_collider;               // n_rvtx
    uint32_t *rp_needs;                  // n_rigid: sort.wgsl:55-86 flag
    unsigned long long *mesh_min;        // cap*64: (distance bits << 32 | collider id) of the closest mesh primitive, ~0 = none
    uint32_t *mesh_aff;                  // cap*64: affinity / sign bits set by mesh primitives
    float h;             // cell width
    float inv_h;
    uint32_t h_pow2;     // cell width is a power of two: x * inv_h == x / h bit for bit
    // Uniform-material mode (3D): every particle has the same (mass, init_volume, lambda, mu), so the four constants
    // travel as kernel arguments instead of being read and re-written with every particle in every substep (the sort
    // is physical): F[8] takes the place of the mass in XM.w and the F2 quad drops out of the step — the fused kernel
    // moves 160 instead of 192 bytes per particle. Decided at creation (all particles equal) or by the caller
    // (wgs_set_uniform_material); the general layout remains for everything else.
    uint32_t uniform;
    float uni_mass, uni_vol, uni_lambda, uni_mu;
    // (round 6) The same for the plasticity parameters, on single-domain data, decided at creation (bitwise comparison of all particles):
    //   uni_dp = 1: one set of DruckerPrager h0..h3 (the DP0 quad) — the plastic fused G2P takes the four from here and neither reads
    //               nor rewrites the quad: both ping-pong buffers hold it for every slot from the upload on;
    //   uni_dp = 2: lambda, mu of the plasticity and max_stretch are shared as well — the whole per-particle plastic STATE is then one
    //               quad, DP1 = (plastic det, plastic hardening, log_vol_gain, phase), and DP2 drops out of the step too.
    // 32 / 64 of the plastic step's 216 + bytes per particle. Whoever else touches these quads (export, checkpoint restore) goes
    // through unpack_slot + fix_uniform / k_import_plastic_state, which know the mode; slabs of a decomposition never use it.
    uint32_t uni_dp;
    float uni_dpv[6], uni_max_stretch;
    int model;           // WGS_MODEL_*
    uint32_t dbg;        // launch-shape A/B switches (env WGS_DEBUG; same results, see capi.hip), 0 in production. The
                         // result-changing ablations only exist in builds with -DWGS_ABLATE (never the shipped library)
};

// Sharded runs keep TWO sets of the three particle counters (CTR_N, CTR_NV, CTR_NPREV; the second set CTR_SET slots
// further): the kernels of substep n read set n & 1 (Dev::ctr_set), and the bookkeeping of the migration round — done by
// the last workgroup of k_g2p_arrivals (kernels_arrivals.h) — writes the set of substep n + 1.
constexpr uint32_t CTR_SET = 16;
__device__ inline uint32_t &ctr_cur(const Dev &d, uint32_t which) { return d.counters[which + CTR_SET * d.ctr_set]; }
__device__ inline uint32_t &ctr_next(const Dev &d, uint32_t which) { return d.counters[which + CTR_SET * (d.ctr_set ^ 1u)]; }
__device__ inline uint32_t num_slots(const Dev &d) { return d.sharded ? ctr_cur(d, CTR_N) : d.n; }
__device__ inline uint32_t num_valid(const Dev &d) { return d.sharded ? ctr_cur(d, CTR_NV) : d.nv; }
// Chunks of 64 sorted particles per XCD eighth of the fused G2P: waves advance `npass` consecutive chunks, an XCD gets
// `per_xcd` consecutive waves' worth (g2p_body.inc; the sort files the visit list by the same rule).
__device__ inline uint32_t g2p_waves_per_xcd(const Dev &d, uint32_t npass) {
    const uint32_t span = 64u * npass;
    return ((num_valid(d) + span - 1u) / span + 7u) >> 3;
}

// Layout of a block's tile slab (Dev::slab, Dev::imp_slab): NOT in tile order but grouped by the block the nodes belong to —
// region o (o in {0,1}^D, bit k = the node lies in the "+1" neighbour along axis k) holds the nodes of the tile that are
// nodes of block b + o, in that block's own (x fastest) order restricted to the two layers the tile reaches where o_k = 1.
// The grid update gathers and rewrites, for every destination block, exactly one region of each source slab: with this
// layout those are contiguous runs of 128 B .. 1 KB instead of 64-byte rows strided through a (BW+2)^D box — the launch
// moves a tile 3.4x its block, so how it touches it is most of its time. P2G stores and the fused G2P loads whole slabs
// (contiguous either way) and translate positions with slab_tile_of.
template <int D> __host__ __device__ constexpr int slab_region_base(int o) {
    // sizes: prod_k (o_k ? 2 : BW); 3D: 64 32 32 16 32 16 16 8, 2D: 64 16 16 4
    int base = 0;
    for (int r = 0; r < o; r++) {
        int sz = 1;
        for (int k = 0; k < D; k++) sz *= ((r >> k) & 1) ? 2 : Dim<D>::BW;
        base += sz;
    }
    return base;
}
// position in the slab of the node with local coordinates l (in its own block b + o) of region o
template <int D> __device__ inline uint32_t slab_pos(int o, const int *l) {
    constexpr int BW = Dim<D>::BW;
    const int ex = (o & 1) ? 2 : BW, ey = (o & 2) ? 2 : BW;
    int base = 0;
#pragma unroll
    for (int r = 0; r < (1 << D); r++)
        if (r == o) base = slab_region_base<D>(r);
    return (uint32_t)(base + l[0] + ex * (l[1] + (D == 3 ? ey * l[2] : 0)));
}
// tile index (x + TW y (+ TW^2 z)) of slab position p
template <int D> __host__ __device__ constexpr uint32_t slab_tile_of(uint32_t p) {
    constexpr int BW = Dim<D>::BW, TW = Dim<D>::TW;
    int o = 0;
    for (int r = 1; r < (1 << D); r++)
        if (p >= (uint32_t)slab_region_base<D>(r)) o = r;
    const uint32_t q = p - (uint32_t)slab_region_base<D>(o);
    const uint32_t sx = (o & 1) ? 1u : (uint32_t)Dim<D>::BSHIFT, sy = (o & 2) ? 1u : (uint32_t)Dim<D>::BSHIFT;   // log2 of the region's extents
    const uint32_t lx = q & ((1u << sx) - 1u), ly = (q >> sx) & ((1u << sy) - 1u), lz = D == 3 ? q >> (sx + sy) : 0u;
    const uint32_t tx = lx + ((o & 1) ? BW : 0), ty = ly + ((o & 2) ? BW : 0), tz = lz + ((o & 4) ? BW : 0);
    return tx + TW * ty + (D == 3 ? TW * TW * tz : 0u);
}
// ... as a table in constant memory: the kernels that walk a whole slab (P2G's store, the fused G2P's tile staging) fetch
// their entries with the slab loads themselves instead of spending ~30 integer instructions per node on the arithmetic
template <int D> struct SlabTileMap {
    uint8_t t[Dim<D>::TILE];
    constexpr SlabTileMap() : t{} {
        for (int p = 0; p < Dim<D>::TILE; p++) t[p] = (uint8_t)slab_tile_of<D>((uint32_t)p);
    }
};
__constant__ SlabTileMap<WGS_DIM> g_slab_tile_map = SlabTileMap<WGS_DIM>();

// Quad access = (one uniform 64-bit buffer base in SGPRs) + (32-bit per-lane byte offset):
// `global_load_dwordx4 v[..], v_off, s[base:base+1]`. Valid while one ping-pong buffer is
// < 4 GiB (checked in wgs_data_create).
// The empty asm pins the 32-bit offset in a VGPR right at the access: otherwise LICM turns
// (base + off) into loop-invariant 64-bit VGPR address pairs, one per quad, live for the
// whole kernel.
__device__ inline float4 ldq(const float *base, uint32_t npad, int q, uint32_t i) {
    uint32_t off = ((uint32_t)q * npad + i) * 16u;
    asm volatile("" : "+v"(off));
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + off);
}
__device__ inline void stq(float *base, uint32_t npad, int q, uint32_t i, float4 v) {
    uint32_t off = ((uint32_t)q * npad + i) * 16u;
    asm volatile("" : "+v"(off));
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + off) = v;
}
// Streaming load (non-temporal hint) for data a kernel touches exactly once: the fused G2P reads every particle's state
// once and nothing else reads it in that launch, so the lines need not displace the node tiles and sort indices in L2.
// Measured at C2 on one box: G2P 46-52 -> 41-44 us, the substep 150-155 -> 143-146 us. The same hint on G2P's STORES
// gives that back (P2G of the next substep finds less in the caches), and on P2G's loads it costs 24 us: the staging
// rounds of a block touch a line several times.
__device__ inline float4 ldq_stream(const float *base, uint32_t npad, int q, uint32_t i) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    uint32_t off = ((uint32_t)q * npad + i) * 16u;
    asm volatile("" : "+v"(off));
    const v4f nv = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(base) + off));
    return make_float4(nv.x, nv.y, nv.z, nv.w);
}
// Hand-over of 16-byte values between workgroups of ONE launch (P2G's slabs to the grid-update waves): every XCD has an L2
// of its own, coherent with the others only at kernel boundaries — or for accesses made at agent scope, which the
// hardware writes through / fetches past the L2 (the sc1 bit). Relaxed atomics compile to such accesses but only up to
// 8 bytes (two half-line transactions per float4); the raw buffer instructions take the cache policy as an operand:
// buffer_{load,store}_dwordx4 ... sc1 over a descriptor of ONE slab (wave-uniform base, 32-bit byte offset), with the
// compiler's own wait counts. No fence is involved (an agent-scope fence is a whole-L2 write-back on this part: NOTES.md 4).
typedef uint32_t wgs_v4u __attribute__((ext_vector_type(4)));
constexpr int WGS_CPOL_SC1 = 16;   // gfx940+: agent scope
__device__ inline __amdgpu_buffer_rsrc_t slab_rsrc(float4 *slab_of_block, uint32_t bytes) {
    // (the base must be wave-uniform: callers pass a pointer computed from a readfirstlane'd block id)
    // (stride 0 = raw buffer: `bytes` records of one byte, every access bounds-checked against it — out of range loads return
    // zeros, stores are dropped; 0x00020000 = word 3 of a raw-buffer descriptor on gfx90a / gfx94x / gfx950: 32-bit data format)
    return __builtin_amdgcn_make_buffer_rsrc(slab_of_block, 0, bytes, 0x00020000);
}
__device__ inline void st_agent(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const float4 v) {
    const wgs_v4u u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, WGS_CPOL_SC1);
}
__device__ inline void st_plain(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const float4 v) {   // (bounds-checked, cached as usual)
    const wgs_v4u u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 0);
}
__device__ inline float4 ld_agent(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    const wgs_v4u u = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, WGS_CPOL_SC1);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
// maximum over the 64 lanes by DPP row shifts and broadcasts: half a dozen VALU instructions where six __shfl_xor steps are six
// dependent ds_bpermute round trips (P2G's longest run of a block: 0.4 of the 0.7 us between a block's record and its first fetch)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8: lane 15 of a row holds the row's maximum
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Inclusive prefix sum over the 64 lanes, likewise (rows of 16 by row_shr 1 / 2 / 4 / 8, then lane 15 of rows 0 and 2 into rows 1 and 3,
// then lane 31 into rows 2 and 3): six VALU instructions where the __shfl_up ladder is six dependent ds_bpermute round trips — a wave
// of launch 2 of the sort made two dozen such round trips per block, a quarter of its 5.6 us at rest. All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl_u32(v), 63); }
// the value of a wave-uniform lane, by v_readlane instead of a ds_bpermute round trip
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int uniform_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, uniform_lane); }
// persistent particle id (the caller's index) lives after the quads
template <int D> __device__ inline uint32_t ldpid(const float *base, uint32_t npad, uint32_t i) {
    uint32_t off = ((uint32_t)Pl<D>::NQ * 4u * npad + i) * 4u;
    asm volatile("" : "+v"(off));
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + off);
}
// Epoch (substep number) at which the particle's CDF quads were last computed. A particle whose block is
// out of reach of every collider has default_cdf() (g2p_cdf.wgsl:246-249); instead of storing 32 bytes of
// zeros for ~all particles every substep, the quads are only written for particles near a collider and
// are VALID only while this stamp equals the current (or, for "previous affinity", the previous) epoch.
// A buffer slot that is not rewritten keeps a stamp at least two epochs old (ping-pong), never a fresh one.
template <int D> __device__ inline uint32_t ldstamp(const float *base, uint32_t npad, uint32_t i) {
    uint32_t off = (((uint32_t)Pl<D>::NQ * 4u + 1u) * npad + i) * 4u;
    asm volatile("" : "+v"(off));
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + off);
}
template <int D> __device__ inline void ststamp(float *base, uint32_t npad, uint32_t i, uint32_t e) {
    uint32_t off = (((uint32_t)Pl<D>::NQ * 4u + 1u) * npad + i) * 4u;
    asm volatile("" : "+v"(off));
    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(base) + off) = e;
}
// A descriptor over one whole particle buffer (< 4 GiB by construction) for the agent-scope accesses of values handed over
// inside a launch (st_agent / ld_agent above): quad q of slot i sits at byte (q * npad + i) * 16.
template <int D> __device__ inline __amdgpu_buffer_rsrc_t particle_rsrc(float *base, uint32_t npad) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, ((uint32_t)Pl<D>::NQ * 4u + 2u) * npad * 4u, 0x00020000);
}
__device__ inline uint32_t quad_off(uint32_t npad, int q, uint32_t i) { return ((uint32_t)q * npad + i) * 16u; }
template <int D> __device__ inline uint32_t *stamp_ptr(float *base, uint32_t npad, uint32_t i) {
    return reinterpret_cast<uint32_t *>(base) + ((size_t)((uint32_t)Pl<D>::NQ * 4u + 1u) * npad + i);
}
template <int D> __device__ inline void stpid(float *base, uint32_t npad, uint32_t i, uint32_t pid) {
    uint32_t off = ((uint32_t)Pl<D>::NQ * 4u * npad + i) * 4u;
    asm volatile("" : "+v"(off));
    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(base) + off) = pid;
}

}  // namespace wgs
