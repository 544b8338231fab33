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

namespace wgs {

constexpr uint32_t NONE = 0xffffffffu;
constexpr int NPB = 64;  // nodes (= cells) per block: 4^3 or 8^2, grid.wgsl:43

template <int D> struct Dim;
template <> struct Dim<3> {
    static constexpr int BW = 4, BSHIFT = 2, TW = 6, TILE = 216, NBH = 27, DD = 9, NNBR = 8, ANG = 3;
};
template <> struct Dim<2> {
    static constexpr int BW = 8, BSHIFT = 3, TW = 10, TILE = 100, NBH = 9, DD = 4, NNBR = 4, ANG = 1;
};

// Plane indices of a particle buffer.
template <int D> struct Pl {
    static constexpr int DD = D * D;
    static constexpr int POS = 0;
    static constexpr int VEL = POS + D;
    static constexpr int F = VEL + D;
    static constexpr int C = F + DD;        // APIC matrix C' (affine), particle3d.wgsl:12
    static constexpr int MASS = C + DD;
    static constexpr int VOL = MASS + 1;    // init_volume
    static constexpr int LAM = VOL + 1;
    static constexpr int MU = LAM + 1;
    static constexpr int PID = MU + 1;      // persistent particle id (caller's index), u32 bits
    static constexpr int NBASE = PID + 1;
    // Plasticity / phase group (models/drucker_prager.wgsl:8-23, particle_update.wgsl:40-43)
    static constexpr int DP = NBASE;        // h0,h1,h2,h3,lambda,mu
    static constexpr int DPS = DP + 6;      // plastic det, hardening, log_vol_gain
    static constexpr int PHASE = DPS + 3;   // phase, max_stretch
    // CDF group (particle3d.wgsl:17-25)
    static constexpr int NRM = PHASE + 2;
    static constexpr int RVEL = NRM + D;
    static constexpr int DIST = RVEL + D;
    static constexpr int AFF = DIST + 1;    // u32 bits
    static constexpr int COUNT = AFF + 1;
};

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

struct NodeCdf {        // grid.wgsl:233-240
    float distance;
    uint32_t affinities;
    uint32_t closest_id;
    uint32_t pad;
};

// Counter slots in Dev::counters
enum { CTR_NBLOCKS = 0, CTR_ERRORS = 1, CTR_COUNT = 8 };
enum { ERRBIT_OVERFLOW = 1u, ERRBIT_KEYRANGE = 2u };

// Everything a kernel needs, passed by value.
struct Dev {
    uint32_t n;          // particles
    uint32_t npad;       // plane stride (floats)
    float *buf[2];       // ping-pong particle buffers
    uint32_t *perm;      // sorted slot -> index in the current buffer
    uint32_t *cellid;    // per particle (current-buffer index): dense block id * 64 + cell in block
    uint32_t *rank;      // per particle: position inside its cell (arrival order; canonicalised later)
    // sparse block grid (grid.wgsl:82-184): open-addressing hash of packed block keys
    uint32_t *hkeys;     // hcap: packed key or NONE
    uint32_t *hvals;     // hcap: dense block id
    uint32_t hmask;      // hcap - 1
    uint32_t cap;        // block capacity
    uint32_t *block_key;   // cap: packed virtual id
    uint32_t *block_count; // cap: particles whose associated cell is in the block
    uint32_t *block_start; // cap: exclusive scan of block_count (first_particle)
    uint32_t *nbr_plus;    // cap*8: dense ids of b + {0,1}^D (always active)
    uint32_t *nbr_minus;   // cap*8: dense ids of b - {0,1}^D or NONE
    uint32_t *cell_count;  // cap*64 (zero outside the sort)
    uint32_t *cell_start;  // cap*64
    uint32_t *cell_cursor; // cap*64: end of the cell's range in perm
    float4 *nodes;         // cap*64: velocity|momentum xyz, mass (2D: vx, vy, mass, 0)
    NodeCdf *node_cdf;     // cap*64
    float4 *slab;          // cap*TILE: per-block P2G tile (block + its "+1" rim)
    uint32_t *block_cdf_flag; // cap: block has a node with non-zero affinity
    uint32_t *counters;    // CTR_COUNT
    const SimParamsDev *sp;
    const ColliderDev *colliders;
    uint32_t n_colliders;
    float h;             // cell width
    float inv_h;
    int model;           // WGS_MODEL_*
};

__host__ __device__ inline float *plane(float *base, uint32_t npad, int p) { return base + (size_t)p * npad; }

}  // namespace wgs
