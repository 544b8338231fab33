// kernels_arrivals.h — sharded runs: G2P + particle update of the particles that ARRIVE with this substep's messages
// (kernels_shard.h). g2p.wgsl:134-238 + particle_update.wgsl:45-141 for a handful of particles whose records lie in
// the inbound messages instead of the sorted buffer: no block tile, no sort entry — every (arrival, stencil node) pair
// looks its node up (the block's velocity if the block is active here, else the neighbour's partial sum from the
// message, which is then the node's total), the arrival is advanced from those 3^D values and written BEHIND the
// sorted output of the fused G2P, in slots [NPREV, N), and binned for the next substep's sort through the hash path like a particle
// that changed block — here (Dev::bin_next), or by launch 1 of that sort (k_rebin; k_bin on a table rebuild) when the fused G2P
// does not bin. A launch of its own behind the fused G2P (capi.hip);
// its last workgroup to finish does the bookkeeping of the migration round (the particle counters of the next substep).
#pragma once
#include "kernels_transfer.h"

namespace wgs {

constexpr int ARR_PER_WG = 8;  // arrivals per 256-thread workgroup: 32 lanes look up the 3^D nodes of one arrival

// (always a launch of its own: inside the fused G2P launch it cost more than the launch it saved)
template <int D, int MODEL, bool PLASTIC>
__global__ __launch_bounds__(256) void k_g2p_arrivals(Dev d, int side, uint32_t epoch) {
    __shared__ float4 s_nv[ARR_PER_WG][Dim<D>::NBH];   // node velocity (, mass)
    __shared__ uint2 s_nc[ARR_PER_WG][Dim<D>::NBH];    // node affinity / sign bits, closest collider
#define ARR_NPW ARR_PER_WG
#define ARR_BX blockIdx.x
#define ARR_GX gridDim.x
#include "arrivals_body.inc"
#undef ARR_NPW
#undef ARR_BX
#undef ARR_GX
}

}  // namespace wgs
