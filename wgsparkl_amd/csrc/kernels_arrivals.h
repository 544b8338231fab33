// kernels_arrivals.h — sharded runs: G2P + particle update of the particles that ARRIVE with this substep's messages
// (kernels_shard.h). g2p.wgsl:134-238 + particle_update.wgsl:45-141 for a handful of particles whose records lie in
// the inbound messages instead of the sorted buffer: no block tile, no sort entry — every (arrival, stencil node) pair
// looks its node up (the block's velocity if the block is active here, else the neighbour's partial sum from the
// message, which is then the node's total), the arrival is advanced from those 3^D values and written BEHIND the
// sorted output of the fused G2P, where the next substep's sort finds it as a new particle (k_bin, tail = 1).
// The last workgroup to finish does the bookkeeping of the migration round.
#pragma once
#include "kernels_transfer.h"

namespace wgs {

constexpr int ARR_PER_WG = 8;  // arrivals per 256-thread workgroup: 32 lanes look up the 3^D nodes of one arrival

template <int D, int MODEL, bool PLASTIC>
__global__ __launch_bounds__(256) void k_g2p_arrivals(Dev d, int side, uint32_t epoch) {
    constexpr int BW = Dim<D>::BW, BS = Dim<D>::BSHIFT, NS = Dim<D>::NBH, DD = D * D, NQ = Pl<D>::NQ, RF = particle_record_floats<D>();
    using P = Pl<D>;
    __shared__ float4 s_nv[ARR_PER_WG][NS];   // node velocity (, mass)
    __shared__ uint2 s_nc[ARR_PER_WG][NS];    // node affinity / sign bits, closest collider
    const int tid = threadIdx.x;
    const float *in_msg[2] = {d.msg.in[0], d.msg.in[1]};
    uint32_t n_in[2];
#pragma unroll
    for (int f = 0; f < 2; f++) n_in[f] = in_msg[f] ? min(reinterpret_cast<const uint32_t *>(in_msg[f])[1], d.msg.mig_cap) : 0u;
    const uint32_t n_arr_all = n_in[0] + n_in[1];
    if (blockIdx.x == 0 && tid < 2 && in_msg[tid] && n_in[tid] != 0u &&
        (reinterpret_cast<const uint32_t *>(in_msg[tid])[2] & MSG_FLAG_UNIFORM) != (d.uniform ? MSG_FLAG_UNIFORM : 0u))
        atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);  // the neighbour's particle records are laid out differently (wgs_set_uniform_material on some ranks only)
    // sorted residents of this substep: the arrivals go behind them (read by every workgroup before the last one to
    // finish rewrites the counters)
    const uint32_t s0 = d.counters[CTR_NV];
    const uint32_t room = d.n > s0 ? d.n - s0 : 0u;  // d.n = allocated capacity in sharded mode
    const uint32_t n_arr = min(n_arr_all, room);
    float *out = d.buf[side ^ 1];
    const uint32_t npad = d.npad;
    const float h = d.h, inv_h = d.inv_h, dt = d.sp->dt;
    const float invd = 4.0f / (h * h);
    const bool cpic = d.n_colliders != 0u;
    auto record_of = [&](uint32_t r) -> const float * {
        return r < n_in[0] ? msg_particles<D>(in_msg[0], d.msg.halo_cap) + (size_t)r * RF
                           : msg_particles<D>(in_msg[1], d.msg.halo_cap) + (size_t)(r - n_in[0]) * RF;
    };
    for (uint32_t base = blockIdx.x * ARR_PER_WG; base < n_arr; base += gridDim.x * ARR_PER_WG) {
        // ---- phase 1: the 3^D stencil nodes of each arrival
        {
            const int a = tid >> 5, n = tid & 31;
            const uint32_t r = base + (uint32_t)a;
            uint32_t key = NONE, b = NONE;
            int tag = 0, q = 0;
            bool need = false;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            NodeCdf nc = {0.f, 0u, NONE, 0u};
            if (r < n_arr && n < NS) {
                const float *rec = record_of(r);
                const int s[3] = {n % 3, (n / 3) % 3, D == 3 ? n / 9 : 0};
                int bcoord[3] = {0, 0, 0};
                uint32_t ln = 0u, shift = 0u;
                float pt[D];
#pragma unroll
                for (int k = 0; k < D; k++) {
                    const int c = assoc_cell(rec[k], h, inv_h, d.h_pow2 != 0u) + s[k];  // node = associated cell + shift
                    bcoord[k] = c >> BS;
                    ln |= (uint32_t)(c & (BW - 1)) << shift;
                    shift += BS;
                    pt[k] = (float)c * h;
                }
                key = pack_key<D>(bcoord);
                b = block_in_key_range<D>(bcoord) ? hmap_find(d, key, epoch) : NONE;
                halo_slot<D>(ln, tag, q);
                if (b != NONE) {  // active here: the grid update left its velocity (all contributions of both ranks)
                    v = d.nodes[(size_t)b * NPB + ln];
                    if (cpic) nc = d.node_cdf[(size_t)b * NPB + ln];
                } else {
                    need = true;
                    // (mesh colliders: the mesh part of a node cdf exists for active blocks only; an arrival that enters an
                    // empty region next to a mesh sees the analytic shapes there for this one substep)
                    if (cpic) nc = node_cdf_eval<D>(d, pt);
                }
            }
            // Nodes of blocks that are NOT active here: nobody on this rank contributes to them, so what the old owner sent is
            // the node's total (grid_update.wgsl:55-64 applied to it here); no record = the node is empty. The 32 lanes of an
            // arrival scan the headers of its message together (rare: the first particles to enter an empty region).
            {
                const int face = r < n_in[0] ? 0 : 1;
                const float *msg = r < n_arr ? in_msg[face] : nullptr;
                const uint32_t n_rec = msg ? min(reinterpret_cast<const uint32_t *>(msg)[0], d.msg.halo_cap) : 0u;
                const float4 *recs = msg ? msg_halo<D>(msg) : nullptr;
                const int sub = tid & 31;
                const unsigned long long needs = __ballot(need);
                const bool grp = ((needs >> (tid & 32)) & 0xffffffffull) != 0ull;   // some lane of my 32-lane group needs a record
                uint32_t found = NONE;
                for (uint32_t rb = 0; __ballot(grp && rb < n_rec) != 0ull; rb += 32u) {
                    const uint32_t rr = rb + (uint32_t)sub;
                    uint32_t hk = NONE, ht = NONE;
                    if (grp && rr < n_rec) {
                        const float4 hd = recs[(size_t)rr * HaloCfg<D>::REC_F4];
                        hk = __float_as_uint(hd.x);
                        ht = __float_as_uint(hd.y);
                    }
                    for (int sft = 0; sft < 32; sft++) {
                        const uint32_t k2 = (uint32_t)__shfl((int)hk, sft, 32), t2 = (uint32_t)__shfl((int)ht, sft, 32);
                        if (need && k2 == key && t2 == (uint32_t)tag && rb + (uint32_t)sft < n_rec) found = rb + (uint32_t)sft;
                    }
                }
                if (need && found != NONE) {
                    const float4 p = recs[(size_t)found * HaloCfg<D>::REC_F4 + 1 + q];
                    const float mass = D == 3 ? p.w : p.z;
                    const float inv_mass = mass > 0.f ? 1.0f / mass : 0.f;
                    const float mom[3] = {p.x, p.y, p.z};
                    const float lim = h / dt;
                    float vel[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        const float t = (mom[k] + mass * d.sp->gravity[k] * dt) * inv_mass;
                        vel[k] = fminf(fmaxf(t, -lim), lim);
                    }
                    v = D == 3 ? make_float4(vel[0], vel[1], vel[2], mass) : make_float4(vel[0], vel[1], mass, 0.f);
                }
            }
            if (r < n_arr && n < NS) {
                s_nv[a][n] = v;
                s_nc[a][n] = make_uint2(nc.affinities, nc.closest_id);
            }
        }
        __syncthreads();
        // ---- phase 2: one thread per arrival
        if (tid < ARR_PER_WG && base + (uint32_t)tid < n_arr) {
            const uint32_t r = base + (uint32_t)tid;
            const float *rec = record_of(r);
            const uint32_t j = s0 + r;  // output slot
            auto quad = [&](int qd) { return make_float4(rec[qd * 4], rec[qd * 4 + 1], rec[qd * 4 + 2], rec[qd * 4 + 3]); };
            float x[D], Fm[DD], pvel[D], mass, vol0, lambda, mu;
            if constexpr (D == 3) {
                const float4 xm = quad(P::XM), f0 = quad(P::F0), f1 = quad(P::F1), f2 = quad(Pl<3>::F2), cv = quad(P::CV2);
                x[0] = xm.x; x[1] = xm.y; x[2] = xm.z;
                Fm[0] = f0.x; Fm[1] = f0.y; Fm[2] = f0.z; Fm[3] = f0.w; Fm[4] = f1.x; Fm[5] = f1.y; Fm[6] = f1.z; Fm[7] = f1.w;
                if (d.uniform) {
                    Fm[8] = xm.w; mass = d.uni_mass; vol0 = d.uni_vol; lambda = d.uni_lambda; mu = d.uni_mu;
                } else {
                    mass = xm.w; Fm[8] = f2.x; vol0 = f2.y; lambda = f2.z; mu = f2.w;
                }
                pvel[0] = cv.y; pvel[1] = cv.z; pvel[2] = cv.w;
            } else {
                const float4 xm = quad(P::XM), f0 = quad(P::F0), vl = quad(P::CV2);
                x[0] = xm.x; x[1] = xm.y; mass = xm.z; vol0 = xm.w;
                Fm[0] = f0.x; Fm[1] = f0.y; Fm[2] = f0.z; Fm[3] = f0.w;
                lambda = vl.z; mu = vl.w;
                pvel[0] = vl.x; pvel[1] = vl.y;
            }
            const uint32_t pid = __float_as_uint(rec[NQ * 4]);
            // particle cdf: valid only if the old owner computed it in this substep's P2G prologue (else default_cdf())
            const bool cdf_live = cpic && __float_as_uint(rec[NQ * 4 + 1]) == epoch;
            float nrm[D], sdist = 0.f;
            uint32_t paff = 0u;
#pragma unroll
            for (int k = 0; k < D; k++) nrm[k] = 0.f;
            if (cdf_live) {
                const float4 c0 = quad(P::CDF0), c1 = quad(P::CDF1);
                nrm[0] = c0.x; nrm[1] = c0.y;
                if constexpr (D == 3) { nrm[2] = c0.z; sdist = c0.w; paff = __float_as_uint(c1.w); }
                else { sdist = c0.z; paff = __float_as_uint(c0.w); }
            }
            // ---- G2P (g2p.wgsl:150-218), the 3^D-term form
            float ref[D], w[D][3];
#pragma unroll
            for (int k = 0; k < D; k++) {
                const int c = assoc_cell(x[k], h, inv_h, d.h_pow2 != 0u);
                ref[k] = (float)c * h - x[k];
                eval_all(-ref[k] * inv_h, w[k]);
            }
            float vel[D], grad[DD];
#pragma unroll
            for (int k = 0; k < D; k++) vel[k] = 0.f;
#pragma unroll
            for (int k = 0; k < DD; k++) grad[k] = 0.f;
#pragma unroll 1
            for (int n = 0; n < NS; n++) {
                const int sx = n % 3, sy = (n / 3) % 3, sz = D == 3 ? n / 9 : 0;
                const float4 nd = s_nv[tid][n];
                const uint2 ncd = s_nc[tid][n];
                float nv[D], dpt[D];
                nv[0] = nd.x; nv[1] = nd.y;
                if constexpr (D == 3) nv[2] = nd.z;
                dpt[0] = ref[0] + (float)sx * h;
                dpt[1] = ref[1] + (float)sy * h;
                if constexpr (D == 3) dpt[2] = ref[2] + (float)sz * h;
                float wgt = w[0][sx] * w[1][sy];
                if constexpr (D == 3) wgt *= (sz == 0 ? w[2][0] : (sz == 1 ? w[2][1] : w[2][2]));
                if (!affinities_are_compatible(paff, ncd.x)) {
                    if (ncd.y != NONE && ncd.y < d.n_colliders) {
                        const ColliderDev &col = d.colliders[ncd.y];
                        float cc[D], bv[D], rel[D], pr[D];
#pragma unroll
                        for (int k = 0; k < D; k++) cc[k] = dpt[k] + x[k];
                        velocity_at_point<D>(col, cc, bv);
#pragma unroll
                        for (int k = 0; k < D; k++) rel[k] = pvel[k] - bv[k];
                        project_velocity<D>(rel, nrm, pr);
#pragma unroll
                        for (int k = 0; k < D; k++) nv[k] = bv[k] + pr[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < D; k++) nv[k] = pvel[k];
                    }
                }
                const float wi = wgt * invd;
#pragma unroll
                for (int k = 0; k < D; k++) vel[k] += nv[k] * wgt;
#pragma unroll
                for (int c = 0; c < D; c++)
#pragma unroll
                    for (int rr = 0; rr < D; rr++) grad[c * D + rr] += wi * (nv[rr] * dpt[c]);
            }
            float rvel[D];
#pragma unroll
            for (int k = 0; k < D; k++) rvel[k] = 0.f;
            if (cdf_live) {  // g2p.wgsl:220-226
                for (uint32_t c = 0; c < d.n_colliders && c < 16u; c++)
                    if (paff & (1u << c)) {
                        float bv[D];
                        velocity_at_point<D>(d.colliders[c], x, bv);
#pragma unroll
                        for (int k = 0; k < D; k++) rvel[k] += bv[k];
                    }
            }
            // ---- particle update (particle_update.wgsl:58-132)
            if (cdf_live && sdist < -0.05f * h) {
                float rel[D], pr[D];
#pragma unroll
                for (int k = 0; k < D; k++) rel[k] = vel[k] - rvel[k];
                project_velocity<D>(rel, nrm, pr);
#pragma unroll
                for (int k = 0; k < D; k++) vel[k] = rvel[k] + pr[k];
            }
            float l2 = 0.f;
#pragma unroll
            for (int k = 0; k < D; k++) l2 += vel[k] * vel[k];
            const float len = sqrtf(l2);
            if (len > h / dt) {
#pragma unroll
                for (int k = 0; k < D; k++) vel[k] = vel[k] / len * h / dt;
            }
            float xn[D];
#pragma unroll
            for (int k = 0; k < D; k++) xn[k] = x[k] + vel[k] * dt;
            {  // already on its way out again? then it is a guest of the next substep
                const int nbx = assoc_cell(xn[0], h, inv_h, d.h_pow2 != 0u) >> BS;
                if (nbx < d.shard_lo || nbx >= d.shard_hi) {
                    const uint32_t ls = atomicAdd(&d.counters[CTR_NLEAVE], 1u);
                    if (ls < d.leavers_cap) d.leavers[ls] = j;
                    else atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);
                }
            }
            if (cdf_live && sdist < -0.05f * h) {
                const float corrected = fmaxf(sdist, -0.3f * h);
                const float imp = dt * -corrected * 1.0e3f;
#pragma unroll
                for (int k = 0; k < D; k++) vel[k] += imp * nrm[k];
            }
            float gdt[DD], prod[DD];
#pragma unroll
            for (int k = 0; k < DD; k++) gdt[k] = grad[k] * dt;
            mat_mul<D>(gdt, Fm, prod);
#pragma unroll
            for (int k = 0; k < DD; k++) Fm[k] += prod[k];
            float tau[DD];
            Svd<D> sv;
            if constexpr (PLASTIC) {
                const float4 d0 = quad(P::DP0), d1 = quad(P::DP1), d2 = quad(P::DP2);
                float dp[6] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y}, st[3] = {d1.z, d1.w, d2.x}, phase = d2.y;
                const float max_stretch = d2.z;
                const bool breakable = phase > 0.f && max_stretch > 0.f;
                if (MODEL != 1 || breakable || (phase == 0.f && dp[4] != 0.f)) svd<D>(Fm, sv);
                if (breakable) {
                    bool broken = false;
#pragma unroll
                    for (int k = 0; k < D; k++) broken = broken || sv.s[k] > max_stretch;
                    if (broken) phase = 0.f;
                }
                if (phase == 0.f && dp[4] != 0.f) drucker_prager_project<D>(dp, st, Fm, sv);
                stq(out, npad, P::DP0, j, make_float4(dp[0], dp[1], dp[2], dp[3]));
                stq(out, npad, P::DP1, j, make_float4(dp[4], dp[5], st[0], st[1]));
                stq(out, npad, P::DP2, j, make_float4(st[2], phase, max_stretch, 0.f));
            }
            if constexpr (MODEL == 1) {
                kirchoff_neo_hookean<D>(lambda, mu, Fm, tau);
            } else {
                if constexpr (!PLASTIC) svd<D>(Fm, sv);
                kirchoff_corotated<D>(lambda, mu, Fm, sv, tau);
            }
            const float coeff = vol0 * invd * dt;
            float Cn[DD];
#pragma unroll
            for (int k = 0; k < DD; k++) Cn[k] = grad[k] * mass - tau[k] * coeff;
            if constexpr (D == 3) {
                stq(out, npad, P::XM, j, make_float4(xn[0], xn[1], xn[2], d.uniform ? Fm[8] : mass));
                stq(out, npad, P::CV0, j, make_float4(Cn[0], Cn[1], Cn[2], Cn[3]));
                stq(out, npad, Pl<3>::CV1, j, make_float4(Cn[4], Cn[5], Cn[6], Cn[7]));
                stq(out, npad, P::CV2, j, make_float4(Cn[8], vel[0], vel[1], vel[2]));
                stq(out, npad, P::F0, j, make_float4(Fm[0], Fm[1], Fm[2], Fm[3]));
                stq(out, npad, Pl<3>::F1, j, make_float4(Fm[4], Fm[5], Fm[6], Fm[7]));
                if (!d.uniform) stq(out, npad, Pl<3>::F2, j, make_float4(Fm[8], vol0, lambda, mu));
                stq(out, npad, P::CDF0, j, make_float4(nrm[0], nrm[1], nrm[2], sdist));
                stq(out, npad, P::CDF1, j, make_float4(rvel[0], rvel[1], rvel[2], __uint_as_float(paff)));
            } else {
                stq(out, npad, P::XM, j, make_float4(xn[0], xn[1], mass, vol0));
                stq(out, npad, P::CV0, j, make_float4(Cn[0], Cn[1], Cn[2], Cn[3]));
                stq(out, npad, P::CV2, j, make_float4(vel[0], vel[1], lambda, mu));
                stq(out, npad, P::F0, j, make_float4(Fm[0], Fm[1], Fm[2], Fm[3]));
                stq(out, npad, P::CDF0, j, make_float4(nrm[0], nrm[1], sdist, __uint_as_float(paff)));
                stq(out, npad, P::CDF1, j, make_float4(rvel[0], rvel[1], 0.f, 0.f));
            }
            if constexpr (!PLASTIC) {  // (the quads travel with the particle even when this simulation never reads them)
                stq(out, npad, P::DP0, j, quad(P::DP0));
                stq(out, npad, P::DP1, j, quad(P::DP1));
                stq(out, npad, P::DP2, j, quad(P::DP2));
            }
            stpid<D>(out, npad, j, pid);
            ststamp<D>(out, npad, j, cdf_live ? epoch : 0u);  // (0: stale — default_cdf(), layout.h)
        }
        __syncthreads();
    }
    // ---- bookkeeping of the migration round, by the last workgroup to get here (every other one has read CTR_NV):
    // the buffer just written holds the sorted output [0, s0) — the guests' slots vacated — and the arrivals behind it
    if (tid == 0) {
        __threadfence();
        const uint32_t ticket = atomicAdd(&d.counters[CTR_TICKET], 1u);
        if (ticket == gridDim.x - 1u) {
            uint32_t sent = 0u;
#pragma unroll
            for (int f = 0; f < 2; f++)
                if (d.msg.out[f]) sent += min(reinterpret_cast<const uint32_t *>(d.msg.out[f])[1], d.msg.mig_cap);
            if (n_arr < n_arr_all) atomicOr(&d.counters[CTR_ERRORS], ERRBIT_SHARD);  // particle capacity of the slab exhausted
            d.counters[CTR_NPREV] = s0;
            d.counters[CTR_N] = s0 + n_arr;
            d.counters[CTR_NV] = s0 - min(sent, s0) + n_arr;
            d.counters[CTR_TICKET] = 0u;
        }
    }
}

}  // namespace wgs
