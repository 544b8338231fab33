/*
 * mpm_oracle.c — CPU restatement of the wgsparkl MLS-MPM substep (see mpm_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY (checker for tests/, smoke(), bench.py cpu_baseline).
 * Parity: pinned for the prefix sum only (reference test prefix_sum.rs:170-231);
 * everything else "parity unpinned" — restated from the WGSL cited at each
 * function, with deterministic (ascending particle id) list order where the
 * reference's order is decided by atomic races (sort.wgsl:126,133).
 *
 * All `file:line` citations are relative to /root/reference/src/.
 */
#include "mpm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define D ORC_DIM
#define DD ORC_DD
#define BW ORC_BLOCK_W
#define NPB ORC_NODES_PER_BLOCK

#define R(x) ((real)(x))
static inline real r_sqrt(real x) { return (real)sqrt((double)x); }
static inline real r_log(real x) { return sizeof(real) == 4 ? (real)logf((float)x) : (real)log((double)x); }
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_sin(real x) { return sizeof(real) == 4 ? (real)sinf((float)x) : (real)sin((double)x); }
static inline real r_abs(real x) { return x < 0 ? -x : x; }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_min(real a, real b) { return a < b ? a : b; }

#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
#else
int orc_num_threads(void) { return 1; }
#endif
int orc_dim(void) { return D; }
int orc_real_size(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------------ */
/* Index math                                                               */
/* ------------------------------------------------------------------------ */

/* grid/grid.wgsl:82-95 */
uint32_t orc_pack_key(const int32_t *b) {
#if D == 2
    return ((uint32_t)(b[0] + 0x00007fff) & 0x0000ffffu) |
           (((uint32_t)(b[1] + 0x00007fff) & 0x0000ffffu) << 16);
#else
    return ((uint32_t)(b[0] + 0x000003ff) & 0x000007ffu) |
           (((uint32_t)(b[1] + 0x000001ff) & 0x000003ffu) << 11) |
           (((uint32_t)(b[2] + 0x000003ff) & 0x000007ffu) << 21);
#endif
}

/* grid/grid.wgsl:98-105 (murmur3 scramble) */
uint32_t orc_hash(uint32_t key) {
    key *= 0xcc9e2d51u;
    key = (key << 15) | (key >> 17);
    key *= 0x1b873593u;
    return key;
}

/* solver/particle3d.wgsl:41-49: assoc_cell = round(pt / h) - 1 (WGSL round = ties to even,
 * true fp32 division). */
void orc_assoc_cell(const float *pt, float h, int32_t *cell) {
    for (int k = 0; k < D; k++) {
        float q = pt[k] / h;
        float r = rintf(q) - 1.0f;
        cell[k] = (int32_t)r;
    }
}

/* grid/grid.wgsl:284-292 block_associated_to_point + particle3d.wgsl:41-45 */
void orc_block_and_local(const float *pt, float h, int32_t *block, uint32_t *local) {
    for (int k = 0; k < D; k++) {
        float assoc_cell = rintf(pt[k] / h) - 1.0f;
        float assoc_block = floorf(assoc_cell / (float)BW);
        block[k] = (int32_t)assoc_block;
        local[k] = (uint32_t)(assoc_cell - assoc_block * (float)BW);
    }
}

static void pos_f32(const orc_particles *p, int i, float *out) {
    for (int k = 0; k < D; k++) out[k] = (float)p->pos[i * D + k];
}

/* grid/grid.wgsl:346-348 node_id */
static inline uint32_t node_local_index(const uint32_t *l) {
#if D == 2
    return l[0] + l[1] * 8u;
#else
    return l[0] + l[1] * 4u + l[2] * 16u;
#endif
}

/* ------------------------------------------------------------------------ */
/* Prefix sum                                                               */
/* ------------------------------------------------------------------------ */

/* grid/prefix_sum.rs:71-83 WgPrefixSum::eval_cpu — exclusive scan "as if a 0 was appended". */
void orc_prefix_sum_eval_cpu(uint32_t *v, int32_t len) {
    if (len <= 0) return;
    for (int32_t i = 0; i < len - 1; i++) v[i + 1] += v[i];
    for (int32_t i = len - 1; i >= 1; i--) v[i] = v[i - 1];
    v[0] = 0;
}

static uint32_t next_pow2(uint32_t v) { /* prefix_sum.wgsl:95-105 */
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    v++;
    return v;
}

/* Restates the GPU algorithm itself: per-256 Blelloch up/down sweep
 * (prefix_sum.wgsl:11-84), recursive scan of the block totals
 * (prefix_sum.rs:36-50) and add_data_grp (prefix_sum.wgsl:86-93). */
void orc_prefix_sum_gpu_algorithm(uint32_t *data, int32_t len) {
    const uint32_t WG = 256;
    if (len <= 0) return;
    uint32_t ngroups = ((uint32_t)len + WG - 1) / WG;
    uint32_t *aux = (uint32_t *)calloc(ngroups, sizeof(uint32_t));
    for (uint32_t bid = 0; bid < ngroups; bid++) {
        uint32_t ws[256];
        uint32_t block_len = (uint32_t)len - bid * WG;
        uint32_t shared_len = next_pow2(block_len);
        if (shared_len < 1) shared_len = 1;
        if (shared_len > WG) shared_len = WG;
        for (uint32_t t = 0; t < WG; t++) {
            uint32_t e = t + bid * WG;
            ws[t] = e < (uint32_t)len ? data[e] : 0u;
        }
        uint32_t offset = 1;
        for (uint32_t d = shared_len / 2; d > 0; d /= 2) {
            for (uint32_t t = 0; t < d; t++) {
                uint32_t ia = t * 2 * offset + offset - 1;
                uint32_t ib = (t * 2 + 1) * offset + offset - 1;
                ws[ib] = ws[ia] + ws[ib];
            }
            offset *= 2;
        }
        aux[bid] = ws[shared_len - 1];
        ws[shared_len - 1] = 0;
        offset = shared_len / 2;
        for (uint32_t d = 1; d < shared_len; d *= 2) {
            for (uint32_t t = 0; t < d; t++) {
                uint32_t ia = t * 2 * offset + offset - 1;
                uint32_t ib = (t * 2 + 1) * offset + offset - 1;
                uint32_t a = ws[ia], b = ws[ib];
                ws[ia] = b;
                ws[ib] = a + b;
            }
            offset /= 2;
        }
        for (uint32_t t = 0; t < WG; t++) {
            uint32_t e = t + bid * WG;
            if (e < (uint32_t)len) data[e] = ws[t];
        }
    }
    if (ngroups > 1) {
        orc_prefix_sum_gpu_algorithm(aux, (int32_t)ngroups);
        for (int32_t i = 0; i < len; i++) data[i] += aux[i / (int32_t)WG];
    }
    free(aux);
}

/* ------------------------------------------------------------------------ */
/* Quadratic B-spline kernel — grid/kernel.wgsl                             */
/* ------------------------------------------------------------------------ */

/* kernel.wgsl:7-17 / 22-50. Order matters: it is the order of the 27 (9)
 * additions in p2g_step / particle_g2p. */
#if D == 2
static const int NBH_SHIFTS[9][2] = {
    {2, 2}, {2, 0}, {2, 1}, {0, 2}, {0, 0}, {0, 1}, {1, 2}, {1, 0}, {1, 1},
};
#else
static const int NBH_SHIFTS[27][3] = {
    {2, 2, 2}, {2, 0, 2}, {2, 1, 2}, {0, 2, 2}, {0, 0, 2}, {0, 1, 2}, {1, 2, 2}, {1, 0, 2}, {1, 1, 2},
    {2, 2, 0}, {2, 0, 0}, {2, 1, 0}, {0, 2, 0}, {0, 0, 0}, {0, 1, 0}, {1, 2, 0}, {1, 0, 0}, {1, 1, 0},
    {2, 2, 1}, {2, 0, 1}, {2, 1, 1}, {0, 2, 1}, {0, 0, 1}, {0, 1, 1}, {1, 2, 1}, {1, 0, 1}, {1, 1, 1},
};
#endif

int orc_nbh_shift(int i, int axis) { return NBH_SHIFTS[i][axis]; }

/* kernel.wgsl:18-20 / 51-53: flattened index of the shift in the (BW+2)^D shared tile. */
int orc_nbh_shift_shared(int i) {
#if D == 2
    return NBH_SHIFTS[i][0] + NBH_SHIFTS[i][1] * 10;
#else
    return NBH_SHIFTS[i][0] + NBH_SHIFTS[i][1] * 6 + NBH_SHIFTS[i][2] * 36;
#endif
}

/* kernel.wgsl:56-58 */
static inline real inv_d(real h) { return R(4.0) / (h * h); }

/* kernel.wgsl:60-66 */
void orc_eval_all(real x, real *w) {
    w[0] = R(0.5) * (R(1.5) - x) * (R(1.5) - x);
    w[1] = R(0.75) - (x - R(1.0)) * (x - R(1.0));
    w[2] = R(0.5) * (x - R(0.5)) * (x - R(0.5));
}

/* particle3d.wgsl:47-57 dir_to_associated_grid_node: (round(x/h) - 1) * h - x.
 * Computed in the oracle's precision from the fp32-exact cell index. */
static void dir_to_assoc(const orc_particles *p, int i, real h, real *ref) {
    float pf[3];
    int32_t cell[3];
    pos_f32(p, i, pf);
    orc_assoc_cell(pf, (float)h, cell);
    for (int k = 0; k < D; k++) ref[k] = (real)cell[k] * h - p->pos[i * D + k];
}

/* kernel.wgsl:84-105 precompute_weights: w[axis] = eval_all(-ref[axis] / h) */
static void precompute_weights(const real *ref, real h, real w[D][3]) {
    for (int k = 0; k < D; k++) orc_eval_all(-ref[k] / h, w[k]);
}

/* ------------------------------------------------------------------------ */
/* Small dense linear algebra (column-major)                                */
/* ------------------------------------------------------------------------ */

static void mat_mul(const real *a, const real *b, real *out) {
    real t[DD];
    for (int c = 0; c < D; c++)
        for (int r = 0; r < D; r++) {
            real s = 0;
            for (int k = 0; k < D; k++) s += a[k * D + r] * b[c * D + k];
            t[c * D + r] = s;
        }
    memcpy(out, t, sizeof(t));
}

static void mat_vec(const real *a, const real *v, real *out) {
    real t[D];
    for (int r = 0; r < D; r++) {
        real s = 0;
        for (int c = 0; c < D; c++) s += a[c * D + r] * v[c];
        t[r] = s;
    }
    memcpy(out, t, sizeof(t));
}

static void mat_transpose(const real *a, real *out) {
    real t[DD];
    for (int c = 0; c < D; c++)
        for (int r = 0; r < D; r++) t[c * D + r] = a[r * D + c];
    memcpy(out, t, sizeof(t));
}

static real mat_det(const real *m) {
#if D == 2
    return m[0] * m[3] - m[2] * m[1];
#else
    /* m(r,c) = m[c*3+r] */
    return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2]) +
           m[6] * (m[1] * m[5] - m[4] * m[2]);
#endif
}

static double det_d(const double *m) {
#if D == 2
    return m[0] * m[3] - m[2] * m[1];
#else
    return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2]) +
           m[6] * (m[1] * m[5] - m[4] * m[2]);
#endif
}

/*
 * SVD F = U * diag(S) * Vt. The reference calls wgebra::svd2/svd3 (third party,
 * dimforge/wgmath rev 6d17942, not on disk) at linear_elasticity.wgsl:15,29,
 * drucker_prager.wgsl:80,139, particle_update.wgsl:103,109. This is our own
 * one-sided Jacobi in fp64, rounded to `real` at the end. Convention (the usual
 * graphics one, assumed for wgebra): U and V are proper rotations; when
 * det F < 0 the sign goes on the singular value of smallest magnitude.
 */
void orc_svd(const real *m, real *u_out, real *s_out, real *vt_out) {
    double a[DD], v[DD];
    for (int i = 0; i < DD; i++) { a[i] = (double)m[i]; v[i] = 0.0; }
    for (int i = 0; i < D; i++) v[i * D + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < D - 1; p++)
            for (int q = p + 1; q < D; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < D; r++) {
                    alpha += a[p * D + r] * a[p * D + r];
                    beta += a[q * D + r] * a[q * D + r];
                    gamma += a[p * D + r] * a[q * D + r];
                }
                if (gamma == 0.0 || fabs(gamma) <= 1e-300) continue;
                if (fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
                off += fabs(gamma);
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < D; r++) {
                    double ap = a[p * D + r], aq = a[q * D + r];
                    a[p * D + r] = c * ap - s * aq;
                    a[q * D + r] = s * ap + c * aq;
                    double vp = v[p * D + r], vq = v[q * D + r];
                    v[p * D + r] = c * vp - s * vq;
                    v[q * D + r] = s * vp + c * vq;
                }
            }
        if (off == 0.0) break;
    }
    double sig[D], u[DD];
    for (int c = 0; c < D; c++) {
        double n = 0;
        for (int r = 0; r < D; r++) n += a[c * D + r] * a[c * D + r];
        sig[c] = sqrt(n);
    }
    /* Normalise columns; rebuild the ones with a vanishing singular value. */
    double smax = 0;
    for (int c = 0; c < D; c++) if (sig[c] > smax) smax = sig[c];
    int ok[D];
    for (int c = 0; c < D; c++) {
        ok[c] = sig[c] > 1e-150 && sig[c] > 1e-14 * smax;
        for (int r = 0; r < D; r++) u[c * D + r] = ok[c] ? a[c * D + r] / sig[c] : 0.0;
        if (!ok[c]) sig[c] = 0.0;
    }
#if D == 2
    for (int c = 0; c < 2; c++)
        if (!ok[c]) {
            int o = 1 - c;
            if (ok[o]) { u[c * 2 + 0] = -u[o * 2 + 1]; u[c * 2 + 1] = u[o * 2 + 0]; }
            else { u[c * 2 + 0] = c == 0; u[c * 2 + 1] = c == 1; }
            ok[c] = 1;
        }
#else
    {
        int nbad = !ok[0] + !ok[1] + !ok[2];
        if (nbad == 3) {
            for (int i = 0; i < 9; i++) u[i] = (i % 4 == 0);
        } else if (nbad == 2) {
            int g = ok[0] ? 0 : (ok[1] ? 1 : 2);
            double *e = &u[g * 3];
            /* pick the axis least aligned with e */
            int ax = 0;
            if (fabs(e[1]) < fabs(e[ax])) ax = 1;
            if (fabs(e[2]) < fabs(e[ax])) ax = 2;
            double t[3] = {0, 0, 0};
            t[ax] = 1.0;
            double d = e[0] * t[0] + e[1] * t[1] + e[2] * t[2];
            double b1[3] = {t[0] - d * e[0], t[1] - d * e[1], t[2] - d * e[2]};
            double n1 = sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
            for (int r = 0; r < 3; r++) b1[r] /= n1;
            double b2[3] = {e[1] * b1[2] - e[2] * b1[1], e[2] * b1[0] - e[0] * b1[2], e[0] * b1[1] - e[1] * b1[0]};
            int c1 = (g + 1) % 3, c2 = (g + 2) % 3;
            for (int r = 0; r < 3; r++) { u[c1 * 3 + r] = b1[r]; u[c2 * 3 + r] = b2[r]; }
        } else if (nbad == 1) {
            int b = !ok[0] ? 0 : (!ok[1] ? 1 : 2);
            int c1 = (b + 1) % 3, c2 = (b + 2) % 3;
            double *x = &u[c1 * 3], *y = &u[c2 * 3];
            u[b * 3 + 0] = x[1] * y[2] - x[2] * y[1];
            u[b * 3 + 1] = x[2] * y[0] - x[0] * y[2];
            u[b * 3 + 2] = x[0] * y[1] - x[1] * y[0];
        }
    }
#endif
    /* Make U and V proper rotations; put the sign on the smallest singular value. */
    int kmin = 0;
    for (int c = 1; c < D; c++) if (sig[c] < sig[kmin]) kmin = c;
    if (det_d(v) < 0) {
        for (int r = 0; r < D; r++) v[kmin * D + r] = -v[kmin * D + r];
        sig[kmin] = -sig[kmin];
    }
    if (det_d(u) < 0) {
        for (int r = 0; r < D; r++) u[kmin * D + r] = -u[kmin * D + r];
        sig[kmin] = -sig[kmin];
    }
    for (int c = 0; c < D; c++) {
        s_out[c] = (real)sig[c];
        for (int r = 0; r < D; r++) {
            u_out[c * D + r] = (real)u[c * D + r];
            vt_out[c * D + r] = (real)v[r * D + c]; /* Vt(r,c) = V(c,r) */
        }
    }
}

/* wgebra Svd::recompose: U * diag(S) * Vt */
static void svd_recompose(const real *u, const real *s, const real *vt, real *out) {
    real us[DD];
    for (int c = 0; c < D; c++)
        for (int r = 0; r < D; r++) us[c * D + r] = u[c * D + r] * s[c];
    mat_mul(us, vt, out);
}

/* ------------------------------------------------------------------------ */
/* Constitutive models                                                      */
/* ------------------------------------------------------------------------ */

/* models/linear_elasticity.wgsl:14-41 (corotated) and
 * models/neo_hookean_elasticity.wgsl:12-25 */
void orc_kirchoff_stress(int model, real lambda, real mu, const real *F, real *tau) {
    if (model == ORC_MODEL_NEO_HOOKEAN) {
        real j = r_max(mat_det(F), R(1.0e-10));
        real diag = lambda * r_log(j) - mu;
        real Ft[DD], FFt[DD];
        mat_transpose(F, Ft);
        mat_mul(F, Ft, FFt);
        for (int i = 0; i < DD; i++) tau[i] = mu * FFt[i];
        for (int k = 0; k < D; k++) tau[k * D + k] += diag;
        return;
    }
    real u[DD], s[D], vt[DD];
    orc_svd(F, u, s, vt);
    real j = s[0];
    for (int k = 1; k < D; k++) j = j * s[k];
    for (int k = 0; k < D; k++) s[k] -= R(1.0);
    real diag = lambda * (j - R(1.0)) * j;
    real rec[DD], Ft[DD], prod[DD];
    svd_recompose(u, s, vt, rec);
    mat_transpose(F, Ft);
    mat_mul(rec, Ft, prod);
    for (int i = 0; i < DD; i++) tau[i] = prod[i] * (R(2.0) * mu);
    for (int k = 0; k < D; k++) tau[k * D + k] += diag;
}

/* models/drucker_prager.wgsl:25-29 */
static real dp_alpha(const real *dp, real q) {
    real angle = dp[0] + (dp[1] * q - dp[3]) * r_exp(-dp[2] * q);
    real s_angle = r_sin(angle);
    return r_sqrt(R(2.0) / R(3.0)) * (R(2.0) * s_angle) / (R(3.0) - s_angle);
}

/* models/drucker_prager.wgsl:42-64 (2D), :112-131 (3D) + project :66-101 / :133-158.
 * Returns 1 when the state/deformation gradient were modified. */
int orc_drucker_prager_project(const real *dp, real *state, real *F) {
    if (dp[4] == 0) return 0; /* plasticity.lambda == 0: disabled (drucker_prager.wgsl:134) */
    real u[DD], sv[D], vt[DD];
    orc_svd(F, u, sv, vt);
    real alpha = dp_alpha(dp, state[1]);
    const real d = (real)D;
    real strain[D], dev[D], new_sv[D];
    real trace = 0;
    for (int k = 0; k < D; k++) { strain[k] = r_log(sv[k]) + state[2] / d; }
    for (int k = 0; k < D; k++) trace += strain[k];
    int all_zero = 1;
    for (int k = 0; k < D; k++) { dev[k] = strain[k] - trace / d; if (dev[k] != 0) all_zero = 0; }
    real hardening;
    if (trace > 0 || all_zero) {
        real n = 0;
        for (int k = 0; k < D; k++) { new_sv[k] = 1; n += strain[k] * strain[k]; }
        hardening = r_sqrt(n);
    } else {
        real n = 0;
        for (int k = 0; k < D; k++) n += dev[k] * dev[k];
        real dev_norm = r_sqrt(n);
        real gamma = dev_norm + (d * dp[4] + R(2.0) * dp[5]) / (R(2.0) * dp[5]) * trace * alpha;
        if (gamma <= 0) return 0; /* valid == false → unchanged */
        for (int k = 0; k < D; k++) new_sv[k] = r_exp(strain[k] - dev[k] * (gamma / dev_norm));
        hardening = gamma;
    }
    real prev_det = sv[0], new_det = new_sv[0];
    for (int k = 1; k < D; k++) { prev_det = prev_det * sv[k]; new_det = new_det * new_sv[k]; }
    state[0] = state[0] * prev_det / new_det;
    state[2] = state[2] + r_log(prev_det) - r_log(new_det);
    state[1] = state[1] + hardening;
    svd_recompose(u, new_sv, vt, F);
    return 1;
}

/* ------------------------------------------------------------------------ */
/* Sparse grid hash map — grid/grid.wgsl:121-184, 323-334                   */
/* ------------------------------------------------------------------------ */

static uint32_t hmap_find(const orc_grid *g, const int32_t *block) {
    uint32_t key = orc_pack_key(block);
    uint32_t mask = (uint32_t)g->hmap_capacity - 1u;
    uint32_t slot = orc_hash(key) & mask;
    for (int32_t k = 0; k < g->hmap_capacity; k++) {
        uint32_t st = g->hmap_state[slot];
        if (st == key) return g->hmap_value[slot];
        if (st == ORC_NONE) return ORC_NONE;
        slot = (slot + 1u) & mask;
    }
    return ORC_NONE;
}

static void mark_block_as_active(orc_grid *g, const int32_t *block) {
    uint32_t key = orc_pack_key(block);
    uint32_t cap = (uint32_t)g->hmap_capacity;
    uint32_t slot = orc_hash(key) & (cap - 1u);
    for (uint32_t k = 0; k < cap; k++) {
        uint32_t st = g->hmap_state[slot];
        if (st == ORC_NONE) {
            if (g->n_blocks >= g->cap_blocks) { g->overflow = 1; return; }
            g->hmap_state[slot] = key;
            uint32_t id = (uint32_t)g->n_blocks++;
            for (int a = 0; a < D; a++) g->block_vid[id * D + a] = block[a];
            g->first_particle[id] = 0;
            g->num_particles[id] = 0;
            g->hmap_value[slot] = id;
            return;
        } else if (st == key) {
            return;
        }
        slot = (slot + 1u) % cap & (cap - 1u);
    }
    g->overflow = 1; /* the reference silently drops the block (grid.wgsl:126-128,163) */
}

/* ------------------------------------------------------------------------ */
/* Sort — grid/grid.rs:30-207, grid/sort.wgsl:26-36,89-137, grid.wgsl:186-203,362-379 */
/* ------------------------------------------------------------------------ */
static void sort_impl(const orc_particles *p, const orc_params *prm, orc_grid *g, orc_rigid *rig) {
    const float h = (float)prm->cell_width;
    /* reset_hmap */
    for (int32_t i = 0; i < g->hmap_capacity; i++) { g->hmap_state[i] = ORC_NONE; g->hmap_value[i] = 0; }
    g->n_blocks = 0;
    g->overflow = 0;
    /* touch_particle_blocks: the associated block and its "+1" neighbours
     * (grid.wgsl:300-320 order). */
    for (int32_t i = 0; i < p->n; i++) {
        float pf[3];
        int32_t b[3];
        uint32_t l[3];
        pos_f32(p, i, pf);
        orc_block_and_local(pf, h, b, l);
#if D == 2
        for (int ox = 0; ox <= 1; ox++)
            for (int oy = 0; oy <= 1; oy++) {
                int32_t nb[2] = {b[0] + ox, b[1] + oy};
                mark_block_as_active(g, nb);
            }
#else
        for (int ox = 0; ox <= 1; ox++)
            for (int oy = 0; oy <= 1; oy++)
                for (int oz = 0; oz <= 1; oz++) {
                    int32_t nb[3] = {b[0] + ox, b[1] + oy, b[2] + oz};
                    mark_block_as_active(g, nb);
                }
#endif
    }
    if (rig) {
        /* mark_rigid_particles_needing_block (sort.wgsl:55-86): the sample's own block is missing but one of
         * its "+1" neighbours exists. All marks are taken before any block is added (two dispatches). */
        for (int32_t i = 0; i < rig->n; i++) {
            float pf[3] = {0, 0, 0};
            int32_t b[3];
            uint32_t l[3];
            for (int k = 0; k < D; k++) pf[k] = (float)rig->world_pts[i * D + k];
            orc_block_and_local(pf, h, b, l);
            int first = 0, count = 0;
#if D == 2
            const int assoc[4][3] = {{0, 0, 0}, {0, 1, 0}, {1, 0, 0}, {1, 1, 0}};
            const int nassoc = 4;
#else
            const int assoc[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {0, 1, 1}, {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {1, 1, 1}};
            const int nassoc = 8;
#endif
            for (first = 0; first < nassoc; first++) {
                int32_t nb[3] = {b[0] + assoc[first][0], b[1] + assoc[first][1], b[2] + assoc[first][2]};
                if (hmap_find(g, nb) != ORC_NONE) break;
            }
            (void)count;
            rig->needs_block[i] = (first > 0 && first < nassoc) ? 1u : 0u;
        }
        /* touch_rigid_particle_blocks (sort.wgsl:38-52): only the sample's own block */
        for (int32_t i = 0; i < rig->n; i++) {
            if (!rig->needs_block[i]) continue;
            float pf[3] = {0, 0, 0};
            int32_t b[3];
            uint32_t l[3];
            for (int k = 0; k < D; k++) pf[k] = (float)rig->world_pts[i * D + k];
            orc_block_and_local(pf, h, b, l);
            mark_block_as_active(g, b);
        }
    }
    /* update_block_particle_count */
    for (int32_t i = 0; i < p->n; i++) {
        float pf[3];
        int32_t b[3];
        uint32_t l[3];
        pos_f32(p, i, pf);
        orc_block_and_local(pf, h, b, l);
        uint32_t id = hmap_find(g, b);
        if (id != ORC_NONE) g->num_particles[id]++;
    }
    /* copy_particles_len_to_scan_value + prefix_sum + copy_scan_values_to_first_particles */
    uint32_t *scan = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(g->n_blocks > 0 ? g->n_blocks : 1));
    for (int32_t b = 0; b < g->n_blocks; b++) scan[b] = g->num_particles[b];
    orc_prefix_sum_eval_cpu(scan, g->n_blocks);
    for (int32_t b = 0; b < g->n_blocks; b++) g->first_particle[b] = scan[b];
    /* reset (grid.wgsl:362-379) */
    size_t nn = (size_t)g->n_blocks * NPB;
    memset(g->node_mv, 0, sizeof(real) * nn * (D + 1));
    for (size_t i = 0; i < nn; i++) {
        g->node_cdf_dist[i] = 0;
        g->node_cdf_aff[i] = 0;
        g->node_cdf_closest[i] = ORC_NONE;
        g->node_head[i] = ORC_NONE;
        g->node_len[i] = 0;
    }
    /* finalize_particles_sort. Deterministic: ascending id inside a block and
     * inside a node list (reference: race order). Lists are built back to front
     * so that head is the smallest id. */
    for (int32_t i = 0; i < p->n; i++) {
        float pf[3];
        int32_t b[3];
        uint32_t l[3];
        pos_f32(p, i, pf);
        orc_block_and_local(pf, h, b, l);
        uint32_t id = hmap_find(g, b);
        if (id == ORC_NONE) continue;
        g->sorted_ids[scan[id]++] = (uint32_t)i;
    }
    for (int32_t i = p->n - 1; i >= 0; i--) {
        float pf[3];
        int32_t b[3];
        uint32_t l[3];
        pos_f32(p, i, pf);
        orc_block_and_local(pf, h, b, l);
        uint32_t id = hmap_find(g, b);
        if (id == ORC_NONE) continue;
        uint32_t node = id * NPB + node_local_index(l);
        g->particle_next[i] = g->node_head[node];
        g->node_head[node] = (uint32_t)i;
        g->node_len[node]++;
    }
    free(scan);
    if (rig) {
        /* sort_rigid_particles (sort.wgsl:139-161): per-node lists; samples whose block does not exist are
         * ignored. Built back to front so that heads are the smallest ids (reference: race order). */
        for (size_t i = 0; i < nn; i++) { rig->node_head[i] = ORC_NONE; rig->node_len[i] = 0; }
        for (int32_t i = rig->n - 1; i >= 0; i--) {
            float pf[3] = {0, 0, 0};
            int32_t b[3];
            uint32_t l[3];
            for (int k = 0; k < D; k++) pf[k] = (float)rig->world_pts[i * D + k];
            orc_block_and_local(pf, h, b, l);
            uint32_t id = hmap_find(g, b);
            rig->next[i] = ORC_NONE;
            if (id == ORC_NONE) continue;
            uint32_t node = id * NPB + node_local_index(l);
            rig->next[i] = rig->node_head[node];
            rig->node_head[node] = (uint32_t)i;
            rig->node_len[node]++;
        }
    }
}

void orc_sort(const orc_particles *p, const orc_params *prm, orc_grid *g) { sort_impl(p, prm, g, NULL); }
void orc_sort_rigid(const orc_particles *p, const orc_params *prm, orc_grid *g, orc_rigid *rig) { sort_impl(p, prm, g, rig); }

/* ------------------------------------------------------------------------ */
/* Rigid-body helpers (third party in the reference: wgrapier body.wgsl,    */
/* wgebra sim2/sim3, wgparry shapes — restated from their published maths)  */
/* ------------------------------------------------------------------------ */

/* wgrapier Body::velocity_at_point: linvel + angvel x (pt - com)
 * (call sites p2g.wgsl:208, g2p.wgsl:191,224). */
static void velocity_at_point(const orc_collider *c, const real *pt, real *out) {
    real d[3] = {0, 0, 0};
    for (int k = 0; k < D; k++) d[k] = pt[k] - c->com[k];
#if D == 2
    out[0] = c->linvel[0] + (-c->angvel[0] * d[1]);
    out[1] = c->linvel[1] + (c->angvel[0] * d[0]);
#else
    out[0] = c->linvel[0] + (c->angvel[1] * d[2] - c->angvel[2] * d[1]);
    out[1] = c->linvel[1] + (c->angvel[2] * d[0] - c->angvel[0] * d[2]);
    out[2] = c->linvel[2] + (c->angvel[0] * d[1] - c->angvel[1] * d[0]);
#endif
}

#if D == 3
static void quat_rotate(const real *q, const real *v, real *out) {
    /* v + 2 w (u x v) + 2 u x (u x v), q = (u, w) */
    real ux = q[0], uy = q[1], uz = q[2], w = q[3];
    real tx = R(2.0) * (uy * v[2] - uz * v[1]);
    real ty = R(2.0) * (uz * v[0] - ux * v[2]);
    real tz = R(2.0) * (ux * v[1] - uy * v[0]);
    out[0] = v[0] + w * tx + (uy * tz - uz * ty);
    out[1] = v[1] + w * ty + (uz * tx - ux * tz);
    out[2] = v[2] + w * tz + (ux * ty - uy * tx);
}
#endif

/* wgebra Sim::mulPt / invMulPt: p_world = rot * (p_local * scale) + trans */
static void pose_to_local(const orc_collider *c, const real *pw, real *pl) {
    real d[3] = {0, 0, 0};
    for (int k = 0; k < D; k++) d[k] = pw[k] - c->trans[k];
#if D == 2
    real cs = c->rot[0], sn = c->rot[1];
    pl[0] = (cs * d[0] + sn * d[1]) / c->scale;
    pl[1] = (-sn * d[0] + cs * d[1]) / c->scale;
#else
    real qi[4] = {-c->rot[0], -c->rot[1], -c->rot[2], c->rot[3]};
    real t[3];
    quat_rotate(qi, d, t);
    for (int k = 0; k < 3; k++) pl[k] = t[k] / c->scale;
#endif
}

static void pose_to_world(const orc_collider *c, const real *pl, real *pw) {
    real s[3] = {0, 0, 0};
    for (int k = 0; k < D; k++) s[k] = pl[k] * c->scale;
#if D == 2
    real cs = c->rot[0], sn = c->rot[1];
    pw[0] = cs * s[0] - sn * s[1] + c->trans[0];
    pw[1] = sn * s[0] + cs * s[1] + c->trans[1];
#else
    real t[3];
    quat_rotate(c->rot, s, t);
    for (int k = 0; k < 3; k++) pw[k] = t[k] + c->trans[k];
#endif
}

/* wgparry Shape::projectPointOnBoundary (call site collide.wgsl:39). Local-space
 * projection on the BOUNDARY (solid = false), parry's published algorithms:
 *  - ball: centre + r * dir (parry Ball::project_local_point)
 *  - cuboid: parry Aabb::do_project_local_point with solid = false
 *  - capsule: segment projection + ball of radius r around it */
static int project_local_on_boundary(const orc_collider *c, const real *pt, real *proj) {
    if (c->shape_type == ORC_SHAPE_BALL) {
        real r = c->shape[0];
        real n2 = 0;
        for (int k = 0; k < D; k++) n2 += pt[k] * pt[k];
        real n = r_sqrt(n2);
        int inside = n2 <= r * r;
        if (n == 0) { /* parry: any direction; use +y ("up") */
            for (int k = 0; k < D; k++) proj[k] = 0;
            proj[1] = r;
        } else {
            for (int k = 0; k < D; k++) proj[k] = pt[k] * (r / n);
        }
        return inside;
    }
    if (c->shape_type == ORC_SHAPE_CAPSULE) {
        real hh = c->shape[0], r = c->shape[1];
        real seg[3] = {0, 0, 0};
        seg[1] = r_max(-hh, r_min(hh, pt[1]));
        real d[3], n2 = 0;
        for (int k = 0; k < D; k++) { d[k] = pt[k] - seg[k]; n2 += d[k] * d[k]; }
        real n = r_sqrt(n2);
        int inside = n2 <= r * r;
        if (n == 0) {
            for (int k = 0; k < D; k++) proj[k] = seg[k];
            proj[0] += r;
        } else {
            for (int k = 0; k < D; k++) proj[k] = seg[k] + d[k] * (r / n);
        }
        return inside;
    }
    /* cuboid */
    real mins_pt[D], pt_maxs[D], shift[D];
    int inside = 1;
    for (int k = 0; k < D; k++) {
        real he = c->shape[k];
        mins_pt[k] = -he - pt[k];
        pt_maxs[k] = pt[k] - he;
        shift[k] = r_max(mins_pt[k], 0) - r_max(pt_maxs[k], 0);
        if (shift[k] != 0) inside = 0;
    }
    if (!inside) {
        for (int k = 0; k < D; k++) proj[k] = pt[k] + shift[k];
        return 0;
    }
    real best = -(sizeof(real) == 4 ? (real)3.402823466e+38F : (real)1.7976931348623157e308);
    int is_mins = 0, best_id = 0;
    for (int k = 0; k < D; k++) {
        if (mins_pt[k] < pt_maxs[k]) {
            if (pt_maxs[k] > best) { best_id = k; is_mins = 0; best = pt_maxs[k]; }
        } else if (mins_pt[k] > best) {
            best_id = k; is_mins = 1; best = mins_pt[k];
        }
    }
    for (int k = 0; k < D; k++) proj[k] = pt[k];
    proj[best_id] += is_mins ? best : -best;
    return 1;
}

/* grid/grid.wgsl:390-404 */
static void project_velocity(const real *vel, const real *n, real *out) {
    real normal_vel = 0;
    for (int k = 0; k < D; k++) normal_vel += vel[k] * n[k];
    if (normal_vel < 0) {
        const real friction = R(20.0);
        real t[D], l2 = 0;
        for (int k = 0; k < D; k++) { t[k] = vel[k] - n[k] * normal_vel; l2 += t[k] * t[k]; }
        real len = r_sqrt(l2);
        real scale = r_max(R(0.0), len + friction * normal_vel);
        for (int k = 0; k < D; k++) {
            real dir = len > R(1.0e-8) ? t[k] / len : R(0.0);
            out[k] = dir * scale;
        }
    } else {
        for (int k = 0; k < D; k++) out[k] = vel[k];
    }
}

/* grid/grid.wgsl:250-255 */
static int affinities_are_compatible(uint32_t a1, uint32_t a2) {
    uint32_t common = a1 & a2 & 0x0000ffffu;
    uint32_t s1 = (a1 >> 16) & common;
    uint32_t s2 = (a2 >> 16) & common;
    return s1 == s2;
}

/* ------------------------------------------------------------------------ */
/* Node CDF — solver/grid_update_cdf.wgsl:16-39 + collision/collide.wgsl:23-56 */
/* ------------------------------------------------------------------------ */
void orc_grid_update_cdf(const orc_params *prm, orc_grid *g) {
    const real h = prm->cell_width;
#pragma omp parallel for schedule(static)
    for (int32_t b = 0; b < g->n_blocks; b++)
        for (uint32_t t = 0; t < NPB; t++) {
            uint32_t l[3];
#if D == 2
            l[0] = t % 8; l[1] = t / 8; l[2] = 0;
#else
            l[0] = t % 4; l[1] = (t / 4) % 4; l[2] = t / 16;
#endif
            real pt[3] = {0, 0, 0};
            for (int k = 0; k < D; k++) pt[k] = (real)(g->block_vid[b * D + k] * BW + (int32_t)l[k]) * h;
            real dist = R(1.0e10);
            uint32_t aff = 0, closest = ORC_NONE;
            real cap = h * R(1.5);
            int nc = prm->n_colliders < ORC_MAX_COLLIDERS ? prm->n_colliders : ORC_MAX_COLLIDERS;
            for (int i = 0; i < nc; i++) {
                const orc_collider *c = &prm->colliders[i];
                if (c->shape_type >= ORC_SHAPE_MESH) continue; /* collide.wgsl:36-38 */
                real pl[3], projl[3], proj[3];
                pose_to_local(c, pt, pl);
                int inside = project_local_on_boundary(c, pl, projl);
                pose_to_world(c, projl, proj);
                real dpt[3], n2 = 0;
                int within = 1;
                for (int k = 0; k < D; k++) {
                    dpt[k] = proj[k] - pt[k];
                    n2 += dpt[k] * dpt[k];
                    if (!(r_abs(dpt[k]) <= cap)) within = 0;
                }
                if (inside || within) {
                    real d = r_sqrt(n2);
                    if (d < dist) closest = (uint32_t)i;
                    dist = r_min(dist, d);
                    aff |= (inside ? 0x00010001u : 0x00000001u) << i;
                }
            }
            uint32_t node = (uint32_t)b * NPB + t;
            g->node_cdf_dist[node] = dist;
            g->node_cdf_aff[node] = aff;
            g->node_cdf_closest[node] = closest;
        }
}

/* Shared tile lookup used by g2p / g2p_cdf (g2p.wgsl:72-132): node at cell
 * offset s in [0, BW+1]^D from block `b`'s origin; absent blocks read as zero. */
static uint32_t tile_node(const orc_grid *g, int32_t b, const int *s) {
    int32_t nb[3];
    uint32_t l[3] = {0, 0, 0};
    for (int k = 0; k < D; k++) {
        int o = s[k] >= BW ? 1 : 0;
        nb[k] = g->block_vid[b * D + k] + o;
        l[k] = (uint32_t)(s[k] - o * BW);
    }
    uint32_t id = hmap_find(g, nb);
    if (id == ORC_NONE) return ORC_NONE;
    return id * NPB + node_local_index(l);
}

#if D == 3
static real det4(const real m[16]) {
    /* m(r,c) = m[c*4+r]; Laplace expansion */
    real s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[9] - m[8] * m[1], s2 = m[0] * m[13] - m[12] * m[1];
    real s3 = m[4] * m[9] - m[8] * m[5], s4 = m[4] * m[13] - m[12] * m[5], s5 = m[8] * m[13] - m[12] * m[9];
    real c5 = m[10] * m[15] - m[14] * m[11], c4 = m[6] * m[15] - m[14] * m[7], c3 = m[6] * m[11] - m[10] * m[7];
    real c2 = m[2] * m[15] - m[14] * m[3], c1 = m[2] * m[11] - m[10] * m[3], c0 = m[2] * m[7] - m[6] * m[3];
    return s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
}
#endif

/* Solve M x = r for the (D+1)x(D+1) normal equations; the reference computes
 * inv(M) * r with wgebra Inv::inv3/inv4 (g2p_cdf.wgsl:236,242). Gaussian
 * elimination with partial pivoting in fp64. */
static void solve_small(const real *m, const real *r, real *x) {
    enum { N = D + 1 };
    double a[N][N + 1];
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < N; j++) a[i][j] = (double)m[j * N + i];
        a[i][N] = (double)r[i];
    }
    for (int c = 0; c < N; c++) {
        int piv = c;
        for (int i = c + 1; i < N; i++) if (fabs(a[i][c]) > fabs(a[piv][c])) piv = i;
        if (piv != c) for (int j = 0; j <= N; j++) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
        for (int i = c + 1; i < N; i++) {
            double f = a[i][c] / a[c][c];
            for (int j = c; j <= N; j++) a[i][j] -= f * a[c][j];
        }
    }
    for (int i = N - 1; i >= 0; i--) {
        double s = a[i][N];
        for (int j = i + 1; j < N; j++) s -= a[i][j] * (double)x[j];
        x[i] = (real)(s / a[i][i]);
    }
}

/* ------------------------------------------------------------------------ */
/* Particle CDF — solver/g2p_cdf.wgsl:124-250                               */
/* ------------------------------------------------------------------------ */
void orc_g2p_cdf(orc_particles *p, const orc_params *prm, const orc_grid *g) {
    const real h = prm->cell_width;
    enum { N = D + 1 };
    #pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < p->n; i++) {
        float pf[3];
        int32_t b[3];
        uint32_t l[3];
        pos_f32(p, i, pf);
        orc_block_and_local(pf, (float)h, b, l);
        uint32_t bid = hmap_find(g, b);
        if (bid == ORC_NONE) continue; /* not in any block: g2p_cdf never visits it */
        real ref[D], w[D][3];
        dir_to_assoc(p, i, h, ref);
        precompute_weights(ref, h, w);
        uint32_t prev_affinity = p->cdf_affinity[i];
        uint32_t particle_affinity = 0;
        real signs[16];
        for (int c = 0; c < 16; c++) signs[c] = 0;
        real ndist[ORC_NBH];
        uint32_t naff[ORC_NBH];
        for (int n = 0; n < ORC_NBH; n++) {
            int s[3] = {0, 0, 0};
            real weight = 1;
            for (int k = 0; k < D; k++) { s[k] = (int)l[k] + NBH_SHIFTS[n][k]; }
            weight = w[0][NBH_SHIFTS[n][0]] * w[1][NBH_SHIFTS[n][1]];
#if D == 3
            weight = weight * w[2][NBH_SHIFTS[n][2]];
#endif
            uint32_t node = tile_node(g, (int32_t)bid, s);
            ndist[n] = node == ORC_NONE ? 0 : g->node_cdf_dist[node];
            naff[n] = node == ORC_NONE ? 0 : g->node_cdf_aff[node];
            particle_affinity |= naff[n] & 0x0000ffffu;
            for (uint32_t c = 0; c < 16; c++) {
                real compatible = (naff[n] & (1u << c)) ? R(1.0) : R(0.0);
                /* shape_has_solid_interior() is constant false (g2p_cdf.wgsl:252-256) */
                real sign = ((naff[n] >> 16) & (1u << c)) ? R(-1.0) : R(1.0);
                signs[c] += compatible * weight * sign * ndist[n];
            }
        }
        for (uint32_t c = 0; c < 16; c++) {
            uint32_t mask = 1u << (c + 16);
            if ((prev_affinity & (1u << c)) == 0) {
                if (signs[c] < 0) particle_affinity |= mask;
            } else {
                particle_affinity |= prev_affinity & mask;
            }
        }
        real qtq[N * N], qtu[N];
        for (int k = 0; k < N * N; k++) qtq[k] = 0;
        for (int k = 0; k < N; k++) qtu[k] = 0;
        for (int n = 0; n < ORC_NBH; n++) {
            real pv[N];
            real weight = w[0][NBH_SHIFTS[n][0]] * w[1][NBH_SHIFTS[n][1]];
#if D == 3
            weight = weight * w[2][NBH_SHIFTS[n][2]];
#endif
            for (int k = 0; k < D; k++) pv[k] = ref[k] + (real)NBH_SHIFTS[n][k] * h;
            pv[D] = 1;
            uint32_t combined = naff[n] & particle_affinity & 0x0000ffffu;
            uint32_t sign_diff = ((naff[n] >> 16) ^ (particle_affinity >> 16)) & combined;
            if (combined != 0) {
                real dist = sign_diff == 0 ? ndist[n] : -ndist[n];
                for (int c = 0; c < N; c++)
                    for (int r = 0; r < N; r++) qtq[c * N + r] += (pv[r] * pv[c]) * weight;
                for (int r = 0; r < N; r++) qtu[r] += pv[r] * weight * dist;
            }
        }
        real det;
#if D == 2
        det = qtq[0] * (qtq[4] * qtq[8] - qtq[7] * qtq[5]) - qtq[3] * (qtq[1] * qtq[8] - qtq[7] * qtq[2]) +
              qtq[6] * (qtq[1] * qtq[5] - qtq[4] * qtq[2]);
#else
        det = det4(qtq);
#endif
        if (det > R(1.0e-8)) {
            real res[N];
            solve_small(qtq, qtu, res);
            real n2 = 0;
            for (int k = 0; k < D; k++) n2 += res[k] * res[k];
            real len = r_sqrt(n2);
            for (int k = 0; k < D; k++) {
#if D == 2
                p->cdf_normal[i * D + k] = len > R(1.0e-6) ? res[k] / len : R(0.0);
#else
                p->cdf_normal[i * D + k] = res[k] / len;
#endif
                p->cdf_rigid_vel[i * D + k] = 0;
            }
            p->cdf_dist[i] = res[D];
            p->cdf_affinity[i] = particle_affinity;
        } else {
            for (int k = 0; k < D; k++) { p->cdf_normal[i * D + k] = 0; p->cdf_rigid_vel[i * D + k] = 0; }
            p->cdf_dist[i] = 0;
            p->cdf_affinity[i] = 0;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* P2G — solver/p2g.wgsl:69-236                                             */
/* ------------------------------------------------------------------------ */
/* rigid_impulses.wgsl:50-58: fixed-point conversion. WGSL's i32(f32) truncates toward zero and
 * saturates (NaN -> 0). */
static int32_t flt2int(float f) {
    float s = f * 1e5f;
    if (!(s == s)) return 0;
    if (s >= 2147483648.0f) return INT32_MAX;
    if (s <= -2147483648.0f) return INT32_MIN;
    return (int32_t)s;
}

void orc_p2g(const orc_particles *p, const orc_params *prm, orc_grid *g) {
    const real h = prm->cell_width;
    /* the accumulators are zeroed by their consumer (rigid_impulses.wgsl:104-109); orc_step, which has
     * no bodies, clears them itself */
    /* blocks are independent (per-node gather); the only shared state is the integer impulse sum */
#pragma omp parallel for schedule(dynamic, 4)
    for (int32_t b = 0; b < g->n_blocks; b++) {
        for (uint32_t t = 0; t < NPB; t++) {
            int tl[3] = {0, 0, 0};
#if D == 2
            tl[0] = (int)(t % 8); tl[1] = (int)(t / 8);
#else
            tl[0] = (int)(t % 4); tl[1] = (int)((t / 4) % 4); tl[2] = (int)(t / 16);
#endif
            uint32_t gid = (uint32_t)b * NPB + t;
            uint32_t node_aff = g->node_cdf_aff[gid];
            uint32_t collider_id = g->node_cdf_closest[gid];
            /* The 27 (9) cells whose particles reach this node, in NBH order, and
             * the cursor of each cell's particle list (p2g.wgsl:341-396). */
            uint32_t cursor[ORC_NBH];
            uint32_t maxlen = 0;
            for (int n = 0; n < ORC_NBH; n++) {
                int32_t cb[3];
                uint32_t cl[3] = {0, 0, 0};
                for (int k = 0; k < D; k++) {
                    int rel = tl[k] + NBH_SHIFTS[n][k] - 2; /* in [-2, BW-1] */
                    int o = rel < 0 ? -1 : 0;
                    cb[k] = g->block_vid[b * D + k] + o;
                    cl[k] = (uint32_t)(rel - o * BW);
                }
                uint32_t id = hmap_find(g, cb);
                if (id == ORC_NONE) { cursor[n] = ORC_NONE; continue; }
                uint32_t node = id * NPB + node_local_index(cl);
                cursor[n] = g->node_head[node];
                if (g->node_len[node] > maxlen) maxlen = g->node_len[node];
            }
            real total[D + 1];
            real imp[D], ang[ORC_ANG];
            for (int k = 0; k <= D; k++) total[k] = 0;
            for (int k = 0; k < D; k++) imp[k] = 0;
            for (int k = 0; k < ORC_ANG; k++) ang[k] = 0;
            for (uint32_t it = 0; it < maxlen; it++) {
                real part[D + 1];
                real pimp[D], pang[ORC_ANG];
                for (int k = 0; k <= D; k++) part[k] = 0;
                for (int k = 0; k < D; k++) pimp[k] = 0;
                for (int k = 0; k < ORC_ANG; k++) pang[k] = 0;
                for (int n = 0; n < ORC_NBH; n++) {
                    uint32_t pid = cursor[n];
                    if (pid == ORC_NONE) continue; /* zero slot: contributes exactly 0 */
                    cursor[n] = g->particle_next[pid];
                    real ref[D], w[D][3], dpt[D], mom[D], ad[D];
                    dir_to_assoc(p, (int)pid, h, ref);
                    precompute_weights(ref, h, w);
                    int sh[3] = {0, 0, 0};
                    for (int k = 0; k < D; k++) sh[k] = 2 - NBH_SHIFTS[n][k];
                    real mass = p->mass[pid];
                    for (int k = 0; k < D; k++) {
                        mom[k] = p->vel[pid * D + k] * mass;
                        dpt[k] = ref[k] + (real)sh[k] * h;
                    }
                    real weight = w[0][sh[0]] * w[1][sh[1]];
#if D == 3
                    weight = weight * w[2][sh[2]];
#endif
                    uint32_t paff = p->cdf_affinity[pid];
                    if (!affinities_are_compatible(node_aff, paff)) {
                        if (collider_id != ORC_NONE && (int)collider_id < prm->n_colliders) {
                            const orc_collider *c = &prm->colliders[collider_id];
                            real cc[3] = {0, 0, 0}, bv[3], rel[D], proj[D], ghost[D], dimp[D];
                            for (int k = 0; k < D; k++) cc[k] = dpt[k] + p->pos[pid * D + k];
                            velocity_at_point(c, cc, bv);
                            for (int k = 0; k < D; k++) rel[k] = p->vel[pid * D + k] - bv[k];
                            project_velocity(rel, &p->cdf_normal[pid * D], proj);
                            for (int k = 0; k < D; k++) {
                                ghost[k] = bv[k] + proj[k];
                                dimp[k] = (p->vel[pid * D + k] - ghost[k]) * (weight * mass);
                            }
                            real lever[3] = {0, 0, 0};
                            for (int k = 0; k < D; k++) lever[k] = c->com[k] - cc[k];
#if D == 2
                            pang[0] += dimp[0] * lever[1] + dimp[1] * (-lever[0]);
#else
                            pang[0] += dimp[1] * lever[2] - dimp[2] * lever[1];
                            pang[1] += dimp[2] * lever[0] - dimp[0] * lever[2];
                            pang[2] += dimp[0] * lever[1] - dimp[1] * lever[0];
#endif
                            for (int k = 0; k < D; k++) pimp[k] += dimp[k];
                        }
                        continue;
                    }
                    mat_vec(&p->affine[pid * DD], dpt, ad);
                    for (int k = 0; k < D; k++) part[k] += (ad[k] + mom[k]) * weight;
                    part[D] += mass * weight;
                }
                for (int k = 0; k <= D; k++) total[k] += part[k];
                for (int k = 0; k < D; k++) imp[k] += pimp[k];
                for (int k = 0; k < ORC_ANG; k++) ang[k] += pang[k];
            }
            for (int k = 0; k <= D; k++) g->node_mv[gid * (D + 1) + k] = total[k];
            if (collider_id != ORC_NONE && collider_id < 16) {
                /* rigid_impulses.wgsl:52-54 flt2int: i32(flt * 1e5), accumulated atomically */
                for (int k = 0; k < D; k++)
                {
                    const int32_t v = flt2int((float)imp[k]);
#pragma omp atomic
                    g->impulses[collider_id * (D + ORC_ANG) + k] += v;
                }
                for (int k = 0; k < ORC_ANG; k++)
                {
                    const int32_t v = flt2int((float)ang[k]);
#pragma omp atomic
                    g->impulses[collider_id * (D + ORC_ANG) + D + k] += v;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Grid update — solver/grid_update.wgsl:20-64                              */
/* ------------------------------------------------------------------------ */
void orc_grid_update(const orc_params *prm, orc_grid *g) {
    const real dt = prm->dt, h = prm->cell_width;
    size_t nn = (size_t)g->n_blocks * NPB;
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < nn; i++) {
        real *mv = &g->node_mv[i * (D + 1)];
        real mass = mv[D];
        real inv_mass = mass > 0 ? R(1.0) / mass : R(0.0);
        real lim = h / dt;
        for (int k = 0; k < D; k++) {
            real v = (mv[k] + mass * prm->gravity[k] * dt) * inv_mass;
            v = r_min(r_max(v, -lim), lim); /* WGSL clamp = min(max(x, lo), hi) */
            mv[k] = v;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* G2P — solver/g2p.wgsl:134-238                                            */
/* ------------------------------------------------------------------------ */
void orc_g2p(orc_particles *p, const orc_params *prm, const orc_grid *g) {
    const real h = prm->cell_width;
    const real invd = inv_d(h);
    #pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < p->n; i++) {
        float pf[3];
        int32_t b[3];
        uint32_t l[3];
        pos_f32(p, i, pf);
        orc_block_and_local(pf, (float)h, b, l);
        uint32_t bid = hmap_find(g, b);
        if (bid == ORC_NONE) continue;
        real ref[D], w[D][3];
        dir_to_assoc(p, i, h, ref);
        precompute_weights(ref, h, w);
        real acc[D + 1], grad[DD], rigid_vel[D];
        for (int k = 0; k <= D; k++) acc[k] = 0;
        for (int k = 0; k < DD; k++) grad[k] = 0;
        for (int k = 0; k < D; k++) rigid_vel[k] = 0;
        const real *pvel = &p->vel[i * D];
        uint32_t paff = p->cdf_affinity[i];
        for (int n = 0; n < ORC_NBH; n++) {
            int s[3] = {0, 0, 0};
            real dpt[D], cell[D + 1];
            for (int k = 0; k < D; k++) {
                s[k] = (int)l[k] + NBH_SHIFTS[n][k];
                dpt[k] = ref[k] + (real)NBH_SHIFTS[n][k] * h;
            }
            uint32_t node = tile_node(g, (int32_t)bid, s);
            uint32_t naff = 0, nclosest = ORC_NONE;
            for (int k = 0; k <= D; k++) cell[k] = node == ORC_NONE ? R(0.0) : g->node_mv[node * (D + 1) + k];
            if (node != ORC_NONE) { naff = g->node_cdf_aff[node]; nclosest = g->node_cdf_closest[node]; }
            if (!affinities_are_compatible(paff, naff)) {
                if (nclosest != ORC_NONE && (int)nclosest < prm->n_colliders) {
                    const orc_collider *c = &prm->colliders[nclosest];
                    real cc[3] = {0, 0, 0}, bv[3], rel[D], proj[D];
                    for (int k = 0; k < D; k++) cc[k] = dpt[k] + p->pos[i * D + k];
                    velocity_at_point(c, cc, bv);
                    for (int k = 0; k < D; k++) rel[k] = pvel[k] - bv[k];
                    project_velocity(rel, &p->cdf_normal[i * D], proj);
                    for (int k = 0; k < D; k++) cell[k] = bv[k] + proj[k];
                } else {
                    for (int k = 0; k < D; k++) cell[k] = pvel[k];
                }
            }
            real weight = w[0][NBH_SHIFTS[n][0]] * w[1][NBH_SHIFTS[n][1]];
#if D == 3
            weight = weight * w[2][NBH_SHIFTS[n][2]];
#endif
            for (int k = 0; k <= D; k++) acc[k] += cell[k] * weight;
            real wi = weight * invd;
            for (int c = 0; c < D; c++)
                for (int r = 0; r < D; r++) grad[c * D + r] += wi * (cell[r] * dpt[c]);
        }
        /* g2p.wgsl:220-226. B9: the reference loops over 16 slots relying on robust
         * buffer access; bounded by the real collider count here. */
        for (int c = 0; c < 16 && c < prm->n_colliders; c++) {
            if (paff & (1u << c)) {
                real pt[3] = {0, 0, 0}, bv[3];
                for (int k = 0; k < D; k++) pt[k] = p->pos[i * D + k];
                velocity_at_point(&prm->colliders[c], pt, bv);
                for (int k = 0; k < D; k++) rigid_vel[k] += bv[k];
            }
        }
        for (int k = 0; k < D; k++) {
            p->cdf_rigid_vel[i * D + k] = rigid_vel[k];
            p->vel[i * D + k] = acc[k];
        }
        for (int k = 0; k < DD; k++) p->affine[i * DD + k] = grad[k];
    }
}

/* ------------------------------------------------------------------------ */
/* Particle update — solver/particle_update.wgsl:45-141                     */
/* ------------------------------------------------------------------------ */
void orc_particle_update(orc_particles *p, const orc_params *prm) {
    const real dt = prm->dt, h = prm->cell_width;
    #pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < p->n; i++) {
        real vel[D], nrm[D], rv[D];
        for (int k = 0; k < D; k++) {
            vel[k] = p->vel[i * D + k];
            nrm[k] = p->cdf_normal[i * D + k];
            rv[k] = p->cdf_rigid_vel[i * D + k];
        }
        real sd = p->cdf_dist[i];
        if (sd < R(-0.05) * h) {
            real rel[D], proj[D];
            for (int k = 0; k < D; k++) rel[k] = vel[k] - rv[k];
            project_velocity(rel, nrm, proj);
            for (int k = 0; k < D; k++) vel[k] = rv[k] + proj[k];
        }
        real l2 = 0;
        for (int k = 0; k < D; k++) l2 += vel[k] * vel[k];
        real len = r_sqrt(l2);
        if (len > h / dt)
            for (int k = 0; k < D; k++) vel[k] = vel[k] / len * h / dt;
        for (int k = 0; k < D; k++) p->pos[i * D + k] = p->pos[i * D + k] + vel[k] * dt;
        if (sd < R(-0.05) * h) {
            real corrected = r_max(sd, R(-0.3) * h);
            real imp = dt * -corrected * R(1.0e3);
            for (int k = 0; k < D; k++) vel[k] += imp * nrm[k];
        }
        /* F <- F + (grad_v * dt) * F ; the velocity gradient sits in `affine`. */
        real *F = &p->def_grad[i * DD];
        real *A = &p->affine[i * DD];
        real adt[DD], prod[DD], newF[DD];
        for (int k = 0; k < DD; k++) adt[k] = A[k] * dt;
        mat_mul(adt, F, prod);
        for (int k = 0; k < DD; k++) newF[k] = F[k] + prod[k];

        real phase = p->phase[i * 2 + 0];
        real max_stretch = p->phase[i * 2 + 1];
        if (phase > 0 && max_stretch > 0) {
            real u[DD], s[D], vt[DD];
            orc_svd(newF, u, s, vt);
            int broken = 0;
            for (int k = 0; k < D; k++) if (s[k] > max_stretch) broken = 1;
            if (broken) { p->phase[i * 2 + 0] = 0; phase = 0; }
        }
        if (phase == 0) orc_drucker_prager_project(&p->dp[i * 6], &p->dp_state[i * 3], newF);
        real tau[DD];
        orc_kirchoff_stress(prm->model, p->lambda[i], p->mu[i], newF, tau);
        real coeff = p->init_volume[i] * inv_d(h) * dt;
        for (int k = 0; k < DD; k++) {
            real a = A[k] * p->mass[i] - tau[k] * coeff;
            F[k] = newF[k];
            A[k] = a;
        }
        for (int k = 0; k < D; k++) p->vel[i * D + k] = vel[k];
    }
}

/* One substep = pipeline.rs:201-280 restricted to the MPM passes:
 * sort, grid_update_cdf, g2p_cdf, p2g, grid_update, g2p, particles_update. */
void orc_step(orc_particles *p, const orc_params *prm, orc_grid *g, int n_substeps) {
    for (int s = 0; s < n_substeps; s++) {
        orc_sort(p, prm, g);
        /* Queued unconditionally by the reference, also with zero colliders
         * (then every particle cdf is reset to default_cdf()). */
        orc_grid_update_cdf(prm, g);
        orc_g2p_cdf(p, prm, g);
        for (int k = 0; k < 16 * (D + ORC_ANG); k++) g->impulses[k] = 0;
        orc_p2g(p, prm, g);
        orc_grid_update(prm, g);
        orc_g2p(p, prm, g);
        orc_particle_update(p, prm);
    }
}

/* ------------------------------------------------------------------------ */
/* Rigid bodies: two-way coupling — solver/rigid_impulses.wgsl              */
/* Body::applyImpulse / integrateVelocity / updateMprops are third party     */
/* (wgrapier, not on disk): restated from rapier's published algorithms      */
/* (RigidBodyVelocity::integrate, MassProperties world-space update).        */
/* ------------------------------------------------------------------------ */

/* rigid_impulses.wgsl:138-149 update_world_mass_properties: world com = pose * local com,
 * world inverse inertia = R I^-1 R^T. */
void orc_update_world_mass_properties(orc_collider *cols, orc_body *bodies, int n) {
    for (int i = 0; i < n; i++) {
        orc_collider *c = &cols[i];
        orc_body *b = &bodies[i];
        real com[3] = {0, 0, 0};
        pose_to_world(c, b->local_com, com);
        for (int k = 0; k < D; k++) c->com[k] = com[k];
#if D == 2
        b->inv_inertia_world[0] = b->inv_inertia_local[0];
#else
        /* columns of R = images of the basis vectors */
        real rm[9], tmp[9];
        for (int col = 0; col < 3; col++) {
            real e[3] = {0, 0, 0}, o[3];
            e[col] = 1;
            quat_rotate(c->rot, e, o);
            for (int r = 0; r < 3; r++) rm[col * 3 + r] = o[r];
        }
        /* tmp = R * Il ; world = tmp * R^T   (column-major, element (r,c) at [c*3+r]) */
        for (int cc = 0; cc < 3; cc++)
            for (int r = 0; r < 3; r++) {
                real s = 0;
                for (int k = 0; k < 3; k++) s += rm[k * 3 + r] * b->inv_inertia_local[cc * 3 + k];
                tmp[cc * 3 + r] = s;
            }
        for (int cc = 0; cc < 3; cc++)
            for (int r = 0; r < 3; r++) {
                real s = 0;
                for (int k = 0; k < 3; k++) s += tmp[k * 3 + r] * rm[k * 3 + cc];
                b->inv_inertia_world[cc * 3 + r] = s;
            }
#endif
    }
}

/* rigid_impulses.wgsl:95-136 update: apply the accumulated impulse, cap the velocities, integrate the
 * pose, then apply gravity to bodies with a non-zero inverse mass. */
void orc_integrate_bodies(const orc_params *prm, orc_collider *cols, const orc_body *bodies, int n, int32_t *impulses) {
    const real dt = prm->dt, h = prm->cell_width;
    for (int i = 0; i < n && i < 16; i++) {
        orc_collider *c = &cols[i];
        const orc_body *b = &bodies[i];
        int32_t *acc = &impulses[i * (D + ORC_ANG)];
        real lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
        for (int k = 0; k < D; k++) lin[k] = (real)((float)acc[k] / 1e5f);
        for (int k = 0; k < ORC_ANG; k++) ang[k] = (real)((float)acc[D + k] / 1e5f);
        for (int k = 0; k < D + ORC_ANG; k++) acc[k] = 0;
        /* Body::applyImpulse */
        real nl[3] = {0, 0, 0}, na[3] = {0, 0, 0};
        for (int k = 0; k < D; k++) nl[k] = c->linvel[k] + b->inv_mass[k] * lin[k];
#if D == 2
        na[0] = c->angvel[0] + b->inv_inertia_world[0] * ang[0];
#else
        for (int r = 0; r < 3; r++) {
            real s = 0;
            for (int k = 0; k < 3; k++) s += b->inv_inertia_world[k * 3 + r] * ang[k];
            na[r] = c->angvel[r] + s;
        }
#endif
        real ln2 = 0, an2 = 0, il2 = 0, ia2 = 0;
        for (int k = 0; k < D; k++) { ln2 += nl[k] * nl[k]; il2 += lin[k] * lin[k]; }
        for (int k = 0; k < ORC_ANG; k++) { an2 += na[k] * na[k]; ia2 += ang[k] * ang[k]; }
        real lnorm = r_sqrt(ln2), anorm = r_sqrt(an2);
        real lin_limit = R(0.1) * h / dt, ang_limit = R(1.0);
        if (r_sqrt(il2) != 0 || r_sqrt(ia2) != 0) {
            if (lnorm > lin_limit)
                for (int k = 0; k < D; k++) nl[k] = nl[k] * (lin_limit / lnorm);
            if (anorm > ang_limit)
                for (int k = 0; k < ORC_ANG; k++) na[k] = na[k] * (ang_limit / anorm);
        }
        /* Body::integrateVelocity (rapier RigidBodyVelocity::integrate): rotate about the world centre of
         * mass by exp(angvel dt), translate by linvel dt, renormalise the rotation. */
        real comw[3] = {0, 0, 0}, arm[3] = {0, 0, 0}, rarm[3] = {0, 0, 0};
        pose_to_world(c, b->local_com, comw);
        for (int k = 0; k < D; k++) arm[k] = c->trans[k] - comw[k];
#if D == 2
        real a = na[0] * dt;
        real ca = (real)cos((double)a), sa = (real)sin((double)a);
        rarm[0] = ca * arm[0] - sa * arm[1];
        rarm[1] = sa * arm[0] + ca * arm[1];
        real nc = ca * c->rot[0] - sa * c->rot[1], ns = sa * c->rot[0] + ca * c->rot[1];
        real nn = r_sqrt(nc * nc + ns * ns);
        c->rot[0] = nc / nn;
        c->rot[1] = ns / nn;
#else
        real ax[3] = {na[0] * dt, na[1] * dt, na[2] * dt};
        real angle = r_sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        real dq[4] = {0, 0, 0, 1};
        if (angle != 0) {
            real s = (real)sin((double)angle * 0.5) / angle;
            dq[0] = ax[0] * s; dq[1] = ax[1] * s; dq[2] = ax[2] * s;
            dq[3] = (real)cos((double)angle * 0.5);
        }
        quat_rotate(dq, arm, rarm);
        /* dq * rot (Hamilton product, (i,j,k,w) storage) */
        const real *q = c->rot;
        real nq[4];
        nq[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
        nq[1] = dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0];
        nq[2] = dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3];
        nq[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
        real nn = r_sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
        for (int k = 0; k < 4; k++) c->rot[k] = nq[k] / nn;
#endif
        for (int k = 0; k < D; k++) c->trans[k] = comw[k] + rarm[k] + nl[k] * dt;
        /* gravity on dynamic axes only (rigid_impulses.wgsl:130-131) */
        for (int k = 0; k < D; k++) c->linvel[k] = nl[k] + (b->inv_mass[k] != 0 ? prm->gravity[k] * dt : 0);
        for (int k = 0; k < ORC_ANG; k++) c->angvel[k] = na[k];
    }
}

/* pipeline.rs:201-280 with bodies: world mass properties first, body integration last. */
void orc_step_bodies(orc_particles *p, const orc_params *prm, orc_grid *g, orc_collider *cols, orc_body *bodies,
                     int n_substeps) {
    for (int k = 0; k < 16 * (D + ORC_ANG); k++) g->impulses[k] = 0;
    for (int s = 0; s < n_substeps; s++) {
        orc_update_world_mass_properties(cols, bodies, prm->n_colliders);
        orc_sort(p, prm, g);
        orc_grid_update_cdf(prm, g);
        orc_g2p_cdf(p, prm, g);
        orc_p2g(p, prm, g);
        orc_grid_update(prm, g);
        orc_g2p(p, prm, g);
        orc_particle_update(p, prm);
        orc_integrate_bodies(prm, cols, bodies, prm->n_colliders, g->impulses);
    }
}

/* ------------------------------------------------------------------------ */
/* Rigid particles of mesh colliders                                         */
/* ------------------------------------------------------------------------ */

/* rigid_particle_update.wgsl:26-49: world = pose * local for the samples and for the mesh vertices */
void orc_update_rigid_particles(const orc_params *prm, orc_rigid *rig) {
    for (int32_t i = 0; i < rig->n; i++) {
        const orc_collider *c = &prm->colliders[rig->ids[i * 4 + 3]];
        real w[3] = {0, 0, 0}, l[3] = {0, 0, 0};
        for (int k = 0; k < D; k++) l[k] = rig->local_pts[i * D + k];
        pose_to_world(c, l, w);
        for (int k = 0; k < D; k++) rig->world_pts[i * D + k] = w[k];
    }
    for (int32_t i = 0; i < rig->nv; i++) {
        const orc_collider *c = &prm->colliders[rig->vtx_collider[i]];
        real w[3] = {0, 0, 0}, l[3] = {0, 0, 0};
        for (int k = 0; k < D; k++) l[k] = rig->local_vtx[i * D + k];
        pose_to_world(c, l, w);
        for (int k = 0; k < D; k++) rig->world_vtx[i * D + k] = w[k];
    }
}

/* p2g_cdf.wgsl:52-190: every node gathers the rigid particles of the 27 (9) cells below it and projects
 * itself on their primitive; a valid projection sets the collider's affinity bit, ORs its sign bit and competes
 * for the closest collider. Equal distances: the smaller collider id wins (reference: list order). */
void orc_p2g_cdf(const orc_params *prm, orc_grid *g, const orc_rigid *rig) {
    const real h = prm->cell_width;
    for (int32_t b = 0; b < g->n_blocks; b++)
        for (uint32_t t = 0; t < NPB; t++) {
            int tl[3] = {0, 0, 0};
#if D == 2
            tl[0] = (int)(t % 8); tl[1] = (int)(t / 8);
#else
            tl[0] = (int)(t % 4); tl[1] = (int)((t / 4) % 4); tl[2] = (int)(t / 16);
#endif
            uint32_t gid = (uint32_t)b * NPB + t;
            real cell[3] = {0, 0, 0};
            for (int k = 0; k < D; k++) cell[k] = (real)(g->block_vid[b * D + k] * BW + tl[k]) * h;
            real ndist = g->node_cdf_dist[gid];
            uint32_t naff = g->node_cdf_aff[gid], nclosest = g->node_cdf_closest[gid];
            for (int n = 0; n < ORC_NBH; n++) {
                int32_t cb[3];
                uint32_t cl[3] = {0, 0, 0};
                for (int k = 0; k < D; k++) {
                    int rel = tl[k] + NBH_SHIFTS[n][k] - 2;
                    int o = rel < 0 ? -1 : 0;
                    cb[k] = g->block_vid[b * D + k] + o;
                    cl[k] = (uint32_t)(rel - o * BW);
                }
                uint32_t id = hmap_find(g, cb);
                if (id == ORC_NONE) continue;
                uint32_t node = id * NPB + node_local_index(cl);
                for (uint32_t rp = rig->node_head[node]; rp != ORC_NONE; rp = rig->next[rp]) {
                    const uint32_t *ids = &rig->ids[rp * 4];
                    uint32_t col = ids[3];
                    const real *va = &rig->world_vtx[ids[0] * D], *vb = &rig->world_vtx[ids[1] * D];
                    int valid = 0, sign = 0;
                    real dist = 0;
#if D == 2
                    /* wgparry Segment::projectLocalPoint (third party): clamp of the orthogonal projection */
                    real ab[2] = {vb[0] - va[0], vb[1] - va[1]}, ap[2] = {cell[0] - va[0], cell[1] - va[1]};
                    real den = ab[0] * ab[0] + ab[1] * ab[1];
                    real tt = den > 0 ? (ap[0] * ab[0] + ap[1] * ab[1]) / den : 0;
                    real proj[2];
                    if (tt <= 0) { proj[0] = va[0]; proj[1] = va[1]; }
                    else if (tt >= 1) { proj[0] = vb[0]; proj[1] = vb[1]; }
                    else { proj[0] = va[0] + ab[0] * tt; proj[1] = va[1] + ab[1] * tt; }
                    if ((proj[0] != va[0] || proj[1] != va[1]) && (proj[0] != vb[0] || proj[1] != vb[1])) {
                        real dp[2] = {cell[0] - proj[0], cell[1] - proj[1]};
                        valid = 1;
                        dist = r_sqrt(dp[0] * dp[0] + dp[1] * dp[1]);
                        sign = (dp[0] * (-ab[1]) + dp[1] * ab[0]) < 0;
                    }
#else
                    const real *vc = &rig->world_vtx[ids[2] * D];
                    real ap[3], bp[3], cp[3], ab[3], ac[3], bc[3], nrm[3], t1[3];
                    for (int k = 0; k < 3; k++) {
                        ap[k] = cell[k] - va[k]; bp[k] = cell[k] - vb[k]; cp[k] = cell[k] - vc[k];
                        ab[k] = vb[k] - va[k]; ac[k] = vc[k] - va[k]; bc[k] = vc[k] - vb[k];
                    }
#define CROSS(o, x, y) do { o[0] = x[1] * y[2] - x[2] * y[1]; o[1] = x[2] * y[0] - x[0] * y[2]; o[2] = x[0] * y[1] - x[1] * y[0]; } while (0)
#define DOT(x, y) (x[0] * y[0] + x[1] * y[1] + x[2] * y[2])
                    CROSS(nrm, ab, ac);
                    real nlen = r_sqrt(DOT(nrm, nrm));
                    if (nlen != 0) {
                        real d1, d2, d3;
                        CROSS(t1, ab, nrm); d1 = DOT(t1, ap);
                        CROSS(t1, bc, nrm); d2 = DOT(t1, bp);
                        CROSS(t1, ac, nrm); d3 = DOT(t1, cp);
                        if (d1 <= 0 && d2 <= 0 && d3 >= 0) {
                            real sd = DOT(nrm, ap) / nlen;
                            valid = 1;
                            sign = sd < 0;
                            dist = r_abs(sd);
                        }
                    }
#undef CROSS
#undef DOT
#endif
                    if (!valid || col >= 16) continue;
                    naff |= (1u << col) | ((uint32_t)sign << (col + 16));
                    if (dist < ndist || (dist == ndist && col < nclosest)) {
                        ndist = dist;
                        nclosest = col;
                    }
                }
            }
            g->node_cdf_dist[gid] = ndist;
            g->node_cdf_aff[gid] = naff;
            g->node_cdf_closest[gid] = nclosest;
        }
}

/* pipeline.rs:201-280, every pass */
void orc_step_full(orc_particles *p, const orc_params *prm, orc_grid *g, orc_collider *cols, orc_body *bodies,
                   orc_rigid *rig, int move_bodies, int n_substeps) {
    for (int k = 0; k < 16 * (D + ORC_ANG); k++) g->impulses[k] = 0;
    for (int s = 0; s < n_substeps; s++) {
        if (move_bodies) orc_update_world_mass_properties(cols, bodies, prm->n_colliders);
        if (rig) orc_update_rigid_particles(prm, rig);
        sort_impl(p, prm, g, rig);
        orc_grid_update_cdf(prm, g);
        if (rig) orc_p2g_cdf(prm, g, rig);
        orc_g2p_cdf(p, prm, g);
        orc_p2g(p, prm, g);
        orc_grid_update(prm, g);
        orc_g2p(p, prm, g);
        orc_particle_update(p, prm);
        /* bodies that can neither move nor be pushed: the pass is the identity (the HIP path skips it too) */
        if (move_bodies) orc_integrate_bodies(prm, cols, bodies, prm->n_colliders, g->impulses);
        else for (int k = 0; k < 16 * (D + ORC_ANG); k++) g->impulses[k] = 0;
    }
}
