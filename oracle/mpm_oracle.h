/*
 * mpm_oracle.h — CPU restatement of the wgsparkl MLS-MPM substep.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (wgsparkl_amd/, the
 * C-ABI library) may include, link or call this. It is the checker used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY PINNING: the only numeric assertion in the reference's own tests is
 * the exclusive prefix sum (src/grid/prefix_sum.rs:170-231); this oracle is
 * pinned against that (tests/test_oracle_golden.py). Every other function is
 * "parity unpinned": the reference cannot be built here (no cargo/rustc, no
 * Vulkan, third-party wgebra/wgparry/wgrapier sources absent), so the oracle is
 * a restatement written from the WGSL files cited per function, cross-checked
 * against an independent numpy restatement (oracle/np_oracle.py) and against
 * physical invariants.
 *
 * Compile-time knobs:  -DORC_DIM=2|3   -DORC_REAL=float|double
 *   ORC_REAL=float  : follows the WGSL fp32 operation order (build with
 *                     -ffp-contract=off so no FMA contraction happens).
 *   ORC_REAL=double : same algorithm with fp64 arithmetic ("truth" used to
 *                     state tolerances). Cell/block indices are always computed
 *                     from the fp32-rounded position with fp32 arithmetic.
 */
#ifndef MPM_ORACLE_H
#define MPM_ORACLE_H

#include <stdint.h>

#ifndef ORC_DIM
#define ORC_DIM 3
#endif
#ifndef ORC_REAL
#define ORC_REAL float
#endif

typedef ORC_REAL real;

#define ORC_NONE 0xffffffffu
#define ORC_D ORC_DIM
#define ORC_DD (ORC_DIM * ORC_DIM)
#if ORC_DIM == 2
#define ORC_ANG 1
#define ORC_BLOCK_W 8 /* grid.wgsl:278 */
#define ORC_NBH 9
#else
#define ORC_ANG 3
#define ORC_BLOCK_W 4 /* grid.wgsl:286 */
#define ORC_NBH 27
#endif
#define ORC_NODES_PER_BLOCK 64 /* grid.wgsl:43 */
#define ORC_MAX_COLLIDERS 16   /* grid.wgsl:230-240 */

/* Constitutive model selector. The reference has no runtime switch
 * (particle_update.wgsl:7-8 picks linear_elasticity at compile time). */
enum { ORC_MODEL_COROTATED = 0, ORC_MODEL_NEO_HOOKEAN = 1 };

/* Analytic collider shapes handled by collide() (collision/collide.wgsl:23-56). */
enum { ORC_SHAPE_BALL = 0, ORC_SHAPE_CUBOID = 1, ORC_SHAPE_CAPSULE = 2,
       ORC_SHAPE_MESH = 3 /* trimesh / heightfield / polyline: no analytic projection (collide.wgsl:36-38), rigid particles instead */ };

/* Particle state, structure-of-arrays. Matrices are column-major like WGSL
 * matNxN / nalgebra (element (r,c) at [c*D + r]).
 * Field meaning follows particle3d.wgsl:7-25 / particle2d.wgsl:7-27,
 * models/drucker_prager.wgsl:8-23, particle_update.wgsl:40-43. */
typedef struct {
    int32_t n;
    real *pos;           /* n*D */
    real *vel;           /* n*D */
    real *def_grad;      /* n*D*D */
    real *affine;        /* n*D*D */
    real *cdf_normal;    /* n*D */
    real *cdf_rigid_vel; /* n*D */
    real *cdf_dist;      /* n */
    uint32_t *cdf_affinity; /* n */
    real *init_volume;   /* n */
    real *mass;          /* n */
    real *lambda;        /* n  ElasticCoefficients.lambda */
    real *mu;            /* n */
    real *dp;            /* n*6 (h0,h1,h2,h3,lambda,mu) */
    real *dp_state;      /* n*3 (plastic det, hardening, log_vol_gain) */
    real *phase;         /* n*2 (phase, max_stretch) */
} orc_particles;

/* One coupled collider = shape + pose + body velocity + body centre of mass
 * (wgrapier GpuBodySet as seen by collide.wgsl / g2p.wgsl / p2g.wgsl). */
typedef struct {
    int32_t shape_type;
    real shape[4];       /* ball: r; cuboid: half extents; capsule: half_height(y axis), r */
    real rot[4];         /* 3D: unit quaternion (i,j,k,w); 2D: (cos, sin, -, -) */
    real trans[3];
    real scale;
    real linvel[3];
    real angvel[3];      /* 2D: angvel[0] */
    real com[3];         /* world-space centre of mass */
} orc_collider;

/* Mass properties of the rigid body a collider is attached to (wgrapier
 * GpuBodySet local_mprops / mprops as used by rigid_impulses.wgsl:81-84).
 * A kinematic or fixed body has inv_mass = 0 and inv_inertia_local = 0. */
typedef struct {
    real inv_mass[3];           /* per-axis inverse mass (rapier: effective_inv_mass) */
    real inv_inertia_local[9];  /* 3D: column-major inverse inertia tensor in the body frame; 2D: [0] */
    real local_com[3];          /* centre of mass in the body frame */
    real inv_inertia_world[9];  /* out: R inv_inertia_local R^T (3D) / copy (2D), by orc_update_world_mass_properties */
} orc_body;

typedef struct {
    real gravity[3];
    real dt;
    real cell_width;
    int32_t model;       /* ORC_MODEL_* */
    int32_t n_colliders;
    const orc_collider *colliders;
} orc_params;

/* Sparse block grid produced by the sort (grid.wgsl:215-228, grid.rs:209-264).
 * Caller allocates with capacity `cap_blocks`. */
typedef struct {
    int32_t cap_blocks;
    int32_t hmap_capacity;     /* power of two, == next_pow2(cap) like grid.rs:283 */
    int32_t n_blocks;          /* out: num_active_blocks */
    int32_t overflow;          /* out: 1 if an insert failed (reference drops silently) */
    uint32_t *hmap_state;      /* hmap_capacity : packed key or NONE */
    uint32_t *hmap_value;      /* hmap_capacity : block header id */
    int32_t *block_vid;        /* cap*D   virtual id */
    uint32_t *first_particle;  /* cap */
    uint32_t *num_particles;   /* cap */
    uint32_t *sorted_ids;      /* n */
    uint32_t *node_head;       /* cap*64 : first particle of node list or NONE */
    uint32_t *node_len;        /* cap*64 */
    uint32_t *particle_next;   /* n */
    real *node_mv;             /* cap*64*(D+1): momentum|velocity, mass */
    real *node_cdf_dist;       /* cap*64 */
    uint32_t *node_cdf_aff;    /* cap*64 */
    uint32_t *node_cdf_closest;/* cap*64 */
    int32_t *impulses;         /* 16*(D+ANG): fixed-point (x1e5) linear+angular impulses accumulated by p2g */
} orc_grid;

/* Rigid particles of the mesh colliders (GpuRigidParticles, src/solver/particle3d.rs:83-150, 2D
 * particle2d.rs:60-125): sample points in the collider's local frame with the primitive (triangle / segment)
 * they were sampled from, and the mesh vertices. Caller allocates everything. */
typedef struct {
    int32_t n;                 /* samples */
    const real *local_pts;     /* n*D */
    real *world_pts;           /* n*D, out of orc_update_rigid_particles */
    const uint32_t *ids;       /* n*4: vertex ids of the primitive (2D: [0..1]), collider id in [3] */
    int32_t nv;                /* mesh vertices */
    const real *local_vtx;     /* nv*D */
    real *world_vtx;           /* nv*D */
    const uint32_t *vtx_collider; /* nv */
    uint32_t *needs_block;     /* n: sort.wgsl:56-86 flag (one word per sample here) */
    uint32_t *node_head;       /* cap*64: per-node rigid particle list (sort.wgsl:140-161) */
    uint32_t *node_len;        /* cap*64 */
    uint32_t *next;            /* n */
} orc_rigid;

#ifdef __cplusplus
extern "C" {
#endif

int orc_dim(void);
int orc_num_threads(void);   /* 1 unless built with -fopenmp (the _omp flavour used as bench.py's CPU baseline) */
int orc_real_size(void);

/* index math — grid.wgsl:82-105, 284-292; particle3d.wgsl:41-45 */
uint32_t orc_pack_key(const int32_t *block);
uint32_t orc_hash(uint32_t packed_key);
void orc_assoc_cell(const float *pt, float cell_width, int32_t *cell_out);
void orc_block_and_local(const float *pt, float cell_width, int32_t *block_out, uint32_t *local_out);

/* prefix_sum.rs:71-83 (eval_cpu) and prefix_sum.wgsl:11-93 (GPU algorithm restated) */
void orc_prefix_sum_eval_cpu(uint32_t *v, int32_t len);
void orc_prefix_sum_gpu_algorithm(uint32_t *v, int32_t len);

/* kernel.wgsl */
void orc_eval_all(real x, real *w3);
int orc_nbh_shift(int i, int axis);
int orc_nbh_shift_shared(int i);

/* linear algebra helpers (own implementation; the reference's come from wgebra) */
void orc_svd(const real *m, real *u, real *s, real *vt);

/* models */
void orc_kirchoff_stress(int model, real lambda, real mu, const real *F, real *tau);
int orc_drucker_prager_project(const real *dp6, real *state3, real *F);

/* passes, in pipeline.rs:201-280 order */
void orc_sort(const orc_particles *p, const orc_params *prm, orc_grid *g);
void orc_grid_update_cdf(const orc_params *prm, orc_grid *g);
void orc_g2p_cdf(orc_particles *p, const orc_params *prm, const orc_grid *g);
void orc_p2g(const orc_particles *p, const orc_params *prm, orc_grid *g);
void orc_grid_update(const orc_params *prm, orc_grid *g);
void orc_g2p(orc_particles *p, const orc_params *prm, const orc_grid *g);
void orc_particle_update(orc_particles *p, const orc_params *prm);
void orc_step(orc_particles *p, const orc_params *prm, orc_grid *g, int n_substeps);

/* rigid bodies (two-way coupling) — rigid_impulses.wgsl:95-149, pipeline.rs:204-205,268-280.
 * orc_step_bodies = the full pipeline.rs:201-280 order for analytic colliders: the colliders of
 * prm (which must point at `cols`) are moved by the impulses p2g accumulated. */
void orc_update_world_mass_properties(orc_collider *cols, orc_body *bodies, int n);
void orc_integrate_bodies(const orc_params *prm, orc_collider *cols, const orc_body *bodies, int n, int32_t *impulses);
void orc_step_bodies(orc_particles *p, const orc_params *prm, orc_grid *g, orc_collider *cols, orc_body *bodies,
                     int n_substeps);

/* rigid particles of mesh colliders — solver/rigid_particle_update.wgsl:26-49, grid/sort.wgsl:38-86,139-161,
 * solver/p2g_cdf.wgsl:52-190. orc_step_full = the complete pipeline.rs:201-280 order. */
void orc_update_rigid_particles(const orc_params *prm, orc_rigid *rig);
void orc_sort_rigid(const orc_particles *p, const orc_params *prm, orc_grid *g, orc_rigid *rig);
void orc_p2g_cdf(const orc_params *prm, orc_grid *g, const orc_rigid *rig);
void orc_step_full(orc_particles *p, const orc_params *prm, orc_grid *g, orc_collider *cols, orc_body *bodies,
                   orc_rigid *rig, int move_bodies, int n_substeps);

#ifdef __cplusplus
}
#endif
#endif
