/*
 * wgsparkl_hip.h — C ABI of the MI355X-native MLS-MPM step (drop-in for the
 * one hot path of dimforge/wgsparkl: `MpmPipeline::queue_step`).
 *
 * Build twice, like the reference builds its two crates from one source tree
 * (Cargo.toml:1-8): -DWGS_DIM=3 -> libwgsparkl3d_hip.so, -DWGS_DIM=2 ->
 * libwgsparkl2d_hip.so. Both export the same symbol names.
 *
 * Every entry point cites the reference interface it replaces (paths relative
 * to /root/reference). Plain pointers and sizes only; no exceptions cross the
 * boundary; every call returns a wgs_status (0 = ok) and leaves a message for
 * wgs_last_error(). Handles are not internally synchronised: one host thread
 * per wgs_data at a time (the reference calls from one Bevy system,
 * src_testbed/lib.rs:60-69).
 */
#ifndef WGSPARKL_HIP_H
#define WGSPARKL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifndef WGS_DIM
#define WGS_DIM 3
#endif

#if WGS_DIM == 2
#define WGS_ANG_DIM 1
#else
#define WGS_ANG_DIM 3
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t wgs_status;
enum {
    WGS_OK = 0,
    WGS_ERR_INVALID_ARGUMENT = 1,
    WGS_ERR_NO_DEVICE = 2,       /* no HIP device / HIP runtime failure at creation */
    WGS_ERR_HIP = 3,             /* a HIP call failed; see wgs_last_error() */
    WGS_ERR_GRID_OVERFLOW = 4,   /* more active blocks than grid_capacity (the reference drops them silently, src/grid/grid.wgsl:126-128,163) */
    WGS_ERR_KEY_RANGE = 5,       /* a block left the packed-key range (src/grid/grid.wgsl:88-95; quirk B5) */
    WGS_ERR_UNSUPPORTED = 6
};

/* src/solver/params.rs:6-16 SimulationParams (2D: gravity[2]; the f32 padding is not exposed). */
typedef struct {
    float gravity[WGS_DIM];
    float dt;
} wgs_sim_params;

/* src/models/mod.rs:65-70 ElasticCoefficients */
typedef struct { float lambda, mu; } wgs_elastic_coefficients;
/* src/models/drucker_prager.rs:6-15 DruckerPrager */
typedef struct { float h0, h1, h2, h3, lambda, mu; } wgs_drucker_prager;
/* src/models/drucker_prager.rs:36-42 DruckerPragerPlasticState */
typedef struct { float plastic_deformation_gradient_det, plastic_hardening, log_vol_gain; } wgs_plastic_state;
/* src/solver/particle_update.rs:35-40 ParticlePhase */
typedef struct { float phase, max_stretch; } wgs_particle_phase;

/* src/solver/particle3d.rs:44-51 / particle2d.rs:42-49 Cdf */
typedef struct {
    float normal[WGS_DIM];
    float rigid_vel[WGS_DIM];
    float signed_distance;
    uint32_t affinity;
} wgs_cdf;

/* src/solver/particle3d.rs:16-26 ParticleDynamics, Rust field order; matrices column-major (nalgebra). */
typedef struct {
    float velocity[WGS_DIM];
    float def_grad[WGS_DIM * WGS_DIM];
    float affine[WGS_DIM * WGS_DIM];
    wgs_cdf cdf;
    float init_volume;
    float init_radius;
    float mass;
} wgs_particle_dynamics;

/* src/solver/particle3d.rs:53-60 Particle. Option<T> is flattened to has_* + value;
 * has_* == 0 applies the reference defaults of src/models/mod.rs:24,33-36. */
typedef struct {
    float position[WGS_DIM];
    wgs_particle_dynamics dynamics;
    wgs_elastic_coefficients model;
    uint32_t has_plasticity;
    wgs_drucker_prager plasticity;
    uint32_t has_phase;
    wgs_particle_phase phase;
} wgs_particle;

/* One coupled collider: what MpmData::new keeps of rapier's (RigidBody, Collider)
 * pair (src/pipeline.rs:107-117 -> wgrapier GpuBodySet: shape, pose, velocity,
 * world mass properties). Only analytic shapes are handled, like collide()
 * (src/collision/collide.wgsl:36-38 skips polylines and trimeshes). */
enum { WGS_SHAPE_BALL = 0, WGS_SHAPE_CUBOID = 1, WGS_SHAPE_CAPSULE = 2,
       WGS_SHAPE_MESH = 3 /* trimesh / heightfield / polyline: no analytic projection, coupled through the rigid
                             particles of wgs_set_rigid_particles */ };
typedef struct {
    float rotation[4];     /* 3D: unit quaternion (i, j, k, w); 2D: (cos, sin, 0, 0) */
    float translation[3];
    float scale;
} wgs_pose;                /* wgebra Sim3 / Sim2 */
typedef struct {
    float linear[3];
    float angular[3];      /* 2D: angular[0] */
} wgs_velocity;            /* wgrapier Body::Velocity */
typedef struct {
    uint32_t shape_type;   /* WGS_SHAPE_* */
    float shape[4];        /* ball: r | cuboid: half extents | capsule: half height (local y), r */
    wgs_pose pose;
    wgs_velocity velocity;
    float com[3];          /* world-space centre of mass (wgrapier MassProperties.com) */
} wgs_collider;
#define WGS_MAX_COLLIDERS 16 /* src/grid/grid.wgsl:230-240: 16 affinity + 16 sign bits */
/* Mass properties of the rigid body behind a collider (wgrapier GpuBodySet local_mprops, bound at
 * src/solver/rigid_impulses.wgsl:81-84). All zero (the default) = kinematic or fixed body: it follows its
 * velocity. Non-zero = dynamic: the particles push it back (two-way coupling, src/solver/p2g.wgsl:200-228,
 * src/solver/rigid_impulses.wgsl:95-136). The centre of mass is wgs_collider.com. */
typedef struct {
    float inv_mass[3];           /* per-axis inverse mass */
    float inv_inertia_local[9];  /* 3D: column-major inverse inertia tensor in the body frame; 2D: [0] = 1/I */
} wgs_mass_properties;

/* Constitutive model (the reference picks at shader-compile time,
 * src/solver/particle_update.wgsl:7-8; default = corotated like the reference). */
enum { WGS_MODEL_COROTATED = 0, WGS_MODEL_NEO_HOOKEAN = 1 };

/* The reference's 10 timestamped passes (src/pipeline.rs:201-271). The fused
 * G2P + particle update reports its time under WGS_PASS_G2P; in collider simulations
 * the same kernel's second launch, over the blocks near a collider, is reported under
 * WGS_PASS_PARTICLES_UPDATE (0 without colliders). */
enum {
    WGS_PASS_UPDATE_RIGID_PARTICLES = 0, WGS_PASS_GRID_SORT = 1, WGS_PASS_GRID_UPDATE_CDF = 2,
    WGS_PASS_P2G_CDF = 3, WGS_PASS_G2P_CDF = 4, WGS_PASS_P2G = 5, WGS_PASS_GRID_UPDATE = 6,
    WGS_PASS_G2P = 7, WGS_PASS_PARTICLES_UPDATE = 8, WGS_PASS_INTEGRATE_BODIES = 9,
    WGS_NUM_PASSES = 10
};

/* Grid node keyed by world cell coordinate (active blocks are numbered by an
 * atomic counter in the reference, src/grid/grid.wgsl:327, so physical ids are
 * not comparable between runs; virtual ids are). */
typedef struct {
    int32_t cell[WGS_DIM];
    float velocity[WGS_DIM];   /* momentum_velocity_mass.xyz after the grid update */
    float mass;
    float cdf_distance;
    uint32_t cdf_affinities;
    uint32_t cdf_closest_id;
} wgs_node_record;

/* src/grid/grid.rs:250-256 GpuActiveBlockHeader */
typedef struct {
    int32_t virtual_id[WGS_DIM];
    uint32_t first_particle;
    uint32_t num_particles;
} wgs_block_record;

typedef struct {
    uint32_t num_particles;
    uint32_t num_active_blocks;   /* of the last executed substep */
    uint32_t grid_capacity;       /* rounded up to a power of two like src/grid/grid.rs:283; grows, see wgs_set_grid_growth */
    uint32_t overflow;            /* sticky: WGS_ERR_GRID_OVERFLOW / WGS_ERR_KEY_RANGE seen on device */
    uint64_t substeps_done;
    uint64_t device_bytes;        /* HBM held by this wgs_data */
    uint32_t num_near_collider_blocks; /* particle-bearing blocks whose tile sees a collider (the CPIC passes' list), last substep */
    uint32_t grid_growths;        /* times the block capacity was doubled (wgs_set_grid_growth) */
    uint64_t cell_changers;       /* particles that changed their associated cell since creation (they are what the sort has to move:
                                     the difference of two reads / (particles x substeps) is the mover fraction per substep) */
    uint64_t table_rebuilds;      /* substeps that rebuilt the table of block ids (a full binning pass in front of the sort: the first substep,
                                     every 1024th, after a growth of the grid, when the ids ran out) — with grid_growths, the fixed-cost events
                                     inside a timed region */
    uint32_t block_ids;           /* physical block ids handed out since the last table rebuild (high-water mark; the capacity bounds it) */
    uint32_t block_ids_free;      /* ... of them on the free list: blocks evicted from the table after 8 substeps without activity, their ids reused */
    uint32_t table_marks;         /* table slots marked "evicted" and not yet reused by an insertion */
    uint32_t table_refreshes;     /* times the table was cleared of those marks: the live blocks re-inserted under their own ids, no particle touched */
} wgs_stats;

typedef struct wgs_pipeline wgs_pipeline;
typedef struct wgs_data wgs_data;

/* Thread-local message of the last failing call on this thread. */
const char *wgs_last_error(void);
/* 2 or 3: the dimension this library was built for (cargo features dim2/dim3, src/lib.rs:4-15). */
int32_t wgs_dim(void);
/* Build identification: dimension, target arch and — never in a shipped library — "WGS_ABLATE" when the kernels were
 * compiled with their ablation switches (bench.py refuses such a build). */
const char *wgs_build_info(void);
/* Version of this header's structs and entry points as the LIBRARY was built (WGS_ABI_VERSION as the caller was): the structs carry no
 * size field — wgs_get_stats writes sizeof(wgs_stats) of ITS header —, so a binding checks wgs_abi_version() == WGS_ABI_VERSION once,
 * after loading the library, and refuses to go on otherwise (include/wgsparkl_hip.hpp and wgsparkl_amd/_ffi.py do). History: 5 =
 * wgs_stats grew by block_ids .. table_refreshes (24 bytes); 6 = this function. New counters will come behind a call of their own. */
#define WGS_ABI_VERSION 6
uint32_t wgs_abi_version(void);

/* MpmPipeline::new(&Device) -> Result<Self, ComposerError>  (src/pipeline.rs:176-193).
 * Binds to HIP device `hip_device`; fails with WGS_ERR_NO_DEVICE when there is none
 * (there is no CPU fallback). */
wgs_status wgs_pipeline_create(int32_t hip_device, wgs_pipeline **out);
void wgs_pipeline_destroy(wgs_pipeline *pipeline);

/* MpmData::new(device, params, &[Particle], &RigidBodySet, &ColliderSet, cell_width, grid_capacity)
 * (src/pipeline.rs:98-128). `particles`/`colliders` are borrowed for the call only. */
wgs_status wgs_data_create(wgs_pipeline *pipeline, const wgs_sim_params *params,
                           const wgs_particle *particles, size_t num_particles,
                           const wgs_collider *colliders, size_t num_colliders,
                           float cell_width, uint32_t grid_capacity, wgs_data **out);
void wgs_data_destroy(wgs_data *data);

/* Runtime replacement for the compile-time import at src/solver/particle_update.wgsl:7-8. */
wgs_status wgs_set_constitutive_model(wgs_data *data, int32_t model);

/* MpmPipeline::queue_step + `for _ in 0..num_substeps { queue.encode(..) }` + submit
 * (src/pipeline.rs:195-281, src_testbed/step.rs:122-128,169): enqueues `num_substeps`
 * substeps on the data's HIP stream and returns without waiting. `timestamps` != 0
 * brackets each pass with HIP events (src/pipeline.rs:201-271 compute_pass(name, add_timestamps)). */
wgs_status wgs_step(wgs_pipeline *pipeline, wgs_data *data, uint32_t num_substeps, int32_t timestamps);
/* device.poll(Maintain::Wait) (src/pipeline.rs:339). Also reports device-side sticky errors. */
wgs_status wgs_sync(wgs_data *data);

/* queue.write_buffer(sim_params) from the UI (src_testbed/ui.rs:91-104) */
wgs_status wgs_set_sim_params(wgs_data *data, const wgs_sim_params *params);
/* queue.write_buffer(bodies.poses()) (src_testbed/step.rs:79-96); com is refreshed by the caller too */
wgs_status wgs_set_collider_poses(wgs_data *data, const wgs_pose *poses, const float *coms /* n*3 or NULL */, size_t n);
/* queue.write_buffer(bodies.vels()) (src_testbed/step.rs:98-119) */
wgs_status wgs_set_body_velocities(wgs_data *data, const wgs_velocity *vels, size_t n);
/* GpuBodySet::from_rapier's local mass properties (src/pipeline.rs:145, wgrapier): which bodies are dynamic.
 * Stream-ordered like the other setters. On sharded data the impulses are reduced over the ranks (wgs_sharded_step). */
wgs_status wgs_set_body_mass_properties(wgs_data *data, const wgs_mass_properties *mprops, size_t n);
/* Rigid particles of the mesh colliders = the buffers GpuRigidParticles::from_rapier builds on the host
 * (src/solver/particle3d.rs:100-150, 2D src/solver/particle2d.rs:75-125; sampling step = cell width,
 * src/pipeline.rs:144) plus the mesh vertex buffers of wgrapier's GpuBodySet: sample points and mesh vertices in
 * the collider's LOCAL frame, per sample the vertex ids of the triangle (2D: segment, vertex[2] unused) it was
 * taken from and its collider. Every substep then runs `update rigid particles`, the rigid-particle block marks
 * and `p2g_cdf` (src/solver/rigid_particle_update.wgsl, src/grid/sort.wgsl:38-86, src/solver/p2g_cdf.wgsl).
 * Copies its inputs; n == 0 removes them. Blocking. On sharded data every rank passes ALL samples (the node cdfs are a
 * function of position and colliders: the two ranks of a face compute the same values for the nodes they share). */
typedef struct { uint32_t vertex[3]; uint32_t collider; } wgs_sample_ids;   /* GpuSampleIds */
wgs_status wgs_set_rigid_particles(wgs_data *data, const float *local_points /* n*DIM */, const wgs_sample_ids *ids, size_t n,
                                   const float *local_vertices /* nv*DIM */, const uint32_t *vertex_collider_ids, size_t nv);
/* poses_staging read-back after the step (src_testbed/step.rs:129-132,175-198): the poses the device
 * integrated (every substep ends with src/solver/rigid_impulses.wgsl:95-136). Blocking. `vels` and `coms`
 * (n*3 floats, world space) may be NULL. */
wgs_status wgs_read_body_poses(wgs_data *data, wgs_pose *poses, wgs_velocity *vels, float *coms, size_t n);

/* positions is the only particle buffer the reference creates COPY_SRC (src/solver/particle3d.rs:197-201).
 * out: num_particles * WGS_DIM floats, in the caller's original particle order. Blocking. */
wgs_status wgs_read_positions(wgs_data *data, float *out);
/* Optional interop view (SURVEY §8b "Ownership": no borrowed device pointers escape except through this call): where the
 * current particle state lives on the device, for a renderer or a coupling code that reads positions without a host round
 * trip (the reference hands its `positions` buffer to the testbed's render pass the same way, src/solver/particle3d.rs:197-201,
 * src_testbed/step.rs:144-163). One float4 per slot — the first WGS_DIM floats are the position — in the SORTED order of the last
 * substep; particle_ids[slot] is the caller's index of the particle in that slot. Read-only; valid until the next call that
 * steps, sets, restores or destroys `data`. Work enqueued on `hip_stream` before this call is what produced the contents:
 * synchronise, or enqueue the reader behind it. Not blocking. Single-domain data. */
typedef struct {
    const float *position_quads;   /* capacity * 4 floats */
    const uint32_t *particle_ids;  /* capacity */
    uint32_t count;                /* slots in use = number of particles */
    uint32_t capacity;
    uint32_t dim;                  /* WGS_DIM of the library */
    uint32_t reserved;
    void *hip_stream;              /* hipStream_t of `data` */
} wgs_device_ptrs;
wgs_status wgs_get_device_ptrs(wgs_data *data, wgs_device_ptrs *out);
/* Full particle state in the caller's original order (tests / checkpoint). Blocking. */
wgs_status wgs_read_particles(wgs_data *data, wgs_particle *out, wgs_plastic_state *plastic_out /* may be NULL */);
/* Render hand-off: src_testbed/prep_vertex_buffer{2,3}d.wgsl `main`, the compute pass the testbed
 * queues after the substeps (src_testbed/step.rs:144-163). One InstanceData per particle in the caller's order
 * (src_testbed/instancing3d.rs:66-74): base_color is an INPUT (set once at startup by the testbed), the rest is
 * written. Modes = RenderMode (src_testbed/prep_vertex_buffer.rs:11-18). */
enum {
    WGS_RENDER_DEFAULT = 0, WGS_RENDER_VOLUME = 1, WGS_RENDER_VELOCITY = 2, WGS_RENDER_CDF_NORMALS = 3,
    WGS_RENDER_CDF_DISTANCES = 4, WGS_RENDER_CDF_SIGNS = 5
};
typedef struct {
    float deformation[3][4]; /* mat3x3 columns, padded to vec4 (2D: F in the xy block, identity elsewhere) */
    float position[4];       /* xyz, 0 (2D: z = 0) */
    float base_color[4];
    float color[4];
} wgs_instance;
/* host buffer of n records: staged to the device, filled, copied back. Blocking. */
wgs_status wgs_prep_vertex_buffer(wgs_data *data, uint32_t mode, wgs_instance *instances);
/* interop form: `device_instances` is a DEVICE pointer to n records that stay resident (the testbed's vertex
 * buffer); stream-ordered, returns immediately. */
wgs_status wgs_prep_vertex_buffer_device(wgs_data *data, uint32_t mode, wgs_instance *device_instances);

/* Checkpoint restore (the reference has no checkpointing, MpmData::new always starts from
 * DruckerPragerPlasticState::default, src/models/drucker_prager.rs:36-53): wgs_read_particles +
 * wgs_read_body_poses -> wgs_data_create (+ wgs_set_body_mass_properties) -> wgs_set_plastic_state continues a
 * run bit-exactly. `states`: one record per particle in the caller's order. Blocking; single-domain data. */
wgs_status wgs_set_plastic_state(wgs_data *data, const wgs_plastic_state *states);
/* Sparse grid of the last substep: nodes of every active block. *count receives the number written. */
wgs_status wgs_read_grid(wgs_data *data, wgs_node_record *out, size_t capacity, size_t *count);
/* Active block headers + the sorted particle ids (GpuParticles.sorted_ids, src/solver/particle3d.rs:178-180)
 * of the last substep. sorted_ids may be NULL. */
wgs_status wgs_read_blocks(wgs_data *data, wgs_block_record *out, size_t capacity, size_t *count, uint32_t *sorted_ids);
/* GpuTimestamps read-back (src_testbed/step.rs:219-251): ms per pass summed over the substeps of the
 * last wgs_step(.., timestamps=1) call. */
wgs_status wgs_read_timings(wgs_data *data, float ms[WGS_NUM_PASSES]);
/* The cost of one timing mark (a HIP event is a barrier packet of its own): the average distance of two marks
 * recorded back to back in the same timestamped substeps. Every pass time of wgs_read_timings includes one; a pass
 * with K launches between its marks took (time - overhead) of kernel time. */
wgs_status wgs_read_timing_overhead(wgs_data *data, float *ms_per_mark);
wgs_status wgs_get_stats(wgs_data *data, wgs_stats *out);
/* Per-material constants are per-PARTICLE data in the reference (GpuModels, src/models/mod.rs:12-50; init_volume and
 * mass in ParticleDynamics), re-read by every pass. When every particle of a simulation has the same (mass,
 * init_volume, lambda, mu) the step keeps them as kernel arguments instead (32 bytes per particle and substep less
 * through HBM; results are bit-identical). wgs_data_create detects this by itself; on SHARDED data a rank only sees
 * its own particles, so the caller asserts it — on every rank, before the first step — with this call. 3D only
 * (a no-op in the 2D library). */
wgs_status wgs_set_uniform_material(wgs_data *data, float mass, float init_volume, float lambda, float mu);
/* The reference sizes the sparse grid once and has a stub where it should grow it ("TODO: resize the hashmap and
 * retry", src/grid/grid.rs:43-45,116-117); blocks beyond the capacity are dropped silently
 * (src/grid/grid.wgsl:126-128). Here, by default, the block capacity DOUBLES whenever a wgs_step / wgs_sharded_step
 * call finds that the previous call ended with more than half of it active (the check reads counters left in pinned
 * host memory: no synchronisation; the growth itself drains the stream once). Particle state is untouched: the next
 * substep rebuilds the table. enabled = 0 restores the fixed capacity; WGS_ERR_GRID_OVERFLOW remains for growth that
 * outruns the check (more than half a capacity of new blocks inside one call). */
wgs_status wgs_set_grid_growth(wgs_data *data, int32_t enabled);
/* Environment (read once per wgs_data_create; developer switches, none of them changes a result): WGS_DEBUG = bit mask of
 * launch-SHAPE choices (two launches instead of one paired launch, the general binning pass on every substep, the grid
 * update and the message packing as launches of their own instead of workgroups of the P2G launch, ...) and
 * WGS_REHASH_PERIOD = substeps between unconditional rebuilds of the block table — used by the A/B tools and by the tests
 * that assert those shapes are bit-identical; WGS_TRACE drains the stream at every pass boundary and says so on stderr.
 * bench.py refuses to run with WGS_DEBUG / WGS_REHASH_PERIOD set. Switches that DO change results (ablations for timing
 * experiments) exist only in builds with -DWGS_ABLATE, which wgs_build_info() names and bench.py refuses. */
/* TEST HOOK (not part of the drop-in surface): runs the device-side exclusive scan that replaces WgPrefixSum
 * (src/grid/prefix_sum.rs:17-152, prefix_sum.wgsl:11-93) on caller data, so that the reference's own scan test
 * vectors (src/grid/prefix_sum.rs:183-229) can be put through the HIP code: out[i] = sum of values[0..i), *total
 * (may be NULL) = sum of all. Blocking. */
wgs_status wgs_debug_scan(wgs_pipeline *pipeline, const uint32_t *values, uint32_t n, uint32_t *out, uint32_t *total);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU (x-slab domain decomposition). NEW DESIGN: the reference is single-GPU (one wgpu::Device,
 * src/pipeline.rs:176-193; SURVEY.md sections 5 and 8e). One process per GPU owns the particles whose
 * associated block has bx in [block_lo, block_hi) — its core range — and ONE neighbour exchange per substep
 * (wgsparkl_amd/csrc/kernels_shard.h): after P2G both neighbours swap the partial (momentum, mass) sums of the node
 * layers they share, and in the SAME message the records of the particles that left the sender's core range in the
 * previous substep; the old owner still transfers such a particle to the grid in this substep, the new owner runs its
 * G2P + particle update from the message and keeps it. No host synchronisation anywhere inside a substep.
 * `global_ids` are the particles' ids in the global scene (the canonical summation order is by id, so a
 * sharded run reproduces the single-GPU sums). All ranks must pass the same `force_plastic`.
 * The caller replays the reference contract — one call per frame, src_testbed/step.rs:122-128:
 *   rank 0: wgs_comm_get_unique_id(id) -> the host broadcasts the 128 bytes (MPI, torch.distributed, a file ...)
 *   every rank: wgs_comm_create(pipeline, id, rank, world, 0, &comm)      (ncclCommInitRank; rank r talks to r-1, r+1)
 *               wgs_data_create_sharded(...); wgs_shard_attach(data, comm, 0, 0, halo_cap, mig_cap)
 *   per frame:  wgs_sharded_step(pipeline, data, num_substeps); ... wgs_sync(data)
 * RCCL point-to-point is the transport, bound at run time with dlopen("librccl.so.1"): the library has no link-time
 * dependency on RCCL and single-GPU users never load it. */
wgs_status wgs_data_create_sharded(wgs_pipeline *pipeline, const wgs_sim_params *params,
                                   const wgs_particle *particles, size_t num_particles, const uint32_t *global_ids,
                                   const wgs_collider *colliders, size_t num_colliders, float cell_width,
                                   uint32_t grid_capacity, uint32_t particle_capacity, int32_t block_lo,
                                   int32_t block_hi, int32_t force_plastic, wgs_data **out);
/* Sizes of the records a message holds (for choosing the capacities of wgs_shard_attach and reading wgs_shard_export):
 * a halo record = key + the partial sums of ONE x-layer pair of a block (2 * BW^(D-1) nodes); a particle record = the
 * quads of wgsparkl_amd/csrc/layout.h + persistent id + cdf epoch — in uniform-material mode (wgs_set_uniform_material;
 * 3D) the fourth word of its first quad holds F[8] instead of the mass and its F2 quad is not maintained. */
uint32_t wgs_shard_halo_record_bytes(void);
uint32_t wgs_shard_particle_record_bytes(void);
uint32_t wgs_shard_buffer_header_bytes(void);    /* every message / export buffer = header + records */
/* Run this wgs_data on the caller's HIP stream, so that its kernels are ordered with the caller's own work without host
 * synchronisation. The handle does not take ownership; the stream must outlive it. */
wgs_status wgs_set_stream(wgs_data *data, void *hip_stream);
/* full records of every particle this rank holds (read-back of a sharded run; BLOCKING). Between two calls of
 * wgs_sharded_step a rank also holds the particles that left its core range in the last substep: they are handed over
 * with the next substep's message, every particle is held by exactly one rank at any time. */
wgs_status wgs_shard_export(wgs_data *data, void *device_buf, uint32_t capacity_records, uint32_t *count);

#define WGS_COMM_ID_BYTES 128          /* ncclUniqueId */
#define WGS_COMM_SELF_NEIGHBOURS 1     /* flag: the rank is its own lower and upper neighbour (one-GPU timing proxy of an interior rank) */
typedef struct wgs_comm wgs_comm;
wgs_status wgs_comm_get_unique_id(uint8_t id[WGS_COMM_ID_BYTES]);
wgs_status wgs_comm_create(wgs_pipeline *pipeline, const uint8_t id[WGS_COMM_ID_BYTES], int32_t rank, int32_t world,
                           int32_t flags, wgs_comm **out);
/* Every wgs_data attached to the communicator must have been synchronised (wgs_sync) or destroyed before. */
void wgs_comm_destroy(wgs_comm *comm);
/* Allocates the message buffers of this slab (one outgoing, one incoming per neighbour; owned by the wgs_data, fixed
 * capacity, sent whole: no size handshake; an overflow is reported by the next wgs_sync). Call before the first step.
 * With a communicator the neighbours are the ranks next to comm's (has_lower / has_upper are ignored); comm == NULL = a
 * slab of a lockstep group inside one process (wgs_sharded_step_lockstep), or a slab without neighbours.
 * `halo_capacity_records`: halo records per message — about twice the active blocks of a face of the slab (every block
 * of the interface layer sends one x-layer pair, the blocks migrating particles touch a few more);
 * `migrant_capacity`: particles that can cross one face in one substep. A slab with TWO neighbours must be at least
 * 3 blocks wide. Two-way coupled (dynamic) bodies: every rank accumulates the fixed-point impulses of its own
 * particles (src/solver/p2g.wgsl:142-155) and the 16 x 8 int32 sums are all-reduced before integrate_bodies
 * (src/solver/rigid_impulses.wgsl:94-137) — integers, so the result does not depend on the order. */
wgs_status wgs_shard_attach(wgs_data *data, wgs_comm *comm, int32_t has_lower, int32_t has_upper,
                            uint32_t halo_capacity_records, uint32_t migrant_capacity);
/* `num_substeps` whole substeps of this rank's slab, asynchronous (MpmPipeline::queue_step + encode x N + submit,
 * src/pipeline.rs:195-281, for one slab of the decomposition). Every rank must call it with the same count. */
wgs_status wgs_sharded_step(wgs_pipeline *pipeline, wgs_data *data, uint32_t num_substeps);
/* The same phases for `num_slabs` slabs that live in ONE process on ONE device (passed in x order, attached with
 * comm == NULL), device-to-device copies as the transport: decomposition tests on a single GPU. For the duration of
 * the call the group's work is ordered on slab 0's stream; afterwards every slab's own stream waits for it. */
wgs_status wgs_sharded_step_lockstep(wgs_pipeline *pipeline, wgs_data **slabs, uint32_t num_slabs, uint32_t num_substeps);

#ifdef __cplusplus
}
#endif
#endif /* WGSPARKL_HIP_H */
