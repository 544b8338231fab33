// wgsparkl_hip.hpp — C++17 host-side mirror of the reference's Rust API for the substep path, header-only, over the C ABI
// of wgsparkl_hip.h (define WGS_DIM = 2 or 3 before including; link libwgsparkl{2,3}d_hip.so).
//
// The reference's host is Rust (src/pipeline.rs, src/solver/particle3d.rs, src/models/mod.rs); this image has no Rust
// toolchain, so the compiled-language host side is C++. Names, argument meaning and error behaviour follow the
// reference's call sites (src/pipeline.rs:98-128,176-200; src_testbed/step.rs:79-132; examples/sand3.rs:28-113):
//
//   reference (Rust)                                           this header
//   ---------------------------------------------------------  -------------------------------------------------------
//   MpmPipeline::new(&device) -> Result<Self, ComposerError>    wgsparkl::MpmPipeline::create(hip_device)   (throws Error)
//   MpmData::new(device, params, &particles, &bodies,           wgsparkl::MpmData::create(pipeline, params, particles,
//                &colliders, cell_width, grid_capacity)                                   colliders, cell_width, grid_capacity)
//   pipeline.queue_step(&mut data, &mut queue, timestamps);     pipeline.queue_step(data, num_substeps, timestamps)
//   for _ in 0..n { queue.encode(..) }; submit                    (records AND submits n substeps; asynchronous)
//   device.poll(Maintain::Wait)                                 data.sync()
//   ParticleDynamics::with_density(radius, density)             wgsparkl::with_density(radius, density)
//   ElasticCoefficients::from_young_modulus(E, nu)              wgsparkl::from_young_modulus(E, nu)
//   DruckerPrager::new(E, nu)                                   wgsparkl::drucker_prager(E, nu)
//   Result / panics                                             wgsparkl::Error (status + wgs_last_error()); no CPU fallback:
//                                                               create() throws WGS_ERR_NO_DEVICE without a HIP device
#ifndef WGSPARKL_HIP_HPP
#define WGSPARKL_HIP_HPP

#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "wgsparkl_hip.h"

namespace wgsparkl {

struct Error : std::runtime_error {
    wgs_status status;
    Error(wgs_status st, const char *what) : std::runtime_error(what ? what : "wgsparkl error"), status(st) {}
};

inline void check(wgs_status st) {
    if (st != WGS_OK) throw Error(st, wgs_last_error());
}

// src/models/mod.rs:52-75 ElasticCoefficients::from_young_modulus / lame_lambda_mu
inline wgs_elastic_coefficients from_young_modulus(float young_modulus, float poisson_ratio) {
    const float d = (1.0f + poisson_ratio) * (1.0f - 2.0f * poisson_ratio);
    return {young_modulus * poisson_ratio / d, young_modulus / (2.0f * (1.0f + poisson_ratio))};
}

// src/models/drucker_prager.rs:17-34 DruckerPrager::new (h0 = 35 deg, h1 = 9 deg, h2 = 0.2, h3 = 10 deg)
inline wgs_drucker_prager drucker_prager(float young_modulus, float poisson_ratio) {
    // (a non-positive modulus keeps the reference's placeholder lambda = mu = -1: "take them from the particle's model")
    const wgs_elastic_coefficients c = young_modulus > 0.0f ? from_young_modulus(young_modulus, poisson_ratio) : wgs_elastic_coefficients{-1.0f, -1.0f};
    const float deg = 3.14159265358979323846f / 180.0f;
    return {35.0f * deg, 9.0f * deg, 0.2f, 10.0f * deg, c.lambda, c.mu};
}

// src/solver/particle3d.rs:28-42 ParticleDynamics::with_density: V0 = (2 r)^dim, m = rho V0, F = I, everything else zero
inline wgs_particle_dynamics with_density(float radius, float density) {
    wgs_particle_dynamics d{};
    for (int k = 0; k < WGS_DIM; k++) d.def_grad[k * WGS_DIM + k] = 1.0f;
    d.init_volume = std::pow(2.0f * radius, (float)WGS_DIM);
    d.init_radius = radius;
    d.mass = density * d.init_volume;
    return d;
}

class MpmData;

// src/pipeline.rs:24-39,176-193 MpmPipeline. Immutable after creation: one pipeline may serve many MpmData.
class MpmPipeline {
  public:
    static MpmPipeline create(int hip_device) {
        // (the structs carry no size field: a library built from another header version is refused here, once — wgsparkl_hip.h)
        if (wgs_abi_version() != WGS_ABI_VERSION) throw Error(WGS_ERR_INVALID_ARGUMENT, "libwgsparkl_hip was built from another version of wgsparkl_hip.h");
        wgs_pipeline *p = nullptr;
        check(wgs_pipeline_create(hip_device, &p));
        return MpmPipeline(p);
    }
    MpmPipeline(MpmPipeline &&o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    MpmPipeline &operator=(MpmPipeline &&o) noexcept {
        if (this != &o) { reset(); h_ = std::exchange(o.h_, nullptr); }
        return *this;
    }
    MpmPipeline(const MpmPipeline &) = delete;
    MpmPipeline &operator=(const MpmPipeline &) = delete;
    ~MpmPipeline() { reset(); }

    // queue_step + num_substeps x queue.encode + submit (src/pipeline.rs:195-281, src_testbed/step.rs:122-128,169):
    // asynchronous, stream-ordered
    inline void queue_step(MpmData &data, uint32_t num_substeps, bool add_timestamps = false) const;

    wgs_pipeline *handle() const { return h_; }

  private:
    explicit MpmPipeline(wgs_pipeline *p) : h_(p) {}
    void reset() {
        if (h_) wgs_pipeline_destroy(h_);
        h_ = nullptr;
    }
    wgs_pipeline *h_ = nullptr;
};

// src/pipeline.rs:84-173 MpmData: owns every device buffer of one simulation (RAII like the reference's fields).
class MpmData {
  public:
    // MpmData::new (src/pipeline.rs:98-128): the particle and collider slices are borrowed for the call only
    static MpmData create(const MpmPipeline &pipeline, const wgs_sim_params &params, const std::vector<wgs_particle> &particles,
                          const std::vector<wgs_collider> &colliders, float cell_width, uint32_t grid_capacity) {
        wgs_data *d = nullptr;
        check(wgs_data_create(pipeline.handle(), &params, particles.data(), particles.size(), colliders.data(), colliders.size(),
                              cell_width, grid_capacity, &d));
        return MpmData(d, particles.size(), colliders.size());
    }
    MpmData(MpmData &&o) noexcept : h_(std::exchange(o.h_, nullptr)), n_(o.n_), nc_(o.nc_) {}
    MpmData &operator=(MpmData &&o) noexcept {
        if (this != &o) { reset(); h_ = std::exchange(o.h_, nullptr); n_ = o.n_; nc_ = o.nc_; }
        return *this;
    }
    MpmData(const MpmData &) = delete;
    MpmData &operator=(const MpmData &) = delete;
    ~MpmData() { reset(); }

    void set_constitutive_model(int32_t model) { check(wgs_set_constitutive_model(h_, model)); }
    // device.poll(Maintain::Wait) (src/pipeline.rs:339); also reports device-side sticky errors (grid overflow, key range)
    void sync() { check(wgs_sync(h_)); }
    // the per-frame host -> device writes of src_testbed/step.rs:79-119 and ui.rs:91-104
    void set_sim_params(const wgs_sim_params &p) { check(wgs_set_sim_params(h_, &p)); }
    void set_collider_poses(const std::vector<wgs_pose> &poses) { check(wgs_set_collider_poses(h_, poses.data(), nullptr, poses.size())); }
    void set_body_velocities(const std::vector<wgs_velocity> &vels) { check(wgs_set_body_velocities(h_, vels.data(), vels.size())); }
    void set_body_mass_properties(const std::vector<wgs_mass_properties> &m) { check(wgs_set_body_mass_properties(h_, m.data(), m.size())); }
    // the blocking reads of src_testbed/step.rs:129-132,175-198 (poses) and of the positions buffer (particle3d.rs:197-201)
    std::vector<float> read_positions() {
        std::vector<float> out(n_ * WGS_DIM);
        check(wgs_read_positions(h_, out.data()));
        return out;
    }
    // the optional interop view: device pointers to the position quads + ids of the current state (sorted order) and their stream
    wgs_device_ptrs device_ptrs() {
        wgs_device_ptrs v{};
        check(wgs_get_device_ptrs(h_, &v));
        return v;
    }
    std::vector<wgs_particle> read_particles() {
        std::vector<wgs_particle> out(n_);
        check(wgs_read_particles(h_, out.data(), nullptr));
        return out;
    }
    std::vector<wgs_pose> read_body_poses() {
        std::vector<wgs_pose> poses(nc_);
        check(wgs_read_body_poses(h_, poses.data(), nullptr, nullptr, nc_));
        return poses;
    }
    // GpuTimestamps (src/pipeline.rs:201-271): milliseconds per named pass of the last timestamped queue_step
    std::vector<float> read_timings() {
        std::vector<float> ms(WGS_NUM_PASSES);
        check(wgs_read_timings(h_, ms.data()));
        return ms;
    }
    wgs_stats stats() {
        wgs_stats s{};
        check(wgs_get_stats(h_, &s));
        return s;
    }
    size_t num_particles() const { return n_; }
    wgs_data *handle() const { return h_; }

  private:
    MpmData(wgs_data *d, size_t n, size_t nc) : h_(d), n_(n), nc_(nc) {}
    void reset() {
        if (h_) wgs_data_destroy(h_);
        h_ = nullptr;
    }
    wgs_data *h_ = nullptr;
    size_t n_ = 0, nc_ = 0;
};

inline void MpmPipeline::queue_step(MpmData &data, uint32_t num_substeps, bool add_timestamps) const {
    check(wgs_step(h_, data.handle(), num_substeps, add_timestamps ? 1 : 0));
}

}  // namespace wgsparkl

#endif  // WGSPARKL_HIP_HPP
