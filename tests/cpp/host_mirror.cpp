// Test program for include/wgsparkl_hip.hpp (the C++ host mirror of the reference's Rust API), built by
// tests/test_cpp_host_mirror.py with g++ against libwgsparkl3d_hip.so.
//   host_mirror nodevice        -> MpmPipeline::create must throw WGS_ERR_NO_DEVICE (a box without a GPU): exit 0 if it does
//   host_mirror smoke OUT.bin   -> the scene of the reference's own smoke test (src/pipeline.rs:302-331: 10^3 particles at
//                                  i / 2, r = h / 4, rho = 1, E = 1e5, nu = 0.33, g = (0, -9.81, 0), dt = (1/60)/10, h = 1,
//                                  capacity 100000), 10 substeps through MpmPipeline::queue_step; positions -> OUT.bin
#define WGS_DIM 3
#include "wgsparkl_hip.hpp"

#include <cstdio>
#include <cstring>

int main(int argc, char **argv) {
    using namespace wgsparkl;
    if (argc >= 2 && !std::strcmp(argv[1], "nodevice")) {
        try {
            MpmPipeline p = MpmPipeline::create(0);
        } catch (const Error &e) {
            std::printf("refused: status %d (%s)\n", (int)e.status, e.what());
            return e.status == WGS_ERR_NO_DEVICE ? 0 : 2;
        }
        std::printf("a pipeline was created: this box has a GPU\n");
        return 3;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "smoke")) {
        try {
            const float h = 1.0f;
            std::vector<wgs_particle> particles;
            for (int i = 0; i < 10; i++)
                for (int j = 0; j < 10; j++)
                    for (int k = 0; k < 10; k++) {
                        wgs_particle p{};
                        p.position[0] = (float)i / h / 2.0f;
                        p.position[1] = (float)j / h / 2.0f;
                        p.position[2] = (float)k / h / 2.0f;
                        p.dynamics = with_density(h / 4.0f, 1.0f);
                        p.model = from_young_modulus(100000.0f, 0.33f);
                        p.has_plasticity = 0;  // plasticity: None, phase: None (src/pipeline.rs:316-318)
                        p.has_phase = 0;
                        particles.push_back(p);
                    }
            wgs_sim_params params{};
            params.gravity[1] = -9.81f;
            params.dt = (1.0f / 60.0f) / 10.0f;
            MpmPipeline pipeline = MpmPipeline::create(0);
            MpmData data = MpmData::create(pipeline, params, particles, {}, h, 100000);
            pipeline.queue_step(data, 10, false);
            data.sync();
            const std::vector<float> pos = data.read_positions();
            const wgs_stats st = data.stats();
            std::FILE *f = std::fopen(argv[2], "wb");
            if (!f) return 4;
            std::fwrite(pos.data(), sizeof(float), pos.size(), f);
            std::fclose(f);
            std::printf("ok: %zu particles, %u active blocks\n", data.num_particles(), (unsigned)st.num_active_blocks);
            return 0;
        } catch (const Error &e) {
            std::printf("error: status %d (%s)\n", (int)e.status, e.what());
            return 1;
        }
    }
    std::printf("usage: host_mirror nodevice | smoke OUT.bin\n");
    return 64;
}
