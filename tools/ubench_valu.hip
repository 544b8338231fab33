// developer utility: VALU issue rate on gfx950 by waves per SIMD — v_fma_f32 against v_pk_fma_f32, independent accumulators.
// build + run on a GPU box: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int PK> __global__ void k(float *out, int iters, float a, float b) {
    float2v acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = float2v{(float)threadIdx.x + i, (float)i};
    float2v A = {a, a * 1.5f}, B = {b, b * 0.5f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(A), "v"(B));
                else asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i].x) : "v"(A.x), "v"(B.x));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int pk = 0; pk < 2; pk++)
        for (int wps = 1; wps <= 4; wps++) {   // waves per SIMD: block = 256 * wps threads, one block per CU
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (pk) k<1><<<256, 256 * wps>>>(out, iters, 1.0001f, 0.5f);
                else k<0><<<256, 256 * wps>>>(out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) {
                    const double inst_per_simd = (double)iters * 32 * wps;   // wave-instructions issued on one SIMD
                    printf("%s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms,
                           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
                }
            }
        }
    return 0;
}
