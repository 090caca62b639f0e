// Issue rate of v_mfma_f32_32x32x2_f32 (the exact-fp32 matrix instruction every product of this repo uses):
// waves per SIMD x independent accumulators.  Build: hipcc -O3 --offload-arch=gfx950 mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 123.456f) out[0] = s;
}
template <int NACC> void run(int threads, float* d) {
    const int iters = 4096 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, threads>>>(d, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NACC><<<256, threads>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 256.0 * (threads / 64) * 4096.0 * 5;        // MFMA instructions
    printf("waves/SIMD %d, %d accumulators: %7.1f TFLOP/s   (%.1f ns per MFMA per SIMD)\n", threads / 256, NACC,
           mf * 4096 / (ms * 1e-3) / 1e12, ms * 1e6 / (mf / 1024));
}
int main() {
    float* d; hipMalloc(&d, 4096);
    run<1>(256, d); run<2>(256, d); run<4>(256, d);
    run<1>(512, d); run<2>(512, d); run<1>(1024, d); run<2>(1024, d);
    return 0;
}
