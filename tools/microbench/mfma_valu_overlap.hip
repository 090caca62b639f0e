// Do VALU instructions of one wave issue while another wave of the SAME SIMD runs a chain of v_mfma_f32_32x32x2_f32?
// 1024 threads = 4 waves per SIMD; role of a wave = its number >> 2 (waves w, w + 4, w + 8, w + 12 share SIMD w & 3).
//   mode 0: every wave MFMA;  1: every wave VALU;  2: waves 0-7 MFMA, waves 8-15 VALU (two of each per SIMD);
//   3: waves 0-7 MFMA, 8-15 idle;  4: waves 0-7 idle, 8-15 VALU.
// If (2) ~ max(3, 4) the units overlap; if (2) ~ (3) + (4) they share the issue port for the whole MFMA.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void k(float* out, int iters, int mode, float a0) {
    const int w = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3) && w < 8);
    const bool do_valu = mode == 1 || ((mode == 2 || mode == 4) && w >= 8);
    float s = 0.f;
    if (do_mfma) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float a = a0 + threadIdx.x, b = 2.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i];
    }
    if (do_valu) {
        float x0 = a0, x1 = a0 + 1, x2 = a0 + 2, x3 = a0 + 3;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {              // 256 independent-ish fmas per iteration = 16 MFMAs' worth of cycles
                x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f);
                x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
            }
        }
        s += x0 + x1 + x2 + x3;
    }
    if (s == 123.456f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    const int iters = 2000;
    const char* names[] = {"all 16 waves MFMA", "all 16 waves VALU", "8 waves MFMA + 8 waves VALU", "8 waves MFMA, 8 idle", "8 idle, 8 waves VALU"};
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<256, 1024>>>(d, iters, mode, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) k<<<256, 1024>>>(d, iters, mode, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-30s %8.3f ms per launch\n", names[mode], ms / 3);
    }
    return 0;
}
