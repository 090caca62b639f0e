// Microbenchmark: throughput of LDS atomics on gfx950 by data type, random rows of a 645x17 image.
// hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(const int* __restrict__ rows, int n_iter, float* out) {
    extern __shared__ unsigned long long lds64[];
    float* f = (float*)lds64;
    unsigned* u = (unsigned*)lds64;
    const int t = threadIdx.x;
    for (int i = t; i < 645 * 17 * 2; i += 1024) u[i] = 0;
    __syncthreads();
    const int sub = t & 3, slot = t >> 2;
    for (int it = 0; it < n_iter; ++it) {
        const int r = rows[(blockIdx.x * n_iter + it) * 256 + slot];
        const int a = r * 17 + sub * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) atomicAdd(f + a + j, 1.0f);
            if (MODE == 1) atomicAdd(u + a + j, 1u);
            if (MODE == 2) atomicAdd(lds64 + a + j, 1ull);
            if (MODE == 3) f[a + j] += 1.0f;                       // racy plain RMW (upper bound)
            if (MODE == 4) atomicMax(u + a + j, (unsigned)it);
            if (MODE == 5) __hip_atomic_fetch_add(f + a + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (t == 0) out[blockIdx.x] = f[3] + (float)u[5];
}

int main() {
    const int n_iter = 512, blocks = 256;
    int* rows; float* out;
    hipMalloc(&rows, sizeof(int) * blocks * n_iter * 256);
    hipMalloc(&out, sizeof(float) * blocks);
    int* h = (int*)malloc(sizeof(int) * blocks * n_iter * 256);
    unsigned s = 12345;
    for (int i = 0; i < blocks * n_iter * 256; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % 645; }
    hipMemcpy(rows, h, sizeof(int) * blocks * n_iter * 256, hipMemcpyHostToDevice);
    const char* names[] = {"ds_add_f32 (atomicAdd float)", "ds_add_u32", "ds_add_u64", "plain rmw f32 (racy)", "ds_max_u32", "fetch_add f32 wg-scope"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            size_t lds = 645 * 17 * 8;
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), lds, 0, rows, n_iter, out);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), lds, 0, rows, n_iter, out);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), lds, 0, rows, n_iter, out);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), lds, 0, rows, n_iter, out);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), lds, 0, rows, n_iter, out);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), lds, 0, rows, n_iter, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double ops = (double)blocks * n_iter * 1024 * 4;
            if (rep) printf("%-32s %8.3f ms  %7.1f G lane-ops/s  %.2f lane-ops/clk/CU\n", names[mode], ms, ops / ms / 1e6,
                            ops / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
