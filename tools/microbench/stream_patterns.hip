// Microbenchmark: how fast can a [1097 x 20640] fp32 matrix (91 MB) be read / written on gfx950 with
// the access patterns a GEMM operand layout forces?  Rows are 82 560 B apart.
//   hipcc --offload-arch=gfx950 -O3 stream_patterns.hip -o /tmp/sp && /tmp/sp
// read patterns (per wave, one "step" = what is issued before the data is consumed):
//   0  linear: the wave reads 16 consecutive KB (dwordx4, fully coalesced)            -- upper bound
//   1  tile32: 32 rows x 128 B with dword loads  (lane = column, 2 rows per load)     -- thin_m / thin_k operand
//   2  tile128: 32 rows x 512 B with dwordx4 loads (lane = 4 columns, 2 rows per load)
//   3  rowlane: 32 rows (lane = row) x 128 B with dwordx4 loads (16 B per row per load) -- kk operand
//   4  tile128 over 8 rows only (4 x4-loads per step, more steps)
// write patterns: 10 linear x4, 11 tile32 dword (2 rows x 128 B per store), 12 tile128 x4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int R = 1097, C = 20640;

template <int MODE>
__global__ __launch_bounds__(256) void rd(const float* __restrict__ a, float* __restrict__ out, int waves_total) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.f;
    if (MODE == 0) {
        const size_t total4 = (size_t)R * C / 4;
        for (size_t i = (size_t)gw * 64 + lane; i < total4; i += (size_t)waves_total * 64) {
            float4 v = reinterpret_cast<const float4*>(a)[i];
            acc += v.x + v.y + v.z + v.w;
        }
    } else if (MODE == 1) {                               // wave = (column tile of 32, row range); steps of 32 rows
        const int n_tiles = C / 32, splits = waves_total / n_tiles;
        const int nt = gw % n_tiles, sp = gw / n_tiles;
        if (sp >= splits) return;
        const int r_lo = sp * ((R + splits - 1) / splits), r_hi = min(R, r_lo + (R + splits - 1) / splits);
        for (int r0 = r_lo; r0 < r_hi; r0 += 32) {
            float v[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) { int r = min(r0 + 2 * kk + (lane >> 5), r_hi - 1); v[kk] = a[(size_t)r * C + nt * 32 + (lane & 31)]; }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc += v[kk];
        }
    } else if (MODE == 2 || MODE == 4) {                  // wave = (column tile of 128, row range)
        constexpr int RS = MODE == 2 ? 32 : 8;
        const int n_tiles = (C + 127) / 128, splits = waves_total / n_tiles;
        const int nt = gw % n_tiles, sp = gw / n_tiles;
        if (sp >= splits) return;
        const int r_lo = sp * ((R + splits - 1) / splits), r_hi = min(R, r_lo + (R + splits - 1) / splits);
        const int col = min(nt * 128 + (lane & 31) * 4, C - 4);
        for (int r0 = r_lo; r0 < r_hi; r0 += RS) {
            float4 v[RS / 2];
#pragma unroll
            for (int kk = 0; kk < RS / 2; ++kk) { int r = min(r0 + 2 * kk + (lane >> 5), r_hi - 1); v[kk] = *reinterpret_cast<const float4*>(a + (size_t)r * C + col); }
#pragma unroll
            for (int kk = 0; kk < RS / 2; ++kk) acc += v[kk].x + v[kk].y + v[kk].z + v[kk].w;
        }
    } else if (MODE == 3) {                               // wave = (row tile of 32, column range); lane = row
        const int m_tiles = (R + 31) / 32, splits = waves_total / m_tiles;
        const int mt = gw % m_tiles, sp = gw / m_tiles;
        if (sp >= splits) return;
        const int per = ((C / splits) + 31) / 32 * 32;
        const int c_lo = sp * per, c_hi = min(C, c_lo + per);
        const int r = min(mt * 32 + (lane & 31), R - 1);
        for (int c0 = c_lo; c0 < c_hi; c0 += 32) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { int c = min(c0 + 8 * q + 4 * (lane >> 5), c_hi - 4); v[q] = *reinterpret_cast<const float4*>(a + (size_t)r * C + c); }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
        }
    }
    if (acc == 12345.678f) out[gw] = acc;                 // never true: keeps the loads alive
}

template <int MODE>
__global__ __launch_bounds__(256) void wr(float* __restrict__ a, int waves_total, const float* __restrict__ src = nullptr) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (MODE == 10) {
        const size_t total4 = (size_t)R * C / 4;
        for (size_t i = (size_t)gw * 64 + lane; i < total4; i += (size_t)waves_total * 64)
            reinterpret_cast<float4*>(a)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    } else if (MODE == 11) {                              // wave = (row tile 32, 4 column tiles of 32): 16 dword stores per tile
        const int m_tiles = (R + 31) / 32, chunks = C / 128;
        const int mt = gw % m_tiles, ch = gw / m_tiles;
        if (ch >= chunks) return;
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) a[(size_t)row * C + ch * 128 + j * 32 + (lane & 31)] = (float)r;
            }
    } else if (MODE == 13 || MODE == 14) {                // tile32 stores behind a chain of 16 fp32 MFMAs (+ 16 operand loads)
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        const int m_tiles = (R + 31) / 32, chunks = C / 128;
        const int mt = gw % m_tiles, ch = gw / m_tiles;
        if (ch >= chunks) return;
        float av[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) av[kk] = (float)(lane + kk);
        for (int j = 0; j < 4; ++j) {
            float bv[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                bv[kk] = MODE == 14 ? src[(size_t)(2 * kk + (lane >> 5)) * C + ch * 128 + j * 32 + (lane & 31)] : (float)(kk + j);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) a[(size_t)row * C + ch * 128 + j * 32 + (lane & 31)] = acc[r];
            }
        }
    } else if (MODE == 20) {                              // as 14 with dwordx4 operand loads and stores: lane j owns columns 4j..4j+3
        typedef float f32x16 __attribute__((ext_vector_type(16)));    // = one column of each of four interleaved 32-column tiles
        const int m_tiles = (R + 31) / 32, chunks = C / 128;
        const int mt = gw % m_tiles, ch = gw / m_tiles;
        if (ch >= chunks) return;
        float av[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) av[kk] = (float)(lane + kk);
        float4 bv[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            bv[kk] = *reinterpret_cast<const float4*>(src + (size_t)(2 * kk + (lane >> 5)) * C + ch * 128 + (lane & 31) * 4);
        f32x16 acc[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s4][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk].w, acc[3], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < R) *reinterpret_cast<float4*>(a + (size_t)row * C + ch * 128 + (lane & 31) * 4) =
                make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
    } else if (MODE == 19) {                              // as 14, but every wave reads the SAME 16 KB of operands (certainly cached)
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        const int m_tiles = (R + 31) / 32, chunks = C / 128;
        const int mt = gw % m_tiles, ch = gw / m_tiles;
        if (ch >= chunks) return;
        float av[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) av[kk] = (float)(lane + kk);
        for (int j = 0; j < 4; ++j) {
            float bv[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                bv[kk] = src[(size_t)(2 * kk + (lane >> 5)) * 128 + j * 32 + (lane & 31)];
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) a[(size_t)row * C + ch * 128 + j * 32 + (lane & 31)] = acc[r];
            }
        }
    } else if (MODE == 17 || MODE == 18) {                // as 14 with NON-TEMPORAL stores (17) / + non-temporal... loads stay cached
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        const int m_tiles = (R + 31) / 32, chunks = C / 128;
        const int mt = gw % m_tiles, ch = gw / m_tiles;
        if (ch >= chunks) return;
        float av[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) av[kk] = (float)(lane + kk);
        for (int j = 0; j < 4; ++j) {
            float bv[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                bv[kk] = src[(size_t)(2 * kk + (lane >> 5)) * C + ch * 128 + j * 32 + (lane & 31)];
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) __builtin_nontemporal_store(acc[r], a + (size_t)row * C + ch * 128 + j * 32 + (lane & 31));
            }
        }
    } else if (MODE == 15 || MODE == 16) {                // as 14, all 64 operand loads issued before the first MFMA / store
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        const int m_tiles = (R + 31) / 32, chunks = C / 128;
        const int mt = gw % m_tiles, ch = gw / m_tiles;
        if (ch >= chunks) return;
        float av[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) av[kk] = (float)(lane + kk);
        float bv[4][16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                bv[j][kk] = src[(size_t)(2 * kk + (lane >> 5)) * C + ch * 128 + j * 32 + (lane & 31)];
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 16) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every load is home before the first store
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[j][kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) a[(size_t)row * C + ch * 128 + j * 32 + (lane & 31)] = acc[r];
            }
        }
    } else if (MODE == 12) {                              // same tile, lane = 4 columns: 16 x4 stores of 2 rows x 512 B
        const int m_tiles = (R + 31) / 32, chunks = C / 128;
        const int mt = gw % m_tiles, ch = gw / m_tiles;
        if (ch >= chunks) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < R) *reinterpret_cast<float4*>(a + (size_t)row * C + ch * 128 + (lane & 31) * 4) = make_float4(1.f, 2.f, 3.f, (float)r);
        }
    }
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}

int main() {
    float *a, *out;
    const size_t bytes = (size_t)R * C * 4;
    hipMalloc(&a, bytes); hipMalloc(&out, 1 << 22);
    hipMemset(a, 0, bytes);
    const double mb = bytes / 1e6;
#define RD(MODE, WAVES, NAME)                                                                   \
    { const int w = (WAVES); float us = timeit([&] { hipLaunchKernelGGL(rd<MODE>, dim3((w + 3) / 4), dim3(256), 0, 0, a, out, w); }); \
      printf("read  %-34s waves %6d  %7.1f us  %.2f TB/s\n", NAME, w, us, mb / us); }
#define WR(MODE, WAVES, NAME)                                                                   \
    { const int w = (WAVES); float us = timeit([&] { hipLaunchKernelGGL(wr<MODE>, dim3((w + 3) / 4), dim3(256), 0, 0, a, w); }); \
      printf("write %-34s waves %6d  %7.1f us  %.2f TB/s\n", NAME, w, us, mb / us); }
    {   // the same patterns over SIX buffers in turn (546 MB > the 256 MB Infinity Cache): what a step sees,
        // where Y / dY are written once and read once
        float* bufs[6];
        for (int i = 0; i < 6; ++i) { hipMalloc(&bufs[i], bytes); hipMemset(bufs[i], 0, bytes); }
        int it = 0;
        { float us = timeit([&] { hipLaunchKernelGGL(wr<10>, dim3(4096), dim3(256), 0, 0, bufs[it++ % 6], 16384); });
          printf("write linear x4, 6 rotating buffers            %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<11>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 6], 35 * 161); });
          printf("write tile32 dword, 6 rotating buffers         %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<13>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 6], 35 * 161, bufs[5]); });
          printf("write tile32 dword + 16 MFMA/tile, 6 rotating  %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<14>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 5], 35 * 161, bufs[5]); });
          printf("write tile32 dword + 16 loads + 16 MFMA/tile   %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<20>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 5], 35 * 161, bufs[5]); });
          printf("  same work with dwordx4 loads / stores        %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<19>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 5], 35 * 161, bufs[5]); });
          printf("  same as +16 loads, all waves read one 16 KB   %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<17>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 5], 35 * 161, bufs[5]); });
          printf("  same as +16 loads, NON-TEMPORAL stores       %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<15>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 5], 35 * 161, bufs[5]); });
          printf("  same, all 64 loads first                     %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(wr<16>, dim3((35 * 161 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 5], 35 * 161, bufs[5]); });
          printf("  same, all loads first + vmcnt(0) before MFMA %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(rd<0>, dim3(4096), dim3(256), 0, 0, bufs[it++ % 6], out, 16384); });
          printf("read  linear x4, 6 rotating buffers            %7.1f us  %.2f TB/s\n", us, mb / us); }
        { float us = timeit([&] { hipLaunchKernelGGL(rd<1>, dim3((645 * 8 + 3) / 4), dim3(256), 0, 0, bufs[it++ % 6], out, 645 * 8); });
          printf("read  tile32 dword, 6 rotating buffers         %7.1f us  %.2f TB/s\n", us, mb / us); }
    }
    RD(0, 4096, "linear x4") RD(0, 8192, "linear x4") RD(0, 16384, "linear x4")
    RD(1, 645 * 4, "tile32 dword, 4 row splits") RD(1, 645 * 8, "tile32 dword, 8 row splits") RD(1, 645 * 16, "tile32 dword, 16 row splits")
    RD(2, 162 * 8, "tile128 x4, 8 row splits") RD(2, 162 * 16, "tile128 x4, 16 row splits") RD(2, 162 * 32, "tile128 x4, 32 row splits")
    RD(4, 162 * 16, "tile128 x4 (8-row steps), 16 splits") RD(4, 162 * 32, "tile128 x4 (8-row steps), 32 splits")
    RD(3, 35 * 57, "rowlane x4, 57 col splits") RD(3, 35 * 114, "rowlane x4, 114 col splits") RD(3, 35 * 228, "rowlane x4, 228 col splits")
    WR(10, 4096, "linear x4") WR(10, 16384, "linear x4")
    WR(11, 35 * 161, "tile32 dword stores") WR(12, 35 * 161, "tile128 x4 stores")
    return 0;
}
