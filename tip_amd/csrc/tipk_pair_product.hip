// The dense half of the pair-form D-D forward pass (include/tipk.h section 2c):
//
//     slab[g][v, c] = sum_{u in group g} sum_b C[u][v][b] * XB[u][b][c]          (then tipk_sum_slabs over g)
//
// C [n_src][n_dst][NB] are the pair cells the wave-stream gather wrote (C[u][v][:] = sum of att[r, :] over the
// relations linking u -> v; 53 MB at BioSNAP, 70 % of it cells of unlinked pairs that stay zero), XB[u] = the
// [NB x d] block of X . basis of source node u.  It is one pass over C with 0.86 GFLOP of fp32 MFMA: as a tiled
// GEMM through LDS (tipk_gemm_f32, 486 workgroups x 8 K-steps behind barriers) it took 21 us, and so did a first
// version of this kernel with one wave per (32 rows, 8 source nodes): at 1.7 waves per SIMD the matrix pipe (0.46 us
// of work per node) sat idle while the next node's 8 KB of operands travelled (1 - 2 us).  Now a WORKGROUP owns
// (32 destination rows, one group of `group` = 4 x PP_UPW source nodes) and each of its four waves takes PP_UPW of
// the nodes: all of a wave's operands are requested at once, straight into the register layout of
// v_mfma_f32_32x32x2_f32 --
//   A = C[u][v0 .. v0+31][:]   4 KB contiguous, 64 bytes per lane (lane = row, the lane halves split the bases)
//   B = XB[u]                  [NB x d], NB / 2 dwords per lane (lane = column)
// (instruction kk multiplies base (NB / 2) * half + kk: a fixed permutation of the reduction index) -- 6.6 waves
// per SIMD keep the pipe fed, and the four partial tiles are added in wave order through LDS.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PP_UPW = 2;              // source nodes per wave

struct PpArgs {
    const float* cells; const float* xb; float* slabs;
    int n_src, n_dst, d, group;        // n_src: multiple of group (padded blocks are zero); group = 4 * PP_UPW
    int row_tiles, n_groups;
    int symmetric;                     // cells hold only source <= destination: C[u][v] for v < u is read at C[v][u]
};

template <int NB>
__global__ __launch_bounds__(256) void pair_product_kernel(PpArgs a) {
    constexpr int KH = NB / 2;                          // bases per lane half
    __shared__ float red[4][1024];
    const int lane = threadIdx.x & 63, row = lane & 31, kh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = (int)blockIdx.x / a.row_tiles, rt = (int)blockIdx.x - g * a.row_tiles;
    const int v0 = rt * 32;
    const int v = v0 + row < a.n_dst ? v0 + row : a.n_dst - 1;            // clamped: rows past the end are not stored
    const int c = row < a.d ? row : a.d - 1;                              // clamped: columns past d are not stored
    const int u0 = g * a.group + wv * PP_UPW;
    // operand addresses of source node u: cells + ((u * n_dst + v) * NB + KH * kh), xb + ((u * NB + KH * kh) * d + c)
    const int64_t a_step = (int64_t)a.n_dst * NB, b_step = (int64_t)NB * a.d;
    const float* ap = a.cells + ((int64_t)u0 * a.n_dst + v) * NB + KH * kh;
    const float* at = a.cells + ((int64_t)v * a.n_dst + u0) * NB + KH * kh;        // the mirrored cell (v, u0)
    const float* bp = a.xb + ((int64_t)u0 * NB + KH * kh) * a.d + c;
    float av[PP_UPW][KH], bv[PP_UPW][KH];
#pragma unroll
    for (int q = 0; q < PP_UPW; ++q) {
#pragma unroll
        for (int i = 0; i < KH / 4; ++i) {
            // symmetric graph: only cells with source <= destination exist (the gather did half the work)
            const bool mirrored = a.symmetric && v < u0 + q && u0 + q < a.n_dst;
            const float4 t = tipk_ld4(mirrored ? at + q * NB + 4 * i : ap + q * a_step + 4 * i);
            av[q][4 * i] = t.x; av[q][4 * i + 1] = t.y; av[q][4 * i + 2] = t.z; av[q][4 * i + 3] = t.w;
        }
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) bv[q][kk] = bp[q * b_step + (int64_t)kk * a.d];
    }
    __builtin_amdgcn_sched_barrier(0);                  // all operands are requested before the first multiply
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int q = 0; q < PP_UPW; ++q)
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][kk], bv[q][kk], acc, 0, 0, 0);
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + row] = acc[r];
    __syncthreads();
    float* o = a.slabs + ((int64_t)g * a.n_dst + v0) * a.d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {                        // 1024 outputs / 256 threads, partials added in wave order
        const int i = e * 256 + (int)threadIdx.x;
        const int rr = i >> 5, cc = i & 31;
        const float s = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        if (v0 + rr < a.n_dst && cc < a.d) o[(int64_t)rr * a.d + cc] = s;
    }
}

}  // namespace

extern "C" int tipk_pair_product_supported(int n_bases, int d) {
    return (n_bases == 8 || n_bases == 16 || n_bases == 32) && d >= 1 && d <= 32;
}

extern "C" int tipk_pair_product(const float* cells, const float* xb, int64_t n_src, int64_t n_dst, int n_bases, int d,
                                 int group, int symmetric, float* slabs, tipk_stream_t stream) {
    if (!cells || !xb || !slabs || n_src <= 0 || n_dst <= 0 || group != 4 * PP_UPW || n_src % group != 0) return TIPK_EINVAL;
    if (!tipk_pair_product_supported(n_bases, d)) return TIPK_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(cells) & 15) || n_src * n_dst * n_bases >= (1LL << 40)) return TIPK_EINVAL;
    PpArgs a;
    a.cells = cells; a.xb = xb; a.slabs = slabs;
    a.n_src = (int)n_src; a.n_dst = (int)n_dst; a.d = d; a.group = group; a.symmetric = symmetric != 0;
    a.row_tiles = (int)tipk_ceil_div(n_dst, 32);
    a.n_groups = (int)(n_src / group);
    const int64_t blocks = (int64_t)a.row_tiles * a.n_groups;
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n_bases == 32) hipLaunchKernelGGL(pair_product_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (n_bases == 16) hipLaunchKernelGGL(pair_product_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pair_product_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    TIPK_RETURN_LAUNCH();
}
