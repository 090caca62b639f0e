// The two large products of the basis decomposition, as streaming kernels (include/tipk.h section 2b):
//
//   expand:  Y[r, j]   = sum_b att[r, b] * XB[b, j]      (R x 32) . (32 x J)  -> writes R*J floats
//   reduce:  dXB[b, j] = sum_r att[r, b] * dY[r, j]      (32 x R) . (R x J)   -> reads  R*J floats
//
// with J = nodes * out_channels (20 640 ... 1.28 M) and exactly 32 bases (TIP's num_base).  Both move
// one 90 MB (BioSNAP) / 10 GB (synthetic) matrix once and are HBM-bound; the generic LDS-tiled GEMM
// reached 2.4 / 2.9 TB/s on them.  Here the MFMA operands come straight from their natural layouts:
//   * expand: XB's 32 x 64 column block sits in registers as the B operand (lane l holds
//     XB[2kk + (l>>5)][j0 + (l&31)]: coalesced 128-B reads); att is staged once per workgroup into LDS
//     k-major, so the A operand is a conflict-free ds_read_b32; the loop only issues MFMAs and stores.
//   * reduce: A[i=b][k=r] = att[r][b] and B[k=r][j] = dY[r][j] are both read with lanes running along
//     the contiguous index -- no LDS at all; K (relations) is split over blockIdx.y into slabs.
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NB = 32;                 // bases
constexpr int RC = 512;                // att rows staged per pass (expand): 32 x 516 floats = 66 KB
constexpr int RC_PAD = 4;

__global__ __launch_bounds__(256) void basis_expand_kernel(const float* __restrict__ att, int64_t n_rel,
                                                           const float* __restrict__ xb, int64_t n_cols,
                                                           float* __restrict__ y, int64_t rows_per_block) {
    __shared__ float As[NB][RC + RC_PAD];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    const int64_t r_lo = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r_hi = r_lo + rows_per_block < n_rel ? r_lo + rows_per_block : n_rel;
    const int kh = lane >> 5, jl = lane & 31;

    // B operand of both 32-column subtiles, all 16 k-steps: 32 registers, loaded once
    float breg[2][16];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int64_t j = j0 + s * 32 + jl;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) breg[s][kk] = j < n_cols ? xb[(int64_t)(2 * kk + kh) * n_cols + j] : 0.f;
    }
    for (int64_t rc = r_lo; rc < r_hi; rc += RC) {
        const int rows = (int)(r_hi - rc < RC ? r_hi - rc : RC);
        __syncthreads();
        // stage att[rc : rc+rows, 0:32] transposed (k-major); batch the loads before the LDS writes
        for (int base = 0; base < rows * 8; base += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int i = base + u * 256 + t;
                i = i < rows * 8 ? i : rows * 8 - 1;
                v[u] = tipk_ld4(att + (rc + (i >> 3)) * NB + (i & 7) * 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256 + t;
                if (i < rows * 8) {
                    const int r = i >> 3, c = (i & 7) * 4;
                    As[c + 0][r] = v[u].x; As[c + 1][r] = v[u].y; As[c + 2][r] = v[u].z; As[c + 3][r] = v[u].w;
                }
            }
        }
        __syncthreads();
        const int m_tiles = (rows + 31) >> 5;
        for (int mt = wid; mt < m_tiles; mt += 4) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            const int m = mt * 32 + jl;                 // A operand row of this lane (padding rows read stale
#pragma unroll                                          // LDS; their results are never stored)
            for (int kk = 0; kk < 16; ++kk) {
                const float av = As[2 * kk + kh][m];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[0][kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[1][kk], acc1, 0, 0, 0);
            }
            // C/D layout: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < rows) {
                    float* o = y + (rc + row) * n_cols + j0 + jl;
                    if (j0 + jl < n_cols) o[0] = acc0[r];
                    if (j0 + 32 + jl < n_cols) o[32] = acc1[r];
                }
            }
        }
    }
}

// One wave per 32-column subtile; blockIdx.y = K slab.  U k-steps (2 relations each) are loaded ahead.
__global__ __launch_bounds__(256) void basis_reduce_kernel(const float* __restrict__ att, int64_t n_rel,
                                                           const float* __restrict__ dy, int64_t n_cols,
                                                           float* __restrict__ slabs, int64_t rows_per_slab) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t j = ((int64_t)blockIdx.x * 4 + wid) * 32 + (lane & 31);
    const int kh = lane >> 5, bl = lane & 31;
    if (((int64_t)blockIdx.x * 4 + wid) * 32 >= n_cols) return;
    const int64_t r_lo = (int64_t)blockIdx.y * rows_per_slab;
    const int64_t r_hi = r_lo + rows_per_slab < n_rel ? r_lo + rows_per_slab : n_rel;
    const bool j_ok = j < n_cols;
    const int64_t jc = j_ok ? j : n_cols - 1;           // clamped: loads stay unconditional
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int U = 8;
    for (int64_t r0 = r_lo; r0 < r_hi; r0 += 2 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + 2 * u + kh;
            const int64_t rcl = r < r_hi ? r : r_hi - 1;
            av[u] = att[rcl * NB + bl];
            bv[u] = dy[rcl * n_cols + jc];
            if (r >= r_hi) { av[u] = 0.f; bv[u] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (!j_ok) return;
    float* o = slabs + (int64_t)blockIdx.y * NB * n_cols;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = (r & 3) + 8 * (r >> 2) + 4 * kh;
        o[(int64_t)b * n_cols + j] = acc[r];
    }
}

}  // namespace

extern "C" int tipk_basis_expand(const float* att, int64_t n_rel, int n_base, const float* xb, int64_t n_cols,
                                 float* y, tipk_stream_t stream) {
    if (n_rel < 0 || n_cols < 0) return TIPK_EINVAL;
    if (n_base != NB) return TIPK_EUNSUPPORTED;
    if (n_rel == 0 || n_cols == 0) return TIPK_OK;
    if (!att || !xb || !y || (reinterpret_cast<uintptr_t>(att) & 15)) return TIPK_EINVAL;
    const int64_t gx = tipk_ceil_div(n_cols, 64);
    int64_t gy = 1;                                     // row blocks: enough workgroups to fill the chip
    while (gx * gy < 1024 && n_rel / (gy * 2) >= 64) gy *= 2;
    const int64_t rows_per_block = tipk_ceil_div(tipk_ceil_div(n_rel, gy), 32) * 32;
    gy = tipk_ceil_div(n_rel, rows_per_block);
    if (gx > 0x7fffffffLL || gy > 65535) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(basis_expand_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, att,
                       n_rel, xb, n_cols, y, rows_per_block);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_basis_reduce_slabs(int64_t n_rel, int64_t n_cols) {
    // number of K slabs tipk_basis_reduce writes (caller sizes the workspace [slabs][32][n_cols])
    const int64_t waves = tipk_ceil_div(n_cols, 32);
    int64_t s = 1;
    while (waves * s < 2048 && n_rel / (s * 2) >= 128) s *= 2;
    return (int)s;
}

extern "C" int tipk_basis_reduce(const float* att, int64_t n_rel, int n_base, const float* dy, int64_t n_cols,
                                 float* slabs, tipk_stream_t stream) {
    if (n_rel < 0 || n_cols < 0) return TIPK_EINVAL;
    if (n_base != NB) return TIPK_EUNSUPPORTED;
    if (n_cols == 0) return TIPK_OK;
    if (!att || !dy || !slabs) return TIPK_EINVAL;
    const int64_t s = tipk_basis_reduce_slabs(n_rel, n_cols);
    const int64_t rows_per_slab = tipk_ceil_div(tipk_ceil_div(n_rel > 0 ? n_rel : 1, s), 2) * 2;
    const int64_t gx = tipk_ceil_div(tipk_ceil_div(n_cols, 32), 4);
    if (gx > 0x7fffffffLL || s > 65535) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(basis_reduce_kernel, dim3((unsigned)gx, (unsigned)s), dim3(256), 0, (hipStream_t)stream, att,
                       n_rel, dy, n_cols, slabs, rows_per_slab);
    TIPK_RETURN_LAUNCH();
}
