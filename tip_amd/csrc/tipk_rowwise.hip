// Small row-wise glue kernels (include/tipk.h section 3).  All are HBM/L2 streaming passes over
// N x d matrices that are tiny next to the edge lists; they exist so that the path needs no torch
// elementwise launches between the aggregation kernels.
#include "tipk_common.h"

namespace {

// 32x32 tiles through LDS (+1 pad: conflict-free column reads), coalesced on both sides.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t rows, int64_t cols,
                                                        float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t r = r0 + ty + j, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + j][tx] = in[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int64_t c = c0 + ty + j, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + j];
    }
}

__global__ __launch_bounds__(256) void rows_affine_kernel(const float* __restrict__ in, int64_t ld_in,
                                                          const float* __restrict__ row_mul,
                                                          const float* __restrict__ row_div,
                                                          const float* __restrict__ gate, int64_t ld_gate,
                                                          float* __restrict__ out, int64_t ld_out, int64_t rows,
                                                          int64_t cols, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i - r * cols;
    float v = in[r * ld_in + c];
    if (row_mul) v *= row_mul[r];
    if (row_div) v /= row_div[r];
    if (gate && !(gate[r * ld_gate + c] > 0.f)) v = 0.f;
    float* o = out + r * ld_out + c;
    *o = accumulate ? *o + v : v;
}

// Stage 1 of the column sum: workgroup g adds rows g, g+G, ... (fixed order) for every column.
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ in, int64_t ld_in,
                                                              int64_t rows, int64_t cols,
                                                              float* __restrict__ scratch) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    const int cl = cols < 256 ? (int)cols : 256;       // columns handled per pass
    const int lanes_r = 256 / cl;                      // row lanes per column (>= 1)
    for (int64_t cb = 0; cb < cols; cb += cl) {
        const int c = t % cl, rl = t / cl;
        float s = 0.f;
        if (rl < lanes_r && cb + c < cols)
            for (int64_t r = (int64_t)blockIdx.x * lanes_r + rl; r < rows; r += (int64_t)gridDim.x * lanes_r)
                s += in[r * ld_in + cb + c];
        red[t] = s;
        __syncthreads();
        if (rl == 0 && cb + c < cols) {
            float tot = 0.f;
            for (int k = 0; k < lanes_r; ++k) tot += red[k * cl + c];
            scratch[(int64_t)blockIdx.x * cols + cb + c] = tot;
        }
        __syncthreads();
    }
}

// ReLU backward gate fused with stage 1 of the column sum of its result (the bias gradient of the
// layer): workgroup g handles rows g*R + rl, stepping by G*R (R = 256 / cols row lanes), writes
// out = in * (gate > 0) and its partial column sums -> scratch[g][c] (fixed order).
__global__ __launch_bounds__(256) void gate_colsum_kernel(const float* __restrict__ in, int64_t ld_in,
                                                          const float* __restrict__ gate, int64_t ld_gate,
                                                          float* __restrict__ out, int64_t ld_out, int64_t rows,
                                                          int cols, float* __restrict__ scratch) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    const int lanes_r = 256 / cols;
    const int c = t % cols, rl = t / cols;
    float s = 0.f;
    if (rl < lanes_r)
        for (int64_t r = (int64_t)blockIdx.x * lanes_r + rl; r < rows; r += (int64_t)gridDim.x * lanes_r) {
            float v = in[r * ld_in + c];
            if (!(gate[r * ld_gate + c] > 0.f)) v = 0.f;
            out[r * ld_out + c] = v;
            s += v;
        }
    red[t] = s;
    __syncthreads();
    if (rl == 0) {
        float tot = 0.f;
        for (int k = 0; k < lanes_r; ++k) tot += red[k * cols + c];
        scratch[(int64_t)blockIdx.x * cols + c] = tot;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The drug feature mix of FMEncoder (src/layers.py:526-539) in ONE forward launch:
//     x0 = cat(xd / d_norm, mean W)    or    x0 = xd / d_norm + mean W        (mean [N x p], W [p x q], p, q <= 64)
// The dense map of MyHierarchyConv (src/layers.py:239) is 645 x 16 x 16 at BioSNAP size: as a GEMM launch plus a
// row-scaling launch the arithmetic is nothing and the second hand-over is 5 us.  One thread per output element; W in LDS.
// (The backward pass keeps its grouped split-K product + ordered slab sum: d W = mean^T g_pd is a reduction over all
// rows, and ONE workgroup doing it -- rows straight from global memory, 128-row LDS tiles, or all rows in one LDS
// tile -- took 22 ... 30 us inside the step against 17 us for the three launches it replaced.)
constexpr int DM_MAX = 64;

__global__ __launch_bounds__(256) void drug_mix_fwd_kernel(const float* __restrict__ xd, int64_t ld_xd,
                                                           const float* __restrict__ d_norm,
                                                           const float* __restrict__ mean, int64_t ld_mean,
                                                           const float* __restrict__ w, int p, int q, int ne, int cat,
                                                           float* __restrict__ out, int64_t ld_out, int64_t rows) {
    __shared__ float wl[DM_MAX * DM_MAX];
    for (int i = threadIdx.x; i < p * q; i += 256) wl[i] = w[i];
    __syncthreads();
    const int cols = cat ? ne + q : ne;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    float v = 0.f;
    if (c < ne) {
        v = xd[r * ld_xd + c];
        if (d_norm) v /= d_norm[r];
    }
    const int j = cat ? c - ne : c;
    if (j >= 0 && j < q) {
        float s = 0.f;
        for (int k = 0; k < p; ++k) s = fmaf(mean[r * ld_mean + k], wl[k * q + j], s);
        v += s;
    }
    out[r * ld_out + c] = v;
}

}  // namespace

extern "C" int tipk_gate_colsum_groups(int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0 || cols > 256) return 0;
    const int64_t lanes_r = 256 / cols;
    int64_t groups = tipk_ceil_div(rows, lanes_r * 4);
    if (groups > 256) groups = 256;                   // the ordered sum of the groups is a single workgroup's chain
    return (int)groups;
}

extern "C" int tipk_gate_colsum(const float* in, int64_t ld_in, const float* gate, int64_t ld_gate, float* out,
                                int64_t ld_out, int64_t rows, int64_t cols, float* scratch, tipk_stream_t stream) {
    if (rows < 0 || cols < 0) return TIPK_EINVAL;
    if (rows == 0 || cols == 0) return TIPK_OK;
    if (!in || !gate || !out || !scratch) return TIPK_EINVAL;
    const int groups = tipk_gate_colsum_groups(rows, cols);
    if (groups == 0) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(gate_colsum_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, in, ld_in, gate,
                       ld_gate, out, ld_out, rows, (int)cols, scratch);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_transpose(const float* in, int64_t rows, int64_t cols, float* out, tipk_stream_t stream) {
    if (rows < 0 || cols < 0) return TIPK_EINVAL;
    if (rows == 0 || cols == 0) return TIPK_OK;
    if (!in || !out) return TIPK_EINVAL;
    const int64_t gx = tipk_ceil_div(cols, 32), gy = tipk_ceil_div(rows, 32);
    if (gx > 0x7fffffffLL || gy > 65535) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, in, rows,
                       cols, out);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_rows_affine(const float* in, int64_t ld_in, const float* row_mul, const float* row_div,
                                const float* gate, int64_t ld_gate, float* out, int64_t ld_out, int64_t rows,
                                int64_t cols, int accumulate, tipk_stream_t stream) {
    if (rows < 0 || cols < 0) return TIPK_EINVAL;
    if (rows == 0 || cols == 0) return TIPK_OK;
    if (!in || !out) return TIPK_EINVAL;
    const int64_t blocks = tipk_ceil_div(rows * cols, 256);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(rows_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ld_in,
                       row_mul, row_div, gate, ld_gate, out, ld_out, rows, cols, accumulate);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_sum_slabs(const float*, int64_t, int64_t, int64_t, float, int, float*, tipk_stream_t);

extern "C" int tipk_col_sum(const float* in, int64_t ld_in, int64_t rows, int64_t cols, float* scratch, float* out,
                            tipk_stream_t stream) {
    if (rows < 0 || cols < 0) return TIPK_EINVAL;
    if (cols == 0) return TIPK_OK;
    if (!in || !out || !scratch) return TIPK_EINVAL;
    const int cl = cols < 256 ? (int)cols : 256;
    const int lanes_r = 256 / cl;
    int64_t groups = tipk_ceil_div(rows, (int64_t)lanes_r * 8);
    if (groups > 256) groups = 256;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, in, ld_in,
                       rows, cols, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tipk_hip_status(e);
    return tipk_sum_slabs(scratch, groups, cols, cols, 1.0f, 0, out, stream);
}

extern "C" int tipk_drug_mix_fwd(const float* xd, int64_t ld_xd, const float* d_norm, const float* mean, int64_t ld_mean,
                                 const float* w, int p, int q, int64_t rows, int ne, int cat, float* out, int64_t ld_out,
                                 tipk_stream_t stream) {
    if (rows < 0 || ne < 0 || p <= 0 || q <= 0 || p > DM_MAX || q > DM_MAX || (!cat && q != ne)) return TIPK_EINVAL;
    if (rows == 0) return TIPK_OK;
    if (!xd || !mean || !w || !out) return TIPK_EINVAL;
    const int64_t cols = cat ? ne + q : ne;
    const int64_t blocks = tipk_ceil_div(rows * cols, 256);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(drug_mix_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xd, ld_xd, d_norm, mean,
                       ld_mean, w, p, q, ne, cat, out, ld_out, rows);
    TIPK_RETURN_LAUNCH();
}
