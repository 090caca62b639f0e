// The hand-over between the two R-GCN layers of the encoder in ONE launch (include/tipk.h section 2g; src/layers.py:545-548):
//
//     x1  = relu( 1/deg * sum_s slab_s + x0 root1 )          the ordered slab sum that ends layer 1's forward pass
//     XB2 = x1 basis2  (node-major, rows padded to 32 columns: the pair product's operand),   x1 root2
//
// Round 4 ran them as two launches on the critical path (4.9 + 6.6 us: a slab sum whose 645 x 32 result the next launch --
// 198 workgroups of a tiled GEMM with K = 32 -- read straight back).  The products are row-local (K = 32, 528 columns per
// row): the workgroup that finishes two rows of x1 -- 64 elements x 16 slab lanes, the layout and the order of additions of
// sum_slabs_kernel<16> -- multiplies them at once, a thread per output column, x1 broadcast from LDS, basis2 / root2 (66 KB)
// out of L2.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

struct SxArgs {
    const float* in; int64_t n_slabs, slab_stride; int n_rows;             // slabs [n_slabs][n_rows][32]
    const float* row_scale; const float* addend; int relu;
    float* x;                                                              // [n_rows][32]
    const float* basis; const float* root; int n_bases, d_out;             // [n_bases][32][d_out], [32][d_out]
    float* xb;                                                             // [..][n_bases][32]  (columns >= d_out untouched: zeros)
    float* xroot;                                                          // [n_rows][d_out]
};

__global__ __launch_bounds__(1024) void sum_slabs_xb_kernel(SxArgs a) {
    __shared__ float red[1024];
    __shared__ float xs[64];
    constexpr int lanes = 16, epb = 64;                                    // slab lanes per element, elements per workgroup
    const int t = threadIdx.x, e = t % epb, j = t / epb;
    const int64_t count = (int64_t)a.n_rows * 32;
    const int64_t i = (int64_t)blockIdx.x * epb + e;
    float s = 0.f;
    if (i < count) {
        const float* p = a.in + i;                                         // four loads in flight, original order of additions
        int64_t k = j;
        for (; k + 3 * lanes < a.n_slabs; k += 4 * lanes) {
            const float v0 = p[k * a.slab_stride], v1 = p[(k + lanes) * a.slab_stride];
            const float v2 = p[(k + 2 * lanes) * a.slab_stride], v3 = p[(k + 3 * lanes) * a.slab_stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < a.n_slabs; k += lanes) s += p[k * a.slab_stride];
    }
    red[t] = s;
    __syncthreads();
    if (j == 0) {
        float v = 0.f;
        if (i < count) {
            v = red[e];
            for (int q = 1; q < lanes; ++q) v += red[q * epb + e];
            if (a.row_scale) v *= a.row_scale[i / 32];
            if (a.addend) v += a.addend[i];
            if (a.relu) v = fmaxf(v, 0.f);
            a.x[i] = v;
        }
        xs[e] = v;
    }
    __syncthreads();
    // the two rows' products: thread = output column n of [XB (n_bases x d_out) | x root (d_out)], K = 32 in index order
    const int row0 = (int)blockIdx.x * 2;
    const int n_xb = a.n_bases * a.d_out, n_cols = n_xb + a.d_out;
    for (int n = t; n < n_cols; n += 1024) {
        const bool is_root = n >= n_xb;
        const int b = is_root ? 0 : n / a.d_out, c = is_root ? n - n_xb : n - b * a.d_out;
        const float* w = is_root ? a.root + c : a.basis + (int64_t)b * 32 * a.d_out + c;
        float wv[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) wv[k] = w[k * a.d_out];
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) { acc0 = fmaf(xs[k], wv[k], acc0); acc1 = fmaf(xs[32 + k], wv[k], acc1); }
        if (is_root) {
            a.xroot[(int64_t)row0 * a.d_out + c] = acc0;
            if (row0 + 1 < a.n_rows) a.xroot[(int64_t)(row0 + 1) * a.d_out + c] = acc1;
        } else {
            a.xb[((int64_t)row0 * a.n_bases + b) * 32 + c] = acc0;
            if (row0 + 1 < a.n_rows) a.xb[((int64_t)(row0 + 1) * a.n_bases + b) * 32 + c] = acc1;
        }
    }
}

}  // namespace

extern "C" int tipk_sum_slabs_xb(const float* slabs, int64_t n_slabs, int64_t slab_stride, int64_t n_rows, int d_in,
                                 const float* row_scale, const float* addend, int relu, float* x,
                                 const float* basis, const float* root, int n_bases, int d_out, float* xb, float* xroot,
                                 tipk_stream_t stream) {
    if (d_in != 32 || d_out < 1 || d_out > 32 || n_bases < 1) return TIPK_EUNSUPPORTED;
    if (!slabs || !x || !basis || !root || !xb || !xroot || n_slabs < 1 || n_rows < 1 || slab_stride < n_rows * 32) return TIPK_EINVAL;
    const int64_t blocks = tipk_ceil_div(n_rows, 2);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    SxArgs a;
    a.in = slabs; a.n_slabs = n_slabs; a.slab_stride = slab_stride; a.n_rows = (int)n_rows;
    a.row_scale = row_scale; a.addend = addend; a.relu = relu; a.x = x;
    a.basis = basis; a.root = root; a.n_bases = n_bases; a.d_out = d_out; a.xb = xb; a.xroot = xroot;
    hipLaunchKernelGGL(sum_slabs_xb_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, a);
    TIPK_RETURN_LAUNCH();
}
