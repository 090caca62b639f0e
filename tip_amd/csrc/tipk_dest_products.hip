// Forward pass of an R-GCN layer on LARGE node sets without materialising Y = att . XB (include/tipk.h section 2f;
// src/layers.py:159-180).  With W_r = sum_b att[r, b] basis_b:
//
//     agg[v, :] = sum_{e -> v} X[src_e] W_{r_e} = sum_b ( sum_{e -> v} att[r_e, b] X[src_e, :] ) basis_b = sum_b T[b, v, :] basis_b
//
// Y [R N x d_out] is 10 GB at config 5 (N = 10 000, R = 2 000, d = 128): written by one product and gathered back by the
// aggregation -- 2.5 + 3.1 ms per layer, both HBM / Infinity-Cache bound.  T [bases][N][d_in] is 164 MB, and the sum that
// builds it is a matrix product per DESTINATION whose reduction runs over the node's incoming edges (5 000 at config 5):
// A = rows of att gathered by the edges' relations (a 256 KB table), B = rows of X gathered by their sources (5 MB), both
// L2-resident -- 0.41 TFLOP of fp32 MFMA per layer instead of 0.16, but no 10 GB round trip.  The second product
// (sum_b T_b basis_b, 10 GFLOP) is a plain batch-reduced tipk_gemm_f32.
//
// One workgroup per (32-column tile of X, destination), launched tile by tile (heaviest destinations first inside a tile):
// while a tile is being worked on, what the chip gathers from is that tile's 128-byte slices of the X rows (1.3 MB at
// config 5) + att -- it stays in every XCD's 4 MB L2, which the whole X (5 MB) would not.  The four waves take every
// fourth batch of 32 edges: a wave loads the batch's edge words once (lane = edge), spreads (relation, source) by wave
// shuffles and issues 16 + 16 dword loads -- lane = base for att, lane = column for X: every load instruction touches
// two full 128-byte lines -- for 16 v_mfma_f32_32x32x2_f32; two batches in flight in statically named register sets.
// The waves' tiles are added through LDS in wave order: bitwise reproducible.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

struct DpfArgs {
    const float* x; int ld_x, d_in;
    const float* att; int ld_att, n_bases;
    const int4* node_desc;               // [n_nodes] {v, first edge, edges, 0}, heaviest first
    int n_nodes;
    const u32* edges;                    // [E] rel | src << rel_bits, grouped by destination
    int rel_bits;
    float* t; int64_t t_sb, t_sv;        // T element (b, v, i) at t[b * t_sb + v * t_sv + i]
};

__device__ __forceinline__ float dpf_ldg(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

__global__ __launch_bounds__(256) void dest_products_kernel(DpfArgs a) {
    __shared__ float red[4 * 1024];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = lane & 31, kh = lane >> 5;
    const int tile = (int)blockIdx.x / a.n_nodes;                      // (tile-major launch order)
    const int c0 = tile * 32;                                          // this workgroup's 32 columns of X
    const int4 nd = a.node_desc[(int)blockIdx.x - tile * a.n_nodes];
    const int v = __builtin_amdgcn_readfirstlane(nd.x);
    const int e0 = __builtin_amdgcn_readfirstlane(nd.y);
    const int cnt = __builtin_amdgcn_readfirstlane(nd.z);
    const u32 rmask = (1u << a.rel_bits) - 1u;
    const u32 ldx4 = (u32)a.ld_x * 4u, lda4 = (u32)a.ld_att * 4u;
    const int col = c0 + n < a.d_in ? c0 + n : a.d_in - 1;            // clamped: columns past the end are not stored
    const u32 col4 = (u32)col * 4u;
    const u32 b4 = (u32)(n < a.n_bases ? n : a.n_bases - 1) * 4u;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int n_batches = (cnt + 31) >> 5;
    // edge word of lane n of batch q (past the end: the node's last edge, multiplied by 0)
    auto getword = [&](int q) -> u32 {
        int i = q * 32 + n;
        i = i < cnt ? i : cnt - 1;
        return a.edges[e0 + i];
    };
    // loads are unconditional (a batch past the end re-reads the node's last edge); what does not exist is cleared
    // bitwise where it is USED (a `cond ? loaded : 0` next to a load makes hipcc wait for every load on the spot)
    auto load = [&](u32 wd, float (&av)[16], float (&bv)[16]) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const u32 we = (u32)__shfl((int)wd, 2 * kk + kh, 64);
            av[kk] = dpf_ldg(a.att, __umul24(we & rmask, lda4) + b4);         // (24-bit multiplies: v_mul_lo_u32 is quarter rate)
            bv[kk] = dpf_ldg(a.x, __umul24(we >> a.rel_bits, ldx4) + col4);
        }
    };
    auto mfma = [&](int q, const float (&av)[16], const float (&bv)[16]) {
        const int left = cnt - q * 32;                                // edges of this batch (<= 0: none)
        if (left >= 32) {                                             // (uniform)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc, 0, 0, 0);
        } else if (left > 0) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int keep = 2 * kk + kh < left ? -1 : 0;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(__float_as_int(av[kk]) & keep), bv[kk], acc, 0, 0, 0);
            }
        }
    };
    if (cnt > 0) {
        float aX[16], bX[16], aY[16], bY[16];
        u32 wX = getword(w), wY = getword(w + 4);
        load(wX, aX, bX);
        for (int q = w; q < n_batches; q += 8) {
            wX = getword(q + 8);
            load(wY, aY, bY);
            __builtin_amdgcn_sched_barrier(0);
            mfma(q, aX, bX);
            __builtin_amdgcn_sched_barrier(0);
            wY = getword(q + 12);
            load(wX, aX, bX);
            __builtin_amdgcn_sched_barrier(0);
            mfma(q + 4, aY, bY);                                      // (a batch past the end is skipped)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the waves' tiles -> one, in wave order; element e = reg * 64 + lane <-> row (base) (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5),
    // column lane & 31 of the MFMA result
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w * 1024 + r * 64 + lane] = acc[r];
    __syncthreads();
    float* o = a.t + (int64_t)v * a.t_sv + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = t + j * 256;
        const float s = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
        const int r = e >> 6, l = e & 63;
        const int b = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), c = l & 31;
        if (b < a.n_bases && c0 + c < a.d_in) o[(int64_t)b * a.t_sb + c] = s;
    }
}

}  // namespace

extern "C" int tipk_rgcn_dest_products_supported(int64_t n_nodes, int64_t n_rel, int n_bases, int d_in) {
    int bits = 1;
    while ((1LL << bits) < n_rel) ++bits;
    return n_bases >= 1 && n_bases <= 32 && d_in >= 1 && d_in <= 1024 && n_nodes > 0 && n_nodes <= (1LL << (32 - bits)) &&
           n_nodes < (1LL << 24) && n_rel < (1LL << 24) ? bits : 0;
}

extern "C" int tipk_rgcn_dest_products(const float* x, int64_t ld_x, int d_in, const float* att, int64_t ld_att, int n_bases,
                                       int64_t n_nodes, int64_t n_rel, const int32_t* node_desc, const uint32_t* edges,
                                       float* t, int64_t t_sb, int64_t t_sv, tipk_stream_t stream) {
    const int bits = tipk_rgcn_dest_products_supported(n_nodes, n_rel, n_bases, d_in);
    if (!bits) return TIPK_EUNSUPPORTED;
    if (!x || !att || !node_desc || !edges || !t || ld_x < d_in || ld_att < n_bases || (reinterpret_cast<uintptr_t>(node_desc) & 15))
        return TIPK_EINVAL;
    if (n_nodes * ld_x * 4 >= (1LL << 32) || n_rel * ld_att * 4 >= (1LL << 32) || ld_x * 4 >= (1LL << 24) || ld_att * 4 >= (1LL << 24))
        return TIPK_EUNSUPPORTED;                                       // 32-bit byte offsets from 24-bit multiplies
    DpfArgs a;
    a.x = x; a.ld_x = (int)ld_x; a.d_in = d_in; a.att = att; a.ld_att = (int)ld_att; a.n_bases = n_bases;
    a.node_desc = reinterpret_cast<const int4*>(node_desc); a.n_nodes = (int)n_nodes; a.edges = edges; a.rel_bits = bits;
    a.t = t; a.t_sb = t_sb; a.t_sv = t_sv;
    const int64_t grid = n_nodes * tipk_ceil_div(d_in, 32);
    if (grid > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(dest_products_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    TIPK_RETURN_LAUNCH();
}
