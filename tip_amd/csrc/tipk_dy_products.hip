// Both consumers of dY in ONE pass (include/tipk.h section 2b):
//
//     dXB [b, c]  = sum_r att[r, b] * dY[r, c]          (b < NB <= 32 bases, c < NC = nodes * channels)
//     datt[r, b]  = sum_c dY[r, c]  * XB[b, c]
//
// dY [R x NC] is the largest tensor of the backward pass (91 MB at BioSNAP layer 1).  As two GEMMs it is
// read twice and each product runs at ~2.7 TB/s (tools/bench_gemm.py); here every 32 x 32 tile of dY is
// loaded once, multiplied by att^T in the register layout it arrives in (B operand: lane = column), and
// transposed through a wave-private LDS tile to serve as the A operand of the second product.
//
// Workgroup = 16 waves = 512 consecutive columns (wave w owns columns c0 .. c0+31) x one range of rows:
//   * dXB: the wave keeps its [NB x 32] accumulator over the whole row range; one slab per row range;
//   * datt: per 32-row tile the 16 waves' [32 x NB] partial products are added through LDS (fixed order)
//     and written as one slab per column chunk.
// Both slab sets are finished by tipk_sum_slabs_group -- ordered sums, reproducible.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

constexpr int DP_WAVES = 16;
constexpr int DP_CHUNK = DP_WAVES * 32;                 // columns per workgroup
constexpr int DP_TLD = 33;                              // transpose tile row stride (conflict-free both ways)

struct DpArgs {
    const float* dy; const float* att; const float* xb;
    int R, NC, NB;
    int64_t ld_dy, ld_att, ld_xb;
    float* dxb; float* datt;                            // slabs: [s_r][NB x NC], [s_c][R x NB]
    const uint32_t* used; int cols_per_node, n_nodes;   // nullable: bit (r & 31) of used[(r >> 5) * n_nodes + c / cols_per_node]
                                                        // = row r of dY holds data in the columns of that node; rows whose
                                                        // bit is clear are NOT read as data (the producer never wrote them)
    int rows_per_range;                                 // multiple of 32
    int dbg;                                            // TIPK_DP_DEBUG: 1 no loads after the first tile, 2 no 2nd product, 4 no 1st
};

__device__ __forceinline__ float ldg(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float and_mask(float v, u32 mask) { return __uint_as_float(__float_as_uint(v) & mask); }

template <bool MASKED>
__global__ __launch_bounds__(1024) void dy_products_kernel(DpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row = lane & 31, kh = lane >> 5;
    float* tile = lds + w * (32 * DP_TLD);              // wave-private transpose tile
    float* red = lds + DP_WAVES * (32 * DP_TLD);        // [16 waves][32 x 32] partial datt tiles
    const int chunk = blockIdx.x, range = blockIdx.y;
    const int c0 = chunk * DP_CHUNK + w * 32;
    const int r_lo = range * a.rows_per_range;
    const int r_hi = r_lo + a.rows_per_range < a.R ? r_lo + a.rows_per_range : a.R;
    const int NC = a.NC, NB = a.NB;
    const int col = c0 + row;
    const u32 col_b = (u32)(col < NC ? col : NC - 1) * 4u;
    const u32 ld_dy = (u32)a.ld_dy * 4u, ld_att = (u32)a.ld_att * 4u;
    const bool wave_on = c0 < NC;                       // the last chunk may have idle waves (they still join barriers)

    // B operand of the second product: XB[b = lane & 31][c0 + 2 kk + kh], zero outside (b, c) range
    float xbv[16];
    {
        const u32 b_off = (u32)(row < NB ? row : NB - 1) * (u32)a.ld_xb * 4u;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int c = c0 + 2 * kk + kh;
            const float v = ldg(a.xb, b_off + (u32)(c < NC ? c : NC - 1) * 4u);
            xbv[kk] = and_mask(v, (row < NB && c < NC) ? 0xffffffffu : 0u);
        }
    }
    f32x16 acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.f;

    float dyv[16], atv[16];
    // Row mask (a.used): rows (relation, node) without edges are never written by the transposed gather.  Their
    // loads are REDIRECTED to a row of the same tile and column that does hold data (a line this lane fetches
    // anyway: no extra traffic, and no exec-masked loads, which hipcc serialises), and the value is cleared
    // bitwise afterwards.  The mask word of a tile is requested one tile ahead of the addresses it steers.
    uint32_t uw_cur = 0xffffffffu, uw_nxt = 0xffffffffu;     // this lane's rows: bit j = row 2 j + kh ... after the shift below
    const u32 used_b = MASKED ? (u32)((col < NC ? col : NC - 1) / a.cols_per_node) * 4u : 0u;
    const int last_word = (r_hi - 1) >> 5;
    auto mask_word = [&](int r0) {
        const int wd = (r0 >> 5) < last_word ? (r0 >> 5) : last_word;
        return __float_as_uint(ldg(reinterpret_cast<const float*>(a.used + (int64_t)wd * a.n_nodes), used_b));
    };
    auto load_tile = [&](int r0) {                      // dY tile in B-operand layout + att^T in A-operand layout
        // 64-bit (wave-uniform) address of the tile's first row + 32-bit offsets inside the 32 rows:
        // dY may be far larger than 4 GB (synthetic config: 10 GB)
        const float* dy_t = a.dy + (int64_t)r0 * a.ld_dy;
        const float* att_t = a.att + (int64_t)r0 * a.ld_att;
        u32 safe_off = 0;
        if (MASKED) safe_off = (uw_cur ? (u32)__builtin_ctz(uw_cur) : 0u) * ld_dy + col_b;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int r = r0 + 2 * kk + kh;
            const u32 rl = (u32)((r < r_hi ? r : r_hi - 1) - r0);
            u32 off = rl * ld_dy + col_b;
            if (MASKED) off = ((uw_cur >> (2 * kk + kh)) & 1u) ? off : safe_off;      // bits beyond r_hi are clear
            dyv[kk] = ldg(dy_t, off);
            atv[kk] = ldg(att_t, rl * ld_att + (u32)(row < NB ? row : NB - 1) * 4u);
        }
    };
    if (r_lo < r_hi) {
        if (MASKED) { uw_cur = mask_word(r_lo); uw_nxt = mask_word(r_lo + 32); }
        load_tile(r_lo);
    }
    for (int r0 = r_lo; r0 < r_hi; r0 += 32) {
        if (MASKED) {
            const int uwl = (int)(uw_cur >> kh);                      // bit 2 kk = this lane's row of step kk
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)                           // one signed 1-bit field extract = 0 / ~0
                dyv[kk] = and_mask(dyv[kk], (u32)__builtin_amdgcn_sbfe(uwl, 2 * kk, 1));
        }
        // (1) dXB += att^T . dY : rows beyond r_hi (and bases beyond NB) are zeroed in the A operand
        if (!TIPK_DBG(a.dbg & 4))
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const u32 mk = (r0 + 2 * kk + kh < r_hi && row < NB) ? 0xffffffffu : 0u;
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(atv[kk], mk), dyv[kk], acc1, 0, 0, 0);
        }
        // transpose the tile through LDS: element (r = 2 kk + kh, c = lane & 31) -> tile[c][r]
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) tile[row * DP_TLD + 2 * kk + kh] = dyv[kk];
        __builtin_amdgcn_sched_barrier(0);
        if (r0 + 32 < r_hi && !TIPK_DBG(a.dbg & 1)) {                       // next tile in flight during the second product
            if (MASKED) { uw_cur = uw_nxt; uw_nxt = mask_word(r0 + 64); }
            load_tile(r0 + 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (TIPK_DBG(a.dbg & 2)) continue;
        // (2) datt tile = dY . XB^T : A operand lane = row r, k = column c
        f32x16 acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float av = tile[(2 * kk + kh) * DP_TLD + row];   // same wave wrote it: program order suffices
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xbv[kk], acc2, 0, 0, 0);
        }
        // add the 16 waves' partial tiles in wave order
        __syncthreads();                                // red is free (previous tile has been summed)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * kh;
            red[w * 1024 + rr * 32 + row] = wave_on ? acc2[r] : 0.f;
        }
        __syncthreads();
        {
            float s = red[t];
#pragma unroll
            for (int q = 1; q < DP_WAVES; ++q) s += red[q * 1024 + t];
            const int rr = r0 + (t >> 5), b = t & 31;
            if (rr < r_hi && b < NB) a.datt[((int64_t)chunk * a.R + rr) * NB + b] = s;
        }
    }
    // dXB slab of this row range: C layout col = lane & 31 (column c), rows = bases
    if (wave_on && col < NC) {
        float* o = a.dxb + (int64_t)range * NB * (int64_t)NC;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (b < NB) o[(int64_t)b * NC + col] = acc1[r];
        }
    }
}

constexpr size_t DP_LDS = (size_t)(DP_WAVES * 32 * DP_TLD + DP_WAVES * 1024) * sizeof(float);

}  // namespace

extern "C" int tipk_rgcn_dy_products_plan(int64_t n_rel, int64_t n_cols, int n_bases, int* col_slabs, int* row_slabs) {
    if (!col_slabs || !row_slabs) return TIPK_EINVAL;
    *col_slabs = *row_slabs = 0;
    if (n_rel <= 0 || n_cols <= 0 || n_bases <= 0 || n_bases > 32 || n_rel > 0x7fffffffLL || n_cols > 0x1fffffffLL) return TIPK_OK;
    const int64_t s_c = tipk_ceil_div(n_cols, DP_CHUNK);  // the datt slabs are always n_bases / 512 = 6 % of dY
    if (s_c > 65535) return TIPK_OK;
    int64_t s_r = 256 / s_c;                            // about one workgroup per CU
    if (s_r < 1) s_r = 1;
    const int64_t per = tipk_ceil_div(tipk_ceil_div(n_rel, s_r), 32) * 32;
    s_r = tipk_ceil_div(n_rel, per);
    *col_slabs = (int)s_c;
    *row_slabs = (int)s_r;
    return TIPK_OK;
}

extern "C" int tipk_rgcn_dy_products(const float* dy, int64_t ld_dy, const float* att, int64_t ld_att, const float* xb,
                                     int64_t ld_xb, int64_t n_rel, int64_t n_cols, int n_bases, const uint32_t* row_used,
                                     int64_t n_nodes, float* dxb_slabs, float* datt_slabs, tipk_stream_t stream) {
    int s_c = 0, s_r = 0;
    const int rc = tipk_rgcn_dy_products_plan(n_rel, n_cols, n_bases, &s_c, &s_r);
    if (rc != TIPK_OK) return rc;
    if (s_c == 0) return TIPK_EUNSUPPORTED;
    if (!dy || !att || !xb || !dxb_slabs || !datt_slabs || ld_dy < n_cols || ld_att < n_bases || ld_xb < n_cols) return TIPK_EINVAL;
    // 32-bit byte offsets inside one 32-row tile of dY / att and inside XB
    if (32 * ld_dy >= (1LL << 30) || 32 * ld_att >= (1LL << 30) || n_bases * ld_xb >= (1LL << 30)) return TIPK_EUNSUPPORTED;
    DpArgs a;
    a.dy = dy; a.att = att; a.xb = xb;
    a.R = (int)n_rel; a.NC = (int)n_cols; a.NB = n_bases;
    a.ld_dy = ld_dy; a.ld_att = ld_att; a.ld_xb = ld_xb;
    a.dxb = dxb_slabs; a.datt = datt_slabs;
    a.used = row_used; a.n_nodes = (int)n_nodes; a.cols_per_node = 1;
    if (row_used) {
        if (n_nodes <= 0 || n_cols % n_nodes != 0) return TIPK_EINVAL;
        a.cols_per_node = (int)(n_cols / n_nodes);
    }
    a.rows_per_range = (int)(tipk_ceil_div(tipk_ceil_div(n_rel, s_r), 32) * 32);
    a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DP_DEBUG));
    auto kern = row_used ? dy_products_kernel<true> : dy_products_kernel<false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DP_LDS);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL(kern, dim3((unsigned)s_c, (unsigned)s_r), dim3(1024), DP_LDS, (hipStream_t)stream, a);
    TIPK_RETURN_LAUNCH();
}
