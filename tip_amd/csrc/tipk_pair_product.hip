// The dense half of the pair-form D-D forward pass (include/tipk.h section 2c):
//
//     slab[g][v, c] = sum_{u in group g} sum_b C[u][v][b] * XB[u][b][c]          (then tipk_sum_slabs over g)
//
// C [n_src][n_dst][NB] are the pair cells the wave-stream gather wrote (C[u][v][:] = sum of att[r, :] over the
// relations linking u -> v; 53 MB at BioSNAP, 70 % of it cells of unlinked pairs that stay zero), XB[u] = the
// [NB x d] block of X . basis of source node u.  It is one pass over C with 0.86 GFLOP of fp32 MFMA: as a tiled
// GEMM through LDS (tipk_gemm_f32, 486 workgroups x 8 K-steps behind barriers) it took 21 us, and so did two
// wave-local versions in which every wave fetched its own XB blocks (dword loads, 54 MB through the texture
// path) and had 2 blocks of work: loads 10.8 + 5.1 us and MFMA 10.0 us added up (tools/microbench/pp_variants.hip).
// Now a WORKGROUP owns (128 destination rows, one group of PP_GROUP source nodes): the group's XB blocks (32 KB)
// are staged in LDS once, coalesced; each of the 4 waves owns 32 of the rows for ALL the group's nodes, so its
// sum needs no cross-wave reduction, and walks the nodes with the next two nodes' cell rows in flight:
//   A = C[u][v0 .. v0+31][:]   4 KB contiguous, 64 bytes per lane, straight into the register layout of
//                              v_mfma_f32_32x32x2_f32 (lane = row, the lane halves split the bases; mirrored cells
//                              of a symmetric graph are read at their transposed place)
//   B = XB[u]                  from LDS (lane = column), conflict-free ds_read_b32; XB rows are stored padded to 32
//                              columns (zeros beyond d), so a group's blocks are one contiguous 32 KB copy
// (instruction kk multiplies base (NB / 2) * half + kk: a fixed permutation of the reduction index).
#include <stdlib.h>
#include <type_traits>
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PP_GROUP = 8;            // source nodes per workgroup = per slab

struct PpArgs {
    const float* cells; const float* xb; float* slabs;
    int n_src, n_dst, d, group;        // n_src: multiple of group (padded blocks are zero)
    int row_tiles, n_groups;           // row_tiles: tiles of 128 destination rows
    int symmetric;                     // cells hold only source <= destination: C[u][v] for v < u is read at C[v][u]
    const uint32_t* links;             // nullable: [n_src][ceil(n_dst / 32)] bit r of word (u, t) = pair (u, 32 t + r) is linked
    const float* zeros;                // >= NB * 4 bytes of zeros (read in place of the cell of an unlinked pair)
    float* xbt;                        // nullable: [n_dst][d][NB] -- XB written back base-innermost (by the row-tile-0 workgroups)
    int dbg;                           // debug builds ("dp_debug"): 256 no cell fetches (every lane reads the zero block), 512 no MFMAs,
};                                     // 1024 no XB staging, 2048 no stores

// NARROW (d <= 16): the same walk on v_mfma_f32_16x16x4_f32 -- two blocks of 16 rows per wave, 16 columns: half the matrix-pipe
// time of a half-empty 32-column tile (the kernel is bound by that pipe: profiles/r06_experiments.md section 8)
template <int NB, bool NARROW>
__global__ __launch_bounds__(256) void pair_product_kernel(PpArgs a) {
    constexpr int KH = NB / 2;                          // bases per lane half
    __shared__ __attribute__((aligned(16))) float xbl[PP_GROUP * NB * 32];     // [node][base][32 columns], columns >= d are zero
    const int t = threadIdx.x, lane = t & 63, row = lane & 31, kh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = (int)blockIdx.x / a.row_tiles, rt = (int)blockIdx.x - g * a.row_tiles;
    const int u0 = g * PP_GROUP;
    // the link words of this wave's 32 destination rows, one per source node of the group (requested first: they steer the
    // cell loads, and travel while XB is staged)
    const int v0 = rt * 128 + wv * 32;
    uint32_t lk[PP_GROUP];
    {
        const int lw = (a.n_dst + 31) >> 5;
        const int tcol = v0 < a.n_dst ? (v0 >> 5) : 0;
#pragma unroll
        for (int q = 0; q < PP_GROUP; ++q) lk[q] = a.links ? a.links[(int64_t)(u0 + q) * lw + tcol] : 0xffffffffu;
    }
    // stage XB[u0 .. u0 + PP_GROUP) : [node][base][d] -> LDS rows of 32 floats
    {
        // stage XB[u0 .. u0 + PP_GROUP): rows are stored padded to 32 columns, so the block IS the LDS image
        constexpr int N4 = PP_GROUP * NB * 32 / 4 / 256;                        // float4 per thread
        const float* src = a.xb + (int64_t)u0 * NB * 32;
        float4 x[N4];
#pragma unroll
        for (int j = 0; j < N4; ++j) x[j] = TIPK_DBG(a.dbg & 1024) ? make_float4(0.f, 0.f, 0.f, 0.f) : tipk_ld4(src + (j * 256 + t) * 4);
#pragma unroll
        for (int j = 0; j < N4; ++j) tipk_st4(xbl + (j * 256 + t) * 4, x[j]);
    }
    __syncthreads();
    // The group's XB blocks are in LDS: the workgroups of row tile 0 write them back TRANSPOSED, xbt[u][c][0 .. NB) -- the
    // layout in which the backward pass reads a column of all bases as one 128-byte line (tipk_rgcn_node_products xbt).
    // 8 lanes = the 8 float4 of one (node, column): full lines.  (As a second output of the XB product -- 4-byte stores,
    // one line each -- the forward pass lost the 5 us the backward pass gained; as a product of its own 4.7 us.)
    if (a.xbt && rt == 0) {
        constexpr int B4 = NB / 4;
        for (int e = t; e < PP_GROUP * a.d * B4; e += 256) {
            const int b4 = e % B4, c = (e / B4) % a.d, q = e / (B4 * a.d);
            if (u0 + q < a.n_dst) {
                const float* x = xbl + (q * NB + 4 * b4) * 32 + c;
                tipk_st4(a.xbt + ((int64_t)(u0 + q) * a.d + c) * NB + 4 * b4, make_float4(x[0], x[32], x[64], x[96]));
            }
        }
    }
    if (v0 >= a.n_dst) return;
    if constexpr (NARROW) {
        // lane = (row r of a 16-row block, quarter kq of the bases): A[r][k = kq] of instruction i is base kq * KQ + i of the lane's
        // row, B[k = kq][column r] the same base of XB[u] -- a fixed permutation of the reduction index, as in the wide body
        constexpr int KQ = NB / 4;
        const int r = lane & 15, kq = lane >> 4;
        const int64_t a_step = (int64_t)a.n_dst * NB;
        int vr[2];
        const float* ap[2];
        const float* at[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            vr[rb] = v0 + rb * 16 + r < a.n_dst ? v0 + rb * 16 + r : a.n_dst - 1;
            ap[rb] = a.cells + ((int64_t)u0 * a.n_dst + vr[rb]) * NB + KQ * kq;
            at[rb] = a.cells + ((int64_t)vr[rb] * a.n_dst + u0) * NB + KQ * kq;
        }
        const float* zp = a.zeros + KQ * kq;
        float a0[2][KQ], a1[2][KQ], a2[2][KQ];
        auto fetch = [&](float (&av)[2][KQ], int q) __attribute__((always_inline)) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const bool mirrored = a.symmetric && vr[rb] < u0 + q && u0 + q < a.n_dst;
                const float* p = mirrored ? at[rb] + q * NB : ap[rb] + q * a_step;
                p = ((lk[q] >> (rb * 16 + r)) & 1u) && !TIPK_DBG(a.dbg & 256) ? p : zp;
#pragma unroll
                for (int i = 0; i < KQ / 4; ++i) {
                    const float4 x = tipk_ld4(p + 4 * i);
                    av[rb][4 * i] = x.x; av[rb][4 * i + 1] = x.y; av[rb][4 * i + 2] = x.z; av[rb][4 * i + 3] = x.w;
                }
            }
        };
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        auto multiply = [&](const float (&av)[2][KQ], int q) __attribute__((always_inline)) {
            if (__builtin_amdgcn_readfirstlane((int)lk[q]) == 0 || TIPK_DBG(a.dbg & 512)) return;
            const float* b = xbl + (q * NB + KQ * kq) * 32 + r;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const float bv = b[i * 32];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][i], bv, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][i], bv, acc1, 0, 0, 0);
            }
        };
        fetch(a0, 0); fetch(a1, 1);
        fetch(a2, 2); __builtin_amdgcn_sched_barrier(0); multiply(a0, 0);
        fetch(a0, 3); __builtin_amdgcn_sched_barrier(0); multiply(a1, 1);
        fetch(a1, 4); __builtin_amdgcn_sched_barrier(0); multiply(a2, 2);
        fetch(a2, 5); __builtin_amdgcn_sched_barrier(0); multiply(a0, 3);
        fetch(a0, 6); __builtin_amdgcn_sched_barrier(0); multiply(a1, 4);
        fetch(a1, 7); __builtin_amdgcn_sched_barrier(0); multiply(a2, 5);
        multiply(a0, 6);
        multiply(a1, 7);
        // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + reg
        if (r < a.d && !TIPK_DBG(a.dbg & 2048)) {
            float* o = a.slabs + ((int64_t)g * a.n_dst + v0) * a.d + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = 4 * kq + i;
                if (v0 + rr < a.n_dst) o[(int64_t)rr * a.d] = acc0[i];
                if (v0 + 16 + rr < a.n_dst) o[(int64_t)(16 + rr) * a.d] = acc1[i];
            }
        }
        return;
    }
    const int v = v0 + row < a.n_dst ? v0 + row : a.n_dst - 1;                  // clamped: rows past the end are not stored
    const int64_t a_step = (int64_t)a.n_dst * NB;
    const float* ap = a.cells + ((int64_t)u0 * a.n_dst + v) * NB + KH * kh;     // cell (u0 + q, v): + q * a_step
    const float* at = a.cells + ((int64_t)v * a.n_dst + u0) * NB + KH * kh;     // mirrored cell (v, u0 + q): + q * NB
    float a0[KH], a1[KH], a2[KH];
    // 30 % of BioSNAP's drug pairs are linked: the cell of an UNLINKED pair is not fetched -- its lane reads one shared
    // block of zeros instead (no branch: the MFMAs run on zeros).  Round 4 skipped whole (32 rows, source node) tiles
    // without a link (19 % of them); per row the cell traffic falls from 43 MB to the 16 MB that hold data.
    const float* zp = a.zeros + KH * kh;
    auto fetch = [&](float (&av)[KH], int q) __attribute__((always_inline)) {       // (q is a literal at every call)
        const bool mirrored = a.symmetric && v < u0 + q && u0 + q < a.n_dst;
        const float* p = mirrored ? at + q * NB : ap + q * a_step;
        p = ((lk[q] >> row) & 1u) && !TIPK_DBG(a.dbg & 256) ? p : zp;
#pragma unroll
        for (int i = 0; i < KH / 4; ++i) {
            const float4 x = tipk_ld4(p + 4 * i);
            av[4 * i] = x.x; av[4 * i + 1] = x.y; av[4 * i + 2] = x.z; av[4 * i + 3] = x.w;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // a (32 rows, source node) tile without any linked pair (19 % of them at BioSNAP) is all zeros: its 16 MFMAs would add
    // exactly 0 -- skipped (the link word is the same for the whole wave: a scalar branch); the kernel is bound by the matrix
    // pipe on its dense work (profiles/r06_experiments.md section 8)
    auto multiply = [&](const float (&av)[KH], int q) __attribute__((always_inline)) {
        if (__builtin_amdgcn_readfirstlane((int)lk[q]) == 0 || TIPK_DBG(a.dbg & 512)) return;
        const float* b = xbl + (q * NB + KH * kh) * 32 + row;
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b[kk * 32], acc, 0, 0, 0);
    };
    static_assert(PP_GROUP == 8, "the node loop below is unrolled for 8 nodes, three register buffers");
    fetch(a0, 0); fetch(a1, 1);
    fetch(a2, 2); __builtin_amdgcn_sched_barrier(0); multiply(a0, 0);
    fetch(a0, 3); __builtin_amdgcn_sched_barrier(0); multiply(a1, 1);
    fetch(a1, 4); __builtin_amdgcn_sched_barrier(0); multiply(a2, 2);
    fetch(a2, 5); __builtin_amdgcn_sched_barrier(0); multiply(a0, 3);
    fetch(a0, 6); __builtin_amdgcn_sched_barrier(0); multiply(a1, 4);
    fetch(a1, 7); __builtin_amdgcn_sched_barrier(0); multiply(a2, 5);
    multiply(a0, 6);
    multiply(a1, 7);
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (row < a.d && !TIPK_DBG(a.dbg & 2048)) {
        float* o = a.slabs + ((int64_t)g * a.n_dst + v0) * a.d + row;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (v0 + rr < a.n_dst) o[(int64_t)rr * a.d] = acc[r];
        }
    }
}

}  // namespace

extern "C" int tipk_pair_product_supported(int n_bases, int d) {
    return (n_bases == 8 || n_bases == 16 || n_bases == 32) && d >= 1 && d <= 32;
}

extern "C" int tipk_pair_product(const float* cells, const float* xb, int64_t n_src, int64_t n_dst, int n_bases, int d,
                                 int group, int symmetric, const uint32_t* links, const float* zeros, float* xbt,
                                 float* slabs, tipk_stream_t stream) {
    if (!cells || !xb || !slabs || n_src <= 0 || n_dst <= 0 || group != PP_GROUP || n_src % group != 0) return TIPK_EINVAL;
    if (links && (!zeros || (reinterpret_cast<uintptr_t>(zeros) & 15))) return TIPK_EINVAL;
    if (xbt && ((reinterpret_cast<uintptr_t>(xbt) & 15) || n_src < n_dst)) return TIPK_EINVAL;   // (sources = destinations = the drugs)
    if (!tipk_pair_product_supported(n_bases, d)) return TIPK_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(cells) & 15) || (reinterpret_cast<uintptr_t>(xb) & 15) || n_src * n_dst * n_bases >= (1LL << 40)) return TIPK_EINVAL;
    PpArgs a;
    a.cells = cells; a.xb = xb; a.slabs = slabs;
    a.n_src = (int)n_src; a.n_dst = (int)n_dst; a.d = d; a.group = group; a.symmetric = symmetric != 0;
    a.links = links; a.zeros = links ? zeros : cells; a.xbt = xbt;
    a.dbg = tipk_option(TIPK_OPT_DP_DEBUG);
    a.row_tiles = (int)tipk_ceil_div(n_dst, 128);
    a.n_groups = (int)(n_src / group);
    const int64_t blocks = (int64_t)a.row_tiles * a.n_groups;
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = d <= 16 && n_bases >= 16;                  // (a quarter of the bases per lane = whole float4s)
    if (n_bases == 32 && narrow) hipLaunchKernelGGL((pair_product_kernel<32, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (n_bases == 32) hipLaunchKernelGGL((pair_product_kernel<32, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (n_bases == 16 && narrow) hipLaunchKernelGGL((pair_product_kernel<16, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (n_bases == 16) hipLaunchKernelGGL((pair_product_kernel<16, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pair_product_kernel<8, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    TIPK_RETURN_LAUNCH();
}
