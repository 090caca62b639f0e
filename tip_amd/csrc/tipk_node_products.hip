// Both consumers of dY on its COMPACT, node-major form (include/tipk.h section 2d):
//
//     dXB [b, u, c]  = sum_{r in rel(u)} att[r, b] * dY[(u, r), c]          (K = the relations that leave node u)
//     datt[r, b]     = sum_u sum_c dY[(u, r), c] * XB[b, u, c]
//
// The transposed D-D gather (tipk_stream_gather) produces one row of dY per (relation, source node) pair that has
// an edge -- 47 % of the R x N pairs of BioSNAP.  tipk_rgcn_dy_products works on the dense [R x N d] matrix: it
// multiplies the empty rows too (masked), adds the 16 waves' d att tiles through LDS behind two barriers per 32-row
// tile, and leaves d XB as one slab per range of relations.  Here the gather writes only the rows that exist, grouped
// by SOURCE NODE (ascending relation inside a node), and the two products are two kinds of workgroups of ONE launch
// whose wavefronts never meet before their final sum:
//
//   role 1 (one workgroup per node u, heaviest nodes first): the node's rows are cut into tiles of 32; a wave takes
//          every 8th tile, A = the att rows of the tile's relations (gathered through row_rel), B = the dY tile as it
//          lies in memory (lane = column), and keeps the [bases x d] block in its accumulators: K runs over REAL rows
//          only and the block is complete when the node is done -- no d XB slabs.
//   role 2 (one workgroup per (tile of 32 relations, range of column chunks)): a wave walks 32-column chunks of the
//          flattened (node, channel) axis; A = for each of its 32 relations the 16 floats of that relation's row at the chunk's node, fetched through the position table `pos` (absent pairs point at one shared zero row: a
//          broadcast line, never HBM traffic), B = XB of that node.  The [32 relations x bases] tile stays in the
//          accumulators over the whole range; one small slab per range.
//
// Both roles read dY straight from global memory in the register layout of v_mfma_f32_32x32x2_f32 (the k index of
// an operand pair may be permuted freely: k = 2 kk + kh <-> column kh * 16 + kk, so role 2 loads 16 consecutive
// floats per lane as four dwordx4), two tiles in flight per wave in statically named register sets.  No LDS
// transposes, no barriers inside the loops, no atomics: sums are in fixed order, bitwise reproducible.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

constexpr int NP_WAVES = 4;                             // waves per workgroup (both roles)
#ifndef NP_THREE_SETS
#define NP_THREE_SETS 1
#endif
constexpr int NP_THREADS = NP_WAVES * 64;

struct NpArgs {
    const float* dyc; int d, log2d;                     // [n_rows + 1][d], row n_rows = 0
    const int4* node_desc; const int32_t* row_rel; const int32_t* pos;
    int n_nodes, n_rows, R, R_pad, NB;
    const float* att; int64_t ld_att;
    const float* xb; int64_t xb_sb, xb_su;
    const float* xbt;                                   // nullable: the same XB as [node][column][base] (role 2 reads it coalesced)
    float* dxb; int64_t dxb_sb, dxb_su;
    float* datt;                                        // slabs [G][R][NB]
    int G, n_rp, n_chunks, chunks_per_wg, n_role2;
    int dbg;                                            // debug builds ("dp_debug"): 8 = role 1 returns at once, 16 = role 2
};

// the waves' accumulator tiles -> one tile, in wave order; element e = reg * 64 + lane  <->  row (reg & 3) +
// 8 (reg >> 2) + 4 (lane >> 5), column lane & 31 of the MFMA result
template <typename Store>
__device__ __forceinline__ void np_reduce_store(float* red, const f32x16& acc, int t, int w, int lane, Store&& store) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w * 1024 + r * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 1024 / NP_THREADS; ++j) {
        const int e = t + j * NP_THREADS;
        float s = red[e];
#pragma unroll
        for (int q = 1; q < NP_WAVES; ++q) s += red[q * 1024 + e];
        const int r = e >> 6, l = e & 63;
        store((r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31, s);
    }
    __syncthreads();
}

__device__ __forceinline__ float np_ldg(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float4 np_ldg4(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// No operand is ever masked: a row / chunk that does not exist reads the ZERO ROW of dyc on one side of the product (its
// partner on the other side is a clamped, finite value), and result rows / columns beyond n_bases or d are not stored.
// A `cond ? loaded : 0` next to its load makes hipcc skip the load under exec and wait for each one on the spot.
// D = the row width d as a template constant: the two R-GCN layers of a model launch with identical grids, and a name per
// width keeps them apart in kernel traces and counter summaries (profiles/*_kernel_by_grid.csv); NCT = 32-column tiles per row
template <int D, bool XBT = false>
__global__ __launch_bounds__(NP_THREADS) void node_products_kernel(NpArgs a) {
    constexpr int NCT = (D + 31) / 32;
    __shared__ float red[NP_WAVES * 1024];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = lane & 31, kh = lane >> 5;
    constexpr int d = D;
    const int NB = a.NB;
    const int nb_c = n < NB ? n : NB - 1;
    const u32 d4 = (u32)d * 4u;

    // role 2 workgroups (uniform, long) are launched first; the per-node ones fill in around them
    if (TIPK_DBG(((int)blockIdx.x >= a.n_role2 ? a.dbg & 8 : a.dbg & 16))) return;
    if ((int)blockIdx.x >= a.n_role2) {
        // ---------------------------------------------------------------- role 1: dXB[:, u, :]
        const int4 nd = a.node_desc[(int)blockIdx.x - a.n_role2];         // uniform index: one scalar load
        const int u = __builtin_amdgcn_readfirstlane(nd.x);
        const int i_lo = __builtin_amdgcn_readfirstlane(nd.y);
        const int i_hi = __builtin_amdgcn_readfirstlane(nd.z);
        const int n_tiles = (i_hi - i_lo + 31) >> 5;
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
        const u32 ld_att4 = (u32)a.ld_att * 4u, nb_c4 = (u32)nb_c * 4u;
        u32 cc4[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { const int c = ct * 32 + n; cc4[ct] = (u32)(c < d ? c : d - 1) * 4u; }

        // relation ids of a tile's rows (lane i of either half holds row i0 + i), requested one stage ahead of the loads
        auto getrel = [&](int tile) {
            tile = tile < n_tiles ? tile : n_tiles - 1;
            int ir = i_lo + tile * 32 + n;
            ir = ir < i_hi ? ir : i_hi - 1;
            return a.row_rel[ir];
        };
        auto load = [&](int tile, int relv, float (&av)[16], float (&bv)[NCT][16]) {
            const bool tile_ok = tile < n_tiles;
            tile = tile_ok ? tile : n_tiles - 1;
            const int i0 = i_lo + tile * 32;
            const int lim = tile_ok ? i_hi : 0;                          // a tile past the end: every row is the zero row
            int rr[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) rr[kk] = __shfl(relv, 2 * kk + kh, 64);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int i = 2 * kk + kh;
                const int row = i0 + i;
                const int r = rr[kk];
                if (!TIPK_DBG(a.dbg & 128)) av[kk] = np_ldg(a.att, (u32)r * ld_att4 + nb_c4);
                const u32 rowb = (u32)(row < lim ? row : a.n_rows) * d4;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) if (!TIPK_DBG(a.dbg & 256)) bv[ct][kk] = np_ldg(a.dyc, rowb + cc4[ct]);
            }
        };
        auto mfma = [&](const float (&av)[16], const float (&bv)[NCT][16]) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[ct][kk], acc[ct], 0, 0, 0);
        };
        if (n_tiles > 0) {
            if constexpr (NCT == 1 && NP_THREE_SETS) {
                // THREE register sets (round 4).  vmcnt counts at most 63 operations and a tile is 32 loads: two tiles in
                // flight behind the one being multiplied.  The relation ids of a tile are requested TWO steps before the loads
                // that need them and IN FRONT of that step's loads: vector-memory operations complete in order, so an id
                // requested behind a tile's loads makes its use wait for that tile -- which kept the two-set form below at
                // one tile in flight (a hub node's 9 tiles per wave: 9 load round trips in a row).
                constexpr int NF = 3;
                float av[NF][16], bv[NF][1][16];
                int rl[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) rl[f] = getrel(w + f * NP_WAVES);
#pragma unroll
                for (int f = 0; f < NF - 1; ++f) load(w + f * NP_WAVES, rl[f], av[f], bv[f]);
                rl[0] = getrel(w + NF * NP_WAVES);
                for (int tile = w; tile < n_tiles; tile += NF * NP_WAVES) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const int cur = tile + f * NP_WAVES;             // multiplied now (set f)
                        if (cur < n_tiles) {                             // (uniform over the wave)
                            rl[(f + 1) % NF] = getrel(cur + (NF + 1) * NP_WAVES);
                            load(cur + (NF - 1) * NP_WAVES, rl[(f + NF - 1) % NF], av[(f + NF - 1) % NF], bv[(f + NF - 1) % NF]);
                            __builtin_amdgcn_sched_barrier(0);
                            mfma(av[f], bv[f]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else {
            float aX[16] = {}, bX[NCT][16] = {}, aY[16] = {}, bY[NCT][16] = {};
            int tile = w;
            int rX = getrel(tile), rY = getrel(tile + NP_WAVES);
            load(tile, rX, aX, bX);
            for (; tile < n_tiles; tile += 2 * NP_WAVES) {
                rX = getrel(tile + 2 * NP_WAVES);
                load(tile + NP_WAVES, rY, aY, bY);
                __builtin_amdgcn_sched_barrier(0);
                mfma(aX, bX);
                __builtin_amdgcn_sched_barrier(0);
                rY = getrel(tile + 3 * NP_WAVES);
                load(tile + 2 * NP_WAVES, rX, aX, bX);
                __builtin_amdgcn_sched_barrier(0);
                mfma(aY, bY);                                            // (a tile past the end multiplies zero rows)
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
        float* o = a.dxb + (int64_t)u * a.dxb_su;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
            np_reduce_store(red, acc[ct], t, w, lane, [&](int b, int c, float s) {
                const int col = ct * 32 + c;
                if (b < NB && col < d) o[(int64_t)b * a.dxb_sb + col] = s;
            });
        return;
    }
    // -------------------------------------------------------------------- role 2: one slab tile of d att
    const int wg = (int)blockIdx.x;
    const int rt = __builtin_amdgcn_readfirstlane(wg / a.G), g = __builtin_amdgcn_readfirstlane(wg % a.G);
    const int q_lo = g * a.chunks_per_wg;
    const int q_hi = q_lo + a.chunks_per_wg < a.n_chunks ? q_lo + a.chunks_per_wg : a.n_chunks;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (q_lo < q_hi) {
        const u32 pos_m4 = (u32)(rt * 32 + n) * 4u, rpad4 = (u32)a.R_pad * 4u;
        const u32 xb_n4 = (u32)nb_c * (u32)a.xb_sb * 4u, xb_su4 = (u32)a.xb_su * 4u;
        auto getpos = [&](int q) {
            q = q < q_hi ? q : q_hi - 1;
            int node = (32 * q + 16 * kh) / D;
            node = node < a.n_nodes ? node : a.n_nodes - 1;
            return __float_as_int(np_ldg(reinterpret_cast<const float*>(a.pos), (u32)node * rpad4 + pos_m4));
        };
        auto load = [&](int q, int p, float4 (&a4)[4], float4 (&b4)[4]) {
            const bool q_ok = q < q_hi;
            q = q_ok ? q : q_hi - 1;
            const int f = 32 * q + 16 * kh;
            const int node = f / D;
            const u32 c04 = (u32)(f & (d - 1)) * 4u;
            const bool ok = q_ok && node < a.n_nodes;                       // the last chunk may run past the last node
            const u32 ab = (u32)(ok ? p : a.n_rows) * d4 + c04;             // not there: the zero row
            const u32 bb = xb_n4 + (u32)(node < a.n_nodes ? node : a.n_nodes - 1) * xb_su4 + c04;
            if constexpr (XBT) {
                // XB as [node][column][base]: lane n reads base n of ONE column per load -- the 32 lanes of a half share a
                // 128-byte line (2 lines per instruction).  From the [base][node][column] operand every lane's 64 bytes sit
                // in a line of their own: 32 lines per 16-byte load, four loads per chunk, and the vector-memory address
                // path was what bounded this role (TA busy 59 %: profiles/r04_node_products_experiments.md)
                const u32 tb = ((u32)(node < a.n_nodes ? node : a.n_nodes - 1) * (u32)D + (u32)(f & (d - 1))) * (u32)NB * 4u + (u32)nb_c * 4u;
                const u32 nb4 = (u32)NB * 4u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!TIPK_DBG(a.dbg & 64)) a4[j] = np_ldg4(a.dyc, ab + 16u * j);
                    if (!TIPK_DBG(a.dbg & 32)) {
                        b4[j].x = np_ldg(a.xbt, tb + (4 * j) * nb4); b4[j].y = np_ldg(a.xbt, tb + (4 * j + 1) * nb4);
                        b4[j].z = np_ldg(a.xbt, tb + (4 * j + 2) * nb4); b4[j].w = np_ldg(a.xbt, tb + (4 * j + 3) * nb4);
                    }
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!TIPK_DBG(a.dbg & 64)) a4[j] = np_ldg4(a.dyc, ab + 16u * j);
                if (!TIPK_DBG(a.dbg & 32)) b4[j] = np_ldg4(a.xb, bb + 16u * j);
            }
        };
        auto mfma = [&](const float4 (&a4)[4], const float4 (&b4)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc, 0, 0, 0);
            }
        };
        float4 aX[4] = {}, bX[4] = {}, aY[4] = {}, bY[4] = {};
        int q = q_lo + w;
        int pX = getpos(q), pY = getpos(q + NP_WAVES);
        load(q, pX, aX, bX);
        for (; q < q_hi; q += 2 * NP_WAVES) {
            pX = getpos(q + 2 * NP_WAVES);
            load(q + NP_WAVES, pY, aY, bY);
            __builtin_amdgcn_sched_barrier(0);
            mfma(aX, bX);
            __builtin_amdgcn_sched_barrier(0);
            pY = getpos(q + 3 * NP_WAVES);
            load(q + 2 * NP_WAVES, pX, aX, bX);
            __builtin_amdgcn_sched_barrier(0);
            mfma(aY, bY);                                                   // (a chunk past the end multiplies the zero row)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* o = a.datt + (int64_t)g * a.R * NB;
    np_reduce_store(red, acc, t, w, lane, [&](int m, int b, float s) {
        const int rel = rt * 32 + m;
        if (rel < a.R && b < NB) o[(int64_t)rel * NB + b] = s;
    });
}

}  // namespace

// relation tiles x ranges of column chunks: about one workgroup (8 waves) per CU, so that every SIMD has the
// same number of role 2 waves -- at most 32 slabs
static int np_ranges(int64_t n_rel, int64_t n_chunks) {
    const int64_t n_rt = tipk_ceil_div(n_rel, 32);
    int64_t g = 512 / n_rt;
    if (g > 32) g = 32;
    if (g > n_chunks / (2 * NP_WAVES)) g = n_chunks / (2 * NP_WAVES);       // at least two chunks per wave
    if (g < 1) g = 1;
    const int64_t per = tipk_ceil_div(n_chunks, g);
    return (int)tipk_ceil_div(n_chunks, per);
}

extern "C" int tipk_rgcn_node_products_plan(int64_t n_nodes, int d, int64_t n_rel, int n_bases, int* att_slabs) {
    if (!att_slabs) return TIPK_EINVAL;
    *att_slabs = 0;
    if (n_nodes <= 0 || n_rel <= 0 || n_bases <= 0 || n_bases > 32) return TIPK_OK;
    if (d < 16 || d > 128 || (d & (d - 1)) != 0) return TIPK_OK;
    if (n_nodes * (int64_t)d >= (1LL << 29) || n_rel >= (1LL << 24)) return TIPK_OK;
    *att_slabs = np_ranges(n_rel, tipk_ceil_div(n_nodes * d, 32));
    return TIPK_OK;
}

extern "C" int tipk_rgcn_node_products(const float* dyc, int64_t n_rows, int d, const int32_t* node_desc,
                                       const int32_t* row_rel, const int32_t* pos,
                                       int64_t n_nodes, int64_t n_rel, const float* att, int64_t ld_att, int n_bases,
                                       const float* xb, int64_t xb_sb, int64_t xb_su, const float* xbt,
                                       float* dxb, int64_t dxb_sb, int64_t dxb_su, float* datt_slabs,
                                       tipk_stream_t stream) {
    int G = 0;
    const int rc = tipk_rgcn_node_products_plan(n_nodes, d, n_rel, n_bases, &G);
    if (rc != TIPK_OK) return rc;
    if (G == 0) return TIPK_EUNSUPPORTED;
    if (!dyc || !node_desc || !row_rel || !pos || !att || !xb || !dxb || !datt_slabs || n_rows <= 0 || ld_att < n_bases)
        return TIPK_EINVAL;
    // 32-bit byte offsets into dyc, att, xb and pos
    if ((n_rows + 1) * (int64_t)d >= (1LL << 29) || n_rel * ld_att >= (1LL << 29) || n_bases * xb_sb >= (1LL << 29) ||
        n_nodes * xb_su >= (1LL << 29) || n_nodes * (n_rel + 64) >= (1LL << 29))
        return TIPK_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dyc) & 15) || (reinterpret_cast<uintptr_t>(xb) & 15) || xb_sb % 4 != 0 || xb_su % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(node_desc) & 15))
        return TIPK_EINVAL;
    NpArgs a;
    a.dyc = dyc; a.d = d; a.log2d = __builtin_ctz((unsigned)d);
    a.node_desc = reinterpret_cast<const int4*>(node_desc); a.row_rel = row_rel; a.pos = pos;
    a.n_nodes = (int)n_nodes; a.n_rows = (int)n_rows; a.R = (int)n_rel; a.R_pad = (int)(tipk_ceil_div(n_rel, 64) * 64);
    a.NB = n_bases;
    a.att = att; a.ld_att = ld_att;
    a.xb = xb; a.xb_sb = xb_sb; a.xb_su = xb_su; a.xbt = xbt;
    a.dxb = dxb; a.dxb_sb = dxb_sb; a.dxb_su = dxb_su;
    a.datt = datt_slabs;
    a.G = G; a.n_rp = a.R_pad / 32;
    a.n_chunks = (int)tipk_ceil_div(n_nodes * d, 32);
    a.chunks_per_wg = (int)tipk_ceil_div(a.n_chunks, G);
    a.n_role2 = a.n_rp * G;
    a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DP_DEBUG));
    const unsigned grid = (unsigned)(a.n_nodes + a.n_role2);
    hipStream_t st = (hipStream_t)stream;
    if (xbt && d == 16) hipLaunchKernelGGL((node_products_kernel<16, true>), dim3(grid), dim3(NP_THREADS), 0, st, a);
    else if (xbt && d == 32) hipLaunchKernelGGL((node_products_kernel<32, true>), dim3(grid), dim3(NP_THREADS), 0, st, a);
    else if (xbt && d == 64) hipLaunchKernelGGL((node_products_kernel<64, true>), dim3(grid), dim3(NP_THREADS), 0, st, a);
    else if (xbt) hipLaunchKernelGGL((node_products_kernel<128, true>), dim3(grid), dim3(NP_THREADS), 0, st, a);
    else if (d == 16) hipLaunchKernelGGL(node_products_kernel<16>, dim3(grid), dim3(NP_THREADS), 0, st, a);
    else if (d == 32) hipLaunchKernelGGL(node_products_kernel<32>, dim3(grid), dim3(NP_THREADS), 0, st, a);
    else if (d == 64) hipLaunchKernelGGL(node_products_kernel<64>, dim3(grid), dim3(NP_THREADS), 0, st, a);
    else hipLaunchKernelGGL(node_products_kernel<128>, dim3(grid), dim3(NP_THREADS), 0, st, a);
    TIPK_RETURN_LAUNCH();
}
