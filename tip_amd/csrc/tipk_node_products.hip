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
constexpr int NP_THREADS = NP_WAVES * 64;

struct NpArgs {
    const float* dyc; int d, log2d;                     // [n_rows + 1][d], row n_rows = 0
    const int4* node_desc; const int32_t* row_rel; const int32_t* pos;
    int n_nodes, n_rows, R, R_pad, NB;
    const float* att; int64_t ld_att;
    const float* xb; int64_t xb_sb, xb_su;
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
template <int D>
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


// =====================================================================================================================
// Round 4: d att through LDS (row widths 16 and 32 -- the R-GCN layers of TIP).
//
// What bounded the kernel above (profiles/r03_pmc_node_products.txt): not the matrix cores (46 % busy) and not HBM, but the
// vector-memory ADDRESS path.  Role 2 fetches both MFMA operands "row per lane": every lane reads 64 contiguous bytes of
// ITS OWN 128-byte row, so a dwordx4 wave-instruction touches 64 different 16-byte pieces in 32-64 different lines and
// occupies the texture addresser for ~64 cycles instead of 16 -- 8 such instructions per 16 MFMAs, four SIMDs sharing one
// addresser: 2 048 addresser cycles per 1 024 MFMA cycles (TA_BUSY 59 % of the launch, 17.8 cache accesses per load).
// Here both operands arrive as FULL LINES by LDS-DMA (global_load_lds_dwordx4: 1 KiB contiguous per wave-instruction, no
// VGPR destination) and the row-per-lane fragments are read from LDS:
//
//   workgroup = NP2_W waves = NP2_W consecutive relation tiles (one per wave: its 32 x n_bases tile of d att stays in the
//               wave's accumulators for the whole range -- no cross-wave reduction) x a range of 32-column chunks of the
//               flattened (node, channel) axis;
//   stage     = one chunk: B = XB of the chunk [32 bases x 128 B], shared by the waves (it was fetched once per wave), A =
//               the rows of dY that EXIST for (the chunk's node(s), the workgroup's 32 NP2_W relations): consecutive rows of
//               the compact matrix, i.e. one contiguous block -- only real rows cross the fabric, each once;
//   ring      = NP2_NS stage slots; stage i + 3 is requested while stage i is multiplied and stage i + 1 read from LDS into
//               the second register set: one raw s_barrier per stage, counted vmcnt (LDS-DMA stays in flight across it);
//   lane m    finds its relation's row through the tile's bit mask (rank = popcount of the lower bits; a relation without a
//               row at this node reads a zero row kept in LDS): no `pos` table, the descriptors are scalar loads;
//   swizzle   128-byte (64-byte) rows would put every lane of a ds_read_b128 group on two (four) bank quads; the 16-byte
//               pieces of row r are stored at piece ^ ((r >> 1) & 7) (piece ^ ((r >> 2) & 3)) -- chosen on the SOURCE address
//               of the DMA, whose LDS side is lane-linear -- which is conflict-free for rows that differ mod 16.
// Role 1 (d XB per node) is the code of the kernel above; its cross-wave reduction buffer aliases the ring.
constexpr int NP2_W = 4;                                 // waves = relation tiles per role-2 workgroup
constexpr int NP2_NS = 3;                                // stage slots of the ring
constexpr int NP2_STAGE = (1 + NP2_W) * 4096;            // bytes: B image 4 KiB + A region NP2_W x 4 KiB
constexpr int NP2_P = 1 + NP2_W;                         // LDS-DMA instructions per wave and stage (20 pieces / 4 waves)
constexpr int NP2_ZERO = NP2_NS * NP2_STAGE;             // the zero row (128 bytes)
constexpr int NP2_REC = NP2_ZERO + 256;                  // the workgroup's descriptor records: NP2_W KiB = 32 NP2_W records of 32 bytes
constexpr int NP2_MAXREC = 32 * NP2_W;
constexpr int NP2_LDS = NP2_REC + 1024 * NP2_W;

struct Np2Args {
    NpArgs a;                                            // the fields of the kernel above (pos unused)
    const int4* recs;                                    // [n_rtg][n_nodes_pad][2]: {first row, rows, offsets of the 4 tiles (8 bits each), 0}, {masks}
    int n_rtg, n_nodes_pad, G2, chunks_per_wg2, n_role2;
};

typedef int np_i4 __attribute__((ext_vector_type(4)));
typedef float np_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ np_i4 np_sload4(const void* p) {                // scalar loads the compiler does not see: wait
    np_i4 v;                                                                // lgkmcnt(0) with the result as an operand before use
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(v) : "s"(p));
    return v;
}
__device__ __forceinline__ int np_sload1(const void* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(p));
    return v;
}
typedef __attribute__((address_space(3))) void np_lds_void_t;
typedef const __attribute__((address_space(1))) void np_global_void_t;

__device__ __forceinline__ void np2_dma(const char* src, char* lds_wave_base, bool on) {
    // (lane 0 is always on: an LDS-DMA whose lanes are ALL off is skipped and would not count in vmcnt)
    if (on) __builtin_amdgcn_global_load_lds((np_global_void_t*)src, (np_lds_void_t*)lds_wave_base, 16, 0, 0);
}
// debug builds ("dp_debug"): 32 = role 2 without MFMAs, 64 = without DMA, 128 = without fragment reads

template <int D>
__global__ __launch_bounds__(NP_THREADS) void node_products_lds_kernel(Np2Args p) {
    static_assert(D == 16 || D == 32, "row widths of the LDS form");
    const NpArgs& a = p.a;
    constexpr int NCT = 1;
    constexpr int NPC = 32 / D;                          // nodes per chunk
    constexpr int RB = D * 4;                            // bytes of a row of dY
    constexpr int PR = RB / 16;                          // 16-byte pieces per row
    constexpr int RPP = 64 / PR;                         // rows per 1-KiB DMA piece
    constexpr int SWS = D == 32 ? 1 : 2;                 // swizzle key of row r = (r >> SWS) & (PR - 1)
    __shared__ __attribute__((aligned(1024))) char smem[NP2_LDS];
    float* red = reinterpret_cast<float*>(smem);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n = lane & 31, kh = lane >> 5;
    constexpr int d = D;
    const int NB = a.NB;
    const int nb_c = n < NB ? n : NB - 1;
    const u32 d4 = (u32)d * 4u;

    if (TIPK_DBG(((int)blockIdx.x >= p.n_role2 ? a.dbg & 8 : a.dbg & 16))) return;
    if ((int)blockIdx.x >= p.n_role2) {
        // ---------------------------------------------------------------- role 1: dXB[:, u, :]  (as in the kernel above)
        const int4 nd = a.node_desc[(int)blockIdx.x - p.n_role2];
        const int u = __builtin_amdgcn_readfirstlane(nd.x);
        const int i_lo = __builtin_amdgcn_readfirstlane(nd.y);
        const int i_hi = __builtin_amdgcn_readfirstlane(nd.z);
        int n_tiles = (i_hi - i_lo + 31) >> 5;
        if (TIPK_DBG(a.dbg & 256)) n_tiles = n_tiles < 8 ? n_tiles : 8;       // debug: at most two tiles per wave
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
        const u32 ld_att4 = (u32)a.ld_att * 4u, nb_c4 = (u32)nb_c * 4u;
        u32 cc4[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { const int c = ct * 32 + n; cc4[ct] = (u32)(c < d ? c : d - 1) * 4u; }
        // THREE register sets per wave (statically named, used round robin): a wave's tiles are a chain of load round trips
        // (~1.3 us each under load) in front of 0.5 us of MFMAs, and the launch lasts as long as the hub node's chain -- 9
        // tiles per wave: 16 us in the kernel above, of which 4 us are arithmetic
        auto getrel = [&](int tile) {
            tile = tile < n_tiles ? tile : n_tiles - 1;
            int ir = i_lo + tile * 32 + n;
            ir = ir < i_hi ? ir : i_hi - 1;
            return a.row_rel[ir];
        };
        auto load = [&](int tile, int relv, float (&av)[16], float (&bv)[16]) {
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
                av[kk] = np_ldg(a.att, (u32)r * ld_att4 + nb_c4);
                const u32 rowb = (u32)(row < lim ? row : a.n_rows) * d4;
                bv[kk] = np_ldg(a.dyc, rowb + cc4[0]);
            }
        };
        auto mfma = [&](const float (&av)[16], const float (&bv)[16]) {
            if (TIPK_DBG(a.dbg & 512)) { for (int kk = 0; kk < 16; ++kk) acc[0][kk] += av[kk] + bv[kk]; return; }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc[0], 0, 0, 0);
        };
        if (n_tiles > 0) {
            // vmcnt counts at most 63 operations and a tile is 32 loads: two tiles in flight behind the one being multiplied.
            // The relation ids of a tile are requested TWO steps before the loads that need them and in front of that step's
            // loads (vector-memory operations complete in order: an id requested behind a tile's loads would make its use wait
            // for that tile -- which is what kept the kernel above at one tile in flight).
            constexpr int NF = 3;                                        // register sets = relation-id slots
            float av[NF][16], bv[NF][16];
            int rl[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) rl[f] = getrel(w + f * NP_WAVES);
#pragma unroll
            for (int f = 0; f < NF - 1; ++f) load(w + f * NP_WAVES, rl[f], av[f], bv[f]);
            rl[0] = getrel(w + NF * NP_WAVES);
            for (int tile = w; tile < n_tiles; tile += NF * NP_WAVES) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int cur = tile + f * NP_WAVES;                 // multiplied now (set f)
                    if (cur < n_tiles) {                                 // (uniform over the wave)
                        rl[(f + 1) % NF] = getrel(cur + (NF + 1) * NP_WAVES);
                        load(cur + (NF - 1) * NP_WAVES, rl[(f + NF - 1) % NF], av[(f + NF - 1) % NF], bv[(f + NF - 1) % NF]);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma(av[f], bv[f]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
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
    // -------------------------------------------------------------------- role 2: NP2_W tiles of d att, one per wave
    const int wg = (int)blockIdx.x;
    const int rtg = __builtin_amdgcn_readfirstlane(wg / p.G2), g = __builtin_amdgcn_readfirstlane(wg % p.G2);
    const int q_lo = g * p.chunks_per_wg2;
    const int q_hi = q_lo + p.chunks_per_wg2 < a.n_chunks ? q_lo + p.chunks_per_wg2 : a.n_chunks;
    const int n_st = q_hi - q_lo;                                           // stages of this workgroup (uniform over its waves)
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    {   // the zero row (visible behind the first barrier; by asm: an LDS store hipcc can see makes it wait for the LDS-DMA)
        const u32 za = (u32)(uintptr_t)(np_lds_void_t*)smem + (u32)NP2_ZERO + (u32)(t & 31) * 4u;
        const float zf = 0.f;
        asm volatile("ds_write_b32 %0, %1" ::"v"(za), "v"(zf) : "memory");
    }
    if (n_st > 0) {
        const int4* recs = p.recs + (int64_t)rtg * p.n_nodes_pad * 2;
        const int nflt = a.n_nodes * d;                                     // floats of a row of XB that exist
        // --- per-lane constants of the DMA: row / piece of this lane inside a 1-KiB piece, swizzled source offset
        const int l_row = lane / PR, l_pc = lane % PR;                      // A pieces: RPP rows x PR pieces
        const int lb_row = lane >> 3, lb_pc = lane & 7;                     // B pieces: 8 rows (bases) x 8 pieces
        const int b_row = w * 8 + lb_row;                                   // the base whose row this lane copies
        const u32 b_src0 = (u32)(b_row < NB ? b_row : NB - 1) * (u32)a.xb_sb * 4u;
        const int b_piece = lb_pc ^ ((b_row >> 1) & 7);                     // source piece that lands at LDS piece lb_pc
        // --- per-lane constants of the fragment reads
        const u32 b_rd = (u32)n * 128u;
        const int b_key = (n >> 1) & 7;
        u32 b_off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b_off[j] = b_rd + (u32)((((kh << 2) | j) ^ b_key) << 4);
        const u32 lt_mask = (1u << n) - 1u;

        // Descriptors.  A record is read ONCE by ONE workgroup, so every fetch from memory is a cold miss (~1 us, more than a
        // stage lasts): the records of all the workgroup's stages -- they are contiguous -- come in by one LDS-DMA piece per
        // wave in the prologue and are read from LDS (uniform address) one stage ahead of their use.  Only the first three
        // stages' records are fetched directly, as SCALAR loads issued by hand (behind the first LDS-DMA hipcc treats every
        // global load as possibly clobbered, makes it a vector load and waits vmcnt(0) for it).
        struct Rec { np_i4 r[NPC]; int m[NPC]; };                           // {first row, rows, tile offsets, 0}, this wave's tile mask
        const int n_rec = n_st * NPC;
        auto fetch = [&](int st) -> Rec {                                   // (valid behind `settle`)
            Rec rc;
            const int q = q_lo + (st < n_st ? st : n_st - 1);
#pragma unroll
            for (int h = 0; h < NPC; ++h) {
                const int4* rp = recs + (q * NPC + h) * 2;                   // (node n_nodes of an odd count: the zero record)
                rc.r[h] = np_sload4(rp);
                rc.m[h] = np_sload1(reinterpret_cast<const int*>(rp + 1) + w);
            }
            return rc;
        };
        auto settle = [&](Rec& rc) {
#pragma unroll
            for (int h = 0; h < NPC; ++h) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rc.r[h]), "+s"(rc.m[h]));
        };
        struct RecV { np_i4 r[NPC]; int m[NPC]; };                          // the same, read from LDS into (uniform) vector registers
        const u32 rec0 = (u32)(uintptr_t)(np_lds_void_t*)smem + (u32)NP2_REC;
        auto fetch_lds = [&](int st) -> RecV {
            RecV rv;
            const int si = st < n_st ? st : n_st - 1;
#pragma unroll
            for (int h = 0; h < NPC; ++h) {
                const u32 ra = rec0 + (u32)(si * NPC + h) * 32u;
                const u32 ma = ra + 16u + 4u * (u32)w;
                asm volatile("ds_read_b128 %0, %1" : "=v"(rv.r[h]) : "v"(ra));
                asm volatile("ds_read_b32 %0, %1" : "=v"(rv.m[h]) : "v"(ma));
            }
            return rv;
        };
        auto settle_lds = [&](RecV& rv) -> Rec {
            Rec rc;
#pragma unroll
            for (int h = 0; h < NPC; ++h) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv.r[h]), "+v"(rv.m[h]));
                rc.r[h].x = __builtin_amdgcn_readfirstlane(rv.r[h].x);
                rc.r[h].y = __builtin_amdgcn_readfirstlane(rv.r[h].y);
                rc.r[h].z = __builtin_amdgcn_readfirstlane(rv.r[h].z);
                rc.r[h].w = 0;
                rc.m[h] = __builtin_amdgcn_readfirstlane(rv.m[h]);
            }
            return rc;
        };
        {   // the records' DMA: piece w = records [32 w, 32 w + 32), 16 bytes per lane (the oldest vector-memory operation of the wave)
            const int ri = 32 * w + (lane >> 1);
            const bool ok = ri < n_rec;
            const char* src = reinterpret_cast<const char*>(recs + (int64_t)(q_lo * NPC + (ok ? ri : 0)) * 2 + (lane & 1));
            np2_dma(src, smem + NP2_REC + w * 1024, ok || lane == 0);
        }
        struct Rd { int o[NPC]; u32 mask[NPC]; bool real; };                // what the fragment reads of a stage need (uniform)
        // Requesting stage st into its slot = P = 5 LDS-DMA instructions per wave, ALWAYS issued (exec-masked when the stage or
        // the piece does not exist; lane 0 stays on) so that vmcnt counts are static.  Part 0 = the wave's piece of B, parts
        // 1 .. 4 = its pieces of A: the parts are issued one by one in the gaps of the running MFMA chain (`half`).
        auto rd_of = [&](int st, const Rec& rc) -> Rd {
            Rd rd;
            rd.real = st < n_st;
#pragma unroll
            for (int h = 0; h < NPC; ++h) {
                rd.o[h] = (rc.r[h].z >> (8 * w)) & 0xff;
                rd.mask[h] = rd.real ? (u32)rc.m[h] : 0u;
            }
            if (TIPK_DBG(a.dbg & 64)) { rd.mask[0] = 0; rd.mask[NPC - 1] = 0; }
            return rd;
        };
        auto request_part = [&](int part, int st, int slot_i, const Rec& rc) {
            if (TIPK_DBG(a.dbg & 64)) return;
            const bool real = st < n_st;
            char* slot = smem + slot_i * NP2_STAGE;
            if (part == 0) {
                // B: 8 rows of the XB image per wave.  floats [32 q, 32 q + 32) of every base's row; a piece past the end of
                // the row (last chunk of an odd node count) re-reads the last piece that exists: finite values under zero rows of A
                const int q = q_lo + (real ? st : n_st - 1);
                int f = 32 * q + 4 * b_piece;
                f = f + 4 <= nflt ? f : nflt - 4;
                np2_dma(reinterpret_cast<const char*>(a.xb) + b_src0 + (u32)f * 4u, slot + w * 1024, real || lane == 0);
                return;
            }
            // A: the block's rows, NP2_W * 4 / NPC pieces per node, dealt round robin to the waves
            constexpr int PCS = NP2_W * 4 / NPC;                            // 1-KiB pieces of a node's region
            constexpr int PJ = PCS / NP2_W;                                 // pieces of a node per wave
            const int h = (part - 1) / PJ, j = (part - 1) % PJ;
            const int p_lo = rc.r[h].x, nblk = rc.r[h].y;
            const int pc = w + NP2_W * j;
            const int r = pc * RPP + l_row;
            const int sp = l_pc ^ ((r >> SWS) & (PR - 1));
            const bool ok = real && r < nblk;
            const u32 off = ok ? (u32)(p_lo + r) * (u32)RB + (u32)(sp << 4) : 0u;
            np2_dma(reinterpret_cast<const char*>(a.dyc) + off, slot + 4096 + h * (NP2_W * 4096 / NPC) + pc * 1024, ok || lane == 0);
        };
        // Fragment reads of a stage (landed, behind a barrier) into one register set; a stage that does not exist reads zeros.
        // The reads are INLINE ASM: hipcc knows that LDS-DMA writes LDS and waits vmcnt(0) in front of every ds_read it can
        // see from the same array -- the counted waits are the real dependency.  `landed` makes the registers valid.
        const u32 lds0 = (u32)(uintptr_t)(np_lds_void_t*)smem;
        struct RdAddr { u32 rowb; int key; bool ex; u32 slot; bool real; };
        auto read_prep = [&](int slot_i, const Rd& rd) -> RdAddr {
            RdAddr ra;
            ra.slot = lds0 + (u32)(slot_i * NP2_STAGE);
            const int h = NPC == 2 ? kh : 0;
            const u32 mask = NPC == 2 ? (kh ? rd.mask[NPC - 1] : rd.mask[0]) : rd.mask[0];
            const int o = NPC == 2 ? (kh ? rd.o[NPC - 1] : rd.o[0]) : rd.o[0];
            ra.ex = (mask >> n) & 1u;
            const int r = o + __popc(mask & lt_mask);
            ra.key = (r >> SWS) & (PR - 1);
            ra.rowb = ra.slot + 4096u + (u32)h * (u32)(NP2_W * 4096 / NPC) + (u32)r * (u32)RB;
            ra.real = rd.real;
            return ra;
        };
        auto read_j = [&](int j, const RdAddr& ra, np_f4 (&a4)[4], np_f4 (&b4)[4]) {
            const int pcs = NPC == 2 ? j : ((kh << 2) | j);                 // d = 16: the lane's node is its half, the row is all its
            const u32 ao = ra.ex ? ra.rowb + (u32)((pcs ^ ra.key) << 4) : lds0 + (u32)NP2_ZERO + 16u * j;
            const u32 bo = ra.real ? ra.slot + b_off[j] : lds0 + (u32)NP2_ZERO + 16u * j;
            if (TIPK_DBG(a.dbg & 128)) { a4[j] = np_f4{0.f, 0.f, 0.f, 0.f}; b4[j] = a4[j]; return; }
            asm volatile("ds_read_b128 %0, %1" : "=v"(a4[j]) : "v"(ao));
            asm volatile("ds_read_b128 %0, %1" : "=v"(b4[j]) : "v"(bo));
        };
        auto landed = [&](np_f4 (&a4)[4], np_f4 (&b4)[4]) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]),
                                                  "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
        };
        auto mfma1 = [&](int k, const np_f4 (&a4)[4], const np_f4 (&b4)[4]) {      // MFMA k of a stage: k-slot 4 j + e
            if (TIPK_DBG(a.dbg & 32)) { acc[k] += a4[k >> 2][k & 3] + b4[k >> 2][k & 3]; return; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[k >> 2][k & 3], b4[k >> 2][k & 3], acc, 0, 0, 0);
        };
        // One stage of the loop: stage `st_mul` (registers cur) is multiplied while stage st_mul + 3 is requested and stage
        // st_mul + 1 (slot slot_rd, descriptor rd_rd) is read into the other register set (nxt).  A wave is in order and the
        // MFMAs of a stage are one dependent chain (64 cycles apart), so everything else of the stage is issued INSIDE the
        // chain, a few instructions per gap -- with two waves per SIMD running the same program in step, nothing else hides it.
        auto half = [&](int st_mul, int slot_free, int slot_rd, RecV& v_in, RecV& v_out, const Rd& rd_rd, Rd& rd_new,
                        const np_f4 (&ca)[4], const np_f4 (&cb)[4], np_f4 (&na4)[4], np_f4 (&nb4)[4]) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP2_P) : "memory");    // my pieces of stage st_mul + 1
            const Rec rc = settle_lds(v_in);                                            // the record of stage st_mul + 3
            __builtin_amdgcn_s_barrier();                                               // all pieces; slot_free has been read by all
            rd_new = rd_of(st_mul + 3, rc);
            const RdAddr ra = read_prep(slot_rd, rd_rd);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                mfma1(k, ca, cb);
                __builtin_amdgcn_sched_barrier(0);
                if (k < 5) request_part(k, st_mul + 3, slot_free, rc);
                else if (k == 5) v_out = fetch_lds(st_mul + 4);
                else if (k < 10) read_j(k - 6, ra, na4, nb4);
                __builtin_amdgcn_sched_barrier(0);
            }
            landed(na4, nb4);                                                           // (issued 6+ MFMAs ago)
        };
        // prologue: stages 0, 1, 2 requested; stage 0 into register set X.  Slot of stage s = s % 3, kept as a counter.
        Rec c0 = fetch(0), c1 = fetch(1), c2 = fetch(2);
        settle(c0); settle(c1); settle(c2);
#pragma unroll
        for (int part = 0; part < NP2_P; ++part) request_part(part, 0, 0, c0);
#pragma unroll
        for (int part = 0; part < NP2_P; ++part) request_part(part, 1, 1, c1);
#pragma unroll
        for (int part = 0; part < NP2_P; ++part) request_part(part, 2, 2, c2);
        const Rd r0 = rd_of(0, c0);
        Rd nx1 = rd_of(1, c1), nx2 = rd_of(2, c2);
        np_f4 aX[4], bX[4], aY[4], bY[4];
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NP2_P) : "memory");     // the records and stage 0
        __builtin_amdgcn_s_barrier();
        RecV vA = fetch_lds(3), vB;
        {
            const RdAddr ra = read_prep(0, r0);
#pragma unroll
            for (int j = 0; j < 4; ++j) read_j(j, ra, aX, bX);
        }
        landed(aX, bX);
        // two stages per trip (an odd count is padded with a stage of zeros)
        int s0 = 0;                                                         // slot of stage i
        for (int i = 0; i < n_st; i += 2) {
            const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
            Rd na, nb;
            half(i, s0, s1, vA, vB, nx1, na, aX, bX, aY, bY);
            half(i + 1, s1, s2, vB, vA, nx2, nb, aY, bY, aX, bX);
            nx1 = na; nx2 = nb;
            s0 = s2;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         // (the tail's dummy requests and fetches)
    }
    const int rt = rtg * NP2_W + w;
    float* o = a.datt + (int64_t)g * a.R * NB;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rel = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (rel < a.R && n < NB) o[(int64_t)rel * NB + n] = acc[r];
    }
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

// the LDS form (d = 16, 32): groups of NP2_W relation tiles x ranges of chunks -- about TWO role-2 workgroups per CU (what the
// LDS ring admits: while one wave of a SIMD issues its DMA pieces and fragment reads, the other multiplies) and at least 4
// stages each
static bool np2_shape(int d) { return d == 16 || d == 32; }
static int np2_ranges(int64_t n_rel, int64_t n_chunks) {
    const int64_t n_rtg = tipk_ceil_div(tipk_ceil_div(n_rel, 32), NP2_W);
    int64_t g = (512 + n_rtg / 2) / n_rtg;
    if (g > 64) g = 64;
    if (g > n_chunks / 4) g = n_chunks / 4;
    if (g < tipk_ceil_div(n_chunks, NP2_MAXREC / 2)) g = tipk_ceil_div(n_chunks, NP2_MAXREC / 2);   // a workgroup's records fit their LDS block
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
    const int64_t n_chunks = tipk_ceil_div(n_nodes * d, 32);
    *att_slabs = np2_shape(d) ? np2_ranges(n_rel, n_chunks) : np_ranges(n_rel, n_chunks);
    return TIPK_OK;
}

extern "C" int tipk_rgcn_node_products(const float* dyc, int64_t n_rows, int d, const int32_t* node_desc,
                                       const int32_t* row_rel, const int32_t* pos, const int32_t* tile_recs,
                                       int64_t n_nodes, int64_t n_rel, const float* att, int64_t ld_att, int n_bases,
                                       const float* xb, int64_t xb_sb, int64_t xb_su,
                                       float* dxb, int64_t dxb_sb, int64_t dxb_su, float* datt_slabs,
                                       tipk_stream_t stream) {
    int G = 0;
    const int rc = tipk_rgcn_node_products_plan(n_nodes, d, n_rel, n_bases, &G);
    if (rc != TIPK_OK) return rc;
    if (G == 0) return TIPK_EUNSUPPORTED;
    const bool lds_form = np2_shape(d);
    if (!dyc || !node_desc || !row_rel || !att || !xb || !dxb || !datt_slabs || n_rows <= 0 || ld_att < n_bases)
        return TIPK_EINVAL;
    if (lds_form ? !tile_recs : !pos) return TIPK_EINVAL;
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
    a.xb = xb; a.xb_sb = xb_sb; a.xb_su = xb_su;
    a.dxb = dxb; a.dxb_sb = dxb_sb; a.dxb_su = dxb_su;
    a.datt = datt_slabs;
    a.G = G; a.n_rp = a.R_pad / 32;
    a.n_chunks = (int)tipk_ceil_div(n_nodes * d, 32);
    a.chunks_per_wg = (int)tipk_ceil_div(a.n_chunks, G);
    a.n_role2 = a.n_rp * G;
    a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DP_DEBUG));
    hipStream_t st = (hipStream_t)stream;
    if (lds_form) {
        // the LDS form reads whole rows of XB [bases][node][d]: the rows of a node's chunk must be contiguous and 16-byte aligned
        if (xb_su != d || (reinterpret_cast<uintptr_t>(tile_recs) & 15)) return TIPK_EINVAL;
        Np2Args p;
        p.a = a;
        p.recs = reinterpret_cast<const int4*>(tile_recs);
        p.n_rtg = (int)tipk_ceil_div(tipk_ceil_div(n_rel, 32), NP2_W);
        p.n_nodes_pad = (int)(tipk_ceil_div(n_nodes, 2) * 2);
        p.G2 = G;
        p.chunks_per_wg2 = a.chunks_per_wg;
        p.n_role2 = p.n_rtg * G;
        const unsigned grid = (unsigned)(a.n_nodes + p.n_role2);
        if (d == 16) hipLaunchKernelGGL(node_products_lds_kernel<16>, dim3(grid), dim3(NP_THREADS), 0, st, p);
        else hipLaunchKernelGGL(node_products_lds_kernel<32>, dim3(grid), dim3(NP_THREADS), 0, st, p);
        TIPK_RETURN_LAUNCH();
    }
    const unsigned grid = (unsigned)(a.n_nodes + a.n_role2);
    if (d == 64) hipLaunchKernelGGL(node_products_kernel<64>, dim3(grid), dim3(NP_THREADS), 0, st, a);
    else hipLaunchKernelGGL(node_products_kernel<128>, dim3(grid), dim3(NP_THREADS), 0, st, a);
    TIPK_RETURN_LAUNCH();
}
