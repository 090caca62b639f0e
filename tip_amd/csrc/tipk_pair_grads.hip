// The dense half of the pair-form D-D BACKWARD pass (include/tipk.h section 2e; autograd of src/layers.py:159-180).
//
// The forward pass (section 2c) is   agg[v] = sum_u C[u, v, :] . XB[u],   C[u, v, :] = sum of att[r, :] over the relations
// linking u -> v.  Its two gradients, with g' = g / deg:
//
//     dXB[b, u, c]  = sum_{v linked to u} C[u, v, b] * g'[v, c]                      (uses the cells the forward pass left)
//     dC [u, v, b]  = sum_c XB[u][b][c] * g'[v, c]      for the LINKED pairs only     ("pair gradient", one 128-byte row)
//
// and d att[r, :] = sum over the pairs r links of dC -- a wave-stream gather of those rows (tipk_stream_gather_parts).
// Round 3/4 went through dY = A_r^T g' (one row per (relation, source node), 329 188 rows = 42 MB at BioSNAP layer 1, a
// gather over all 8.3 M directed edges) and two products on it, of which the d att one multiplied 53 % zero rows
// (tipk_rgcn_node_products: 31 us + a 28 us gather).  Here nothing is per relation: a drug pair is one K-step.
//
// A node's linked pairs are cut into tiles of 32 SLOTS (a slot = one neighbour v_j; plan: tip_amd/plan.py
// `build_pair_bwd_plan`); the two products are two kinds of workgroups of ONE launch:
//   role 1, one workgroup per source node u (heaviest first, launched first), a wave takes every 4th tile: dXB[:, u, :].
//     g' rows of a tile's 32 neighbours -- lane (j, half) loads 64 (32) contiguous bytes of row v_j, scaled by 1 / deg(v_j)
//     (kept in the slot) -- are transposed through a wave-private LDS tile (no barrier: LDS operations of a wave complete in
//     order) into the B operand; A = the neighbours' cell lines, one dword per lane and K-step, two full lines per load
//     (lane = base); v_mfma_f32_32x32x2_f32 (d = 32) / v_mfma_f32_16x16x4_f32 (d = 16: no half-empty 32-column tile); K
//     runs over LINKED pairs only.  Two tiles in flight per wave in statically named register sets, slot words requested one
//     tile further ahead and in front of that step's loads (vector-memory operations retire in order).  The 4 waves' tiles
//     are added through LDS in wave order: bitwise reproducible.
//   role 2, one WAVE per tile: its 32 pair-gradient rows, dC = g' tile (A, as it is loaded) . XB[u]^T (B); the [32 slots x 32
//     bases] result leaves as 16 stores of two full 128-byte lines each.  Round 5, first version: the same waves did both
//     products, 32 MFMAs + 16 stores per tile -- a hub node's wave walked 5 such tiles in a row while most of the chip had
//     finished (debug decomposition, tools/bench_pair_grads.py: 21.8 us, 11.7 with at most one tile per wave); the tiles
//     of this half are independent, so they are dealt one per wave over the whole chip.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

constexpr int PG_WAVES = 4;
constexpr int PG_THREADS = PG_WAVES * 64;

struct PgArgs {
    const float* cells;                 // [..][n_bases = 32] lines, addressed by slot.line
    const float* xb;                    // [n_nodes][32 bases][32]  (rows padded to 32 columns)
    const float* g; int ld_g;           // [n_nodes][d]
    const int4* node_desc;              // [n_nodes] {u, first slot, tiles, 0}, heaviest first
    const int32_t* tile_node;           // [n_tiles] node of every tile of 32 slots
    int n_nodes, n_tiles;
    const int4* slots;                  // [n_slots] {v, bits of 1 / deg(v) (0 for a pad), cell line of (u, v), row of pg}
    float* dxb; int64_t dxb_sb, dxb_su;
    float* pg;                          // [..][32]: slot s writes row slots[s].w
    int dbg;                            // debug builds ("dp_debug"): 1 no dC stores, 2 no dC product, 4 no dXB product (cells unread),
};                                      // 8 one tile per wave at most, 16 no g' loads, 32 no cell loads, 64 no role 2, 128 no role 1

__device__ __forceinline__ float pg_ldg(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float4 pg_ldg4(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
// LDS accesses of ONE wave complete in issue order; what has to be stopped is the compiler moving a read of the tile in
// front of the other lanes' writes
__device__ __forceinline__ void pg_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ROLE 2 of the launch (workgroups behind the per-node ones): the pair-gradient rows, one tile of 32 slots per WAVE -- every
// tile is independent (its node's XB block is the only shared operand), so this half of the work is spread evenly over the
// chip instead of queueing behind a hub node's tiles in the per-node workgroups.
template <int D>
__device__ __forceinline__ void pg_role_dc(const PgArgs& a, int wg, int lane, int w) {
    constexpr int H = D / 2;
    const int n = lane & 31, kh = lane >> 5;
    const int tl = wg * PG_WAVES + w;
    if (tl >= a.n_tiles) return;                                            // (uniform per wave; this role has no barrier)
    const int u = a.tile_node[tl];                                          // uniform index: a scalar load
    const int4 s = a.slots[tl * 32 + n];
    float xbf[H], gs[H];
    const u32 off = ((u32)u * 1024u + (u32)n * 32u + (u32)(H * kh)) * 4u;
    const u32 goff = (u32)s.x * (u32)a.ld_g * 4u + (u32)(H * kh) * 4u;
#pragma unroll
    for (int i = 0; i < H / 4; ++i) {
        const float4 x = pg_ldg4(a.xb, off + 16u * i);
        xbf[4 * i] = x.x; xbf[4 * i + 1] = x.y; xbf[4 * i + 2] = x.z; xbf[4 * i + 3] = x.w;
        const float4 y = pg_ldg4(a.g, goff + 16u * i);
        gs[4 * i] = y.x; gs[4 * i + 1] = y.y; gs[4 * i + 2] = y.z; gs[4 * i + 3] = y.w;
    }
    const float ss = __int_as_float(s.y);
    // dC tile [32 slots x 32 bases] = g' tile (A: lane = slot) . XB[u]^T (B: lane = base), K = d
    f32x16 pc;
#pragma unroll
    for (int i = 0; i < 16; ++i) pc[i] = 0.f;
    if (!TIPK_DBG(a.dbg & 2)) {
#pragma unroll
        for (int i = 0; i < H; ++i) pc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[i] * ss, xbf[i], pc, 0, 0, 0);
    }
    // rows of dC: C/D layout of the 32x32 MFMA -- column = lane & 31 (base), row = (reg & 3) + 8 (reg >> 2) + 4 kh (slot);
    // every row goes where the d att gather stages it from (the slot's own word): two full 128-byte lines per store
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (TIPK_DBG(a.dbg & 1)) break;
        const int dj = __shfl(s.w, (r & 3) + 8 * (r >> 2) + 4 * kh, 64);
        *reinterpret_cast<float*>(reinterpret_cast<char*>(a.pg) + ((u32)dj * 128u + (u32)n * 4u)) = pc[r];
    }
}

template <int D>
__global__ __launch_bounds__(PG_THREADS) void pair_grads_kernel(PgArgs a) {
    static_assert(D == 16 || D == 32, "d = 16 | 32");
    constexpr int H = D / 2;                               // columns of g' per lane half
    constexpr int LDT = D + 4;                             // floats per row of the wave's g' tile: 16-byte aligned rows,
                                                           // neighbouring rows start 4 banks apart
    constexpr int NA = 16;                                 // cell dwords per lane and tile (both widths)
    __shared__ __attribute__((aligned(16))) float lds[PG_WAVES * 1024 > PG_WAVES * 32 * LDT ? PG_WAVES * 1024 : PG_WAVES * 32 * LDT];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    if ((int)blockIdx.x >= a.n_nodes) {                    // the per-node workgroups (long, heaviest first) are launched first
        if (!TIPK_DBG(a.dbg & 64)) pg_role_dc<D>(a, (int)blockIdx.x - a.n_nodes, lane, w);
        return;
    }
    if (TIPK_DBG(a.dbg & 128)) return;
    // ROLE 1: dXB[:, u, :] of one node
    const int n = lane & 31, kh = lane >> 5;               // 32x32x2 operands: row / column, k half
    const int m16 = lane & 15, q16 = lane >> 4;            // 16x16x4 operands: row / column, k quarter
    const int4 nd = a.node_desc[blockIdx.x];               // uniform index: one scalar load
    const int u = __builtin_amdgcn_readfirstlane(nd.x);
    const int s_first = __builtin_amdgcn_readfirstlane(nd.y);
    int n_tiles = __builtin_amdgcn_readfirstlane(nd.z);
    if (TIPK_DBG(a.dbg & 8)) n_tiles = n_tiles < PG_WAVES ? n_tiles : PG_WAVES;
    float* tile = lds + w * 32 * LDT;
    const u32 ldg4 = (u32)a.ld_g * 4u;
    f32x16 acc;                                            // d = 32: dXB[:, u, :] as one 32 x 32 tile
    f32x4 acc_lo, acc_hi;                                  // d = 16: bases 0-15 / 16-31 x 16 columns
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc_lo[i] = 0.f; acc_hi[i] = 0.f; }

    auto getslot = [&](int tl) {
        tl = tl < n_tiles ? tl : n_tiles - 1;
        return a.slots[s_first + tl * 32 + n];
    };
    // operands of tile `tl` (clamped: a tile past the end re-reads the last one and is never multiplied)
    auto load = [&](const int4& s, float (&cv)[NA], float (&gk)[H], float& ss) {
        if (TIPK_DBG(a.dbg & 32)) {
        } else if constexpr (D == 32) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int lj = __shfl(s.z, 2 * kk + kh, 64);
                cv[kk] = pg_ldg(a.cells, (u32)lj * 128u + (u32)n * 4u);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int lj = __shfl(s.z, 4 * ks + q16, 64);
                cv[2 * ks] = pg_ldg(a.cells, (u32)lj * 128u + (u32)m16 * 4u);
                cv[2 * ks + 1] = pg_ldg(a.cells, (u32)lj * 128u + (u32)(16 + m16) * 4u);
            }
        }
        const u32 goff = (u32)s.x * ldg4 + (u32)(H * kh) * 4u;
#pragma unroll
        for (int i = 0; i < H / 4; ++i) {
            if (TIPK_DBG(a.dbg & 16)) break;
            const float4 x = pg_ldg4(a.g, goff + 16u * i);
            gk[4 * i] = x.x; gk[4 * i + 1] = x.y; gk[4 * i + 2] = x.z; gk[4 * i + 3] = x.w;
        }
        ss = __int_as_float(s.y);
    };
    auto compute = [&](const float (&cv)[NA], const float (&gk)[H], float ss) {
        // the g' tile (lane = slot as loaded) -> LDS -> B operand (lane = column)
        pg_wave_sync();                                                     // (the previous tile's reads are issued)
#pragma unroll
        for (int i = 0; i < H / 4; ++i)
            tipk_st4(tile + n * LDT + H * kh + 4 * i,
                     make_float4(gk[4 * i] * ss, gk[4 * i + 1] * ss, gk[4 * i + 2] * ss, gk[4 * i + 3] * ss));
        pg_wave_sync();
        // dXB += cells^T (A: lane = base) . g' tile (B: lane = column), K = the tile's 32 slots
        if (TIPK_DBG(a.dbg & 4)) {
        } else if constexpr (D == 32) {
            float bv[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) bv[kk] = tile[(2 * kk + kh) * LDT + n];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv[kk], bv[kk], acc, 0, 0, 0);
        } else {
            float bv[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) bv[ks] = tile[(4 * ks + q16) * LDT + m16];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                acc_lo = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[2 * ks], bv[ks], acc_lo, 0, 0, 0);
                acc_hi = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[2 * ks + 1], bv[ks], acc_hi, 0, 0, 0);
            }
        }
    };

    if (n_tiles > 0) {
        float cX[NA], gX[H], cY[NA], gY[H];
        float sX = 0.f, sY = 0.f;
        int tl = w;
        int4 slX = getslot(tl), slY = getslot(tl + PG_WAVES);
        load(slX, cX, gX, sX);
        for (; tl < n_tiles; tl += 2 * PG_WAVES) {
            slX = getslot(tl + 2 * PG_WAVES);
            load(slY, cY, gY, sY);
            __builtin_amdgcn_sched_barrier(0);
            compute(cX, gX, sX);
            __builtin_amdgcn_sched_barrier(0);
            if (tl + PG_WAVES < n_tiles) {                                  // (uniform)
                slY = getslot(tl + 3 * PG_WAVES);
                load(slX, cX, gX, sX);
                __builtin_amdgcn_sched_barrier(0);
                compute(cY, gY, sY);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // the waves' tiles -> one, in wave order
    __syncthreads();
    float* red = lds;
    float* o = a.dxb + (int64_t)u * a.dxb_su;
    if constexpr (D == 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[w * 1024 + r * 64 + lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 1024 / PG_THREADS; ++j) {
            const int e = t + j * PG_THREADS;
            float s = red[e];
#pragma unroll
            for (int q = 1; q < PG_WAVES; ++q) s += red[q * 1024 + e];
            const int r = e >> 6, l = e & 63;
            o[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * a.dxb_sb + (l & 31)] = s;
        }
    } else {
        // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
        for (int i = 0; i < 4; ++i) { red[w * 512 + i * 64 + lane] = acc_lo[i]; red[w * 512 + (4 + i) * 64 + lane] = acc_hi[i]; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 512 / PG_THREADS; ++j) {
            const int e = t + j * PG_THREADS;
            float s = red[e];
#pragma unroll
            for (int q = 1; q < PG_WAVES; ++q) s += red[q * 512 + e];
            const int r = e >> 6, l = e & 63;
            o[(int64_t)(16 * (r >> 2) + 4 * (l >> 4) + (r & 3)) * a.dxb_sb + (l & 15)] = s;
        }
    }
}

}  // namespace

extern "C" int tipk_rgcn_pair_grads_supported(int n_bases, int d) {
    return n_bases == 32 && (d == 16 || d == 32);
}

extern "C" int tipk_rgcn_pair_grads(const float* cells, int64_t n_lines, const float* xb, const float* g, int64_t ld_g,
                                    int64_t n_nodes, int n_bases, int d, const int32_t* node_desc, const int32_t* slots,
                                    const int32_t* tile_node, int64_t n_slots, float* dxb, int64_t dxb_sb, int64_t dxb_su, float* pg, int64_t pg_rows,
                                    tipk_stream_t stream) {
    if (!tipk_rgcn_pair_grads_supported(n_bases, d)) return TIPK_EUNSUPPORTED;
    if (!cells || !xb || !g || !node_desc || !slots || !tile_node || !dxb || !pg || n_nodes <= 0 || n_slots <= 0 || n_slots % 32 != 0 ||
        ld_g < d || ld_g % 4 != 0)
        return TIPK_EINVAL;
    if ((reinterpret_cast<uintptr_t>(cells) & 15) || (reinterpret_cast<uintptr_t>(xb) & 15) || (reinterpret_cast<uintptr_t>(g) & 15) ||
        (reinterpret_cast<uintptr_t>(node_desc) & 15) || (reinterpret_cast<uintptr_t>(slots) & 15))
        return TIPK_EINVAL;
    // 32-bit byte offsets into cells, xb and g
    if (pg_rows <= 0 || pg_rows * 128 >= (1LL << 32) || n_lines <= 0 || n_lines * 128 >= (1LL << 32) || n_nodes * 4096 >= (1LL << 32) || n_nodes * ld_g * 4 >= (1LL << 32) ||
        n_nodes > 0x7fffffffLL)
        return TIPK_EUNSUPPORTED;
    PgArgs a;
    a.cells = cells; a.xb = xb; a.g = g; a.ld_g = (int)ld_g;
    a.node_desc = reinterpret_cast<const int4*>(node_desc); a.slots = reinterpret_cast<const int4*>(slots);
    a.tile_node = tile_node; a.n_nodes = (int)n_nodes; a.n_tiles = (int)(n_slots / 32);
    a.dxb = dxb; a.dxb_sb = dxb_sb; a.dxb_su = dxb_su; a.pg = pg;
    a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DP_DEBUG));
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(n_nodes + tipk_ceil_div(a.n_tiles, PG_WAVES));
    if (d == 32) hipLaunchKernelGGL(pair_grads_kernel<32>, dim3(grid), dim3(PG_THREADS), 0, st, a);
    else hipLaunchKernelGGL(pair_grads_kernel<16>, dim3(grid), dim3(PG_THREADS), 0, st, a);
    TIPK_RETURN_LAUNCH();
}
