// Large node sets (config 5: N = 10 000, R = 2 000, E = 50 M): the (relation, node) ROW SUMS of an R-GCN layer are
// assembled in LDS and multiplied where they are -- they never exist in HBM (include/tipk.h section 2h).
//
//     S[(r, v), :]  = sum_{e in (r, v)} table[other_e, :]           (~2.5 edges per row at config 5)
//     T[b, v, :]    = sum_r att[r, b] S[(r, v), :]                  forward: table = X, rows by destination  -> agg = sum_b T_b basis_b
//                                                                   backward: table = D^-1 g', rows by source -> T = d XB
//     d att[r, b]   = sum_v <S[(r, v), :], XB[b, v, :]>             backward only (P2)
//
// Round 4 wrote S (= dY, 10 GB) with a CSR gather (3.4 ms) and read it back for both products (`dy_products`, 3.8 ms);
// the round-5 forward pass multiplied per EDGE (`dest_products`: 0.41 TFLOP, 4.2 ms).  Multiplying the row sums is
// 0.16 TFLOP per product, and the sums cost LDS adds instead of HBM round trips.
//
// Workgroup = 8 waves = 8 nodes x ONE tile of 32 channels (launch order: channel tile major, so that the gathered
// slices of the table -- N x 128 bytes -- stay inside an XCD's L2); TWO workgroups per CU: product 2's cross-wave sum
// puts two barriers into every tile, so the waves of a workgroup gather at the same time and multiply at the same time --
// the other workgroup of the CU is in another phase.  Wave w owns node v = 8 g + w and walks the relation
// tiles r0 = 0, 32, ...:
//   * the tile's edges come as batches of 16 ENTRY WORDS per lane half (half kh owns rows r0 + 2 kk + kh, the row
//     split of the v_mfma_f32_32x32x2_f32 operands), sorted by row: word = inside << 24 | other << 8 | byte offset of the
//     row inside the tile (128 = the unused 33rd column: padding; inside = 0 at the first entry of a row).  One coalesced
//     load per batch; lane j of every 16-lane row holds entry j, and entry j reaches the half's lanes as the DPP row
//     broadcast folded into the instruction that unpacks it;
//   * per entry ONE 128-byte row piece per lane half is loaded (table[other, 32 channels]) and summed in a REGISTER per
//     lane (acc = acc * inside + piece: v_cvt_f32_ubyte3 + v_fma), and the running sum is stored to the wave's private
//     [32 channels][33] tile -- the last store of a row is its sum.  (ds_add_f32 into the tile: 45 ms instead of 4: LDS
//     float atomics run lane by lane on this chip.)  Fixed order: the sums are reproducible;
//   * the tile is the B operand of product 1 (lane = channel) and, read transposed, the A operand of product 2
//     (lane = row); product 2's 8 partial tiles (one per node) are added through LDS in wave order.
#include <stdlib.h>
#include <type_traits>
#include "tipk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;
typedef __attribute__((address_space(3))) float rp_lds_f32_t;

constexpr int RP_WAVES = 8;                             // nodes per workgroup: two workgroups share a CU (see the kernel's header)
constexpr int RP_TLD = 33;                              // tile row stride: conflict-free both ways; column 32 = dump

struct RpArgs {
    const float* table; u32 ld_table_b; int n_nodes, ch;        // gathered rows [n_nodes][ld], ch channels (multiple of 32)
    const float* att; u32 ld_att_b; int R, NB;
    const float* xb; int64_t ld_xb;                             // P2: [NB][n_nodes * ch]
    const int32_t* entries; const int32_t* desc; int n_tiles;   // desc[(node * n_tiles + tile) * 2] = {first batch, batches}
    float* t_out;                                               // [NB][n_nodes * ch]
    float* datt;                                                // P2 slabs: [gridDim.x][R][NB]
    int n_groups;
    int dbg;                                                    // TIPK_DP_DEBUG: 1 no table loads, 2 no LDS adds, 4 no products, 8 plain read-add-write
};

__device__ __forceinline__ float rp_ldg(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float rp_and(float v, u32 mask) { return __uint_as_float(__float_as_uint(v) & mask); }
template <int I>
__device__ __forceinline__ u32 rp_row_bcast(u32 x) {      // lane I of every 16-lane DPP row, broadcast to its row
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + (I & 15), 0xf, 0xf, true);
}
template <int I>
__device__ __forceinline__ void rp_fmac_row_bcast(float& d, float x, float y) {    // d += (lane I of the row's x) * y
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(d) : "v"(x), "v"(y), "n"(I & 15));
}
// LDS words written by one lane of the wave and read by another: LDS operations of a wave execute in order; the fences keep
// the COMPILER from moving them across this point
__device__ __forceinline__ void rp_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int N, int I = 0, typename F>
__device__ __forceinline__ void rp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rp_static_for<N, I + 1>(f);
    }
}

template <int CPL> struct RpGeom {
    static constexpr int CW = 32 * CPL;                         // channels per wave
    static constexpr int TS = CW + (CPL == 1 ? 1 : 2);          // tile row stride in floats (row 32 = padding's dump row)
    static constexpr int TILE = (33 * TS + 3) / 4 * 4;          // floats per wave, 16-byte multiple
};

template <bool P2, int CPL>
__global__ __launch_bounds__(RP_WAVES * 64, 4) void row_products_kernel(RpArgs a) {
    using G = RpGeom<CPL>;
    constexpr int CW = G::CW, TS = G::TS;
    typedef float vec_t __attribute__((ext_vector_type(CPL)));
    typedef __attribute__((address_space(3))) vec_t lds_vec_t;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row = lane & 31, kh = lane >> 5;
    float* tile = lds + w * G::TILE;                    // wave-private: [row of the relation tile (+ dump row)][TS channels]
    float* attl = lds + RP_WAVES * G::TILE;             // P2: [2][32 rows][33] att tiles shared by the workgroup (see begin_tile)
    const int cb = (int)blockIdx.x / a.n_groups, grp = (int)blockIdx.x - cb * a.n_groups;   // channel block, node group
    const int node_raw = grp * RP_WAVES + w;
    const bool wave_on = node_raw < a.n_nodes;          // the last group may have idle waves (they still join barriers)
    const int node = wave_on ? node_raw : a.n_nodes - 1;
    const int NB = a.NB, R = a.R;
    const int64_t NC = (int64_t)a.n_nodes * a.ch;
    const int64_t col0 = (int64_t)node * a.ch + cb * CW;                 // first column of [NB][NC] of this wave
    const u32 tcol_b = (u32)(cb * CW + row * CPL) * 4u;                  // this lane's channels inside a table row
    const u32 tile_c = (u32)(uintptr_t)(rp_lds_f32_t*)tile + (u32)row * (CPL * 4u);   // LDS byte address of tile[0][lane's channels]
    const u32 att_b = (u32)(row < NB ? row : NB - 1) * 4u;

    float xbv[CPL][16];
    if (P2) {                                           // B operand of product 2: XB[b = lane & 31][channel 32 q + 2 kk + kh]
        const int64_t b_off = (int64_t)(row < NB ? row : NB - 1) * a.ld_xb + col0;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                xbv[q][kk] = rp_and(a.xb[b_off + 32 * q + 2 * kk + kh], (row < NB && wave_on) ? 0xffffffffu : 0u);
    }
    f32x16 acc1[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[q][i] = 0.f;

    // The node's batches are consecutive in `entries` (tile after tile) and every tile has at least one: ONE load pipeline
    // over all of them, always two parts of a batch in flight and the entry words some batches ahead.  Every wait for a load
    // then has a FIXED number of younger loads behind it on every path (vmcnt counts in order, and the compiler must assume
    // the path with the fewest): with tiles of zero batches, or loads issued under a condition, each tile's att operand cost
    // a drain of the whole pipeline.  (The loads issued past the node's last batch fetch the next node's pieces -- the array
    // ends with batches of padding -- and are never used.)
    const int32_t* dsc = a.desc + (int64_t)node * a.n_tiles * 2;
    int b = dsc[0];
    int nbat = dsc[1];                                  // (the scalar load of a tile's length travels a tile ahead)
    int tl = 0;
    const u32 word_lane = (u32)(kh * 16 + (lane & 15));
    auto load_word = [&](int batch) { return (u32)a.entries[(int64_t)batch * 32 + word_lane]; };
    const u32 ld_mul = a.ld_table_b >> 8;               // (entry & 0xffff00) * ld_mul = other * bytes per table row
    float acc[CPL];                                     // running sum of the current row of this lane's half
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    float atv[16];
    // Lane j of a 16-lane row unpacks entry j ONCE per batch (row offset in the table, LDS offset, the 0.0 / 1.0 of byte 3);
    // per entry the unpacked value reaches the half's lanes as the DPP row broadcast inside the instruction that uses it:
    // v_add_u32_dpp (load address), v_fmac_f32_dpp per channel (running sum), v_add_u32_dpp (LDS address).
    // A batch is consumed in 16 / DEPTH parts, part g through buffer g & 1.
    constexpr int DEPTH = CPL == 2 ? 4 : 8, NQ = 16 / DEPTH;
    auto issue = [&](auto lo_c, float (&val)[DEPTH][CPL], u32 wv) {
        constexpr int LO = decltype(lo_c)::value;
        const u32 rowb = __umul24(wv & 0x00ffff00u, ld_mul);
        rp_static_for<DEPTH>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const u32 off = rp_row_bcast<LO + j>(rowb) + tcol_b;
            if (TIPK_DBG(a.dbg & 1)) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) val[j][c] = __uint_as_float(off);
            } else {
                const vec_t x = *reinterpret_cast<const vec_t*>(reinterpret_cast<const char*>(a.table) + off);
#pragma unroll
                for (int c = 0; c < CPL; ++c) val[j][c] = x[c];
            }
        });
    };
    auto consume = [&](auto lo_c, float (&val)[DEPTH][CPL], u32 wv) {
        constexpr int LO = decltype(lo_c)::value;
        float keep = (float)(wv >> 24);                 // v_cvt_f32_ubyte3: 0.0 at the first entry of a row (and padding), else 1.0
        const u32 offl = __umul24(wv & 0xffu, (u32)TS); // 4 * row of the tile -> byte offset of that row
        asm volatile("s_nop 1" : "+v"(keep));           // (hand-written DPP below: 2 wait states after the VALU write)
        rp_static_for<DEPTH>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                rp_fmac_row_bcast<LO + j>(val[j][c], keep, acc[c]);    // val = inside * acc + piece
                acc[c] = val[j][c];
            }
            const u32 off = rp_row_bcast<LO + j>(offl) + tile_c;
            vec_t o;
#pragma unroll
            for (int c = 0; c < CPL; ++c) o[c] = acc[c];
            if (!TIPK_DBG(a.dbg & 2)) *reinterpret_cast<lds_vec_t*>((uintptr_t)off) = o;     // the row's last write is its sum
        });
    };
    // A operand of product 1 = att^T (lane = base).  Forward pass: 16 registers per tile, requested when the tile begins.
    // P2 (no registers to spare; the waves meet at two barriers per tile anyway): ONE copy per workgroup in LDS, double
    // buffered -- thread t fetches elements of the NEXT tile when a tile begins and stores them before the tile's first
    // barrier; rows beyond R / bases beyond NB are stored as zeros.
    constexpr int NT = RP_WAVES * 64, PER = 1024 / NT;  // elements of a 32 x 32 tile per thread
    float att_next[PER];
    auto att_elem = [&](int tile_i, int i) {
        const int e = i * NT + t, r = tile_i * 32 + (e >> 5), bb = e & 31;
        const float v = a.att[(int64_t)(r < R ? r : R - 1) * (a.ld_att_b >> 2) + (bb < NB ? bb : NB - 1)];
        return rp_and(v, (r < R && bb < NB && tile_i < a.n_tiles) ? 0xffffffffu : 0u);
    };
    auto att_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) attl[buf * (32 * RP_TLD) + ((i * NT + t) >> 5) * RP_TLD + (t & 31)] = att_next[i];
    };
    if (P2) {
#pragma unroll
        for (int i = 0; i < PER; ++i) att_next[i] = att_elem(0, i);
        att_store(0);
        __syncthreads();
    }
    auto begin_tile = [&]() {
        const int r0 = tl * 32;
        if (P2) {
#pragma unroll
            for (int i = 0; i < PER; ++i) att_next[i] = att_elem(tl + 1 < a.n_tiles ? tl + 1 : tl, i);
        } else {
            const float* att_t = a.att + (int64_t)r0 * (a.ld_att_b >> 2);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int r = r0 + 2 * kk + kh;
                atv[kk] = rp_ldg(att_t, (u32)((r < R ? r : R - 1) - r0) * a.ld_att_b + att_b);
            }
        }
        // zero rows 0 .. 31 of the tile (rows without edges): 32 * TS floats as float4
        constexpr int N4 = 32 * TS / 4;
        static_assert(32 * TS % 4 == 0, "tile zeroing");
#pragma unroll
        for (int i = 0; i < (N4 + 63) / 64; ++i)
            if (i * 64 + lane < N4) tipk_st4(tile + (i * 64 + lane) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        rp_wave_sync();
    };
    auto finish_tile = [&]() {
        const int r0 = tl * 32;
        rp_wave_sync();
        // (1) T += att^T . S : B operand lane = channel 32 q + (lane & 31), k = row
        if (!TIPK_DBG(a.dbg & 4))
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const u32 mk = (r0 + 2 * kk + kh < R && row < NB) ? 0xffffffffu : 0u;
            const float av = P2 ? attl[(tl & 1) * (32 * RP_TLD) + (2 * kk + kh) * RP_TLD + row] : rp_and(atv[kk], mk);
#pragma unroll
            for (int q = 0; q < CPL; ++q)
                acc1[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, tile[(2 * kk + kh) * TS + 32 * q + row], acc1[q], 0, 0, 0);
        }
        if (P2) {
            // (2) d att tile = S . XB^T : A operand lane = row r, k = channel
            f32x16 acc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[row * TS + 32 * q + 2 * kk + kh], xbv[q][kk], acc2, 0, 0, 0);
            // the 8 waves' partial tiles are added through LDS in wave order: each wave puts its own where its tile of row
            // sums was (read for the last time just above; zeroed again when the next tile begins, behind the second barrier)
            att_store((tl + 1) & 1);                        // (free: read two barriers ago)
            rp_wave_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * kh;
                tile[rr * 32 + row] = acc2[r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = i * NT + t;
                float sum = lds[e];
#pragma unroll
                for (int q = 1; q < RP_WAVES; ++q) sum += lds[q * G::TILE + e];
                const int rr = r0 + (e >> 5), bb = e & 31;
                if (rr < R && bb < NB) a.datt[((int64_t)blockIdx.x * R + rr) * NB + bb] = sum;
            }
            __syncthreads();
        }
    };
    float va[DEPTH][CPL], vb[DEPTH][CPL];
    constexpr int WQ = 2;                               // entry words on their way: batches b + 2 .. b + 1 + WQ
    u32 wa = load_word(b), wb = load_word(b + 1), wq[WQ];
#pragma unroll
    for (int q = 0; q < WQ; ++q) wq[q] = load_word(b + 2 + q);
    issue(std::integral_constant<int, 0>{}, va, wa);
    issue(std::integral_constant<int, DEPTH>{}, vb, wa);
    for (tl = 0; tl < a.n_tiles; ++tl) {
        const int nbat_next = dsc[2 * (tl + 1 < a.n_tiles ? tl + 1 : tl) + 1];
        begin_tile();
        int i = 0;
        do {                                            // (at least one batch: see above)
            rp_static_for<NQ>([&](auto gc) {
                constexpr int g = decltype(gc)::value;  // part g of this batch; part g + 2 takes its buffer
                constexpr int nx = (g + 2) % NQ;
                auto lo = std::integral_constant<int, g * DEPTH>{};
                auto lo_nx = std::integral_constant<int, nx * DEPTH>{};
                if constexpr ((g & 1) == 0) { consume(lo, va, wa); issue(lo_nx, va, g + 2 < NQ ? wa : wb); }
                else { consume(lo, vb, wa); issue(lo_nx, vb, g + 2 < NQ ? wa : wb); }
            });
            wa = wb; wb = wq[0];
#pragma unroll
            for (int q = 0; q + 1 < WQ; ++q) wq[q] = wq[q + 1];
            wq[WQ - 1] = load_word(b + 2 + WQ);
            ++b;
        } while (++i < nbat);
        finish_tile();
        nbat = nbat_next;
    }
    // T: C layout col = lane & 31 (channel), rows = bases
    if (wave_on) {
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bb = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (bb < NB) a.t_out[(int64_t)bb * NC + col0 + 32 * q + row] = acc1[q][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// WAVE-UNIFORM entries (channels % 64 == 0).  The fp32 MFMA and the VALU instructions of a SIMD do not overlap on this chip
// (tools/microbench/mfma_valu_overlap.hip): a launch costs at least MFMA busy + VALU busy, and the kernel above spends 2.5
// VALU instructions per entry and 64 channels (address add, two running-sum fmacs, LDS address add for two entries a step).
// Here ONE entry occupies the whole wave -- lane = channel of the 64-channel block -- so everything that depends on the
// entry alone is SCALAR: the entry words come through s_load, the table row offset is the soffset of `buffer_load_dword`,
// the LDS row offset goes into M0 for `ds_write_addtid_b32` (address = M0 + lane * 4) and the inside-row flag is an SGPR
// operand of the one v_fma that is left: 1 VALU instruction per entry (+ 4 on the scalar port).
// Entry format (plan.RowStreamPlan.scalar()): [batch][2][16] int32, plane 0 = byte offset of the table row (other * bytes
// per row), plane 1 = byte offset of the tile row (4 * 65 * row; row 32 = padding's dump row) | 0x3f800000 inside a row;
// one list per (node, tile), sorted by row, padded to 16; desc as above.
constexpr int RPS_TS = 65;                              // tile row stride in floats (conflict-free operand reads both ways)
constexpr int RPS_TILE = (33 * RPS_TS + 3) / 4 * 4;     // floats per wave

template <bool P2>
__global__ __launch_bounds__(RP_WAVES * 64, 4) void row_products_s_kernel(RpArgs a) {
    constexpr int TS = RPS_TS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row = lane & 31, kh = lane >> 5;
    float* tile = lds + w * RPS_TILE;                   // wave-private: [row of the relation tile (+ dump row)][65]
    float* attl = lds + RP_WAVES * RPS_TILE;            // P2: [2][32 rows][33] att tiles shared by the workgroup
    const int cb = (int)blockIdx.x / a.n_groups, grp = (int)blockIdx.x - cb * a.n_groups;
    const int node_raw = grp * RP_WAVES + w;
    const bool wave_on = node_raw < a.n_nodes;
    const int node = wave_on ? node_raw : a.n_nodes - 1;
    const int NB = a.NB, R = a.R;
    const int64_t NC = (int64_t)a.n_nodes * a.ch;
    const int64_t col0 = (int64_t)node * a.ch + cb * 64;
    const u32 voff = (u32)(cb * 64 + lane) * 4u;                         // this lane's channel inside a table row
    u32 tile_base = __builtin_amdgcn_readfirstlane((u32)(uintptr_t)(rp_lds_f32_t*)tile);   // (not const: an asm operand inside a lambda)
    const u32 att_b = (u32)(row < NB ? row : NB - 1) * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.table), 0,
                                                                           (int)((u32)a.n_nodes * a.ld_table_b), 0x00020000);
    float xbv[2][16];
    if (P2) {
        const int64_t b_off = (int64_t)(row < NB ? row : NB - 1) * a.ld_xb + col0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                xbv[q][kk] = rp_and(a.xb[b_off + 32 * q + 2 * kk + kh], (row < NB && wave_on) ? 0xffffffffu : 0u);
    }
    f32x16 acc1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[q][i] = 0.f;
    const int32_t* dsc = a.desc + (int64_t)node * a.n_tiles * 2;
    int b = __builtin_amdgcn_readfirstlane(dsc[0]);
    int nbat = __builtin_amdgcn_readfirstlane(dsc[1]);
    int tl = 0;
    float acc = 0.f;
    float atv[16];
    float va[8], vb[8];
    int w1c[16], wn[32];                                // plane 1 of the current batch; both planes of the next one (SGPRs)
    // (the constant address space makes these s_load_dwordx16: `entries` is read-only for the whole launch)
    typedef const __attribute__((address_space(4))) int32_t* rp_const_i32_t;
    const rp_const_i32_t entries_c = (rp_const_i32_t)(uintptr_t)a.entries;
    auto load_words = [&](int batch, int (&dst)[32]) __attribute__((always_inline)) {
        const rp_const_i32_t eb = entries_c + (int64_t)batch * 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) dst[j] = eb[j];
    };
    auto issue = [&](auto lo_c, float (&val)[8], const int (&w0)[32]) __attribute__((always_inline)) {
        constexpr int LO = decltype(lo_c)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            val[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, w0[LO + j], 0));
    };
    auto consume = [&](auto lo_c, float (&val)[8], const int (&w1)[16]) __attribute__((always_inline)) {
        constexpr int LO = decltype(lo_c)::value;
        const u32 tb = tile_base;                       // (a generic lambda does not capture what only an asm operand names)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int wd = w1[LO + j];
            const float keep = __int_as_float(wd & 0x3f800000);
            acc = fmaf(acc, keep, val[j]);
            // M0 = tile base + the entry's plane-1 word as it is: the inside-row bits (23 .. 29) sit above the LDS address
            // bits the add-TID instruction looks at (tools/microbench/ds_addtid.hip); SALU write of M0 -> add-TID LDS
            // instruction needs one wait state, which the compiler cannot see inside inline assembly
            if (!TIPK_DBG(a.dbg & 2))
                asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0" : : "v"(acc), "s"(tb), "s"(wd) : "memory", "scc");
        }
    };
    constexpr int NT = RP_WAVES * 64, PER = 1024 / NT;
    float att_next[PER];
    auto att_elem = [&](int tile_i, int i) __attribute__((always_inline)) {
        const int e = i * NT + t, r = tile_i * 32 + (e >> 5), bb = e & 31;
        const float v = a.att[(int64_t)(r < R ? r : R - 1) * (a.ld_att_b >> 2) + (bb < NB ? bb : NB - 1)];
        return rp_and(v, (r < R && bb < NB && tile_i < a.n_tiles) ? 0xffffffffu : 0u);
    };
    auto att_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i) attl[buf * (32 * RP_TLD) + ((i * NT + t) >> 5) * RP_TLD + (t & 31)] = att_next[i];
    };
    if (P2) {
#pragma unroll
        for (int i = 0; i < PER; ++i) att_next[i] = att_elem(0, i);
        att_store(0);
        __syncthreads();
    }
    auto begin_tile = [&]() __attribute__((always_inline)) {
        const int r0 = tl * 32;
        if (P2) {
#pragma unroll
            for (int i = 0; i < PER; ++i) att_next[i] = att_elem(tl + 1 < a.n_tiles ? tl + 1 : tl, i);
        } else {
            const float* att_t = a.att + (int64_t)r0 * (a.ld_att_b >> 2);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int r = r0 + 2 * kk + kh;
                atv[kk] = rp_ldg(att_t, (u32)((r < R ? r : R - 1) - r0) * a.ld_att_b + att_b);
            }
            // rows beyond R / bases beyond NB are cleared HERE, and only in a tile that has any (the last one of a relation
            // count that is no multiple of 32; fewer than 32 bases): per MFMA the masks were 64 of a tile's ~200 VALU instructions
            if (r0 + 32 > R || NB < 32) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    atv[kk] = rp_and(atv[kk], (r0 + 2 * kk + kh < R && row < NB) ? 0xffffffffu : 0u);
            }
        }
        constexpr int N4 = 32 * TS / 4;                                  // rows 0 .. 31 as float4
        static_assert(32 * TS % 4 == 0, "tile zeroing");
#pragma unroll
        for (int i = 0; i < (N4 + 63) / 64; ++i)
            if (i * 64 + lane < N4) tipk_st4(tile + (i * 64 + lane) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        rp_wave_sync();
    };
    auto finish_tile = [&]() __attribute__((always_inline)) {
        const int r0 = tl * 32;
        rp_wave_sync();
        if (!TIPK_DBG(a.dbg & 4))
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float av = P2 ? attl[(tl & 1) * (32 * RP_TLD) + (2 * kk + kh) * RP_TLD + row] : atv[kk];
#pragma unroll
            for (int q = 0; q < 2; ++q)
                acc1[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, tile[(2 * kk + kh) * TS + 32 * q + row], acc1[q], 0, 0, 0);
        }
        if (P2) {
            f32x16 acc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[row * TS + 32 * q + 2 * kk + kh], xbv[q][kk], acc2, 0, 0, 0);
            att_store((tl + 1) & 1);
            rp_wave_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * kh;
                tile[rr * 32 + row] = acc2[r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = i * NT + t;
                float sum = lds[e];
#pragma unroll
                for (int q = 1; q < RP_WAVES; ++q) sum += lds[q * RPS_TILE + e];
                const int rr = r0 + (e >> 5), bb = e & 31;
                if (rr < R && bb < NB) a.datt[((int64_t)blockIdx.x * R + rr) * NB + bb] = sum;
            }
            __syncthreads();
        }
    };
    constexpr std::integral_constant<int, 0> c0{};
    constexpr std::integral_constant<int, 8> c8{};
    load_words(b, wn);
#pragma unroll
    for (int j = 0; j < 16; ++j) w1c[j] = wn[16 + j];
    issue(c0, va, wn);
    issue(c8, vb, wn);
    for (tl = 0; tl < a.n_tiles; ++tl) {
        const int nbat_next = __builtin_amdgcn_readfirstlane(dsc[2 * (tl + 1 < a.n_tiles ? tl + 1 : tl) + 1]);
        begin_tile();
        int i = 0;
        do {                                            // (at least one batch per tile)
            load_words(b + 1, wn);                      // (past the node's last batch: the next node's, or padding; unused)
            consume(c0, va, w1c);
            issue(c0, va, wn);
            consume(c8, vb, w1c);
            issue(c8, vb, wn);
#pragma unroll
            for (int j = 0; j < 16; ++j) w1c[j] = wn[16 + j];
            ++b;
        } while (++i < nbat);
        finish_tile();
        nbat = nbat_next;
    }
    if (wave_on) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bb = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (bb < NB) a.t_out[(int64_t)bb * NC + col0 + 32 * q + row] = acc1[q][r];
            }
    }
}

template <bool P2>
int rps_launch(const RpArgs& a, unsigned blocks, hipStream_t st) {
    const size_t lds = (size_t)(RP_WAVES * RPS_TILE + (P2 ? 2 * 32 * RP_TLD : 0)) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)row_products_s_kernel<P2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL((row_products_s_kernel<P2>), dim3(blocks), dim3(RP_WAVES * 64), lds, st, a);
    TIPK_RETURN_LAUNCH();
}

template <int CPL> constexpr size_t rp_lds_bytes(bool p2) {
    return (size_t)(RP_WAVES * RpGeom<CPL>::TILE + (p2 ? 2 * 32 * RP_TLD : 0)) * sizeof(float);
}

template <bool P2, int CPL>
int rp_launch(const RpArgs& a, unsigned blocks, hipStream_t st) {
    const size_t lds = rp_lds_bytes<CPL>(P2);
    hipError_t e = hipFuncSetAttribute((const void*)row_products_kernel<P2, CPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL((row_products_kernel<P2, CPL>), dim3(blocks), dim3(RP_WAVES * 64), lds, st, a);
    TIPK_RETURN_LAUNCH();
}

// channels per lane: two (64-channel blocks, 8-byte loads and LDS stores: half the instructions per gathered byte) where the
// width allows
inline int rp_cpl(int channels) { return channels % 64 == 0 ? 2 : 1; }

}  // namespace

extern "C" int tipk_rgcn_row_products_supported(int64_t n_nodes, int64_t n_rel, int n_bases, int channels) {
    if (n_nodes <= 0 || n_rel <= 0 || n_bases <= 0 || n_bases > 32 || channels <= 0 || channels % 32 != 0) return 0;
    if (n_nodes > 65536 || n_rel > 0x7fffffffLL) return 0;                            // entry word: inside << 24 | other << 8 | row offset
    const int64_t slabs = tipk_ceil_div(n_nodes, RP_WAVES) * (channels / (32 * rp_cpl(channels)));
    if (slabs > 0x7fffffffLL) return 0;
    return 1;
}

extern "C" int64_t tipk_rgcn_row_products_slabs(int64_t n_nodes, int channels) {
    return tipk_ceil_div(n_nodes, RP_WAVES) * (channels / (32 * rp_cpl(channels)));
}

extern "C" int tipk_rgcn_row_products(const float* table, int64_t ld_table, int64_t n_nodes, int channels, const float* att,
                                      int64_t ld_att, int64_t n_rel, int n_bases, const int32_t* entries, const int32_t* desc,
                                      const float* xb, int64_t ld_xb, float* t_out, float* datt_slabs, tipk_stream_t stream) {
    if (!tipk_rgcn_row_products_supported(n_nodes, n_rel, n_bases, channels)) return TIPK_EUNSUPPORTED;
    if (!table || !att || !entries || !desc || !t_out || ld_table < channels || ld_att < n_bases) return TIPK_EINVAL;
    if ((xb == nullptr) != (datt_slabs == nullptr) || (xb && ld_xb < n_nodes * channels)) return TIPK_EINVAL;
    if (n_nodes * ld_table * 4 >= (1LL << 32) || 32 * ld_att * 4 >= (1LL << 31)) return TIPK_EUNSUPPORTED;   // 32-bit byte offsets
    if (ld_table % 64 != 0 || ld_table * 4 / 256 >= 256) return TIPK_EUNSUPPORTED;     // (other << 8) * (row bytes / 256) in 24-bit arithmetic
    if (reinterpret_cast<uintptr_t>(table) & 7) return TIPK_EINVAL;                    // 8-byte loads
    RpArgs a;
    a.table = table; a.ld_table_b = (u32)(ld_table * 4); a.n_nodes = (int)n_nodes; a.ch = channels;
    a.att = att; a.ld_att_b = (u32)(ld_att * 4); a.R = (int)n_rel; a.NB = n_bases;
    a.xb = xb; a.ld_xb = ld_xb;
    a.entries = entries; a.desc = desc; a.n_tiles = (int)tipk_ceil_div(n_rel, 32);
    a.t_out = t_out; a.datt = datt_slabs;
    a.n_groups = (int)tipk_ceil_div(n_nodes, RP_WAVES);
    a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DP_DEBUG));
    const int cpl = rp_cpl(channels);
    const unsigned blocks = (unsigned)(a.n_groups * (channels / (32 * cpl)));
    hipStream_t st = (hipStream_t)stream;
    if (xb) return cpl == 2 ? rp_launch<true, 2>(a, blocks, st) : rp_launch<true, 1>(a, blocks, st);
    return cpl == 2 ? rp_launch<false, 2>(a, blocks, st) : rp_launch<false, 1>(a, blocks, st);
}

extern "C" int tipk_rgcn_row_products_s_supported(int64_t n_nodes, int64_t n_rel, int n_bases, int channels) {
    if (n_nodes <= 0 || n_rel <= 0 || n_bases <= 0 || n_bases > 32 || channels <= 0 || channels % 64 != 0) return 0;
    if (n_rel > 0x7fffffffLL || tipk_ceil_div(n_nodes, RP_WAVES) * (channels / 64) > 0x7fffffffLL) return 0;
    return 1;
}

extern "C" int tipk_rgcn_row_products_s(const float* table, int64_t ld_table, int64_t n_nodes, int channels, const float* att,
                                        int64_t ld_att, int64_t n_rel, int n_bases, const int32_t* entries, const int32_t* desc,
                                        const float* xb, int64_t ld_xb, float* t_out, float* datt_slabs, tipk_stream_t stream) {
    if (!tipk_rgcn_row_products_s_supported(n_nodes, n_rel, n_bases, channels)) return TIPK_EUNSUPPORTED;
    if (!table || !att || !entries || !desc || !t_out || ld_table < channels || ld_att < n_bases) return TIPK_EINVAL;
    if ((xb == nullptr) != (datt_slabs == nullptr) || (xb && ld_xb < n_nodes * channels)) return TIPK_EINVAL;
    if (n_nodes * ld_table * 4 >= (1LL << 31) || 32 * ld_att * 4 >= (1LL << 31)) return TIPK_EUNSUPPORTED;   // 32-bit byte offsets
    RpArgs a;
    a.table = table; a.ld_table_b = (u32)(ld_table * 4); a.n_nodes = (int)n_nodes; a.ch = channels;
    a.att = att; a.ld_att_b = (u32)(ld_att * 4); a.R = (int)n_rel; a.NB = n_bases;
    a.xb = xb; a.ld_xb = ld_xb;
    a.entries = entries; a.desc = desc; a.n_tiles = (int)tipk_ceil_div(n_rel, 32);
    a.t_out = t_out; a.datt = datt_slabs;
    a.n_groups = (int)tipk_ceil_div(n_nodes, RP_WAVES);
    a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DP_DEBUG));
    const unsigned blocks = (unsigned)(a.n_groups * (channels / 64));
    return xb ? rps_launch<true>(a, blocks, (hipStream_t)stream) : rps_launch<false>(a, blocks, (hipStream_t)stream);
}
