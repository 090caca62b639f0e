// Dense fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: bit-exact fp32 fma chain,
// 64 FLOP/clk/SIMD = the fp32 vector peak, MI355X_MICROARCH "Matrix cores").  Contract:
// include/tipk.h section 2.  All products of the TIP path are small or skinny (K = num_bases = 32,
// or N = out channels <= 128), so the kernel is a plain LDS-tiled loop with register prefetch of
// the next K tile; generic element strides make every transpose / basis reshape copy-free.
//
// Tile: WM x WN waves of 32x32 (one MFMA accumulator each), BK = 32, LDS double-buffered (one
// barrier per K step; the next tile's global loads are in flight during the MFMAs).  LDS tiles
// are k-major (As[k][m], Bs[k][n]) so the MFMA operand fetch (lane l: row l&31 of
// k = 2*kk + (l>>5)) is a conflict-free ds_read_b32 of 32 consecutive floats per half-wave.
#include <stdlib.h>
#include "tipk_common.h"
#include "tipk_slabs.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int PAD = 4;

struct GemmArgs {
    int64_t m, n, k, kbatch, ksplit, kchunk;
    const float* a; int64_t a_sm, a_sk, a_sq, a_sz;
    const float* b; int64_t b_sk, b_sn, b_sq, b_sz;
    float* c; int64_t c_sm, c_sz, c_ss;
    const float* c_in; int64_t cin_sm, cin_sz;
    float alpha;
    int relu;
};

typedef unsigned int u32;

// uniform base pointer + 32-bit byte offset: compiles to `global_load v, v_off, s[base]`
__device__ __forceinline__ float ldg(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float4 ldg4(const float* base, u32 byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Loads a (ROWS x BK) operand tile into registers: element (r, kk) = tile[r*s_r + kk*s_k] or 0, with
// `tile` the (wave-uniform) address of the tile origin and 32-bit offsets inside the tile (the host
// checks 128*s_r + 32*s_k < 2^30) -- one VGPR per address; 64-bit per-element address arithmetic made
// each specialised body 11 KB of code and the grouped kernel twice the instruction cache.
// kfast: consecutive threads walk k (operand is k-contiguous), else they walk the row index.
template <int ROWS>
struct TileLoader {
    static constexpr int PER = ROWS * BK / 256;
    float v[PER];
    __device__ __forceinline__ static void coord(bool kfast, int i, int t, int& r, int& kk) {
        // selects, not if/else: with a runtime flag hipcc built the two candidates as a scratch array
        r = kfast ? t / BK + (256 / BK) * i : t % ROWS;
        kk = kfast ? t % BK : t / ROWS + (256 / ROWS) * i;
    }
    __device__ __forceinline__ void load(bool kfast, const float* tile, u32 s_r, u32 s_k, int rows_left, int k_left, int t) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int r, kk;
            coord(kfast, i, t, r, kk);
            v[i] = (r < rows_left && kk < k_left) ? ldg(tile, ((u32)r * s_r + (u32)kk * s_k) * 4u) : 0.f;
        }
    }
    __device__ __forceinline__ void store(bool kfast, float (*lds)[ROWS + PAD], int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int r, kk;
            coord(kfast, i, t, r, kk);
            lds[kk][r] = v[i];
        }
    }
};

// NBUF = LDS buffers: 2 overlaps the next tile's staging with the MFMAs of a K loop; 1 for products
// whose K fits one tile (Y = att . XB, K = 32): half the LDS, twice the resident workgroups.
// R = register tile per wave: R x R accumulators of 32x32 (R = 2: a 128x128 workgroup tile, each LDS
// operand read feeds two MFMAs -- used for the large square-ish products such as Y = att . XB).
// A_KFAST / B_KFAST: plain arguments -- the single-product kernels pass template constants, the
// grouped kernel runtime flags.
template <int WM, int WN, int NBUF, int R = 1>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, int64_t bx, int64_t by, int64_t bz, float* smem,
                                          const bool A_KFAST, const bool B_KFAST) {
    constexpr int BM = WM * 32 * R, BN = WN * 32 * R;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    float (*As)[BK][BM + PAD] = reinterpret_cast<float (*)[BK][BM + PAD]>(smem);
    float (*Bs)[BK][BN + PAD] = reinterpret_cast<float (*)[BK][BN + PAD]>(smem + NBUF * BK * (BM + PAD));

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int64_t m0 = by * BM, n0 = bx * BN;
    const int64_t z = bz / g.ksplit, slab = bz % g.ksplit;
    const int64_t k_lo = slab * g.kchunk;
    const int64_t k_hi = (k_lo + g.kchunk < g.k) ? k_lo + g.kchunk : g.k;
    const int tiles_per_q = (k_hi > k_lo) ? (int)((k_hi - k_lo + BK - 1) / BK) : 0;
    const int n_tiles = tiles_per_q * (int)g.kbatch;

    // tile origins (uniform); the per-element offsets inside a tile are 32-bit
    const float* a_z = g.a + z * g.a_sz + m0 * g.a_sm + k_lo * g.a_sk;
    const float* b_z = g.b + z * g.b_sz + n0 * g.b_sn + k_lo * g.b_sk;
    const u32 a_sm = (u32)g.a_sm, a_sk = (u32)g.a_sk, b_sn = (u32)g.b_sn, b_sk = (u32)g.b_sk;
    const int m_left = (int)(g.m - m0 < BM ? g.m - m0 : BM), n_left = (int)(g.n - n0 < BN ? g.n - n0 : BN);

    TileLoader<BM> la;
    TileLoader<BN> lb;
    f32x16 acc[R][R];
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    auto fetch = [&](int tile) {
        const int q = tile / tiles_per_q;
        const int kt = tile - q * tiles_per_q;
        const int k_left = (int)(k_hi - k_lo) - kt * BK;
        la.load(A_KFAST, a_z + (int64_t)q * g.a_sq + (int64_t)kt * BK * g.a_sk, a_sm, a_sk, m_left, k_left, t);
        lb.load(B_KFAST, b_z + (int64_t)q * g.b_sq + (int64_t)kt * BK * g.b_sk, b_sn, b_sk, n_left, k_left, t);
    };

    if (n_tiles > 0) fetch(0);
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int buf = NBUF == 2 ? (tile & 1) : 0;
        // buffer `buf` was last read two steps ago; the barrier of the previous step fences it
        if (NBUF == 1 && tile > 0) __syncthreads();
        la.store(A_KFAST, As[buf], t);
        lb.store(B_KFAST, Bs[buf], t);
        __syncthreads();
        if (tile + 1 < n_tiles) fetch(tile + 1);
        const int row = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[R], bv[R];
#pragma unroll
            for (int a = 0; a < R; ++a) av[a] = As[buf][2 * kk + kh][(wm * R + a) * 32 + row];
#pragma unroll
            for (int b = 0; b < R; ++b) bv[b] = Bs[buf][2 * kk + kh][(wn * R + b) * 32 + row];
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < R; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    float* c_z = g.c + z * g.c_sz + slab * g.c_ss;
    const float* cin_z = g.c_in ? g.c_in + z * g.cin_sz : nullptr;
#pragma unroll
    for (int a = 0; a < R; ++a) {
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const int64_t col = n0 + (wn * R + b) * 32 + (lane & 31);
            if (col >= g.n) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t rowi = m0 + (wm * R + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (rowi < g.m) {
                    float v = g.alpha * acc[a][b][r];
                    if (cin_z) v += cin_z[rowi * g.cin_sm + col];
                    if (g.relu) v = fmaxf(v, 0.f);
                    c_z[rowi * g.c_sm + col] = v;
                }
            }
        }
    }
}

template <int WM, int WN, bool A_KFAST, bool B_KFAST, int NBUF, int R = 1>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[NBUF * BK * (WM * 32 * R + WN * 32 * R + 2 * PAD)];
    gemm_body<WM, WN, NBUF, R>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem, A_KFAST, B_KFAST);
}

// ---------------------------------------------------------------------------------------------
// Streaming products: the three large GEMMs of an R-GCN layer are not compute problems but one
// pass over a [relations x nodes*channels] matrix (91 MB at BioSNAP layer 1) with a 32-wide
// reduction or output:
//     Y    = att . XB        m = R,  n = N*d, k = B     (writes Y)        -> thin_k
//     dXB  = att^T . dY      m = B,  n = N*d, k = R     (reads dY)        -> thin_m
//     datt = dY . XB^T       m = R,  n = B,   k = N*d   (reads dY)        -> kk
// The LDS-tiled kernel above runs them at 2.3-2.7 TB/s: a workgroup loads, multiplies and stores
// in turn behind barriers, and at 129-212 VGPRs only 2-3 workgroups share a CU.  Here every WAVE
// works alone -- MFMA operands are loaded straight from global memory in the register layout of
// v_mfma_f32_32x32x2_f32 (A: lane = row, B: lane = column, k split over lane halves and registers),
// the next operands are in flight during the MFMAs, no LDS, no barriers, ~80 VGPRs.
// Each body gets the linear workgroup index inside its problem (so it also runs as a member of
// gemm_f32_group_kernel).
// ---------------------------------------------------------------------------------------------
constexpr int THIN_K_NT = 4;           // 32-column tiles per wave (128 consecutive columns)

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

__device__ __forceinline__ void store_tile(const GemmArgs& g, const f32x16& acc, float* c_z, const float* cin_z,
                                           int64_t m0, int64_t n0, int lane) {
    const int64_t col = n0 + (lane & 31);
    if (col >= g.n) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t rowi = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (rowi < g.m) {
            float v = g.alpha * acc[r];
            if (cin_z) v += cin_z[rowi * g.cin_sm + col];
            if (g.relu) v = fmaxf(v, 0.f);
            c_z[rowi * g.c_sm + col] = v;
        }
    }
}

// Coding rules that keep hipcc from undoing the design (all seen in the ISA of earlier versions):
//  * wave-uniform quantities go through readfirstlane so that operand bases live in SGPRs, and lane
//    offsets are 32-bit BYTE offsets (stream_kind checks the extents): every load is
//    `global_load v, v_off, s[base]` -- one VGPR per address instead of a 64-bit pair;
//  * loads are never guarded and never feed a select: out-of-range rows / columns are clamped (their
//    products land in output rows / columns that are not stored) and out-of-range k is clamped and
//    its A value is zeroed with a bitwise AND (a `cond ? v : 0` makes the compiler sink the load into
//    a branch with a vmcnt(0) wait behind every single load);
//  * the k loop handles both register buffers per iteration (no if/else on the buffer index, which
//    bounced the accumulator between AGPRs and VGPRs every step);
//  * a sched_barrier separates "issue the next operands" from "multiply the current ones": left alone
//    the scheduler sinks each load next to its use to save registers (36 VGPRs, one exposed memory
//    round trip per load).
__device__ __forceinline__ float and_mask(float v, u32 mask) { return __uint_as_float(__float_as_uint(v) & mask); }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// store of one 32x32 accumulator: 64-bit (wave-uniform) address of the tile origin + 32-bit byte
// offsets inside the tile (32 rows of c_sm floats: the host checks 32 * c_sm * 4 < 2^32), so outputs
// larger than 4 GB (synthetic config: Y is 10 GB) are addressed correctly
__device__ __forceinline__ void store_tile32(const GemmArgs& g, const f32x16& acc, float* c_z, const float* cin_z,
                                             int m0, int n0, int lane) {
    const int col = n0 + (lane & 31);
    if (col >= (int)g.n) return;
    float* c_t = c_z + (int64_t)m0 * g.c_sm + n0;
    const float* cin_t = cin_z ? cin_z + (int64_t)m0 * g.cin_sm + n0 : nullptr;
    const u32 c_sm = (u32)g.c_sm * 4u, cin_sm = (u32)g.cin_sm * 4u;
    const u32 cb = (u32)(lane & 31) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m0 + rl < (int)g.m) {
            float v = g.alpha * acc[r];
            if (cin_t) v += ldg(cin_t, (u32)rl * cin_sm + cb);
            if (g.relu) v = fmaxf(v, 0.f);
            *reinterpret_cast<float*>(reinterpret_cast<char*>(c_t) + ((u32)rl * c_sm + cb)) = v;
        }
    }
}

// k <= 32, B rows contiguous in n, one k step: wave = (32-row tile, THIN_K_NT column tiles)
template <bool WIDE>
__device__ __forceinline__ void gemm_thin_k_body(const GemmArgs& g, int64_t block) {
    const int lane = threadIdx.x & 63;
    const int row = lane & 31, kh = lane >> 5;
    const int M = (int)g.m, N = (int)g.n, K = (int)g.k;
    const int m_tiles = (M + 31) / 32;
    const int n_chunks = (N + 32 * THIN_K_NT - 1) / (32 * THIN_K_NT);
    const int gw = uniform((int)block * 4 + (int)(threadIdx.x >> 6));   // the waves of a workgroup share the column chunk
    const int z = uniform(gw / (m_tiles * n_chunks));
    const int rem = gw - z * (m_tiles * n_chunks);
    const int chunk = uniform(rem / m_tiles), mt = uniform(rem - chunk * m_tiles);
    if (z >= g.kbatch) return;                             // kbatch holds the batch count here
    const int m0 = mt * 32;
    const float* __restrict__ a_z = g.a + z * g.a_sz;
    const float* __restrict__ b_z = g.b + z * g.b_sz;
    float* c_z = g.c + z * g.c_sz;
    const float* cin_z = g.c_in ? g.c_in + z * g.cin_sz : nullptr;
    const u32 a_sk = (u32)g.a_sk * 4u, b_sk = (u32)g.b_sk * 4u;
    const u32 aoff = (u32)(m0 + row < M ? m0 + row : M - 1) * (u32)g.a_sm * 4u;
    float av[16];
    u32 koff[16];                                          // row offsets of B, shared by every tile
    // MFMA step kk multiplies k = kmap(kk, lane half).  Narrow body: k = 2 kk + kh (the tiled kernel's order,
    // bit-identical results).  Wide body: k = 16 kh + kk, so a lane's 16 values of A are 16 CONSECUTIVE
    // floats of its row = 4 dwordx4 loads instead of 16 strided dword loads (any pairing of k is a valid
    // order of the fp32 sum as long as A and B agree).
    const bool a_vec = WIDE && g.a_sk == 1 && (g.a_sm & 3) == 0 && (g.a_sz & 3) == 0 && K == 32 &&
                       (reinterpret_cast<uintptr_t>(g.a) & 15) == 0;
    if (a_vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = ldg4(a_z, aoff + (u32)(16 * kh + 4 * q) * 4u);
            av[4 * q + 0] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
        }
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = WIDE ? 16 * kh + kk : 2 * kk + kh;
        const u32 kc = (u32)(k < K ? k : K - 1);
        if (!a_vec) av[kk] = and_mask(ldg(a_z, aoff + kc * a_sk), k < K ? 0xffffffffu : 0u);
        koff[kk] = kc * b_sk;
    }
    const int nb = chunk * (32 * THIN_K_NT);
    // WIDE path (the normal case: 16-byte aligned rows, n % 4 == 0): the wave's 128 columns are taken as
    // four INTERLEAVED 32-column tiles -- tile s = columns 4j + s -- so lane j owns four consecutive columns:
    // the B operand arrives as 16 dwordx4 loads instead of 64 dword loads and the result leaves as 16
    // dwordx4 stores instead of 64.  Measured (tools/microbench/stream_patterns.hip, same work): 24 us vs
    // 36 us -- it is the NUMBER of vector-memory instructions, not their bytes, that bounds these streams.
    // (the host checks: n % 4 == 0, row strides % 4 == 0, 16-byte aligned bases -- thin_k_wide_ok)
    if (WIDE) {
        const int col = nb + 4 * row;                       // first of the lane's four columns
        const u32 cc = (u32)(col < N ? col : N - 4) * 4u;
        float4 b4[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) b4[kk] = ldg4(b_z, koff[kk] + cc);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 a0 = zero16(), a1 = zero16(), a2 = zero16(), a3 = zero16();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b4[kk].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b4[kk].y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b4[kk].z, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], b4[kk].w, a3, 0, 0, 0);
        }
        if (col >= N) return;
        float* c_t = c_z + (int64_t)m0 * g.c_sm + nb;       // 64-bit tile origin + 32-bit offsets (outputs > 4 GB)
        const float* cin_t = cin_z ? cin_z + (int64_t)m0 * g.cin_sm + nb : nullptr;
        const u32 c_sm = (u32)g.c_sm * 4u, cin_sm = (u32)g.cin_sm * 4u, cb = (u32)row * 16u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m0 + rl < M) {
                float4 v = make_float4(g.alpha * a0[r], g.alpha * a1[r], g.alpha * a2[r], g.alpha * a3[r]);
                if (cin_t) {
                    const float4 ci = ldg4(cin_t, (u32)rl * cin_sm + cb);
                    v.x += ci.x; v.y += ci.y; v.z += ci.z; v.w += ci.w;
                }
                if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(reinterpret_cast<char*>(c_t) + ((u32)rl * c_sm + cb)) = v;
            }
        }
        return;
    }
    // All THIN_K_NT operand tiles are requested before the first product: vmcnt counts loads AND stores
    // (gfx9), so a load issued after a tile's stores cannot be waited for without also waiting for the
    // write acknowledgements of those stores -- a ~2 us stall per tile when loads and stores alternate.
    float bt[THIN_K_NT][16];
#pragma unroll
    for (int j = 0; j < THIN_K_NT; ++j) {
        const int col = nb + 32 * j + row;
        const u32 cc = (u32)(col < N ? col : N - 1) * 4u;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) bt[j][kk] = ldg(b_z, koff[kk] + cc);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < THIN_K_NT; ++j) {
        f32x16 acc = zero16();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bt[j][kk], acc, 0, 0, 0);
        store_tile32(g, acc, c_z, cin_z, m0, nb + 32 * j, lane);
    }
}

// m <= 32, B rows contiguous in n, long k (split into slabs): wave = (slab, 32-column tile)
__device__ __forceinline__ void gemm_thin_m_body(const GemmArgs& g, int64_t block) {
    const int lane = threadIdx.x & 63;
    const int row = lane & 31, kh = lane >> 5;
    const int M = (int)g.m, N = (int)g.n;
    const int n_tiles = (N + 31) / 32;
    const int gw = uniform((int)block * 4 + (int)(threadIdx.x >> 6));   // consecutive waves: consecutive column tiles of one slab
    const int zs = uniform(gw / n_tiles);
    const int nt = gw - zs * n_tiles;
    const int z = uniform(zs / (int)g.ksplit);
    const int slab = zs - z * (int)g.ksplit;
    if (z >= g.kbatch) return;
    const int k_lo = slab * (int)g.kchunk;
    const int k_hi = (k_lo + (int)g.kchunk < (int)g.k) ? k_lo + (int)g.kchunk : (int)g.k;
    const float* __restrict__ a_z = g.a + z * g.a_sz;
    const float* __restrict__ b_z = g.b + z * g.b_sz;
    const int n0 = nt * 32;
    const int col = n0 + row;
    const u32 cc = (u32)(col < N ? col : N - 1) * 4u;
    const u32 a_sk = (u32)g.a_sk * 4u, b_sk = (u32)g.b_sk * 4u;
    const u32 ar = (u32)(row < M ? row : M - 1) * (u32)g.a_sm * 4u;
    // two operand buffers (a 4-deep ring was measured 30 % slower: the pass is not latency-bound)
    float a0[16], b0[16], a1[16], b1[16];
#define TIPK_LOAD_AB(A, B, K0)                                                 \
    _Pragma("unroll") for (int kk = 0; kk < 16; ++kk) {                        \
        const int k_ = (K0) + 2 * kk + kh;                                     \
        const u32 kc_ = (u32)(k_ < k_hi ? k_ : k_hi - 1);                      \
        A[kk] = ldg(a_z, ar + kc_ * a_sk);                                     \
        B[kk] = ldg(b_z, kc_ * b_sk + cc);                                     \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define TIPK_MMA(A, B, K0)                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 16; ++kk)                          \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(A[kk], (K0) + 2 * kk + kh < k_hi ? 0xffffffffu : 0u), \
                                                   B[kk], acc, 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc = zero16();
    if (k_lo < k_hi) {
        TIPK_LOAD_AB(a0, b0, k_lo)
#pragma unroll 1
        for (int k0 = k_lo; k0 < k_hi; k0 += 64) {         // a step beyond k_hi multiplies zeros (k mask applied at use)
            TIPK_LOAD_AB(a1, b1, k0 + 32)
            TIPK_MMA(a0, b0, k0)
            TIPK_LOAD_AB(a0, b0, k0 + 64)
            TIPK_MMA(a1, b1, k0 + 32)
        }
    }
#undef TIPK_LOAD_AB
#undef TIPK_MMA
    float* c_z = g.c + z * g.c_sz + (int64_t)slab * g.c_ss;
    const float* cin_z = g.c_in ? g.c_in + z * g.cin_sz : nullptr;
    store_tile32(g, acc, c_z, cin_z, 0, n0, lane);
}

// n <= 32, BOTH operands contiguous in k (A [m x k] rows, B given as [n x k] rows), long k in slabs:
// wave = (slab, 32-row tile).  A lane loads 4 consecutive k of its row per dwordx4; lane half h takes
// k0 + 8q + 4h .. +3, so MFMA step (q, s) multiplies k = k0 + 8q + s (h = 0) and k0 + 8q + 4 + s
// (h = 1) -- any pairing of k is a valid order of the fp32 sum as long as A and B agree.
__device__ __forceinline__ void gemm_kk_body(const GemmArgs& g, int64_t block) {
    const int lane = threadIdx.x & 63;
    const int row = lane & 31, kh = lane >> 5;
    const int M = (int)g.m, N = (int)g.n;
    const int m_tiles = (M + 31) / 32;
    const int gw = uniform((int)block * 4 + (int)(threadIdx.x >> 6));
    const int zs = uniform(gw / m_tiles);
    const int mt = gw - zs * m_tiles;
    const int z = uniform(zs / (int)g.ksplit);
    const int slab = zs - z * (int)g.ksplit;
    if (z >= g.kbatch) return;
    const int k_lo = slab * (int)g.kchunk;
    const int k_hi = (k_lo + (int)g.kchunk < (int)g.k) ? k_lo + (int)g.kchunk : (int)g.k;     // multiples of 4 (host checks)
    const int m0 = mt * 32;
    const float* __restrict__ a_z = g.a + z * g.a_sz;
    const float* __restrict__ b_z = g.b + z * g.b_sz;
    const u32 ar = (u32)(m0 + row < M ? m0 + row : M - 1) * (u32)g.a_sm * 4u;
    const u32 br = (u32)(row < N ? row : N - 1) * (u32)g.b_sn * 4u;
    float4 a0[4], b0[4], a1[4], b1[4];
#define TIPK_LOAD_AB(A, B, K0)                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                            \
        const int k_ = (K0) + 8 * q + 4 * kh;                                  \
        const u32 kc_ = (u32)(k_ < k_hi ? k_ : k_hi - 4) * 4u;                 \
        A[q] = ldg4(a_z, ar + kc_);                                            \
        B[q] = ldg4(b_z, br + kc_);                                            \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define TIPK_KK_STEP(A, B, K0)                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                            \
        const u32 mk_ = (K0) + 8 * q + 4 * kh < k_hi ? 0xffffffffu : 0u;       \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(A[q].x, mk_), B[q].x, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(A[q].y, mk_), B[q].y, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(A[q].z, mk_), B[q].z, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(A[q].w, mk_), B[q].w, acc, 0, 0, 0); \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc = zero16();
    if (k_lo < k_hi) {
        TIPK_LOAD_AB(a0, b0, k_lo)
#pragma unroll 1
        for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
            TIPK_LOAD_AB(a1, b1, k0 + 32)
            TIPK_KK_STEP(a0, b0, k0)
            TIPK_LOAD_AB(a0, b0, k0 + 64)
            TIPK_KK_STEP(a1, b1, k0 + 32)
        }
    }
#undef TIPK_LOAD_AB
#undef TIPK_KK_STEP
    float* c_z = g.c + z * g.c_sz + (int64_t)slab * g.c_ss;
    const float* cin_z = g.c_in ? g.c_in + z * g.cin_sz : nullptr;
    store_tile32(g, acc, c_z, cin_z, m0, 0, lane);
}

enum { STREAM_NONE = 0, STREAM_THIN_K = 1, STREAM_THIN_M = 2, STREAM_KK = 3, STREAM_THIN_K4 = 4 };

inline bool thin_k_wide_ok(const GemmArgs& g, int64_t batch) {
    static_assert(THIN_K_NT == 4, "the wide body owns four interleaved tiles");
    if ((g.n & 3) || ((g.b_sk | g.c_sm | g.b_sz | g.c_sz) & 3)) return false;
    if (g.c_in && ((g.cin_sm | g.cin_sz) & 3)) return false;
    return ((reinterpret_cast<uintptr_t>(g.b) | reinterpret_cast<uintptr_t>(g.c) | reinterpret_cast<uintptr_t>(g.c_in)) & 15) == 0;
}

// which streaming body serves this product (GemmArgs as filled by fill_args, batch = z count)
inline int stream_kind(const GemmArgs& g, int64_t batch) {
    if (g.kbatch != 1) return STREAM_NONE;
    const bool big = g.m * g.n >= (1 << 20) || g.n * g.k >= (1 << 20) || g.m * g.k >= (1 << 20);
    if (!big) return STREAM_NONE;
    // 32-bit element offsets inside one operand (positive strides only)
    const int64_t lim = 0x3fffffffLL;                     // byte offsets fit 32 bits
    if (g.a_sm < 0 || g.a_sk < 0 || g.b_sk < 0 || g.b_sn < 0 || g.m > lim || g.n > lim || g.k > lim) return STREAM_NONE;
    if ((g.m - 1) * g.a_sm + (g.k - 1) * g.a_sk >= lim || (g.k - 1) * g.b_sk + (g.n - 1) * g.b_sn >= lim) return STREAM_NONE;
    if (g.c_sm < 0 || g.cin_sm < 0 || 32 * g.c_sm >= lim || 32 * g.cin_sm >= lim) return STREAM_NONE;   // tile-local C offsets
    if (g.k <= 32 && g.ksplit == 1 && g.b_sn == 1 && (g.m >= 256 || (g.m >= 8 && g.n >= (1 << 16))) && g.n >= 1024)
        return (thin_k_wide_ok(g, batch) && !tipk_option(TIPK_OPT_GEMM_THIN_K_NARROW)) ? STREAM_THIN_K4 : STREAM_THIN_K;
    if (g.m <= 32 && g.b_sn == 1 && g.n >= 1024 && g.k >= 256) return STREAM_THIN_M;
    // lane-per-row dwordx4 loads keep the texture addresser 70 % busy (PMC) and the LDS-tiled kernel is as
    // fast on this shape: the kk body only runs when asked for (TIPK_STREAM_KK=1, tests)
    if (tipk_option(TIPK_OPT_GEMM_STREAM_KK) && g.n <= 32 && g.a_sk == 1 && g.b_sk == 1 && g.m >= 256 && g.k >= 1024 && g.k % 4 == 0 && g.a_sm % 4 == 0 &&
        g.b_sn % 4 == 0 && g.a_sz % 4 == 0 && g.b_sz % 4 == 0 && (reinterpret_cast<uintptr_t>(g.a) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(g.b) & 15) == 0)
        return STREAM_KK;
    return STREAM_NONE;
}

// workgroups (4 waves each) of a streaming problem
inline int64_t stream_blocks(int kind, const GemmArgs& g, int64_t batch) {
    int64_t waves = 0;
    if (kind == STREAM_THIN_K || kind == STREAM_THIN_K4) waves = batch * tipk_ceil_div(g.m, 32) * tipk_ceil_div(g.n, 32 * THIN_K_NT);
    if (kind == STREAM_THIN_M) waves = batch * g.ksplit * tipk_ceil_div(g.n, 32);
    if (kind == STREAM_KK) waves = batch * g.ksplit * tipk_ceil_div(g.m, 32);
    return tipk_ceil_div(waves, 4);
}

template <int KIND>
__global__ __launch_bounds__(256) void gemm_stream_kernel(GemmArgs g) {
    if (KIND == STREAM_THIN_K) gemm_thin_k_body<false>(g, blockIdx.x);
    if (KIND == STREAM_THIN_K4) gemm_thin_k_body<true>(g, blockIdx.x);
    if (KIND == STREAM_THIN_M) gemm_thin_m_body(g, blockIdx.x);
    if (KIND == STREAM_KK) gemm_kk_body(g, blockIdx.x);
}

// Several independent products in ONE launch (tipk_gemm_f32_group): the backward of an R-GCN layer
// needs d basis, d root and the two halves of dX at the same moment, each far too small to fill
// 256 CUs; as separate launches they cost ~5 us apiece on the dependent chain of the step.
// Workgroup -> (problem, tile) through a prefix of block counts; every problem keeps the tile shape
// tipk_gemm_f32 would have picked for it, so the results are bit-identical to separate launches.
constexpr int GROUP_MAX = TIPK_GROUP_MAX;

struct GemmGroupArgs {
    int count;
    int first_block[GROUP_MAX + 1];
    int cfg[GROUP_MAX];                    // shape * 4 + a_kfast * 2 + b_kfast
    int gx[GROUP_MAX], gy[GROUP_MAX];
    GemmArgs g[GROUP_MAX];
};

__global__ __launch_bounds__(256) void gemm_f32_group_kernel(GemmGroupArgs ga) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (160 + 2 * PAD)];
    // constant indices only: a dynamic index into the by-value argument would copy it to scratch
    GemmArgs g = ga.g[0];
    int first = 0, gx = ga.gx[0], gy = ga.gy[0], cfg = ga.cfg[0];
#pragma unroll
    for (int q = 1; q < GROUP_MAX; ++q)
        if (q < ga.count && (int)blockIdx.x >= ga.first_block[q]) {
            g = ga.g[q]; first = ga.first_block[q]; gx = ga.gx[q]; gy = ga.gy[q]; cfg = ga.cfg[q];
        }
    const int local = (int)blockIdx.x - first;
    const int64_t bx = local % gx, by = (local / gx) % gy, bz = local / (gx * gy);
    const bool akf = (cfg & 2) != 0, bkf = (cfg & 1) != 0;
    switch (cfg >= 100 ? cfg : cfg >> 2) {
    case 0: gemm_body<4, 1, 2>(g, bx, by, bz, smem, akf, bkf); break;
    case 1: gemm_body<1, 4, 2>(g, bx, by, bz, smem, akf, bkf); break;
    case 2: gemm_body<2, 2, 2>(g, bx, by, bz, smem, akf, bkf); break;
    case 100 + STREAM_THIN_M: gemm_thin_m_body(g, local); break;
    case 100 + STREAM_KK: gemm_kk_body(g, local); break;
    }
}

template <int WM, int WN, int R = 1>
int launch(const GemmArgs& g, int64_t batch, hipStream_t st) {
    constexpr int BM = WM * 32 * R, BN = WN * 32 * R;
    const int64_t gx = tipk_ceil_div(g.n, BN), gy = tipk_ceil_div(g.m, BM), gz = batch * g.ksplit;
    if (gx > 0x7fffffffLL || gy > 65535 || gz > 65535) return TIPK_EUNSUPPORTED;
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz), block(256);
    const bool akf = g.a_sk == 1 || g.a_sm != 1;      // k-contiguous (or generic) -> walk k
    const bool bkf = g.b_sk == 1 && g.b_sn != 1;      // only when B is truly k-contiguous
    const bool one_tile = g.kbatch == 1 && g.kchunk <= BK;
#define TIPK_GEMM_GO(A, B)                                                                              \
    do {                                                                                                \
        if (one_tile) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, A, B, 1, R>), grid, block, 0, st, g); \
        else hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, A, B, (R > 1 ? 1 : 2), R>), grid, block, 0, st, g); \
    } while (0)
    if (akf && bkf) TIPK_GEMM_GO(true, true);
    else if (akf) TIPK_GEMM_GO(true, false);
    else if (bkf) TIPK_GEMM_GO(false, true);
    else TIPK_GEMM_GO(false, false);
#undef TIPK_GEMM_GO
    TIPK_RETURN_LAUNCH();
}

// 64 consecutive elements x 4 slab lanes per workgroup: lane j adds slabs j, j+4, ... in order,
// then the four lanes are combined in a fixed order through LDS (deterministic).  Optional fused
// epilogue: out = relu?( alpha * row_scale[i / cols] * sum + addend[i] (+ out[i]) ).
template <int LANES>   // slab lanes per element: 4 (few slabs, many elements) or 16 (many slabs, few elements)
__global__ __launch_bounds__(64 * LANES) void sum_slabs_kernel(const float* __restrict__ in, int64_t n_slabs,
                                                               int64_t slab_stride, int64_t count, float alpha,
                                                               int accumulate, const float* __restrict__ row_scale,
                                                               int64_t cols, const float* __restrict__ addend,
                                                               int relu, float* __restrict__ out) {
    __shared__ float red[LANES][64];
    const int e = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    float s = 0.f;
    if (i < count) {
        // four loads in flight per lane, added in the original order (bitwise the same sum): a slab sum is a
        // chain of dependent ~1 us round trips otherwise (256 partial slabs / 16 lanes = 16 of them)
        const float* p = in + i;
        int64_t k = j;
        for (; k + 3 * LANES < n_slabs; k += 4 * LANES) {
            const float v0 = p[k * slab_stride], v1 = p[(k + LANES) * slab_stride];
            const float v2 = p[(k + 2 * LANES) * slab_stride], v3 = p[(k + 3 * LANES) * slab_stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < n_slabs; k += LANES) s += p[k * slab_stride];
    }
    red[j][e] = s;
    __syncthreads();
    if (j == 0 && i < count) {
        s = red[0][e];
#pragma unroll
        for (int q = 1; q < LANES; ++q) s += red[q][e];
        s *= alpha;
        if (row_scale) s *= row_scale[i / cols];
        if (addend) s += addend[i];
        if (accumulate) s += out[i];
        if (relu) s = fmaxf(s, 0.f);
        out[i] = s;
    }
}

struct SlabGroupArgs {
    int count;
    int first_block[GROUP_MAX + 1];
    SlabArgs s[GROUP_MAX];
};

// tipk_sum_slabs_group: several ordered slab sums in one launch.  1024 threads = `lanes` slab lanes
// x (1024 / lanes) elements; per element the additions happen in exactly the order of
// sum_slabs_kernel<lanes>.
__global__ __launch_bounds__(1024) void sum_slabs_group_kernel(SlabGroupArgs sa) {
    __shared__ float red[1024];
    SlabArgs a = sa.s[0];
    int first = 0;
#pragma unroll
    for (int q = 1; q < GROUP_MAX; ++q)
        if (q < sa.count && (int)blockIdx.x >= sa.first_block[q]) { a = sa.s[q]; first = sa.first_block[q]; }
    slab_sum_body(a, (int)blockIdx.x - first, red);
}

// ---------------------------------------------------------------------------------------------
// Products whose reduction is split over the 16 WAVES OF ONE WORKGROUP (tipk_gemm_wg_group).
//
// The backward pass of an R-GCN layer ends in four small products with few output tiles and a reduction of
// 645 .. 1 056 terms (d basis = X^T dXB per base, d root = X^T g, dX = sum_b dXB_b basis_b^T + g root^T).  As grouped
// split-K GEMMs they were cut into slabs over workgroups and finished by a second launch (tipk_sum_slabs_group): 10.4 +
// 7.6 us of the step per layer for 0.2 GFLOP, both launches a chain of dependent round trips.  Here ONE workgroup owns a
// 32 x 32 output tile: its K tiles (32 terms each; batch terms of a reduce-batch product and an optional SECOND product
// added on top count as further tiles) are dealt to the 16 waves in contiguous runs, every wave loads its operands
// straight from global memory in the register layout of v_mfma_f32_32x32x2_f32 (next tile in flight during the MFMAs),
// the partial tiles are added through LDS in wave order (deterministic) and the epilogue (alpha, c_in, ReLU, gate) is
// applied once.  Ordered slab sums that are ready at the same point (the d att slabs of tipk_rgcn_node_products) ride in
// the same launch as further workgroups.
constexpr int WGK_MAX = 4, WGK_SUMS = 3, WGK_WAVES = 16;

struct WgkJob {
    const float* a; int64_t a_sm, a_sk, a_sq, a_sz;
    const float* b; int64_t b_sk, b_sn, b_sq, b_sz;
    const float* a2; int64_t a2_sm, a2_sk;              // nullable second product (k2 terms), added to the same tile
    const float* b2; int64_t b2_sk, b2_sn;
    float* c; int64_t c_sm, c_sz;
    const float* c_in; int64_t cin_sm, cin_sz;
    const float* gate; int64_t gate_sm, gate_sz;        // nullable: out = gate > 0 ? value : 0
    int m, n, k, k2, kbatch;
    int mt, nt;                                         // output tiles
    int tiles_q, n_kt, per, nw;                         // K tiles per batch term, in all, per wave; waves with work
    int vec1, vec2;                                     // both operands of the (first / second) product are contiguous in k
    float alpha; int relu;
};
struct WgkArgs {
    int n_gemm, n_sum;
    int first[WGK_MAX + WGK_SUMS + 1];
    WgkJob g[WGK_MAX];
    SlabArgs s[WGK_SUMS];
};

__global__ __launch_bounds__(1024) void gemm_wgk_group_kernel(WgkArgs wa) {
    __shared__ __attribute__((aligned(16))) float red[WGK_WAVES * 1024];
    const int blk = (int)blockIdx.x;
    if (blk >= wa.first[WGK_MAX]) {                     // the slab sums follow the products
        SlabArgs sa = wa.s[0];
        int first = wa.first[WGK_MAX];
#pragma unroll
        for (int q = 1; q < WGK_SUMS; ++q)
            if (q < wa.n_sum && blk >= wa.first[WGK_MAX + q]) { sa = wa.s[q]; first = wa.first[WGK_MAX + q]; }
        slab_sum_body(sa, blk - first, red);
        return;
    }
    WgkJob g = wa.g[0];
    int first = 0;
#pragma unroll
    for (int q = 1; q < WGK_MAX; ++q)
        if (q < wa.n_gemm && blk >= wa.first[q]) { g = wa.g[q]; first = wa.first[q]; }
    const int local = blk - first;
    const int nt_i = local % g.nt, mt_i = (local / g.nt) % g.mt, z = local / (g.nt * g.mt);
    const int t = threadIdx.x, lane = t & 63, row = lane & 31, kh = lane >> 5;
    const int w = uniform(t >> 6);
    const int m0 = mt_i * 32, n0 = nt_i * 32;
    const int n_first = g.kbatch * g.tiles_q;           // K tiles of the first product
    const u32 am = (u32)(m0 + row < g.m ? m0 + row : g.m - 1), bn = (u32)(n0 + row < g.n ? n0 + row : g.n - 1);
    const u32 m_mask = m0 + row < g.m ? 0xffffffffu : 0u;

    f32x16 acc = zero16();
    float a0[16], b0[16], a1[16], b1[16];
    int left0 = 0, left1 = 0;                           // valid k of the tile in each buffer (negative: the vector layout)
    // Which k a lane's register kk holds is free as long as both operands agree: when BOTH are contiguous in k (dX: rows of
    // dXB against rows of basis) lane half kh takes k0 + 16 kh .. + 15 as four 16-byte loads -- a dword per (row, k) made
    // every load touch 32 lines for 4 useful bytes each: 31.7 us for the dX product alone (tools/bench_wg_gemm.py)
    auto load = [&](int tile, float (&av)[16], float (&bv)[16], int& left) {
        const bool second = tile >= n_first;
        const int tq = second ? tile - n_first : tile;
        const int q = second ? 0 : tq / g.tiles_q;
        const int kt = second ? tq : tq - q * g.tiles_q;
        const int kk_total = second ? g.k2 : g.k;
        const int k0 = kt * 32;
        left = kk_total - k0 < 32 ? kk_total - k0 : 32;
        const float* ap = second ? g.a2 : g.a + (int64_t)z * g.a_sz + (int64_t)q * g.a_sq;
        const float* bp = second ? g.b2 : g.b + (int64_t)z * g.b_sz + (int64_t)q * g.b_sq;
        const u32 s_am = (u32)(second ? g.a2_sm : g.a_sm), s_ak = (u32)(second ? g.a2_sk : g.a_sk);
        const u32 s_bk = (u32)(second ? g.b2_sk : g.b_sk), s_bn = (u32)(second ? g.b2_sn : g.b_sn);
        const u32 ao = am * s_am, bo = bn * s_bn;
        if (second ? g.vec2 : g.vec1) {                 // (uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kb = 16 * kh + 4 * j;
                const u32 kc = (u32)(k0 + (kb < left ? kb : left - 4));
                const float4 xa = ldg4(ap, (ao + kc) * 4u), xb = ldg4(bp, (bo + kc) * 4u);
                av[4 * j] = xa.x; av[4 * j + 1] = xa.y; av[4 * j + 2] = xa.z; av[4 * j + 3] = xa.w;
                bv[4 * j] = xb.x; bv[4 * j + 1] = xb.y; bv[4 * j + 2] = xb.z; bv[4 * j + 3] = xb.w;
            }
            left = -left;
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int kl = 2 * kk + kh;
            const u32 kc = (u32)(k0 + (kl < left ? kl : left - 1));
            av[kk] = ldg(ap, (ao + kc * s_ak) * 4u);
            bv[kk] = ldg(bp, (kc * s_bk + bo) * 4u);
        }
    };
    auto mma = [&](const float (&av)[16], const float (&bv)[16], int left) {
        const int lim = left < 0 ? -left - 16 * kh : left - kh;     // register kk is valid: kk < lim (vector) / 2 kk < lim
        const int step = left < 0 ? 1 : 2;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(and_mask(av[kk], step * kk < lim ? m_mask : 0u), bv[kk], acc, 0, 0, 0);
    };
    const int t_lo = w * g.per, t_hi = t_lo + g.per < g.n_kt ? t_lo + g.per : g.n_kt;
    if (t_lo < t_hi) {
        load(t_lo, a0, b0, left0);
#pragma unroll 1
        for (int tile = t_lo; tile < t_hi; tile += 2) {
            if (tile + 1 < t_hi) load(tile + 1, a1, b1, left1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0, left0);
            __builtin_amdgcn_sched_barrier(0);
            if (tile + 1 < t_hi) {
                if (tile + 2 < t_hi) load(tile + 2, a0, b0, left0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1, left1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) red[w * 1024 + r * 64 + lane] = acc[r];
    }
    __syncthreads();
    {
        float s = g.nw > 0 ? red[t] : 0.f;
        for (int q = 1; q < g.nw; ++q) s += red[q * 1024 + t];
        const int r = t >> 6, l = t & 63;
        const int rr = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), cc = n0 + (l & 31);
        if (rr < g.m && cc < g.n) {
            s *= g.alpha;
            if (g.c_in) s += g.c_in[(int64_t)z * g.cin_sz + (int64_t)rr * g.cin_sm + cc];
            if (g.relu) s = fmaxf(s, 0.f);
            if (g.gate && !(g.gate[(int64_t)z * g.gate_sz + (int64_t)rr * g.gate_sm + cc] > 0.f)) s = 0.f;
            g.c[(int64_t)z * g.c_sz + (int64_t)rr * g.c_sm + cc] = s;
        }
    }
}

}  // namespace

// validates a descriptor and converts it; returns 1 when there is nothing to compute
static int fill_args(const tipk_gemm_desc* d, GemmArgs& g) {
    if (!d || d->m < 0 || d->n < 0 || d->k < 0 || d->batch < 0 || d->kbatch < 1 || d->ksplit < 1) return TIPK_EINVAL;
    if (d->m == 0 || d->n == 0 || d->batch == 0) return 1;
    if (!d->a || !d->b || !d->c) return TIPK_EINVAL;
    if (d->ksplit > 1 && (d->c_in || d->relu)) return TIPK_EINVAL;
    // offsets inside one operand tile are 32-bit byte offsets (strides must be non-negative)
    if (d->a_sm < 0 || d->a_sk < 0 || d->b_sk < 0 || d->b_sn < 0) return TIPK_EINVAL;
    if (128 * d->a_sm + 32 * d->a_sk >= (1LL << 30) || 128 * d->b_sn + 32 * d->b_sk >= (1LL << 30)) return TIPK_EUNSUPPORTED;
    g.m = d->m; g.n = d->n; g.k = d->k; g.kbatch = d->kbatch; g.ksplit = d->ksplit;
    g.kchunk = tipk_ceil_div(tipk_ceil_div(d->k, d->ksplit), BK) * BK;
    if (g.kchunk == 0) g.kchunk = BK;
    g.a = d->a; g.a_sm = d->a_sm; g.a_sk = d->a_sk; g.a_sq = d->a_sq; g.a_sz = d->a_sz;
    g.b = d->b; g.b_sk = d->b_sk; g.b_sn = d->b_sn; g.b_sq = d->b_sq; g.b_sz = d->b_sz;
    g.c = d->c; g.c_sm = d->c_sm; g.c_sz = d->c_sz; g.c_ss = d->c_ss;
    g.c_in = d->c_in; g.cin_sm = d->cin_sm; g.cin_sz = d->cin_sz;
    g.alpha = d->alpha; g.relu = d->relu;
    return TIPK_OK;
}

extern "C" int tipk_gemm_f32(const tipk_gemm_desc* d, tipk_stream_t stream) {
    GemmArgs g;
    const int rc = fill_args(d, g);
    if (rc != TIPK_OK) return rc > 0 ? TIPK_OK : rc;
    hipStream_t st = (hipStream_t)stream;
    const int kind = tipk_option(TIPK_OPT_GEMM_NO_STREAM) ? STREAM_NONE : stream_kind(g, d->batch);
    if (kind != STREAM_NONE) {
        const int64_t blocks = stream_blocks(kind, g, d->batch);
        if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
        g.kbatch = d->batch;                            // the streaming bodies find the batch count here
        if (kind == STREAM_THIN_K)
            hipLaunchKernelGGL(gemm_stream_kernel<STREAM_THIN_K>, dim3((unsigned)blocks), dim3(256), 0, st, g);
        else if (kind == STREAM_THIN_K4)
            hipLaunchKernelGGL(gemm_stream_kernel<STREAM_THIN_K4>, dim3((unsigned)blocks), dim3(256), 0, st, g);
        else if (kind == STREAM_THIN_M)
            hipLaunchKernelGGL(gemm_stream_kernel<STREAM_THIN_M>, dim3((unsigned)blocks), dim3(256), 0, st, g);
        else
            hipLaunchKernelGGL(gemm_stream_kernel<STREAM_KK>, dim3((unsigned)blocks), dim3(256), 0, st, g);
        TIPK_RETURN_LAUNCH();
    }
    if (d->n <= 32) return launch<4, 1>(g, d->batch, st);
    if (d->m <= 32) return launch<1, 4>(g, d->batch, st);
    if (d->m >= 512 && d->n >= 512 && d->ksplit == 1) return launch<2, 2, 2>(g, d->batch, st);   // 128 x 128 tiles
    return launch<2, 2>(g, d->batch, st);
}

extern "C" int tipk_gemm_f32_group(const tipk_gemm_desc* descs, int32_t count, tipk_stream_t stream) {
    if (count < 0 || count > GROUP_MAX || (count > 0 && !descs)) return TIPK_EINVAL;
    GemmGroupArgs ga;
    ga.count = 0;
    int64_t blocks = 0;
    const bool no_stream = tipk_option(TIPK_OPT_GEMM_NO_STREAM) != 0;
    for (int i = 0; i < count; ++i) {
        GemmArgs& g = ga.g[ga.count];
        const int rc = fill_args(descs + i, g);
        if (rc < 0) return rc;
        if (rc > 0) continue;
        int kind = no_stream ? STREAM_NONE : stream_kind(g, descs[i].batch);
        if (kind == STREAM_THIN_K || kind == STREAM_THIN_K4) kind = STREAM_NONE;   // never grouped in the path: keeps the grouped kernel's registers down
        if (kind != STREAM_NONE) {
            const int64_t nb = stream_blocks(kind, g, descs[i].batch);
            if (nb > 0x3fffffffLL) return TIPK_EUNSUPPORTED;
            g.kbatch = descs[i].batch;
            ga.cfg[ga.count] = 100 + kind;
            ga.gx[ga.count] = 1;
            ga.gy[ga.count] = 1;
            ga.first_block[ga.count] = (int)blocks;
            blocks += nb;
            if (blocks > 0x3fffffffLL) return TIPK_EUNSUPPORTED;
            ++ga.count;
            continue;
        }
        const int shape = g.n <= 32 ? 0 : (g.m <= 32 ? 1 : 2);
        const int bm = shape == 0 ? 128 : (shape == 1 ? 32 : 64), bn = shape == 0 ? 32 : (shape == 1 ? 128 : 64);
        const bool akf = g.a_sk == 1 || g.a_sm != 1;
        const bool bkf = g.b_sk == 1 && g.b_sn != 1;
        const int64_t gx = tipk_ceil_div(g.n, bn), gy = tipk_ceil_div(g.m, bm), gz = descs[i].batch * g.ksplit;
        if (gx > 0x7fffffLL || gy > 65535 || gz > 65535 || gx * gy * gz > 0x3fffffffLL) return TIPK_EUNSUPPORTED;
        ga.cfg[ga.count] = shape * 4 + (akf ? 2 : 0) + (bkf ? 1 : 0);
        ga.gx[ga.count] = (int)gx;
        ga.gy[ga.count] = (int)gy;
        ga.first_block[ga.count] = (int)blocks;
        blocks += gx * gy * gz;
        if (blocks > 0x3fffffffLL) return TIPK_EUNSUPPORTED;
        ++ga.count;
    }
    if (ga.count == 0) return TIPK_OK;
    ga.first_block[ga.count] = (int)blocks;
    hipLaunchKernelGGL(gemm_f32_group_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ga);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_sum_slabs_group(const tipk_slab_sum_desc* descs, int32_t count, tipk_stream_t stream) {
    if (count < 0 || count > GROUP_MAX || (count > 0 && !descs)) return TIPK_EINVAL;
    SlabGroupArgs sa;
    sa.count = 0;
    int64_t blocks = 0;
    for (int i = 0; i < count; ++i) {
        int rc;
        const int64_t nb = fill_slab_args(descs[i], sa.s[sa.count], &rc);
        if (rc != TIPK_OK) return rc;
        if (nb == 0) continue;
        sa.first_block[sa.count] = (int)blocks;
        blocks += nb;
        if (blocks > 0x3fffffffLL) return TIPK_EUNSUPPORTED;
        ++sa.count;
    }
    if (sa.count == 0) return TIPK_OK;
    sa.first_block[sa.count] = (int)blocks;
    hipLaunchKernelGGL(sum_slabs_group_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, sa);
    TIPK_RETURN_LAUNCH();
}

static int fill_wgk(const tipk_wg_gemm_desc& d, WgkJob& j) {
    const tipk_gemm_desc& p = d.p;
    if (p.m <= 0 || p.n <= 0 || p.k < 0 || p.batch <= 0 || p.kbatch < 1 || p.ksplit != 1) return TIPK_EINVAL;
    if (!p.a || !p.b || !p.c || p.a_sm < 0 || p.a_sk < 0 || p.b_sk < 0 || p.b_sn < 0) return TIPK_EINVAL;
    const bool second = d.a2 != nullptr;
    if (second && (!d.b2 || d.k2 <= 0 || p.batch != 1 || d.a2_sm < 0 || d.a2_sk < 0 || d.b2_sk < 0 || d.b2_sn < 0)) return TIPK_EINVAL;
    const int64_t lim = 1LL << 30;                       // element offsets inside one term: 32-bit byte offsets
    if (p.m > lim || p.n > lim || p.k > lim || p.m * p.a_sm + p.k * p.a_sk >= lim || p.k * p.b_sk + p.n * p.b_sn >= lim) return TIPK_EUNSUPPORTED;
    if (second && (d.k2 > lim || p.m * d.a2_sm + d.k2 * d.a2_sk >= lim || d.k2 * d.b2_sk + p.n * d.b2_sn >= lim)) return TIPK_EUNSUPPORTED;
    const int64_t tiles_q = tipk_ceil_div(p.k, 32);
    const int64_t n_kt = p.kbatch * tiles_q + (second ? tipk_ceil_div(d.k2, 32) : 0);
    const int64_t mt = tipk_ceil_div(p.m, 32), nt = tipk_ceil_div(p.n, 32);
    // (one workgroup = one CU = 4 matrix pipes: 64 K tiles are 16 per pipe = 7.5 us of MFMA time; a 114-tile reduction measured 33 us)
    if (n_kt < 1 || n_kt > WGK_WAVES * 4 || mt * nt * p.batch > 4096) return TIPK_EUNSUPPORTED;
    j.a = p.a; j.a_sm = p.a_sm; j.a_sk = p.a_sk; j.a_sq = p.a_sq; j.a_sz = p.a_sz;
    j.b = p.b; j.b_sk = p.b_sk; j.b_sn = p.b_sn; j.b_sq = p.b_sq; j.b_sz = p.b_sz;
    j.a2 = d.a2; j.a2_sm = d.a2_sm; j.a2_sk = d.a2_sk; j.b2 = d.b2; j.b2_sk = d.b2_sk; j.b2_sn = d.b2_sn;
    j.c = p.c; j.c_sm = p.c_sm; j.c_sz = p.c_sz;
    j.c_in = p.c_in; j.cin_sm = p.cin_sm; j.cin_sz = p.cin_sz;
    j.gate = d.gate; j.gate_sm = d.gate_sm; j.gate_sz = d.gate_sz;
    j.m = (int)p.m; j.n = (int)p.n; j.k = (int)p.k; j.k2 = second ? (int)d.k2 : 0; j.kbatch = (int)p.kbatch;
    j.mt = (int)mt; j.nt = (int)nt;
    j.tiles_q = (int)tiles_q; j.n_kt = (int)n_kt;
    j.per = (int)tipk_ceil_div(n_kt, WGK_WAVES);
    j.nw = (int)tipk_ceil_div(n_kt, j.per);
    j.alpha = p.alpha; j.relu = p.relu;
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    j.vec1 = p.a_sk == 1 && p.b_sk == 1 && p.k % 4 == 0 && ((p.a_sm | p.b_sn | p.a_sq | p.b_sq | p.a_sz | p.b_sz) & 3) == 0 && al16(p.a) && al16(p.b);
    j.vec2 = second && d.a2_sk == 1 && d.b2_sk == 1 && d.k2 % 4 == 0 && ((d.a2_sm | d.b2_sn) & 3) == 0 && al16(d.a2) && al16(d.b2);
    return TIPK_OK;
}

extern "C" int tipk_gemm_wg_group_supported(const tipk_wg_gemm_desc* d) {
    WgkJob j;
    return d && fill_wgk(*d, j) == TIPK_OK;
}

extern "C" int tipk_gemm_wg_group(const tipk_wg_gemm_desc* descs, int32_t count, const tipk_slab_sum_desc* sums,
                                  int32_t n_sums, tipk_stream_t stream) {
    if (count < 0 || count > WGK_MAX || n_sums < 0 || n_sums > WGK_SUMS || (count > 0 && !descs) || (n_sums > 0 && !sums)) return TIPK_EINVAL;
    WgkArgs wa;
    wa.n_gemm = count; wa.n_sum = 0;
    int64_t blocks = 0;
    for (int i = 0; i < count; ++i) {
        const int rc = fill_wgk(descs[i], wa.g[i]);
        if (rc != TIPK_OK) return rc;
        wa.first[i] = (int)blocks;
        blocks += (int64_t)wa.g[i].mt * wa.g[i].nt * descs[i].p.batch;
    }
    for (int i = count; i <= WGK_MAX; ++i) wa.first[i] = (int)blocks;
    for (int i = 0; i < n_sums; ++i) {
        int rc;
        const int64_t nb = fill_slab_args(sums[i], wa.s[wa.n_sum], &rc);
        if (rc != TIPK_OK) return rc;
        if (nb == 0) continue;
        wa.first[WGK_MAX + wa.n_sum] = (int)blocks;
        blocks += nb;
        if (blocks > 0x3fffffffLL) return TIPK_EUNSUPPORTED;
        ++wa.n_sum;
    }
    for (int i = wa.n_sum; i <= WGK_SUMS; ++i) wa.first[WGK_MAX + i] = (int)blocks;
    if (blocks == 0) return TIPK_OK;
    hipLaunchKernelGGL(gemm_wgk_group_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, wa);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_sum_slabs_ex(const float* in, int64_t n_slabs, int64_t slab_stride, int64_t count, float alpha,
                                 int accumulate, const float* row_scale, int64_t cols, const float* addend, int relu,
                                 float* out, tipk_stream_t stream) {
    if (n_slabs < 0 || count < 0 || (row_scale && cols <= 0)) return TIPK_EINVAL;
    if (count == 0) return TIPK_OK;
    if (!out || (n_slabs > 0 && !in)) return TIPK_EINVAL;
    const int64_t blocks = tipk_ceil_div(count, 64);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    if (n_slabs >= 32 && blocks < 2048)
        hipLaunchKernelGGL(sum_slabs_kernel<16>, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, in, n_slabs,
                           slab_stride, count, alpha, accumulate, row_scale, cols, addend, relu, out);
    else
        hipLaunchKernelGGL(sum_slabs_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, n_slabs,
                           slab_stride, count, alpha, accumulate, row_scale, cols, addend, relu, out);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_sum_slabs(const float* in, int64_t n_slabs, int64_t slab_stride, int64_t count, float alpha,
                              int accumulate, float* out, tipk_stream_t stream) {
    return tipk_sum_slabs_ex(in, n_slabs, slab_stride, count, alpha, accumulate, nullptr, 0, nullptr, 0, out, stream);
}
