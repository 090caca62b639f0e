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
#include "tipk_common.h"

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

// Loads a (ROWS x BK) operand tile into registers: element (r, kk) = p[r*s_r + kk*s_k] or 0.
// K_FAST: consecutive threads walk k (operand is k-contiguous), else they walk the row index.
template <int ROWS, bool K_FAST>
struct TileLoader {
    static constexpr int PER = ROWS * BK / 256;
    float v[PER];
    __device__ __forceinline__ static void coord(int i, int t, int& r, int& kk) {
        if (K_FAST) { kk = t % BK; r = t / BK + (256 / BK) * i; }
        else        { r = t % ROWS; kk = t / ROWS + (256 / ROWS) * i; }
    }
    __device__ __forceinline__ void load(const float* p, int64_t s_r, int64_t s_k, int64_t r0, int64_t r_end,
                                         int64_t k0, int64_t k_end, int t) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int r, kk;
            coord(i, t, r, kk);
            const int64_t gr = r0 + r, gk = k0 + kk;
            v[i] = (gr < r_end && gk < k_end) ? p[gr * s_r + gk * s_k] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float (*lds)[ROWS + PAD], int t) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int r, kk;
            coord(i, t, r, kk);
            lds[kk][r] = v[i];
        }
    }
};

// NBUF = LDS buffers: 2 overlaps the next tile's staging with the MFMAs of a K loop; 1 for products
// whose K fits one tile (Y = att . XB, K = 32): half the LDS, twice the resident workgroups.
// R = register tile per wave: R x R accumulators of 32x32 (R = 2: a 128x128 workgroup tile, each LDS
// operand read feeds two MFMAs -- used for the large square-ish products such as Y = att . XB).
template <int WM, int WN, bool A_KFAST, bool B_KFAST, int NBUF, int R = 1>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = WM * 32 * R, BN = WN * 32 * R;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    __shared__ float As[NBUF][BK][BM + PAD];
    __shared__ float Bs[NBUF][BK][BN + PAD];

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const int64_t z = blockIdx.z / g.ksplit, slab = blockIdx.z % g.ksplit;
    const int64_t k_lo = slab * g.kchunk;
    const int64_t k_hi = (k_lo + g.kchunk < g.k) ? k_lo + g.kchunk : g.k;
    const int64_t tiles_per_q = (k_hi > k_lo) ? (k_hi - k_lo + BK - 1) / BK : 0;
    const int64_t n_tiles = tiles_per_q * g.kbatch;

    const float* a_z = g.a + z * g.a_sz;
    const float* b_z = g.b + z * g.b_sz;

    TileLoader<BM, A_KFAST> la;
    TileLoader<BN, B_KFAST> lb;
    f32x16 acc[R][R];
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    auto fetch = [&](int64_t tile) {
        const int64_t q = tile / tiles_per_q;
        const int64_t k0 = k_lo + (tile % tiles_per_q) * BK;
        la.load(a_z + q * g.a_sq, g.a_sm, g.a_sk, m0, g.m, k0, k_hi, t);
        lb.load(b_z + q * g.b_sq, g.b_sn, g.b_sk, n0, g.n, k0, k_hi, t);
    };

    if (n_tiles > 0) fetch(0);
    for (int64_t tile = 0; tile < n_tiles; ++tile) {
        const int buf = NBUF == 2 ? (int)(tile & 1) : 0;
        // buffer `buf` was last read two steps ago; the barrier of the previous step fences it
        if (NBUF == 1 && tile > 0) __syncthreads();
        la.store(As[buf], t);
        lb.store(Bs[buf], t);
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
    if (i < count)
        for (int64_t k = j; k < n_slabs; k += LANES) s += in[k * slab_stride + i];
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

}  // namespace

extern "C" int tipk_gemm_f32(const tipk_gemm_desc* d, tipk_stream_t stream) {
    if (!d || d->m < 0 || d->n < 0 || d->k < 0 || d->batch < 0 || d->kbatch < 1 || d->ksplit < 1) return TIPK_EINVAL;
    if (d->m == 0 || d->n == 0 || d->batch == 0) return TIPK_OK;
    if (!d->a || !d->b || !d->c) return TIPK_EINVAL;
    if (d->ksplit > 1 && (d->c_in || d->relu)) return TIPK_EINVAL;
    GemmArgs g;
    g.m = d->m; g.n = d->n; g.k = d->k; g.kbatch = d->kbatch; g.ksplit = d->ksplit;
    g.kchunk = tipk_ceil_div(tipk_ceil_div(d->k, d->ksplit), BK) * BK;
    if (g.kchunk == 0) g.kchunk = BK;
    g.a = d->a; g.a_sm = d->a_sm; g.a_sk = d->a_sk; g.a_sq = d->a_sq; g.a_sz = d->a_sz;
    g.b = d->b; g.b_sk = d->b_sk; g.b_sn = d->b_sn; g.b_sq = d->b_sq; g.b_sz = d->b_sz;
    g.c = d->c; g.c_sm = d->c_sm; g.c_sz = d->c_sz; g.c_ss = d->c_ss;
    g.c_in = d->c_in; g.cin_sm = d->cin_sm; g.cin_sz = d->cin_sz;
    g.alpha = d->alpha; g.relu = d->relu;
    hipStream_t st = (hipStream_t)stream;
    if (d->n <= 32) return launch<4, 1>(g, d->batch, st);
    if (d->m <= 32) return launch<1, 4>(g, d->batch, st);
    if (d->m >= 512 && d->n >= 512 && d->ksplit == 1) return launch<2, 2, 2>(g, d->batch, st);   // 128 x 128 tiles
    return launch<2, 2>(g, d->batch, st);
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
