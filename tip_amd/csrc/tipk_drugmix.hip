// The P -> D stage of FMEncoder fused with the drug feature mix (src/layers.py:526-539 with MyHierarchyConv, :229-242):
//
//     mean[d, :] = 1 / max(1, #targets(d)) * sum_{p -> d} H[p, :]            MyHierarchyConv.propagate (aggr = 'mean')
//     x0[d, :]   = cat(xd[d] / d_norm[d], mean[d] W)    or    xd[d] / d_norm[d] + mean[d] W
//
// Round 3 ran this as a gather launch (18 596 edges: 8 us, a launch floor) + the mix launch (5 us) forward and as four
// launches backward (row scaling, a grouped split-K product for d W and d mean, its slab sum, the transposed gather:
// 22 us).  Here each pass is ONE launch:
//   forward   a wavefront per drug: its targets' rows (64 B each at BioSNAP) are summed by 64 / P edge slots of P lanes,
//             the slots are combined by shuffles, the dense map W [p x q] comes from LDS; `mean` is written for the backward.
//   backward  ONE launch for d xd = g / d_norm, d mean = g_pd W^T and d W = mean^T g_pd (+ the transposed gather of d mean
//             on its plan, as before).  Workgroup 0 computes d W for ALL drugs: the reduction over 645 rows that a single
//             workgroup "could not do" in round 3 (22 ... 30 us: one dependent LDS round trip per row and thread) takes the
//             rows through LDS in 2 tiles (coalesced loads, one round trip per tile), a 2 x 2 output patch per thread and
//             1024 / (p q / 4) groups of rows side by side; the other workgroups do the two element-wise maps meanwhile.
//             (Measured and dropped: the transposed gather inside the same launch, a wavefront per source row -- its three
//             dependent round trips (row pointer -> edge list -> rows of g) took 19 us against 5.5 us for the plan kernel.)
// All sums in fixed order: bitwise reproducible.
#include "tipk_common.h"

namespace {

constexpr int DG_MAX = 64;                                    // p, q <= 64 (as tipk_drug_mix_fwd)

struct DgFwdArgs {
    const float* xd; int64_t ld_xd; const float* d_norm;
    const float* h; int64_t ld_h;                             // source rows [n_src x p]
    const int32_t* ptr; const int32_t* src; const float* scale;   // CSR by drug
    const int2* wg;                                           // per workgroup: {first drug, drugs (<= 16)}; 1 drug = all 16 waves on it
    const float* w; int p, q, ne, cat;
    float* out; int64_t ld_out; float* mean;                  // mean [rows x p] contiguous
    int rows;
};

// sum of h[src[e], c] over e in [e0, e1) for this lane's column c = lane % PL, edge slots side by side: 64 edges per batch --
// ONE coalesced load of their source ids (requested a batch ahead), then 64 / SLOTS row loads per lane issued back to back.
// A per-edge chain (id -> row -> add) is two dependent misses per edge: the hub drug of BioSNAP has 2 834 targets.
template <int PL>
__device__ __forceinline__ float dg_row_sum(const float* __restrict__ h, int64_t ld_h, const int32_t* __restrict__ idx, int e0,
                                            int e1, int cc, int lane) {
    constexpr int SLOTS = 64 / PL, STEPS = PL;               // 64 edges = STEPS steps of SLOTS edges
    const int s = lane / PL;
    float acc = 0.f;
    int nxt = e0 + lane < e1 ? idx[e0 + lane] : 0;
    for (int b = e0; b < e1; b += 64) {
        const int cur = nxt;
        nxt = b + 64 + lane < e1 ? idx[b + 64 + lane] : 0;
        float v[STEPS];
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            const int k = SLOTS * t + s;                      // edge of the batch
            const int row = __shfl(cur, k, 64);
            v[t] = b + k < e1 ? h[(int64_t)row * ld_h + cc] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < STEPS; ++t) acc += v[t];
    }
#pragma unroll
    for (int off = PL; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);    // the slots, in a fixed tree
    return acc;
}

template <int PL>                                             // lanes per edge slot: the power of two >= p
__global__ __launch_bounds__(1024) void drug_mix_gather_fwd_kernel(DgFwdArgs a) {
    __shared__ float wl[DG_MAX * DG_MAX];
    __shared__ float ml[16][DG_MAX];
    for (int i = threadIdx.x; i < a.p * a.q; i += 1024) wl[i] = a.w[i];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int2 desc = a.wg[blockIdx.x];
    const bool coop = desc.y == 1;
    const int c = lane % PL;
    const int cc = c < a.p ? c : a.p - 1;
    int d = desc.x + (coop ? 0 : wv);
    const bool live = coop || wv < desc.y;
    float m = 0.f;
    if (live) {
        int e0 = a.ptr[d], e1 = a.ptr[d + 1];
        if (coop) {                                          // the 16 waves share the drug's edges, partial sums through LDS
            const int per = ((e1 - e0 + 15) / 16 + 63) & ~63;
            e0 = e0 + wv * per;
            e1 = e0 + per < e1 ? e0 + per : e1;
        }
        m = e0 < e1 ? dg_row_sum<PL>(a.h, a.ld_h, a.src, e0, e1, cc, lane) : 0.f;
        if (lane < a.p) ml[wv][lane] = m;
    }
    __syncthreads();
    if (coop) {
        if (wv != 0) return;
        m = 0.f;
        if (lane < a.p)
            for (int k = 0; k < 16; ++k) m += ml[k][lane];   // the waves' shares, in order
    } else if (!live) {
        return;
    }
    m *= a.scale[d];
    if (lane < a.p) {
        a.mean[(int64_t)d * a.p + lane] = m;
        ml[wv][lane] = m;
    }
    __builtin_amdgcn_wave_barrier();                          // (ml[wv] is written and read by this wavefront only)
    const int cols = a.cat ? a.ne + a.q : a.ne;
    float* o = a.out + (int64_t)d * a.ld_out;
    const float* x = a.xd + (int64_t)d * a.ld_xd;
    for (int col = lane; col < cols; col += 64) {
        float v = 0.f;
        if (col < a.ne) v = a.d_norm ? x[col] / a.d_norm[d] : x[col];
        const int j = a.cat ? col - a.ne : col;
        if (j >= 0 && j < a.q) {
            float t = 0.f;
            for (int k = 0; k < a.p; ++k) t = fmaf(ml[wv][k], wl[k * a.q + j], t);
            v += t;
        }
        o[col] = v;
    }
}

struct DgBwdArgs {
    const float* g; int64_t ld_g;                             // upstream gradient [rows x (cat ? ne + q : ne)]
    const float* d_norm; const float* mean;                   // mean [rows x p]
    const float* w; int p, q, ne, cat;
    float* g_xd; int64_t ld_gxd;                              // [rows x ne]
    float* g_mean;                                            // [rows x p] = g_pd W^T (contiguous)
    float* g_w;                                               // [p x q]
    int rows, n_wg;                                           // n_wg = workgroups besides workgroup 0
    int tp, kg, tile;                                         // d W: threads per row group (p q / 4), row groups, rows per LDS tile
};

constexpr int DG_STAGE = 32768;                               // floats of an LDS tile of (mean | g_pd) rows: 128 KiB (all 645 rows at p = q = 16)

__global__ __launch_bounds__(1024) void drug_mix_bwd_kernel(DgBwdArgs a) {
    __shared__ float sm[DG_STAGE + 4096];                     // workgroup 0: the tile + partial patches [kg][p q]; others: W
    const int t = threadIdx.x;
    const int qoff = a.cat ? a.ne : 0;                        // columns of g that are g_pd
    if (blockIdx.x == 0) {
        // ------------------------------------------------------------ d W = mean^T g_pd over all rows
        // The rows come through LDS in tiles (coalesced loads, every thread a few elements: ONE round trip per tile); a
        // thread owns a 2 x 2 patch of d W for every kg-th row of the tile.
        const int hq = a.q / 2;
        const int kgi = t / a.tp, pt = t % a.tp;
        const int pi = pt / hq, pj = pt % hq;                 // the patch: rows 2 pi, 2 pi + 1 of W, columns 2 pj, 2 pj + 1
        const int pw = a.p + a.q;                             // floats of a tile row: mean | g_pd
        float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
        const int lr = t / pw, lc = t - lr * pw, lrs = 1024 / pw;          // loader role: column lc of rows lr, lr + lrs, ...
        for (int r0 = 0; r0 < a.rows; r0 += a.tile) {
            const int nr = a.rows - r0 < a.tile ? a.rows - r0 : a.tile;
            __syncthreads();                                  // (the previous tile has been read)
            if (lr < lrs) {
                const float* src = lc < a.p ? a.mean + (int64_t)r0 * a.p + lc : a.g + (int64_t)r0 * a.ld_g + qoff + (lc - a.p);
                const int64_t ld = lc < a.p ? a.p : a.ld_g;
                int r = lr;
                for (; r + 3 * lrs < nr; r += 4 * lrs) {      // four rows in flight per thread
                    const float v0 = src[(int64_t)r * ld], v1 = src[(int64_t)(r + lrs) * ld];
                    const float v2 = src[(int64_t)(r + 2 * lrs) * ld], v3 = src[(int64_t)(r + 3 * lrs) * ld];
                    sm[r * pw + lc] = v0; sm[(r + lrs) * pw + lc] = v1;
                    sm[(r + 2 * lrs) * pw + lc] = v2; sm[(r + 3 * lrs) * pw + lc] = v3;
                }
                for (; r < nr; r += lrs) sm[r * pw + lc] = src[(int64_t)r * ld];
            }
            __syncthreads();
            if (kgi < a.kg) {
                const float* mrow = sm + 2 * pi;
                const float* grow = sm + a.p + 2 * pj;
                int r = kgi;
                for (; r + 3 * a.kg < nr; r += 4 * a.kg) {    // four rows' reads in flight
                    const float2 m0 = *reinterpret_cast<const float2*>(mrow + r * pw), g0 = *reinterpret_cast<const float2*>(grow + r * pw);
                    const float2 m1 = *reinterpret_cast<const float2*>(mrow + (r + a.kg) * pw), g1 = *reinterpret_cast<const float2*>(grow + (r + a.kg) * pw);
                    const float2 m2 = *reinterpret_cast<const float2*>(mrow + (r + 2 * a.kg) * pw), g2 = *reinterpret_cast<const float2*>(grow + (r + 2 * a.kg) * pw);
                    const float2 m3 = *reinterpret_cast<const float2*>(mrow + (r + 3 * a.kg) * pw), g3 = *reinterpret_cast<const float2*>(grow + (r + 3 * a.kg) * pw);
                    s00 = fmaf(m0.x, g0.x, s00); s01 = fmaf(m0.x, g0.y, s01); s10 = fmaf(m0.y, g0.x, s10); s11 = fmaf(m0.y, g0.y, s11);
                    s00 = fmaf(m1.x, g1.x, s00); s01 = fmaf(m1.x, g1.y, s01); s10 = fmaf(m1.y, g1.x, s10); s11 = fmaf(m1.y, g1.y, s11);
                    s00 = fmaf(m2.x, g2.x, s00); s01 = fmaf(m2.x, g2.y, s01); s10 = fmaf(m2.y, g2.x, s10); s11 = fmaf(m2.y, g2.y, s11);
                    s00 = fmaf(m3.x, g3.x, s00); s01 = fmaf(m3.x, g3.y, s01); s10 = fmaf(m3.y, g3.x, s10); s11 = fmaf(m3.y, g3.y, s11);
                }
                for (; r < nr; r += a.kg) {
                    const float2 mv = *reinterpret_cast<const float2*>(mrow + r * pw);
                    const float2 gv = *reinterpret_cast<const float2*>(grow + r * pw);
                    s00 = fmaf(mv.x, gv.x, s00); s01 = fmaf(mv.x, gv.y, s01);
                    s10 = fmaf(mv.y, gv.x, s10); s11 = fmaf(mv.y, gv.y, s11);
                }
            }
        }
        float* part = sm + DG_STAGE;
        if (kgi < a.kg) {
            float* sp = part + kgi * (a.p * a.q);
            sp[(2 * pi) * a.q + 2 * pj] = s00; sp[(2 * pi) * a.q + 2 * pj + 1] = s01;
            sp[(2 * pi + 1) * a.q + 2 * pj] = s10; sp[(2 * pi + 1) * a.q + 2 * pj + 1] = s11;
        }
        __syncthreads();
        for (int o = t; o < a.p * a.q; o += 1024) {
            float s = part[o];
            for (int k = 1; k < a.kg; ++k) s += part[k * (a.p * a.q) + o];       // the row groups, in order
            a.g_w[o] = s;
        }
        return;
    }
    const int b = blockIdx.x - 1;
    float* wl = sm;
    if (a.g_mean)
        for (int i = t; i < a.p * a.q; i += 1024) wl[i] = a.w[i];
    // ---------------------------------------------------------------- d xd = g[:, :ne] / d_norm (a slice per workgroup)
    if (a.g_xd) {
        const int tot = a.rows * a.ne;
        const int per = (tot + a.n_wg - 1) / a.n_wg;
        const int i1 = (b + 1) * per < tot ? (b + 1) * per : tot;
        for (int i = b * per + t; i < i1; i += 1024) {
            const int r = i / a.ne;
            const int c = i - r * a.ne;
            const float v = a.g[(int64_t)r * a.ld_g + c];
            a.g_xd[(int64_t)r * a.ld_gxd + c] = a.d_norm ? v / a.d_norm[r] : v;
        }
    }
    // ---------------------------------------------------------------- d mean = g_pd W^T (a slice per workgroup)
    if (!a.g_mean) return;
    __syncthreads();
    const int tot = a.rows * a.p;
    const int per = (tot + a.n_wg - 1) / a.n_wg;
    const int i1 = (b + 1) * per < tot ? (b + 1) * per : tot;
    for (int i = b * per + t; i < i1; i += 1024) {
        const int r = i / a.p;
        const int c = i - r * a.p;
        const float* gr = a.g + (int64_t)r * a.ld_g + qoff;
        float v = 0.f;
        for (int j = 0; j < a.q; ++j) v = fmaf(gr[j], wl[c * a.q + j], v);
        a.g_mean[i] = v;
    }
}

}  // namespace

extern "C" int tipk_drug_mix_gather_supported(int p, int q) {
    if (p <= 0 || q <= 0 || p > DG_MAX || q > DG_MAX || (p & 1) || (q & 1)) return 0;
    if (p * q / 4 > 1024 || p * q > 4096) return 0;
    return 1;
}

extern "C" int tipk_drug_mix_gather_fwd(const float* xd, int64_t ld_xd, const float* d_norm, const float* h, int64_t ld_h,
                                        const int32_t* ptr, const int32_t* src, const float* scale, const int32_t* wg_desc,
                                        int64_t n_wg, const float* w, int p, int q, int64_t rows, int ne, int cat, float* out,
                                        int64_t ld_out, float* mean, tipk_stream_t stream) {
    if (rows < 0 || ne < 0 || !tipk_drug_mix_gather_supported(p, q) || (!cat && q != ne)) return TIPK_EINVAL;
    if (rows == 0) return TIPK_OK;
    if (!xd || !h || !ptr || !src || !scale || !w || !out || !mean || !wg_desc || n_wg <= 0 || rows > 0x7fffffffLL ||
        n_wg > 0x7fffffffLL || (reinterpret_cast<uintptr_t>(wg_desc) & 7))
        return TIPK_EINVAL;
    DgFwdArgs a;
    a.xd = xd; a.ld_xd = ld_xd; a.d_norm = d_norm; a.h = h; a.ld_h = ld_h; a.ptr = ptr; a.src = src; a.scale = scale;
    a.wg = reinterpret_cast<const int2*>(wg_desc);
    a.w = w; a.p = p; a.q = q; a.ne = ne; a.cat = cat; a.out = out; a.ld_out = ld_out; a.mean = mean; a.rows = (int)rows;
    const dim3 grid((unsigned)n_wg);
    hipStream_t st = (hipStream_t)stream;
    if (p <= 8) hipLaunchKernelGGL(drug_mix_gather_fwd_kernel<8>, grid, dim3(1024), 0, st, a);
    else if (p <= 16) hipLaunchKernelGGL(drug_mix_gather_fwd_kernel<16>, grid, dim3(1024), 0, st, a);
    else if (p <= 32) hipLaunchKernelGGL(drug_mix_gather_fwd_kernel<32>, grid, dim3(1024), 0, st, a);
    else hipLaunchKernelGGL(drug_mix_gather_fwd_kernel<64>, grid, dim3(1024), 0, st, a);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_drug_mix_bwd(const float* g, int64_t ld_g, const float* d_norm, const float* mean, const float* w, int p, int q,
                                 int64_t rows, int ne, int cat, float* g_xd, int64_t ld_gxd, float* g_mean, float* g_w,
                                 tipk_stream_t stream) {
    if (rows < 0 || ne < 0 || !tipk_drug_mix_gather_supported(p, q) || (!cat && q != ne)) return TIPK_EINVAL;
    if (rows == 0) return TIPK_OK;
    if (!g || !mean || !w || !g_w || rows * (int64_t)(ne + p + q) > 0x7fffffffLL) return TIPK_EINVAL;
    DgBwdArgs a;
    a.g = g; a.ld_g = ld_g; a.d_norm = d_norm; a.mean = mean; a.w = w; a.p = p; a.q = q; a.ne = ne; a.cat = cat;
    a.g_xd = g_xd; a.ld_gxd = ld_gxd; a.g_mean = g_mean; a.g_w = g_w;
    a.rows = (int)rows;
    a.tp = p * q / 4;
    a.kg = 1024 / a.tp;
    if (a.kg > 4096 / (p * q)) a.kg = 4096 / (p * q);                        // the partial patches fit their LDS block
    if (a.kg > 16) a.kg = 16;
    if (a.kg < 1) a.kg = 1;
    a.tile = DG_STAGE / (p + q);
    int64_t n_wg = tipk_ceil_div(rows * (int64_t)(ne > p ? ne : p), 1024);
    if (n_wg < 1) n_wg = 1;
    if (n_wg > 240) n_wg = 240;
    a.n_wg = (int)n_wg;
    hipLaunchKernelGGL(drug_mix_bwd_kernel, dim3((unsigned)(1 + n_wg)), dim3(1024), 0, (hipStream_t)stream, a);
    TIPK_RETURN_LAUNCH();
}
