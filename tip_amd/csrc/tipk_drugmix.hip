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
    const int2* wg;                                           // per workgroup: {first, n | W << 8}: n <= 16 / W rows, W waves each
    const int32_t* order;                                     // rows in the order the workgroups take them (nullable: identity)
    const float* w; int p, q, ne, cat;
    float* out; int64_t ld_out; float* mean;                  // mean [rows x p] contiguous
    int rows;
    // XB tail (tipk_drug_mix_gather_xb_fwd): the first R-GCN layer's row-local products of the rows this workgroup finishes
    const float* basis; const float* root; int n_bases, d_out;    // [n_bases][cols][d_out], [cols][d_out]; cols = width of out
    float* xb;                                                // [..][n_bases][32] node-major, rows padded to 32 columns
    float* xroot;                                             // [rows][d_out]
};

constexpr int DG_XMAX = 128;                                  // widest mixed row the XB tail takes
constexpr int DG_XLD = DG_XMAX + 2;                           // LDS row stride of the staged rows (A-operand reads: no 4-way conflicts)
constexpr int DG_CP = 4;                                      // workgroups per block of rows in the XB launch (column parts)

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

template <int PL, bool XB>                                    // lanes per edge slot: the power of two >= p
__global__ __launch_bounds__(1024) void drug_mix_gather_fwd_kernel(DgFwdArgs a) {
    __shared__ float wl[DG_MAX * DG_MAX];
    __shared__ float ml[16][DG_MAX];
    __shared__ float xl[XB ? 16 * DG_XLD : 1];                // XB: the workgroup's rows of x0 (rows it does not own stay zero)
    __shared__ int rowid[16];                                 // XB: the rows' ids
    for (int i = threadIdx.x; i < a.p * a.q; i += 1024) wl[i] = a.w[i];
    if constexpr (XB)
        for (int i = threadIdx.x; i < 16 * DG_XLD; i += 1024) xl[i] = 0.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XB: DG_CP workgroups share a block of rows -- each repeats the (cheap) gather and takes every DG_CP-th tile of the products,
    // part 0 writes x0 / mean.  The products of 16 rows are 3.5 us of fp32 MFMA on ONE CU; 41 row blocks alone left 215 CUs idle
    const int cpart = XB ? (int)(blockIdx.x % DG_CP) : 0;
    const int2 desc = a.wg[XB ? blockIdx.x / DG_CP : blockIdx.x];
    // W wavefronts share a row's edges (partial sums through LDS, added in wave order): rows are dealt by edge count -- one
    // with more than 512 edges has the workgroup to itself (W = 16: one BioSNAP drug has 2 834 protein targets), rows with
    // 65 ... 512 edges go four to a workgroup (W = 4; 84 drugs have 130 ... 250 targets: a wavefront alone walked them as four
    // dependent batches, the long pole of the launch), the others sixteen (W = 1)
    const int nrow = desc.y & 255;
    const int W = (desc.y >> 8) ? (desc.y >> 8) : (nrow == 1 ? 16 : 1);
    const int li = wv / W, sub = wv - li * W;
    const int c = lane % PL;
    const int cc = c < a.p ? c : a.p - 1;
    const bool live = li < nrow;
    int d = 0;
    if (live) d = a.order ? a.order[desc.x + li] : desc.x + li;
    float m = 0.f;
    if (live) {
        int e0 = a.ptr[d], e1 = a.ptr[d + 1];
        if (W > 1) {
            const int per = ((e1 - e0 + W - 1) / W + 63) & ~63;
            e0 = e0 + sub * per;
            e1 = e0 + per < e1 ? e0 + per : e1;
        }
        m = e0 < e1 ? dg_row_sum<PL>(a.h, a.ld_h, a.src, e0, e1, cc, lane) : 0.f;
        if (lane < a.p) ml[wv][lane] = m;
    }
    __syncthreads();
    const int cols = a.cat ? a.ne + a.q : a.ne;
    const bool mine = live && sub == 0;                       // this wavefront finishes row d
    if (mine && W > 1) {
        m = 0.f;
        if (lane < a.p)
            for (int k = 0; k < W; ++k) m += ml[wv + k][lane];   // the waves' shares, in order
    }
    if (!XB && !mine) return;
    if (mine) {
        m *= a.scale[d];
        if (lane < a.p) {
            if (cpart == 0) a.mean[(int64_t)d * a.p + lane] = m;
            ml[wv][lane] = m;
        }
        __builtin_amdgcn_wave_barrier();                      // (ml[wv] is written and read by this wavefront only)
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
            if (cpart == 0) o[col] = v;
            if constexpr (XB) xl[li * DG_XLD + col] = v;
        }
        if constexpr (XB)
            if (lane == 0) rowid[li] = d;
    }
    if constexpr (XB) {
        // XB[d] = x0[d] basis (node-major, 32-column rows) and x0[d] root for the workgroup's <= 16 rows: 16 x 16 tiles of
        // v_mfma_f32_16x16x4_f32 (A: lane = row l & 15, k = l >> 4, out of LDS once; B: the weights straight from L2 in the
        // operand layout, lane = column l & 15, k = l >> 4), K = the mixed row in index order -- a k-ordered fma chain
        __syncthreads();
        const int m16 = lane & 15, q16 = lane >> 4;
        const int ksteps = cols >> 2;                         // cols % 4 == 0, <= DG_XMAX (host)
        const int tpb = a.d_out >> 4;                         // 16-column tiles per basis (d_out = 16 | 32)
        const int xb_tiles = a.n_bases * tpb, n_tiles = xb_tiles + tpb;
        const float* xrow = xl + m16 * DG_XLD + q16;
        const int ldw = a.d_out;
        // ALL the weight loads of a wavefront's tiles are requested before the first product
        constexpr int TPW = 2;                                // tiles of one batch per wavefront (66 tiles / 4 parts / 16 waves at BioSNAP)
        constexpr int TS = 16 * DG_CP;                        // tile stride of a wavefront
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        for (int t0 = cpart + DG_CP * wv; t0 < n_tiles; t0 += TS * TPW) {
            f32x4 acc[TPW];
            const float* wp[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                int tile = t0 + TS * i;
                tile = tile < n_tiles ? tile : n_tiles - 1;    // clamped, unconditional loads
                const bool is_root = tile >= xb_tiles;
                const int b = is_root ? 0 : tile / tpb;
                const int c0 = (is_root ? tile - xb_tiles : tile - b * tpb) << 4;
                wp[i] = (is_root ? a.root : a.basis + (int64_t)b * cols * ldw) + c0 + m16 + q16 * ldw;
            }
            for (int k0 = 0; k0 < ksteps; k0 += 16) {         // 16 k-steps (64 columns of the row) per batch of loads
                float av[16], bw[TPW][16];
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const int kc = k0 + ks < ksteps ? k0 + ks : ksteps - 1;
                        bw[i][ks] = wp[i][4 * kc * ldw];
                    }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const int kc = k0 + ks < ksteps ? k0 + ks : ksteps - 1;
                    const float v = xrow[4 * kc];
                    av[ks] = k0 + ks < ksteps ? v : 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                    for (int i = 0; i < TPW; ++i)             // five independent accumulators: the 40-cycle latency is covered
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bw[i][ks], acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int tile = t0 + TS * i;
                if (tile < n_tiles) {                          // (wave-uniform)
                    const bool is_root = tile >= xb_tiles;
                    const int b = is_root ? 0 : tile / tpb;
                    const int c0 = (is_root ? tile - xb_tiles : tile - b * tpb) << 4;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = 4 * q16 + v;            // C: row = 4 (l >> 4) + v, column = l & 15
                        if (r < nrow) {
                            if (is_root) a.xroot[(int64_t)rowid[r] * a.d_out + c0 + m16] = acc[i][v];
                            else a.xb[((int64_t)rowid[r] * a.n_bases + b) * 32 + c0 + m16] = acc[i][v];
                        }
                    }
                }
            }
        }
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


// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the backward pass of the whole P -> D stage in ONE launch (it was three: tipk_drug_mix_bwd, the transposed
// gather of d mean on its plan, the grouped products of GCNConv 2's backward pass -- 7.9 + 9.1 + 8.8 us for KFLOPs):
//
//     d xd   = g[:, :ne] / d_norm                                   (slices, all row workgroups)
//     d W_h  = mean^T g_pd                                          (workgroup 0: the 645-row reduction, as tipk_drug_mix_bwd)
//     d mean = g_pd W_h^T                                           [rows x p]: every row workgroup computes ALL of it into its
//                                                                   LDS (165 K fma, 41 KB of g out of L2) -- nothing crosses
//                                                                   workgroups inside the launch
//     g_h[s] = sum_{e: s -> d} tw[e] d mean[d]                      the transposed P -> D gather, a wavefront per kept source row,
//                                                                   CSR by source row, rows of d mean out of LDS
//     gw[s]  = (g_h[s] W2) * row_scale[s]                           GCNConv 2's  g W  on the row it belongs to (p -> c1 columns)
//     d W2 partial [c1 x p] = agg[rows of the workgroup]^T g_h,  d b2 partial [p] = column sums of g_h
//
// The partials are slabs, one per row workgroup, summed in order by the next launch's riders (tipk_gather_sum_riders).
// Fixed order everywhere: bitwise reproducible.
struct PdBwdArgs {
    DgBwdArgs m;                                              // the drug-mix half (g_mean unused)
    const int32_t* tptr; const int32_t* tdst; const float* tw; int n_src;     // CSR by kept source row: drug, 1 / #targets(drug)
    const float* agg; int64_t ld_agg; int c1;                 // [n_src x c1]: GCNConv 2's aggregated input rows
    const float* w2; int64_t w2_sk, w2_sn;                    // W2 (k < p, n < c1) at w2[k * w2_sk + n * w2_sn]
    const float* row_scale;                                   // nullable
    float* gw; int64_t ld_gw;                                 // [n_src x c1]
    float* dw2; float* db2;                                   // slabs [n_row_wg][c1][p], [n_row_wg][p]
    const int32_t* wg_rows;                                   // [n_row_wg + 1] first source row of every row workgroup
    int rows_per_wg;
    int dbg;                                                  // debug builds: 1 = no d W_h, 2 = no gather, 4 = no d W2 partials, 8 = no prelude
};

// LDS of a row workgroup (floats): W_h^T [q x p] | W2 [p x c1] | g_h [64 x p] | the workgroup's rows of agg [64 x c1] | the gathered
// rows [64 x q] | a chunk of weighted rows of g_pd [PD_CHUNK_E x q'] | the chunk's edge ids, weights + the row pointers;
// workgroups 0 .. PD_WH_WGS - 1 use the first DG_STAGE + 4096 floats as drug_mix_bwd_kernel does
constexpr int PD_OFF_WH = 0;                                  // <= 4096
constexpr int PD_OFF_W2 = 4096;                               // <= 4096
constexpr int PD_OFF_GH = 8192;                               // <= 4096
constexpr int PD_OFF_AGG = 12288;                             // <= 4096
constexpr int PD_OFF_TQ = 16384;                              // 64 x q <= 4096
constexpr int PD_CHUNK_E = 512;                               // edges per chunk at most (fewer when 512 q > PD_PROD floats)
constexpr int PD_PROD = 16384;
constexpr int PD_OFF_PROD = 20480;                            // chunk x q <= PD_PROD floats
constexpr int PD_OFF_EID = PD_OFF_PROD + PD_PROD;             // ids [512] | weights [512] | row pointers [65]
constexpr int PD_ROWS = 4;                                    // (rows_per_wg = 64)
constexpr int PD_WH_WGS = 8;                                  // workgroups (= slabs) of d W_h
constexpr int PD_LDS = PD_OFF_EID + 2 * PD_CHUNK_E + 128;
static_assert(PD_LDS >= DG_STAGE + 4096 && PD_LDS * 4 <= 160 * 1024, "LDS layout");

template <int PL>
__global__ __launch_bounds__(1024) void pd_stage_bwd_kernel(PdBwdArgs A) {
    __shared__ float sm[PD_LDS];
    const DgBwdArgs& a = A.m;
    const int t = threadIdx.x;
    const int qoff = a.cat ? a.ne : 0;
    if (blockIdx.x < PD_WH_WGS) {
        if (TIPK_DBG(A.dbg & 1)) return;
        // ------------------------------------------------------------ d W_h = mean^T g_pd: PD_WH_WGS workgroups take a share of the
        // rows each and leave a slab [p x q] (summed in order by the caller's riders).  One workgroup for all 645 rows was the
        // long pole of the launch (five dependent tile-load round trips + a 40-step chain per thread).
        const int share = (a.rows + PD_WH_WGS - 1) / PD_WH_WGS;
        const int ra0 = (int)blockIdx.x * share;
        const int ra1 = ra0 + share < a.rows ? ra0 + share : a.rows;
        const int hq = a.q / 2;
        const int kgi = t / a.tp, pt = t % a.tp;
        const int pi = pt / hq, pj = pt % hq;
        const int pw = a.p + a.q;
        float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
        const int lr = t / pw, lc = t - lr * pw, lrs = 1024 / pw;
        for (int r0 = ra0; r0 < ra1; r0 += a.tile) {
            const int nr = ra1 - r0 < a.tile ? ra1 - r0 : a.tile;
            __syncthreads();
            if (lr < lrs) {
                const float* src = lc < a.p ? a.mean + (int64_t)r0 * a.p + lc : a.g + (int64_t)r0 * a.ld_g + qoff + (lc - a.p);
                const int64_t ld = lc < a.p ? a.p : a.ld_g;
                int r = lr;
                for (; r + 3 * lrs < nr; r += 4 * lrs) {
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
                for (; r + a.kg < nr; r += 2 * a.kg) {
                    const float2 m0 = *reinterpret_cast<const float2*>(mrow + r * pw), g0 = *reinterpret_cast<const float2*>(grow + r * pw);
                    const float2 m1 = *reinterpret_cast<const float2*>(mrow + (r + a.kg) * pw), g1 = *reinterpret_cast<const float2*>(grow + (r + a.kg) * pw);
                    s00 = fmaf(m0.x, g0.x, s00); s01 = fmaf(m0.x, g0.y, s01); s10 = fmaf(m0.y, g0.x, s10); s11 = fmaf(m0.y, g0.y, s11);
                    s00 = fmaf(m1.x, g1.x, s00); s01 = fmaf(m1.x, g1.y, s01); s10 = fmaf(m1.y, g1.x, s10); s11 = fmaf(m1.y, g1.y, s11);
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
        float* slab = a.g_w + (int64_t)blockIdx.x * (a.p * a.q);
        for (int o = t; o < a.p * a.q; o += 1024) {
            float s = part[o];
            for (int k = 1; k < a.kg; ++k) s += part[k * (a.p * a.q) + o];
            slab[o] = s;
        }
        return;
    }
    const int b = blockIdx.x - PD_WH_WGS;
    float* wlt = sm + PD_OFF_WH;                              // W_h^T [q x p]
    float* w2l = sm + PD_OFF_W2;                              // W2 [p x c1]
    float* gh = sm + PD_OFF_GH;
    float* aggl = sm + PD_OFF_AGG;                            // the workgroup's rows of agg
    float* tq = sm + PD_OFF_TQ;                               // the workgroup's gathered rows [rows_per_wg][q]
    float* prod = sm + PD_OFF_PROD;                           // weighted rows of g_pd of a chunk of edges [PD_CHUNK_E][q]
    int* eid = reinterpret_cast<int*>(sm + PD_OFF_EID);       // ids | weights of the chunk's edges
    float* ewt = sm + PD_OFF_EID + PD_CHUNK_E;
    int* rptr = reinterpret_cast<int*>(sm + PD_OFF_EID + 2 * PD_CHUNK_E);   // row pointers of the workgroup's rows [rows_per_wg + 1]
    const int p = a.p, q = a.q, c1 = A.c1;
    // a workgroup's rows: consecutive, at most 64 of them and -- unless a single row has more -- at most PD_CHUNK_E edges
    // (the plan's deal: BioSNAP has a block of proteins that 65 ... 94 drugs target each -- 64 ROWS per workgroup gave three
    // workgroups 2 200 ... 3 300 edges, seven chunks each, against a mean of 326)
    const int s_first = A.wg_rows[b];
    const int n_mine = A.wg_rows[b + 1] - s_first;
    // Everything that depends on nothing inside the launch is requested FIRST, in one batch: the weights, the workgroup's rows of
    // agg, the row pointers of its source rows.
    for (int i = t; i < p * q; i += 1024) wlt[(i % q) * p + i / q] = a.w[i];
    for (int i = t; i < p * c1; i += 1024) w2l[i] = A.w2[(int64_t)(i / c1) * A.w2_sk + (int64_t)(i % c1) * A.w2_sn];
    for (int i = t; i < n_mine * c1; i += 1024) aggl[i] = A.agg[(int64_t)(s_first + i / c1) * A.ld_agg + (i % c1)];
    if (t <= n_mine) rptr[t] = A.tptr[s_first + t];
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
    if (TIPK_DBG(A.dbg & 2)) return;
    __syncthreads();
    // ---------------------------------------------------------------- the transposed gather on g_pd ITSELF (q columns), the dense map
    // after it:  g_h[s] = (sum_e tw[e] g_pd[dst[e]]) W_h^T  -- 256 fma per source row instead of d mean for all 645 drugs in every
    // workgroup.  EDGE-parallel: the workgroup's rows are consecutive, so are their edges; (edge, column) pairs are dealt to the
    // threads -- every row load of up to 512 edges in flight at once, whatever the rows' lengths (a hub protein has 94 edges, the
    // mean row 5: a wavefront per row walked a hub as a chain of dependent loads, 12 of the launch's 18 us) -- the weighted rows
    // go to LDS, and thread (row, column) adds its row's entries there in edge order: fixed order, reproducible.
    const int E0 = rptr[0], E1 = rptr[n_mine];
    constexpr int EPP = 1024 / PL;                            // edges per pass (PL threads per edge: the power of two >= q)
    int chunk_e = PD_PROD / q;                                // edges per chunk: whole passes, <= PD_CHUNK_E, chunk_e q <= PD_PROD
    chunk_e = chunk_e < PD_CHUNK_E ? chunk_e : PD_CHUNK_E;
    chunk_e = chunk_e / EPP * EPP;                            // (EPP <= 128, PD_PROD / q >= 256)
    const int passes = chunk_e / EPP;
    const int cl = t % PL, el = t / PL;
    const int cq = cl < q ? cl : q - 1;
    // thread -> the (row, column) sums it owns: item = t + 1024 k  (rows_per_wg x q <= 4096 items)
    float racc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = E0; c0 < E1; c0 += chunk_e) {
        const int ne_c = E1 - c0 < chunk_e ? E1 - c0 : chunk_e;
        if (t < chunk_e) {
            const bool in = t < ne_c;
            eid[t] = in ? A.tdst[c0 + t] : 0;
            ewt[t] = in ? A.tw[c0 + t] : 0.f;
        }
        __syncthreads();
        if (TIPK_DBG(A.dbg & 16)) return;
        for (int p0 = 0; p0 < passes; p0 += 8) {              // 8 row loads per thread in flight
            float v[8], w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int e = (p0 + u) * EPP + el;
                e = e < chunk_e ? e : chunk_e - 1;            // clamped, unconditional
                w8[u] = ewt[e];
                v[u] = a.g[(int64_t)eid[e] * a.ld_g + qoff + cq];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = (p0 + u) * EPP + el;
                if (cl < q && e < chunk_e) prod[e * q + cl] = w8[u] * v[u];
            }
            if ((p0 + 8) * EPP >= ne_c) break;                // (uniform: the rest of the chunk is padding)
        }
        __syncthreads();
        if (TIPK_DBG(A.dbg & 32)) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = t + 1024 * k;
            if (item < n_mine * q) {
                const int r = item / q, c = item - r * q;
                int lo = rptr[r] - c0, hi = rptr[r + 1] - c0;
                lo = lo < 0 ? 0 : lo;
                hi = hi > ne_c ? ne_c : hi;
                float sacc = racc[k];
                int e = lo;
                for (; e + 8 <= hi; e += 8) {                 // eight reads in flight, added in edge order
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = prod[(e + u) * q + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sacc += v[u];
                }
                for (; e < hi; ++e) sacc += prod[e * q + c];
                racc[k] = sacc;
            }
        }
        __syncthreads();                                      // (the chunk has been read)
        if (TIPK_DBG(A.dbg & 64)) return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int item = t + 1024 * k;
        if (item < n_mine * q) tq[item] = racc[k];
    }
    __syncthreads();
    // g_h = tq W_h^T: thread (row, column < p)
    for (int item = t; item < n_mine * p; item += 1024) {
        const int r = item / p, c = item - r * p;
        float v = 0.f;
        for (int j = 0; j < q; ++j) v = fmaf(tq[r * q + j], wlt[j * p + c], v);
        gh[item] = v;
    }
    __syncthreads();
    // gw = (g_h W2) * row_scale: thread (row, column < c1)
    for (int item = t; item < n_mine * c1; item += 1024) {
        const int r = item / c1, n = item - r * c1;
        const int sr = s_first + r;
        float v = 0.f;
        for (int k = 0; k < p; ++k) v = fmaf(gh[r * p + k], w2l[k * c1 + n], v);
        A.gw[(int64_t)sr * A.ld_gw + n] = A.row_scale ? v * A.row_scale[sr] : v;
    }
    // ---------------------------------------------------------------- partial d W2 [c1 x p] = agg^T g_h and d b2 [p] over the rows
    if (TIPK_DBG(A.dbg & 4)) return;
    float* dw = A.dw2 + (int64_t)b * c1 * p;
    for (int o = t; o < c1 * p + p; o += 1024) {
        float sacc = 0.f;
        if (o < c1 * p) {
            const int k = o / p, c = o - k * p;
            for (int r = 0; r < n_mine; ++r) sacc = fmaf(aggl[r * c1 + k], gh[r * p + c], sacc);
            dw[o] = sacc;
        } else {
            const int c = o - c1 * p;
            for (int r = 0; r < n_mine; ++r) sacc += gh[r * p + c];
            A.db2[(int64_t)b * p + c] = sacc;
        }
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
                                        const int32_t* order, int64_t n_wg, const float* w, int p, int q, int64_t rows, int ne, int cat,
                                        float* out, int64_t ld_out, float* mean, tipk_stream_t stream) {
    if (rows < 0 || ne < 0 || !tipk_drug_mix_gather_supported(p, q) || (!cat && q != ne)) return TIPK_EINVAL;
    if (rows == 0) return TIPK_OK;
    if (!xd || !h || !ptr || !src || !scale || !w || !out || !mean || !wg_desc || n_wg <= 0 || rows > 0x7fffffffLL ||
        n_wg > 0x7fffffffLL || (reinterpret_cast<uintptr_t>(wg_desc) & 7))
        return TIPK_EINVAL;
    DgFwdArgs a;
    a.xd = xd; a.ld_xd = ld_xd; a.d_norm = d_norm; a.h = h; a.ld_h = ld_h; a.ptr = ptr; a.src = src; a.scale = scale;
    a.wg = reinterpret_cast<const int2*>(wg_desc); a.order = order;
    a.w = w; a.p = p; a.q = q; a.ne = ne; a.cat = cat; a.out = out; a.ld_out = ld_out; a.mean = mean; a.rows = (int)rows;
    a.basis = nullptr; a.root = nullptr; a.n_bases = 0; a.d_out = 0; a.xb = nullptr; a.xroot = nullptr;
    const dim3 grid((unsigned)n_wg);
    hipStream_t st = (hipStream_t)stream;
    if (p <= 8) hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<8, false>), grid, dim3(1024), 0, st, a);
    else if (p <= 16) hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<16, false>), grid, dim3(1024), 0, st, a);
    else if (p <= 32) hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<32, false>), grid, dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<64, false>), grid, dim3(1024), 0, st, a);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_drug_mix_gather_xb_supported(int p, int q, int ne, int cat, int n_bases, int d_out) {
    if (!tipk_drug_mix_gather_supported(p, q) || ne < 0 || (!cat && q != ne)) return 0;
    const int cols = cat ? ne + q : ne;
    if (cols <= 0 || cols > DG_XMAX || (cols & 3)) return 0;
    if (n_bases < 1 || n_bases > 4096 || (d_out != 16 && d_out != 32)) return 0;
    return 1;
}

extern "C" int tipk_drug_mix_gather_xb_fwd(const float* xd, int64_t ld_xd, const float* d_norm, const float* h, int64_t ld_h,
                                           const int32_t* ptr, const int32_t* src, const float* scale, const int32_t* wg_desc,
                                           const int32_t* order, int64_t n_wg, const float* w, int p, int q, int64_t rows, int ne,
                                           int cat, float* out, int64_t ld_out, float* mean, const float* basis, const float* root,
                                           int n_bases, int d_out, float* xb, float* xroot, tipk_stream_t stream) {
    if (rows < 0 || ne < 0) return TIPK_EINVAL;
    if (!tipk_drug_mix_gather_xb_supported(p, q, ne, cat, n_bases, d_out)) return TIPK_EUNSUPPORTED;
    if (rows == 0) return TIPK_OK;
    if (!xd || !h || !ptr || !src || !scale || !w || !out || !mean || !wg_desc || !basis || !root || !xb || !xroot || n_wg <= 0 ||
        rows > 0x7fffffffLL || n_wg > 0x7fffffffLL || (reinterpret_cast<uintptr_t>(wg_desc) & 7))
        return TIPK_EINVAL;
    DgFwdArgs a;
    a.xd = xd; a.ld_xd = ld_xd; a.d_norm = d_norm; a.h = h; a.ld_h = ld_h; a.ptr = ptr; a.src = src; a.scale = scale;
    a.wg = reinterpret_cast<const int2*>(wg_desc); a.order = order;
    a.w = w; a.p = p; a.q = q; a.ne = ne; a.cat = cat; a.out = out; a.ld_out = ld_out; a.mean = mean; a.rows = (int)rows;
    a.basis = basis; a.root = root; a.n_bases = n_bases; a.d_out = d_out; a.xb = xb; a.xroot = xroot;
    if (n_wg * DG_CP > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    const dim3 grid((unsigned)(n_wg * DG_CP));
    hipStream_t st = (hipStream_t)stream;
    if (p <= 8) hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<8, true>), grid, dim3(1024), 0, st, a);
    else if (p <= 16) hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<16, true>), grid, dim3(1024), 0, st, a);
    else if (p <= 32) hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<32, true>), grid, dim3(1024), 0, st, a);
    else hipLaunchKernelGGL((drug_mix_gather_fwd_kernel<64, true>), grid, dim3(1024), 0, st, a);
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

extern "C" int tipk_pd_stage_bwd_supported(int p, int q, int64_t rows, int c1) {
    if (!tipk_drug_mix_gather_supported(p, q) || rows <= 0) return 0;
    if (c1 <= 0 || c1 > 64 || (int64_t)p * c1 > 4096) return 0;                 // (a workgroup's 64 rows of agg and of g_h in LDS)
    return 1;
}

extern "C" int tipk_pd_stage_bwd_wh_slabs(void) { return PD_WH_WGS; }

extern "C" int tipk_pd_stage_bwd_limits(int* max_rows, int* max_edges) {
    if (max_rows) *max_rows = 16 * PD_ROWS;
    if (max_edges) *max_edges = PD_CHUNK_E;
    return TIPK_OK;
}

extern "C" int tipk_pd_stage_bwd(const float* g, int64_t ld_g, const float* d_norm, const float* mean, const float* w, int p, int q,
                                 int64_t rows, int ne, int cat, float* g_xd, int64_t ld_gxd, float* g_w_slabs,
                                 const int32_t* tptr, const int32_t* tdst, const float* tw, int64_t n_src,
                                 const int32_t* wg_rows, int64_t n_row_wg,
                                 const float* agg, int64_t ld_agg, int c1, const float* w2, int64_t w2_sk, int64_t w2_sn,
                                 const float* row_scale, float* gw, int64_t ld_gw, float* dw2_slabs, float* db2_slabs,
                                 tipk_stream_t stream) {
    if (rows < 0 || ne < 0 || n_src < 0) return TIPK_EINVAL;
    if (!tipk_pd_stage_bwd_supported(p, q, rows, c1) || (!cat && q != ne)) return TIPK_EUNSUPPORTED;
    if (!g || !mean || !w || !g_w_slabs || !tptr || !tdst || !tw || !agg || !w2 || !gw || !dw2_slabs || !db2_slabs || n_src == 0 ||
        !wg_rows || n_row_wg <= 0 || n_row_wg > 0x7fffffLL ||
        rows * (int64_t)(ne + p + q) > 0x7fffffffLL || n_src > 0x7fffffffLL)
        return TIPK_EINVAL;
    PdBwdArgs A;
    DgBwdArgs& a = A.m;
    a.g = g; a.ld_g = ld_g; a.d_norm = d_norm; a.mean = mean; a.w = w; a.p = p; a.q = q; a.ne = ne; a.cat = cat;
    a.g_xd = g_xd; a.ld_gxd = ld_gxd; a.g_mean = nullptr; a.g_w = g_w_slabs;
    a.rows = (int)rows;
    a.tp = p * q / 4;
    a.kg = 1024 / a.tp;
    if (a.kg > 4096 / (p * q)) a.kg = 4096 / (p * q);
    if (a.kg > 16) a.kg = 16;
    if (a.kg < 1) a.kg = 1;
    a.tile = DG_STAGE / (p + q);
    A.rows_per_wg = 16 * PD_ROWS;                                               // 64 rows at most: 64 p <= 4096 floats of g_h
    A.wg_rows = wg_rows;
    const int64_t n_wg = n_row_wg;
    a.n_wg = (int)n_wg;
    A.tptr = tptr; A.tdst = tdst; A.tw = tw; A.n_src = (int)n_src;
    A.agg = agg; A.ld_agg = ld_agg; A.c1 = c1; A.w2 = w2; A.w2_sk = w2_sk; A.w2_sn = w2_sn; A.row_scale = row_scale;
    A.gw = gw; A.ld_gw = ld_gw; A.dw2 = dw2_slabs; A.db2 = db2_slabs;
    A.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DM_DEBUG));
    const dim3 grid((unsigned)(PD_WH_WGS + n_wg));
    hipStream_t st = (hipStream_t)stream;
    if (q <= 8) hipLaunchKernelGGL(pd_stage_bwd_kernel<8>, grid, dim3(1024), 0, st, A);
    else if (q <= 16) hipLaunchKernelGGL(pd_stage_bwd_kernel<16>, grid, dim3(1024), 0, st, A);
    else if (q <= 32) hipLaunchKernelGGL(pd_stage_bwd_kernel<32>, grid, dim3(1024), 0, st, A);
    else hipLaunchKernelGGL(pd_stage_bwd_kernel<64>, grid, dim3(1024), 0, st, A);
    TIPK_RETURN_LAUNCH();
}
