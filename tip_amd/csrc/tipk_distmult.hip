// DistMult decoder (include/tipk.h section 4): score = sigma(sum_k z[u,k] z[v,k] w[r,k]) and its
// gradients, plus the fused TIP training objective.  z (N x k) and w (R x k) are a few tens of KB:
// they stay L1/L2-resident, so the compulsory HBM traffic is the triple list itself
// (2 indices + relation id per triple) plus one score.
//
// Gradients: d z is scattered over at most N rows that every workgroup hits -> accumulated in a
// per-workgroup LDS image with ds_add_f32 and flushed once; d w[r] is reduced across the wave with
// shuffles whenever the wave's triples share one relation (triples arrive grouped by relation,
// src/utils.py:57-63), so global float atomics are per wave, not per triple (Guideline 12).
#include <type_traits>
#include "tipk_common.h"

namespace {

constexpr float TIP_EPS = 1e-13f;       // src/layers.py:15

template <typename T>
__device__ __forceinline__ int64_t ld_idx(const void* p, int64_t i) { return (int64_t) reinterpret_cast<const T*>(p)[i]; }

template <int V>
__device__ __forceinline__ float dot3(const float* a, const float* b, const float* c, int k) {
    float s = 0.f;
    if (V == 4) {
        for (int j = 0; j < k; j += 4) {
            const float4 x = tipk_ld4(a + j), y = tipk_ld4(b + j), w = tipk_ld4(c + j);
            s = fmaf(x.x * y.x, w.x, s); s = fmaf(x.y * y.y, w.y, s);
            s = fmaf(x.z * y.z, w.z, s); s = fmaf(x.w * y.w, w.w, s);
        }
    } else {
        for (int j = 0; j < k; ++j) s = fmaf(a[j] * b[j], c[j], s);
    }
    return s;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

template <typename IT, typename ET, int V>
__global__ __launch_bounds__(256) void distmult_fwd_kernel(const float* __restrict__ z, int k,
                                                           const float* __restrict__ w, const void* iu,
                                                           const void* iv, const void* et, int64_t n, int sig,
                                                           float* __restrict__ score) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t u = ld_idx<IT>(iu, e), v = ld_idx<IT>(iv, e), r = ld_idx<ET>(et, e);
    const float s = dot3<V>(z + u * k, z + v * k, w + r * k, k);
    score[e] = sig ? sigmoidf(s) : s;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// Adds coef * (zb (.) wr) into row `ua` of the d z image (LDS or global).
template <bool LDS>
__device__ __forceinline__ void add_row(float* gz, int64_t row, int k, int j, float a) {
    if (LDS) atomicAdd(gz + row * k + j, a);            // ds_add_f32
    else atomicAdd(gz + row * k + j, a);                // global_atomic_add_f32
}

// MODE 0: upstream gradient per triple given (backward of `distmult_fwd`).
// MODE 1: fused objective over (pos, neg) pairs; writes the loss and the gradients in one pass.
template <typename IT, typename ET, int V, int MODE, bool USE_LDS>
__global__ __launch_bounds__(256) void distmult_grad_kernel(
    const float* __restrict__ g_score, const float* __restrict__ score, const float* __restrict__ z,
    int64_t n_nodes, int k, const float* __restrict__ w, const void* iu, const void* iv, const void* ju,
    const void* jv, const void* et, int64_t n, int64_t per_block, int sig, float* __restrict__ loss_out,
    float* __restrict__ g_z, float* __restrict__ g_w) {
    extern __shared__ float lds[];
    __shared__ float red[4];
    const int t = threadIdx.x;
    const int64_t nk = n_nodes * k;
    float* gz = USE_LDS ? lds : g_z;
    const bool want_grad = g_z != nullptr;
    if (USE_LDS && want_grad) {
        for (int64_t i = t; i < nk; i += blockDim.x) lds[i] = 0.f;
        __syncthreads();
    }
    const int64_t lo = (int64_t)blockIdx.x * per_block;
    const int64_t hi = lo + per_block < n ? lo + per_block : n;
    const float inv_n = 1.f / (float)n;
    float loss = 0.f;

    for (int64_t base = lo; base < hi; base += blockDim.x) {
        const int64_t e = base + t;
        const bool valid = e < hi;
        int64_t r = -1, u[2] = {0, 0}, v[2] = {0, 0};
        float coef[2] = {0.f, 0.f};
        constexpr int NT = MODE == 1 ? 2 : 1;
        if (valid) {
            r = ld_idx<ET>(et, e);
            u[0] = ld_idx<IT>(iu, e); v[0] = ld_idx<IT>(iv, e);
            if (MODE == 1) {
                u[1] = ld_idx<IT>(ju, e); v[1] = ld_idx<IT>(jv, e);
                const float sp = sigmoidf(dot3<V>(z + u[0] * k, z + v[0] * k, w + r * k, k));
                const float sn = sigmoidf(dot3<V>(z + u[1] * k, z + v[1] * k, w + r * k, k));
                loss -= logf(sp + TIP_EPS) + logf(1.f - sn + TIP_EPS);
                coef[0] = -inv_n * sp * (1.f - sp) / (sp + TIP_EPS);
                coef[1] = inv_n * sn * (1.f - sn) / (1.f - sn + TIP_EPS);
            } else {
                coef[0] = g_score[e];
                if (sig) { const float s = score[e]; coef[0] *= s * (1.f - s); }
            }
        }
        if (!want_grad) continue;
        // is the relation uniform over the wave's valid lanes?
        const unsigned long long vm = __ballot(valid);
        if (vm == 0ULL) continue;
        const int first = __ffsll((long long)vm) - 1;
        const int64_t r0 = __shfl(r, first);
        const bool uniform = __all(!valid || r == r0);
        for (int j = 0; j < k; ++j) {
            float gw = 0.f;
            if (valid) {
                const float wr = w[r * k + j];
#pragma unroll
                for (int q = 0; q < NT; ++q) {
                    const float zu = z[u[q] * k + j], zv = z[v[q] * k + j];
                    add_row<USE_LDS>(gz, u[q], k, j, coef[q] * zv * wr);
                    add_row<USE_LDS>(gz, v[q], k, j, coef[q] * zu * wr);
                    gw = fmaf(coef[q], zu * zv, gw);
                }
            }
            if (uniform) {
                gw = wave_sum(gw);
                if (tipk_lane() == 0) atomicAdd(g_w + r0 * k + j, gw);
            } else if (valid) {
                atomicAdd(g_w + r * k + j, gw);
            }
        }
    }

    if (MODE == 1) {
        loss = wave_sum(loss);
        if (tipk_lane() == 0) red[t >> 6] = loss;
        __syncthreads();
        if (t == 0) atomicAdd(loss_out, (red[0] + red[1] + red[2] + red[3]) * inv_n);
    }
    if (USE_LDS && want_grad) {
        __syncthreads();
        for (int64_t i = t; i < nk; i += blockDim.x) {
            const float a = lds[i];
            if (a != 0.f) atomicAdd(g_z + i, a);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Task kernel: the fast path when triples are grouped by relation (TIP's layout).
//
// A *task* = (relation, begin, end) with at most a few thousand positions of ONE relation, so the
// relation row w[r] lives in registers and d w[r] is reduced inside the workgroup (one set of k
// global atomics per task).  z is staged once per workgroup into LDS (row stride k+4 floats:
// 16-byte aligned float4 reads that spread over the banks).  KL = k/4 lanes share one position:
// each reads one float4 of every row, the dot product is finished with KL-wide shuffles.
//
// d z is scattered over N rows that every position hits.  Measured on gfx950
// (tools/microbench/lds_atomics.hip): ds_add_f32 sustains 0.33 lane-ops/clk/CU, ds_add_u64 5.0 --
// 15x more.  The d z image is therefore a 64-bit FIXED-POINT accumulator in LDS: every fp32
// contribution is scaled by a power of two chosen from a bound on the sum of |contributions|
// (exact conversion, 60 bits of headroom), integer-added (associative: the result does not depend on
// scheduling), and converted back once per workgroup.
// Persistent grid (one 1024-thread workgroup per CU), tasks dealt round-robin, largest first.
__device__ __forceinline__ float block_max_1024(float v, float* red, int t) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    float m = red[0];
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    __syncthreads();
    return m;
}

__device__ __forceinline__ float block_sum_1024(float v, float* red, int t) {
    v = wave_sum(v);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    float m = 0.f;
    for (int i = 0; i < 16; ++i) m += red[i];
    __syncthreads();
    return m;
}

__device__ __forceinline__ void fx_add(unsigned long long* p, float c, double scale) {
    atomicAdd(p, (unsigned long long)(long long)((double)c * scale));      // ds_add_u64
}

// The fused objective's fixed point (MODE 1 with a workspace): every term added to d z is bounded INDIVIDUALLY --
// |q| <= 2 / n (weight-2 positives), |z w| <= zmax wmax -- so the scale is chosen per TERM: scaled terms stay below 2^30
// in magnitude, one fp32 multiply by a power of two, a round to nearest and a 32-bit convert give the integer (the eight
// double-precision instructions `(long long)(double)` expands to -- v_cvt_f64_f32, v_mul_f64, v_trunc / ldexp / floor /
// fmac _f64, two converts back -- were a third of the kernel: tools/bench_decoder.py), and the 64-bit accumulators
// hold up to 2^33 such terms.  Resolution: 2^-30 of the largest possible term, rounding unbiased.
__device__ __forceinline__ void fx_add_term(unsigned long long* p, float c, float scale_f) {
    const int v = (int)rintf(c * scale_f);                                 // |c * scale| < 2^30 by construction
    atomicAdd(p, (unsigned long long)(long long)v);                        // ds_add_u64
}

// sigma, log and the quotient of the objective on the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32,
// ~1 ulp each) instead of the IEEE-accurate library sequences: the objective is compared at 2e-5, its gradient at 2e-6
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

constexpr int TASK_MAX = 2048;          // positions per task (ids staged in LDS: 4 x 2048 x 2 B)

template <typename IT, int MODE>
__global__ __launch_bounds__(1024) void distmult_task_kernel(
    const float* __restrict__ z, int n_nodes, int k, const float* __restrict__ w, int n_rel,
    const int32_t* __restrict__ tasks, int n_tasks, const IT* __restrict__ pu, const IT* __restrict__ pv,
    const IT* __restrict__ nu, const IT* __restrict__ nv, const float* __restrict__ g_score, int sig,
    int64_t n_total, float* __restrict__ loss_out, float* __restrict__ g_z, float* __restrict__ g_w,
    unsigned long long* __restrict__ ws, int dbg) {
    // ws != NULL (fused objective only): the cross-workgroup sums of loss, d z and d w go through 64-bit FIXED-POINT
    // integer atomics into `ws` (exact, order-independent) and det_finalize_kernel converts them: the training
    // objective and its gradients are then bitwise reproducible; with float atomics they differ at ~1e-7.
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
    const int t = threadIdx.x;
    const int ld = k + 4;                                     // z image: float4 reads, 16-byte aligned rows
    const int lg = k + 1;                                     // d z image: odd stride = all banks
    unsigned long long* gzl = lds64;                          // [n_nodes][k+1] fixed point
    float* zl = (float*)(gzl + (((int64_t)n_nodes * lg + 1) & ~1LL));   // [n_nodes][k+4], 16-BYTE ALIGNED: rows are read as b128 (an
                                                              // image that starts on an odd 8-byte word costs SQ_LDS_UNALIGNED_STALL on every read)
    float* red = zl + (((int64_t)n_nodes * ld + 3) & ~3LL);   // [16 waves][k] + [16]
    uint16_t* ixl = reinterpret_cast<uint16_t*>(red + 16 * k + 16);   // [4][TASK_MAX] ids of the current task
    const bool want_grad = g_z != nullptr;
    float zmax = 0.f;
    for (int i = t; i < n_nodes * k; i += 1024) {
        const int r = i / k, c = i - r * k;
        const float v = z[i];
        zl[r * ld + c] = v;
        zmax = fmaxf(zmax, fabsf(v));
    }
    for (int i = t; i < n_nodes * lg; i += 1024) gzl[i] = 0ull;
    double scale = 1.0;
    if (want_grad) {
        float wmax = 0.f;
        for (int i = t; i < n_rel * k; i += 1024) wmax = fmaxf(wmax, fabsf(w[i]));
        zmax = block_max_1024(zmax, red, t);
        wmax = block_max_1024(wmax, red, t);
        // bound on sum |contribution| into any element: every position adds <= 4 |q| zmax wmax with
        // sum |q| <= 1 (fused objective: |q| <= 1/n each) or <= sum |g| over this workgroup's positions
        float bound = 4.f * zmax * wmax;
        if (MODE == 0) {
            float sg = 0.f;
            for (int task = blockIdx.x; task < n_tasks; task += gridDim.x)
                for (int p = tasks[4 * task + 1] + t; p < tasks[4 * task + 2]; p += 1024) sg += fabsf(g_score[p]);
            bound *= block_sum_1024(sg, red, t);
        }
        int ex = 60;
        if (bound > 0.f && bound < 3.0e38f) ex = 60 - (ilogbf(bound) + 1);
        ex = ex > 180 ? 180 : ex;
        if (MODE == 1 && ws) {                                  // per-TERM scale (fx_add_term): a term is <= bound / n
            const float tb = bound / (float)n_total;
            ex = 100;
            if (tb > 0.f && tb < 3.0e38f) ex = 30 - (ilogbf(tb) + 1);
            ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);     // (the scale also has to fit a float)
        }
        scale = ldexp(1.0, ex);
    }
    // fixed-point scales of the deterministic path: identical in every workgroup (they depend on z and w only)
    double scale_w = 1.0;
    const double scale_l = 1125899906842624.0;                // 2^50: the objective is < 2^7
    if (ws && MODE == 1) {
        const float bw = 4.f * zmax * zmax;
        int ex = 60;
        if (bw > 0.f && bw < 3.0e38f) ex = 60 - (ilogbf(bw) + 1);
        scale_w = ldexp(1.0, ex > 180 ? 180 : ex);
        if (blockIdx.x == 0 && t == 0) {                      // the finalize kernel divides by them
            const int64_t base = (int64_t)n_nodes * k + (int64_t)n_rel * k + 1;
            reinterpret_cast<double*>(ws)[base] = scale;
            reinterpret_cast<double*>(ws)[base + 1] = scale_w;
            reinterpret_cast<double*>(ws)[base + 2] = scale_l;
        }
    }
    __syncthreads();
    const bool per_term = MODE == 1 && ws != nullptr;
    const float scale_f = (float)scale;
    float dbg_sink = 0.f;
    auto fxa = [&](unsigned long long* p, float c) {
        if (TIPK_DBG(dbg & 1)) { dbg_sink += c; return; }                 // debug builds: no conversion, no atomic
        if (per_term) fx_add_term(p, c, scale_f); else fx_add(p, c, scale);
    };
    const int KL = k >> 2;                                    // lanes per position (power of two <= 16)
    const int sub = t & (KL - 1);
    const int slot = t / KL;
    const int n_slots = 1024 / KL;
    const int c0 = sub * 4;
    const float inv_n = 1.f / (float)n_total;
    float loss = 0.f;

    for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int rel = tasks[4 * task], tb = tasks[4 * task + 1], te = tasks[4 * task + 2];
        // weight of the task's POSITIVE triples in the fused objective: 1, or -- when the caller has verified
        // that every relation lists each pair in both directions (the data contract of the path:
        // [u<v half | mirrored half], src/utils.py:35-65) -- 2 for the first half and 0 for the mirrored
        // half, whose scores and gradients are identical: a quarter of the LDS atomics and of the
        // transcendentals disappear.  The negatives of every position are always evaluated.
        const int pos_w = MODE == 1 ? tasks[4 * task + 3] : 1;
        const float4 wr = tipk_ld4(w + (int64_t)rel * k + c0);
        float4 gw = make_float4(0.f, 0.f, 0.f, 0.f);
        // The task's ids are staged into LDS as 16-bit values with ONE batch of coalesced loads per
        // thread (a per-iteration prefetch left the loop bound by one HBM latency per 256 positions).
        {
            const int cnt = te - tb;
            int64_t g[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = h * 1024 + t;
                const int64_t q = (int64_t)tb + (i < cnt ? i : cnt - 1);
                g[h * 4 + 0] = (int64_t)pu[q];
                g[h * 4 + 1] = (int64_t)pv[q];
                g[h * 4 + 2] = MODE == 1 ? (int64_t)nu[q] : 0;
                g[h * 4 + 3] = MODE == 1 ? (int64_t)nv[q] : 0;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = h * 1024 + t;
                if (i < cnt) {
                    ixl[0 * TASK_MAX + i] = (uint16_t)g[h * 4 + 0];
                    ixl[1 * TASK_MAX + i] = (uint16_t)g[h * 4 + 1];
                    ixl[2 * TASK_MAX + i] = (uint16_t)g[h * 4 + 2];
                    ixl[3 * TASK_MAX + i] = (uint16_t)g[h * 4 + 3];
                }
            }
        }
        __syncthreads();
        for (int64_t base = tb; base < te; base += n_slots) {
            const int64_t pos = base + slot;
            const bool valid = pos < te;
            const int li = valid ? (int)(pos - tb) : 0;
            const int cu0 = ixl[li], cv0 = ixl[TASK_MAX + li];
            const int cu1 = MODE == 1 ? ixl[2 * TASK_MAX + li] : 0, cv1 = MODE == 1 ? ixl[3 * TASK_MAX + li] : 0;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), b0 = a0, a1 = a0, b1 = a0;
            float d0 = 0.f, d1 = 0.f;
            if (valid) {
                if (pos_w != 0) {
                    a0 = tipk_ld4(zl + cu0 * ld + c0);
                    b0 = tipk_ld4(zl + cv0 * ld + c0);
                    d0 = a0.x * b0.x * wr.x + a0.y * b0.y * wr.y + a0.z * b0.z * wr.z + a0.w * b0.w * wr.w;
                }
                if (MODE == 1) {
                    a1 = tipk_ld4(zl + cu1 * ld + c0);
                    b1 = tipk_ld4(zl + cv1 * ld + c0);
                    d1 = a1.x * b1.x * wr.x + a1.y * b1.y * wr.y + a1.z * b1.z * wr.z + a1.w * b1.w * wr.w;
                }
            }
            for (int o = 1; o < KL; o <<= 1) {
                d0 += __shfl_xor(d0, o);
                if (MODE == 1) d1 += __shfl_xor(d1, o);
            }
            float q0 = 0.f, q1 = 0.f;
            if (MODE == 1) {
                // the positive's and the negative's sigmoid / log / divide are evaluated by DIFFERENT
                // lanes of the position (sub 0 and sub 1) and exchanged with two shuffles: the SIMD
                // executes one transcendental sequence per step instead of two
                const bool neg_lane = KL > 1 && sub == 1;
                const float x = neg_lane ? d1 : d0;
                const float sg = fast_sigmoid(x);
                const float val = neg_lane ? 1.f - sg : sg;
                const float lg = __logf(val + TIP_EPS);
                const float qq = inv_n * sg * (1.f - sg) * __builtin_amdgcn_rcpf(val + TIP_EPS);
                const float pw = (float)pos_w;
                if (KL > 1) {
                    if (valid && sub < 2) loss -= neg_lane ? lg : pw * lg;
                    q0 = -pw * __shfl(qq, 0, KL);
                    q1 = __shfl(qq, 1, KL);
                } else {                                   // k = 4: one lane per position does both
                    const float sn = fast_sigmoid(d1);
                    if (valid) loss -= pw * lg + __logf(1.f - sn + TIP_EPS);
                    q0 = -pw * qq;
                    q1 = inv_n * sn * (1.f - sn) * __builtin_amdgcn_rcpf(1.f - sn + TIP_EPS);
                }
                if (!valid) { q0 = 0.f; q1 = 0.f; }
            } else if (valid) {
                q0 = g_score[pos];
                if (sig) { const float sg = sigmoidf(d0); q0 *= sg * (1.f - sg); }
            }
            if (want_grad && valid) {
                if (pos_w != 0) {
                    unsigned long long* gu = gzl + cu0 * lg + c0;
                    unsigned long long* gv = gzl + cv0 * lg + c0;
                    fxa(gu + 0, q0 * b0.x * wr.x); fxa(gu + 1, q0 * b0.y * wr.y);
                    fxa(gu + 2, q0 * b0.z * wr.z); fxa(gu + 3, q0 * b0.w * wr.w);
                    fxa(gv + 0, q0 * a0.x * wr.x); fxa(gv + 1, q0 * a0.y * wr.y);
                    fxa(gv + 2, q0 * a0.z * wr.z); fxa(gv + 3, q0 * a0.w * wr.w);
                    gw.x = fmaf(q0, a0.x * b0.x, gw.x); gw.y = fmaf(q0, a0.y * b0.y, gw.y);
                    gw.z = fmaf(q0, a0.z * b0.z, gw.z); gw.w = fmaf(q0, a0.w * b0.w, gw.w);
                }
                if (MODE == 1) {
                    unsigned long long* hu = gzl + cu1 * lg + c0;
                    unsigned long long* hv = gzl + cv1 * lg + c0;
                    fxa(hu + 0, q1 * b1.x * wr.x); fxa(hu + 1, q1 * b1.y * wr.y);
                    fxa(hu + 2, q1 * b1.z * wr.z); fxa(hu + 3, q1 * b1.w * wr.w);
                    fxa(hv + 0, q1 * a1.x * wr.x); fxa(hv + 1, q1 * a1.y * wr.y);
                    fxa(hv + 2, q1 * a1.z * wr.z); fxa(hv + 3, q1 * a1.w * wr.w);
                    gw.x = fmaf(q1, a1.x * b1.x, gw.x); gw.y = fmaf(q1, a1.y * b1.y, gw.y);
                    gw.z = fmaf(q1, a1.z * b1.z, gw.z); gw.w = fmaf(q1, a1.w * b1.w, gw.w);
                }
            }
        }
        if (!want_grad) __syncthreads();                       // ids of this task are no longer needed
        if (want_grad) {                                       // d w[rel]: wave shuffle, then 16 waves via LDS
            for (int o = KL; o < TIPK_WAVE; o <<= 1) {
                gw.x += __shfl_xor(gw.x, o); gw.y += __shfl_xor(gw.y, o);
                gw.z += __shfl_xor(gw.z, o); gw.w += __shfl_xor(gw.w, o);
            }
            if (tipk_lane() < KL) tipk_st4(red + (t >> 6) * k + c0, gw);
            __syncthreads();
            if (t < k) {
                float tot = 0.f;
                for (int wv = 0; wv < 16; ++wv) tot += red[wv * k + t];
                if (ws && MODE == 1)
                    atomicAdd(ws + (int64_t)n_nodes * k + (int64_t)rel * k + t, (unsigned long long)(long long)((double)tot * scale_w));
                else
                    atomicAdd(g_w + (int64_t)rel * k + t, tot);
            }
            __syncthreads();
        }
    }
    if (TIPK_DBG(dbg) && dbg_sink == 12345.f) loss += 1.f;
    if (MODE == 1) {
        loss = wave_sum(loss);
        if (tipk_lane() == 0) red[16 * k + (t >> 6)] = loss;
        __syncthreads();
        if (t == 0) {
            float tot = 0.f;
            for (int wv = 0; wv < 16; ++wv) tot += red[16 * k + wv];
            if (ws) atomicAdd(ws + (int64_t)n_nodes * k + (int64_t)n_rel * k, (unsigned long long)(long long)((double)(tot * inv_n) * scale_l));
            else atomicAdd(loss_out, tot * inv_n);
        }
    }
    if (want_grad) {
        __syncthreads();
        const double inv_scale = 1.0 / scale;
        for (int i = t; i < n_nodes * k; i += 1024) {
            const int r = i / k, c = i - r * k;
            const long long a = (long long)gzl[r * lg + c];
            if (TIPK_DBG(dbg & 4)) continue;                           // debug builds: no flush of the d z image
            if (ws && MODE == 1) { if (a != 0) atomicAdd(ws + i, (unsigned long long)a); }
            else if (a != 0) atomicAdd(g_z + i, (float)((double)a * inv_scale));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The fused objective, one LANE per position (deterministic path: tasks + workspace; k in {4, 8, 16}).
//
// distmult_task_kernel gives a position to k / 4 lanes: every lane of the group executes the position's
// transcendentals, the dot product ends in shuffles, and the task's ids are staged through LDS behind a barrier -- a
// load phase nothing overlaps.  Measured at BioSNAP size (tools/bench_decoder.py, rocprofv3): 280 us for the objective
// alone (no gradients), against 88 us per 8.3 M scores for the plain one-thread-per-triple score kernel.  Here a lane
// owns a position outright for the EVALUATION: its four ids come straight from global memory (coalesced across the
// wave, the next task's ids requested before the current one is evaluated), the rows of z from the LDS image as b128
// reads, the dot products, sigma / log / quotient once.  The gradient terms are then scattered with k LANES PER ROW
// (phase 2 below: ids and coefficients travel by wave shuffles), so that one LDS atomic instruction touches 64 / k
// rows of the fixed-point d z image as contiguous pieces; d w[r] is summed per task: a column per lane -> the
// wave's groups by shuffles -> 16 waves through LDS -> one fixed-point add per column.
// idx_bytes = 2 of tipk_distmult_loss: a pair of 16-bit node ids in one 32-bit word (u | v << 16), one array per triple
// list -- the positives of the path are static and narrowed once, the negatives come from the sampler in this form:
// 66 MB of ids per BioSNAP step instead of the 266 MB of four int64 arrays
struct PackedPair { uint32_t w; };

template <int I>
__device__ __forceinline__ int pr_row_bcast(int x) {      // lane I of every 16-lane DPP row, broadcast to its row
    // (bound_ctrl with every row and bank enabled: the combiner folds this move into the VOP2 instruction that uses it --
    // v_add_u32_dpp / v_mul_f32_dpp; the plain mov_dpp builtin stays a separate v_mov_b32_dpp: 5 of the 20 VALU
    // instructions of a scatter step in the round-4 kernel)
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + (I & 15), 0xf, 0xf, true);
}
// float -> int32, round to nearest (ties up: floor(x + 0.5)) in ONE instruction; rintf + the conversion are two
__device__ __forceinline__ int pr_round_i32(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
template <int I>
__device__ __forceinline__ void pr_fmac_row_bcast(float& acc, float x, float y) {   // acc += (lane I of the row's x) * y
    // (the combiner folds a broadcast into v_add / v_mul but not into the tied-accumulator v_fmac)
    // DPP reads `x` from OTHER lanes: a VALU write of x needs 2 wait states before it, and the hazard recognizer does not look
    // inside inline assembly.  The FIRST instruction of every chain (I % 16 == 0: the chains below run I = 0 .. K - 1 over one
    // unchanged x) carries the s_nop itself, so a register copy the compiler may place in front of the chain is covered too;
    // inside a chain x is read-only.  The row-of-16 lane mapping (I & 15) is why the kernel asserts K <= 16.
    if constexpr ((I & 15) == 0)
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y), "n"(I & 15));
    else
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y), "n"(I & 15));
}
typedef float pr_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float pr_lds_f32_t;
typedef __attribute__((address_space(3))) unsigned long long pr_lds_u64_t;
template <int N, int I = 0, typename F>
__device__ __forceinline__ void pr_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pr_static_for<N, I + 1>(f);
    }
}

template <typename IT, int K>
__global__ __launch_bounds__(1024) void distmult_objective_kernel(
    const float* __restrict__ z, int n_nodes, const float* __restrict__ w, int n_rel,
    const int32_t* __restrict__ tasks, int n_tasks, const IT* __restrict__ pu, const IT* __restrict__ pv,
    const IT* __restrict__ nu, const IT* __restrict__ nv, int64_t n_total, int want_grad,
    unsigned long long* __restrict__ ws, int dbg) {
    static_assert(K <= 16 && (K & (K - 1)) == 0, "the w row travels as the DPP broadcast of a 16-lane row");
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
    const int t = threadIdx.x;
    constexpr int ld = K + 4, lg = K + 1;
    unsigned long long* gzl = lds64;                                      // [n_nodes][K + 1] fixed point
    float* zl = (float*)(gzl + (((int64_t)n_nodes * lg + 1) & ~1LL));     // [n_nodes][K + 4], 16-byte aligned (b128 row reads)
    float* red = zl + (((int64_t)n_nodes * ld + 3) & ~3LL);               // [16 waves][K] + [16]
    float zmax = 0.f, wmax = 0.f;
    for (int i = t; i < n_nodes * K; i += 1024) {
        const int r = i / K, c = i - r * K;
        const float v = z[i];
        zl[r * ld + c] = v;
        zmax = fmaxf(zmax, fabsf(v));
    }
    for (int i = t; i < n_nodes * lg; i += 1024) gzl[i] = 0ull;
    for (int i = t; i < n_rel * K; i += 1024) wmax = fmaxf(wmax, fabsf(w[i]));
    zmax = block_max_1024(zmax, red, t);
    wmax = block_max_1024(wmax, red, t);
    // scales: identical in every workgroup (they depend on z and w only); per-TERM scale for d z (fx_add_term)
    double scale, scale_w;
    {
        const float tb = 4.f * zmax * wmax / (float)n_total;
        int ex = 100;
        if (tb > 0.f && tb < 3.0e38f) ex = 30 - (ilogbf(tb) + 1);
        scale = ldexp(1.0, ex > 100 ? 100 : (ex < -100 ? -100 : ex));
        const float bw = 4.f * zmax * zmax;
        ex = 60;
        if (bw > 0.f && bw < 3.0e38f) ex = 60 - (ilogbf(bw) + 1);
        scale_w = ldexp(1.0, ex > 180 ? 180 : ex);
    }
    const double scale_l = 1125899906842624.0;                            // 2^50: the objective is < 2^7
    if (blockIdx.x == 0 && t == 0) {                                      // the finalize kernel divides by them
        const int64_t base = (int64_t)n_nodes * K + (int64_t)n_rel * K + 1;
        reinterpret_cast<double*>(ws)[base] = scale;
        reinterpret_cast<double*>(ws)[base + 1] = scale_w;
        reinterpret_cast<double*>(ws)[base + 2] = scale_l;
    }
    __syncthreads();
    const float scale_f = (float)scale;
    const float inv_n = 1.f / (float)n_total;
    float loss = 0.f;

    // ids of (task, half h) for this lane: positions tb + h * 1024 + t (clamped; `valid` says whether it exists)
    struct Ids { int pu, pv, nu, nv; };
    const int col = t & (K - 1);
    auto fetch_wcol = [&](int task) -> float {                            // this lane's column of the task's w row
        task = task < n_tasks ? task : n_tasks - 1;
        return w[(int64_t)tasks[4 * task] * K + col];
    };
    // LDS byte addresses of this lane's column in row 0 of the two images (phase 2 adds a row's byte offset: one VOP2
    // add with the DPP broadcast folded in, instead of a 64-bit multiply-add per address)
    const unsigned c4 = (unsigned)(uintptr_t)(pr_lds_f32_t*)zl + (unsigned)col * 4u;
    const unsigned c8 = (unsigned)(uintptr_t)(pr_lds_u64_t*)gzl + (unsigned)col * 8u;
    auto fetch = [&](int task, int h, Ids& o) {
        task = task < n_tasks ? task : n_tasks - 1;
        const int tb = tasks[4 * task + 1], te = tasks[4 * task + 2];
        int64_t q = (int64_t)tb + h * 1024 + t;
        q = q < te ? q : (int64_t)te - 1;
        if constexpr (std::is_same<IT, PackedPair>::value) {               // one 32-bit word per pair: u | v << 16
            const uint32_t pw = pu[q].w, nw = nu[q].w;
            o.pu = (int)(pw & 0xffffu); o.pv = (int)(pw >> 16); o.nu = (int)(nw & 0xffffu); o.nv = (int)(nw >> 16);
        } else {
            o.pu = (int)pu[q]; o.pv = (int)pv[q]; o.nu = (int)nu[q]; o.nv = (int)nv[q];
        }
    };
    Ids cur0, cur1, nx0, nx1;
    fetch(blockIdx.x, 0, cur0);
    fetch(blockIdx.x, 1, cur1);
    float wcol = fetch_wcol(blockIdx.x), nx_wcol;
    for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int rel = tasks[4 * task], tb = tasks[4 * task + 1], te = tasks[4 * task + 2];
        const int pos_w = tasks[4 * task + 3];
        const float pw = (float)pos_w;
        fetch(task + gridDim.x, 0, nx0);                                  // the next task's ids (and this lane's column of
        fetch(task + gridDim.x, 1, nx1);                                  // its w row) travel during this one: nothing the
        nx_wcol = fetch_wcol(task + gridDim.x);                           // task itself uses comes from a vector load
        // (the task's w row is NOT loaded: column j sits in lane j of every 16-lane row -- `wcol`, prefetched a task ago --
        // and enters the dot product as the broadcast operand of v_fmac_f32_dpp; the s_nop covers the DPP read-after-write
        // hazard that the compiler does not see inside inline assembly)
        asm volatile("s_nop 1" : "+v"(wcol));
        constexpr int G = 64 / K;                                         // positions one scatter instruction covers
        const int grp = (t & 63) / K;
        const float wcs = wcol * scale_f;                                 // (a power of two: the products round the same)
        float gwc = 0.f;                                                  // this lane's column of d w[rel]
        // PHASE 1, one lane per position: score and the coefficient q of its gradient terms (0: nothing to add)
        auto triple = [&](int u, int v, bool negative, float weight) -> float {
            float d = 0.f;
            pr_static_for<K / 4>([&](auto jc) {
                constexpr int j = 4 * decltype(jc)::value;
                const float4 x = tipk_ld4(zl + u * ld + j), y = tipk_ld4(zl + v * ld + j);
                pr_fmac_row_bcast<j>(d, wcol, x.x * y.x); pr_fmac_row_bcast<j + 1>(d, wcol, x.y * y.y);
                pr_fmac_row_bcast<j + 2>(d, wcol, x.z * y.z); pr_fmac_row_bcast<j + 3>(d, wcol, x.w * y.w);
            });
            const float sg = fast_sigmoid(d);
            const float val = negative ? 1.f - sg : sg;
            loss -= weight * __logf(val + TIP_EPS);
            const float q = weight * inv_n * sg * (1.f - sg) * __builtin_amdgcn_rcpf(val + TIP_EPS);
            return negative ? q : -q;
        };
        // PHASE 2, K lanes per row: the wave's 64 positions are revisited G at a time; lane `col` of group `grp` adds
        // column `col` of both gradient terms of position i * G + grp -- one scatter instruction touches G rows of the
        // d z image, each as one contiguous 8 K-byte piece (a lane per position made every instruction touch 64 rows:
        // twice the time, tools/bench_decoder.py) -- and keeps its column of d w.  What travels per position is its
        // coefficient and the BYTE OFFSETS of its two rows in the two images (multiplied once, by the position's lane):
        // an address is one add.  Positions that do not exist carry q = 0 and clamped ids: they add zeros, no branch.
        // The steps are software-pipelined: the two row reads of step i + 1 are issued BEFORE the two adds of step i (LDS
        // operations complete in order, and the compiler will not move a read over an atomic it cannot prove disjoint):
        // a wave waits for reads that have only its own previous reads in front of them, not for its adds to drain.
        // NS streams (the positive's and the negative's terms) run through ONE pipeline of NS * 64 / G steps.
        auto scatter = [&](auto ns_c, float q_a, int u_a, int v_a, float q_b, int u_b, int v_b) {
            constexpr int NS = decltype(ns_c)::value, STEPS = 64 / G, TOTAL = NS * STEPS;
            // (second registers with the coefficients: each broadcast then has ONE user and folds into it -- v_mul_f32_dpp
            // for the scaled coefficient, v_fmac_f32_dpp for d w.  The s_nop: the hand-written DPP instruction reads them
            // from other lanes and the hazard recognizer does not look inside inline assembly -- 2 wait states)
            float q_a2 = q_a, q_b2 = q_b;
            asm volatile("s_nop 1" : "+v"(q_a2), "+v"(q_b2));
            const int zu_a = u_a * (ld * 4), zv_a = v_a * (ld * 4), gu_a = u_a * (lg * 8), gv_a = v_a * (lg * 8);
            const int zu_b = u_b * (ld * 4), zv_b = v_b * (ld * 4), gu_b = u_b * (lg * 8), gv_b = v_b * (lg * 8);
            float ua[TOTAL], vb[TOTAL];
            auto issue_reads = [&](auto jc) {
                constexpr int j = decltype(jc)::value, i = j % STEPS;
                constexpr bool second = j >= STEPS;
                unsigned zu, zv;
                if constexpr (K == 16) {
                    zu = (unsigned)pr_row_bcast<i>(second ? zu_b : zu_a) + c4;
                    zv = (unsigned)pr_row_bcast<i>(second ? zv_b : zv_a) + c4;
                } else {
                    const int src = i * G + grp;
                    zu = (unsigned)__shfl(second ? zu_b : zu_a, src, 64) + c4;
                    zv = (unsigned)__shfl(second ? zv_b : zv_a, src, 64) + c4;
                }
                ua[j] = *reinterpret_cast<pr_lds_f32_t*>((uintptr_t)zu);
                vb[j] = *reinterpret_cast<pr_lds_f32_t*>((uintptr_t)zv);
            };
            issue_reads(std::integral_constant<int, 0>{});
            pr_static_for<TOTAL>([&](auto jc) {
                constexpr int j = decltype(jc)::value, i = j % STEPS;
                constexpr bool second = j >= STEPS;
                if constexpr (j + 1 < TOTAL) issue_reads(std::integral_constant<int, j + 1>{});
                float qs, q = 0.f;
                unsigned gu, gv;
                if constexpr (K == 16) {
                    // group = one DPP row of 16 lanes: lane i of every row is broadcast to its row (row_newbcast:i, folded
                    // into the instruction that uses it) -- group g revisits the positions of lanes 16 g .. 16 g + 15
                    qs = __int_as_float(pr_row_bcast<i>(__float_as_int(second ? q_b : q_a))) * wcs;
                    gu = (unsigned)pr_row_bcast<i>(second ? gu_b : gu_a) + c8;
                    gv = (unsigned)pr_row_bcast<i>(second ? gv_b : gv_a) + c8;
                } else {
                    const int src = i * G + grp;
                    q = __shfl(second ? q_b : q_a, src, 64);
                    qs = q * wcs;
                    gu = (unsigned)__shfl(second ? gu_b : gu_a, src, 64) + c8;
                    gv = (unsigned)__shfl(second ? gv_b : gv_a, src, 64) + c8;
                }
                if (!TIPK_DBG(dbg & 1)) {
                    const pr_f32x2 term = (pr_f32x2){vb[j], ua[j]} * (pr_f32x2){qs, qs};   // v_pk_mul_f32
                    const int tu = pr_round_i32(term.x), tv = pr_round_i32(term.y);        // |term * scale| < 2^30 by construction
                    __hip_atomic_fetch_add(reinterpret_cast<pr_lds_u64_t*>((uintptr_t)gu), (unsigned long long)(long long)tu,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(reinterpret_cast<pr_lds_u64_t*>((uintptr_t)gv), (unsigned long long)(long long)tv,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if constexpr (K == 16) pr_fmac_row_bcast<i>(gwc, second ? q_b2 : q_a2, ua[j] * vb[j]);
                else gwc = fmaf(q, ua[j] * vb[j], gwc);
            });
        };
        const int cnt = te - tb;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const Ids& id = h ? cur1 : cur0;
            const bool valid = t + h * 1024 < cnt;
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;     // wave-uniform
            // (every lane evaluates -- the w row travels by DPP from lanes that must be active; a position that does not
            // exist has clamped ids and weight 0: it adds 0 to the objective and its coefficient is 0)
            float qp = 0.f, qn;
            if (pos_w != 0) qp = triple(id.pu, id.pv, false, valid ? pw : 0.f);
            qn = triple(id.nu, id.nv, true, valid ? 1.f : 0.f);
            if (want_grad) {
                if (pos_w != 0) scatter(std::integral_constant<int, 2>{}, qp, id.pu, id.pv, qn, id.nu, id.nv);
                else scatter(std::integral_constant<int, 1>{}, qn, id.nu, id.nv, 0.f, 0, 0);
            }
        }
        cur0 = nx0;
        cur1 = nx1;
        wcol = nx_wcol;
        if (want_grad) {
            // d w[rel]: the wave's G groups by shuffles, then ONE fixed-point add per column and WAVE straight into the
            // workspace.  No barrier anywhere in the task loop: the 16 waves of the workgroup drift apart, and the waves
            // that are evaluating positions (VALU, transcendentals) overlap the waves that are scattering (LDS) -- with the
            // per-task reduction through LDS (two barriers) every wave was in the same phase at the same time
#pragma unroll
            for (int o = K; o < 64; o <<= 1) gwc += __shfl_xor(gwc, o, 64);
            if (tipk_lane() < K)
                atomicAdd(ws + (int64_t)n_nodes * K + (int64_t)rel * K + col, (unsigned long long)(long long)((double)gwc * scale_w));
        }
    }
    __syncthreads();
    loss = block_sum_1024(loss, red, t);
    if (t == 0)
        atomicAdd(ws + (int64_t)n_nodes * K + (int64_t)n_rel * K, (unsigned long long)(long long)((double)(loss * inv_n) * scale_l));
    if (want_grad) {
        __syncthreads();
        // (one slab per workgroup summed by the finalize kernel instead of these atomics: measured 20 us SLOWER)
        for (int i = t; i < n_nodes * K; i += 1024) {
            const int r = i / K, c = i - r * K;
            const unsigned long long a = gzl[r * lg + c];
            if (a != 0ull) atomicAdd(ws + i, a);
        }
    }
}

inline bool task_path_ok(int64_t n_nodes, int k, int64_t* lds_bytes) {
    if (k % 4 != 0 || k < 4 || k > 64 || (k & (k - 1)) != 0) return false;
    if (n_nodes > 65535) return false;                    // ids are staged as 16-bit values
    *lds_bytes = ((n_nodes * (k + 1) + 1) & ~1LL) * 8 + (((n_nodes * (k + 4) + 3) & ~3LL) + 16 * k + 16) * (int64_t)sizeof(float) +
                 4 * TASK_MAX * 2;
    return *lds_bytes <= 158 * 1024;
}

// ws layout (u64 words): [n_nodes*k] d z | [n_rel*k] d w | [1] loss | 3 doubles: the three scales.
// Converts, ADDS into the outputs (same contract as the float path) and zeroes the words again, so the caller's
// workspace is reusable without a memset.
// store != 0: the outputs are OVERWRITTEN (0 + value, the same bits as adding into zeroed outputs) -- no zero fills.
__global__ __launch_bounds__(256) void det_finalize_kernel(unsigned long long* ws, int64_t n_z, int64_t n_w,
                                                           float* loss_out, float* g_z, float* g_w, int store) {
    const double* sc = reinterpret_cast<const double*>(ws + n_z + n_w + 1);
    const double inv_z = 1.0 / sc[0], inv_w = 1.0 / sc[1], inv_l = 1.0 / sc[2];
    const int64_t total = n_z + n_w + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const long long a = (long long)ws[i];
        if (a == 0 && !store) continue;
        if (a != 0) ws[i] = 0ull;
        if (i < n_z) { if (g_z) g_z[i] = (store ? 0.0f : g_z[i]) + (float)((double)a * inv_z); }
        else if (i < n_z + n_w) { if (g_w) g_w[i - n_z] = (store ? 0.0f : g_w[i - n_z]) + (float)((double)a * inv_w); }
        else loss_out[0] = (store ? 0.0f : loss_out[0]) + (float)((double)a * inv_l);
    }
}

int launch_finalize(int64_t n_nodes, int k, int64_t n_rel, float* loss_out, float* g_z, float* g_w, unsigned long long* ws,
                    hipStream_t st, int store) {
    const int64_t n_z = n_nodes * k, n_w = n_rel * k;
    const int64_t blocks = tipk_ceil_div(n_z + n_w + 1, 256);
    hipLaunchKernelGGL(det_finalize_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, ws, n_z,
                       n_w, loss_out, g_z, g_w, store);
    TIPK_RETURN_LAUNCH();
}

template <typename IT, int MODE>
int launch_tasks(const float* z, int64_t n_nodes, int k, const float* w, int64_t n_rel, const int32_t* tasks,
                 int64_t n_tasks, const void* pu, const void* pv, const void* nu, const void* nv,
                 const float* g_score, int sig, int64_t n_total, float* loss_out, float* g_z, float* g_w,
                 int64_t lds_bytes, hipStream_t st, unsigned long long* ws = nullptr, int store = 0) {
    auto kern = distmult_task_kernel<IT, MODE>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return tipk_hip_status(e);
    int64_t grid = n_tasks < 256 ? n_tasks : 256;              // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(1024), (size_t)lds_bytes, st, z, (int)n_nodes, k, w, (int)n_rel,
                       tasks, (int)n_tasks, (const IT*)pu, (const IT*)pv, (const IT*)nu, (const IT*)nv, g_score, sig,
                       n_total, loss_out, g_z, g_w, MODE == 1 ? ws : nullptr, TIPK_DBG(tipk_option(TIPK_OPT_DM_DEBUG)));
    if (ws && MODE == 1) return launch_finalize(n_nodes, k, n_rel, loss_out, g_z, g_w, ws, st, store);
    TIPK_RETURN_LAUNCH();
}

int launch_objective(const float* z, int64_t n_nodes, int k, const float* w, int64_t n_rel, const int32_t* tasks,
                     int64_t n_tasks, const void* pu, const void* pv, const void* nu, const void* nv, int idx_bytes,
                     int64_t n_total, float* loss_out, float* g_z, float* g_w, unsigned long long* ws, hipStream_t st,
                     int store) {
    const size_t lds = (size_t)(((n_nodes * (k + 1) + 1) & ~1LL) * 8 + (((n_nodes * (k + 4) + 3) & ~3LL) + 16 * k + 16) * 4);
    const int64_t grid = n_tasks < 256 ? n_tasks : 256;              // one persistent workgroup per CU
    const int dbg = TIPK_DBG(tipk_option(TIPK_OPT_DM_DEBUG));
#define OBJ(IT, KK)                                                                                                      \
    {                                                                                                                    \
        auto kern = distmult_objective_kernel<IT, KK>;                                                                   \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        if (e != hipSuccess) return tipk_hip_status(e);                                                                  \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(1024), lds, st, z, (int)n_nodes, w, (int)n_rel, tasks,      \
                           (int)n_tasks, (const IT*)pu, (const IT*)pv, (const IT*)nu, (const IT*)nv, n_total,           \
                           g_z != nullptr ? 1 : 0, ws, dbg);                                                             \
    }
    if (idx_bytes == 8) { if (k == 4) OBJ(int64_t, 4) else if (k == 8) OBJ(int64_t, 8) else OBJ(int64_t, 16) }
    else if (idx_bytes == 4) { if (k == 4) OBJ(int32_t, 4) else if (k == 8) OBJ(int32_t, 8) else OBJ(int32_t, 16) }
    else { if (k == 4) OBJ(PackedPair, 4) else if (k == 8) OBJ(PackedPair, 8) else OBJ(PackedPair, 16) }
#undef OBJ
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return tipk_hip_status(le);
    return launch_finalize(n_nodes, k, n_rel, loss_out, g_z, g_w, ws, st, store);
}

constexpr int64_t LDS_GZ_LIMIT = 96 * 1024;

template <typename IT, typename ET, int V, int MODE>
int launch_grad(const float* g_score, const float* score, const float* z, int64_t n_nodes, int k, const float* w,
                const void* iu, const void* iv, const void* ju, const void* jv, const void* et, int64_t n, int sig,
                float* loss_out, float* g_z, float* g_w, hipStream_t st) {
    const int64_t lds_bytes = n_nodes * k * (int64_t)sizeof(float);
    const bool use_lds = g_z && lds_bytes <= LDS_GZ_LIMIT;
    // enough workgroups to fill 256 CUs a few times over, but each long enough to amortise the
    // LDS image flush (n_nodes*k atomics)
    int64_t per_block = tipk_ceil_div(n, 1024);
    if (per_block < 4096) per_block = 4096;
    per_block = tipk_ceil_div(per_block, 256) * 256;
    const int64_t blocks = tipk_ceil_div(n, per_block);
    if (use_lds) {
        auto kern = distmult_grad_kernel<IT, ET, V, MODE, true>;
        if (lds_bytes > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds_bytes);
            if (e != hipSuccess) return tipk_hip_status(e);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), (size_t)lds_bytes, st, g_score, score, z,
                           n_nodes, k, w, iu, iv, ju, jv, et, n, per_block, sig, loss_out, g_z, g_w);
    } else {
        hipLaunchKernelGGL((distmult_grad_kernel<IT, ET, V, MODE, false>), dim3((unsigned)blocks), dim3(256), 0, st,
                           g_score, score, z, n_nodes, k, w, iu, iv, ju, jv, et, n, per_block, sig, loss_out, g_z,
                           g_w);
    }
    TIPK_RETURN_LAUNCH();
}

inline bool vec_ok(const float* z, const float* w, int k) {
    return k % 4 == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
}

}  // namespace

#define TIPK_DISPATCH_IDX(CALL)                                      \
    do {                                                             \
        if (idx_bytes == 4 && et_bytes == 4) { CALL(int32_t, int32_t); } \
        if (idx_bytes == 8 && et_bytes == 8) { CALL(int64_t, int64_t); } \
        if (idx_bytes == 4 && et_bytes == 8) { CALL(int32_t, int64_t); } \
        if (idx_bytes == 8 && et_bytes == 4) { CALL(int64_t, int32_t); } \
        return TIPK_EINVAL;                                          \
    } while (0)

extern "C" int tipk_distmult_fwd(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                                 const void* idx_u, const void* idx_v, int idx_bytes, const void* edge_type,
                                 int et_bytes, int64_t n_triples, int sigmoid, float* score, tipk_stream_t stream) {
    if (n_triples < 0 || k <= 0 || n_nodes < 0 || n_rel < 0) return TIPK_EINVAL;
    if (n_triples == 0) return TIPK_OK;
    if (!z || !rel_w || !idx_u || !idx_v || !edge_type || !score) return TIPK_EINVAL;
    const int64_t blocks = tipk_ceil_div(n_triples, 256);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(z, rel_w, k);
#define CALL(IT, ET)                                                                                             \
    {                                                                                                            \
        if (vec) hipLaunchKernelGGL((distmult_fwd_kernel<IT, ET, 4>), dim3((unsigned)blocks), dim3(256), 0, st, z, k, \
                                    rel_w, idx_u, idx_v, edge_type, n_triples, sigmoid, score);                 \
        else hipLaunchKernelGGL((distmult_fwd_kernel<IT, ET, 1>), dim3((unsigned)blocks), dim3(256), 0, st, z, k,   \
                                rel_w, idx_u, idx_v, edge_type, n_triples, sigmoid, score);                     \
        TIPK_RETURN_LAUNCH();                                                                                    \
    }
    TIPK_DISPATCH_IDX(CALL);
#undef CALL
}

extern "C" int tipk_distmult_bwd(const float* g_score, const float* score, const float* z, int64_t n_nodes, int k,
                                 const float* rel_w, int64_t n_rel, const void* idx_u, const void* idx_v,
                                 int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples, int sigmoid,
                                 const int32_t* tasks, int64_t n_tasks, float* g_z, float* g_w,
                                 tipk_stream_t stream) {
    if (n_triples < 0 || k <= 0 || n_nodes < 0 || n_rel < 0 || n_tasks < 0) return TIPK_EINVAL;
    if (n_triples == 0) return TIPK_OK;
    if (!g_score || !z || !rel_w || !idx_u || !idx_v || !edge_type || !g_z || !g_w || (sigmoid && !score))
        return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(z, rel_w, k);
    int64_t lds_bytes = 0;
    if (tasks && n_tasks > 0 && vec && n_tasks < 0x7fffffffLL && task_path_ok(n_nodes, k, &lds_bytes)) {
        if (idx_bytes == 8)
            return launch_tasks<int64_t, 0>(z, n_nodes, k, rel_w, n_rel, tasks, n_tasks, idx_u, idx_v, nullptr, nullptr,
                                            g_score, sigmoid, n_triples, nullptr, g_z, g_w, lds_bytes, st);
        if (idx_bytes == 4)
            return launch_tasks<int32_t, 0>(z, n_nodes, k, rel_w, n_rel, tasks, n_tasks, idx_u, idx_v, nullptr, nullptr,
                                            g_score, sigmoid, n_triples, nullptr, g_z, g_w, lds_bytes, st);
        return TIPK_EINVAL;
    }
#define CALL(IT, ET)                                                                                          \
    {                                                                                                         \
        if (vec) return launch_grad<IT, ET, 4, 0>(g_score, score, z, n_nodes, k, rel_w, idx_u, idx_v, nullptr, \
                                                  nullptr, edge_type, n_triples, sigmoid, nullptr, g_z, g_w, st); \
        return launch_grad<IT, ET, 1, 0>(g_score, score, z, n_nodes, k, rel_w, idx_u, idx_v, nullptr, nullptr,    \
                                         edge_type, n_triples, sigmoid, nullptr, g_z, g_w, st);                \
    }
    TIPK_DISPATCH_IDX(CALL);
#undef CALL
}

extern "C" int64_t tipk_distmult_workspace_bytes(int64_t n_nodes, int k, int64_t n_rel) {
    if (n_nodes < 0 || k <= 0 || n_rel < 0) return 0;
    return (n_nodes * k + n_rel * k + 1 + 3) * 8;
}

static int distmult_loss_impl(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                              const void* pos_u, const void* pos_v, const void* neg_u, const void* neg_v,
                              int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples,
                              const int32_t* tasks, int64_t n_tasks, float* loss_out, float* g_z, float* g_w,
                              void* workspace, tipk_stream_t stream, int store) {
    if (n_triples <= 0 || k <= 0 || n_nodes < 0 || n_rel < 0 || n_tasks < 0) return TIPK_EINVAL;
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 7)) return TIPK_EINVAL;
    unsigned long long* ws = reinterpret_cast<unsigned long long*>(workspace);
    if (!z || !rel_w || !pos_u || !neg_u || !edge_type || !loss_out) return TIPK_EINVAL;
    if (idx_bytes != 2 && (!pos_v || !neg_v)) return TIPK_EINVAL;
    if ((g_z == nullptr) != (g_w == nullptr)) return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(z, rel_w, k);
    int64_t lds_bytes = 0;
    if (idx_bytes == 2 && !(tasks && n_tasks > 0 && vec)) return TIPK_EUNSUPPORTED;
    if (tasks && n_tasks > 0 && vec && n_tasks < 0x7fffffffLL && task_path_ok(n_nodes, k, &lds_bytes)) {
        if (idx_bytes == 2 && !(ws && (k == 4 || k == 8 || k == 16) && n_nodes <= 65535)) return TIPK_EUNSUPPORTED;
        if (ws && (k == 4 || k == 8 || k == 16) && (idx_bytes == 8 || idx_bytes == 4 || idx_bytes == 2) && n_nodes <= 0x7fffffffLL &&
            (idx_bytes == 2 || !tipk_option(TIPK_OPT_DM_TASK_KERNEL)))
            return launch_objective(z, n_nodes, k, rel_w, n_rel, tasks, n_tasks, pos_u, pos_v, neg_u, neg_v, idx_bytes,
                                    n_triples, loss_out, g_z, g_w, ws, st, store);
        if (store && !ws) return TIPK_EUNSUPPORTED;       // only the finalize launch of the workspace path can overwrite
        if (idx_bytes == 8)
            return launch_tasks<int64_t, 1>(z, n_nodes, k, rel_w, n_rel, tasks, n_tasks, pos_u, pos_v, neg_u, neg_v, nullptr,
                                            1, n_triples, loss_out, g_z, g_w, lds_bytes, st, ws, store);
        if (idx_bytes == 4)
            return launch_tasks<int32_t, 1>(z, n_nodes, k, rel_w, n_rel, tasks, n_tasks, pos_u, pos_v, neg_u, neg_v, nullptr,
                                            1, n_triples, loss_out, g_z, g_w, lds_bytes, st, ws, store);
        return TIPK_EINVAL;
    }
    if (idx_bytes == 2 || store) return TIPK_EUNSUPPORTED;     // packed pairs / overwriting outputs: the workspace path only
#define CALL(IT, ET)                                                                                             \
    {                                                                                                            \
        if (vec) return launch_grad<IT, ET, 4, 1>(nullptr, nullptr, z, n_nodes, k, rel_w, pos_u, pos_v, neg_u, neg_v, \
                                                  edge_type, n_triples, 1, loss_out, g_z, g_w, st);             \
        return launch_grad<IT, ET, 1, 1>(nullptr, nullptr, z, n_nodes, k, rel_w, pos_u, pos_v, neg_u, neg_v,          \
                                         edge_type, n_triples, 1, loss_out, g_z, g_w, st);                      \
    }
    TIPK_DISPATCH_IDX(CALL);
#undef CALL
}

extern "C" int tipk_distmult_loss(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                                  const void* pos_u, const void* pos_v, const void* neg_u, const void* neg_v,
                                  int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples,
                                  const int32_t* tasks, int64_t n_tasks, float* loss_out, float* g_z, float* g_w,
                                  void* workspace, tipk_stream_t stream) {
    return distmult_loss_impl(z, n_nodes, k, rel_w, n_rel, pos_u, pos_v, neg_u, neg_v, idx_bytes, edge_type, et_bytes, n_triples,
                              tasks, n_tasks, loss_out, g_z, g_w, workspace, stream, 0);
}

extern "C" int tipk_distmult_loss_store(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                                        const void* pos_u, const void* pos_v, const void* neg_u, const void* neg_v,
                                        int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples,
                                        const int32_t* tasks, int64_t n_tasks, float* loss_out, float* g_z, float* g_w,
                                        void* workspace, tipk_stream_t stream) {
    return distmult_loss_impl(z, n_nodes, k, rel_w, n_rel, pos_u, pos_v, neg_u, neg_v, idx_bytes, edge_type, et_bytes, n_triples,
                              tasks, n_tasks, loss_out, g_z, g_w, workspace, stream, 1);
}
