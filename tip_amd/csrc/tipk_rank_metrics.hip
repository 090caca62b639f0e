// Per-relation ranking metrics on device (include/tipk.h section 6): AUPRC, AUROC, AP of every
// relation's [positives | negatives] score block, replacing the reference's 1 097 device->host copies
// and sklearn calls in `TIP.compute_auprc_auroc_ap_by_et` (src/layers.py:355-375, src/utils.py:86-93).
//
// One 1024-thread workgroup per relation.  The 2n scores are turned into 64-bit keys
// (order-preserving float bits << 1 | label), bitonic-sorted DESCENDING in LDS (up to 16 384 keys =
// 128 KB), then every thread walks 16 consecutive ranks: cumulative TP at a rank comes from a block
// scan of per-thread label counts, operating points are the ends of tie groups (sklearn's
// `_binary_clf_curve`: equal scores share one threshold), and the three metrics are sums over
// operating points of terms in (tp_k, fp_k, tp_{k-1}, fp_{k-1}) accumulated in fp64:
//   AUROC = sum (fpr_k - fpr_{k-1}) (tpr_k + tpr_{k-1}) / 2
//   AP    = sum (rec_k - rec_{k-1}) prec_k
//   AUPRC = trapezoid of (recall, precision) from (0, 1) up to the first point with full recall
//           (= metrics.auc over metrics.precision_recall_curve).
#include "tipk_common.h"

namespace {

constexpr int RM_MAX = 16384;           // keys per relation (LDS: 8 B each)
constexpr int RM_T = 1024;

__device__ __forceinline__ uint32_t orderable(float f) {       // larger float -> larger uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Carry {                           // last operating point at or before a position
    int idx;                             // -1 = none
    int tp;
};

__global__ __launch_bounds__(RM_T) void rank_metrics_kernel(const float* __restrict__ pos,
                                                            const float* __restrict__ neg,
                                                            const int64_t* __restrict__ range_ptr, int64_t n_rel,
                                                            double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    __shared__ int s_cnt[RM_T];
    __shared__ Carry s_carry[RM_T];
    __shared__ double s_red[3][16];
    const int t = threadIdx.x;
    const int64_t rel = blockIdx.x;
    const int64_t a = range_ptr[rel], b = range_ptr[rel + 1];
    const int n = (int)(b - a);                       // positives = negatives = n
    const int m = 2 * n;
    if (n <= 0) {                                     // empty relation: sklearn would raise; report NaN
        if (t < 3) out[t * n_rel + rel] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    int npad = 1;
    while (npad < m) npad <<= 1;
    for (int i = t; i < npad; i += RM_T) {
        unsigned long long k = 0ull;                  // padding sorts to the end (smallest key)
        if (i < n) k = ((unsigned long long)orderable(pos[a + i]) << 1) | 1ull;
        else if (i < m) k = ((unsigned long long)orderable(neg[a + i - n]) << 1);
        keys[i] = i < m ? k + 2ull : 0ull;            // +2: every real key is above the padding key 0
    }
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < npad; i += RM_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool desc = (i & k) == 0;   // this block sorts descending
                    if (desc ? x < y : x > y) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    // each thread owns `per` consecutive ranks
    const int per = (npad + RM_T - 1) / RM_T;         // 1..16
    const int lo = t * per;
    unsigned lab = 0, endf = 0;                       // bit e: label / "last of its tie group"
    int cnt = 0;
    for (int e = 0; e < per; ++e) {
        const int i = lo + e;
        if (i < m) {
            const unsigned long long k = keys[i];
            const unsigned long long kn = i + 1 < m ? keys[i + 1] : 0ull;
            if (k & 1ull) { lab |= 1u << e; ++cnt; }
            if (i + 1 >= m || (k >> 1) != (kn >> 1)) endf |= 1u << e;
        }
    }
    s_cnt[t] = cnt;
    __syncthreads();
    // exclusive scan of the per-thread positive counts (Hillis-Steele over 1024 entries)
    for (int off = 1; off < RM_T; off <<= 1) {
        const int v = t >= off ? s_cnt[t - off] : 0;
        __syncthreads();
        s_cnt[t] += v;
        __syncthreads();
    }
    int tp = s_cnt[t] - cnt;                          // positives ranked before this thread's chunk
    // last operating point inside this chunk (for the threads to the right)
    {
        Carry c = {-1, 0};
        int run = tp;
        for (int e = 0; e < per; ++e) {
            if (lab >> e & 1u) ++run;
            if (endf >> e & 1u) { c.idx = lo + e; c.tp = run; }
        }
        s_carry[t] = c;
    }
    __syncthreads();
    for (int off = 1; off < RM_T; off <<= 1) {        // inclusive "latest valid" scan
        Carry left = {-1, 0};
        if (t >= off) left = s_carry[t - off];
        __syncthreads();
        if (s_carry[t].idx < 0) s_carry[t] = left;
        __syncthreads();
    }
    Carry prev = {-1, 0};
    if (t > 0) prev = s_carry[t - 1];
    const double P = (double)n, N = (double)n;
    double tp_prev = prev.idx >= 0 ? (double)prev.tp : 0.0;
    double fp_prev = prev.idx >= 0 ? (double)(prev.idx + 1 - prev.tp) : 0.0;
    bool have_prev = prev.idx >= 0;
    double auroc = 0.0, ap = 0.0, auprc = 0.0;
    for (int e = 0; e < per; ++e) {
        if (lab >> e & 1u) ++tp;
        if (endf >> e & 1u) {
            const int i = lo + e;
            const double tpk = (double)tp, fpk = (double)(i + 1 - tp);
            const double prec = tpk / (tpk + fpk), rec = tpk / P;
            const double rec_p = tp_prev / P;
            const double prec_p = have_prev ? tp_prev / (tp_prev + fp_prev) : 1.0;
            auroc += (fpk / N - fp_prev / N) * (tpk / P + tp_prev / P) * 0.5;
            ap += (rec - rec_p) * prec;
            if (tp_prev < P) auprc += (rec - rec_p) * (prec + prec_p) * 0.5;   // stop after full recall
            tp_prev = tpk;
            fp_prev = fpk;
            have_prev = true;
        }
    }
    // block reduction in a fixed order (wave shuffles, then 16 waves)
    double v3[3] = {auprc, auroc, ap};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double v = v3[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) s_red[q][t >> 6] = v;
    }
    __syncthreads();
    if (t < 3) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += s_red[t][w];
        out[t * n_rel + rel] = s;
    }
}

}  // namespace

extern "C" int tipk_rank_metrics(const float* pos_score, const float* neg_score, const int64_t* range_ptr,
                                 int64_t n_rel, int64_t max_pairs, double* out, tipk_stream_t stream) {
    if (n_rel < 0 || max_pairs < 0) return TIPK_EINVAL;
    if (n_rel == 0) return TIPK_OK;
    if (!pos_score || !neg_score || !range_ptr || !out) return TIPK_EINVAL;
    if (2 * max_pairs > RM_MAX || n_rel > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    int npad = 2;
    while (npad < 2 * max_pairs) npad <<= 1;
    const size_t lds = (size_t)npad * sizeof(unsigned long long);
    hipError_t e = hipFuncSetAttribute((const void*)rank_metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL(rank_metrics_kernel, dim3((unsigned)n_rel), dim3(RM_T), lds, (hipStream_t)stream, pos_score,
                       neg_score, range_ptr, n_rel, out);
    TIPK_RETURN_LAUNCH();
}
