// Relation-indexed pair scores from two dense node x relation tables (include/tipk.h section 4b):
//     score[e] = sigma( s1[u_e, r_e] + s2[v_e, r_e] )
// This is the per-triple part of the reference's NNDecoder (src/layers.py:598-637): with
// P = relu(z w1_l1), Q = relu(z w2_l1) the two relation-specific dot products of every triple are
// entries of S1 = P w1_l2^T and S2 = Q w2_l2^T (N x R, 2.8 MB for BioSNAP: L2-resident), so the
// E x l1 gathers and products of the reference collapse into two dense GEMMs plus one scalar
// gather per endpoint.  Backward scatters d score into d S1 / d S2 with global float atomics
// (8.3 M adds over 0.7 M cells: 33 MB of atomic traffic, light contention).
#include "tipk_common.h"

namespace {

template <typename T>
__device__ __forceinline__ int64_t ldi(const void* p, int64_t i) { return (int64_t) reinterpret_cast<const T*>(p)[i]; }

template <typename IT, typename ET>
__global__ __launch_bounds__(256) void pair_table_fwd_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                                             int64_t ld, const void* iu, const void* iv, const void* et,
                                                             int64_t n, int sig, float* __restrict__ score) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t r = ldi<ET>(et, e);
    const float x = s1[ldi<IT>(iu, e) * ld + r] + s2[ldi<IT>(iv, e) * ld + r];
    score[e] = sig ? 1.f / (1.f + expf(-x)) : x;
}

template <typename IT, typename ET>
__global__ __launch_bounds__(256) void pair_table_bwd_kernel(const float* __restrict__ g_score,
                                                             const float* __restrict__ score, int64_t ld,
                                                             const void* iu, const void* iv, const void* et, int64_t n,
                                                             int sig, float* __restrict__ g_s1, float* __restrict__ g_s2) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t r = ldi<ET>(et, e);
    float q = g_score[e];
    if (sig) { const float s = score[e]; q *= s * (1.f - s); }
    atomicAdd(g_s1 + ldi<IT>(iu, e) * ld + r, q);
    atomicAdd(g_s2 + ldi<IT>(iv, e) * ld + r, q);
}

}  // namespace

#define TIPK_PT_DISPATCH(KERN, ...)                                                                                   \
    do {                                                                                                              \
        if (idx_bytes == 8 && et_bytes == 8) hipLaunchKernelGGL((KERN<int64_t, int64_t>), grid, dim3(256), 0, st, __VA_ARGS__); \
        else if (idx_bytes == 4 && et_bytes == 4) hipLaunchKernelGGL((KERN<int32_t, int32_t>), grid, dim3(256), 0, st, __VA_ARGS__); \
        else if (idx_bytes == 8 && et_bytes == 4) hipLaunchKernelGGL((KERN<int64_t, int32_t>), grid, dim3(256), 0, st, __VA_ARGS__); \
        else if (idx_bytes == 4 && et_bytes == 8) hipLaunchKernelGGL((KERN<int32_t, int64_t>), grid, dim3(256), 0, st, __VA_ARGS__); \
        else return TIPK_EINVAL;                                                                                      \
    } while (0)

extern "C" int tipk_pair_table_fwd(const float* s1, const float* s2, int64_t ld, const void* idx_u, const void* idx_v,
                                   int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples, int sigmoid,
                                   float* score, tipk_stream_t stream) {
    if (n_triples < 0 || ld <= 0) return TIPK_EINVAL;
    if (n_triples == 0) return TIPK_OK;
    if (!s1 || !s2 || !idx_u || !idx_v || !edge_type || !score) return TIPK_EINVAL;
    const int64_t blocks = tipk_ceil_div(n_triples, 256);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    TIPK_PT_DISPATCH(pair_table_fwd_kernel, s1, s2, ld, idx_u, idx_v, edge_type, n_triples, sigmoid, score);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_pair_table_bwd(const float* g_score, const float* score, int64_t ld, const void* idx_u,
                                   const void* idx_v, int idx_bytes, const void* edge_type, int et_bytes,
                                   int64_t n_triples, int sigmoid, float* g_s1, float* g_s2, tipk_stream_t stream) {
    if (n_triples < 0 || ld <= 0) return TIPK_EINVAL;
    if (n_triples == 0) return TIPK_OK;
    if (!g_score || !idx_u || !idx_v || !edge_type || !g_s1 || !g_s2 || (sigmoid && !score)) return TIPK_EINVAL;
    const int64_t blocks = tipk_ceil_div(n_triples, 256);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    TIPK_PT_DISPATCH(pair_table_bwd_kernel, g_score, score, ld, idx_u, idx_v, edge_type, n_triples, sigmoid, g_s1, g_s2);
    TIPK_RETURN_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Fused TIP objective of the table decoder (include/tipk.h section 4b; src/layers.py:335-340 with the NNDecoder of
// src/layers.py:598-637 as the scorer, model/ddm-nn.py:65-102):
//
//     loss = -mean log(sigma(x_pos) + eps) - mean log(1 - sigma(x_neg) + eps),   x = S1[u, r] + S2[v, r]
//
// on TRANSPOSED tables s1t / s2t [R][ld]: row r = the scores of every node under relation r, 2.6 KB at BioSNAP.  One
// workgroup per relation (heaviest first): both rows are staged in LDS, a thread evaluates a position (positive and
// negative pair of the same place in the relation's block: two LDS reads each, sigma / log once on the transcendental
// units) and adds d loss / d x to the relation's two GRADIENT ROWS in LDS as 64-bit fixed point (ds_add_u64: exact, so
// the sums do not depend on the order the lanes arrive in -- bitwise reproducible; LDS float atomics are 16 x slower on
// gfx950 and order-dependent).  The rows leave as contiguous fp32 rows of g_s1t / g_s2t: no global atomics, no workspace,
// no zero fills (every row of the outputs is written by its relation's workgroup).  The two sums of logs of a relation
// leave as doubles, added over the relations by the caller.
namespace {

constexpr int PTL_THREADS = 1024;
constexpr float PTL_FIX = 68719476736.0f;           // 2^36: |d loss / d x| * n <= 1 -> a term < 2^37, 2^26 terms fit in 63 bits

struct PtlArgs {
    const float* s1t; const float* s2t; int64_t ld;
    int n_nodes, n_rel;
    const uint32_t* pos; const uint32_t* neg;       // packed pairs u | v << 16, grouped by relation
    const int64_t* rel_ptr;                         // [n_rel + 1]
    const int32_t* order;                           // [n_rel] relations by decreasing size
    float eps; double inv_n;                        // 1 / number of positions (of either sign)
    double* loss_parts;                             // [n_rel][2]
    float* g_s1t; float* g_s2t;                     // nullable (objective only)
};

__global__ __launch_bounds__(PTL_THREADS) void pair_table_loss_kernel(PtlArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ptl_lds[];
    const int n = a.n_nodes, t = threadIdx.x;
    unsigned long long* g1 = reinterpret_cast<unsigned long long*>(ptl_lds);      // [n] fixed-point d S1[:, r]
    unsigned long long* g2 = g1 + n;
    float* s1 = reinterpret_cast<float*>(g2 + n);                                  // [n] S1[:, r]
    float* s2 = s1 + n;
    double* red = reinterpret_cast<double*>(ptl_lds);                              // (after the rows are written out)
    const int r = a.order[blockIdx.x];
    const float* r1 = a.s1t + (int64_t)r * a.ld;
    const float* r2 = a.s2t + (int64_t)r * a.ld;
    for (int i = t; i < n; i += PTL_THREADS) { s1[i] = r1[i]; s2[i] = r2[i]; g1[i] = 0ull; g2[i] = 0ull; }
    __syncthreads();
    const int64_t b = a.rel_ptr[r], e = a.rel_ptr[r + 1];
    const bool grad = a.g_s1t != nullptr;
    double lp = 0.0, ln = 0.0;
    for (int64_t p = b + t; p < e; p += PTL_THREADS) {
        const uint32_t wp = a.pos[p], wn = a.neg[p];
        const int pu = wp & 0xffffu, pv = wp >> 16, nu = wn & 0xffffu, nv = wn >> 16;
        const float sp = 1.f / (1.f + __expf(-(s1[pu] + s2[pv])));
        const float sn = 1.f / (1.f + __expf(-(s1[nu] + s2[nv])));
        lp += (double)__logf(sp + a.eps);
        ln += (double)__logf(1.f - sn + a.eps);
        if (grad) {
            // n * d loss / d x: -(1 - sigma) sigma / (sigma + eps) for a positive, +sigma (1 - sigma) / (1 - sigma + eps) for a negative
            const long long qp = -__float2ll_rn(sp * (1.f - sp) / (sp + a.eps) * PTL_FIX);
            const long long qn = __float2ll_rn(sn * (1.f - sn) / (1.f - sn + a.eps) * PTL_FIX);
            atomicAdd(g1 + pu, (unsigned long long)qp); atomicAdd(g2 + pv, (unsigned long long)qp);
            atomicAdd(g1 + nu, (unsigned long long)qn); atomicAdd(g2 + nv, (unsigned long long)qn);
        }
    }
    __syncthreads();
    if (grad) {
        const double k = a.inv_n / (double)PTL_FIX;
        float* o1 = a.g_s1t + (int64_t)r * a.ld;
        float* o2 = a.g_s2t + (int64_t)r * a.ld;
        for (int i = t; i < n; i += PTL_THREADS) {
            o1[i] = (float)((double)(long long)g1[i] * k);
            o2[i] = (float)((double)(long long)g2[i] * k);
        }
        __syncthreads();
    }
    // the relation's two sums of logs: threads in index order, fixed tree
    red[t] = lp; red[PTL_THREADS + t] = ln;
    __syncthreads();
    for (int s = PTL_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) { red[t] += red[t + s]; red[PTL_THREADS + t] += red[PTL_THREADS + t + s]; }
        __syncthreads();
    }
    if (t == 0) { a.loss_parts[2 * r] = red[0]; a.loss_parts[2 * r + 1] = red[PTL_THREADS]; }
}

}  // namespace

extern "C" int tipk_pair_table_loss(const float* s1t, const float* s2t, int64_t ld, int64_t n_nodes, int64_t n_rel,
                                    const uint32_t* pos_pairs, const uint32_t* neg_pairs, const int64_t* rel_ptr,
                                    const int32_t* order, int64_t n_positions, float eps, double* loss_parts,
                                    float* g_s1t, float* g_s2t, tipk_stream_t stream) {
    if (n_rel <= 0 || n_nodes <= 0 || n_nodes > 65535 || ld < n_nodes || n_positions <= 0) return TIPK_EINVAL;
    if (!s1t || !s2t || !pos_pairs || !neg_pairs || !rel_ptr || !order || !loss_parts || ((g_s1t == nullptr) != (g_s2t == nullptr)))
        return TIPK_EINVAL;
    size_t lds = (size_t)n_nodes * 24;
    if (lds < 2 * PTL_THREADS * sizeof(double)) lds = 2 * PTL_THREADS * sizeof(double);
    if (lds > 150 * 1024) return TIPK_EUNSUPPORTED;
    if (n_rel > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    PtlArgs a;
    a.s1t = s1t; a.s2t = s2t; a.ld = ld; a.n_nodes = (int)n_nodes; a.n_rel = (int)n_rel;
    a.pos = pos_pairs; a.neg = neg_pairs; a.rel_ptr = rel_ptr; a.order = order;
    a.eps = eps; a.inv_n = 1.0 / (double)n_positions; a.loss_parts = loss_parts; a.g_s1t = g_s1t; a.g_s2t = g_s2t;
    hipError_t e = hipFuncSetAttribute((const void*)pair_table_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL(pair_table_loss_kernel, dim3((unsigned)n_rel), dim3(PTL_THREADS), lds, (hipStream_t)stream, a);
    TIPK_RETURN_LAUNCH();
}
