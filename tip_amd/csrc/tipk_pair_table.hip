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
