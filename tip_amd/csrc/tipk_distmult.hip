// DistMult decoder (include/tipk.h section 4): score = sigma(sum_k z[u,k] z[v,k] w[r,k]) and its
// gradients, plus the fused TIP training objective.  z (N x k) and w (R x k) are a few tens of KB:
// they stay L1/L2-resident, so the compulsory HBM traffic is the triple list itself
// (2 indices + relation id per triple) plus one score.
//
// Gradients: d z is scattered over at most N rows that every workgroup hits -> accumulated in a
// per-workgroup LDS image with ds_add_f32 and flushed once; d w[r] is reduced across the wave with
// shuffles whenever the wave's triples share one relation (triples arrive grouped by relation,
// src/utils.py:57-63), so global float atomics are per wave, not per triple (Guideline 12).
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
                                 float* g_z, float* g_w, tipk_stream_t stream) {
    if (n_triples < 0 || k <= 0 || n_nodes < 0 || n_rel < 0) return TIPK_EINVAL;
    if (n_triples == 0) return TIPK_OK;
    if (!g_score || !z || !rel_w || !idx_u || !idx_v || !edge_type || !g_z || !g_w || (sigmoid && !score))
        return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(z, rel_w, k);
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

extern "C" int tipk_distmult_loss(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                                  const void* pos_u, const void* pos_v, const void* neg_u, const void* neg_v,
                                  int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples,
                                  float* loss_out, float* g_z, float* g_w, tipk_stream_t stream) {
    if (n_triples <= 0 || k <= 0 || n_nodes < 0 || n_rel < 0) return TIPK_EINVAL;
    if (!z || !rel_w || !pos_u || !pos_v || !neg_u || !neg_v || !edge_type || !loss_out) return TIPK_EINVAL;
    if ((g_z == nullptr) != (g_w == nullptr)) return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = vec_ok(z, rel_w, k);
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
