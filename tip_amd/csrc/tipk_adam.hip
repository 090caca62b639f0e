// One Adam step over a LIST of fp32 tensors in one launch (include/tipk.h section 9) -- the optimizer step of the
// reference's training loop (tip.py:24-30: torch.optim.Adam(model.parameters(), lr=0.01), amsgrad off).
//
// TIP-cat has 13 parameter tensors, 1.4 M floats in all.  torch's fused multi-tensor Adam gives every workgroup a chunk
// of 65536 elements: 22 workgroups on 256 CUs, 44 us of the 0.9 ms graphed epoch.  Here a workgroup owns 1024 elements
// (256 threads x float4), ~1400 workgroups: the step is four streams in, three out (39 MB) and takes a launch floor.
//
// The tensors' addresses travel BY VALUE in the kernel arguments (up to ADAM_MAX_TENSORS per launch; longer lists take
// several launches): no descriptor table in device memory, nothing to upload, and a captured hipGraph keeps what it
// captured.  The step counts live in device memory, one per tensor as in torch.optim.Adam (a parameter without gradient
// sits a step out), plus a ticket word: every workgroup reads its tensor's count, the LAST one to finish (ticket)
// advances the counts of the launch's tensors -- a captured step keeps counting on replay without a launch of its own.
#include <math.h>
#include "tipk_common.h"

namespace {

constexpr int ADAM_MAX_TENSORS = 48;
constexpr int ADAM_CHUNK = 1024;

struct AdamArgs {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    int32_t first_wg[ADAM_MAX_TENSORS + 1];   // workgroups [first_wg[i], first_wg[i + 1]) own tensor i
    int64_t n[ADAM_MAX_TENSORS];
    unsigned long long* step[ADAM_MAX_TENSORS];   // device: steps done so far, per tensor
    unsigned long long* ticket;               // device [33 * 16], 0 between launches
    int n_tensors;
    int dbg;                                  // -DTIPK_DEBUG builds: 1 no ticket, 2 no bias-correction arithmetic, 4 no stores
    double lr, beta1, beta2, eps, weight_decay;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1, float b2, float wd, float step_size,
                                         float inv_sqrt_bc2, float eps) {
    if (wd != 0.0f) g = fmaf(wd, p, g);                       // L2 form (torch.optim.Adam, not AdamW)
    m = fmaf(1.0f - b1, g - m, m);                            // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(1.0f - b2, g * g, b2 * v);                       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a) {
    __shared__ float sc[2];
    const int t = threadIdx.x, wg = blockIdx.x;
    int ti = 0;
    while (ti + 1 < a.n_tensors && wg >= a.first_wg[ti + 1]) ++ti;
    const unsigned long long step = *a.step[ti] + 1ull;        // (uniform: one scalar load)
    if (t == 0 && TIPK_DBG(a.dbg & 2)) { sc[0] = (float)a.lr; sc[1] = 1.0f; }
    else if (t == 0) {
        // beta^step by squaring (<= 64 dependent multiplies; the library pow is a few hundred double instructions on one lane)
        double p1 = 1.0, p2 = 1.0, s1 = a.beta1, s2 = a.beta2;
        for (unsigned long long e = step; e != 0ull; e >>= 1) {
            if (e & 1ull) { p1 *= s1; p2 *= s2; }
            s1 *= s1; s2 *= s2;
        }
        const double bc1 = 1.0 - p1, bc2 = 1.0 - p2;
        sc[0] = (float)(a.lr / bc1);
        sc[1] = (float)(1.0 / sqrt(bc2));
    }
    const int64_t i0 = (int64_t)(wg - a.first_wg[ti]) * ADAM_CHUNK + t * 4;
    const int64_t n = a.n[ti];
    float* __restrict__ p = a.p[ti];
    const float* __restrict__ g = a.g[ti];
    float* __restrict__ m = a.m[ti];
    float* __restrict__ v = a.v[ti];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const float b1 = (float)a.beta1, b2 = (float)a.beta2, wd = (float)a.weight_decay, eps = (float)a.eps;
    float pr[4], gr[4], mr[4], vr[4];
    const bool full = vec && i0 + 4 <= n;
    if (full) {
        *reinterpret_cast<float4*>(pr) = *reinterpret_cast<const float4*>(p + i0);
        *reinterpret_cast<float4*>(gr) = *reinterpret_cast<const float4*>(g + i0);
        *reinterpret_cast<float4*>(mr) = *reinterpret_cast<const float4*>(m + i0);
        *reinterpret_cast<float4*>(vr) = *reinterpret_cast<const float4*>(v + i0);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = i0 + j < n;
            pr[j] = in ? p[i0 + j] : 0.0f; gr[j] = in ? g[i0 + j] : 0.0f;
            mr[j] = in ? m[i0 + j] : 0.0f; vr[j] = in ? v[i0 + j] : 0.0f;
        }
    }
    __syncthreads();
    const float step_size = sc[0], inv_sqrt_bc2 = sc[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) adam_one(pr[j], gr[j], mr[j], vr[j], b1, b2, wd, step_size, inv_sqrt_bc2, eps);
    if (TIPK_DBG(a.dbg & 4) && pr[0] != 12345.f) {}
    else if (full) {
        *reinterpret_cast<float4*>(p + i0) = *reinterpret_cast<const float4*>(pr);
        *reinterpret_cast<float4*>(m + i0) = *reinterpret_cast<const float4*>(mr);
        *reinterpret_cast<float4*>(v + i0) = *reinterpret_cast<const float4*>(vr);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) { p[i0 + j] = pr[j]; m[i0 + j] = mr[j]; v[i0 + j] = vr[j]; }
    }
    // the last workgroup to get here advances the counts: every workgroup read its own (and waited for it) before this point.
    // Two-level ticket: all workgroups finish together, and 800 returning atomics on ONE word are served one after the
    // other (9 of the kernel's 14.6 us, tools/bench_adam.py) -- a workgroup takes a ticket of its group (wg mod 32), the
    // last of a group one of the top word: 25 + 32 atomics deep instead of 800.
    __shared__ int last;
    if (TIPK_DBG(a.dbg & 1)) return;
    if (t == 0) {
        const unsigned n_wg = gridDim.x, grp = (unsigned)wg & 31u;
        const unsigned in_grp = (n_wg + 31u - grp) / 32u, n_grp = n_wg < 32u ? n_wg : 32u;
        int l = 0;
        if (atomicAdd(a.ticket + 16 * (1 + grp), 1ull) == (unsigned long long)in_grp - 1ull) {   // (a cache line per word)
            a.ticket[16 * (1 + grp)] = 0ull;
            l = atomicAdd(a.ticket, 1ull) == (unsigned long long)n_grp - 1ull;
        }
        last = l;
    }
    __syncthreads();
    if (last) {
        if (t < a.n_tensors) *a.step[t] += 1ull;
        if (t == 0) *a.ticket = 0ull;
    }
}

}  // namespace

extern "C" int tipk_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, uint64_t* const* steps, uint64_t* ticket,
                              double lr, double beta1, double beta2, double eps, double weight_decay, tipk_stream_t stream) {
    if (n_tensors < 0 || !ticket || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !steps)))
        return TIPK_EINVAL;
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0))
        return TIPK_EINVAL;
    for (int i = 0; i < n_tensors; ++i)
        if (numel[i] < 0 || (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || !steps[i]))) return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int last = -1;
    for (int i = 0; i < n_tensors; ++i) if (numel[i] > 0) last = i;
    if (last < 0) return TIPK_OK;
    int i = 0;
    while (i <= last) {
        AdamArgs a;
        a.n_tensors = 0; a.first_wg[0] = 0;
        a.ticket = reinterpret_cast<unsigned long long*>(ticket);
        a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_DM_DEBUG));
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
        int64_t wgs = 0;
        for (; i <= last && a.n_tensors < ADAM_MAX_TENSORS; ++i) {
            if (numel[i] == 0) continue;
            const int64_t c = tipk_ceil_div(numel[i], ADAM_CHUNK);
            if (wgs + c > 0x3fffffffLL) { if (a.n_tensors == 0) return TIPK_EUNSUPPORTED; break; }
            const int k = a.n_tensors++;
            a.p[k] = params[i]; a.g[k] = grads[i]; a.m[k] = exp_avg[i]; a.v[k] = exp_avg_sq[i]; a.n[k] = numel[i];
            a.step[k] = reinterpret_cast<unsigned long long*>(steps[i]);
            wgs += c;
            a.first_wg[k + 1] = (int32_t)wgs;
        }
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return tipk_hip_status(e);
    }
    return TIPK_OK;
}
