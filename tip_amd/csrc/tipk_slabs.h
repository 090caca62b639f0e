// Ordered slab sums as a building block (tipk_sum_slabs_group, and as riders of tipk_gemm_wg_group / tipk_gather_sum_riders):
// the argument record, the host-side conversion of a tipk_slab_sum_desc and the body one 1024-thread workgroup runs.
#pragma once
#include "tipk_common.h"

namespace {

struct SlabArgs {
    const float* in; int64_t n_slabs, slab_stride, count; float alpha; int accumulate;
    const float* row_scale; int64_t cols; const float* addend; int relu; const float* gate; float* out; int lanes;
};

// one workgroup (1024 threads) of an ordered slab sum: `block` = its index inside the sum, red = 1024 floats of LDS
__device__ __forceinline__ void slab_sum_body(const SlabArgs& a, int block, float* red) {
    const int first = 0;
    const float* __restrict__ in = a.in;
    if (a.lanes == 1) {
        // 4 slab lanes, vectorised: ONE thread owns 4 consecutive elements and plays all four slab lanes itself
        // (lane j = slabs j, j+4, ... in order; then ((s0 + s1) + s2) + s3: bit for bit the sums of the 4-lane
        // layout below) -- 16-byte loads, eight of them in flight, no LDS and no barrier.  The dword layout
        // moved the 21 MB of the layer-1 d XB / d att slabs at 1.7 TB/s.
        const int64_t i = ((int64_t)(block - first) * 1024 + threadIdx.x) * 4;
        if (i >= a.count) return;
        const float4* p = reinterpret_cast<const float4*>(in + i);
        const int64_t st4 = a.slab_stride / 4;
        float4 s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t k = 0;
        for (; k + 8 <= a.n_slabs; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(k + u) * st4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s[u & 3].x += v[u].x; s[u & 3].y += v[u].y; s[u & 3].z += v[u].z; s[u & 3].w += v[u].w; }
        }
        {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t kc = k + u < a.n_slabs ? k + u : (a.n_slabs > 0 ? a.n_slabs - 1 : 0);   // clamped, unconditional
                v[u] = a.n_slabs > 0 ? p[kc * st4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < a.n_slabs) { s[u & 3].x += v[u].x; s[u & 3].y += v[u].y; s[u & 3].z += v[u].z; s[u & 3].w += v[u].w; }
        }
        float r[4] = {((s[0].x + s[1].x) + s[2].x) + s[3].x, ((s[0].y + s[1].y) + s[2].y) + s[3].y,
                      ((s[0].z + s[1].z) + s[2].z) + s[3].z, ((s[0].w + s[1].w) + s[2].w) + s[3].w};
        const float rs = a.row_scale ? a.row_scale[i / a.cols] : 1.f;              // cols % 4 == 0: one row per thread
        float4 ad = make_float4(0.f, 0.f, 0.f, 0.f), ac = ad, gt = make_float4(1.f, 1.f, 1.f, 1.f);
        if (a.addend) ad = *reinterpret_cast<const float4*>(a.addend + i);
        if (a.accumulate) ac = *reinterpret_cast<const float4*>(a.out + i);
        if (a.gate) gt = *reinterpret_cast<const float4*>(a.gate + i);
        const float adv[4] = {ad.x, ad.y, ad.z, ad.w}, acv[4] = {ac.x, ac.y, ac.z, ac.w}, gtv[4] = {gt.x, gt.y, gt.z, gt.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t = r[c] * a.alpha;
            if (a.row_scale) t *= rs;
            if (a.addend) t += adv[c];
            if (a.accumulate) t += acv[c];
            if (a.relu) t = fmaxf(t, 0.f);
            if (a.gate && !(gtv[c] > 0.f)) t = 0.f;
            r[c] = t;
        }
        *reinterpret_cast<float4*>(a.out + i) = make_float4(r[0], r[1], r[2], r[3]);
        return;
    }
    const int lanes = a.lanes, epb = 1024 / lanes;
    const int e = threadIdx.x % epb, j = threadIdx.x / epb;
    const int64_t i = (int64_t)(block - first) * epb + e;
    float s = 0.f;
    if (i < a.count) {
        const float* p = in + i;                           // four loads in flight, original order of additions
        int64_t k = j;
        for (; k + 3 * lanes < a.n_slabs; k += 4 * lanes) {
            const float v0 = p[k * a.slab_stride], v1 = p[(k + lanes) * a.slab_stride];
            const float v2 = p[(k + 2 * lanes) * a.slab_stride], v3 = p[(k + 3 * lanes) * a.slab_stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < a.n_slabs; k += lanes) s += p[k * a.slab_stride];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (j == 0 && i < a.count) {
        s = red[e];
        for (int q = 1; q < lanes; ++q) s += red[q * epb + e];
        s *= a.alpha;
        if (a.row_scale) s *= a.row_scale[i / a.cols];
        if (a.addend) s += a.addend[i];
        if (a.accumulate) s += a.out[i];
        if (a.relu) s = fmaxf(s, 0.f);
        if (a.gate && !(a.gate[i] > 0.f)) s = 0.f;
        a.out[i] = s;
    }
}

// converts one slab-sum descriptor (shared by tipk_sum_slabs_group and tipk_gemm_wg_group); returns its workgroups, 0 = nothing to do
inline int64_t fill_slab_args(const tipk_slab_sum_desc& d, SlabArgs& a, int* rc) {
    *rc = TIPK_OK;
    if (d.n_slabs < 0 || d.count < 0 || (d.row_scale && d.cols <= 0)) { *rc = TIPK_EINVAL; return 0; }
    if (d.count == 0) return 0;
    if (!d.out || (d.n_slabs > 0 && !d.in)) { *rc = TIPK_EINVAL; return 0; }
    a.in = d.in; a.n_slabs = d.n_slabs; a.slab_stride = d.slab_stride; a.count = d.count; a.alpha = d.alpha;
    a.accumulate = d.accumulate; a.row_scale = d.row_scale; a.cols = d.cols; a.addend = d.addend; a.relu = d.relu;
    a.gate = d.gate; a.out = d.out;
    // same slab-lane rule as tipk_sum_slabs_ex, so grouped and single launches add in the same order
    a.lanes = (d.n_slabs >= 32 && tipk_ceil_div(d.count, 64) < 2048) ? 16 : 4;
    // the 4-lane sums run vectorised (lanes = 1: same order of additions, 16-byte accesses) when alignment allows
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (a.lanes == 4 && d.count % 4 == 0 && d.slab_stride % 4 == 0 && al16(d.in) && al16(d.out) && al16(d.addend) &&
        al16(d.gate) && (!d.row_scale || d.cols % 4 == 0) && d.count >= 4096)
        a.lanes = 1;
    return a.lanes == 1 ? tipk_ceil_div(d.count, 4096) : tipk_ceil_div(d.count, 1024 / a.lanes);
}

}  // namespace
