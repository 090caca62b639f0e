// Shared helpers of libtipk (gfx950 only: 64-wide wavefronts are hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tipk.h"

#define TIPK_WAVE 64

static inline int tipk_hip_status(hipError_t e) {
    return e == hipSuccess ? TIPK_OK : TIPK_EHIP_BASE - (int)e;
}

// Launch-error check that does not synchronise (safe under stream capture).
#define TIPK_RETURN_LAUNCH()                          \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        return tipk_hip_status(e__);                  \
    } while (0)

static inline int64_t tipk_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
__device__ __forceinline__ float4 tipk_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void tipk_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ int tipk_lane() { return threadIdx.x & (TIPK_WAVE - 1); }
#endif
