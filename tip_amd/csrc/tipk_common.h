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

// Host-side options of the library (tipk_api.cpp; include/tipk.h section 0): set explicitly through
// tipk_set_option -- no launch path reads the environment.  The *_DEBUG ids skip parts of a kernel's
// work (timing decompositions, tools/bench_*.py): they exist only in -DTIPK_DEBUG builds, a release
// library refuses them and its kernels carry no skip code.
enum {
    TIPK_OPT_GEMM_NO_STREAM = 0,      // every product through the LDS-tiled kernel
    TIPK_OPT_GEMM_THIN_K_NARROW = 1,  // dword body of the Y = att.XB streaming kernel
    TIPK_OPT_GEMM_STREAM_KK = 2,      // lane-per-row streaming body for d att
    TIPK_OPT_RG_DEBUG = 3,            // debug builds only
    TIPK_OPT_DP_DEBUG = 4,            // debug builds only
    TIPK_OPT_RG_OCCUPANCY = 5,        // tipk_rel_gather: workgroups per CU to aim for (0 = default, 1, 2)
    TIPK_OPT_DM_DEBUG = 6,            // debug builds only (decoder kernels)
    TIPK_OPT_DM_TASK_KERNEL = 7,      // fused objective through distmult_task_kernel (k / 4 lanes per position) -- A/B runs
    TIPK_OPT_COUNT = 8
};
int tipk_option(int id);
#ifdef TIPK_DEBUG
#define TIPK_DBG(expr) (expr)
#else
#define TIPK_DBG(expr) 0
#endif

#ifdef __HIPCC__
__device__ __forceinline__ float4 tipk_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void tipk_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ int tipk_lane() { return threadIdx.x & (TIPK_WAVE - 1); }
#endif
