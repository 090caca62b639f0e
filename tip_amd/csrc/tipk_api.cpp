// libtipk housekeeping entry points (include/tipk.h): version, status strings, device query.
#include <stdio.h>
#include <string.h>
#include "tipk_common.h"

extern "C" int tipk_abi_version(void) { return TIPK_ABI_VERSION; }

#ifndef TIPK_BUILD_ID
#define TIPK_BUILD_ID "unknown"
#endif
// digest of the sources this binary was compiled from (the Makefile passes it); the Python binding
// compares it with the digest of the sources next to it and refuses a stale library
// (the marker string lets the binding read the id from the FILE without loading it: dlopen of a path
// that is already loaded returns the old image, so a rebuilt library could not be told from a stale one)
#ifdef TIPK_DEBUG
static const char tipk_build_marker[] = "TIPK_BUILD_ID=" TIPK_BUILD_ID "+debug";
#else
static const char tipk_build_marker[] = "TIPK_BUILD_ID=" TIPK_BUILD_ID;
#endif
extern "C" const char* tipk_build_id(void) { return tipk_build_marker + 14; }

static int g_options[TIPK_OPT_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
static const char* const g_option_names[TIPK_OPT_COUNT] = {"gemm_no_stream", "gemm_thin_k_narrow", "gemm_stream_kk",
                                                            "rg_debug", "dp_debug", "rg_occupancy", "dm_debug", "dm_task_kernel"};

int tipk_option(int id) { return (id >= 0 && id < TIPK_OPT_COUNT) ? g_options[id] : 0; }

extern "C" int tipk_set_option(const char* name, int value) {
    if (!name) return TIPK_EINVAL;
    for (int i = 0; i < TIPK_OPT_COUNT; ++i) {
        if (strcmp(name, g_option_names[i]) != 0) continue;
#ifndef TIPK_DEBUG
        if ((i == TIPK_OPT_RG_DEBUG || i == TIPK_OPT_DP_DEBUG || i == TIPK_OPT_DM_DEBUG) && value != 0) return TIPK_EUNSUPPORTED;
#endif
        g_options[i] = value;
        return TIPK_OK;
    }
    return TIPK_EINVAL;
}

extern "C" int tipk_get_option(const char* name, int* value) {
    if (!name || !value) return TIPK_EINVAL;
    for (int i = 0; i < TIPK_OPT_COUNT; ++i)
        if (strcmp(name, g_option_names[i]) == 0) { *value = g_options[i]; return TIPK_OK; }
    return TIPK_EINVAL;
}

extern "C" const char* tipk_strerror(int status) {
    if (status == TIPK_OK) return "ok";
    if (status == TIPK_EINVAL) return "invalid argument";
    if (status == TIPK_EUNSUPPORTED) return "unsupported shape (see include/tipk.h limits)";
    if (status <= TIPK_EHIP_BASE) return hipGetErrorString((hipError_t)(TIPK_EHIP_BASE - status));
    return "unknown tipk status";
}

extern "C" int tipk_device_info(int device, int* n_cu, int* lds_bytes_per_cu, int* wavefront, char* arch,
                                int arch_len) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return tipk_hip_status(e);
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (wavefront) *wavefront = p.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return TIPK_OK;
}
