// libtipk housekeeping entry points (include/tipk.h): version, status strings, device query.
#include <stdio.h>
#include <string.h>
#include "tipk_common.h"

extern "C" int tipk_abi_version(void) { return TIPK_ABI_VERSION; }

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
