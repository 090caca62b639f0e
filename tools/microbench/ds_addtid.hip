// What does ds_write_addtid_b32 add to M0: the lane number or the thread number inside the workgroup?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out, int hi) {
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -1;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)lds;
    const int v = 1000 * (threadIdx.x >> 6) + (threadIdx.x & 63);
    const unsigned m0v = hi ? (base | 0x3f800000u) : base;     // do the upper 16 bits of M0 matter?
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\ts_waitcnt lgkmcnt(0)" : : "v"(v), "s"(m0v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main() {
    int* d; hipMalloc(&d, 4096);
    for (int hi = 0; hi < 2; ++hi) {
        k<<<1, 256>>>(d, hi);
        int h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
        printf("M0 %s: ", hi ? "= base | 0x3f800000" : "= base");
        for (int i = 0; i < 160; i += 32) printf("lds[%d] = %d  ", i, h[i]);
        printf("\n");
    }
    return 0;
}
