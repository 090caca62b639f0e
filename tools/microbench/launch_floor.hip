// What a dependent kernel costs inside a replayed hipGraph on this box: N-node chains of
//   (a) an empty kernel, (b) one 64-thread wave doing load -> add -> store on one cache line,
//   (c) 1024 workgroups x 256 threads streaming 1 MB, (d) a 256-deep dependent load chain in one wave.
// Build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o build/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_rmw(float* p) { if (threadIdx.x == 0) p[0] += 1.f; }
__global__ void k_stream(const float4* a, float4* b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = a[i]; v.x += 1.f; b[i] = v;
}
__global__ void k_chain(const int* nxt, int* out, int depth) {
    int j = threadIdx.x;
    for (int i = 0; i < depth; ++i) j = nxt[j];
    out[threadIdx.x] = j;
}
// 256 slabs of 8192 floats summed in order: (i) one thread per element walking the slabs with 16 loads in flight
__global__ void k_slabs_serial(const float* s, float* o, int n, int slabs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int k = 0; k < slabs; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s[(size_t)(k + u) * n + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    o[i] = acc;
}

template <class F> float time_graph(hipStream_t st, int n_nodes, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n_nodes; ++i) launch(i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int w = 0; w < 10; ++w) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1000.f / 10 / n_nodes;
}

__global__ void k_wr(float4* b) { b[blockIdx.x * blockDim.x + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); }
__global__ void k_rd(const float4* a, float* o) {
    float4 v = a[blockIdx.x * blockDim.x + threadIdx.x];
    if (v.x == 123.f) o[0] = v.y;
}

__global__ void k_copy(const float4* a, float4* b) { const int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[i]; }
__global__ void k_indep(const float4* a, float4* b, float* o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = a[i];
    b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    if (v.x == 123.f) o[0] = v.y;
}
__global__ void k_copy1(const float* a, float* b) { const int i = blockIdx.x * blockDim.x + threadIdx.x; b[i] = a[i]; }

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *a, *b; int *nxt, *out;
    CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&b, 64 << 20)); CK(hipMemset(a, 0, 64 << 20));
    CK(hipMalloc(&nxt, 4096)); CK(hipMalloc(&out, 4096));
    std::vector<int> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (i * 17 + 5) % 1024;
    CK(hipMemcpy(nxt, h.data(), 4096, hipMemcpyHostToDevice));
    const int N = 100;
    printf("empty kernel                         %6.2f us / node\n", time_graph(st, N, [&](int) { k_empty<<<1, 64, 0, st>>>(); }));
    printf("empty kernel, 256 x 1024 threads     %6.2f us / node\n", time_graph(st, N, [&](int) { k_empty<<<256, 1024, 0, st>>>(); }));
    printf("one-lane read-modify-write           %6.2f us / node\n", time_graph(st, N, [&](int) { k_rmw<<<1, 64, 0, st>>>(a); }));
    printf("stream 1 MB -> 1 MB (ping-pong)      %6.2f us / node\n", time_graph(st, N, [&](int i) {
        k_stream<<<256, 256, 0, st>>>((const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b)); }));
    printf("stream 16 MB -> 16 MB (ping-pong)    %6.2f us / node\n", time_graph(st, N, [&](int i) {
        k_stream<<<4096, 256, 0, st>>>((const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b)); }));
    for (int kb : {16, 64, 256, 1024, 4096, 16384}) {
        const int blocks = kb * 1024 / 16 / 256;
        float t_w = time_graph(st, N, [&](int) { k_wr<<<blocks, 256, 0, st>>>((float4*)b); });
        float t_r = time_graph(st, N, [&](int) { k_rd<<<blocks, 256, 0, st>>>((const float4*)a, b + (32 << 18)); });
        float t_p = time_graph(st, N, [&](int i) { k_stream<<<blocks, 256, 0, st>>>((const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b)); });
        float t_s = time_graph(st, N, [&](int i) { k_stream<<<blocks, 256, 0, st>>>((const float4*)a, (float4*)b); });
        printf("%6d KB: write-only %6.2f  read-only %6.2f  ping-pong %6.2f  a->b every node %6.2f us / node\n", kb, t_w, t_r, t_p, t_s);
    }
    {
        const int blocks = 256;
        printf("1 MB variants (us / node):\n");
        printf("  copy a -> b                      %6.2f\n", time_graph(st, N, [&](int) { k_copy<<<blocks, 256, 0, st>>>((const float4*)a, (float4*)b); }));
        printf("  copy in place a -> a             %6.2f\n", time_graph(st, N, [&](int) { k_copy<<<blocks, 256, 0, st>>>((const float4*)a, (float4*)a); }));
        printf("  read a, write const to b         %6.2f\n", time_graph(st, N, [&](int) { k_indep<<<blocks, 256, 0, st>>>((const float4*)a, (float4*)b, b + (32 << 18)); }));
        printf("  copy a -> a + 1 MB               %6.2f\n", time_graph(st, N, [&](int) { k_copy<<<blocks, 256, 0, st>>>((const float4*)a, (float4*)(a + (1 << 18))); }));
        printf("  copy a -> b, 64 x 1024 threads   %6.2f\n", time_graph(st, N, [&](int) { k_copy<<<64, 1024, 0, st>>>((const float4*)a, (float4*)b); }));
        printf("  copy a -> b dwords, 1024 x 256   %6.2f\n", time_graph(st, N, [&](int) { k_copy1<<<1024, 256, 0, st>>>(a, b); }));
        printf("  copy 64 KB a -> b, 16 x 256      %6.2f\n", time_graph(st, N, [&](int) { k_copy<<<16, 256, 0, st>>>((const float4*)a, (float4*)b); }));
        printf("  copy 64 KB a -> b, 256 x 16...   %6.2f\n", time_graph(st, N, [&](int) { k_copy<<<64, 64, 0, st>>>((const float4*)a, (float4*)b); }));
    }
    for (int depth : {1, 4, 16, 64})
        printf("dependent load chain depth %3d       %6.2f us / node\n", depth, time_graph(st, N, [&](int) { k_chain<<<1, 64, 0, st>>>(nxt, out, depth); }));
    for (int slabs : {16, 64, 256})
        printf("ordered sum of %3d slabs x 8192      %6.2f us / node\n", slabs, time_graph(st, N, [&](int) { k_slabs_serial<<<32, 256, 0, st>>>(a, b, 8192, slabs); }));
    printf("ordered sum of 256 slabs x 20640     %6.2f us / node\n", time_graph(st, N, [&](int) { k_slabs_serial<<<81, 256, 0, st>>>(a, b, 20640, 256); }));
    return 0;
}
