// One-shot all-reduce over peer-mapped mailboxes (include/tipk.h section 8) -- the xGMI-aware exchange of
// SURVEY.md section 8(e) for the step's small collectives (41 KB ... 5 MB).
//
// The eight GPUs of an MI355X node are fully connected (7 xGMI links per GPU, ~153 GB/s each).  A ring all-reduce moves
// 2 (w - 1) / w of the buffer over ONE link per hop in 2 (w - 1) dependent steps: at BioSNAP sizes each step is pure
// latency.  Here every rank owns a MAILBOX (uncached device memory, mapped into the other ranks' address spaces through
// hipIpc) with one slot per rank; an all-reduce is ONE kernel per rank:
//
//   1. write this rank's buffer into slot `rank` of every mailbox -- w - 1 remote streams that leave over w - 1
//      DIFFERENT links at the same time (plus the local copy);
//   2. system-scope release, then store the call's sequence number into flag (rank, chunk) of every mailbox;
//   3. wait until the own mailbox shows the sequence number in the flags of all w ranks for this chunk;
//   4. add the w slots IN RANK ORDER into the buffer: every rank computes the same bits, run after run.
//
// Chunks (one workgroup each) are independent: no grid-wide barrier.  The sequence number lives in device memory and is
// advanced by a one-thread kernel behind the exchange (the sampler's scheme), so a captured hipGraph that contains the
// exchange replays correctly.  Two mailbox halves alternate with the parity of the sequence number: a rank that is
// already writing call s + 1 cannot overwrite what a slower rank still reads for call s, and it cannot reach call
// s + 2 before every rank has posted s + 1, i.e. has finished reading s.
//
// BOUNDED WAIT: a rank that died, skipped a call or took another collective would leave the others spinning for ever --
// on a shared node a hung lease, not an error.  The wait of step 3 therefore has a wall-clock budget (s_memrealtime,
// 100 MHz; tipk_peer_set_timeout_ms, default 2 000 ms): when it runs out the workgroup records {sequence number, first
// missing rank, chunk} in the mailbox's ERROR WORD and goes on with whatever the slots hold.  The result of that call is
// garbage BY DESIGN; the host reads the error word at its next synchronisation point (tipk_peer_status) and the rank
// exits non-zero (tip_amd.dist.DirectExchange.check) -- never a re-exec of a process that touched the GPU.
//
// Mailboxes are allocated and freed through explicit entry points (tipk_peer_alloc / tipk_peer_free: uncached memory
// cannot come from the caller's allocator); the exchange itself allocates nothing.
#include <string.h>
#include "tipk_common.h"

namespace {

constexpr int PEER_CHUNK = 4096;          // floats per workgroup
constexpr int PEER_MAX_WORLD = 16;

struct PeerArgs {
    float* data; int64_t n;
    char* box[PEER_MAX_WORLD];            // the ranks' mailboxes as mapped HERE (own mailbox at [rank])
    int rank, world;
    int64_t max_floats, half_bytes, n_chunks_max;
    const unsigned long long* seq;        // device: number of exchanges done so far
    unsigned long long* err;              // device (own mailbox): 0, or the record of the first wait that timed out
    unsigned long long budget;            // s_memrealtime ticks (100 MHz) a flag wait may take
};

// mailbox layout (one half): slots [world][max_floats] fp32, padded to 256 bytes | flags [world][n_chunks_max] u64
// (the flag area starts on a 256-byte boundary whatever world and max_floats are: u64 system-scope atomics on it are aligned)
__host__ __device__ __forceinline__ int64_t peer_slot_bytes(int world, int64_t max_floats) {
    return ((int64_t)world * max_floats * 4 + 255) / 256 * 256;
}
__device__ __forceinline__ float* peer_slot(char* box, int half, int64_t half_bytes, int64_t max_floats, int r) {
    return reinterpret_cast<float*>(box + half * half_bytes) + (int64_t)r * max_floats;
}
__device__ __forceinline__ unsigned long long* peer_flag(char* box, int half, int64_t half_bytes, int64_t max_floats, int world,
                                                         int64_t n_chunks_max, int r, int64_t chunk) {
    return reinterpret_cast<unsigned long long*>(box + half * half_bytes + peer_slot_bytes(world, max_floats)) + (int64_t)r * n_chunks_max + chunk;
}

__global__ __launch_bounds__(256) void peer_allreduce_kernel(PeerArgs a) {
    const int t = threadIdx.x;
    const int64_t chunk = blockIdx.x;
    const int64_t i0 = chunk * PEER_CHUNK;
    const int cnt = (int)(a.n - i0 < PEER_CHUNK ? a.n - i0 : PEER_CHUNK);
    const unsigned long long seq = *a.seq + 1ull;
    const int half = (int)(seq & 1ull);
    // 1. this rank's chunk -> slot `rank` of every mailbox (remote ones first: their links start early)
    for (int pp = 1; pp <= a.world; ++pp) {
        const int p = (a.rank + pp) % a.world;
        float* dst = peer_slot(a.box[p], half, a.half_bytes, a.max_floats, a.rank) + i0;
        for (int i = t; i < cnt; i += 256) dst[i] = a.data[i0 + i];
    }
    // 2. make the stores visible system-wide, then post the flags
    __threadfence_system();
    __syncthreads();
    if (t < a.world) {
        unsigned long long* f = peer_flag(a.box[t], half, a.half_bytes, a.max_floats, a.world, a.n_chunks_max, a.rank, chunk);
        __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 3. wait for all ranks' flags of this chunk in the own mailbox
    if (t < a.world) {
        unsigned long long* f = peer_flag(a.box[a.rank], half, a.half_bytes, a.max_floats, a.world, a.n_chunks_max, t, chunk);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.budget) {
                // record (first writer wins) and give up: bits 63..32 = sequence number, 31..16 = missing rank + 1, 15..0 = chunk
                unsigned long long rec = (seq << 32) | ((unsigned long long)(t + 1) << 16) | (unsigned long long)(chunk & 0xffff);
                unsigned long long expect = 0ull;
                __hip_atomic_compare_exchange_strong(a.err, &expect, rec, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    // 4. ordered sum of the slots (uncached memory: the loads see what the peers wrote)
    for (int i = t; i < cnt; i += 256) {
        float s = peer_slot(a.box[a.rank], half, a.half_bytes, a.max_floats, 0)[i0 + i];
        for (int r = 1; r < a.world; ++r) s += peer_slot(a.box[a.rank], half, a.half_bytes, a.max_floats, r)[i0 + i];
        a.data[i0 + i] = s;
    }
}

__global__ void peer_seq_advance_kernel(unsigned long long* seq) { *seq += 1ull; }

inline int64_t peer_chunks(int64_t max_floats) { return tipk_ceil_div(max_floats, PEER_CHUNK); }
inline int64_t peer_half_bytes(int world, int64_t max_floats) {
    return peer_slot_bytes(world, max_floats) + ((int64_t)world * peer_chunks(max_floats) * 8 + 255) / 256 * 256;
}
constexpr int64_t PEER_TAIL_BYTES = 256;       // behind the two halves: +0 the sequence counter, +64 the error word
unsigned long long g_peer_budget = 200000000ull;   // 2 s of s_memrealtime ticks

}  // namespace

extern "C" int64_t tipk_peer_mailbox_bytes(int world, int64_t max_floats) {
    if (world < 1 || world > PEER_MAX_WORLD || max_floats < 1) return 0;
    return 2 * peer_half_bytes(world, max_floats) + PEER_TAIL_BYTES;       // two halves + the sequence counter and the error word
}

extern "C" int tipk_peer_alloc(int64_t bytes, void** ptr) {
    if (!ptr || bytes <= 0) return TIPK_EINVAL;
    hipError_t e = hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return tipk_hip_status(e);
    e = hipMemset(*ptr, 0, (size_t)bytes);                                 // flags and the sequence counter start at 0
    if (e != hipSuccess) return tipk_hip_status(e);
    return tipk_hip_status(hipDeviceSynchronize());
}

extern "C" int tipk_peer_free(void* ptr) { return ptr ? tipk_hip_status(hipFree(ptr)) : TIPK_OK; }

extern "C" int tipk_ipc_get_handle(void* ptr, void* handle_out /* 64 bytes */) {
    if (!ptr || !handle_out) return TIPK_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    return tipk_hip_status(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), ptr));
}

extern "C" int tipk_ipc_open(const void* handle /* 64 bytes */, void** ptr) {
    if (!handle || !ptr) return TIPK_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    return tipk_hip_status(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}

extern "C" int tipk_ipc_close(void* ptr) { return ptr ? tipk_hip_status(hipIpcCloseMemHandle(ptr)) : TIPK_OK; }

extern "C" int tipk_peer_allreduce(float* data, int64_t n, void* const* mailboxes /* host [world] */, int rank, int world,
                                   int64_t max_floats, tipk_stream_t stream) {
    if (!data || !mailboxes || world < 1 || world > PEER_MAX_WORLD || rank < 0 || rank >= world || n < 0 || n > max_floats)
        return TIPK_EINVAL;
    if (n == 0) return TIPK_OK;
    PeerArgs a;
    a.data = data; a.n = n; a.rank = rank; a.world = world;
    a.max_floats = max_floats; a.half_bytes = peer_half_bytes(world, max_floats); a.n_chunks_max = peer_chunks(max_floats);
    for (int r = 0; r < PEER_MAX_WORLD; ++r) a.box[r] = r < world ? static_cast<char*>(mailboxes[r]) : nullptr;
    for (int r = 0; r < world; ++r) if (!a.box[r]) return TIPK_EINVAL;
    unsigned long long* seq = reinterpret_cast<unsigned long long*>(a.box[rank] + 2 * a.half_bytes);
    a.seq = seq;
    a.err = reinterpret_cast<unsigned long long*>(a.box[rank] + 2 * a.half_bytes + 64);
    a.budget = g_peer_budget;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3((unsigned)tipk_ceil_div(n, PEER_CHUNK)), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL(peer_seq_advance_kernel, dim3(1), dim3(1), 0, st, seq);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_peer_set_timeout_ms(int64_t ms) {
    if (ms < 1 || ms > 600000) return TIPK_EINVAL;
    g_peer_budget = (unsigned long long)ms * 100000ull;                    // s_memrealtime counts at 100 MHz
    return TIPK_OK;
}

// Error word of the own mailbox (0 = every wait so far was served): a SYNCHRONOUS 8-byte copy -- call it at a point where
// the host synchronises anyway (after a step's loss has been read), never inside a stream capture.
extern "C" int tipk_peer_status(void* mailbox, int world, int64_t max_floats, uint64_t* error_word) {
    if (!mailbox || !error_word || world < 1 || world > PEER_MAX_WORLD || max_floats < 1) return TIPK_EINVAL;
    const char* p = static_cast<const char*>(mailbox) + 2 * peer_half_bytes(world, max_floats) + 64;
    return tipk_hip_status(hipMemcpy(error_word, p, 8, hipMemcpyDeviceToHost));
}
