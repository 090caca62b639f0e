// Train / test split of the D-D graph on device (include/tipk.h section 7; reference
// src/utils.py:35-65 `process_edges`, data/utils.py:212-229 `process_prot_edge`, done there on the host
// with numpy's global Mersenne state, one relation at a time, through Python lists).
//
//   pair i of the concatenated undirected pair list is a TRAINING pair iff
//       Philox4x32-10(counter = (i lo, i hi, 0, 'SPLT'), key = seed).x0  <  floor(p * 2^32)
//   (bit-exact specification: oracle/philox_split.py).  Per relation r the kept pairs, in list order,
//   form the block [ (u,v) ... | (v,u) ... ] of the training tensors -- the layout
//   `to_bidirection` gives them (src/utils.py:17-23, :53) -- and the rest the same block of the test
//   tensors; `edge_type` is the relation id.  Two launches: flags + per-relation counts (integer
//   atomics: exact), then an ordered in-workgroup compaction per relation.  Deterministic.
#include "tipk_common.h"

namespace {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;

__device__ __forceinline__ uint32_t philox_x0(uint64_t ctr, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(PHILOX_M0, c0), lo0 = PHILOX_M0 * c0;
        const uint32_t hi1 = __umulhi(PHILOX_M1, c2), lo1 = PHILOX_M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    return c0;
}

__global__ __launch_bounds__(256) void split_flags_kernel(const int64_t* __restrict__ rel_ptr, int64_t n_rel,
                                                          uint32_t threshold, uint32_t all, uint64_t seed,
                                                          uint8_t* __restrict__ take,
                                                          unsigned long long* __restrict__ n_train) {
    const int64_t total = rel_ptr[n_rel];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t x = philox_x0((uint64_t)i, 0u, 0x53504C54u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const bool t = all || x < threshold;
    take[i] = t ? 1 : 0;
    if (!t) return;
    int64_t lo = 0, hi = n_rel;                              // relation of pair i
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rel_ptr[mid] <= i) lo = mid; else hi = mid;
    }
    atomicAdd(n_train + lo, 1ull);
}

// one workgroup per relation: ordered compaction of its pairs into the two mirrored blocks
template <typename IT>
__global__ __launch_bounds__(256) void split_scatter_kernel(const IT* __restrict__ pu, const IT* __restrict__ pv,
                                                            const int64_t* __restrict__ rel_ptr,
                                                            const uint8_t* __restrict__ take,
                                                            const int64_t* __restrict__ train_ptr,
                                                            const int64_t* __restrict__ test_ptr,
                                                            int64_t* __restrict__ tr_u, int64_t* __restrict__ tr_v,
                                                            int64_t* __restrict__ tr_et, int64_t* __restrict__ te_u,
                                                            int64_t* __restrict__ te_v, int64_t* __restrict__ te_et) {
    __shared__ int wave_sum[4];
    __shared__ int base_tr, base_te;
    const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t a = rel_ptr[r], b = rel_ptr[r + 1];
    const int64_t tr0 = train_ptr[r], ntr = (train_ptr[r + 1] - tr0) / 2;   // directed blocks: [kept | mirrored]
    const int64_t te0 = test_ptr[r], nte = (test_ptr[r + 1] - te0) / 2;
    if (t == 0) { base_tr = 0; base_te = 0; }
    __syncthreads();
    for (int64_t c = a; c < b; c += 256) {
        const int64_t i = c + t;
        const bool in = i < b;
        const bool k = in && take[i];
        const unsigned long long m = __ballot(k);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_sum[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) woff += wave_sum[w]; tot += wave_sum[w]; }
        const int btr = base_tr, bte = base_te;
        const int n_in = (int)((b - c) < 256 ? (b - c) : 256);
        if (in) {
            const int64_t u = (int64_t)pu[i], v = (int64_t)pv[i];
            if (k) {
                const int64_t q = tr0 + btr + woff + before;
                tr_u[q] = u; tr_v[q] = v; tr_et[q] = r;
                tr_u[q + ntr] = v; tr_v[q + ntr] = u; tr_et[q + ntr] = r;
            } else {
                const int64_t q = te0 + bte + (t - (woff + before));         // rank among the dropped = index - kept before
                te_u[q] = u; te_v[q] = v; te_et[q] = r;
                te_u[q + nte] = v; te_v[q + nte] = u; te_et[q + nte] = r;
            }
        }
        __syncthreads();
        if (t == 0) { base_tr = btr + tot; base_te = bte + (n_in - tot); }
        __syncthreads();
    }
}

}  // namespace

extern "C" int tipk_split_flags(const int64_t* rel_ptr, int64_t n_rel, int64_t n_pairs, double p_train, uint64_t seed,
                                uint8_t* take, uint64_t* n_train, tipk_stream_t stream) {
    if (!rel_ptr || !take || !n_train || n_rel < 0 || n_pairs < 0 || !(p_train >= 0.0 && p_train <= 1.0)) return TIPK_EINVAL;
    if (n_pairs == 0) return TIPK_OK;
    const double scaled = p_train * 4294967296.0;
    const uint32_t all = scaled >= 4294967296.0 ? 1u : 0u;
    const uint32_t threshold = all ? 0xffffffffu : (uint32_t)scaled;          // floor(p * 2^32)
    const int64_t blocks = tipk_ceil_div(n_pairs, 256);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(split_flags_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rel_ptr, n_rel,
                       threshold, all, seed, take, reinterpret_cast<unsigned long long*>(n_train));
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_split_scatter(const void* pairs_u, const void* pairs_v, int idx_bytes, const int64_t* rel_ptr,
                                  int64_t n_rel, const uint8_t* take, const int64_t* train_ptr, const int64_t* test_ptr,
                                  int64_t* train_u, int64_t* train_v, int64_t* train_et, int64_t* test_u, int64_t* test_v,
                                  int64_t* test_et, tipk_stream_t stream) {
    if (!pairs_u || !pairs_v || !rel_ptr || !take || !train_ptr || !test_ptr || n_rel < 0 || n_rel > 0x7fffffffLL)
        return TIPK_EINVAL;
    if (n_rel == 0) return TIPK_OK;
    hipStream_t st = (hipStream_t)stream;
    if (idx_bytes == 8)
        hipLaunchKernelGGL(split_scatter_kernel<int64_t>, dim3((unsigned)n_rel), dim3(256), 0, st, (const int64_t*)pairs_u,
                           (const int64_t*)pairs_v, rel_ptr, take, train_ptr, test_ptr, train_u, train_v, train_et, test_u,
                           test_v, test_et);
    else if (idx_bytes == 2)
        hipLaunchKernelGGL(split_scatter_kernel<uint16_t>, dim3((unsigned)n_rel), dim3(256), 0, st, (const uint16_t*)pairs_u,
                           (const uint16_t*)pairs_v, rel_ptr, take, train_ptr, test_ptr, train_u, train_v, train_et, test_u,
                           test_v, test_et);
    else
        return TIPK_EINVAL;
    TIPK_RETURN_LAUNCH();
}
