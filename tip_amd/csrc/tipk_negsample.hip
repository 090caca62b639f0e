// Typed negative sampling on device (include/tipk.h section 5; reference src/neg_sampling.py:5-26).
// Counter-based Philox4x32-10 (no state, replayable); a candidate pair is uniform on [0, n^2) and rejected while it is a
// positive of the SAME relation.  n^2 < 2^32 (round 5): one Philox call serves FOUR positions, a candidate is the high
// word of x * n^2 with Lemire's rejection of the low word (exactly uniform); larger node sets: one 64-bit candidate per
// call, mulhi64(random64, n^2).  Bit-exact specification: oracle/philox_sampler.py.
#include <type_traits>
#include "tipk_common.h"

namespace {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;
constexpr int MAX_ATTEMPTS = 64;

// 32 x 32 -> 64-bit product in ONE quarter-rate instruction (hipcc splits the product into v_mul_hi_u32 + v_mul_lo_u32,
// two of them: the multiplies are two thirds of the sampler's issue slots)
__device__ __forceinline__ uint64_t mul_wide(uint32_t m, uint32_t x) {
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "s"(m), "v"(x) : "vcc");
    return r;
}

struct Philox4 { uint32_t x[4]; };

__device__ __forceinline__ Philox4 philox4(uint64_t ctr, uint32_t attempt, uint32_t k0, uint32_t k1) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = attempt, c3 = 0u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = mul_wide(PHILOX_M0, c0), p1 = mul_wide(PHILOX_M1, c2);
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    Philox4 r;
    r.x[0] = c0; r.x[1] = c1; r.x[2] = c2; r.x[3] = c3;
    return r;
}

__device__ __forceinline__ uint64_t philox64(uint64_t ctr, uint32_t attempt, uint32_t k0, uint32_t k1) {
    const Philox4 r = philox4(ctr, attempt, k0, k1);
    return (uint64_t)r.x[0] | ((uint64_t)r.x[1] << 32);
}

// word (c & 3) of the Philox call of counter c >> 2 (constant indices: no scratch array)
__device__ __forceinline__ uint32_t philox_word(uint64_t c, uint32_t attempt, uint32_t k0, uint32_t k1) {
    const Philox4 r = philox4(c >> 2, attempt, k0, k1);
    const uint32_t w = (uint32_t)c & 3u;
    return w == 0u ? r.x[0] : (w == 1u ? r.x[1] : (w == 2u ? r.x[2] : r.x[3]));
}

// Philox key of call n of a sampler stream: splitmix64(seed + (n + 1) * golden) -- the same function
// tip_amd/neg_sampling.py applies on the host; with a device-resident call counter a captured
// hipGraph draws fresh negatives on every replay.
__device__ __forceinline__ uint64_t call_key(uint64_t seed, uint64_t n) {
    uint64_t z = seed + (n + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void counter_advance_kernel(uint64_t* counter) { *counter += 1; }

// In-kernel advance of a sampler stream (state[0] = position, state[2] = ticket): the LAST workgroup to finish moves
// the position on.  Every workgroup derived its key from `call_no` (a use: the load has completed) before it gets here,
// so the store cannot be seen by this launch; the next launch on the stream sees it.
__device__ __forceinline__ void stream_advance(uint64_t* state, int advance, uint64_t call_no) {
    if (!advance || !state) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* s = reinterpret_cast<unsigned long long*>(state);
        if (atomicAdd(s + 2, 1ull) == (unsigned long long)gridDim.x - 1ull) {
            s[2] = 0ull;
            s[0] = call_no + 1ull;
        }
    }
}

// (u, v) = (cand / n, cand % n).  n^2 < 2^32 for every graph whose ids fit 16 bits: one 32-bit division (a 64-bit
// division by a run-time value is ~100 instructions per position) -- the same integers either way.
struct PackedOut { uint32_t w; };       // idx_bytes = 2: the pair as ONE 32-bit word u | v << 16 (n_nodes <= 65535), out_v unused

template <typename OT>
__device__ __forceinline__ void split_pair(uint64_t cand, int64_t n_nodes, OT* u, OT* v) {
    if constexpr (std::is_same<OT, PackedOut>::value) {
        const uint32_t c = (uint32_t)cand, n = (uint32_t)n_nodes;
        const uint32_t q = c / n;
        u->w = q | ((c - q * n) << 16);
    } else if (n_nodes <= 65535) {
        const uint32_t c = (uint32_t)cand, n = (uint32_t)n_nodes;
        const uint32_t q = c / n;
        *u = (OT)q;
        *v = (OT)(c - q * n);
    } else {
        *u = (OT)(cand / (uint64_t)n_nodes);
        *v = (OT)(cand % (uint64_t)n_nodes);
    }
}

template <typename OT>
__global__ __launch_bounds__(256) void neg_sample_kernel(const int64_t* __restrict__ keys,
                                                         const int64_t* __restrict__ rel_ptr, int64_t n_rel,
                                                         int64_t n_nodes, uint64_t seed,
                                                         uint64_t* __restrict__ call_counter, int advance,
                                                         const int64_t* __restrict__ pos_offset,
                                                         OT* __restrict__ out_u, OT* __restrict__ out_v) {
    const uint64_t call_no = call_counter ? call_counter[0] : 0ull;
    const uint64_t key = call_counter ? call_key(call_counter[1], call_no) : seed;
    const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    const int64_t total = rel_ptr[n_rel];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) {
    // relation of position e: largest r with rel_ptr[r] <= e
    int64_t lo = 0, hi = n_rel;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (rel_ptr[mid] <= e) lo = mid; else hi = mid;
    }
    const int64_t a = rel_ptr[lo], b = rel_ptr[lo + 1];
    const uint64_t nn = (uint64_t)n_nodes * (uint64_t)n_nodes;
    const uint64_t ctr = (uint64_t)(e + (pos_offset ? pos_offset[lo] : 0));      // the position's number in the WHOLE triple list
    const bool narrow = nn < (1ull << 32);                                       // four positions per Philox call (spec)
    const uint32_t nn32 = (uint32_t)nn, thresh = narrow ? (uint32_t)(((1ull << 32) - nn) % nn) : 0u;
    uint64_t cand = 0;
    for (int attempt = 0; attempt < MAX_ATTEMPTS; ++attempt) {
        if (narrow) {
            const uint64_t m = mul_wide(nn32, philox_word(ctr, (uint32_t)attempt, k0, k1));
            cand = m >> 32;
            if ((uint32_t)m < thresh && attempt + 1 < MAX_ATTEMPTS) continue;
        } else {
            cand = __umul64hi(philox64(ctr, (uint32_t)attempt, k0, k1), nn);
        }
        int64_t l = a, h = b;                       // lower_bound(keys[a:b], cand)
        while (l < h) {
            const int64_t mid = (l + h) >> 1;
            if ((uint64_t)keys[mid] < cand) l = mid + 1; else h = mid;
        }
        if (l == b || (uint64_t)keys[l] != cand) break;
    }
    split_pair(cand, n_nodes, out_u + e, out_v + e);
    }
    stream_advance(call_counter, advance, call_no);
}

// Bitmap variant (n_nodes^2 bits fit in LDS: BioSNAP 645^2 bits = 52 KB): a persistent 1024-thread
// workgroup per CU walks its relations (edge-balanced deal, as in tipk_rel_gather); per relation it
// sets one bit per positive pair (ds_or_b32: integer LDS atomics run at 5 lane-ops/clk/CU) and every
// candidate is then tested with ONE LDS read instead of a 16-step binary search through L2.
// Same candidates, same acceptance rule, same output as `neg_sample_kernel` (bit-exact).
constexpr int NS_QCAP = 4096;                             // rejected positions a workgroup parks per unit (LDS queue)
constexpr int NS_PRE = 8;                                 // keys per lane requested a unit ahead

template <typename OT>
__global__ __launch_bounds__(1024) void neg_sample_bitmap_kernel(
    const int64_t* __restrict__ keys, const int64_t* __restrict__ rel_ptr, const int32_t* __restrict__ wg_unit_ptr,
    const int32_t* __restrict__ wg_units, int64_t n_nodes, uint64_t seed, uint64_t* __restrict__ call_counter, int advance,
    const int64_t* __restrict__ pos_offset, const uint32_t* __restrict__ keys32, OT* __restrict__ out_u, OT* __restrict__ out_v,
    int dbg) {
    extern __shared__ unsigned bm[];                       // the bitmap | the retry queue [NS_QCAP] | its length
    const uint64_t call_no = call_counter ? call_counter[0] : 0ull;
    const uint64_t key = call_counter ? call_key(call_counter[1], call_no) : seed;
    const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    const uint64_t nn = (uint64_t)n_nodes * (uint64_t)n_nodes;
    const int words = (int)((nn + 31) >> 5);
    unsigned* queue = bm + words;
    unsigned* qlen = queue + NS_QCAP;
    const uint32_t nn32 = (uint32_t)nn, n32 = (uint32_t)n_nodes;
    const uint32_t thresh = (uint32_t)(((1ull << 32) - nn) % nn);
    const float inv_n = 1.0f / (float)n32;
    const int t = threadIdx.x;
    const int NT = (int)blockDim.x;                        // 1024, or 512 when two workgroups (two bitmaps) fit one CU
    int have = -1;                                         // relation whose bitmap is in LDS
    const int u_first = wg_unit_ptr[blockIdx.x], u_end = wg_unit_ptr[blockIdx.x + 1];
    if (t == 0) *qlen = 0u;
    __syncthreads();
    auto store_pair = [&](int64_t e, uint32_t cand) {
        uint32_t q = (uint32_t)((float)cand * inv_n);
        int32_t r = (int32_t)(cand - __umul24(q, n32));
        if (r < 0) { --q; r += (int32_t)n32; }
        else if (r >= (int32_t)n32) { ++q; r -= (int32_t)n32; }
        if constexpr (std::is_same<OT, PackedOut>::value) out_u[e].w = q | ((uint32_t)r << 16);
        else { out_u[e] = (OT)q; out_v[e] = (OT)r; }
    };
    // A unit's descriptor, its relation's range and the first NS_PRE keys of every lane are requested ONE UNIT AHEAD: the
    // bitmap build of a unit then starts from registers instead of three dependent round trips (unit -> range -> keys) that
    // nothing overlapped (round 6: half of the launch was the builds)
    struct Unit { int rel; int64_t ub, ue, a, b, off; uint32_t kk[NS_PRE]; };
    auto fetch_unit = [&](int ui, Unit& u) {
        const int uc = ui < u_end ? ui : u_end - 1;        // clamped, unconditional
        u.rel = wg_units[3 * uc];
        u.ub = wg_units[3 * uc + 1]; u.ue = wg_units[3 * uc + 2];
        u.a = rel_ptr[u.rel]; u.b = rel_ptr[u.rel + 1];
        u.off = pos_offset ? pos_offset[u.rel] : 0;
        if (keys32) {
#pragma unroll
            for (int j = 0; j < NS_PRE; ++j) {
                int64_t e = u.a + j * NT + t;
                e = e < u.b ? e : u.b - 1;
                u.kk[j] = keys32[e];
            }
        }
    };
    Unit cur, nxt;
    if (u_first < u_end) fetch_unit(u_first, cur);
    for (int ui = u_first; ui < u_end; ++ui) {
        // unit = (relation, first position, end position): a relation larger than the per-workgroup share is cut into
        // several units (every one of them builds the relation's bitmap; BioSNAP's largest relation alone is 1.6 x
        // the mean load of a workgroup)
        const int rel = cur.rel;
        const int64_t ub = cur.ub, ue = cur.ue, a = cur.a, b = cur.b, off = cur.off;
        if (rel != have && !TIPK_DBG(dbg & 2)) {           // (debug builds: 2 = no bitmap, 1 = no draws, 4 = no clear / barriers)
            __syncthreads();                               // the previous relation's tests are done
            for (int i = t; i < words; i += NT) bm[i] = 0u;
            __syncthreads();
            if (keys32) {                                  // (4 bytes per positive instead of 8)
#pragma unroll
                for (int j = 0; j < NS_PRE; ++j)           // predicated atomics (a thousand lanes on ONE word serialise)
                    if (a + j * NT + t < b) atomicOr(&bm[cur.kk[j] >> 5], 1u << (cur.kk[j] & 31));
                for (int64_t e0 = a + (int64_t)NS_PRE * NT; e0 < b; e0 += 8 * NT) {      // relations beyond NS_PRE x NT positives
                    uint32_t kk[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        int64_t e = e0 + j * NT + t;
                        e = e < b ? e : b - 1;
                        kk[j] = keys32[e];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (e0 + j * NT + t < b) atomicOr(&bm[kk[j] >> 5], 1u << (kk[j] & 31));
                }
            } else {
                for (int64_t e = a + t; e < b; e += NT) {
                    const uint64_t k = (uint64_t)keys[e];
                    atomicOr(&bm[k >> 5], 1u << (k & 31));
                }
            }
            __syncthreads();
            have = rel;
        }
        fetch_unit(ui + 1, nxt);                           // travels during this unit's draws
        if (!TIPK_DBG(dbg & 1)) {
        // A lane takes the FOUR positions of one Philox counter (c >> 2 = Q): one call of 20 wide multiplies draws all four
        // first attempts.  n^2 < 2^24 here (the bitmap fits LDS): a candidate is the high word of x * n^2, rejected (exact
        // uniformity: Lemire) when the low word is below (2^32 - n^2) mod n^2, or when its bit is set; (u, v) = (cand / n,
        // cand % n) is an exact float quotient with a one-step correction.
        // A REJECTED position (1.8 % of them at BioSNAP) is parked in an LDS queue and drawn again afterwards, a lane per
        // parked position: re-drawing in place made 69 % of the WAVES repeat the Philox call for each of the four positions
        // (round 5: five calls per four positions), per attempt level still one more call per lane for 7 % of the lanes.
        // Attempt k of a position is word (c & 3) of philox4(c >> 2, k) wherever it is drawn: bit-exact with the spec.
        const int64_t q_first = (ub + off) >> 2, q_last = (ue - 1 + off) >> 2;
        const bool vec_ok = (off & 3) == 0 && (reinterpret_cast<uintptr_t>(out_u) & 15) == 0;
        for (int64_t Q = q_first + t; Q <= q_last; Q += NT) {
            const Philox4 rw = philox4((uint64_t)Q, 0u, k0, k1);
            uint32_t word[4];
            bool done[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t e = 4 * Q + j - off;
                done[j] = false;
                word[j] = 0u;
                if (e < ub || e >= ue) continue;
                uint64_t m = mul_wide(nn32, rw.x[j]);
                uint32_t cand = (uint32_t)(m >> 32);
                if ((uint32_t)m < thresh || ((bm[cand >> 5] >> (cand & 31)) & 1u)) {
                    const unsigned slot = atomicAdd(qlen, 1u);
                    if (slot < (unsigned)NS_QCAP) { queue[slot] = (unsigned)(e - ub); continue; }
                    for (int attempt = 1; attempt < MAX_ATTEMPTS; ++attempt) {        // queue full: in place (same words)
                        m = mul_wide(nn32, philox_word((uint64_t)(4 * Q + j), (uint32_t)attempt, k0, k1));
                        cand = (uint32_t)(m >> 32);
                        if ((uint32_t)m >= thresh && !((bm[cand >> 5] >> (cand & 31)) & 1u)) break;
                    }
                }
                if constexpr (std::is_same<OT, PackedOut>::value) {
                    uint32_t q = (uint32_t)((float)cand * inv_n);
                    int32_t r = (int32_t)(cand - __umul24(q, n32));
                    if (r < 0) { --q; r += (int32_t)n32; }
                    else if (r >= (int32_t)n32) { ++q; r -= (int32_t)n32; }
                    word[j] = q | ((uint32_t)r << 16);
                    done[j] = true;
                } else {
                    store_pair(e, cand);
                }
            }
            if constexpr (std::is_same<OT, PackedOut>::value) {
                // the lane's four words as ONE 16-byte store when all four exist and the slot is aligned (four dword stores a
                // lane make every 64-byte line arrive in four pieces)
                const int64_t e0 = 4 * Q - off;
                if (done[0] && done[1] && done[2] && done[3] && vec_ok) {
                    *reinterpret_cast<uint4*>(&out_u[e0].w) = make_uint4(word[0], word[1], word[2], word[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (done[j]) out_u[e0 + j].w = word[j];
                }
            }
        }
        __syncthreads();                                   // the queue is complete
        const unsigned parked = *qlen < (unsigned)NS_QCAP ? *qlen : (unsigned)NS_QCAP;
        for (unsigned i = t; i < parked; i += NT) {
            const int64_t e = ub + queue[i];
            const uint64_t c = (uint64_t)(e + off);
            uint32_t cand = 0;
            for (int attempt = 1; attempt < MAX_ATTEMPTS; ++attempt) {
                const uint64_t m = mul_wide(nn32, philox_word(c, (uint32_t)attempt, k0, k1));
                cand = (uint32_t)(m >> 32);
                if ((uint32_t)m >= thresh && !((bm[cand >> 5] >> (cand & 31)) & 1u)) break;
            }
            store_pair(e, cand);
        }
        __syncthreads();                                   // (the queue has been read)
        if (t == 0) *qlen = 0u;
        }
        cur = nxt;
    }
    stream_advance(call_counter, advance, call_no);
}

}  // namespace

extern "C" int tipk_counter_advance(uint64_t* counter, tipk_stream_t stream) {
    if (!counter) return TIPK_EINVAL;
    hipLaunchKernelGGL(counter_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
    TIPK_RETURN_LAUNCH();
}

extern "C" int tipk_negsample_wgs_per_cu(int64_t n_nodes) {
    if (n_nodes <= 0 || n_nodes > 4095) return 0;
    const int64_t bm_bytes = ((n_nodes * n_nodes + 31) / 32) * 4;
    if (bm_bytes + (NS_QCAP + 4) * 4 > 158 * 1024) return 0;
    return 2 * (bm_bytes + (NS_QCAP + 4) * 4) <= 156 * 1024 ? 2 : 1;
}

extern "C" int tipk_typed_negative_sampling(const int64_t* pos_key_sorted, const int64_t* rel_ptr, int64_t n_rel,
                                            int64_t n_nodes, uint64_t seed, uint64_t* call_counter, int advance,
                                            const int32_t* wg_unit_ptr, const int32_t* wg_units, int64_t n_wg,
                                            const int64_t* pos_offset, const uint32_t* pos_key32,
                                            void* out_u, void* out_v, int idx_bytes, int64_t n_positions,
                                            tipk_stream_t stream) {
    if (n_rel < 0 || n_nodes <= 0 || n_positions < 0 || n_nodes > 0xffffffffLL) return TIPK_EINVAL;
    if (n_positions == 0 || n_rel == 0) return TIPK_OK;
    if (!pos_key_sorted || !rel_ptr || !out_u || (!out_v && idx_bytes != 2)) return TIPK_EINVAL;
    const int64_t blocks = tipk_ceil_div(n_positions, 256);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t bm_bytes = (((int64_t)n_nodes * n_nodes + 31) / 32) * 4;
    const int dbg = TIPK_DBG(tipk_option(TIPK_OPT_DM_DEBUG));
    const int64_t lds_bytes = bm_bytes + (NS_QCAP + 4) * 4;                    // + the retry queue and its length
    // two 512-thread workgroups per CU when two bitmaps fit its LDS: one builds its relation's bitmap (key loads, barriers) while
    // the other draws -- a single 1024-thread workgroup per CU left the CU idle through every build (round 6)
    const unsigned nt = tipk_negsample_wgs_per_cu(n_nodes) == 2 ? 512u : 1024u;
    if (wg_unit_ptr && wg_units && n_wg > 0 && n_wg <= 65535 && lds_bytes <= 158 * 1024 && n_nodes <= 4095) {
        if (idx_bytes == 8) {
            auto kern = neg_sample_bitmap_kernel<int64_t>;
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return tipk_hip_status(e);
            hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(nt), (size_t)lds_bytes, st, pos_key_sorted, rel_ptr,
                               wg_unit_ptr, wg_units, n_nodes, seed, call_counter, advance, pos_offset, pos_key32, (int64_t*)out_u, (int64_t*)out_v, dbg);
        } else if (idx_bytes == 4) {
            auto kern = neg_sample_bitmap_kernel<int32_t>;
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return tipk_hip_status(e);
            hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(nt), (size_t)lds_bytes, st, pos_key_sorted, rel_ptr,
                               wg_unit_ptr, wg_units, n_nodes, seed, call_counter, advance, pos_offset, pos_key32, (int32_t*)out_u, (int32_t*)out_v, dbg);
        } else if (idx_bytes == 2 && n_nodes <= 65535) {
            auto kern = neg_sample_bitmap_kernel<PackedOut>;
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return tipk_hip_status(e);
            hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(nt), (size_t)lds_bytes, st, pos_key_sorted, rel_ptr,
                               wg_unit_ptr, wg_units, n_nodes, seed, call_counter, advance, pos_offset, pos_key32, (PackedOut*)out_u, (PackedOut*)out_u, dbg);
        } else {
            return TIPK_EINVAL;
        }
        TIPK_RETURN_LAUNCH();
    }
    if (idx_bytes == 8)
        hipLaunchKernelGGL(neg_sample_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), 0, st, pos_key_sorted,
                           rel_ptr, n_rel, n_nodes, seed, call_counter, advance, pos_offset, (int64_t*)out_u, (int64_t*)out_v);
    else if (idx_bytes == 4)
        hipLaunchKernelGGL(neg_sample_kernel<int32_t>, dim3((unsigned)blocks), dim3(256), 0, st, pos_key_sorted,
                           rel_ptr, n_rel, n_nodes, seed, call_counter, advance, pos_offset, (int32_t*)out_u, (int32_t*)out_v);
    else if (idx_bytes == 2 && n_nodes <= 65535)
        hipLaunchKernelGGL(neg_sample_kernel<PackedOut>, dim3((unsigned)blocks), dim3(256), 0, st, pos_key_sorted,
                           rel_ptr, n_rel, n_nodes, seed, call_counter, advance, pos_offset, (PackedOut*)out_u, (PackedOut*)out_u);
    else
        return TIPK_EINVAL;
    TIPK_RETURN_LAUNCH();
}
