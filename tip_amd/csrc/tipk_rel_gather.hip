// Relation-local gather: the D-D aggregation when the node table of ONE relation fits in LDS
// (BioSNAP: 645 drugs x 32 floats = 82 KB).  include/tipk.h section 1b.
//
// The generic `gather_sum` streams every gathered row through the fabric (1.07 GB per launch for the
// 8.3 M D-D edges: 150 us at the HBM roofline).  Here a persistent 1024-thread workgroup per CU walks
// its relations; per relation it stages the table rows (Y_r, forward) once with coalesced loads --
// or keeps them for the whole launch (g', backward) -- plus the relation's run table, node order and
// 16-bit edge ids, and every gathered row is then a ds_read_b128 from LDS instead of an L2/fabric
// transaction.
//
// Inside a relation the output nodes are ordered by DECREASING run length (edges of one
// (relation, node) pair = one contiguous run), and slot s (L = columns/4 lanes) takes positions
// s, s+S, s+2S, ...: the 64/L slots of a wavefront always work on runs of nearly equal length, empty
// rows sit at the end, and no node is touched by two slots of one relation -> no atomics anywhere.
//   FWD: the run sum is added to a per-workgroup fp32 accumulator image in LDS (plain read-modify-
//        write, exclusive by construction); one partial slab per workgroup, combined in order by
//        tipk_sum_slabs.  When table + accumulators exceed the LDS the columns are processed in
//        `n_split` independent column blocks (blockIdx.y).
//   BWD: the run sum IS the output row (relation, node): written straight to dY.
// Results are bitwise reproducible (fixed order everywhere).
#include <stdlib.h>
#include "tipk_common.h"

namespace {

constexpr int RG_CHUNK = 16384;        // edge ids staged per pass (uint16: 32 KB)

struct RgArgs {
    const float* table; int64_t ld_t;
    int n_nodes, dc;                   // dc = columns handled by one column block
    const int32_t* wg_rel_ptr; const int32_t* wg_rels;
    const int64_t* rel_idx_off; const int32_t* rel_len;
    const uint16_t* idx; const int32_t* runs; const uint16_t* node_at;
    float* out; int64_t ld_out;
    int dbg;
};

template <int L, bool BWD>
__global__ __launch_bounds__(1024) void rel_gather_kernel(RgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = 1024 / L;                       // slots per workgroup
    const int t = threadIdx.x;
    const int n_nodes = a.n_nodes, dc = a.dc;
    const int ldt = dc + 4;                            // odd multiple of 16 B: ds_read_b128 spreads over banks
    const int q4 = dc >> 2;                            // float4 per row (== L)
    float* tab = lds;                                                   // [n_nodes + 1][dc+4], last row = 0
    float* accl = tab + (int64_t)(n_nodes + 1) * ldt;                                  // FWD: [n_nodes][dc]
    int32_t* run_l = reinterpret_cast<int32_t*>(accl + (BWD ? 0 : (int64_t)n_nodes * dc));     // [n_nodes][2]
    uint16_t* node_l = reinterpret_cast<uint16_t*>(run_l + ((2 * n_nodes + 3) & ~3));  // [n_nodes] (+pad)
    uint16_t* idx_l = node_l + ((n_nodes + 7) & ~7);                                   // [RG_CHUNK], 16-B aligned
    const int slot = t / L, sub = t & (L - 1), c0 = sub * 4;
    const int col0 = blockIdx.y * dc;                  // column block of this workgroup
    const float* table = a.table + col0;
    float* out = a.out + col0;

    // Staging is software-pipelined across relations: the NEXT relation's table rows, run table,
    // node order and first id chunk are requested into registers before the current relation is
    // processed and written to LDS afterwards, so HBM latency hides behind the LDS-bound compute
    // (one workgroup per CU: there is no other wave to hide it).
    constexpr int TU = 8;                              // float4 of table per thread (n_nodes*q4 <= 8192)
    float4 tv[TU];
    int rv[2];
    uint16_t nv = 0;
    uint4 iv[2];
    const int total4 = n_nodes * q4;
    auto prefetch = [&](int rel, bool with_table) {
        const int32_t* rsrc = a.runs + (int64_t)rel * n_nodes * 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; rv[u] = i < 2 * n_nodes ? rsrc[i] : 0; }
        nv = t < n_nodes ? a.node_at[(int64_t)rel * n_nodes + t] : (uint16_t)0;
        const int ne = a.rel_len[rel];
        const int n8 = ((ne < RG_CHUNK ? ne : RG_CHUNK) + 7) >> 3;
        const uint4* isrc = reinterpret_cast<const uint4*>(a.idx + a.rel_idx_off[rel]);
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < n8) iv[u] = isrc[i]; }
        if (with_table) {
            const float* src = table + (BWD ? 0 : (int64_t)rel * n_nodes * a.ld_t);
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int i = u * 1024 + t;
                if (i < total4) { const int r = i / q4, c = (i - r * q4) * 4; tv[u] = tipk_ld4(src + (int64_t)r * a.ld_t + c); }
            }
        }
    };
    auto commit = [&](int rel, bool with_table) {      // registers -> LDS
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < 2 * n_nodes) run_l[i] = rv[u]; }
        if (t < n_nodes) node_l[t] = nv;
        const int ne = a.rel_len[rel];
        const int n8 = ((ne < RG_CHUNK ? ne : RG_CHUNK) + 7) >> 3;
        uint4* idst = reinterpret_cast<uint4*>(idx_l);
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < n8) idst[i] = iv[u]; }
        if (with_table) {
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int i = u * 1024 + t;
                if (i < total4) { const int r = i / q4, c = (i - r * q4) * 4; tipk_st4(tab + r * ldt + c, tv[u]); }
            }
        }
    };
    if (t < ldt) tab[(int64_t)n_nodes * ldt + t] = 0.f;                    // the sentinel's row
    if (!BWD)
        for (int i = t; i < n_nodes * q4; i += 1024) tipk_st4(accl + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    const int wg = blockIdx.x;
    const int ri0 = a.wg_rel_ptr[wg], ri1 = a.wg_rel_ptr[wg + 1];
    if (ri0 < ri1) prefetch(a.wg_rels[ri0], true);     // BWD: g' is staged once, with the first relation
    for (int ri = ri0; ri < ri1; ++ri) {
        const int rel = a.wg_rels[ri];
        const int64_t e0 = a.rel_idx_off[rel];         // multiple of 8 ids: 16-byte aligned segment
        const int ne = a.rel_len[rel];
        __syncthreads();                               // readers of the previous relation are done
        commit(rel, !BWD || ri == ri0);
        __syncthreads();
        if (ri + 1 < ri1) prefetch(a.wg_rels[ri + 1], !BWD);      // in flight during the compute below
        for (int cb = 0; cb == 0 || cb < ne; cb += RG_CHUNK) {
            const int cn = ne - cb < RG_CHUNK ? ne - cb : RG_CHUNK;
            if (cb > 0) {                              // rare: a relation with more than RG_CHUNK ids
                __syncthreads();
                const uint4* isrc = reinterpret_cast<const uint4*>(a.idx + e0 + cb);
                uint4* idst = reinterpret_cast<uint4*>(idx_l);
                const int n8 = (cn + 7) >> 3;
                uint4 jv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < n8) jv[u] = isrc[i]; }
#pragma unroll
                for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < n8) idst[i] = jv[u]; }
                __syncthreads();
            }
            if (a.dbg & 1) continue;
            for (int p = slot; p < n_nodes; p += NS) {
                const int b = run_l[2 * p], len = run_l[2 * p + 1];
                if (!BWD && len == 0) break;           // rows are sorted by length: the rest is empty
                int lo = b > cb ? b : cb;                // b, len, cb, cn are multiples of 8
                int hi = b + len < cb + cn ? b + len : cb + cn;
                lo -= cb;
                hi -= cb;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                // 8 edges per step.  Runs are padded to multiples of 8 ids with the sentinel id
                // n_nodes, whose table row is all zeros: no masks, no tails -- a step is one 16-byte id
                // read (same address for the slot's lanes = broadcast) + 8 row reads + 8 adds.
                for (int eb = lo; eb < hi; eb += 8) {
                    const uint4 pk = *reinterpret_cast<const uint4*>(idx_l + eb);
                    const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
                    float4 v[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const unsigned idj = (w4[jj >> 1] >> (16 * (jj & 1))) & 0xffffu;
                        v[jj] = tipk_ld4(tab + idj * ldt + c0);
                    }
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        acc.x += v[jj].x; acc.y += v[jj].y; acc.z += v[jj].z; acc.w += v[jj].w;
                    }
                }
                const int node = node_l[p];
                if (BWD) {
                    float* o = out + ((int64_t)rel * n_nodes + node) * a.ld_out + c0;
                    if (cb > 0 && lo < hi) {           // a run continued from the previous id chunk
                        const float4 old = tipk_ld4(o);
                        acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
                    }
                    if (cb == 0 || lo < hi) tipk_st4(o, acc);
                } else if (lo < hi) {
                    float* o = accl + node * dc + c0;  // this slot is the only one touching `node` now
                    const float4 old = tipk_ld4(o);
                    acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
                    tipk_st4(o, acc);
                }
            }
        }
    }
    if (!BWD) {
        __syncthreads();
        float* o = out + (int64_t)wg * n_nodes * a.ld_out;
        for (int i = t; i < n_nodes * q4; i += 1024) {
            const int r = i / q4, c = (i - r * q4) * 4;
            tipk_st4(o + (int64_t)r * a.ld_out + c, tipk_ld4(accl + r * dc + c));
        }
    }
}

inline int64_t rel_gather_lds(int64_t n_nodes, int dc, bool bwd) {
    return (n_nodes + 1) * (dc + 4) * 4 + (bwd ? 0 : n_nodes * dc * 4) + ((2 * n_nodes + 3) & ~3LL) * 4 +
           ((n_nodes + 7) & ~7LL) * 2 + RG_CHUNK * 2;
}

constexpr int64_t RG_LDS_LIMIT = 158 * 1024;

// column blocks needed so that one block's table (+ accumulators) fits in LDS; 0 = impossible
inline int rel_gather_split(int64_t n_nodes, int d, bool bwd) {
    if (n_nodes <= 0 || n_nodes > 65534 || d < 4 || d > 256 || (d & (d - 1)) != 0) return 0;
    for (int split = 1; d / split >= 4; split *= 2) {
        const int dc = d / split;
        if (dc > 64) continue;                          // L = dc/4 <= 16 lanes per slot
        if (n_nodes * (dc / 4) > 8192) continue;       // table rows are prefetched in 8 float4 per thread
        if (rel_gather_lds(n_nodes, dc, bwd) <= RG_LDS_LIMIT) return split;
    }
    return 0;
}

template <int L>
int launch_rg(bool bwd, const RgArgs& a, int n_wg, int split, hipStream_t st) {
    const size_t lds = (size_t)rel_gather_lds(a.n_nodes, a.dc, bwd);
    hipError_t e;
    dim3 grid((unsigned)n_wg, (unsigned)split);
    if (bwd) {
        auto kern = rel_gather_kernel<L, true>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return tipk_hip_status(e);
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, a);
    } else {
        auto kern = rel_gather_kernel<L, false>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return tipk_hip_status(e);
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, a);
    }
    TIPK_RETURN_LAUNCH();
}

}  // namespace

extern "C" int tipk_rel_gather_supported(int64_t n_nodes, int d, int backward) {
    return rel_gather_split(n_nodes, d, backward != 0) > 0 ? 1 : 0;
}

extern "C" int tipk_rel_gather(int backward, const float* table, int64_t ld_table, int64_t n_nodes, int d,
                               int64_t n_wg, const int32_t* wg_rel_ptr, const int32_t* wg_rels,
                               const int64_t* rel_idx_off, const int32_t* rel_len, const uint16_t* idx,
                               const int32_t* runs, const uint16_t* node_at, float* out, int64_t ld_out,
                               tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table || !wg_rel_ptr || !wg_rels || !rel_idx_off || !rel_len || !idx || !runs ||
        !node_at || !out || (reinterpret_cast<uintptr_t>(idx) & 15))
        return TIPK_EINVAL;
    const int split = rel_gather_split(n_nodes, d, backward != 0);
    if (split == 0) return TIPK_EUNSUPPORTED;
    if (ld_table % 4 != 0 || ld_out % 4 != 0 || (reinterpret_cast<uintptr_t>(table) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return TIPK_EINVAL;
    const char* dbg_env = getenv("TIPK_RG_DEBUG");
    RgArgs a;
    a.table = table; a.ld_t = ld_table; a.n_nodes = (int)n_nodes; a.dc = d / split;
    a.wg_rel_ptr = wg_rel_ptr; a.wg_rels = wg_rels; a.rel_idx_off = rel_idx_off; a.rel_len = rel_len;
    a.idx = idx; a.runs = runs; a.node_at = node_at; a.out = out; a.ld_out = ld_out;
    a.dbg = dbg_env ? atoi(dbg_env) : 0;
    hipStream_t st = (hipStream_t)stream;
    switch (a.dc / 4) {
        case 1: return launch_rg<1>(backward != 0, a, (int)n_wg, split, st);
        case 2: return launch_rg<2>(backward != 0, a, (int)n_wg, split, st);
        case 4: return launch_rg<4>(backward != 0, a, (int)n_wg, split, st);
        case 8: return launch_rg<8>(backward != 0, a, (int)n_wg, split, st);
        default: return launch_rg<16>(backward != 0, a, (int)n_wg, split, st);
    }
}
