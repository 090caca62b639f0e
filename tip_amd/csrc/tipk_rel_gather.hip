// Relation-local gather: the D-D aggregation when the node table of ONE relation fits in LDS
// (BioSNAP: 645 drugs x 32 floats = 82 KB).  include/tipk.h section 1b.
//
// The generic `gather_sum` streams every gathered row through the fabric (1.07 GB per launch for the
// 8.3 M D-D edges: 150 us at the HBM roofline).  Here a persistent 1024-thread workgroup per CU walks
// its relations; per relation it stages the table rows (Y_r, forward) once with coalesced loads --
// or keeps them for the whole launch (g', backward) -- plus the relation's run table and 16-bit edge
// ids, and every gathered row is then a ds_read_b128 from LDS (256 B/clk/CU) instead of an L2/fabric
// transaction.  A slot of L = d/4 lanes OWNS a fixed set of output nodes (positions slot, slot+S,
// ... of a degree-sorted node order, so heavy and light nodes are dealt evenly) and keeps their sums
// in registers: no atomics, no LDS accumulators, bitwise reproducible.
//   FWD: sums persist across the workgroup's relations -> one partial [N x d] slab per workgroup,
//        combined in order by tipk_sum_slabs.
//   BWD: one output row per (relation, node) written straight to dY; sums reset per relation.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

constexpr int RG_CHUNK = 16384;        // edge ids staged per pass (uint16: 32 KB)

template <int L, int J, bool BWD>
__global__ __launch_bounds__(1024) void rel_gather_kernel(
    const float* __restrict__ table, int64_t ld_t, int n_nodes, int d, const int32_t* __restrict__ wg_rel_ptr,
    const int32_t* __restrict__ wg_rels, const int64_t* __restrict__ rel_idx_off, const int32_t* __restrict__ rel_len,
    const uint16_t* __restrict__ idx, const int32_t* __restrict__ runs, const int32_t* __restrict__ node_at,
    float* __restrict__ out, int64_t ld_out, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = 1024 / L;                       // slots per workgroup
    const int t = threadIdx.x;
    const int ldt = d + 4;                             // odd multiple of 16 B: ds_read_b128 spreads over banks
    float* tab = lds;                                  // [n_nodes][d+4]
    int32_t* run_l = reinterpret_cast<int32_t*>(tab + (int64_t)n_nodes * ldt);      // [n_nodes][2] (+pad to 16 B)
    uint16_t* idx_l = reinterpret_cast<uint16_t*>(run_l + ((2 * n_nodes + 3) & ~3)); // [RG_CHUNK], 16-byte aligned
    const int slot = t / L, sub = t & (L - 1), c0 = sub * 4;
    const int q4 = d >> 2;                             // float4 per row
    const bool col_ok = c0 < d;

    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);

    // Staging loops issue a whole batch of global loads before the first LDS write: a plain
    // load->ds_write loop waits for every load in turn (one HBM latency per iteration).
    auto stage_table = [&](const float* src) {
        constexpr int U = 8;
        const int total = n_nodes * q4;
        for (int base = 0; base < total; base += 1024 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 1024 + t;
                if (i < total) { const int r = i / q4, c = (i - r * q4) * 4; v[u] = tipk_ld4(src + (int64_t)r * ld_t + c); }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 1024 + t;
                if (i < total) { const int r = i / q4, c = (i - r * q4) * 4; tipk_st4(tab + r * ldt + c, v[u]); }
            }
        }
    };
    if (BWD) stage_table(table);                       // g' stays resident for the whole launch
    const int wg = blockIdx.x;
    for (int ri = wg_rel_ptr[wg]; ri < wg_rel_ptr[wg + 1]; ++ri) {
        const int rel = wg_rels[ri];
        const int64_t e0 = rel_idx_off[rel];           // multiple of 8 ids: 16-byte aligned segment
        const int ne = rel_len[rel];
        __syncthreads();                               // readers of the previous relation are done
        {   // run table (2*n_nodes ints) and the first chunk of ids are requested together with Y_r
            const int32_t* rsrc = runs + (int64_t)rel * n_nodes * 2;
            int rv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; rv[u] = i < 2 * n_nodes ? rsrc[i] : 0; }
            if (!BWD && !(dbg & 2)) stage_table(table + (int64_t)rel * n_nodes * ld_t);
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < 2 * n_nodes) run_l[i] = rv[u]; }
            for (int i = 2048 + t; i < 2 * n_nodes; i += 1024) run_l[i] = rsrc[i];
        }
        for (int cb = 0; cb < ne; cb += RG_CHUNK) {
            const int cn = ne - cb < RG_CHUNK ? ne - cb : RG_CHUNK;
            if (cb > 0) __syncthreads();               // readers of the previous chunk are done
            {   // 8 ids (16 B) per lane per load; the segment is padded to a multiple of 8 ids
                const uint4* isrc = reinterpret_cast<const uint4*>(idx + e0 + cb);
                uint4* idst = reinterpret_cast<uint4*>(idx_l);
                const int n8 = (cn + 7) >> 3;
                uint4 iv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < n8 && !(dbg & 4)) iv[u] = isrc[i]; }
#pragma unroll
                for (int u = 0; u < 2; ++u) { const int i = u * 1024 + t; if (i < n8) idst[i] = iv[u]; }
            }
            __syncthreads();
            if (dbg & 1) continue;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                // snake deal of the degree-sorted positions: even bands ascending, odd bands descending
                const int p = NS * j + ((j & 1) ? NS - 1 - slot : slot);
                if (p < n_nodes) {
                    const int b = run_l[2 * p], len = run_l[2 * p + 1];
                    int lo = b > cb ? b : cb;
                    int hi = b + len < cb + cn ? b + len : cb + cn;
                    lo -= cb;
                    hi -= cb;
                    // 8 edges per step: the 8 ids come as ONE 16-byte LDS read (same address for the
                    // slot's lanes = broadcast), so a step costs 1 + 8 LDS reads and no ds_bpermute;
                    // all row reads are issued before the first add (no per-row latency).
                    for (int eb = lo & ~7; eb < hi; eb += 8) {
                        const uint4 pk = *reinterpret_cast<const uint4*>(idx_l + eb);
                        const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
                        float4 v[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const int e = eb + jj;
                            const int idj = (int)((w4[jj >> 1] >> (16 * (jj & 1))) & 0xffffu);
                            const bool ok = e >= lo && e < hi;
                            v[jj] = tipk_ld4(tab + (ok ? idj : 0) * ldt + c0);
                            if (!ok) v[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            acc[j].x += v[jj].x; acc[j].y += v[jj].y; acc[j].z += v[jj].z; acc[j].w += v[jj].w;
                        }
                    }
                }
            }
        }
        if (BWD) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int p = NS * j + ((j & 1) ? NS - 1 - slot : slot);
                if (p < n_nodes && col_ok)
                    tipk_st4(out + ((int64_t)rel * n_nodes + node_at[p]) * ld_out + c0, acc[j]);
                acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int p = NS * j + ((j & 1) ? NS - 1 - slot : slot);
            if (p < n_nodes && col_ok) tipk_st4(out + ((int64_t)wg * n_nodes + node_at[p]) * ld_out + c0, acc[j]);
        }
    }
}

inline int64_t rel_gather_lds(int64_t n_nodes, int d) {
    return n_nodes * (d + 4) * 4 + ((2 * n_nodes + 3) & ~3LL) * 4 + RG_CHUNK * 2;
}

template <int L, int J>
int launch_rg(bool bwd, const float* table, int64_t ld_t, int n_nodes, int d, int n_wg, const int32_t* wg_rel_ptr,
              const int32_t* wg_rels, const int64_t* rel_idx_off, const int32_t* rel_len, const uint16_t* idx,
              const int32_t* runs, const int32_t* node_at, float* out, int64_t ld_out, hipStream_t st) {
    const char* dbg_env = getenv("TIPK_RG_DEBUG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    const size_t lds = (size_t)rel_gather_lds(n_nodes, d);
    hipError_t e;
    if (bwd) {
        auto kern = rel_gather_kernel<L, J, true>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return tipk_hip_status(e);
        hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(1024), lds, st, table, ld_t, n_nodes, d, wg_rel_ptr,
                           wg_rels, rel_idx_off, rel_len, idx, runs, node_at, out, ld_out, dbg);
    } else {
        auto kern = rel_gather_kernel<L, J, false>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return tipk_hip_status(e);
        hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(1024), lds, st, table, ld_t, n_nodes, d, wg_rel_ptr,
                           wg_rels, rel_idx_off, rel_len, idx, runs, node_at, out, ld_out, dbg);
    }
    TIPK_RETURN_LAUNCH();
}

}  // namespace

extern "C" int tipk_rel_gather_supported(int64_t n_nodes, int d) {
    if (n_nodes <= 0 || n_nodes > 65535 || d < 4 || d > 64 || (d & (d - 1)) != 0) return 0;
    const int L = d / 4;
    const int64_t slots = 1024 / L;
    if (n_nodes > 8 * slots) return 0;
    return rel_gather_lds(n_nodes, d) <= 158 * 1024 ? 1 : 0;
}

extern "C" int tipk_rel_gather(int backward, const float* table, int64_t ld_table, int64_t n_nodes, int d,
                               int64_t n_wg, const int32_t* wg_rel_ptr, const int32_t* wg_rels,
                               const int64_t* rel_idx_off, const int32_t* rel_len, const uint16_t* idx,
                               const int32_t* runs, const int32_t* node_at, float* out, int64_t ld_out,
                               tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table || !wg_rel_ptr || !wg_rels || !rel_idx_off || !rel_len || !idx || !runs ||
        !node_at || !out || (reinterpret_cast<uintptr_t>(idx) & 15))
        return TIPK_EINVAL;
    if (!tipk_rel_gather_supported(n_nodes, d)) return TIPK_EUNSUPPORTED;
    if (ld_table % 4 != 0 || ld_out % 4 != 0 || (reinterpret_cast<uintptr_t>(table) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int L = d / 4;
    const int slots = 1024 / L;
    const int j = (int)((n_nodes + slots - 1) / slots);          // nodes owned per slot
#define TIPK_RG(LL, JJ)                                                                                         \
    return launch_rg<LL, JJ>(backward != 0, table, ld_table, (int)n_nodes, d, (int)n_wg, wg_rel_ptr, wg_rels,      \
                             rel_idx_off, rel_len, idx, runs, node_at, out, ld_out, st)
#define TIPK_RG_J(LL)                    \
    do {                                 \
        if (j <= 1) { TIPK_RG(LL, 1); }  \
        if (j <= 2) { TIPK_RG(LL, 2); }  \
        if (j <= 3) { TIPK_RG(LL, 3); }  \
        if (j <= 4) { TIPK_RG(LL, 4); }  \
        if (j <= 6) { TIPK_RG(LL, 6); }  \
        TIPK_RG(LL, 8);                  \
    } while (0)
    switch (L) {
        case 1: TIPK_RG_J(1);
        case 2: TIPK_RG_J(2);
        case 4: TIPK_RG_J(4);
        case 8: TIPK_RG_J(8);
        default: TIPK_RG_J(16);
    }
#undef TIPK_RG_J
#undef TIPK_RG
}
