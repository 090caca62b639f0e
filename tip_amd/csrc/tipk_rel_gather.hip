// Relation-local gather: the D-D aggregation when the node table of ONE relation fits in LDS
// (BioSNAP: 645 drugs x 32 floats = 82 KB).  include/tipk.h section 1b.
//
// The generic `gather_sum` streams every gathered row through the fabric (1.07 GB per launch for the
// 8.3 M D-D edges: 150 us at the HBM roofline).  Here a persistent 1024-thread workgroup per CU walks
// its work units (a unit = one relation, or a share of a big one); everything a unit needs -- its table rows
// (Y_r, forward; backward g' stays for the whole launch), run table, node order and 16-bit edge ids --
// sits in LDS, and every gathered row is a ds_read_b128 from LDS instead of an L2/fabric transaction.
//
// Inside a unit the output nodes are ordered by DECREASING run length (edges of one (relation, node)
// pair = one contiguous run), and slot s (L = columns/4 lanes) takes positions s, s+S, s+2S, ...
// (snake): the 64/L slots of a wavefront work on runs of nearly equal length, empty rows sit at the
// end, and no node is touched by two slots of one unit -> no atomics anywhere.
//   FWD: the run sum is added to a per-workgroup fp32 accumulator image in LDS (plain read-modify-
//        write, exclusive by construction); one partial slab per workgroup, combined in order by
//        tipk_sum_slabs.  When table + accumulators exceed the LDS the columns are processed in
//        `n_split` independent column blocks (blockIdx.y).
//   BWD: the run sum IS the output row (relation, node): written straight to dY.
// Work units: a relation much larger than the per-workgroup average would set the length of the whole
// launch, so the plan deals the positions of such a relation round-robin to k units; units are assigned
// to workgroups by a longest-processing-time deal.
//
// Staging (round 2, after in-kernel cycle stamps -- tools/rg_stamps.py -- showed 45-60 % of a wave's life
// outside the gather loops: two barriers per unit with 1.5x unequal waves, a register->LDS commit pass,
// synchronous id-chunk reloads, s_waitcnt vmcnt that hipcc attached to register prefetches):
//   * every per-stage array arrives by LDS-DMA (global_load_lds_dwordx4: no registers, no copy pass) into
//     the buffer the stage before last used: ids in two chunk buffers, run table / node order / table image
//     (forward) in two unit buffers.  All LDS images are plain linear copies (the table unpadded), so one
//     wave-instruction's 64 x 16 bytes land contiguously (destination = wave-uniform base + lane x 16);
//   * a STAGE = (unit, id chunk); ONE barrier per stage: it makes stage s visible (hipcc drains vmcnt before
//     it) and certifies that everyone has left stage s-1, whose buffers the DMA for stage s+1 -- issued right
//     behind the barrier -- overwrites while stage s is walked.
// Results are bitwise reproducible (fixed order everywhere).
#include "tipk_common.h"

#ifdef TIPK_DEBUG
// debug builds only (make debug): per-wave cycle stamps of the last rel_gather launch, read back by
// tipk_debug_rg_stamps (tools/rg_stamps.py).  [workgroup][wave][8] = { total, position loops, --, waiting
// at the stage barrier (incl. the DMA drain), start -> first stage, epilogue, issuing DMA, stages }.
__device__ unsigned long long tipk_rg_stamps[512 * 16 * 8];
#define RG_STAMP(var) unsigned long long var = __builtin_readcyclecounter()
#else
#define RG_STAMP(var)
#endif

namespace {

constexpr int RG_CHUNK_MAX = 16384;    // edge ids per chunk buffer (uint16: 32 KB), less when LDS is short
constexpr int RG_CHUNK_MIN = 4096;
constexpr int RG_META = 16;            // unit descriptors staged per batch (LDS: 512 B)

struct RgArgs {
    const float* table; int64_t ld_t;
    int n_nodes, dc;                   // dc = columns handled by one column block
    int np;                            // entries per unit of runs / node_at (n_nodes rounded up to 8)
    const int32_t* wg_unit_ptr;        // [n_wg + 1] range of every workgroup in unit_meta
    const int32_t* unit_meta;          // [n_units][8] in workgroup order: unit, relation, n_pos, n_ids, idx_off lo/hi
    const uint16_t* idx; const uint32_t* runs; const uint16_t* node_at;
    float* out; int64_t ld_out;
    const float* row_scale;            // BWD: g' = row_scale[node] * table[node] applied while staging (nullable)
    int chunk;                         // ids per chunk buffer
    int idx_mul;                       // byte offset of a table row = idx value * idx_mul (plan stores node * idx_unit)
    int dbg;
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;

// TU = float4 of one relation's table per thread (ceil(n_nodes*dc/4 / 1024)): LDS-DMA instructions per thread
// and unit (forward pass).
template <int L, bool BWD, int TU, bool UNIT>
__global__ __launch_bounds__(1024) void rel_gather_kernel(RgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = 1024 / L;                       // slots per workgroup
    constexpr int q4 = L;                              // float4 per row of a column block
    const int t = threadIdx.x;
    const int n_nodes = a.n_nodes, dc = a.dc, np = a.np;
    // unpadded power-of-two rows: a row's 64-byte quarter of the 64 banks is (node mod C); the PLAN orders every
    // run so that the slots of one ds_read_b128 lane group hit different quarters (tip_amd/plan.py
    // `bank_rotation`) -- padding only randomises the conflicts
    const int ldt = dc;
    const int CH = a.chunk;
    float* tab = lds;                                                    // [1 | 2][n_nodes + 1][dc], last row = 0
    const int tab_floats = (n_nodes + 1) * ldt;
    float* accl = tab + (BWD ? 1 : 2) * (int64_t)tab_floats;             // FWD: [n_nodes][dc]
    // run table: one word per position = (begin / 8) | (length / 8) << 16 (both multiples of 8 ids; a unit
    // has < 2^19 ids), as the plan stores it
    uint32_t* run_l = reinterpret_cast<uint32_t*>(accl + (BWD ? 0 : (int64_t)n_nodes * dc));     // [2][np]
    uint16_t* node_l = reinterpret_cast<uint16_t*>(run_l + 2 * np);      // [2][np]
    uint16_t* idx_l = node_l + 2 * np;                                   // [2][CH], 16-B aligned (np % 8 == 0)
    int32_t* meta_l = reinterpret_cast<int32_t*>(idx_l + 2 * CH);        // [RG_META][8]
    const int slot = t / L, sub = t & (L - 1), c0 = sub * 4;
    const int col0 = blockIdx.y * dc;                  // column block of this workgroup
    const float* table = a.table + col0;
    float* out = a.out + col0;
    const int total4 = n_nodes * q4;

    // descriptor words are wave-uniform: reading them through readfirstlane keeps the address arithmetic of
    // the DMA on the scalar unit
    auto sget = [&](const int32_t* m, int k) { return __builtin_amdgcn_readfirstlane(m[k]); };
    // linear global -> LDS copy of n16 16-byte pieces: piece i = k * 1024 + t lands at dst + i * 16
    auto dma = [&](const void* src, void* dst, int n16) {
        for (int k = 0; k * 1024 < n16; ++k) {
            const int i = k * 1024 + t;
            if (i < n16)
                __builtin_amdgcn_global_load_lds((global_void_t*)(reinterpret_cast<const char*>(src) + (uint32_t)i * 16u),
                                                 (lds_void_t*)(reinterpret_cast<char*>(dst) + (k * 1024 + (t & ~63)) * 16), 16, 0, 0);
        }
    };
    // everything stage (unit m, chunk cb) needs -> the buffers `sbuf` (ids) and `ubuf` (per-unit arrays)
    auto issue = [&](const int32_t* m, int cb, int sbuf, int ubuf) {
        const int ne = sget(m, 3);
        const int64_t off = ((int64_t)(uint32_t)sget(m, 4) | ((int64_t)sget(m, 5) << 32)) + cb;
        const int cn = ne - cb < CH ? ne - cb : CH;
        dma(a.idx + off, idx_l + sbuf * CH, cn >> 3);
        if (cb == 0) {
            const int unit = sget(m, 0);
            dma(a.runs + (int64_t)unit * np, run_l + ubuf * np, np >> 2);
            dma(a.node_at + (int64_t)unit * np, node_l + ubuf * np, np >> 3);
            if (!BWD) {
                // table rows of this column block: source rows are ld_t apart, the image is linear in the float4
                // index i = row * L + column / 4
                const float* src = table + (int64_t)sget(m, 1) * n_nodes * a.ld_t;
                float* dst = tab + (int64_t)ubuf * tab_floats;
#pragma unroll
                for (int u = 0; u < TU; ++u) {
                    const int i = u * 1024 + t;
                    if (i < total4) {
                        const int r = i / q4, c = (i - r * q4) * 4;
                        __builtin_amdgcn_global_load_lds((global_void_t*)(src + (uint32_t)(r * (int)a.ld_t + c)),
                                                         (lds_void_t*)(dst + (int64_t)(u * 1024 + (t & ~63)) * 4), 16, 0, 0);
                    }
                }
            }
        }
    };

    if (t < ldt) {                                                        // the sentinel's row (of both images)
        tab[(int64_t)n_nodes * ldt + t] = 0.f;
        if (!BWD) tab[(int64_t)tab_floats + (int64_t)n_nodes * ldt + t] = 0.f;
    }
    if (!BWD) {
        for (int i = t; i < n_nodes * q4; i += 1024) tipk_st4(accl + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    } else {                                           // g' is staged once and stays for the whole launch
        for (int base = 0; base < total4; base += 4096) {
            float4 gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int i = base + u * 1024 + t;
                i = i < total4 ? i : total4 - 1;
                const int r = i / q4, c = (i - r * q4) * 4;
                gv[u] = tipk_ld4(table + (int64_t)r * a.ld_t + c);
                if (a.row_scale) { const float sc = a.row_scale[r]; gv[u].x *= sc; gv[u].y *= sc; gv[u].z *= sc; gv[u].w *= sc; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 1024 + t;
                if (i < total4) { const int r = i / q4, c = (i - r * q4) * 4; tipk_st4(tab + r * ldt + c, gv[u]); }
            }
        }
    }
    const int wg = blockIdx.x;
#ifdef TIPK_DEBUG
    unsigned long long st_loop = 0, st_wait = 0, st_issue = 0, st_first = 0, st_stages = 0;
    RG_STAMP(st_t0);
#endif
    const int ri0 = a.wg_unit_ptr[wg], ri1 = a.wg_unit_ptr[wg + 1];
    int stage = 0;                                     // parity of the id chunk buffer, continuous over the launch
    for (int rb = ri0; rb < ri1; rb += RG_META) {      // batches of RG_META units (one batch in practice)
        const int rn = ri1 - rb < RG_META ? ri1 - rb : RG_META;
        __syncthreads();                               // the previous batch's descriptors / buffers are no longer read
        if (t < rn * 8) meta_l[t] = a.unit_meta[(int64_t)rb * 8 + t];
        __syncthreads();
        issue(meta_l, 0, stage & 1, 0);
        int ri = 0, cb = 0;
        while (ri < rn) {
            const int32_t* m = meta_l + ri * 8;        // a work unit: one relation, or a share of a big one
            const int npos = sget(m, 2), ne = sget(m, 3);
            const int64_t row0 = (int64_t)sget(m, 1) * n_nodes;
            RG_STAMP(st_a);
#ifdef TIPK_DEBUG
            if (st_first == 0) st_first = st_a - st_t0;
#endif
            __syncthreads();                           // stage visible (vmcnt drained); everyone has left the stage before
            RG_STAMP(st_b);
            // the NEXT stage travels while this one is walked
            int nri = ri, ncb = cb + CH;
            if (ncb >= ne) { nri = ri + 1; ncb = 0; }
            if (nri < rn && !TIPK_DBG(a.dbg & 2)) issue(meta_l + nri * 8, ncb, (stage + 1) & 1, nri & 1);
            asm volatile("" ::: "memory");            // keep the DMA issues above the position loops
            RG_STAMP(st_c);
#ifdef TIPK_DEBUG
            st_wait += st_b - st_a;
            st_issue += st_c - st_b;
            st_stages += 1;
#endif
            const int cn = ne - cb < CH ? ne - cb : CH;
            const uint32_t* runs_u = run_l + (ri & 1) * np;
            const uint16_t* nodes_u = node_l + (ri & 1) * np;
            const uint16_t* ids_s = idx_l + (stage & 1) * CH;
            // Runs are short (BioSNAP: 13.6 padded ids = 1.7 steps on average), so what a slot does AROUND
            // a run -- fetch (begin, length) and the node, read-modify-write the accumulator -- is a chain
            // of dependent LDS round trips as long as the run itself.  The next position's descriptor is
            // therefore requested one band ahead, and the forward pass requests the old accumulator value
            // before the run instead of after it.
            if (npos > 0 && !TIPK_DBG(a.dbg & 1)) {
                int p_next = slot;                      // band 0 is ascending
                uint32_t run_next;
                unsigned node_next;
                {
                    const int pc = p_next < npos ? p_next : npos - 1;
                    run_next = runs_u[pc];
                    node_next = nodes_u[pc];
                }
                const char* tabb = reinterpret_cast<const char*>(tab + (BWD ? 0 : (ri & 1) * tab_floats) + c0);
                const unsigned ldt4 = (unsigned)a.idx_mul;   // 1 when the plan pre-scaled the ids to byte offsets
                for (int pb = 0, band = 0; pb < npos; pb += NS, ++band) {
                    // snake deal of the length-sorted rows: band 0 ascending, band 1 descending, ... so the
                    // slot that got the longest row of one band gets the shortest of the next
                    const int p = p_next;
                    const int b = (int)(run_next & 0xffffu) << 3, len = (int)(run_next >> 16) << 3;
                    const unsigned node = node_next;
                    p_next = pb + NS + (((band + 1) & 1) ? NS - 1 - slot : slot);
                    {
                        const int pc = p_next < npos ? p_next : npos - 1;     // clamped, unconditional
                        run_next = runs_u[pc];
                        node_next = nodes_u[pc];
                    }
                    if (p >= npos) continue;
                    if (!BWD && len == 0) continue;
                    int lo = b > cb ? b : cb;            // b, len, cb, cn are multiples of 8
                    int hi = b + len < cb + cn ? b + len : cb + cn;
                    lo -= cb;
                    hi -= cb;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 old_acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!BWD && lo < hi) old_acc = tipk_ld4(accl + node * dc + c0);   // this slot is the only one touching `node` now
                    // 8 edges per step.  Runs are padded to multiples of 8 ids with the sentinel, whose table row is
                    // all zeros: no masks, no tails.  A step = one id read through a running pointer (16 bytes past
                    // the run is harmless: the next buffer follows), 8 SDWA adds that turn the pre-scaled 16-bit ids
                    // into row addresses, 8 row reads, ONE wait (the last row requested is added first: LDS returns in
                    // order), 16 packed adds.
                    const uint16_t* idp = ids_s + lo;
                    uint4 pk = lo < hi ? *reinterpret_cast<const uint4*>(idp) : make_uint4(0, 0, 0, 0);
                    for (int eb = lo; eb < hi; eb += 8) {
                        const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
                        const char* ad[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const unsigned idj = (jj & 1) ? (w4[jj >> 1] >> 16) : (w4[jj >> 1] & 0xffffu);
                            ad[jj] = UNIT ? tabb + idj : tabb + __umul24(idj, ldt4);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        idp += 8;
                        pk = *reinterpret_cast<const uint4*>(idp);           // next step's ids (unconditional)
                        float4 v[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) v[jj] = *reinterpret_cast<const float4*>(ad[jj]);
#pragma unroll
                        for (int jj = 7; jj >= 0; --jj) {
                            acc.x += v[jj].x; acc.y += v[jj].y; acc.z += v[jj].z; acc.w += v[jj].w;
                        }
                    }
                    if (BWD) {
                        float* o = out + (row0 + node) * a.ld_out + c0;
                        if (cb > 0 && lo < hi) {       // a run continued from the previous id chunk
                            const float4 old = tipk_ld4(o);
                            acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
                        }
                        if (cb == 0 || lo < hi) tipk_st4(o, acc);
                    } else if (lo < hi) {
                        acc.x += old_acc.x; acc.y += old_acc.y; acc.z += old_acc.z; acc.w += old_acc.w;
                        tipk_st4(accl + node * dc + c0, acc);
                    }
                }
            }
#ifdef TIPK_DEBUG
            st_loop += __builtin_readcyclecounter() - st_c;
#endif
            ri = nri;
            cb = ncb;
            ++stage;
        }
    }
    RG_STAMP(st_epi);
    if (!BWD) {
        __syncthreads();
        float* o = out + (int64_t)wg * n_nodes * a.ld_out;
        for (int i = t; i < n_nodes * q4; i += 1024) {
            const int r = i / q4, c = (i - r * q4) * 4;
            tipk_st4(o + (int64_t)r * a.ld_out + c, tipk_ld4(accl + r * dc + c));
        }
    }
#ifdef TIPK_DEBUG
    if ((t & 63) == 0) {
        const int w = (blockIdx.y * gridDim.x + blockIdx.x) * 16 + (t >> 6);
        if (w < 512 * 16) {
            const unsigned long long now = __builtin_readcyclecounter();
            tipk_rg_stamps[w * 8 + 0] = now - st_t0;
            tipk_rg_stamps[w * 8 + 1] = st_loop;
            tipk_rg_stamps[w * 8 + 2] = 0;
            tipk_rg_stamps[w * 8 + 3] = st_wait;
            tipk_rg_stamps[w * 8 + 4] = st_first;
            tipk_rg_stamps[w * 8 + 5] = now - st_epi;
            tipk_rg_stamps[w * 8 + 6] = st_issue;
            tipk_rg_stamps[w * 8 + 7] = st_stages;
        }
    }
#endif
}

inline int64_t rg_np(int64_t n_nodes) { return (n_nodes + 7) & ~7LL; }

inline int64_t rel_gather_lds(int64_t n_nodes, int dc, bool bwd, int chunk) {
    const int64_t np = rg_np(n_nodes);
    return (bwd ? 1 : 2) * (n_nodes + 1) * dc * 4 + (bwd ? 0 : n_nodes * dc * 4) + 2 * np * 4 + 2 * np * 2 +
           2 * (int64_t)chunk * 2 + RG_META * 8 * 4;
}

constexpr int64_t RG_LDS_LIMIT = 158 * 1024;

// The launch shape: `split` column blocks (grid.y), `occ` workgroups per CU the LDS footprint
// admits (1 or 2), `chunk` ids per chunk buffer (the largest multiple of 1024 in [4096, 16384] that fits).
// Cutting the columns finer lets TWO 1024-thread workgroups share a CU (8 waves per SIMD; <= 64 VGPRs)
// at the price of walking the run tables once more per extra column block -- measured slower on BioSNAP
// (83 us against 49 us for the d = 32 forward launch); `want_occ` (option "rg_occupancy") picks.
// split = 0: shape not supported.
struct RgShape { int split, occ, chunk; };
inline RgShape rel_gather_shape(int64_t n_nodes, int d, bool bwd, int want_occ) {
    RgShape none = {0, 0, 0};
    if (n_nodes <= 0 || n_nodes > 1024 || d < 4 || d > 256 || (d & (d - 1)) != 0) return none;
    for (int occ = want_occ >= 2 ? 2 : 1; occ >= 1; --occ)
        for (int split = 1; d / split >= 4; split *= 2) {
            const int dc = d / split;
            if (dc > 64) continue;                          // L = dc/4 <= 16 lanes per slot
            if (n_nodes * (dc / 4) > 8192) continue;       // table rows: at most 8 DMA pieces per thread
            if (occ == 2 && dc < 8) continue;               // 4-column blocks: the run tables dominate
            for (int chunk = RG_CHUNK_MAX; chunk >= RG_CHUNK_MIN; chunk -= 1024)
                if (rel_gather_lds(n_nodes, dc, bwd, chunk) * occ <= RG_LDS_LIMIT + (occ - 1) * 2048) {
                    RgShape s = {split, occ, chunk};
                    return s;
                }
        }
    return none;
}
inline int rg_want_occ() { const int o = tipk_option(TIPK_OPT_RG_OCCUPANCY); return o == 0 ? 1 : o; }

template <int L, bool BWD, int TU>
int launch_rg3(const RgArgs& a, int n_wg, int split, hipStream_t st) {
    const size_t lds = (size_t)rel_gather_lds(a.n_nodes, a.dc, BWD, a.chunk);
    auto kern = a.idx_mul == 1 ? rel_gather_kernel<L, BWD, TU, true> : rel_gather_kernel<L, BWD, TU, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg, (unsigned)split), dim3(1024), lds, st, a);
    TIPK_RETURN_LAUNCH();
}

template <int L>
int launch_rg(bool bwd, const RgArgs& a, int n_wg, int split, hipStream_t st) {
    if (bwd) return launch_rg3<L, true, 1>(a, n_wg, split, st);
    const int64_t per_thread = ((int64_t)a.n_nodes * (a.dc / 4) + 1023) / 1024;
    if (per_thread <= 2) return launch_rg3<L, false, 2>(a, n_wg, split, st);
    if (per_thread <= 4) return launch_rg3<L, false, 4>(a, n_wg, split, st);
    return launch_rg3<L, false, 8>(a, n_wg, split, st);
}

}  // namespace

#ifdef TIPK_DEBUG
extern "C" int tipk_debug_rg_stamps(unsigned long long* host_out /* [512*16*8] */) {
    return tipk_hip_status(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tipk_rg_stamps), sizeof(unsigned long long) * 512 * 16 * 8));
}
#endif

extern "C" int tipk_rel_gather_supported(int64_t n_nodes, int d, int backward) {
    return rel_gather_shape(n_nodes, d, backward != 0, rg_want_occ()).split;
}

extern "C" int tipk_rel_gather_chunk(int64_t n_nodes, int d, int backward) {
    return rel_gather_shape(n_nodes, d, backward != 0, rg_want_occ()).chunk;
}

extern "C" int tipk_rel_gather_occupancy(int64_t n_nodes, int d, int backward) {
    return rel_gather_shape(n_nodes, d, backward != 0, rg_want_occ()).occ;
}

extern "C" int tipk_rel_gather(int backward, const float* table, int64_t ld_table, int64_t n_nodes, int d,
                               int64_t n_wg, const int32_t* wg_unit_ptr, const int32_t* unit_meta,
                               const uint16_t* idx, int idx_unit, const uint32_t* runs, const uint16_t* node_at,
                               const float* row_scale, float* out, int64_t ld_out, tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table || !wg_unit_ptr || !unit_meta || !idx || !runs || !node_at || !out ||
        (reinterpret_cast<uintptr_t>(idx) & 15) || (reinterpret_cast<uintptr_t>(runs) & 15) ||
        (reinterpret_cast<uintptr_t>(node_at) & 15))
        return TIPK_EINVAL;
    const RgShape shape = rel_gather_shape(n_nodes, d, backward != 0, rg_want_occ());
    const int split = shape.split;
    if (split == 0) return TIPK_EUNSUPPORTED;
    if (ld_table % 4 != 0 || ld_out % 4 != 0 || (reinterpret_cast<uintptr_t>(table) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return TIPK_EINVAL;
    RgArgs a;
    a.table = table; a.ld_t = ld_table; a.n_nodes = (int)n_nodes; a.dc = d / split; a.np = (int)rg_np(n_nodes);
    a.wg_unit_ptr = wg_unit_ptr; a.unit_meta = unit_meta;
    a.idx = idx; a.runs = runs; a.node_at = node_at; a.out = out; a.ld_out = ld_out;
    a.row_scale = backward ? row_scale : nullptr;
    a.chunk = shape.chunk;
    // idx holds node * idx_unit (the sentinel: n_nodes * idx_unit <= 65535); a row is dc * 4 bytes
    if (idx_unit <= 0 || (a.dc * 4) % idx_unit != 0 || (int64_t)n_nodes * idx_unit > 65535) return TIPK_EINVAL;
    a.idx_mul = a.dc * 4 / idx_unit;
    a.dbg = TIPK_DBG(tipk_option(TIPK_OPT_RG_DEBUG));
    hipStream_t st = (hipStream_t)stream;
    switch (a.dc / 4) {
        case 1: return launch_rg<1>(backward != 0, a, (int)n_wg, split, st);
        case 2: return launch_rg<2>(backward != 0, a, (int)n_wg, split, st);
        case 4: return launch_rg<4>(backward != 0, a, (int)n_wg, split, st);
        case 8: return launch_rg<8>(backward != 0, a, (int)n_wg, split, st);
        default: return launch_rg<16>(backward != 0, a, (int)n_wg, split, st);
    }
}
