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
// Work units: a relation much larger than the per-workgroup average (BioSNAP: 51 466 edges against
// 32 525) would set the length of the whole launch, so the plan deals the positions of such a relation
// round-robin to k units (each stages the relation's table again and walks every k-th position); units
// are assigned to workgroups by a longest-processing-time deal.
// Results are bitwise reproducible (fixed order everywhere).
#include <stdlib.h>
#include "tipk_common.h"

#ifdef TIPK_DEBUG
// debug builds only (make debug): per-wave cycle stamps of the last rel_gather launch, read back by
// tipk_debug_rg_stamps (tools/rg_stamps.py).  [workgroup][wave][6] = { total, position loops, commit
// (barrier to barrier), waiting at the unit's first barrier, prologue, epilogue }.  Nothing here exists in the release library.
__device__ unsigned long long tipk_rg_stamps[512 * 16 * 8];
#define RG_STAMP(var) unsigned long long var = __builtin_readcyclecounter()
#else
#define RG_STAMP(var)
#endif

namespace {

constexpr int RG_CHUNK_MAX = 16384;    // edge ids staged per pass (uint16: 32 KB); 8192 when that lets two workgroups share a CU
constexpr int RG_META = 16;            // unit descriptors staged per batch (LDS: 512 B)

struct RgArgs {
    const float* table; int64_t ld_t;
    int n_nodes, dc;                   // dc = columns handled by one column block
    const int32_t* wg_unit_ptr;        // [n_wg + 1] range of every workgroup in unit_meta
    const int32_t* unit_meta;          // [n_units][8] in workgroup order: unit, relation, n_pos, n_ids, idx_off lo/hi
    const uint16_t* idx; const int32_t* runs; const uint16_t* node_at;
    float* out; int64_t ld_out;
    const float* row_scale;            // BWD: g' = row_scale[node] * table[node] applied while staging (nullable)
    int chunk;                         // ids staged per pass (8192 or 16384)
    int idx_mul;                       // byte offset of a table row = idx value * idx_mul (plan stores node * idx_unit)
    int dbg;
};

// TU = float4 of one relation's table per thread (ceil(n_nodes*dc/4 / 1024)): LDS-DMA instructions per thread
// and unit (forward pass).
template <int L, bool BWD, int TU, bool UNIT>
__global__ __launch_bounds__(1024) void rel_gather_kernel(RgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = 1024 / L;                       // slots per workgroup
    const int t = threadIdx.x;
    const int n_nodes = a.n_nodes, dc = a.dc;
    const int ldt = dc;                                // unpadded power-of-two rows: a row's 64-byte quarter of the 64 banks is (node mod C);
                                                       // the PLAN orders every run so that the slots of one ds_read_b128 lane group hit different
                                                       // quarters (tip_amd/plan.py `bank_rotation`) -- padding only randomises the conflicts
    constexpr int q4 = L;                              // float4 per row of a column block (compile-time: the row/column
                                                       // split of a linear index below is a shift, not a 40-instruction division)
    // FWD: TWO table images -- the next unit's rows arrive by LDS-DMA (global_load_lds_dwordx4: no
    // registers, no copy pass) into the image the previous unit used, while the current unit is gathered
    float* tab = lds;                                                   // [n_nodes + 1][dc], last row = 0
    const int tab_floats = (n_nodes + 1) * ldt;
    float* accl = tab + (BWD ? 1 : 2) * (int64_t)tab_floats;                           // FWD: [n_nodes][dc]
    // run table in LDS: one word per position = (begin / 8) | (length / 8) << 16 (both are multiples of 8 ids;
    // a unit has < 2^19 ids) -- half the bytes of the plan's int32 pairs, which is what lets a second table
    // image fit next to a 16 K id chunk
    uint32_t* run_l = reinterpret_cast<uint32_t*>(accl + (BWD ? 0 : (int64_t)n_nodes * dc));   // [n_nodes]
    uint16_t* node_l = reinterpret_cast<uint16_t*>(run_l + ((n_nodes + 3) & ~3));      // [n_nodes] (+pad)
    const int RG_CHUNK = a.chunk;
    uint16_t* idx_l = node_l + ((n_nodes + 7) & ~7);                                   // [chunk], 16-B aligned
    // unit descriptors of this workgroup, staged once: reading them from global memory per unit cost
    // four dependent round trips (unit id -> offsets -> ...) that nothing could hide (one workgroup per CU)
    int32_t* meta_l = reinterpret_cast<int32_t*>(idx_l + RG_CHUNK);                    // [RG_META][8]
    const int slot = t / L, sub = t & (L - 1), c0 = sub * 4;
    const int col0 = blockIdx.y * dc;                  // column block of this workgroup
    const float* table = a.table + col0;
    float* out = a.out + col0;

    // Staging is software-pipelined across relations: the NEXT relation's table rows, run table,
    // node order and first id chunk are requested into registers before the current relation is
    // processed and written to LDS afterwards, so HBM latency hides behind the LDS-bound compute
    // (one workgroup per CU: there is no other wave to hide it).
    // (named scalars, not arrays: hipcc keeps small arrays that are written in one lambda and read in
    // another in scratch memory, which puts a vmcnt(0) wait right behind every prefetch load)
    int2 rv = make_int2(0, 0);
    uint16_t nv = 0;
    uint4 iv0, iv1;
    const int total4 = n_nodes * q4;
    // The small per-unit arrays (run table, node order, first id chunk) are prefetched into registers one
    // unit ahead; their loads are UNCONDITIONAL (indices clamped into the valid range): a load inside an
    // exec-masked branch makes hipcc wait vmcnt(0) at the end of the branch.  The TABLE (41 KB per unit at
    // BioSNAP) does not pass through registers: measured with in-kernel stamps (tools/rg_stamps.py), the
    // register-staged version spent 36 % of a wave's life behind the s_waitcnt vmcnt that hipcc put after the
    // prefetch (it moved half-loaded table registers around) and 11 % writing them to LDS.
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void global_void_t;
    // descriptor words are wave-uniform: reading them through readfirstlane keeps the address arithmetic of
    // the prefetches on the scalar unit (stamps: issuing a unit's prefetch cost 1 200 cycles per wave with
    // 64-bit vector address math, 13 % of the forward launch)
    auto sget = [&](const int32_t* m, int k) { return __builtin_amdgcn_readfirstlane(m[k]); };
    // the ids of one chunk of one unit -> iv0 / iv1 (the plan pads the id array by a whole chunk, so the
    // loads are unconditional and unclamped; words past the unit's end are never used)
    auto prefetch_ids = [&](const int32_t* m, int cb) {
        const int64_t off = ((int64_t)(uint32_t)sget(m, 4) | ((int64_t)sget(m, 5) << 32)) + cb;
        const uint4* isrc = reinterpret_cast<const uint4*>(a.idx + off);
        iv0 = isrc[t];
        iv1 = isrc[1024 + t];
    };
    auto prefetch_unit = [&](const int32_t* m, int which) {  // run table, node order, table image (by LDS-DMA)
        const int unit = sget(m, 0);
        const int tc = t < n_nodes ? t : n_nodes - 1;
        rv = reinterpret_cast<const int2*>(a.runs + (int64_t)unit * n_nodes * 2)[tc];
        nv = (a.node_at + (int64_t)unit * n_nodes)[tc];
        if (!BWD) {
            // the LDS image is the linear order of the float4 index i = row * L + column/4, so one
            // wave-instruction's 64 x 16 bytes land contiguously (destination = wave-uniform base + lane * 16)
            const float* src = table + (int64_t)sget(m, 1) * n_nodes * a.ld_t;
            float* dst = tab + (int64_t)which * tab_floats;
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
    };
    auto commit_ids = [&]() {                          // the whole buffer (entries past the chunk are never read)
        uint4* idst = reinterpret_cast<uint4*>(idx_l);
        idst[t] = iv0;
        if (RG_CHUNK > 8192) idst[1024 + t] = iv1;     // wave-uniform
    };
    auto commit_unit = [&]() {
        if (t < n_nodes) {
            run_l[t] = ((uint32_t)rv.x >> 3) | (((uint32_t)rv.y >> 3) << 16);
            node_l[t] = nv;
        }
    };
    if (t < ldt) {                                                         // the sentinel's row (of both images)
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
    unsigned long long st_loop = 0, st_commit = 0, st_wait = 0, st_pref = 0, st_reload = 0, st_first = 0;
    RG_STAMP(st_t0);
#endif
    const int ri0 = a.wg_unit_ptr[wg], ri1 = a.wg_unit_ptr[wg + 1];
#ifdef TIPK_DEBUG
    const unsigned long long st_pro = __builtin_readcyclecounter() - st_t0;
#endif
    for (int rb = ri0; rb < ri1; rb += RG_META) {      // batches of RG_META units (one batch in practice)
    const int rn = ri1 - rb < RG_META ? ri1 - rb : RG_META;
    __syncthreads();                                   // the previous batch's descriptors are no longer read
    if (t < rn * 8) meta_l[t] = a.unit_meta[(int64_t)rb * 8 + t];
    __syncthreads();
    prefetch_unit(meta_l, 0);
    prefetch_ids(meta_l, 0);
    for (int ri = 0; ri < rn; ++ri) {
        const int32_t* m = meta_l + ri * 8;            // a work unit: one relation, or a share of a big one
        const int npos = sget(m, 2), ne = sget(m, 3);
        const int64_t row0 = (int64_t)sget(m, 1) * n_nodes;
        RG_STAMP(st_a);
#ifdef TIPK_DEBUG
        if (st_first == 0) st_first = st_a - st_t0;
#endif
        __syncthreads();                               // readers of the previous unit are done
        RG_STAMP(st_b);
        commit_unit();
        commit_ids();
        __syncthreads();
        RG_STAMP(st_c);
#ifdef TIPK_DEBUG
        st_wait += st_b - st_a;
        st_commit += st_c - st_b;
#endif
        const bool more = ri + 1 < rn && !TIPK_DBG(a.dbg & 2);
        if (more) prefetch_unit(m + 8, (ri + 1) & 1);  // in flight during the compute below
        asm volatile("" ::: "memory");                // hipcc otherwise sinks some of the DMA issues below the position loops
#ifdef TIPK_DEBUG
        st_pref += __builtin_readcyclecounter() - st_c;
#endif
        for (int cb = 0; cb == 0 || cb < ne; cb += RG_CHUNK) {
            const int cn = ne - cb < RG_CHUNK ? ne - cb : RG_CHUNK;
            if (cb > 0) {                              // a unit with more ids than one chunk: the next chunk was
                RG_STAMP(st_r0);                       // requested into registers while the previous one was walked
                __syncthreads();
                commit_ids();
                __syncthreads();
#ifdef TIPK_DEBUG
                st_reload += __builtin_readcyclecounter() - st_r0;
#endif
            }
            // ids of the NEXT stage (next chunk of this unit, or the first chunk of the next unit) travel
            // while this chunk is walked: no id load is ever waited for with nothing else to do
            if (cb + RG_CHUNK < ne) prefetch_ids(m, cb + RG_CHUNK);
            else if (more) prefetch_ids(m + 8, 0);
            asm volatile("" ::: "memory");
            if (TIPK_DBG(a.dbg & 1)) continue;
            // Runs are short (BioSNAP: 13.6 padded ids = 1.7 steps on average), so what a slot does AROUND
            // a run -- fetch (begin, length) and the node, read-modify-write the accumulator -- is a chain
            // of dependent LDS round trips as long as the run itself.  The next position's descriptor is
            // therefore requested one band ahead (one 8-byte read + the node), and the forward pass
            // requests the old accumulator value before the run instead of after it.
            if (npos <= 0) continue;
            RG_STAMP(st_l0);
            int p_next = slot;                          // band 0 is ascending
            uint32_t run_next;
            unsigned node_next;
            {
                const int pc = p_next < npos ? p_next : npos - 1;
                run_next = run_l[pc];
                node_next = node_l[pc];
            }
            for (int pb = 0, band = 0; pb < npos; pb += NS, ++band) {
                // snake deal of the length-sorted rows: band 0 ascending, band 1 descending, ... so the
                // slot that got the longest row of one band gets the shortest of the next
                const int p = p_next;
                const int b = (int)(run_next & 0xffffu) << 3, len = (int)(run_next >> 16) << 3;
                const unsigned node = node_next;
                p_next = pb + NS + (((band + 1) & 1) ? NS - 1 - slot : slot);
                {
                    const int pc = p_next < npos ? p_next : npos - 1;         // clamped, unconditional
                    run_next = run_l[pc];
                    node_next = node_l[pc];
                }
                if (p >= npos) continue;
                if (!BWD && len == 0) continue;
                int lo = b > cb ? b : cb;                // b, len, cb, cn are multiples of 8
                int hi = b + len < cb + cn ? b + len : cb + cn;
                lo -= cb;
                hi -= cb;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 old_acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!BWD && lo < hi) old_acc = tipk_ld4(accl + node * dc + c0);   // this slot is the only one touching `node` now
                // 8 edges per step.  Runs are padded to multiples of 8 ids with the sentinel id
                // n_nodes, whose table row is all zeros: no masks, no tails -- a step is one 16-byte id
                // read (same address for the slot's lanes = broadcast) + 8 row reads + 8 adds.
                // (the next step's ids are requested before this step's rows; row address = 24-bit multiply:
                // a plain `idj * ldt` compiled to the quarter-rate v_mul_lo_u32 -- 8 of them were half of
                // the loop's VALU time, which is what bounds the kernel together with the LDS pipe)
                const char* tabb = reinterpret_cast<const char*>(tab + (BWD ? 0 : (ri & 1) * tab_floats) + c0);
                const unsigned ldt4 = (unsigned)a.idx_mul;                   // 1 when the plan pre-scaled the ids to byte offsets
                // The loop is bound by instruction ISSUE (PMC, profiles/r02a_lds.json: the SIMDs issue 72 % of
                // the time, VALU 41 % + scalar/waits 15 % + LDS 12 %), so every step is kept to: one id read
                // through a running pointer (reading 16 bytes past the run is harmless: the id buffer is
                // followed by the descriptor buffer), 8 SDWA adds that turn the pre-scaled 16-bit ids into row
                // addresses, 8 row reads, ONE wait for all of them, 16 packed adds (rows added last to first).
                const uint16_t* idp = idx_l + lo;
                uint4 pk = lo < hi ? *reinterpret_cast<const uint4*>(idp) : make_uint4(0, 0, 0, 0);
                for (int eb = lo; eb < hi; eb += 8) {
                    // addresses first, THEN the next ids into the same registers (no copy, no wait for the
                    // read that was just issued), then the rows
                    const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
                    const char* ad[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const unsigned idj = (jj & 1) ? (w4[jj >> 1] >> 16) : (w4[jj >> 1] & 0xffffu);
                        ad[jj] = UNIT ? tabb + idj : tabb + __umul24(idj, ldt4);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    idp += 8;
                    pk = *reinterpret_cast<const uint4*>(idp);               // next step's ids (unconditional)
                    float4 v[8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) v[jj] = *reinterpret_cast<const float4*>(ad[jj]);
                    // the LAST row requested is added first: its arrival implies all the others (LDS returns in
                    // order), so the step has one s_waitcnt instead of eight counted ones (each is an issue slot)
#pragma unroll
                    for (int jj = 7; jj >= 0; --jj) {
                        acc.x += v[jj].x; acc.y += v[jj].y; acc.z += v[jj].z; acc.w += v[jj].w;
                    }
                }
                if (BWD) {
                    float* o = out + (row0 + node) * a.ld_out + c0;
                    if (cb > 0 && lo < hi) {           // a run continued from the previous id chunk
                        const float4 old = tipk_ld4(o);
                        acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
                    }
                    if (cb == 0 || lo < hi) tipk_st4(o, acc);
                } else if (lo < hi) {
                    acc.x += old_acc.x; acc.y += old_acc.y; acc.z += old_acc.z; acc.w += old_acc.w;
                    tipk_st4(accl + node * dc + c0, acc);
                }
            }
#ifdef TIPK_DEBUG
            st_loop += __builtin_readcyclecounter() - st_l0;
#endif
        }
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
        (void)st_pro;
        if (w < 512 * 16) {
            const unsigned long long now = __builtin_readcyclecounter();
            tipk_rg_stamps[w * 8 + 0] = now - st_t0;
            tipk_rg_stamps[w * 8 + 1] = st_loop;
            tipk_rg_stamps[w * 8 + 2] = st_commit;
            tipk_rg_stamps[w * 8 + 3] = st_wait;
            tipk_rg_stamps[w * 8 + 4] = st_first;
            tipk_rg_stamps[w * 8 + 5] = now - st_epi;
            tipk_rg_stamps[w * 8 + 6] = st_pref;
            tipk_rg_stamps[w * 8 + 7] = st_reload;
        }
    }
#endif
}

inline int64_t rel_gather_lds(int64_t n_nodes, int dc, bool bwd, int chunk) {
    return (bwd ? 1 : 2) * (n_nodes + 1) * dc * 4 + (bwd ? 0 : n_nodes * dc * 4) + ((n_nodes + 3) & ~3LL) * 4 +
           ((n_nodes + 7) & ~7LL) * 2 + (int64_t)chunk * 2 + RG_META * 8 * 4;
}

constexpr int64_t RG_LDS_LIMIT = 158 * 1024;

// The launch shape: `split` column blocks (grid.y), `occ` workgroups per CU the LDS footprint
// admits (1 or 2), `chunk` ids staged per pass.  The kernel is a chain of dependent LDS round trips at
// 4 waves per SIMD (PMC: half of the wave cycles wait, profiles/r02a_lds.json); cutting the columns
// finer lets TWO 1024-thread workgroups share a CU (8 waves per SIMD; <= 64 VGPRs) at the price of
// walking the run tables once more per extra column block -- `want_occ` (option "rg_occupancy",
// default measured best) picks.  split = 0: shape not supported.
struct RgShape { int split, occ, chunk; };
inline RgShape rel_gather_shape(int64_t n_nodes, int d, bool bwd, int want_occ) {
    RgShape none = {0, 0, 0};
    if (n_nodes <= 0 || n_nodes > 1024 || d < 4 || d > 256 || (d & (d - 1)) != 0) return none;
    for (int occ = want_occ >= 2 ? 2 : 1; occ >= 1; --occ)
        for (int split = 1; d / split >= 4; split *= 2) {
            const int dc = d / split;
            if (dc > 64) continue;                          // L = dc/4 <= 16 lanes per slot
            if (n_nodes * (dc / 4) > 8192) continue;       // table rows are prefetched in 8 float4 per thread
            if (occ == 2 && dc < 8) continue;               // 4-column blocks: the run tables dominate
            for (int chunk = RG_CHUNK_MAX; chunk >= 8192; chunk /= 2)
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
extern "C" int tipk_debug_rg_stamps(unsigned long long* host_out /* [512*16*4] */) {
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
                               const uint16_t* idx, int idx_unit, const int32_t* runs, const uint16_t* node_at,
                               const float* row_scale, float* out, int64_t ld_out, tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table || !wg_unit_ptr || !unit_meta || !idx || !runs || !node_at || !out ||
        (reinterpret_cast<uintptr_t>(idx) & 15))
        return TIPK_EINVAL;
    const RgShape shape = rel_gather_shape(n_nodes, d, backward != 0, rg_want_occ());
    const int split = shape.split;
    if (split == 0) return TIPK_EUNSUPPORTED;
    if (ld_table % 4 != 0 || ld_out % 4 != 0 || (reinterpret_cast<uintptr_t>(table) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return TIPK_EINVAL;
    RgArgs a;
    a.table = table; a.ld_t = ld_table; a.n_nodes = (int)n_nodes; a.dc = d / split;
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
