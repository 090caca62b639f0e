// Wave-stream gather (include/tipk.h section 1d):  out[row] = sum_{e: row(e) = row} table[tab(e)]  for a table
// that fits in LDS, organised so that the 16 wavefronts of a workgroup never meet after the table is staged.
// Two uses on the TIP path: the TRANSPOSED D-D pass (dY_r = A_r^T g': table = g', rows = (relation, source)
// -- the sums of tipk_rel_gather(backward = 1)) and the forward D-D pass in its pair form (table = att, rows =
// (destination, source) drug pairs: tip_amd/ops.py `_RGCN.forward`).
//
// tipk_rel_gather walks work UNITS (relations): per unit the workgroup stages ids and run tables together
// and meets at two barriers.  In-kernel stamps (tools/rg_stamps.py) put 14-19 % of a wave's life into
// waiting at those barriers (the unit lasts as long as its hub node's run), 9-12 % into the staging between
// them and 3-5 % into id-chunk reloads, while the LDS pipe -- the resource that bounds the row reads -- idles.
// In the transposed pass nothing forces that structure: g' is the SAME table for every relation and every
// output row (relation, source node) is an independent sum.  So the plan (tip_amd/plan.py
// `build_stream_plan`) cuts the whole pass into per-WAVEFRONT streams of fixed-size records:
//
//   band  = one cell per slot (a slot = L lanes = one output row at a time, 64 / L slots per wavefront)
//   cell  = up to RS_PIECE steps of 8 edge ids of ONE output row: (row | steps << 24 | first << 28 | last << 29 | log2 k << 30);
//           a longer run continues in the same slot of the next band, its sum stays in registers
//   ids   = [band][step][slot][8] uint16, pre-scaled row offsets: one 16-byte load per slot and step, the
//           slots of a wavefront read one contiguous block per step
//
// Runs are sorted by length before they are dealt, so the slots of a band hold equally long pieces (no
// idle lanes behind a hub run), rows of bands are dealt to the wavefronts by longest-processing-time, and
// a wavefront just streams: next band's cell + ids are in flight (registers) while the current band is
// gathered from LDS; no LDS id buffers, no run tables, no barriers.  Rows without edges are written as
// zeros from a per-wavefront list.  Fixed order everywhere: bitwise reproducible.
#include <stdlib.h>
#include "tipk_common.h"

#ifdef TIPK_DEBUG
// debug builds only: per-wave cycle stamps of the last stream-gather launch (tools/rs_stamps.py):
// [wave][4] = { total, table staging + barrier, band loop, bands walked }
__device__ unsigned long long tipk_rs_stamps[4096 * 4];
#endif

namespace {

constexpr int RS_PIECE = 4;            // steps (8 ids each) per cell
constexpr int RS_STAGE = 10;           // float4 per thread requested at once while the table is staged
constexpr int RS_DEPTH = 2;            // bands a record is requested ahead of its use
constexpr int64_t RS_LDS_LIMIT = 158 * 1024;

struct RsArgs {
    const float* table; int64_t ld_t;
    int n_nodes, dc;                   // dc = columns of one column block
    const int32_t* wave_ptr;           // [n_waves + 1] band range of every wavefront
    const uint32_t* cells;             // [n_bands][64 / L]
    const uint16_t* ids;               // [n_bands][RS_PIECE][64 / L][8]
    const int32_t* zero_ptr;           // [n_waves + 1] range in zero_rows
    const int32_t* zero_rows;          // output rows without edges
    float* out; int64_t ld_out;
    const float* row_scale;            // g' = row_scale[node] * table[node], applied while staging (nullable)
    int idx_mul;                       // byte offset of a table row = id * idx_mul
    // epilogue of a finished row: relu?(out_scale[row] * sum + bias[col]) -- the GCN layer of the P-P graph
    // (D^-1/2 (A + I) D^-1/2 = a row scaling on either side of the plain sum); all nullable / 0
    const float* out_scale;
    const float* bias;
    int relu;
    // KIND 2 (tipk_stream_gather_parts): every workgroup stages ITS partition of a table that exists as two halves:
    // LDS row i of partition p = table[part_first[p] + i] + table[second + part_first[p] + i]
    // two tables in one launch (tipk_stream_gather_two): blockIdx.y = 1 gathers from table1 into out1 on the SAME plan
    const float* table1; float* out1;
    const int32_t* part_first;         // [n_parts] first table row of a partition
    const int32_t* wg_part;            // [n_wg] partition of a workgroup
    int64_t second;                    // rows between the two halves
};

template <typename T> using rs_lds_t = const __attribute__((address_space(3))) T;

// a lane's piece of a row: float4 (16-byte rows and wider) or float2 (8-byte rows: tables of up to 19 000 nodes)
template <int VW> struct RsVec;
template <> struct RsVec<4> {
    typedef float4 T;
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ T lds_load(unsigned addr) {             // ds_read_b128 at a 32-bit LDS byte address
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 x = *reinterpret_cast<const __attribute__((address_space(3))) f4*>((uintptr_t)addr);
        return make_float4(x.x, x.y, x.z, x.w);
    }
    static __device__ __forceinline__ void add(T& a, const T& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    static __device__ __forceinline__ void scale(T& a, float s) { a.x *= s; a.y *= s; a.z *= s; a.w *= s; }
    static __device__ __forceinline__ T shfl_down(const T& a, int d) {
        T o; o.x = __shfl_down(a.x, d, 64); o.y = __shfl_down(a.y, d, 64); o.z = __shfl_down(a.z, d, 64); o.w = __shfl_down(a.w, d, 64); return o;
    }
    static __device__ __forceinline__ void relu(T& a) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
};
template <> struct RsVec<2> {
    typedef float2 T;
    static __device__ __forceinline__ T zero() { return make_float2(0.f, 0.f); }
    static __device__ __forceinline__ T lds_load(unsigned addr) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 x = *reinterpret_cast<const __attribute__((address_space(3))) f2*>((uintptr_t)addr);
        return make_float2(x.x, x.y);
    }
    static __device__ __forceinline__ void add(T& a, const T& b) { a.x += b.x; a.y += b.y; }
    static __device__ __forceinline__ void scale(T& a, float s) { a.x *= s; a.y *= s; }
    static __device__ __forceinline__ T shfl_down(const T& a, int d) {
        T o; o.x = __shfl_down(a.x, d, 64); o.y = __shfl_down(a.y, d, 64); return o;
    }
    static __device__ __forceinline__ void relu(T& a) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); }
};

// KIND 0 / 1 only name the launch in profiles (0: rows = (relation, node), the transposed pass; 1: rows = node pairs);
// KIND 2: the table is per workgroup -- a partition of the symmetrised pair gradients (tipk_stream_gather_parts)
template <int L, bool UNIT, int KIND, int VW = 4>
__global__ __launch_bounds__(1024) void stream_gather_kernel(RsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tab[];        // [n_nodes + 1][dc], last row = 0 (the pad id's row)
    typedef RsVec<VW> V;
    typedef typename V::T vec_t;
    constexpr int SPW = 64 / L;
    constexpr int q4 = L;                                              // vectors per row of the column block
    const int t = threadIdx.x, lane = t & 63;
#ifdef TIPK_DEBUG
    const unsigned long long st0 = __builtin_readcyclecounter();
#endif
    const int n_nodes = a.n_nodes, dc = a.dc;
    const int slot = lane / L, c0 = (lane & (L - 1)) * VW;
    const bool second = a.table1 != nullptr && blockIdx.y == 1;        // (uniform)
    const int col0 = second ? 0 : blockIdx.y * dc;
    const float* table = second ? a.table1 : a.table + col0;
    float* out = second ? a.out1 : a.out + col0;
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 16 + (t >> 6));
    int b = __builtin_amdgcn_readfirstlane(a.wave_ptr[gw]);
#ifdef TIPK_DEBUG
    const int b_first = b;
#endif
    const int b1 = __builtin_amdgcn_readfirstlane(a.wave_ptr[gw + 1]);
    const unsigned tab32 = (unsigned)(uintptr_t)(rs_lds_t<float>*)(tab + c0);     // LDS byte address of this lane's piece of row 0
    const unsigned ldt4 = (unsigned)__builtin_amdgcn_readfirstlane(a.idx_mul);      // (<= 64: a 16-bit factor)

    // A band is walked in about a microsecond; its record (cell word + ids) comes from HBM and takes two under
    // load, so records are requested RS_DEPTH bands ahead (loads are unconditional: the band index is clamped
    // to the wavefront's last band).  Registers: 2 x RS_DEPTH record sets used round robin -- band j is walked
    // out of set j mod 4 while band j + 2 lands in set (j + 2) mod 4 -- so a record is never copied: rotating
    // the records through one set of names makes hipcc move registers whose loads are still in flight, i.e.
    // drain the whole queue once per trip (measured: 42 us inside the step with one record in flight, 31 alone).
    static_assert(RS_DEPTH == 2, "four record sets below");
    uint32_t c0q = 0, c1q = 0, c2q = 0, c3q = 0;
    uint4 i0q[RS_PIECE], i1q[RS_PIECE], i2q[RS_PIECE], i3q[RS_PIECE];
    auto fetch = [&](int band, uint32_t& cw, uint4 (&iw)[RS_PIECE]) {
        band = band < b1 ? band : b1 - 1;
        cw = a.cells[(int64_t)band * SPW + slot];
        const uint4* p = reinterpret_cast<const uint4*>(a.ids) + ((int64_t)band * (RS_PIECE * SPW) + slot);
#pragma unroll
        for (int k = 0; k < RS_PIECE; ++k) iw[k] = p[k * SPW];
    };
    // the first two records travel while the table is staged
    if (b < b1) {
        fetch(b, c0q, i0q);
        fetch(b + 1, c1q, i1q);
    }
    const int total4 = n_nodes * q4;
    if (t < dc) tab[(int64_t)n_nodes * dc + t] = 0.f;
    if constexpr (KIND == 2) {
        // the partition in ONE batch of requests: both halves of every piece, two contiguous streams
        constexpr int PSTAGE = 8;
        const int part = __builtin_amdgcn_readfirstlane(a.wg_part[blockIdx.x]);
        const float* ta = table + (int64_t)__builtin_amdgcn_readfirstlane(a.part_first[part]) * a.ld_t;
        const float* tb = ta + a.second * a.ld_t;
        for (int base = 0; base < total4; base += 1024 * PSTAGE) {
            vec_t ga[PSTAGE], gb[PSTAGE];
#pragma unroll
            for (int u = 0; u < PSTAGE; ++u) {
                int i = base + u * 1024 + t;
                i = i < total4 ? i : total4 - 1;
                const int64_t off = (int64_t)(i / q4) * a.ld_t + (i % q4) * VW;
                ga[u] = *reinterpret_cast<const vec_t*>(ta + off);
                gb[u] = *reinterpret_cast<const vec_t*>(tb + off);
            }
#pragma unroll
            for (int u = 0; u < PSTAGE; ++u) {
                const int i = base + u * 1024 + t;
                V::add(ga[u], gb[u]);
                if (i < total4) *reinterpret_cast<vec_t*>(tab + (i / q4) * dc + (i % q4) * VW) = ga[u];
            }
        }
    } else
    // the whole table in ONE round trip: up to RS_STAGE float4 per thread are requested before the first is stored
    // (the LDS holds at most 10 112 float4; a loop of 4-deep batches cost a dependent round trip per 64 KB)
    for (int base = 0; base < total4; base += 1024 * RS_STAGE) {
        vec_t gv[RS_STAGE];
#pragma unroll
        for (int u = 0; u < RS_STAGE; ++u) {
            int i = base + u * 1024 + t;
            i = i < total4 ? i : total4 - 1;
            const int r = i / q4, c = (i - r * q4) * VW;
            gv[u] = *reinterpret_cast<const vec_t*>(table + (int64_t)r * a.ld_t + c);
        }
        if (a.row_scale) {
#pragma unroll
            for (int u = 0; u < RS_STAGE; ++u) {
                int i = base + u * 1024 + t;
                i = i < total4 ? i : total4 - 1;
                V::scale(gv[u], a.row_scale[i / q4]);
            }
        }
#pragma unroll
        for (int u = 0; u < RS_STAGE; ++u) {
            const int i = base + u * 1024 + t;
            if (i < total4) { const int r = i / q4, c = (i - r * q4) * VW; *reinterpret_cast<vec_t*>(tab + r * dc + c) = gv[u]; }
        }
    }
    __syncthreads();                                   // the only barrier of the launch
#ifdef TIPK_DEBUG
    const unsigned long long st1 = __builtin_readcyclecounter();
#endif
    vec_t acc = V::zero();
    // a finished row: relu?(out_scale[row] * sum + bias) (all optional), one store
    vec_t bias_v = V::zero();
    if (a.bias) bias_v = *reinterpret_cast<const vec_t*>(a.bias + col0 + c0);
    auto finish = [&](int64_t row, vec_t v) {
        if (a.out_scale) V::scale(v, a.out_scale[row]);
        V::add(v, bias_v);
        if (a.relu) V::relu(v);
        *reinterpret_cast<vec_t*>(out + row * a.ld_out + c0) = v;
    };
    // walk band `band` out of (cw, iw) and request band + RS_DEPTH into (nw, niw)
    auto walk = [&](int band, const uint32_t& cw, const uint4 (&iw)[RS_PIECE], uint32_t& nw, uint4 (&niw)[RS_PIECE]) {
        const uint32_t cell = band < b1 ? cw : 0u;                             // past the end: an idle cell
        fetch(band + RS_DEPTH, nw, niw);
        __builtin_amdgcn_sched_barrier(0);
        const int len = (int)((cell >> 24) & 15u);
        if (cell & (1u << 28)) acc = V::zero();                                // first piece of its row
#pragma unroll
        for (int k = 0; k < RS_PIECE; ++k) {
            if (k < len) {
                // one step: 8 pre-scaled ids -> 8 row addresses (SDWA adds) -> 8 ds_read_b128 -> one wait, rows
                // added last to first (the last row's arrival implies the others: LDS returns in order)
                // (round 6) a row address = base + id16 * idx_mul in ONE instruction: v_mad_u32_u16 takes the 16-bit half of the
                // id word it is told to (op_sel) -- unpack (v_and / v_lshrrev) + v_mad_u32_u24 were 16 of a step's ~51 VALU
                // instructions, and the VALU time of these launches ADDS to their LDS time (profiles/r06_lds.json)
                const unsigned w4[4] = {iw[k].x, iw[k].y, iw[k].z, iw[k].w};
                unsigned ad[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    if (jj & 1) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ad[jj]) : "v"(w4[jj >> 1]), "s"(ldt4), "v"(tab32));
                    else asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(ad[jj]) : "v"(w4[jj >> 1]), "s"(ldt4), "v"(tab32));
                }
                vec_t v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = V::lds_load(ad[jj]);
#pragma unroll
                for (int jj = 7; jj >= 0; --jj) V::add(acc, v[jj]);
            }
        }
        // a WIDE run was cut into 2^klog sub-runs in adjacent (aligned) slots: on its last band the partial sums are
        // added in a fixed tree order, slot s <- slot s + 2^j, and the set's first slot holds the row
        const unsigned klog = cell >> 30;
        if (__builtin_amdgcn_ballot_w64(klog != 0u) != 0ull) {                   // wave-uniform
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                constexpr int LL = L;
                const int delta = LL << j;
                if (delta < 64) {
                    const vec_t o = V::shfl_down(acc, delta);
                    if ((int)klog > j && (slot & ((2 << j) - 1)) == 0) V::add(acc, o);
                }
            }
        }
        if (cell & (1u << 29))                                                   // the row is complete
            finish((int64_t)(cell & 0xffffffu), acc);
    };
    for (; b < b1; b += 4) {
        walk(b, c0q, i0q, c2q, i2q);
        walk(b + 1, c1q, i1q, c3q, i3q);
        walk(b + 2, c2q, i2q, c0q, i0q);
        walk(b + 3, c3q, i3q, c1q, i1q);
    }
#ifdef TIPK_DEBUG
    const int gws = gw + (int)blockIdx.y * (int)gridDim.x * 16;         // (column blocks / the second table behind the first)
    if (lane == 0 && gws < 4096) {
        const unsigned long long now = __builtin_readcyclecounter();
        tipk_rs_stamps[gws * 4 + 0] = now - st0;
        tipk_rs_stamps[gws * 4 + 1] = st1 - st0;
        tipk_rs_stamps[gws * 4 + 2] = now - st1;
        tipk_rs_stamps[gws * 4 + 3] = (unsigned long long)(b1 - b_first) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4 /* HW_ID: cu/se */) << 32);
    }
#endif
    // rows (relation, node) without edges
    if (!a.zero_ptr) return;                           // the consumer masks those rows (tipk_rgcn_dy_products row_used)
    const int z0 = __builtin_amdgcn_readfirstlane(a.zero_ptr[gw]), z1 = __builtin_amdgcn_readfirstlane(a.zero_ptr[gw + 1]);
    for (int z = z0 + slot; z < z1; z += SPW)
        finish((int64_t)a.zero_rows[z], V::zero());
}

// column blocks of the launch (grid.y): the table of one block must fit in LDS; 0 = not supported.
// max_split: every column block walks all the ids again -- 4 for the D-D passes (more blocks than that and the other
// routes win), 16 for the P-P graph (19 081 proteins: 2-column blocks of 8-byte rows, 152 KB)
inline int rel_stream_split(int64_t n_nodes, int d, int max_split) {
    if (n_nodes <= 0 || n_nodes > 65535 || d < 4 || d > 256 || (d & (d - 1)) != 0) return 0;
    for (int split = 1; split <= max_split && d / split >= (max_split > 4 ? 2 : 4); split *= 2) {
        const int dc = d / split;
        if (dc > 64) continue;
        if ((n_nodes + 1) * dc * 4 <= RS_LDS_LIMIT) return split;
    }
    return 0;
}

template <int L, int VW = 4>
int launch_rs(const RsArgs& a, int n_wg, int split, int kind, hipStream_t st) {
    const size_t lds = (size_t)(a.n_nodes + 1) * a.dc * 4;
    void (*kern)(RsArgs) = kind ? (a.idx_mul == 1 ? stream_gather_kernel<L, true, 1, VW> : stream_gather_kernel<L, false, 1, VW>)
                                : (a.idx_mul == 1 ? stream_gather_kernel<L, true, 0, VW> : stream_gather_kernel<L, false, 0, VW>);
    if constexpr (L == 8 && VW == 4) {                                 // per-workgroup tables: 128-byte rows only
        if (kind == 2) kern = a.idx_mul == 1 ? stream_gather_kernel<L, true, 2, VW> : stream_gather_kernel<L, false, 2, VW>;
    }
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg, (unsigned)split), dim3(1024), lds, st, a);
    TIPK_RETURN_LAUNCH();
}

}  // namespace

#ifdef TIPK_DEBUG
extern "C" int tipk_debug_rs_stamps(unsigned long long* host_out /* [4096 * 4] */) {
    return tipk_hip_status(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tipk_rs_stamps), sizeof(unsigned long long) * 4096 * 4));
}
#endif

extern "C" int tipk_stream_gather_supported(int64_t n_table, int d, int max_split) {
    return rel_stream_split(n_table, d, max_split < 1 ? 1 : max_split);
}

extern "C" int tipk_stream_gather_piece(void) { return RS_PIECE; }

extern "C" int tipk_stream_gather(const float* table, int64_t ld_table, int64_t n_nodes, int d, int64_t n_wg,
                                   const int32_t* wave_ptr, const uint32_t* cells, const uint16_t* ids, int idx_unit,
                                   const int32_t* zero_ptr, const int32_t* zero_rows, const float* row_scale,
                                   float* out, int64_t ld_out, int kind, int max_split, const float* out_scale,
                                   const float* bias, int relu, tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table || !wave_ptr || !cells || !ids || (zero_ptr && !zero_rows) || !out ||
        (reinterpret_cast<uintptr_t>(ids) & 15))
        return TIPK_EINVAL;
    const int split = rel_stream_split(n_nodes, d, max_split < 1 ? 1 : max_split);
    if (split == 0) return TIPK_EUNSUPPORTED;
    const int dc_ = d / split;
    const int al = dc_ >= 4 ? 4 : 2;                                   // floats per lane piece
    if (ld_table % al != 0 || ld_out % al != 0 || (reinterpret_cast<uintptr_t>(table) & (4 * al - 1)) ||
        (reinterpret_cast<uintptr_t>(out) & (4 * al - 1)) || (bias && (reinterpret_cast<uintptr_t>(bias) & (4 * al - 1))))
        return TIPK_EINVAL;
    RsArgs a;
    a.table = table; a.ld_t = ld_table; a.n_nodes = (int)n_nodes; a.dc = d / split;
    a.wave_ptr = wave_ptr; a.cells = cells; a.ids = ids; a.zero_ptr = zero_ptr; a.zero_rows = zero_rows;
    a.out = out; a.ld_out = ld_out; a.row_scale = row_scale;
    a.out_scale = out_scale; a.bias = bias; a.relu = relu;
    a.part_first = nullptr; a.wg_part = nullptr; a.second = 0; a.table1 = nullptr; a.out1 = nullptr;
    if (idx_unit <= 0 || (a.dc * 4) % idx_unit != 0 || (int64_t)n_nodes * idx_unit > 65535) return TIPK_EINVAL;
    a.idx_mul = a.dc * 4 / idx_unit;
    hipStream_t st = (hipStream_t)stream;
    if (a.dc == 2) return launch_rs<1, 2>(a, (int)n_wg, split, kind != 0, st);
    switch (a.dc / 4) {
        case 1: return launch_rs<1>(a, (int)n_wg, split, kind != 0, st);
        case 2: return launch_rs<2>(a, (int)n_wg, split, kind != 0, st);
        case 4: return launch_rs<4>(a, (int)n_wg, split, kind != 0, st);
        case 8: return launch_rs<8>(a, (int)n_wg, split, kind != 0, st);
        default: return launch_rs<16>(a, (int)n_wg, split, kind != 0, st);
    }
}

// d att of the pair-form backward pass (include/tipk.h section 2e): out[p * n_rel + r] = sum over the pairs t of partition p
// that relation r links of (table[t] + table[second + t]).  A partition's sums are staged in LDS by each of its
// workgroups (wg_part); everything else is tipk_stream_gather.
extern "C" int tipk_stream_gather_parts(const float* table, int64_t ld_table, int d, int64_t second, const int32_t* part_first,
                                         int64_t part_len, const int32_t* wg_part, int64_t n_wg, const int32_t* wave_ptr,
                                         const uint32_t* cells, const uint16_t* ids, int idx_unit, const int32_t* zero_ptr,
                                         const int32_t* zero_rows, float* out, int64_t ld_out, tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table || !part_first || !wg_part || !wave_ptr || !cells || !ids || (zero_ptr && !zero_rows) ||
        !out || (reinterpret_cast<uintptr_t>(ids) & 15) || second < 0)
        return TIPK_EINVAL;
    if (d != 32) return TIPK_EUNSUPPORTED;                             // one 128-byte row per pair (n_bases = 32)
    if (part_len <= 0 || (part_len + 1) * d * 4 > RS_LDS_LIMIT) return TIPK_EUNSUPPORTED;
    if (ld_table % 4 != 0 || ld_out % 4 != 0 || (reinterpret_cast<uintptr_t>(table) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return TIPK_EINVAL;
    RsArgs a;
    a.table = table; a.ld_t = ld_table; a.n_nodes = (int)part_len; a.dc = d;
    a.wave_ptr = wave_ptr; a.cells = cells; a.ids = ids; a.zero_ptr = zero_ptr; a.zero_rows = zero_rows;
    a.out = out; a.ld_out = ld_out; a.row_scale = nullptr;
    a.out_scale = nullptr; a.bias = nullptr; a.relu = 0;
    a.part_first = part_first; a.wg_part = wg_part; a.second = second; a.table1 = nullptr; a.out1 = nullptr;
    if (idx_unit <= 0 || (a.dc * 4) % idx_unit != 0 || part_len * idx_unit > 65535) return TIPK_EINVAL;
    a.idx_mul = a.dc * 4 / idx_unit;
    return launch_rs<8>(a, (int)n_wg, 1, 2, (hipStream_t)stream);
}

// d att of BOTH R-GCN layers in one launch (round 6): the layers share the pair-backward plan; blockIdx.y = 1 stages its
// partitions from table1 and writes out1.  One launch ramp and one tail instead of two (the gathers are off the critical
// path of the dense products between them: an encoder-level schedule runs them side by side once both tables exist).
extern "C" int tipk_stream_gather_parts_two(const float* table0, const float* table1, int64_t ld_table, int d, int64_t second,
                                             const int32_t* part_first, int64_t part_len, const int32_t* wg_part, int64_t n_wg,
                                             const int32_t* wave_ptr, const uint32_t* cells, const uint16_t* ids, int idx_unit,
                                             const int32_t* zero_ptr, const int32_t* zero_rows, float* out0, float* out1,
                                             int64_t ld_out, tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table0 || !table1 || !part_first || !wg_part || !wave_ptr || !cells || !ids ||
        (zero_ptr && !zero_rows) || !out0 || !out1 || (reinterpret_cast<uintptr_t>(ids) & 15) || second < 0)
        return TIPK_EINVAL;
    if (d != 32) return TIPK_EUNSUPPORTED;
    if (part_len <= 0 || (part_len + 1) * d * 4 > RS_LDS_LIMIT) return TIPK_EUNSUPPORTED;
    if (ld_table % 4 != 0 || ld_out % 4 != 0 || ((reinterpret_cast<uintptr_t>(table0) | reinterpret_cast<uintptr_t>(table1) |
                                                   reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(out1)) & 15))
        return TIPK_EINVAL;
    RsArgs a;
    a.table = table0; a.ld_t = ld_table; a.n_nodes = (int)part_len; a.dc = d;
    a.wave_ptr = wave_ptr; a.cells = cells; a.ids = ids; a.zero_ptr = zero_ptr; a.zero_rows = zero_rows;
    a.out = out0; a.ld_out = ld_out; a.row_scale = nullptr;
    a.out_scale = nullptr; a.bias = nullptr; a.relu = 0;
    a.part_first = part_first; a.wg_part = wg_part; a.second = second; a.table1 = table1; a.out1 = out1;
    if (idx_unit <= 0 || (a.dc * 4) % idx_unit != 0 || part_len * idx_unit > 65535) return TIPK_EINVAL;
    a.idx_mul = a.dc * 4 / idx_unit;
    return launch_rs<8>(a, (int)n_wg, 2, 2, (hipStream_t)stream);
}

// The pair cells of BOTH R-GCN layers of an encoder in one launch (include/tipk.h section 1d): the layers share the graph,
// hence the plan; a workgroup stages att of layer blockIdx.y.  (Two launches of 256 workgroups each left the chip idle
// through two launch ramps and two tails; the cells depend on the parameters only, so they can lead the step.)
extern "C" int tipk_stream_gather_two(const float* table0, const float* table1, int64_t ld_table, int64_t n_table, int d,
                                       int64_t n_wg, const int32_t* wave_ptr, const uint32_t* cells, const uint16_t* ids,
                                       int idx_unit, float* out0, float* out1, int64_t ld_out, tipk_stream_t stream) {
    if (n_wg <= 0 || n_wg > 65535 || !table0 || !table1 || !wave_ptr || !cells || !ids || !out0 || !out1 ||
        (reinterpret_cast<uintptr_t>(ids) & 15))
        return TIPK_EINVAL;
    if (rel_stream_split(n_table, d, 1) != 1 || d < 16) return TIPK_EUNSUPPORTED;      // one column block, 16-byte lane pieces
    if (ld_table % 4 != 0 || ld_out % 4 != 0 || ((reinterpret_cast<uintptr_t>(table0) | reinterpret_cast<uintptr_t>(table1) |
                                                   reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(out1)) & 15))
        return TIPK_EINVAL;
    RsArgs a;
    a.table = table0; a.ld_t = ld_table; a.n_nodes = (int)n_table; a.dc = d;
    a.wave_ptr = wave_ptr; a.cells = cells; a.ids = ids; a.zero_ptr = nullptr; a.zero_rows = nullptr;
    a.out = out0; a.ld_out = ld_out; a.row_scale = nullptr; a.out_scale = nullptr; a.bias = nullptr; a.relu = 0;
    a.part_first = nullptr; a.wg_part = nullptr; a.second = 0; a.table1 = table1; a.out1 = out1;
    if (idx_unit <= 0 || (a.dc * 4) % idx_unit != 0 || n_table * idx_unit > 65535) return TIPK_EINVAL;
    a.idx_mul = a.dc * 4 / idx_unit;
    hipStream_t st = (hipStream_t)stream;
    switch (a.dc / 4) {
        case 4: return launch_rs<4>(a, (int)n_wg, 2, 1, st);
        case 8: return launch_rs<8>(a, (int)n_wg, 2, 1, st);
        case 16: return launch_rs<16>(a, (int)n_wg, 2, 1, st);
        default: return TIPK_EUNSUPPORTED;
    }
}
