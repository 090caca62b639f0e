// Segmented gather-sum: the sparse aggregation of all three TIP graphs (P-P GCN, P->D mean, D-D
// R-GCN) and of their backward passes.  See include/tipk.h section 1 for the contract and
// DESIGN.md "Kernels / gather_sum" for the roofline.
//
// Mapping (gfx950, wave = 64 lanes): a *slot* of L lanes owns one work item (<= chunk edges of one
// output row) and keeps its running sum in registers, 4 floats (one dwordx4) per lane, so one
// gathered feature row is ONE coalesced L*16-byte read.  A wave holds 64/L slots.  The slot's
// lanes fetch L edge ids with one coalesced load and hand them round with ds_bpermute (`__shfl`),
// so per 64 edges a wave issues 1 index load + 64/L... row loads of full width and no atomics at
// all: rows split over several items are combined in slot order by `gather_sum_finalize`
// (bitwise reproducible).  Items are pre-sorted by length so the slots of a wave finish together.
#include "tipk_common.h"
#include "tipk_slabs.h"

namespace {

// items[:, 3]: how a work item delivers its sum (tip_amd/plan.py)
constexpr int ITEM_DIRECT = 1;      // the whole row: epilogue + store
constexpr int ITEM_PIECE = 2;       // piece of a row combined inside the workgroup
constexpr int ITEM_LEADER = 4;      // first piece: adds the next (flags >> 8) - 1 slots and stores
constexpr int ITEM_NULL = 8;        // padding of a block
                                    // 0: piece of a row combined by tipk_gather_sum_finalize

// Ordered slab sums that are ready when a grouped gather is launched ride in it as further 1024-thread workgroups
// (tipk_gather_sum_riders): a dependent 4-us launch less each.
constexpr int GS_RIDERS = 3;
struct Riders {
    int count;
    int first[GS_RIDERS + 1];          // first[0] = workgroups of the gather itself
    SlabArgs s[GS_RIDERS];
};

struct Epilogue {
    const float* row_scale;
    const float* bias;
    int relu;
    // LIN kernels only (tipk_gather_sum_lin): out2[row, 0 .. d2) = relu2?(finished row . w^T + bias2), w element (o, i) at
    // w[o * w_so + i * w_si]
    const float* w; int64_t w_so, w_si;
    const float* bias2; int relu2;
    float* out2; int64_t ld_out2;
    // tipk_gather_sum_riders only: gate [n_out x d] (row stride ld_gate): finished value = gate > 0 ? value : 0 -- the ReLU backward
    // of the layer whose output the rows are gradients of; colsum [workgroups x d]: column sums of the workgroup's finished rows
    const float* gate; int64_t ld_gate;
    float* colsum;
};

// A linear map of the FINISHED row, applied by the L lanes that hold it (4 columns each, d == 4 L): every lane forms its
// 4-column share of all N = L * PER outputs, then a reduce-scatter over the slot (log2 L exchange steps, the kept half
// halves every step: N - PER shuffles instead of N log2 L) leaves PER complete outputs in every lane -- fixed order.
constexpr int lin_log2(int x) { return x <= 1 ? 0 : 1 + lin_log2(x / 2); }

template <int N> struct LinVec { typedef float type __attribute__((ext_vector_type(N))); };

template <int L, int PER>
__device__ __forceinline__ void lin_store(const float4& v, const Epilogue& ep, const float* wl, const float (&b2)[PER], int64_t row, int col, int sub) {
    constexpr int N = L * PER;
    static_assert(N % 4 == 0, "outputs are read four at a time");
    // w sits in LDS as wl[input column][N + 4] (staged by the whole workgroup before the launch's barrier): global loads
    // inside this exec-masked branch were waited for one by one -- 64 dependent round trips, 16 us for the 3 640 rows of conv2
    // (a register VECTOR, constant element numbers only: with `float p[N]` hipcc folded `hi ? p[j] : p[j + half]` into ONE
    // load at a lane-dependent index and expanded that into compare / select chains over all N elements -- +6.4 us)
    typename LinVec<N>::type p;
    const float vc[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* wr = wl + (col + c) * (N + 4);
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(wr + 4 * q);
            if (c == 0) { p[4 * q] = vc[0] * t.x; p[4 * q + 1] = vc[0] * t.y; p[4 * q + 2] = vc[0] * t.z; p[4 * q + 3] = vc[0] * t.w; }
            else {
                p[4 * q] = fmaf(vc[c], t.x, p[4 * q]); p[4 * q + 1] = fmaf(vc[c], t.y, p[4 * q + 1]);
                p[4 * q + 2] = fmaf(vc[c], t.z, p[4 * q + 2]); p[4 * q + 3] = fmaf(vc[c], t.w, p[4 * q + 3]);
            }
        }
    }
    int base = 0;
#pragma unroll
    for (int st = 0; st < lin_log2(L); ++st) {
        const int o = 1 << st;
        const int half = N >> (st + 1);
        const bool hi = (sub & o) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const float lo_v = p[j], hi_v = p[j + half];
            const float send = hi ? lo_v : hi_v;
            const float keep = hi ? hi_v : lo_v;
            p[j] = keep + __shfl_xor(send, o, 64);
        }
        base += hi ? half : 0;
    }
    // (base == lin_base<L, PER>(sub): the caller loaded b2 = bias2[base ..] before any divergence -- a global load in here is
    // waited for on the spot)
    float* o2 = ep.out2 + row * ep.ld_out2 + base;
    float r[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        r[k] = p[k] + b2[k];
        if (ep.relu2) r[k] = fmaxf(r[k], 0.f);
    }
    if constexpr (PER == 4) tipk_st4(o2, make_float4(r[0], r[1], r[2], r[3]));
    else if constexpr (PER == 2) *reinterpret_cast<float2*>(o2) = make_float2(r[0], r[1]);
    else o2[0] = r[0];
}

// first output a lane of the slot ends up with after lin_store's reduce-scatter
template <int L, int PER>
__device__ __forceinline__ int lin_base(int sub) {
    int base = 0;
#pragma unroll
    for (int st = 0; st < lin_log2(L); ++st)
        if (sub & (1 << st)) base += (L * PER) >> (st + 1);
    return base;
}

template <int V>
struct Acc;
template <>
struct Acc<4> {
    float4 v;
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void fma_row(const float* p, float w) {
        float4 t = tipk_ld4(p);
        v.x = fmaf(w, t.x, v.x); v.y = fmaf(w, t.y, v.y); v.z = fmaf(w, t.z, v.z); v.w = fmaf(w, t.w, v.w);
    }
    __device__ __forceinline__ void add_row(const float* p) {
        float4 t = tipk_ld4(p);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    __device__ __forceinline__ void add(const Acc& o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
    __device__ __forceinline__ void load(const float* p) { v = tipk_ld4(p); }
    __device__ __forceinline__ void load_off(const float* base, unsigned byte_off) {   // scalar base + 32-bit lane offset
        v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
    }
    __device__ __forceinline__ void fma_acc(const Acc& o, float w) {
        v.x = fmaf(w, o.v.x, v.x); v.y = fmaf(w, o.v.y, v.y); v.z = fmaf(w, o.v.z, v.z); v.w = fmaf(w, o.v.w, v.w);
    }
    __device__ __forceinline__ void add_shfl_xor(int o) {
        v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
    }
    __device__ __forceinline__ void epilogue(const Epilogue& ep, int row, int col) {
        if (ep.row_scale) { float s = ep.row_scale[row]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
        if (ep.bias) { float4 b = tipk_ld4(ep.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (ep.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (ep.gate) {
            const float4 g = tipk_ld4(ep.gate + (int64_t)row * ep.ld_gate + col);
            v.x = g.x > 0.f ? v.x : 0.f; v.y = g.y > 0.f ? v.y : 0.f; v.z = g.z > 0.f ? v.z : 0.f; v.w = g.w > 0.f ? v.w : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* p) const { tipk_st4(p, v); }
};
template <>
struct Acc<1> {
    float v;
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ void fma_row(const float* p, float w) { v = fmaf(w, *p, v); }
    __device__ __forceinline__ void add_row(const float* p) { v += *p; }
    __device__ __forceinline__ void add(const Acc& o) { v += o.v; }
    __device__ __forceinline__ void load(const float* p) { v = *p; }
    __device__ __forceinline__ void load_off(const float* base, unsigned byte_off) {
        v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
    }
    __device__ __forceinline__ void fma_acc(const Acc& o, float w) { v = fmaf(w, o.v, v); }
    __device__ __forceinline__ void add_shfl_xor(int o) { v += __shfl_xor(v, o); }
    __device__ __forceinline__ void epilogue(const Epilogue& ep, int row, int col) {
        if (ep.row_scale) v *= ep.row_scale[row];
        if (ep.bias) v += ep.bias[col];
        if (ep.relu) v = fmaxf(v, 0.f);
        if (ep.gate && !(ep.gate[(int64_t)row * ep.ld_gate + col] > 0.f)) v = 0.f;
    }
    __device__ __forceinline__ void store(float* p) const { *p = v; }
};

// GROUPED: the workgroup is exactly one block of G = blockDim / L items of a `group_slots` plan.
// SMALL: the table is < 4 GB: a gathered row's address is the (scalar) table base + a 32-bit byte offset =
// ONE 24-bit or 32-bit multiply per row; the general path pays a 64-bit multiply-add (three quarter-rate
// v_mul_lo_u32 / v_mad_u64_u32 and two adds) per row, which made the narrow-row launches (P-P graph: 32- and
// 16-float rows, 8 rows per wave-instruction) issue-bound rather than L2-bound.
template <int V, int L, bool HAS_W, bool GROUPED, bool SMALL, int LIN = 0>   // LIN = outputs per lane of the fused linear map (0: none)
__global__ __launch_bounds__(GROUPED ? 1024 : 256) void gather_sum_kernel(
    const float* __restrict__ table, int64_t ld_table, const int32_t* __restrict__ row_id,
    const float* __restrict__ edge_w, const int4* __restrict__ items, int64_t n_items,
    float* __restrict__ out, int64_t ld_out, float* __restrict__ partial, Epilogue ep, int d, Riders rd) {
    if (GROUPED && rd.count > 0 && (int)blockIdx.x >= rd.first[0]) {      // a rider workgroup: one block of an ordered slab sum
        extern __shared__ __attribute__((aligned(16))) unsigned char rider_raw[];
        SlabArgs sa = rd.s[0];
        int first = rd.first[0];
#pragma unroll
        for (int q = 1; q < GS_RIDERS; ++q)
            if (q < rd.count && (int)blockIdx.x >= rd.first[q]) { sa = rd.s[q]; first = rd.first[q]; }
        slab_sum_body(sa, (int)blockIdx.x - first, reinterpret_cast<float*>(rider_raw));
        return;
    }
    constexpr int SLOTS = TIPK_WAVE / L;
    constexpr int U = 8;                                   // row loads kept in flight per lane
    constexpr int IPL = L < U ? U / L : 1;                 // edge ids held per lane: narrow slots (d <= 16) used to
    constexpr int STEP = L * IPL;                          // keep only L rows in flight -> twice the dependent batches
    const int lane = tipk_lane();
    const int sub = lane & (L - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / TIPK_WAVE;
    const int64_t w = wave * SLOTS + lane / L;
    const bool active = w < n_items;
    int4 it = make_int4(0, 0, 0, 1);
    if (active) it = items[w];
    const int col = sub * V;
    const bool col_ok = col < d;
    const unsigned ldb = (unsigned)ld_table * 4u;          // SMALL: row stride in bytes (table bytes < 2^32)
    const float* table_c = table + (col_ok ? col : 0);

    Acc<V> acc;
    acc.zero();
    if constexpr (LIN > 0) {
        // the dense map's matrix -> LDS, behind the pieces' combine buffer: wl[input column][L LIN + 4] (ordered by the barrier
        // every thread of a grouped launch passes before a row is finished)
        extern __shared__ __attribute__((aligned(16))) unsigned char lin_raw[];
        float* wl = reinterpret_cast<float*>(lin_raw + (size_t)blockDim.x * sizeof(Acc<V>));
        constexpr int N = L * LIN;
        for (int i = threadIdx.x; i < d * N; i += blockDim.x) {
            const int ii = i / N, o = i - ii * N;
            wl[ii * (N + 4) + o] = ep.w[(int64_t)o * ep.w_so + (int64_t)ii * ep.w_si];
        }
    }
    float lin_b[LIN > 0 ? LIN : 1];
    if constexpr (LIN > 0) {
#pragma unroll
        for (int k = 0; k < LIN; ++k) lin_b[k] = ep.bias2 ? ep.bias2[lin_base<L, LIN>(sub) + k] : 0.f;
    }
    // software-pipelined ids: the slot's next STEP edge ids are requested before the current rows;
    // lane `sub` holds the ids of edges e0 + q * L + sub (coalesced per q)
    int id_next[IPL];
    float w_next[IPL];
#pragma unroll
    for (int q = 0; q < IPL; ++q) {
        id_next[q] = -1;
        w_next[q] = 0.f;
        const int e = it.x + q * L + sub;
        if (e < it.y) {
            id_next[q] = row_id[e];
            if (HAS_W) w_next[q] = edge_w[e];
        }
    }
    for (int e0 = it.x; e0 < it.y; e0 += STEP) {
        int id[IPL];
        float wgt[IPL];
#pragma unroll
        for (int q = 0; q < IPL; ++q) {
            id[q] = id_next[q];
            wgt[q] = w_next[q];
            const int nxt = e0 + STEP + q * L + sub;
            id_next[q] = -1;
            if (nxt < it.y) {
                id_next[q] = row_id[nxt];
                if (HAS_W) w_next[q] = edge_w[nxt];
            }
        }
#pragma unroll 1                          // one batch of U rows in flight: unrolling L/U batches cost 167 VGPRs at L = 32
        for (int j0 = 0; j0 < STEP; j0 += U) {
            int ids[U];
            float ws[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (IPL == 1) {
                    ids[j] = __shfl(id[0], j0 + j, L);
                    if (HAS_W) ws[j] = __shfl(wgt[0], j0 + j, L);
                } else {                                   // STEP == U: edge j of the step sits in register j / L of lane j % L
                    ids[j] = __shfl(id[j / L], j % L, L);
                    if (HAS_W) ws[j] = __shfl(wgt[j / L], j % L, L);
                }
            }
            // all row loads of the batch are issued before the first use: a load whose result is
            // consumed inside its own exec-masked branch is waited for on the spot (U dependent memory
            // round trips per batch).  Lanes without an edge skip the load (short rows -- the backward
            // plans average 2.5 edges per row -- must not pay for 8).
            Acc<V> rows[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                rows[j].zero();
                if (ids[j] >= 0 && col_ok) {
                    if (SMALL) rows[j].load_off(table_c, (unsigned)ids[j] * ldb);
                    else rows[j].load(table + col + (int64_t)ids[j] * ld_table);
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (HAS_W) acc.fma_acc(rows[j], ws[j]); else acc.add(rows[j]);
            }
        }
    }
    const int fl = active ? it.w : ITEM_NULL;
    extern __shared__ __attribute__((aligned(16))) unsigned char comb_raw[];
    bool finished = false;                              // this slot holds a complete row (ONE epilogue site for both cases)
    if (GROUPED) {
        // pieces of a split row sit in consecutive slots of this workgroup: the leader adds them in
        // slot order through LDS (fixed order -> reproducible; no partial buffer, no second launch)
        Acc<V>* comb = reinterpret_cast<Acc<V>*>(comb_raw);
        if (fl & ITEM_PIECE) comb[threadIdx.x] = acc;
        __syncthreads();
        if ((fl & ITEM_LEADER) && col_ok) {
            const int cnt = fl >> 8;
            for (int j = 1; j < cnt; ++j) acc.add(comb[threadIdx.x + j * L]);
            finished = true;
        }
    }
    if constexpr (LIN == -1 && V == 4 && GROUPED) {
        // COLUMN SUMS of the workgroup's finished rows next to the rows themselves (the bias gradient of the layer the rows
        // are gradients of, behind the gate): no early exits -- slot order, two levels through the combine buffer
        const bool fin = finished || (col_ok && !(fl & (ITEM_PIECE | ITEM_NULL)) && (fl & ITEM_DIRECT));
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fin) {
            acc.epilogue(ep, it.z, col);
            acc.store(out + (int64_t)it.z * ld_out + col);
            cs = acc.v;
        }
        Acc<V>* comb = reinterpret_cast<Acc<V>*>(comb_raw);
        __syncthreads();                                // (the leaders have read the pieces)
        comb[threadIdx.x].v = cs;
        __syncthreads();
        constexpr int PARTS = 16;
        const int per = ((int)blockDim.x / L) / PARTS;  // slots per part (1024 threads: a multiple of 16 slots)
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)threadIdx.x < PARTS * L) {
            const int p = (int)threadIdx.x / L;
            for (int q = p * per; q < (p + 1) * per; ++q) { const float4 t4 = comb[q * L + sub].v; s1.x += t4.x; s1.y += t4.y; s1.z += t4.z; s1.w += t4.w; }
        }
        __syncthreads();
        if ((int)threadIdx.x < PARTS * L) comb[threadIdx.x].v = s1;
        __syncthreads();
        if ((int)threadIdx.x < L && col_ok) {
            float4 tot = comb[sub].v;
            for (int p = 1; p < PARTS; ++p) { const float4 t4 = comb[p * L + sub].v; tot.x += t4.x; tot.y += t4.y; tot.z += t4.z; tot.w += t4.w; }
            tipk_st4(ep.colsum + (int64_t)blockIdx.x * d + col, tot);
        }
        return;
    }
    if (!finished) {
        if (!col_ok || (fl & (ITEM_PIECE | ITEM_NULL))) return;
        if (!(fl & ITEM_DIRECT)) {
            acc.store(partial + (int64_t)it.z * d + col);
            return;
        }
    }
    acc.epilogue(ep, it.z, col);
    acc.store(out + (int64_t)it.z * ld_out + col);
    if constexpr (LIN > 0 && V == 4)                    // (d == 4 L: all lanes of the slot are here)
        lin_store<L, LIN>(acc.v, ep, reinterpret_cast<const float*>(comb_raw + (size_t)blockDim.x * sizeof(Acc<V>)), lin_b, it.z, col, sub);
}

// One WAVE per split row (4 rows per workgroup): the wave's 64/LPR lane groups add the row's slots
// g, g+G, g+2G, ... in order, then the groups are combined with a fixed xor-shuffle tree, so the result
// does not depend on scheduling.  (A workgroup per row cost 9 000 tiny workgroups on the P-P graph.)
template <int V>
__global__ __launch_bounds__(256) void gather_sum_finalize_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ rows, int64_t n_rows, float* __restrict__ out,
    int64_t ld_out, Epilogue ep, int d, int lanes_per_row) {
    const int lane = tipk_lane();
    const int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / TIPK_WAVE;
    if (m >= n_rows) return;
    const int sub = lane % lanes_per_row;
    const int part = lane / lanes_per_row;
    const int n_part = TIPK_WAVE / lanes_per_row;
    const int col = sub * V;
    const int row = rows[3 * m + 0], s0 = rows[3 * m + 1], s1 = rows[3 * m + 2];
    Acc<V> acc;
    acc.zero();
    const bool ok = col < d;
    if (ok)
        for (int s = s0 + part; s < s1; s += n_part) acc.add_row(partial + (int64_t)s * d + col);
    for (int o = lanes_per_row; o < TIPK_WAVE; o <<= 1) acc.add_shfl_xor(o);
    if (part == 0 && ok) {
        acc.epilogue(ep, row, col);
        acc.store(out + (int64_t)row * ld_out + col);
    }
}

// Rows cut into only a few slots (P-P: 2-4): one L-lane slot per row, 64/L rows per wave.
template <int V, int L>
__global__ __launch_bounds__(256) void gather_sum_finalize_small_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ rows, int64_t n_rows, float* __restrict__ out,
    int64_t ld_out, Epilogue ep, int d) {
    const int lane = tipk_lane();
    const int sub = lane & (L - 1);
    const int64_t m = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / TIPK_WAVE) * (TIPK_WAVE / L) + lane / L;
    const int col = sub * V;
    if (m >= n_rows || col >= d) return;
    const int row = rows[3 * m + 0], s0 = rows[3 * m + 1], s1 = rows[3 * m + 2];
    Acc<V> acc;
    acc.zero();
    for (int s = s0; s < s1; ++s) acc.add_row(partial + (int64_t)s * d + col);
    acc.epilogue(ep, row, col);
    acc.store(out + (int64_t)row * ld_out + col);
}

template <int V, int L>
int launch_finalize_small(const float* partial, const int32_t* rows, int64_t n_rows, float* out, int64_t ld_out,
                          Epilogue ep, int d, hipStream_t st) {
    const int64_t waves = tipk_ceil_div(n_rows, TIPK_WAVE / L);
    const int64_t blocks = tipk_ceil_div(waves, 4);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL((gather_sum_finalize_small_kernel<V, L>), dim3((unsigned)blocks), dim3(256), 0, st, partial,
                       rows, n_rows, out, ld_out, ep, d);
    TIPK_RETURN_LAUNCH();
}

template <int V, int L>
int launch_gather(const float* table, int64_t ld_table, const int32_t* row_id, const float* edge_w,
                  const int32_t* items, int64_t n_items, float* out, int64_t ld_out, float* partial,
                  Epilogue ep, int d, int group_slots, bool small, hipStream_t st, Riders rd = Riders{}) {
    constexpr int SLOTS = TIPK_WAVE / L;
    const int4* it4 = reinterpret_cast<const int4*>(items);
    if (group_slots > 0) {                              // one workgroup = one block of the plan
        const int threads = group_slots * L;
        if (threads > 1024 || threads % TIPK_WAVE != 0) return TIPK_EUNSUPPORTED;
        const int64_t blocks = tipk_ceil_div(n_items, group_slots);
        if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
        const size_t lds = (size_t)threads * sizeof(Acc<V>);
        int64_t grid = blocks;
        if (rd.count > 0) {                             // riders: 1024-thread workgroups behind the gather's own
            if (threads != 1024) return TIPK_EUNSUPPORTED;
            const int rider_blocks = rd.first[rd.count];        // (filled relative to 0 by the caller)
            for (int q = 0; q <= rd.count; ++q) rd.first[q] += (int)blocks;
            grid += rider_blocks;
            if (grid > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
        }
#define TIPK_GS_LAUNCH(W, G, S, GRID, BLOCK, LDS)                                                             \
        hipLaunchKernelGGL((gather_sum_kernel<V, L, W, G, S>), dim3((unsigned)(GRID)), dim3(BLOCK), LDS, st, table, \
                           ld_table, row_id, edge_w, it4, n_items, out, ld_out, partial, ep, d, rd)
        if (ep.colsum) {                                // (tipk_gather_sum_riders checked: V == 4, 1024 threads, 32-bit row offsets)
            if constexpr (V == 4) {
                if (threads != 1024 || !small) return TIPK_EUNSUPPORTED;
                if (edge_w)
                    hipLaunchKernelGGL((gather_sum_kernel<V, L, true, true, true, -1>), dim3((unsigned)grid), dim3(threads), lds, st, table,
                                       ld_table, row_id, edge_w, it4, n_items, out, ld_out, partial, ep, d, rd);
                else
                    hipLaunchKernelGGL((gather_sum_kernel<V, L, false, true, true, -1>), dim3((unsigned)grid), dim3(threads), lds, st, table,
                                       ld_table, row_id, edge_w, it4, n_items, out, ld_out, partial, ep, d, rd);
                TIPK_RETURN_LAUNCH();
            } else {
                return TIPK_EUNSUPPORTED;
            }
        }
        if (edge_w) { if (small) TIPK_GS_LAUNCH(true, true, true, grid, threads, lds); else TIPK_GS_LAUNCH(true, true, false, grid, threads, lds); }
        else { if (small) TIPK_GS_LAUNCH(false, true, true, grid, threads, lds); else TIPK_GS_LAUNCH(false, true, false, grid, threads, lds); }
        TIPK_RETURN_LAUNCH();
    }
    if (ep.colsum) return TIPK_EUNSUPPORTED;
    if (rd.count > 0) return TIPK_EUNSUPPORTED;         // riders need the 1024-thread workgroups of a grouped plan
    const int64_t waves = tipk_ceil_div(n_items, SLOTS);
    const int64_t blocks = tipk_ceil_div(waves, 4);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    if (edge_w) { if (small) TIPK_GS_LAUNCH(true, false, true, blocks, 256, 0); else TIPK_GS_LAUNCH(true, false, false, blocks, 256, 0); }
    else { if (small) TIPK_GS_LAUNCH(false, false, true, blocks, 256, 0); else TIPK_GS_LAUNCH(false, false, false, blocks, 256, 0); }
#undef TIPK_GS_LAUNCH
    TIPK_RETURN_LAUNCH();
}

inline int pow2_at_least(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int gather_sum_impl(const float* table, int64_t ld_table, int64_t n_table, const int32_t* row_id,
                           const float* edge_w, const int32_t* items, int64_t n_items, float* out,
                           int64_t ld_out, float* partial, const float* row_scale, const float* bias,
                           int relu, int d, int group_slots, const Riders& rd, tipk_stream_t stream,
                           const float* gate = nullptr, int64_t ld_gate = 0, float* colsum = nullptr) {
    if (n_items < 0 || d <= 0 || group_slots < 0 || !items || !out || (n_items > 0 && (!table || !row_id)))
        return TIPK_EINVAL;
    if (n_items == 0) return TIPK_OK;
    if (!aligned16(items)) return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Epilogue ep{row_scale, bias, relu, nullptr, 0, 0, nullptr, 0, nullptr, 0, gate, ld_gate, colsum};
    if (gate && (ld_gate % 4 != 0 || !aligned16(gate))) return TIPK_EINVAL;
    if (colsum && (d % 4 != 0 || !aligned16(colsum))) return TIPK_EUNSUPPORTED;
    const bool small = n_table > 0 && ld_table > 0 && n_table * ld_table * 4 < (1LL << 32);   // 32-bit row offsets
    const bool vec = d % 4 == 0 && ld_table % 4 == 0 && ld_out % 4 == 0 && aligned16(table) && aligned16(out) &&
                     (!partial || aligned16(partial)) && (!bias || aligned16(bias));
#define TIPK_GS(V, L) \
    return launch_gather<V, L>(table, ld_table, row_id, edge_w, items, n_items, out, ld_out, partial, ep, d, \
                               group_slots, small, st, rd)
    if (vec) {
        if (d > 256) return TIPK_EUNSUPPORTED;
        switch (pow2_at_least(d / 4)) {
            case 1: TIPK_GS(4, 1);
            case 2: TIPK_GS(4, 2);
            case 4: TIPK_GS(4, 4);
            case 8: TIPK_GS(4, 8);
            case 16: TIPK_GS(4, 16);
            case 32: TIPK_GS(4, 32);
            default: TIPK_GS(4, 64);
        }
    }
    if (d > 64) return TIPK_EUNSUPPORTED;
    switch (pow2_at_least(d)) {
        case 1: TIPK_GS(1, 1);
        case 2: TIPK_GS(1, 2);
        case 4: TIPK_GS(1, 4);
        case 8: TIPK_GS(1, 8);
        case 16: TIPK_GS(1, 16);
        case 32: TIPK_GS(1, 32);
        default: TIPK_GS(1, 64);
    }
#undef TIPK_GS
}

extern "C" int tipk_gather_sum(const float* table, int64_t ld_table, int64_t n_table, const int32_t* row_id,
                               const float* edge_w, const int32_t* items, int64_t n_items, float* out,
                               int64_t ld_out, float* partial, const float* row_scale, const float* bias,
                               int relu, int d, int group_slots, tipk_stream_t stream) {
    return gather_sum_impl(table, ld_table, n_table, row_id, edge_w, items, n_items, out, ld_out, partial, row_scale, bias,
                           relu, d, group_slots, Riders{}, stream);
}

extern "C" int tipk_gather_sum_riders_supported(int d, int group_slots) {
    if (group_slots <= 0 || d <= 0) return 0;
    const int lanes = d % 4 == 0 ? pow2_at_least(d / 4) : pow2_at_least(d);
    return group_slots * lanes == 1024;
}

extern "C" int tipk_gather_sum_riders(const float* table, int64_t ld_table, int64_t n_table, const int32_t* row_id,
                                      const float* edge_w, const int32_t* items, int64_t n_items, float* out,
                                      int64_t ld_out, const float* row_scale, const float* bias, int relu, int d,
                                      int group_slots, const float* gate, int64_t ld_gate, float* colsum,
                                      const tipk_slab_sum_desc* sums, int32_t n_sums, tipk_stream_t stream) {
    if (n_sums < 0 || n_sums > GS_RIDERS || (n_sums > 0 && !sums)) return TIPK_EINVAL;
    if (!tipk_gather_sum_riders_supported(d, group_slots) || n_items <= 0) return TIPK_EUNSUPPORTED;
    Riders rd;
    rd.count = 0;
    int64_t blocks = 0;
    for (int i = 0; i < n_sums; ++i) {
        int rc;
        const int64_t nb = fill_slab_args(sums[i], rd.s[rd.count], &rc);
        if (rc != TIPK_OK) return rc;
        if (nb == 0) continue;
        rd.first[rd.count] = (int)blocks;
        blocks += nb;
        if (blocks > 0x3fffffffLL) return TIPK_EUNSUPPORTED;
        ++rd.count;
    }
    for (int i = rd.count; i <= GS_RIDERS; ++i) rd.first[i] = (int)blocks;
    return gather_sum_impl(table, ld_table, n_table, row_id, edge_w, items, n_items, out, ld_out, nullptr, row_scale, bias,
                           relu, d, group_slots, rd, stream, gate, ld_gate, colsum);
}

namespace {

template <int L, int PER>
int launch_gather_lin(const float* table, int64_t ld_table, const int32_t* row_id, const float* edge_w, const int32_t* items,
                      int64_t n_items, float* out, int64_t ld_out, const Epilogue& ep, int d, int group_slots, hipStream_t st) {
    const int threads = group_slots * L;
    if (threads > 1024 || threads % TIPK_WAVE != 0) return TIPK_EUNSUPPORTED;
    const int64_t blocks = tipk_ceil_div(n_items, group_slots);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    const size_t lds = (size_t)threads * sizeof(Acc<4>) + (size_t)d * (L * PER + 4) * sizeof(float);
    const int4* it4 = reinterpret_cast<const int4*>(items);
    if (edge_w)
        hipLaunchKernelGGL((gather_sum_kernel<4, L, true, true, true, PER>), dim3((unsigned)blocks), dim3(threads), lds, st, table,
                           ld_table, row_id, edge_w, it4, n_items, out, ld_out, (float*)nullptr, ep, d, Riders{});
    else
        hipLaunchKernelGGL((gather_sum_kernel<4, L, false, true, true, PER>), dim3((unsigned)blocks), dim3(threads), lds, st, table,
                           ld_table, row_id, edge_w, it4, n_items, out, ld_out, (float*)nullptr, ep, d, Riders{});
    TIPK_RETURN_LAUNCH();
}

}  // namespace

// d = 4 L floats per row with L in {4, 8, 16} lanes, d2 = L, 2 L or 4 L <= 16 outputs, grouped plans (rows finished inside the launch)
extern "C" int tipk_gather_sum_lin_supported(int d, int d2, int group_slots) {
    if (group_slots <= 0 || (d != 16 && d != 32 && d != 64)) return 0;
    const int L = d / 4;
    if (group_slots * L > 1024 || (group_slots * L) % TIPK_WAVE != 0) return 0;
    return (d2 == L || d2 == 2 * L || d2 == 4 * L) && d2 <= 16;          // (32 outputs per lane spill: 128 VGPRs at 1024 threads)
}

extern "C" int tipk_gather_sum_lin(const float* table, int64_t ld_table, int64_t n_table, const int32_t* row_id,
                                   const float* edge_w, const int32_t* items, int64_t n_items, float* out, int64_t ld_out,
                                   const float* row_scale, const float* w, int64_t w_so, int64_t w_si, const float* bias2,
                                   int relu2, float* out2, int64_t ld_out2, int d, int d2, int group_slots,
                                   tipk_stream_t stream) {
    if (n_items < 0 || !items || !out || !w || !out2 || (n_items > 0 && (!table || !row_id))) return TIPK_EINVAL;
    if (!tipk_gather_sum_lin_supported(d, d2, group_slots)) return TIPK_EUNSUPPORTED;
    if (n_items == 0) return TIPK_OK;
    if (!(n_table > 0 && ld_table > 0 && n_table * ld_table * 4 < (1LL << 32))) return TIPK_EUNSUPPORTED;   // 32-bit row offsets
    if (!aligned16(items) || !aligned16(table) || !aligned16(out) || ld_table % 4 != 0 || ld_out % 4 != 0) return TIPK_EINVAL;
    Epilogue ep{row_scale, nullptr, 0, w, w_so, w_si, bias2, relu2, out2, ld_out2, nullptr, 0, nullptr};
    hipStream_t st = (hipStream_t)stream;
    const int L = d / 4, per = d2 / L;
#define TIPK_GL(LL, PP) return launch_gather_lin<LL, PP>(table, ld_table, row_id, edge_w, items, n_items, out, ld_out, ep, d, group_slots, st)
    if (L == 4 && per == 1) TIPK_GL(4, 1);
    if (L == 4 && per == 2) TIPK_GL(4, 2);
    if (L == 4) TIPK_GL(4, 4);
    if (L == 8 && per == 1) TIPK_GL(8, 1);
    if (L == 8) TIPK_GL(8, 2);
    TIPK_GL(16, 1);
#undef TIPK_GL
}

extern "C" int tipk_gather_sum_finalize(const float* partial, const int32_t* rows, int64_t n_rows, float* out,
                                        int64_t ld_out, const float* row_scale, const float* bias, int relu,
                                        int d, int max_slots, tipk_stream_t stream) {
    if (n_rows < 0 || d <= 0) return TIPK_EINVAL;
    if (n_rows == 0) return TIPK_OK;
    if (!partial || !rows || !out || n_rows > 0x7fffffffLL) return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    Epilogue ep{row_scale, bias, relu, nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr};
    const bool vec = d % 4 == 0 && ld_out % 4 == 0 && aligned16(partial) && aligned16(out) && (!bias || aligned16(bias));
    if (vec && max_slots > 0 && max_slots <= 8 && d <= 128) {
        switch (pow2_at_least(d / 4)) {
            case 1: return launch_finalize_small<4, 1>(partial, rows, n_rows, out, ld_out, ep, d, st);
            case 2: return launch_finalize_small<4, 2>(partial, rows, n_rows, out, ld_out, ep, d, st);
            case 4: return launch_finalize_small<4, 4>(partial, rows, n_rows, out, ld_out, ep, d, st);
            case 8: return launch_finalize_small<4, 8>(partial, rows, n_rows, out, ld_out, ep, d, st);
            case 16: return launch_finalize_small<4, 16>(partial, rows, n_rows, out, ld_out, ep, d, st);
            default: return launch_finalize_small<4, 32>(partial, rows, n_rows, out, ld_out, ep, d, st);
        }
    }
    if (vec) {
        if (d > 256) return TIPK_EUNSUPPORTED;
        const int lpr = pow2_at_least(d / 4);
        hipLaunchKernelGGL((gather_sum_finalize_kernel<4>), dim3((unsigned)tipk_ceil_div(n_rows, 4)), dim3(256), 0, st,
                           partial, rows, n_rows, out, ld_out, ep, d, lpr);
    } else {
        if (d > 64) return TIPK_EUNSUPPORTED;
        const int lpr = pow2_at_least(d);
        hipLaunchKernelGGL((gather_sum_finalize_kernel<1>), dim3((unsigned)tipk_ceil_div(n_rows, 4)), dim3(256), 0, st,
                           partial, rows, n_rows, out, ld_out, ep, d, lpr);
    }
    TIPK_RETURN_LAUNCH();
}

namespace {

// 16-byte store that does NOT stay in the XCD's L2 (`sc1`, MI355X_MICROARCH.md "stores of each flavour"):
// the output of the transposed pass is a 10 GB write-once stream in config 5; kept in L2 it evicts the
// 5 MB table every gathered row comes from.
typedef float tipk_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
    tipk_f4 q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(q) : "memory");
}

// ---------------------------------------------------------------------------------------------
// CSR rows: out[r] = sum_{e in [row_ptr[r], row_ptr[r+1])} table[row_id[e]] for EVERY row r, rows short
// (include/tipk.h section 1c).  The transposed D-D pass of a large graph writes R*N rows of ~2.5 edges
// each (BASELINE config 5: 20 M rows, 50 M edges): as work items of the plan above that is one 16-byte
// descriptor, one dependent id fetch and one scattered 512-byte store per 2.5 gathered rows -- the launch
// is a chain of dependent round trips (4.9 ms, 0.64 of the HBM roofline).  Here a slot of L lanes takes
// RP CONSECUTIVE rows: their RP + 1 row pointers arrive with one coalesced load, the edge ids of the
// whole range are contiguous, 8 gathered rows are in flight per lane, and the RP output rows are one
// contiguous RP * d * 4-byte store stream.  No descriptors, no atomics; sums in edge order.
template <int L>
__global__ __launch_bounds__(256) void gather_rows_csr_kernel(const float* __restrict__ table, int64_t ld_table,
                                                              const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ row_id, int64_t n_out,
                                                              float* __restrict__ out, int64_t ld_out, int d, int rp) {
    constexpr int SLOTS = TIPK_WAVE / L;
    constexpr int U = L < 8 ? L : 8;                       // gathered rows in flight per lane (ids come from U lanes)
    const int lane = tipk_lane();
    const int sub = lane & (L - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / TIPK_WAVE;
    const int64_t task = wave * SLOTS + lane / L;
    const int64_t r0 = task * rp;
    if (r0 >= n_out) return;
    const int nr = (int)(n_out - r0 < rp ? n_out - r0 : rp);
    const int col = sub * 4;
    const bool col_ok = col < d;
    const float* table_c = table + (col_ok ? col : 0);
    const unsigned ldb = (unsigned)ld_table * 4u;
    // lane k of the slot holds row_ptr[r0 + k] (k <= nr <= L - 1): one coalesced load for the whole task
    const int pk = sub <= nr ? sub : nr;
    const int myptr = row_ptr[r0 + pk];
    const int e_begin = __shfl(myptr, 0, L), e_end = __shfl(myptr, nr, L);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int row = 0;                                           // current output row (relative to r0)
    int next = __shfl(myptr, 1, L);                        // first edge of the row after it
    float* o = out + r0 * ld_out + col;
    // ids of up to U edges per batch: lane j < U of the slot loads one, all lanes read them back by
    // shuffle; the NEXT batch's ids are requested before this batch's rows (one round trip per batch)
    int idn = 0;
    if (e_begin < e_end) {
        const int ej = e_begin + (sub & (U - 1));
        idn = row_id[ej < e_end ? ej : e_end - 1];
    }
    for (int e0 = e_begin; e0 < e_end; e0 += U) {
        const int idv = idn;
        {
            const int ej = e0 + U + (sub & (U - 1));
            idn = row_id[ej < e_end ? ej : e_end - 1];                        // clamped, unconditional
        }
        // all ids first (U shuffles, one wait), then all rows; lanes beyond d read column 0 and are masked
        // at the store; a row address = scalar table base + 32-bit byte offset (host checks the table size)
        int ids[U];
#pragma unroll
        for (int j = 0; j < U; ++j) ids[j] = __shfl(idv, j, L);
        float4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j)
            v[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table_c) + (unsigned)ids[j] * ldb);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e = e0 + j;
            if (e < e_end) {
                while (e >= next) {                        // rows that end before this edge (also empty ones)
                    if (col_ok) st4_stream(o + (int64_t)row * ld_out, acc);
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    ++row;
                    next = __shfl(myptr, row + 1 < nr ? row + 1 : nr, L);
                }
                acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
        }
    }
    for (; row < nr; ++row) {                              // the last row with edges and any empty rows behind it
        if (col_ok) st4_stream(o + (int64_t)row * ld_out, acc);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int L>
int launch_rows_csr(const float* table, int64_t ld_table, const int32_t* row_ptr, const int32_t* row_id, int64_t n_out,
                    float* out, int64_t ld_out, int d, hipStream_t st) {
    const int rp = L <= 16 ? L - 1 : 16;                   // rows per slot (rp + 1 pointers on the slot's L lanes)
    const int64_t tasks = tipk_ceil_div(n_out, rp);
    const int64_t waves = tipk_ceil_div(tasks, TIPK_WAVE / L);
    const int64_t blocks = tipk_ceil_div(waves, 4);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL((gather_rows_csr_kernel<L>), dim3((unsigned)blocks), dim3(256), 0, st, table, ld_table, row_ptr,
                       row_id, n_out, out, ld_out, d, rp);
    TIPK_RETURN_LAUNCH();
}

}  // namespace

extern "C" int tipk_gather_rows_csr(const float* table, int64_t ld_table, int64_t n_table, const int32_t* row_ptr,
                                    const int32_t* row_id, int64_t n_out, float* out, int64_t ld_out, int d,
                                    tipk_stream_t stream) {
    if (n_out < 0 || d < 8 || d > 256 || d % 4 != 0) return n_out < 0 || d <= 0 ? TIPK_EINVAL : TIPK_EUNSUPPORTED;
    if (n_table <= 0 || n_table * ld_table * 4 >= (1LL << 32)) return TIPK_EUNSUPPORTED;   // 32-bit row offsets
    if (n_out == 0) return TIPK_OK;
    if (!table || !row_ptr || !out || ld_table % 4 != 0 || ld_out % 4 != 0 ||      // row_id may be NULL for E = 0
        (reinterpret_cast<uintptr_t>(table) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return TIPK_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int lanes = 2;
    while (lanes < d / 4) lanes *= 2;
    switch (lanes) {
        case 2: return launch_rows_csr<2>(table, ld_table, row_ptr, row_id, n_out, out, ld_out, d, st);
        case 4: return launch_rows_csr<4>(table, ld_table, row_ptr, row_id, n_out, out, ld_out, d, st);
        case 8: return launch_rows_csr<8>(table, ld_table, row_ptr, row_id, n_out, out, ld_out, d, st);
        case 16: return launch_rows_csr<16>(table, ld_table, row_ptr, row_id, n_out, out, ld_out, d, st);
        case 32: return launch_rows_csr<32>(table, ld_table, row_ptr, row_id, n_out, out, ld_out, d, st);
        default: return launch_rows_csr<64>(table, ld_table, row_ptr, row_id, n_out, out, ld_out, d, st);
    }
}
