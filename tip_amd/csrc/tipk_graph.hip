// The op-level face of the library (include/tipk.h section 9): an opaque graph handle that OWNS the
// preprocessed buffers of a D-D graph, and one forward / one backward entry per R-GCN layer that take raw device pointers,
// the handle, a caller-supplied workspace and a stream -- the whole layer (reference MyRGCNConv2.forward / MyRGCNConv.forward,
// src/layers.py:157-188 / :76-99, and their autograd) behind two C calls, for a host that is not this package's Python.
//
// The handle implements the GENERIC route natively (any node count, any widths, symmetric or not): basis-first,
// transform-then-gather, explicit backward, nothing of size E x d:
//
//     forward    XB_b = X basis_b,  Y = att . XB  [R N, d_out],  out = relu?( 1/deg * sum_{e -> v} Y[(r_e, src_e)] + X root )
//     backward   g' = g (.) [out > 0],  dY[(r, u)] = sum_{e = (r, u) -> v} g'[v] / deg[v]
//                d att = dY . XB^T,  d XB = att^T . dY,  d basis_b = X^T d XB_b,  d root = X^T g',
//                dX = g' root^T + sum_b d XB_b basis_b^T
//
// tipk_graph_build sorts the edge list ON THE HOST, once (stable counting sorts: the order of additions inside a row is the
// order of the edge list -- every pass is bitwise reproducible), and keeps two CSRs on the device: rows of the output by
// destination -> rows (relation, source) of Y, and rows (relation, source) of dY -> destinations.  The gathers are a
// wavefront per output row; the dense products are the library's own MFMA GEMM (tipk_gemm_f32).
//
// The PAIR-FORM route (the one the PyTorch modules take at BioSNAP size: <= 1 024 nodes, an att table that fits in LDS) is behind
// the handle as well: tipk_graph_prepare_rgcn builds its plans on the host with tipk_pairplan.hip -- the arrays tip_amd/plan.py
// builds, bit for bit -- and tipk_rgcn_fwd / _bwd then run the wave-stream cell gather, the pair product, the pair gradients and
// the partitioned d att gather (include/tipk.h sections 1d, 2c, 2e) instead of the generic passes.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>
#include "tipk_common.h"
#include "tipk_pairplan.h"

namespace {

constexpr int PAIR_KGROUP = 8;             // source nodes per slab of tipk_pair_product

// device arrays of a wave-stream plan
struct StreamDev {
    int32_t* wave_ptr = nullptr; uint32_t* cells = nullptr; uint16_t* ids = nullptr; int32_t* zero_ptr = nullptr; int32_t* zero_rows = nullptr;
    int64_t n_wg = 0, n_bands = 0;
    int idx_unit = 1;
};

// the pair-form plans of one (n_bases, symmetric) shape: forward cells plan, link words, and -- once a width with pair
// gradients was prepared -- the backward plan with its gradient table (transient inside a backward call: owned here)
struct PairRoute {
    int n_bases = 0, lanes = 0;
    bool symmetric = false, has_bwd = false;
    StreamDev fwd;
    uint32_t* links = nullptr;
    float* zeros = nullptr;                // 64 floats of zeros (tipk_pair_product reads them in place of unlinked cells)
    int32_t* slots = nullptr; int32_t* node_desc = nullptr; int32_t* tile_node = nullptr; int32_t* part_first = nullptr; int32_t* wg_part = nullptr;
    StreamDev gather;
    int64_t n_slots = 0, n_parts = 0, part_len = 0, n_alloc = 0;
    float* pg = nullptr;                   // [2 n_alloc + 1][n_bases], zeroed once: a call rewrites the rows of the linked pairs
    std::vector<void*> owned;
};

struct HostEdges { std::vector<int32_t> src, dst, rel; };

// a grouped gather plan of tipk_gather_sum on the device (G = 128: rows of 4 .. 32 floats), built in C++ (tipk_pairplan.hip)
struct GatherDev {
    int32_t* row_id = nullptr; float* edge_w = nullptr; int32_t* items = nullptr;
    int64_t n_items = 0, n_table = 0;
    int G = 0;
};

}  // namespace

struct tipk_graph {
    int kind;              // 0: D-D relation graph (R-GCN), 1: normalised adjacency (GCNConv), 2: bipartite mean (MyHierarchyConv)
    int64_t n_nodes, n_rel, n_edges;
    int32_t* fwd_ptr;      // [n_out + 1]              edges by output row (R-GCN: destination)
    int32_t* fwd_row;      // [n_edges]                table row (R-GCN: relation * n_nodes + source, a row of Y)
    int32_t* bwd_ptr;      // [n_table + 1]            the transpose (R-GCN: by (relation, source))
    int32_t* bwd_row;      // [n_edges]
    float* inv_deg;        // R-GCN: [n_nodes] 1 / max(1, in-degree over all relations)
    float* fwd_w;          // kinds 1, 2: weight of every edge in fwd order / in bwd order (GCN norm; 1 / #edges of the target)
    float* bwd_w;
    int64_t n_out, n_table, n_source;      // kinds 1, 2: rows of the output / of the table; kind 2: first target row
    HostEdges* host;                       // kind 0: the edge list on the host until tipk_graph_release_host (plans are built from it)
    std::vector<PairRoute*>* routes;       // kind 0: prepared pair-form routes
    int symmetric_known, symmetric;        // kind 0: every relation links u -> v as often as v -> u (computed on first use)
    GatherDev* pf;                         // work-item plans of the two CSRs for rows of <= 32 floats (hub rows cut into pieces that one
    GatherDev* pb;                         // workgroup combines): the wavefront-per-row kernel below is the route for everything else
};

namespace {

struct GrArgs {
    const float* table; int64_t ld_t;
    const int32_t* ptr; const int32_t* row; int64_t n_out;
    const float* table_scale;                  // nullable: table row i is used as table_scale[i] * table[i]
    const float* out_scale;                    // nullable: per output row
    const float* addend; int64_t ld_add;       // nullable
    const float* gate; int64_t ld_gate;        // nullable: the TABLE row is masked with gate[i] > 0 (ReLU backward)
    const float* edge_w;                       // nullable: per edge (in list order)
    const float* bias;                         // nullable: per column, added before the ReLU
    int relu, d;
    float* out; int64_t ld_out;
    const GatherDev* plan;                     // nullable: the same sums as work items of tipk_gather_sum (used when it applies)
};

// one wavefront per output row, lane = column (blocks of 64 columns), the row's edges in list order, 4 rows of the table in flight
__global__ __launch_bounds__(256) void graph_rows_kernel(GrArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.n_out) return;
    const int e0 = a.ptr[r], e1 = a.ptr[r + 1];
    for (int c = lane; c < a.d; c += 64) {
        float s = 0.f;
        int e = e0;
        for (; e + 4 <= e1; e += 4) {
            const int i0 = a.row[e], i1 = a.row[e + 1], i2 = a.row[e + 2], i3 = a.row[e + 3];
            float v0 = a.table[(int64_t)i0 * a.ld_t + c], v1 = a.table[(int64_t)i1 * a.ld_t + c];
            float v2 = a.table[(int64_t)i2 * a.ld_t + c], v3 = a.table[(int64_t)i3 * a.ld_t + c];
            if (a.gate) {
                v0 = a.gate[(int64_t)i0 * a.ld_gate + c] > 0.f ? v0 : 0.f; v1 = a.gate[(int64_t)i1 * a.ld_gate + c] > 0.f ? v1 : 0.f;
                v2 = a.gate[(int64_t)i2 * a.ld_gate + c] > 0.f ? v2 : 0.f; v3 = a.gate[(int64_t)i3 * a.ld_gate + c] > 0.f ? v3 : 0.f;
            }
            if (a.table_scale) { v0 *= a.table_scale[i0]; v1 *= a.table_scale[i1]; v2 *= a.table_scale[i2]; v3 *= a.table_scale[i3]; }
            if (a.edge_w) { v0 *= a.edge_w[e]; v1 *= a.edge_w[e + 1]; v2 *= a.edge_w[e + 2]; v3 *= a.edge_w[e + 3]; }
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; e < e1; ++e) {
            const int i0 = a.row[e];
            float v0 = a.table[(int64_t)i0 * a.ld_t + c];
            if (a.gate) v0 = a.gate[(int64_t)i0 * a.ld_gate + c] > 0.f ? v0 : 0.f;
            if (a.table_scale) v0 *= a.table_scale[i0];
            if (a.edge_w) v0 *= a.edge_w[e];
            s += v0;
        }
        if (a.out_scale) s *= a.out_scale[r];
        if (a.addend) s += a.addend[r * a.ld_add + c];
        if (a.bias) s += a.bias[c];
        if (a.relu) s = fmaxf(s, 0.f);
        a.out[r * a.ld_out + c] = s;
    }
}

// out = in (.) [gate > 0]   (the ReLU backward of the layer's own epilogue, for the root / d root terms)
__global__ __launch_bounds__(256) void graph_gate_kernel(const float* in, int64_t ld_in, const float* gate, int64_t ld_gate, float* out,
                                                         int64_t rows, int d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * d) return;
    const int64_t r = i / d;
    const int c = (int)(i - r * d);
    const float v = in[r * ld_in + c];
    out[i] = gate[r * ld_gate + c] > 0.f ? v : 0.f;
}

// the work-item plan applies: rows of 4 .. 32 floats, 16-byte aligned, no epilogue beyond row scale / bias / ReLU
bool plan_applies(const GrArgs& a) {
    return a.plan && !a.addend && !a.gate && !a.table_scale && a.d % 4 == 0 && a.d <= 32 && tipk_plan::group_slots_for(a.d) == a.plan->G &&
           a.ld_t % 4 == 0 && a.ld_out % 4 == 0 && !(reinterpret_cast<uintptr_t>(a.table) & 15) && !(reinterpret_cast<uintptr_t>(a.out) & 15);
}

int gr_launch(const GrArgs& a, hipStream_t st) {
    if (a.n_out <= 0) return TIPK_OK;
    if (plan_applies(a))
        return tipk_gather_sum(a.table, a.ld_t, a.plan->n_table, a.plan->row_id, a.edge_w ? a.plan->edge_w : nullptr, a.plan->items,
                               a.plan->n_items, a.out, a.ld_out, nullptr, a.out_scale, a.bias, a.relu, a.d, a.plan->G, (tipk_stream_t)st);
    const int64_t blocks = tipk_ceil_div(a.n_out, (int64_t)4);
    if (blocks > 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    hipLaunchKernelGGL(graph_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    TIPK_RETURN_LAUNCH();
}

tipk_gemm_desc gemm_desc(int64_t m, int64_t n, int64_t k, const float* a, int64_t a_sm, int64_t a_sk, const float* b, int64_t b_sk,
                         int64_t b_sn, float* c, int64_t c_sm) {
    tipk_gemm_desc g;
    memset(&g, 0, sizeof(g));
    g.m = m; g.n = n; g.k = k; g.batch = 1; g.kbatch = 1; g.ksplit = 1;
    g.a = a; g.a_sm = a_sm; g.a_sk = a_sk; g.b = b; g.b_sk = b_sk; g.b_sn = b_sn; g.c = c; g.c_sm = c_sm;
    g.alpha = 1.f;
    return g;
}

inline int64_t align256(int64_t bytes) { return (bytes + 255) / 256 * 256; }

// C [m x n] (rows contiguous) = A . B with a LONG reduction and few output tiles (weight gradients: K = the node count): the k
// range is cut into slabs that fill the chip, added in order afterwards (tipk_gemm_f32 ksplit + tipk_sum_slabs_ex) -- one
// workgroup walking 19 081 terms took 0.8 ms.  reduce_slabs = slabs of the split (0: the plain product), `slabs` = that many
// m x n blocks of the caller's workspace
int64_t reduce_slabs(int64_t m, int64_t n, int64_t k) {
    const int64_t tiles = ((m + 31) / 32) * ((n + 31) / 32);
    const int64_t want = std::min((k + 63) / 64, (512 + tiles - 1) / tiles);
    if (want < 2 || tiles >= 256 || m * n > 65536) return 0;
    return std::min<int64_t>(320, want);
}

int reduce_gemm(tipk_gemm_desc d, float* slabs, tipk_stream_t stream) {
    const int64_t ns = reduce_slabs(d.m, d.n, d.k);
    if (!ns || d.c_sm != d.n || d.batch != 1 || d.kbatch != 1 || d.c_in || d.relu || d.alpha != 1.f) return tipk_gemm_f32(&d, stream);
    float* out = d.c;
    d.ksplit = ns; d.c = slabs; d.c_ss = d.m * d.n;
    const int st = tipk_gemm_f32(&d, stream);
    if (st != TIPK_OK) return st;
    return tipk_sum_slabs_ex(slabs, ns, d.m * d.n, d.m * d.n, 1.f, 0, nullptr, d.n, nullptr, 0, out, stream);
}

void free_gather(GatherDev* p) {
    if (!p) return;
    if (p->row_id) (void)hipFree(p->row_id);
    if (p->edge_w) (void)hipFree(p->edge_w);
    if (p->items) (void)hipFree(p->items);
    delete p;
}

// work items for out[o] = sum_e w[e] table[t_e] (grouped plan, G = 128), uploaded; *out stays null when the graph is too large
// for 32-bit plans or an allocation fails (the wavefront-per-row kernel then runs)
void build_gather_dev(const std::vector<int64_t>& orow, const std::vector<int64_t>& trow, const float* w, int64_t n_out, int64_t n_table,
                      GatherDev** out) {
    *out = nullptr;
    const int64_t e = (int64_t)orow.size();
    if (e >= 0x7fffffffLL || n_out >= (1 << 26) || n_table >= 0x7fffffffLL) return;
    tipk_plan::GatherPlanH gp;
    tipk_plan::build_gather_plan(orow.data(), trow.data(), w, e, n_out, n_table, 0, 128, gp);
    GatherDev* d = new (std::nothrow) GatherDev;
    if (!d) return;
    auto up = [&](void** dev, const void* host, size_t bytes) {
        if (hipMalloc(dev, bytes ? bytes : 4) != hipSuccess) { *dev = nullptr; return false; }
        return !bytes || hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = up((void**)&d->row_id, gp.row_id.data(), gp.row_id.size() * 4) && up((void**)&d->items, gp.items.data(), gp.items.size() * 4);
    if (ok && w) ok = up((void**)&d->edge_w, gp.edge_w.data(), gp.edge_w.size() * 4);
    if (!ok) { free_gather(d); return; }
    d->n_items = gp.n_items; d->n_table = n_table; d->G = 128;
    *out = d;
}

// workspace layout (floats): XB [nb][N][d_out] | Y or dY [R N][d_out] | xroot / g' [N][d_out] | d XB [nb][N][d_out]
struct Ws { float* xb; float* y; float* t; float* dxb; int64_t bytes; };

Ws carve(void* base, int64_t n, int64_t r, int nb, int d_out) {
    Ws w;
    char* p = (char*)base;
    int64_t off = 0;
    w.xb = (float*)(p + off); off += align256((int64_t)nb * n * d_out * 4);
    w.y = (float*)(p + off); off += align256(r * n * d_out * 4);
    w.t = (float*)(p + off); off += align256(n * d_out * 4);
    w.dxb = (float*)(p + off); off += align256((int64_t)nb * n * d_out * 4);
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" int tipk_graph_build(const void* edge_index, const void* edge_type, const void* range_list, int idx_bytes,
                                int64_t n_edges, int64_t n_nodes, int64_t n_rel, const float* in_degree, tipk_graph** out) {
    if (!out) return TIPK_EINVAL;
    *out = nullptr;
    if ((idx_bytes != 4 && idx_bytes != 8) || n_edges < 0 || n_nodes <= 0 || n_rel < 0 || (n_edges > 0 && !edge_index) ||
        (!edge_type && !range_list && n_edges > 0))
        return TIPK_EINVAL;
    if (n_rel * n_nodes >= 0x7fffffffLL || n_edges >= 0x7fffffffLL) return TIPK_EUNSUPPORTED;      // 32-bit row ids / offsets
    // the index tensors may live on the device or on the host (hipMemcpyDefault resolves either)
    auto fetch = [&](const void* p, int64_t count, std::vector<int64_t>& dst) -> int {
        dst.resize((size_t)count);
        if (count == 0) return TIPK_OK;
        if (idx_bytes == 8) return tipk_hip_status(hipMemcpy(dst.data(), p, (size_t)count * 8, hipMemcpyDefault));
        std::vector<int32_t> tmp((size_t)count);
        const int st = tipk_hip_status(hipMemcpy(tmp.data(), p, (size_t)count * 4, hipMemcpyDefault));
        for (int64_t i = 0; i < count; ++i) dst[(size_t)i] = tmp[(size_t)i];
        return st;
    };
    std::vector<int64_t> ei, rel;
    int st = fetch(edge_index, 2 * n_edges, ei);
    if (st != TIPK_OK) return st;
    if (range_list) {                                                          // MyRGCNConv2: relation r owns edges [start, end)
        std::vector<int64_t> rg;
        st = fetch(range_list, 2 * n_rel, rg);
        if (st != TIPK_OK) return st;
        rel.assign((size_t)n_edges, -1);
        for (int64_t r = 0; r < n_rel; ++r) {
            const int64_t s = rg[(size_t)(2 * r)], e = rg[(size_t)(2 * r + 1)];
            if (s < 0 || e < s || e > n_edges) return TIPK_EINVAL;
            for (int64_t i = s; i < e; ++i) rel[(size_t)i] = r;
        }
    } else {
        st = fetch(edge_type, n_edges, rel);
        if (st != TIPK_OK) return st;
    }
    const int64_t* src = ei.data();
    const int64_t* dst = ei.data() + n_edges;
    for (int64_t i = 0; i < n_edges; ++i)
        if (src[i] < 0 || src[i] >= n_nodes || dst[i] < 0 || dst[i] >= n_nodes || rel[(size_t)i] < 0 || rel[(size_t)i] >= n_rel)
            return TIPK_EINVAL;                                                // (the reference raises IndexError)
    // stable counting sorts: by destination, and by (relation, source)
    std::vector<int32_t> fptr((size_t)n_nodes + 1, 0), frow((size_t)n_edges), bptr((size_t)(n_rel * n_nodes) + 1, 0), brow((size_t)n_edges);
    for (int64_t i = 0; i < n_edges; ++i) { ++fptr[(size_t)dst[i] + 1]; ++bptr[(size_t)(rel[(size_t)i] * n_nodes + src[i]) + 1]; }
    for (size_t i = 1; i < fptr.size(); ++i) fptr[i] += fptr[i - 1];
    for (size_t i = 1; i < bptr.size(); ++i) bptr[i] += bptr[i - 1];
    {
        std::vector<int32_t> fpos(fptr.begin(), fptr.end() - 1), bpos(bptr.begin(), bptr.end() - 1);
        for (int64_t i = 0; i < n_edges; ++i) {
            const int64_t yr = rel[(size_t)i] * n_nodes + src[i];
            frow[(size_t)fpos[(size_t)dst[i]]++] = (int32_t)yr;
            brow[(size_t)bpos[(size_t)yr]++] = (int32_t)dst[i];
        }
    }
    std::vector<float> inv((size_t)n_nodes);
    if (in_degree) {                                                           // a shard of the relations: the GLOBAL in-degree
        st = tipk_hip_status(hipMemcpy(inv.data(), in_degree, (size_t)n_nodes * 4, hipMemcpyDefault));
        if (st != TIPK_OK) return st;
        for (auto& v : inv) v = 1.f / (v < 1.f ? 1.f : v);
    } else {
        for (int64_t v = 0; v < n_nodes; ++v) {
            const int deg = fptr[(size_t)v + 1] - fptr[(size_t)v];
            inv[(size_t)v] = 1.f / (float)(deg < 1 ? 1 : deg);                 // torch-scatter 'mean': clamp(count, 1)
        }
    }
    tipk_graph* g = new (std::nothrow) tipk_graph;
    if (!g) return TIPK_EINVAL;
    memset(g, 0, sizeof(*g));
    g->kind = 0; g->n_nodes = n_nodes; g->n_rel = n_rel; g->n_edges = n_edges; g->n_out = n_nodes; g->n_table = n_rel * n_nodes;
    auto up = [&](void** d, const void* h, size_t bytes) -> int {
        hipError_t e = hipMalloc(d, bytes ? bytes : 4);
        if (e != hipSuccess) return tipk_hip_status(e);
        return bytes ? tipk_hip_status(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice)) : TIPK_OK;
    };
    st = up((void**)&g->fwd_ptr, fptr.data(), fptr.size() * 4);
    if (st == TIPK_OK) st = up((void**)&g->fwd_row, frow.data(), frow.size() * 4);
    if (st == TIPK_OK) st = up((void**)&g->bwd_ptr, bptr.data(), bptr.size() * 4);
    if (st == TIPK_OK) st = up((void**)&g->bwd_row, brow.data(), brow.size() * 4);
    if (st == TIPK_OK) st = up((void**)&g->inv_deg, inv.data(), inv.size() * 4);
    if (st != TIPK_OK) { tipk_graph_destroy(g); return st; }
    // work-item plans of the same two gathers (rows of <= 32 floats): few destination rows with thousands of edges each -- BioSNAP:
    // 645 rows of 12 900 -- are 645 wavefronts on the per-row kernel; 1 / deg of the destination is folded into the transposed
    // plan's edge weights.  Not for (relation, node) spaces beyond 4 M rows (config 5: the plan would outweigh the CSR)
    if (n_rel * n_nodes <= (1 << 22) && n_edges > 0) {
        std::vector<int64_t> orow((size_t)n_edges), trow((size_t)n_edges);
        std::vector<float> w((size_t)n_edges);
        for (int64_t i = 0; i < n_edges; ++i) {
            orow[(size_t)i] = dst[i]; trow[(size_t)i] = rel[(size_t)i] * n_nodes + src[i]; w[(size_t)i] = inv[(size_t)dst[i]];
        }
        build_gather_dev(orow, trow, nullptr, n_nodes, n_rel * n_nodes, &g->pf);
        build_gather_dev(trow, orow, w.data(), n_rel * n_nodes, n_nodes, &g->pb);
    }
    // the edge list stays on the host (12 bytes per edge) for tipk_graph_prepare_rgcn, until tipk_graph_release_host
    g->host = new (std::nothrow) HostEdges;
    g->routes = new (std::nothrow) std::vector<PairRoute*>;
    if (!g->host || !g->routes) { tipk_graph_destroy(g); return TIPK_EINVAL; }
    g->host->src.resize((size_t)n_edges); g->host->dst.resize((size_t)n_edges); g->host->rel.resize((size_t)n_edges);
    for (int64_t i = 0; i < n_edges; ++i) {
        g->host->src[(size_t)i] = (int32_t)src[i]; g->host->dst[(size_t)i] = (int32_t)dst[i]; g->host->rel[(size_t)i] = (int32_t)rel[(size_t)i];
    }
    *out = g;
    return TIPK_OK;
}

extern "C" int tipk_graph_destroy(tipk_graph* g) {
    if (!g) return TIPK_OK;
    int st = TIPK_OK;
    void* bufs[7] = {g->fwd_ptr, g->fwd_row, g->bwd_ptr, g->bwd_row, g->inv_deg, g->fwd_w, g->bwd_w};
    for (void* b : bufs)
        if (b) { const int s = tipk_hip_status(hipFree(b)); if (s != TIPK_OK) st = s; }
    if (g->routes) {
        for (PairRoute* pr : *g->routes) {
            for (void* b : pr->owned)
                if (b) { const int s = tipk_hip_status(hipFree(b)); if (s != TIPK_OK) st = s; }
            delete pr;
        }
        delete g->routes;
    }
    free_gather(g->pf);
    free_gather(g->pb);
    delete g->host;
    delete g;
    return st;
}

extern "C" int tipk_graph_release_host(tipk_graph* g) {
    if (!g) return TIPK_EINVAL;
    delete g->host;
    g->host = nullptr;
    return TIPK_OK;
}

extern "C" int tipk_graph_info(const tipk_graph* g, int64_t* n_nodes, int64_t* n_rel, int64_t* n_edges, const float** inv_degree) {
    if (!g) return TIPK_EINVAL;
    if (n_nodes) *n_nodes = g->n_nodes;
    if (n_rel) *n_rel = g->n_rel;
    if (n_edges) *n_edges = g->n_edges;
    if (inv_degree) *inv_degree = g->inv_deg;
    return TIPK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The pair-form route behind the handle.

namespace {

template <class T> int upload(PairRoute* pr, T** dev, const void* host, size_t count) {
    void* d = nullptr;
    const size_t bytes = count * sizeof(T);
    hipError_t e = hipMalloc(&d, bytes ? bytes : 4);
    if (e != hipSuccess) return tipk_hip_status(e);
    pr->owned.push_back(d);
    *dev = (T*)d;
    return bytes && host ? tipk_hip_status(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice)) : TIPK_OK;
}

int upload_stream(PairRoute* pr, StreamDev& sd, const tipk_plan::StreamPlanH& sp) {
    int st = upload(pr, &sd.wave_ptr, sp.wave_ptr.data(), sp.wave_ptr.size());
    if (st == TIPK_OK) st = upload(pr, &sd.cells, sp.cells.data(), sp.cells.size());
    if (st == TIPK_OK) st = upload(pr, &sd.ids, sp.ids.data(), sp.ids.size());
    if (st == TIPK_OK) st = upload(pr, &sd.zero_ptr, sp.zero_ptr.data(), sp.zero_ptr.size());
    if (st == TIPK_OK) st = upload(pr, &sd.zero_rows, sp.zero_rows.data(), sp.zero_rows.size());
    sd.n_wg = sp.n_wg; sd.n_bands = sp.n_bands; sd.idx_unit = sp.idx_unit;
    return st;
}

// the route prepared for this layer shape (nullptr: the generic route)
const PairRoute* route_for(const tipk_graph* g, int n_bases, int d_out) {
    if (!g->routes || !tipk_pair_product_supported(n_bases, d_out)) return nullptr;
    for (const PairRoute* pr : *g->routes)
        if (pr->n_bases == n_bases) return pr;
    return nullptr;
}

inline int64_t pad_group(int64_t n) { return (n + PAIR_KGROUP - 1) / PAIR_KGROUP * PAIR_KGROUP; }

// workspace of the pair route (floats): cells [N_pad][N][nb] | XB [N_pad][nb][32] | X root / g' [N][d_out] |
// product slabs [N_pad / 8][N][d_out] | d XB [nb][N][d_out] | d att slabs [parts][R][nb]
struct PairWs { float* cells; float* xb; float* t; float* slabs; float* dxb; float* att_slabs; int64_t bytes; };

PairWs carve_pair(void* base, int64_t n, int64_t r, int nb, int d_out, int64_t n_parts) {
    PairWs w;
    char* p = (char*)base;
    int64_t off = 0;
    const int64_t n_pad = pad_group(n);
    w.cells = (float*)(p + off); off += align256(n_pad * n * nb * 4);
    w.xb = (float*)(p + off); off += align256(n_pad * nb * 32 * 4);
    w.t = (float*)(p + off); off += align256(n * d_out * 4);
    w.slabs = (float*)(p + off); off += align256(n_pad / PAIR_KGROUP * n * d_out * 4);
    w.dxb = (float*)(p + off); off += align256((int64_t)nb * n * d_out * 4);
    w.att_slabs = (float*)(p + off); off += align256(std::max<int64_t>(n_parts, 1) * r * nb * 4);
    w.bytes = off;
    return w;
}

// XB into the node-major buffer of the pair product (rows padded to 32 columns, everything the product does not write zero)
// and the pair cells C[u][v][:] = sum of att[r, :] over the relations linking u -> v
// (root != nullptr: X root into w.t in the same grouped launch)
int pair_operands(const tipk_graph* g, const PairRoute* pr, const PairWs& w, const float* x, int64_t ld_x, int d_in, const float* basis,
                  const float* att, const float* root, int d_out, tipk_stream_t stream) {
    const int64_t n = g->n_nodes, r = g->n_rel;
    const int nb = pr->n_bases;
    int st = tipk_hip_status(hipMemsetAsync(w.xb, 0, (size_t)(pad_group(n) * nb * 32 * 4), (hipStream_t)stream));
    if (st != TIPK_OK) return st;
    tipk_gemm_desc d[2];
    d[0] = gemm_desc(n, d_out, d_in, x, ld_x, 1, basis, d_out, 1, w.xb, (int64_t)nb * 32);
    d[0].batch = nb; d[0].a_sz = 0; d[0].b_sz = (int64_t)d_in * d_out; d[0].c_sz = 32;
    d[1] = gemm_desc(n, d_out, d_in, x, ld_x, 1, root, d_out, 1, w.t, d_out);
    if ((st = tipk_gemm_f32_group(d, root ? 2 : 1, stream)) != TIPK_OK) return st;
    return tipk_stream_gather(att, nb, r, nb, pr->fwd.n_wg, pr->fwd.wave_ptr, pr->fwd.cells, pr->fwd.ids, pr->fwd.idx_unit, nullptr,
                              pr->fwd.zero_rows, nullptr, w.cells, nb, 1, 4, nullptr, nullptr, 0, stream);
}

// the dense products of d XB [nb][N][d_out] and g' that every route ends with, plus (pair form) the ordered sum of the d att
// slabs: ONE launch when the reductions fit the waves of a workgroup per output tile (tipk_gemm_wg_group: BioSNAP does),
// else product by product
int dense_grads(const float* x, int64_t ld_x, int d_in, const float* basis, const float* root, int n_bases, int d_out, int64_t n,
                const float* gp, int64_t ld_gp, const float* dxb, float* g_x, int64_t ld_gx, float* g_basis, float* g_root,
                const tipk_slab_sum_desc* att_sum, tipk_stream_t stream) {
    int st;
    // d basis_b = X^T d XB_b (batched, A shared),  d root = X^T g',  dX = sum_b d XB_b basis_b^T + g' root^T
    tipk_wg_gemm_desc w[3];
    memset(w, 0, sizeof(w));
    w[0].p = gemm_desc(d_in, d_out, n, x, 1, ld_x, dxb, d_out, 1, g_basis, d_out);
    w[0].p.batch = n_bases; w[0].p.a_sz = 0; w[0].p.b_sz = n * d_out; w[0].p.c_sz = (int64_t)d_in * d_out;
    w[1].p = gemm_desc(d_in, d_out, n, x, 1, ld_x, gp, ld_gp, 1, g_root, d_out);
    w[2].p = gemm_desc(n, d_in, d_out, dxb, d_out, 1, basis, 1, d_out, g_x, ld_gx);
    w[2].p.kbatch = n_bases; w[2].p.a_sq = n * d_out; w[2].p.b_sq = (int64_t)d_in * d_out;
    w[2].a2 = gp; w[2].a2_sm = ld_gp; w[2].a2_sk = 1; w[2].b2 = root; w[2].b2_sk = 1; w[2].b2_sn = d_out; w[2].k2 = d_out;
    if (tipk_gemm_wg_group_supported(&w[0]) && tipk_gemm_wg_group_supported(&w[1]) && tipk_gemm_wg_group_supported(&w[2]))
        return tipk_gemm_wg_group(w, 3, att_sum, att_sum ? 1 : 0, stream);
    if (att_sum && (st = tipk_sum_slabs_group(att_sum, 1, stream)) != TIPK_OK) return st;
    if ((st = tipk_gemm_f32(&w[0].p, stream)) != TIPK_OK) return st;
    if ((st = tipk_gemm_f32(&w[1].p, stream)) != TIPK_OK) return st;
    // dX = g' root^T, then += sum_b d XB_b basis_b^T (the batch reduced inside the product, on top of the first term)
    tipk_gemm_desc d = gemm_desc(n, d_in, d_out, gp, ld_gp, 1, root, 1, d_out, g_x, ld_gx);
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    d = w[2].p;
    d.c_in = g_x; d.cin_sm = ld_gx;
    return tipk_gemm_f32(&d, stream);
}

int gate_rows(const float* in, int64_t ld_in, const float* gate, int64_t ld_gate, float* out, int64_t rows, int d, hipStream_t hs);

}  // namespace

extern "C" int tipk_graph_prepare_rgcn(tipk_graph* g, int n_bases, int d_out) {
    if (!g || g->kind != 0 || n_bases <= 0 || d_out <= 0) return TIPK_EINVAL;
    const int64_t n = g->n_nodes, r = g->n_rel, e = g->n_edges;
    if (!g->routes || n > 1024 || r <= 0 || e <= 0 || n * n >= (1 << 24) || !tipk_pair_product_supported(n_bases, d_out) || d_out > 32)
        return TIPK_EUNSUPPORTED;
    const int split = tipk_stream_gather_supported(r, n_bases, 4);
    if (!split || (n_bases / split) % 4) return TIPK_EUNSUPPORTED;
    const int lanes = (n_bases / split) / 4, piece = tipk_stream_gather_piece();
    PairRoute* pr = nullptr;
    for (PairRoute* q : *g->routes)
        if (q->n_bases == n_bases) pr = q;
    const bool want_bwd = tipk_rgcn_pair_grads_supported(n_bases, d_out) != 0;
    if (pr && (pr->has_bwd || !want_bwd)) return TIPK_OK;
    if (!g->host) return TIPK_EINVAL;                                          // tipk_graph_release_host was called
    int dev = 0, n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n_cu = prop.multiProcessorCount;
    std::vector<int64_t> src((size_t)e), dst((size_t)e), rel((size_t)e);
    for (int64_t i = 0; i < e; ++i) { src[(size_t)i] = g->host->src[(size_t)i]; dst[(size_t)i] = g->host->dst[(size_t)i]; rel[(size_t)i] = g->host->rel[(size_t)i]; }
    if (!g->symmetric_known) {
        g->symmetric = tipk_plan::relations_symmetric(src.data(), dst.data(), rel.data(), e, n);
        g->symmetric_known = 1;
    }
    int st = TIPK_OK;
    if (!pr) {
        pr = new (std::nothrow) PairRoute;
        if (!pr) return TIPK_EINVAL;
        g->routes->push_back(pr);                                              // (owned by the handle from here on, whatever happens)
        pr->n_bases = n_bases; pr->lanes = lanes; pr->symmetric = g->symmetric != 0;
        // every relation symmetric => cell (u, v) == cell (v, u): the cells with u <= v only, from half the edges
        std::vector<int64_t> out_row, tab_row;
        out_row.reserve((size_t)e); tab_row.reserve((size_t)e);
        for (int64_t i = 0; i < e; ++i)
            if (!pr->symmetric || src[(size_t)i] <= dst[(size_t)i]) { out_row.push_back(src[(size_t)i] * n + dst[(size_t)i]); tab_row.push_back(rel[(size_t)i]); }
        tipk_plan::StreamPlanH sp;
        tipk_plan::build_stream_plan_rows(out_row.data(), tab_row.data(), (int64_t)out_row.size(), n * n, r, n_cu, lanes, piece, 0, 0, sp);
        std::vector<uint32_t> links;
        tipk_plan::pair_link_words(src.data(), dst.data(), e, n, links);
        st = upload_stream(pr, pr->fwd, sp);
        if (st == TIPK_OK) st = upload(pr, &pr->links, links.data(), links.size());
        if (st == TIPK_OK) st = upload(pr, &pr->zeros, (const void*)nullptr, (size_t)64);
        if (st == TIPK_OK) st = tipk_hip_status(hipMemset(pr->zeros, 0, 64 * 4));
        if (st != TIPK_OK) { pr->n_bases = -1; return st; }                    // (a route no shape matches; freed with the handle)
    }
    if (want_bwd && !pr->has_bwd) {
        std::vector<float> scale((size_t)n);
        st = tipk_hip_status(hipMemcpy(scale.data(), g->inv_deg, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (st != TIPK_OK) return st;
        tipk_plan::PairBwdH pb;
        if (!tipk_plan::build_pair_bwd_plan(src.data(), dst.data(), rel.data(), e, n, r, scale.data(), pr->symmetric, n_cu, lanes, piece, pb))
            return TIPK_OK;                                                    // forward in pair form, backward on the generic route
        st = upload(pr, &pr->slots, pb.slots.data(), pb.slots.size());
        if (st == TIPK_OK) st = upload(pr, &pr->node_desc, pb.node_desc.data(), pb.node_desc.size());
        if (st == TIPK_OK) st = upload(pr, &pr->tile_node, pb.tile_node.data(), pb.tile_node.size());
        if (st == TIPK_OK) st = upload(pr, &pr->part_first, pb.part_first.data(), pb.part_first.size());
        if (st == TIPK_OK) st = upload(pr, &pr->wg_part, pb.wg_part.data(), pb.wg_part.size());
        if (st == TIPK_OK) st = upload_stream(pr, pr->gather, pb.gather);
        const size_t pg_count = (size_t)(2 * pb.n_alloc + 1) * (size_t)n_bases;
        if (st == TIPK_OK) st = upload(pr, &pr->pg, (const void*)nullptr, pg_count);
        if (st == TIPK_OK) st = tipk_hip_status(hipMemset(pr->pg, 0, pg_count * 4));
        if (st != TIPK_OK) return st;
        pr->n_slots = pb.n_slots; pr->n_parts = pb.n_parts; pr->part_len = pb.part_len; pr->n_alloc = pb.n_alloc;
        pr->has_bwd = true;
    }
    return TIPK_OK;
}

extern "C" int tipk_graph_rgcn_route(const tipk_graph* g, int n_bases, int d_out) {
    if (!g || g->kind != 0) return TIPK_EINVAL;
    const PairRoute* pr = route_for(g, n_bases, d_out);
    if (!pr) return 0;
    return 1 + (pr->has_bwd && tipk_rgcn_pair_grads_supported(n_bases, d_out) ? 1 : 0);
}

extern "C" int64_t tipk_rgcn_workspace_bytes(const tipk_graph* g, int d_in, int d_out, int n_bases) {
    if (!g || g->kind != 0 || d_in <= 0 || d_out <= 0 || n_bases <= 0) return -1;
    const int64_t generic = carve(nullptr, g->n_nodes, g->n_rel, n_bases, d_out).bytes;
    const PairRoute* pr = route_for(g, n_bases, d_out);
    if (!pr) return generic;
    // (a route whose backward pass runs on the generic passes needs their workspace as well)
    return std::max(generic, carve_pair(nullptr, g->n_nodes, g->n_rel, n_bases, d_out, pr->n_parts).bytes);
}

extern "C" int tipk_rgcn_fwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                             const float* root, int n_bases, int d_out, int relu, float* out, int64_t ld_out, void* workspace,
                             int64_t workspace_bytes, tipk_stream_t stream) {
    if (!g || g->kind != 0 || !x || !basis || !root || !out || !workspace || d_in <= 0 || d_out <= 0 || n_bases <= 0 || ld_x < d_in || ld_out < d_out ||
        (g->n_rel > 0 && !att))
        return TIPK_EINVAL;
    const int64_t n = g->n_nodes, r = g->n_rel;
    if (workspace_bytes < tipk_rgcn_workspace_bytes(g, d_in, d_out, n_bases) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    int st;
    const PairRoute* pr = route_for(g, n_bases, d_out);
    if (pr && ld_out == d_out) {
        // PAIR FORM (tip_amd/ops.py `_RGCN.forward`, same launches): sum_r A_r X W_r = sum over the linked pairs (u -> v) of
        // C[u, v, :] . XB[u],  C[u, v, :] = sum of att[r, :] over the relations linking u -> v
        const PairWs pw = carve_pair(workspace, n, r, n_bases, d_out, pr->n_parts);
        if ((st = pair_operands(g, pr, pw, x, ld_x, d_in, basis, att, root, d_out, stream)) != TIPK_OK) return st;
        const int64_t n_pad = pad_group(n);
        if ((st = tipk_pair_product(pw.cells, pw.xb, n_pad, n, n_bases, d_out, PAIR_KGROUP, pr->symmetric, pr->links, pr->zeros, nullptr,
                                    pw.slabs, stream)) != TIPK_OK)
            return st;
        return tipk_sum_slabs_ex(pw.slabs, n_pad / PAIR_KGROUP, n * d_out, n * d_out, 1.f, 0, g->inv_deg, d_out, pw.t, relu, out, stream);
    }
    const Ws w = carve(workspace, n, r, n_bases, d_out);
    // XB_b = X basis_b (one batched product, A shared), X root
    tipk_gemm_desc d = gemm_desc(n, d_out, d_in, x, ld_x, 1, basis, d_out, 1, w.xb, d_out);
    d.batch = n_bases; d.a_sz = 0; d.b_sz = (int64_t)d_in * d_out; d.c_sz = n * d_out;
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    d = gemm_desc(n, d_out, d_in, x, ld_x, 1, root, d_out, 1, w.t, d_out);
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    if (r > 0) {
        // Y = att . XB  [R, N d_out]
        d = gemm_desc(r, n * d_out, n_bases, att, n_bases, 1, w.xb, n * d_out, 1, w.y, n * d_out);
        if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    }
    GrArgs a;
    memset(&a, 0, sizeof(a));
    a.table = w.y; a.ld_t = d_out; a.ptr = g->fwd_ptr; a.row = g->fwd_row; a.n_out = n;
    a.out_scale = g->inv_deg; a.d = d_out; a.plan = g->pf;
    a.out = w.dxb; a.ld_out = d_out;                                            // (d XB's place is free in the forward pass)
    if (ld_out == d_out && plan_applies(a)) {
        // work items: the scaled aggregate first, then + X root and the ReLU in the ordered-sum kernel (one "slab")
        if ((st = gr_launch(a, (hipStream_t)stream)) != TIPK_OK) return st;
        return tipk_sum_slabs_ex(w.dxb, 1, n * d_out, n * d_out, 1.f, 0, nullptr, d_out, w.t, relu, out, stream);
    }
    a.plan = nullptr;
    a.addend = w.t; a.ld_add = d_out; a.relu = relu; a.out = out; a.ld_out = ld_out;
    return gr_launch(a, (hipStream_t)stream);
}

extern "C" int tipk_rgcn_bwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                             const float* root, int n_bases, int d_out, const float* grad_out, int64_t ld_g, const float* out_relu,
                             int64_t ld_relu, float* g_x, int64_t ld_gx, float* g_basis, float* g_att, float* g_root, void* workspace,
                             int64_t workspace_bytes, tipk_stream_t stream) {
    return tipk_rgcn_bwd_ex(g, x, ld_x, d_in, basis, att, root, n_bases, d_out, grad_out, ld_g, out_relu, ld_relu, g_x, ld_gx, g_basis, g_att,
                            g_root, workspace, workspace_bytes, 0, stream);
}

extern "C" int tipk_rgcn_bwd_ex(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                                const float* root, int n_bases, int d_out, const float* grad_out, int64_t ld_g, const float* out_relu,
                                int64_t ld_relu, float* g_x, int64_t ld_gx, float* g_basis, float* g_att, float* g_root, void* workspace,
                                int64_t workspace_bytes, int flags, tipk_stream_t stream) {
    if (!g || g->kind != 0 || !x || !basis || !root || !grad_out || !g_x || !g_basis || !g_root || !workspace || d_in <= 0 || d_out <= 0 || n_bases <= 0 ||
        ld_x < d_in || ld_g < d_out || ld_gx < d_in || (g->n_rel > 0 && (!att || !g_att)) || (out_relu && ld_relu < d_out) ||
        (flags & ~TIPK_RGCN_WORKSPACE_FROM_FWD))
        return TIPK_EINVAL;
    const int64_t n = g->n_nodes, r = g->n_rel;
    if (workspace_bytes < tipk_rgcn_workspace_bytes(g, d_in, d_out, n_bases) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    int st;
    const PairRoute* pr = route_for(g, n_bases, d_out);
    if (pr && pr->has_bwd && tipk_rgcn_pair_grads_supported(n_bases, d_out)) {
        // PAIR-FORM BACKWARD (tip_amd/ops.py `pair_backward`): per linked pair one gradient row pg[(u, v), :] = XB[u] . g'[v] / deg(v)
        // and d XB[u] += C[u, v, :] (x) g'[v] / deg(v) in one launch; d att[r, :] = sum of the pair rows r links (partitioned
        // wave-stream gather -> slabs -> ordered sum); then the dense products of d XB as on the generic route
        const PairWs pw = carve_pair(workspace, n, r, n_bases, d_out, pr->n_parts);
        const float* gp = grad_out;
        int64_t ld_gp = ld_g;
        if (out_relu) {
            if ((st = gate_rows(grad_out, ld_g, out_relu, ld_relu, pw.t, n, d_out, hs)) != TIPK_OK) return st;
            gp = pw.t; ld_gp = d_out;
        }
        if (!(flags & TIPK_RGCN_WORKSPACE_FROM_FWD) && (st = pair_operands(g, pr, pw, x, ld_x, d_in, basis, att, nullptr, d_out, stream)) != TIPK_OK)
            return st;
        if ((st = tipk_rgcn_pair_grads(pw.cells, pad_group(n) * n, pw.xb, gp, ld_gp, n, n_bases, d_out, pr->node_desc, pr->slots, pr->tile_node,
                                       pr->n_slots, pw.dxb, n * d_out, d_out, pr->pg, 2 * pr->n_alloc + 1, stream)) != TIPK_OK)
            return st;
        if ((st = tipk_stream_gather_parts(pr->pg, n_bases, n_bases, pr->n_alloc, pr->part_first, pr->part_len, pr->wg_part, pr->gather.n_wg,
                                           pr->gather.wave_ptr, pr->gather.cells, pr->gather.ids, pr->gather.idx_unit, pr->gather.zero_ptr,
                                           pr->gather.zero_rows, pw.att_slabs, n_bases, stream)) != TIPK_OK)
            return st;
        tipk_slab_sum_desc sum;
        memset(&sum, 0, sizeof(sum));
        sum.in = pw.att_slabs; sum.n_slabs = pr->n_parts; sum.slab_stride = r * n_bases; sum.count = r * n_bases; sum.alpha = 1.f;
        sum.cols = n_bases; sum.out = g_att;
        return dense_grads(x, ld_x, d_in, basis, root, n_bases, d_out, n, gp, ld_gp, pw.dxb, g_x, ld_gx, g_basis, g_root, &sum, stream);
    }
    const Ws w = carve(workspace, n, r, n_bases, d_out);
    // g' = g (.) [out > 0] (only when the layer applied the ReLU itself)
    const float* gp = grad_out;
    int64_t ld_gp = ld_g;
    if (out_relu) {
        const int64_t tot = n * d_out;
        hipLaunchKernelGGL(graph_gate_kernel, dim3((unsigned)tipk_ceil_div(tot, (int64_t)256)), dim3(256), 0, hs, grad_out, ld_g, out_relu,
                           ld_relu, w.t, n, d_out);
        if ((st = tipk_hip_status(hipGetLastError())) != TIPK_OK) return st;
        gp = w.t; ld_gp = d_out;
    }
    // XB again (nothing of the forward pass is kept besides what the caller holds: X and the parameters)
    tipk_gemm_desc d = gemm_desc(n, d_out, d_in, x, ld_x, 1, basis, d_out, 1, w.xb, d_out);
    d.batch = n_bases; d.a_sz = 0; d.b_sz = (int64_t)d_in * d_out; d.c_sz = n * d_out;
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    if (r > 0) {
        // dY[(r, u)] = sum over the edges (r, u) -> v of g'[v] / deg[v]
        GrArgs a;
        memset(&a, 0, sizeof(a));
        a.table = gp; a.ld_t = ld_gp; a.ptr = g->bwd_ptr; a.row = g->bwd_row; a.n_out = r * n;
        a.d = d_out; a.out = w.y; a.ld_out = d_out; a.plan = g->pb;
        if (plan_applies(a)) a.edge_w = g->inv_deg;                            // (non-null = "weighted": the plan carries 1 / deg per edge)
        else { a.plan = nullptr; a.table_scale = g->inv_deg; }
        if ((st = gr_launch(a, hs)) != TIPK_OK) return st;
        // d att = dY . XB^T  [R, bases],  d XB = att^T . dY  [bases, N d_out]
        d = gemm_desc(r, n_bases, n * d_out, w.y, n * d_out, 1, w.xb, 1, n * d_out, g_att, n_bases);
        if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
        d = gemm_desc(n_bases, n * d_out, r, att, 1, n_bases, w.y, n * d_out, 1, w.dxb, n * d_out);
        if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    } else {
        if ((st = tipk_hip_status(hipMemsetAsync(w.dxb, 0, (size_t)n_bases * n * d_out * 4, hs))) != TIPK_OK) return st;
    }
    return dense_grads(x, ld_x, d_in, basis, root, n_bases, d_out, n, gp, ld_gp, w.dxb, g_x, ld_gx, g_basis, g_root, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// GCNConv and MyHierarchyConv behind the same kind of handle (include/tipk.h section 10b).

namespace {

// weighted pair of CSRs from (out row, table row, weight) triples, stable in list order; everything uploaded into `g`
int build_weighted(tipk_graph* g, const std::vector<int64_t>& orow, const std::vector<int64_t>& trow, const std::vector<float>& w,
                   int64_t n_out, int64_t n_table) {
    const size_t e = orow.size();
    std::vector<int32_t> fptr((size_t)n_out + 1, 0), frow(e), bptr((size_t)n_table + 1, 0), brow(e);
    std::vector<float> fw(e), bw(e);
    for (size_t i = 0; i < e; ++i) { ++fptr[(size_t)orow[i] + 1]; ++bptr[(size_t)trow[i] + 1]; }
    for (size_t i = 1; i < fptr.size(); ++i) fptr[i] += fptr[i - 1];
    for (size_t i = 1; i < bptr.size(); ++i) bptr[i] += bptr[i - 1];
    std::vector<int32_t> fpos(fptr.begin(), fptr.end() - 1), bpos(bptr.begin(), bptr.end() - 1);
    for (size_t i = 0; i < e; ++i) {
        const int32_t f = fpos[(size_t)orow[i]]++, b = bpos[(size_t)trow[i]]++;
        frow[(size_t)f] = (int32_t)trow[i]; fw[(size_t)f] = w[i];
        brow[(size_t)b] = (int32_t)orow[i]; bw[(size_t)b] = w[i];
    }
    auto up = [&](void** d, const void* h, size_t bytes) -> int {
        hipError_t er = hipMalloc(d, bytes ? bytes : 4);
        if (er != hipSuccess) return tipk_hip_status(er);
        return bytes ? tipk_hip_status(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice)) : TIPK_OK;
    };
    int st = up((void**)&g->fwd_ptr, fptr.data(), fptr.size() * 4);
    if (st == TIPK_OK) st = up((void**)&g->fwd_row, frow.data(), e * 4);
    if (st == TIPK_OK) st = up((void**)&g->bwd_ptr, bptr.data(), bptr.size() * 4);
    if (st == TIPK_OK) st = up((void**)&g->bwd_row, brow.data(), e * 4);
    if (st == TIPK_OK) st = up((void**)&g->fwd_w, fw.data(), e * 4);
    if (st == TIPK_OK) st = up((void**)&g->bwd_w, bw.data(), e * 4);
    g->n_out = n_out; g->n_table = n_table; g->n_edges = (int64_t)e;
    if (st == TIPK_OK && e > 0) {
        build_gather_dev(orow, trow, w.data(), n_out, n_table, &g->pf);
        build_gather_dev(trow, orow, w.data(), n_table, n_out, &g->pb);
    }
    return st;
}

int fetch_index(const void* p, int idx_bytes, int64_t count, std::vector<int64_t>& dst) {
    dst.resize((size_t)count);
    if (count == 0) return TIPK_OK;
    if (idx_bytes == 8) return tipk_hip_status(hipMemcpy(dst.data(), p, (size_t)count * 8, hipMemcpyDefault));
    std::vector<int32_t> tmp((size_t)count);
    const int st = tipk_hip_status(hipMemcpy(tmp.data(), p, (size_t)count * 4, hipMemcpyDefault));
    for (int64_t i = 0; i < count; ++i) dst[(size_t)i] = tmp[(size_t)i];
    return st;
}

int gate_rows(const float* in, int64_t ld_in, const float* gate, int64_t ld_gate, float* out, int64_t rows, int d, hipStream_t hs) {
    const int64_t tot = rows * d;
    if (tot <= 0) return TIPK_OK;
    hipLaunchKernelGGL(graph_gate_kernel, dim3((unsigned)tipk_ceil_div(tot, (int64_t)256)), dim3(256), 0, hs, in, ld_in, gate, ld_gate, out,
                       rows, d);
    return tipk_hip_status(hipGetLastError());
}

}  // namespace

// GCNConv (PyG 2.0.1 semantics, as PPEncoder uses it: src/layers.py:386-394): A_hat = D^-1/2 (A + I) D^-1/2 with existing self
// loops replaced by exactly one unit loop per node, D = in-degree including the loop, flow source -> target.
extern "C" int tipk_gcn_graph_build(const void* edge_index, int idx_bytes, int64_t n_edges, int64_t n_nodes, tipk_graph** out) {
    if (!out) return TIPK_EINVAL;
    *out = nullptr;
    if ((idx_bytes != 4 && idx_bytes != 8) || n_edges < 0 || n_nodes <= 0 || (n_edges > 0 && !edge_index)) return TIPK_EINVAL;
    if (n_edges + n_nodes >= 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    std::vector<int64_t> ei;
    int st = fetch_index(edge_index, idx_bytes, 2 * n_edges, ei);
    if (st != TIPK_OK) return st;
    std::vector<int64_t> src, dst;
    src.reserve((size_t)(n_edges + n_nodes)); dst.reserve((size_t)(n_edges + n_nodes));
    for (int64_t i = 0; i < n_edges; ++i) {
        const int64_t s = ei[(size_t)i], d = ei[(size_t)(n_edges + i)];
        if (s < 0 || s >= n_nodes || d < 0 || d >= n_nodes) return TIPK_EINVAL;
        if (s != d) { src.push_back(s); dst.push_back(d); }                   // add_remaining_self_loops: one loop per node below
    }
    for (int64_t v = 0; v < n_nodes; ++v) { src.push_back(v); dst.push_back(v); }
    std::vector<double> deg((size_t)n_nodes, 0.0);
    for (size_t i = 0; i < dst.size(); ++i) deg[(size_t)dst[i]] += 1.0;
    std::vector<float> dis((size_t)n_nodes), w(src.size());
    for (int64_t v = 0; v < n_nodes; ++v) dis[(size_t)v] = deg[(size_t)v] > 0 ? 1.0f / sqrtf((float)deg[(size_t)v]) : 0.f;
    for (size_t i = 0; i < src.size(); ++i) w[i] = dis[(size_t)src[i]] * dis[(size_t)dst[i]];
    tipk_graph* g = new (std::nothrow) tipk_graph;
    if (!g) return TIPK_EINVAL;
    memset(g, 0, sizeof(*g));
    g->kind = 1; g->n_nodes = n_nodes;
    st = build_weighted(g, dst, src, w, n_nodes, n_nodes);                    // out[target] += w * table[source]
    if (st != TIPK_OK) { tipk_graph_destroy(g); return st; }
    *out = g;
    return TIPK_OK;
}

extern "C" int64_t tipk_gcn_workspace_bytes(const tipk_graph* g, int d_in, int d_out) {
    if (!g || g->kind != 1 || d_out <= 0) return -1;
    // [g' | d lin | column-sum scratch | slabs of the split d W product (dense features of modest width)]
    const int64_t slabs = d_in > 0 ? reduce_slabs(d_out, d_in, g->n_nodes) : 0;
    return 2 * align256(g->n_nodes * (int64_t)d_out * 4) + align256(256 * (int64_t)d_out * 4) + align256(slabs * d_out * (int64_t)d_in * 4);
}

// out = relu?( A_hat (x W^T) + bias );  x = NULL: identity features (lin(I) = W^T: d_in = n_nodes); weight element (o, i) at
// weight[o * w_so + i * w_si]
extern "C" int tipk_gcn_fwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* weight, int64_t w_so, int64_t w_si,
                            const float* bias, int d_out, int relu, float* out, int64_t ld_out, void* workspace,
                            int64_t workspace_bytes, tipk_stream_t stream) {
    if (!g || g->kind != 1 || !weight || !out || !workspace || d_out <= 0 || ld_out < d_out || (x && (d_in <= 0 || ld_x < d_in)))
        return TIPK_EINVAL;
    const int64_t n = g->n_nodes;
    if (workspace_bytes < tipk_gcn_workspace_bytes(g, d_in, d_out) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    float* xl = (float*)workspace;
    GrArgs a;
    memset(&a, 0, sizeof(a));
    if (x) {
        tipk_gemm_desc d = gemm_desc(n, d_out, d_in, x, ld_x, 1, weight, w_si, w_so, xl, d_out);          // x W^T
        const int st = tipk_gemm_f32(&d, stream);
        if (st != TIPK_OK) return st;
        a.table = xl; a.ld_t = d_out;
    } else {
        // identity features: lin(I) = W^T read in place -- table row i, column o = weight[o * w_so + i * w_si]; the gather wants
        // unit-stride columns (this package stores lin.weight as [in, out] memory behind its [out, in] shape: w_so = 1)
        if (w_so != 1) return TIPK_EUNSUPPORTED;
        a.table = weight; a.ld_t = w_si;
    }
    a.ptr = g->fwd_ptr; a.row = g->fwd_row; a.n_out = n; a.edge_w = g->fwd_w; a.bias = bias; a.relu = relu; a.d = d_out; a.plan = g->pf;
    a.out = out; a.ld_out = ld_out;
    return gr_launch(a, (hipStream_t)stream);
}

extern "C" int tipk_gcn_bwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* weight, int64_t w_so, int64_t w_si,
                            int d_out, const float* grad_out, int64_t ld_g, const float* out_relu, int64_t ld_relu, float* g_x,
                            int64_t ld_gx, float* g_weight, int64_t gw_so, int64_t gw_si, float* g_bias, void* workspace,
                            int64_t workspace_bytes, tipk_stream_t stream) {
    if (!g || g->kind != 1 || !weight || !grad_out || !g_weight || !workspace || d_out <= 0 || ld_g < d_out ||
        (x && (d_in <= 0 || ld_x < d_in)) || (g_x && (!x || ld_gx < d_in)) || (out_relu && ld_relu < d_out))
        return TIPK_EINVAL;
    const int64_t n = g->n_nodes;
    if (workspace_bytes < tipk_gcn_workspace_bytes(g, d_in, d_out) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    char* wp = (char*)workspace;
    float* gp_buf = (float*)wp;
    float* gxl = (float*)(wp + align256(n * (int64_t)d_out * 4));
    float* scratch = (float*)(wp + 2 * align256(n * (int64_t)d_out * 4));
    int st;
    const float* gp = grad_out;
    int64_t ld_gp = ld_g;
    if (out_relu) {
        if ((st = gate_rows(grad_out, ld_g, out_relu, ld_relu, gp_buf, n, d_out, hs)) != TIPK_OK) return st;
        gp = gp_buf; ld_gp = d_out;
    }
    if (g_bias && (st = tipk_col_sum(gp, ld_gp, n, d_out, scratch, g_bias, stream)) != TIPK_OK) return st;
    // d (x W^T) = A_hat^T g'
    float* gxl_out = gxl;
    int64_t ld_gxl = d_out;
    if (!x && gw_so == 1) { gxl_out = g_weight; ld_gxl = gw_si; }              // identity features: d W = (d lin)^T, written in place
    GrArgs a;
    memset(&a, 0, sizeof(a));
    a.table = gp; a.ld_t = ld_gp; a.ptr = g->bwd_ptr; a.row = g->bwd_row; a.n_out = n; a.edge_w = g->bwd_w; a.d = d_out; a.plan = g->pb;
    a.out = gxl_out; a.ld_out = ld_gxl;
    if ((st = gr_launch(a, hs)) != TIPK_OK) return st;
    if (!x) return gw_so == 1 ? TIPK_OK : TIPK_EUNSUPPORTED;
    // d W (o, i) = sum_v gxl[v, o] x[v, i];  d x = gxl W
    tipk_gemm_desc d = gemm_desc(d_out, d_in, n, gxl, 1, d_out, x, ld_x, 1, g_weight, gw_so);
    if (gw_si != 1) return TIPK_EUNSUPPORTED;
    if ((st = reduce_gemm(d, scratch + align256(256 * (int64_t)d_out * 4) / 4, stream)) != TIPK_OK) return st;
    if (g_x) {
        d = gemm_desc(n, d_in, d_out, gxl, d_out, 1, weight, w_so, w_si, g_x, ld_gx);
        if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    }
    return TIPK_OK;
}

// MyHierarchyConv (src/layers.py:196-247): mean over the incoming edges of the rows [n_source, n_all) of the concatenated
// node space (edges that end below n_source are ignored, as `aggr_out[self.unique_source_num:]` drops them), then . weight
extern "C" int tipk_hier_graph_build(const void* edge_index, int idx_bytes, int64_t n_edges, int64_t n_all, int64_t n_source,
                                     tipk_graph** out) {
    if (!out) return TIPK_EINVAL;
    *out = nullptr;
    if ((idx_bytes != 4 && idx_bytes != 8) || n_edges < 0 || n_all <= 0 || n_source < 0 || n_source >= n_all || (n_edges > 0 && !edge_index))
        return TIPK_EINVAL;
    if (n_edges >= 0x7fffffffLL) return TIPK_EUNSUPPORTED;
    std::vector<int64_t> ei;
    int st = fetch_index(edge_index, idx_bytes, 2 * n_edges, ei);
    if (st != TIPK_OK) return st;
    const int64_t n_t = n_all - n_source;
    std::vector<int64_t> orow, trow;
    std::vector<double> cnt((size_t)n_t, 0.0);
    for (int64_t i = 0; i < n_edges; ++i) {
        const int64_t s = ei[(size_t)i], d = ei[(size_t)(n_edges + i)];
        if (s < 0 || s >= n_all || d < 0 || d >= n_all) return TIPK_EINVAL;
        if (d >= n_source) { orow.push_back(d - n_source); trow.push_back(s); cnt[(size_t)(d - n_source)] += 1.0; }
    }
    std::vector<float> w(orow.size());
    for (size_t i = 0; i < orow.size(); ++i) w[i] = 1.0f / (float)cnt[(size_t)orow[i]];      // 'mean': sum / count
    tipk_graph* g = new (std::nothrow) tipk_graph;
    if (!g) return TIPK_EINVAL;
    memset(g, 0, sizeof(*g));
    g->kind = 2; g->n_nodes = n_all; g->n_source = n_source;
    st = build_weighted(g, orow, trow, w, n_t, n_all);
    if (st != TIPK_OK) { tipk_graph_destroy(g); return st; }
    *out = g;
    return TIPK_OK;
}

extern "C" int64_t tipk_hier_workspace_bytes(const tipk_graph* g, int d_in, int d_out) {
    if (!g || g->kind != 2 || d_in <= 0 || d_out <= 0) return -1;
    return 2 * align256(g->n_out * (int64_t)d_in * 4) + align256(reduce_slabs(d_in, d_out, g->n_out) * d_in * (int64_t)d_out * 4);
}

// out [n_target x d_out] = mean(x over incoming edges) . weight;  x [n_all x d_in], weight [d_in x d_out] contiguous
extern "C" int tipk_hier_fwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* weight, int d_out, float* out,
                             int64_t ld_out, void* workspace, int64_t workspace_bytes, tipk_stream_t stream) {
    if (!g || g->kind != 2 || !x || !weight || !out || !workspace || d_in <= 0 || d_out <= 0 || ld_x < d_in || ld_out < d_out) return TIPK_EINVAL;
    if (workspace_bytes < tipk_hier_workspace_bytes(g, d_in, d_out) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    float* mean = (float*)workspace;
    GrArgs a;
    memset(&a, 0, sizeof(a));
    a.table = x; a.ld_t = ld_x; a.ptr = g->fwd_ptr; a.row = g->fwd_row; a.n_out = g->n_out; a.edge_w = g->fwd_w; a.d = d_in; a.plan = g->pf;
    a.out = mean; a.ld_out = d_in;
    int st = gr_launch(a, (hipStream_t)stream);
    if (st != TIPK_OK) return st;
    tipk_gemm_desc d = gemm_desc(g->n_out, d_out, d_in, mean, d_in, 1, weight, d_out, 1, out, ld_out);
    return tipk_gemm_f32(&d, stream);
}

extern "C" int tipk_hier_bwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* weight, int d_out,
                             const float* grad_out, int64_t ld_g, float* g_x, int64_t ld_gx, float* g_weight, void* workspace,
                             int64_t workspace_bytes, tipk_stream_t stream) {
    if (!g || g->kind != 2 || !x || !weight || !grad_out || !g_weight || !workspace || d_in <= 0 || d_out <= 0 || ld_x < d_in ||
        ld_g < d_out || (g_x && ld_gx < d_in))
        return TIPK_EINVAL;
    if (workspace_bytes < tipk_hier_workspace_bytes(g, d_in, d_out) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    float* mean = (float*)workspace;
    float* g_mean = (float*)((char*)workspace + align256(g->n_out * (int64_t)d_in * 4));
    GrArgs a;
    memset(&a, 0, sizeof(a));
    a.table = x; a.ld_t = ld_x; a.ptr = g->fwd_ptr; a.row = g->fwd_row; a.n_out = g->n_out; a.edge_w = g->fwd_w; a.d = d_in; a.plan = g->pf;
    a.out = mean; a.ld_out = d_in;
    int st = gr_launch(a, hs);                                                 // the mean again (nothing is kept between the calls)
    if (st != TIPK_OK) return st;
    tipk_gemm_desc d = gemm_desc(d_in, d_out, g->n_out, mean, 1, d_in, grad_out, ld_g, 1, g_weight, d_out);     // mean^T g
    if ((st = reduce_gemm(d, g_mean + align256(g->n_out * (int64_t)d_in * 4) / 4, stream)) != TIPK_OK) return st;
    if (!g_x) return TIPK_OK;
    d = gemm_desc(g->n_out, d_in, d_out, grad_out, ld_g, 1, weight, 1, d_out, g_mean, d_in);                     // g W^T
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    memset(&a, 0, sizeof(a));
    a.table = g_mean; a.ld_t = d_in; a.ptr = g->bwd_ptr; a.row = g->bwd_row; a.n_out = g->n_table; a.edge_w = g->bwd_w; a.d = d_in; a.plan = g->pb;
    a.out = g_x; a.ld_out = ld_gx;
    return gr_launch(a, hs);                                                   // rows nobody reads from: zeros
}
