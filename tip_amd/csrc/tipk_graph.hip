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
// wavefront per output row; the dense products are the library's own MFMA GEMM (tipk_gemm_f32).  The LDS-resident pair form
// that the PyTorch modules take at BioSNAP size (tip_amd/encoder.py) is faster there; its plans are built by tip_amd/plan.py.
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "tipk_common.h"

struct tipk_graph {
    int64_t n_nodes, n_rel, n_edges;
    int32_t* fwd_ptr;      // [n_nodes + 1]            edges by destination
    int32_t* fwd_row;      // [n_edges]                row (relation * n_nodes + source) of Y
    int32_t* bwd_ptr;      // [n_rel * n_nodes + 1]    edges by (relation, source)
    int32_t* bwd_row;      // [n_edges]                destination
    float* inv_deg;        // [n_nodes]                1 / max(1, in-degree over all relations)
};

namespace {

struct GrArgs {
    const float* table; int64_t ld_t;
    const int32_t* ptr; const int32_t* row; int64_t n_out;
    const float* table_scale;                  // nullable: table row i is used as table_scale[i] * table[i]
    const float* out_scale;                    // nullable: per output row
    const float* addend; int64_t ld_add;       // nullable
    const float* gate; int64_t ld_gate;        // nullable: the TABLE row is masked with gate[i] > 0 (ReLU backward)
    int relu, d;
    float* out; int64_t ld_out;
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
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; e < e1; ++e) {
            const int i0 = a.row[e];
            float v0 = a.table[(int64_t)i0 * a.ld_t + c];
            if (a.gate) v0 = a.gate[(int64_t)i0 * a.ld_gate + c] > 0.f ? v0 : 0.f;
            if (a.table_scale) v0 *= a.table_scale[i0];
            s += v0;
        }
        if (a.out_scale) s *= a.out_scale[r];
        if (a.addend) s += a.addend[r * a.ld_add + c];
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

int gr_launch(const GrArgs& a, hipStream_t st) {
    if (a.n_out <= 0) return TIPK_OK;
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
    g->n_nodes = n_nodes; g->n_rel = n_rel; g->n_edges = n_edges;
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
    *out = g;
    return TIPK_OK;
}

extern "C" int tipk_graph_destroy(tipk_graph* g) {
    if (!g) return TIPK_OK;
    int st = TIPK_OK;
    void* bufs[5] = {g->fwd_ptr, g->fwd_row, g->bwd_ptr, g->bwd_row, g->inv_deg};
    for (void* b : bufs)
        if (b) { const int s = tipk_hip_status(hipFree(b)); if (s != TIPK_OK) st = s; }
    delete g;
    return st;
}

extern "C" int tipk_graph_info(const tipk_graph* g, int64_t* n_nodes, int64_t* n_rel, int64_t* n_edges, const float** inv_degree) {
    if (!g) return TIPK_EINVAL;
    if (n_nodes) *n_nodes = g->n_nodes;
    if (n_rel) *n_rel = g->n_rel;
    if (n_edges) *n_edges = g->n_edges;
    if (inv_degree) *inv_degree = g->inv_deg;
    return TIPK_OK;
}

extern "C" int64_t tipk_rgcn_workspace_bytes(const tipk_graph* g, int d_in, int d_out, int n_bases) {
    if (!g || d_in <= 0 || d_out <= 0 || n_bases <= 0) return -1;
    return carve(nullptr, g->n_nodes, g->n_rel, n_bases, d_out).bytes;
}

extern "C" int tipk_rgcn_fwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                             const float* root, int n_bases, int d_out, int relu, float* out, int64_t ld_out, void* workspace,
                             int64_t workspace_bytes, tipk_stream_t stream) {
    if (!g || !x || !basis || !root || !out || !workspace || d_in <= 0 || d_out <= 0 || n_bases <= 0 || ld_x < d_in || ld_out < d_out ||
        (g->n_rel > 0 && !att))
        return TIPK_EINVAL;
    const int64_t n = g->n_nodes, r = g->n_rel;
    const Ws w = carve(workspace, n, r, n_bases, d_out);
    if (workspace_bytes < w.bytes || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    int st;
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
    a.out_scale = g->inv_deg; a.addend = w.t; a.ld_add = d_out; a.relu = relu; a.d = d_out; a.out = out; a.ld_out = ld_out;
    return gr_launch(a, (hipStream_t)stream);
}

extern "C" int tipk_rgcn_bwd(const tipk_graph* g, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                             const float* root, int n_bases, int d_out, const float* grad_out, int64_t ld_g, const float* out_relu,
                             int64_t ld_relu, float* g_x, int64_t ld_gx, float* g_basis, float* g_att, float* g_root, void* workspace,
                             int64_t workspace_bytes, tipk_stream_t stream) {
    if (!g || !x || !basis || !root || !grad_out || !g_x || !g_basis || !g_root || !workspace || d_in <= 0 || d_out <= 0 || n_bases <= 0 ||
        ld_x < d_in || ld_g < d_out || ld_gx < d_in || (g->n_rel > 0 && (!att || !g_att)) || (out_relu && ld_relu < d_out))
        return TIPK_EINVAL;
    const int64_t n = g->n_nodes, r = g->n_rel;
    const Ws w = carve(workspace, n, r, n_bases, d_out);
    if (workspace_bytes < w.bytes || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TIPK_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    int st;
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
        a.table = gp; a.ld_t = ld_gp; a.ptr = g->bwd_ptr; a.row = g->bwd_row; a.n_out = r * n; a.table_scale = g->inv_deg;
        a.d = d_out; a.out = w.y; a.ld_out = d_out;
        if ((st = gr_launch(a, hs)) != TIPK_OK) return st;
        // d att = dY . XB^T  [R, bases],  d XB = att^T . dY  [bases, N d_out]
        d = gemm_desc(r, n_bases, n * d_out, w.y, n * d_out, 1, w.xb, 1, n * d_out, g_att, n_bases);
        if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
        d = gemm_desc(n_bases, n * d_out, r, att, 1, n_bases, w.y, n * d_out, 1, w.dxb, n * d_out);
        if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    } else {
        if ((st = tipk_hip_status(hipMemsetAsync(w.dxb, 0, (size_t)n_bases * n * d_out * 4, hs))) != TIPK_OK) return st;
    }
    // d basis_b = X^T d XB_b (batched, A shared),  d root = X^T g'
    d = gemm_desc(d_in, d_out, n, x, 1, ld_x, w.dxb, d_out, 1, g_basis, d_out);
    d.batch = n_bases; d.a_sz = 0; d.b_sz = n * d_out; d.c_sz = (int64_t)d_in * d_out;
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    d = gemm_desc(d_in, d_out, n, x, 1, ld_x, gp, ld_gp, 1, g_root, d_out);
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    // dX = g' root^T, then += sum_b d XB_b basis_b^T (the batch reduced inside the product, on top of the first term)
    d = gemm_desc(n, d_in, d_out, gp, ld_gp, 1, root, 1, d_out, g_x, ld_gx);
    if ((st = tipk_gemm_f32(&d, stream)) != TIPK_OK) return st;
    d = gemm_desc(n, d_in, d_out, w.dxb, d_out, 1, basis, 1, d_out, g_x, ld_gx);
    d.kbatch = n_bases; d.a_sq = n * d_out; d.b_sq = (int64_t)d_in * d_out; d.c_in = g_x; d.cin_sm = ld_gx;
    return tipk_gemm_f32(&d, stream);
}
