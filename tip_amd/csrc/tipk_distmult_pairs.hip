// The POSITIVE triples' share of d z of the fused DistMult objective, in PAIR-MAJOR order (include/tipk.h section 4c;
// src/layers.py:335-340 with the decoder of :581-595 under autograd).
//
// distmult_objective_kernel walks the triples relation by relation and scatters two gradient rows per triple into an LDS
// image of d z with 64-bit integer atomics: 25 M row-adds per BioSNAP step, and the number of wave-level ds_add_u64
// instructions is the kernel's floor (195 us).  A third of them belong to the positives -- and the positives are the
// edges of the D-D graph, 66 relations per linked drug pair: with p = z[u] o z[v] fixed per pair,
//
//     s_r = <p, D[r]>,  q_r = d loss / d s_r                        per (pair, relation): one row of D out of LDS
//     W[u, v, :] = sum_{r links u-v} q_r D[r, :]                    a wave-stream gather (the plan of the pair cells, 64-byte rows)
//     d z[u] += sum_{v linked to u} W[u, v, :] o z[v]               one pass over the node's neighbours, no atomics
//
// (DistMult's score is symmetric in (u, v), so W is, and the mirrored triple of a symmetric graph is the same term: the
// objective's weight 2).  The objective kernel keeps the positives' loss and d D (they need no atomics) and scatters the
// negatives only.
#include <stdlib.h>
#include "tipk_common.h"

namespace {

constexpr int DPR_PIECE = 4;            // steps (8 ids each) per cell: the wave-stream plan's piece
constexpr float DPR_EPS = 1e-13f;       // src/layers.py:15

struct DprArgs {
    const float* z; int n_nodes; const float* w; int n_rel;            // z [N][16], D [R][16]
    const int32_t* wave_ptr; const uint32_t* cells; const uint16_t* ids; int idx_mul;
    float coef;                                                          // weight / n_total of a positive (2 / n: mirrored pairs)
    float* wrows;                                                        // [N * N][16]: rows of linked pairs (u <= v) are written
};

__device__ __forceinline__ float dpr_quad_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
    x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xf, 0xf, true));      // quad_perm [2, 3, 0, 1]
    return x;
}

// rows = drug pairs (u * N + v, u <= v), table = D in LDS, 16 slots of 4 lanes per wavefront: tipk_stream_gather's walk with
// a weight per gathered row that depends on the row itself
template <bool UNIT>
__global__ __launch_bounds__(1024) void distmult_pair_rows_kernel(DprArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dpr_lds[];
    float* tab = dpr_lds;                                               // [R + 1][16], last row = 0 (the pad id's row)
    float* zt = tab + (a.n_rel + 1) * 16;                               // [N][16]
    constexpr int L = 4, SPW = 16;
    const int t = threadIdx.x, lane = t & 63;
    const int slot = lane / L, c0 = (lane & (L - 1)) * 4;
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 16 + (t >> 6));
    int b = __builtin_amdgcn_readfirstlane(a.wave_ptr[gw]);
    const int b1 = __builtin_amdgcn_readfirstlane(a.wave_ptr[gw + 1]);
    uint32_t c0q = 0, c1q = 0, c2q = 0, c3q = 0;
    uint4 i0q[DPR_PIECE], i1q[DPR_PIECE], i2q[DPR_PIECE], i3q[DPR_PIECE];
    auto fetch = [&](int band, uint32_t& cw, uint4 (&iw)[DPR_PIECE]) {
        band = band < b1 ? band : b1 - 1;
        cw = a.cells[(int64_t)band * SPW + slot];
        const uint4* p = reinterpret_cast<const uint4*>(a.ids) + ((int64_t)band * (DPR_PIECE * SPW) + slot);
#pragma unroll
        for (int k = 0; k < DPR_PIECE; ++k) iw[k] = p[k * SPW];
    };
    if (b < b1) { fetch(b, c0q, i0q); fetch(b + 1, c1q, i1q); }
    for (int i = t; i < a.n_rel * 4; i += 1024) tipk_st4(tab + 4 * i, tipk_ld4(a.w + 4 * i));
    if (t < 16) tab[a.n_rel * 16 + t] = 0.f;
    for (int i = t; i < a.n_nodes * 4; i += 1024) tipk_st4(zt + 4 * i, tipk_ld4(a.z + 4 * i));
    __syncthreads();                                                    // the only barrier of the launch
    const char* tabb = reinterpret_cast<const char*>(tab + c0);
    const unsigned mul = (unsigned)a.idx_mul;
    const float inv_n = 1.0f / (float)a.n_nodes;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), p = acc;
    auto walk = [&](int band, const uint32_t& cw, const uint4 (&iw)[DPR_PIECE], uint32_t& nw, uint4 (&niw)[DPR_PIECE]) {
        const uint32_t cell = band < b1 ? cw : 0u;                      // past the end: an idle cell
        fetch(band + 2, nw, niw);
        __builtin_amdgcn_sched_barrier(0);
        const int len = (int)((cell >> 24) & 15u);
        if (cell & (1u << 28)) {                                        // first piece of its pair: p = z[u] o z[v]
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const uint32_t row = cell & 0xffffffu;
            uint32_t u = (uint32_t)((float)row * inv_n);                // exact quotient (row < 2^24) up to one step
            int32_t v = (int32_t)(row - u * (uint32_t)a.n_nodes);
            if (v < 0) { --u; v += a.n_nodes; } else if (v >= a.n_nodes) { ++u; v -= a.n_nodes; }
            const float4 zu = tipk_ld4(zt + u * 16 + c0), zv = tipk_ld4(zt + v * 16 + c0);
            p = make_float4(zu.x * zv.x, zu.y * zv.y, zu.z * zv.z, zu.w * zv.w);
        }
#pragma unroll
        for (int k = 0; k < DPR_PIECE; ++k) {
            if (k < len) {
                const unsigned w4[4] = {iw[k].x, iw[k].y, iw[k].z, iw[k].w};
                float4 v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const unsigned idj = (jj & 1) ? (w4[jj >> 1] >> 16) : (w4[jj >> 1] & 0xffffu);
                    v[jj] = *reinterpret_cast<const float4*>(UNIT ? tabb + idj : tabb + __umul24(idj, mul));
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    // s = <p, D[r]> over the slot's 4 lanes; q = -coef (1 - sigma) sigma / (sigma + eps); a pad id reads the zero row:
                    // it adds q * 0
                    const float s = dpr_quad_sum(fmaf(p.x, v[jj].x, fmaf(p.y, v[jj].y, fmaf(p.z, v[jj].z, p.w * v[jj].w))));
                    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-s));
                    const float q = -a.coef * sg * (1.f - sg) * __builtin_amdgcn_rcpf(sg + DPR_EPS);
                    acc.x = fmaf(q, v[jj].x, acc.x); acc.y = fmaf(q, v[jj].y, acc.y);
                    acc.z = fmaf(q, v[jj].z, acc.z); acc.w = fmaf(q, v[jj].w, acc.w);
                }
            }
        }
        // a WIDE run was cut into 2^klog sub-runs in adjacent (aligned) slots: fixed tree, slot s <- slot s + 2^j
        const unsigned klog = cell >> 30;
        if (__builtin_amdgcn_ballot_w64(klog != 0u) != 0ull) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int delta = L << j;
                float4 o;
                o.x = __shfl_down(acc.x, delta, 64); o.y = __shfl_down(acc.y, delta, 64);
                o.z = __shfl_down(acc.z, delta, 64); o.w = __shfl_down(acc.w, delta, 64);
                if ((int)klog > j && (slot & ((2 << j) - 1)) == 0) { acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
            }
        }
        if (cell & (1u << 29)) tipk_st4(a.wrows + (int64_t)(cell & 0xffffffu) * 16 + c0, acc);
    };
    for (; b < b1; b += 4) {
        walk(b, c0q, i0q, c2q, i2q);
        walk(b + 1, c1q, i1q, c3q, i3q);
        walk(b + 2, c2q, i2q, c0q, i0q);
        walk(b + 3, c3q, i3q, c1q, i1q);
    }
}

// d z[u, :] = sum over the neighbours v of u of W[line(u, v), :] o z[v, :]   (k = 16): a workgroup of 4 wavefronts per node,
// a wavefront takes batches of 64 neighbours -- their (v, line) words in ONE coalesced load, then 16 x 2 dword loads per
// lane (lane = column, 4 neighbours side by side) issued back to back
struct DpzArgs {
    const float* z; const float* wrows; const int32_t* nbr_ptr; const int2* nbr; float* g_z; int n_nodes;
};

__global__ __launch_bounds__(256) void distmult_pair_dz_kernel(DpzArgs a) {
    __shared__ float red[4][16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int c = lane & 15, j = lane >> 4;
    const int u = (int)blockIdx.x;
    const int e0 = a.nbr_ptr[u], e1 = a.nbr_ptr[u + 1];
    float acc = 0.f;
    for (int b = e0 + wv * 64; b < e1; b += 256) {
        const int i = b + lane;
        const int2 me = a.nbr[i < e1 ? i : e1 - 1];
        float wv_[16], zv_[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int src = 4 * s + j;
            const int v = __shfl(me.x, src, 64), line = __shfl(me.y, src, 64);
            wv_[s] = a.wrows[(int64_t)line * 16 + c];
            zv_[s] = a.z[(int64_t)v * 16 + c];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (b + 4 * s + j < e1) acc = fmaf(wv_[s], zv_[s], acc);
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 16) red[wv][lane] = acc;
    __syncthreads();
    if (t < 16) a.g_z[(int64_t)u * 16 + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
}

}  // namespace

extern "C" int tipk_distmult_pair_dz(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel, int64_t n_wg,
                                     const int32_t* wave_ptr, const uint32_t* cells, const uint16_t* ids, int idx_unit,
                                     float coef, float* wrows, const int32_t* nbr_ptr, const int32_t* nbr, float* g_z,
                                     tipk_stream_t stream) {
    if (k != 16) return TIPK_EUNSUPPORTED;
    if (!z || !rel_w || !wave_ptr || !cells || !ids || !wrows || !nbr_ptr || !nbr || !g_z || n_nodes <= 0 || n_rel <= 0 ||
        n_wg <= 0 || n_wg > 65535)
        return TIPK_EINVAL;
    if (n_nodes * n_nodes >= (1LL << 24) || n_nodes > 4095) return TIPK_EUNSUPPORTED;
    if (((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(rel_w) | reinterpret_cast<uintptr_t>(ids) |
          reinterpret_cast<uintptr_t>(wrows)) & 15) || (reinterpret_cast<uintptr_t>(nbr) & 7))
        return TIPK_EINVAL;
    const size_t lds = (size_t)((n_rel + 1) * 16 + n_nodes * 16) * 4;
    if (lds > 158 * 1024) return TIPK_EUNSUPPORTED;
    if (idx_unit <= 0 || 64 % idx_unit != 0 || n_rel * idx_unit > 65535) return TIPK_EINVAL;
    DprArgs a;
    a.z = z; a.n_nodes = (int)n_nodes; a.w = rel_w; a.n_rel = (int)n_rel;
    a.wave_ptr = wave_ptr; a.cells = cells; a.ids = ids; a.idx_mul = 64 / idx_unit; a.coef = coef; a.wrows = wrows;
    hipStream_t st = (hipStream_t)stream;
    auto kern = a.idx_mul == 1 ? distmult_pair_rows_kernel<true> : distmult_pair_rows_kernel<false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return tipk_hip_status(e);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(1024), lds, st, a);
    DpzArgs b;
    b.z = z; b.wrows = wrows; b.nbr_ptr = nbr_ptr; b.nbr = reinterpret_cast<const int2*>(nbr); b.g_z = g_z; b.n_nodes = (int)n_nodes;
    hipLaunchKernelGGL(distmult_pair_dz_kernel, dim3((unsigned)n_nodes), dim3(256), 0, st, b);
    TIPK_RETURN_LAUNCH();
}
