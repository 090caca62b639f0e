// Host-side plan construction of the pair-form D-D passes (see tipk_pairplan.h): the work lists of the wave-stream gathers
// (tipk_rel_stream.hip) and of the pair-gradient kernel (tipk_pair_grads.hip) from a relation-typed edge list.  Host code only:
// stable counting sorts, one longest-processing-time deal per plan.  The Python package builds the same arrays in
// tip_amd/plan.py (torch ops); tests/test_host_plans.py holds the two against each other array by array.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <map>
#include <new>
#include <numeric>
#include <queue>
#include <string>
#include <utility>
#include "tipk_common.h"
#include "tipk_pairplan.h"

namespace tipk_plan {

namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t pymod(int64_t a, int64_t b) { const int64_t m = a % b; return m < 0 ? m + b : m; }

// indices 0 .. n-1 sorted by key, descending, ties in index order
template <class K> std::vector<int64_t> order_desc(const std::vector<K>& key, const std::vector<int64_t>* subset = nullptr) {
    std::vector<int64_t> o;
    if (subset) o = *subset;
    else { o.resize(key.size()); std::iota(o.begin(), o.end(), (int64_t)0); }
    std::stable_sort(o.begin(), o.end(), [&](int64_t a, int64_t b) { return key[(size_t)a] > key[(size_t)b]; });
    return o;
}

// which of the two 16-lane groups of a half-wave a lane's ds_read_b128 is served in (plan.py _B128_GROUP_OF_LANE)
inline int b128_group_of_lane(int l) {
    const int m = l % 32;
    return (m <= 3 || (m >= 12 && m <= 15) || (m >= 20 && m <= 27)) ? 0 : 1;
}

// class count and per-slot class offset for conflict-free LDS reads (plan.py bank_rotation)
void bank_rotation(int lanes, int& n_cls, std::vector<int>& rot) {
    const int per_wave = 64 / lanes;
    rot.clear();
    if (lanes >= 16) { n_cls = 1; rot.assign((size_t)per_wave, 0); return; }
    if (lanes == 8) {
        static const int r4[4] = {0, 0, 1, 1};
        n_cls = 2;
        for (int k = 0; k < per_wave; ++k) rot.push_back(r4[k % 4]);
        return;
    }
    std::map<int, int> seen;
    for (int k = 0; k < per_wave; ++k) {
        const int g = (k * lanes) / 32 * 2 + b128_group_of_lane(k * lanes);
        const int v = seen.count(g) ? seen[g] : 0;
        rot.push_back(v);
        seen[g] = v + 1;
    }
    n_cls = 16 / lanes;
}

}  // namespace

void build_stream_plan_rows(const int64_t* out_row, const int64_t* tab_row, int64_t E, int64_t n_rows, int64_t T, int n_wg, int lanes,
                            int piece, int wide_steps, int row_bytes, StreamPlanH& sp) {
    const int S = 64 / lanes;
    const int64_t W = (int64_t)n_wg * 16;
    if (wide_steps <= 0) wide_steps = STREAM_WIDE_STEPS;
    if (row_bytes <= 0) row_bytes = lanes * 16;
    // ---- runs: one per output row with edges, in row order
    std::vector<int64_t> cnt_rows((size_t)n_rows, 0);
    for (int64_t e = 0; e < E; ++e) ++cnt_rows[(size_t)out_row[e]];
    std::vector<int64_t> zero_rows, run_row, run_cnt;
    for (int64_t r = 0; r < n_rows; ++r) {
        if (cnt_rows[(size_t)r]) { run_row.push_back(r); run_cnt.push_back(cnt_rows[(size_t)r]); }
        else zero_rows.push_back(r);
    }
    const int64_t n_runs = (int64_t)run_row.size();
    std::vector<int64_t> run_steps((size_t)n_runs), k_run((size_t)n_runs), q_run((size_t)n_runs);
    const int kmax = S < 8 ? S : 8;
    for (int64_t i = 0; i < n_runs; ++i) {
        run_steps[(size_t)i] = (run_cnt[(size_t)i] + 7) / 8;
        const int64_t need = cdiv(run_steps[(size_t)i], wide_steps);
        int64_t k = 1;
        for (int kk = 2; kk <= 8; kk *= 2)
            if (kk <= kmax && need > kk / 2) k = kk;
        k_run[(size_t)i] = k;
        q_run[(size_t)i] = cdiv(run_steps[(size_t)i], k);
    }
    // ---- the slot list in layout order: wide sets first (largest k first, k-aligned), then the plain runs by decreasing steps
    std::vector<int64_t> v_run, v_pos;
    for (int kk = 8; kk >= 2; kk /= 2) {
        std::vector<int64_t> sel;
        for (int64_t i = 0; i < n_runs; ++i)
            if (k_run[(size_t)i] == kk) sel.push_back(i);
        if (sel.empty()) continue;
        sel = order_desc(q_run, &sel);
        for (int64_t s : sel)
            for (int p = 0; p < kk; ++p) { v_run.push_back(s); v_pos.push_back(p); }
        const int64_t pad = pymod(-(int64_t)sel.size() * kk, S);
        for (int64_t i = 0; i < pad; ++i) { v_run.push_back(-1); v_pos.push_back(0); }
    }
    {
        std::vector<int64_t> plain;
        for (int64_t i = 0; i < n_runs; ++i)
            if (k_run[(size_t)i] == 1) plain.push_back(i);
        plain = order_desc(run_steps, &plain);
        for (int64_t s : plain) { v_run.push_back(s); v_pos.push_back(0); }
        const int64_t pad = pymod(-(int64_t)v_run.size(), S);
        for (int64_t i = 0; i < pad; ++i) { v_run.push_back(-1); v_pos.push_back(0); }
    }
    const int64_t V = (int64_t)v_run.size(), G = V / S;
    std::vector<int64_t> v_q((size_t)V), v_steps((size_t)V), v_nb((size_t)V), v_klog((size_t)V), v_rowid((size_t)V);
    for (int64_t i = 0; i < V; ++i) {
        const bool real = v_run[(size_t)i] >= 0;
        const int64_t vr = real ? v_run[(size_t)i] : 0;
        const int64_t k = real ? k_run[(size_t)vr] : 1, q = real ? q_run[(size_t)vr] : 0, tot = real ? run_steps[(size_t)vr] : 0;
        int64_t st = tot - v_pos[(size_t)i] * q;
        if (st < 0) st = 0;
        if (st > q) st = q;
        v_q[(size_t)i] = q;
        v_steps[(size_t)i] = st;
        v_nb[(size_t)i] = real ? std::max<int64_t>(cdiv(q, piece), 1) : 0;
        v_klog[(size_t)i] = k >= 8 ? 3 : k >= 4 ? 2 : k >= 2 ? 1 : 0;
        v_rowid[(size_t)i] = real ? run_row[(size_t)vr] : 0;
    }
    // ---- groups of S slots -> wavefronts: longest processing time first on steps + 2 x bands
    std::vector<int64_t> g_bands((size_t)G);
    std::vector<double> cost((size_t)G);
    for (int64_t g = 0; g < G; ++g) {
        int64_t m = 0;
        for (int s = 0; s < S; ++s) m = std::max(m, v_q[(size_t)(g * S + s)]);
        g_bands[(size_t)g] = std::max<int64_t>(cdiv(m, piece), 1);
        cost[(size_t)g] = (double)m + 2.0 * (double)g_bands[(size_t)g];
    }
    const std::vector<int64_t> by_cost = order_desc(cost);
    std::vector<int64_t> wave_of((size_t)G);
    {
        typedef std::pair<double, int64_t> Load;
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        for (int64_t w = 0; w < W; ++w) heap.push(Load(0.0, w));
        for (int64_t g : by_cost) {
            const Load top = heap.top();
            heap.pop();
            wave_of[(size_t)g] = top.second;
            heap.push(Load(top.first + cost[(size_t)g], top.second));
        }
    }
    std::vector<int64_t> g_order((size_t)G);
    std::iota(g_order.begin(), g_order.end(), (int64_t)0);
    std::stable_sort(g_order.begin(), g_order.end(), [&](int64_t a, int64_t b) { return wave_of[(size_t)a] < wave_of[(size_t)b]; });
    std::vector<int64_t> band0_l((size_t)G), g_band0((size_t)G), per_wave((size_t)W, 0);
    int64_t n_bands = 0;
    for (int64_t i = 0; i < G; ++i) {
        const int64_t g = g_order[(size_t)i];
        band0_l[(size_t)i] = n_bands;
        g_band0[(size_t)g] = n_bands;
        n_bands += g_bands[(size_t)g];
        per_wave[(size_t)wave_of[(size_t)g]] += g_bands[(size_t)g];
    }
    sp.wave_ptr.assign((size_t)W + 1, 0);
    for (int64_t w = 0; w < W; ++w) sp.wave_ptr[(size_t)w + 1] = sp.wave_ptr[(size_t)w] + (int32_t)per_wave[(size_t)w];
    // ---- cells
    sp.cells.assign((size_t)(n_bands * S), 0);
    for (int64_t i = 0; i < G; ++i) {
        const int64_t g = g_order[(size_t)i];
        for (int64_t k = 0; k < g_bands[(size_t)g]; ++k) {
            const int64_t band = band0_l[(size_t)i] + k;
            for (int s = 0; s < S; ++s) {
                const int64_t vi = g * S + s;
                const bool rl = v_run[(size_t)vi] >= 0;
                const int64_t nb_set = v_nb[(size_t)vi];
                if (!(rl && k < nb_set)) continue;
                int64_t length = v_steps[(size_t)vi] - k * piece;
                if (length < 0) length = 0;
                if (length > piece) length = piece;
                const bool final_ = k == nb_set - 1;
                const uint32_t first = k == 0, last = final_ && v_pos[(size_t)vi] == 0;
                const uint32_t c = (uint32_t)v_rowid[(size_t)vi] | ((uint32_t)length << 24) | (first << 28) | (last << 29) |
                                   ((final_ ? (uint32_t)v_klog[(size_t)vi] : 0u) << 30);
                sp.cells[(size_t)(band * S + s)] = (int32_t)c;
            }
        }
    }
    // ---- ids: pre-scaled to row offsets, every sub-run ordered by bank class
    int idx_unit = 1;
    while (idx_unit * 2 <= row_bytes && T * idx_unit * 2 <= 65535) idx_unit *= 2;
    std::vector<int64_t> v_first((size_t)std::max<int64_t>(n_runs, 1), -1);
    for (int64_t i = 0; i < V; ++i)
        if (v_run[(size_t)i] >= 0 && v_pos[(size_t)i] == 0) v_first[(size_t)v_run[(size_t)i]] = i;
    std::vector<int64_t> run_of_row((size_t)n_rows, -1), run_first((size_t)n_runs + 1, 0);
    for (int64_t i = 0; i < n_runs; ++i) {
        run_of_row[(size_t)run_row[(size_t)i]] = i;
        run_first[(size_t)i + 1] = run_first[(size_t)i] + run_cnt[(size_t)i];
    }
    int n_cls;
    std::vector<int> rot;
    if (row_bytes == 8) {
        n_cls = 32;
        for (int k = 0; k < S; ++k) rot.push_back(k % 32);
    } else {
        bank_rotation(lanes, n_cls, rot);
    }
    // edges by run (edge order inside a run): virtual run, table row and class of every edge
    std::vector<int64_t> ve((size_t)E), tab1((size_t)E);
    {
        std::vector<int64_t> pos(run_first.begin(), run_first.end() - 1);
        for (int64_t e = 0; e < E; ++e) {
            const int64_t ri = run_of_row[(size_t)out_row[e]];
            const int64_t p = pos[(size_t)ri]++;
            const int64_t j1 = p - run_first[(size_t)ri];
            ve[(size_t)p] = v_first[(size_t)ri] + j1 / (q_run[(size_t)ri] * 8);
            tab1[(size_t)p] = tab_row[e];
        }
    }
    // stable counting sort by (virtual run, class)
    std::vector<int64_t> bucket((size_t)(V * n_cls) + 1, 0);
    std::vector<int32_t> key((size_t)E);
    for (int64_t p = 0; p < E; ++p) {
        const int64_t cls = pymod(tab1[(size_t)p] % n_cls - rot[(size_t)(ve[(size_t)p] % S)], n_cls);
        key[(size_t)p] = (int32_t)(ve[(size_t)p] * n_cls + cls);
        ++bucket[(size_t)key[(size_t)p] + 1];
    }
    for (size_t i = 1; i < bucket.size(); ++i) bucket[i] += bucket[i - 1];
    sp.ids.assign((size_t)(std::max<int64_t>(n_bands, 1) * piece * S * 8), (uint16_t)(T * idx_unit));
    {
        std::vector<int64_t> v_start((size_t)V);
        for (int64_t v = 0; v < V; ++v) v_start[(size_t)v] = bucket[(size_t)(v * n_cls)];
        std::vector<int64_t> pos(bucket.begin(), bucket.end() - 1);
        for (int64_t p = 0; p < E; ++p) {
            const int64_t at = pos[(size_t)key[(size_t)p]]++;
            const int64_t v = ve[(size_t)p];
            const int64_t jr = at - v_start[(size_t)v];
            const int64_t step = jr / 8;
            const int64_t band = g_band0[(size_t)(v / S)] + step / piece;
            const int64_t dest = ((band * piece + step % piece) * S + v % S) * 8 + jr % 8;
            sp.ids[(size_t)dest] = (uint16_t)(tab1[(size_t)p] * idx_unit);
        }
    }
    const int64_t nz = (int64_t)zero_rows.size();
    sp.zero_ptr.resize((size_t)W + 1);
    for (int64_t w = 0; w <= W; ++w) sp.zero_ptr[(size_t)w] = (int32_t)((w * nz) / W);
    sp.zero_rows.clear();
    for (int64_t r : zero_rows) sp.zero_rows.push_back((int32_t)r);
    if (sp.zero_rows.empty()) sp.zero_rows.push_back(0);
    sp.n_rows = n_rows; sp.n_table = T; sp.n_bands = n_bands; sp.n_edges = E;
    sp.n_wg = n_wg; sp.lanes = lanes; sp.piece = piece; sp.idx_unit = idx_unit; sp.row_bytes = row_bytes;
}

bool build_pair_bwd_plan(const int64_t* src, const int64_t* dst, const int64_t* rel, int64_t n_edges, int64_t N, int64_t R,
                         const float* scale, bool symmetric, int n_wg_in, int lanes, int piece, PairBwdH& pb) {
    const int64_t ls = N;
    // ---- directed pairs, grouped by source node, neighbours ascending; 32 slots per tile
    std::vector<int64_t> key((size_t)n_edges);
    for (int64_t e = 0; e < n_edges; ++e) key[(size_t)e] = src[e] * N + dst[e];
    std::sort(key.begin(), key.end());
    key.erase(std::unique(key.begin(), key.end()), key.end());
    const int64_t n_dp = (int64_t)key.size();
    std::vector<int64_t> pu((size_t)n_dp), pv((size_t)n_dp), deg((size_t)N, 0);
    for (int64_t i = 0; i < n_dp; ++i) { pu[(size_t)i] = key[(size_t)i] / N; pv[(size_t)i] = key[(size_t)i] % N; ++deg[(size_t)pu[(size_t)i]]; }
    std::vector<int64_t> tiles((size_t)N), first_tile((size_t)N), first_pair((size_t)N);
    int64_t n_tiles = 0, acc_p = 0;
    for (int64_t u = 0; u < N; ++u) {
        tiles[(size_t)u] = (deg[(size_t)u] + 31) / 32;
        first_tile[(size_t)u] = n_tiles; n_tiles += tiles[(size_t)u];
        first_pair[(size_t)u] = acc_p; acc_p += deg[(size_t)u];
    }
    const int64_t n_slots = n_tiles * 32;
    if (n_slots <= 0) return false;
    std::vector<int64_t> slot((size_t)n_dp), line((size_t)n_dp);
    for (int64_t i = 0; i < n_dp; ++i) {
        const int64_t u = pu[(size_t)i], v = pv[(size_t)i];
        slot[(size_t)i] = first_tile[(size_t)u] * 32 + (i - first_pair[(size_t)u]);
        line[(size_t)i] = symmetric ? (u <= v ? u * ls + v : v * ls + u) : u * ls + v;
        if (line[(size_t)i] >= (1 << 24)) return false;
    }
    // ---- the table of the d att gather: one row per pair walked
    std::vector<int64_t> t_key, e_key, e_rel, row_of((size_t)n_dp);
    if (symmetric) {
        for (int64_t i = 0; i < n_dp; ++i)
            if (pu[(size_t)i] <= pv[(size_t)i]) t_key.push_back(key[(size_t)i]);
        for (int64_t e = 0; e < n_edges; ++e)
            if (src[e] <= dst[e]) { e_key.push_back(src[e] * N + dst[e]); e_rel.push_back(rel[e]); }
        for (int64_t i = 0; i < n_dp; ++i) {
            const int64_t lo = std::min(pu[(size_t)i], pv[(size_t)i]), hi = std::max(pu[(size_t)i], pv[(size_t)i]);
            const auto it = std::lower_bound(t_key.begin(), t_key.end(), lo * N + hi);
            if (it == t_key.end() || *it != lo * N + hi) return false;         // the graph is not symmetric
            row_of[(size_t)i] = it - t_key.begin();
        }
    } else {
        t_key = key;
        e_key.resize((size_t)n_edges); e_rel.assign(rel, rel + n_edges);
        for (int64_t e = 0; e < n_edges; ++e) e_key[(size_t)e] = src[e] * N + dst[e];
        std::iota(row_of.begin(), row_of.end(), (int64_t)0);
    }
    const int64_t n_t = (int64_t)t_key.size(), E = (int64_t)e_key.size();
    std::vector<int64_t> e_row((size_t)E);
    for (int64_t e = 0; e < E; ++e) e_row[(size_t)e] = std::lower_bound(t_key.begin(), t_key.end(), e_key[(size_t)e]) - t_key.begin();
    // ---- partitions: equal numbers of rows AND of edges -- the pairs dealt like cards, heaviest first, back and forth
    const int64_t cap = PAIR_PART_ROWS;
    if ((cap + 1) * lanes * 16 > 158 * 1024) return false;
    int64_t n_parts = std::max<int64_t>(1, cdiv(n_t, cap));
    if (E >= 2 * (int64_t)PAIR_PART_WGS * PAIR_PART_EDGES_PER_WG)
        n_parts = std::max(n_parts, std::min<int64_t>(n_wg_in / PAIR_PART_WGS, cdiv(E, (int64_t)PAIR_PART_WGS * PAIR_PART_EDGES_PER_WG)));
    const int64_t part_len = cdiv(cdiv(n_t, n_parts), 8) * 8;
    std::vector<int64_t> row_edges((size_t)n_t, 0);
    for (int64_t e = 0; e < E; ++e) ++row_edges[(size_t)e_row[(size_t)e]];
    const std::vector<int64_t> by_load = order_desc(row_edges);
    std::vector<int64_t> part_of((size_t)n_t), p_cnt((size_t)n_parts, 0);
    for (int64_t i = 0; i < n_t; ++i) {
        const int64_t k = i / n_parts, j = i % n_parts;
        part_of[(size_t)by_load[(size_t)i]] = (k % 2 == 0) ? j : n_parts - 1 - j;
    }
    for (int64_t i = 0; i < n_t; ++i) ++p_cnt[(size_t)part_of[(size_t)i]];
    std::vector<int64_t> new_row((size_t)n_t);
    {
        // inside a partition the pairs keep their (u, v) order
        std::vector<int64_t> fill((size_t)n_parts, 0);
        for (int64_t i = 0; i < n_t; ++i) {
            const int64_t p = part_of[(size_t)i];
            new_row[(size_t)i] = p * part_len + fill[(size_t)p]++;
        }
    }
    for (int64_t e = 0; e < E; ++e) e_row[(size_t)e] = new_row[(size_t)e_row[(size_t)e]];
    for (int64_t i = 0; i < n_dp; ++i) row_of[(size_t)i] = new_row[(size_t)row_of[(size_t)i]];
    const int64_t n_alloc = n_parts * part_len;
    if (2 * n_alloc + 1 >= (1 << 25) || n_parts * R >= (1 << 24)) return false;
    pb.part_first.resize((size_t)n_parts);
    for (int64_t p = 0; p < n_parts; ++p) pb.part_first[(size_t)p] = (int32_t)(p * part_len);
    // ---- slots {v, 1 / deg(v), cell line, destination row of the gradient row}; pads: the node's own first neighbour / line
    // with the factor 0, their gradient rows go to the dump row
    pb.slots.assign((size_t)n_slots * 4, 0);
    for (int64_t u = 0; u < N; ++u) {
        if (!tiles[(size_t)u]) continue;
        const int64_t fp = std::min(first_pair[(size_t)u], std::max<int64_t>(n_dp - 1, 0));
        for (int64_t s = first_tile[(size_t)u] * 32; s < (first_tile[(size_t)u] + tiles[(size_t)u]) * 32; ++s) {
            pb.slots[(size_t)s * 4 + 0] = (int32_t)pv[(size_t)fp];
            pb.slots[(size_t)s * 4 + 1] = 0;
            pb.slots[(size_t)s * 4 + 2] = (int32_t)line[(size_t)fp];
            pb.slots[(size_t)s * 4 + 3] = (int32_t)(2 * n_alloc);
        }
    }
    for (int64_t i = 0; i < n_dp; ++i) {
        const int64_t s = slot[(size_t)i];
        const float sc = scale[(size_t)pv[(size_t)i]];
        int32_t bits;
        memcpy(&bits, &sc, 4);
        pb.slots[(size_t)s * 4 + 0] = (int32_t)pv[(size_t)i];
        pb.slots[(size_t)s * 4 + 1] = bits;
        pb.slots[(size_t)s * 4 + 2] = (int32_t)line[(size_t)i];
        pb.slots[(size_t)s * 4 + 3] = (int32_t)(symmetric ? (pu[(size_t)i] <= pv[(size_t)i] ? row_of[(size_t)i] : n_alloc + row_of[(size_t)i])
                                                          : row_of[(size_t)i]);
    }
    {
        const std::vector<int64_t> order = order_desc(tiles);
        pb.node_desc.resize((size_t)N * 4);
        for (int64_t i = 0; i < N; ++i) {
            const int64_t u = order[(size_t)i];
            pb.node_desc[(size_t)i * 4 + 0] = (int32_t)u;
            pb.node_desc[(size_t)i * 4 + 1] = (int32_t)(first_tile[(size_t)u] * 32);
            pb.node_desc[(size_t)i * 4 + 2] = (int32_t)tiles[(size_t)u];
            pb.node_desc[(size_t)i * 4 + 3] = 0;
        }
        pb.tile_node.clear();
        for (int64_t u = 0; u < N; ++u)
            for (int64_t t = 0; t < tiles[(size_t)u]; ++t) pb.tile_node.push_back((int32_t)u);
    }
    // ---- workgroups per partition; one wave-stream plan per partition, cut into its workgroups' pieces
    const int64_t per = std::max<int64_t>(1, std::min<int64_t>(n_wg_in / n_parts, E ? cdiv(E, n_parts * PAIR_PART_EDGES_PER_WG) : 1));
    const int S = 64 / lanes;
    std::vector<std::vector<int64_t>> sel_rel((size_t)n_parts), sel_loc((size_t)n_parts);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t p = e_row[(size_t)e] / part_len;
        sel_rel[(size_t)p].push_back(e_rel[(size_t)e]);
        sel_loc[(size_t)p].push_back(e_row[(size_t)e] % part_len);
    }
    std::vector<StreamPlanH> plans((size_t)n_parts);
    int idx_unit = 0;
    for (int64_t p = 0; p < n_parts; ++p) {
        build_stream_plan_rows(sel_rel[(size_t)p].data(), sel_loc[(size_t)p].data(), (int64_t)sel_rel[(size_t)p].size(), R, part_len, (int)per,
                               lanes, piece, 0, 0, plans[(size_t)p]);
        if (idx_unit && idx_unit != plans[(size_t)p].idx_unit) return false;
        idx_unit = plans[(size_t)p].idx_unit;
        std::vector<int64_t>().swap(sel_rel[(size_t)p]);
        std::vector<int64_t>().swap(sel_loc[(size_t)p]);
    }
    // ... laid out in LAUNCH order: workgroup b runs on XCD b mod 8, and the workgroups that stage the same partition share an XCD
    const int n_xcd = 8;
    std::vector<std::vector<std::pair<int64_t, int64_t>>> per_xcd((size_t)n_xcd);
    std::vector<size_t> head((size_t)n_xcd, 0);
    for (int x = 0; x < n_xcd; ++x)
        for (int64_t p = x; p < n_parts; p += n_xcd)
            for (int64_t q = 0; q < per; ++q) per_xcd[(size_t)x].push_back(std::make_pair(p, q));
    std::vector<std::pair<int64_t, int64_t>> order;
    auto left = [&](int x) { return per_xcd[(size_t)x].size() - head[(size_t)x]; };
    for (;;) {
        bool any = false;
        for (int x = 0; x < n_xcd; ++x) any = any || left(x) > 0;
        if (!any) break;
        for (int x = 0; x < n_xcd; ++x) {
            int from = x;
            if (left(x) == 0) {                                        // the first list with the most entries left
                from = 0;
                for (int y = 1; y < n_xcd; ++y)
                    if (left(y) > left(from)) from = y;
            }
            if (left(from) > 0) order.push_back(per_xcd[(size_t)from][head[(size_t)from]++]);
        }
    }
    StreamPlanH& gp = pb.gather;
    gp = StreamPlanH();
    int64_t band0 = 0, z0 = 0;
    pb.wg_part.clear();
    const int64_t band_ids = (int64_t)piece * S * 8;
    for (const auto& pq : order) {
        const int64_t p = pq.first, q = pq.second;
        const StreamPlanH& sp = plans[(size_t)p];
        const int64_t b0 = sp.wave_ptr[(size_t)(16 * q)], b1 = sp.wave_ptr[(size_t)(16 * q + 16)];
        const int64_t y0 = sp.zero_ptr[(size_t)(16 * q)], y1 = sp.zero_ptr[(size_t)(16 * q + 16)];
        for (int w = 0; w < 16; ++w) {
            gp.wave_ptr.push_back((int32_t)(sp.wave_ptr[(size_t)(16 * q + w)] - b0 + band0));
            gp.zero_ptr.push_back((int32_t)(sp.zero_ptr[(size_t)(16 * q + w)] - y0 + z0));
        }
        for (int64_t i = b0 * S; i < b1 * S; ++i) {
            const uint32_t c = (uint32_t)sp.cells[(size_t)i];
            gp.cells.push_back((int32_t)(c != 0 ? c + (uint32_t)(p * R) : c));            // output row = p * R + relation
        }
        gp.ids.insert(gp.ids.end(), sp.ids.begin() + b0 * band_ids, sp.ids.begin() + b1 * band_ids);
        for (int64_t i = y0; i < y1; ++i) gp.zero_rows.push_back((int32_t)(sp.zero_rows[(size_t)i] + p * R));
        band0 += b1 - b0;
        z0 += y1 - y0;
        pb.wg_part.push_back((int32_t)p);
    }
    gp.wave_ptr.push_back((int32_t)band0);
    gp.zero_ptr.push_back((int32_t)z0);
    if (!band0) gp.ids.assign((size_t)band_ids, (uint16_t)(part_len * idx_unit));
    if (!z0) gp.zero_rows.assign(1, 0);
    gp.n_rows = n_parts * R; gp.n_table = part_len; gp.n_bands = band0; gp.n_edges = E;
    gp.n_wg = (int)(per * n_parts); gp.lanes = lanes; gp.piece = piece; gp.idx_unit = idx_unit; gp.row_bytes = lanes * 16;
    pb.n_nodes = N; pb.n_rel = R; pb.n_slots = n_slots; pb.n_parts = n_parts; pb.part_len = part_len; pb.n_alloc = n_alloc;
    pb.symmetric = symmetric;
    return true;
}

int group_slots_for(int d) {
    int lanes = 1;
    const int need = d % 4 == 0 ? d / 4 : d;
    while (lanes < need) lanes *= 2;
    return std::max(std::min(128, 1024 / lanes), 64 / lanes);
}

void build_gather_plan(const int64_t* out_row, const int64_t* table_row, const float* edge_w, int64_t E, int64_t n_out, int64_t n_table,
                       int chunk, int G, GatherPlanH& gp) {
    enum { ITEM_DIRECT = 1, ITEM_PIECE = 2, ITEM_LEADER = 4, ITEM_NULL = 8 };
    if (chunk <= 0) {                                                          // plan.py auto_chunk
        const int64_t target = E <= (1 << 22) ? 65536 * 2 : 65536;
        chunk = 16;
        while (chunk < 128 && E / chunk > target) chunk *= 2;
    }
    // edges by output row, stable
    std::vector<int64_t> row_ptr((size_t)n_out + 1, 0);
    for (int64_t e = 0; e < E; ++e) ++row_ptr[(size_t)out_row[e] + 1];
    for (int64_t r = 0; r < n_out; ++r) row_ptr[(size_t)r + 1] += row_ptr[(size_t)r];
    gp.perm.resize((size_t)E); gp.row_id.resize((size_t)E);
    gp.edge_w.clear();
    if (edge_w) gp.edge_w.resize((size_t)E);
    {
        std::vector<int64_t> pos(row_ptr.begin(), row_ptr.end() - 1);
        for (int64_t e = 0; e < E; ++e) {
            const int64_t p = pos[(size_t)out_row[e]]++;
            gp.perm[(size_t)p] = e;
            gp.row_id[(size_t)p] = (int32_t)table_row[e];
            if (edge_w) gp.edge_w[(size_t)p] = edge_w[e];
        }
    }
    // pieces per row (hub rows: at most G balanced pieces, one workgroup)
    std::vector<int64_t> n_chunks((size_t)n_out), piece_len((size_t)n_out);
    for (int64_t r = 0; r < n_out; ++r) {
        const int64_t cnt = row_ptr[(size_t)r + 1] - row_ptr[(size_t)r];
        int64_t nc = std::max<int64_t>(cdiv(cnt, chunk), 1);
        nc = std::min<int64_t>(nc, G);
        n_chunks[(size_t)r] = nc;
        piece_len[(size_t)r] = std::max<int64_t>(cdiv(cnt, nc), 1);
    }
    auto piece = [&](int64_t r, int64_t local, int64_t& begin, int64_t& end) {
        const int64_t row_end = row_ptr[(size_t)r + 1];
        begin = std::min(row_ptr[(size_t)r] + local * piece_len[(size_t)r], row_end);
        end = std::min(begin + piece_len[(size_t)r], row_end);
    };
    // split rows: best-fit-decreasing packing of their pieces into blocks of G consecutive slots (plan.py pack_blocks)
    std::vector<int64_t> split;
    for (int64_t r = 0; r < n_out; ++r)
        if (n_chunks[(size_t)r] > 1) split.push_back(r);
    std::vector<int64_t> order(split.size());
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return n_chunks[(size_t)split[(size_t)a]] > n_chunks[(size_t)split[(size_t)b]]; });
    std::vector<std::vector<int64_t>> free_((size_t)G + 1);
    std::vector<int64_t> used, block(split.size()), offset(split.size());
    for (int64_t i : order) {
        const int64_t pc = n_chunks[(size_t)split[(size_t)i]];
        int64_t b = -1;
        for (int64_t c = pc; c <= G; ++c)
            if (!free_[(size_t)c].empty()) { b = free_[(size_t)c].back(); free_[(size_t)c].pop_back(); break; }
        if (b < 0) { b = (int64_t)used.size(); used.push_back(0); }
        block[(size_t)i] = b; offset[(size_t)i] = used[(size_t)b];
        used[(size_t)b] += pc;
        if (G - used[(size_t)b] > 0) free_[(size_t)(G - used[(size_t)b])].push_back(b);
    }
    const int64_t n_blocks = (int64_t)used.size();
    // direct rows, longest first (stable)
    std::vector<int64_t> direct;
    for (int64_t r = 0; r < n_out; ++r)
        if (n_chunks[(size_t)r] == 1) direct.push_back(r);
    std::stable_sort(direct.begin(), direct.end(), [&](int64_t a, int64_t b) {
        return row_ptr[(size_t)a + 1] - row_ptr[(size_t)a] > row_ptr[(size_t)b + 1] - row_ptr[(size_t)b];
    });
    gp.n_items = n_blocks * G + (int64_t)direct.size();
    gp.items.assign((size_t)gp.n_items * 4, 0);
    for (int64_t i = 0; i < n_blocks * G; ++i) gp.items[(size_t)i * 4 + 3] = ITEM_NULL;
    for (size_t i = 0; i < split.size(); ++i) {
        const int64_t r = split[i], pc = n_chunks[(size_t)r], base = block[i] * G + offset[i];
        for (int64_t l = 0; l < pc; ++l) {
            int64_t b, e;
            piece(r, l, b, e);
            int32_t* it = &gp.items[(size_t)(base + l) * 4];
            it[0] = (int32_t)b; it[1] = (int32_t)e; it[2] = (int32_t)r;
            it[3] = (int32_t)(l == 0 ? (ITEM_PIECE | ITEM_LEADER | (pc << 8)) : ITEM_PIECE);
        }
    }
    for (size_t i = 0; i < direct.size(); ++i) {
        const int64_t r = direct[i];
        int64_t b, e;
        piece(r, 0, b, e);
        int32_t* it = &gp.items[(size_t)(n_blocks * G + (int64_t)i) * 4];
        it[0] = (int32_t)b; it[1] = (int32_t)e; it[2] = (int32_t)r; it[3] = ITEM_DIRECT;
    }
    gp.n_out = n_out; gp.n_table = n_table; gp.n_edges = E; gp.chunk = chunk; gp.group_slots = G;
}

void pair_link_words(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n, std::vector<uint32_t>& words) {
    const int64_t n8 = cdiv(n, 8) * 8, lw = cdiv(n, 32);
    words.assign((size_t)(n8 * lw), 0u);
    for (int64_t e = 0; e < n_edges; ++e) words[(size_t)(src[e] * lw + (dst[e] >> 5))] |= 1u << (dst[e] & 31);
}

bool relations_symmetric(const int64_t* src, const int64_t* dst, const int64_t* rel, int64_t n_edges, int64_t N) {
    std::vector<int64_t> fw((size_t)n_edges), bw((size_t)n_edges);
    for (int64_t e = 0; e < n_edges; ++e) {
        fw[(size_t)e] = (rel[e] * N + src[e]) * N + dst[e];
        bw[(size_t)e] = (rel[e] * N + dst[e]) * N + src[e];
    }
    std::sort(fw.begin(), fw.end());
    std::sort(bw.begin(), bw.end());
    return fw == bw;
}

}  // namespace tipk_plan

// ---------------------------------------------------------------------------------------------------------------------
// The plans as host arrays through the C ABI (include/tipk.h section 10c): what tests/test_host_plans.py compares with
// tip_amd/plan.py, and what a host that keeps its own device buffers can build its launches from.

struct tipk_host_plan {
    std::map<std::string, std::vector<char>> arrays;
    std::map<std::string, int> elem;
    std::map<std::string, int64_t> scalars;
    template <class T> void put(const char* name, const std::vector<T>& v) {
        std::vector<char>& a = arrays[name];
        a.resize(v.size() * sizeof(T));
        if (!v.empty()) memcpy(a.data(), v.data(), a.size());
        elem[name] = (int)sizeof(T);
    }
};

namespace {

void put_stream(tipk_host_plan* h, const tipk_plan::StreamPlanH& sp, const std::string& pre) {
    h->put((pre + "wave_ptr").c_str(), sp.wave_ptr);
    h->put((pre + "cells").c_str(), sp.cells);
    h->put((pre + "ids").c_str(), sp.ids);
    h->put((pre + "zero_ptr").c_str(), sp.zero_ptr);
    h->put((pre + "zero_rows").c_str(), sp.zero_rows);
    h->scalars[pre + "n_rows"] = sp.n_rows; h->scalars[pre + "n_table"] = sp.n_table; h->scalars[pre + "n_bands"] = sp.n_bands;
    h->scalars[pre + "n_edges"] = sp.n_edges; h->scalars[pre + "n_wg"] = sp.n_wg; h->scalars[pre + "lanes"] = sp.lanes;
    h->scalars[pre + "piece"] = sp.piece; h->scalars[pre + "idx_unit"] = sp.idx_unit; h->scalars[pre + "row_bytes"] = sp.row_bytes;
}

}  // namespace

extern "C" int tipk_plan_stream_rows(const int64_t* out_row, const int64_t* tab_row, int64_t n_edges, int64_t n_rows, int64_t n_table,
                                     int n_wg, int lanes, int piece, int wide_steps, int row_bytes, tipk_host_plan** out) {
    if (!out) return TIPK_EINVAL;
    *out = nullptr;
    if (n_edges < 0 || (n_edges > 0 && (!out_row || !tab_row)) || n_rows <= 0 || n_rows >= (1 << 24) || n_table <= 0 || n_table > 65535 ||
        n_wg <= 0 || piece <= 0 || (lanes != 1 && lanes != 2 && lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32 && lanes != 64) ||
        !(row_bytes == 0 || row_bytes == lanes * 16 || (row_bytes == 8 && lanes == 1)))
        return TIPK_EINVAL;
    for (int64_t e = 0; e < n_edges; ++e)
        if (out_row[e] < 0 || out_row[e] >= n_rows || tab_row[e] < 0 || tab_row[e] >= n_table) return TIPK_EINVAL;
    tipk_host_plan* h = new (std::nothrow) tipk_host_plan;
    if (!h) return TIPK_EINVAL;
    tipk_plan::StreamPlanH sp;
    tipk_plan::build_stream_plan_rows(out_row, tab_row, n_edges, n_rows, n_table, n_wg, lanes, piece, wide_steps, row_bytes, sp);
    put_stream(h, sp, "");
    *out = h;
    return TIPK_OK;
}

extern "C" int tipk_plan_pair_bwd(const int64_t* src, const int64_t* dst, const int64_t* rel, int64_t n_edges, int64_t n_nodes,
                                  int64_t n_rel, const float* scale, int symmetric, int n_wg, int lanes, int piece, tipk_host_plan** out) {
    if (!out) return TIPK_EINVAL;
    *out = nullptr;
    if (n_edges <= 0 || !src || !dst || !rel || !scale || n_nodes <= 0 || n_rel <= 0 || n_wg <= 0 || piece <= 0 ||
        (lanes != 4 && lanes != 8 && lanes != 16))
        return TIPK_EINVAL;
    for (int64_t e = 0; e < n_edges; ++e)
        if (src[e] < 0 || src[e] >= n_nodes || dst[e] < 0 || dst[e] >= n_nodes || rel[e] < 0 || rel[e] >= n_rel) return TIPK_EINVAL;
    tipk_host_plan* h = new (std::nothrow) tipk_host_plan;
    if (!h) return TIPK_EINVAL;
    tipk_plan::PairBwdH pb;
    if (!tipk_plan::build_pair_bwd_plan(src, dst, rel, n_edges, n_nodes, n_rel, scale, symmetric != 0, n_wg, lanes, piece, pb)) {
        delete h;
        return TIPK_EUNSUPPORTED;
    }
    h->put("slots", pb.slots); h->put("node_desc", pb.node_desc); h->put("tile_node", pb.tile_node);
    h->put("part_first", pb.part_first); h->put("wg_part", pb.wg_part);
    h->scalars["n_slots"] = pb.n_slots; h->scalars["n_parts"] = pb.n_parts; h->scalars["part_len"] = pb.part_len;
    h->scalars["n_alloc"] = pb.n_alloc; h->scalars["symmetric"] = pb.symmetric;
    put_stream(h, pb.gather, "gather.");
    *out = h;
    return TIPK_OK;
}

extern "C" int tipk_plan_gather(const int64_t* out_row, const int64_t* table_row, const float* edge_w, int64_t n_edges, int64_t n_out,
                                int64_t n_table, int chunk, int group_slots, tipk_host_plan** out) {
    if (!out) return TIPK_EINVAL;
    *out = nullptr;
    if (n_edges < 0 || (n_edges > 0 && (!out_row || !table_row)) || n_out <= 0 || n_table <= 0 || chunk < 0 || group_slots <= 0 ||
        group_slots > 1024 || n_edges >= 0x7fffffffLL || n_out >= 0x7fffffffLL || n_table >= 0x7fffffffLL)
        return TIPK_EINVAL;
    for (int64_t e = 0; e < n_edges; ++e)
        if (out_row[e] < 0 || out_row[e] >= n_out || table_row[e] < 0 || table_row[e] >= n_table) return TIPK_EINVAL;
    tipk_host_plan* h = new (std::nothrow) tipk_host_plan;
    if (!h) return TIPK_EINVAL;
    tipk_plan::GatherPlanH gp;
    tipk_plan::build_gather_plan(out_row, table_row, edge_w, n_edges, n_out, n_table, chunk, group_slots, gp);
    h->put("row_id", gp.row_id); h->put("edge_w", gp.edge_w); h->put("items", gp.items); h->put("perm", gp.perm);
    h->scalars["n_items"] = gp.n_items; h->scalars["chunk"] = gp.chunk; h->scalars["group_slots"] = gp.group_slots;
    h->scalars["n_edges"] = gp.n_edges; h->scalars["n_out"] = gp.n_out; h->scalars["n_table"] = gp.n_table;
    *out = h;
    return TIPK_OK;
}

extern "C" int tipk_plan_link_words(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, tipk_host_plan** out) {
    if (!out) return TIPK_EINVAL;
    *out = nullptr;
    if (n_edges < 0 || (n_edges > 0 && (!src || !dst)) || n_nodes <= 0) return TIPK_EINVAL;
    for (int64_t e = 0; e < n_edges; ++e)
        if (src[e] < 0 || src[e] >= n_nodes || dst[e] < 0 || dst[e] >= n_nodes) return TIPK_EINVAL;
    tipk_host_plan* h = new (std::nothrow) tipk_host_plan;
    if (!h) return TIPK_EINVAL;
    std::vector<uint32_t> w;
    tipk_plan::pair_link_words(src, dst, n_edges, n_nodes, w);
    h->put("links", w);
    *out = h;
    return TIPK_OK;
}

extern "C" int tipk_host_plan_array(const tipk_host_plan* h, const char* name, const void** data, int64_t* count, int* elem_bytes) {
    if (!h || !name) return TIPK_EINVAL;
    const auto it = h->arrays.find(name);
    if (it == h->arrays.end()) return TIPK_EINVAL;
    const int eb = h->elem.at(name);
    if (data) *data = it->second.data();
    if (count) *count = (int64_t)(it->second.size() / (size_t)eb);
    if (elem_bytes) *elem_bytes = eb;
    return TIPK_OK;
}

extern "C" int64_t tipk_host_plan_scalar(const tipk_host_plan* h, const char* name) {
    if (!h || !name) return -1;
    const auto it = h->scalars.find(name);
    return it == h->scalars.end() ? -1 : it->second;
}

extern "C" void tipk_host_plan_free(tipk_host_plan* h) { delete h; }
