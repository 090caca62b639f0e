// Host-side plan construction of the pair-form D-D passes, in C++ (internal header of libtipk; the op-level graph handle of
// tipk_graph.hip builds its plans with it).  The same arrays, bit for bit, as tip_amd/plan.py `build_stream_plan_rows` /
// `build_pair_bwd_plan` and tip_amd/layers.py `pair_link_words` produce -- tests/test_host_plans.py compares them on the CPU.
#pragma once
#include <stdint.h>
#include <vector>

namespace tipk_plan {

// arrays of tipk_stream_gather (include/tipk.h section 1d)
struct StreamPlanH {
    int64_t n_rows = 0, n_table = 0, n_bands = 0, n_edges = 0;
    int n_wg = 0, lanes = 0, piece = 0, idx_unit = 1, row_bytes = 0;
    std::vector<int32_t> wave_ptr;          // [n_wg * 16 + 1]
    std::vector<int32_t> cells;             // [n_bands][64 / lanes] (uint32 bit patterns)
    std::vector<uint16_t> ids;              // [max(n_bands, 1) * piece * (64 / lanes) * 8]
    std::vector<int32_t> zero_ptr;          // [n_wg * 16 + 1]
    std::vector<int32_t> zero_rows;         // rows without edges ([0] when there are none)
};

// arrays of tipk_rgcn_pair_grads + tipk_stream_gather_parts (include/tipk.h section 2e)
struct PairBwdH {
    int64_t n_nodes = 0, n_rel = 0, n_slots = 0, n_parts = 0, part_len = 0, n_alloc = 0;
    bool symmetric = false;
    std::vector<int32_t> slots;             // [n_slots][4] {v, bits of 1 / deg(v), cell line, row of pg}
    std::vector<int32_t> node_desc;         // [n_nodes][4] {u, first slot, tiles, 0}
    std::vector<int32_t> tile_node;         // node of every tile of 32 slots
    std::vector<int32_t> part_first;        // [n_parts]
    std::vector<int32_t> wg_part;           // [gather.n_wg]
    StreamPlanH gather;
};

// arrays of tipk_gather_sum on a GROUPED plan (include/tipk.h section 1; tip_amd/plan.py `build_gather_plan` with group_slots > 0)
struct GatherPlanH {
    int64_t n_out = 0, n_table = 0, n_edges = 0, n_items = 0;
    int chunk = 0, group_slots = 0;
    std::vector<int32_t> row_id;            // [n_edges] table row per edge, edges sorted by output row (stable)
    std::vector<float> edge_w;              // [n_edges] in the same order (empty: no weights)
    std::vector<int32_t> items;             // [n_items][4] {begin, end, target, flags}
    std::vector<int64_t> perm;              // plan order -> caller's edge order
};

constexpr int STREAM_WIDE_STEPS = 16;
constexpr int PAIR_PART_ROWS = 1016, PAIR_PART_WGS = 4, PAIR_PART_EDGES_PER_WG = 16384;

// out[o] = sum over the edges with out_row[e] == o of table[tab_row[e]]; row_bytes 0 = lanes * 16
void build_stream_plan_rows(const int64_t* out_row, const int64_t* tab_row, int64_t n_edges, int64_t n_rows, int64_t n_table, int n_wg,
                            int lanes, int piece, int wide_steps, int row_bytes, StreamPlanH& sp);

// plan of the pair-form backward pass (scale [n_nodes] = 1 / in-degree as the layer applies it); returns false when the
// graph was declared symmetric and is not
bool build_pair_bwd_plan(const int64_t* src, const int64_t* dst, const int64_t* rel, int64_t n_edges, int64_t n_nodes, int64_t n_rel,
                         const float* scale, bool symmetric, int n_wg, int lanes, int piece, PairBwdH& pb);

// out[o] = sum over the edges with out_row[e] == o of edge_w[e] * table[table_row[e]]; chunk 0 = chosen from the edge count;
// group_slots = the block size G (split rows are combined inside one workgroup): `group_slots_for(d)`
void build_gather_plan(const int64_t* out_row, const int64_t* table_row, const float* edge_w /* nullable */, int64_t n_edges, int64_t n_out,
                       int64_t n_table, int chunk, int group_slots, GatherPlanH& gp);
int group_slots_for(int d);

// uint32 [n_nodes padded to 8][ceil(n_nodes / 32)]: bit r of word (u, t) = some edge links u -> 32 t + r
void pair_link_words(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, std::vector<uint32_t>& words);

// every relation links u -> v as often as v -> u
bool relations_symmetric(const int64_t* src, const int64_t* dst, const int64_t* rel, int64_t n_edges, int64_t n_nodes);

}  // namespace tipk_plan
