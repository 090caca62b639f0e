/* tipk.h -- C ABI of libtipk.so: the MI355X (gfx950) kernels behind the TIP hot path.
 *
 * The reference (NYXFLOWER/TIP) has no FFI/plugin layer for this path: its boundary is the Python
 * `nn.Module` surface of `src/layers.py`, and all device arithmetic is delegated to torch, PyG 2.0.1
 * and torch-scatter 2.0.8.  Each entry point below therefore names the reference call site whose
 * implicit torch/PyG kernels it replaces (paths relative to the reference root; K-numbers are
 * SURVEY.md section 2.1).  `tip_amd/_lib.py` binds every symbol with ctypes; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked host; fp32 values, int32 indices (the Python
 *     side narrows the reference's int64 tensors once, when it builds a plan);
 *   - `stream` is a hipStream_t (NULL = the legacy default stream); calls only enqueue work: they
 *     never allocate, free, synchronise or throw, so a caller may capture them into a hipGraph;
 *   - return value: TIPK_OK, TIPK_EINVAL / TIPK_EUNSUPPORTED for bad arguments, or
 *     -(1000 + hipError_t) when a HIP runtime call failed; `tipk_strerror` names it;
 *   - workspaces (`partial`, slabs) are supplied by the caller; the library borrows pointers for the
 *     duration of the call only;
 *   - thread safety: the launch entry points are re-entrant per (device, stream); one process per GPU for multi-GPU jobs.
 *     Two pieces of state are PROCESS-WIDE (plain globals, read at launch time, no locking): the options of section 0
 *     (`tipk_set_option`) and the flag-wait budget of section 8 (`tipk_peer_set_timeout_ms`) -- set them before launching
 *     from other threads; every exchange of the process shares the one budget.
 */
#ifndef TIPK_H
#define TIPK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIPK_ABI_VERSION 23

#define TIPK_OK            0
#define TIPK_EINVAL      (-1)
#define TIPK_EUNSUPPORTED (-2)
#define TIPK_EHIP_BASE   (-1000)      /* status = TIPK_EHIP_BASE - hipError_t */

typedef void* tipk_stream_t;          /* hipStream_t */

int         tipk_abi_version(void);
const char* tipk_strerror(int status);
/* digest (16 hex digits) of the sources the library was compiled from; "+debug" appended in
 * -DTIPK_DEBUG builds.  tip_amd/_lib.py refuses a library whose digest differs from the sources
 * next to it (a stale prebuilt .so would silently run old kernels). */
const char* tipk_build_id(void);
/* 0. Host-side options (process-wide; set before launching, never read from the environment):
 *      "gemm_no_stream"      1 = every product through the LDS-tiled kernel (cross-check in tests)
 *      "gemm_thin_k_narrow"  1 = dword body of the Y = att.XB streaming kernel
 *      "gemm_stream_kk"      1 = lane-per-row streaming body for d att
 *      "rg_occupancy"        workgroups per CU tipk_rel_gather aims for: 1, 2 (0 = library default)
 *      "dm_task_kernel"      1 = fused objective through the k/4-lanes-per-position task kernel (the round-2 kernel; A/B runs)
 *      "rg_debug", "dp_debug", "dm_debug"  bit masks that SKIP parts of tipk_rel_gather / tipk_rgcn_dy_products / the decoder kernels
 *                            (timing decompositions): accepted by -DTIPK_DEBUG builds only; a release
 *                            library returns TIPK_EUNSUPPORTED for a non-zero value and its kernels
 *                            contain no skip code.
 *    Unknown name: TIPK_EINVAL. */
int         tipk_set_option(const char* name, int value);
int         tipk_get_option(const char* name, int* value);
/* host query: fills whatever is non-NULL; returns TIPK_OK or a HIP error (e.g. no device). */
int         tipk_device_info(int device, int* n_cu, int* lds_bytes_per_cu, int* wavefront, char* arch, int arch_len);

/* --------------------------------------------------------------------------------------------
 * 1. Segmented gather-sum -- the one sparse-aggregation kernel of the path.
 *
 *      out[row(w)] (or partial[slot(w)]) = sum_{e in [begin_w, end_w)} edge_w[e] * table[row_id[e]]
 *
 * replaces PyG `MessagePassing.propagate` = `index_select` + `torch_scatter.scatter` at
 *   GCNConv            src/layers.py:392,394   (K1: P-P aggregation, edge_w = gcn norm)
 *   MyHierarchyConv    src/layers.py:229-233   (K2: P->D mean)
 *   MyRGCNConv2/Conv   src/layers.py:159-180 / :78-86  (K5/K6: D-D aggregation over Y = att.XB)
 * and their autograd backward (the transposed plan).
 *
 * A *plan* (built once per static graph by tip_amd/plan.py) sorts the edges by output row and cuts
 * every row's edge list into work items of at most `chunk` edges:
 *   items[w] = { begin, end, target, flags }     int32 x 4, items ordered by decreasing length
 *     flags bit0 = 1: the row has one item; `target` is the output row and the epilogue
 *                     (row_scale, bias, relu) is applied here;
 *     flags = 0:      the row is split; `target` is a slot of `partial` ([n_slots x d], dense);
 *                     `tipk_gather_sum_finalize` adds the row's slots in order (deterministic).
 *   Plans built with group_slots = G > 0 never use `partial`: the pieces of a split row occupy
 *   consecutive items inside one aligned block of G items, padded with null items, and are added
 *   in item order through LDS by the kernel itself:
 *     flags bit1 (2): piece; flags bit2 (4): first piece of its row, (flags >> 8) pieces in all,
 *                     `target` = output row, epilogue applied; flags bit3 (8): null item.
 *   One workgroup runs one block: G * lanes-per-item threads (lanes per item = next pow2 of d/4,
 *   or of d when d % 4 != 0) must be a multiple of 64 and <= 1024 (G = 128: d <= 32).
 * d = floats per row (d % 4 == 0: 4..256, or any d <= 64); ld_* = row strides in floats
 * (multiples of 4 when d % 4 == 0; table/out/partial 16-byte aligned in that case).
 * n_table = rows of `table`: below 4 GB a gathered row is addressed as table base + 32-bit byte offset
 * (one multiply per row instead of a 64-bit multiply-add); 0 = unknown (general path).
 */
int tipk_gather_sum(const float* table, int64_t ld_table, int64_t n_table /* rows of table (0 = unknown) */,
                    const int32_t* row_id, const float* edge_w /* nullable */,
                    const int32_t* items, int64_t n_items,
                    float* out, int64_t ld_out,
                    float* partial /* nullable when no item is split */,
                    const float* row_scale /* nullable, per out row */,
                    const float* bias /* nullable, [d] */, int relu,
                    int d, int group_slots /* G of the plan, 0 = none */, tipk_stream_t stream);

/* rows[m] = { out_row, first_slot, end_slot } int32 x 3: out[out_row] = epi(sum partial[slots]).
 * max_slots: largest slot count of any row (host knows it from the plan; 0 = unknown) -- picks
 * a slot-per-row kernel for lightly split rows, a workgroup-per-row kernel otherwise. */
int tipk_gather_sum_finalize(const float* partial, const int32_t* rows, int64_t n_rows,
                             float* out, int64_t ld_out,
                             const float* row_scale, const float* bias, int relu,
                             int d, int max_slots, tipk_stream_t stream);

/* tipk_gather_sum on a GROUPED plan whose workgroups have 1024 threads (group_slots x lanes per item; `_supported`), with up
 * to 3 ordered slab sums (section 2: tipk_sum_slabs_group) that are READY at the same point riding in the launch as further
 * workgroups -- the bias-gradient partials of GCNConv 1 next to its transposed aggregation, the split-K slabs of conv2's
 * d W / d bias next to its transposed aggregation (src/layers.py:392-394 under autograd): a dependent 4-us launch less each.
 * Two optional extensions of the epilogue, for the transposed aggregation whose rows are gradients of a ReLU layer's output:
 *   gate [n_out x d] (row stride ld_gate):  out[row] = gate[row] > 0 ? value : 0  (the ReLU backward, src/layers.py:393);
 *   colsum [workgroups x d] (workgroups = ceil(n_items / group_slots), d % 4 == 0): column sums of the finished rows of every
 *   workgroup, in slot order -- the bias gradient's partial rows, added in order by a slab sum (riding in the NEXT gather). */
struct tipk_slab_sum_desc;                         /* section 2 */
int tipk_gather_sum_riders_supported(int d, int group_slots);
int tipk_gather_sum_riders(const float* table, int64_t ld_table, int64_t n_table, const int32_t* row_id, const float* edge_w,
                           const int32_t* items, int64_t n_items, float* out, int64_t ld_out, const float* row_scale,
                           const float* bias, int relu, int d, int group_slots,
                           const float* gate /* nullable */, int64_t ld_gate, float* colsum /* nullable */,
                           const struct tipk_slab_sum_desc* sums /* host, [n_sums <= 3] */, int32_t n_sums,
                           tipk_stream_t stream);

/* tipk_gather_sum on a GROUPED plan with a linear map of every finished row in the same launch:
 *     out[row]  = row_scale[row] * sum (as tipk_gather_sum, no bias / ReLU)          [n_out x d]
 *     out2[row] = relu2?( out[row] . w^T + bias2 )                                     [n_out x d2],  w element (o, i) at w[o w_so + i w_si]
 * Aggregate-then-transform of a GCN layer whose output is needed for FEW rows (conv2 of the P-P encoder: 3 640 of 19 081
 * proteins are read by the P->D stage): A_hat (x W^T) = (A_hat x) W^T, so the dense map runs on the rows that are kept
 * (src/layers.py:392-394 with the rows nobody reads left out), and the layer's forward pass is ONE launch.
 * d in {16, 32, 64} (L = d / 4 lanes hold a row), d2 in {L, 2 L, 4 L} and <= 16, group_slots > 0 (`tipk_gather_sum_lin_supported`). */
int tipk_gather_sum_lin_supported(int d, int d2, int group_slots);
int tipk_gather_sum_lin(const float* table, int64_t ld_table, int64_t n_table, const int32_t* row_id, const float* edge_w,
                        const int32_t* items, int64_t n_items, float* out, int64_t ld_out, const float* row_scale,
                        const float* w, int64_t w_so, int64_t w_si, const float* bias2 /* nullable */, int relu2,
                        float* out2, int64_t ld_out2, int d, int d2, int group_slots, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 1c. CSR rows -- the TRANSPOSED D-D pass of a large graph (autograd backward of K5/K6,
 *     src/layers.py:159-180): out[r] = sum_{e in [row_ptr[r], row_ptr[r+1])} table[row_id[e]] for every one
 *     of the n_out rows (all written, also empty ones), where rows are short (config 5: R*N = 20 M rows
 *     (relation, source), 2.5 edges each, table = g' [N x d] cache-resident).  A slot takes 8
 *     consecutive rows: one coalesced pointer load, contiguous ids, 8 gathered rows in flight, one
 *     contiguous output stream -- no work-item descriptors.  Edges sorted by output row (stable);
 *     row_ptr int32 [n_out + 1]; d % 4 == 0, 8 <= d <= 256. */
int tipk_gather_rows_csr(const float* table, int64_t ld_table, int64_t n_table /* rows; table < 4 GB */,
                         const int32_t* row_ptr, const int32_t* row_id,
                         int64_t n_out, float* out, int64_t ld_out, int d, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 1b. Relation-local gather -- the D-D aggregation (K5/K6, src/layers.py:159-180) when one
 *     relation's node table fits in LDS (BioSNAP: 645 drugs x 32 floats).  Same sums as
 *     `tipk_gather_sum` over the plan of the same graph, with every gathered row read from LDS:
 *
 *       backward = 0:  table = Y [n_rel * n_nodes, d];  out = partial [n_wg, n_nodes, d] with
 *                      sum_wg partial[wg, o] = sum_r sum_{e in r: out(e)=o} Y[r * n_nodes + tab(e)]
 *                      (combine with tipk_sum_slabs);
 *       backward = 1:  table = g' [n_nodes, d];  out = dY [n_rel * n_nodes, d],
 *                      dY[r * n_nodes + o] = sum_{e in r: out(e)=o} g'[tab(e)]   (every row written).
 *
 *     d: power of two >= 4; n_nodes <= 65535; the columns are cut into blocks until a block's table
 *     (+ accumulators when backward = 0) fits in 158 KB of LDS -- `tipk_rel_gather_supported`.
 *
 *     Relation-local plan (tip_amd/plan.py `build_rel_plan`), all device arrays.  The plan is a
 *     list of n_units WORK UNITS: a unit is one relation, or -- for relations much larger than the
 *     per-workgroup average, which would otherwise set the length of the launch -- every k-th output
 *     position of one relation:
 *       node_at[n_units][n_nodes] uint16: output node at position p of the unit; positions are
 *                               ordered by decreasing run length
 *       runs[n_units][n_nodes][2] (begin relative to the unit's first id, padded length) per position
 *       idx[..]                 uint16 table node of each edge TIMES idx_unit (16-byte aligned array;
 *                               idx_unit = 1, or a power of two up to the bytes of one column-block
 *                               row with n_nodes * idx_unit <= 65535: then a row's LDS address is
 *                               base + idx with no multiply); inside a unit the edges are sorted by
 *                               the position of their OUTPUT node; runs are padded to multiples of 8
 *                               ids with the sentinel n_nodes * idx_unit, whose table row is zero.
 *                               The order of the ids INSIDE a run is free (it only fixes the order of
 *                               the fp32 sum): tip_amd/plan.py orders them so that the slots of one
 *                               ds_read_b128 lane group read different bank quarters (table rows are
 *                               unpadded: node mod (256 / row bytes) is the quarter)
 *       unit_meta[n_units][8]   int32 descriptors, listed in the order the workgroups process them:
 *                               { unit (row of node_at / runs), relation (selects the Y_r block / the
 *                               dY rows), n_pos (positions to walk: backward all of the unit's, so
 *                               every dY row is written exactly once; forward those with edges),
 *                               n_ids (padded), idx offset low, idx offset high (multiple of 8), 0, 0 }
 *       wg_unit_ptr[n_wg+1]     range of unit_meta handled by each of the n_wg workgroups
 *                               (longest-processing-time deal; n_wg = number of CUs)
 */
/* host predicate: number of column blocks the launch will use (grid = n_wg x blocks), 0 = the
 * shape is not supported (use tipk_gather_sum).  row_scale (backward only, nullable): the table
 * rows are multiplied by row_scale[node] while they are staged (g' = g / deg fused). */
int tipk_rel_gather_supported(int64_t n_nodes, int d, int backward);
/* workgroups per CU (1 or 2) the launch of this shape reaches: with 2 the columns are cut finer so that
 * two 1024-thread workgroups share a CU (8 waves per SIMD hide the kernel's LDS round trips); the host
 * builds the plan for  occupancy * CUs / column-blocks  workgroups.  Option "rg_occupancy" (1 | 2). */
int tipk_rel_gather_occupancy(int64_t n_nodes, int d, int backward);
/* ids staged into LDS per pass for this shape (8192 or 16384): a forward work unit with more ids than
 * this reloads its id chunk synchronously, so the host cuts forward units at this size. */
int tipk_rel_gather_chunk(int64_t n_nodes, int d, int backward);
int tipk_rel_gather(int backward, const float* table, int64_t ld_table, int64_t n_nodes, int d,
                    int64_t n_wg, const int32_t* wg_unit_ptr, const int32_t* unit_meta,
                    const uint16_t* idx, int idx_unit, const int32_t* runs, const uint16_t* node_at,
                    const float* row_scale, float* out, int64_t ld_out, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 1d. Wave-stream gather:   out[row] = sum_{e: row(e) = row} table[tab(e)]   for a table [n_table x d] that fits in LDS
 *     (one column block of it: `tipk_stream_gather_supported`).  After the table is staged (one barrier) every
 *     wavefront streams through its own list of fixed-size records and never meets the others
 *     (tip_amd/csrc/tipk_rel_stream.hip explains why).  Two uses on the TIP path:
 *       - the TRANSPOSED D-D pass (autograd of src/layers.py:159-180): table = g' [nodes x out], row = relation *
 *         nodes + source node -- the sums of tipk_rel_gather(backward = 1), without its work units;
 *       - the FORWARD D-D pass (src/layers.py:159-180) in pair form: table = att [relations x bases], row =
 *         destination * nodes + source: cell (v, u) = sum of att[r, :] over the relations r that link u -> v
 *         (a drug pair of BioSNAP is linked by 66 relations on average), followed by ONE dense product with
 *         X . basis (tip_amd/ops.py `_RGCN.forward`): Y = att . XB is never formed.
 *       - (round 3) the GCN layers of the P-P graph, both passes (GCNConv.forward via PPEncoder, src/layers.py:392-394):
 *         D^-1/2 (A + I) D^-1/2 X = a row scaling on either side of the plain sum, i.e. row_scale = out_scale =
 *         deg^-1/2, with bias and ReLU in the epilogue.  19 081 proteins fit as 2-column blocks of 8-byte rows
 *         (max_split = 16): the rows come out of LDS instead of through the L2 gather path of tipk_gather_sum.
 *     A lane holds 4 columns (2 when the column block is 2 wide); L = (d / column blocks) / 4 lanes (or 1) form a slot, S = 64 / L slots a wavefront, P = tipk_stream_gather_piece()
 *     steps of 8 ids a cell.  Plan (tip_amd/plan.py `build_stream_plan_rows`), all device arrays:
 *       wave_ptr[n_wg * 16 + 1]  int32: range of bands of every wavefront (wavefront = workgroup * 16 + wave)
 *       cells[n_bands][S]        uint32: row (24 bits) | steps << 24 (0 .. P) | first << 28 | last << 29 | log2 k << 30.
 *                                A slot adds the steps' rows to a register sum that is cleared on `first` and
 *                                written to out[row] on `last`: a run longer than P steps continues in the SAME
 *                                slot of the wavefront's next band; 0 = idle cell.  log2 k > 0 (on the last band
 *                                of a WIDE run, in all its slots): the run was cut into k = 2, 4 or 8 sub-runs in
 *                                adjacent slots s0 .. s0 + k - 1 (s0 a multiple of k); their sums are added in
 *                                the fixed order  s <- s + 2^j,  j = 0, 1, ..,  and slot s0 (the one with `last`)
 *                                writes the row -- a slot's run is a chain of dependent steps, and the longest
 *                                run of the graph would otherwise set the length of the launch
 *       ids[n_bands][P][S][8]    uint16: table row * idx_unit of the edges (see 1b), runs padded to 8 with the
 *                                sentinel n_table * idx_unit (a zero row); steps beyond a cell's count are not read
 *       zero_ptr[n_wg * 16 + 1], zero_rows[]  int32: the rows without edges, dealt to the wavefronts; zero_ptr
 *                                = NULL: those rows are left untouched (the consumer masks them -- section 2b
 *                                row_used -- or they live in a buffer that was zeroed once)
 *     rows < 2^24.  row_scale (nullable): row_scale[i] * table[i] is what is staged.
 *     Epilogue of a finished row (rows without edges included): relu?(out_scale[row] * sum + bias[col]); out_scale /
 *     bias nullable.  max_split: the largest number of column blocks the caller accepts (every block walks all the ids
 *     again): 4 on the D-D passes, 16 for the P-P graph (which also admits 2-column blocks).
 */
/* column blocks of the launch (grid = n_wg x blocks); 0 = the table does not fit in LDS */
int tipk_stream_gather_supported(int64_t n_table, int d, int max_split);
/* two tables of the same shape on ONE plan in one launch (no zero rows, no scaling, no epilogue): out0 <- table0, out1 <- table1;
 * the pair cells of both R-GCN layers of an encoder (they share the graph; the cells depend on the parameters only).  One column
 * block only (tipk_stream_gather_supported(n_table, d, 1) == 1), d in {16, 32, 64}. */
int tipk_stream_gather_two(const float* table0, const float* table1, int64_t ld_table, int64_t n_table, int d, int64_t n_wg,
                           const int32_t* wave_ptr, const uint32_t* cells, const uint16_t* ids, int idx_unit,
                           float* out0, float* out1, int64_t ld_out, tipk_stream_t stream);
int tipk_stream_gather_piece(void);
int tipk_stream_gather(const float* table, int64_t ld_table, int64_t n_table, int d, int64_t n_wg,
                        const int32_t* wave_ptr, const uint32_t* cells, const uint16_t* ids, int idx_unit,
                        const int32_t* zero_ptr, const int32_t* zero_rows, const float* row_scale,
                        float* out, int64_t ld_out, int kind /* 0 | 1: names the kernel instance in profiles, nothing else */,
                        int max_split, const float* out_scale, const float* bias, int relu, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2. Dense fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
 *
 *   C[z] = relu?( alpha * sum_{q<kbatch} A[z,q] (m x k) . B[z,q] (k x n) + C_in[z] ),  z < batch
 *
 * With ksplit > 1 the k range is cut into `ksplit` slabs: slab s holds the partial product over
 * its k range at c + s*c_ss (c_in/relu must be unset); `tipk_sum_slabs` combines them in order.
 *
 * replaces the `torch.matmul`/`mm` calls of the path:
 *   W = att @ basis.view(B,-1), x_j[s:e] @ w[et], x @ root   src/layers.py:163-172,184  (K4,K5,K7)
 *   aggr_out[n_src:] @ weight                                 src/layers.py:239          (K2)
 *   GCNConv.lin                                               src/layers.py:392,394      (K1)
 * and their backward products.  Element (i,j) of A[z,q] is a[z*a_sz + q*a_sq + i*a_sm + j*a_sk]
 * (strides in floats, so transposes and the basis-decomposition reshapes need no copies).
 */
typedef struct tipk_gemm_desc {
    int64_t m, n, k;
    int64_t batch, kbatch;
    int64_t ksplit;                                /* >= 1: split K into this many slabs */
    const float* a; int64_t a_sm, a_sk, a_sq, a_sz;
    const float* b; int64_t b_sk, b_sn, b_sq, b_sz;
    float*       c; int64_t c_sm, c_sz, c_ss;      /* C row-major (column stride 1); c_ss = slab stride */
    const float* c_in; int64_t cin_sm, cin_sz;     /* nullable; may alias c */
    float alpha;
    int   relu;
} tipk_gemm_desc;

int tipk_gemm_f32(const tipk_gemm_desc* desc /* host */, tipk_stream_t stream);

/* Up to TIPK_GROUP_MAX independent products in ONE launch (no ordering between them; outputs must
 * not overlap any input of the group).  Each product is computed exactly as tipk_gemm_f32 would
 * (same tile shape and k order -> bit-identical).  Used where the path needs several small
 * products at the same point of the dependent chain: XB and X root going forward
 * (src/layers.py:163-172 and :184), d basis / d root / both halves of dX going back. */
#define TIPK_GROUP_MAX 6
int tipk_gemm_f32_group(const tipk_gemm_desc* descs /* host, [count] */, int32_t count, tipk_stream_t stream);

/* out[i] = alpha * sum_{s<n_slabs} in[s*slab_stride + i] (+ out[i] if accumulate), i < count.
 * Ordered (deterministic) reduction of split-K slabs / per-workgroup partials. */
int tipk_sum_slabs(const float* in, int64_t n_slabs, int64_t slab_stride, int64_t count,
                   float alpha, int accumulate, float* out, tipk_stream_t stream);
/* same with a fused epilogue: out[i] = relu?( alpha * row_scale[i / cols] * sum_s in[s][i] + addend[i]
 * (+ out[i]) ); row_scale / addend nullable.  Finishes an R-GCN layer in one pass:
 * relu( D^-1 sum_partials + X root )  (src/layers.py:184, :547). */
int tipk_sum_slabs_ex(const float* in, int64_t n_slabs, int64_t slab_stride, int64_t count,
                      float alpha, int accumulate, const float* row_scale, int64_t cols,
                      const float* addend, int relu, float* out, tipk_stream_t stream);
/* the same for up to TIPK_GROUP_MAX independent slab sets in one launch (arguments as above);
 * gate (nullable, [count]): out[i] = gate[i] > 0 ? value : 0 -- the ReLU backward of the layer that
 * produced the gradient's input (src/layers.py:547), applied while the gradient is finished. */
typedef struct tipk_slab_sum_desc {
    const float* in; int64_t n_slabs, slab_stride, count;
    float alpha; int accumulate;
    const float* row_scale; int64_t cols;
    const float* addend; int relu;
    const float* gate;
    float* out;
} tipk_slab_sum_desc;
int tipk_sum_slabs_group(const tipk_slab_sum_desc* descs /* host, [count] */, int32_t count, tipk_stream_t stream);

/* Products whose reduction is split over the 16 waves of ONE workgroup, plus ordered slab sums that are ready at
 * the same point, in ONE launch.  The backward pass of an R-GCN layer (autograd of src/layers.py:159-184) ends in
 * d basis[b] = X^T dXB[b], d root = X^T g and dX = sum_b dXB[b] basis[b]^T + g root^T: few output tiles, reductions of
 * 645 .. 1 056 terms.  A workgroup owns a 32 x 32 output tile, deals its K tiles (32 terms; the kbatch terms of p and the
 * optional second product a2 . b2 are further tiles) to its waves in contiguous runs and adds the partial tiles in
 * wave order (deterministic) -- no slabs, no second launch.
 *   p:     as tipk_gemm_f32 (batch, kbatch, strides, c_in, alpha, relu); ksplit must be 1
 *   a2/b2: nullable second product [m x k2] . [k2 x n] added to the same output (batch must be 1)
 *   gate:  nullable, laid out like the output: out = gate > 0 ? value : 0
 * Shapes: at most 64 K tiles in all (2 048 terms), at most 4096 output tiles (TIPK_EUNSUPPORTED otherwise: tipk_gemm_f32_group). */
typedef struct tipk_wg_gemm_desc {
    tipk_gemm_desc p;
    const float* a2; int64_t a2_sm, a2_sk;
    const float* b2; int64_t b2_sk, b2_sn;
    int64_t k2;
    const float* gate; int64_t gate_sm, gate_sz;
} tipk_wg_gemm_desc;
#define TIPK_WG_GEMM_MAX 4
#define TIPK_WG_SUMS_MAX 3
int tipk_gemm_wg_group_supported(const tipk_wg_gemm_desc* desc /* host */);
int tipk_gemm_wg_group(const tipk_wg_gemm_desc* descs /* host, [count <= TIPK_WG_GEMM_MAX] */, int32_t count,
                       const tipk_slab_sum_desc* sums /* host, [n_sums <= TIPK_WG_SUMS_MAX], nullable */, int32_t n_sums,
                       tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2c. The dense half of the pair-form D-D forward pass (src/layers.py:159-180; tip_amd/ops.py `_RGCN.forward`):
 *
 *        slabs[g][v, c] = sum_{u in group g} sum_b cells[u][v][b] * xb[u][b][c]        g < n_src / group
 *
 *     cells [n_src][n_dst][n_bases] = the pair cells written by tipk_stream_gather (table = att), xb
 *     [n_src][n_bases][32] = X . basis per source node, rows PADDED to 32 columns with zeros beyond d; n_src is the node count rounded up to a multiple of
 *     `group` with blocks that stay zero.  slabs [n_src / group][n_dst][d] are added in order by
 *     tipk_sum_slabs_ex (which also applies 1 / deg, + X root and the ReLU).  n_bases in {8, 16, 32}, d <= 32
 *     (`tipk_pair_product_supported`; otherwise tipk_gemm_f32 with kbatch = group does the same sums).
 *     symmetric != 0: every relation links u -> v iff it links v -> u (BioSNAP), so cells[u][v] == cells[v][u]
 *     and the gather only built the cells with u <= v (half the edges); cell (u, v) with v < u is read at (v, u).
 *     links (nullable): uint32 [n_src][ceil(n_dst / 32)], bit r of word (u, t) = "pair (u, 32 t + r) is linked" (rows of the
 *     padding nodes zero; on a symmetric graph bit (u, v) == bit (v, u)): the cell of an unlinked pair is not fetched (it is
 *     zero by construction; `zeros` = >= n_bases * 4 bytes of zeros (128 B at n_bases = 32), 16-byte aligned, read in its place).
 *     xbt (nullable, [n_dst][d][n_bases], 16-byte aligned): XB of the first n_dst source nodes written back with the bases
 *     innermost -- the operand layout of the backward pass (tipk_rgcn_node_products xbt), from the LDS stage of this kernel.
 */
int tipk_pair_product_supported(int n_bases, int d);
int tipk_pair_product(const float* cells, const float* xb, int64_t n_src, int64_t n_dst, int n_bases, int d,
                      int group, int symmetric, const uint32_t* links, const float* zeros, float* xbt /* nullable */,
                      float* slabs, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2b. Both consumers of dY (the gradient of Y = att . XB, src/layers.py:163-172 under autograd) in
 *     one pass over dY [n_rel x n_cols] (n_cols = nodes * out channels; 91 MB at BioSNAP layer 1):
 *
 *        dXB [b, c] = sum_r att[r, b] * dY[r, c]       datt[r, b] = sum_c dY[r, c] * XB[b, c]
 *
 *     Results arrive as slabs to be added in order (tipk_sum_slabs_group):
 *        dxb_slabs  [row_slabs][n_bases x n_cols]   one per range of relations
 *        datt_slabs [col_slabs][n_rel x n_bases]    one per chunk of 512 columns
 *     `tipk_rgcn_dy_products_plan` returns the slab counts (both 0: shape not supported -- n_bases > 32;
 *     use two tipk_gemm_f32 then).  The datt slabs are n_bases / 512 of the size of dY.
 *     row_used (nullable; then n_nodes = n_cols / columns per node): uint32 [ceil(n_rel / 32)][n_nodes], bit
 *     (r & 31) of word (r >> 5, node) = "relation r has an edge leaving node" = row (r, node) of dY holds data.
 *     More than half of the (relation, source) rows of BioSNAP have no edge: with the mask the transposed
 *     gather (section 1d, zero_ptr = NULL) does not write their zeros -- 48 of 91 MB at layer 1 -- and
 *     whatever the buffer holds there is cleared bitwise after it is loaded.
 */
int tipk_rgcn_dy_products_plan(int64_t n_rel, int64_t n_cols, int n_bases, int* col_slabs, int* row_slabs);
int tipk_rgcn_dy_products(const float* dy, int64_t ld_dy, const float* att, int64_t ld_att,
                          const float* xb, int64_t ld_xb, int64_t n_rel, int64_t n_cols, int n_bases,
                          const uint32_t* row_used, int64_t n_nodes,
                          float* dxb_slabs, float* datt_slabs, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2d. The same two products as 2b on the COMPACT, node-major form of dY (autograd of src/layers.py:163-172 and of
 *     the scatter-mean at :159-180, for graphs whose transposed pass runs as tipk_stream_gather):
 *
 *        dXB [b, u, c] = sum_{r leaves u} att[r, b] * dY[(u, r), c]      datt[r, b] = sum_u sum_c dY[(u, r), c] * XB[b, u, c]
 *
 *     dyc [n_rows + 1][d]: one row per (source node u, relation r) pair that has an edge (BioSNAP: 47 % of the
 *       R x N pairs), grouped by node, ascending relation inside a node -- the row numbers are what the plan of the
 *       transposed gather (section 1d) writes into its cells, so the gather produces this layout directly; row
 *       n_rows must hold zeros (pairs without an edge read it).
 *     node_desc [n_nodes][4]  { node u, first row, end row, 0 } of the nodes by DECREASING row count (launch order of
 *       the per-node workgroups; 16-byte aligned);   row_rel [n_rows]  relation of a row;
 *     pos [n_nodes][ceil(n_rel / 64) * 64]  row of (u, r), or n_rows when the pair has no edge / r >= n_rel.
 *     xb element (b, u, c) at xb[b * xb_sb + u * xb_su + c] (strides multiples of 4 floats), dxb likewise: written
 *     COMPLETE (no slabs); datt_slabs [att_slabs][n_rel][n_bases] are added in order by tipk_sum_slabs(_group).
 *     xbt (optional): the same values as [n_nodes][d][n_bases] (contiguous) -- the d att product then reads a column of
 *     all bases as ONE 128-byte line instead of 32 lines (the vector-memory address path bounds that product).
 *     d in {16, 32, 64, 128}, n_bases <= 32 (`tipk_rgcn_node_products_plan` returns att_slabs = 0 otherwise: use the
 *     dense form 2b).  All sums in fixed order: bitwise reproducible.
 */
int tipk_rgcn_node_products_plan(int64_t n_nodes, int d, int64_t n_rel, int n_bases, int* att_slabs);
int tipk_rgcn_node_products(const float* dyc, int64_t n_rows, int d, const int32_t* node_desc, const int32_t* row_rel,
                            const int32_t* pos, int64_t n_nodes, int64_t n_rel,
                            const float* att, int64_t ld_att, int n_bases,
                            const float* xb, int64_t xb_sb, int64_t xb_su, const float* xbt /* nullable */,
                            float* dxb, int64_t dxb_sb, int64_t dxb_su, float* datt_slabs, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2e. The BACKWARD pass of the pair form 2c (autograd of src/layers.py:159-180 for graphs whose forward pass ran as
 *     cells + tipk_pair_product): with g' = g / deg and the cells C the forward pass left in place,
 *
 *        dxb[b, u, c]     = sum_{v linked to u} C[u, v, b] * g'[v, c]
 *        pg [slot(u, v)]  = XB[u] . g'[v]            one row of n_bases floats per LINKED (u, v): d C[u, v, :]
 *        datt[r, :]       = sum over the pairs (u, v) that relation r links of pg[slot(u, v)]
 *
 *     -- no row of dY = A_r^T g' is ever formed (sections 2b / 2d), nothing is per relation until the last line, and on a
 *     symmetric graph that line walks HALF the edges: datt[r] = sum_{u <= v} (pg[slot(u, v)] + pg[slot(v, u)]).
 *     tipk_rgcn_pair_grads (the first two lines), plan arrays of tip_amd/plan.py `build_pair_bwd_plan`:
 *       slots [n_slots][4]  { v, float bits of 1 / deg(v), cell line of (u, v), row of pg }: the neighbours of a node in runs
 *          of 32 (n_slots % 32 == 0; pads carry a valid v / line, the factor 0.0f and a dump row, so they add zeros -- g must
 *          be finite); cell line L = the n_bases floats at cells + L * n_bases (a symmetric forward pass keeps (min, max) only);
 *       node_desc [n_nodes][4]  { node u, its first slot, its tiles of 32 slots, 0 } by DECREASING tile count;
 *       tile_node [n_slots / 32]  the node of every tile (the pair-gradient rows are computed one tile per wavefront);
 *       cells: n_lines lines of n_bases floats; xb [n_nodes][n_bases][32] as in 2c; g [n_nodes][ld_g]; dxb element
 *       (b, u, c) at dxb[b * dxb_sb + u * dxb_su + c], written COMPLETE; pg [pg_rows][n_bases]: the gradient row of a slot
 *       is written to row slots[..][3] -- the place the gather below stages it from, so pg needs no index on the way out.
 *     n_bases = 32, d in {16, 32} (`tipk_rgcn_pair_grads_supported`).  All sums in fixed order: bitwise reproducible.
 *     tipk_stream_gather_parts (the last line) is tipk_stream_gather (1d) with a table PER WORKGROUP: the pairs (sorted) are
 *     cut into partitions of at most part_len consecutive rows that fit in LDS, row i of partition p = table[part_first[p] + i]
 *     + table[second + part_first[p] + i] (table = pg: first half = the rows of (u, v), u <= v, second half = the rows of
 *     their mirrors; rows nobody writes hold zeros), wg_part [n_wg] = the partition a workgroup stages, ids = rows counted
 *     from the partition's first, output rows = p * n_rel + r (out [n_parts * n_rel][d], added over p in order by
 *     tipk_sum_slabs(_group)).  d = 32 (one 128-byte row per pair); every partition is staged part_len rows long.
 */
int tipk_rgcn_pair_grads_supported(int n_bases, int d);
int tipk_rgcn_pair_grads(const float* cells, int64_t n_lines, const float* xb, const float* g, int64_t ld_g,
                         int64_t n_nodes, int n_bases, int d, const int32_t* node_desc, const int32_t* slots,
                         const int32_t* tile_node, int64_t n_slots, float* dxb, int64_t dxb_sb, int64_t dxb_su, float* pg, int64_t pg_rows,
                         tipk_stream_t stream);
int tipk_stream_gather_parts(const float* table, int64_t ld_table, int d, int64_t second, const int32_t* part_first,
                             int64_t part_len, const int32_t* wg_part, int64_t n_wg, const int32_t* wave_ptr,
                             const uint32_t* cells, const uint16_t* ids, int idx_unit, const int32_t* zero_ptr /* nullable */,
                             const int32_t* zero_rows, float* out, int64_t ld_out, tipk_stream_t stream);
/* Round 6: the d att gathers of BOTH R-GCN layers of an encoder in one launch (the layers share the plan): grid.y = 1
 * stages its partitions from table1 and writes out1; everything else as tipk_stream_gather_parts. */
int tipk_stream_gather_parts_two(const float* table0, const float* table1, int64_t ld_table, int d, int64_t second,
                                 const int32_t* part_first, int64_t part_len, const int32_t* wg_part, int64_t n_wg,
                                 const int32_t* wave_ptr, const uint32_t* cells, const uint16_t* ids, int idx_unit,
                                 const int32_t* zero_ptr, const int32_t* zero_rows, float* out0, float* out1,
                                 int64_t ld_out, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2f. Forward aggregation of an R-GCN layer on LARGE node sets without Y = att . XB (src/layers.py:159-180 with
 *     W_r = sum_b att[r, b] basis_b):   agg[v] = sum_b T[b, v, :] basis_b,
 *
 *        T[b, v, i] = sum_{e -> v} att[r_e, b] * x[src_e, i]          one matrix product per DESTINATION, K = its edges
 *
 *     -- 164 MB instead of the 10 GB of Y at config 5 (N = 10 000, R = 2 000, d = 128); the second product is a
 *     batch-reduced tipk_gemm_f32.  edges [E]: uint32 rel | src << bits, grouped by destination (bits = the return value of
 *     `tipk_rgcn_dest_products_supported`, 0 = shape not supported: n_bases > 32, or ids that do not fit 32 bits);
 *     node_desc [n_nodes][4] { node v, its first edge, its edges, 0 } by DECREASING edge count (16-byte aligned);
 *     T element (b, v, i) at t[b * t_sb + v * t_sv + i], every element written.  Sums in edge order per wave, the four
 *     waves of a node in wave order: bitwise reproducible.
 */
int tipk_rgcn_dest_products_supported(int64_t n_nodes, int64_t n_rel, int n_bases, int d_in);
int tipk_rgcn_dest_products(const float* x, int64_t ld_x, int d_in, const float* att, int64_t ld_att, int n_bases,
                            int64_t n_nodes, int64_t n_rel, const int32_t* node_desc, const uint32_t* edges,
                            float* t, int64_t t_sb, int64_t t_sv, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2g. The hand-over between two R-GCN layers in one launch (src/layers.py:545-548): the ordered slab sum that ends a
 *     layer's forward pass (the 16 slab lanes and the order of additions of tipk_sum_slabs_ex, epilogue relu?(row_scale * sum +
 *     addend)) -> x [n_rows][32], and at once the next layer's row-local products: xb [row][n_bases][32] = x basis (node-major,
 *     rows padded to 32 columns: the operand of 2c; columns >= d_out are not written) and xroot [n_rows][d_out] = x root.
 *     slabs [n_slabs][n_rows][32] (slab_stride floats apart); basis [n_bases][32][d_out], root [32][d_out], contiguous.
 *     d_in = 32 only (TIPK_EUNSUPPORTED otherwise: tipk_sum_slabs_ex + tipk_gemm_f32_group do the same).
 */
int tipk_sum_slabs_xb(const float* slabs, int64_t n_slabs, int64_t slab_stride, int64_t n_rows, int d_in,
                      const float* row_scale /* nullable */, const float* addend /* nullable */, int relu, float* x,
                      const float* basis, const float* root, int n_bases, int d_out, float* xb, float* xroot,
                      tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 2h. LARGE node sets, both passes: the (relation, node) row sums are assembled in LDS and multiplied there -- neither
 *     Y = att . XB nor dY = A_r^T g' exists in memory (src/layers.py:159-180 and its autograd):
 *
 *        S[(r, v), :] = sum_{e in (r, v)} table[other_e, :]
 *        t[b, v, :]   = sum_r att[r, b] S[(r, v), :]               forward: table = X, rows by destination (agg = sum_b t_b basis_b);
 *                                                                   backward: table = D^-1 g', rows by source (t = d XB)
 *        d att[r, b]  = sum_v < S[(r, v), :], xb[b, v, :] >         only with xb != NULL: slab s of datt_slabs [n_slabs][n_rel][n_bases]
 *                                                                   holds the partial sum of 8 nodes x 32 channels (sum the slabs)
 *
 *     entries [n_batches][2][16] int32: per (node, tile of 32 relations) batches of 16 words per HALF (half = rel & 1),
 *     a half's list sorted by relation, word = inside << 24 | other << 8 | 4 * (rel % 32) with inside = 0 at the first word
 *     of a (relation, node) row and 1 at its other words, padding word 128 (only at the end of a list); a node's batches are
 *     consecutive, tile after tile; desc [n_nodes][ceil(n_rel / 32)][2] = { first batch, batches >= 1 }; `entries` ends with 8
 *     batches of padding behind the last node's (the load pipeline runs 8 batches ahead of the sums).
 *     channels % 32 == 0, n_bases <= 32, n_nodes <= 65 536 (`_supported`), ld_table % 64 == 0 and < 16 384;
 *     t [n_bases][n_nodes * channels], every element written; n_slabs = `tipk_rgcn_row_products_slabs`.  Sums in entry
 *     order: bitwise reproducible.
 */
int tipk_rgcn_row_products_supported(int64_t n_nodes, int64_t n_rel, int n_bases, int channels);
int64_t tipk_rgcn_row_products_slabs(int64_t n_nodes, int channels);
int tipk_rgcn_row_products(const float* table, int64_t ld_table, int64_t n_nodes, int channels, const float* att,
                           int64_t ld_att, int64_t n_rel, int n_bases, const int32_t* entries, const int32_t* desc,
                           const float* xb /* nullable */, int64_t ld_xb, float* t, float* datt_slabs /* nullable */,
                           tipk_stream_t stream);

/* The same two products with WAVE-UNIFORM entries (channels % 64 == 0; any node count whose table fits 2 GB): a wave =
 * (node, 64 channels), one entry per step for the whole wave, so the entry's table offset, LDS row offset and inside-row flag
 * are scalar operands (1 VALU instruction per entry; the fp32 MFMA and the VALU of a SIMD do not overlap on gfx950).
 * entries [n_batches][2][16] int32: ONE list per (node, tile of 32 relations) sorted by relation and padded to 16 entries,
 * plane 0 = BYTE offset of the gathered table row (other * ld_table * 4), plane 1 = 260 * (rel % 32) | 0x3f800000 at every entry
 * of a (relation, node) row but its first (padding: 0 / 260 * 32); desc, padding batches at the end, t and datt_slabs as above
 * (a slab = 8 nodes x 64 channels). */
int tipk_rgcn_row_products_s_supported(int64_t n_nodes, int64_t n_rel, int n_bases, int channels);
int tipk_rgcn_row_products_s(const float* table, int64_t ld_table, int64_t n_nodes, int channels, const float* att,
                             int64_t ld_att, int64_t n_rel, int n_bases, const int32_t* entries, const int32_t* desc,
                             const float* xb /* nullable */, int64_t ld_xb, float* t, float* datt_slabs /* nullable */,
                             tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 3. Small row-wise glue (each replaces one or more torch elementwise/copy kernels, K3/K8).
 */
/* out[c, r] = in[r, c]  -- `lin(x)` for identity features is W^T (src/layers.py:392 with
 * prepare.py:22-23), and dW = (dXlin)^T on the way back. */
int tipk_transpose(const float* in, int64_t rows, int64_t cols, float* out, tipk_stream_t stream);

/* out[r, c] (+)= in[r, c] * (row_mul ? row_mul[r] : 1) / (row_div ? row_div[r] : 1)
 *               * (gate ? (gate[r, c] > 0) : 1)
 * covers x/d_norm + cat/add (src/layers.py:534-539), 1/deg scaling of upstream gradients and the
 * ReLU backward gate (src/layers.py:393,547). */
int tipk_rows_affine(const float* in, int64_t ld_in,
                     const float* row_mul, const float* row_div,
                     const float* gate, int64_t ld_gate,
                     float* out, int64_t ld_out,
                     int64_t rows, int64_t cols, int accumulate, tipk_stream_t stream);

/* out[r, c] = in[r, c] * (gate[r, c] > 0)  AND  scratch[g, c] = the column sums of out over the rows
 * of workgroup g (g < tipk_gate_colsum_groups(rows, cols); 0 = unsupported, cols > 256): the ReLU
 * backward of GCNConv 1 (src/layers.py:393) and stage 1 of its bias gradient in one pass; the groups
 * are added in order by tipk_sum_slabs. */
int tipk_gate_colsum_groups(int64_t rows, int64_t cols);
int tipk_gate_colsum(const float* in, int64_t ld_in, const float* gate, int64_t ld_gate,
                     float* out, int64_t ld_out, int64_t rows, int64_t cols, float* scratch,
                     tipk_stream_t stream);

/* The drug feature mix of FMEncoder in one forward launch (src/layers.py:526-539 with the dense map of
 * MyHierarchyConv, :239):   x0 = cat(xd / d_norm, mean W)  (cat != 0)   or   x0 = xd / d_norm + mean W  (q == ne).
 * xd [rows x ne], mean [rows x p], W [p x q] row-major contiguous, p, q <= 64; d_norm nullable (= 1). */
int tipk_drug_mix_fwd(const float* xd, int64_t ld_xd, const float* d_norm, const float* mean, int64_t ld_mean,
                      const float* w, int p, int q, int64_t rows, int ne, int cat, float* out, int64_t ld_out,
                      tipk_stream_t stream);

/* The P -> D stage fused with the drug feature mix, one launch per pass (src/layers.py:526-539 with MyHierarchyConv,
 * :229-242: mean over a drug's protein targets, dense map, /d_norm, cat | add):
 *     mean[d, :] = scale[d] * sum_{e in [ptr[d], ptr[d + 1])} h[src[e], :]         (scale = 1 / max(1, #targets))
 *     out[d, :]  = cat(xd[d] / d_norm[d], mean[d] W)   (cat != 0)   or   xd[d] / d_norm[d] + mean[d] W   (q == ne)
 *   h [n_src x p] (row stride ld_h), W [p x q] contiguous, p and q even and <= 64 (tipk_drug_mix_gather_supported),
 *   CSR by drug: ptr int32 [rows + 1], src int32 [edges]; `mean` [rows x p] contiguous is an OUTPUT (kept for the backward).
 *   wg_desc int32 [n_wg][2] = { first, n | W << 8 } of a 16-wave workgroup: rows order[first .. first + n) (order = NULL: the
 *   drugs first .. first + n - 1 themselves), W wavefronts per row sharing its edges, n <= 16 / W, W in {1, 4, 16} (W = 0: the
 *   round-5 form -- n consecutive drugs a wavefront each, or ONE drug on all 16) -- every drug exactly once.  Round 6 deals the
 *   drugs by edge count (tip_amd.layers.hier_graph): > 512 edges alone, 65 ... 512 four to a workgroup, the rest sixteen.
 * Backward, from g = d out [rows x (cat ? ne + q : ne)], one launch (tipk_drug_mix_bwd):
 *     g_xd   = g[:, :ne] / d_norm                          (nullable: not wanted)
 *     g_mean = g_pd W^T                                    [rows x p] contiguous (nullable), g_pd = the last q (cat) / all
 *                                                          (add) columns of g; the caller gathers it back to the source rows
 *                                                          on the transposed plan (tipk_gather_sum, edge weights = scale)
 *     g_w    = mean^T g_pd                                 [p x q]
 * All sums in fixed order. */
int tipk_drug_mix_gather_supported(int p, int q);
int tipk_drug_mix_gather_fwd(const float* xd, int64_t ld_xd, const float* d_norm, const float* h, int64_t ld_h,
                             const int32_t* ptr, const int32_t* src, const float* scale, const int32_t* wg_desc,
                             const int32_t* order /* nullable */, int64_t n_wg,
                             const float* w, int p, int q, int64_t rows, int ne, int cat, float* out, int64_t ld_out,
                             float* mean, tipk_stream_t stream);
int tipk_drug_mix_bwd(const float* g, int64_t ld_g, const float* d_norm, const float* mean, const float* w, int p, int q,
                      int64_t rows, int ne, int cat, float* g_xd, int64_t ld_gxd, float* g_mean, float* g_w,
                      tipk_stream_t stream);

/* Round 6 -- the same forward launch ALSO leaves the first R-GCN layer's row-local products of the rows it finishes
 * (src/layers.py:545 applied to the output of :532-539; reference MyRGCNConv2.forward :159-188, basis-first):
 *     xb[d, b, :d_out] = out[d, :] basis[b]      node-major [.. x n_bases x 32] (rows padded to 32 columns: the operand of
 *                                                tipk_pair_product / tipk_rgcn_pair_grads; columns >= d_out untouched)
 *     xroot[d, :]      = out[d, :] root          [rows x d_out] contiguous
 *   basis [n_bases x cols x d_out], root [cols x d_out] contiguous, cols = cat ? ne + q : ne (a multiple of 4, <= 128),
 *   d_out 16 or 32 (tipk_drug_mix_gather_xb_supported).  v_mfma_f32_16x16x4_f32 on the workgroup's <= 16 rows: a k-ordered
 *   fp32 fma chain per element.  Saves the launch of the two products (9.6 us at BioSNAP for 88 MFLOP). */
int tipk_drug_mix_gather_xb_supported(int p, int q, int ne, int cat, int n_bases, int d_out);
int tipk_drug_mix_gather_xb_fwd(const float* xd, int64_t ld_xd, const float* d_norm, const float* h, int64_t ld_h,
                                const int32_t* ptr, const int32_t* src, const float* scale, const int32_t* wg_desc,
                                const int32_t* order /* nullable */, int64_t n_wg, const float* w, int p, int q, int64_t rows, int ne, int cat, float* out,
                                int64_t ld_out, float* mean, const float* basis, const float* root, int n_bases, int d_out,
                                float* xb, float* xroot, tipk_stream_t stream);

/* Round 6 -- the backward pass of the whole P -> D stage in ONE launch: tipk_drug_mix_bwd + the transposed gather of
 * d mean + the products of GCNConv 2's backward pass (autograd of src/layers.py:526-539 and of PPEncoder.conv2, :394):
 *     g_xd = g[:, :ne] / d_norm                                             as tipk_drug_mix_bwd
 *     g_w_slabs[j] = mean^T g_pd over the j-th share of the rows  [p x q], tipk_pd_stage_bwd_wh_slabs() slabs
 *     g_h[s] = sum_{e in [tptr[s], tptr[s + 1])} tw[e] * (g_pd W^T)[tdst[e]]     per kept source row s (CSR by source row:
 *                                                                                drug ids, 1 / #targets(drug)); never stored
 *     gw[s, :] = (g_h[s] W2) * row_scale[s]        [n_src x c1]; W2 (k < p, n < c1) at w2[k * w2_sk + n * w2_sn]
 *     dw2_slabs[j] = agg[rows of workgroup j]^T g_h    [c1 x p] per row workgroup,  db2_slabs[j] = column sums of g_h  [p]
 *   agg [n_src x c1]: GCNConv 2's aggregated input rows (tipk_gather_sum_lin's first output).  wg_rows int32 [n_row_wg + 1]:
 *   the deal of the source rows to workgroups -- consecutive rows, at most max_rows of them and (unless one row alone has more)
 *   at most max_edges edges per workgroup (tipk_pd_stage_bwd_limits), wg_rows[0] = 0, wg_rows[n_row_wg] = n_src.  The slabs --
 *   n_row_wg of each -- are summed in order by the caller (riders of the next launch).
 *   Supported: tipk_drug_mix_gather_supported(p, q), c1 <= 64, p * c1 <= 4 096. */
int tipk_pd_stage_bwd_supported(int p, int q, int64_t rows, int c1);
int tipk_pd_stage_bwd_limits(int* max_rows, int* max_edges);
int tipk_pd_stage_bwd_wh_slabs(void);
int tipk_pd_stage_bwd(const float* g, int64_t ld_g, const float* d_norm, const float* mean, const float* w, int p, int q,
                      int64_t rows, int ne, int cat, float* g_xd, int64_t ld_gxd, float* g_w_slabs,
                      const int32_t* tptr, const int32_t* tdst, const float* tw, int64_t n_src,
                      const int32_t* wg_rows, int64_t n_row_wg,
                      const float* agg, int64_t ld_agg, int c1, const float* w2, int64_t w2_sk, int64_t w2_sn,
                      const float* row_scale, float* gw, int64_t ld_gw, float* dw2_slabs, float* db2_slabs,
                      tipk_stream_t stream);

/* out[c] = sum_r in[r, c]  (bias gradients of GCNConv).  `scratch` holds >= 256*cols floats. */
int tipk_col_sum(const float* in, int64_t ld_in, int64_t rows, int64_t cols,
                 float* scratch, float* out, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 4. DistMult decoder  score(u,v,r) = sigma( sum_k z[u,k] z[v,k] w[r,k] )
 *    replaces MultiInnerProductDecoder.forward, src/layers.py:590-592 (K9) and its backward.
 *    idx_bytes / et_bytes: 4 (int32) or 8 (int64, the reference's dtype) -- read in place.
 *    tipk_distmult_loss and tipk_typed_negative_sampling also take idx_bytes = 2: PACKED pairs, one uint32 word
 *    u | v << 16 per triple in the `_u` array (the `_v` pointer is unused; n_nodes <= 65535) -- the static positives
 *    are narrowed once, the sampler emits its negatives in this form: 66 MB of ids per BioSNAP step instead of 266 MB.
 */
int tipk_distmult_fwd(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                      const void* idx_u, const void* idx_v, int idx_bytes,
                      const void* edge_type, int et_bytes, int64_t n_triples,
                      int sigmoid, float* score, tipk_stream_t stream);

/* g_z [n_nodes x k] and g_w [n_rel x k] are ACCUMULATED into (caller zeroes them);
 * `score` is the forward output when sigmoid != 0 (unused otherwise).
 * tasks (nullable): int32 [n_tasks, 4] = (relation, begin, end, pos_weight) -- ranges of AT MOST 2048
 * positions that share one relation, covering [0, n_triples), largest first.  pos_weight is read by
 * tipk_distmult_loss only: the weight (1, or 2 / 0) of the task's POSITIVE triples -- when every relation
 * lists each pair in both directions ([u<v half | mirrored half], src/utils.py:35-65) the host may give
 * the first half weight 2 and the mirrored half weight 0 (identical scores and gradients): the objective
 * and all gradients are unchanged, a quarter of the work disappears.  TIP's triples are grouped by relation
 * (src/utils.py:57-63), so the host builds this once; with it (and k a power of two in 4..64,
 * 2*n_nodes*(k+4)*4 B <= 150 KB) the LDS-resident fast kernel runs, otherwise the generic one. */
int tipk_distmult_bwd(const float* g_score, const float* score,
                      const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                      const void* idx_u, const void* idx_v, int idx_bytes,
                      const void* edge_type, int et_bytes, int64_t n_triples,
                      int sigmoid, const int32_t* tasks, int64_t n_tasks,
                      float* g_z, float* g_w, tipk_stream_t stream);

/* Fused training objective of TIP.forward (src/layers.py:335-340, K9+K10): one pass over the
 * positive triples and their negatives (same relation per position):
 *   loss = -mean log(sigma(pos)+1e-13) - mean log(1-sigma(neg)+1e-13)
 * loss_out[0] += loss (caller zeroes); if g_z/g_w are non-NULL they receive d loss / d z, d w
 * (accumulated; caller zeroes), so the backward pass is a scale by the upstream scalar.
 *
 * workspace (nullable; tipk_distmult_workspace_bytes(...) bytes, 8-byte aligned, ZEROED before its first use;
 * the call leaves it zeroed): with it -- and a task table, i.e. on the LDS-resident fast kernel -- the sums across
 * workgroups of loss, d z and d w are 64-bit fixed-point integer atomics (exact, order-independent) converted by
 * a finalize launch: the objective and its gradients are BITWISE REPRODUCIBLE from run to run (the float
 * atomics of the default path differ at the 1e-7 level with the arrival order of 256 workgroups).
 * With k in {4, 8, 16} this path runs distmult_objective_kernel: one lane evaluates a position (ids straight from global
 * memory, sigma / log / quotient on the hardware transcendental units), the gradient terms are scattered with k lanes per
 * row of the LDS image, and every term is converted to fixed point with a PER-TERM scale (|term| <= 4 zmax wmax / n:
 * one multiply, one round, one 32-bit convert instead of a double-precision sequence); option "dm_task_kernel" = 1
 * keeps the k/4-lanes-per-position kernel of round 2 for A/B runs. */
int64_t tipk_distmult_workspace_bytes(int64_t n_nodes, int k, int64_t n_rel);
int tipk_distmult_loss(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                       const void* pos_u, const void* pos_v, const void* neg_u, const void* neg_v,
                       int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples,
                       const int32_t* tasks, int64_t n_tasks,
                       float* loss_out, float* g_z, float* g_w, void* workspace, tipk_stream_t stream);
/* Same, but loss_out / g_z / g_w are OVERWRITTEN (they need not be zeroed: three fill launches less per training
 * step; the bits are those of tipk_distmult_loss on zeroed outputs).  Only the workspace path can do that (its finalize
 * launch visits every output element): TIPK_EUNSUPPORTED otherwise, nothing launched. */
int tipk_distmult_loss_store(const float* z, int64_t n_nodes, int k, const float* rel_w, int64_t n_rel,
                             const void* pos_u, const void* pos_v, const void* neg_u, const void* neg_v,
                             int idx_bytes, const void* edge_type, int et_bytes, int64_t n_triples,
                             const int32_t* tasks, int64_t n_tasks,
                             float* loss_out, float* g_z, float* g_w, void* workspace, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 4b. NNDecoder triple scoring (reference src/layers.py:598-637, the paper's DR-NN ablation; SURVEY
 *     section 8(f) item 1):  score[e] = sigma( s1[u_e, r_e] + s2[v_e, r_e] )  with the dense tables
 *     s1 = relu(z w1_l1) w1_l2^T and s2 = relu(z w2_l1) w2_l2^T ([n_nodes x n_rel], row stride ld)
 *     produced by tipk_gemm_f32.  bwd ACCUMULATES d s1 / d s2 (caller zeroes) with float atomics.
 */
int tipk_pair_table_fwd(const float* s1, const float* s2, int64_t ld,
                        const void* idx_u, const void* idx_v, int idx_bytes,
                        const void* edge_type, int et_bytes, int64_t n_triples,
                        int sigmoid, float* score, tipk_stream_t stream);
int tipk_pair_table_bwd(const float* g_score, const float* score, int64_t ld,
                        const void* idx_u, const void* idx_v, int idx_bytes,
                        const void* edge_type, int et_bytes, int64_t n_triples,
                        int sigmoid, float* g_s1, float* g_s2, tipk_stream_t stream);
/*     The fused TIP objective on the tables (src/layers.py:335-340 with this decoder, model/ddm-nn.py:65-102):
 *        loss = -mean log(sigma(x_pos) + eps) - mean log(1 - sigma(x_neg) + eps),  x = S1[u, r] + S2[v, r]
 *     on TRANSPOSED tables s1t / s2t [n_rel][ld] (row r = every node's score under relation r: w_l2 . relu(z w_l1)^T from
 *     tipk_gemm_f32).  pos_pairs / neg_pairs [n_positions]: u | v << 16 per triple (n_nodes <= 65 535), both grouped by
 *     relation with the SAME blocks rel_ptr [n_rel + 1] (the sampler's layout); order [n_rel]: the relations by decreasing
 *     block size (launch order of the per-relation workgroups).  Outputs: loss_parts [n_rel][2] (double) = a relation's
 *     sums of log(sigma_pos + eps) and log(1 - sigma_neg + eps): loss = -(sum of all) / n_positions; g_s1t / g_s2t
 *     [n_rel][ld] (both or neither; columns < n_nodes of EVERY row are written) = d loss / d S1^T, d S2^T, accumulated in
 *     LDS as 64-bit fixed point: no atomics on global memory, bitwise reproducible. */
int tipk_pair_table_loss(const float* s1t, const float* s2t, int64_t ld, int64_t n_nodes, int64_t n_rel,
                         const uint32_t* pos_pairs, const uint32_t* neg_pairs, const int64_t* rel_ptr,
                         const int32_t* order, int64_t n_positions, float eps, double* loss_parts,
                         float* g_s1t /* nullable */, float* g_s2t /* nullable */, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 5. Typed negative sampling on device -- replaces typed_negative_sampling / negative_sampling,
 *    src/neg_sampling.py:5-26 (K11: host numpy + one D2H copy per relation).
 *
 * For every positive position e of relation r (rel_ptr[r] <= e < rel_ptr[r+1]) draw a pair
 * uniformly from n_nodes^2 (with replacement, self pairs allowed, as np.random.choice does) and
 * redraw while it equals a positive pair OF THE SAME RELATION.  Randomness: Philox4x32-10, key = (seed_lo, seed_hi);
 * n^2 < 2^32: ONE call (counter = (e >> 2, attempt, 0)) serves the four positions 4q .. 4q + 3, a candidate is the high
 * word of x * n^2 and is redrawn when the low word is below (2^32 - n^2) mod n^2 (exactly uniform); larger node sets:
 * counter = (e, attempt, 0), candidate = mulhi64(r0 | r1<<32, n^2) -- specified bit-exactly in
 * oracle/philox_sampler.py.  `pos_key_sorted` holds u*n+v of each relation's positives sorted ascending within the
 * relation (int64); `pos_key32` (nullable) the same keys as uint32, in any order inside a relation: what the bitmap
 * route reads when given (half the bytes).  After 64 rejected attempts the 64th candidate is kept (probability <
 * density^64).
 * wg_unit_ptr [n_wg + 1] / wg_units [n_units][3] (nullable, int32): edge-balanced deal of UNITS = (relation, first
 * position, end position) to n_wg workgroups -- the units tile [0, n_positions), a unit lies inside one relation (a
 * relation larger than a workgroup's share is cut into several units); with it and n_nodes^2 bits <= 150 KB each
 * workgroup tests candidates against an LDS bitmap of its relation's positives instead of searching the sorted keys
 * (same output bit for bit, ~6x faster on BioSNAP).
 * call_counter != NULL: a sampler STREAM whose state lives on the device, uint64[2] = { position, seed }:
 * the Philox key is splitmix64(state[1] + (state[0] + 1) * 0x9E3779B97F4A7C15) and the host `seed`
 * argument is ignored -- a captured hipGraph draws new negatives on every replay
 * and re-seeding after capture (a device-side write of the words) takes effect in the replays.  The position moves
 * on either by `tipk_counter_advance` on state[0] (a launch of its own), or -- advance != 0, state then is uint64[3] =
 * { position, seed, ticket (0) } -- inside the sampling launch: its last workgroup to finish stores position + 1.
 * pos_offset (nullable, int64 [n_rel]): the Philox counter of position e of relation r is e + pos_offset[r] -- a rank
 * of a relation-sharded run passes (start of r's block in the WHOLE triple list) - rel_ptr[r], so that its negatives
 * are exactly the negatives the unsharded run draws for those relations, whatever the number of ranks.
 */
/* workgroups per CU the bitmap sampler is launched with for this node count (plan the units for CUs x this many workgroups):
 * 2 when two bitmaps of n_nodes^2 bits fit the LDS of a CU (512-thread workgroups), 1 (1024 threads), 0 = no bitmap route */
int tipk_negsample_wgs_per_cu(int64_t n_nodes);
int tipk_typed_negative_sampling(const int64_t* pos_key_sorted, const int64_t* rel_ptr /* [n_rel+1] */,
                                 int64_t n_rel, int64_t n_nodes, uint64_t seed,
                                 uint64_t* call_counter /* nullable device uint64[2 or 3], see above */, int advance,
                                 const int32_t* wg_unit_ptr /* nullable */, const int32_t* wg_units, int64_t n_wg,
                                 const int64_t* pos_offset /* nullable */, const uint32_t* pos_key32 /* nullable */,
                                 void* out_u, void* out_v, int idx_bytes,
                                 int64_t n_positions /* = rel_ptr[n_rel], host copy */,
                                 tipk_stream_t stream);

int tipk_counter_advance(uint64_t* call_counter /* device */, tipk_stream_t stream);   /* *counter += 1 */

/* --------------------------------------------------------------------------------------------
 * 6. Per-relation ranking metrics on device -- replaces the loop of
 *    TIP.compute_auprc_auroc_ap_by_et (src/layers.py:355-375) over sklearn's roc_auc_score,
 *    average_precision_score and auc(precision_recall_curve) (src/utils.py:86-93): 1 097 D2H copies
 *    and sorts on the host.
 *
 * Relation r scores pos_score[range_ptr[r] : range_ptr[r+1]] against the same slice of neg_score
 * (labels 1 / 0).  out is fp64 [3][n_rel] = (AUPRC, AUROC, AP) rows.  Ties share one operating
 * point, exactly as sklearn's curves do.  max_pairs (host) = largest slice; 2*max_pairs <= 16384
 * (one workgroup sorts a relation in LDS), else TIPK_EUNSUPPORTED and the caller evaluates on the host.
 */
int tipk_rank_metrics(const float* pos_score, const float* neg_score, const int64_t* range_ptr /* [n_rel+1] */,
                      int64_t n_rel, int64_t max_pairs, double* out, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 7. Train / test split on device -- replaces `process_edges`, src/utils.py:35-65 (host numpy: one
 *    `np.random.binomial(1, p, E_r)` per relation from the global Mersenne state, Python lists, then
 *    `to_bidirection`), SURVEY.md section 8(f) item 2.
 *
 *    The undirected pairs of all relations arrive concatenated (pairs_u/pairs_v [n_pairs], 2-byte or
 *    8-byte ids; rel_ptr [n_rel + 1]).  Pair i is a TRAINING pair iff
 *        Philox4x32-10(counter = (i lo, i hi, 0, 0x53504C54), key = (seed lo, seed hi)).x0 < floor(p * 2^32)
 *    (p = 1 keeps every pair) -- specified bit-exactly in oracle/philox_split.py.
 *    tipk_split_flags  writes take[i] (1 = train) and ADDS the kept pairs per relation to n_train[r]
 *                      (uint64, caller zeroes).
 *    tipk_split_scatter  with train_ptr / test_ptr [n_rel + 1] = offsets of every relation's block of
 *                      DIRECTED edges in the outputs (2 * kept, 2 * dropped: exclusive sums the caller
 *                      forms from n_train), writes per relation r the kept pairs in list order as
 *                      [ (u,v) ... | (v,u) ... ] into train_u/v with train_et = r, the dropped ones the same
 *                      way into test_* -- the layout of `to_bidirection` (src/utils.py:17-23, :53).
 *    Outputs are int64 (the reference's dtype). */
int tipk_split_flags(const int64_t* rel_ptr, int64_t n_rel, int64_t n_pairs, double p_train, uint64_t seed,
                     uint8_t* take, uint64_t* n_train, tipk_stream_t stream);
int tipk_split_scatter(const void* pairs_u, const void* pairs_v, int idx_bytes, const int64_t* rel_ptr, int64_t n_rel,
                       const uint8_t* take, const int64_t* train_ptr, const int64_t* test_ptr,
                       int64_t* train_u, int64_t* train_v, int64_t* train_et,
                       int64_t* test_u, int64_t* test_v, int64_t* test_et, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 8. One-shot all-reduce over peer-mapped mailboxes -- the xGMI-aware exchange of the relation-sharded step
 *    (north_star: "the D-D pass shards by relation-id across the 8 GPUs of one node with an ... all-reduce of drug
 *    embeddings over xGMI"; the reference is single-GPU, README.md:58).  SUM of one fp32 buffer of n <= max_floats
 *    elements over `world` ranks (one process per GPU), result identical bit for bit on every rank (slots added in
 *    rank order).  tip_amd/csrc/tipk_peer.hip describes the protocol.
 *      tipk_peer_mailbox_bytes   size of one rank's mailbox (two halves of world slots + flags, + the call counter);
 *      tipk_peer_alloc / _free   the mailbox: UNCACHED device memory, zeroed (explicit allocation entry points: the
 *                                only ones of the library -- such memory cannot come from the caller's allocator);
 *      tipk_ipc_get_handle / _open / _close   hipIpc handle (64 bytes) of the own mailbox / mapping of a peer's;
 *      tipk_peer_allreduce       data [n] <- sum over ranks; mailboxes = HOST array [world] of the mailboxes' addresses
 *                                in THIS process (own one at [rank]); every rank must call with the same n, in the same
 *                                order.  Two launches (exchange + call counter), no host synchronisation: capturable.
 */
int64_t tipk_peer_mailbox_bytes(int world, int64_t max_floats);
int tipk_peer_alloc(int64_t bytes, void** ptr);
int tipk_peer_free(void* ptr);
int tipk_ipc_get_handle(void* ptr, void* handle_out /* 64 bytes, host */);
int tipk_ipc_open(const void* handle /* 64 bytes, host */, void** ptr);
int tipk_ipc_close(void* ptr);
int tipk_peer_allreduce(float* data, int64_t n, void* const* mailboxes /* host [world] */, int rank, int world,
                        int64_t max_floats, tipk_stream_t stream);
/*      BOUNDED WAIT: the flag wait of an exchange has a wall-clock budget (default 2 000 ms).  A rank whose peer died,
 *      skipped a call or took another collective does not spin for ever: the kernel records {sequence number << 32 |
 *      (missing rank + 1) << 16 | chunk} in the ERROR WORD of its own mailbox (first record wins) and finishes with
 *      whatever the slots hold.  tipk_peer_status copies that word to the host (0 = every wait was served; a synchronous
 *      8-byte copy: call it where the host synchronises anyway, never inside a capture); the caller must treat a
 *      non-zero word as fatal for the process group.  tipk_peer_set_timeout_ms: the budget of later launches (1 ... 600 000),
 *      ONE value for every exchange of the process (a short budget for a self-test, then a production budget that outlasts
 *      the host-side skew between ranks: tip_amd/dist.py leaves 60 000 ms behind). */
int tipk_peer_set_timeout_ms(int64_t ms);
int tipk_peer_status(void* mailbox, int world, int64_t max_floats, uint64_t* error_word /* host */);

/* --------------------------------------------------------------------------------------------
 * 9. Optimizer step of the training loop -- replaces `optimizer.step()` of tip.py:24-30
 *    (torch.optim.Adam(model.parameters(), lr=0.01): amsgrad off, maximize off, L2 weight decay).
 *    ONE launch over a list of fp32 tensors (host arrays [n_tensors] of device pointers; element i of params, grads,
 *    exp_avg, exp_avg_sq must share one dense layout of numel[i] elements):
 *        g += weight_decay * p;  m += (1 - beta1) (g - m);  v = beta2 v + (1 - beta2) g g;
 *        p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),      t = *steps[i] + 1
 *    steps: host array [n_tensors] of device uint64 words, the steps tensor i has made (torch keeps the count per
 *    parameter: one without a gradient sits the step out); ticket: device uint64[528], zeros (left zeroed; 33 ticket words, one per 128-byte line).  The launch's last workgroup adds 1 to
 *    the counts of its tensors -- a captured hipGraph keeps counting on replay.  Tensors with numel 0 are skipped (their
 *    count does not move).  The addresses are kernel arguments (48 tensors per launch, longer lists take several
 *    launches): nothing is uploaded, nothing allocated. */
int tipk_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, uint64_t* const* steps, uint64_t* ticket /* device */,
                   double lr, double beta1, double beta2, double eps, double weight_decay, tipk_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * 10. Op-level entries (round 6): a graph handle that OWNS the preprocessed buffers of a D-D graph
 *     and ONE call per pass of an R-GCN layer -- reference MyRGCNConv2.forward(x, edge_index, edge_type, range_list)
 *     (src/layers.py:157-188; with edge_type instead of range_list: MyRGCNConv.forward, :76-99) and its autograd -- for a host
 *     that is not this package's Python: raw device pointers + handle + caller-supplied workspace + stream.
 *
 *     tipk_graph_build (the only entry of the library that allocates, and that synchronises: preprocessing, once per graph):
 *         edge_index [2][n_edges] (row 0 sources, row 1 destinations), and either range_list [n_rel][2] (start, end) of
 *         consecutive blocks, as MyRGCNConv2 takes it, or edge_type [n_edges]; idx_bytes 4 | 8 for all three; the arrays may
 *         live on the device or on the host.  in_degree (nullable, float [n_nodes]): the in-degree of the WHOLE graph when the
 *         edge list is one rank's shard of the relations.  Ids out of range -> TIPK_EINVAL (the reference raises IndexError).
 *         The handle holds two CSRs (rows of the output by destination; rows (relation, source) by destination list) and
 *         1 / max(1, in-degree): edge order inside a row = order of the edge list, so every pass is bitwise reproducible.
 *     tipk_rgcn_fwd:   out = relu?( D^-1 sum_r A_r X W_r + X root ),  W_r = sum_b att[r, b] basis[b]
 *         x [N x d_in] (row stride ld_x), basis [n_bases x d_in x d_out], att [R x n_bases], root [d_in x d_out] contiguous.
 *     tipk_rgcn_bwd:   from grad_out [N x d_out]; out_relu (nullable) = the forward output when relu was set (its mask is
 *         applied first) -> g_x [N x d_in], g_basis, g_att, g_root (shapes of the parameters, contiguous, overwritten).
 *     workspace: tipk_rgcn_workspace_bytes(graph, d_in, d_out, n_bases) bytes, 16-byte aligned, caller-owned, reusable
 *         between calls on one stream (the backward pass recomputes XB: nothing is kept across the two calls but the
 *         caller's own X, parameters and -- with relu -- the output).
 *     Routes.  Generic (any shapes): basis-first, transform-then-gather, Y = att . XB [R N x d_out] the largest temporary.
 *     PAIR FORM (what the PyTorch modules take at BioSNAP size; both D-D layers forward + backward through these entries:
 *     1.9 ms on the generic route, 0.18 ms in pair form -- tools/bench_c_abi.py, `op_level_c_abi` on the bench line):
 *     tipk_graph_prepare_rgcn(graph, n_bases, d_out) builds the plans of the LDS-resident pair form for that layer shape --
 *     on the host, with the C++ builders of section 10c; allocates and synchronises like tipk_graph_build -- when the graph
 *     qualifies (<= 1 024 nodes, an att table [R x n_bases] that fits in LDS, n_bases and d_out the pair kernels support);
 *     TIPK_EUNSUPPORTED otherwise, and the generic route stays.  Call it once per layer shape BEFORE asking for the
 *     workspace size; tipk_rgcn_fwd / _bwd then run sections 1d + 2c (+ 2e) on the handle's plans.  tipk_graph_rgcn_route:
 *     0 generic, 1 pair-form forward (backward generic), 2 pair form both ways.  The handle keeps the edge list on the host
 *     (12 bytes per edge) for later prepare calls until tipk_graph_release_host.
 *     tipk_rgcn_bwd_ex(..., flags, stream): TIPK_RGCN_WORKSPACE_FROM_FWD = the workspace still holds what tipk_rgcn_fwd of THIS
 *     layer left in it (pair cells and XB: the caller ran nothing else on that workspace in between) -- the backward pass then
 *     skips recomputing them; tipk_rgcn_bwd = flags 0 (always safe).  On a prepared route the handle's gradient table is
 *     scratch of the backward call: one backward pass per handle at a time (calls on one stream are ordered anyway).
 */
#define TIPK_RGCN_WORKSPACE_FROM_FWD 1
typedef struct tipk_graph tipk_graph;
int tipk_graph_build(const void* edge_index, const void* edge_type /* nullable */, const void* range_list /* nullable */,
                     int idx_bytes, int64_t n_edges, int64_t n_nodes, int64_t n_rel, const float* in_degree /* nullable */,
                     tipk_graph** graph_out);
int tipk_graph_destroy(tipk_graph* graph);
int tipk_graph_info(const tipk_graph* graph, int64_t* n_nodes, int64_t* n_rel, int64_t* n_edges,
                    const float** inv_degree /* device, owned by the handle */);
int64_t tipk_rgcn_workspace_bytes(const tipk_graph* graph, int d_in, int d_out, int n_bases);
int tipk_rgcn_fwd(const tipk_graph* graph, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                  const float* root, int n_bases, int d_out, int relu, float* out, int64_t ld_out,
                  void* workspace, int64_t workspace_bytes, tipk_stream_t stream);
int tipk_rgcn_bwd(const tipk_graph* graph, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                  const float* root, int n_bases, int d_out, const float* grad_out, int64_t ld_g,
                  const float* out_relu /* nullable */, int64_t ld_relu, float* g_x, int64_t ld_gx, float* g_basis, float* g_att,
                  float* g_root, void* workspace, int64_t workspace_bytes, tipk_stream_t stream);
int tipk_rgcn_bwd_ex(const tipk_graph* graph, const float* x, int64_t ld_x, int d_in, const float* basis, const float* att,
                     const float* root, int n_bases, int d_out, const float* grad_out, int64_t ld_g,
                     const float* out_relu /* nullable */, int64_t ld_relu, float* g_x, int64_t ld_gx, float* g_basis, float* g_att,
                     float* g_root, void* workspace, int64_t workspace_bytes, int flags, tipk_stream_t stream);
int tipk_graph_prepare_rgcn(tipk_graph* graph, int n_bases, int d_out);
int tipk_graph_rgcn_route(const tipk_graph* graph, int n_bases, int d_out);
int tipk_graph_release_host(tipk_graph* graph);

/* 10b. The other two layer kinds of the path behind the same kind of handle (tipk_graph_destroy frees them all):
 *   GCNConv as PPEncoder uses it (PyG 2.0.1 semantics, src/layers.py:386-394): tipk_gcn_graph_build forms
 *   A_hat = D^-1/2 (A + I) D^-1/2 -- existing self loops replaced by exactly one unit loop per node, D = in-degree incl. the
 *   loop, flow source -> target -- as two weighted CSRs.
 *     tipk_gcn_fwd:  out = relu?( A_hat (x W^T) + bias );  x = NULL: identity features (lin(I) = W^T, read in place: w_so must be 1);
 *                    weight element (o, i) at weight[o * w_so + i * w_si] (any layout of the [out, in] parameter).
 *     tipk_gcn_bwd:  grad_out (masked with out_relu > 0 when given) -> g_x (nullable; needs x), g_weight (element (o, i) at
 *                    g_weight[o * gw_so + i * gw_si]; dense x: gw_si must be 1; identity: gw_so must be 1 -- d W is the transposed
 *                    aggregate written in place), g_bias (nullable).
 *   MyHierarchyConv (src/layers.py:196-247): tipk_hier_graph_build keeps the edges that end in rows [n_source, n_all) of the
 *   concatenated node space; tipk_hier_fwd: out [n_all - n_source x d_out] = mean over incoming edges of x [n_all x d_in] . weight
 *   [d_in x d_out]; tipk_hier_bwd -> g_x [n_all x d_in] (nullable), g_weight.
 *   Workspaces: tipk_gcn_workspace_bytes / tipk_hier_workspace_bytes, 16-byte aligned, caller-owned.  (The fused objective of the
 *   DistMult decoder is one C call already: tipk_distmult_loss_store, section 4.) */
int tipk_gcn_graph_build(const void* edge_index, int idx_bytes, int64_t n_edges, int64_t n_nodes, tipk_graph** graph_out);
int64_t tipk_gcn_workspace_bytes(const tipk_graph* graph, int d_in, int d_out);
int tipk_gcn_fwd(const tipk_graph* graph, const float* x /* nullable */, int64_t ld_x, int d_in, const float* weight, int64_t w_so,
                 int64_t w_si, const float* bias /* nullable */, int d_out, int relu, float* out, int64_t ld_out,
                 void* workspace, int64_t workspace_bytes, tipk_stream_t stream);
int tipk_gcn_bwd(const tipk_graph* graph, const float* x /* nullable */, int64_t ld_x, int d_in, const float* weight, int64_t w_so,
                 int64_t w_si, int d_out, const float* grad_out, int64_t ld_g, const float* out_relu /* nullable */, int64_t ld_relu,
                 float* g_x /* nullable */, int64_t ld_gx, float* g_weight, int64_t gw_so, int64_t gw_si, float* g_bias /* nullable */,
                 void* workspace, int64_t workspace_bytes, tipk_stream_t stream);
int tipk_hier_graph_build(const void* edge_index, int idx_bytes, int64_t n_edges, int64_t n_all, int64_t n_source,
                          tipk_graph** graph_out);
int64_t tipk_hier_workspace_bytes(const tipk_graph* graph, int d_in, int d_out);
int tipk_hier_fwd(const tipk_graph* graph, const float* x, int64_t ld_x, int d_in, const float* weight, int d_out, float* out,
                  int64_t ld_out, void* workspace, int64_t workspace_bytes, tipk_stream_t stream);
int tipk_hier_bwd(const tipk_graph* graph, const float* x, int64_t ld_x, int d_in, const float* weight, int d_out,
                  const float* grad_out, int64_t ld_g, float* g_x /* nullable */, int64_t ld_gx, float* g_weight,
                  void* workspace, int64_t workspace_bytes, tipk_stream_t stream);

/* 10c. Plan construction in C++ (tip_amd/csrc/tipk_pairplan.hip), exposed as HOST arrays: the work lists of the wave-stream
 *     gathers (section 1d) and of the pair-form backward pass (section 2e) from index arrays in host memory -- the same arrays,
 *     bit for bit, as tip_amd/plan.py `build_stream_plan_rows` / `build_pair_bwd_plan` and tip_amd/layers.py `pair_link_words`
 *     (tests/test_host_plans.py).  The graph handle of section 10 builds its pair-form plans with these (tipk_graph_prepare_rgcn);
 *     a host that manages its own device buffers can upload them and call the kernel-level entries itself.
 *     Arrays by name -- stream plan: "wave_ptr", "cells", "ids" (uint16), "zero_ptr", "zero_rows"; pair backward plan: "slots",
 *     "node_desc", "tile_node", "part_first", "wg_part" and the gather's under "gather.<name>"; link words: "links".
 *     Scalars by name: "n_rows", "n_table", "n_bands", "n_edges", "n_wg", "lanes", "piece", "idx_unit", "row_bytes" (prefixed with
 *     "gather." on a pair plan) and "n_slots", "n_parts", "part_len", "n_alloc", "symmetric"; -1 = unknown name.
 *     wide_steps / row_bytes 0 = the defaults (16; lanes * 16). */
typedef struct tipk_host_plan tipk_host_plan;
int tipk_plan_stream_rows(const int64_t* out_row, const int64_t* tab_row, int64_t n_edges, int64_t n_rows, int64_t n_table,
                          int n_wg, int lanes, int piece, int wide_steps, int row_bytes, tipk_host_plan** plan_out);
int tipk_plan_pair_bwd(const int64_t* src, const int64_t* dst, const int64_t* rel, int64_t n_edges, int64_t n_nodes, int64_t n_rel,
                       const float* scale /* [n_nodes] 1 / in-degree */, int symmetric, int n_wg, int lanes, int piece,
                       tipk_host_plan** plan_out);
/* grouped gather plan of section 1 (plan.py `build_gather_plan` with group_slots = G > 0; chunk 0 = from the edge count):
 * arrays "row_id", "edge_w" (float; empty without weights), "items", "perm" (int64); scalars "n_items", "chunk", "group_slots". */
int tipk_plan_gather(const int64_t* out_row, const int64_t* table_row, const float* edge_w /* nullable */, int64_t n_edges,
                     int64_t n_out, int64_t n_table, int chunk, int group_slots, tipk_host_plan** plan_out);
int tipk_plan_link_words(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, tipk_host_plan** plan_out);
int tipk_host_plan_array(const tipk_host_plan* plan, const char* name, const void** data, int64_t* count, int* elem_bytes);
int64_t tipk_host_plan_scalar(const tipk_host_plan* plan, const char* name);
void tipk_host_plan_free(tipk_host_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* TIPK_H */
