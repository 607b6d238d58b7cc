/*
 * gsat_hip.h -- C ABI of libgsat_hip.so, the MI355X (gfx950) implementation of GSAT's
 * edge-attention -> stochastic mask -> masked message passing hot path.
 *
 * The reference (mihikamd/DP-GSAT) has no FFI: its boundary is a Python nn.Module protocol whose
 * arithmetic runs inside torch-geometric / torch-scatter / torch-sparse kernels.  Each entry point
 * below replaces the chain of third-party launches behind one reference call site, cited as
 * "replaces: <file:line>" (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success or a negative GSAT_ERR_* code; no exception crosses the
 *     ABI; gsat_last_error() returns a thread-local message for the last failure on this thread.
 *   - all tensor pointers are DEVICE pointers, contiguous row-major, fp32 unless typed otherwise;
 *     edge_index / batch keep the reference's int64 API type, everything the library produces for
 *     its own kernels (CSR arrays, permutations, segment pointers) is int32.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never allocates,
 *     frees or retains device memory, never synchronises the host, and only enqueues work on the
 *     `stream` argument (a hipStream_t passed as void*), so calls are graph-capturable and
 *     re-entrant (forward on the Python thread, backward on autograd's worker thread).
 *   - "nullable" pointers may be NULL to switch the corresponding term off.
 */
#ifndef GSAT_HIP_H
#define GSAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSAT_ABI_VERSION 4

#define GSAT_OK 0
#define GSAT_ERR_HIP (-1)        /* a HIP runtime call failed */
#define GSAT_ERR_ARG (-2)        /* bad argument (null pointer, negative extent, unsupported width) */
#define GSAT_ERR_WORKSPACE (-3)  /* workspace too small */
#define GSAT_ERR_UNSUPPORTED (-4)
#define GSAT_ERR_BLAS (-5)         /* reserved: no entry point returns it since the library GEMMs were replaced (round 1) */

int gsat_abi_version(void);

/* Device-resident seed stream for captured steps: state[0] = base seed (written once by the host), state[1] = counter; every call bumps
 * the counter and writes splitmix64(base, counter) (63 bits) to out[0], which the dropout / sampler kernels read through their seed_dev
 * argument.  One launch; deterministic given the base seed and the number of calls. */
int gsat_seed_next(uint64_t* state, uint64_t* out, void* stream);
const char* gsat_last_error(void);

/* =============================== integer edge bookkeeping ==================================== */

/* Workspace (bytes) needed by gsat_build_csr / gsat_reverse_edge_perm for E edges. */
size_t gsat_csr_workspace_bytes(int64_t num_edges, int64_t num_rows);
size_t gsat_rev_workspace_bytes(int64_t num_edges);

/*
 * Group the E edge slots by `rows` (stable): perm = edge ids sorted by (rows[e], e),
 * rowptr[r]..rowptr[r+1] = slots of row r, other_sorted[k] = (int32) other[perm[k]].
 * replaces: the scatter index handling inside MessagePassing.propagate
 *           (src/models/conv_layers.py:21,44,163) -- done once per batch instead of per launch.
 * rows/other: int64[E]; rowptr: int32[num_rows+1]; other_sorted, perm: int32[E].
 * err_flag (int32[1], device, caller-zeroed): incremented for every out-of-range row id.
 */
int gsat_build_csr(const int64_t* rows, const int64_t* other, int64_t num_edges, int64_t num_rows,
                   int32_t* rowptr, int32_t* other_sorted, int32_t* perm, int32_t* err_flag,
                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * Both CSRs of one collated batch from ONE stable sort of 2E (row, edge id) keys -- by destination (forward aggregation) and by
 * source (every backward) -- plus what the host side derives from them: the hub-row chunk lists of both (gsat_row_chunks), the
 * by-source-slot -> by-destination-slot map used by the PNA backward, and int32 copies of the two edge_index rows.  Results are
 * identical to two gsat_build_csr calls (same stable order); the point is half the dependent launches per fresh batch.
 * edge_index [2,E] int64 (row 0 = source, row 1 = target).  replaces: the scatter index of MessagePassing.propagate
 * (src/models/conv_layers.py:21,44,163) for a whole batch.  workspace: gsat_csr_pair_workspace_bytes(E, N).
 * err_flag: int32[4], zeroed BY THE CALL (its first launch): [0] = number of out-of-range node ids, [1] / [2] = number of hub chunks of the
 * by-destination / by-source CSR (chunk_ptr_*[N]), so one small device->host read answers every host-side question.
 */
size_t gsat_csr_pair_workspace_bytes(int64_t E, int64_t num_nodes);
int gsat_build_csr_pair(const int64_t* edge_index, int64_t E, int64_t num_nodes, int32_t* rowptr_dst, int32_t* src_by_dst,
                        int32_t* eid_by_dst, int32_t* rowptr_src, int32_t* dst_by_src, int32_t* eid_by_src,
                        int32_t* slot_dst_of_srcslot, int32_t* chunk_ptr_dst, int32_t* chunk_ptr_src, int32_t* src32,
                        int32_t* dst32, int32_t* err_flag, void* workspace, size_t workspace_bytes, void* stream);

/*
 * rev[k] = id of the edge (dst_k, src_k); flags[0] = 1 iff the edge multiset equals its transpose
 * (is_undirected), flags[1] = number of sorted positions where edge and transposed keys differ.
 * Pairing rule for duplicate edges: stable (key, edge id) order on both sides.
 * replaces: is_undirected + torch_sparse.transpose + reorder_like
 *           (example/gsat.py:80-83, src/run_gsat.py:231-247, src/utils/utils.py:19-25).
 * edge_index: int64[2,E]; rev: int32[E]; flags: int32[2].
 */
int gsat_reverse_edge_perm(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes,
                           int32_t* rev, int32_t* flags, void* workspace, size_t workspace_bytes,
                           void* stream);

/*
 * The same permutation and flags from the two CSRs of gsat_build_csr_pair (its clamped src32 / dst32 included), without a sort:
 * the k-th copy (by edge id) of (s,d) pairs with the k-th copy of (d,s); an edge without a partner keeps rev[k] = k and counts
 * in flags[1] (so flags[1] = number of unpaired edges here, 0 iff undirected -- the only property the callers use).
 * Each edge walks the shorter of the two rows that hold its copies, so hub-leaf edges cost the leaf's degree.
 */
int gsat_reverse_edge_perm_csr(const int32_t* src32, const int32_t* dst32, const int32_t* rowptr_dst, const int32_t* src_by_dst,
                               const int32_t* eid_by_dst, const int32_t* rowptr_src, const int32_t* dst_by_src,
                               const int32_t* eid_by_src, int64_t num_edges, int64_t num_nodes, int32_t* rev, int32_t* flags,
                               void* stream);

/*
 * Segment pointer of a non-decreasing id vector (PyG `batch`): ptr[g]..ptr[g+1] = rows of
 * segment g.  flags[0] counts order violations / out-of-range ids (0 = valid input).
 * replaces: the per-call `degree(batch)` + scatter index of InstanceNorm and global pools
 *           (src/utils/get_model.py:50-51, src/models/gin.py:34, src/models/pna.py:47).
 */
int gsat_segment_ptr(const int64_t* seg_ids, int64_t num_rows, int64_t num_segments, int32_t* ptr,
                     int32_t* flags, void* stream);
/* same in ONE launch, plus the int32 copy seg_ids32[n] of the ids; *flags is only incremented: the caller zeroes it */
int gsat_segment_ptr32(const int64_t* seg_ids, int64_t n, int64_t num_segments, int32_t* ptr, int32_t* seg_ids32,
                       int32_t* flags, void* stream);

/* out[e] = batch[index[e]]  (edge -> graph id, `batch[col]` of example/gsat.py:136). int64 out; index values outside
 * [0, table_len) are clamped into the table (they are counted as range errors by the CSR builders). */
int gsat_gather_i64(const int64_t* table, int64_t table_len, const int64_t* index, int64_t n, int64_t* out, void* stream);

/* ============================ masked message passing: sum (GIN / GINE) ====================== */

/*
 * Rows with more than GSAT_LONG_ROW_EDGES entries (power-law hubs) are split into chunks of that many
 * consecutive CSR slots: chunk sums go to a partial workspace and are added per row in chunk order, so the
 * result stays atomics-free and bitwise reproducible.  chunk_ptr[r]..chunk_ptr[r+1] = chunk ids of row r
 * (empty for short rows); built once per CSR by gsat_row_chunks.  Pass chunk_ptr = NULL to process every
 * row whole.
 */
#define GSAT_LONG_ROW_EDGES 256
size_t gsat_row_chunks_workspace_bytes(int64_t num_rows);
int gsat_row_chunks(const int32_t* rowptr, int64_t num_rows, int32_t* chunk_ptr /* [num_rows+1] */,
                    void* workspace, size_t workspace_bytes, void* stream);
/* floats needed for the partial-sum workspace of a CSR with num_edges entries and row width H */
size_t gsat_long_row_partial_floats(int64_t num_edges, int64_t H);

/*
 * out[i,:] = self_coef * self_rows[i,:] + sum_{k in row i} w_k * msg_k
 *   msg_k = x[col[k],:]                                   (edge_emb == NULL : GINConv)
 *         = relu(x[col[k],:] + edge_emb[eid[k],:])         (edge_emb != NULL : GINEConv)
 *   w_k   = att[eid[k]]  (att == NULL -> 1)
 * replaces: GINConv.forward/message and GINEConv.forward/message up to `self.nn`
 *           (src/models/conv_layers.py:14-34, 37-66): index_select + mul + scatter_sum + add.
 * x: [*,H]; self_rows: [N,H] (NULL -> x; not read when self_coef == 0); att: [E] nullable;
 * edge_emb: [E,H] nullable; rowptr int32[N+1]; col, eid: int32[E]; out: [N,H].  H % 4 == 0, H <= 2048.
 * chunk_ptr / partial: long-row support (both nullable together), see above.
 */
int gsat_aggr_sum_fwd(const float* x, const float* self_rows, const float* att, const float* edge_emb,
                      const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                      int64_t num_rows, int64_t num_edges, int64_t H, float self_coef, float* out,
                      const int32_t* chunk_ptr, float* partial, void* stream);

/*
 * Backward of gsat_aggr_sum_fwd over the TRANSPOSED structure (edges grouped by source node j):
 *   dx[j,:]   = self_coef * dout[j,:] + sum_{k in srcrow j} att_k * dout[dst[k],:] (* [pre_k > 0])
 *   datt[eid[k]]        = < msg_k , dout[dst[k],:] >
 *   dedge_emb[eid[k],:] = att_k * dout[dst[k],:] * [pre_k > 0]        (GINE only)
 * replaces: autograd backward of the above (example/trainer.py:34, src/run_gsat.py:634).
 * rowptr_src int32[N+1]; dst_sorted, eid_src int32[E]; datt nullable; dedge_emb nullable;
 * chunk_ptr / partial: long-row support for the by-source CSR.
 */
int gsat_aggr_sum_bwd(const float* x, const float* att, const float* edge_emb, const float* dout,
                      const int32_t* rowptr_src, const int32_t* dst_sorted, const int32_t* eid_src,
                      int64_t num_rows, int64_t num_edges, int64_t H, float self_coef, float* dx, float* datt,
                      float* dedge_emb, const int32_t* chunk_ptr, float* partial, void* stream);

/* ============================ masked message passing: PNA multi-aggregation ================== */

/* aggregator codes (order of `aggregators` = order of the YAML list, src/configs/PNA-*.yml) */
#define GSAT_AGG_SUM 0
#define GSAT_AGG_MEAN 1
#define GSAT_AGG_MIN 2
#define GSAT_AGG_MAX 3
#define GSAT_AGG_VAR 4
#define GSAT_AGG_STD 5
/* scaler codes (src/models/conv_layers.py:229-259) */
#define GSAT_SCALE_IDENTITY 0
#define GSAT_SCALE_AMPLIFICATION 1
#define GSAT_SCALE_ATTENUATION 2
#define GSAT_SCALE_LINEAR 3
#define GSAT_SCALE_INVERSE_LINEAR 4

/*
 * out[i, ((s*A + a)*F + p*H) : +H] = scaler_s(deg_i) * aggr_a over in-edges k of row i of
 *   m_k = att_k * [ x[i,:] || x[col[k],:] (|| edge_emb[eid[k],:]) ],   F = 2H (3H with edge_emb)
 * mean: count clamped >= 1; min/max: 0 for empty rows; var = mean(m^2) - mean(m)^2;
 * std = sqrt(relu(var) + 1e-5) (so sqrt(1e-5) for empty rows); deg_i = unweighted in-degree.
 * replaces: PNAConvSimple.message + aggregate up to `post_nn`
 *           (src/models/conv_layers.py:160-185, aggregators :193-226, scalers :229-259).
 * aggregators / scalers are HOST int32 arrays (1..8 entries).  H % 4 == 0, H <= 512 (widths above 256 run as one launch per
 * 256-channel chunk; the tiled backward below covers 64 <= H <= 256).
 */
int gsat_pna_fwd(const float* x, const float* att, const float* edge_emb, const int32_t* rowptr,
                 const int32_t* col, const int32_t* eid, int64_t num_rows, int64_t H,
                 const int32_t* aggregators, int num_aggregators, const int32_t* scalers, int num_scalers,
                 float avg_deg_lin, float avg_deg_log, float* out, void* stream);

/*
 * Backward, destination-row pass.  For every in-edge slot k (CSR-by-destination order):
 *   dmsg[k,:]           = gradient w.r.t. the gathered row x[col[k],:]      ([E,H], slot order)
 *   datt[eid[k]]        = gradient w.r.t. att of that edge                    (nullable)
 *   dedge_emb[eid[k],:] = gradient w.r.t. edge_emb                            (nullable)
 *   dx_self[i,:]        = gradient w.r.t. x[i,:] through the x_i third of the message
 * min/max route to the FIRST slot attaining the extremum (torch-scatter CPU rule); std's relu has
 * zero slope at var <= 0.  The caller finishes dx[j,:] = dx_self[j,:] + sum_{slots of source j}
 * dmsg[slot,:] with gsat_aggr_sum_fwd over the by-source CSR (col = slot map, att = NULL).
 * replaces: autograd backward of the PNA scatter passes (example/trainer.py:34).
 */
int gsat_pna_bwd(const float* x, const float* att, const float* edge_emb, const float* dout,
                 const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t num_rows, int64_t H,
                 const int32_t* aggregators, int num_aggregators, const int32_t* scalers, int num_scalers,
                 float avg_deg_lin, float avg_deg_log, float* dx_self, float* dmsg, float* datt,
                 float* dedge_emb, void* stream);

/*
 * The same two passes for batches with long (hub) rows: chunk_ptr = the by-destination hub-chunk list of gsat_build_csr_pair /
 * gsat_row_chunks (rows of more than GSAT_LONG_ROW_EDGES in-edges split into chunks of that many), workspace =
 * gsat_pna_long_row_floats(E, H, edge_emb != NULL) floats.  A long row is no longer walked by one lane group: its chunks are
 * reduced in parallel (statistics + first min/max slots per chunk), folded in chunk order by the row's group, and -- backward --
 * its per-edge gradients are written chunk-parallel from the row's routing record.  Bitwise reproducible; short rows unchanged.
 * chunk_ptr == NULL behaves exactly like gsat_pna_fwd / gsat_pna_bwd.
 */
size_t gsat_pna_long_row_floats(int64_t num_edges, int64_t H, int has_edge_emb);
int gsat_pna_fwd_long(const float* x, const float* att, const float* edge_emb, const int32_t* rowptr,
                      const int32_t* col, const int32_t* eid, int64_t num_rows, int64_t num_edges, int64_t H,
                      const int32_t* aggregators, int num_aggregators, const int32_t* scalers, int num_scalers,
                      float avg_deg_lin, float avg_deg_log, float* out, const int32_t* chunk_ptr, float* workspace, void* stream);
int gsat_pna_bwd_long(const float* x, const float* att, const float* edge_emb, const float* dout,
                      const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t num_rows, int64_t num_edges, int64_t H,
                      const int32_t* aggregators, int num_aggregators, const int32_t* scalers, int num_scalers,
                      float avg_deg_lin, float avg_deg_log, float* dx_self, float* dmsg, float* datt,
                      float* dedge_emb, const int32_t* chunk_ptr, float* workspace, void* stream);

/*
 * Tiled backward for the reference's aggregator lists (mean,min,max,std[,sum]; identity scaler; no edge_attr third): one
 * launch writes dx directly.  A workgroup owns a window ("tile") of consecutive destination rows and their in-edges; the
 * per-edge gradient rows stay in LDS and are summed per source there, so dmsg [E,H] is only touched by edges whose source lies
 * outside the window ("spilled"); a second small launch adds those.  Summation
 * order per source: dx_self, in-window rows in by-source slot order, spilled rows in by-source slot order (bitwise reproducible).
 *   gsat_pna_tile_plan:   window geometry for width H and an LDS budget per workgroup (0 -> 80 KiB): windows span
 *                         rows_nominal .. rows_nominal + rows_slack rows and hold edges_cap in-edges in LDS.
 *   gsat_pna_build_tiles: tile_desc int32[T+1][4] (16-byte aligned) = (first row, rowptr[row], rowptr_src[row], 0) of every window,
 *                         T = ceil(N / rows_nominal); window t starts at max(start of the graph containing row
 *                         t*rows_nominal, t*rows_nominal - rows_slack) when node_ptr [G+1] / node_seg [N] (gsat_segment_ptr32) are
 *                         given -- block-diagonal batches then lose almost no edge to a window boundary -- else at t*rows_nominal.
 *                         Also lists, once per batch, the sources that have a spilled edge (spill_rows[0 .. *spill_count), any order).
 *   gsat_pna_bwd_tiled:   rows_cap >= the longest window (rows_nominal + rows_slack <= 2 rows_nominal); dmsg [E,H] is scratch.
 * replaces: autograd backward of PNAConvSimple.message/aggregate (src/models/conv_layers.py:166-185).
 */
int gsat_pna_tile_plan(int64_t H, int64_t lds_budget_bytes, int32_t* rows_nominal, int32_t* rows_slack, int32_t* edges_cap);
int gsat_pna_build_tiles(const int32_t* node_ptr, const int32_t* node_seg, const int32_t* rowptr, const int32_t* rowptr_src,
                         const int32_t* slot_dst_of_srcslot, int64_t num_rows, int rows_nominal, int rows_slack, int edges_cap,
                         int32_t* tile_desc, int32_t* spill_rows /* [num_rows] */, int32_t* spill_count /* [1] */, void* stream);
int gsat_pna_bwd_tiled(const float* x, const float* att, const float* dout, const int32_t* rowptr, const int32_t* col,
                       const int32_t* eid, const int32_t* tile_desc, int64_t num_tiles, int rows_nominal, int rows_cap, int edges_cap,
                       const int32_t* rowptr_src, const int32_t* slot_dst_of_srcslot, int64_t num_rows, int64_t num_edges, int64_t H,
                       const int32_t* aggregators, int num_aggregators, const int32_t* scalers, int num_scalers,
                       const int32_t* spill_rows, const int32_t* spill_count, float* dx, float* dmsg, float* datt,
                       const float* dx_add /* [N,H] or NULL: added to dx (a gradient reaching x by another path, e.g. the layer's residual) */,
                       void* stream);

/*
 * Node attention without the lift (example/gsat.py:112-117 `edge_att = node_att[src] * node_att[dst]`, then PNAConvSimple.message): the
 * aggregation kernels form the edge weight node_att[row] * node_att[source] at the load -- the same product, bit for bit, as
 * gsat_lift_fwd followed by gsat_pna_fwd -- so there is no [E] attention tensor, no edge-id indirection and no lift kernel either way.
 *   gsat_pna_fwd_node_att:        as gsat_pna_fwd without edge_emb / hub chunks; node_att [N].
 *   gsat_pna_bwd_tiled_node_att:  as gsat_pna_bwd_tiled; dnode_att [N] (may be NULL) = sum over the edges of a node, as destination and as
 *                                 source, of d w_e * node_att[other end], summed in CSR order inside the two passes of the tile kernel
 *                                 (bitwise reproducible); dw [E] is scratch for the spilled edges' shares (needed when dnode_att != NULL).
 * replaces: GSAT.lift_node_att_to_edge_att + its autograd backward fused into PNAConvSimple's aggregation (src/models/pna.py:57-59).
 */
/*
 * PNAConvSimple without the x_i half of the aggregate (src/models/conv_layers.py:148-153 post_nn(aggregate(message))).  The x_i part of
 * the message is the same vector for every in-edge of a row, so its aggregates are a closed form of x_i and four statistics of the row's
 * edge weights; the [N, A*2*H] aggregate is therefore never written: the forward emits the x_j parts and the four scalars, and the post_nn
 * GEMMs rebuild the x_i columns in their operand loader (the arithmetic of gsat_pna_fwd up to the order of two multiplications).
 *   gsat_pna_fwd_compact:  aggj [N, A*H] (aggregator-major), scal [N,8] = (sum a / n, sum a^2 / n, min a, max a, sum a, 0, 0, 0) of the row's
 *                          edge weights, n = max(in-degree, 1), min / max stored as 0 for a row without in-edges.  att NULL: unit weights;
 *                          att [E] with eid; att = node attention [N] with eid NULL (as gsat_pna_fwd_node_att).  Aggregators
 *                          (mean,min,max,std[,sum]), identity scaler, no edge features.
 *   gsat_pna_post_fwd:     out [N, Ho] = Agg W^T + bias, W [Ho, A*2*H] row-major (nn.Linear), split-bf16 x 3 MFMA.  H % 32 == 0.
 *   gsat_pna_post_dw:      dW [Ho, A*2*H] = dout^T Agg, reduced over the rows in ordered slabs (workspace: *_workspace_floats).  H % 64 == 0.
 * The backward's dAgg = dout W stays a plain GEMM into [N, A*2*H] (gsat_gemm_bf16x3) and feeds gsat_pna_bwd_tiled[_node_att].
 */
int gsat_pna_fwd_compact(const float* x, const float* att, const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t num_rows,
                         int64_t H, const int32_t* aggregators, int num_aggregators, float* aggj, float* scal, void* stream);
int gsat_pna_post_fwd(const float* x, const float* aggj, const float* scal, int64_t num_rows, int64_t H,
                      int num_aggregators, const float* W, int64_t ldw, const float* bias, int64_t Ho, float* out, void* stream);
size_t gsat_pna_post_dw_workspace_floats(int64_t num_rows, int64_t H, int num_aggregators, int64_t Ho);
int gsat_pna_post_dw(const float* x, const float* aggj, const float* scal, int64_t num_rows, int64_t H,
                     int num_aggregators, const float* dout, int64_t Ho, float* dW, float* workspace, size_t workspace_floats, void* stream);

int gsat_pna_fwd_node_att(const float* x, const float* node_att, const int32_t* rowptr, const int32_t* col, int64_t num_rows, int64_t H,
                          const int32_t* aggregators, int num_aggregators, const int32_t* scalers, int num_scalers, float avg_deg_lin,
                          float avg_deg_log, float* out, void* stream);
int gsat_pna_bwd_tiled_node_att(const float* x, const float* node_att, const float* dout, const int32_t* rowptr, const int32_t* col,
                                const int32_t* tile_desc, int64_t num_tiles, int rows_nominal, int rows_cap, int edges_cap,
                                const int32_t* rowptr_src, const int32_t* slot_dst_of_srcslot, int64_t num_rows, int64_t num_edges, int64_t H,
                                const int32_t* aggregators, int num_aggregators, const int32_t* scalers, int num_scalers,
                                const int32_t* spill_rows, const int32_t* spill_count, float* dx, float* dmsg, float* dnode_att, float* dw,
                                const float* dx_add, int accumulate_dnode_att /* 1: dnode_att += (the layers of a model share one attention) */,
                                void* stream);

/* ================================ BatchNorm1d over node rows ================================= */

/*
 * y = [relu]( (x - mean) * rstd * gamma + beta ), per channel over the N rows.  training: batch statistics
 * (two-pass mean / biased variance), running_mean / running_var updated in place with `momentum` (running_var
 * takes the unbiased variance, as torch); eval: running statistics.  save_mean / save_rstd [C] feed the backward.
 * replaces: nn.BatchNorm1d in GIN.MLP (src/models/gin.py:55-62) and PyG BatchNorm + F.relu in PNA
 *           (src/models/pna.py:45,57).  workspace: gsat_bn_workspace_floats() floats.  C % 4 == 0.
 */
size_t gsat_bn_workspace_floats(int64_t N, int64_t C);
int gsat_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                int64_t N, int64_t C, int training, float momentum, float eps, int relu, float* y,
                float* save_mean, float* save_rstd, float* workspace, void* stream);
int gsat_bn_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* save_mean,
                const float* save_rstd, int64_t N, int64_t C, int training, int relu, float* dx, float* dgamma,
                float* dbeta, float* workspace, void* stream);

/*
 * The tail of a PNA layer in the same passes:  y = dropout_p( [relu](BN(x)) + residual )
 * replaces: `h = F.relu(batch_norm(conv(...))); x = h + x; x = F.dropout(x, p, training)` (src/models/pna.py:57-59).
 * residual may be NULL; dropout_p = 0 disables dropout (eval).  The keep mask is Philox stream 3, i.e.
 * gsat_philox_keep_mask(seed, 3, N, C, p) -- by value, or read from *seed_dev when that is not NULL (hipGraph replays) --
 * and is recomputed in the backward, which also returns dresidual = dy * keep / (1 - p) (NULL to skip).
 */
int gsat_bn_act_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    int64_t N, int64_t C, int training, float momentum, float eps, int relu, const float* residual,
                    float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* y, float* save_mean,
                    float* save_rstd, float* workspace, void* stream);
int gsat_bn_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* save_mean,
                    const float* save_rstd, int64_t N, int64_t C, int training, int relu, float dropout_p,
                    uint64_t seed, const uint64_t* seed_dev, float* dx, float* dresidual, float* dgamma,
                    float* dbeta, float* workspace, void* stream);

/*
 * BatchNorm over a batch that is sharded across ranks: the same kernels in separate steps, with the two tiny cross-rank
 * reductions left to the caller (torch.distributed all-reduce of [C] vectors), so a sharded run reproduces the single-process
 * statistics at the same global batch:
 *   forward:  S = all_reduce(gsat_bn_local_sum(x, NULL)); mean = S / N_global;
 *             Q = all_reduce(gsat_bn_local_sum(x, mean)); rstd = 1/sqrt(Q / N_global + eps); gsat_bn_apply_fwd(...)
 *   backward: (s1, s2) = all_reduce(gsat_bn_local_bwd_sums(...)); gsat_bn_apply_bwd(..., s1, s2, N_global, ...);
 *             the parameter gradients dbeta / dgamma are the LOCAL s1 / s2 (they are averaged with all other gradients).
 * workspace: gsat_bn_workspace_floats(N, C) floats.  replaces: what torch.nn.SyncBatchNorm would do for src/models/gin.py:58, pna.py:45.
 */
int gsat_bn_local_sum(const float* x, const float* centre, int64_t N, int64_t C, float* out, float* workspace, void* stream);
int gsat_bn_apply_fwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd, int64_t N, int64_t C,
                      int relu, const float* residual, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* y, void* stream);
int gsat_bn_local_bwd_sums(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                           int64_t N, int64_t C, int relu, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* sum_dy,
                           float* sum_dy_xhat, float* workspace, void* stream);
int gsat_bn_apply_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                      const float* sum_dy, const float* sum_dy_xhat, int64_t global_rows, const float* global_rows_dev /* nullable: device
                      float overriding global_rows, so the count reduced across ranks never visits the host */, int64_t N, int64_t C,
                      int relu, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* dx, float* dresidual, void* stream);

/*
 * out[c] = sum_r x[r,c], two-stage fixed-order reduction (bitwise reproducible).  Bias gradients of the Linear layers
 * (replaces the autograd `sum(0)` of nn.Linear in src/models/gin.py:55-62, src/models/conv_layers.py:153-155).
 * workspace: gsat_colsum_workspace_floats(C) floats.
 */
size_t gsat_colsum_workspace_floats(int64_t C);
int gsat_colsum(const float* x, int64_t R, int64_t C, float* out, float* workspace, void* stream);

/* =============================== categorical encoders (ogb) ================================= */

/*
 * out[n,:] = sum_col W[offset(col) + x[n,col], :],  offset(col) = dims[0] + ... + dims[col-1]
 * replaces: ogb AtomEncoder / BondEncoder forward (src/models/gin.py:22-25,45-47; src/models/pna.py:19-22,53-55).
 * x: int64 [N,ncol]; dims: HOST int32[ncol] table sizes; W: the ncol tables concatenated [sum(dims), H].
 * Out-of-range categories are clamped into their table.
 */
int gsat_embsum_fwd(const int64_t* x, const int32_t* dims, int ncol, const float* W, int64_t N, int64_t H,
                    float* out, void* stream);
/*
 * One-hot matrix O [N, R_padded] (R_padded >= sum(dims), multiple of 4): O[n, offset(col)+x[n,col]] = 1.
 * The encoder backward is dW = O^T dout (gsat_gemm_f32 with a_t = 1) instead of a sorted scatter-add.
 */
int gsat_onehot_rows(const int64_t* x, const int32_t* dims, int ncol, int64_t N, int64_t R_padded, float* O,
                     void* stream);

/* ================================ dense fp32 MFMA GEMM ======================================== */

/*
 * C[M,N] = (accumulate ? C : 0) + op(A) op(B) + bias,   exact fp32 (v_mfma_f32_32x32x2_f32).
 *   a_t == 0: A is [M,K] row-major;  a_t != 0: A is [K,M] row-major (reduction over rows, split-K slabs)
 *   b_t != 0: B is [N,K] row-major (nn.Linear weight);  b_t == 0: B is [K,N] row-major
 * replaces: the `addmm` launches of nn.Linear inside the attention MLP (src/utils/get_model.py:61).
 * Contiguous extents and leading dimensions must be multiples of 4; workspace (floats) from
 * gsat_gemm_workspace_floats() is only needed when a_t != 0; bias (nullable, [N]) only when a_t == 0.
 */
size_t gsat_gemm_workspace_floats(int a_t, int64_t M, int64_t N, int64_t K);
int gsat_gemm_f32(int a_t, int b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int accumulate,
                  float* workspace, size_t workspace_floats, void* stream);
/*
 * Same contract, but products of >= 2 GFLOP run as split-bf16: every fp32 operand x = hi + lo (two bf16), the product is
 * hi*hi + hi*lo + lo*hi accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (relative error ~1e-5 of |A||B|, 2-5x faster).
 * For callers whose outputs are not re-normalised by a small per-graph sigma (the backbone's Linear layers).
 */
int gsat_gemm_bf16x3(int a_t, int b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                     const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int accumulate,
                     float* workspace, size_t workspace_floats, void* stream);

/* =============================== attention extractor MLP ===================================== */

/*
 * ExtractorMLP + concrete sampler:
 *   edge mode:  z_e = MLP([emb[src_e] || emb[dst_e]], segment = batch[src_e])   (M = E rows)
 *   node mode:  z_n = MLP(emb[n], segment = batch[n])                            (M = N rows)
 *   MLP = Linear(C0,C1) -> InstanceNorm(per graph) -> ReLU -> Dropout(p)
 *      -> Linear(C1,C2) -> InstanceNorm -> ReLU -> Dropout(p) -> Linear(C2,1)
 *   InstanceNorm: per-graph, per-channel mean, biased variance of the centred values, eps 1e-5,
 *   no affine, batch statistics in train and eval.
 *   att = sigmoid(z + log u - log(1-u)) when `training` and `u` given, else sigmoid(z).
 * replaces: ExtractorMLP.forward (example/gsat.py:131-139; src/run_gsat.py:909-927), MLP /
 *   BatchSequential / InstanceNorm / ReLU / Dropout (src/utils/get_model.py:47-68) and
 *   GSAT.sampling / concrete_sample (example/gsat.py:94-103; src/run_gsat.py:877-885).
 * In edge mode layer 1 is evaluated on NODES (P = emb W1[:, :H]^T, Q = emb W1[:, H:]^T, then
 * h1_e = P[src_e] + Q[dst_e] + b1), so the [E, 2H] concat and the E-row first GEMM never exist.
 * Dropout keep-masks: `mask1/mask2` (float 0/1) if given, else Philox4x32-10 of (seed, layer, row, col).
 */
typedef struct gsat_attn_args {
    int64_t M, N, G;
    int32_t H, C1, C2;
    int32_t edge_mode;
    int32_t training;
    float p_drop;
    uint64_t seed;
    const int32_t* src;        /* [E] int32 edge_index[0]   (edge mode) */
    const int32_t* dst;        /* [E] int32 edge_index[1]   (edge mode) */
    const int32_t* seg_ptr;    /* [G+1] MLP rows grouped by graph */
    const int32_t* seg_order;  /* [M] row ids in grouped order, NULL = identity */
    const int32_t* row_seg;    /* [M] graph id of every row */
    const float *W1, *b1, *W2, *b2, *W3, *b3;   /* nn.Linear layout [out, in] */
    const float* emb;          /* [N,H] */
    const float* mask1;        /* nullable [M,C1] */
    const float* mask2;        /* nullable [M,C2] */
    const float* u;            /* nullable [M] */
    float* P;                  /* [N,C1] saved: edge P ; node h1 (no bias) */
    float* Q;                  /* [N,C1] saved: edge Q ; node: NULL */
    float* a1;                 /* [M,C1] saved: dropout(relu(norm(h1))) */
    float* h2;                 /* [M,C2] saved: a1 W2^T (no bias) */
    float* stats;              /* [G*(2*C1+2*C2)] saved: mean1 | rstd1 | mean2 | rstd2 */
    float* logits;             /* [M] out */
    float* att;                /* [M] out, nullable */
    void* fwd_workspace;       /* gsat_attn_fwd_workspace_bytes() bytes (0 for batches of small graphs) */
    size_t fwd_workspace_bytes;
    const uint64_t* seed_dev;  /* nullable DEVICE word overriding `seed` (hipGraph replays: new dropout mask per replay) */
    int32_t noise_philox;      /* != 0 with `training` and u == NULL: draw the concrete sampler's u in the kernel (Philox stream 4 of
                                  `seed`, row-keyed) instead of reading a tensor -- the reference's uniform_ launch (example/gsat.py:96) */
    int32_t fused;             /* one-launch forward (whole graphs per workgroup, attn_fused.hip): 1 = whenever the shapes allow it, -1 = never,
                                  0 = automatic (GSAT_ATTN_FUSED=1 / 0 overrides): node mode, H % 16 == 0, M >= 32768 rows; its layer products
                                  are split-bf16 x 6 on the bf16 matrix pipe (fp32-level accuracy; GSAT_ATTN_FUSED_X6=0: exact fp32 MFMA) */
    const int32_t* node_ptr;   /* [G+1] node segments of the batch; edge mode needs it for the fused forward (NULL: staged pipeline) */
} gsat_attn_args;

typedef struct gsat_attn_grads {
    const float* dlogits;      /* nullable [M] gradient w.r.t. logits */
    const float* datt;         /* nullable [M] gradient w.r.t. att (needs args->att) */
    const int32_t* rowptr_src; /* edge mode: by-source CSR + edge ids */
    const int32_t* eid_by_src;
    const int32_t* rowptr_dst; /* edge mode: by-destination CSR + edge ids */
    const int32_t* eid_by_dst;
    const int32_t* chunk_ptr_src; /* nullable: long-row chunks of the two CSRs (gsat_row_chunks) */
    const int32_t* chunk_ptr_dst;
    float *demb, *dW1, *db1, *dW2, *db2, *dW3, *db3;
    void* workspace;
    size_t workspace_bytes;
} gsat_attn_grads;

size_t gsat_attn_fwd_workspace_bytes(const gsat_attn_args* args);
/* which forward gsat_attn_fwd will run for these arguments (host-side, no launch): 0 = staged pipeline (layer products exact fp32 MFMA),
   1 = one-launch forward with exact fp32 MFMA, 2 = one-launch forward with split-bf16 x 6 products.  For logs and benchmark lines. */
int gsat_attn_fwd_kind(const gsat_attn_args* args);
size_t gsat_attn_bwd_workspace_bytes(const gsat_attn_args* args);
int gsat_attn_fwd(const gsat_attn_args* args, void* stream);
int gsat_attn_bwd(const gsat_attn_args* args, const gsat_attn_grads* grads, void* stream);

/* Standalone per-graph InstanceNorm (src/utils/get_model.py:50-51,64) for BatchSequential users.
 * y = (x - mean_g) * rstd_g ; stats [G*2*C] = mean | rstd (saved for backward). */
int gsat_instance_norm_fwd(const float* x, const int32_t* seg_ptr, const int32_t* seg_order,
                           const int32_t* row_seg, int64_t M, int64_t G, int64_t C, float* y,
                           float* stats, void* stream);
int gsat_instance_norm_bwd(const float* y, const float* dy, const float* stats, const int32_t* seg_ptr,
                           const int32_t* seg_order, const int32_t* row_seg, int64_t M, int64_t G,
                           int64_t C, float* dx, float* workspace /* [G*2*C] */, void* stream);

/* u[m]: the concrete sampler's noise gsat_attn_fwd draws for row m when noise_philox is set (parity tests pass it back explicitly) */
int gsat_philox_noise(uint64_t seed, int64_t M, float* u, void* stream);

/* keep[m,c] in {0,1}: the Philox dropout mask gsat_attn_* uses for (seed, layer, row m, column c). */
int gsat_philox_keep_mask(uint64_t seed, int32_t layer, int64_t M, int64_t C, float p_drop, float* keep,
                          void* stream);

/* GIN layer tail (src/models/gin.py:49-52: x = F.dropout(relu(conv(x)), p)): y = relu(x) * keep / (1 - p), keep drawn from the Philox
 * stream 5 of `seed` keyed by (row, column) -- or from the device word `seed_dev` (captured steps).  The backward needs y alone:
 * dx = dy / (1 - p) where y > 0, else 0.  C % 4 == 0. */
int gsat_relu_dropout_fwd(const float* x, int64_t N, int64_t C, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* y,
                          void* stream);
int gsat_relu_dropout_bwd(const float* y, const float* dy, int64_t N, int64_t C, float dropout_p, float* dx, void* stream);

/* ============================ samplers, lift, symmetrise, info loss ========================== */

/*
 * att = sigmoid((logits + noise_term) / temp);  mode 0: no noise (eval);
 * mode 1 (concrete): noise_term = log(u) - log(1-u), u = noise[m]   (example/gsat.py:94-103)
 * mode 2 (gumbel):   noise_term = -log(-log(U + eps) + eps)          (src/run_gsat.py:182-187)
 * backward: dlogits = datt * att * (1 - att) / temp.
 */
int gsat_sample_fwd(const float* logits, const float* noise, int mode, float temp, float eps, int64_t M,
                    float* att, void* stream);
int gsat_sample_bwd(const float* att, const float* datt, float temp, int64_t M, float* dlogits, void* stream);

/*
 * edge_att[e] = node_att[src_e] * node_att[dst_e]
 * replaces: lift_node_att_to_edge_att (example/gsat.py:112-117, src/run_gsat.py:870-875).
 * backward walks both CSRs (no atomics): dnode[n] = sum_out d_e a[dst_e] + sum_in d_e a[src_e].
 */
int gsat_lift_fwd(const float* node_att, const int32_t* src, const int32_t* dst, int64_t num_edges,
                  float* edge_att, void* stream);
int gsat_lift_bwd(const float* node_att, const float* dedge_att, const int32_t* rowptr_src,
                  const int32_t* dst_by_src, const int32_t* eid_by_src, const int32_t* rowptr_dst,
                  const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t num_nodes,
                  float* dnode_att, void* stream);

/*
 * out[k] = (att[k] + att[rev[k]]) / 2 ; rev from gsat_reverse_edge_perm (an involution, so the same
 * call maps d(out) to d(att)).  replaces: example/gsat.py:81-83, src/run_gsat.py:233-235,243-245.
 */
int gsat_symmetrise(const float* att, const int32_t* rev, const int32_t* undirected_flag /* nullable device int32: 0 -> out = att */,
                    int64_t num_edges, float* out, void* stream);

/*
 * out[0] = mean_m [ a log(a/r + 1e-6) + (1-a) log((1-a)/(1-r+1e-6) + 1e-6) ], r = r_vec[m] if r_vec
 * else r_scalar.  replaces: example/gsat.py:31; src/run_gsat.py:127,132 (tensor prior).
 * partial: scratch float[1024].  backward: datt[m] = gout[0]/M * d term / d a (r is detached).
 */
int gsat_info_loss_fwd(const float* att, const float* r_vec, float r_scalar, int64_t M, float* partial,
                       float* out, void* stream);
int gsat_info_loss_bwd(const float* att, const float* r_vec, float r_scalar, const float* gout, int64_t M,
                       float* datt, void* stream);

/* out[i] = (int32) in[i] */
int gsat_narrow_i64(const int64_t* in, int64_t n, int32_t* out, void* stream);

/* ================================ device-side batch assembly ================================= */

/*
 * Collate the graphs `graph_ids` of a packed dataset into one batch, on the device.
 * Packed dataset: nodes / edges of all graphs concatenated; node_ptr_all / edge_ptr_all int64[G_all+1];
 * edge_local_all int64[2, num_edges_all] with node ids LOCAL to their graph.
 * out_node_ptr / out_edge_ptr int64[num_graphs+1] = exclusive scans of the selected graphs' node / edge counts.
 * Outputs: batch[N] (graph id per node), node_src_row[N] (row of x_all to copy), edge_index[2,E] (offset to batch ids),
 * edge_src_slot[E] (slot of edge_attr_all / edge_label_all to copy).
 * replaces: PyG DataLoader collation / Batch.from_data_list (src/utils/get_data_loaders.py:130-145) [3P]: node offset of
 * edge_index by the cumulative node count, batch[n] = graph id, graph order = order of graph_ids.
 */
int gsat_collate(const int64_t* graph_ids, int64_t num_graphs, const int64_t* node_ptr_all, const int64_t* edge_ptr_all,
                 const int64_t* edge_local_all, int64_t num_edges_all, const int64_t* out_node_ptr,
                 const int64_t* out_edge_ptr, int64_t N, int64_t E, int64_t* batch, int64_t* node_src_row,
                 int64_t* edge_index, int64_t* edge_src_slot, void* stream);

/*
 * Line ("dual") graph: dual node k = primal directed edge k; dual edges join, in both directions, every two primal
 * edges that leave the same node.  counts[n] = d_n (d_n - 1) / 2; pair_ptr = exclusive scan of counts (int64[N+1]);
 * dual_edge_index int64[2, 2*num_pairs], ordered: source node ascending, pairs (i<j) in edge-id order, (e_i,e_j),(e_j,e_i).
 * replaces: the pure-Python pair loops of src/datasets/mutag_dual.py:345-377 (`group_by_first`).
 */
int gsat_line_graph_pair_counts(const int32_t* rowptr_src, int64_t num_nodes, int64_t* counts, void* stream);
int gsat_line_graph(const int32_t* rowptr_src, const int32_t* eid_by_src, const int64_t* pair_ptr, int64_t num_nodes,
                    int64_t num_pairs, int64_t* dual_edge_index, void* stream);

/*
 * Line graph with one dual node per UNDIRECTED primal edge (the ba_2motifs dual dataset of the fork).
 * Undirected edges are numbered in row-major order of (smaller endpoint, larger endpoint); dual nodes i != j are adjacent
 * when their primal edges share an endpoint; dual_edge_index is in (i, j) row-major order (dense_to_sparse order).
 * replaces: the dense-matrix Python loops of src/datasets/ba_2motifs_dual.py:35-62.
 * Step 1, gsat_und_edges: sorted_keys uint64[E] (src*N+dst ascending), rowptr int32[N+1] over that list, und_of_slot int32[E]
 *   (undirected id of every sorted slot, -1 for self loops), und_of_edge int32[E] (same by ORIGINAL edge id: maps primal edge
 *   attention onto dual nodes), und_src/und_dst int32[>= E/2 + 1, pass E] endpoints a < b of every undirected edge,
 *   status int32[4]: [0] = M (number of undirected edges), [1] = directed edges whose reverse is missing (must be 0: the rule is
 *   defined on symmetric edge sets), [2] = out-of-range ids (clamped).
 * Step 2, gsat_und_line_graph_counts: counts[i] = dual degree of dual node i (the caller scans them into dual_ptr int64[M+1]).
 * Step 3, gsat_und_line_graph: fills dual_edge_index int64[2, total], total = dual_ptr[M].
 */
size_t gsat_und_edges_workspace_bytes(int64_t num_edges);
int gsat_und_edges(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes, uint64_t* sorted_keys, int32_t* rowptr,
                   int32_t* und_of_slot, int32_t* und_of_edge, int32_t* und_src, int32_t* und_dst, int32_t* status,
                   void* workspace, size_t workspace_bytes, void* stream);
int gsat_und_line_graph_counts(const int32_t* rowptr, const int32_t* und_of_slot, const int32_t* und_src, const int32_t* und_dst,
                               int64_t num_und, int64_t* counts, void* stream);
int gsat_und_line_graph(const int32_t* rowptr, const int32_t* und_of_slot, const int32_t* und_src, const int32_t* und_dst,
                        const int64_t* dual_ptr, int64_t num_und, int64_t total, int64_t* dual_edge_index, void* stream);

/* ================================ global pools / segment ops ================================ */

/*
 * out[g,:] = sum (mean != 0: mean, count clamped >= 1) of x[ptr[g]..ptr[g+1],:].
 * replaces: global_add_pool / global_mean_pool (src/models/gin.py:34,53; src/models/pna.py:47,62).
 */
int gsat_segment_pool_fwd(const float* x, const int32_t* ptr, int64_t num_segments, int64_t H,
                          int mean, float* out, void* stream);
int gsat_segment_pool_bwd(const float* dout, const int32_t* ptr, int64_t num_segments, int64_t H,
                          int mean, float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSAT_HIP_H */
