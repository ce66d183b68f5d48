/* plnlp_hip.h -- C ABI of libplnlp_hip.so, the MI355X (gfx950) kernels behind the
 * PLNLP training hot path.
 *
 * The reference (zhitao-wang/PLNLP) has no FFI layer: its hot path is Python
 * calling un-vendored third-party kernels (torch_sparse SpMM, cuBLAS, ATen).
 * Each entry point below replaces one of those call sites; the citation names
 * the reference line whose work it does (paths relative to /root/reference).
 *
 * Contract (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless said otherwise;
 *   - the caller owns every buffer, including workspaces; the library never
 *     allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and is stream-ordered; entry points are re-entrant and
 *     keep no global state that affects results (safe for one-process-per-GPU data parallelism) -- the two
 *     exceptions are measurement aids: the launch counters (plnlp_launch_counts) and plnlp_gemm_stationary_tuning / _block_tuning;
 *   - return value: 0 = enqueued; PLNLP_E_* (negative) = argument rejected,
 *     nothing enqueued; positive = hipError_t reported by the launch;
 *   - matrices are row-major fp32 with an explicit leading dimension (elements);
 *   - graph indices are int32 (col) / int64 (rowptr), see Graph in
 *     plnlp_amd/graph.py.
 */
#ifndef PLNLP_HIP_H
#define PLNLP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 7 (round 3): plnlp_rmat_edges, plnlp_host_randperm_*, plnlp_adam_step_scalars, device-side step scalars in
 *              plnlp_epilogue / plnlp_adam_tensor; the short-rows aggregation form retired.
 * 8 (round 3): plnlp_row_split.seg_* (explicit chunks), PLNLP_AGG_SLABS_XCD / _HUB_XCD / _FUSED_PASSES,
 *              plnlp_pairwise_loss_tail_f32, plnlp_sqnorm_multi_sum_f32, plnlp_edge_endpoints, plnlp_compact_endpoints.
 * 9 (round 4): plnlp_gemm_operand.b_terms / b_terms_bytes + plnlp_gemm_b_terms_bytes (the stationary-weights GEMM);
 *              plnlp_edge_lists_build / _workspace / _supported (the batch's index structures without a library sort).
 * 10 (round 5): plnlp_edge_lists_* removed (the sort-free builder lost its third same-box A/B); plnlp_gemm_stationary_applies (the host asks the library's own rule before it lends b_terms and drops
 *              split-K), plnlp_launch_counts / plnlp_launch_kind_name (which kernel families have been launched),
 *              plnlp_mlp_head_backward_f32 (the 1-output head's backward in one pass over the hidden activation),
 *              PLNLP_EPI_ROWDOT + plnlp_gemm_rowdot_tiles / plnlp_rowdot_finish_f32 (its forward in the hidden GEMM's epilogue).
 * 11 (round 5): plnlp_gemm_wide_wgrad_slices (the weight gradient with whole 224- / 256-wide blocks of the result held by one
 *              workgroup per K slice: the host asks how many slices that form wants and cuts K accordingly).
 * 12 (round 6): plnlp_gemm_block_tuning (the stationary-weights product with a whole 256-row block per workgroup, gemm_x3b.hip);
 *              launch kind "gemm_x3b"; plnlp_gemm_operand.reserved -> flags (PLNLP_GEMM_FLAG_WIDE_WGRAD: the wide weight-gradient
 *              form is asked for explicitly, no longer implied by the slice count); plnlp_gemm_operand.a_colsum (the bias gradient
 *              out of the wide weight-gradient kernel); plnlp_dense_aggregate_f32 / _scratch_bytes (a dense graph's aggregation on
 *              the matrix cores); launch kind "agg_dense"; plnlp_edge_segment_tuning (the segment backward's forms by segment count); plnlp_dense_aggregate_tuning. */
#define PLNLP_ABI_VERSION 12

#define PLNLP_E_NULL      (-1)   /* required pointer is NULL                */
#define PLNLP_E_SHAPE     (-2)   /* negative / inconsistent size            */
#define PLNLP_E_ALIGN     (-3)   /* pointer or leading dimension misaligned */
#define PLNLP_E_UNSUPPORTED (-4) /* flag / size combination not implemented */
#define PLNLP_E_WORKSPACE (-5)   /* workspace too small                     */

int         plnlp_abi_version(void);
const char* plnlp_error_string(int code);

/* ---- epilogue flags shared by aggregate / linear -------------------------- */
#define PLNLP_EPI_BIAS      1u   /* + bias[f]                                         */
#define PLNLP_EPI_RELU      2u   /* max(.,0)                    layer.py:21,25,84     */
#define PLNLP_EPI_DROPOUT   4u   /* counter-RNG dropout(p,seed) layer.py:22,26,85     */
#define PLNLP_EPI_ACCUM     8u   /* out += result instead of out = result             */
#define PLNLP_EPI_GATE     16u   /* result = gate[r,f] > 0 ? result*gate_scale : 0
                                    (backward of relu+dropout given the forward output);
                                    applied LAST, i.e. after PLNLP_EPI_ACCUM / ADDEND */
#define PLNLP_EPI_ADDEND   32u   /* result += addend[a, f], a = addend_index ? addend_index[r] : r,
                                    skipped when a < 0 (a row-sparse term joining a dense result);
                                    applied where PLNLP_EPI_ACCUM is */
#define PLNLP_EPI_ADAM     64u   /* the result is the gradient of the parameter `out`: one Adam step on it instead of
                                    a store (see the adam_* fields; plnlp_csr_aggregate_f32 only) */
#define PLNLP_EPI_ROWDOT 128u   /* also emit per-tile row dot products with rowdot_w (see plnlp_epilogue; the stationary GEMM only) */

typedef struct plnlp_epilogue {
    uint32_t     flags;
    float        dropout_p;      /* PLNLP_EPI_DROPOUT                         */
    uint64_t     dropout_seed;   /* PLNLP_EPI_DROPOUT: per-call 64-bit seed   */
    const float* bias;           /* PLNLP_EPI_BIAS: [n_cols]                  */
    const float* gate;           /* PLNLP_EPI_GATE: [n_rows, ld_gate]         */
    int64_t      ld_gate;
    float        gate_scale;
    const int32_t* gate_index;   /* nullable: the gate row of result row r is gate_index[r] (the result
                                    holds only the rows of a longer matrix listed in gate_index)    */
    const float* addend;         /* PLNLP_EPI_ADDEND: [*, ld_addend]           */
    int64_t      ld_addend;
    const int32_t* addend_index; /* nullable: [n_rows], -1 = no addend row     */
    const int32_t* dropout_row_index; /* nullable: the dropout counter of result row r is taken at row
                                    dropout_row_index[r] (the result holds only SOME rows of the full matrix
                                    and must draw the mask the full matrix would)                     */
    /* PLNLP_EPI_ADAM (plnlp_csr_aggregate_f32, vector path only): the result is the GRADIENT of `out`, which is
     * a parameter -- it is not stored; out, adam_m, adam_v (same leading dimension as out) take one Adam step
     * (torch.optim.Adam, no weight decay, no clipping: the embedding table of plnlp/model.py:163-167) with
     * exactly the arithmetic of plnlp_adam_multi_f32.  Not combinable with PLNLP_EPI_ACCUM.               */
    float*       adam_m;
    float*       adam_v;
    int64_t      adam_step;      /* step count AFTER this update (>= 1): bias corrections 1 - beta^step     */
    float        adam_lr, adam_beta1, adam_beta2, adam_eps;
    /* per-step scalars in DEVICE memory (both nullable): what changes from one training step to the next while
     * every pointer and size stays the same -- so that a step can be captured in a hipGraph once and replayed, its
     * host uploading 40 bytes per step instead of rebuilding kernel arguments.  Values written by the host are used
     * verbatim: a replayed step is bit-identical to the eager one.
     *   dropout_seed_ptr : the 64-bit seed of PLNLP_EPI_DROPOUT (overrides dropout_seed)
     *   adam_scalars     : float[3] = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} as plnlp_adam_step_scalars computes them
     *                      (overrides adam_lr / adam_step)                                                        */
    const uint64_t* dropout_seed_ptr;
    const float*    adam_scalars;
    /* PLNLP_EPI_ROWDOT (plnlp_gemm_f32 on the stationary-weights kernel only, else PLNLP_E_UNSUPPORTED): besides storing the
     * result, the launch writes, per column tile t of the result, the partial row dot products
     *     rowdot_out[t * rowdot_ld + r] = sum_{c in tile t} y[r, c] * rowdot_w[c]        (y = the stored, post-epilogue value)
     * -- MLPPredictor's 1-output last linear (plnlp/layer.py:86) evaluated in the epilogue of the hidden layer's product
     * instead of by a second pass over the [rows, hidden] activation.  plnlp_gemm_rowdot_tiles(m, n) = the number of
     * tiles t (0: the form does not apply to this shape); plnlp_rowdot_finish_f32 adds the tiles in order (+ the bias). */
    const float*    rowdot_w;      /* [n_cols] */
    float*          rowdot_out;    /* [tiles, rowdot_ld] */
    int64_t         rowdot_ld;     /* >= n_rows */
} plnlp_epilogue;

/* ---- K1/K2: CSR neighbour gather-and-reduce --------------------------------
 * out[r, :] = EPI( red_{e in [rowptr[r], rowptr[r+1])} w_e * x[col[e], :] )
 *   w_e = (val ? val[val_index ? val_index[e] : e] : 1) * (src_scale ? src_scale[col[e]] : 1)
 *   reduce = PLNLP_REDUCE_SUM | PLNLP_REDUCE_MEAN (divide by max(rowlen,1))
 * Replaces torch_sparse.matmul(adj_t, x, reduce) reached from SAGEConv /
 * GCNConv (plnlp/layer.py:20,23,36,45) and, run on the transposed CSR, its
 * autograd backward.  Atomic-free, deterministic.
 * `x` and `out` must not alias.  feat % 4 == 0 and 16-byte aligned rows take
 * the vector path; anything else takes a scalar path.
 *
 * Long rows (hubs of a power-law graph) would serialise on one wave; an optional
 * plnlp_row_split moves every row longer than `threshold` out of the main pass:
 * each such row is cut into chunks of `threshold` edges, one wave per chunk writes
 * a partial sum to the workspace, and one wave per long row adds the partials in
 * chunk order (fixed order -> still deterministic).  The tables are built by the
 * caller (once per static graph; per batch, with upper-bound sizes and -1 padding,
 * for the per-step incidence lists): see plnlp_amd/graph.py::RowSplit.
 */
typedef struct plnlp_row_split {
    int64_t        threshold;        /* >= 64                                            */
    int64_t        n_long;           /* entries in long_rows (slots holding -1 are idle)  */
    const int64_t* long_rows;        /* [n_long] row ids                                  */
    const int64_t* chunk_beg;        /* [n_long] first chunk id of each long row          */
    const int32_t* chunk_cnt;        /* [n_long] chunks of each long row                  */
    int64_t        n_chunks;         /* entries in chunk_long                             */
    const int32_t* chunk_long;       /* [n_chunks] slot in long_rows owning the chunk, -1 = idle */
    float*         workspace;        /* [n_chunks, feat], 16-byte aligned                 */
    int64_t        workspace_floats;
    /* optional (all three or none; plnlp_csr_aggregate_f32 only) -- EXPLICIT chunks, processed in table order:
     * chunk c covers entries [seg_beg[c], seg_beg[c] + seg_len[c]) of col / val and adds into workspace slot
     * seg_slot[c] (-1 = idle); the slots of long row l are chunk_beg[l] .. chunk_beg[l] + chunk_cnt[l] - 1 as
     * before, so the partial sums of a row are still added in slot order.  Lets the caller cut hub rows by SOURCE
     * RANGE and order the chunks range-major (plnlp_amd/graph.py::SourceOrderedSplit): waves that run together
     * then gather from the same few thousand source rows.  chunk_long is not read in this form. */
    const int64_t* seg_beg;
    const int32_t* seg_len;
    const int32_t* seg_slot;
} plnlp_row_split;

/* Build the tables above on the device (no host sync): capacities are upper bounds
 * (a CSR with nnz entries has at most nnz/threshold long rows and nnz/threshold + that
 * many chunks); counters[0..2] receive {long rows - 1, chunks - 1, overflow count - 1} (they count up
 * from -1: tables and counters are cleared by one 0xFF fill when the caller packs
 * [counters(4 x i64) | long_rows | chunk_beg | chunk_cnt | chunk_long] back to back). */
int plnlp_row_split_build(const int64_t* rowptr, int64_t n_rows, int64_t threshold,
                          int64_t n_long_cap, int64_t n_chunks_cap,
                          int64_t* long_rows, int64_t* chunk_beg, int32_t* chunk_cnt,
                          int32_t* chunk_long, int64_t* counters /* [4] */, void* stream);

#define PLNLP_REDUCE_SUM  0
#define PLNLP_REDUCE_MEAN 1
/* (flag value 1 is retired: a several-rows-per-wave form for very short rows lost every measurement to the
 * one-row-per-wave form -- rounds 1 and 2 -- and was removed in ABI 7; the bit is ignored) */
#define PLNLP_AGG_LDS_STAGE  2   /* flags: small dense graph -> stage feature slabs of x in LDS (needs
                                    n_src * 16 B <= ~150 KiB, 16-byte aligned rows); gathers then hit LDS */
#define PLNLP_AGG_NT_LOADS   4   /* flags: gathered rows are loaded with the streaming (non-temporal) hint -- for a
                                    source matrix far beyond the 256 MiB Infinity Cache (F = 256 / 512 forms)   */
#define PLNLP_AGG_FEW_IN_FLIGHT 8 /* flags: 4 instead of 8 neighbour rows in flight per wave at F = 512 (fewer
                                    registers -> more waves per SIMD)                                            */
#define PLNLP_AGG_SLABS_128 16    /* flags: one wave per (row, 128-column slab) instead of one per row: more    */
#define PLNLP_AGG_SLABS_256 32    /* (256-column slab) independent gather chains in flight -- for a source matrix
                                     far beyond the caches (HBM-bound); not combined with a dropout epilogue  */
#define PLNLP_AGG_SLABS_XCD 64    /* flags: eight feat/8-column slabs, slab = workgroup id mod 8 -> pinned to the XCD
                                     the workgroup lands on (F = 256 / 512 / 1024): each XCD's L2 sees an eighth of
                                     every source row -- for a source matrix the Infinity Cache holds but one L2
                                     does not; not combined with a dropout epilogue                            */
#define PLNLP_AGG_HUB_XCD 128     /* flags: the same pinned slabs for the LONG rows' chunk pass only (the main pass
                                     keeps one full-width wave per row)                                        */
#define PLNLP_AGG_FUSED_PASSES 256 /* flags: the long rows' chunk pass runs inside the main pass's launch (its workgroups
                                     first), F > 128 full-width forms: same sums, same bits, two launches instead of
                                     three -- the chunks' latency chains hide behind the short rows' traffic  */
int plnlp_csr_aggregate_f32(const int64_t* rowptr, const int32_t* col,
                            const float* val,        /* nullable: [nnz], or indexed through val_index */
                            const int32_t* val_index,/* nullable: [nnz]; weight of entry e = val[val_index[e]] */
                            const float* src_scale,  /* nullable: [n_src]    */
                            const int32_t* src_map,  /* nullable: [n_src]; x holds only SOME source rows: entry e
                                                        reads x[src_map[col[e]]] and is skipped when that is < 0
                                                        (src_scale still indexes col[e]) -- the transposed
                                                        aggregation of a row-sparse gradient                */
                            const int32_t* row_index,/* nullable: [n_rows]; result row r aggregates CSR row row_index[r]
                                                        (only the rows an edge batch reads are produced)          */
                            const int32_t* split_out_map, /* with row_index and split: [CSR rows] result row of a
                                                        long CSR row, < 0 = not produced                           */
                            const float* x, int64_t ldx,
                            float* out, int64_t ldo,
                            int64_t n_rows, int64_t n_src /* rows of x */, int64_t feat, int reduce, int flags,
                            const plnlp_epilogue* epi /* nullable, HOST ptr */,
                            const plnlp_row_split* split /* nullable, HOST ptr */,
                            void* stream);

/* The same aggregation for a DENSE graph, as a product on the matrix cores (csrc/aggregate_dense.hip; replaces torch_sparse.matmul
 * under SAGEConv, plnlp/layer.py:30-36, where the adjacency is dense enough that a matrix of counts beats the CSR gather --
 * ogbl-ddi: 4 267 nodes, 11.7 % of all pairs are edges):
 *   out[r, :] = EPI( row_scale[r] * sum_k counts[r, k] * (src_scale[k] * x[k, :]) )
 * counts: bf16 [n_rows, ld_counts] -- entry (r, k) = the number of CSR entries (r, k), exact up to 256; ld_counts a multiple of
 * 16 >= n_src rounded up to 16, zeros beyond n_src; built once per static graph by the caller.  src_scale (nullable [n_src]): the
 * per-source weight of a valued graph whose values depend on the column only (the mean's backward: 1 / deg); row_scale (nullable
 * [n_rows]): the mean's 1 / max(deg, 1).  x is split into three bf16 terms (the GEMMs' split: what is dropped is below 2^-24 |x|), the
 * counts are exact; f32 accumulation in two levels (four K-steps on the matrix pipe, then plain f32 adds): against float64 the rms
 * error is below the CSR kernels', the sums are the same to f32 round-off in another order.
 * epi: every flag plnlp_csr_aggregate_f32 takes (PLNLP_EPI_ADAM included); operands 16-byte aligned.  scratch: DEVICE memory of
 * plnlp_dense_aggregate_scratch_bytes(n_rows, n_src, feat) bytes the launch may overwrite (x's image + the K slices' partials).
 * feat % 4 == 0; x / out 16-byte aligned with ldx, ldo multiples of 4.  Deterministic (fixed slice order; the number of K slices
 * is a function of the shape: whole rounds of the 512 workgroups the chip holds).
 * plnlp_dense_aggregate_tuning(s): s > 0 forces the slice count (measurement knob, process-global; ask for the scratch size after
 * setting it), 0 = the rule. */
void plnlp_dense_aggregate_tuning(int slices);
int64_t plnlp_dense_aggregate_scratch_bytes(int64_t n_rows, int64_t n_src, int64_t feat);
int plnlp_dense_aggregate_f32(const void* counts, int64_t ld_counts, const float* src_scale, const float* row_scale,
                              const float* x, int64_t ldx, float* out, int64_t ldo, int64_t n_rows, int64_t n_src, int64_t feat,
                              const plnlp_epilogue* epi /* nullable, HOST ptr */, void* scratch, int64_t scratch_bytes, void* stream);

/* reduce = max (torch_sparse.matmul(adj_t, x, reduce='max'); SAGEConv(aggr='max') in PyG, the third
 * reduction SURVEY.md 8(b) lists next to sum / mean):
 *   out[r, f] = max_e w_e * x[col[e], f], 0 for a row without entries;
 *   arg[r, f] = row-relative position of the FIRST entry attaining the maximum, -1 for an empty row
 *               (int32 [n_rows, ld_arg]; what the backward needs).
 * Lane groups fold their candidates with a wavefront-shuffle (max, arg) tree; long rows go through
 * `split` like the sum kernel, with arg_workspace (int32 [n_chunks, feat]) next to split->workspace. */
int plnlp_csr_aggregate_max_f32(const int64_t* rowptr, const int32_t* col,
                                const float* val /* nullable: [nnz] */,
                                const float* x, int64_t ldx, float* out, int64_t ldo,
                                int32_t* arg, int64_t ld_arg, int64_t n_rows, int64_t feat,
                                const plnlp_row_split* split /* nullable, HOST ptr */,
                                int32_t* arg_workspace /* nullable unless split */, void* stream);
/* its backward as a gather over the TRANSPOSED CSR (no atomics, fixed order):
 *   gx[j, f] = sum_{t in [rowptr_t[j], rowptr_t[j+1])} [arg[col_t[t], f] == pos_t[t]] * w_t * gy[col_t[t], f]
 * col_t[t] = the row i fed by source j, pos_t[t] = position of that (i <- j) entry in row i of the
 * forward CSR, val_t[t] = its weight (nullable). */
int plnlp_csr_aggregate_max_bwd_f32(const int64_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                                    const float* val_t, const float* gy, int64_t ldg,
                                    const int32_t* arg, int64_t ld_arg, float* gx, int64_t ldgx,
                                    int64_t n_src, int64_t feat, void* stream);

/* ---- K4: dense fp32 linear on the MFMA (f32-input form or three-term bf16 split: see PLNLP_GEMM_MATH_*) --
 * C[M,N] = EPI( sum_s  op(A_s)[M,K_s] * op(B_s)[K_s,N] )     s = 0 .. n_seg-1 (<= 2)
 *   a_trans = 0: A_s stored [M,K_s] (lda = row stride)   1: stored [K_s,M]
 *   b_trans = 1: B_s stored [N,K_s] (nn.Linear weight)   0: stored [K_s,N]
 * Replaces F.linear / addmm inside SAGEConv (lin_l(agg)+lin_r(x) as ONE
 * concat-K product, n_seg = 2), GCNConv.lin and MLPPredictor.lins
 * (plnlp/layer.py:36,45,83,86) and their autograd dgrad / wgrad GEMMs.
 * split_k > 1: partial products go to `workspace` ([split_k, M, N] floats) and are reduced
 * in fixed order by a second kernel (deterministic); the epilogue runs in that
 * second kernel.
 */
#define PLNLP_GEMM_FLAG_WIDE_WGRAD 1   /* the caller asks for the whole-block weight-gradient form (csrc/gemm_wgw.hip) and has cut K
                                         into plnlp_gemm_wide_wgrad_slices(...) slices: the launch takes the form when BOTH hold.
                                         Without the flag a split_k that merely happens to equal that number runs the 128 x 128
                                         kernels (the choice used to be implicit in the slice count)                 */
typedef struct plnlp_gemm_operand {
    const float* a; int64_t lda;
    const float* b; int64_t ldb;
    int64_t k;
    const int32_t* b_index;   /* nullable, b_trans = 0 and a_trans = 1 only: B's row for reduction index j is
                                 b_index[j] (weight gradient over the rows of a row-sparse dz: B = the
                                 layer input, gathered in place; applies to b2 of plnlp_gemm_concat_b_f32 too) */
    const int32_t* a_index;   /* nullable, a_trans = 0 and b_trans = 1 only: A's row for result row i is
                                 a_index[i] (a layer evaluated only at the rows an edge batch reads: the
                                 root operand x is gathered in the loader)                              */
    int32_t math;             /* PLNLP_GEMM_MATH_* of the launch (read from segs[0])                     */
    int32_t flags;            /* PLNLP_GEMM_FLAG_* (read from segs[0]); 0 = none */
    const int32_t* a_index2;  /* nullable, with a_index (one segment, BF16X3 only): A's row for result row i is the
                                 ELEMENTWISE PRODUCT a[a_index[i], :] * a[a_index2[i], :] -- the Hadamard of the two
                                 endpoint rows of edge i (plnlp/model.py:155-156 + layer.py:81) formed in the loader
                                 of MLPPredictor's first linear instead of being written out and read back       */
    const int32_t* b_index2;  /* nullable, with b_index: likewise for B's row of reduction index j (the weight
                                 gradient of that linear, dz^T (h[src] * h[dst]))                                */
    void* b_terms;            /* nullable (read from segs[0]), BF16X3 and a_trans = 0: DEVICE scratch of at least
                                 plnlp_gemm_b_terms_bytes(m, n, k0, k1) bytes, 16-byte aligned, that the launch may
                                 overwrite: B -- the weights of the layer -- is split into its three bf16 terms ONCE
                                 into this buffer instead of once per row panel, and the product runs on the
                                 stationary-weights kernel (csrc/gemm_x3s.hip).  Same result bits either way; launches
                                 the form does not cover (gathered B rows, unaligned operands, m < 16384) ignore it.   */
    int64_t b_terms_bytes;
    float* a_colsum;          /* nullable (read from segs[0]); only with PLNLP_GEMM_FLAG_WIDE_WGRAD on a launch that takes that form
                                 (else PLNLP_E_UNSUPPORTED): the launch also writes a_colsum[0 .. m) = sum over the reduction
                                 index of A[:, i] -- for dW = dz^T x that is the bias gradient sum_rows dz (the column sums
                                 autograd computes for F.linear's bias, plnlp/layer.py:83; PyG's conv bias) -- from the rows of dz
                                 the kernel stages anyway instead of a second pass over dz.  16-byte aligned; the workspace then
                                 holds split_k * (m * n + m) floats.  Fixed summation order: same bits every launch.          */
} plnlp_gemm_operand;

/* bytes of scratch plnlp_gemm_operand.b_terms needs for an [m, n] result over K-segments k0 (+ k1, 0 = one segment) */
int64_t plnlp_gemm_b_terms_bytes(int64_t m, int64_t n, int64_t k0, int64_t k1);
/* 1 when plnlp_gemm_f32 / _pair_f32 / _split_out_f32 / _concat_b_f32 called with these arguments (and a b_terms buffer of
 * plnlp_gemm_b_terms_bytes) run the stationary-weights kernel, else 0: the library's OWN rule (csrc/gemm_f32.hip::
 * stationary_form), so a host that lends the buffer and leaves K uncut only when this says 1 cannot disagree with the
 * launch (the rule reads segs[].a / lda / k / math / the index pointers, c / ldc / c2 / ldc2 / n_split, m, n; it replaces the
 * host-side mirror of it that `F.linear`'s callers kept, plnlp/layer.py:83,86).  No launch, no device access. */
int plnlp_gemm_stationary_applies(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans, const float* c,
                                  int64_t ldc, int64_t m, int64_t n, const float* c2, int64_t ldc2, int64_t n_split);
/* K slices the WIDE weight-gradient form wants for C[m, n] = A^T B (a_trans = 1, b_trans = 0, one segment, A as stored, BF16X3
 * math, 16-byte aligned operands, k >= 32 768), 0 where the form does not apply.  Whole blocks of the result per workgroup and
 * K slice -- every operand row is read once and split once per block instead of once per 128-wide tile (csrc/gemm_wgw.hip):
 *   128 < m, n <= 224, both multiples of 4: ONE block (7 waves); B one buffer, not gathered;
 *   m and n multiples of 256, at most 8 blocks of 256 x 256 (8 waves): B may be two buffers side by side along n (b2 / ldb2 /
 *     nb_split as in plnlp_gemm_pair_f32, nb_split a multiple of 256; b2 = NULL: one buffer) and its rows may be gathered
 *     (seg->b_index, 16-byte aligned; b_index_on as in plnlp_gemm_pair_f32, 3 for one buffer).
 * A plnlp_gemm_f32 / plnlp_gemm_pair_f32 launch takes the form exactly when segs[0].flags carries PLNLP_GEMM_FLAG_WIDE_WGRAD and its
 * split_k equals this number (and its workspace
 * holds split_k * m * n floats, as for every split-K launch); any other split_k runs the 128 x 128 kernels.  Replaces nothing in
 * the reference (`F.linear`'s weight gradient is one cuBLAS call, plnlp/layer.py:83,86).  No launch, no device access. */
int plnlp_gemm_wide_wgrad_slices(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans, int64_t m, int64_t n,
                                 const float* b2, int64_t ldb2, int64_t nb_split, int b_index_on);
/* column tiles of the stationary-weights kernel's result for an [m, n] product = rows of plnlp_epilogue.rowdot_out that a
 * PLNLP_EPI_ROWDOT launch writes; 0 when the row-dot epilogue is not available for this shape (tile widths 128 / 256 only) */
int plnlp_gemm_rowdot_tiles(int64_t m, int64_t n);
/* out[r] = (bias ? *bias : 0) + sum_t partial[t * ld + r], t = 0 .. tiles-1 in order (the tail of PLNLP_EPI_ROWDOT) */
int plnlp_rowdot_finish_f32(const float* partial, int64_t ld, int tiles, int64_t n_rows, const float* bias /* nullable, DEVICE */,
                            float* out, void* stream);
/* measurement knob of the stationary-weights form (process-global; A/B runs only, not for concurrent launches):
 * nb = 1 / 2 / 4 / 7 / 8 forces the column-tile width (x 32 columns), 0 = automatic; min_rows > 0 changes the number of
 * rows of A from which the form is used at all (default 16 384; the caller must lend b_terms for such launches too) */
void plnlp_gemm_stationary_tuning(int nb, int min_rows);
/* measurement knob of the stationary-weights form's whole-block kernel (csrc/gemm_x3b.hip: one 256-row x 224-column block of the
 * result per workgroup, 8 waves on ONE weight-image stream per CU, plain loads only, one persistent workgroup per CU walking its
 * share of the rows with the K loop pipelined across blocks; taken by the launches of >= 32 768 rows whose column tile is 224 wide
 * -- a layer 193 .. 224 wide: citation2's h = 200 -- same bits as the 128-row kernel).  Process-global, A/B runs only; bits:
 * 1 = never (csrc/gemm_x3s.hip / the tile kernels everywhere), 2 = no leading half blocks, 4 = also the 256-column tiles.
 * Replaces nothing in the reference (`F.linear` is one cuBLAS call, plnlp/layer.py:83,86). */
void plnlp_gemm_block_tuning(int mode);

/* how the products are formed.  Both take and return fp32 and accumulate in fp32:
 *   F32    -- v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain over k (157 TFLOP/s peak)
 *   BF16X3 -- every operand element split in the loader into three bf16 terms (x = hi + mid + lo, residuals
 *             exact), six bf16 MFMAs per product block (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi): each
 *             product reproduced to 2^-22 relative in the worst case (bf16 unit round-off 2^-8: dropped cross terms
 *             and operand residuals of 2^-24 each), measured at or below the f32 MFMA's error against fp64 (which
 *             rounds every product at 2^-24); 16/6 of the f32 MFMA rate.
 * Non-finite operands (a diverged run): F32 follows IEEE like the reference's sgemm -- an inf operand element gives
 * +-inf in the result elements it feeds, NaN where it meets a 0 or an opposite inf, NaN propagates.  BF16X3 turns
 * EVERY result element that depends on a non-finite operand element into NaN (the split forms inf - inf); result
 * elements that do not depend on it are bit-identical to the clean product.  Either way the divergence is visible
 * (tests/test_hip_round3.py::test_gemm_non_finite_operands).  Finite operands up to FLT_MAX are fine in both. */
#define PLNLP_GEMM_MATH_F32    0
#define PLNLP_GEMM_MATH_BF16X3 1

int plnlp_gemm_f32(const plnlp_gemm_operand* segs /* HOST ptr */, int n_seg,
                   int a_trans, int b_trans,
                   float* c, int64_t ldc, int64_t m, int64_t n,
                   const plnlp_epilogue* epi /* nullable, HOST ptr */,
                   int split_k, float* workspace, int64_t workspace_floats,
                   void* stream);

/* same product with the result columns split between two buffers: columns [0, n_split) go to
 * c, columns [n_split, n) to c2 -- one pass over A yields both data gradients of SAGEConv
 * ([gx | gagg] = dz @ [Wr | Wl]) without a strided view to re-pack.  No epilogue, no split-K. */
int plnlp_gemm_split_out_f32(const plnlp_gemm_operand* segs /* HOST ptr */, int n_seg,
                             int a_trans, int b_trans,
                             float* c, int64_t ldc, float* c2, int64_t ldc2, int64_t n_split,
                             int64_t m, int64_t n, const plnlp_epilogue* epi /* must be NULL / empty */,
                             void* stream);

/* B operand concatenated along N from two buffers: columns [0, nb_split) of op(B) come from seg->b,
 * columns [nb_split, n) from b2 (nb_split % 128 == 0).  One pass over A yields both weight
 * gradients of SAGEConv: [dWl | dWr] = dz^T [agg | x] (a_trans = 1, b_trans = 0, split-K). */
int plnlp_gemm_concat_b_f32(const plnlp_gemm_operand* seg /* HOST ptr, one segment */,
                            const float* b2, int64_t ldb2, int64_t nb_split,
                            int a_trans, int b_trans, float* c, int64_t ldc, int64_t m, int64_t n,
                            const plnlp_epilogue* epi, int split_k, float* workspace,
                            int64_t workspace_floats, void* stream);

/* both at once -- B read from two buffers along N, the result written to two buffers along N, optional
 * split-K (the reduce kernel writes the two halves): the two data gradients of SAGEConv without first
 * concatenating the weights ([gx | gagg] = dz [Wr | Wl]: b = Wr, b2 = Wl), and its two weight gradients
 * as two contiguous tensors ([dWl | dWr] = dz^T [agg | x]).  b2 nullable (B is then one buffer); no epilogue. */
int plnlp_gemm_pair_f32(const plnlp_gemm_operand* seg /* HOST ptr, one segment */,
                        const float* b2, int64_t ldb2, int64_t nb_split,
                        int b_index_on /* seg->b_index applies to: 3 = b and b2, 1 = b only, 2 = b2 only */,
                        int a_trans, int b_trans, float* c, int64_t ldc, float* c2, int64_t ldc2,
                        int64_t n_split, int64_t m, int64_t n, int split_k, float* workspace,
                        int64_t workspace_floats, void* stream);

/* column sums: out[f] = sum_r x[r,f] (bias gradients; mean row for eval,
 * plnlp/model.py:193).  workspace: [n_blocks, feat] floats, n_blocks returned by
 * plnlp_colsum_workspace_floats / feat. */
int64_t plnlp_colsum_workspace_floats(int64_t n_rows, int64_t feat);
int plnlp_colsum_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t feat,
                     const float* row_weight /* nullable: out[f] = sum_r row_weight[r]*x[r,f] */,
                     float scale, float* out, float* workspace, int64_t workspace_floats,
                     void* stream);

/* ---- single-output linear head (MLPPredictor's last layer, plnlp/layer.py:86, out_channels = 1) --
 * forward : out[r] = <x[r,:], w> + (bias ? *bias : 0)                 (a GEMM with N = 1 would waste
 *           127/128 of every MFMA tile; this is a bandwidth-bound row reduction)
 * backward: dx[r,:] = EPI( g[r] * w[:] )   (EPI: gate by the previous layer's output)
 *           dw[k]   = sum_r g[r] * x[r,k]  -> plnlp_colsum_f32 with row_weight = g */
int plnlp_matvec_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t feat,
                     const float* w, const float* bias /* nullable, DEVICE scalar */,
                     float* out, void* stream);
int plnlp_outer_f32(const float* g, const float* w, int64_t n_rows, int64_t feat,
                    float* dx, int64_t lddx, const plnlp_epilogue* epi /* nullable: GATE */,
                    void* stream);
/* The whole backward of that head behind a relu / dropout hidden layer in ONE pass over the hidden activation a
 * (plnlp/layer.py:82-86: x = dropout(relu(lin(x))); x = lins[-1](x) -- autograd's four passes over [rows, feat]):
 *   dz[r,f] = a[r,f] > 0 ? g[r] w[f] gate_scale : 0      (= plnlp_outer_f32 with the GATE epilogue, same bits)
 *   sums[0 .. feat)        = dw[f]  = sum_r g[r] a[r,f]  (= plnlp_colsum_f32(a, row_weight = g), same bits)
 *   sums[feat .. 2 feat)   = dbp[f] = sum_r dz[r,f]      (= plnlp_colsum_f32(dz), same bits: the hidden layer's bias gradient)
 * (sums has 2 feat floats; the head's own bias gradient sum_r g[r] stays plnlp_colsum_f32 on the [rows, 1] vector)
 * feat % 4 == 0, feat <= 1024, 16-byte aligned a / dz / w / workspace (else PLNLP_E_UNSUPPORTED / _ALIGN: the caller keeps
 * the separate entry points); workspace: plnlp_mlp_head_backward_workspace_floats(n_rows, feat) floats. */
int64_t plnlp_mlp_head_backward_workspace_floats(int64_t n_rows, int64_t feat);
int plnlp_mlp_head_backward_f32(const float* a, int64_t lda, const float* g, const float* w, float gate_scale,
                                int64_t n_rows, int64_t feat, float* dz, int64_t lddz, float* sums,
                                float* workspace, int64_t workspace_floats, void* stream);

/* ---- K3: edge endpoint gather + score --------------------------------------
 * Replaces h[edge[0]], h[edge[1]] (plnlp/model.py:155-156,179-180) fused with
 * DotPredictor.forward (layer.py:174-176) or the Hadamard product that opens
 * MLPPredictor.forward (layer.py:81).  src/dst are int64 like the reference's
 * edge tensors; h has n_rows rows and a negative index i addresses row n_rows + i
 * (the reference appends a mean row so that -1 means "unseen node", model.py:191-194).
 * plnlp_edge_hadamard_fwd_f32 on a table of 4 .. 24 MB at 256 / 512 columns (beyond one XCD's L2, an eighth of it inside: ogbl-ddi)
 * runs in eight XCD-pinned column slabs (workgroup b: slab b % 8) -- the same products; plnlp_edge_segment_tuning(3) turns that off.
 */
int plnlp_edge_dot_fwd_f32(const float* h, int64_t ldh, int64_t n_rows,
                           const int64_t* src, const int64_t* dst, int64_t n_edges,
                           int64_t feat, float* out, void* stream);
int plnlp_edge_hadamard_fwd_f32(const float* h, int64_t ldh, int64_t n_rows,
                                const int64_t* src, const int64_t* dst, int64_t n_edges,
                                int64_t feat, float* out, int64_t ldo, void* stream);
/* backward of both: gh[src] += g (.) h[dst], gh[dst] += g (.) h[src]
 * (index_put_(accumulate=True) in the reference's autograd).
 *   g_is_vector = 0: g is [n_edges] (dot)   1: g is [n_edges, ldg] (hadamard)
 * Scatter by fp32 atomics: summation order is not fixed. gh must
 * be zero-initialised (or hold a gradient to accumulate into). */
int plnlp_edge_scatter_bwd_f32(const float* h, int64_t ldh,
                               const int64_t* src, const int64_t* dst, int64_t n_edges,
                               int64_t feat, const float* g, int64_t ldg, int g_is_vector,
                               float* gh, int64_t ldgh, void* stream);
/* node-sorted incidence list of an edge batch, built on the device without a host sync
 * (keys-only radix sort of unique (node, item) keys -> a fully determined order):
 *   seg_ptr[n_nodes+1], and per sorted item the batch edge id and the OTHER endpoint.
 * keys_a / keys_b: [2*n_edges] uint64 scratch; temp: plnlp_incidence_temp_bytes(n_edges) bytes. */
int64_t plnlp_incidence_temp_bytes(int64_t n_edges);
int plnlp_incidence_build(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes,
                          uint64_t* keys_a, uint64_t* keys_b, void* temp, int64_t temp_bytes,
                          int32_t* item_edge, int32_t* item_other, int64_t* seg_ptr, void* stream);
/* rows of a CSR that are not empty, in increasing order (the nodes an edge batch touches: only those
 * rows of the gathered matrix receive a gradient, so the encoder's last backward step runs on
 * `count` rows instead of n_rows).  rows[i] = i-th non-empty row, node_map[r] = its position or -1,
 * rowptr_c[i] = rowptr[rows[i]] and rowptr_c[count] = rowptr[n_rows] (the CSR without its empty
 * rows), *count = their number; past the count the lists are padded as empty rows (rows[i] = 0,
 * rowptr_c[i+1] = rowptr[n_rows]) so the compact CSR can be walked over its capacity without the count.
 * Capacity of rows / rowptr_c: n_rows (+1).  block_ws: int32
 * [plnlp_compact_rows_workspace(n_rows)].  Three small launches, deterministic. */
int64_t plnlp_compact_rows_workspace(int64_t n_rows);
int plnlp_compact_rows(const int64_t* rowptr, int64_t n_rows, int32_t* rows, int32_t* node_map,
                       int64_t* rowptr_c, int64_t* count, int32_t* block_ws, void* stream);
/* The endpoint lists of a batch, one launch each (the side stream ran them as 2 cat + 3 gather + 3 cast launches of
 * stock ops per step):
 *   plnlp_edge_endpoints   : src / dst [n_pos + n_neg] = column 0 / 1 of [pos ; neg], both row-major [*, 2] int64 --
 *                            the `pos_edge[:, 0]`, `neg_edge[..., 0]` ... of plnlp/model.py:152-156 as two lists;
 *   plnlp_compact_endpoints: src_c / dst_c = node_map[src / dst] (int64), other_c = node_map[item_other] (int32) --
 *                            the same endpoints as rows of a matrix that holds only the touched nodes
 *                            (plnlp_compact_rows' node_map; every endpoint of the batch is a touched node). */
int plnlp_edge_endpoints(const int64_t* pos, int64_t n_pos, const int64_t* neg, int64_t n_neg,
                         int64_t* src, int64_t* dst, void* stream);
int plnlp_compact_endpoints(const int32_t* node_map, const int64_t* src, const int64_t* dst, int64_t n_edges,
                            const int32_t* item_other, int64_t n_items, int64_t* src_c, int64_t* dst_c,
                            int32_t* other_c, void* stream);
/* uniform random walks for the random-walk pair augmentation (main.py:241-253; replaces
 * torch_cluster.random_walk): walks[w, 0] = start[w], walks[w, l+1] = a uniformly chosen neighbour of
 * walks[w, l] (the node itself if it has none).  Randomness: counter hash of (seed, w*L + l). */
int plnlp_random_walk(const int64_t* rowptr, const int32_t* col, const int64_t* start,
                      int64_t n_walkers, int walk_length, uint64_t seed,
                      int64_t* walks /* [n_walkers, walk_length + 1] */, void* stream);

/* R-MAT (a, b, c, d) edge stream -- the synthetic graph of BASELINE.json config 5 ("scale-26 R-MAT
 * .57/.19/.19/.05, ids mod N, generated on-GPU in chunks"; the reference itself has no generator: it loads OGB
 * files, main.py:74-95).  Edges [edge_lo, edge_lo + n_edges) of the stream defined by (scale, seed, thresholds):
 * bit l of an edge's raw row / column id comes from the counter hash of (seed, edge * 64 + l) compared with the
 * integer thresholds t_a = a * 2^32, t_ab = (a + b) * 2^32, t_abc = (a + b + c) * 2^32; relabel != 0 sends the raw
 * ids through a seeded bijection of [0, 2^scale) first; ids are folded mod n_nodes.  Pure integer arithmetic: any
 * rank replays any part of the stream bit-exactly (a row-sharded run keeps only the edges of its own rows), and
 * oracle/reference_path.py::rmat_edges_ref restates it in numpy. */
int plnlp_rmat_edges(int scale, int64_t n_nodes, int64_t edge_lo, int64_t n_edges, uint64_t seed,
                     uint32_t t_a, uint32_t t_ab, uint32_t t_abc, int relabel,
                     int32_t* rows /* [n_edges] */, int32_t* cols /* [n_edges] */, void* stream);

/* deterministic variant over a node-sorted incidence list built once per batch:
 * for node slot s (seg_node[s] = node id, or s itself when seg_node is NULL), items
 * [seg_ptr[s], seg_ptr[s+1]) each
 * name an edge id and the OTHER endpoint:
 *   gh[seg_node[s], :] = EPI( sum_items g[edge] (.) h[other, :] )
 * Vector rows (feat % 4 == 0, 16-byte aligned): a LONG segment is shared by the waves of a workgroup (one wave per segment
 * makes the launch as long as the hub's serial chain of items) -- from 8 192 segments on, eight segments per workgroup of
 * eight waves: a wave sums its own if it has at most 64 items (item order), the longer ones are summed by all eight waves
 * (contiguous eighths, partial sums added in wave order); below 8 192 segments every segment is shared by the four waves of
 * its own workgroup (a dense graph: ddi).  Deterministic: the association depends on the segment's length only.
 * plnlp_edge_segment_tuning(1): one wave per segment whatever its length (the round-5 form; measurement knob, process-global);
 * 2: groups of four waves / four segments instead of eight / eight; 3: few segments without the XCD-pinned column slabs (below);
 * 0 = the rule above.  Few segments (< 8 192), rows of 256 / 512 floats, a table of more than 4 MB (one XCD's L2): eight workgroups
 * per segment, one eighth of the columns each (workgroup b: slab b % 8 -- consecutive workgroups run on consecutive XCDs, so an
 * XCD's L2 holds one slab of h), the partial sums of the workgroup's lane groups added in group order. */
void plnlp_edge_segment_tuning(int form);

int plnlp_edge_segment_bwd_f32(const float* h, int64_t ldh,
                               const int64_t* seg_ptr, const int64_t* seg_node, int64_t n_seg,
                               const int32_t* item_edge, const int32_t* item_other,
                               int64_t feat, const float* g, int64_t ldg, int g_is_vector,
                               float* gh, int64_t ldgh,
                               const plnlp_epilogue* epi /* nullable: GATE only */,
                               void* stream);

/* ---- K5: pairwise ranking loss, forward + backward in one pass -------------
 * d = pos[b] - neg[b*k + n];  loss = sum (or mean) over the [B,k] grid.
 * Replaces plnlp/loss.py:5-48 + their autograd.  kind: PLNLP_LOSS_*.
 * weight: [B] (weighted / adaptive kinds), nullable otherwise.
 * Outputs: loss[1], gpos[B], gneg[B*k] (d loss / d input, already scaled by
 * grad_scale).  workspace: plnlp_loss_workspace_floats(B) floats; reduction order
 * is fixed (deterministic).
 */
#define PLNLP_LOSS_AUC                 0   /* loss.py:5-8   */
#define PLNLP_LOSS_HINGE_AUC           1   /* loss.py:11-14 */
#define PLNLP_LOSS_WEIGHTED_AUC        2   /* loss.py:17-21 */
#define PLNLP_LOSS_ADAPTIVE_AUC        3   /* loss.py:24-28 */
#define PLNLP_LOSS_WEIGHTED_HINGE_AUC  4   /* loss.py:31-35 */
#define PLNLP_LOSS_ADAPTIVE_HINGE_AUC  5   /* loss.py:38-42 */
#define PLNLP_LOSS_LOG_RANK            6   /* loss.py:45-48 (mean) */
int64_t plnlp_loss_workspace_floats(int64_t batch);
int plnlp_pairwise_loss_f32(int kind, const float* pos, const float* neg, const float* weight,
                            int64_t batch, int64_t num_neg, float grad_scale,
                            float* loss, float* gpos, float* gneg,
                            float* workspace, int64_t workspace_floats, void* stream);
/* The same in ONE launch: the last workgroup to finish adds the partial sums (same order, same bits as the
 * two-launch form) -- a dependent launch of a one-block kernel costs 5-9 us on this part.
 *   block_counter: one DEVICE word, 0 at launch, left 0 (one persistent word per caller and stream); null = the
 *                  two-launch form above;
 *   loss_acc     : nullable DEVICE double: *loss_acc += (double)loss * acc_weight -- the epoch's running sum
 *                  `total_loss += loss.item() * num_examples` (plnlp/model.py:169) without its three element-wise
 *                  launches and without the read-back. */
int plnlp_pairwise_loss_tail_f32(int kind, const float* pos, const float* neg, const float* weight,
                                 int64_t batch, int64_t num_neg, float grad_scale,
                                 float* loss, float* gpos, float* gneg,
                                 float* workspace, int64_t workspace_floats,
                                 unsigned int* block_counter, double* loss_acc, double acc_weight, void* stream);

/* ---- optimiser step (plnlp/model.py:163-167) -------------------------------
 * sum of squares of a gradient tensor into partial[n_partial] (fixed order), so
 * that clip_grad_norm_'s total norm over a parameter group is
 * sqrt(sum over tensors and partials); then one fused Adam update with the clip
 * coefficient read from DEVICE memory (no host sync):
 *   coef = clip_coef ? min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) : 1
 */
int64_t plnlp_sqnorm_partials(int64_t n);
int plnlp_sqnorm_f32(const float* g, int64_t n, float* partial, int64_t n_partial, void* stream);
int plnlp_sum_partials_f32(const float* partial, int64_t n, float* out /* [1] */, int accumulate,
                           void* stream);
int plnlp_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                        int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int decoupled_wd, int64_t step,
                        const float* sqnorm /* nullable DEVICE ptr */, float max_norm,
                        float grad_scale, void* stream);

/* The same two steps over a LIST of tensors in one launch each (a model has a handful of small
 * parameters next to one large one; every dependent launch costs >= 4.5 us whatever it does):
 *   plnlp_sqnorm_multi_f32: partials of all tensors back to back (tensor i gets
 *     plnlp_sqnorm_partials(n_i) entries), to be summed by plnlp_sum_partials_f32;
 *   plnlp_adam_multi_f32: one fused clip + Adam update per listed tensor (each with its own step
 *     count and clip group).  At most PLNLP_MULTI_MAX tensors per call. */
#define PLNLP_MULTI_MAX 16
typedef struct plnlp_adam_tensor {
    float*       param;
    const float* grad;
    float*       exp_avg;
    float*       exp_avg_sq;
    int64_t      n;
    int64_t      step;        /* >= 1 */
    const float* sqnorm;      /* nullable: squared norm of the tensor's clip group (device scalar) */
    float        max_norm;
    const float* step_scalars;/* nullable DEVICE float[3] = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} of this step
                                 (plnlp_adam_step_scalars): overrides `lr` and `step` -- see plnlp_epilogue.adam_scalars */
} plnlp_adam_tensor;
/* HOST helper: out[0..2] = {lr, 1 - beta1^step, sqrt(1 - beta2^step)} with exactly the arithmetic the launchers
 * apply to a by-value step count -- what a caller uploads into adam_scalars / step_scalars */
int plnlp_adam_step_scalars(float lr, float beta1, float beta2, int64_t step, float* out /* HOST [3] */);
int plnlp_sqnorm_multi_f32(const float* const* grads /* HOST array */, const int64_t* sizes /* HOST */,
                           int n_tensors, float* partial, int64_t n_partial, void* stream);
/* ... with the sum of the partials in the same launch (last-workgroup pattern, see plnlp_pairwise_loss_tail_f32):
 * out[0] = sum of this call's partials; block_counter null = plnlp_sqnorm_multi_f32 */
int plnlp_sqnorm_multi_sum_f32(const float* const* grads /* HOST array */, const int64_t* sizes /* HOST */,
                               int n_tensors, float* partial, int64_t n_partial, float* out /* DEVICE [1] */,
                               unsigned int* block_counter, void* stream);
int plnlp_adam_multi_f32(const plnlp_adam_tensor* tensors /* HOST array */, int n_tensors,
                         float lr, float beta1, float beta2, float eps, float weight_decay,
                         int decoupled_wd, float grad_scale, void* stream);
/* grad *= min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) -- clip_grad_norm_ for callers that
 * keep torch.optim as the optimiser */
int plnlp_clip_scale_f32(float* grad, int64_t n, const float* sqnorm, float max_norm, void* stream);

/* ---- HOST side: the DataLoader batch permutation (plnlp/model.py:147), produced incrementally ------------
 * torch.randperm(n, generator=torch.Generator().manual_seed(seed)) on the CPU -- what DataLoader(range(E), B,
 * shuffle=True) permutes with -- is a forward Fisher-Yates shuffle driven by MT19937: after iteration i the entries
 * perm[0..i] are final.  These two functions reproduce it bit for bit in HOST memory (perm: int64 [n], mt_state:
 * uint32 [625], both caller-owned; no GPU work, no stream) and in slices: _init writes the identity and seeds the
 * generator, _advance(from, to) runs iterations [from, to) in order -- after it perm[0 .. to) is final (all of it when
 * to == n).  Calls must cover [0, n) in increasing, contiguous slices.  A trainer shuffles a few batches ahead of the
 * GPU on a host thread instead of running the whole shuffle before the first step.  n < 2^32 / 20 (ATen's own branch
 * for this algorithm); larger n: PLNLP_E_UNSUPPORTED. */
int plnlp_host_randperm_init(uint64_t seed, int64_t n, int64_t* perm /* HOST */, uint32_t* mt_state /* HOST [625] */);
int plnlp_host_randperm_advance(int64_t n, int64_t* perm /* HOST */, uint32_t* mt_state /* HOST [625] */,
                                int64_t from, int64_t to);

/* ---- small element-wise helpers -------------------------------------------- */
/* y = gate>0 ? g*scale : 0  (relu+dropout backward as a stand-alone pass) */
int plnlp_gate_f32(const float* g, const float* gate, float scale, float* y, int64_t n, void* stream);
/* out[r, :] = dropout(x[r, :]) -- stand-alone counter-RNG dropout (tests) */
int plnlp_dropout_f32(const float* x, float* y, int64_t n_rows, int64_t n_cols,
                      float p, uint64_t seed, void* stream);
int plnlp_transpose_f32(const float* x, int64_t ldx, float* y, int64_t ldy,
                        int64_t n_rows, int64_t n_cols, void* stream);

/* ---- diagnostics: which kernel families this process has launched -------------- */
/* Process-wide counters (relaxed atomics; a count, not a synchronisation), one per kernel FAMILY an entry point can choose
 * between: the stationary-weights / tile split-bf16 GEMMs, the f32 tile GEMM, the aggregation's one-wave-per-row, slab,
 * XCD-pinned, fused (main + hub chunks in one launch), chunk, finalize, LDS-staged and scalar forms.  The reference has no
 * counterpart (its kernels are chosen inside torch / torch_sparse); the parity tests use it to assert that a multi-step run
 * went through the forms the benchmark measures (tests/test_hip_round5.py).
 * plnlp_launch_counts copies min(n, kinds) counters into out (nullable) and returns the number of kinds;
 * plnlp_launch_kind_name(i) names kind i (NULL past the end). */
int         plnlp_launch_counts(int64_t* out /* HOST */, int n);
const char* plnlp_launch_kind_name(int kind);

#ifdef __cplusplus
}
#endif
#endif /* PLNLP_HIP_H */
