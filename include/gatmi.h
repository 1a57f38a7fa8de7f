/*
 * gatmi.h — C ABI of libgatmi.so: a graph-attention (GAT) layer for AMD MI355X (gfx950).
 *
 * The reference (zhangtianer521/GAT, TensorFlow 1.x, pure Python) has no FFI of its own; its
 * boundary for this path is the Python functions
 *     utils/layers.py:6      attn_head(seq, out_sz, bias_mat, activation, in_drop, coef_drop, residual)
 *     models/sp_gat.py:13    layers.sp_attn_head(seq, out_sz, adj_mat, activation, nb_nodes, ...)
 *     utils/process.py:14    adj_to_bias(adj, sizes, nhood)
 *     utils/process.py:179   preprocess_adj_bias(adj)
 * The entry points below are what a ctypes binding placed under those functions calls.  Each
 * one names the reference lines whose work it replaces.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions cross the boundary;
 *   - every function returns GATMI_OK (0) or a negative GATMI_E* code; the message for the last
 *     failure on the calling thread is returned by gatmi_last_error();
 *   - the caller owns every buffer; the library allocates no device memory and keeps no pointer
 *     after it returns; scratch is a caller-provided workspace sized by gatmi_*_ws_floats();
 *   - device entry points ENQUEUE on `stream` (a hipStream_t passed as void*) and return without
 *     synchronising; the current device must be the one that owns the buffers;
 *   - feature matrices are row-major fp32 with `hd = H * dp` columns, where dp (the padded
 *     per-head width) is a power of two >= 4 and hd / 4 is a power of two <= 64; padded columns
 *     must be zero in W, a1, a2, bias;
 *   - rowptr/colptr are int64, node ids / edge offsets inside work items are int32 (a single
 *     GPU's shard must have fewer than 2^31 stored edges and nodes).
 *
 * Graph orientation: CSR row i = DESTINATION node i, col = SOURCE ids j, i.e. entry (i, j) of the
 * reference's bias_mat / SparseTensor adjacency (utils/layers.py:16: logits[i][j] = f_1[i] + f_2[j]).
 */
#ifndef GATMI_H
#define GATMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GATMI_OK            0
#define GATMI_EINVAL       -1   /* null pointer, negative size, inconsistent dims            */
#define GATMI_EALIGN       -2   /* device pointer not 16-byte aligned                        */
#define GATMI_EUNSUPPORTED -3   /* (H, dp) combination not built, unknown activation, ...    */
#define GATMI_EHIP         -4   /* HIP launch / runtime error (text in gatmi_last_error)     */
#define GATMI_ERANGE       -5   /* index out of range in host input, or 32-bit overflow      */

#define GATMI_ACT_IDENTITY 0    /* `lambda x: x`, models/sp_gat.py:30 */
#define GATMI_ACT_ELU      1    /* tf.nn.elu,     models/sp_gat.py:9  */
#define GATMI_ACT_SIGMOID  2    /* tf.nn.sigmoid, utils/layers.py:307 (inner-product decoder only) */
#define GATMI_ACT_RELU     3    /* tf.nn.relu, the `activation` of the brain-network models (models/gat.py:191) */

#define GATMI_DTYPE_F32  0      /* storage of a gathered table: fp32 rows ...                      */
#define GATMI_DTYPE_BF16 1      /* ... or bf16 rows (half the gather traffic; arithmetic stays fp32) */

#define GATMI_REDUCE_CONCAT 0   /* tf.concat(attns, -1), models/sp_gat.py:17 */
#define GATMI_REDUCE_MEAN   1   /* tf.add_n(out) / n,    models/sp_gat.py:32 (identity activation only) */

int gatmi_version(void);
const char *gatmi_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Host integer API (pure C++ inside, no GPU needed).  Replaces the adjacency handling of
 * utils/process.py:179-188 downstream of its (indices, values, shape) triple.
 * ---------------------------------------------------------------------------------------- */

/* COO -> CSR, destination-major, each row sorted by source id.
 *   dst[k], src[k] : entry k (for the reference's triple: dst = indices[:,0], src = indices[:,1])
 *   add_self_loops : insert (i,i) for every i that lacks it (process.py:181 already does)
 *   dedup          : merge repeated (dst,src) pairs into one entry (process.py:182 binarises)
 * Outputs: rowptr_out[n+1]; col_out[cap]; perm_out[cap] (input position of each output entry,
 * -1 for an inserted self-loop; may be NULL); *nnz_out = entries written.  cap must be
 * >= nnz + (add_self_loops ? n : 0). */
int gatmi_csr_from_coo(int64_t n, int64_t nnz, const int32_t *dst, const int32_t *src,
                       int add_self_loops, int dedup, int64_t cap,
                       int64_t *rowptr_out, int32_t *col_out, int64_t *perm_out, int64_t *nnz_out);

/* CSR -> CSC of the same matrix: for every source j the destinations i that read it, sorted by
 * i, and perm_out[k] = CSR position of CSC entry k (used to key per-edge dropout). */
int gatmi_csc_from_csr(int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int32_t *col,
                       int64_t *colptr_out, int32_t *rowidx_out, int32_t *perm_out);

/* Work items for the row kernels: rows are cut into pieces of at most `chunk` entries and the
 * pieces are ordered longest first (stable by row id), so that a wave's row groups carry similar
 * lengths and long rows start early.  A row of <= chunk entries is one item with slot = -1; a
 * longer row becomes ceil(len/chunk) items that own consecutive partial-result slots and one
 * record in the split list.  item = {row, first_entry, n_entries, slot} (4 x int32);
 * split record = {row, first_slot, n_slots, 0}.
 * gatmi_work_items_count gives the array sizes; gatmi_work_items fills them. */
int gatmi_work_items_count(int64_t n, const int64_t *ptr, int32_t chunk,
                           int64_t *n_items_out, int64_t *n_split_out, int64_t *n_slots_out);
int gatmi_work_items(int64_t n, const int64_t *ptr, int32_t chunk,
                     int32_t *items_out, int32_t *split_out);

/* Two-phase work items for a row-partitioned shard whose rows keep their owned sources first:
 * entries [ptr[i], cut[i]) of row i read rows this GPU owns, [cut[i], ptr[i+1]) read halo rows that
 * arrive from peers.  items_a cover the first parts (they can be swept while the halo rows are still
 * in flight), items_b the second parts; both lists are ordered longest first.  The last field of an
 * item says what happens to the row's running softmax state after the piece:
 *     -1        the row is finished and written (all its entries were in this piece);
 *     s >= 0    the state is parked in partial-result slot s;
 *     -(s + 2)  the state parked in slot s is taken up first, then the row is finished.
 * A row with one piece in each phase uses the third form (one slot, no merge); rows with more pieces
 * (longer than `chunk` on either side) park every piece and get a split record {row, first_slot,
 * n_slots, 0} for the merge step of gatmi_gat_fwd.  Run gatmi_gat_fwd with items_a and n_split = 0,
 * then, once the halo rows have arrived, with items_b and the split list, both with the same `ws`
 * of gatmi_gat_fwd_ws_floats(n_slots, H, dp) floats. */
int gatmi_work_items_phased_count(int64_t n, const int64_t *ptr, const int64_t *cut, int32_t chunk,
                                  int64_t *n_a_out, int64_t *n_b_out, int64_t *n_split_out, int64_t *n_slots_out);
int gatmi_work_items_phased(int64_t n, const int64_t *ptr, const int64_t *cut, int32_t chunk,
                            int32_t *items_a_out, int32_t *items_b_out, int32_t *split_out);

/* Contiguous 1-D row partition into nparts blocks: part_out[nparts+1], part p owns rows
 * [part_out[p], part_out[p+1]).  balance = 0: equal row counts; 1: equal stored-entry counts. */
int gatmi_partition_rows(int64_t n, int32_t nparts, const int64_t *rowptr, int32_t balance,
                         int64_t *part_out);

/* Halo plan of one partition: its CSR rows [part[rank], part[rank+1]) with GLOBAL source ids in
 * col[0..nnz) are rewritten to local ids: owned j -> j - part[rank]; remote j -> n_own + (rank
 * of j in the sorted list of distinct remote ids).  halo_ids_out (capacity nnz) receives that
 * sorted list (ascending global id, hence grouped by owner), halo_counts_out[nparts] the number
 * of halo rows owned by each peer, *n_halo_out the list length. */
int gatmi_halo_plan(int32_t nparts, int32_t rank, const int64_t *part, int64_t nnz,
                    const int64_t *col_global, int32_t *col_local_out,
                    int64_t *halo_ids_out, int64_t *halo_counts_out, int64_t *n_halo_out);

/* ------------------------------------------------------------------------------------------
 * Device API
 * ---------------------------------------------------------------------------------------- */

/* Feature projection with the logit halves fused: replaces utils/layers.py:11 (seq_fts = X.W,
 * no bias), :14 (f_1 = seq_fts.a1 + b1) and :15 (f_2 = seq_fts.a2 + b2) for all H heads at once.
 *   X [n, f_in] (row stride ldx floats), W [f_in, hd], a1,a2 [hd], b1,b2 [H]
 *   Wh [n, hd], f1,f2 [n, H].  Wh16: optional bf16 gather table for GATMI_DTYPE_BF16 (NULL to skip), written
 *   from the same registers: one record per row, ld16 bf16 elements apart (ld16 >= hd + 2H, a multiple of 8),
 *   [hd bf16 values of Wh, round to nearest even | the row's H fp32 f2 values | padding, left untouched].
 *   W_res/b_res/res_out: optional second projection of the same X (residual branch, utils/layers.py:30),
 *   NULL to skip.
 *   in_keep < 1: dropout of the layer's input (utils/layers.py:8-9, keep probability in_keep) applied to the
 *   tile of X as it is loaded — X' = X * mask / in_keep is never written.  The mask is a counter hash of
 *   (seed, row, column / 4), 16 bits per element; gatmi_proj_bwd_params and gatmi_proj_bwd_input called with
 *   the same in_keep and seed redraw it.  Needs f_in % 4 == 0 and 16-byte aligned rows of X
 *   (GATMI_EUNSUPPORTED otherwise: mask the input outside); in_keep = 1: no mask, any shape. */
int gatmi_proj_fwd(const float *X, int64_t n, int32_t f_in, int64_t ldx,
                   const float *W, const float *a1, const float *a2,
                   const float *b1, const float *b2, int32_t H, int32_t dp,
                   float *Wh, uint16_t *Wh16, int64_t ld16, float *f1, float *f2,
                   const float *W_res, const float *b_res, float *res_out,
                   float in_keep, uint64_t seed, void *stream);

/* Parameter gradients of the projection: dW = X^T . dWh [f_in, hd]; and, when Wh != NULL,
 * da1 = sum_i df1[i,h] * Wh[i,h,:], da2 likewise with df2, db1 = sum_i df1, db2 = sum_i df2
 * (the backward of layers.py:11,14,15).  ws: gatmi_proj_bwd_ws_floats(f_in, H, dp) floats.
 * in_keep, seed: as passed to gatmi_proj_fwd (X stands for the dropped input X' of that call). */
size_t gatmi_proj_bwd_ws_floats(int32_t f_in, int32_t H, int32_t dp);
int gatmi_proj_bwd_params(const float *X, int64_t n, int32_t f_in, int64_t ldx,
                          const float *dWh, const float *Wh, const float *df1, const float *df2,
                          int32_t H, int32_t dp,
                          float *dW, float *da1, float *da2, float *db1, float *db2,
                          float *ws, float in_keep, uint64_t seed, void *stream);

/* Input gradient of the projection: dX [n, f_in] (row stride lddx) = dWh . W^T
 * (+= when accumulate != 0).  in_keep < 1 (with the seed of gatmi_proj_fwd): the gradient with respect to the
 * UNdropped input, i.e. the product times mask / in_keep, applied as the rows are stored. */
int gatmi_proj_bwd_input(const float *dWh, int64_t n, int32_t f_in, const float *W,
                         int32_t H, int32_t dp, float *dX, int64_t lddx, int accumulate,
                         float in_keep, uint64_t seed, void *stream);

/* Parameter gradients of an ordinary dense layer Y = X.W + b applied to every node row: the width-1 convolutions
 * around the brain-network attention step (tf.layers.conv1d(., out_sz, 1), utils/layers.py:180,183,187,223-224;
 * models/gat.py:212,231).  With n in the tens of thousands and widths of 16..384 these are reductions over n, which
 * a general GEMM library runs in a dozen workgroups; here the rows are split over the grid and summed in a fixed order.
 *   X [n, f_in] (row stride ldx floats), dY [n, w] contiguous  ->  dW [f_in, w] = X^T . dY,  db [w] = column sums of dY
 *   (db = NULL to skip).  ws: gatmi_linear_bwd_ws_floats(n, f_in, w) floats, 16-byte aligned. */
size_t gatmi_linear_bwd_ws_floats(int64_t n, int32_t f_in, int32_t w);
int gatmi_linear_bwd_params(const float *X, int64_t n, int32_t f_in, int64_t ldx, const float *dY, int32_t w,
                            float *dW, float *db, float *ws, void *stream);

/* out [w] = column sums of Y [n, w] (row stride ld floats): the gradient of a bias added to every node row
 * (the bias of tf.layers.conv1d, utils/layers.py:180), in a fixed order.  ws: gatmi_column_sums_ws_floats(n, w). */
size_t gatmi_column_sums_ws_floats(int64_t n, int32_t w);
int gatmi_column_sums(const float *Y, int64_t n, int32_t w, int64_t ld, float *out, float *ws, void *stream);

/* Bias and activation of a width-1 convolution over the node rows (tf.layers.conv1d(., w, 1, activation),
 * utils/layers.py:180,183,223-224): Y = act(X + b), X and Y [n, w] with rows ld floats apart, activation =
 * GATMI_ACT_IDENTITY / _ELU / _RELU; and its backward from the OUTPUT: dPre = dY * act'(Y), db [w] = column sums of dPre,
 * in one pass (ws: gatmi_column_sums_ws_floats(n, w) floats). */
int gatmi_bias_act_fwd(const float *X, int64_t n, int32_t w, int64_t ld, const float *b, int32_t activation,
                       float *Y, void *stream);
int gatmi_bias_act_bwd(const float *dY, const float *Y, int64_t n, int32_t w, int64_t ld, int32_t activation,
                       float *dPre, float *db, float *ws, void *stream);

/* The projection for SPARSE node features: X [n, f_in] in CSR (xrowptr [n+1], xcol, xval in DEVICE
 * memory), as the sparse tuple of preprocess_features (utils/process.py:155-162, the second return
 * value) describes it; Cora's features are 98.7 % zeros.  Wh = X . W summed over stored entries
 * in stored order, logit halves as in gatmi_proj_fwd.  No residual projection. */
int gatmi_proj_fwd_csr(const int64_t *xrowptr, const int32_t *xcol, const float *xval, int64_t n, int32_t f_in,
                       const float *W, const float *a1, const float *a2, const float *b1, const float *b2,
                       int32_t H, int32_t dp, float *Wh, float *f1, float *f2, void *stream);

/* preprocess_features (utils/process.py:155-162) on a device CSR matrix: out[k] = val[k] * (1 / sum
 * of the row's stored values), rows whose sum is 0 stay 0 (the reference replaces the infinite
 * reciprocal by 0, :159).  val and out may be the same array. */
int gatmi_csr_row_normalize(int64_t n, const int64_t *rowptr, const float *val, float *out, void *stream);

/* Parameter gradients for sparse features: dW = X^T . dWh walks the CSR of X^T (xt_rowptr [f_in+1],
 * xt_row = node ids, xt_val; fixed summation order); da1, da2, db1, db2 as in gatmi_proj_bwd_params
 * (ws: gatmi_proj_bwd_ws_floats floats, needed only when Wh is given). */
int gatmi_proj_bwd_params_csr(const int64_t *xt_rowptr, const int32_t *xt_row, const float *xt_val,
                              int64_t n, int32_t f_in, const float *dWh, const float *Wh,
                              const float *df1, const float *df2, int32_t H, int32_t dp,
                              float *dW, float *da1, float *da2, float *db1, float *db2,
                              float *ws, void *stream);

/* Fused CSR edge-softmax + neighbour aggregation, forward.  Replaces utils/layers.py:16-17
 * (logits, LeakyReLU, additive mask, softmax), :20 (coefficient dropout), :22 (feature dropout),
 * :24 (coefs . seq_fts), :25 (bias_add), :30 (residual add), :34 (activation) and the head
 * concat / mean of models/sp_gat.py:17,32 — for all H heads.
 *   items/n_items, split/n_split : gatmi_work_items of the CSR rowptr (or a sub-list of them: only
 *       the rows named by the items are computed and written; the merge step runs for the rows in
 *       `split`; see gatmi_work_items_phased for the item encoding); col: CSR source ids
 *   feat [n_src, hd] gathered rows (Wh), row stride ldfeat elements, stored as feat_dtype: GATMI_DTYPE_F32,
 *       or GATMI_DTYPE_BF16 (the records gatmi_proj_fwd / gatmi_to_bf16_records write); f1 [n_dst, H]
 *   the source half of the logits (layers.py:15), one of three forms:
 *       f2 [n_src, H], a table the kernel gathers from (logit_a2 = logit_b2 = NULL), or
 *       fp32 rows, f2 = NULL and logit_a2 [hd] (+ logit_b2 [H], NULL = zeros): the kernel forms
 *       f2_j = <feat_j, logit_a2> + logit_b2 per head from the row it gathers anyway, which saves one
 *       scattered request per entry; gatmi_gat_bwd_csc must be called in the same form, so that both
 *       passes see bit-identical logits; or
 *       bf16 rows, f2 = NULL and logit_a2 = NULL: the exact fp32 f2 values of row j follow its hd bf16
 *       values inside the record (ldfeat >= hd + 2H); gatmi_gat_bwd_csc then takes the f2 table.
 *   bias [hd] or NULL; resid [n_dst, hd] or NULL (per head, added before the activation; with
 *       GATMI_REDUCE_MEAN it is averaged over the heads together with the aggregate and the bias)
 *   out [n_dst, hd] (concat) or [n_dst, dp] (mean)
 *   agg, pos [n_dst, hd], lse, q [n_dst, H]: saved for the backward pass, all NULL for inference
 *       agg = sum_j alpha_ij feat_j (before bias), pos = same sum over entries with positive
 *       pre-LeakyReLU logit, q = sum of alpha over those entries, lse = log sum exp of the row
 *   ws: gatmi_gat_fwd_ws_floats(n_slots, H, dp) floats when n_split > 0
 *   coef_keep / feat_keep in (0,1]: keep probabilities of layers.py:20 / :22 (1 = no dropout);
 *   masks are a counter-based hash of (seed, CSR entry position, head) / (seed, src, column); the feature
 *   mask draws one hash per four consecutive columns (16 bits each: keep probabilities resolve to 2^-16). */
size_t gatmi_gat_fwd_ws_floats(int64_t n_slots, int32_t H, int32_t dp);
int gatmi_gat_fwd(int64_t n_dst, int32_t H, int32_t dp,
                  const int32_t *items, int64_t n_items, const int32_t *split, int64_t n_split,
                  const int32_t *col, const void *feat, int32_t feat_dtype, int64_t ldfeat, const float *f1, const float *f2,
                  const float *logit_a2, const float *logit_b2,
                  const float *bias, const float *resid, int32_t activation, int32_t reduce,
                  float slope, float coef_keep, float feat_keep, uint64_t seed,
                  float *out, float *agg, float *pos, float *lse, float *q,
                  float *ws, void *stream);

/* The same forward pass for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses with a no-allocation rule) and therefore cannot bring the
 * workspace of gatmi_gat_fwd.  It has NO counterpart in the reference.  The partial results of split
 * rows go to a fixed store of GATMI_NOALLOC_SCRATCH_BYTES inside the library's code object (one copy
 * per GPU, held until the process ends) instead of `ws`; everything else, and every result bit, is as
 * in gatmi_gat_fwd.  n_slots is the slot count gatmi_work_items_count (or, for two-phase items,
 * gatmi_work_items_phased_count, whose parked rows take slots without a split record) reported for
 * these items (0 when no row is split or parked); the call fails with GATMI_ERANGE when
 * gatmi_gat_fwd_ws_floats(n_slots, H, dp) * 4 exceeds the store (13 797 slots at H = 8, dp = 8).
 * The function calls no hipMalloc / hipFree and no allocating routine.  Its first call on a device
 * looks up that device's address of the store (hipGetSymbolAddress): make that first call outside
 * stream capture.  All calls on one GPU share the store: calls that overlap in time on one GPU
 * (different streams, or threads) race on it; calls ordered on one stream do not. */
#define GATMI_NOALLOC_SCRATCH_BYTES (8u << 20)
int gatmi_gat_fwd_noalloc(int64_t n_dst, int32_t H, int32_t dp,
                          const int32_t *items, int64_t n_items, const int32_t *split, int64_t n_split,
                          int64_t n_slots,
                          const int32_t *col, const void *feat, int32_t feat_dtype, int64_t ldfeat, const float *f1, const float *f2,
                          const float *logit_a2, const float *logit_b2,
                          const float *bias, const float *resid, int32_t activation, int32_t reduce,
                          float slope, float coef_keep, float feat_keep, uint64_t seed,
                          float *out, float *agg, float *pos, float *lse, float *q, void *stream);

/* Dropout under HIP-graph replay.  NO counterpart in the reference (TensorFlow draws its masks inside
 * the session, utils/layers.py:9,20,22).  Every entry point takes its dropout seed by value, and a HIP
 * graph freezes kernel arguments at capture: a captured training step would replay one set of masks.
 * After gatmi_dropout_step_source(word), every launch with dropout switched on (a keep probability
 * below 1) that is issued on the CURRENT device, from any thread, also carries `word` — the address of
 * a 64-bit integer in that device's memory, 8-byte aligned — and its kernels read it when they run:
 * they use seed + *word * gatmi_dropout_step_multiplier() (mod 2^64) in place of seed.  Bump the word
 * on the stream before each step (inside the captured region, so that every replay does) and each
 * step draws fresh masks; forward and backward launches of one step must see the same value.
 * word = NULL ends it for later launches; launches already captured keep reading the word they were
 * captured with, which must stay allocated for as long as they can run.  With no word registered the
 * masks are exactly those of the seed passed (bit-identical to a library without this call).
 * The registration is DEVICE-WIDE (one word per device for the whole process: autograd issues the
 * backward launches from a thread of its own): while it stands, dropout launches from every thread and
 * stream on that device carry the word.  gatmi_dropout_step_source_current() returns the word
 * registered for the current device (NULL: none), so that a scope can put back what it found. */
int gatmi_dropout_step_source(const uint64_t *device_word);
const uint64_t *gatmi_dropout_step_source_current(void);
uint64_t gatmi_dropout_step_multiplier(void);

/* Backward, per-destination part (no gather).  From dOut and the saved forward tensors:
 *   g [n_dst, hd]    = dOut * activation'(out)              (dOut / H per head for mean), row stride ldg floats;
 *       g16: the same rows in bf16 (row stride ldg16 bf16 elements) for gatmi_gat_bwd_csc's gather; either of
 *       g / g16 may be NULL
 *   stats [n_dst, 3, H] = {f1, lse, c}, c = <g, agg> per head, row stride lds floats.  g and stats may
 *       interleave in one buffer of records [g hd | f1 H | lse H | c H | pad] (stats = g + hd, ldg = lds >= hd + 3H;
 *       bf16: stats right behind the hd bf16 values of g16, ldg16 = 2 lds): the two gathers per entry of
 *       gatmi_gat_bwd_csc then read one contiguous record.  With H % 4 == 0 the kernel writes such records,
 *       padding included (zeros), as whole 16-byte pieces: pad them to whole 128-byte lines.
 *   df1 [n_dst, H]   = (1 - slope) * (<g, pos> - c * q)     = sum_j dlogit_ij  (gradient of f_1)
 *   dbias [hd]       = sum_i g (NULL to skip);  dresid is g itself.
 *   activation': ELU / ReLU need the layer's pre-activation.  Pass `out` (the saved output; required
 *       when a residual was added) -- or out = NULL and the layer's `bias` [hd] (NULL = zeros): without
 *       a residual the pre-activation is agg + bias, which saves reading `out` (a fifth of the traffic).
 *   ws: gatmi_gat_bwd_node_ws_floats(H, dp). */
size_t gatmi_gat_bwd_node_ws_floats(int32_t H, int32_t dp);
int gatmi_gat_bwd_node(int64_t n_dst, int32_t H, int32_t dp,
                       const float *dout, const float *out, const float *agg, const float *pos,
                       const float *q, const float *f1, const float *lse, const float *bias,
                       int32_t activation, int32_t reduce, float slope,
                       float *g, int64_t ldg, uint16_t *g16, int64_t ldg16, float *stats, int64_t lds, float *df1, float *dbias,
                       float *ws, void *stream);

/* Backward, per-source part: transposed SpMM + SDDMM over the CSC structure.
 *   items/split: gatmi_work_items of colptr; rowidx: CSC destination ids; perm: CSR position of
 *   each CSC entry (only read when coef_keep < 1, may be NULL otherwise)
 *   g [n_dst, hd] (fp32 or bf16; row stride ldg elements) and stats [n_dst, 3, H] (row stride lds floats)
 *       from gatmi_gat_bwd_node
 *   dfeat [n_src, hd] = sum_i alpha_ij g_i  (+ df1_j a1 + df2_j a2 when a1 != NULL: then dfeat
 *       is the full gradient of Wh = seq_fts and df1 [n_src, H] must be given)
 *   df2 [n_src, H]    = sum_i dlogit_ij
 *   f2 / logit_a2 / logit_b2: the source half of the logits in the form the forward call used (table, or
 *       formed from the rows of feat; with g_dtype GATMI_DTYPE_BF16 from those rows rounded to bf16, which
 *       is what a forward pass over the bf16 table saw). */
size_t gatmi_gat_bwd_ws_floats(int64_t n_slots, int32_t H, int32_t dp);
int gatmi_gat_bwd_csc(int64_t n_src, int32_t H, int32_t dp,
                      const int32_t *items, int64_t n_items, const int32_t *split, int64_t n_split,
                      const int32_t *rowidx, const int32_t *perm,
                      const void *g, int32_t g_dtype, int64_t ldg, const float *stats, int64_t lds, const float *feat, const float *f2,
                      const float *logit_a2, const float *logit_b2,
                      const float *df1, const float *a1, const float *a2,
                      float slope, float coef_keep, float feat_keep, uint64_t seed,
                      float *dfeat, float *df2, float *ws, void *stream);

/* fp32 -> bf16 (round to nearest even) copy of a table, n_elems a multiple of 4. */
int gatmi_to_bf16(const float *src, int64_t n_elems, uint16_t *dst, void *stream);

/* The bf16 gather records of gatmi_gat_fwd from finished fp32 rows (what gatmi_proj_fwd's Wh16 output writes
 * from its epilogue): row r of dst = [src[r, 0..hd) as bf16 | f2[r, 0..H) as fp32 | padding, untouched],
 * rows ld16 bf16 elements apart (ld16 >= hd + 2H, a multiple of 4). */
int gatmi_to_bf16_records(const float *src, const float *f2, int64_t n, int32_t H, int32_t dp, int64_t ld16,
                          uint16_t *dst, void *stream);

/* Last step of the backward pass for owned rows when gatmi_gat_bwd_csc ran with a1 == NULL
 * (row-partitioned runs add the peers' halo contributions to dfeat / df2 first):
 *   dfeat[j,:] += df1[j,h] * a1 + df2[j,h] * a2   — the gradient of layers.py:14-15 w.r.t. seq_fts. */
int gatmi_gat_bwd_finalize(int64_t n, int32_t H, int32_t dp, const float *df1, const float *df2,
                           const float *a1, const float *a2, float *dfeat, void *stream);

/* gatmi_halo_unpack_sum on dfeat and df2 followed by gatmi_gat_bwd_finalize, in one pass over the owned
 * rows (no counterpart in the reference: row-partitioned runs): owned row r first takes in the rows
 * recv_dfeat[pos[k], :] / recv_df2[pos[k], :], k in [rptr[r], rptr[r+1]), that the reverse halo exchange
 * returned for it (list order: the sums are those of gatmi_halo_unpack_sum), then the logit terms.
 *   rptr [n + 1] over ALL owned rows (rows nobody read have an empty range); df2 is updated in place. */
int gatmi_gat_bwd_finalize_halo(int64_t n, int32_t H, int32_t dp, const float *df1, float *df2,
                                const float *a1, const float *a2, float *dfeat,
                                const int64_t *rptr, const int64_t *pos, const float *recv_dfeat, const float *recv_df2,
                                void *stream);

/* Batched dense masked attention (small graphs).  Replaces utils/layers.py:16-25,34 on
 * [B, N, *] tensors with the additive bias_mat [B, N, N] of utils/process.py:14-25 (any additive
 * values, not only 0 / -1e9).
 *   Wh [B,N,hd], f1,f2 [B,N,H], bias [hd] or NULL, resid [B,N,hd] or NULL; out [B,N,hd];
 *   agg, pos [B,N,hd], lse, q [B,N,H]: saved for backward (meaning as in gatmi_gat_fwd), or all NULL.
 *   coef_keep / feat_keep / seed: dropout of layers.py:20,22 as in gatmi_gat_fwd; the masks are keyed
 *   on (graph, i, j, head) and (graph, j, column); dropout needs the saved tensors (training). */
int gatmi_dense_attn_fwd(int64_t B, int32_t N, int32_t H, int32_t dp,
                         const float *Wh, const float *f1, const float *f2, const float *bias_mat,
                         const float *bias, const float *resid, int32_t activation, float slope,
                         float coef_keep, float feat_keep, uint64_t seed,
                         float *out, float *agg, float *pos, float *lse, float *q, void *stream);

/* Backward of the above, per-source part.  g, stats and df1 come from gatmi_gat_bwd_node run on
 * the B*N rows.  dWh [B,N,hd] = sum_i alpha_ij g_i (+ df1_j a1 + df2_j a2 when a1 != NULL),
 * df2 [B,N,H] = sum_i dlogit_ij. */
int gatmi_dense_attn_bwd(int64_t B, int32_t N, int32_t H, int32_t dp,
                         const float *g, const float *stats, const float *Wh, const float *f2,
                         const float *bias_mat, const float *df1, const float *a1, const float *a2,
                         float slope, float coef_keep, float feat_keep, uint64_t seed,
                         float *dWh, float *df2, void *stream);

/* The same two operations for SPARSE masks (same results): instead of sweeping all N x N entries
 * through the matrix cores, the forward kernel reads the mask rows once, compacts the column indices
 * that can contribute into LDS and visits only those.  It visits (i, j) iff
 * bias_mat[b,i,j] > max_j' bias_mat[b,i,j'] - 1e6; every skipped entry has a softmax coefficient of
 * exactly 0 in fp32 as long as the logits stay below 4e5 in magnitude, so adj_to_bias masks
 * (0 / -1e9, utils/process.py:25), soft biases and rows that are masked everywhere (all entries
 * kept) give what the dense kernels give.  `present` [B*N, gatmi_dense_mask_words(N)] uint32 (may be
 * NULL when agg..q are NULL) receives one bit per visited entry plus a flag word per row; the
 * backward kernel walks those bits instead of reading the fp32 mask again (it fetches bias values
 * only in rows whose visited entries are not all zero).  Faster than the dense kernels when about a
 * third or fewer of the entries of a row contribute — and, for batches of at least 256 graphs of 4..320
 * nodes (N % 4 == 0) with rows of 64 floats in heads of >= 8, whose graphs the kernels keep resident in
 * LDS, at every density of a hard mask (equal on a full one); H*dp >= 32, N <= 32767. */
size_t gatmi_dense_mask_words(int32_t N);
int gatmi_dense_attn_fwd_masked(int64_t B, int32_t N, int32_t H, int32_t dp,
                                const float *Wh, const float *f1, const float *f2, const float *bias_mat,
                                const float *bias, const float *resid, int32_t activation, float slope,
                                float coef_keep, float feat_keep, uint64_t seed,
                                float *out, float *agg, float *pos, float *lse, float *q,
                                uint32_t *present, void *stream);
int gatmi_dense_attn_bwd_masked(int64_t B, int32_t N, int32_t H, int32_t dp,
                                const float *g, const float *stats, const float *Wh, const float *f2,
                                const float *bias_mat, const float *df1, const float *a1, const float *a2,
                                const uint32_t *present,
                                float slope, float coef_keep, float feat_keep, uint64_t seed,
                                float *dWh, float *df2, void *stream);

/* gatmi_gat_bwd_node and gatmi_dense_attn_bwd_masked in one launch, for the shapes the graph-resident list
 * kernels cover (batches of at least 256 graphs of 4..320 nodes, N % 4 == 0, rows of 64 floats, heads of at
 * least 8; dropout of utils/layers.py:20,22 included: gatmi_dense_attn_bwd_fused_ok says 1): the block that owns
 * a graph forms g, the row statistics and df1 of its nodes while it stages them, so they make no round trip
 * through memory and one launch per step goes away.  Arguments as in those two calls; g may be NULL (it is only
 * needed by a residual branch), dbias may be NULL, ws: gatmi_dense_attn_bwd_fused_ws_floats(B, H, dp) floats when
 * dbias or da1.. are given.  The structure of the mask comes EITHER from bias_mat + the forward pass's `present` bits
 * (cptr = crow = cord = NULL) OR from prebuilt lists (cptr, crow, cord, max_entries from the calls below;
 * bias_mat and present may then be NULL; the shape is then covered whenever gatmi_dense_lists_capacity(N, H, dp, 1)
 * is not 0).  Other shapes: GATMI_EUNSUPPORTED (use the two calls).
 * da1, da2 [H*dp], db1, db2 [H] (all four or none; need a1, a2 and ws): the gradients of the attention vectors and
 * their scalar biases (utils/layers.py:14-15), da1 = sum_j df1_j Wh_j, da2 = sum_j df2_j Wh_j, db = sum_j df_j, summed
 * by the block that holds the graph's Wh rows and df anyway, so that gatmi_proj_bwd_params need not read Wh again
 * (call it with Wh = df1 = df2 = da1.. = NULL for dW alone).  Only where gatmi_dense_attn_bwd_fused_param_grads says 1
 * (prebuilt lists, at least two heads); elsewhere pass NULL. */
int gatmi_dense_attn_bwd_fused_ok(int64_t B, int32_t N, int32_t H, int32_t dp, float coef_keep, float feat_keep);
int gatmi_dense_attn_bwd_fused_param_grads(int32_t N, int32_t H, int32_t dp, int32_t lists);
size_t gatmi_dense_attn_bwd_fused_ws_floats(int64_t B, int32_t H, int32_t dp);
int gatmi_dense_attn_bwd_fused(int64_t B, int32_t N, int32_t H, int32_t dp,
                               const float *dout, const float *out, const float *agg, const float *pos,
                               const float *q, const float *f1, const float *lse, const float *bias, int32_t activation,
                               const float *Wh, const float *f2, const float *bias_mat, const uint32_t *present,
                               const int32_t *cptr, const uint16_t *crow, const uint16_t *cord, int64_t max_entries,
                               const float *a1, const float *a2, float slope,
                               float coef_keep, float feat_keep, uint64_t seed,
                               float *g, float *df1, float *dbias,
                               float *da1, float *da2, float *db1, float *db2, float *ws,
                               float *dWh, float *df2, void *stream);

/* A mask that stays the same step after step (the reference feeds one bias_mat every epoch,
 * execute_cora.py:100-107; utils/layers.py:16-24 only ever adds it to the logits) need not be read and compacted
 * again: its structure can be turned into lists once, and the graph-resident kernels then do nothing but walk them
 * (no mask read, no compaction; rows are taken in the order of their list lengths so that the eight rows a wave
 * walks together finish together).  For hard masks only: every visited entry of a row carries the row's maximum,
 * so the bias cancels in the softmax (adj_to_bias masks, utils/process.py:14-25; gatmi_dense_mask_bits reports
 * rows for which this does not hold in their flag word, and such masks stay with gatmi_dense_attn_fwd_masked).
 *   gatmi_dense_mask_bits     bits [B*N, gatmi_dense_mask_words(N)]: bit j of row (b, i) = entry (i, j) is within
 *                             1e6 of its row's maximum (the entries gatmi_dense_attn_fwd_masked visits); last
 *                             word of a row = 1 when some visited entry differs from the row's maximum, or when
 *                             several entries share a maximum beyond shared_limit in magnitude (fp32 then rounds
 *                             the logits added to it: a row masked everywhere attends uniformly)
 *   gatmi_dense_lists_count   row_cnt, col_cnt [B*N] int32: list lengths (visited columns of a row / visiting rows
 *                             of a column), padded to whole batches of four
 *   (caller)                  lptr, cptr [B*N + 1] int32 = exclusive prefix sums of the counts; lcol, crow: that
 *                             many uint16, 8-byte aligned; lord / cord [B*N] uint16: per graph, its rows /
 *                             columns ordered by list length, longest first; max_entries: the largest number of
 *                             list entries one graph holds (forward: rows; backward: columns)
 *   gatmi_dense_lists_fill    lcol, crow: the lists, padded with the id N, in the order in which the compacting
 *                             kernels build theirs (results are bit-identical to the mask-reading kernels)
 *   gatmi_dense_lists_capacity  list entries one graph may hold for this shape (backward != 0: for
 *                             gatmi_dense_attn_bwd_fused); 0: shape not covered.  The forward kernel and, with at
 *                             least two heads, the backward kernel read their lists from memory, so every list fits
 *                             ((N + 4) * N); one head of 64 floats in the backward pass: what fits beside the graph's
 *                             rows in LDS
 *   gatmi_dense_attn_fwd_lists  gatmi_dense_attn_fwd_masked on the lists; GATMI_ERANGE when max_entries exceeds
 *                             the capacity.  A caller that understates max_entries gets wrong rows for the graphs
 *                             that exceed it, never an out-of-bounds access.
 * Shapes: those of the graph-resident kernels (see gatmi_dense_attn_bwd_fused), except that N need not be a multiple
 * of 4 (that is what the 16-byte mask loads of the mask-reading kernels ask for; the reference's atlases have 246 and
 * 90 nodes, Data_processing.py:46,518): at least 256 graphs of 4..320 nodes, rows of 64 floats in heads of >= 8
 * (one head of 64 floats: N % 4 == 0 for the backward pass). */
int gatmi_dense_mask_bits(int64_t B, int32_t N, const float *bias_mat, float shared_limit, uint32_t *bits, void *stream);
int gatmi_dense_lists_count(int64_t B, int32_t N, const uint32_t *bits, int32_t *row_cnt, int32_t *col_cnt, void *stream);
int gatmi_dense_lists_fill(int64_t B, int32_t N, const uint32_t *bits, const int32_t *lptr, const int32_t *cptr,
                           uint16_t *lcol, uint16_t *crow, void *stream);
int64_t gatmi_dense_lists_capacity(int32_t N, int32_t H, int32_t dp, int32_t backward);
int gatmi_dense_attn_fwd_lists(int64_t B, int32_t N, int32_t H, int32_t dp,
                               const float *Wh, const float *f1, const float *f2,
                               const int32_t *lptr, const uint16_t *lcol, const uint16_t *lord, int64_t max_entries,
                               const float *bias, const float *resid, int32_t activation, float slope,
                               float coef_keep, float feat_keep, uint64_t seed,
                               float *out, float *agg, float *pos, float *lse, float *q, void *stream);

/* Masked softmax cross-entropy, its gradient and the masked accuracy in one sweep over the logits
 * (models/base_gattn.py:140-146 masked_softmax_cross_entropy, :158-165 masked_accuracy; called at
 * execute_cora.py:74-75):
 *     loss     = mean_i(CE_i * mask_i / mean(mask)),  CE_i = -sum_c labels[i,c] * log_softmax(logits[i,:])[c]
 *     accuracy = mean_i([argmax logits[i,:] == argmax labels[i,:]] * mask_i / mean(mask))
 *     dlogits[i,c] = mask_i / sum(mask) * (softmax(logits[i,:])[c] * sum_c' labels[i,c'] - labels[i,c])
 *   logits [n, C] with row stride ld_logits floats (the layer's padded rows are fine: only C columns are read),
 *   labels [n, C] one-hot or soft, row stride ld_labels; mask [n] floats or NULL (all ones); C <= 256.
 *   dlogits (may be NULL) has the layout of logits; its columns C .. ld_logits - 1 are set to zero, so it can be
 *   handed to the layer's backward pass as it is (ld_logits at most the padded width of C).
 *   out3 = {loss, accuracy, sum(mask)} in DEVICE memory; ws: gatmi_masked_ce_ws_floats() floats.
 * All sums run in a fixed order.  gatmi_scale_by_device_scalar multiplies x[0..n) (n % 4 == 0) by *scale, a float
 * in device memory, and does nothing when it is exactly 1: the upstream gradient of a loss. */
size_t gatmi_masked_ce_ws_floats(void);
int gatmi_masked_ce(int64_t n, int32_t C, const float *logits, int64_t ld_logits, const float *labels, int64_t ld_labels,
                    const float *mask, float *dlogits, float *out3, float *ws, void *stream);
int gatmi_scale_by_device_scalar(float *x, int64_t n, const float *scale, void *stream);

/* Brain-network variant of the attention step: the inner loop body of attn_e2v_BNF_V2
 * (utils/layers.py:174-179) with the coefficients of attn_v2e_BNF_V2 (:229-237), for H heads that
 * share X and the weighted adjacency A but own their projection (f1, f2) and the scalars m1, m2
 * (stimulate_mag_1 / _2):
 *     out[b,i,h,:] = sum_j (m2[h]*[A_ij > 0] + A_ij) * (m1[h] + softmax_j(LeakyReLU(f1_i + f2_j))) * X[b,j,:]
 *   X [B,N,fp] (fp = feature width padded to 4, 8, 16, 32 or 64; padded columns zero), A [B,N,N],
 *   f1,f2 [B,N,H] (the logit halves of :232-233: x . (W a_s) + b_s), m1,m2 [H] in DEVICE memory, H = 1 or 2;
 *   out [H,B,N,fp] (head-major: the width-1 convolution that follows, :180, is a product per head, and its gradient
 *   arrives in the same layout); t2n [B,N,H,fp] (the softmax-weighted part of out), lse [B,N,H] (log-sum-exp of a
 *   row's logits) and q [B,N,H,2] (the shares of its softmax mass on positive and on negative logits): saved for
 *   backward, or all NULL.
 *   groups > 1: one launch for groups x H heads of the same step (the reference runs n_heads of these steps on one
 *   input, :173; at its batch of 50 a launch for two heads fills a fraction of the chip): f1, f2, lse are
 *   [groups,B,N,H], q [groups,B,N,H,2], t2n [groups,B,N,H,fp], m1, m2 [groups*H], out [groups*H,B,N,fp]; the two sums that do not depend
 *   on the head ([A>0].X and A.X) are then formed once, by a pass of their own when the launch is large, through ws:
 *   gatmi_bnf_attn_fwd_ws_floats(B, N, fp, groups) floats (NULL for groups = 1). */
size_t gatmi_bnf_attn_fwd_ws_floats(int64_t B, int32_t N, int32_t fp, int32_t groups);
int gatmi_bnf_attn_fwd(int64_t B, int32_t N, int32_t H, int32_t fp,
                       const float *X, const float *A, const float *f1, const float *f2,
                       const float *m1, const float *m2, float slope,
                       float *out, float *t2n, float *lse, float *q, float *ws, int32_t groups, void *stream);

/* Backward of the above.  G [H,B,N,fp] = gradient of out (head-major like out); t2n, f1, lse, q, f2 as the forward
 * call had them.  Outputs: dX [B,N,fp] (aggregation part only), dA [B,N,N], df1, df2 [B,N,H] (gradients of the logit halves:
 * df2_j = sum_i dlogit_ij; df1_i = sum_j dlogit_ij in the form (1 - slope)(qneg_i sum_{z>0} - qpos_i sum_{z<=0}) of
 * C_ij n_ij <G_i, X_j>, term by term 0 for a row whose logits lie on one side of LeakyReLU's kink),
 * dm [2H] = {dm1[H], dm2[H]}.
 * accumulate != 0: dX and dA are added to (the heads of one step share X and A).
 * groups > 1: G is [groups*H,B,N,fp]; t2n, f1, lse, q, f2, df1, df2 and dm carry a leading [groups] axis, m1, m2 are [groups*H]; dX and dA are
 * the sums over the groups (each group's share goes through ws and is added in group order).
 * ws: gatmi_bnf_attn_bwd_ws_floats(B, N, H, fp, groups) floats, 16-byte aligned. */
size_t gatmi_bnf_attn_bwd_ws_floats(int64_t B, int32_t N, int32_t H, int32_t fp, int32_t groups);
int gatmi_bnf_attn_bwd(int64_t B, int32_t N, int32_t H, int32_t fp,
                       const float *G, const float *t2n, const float *f1, const float *lse, const float *q,
                       const float *X, const float *A,
                       const float *f2, const float *m1, const float *m2, float slope,
                       float *dX, float *dA, int accumulate, float *df1, float *df2, float *dm, float *ws, int32_t groups, void *stream);

/* InnerProductDecoder (utils/layers.py:307-313): out[b,i,j] = act(<Zd[b,i,:], Zd[b,j,:]>) with
 * Zd = dropout(Z, keep) (:308; the same counter-based mask as the feature dropout of the attention
 * layers, keyed on seed and the element index; keep = 1 disables it) and act = GATMI_ACT_SIGMOID
 * (the reference's default) or GATMI_ACT_IDENTITY.  Z [B,N,F], F <= 128; out [B,N,N] (the
 * reference returns it flattened, :311: same memory). */
int gatmi_ipd_fwd(int64_t B, int32_t N, int32_t F, const float *Z, int32_t activation,
                  float keep, uint64_t seed, float *out, void *stream);

/* Backward of the above: dZ [B,N,F] from dOut [B,N,N], the saved output (sigmoid only; may be NULL
 * for the identity) and Z; same keep and seed as the forward call. */
int gatmi_ipd_bwd(int64_t B, int32_t N, int32_t F, const float *dOut, const float *out,
                  const float *Z, int32_t activation, float keep, uint64_t seed, float *dZ, void *stream);

/* Halo exchange helpers (row-partitioned multi-GPU).  pack: dst[k, :] = src[idx[k], :] for
 * k < n_idx, rows of `width` floats (send buffer of Wh / f2 rows requested by a peer); rows are
 * moved 16 bytes per lane when width is a multiple of 4, one float per lane otherwise.
 * unpack_add: dst[idx[k], :] += src[k, :] (gradient rows returned by a peer; idx may repeat
 * across peers but not inside one call). */
int gatmi_halo_pack(const float *src, const int64_t *idx, int64_t n_idx, int32_t width,
                    float *dst, void *stream);
int gatmi_halo_unpack_add(const float *src, const int64_t *idx, int64_t n_idx, int32_t width,
                          float *dst, void *stream);

/* All peers' returned rows in ONE launch: for u < n_rows, dst[rows[u], :] += src[pos[k], :] summed over
 * k in [ptr[u], ptr[u+1]) in list order.  rows = the distinct owned rows that peers read, pos = for each
 * of them the positions of its copies in the receive buffer (the send list inverted).  A target row
 * belongs to one thread group, so rows returned by several peers need neither atomics nor one launch per
 * peer, and the sum has a fixed order. */
int gatmi_halo_unpack_sum(const float *src, const int64_t *rows, const int64_t *ptr, const int64_t *pos,
                          int64_t n_rows, int32_t width, float *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GATMI_H */
