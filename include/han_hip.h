/* han_hip.h -- C ABI of libhan_hip.so: the MI355X (gfx950) HAN hot path.
 *
 * Every entry point takes raw DEVICE pointers plus a hipStream_t (passed as
 * void*), launches on that stream, never allocates, never synchronises, and
 * returns 0 on success or a negative HAN_E_* / positive hipError_t code.
 * The caller owns every buffer.  Shapes use the notation of SURVEY.md sec. 8:
 *   N   destination rows owned by this call (local rows of a node partition)
 *   NT  rows of the gather tables (== N on one GPU; local + halo otherwise)
 *   F   input feature width          K  attention heads (n_heads[0])
 *   FP  features per head (hid_units[0])     D = K*FP  (this build: D == 64)
 *   P   meta-paths     A  mp_att_size     C  classes
 * All floating-point data is fp32, row-major, dense; indices are int64
 * (row/col pointers) and int32 (neighbour ids).
 *
 * The reference has no native interface for this path: it is Python on
 * TensorFlow 1.x.  Each function names the reference call site it replaces
 * (paths relative to the reference root); the Python binding that mirrors the
 * reference's own layer API is han_amd/layers.py.
 */
#ifndef HAN_HIP_H
#define HAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAN_ABI_VERSION 6

#define HAN_E_BADARG   (-1)   /* null pointer, negative size, inconsistent shape.  The forward
                              * entry points return 0 at once for N == 0 (empty tensors may
                              * carry null data pointers)                                  */
#define HAN_E_UNSUPPORTED (-2) /* shape outside this build (e.g. K*FP != 64)     */
#define HAN_E_WORKSPACE (-3)  /* workspace too small                            */

#define HAN_ACT_IDENTITY 0
#define HAN_ACT_ELU      1
/* `flags` of han_node_attn_fwd / han_node_attn_bwd_cols */
#define HAN_FLAG_XCD_ORDER 1   /* graph has locality (neighbour ids close to the row id): give the blocks of one
                                  XCD a contiguous share of the rows in flight, so that the eight L2s cache eight
                                  different source windows.  Speed only; off for graphs without structure, where it
                                  measured 9 % slower in the HBM regime (eight separate index / output streams). */

#define HAN_FLAG_LEAN 256      /* small graphs whose table lives in the L2s (a few thousand rows) with long rows -- the reference's own
                                  data sets -- are bound by vector-instruction issue, not by memory.  han_node_attn_fwd then reads
                                  the neighbour scores from f2_src (a 4-byte gather per edge and head) instead of recomputing them,
                                  runs the softmax in log2 units and computes ONE attention-dropout hash per (edge, four heads);
                                  for 8 heads x 8 columns one lane owns a whole head of an edge.  han_node_attn_bwd_cols: the same
                                  hash sharing, and the one-lane-per-head map for 8 x 8.  fp32 tables, table_gid NULL.  Not for
                                  large tables: the score gather would cost a memory line per edge there.                      */
#define HAN_FLAG_K2_DEEP 512    /* measurements only: the bf16 eval forward with 8 steps (32 rows) in flight per wave instead of 4
                                  (rounds 2-3's form: more registers, fewer waves; tools/k2_regimes.py --deep)               */
#define HAN_FLAG_K2_SHARED_HASH 1024 /* measurements only: han_node_attn_fwd, training forward of the 8 x 8 shape with both dropouts
                                  on, with ONE attention-dropout hash per lane and 4-edge step, the other three edges' words taken
                                  from the lane's DPP quad -- bitwise the same draws; no faster          */
#define HAN_FLAG_MASKED_EDGES 64 /* han_node_attn_bwd_cols: entries of rowidx below 0 are skipped IN PLACE (their destination's
                                  g row is identically zero -- a destination outside the loss mask of a one-layer model);
                                  the remaining terms are summed in the positions and order of the full pass, so the
                                  results are bit-identical to it.  Opt-in (HANTrainer(masked_backward=True)).          */
/* `flags` of han_project_fwd: which matrix pipe runs the projection (default 0: the library chooses --
 * the bf16 x 6 kernel for training forwards and bf16 features, the exact-fp32 kernel otherwise).
 * Both have fp32-class accuracy; the switches exist for measurements and tests.                     */
#define HAN_FLAG_K1_EXACT_PIPE  2   /* v_mfma_f32_16x16x4_f32 kernels only                                 */
#define HAN_FLAG_K1_MATRIX_PIPE 4   /* the bf16 x 6 kernel wherever it applies (also the fp32 eval forward) */
#define HAN_FLAG_K1_PAIRS      32   /* measurements only: han_project_fwd_multi fuses 2 meta-paths per block, not 4 */
#define HAN_FLAG_K1_4WAVE      16   /* measurements only: the round-2 form of the bf16 x 6 kernel (4 waves x 2 row tiles,
                                       two waves per SIMD) instead of 8 waves x 1 tile (four per SIMD)                     */
/* han_project_fwd, heads wider than the 64 columns of a K1 / K2 row (hid_units > 64, models/gat.py:42-57): such a head runs
 * as ceil(F'/64) column slices, each a K = 1, F' = 64 call with the SAME seed -- so the slices share the per-head
 * input-dropout and attention-dropout draws, as one head must -- and slice s draws its projected-row dropout
 * (layers.py:31-32, one draw per column) from stream 2 + 4 s.  The caller adds the slices' partial scores
 * (b1 / b2 in slice 0 only) and hands the totals to K2 (han_node_attn_fwd: f2_src).                              */
#define HAN_FLAG_FTS_SLICE(s)     (((s) & 0xFF) << 8)
#define HAN_FLAG_FTS_SLICE_OF(fl) (((fl) >> 8) & 0xFF)
/* `flags` of han_sem_attn_fwd / han_sem_attn_bwd */
#define HAN_FLAG_K3_EXACT_PIPE  8   /* fp32 MFMA kernels also for large inputs (default: bf16 x 6 from 65 536 rows) */
#define HAN_FLAG_K3_PAIRS      16   /* measurements only: han_sem_attn_bwd at mp_att_size = 128 with two waves sharing a tile
                                     * (two waves per SIMD, 64 attention columns each) -- not faster: DESIGN.md section 3 */
#define HAN_FLAG_K3_G3_F32     32   /* measurements only: han_sem_attn_bwd's dW product on the fp32 matrix pipe, tile by tile
                                     * (the round-3 form) instead of two tiles per step on the bf16 pipe                   */

/* storage type of X and of the gather tables H / g ("bf16 feats" of the 10M-node
 * config): everything is accumulated in fp32; any head shape (8 x 8 is the tuned one) */
#define HAN_DTYPE_F32  0
#define HAN_DTYPE_BF16 1

/* Seeds.  Every dropout site is a counter-based hash of (seed, stream, global ids), so a
 * forward, its backward and every node partition regenerate identical masks.  `seed` is
 * passed by value; `seed_dev` (device pointer, or NULL) exists for captured hipGraphs,
 * whose kernel arguments are frozen at capture: when non-NULL the effective seed is
 * splitmix64(seed + *seed_dev), read on the device at kernel start, and the host bumps
 * that device word between replays.  Pass NULL for plain stream launches.            */

int han_abi_version(void);
const char *han_error_string(int code);

/* ---- K1: feature projection + attention scores ---------------------------
 * utils/layers.py:18-24 for all K heads of one meta-path:
 *   Xk = dropout_k(X, keep=1-in_drop)      (re-sampled per head, :18-19)
 *   H[:, k*FP:(k+1)*FP] = Xk @ W[:, k*FP:(k+1)*FP]          (:20)
 *   f1[:,k] = H_k . a1[k] + b1[k];  f2[:,k] = H_k . a2[k] + b2[k]   (:23-24)
 * and the Bernoulli draw of the projected-row dropout (:31-32, which the
 * reference applies AFTER the scores were taken from the undropped rows): when
 * fts_drop > 0 the keep bit of H[n][d] is stamped into mantissa bit 0 of that
 * float (a 1-ulp perturbation seen consistently by f1/f2, K2 and the backward),
 * so that K2 applies the mask while it gathers at no extra memory traffic.
 * X (N,F) ldx>=F (elements; x_dtype fp32 or bf16); W (F,D); a1,a2 (K,FP); b1,b2 (K);
 * H (N,D) in table_dtype (f1/f2 are taken from the rows as stored); f1,f2 (N,K).
 * in_drop == 0 -> no input dropout.  row_offset = global id of row 0 (RNG key).
 * workspace: han_project_fwd_workspace() bytes (short inputs -- fewer 128-row tiles than CUs -- split the
 * reduction over F and sum partial tiles from it in a fixed order; long inputs keep the pre-split image of W
 * there, see below); may be NULL when that is 0.                            */
size_t han_project_fwd_workspace(int64_t N, int F, int K, int FP);
/* Round 4: long inputs (at least 16 384 rows, no split over F) may run on the bf16 x 6 matrix-pipe kernels, which
 * read W from a pre-split, LDS-ready image built in the workspace by a small launch in front of the projection
 * (every block used to split the same W tiles again and store them through 4-way bank conflicts):
 * han_project_fwd_workspace() then returns the image size, ceil(F / 32) * 12 288 bytes -- no longer 0 --, and
 * han_project_fwd_multi_workspace() the size for the P meta-paths of one han_project_fwd_multi call.        */
size_t han_project_fwd_multi_workspace(int64_t N, int F, int K, int FP, int P);
/* Keep table of the per-head input dropout (layers.py:18-19), written by the training forward and
 * read by han_project_bwd so that dW does not regenerate the draws: N rows of F bytes (F % 8 == 0);
 * the 8 bytes of features 8o .. 8o+7 of row n form one little-endian 64-bit word whose bit
 *     32*q + 16*(k % 2) + 4*(k / 2) + i        (q = 0,1; i = 0..3; head k = 0..7)
 * is 1 iff head k keeps feature f = 8o + 4q + i of row n -- i.e. the word IS the 64-lane mask of the
 * dW kernel's v_mfma_f32_4x4x1_16B_f32 A operand.  han_project_keep_bytes() returns the size the
 * caller allocates (N*F + 128 B of slack for the kernel's whole-tile mask loads), or 0 when this
 * shape has no table (then pass NULL and the backward regenerates the draws from the seed):
 * built for K == 8, FP == 8, F % 8 == 0, ldx % 4 == 0, N >= 32768 (the matrix-pipe forward).        */
size_t han_project_keep_bytes(int64_t N, int F, int64_t ldx, int K, int FP);
int han_project_fwd(const void *X, int x_dtype, int64_t ldx, const float *W, const float *a1,
                    const float *a2, const float *b1, const float *b2, void *H,
                    int table_dtype, float *f1, float *f2, void *workspace,
                    size_t workspace_bytes, int64_t N, int F, int K, int FP,
                    float in_drop, float fts_drop, uint64_t seed, const uint64_t *seed_dev, int64_t row_offset,
                    uint8_t *keep, int flags, void *stream);

/* All P meta-paths of ONE shared feature matrix in one call (the reference feeds the same matrix to every
 * meta-path: ex_acm3025.py:86, models/gat.py:39).  W (P,F,D), a1/a2 (P,K,FP), b1/b2 (P,K), H (P,N,D),
 * f1/f2 (P,N,K), all contiguous over p; seeds: HOST array of P seeds (may be NULL when in_drop == fts_drop == 0);
 * keep: NULL or P tables, han_project_keep_bytes() apart.  The eval forward of long inputs (no dropout,
 * N >= 16384, 16-byte aligned rows) runs fused: X is read, split and staged once for 4 (or 2) meta-paths per
 * block; every other case is a loop over han_project_fwd, with the same results.                      */
int han_project_fwd_multi(const void *X, int x_dtype, int64_t ldx, const float *W, const float *a1,
                          const float *a2, const float *b1, const float *b2, void *H, int table_dtype,
                          float *f1, float *f2, void *workspace, size_t workspace_bytes, int64_t N,
                          int F, int K, int FP, int P, float in_drop, float fts_drop,
                          const uint64_t *seeds, const uint64_t *seed_dev, int64_t row_offset,
                          uint8_t *keep, int flags, void *stream);

/* dW = Xk^T dH.  keep: the table the forward of the SAME seed wrote (then no draw is regenerated), or
 * NULL: per head dropout masks regenerated from the seed.
 * workspace: han_project_bwd_workspace() bytes, any contents.              */
size_t han_project_bwd_workspace(int64_t N, int F, int K, int FP);
int han_project_bwd(const void *X, int x_dtype, int64_t ldx, const float *dH, float *dW,
                    void *workspace, size_t workspace_bytes, int64_t N, int F, int K,
                    int FP, float in_drop, uint64_t seed, const uint64_t *seed_dev, int64_t row_offset,
                    const uint8_t *keep, void *stream);

/* dX = sum_k mask_k/keep * (dH_k W_k^T): gradient w.r.t. the layer INPUT, needed only
 * for layers >= 1 of a multi-layer stack (models/gat.py:48-57).  dH (N,D); W (F,D);
 * dX (N,F) with row stride ldo (a slice of the previous layer's dM).            */
int han_project_bwd_input(const float *dH, const float *W, float *dX, int64_t ldo, int64_t N,
                          int F, int K, int FP, float in_drop, uint64_t seed, const uint64_t *seed_dev,
                          int64_t row_offset, void *stream);

/* Degree bins and row splitting for skewed graphs (optional; pass NULL for none: the library then picks ONE row
 * shape for the whole launch from E / N).
 * Bins (round 4): real meta-path graphs mix rows of a few entries with rows of thousands (DBLP APA: mean 2.7, max
 * 46; the power-law workload of SURVEY.md section 8d).  With n_short + n_mid > 0 the launch is degree-binned:
 *   short_rows (n_short, int32 row ids): rows with fewer than HAN_SHORT_DEG entries (incl. empty rows) -- one 16-lane
 *       group per row, four rows per wave, the row's ids one coalesced load; best ordered by ceil(deg / 4), then id,
 *       so that the four rows of a wave need the same number of steps;
 *   mid_rows (n_mid): rows of HAN_SHORT_DEG .. split_deg entries -- a wave per row;
 *   rows beyond split_deg: the chunks below.
 * A list pointer may be NULL when its bin holds every row of the launch (n == N: identity).  Every row must be in
 * exactly one of the three sets; each row is processed by one fixed kernel in a fixed order (bitwise reproducible).
 * Splitting: rows (sources, in the backward) with more than split_deg stored entries are skipped by the main
 * launch; each is cut into chunks of consecutive edges [chunk_start, chunk_end),
 * one wave per chunk produces a partial state in `workspace`, and a finishing
 * launch merges a row's chunks in order (deterministic).  chunk_long[c] indexes
 * long_rows; long_ptr (n_long+1) gives each long row's chunk range.  n_long == 0: no row is split (the chunk
 * fields are then not read; split_deg must still bound the rows listed in mid_rows).            */
#define HAN_SHORT_DEG 16
typedef struct han_row_split {
    int64_t split_deg, n_long, n_chunks;
    const int64_t *long_rows, *long_ptr;
    const int32_t *chunk_long;
    const int64_t *chunk_start, *chunk_end;
    void *workspace;
    size_t workspace_bytes;       /* >= han_row_split_workspace(n_chunks) */
    int64_t n_short, n_mid;       /* degree bins; both 0: not binned */
    const int32_t *short_rows, *mid_rows;
} han_row_split_t;
size_t han_row_split_workspace(int64_t n_chunks);

/* Small dense graphs on the matrix pipe (round 4; optional, pass NULL for none).  The reference computes its heads
 * densely -- N x N logits, an additive mask, matmul(coefs, seq_fts): utils/layers.py:26-34 -- and its own data sets ARE
 * dense (ACM PSP 24 % of all pairs, DBLP APCPA / APTPA 30 % / 78 %).  With a han_dense_t the forward / the transposed-
 * graph backward run as 16 x 16 x 4 fp32 MFMA tiles over an adjacency BIT MASK, with the exponential taken out of the
 * inner loop (exp(LeakyReLU(x)) = max(e^x, e^0.2x): per-node factors).  Built for 8 heads x 8 columns, fp32 tables,
 * table_gid NULL, edge_val NULL, graphs without repeated entries (han_csr_to_bitmask counts them), HAN_FLAG_LEAN set and
 * f2_src given: when the per-head range of the scores exceeds 80 (checked on the device) the lean CSR kernel runs
 * instead, in the same call.  bits: [rows][ld_words] 32-bit words, bit (j & 31) of word (j >> 5) of row i set iff the
 * graph stores the entry (i, j); n_table = rows of the table the columns index.                                        */
typedef struct han_dense {
    const uint32_t *bits;
    int64_t ld_words;             /* >= ceil(n_table / 32) */
    int64_t n_table;
    void *workspace;              /* 16-byte aligned */
    size_t workspace_bytes;       /* >= han_node_attn_dense_workspace(rows, n_table, train) */
} han_dense_t;
size_t han_node_attn_dense_workspace(int64_t rows, int64_t n_table, int train);
/* CSR -> bit mask.  *repeated (device int, or NULL) receives the number of entries that were already present (entries outside
 * [0, n_table) -- e.g. the -1 of HAN_FLAG_MASKED_EDGES graphs -- are skipped and counted too: such a graph has no dense form).
 * Launches a memset of bits (and of *repeated) and one kernel on `stream`.                                     */
int han_csr_to_bitmask(const int64_t *rowptr, const int32_t *colidx, int64_t N, int64_t n_table,
                       uint32_t *bits, int64_t ld_words, int *repeated, void *stream);

/* ---- K2: node-level attention ---------------------------------------------
 * utils/layers.py:26-35,46 (dense mask form) == :95-118,127 (sparse form) over
 * the stored neighbours only:
 *   e_ij = LeakyReLU(f1_i + f2_j), alpha = softmax_j, out_i = act(sum_j
 *   drop(alpha_ij) drop(H_j) + c).
 * rowptr (N+1) int64, colidx (E) int32 indexing the NT-row table H (NT,D) (the
 * UNDROPPED projected rows; with fts_drop > 0 bit 0 of every element is its keep
 * bit, as han_project_fwd stamped it); the neighbour score
 * f2_j = H_j[k].a2[k] + b2[k] is recomputed from the gathered row.  table_gid (NT)
 * int32 or NULL: the GLOBAL node id of each table row when the table is a
 * [local | halo] table of a node partition (the dropout RNG is keyed by global
 * ids); NULL means the table index is the global id.  res (N,D) or NULL: the
 * residual term conv1d(seq, F', 1) of layers.py:38-40, added before the activation.
 * edge_val (E) fp32 in colidx order or NULL: the stored values of sp_attn_head's
 * SparseTensor adj_mat, which SCALE the logits (e_ij = LeakyReLU(v_ij*(f1_i+f2_j)),
 * layers.py:95-98) -- NULL is the binary adjacency every shipped config uses.
 * f1 (N,K) for the local rows; a2 (K,FP); b2 (K); c (D).  out row i is written
 * at out + i*out_stride (so the K heads land directly in M[:,p,:],
 * models/gat.py:46,58-60).  Training extras (lse, aggp, tsum: all or none NULL; pre: optional, the
 * backward does not need it): pre (N,D)
 * pre-activation, lse (N,K) log-sum-exp of the scores, aggp (N,D) and tsum (N,K)
 * -- the LeakyReLU'-weighted aggregates that make df1 row-local in the backward.
 * f2_src (NT,K) or NULL: the scores of the table rows.  With K = 1, F' = 64 the neighbour score is GATHERED from
 * this table instead of being recomputed from the row -- the slices of a head wider than 64 columns share the
 * head's scores (f1 / f2_src then hold the totals over the slices; a2 / b2 are not read); with HAN_FLAG_LEAN the
 * lean kernels read it for any head shape; otherwise it is not read.  The backward needs nothing
 * new: han_node_attn_bwd_cols already takes f2 and df1 as inputs, and its slices' df1 / df2 add up (the
 * softmax backward is linear in d alpha); the caller adds (df2_total - df2_slice) x a2_slice to dH.          */
int han_node_attn_fwd(const int64_t *rowptr, const int32_t *colidx, const float *edge_val,
                      const void *H,
                      int table_dtype, const int32_t *table_gid, const float *f1, const float *f2_src,
                      const float *a2, const float *b2, const float *c, const float *res,
                      float *out, int64_t out_stride, float *pre, float *lse, float *aggp,
                      float *tsum, int64_t N, int64_t E, int K, int FP, float slope,
                      float coef_drop, float fts_drop, uint64_t seed, const uint64_t *seed_dev, int64_t row_offset,
                      int activation, int flags, const han_row_split_t *split, const han_dense_t *dense, void *stream);

/* The attention coefficients themselves (attn_head(..., return_coef=True),
 * utils/layers.py:27-30,43-44; models/gat.py:143-172 averages them over the heads):
 * coef[e*K+k] = drop(alpha_ij)[k] for the stored entry e = (i,j) in CSR order, or with
 * mean_heads != 0 coef[e] = mean_k of those.  f1, f2 (N,K)/(NT,K) as han_project_fwd
 * wrote them; coef_drop/seed/row_offset/table_gid as in the forward (so the draws are
 * the ones that forward used).  Diagnostic output: not used by training.            */
int han_node_attn_coefs(const int64_t *rowptr, const int32_t *colidx, const float *edge_val,
                        const int32_t *table_gid, const float *f1, const float *f2, float *coef,
                        int mean_heads, int64_t N, int64_t E, int K, int FP, float slope,
                        float coef_drop, uint64_t seed, const uint64_t *seed_dev, int64_t row_offset, void *stream);

/* Backward tables: ONE fused row per destination i,
 *   [ g_i : D elements of table_dtype | (f1_i, lse_i, s_i, 0)[k] : 4 fp32 per head k ]
 * padded to whole 128-B lines; han_gs_row_bytes() gives the row size (K = 8, F' = 8: 384 B with
 * fp32 g, 256 B with bf16 g).  Under a node partition this is the ONLY table the backward moves
 * per meta-path (one collective instead of one for g and one for the statistics).            */
size_t han_gs_row_bytes(int K, int FP, int table_dtype);

/* Backward, step 1 (row-local): from dOut (N,D; row stride dout_stride), the forward's OUTPUT rows `out`
 * (N,D; row stride out_stride -- e.g. M[:,p,:]) and the saved aggp/tsum/f1/lse compute
 *   pre = act^-1(out): identity, or for ELU  out > 0: out,  out <= 0: log(out + 1)  (round 3: the pre-activation
 *         is no longer stored -- 256 B per row less to write, to keep and to read; act'(pre) = out > 0 ? 1 : out + 1)
 *   g = dOut * act'(pre)   (rounded to table_dtype; the sums below use the rounded value)
 *   s_i[k] = g_i[k] . (pre_i - c)[k]
 *   df1_i[k] = g_i[k] . aggp_i[k] - s_i[k] * tsum_i[k]        -> df1 (N,K)
 *   gs row i = [ g_i | (f1, lse, s, 0)[k] ]                    -> gs (N rows of han_gs_row_bytes())
 *   dc += sum_i g_i  (written, not accumulated; unrounded g)  -> dc (D)
 * workspace: han_node_attn_bwd_workspace() bytes.                          */
size_t han_node_attn_bwd_workspace(int64_t N, int K, int FP);
int han_node_attn_bwd_rows(const float *dOut, int64_t dout_stride, const float *out, int64_t out_stride,
                           const float *aggp, const float *tsum, const float *f1,
                           const float *lse, const float *c, const float *res, void *gs,
                           int table_dtype, float *df1, float *dc, void *workspace,
                           size_t workspace_bytes, int64_t N, int K, int FP, int activation,
                           void *stream);

/* Backward, step 2 (gather over the TRANSPOSED graph, no float atomics):
 * colptr (NS+1) / rowidx (E) list, for each source row j owned by this call,
 * the destination rows i (indices into the NT-row fused table gs).  Produces for each source j
 *   df2_j[k] = sum_i dl_ij,   dH_j = mH_j/keep * sum_i drop(alpha_ij) g_i
 *                                     + df1_j a1 + df2_j a2
 * H (keep bits in bit 0 when fts_drop > 0), f2, df1 are the NS local source rows;
 * dH (NS,D), df2 (NS,K) outputs.
 * src_offset / the ids in rowidx + dst_offset are the global ids used as RNG
 * keys (must match the forward).  edge_val (E) or NULL: the forward's adjacency
 * values permuted into the transposed graph's order.                         */
int han_node_attn_bwd_cols(const int64_t *colptr, const int32_t *rowidx, const float *edge_val,
                           const void *gs, const int32_t *table_gid, const void *H,
                           int table_dtype, const float *f2,
                           const float *df1, const float *a1, const float *a2,
                           float *dH, float *df2, int64_t NS, int64_t E, int K, int FP, float slope,
                           float coef_drop, float fts_drop, uint64_t seed, const uint64_t *seed_dev,
                           int64_t src_offset, int64_t dst_offset, int flags,
                           const han_row_split_t *split, const han_dense_t *dense, void *stream);

/* Backward, step 3: gradients of the score parameters
 *   da1[k,f] = sum_n df1[n,k] H[n,k,f]   da2 likewise with df2
 *   db1[k] = sum_n df1[n,k]              db2[k] = sum_n df2[n,k]            */
size_t han_score_param_bwd_workspace(int64_t N, int K, int FP);
int han_score_param_bwd(const void *H, int table_dtype, const float *df1, const float *df2,
                        float *da1, float *da2, float *db1, float *db2, void *workspace,
                        size_t workspace_bytes, int64_t N, int K, int FP, void *stream);

/* ---- K3: semantic-level (meta-path) attention ------------------------------
 * utils/layers.py:152-159: v = tanh(M Womega + bomega); s = v . uomega;
 * beta = softmax over P PER NODE; Z = sum_p beta_p M_p.
 * M (N,P,D); Womega (D,A); bomega,uomega (A); Z (N,D); beta (N,P).
 * D and A: any multiples of 64 (zero-pad narrower ones: exact).  D in {64,128} with
 * A <= 256 run the tuned kernels, anything wider the run-time-width kernels
 * (models/gat.py:42-61 leaves the last layer's width and mp_att_size free).   */
int han_sem_attn_fwd(const float *M, const float *w_omega, const float *b_omega,
                     const float *u_omega, float *Z, float *beta, int64_t N, int P,
                     int D, int A, int flags, void *stream);

/* dZ (N,D) -> dM (N,P,D), dWomega (D,A), dbomega (A), duomega (A).
 * beta is the forward's output.                                            */
size_t han_sem_attn_bwd_workspace(int64_t N, int P, int D, int A);
int han_sem_attn_bwd(const float *M, const float *w_omega, const float *b_omega,
                     const float *u_omega, const float *beta, const float *dZ, float *dM,
                     float *dw_omega, float *db_omega, float *du_omega, void *workspace,
                     size_t workspace_bytes, int64_t N, int P, int D, int A, int flags, void *stream);

/* ---- classifier + masked loss ----------------------------------------------
 * models/gat.py:65-72: logits = (1/HC) sum_h (Z Wc[h] + bc[h]);  Wc (HC,D,C).
 * models/base_gattn.py:41-48,61-69: per-row masked softmax-CE and accuracy
 * terms with row weight w_i = mask_i * inv_mask_mean / n_total:
 *   loss_acc[0] = sum_i w_i CE_i,  loss_acc[1] = sum_i w_i [argmax == label]
 * labels are class ids (int32, argmax of the one-hot rows); mask uint8.
 * If dZ != NULL also writes dlogits (N,C) = w_i (softmax - onehot) and
 * dZ = dlogits @ mean_h(Wc[h])^T, dWc (HC,D,C), dbc (HC,C).
 * D: any multiple of 64; C: any class count (D in {64,128} with C <= 64: the fused
 * single-launch kernels; otherwise head average -> row kernel -> Z^T dlogits over
 * the masked rows).                                                          */
size_t han_classifier_workspace(int64_t N, int D, int C, int HC);
int han_classifier_loss(const float *Z, const float *Wc, const float *bc,
                        const int32_t *labels, const uint8_t *mask, float row_weight,
                        float *logits, float *loss_acc, float *dZ, float *dWc, float *dbc,
                        void *workspace, size_t workspace_bytes, int64_t N, int D, int C,
                        int HC, void *stream);

/* The backward of the logits alone (HeteGAT_multi.inference returns logits; a caller
 * that forms its own loss hands back dlogits (N,C)):  dZ = dlogits @ mean_h(Wc[h])^T,
 * dWc[h] = Z^T dlogits / HC, dbc[h] = sum_n dlogits / HC.  Same shape rules.      */
size_t han_classifier_bwd_workspace(int64_t N, int D, int C, int HC);
int han_classifier_bwd(const float *Z, const float *Wc, const float *bc, const float *dlogits,
                       float *dZ, float *dWc, float *dbc, void *workspace,
                       size_t workspace_bytes, int64_t N, int D, int C, int HC, void *stream);

/* ---- optimiser --------------------------------------------------------------
 * models/base_gattn.py:12-24: L2 on every trainable + tf.train.AdamOptimizer:
 *   g' = g + l2_coef * p;  m,v EMAs;  p -= lr_t * m / (sqrt(v) + eps)
 * with lr_t = lr * sqrt(1-beta2^t)/(1-beta1^t) computed by the caller when step_dev is
 * NULL; when step_dev (device int64, the step count t >= 1) is given, `lr_t` is the BASE
 * rate lr and the bias correction is computed on the device (captured hipGraphs).  */
int han_adam_step(float *param, const float *grad, float *m, float *v, int64_t n,
                  float lr_t, float beta1, float beta2, float eps, float l2_coef,
                  const int64_t *step_dev, void *stream);

/* sum over all parameters of p^2/2 (the L2 term of the reported loss).
 * out[0] written.  workspace: 4096 floats.                                 */
int han_l2_half_sumsq(const float *param, int64_t n, float *out, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---- input format: additive bias matrix -> CSR ----------------------------
 * utils/process.py:14-25 produces bias (N,N) in {0,-1e9}; an edge is an entry
 * > -1e8.  Two launches: counts per row (then the caller scans) and fill.  */
int han_bias_row_counts(const float *bias, int64_t N, int64_t ld, int64_t *counts,
                        void *stream);
int han_bias_fill_csr(const float *bias, int64_t N, int64_t ld, const int64_t *rowptr,
                      int32_t *colidx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HAN_HIP_H */
