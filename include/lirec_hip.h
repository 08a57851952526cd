/*
 * lirec_hip.h -- C ABI of the MI355X (gfx950) hot-path library for LIReC.
 *
 * The reference (Annusha/LIReC) has no FFI layer: its hot path is the Python
 * object protocol between mlp/train.py / mlp/test.py and mlp/model.py, executed
 * by ATen.  Each entry point below replaces the ATen op sequence of one region
 * of mlp/model.py (cited per function; paths relative to the reference).  The
 * Python host (lirec_amd/) binds these with ctypes and keeps the reference's
 * create_model / model(batch) / loss(out, batch) protocol on top.
 *
 * Conventions
 *   - extern "C"; plain pointers and sizes; no torch types.
 *   - every function returns 0 on success or a hipError_t / LIREC_E* code; never throws.
 *   - all pointers are DEVICE pointers unless stated; the caller owns every buffer;
 *     nothing is allocated inside (scratch comes in through `workspace`, sized by
 *     lirec_workspace_bytes); every call is enqueued on `stream` and returns
 *     without synchronising.
 *   - matrices are row-major fp32; `ld*` are row strides in elements.
 *   - Linear weights are [out, in] row-major with a bias of [out], as
 *     torch.nn.Linear stores them (state_dict layout).
 *   - gradient outputs ACCUMULATE (+=) into the caller's gradient buffers, like
 *     autograd's .grad accumulation; zero them with hipMemsetAsync between steps.
 *   - dropout masks come from a counter-based generator (Philox4x32-10): element
 *     (row, col) of dropout site s is kept iff word[row & 3] of
 *     philox(counter = (col, row >> 2, s, 0), key = seed) >= floor(p * 2^32); kept values
 *     are scaled by 1/(1-p) (torch.nn.Dropout train-mode semantics, mlp/model.py:52).
 *     p == 0 (eval mode) disables it.
 */
#ifndef LIREC_HIP_H
#define LIREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lirec_stream_t;            /* hipStream_t */
typedef void* lirec_ctx_t;               /* library context (lirec_ctx_create); NULL = the default context */

#define LIREC_VERSION 122                /* 0.1.8 */
#define LIREC_MAX_SEG 4

enum {
  LIREC_OK = 0,
  LIREC_EINVAL = 10001,                  /* bad argument (null pointer, size <= 0, unsupported combination) */
  LIREC_EWORKSPACE = 10002               /* workspace too small */
};

/* dropout sites (one counter stream each) */
enum { LIREC_SITE_H1_INTS = 0, LIREC_SITE_H1_CTX = 1, LIREC_SITE_E_INTS = 2,
       LIREC_SITE_E_CTX = 3, LIREC_SITE_GATE = 4,
       LIREC_SITE_TRACK_SAMPLE = 5 };    /* the uniform draw of the positive-track sampler (lirec_margin_loss, sample) */

/* Row selection inside the feature block.  Logical row n of an operand maps to
 * physical row (n / group) * group_stride + (n % group) + group_off of the
 * (B*T, R+1, D) feature tensor: the interaction head reads row 0 of every
 * candidate (group=1, stride=R+1, off=0; mlp/model.py:279), the context head rows
 * 1..R (group=R, stride=R+1, off=1; mlp/model.py:305).  group == 0 means identity. */
typedef struct {
  int32_t group, group_stride, group_off;
} lirec_rowsel;

typedef struct {
  uint64_t seed;                          /* Philox key */
  float p;                                /* drop probability; 0 disables */
  int32_t site;                           /* LIREC_SITE_* of the first-layer activation */
  int32_t site2;                          /* site of the embedding dropout (embed epilogue 1 / pool) */
  const uint64_t* seed_dev;               /* optional device counter: the key is seed + *seed_dev, read by the kernels.
                                           * Lets a recorded train step (command list) draw new masks on every replay
                                           * (lirec_counter_add inside the graph); NULL = seed alone */
} lirec_dropout;

/* ---- embedding MLPs -----------------------------------------------------
 * Replaces, for one head h in {ints, ctx}, the four two-layer branches
 *   Lin -> dropout -> relu -> Lin      (mlp/model.py:279-294 / :305-322,
 *                                       :152-167 / :177-194, :59-76)
 * on `rows` logical rows of X selected by `sel`.
 * Segment s reads X[:, in_off[s] : in_off[s]+in_dim[s]], first layer W1[s]
 * [J, in_dim[s]], second layer W2[s] [out_dim[s], J].
 *   H1 [rows, nseg*J]   = relu(dropout(X_s W1_s^T + b1_s))    (saved for backward)
 *   Z2 [rows, sum out_dim] at (Z2, ldz2):
 *       epilogue 0: Z2 = H1_s W2_s^T + b2_s
 *       epilogue 1: Tn = tanh(Z2) -> Tn_out; Z2 := dropout(Tn)  (cat -> tanh -> dropout, mlp/model.py:296-297,
 *                                                                :326-327; dropout site = drop.site2)
 *
 * Pooled form (context head, mask != NULL): rows = n*R.  The masked mean over the R context
 * clips of a candidate (mlp/model.py:309,315,323-324) is linear and sits directly behind the
 * second Linear, so it is moved in front of it -- exact algebra, 1/R of the layer-2 work:
 *   Hbar[c,:] = sum_r mask[c,r] H1[c,r,:] / div[c]     (the pooling pass, now over H1)
 *   f[c]      = (sum_r mask[c,r]) / div[c]              (1, or 0 for an all-masked candidate with
 *                                                        clamp_zero; NaN without it, as the reference)
 *   Z2[c,:]   = Hbar[c,:] W2^T + f[c] b2                (n rows), then the epilogue as above
 *   with div = sum_r mask (clamp_zero: 0 -> 1, mlp/model.py:303; MidFusionMultiClip has none, :175).
 */
struct lirec_pieces_s;                    /* piece tables + index (defined with lirec_embed_l1_indexed below) */
typedef struct {
  const float* X; int64_t ldx;
  const float* W1[LIREC_MAX_SEG]; const float* b1[LIREC_MAX_SEG];
  const float* W2[LIREC_MAX_SEG]; const float* b2[LIREC_MAX_SEG];
  float* H1;                              /* [rows, nseg*J], ld = nseg*J */
  float* Z2; int64_t ldz2;
  float* Tn; int64_t ldtn;                /* epilogue 1 only */
  const float* mask;                      /* pooled form: fp32 [n, R]; NULL = plain form */
  float* Hbar;                            /* pooled form: [n, nseg*J] out (saved for backward) */
  float* fscale;                          /* pooled form: [n] out (saved for backward) */
  /* compact pooled form (all three from lirec_compact_rows, or all NULL): only the context rows with
   * a non-zero mask are run through layer 1; H1 then holds one row per VALID context row, in rowmap
   * order (allocate it for n*R rows; the tail is left untouched). */
  const int32_t* rowmap; const int32_t* cstart; const int32_t* count;
  const float* wts;                       /* compact form, optional: the mask value of each compact row (lirec_compact_rows2);
                                           * when given, `mask` may be NULL */
  int32_t in_off[LIREC_MAX_SEG], in_dim[LIREC_MAX_SEG], out_dim[LIREC_MAX_SEG];
  int32_t rows, nseg, J, epilogue;
  int32_t R, clamp_zero;                  /* pooled form */
  lirec_rowsel sel;
  lirec_dropout drop;
  /* 1: X is stored as bf16 (ldx in elements; BASELINE config 5 "bf16 storage"): half the feature bytes, and the
   * split-precision core runs two MFMAs per product instead of three (X has no low part).  Default core only.
   * With `planes` (lirec_planes_bytes(.., x_mode = 1); ABI 119) the staging launch copies the valid rows into q16b there and the
   * one-plane persistent kernels read them: bit-identical to the block stored as q16b (x_q32 = 2). */
  int32_t x_bf16;
  int32_t parts;                          /* 0: the whole call; 1: layer 1 (+ the pooling pass of the pooled form) only; 2: layer 2 only; 3: the pooling
                                           * pass + layer 2 (layer 1 done elsewhere: lirec_embed_l1_indexed); 4: stage the ROWS into `planes` and nothing
                                           * else (see rows_staged); lirec_embed_fwd2 wants the same value in both */
  /* Optional workspace (lirec_planes_bytes) for the PRE-SPLIT bf16 operand planes of layer 1.  When given (default GEMM
   * core, segments adjacent in the feature row, in_dim % 32 == 0, J % 128 == 0, aligned X) the forward first writes the
   * selected feature rows as dense hi / lo bf16 planes -- compacted, when the compact form is used -- and the first-layer
   * weights likewise, and layer 1 runs on those planes with LDS-DMA staging (no conversion in the k-loop; same three
   * MFMAs per product, bit-identical results).  The backward call must be handed the same buffer: the weight gradient
   * reads the feature planes again.  NULL: operands are split on the fly per k-tile. */
  void* planes; int64_t planes_bytes;
  /* Optional, with `planes` only (ABI 115): the feature rows come as PIECE TABLES + INDEX (the inputs of lirec_gather_features:
   * text | clip-visual | track-1 | track-2, nseg = 4, in_off[0] = 0) instead of X -- the staging pass writes the q32b rows of
   * layer 1 straight from the tables, the (B, T, R+1, D) block is never built and X may be NULL.  The row selector then
   * addresses index rows.  The backward call gets the same `planes` buffer and needs neither X nor the pieces.  When the
   * q32b form does not apply the call fails with LIREC_EINVAL (use lirec_embed_l1_indexed). */
  const struct lirec_pieces_s* pieces;
  /* Optional, pooled form only (ABI 118): lirec_hbits_bytes(rows, nseg * J) bytes.  The pooling pass, which reads every valid row
   * of H1 anyway, also leaves one bit per element, [H1 > 0] -- all that backward needs of H1 (the relu / dropout derivative in the
   * un-pooling pass: relu(dropout(z)) > 0 <=> kept and z > 0).  Hand the same buffer to lirec_embed_bwd and H1 need not be kept
   * from forward to backward (151 MB at the bench shape) nor be read again (58 MB).  Needs R <= 64, (nseg * J) % 4 == 0 and
   * 16-byte aligned H1 / Hbar (LIREC_EINVAL otherwise); bit-identical results. */
  void* hbits;
  /* 1 (ABI 118, with `planes`): X is not an fp32 block but the SAME block stored as q32b (lirec_to_q32b: blocked bf16 hi / lo,
   * the fp32 footprint; ldx = its columns, rows padded to 32) -- the storage layer 1 reads.  Nothing is staged: layer 1 and its
   * weight gradient gather their rows from X through a row list the staging launch writes (which then only stages W1 and the
   * dropout keep bytes).  Bit-identical to the staged path.
   * 2 (ABI 119): X is the block stored as q16b (lirec_to_q16b: its values rounded to bf16, blocked, HALF the fp32 footprint --
   * "bf16 feature storage", BASELINE config 5): gathered the same way by the ONE-PLANE forms of the two kernels (the stored value
   * is the hi half of the product's split, there is no lo half: two MFMAs per product instead of three).  Block form only.
   * 3 (ABI 122): X is the block stored as q16c (lirec_to_q16c: the same bf16 values in 32 x 64 blocks of 4 KiB, 128-byte rows) --
   * the storage of the SINGLE-PASS mode (lirec_set_gemm_mode(3)) and of that mode only: its forward kernel takes 64 of k per step,
   * whole 128-byte lines of both operands (rows as q16c, first-layer weights staged -- or kept by the caller, W1q -- as q16c), one
   * MFMA per product.  Mode 3 refuses q16b / q32b rows (LIREC_EINVAL), mode 2 refuses q16c; a row-major bf16 block (x_bf16) is
   * staged in the form of the mode in force.  Block form only; ldx % 64 == 0, segments start at multiples of 64 columns. */
  int32_t x_q32;
  /* 1 (ABI 118): the feature rows, the dropout keep bytes and the partition bound are ALREADY in `planes` -- an earlier call with
   * parts = 4 (same arguments, same `planes`, drop.seed = the key THIS call draws its masks from) put them there, on a stream this
   * one is ordered behind.  The call then stages the first-layer weights only.  That is how a training loop stages the rows of
   * batch t + 1 beside the MFMA-bound backward of batch t (an input pipeline: the rows do not depend on the weights).
   * q32b path only: LIREC_EINVAL where the call would have fallen back to the on-the-fly kernels. */
  int32_t rows_staged;
  /* Optional (ABI 119, with `planes`): W1q[i] = the first-layer weights of segment i ALREADY in the q32b form ([J][in_dim[i]],
   * lirec_to_q32b's layout, 256-byte aligned) -- kept current by the caller: by lirec_to_q32b once, then by the fused update of
   * lirec_embed_bwd_args::adam (lirec_fused_adam::wq).  The staging launch then leaves the weights alone (with rows_staged there is
   * no staging launch at all); all segments or none.  Bit-identical.  (ABI 122) In the single-pass mode the form is q16c
   * (lirec_to_q16c; at the same addresses, the first half of each matrix's bytes) -- made and read under that mode only. */
  const void* W1q[LIREC_MAX_SEG];
} lirec_embed_fwd_args;
int64_t lirec_hbits_bytes(int32_t rows, int32_t W);
int lirec_embed_fwd(const lirec_embed_fwd_args* a, lirec_stream_t stream);
/* Both heads of one model in one call (same results as two lirec_embed_fwd calls): the second layers of the two
 * heads -- small GEMMs on the candidate rows -- share one grouped launch. */
int lirec_embed_fwd2(const lirec_embed_fwd_args* a, const lirec_embed_fwd_args* b, lirec_stream_t stream);

/* Layer 1 on the UNIQUE feature pieces (second half).  A row of the loader's (B, T, R+1, D) block is
 * [clip piece (text | clip-visual) | track-1 piece | track-2 piece] and every piece is shared by many rows
 * (mixed_utils/classification_dataloader.py:336-349, :477-478, :531-533); the first Linear of a modality branch
 * (mlp/model.py:279-292, :305-320) acts on one piece only.  With the tables and the index of lirec_gather_features the
 * pre-activations are computed once per PIECE and expanded per row with that row's dropout mask: H1 -- and everything
 * after it -- is bit-identical to lirec_embed_fwd on the expanded block, which is never built.
 *   heads[h]: the arguments lirec_embed_fwd would get (X ignored; nseg = 4: text, visual, tracks1, tracks2; J % 256 == 0);
 *   zclip[h] [n_clip, 2J], ztrk[h] [n_track, 2J]: caller-provided scratch (kept for nothing: backward does not need it).
 * Does layer 1 only; the caller continues with parts = 3 (pooling pass + layer 2: lirec_embed_fwd / lirec_embed_fwd2). */
typedef struct lirec_pieces_s {
  const float* clip; int64_t ld_clip; int32_t n_clip;       /* [n_clip, text_dim + visual_dim] fp32 */
  const float* track; int64_t ld_track; int32_t n_track;    /* [n_track, track_dim] fp32 */
  const int32_t* index;                                     /* [physical rows, 3]: clip, track-1, track-2 piece (or < 0) */
  int32_t text_dim, visual_dim, track_dim;
  /* Optional (ABI 118): the same tables stored as q32b (lirec_to_q32b; rows n_clip + 1 / n_track + 1 -- the zero row included --
   * padded to 32).  With them (and `planes`) layer 1 and its weight gradient GATHER their rows from the tables through the index:
   * no staged copy of the rows exists, `clip` / `track` may be NULL.  clip_rows / track_rows (optional): a second level for a
   * store of ALL pieces resident in HBM -- an index value v >= 0 then names table row clip_rows[v] / track_rows[v] (v < 0: the
   * zero row n_clip / n_track), so a batch brings two short lists and the index, and no table is cut. */
  const void* clip_q; const void* track_q;
  const int32_t* clip_rows; const int32_t* track_rows;
} lirec_pieces;
/* fp32 [rows][cols] (cols % 32 == 0, 16-byte aligned) -> q32b with the rows padded to a multiple of 32 by zero rows;
 * lirec_q32b_bytes(rows, cols) bytes at dst (256-byte aligned). */
int64_t lirec_q32b_bytes(int64_t rows, int64_t cols);
int lirec_to_q32b(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, lirec_stream_t stream);
/* The same to q16b (ABI 119): every value rounded to bf16 (nearest even), 32 x 32 blocks of 2 KiB, row r of a block = 64 B;
 * lirec_q16b_bytes(rows, cols) bytes at dst (256-byte aligned).  What lirec_embed_fwd_args::x_q32 = 2 reads. */
int64_t lirec_q16b_bytes(int64_t rows, int64_t cols);
int lirec_to_q16b(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, lirec_stream_t stream);
/* The same to q16c (ABI 122): the bf16 values in 32 x 64 blocks of 4 KiB, row r of a block = 128 B (cols % 64 == 0; the same
 * lirec_q16b_bytes).  What x_q32 = 3 reads, and the form of W1q in the single-pass mode. */
int lirec_to_q16c(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, lirec_stream_t stream);
int lirec_embed_l1_indexed(const lirec_embed_fwd_args* const* heads, int32_t nh, const lirec_pieces* pieces,
                           float* const* zclip, float* const* ztrk, lirec_stream_t stream);

/* Row compaction for the pooled form.  A context row whose mask is 0 cannot influence any output
 * (the masked mean multiplies it by 0 and its gradient is 0; mlp/model.py:309-324), so it need not
 * be computed at all.  From mask [n, R] this writes rowmap[j] = c*R + r of the j-th row with a
 * non-zero mask (ascending; size n*R), cstart[c] = first compact row of candidate c (size n+1) and
 * count[0] = number of valid rows.  The count stays on the device: the GEMMs read it there and size
 * their work at run time, so nothing synchronises with the host.  Dropout counters keep the
 * original row ids, so every value equals the uncompacted computation. */
int lirec_compact_rows(const float* mask, int32_t n, int32_t R, int32_t* rowmap, int32_t* cstart, int32_t* count,
                       lirec_stream_t stream);
/* The same reading the mask in the dtype the DataLoader delivers it (mask_dtype: 0 fp32, 1 int64 -- rels_mask, SURVEY
 * appendix B --, 2 float64; no cast kernel) and also writing wts[j] = (float)mask[rowmap[j]] (size n*R, optional): the
 * pooling passes then read each candidate's weights as one contiguous run. */
int lirec_compact_rows2(const void* mask, int32_t mask_dtype, int32_t n, int32_t R, int32_t* rowmap, int32_t* cstart,
                        int32_t* count, float* wts, lirec_stream_t stream);
/* cstart of lirec_compact_rows2 must have room for 2 n + 1 ints: the entries behind n + 1 are scratch while the call runs
 * (one wave per candidate, two launches; lirec_compact_rows keeps n + 1 and a single-workgroup kernel). */

/* Backward of lirec_embed_fwd (replaces autograd through the same lines):
 *   given dZ2 [rows, sum out_dim] (ld lddz2) -- already multiplied by the
 *   tanh/dropout derivative when epilogue was 1 -- accumulates
 *   dW2_s += dZ2_s^T H1_s, db2_s += colsum dZ2_s,
 *   dZ1 = (dZ2_s W2_s) * [H1 > 0] / (1-p)   (into workspace, rows*nseg*J floats)
 *   dW1_s += dZ1_s^T X_s,  db1_s += colsum dZ1_s.
 * dX is never formed: the features do not require grad.
 * Pooled form (mask != NULL; rows = n*R, dZ2 is [n, sum out_dim]):
 *   dW2_s += dZ2_s^T Hbar_s, db2_s += sum_c f[c] dZ2_s[c,:], dHbar = dZ2_s W2_s  (n rows),
 *   dZ1[c,r,:] = dHbar[c,:] * mask[c,r]/div[c] * [H1[c,r,:] > 0] / (1-p), then dW1/db1 as above. */
typedef struct {
  const float* X; int64_t ldx;
  const float* W2[LIREC_MAX_SEG];
  const float* H1;
  const float* dZ2; int64_t lddz2;
  float* dW1[LIREC_MAX_SEG]; float* db1[LIREC_MAX_SEG];
  float* dW2[LIREC_MAX_SEG]; float* db2[LIREC_MAX_SEG];
  void* workspace; int64_t workspace_bytes;
  const float* mask; const float* Hbar; const float* fscale;   /* pooled form (as saved by lirec_embed_fwd) */
  const int32_t* rowmap; const int32_t* cstart; const int32_t* count;   /* compact pooled form, as in forward */
  const float* wts;                                                      /* as in forward */
  int32_t in_off[LIREC_MAX_SEG], in_dim[LIREC_MAX_SEG], out_dim[LIREC_MAX_SEG];
  /* parts: 0 the whole backward; 1 only the second-layer weight gradient (dW2, db2); 2 everything else = 3 then 4;
   * 3 only the hidden-layer gradient (dZ1, or dHbar in the pooled form); 4 only what follows it (un-pooling, dW1 / db1).
   * 1 is independent of 2 given dZ2, and the 4 of one head is independent of the 4 of another: a caller may enqueue
   * them on two streams (each stream with its own library context: both may split K into their context's scratch).
   * lirec_embed_bwd only: 5 = the un-pooling pass of 4 alone (then lirec_embed_dw1_indexed). */
  int32_t rows, nseg, J, parts;
  int32_t R, clamp_zero;
  lirec_rowsel sel;
  lirec_dropout drop;
  int32_t x_bf16, reserved2_;             /* as in lirec_embed_fwd_args */
  void* planes; int64_t planes_bytes;     /* the buffer the forward call filled (or NULL), see lirec_embed_fwd_args */
  const void* hbits;                      /* pooled form: the sign bits the forward call left (see lirec_embed_fwd_args); H1 may then be NULL */
  const struct lirec_pieces_s* pieces;    /* as in the forward call when its rows were gathered from q32b piece tables (else NULL) */
  int32_t x_q32, reserved3_;              /* as in the forward call */
  /* Optional (ABI 119; parts 0, 2 or 4 on the q32b path, single GPU): fold the Adam update of the first-layer parameters into the
   * kernel that finishes their gradients -- see lirec_fused_adam.  Taken from the first head of the call; LIREC_EINVAL where the
   * weight gradient would not run on the gemm_p2 kernels or the problems of the call do not cover `n_params`. */
  const struct lirec_fused_adam_s* adam;
} lirec_embed_bwd_args;
/* An Adam update folded into the launch that FINISHES the gradients it consumes (lirec_embed_bwd_args::adam: W1, b1 of every
 * segment of every head of the call, in the launch that sums the stream-K partial tiles of dW1 -- 10 M of the 34 M parameters at
 * the bench shape):
 * the thread that owns four gradient elements applies Adam to the parameters and moments at the same offsets of their flat
 * buffers (lirec_adam_step's arithmetic and op order: bit-identical), still stores the gradient, and -- `wq` -- writes the new
 * weights' q32b form into a shadow buffer (what lirec_embed_fwd_args::W1q points into): the weights at element offset o of the
 * flat buffers go to byte 4 * (o - wq_first) of `wq`, which must come out 256-byte aligned for every W1 (single-pass mode, ABI 122:
 * the q16c form at the same byte, half as long).  Saves the
 * gradient's round trip (written by the reduce, read back by lirec_adam_step), one launch, and the next forward's W1 staging.
 * p, g, m, v: buffers with ONE layout; every dW1[i] / db1[i] of the call must point into [g, g + n).  `n_params` = the number of
 * parameter elements the caller expects the call to update (sum of J * in_dim + J over the segments): checked, so that a
 * parameter the call does not reach cannot silently miss its update.  step / step_dev as in lirec_adam_step. */
typedef struct lirec_fused_adam_s {
  float* p; const float* g; float* m; float* v;
  void* wq; int64_t wq_first;
  int64_t n, n_params;
  int32_t step; float lr, beta1, beta2, eps, weight_decay, grad_scale;
  const int64_t* step_dev;
} lirec_fused_adam;
int lirec_embed_bwd(const lirec_embed_bwd_args* a, lirec_stream_t stream);
/* Both heads in one call: dW2 of both heads in one grouped launch, likewise the hidden-layer gradients; the two
 * first-layer weight gradients stay separate launches (a's first). */
int lirec_embed_bwd2(const lirec_embed_bwd_args* a, const lirec_embed_bwd_args* b, lirec_stream_t stream);
/* First-layer weight gradients from the unique pieces (the backward of lirec_embed_l1_indexed):
 *   dW1_seg += sum over pieces u of ( sum of the dZ1 rows whose index names u ) (x) piece_u ,  db1_seg += sum of the dZ1 rows.
 * The inner sums are  S = P^T dZ1  with P the 0/1 incidence matrix of the index (written by the call), an ordinary weight-
 * gradient GEMM: no sort, no atomics, fixed summation order.  heads[h]: the arguments of lirec_embed_bwd after its parts 3
 * (and 5, the un-pooling pass alone, for the pooled form) have run -- dZ1 sits in the workspace; X is not used.
 * The piece tables must carry ONE EXTRA ZERO ROW behind their n_clip / n_track rows (the piece of a negative index).
 * Scratch: P[h] rows_h x round4((n_clip + 1) + 2 (n_track + 1)) floats, S[h] ((n_clip + 1) + (n_track + 1)) x 2J floats. */
int lirec_embed_dw1_indexed(const lirec_embed_bwd_args* const* heads, int32_t nh, const lirec_pieces* pieces,
                            float* const* P, float* const* S, lirec_stream_t stream);
/* scratch lirec_embed_bwd needs: `rows` = the logical row count, plus n for the pooled form
 * (pass rows = n*R + n).  (Twice the fp32 gradient of the hidden layer, rows rounded up to 32: the plain form keeps
 * the fp32 gradient and its bf16 planes side by side when layer 1 runs on planes.) */
int64_t lirec_workspace_bytes(int32_t rows, int32_t nseg, int32_t J);
/* bytes of the `planes` workspace of one head: feature planes for `rows` rows (rounded up to 32) of `dsum` = sum of
 * in_dim columns (hi + lo; hi only when x_bf16) + weight planes for J x dsum (hi + lo); 256-byte aligned parts */
int64_t lirec_planes_bytes(int32_t rows, int32_t dsum, int32_t J, int32_t x_mode);
/* (x_mode: 0 = fp32 rows, staged into the workspace; 2 = rows gathered from q32b storage -- x_q32 or q32b piece tables --, for
 *  which the workspace holds no row copy; 1 = a row-major bf16 block (x_bf16): its rows are staged as ONE plane, q16b) */

/* ---- masked mean over context clips ("pairwise" pooling pass) ------------
 * Replaces (z.view(n, R, W) * mask).sum(1) / divider followed by tanh and
 * dropout (mlp/model.py:301-327; :174-199 without the divider clamp):
 *   P[c,:] = sum_r mask[c,r] Z2[c,r,:] / div[c],  div = sum_r mask (clamp_zero: 0 -> 1)
 *   Tn = tanh(P);  E = dropout(Tn)   (site = drop.site2, row = c)
 * Z2 is [n*R, W] with row stride ldz; mask is fp32 [n, R]. */
int lirec_pool_fwd(const float* Z2, int64_t ldz, const float* mask, int32_t n, int32_t R, int32_t W,
                   int32_t clamp_zero, float* Tn, int64_t ldtn, float* E, int64_t lde,
                   const lirec_dropout* drop, lirec_stream_t stream);
/* dZ2[c,r,:] = dP[c,:] * mask[c,r] / div[c]   (dP already carries the tanh/dropout derivative) */
int lirec_pool_bwd(const float* dP, int64_t lddp, const float* mask, int32_t n, int32_t R, int32_t W,
                   int32_t clamp_zero, float* dZ2, int64_t lddz, lirec_stream_t stream);

/* ---- gating unit ----------------------------------------------------------
 * Replaces GatingUnit.forward (mlp/model.py:349-354):
 *   G = dropout(relu(EE Wg^T + bg)),  EE = [E_ctx | E_ints]  [n, K]  (ctx first, :352) */
int lirec_gate_fwd(const float* EE, int64_t ldee, const float* Wg, const float* bg, int32_t n, int32_t K,
                   int32_t N, float* G, int64_t ldg, const lirec_dropout* drop, lirec_stream_t stream);
/* Backward: dZg = dG * [G > 0] / (1-p) must already be in dZg (lirec_heads_bwd writes it);
 *   dWg += dZg^T EE, dbg += colsum dZg, and
 *   dEE[:, j] (op)= (dZg Wg)[:, j] * tanh'/dropout factor of column j, where
 *   columns [0, split) use (Tn_ctx, site_ctx) and [split, K) use (Tn_ints, site_ints);
 *   acc_first != 0 adds the existing contents of dEE[:, :split) (the relationship-head
 *   gradient) before applying the factor. */
int lirec_gate_bwd(const float* dZg, int64_t lddzg, const float* EE, int64_t ldee, const float* Wg,
                   int32_t n, int32_t K, int32_t N, int32_t split,
                   const float* Tn, int64_t ldtn, float* dWg, float* dbg, float* dEE, int64_t lddee,
                   int32_t acc_first, const lirec_dropout* drop, int32_t site_ctx, int32_t site_ints,
                   lirec_stream_t stream);
/* The same with `parts`: 0 both; 1 only dWg / dbg; 2 only dEE (independent given dZg: two streams, two contexts). */
int lirec_gate_bwd_parts(const float* dZg, int64_t lddzg, const float* EE, int64_t ldee, const float* Wg,
                         int32_t n, int32_t K, int32_t N, int32_t split,
                         const float* Tn, int64_t ldtn, float* dWg, float* dbg, float* dEE, int64_t lddee,
                         int32_t acc_first, const lirec_dropout* drop, int32_t site_ctx, int32_t site_ints,
                         int32_t parts, lirec_stream_t stream);

/* The gate's three GEMMs on STAGED q32b operands (ABI 118; ABI 120: one kernel form for all three).  `ws` =
 * lirec_gate_ws_bytes(n, K, N) bytes, 256-byte aligned, kept by the caller from the forward call to the backward call: the forward
 * stages Wg and EE there (blocked bf16 hi / lo, the fp32 footprint) AND their transposes (one read, two forms), the backward stages dZg
 * and its transpose -- so that forward (EE . Wg^T), data gradient (dZg . Wg, through the rows of Wg^T) and weight gradient (dZg^T . EE,
 * through the rows of dZg^T and EE^T) all read k-contiguous rows (the weights must not change between the two calls).  Kernel:
 * gemm_p3.hpp (wave-specialised workgroups, 128 x 96 or 128 x 128 tiles, persistent over tiles) when n % 128 == 0, N % 128 == 0,
 * N % 96 == 0, K % 96 == 0, split % 96 == 0 and K / 96 >= 8; else gemm_p2.hpp's forward / data-gradient kernels (n % 32 == 0, K and N
 * % 256, split % 256) with the plain weight-gradient kernel; else -- or with ws NULL -- the calls ARE the plain ones.  Same arithmetic
 * as lirec_gate_fwd / lirec_gate_bwd_parts (three bf16 products per element pair, fp32 accumulate, same dropout counters), another
 * summation order.  2 * split == K and contiguous EE / dZg are required throughout.  parts as in lirec_gate_bwd_parts. */
int64_t lirec_gate_ws_bytes(int32_t n, int32_t K, int32_t N);
int lirec_gate_fwd_ws(const float* EE, int64_t ldee, const float* Wg, const float* bg, int32_t n, int32_t K,
                      int32_t N, float* G, int64_t ldg, const lirec_dropout* drop, void* ws, int64_t ws_bytes,
                      int32_t weights_staged, lirec_stream_t stream);
/* Stages Wg (and Wg^T) into `ws` on its own (37.7 MB read, 75 MB written for the 3072 x 3072 gate: 25-45 us of HBM traffic that depends
 * on nothing else in the step).  A caller puts it on another stream beside the MFMA-bound layer-1 launch and then passes weights_staged = 1 to
 * lirec_gate_fwd_ws, which in that case stages the rows only and fails with LIREC_EINVAL where it would have fallen back.
 * LIREC_EINVAL when the shapes do not qualify (same rule as lirec_gate_fwd_ws: ask before relying on it). */
int lirec_gate_stage_weights(const float* Wg, int32_t n, int32_t K, int32_t N, void* ws, int64_t ws_bytes, lirec_stream_t stream);
int lirec_gate_bwd_ws(const float* dZg, int64_t lddzg, const float* EE, int64_t ldee, const float* Wg,
                      int32_t n, int32_t K, int32_t N, int32_t split,
                      const float* Tn, int64_t ldtn, float* dWg, float* dbg, float* dEE, int64_t lddee,
                      int32_t acc_first, const lirec_dropout* drop, int32_t site_ctx, int32_t site_ints,
                      int32_t parts, void* ws, int64_t ws_bytes, int32_t rows_staged, lirec_stream_t stream);
/* (parts 4 = stage the rows of dZg into `ws` and nothing else; a later call with rows_staged = 1 -- on any stream ordered behind
 *  it -- then skips that pass: this is how the weight gradient (part 1) runs on another stream beside the data gradient (part 2),
 *  both reading ONE staged copy.  Where the shapes do not qualify part 4 does nothing and the other parts are the plain kernels.) */

/* ---- output heads ----------------------------------------------------------
 * Replaces out_ints / out_ctx (mlp/model.py:332-336, :205-209, :90): Y = A W^T + b. */
int lirec_linear_fwd(const float* A, int64_t lda, const float* W, const float* b, int32_t n, int32_t K,
                     int32_t N, float* Y, int64_t ldy, lirec_stream_t stream);
/* Backward of one head: dW += dY^T A, db += colsum dY, and (if dA != NULL)
 *   dA (op)= dY W  with epilogue `mode`:
 *     0 store            1 relu-dropout backward: * [act > 0] / (1-p)   (act = G)
 *     2 tanh-dropout backward: * keep/(1-p) * (1 - act^2)               (act = Tn; site from drop->site2)
 *   accumulate != 0 adds the previous contents of dA before the factor is applied. */
int lirec_linear_bwd(const float* dY, int64_t lddy, const float* A, int64_t lda, const float* W,
                     int32_t n, int32_t K, int32_t N, float* dW, float* db,
                     float* dA, int64_t ldda, int32_t mode, const float* act, int64_t ldact,
                     int32_t accumulate, const lirec_dropout* drop, lirec_stream_t stream);

/* The same for several heads at once (out_ints and out_ctx are independent given their inputs): one grouped
 * launch per GEMM kind instead of one per head.  Field meaning as in lirec_linear_fwd / lirec_linear_bwd. */
typedef struct {
  const float* A; int64_t lda; const float* W; const float* b; float* Y; int64_t ldy;
  int32_t n, K, N, reserved_;
} lirec_linear_fwd_args;
int lirec_linear_fwd_group(const lirec_linear_fwd_args* v, int32_t count, lirec_stream_t stream);
typedef struct {
  const float* dY; int64_t lddy; const float* A; int64_t lda; const float* W;
  float* dW; float* db; float* dA; int64_t ldda; const float* act; int64_t ldact;
  int32_t n, K, N, mode, accumulate;
  int32_t parts;                          /* 0 both; 1 only dW / db; 2 only dA (independent of each other: two streams) */
  lirec_dropout drop;
} lirec_linear_bwd_args;
int lirec_linear_bwd_group(const lirec_linear_bwd_args* v, int32_t count, lirec_stream_t stream);

/* ---- losses ---------------------------------------------------------------
 * One fused forward+backward per loss: writes the scalar loss and d(loss)/d(logits).
 *
 * lirec_margin_loss covers the four max-margin losses of mlp/model.py:
 *   MaxMarginCrossEntropyLoss :427-441   (T=1, rels=NULL, lymbda=1, margin=opt.margin)
 *   MultiTaskMaxMargin        :387-419   (T=1, rels_mean_valid=1, margin=opt.margin)
 *   MarginLoss                :450-494   (rels=NULL, lymbda=1, margin=opt.tr_margin)
 *   MarginTrackRelsLoss       :503-575   (margin=opt.tr_margin)
 * Per clip b: padded tracks (mem==0) get logit -inf (written back into `ints` when
 * mask_inplace, as the reference's in-place masking does, :460,:512); S=sigmoid;
 * positive track k:
 *   sel[b] when sel != NULL and sel[b] >= 0 (a draw injected by the caller), else
 *   0 when tr_correct (:476,:550), else
 *   sample != 0 (opt.tr_cat_distr, :468-471,:538-543): drawn in the kernel from the categorical distribution
 *       p_t = softmax_t(ints[t,y])                                   (MarginLoss; logits with -inf on padded tracks)
 *       p_t = (softmax_t(ints[t,y]) + nan->0(softmax_t(rels'[t,r0]))) / 2      (MarginTrackRelsLoss; rels' = rels with
 *             the None column appended and -inf wherever the track is padded or its label is None, :516-524, :542)
 *     one wave per clip: lane t holds track t, max / sum by wave shuffles, u = (word 0 of
 *     philox(counter = (b, 0, LIREC_SITE_TRACK_SAMPLE, 0), key = sample_seed [+ *sample_seed_dev]) >> 8) * 2^-24,
 *     k = the first t with cumsum(p)_t > u * sum(p) (torch.multinomial normalises the same way; its own draw comes
 *     from torch's global generator, which nothing else can reproduce: the caller may inject one through `sel`);
 *     probs_out[b,t] (optional) receives p_t, the tensor the reference hands to torch.multinomial;
 *   otherwise argmax_t (S[t,y] + Q[t,r0]) * mem[t]  (:479,:552-553);
 * negatives = every (t,c) not masked by mem, multilab
 * weights or the target-column rules (:462-467,:526-537);
 *   sum variant: sum relu(m - pos + S) over negatives          (:488-492,:563-573)
 *   max variant: sum_t relu(m - pos + max_c S*mask)            (:483-486,:557-562)
 * loss = lymbda * mean_b(ints part) + mean(rels part), the latter over B clips or,
 * with rels_mean_valid, over the clips whose label != NR (:407-418).
 */
typedef struct {
  float* ints; int64_t ld_ints;           /* [B*T, C] logits (modified in place when mask_inplace) */
  const float* rels; int64_t ld_rels;     /* [B*T, NR] or NULL */
  const float* mem;                       /* [B, T] fp32 0/1, or NULL = all ones */
  const float* w;                         /* [B, C] multilab weights fp32 0/1, or NULL = all ones */
  const int32_t* y;                       /* [B] interaction labels */
  const int32_t* r;                       /* [B, T] relationship labels (NR = None), or NULL */
  const int32_t* g;                       /* [B, 2] gt_tracks, or NULL = (0,0) */
  const int32_t* sel;                     /* [B] forced positive track, <0 = argmax; or NULL = argmax */
  float* d_ints; int64_t ld_dints;        /* [B*T, C] out */
  float* d_rels; int64_t ld_drels;        /* [B*T, NR] out (when rels) */
  float* loss;                            /* [1] out */
  float* partial;                         /* [2*B + 2] scratch */
  int32_t* sel_out;                       /* [B] chosen track (out, optional) */
  int32_t B, T, C, NR;
  float margin, lymbda;
  int32_t max_neg, tr_correct, mask_inplace, rels_mean_valid;
  /* 1: mem and w point to float64, y / r / g to int64 -- the dtypes the reference's DataLoader delivers
   *  -- and are read in place: no cast kernels between the loader batch and the loss. */
  int32_t loader_types;
  /* 0 argmax / forced; 1 draw the positive track in the kernel (tr_cat_distr); 2 as 1, but only probs_out and sel_out
   * are written (no loss, no gradients): for a caller that wants to draw from the probabilities itself */
  int32_t sample;
  uint64_t sample_seed;                   /* Philox key of the draw */
  const uint64_t* sample_seed_dev;        /* optional device counter added to the key (graph replay), as lirec_dropout */
  float* probs_out;                       /* [B, T] out, optional */
  /* Optional arrival counter (device int32, ZERO on entry, left zero on exit): when given, the clip whose workgroup
   * arrives last sums the per-clip partials in a fixed order and writes `loss` -- one launch instead of two. */
  int32_t* arrive;
  /* Data-parallel form of the batch means (the reference is single-device, mlp/train.py:42).  A rank that holds B
   * of the B_global clips of a global batch, and whose gradients are then AVERAGED over `world` ranks, passes the denominators
   * of the GLOBAL means divided by world:
   *   batch_divisor = B_global / world                                        (0: B, the local clip count)
   *   rels_divisor  = (clips of the global batch whose label != NR) / world   (0: the local count; rels_mean_valid only, :407-418)
   * The average over the ranks of the per-rank losses / gradients then IS the single-process loss / gradient of the global
   * batch, whatever the ranks' own counts.  divisors_dev: optional device float[2] = {batch_divisor, rels_divisor} read by the
   * kernel instead of the two values (the result of an all-reduce that never visits the host); entries <= 0: the defaults. */
  float batch_divisor, rels_divisor;
  const float* divisors_dev;
  /* stride of `y` in elements (0 = 1): the clip's label read IN PLACE out of the loader's [B, R+1, 1] labels tensor of the
   * multi-clip recipe (mlp/model.py:393: labels[:, 0]) -- no gathered copy, so a recorded step sees a refilled batch's labels */
  int32_t y_stride, reserved_;
} lirec_margin_loss_args;
int lirec_margin_loss(const lirec_margin_loss_args* a, lirec_stream_t stream);

/* SURVEY 8(b)'s K5 boundary call: the output heads, the loss and the heads' data gradient as ONE library call --
 *   out_ints / out_ctx (mlp/model.py:332-336, :205-209)   heads[0..n_heads): Y = A W^T + b          (one grouped launch)
 *   the max-margin loss on those logits (:381-575)         loss: fused forward + d(loss)/d(logits)   (one launch)
 *   dA = dlogits W of every head                           back[0..n_heads), issued with parts = 2   (one grouped launch)
 * i.e. what a host that drives forward, loss and backward together (an inference-free training loop, the recorded step) issues
 * between the gate product and the gate's data gradient.  `loss->ints` / `loss->rels` must be the heads' Y, `back[h].dY` the
 * loss's d_ints / d_rels.  The heads' WEIGHT gradients are not part of it (independent of the chain: lirec_linear_bwd_group
 * with parts = 1, on a second stream).  Three launches, not one: a single-launch form -- one workgroup per clip pulling the
 * 1.2 MB of out_ints.weight through its CU twice -- is slower than these on this machine.  NULL `loss` (evaluation):
 * the heads only. */
int lirec_heads_loss_fwd_bwd(const lirec_linear_fwd_args* heads, const lirec_linear_bwd_args* back, int32_t n_heads,
                             const lirec_margin_loss_args* loss, lirec_stream_t stream);

/* MultiTaskCrossEntropyLoss.forward (mlp/model.py:367-378): mean CE over `ints` rows plus
 * mean CE over the `rels` rows whose label != NR.  class_w ([C]) may be NULL.
 * den_ints / den_rels / dens_dev: the data-parallel form, as lirec_margin_loss_args::batch_divisor -- the denominators of the
 * two means given by the caller (the global batch's sum of target class weights / its count of labelled rows, each divided by
 * world); 0 / NULL: this batch's own (the reference's single-device form).  dens_dev: device float[2] read instead. */
int lirec_ce_loss(const float* ints, int64_t ld_ints, const float* rels, int64_t ld_rels,
                  const int32_t* y, const int32_t* r, const float* class_w,
                  int32_t B, int32_t C, int32_t NR, float* d_ints, int64_t ld_dints,
                  float* d_rels, int64_t ld_drels, float* loss, float* partial,
                  float den_ints, float den_rels, const float* dens_dev, lirec_stream_t stream);

/* ---- optimiser --------------------------------------------------------------
 * torch.optim.Adam(lr, weight_decay) as configured at mlp/model.py:599-601, fused over
 * one flat fp32 buffer: g += wd*p; m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  grad_scale multiplies g first
 * (1/world_size after a summing all-reduce). `step` is 1-based. */
int lirec_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int32_t step,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    float grad_scale, const int64_t* step_dev, lirec_stream_t stream);
/* The same update with the step taken from a device counter of COMPLETED steps that the launch itself maintains: step =
 * *count_dev + 1, and -- advance != 0 -- the workgroup that finishes last stores it back (`ticket`: a device int32, zero on entry,
 * left zero).  For a stream that counts its own steps: a replayed train step leaves its weight-gradient side stream running into
 * the next step, whose first launch advances the shared step counter (lirec_amd/graph.py) -- the side stream's share of the update
 * reads its own.  An update cut into several calls passes advance on the last one only. */
int lirec_adam_step_counted(float* p, const float* g, float* m, float* v, int64_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay,
                            float grad_scale, int64_t* count_dev, int32_t* ticket, int32_t advance, lirec_stream_t stream);
/* `step_dev` (optional, device): when not NULL the 1-based step is read from it by the kernel instead of `step`
 * (bias corrections computed on the device), so that a captured graph advances through the steps.
 * lirec_counter_add: ctr[i] += inc[i] for i < n (n <= 4), one tiny kernel -- the "next step" node of such a graph. */
int lirec_counter_add(int64_t* ctr, const int64_t* inc_host, int32_t n, lirec_stream_t stream);

/* ---- evaluation counters on the device -------------------------
 * Precision.update_probs_max_tracks (utils/evaluation.py:114-176) and, with `rels`,
 * update_probs_max_tracks_rels (:179-271): per clip, padded tracks -> -inf, S = sigmoid; predicted track
 * = argmax_t S[t,y] (+ Q[t,r0] with the "None" column appended as 0, :220-222); joint prediction = first
 * flat argmax over (t,c) of S, or over (t,c,r) of S[t,c]+Q[t,r] (:229-235); then the two-pass bookkeeping over
 * the two ground-truth tracks (:150-175, :239-270).  The reference copies the logits to the host every batch
 * (mlp/test.py:50-67) and tiles a (B*T, C, NR) tensor in numpy; here one workgroup per clip adds into
 *   counters[0..6] = total, total_cl, total_rels, top1, trks_top1, cls_top1, rels_top1   (int64, device)
 * which the eval loop reads once at the end.  mem, y, r, g: as for the margin loss; with loader_types f64 /
 * i64 are read in place.  just_zeros: [B] bytes, NULL = none. */
typedef struct {
  const float* ints; int64_t ld_ints;     /* [B*T, C] logits */
  const float* rels; int64_t ld_rels;     /* [B*T, NR] or NULL */
  const float* mem;                       /* [B, T] 0/1 or NULL */
  const int32_t* y;                       /* [B] */
  const int32_t* r;                       /* [B, T] (NR = None); required with rels */
  const int32_t* g;                       /* [B, 2] gt_tracks */
  const uint8_t* just_zeros;              /* [B] or NULL */
  int64_t* counters;                      /* [8] in/out (accumulated) */
  int32_t B, T, C, NR;
  int32_t loader_types, reserved_;
} lirec_eval_args;
int lirec_eval_max_tracks(const lirec_eval_args* a, lirec_stream_t stream);

/* ---- feature assembly ---------------------------------------------
 * The reference's loader tiles every row of the (B, T, R+1, D) feature block on the host as
 *   [ clip piece (text | clip-visual) | track-1 piece | track-2 piece ]
 * (mixed_utils/classification_dataloader.py:336-349, :419, :477-478, :496-497, :531-533, :558-565;
 * mixed_utils/mixed_features.py:115-125) and the block crosses PCIe every step (mlp/model.py:279-280).  Here only the
 * de-duplicated piece tables and an index cross it:
 *   out[row, :] = [ clip[index[row,0], :clip_dim] | track[index[row,1], :track_dim] | track[index[row,2], :track_dim] ]
 * with zeros for a negative index.  clip / track: fp32 (table_f64 = 0) or float64 tables (row strides ld_* in
 * elements); out: fp32 [rows, clip_dim + 2 track_dim] (ld_out in elements); clip_dim and track_dim multiples of 4,
 * tables / out 16-byte (float64: 16-byte) aligned. */
int lirec_gather_features(const void* clip, int64_t ld_clip, const void* track, int64_t ld_track, int32_t table_f64,
                          const int32_t* index, int64_t rows, int32_t clip_dim, int32_t track_dim,
                          float* out, int64_t ld_out, lirec_stream_t stream);
/* The same block written as bf16 (round to nearest even; `out`: [rows, clip_dim + 2 track_dim] bf16, ld_out in elements):
 * "bf16 feature storage" (x_bf16 of lirec_embed_fwd_args) fed straight from the piece tables. */
int lirec_gather_features_bf16(const void* clip, int64_t ld_clip, const void* track, int64_t ld_track, int32_t table_f64,
                               const int32_t* index, int64_t rows, int32_t clip_dim, int32_t track_dim,
                               void* out, int64_t ld_out, lirec_stream_t stream);

/* ---- raw feature pooling ------------------------------------------
 * What the reference's feature classes compute with numpy for a clip or a track that is not in their cache:
 * lirec_grid_pool: out[o, c] = max over the elements e in [estart[o], estart[o+1]) of the MEAN of
 *   grid[frame_e, c, y0_e:y1_e, x0_e:x1_e]   (boxes[e] = {frame, y0, y1, x0, x1}, half-open ranges)
 * = the clip-visual feature with one full-grid box per frame of the clip's frame range (visual_utils/
 * visual_features.py:60-103 spatial mean, mixed_utils/mixed_features.py:54 temporal max), and the track feature with the
 * person box of every track element (visual_features.py:105-134, mixed_features.py:104-105).  grid: fp32 [F, C, H, W];
 * a frame outside [0, F) contributes a zero row (:129), an empty box NaN, no element zeros.  The mean follows numpy's
 * float32 pairwise summation order: results equal np.mean bit for bit.
 * lirec_rows_max: out[o, :] = max over the rows idx[e], e in [estart[o], estart[o+1]), of src [*, ld] (the text feature
 * of a clip: tokens in its time range, text_utils/text_features.py:140-165, then np.max, mixed_features.py:61). */
int lirec_grid_pool(const float* grid, int32_t F, int32_t C, int32_t H, int32_t W, const int32_t* boxes,
                    const int32_t* estart, int32_t n_out, float* out, int64_t ld_out, lirec_stream_t stream);
int lirec_rows_max(const float* src, int64_t ld, const int32_t* idx, const int32_t* estart, int32_t n_out, int32_t dim,
                   float* out, int64_t ld_out, lirec_stream_t stream);

/* ---- utilities ---------------------------------------------------------------- */
/* float64 -> float32 (the DataLoader delivers float64, mlp/model.py:279 `.float()`) */
int lirec_cast_f64_f32(const double* src, float* dst, int64_t n, lirec_stream_t stream);
/* keep[row, col] (uint8) of one dropout site, for tests */
int lirec_dropout_mask(uint8_t* keep, int32_t rows, int32_t cols, const lirec_dropout* drop,
                       int32_t site, lirec_stream_t stream);
int lirec_version(void);
/* sizeof() of ABI struct `which` (5 = eval; 0 embed_fwd, 1 embed_bwd, 2 margin_loss, 3 dropout,
 * 4 rowsel) so a binding can verify its mirror; -1 if unknown */
int lirec_abi_sizeof(int which);
/* GEMM core: 0 exact f32-input MFMA   1 one-thread-per-output HIP GEMM (bring-up cross-check)
 *            2 split-precision bf16x3 MFMA (fp32 in/out, ~2^-16 per product, up to 5.3x the f32 core)
 *            3 the bf16 core with ONE pass on the large GEMMs -- layer 1 and its weight gradient, the gate's forward / data /
 *              weight gradients: operands rounded to bf16 once, fp32 accumulate (~2^-9 per operand); every other GEMM as in 2.
 *              BASELINE config 5's arithmetic ("bf16, 32 tracks/clip stress"); outside the 1e-4 parity contract by design,
 *              never the headline (tests/test_gpu_onepass.py states and checks its tolerance) */
int lirec_set_gemm_mode(int mode);
int lirec_get_gemm_mode(void);
/* on != 0: every weight / bias gradient launched from now on OVERWRITES its buffer (dW = ..., db = ...) instead of accumulating
 * (+=).  For a caller that issues zero_grad + forward + backward + step as one unit (the recorded train step): the zeroing pass
 * over the gradient buffer (76 MB per step here) is then not needed.  Every parameter must receive exactly one gradient launch
 * per step (true for the models of this library; lirec_amd.graph checks it once per recording).  PER CALLING THREAD (like
 * lirec_grad_overwrite_conflicts): a backward that runs on another host thread -- an autograd engine thread, a loader thread -- is not
 * switched and accumulates; the Python side only enters the mode where loss.backward() takes the direct path on the calling thread
 * (lirec_amd.graph refuses to record anything else).  Default off. */
int lirec_set_grad_overwrite(int on);
/* The number of gradient buffers that were the target of MORE than one weight- / bias-gradient launch since the mode was last
 * switched on (calling thread): non-zero means overwriting would lose contributions -- keep the zeroing pass. */
int lirec_grad_overwrite_conflicts(void);
const char* lirec_error_string(int code);
/* Optional device scratch for split-K: the GEMMs whose output is small but whose reduction is deep
 * (weight gradients dW = dY^T X over all rows, the skinny head GEMMs) cut K into chunks so that
 * they fill the chip; partial tiles go to this buffer and a fixed-order reduce kernel sums them
 * (bitwise reproducible, no atomics).  Without scratch (NULL / 0) nothing is split.  The buffer is
 * used by the next GEMM launch, so all launches must be on one stream while it is registered.
 * 128 MiB covers the full-size model. */
int lirec_set_scratch(void* ptr, int64_t bytes);

/* Contexts.  The GEMM core (lirec_set_gemm_mode), the split-K scratch (lirec_set_scratch) and the diagnostic switches
 * (lirec_debug_set) are state of a context, and every thread has a current one -- the default context until
 * lirec_ctx_set_current(ctx) is called on that thread (NULL switches back).  All launches that use one context's scratch
 * must be on one stream; a host with two concurrent streams (training + evaluation) gives each its own context.
 * A new context starts with the default context's GEMM core and no scratch. */
int lirec_ctx_create(lirec_ctx_t* out);
int lirec_ctx_destroy(lirec_ctx_t ctx);
int lirec_ctx_set_current(lirec_ctx_t ctx);
lirec_ctx_t lirec_ctx_get_current(void);
/* Command lists: a recorded step re-issued from C.  Between lirec_record_begin() and lirec_record_end() every kernel launch,
 * memset and lirec_stream_wait made by the calling thread through this library is executed as usual AND appended -- kernel,
 * grid, arguments by value, stream -- to a list; lirec_cmdlist_replay(list, from, to) re-issues commands [from, to) (to < 0:
 * to the end) as plain launches on the streams they were recorded on.  The caller keeps every buffer the recorded calls
 * were given alive and at its address, and passes per-step scalars through device memory (lirec_dropout.seed_dev,
 * lirec_adam_step's step_dev, lirec_counter_add).  lirec_record_mark() = number of commands recorded so far: the host notes
 * it where it has work of its own to do between commands at replay time (the gradient all-reduces of a data-parallel step).
 * Replaces nothing in the reference (a plain eager loop, mlp/train.py:57-63); host time per step 0.9 ms -> ~0.1 ms with the
 * eager loop's kernel timeline (lirec_amd/graph.py, RecordedTrainStep). */
typedef struct lirec_cmdlist* lirec_cmdlist_t;
int lirec_record_begin(void);
int32_t lirec_record_mark(void);
int lirec_record_end(lirec_cmdlist_t* out);
int32_t lirec_cmdlist_size(lirec_cmdlist_t list);
int lirec_cmdlist_replay(lirec_cmdlist_t list, int32_t from, int32_t to);
/* Diagnostics: the same replay with the stream of command `lag_at` held back by `ticks` of the 100 MHz clock in front of that
 * command (tests: a recorded step's cross-stream dependencies must be events, not timing); lirec_cmdlist_command: the stream
 * command i is issued on (a stream wait: the signalling stream) and its kind (0 launch / memset, 1 stream wait, 2 profiling bracket). */
int lirec_cmdlist_replay_lagged(lirec_cmdlist_t list, int32_t from, int32_t to, int32_t lag_at, int64_t ticks);
int lirec_cmdlist_command(lirec_cmdlist_t list, int32_t i, lirec_stream_t* stream, int32_t* kind);
int lirec_cmdlist_destroy(lirec_cmdlist_t list);
/* `waiter` waits for everything enqueued on `signaller` so far (event record + stream wait; recorded like a launch). */
int lirec_stream_wait(lirec_stream_t waiter, lirec_stream_t signaller);
/* The same for n <= 4 waiters behind ONE event record on `signaller` (a record costs the signalling stream a ~6 us bubble on
 * this runtime; a fork that orders two side streams behind the same point pays it once). */
int lirec_stream_wait_many(const lirec_stream_t* waiters, int32_t n, lirec_stream_t signaller);
/* hipMemsetAsync(p, 0, bytes) on `stream`, recorded like a launch (optimizer.zero_grad inside a recorded step). */
int lirec_memset_zero(void* p, int64_t bytes, lirec_stream_t stream);
/* One kernel: zero `bytes` bytes at `p` (16-byte aligned) and ctr[i] += inc[i] for i < n <= 4 (n = 0: no counters).  The
 * first launch of a replayed step: optimizer.zero_grad() and lirec_counter_add() without a second launch between them. */
int lirec_zero_count(void* p, int64_t bytes, int64_t* ctr, const int64_t* inc, int32_t n, lirec_stream_t stream);

/* Diagnostics (current context): `ablate` = k-loop ablation mask (4: no k-loop; planes kernels 16: no LDS-DMA, 32: no LDS
 * reads / MFMAs; 8: planes path off) -- results are garbage, only the timing is meaningful; 64: run-time split-K rule for the
 * row-compacted dW1, 128: split-K for GEMMs with an epilogue (both correct, both measured null / negative); 256 / 512: the
 * pre-round-2 tile orders of grouped NT / TN launches (correct; more L2 misses: see DESIGN 4.4); 4096: the split-K reduce kernel that
 * walks the problems inside every thread (same results, slower); 131072: lirec_linear_bwd_group puts a ~2 ms idle kernel in front of
 * its weight-gradient launches, 262144: the same in front of a row staging pass run ahead of its step (lirec_embed_fwd parts = 4) -- streams
 * made to lag on purpose: tests/test_gpu_recorded_bench_shape.py; `force_cfg` >= 0 forces one
 * tile configuration of the on-the-fly cores, -1 = automatic.  Never set by the product path. */
int lirec_debug_set(int ablate, int force_cfg);

/* Per-call-site timing with HIP events recorded on the launch stream (off by default).
 * enable(1) clears the accumulators; read() waits for the recorded events and returns, for
 * `site` in [0, lirec_profile_sites()): device milliseconds, launches, algorithmic FLOPs
 * (GEMM sites) and algorithmic bytes (HBM-bound sites) accumulated since enable(1). */
int lirec_profile_enable(int on);
int lirec_profile_sites(void);
const char* lirec_profile_site_name(int site);
int lirec_profile_read(int site, double* ms, int64_t* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* LIREC_HIP_H */
