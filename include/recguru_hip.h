/* recguru_hip.h -- C ABI of librecguru_hip.so (gfx950 / MI355X).
 *
 * The reference (Chain123/RecGURU) has no operator / FFI layer: its hot path is torch.nn modules
 * (SURVEY.md 8b).  This header is therefore the native boundary the build defines: one launcher
 * per fused op of the AE+GAN training step, forward and backward.  Each entry point names the
 * reference code it replaces (paths relative to /root/reference/GURU).
 *
 * Conventions: plain pointers and sizes, no torch types.  All pointers are DEVICE pointers unless
 * a parameter is documented as host.  The caller owns every buffer; kernels never allocate, never
 * synchronise and run on the stream passed as `void* stream` (a hipStream_t; NULL = default).
 * `dtype` selects the activation/operand tier: RG_F32 (exact f32 MFMA, parity tier) or RG_BF16
 * (bf16 MFMA operands, f32 accumulation/softmax/LayerNorm).  Parameters, gradients, optimizer
 * state, losses and statistics are always f32.  Return 0 on success, a negative RG_ERR_* code
 * otherwise; rg_last_error() gives the message (thread-local).
 */
#ifndef RECGURU_HIP_H
#define RECGURU_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RG_F32 0
#define RG_BF16 1
#define RG_X3 2 /* "bf16x3": buffers are f32 exactly as for RG_F32 (the two tiers can be mixed call by call); every MFMA operand
                   is split into a bf16 pair hi + lo and a product is three bf16 MFMAs (lo.hi + hi.lo + hi.hi, f32 accumulate):
                   ~16 significant bits per operand at 3/16 of the exact-f32 matrix-pipe time.  Taken by the GEMM-class entry
                   points (rg_gemm_nt, rg_gemm_tn, rg_attn_fwd / bwd, rg_post_attn_fwd, rg_ffn_bwd_data, rg_attn_out_bwd,
                   rg_disc_rows); everything else has no matrix product and is called with RG_F32. */

#define RG_ERR_INVALID (-1)
#define RG_ERR_UNSUPPORTED (-2)
#define RG_ERR_HIP (-3)

const char* rg_last_error(void);
int rg_version(void);

/* ---- deterministic reductions (SURVEY.md 5.2 "offer sorted segmented-reduce mode for tests"; no reference counterpart: torch's own
 * index_add / scatter backward on the reference's path are float atomics too, GURU/AutoEnc4Rec_cross.py:98-102 through autograd).
 * librecguru_hip.so sums parameter gradients, loss sums and the discriminator's scalars with float atomics (order = workgroup
 * schedule).  librecguru_hip_det.so -- the same sources built with -DRG_DETERMINISTIC -- sends every such add to a 64-bit fixed-point
 * shadow instead (integer atomics: any order gives the same bits).  Destinations must then lie in one of two arenas the host
 * registers: a float buffer [bytes] at fbase (still readable by the kernels) whose shadow of long long [2 * bytes] starts at sbase,
 * in units of 2^-bits; the host converts shadow -> float after the launch (recguru_amd/hip.py `_DetArena`).
 *   rg_det_enabled()  number of translation units built deterministic (0: this is the float-atomic library)
 *   rg_det_set_arenas()  arena 0: gradients, arena 1: loss sums / scalars (current device)
 *   rg_det_fault(clear)  0 = every add so far went to a shadow; 1 = a destination outside the arenas was added to with a float
 *                        atomic (host omission); 2 = a contribution outside the fixed-point range (or NaN); -1 = HIP error */
int rg_det_enabled(void);
int rg_det_set_arenas(void* fbase0, void* sbase0, unsigned long long bytes0, int bits0,
                      void* fbase1, void* sbase1, unsigned long long bytes1, int bits1);
int rg_det_fault(int clear);

/* ---- generic Linear-shaped GEMMs --------------------------------------------------------------
 * replaces nn.Linear forward / backward in MultiHeadAttention (Transformer/transformer.py:136-161),
 * PositionWiseFeedForwardNet (:173-188) and Discriminator (tools/utils.py:41-57). */
#define RG_PRO_NONE 0
#define RG_PRO_GELU 1 /* operand = gelu_tanh(stored pre-activation), transformer.py:81-84 */

#define RG_EPI_NONE 0        /* C = acc + bias */
#define RG_EPI_RELU 1        /* C = max(acc + bias, 0)                       tools/utils.py:43 */
#define RG_EPI_MUL_POSMASK 2 /* C = (acc + bias) * [aux > 0]   (ReLU backward / GP mask chain) */
#define RG_EPI_GELU_GRAD 3   /* C = (acc + bias) * gelu'(aux)                (FFN backward)   */
#define RG_EPI_ADD 4         /* C = acc + bias + aux                         (residual grads) */
#define RG_EPI_RESID_LN 5    /* C = LayerNorm(acc + bias + aux; gamma, beta, eps) * rowmask;
                                rstd_out[m] saved.  transformer.py:161,188 and :594,:539      */
#define RG_EPI_DROP_GELU 6   /* weight-stationary path only: C = dropout(acc + bias; drop_p, drop_seed) AND C2 = gelu(C as stored) --
                                the FFN's first product with both of its consumers' operands in one epilogue (transformer.py:181-184) */

typedef struct {
  const void* A; int lda;     /* [M,K] dtype */
  const void* W; int ldw;     /* [N,K] dtype (torch Linear layout [out,in]) */
  const float* bias;          /* [N] or NULL */
  void* C; int ldc;           /* [M,N] dtype, or f32 if c_is_f32 */
  int c_is_f32;
  int M, N, K;                /* K multiple of 32 */
  int prologue, epilogue;
  const void* aux; int ldaux; /* [M,N] dtype */
  const float* gamma; const float* beta; const float* rowmask; float* rstd_out; float ln_eps;
  int debug_ablate;           /* 0 in production; tools/kbench.py phase ablation bits */
  /* dropout hooks (all off when 0): */
  float epi_scale;            /* MUL_POSMASK: kept elements are multiplied by it (1/(1-p)); 0 means 1   */
  float epi_nonzero_scale;    /* GELU_GRAD: also multiply by (aux != 0 ? this : 0) -- backward of the
                                 dropout that precedes GELU (transformer.py:182-184), aux = dropped h1 */
  float drop_p; unsigned long long drop_seed;   /* RELU: C = dropout(relu(.)), tools/utils.py:43-44 */
  const int* live16;          /* optional list of live 16-row tiles (rg_live_tiles): rows of the other tiles are not read and
                                 their rows of C are written as zeros; honoured by the weight-stationary kernel, an error
                                 (RG_ERR_UNSUPPORTED) where only the generic one takes the problem */
  int skip_dead_fill;         /* with live16: 0 = rows of C of the padded tiles are written as zeros; 1 = left UNWRITTEN -- for
                                 outputs whose every consumer is list- or rowmask-driven (the list-driven GEMMs never read
                                 them; rg_attn_bwd takes dctx rows with rowmask == 0 as zero); 2 = written as the BIAS row
                                 (epilogue NONE) -- exact when the caller guarantees that those rows of A are all zero: the
                                 padded positions of a layer input (transformer.py:594 / :539 multiply every layer output by
                                 the pad mask, and the embedding by it, :105), whose Q / K / V rows are the biases */
  /* c_hm_L > 0 (weight-stationary path only: bf16, N = 3 * H * 32 with H % 4 == 0, M = B * c_hm_L): C is written HEAD-MAJOR, as
   * the three tensors q | k | v [B][H][c_hm_L][32] one after the other -- column c of row m = b * L + l goes to
   * (((c / (32 H)) * B + b) * H + (c / 32) % H) * L * 32 + l * 32 + c % 32.  A head's K (V, Q) tile is ONE contiguous run of
   * L * 64 bytes, which the attention kernels stage by LDS-DMA (rg_attn_args.qkv_hm).  ldc is ignored. */
  int c_hm_L;
  void* C2;                   /* RG_EPI_DROP_GELU: second output [M,N] of the tier dtype, row pitch ldc */
  int w_packed;               /* RG_X3 with K = 256 only (the weight-stationary kernel streams its K x 128 weight slice per row tile and splits it there):
                                 W is the PRESPLIT fragment-packed copy of the [N,K] matrix (rg_cast RG_CAST_PACK | RG_CAST_SPLIT, K = ldw) -- the slice
                                 stays in registers, no split arithmetic per tile.  Refused (RG_ERR_UNSUPPORTED) where another kernel would take the problem. */
} rg_gemm_nt_args;
int rg_gemm_nt(const rg_gemm_nt_args* args /* host */, int dtype, void* stream);

typedef struct {
  const void* Y; int ldy;     /* [T,N1] dtype  (upstream gradient) */
  const void* X; int ldx;     /* [T,N2] dtype  (layer input; prologue_x applied) */
  float* dW; int lddw;        /* [N1,N2] f32, ACCUMULATED into (atomics) */
  float* colsum;              /* [N1] f32 += sum_t Y[t,:]  (bias gradient) or NULL */
  int T, N1, N2;
  int prologue_x;
  float scale;
  int splits;                 /* token splits (0 = auto) */
  int use_tr;                 /* bf16: 1 = ds_read_tr16_b64 fragments, 0 = scalar LDS reads */
  const int* live16;          /* optional list of live 16-row tiles (rg_live_tiles): only those rows are summed (the others
                                 carry zero Y); honoured by the big-shape kernel, ignored (all rows summed) otherwise */
  float* partials;            /* optional caller-owned device scratch of >= rg_gemm_tn_workspace() bytes: the big-shape kernel
                                 then writes each workgroup's partial dW with plain coalesced stores and a second launch
                                 sums them into dW (256 x |dW| of 64-byte-segment float atomics ran at a fraction of the
                                 memory-side atomic rate and cost more than the GEMM itself).  NULL: atomics.  Must not be
                                 shared by calls that may run concurrently on different streams. */
  int colsum_T;               /* > 0: colsum sums rows t < colsum_T only (the fused discriminator stacks rows whose bias
                                 gradient is zero by construction behind the ones that have one); 0: all T rows */
} rg_gemm_tn_args;
int rg_gemm_tn(const rg_gemm_tn_args* args /* host */, int dtype, void* stream);
/* bytes of `partials` scratch rg_gemm_tn can use for these arguments (0: the kernel it would run has no use for it) */
size_t rg_gemm_tn_workspace(const rg_gemm_tn_args* args /* host */, int dtype);
/* The FOUR weight-gradient products of one transformer layer's backward (autograd of the nn.Linear members of
 * MultiHeadAttention and PositionWiseFeedForwardNet, Transformer/transformer.py:136-188) in ONE launch + one reduce launch:
 *   slot 0  dW2   [128 x 512] += dl2^T gelu(h1)   (prologue_x = RG_PRO_GELU)      slot 1  dW1 [512 x 128] += dh1^T y
 *   slot 2  dWqkv [384 x 128] += dqkv^T x                                          slot 3  dWo [128 x 128] += dz^T ctx
 * p[4] in that order (T = 0: empty slot), RG_BF16 or RG_X3, T >= 8192, every used slot with `partials` of
 * rg_gemm_tn_layer_workspace(slot, wgs[slot]) bytes; wgs[i] = workgroups dealt to slot i (sum <= 256).  colsum / live16 as in
 * rg_gemm_tn.  Same sums as four rg_gemm_tn calls (different f32 summation order). */
int rg_gemm_tn_layer_supported(const rg_gemm_tn_args* p /* host, 4 */, const int* wgs /* host, 4 */, int dtype);
size_t rg_gemm_tn_layer_workspace(int slot, int wgs);
int rg_gemm_tn_layer(const rg_gemm_tn_args* p /* host, 4 */, const int* wgs /* host, 4 */, int dtype, void* stream);
/* Plan queries (no launch): the name of the kernel rg_gemm_nt / rg_gemm_tn would run for these arguments --
 * "gemm_ws_kernel<K/128,N/128>" (persistent weight-stationary), "gemm_nt_kernel<dtype,NTW>" (generic tiles),
 * "gemm_tn_dma_kernel<N1,N2>" / "gemm_tn_big_kernel<N1,N2>" (big shapes: LDS-DMA ring / register-staged) or
 * "gemm_tn_kernel<dtype>" -- written NUL-terminated into name[cap].  Used by the
 * host-side profiler so that per-kernel times match rocprofv3's kernel names. */
int rg_gemm_nt_plan(const rg_gemm_nt_args* args /* host */, int dtype, char* name, int cap);
int rg_gemm_tn_plan(const rg_gemm_tn_args* args /* host */, int dtype, char* name, int cap);


/* ---- attention core (d_k = d_v = 32) --------------------------------------------------------------
 * replaces ScaledDotProductAttention.forward (Transformer/transformer.py:119-129) + the mask
 * builders get_attn_pad_mask / get_attn_subsequent_mask (:54-78) + head repeat (:157).
 * masked(q,key) = (key_ids[b,key] == pad_value) || (causal && key > q); masked scores are REPLACED
 * by -1e9 before softmax.  qkv: [B,L,3*H*32] (Q|K|V), ctx: [B,L,H*32], lse: [B,H,L] f32. */
typedef struct {
  const void* qkv;
  const int64_t* key_ids; int64_t pad_value; int causal;
  void* ctx; float* lse;      /* lse may be NULL (inference) */
  int B, L, H, dk;
  float scale;                /* 1/sqrt(d_k), transformer.py:120 */
  float drop_p; unsigned long long seed;   /* attention-probability dropout, transformer.py:126-127 */
  /* optional [B*L] f32: rows with rowmask == 0 are the padded positions whose layer output the caller multiplies by
   * the pad mask (transformer.py:594 / :539) -- a 16-query tile made only of such rows is skipped (ctx rows = 0). */
  const float* rowmask;
  /* x-input form (inference only, lse == NULL): x != NULL replaces qkv by the LAYER INPUT x [B,L,d] and the kernel
   * projects its head's Q, K, V itself with wqkv [3P,d] (rows Q | K | V, torch Linear layout, dtype) and bqkv [3P] f32 --
   * MultiHeadAttention.forward's WQ / WK / WV (transformer.py:151-156) fused into the attention core.  x_masked: the
   * caller guarantees that the rows of x at positions with rowmask == 0 are all zero (their K / V are then the bias rows,
   * not computed).  rg_attn_fwd_x_supported(d, dtype, drop_p): bf16, d = 128, dropout off or 0.5. */
  const void* x; const void* wqkv; const float* bqkv; int d;
  /* x_masked, qkv form: the caller guarantees that the K and V rows at positions with rowmask == 0 are all IDENTICAL (the
   * projection of an all-zero layer-input row is the bias row).  A non-causal head then evaluates a leading run of such
   * keys (left padding) once: same max, (count of kept copies) more terms in the row sum and the context.
   * x_masked == 2: additionally the rows of qkv in 16-row tiles (of the flattened [B*L] rows) made of such positions
   * only may be UNWRITTEN (rg_gemm_nt skip_dead_fill = 1 on the projection): the kernel substitutes the bias rows bqkv
   * [3*H*32] f32 for every position with rowmask == 0 instead of reading -- the projection then neither computes nor
   * writes the padded tiles (a third of its traffic at the bench shape). */
  int x_masked;
  /* optional [B] (rg_first_live): index of the first position of each sequence with rowmask != 0 (L if none).  With
   * x_masked == 2 the rows before it are not even addressed (their loads are pointed at that first live row, whose lines
   * are hot, and the data replaced by the bias rows) -- unwritten rows are cold lines, reading them cost more than the
   * projection saved by not writing them. */
  const int* first_live;
  /* qkv_hm = 1 (bf16, qkv form, d_model = 128 / H = 4 not required): qkv is the head-major triple q | k | v, each [B, H, L, 32]
   * (rg_gemm_nt_args.c_hm_L).  pad_rows (required then): [3 * H + 1][32] elements of the tier dtype -- row h: the Q bias of
   * head h, row H + h: its K bias, row 2 H + h: its V bias, row 3 H: zeros.  Rows the kernel must not read from qkv
   * (x_masked == 2: positions before first_live; rows >= L of the padded key range) are fetched from there instead: the
   * K / V tiles are filled by LDS-DMA (one contiguous 1 KB per wave instruction), with nothing staged through registers. */
  int qkv_hm;
  const void* pad_rows;
} rg_attn_args;
int rg_attn_fwd(const rg_attn_args* args /* host */, int dtype, void* stream);
int rg_attn_fwd_x_supported(int d, int dtype, float drop_p);

typedef struct {
  const void* qkv; const void* dctx; const void* ctx; const float* lse;
  const int64_t* key_ids; int64_t pad_value; int causal;
  void* dqkv;                 /* [B,L,3*H*32] dtype, fully overwritten */
  int B, L, H, dk;
  float scale;
  float drop_p; unsigned long long seed;   /* must equal the forward's */
  const float* rowmask;       /* optional, as in the forward: dctx of those rows is zero, their tiles are skipped */
  /* x_masked == 2 (as in rg_attn_args): the Q / K / V rows at positions with rowmask == 0 are NOT read -- the bias rows
   * bqkv [3*H*32] f32 (Q | K | V) stand in, bit for bit what the projection of an all-zero input row gives */
  const float* bqkv; int x_masked;
  const int* first_live;      /* optional [B], as in rg_attn_args */
  int qkv_hm;                 /* 1: qkv is the head-major triple q | k | v, each [B, H, L, 32] (rg_attn_args.qkv_hm); dqkv stays
                               * token-major [B, L, 3P].  bf16 tier. */
} rg_attn_bwd_args;
int rg_attn_bwd(const rg_attn_bwd_args* args /* host */, int dtype, void* stream);

/* ---- K1: embedding gather + positional add + pad mask, and its scatter-add backward ---------------
 * replaces nn.Embedding lookups (AutoEnc4Rec.py:180,192; AutoEnc4Rec_cross.py:98,101,124,127) fused
 * with PositionalEncoding.forward (Transformer/transformer.py:104-106).
 * table: [rows,d] dtype (operand-tier copy of the f32 master), pe: [>=L,d] f32, ids: [ntok] i64,
 * mask: [ntok] f32, out: [ntok,d] dtype.  Backward accumulates dx*mask into the dense f32 gradient
 * dE[rows,d]; skip_row (e.g. padding_idx 0 of AutoEnc4Rec.py:153) receives nothing (-1 = none). */
int rg_embed_pe_fwd(const void* table, const float* pe, const int64_t* ids, const float* mask, void* out,
                    long long ntok, int L, int d, float drop_p, unsigned long long seed, int dtype, void* stream);
/* The same, told how many rows the table has (0 = unknown).  Results are identical; the launcher uses it for its STORE POLICY only: output
 * rows leave by nontemporal stores when table + output exceed the 256 MB Infinity Cache (the 1 GiB config-5 table: 0.57 -> 0.68 of 8 TB/s),
 * by ordinary stores otherwise.  rg_embed_pe_fwd decides from the output size alone. */
int rg_embed_pe_fwd_rows(const void* table, long long table_rows, const float* pe, const int64_t* ids, const float* mask, void* out,
                         long long ntok, int L, int d, float drop_p, unsigned long long seed, int dtype, void* stream);
/* The same with a second, bf16 copy of the output rows, out2 [ntok,d] (NULL: none) -- the "mixed" tier (f32 forward tensors, bf16
 * backward operands: DESIGN.md 2) keeps the layer input twice, f32 for the forward and bf16 as the X operand of the first layer's
 * weight-gradient product.  out2 needs d in {128, 256}. */
int rg_embed_pe_fwd2(const void* table, const float* pe, const int64_t* ids, const float* mask, void* out, void* out2,
                     long long ntok, int L, int d, float drop_p, unsigned long long seed, int dtype, void* stream);
/* Split-residual form (bf16 tier, see rg_post_attn_args.x_lo): the rows are gathered from the F32 MASTER table
 * (table_f32 [rows,d]; the embedding feeds the residual stream directly, so rounding the table to bf16 first would cap its
 * precision at 8 bits) and leave as the pair out = bf16(v), out_lo = bf16(v - out). */
int rg_embed_pe_fwd_split(const float* table_f32, const float* pe, const int64_t* ids, const float* mask, void* out,
                          void* out_lo, long long ntok, int L, int d, float drop_p, unsigned long long seed, void* stream);
int rg_embed_scatter_bwd(const void* dx, const int64_t* ids, const float* mask, float* dE, long long ntok, int d,
                         long long skip_row, float drop_p, unsigned long long seed, int dtype, void* stream);
/* Same result as rg_embed_scatter_bwd with the gradient rows summed per 64-row bin of the table in LDS (counting sort of
 * the positions by bin, csrc/loss.hip) instead of one atomic row per live position.  table_rows = rows of dE.
 * workspace >= rg_embed_scatter_binned_workspace() bytes (0: shape not supported -- d not in {64,128,256}, > 8192 bins). */
size_t rg_embed_scatter_binned_workspace(long long ntok, int d, long long table_rows);
int rg_embed_scatter_bwd_binned(const void* dx, const int64_t* ids, const float* mask, float* dE, long long ntok, int d,
                                long long table_rows, long long skip_row, float drop_p, unsigned long long seed,
                                void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* drop_p / seed: nn.Dropout of PositionalEncoding.forward (transformer.py:106); the mask is a stateless
 * hash of (seed, element index), regenerated by the backward.  drop_p = 0 disables it. */

/* ---- LayerNorm backward (from the saved output y and rstd) ---------------------------------------
 * replaces autograd of nn.LayerNorm(d, eps=1e-8) at transformer.py:161,188 incl. the row mask of
 * :594/:539.  dz fully overwritten; dgamma/dbeta ACCUMULATED (atomics). */
typedef struct {
  const void* dy; const void* y; const float* rstd; const float* gamma; const float* beta;
  const float* rowmask;       /* [M] or NULL */
  void* dz; float* dgamma; float* dbeta;
  long long M; int N; int ld;
  void* dz_drop; float drop_p; unsigned long long drop_seed;  /* optional 2nd output dz * dropmask/(1-p):
                                 backward of the dropout on the l2 output (transformer.py:186-187) */
  const int* live16;          /* optional list of live 16-row tiles (rg_live_tiles): only the rows of listed tiles are
                                 processed; the rows of dz / dz_drop of the other (fully padded) tiles stay UNWRITTEN --
                                 for callers whose consumers of dz are all list-driven */
  float* dz_colsum;           /* optional [N] f32 += column sums of dz (the bias gradient of whatever produced the LN input:
                                 saves a separate pass over dz) */
  float* partials;            /* optional caller-owned device scratch of >= rg_ln_bwd_workspace(M, N) bytes: per-block column
                                 sums go there and a second launch adds them to dgamma / dbeta (instead of one float atomic
                                 per block and column onto the same 2N addresses).  Not to be shared across streams. */
} rg_ln_bwd_args;
int rg_ln_bwd(const rg_ln_bwd_args* args /* host */, int dtype, void* stream);
size_t rg_ln_bwd_workspace(long long M, int N);

/* ---- collapsed decoder cross-attention (quirk Q1) ---------------------------------------------------
 * MultiHeadAttention(Q=x, K=V=rep(u)) of transformer.py:259 with AutoEnc4Rec_cross.py:122: context is
 * WV u + bV for every query, so the block is y = LayerNorm(x + o[b]) with o = linear(WV u + bV).
 * rg_bcast_add_ln: x [B*L,N] dtype, o [B,N] f32 -> y dtype, rstd f32.  rg_seq_sum: out[b,:] = sum_t x[b,t,:]. */
int rg_bcast_add_ln(const void* x, const float* o, const float* gamma, const float* beta, void* y, float* rstd,
                    long long M, int L, int N, float eps, int dtype, void* stream);
int rg_seq_sum(const void* x, void* out /* [B,N] dtype */, int B, int L, int N, int dtype, void* stream);
/* out[i] = (ids[i] != pad) as f32: get_pad_mask (gan_training.py:347-350) and its inline copies (:399-401, tools/utils.py:76) */
int rg_pad_mask(const int64_t* ids, int64_t pad, float* out, long long n, void* stream);
/* x_last[b,:] = x[b,L-1,:] ([B,d] dtype), m_last[b] = rowmask[b*L + L-1] (may be NULL): the rows the last encoder layer
 * evaluates (every hot-path caller reads enc_outputs[:, -1, :], AutoEnc4Rec_cross.py:122,154) */
int rg_last_rows(const void* x, const float* rowmask, void* x_last, float* m_last, int B, int L, int d, int dtype, void* stream);

/* ---- discriminator / W-GAN gradient-penalty helpers ----------------------------------------------
 * tools/utils.py:41-57 and gan_training.py:38-55 (closed-form double backward). */
int rg_colsum(const void* x, const void* aux /* or NULL */, const float* coef /* [M] or NULL */, float* out,
              long long M, int N, int ld, float scale, int dtype,
              void* stream);                     /* out[n] += scale*sum_m coef[m]*x[m,n]*[aux[m,n]>0] */
int rg_outer_posmask(const float* coef /* [M] or NULL */, const float* w /* [N] */, const void* aux, void* out,
                     long long M, int N, float scale, int dtype,
                     void* stream);                             /* out = scale*coef[m]*w[n]*[aux>0]     */
/* Fused discriminator + gradient penalty (csrc/disc.hip): ONE row-parallel launch walks, per tile of rows and with the
 * activations in LDS, the forward MLP (tools/utils.py:41-57), the backward chain of the W-loss rows (stacked
 * [real; fake], d loss / d D(x_r) = coef_real | coef_fake) and, when alpha != NULL, the whole gradient-penalty branch of
 * calc_gradient_penalty (gan_training.py:38-55) in closed form (interpolation, forward, dD/dxhat, penalty, second-order
 * chain).  It leaves the row-stacked operand pairs of the weight gradients in HBM -- rows [0, 2B) from the W rows,
 * rows [2B, 3B) from the GP rows:
 *     dW1 += Y1^T X1 ([3B,n1] x [3B,d]),  dW2 += Y2^T X2 ([3B,n2] x [3B,n1]),  dW3 += Y3^T X3 ([3B,n3] x [3B,n2])
 * (three rg_gemm_tn calls by the caller, with colsum = db_i and colsum_T = 2B: only the W rows have a bias gradient;
 * T = 2B rows when alpha == NULL) -- and accumulates itself (f32 atomics) dw4, db4 and scalars[0] += mean(D(real)),
 * scalars[1] += mean(D(fake)), scalars[2] += gp_coef * mean((|g|-1)^2).
 * Weights in operand dtype, FRAGMENT-PACKED (rg_cast with RG_CAST_PACK): W_i = pack(W_i [out,in]) and W_it =
 * pack(W_i^T [in,out]); biases, w4 f32.  drop_p: Dropout(0.2) of
 * the discriminator in train mode (stateless hash of seed_* and the element index); 0 = eval.
 * rg_disc_supported: widths multiples of 32, n2 the widest, tile fits the 160 KB LDS (d_model 32..128 on both tiers,
 * 256 on none: callers fall back to rg_gemm_nt chains). */
typedef struct {
  const void* real; const void* fake;          /* [B,d] dtype */
  const float* alpha;                          /* [B] or NULL (no gradient-penalty rows) */
  const void *W1, *W2, *W3, *W1t, *W2t, *W3t;
  const float *b1, *b2, *b3, *w4, *b4;
  int B, d, n1, n2, n3;
  float drop_p;
  unsigned long long seed_w[3];                /* dropout seeds of the W rows, one per hidden layer */
  unsigned long long seed_g[3];                /* ... of the GP rows */
  float coef_real, coef_fake;                  /* d loss / d D(x_r) for r < B, r >= B */
  float gp_coef;                               /* lambda (x data-parallel scale): penalty = gp_coef * mean((|g|-1)^2) */
  float* out;                                  /* [2B] f32 D(x) or NULL */
  float* scalars;                              /* [3] f32, accumulated */
  void *Y1, *X1, *Y2, *X2, *Y3, *X3;           /* stacked operands, dtype; any may be NULL when no weight gradient is wanted */
  float *db1, *db2, *db3, *dw4, *db4;          /* dw4, db4: accumulated when need_wgrad; db1..db3 unused (rg_gemm_tn colsum) */
  void* dx;                                    /* [2B,d] dtype or NULL: d loss / d x of the W rows */
  void* hscratch;                              /* unused (the masks live in LDS as bit masks) */
  int need_wgrad;
  int debug_ablate;                            /* profiling only (tools/kb_disc.py): 1 no column-sum atomics, 2 weight loads
                                                  of one k-step only, 4 no MFMAs, 8 no flushes to HBM; 0 in production */
  long long* stamps;                           /* profiling only: [grid][16] s_memtime stamps at the stage boundaries, or NULL */
} rg_disc_args;
int rg_disc_supported(int d, int n1, int n2, int n3, int dtype);
int rg_disc_rows(const rg_disc_args* args /* host */, int dtype, void* stream);
int rg_interpolate(const float* alpha, const void* real, const void* fake, void* out, long long B, int d,
                   int dtype, void* stream);                   /* gan_training.py:39-43               */
int rg_gp_penalty(const float* g, void* dg, float* gp, long long B, int d, float lambda, int dtype,
                  void* stream);                               /* gan_training.py:54 + d/dg           */
int rg_sum(const float* x, float* out, long long n, float scale, void* stream); /* out[0] += scale*sum */

/* ---- optimizer ---------------------------------------------------------------------------------
 * torch.optim.Adam as configured at train_gan.py:126-134 / gan_training.py:359 (no amsgrad / decay).
 * step counts from 1.  shadow (optional): operand-tier copy refreshed in the same pass. */
int rg_adam(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long long n, float lr,
            float beta1, float beta2, float eps, int step, void* stream);
/* Multi-tensor Adam: ONE launch for a whole optimizer step.  segs is a DEVICE array; block i updates segment i
 * (callers split large tensors into chunks, e.g. 64K elements).  Same arithmetic as rg_adam with
 * step_lr = lr / (1 - beta1^t) and inv_bc2_sqrt = 1 / sqrt(1 - beta2^t) of the parameter's own step count t. */
typedef struct {
  float* p; const float* g; float* m; float* v;
  long long n;
  float step_lr; float inv_bc2_sqrt;
} rg_adam_seg;
int rg_adam_multi(const rg_adam_seg* segs /* device */, int nsegs, float beta1, float beta2, float eps, void* stream);
/* Same update with the per-parameter STEP COUNT kept in the device-resident table: block b reads segs[b].step (the
 * number of updates this parameter has had), computes the bias corrections 1 - beta^(step+1) itself (in double, like
 * torch.optim.Adam's Python arithmetic), updates its chunk and stores step + 1 back.  Nothing crosses PCIe per
 * optimizer step -- a per-step table upload from pageable host memory blocks the host until the stream drains --
 * and the learning rate (ScheduledOptim changes it every step, transformer.py:43-51) is a launch argument. */
typedef struct {
  float* p; const float* g; float* m; float* v;
  long long n;
  long long step;
} rg_adam_seg_dev;
int rg_adam_multi_dev(rg_adam_seg_dev* segs /* device, updated in place */, int nsegs, double lr, double beta1, double beta2,
                      double eps, void* stream);
/* transpose: bit 0 = dst is the transpose of src; bit 1 (RG_CAST_PACK) = dst holds the logical operand M (= src or src^T)
 * [N][K] as consecutive MFMA fragments: block (n / 16, k / 32) = 64 lanes x 8 elements, lane 16 g + i = M[16 nb + i]
 * [32 kb + 8 g .. + 8], blocks ordered nb-major -- a wave's operand fragment is then one contiguous 1 KB (bf16) read.
 * Needs R, C multiples of 32.  In rg_cast_seg the same bits apply; ld = logical K of the (possibly concatenated) dst,
 * row_off / col_off = n / k offsets (multiples of 16 / 32). */
#define RG_CAST_TRANSPOSE 1
#define RG_CAST_PACK 2
#define RG_CAST_SPLIT 4 /* with RG_CAST_PACK and an f32 destination: the RG_X3 tier's PRESPLIT operand -- the 2 KB slot of a fragment
                           holds the 64 lanes' hi parts bf16(v) (1 KB: 16 bytes per lane), then their lo parts bf16(v - hi) (1 KB).
                           What rg_post_attn_fwd / rg_ffn_bwd_data / rg_attn_out_bwd take as w_packed weights under RG_X3. */
int rg_cast(const float* src, void* dst, int R, int C, int transpose, int dtype, void* stream);
/* Multi-tensor cast: ONE launch refreshes every operand-tier weight copy after an optimizer step (136 per-tensor casts
 * per training iteration otherwise).  Segment s: dst[(r + row_off) * ld + c + col_off] = src[r, c], or with transpose
 * dst[(c + row_off) * ld + r + col_off] = src[r, c]  (a concatenated copy such as the fused QKV weight is several
 * segments into one dst).  tiles: DEVICE int2 array, one entry per workgroup = (segment, 32 x 32 tile index). */
typedef struct {
  const float* src; void* dst;
  int R, C, ld, row_off, col_off, transpose;
} rg_cast_seg;
int rg_cast_multi(const rg_cast_seg* segs /* device */, const int* tiles /* device, 2 ints per workgroup */, int ntiles,
                  int dtype, void* stream);

/* ---- K7/K8: fused gather-dot-loss over the item catalogue --------------------------------------
 * mode RG_LOSS_SAMPLED_CE: AutoEnc4Rec_cross.py:201-215 + tools/lossfunctions.py:36-49 (label 0).
 * mode RG_LOSS_BPR:        tools/utils.py:114-126 + tools/lossfunctions.py:56-72.
 * sums[0] += sum_t mask*loss_t, sums[1] += sum_t mask (caller zeroes sums; loss = sums[0]/sums[1]).
 * aux_tok[t] saves the per-position log-sum-exp (CE) or score margin (BPR) for the backward.
 * Backward: dh fully overwritten; dE ACCUMULATED with gout[0]/sums[1] folded in. */
#define RG_LOSS_SAMPLED_CE 0
#define RG_LOSS_BPR 1
#define RG_LOSS_BPR_SAS 2        /* BPRLoss_sas, tools/lossfunctions.py:79-96 (train_auto.py:26): the two-term form */
typedef struct {
  const void* h;              /* [ntok,d] dtype : decoder states */
  const void* table;          /* [rows,d] dtype */
  const int64_t* pos;         /* [ntok] */
  const int64_t* neg;         /* [ntok,k] */
  const float* mask;          /* [ntok] */
  float* aux_tok;             /* [ntok] */
  float* sums;                /* [2] */
  const float* gout;          /* [1] upstream gradient (backward only) */
  void* dh; float* dE;        /* backward outputs */
  long long ntok; int d; int k; int mode; long long skip_row;
} rg_item_loss_args;
int rg_item_loss_fwd(const rg_item_loss_args* args /* host */, int dtype, void* stream);
int rg_item_loss_bwd(const rg_item_loss_args* args /* host */, int dtype, void* stream);
/* Same outputs as rg_item_loss_bwd, but the table gradient is built by counting-sorting the (position, item)
 * pairs into bins of 64 table rows and accumulating each bin in LDS (one flush per bin chunk) instead of one
 * global atomic row per pair -- the atomic form is bound by the chip-wide float-atomic rate.  table_rows = rows
 * of the table (V+2).  workspace: caller-owned device scratch of >= rg_item_loss_bwd_binned_workspace() bytes
 * (returns 0 when the shape is not supported: d not in {64,128,256}, more than 8192 bins, >= 2^31 pairs). */
size_t rg_item_loss_bwd_binned_workspace(long long ntok, int k, int d, long long table_rows);
int rg_item_loss_bwd_binned(const rg_item_loss_args* args /* host */, long long table_rows, void* workspace,
                            size_t workspace_bytes, int dtype, void* stream);
/* Training form (replaces rg_item_loss_fwd + the rows pass of rg_item_loss_bwd_binned when a gradient will be asked
 * for): ONE walk over the 1+k rows of every position yields the loss sum and, for an upstream gradient of 1, the
 * coefficients coef[t*(1+k)+j] = dloss/dlogit_j * mask_t / sums[1] and dh = sum_j coef_j E[j].  sums[1] must hold
 * the mask count ON ENTRY (rg_sum over mask; under data parallelism the all-reduced count); sums[0] += sum mask*loss.
 * aux_tok, gout, dE are not used.  Supported: d in {64,128,256} and 1+k rows that fit four register batches
 * (rg_item_loss_train_supported).  The backward is rg_scale_dev(dh, gout) + rg_item_loss_scatter_binned, which adds
 * gout[0] * (table gradient of coef) to dE by the same counting sort + LDS accumulation as rg_item_loss_bwd_binned
 * (same workspace size).  Results equal those of the two-call form bit for bit when gout == 1.
 * rg_item_loss_train_supported: 1 = that register form; 2 = the ONLINE form (any k, sampled softmax only, e.g. config-5's
 * k = 1024): running max / sum / sum_j exp(l_j - max) E[j] per position, rows still gathered once; it needs aux_tok [ntok]
 * (receives the log-sum-exp) and leaves the RAW logits in coef; rg_item_loss_scatter_binned given the same aux_tok and sums
 * turns them into the coefficients while it sorts the pairs. */
int rg_item_loss_train_supported(int k, int d);
int rg_item_loss_train(const rg_item_loss_args* args /* host */, float* coef, int dtype, void* stream);
int rg_item_loss_scatter_binned(const rg_item_loss_args* args /* host */, const float* coef, long long table_rows,
                                void* workspace, size_t workspace_bytes, int dtype, void* stream);
/* nn.MSELoss()(a, b) (the overlapped-user term of the generator update, GURU/gan_training.py:28-35,:494-507) and its gradient
 * in one pass: out[0] += mean((a - b)^2); da = 2 (a - b) / n = -db for an upstream gradient of 1 (either may be NULL).
 * a, b, da, db: [n] of dtype, n % 8 == 0. */
int rg_mse(const void* a, const void* b, float* out, void* da, void* db, long long n, int dtype, void* stream);
/* x[0..n) *= s[0] with s on the device; no memory traffic when s[0] == 1.  n % 8 == 0. */
int rg_scale_dev(void* x, long long n, const float* s /* device */, int dtype, void* stream);


/* ---- dropout pieces of the block paths that are not fused (widths other than d_model = 128) ---------------
 * rg_dropout: x [M,N] *= keep(seed, m*N + c) / (1-p) in place (nn.Dropout; the mask is a stateless hash, the same
 *   index space as the fused kernel and rg_ln_bwd's dz_drop, so backward kernels regenerate it from the seed).
 * rg_cross_rows: out [M,N] f32 = bo + sum_h s[m,h] * oh[m/L,h,:] -- the collapsed cross-attention output per row
 *   under attention-map dropout (s from rg_cross_drop_scale). */
int rg_dropout(void* x, long long M, int N, float drop_p, unsigned long long seed, int dtype, void* stream);
/* The wide FFN block (d_model = 256) around the weight-stationary GEMM, which has neither a prologue nor a LayerNorm epilogue
 * (PositionWiseFeedForwardNet.forward, transformer.py:179-188: l1 -> dropout -> GELU -> l2 -> dropout -> + residual -> LayerNorm):
 * rg_dropout_gelu: rg_dropout on x [M,N] in place (drop_p == 0: x untouched) and g [M,N] = gelu(x as stored).  N % 8 == 0.
 * rg_add_drop_ln: y = LayerNorm(x + dropout(z)) * rowmask (rowmask [M] f32 or NULL), rstd [M] for rg_ln_bwd; the mask of z is
 *   the one rg_dropout(z, seed) would apply, z is not rewritten.  x, z, y [M,N] of dtype, N in {128, 256}. */
int rg_dropout_gelu(void* x, void* g, long long M, int N, float drop_p, unsigned long long seed, int dtype, void* stream);
int rg_add_drop_ln(const void* x, const void* z, const float* gamma, const float* beta, const float* rowmask, void* y, float* rstd,
                   long long M, int N, float drop_p, unsigned long long seed, float eps, int dtype, void* stream);
int rg_cross_rows(const float* s, const float* oh, const float* bo, float* out, long long M, int L, int H, int N,
                  void* stream);
/* rg_cross_rows + the LayerNorm that follows it (DecoderLayer's dec_enc_attn under attention-map dropout, transformer.py:160-161,
 * :259) in one row pass: y = LayerNorm(x + bo + sum_h s[m,h] * oh[m/L,h,:]), rstd [M] for rg_ln_bwd; the f32 [M,N] matrix of
 * rg_cross_rows is never formed.  x, y [M,N] of dtype, N in {128, 256}. */
int rg_cross_add_ln(const void* x, const float* s, const float* oh, const float* bo, const float* gamma, const float* beta, void* y,
                    float* rstd, long long M, int L, int H, int N, float eps, int dtype, void* stream);

/* ---- input side: batch assembly and negative sampling on the device ------------------
 * Users are CSR rows: items[offsets[u] .. offsets[u+1]) is user u's chronological item sequence, and
 * excl[excl_off[u] .. excl_off[u+1]) its SORTED, UNIQUE exclusion set (ids in 1..V).  `users` [B] picks the batch.
 * rg_assemble_batch = seq_padding (GURU/data/data_loader.py:25-36) for B users: enc_in [B,L_enc] = left-padded last
 *   L_enc-1 items + eos; dec_in / dec_out [B,L_dec] = the last L_dec entries of [0,0]+enc_in[:-2] / [0,0]+enc_in[1:-1].
 * rg_sample_negatives = pickle_loader.__getitem__'s torch.multinomial(weights, n, replacement=True) with weights
 *   uniform over 1..V and zero on the exclusion set (data_loader.py:298-314): out [B,n]; exact uniform over the
 *   allowed items (the u-th allowed item by binary search, no rejection), counter-based hash of (seed, draw index).
 * rg_sample_negatives_alias = the frequency^0.75-weighted variant (:251-254): Walker alias table over slots ids
 *   (prob [slots] f32, alias [slots] int32), excluded draws rejected. */
int rg_assemble_batch(const int64_t* items, const int64_t* offsets, const int64_t* users, int B, int L_enc, int L_dec,
                      int64_t eos, int64_t* enc_in, int64_t* dec_in, int64_t* dec_out, void* stream);
int rg_sample_negatives(const int64_t* excl, const int64_t* excl_off, const int64_t* users, int B, int n, int64_t V,
                        unsigned long long seed, int64_t* out, void* stream);
int rg_sample_negatives_alias(const float* prob, const int* alias, int64_t slots, const int64_t* excl,
                              const int64_t* excl_off, const int64_t* users, int B, int n, int64_t V,
                              unsigned long long seed, int64_t* out, void* stream);

/* ---- ranking evaluation -------------------------------------------------------------
 * gan_training.py:58-87 (get_scores) + the double argsort of evaluation_2 (:129-132), fused: per user b the
 * scores h[b].E[target[b]] and h[b].E[cand[b,j]] (scores[b,0] = target, optional) and rank[b] = number of
 * candidates scoring strictly higher than the target (= the reference's 0-based rank when scores are distinct).
 * d in {64, 128, 256}; h and table in the operand dtype. */
typedef struct {
  const void* h;             /* [B, d] last recommender-decoder state */
  const void* table;         /* [rows, d] item embeddings */
  const int64_t* target;     /* [B] */
  const int64_t* cand;       /* [B, C] */
  float* scores;             /* [B, 1 + C] or NULL */
  int* rank;                 /* [B] or NULL */
  int B, d, C;
} rg_rank_args;
int rg_rank_scores(const rg_rank_args* args /* host */, int dtype, void* stream);

/* ---- fused post-attention block, forward -------------------------------------------------------------
 * y = LN(ctx.Wo^T + bo + x) [; y = LN(y + o_bcast[b])] ; out = LN(gelu(y.W1^T + b1).W2^T + b2 + y) * rowmask
 * = MultiHeadAttention tail (Transformer/transformer.py:160-161) [+ collapsed dec_enc_attn, :259, Q1]
 *   + PositionWiseFeedForwardNet (:179-188) + the `* pad_mask` of :594 / :539, in ONE launch.
 * Needs d == P == 128 and d_ff % 128 == 0.  *_save / rstd* may be NULL (inference: nothing but `out`
 * is written).  Weights in torch Linear layout, operand dtype; biases / LN parameters f32. */
typedef struct {
  const void* ctx; const void* x;                       /* [M,P], [M,d] dtype */
  const void* Wo; const float* bo; const float* g1; const float* be1;
  const float* o_bcast; const float* gc; const float* bec; int L;   /* optional cross stage: o [M/L, d] f32 */
  const void* W1; const float* b1; const void* W2; const float* b2; const float* g2; const float* be2;
  const float* rowmask;                                  /* [M] or NULL */
  void* out;                                             /* [M,d] dtype */
  void* y_save; float* rstd1; void* y2_save; float* rstd_c; void* h1_save; float* rstd2;
  int M, d, P, dff; float eps;
  /* training-mode dropout (transformer.py:182-183,186-187; all off when drop_p == 0).  With dropout
   * the collapsed cross-attention is no longer one vector per sequence: the attention-map dropout
   * leaves context = s[b,h,q] * (WV u + bV)_h, so the stage takes cross_s [M,H] (rg_cross_drop_scale),
   * cross_oh [M/L, H, d] f32 (per-head output projections) and cross_bo [d] instead of o_bcast. */
  float drop_p; unsigned long long seed_h1; unsigned long long seed_out;
  const float* cross_s; const float* cross_oh; const float* cross_bo; int H;
  /* optional: the list of live 16-row tiles from rg_live_tiles.  Padded row tiles are then compacted away -- a work
   * tile is 4 consecutive LIST entries instead of 64 consecutive rows -- and the rows of the padded tiles are written
   * as zeros (out, y_save, y2_save, h1_save, rstd*). */
  const int* live16;
  int skip_dead_saves;   /* with live16: 1 = leave the padded tiles' rows of the *_save / rstd* buffers untouched (every
                            consumer is list-driven as well), 0 = write zeros / finite placeholders there */
  int w_packed;          /* 1: Wo, W1, W2 are fragment-packed copies (rg_cast RG_CAST_PACK: contiguous 1 KB operand fragments;
                            under RG_X3: RG_CAST_PACK | RG_CAST_SPLIT, the presplit form) */
  /* Split residual stream (bf16 tier only; both or neither): the layer input and output travel through HBM as a bf16
   * PAIR  value = hi + lo  (hi = bf16(value): the tensor every MFMA-operand consumer reads; lo = bf16(value - hi): read
   * only here, as part of the residual addend) -- ~16 significant bits for the residual stream at 2 x 2 bytes, where one
   * bf16 tensor carries 8 (SURVEY.md 7 "keep the residual stream fp32, feed bf16 only to MFMA operands").  Inside the
   * kernel the LayerNorm outputs that serve as residuals (y, y2) are kept as hi + lo tiles as well.
   * x_lo [M,d] : lo part of x (NULL: x is exact);  out_lo [M,d] : receives the lo part of out (NULL: not produced). */
  const void* x_lo;
  void* out_lo;
} rg_post_attn_args;
int rg_post_attn_fwd(const rg_post_attn_args* args /* host */, int dtype, void* stream);

/* ---- FFN block, backward data path in one launch ---------------------------------------------------
 * Backward of  out = l2(gelu(dropout(l1(y))))  (PositionWiseFeedForwardNet, Transformer/transformer.py:181-188) with
 * respect to its input, after the LayerNorm backward has produced dl2 (gradient at the l2 output, output-dropout mask
 * applied) and dz (gradient through the residual):
 *     dh1 = (dl2 . W2) * gelu'(h1) [* (h1 != 0 ? nz_scale : 0)]        [M, dff]   written for the dW1 product
 *     dy  = dh1 . W1 + dz                                                [M, 128]
 * 64-token tiles, d_ff streamed in 128-wide chunks: dh1 is written once and never read back (the two separate
 * products read 3.75 KB and this kernel 2.75 KB per token at d_ff = 512, bf16).  W2t = W2^T [dff,128] and
 * W1t = W1^T [128,dff] as the [out][in] operands of the two products, row-major or fragment-packed (w_packed).
 * Needs d == 128, dff % 128 == 0.  live16 (rg_live_tiles): only live 16-row tiles are computed, the padded tiles'
 * rows of dy are written as zeros and those of dh1 left untouched. */
typedef struct {
  const void* dl2;
  const void* dz;      /* may alias dl2 (no dropout) */
  const void* h1;      /* [M,dff] saved pre-activation (zeros where dropped) */
  const void* W2t;
  const void* W1t;
  void* dh1;
  void* dy;
  int M, d, dff, w_packed;
  float nz_scale;      /* > 0: dropout on h1 was active, 1/(1-p) */
  const int* live16;
  /* Optional: the LayerNorm backward in front (rg_ln_bwd: nn.LayerNorm of transformer.py:188 from its saved OUTPUT, the
   * `* pad_mask` of :594 / :539 and the output dropout of :186-187) computed inside the kernel from the rows it stages
   * anyway.  ln_dout != NULL: dl2 / dz above are ignored as inputs; dz stays on chip, dl2 (= dz, or dz * dropmask /
   * (1 - p)) is WRITTEN to dl2_out [M,128] for the l2 weight-gradient product, dgamma / dbeta are accumulated (through
   * ln_partials, >= rg_ffn_bwd_ln_workspace() bytes, then a reduce launch).  Rows of padded tiles (live16) of dl2_out are
   * left untouched.  Saves the dz / dl2 round trip through HBM: 3.75 KB -> 3 KB per token for the pair of launches. */
  const void* ln_dout; const void* ln_out; const float* ln_rstd; const float* ln_gamma; const float* ln_beta;
  const float* ln_rowmask;     /* optional [M] */
  void* dl2_out; float* ln_dgamma; float* ln_dbeta; float* ln_partials;
  float ln_drop_p; unsigned long long ln_drop_seed;
} rg_ffn_bwd_args;
size_t rg_ffn_bwd_ln_workspace(int M);

/* ---- attention block, backward of its tail in one launch ----------------------------------------------
 * y = LayerNorm(ctx . Wo^T + bo + x) (MultiHeadAttention tail, Transformer/transformer.py:160-161), backwards with respect
 * to ctx:   dz = LayerNorm backward of dy (rg_ln_bwd's arithmetic, from the saved OUTPUT y; rows with rowmask == 0 get 0),
 *           dctx = dz . Wo.
 * dz [M,128] is written (it is the residual gradient and the Y operand of dWo += dz^T ctx), dctx [M,P=128] is written,
 * dgamma / dbeta are accumulated through ln_partials (>= rg_attn_out_bwd_workspace(M) bytes) and a reduce launch.
 * Wot = Wo^T [P,d] as the [out][in] operand, row-major or fragment-packed (w_packed).  d == P == 128.
 * live16: rows of padded tiles of dz / dctx stay unwritten.  Replaces rg_ln_bwd + rg_gemm_nt (1.25 KB -> 1 KB per token,
 * and the LayerNorm backward no longer runs as a launch of its own at 2.7 TB/s). */
typedef struct {
  const void* dy; const void* y; const float* rstd; const float* gamma; const float* beta;
  const float* rowmask;      /* optional [M] */
  const void* Wot;
  void* dz; void* dctx;
  float* dgamma; float* dbeta; float* ln_partials;
  int M, d, P, w_packed;
  const int* live16;
} rg_attn_out_bwd_args;
int rg_attn_out_bwd(const rg_attn_out_bwd_args* args /* host */, int dtype, void* stream);
size_t rg_attn_out_bwd_workspace(int M);
int rg_ffn_bwd_data(const rg_ffn_bwd_args* args /* host */, int dtype, void* stream);
int rg_ffn_bwd_data_supported(int d, int dff);
/* list [1 + 2*nt + 4], nt = ceil(M/16): list[0] = number of 16-row tiles holding a row with rowmask != 0, list[1..] their
 * indices ascending; the remaining (padded) tiles are listed from the far end backwards (list[nt], list[nt-1], ..);
 * list[1+nt ..] is scratch (64-tile bit masks). */
int rg_live_tiles(const float* rowmask, long long M, int* list, void* stream);
/* first[b] = smallest l with rowmask[b*L + l] != 0, L if the sequence has none (left padding: the padded prefix) */
int rg_first_live(const float* rowmask, int B, int L, int* first, void* stream);

/* ---- single-query attention for the last encoder layer --------------------------------------------
 * Only enc_outputs[:, -1, :] is consumed on the hot path (AutoEnc4Rec_cross.py:122,154;
 * gan_training.py:157-161): row L-1 of ScaledDotProductAttention (Transformer/transformer.py:119-129)
 * with the key-pad replace-fill.  qlast [B,H*32], kv [B,L,2*H*32] (K | V), ctx / dq [B,H*32],
 * dkv [B,L,2*H*32] fully overwritten. */
int rg_attn_lastq_fwd(const void* qlast, const void* kv, const int64_t* key_ids, int64_t pad_value, void* ctx,
                      int B, int L, int H, float scale, float drop_p, unsigned long long seed, int dtype, void* stream,
                      const float* bkv, const int* first_live);
int rg_attn_lastq_bwd(const void* qlast, const void* kv, const void* dctx, const int64_t* key_ids, int64_t pad_value,
                      void* dq, void* dkv, int B, int L, int H, float scale, float drop_p, unsigned long long seed,
                      int dtype, void* stream, const float* bkv, const int* first_live);

/* Single-query attention of the last encoder layer straight from the layer input x (no K / V projection: WK is
 * absorbed into the query, WV into the output -- csrc/attention_lastq_x.hip).  Same results as rg_gemm_nt(x, [WK;WV])
 * + rg_attn_lastq_fwd/bwd up to operand rounding.  bf16 (f32: rg_attn_lastq_xf_* below), d_model = H*32 = 128, L <= 256 (rg_attn_lastq_x_supported).
 * qlast = WQ x[:, L-1] + bQ [B,128]; wk / wv [128,128] bf16 row-major [out,in]; first_live[b] (or NULL) = first row of
 * sequence b that is not a zero row of x (rows before it are not read).
 * Backward outputs: dx [B,L,128] (every row written; the caller adds the query path's dx of row L-1), dq [B,128], and
 * the operands of the two weight-gradient products over T = B*4 rows (row b*4+h):
 *   dWV += ym_v^T xbar,  dWK += ym_q^T dqp      (rg_gemm_tn; ym_* = dctx / qlast with the other heads' blocks zeroed)
 * dbv [128] is ACCUMULATED (dbK is exactly zero: a constant added to every score of a softmax row). */
typedef struct {
  const void* x; const void* qlast; const void* wk; const void* wv;
  const float* bk; const float* bv;
  const int64_t* key_ids; int64_t pad_value;
  const int* first_live;
  void* ctx;                                   /* forward out [B,128] */
  const void* dctx;                            /* backward in [B,128] */
  void* dx; void* dq; void* ym_v; void* ym_q; void* xbar; void* dqp; float* dbv;
  int B, L; float scale; float drop_p; unsigned long long seed;
} rg_lastq_x_args;
int rg_attn_lastq_x_supported(int d, int P, int H, int L, int dtype);
int rg_attn_lastq_x_fwd(const rg_lastq_x_args* args /* host */, void* stream);
int rg_attn_lastq_x_bwd(const rg_lastq_x_args* args /* host */, void* stream);
/* The f32 form of the same two (round 6): every tensor of the argument block -- x, qlast, wk, wv, ctx, dctx, dx, dq, ym_*, xbar, dqp -- is
 * f32 and the arithmetic is exact f32 on the vector ALU (per sequence ~2 k FMAs per lane against 100 KB of rows to fetch: nothing for the
 * matrix pipe to win, no operand split to pay).  The f32 AND bf16x3 tiers use it (rg_attn_lastq_x_supported answers for RG_F32 / RG_X3 too):
 * it replaces their K | V projection + rg_attn_lastq_fwd/bwd + dkv -> dx product + the B L-row dWK | dWV product. */
int rg_attn_lastq_xf_fwd(const rg_lastq_x_args* args /* host */, void* stream);
int rg_attn_lastq_xf_bwd(const rg_lastq_x_args* args /* host */, void* stream);
/* bkv [2*H*32] f32 (K | V bias) + first_live [B] (rg_first_live), both optional (NULL): the caller guarantees that the
 * K / V rows before a sequence's first live position are the bias rows (x_masked contract of rg_attn_args): the kernels
 * then do not fetch them (one score, one probability mass for the whole padded prefix). */

/* ---- decoder cross-attention under dropout ---------------------------------------------------------
 * s[b*L+q, h] = (1/n_b) * sum_{keys j live} keep(seed, ((b*H+h)*L+q)*L+j)   (keep = 0 or 1/(1-p)),
 * n_b = number of live keys (enc_ids != pad; all L keys if none is live -- replace-fill, Q3):
 * the sum of the dropped uniform attention row of MultiHeadAttention(Q, rep(u), rep(u)) (transformer.py:259).
 * rg_seq_wsum: out[b,h,:] = sum_q s[b*L+q,h] * x[b,q,:]  (backward of that stage). */
int rg_cross_drop_scale(const int64_t* enc_ids, int64_t pad_value, float* s, int B, int L, int H, float drop_p,
                        unsigned long long seed, void* stream);
int rg_seq_wsum(const void* x, const float* s, void* out, int B, int L, int H, int N, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
