/* libw2vs - C ABI of the MI355X (gfx950) wav2vec-S pre-training hot path.
 *
 * The reference (biaofuxmu/wav2vec-S) has no FFI for this path: it is Python on ATen
 * (SURVEY.md section 8b).  This header is therefore the boundary a native replacement
 * exports *beneath* the fairseq model API: one entry point per fused kernel, plain
 * pointers + sizes, no torch types.  Each entry cites the reference code it replaces;
 * paths are relative to /root/reference/fairseq/fairseq/ ("fs/").
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says host;
 *  - activations/params are bf16 ("void*"), statistics and gradient accumulators fp32;
 *  - layouts are channel-last / token-major: conv activations [B, L, C], encoder tokens
 *    [B, N, C] (N = T' + R right-context copies), QKV fused [B, N, 3C];
 *  - work is enqueued on `stream` (a hipStream_t passed as void*); no entry synchronises;
 *  - return 0 on success, W2VS_ERR_INVALID (-1) for rejected arguments, W2VS_ERR_HIP (-2)
 *    for a HIP runtime error; w2vs_last_error() returns the message (thread local);
 *  - "accumulate" outputs (float* d...) are ADDED to with fp32 atomics; the caller zeroes.
 */
#ifndef W2VS_H
#define W2VS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2VS_ABI_VERSION 1
#define W2VS_ERR_INVALID (-1)
#define W2VS_ERR_HIP (-2)

int w2vs_abi_version(void);
const char* w2vs_last_error(void);
/* sizeof() of the descriptor structs as compiled, so a binding can verify its mirror:
 * which = 0 gemm, 1 ln_fwd, 2 ln_bwd, 3 enc_prologue, 4 attn, 5 quant, 6 nce, 7 layer, 8 collate */
int w2vs_sizeof(int which);

/* ---- GEMM family ----------------------------------------------------------------------------
 * Replaces every nn.Linear / Conv1d(layers 1-6) matmul on the path:
 *   F.linear in fs/models/wav2vec/wav2vec2.py:567-568 (post_extract_proj), :613 (project_q),
 *   :647 (final_proj); q/k/v/out_proj in fs/modules/multihead_attention.py:161-193; fc1/fc2
 *   in wav2vec2.py:971-973; quantizer weight_proj gumbel_vector_quantizer.py:149;
 *   nn.Conv1d in wav2vec2.py:724-727 (as a strided-A GEMM, see gemm.hip).
 * NT: C[M,N] = epi(A[M,K] * B[N,K]^T).  TN: Cf[M,N] += alpha * A[K,M]^T * B[K,N]. */
enum {
  W2VS_EPI_NONE = 0,           /* C = acc                                            */
  W2VS_EPI_BIAS = 1,           /* C = acc + bias[n]                                  */
  W2VS_EPI_BIAS_GELU = 2,      /* C = gelu(acc + bias[n])          (exact erf GELU)  */
  W2VS_EPI_BIAS_GELU_SAVE = 3, /* C2 = acc + bias (pre-activation), C = gelu(C2)     */
  W2VS_EPI_DGELU = 4,          /* C = acc * gelu'(aux[m,n])                          */
  W2VS_EPI_F32 = 5,            /* Cf = alpha * acc (fp32 output)                     */
  W2VS_EPI_ADD = 6,            /* C = acc + aux[m,n]                                 */
  W2VS_EPI_BIAS_GELU_SAVEG = 7,/* C = gelu(pre), C2 = gelu'(pre), pre = bf16(acc + bias): what the backward needs     */
  W2VS_EPI_MUL = 8             /* C = acc * aux[m,n]              (aux = the saved gelu')                              */
};
typedef struct w2vs_gemm_desc {
  const void* A; const void* B;
  void* C; void* C2; float* Cf;
  const void* bias; const void* aux;
  int32_t M, N, K, batch;
  int64_t lda, ldb, ldc;   /* elements */
  int64_t a_off;           /* element offset of A row 0, may be negative (reads zero) */
  int64_t sA, sB, sC;      /* batch strides, elements */
  int64_t a_bytes, b_bytes;/* valid bytes per batch (0 = derive) : reads beyond return 0 */
  int64_t c_elems;         /* valid output elements per batch (0 = derive)            */
  int32_t epi; float alpha;
  float* colsum;           /* TN only, optional: colsum[m] += alpha * sum_k A[k, m] (bias gradient fused in) */
  void* ws; int64_t ws_bytes; /* TN only, optional scratch (36 MB always suffices): split-K partial tiles are stored
                                 there and summed by a second launch instead of fp32 atomics into Cf */
  /* NT only, optional promise about a STRUCTURAL zero block of B: B[n][k] == 0 for every n >= zk_col and k < zk_k (both
   * multiples of 128; 0 = no promise).  Output tiles that lie wholly at columns >= zk_col then start their K loop at zk_k.
   * The (3,2)-conv input gradient is such a product: pair p = [dY[p-1] | dY[p]] . [[W2, 0], [W0, W1]] - a quarter of its
   * multiply-adds are with that zero block.  Results are identical with and without the promise when it is true. */
  int32_t zk_col, zk_k;
  /* TN only: 1 = Cf = alpha * A^T B instead of Cf += ... (colsum stays an accumulation).  The grouped single-writer launches
   * then skip the read of their output tile, and a trainer need not zero those gradient ranges at the start of a step; the
   * general kernels (atomics / partial-tile sums) clear the target first.  Needs ldc == N, no batch. */
  int32_t overwrite;
} w2vs_gemm_desc;
int w2vs_gemm_nt(const w2vs_gemm_desc* d, void* stream);
/* Tests / tuning: force a kernel variant process-wide (not thread safe).  nt_mode: -1 auto, 0/1/2 the 128x128 forms
 * (register double buffer / single buffer / LDS-DMA), 3 loader-consumer, 5 persistent loader-consumer; lc_height: 0 auto or
 * 256 / 192 / 160; tn_lc: -1 auto, 0 the 128x128 atomics kernel, 1 the loader-consumer kernel.  Returns 0. */
int w2vs_gemm_tune(int32_t nt_mode, int32_t lc_height, int32_t tn_lc);
/* Measurement hooks: bracket every stride-th GEMM launch with HIP events on its own stream (0 = off).
 * id = one kernel symbol family (NT: 16 * form + epilogue; weight gradients 10..13); read returns summed launch time [ms],
 * algorithmic FLOPs and the number of launches timed since the last enable.  An event pair also times any gap in which the
 * stream waited for the host to enqueue the launch: samples whose time per FLOP exceeds 3 x the family's median are left out
 * of the totals (never with fewer than four samples). */
int w2vs_prof_enable(int stride);
int w2vs_prof_read(int id, double* total_ms, double* total_flops, int* launches);
/* the same totals over ALL timed samples (no outlier filter): report both, and the difference in sample counts */
int w2vs_prof_read_raw(int id, double* total_ms, double* total_flops, int* launches);
/* every launch of that kernel id since w2vs_prof_enable (the timed ones are a 1-in-stride sample of them), and the algorithmic
 * FLOPs of all of them (scales a sample's time to its whole family when the family mixes shapes) */
int64_t w2vs_prof_launches(int id);
double w2vs_prof_flops(int id);
int w2vs_gemm_tn(const w2vs_gemm_desc* d, int num_cu_hint, void* stream);
/* n <= 12 weight-gradient GEMMs (e.g. the four of one encoder layer: fused QKV, out_proj, fc1, fc2; more than four only as the
 * 8-phase form: 256 x 256 tiles, all of them in one round of the chip - otherwise groups of four) as ONE launch without
 * a K split: together their 256x128 tiles fill the chip, every tile has a single writer (C += A^T B with plain stores:
 * no partial-tile workspace, no summing launch, no atomics except the optional column sums).  Same result as n calls of
 * w2vs_gemm_tn; falls back to exactly that when the group does not qualify. */
int w2vs_gemm_tn_group(const w2vs_gemm_desc* descs, int32_t n, int32_t num_cu_hint, void* stream);
/* The 8-phase form of the grouped launch may split K over workgroup PAIRS that wait for each other's half tile; it does so
 * only when the whole grid is co-resident (occupancy x CU count of the current device, queried once; num_cu_hint can only
 * lower the count).  max_split = 1 forbids the pairs altogether - the training step sets it while a gradient all-reduce may
 * hold CUs beside the backward (wav2vec-s_amd/trainer.py); 2 (the default) allows them.  Returns 0. */
int w2vs_gemm_tn8_max_split(int32_t max_split);
/* Which form the LAST w2vs_gemm_tn_group call on this thread's process took (tests): 0 one launch per problem, 12 the
 * 256x128 single-writer group, 13 the 8-phase 256x256 group without a K split, 14 with the pairwise split. */
int w2vs_gemm_last_group_form(void);

/* ---- conv layer 0: Conv1d(1->C,k,s) + Fp32LayerNorm(C) + GELU ------------------------------
 * fs/models/wav2vec/wav2vec2.py:733-743, 773-781 (layer 0 of ConvFeatureExtractionModel).
 * wave [B, L] bf16 ; w [C, 1, k] ; y [B, L0, C] channel-last ; mean/rstd [B*L0] fp32.      */
int w2vs_conv0_fwd(const void* wave, const void* w, const void* conv_bias, const void* ln_w, const void* ln_b,
                   void* y, float* mean, float* rstd, int B, int L, int C, int k, int s, void* stream);
int w2vs_conv0_bwd(const void* wave, const void* w, const void* conv_bias, const void* ln_w, const void* ln_b,
                   const float* mean, const float* rstd, const void* dy, float* dw, float* dconv_bias,
                   float* dln_w, float* dln_b, int B, int L, int C, int k, int s, void* stream);

/* conv layer 0 of extractor_mode="default": Conv1d + Fp32GroupNorm(C, C) + GELU (wav2vec2.py:744-750).
 * Statistics run over time per (b, c): stat [B, C, 2] = {sum, sumsq} written by fwd and read by bwd;
 * bstat [B, C, 2] is backward scratch.  Each direction is two passes (stats, apply) over the waveform. */
int w2vs_conv0_gn_fwd(const void* wave, const void* w, const void* conv_bias, const void* gn_w, const void* gn_b,
                      void* y, float* stat, int B, int L, int C, int k, int s, void* stream);
int w2vs_conv0_gn_bwd(const void* wave, const void* w, const void* conv_bias, const void* gn_w, const void* gn_b,
                      const float* stat, const void* dy, float* bstat, float* dw, float* dconv_bias, float* dgn_w,
                      float* dgn_b, int B, int L, int C, int k, int s, void* stream);

/* ---- row LayerNorm family --------------------------------------------------------------------
 * fwd:  s = dropout(x) [+ res] ; sum_out = s ; y = [gelu] LN(s)
 *   self.layer_norm wav2vec2.py:556-557 (+ features_pen :554 via sumsq),
 *   dropout1/3 + residual + self_attn_layer_norm/final_layer_norm wav2vec2.py:945-976,
 *   Fp32LayerNorm+GELU of conv layers i < layer_norm_num (large) wav2vec2.py:733-743.
 * bwd:  ds = LNbwd(dy) [+ dsum] ; dres = ds ; dx = ((ds + 2*x*pen_coef) * out_scale) o dropmask
 *       [o gelu'(aux)] ; dgamma/dbeta accumulate.  x is the LN input (the saved sum).            */
typedef struct w2vs_ln_fwd_desc {
  const void* x; const void* res; const void* gamma; const void* beta;
  void* y; void* sum_out; float* mean; float* rstd; float* sumsq;
  int64_t rows; int32_t C; int32_t gelu; float p_drop; uint64_t seed;
} w2vs_ln_fwd_desc;
typedef struct w2vs_ln_bwd_desc {
  const void* x; const void* gamma; const void* beta; const float* mean; const float* rstd;
  const void* dy; const void* dsum; const void* aux;
  void* dx; void* dres; float* dgamma; float* dbeta;
  int64_t rows; int32_t C; int32_t gelu; float p_drop; uint64_t seed; float out_scale; float pen_coef;
  const float* pen_coef_dev;   /* optional device scalar multiplied into pen_coef (no host sync) */
  void* ws; int64_t ws_bytes;  /* optional scratch (>= 512*2*C*4 B is enough): dgamma/dbeta partials are reduced
                                  through it in a second launch; without it every block adds them atomically */
} w2vs_ln_bwd_desc;
int w2vs_ln_fwd(const w2vs_ln_fwd_desc* d, void* stream);
int w2vs_ln_bwd(const w2vs_ln_bwd_desc* d, void* stream);

/* ---- encoder prologue --------------------------------------------------------------------------
 * dropout_input (wav2vec2.py:570) -> x[mask] = mask_emb (:446) -> zero padded frames, + sinusoid
 * (wav2vec_S.py:357-369) -> LayerNorm (:371-372, post-LN only) -> zero pad frame (:375-384) ->
 * dropout (:385) -> right-context copies (gen_block_attn_mask :465-484), written as [B, N, C].
 * src [N] int32: source frame of every token row; pos [B, T] int32 sinusoid row per frame;
 * copy_start [Tp+1], copy_list [R]: CSR of the copies of each frame (for the backward gather). */
typedef struct w2vs_enc_prologue_desc {
  const void* x; const uint8_t* mask; const uint8_t* pad; const int32_t* pos;
  const void* mask_emb; const float* pos_table; const void* gamma; const void* beta;
  void* out; float* mean; float* rstd; const int32_t* src;
  const void* dout; void* dx; float* dmask_emb; float* dgamma; float* dbeta;
  const int32_t* copy_start; const int32_t* copy_list;
  float p_in; float p_enc; uint64_t seed_in; uint64_t seed_enc;
  int32_t apply_ln, B, T, Tp, N, C;
} w2vs_enc_prologue_desc;
int w2vs_enc_prologue_fwd(const w2vs_enc_prologue_desc* d, void* stream);
int w2vs_enc_prologue_bwd(const w2vs_enc_prologue_desc* d, void* stream);

/* ---- block-causal attention with right-context copies ----------------------------------------
 * gen_block_attn_mask (wav2vec_S.py:444-489) + scaled-dot-product core of
 * F.multi_head_attention_forward (multihead_attention.py:161-193) incl. attention dropout.
 * q/k/v: [B, N, ld] with head h at columns h*64..h*64+63 (ld = 3C for a fused QKV buffer),
 * o: [B, N, ldo]; lse [B, H, N] fp32; kpad [B, N] (1 = padded key) or NULL.
 * bwd: dq/dk/dv in the q/k/v layout; delta is a [B, H, N] fp32 scratch.                        */
typedef struct w2vs_attn_desc {
  const void* q; const void* k; const void* v; void* o; float* lse; const uint8_t* kpad;
  const void* dout; float* delta; void* dq; void* dk; void* dv;
  int64_t ld, ldo, sb, sbo;
  int32_t B, H, N, Tp, m, r, head_dim;
  float scale; float p_drop; uint64_t seed;
  int32_t Nq;   /* 0 = N.  Otherwise only positions 0..Nq-1 are queries (Nq <= Tp: the main frames); o / lse / dq rows
                 * past Nq are left untouched - used by the last encoder layer, whose right-context outputs are dead */
  /* Cross ("group prefix") mode, mq > 0: ExpandMultiheadAttention of the CAAT joiner
   * (rain/layers/attention_transducer.py:642-715 with the group mask of MHAJointNet._gen_group_mask, :810-824).
   * The Nq queries live in their own buffer q [B, Nq, ldq] (batch stride sbq), rows ordered (group g, target u) with
   * mq = U targets per group; keys / values k, v [B, N, ld] are the encoder frames (r must be 0, Tp == N).  Query row
   * qi belongs to group qi / mq and attends the keys < min((qi / mq + 1) * m, N) that kpad does not mark (m = the
   * joiner's downsample).  o / dout [B, Nq, ldo], lse / delta [B, H, Nq]; dq in the q layout, dk / dv in the k / v layout. */
  int32_t mq; int64_t ldq, sbq;
  /* Optional keep-mask store for the attention dropout (32-row kernels only): w2vs_attn_drop_bits_bytes(B, H, N, Nq) bytes.
   * With p_drop > 0 the forward records its keep decisions there (one bit per visible (query, key) pair, 128 bytes per
   * 32 x 32 block) and w2vs_attn_bwd, given the same buffer, reads them instead of re-evaluating the counter hash per
   * score - the same decisions either way; NULL on both calls = recompute.  Scratch owned by the caller between the calls. */
  void* drop_bits;
} w2vs_attn_desc;
int w2vs_attn_fwd(const w2vs_attn_desc* d, void* stream);
int w2vs_attn_bwd(const w2vs_attn_desc* d, void* stream);
/* tests / A-B runs: 1 = the 128-query-workgroup kernels (csrc/attention.hip), 2 = the 32-row split kernels (csrc/attention2.hip,
 * default), -1 = default again */
int w2vs_attn_tune(int32_t variant);
/* size of w2vs_attn_desc.drop_bits for a call with these dimensions (Nq = 0 means N) */
int64_t w2vs_attn_drop_bits_bytes(int32_t B, int32_t H, int32_t N, int32_t Nq);

/* ---- composite: one post-LN Transformer encoder layer -------------------------------------------
 * TransformerSentenceEncoderLayer.forward (fs/models/wav2vec/wav2vec2.py:932-976: post-LN :955-976, pre-LN :932-953) with the fused QKV
 * projection, block attention, out_proj, dropout+residual+LayerNorm, fc1+GELU, fc2,
 * dropout+residual+LayerNorm enqueued by ONE call (and the whole backward by one more), so the host
 * issues 2 calls per layer.  R = B*N rows.  Saved activations are written by fwd and read by bwd.
 * hpre = NULL: inference form of the forward (gelu'(fc1 pre-activation) is not stored; layer_bwd then refuses the descriptor).
 * wqkv [3E,E] / bqkv [3E] = q,k,v projections stacked.  Scratch (bwd): ws_e0..2 [R,E], ws_f [R,F],
 * ws_qkv [R,3E], delta [B,H,N] fp32, wt_scratch >= max(3E*E, E*F) bf16.  Gradients accumulate.      */
typedef struct w2vs_layer_desc {
  int32_t B, N, E, F, H, Tp, m, r, post_ln, num_cu;
  float p_drop, p_attn; uint64_t seed_attn, seed_drop1, seed_drop2;
  const uint8_t* kpad;
  const void *wqkv, *bqkv, *wo, *bo, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
  const void* x_in;
  void *qkv, *ctx; float* lse;
  void* s1; float *mean1, *rstd1; void* x1;
  void *hpre, *h;
  void* s2; float *mean2, *rstd2; void* x_out;
  void* tmp;
  const void* d_out; void* d_in;
  float *g_wqkv, *g_bqkv, *g_wo, *g_bo, *g_ln1_g, *g_ln1_b, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln2_g, *g_ln2_b;
  void *ws_e0, *ws_e1, *ws_e2, *ws_f, *ws_qkv, *wt_scratch; float* delta;
  /* optional: the four weights already transposed ([E,3E], [E,E], [E,F], [F,E]), e.g. by ONE w2vs_transpose_multi
   * for every layer of the step; layer_bwd then skips its own four transposes (wt_scratch may be NULL) */
  const void *wqkv_t, *wo_t, *w1_t, *w2_t;
  void* tn_ws; int64_t tn_ws_bytes;   /* optional scratch handed to the weight-gradient GEMMs (w2vs_gemm_desc.ws) */
  /* optional "selected rows" mode for the LAST layer of a pre-training step: only the n_sel token rows sel_idx[]
   * (= the masked frames) of its output are ever read, so everything after the attention runs on those rows only
   * (s1, x1, hpre, h, s2, x_out, mean/rstd then hold n_sel rows, in sel_idx order) and the attention takes the
   * n_q = Tp main frames as queries.  ctx_sel / xin_sel: [n_sel, E] scratch kept for the backward. */
  const int32_t* sel_idx; int32_t n_sel, n_q; void* ctx_sel; void* xin_sel;
  /* optional fourth [R,E] backward scratch: with it (and without sel_idx) layer_bwd keeps every weight-gradient operand
   * alive to the end of the layer and computes the four weight gradients as ONE w2vs_gemm_tn_group launch */
  void* ws_e3;
  /* pre-LN form (post_ln = 0, wav2vec-S large: layer_norm_first; wav2vec2.py:932-953): x_in = LN(stream_in) as produced by the
   * previous layer (x_out of this call is the NEXT layer's x_in), ln1_* = this layer's final_layer_norm, ln2_* = the next norm
   * (next layer's self_attn_layer_norm, or encoder.layer_norm behind the last layer); s1 / s2 are the stream after the
   * attention / FFN residual.  bwd: d_out = dL/d x_out, d_stream_out = dL/d s2 (NULL behind the last layer),
   * d_in = dL/d x_in, d_stream_in = dL/d stream_in. */
  const void* stream_in; const void* d_stream_out; void* d_stream_in;
  void* drop_bits;   /* optional: w2vs_attn_desc.drop_bits of this layer's attention (written by fwd, read by bwd) */
  /* 1 (needs ws_e3, no sel_idx): layer_bwd does NOT launch the four weight gradients; their operands stay in ws_f (d fc1-out),
   * ws_e0 (d fc2-out), ws_qkv (d qkv), ws_e3 (d out_proj-out) until the caller passes this descriptor to w2vs_layer_wgrads -
   * the next layer's backward must therefore run on OTHER ws_f / ws_e0 / ws_qkv / ws_e3 buffers */
  int32_t defer_wgrads;
  /* 1: the deferred weight gradients of this layer OVERWRITE g_wqkv / g_wo / g_w1 / g_w2 (w2vs_gemm_desc.overwrite; the bias
   * gradients still accumulate): the first micro-batch of an update, whose caller then does not zero those ranges */
  int32_t wgrad_overwrite;
  /* optional, selected-rows mode only: two more [R,E] backward scratch buffers.  With them (and ws_e3, defer_wgrads = 1) the
   * scatter back to token rows no longer reuses ws_e0 / ws_f, the four weight-gradient operands of the LAST layer stay alive
   * too, and w2vs_layer_wgrads takes the layer (three of its four GEMMs contract over the n_sel selected rows) */
  void *ws_s0, *ws_s1;
  /* optional, with defer_wgrads = 1: a slab of ln_part_bytes (two halves of up to 768 x 2E fp32 each, i.e. 2 x 4.7 MB at E = 768)
   * that this layer OWNS until its w2vs_layer_wgrads call: the two LayerNorm backward kernels leave their per-block
   * dgamma / dbeta partial rows there and w2vs_layer_wgrads sums the partials of all its layers' norms in one launch
   * (g_ln1_* / g_ln2_* are final only after that call, like the weight gradients).  NULL: each norm reduces at once. */
  void* ln_part; int64_t ln_part_bytes;
} w2vs_layer_desc;
int w2vs_layer_fwd(const w2vs_layer_desc* d, void* stream);
int w2vs_layer_bwd(const w2vs_layer_desc* d, void* stream);
/* The weight gradients (+ bias gradients) of n = 1 or 2 layers whose layer_bwd ran with defer_wgrads = 1, as ONE grouped
 * launch.  Two base-model layers are 216 output tiles of 256 x 256 with full-length K loops: no K split, no exchange between
 * workgroups - a single layer's 108 tiles need the split to fill the chip.  Falls back to one launch per layer when the pair
 * does not fit one workgroup per CU. */
int w2vs_layer_wgrads(const w2vs_layer_desc* layers, int32_t n, void* stream);
/* The same for a caller that packs its launches itself (round 4): parts[i] selects which of layers[i]'s gradients go into THIS
 * launch - bit 0 fc1, bit 1 fc2, bit 2 fused QKV, bit 3 out_proj (weight + bias each), bit 4 the layer's LayerNorm partial sums
 * (ln_part) - n <= 6 layers, <= 12 GEMMs and <= 8 norms per call.  One base-model layer is 36 + 36 + 27 + 9 output tiles of
 * 256 x 256 and a launch is ONE round of the chip whatever its tile count (<= the CU count), so seven 36-tile GEMMs (or
 * 6 x 36 + 27 + 9) = 252 tiles use 252 of 256 CUs where a layer pair uses 216.  Every selected gradient must come from a layer
 * whose layer_bwd ran with defer_wgrads = 1 and whose operand buffers (ws_f, ws_e0, ws_qkv, ws_e3, ln_part) are still intact.
 * Groups that do not fill 5/8 of the chip run as the smaller forms (K split through a slab), as in w2vs_gemm_tn_group. */
int w2vs_layer_wgrads_parts(const w2vs_layer_desc* layers, const int32_t* parts, int32_t n, void* stream);

/* ---- Gumbel vector quantizer ---------------------------------------------------------------------
 * fs/modules/gumbel_vector_quantizer.py:141-202 after the weight_proj GEMM: hard argmax +
 * code/prob perplexities (:152-169), gumbel-softmax hard sample (:173-176), codebook product
 * (:192-195) done as a 2-row gather instead of a (B*M) x 640 x 128 broadcast product.
 * logits [R, G*V] bf16 -- or, preferred, logits_f32 [R, G*V] fp32 (the weight_proj product through
 * W2VS_EPI_F32, no rounding of the sums) plus logit_bias [G*V] bf16 added here: the code SELECTION is an argmax, and a
 * bf16-rounded logit (ulp 0.25 at |x| ~ 60) flips near-ties that fp32 keeps;
 * noise [R*G, V] fp32 Gumbel samples or NULL (device RNG from seed);
 * vars [G*V, D]; q [R, G*D]; idx [R, G]; hard_cnt/prob_sum [G*V] fp32 scratch kept for bwd;
 * ppl_out[2] = {prob_perplexity, code_perplexity}; cvec_out [G*V] = d prob_ppl / d avg_prob.
 * bwd: dsoft [R, G*V] = dq . vars^T (one batched w2vs_gemm_nt per group);
 *      ppl_grad = dLoss/d prob_perplexity.                                                        */
typedef struct w2vs_quant_desc {
  const void* logits; const float* noise; const void* vars;
  void* q; int32_t* idx; float* hard_cnt; float* prob_sum; float* ppl_out; float* cvec_out;
  const void* dq; const void* dsoft; const float* cvec; void* dlogits; float* dvars;
  float ppl_grad; float tau; int32_t R, G, V, D, training; uint64_t seed;
  const float* ppl_grad_dev;   /* optional device scalar multiplied into ppl_grad (no host sync) */
  const float* logits_f32;     /* optional: fp32 logits WITHOUT bias (then `logits` may be NULL) */
  const void* logit_bias;      /* bf16 [G*V] added to logits_f32 (NULL = none) */
  const float* dsoft_f32;      /* bwd, optional: dsoft as fp32 (W2VS_EPI_F32 product); then `dsoft` may be NULL */
} w2vs_quant_desc;
int w2vs_quant_fwd(const w2vs_quant_desc* d, void* stream);
int w2vs_quant_bwd(const w2vs_quant_desc* d, void* stream);

/* ---- InfoNCE logits -----------------------------------------------------------------------------
 * sample_negatives' gather (wav2vec2.py:521-526) + compute_preds (:529-542) fused: the
 * 100 x B x M x C negatives tensor is never materialised.  x, y [B*M, C] bf16, neg_idx [B, K*M]
 * int64 (the reference's index tensor, rows of y.view(-1, C)), logits [B*M, K+1] fp32 with
 * column 0 the positive; neg==pos entries are -inf.  bwd reads the saved logits and norms; dy is
 * accumulated per utterance in LDS (negatives never cross utterances) - no global atomics.    */
typedef struct w2vs_nce_desc {
  const void* x; const void* y; const int64_t* neg_idx; float* logits;
  float* xn; float* yn;              /* [B*M] fp32 row norms, written by fwd, read by bwd */
  const float* dlogits; void* dx; void* dy;   /* bwd: dx, dy bf16 [B*M, C] */
  float* dy_ws;                      /* optional fp32 [B*M, C] scratch: lets bwd split rows over more blocks */
  int32_t B, M, K, C; float temp;
  void* ws; int64_t ws_bytes;        /* optional scratch, >= 4*B*Mp*(Mp + 2*C + 2) bytes with Mp = M rounded up to 64: bwd then
                                        runs as two small matrix products per utterance instead of an LDS-atomic scatter */
} w2vs_nce_desc;
int w2vs_nce_fwd(const w2vs_nce_desc* d, void* stream);
int w2vs_nce_bwd(const w2vs_nce_desc* d, void* stream);

/* ---- cross entropy (target 0, reduction=sum) + accuracy counters --------------------------------
 * fs/criterions/wav2vec_criterion.py:68, 133-155.  out3 = {loss, #argmax==0, #(argmax==0 &&
 * argmin==0)}; dlogits (optional) = softmax - onehot(0).                                          */
int w2vs_ce_rows(const float* logits, int64_t R, int32_t W, float* out3, float* dlogits, void* stream);

/* ---- the whole InfoNCE criterion in one launch -------------------------------------------------------
 * fs/criterions/wav2vec_criterion.py:64-100: cross entropy as above, then (by the block that finishes last)
 *   loss = ce + w_ppl * ((num_vars - prob_ppl) / num_vars) * sample_size + w_pen * features_pen * sample_size
 * with features_pen = pen_acc[0] * pen_norm (wav2vec2.py:571) and the extra losses in the order of
 * wav2vec2.py:669-679 (get_extra_losses).  loss[1] and vec[8] = {loss, ce, ppl term, pen term, correct
 * (:141-151: #max0 - #both0), prob_perplexity, code_perplexity, features_pen}.  scratch: 4 words, zero on entry,
 * zero again on exit (one buffer serves every launch of a stream).  dlogits optional, as for ce_rows.      */
typedef struct w2vs_infonce_loss_desc {
  const float* logits; int64_t R; int32_t W;
  const float* pen_acc; const float* ppl;
  float w_ppl, w_pen, num_vars, pen_norm, sample_size;
  float* loss; float* vec; float* dlogits; float* scratch;
} w2vs_infonce_loss_desc;
int w2vs_infonce_loss(const w2vs_infonce_loss_desc* d, void* stream);
/* its backward: dlogits[n] *= g[0] in place, dsc = {g * c_pen, g * c_ppl} - the gradients of features_pen and
 * prob_perplexity for c_pen = w_pen * sample_size, c_ppl = -w_ppl * sample_size / num_vars               */
int w2vs_infonce_loss_bwd(const float* g, float* dlogits, int64_t n, float c_pen, float c_ppl, float* dsc, void* stream);

/* ---- small movers ------------------------------------------------------------------------------- */
/* dst[i] = src[idx[i]] (scatter=0) or dst[idx[i]] = src[i] (scatter=1); rows of C bf16.
 * x[mask_indices] / unmasked_features[mask_indices] (wav2vec2.py:590-592, 641) and their grads.  */
int w2vs_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t R, int32_t C, int32_t scatter, void* stream);
int w2vs_transpose2d(const void* in, void* out, int32_t R, int32_t C, int32_t batch, void* stream);
/* many independent 2-D transposes in one launch (n <= 64): out[i] [C_i, R_i] = in[i] [R_i, C_i]^T */
typedef struct w2vs_transpose_item { const void* in; void* out; int32_t R, C; int64_t ld_in, ld_out; /* row strides in elements, 0 = dense (C / R) */ } w2vs_transpose_item;
int w2vs_transpose_multi(const w2vs_transpose_item* items, int32_t n, void* stream);
int w2vs_f32_to_bf16(const float* in, void* out, int64_t n, float scale, void* stream);
/* out[i] = float(in[i]) (bf16 -> fp32): with w2vs_f32_to_bf16 the pack / unpack pair of a bf16-compressed gradient exchange
 * (the reference all-reduces gradients in the model dtype, fs/distributed/legacy_distributed_data_parallel.py:100-115) */
int w2vs_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);
/* out[i] = in[i] * keep(seed, i) / (1 - p) : nn.Dropout (dropout_features, wav2vec2.py:571);
 * the backward is the same call on the gradient with the same seed. */
int w2vs_dropout(const void* in, void* out, int64_t n, float p, uint64_t seed, void* stream);
/* out[i] = gate[i] > 0 ? x[i] : 0 (bf16): the ReLU of the CAAT joiner's FFN (rain/layers/attention_transducer.py:772) with
 * gate == x, and its backward with x = d(out), gate = the forward's output. */
int w2vs_relu_gate(const void* x, const void* gate, void* out, int64_t n, void* stream);
/* Fused Adam on flat arrays: fairseq Adam (fs/optim/adam.py:205-229: decoupled weight decay,
 * bias-corrected step) + fp32 master / bf16 working copy (fs/optim/fp16_optimizer.py:205-218).
 * g = fp32 gradient arena; effective gradient = g * scale_host * (scale_dev ? *scale_dev : 1).
 * An effective scale of exactly 0 (what w2vs_clip_scale emits for a non-finite gradient norm) SKIPS the update:
 * master, moments and bf16 image are left untouched, no weight decay (fs/trainer.py:791-793 raises before
 * optimizer.step in that case).  */
int w2vs_adam_step(float* p32, void* p16, float* m, float* v, const float* g, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int32_t step, const float* scale_dev, float scale_host,
                   void* stream);
/* out[0] += sum x^2 : gradient norm (fs/utils.py:341-386) */
int w2vs_sumsq(const float* x, int64_t n, float* out, void* stream);
/* clip_grad_norm_ (fs/utils.py:341-386) applied after multiply_grads(1/sample_size) (fs/trainer.py:769-774), with the
 * norm, the comparison and the factor kept on the device (no host read): inv = scale_host * (scale_dev ? *scale_dev : 1);
 * gnorm = sqrt(*sumsq) * inv; out3 = {inv * min(1, clip / (gnorm + 1e-6)) [inv when clip <= 0], gnorm, non-finite flag}.
 * out3[0] is what w2vs_adam_step takes as scale_dev; a non-finite norm gives scale 0 and flag 1
 * (the reference raises FloatingPointError, fs/trainer.py:791-793; the caller reads the flag when it chooses to). */
int w2vs_clip_scale(const float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, void* stream);
/* The same, for a training loop that calls it once per update: additionally resets *sumsq to 0 (the next w2vs_sumsq needs
 * no separate clear) and, when bad_acc is given, adds the non-finite flag to bad_acc[0] (a sticky count of skipped updates). */
int w2vs_clip_scale_acc(float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, float* bad_acc,
                        void* stream);
/* out[n] += sum_m in[m, n] : bias gradients */
int w2vs_colsum(const void* in, float* out, int64_t M, int32_t N, int64_t ld, void* stream);

/* ---- input side --------------------------------------------------
 * w2vs_batch_by_size: HOST function, no GPU work.  fs/data/data_utils_fast.pyx:19-98 batch_by_size_vec
 * (the Cython batcher behind fs/data/data_utils.py:281-355 and FairseqDataset.batch_by_size,
 * fs/data/fairseq_dataset.py:104-153).  num_tokens[n] (host) = the sizes of the ordered indices; batch_ends[n]
 * (host, out) receives the exclusive end position of every batch, *n_batches their count; the batches are
 * indices[0:e0], indices[e0:e1], ...  max_tokens / max_sentences <= 0 = unlimited; bsz_mult >= 1.
 * Returns W2VS_ERR_INVALID when a single sample exceeds max_tokens (the reference asserts, :30-32).   */
int w2vs_batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences,
                       int32_t bsz_mult, int32_t* batch_ends, int32_t* n_batches);

/* w2vs_collate: RawAudioDataset.collater (fs/data/audio/raw_audio_dataset.py:123-192) for one batch, with
 * postprocess's per-utterance normalisation (:60-72: F.layer_norm(feats, feats.shape), eps 1e-5, over the WHOLE
 * utterance, before any crop) folded in.  flat = the batch's utterances back to back as read from disk (fp32);
 * row b of out is utterance b cropped to [crop_start[b], crop_start[b] + target) when it is longer than target
 * (crop_to_max_size :73-81; the start is the caller's np.random.randint draw), copied when equal, and zero-padded
 * with padding_mask = 1 when shorter (pad=True datasets).  Rows are `width` >= target long; columns past target are
 * padding too (bucketed batch shapes, :157-167).  out: bf16 [B, width] (out_f32 = 0) or fp32 (1); padding_mask: uint8
 * [B, width] or NULL; partial: scratch, >= B * w2vs_collate_chunks(max_size) * 2 doubles (normalize only).        */
typedef struct w2vs_collate_desc {
  const float* flat; const int64_t* offset; const int32_t* size; const int32_t* crop_start;
  void* out; uint8_t* padding_mask; double* partial;
  int32_t B, target, width, max_size, normalize, out_f32;
} w2vs_collate_desc;
int32_t w2vs_collate_chunks(int32_t max_size);
int w2vs_collate(const w2vs_collate_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* W2VS_H */
