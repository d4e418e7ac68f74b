// Host-side composite entry points: one C call enqueues every kernel of a Transformer encoder
// layer (forward or backward) on the caller's stream.  The Python driver then issues ~2 calls per
// layer instead of ~45, which removes the host launch path from the critical path (the GPU work of
// a layer is ~1 ms; 45 ctypes round trips were costing more than that).
//
// Post-LN layer (wav2vec-S base), fs/models/wav2vec/wav2vec2.py:955-976:
//     a  = out_proj(attn(qkv(x)))            s1 = x + drop(a)      x1 = LN1(s1)
//     f  = fc2(gelu(fc1(x1)))                s2 = x1 + drop(f)     y  = LN2(s2)
// (the buffer called hpre holds gelu'(fc1 pre-activation), written by the forward's own erf evaluation)
//
// Pre-LN layer (wav2vec-S large, `layer_norm_first`), wav2vec2.py:932-953, on a residual STREAM s whose normalised image
// n = LN(s) the previous layer (or a standalone LayerNorm for layer 0) has already produced - every "residual add + next
// LayerNorm" pair is one fused kernel:
//     a  = out_proj(attn(qkv(n_in)))         s1 = s_in + drop(a)   x1 = LN_final(s1)            [ln1_* = final_layer_norm]
//     f  = fc2(gelu(fc1(x1)))                s2 = s1 + drop(f)     y  = LN_next(s2)             [ln2_* = the NEXT norm:
//                                                                     next layer's self_attn_layer_norm / encoder.layer_norm]
// x_in = n_in, stream_in = s_in; outputs s2 (the stream) and x_out = y (the next layer's n_in).  Backward takes d_out = dL/dy
// and d_stream_out = dL/ds2 (NULL for the last layer) and returns d_in = dL/dn_in and d_stream_in = dL/ds_in.
#include "w2vs_internal.h"

namespace w2vs {

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_GELU_SAVE = 3, EPI_DGELU = 4, EPI_F32 = 5, EPI_ADD = 6,
       EPI_BIAS_GELU_SAVEG = 7, EPI_MUL = 8 };

static int lin_fwd(const void* x, const void* w, const void* b, void* y, void* pre, int R, int N, int K, int epi, hipStream_t s) {
  GemmDesc d{};
  d.A = x; d.B = w; d.C = y; d.C2 = pre; d.bias = b;
  d.M = R; d.N = N; d.K = K; d.batch = 1; d.lda = K; d.ldb = K; d.ldc = N; d.epi = epi; d.alpha = 1.f;
  return gemm_nt(d, s);
}
// dx[R,K] = dy[R,N] @ W[N,K], W given transposed (wt = [K,N]); aux for DGELU / ADD epilogues
static int lin_dgrad(const void* dy, const void* wt, void* dx, const void* aux, int R, int N, int K, int epi, hipStream_t s) {
  GemmDesc d{};
  d.A = dy; d.B = wt; d.C = dx; d.aux = aux;
  d.M = R; d.N = K; d.K = N; d.batch = 1; d.lda = N; d.ldb = N; d.ldc = K; d.epi = epi; d.alpha = 1.f;
  return gemm_nt(d, s);
}
// dw[N,K] += dy[R,N]^T x[R,K] ; db[N] += colsum(dy)
static int lin_wgrad(const void* dy, const void* x, float* dw, float* db, int R, int N, int K, int num_cu, hipStream_t s,
                     void* ws, int64_t ws_bytes) {
  GemmDesc d{};
  d.ws = ws; d.ws_bytes = ws_bytes;
  d.A = dy; d.B = x; d.Cf = dw;
  d.M = N; d.N = K; d.K = R; d.batch = 1; d.lda = N; d.ldb = K; d.ldc = K; d.alpha = 1.f;
  d.colsum = db;  // bias gradient rides along in the wgrad GEMM's A-tile staging
  return gemm_tn(d, num_cu, s);
}

static GemmDesc wgrad_desc(const void* dy, const void* x, float* dw, float* db, int R, int N, int K, void* ws, int64_t ws_bytes) {
  GemmDesc d{};
  d.ws = ws; d.ws_bytes = ws_bytes;
  d.A = dy; d.B = x; d.Cf = dw;
  d.M = N; d.N = K; d.K = R; d.batch = 1; d.lda = N; d.ldb = K; d.ldc = K; d.alpha = 1.f;
  d.colsum = db;
  return d;
}

#define TRY(x) do { if (int e_ = (x)) return e_; } while (0)

// the four weight gradients of a layer whose backward kept their operands alive (ws_e3 given).  Selected-rows mode (round 4):
// everything behind the attention ran on the n_sel selected rows, so three of the four contract over n_sel rows (the attention
// context operand is ctx_sel) and only the fused QKV projection over all R token rows - the grouped kernels take a K per problem
static void layer_wgrad_descs(const w2vs_layer_desc& L, GemmDesc* g) {
  const int R = L.B * L.N, E = L.E, F = L.F;
  const bool sel = L.sel_idx != nullptr;
  const int Rt = sel ? L.n_sel : R;
  g[0] = wgrad_desc(L.ws_f, L.x1, L.g_w1, L.g_b1, Rt, F, E, L.tn_ws, L.tn_ws_bytes);            // fc1   [F,E]
  g[1] = wgrad_desc(L.ws_e0, L.h, L.g_w2, L.g_b2, Rt, E, F, L.tn_ws, L.tn_ws_bytes);            // fc2   [E,F]
  g[2] = wgrad_desc(L.ws_qkv, L.x_in, L.g_wqkv, L.g_bqkv, R, 3 * E, E, L.tn_ws, L.tn_ws_bytes); // qkv   [3E,E]
  g[3] = wgrad_desc(L.ws_e3, sel ? L.ctx_sel : L.ctx, L.g_wo, L.g_bo, Rt, E, E, L.tn_ws, L.tn_ws_bytes);   // out_proj [E,E]
  for (int i = 0; i < 4; ++i) g[i].overwrite = L.wgrad_overwrite ? 1 : 0;
}

// the dgamma / dbeta partial slabs of a deferred layer: LN2's in the first half of ln_part, LN1's in the second
static bool ln_deferred(const w2vs_layer_desc& L) {
  return L.defer_wgrads && L.ln_part && (L.ln_part_bytes / 2 & ~(int64_t)255) >= ln_min_slab_bytes(L.E);
}
static int64_t ln_half_bytes(const w2vs_layer_desc& L) { return L.ln_part_bytes / 2 & ~(int64_t)255; }

static int layer_check(const w2vs_layer_desc& L) {
  if (L.B <= 0 || L.N <= 0 || L.E <= 0 || L.F <= 0 || L.H <= 0) return set_error("layer: bad dims");
  if (L.E % 8 || L.F % 8 || L.E / L.H != 64) return set_error("layer: need E%8==0, F%8==0, head_dim 64");
  if (!L.post_ln && !L.stream_in) return set_error("layer: the pre-LN form needs stream_in (the residual stream)");
  if (!L.x_in || !L.wqkv || !L.bqkv || !L.wo || !L.bo || !L.w1 || !L.b1 || !L.w2 || !L.b2 || !L.ln1_g || !L.ln1_b ||
      !L.ln2_g || !L.ln2_b)
    return set_error("layer: null weight/input pointer");
  if (!L.qkv || !L.ctx || !L.lse || !L.s1 || !L.mean1 || !L.rstd1 || !L.x1 || !L.h || !L.s2 || !L.mean2 ||
      !L.rstd2 || !L.x_out || !L.tmp)
    return set_error("layer: null activation pointer");
  return 0;
}

static void fill_attn(const w2vs_layer_desc& L, AttnDesc& a) {
  const long E = L.E;
  a.q = L.qkv; a.k = (const char*)L.qkv + 2 * E; a.v = (const char*)L.qkv + 4 * E;
  a.o = L.ctx; a.lse = L.lse; a.kpad = L.kpad;
  a.ld = 3 * E; a.ldo = E; a.sb = (long)L.N * 3 * E; a.sbo = (long)L.N * E;
  a.B = L.B; a.H = L.H; a.N = L.N; a.Tp = L.Tp; a.m = L.m; a.r = L.r; a.head_dim = 64;
  a.scale = 0.125f; a.p_drop = L.p_attn; a.seed = L.seed_attn; a.drop_bits = L.drop_bits;
}

static int sel_check(const w2vs_layer_desc& L) {
  if (!L.sel_idx) return 0;
  if (L.n_sel <= 0 || L.n_sel > L.B * L.N || !L.ctx_sel || !L.xin_sel) return set_error("layer: selected-rows mode needs n_sel, ctx_sel, xin_sel");
  if (L.n_q <= 0 || (L.n_q != L.N && L.n_q > L.Tp)) return set_error("layer: n_q must be N or <= Tp");
  return 0;
}

int layer_fwd(const w2vs_layer_desc& L, hipStream_t s) {
  TRY(layer_check(L));
  TRY(sel_check(L));
  const int R = L.B * L.N, E = L.E, F = L.F;
  const bool sel = L.sel_idx != nullptr;
  const int Rt = sel ? L.n_sel : R;      // rows of everything behind the attention
  TRY(lin_fwd(L.x_in, L.wqkv, L.bqkv, L.qkv, nullptr, R, 3 * E, E, EPI_BIAS, s));
  AttnDesc a{};
  fill_attn(L, a);
  if (sel) a.Nq = L.n_q;
  TRY(attn_fwd(a, s));
  const void* ctx = L.ctx;
  const void* xin = L.post_ln ? L.x_in : L.stream_in;      // what the attention branch is added to
  if (sel) {   // only these token rows of the layer output are read downstream (the masked frames)
    TRY(gather_rows(L.ctx, L.sel_idx, L.ctx_sel, Rt, E, 0, s));
    TRY(gather_rows(xin, L.sel_idx, L.xin_sel, Rt, E, 0, s));
    ctx = L.ctx_sel; xin = L.xin_sel;
  }
  TRY(lin_fwd(ctx, L.wo, L.bo, L.tmp, nullptr, Rt, E, E, EPI_BIAS, s));
  LnFwdDesc n1{};
  n1.x = L.tmp; n1.res = xin; n1.gamma = L.ln1_g; n1.beta = L.ln1_b; n1.y = L.x1; n1.sum_out = L.s1;
  n1.mean = L.mean1; n1.rstd = L.rstd1; n1.rows = Rt; n1.C = E; n1.p_drop = L.p_drop; n1.seed = L.seed_drop1;
  TRY(ln_fwd(n1, s));
  // hpre <- gelu'(pre) for the backward; an inference call passes hpre = NULL and skips that store (half the epilogue's bytes)
  TRY(lin_fwd(L.x1, L.w1, L.b1, L.h, L.hpre, Rt, F, E, L.hpre ? EPI_BIAS_GELU_SAVEG : EPI_BIAS_GELU, s));
  TRY(lin_fwd(L.h, L.w2, L.b2, L.tmp, nullptr, Rt, E, F, EPI_BIAS, s));
  LnFwdDesc n2{};
  n2.x = L.tmp; n2.res = L.post_ln ? L.x1 : L.s1; n2.gamma = L.ln2_g; n2.beta = L.ln2_b; n2.y = L.x_out; n2.sum_out = L.s2;
  n2.mean = L.mean2; n2.rstd = L.rstd2; n2.rows = Rt; n2.C = E; n2.p_drop = L.p_drop; n2.seed = L.seed_drop2;
  TRY(ln_fwd(n2, s));
  return 0;
}

int layer_bwd(const w2vs_layer_desc& L, hipStream_t s) {
  TRY(layer_check(L));
  if (!L.hpre) return set_error("layer_bwd: the forward ran without hpre (inference form): nothing to differentiate");
  TRY(sel_check(L));
  const bool pre_t = L.wqkv_t && L.wo_t && L.w1_t && L.w2_t;
  if (!L.post_ln && !L.d_stream_in) return set_error("layer_bwd: the pre-LN form needs d_stream_in");
  if (!L.d_out || !L.d_in || (!L.wt_scratch && !pre_t) || !L.ws_e0 || !L.ws_e1 || !L.ws_e2 || !L.ws_f || !L.ws_qkv || !L.delta)
    return set_error("layer_bwd: null scratch pointer");
  if (!L.g_wqkv || !L.g_bqkv || !L.g_wo || !L.g_bo || !L.g_w1 || !L.g_b1 || !L.g_w2 || !L.g_b2 || !L.g_ln1_g ||
      !L.g_ln1_b || !L.g_ln2_g || !L.g_ln2_b)
    return set_error("layer_bwd: null gradient pointer");
  const int R = L.B * L.N, E = L.E, F = L.F, cu = L.num_cu > 0 ? L.num_cu : 256;
  const bool sel = L.sel_idx != nullptr;
  const int Rt = sel ? L.n_sel : R;      // d_out, and everything down to d_ctx, has Rt rows
  const void* ctx = sel ? L.ctx_sel : L.ctx;
  // LN2 backward: d_f = ds2 o dropmask (ws_e0), d_x1a = ds2 (ws_e1)
  LnBwdDesc b2{};
  b2.x = L.s2; b2.gamma = L.ln2_g; b2.beta = L.ln2_b; b2.mean = L.mean2; b2.rstd = L.rstd2; b2.dy = L.d_out;
  if (!L.post_ln) b2.dsum = L.d_stream_out;            // pre-LN: the stream's own gradient joins here (NULL: last layer)
  b2.dx = L.ws_e0; b2.dres = L.ws_e1; b2.dgamma = L.g_ln2_g; b2.dbeta = L.g_ln2_b; b2.rows = Rt; b2.C = E;
  b2.p_drop = L.p_drop; b2.seed = L.seed_drop2; b2.out_scale = 1.f;
  const bool ln_defer = ln_deferred(L) && L.ws_e3 && (!L.sel_idx || (L.ws_s0 && L.ws_s1));   // the partial rows wait for w2vs_layer_wgrads in the layer's own slab
  b2.ws = L.ws_f; b2.ws_bytes = (int64_t)R * F * 2;   // ws_f is not live yet: dgamma/dbeta partial slab
  if (ln_defer) { b2.ws = L.ln_part; b2.ws_bytes = ln_half_bytes(L); }
  TRY(ln_bwd(b2, s, ln_defer));
  // With a fourth [R,E] scratch (ws_e3) every operand of the four weight gradients stays alive to the end of the layer
  // (d_f in ws_e0, d_hpre in ws_f, d_a in ws_e3, d_qkv in ws_qkv) and they run as ONE grouped launch without a K split
  // (gemm_tn_group: 216 tiles of full-length K loops, no partial-tile slab, no summing launches).  Not in selected-rows
  // mode, whose scatter step reuses ws_e0 / ws_f - unless the caller hands it two more [R,E] buffers (ws_s0 / ws_s1, round 4)
  // and asks for deferral: the last layer's weight gradients then join the grouped launch of its neighbour.
  const bool sel_defer = sel && L.ws_e3 != nullptr && L.defer_wgrads && L.ws_s0 && L.ws_s1;
  const bool defer = L.ws_e3 != nullptr && (!sel || sel_defer);
  void* d_a = defer ? L.ws_e3 : L.ws_e0;
  // fc2: wgrad, bias, dgrad chained through GELU -> d_hpre (ws_f)
  if (!defer) TRY(lin_wgrad(L.ws_e0, L.h, L.g_w2, L.g_b2, Rt, E, F, cu, s, L.tn_ws, L.tn_ws_bytes));
  if (!pre_t) TRY(transpose2d(L.w2, L.wt_scratch, E, F, 1, s));           // [E,F] -> [F,E]
  TRY(lin_dgrad(L.ws_e0, pre_t ? L.w2_t : L.wt_scratch, L.ws_f, L.hpre, Rt, E, F, EPI_MUL, s));
  // fc1: wgrad, bias, dgrad + residual branch -> d_x1 (ws_e2)
  if (!defer) TRY(lin_wgrad(L.ws_f, L.x1, L.g_w1, L.g_b1, Rt, F, E, cu, s, L.tn_ws, L.tn_ws_bytes));
  if (!pre_t) TRY(transpose2d(L.w1, L.wt_scratch, F, E, 1, s));           // [F,E] -> [E,F]
  // post-LN: x1 feeds fc1 AND the second residual -> add its branch gradient; pre-LN: x1 = LN(s1) feeds fc1 only
  TRY(lin_dgrad(L.ws_f, pre_t ? L.w1_t : L.wt_scratch, L.ws_e2, L.post_ln ? L.ws_e1 : nullptr, Rt, F, E,
                L.post_ln ? EPI_ADD : EPI_NONE, s));
  // LN1 backward: d_a = ds1 o dropmask, d_xin_a = ds1 (ws_e1); pre-LN: ds1 = LNbwd(d_x1) + d_s1 (ws_e1, from LN2's dres)
  LnBwdDesc b1{};
  b1.x = L.s1; b1.gamma = L.ln1_g; b1.beta = L.ln1_b; b1.mean = L.mean1; b1.rstd = L.rstd1; b1.dy = L.ws_e2;
  if (!L.post_ln) b1.dsum = L.ws_e1;
  b1.dx = d_a; b1.dres = (L.post_ln || sel) ? L.ws_e1 : L.d_stream_in; b1.dgamma = L.g_ln1_g; b1.dbeta = L.g_ln1_b; b1.rows = Rt; b1.C = E;
  b1.p_drop = L.p_drop; b1.seed = L.seed_drop1; b1.out_scale = 1.f;
  if (defer) {                                         // ws_f still holds d_hpre: the partial slab goes to the GEMM scratch
    b1.ws = L.tn_ws; b1.ws_bytes = L.tn_ws_bytes;
  } else {
    b1.ws = L.ws_f; b1.ws_bytes = (int64_t)R * F * 2;   // fc1's wgrad/dgrad (enqueued above) were its last readers
  }
  if (ln_defer) { b1.ws = (char*)L.ln_part + ln_half_bytes(L); b1.ws_bytes = ln_half_bytes(L); }
  TRY(ln_bwd(b1, s, ln_defer));
  // out_proj
  if (!defer) TRY(lin_wgrad(L.ws_e0, ctx, L.g_wo, L.g_bo, Rt, E, E, cu, s, L.tn_ws, L.tn_ws_bytes));
  if (!pre_t) TRY(transpose2d(L.wo, L.wt_scratch, E, E, 1, s));
  TRY(lin_dgrad(d_a, pre_t ? L.wo_t : L.wt_scratch, L.ws_e2, nullptr, Rt, E, E, EPI_NONE, s));   // d_ctx
  const void* d_ctx = L.ws_e2;
  const void* d_res = L.ws_e1;
  if (sel) {
    // back to token rows: d_ctx and the residual-branch gradient are zero outside the selected rows
    const size_t bytes = (size_t)R * E * 2;
    void* s0 = sel_defer ? L.ws_s0 : L.ws_e0;                       // ws_e0 (d_a) is dead after the out_proj pair ...
    void* s1 = sel_defer ? L.ws_s1 : L.ws_f;                        // ... ws_f (d_hpre, LN slab) too - unless the wgrads are deferred
    if (hipMemsetAsync(s0, 0, bytes, s) != hipSuccess || hipMemsetAsync(s1, 0, bytes, s) != hipSuccess)
      return set_error("layer_bwd: memset failed");
    TRY(gather_rows(L.ws_e2, L.sel_idx, s0, Rt, E, 1, s));
    TRY(gather_rows(L.ws_e1, L.sel_idx, s1, Rt, E, 1, s));
    d_ctx = s0; d_res = s1;
    if (!L.post_ln && hipMemcpyAsync(L.d_stream_in, s1, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return set_error("layer_bwd: copy failed");
    // dq rows past n_q are not written by the attention backward: the QKV GEMMs read them
    if (hipMemsetAsync(L.ws_qkv, 0, (size_t)R * 3 * E * 2, s) != hipSuccess) return set_error("layer_bwd: memset failed");
  }
  // attention
  AttnDesc a{};
  fill_attn(L, a);
  if (sel) a.Nq = L.n_q;
  const long E2 = 2L * E;
  a.dout = d_ctx; a.delta = L.delta;
  a.dq = L.ws_qkv; a.dk = (char*)L.ws_qkv + E2; a.dv = (char*)L.ws_qkv + 2 * E2;
  TRY(attn_bwd(a, s));
  // fused QKV projection
  if (!defer) TRY(lin_wgrad(L.ws_qkv, L.x_in, L.g_wqkv, L.g_bqkv, R, 3 * E, E, cu, s, L.tn_ws, L.tn_ws_bytes));
  if (!pre_t) TRY(transpose2d(L.wqkv, L.wt_scratch, 3 * E, E, 1, s));     // [3E,E] -> [E,3E]
  // post-LN: x_in is also the residual -> + d_res; pre-LN: x_in = LN(s_in) feeds the projection only, d_res went to d_stream_in
  TRY(lin_dgrad(L.ws_qkv, pre_t ? L.wqkv_t : L.wt_scratch, L.d_in, L.post_ln ? d_res : nullptr, R, 3 * E, E,
                L.post_ln ? EPI_ADD : EPI_NONE, s));
  if (L.defer_wgrads && !defer) return set_error("layer_bwd: defer_wgrads needs ws_e3 (and, with sel_idx, ws_s0 / ws_s1)");
  if (defer && !L.defer_wgrads) {
    GemmDesc g[4];
    layer_wgrad_descs(L, g);
    TRY(gemm_tn_group(g, 4, cu, s));
  }
  return 0;
}

// parts[i]: bit 0 fc1, bit 1 fc2, bit 2 fused QKV, bit 3 out_proj weight (+ bias) gradient of Ls[i]; bit 4: sum that layer's
// LayerNorm partial slabs.  NULL = everything of every layer.
int layer_wgrads(const w2vs_layer_desc* Ls, const int32_t* parts, int n, hipStream_t s) {
  if (!Ls || n < 1 || n > 6) return set_error("layer_wgrads: 1 to 6 layers");
  GemmDesc g[12];
  LnPartial lp[8];
  int ng = 0, nl = 0;
  for (int i = 0; i < n; ++i) {
    const w2vs_layer_desc& L = Ls[i];
    const int part = parts ? parts[i] : 31;
    if (part & ~31) return set_error("layer_wgrads: parts has bits 0-4 only");
    TRY(layer_check(L));
    if (!L.ws_e3 || (L.sel_idx && (!L.ws_s0 || !L.ws_s1 || !L.ctx_sel)) || !L.ws_f || !L.ws_e0 || !L.ws_qkv || !L.g_wqkv || !L.g_bqkv ||
        !L.g_wo || !L.g_bo || !L.g_w1 || !L.g_b1 || !L.g_w2 || !L.g_b2)
      return set_error("layer_wgrads: the layer was not run with defer_wgrads (ws_e3; ws_s0 / ws_s1 with sel_idx) or lacks gradient pointers");
    GemmDesc four[4];
    layer_wgrad_descs(L, four);
    for (int k = 0; k < 4; ++k)
      if (part >> k & 1) {
        if (ng == 12) return set_error("layer_wgrads: at most 12 GEMMs per call");
        g[ng++] = four[k];
      }
    if ((part & 16) && ln_deferred(L)) {
      if (nl + 2 > 8) return set_error("layer_wgrads: at most 8 LayerNorm reductions per call");
      if (L.E != Ls[0].E) return set_error("layer_wgrads: layers of different width");
      if (!L.g_ln1_g || !L.g_ln1_b || !L.g_ln2_g || !L.g_ln2_b) return set_error("layer_wgrads: null LayerNorm gradient pointer");
      const int G = ln_bwd_grid(L.sel_idx ? L.n_sel : L.B * L.N, L.E, ln_half_bytes(L));
      lp[nl++] = LnPartial{(const float*)L.ln_part, L.g_ln2_g, L.g_ln2_b, G};
      lp[nl++] = LnPartial{(const float*)((const char*)L.ln_part + ln_half_bytes(L)), L.g_ln1_g, L.g_ln1_b, G};
    }
  }
  if (nl) TRY(ln_reduce_many(lp, nl, Ls[0].E, s));
  if (!ng) return 0;
  return gemm_tn_group(g, ng, Ls[0].num_cu > 0 ? Ls[0].num_cu : 256, s);
}

}  // namespace w2vs
