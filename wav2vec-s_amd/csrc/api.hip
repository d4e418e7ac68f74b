// extern "C" surface of libw2vs (declared in include/w2vs.h).
#include <string.h>
#include <string>
#include "w2vs_internal.h"

namespace w2vs {
static thread_local std::string g_err;
int set_error(const char* msg) { g_err = msg; return W2VS_ERR_INVALID; }
int hip_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return W2VS_ERR_HIP;
}
}  // namespace w2vs

using namespace w2vs;
#define ST(s) ((hipStream_t)(s))
#define NONNULL(d) if (!(d)) return set_error("null descriptor")

extern "C" {
int w2vs_abi_version(void) { return W2VS_ABI_VERSION; }
const char* w2vs_last_error(void) { return g_err.c_str(); }
int w2vs_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(w2vs_gemm_desc);
    case 1: return (int)sizeof(w2vs_ln_fwd_desc);
    case 2: return (int)sizeof(w2vs_ln_bwd_desc);
    case 3: return (int)sizeof(w2vs_enc_prologue_desc);
    case 4: return (int)sizeof(w2vs_attn_desc);
    case 5: return (int)sizeof(w2vs_quant_desc);
    case 6: return (int)sizeof(w2vs_nce_desc);
    case 7: return (int)sizeof(w2vs_layer_desc);
    case 8: return (int)sizeof(w2vs_collate_desc);
    case 9: return (int)sizeof(w2vs_infonce_loss_desc);
  }
  return -1;
}
int w2vs_prof_enable(int stride) { prof_enable(stride); return 0; }
int w2vs_prof_read(int id, double* ms, double* flops, int* n) { return prof_read(id, ms, flops, n); }
int w2vs_prof_read_raw(int id, double* ms, double* fl, int* n) { return prof_read_raw(id, ms, fl, n); }
int64_t w2vs_prof_launches(int id) { return prof_launches(id); }
double w2vs_prof_flops(int id) { return prof_flops_all(id); }
int w2vs_gemm_nt(const w2vs_gemm_desc* d, void* s) { NONNULL(d); return gemm_nt(*d, ST(s)); }
int w2vs_gemm_tn(const w2vs_gemm_desc* d, int cu, void* s) { NONNULL(d); return gemm_tn(*d, cu, ST(s)); }
int w2vs_gemm_tn_group(const w2vs_gemm_desc* d, int32_t n, int32_t cu, void* s) { NONNULL(d); return gemm_tn_group(d, n, cu, ST(s)); }
int w2vs_conv0_fwd(const void* wave, const void* w, const void* cb, const void* lw, const void* lb, void* y, float* mean,
                   float* rstd, int B, int L, int C, int k, int st, void* s) {
  return conv0_fwd(wave, w, cb, lw, lb, y, mean, rstd, B, L, C, k, st, ST(s));
}
int w2vs_conv0_bwd(const void* wave, const void* w, const void* cb, const void* lw, const void* lb, const float* mean,
                   const float* rstd, const void* dy, float* dw, float* dcb, float* dlw, float* dlb, int B, int L, int C,
                   int k, int st, void* s) {
  return conv0_bwd(wave, w, cb, lw, lb, mean, rstd, dy, dw, dcb, dlw, dlb, B, L, C, k, st, ST(s));
}
int w2vs_conv0_gn_fwd(const void* wave, const void* w, const void* cb, const void* g, const void* b, void* y, float* stat,
                      int B, int L, int C, int k, int st, void* s) {
  return conv0_gn_fwd(wave, w, cb, g, b, y, stat, B, L, C, k, st, ST(s));
}
int w2vs_conv0_gn_bwd(const void* wave, const void* w, const void* cb, const void* g, const void* b, const float* stat,
                      const void* dy, float* bstat, float* dw, float* dcb, float* dg, float* db, int B, int L, int C, int k,
                      int st, void* s) {
  return conv0_gn_bwd(wave, w, cb, g, b, stat, dy, bstat, dw, dcb, dg, db, B, L, C, k, st, ST(s));
}
int w2vs_ln_fwd(const w2vs_ln_fwd_desc* d, void* s) { NONNULL(d); return ln_fwd(*d, ST(s)); }
int w2vs_ln_bwd(const w2vs_ln_bwd_desc* d, void* s) { NONNULL(d); return ln_bwd(*d, ST(s)); }
int w2vs_enc_prologue_fwd(const w2vs_enc_prologue_desc* d, void* s) { NONNULL(d); return enc_prologue_fwd(*d, ST(s)); }
int w2vs_enc_prologue_bwd(const w2vs_enc_prologue_desc* d, void* s) { NONNULL(d); return enc_prologue_bwd(*d, ST(s)); }
int w2vs_attn_fwd(const w2vs_attn_desc* d, void* s) { NONNULL(d); return attn_fwd(*d, ST(s)); }
int w2vs_attn_bwd(const w2vs_attn_desc* d, void* s) { NONNULL(d); return attn_bwd(*d, ST(s)); }
int w2vs_layer_fwd(const w2vs_layer_desc* d, void* s) { NONNULL(d); return layer_fwd(*d, ST(s)); }
int w2vs_layer_bwd(const w2vs_layer_desc* d, void* s) { NONNULL(d); return layer_bwd(*d, ST(s)); }
int w2vs_layer_wgrads(const w2vs_layer_desc* d, int32_t n, void* s) { NONNULL(d); if (n > 2) return set_error("layer_wgrads: 1 or 2 layers (w2vs_layer_wgrads_parts takes more)"); return layer_wgrads(d, nullptr, n, ST(s)); }
int w2vs_layer_wgrads_parts(const w2vs_layer_desc* d, const int32_t* parts, int32_t n, void* s) { NONNULL(d); NONNULL(parts); return layer_wgrads(d, parts, n, ST(s)); }
int w2vs_quant_fwd(const w2vs_quant_desc* d, void* s) { NONNULL(d); return quant_fwd(*d, ST(s)); }
int w2vs_quant_bwd(const w2vs_quant_desc* d, void* s) { NONNULL(d); return quant_bwd(*d, ST(s)); }
int w2vs_nce_fwd(const w2vs_nce_desc* d, void* s) { NONNULL(d); return nce_fwd(*d, ST(s)); }
int w2vs_nce_bwd(const w2vs_nce_desc* d, void* s) { NONNULL(d); return nce_bwd(*d, ST(s)); }
int w2vs_ce_rows(const float* logits, int64_t R, int32_t W, float* out3, float* dl, void* s) { return ce_rows(logits, R, W, out3, dl, ST(s)); }
int w2vs_infonce_loss(const w2vs_infonce_loss_desc* d, void* s) { NONNULL(d); return infonce_loss(*d, ST(s)); }
int w2vs_infonce_loss_bwd(const float* g, float* dl, int64_t n, float c_pen, float c_ppl, float* dsc, void* s) {
  return infonce_loss_bwd(g, dl, n, c_pen, c_ppl, dsc, ST(s));
}
int w2vs_gather_rows(const void* src, const int32_t* idx, void* dst, int64_t R, int32_t C, int32_t sc, void* s) {
  return gather_rows(src, idx, dst, R, C, sc, ST(s));
}
int w2vs_transpose2d(const void* in, void* out, int32_t R, int32_t C, int32_t batch, void* s) { return transpose2d(in, out, R, C, batch, ST(s)); }
int w2vs_gemm_tune(int32_t nt_mode, int32_t lc_height, int32_t tn_lc) { gemm_tune(nt_mode, lc_height, tn_lc); return 0; }
int w2vs_gemm_tn8_max_split(int32_t s) { gemm_tn8_max_split(s); return 0; }
int w2vs_gemm_last_group_form(void) { return gemm_last_group_form(); }
int w2vs_attn_tune(int32_t variant) { attn_tune(variant); return 0; }
int64_t w2vs_attn_drop_bits_bytes(int32_t B, int32_t H, int32_t N, int32_t Nq) {
  const int64_t nq = ((Nq > 0 ? Nq : N) + 31) / 32, nk = (N + 31) / 32;
  return (int64_t)B * H * nq * nk * 128;
}
int w2vs_transpose_multi(const w2vs_transpose_item* items, int32_t n, void* s) { return transpose_multi(items, n, ST(s)); }
int w2vs_f32_to_bf16(const float* in, void* out, int64_t n, float scale, void* s) { return f32_to_bf16(in, out, n, scale, ST(s)); }
int w2vs_bf16_to_f32(const void* in, float* out, int64_t n, void* s) { return bf16_to_f32(in, out, n, ST(s)); }
int w2vs_dropout(const void* in, void* out, int64_t n, float p, uint64_t seed, void* s) { return dropout(in, out, n, p, seed, ST(s)); }
int w2vs_relu_gate(const void* x, const void* gate, void* out, int64_t n, void* s) { return relu_gate(x, gate, out, n, ST(s)); }
int w2vs_adam_step(float* p32, void* p16, float* m, float* v, const float* g, int64_t n, float lr, float b1, float b2, float eps,
                   float wd, int32_t step, const float* scale_dev, float scale_host, void* s) {
  return adam_step(p32, p16, m, v, g, n, lr, b1, b2, eps, wd, step, scale_dev, scale_host, ST(s));
}
int w2vs_batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences, int32_t bsz_mult,
                       int32_t* batch_ends, int32_t* n_batches) {
  return batch_by_size(num_tokens, n, max_tokens, max_sentences, bsz_mult, batch_ends, n_batches);
}
int32_t w2vs_collate_chunks(int32_t max_size) { return collate_chunks(max_size); }
int w2vs_collate(const w2vs_collate_desc* d, void* s) { NONNULL(d); return collate(*d, ST(s)); }
int w2vs_sumsq(const float* x, int64_t n, float* out, void* s) { return sumsq(x, n, out, ST(s)); }
int w2vs_clip_scale(const float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, void* s) {
  return clip_scale(sumsq, scale_dev, scale_host, clip, out3, ST(s));
}
int w2vs_clip_scale_acc(float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, float* bad_acc, void* s) {
  return clip_scale_acc(sumsq, scale_dev, scale_host, clip, out3, bad_acc, ST(s));
}
int w2vs_colsum(const void* in, float* out, int64_t M, int32_t N, int64_t ld, void* s) { return colsum(in, out, M, N, ld, ST(s)); }
}
