// Host-side internal declarations shared by the translation units of libw2vs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/w2vs.h"

// Tuning switches.  libw2vs.so reads NO environment variable: every selector runs on its measured default, the configuration
// the tests cover.  `make -C wav2vec-s_amd/csrc tuning` builds libw2vs_tuning.so with -DW2VS_TUNING, where the same
// expressions read the environment once (tools/*probe*.py, tools/nt_force_sweep*.sh load it through W2VS_LIB).
#ifdef W2VS_TUNING
#include <stdlib.h>
#define W2VS_ENV_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#define W2VS_ENV_SET(name) (getenv(name) != nullptr)
#define W2VS_ENV_STR(name) (getenv(name))
#else
#define W2VS_ENV_INT(name, dflt) (dflt)
#define W2VS_ENV_SET(name) (false)
#define W2VS_ENV_STR(name) ((const char*)nullptr)
#endif

namespace w2vs {

int set_error(const char* msg);                 // records msg, returns W2VS_ERR_INVALID
int hip_check(hipError_t e, const char* what);  // 0, or records + returns W2VS_ERR_HIP

typedef w2vs_gemm_desc GemmDesc;
typedef w2vs_ln_fwd_desc LnFwdDesc;
typedef w2vs_ln_bwd_desc LnBwdDesc;
typedef w2vs_enc_prologue_desc EncPrologueDesc;
typedef w2vs_attn_desc AttnDesc;
typedef w2vs_quant_desc QuantDesc;
typedef w2vs_nce_desc NceDesc;
typedef w2vs_infonce_loss_desc InfonceLossDesc;

int gemm_nt(const GemmDesc& d, hipStream_t s);
int gemm_tn(const GemmDesc& d, int num_cu_hint, hipStream_t s);
int gemm_tn_group(const GemmDesc* ds, int n, int num_cu_hint, hipStream_t s);
void gemm_tune(int nt_mode, int lc_height, int tn_lc);
void gemm_tn8_max_split(int s);
int gemm_last_group_form();
void attn_tune(int variant);
void prof_enable(int stride);
int prof_read(int id, double* total_ms, double* total_flops, int* launches);
int prof_read_raw(int id, double* total_ms, double* total_flops, int* launches);
long prof_launches(int id);
double prof_flops_all(int id);
int conv0_fwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, void* y,
              float* mean, float* rstd, int B, int L, int C, int k, int s, hipStream_t st);
int conv0_bwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, const float* mean,
              const float* rstd, const void* dy, float* dw, float* dcbias, float* dlnw, float* dlnb, int B, int L, int C,
              int k, int s, hipStream_t st);
// conv0.hip: matrix-core form for C == 512, k <= 10 (what conv0_fwd/bwd dispatch to when it applies)
bool conv0_mfma_ok(int C, int k);
int conv0_mfma_fwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, void* y,
                   float* mean, float* rstd, int B, int L, int k, int s, hipStream_t st);
int conv0_mfma_bwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, const float* mean,
                   const float* rstd, const void* dy, float* dw, float* dcbias, float* dlnw, float* dlnb, int B, int L, int k,
                   int s, hipStream_t st);
int conv0_gn_fwd(const void* wave, const void* w, const void* cbias, const void* g, const void* b, void* y, float* stat,
                 int B, int L, int C, int k, int s, hipStream_t st);
int conv0_gn_bwd(const void* wave, const void* w, const void* cbias, const void* g, const void* b, const float* stat,
                 const void* dy, float* bstat, float* dw, float* dcbias, float* dg, float* db, int B, int L, int C, int k,
                 int s, hipStream_t st);
int ln_fwd(const LnFwdDesc& d, hipStream_t st);
// leave_partials: the dgamma / dbeta partial rows stay in d.ws ([ln_bwd_grid(...)][2C] fp32) and ln_reduce_many sums them later
int ln_bwd(const LnBwdDesc& d, hipStream_t st, bool leave_partials = false);
struct LnPartial { const float* part; float* dg; float* db; int G; };
inline int64_t ln_min_slab_bytes(int C) { return (int64_t)sizeof(float) * 2 * C * 64; }
int ln_bwd_grid(long rows, int C, int64_t ws_bytes);
int ln_reduce_many(const LnPartial* r, int n, int C, hipStream_t st);
int enc_prologue_fwd(const EncPrologueDesc& d, hipStream_t st);
int enc_prologue_bwd(const EncPrologueDesc& d, hipStream_t st);
int attn_fwd(const AttnDesc& d, hipStream_t st);
int attn_bwd(const AttnDesc& d, hipStream_t st);
int layer_fwd(const w2vs_layer_desc& L, hipStream_t st);
int layer_bwd(const w2vs_layer_desc& L, hipStream_t st);
int layer_wgrads(const w2vs_layer_desc* Ls, const int32_t* parts, int n, hipStream_t st);
int quant_fwd(const QuantDesc& d, hipStream_t st);
int quant_bwd(const QuantDesc& d, hipStream_t st);
int nce_fwd(const NceDesc& d, hipStream_t st);
int nce_bwd(const NceDesc& d, hipStream_t st);
int ce_rows(const float* logits, long R, int W, float* out3, float* dlogits, hipStream_t st);
int infonce_loss(const InfonceLossDesc& d, hipStream_t st);
int infonce_loss_bwd(const float* g, float* dlogits, long n, float c_pen, float c_ppl, float* dsc, hipStream_t st);
int gather_rows(const void* src, const int* idx, void* dst, long R, int C, int scatter, hipStream_t st);
int transpose2d(const void* in, void* out, int R, int C, int batch, hipStream_t st);
int transpose_multi(const w2vs_transpose_item* items, int n, hipStream_t st);
int f32_to_bf16(const float* in, void* out, long n, float scale, hipStream_t st);
int bf16_to_f32(const void* in, float* out, long n, hipStream_t st);
int dropout(const void* in, void* out, long n, float p, uint64_t seed, hipStream_t st);
int relu_gate(const void* x, const void* gate, void* out, long n, hipStream_t st);
// data.hip (row f3)
int batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences, int32_t bsz_mult,
                  int32_t* ends, int32_t* n_batches);
int collate_chunks(int max_size);
int collate(const w2vs_collate_desc& d, hipStream_t st);
int adam_step(float* p32, void* p16, float* m, float* v, const float* g, long n, float lr, float b1, float b2, float eps,
              float wd, int step, const float* scale_dev, float scale_host, hipStream_t st);
int sumsq(const float* x, long n, float* out, hipStream_t st);
int clip_scale(const float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, hipStream_t st);
int clip_scale_acc(float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, float* bad_acc, hipStream_t st);
int colsum(const void* in, float* out, long M, int N, long ld, hipStream_t st);

}  // namespace w2vs
