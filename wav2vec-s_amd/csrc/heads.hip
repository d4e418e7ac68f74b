// Loss-head kernels: Gumbel vector quantizer, InfoNCE logits, cross-entropy, row gathers.
// All HBM / latency bound, one wave per row, fp32 math, wavefront-shuffle reductions.
//
//   gumbel_quantize   a14  fs/modules/gumbel_vector_quantizer.py:141-202
//   infonce_logits    a16 gather + a17 compute_preds   fs/models/wav2vec/wav2vec2.py:521-542
//   ce_rows           a19  fs/criterions/wav2vec_criterion.py:64-68, 133-155
#include <mutex>
#include <stdlib.h>
#include "common.h"
#include "w2vs_internal.h"

namespace w2vs {

// ------------------------------------------------------------------------------------------------
// argmax with torch's tie rule (first index wins)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

struct QuantP {
  const bf16* logits;   // [R, G*V]
  const float* logits32;  // [R, G*V] fp32 without bias (preferred: no rounding before the argmax), or null
  const bf16* lbias;      // [G*V] bias added to logits32, or null
  const float* noise;   // [R*G, V] gumbel samples or null (then seed is used when training)
  const bf16* vars;     // [G*V, D]
  bf16* q;              // [R, G*D]
  int* idx;             // [R, G] chosen code
  float* hard_cnt;      // [G*V] histogram of argmax(logits)
  float* prob_sum;      // [G*V] sum over rows of softmax(logits)
  // backward
  const bf16* dq;       // [R, G*D]
  const bf16* dsoft;    // [R, G*V]  dq . vars^T  (from the GEMM)
  const float* dsoft32; // the same product as fp32 (preferred: (ds - <soft, ds>) cancels, bf16 ds costs ~5 % there)
  const float* cvec;    // [G*V] d(prob_ppl)/d(avg_prob)
  bf16* dlogits;        // [R, G*V]
  float* dvars;         // [G*V, D] fp32 accumulators
  float ppl_grad;       // dLoss/d(prob_ppl)
  const float* ppl_dev;  // optional device multiplier
  int R, G, V, D;
  float tau; int training; uint64_t seed;
};

constexpr int QV_MAX = 5;  // V <= 320 : up to 5 codes per lane

template <bool BWD>
__global__ __launch_bounds__(256) void quant_kernel(QuantP p) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const int V = p.V, G = p.G, D = p.D;
  const long rows = (long)p.R * G;
  // forward: softmax probabilities are summed per code.  When every wave keeps one group (nwaves % G == 0) the sums
  // stay in registers over the wave's rows and reach memory once per block; per-row atomics put 2000 adders on
  // each of the 640 words.
  const bool keep = !BWD && (nwaves % G) == 0;
  float pacc[QV_MAX];
#pragma unroll
  for (int j = 0; j < QV_MAX; ++j) pacc[j] = 0.f;
  for (long rg = wave_id; rg < rows; rg += nwaves) {
    const long row = rg / G;
    const int g = (int)(rg % G);
    const long lofs = row * (long)(G * V) + (long)g * V;
    float x[QV_MAX], y[QV_MAX];
    float xmax = -INFINITY;
    int xarg = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < QV_MAX; ++j) {
      int v = lane + 64 * j;
      if (v < V) {
        if (p.logits32) x[j] = p.logits32[lofs + v] + (p.lbias ? bf2f(p.lbias[g * V + v]) : 0.f);
        else x[j] = bf2f(p.logits[lofs + v]);
      } else x[j] = -INFINITY;
      if (x[j] > xmax) { xmax = x[j]; xarg = v; }
    }
    wave_argmax(xmax, xarg);
    // softmax(logits) for the average-probability perplexity
    float ps[QV_MAX], se = 0.f;
#pragma unroll
    for (int j = 0; j < QV_MAX; ++j) { ps[j] = __expf(x[j] - xmax); se += ps[j]; }
    se = wave_sum(se);
    const float inv_se = 1.f / se;
    int sel = xarg;
    float ys[QV_MAX], yinv = 0.f;
    if (p.training) {
      float ymax = -INFINITY;
      int yarg = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < QV_MAX; ++j) {
        int v = lane + 64 * j;
        float gn = 0.f;
        if (v < V) {
          if (p.noise) gn = p.noise[rg * V + v];
          else {
            float e = -__logf(u01(hash32(p.seed, (uint64_t)rg * V + v)));  // Exp(1)
            gn = -__logf(fmaxf(e, 1e-20f));
          }
        }
        y[j] = (v < V) ? (x[j] + gn) / p.tau : -INFINITY;
        if (y[j] > ymax) { ymax = y[j]; yarg = v; }
      }
      wave_argmax(ymax, yarg);
      sel = yarg;
      float sy = 0.f;
#pragma unroll
      for (int j = 0; j < QV_MAX; ++j) { ys[j] = __expf(y[j] - ymax); sy += ys[j]; }
      yinv = 1.f / wave_sum(sy);
    }
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < QV_MAX; ++j) {
        int v = lane + 64 * j;
        if (keep) pacc[j] += ps[j] * inv_se;
        else if (v < V) atomicAdd(&p.prob_sum[g * V + v], ps[j] * inv_se);
      }
      if (lane == 0) {
        atomicAdd(&p.hard_cnt[g * V + xarg], 1.0f);
        p.idx[rg] = sel;
      }
      // codebook row copy: D bf16 = D/8 chunks of 16 B
      const bf16* src = p.vars + ((long)g * V + sel) * D;
      bf16* dst = p.q + row * (long)(G * D) + (long)g * D;
      for (int ch = lane; ch < D / 8; ch += 64) *(u32x4*)(dst + ch * 8) = *(const u32x4*)(src + ch * 8);
    } else {
      // d logits = [softmax-bwd of the straight-through gumbel path] / tau
      //          + ppl_grad/R * p o (c - <p, c>)          (diversity term through avg_probs)
      const long dofs = row * (long)(G * V) + (long)g * V;
      float ds[QV_MAX], dot1 = 0.f, dot2 = 0.f;
#pragma unroll
      for (int j = 0; j < QV_MAX; ++j) {
        int v = lane + 64 * j;
        float soft = p.training ? ys[j] * yinv : 0.f;
        ds[j] = (v < V && p.training) ? (p.dsoft32 ? p.dsoft32[dofs + v] : bf2f(p.dsoft[dofs + v])) : 0.f;
        dot1 += soft * ds[j];
        float pj = ps[j] * inv_se;
        dot2 += (v < V) ? pj * p.cvec[g * V + v] : 0.f;
      }
      dot1 = wave_sum(dot1);
      dot2 = wave_sum(dot2);
      bf16* dl = p.dlogits + row * (long)(G * V) + (long)g * V;
      const float sc = (p.ppl_dev ? p.ppl_grad * p.ppl_dev[0] : p.ppl_grad) / (float)p.R;
#pragma unroll
      for (int j = 0; j < QV_MAX; ++j) {
        int v = lane + 64 * j;
        if (v < V) {
          float soft = p.training ? ys[j] * yinv : 0.f;
          float pj = ps[j] * inv_se;
          float gr = soft * (ds[j] - dot1) / p.tau + sc * pj * (p.cvec[g * V + v] - dot2);
          dl[v] = f2bf(gr);
        }
      }
      // d vars[g*V + sel] += dq[row, g]
      const bf16* dqp = p.dq + row * (long)(G * D) + (long)g * D;
      float* dv = p.dvars + ((long)g * V + sel) * D;
      for (int e = lane; e < D; e += 64) atomicAdd(&dv[e], bf2f(dqp[e]));
    }
  }
  if (keep) {   // block-uniform
    __shared__ float pred[4][QV_MAX * 64];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < QV_MAX; ++j) pred[w][lane + 64 * j] = pacc[j];
    __syncthreads();
    // waves w and w' of a block share a group iff (w - w') % G == 0; the first wave of each group folds and flushes
    const int g = (int)(wave_id % G);
    bool first = true;
    for (int w2 = 0; w2 < w; ++w2) first = first && (((blockIdx.x * 4 + w2) % G) != g);
    if (first && wave_id < rows) {
#pragma unroll
      for (int j = 0; j < QV_MAX; ++j) {
        const int v = lane + 64 * j;
        float sum = 0.f;
        for (int w2 = w; w2 < 4; ++w2)
          if (((blockIdx.x * 4 + w2) % G) == g) sum += pred[w2][v];
        if (v < V) atomicAdd(&p.prob_sum[g * V + v], sum);
      }
    }
  }
}

// perplexities from the two histograms; also the vector c = d(prob_ppl)/d(avg_prob)
__global__ void quant_finalize_kernel(const float* hard_cnt, const float* prob_sum, int R, int G, int V, float* out2,
                                      float* cvec) {
  __shared__ float red[2][64];
  const int tid = threadIdx.x;  // 256 threads, one block
  float code_ppl = 0.f, prob_ppl = 0.f;
  for (int g = 0; g < G; ++g) {
    float hc = 0.f, hp = 0.f;
    for (int v = tid; v < V; v += 256) {
      float a = hard_cnt[g * V + v] / (float)R;
      float b = prob_sum[g * V + v] / (float)R;
      hc += a * logf(a + 1e-7f);
      hp += b * logf(b + 1e-7f);
    }
    hc = wave_sum(hc); hp = wave_sum(hp);
    if ((tid & 63) == 0) { red[0][tid >> 6] = hc; red[1][tid >> 6] = hp; }
    __syncthreads();
    float tc = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    float tp = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    __syncthreads();
    const float eg = expf(-tp);
    code_ppl += expf(-tc);
    prob_ppl += eg;
    if (cvec)
      for (int v = tid; v < V; v += 256) {
        float b = prob_sum[g * V + v] / (float)R;
        cvec[g * V + v] = -eg * (logf(b + 1e-7f) + b / (b + 1e-7f));
      }
  }
  if (tid == 0) { out2[0] = prob_ppl; out2[1] = code_ppl; }
}

static int quant_fill(const QuantDesc& d, QuantP& p) {
  p.logits = (const bf16*)d.logits; p.noise = d.noise; p.vars = (const bf16*)d.vars; p.q = (bf16*)d.q; p.idx = d.idx;
  p.hard_cnt = d.hard_cnt; p.prob_sum = d.prob_sum; p.dq = (const bf16*)d.dq; p.dsoft = (const bf16*)d.dsoft; p.cvec = d.cvec;
  p.dlogits = (bf16*)d.dlogits; p.dvars = d.dvars; p.ppl_grad = d.ppl_grad; p.ppl_dev = d.ppl_grad_dev; p.R = d.R; p.G = d.G; p.V = d.V; p.D = d.D;
  p.tau = d.tau; p.training = d.training; p.seed = d.seed; p.dsoft32 = d.dsoft_f32;
  p.logits32 = d.logits_f32; p.lbias = (const bf16*)d.logit_bias;
  if ((!p.logits && !p.logits32) || !p.vars) return set_error("quantizer: null pointer");
  if (p.V > 64 * QV_MAX || p.V < 1) return set_error("quantizer: num_vars per group must be in [1, 320]");
  if (p.D % 8) return set_error("quantizer: var_dim must be a multiple of 8");
  if (p.R <= 0 || p.G <= 0) return set_error("quantizer: bad R/G");
  if (p.training && !(p.tau > 0.f)) return set_error("quantizer: tau must be positive");
  return 0;
}

__global__ __launch_bounds__(256) void zero2_kernel(float* a, float* b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { a[i] = 0.f; b[i] = 0.f; }
}

int quant_fwd(const QuantDesc& d, hipStream_t st) {
  QuantP p{};
  if (int e = quant_fill(d, p)) return e;
  if (!p.q || !p.idx || !p.hard_cnt || !p.prob_sum || !d.ppl_out) return set_error("quant_fwd: null pointer");
  // one small launch instead of two hipMemsetAsync (a fill kernel of ~5 us each for 2.5 KB)
  hipLaunchKernelGGL(zero2_kernel, dim3(std::min((p.G * p.V + 255) / 256, 64)), dim3(256), 0, st, p.hard_cnt, p.prob_sum, (long)p.G * p.V);
  long rows = (long)p.R * p.G;
  int grid = (int)std::min<long>((rows + 3) / 4, 256);   // few blocks: the probability sums flush once per block
  hipLaunchKernelGGL(quant_kernel<false>, dim3(grid), dim3(256), 0, st, p);
  hipLaunchKernelGGL(quant_finalize_kernel, dim3(1), dim3(256), 0, st, p.hard_cnt, p.prob_sum, p.R, p.G, p.V, d.ppl_out,
                     d.cvec_out);
  return hip_check(hipGetLastError(), "quant_fwd");
}

int quant_bwd(const QuantDesc& d, hipStream_t st) {
  QuantP p{};
  if (int e = quant_fill(d, p)) return e;
  if (!p.dq || !p.cvec || !p.dlogits || !p.dvars || !p.prob_sum) return set_error("quant_bwd: null pointer");
  if (p.training && !p.dsoft && !p.dsoft32) return set_error("quant_bwd: dsoft required in training mode");
  long rows = (long)p.R * p.G;
  int grid = (int)std::min<long>((rows + 3) / 4, 2048);
  hipLaunchKernelGGL(quant_kernel<true>, dim3(grid), dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "quant_bwd");
}

// ================================================================================================
// InfoNCE logits.  Row i = (b, j): targets t_0 = y[i], t_k = y[neg_idx[b, j*K + k-1]].
// 16 lanes per target (4 targets in flight per wave), C <= 1024.
// ================================================================================================
struct NceP {
  const bf16* x; const bf16* y; const long long* neg;  // neg: [B, K*M] int64 (the reference's tensor)
  float* logits;                                       // [R, K+1]
  float* xn; float* yn;                                // [R] row norms (clamped at eps), saved by fwd
  const float* dlogits; bf16* dx; bf16* dy;            // bwd outputs bf16 [R, C]
  float* dy_ws; int nsplit;                            // fp32 [R, C] accumulator when rows are split over blocks
  int B, M, K, C; float inv_temp; int cw;
};

__device__ __forceinline__ float sum16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool BWD, int PER>
__global__ __launch_bounds__(256) void nce_kernel(NceP p) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const int C = p.C, K = p.K, M = p.M;
  const long R = (long)p.B * M;
  constexpr float EPS = 1e-8f;
  constexpr int per = PER;  // elements per lane within a 16-lane group (C = 16*PER, PER % 8 == 0)
  for (long row = wave_id; row < R; row += nwaves) {
    const int b = (int)(row / M), j = (int)(row % M);
    // each 16-lane group holds the full x row and the positive y row
    float xs[PER], ys[PER];
    float xx = 0.f;
#pragma unroll
    for (int e8 = 0; e8 < per / 8; ++e8) {
      bf16x8 t = *(const bf16x8*)(p.x + row * C + gl * per + e8 * 8);
      bf16x8 u = *(const bf16x8*)(p.y + row * C + gl * per + e8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { xs[e8 * 8 + e] = bf2f(t[e]); ys[e8 * 8 + e] = bf2f(u[e]); xx += xs[e8 * 8 + e] * xs[e8 * 8 + e]; }
    }
    xx = sum16(xx);
    const float xn = fmaxf(sqrtf(xx), EPS);
    for (int k0 = 0; k0 <= K; k0 += 4) {
      const int k = k0 + grp;
      const bool valid = k <= K;
      long trow = row;
      if (valid && k > 0) trow = (long)p.neg[(long)b * K * M + (long)j * K + (k - 1)];
      float dot = 0.f, tt = 0.f;
      bool same = true;
      float ts[PER];
#pragma unroll
      for (int e8 = 0; e8 < per / 8; ++e8) {
        bf16x8 t = *(const bf16x8*)(p.y + trow * C + gl * per + e8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float tv = bf2f(t[e]);
          ts[e8 * 8 + e] = tv;
          dot += xs[e8 * 8 + e] * tv;
          tt += tv * tv;
          same = same && (tv == ys[e8 * 8 + e]);
        }
      }
      dot = sum16(dot);
      tt = sum16(tt);
      // all-lanes-equal within the 16-lane group
      unsigned long long bal = __ballot(same);
      const bool all_same = ((bal >> (grp * 16)) & 0xFFFFull) == 0xFFFFull;
      const float tn = fmaxf(sqrtf(tt), EPS);
      const float cosv = dot / (xn * tn);
      const bool neg_is_pos = (k > 0) && all_same;
      if (!BWD) {
        if (valid && gl == 0) {
          p.logits[row * (K + 1) + k] = neg_is_pos ? -INFINITY : cosv * p.inv_temp;
          if (k == 0) { p.xn[row] = xn; p.yn[row] = tn; }
        }
      }
    }
  }
}

// Backward.  The forward saved cos (= logits * temp) and the row norms, so the gradient needs no
// dot products:  d cos/dx = t/(|x||t|) - cos x/|x|^2,  d cos/dt = x/(|x||t|) - cos t/|t|^2.
// One 1024-thread block per (channel slice, utterance): negatives never leave their utterance
// (wav2vec2.py:512-514), so dy of the utterance's M rows x cw channels is accumulated in LDS with
// LDS atomics and written once - no global atomics, bf16 results directly.
template <int EPL>  // elements per lane: cw = 64 * EPL
__global__ __launch_bounds__(1024) void nce_bwd_kernel(NceP p) {
  extern __shared__ float dyl[];  // [M][cw]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.y, c0 = blockIdx.x * p.cw;
  const int M = p.M, K = p.K, C = p.C, cw = p.cw;
  const int split = blockIdx.z, nsplit = p.nsplit;   // query rows are dealt over nsplit blocks
  const float temp = 1.f / p.inv_temp;
  for (int i = threadIdx.x; i < M * cw; i += 1024) dyl[i] = 0.f;
  __syncthreads();
  bool act[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) act[e] = (c0 + lane + 64 * e) < C;
  for (int i = wid * nsplit + split; i < M; i += 16 * nsplit) {
    const long row = (long)b * M + i;
    float xs[EPL], gx[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { xs[e] = act[e] ? bf2f(p.x[row * C + c0 + lane + 64 * e]) : 0.f; gx[e] = 0.f; }
    const float xn = p.xn[row];
    const float* lg = p.logits + row * (K + 1);
    const float* dl = p.dlogits + row * (K + 1);
    const long long* ng = p.neg + (long)b * K * M + (long)i * K;
    // phase 1: the (K+1) per-target scalars, lanes in parallel (k = lane, lane+64); entries without
    // gradient (g == 0, neg==pos, k > K) become a no-op on the own row
    int tq[2]; float aq[2], bq[2];
    float sbx = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = lane + 64 * h;
      tq[h] = i; aq[h] = 0.f; bq[h] = 0.f;
      if (k <= K) {
        const float lv = lg[k];
        const float g = (lv == -INFINITY) ? 0.f : dl[k] * p.inv_temp;
        if (g != 0.f) {
          const int t = (k == 0) ? i : (int)(ng[k - 1] - (long long)b * M);
          const float cosv = lv * temp, tn = p.yn[(long)b * M + t];
          tq[h] = t; aq[h] = g / (xn * tn); bq[h] = g * cosv / (tn * tn);
          sbx += g * cosv / (xn * xn);
        }
      }
    }
    sbx = wave_sum(sbx);
    // phase 2: straight-line over targets; scalars broadcast from lane registers, rows of y stream in
    const int KK = K + 1;
#pragma unroll 4
    for (int k = 0; k < KK; ++k) {
      const int h = k >> 6, src = k & 63;
      const int t = __builtin_amdgcn_readlane(h ? tq[1] : tq[0], src);
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h ? aq[1] : aq[0]), src));
      const float bt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h ? bq[1] : bq[0]), src));
      const bf16* yr = p.y + ((long)b * M + t) * C + c0 + lane;
      float* dr = dyl + t * cw + lane;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        if (act[e]) {
          float tv = bf2f(yr[64 * e]);
          gx[e] += a * tv;
          atomicAdd(&dr[64 * e], a * xs[e] - bt * tv);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
      if (act[e]) p.dx[row * C + c0 + lane + 64 * e] = f2bf(gx[e] - sbx * xs[e]);
  }
  __syncthreads();
  if (nsplit == 1) {
    for (int i = threadIdx.x; i < M * cw; i += 1024) {
      int t = i / cw, c = i % cw;
      if (c0 + c < C) p.dy[((long)b * M + t) * C + c0 + c] = f2bf(dyl[i]);
    }
  } else {  // partial sums of this row split: 256-B contiguous fp32 atomics per wave instruction
    for (int i = threadIdx.x; i < M * cw; i += 1024) {
      int t = i / cw, c = i % cw;
      float v = dyl[i];
      if (c0 + c < C && v != 0.f) atomicAdd(&p.dy_ws[((long)b * M + t) * C + c0 + c], v);
    }
  }
}

static int nce_fill(const NceDesc& d, NceP& p) {
  p.x = (const bf16*)d.x; p.y = (const bf16*)d.y; p.neg = (const long long*)d.neg_idx; p.logits = d.logits;
  p.xn = d.xn; p.yn = d.yn;
  p.dlogits = d.dlogits; p.dx = (bf16*)d.dx; p.dy = (bf16*)d.dy; p.B = d.B; p.M = d.M; p.K = d.K; p.C = d.C;
  if (!p.xn || !p.yn) return set_error("infonce: norm buffers xn/yn required");
  if (!p.x || !p.y || !p.neg) return set_error("infonce: null pointer");
  if (p.C != 128 && p.C != 256 && p.C != 512 && p.C != 768) return set_error("infonce: final_dim must be one of 128, 256, 512, 768");
  if (p.B <= 0 || p.M <= 1 || p.K < 0) return set_error("infonce: need B>0, M>1, K>=0");
  if (!(d.temp > 0.f)) return set_error("infonce: logit_temp must be positive");
  p.inv_temp = 1.0f / d.temp;
  return 0;
}

int nce_fwd(const NceDesc& d, hipStream_t st) {
  NceP p{};
  if (int e = nce_fill(d, p)) return e;
  if (!p.logits) return set_error("infonce_fwd: null logits");
  long R = (long)p.B * p.M;
  int grid = (int)std::min<long>((R + 3) / 4, 4096);
  switch (p.C) {
    case 128: hipLaunchKernelGGL((nce_kernel<false, 8>), dim3(grid), dim3(256), 0, st, p); break;
    case 256: hipLaunchKernelGGL((nce_kernel<false, 16>), dim3(grid), dim3(256), 0, st, p); break;
    case 512: hipLaunchKernelGGL((nce_kernel<false, 32>), dim3(grid), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((nce_kernel<false, 48>), dim3(grid), dim3(256), 0, st, p); break;
  }
  return hip_check(hipGetLastError(), "infonce_fwd");
}

// ---------------------------------------------------------------------------------------------
// Dense form of the backward (used when the caller hands over a workspace).  Per utterance the targets of all
// (query i, slot k) pairs are folded into a coefficient matrix A[i][t] = sum of g/(|x_i||y_t|) over the slots of i
// that point at t (negatives are sampled with replacement: ~20 repeats per row), and
//     dX = A . Y - diag(sbx) X          dY = A^T . X - diag(Bv) Y
// are two small matrix products per utterance on the existing TN GEMM (A and A^T are both written, bf16, so each
// product finds its contraction index on the rows).  The scatter form below issues 820 k LDS float atomics
// (~130 cycles each on gfx950): 264 us; this form: a 2-atomics-per-row build + two batched GEMMs + a finish pass.
// ---------------------------------------------------------------------------------------------
struct NceDenseP {
  NceP n;
  bf16* A; bf16* AT;          // [B][Mp][Mp]
  float* Bv; float* sbx;      // [B][Mp]
  float* dxw; float* dyw;     // [B][Mp][C]
  int Mp;
};

// grid (Mp/16, B), 1024 threads: wave w owns query row i = 16*blockIdx.x + w
__global__ __launch_bounds__(1024) void nce_coef_kernel(NceDenseP q) {
  extern __shared__ float rowbuf[];                 // [16][Mp + 1]
  const NceP& p = q.n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y, i0 = blockIdx.x * 16, i = i0 + w;
  const int M = p.M, K = p.K, Mp = q.Mp, pitch = Mp + 1;
  const float temp = 1.f / p.inv_temp;
  for (int j = threadIdx.x; j < 16 * pitch; j += 1024) rowbuf[j] = 0.f;
  __syncthreads();
  if (i < M) {
    const long row = (long)b * M + i;
    const float xn = p.xn[row];
    const float* lg = p.logits + row * (K + 1);
    const float* dl = p.dlogits + row * (K + 1);
    const long long* ng = p.neg + (long)b * K * M + (long)i * K;
    float sbx = 0.f;
    for (int k = lane; k <= K; k += 64) {
      const float lv = lg[k];
      const float g = (lv == -INFINITY) ? 0.f : dl[k] * p.inv_temp;
      if (g != 0.f) {
        const int t = (k == 0) ? i : (int)(ng[k - 1] - (long long)b * M);
        const float cosv = lv * temp, tn = p.yn[(long)b * M + t];
        atomicAdd(&rowbuf[w * pitch + t], g / (xn * tn));          // repeats of t inside this row fold here
        atomicAdd(&q.Bv[(long)b * Mp + t], g * cosv / (tn * tn));
        sbx += g * cosv / (xn * xn);
      }
    }
    sbx = wave_sum(sbx);
    if (lane == 0) q.sbx[(long)b * Mp + i] = sbx;
  }
  __syncthreads();
  // A rows (zero rows past M keep the padded products clean) and the 16-column strip of A^T
  bf16* Ar = q.A + ((long)b * Mp + i) * Mp;
  for (int t = lane; t < Mp; t += 64) Ar[t] = f2bf(rowbuf[w * pitch + t]);
  for (int j = threadIdx.x; j < 16 * Mp; j += 1024) {
    const int w2 = j & 15, t = j >> 4;
    q.AT[((long)b * Mp + t) * Mp + i0 + w2] = f2bf(rowbuf[w2 * pitch + t]);
  }
}

// dx = bf16(dxw - sbx * x), dy = bf16(dyw - Bv * y); one thread per 8 channels
__global__ __launch_bounds__(256) void nce_finish_kernel(NceDenseP q) {
  const NceP& p = q.n;
  const int C8 = p.C >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.B * p.M * C8;
  if (idx >= total) return;
  const int c8 = (int)(idx % C8);
  const long row = idx / C8;
  const int b = (int)(row / p.M), i = (int)(row - (long)b * p.M);
  const float sb = q.sbx[(long)b * q.Mp + i], bv = q.Bv[(long)b * q.Mp + i];
  const float* dxw = q.dxw + ((long)b * q.Mp + i) * p.C + c8 * 8;
  const float* dyw = q.dyw + ((long)b * q.Mp + i) * p.C + c8 * 8;
  const bf16x8 xv = *(const bf16x8*)(p.x + row * p.C + c8 * 8), yv = *(const bf16x8*)(p.y + row * p.C + c8 * 8);
  bf16x8 ox, oy;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ox[e] = f2bf(dxw[e] - sb * bf2f(xv[e]));
    oy[e] = f2bf(dyw[e] - bv * bf2f(yv[e]));
  }
  *(bf16x8*)(p.dx + row * p.C + c8 * 8) = ox;
  *(bf16x8*)(p.dy + row * p.C + c8 * 8) = oy;
}

static int nce_bwd_dense(const NceDesc& d, const NceP& p, hipStream_t st) {
  NceDenseP q{};
  q.n = p;
  const int Mp = (p.M + 63) / 64 * 64;
  q.Mp = Mp;
  const long nA = (long)p.B * Mp * Mp, nW = (long)p.B * Mp * p.C, nV = (long)p.B * Mp;
  char* w = (char*)d.ws;
  q.dxw = (float*)w; q.dyw = q.dxw + nW; q.Bv = q.dyw + nW; q.sbx = q.Bv + nV;     // fp32 part first (one memset)
  q.A = (bf16*)(q.sbx + nV); q.AT = q.A + nA;
  if (int e = hip_check(hipMemsetAsync(q.dxw, 0, sizeof(float) * (2 * nW + nV), st), "memset")) return e;
  const size_t lds = (size_t)16 * (Mp + 1) * sizeof(float);
  hipLaunchKernelGGL(nce_coef_kernel, dim3(Mp / 16, p.B), dim3(1024), lds, st, q);
  if (int e = hip_check(hipGetLastError(), "infonce_bwd coef")) return e;
  GemmDesc g{};
  g.M = Mp; g.N = p.C; g.K = p.M; g.batch = p.B; g.lda = Mp; g.ldb = p.C; g.ldc = p.C; g.alpha = 1.f;
  g.sA = (long)Mp * Mp; g.sB = (long)p.M * p.C; g.sC = (long)Mp * p.C;
  g.a_bytes = (long)Mp * Mp * 2; g.b_bytes = (long)p.M * p.C * 2;
  g.A = q.A; g.B = p.x; g.Cf = q.dyw;                      // dY = A^T X : contraction over query rows i
  if (int e = gemm_tn(g, 0, st)) return e;
  g.A = q.AT; g.B = p.y; g.Cf = q.dxw;                     // dX = A Y   : contraction over target rows t
  if (int e = gemm_tn(g, 0, st)) return e;
  const long total = (long)p.B * p.M * (p.C / 8);
  hipLaunchKernelGGL(nce_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q);
  return hip_check(hipGetLastError(), "infonce_bwd finish");
}

int nce_bwd(const NceDesc& d, hipStream_t st) {
  NceP p{};
  if (int e = nce_fill(d, p)) return e;
  if (!p.dlogits || !p.dx || !p.dy || !p.logits) return set_error("infonce_bwd: null pointer");
  {
    const long Mp = (p.M + 63) / 64 * 64;
    const long need = 4L * p.B * Mp * (2L * p.C + 2) + 2L * 2 * p.B * Mp * Mp;
    static const int dense_env = W2VS_ENV_INT("W2VS_NCE_DENSE", -1);
    bool dense = d.ws && d.ws_bytes >= need && ((uintptr_t)d.ws % 16) == 0 && Mp <= 1024 && (p.C % 8) == 0;
    if (dense_env == 0) dense = false;
    if (dense) return nce_bwd_dense(d, p, st);
  }
  // widest channel slice whose [M][cw] fp32 accumulator fits in LDS (keep 16 KiB headroom)
  int cw = 0;
  for (int c = 64; c >= 64; c -= 64)   // 64-wide slices: C/64 x B blocks keep more CUs busy than wider ones
    if ((long)p.M * c * 4 <= 144 * 1024) { cw = c; break; }
  if (!cw) return set_error("infonce_bwd: too many masked frames per utterance for the LDS accumulator (M > 576)");
  if (cw > p.C) cw = ((p.C + 63) / 64) * 64;
  p.cw = cw;
  const int slices = (p.C + cw - 1) / cw;
  // enough blocks to occupy the chip: split the query rows of an utterance over several blocks whose
  // partial dy are combined with fp32 atomics into dy_ws (then rounded to bf16 once)
  int nsplit = 1;
  if (d.dy_ws) {
    while (nsplit < 16 && slices * p.B * nsplit < 192) nsplit *= 2;
  }
  p.nsplit = nsplit; p.dy_ws = d.dy_ws;
  const long R = (long)p.B * p.M;
  if (nsplit > 1)
    if (int e = hip_check(hipMemsetAsync(p.dy_ws, 0, sizeof(float) * R * p.C, st), "memset")) return e;
  dim3 grid(slices, p.B, nsplit);
  size_t lds = (size_t)p.M * cw * 4;
#define NCE_LAUNCH(E)                                                                                          \
  do {                                                                                                         \
    if (int e = hip_check(hipFuncSetAttribute((const void*)nce_bwd_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)lds), "hipFuncSetAttribute")) return e;                     \
    hipLaunchKernelGGL(nce_bwd_kernel<E>, grid, dim3(1024), lds, st, p);                                        \
  } while (0)
  switch (cw / 64) {
    case 1: NCE_LAUNCH(1); break;
    case 2: NCE_LAUNCH(2); break;
    case 3: NCE_LAUNCH(3); break;
    default: NCE_LAUNCH(4); break;
  }
#undef NCE_LAUNCH
  if (nsplit > 1)
    if (int e = f32_to_bf16(p.dy_ws, p.dy, R * p.C, 1.0f, st)) return e;
  return hip_check(hipGetLastError(), "infonce_bwd");
}

// ================================================================================================
// Cross entropy with target class 0 over rows of fp32 logits (reduction = sum), accuracy
// counters as the criterion computes them, and dlogits = softmax - onehot(0).
// out[0] = loss sum, out[1] = #(argmax==0), out[2] = #(argmax==0 && argmin==0)
// ================================================================================================
// TAIL: the block that finishes last also does the criterion's scalar arithmetic (fs/criterions/wav2vec_criterion.py:80-100),
// which composed from framework ops is ~25 one-element launches per step:
//   loss = ce + w_ppl * ((num_vars - prob_ppl) / num_vars) * sample_size + w_pen * features_pen * sample_size
// and leaves the three accumulators and the ticket at zero again for the next launch.
struct LossTail {
  const float* pen_acc; const float* ppl;       // sum of squared features; {prob_perplexity, code_perplexity}
  float w_ppl, w_pen, num_vars, pen_norm, sample_size;
  float* loss; float* vec;                      // vec[8] = {loss, ce, ppl term, pen term, correct, prob_ppl, code_ppl, features_pen}
  unsigned* ticket;
};
template <bool TAIL>
__global__ __launch_bounds__(256) void ce_kernel(const float* logits, long R, int W, float* out, float* dlogits, LossTail tail) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  float loss_acc = 0.f, max0 = 0.f, both0 = 0.f;
  for (long row = wave_id; row < R; row += nwaves) {
    const float* lp = logits + row * W;
    float mx = -INFINITY, mn = INFINITY;
    int amx = 0x7fffffff, amn = 0x7fffffff;
    for (int v = lane; v < W; v += 64) {
      float x = lp[v];
      if (x > mx) { mx = x; amx = v; }
      if (x < mn) { mn = x; amn = v; }
    }
    wave_argmax(mx, amx);
    float nmn = -mn;
    wave_argmax(nmn, amn);
    float se = 0.f;
    for (int v = lane; v < W; v += 64) se += __expf(lp[v] - mx);
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    if (dlogits)
      for (int v = lane; v < W; v += 64) dlogits[row * W + v] = __expf(lp[v] - lse) - (v == 0 ? 1.f : 0.f);
    if (lane == 0) {
      loss_acc += lse - lp[0];
      if (amx == 0) { max0 += 1.f; if (amn == 0) both0 += 1.f; }
    }
  }
  // one atomic triple per BLOCK: thousands of waves adding into the same three words serialise at the memory
  // side (this tail was 60 of the kernel's 78 us)
  __shared__ float red[4][3];
  if (lane == 0) { red[threadIdx.x >> 6][0] = loss_acc; red[threadIdx.x >> 6][1] = max0; red[threadIdx.x >> 6][2] = both0; }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(&out[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
  if (TAIL) {
    __shared__ int is_last;
    if (threadIdx.x < 3) __threadfence();        // this block's three sums are out before its ticket
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(tail.ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (is_last && threadIdx.x == 0) {
      __threadfence();
      const float ce = atomicExch(&out[0], 0.f), n_max0 = atomicExch(&out[1], 0.f), n_both0 = atomicExch(&out[2], 0.f);
      atomicExch(tail.ticket, 0u);
      const float prob_ppl = tail.ppl[0], code_ppl = tail.ppl[1];
      const float pen = tail.pen_acc[0] * tail.pen_norm;
      const float l1 = tail.w_ppl * ((tail.num_vars - prob_ppl) / tail.num_vars) * tail.sample_size;
      const float l2 = tail.w_pen * pen * tail.sample_size;
      const float loss = (ce + l1) + l2;
      tail.loss[0] = loss;
      tail.vec[0] = loss; tail.vec[1] = ce; tail.vec[2] = l1; tail.vec[3] = l2; tail.vec[4] = n_max0 - n_both0;
      tail.vec[5] = prob_ppl; tail.vec[6] = code_ppl; tail.vec[7] = pen;
    }
  }
}

// dlogits *= g[0];  dsc = {g * c_pen, g * c_ppl}: everything the backward of the scalar arithmetic above hands on
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* g, float* dl, long n, float c_pen, float c_ppl, float* dsc) {
  const float gv = g[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) { dsc[0] = gv * c_pen; dsc[1] = gv * c_ppl; }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dl[i] *= gv;
}

int ce_rows(const float* logits, long R, int W, float* out3, float* dlogits, hipStream_t st) {
  if (!logits || !out3) return set_error("ce_rows: null pointer");
  if (R <= 0 || W <= 0) return set_error("ce_rows: bad shape");
  if (int e = hip_check(hipMemsetAsync(out3, 0, 3 * sizeof(float), st), "memset")) return e;
  int grid = (int)std::min<long>((R + 3) / 4, 256);
  hipLaunchKernelGGL(ce_kernel<false>, dim3(grid), dim3(256), 0, st, logits, R, W, out3, dlogits, LossTail{});
  return hip_check(hipGetLastError(), "ce_rows");
}

int infonce_loss(const InfonceLossDesc& d, hipStream_t st) {
  if (!d.logits || !d.pen_acc || !d.ppl || !d.loss || !d.vec || !d.scratch) return set_error("infonce_loss: null pointer");
  if (d.R <= 0 || d.W <= 0 || d.num_vars <= 0.f) return set_error("infonce_loss: bad shape");
  LossTail t{};
  t.pen_acc = d.pen_acc; t.ppl = d.ppl; t.w_ppl = d.w_ppl; t.w_pen = d.w_pen; t.num_vars = d.num_vars; t.pen_norm = d.pen_norm;
  t.sample_size = d.sample_size; t.loss = d.loss; t.vec = d.vec; t.ticket = (unsigned*)(d.scratch + 3);
  int grid = (int)std::min<long>((d.R + 3) / 4, 256);
  hipLaunchKernelGGL(ce_kernel<true>, dim3(grid), dim3(256), 0, st, d.logits, (long)d.R, d.W, d.scratch, d.dlogits, t);
  return hip_check(hipGetLastError(), "infonce_loss");
}

int infonce_loss_bwd(const float* g, float* dlogits, long n, float c_pen, float c_ppl, float* dsc, hipStream_t st) {
  if (!g || !dlogits || !dsc) return set_error("infonce_loss_bwd: null pointer");
  if (n <= 0) return set_error("infonce_loss_bwd: n must be positive");
  const int grid = (int)std::min<long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid), dim3(256), 0, st, g, dlogits, n, c_pen, c_ppl, dsc);
  return hip_check(hipGetLastError(), "infonce_loss_bwd");
}

// ================================================================================================
// misc: row gather / scatter by index, transpose, fp32 -> bf16, column sums
// ================================================================================================
__global__ void gather_rows_kernel(const bf16* src, const int* idx, bf16* dst, long R, int C, int scatter) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  int cpr = C / 8;
  if (i >= R * cpr) return;
  long row = i / cpr;
  int ch = (int)(i % cpr);
  long s = scatter ? row : idx[row], d = scatter ? idx[row] : row;
  *(u32x4*)(dst + d * C + ch * 8) = *(const u32x4*)(src + s * C + ch * 8);
}
int gather_rows(const void* src, const int* idx, void* dst, long R, int C, int scatter, hipStream_t st) {
  if (!src || !idx || !dst) return set_error("gather_rows: null pointer");
  if (C % 8 || R <= 0) return set_error("gather_rows: C must be a multiple of 8 and R positive");
  long n = R * (C / 8);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const bf16*)src, idx, (bf16*)dst, R, C, scatter);
  return hip_check(hipGetLastError(), "gather_rows");
}

// out[c][r] = in[r][c]  (bf16, 32x32 LDS tiles) ; batched over blockIdx.z
__global__ void transpose_kernel(const bf16* in, bf16* out, int R, int C, long sin, long sout) {
  __shared__ bf16 t[32][33];
  in += (long)blockIdx.z * sin; out += (long)blockIdx.z * sout;
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    int r = r0 + i, c = c0 + threadIdx.x;
    t[i][threadIdx.x] = (r < R && c < C) ? in[(long)r * C + c] : f2bf(0.f);
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < C) out[(long)c * R + r] = t[threadIdx.x][i];
  }
}
int transpose2d(const void* in, void* out, int R, int C, int batch, hipStream_t st) {
  if (!in || !out) return set_error("transpose2d: null pointer");
  if (R <= 0 || C <= 0 || batch <= 0) return set_error("transpose2d: bad shape");
  dim3 grid((C + 31) / 32, (R + 31) / 32, batch), block(32, 8);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, (const bf16*)in, (bf16*)out, R, C, (long)R * C, (long)R * C);
  return hip_check(hipGetLastError(), "transpose2d");
}

// Up to 64 transposes in one launch: the four weight matrices of every encoder layer are turned once per step
// (the dgrad GEMMs want W^T), 48 launches of 5 us each before.  64x64 tiles, 16-byte loads and stores.
struct TrBatch {
  const bf16* in[64]; bf16* out[64];
  int R[64], C[64], tile0[65], tc[64];
  long ldi[64], ldo[64];
  int n;
};
__global__ __launch_bounds__(256) void transpose_multi_kernel(TrBatch b) {
  __shared__ bf16 t[64][72];
  int it = 0;
  while (it + 1 < b.n && (int)blockIdx.x >= b.tile0[it + 1]) ++it;
  const int local = blockIdx.x - b.tile0[it];
  const int R = b.R[it], C = b.C[it];
  const int r0 = (local / b.tc[it]) * 64, c0 = (local % b.tc[it]) * 64;
  const bf16* in = b.in[it];
  bf16* out = b.out[it];
  const long ldi = b.ldi[it], ldo = b.ldo[it];
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 2; ++j) {       // 64 rows x 8 chunks of 8 elements
    const int ch = tid + 256 * j, rr = ch >> 3, cc = (ch & 7) * 8;
    bf16x8 v;
    if (r0 + rr < R && c0 + cc + 8 <= C && (ldi % 8) == 0) v = *(const bf16x8*)(in + (long)(r0 + rr) * ldi + c0 + cc);
    else
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (r0 + rr < R && c0 + cc + e < C) ? in[(long)(r0 + rr) * ldi + c0 + cc + e] : f2bf(0.f);
#pragma unroll
    for (int e = 0; e < 8; ++e) t[rr][cc + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {       // output rows = input columns
    const int ch = tid + 256 * j, oc = ch >> 3, rr = (ch & 7) * 8;
    if (c0 + oc >= C) continue;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = t[rr + e][oc];
    bf16* dst = out + (long)(c0 + oc) * ldo + r0 + rr;
    if (r0 + rr + 8 <= R && (ldo % 8) == 0) *(bf16x8*)dst = v;
    else
#pragma unroll
      for (int e = 0; e < 8; ++e) if (r0 + rr + e < R) dst[e] = v[e];
  }
}
int transpose_multi(const w2vs_transpose_item* items, int n, hipStream_t st) {
  if (!items || n <= 0 || n > 64) return set_error("transpose_multi: need 1..64 items");
  TrBatch b{};
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (!items[i].in || !items[i].out || items[i].R <= 0 || items[i].C <= 0) return set_error("transpose_multi: bad item");
    b.in[i] = (const bf16*)items[i].in; b.out[i] = (bf16*)items[i].out; b.R[i] = items[i].R; b.C[i] = items[i].C;
    b.ldi[i] = items[i].ld_in > 0 ? items[i].ld_in : items[i].C;
    b.ldo[i] = items[i].ld_out > 0 ? items[i].ld_out : items[i].R;
    b.tile0[i] = tiles;
    b.tc[i] = (items[i].C + 63) / 64;
    tiles += b.tc[i] * ((items[i].R + 63) / 64);
  }
  b.tile0[n] = tiles;
  b.n = n;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(tiles), dim3(256), 0, st, b);
  return hip_check(hipGetLastError(), "transpose_multi");
}

__global__ void f32_to_bf16_kernel(const float* in, bf16* out, long n, float scale) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    f32x4 v = *(const f32x4*)(in + i);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e] * scale);
    *(bf16x4*)(out + i) = o;
  } else {
    for (; i < n; ++i) out[i] = f2bf(in[i] * scale);
  }
}
// the same for ranges that do not start on a 16-byte boundary (a gradient-exchange bucket may start anywhere)
__global__ void f32_to_bf16_scalar_kernel(const float* in, bf16* out, long n, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = f2bf(in[i] * scale);
}
int f32_to_bf16(const float* in, void* out, long n, float scale, hipStream_t st) {
  if (!in || !out || n <= 0) return set_error("f32_to_bf16: bad arguments");
  if (((uintptr_t)in & 15) || ((uintptr_t)out & 7))
    hipLaunchKernelGGL(f32_to_bf16_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, (bf16*)out, n, scale);
  else
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, in, (bf16*)out, n, scale);
  return hip_check(hipGetLastError(), "f32_to_bf16");
}
// out = float(in): the way back of a bf16-compressed gradient exchange (trainer.GradExchange, wire_dtype = "bf16")
__global__ void bf16_to_f32_kernel(const bf16* in, float* out, long n, bool vec) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * (vec ? 4 : 1);
  if (vec && i + 3 < n) {
    const bf16x4 v = *(const bf16x4*)(in + i);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = bf2f(v[e]);
    *(f32x4*)(out + i) = o;
  } else if (vec) {
    for (; i < n; ++i) out[i] = bf2f(in[i]);
  } else if (i < n) {
    out[i] = bf2f(in[i]);
  }
}
int bf16_to_f32(const void* in, float* out, long n, hipStream_t st) {
  if (!in || !out || n <= 0) return set_error("bf16_to_f32: bad arguments");
  const bool vec = !(((uintptr_t)out & 15) || ((uintptr_t)in & 7));
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + (vec ? 1023 : 255)) / (vec ? 1024 : 256))), dim3(256), 0, st,
                     (const bf16*)in, out, n, vec);
  return hip_check(hipGetLastError(), "bf16_to_f32");
}

// out = in * keep(seed, i) / (1-p) : standalone dropout (dropout_features), same call for the bwd
__global__ void dropout_kernel(const bf16* in, bf16* out, long n, Drop D) {
  long c = (long)blockIdx.x * 256 + threadIdx.x, i = c * 8;
  if (i >= n) return;
  bf16x8 v = *(const bf16x8*)(in + i), o;
  float m[8];
  drop8(D, (uint32_t)c, m);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(v[e]) * m[e]);
  *(bf16x8*)(out + i) = o;
}
int dropout(const void* in, void* out, long n, float p, uint64_t seed, hipStream_t st) {
  if (!in || !out || n <= 0 || (n % 8)) return set_error("dropout: n must be a positive multiple of 8");
  if (p < 0.f || p >= 1.f) return set_error("dropout: p must be in [0,1)");
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, (const bf16*)in, (bf16*)out, n,
                     make_drop(p, seed));
  return hip_check(hipGetLastError(), "dropout");
}

// out = gate > 0 ? x : 0.  ReLU of the CAAT joiner's FFN (rain/layers/attention_transducer.py:772, activation_fn "relu"):
// forward with gate == x, backward with x = d(out) and gate = the forward's output.
__global__ void relu_gate_kernel(const bf16* x, const bf16* gate, bf16* out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  const bf16x8 v = *(const bf16x8*)(x + i), g = *(const bf16x8*)(gate + i);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = bf2f(g[e]) > 0.f ? v[e] : f2bf(0.f);
  *(bf16x8*)(out + i) = o;
}
int relu_gate(const void* x, const void* gate, void* out, long n, hipStream_t st) {
  if (!x || !gate || !out || n <= 0 || (n % 8)) return set_error("relu_gate: n must be a positive multiple of 8");
  hipLaunchKernelGGL(relu_gate_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, (const bf16*)x, (const bf16*)gate,
                     (bf16*)out, n);
  return hip_check(hipGetLastError(), "relu_gate");
}

// Fused Adam over flat arrays (fs/optim/adam.py:205-229: decoupled weight decay, bias-corrected
// step size) with the fp32 master / bf16 working copy split of fs/optim/fp16_optimizer.py:205-218.
// g is the fp32 gradient arena; grad_scale folds the 1/sample_size (and clip) factor in.
template <int NT_>
__global__ void adam_kernel(float* p32, bf16* p16, float* m, float* v, const float* g, long n, float lr_wd, float step_size,
                            float b1, float b2, float eps, const float* scale_dev, float scale_host) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float gs = scale_dev ? scale_host * scale_dev[0] : scale_host;
  // a gradient scale of exactly 0 is w2vs_clip_scale's mark for a non-finite gradient norm: the update is SKIPPED - master,
  // moments and the bf16 image keep their values, no weight decay (NaN * 0 would poison all four).  The reference raises
  // FloatingPointError before optimizer.step in that case (fs/trainer.py:791-793); the host raises it from the flag later.
  if (gs == 0.f) return;
  f32x4 pv, mv, vv, gv;
  if (NT_) {   // every stream is touched once per update: no reuse to keep in L2
    pv = __builtin_nontemporal_load((f32x4*)(p32 + i)); mv = __builtin_nontemporal_load((f32x4*)(m + i));
    vv = __builtin_nontemporal_load((f32x4*)(v + i)); gv = __builtin_nontemporal_load((const f32x4*)(g + i));
  } else {
    pv = *(f32x4*)(p32 + i); mv = *(f32x4*)(m + i); vv = *(f32x4*)(v + i); gv = *(const f32x4*)(g + i);
  }
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float gr = gv[e] * gs;
    mv[e] = b1 * mv[e] + (1.f - b1) * gr;
    vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
    float pe = pv[e] - lr_wd * pv[e];
    pe -= step_size * mv[e] / (sqrtf(vv[e]) + eps);
    pv[e] = pe;
    o[e] = f2bf(pe);
  }
  if (NT_) {
    __builtin_nontemporal_store(pv, (f32x4*)(p32 + i)); __builtin_nontemporal_store(mv, (f32x4*)(m + i));
    __builtin_nontemporal_store(vv, (f32x4*)(v + i));
  } else {
    *(f32x4*)(p32 + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv;
  }
  *(bf16x4*)(p16 + i) = o;          // the working copy IS read again (next forward): normal policy
}
int adam_step(float* p32, void* p16, float* m, float* v, const float* g, long n, float lr, float b1, float b2, float eps,
              float wd, int step, const float* scale_dev, float scale_host, hipStream_t st) {
  if (!p32 || !p16 || !m || !v || !g || n <= 0 || (n % 4)) return set_error("adam_step: bad arguments (n must be a multiple of 4)");
  if (step < 1) return set_error("adam_step: step counts from 1");
  double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  float step_size = (float)(lr * sqrt(bc2) / bc1);
  // nontemporal loads / stores of the four fp32 streams (each touched once per update: 2.7 GB that would otherwise displace
  // what the next forward wants from L2 / the Infinity Cache): kernel 474 -> 425 us (6.0 -> 6.7 TB/s), step 8.60 -> 8.46 ms -
  // more than the kernel's own gain.  The same hint on the saved gelu' / LayerNorm inputs / gradient-norm read measured
  // nothing (EXPERIMENTS.md).  W2VS_ADAM_NT=0 is the A/B.
  static const int nt_env = W2VS_ENV_INT("W2VS_ADAM_NT", 1);
  if (nt_env)
    hipLaunchKernelGGL(adam_kernel<1>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, p32, (bf16*)p16, m, v, g, n,
                       wd * lr, step_size, b1, b2, eps, scale_dev, scale_host);
  else
    hipLaunchKernelGGL(adam_kernel<0>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, p32, (bf16*)p16, m, v, g, n,
                       wd * lr, step_size, b1, b2, eps, scale_dev, scale_host);
  return hip_check(hipGetLastError(), "adam_step");
}

// out[0] += sum x^2 (fp32): gradient norm for clip_grad_norm_ (fs/utils.py:341-386).
// DETERMINISTIC (round 4): every block leaves its partial sum in a scratch row; a one-block launch adds the partials in
// index order.  With float atomics from 1 024 blocks the norm differed in its last bit from run to run - and so
// did the clip coefficient, and with it the whole Adam update, between two data-parallel ranks holding the SAME summed
// gradient (tests/test_a_dist_gpu.py::test_two_ranks_on_one_gpu...): the reference's replicas stay bit-identical.
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, long n, float* part) {
  float s = 0.f;
  const long n4 = n >> 2, stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = *(const f32x4*)(x + 4 * i);
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float t = x[(n4 << 2) + threadIdx.x]; s += t * t; }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// second launch (a "last block done" tail inside the first kernel needs a device-scope release per block: 96 us against 68):
// one block adds the partials in index order
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* part, int nparts, float* out) {
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
  a = wave_sum(a);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}
// scratch of the kernels above: 16 regions per device (launches in flight on different streams do not share one), zeroed
// synchronously when made
struct SumsqScratch { float* part = nullptr; unsigned* cnt = nullptr; unsigned next = 0; };
static SumsqScratch* sumsq_scratch() {
  static std::mutex mu;
  static SumsqScratch per_dev[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  SumsqScratch& sc = per_dev[dev];
  if (!sc.part) {
    void* p = nullptr;
    const size_t bytes = 16 * (SUMSQ_BLOCKS + 64) * sizeof(float);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    sc.part = (float*)p;
    sc.cnt = (unsigned*)p + 16 * SUMSQ_BLOCKS;
  }
  return &sc;
}
int sumsq(const float* x, long n, float* out, hipStream_t st) {
  if (!x || !out || n <= 0 || ((uintptr_t)x & 15)) return set_error("sumsq: bad arguments (x must be 16-byte aligned)");
  SumsqScratch* sc = sumsq_scratch();
  if (!sc) return set_error("sumsq: scratch allocation failed");
  const unsigned r = sc->next++ & 15;
  float* part = sc->part + r * SUMSQ_BLOCKS;
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, st, x, n, part);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, part, SUMSQ_BLOCKS, out);
  return hip_check(hipGetLastError(), "sumsq");
}

// clip_grad_norm_ without a host read (fs/utils.py:341-386 after fs/trainer.py:769-774's multiply_grads(1/sample_size)):
//   inv = scale_host * (scale_dev ? *scale_dev : 1)          the 1 / sample_size factor
//   gnorm = sqrt(sumsq) * inv ;  coef = clip > 0 ? min(1, clip / (gnorm + 1e-6)) : 1
//   out = {inv * coef (what Adam multiplies the arena with), gnorm, non-finite flag}; non-finite norm -> scale 0
__global__ void clip_scale_kernel(float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, float* bad_acc,
                                  int consume) {
  const float inv = scale_dev ? scale_host * scale_dev[0] : scale_host;
  const float gnorm = sqrtf(sumsq[0]) * inv;
  const bool ok = isfinite(gnorm);
  float coef = 1.f;
  if (clip > 0.f) coef = fminf(1.f, clip / (gnorm + 1e-6f));
  out3[0] = ok ? inv * coef : 0.f;
  out3[1] = gnorm;
  out3[2] = ok ? 0.f : 1.f;
  if (consume) sumsq[0] = 0.f;                   // ready for the next update's w2vs_sumsq
  if (bad_acc && !ok) bad_acc[0] += 1.f;         // sticky count of skipped updates
}
int clip_scale(const float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, hipStream_t st) {
  if (!sumsq || !out3) return set_error("clip_scale: null pointer");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, st, const_cast<float*>(sumsq), scale_dev, scale_host, clip, out3,
                     (float*)nullptr, 0);
  return hip_check(hipGetLastError(), "clip_scale");
}
int clip_scale_acc(float* sumsq, const float* scale_dev, float scale_host, float clip, float* out3, float* bad_acc, hipStream_t st) {
  if (!sumsq || !out3) return set_error("clip_scale_acc: null pointer");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, st, sumsq, scale_dev, scale_host, clip, out3, bad_acc, 1);
  return hip_check(hipGetLastError(), "clip_scale_acc");
}

// out[n] += sum_m in[m][n]  (bf16 in, fp32 atomics out): bias gradients
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* in, float* out, long M, int N, long ld, int rows_per_block) {
  __shared__ float red[8][33][8];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;  // 32 column chunks x 8 row lanes
  const int col = (blockIdx.x * 32 + cx) * 8;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N)
    for (long r = r0 + ry; r < r1; r += 8) {
      bf16x8 v = *(const bf16x8*)(in + r * ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ry][cx][e] = acc[e];
  __syncthreads();
  if (ry == 0 && col < N) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += red[k][cx][e];
      atomicAdd(&out[col + e], s);
    }
  }
}
int colsum(const void* in, float* out, long M, int N, long ld, hipStream_t st) {
  if (!in || !out) return set_error("colsum: null pointer");
  if (N % 8 || ld % 8 || M <= 0) return set_error("colsum: N and ld must be multiples of 8");
  int rpb = 256;
  dim3 grid((N / 8 + 31) / 32, (unsigned)((M + rpb - 1) / rpb));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, (const bf16*)in, out, M, N, ld, rpb);
  return hip_check(hipGetLastError(), "colsum");
}

}  // namespace w2vs
