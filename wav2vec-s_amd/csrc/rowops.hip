// Row-wise (channel-last) HBM-bound kernels: one 64-lane wave per row, 16-B vector access,
// fp32 statistics via wavefront shuffles.  Every tensor here is [rows, C] bf16 with C in
// {512, 768, 1024} (C % 8 == 0, C <= 1024): a lane owns chunks {lane, lane+64} of 8 channels.
//
//   conv0_ln_gelu      a1 layer 0   (wav2vec2.py:733-743)   waveform -> [B, L0, 512]
//   feat_ln            a3 + a4      (wav2vec2.py:554-558)   features_pen + LayerNorm(512)
//   enc_prologue       a6 a7 a8 a10 (wav2vec2.py:446, wav2vec_S.py:355-388, 465-484)
//   add_dropout_ln     a11 glue     (wav2vec2.py:965-976)
#include "common.h"
#include "w2vs_internal.h"

namespace w2vs {

constexpr float LN_EPS = 1e-5f;
constexpr int ROWS_PER_BLOCK = 4;  // 256 threads = 4 waves = 4 rows in flight per block

struct Row {  // up to 1024 channels: 2 chunks of 8 per lane
  float v[2][8];
};

__device__ __forceinline__ int nchunks(int C) { return C >> 3; }

__device__ __forceinline__ void load_row(const bf16* p, int C, int lane, Row& r) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int ch = lane + 64 * h;
    if (ch < nchunks(C)) {
      bf16x8 t = *(const bf16x8*)(p + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[h][e] = bf2f(t[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[h][e] = 0.f;
    }
  }
}
struct RawRow {  // the same two chunks as undecoded bf16 bits (8 registers instead of 16)
  u32x4 v[2];
};
__device__ __forceinline__ void load_raw(const bf16* p, int C, int lane, RawRow& r) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int ch = lane + 64 * h;
    r.v[h] = u32x4{0u, 0u, 0u, 0u};
    if (ch < nchunks(C)) r.v[h] = *(const u32x4*)(p + ch * 8);
  }
}
__device__ __forceinline__ void unpack_row(const RawRow& r, Row& o) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o.v[h][2 * j] = __uint_as_float(r.v[h][j] << 16);
      o.v[h][2 * j + 1] = __uint_as_float(r.v[h][j] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ void store_row(bf16* p, int C, int lane, const Row& r) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int ch = lane + 64 * h;
    if (ch < nchunks(C)) {
      bf16x8 t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = f2bf(r.v[h][e]);
      *(bf16x8*)(p + ch * 8) = t;
    }
  }
}
__device__ __forceinline__ void row_stats(const Row& r, int C, int lane, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += r.v[h][e];  // chunks beyond C hold zeros
  mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (lane + 64 * h < nchunks(C)) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { float d = r.v[h][e] - mean; q += d * d; }
    }
  }
  rstd = rsqrtf(wave_sum(q) / (float)C + LN_EPS);
}

// dropout: element index = row_id*C + channel; identical in fwd and bwd
// multiply a row by its dropout keep/scale factors; chunk index = row * (C/8) + chunk
__device__ __forceinline__ void drop_row(const Drop& D, long row, int C, int lane, Row& x) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float m[8];
    drop8(D, (uint32_t)(row * (C >> 3) + lane + 64 * h), m);
#pragma unroll
    for (int e = 0; e < 8; ++e) x.v[h][e] *= m[e];
  }
}

// =====================================================================================
// conv layer 0: Conv1d(1->C, k=10, s=5, no pad) -> LayerNorm(C) over channels -> GELU
// =====================================================================================
struct Conv0P {
  const bf16* wave; const bf16* w; const bf16* cbias; const bf16* lnw; const bf16* lnb;
  bf16* y; float* mean; float* rstd;
  const bf16* dy; float* dw; float* dcbias; float* dlnw; float* dlnb;
  int B, L, L0, C, k, s;
};

template <bool BWD>
__global__ __launch_bounds__(256) void conv0_kernel(Conv0P p) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int C = p.C, K = p.k;
  const bool act = lane < nchunks(C);  // C <= 512 for the conv stack: one chunk per lane
  // per-lane weights for its 8 channels (k <= 16 taps)
  float w[8][10];
  float cb[8], g[8], bb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = lane * 8 + e;
#pragma unroll
    for (int j = 0; j < 10; ++j) w[e][j] = (act && j < K) ? bf2f(p.w[c * K + j]) : 0.f;
    cb[e] = (act && p.cbias) ? bf2f(p.cbias[c]) : 0.f;
    g[e] = act ? bf2f(p.lnw[c]) : 0.f;
    bb[e] = act ? bf2f(p.lnb[c]) : 0.f;
  }
  float dw[8][10], dg[8], db[8], dcb[8];
  if (BWD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dg[e] = db[e] = dcb[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 10; ++j) dw[e][j] = 0.f;
    }
  }
  const long rows = (long)p.B * p.L0;
  for (long row = wave_id; row < rows; row += nwaves) {
    const int b = (int)(row / p.L0), t = (int)(row % p.L0);
    const bf16* xw = p.wave + (long)b * p.L + (long)t * p.s;
    float x[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) x[j] = (j < K) ? bf2f(xw[j]) : 0.f;
    float c0[8];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = cb[e];
#pragma unroll
      for (int j = 0; j < 10; ++j) a = fmaf(w[e][j], x[j], a);
      c0[e] = act ? a : 0.f;
      s += c0[e];
    }
    float mean, rstd;
    if (!BWD) {
      mean = wave_sum(s) / (float)C;
      float q = 0.f;
      if (act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = c0[e] - mean; q += d * d; }
      }
      rstd = rsqrtf(wave_sum(q) / (float)C + LN_EPS);
      if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
      if (act) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(gelu_exact((c0[e] - mean) * rstd * g[e] + bb[e]));
        *(bf16x8*)(p.y + row * C + lane * 8) = o;
      }
    } else {
      mean = p.mean[row]; rstd = p.rstd[row];
      float dxh[8], xh[8];
      float s1 = 0.f, s2 = 0.f;
      bf16x8 dyv;
      if (act) dyv = *(const bf16x8*)(p.dy + row * C + lane * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[e] = (c0[e] - mean) * rstd;
        float z = xh[e] * g[e] + bb[e];
        float dz = act ? bf2f(dyv[e]) * gelu_grad(z) : 0.f;
        dg[e] += dz * xh[e];
        db[e] += dz;
        dxh[e] = dz * g[e];
        s1 += dxh[e];
        s2 += dxh[e] * xh[e];
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float dc = act ? rstd * (dxh[e] - s1 - xh[e] * s2) : 0.f;
        dcb[e] += dc;
#pragma unroll
        for (int j = 0; j < 10; ++j) dw[e][j] = fmaf(dc, x[j], dw[e][j]);
      }
    }
  }
  if (BWD) {
    // block-level reduction in LDS first: one global atomic per value per BLOCK, not per wave
    __shared__ float red[512 * 13];
    for (int i = threadIdx.x; i < C * 13; i += 256) red[i] = 0.f;
    __syncthreads();
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int c = lane * 8 + e;
        atomicAdd(&red[c], dg[e]);
        atomicAdd(&red[C + c], db[e]);
        atomicAdd(&red[2 * C + c], dcb[e]);
#pragma unroll
        for (int j = 0; j < 10; ++j)
          if (j < K) atomicAdd(&red[3 * C + c * K + j], dw[e][j]);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
      atomicAdd(&p.dlnw[i], red[i]);
      atomicAdd(&p.dlnb[i], red[C + i]);
      if (p.dcbias) atomicAdd(&p.dcbias[i], red[2 * C + i]);
    }
    for (int i = threadIdx.x; i < C * K; i += 256) atomicAdd(&p.dw[i], red[3 * C + i]);
  }
}

int conv0_fwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, void* y,
              float* mean, float* rstd, int B, int L, int C, int k, int s, hipStream_t st) {
  if (!wave || !w || !lnw || !lnb || !y || !mean || !rstd) return set_error("conv0_fwd: null pointer");
  if (C % 8 || C > 512 || k > 10 || k < 1 || s < 1 || L < k) return set_error("conv0_fwd: need C%8==0, C<=512, k<=10, L>=k");
  if ((long)B * ((L - k) / s + 1) >= 0x7FFFFFFFL) return set_error("conv0_fwd: more than 2^31 output frames");
  if (conv0_mfma_ok(C, k)) return conv0_mfma_fwd(wave, w, cbias, lnw, lnb, y, mean, rstd, B, L, k, s, st);
  Conv0P p{};
  p.wave = (const bf16*)wave; p.w = (const bf16*)w; p.cbias = (const bf16*)cbias; p.lnw = (const bf16*)lnw; p.lnb = (const bf16*)lnb;
  p.y = (bf16*)y; p.mean = mean; p.rstd = rstd; p.B = B; p.L = L; p.L0 = (L - k) / s + 1; p.C = C; p.k = k; p.s = s;
  long rows = (long)B * p.L0;
  int grid = (int)std::min<long>((rows + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(conv0_kernel<false>, dim3(grid), dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "conv0_fwd");
}

int conv0_bwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, const float* mean,
              const float* rstd, const void* dy, float* dw, float* dcbias, float* dlnw, float* dlnb, int B, int L, int C,
              int k, int s, hipStream_t st) {
  if (!wave || !w || !lnw || !lnb || !dy || !mean || !rstd || !dw || !dlnw || !dlnb) return set_error("conv0_bwd: null pointer");
  if (C % 8 || C > 512 || k > 10 || k < 1 || s < 1 || L < k) return set_error("conv0_bwd: need C%8==0, C<=512, k<=10, L>=k");
  if ((long)B * ((L - k) / s + 1) >= 0x7FFFFFFFL) return set_error("conv0_bwd: more than 2^31 output frames");
  if (conv0_mfma_ok(C, k)) return conv0_mfma_bwd(wave, w, cbias, lnw, lnb, mean, rstd, dy, dw, dcbias, dlnw, dlnb, B, L, k, s, st);
  Conv0P p{};
  p.wave = (const bf16*)wave; p.w = (const bf16*)w; p.cbias = (const bf16*)cbias; p.lnw = (const bf16*)lnw; p.lnb = (const bf16*)lnb;
  p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd); p.dy = (const bf16*)dy;
  p.dw = dw; p.dcbias = dcbias; p.dlnw = dlnw; p.dlnb = dlnb;
  p.B = B; p.L = L; p.L0 = (L - k) / s + 1; p.C = C; p.k = k; p.s = s;
  long rows = (long)B * p.L0;
  int grid = (int)std::min<long>((rows + 3) / 4, 256);  // one block per CU: long waves, few final atomics
  hipLaunchKernelGGL(conv0_kernel<true>, dim3(grid), dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "conv0_bwd");
}

// =====================================================================================
// conv layer 0, extractor_mode="default": Conv1d(1->C,k,s) -> Fp32GroupNorm(C groups = C channels)
// -> GELU (fs/models/wav2vec/wav2vec2.py:744-750).  The statistics run over TIME per (b, c), so
// each direction is two passes over the (tiny) waveform; the conv output is recomputed, never stored.
//   FWD_STATS : stat[b][c] += {sum, sumsq} of the conv output
//   FWD_APPLY : y = gelu((c0 - mean) * rstd * gamma + beta)
//   BWD_STATS : s[b][c] += {sum dxhat, sum dxhat*xhat};  dgamma, dbeta
//   BWD_APPLY : dconv = rstd * (dxhat - s1/L0 - xhat * s2/L0);  dW, dbias
// =====================================================================================
enum { GN_FWD_STATS = 0, GN_FWD_APPLY = 1, GN_BWD_STATS = 2, GN_BWD_APPLY = 3 };
struct Conv0GnP {
  const bf16* wave; const bf16* w; const bf16* cbias; const bf16* g; const bf16* b;
  bf16* y; float* stat;          // [B][C][2] sum, sumsq
  const bf16* dy; float* bstat;  // [B][C][2] s1, s2
  float* dw; float* dcbias; float* dg; float* db;
  int B, L, L0, C, k, s, rows_per_block;
};

template <int MODE>
__global__ __launch_bounds__(256) void conv0_gn_kernel(Conv0GnP p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int C = p.C, K = p.k, bi = blockIdx.y;
  const bool act = lane < nchunks(C);
  float w[8][10], cb[8], g[8], bb[8], mean[8], rstd[8], m1[8], m2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int c = lane * 8 + e;
#pragma unroll
    for (int j = 0; j < 10; ++j) w[e][j] = (act && j < K) ? bf2f(p.w[c * K + j]) : 0.f;
    cb[e] = (act && p.cbias) ? bf2f(p.cbias[c]) : 0.f;
    g[e] = act ? bf2f(p.g[c]) : 0.f;
    bb[e] = act ? bf2f(p.b[c]) : 0.f;
    mean[e] = 0.f; rstd[e] = 0.f; m1[e] = 0.f; m2[e] = 0.f;
    if (MODE != GN_FWD_STATS && act) {
      float sm = p.stat[((long)bi * C + c) * 2], sq = p.stat[((long)bi * C + c) * 2 + 1];
      mean[e] = sm / (float)p.L0;
      rstd[e] = rsqrtf(fmaxf(sq / (float)p.L0 - mean[e] * mean[e], 0.f) + LN_EPS);
    }
    if (MODE == GN_BWD_APPLY && act) {
      m1[e] = p.bstat[((long)bi * C + c) * 2] / (float)p.L0;
      m2[e] = p.bstat[((long)bi * C + c) * 2 + 1] / (float)p.L0;
    }
  }
  float a0[8], a1[8], dwv[8][10];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a0[e] = a1[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j) dwv[e][j] = 0.f;
  }
  float a2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int t0 = blockIdx.x * p.rows_per_block, t1 = min(p.L0, t0 + p.rows_per_block);
  for (int t = t0 + wid; t < t1; t += 4) {
    const bf16* xw = p.wave + (long)bi * p.L + (long)t * p.s;
    float x[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) x[j] = (j < K) ? bf2f(xw[j]) : 0.f;
    const long row = (long)bi * p.L0 + t;
    bf16x8 dyv;
    if ((MODE == GN_BWD_STATS || MODE == GN_BWD_APPLY) && act) dyv = *(const bf16x8*)(p.dy + row * C + lane * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float c0 = cb[e];
#pragma unroll
      for (int j = 0; j < 10; ++j) c0 = fmaf(w[e][j], x[j], c0);
      if (MODE == GN_FWD_STATS) { a0[e] += c0; a1[e] += c0 * c0; continue; }
      const float xh = (c0 - mean[e]) * rstd[e];
      const float z = xh * g[e] + bb[e];
      if (MODE == GN_FWD_APPLY) { o[e] = f2bf(gelu_exact(z)); continue; }
      const float dz = act ? bf2f(dyv[e]) * gelu_grad(z) : 0.f;
      const float dxh = dz * g[e];
      if (MODE == GN_BWD_STATS) { a0[e] += dxh; a1[e] += dxh * xh; a2[e] += dz * xh; a3[e] += dz; continue; }
      const float dc = rstd[e] * (dxh - m1[e] - xh * m2[e]);
      a0[e] += dc;
#pragma unroll
      for (int j = 0; j < 10; ++j) dwv[e][j] = fmaf(dc, x[j], dwv[e][j]);
    }
    if (MODE == GN_FWD_APPLY && act) *(bf16x8*)(p.y + row * C + lane * 8) = o;
  }
  if (MODE == GN_FWD_APPLY) return;
  // block reduction through LDS, then one global atomic per value per block
  __shared__ float red[512 * 14];
  const int nval = (MODE == GN_BWD_APPLY) ? (1 + K) : (MODE == GN_BWD_STATS ? 4 : 2);
  for (int i = threadIdx.x; i < C * nval; i += 256) red[i] = 0.f;
  __syncthreads();
  if (act) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = lane * 8 + e;
      atomicAdd(&red[c * nval + 0], a0[e]);
      if (MODE != GN_BWD_APPLY) atomicAdd(&red[c * nval + 1], a1[e]);
      if (MODE == GN_BWD_STATS) { atomicAdd(&red[c * nval + 2], a2[e]); atomicAdd(&red[c * nval + 3], a3[e]); }
      if (MODE == GN_BWD_APPLY) {
#pragma unroll
        for (int j = 0; j < 10; ++j)
          if (j < K) atomicAdd(&red[c * nval + 1 + j], dwv[e][j]);
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    if (MODE == GN_FWD_STATS) {
      atomicAdd(&p.stat[((long)bi * C + c) * 2], red[c * nval]);
      atomicAdd(&p.stat[((long)bi * C + c) * 2 + 1], red[c * nval + 1]);
    } else if (MODE == GN_BWD_STATS) {
      atomicAdd(&p.bstat[((long)bi * C + c) * 2], red[c * nval]);
      atomicAdd(&p.bstat[((long)bi * C + c) * 2 + 1], red[c * nval + 1]);
      atomicAdd(&p.dg[c], red[c * nval + 2]);
      atomicAdd(&p.db[c], red[c * nval + 3]);
    } else {
      if (p.dcbias) atomicAdd(&p.dcbias[c], red[c * nval]);
      for (int j = 0; j < K; ++j) atomicAdd(&p.dw[c * K + j], red[c * nval + 1 + j]);
    }
  }
}

static int conv0_gn_fill(Conv0GnP& p, const void* wave, const void* w, const void* cbias, const void* g, const void* b,
                         float* stat, int B, int L, int C, int k, int s) {
  if (!wave || !w || !g || !b || !stat) return set_error("conv0_gn: null pointer");
  if (C % 8 || C > 512 || k > 10 || k < 1 || s < 1 || L < k) return set_error("conv0_gn: need C%8==0, C<=512, k<=10, L>=k");
  p.wave = (const bf16*)wave; p.w = (const bf16*)w; p.cbias = (const bf16*)cbias; p.g = (const bf16*)g; p.b = (const bf16*)b;
  p.stat = stat; p.B = B; p.L = L; p.L0 = (L - k) / s + 1; p.C = C; p.k = k; p.s = s;
  p.rows_per_block = std::max(64, (p.L0 * B + 1023) / 1024);
  return 0;
}

int conv0_gn_fwd(const void* wave, const void* w, const void* cbias, const void* g, const void* b, void* y, float* stat,
                 int B, int L, int C, int k, int s, hipStream_t st) {
  Conv0GnP p{};
  if (int e = conv0_gn_fill(p, wave, w, cbias, g, b, stat, B, L, C, k, s)) return e;
  if (!y) return set_error("conv0_gn_fwd: null output");
  p.y = (bf16*)y;
  if (int e = hip_check(hipMemsetAsync(stat, 0, sizeof(float) * 2 * B * C, st), "memset")) return e;
  dim3 grid((p.L0 + p.rows_per_block - 1) / p.rows_per_block, B);
  hipLaunchKernelGGL(conv0_gn_kernel<GN_FWD_STATS>, grid, dim3(256), 0, st, p);
  hipLaunchKernelGGL(conv0_gn_kernel<GN_FWD_APPLY>, grid, dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "conv0_gn_fwd");
}

int conv0_gn_bwd(const void* wave, const void* w, const void* cbias, const void* g, const void* b, const float* stat,
                 const void* dy, float* bstat, float* dw, float* dcbias, float* dg, float* db, int B, int L, int C, int k,
                 int s, hipStream_t st) {
  Conv0GnP p{};
  if (int e = conv0_gn_fill(p, wave, w, cbias, g, b, const_cast<float*>(stat), B, L, C, k, s)) return e;
  if (!dy || !bstat || !dw || !dg || !db) return set_error("conv0_gn_bwd: null pointer");
  p.dy = (const bf16*)dy; p.bstat = bstat; p.dw = dw; p.dcbias = dcbias; p.dg = dg; p.db = db;
  if (int e = hip_check(hipMemsetAsync(bstat, 0, sizeof(float) * 2 * B * C, st), "memset")) return e;
  dim3 grid((p.L0 + p.rows_per_block - 1) / p.rows_per_block, B);
  hipLaunchKernelGGL(conv0_gn_kernel<GN_BWD_STATS>, grid, dim3(256), 0, st, p);
  hipLaunchKernelGGL(conv0_gn_kernel<GN_BWD_APPLY>, grid, dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "conv0_gn_bwd");
}

// =====================================================================================
// Generic LayerNorm forward / backward over rows (used by feat_ln, conv LN layers of the
// large model, add_dropout_ln).  Modes are compile-time flags.
// =====================================================================================
struct LnP {
  const bf16* x;      // main input  [rows, C]
  const bf16* res;    // residual (added) or nullptr
  const bf16* g; const bf16* b;
  bf16* y;            // LN output (may be null when only the sum is wanted)
  bf16* sum_out;      // x(+dropout) + res, bf16 (pre-LN residual stream) or nullptr
  float* mean; float* rstd;
  float* sumsq;       // features_pen accumulator (sum of x^2), or nullptr
  // dropout on x before the add
  float p_drop; uint64_t seed;
  // backward
  const bf16* dy; const bf16* dsum; const bf16* aux;  // aux: pre-activation for the fused gelu'
  bf16* dx; bf16* dres; float* dg; float* db;
  float out_scale;    // multiplies dx (GradMultiply)
  float pen_coef;     // d(loss)/d(sum x^2) : adds 2*x*pen_coef to dx
  const float* pen_dev;  // optional device multiplier of pen_coef
  int gelu;           // fwd: apply GELU after LN; bwd: dy is grad wrt GELU output
  long rows; int C;
};

__global__ __launch_bounds__(256) void ln_fwd_kernel(LnP p) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int C = p.C;
  const Drop D = make_drop(p.p_drop, p.seed);
  Row g, b;
  load_row(p.g, C, lane, g);
  load_row(p.b, C, lane, b);
  float sq = 0.f;
  // two or three rows per wave (ln_fwd caps the grid): the raw bf16 bits of the NEXT row (and of its residual) are in flight
  // while this one is reduced, and gamma / beta are fetched once per wave instead of once per row
  RawRow nx, nres;
  if (wave_id < p.rows) {
    load_raw(p.x + (long)wave_id * C, C, lane, nx);
    if (p.res) load_raw(p.res + (long)wave_id * C, C, lane, nres);
  }
  for (long row = wave_id; row < p.rows; row += nwaves) {
    Row x, rr;
    unpack_row(nx, x);
    if (p.res) unpack_row(nres, rr);
    if (row + nwaves < p.rows) {
      load_raw(p.x + (row + nwaves) * C, C, lane, nx);
      if (p.res) load_raw(p.res + (row + nwaves) * C, C, lane, nres);
    }
    if (p.sumsq) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) sq += x.v[h][e] * x.v[h][e];
    }
    if (p.p_drop > 0.f) {
      drop_row(D, row, C, lane, x);
    }
    if (p.res) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) x.v[h][e] += rr.v[h][e];
    }
    if (p.sum_out) {
      store_row(p.sum_out + row * C, C, lane, x);
      // the LN below sees the value that was stored (bf16), as a two-kernel pipeline would
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) x.v[h][e] = bf2f(f2bf(x.v[h][e]));
    }
    if (p.y) {
      float mean, rstd;
      row_stats(x, C, lane, mean, rstd);
      if (lane == 0 && p.mean) { p.mean[row] = mean; p.rstd[row] = rstd; }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float z = (x.v[h][e] - mean) * rstd * g.v[h][e] + b.v[h][e];
          x.v[h][e] = p.gelu ? gelu_exact(z) : z;
        }
      store_row(p.y + row * C, C, lane, x);
    }
  }
  if (p.sumsq) {   // one same-address atomic per block, not per wave
    __shared__ float sred[4];
    sq = wave_sum(sq);
    if (lane == 0) sred[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(p.sumsq, (sred[0] + sred[1]) + (sred[2] + sred[3]));
  }
}

// The encoder layers' form (dropout + residual + LayerNorm, no GELU, no penalty sum) with ONE row per wave and few enough
// registers for seven waves per SIMD: every row of a 6544-row activation is in flight at once, where the general kernel above
// (124 registers, four waves per SIMD) takes two rounds of dependent load -> reduce -> store chains.
__global__ __launch_bounds__(256, 7) void ln_fwd_lean_kernel(LnP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int C = p.C, nch = C >> 3;
  const bool two = lane + 64 < nch, one = lane < nch;
  const bf16* xp = p.x + row * C;
  u32x4 rx[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}, rr[2] = {rx[0], rx[0]};
  if (one) rx[0] = *(const u32x4*)(xp + lane * 8);
  if (two) rx[1] = *(const u32x4*)(xp + (lane + 64) * 8);
  if (p.res) {
    const bf16* rp = p.res + row * C;
    if (one) rr[0] = *(const u32x4*)(rp + lane * 8);
    if (two) rr[1] = *(const u32x4*)(rp + (lane + 64) * 8);
  }
  const Drop D = make_drop(p.p_drop, p.seed);
  float v[2][8];
  float sum = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float m[8];
    if (p.p_drop > 0.f) drop8(D, (uint32_t)(row * nch + lane + 64 * h), m);
    u32x4 packed;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a0 = __uint_as_float(rx[h][j] << 16), a1 = __uint_as_float(rx[h][j] & 0xFFFF0000u);
      if (p.p_drop > 0.f) { a0 *= m[2 * j]; a1 *= m[2 * j + 1]; }
      a0 += __uint_as_float(rr[h][j] << 16); a1 += __uint_as_float(rr[h][j] & 0xFFFF0000u);
      if (p.sum_out) {          // the LN sees the value that is stored (bf16), as a two-kernel pipeline would
        bf16x2 t; t[0] = f2bf(a0); t[1] = f2bf(a1);
        packed[j] = __builtin_bit_cast(uint32_t, t);
        a0 = __uint_as_float(packed[j] << 16); a1 = __uint_as_float(packed[j] & 0xFFFF0000u);
      }
      v[h][2 * j] = a0; v[h][2 * j + 1] = a1;
      sum += a0 + a1;
    }
    if (p.sum_out && (h == 0 ? one : two)) *(u32x4*)(p.sum_out + row * C + (lane + 64 * h) * 8) = packed;
  }
  const float mean = wave_sum(sum) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (h == 0 ? one : two) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[h][e] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + LN_EPS);
  if (lane == 0 && p.mean) { p.mean[row] = mean; p.rstd[row] = rstd; }
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (h == 0 ? one : two) {
      const int ch = lane + 64 * h;
      const u32x4 g = *(const u32x4*)(p.g + ch * 8), b = *(const u32x4*)(p.b + ch * 8);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z0 = (v[h][2 * j] - mean) * rstd * __uint_as_float(g[j] << 16) + __uint_as_float(b[j] << 16);
        const float z1 = (v[h][2 * j + 1] - mean) * rstd * __uint_as_float(g[j] & 0xFFFF0000u) + __uint_as_float(b[j] & 0xFFFF0000u);
        bf16x2 t; t[0] = f2bf(z0); t[1] = f2bf(z1);
        o[j] = __builtin_bit_cast(uint32_t, t);
      }
      *(u32x4*)(p.y + row * C + ch * 8) = o;
    }
}

// dx = LNbwd(dy) [+ dsum];   dres = dx;   d(x) = dx * dropmask/(1-p)
// `x` here must be the LN *input* (sum).  When gelu: dy is wrt gelu(LN(x)).
// One wave per row; NW waves per block.  FULL = false is the lean encoder-layer form (no fused GELU, no
// penalty / GradMultiply / aux chain): half the registers, so 16 waves per CU cover the load latency.
// dgamma / dbeta: per-lane register partials -> one LDS reduction per block -> either a [grid][2C]
// partial slab (`part`, reduced by ln_bwd_reduce_kernel) or, without a workspace, atomics per block.
template <bool FULL, int NW>
__global__ __launch_bounds__(NW * 64, FULL ? 2 : 3) void ln_bwd_kernel(LnP p, float* part) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * NW + (threadIdx.x >> 6), nwaves = gridDim.x * NW;
  const int C = p.C;
  const Drop D = make_drop(p.p_drop, p.seed);
  Row g, b, dg, db;
  load_row(p.g, C, lane, g);
  if (FULL) load_row(p.b, C, lane, b);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) dg.v[h][e] = db.v[h][e] = 0.f;
  // software pipeline over rows: the raw bf16 bits of the NEXT row are in flight while this one is reduced
  RawRow nx, ndy;
  float nmean = 0.f, nrstd = 0.f;
  if (wave_id < p.rows) {
    load_raw(p.x + (long)wave_id * C, C, lane, nx);
    if (p.dy) load_raw(p.dy + (long)wave_id * C, C, lane, ndy);
    nmean = p.mean[wave_id]; nrstd = p.rstd[wave_id];
  }
  for (long row = wave_id; row < p.rows; row += nwaves) {
    Row x, dy;
    unpack_row(nx, x);
    if (p.dy) unpack_row(ndy, dy);
    const float mean = nmean, rstd = nrstd;
    if (row + nwaves < p.rows) {
      load_raw(p.x + (row + nwaves) * C, C, lane, nx);
      if (p.dy) load_raw(p.dy + (row + nwaves) * C, C, lane, ndy);
      nmean = p.mean[row + nwaves]; nrstd = p.rstd[row + nwaves];
    }
    float s1 = 0.f, s2 = 0.f;
    Row dxh;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xh = (x.v[h][e] - mean) * rstd;
        float d = p.dy ? dy.v[h][e] : 0.f;
        if (FULL && p.gelu) d *= gelu_grad(xh * g.v[h][e] + b.v[h][e]);
        if (lane + 64 * h >= nchunks(C)) { d = 0.f; xh = 0.f; }
        dg.v[h][e] += d * xh;
        db.v[h][e] += d;
        float t = d * g.v[h][e];
        dxh.v[h][e] = t;
        s1 += t;
        s2 += t * xh;
        x.v[h][e] = xh;
      }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    Row dx;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) dx.v[h][e] = rstd * (dxh.v[h][e] - s1 - x.v[h][e] * s2);
    if (p.dsum) {
      Row ds;
      load_row(p.dsum + row * C, C, lane, ds);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) dx.v[h][e] += ds.v[h][e];
    }
    if (p.dres) store_row(p.dres + row * C, C, lane, dx);
    if (p.dx) {
      if (FULL) {
        const float pen = p.pen_dev ? p.pen_coef * p.pen_dev[0] : p.pen_coef;
        if (pen != 0.f || p.out_scale != 1.f) {
          // x.v currently holds xhat: rebuild the raw input for the penalty term
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float raw = x.v[h][e] / rstd + mean;
              dx.v[h][e] = (dx.v[h][e] + 2.f * raw * pen) * p.out_scale;
            }
        }
      }
      if (p.p_drop > 0.f) {
        drop_row(D, row, C, lane, dx);
      }
      if (FULL && p.aux) {  // chain through the producing layer's GELU: dx *= gelu'(pre)
        Row a;
        load_row(p.aux + row * C, C, lane, a);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) dx.v[h][e] *= gelu_grad(a.v[h][e]);
      }
      store_row(p.dx + row * C, C, lane, dx);
    }
  }
  // block reduction without LDS atomics (ds_add_f32 costs ~130 cycles per wave-instruction on gfx950: measured
  // 25 of this kernel's 35 us): every wave stores its partials to its own slab red[w][kind][e][chunk] (row pitch
  // 136 floats: conflict-free for the lane-per-chunk stores and for the column-order reads), then each thread
  // sums its columns over the NW slabs.
  constexpr int RP = 136, SLAB = 2 * 8 * RP;
  __shared__ float red[NW * SLAB];
  {
    float* mine = red + (threadIdx.x >> 6) * SLAB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ch = lane + 64 * h;
      if (ch < nchunks(C)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          mine[e * RP + ch] = dg.v[h][e];
          mine[8 * RP + e * RP + ch] = db.v[h][e];
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += NW * 64) {
    const int kind = i >= C, col = kind ? i - C : i;
    const int off = kind * 8 * RP + (col & 7) * RP + (col >> 3);
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[w * SLAB + off];
    if (part) part[(long)blockIdx.x * 2 * C + i] = sum;
    else atomicAdd(kind ? &p.db[col] : &p.dg[col], sum);
  }
}

// dgamma[c] += sum_g part[g][c], dbeta[c] += sum_g part[g][C + c].  grid (ceil(2C/64), RY); a block is
// 64 columns x 4 row lanes; every thread sums a strided subset of the G partial rows.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* part, int G, int C, float* dg, float* db) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx, C2 = 2 * C;
  float s = 0.f;
  if (col < C2) {
    const int step = 4 * gridDim.y;
    int gi = blockIdx.y * 4 + ty;
    for (; gi + 3 * step < G; gi += 4 * step) {
      float a0 = part[(long)gi * C2 + col], a1 = part[(long)(gi + step) * C2 + col];
      float a2 = part[(long)(gi + 2 * step) * C2 + col], a3 = part[(long)(gi + 3 * step) * C2 + col];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; gi < G; gi += step) s += part[(long)gi * C2 + col];
  }
  __shared__ float red[4][64];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < C2) {
    s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    if (col < C) atomicAdd(&dg[col], s);
    else atomicAdd(&db[col - C], s);
  }
}

// the same sums for up to eight LayerNorms in one launch (blockIdx.z picks the norm): the encoder layers leave their partial slabs
// in place and the grouped weight-gradient call of a layer pair reduces all of them together (round 4: 4 launches -> 1)
struct LnRedMany { LnPartial r[8]; int n, C; };
__global__ __launch_bounds__(256) void ln_bwd_reduce_many_kernel(LnRedMany m) {
  const LnPartial r = m.r[blockIdx.z];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx, C = m.C, C2 = 2 * C, G = r.G;
  const float* part = r.part;
  float s = 0.f;
  if (col < C2) {
    const int step = 4 * gridDim.y;
    int gi = blockIdx.y * 4 + ty;
    for (; gi + 3 * step < G; gi += 4 * step) {
      float a0 = part[(long)gi * C2 + col], a1 = part[(long)(gi + step) * C2 + col];
      float a2 = part[(long)(gi + 2 * step) * C2 + col], a3 = part[(long)(gi + 3 * step) * C2 + col];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; gi < G; gi += step) s += part[(long)gi * C2 + col];
  }
  __shared__ float red[4][64];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < C2) {
    s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    if (col < C) atomicAdd(&r.dg[col], s);
    else atomicAdd(&r.db[col - C], s);
  }
}

int ln_reduce_many(const LnPartial* r, int n, int C, hipStream_t st) {
  if (n < 1 || n > 8) return set_error("ln_reduce_many: 1..8 norms");
  LnRedMany m{};
  int gmax = 0;
  for (int i = 0; i < n; ++i) {
    if (!r[i].part || !r[i].dg || !r[i].db || r[i].G < 1) return set_error("ln_reduce_many: null pointer");
    m.r[i] = r[i];
    gmax = std::max(gmax, r[i].G);
  }
  m.n = n; m.C = C;
  const int ry = std::max(1, std::min(16, gmax / 16));
  hipLaunchKernelGGL(ln_bwd_reduce_many_kernel, dim3((2 * C + 63) / 64, ry, n), dim3(256), 0, st, m);
  return hip_check(hipGetLastError(), "ln_reduce_many");
}

// the grid ln_bwd chooses for `rows` rows when it is given a partial slab of ws_bytes (shared with the deferred reduction)
int ln_bwd_grid(long rows, int C, int64_t ws_bytes) {
  constexpr int NW = 4;
  int grid = (int)std::min<long>((rows + NW - 1) / NW, rows > 65536 ? 2048 : 768);
  return (int)std::min<long>(grid, ws_bytes / ((long)sizeof(float) * 2 * C));
}

static int ln_check(const LnP& p, const char* who) {
  if (p.C % 8 || p.C > 1024 || p.C < 8) return set_error("layernorm: need C % 8 == 0 and C <= 1024");
  if (p.rows <= 0) return set_error("layernorm: rows must be positive");
  (void)who;
  return 0;
}

int ln_fwd(const LnFwdDesc& d, hipStream_t st) {
  LnP p{};
  p.x = (const bf16*)d.x; p.res = (const bf16*)d.res; p.g = (const bf16*)d.gamma; p.b = (const bf16*)d.beta;
  p.y = (bf16*)d.y; p.sum_out = (bf16*)d.sum_out; p.mean = d.mean; p.rstd = d.rstd; p.sumsq = d.sumsq;
  p.p_drop = d.p_drop; p.seed = d.seed; p.gelu = d.gelu; p.rows = d.rows; p.C = d.C;
  if (!p.x || !p.g || !p.b) return set_error("ln_fwd: null pointer");
  if (p.y && !p.mean) return set_error("ln_fwd: mean/rstd buffers required");
  if (int e = ln_check(p, "ln_fwd")) return e;
  static const int cap = W2VS_ENV_INT("W2VS_LN_FWD_GRID", 256 * 4);   // ~1.6 rows per wave at the encoder size: measured best of 512 / 768 / 1024 / 2048
  static const int lean_env = W2VS_ENV_INT("W2VS_LN_LEAN", 1);
  if (lean_env && p.y && !p.sumsq && !p.gelu && p.rows >= 1024 && p.rows < (1L << 30)) {
    hipLaunchKernelGGL(ln_fwd_lean_kernel, dim3((unsigned)((p.rows + 3) / 4)), dim3(256), 0, st, p);
    return hip_check(hipGetLastError(), "ln_fwd");
  }
  int grid = (int)std::min<long>((p.rows + 3) / 4, cap);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid), dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "ln_fwd");
}

int ln_bwd(const LnBwdDesc& d, hipStream_t st, bool leave_partials) {
  LnP p{};
  p.x = (const bf16*)d.x; p.g = (const bf16*)d.gamma; p.b = (const bf16*)d.beta; p.mean = const_cast<float*>(d.mean);
  p.rstd = const_cast<float*>(d.rstd); p.dy = (const bf16*)d.dy; p.dsum = (const bf16*)d.dsum; p.aux = (const bf16*)d.aux;
  p.dx = (bf16*)d.dx; p.dres = (bf16*)d.dres; p.dg = d.dgamma; p.db = d.dbeta; p.p_drop = d.p_drop; p.seed = d.seed;
  p.out_scale = d.out_scale; p.pen_coef = d.pen_coef; p.pen_dev = d.pen_coef_dev; p.gelu = d.gelu; p.rows = d.rows; p.C = d.C;
  if (!p.x || !p.g || !p.b || !p.mean || !p.rstd || !p.dg || !p.db) return set_error("ln_bwd: null pointer");
  if (int e = ln_check(p, "ln_bwd")) return e;
  constexpr int NW = 4;
  const bool full = p.gelu || p.aux || p.pen_coef != 0.f || p.pen_dev || p.out_scale != 1.f;
  int grid = (int)std::min<long>((p.rows + NW - 1) / NW, p.rows > 65536 ? 2048 : 768);
  // partial slab [grid][2C] fp32 in the caller's workspace; shrink the grid to what the workspace holds
  float* part = nullptr;
  if (d.ws && d.ws_bytes >= ln_min_slab_bytes(p.C)) {
    grid = ln_bwd_grid(p.rows, p.C, d.ws_bytes);
    part = (float*)d.ws;
  } else {
    grid = std::min(grid, 256);   // no workspace: per-block atomics, keep their number down
  }
  if (leave_partials && !part) return set_error("ln_bwd: a deferred reduction needs a partial slab");
  static const int atomic_env = W2VS_ENV_INT("W2VS_LN_BWD_ATOMIC", 0);   // A/B: N > 0 = atomics from N blocks
  if (atomic_env > 0 && !leave_partials) { part = nullptr; grid = std::min<long>((p.rows + NW - 1) / NW, atomic_env); }
  if (full) hipLaunchKernelGGL((ln_bwd_kernel<true, NW>), dim3(grid), dim3(NW * 64), 0, st, p, part);
  else hipLaunchKernelGGL((ln_bwd_kernel<false, NW>), dim3(grid), dim3(NW * 64), 0, st, p, part);
  if (part && !leave_partials) {
    const int ry = std::max(1, std::min(16, grid / 16));
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * p.C + 63) / 64, ry), dim3(256), 0, st, part, grid, p.C, p.dg, p.db);
  }
  return hip_check(hipGetLastError(), "ln_bwd");
}

// =====================================================================================
// Encoder prologue: mask fill + sinusoidal position + LayerNorm + dropout + pad frame +
// right-context copies, written straight into the [B, N, C] token layout the encoder uses.
// =====================================================================================
struct EncProP {
  const bf16* x;          // [B, T, C] projected features (before dropout_input)
  const uint8_t* mask;    // [B, T] 1 = replace by mask_emb ; may be null
  const uint8_t* pad;     // [B, T] 1 = padded frame ; may be null
  const int* pos;         // [B, T] row of the sinusoid table (host: 1 + cumsum of non-pad, 1 for pad)
  const bf16* mask_emb; const float* pos_table;  // [n_pos, C] fp32
  const bf16* g; const bf16* b;
  bf16* out;              // [B, N, C]
  float* mean; float* rstd;  // [B, T]
  const int* src;         // [N] source frame of every token row (0..Tp-1)
  float p_in; uint64_t seed_in; float p_enc; uint64_t seed_enc;
  int apply_ln;           // 0 for the pre-LN (large) variant
  int B, T, Tp, N, C;
  // backward
  const bf16* dout; bf16* dx; float* dmask_emb; float* dg; float* db;
  const int* copy_start; const int* copy_list;  // CSR: copies of frame t are copy_list[copy_start[t]..copy_start[t+1])
};

__global__ __launch_bounds__(256) void enc_prologue_fwd_kernel(EncProP p) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int C = p.C;
  const Drop Din = make_drop(p.p_in, p.seed_in), Denc = make_drop(p.p_enc, p.seed_enc);
  Row g, b, me;
  load_row(p.g, C, lane, g);
  load_row(p.b, C, lane, b);
  load_row(p.mask_emb, C, lane, me);
  const long rows = (long)p.B * p.N;
  for (long row = wave_id; row < rows; row += nwaves) {
    const int bi = (int)(row / p.N), n = (int)(row % p.N);
    const int t = p.src[n];
    Row x;
    if (t >= p.T) {  // the zero frame appended by pad_to_multiple (after the LN, wav2vec_S.py:375-377)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) x.v[h][e] = 0.f;
      store_row(p.out + row * C, C, lane, x);
      continue;
    }
    const long fr = (long)bi * p.T + t;
    const bool is_pad = p.pad && p.pad[fr];
    const bool is_mask = p.mask && p.mask[fr];
    load_row(p.x + fr * C, C, lane, x);
    const float* pt = p.pos_table + (long)p.pos[fr] * C;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ch = lane + 64 * h;
      if (ch >= nchunks(C)) continue;
      float mi[8];
      if (p.p_in > 0.f) drop8(Din, (uint32_t)(fr * (C >> 3) + ch), mi);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = x.v[h][e];
        if (p.p_in > 0.f) v = bf2f(f2bf(v * mi[e]));
        if (is_mask) v = me.v[h][e];
        if (is_pad) v = 0.f;
        x.v[h][e] = bf2f(f2bf(v + pt[ch * 8 + e]));
      }
    }
    if (p.apply_ln) {
      float mean, rstd;
      row_stats(x, C, lane, mean, rstd);
      if (lane == 0 && n < p.T) { p.mean[fr] = mean; p.rstd[fr] = rstd; }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) x.v[h][e] = bf2f(f2bf((x.v[h][e] - mean) * rstd * g.v[h][e] + b.v[h][e]));
    }
    if (p.p_enc > 0.f) {
      drop_row(Denc, fr, C, lane, x);
    }
    store_row(p.out + row * C, C, lane, x);
  }
}

// one wave per source frame (b, t): gathers the gradient of the frame and of all its copies.  EPB_NW waves per block: the
// number of BLOCKS is what the closing atomics cost (2 304 per block), the number of WAVES what covers a frame's dependent
// loads (copy list -> gradient rows -> source row -> position row: ~5 us per frame for a wave alone on its SIMD)
constexpr int EPB_NW = 8;
__global__ __launch_bounds__(EPB_NW * 64) void enc_prologue_bwd_kernel(EncProP p) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * EPB_NW + (threadIdx.x >> 6), nwaves = gridDim.x * EPB_NW;
  const int C = p.C;
  const Drop Din = make_drop(p.p_in, p.seed_in), Denc = make_drop(p.p_enc, p.seed_enc);
  Row g, me, dgm, dbt, dme;
  load_row(p.g, C, lane, g);
  load_row(p.mask_emb, C, lane, me);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) dgm.v[h][e] = dbt.v[h][e] = dme.v[h][e] = 0.f;
  const long rows = (long)p.B * p.T;
  for (long fr = wave_id; fr < rows; fr += nwaves) {
    const int bi = (int)(fr / p.T), t = (int)(fr % p.T);
    const bool is_pad = p.pad && p.pad[fr];
    const bool is_mask = p.mask && p.mask[fr];
    Row d;
    load_row(p.dout + ((long)bi * p.N + t) * C, C, lane, d);
    for (int ci = p.copy_start[t]; ci < p.copy_start[t + 1]; ++ci) {
      Row dc;
      load_row(p.dout + ((long)bi * p.N + p.copy_list[ci]) * C, C, lane, dc);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) d.v[h][e] += dc.v[h][e];
    }
    if (p.p_enc > 0.f) {
      drop_row(Denc, fr, C, lane, d);
    }
    Row dx = d;
    if (p.apply_ln) {
      // rebuild the LN input exactly as the forward did
      Row x;
      load_row(p.x + fr * C, C, lane, x);
      const float* pt = p.pos_table + (long)p.pos[fr] * C;
      const float mean = p.mean[fr], rstd = p.rstd[fr];
      float s1 = 0.f, s2 = 0.f;
      Row dxh;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int ch = lane + 64 * h;
        float mi[8];
        if (p.p_in > 0.f) drop8(Din, (uint32_t)(fr * (C >> 3) + ch), mi);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = x.v[h][e], xh = 0.f, t1 = 0.f;
          if (ch < nchunks(C)) {
            if (p.p_in > 0.f) v = bf2f(f2bf(v * mi[e]));
            if (is_mask) v = me.v[h][e];
            if (is_pad) v = 0.f;
            v = bf2f(f2bf(v + pt[ch * 8 + e]));
            xh = (v - mean) * rstd;
            dgm.v[h][e] += d.v[h][e] * xh;
            dbt.v[h][e] += d.v[h][e];
            t1 = d.v[h][e] * g.v[h][e];
          }
          x.v[h][e] = xh;
          dxh.v[h][e] = t1;
          s1 += t1;
          s2 += t1 * xh;
        }
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) dx.v[h][e] = rstd * (dxh.v[h][e] - s1 - x.v[h][e] * s2);
    }
    // through the mask fill / padding zero / dropout_input
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ch = lane + 64 * h;
      float mi[8];
      if (p.p_in > 0.f) drop8(Din, (uint32_t)(fr * (C >> 3) + ch), mi);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = dx.v[h][e];
        if (is_pad) v = 0.f;
        if (is_mask) { dme.v[h][e] += v; v = 0.f; }
        if (p.p_in > 0.f && ch < nchunks(C)) v *= mi[e];
        dx.v[h][e] = v;
      }
    }
    store_row(p.dx + fr * C, C, lane, dx);
  }
  // per-wave slabs red[w][kind][e][chunk] (pitch 136), folded by column afterwards - see ln_bwd_kernel
  constexpr int RP = 136, KIND = 8 * RP, SLAB = 3 * KIND;
  __shared__ float red[EPB_NW * SLAB];
  {
    float* mine = red + (threadIdx.x >> 6) * SLAB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ch = lane + 64 * h;
      if (ch < nchunks(C)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          mine[e * RP + ch] = dgm.v[h][e];
          mine[KIND + e * RP + ch] = dbt.v[h][e];
          mine[2 * KIND + e * RP + ch] = dme.v[h][e];
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += EPB_NW * 64) {
    const int kind = i / C, col = i - kind * C;
    if (kind < 2 && !p.apply_ln) continue;
    const int off = kind * KIND + (col & 7) * RP + (col >> 3);
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < EPB_NW; ++w) sum += red[w * SLAB + off];
    atomicAdd(kind == 0 ? &p.dg[col] : (kind == 1 ? &p.db[col] : &p.dmask_emb[col]), sum);
  }
}

static int encpro_fill(const EncPrologueDesc& d, EncProP& p) {
  p.x = (const bf16*)d.x; p.mask = d.mask; p.pad = d.pad; p.pos = d.pos; p.mask_emb = (const bf16*)d.mask_emb;
  p.pos_table = d.pos_table; p.g = (const bf16*)d.gamma; p.b = (const bf16*)d.beta; p.out = (bf16*)d.out;
  p.mean = d.mean; p.rstd = d.rstd; p.src = d.src; p.p_in = d.p_in; p.seed_in = d.seed_in; p.p_enc = d.p_enc;
  p.seed_enc = d.seed_enc; p.apply_ln = d.apply_ln; p.B = d.B; p.T = d.T; p.Tp = d.Tp; p.N = d.N; p.C = d.C;
  p.dout = (const bf16*)d.dout; p.dx = (bf16*)d.dx; p.dmask_emb = d.dmask_emb; p.dg = d.dgamma; p.db = d.dbeta;
  p.copy_start = d.copy_start; p.copy_list = d.copy_list;
  if (!p.x || !p.pos || !p.mask_emb || !p.pos_table || !p.g || !p.b || !p.src || !p.mean || !p.rstd)
    return set_error("enc_prologue: null pointer");
  if (p.C % 8 || p.C > 1024) return set_error("enc_prologue: need C % 8 == 0 and C <= 1024");
  if (p.B <= 0 || p.T <= 0 || p.Tp < p.T || p.N < p.Tp) return set_error("enc_prologue: bad B/T/Tp/N");
  return 0;
}

int enc_prologue_fwd(const EncPrologueDesc& d, hipStream_t st) {
  EncProP p{};
  if (int e = encpro_fill(d, p)) return e;
  if (!p.out) return set_error("enc_prologue_fwd: null output");
  long rows = (long)p.B * p.N;
  int grid = (int)std::min<long>((rows + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(enc_prologue_fwd_kernel, dim3(grid), dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "enc_prologue_fwd");
}

int enc_prologue_bwd(const EncPrologueDesc& d, hipStream_t st) {
  EncProP p{};
  if (int e = encpro_fill(d, p)) return e;
  if (!p.dout || !p.dx || !p.dmask_emb || !p.dg || !p.db || !p.copy_start || !p.copy_list)
    return set_error("enc_prologue_bwd: null pointer");
  long rows = (long)p.B * p.T;
  int grid = (int)std::min<long>((rows + EPB_NW - 1) / EPB_NW, 256);
  hipLaunchKernelGGL(enc_prologue_bwd_kernel, dim3(grid), dim3(EPB_NW * 64), 0, st, p);
  return hip_check(hipGetLastError(), "enc_prologue_bwd");
}

}  // namespace w2vs
