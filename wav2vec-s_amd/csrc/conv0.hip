// conv layer 0 on the matrix cores (gfx950): Conv1d(1 -> 512, k <= 10, stride s, no padding) -> LayerNorm(512)
// -> GELU, forward and backward (fs/models/wav2vec/wav2vec2.py:733-743, extractor_mode="layer_norm").
//
// A wave owns 16 consecutive output frames ("rows") at a time.  The conv is a [16 x 16] . [16 x 512] product
// (K = 16: the k taps, one all-ones column that carries the conv bias, zero padding) = 32 MFMA 16x16x16 tiles.
// Tile t, accumulator lane (n = lane & 15, rg = lane >> 4) holds rows rg*4..rg*4+3 of channel ch_of(t, n), so a
// lane owns 4 x 8 consecutive channels of 4 rows:
//   * LayerNorm statistics are an in-lane sum over the 32 tiles + a 16-lane butterfly,
//   * y / dy move as 16-byte pieces, 256 contiguous bytes of a row per instruction and row group,
//   * gamma/beta and their gradients are 32 registers per lane,
//   * the backward's dconv (4 rows of one channel, bf16) is ALREADY the A fragment of the weight-gradient
//     MFMA  dW[ch][tap] += dconv[rows][ch]^T . x[rows][tap]  - no transpose, no LDS round trip.
// The previous VALU form spent 0.34 / 0.80 ms per step (fwd / bwd) in per-row dependent chains
// (10 scalar taps -> 80 FMAs -> two 64-lane reductions, one row per wave at a time).
#include <algorithm>
#include "common.h"
#include "w2vs_internal.h"

namespace w2vs {

namespace {

constexpr int CC = 512, NT = 32;   // channels, 16-channel MFMA tiles
// Channel of column n (= lane & 15) of MFMA tile t.  Round 5: tile t = 8u + e takes channel u*128 + n*8 + e, so for a fixed u a
// lane's eight tiles are 8 CONSECUTIVE channels (one 16-byte piece) and the 16 lanes of a row group cover 256 contiguous
// bytes of the row: every y store / dy load instruction moves whole 128-byte lines.  (Rounds 1-4 gave a lane 32 consecutive
// channels, n*32 + t: the same instruction then touched 16 pieces of 16 bytes at a 64-byte stride per row - four times the
// lines through the address path for the same bytes; the statistics and the MFMAs do not care which channel sits where.)
__device__ __forceinline__ constexpr int ch_of(int t, int n) { return (t >> 3) * 128 + n * 8 + (t & 7); }
// MAP 0: the layout above; MAP 1: the rounds 1-4 layout (a lane owns 32 consecutive channels) - kept for the A/B of the backward
template <int MAP> __device__ __forceinline__ constexpr int ch_map(int t, int n) { return MAP == 0 ? ch_of(t, n) : n * 32 + t; }
constexpr int BIAS_TAP = 10;       // K index of the all-ones column (needs k <= 10)

struct Conv0M {
  const bf16* wave; const bf16* w; const bf16* cbias; const bf16* lnw; const bf16* lnb;
  bf16* y; float* mean; float* rstd;
  const bf16* dy; float* dw; float* dcbias; float* dlnw; float* dlnb;
  int B, L, L0, k, s;
  long rows;
};

__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ short bf_bits(float v) {
  bf16 h = f2bf(v);
  return __builtin_bit_cast(short, h);
}
__device__ __forceinline__ float reduce16(float v) {  // sum over the 16 lanes that share lane >> 4
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
// ---- gelu' by table (backward) ------------------------------------------------------------------------------------
// The backward was VALU-bound on the erf (~25 of its ~49 vector instructions per element): gelu' comes from an LDS table built
// per workgroup with the formula the other kernels evaluate (gelu_grad), indexed by the fp32 argument rounded to MB mantissa
// bits: 24 binades |z| in [2^-20, 2^4) x 2^MB mantissas x 2 signs, fp32 entries.  The argument's rounding (2^-(MB+2)
// relative) is the only approximation; below the range the clamped entry is within 5e-7 of the true value, above it gelu' is
// exactly 0 / 1 (the clamped entries).  Same box, 8 x 175 000 samples: 271.5 -> 247 us.  The FORWARD keeps the erf: the same
// table for gelu (bf16 entries, MB = 9) measured 159 us against 151 - 64 random 2-byte LDS reads per wave-instruction cost
// more than the ~20 VALU instructions they replace when nothing else in the kernel waits on the LDS.
constexpr int GT_E0 = 127 - 20, GT_NE = 24;
template <int MB> struct GT {
  static constexpr int LO = GT_E0 << MB, HI = ((GT_E0 + GT_NE) << MB) - 1, N = GT_NE << MB;
  static __device__ __forceinline__ float arg(int i) {            // the argument entry i stands for
    const uint32_t mag = (uint32_t)(LO + (i >= N ? i - N : i));
    return __uint_as_float((mag << (23 - MB)) | (i >= N ? 0x80000000u : 0u));
  }
  // fp32 -> (sign | exponent | MB mantissa bits), round to nearest (ties up; a carry moves into the exponent as it should)
  static __device__ __forceinline__ uint32_t key(float z) { return (__float_as_uint(z) + (1u << (22 - MB))) >> (23 - MB); }
  static __device__ __forceinline__ int index(uint32_t r) {
    const int a = (int)(r & ((1u << (8 + MB)) - 1u));
    const int t = min(max(a, LO), HI);                               // v_med3
    return t - LO + (int)(r >> (8 + MB)) * N;
  }
};
constexpr int MB_BWD = 8;                     // 48 KiB of fp32 entries

// waveform offset of output frame `row` (flattened b*L0 + t), or -1 past the end
__device__ __forceinline__ long frame_base(const Conv0M& p, long row) {
  if (row >= p.rows) return -1;
  const uint32_t b = (uint32_t)row / (uint32_t)p.L0, t = (uint32_t)row - b * (uint32_t)p.L0;   // rows < 2^31 (host check)
  return (long)b * p.L + (long)t * p.s;
}
// W fragments, B operand of the conv MFMA: lane (n = lane&15, kg = lane>>4) of tile t holds
// W[ch_of(t, n)][kg*4 .. kg*4+3]; K index BIAS_TAP carries the conv bias.
template <int MAP = 0>
__device__ __forceinline__ void build_w_frags(const Conv0M& p, s16x4* wl, int tid) {
  for (int idx = tid; idx < NT * 64; idx += 256) {
    const int t = idx >> 6, l = idx & 63, n = l & 15, kg = l >> 4, ch = ch_map<MAP>(t, n);
    s16x4 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tap = kg * 4 + i;
      float v = 0.f;
      if (tap < p.k) v = bf2f(p.w[ch * p.k + tap]);
      else if (tap == BIAS_TAP && p.cbias) v = bf2f(p.cbias[ch]);
      f[i] = bf_bits(v);
    }
    wl[idx] = f;
  }
}
// A operand of the conv MFMA: lane (m = lane&15 -> row, kg) holds x[row][kg*4 .. kg*4+3] (+ the ones column)
__device__ __forceinline__ s16x4 load_xa(const Conv0M& p, long r0, int lane) {
  const long base = frame_base(p, r0 + (lane & 15));
  const int kg = lane >> 4;
  s16x4 f = {0, 0, 0, 0};
  if (base >= 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tap = kg * 4 + i;
      if (tap < p.k) f[i] = __builtin_bit_cast(short, p.wave[base + tap]);
      else if (tap == BIAS_TAP) f[i] = (short)0x3F80;  // bf16 1.0
    }
  }
  return f;
}

// =================================================================================================
// forward
// =================================================================================================
__global__ __launch_bounds__(256, 2) void conv0_mfma_fwd_kernel(Conv0M p) {
  __shared__ s16x4 wl[NT * 64];
  const int tid = threadIdx.x, lane = tid & 63, lm = lane & 15, lg = lane >> 4;
  build_w_frags(p, wl, tid);
  float g[NT], be[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { g[t] = bf2f(p.lnw[ch_of(t, lm)]); be[t] = bf2f(p.lnb[ch_of(t, lm)]); }
  __syncthreads();
  const long nsteps = (p.rows + 15) / 16;
  const long wave_id = (long)blockIdx.x * 4 + (tid >> 6), nwaves = (long)gridDim.x * 4;
  s16x4 xa = {0, 0, 0, 0};
  if (wave_id < nsteps) xa = load_xa(p, wave_id * 16, lane);
  for (long st = wave_id; st < nsteps; st += nwaves) {
    const long r0 = st * 16;
    asm volatile("" ::: "memory");   // keep the W-fragment LDS reads inside the loop (hoisted they pin 64 registers)
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma16(xa, wl[t * 64 + lane], f32x4{0.f, 0.f, 0.f, 0.f});
    if (st + nwaves < nsteps) xa = load_xa(p, (st + nwaves) * 16, lane);   // next step's taps fly meanwhile
    float mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) s += acc[t][i];
      mean[i] = reduce16(s) * (1.f / CC);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) { float d = acc[t][i] - mean[i]; q = fmaf(d, d, q); }
      rstd[i] = rsqrtf(reduce16(q) * (1.f / CC) + 1e-5f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long row = r0 + lg * 4 + i;
      if (row < p.rows) {
        if (lm == 0) { p.mean[row] = mean[i]; p.rstd[row] = rstd[i]; }
        bf16* yr = p.y + row * CC + lm * 8;
        const float mr = -mean[i] * rstd[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e += 2) {              // two channels at a time: packed FMAs (common.h, gelu_exact2)
            const int t = 8 * u + e;
            const f32x2 xh = pk_fma(f32x2{acc[t][i], acc[t + 1][i]}, splat2(rstd[i]), splat2(mr));
            const f32x2 y = gelu_exact2(pk_fma(xh, f32x2{g[t], g[t + 1]}, f32x2{be[t], be[t + 1]}));
            o[e] = f2bf(y.x); o[e + 1] = f2bf(y.y);
          }
          *(bf16x8*)(yr + 128 * u) = o;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}

// =================================================================================================
// backward: dW, dbias, dgamma, dbeta (the waveform needs no gradient)
// =================================================================================================
// MAP: channel layout (ch_map).  Round 5 measured, same box, three interleaved rounds (tools/conv0_sweep.sh): the rounds 1-4
// layout (MAP 1: a lane owns 32 consecutive channels) 248-257 us, the forward's new layout (MAP 0) 260-264 us - the dy loads
// are prefetched a whole step ahead, their instruction count does not matter here and four loads of one 64-byte run hit L1 -
// wider scheduling windows (a barrier every 2 / 4 tile pairs instead of every pair) 251-262 us: no effect; a form with 2 or 4
// waves sharing a 16-row step (two waves per SIMD; tools/probes/conv0_bwd_split_round5.diff) 250-256 us / 298-321 us.
// Counters (tools/conv0_pmc.sh): this kernel issues vector instructions 47 % of its wave-cycles and waits 40 %.
// PIPE (round 5, the product's choice): pass 1 as a software pipeline over tile pairs inside the wave - 252-257 -> 239-242 us on
// the same box (tools/conv0_sweep.sh).  One stage deeper (MFMAs two pairs ahead) 243-245 us; the same treatment of pass 2: nothing.
template <int MAP, int PIPE>
__global__ __launch_bounds__(256, 1) void conv0_mfma_bwd_kernel(Conv0M p) {
  __shared__ s16x4 wl[NT * 64];
  __shared__ s16x4 dzs[4][NT * 64];      // per wave: dz fragments of the current 16 rows (pass 1 -> pass 2)
  using G = GT<MB_BWD>;
  __shared__ float gdtab[2 * G::N];      // 48 KiB: gelu'
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lm = lane & 15, lg = lane >> 4;
  s16x4* dzl = dzs[wid];
  // after the row loop the dz buffers are dead and hold the block reductions instead
  float* slab = (float*)&dzs[0][0];                       // dW/dbias sums [ch][16 taps] (+16 floats of skew per 128 ch)
  float (*gsl)[2][CC] = (float (*)[2][CC])((float*)&dzs[0][0] + CC * 16 + 64);   // per-wave dgamma / dbeta
  build_w_frags<MAP>(p, wl, tid);
  for (int i = tid; i < 2 * G::N; i += 256) gdtab[i] = gelu_grad(G::arg(i));
  // gamma (low half) and beta (high half) of channel lm*32 + t as bf16 bits: 32 registers instead of 64
  uint32_t gb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bf16x2 v; v[0] = p.lnw[ch_map<MAP>(t, lm)]; v[1] = p.lnb[ch_map<MAP>(t, lm)];
    gb[t] = __builtin_bit_cast(uint32_t, v);
  }
  // the pipelined pass 1 has no registers for them: the same words as a [tile][lm] table in LDS (2 KiB), read with the W fragments
  __shared__ uint32_t gbl[PIPE ? NT * 16 : 1];
  if constexpr (PIPE) {
    if (tid < 16) {
#pragma unroll
      for (int t = 0; t < NT; ++t) gbl[t * 16 + tid] = gb[t];
    }
  }
  f32x4 dwacc[NT];   // tile t: lane (tap = lm, rg) holds channels ch_of(t, rg*4+i)
#pragma unroll
  for (int t = 0; t < NT; ++t) dwacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dgamma / dbeta also ride the matrix cores: tile t's row sums go to column (t & 15) of accumulator (t >> 4)
  // through a one-hot B operand, so 512 channels cost 8 accumulator registers each instead of 32 + 32 VGPRs.
  // Lane (n = lm, rg) of accumulator h then holds channels ch_of(h*16 + lm, rg*4+i).
  f32x4 dgacc[2], dbacc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { dgacc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; dbacc[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  __syncthreads();
  const long nsteps = (p.rows + 15) / 16;
  const long wave_id = (long)blockIdx.x * 4 + wid, nwaves = (long)gridDim.x * 4;
  // Operands of one 16-row step.  They are fetched one step ahead: dy right after pass 1 (its registers are free
  // from there on), the rest into a second set, so a wave (alone on its SIMD) never waits on HBM at a step's top.
  //   xa: A operand of the conv MFMA;  xb: B operand of the dW MFMA - lane (n = tap = lm, kg = lg) holds
  //   x[row lg*4+i][tap], tap BIAS_TAP = 1 -> dbias;  dyb[u][i]: channels u*128 + lm*8 .. +7 of row lg*4+i
  s16x4 xa = {0, 0, 0, 0}, xb = {0, 0, 0, 0}, nxa = {0, 0, 0, 0}, nxb = {0, 0, 0, 0};
  float mr[4], rstd[4], nmr[4], nrstd[4];
  u32x4 dyb[4][4];
  auto load_small = [&](long r0, s16x4& xa_, s16x4& xb_, float (&mr_)[4], float (&rstd_)[4]) {
    xa_ = load_xa(p, r0, lane);
    xb_ = s16x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long row = r0 + lg * 4 + i;
      const long base = frame_base(p, row);
      rstd_[i] = 0.f; mr_[i] = 0.f;
      if (base >= 0) {
        if (lm < p.k) xb_[i] = __builtin_bit_cast(short, p.wave[base + lm]);
        else if (lm == BIAS_TAP) xb_[i] = (short)0x3F80;
        rstd_[i] = p.rstd[row];
        mr_[i] = -p.mean[row] * rstd_[i];
      }
    }
  };
  auto load_dy = [&](long r0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long row = r0 + lg * 4 + i;
      const bf16* dyr = p.dy + row * CC + (MAP == 0 ? lm * 8 : lm * 32);
#pragma unroll
      for (int u = 0; u < 4; ++u) dyb[u][i] = row < p.rows ? *(const u32x4*)(dyr + (MAP == 0 ? 128 : 8) * u) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  if (wave_id < nsteps) {
    load_small(wave_id * 16, xa, xb, mr, rstd);
    load_dy(wave_id * 16);
  }
  for (long st = wave_id; st < nsteps; st += nwaves) {
    // The W fragments and the one-hot operands are loop invariant; hoisted they would pin 128 registers.
    // The clobber makes the LDS reads stay where they are used, the laundered lane id does the same for one-hot.
    asm volatile("" ::: "memory");
    int lmv = lm;
    asm volatile("" : "+v"(lmv));
    const bool more = st + nwaves < nsteps;
    if (more) load_small((st + nwaves) * 16, nxa, nxb, nmr, nrstd);
    // ---- pass 1: dz = dy * gelu'(z), rounded to bf16 and parked in LDS as the [4 rows] fragment of its channel
    // (16 KiB per wave; pass 2 reads it back both as floats and, unchanged, as an MFMA A operand);
    // row sums of dxhat and dxhat*xhat
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE) {
      // The wave is alone on its SIMD: nothing covers its LDS and MFMA latencies but its own instructions.  Three tile pairs
      // are in flight: while the vector work of pair t is issued (F), the gelu' lookups of pair t+2 are on their way (L, behind
      // its conv MFMAs M) and the W fragments and gamma / beta words of pair t+4 are being read (R).
      s16x4 wa[2];
      uint32_t gv[3][2];
      float xh[2][2][4], gd[2][2][4];
      auto R = [&](int t, uint32_t (&g)[2]) {
        wa[0] = wl[t * 64 + lane]; wa[1] = wl[(t + 1) * 64 + lane];
        g[0] = gbl[t * 16 + lm]; g[1] = gbl[(t + 1) * 16 + lm];
      };
      auto ML = [&](const uint32_t (&g)[2], float (&x)[2][4], float (&d)[2][4], int t_next, uint32_t (&g_next)[2], bool rd) {
        const f32x4 c0 = mfma16(xa, wa[0], f32x4{0.f, 0.f, 0.f, 0.f});
        const f32x4 c1 = mfma16(xa, wa[1], f32x4{0.f, 0.f, 0.f, 0.f});
        if (rd) R(t_next, g_next);                       // into the registers the MFMAs have just read
        const float g0 = __uint_as_float(g[0] << 16), b0 = __uint_as_float(g[0] & 0xFFFF0000u);
        const float g1 = __uint_as_float(g[1] << 16), b1 = __uint_as_float(g[1] & 0xFFFF0000u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[0][i] = fmaf(c0[i], rstd[i], mr[i]); x[1][i] = fmaf(c1[i], rstd[i], mr[i]);
          d[0][i] = gdtab[G::index(G::key(fmaf(x[0][i], g0, b0)))];
          d[1][i] = gdtab[G::index(G::key(fmaf(x[1][i], g1, b1)))];
        }
      };
      auto F = [&](int t, const uint32_t (&g)[2], const float (&x)[2][4], const float (&d)[2][4]) {
        const int u = t >> 3, rgi = (t & 7) >> 1;
        const float g0 = __uint_as_float(g[0] << 16), g1 = __uint_as_float(g[1] << 16);
        s16x4 z0, z1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t pk = dyb[u][i][rgi];
          const float dz0 = bf2f(f2bf(__uint_as_float(pk << 16) * d[0][i]));
          const float dz1 = bf2f(f2bf(__uint_as_float(pk & 0xFFFF0000u) * d[1][i]));
          const float dx0 = dz0 * g0, dx1 = dz1 * g1;
          s1[i] += dx0 + dx1;
          s2[i] = fmaf(dx0, x[0][i], fmaf(dx1, x[1][i], s2[i]));
          z0[i] = bf_bits(dz0); z1[i] = bf_bits(dz1);
        }
        dzl[t * 64 + lane] = z0;
        dzl[(t + 1) * 64 + lane] = z1;
      };
      R(0, gv[0]);
      ML(gv[0], xh[0], gd[0], 2, gv[1], true);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; t += 2) {
        const int k = (t >> 1) & 1, q = (t >> 1) % 3;
        if (t + 2 < NT) ML(gv[(q + 1) % 3], xh[k ^ 1], gd[k ^ 1], t + 4, gv[(q + 2) % 3], t + 4 < NT);
        __builtin_amdgcn_sched_barrier(0);
        F(t, gv[q], xh[k], gd[k]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
      const f32x4 c0 = mfma16(xa, wl[t * 64 + lane], f32x4{0.f, 0.f, 0.f, 0.f});
      const f32x4 c1 = mfma16(xa, wl[(t + 1) * 64 + lane], f32x4{0.f, 0.f, 0.f, 0.f});
      const int u = t >> 3, rgi = (t & 7) >> 1;
      uint32_t gv0 = gb[t], gv1 = gb[t + 1];   // laundered: the unpacked floats must not be hoisted out of the loop
      asm volatile("" : "+v"(gv0), "+v"(gv1));
      const float g0 = __uint_as_float(gv0 << 16), b0 = __uint_as_float(gv0 & 0xFFFF0000u);
      const float g1 = __uint_as_float(gv1 << 16), b1 = __uint_as_float(gv1 & 0xFFFF0000u);
      float dz0[4], dz1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t pk = dyb[u][i][rgi];
        const float xh0 = fmaf(c0[i], rstd[i], mr[i]), xh1 = fmaf(c1[i], rstd[i], mr[i]);
        const float gd0 = gdtab[G::index(G::key(fmaf(xh0, g0, b0)))];
        const float gd1 = gdtab[G::index(G::key(fmaf(xh1, g1, b1)))];
        dz0[i] = bf2f(f2bf(__uint_as_float(pk << 16) * gd0));
        dz1[i] = bf2f(f2bf(__uint_as_float(pk & 0xFFFF0000u) * gd1));
        const float dx0 = dz0[i] * g0, dx1 = dz1[i] * g1;
        s1[i] += dx0 + dx1;
        s2[i] = fmaf(dx0, xh0, fmaf(dx1, xh1, s2[i]));
      }
      s16x4 z0, z1;
#pragma unroll
      for (int i = 0; i < 4; ++i) { z0[i] = bf_bits(dz0[i]); z1[i] = bf_bits(dz1[i]); }
      dzl[t * 64 + lane] = z0;
      dzl[(t + 1) * 64 + lane] = z1;
      __builtin_amdgcn_sched_barrier(0);   // keep tile pairs in program order: hoisting MFMAs blows the register budget
    }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[i] = reduce16(s1[i]) * (1.f / CC); s2[i] = reduce16(s2[i]) * (1.f / CC); }
    asm volatile("" ::: "memory");
    if (more) load_dy((st + nwaves) * 16);   // lands during pass 2
    // ---- pass 2: dgamma, dbeta, dconv -> dW (the conv is recomputed: 32 MFMAs are cheaper than 128 registers)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 c = mfma16(xa, wl[t * 64 + lane], f32x4{0.f, 0.f, 0.f, 0.f});
      const s16x4 za = dzl[t * 64 + lane];
      uint32_t gv = PIPE ? gbl[t * 16 + lm] : gb[t];
      asm volatile("" : "+v"(gv));
      const float gt = __uint_as_float(gv << 16);
      s16x4 dc, ph, pl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dz = __uint_as_float((uint32_t)(uint16_t)za[i] << 16);
        const float xh = fmaf(c[i], rstd[i], mr[i]);
        const float pr = dz * xh;
        dc[i] = bf_bits(rstd[i] * (fmaf(dz, gt, -s1[i]) - xh * s2[i]));
        ph[i] = bf_bits(pr);                                        // dz*xhat as hi + lo bf16 parts:
        pl[i] = bf_bits(pr - __uint_as_float((uint32_t)(uint16_t)ph[i] << 16));   // ~16 mantissa bits through the MFMA
      }
      const uint32_t ohw = (lmv == (t & 15)) ? 0x3F803F80u : 0u;
      const u32x2 ohv = {ohw, ohw};
      const s16x4 oh = __builtin_bit_cast(s16x4, ohv);
      dbacc[t >> 4] = mfma16(za, oh, dbacc[t >> 4]);   // dz is bf16-exact: the stored fragment IS the A operand
      dgacc[t >> 4] = mfma16(ph, oh, dgacc[t >> 4]);
      dgacc[t >> 4] = mfma16(pl, oh, dgacc[t >> 4]);
      dwacc[t] = mfma16(dc, xb, dwacc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    xa = nxa; xb = nxb;
#pragma unroll
    for (int i = 0; i < 4; ++i) { mr[i] = nmr[i]; rstd[i] = nrstd[i]; }
  }
  __syncthreads();   // the dz buffers are reused below
  // ---- block reduction (no LDS float atomics) and one global atomic per value per block
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = ch_map<MAP>(h * 16 + lm, lg * 4 + i);
      gsl[wid][0][ch] = dgacc[h][i];
      gsl[wid][1][ch] = dbacc[h][i];
    }
  for (int w = 0; w < 4; ++w) {   // waves take turns adding their dW tiles into the slab
    if (wid == w) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ch = ch_map<MAP>(t, lg * 4 + i);
          float* a = slab + ch * 16 + (ch >> 7) * 16 + lm;
          *a = (w == 0 ? 0.f : *a) + dwacc[t][i];
        }
    }
    __syncthreads();
  }
  for (int i = tid; i < CC; i += 256) {
    atomicAdd(&p.dlnw[i], (gsl[0][0][i] + gsl[1][0][i]) + (gsl[2][0][i] + gsl[3][0][i]));
    atomicAdd(&p.dlnb[i], (gsl[0][1][i] + gsl[1][1][i]) + (gsl[2][1][i] + gsl[3][1][i]));
    if (p.dcbias) atomicAdd(&p.dcbias[i], slab[i * 16 + (i >> 7) * 16 + BIAS_TAP]);
  }
  for (int i = tid; i < CC * p.k; i += 256) {
    const int ch = i / p.k, tap = i - ch * p.k;
    atomicAdd(&p.dw[i], slab[ch * 16 + (ch >> 7) * 16 + tap]);
  }
}


}  // namespace

bool conv0_mfma_ok(int C, int k) { return C == CC && k >= 1 && k <= 10; }

int conv0_mfma_fwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, void* y,
                   float* mean, float* rstd, int B, int L, int k, int s, hipStream_t st) {
  Conv0M p{};
  p.wave = (const bf16*)wave; p.w = (const bf16*)w; p.cbias = (const bf16*)cbias; p.lnw = (const bf16*)lnw; p.lnb = (const bf16*)lnb;
  p.y = (bf16*)y; p.mean = mean; p.rstd = rstd; p.B = B; p.L = L; p.L0 = (L - k) / s + 1; p.k = k; p.s = s;
  p.rows = (long)B * p.L0;
  const long nsteps = (p.rows + 15) / 16;
  const int grid = (int)std::min<long>((nsteps + 3) / 4, 512);
  hipLaunchKernelGGL(conv0_mfma_fwd_kernel, dim3(grid), dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "conv0_fwd");
}

int conv0_mfma_bwd(const void* wave, const void* w, const void* cbias, const void* lnw, const void* lnb, const float* mean,
                   const float* rstd, const void* dy, float* dw, float* dcbias, float* dlnw, float* dlnb, int B, int L, int k,
                   int s, hipStream_t st) {
  Conv0M p{};
  p.wave = (const bf16*)wave; p.w = (const bf16*)w; p.cbias = (const bf16*)cbias; p.lnw = (const bf16*)lnw; p.lnb = (const bf16*)lnb;
  p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd); p.dy = (const bf16*)dy;
  p.dw = dw; p.dcbias = dcbias; p.dlnw = dlnw; p.dlnb = dlnb;
  p.B = B; p.L = L; p.L0 = (L - k) / s + 1; p.k = k; p.s = s;
  p.rows = (long)B * p.L0;
  const long nsteps = (p.rows + 15) / 16;
  const int grid = (int)std::min<long>((nsteps + 3) / 4, 256);
  static const int map_env = W2VS_ENV_INT("W2VS_CONV0_BWD_MAP", 1);     // tuning build: 0 = the forward's channel layout (A/B)
  static const int pipe_env = W2VS_ENV_INT("W2VS_CONV0_BWD_PIPE", 1);   // tuning build: 0 = pass 1 one tile pair at a time (A/B)
  if (map_env == 0) hipLaunchKernelGGL((conv0_mfma_bwd_kernel<0, 0>), dim3(grid), dim3(256), 0, st, p);
  else if (pipe_env == 0) hipLaunchKernelGGL((conv0_mfma_bwd_kernel<1, 0>), dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv0_mfma_bwd_kernel<1, 1>), dim3(grid), dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "conv0_bwd");
}

}  // namespace w2vs
