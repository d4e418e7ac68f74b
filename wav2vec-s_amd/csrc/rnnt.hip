// Transducer loss for gfx950: RNN-T and the delay transducer of
// /root/reference/warp_transducer (include/detail/gpu_rnnt_kernel.h, reduce.h, gpu_rnnt.h, delay_transducer.h),
// behind the reference's own C interface (include/w2vs_rnnt.h).
//
// Three launches instead of the reference's seven (two reductions, four lattice kernels, one gradient kernel + a
// memset of the gradient):
//   rows_kernel     one wave per (b, t, u) row: log-softmax denominator in ONE pass over the V activations (online
//                   max / sum), and the two log-probabilities the lattice needs (blank, next label) are extracted here
//                   into anti-diagonal-major arrays, so the lattice kernels never touch the [B,T,U,V] tensor again.
//   lattice_kernel  grid (B, 2): block (b, 0) runs the alpha AND alpha-delay recursions, block (b, 1) beta AND beta-delay.
//                   Thread u walks the anti-diagonals; a cell's two predecessors are its own previous value (a register)
//                   and its neighbour's previous value (LDS, double buffered: one barrier per diagonal); the
//                   log-probabilities of diagonal n are contiguous in memory and are loaded four diagonals ahead.
//   grad_kernel     one wave per row: every per-cell scalar is computed once, then the row is streamed with 16-byte
//                   loads / stores (1-2 exp per element); rows outside a sample's T x U are written as zeros here, which
//                   replaces the reference's memset pass over the whole gradient.
// HBM-bound: rows_kernel reads 4 V bytes per valid row, grad_kernel reads 4 V and writes 4 V bytes per row.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include "../../include/w2vs_rnnt.h"

namespace {

constexpr int WAVE = 64;
constexpr int ROWS_PER_BLOCK = 4;     // 256 threads, one wave per row

struct Dims {
  int B, maxT, maxU, V, blank, D;     // D = maxT + maxU - 1 anti-diagonals
  const int* cells;                   // compact mode: acts / grads hold ONLY these cells (b*maxT*maxU + t*maxU + u), in this
  long n_rows;                        // order; NULL = dense [B, maxT, maxU, V].  n_rows = rows of acts / grads
};

struct Work {                         // workspace carve-up (floats)
  float *denom, *lpb, *lpl, *alpha, *beta, *adel, *bdel, *ll, *llb, *dexp, *dexpb;
};

__host__ __device__ inline size_t align64(size_t n) { return (n + 63) & ~size_t(63); }

inline size_t work_floats(int B, int maxT, int maxU, bool delay) {
  const size_t cells = (size_t)B * maxT * maxU, diag = (size_t)B * (maxT + maxU - 1) * maxU;
  return align64(cells) + align64(diag) * (delay ? 6 : 4) + align64(B) * 4;
}

inline Work carve(void* ws, int B, int maxT, int maxU, bool delay) {
  const size_t cells = (size_t)B * maxT * maxU, diag = (size_t)B * (maxT + maxU - 1) * maxU;
  float* p = (float*)ws;
  Work w{};
  w.denom = p; p += align64(cells);
  w.lpb = p; p += align64(diag);
  w.lpl = p; p += align64(diag);
  w.alpha = p; p += align64(diag);
  w.beta = p; p += align64(diag);
  if (delay) {
    w.adel = p; p += align64(diag);
    w.bdel = p; p += align64(diag);
  }
  w.ll = p; p += align64(B);
  w.llb = p; p += align64(B);
  w.dexp = p; p += align64(B);
  w.dexpb = p;
  return w;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ long dix(const Dims& d, int b, int t, int u) {       // anti-diagonal-major cell index
  return ((long)b * d.D + (t + u)) * d.maxU + u;
}

// rnnt_helper.h:17-26 log_sum_exp.  The correction log(1 + e^-|a-b|) lies in (0, ln 2]; evaluated with the hardware
// exp2 / log2 (v_exp_f32, v_log_f32: ~1 ulp) its absolute error is ~1e-7, far below one ulp of a lattice value
// (|alpha| reaches the 100s: ulp 1.5e-5) - and the recursion's critical path per anti-diagonal shrinks from ~100 to ~15
// dependent instructions, which is what bounds this latency-bound kernel.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float lse(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b), dlt = -fabsf(a - b);
  return m + __builtin_amdgcn_logf(1.f + fast_exp(dlt)) * 0.69314718055994530942f;
}

// ------------------------------------------------------------------------------------------------ rows: denominators
// online (max, sum) so that the row is read once; merged across lanes with a butterfly
__device__ __forceinline__ void online_add(float& m, float& s, float x) {
  if (x > m) {
    s = s * expf(m - x) + 1.f;
    m = x;
  } else {
    s += expf(x - m);
  }
}

__global__ __launch_bounds__(WAVE* ROWS_PER_BLOCK) void rows_kernel(const float* __restrict__ acts,
                                                                    const int* __restrict__ labels,
                                                                    const int* __restrict__ xlen,
                                                                    const int* __restrict__ ylen, Work w, Dims d) {
  const int lane = threadIdx.x & (WAVE - 1);
  const long arow = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);    // row of acts
  if (arow >= d.n_rows) return;
  const long row = d.cells ? (long)d.cells[arow] : arow;                        // lattice cell
  if (row < 0 || row >= (long)d.B * d.maxT * d.maxU) return;
  const int u = (int)(row % d.maxU);
  const long bt = row / d.maxU;
  const int t = (int)(bt % d.maxT), b = (int)(bt / d.maxT);
  const int T = clampi(xlen[b], 1, d.maxT), U = clampi(ylen[b] + 1, 1, d.maxU);   // lengths past the tensor would read out of bounds
  if (t >= T || u >= U) return;                       // never read: the lattice and the gradient skip these cells
  const float* x = acts + arow * d.V;
  float m = -INFINITY, s = 0.f;
  if ((d.V & 3) == 0) {
    const float4* x4 = (const float4*)x;
    for (int i = lane; i < (d.V >> 2); i += WAVE) {
      const float4 v = x4[i];
      online_add(m, s, v.x); online_add(m, s, v.y); online_add(m, s, v.z); online_add(m, s, v.w);
    }
  } else {
    for (int i = lane; i < d.V; i += WAVE) online_add(m, s, x[i]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, WAVE), s2 = __shfl_xor(s, o, WAVE);
    const float mm = fmaxf(m, m2);
    const float sa = (m == -INFINITY) ? 0.f : s * expf(m - mm), sb = (m2 == -INFINITY) ? 0.f : s2 * expf(m2 - mm);
    m = mm;
    s = sa + sb;
  }
  if (lane == 0) {
    const float den = -m - logf(s);                   // reduce.h:102: -max - log(sum exp(x - max))
    w.denom[row] = den;
    const long c = dix(d, b, t, u);
    w.lpb[c] = den + x[d.blank];
    w.lpl[c] = (u < U - 1) ? den + x[labels[(long)b * (d.maxU - 1) + u]] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ lattice recursions
constexpr int PF = 4;                                  // diagonals of inputs in flight per thread

struct LatIn { float pb, pl, dv; };                    // log p(blank), log p(label), emission cost feeding one step

template <bool DELAY>
__global__ void lattice_kernel(const int* __restrict__ xlen, const int* __restrict__ ylen,
                               const float* __restrict__ delay_values, Work w, Dims d) {
  extern __shared__ float sh[];                        // [2][blockDim] values, [2][blockDim] delays
  const int b = blockIdx.x, u = threadIdx.x, W = blockDim.x;
  const int T = clampi(xlen[b], 1, d.maxT), U = clampi(ylen[b] + 1, 1, d.maxU);   // lengths past the tensor would read out of bounds
  float* shv = sh;
  float* shd = sh + 2 * W;
  const long base = (long)b * d.D * d.maxU;
  const float* lpb = w.lpb + base;
  const float* lpl = w.lpl + base;
  const float* dv = DELAY ? delay_values + (long)b * d.maxT * d.maxU : nullptr;
  const int N = T + U - 1;                             // diagonals 0 .. N-1
  const bool fwd = blockIdx.y == 0;
  float* out = (fwd ? w.alpha : w.beta) + base;
  float* outd = DELAY ? (fwd ? w.adel : w.bdel) + base : nullptr;

  // Inputs of step n do not depend on the recursion, so they are fetched PF diagonals ahead (a step is ~300 cycles of
  // LDS + exp/log1p latency; a global load issued in the same step would add its full latency to every diagonal).
  //  alpha step n, cell (t,u) = (n-u, u): blank from (t-1,u) = diagonal n-1 column u; label from (t,u-1) = diagonal n-1
  //                                       column u-1; emission cost dv[t,u] (gpu_rnnt_kernel.h:76, :81)
  //  beta  step n, cell (t,u):            blank / label / emission cost of the cell itself (:186, :193)
  // Unconditional loads from clamped (always in-bounds) addresses: a load under a per-lane condition makes the compiler
  // resolve the "else 0" right away, i.e. wait for the load it just issued.  Values of cells outside the lattice are
  // loaded (workspace garbage) and never used.
  const int uc = min(u, d.maxU - 1), um1 = clampi(u - 1, 0, d.maxU - 1);
  auto fetch = [&](int n) -> LatIn {
    LatIn r;
    const int row = clampi(fwd ? n - 1 : n, 0, d.D - 1);
    r.pb = lpb[(long)row * d.maxU + uc];
    r.pl = lpl[(long)row * d.maxU + (fwd ? um1 : uc)];
    r.dv = 0.f;
    if (DELAY) r.dv = dv[(long)clampi(n - u, 0, T - 1) * d.maxU + uc];
    return r;
  };

  float self = 0.f, selfd = 0.f;                       // alpha(t-1,u) / beta(t+1,u) and the matching delay
  shv[u] = 0.f; shd[u] = 0.f; shv[W + u] = 0.f; shd[W + u] = 0.f;
  if (fwd && u == 0) {                                 // alpha(0,0) = 0, alpha_delay(0,0) = 0
    out[0] = 0.f;
    if (DELAY) outd[0] = 0.f;
  }
  const int dir = fwd ? 1 : -1;
  const int first = fwd ? 1 : N - 1, steps = fwd ? N - 1 : N;
  LatIn cur[PF], nxt[PF];
  float resv[PF], resd[PF];
  int resn[PF];                                        // diagonal of a finished cell, -1 = nothing to write
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    cur[j] = fetch(first + dir * j);
    resn[j] = -1;
  }
  // Loads and stores share one in-order counter (vmcnt) on gfx9: a wait for prefetched inputs also waits for every store
  // issued before it.  So results are parked in registers and written at the START of the next chunk, BEFORE that chunk's
  // prefetches: whatever a later wait covers was issued at least PF diagonals earlier and has long completed.
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      if (resn[j] >= 0) {
        out[(long)resn[j] * d.maxU + u] = resv[j];
        if (DELAY) outd[(long)resn[j] * d.maxU + u] = resd[j];
      }
      resn[j] = -1;
    }
  };
  __syncthreads();
  for (int s0 = 0; s0 < steps; s0 += PF) {
    flush();
#pragma unroll
    for (int j = 0; j < PF; ++j) nxt[j] = fetch(first + dir * (s0 + PF + j));
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      if (s0 + j >= steps) break;                      // block-uniform
      const int n = first + dir * (s0 + j);
      const int t = n - u;
      const float* prv = shv + ((n - dir) & 1) * W;    // the neighbouring diagonal computed in the previous step
      const float* prd = shd + ((n - dir) & 1) * W;
      const bool valid = u < U && t >= 0 && t < T;
      const LatIn in = cur[j];
      // branch-free: every lane evaluates the general cell, boundary cells are selected afterwards
      const int nb = fwd ? max(u - 1, 0) : min(u + 1, W - 1);
      const float nbv = prv[nb], nbd = prd[nb];        // alpha(t,u-1) / beta(t,u+1) and its delay
      const float no_emit = self + in.pb, emit = nbv + in.pl;
      const float gen = lse(emit, no_emit);
      float gend = 0.f;
      if (DELAY) gend = selfd + fast_exp(emit - gen) * (nbd + in.dv - selfd);   // convex form: the two path weights sum to 1,
                                                       // so fp32 rounding of `gen` scales only the DIFFERENCE of the delays
      float v, vd;
      if (fwd) {
        // u == 0: only the blank transition (t > 0 here); t == 0: only the label transition (gpu_rnnt_kernel.h:27-38, :75-84)
        v = (u == 0) ? no_emit : (t == 0 ? emit : gen);
        vd = (u == 0) ? 0.f : (t == 0 ? nbd + in.dv : gend);
      } else {
        // (T-1,U-1): log p(blank); u == U-1: blank only; t == T-1: label only (:136-150, :180-195)
        const bool lastu = u == U - 1, lastt = t == T - 1;
        v = lastu ? (lastt ? in.pb : no_emit) : (lastt ? emit : gen);
        vd = lastu ? (lastt ? 0.f : selfd) : (lastt ? nbd + in.dv : gend);
      }
      v = valid ? v : 0.f;
      vd = valid ? vd : 0.f;
      resv[j] = v;
      resd[j] = vd;
      resn[j] = valid ? n : -1;
      self = valid ? v : self;
      selfd = valid ? vd : selfd;
      shv[(n & 1) * W + u] = v;
      shd[(n & 1) * W + u] = vd;
      // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait every diagonal for this step's global stores
      // and for the prefetches just issued (~1 us per diagonal, measured)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) cur[j] = nxt[j];
  }
  flush();
  if (fwd) {
    if (u == U - 1) {                                  // `self` is alpha(T-1, U-1)
      w.ll[b] = self + lpb[(long)(N - 1) * d.maxU + (U - 1)];
      if (DELAY) w.dexp[b] = selfd;
    }
  } else if (u == 0) {                                 // `self` is beta(0, 0)
    w.llb[b] = self;
    if (DELAY) w.dexpb[b] = selfd;
  }
}

// ------------------------------------------------------------------------------------------------ gradient rows
struct RowK {                 // per-row constants
  float den, k_main, e0c0, e1c1, sub_blank, sub_label, add_blank, add_label, smooth, scale, up;
  int blank, label;           // label = -1 when the row has no label transition
};

__device__ __forceinline__ float grad_elem(const RowK& k, float x, int v) {
  const float logpk = k.den + x;
  const float p = expf(logpk);
  float g = (k.smooth == 1.f) ? p * k.k_main : expf(k.k_main + logpk);   // k_main: exp(..) resp. the exponent itself
  float g2 = -p * (k.e0c0 + k.e1c1);
  if (v == k.blank) {
    g -= k.sub_blank;
    g2 += k.add_blank;
  }
  if (v == k.label) {
    g -= k.sub_label;
    g2 += k.add_label;
  }
  return (g + k.scale * g2) * k.up;
}

// OUT16: gradients as bf16 (what the projection's dgrad / wgrad GEMMs consume) - halves the kernel's write traffic
__device__ __forceinline__ unsigned short to_bf16(float v) {                // round to nearest even
  unsigned int x = __float_as_uint(v);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((x >> 16) | 0x40);
  return (unsigned short)((x + 0x7fffu + ((x >> 16) & 1u)) >> 16);
}

template <bool DELAY, bool OUT16>
__global__ __launch_bounds__(WAVE* ROWS_PER_BLOCK) void grad_kernel(const float* __restrict__ acts, void* __restrict__ grads_,
                                                                    const int* __restrict__ labels,
                                                                    const int* __restrict__ xlen, const int* __restrict__ ylen,
                                                                    const float* __restrict__ delay_values, Work w, Dims d,
                                                                    float delay_scale, float smooth, int consistent_index,
                                                                    const float* __restrict__ up_dev, int up_n, float up_host) {
  const int lane = threadIdx.x & (WAVE - 1);
  const long arow = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);    // row of acts / grads
  if (arow >= d.n_rows) return;
  long row = d.cells ? (long)d.cells[arow] : arow;                              // lattice cell
  const bool in_range = row >= 0 && row < (long)d.B * d.maxT * d.maxU;
  if (!in_range) row = 0;
  const int u = (int)(row % d.maxU);
  const long bt = row / d.maxU;
  const int t = (int)(bt % d.maxT), b = (int)(bt / d.maxT);
  const int T = clampi(xlen[b], 1, d.maxT), U = clampi(ylen[b] + 1, 1, d.maxU);   // lengths past the tensor would read out of bounds
  float* g = (float*)grads_ + (OUT16 ? 0 : arow * d.V);
  unsigned short* g16 = (unsigned short*)grads_ + (OUT16 ? arow * d.V : 0);
  const bool vec = (d.V & 3) == 0;
  if (!in_range || t >= T || u >= U) {                              // the reference zeroes the whole tensor first
    if (OUT16) {
      if (vec) {
        uint2* g2 = (uint2*)g16;
        for (int i = lane; i < (d.V >> 2); i += WAVE) g2[i] = make_uint2(0u, 0u);
      } else {
        for (int i = lane; i < d.V; i += WAVE) g16[i] = 0;
      }
    } else if (vec) {
      float4* g4 = (float4*)g;
      for (int i = lane; i < (d.V >> 2); i += WAVE) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int i = lane; i < d.V; i += WAVE) g[i] = 0.f;
    }
    return;
  }
  const long c = dix(d, b, t, u);
  const float a = w.alpha[c], bb = w.beta[c], ll = w.ll[b];
  const float logpb = w.lpb[c];
  const bool has_t = t < T - 1, has_u = u < U - 1;
  const float logpy = has_u ? w.lpl[c] : 0.f;
  const float b_t1 = has_t ? w.beta[dix(d, b, t + 1, u)] : 0.f;
  const float b_u1 = has_u ? w.beta[dix(d, b, t, u + 1)] : 0.f;
  RowK k;
  k.den = w.denom[row];
  k.smooth = smooth;
  k.scale = DELAY ? delay_scale : 0.f;
  k.blank = d.blank;
  k.label = has_u ? labels[(long)b * (d.maxU - 1) + u] : -1;
  const float occ = (a + bb - ll) * smooth;            // gpu_rnnt_kernel.h:399
  k.k_main = (smooth == 1.f) ? expf(occ) : occ;
  k.up = up_host * (up_n == 0 ? 1.f : up_dev[up_n == 1 ? 0 : b]);     // d(loss)/d(cost_b) folded into the only pass
  float c0 = 0.f, c1 = 0.f;
  k.e0c0 = k.e1c1 = k.add_blank = k.add_label = 0.f;
  if (DELAY) {
    const float ad = w.adel[c], dexp = w.dexp[b];
    if (has_t) {
      c0 = ad + w.bdel[dix(d, b, t + 1, u)] - dexp;                                    // :403-406
      k.e0c0 = expf(a + b_t1 - ll + logpb) * c0;
      k.add_blank = expf(a + b_t1 + logpb - ll) * c0;                                  // :417
    }
    if (has_u) {
      const float dvv = consistent_index ? delay_values[row] : delay_values[bt];       // :409 reads [bt]
      c1 = ad + dvv + w.bdel[dix(d, b, t, u + 1)] - dexp;
      k.e1c1 = expf(a + b_u1 - ll + logpy) * c1;
      k.add_label = expf(a + b_u1 + logpy - ll) * c1;                                  // :421
    }
  }
  k.sub_blank = 0.f;
  if (!has_t && !has_u) k.sub_blank += expf(smooth * (a - ll + logpb));                // :413-415 (t = T-1, u = U-1)
  if (has_t) k.sub_blank += expf(smooth * (a - ll + b_t1 + logpb));                    // :416
  k.sub_label = has_u ? expf(smooth * (a + b_u1 - ll + logpy)) : 0.f;                  // :420
  if (!has_t && has_u) { /* t = T-1, u < U-1: only the label transition leaves the cell */ }
  const float* x = acts + arow * d.V;
  if (vec) {
    const float4* x4 = (const float4*)x;
    float4* g4 = (float4*)g;
    for (int i = lane; i < (d.V >> 2); i += WAVE) {
      const float4 v = x4[i];
      float4 o;
      o.x = grad_elem(k, v.x, 4 * i);
      o.y = grad_elem(k, v.y, 4 * i + 1);
      o.z = grad_elem(k, v.z, 4 * i + 2);
      o.w = grad_elem(k, v.w, 4 * i + 3);
      if (OUT16)
        ((uint2*)g16)[i] = make_uint2((unsigned)to_bf16(o.x) | ((unsigned)to_bf16(o.y) << 16),
                                      (unsigned)to_bf16(o.z) | ((unsigned)to_bf16(o.w) << 16));
      else
        g4[i] = o;
    }
  } else {
    for (int i = lane; i < d.V; i += WAVE) {
      const float o = grad_elem(k, x[i], i);
      if (OUT16) g16[i] = to_bf16(o); else g[i] = o;
    }
  }
}

// costs_dev [3, B]: NLL, expected delay, total
__global__ void costs_kernel(Work w, float* costs, int B, float delay_scale, int delay) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float nll = -w.ll[b];
  const float de = delay ? w.dexp[b] : 0.f;
  costs[b] = nll;
  costs[B + b] = de;
  costs[2 * B + b] = nll + delay_scale * de;
}

__global__ void delay_values_kernel(int kind, const int* __restrict__ src, const int* __restrict__ tgt,
                                    float* __restrict__ out, int B, int T, int U) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * T * U) return;
  const int u = (int)(i % U);
  const int s = (int)((i / U) % T), b = (int)(i / ((long)U * T));
  const float sl = (float)src[b], tl = (float)tgt[b];
  float v;
  if (kind == 0) {
    v = (float)s / sl;
  } else {
    v = ((float)s + 1.f) * (tl / sl) - ((float)u + 1.f);
    v = (kind == 1) ? fabsf(v) : fmaxf(v, 0.f);
    v = v / tl;
  }
  out[i] = v;
}

// ------------------------------------------------------------------------------------------------ label-smoothed CE rows
// fairseq label_smoothed_nll_loss (fs/criterions/label_smoothed_cross_entropy.py:33-50) on log_softmax(logits), summed
// over rows whose target is not `pad`, as TransducerOut.cross_entropy uses it (rain/layers/attention_transducer.py:339-360):
//   nll = lse - x_y ; smooth = V lse - sum x ; loss = (1 - eps - eps_i) nll + eps_i smooth,  eps_i = eps / (V - 1)
// One wave per row: statistics in one pass, then (optionally) the gradient row
//   d loss / d x_v = (1 - eps - eps_i) (p_v - [v = y]) + eps_i (V p_v - 1), times `scale`.
template <bool OUT16>
__global__ __launch_bounds__(WAVE* ROWS_PER_BLOCK) void ls_ce_kernel(const float* __restrict__ logits, const int* __restrict__ target,
                                                                     void* __restrict__ grads_, float* __restrict__ sums, long R,
                                                                     int V, int pad, float eps, float scale) {
  const int lane = threadIdx.x & (WAVE - 1);
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= R) return;
  const int y = target[row];
  const float* x = logits + row * V;
  float* g = (float*)grads_ + (OUT16 ? 0 : row * V);
  unsigned short* g16 = (unsigned short*)grads_ + (OUT16 ? row * V : 0);
  if (y == pad || y < 0 || y >= V) {
    if (grads_)
      for (int i = lane; i < V; i += WAVE) { if (OUT16) g16[i] = 0; else g[i] = 0.f; }
    return;
  }
  float m = -INFINITY, s = 0.f, sx = 0.f;
  for (int i = lane; i < V; i += WAVE) {
    const float v = x[i];
    online_add(m, s, v);
    sx += v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, WAVE), s2 = __shfl_xor(s, o, WAVE);
    const float mm = fmaxf(m, m2);
    s = ((m == -INFINITY) ? 0.f : s * expf(m - mm)) + ((m2 == -INFINITY) ? 0.f : s2 * expf(m2 - mm));
    m = mm;
    sx += __shfl_xor(sx, o, WAVE);
  }
  const float lse = m + logf(s);
  const float eps_i = eps / (float)(V - 1), c_nll = 1.f - eps - eps_i;
  if (lane == 0) {
    const float nll = lse - x[y], smooth = (float)V * lse - sx;
    atomicAdd(&sums[0], c_nll * nll + eps_i * smooth);
    atomicAdd(&sums[1], nll);
  }
  if (!grads_) return;
  const float cp = (c_nll + eps_i * (float)V) * scale;
  for (int i = lane; i < V; i += WAVE) {
    const float p = expf(x[i] - lse);
    const float o = cp * p - (i == y ? c_nll * scale : 0.f) - eps_i * scale;
    if (OUT16) g16[i] = to_bf16(o); else g[i] = o;
  }
}

rnntStatus_t check(const float* acts, const int* labels, const int* ylen, const int* xlen, void* workspace, int V, int B,
                   const rnntOptions& opt) {
  if (!acts || !labels || !ylen || !xlen || !workspace || V <= 0 || B <= 0 || opt.maxT <= 0 || opt.maxU <= 0)
    return RNNT_STATUS_INVALID_VALUE;
  if (opt.loc != RNNT_GPU) return RNNT_STATUS_EXECUTION_FAILED;       // no CPU path in this build
  if (opt.maxU > 1024 || opt.blank_label < 0 || opt.blank_label >= V) return RNNT_STATUS_INVALID_VALUE;
  if (((uintptr_t)acts & 15) || ((uintptr_t)workspace & 3)) return RNNT_STATUS_INVALID_VALUE;
  return RNNT_STATUS_SUCCESS;
}

// rows + lattice (+ device costs)
rnntStatus_t run_fwd(const float* acts, const int* labels, const int* ylen, const int* xlen, const float* delay_values, int V,
                     int B, float* costs_dev, void* workspace, float delay_scale, const rnntOptions& opt,
                     const int* cells = nullptr, long n_cells = 0) {
  const rnntStatus_t rc = check(acts, labels, ylen, xlen, workspace, V, B, opt);
  if (rc != RNNT_STATUS_SUCCESS) return rc;
  const bool delay = delay_values != nullptr;
  hipStream_t st = (hipStream_t)opt.stream;
  if (cells && n_cells <= 0) return RNNT_STATUS_INVALID_VALUE;
  const long rows = cells ? n_cells : (long)B * opt.maxT * opt.maxU;
  Dims d{B, opt.maxT, opt.maxU, V, opt.blank_label, opt.maxT + opt.maxU - 1, cells, rows};
  Work w = carve(workspace, B, opt.maxT, opt.maxU, delay);
  const unsigned row_blocks = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  hipLaunchKernelGGL(rows_kernel, dim3(row_blocks), dim3(WAVE * ROWS_PER_BLOCK), 0, st, acts, labels, xlen, ylen, w, d);
  const int W = ((opt.maxU + WAVE - 1) / WAVE) * WAVE;
  const size_t shb = sizeof(float) * 4 * W;
  if (delay)
    hipLaunchKernelGGL(lattice_kernel<true>, dim3(B, 2), dim3(W), shb, st, xlen, ylen, delay_values, w, d);
  else
    hipLaunchKernelGGL(lattice_kernel<false>, dim3(B, 2), dim3(W), shb, st, xlen, ylen, delay_values, w, d);
  if (costs_dev)
    hipLaunchKernelGGL(costs_kernel, dim3((B + 63) / 64), dim3(64), 0, st, w, costs_dev, B, delay_scale, delay ? 1 : 0);
  return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

// gradient rows; needs the workspace run_fwd filled for the same arguments
rnntStatus_t run_bwd(const float* acts, void* grads, const int* labels, const int* ylen, const int* xlen,
                     const float* delay_values, int V, int B, void* workspace, float delay_scale, float smooth, int flags,
                     const float* up_dev, int up_n, float up_host, const rnntOptions& opt, const int* cells = nullptr,
                     long n_cells = 0) {
  const rnntStatus_t rc = check(acts, labels, ylen, xlen, workspace, V, B, opt);
  if (rc != RNNT_STATUS_SUCCESS) return rc;
  if (!grads || ((uintptr_t)grads & 15) || (up_n != 0 && up_n != 1 && up_n != B) || (up_n != 0 && !up_dev))
    return RNNT_STATUS_INVALID_VALUE;
  const bool delay = delay_values != nullptr;
  hipStream_t st = (hipStream_t)opt.stream;
  if (cells && n_cells <= 0) return RNNT_STATUS_INVALID_VALUE;
  const long rows = cells ? n_cells : (long)B * opt.maxT * opt.maxU;
  Dims d{B, opt.maxT, opt.maxU, V, opt.blank_label, opt.maxT + opt.maxU - 1, cells, rows};
  Work w = carve(workspace, B, opt.maxT, opt.maxU, delay);
  const unsigned row_blocks = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  const dim3 gr(row_blocks), bl(WAVE * ROWS_PER_BLOCK);
  const bool o16 = (flags & 2) != 0;
#define W2VS_GRAD(DL, O16)                                                                                              \
  hipLaunchKernelGGL((grad_kernel<DL, O16>), gr, bl, 0, st, acts, grads, labels, xlen, ylen, delay_values, w, d,         \
                     DL ? delay_scale : 0.f, smooth, DL ? (flags & 1) : 0, up_dev, up_n, up_host)
  if (delay) { if (o16) W2VS_GRAD(true, true); else W2VS_GRAD(true, false); }
  else       { if (o16) W2VS_GRAD(false, true); else W2VS_GRAD(false, false); }
#undef W2VS_GRAD
  return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

// the reference's synchronous contract: costs to the host, stream synchronised (gpu_rnnt.h:208-213, delay_transducer.h:366-372)
rnntStatus_t run(const float* acts, float* grads, const int* labels, const int* ylen, const int* xlen,
                 const float* delay_values, int V, int B, float* costs_host, void* workspace, float delay_scale, float smooth,
                 const rnntOptions& opt, bool want3) {
  if (!costs_host) return RNNT_STATUS_INVALID_VALUE;
  rnntStatus_t rc = run_fwd(acts, labels, ylen, xlen, delay_values, V, B, nullptr, workspace, delay_scale, opt);
  if (rc != RNNT_STATUS_SUCCESS) return rc;
  if (grads) {
    rc = run_bwd(acts, grads, labels, ylen, xlen, delay_values, V, B, workspace, delay_scale, smooth, 0, nullptr, 0, 1.f, opt);
    if (rc != RNNT_STATUS_SUCCESS) return rc;
  }
  hipStream_t st = (hipStream_t)opt.stream;
  Work w = carve(workspace, B, opt.maxT, opt.maxU, delay_values != nullptr);
  if (hipMemcpyAsync(costs_host, w.ll, sizeof(float) * B, hipMemcpyDeviceToHost, st) != hipSuccess)
    return RNNT_STATUS_MEMOPS_FAILED;
  if (want3 && hipMemcpyAsync(costs_host + B, w.dexp, sizeof(float) * B, hipMemcpyDeviceToHost, st) != hipSuccess)
    return RNNT_STATUS_MEMOPS_FAILED;
  if (hipStreamSynchronize(st) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
  for (int mb = 0; mb < B; ++mb) {
    costs_host[mb] = -costs_host[mb];
    if (want3) costs_host[2 * B + mb] = costs_host[mb] + delay_scale * costs_host[B + mb];
  }
  return RNNT_STATUS_SUCCESS;
}

__global__ void f64_to_f32_kernel(const double* in, float* out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
__global__ void f32_to_f64_kernel(const float* in, double* out, long lo, long hi) {   // elements [lo, hi)
  const long i = lo + (long)blockIdx.x * 256 + threadIdx.x;
  if (i < hi) out[i] = (double)in[i];
}

}  // namespace

extern "C" {

int get_warprnnt_version(void) { return 1; }

const char* rnntGetStatusString(rnntStatus_t status) {
  switch (status) {
    case RNNT_STATUS_SUCCESS: return "no error";
    case RNNT_STATUS_MEMOPS_FAILED: return "hip memcpy or memset failed";
    case RNNT_STATUS_INVALID_VALUE: return "invalid value";
    case RNNT_STATUS_EXECUTION_FAILED: return "execution failed";
    default: return "unknown error";
  }
}

rnntStatus_t compute_rnnt_loss(const float* const activations, float* gradients, const int* const flat_labels,
                               const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                               int minibatch, float* costs, void* workspace, rnntOptions options) {
  return run(activations, gradients, flat_labels, label_lengths, input_lengths, nullptr, alphabet_size, minibatch, costs,
             workspace, 0.f, 1.f, options, false);
}

rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, bool gpu, size_t* size_bytes, size_t dtype_size) {
  if (minibatch <= 0 || maxT <= 0 || maxU <= 0 || !size_bytes || (dtype_size != sizeof(float) && dtype_size != sizeof(double)) ||
      !gpu)
    return RNNT_STATUS_INVALID_VALUE;
  // the fp64 entry computes in fp32 (see compute_rnnt_loss_fp64): its workspace is the fp32 one; reporting dtype_size
  // times the float count keeps a caller that sizes by element width on the safe side
  *size_bytes = work_floats(minibatch, maxT, maxU, false) * dtype_size;
  return RNNT_STATUS_SUCCESS;
}

// rnnt.h:115-124, called by pytorch_binding/src/binding.cpp:69 / :141 for double tensors.  A CONVERTING WRAPPER: the
// lattice runs in fp32 exactly as compute_rnnt_loss (MI355X has no use for an fp64 transducer: the reference's fp64
// instantiation exists for its gradient checks - results here carry fp32 accuracy, so a finite-difference check through
// this entry needs fp32-sized steps and tolerances).  No allocation on the gradient path: the caller's fp64 gradient
// buffer (8 n bytes) doubles as the staging area - fp32 gradients in its first half, the narrowed activations in its
// second - and the gradients are widened in place, top half first (pass [ceil(hi/2), hi) writes bytes >= 4 hi, which
// only holds values already widened or the dead activations).  Costs-only calls (gradients == NULL) have no such buffer,
// and an element count that is not a multiple of 4 leaves the second half misaligned for the kernels' 16-byte accesses:
// those two cases allocate the activation staging.  Synchronous, like the reference's entry (costs land on the host).
rnntStatus_t compute_rnnt_loss_fp64(const double* const activations, double* gradients, const int* const flat_labels,
                                    const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                                    int minibatch, double* costs, void* workspace, rnntOptions options) {
  if (!activations || !costs || minibatch <= 0 || alphabet_size <= 0 || options.maxT <= 0 || options.maxU <= 0)
    return RNNT_STATUS_INVALID_VALUE;
  if (options.loc != RNNT_GPU) return RNNT_STATUS_EXECUTION_FAILED;
  hipStream_t st = (hipStream_t)options.stream;
  const long n = (long)minibatch * options.maxT * options.maxU * alphabet_size;
  float* a32 = nullptr;
  float* g32 = nullptr;
  // (the kernels take 16-byte aligned activations / gradients: the second half of the gradient buffer qualifies when n % 4 == 0)
  bool own_a32 = true;
  if (gradients) {
    if ((uintptr_t)gradients & 15) return RNNT_STATUS_INVALID_VALUE;
    g32 = (float*)gradients;
    if ((n & 3) == 0) { a32 = g32 + n; own_a32 = false; }
  }
  if (own_a32 && hipMalloc((void**)&a32, sizeof(float) * n) != hipSuccess) return RNNT_STATUS_MEMOPS_FAILED;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3(blocks), dim3(256), 0, st, activations, a32, n);
  std::vector<float> c32((size_t)minibatch);
  rnntStatus_t rc = run(a32, g32, flat_labels, label_lengths, input_lengths, nullptr, alphabet_size, minibatch, c32.data(),
                        workspace, 0.f, 1.f, options, false);
  if (rc == RNNT_STATUS_SUCCESS && gradients) {
    for (long hi = n; hi > 0;) {
      const long lo = hi == 1 ? 0 : (hi + 1) / 2;
      hipLaunchKernelGGL(f32_to_f64_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, st, g32, gradients, lo, hi);
      hi = lo;
    }
    if (hipStreamSynchronize(st) != hipSuccess) rc = RNNT_STATUS_EXECUTION_FAILED;
  }
  if (rc == RNNT_STATUS_SUCCESS)
    for (int b = 0; b < minibatch; ++b) costs[b] = (double)c32[b];
  if (own_a32) (void)hipFree(a32);
  return rc;
}

rnntStatus_t compute_rnnt_delay_loss(const float* const activations, float* gradients, const int* const flat_labels,
                                     const int* const label_lengths, const int* const input_lengths,
                                     const float* delay_values, int alphabet_size, int minibatch, float* costs,
                                     void* workspace, float delay_scale, float smooth, rnntOptions options) {
  if (!costs || !delay_values || delay_scale < -1e8f) return RNNT_STATUS_INVALID_VALUE;    // attent_entrypoint.cu:28-39
  return run(activations, gradients, flat_labels, label_lengths, input_lengths, delay_values, alphabet_size, minibatch,
             costs, workspace, delay_scale, smooth, options, true);
}

rnntStatus_t get_delay_workspace_size(int maxT, int maxU, int minibatch, bool gpu, size_t* size_bytes, size_t dtype_size) {
  if (minibatch <= 0 || maxT <= 0 || maxU <= 0 || !size_bytes || dtype_size != sizeof(float) || !gpu)
    return RNNT_STATUS_INVALID_VALUE;
  *size_bytes = work_floats(minibatch, maxT, maxU, true) * sizeof(float);
  return RNNT_STATUS_SUCCESS;
}

rnntStatus_t w2vs_rnnt_forward_async(const float* activations, const int* flat_labels, const int* label_lengths,
                                     const int* input_lengths, const float* delay_values, int alphabet_size, int minibatch,
                                     float* costs_dev, void* workspace, float delay_scale, rnntOptions options,
                                     const int* cell_index, int64_t n_cells) {
  if (!costs_dev) return RNNT_STATUS_INVALID_VALUE;
  return run_fwd(activations, flat_labels, label_lengths, input_lengths, delay_values, alphabet_size, minibatch, costs_dev,
                 workspace, delay_scale, options, cell_index, (long)n_cells);
}

rnntStatus_t w2vs_rnnt_backward_async(const float* activations, void* gradients, const int* flat_labels,
                                      const int* label_lengths, const int* input_lengths, const float* delay_values,
                                      int alphabet_size, int minibatch, void* workspace, float delay_scale, float smooth,
                                      int flags, const float* grad_scale_dev, int grad_scale_n, float grad_scale_host,
                                      rnntOptions options, const int* cell_index, int64_t n_cells) {
  return run_bwd(activations, gradients, flat_labels, label_lengths, input_lengths, delay_values, alphabet_size, minibatch,
                 workspace, delay_scale, smooth, flags, grad_scale_dev, grad_scale_n, grad_scale_host, options, cell_index,
                 (long)n_cells);
}

rnntStatus_t w2vs_ls_ce_rows(const float* logits, const int* target, void* grads, float* sums2, int64_t rows, int V, int pad,
                             float epsilon, float grad_scale, int grads_bf16, void* stream) {
  if (!logits || !target || !sums2 || rows <= 0 || V <= 1) return RNNT_STATUS_INVALID_VALUE;
  const dim3 gr((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), bl(WAVE * ROWS_PER_BLOCK);
  if (grads_bf16)
    hipLaunchKernelGGL(ls_ce_kernel<true>, gr, bl, 0, (hipStream_t)stream, logits, target, grads, sums2, (long)rows, V, pad, epsilon, grad_scale);
  else
    hipLaunchKernelGGL(ls_ce_kernel<false>, gr, bl, 0, (hipStream_t)stream, logits, target, grads, sums2, (long)rows, V, pad, epsilon, grad_scale);
  return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

rnntStatus_t w2vs_rnnt_delay_values(int kind, const int* src_lens, const int* tgt_lens, float* out, int minibatch, int maxT,
                                    int maxU, void* stream) {
  if (kind < 0 || kind > 2 || !src_lens || !tgt_lens || !out || minibatch <= 0 || maxT <= 0 || maxU <= 0)
    return RNNT_STATUS_INVALID_VALUE;
  const long n = (long)minibatch * maxT * maxU;
  hipLaunchKernelGGL(delay_values_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kind,
                     src_lens, tgt_lens, out, minibatch, maxT, maxU);
  return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

}  // extern "C"
