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
//                   log-probabilities of diagonal n are contiguous in memory and are loaded one diagonal ahead.
//   grad_kernel     one wave per row: every per-cell scalar is computed once, then the row is streamed with 16-byte
//                   loads / stores (1-2 exp per element); rows outside a sample's T x U are written as zeros here, which
//                   replaces the reference's memset pass over the whole gradient.
// HBM-bound: rows_kernel reads 4 V bytes per valid row, grad_kernel reads 4 V and writes 4 V bytes per row.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/w2vs_rnnt.h"

namespace {

constexpr int WAVE = 64;
constexpr int ROWS_PER_BLOCK = 4;     // 256 threads, one wave per row

struct Dims {
  int B, maxT, maxU, V, blank, D;     // D = maxT + maxU - 1 anti-diagonals
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

__device__ __forceinline__ float lse(float a, float b) {                         // rnnt_helper.h:17-26
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  return a > b ? log1pf(expf(b - a)) + a : log1pf(expf(a - b)) + b;
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
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const long rows = (long)d.B * d.maxT * d.maxU;
  if (row >= rows) return;
  const int u = (int)(row % d.maxU);
  const long bt = row / d.maxU;
  const int t = (int)(bt % d.maxT), b = (int)(bt / d.maxT);
  const int T = clampi(xlen[b], 1, d.maxT), U = clampi(ylen[b] + 1, 1, d.maxU);   // lengths past the tensor would read out of bounds
  if (t >= T || u >= U) return;                       // never read: the lattice and the gradient skip these cells
  const float* x = acts + row * d.V;
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
  if (blockIdx.y == 0) {
    // ---- alpha(t,u) and alpha_delay(t,u): gpu_rnnt_kernel.h:12-51, 54-100
    float* out = w.alpha + base;
    float* outd = DELAY ? w.adel + base : nullptr;
    float self = 0.f, selfd = 0.f;                     // alpha(t-1, u), alpha_delay(t-1, u)
    if (u == 0) {
      out[0] = 0.f;
      if (DELAY) outd[0] = 0.f;
    }
    shv[u] = 0.f;                                      // buffer 0 holds diagonal 0: only (0,0) = 0 matters
    shd[u] = 0.f;
    // log-probabilities of diagonal n-1 feed step n: blank at column u (cell (t-1,u)), label at column u-1 (cell (t,u-1))
    float pb = (u < d.maxU && N > 1) ? lpb[u] : 0.f;
    float pl = (u >= 1 && u <= d.maxU && N > 1) ? lpl[u - 1] : 0.f;
    __syncthreads();
    for (int n = 1; n < N; ++n) {
      const int t = n - u;
      const float* prv = shv + ((n - 1) & 1) * W;
      const float* prd = shd + ((n - 1) & 1) * W;
      const bool valid = u < U && t >= 0 && t < T;
      float npb = 0.f, npl = 0.f;                      // next diagonal's inputs, loaded before this step's math
      if (n + 1 < N) {
        if (u < d.maxU) npb = lpb[(long)n * d.maxU + u];
        if (u >= 1 && u <= d.maxU) npl = lpl[(long)n * d.maxU + u - 1];
      }
      float dval = 0.f;
      if (DELAY && valid && u > 0) dval = dv[(long)t * d.maxU + u];
      float a = 0.f, ad = 0.f;
      if (valid) {
        if (u == 0) {
          a = self + pb;                               // t > 0 here (n >= 1)
          ad = 0.f;
        } else {
          const float left = prv[u - 1], leftd = prd[u - 1];
          if (t == 0) {
            a = left + pl;
            ad = leftd + dval;
          } else {
            const float no_emit = self + pb, emit = left + pl;
            a = lse(emit, no_emit);
            // the two path weights exp(no_emit - a), exp(emit - a) sum to 1: written as a convex combination the rounding of
            // a (|a| ~ 100s in fp32) only scales the DIFFERENCE of the two delays, not their magnitude
            if (DELAY) ad = selfd + expf(emit - a) * (leftd + dval - selfd);
          }
        }
        out[(long)n * d.maxU + u] = a;
        if (DELAY) outd[(long)n * d.maxU + u] = ad;
        self = a;
        selfd = ad;
      }
      shv[(n & 1) * W + u] = a;
      shd[(n & 1) * W + u] = ad;
      pb = npb;
      pl = npl;
      __syncthreads();
    }
    if (u == U - 1) {                                  // this thread holds alpha(T-1, U-1) in `self`
      w.ll[b] = self + lpb[(long)(N - 1) * d.maxU + (U - 1)];
      if (DELAY) w.dexp[b] = selfd;
    }
  } else {
    // ---- beta(t,u) and beta_delay(t,u): gpu_rnnt_kernel.h:126-163, 166-213
    float* out = w.beta + base;
    float* outd = DELAY ? w.bdel + base : nullptr;
    float self = 0.f, selfd = 0.f;                     // beta(t+1, u), beta_delay(t+1, u)
    shv[u] = 0.f; shd[u] = 0.f; shv[W + u] = 0.f; shd[W + u] = 0.f;
    float pb = 0.f, pl = 0.f;                          // log-probabilities of THIS diagonal's cell (t, u)
    if (u < d.maxU) {
      pb = lpb[(long)(N - 1) * d.maxU + u];
      pl = lpl[(long)(N - 1) * d.maxU + u];
    }
    __syncthreads();
    for (int n = N - 1; n >= 0; --n) {
      const int t = n - u;
      const float* prv = shv + ((n + 1) & 1) * W;
      const float* prd = shd + ((n + 1) & 1) * W;
      const bool valid = u < U && t >= 0 && t < T;
      float npb = 0.f, npl = 0.f;
      if (n > 0 && u < d.maxU) {
        npb = lpb[(long)(n - 1) * d.maxU + u];
        npl = lpl[(long)(n - 1) * d.maxU + u];
      }
      float dval = 0.f;
      if (DELAY && valid && u < U - 1) dval = dv[(long)t * d.maxU + u];
      float bb = 0.f, bd = 0.f;
      if (valid) {
        if (u == U - 1) {
          bb = (t == T - 1) ? pb : self + pb;
          bd = (t == T - 1) ? 0.f : selfd;
        } else {
          const float right = prv[u + 1], rightd = prd[u + 1];
          if (t == T - 1) {
            bb = right + pl;
            bd = rightd + dval;
          } else {
            const float no_emit = self + pb, emit = right + pl;
            bb = lse(emit, no_emit);
            if (DELAY) bd = selfd + expf(emit - bb) * (rightd + dval - selfd);
          }
        }
        out[(long)n * d.maxU + u] = bb;
        if (DELAY) outd[(long)n * d.maxU + u] = bd;
        self = bb;
        selfd = bd;
      }
      shv[(n & 1) * W + u] = bb;
      shd[(n & 1) * W + u] = bd;
      pb = npb;
      pl = npl;
      __syncthreads();
    }
    if (u == 0) {
      w.llb[b] = self;
      if (DELAY) w.dexpb[b] = selfd;
    }
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

template <bool DELAY>
__global__ __launch_bounds__(WAVE* ROWS_PER_BLOCK) void grad_kernel(const float* __restrict__ acts, float* __restrict__ grads,
                                                                    const int* __restrict__ labels,
                                                                    const int* __restrict__ xlen, const int* __restrict__ ylen,
                                                                    const float* __restrict__ delay_values, Work w, Dims d,
                                                                    float delay_scale, float smooth, int consistent_index,
                                                                    const float* __restrict__ up_dev, int up_n, float up_host) {
  const int lane = threadIdx.x & (WAVE - 1);
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const long rows = (long)d.B * d.maxT * d.maxU;
  if (row >= rows) return;
  const int u = (int)(row % d.maxU);
  const long bt = row / d.maxU;
  const int t = (int)(bt % d.maxT), b = (int)(bt / d.maxT);
  const int T = clampi(xlen[b], 1, d.maxT), U = clampi(ylen[b] + 1, 1, d.maxU);   // lengths past the tensor would read out of bounds
  float* g = grads + row * d.V;
  const bool vec = (d.V & 3) == 0;
  if (t >= T || u >= U) {                              // the reference zeroes the whole tensor first
    if (vec) {
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
  const float* x = acts + row * d.V;
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
      g4[i] = o;
    }
  } else {
    for (int i = lane; i < d.V; i += WAVE) g[i] = grad_elem(k, x[i], i);
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
                     int B, float* costs_dev, void* workspace, float delay_scale, const rnntOptions& opt) {
  const rnntStatus_t rc = check(acts, labels, ylen, xlen, workspace, V, B, opt);
  if (rc != RNNT_STATUS_SUCCESS) return rc;
  const bool delay = delay_values != nullptr;
  hipStream_t st = (hipStream_t)opt.stream;
  Dims d{B, opt.maxT, opt.maxU, V, opt.blank_label, opt.maxT + opt.maxU - 1};
  Work w = carve(workspace, B, opt.maxT, opt.maxU, delay);
  const long rows = (long)B * opt.maxT * opt.maxU;
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
rnntStatus_t run_bwd(const float* acts, float* grads, const int* labels, const int* ylen, const int* xlen,
                     const float* delay_values, int V, int B, void* workspace, float delay_scale, float smooth, int flags,
                     const float* up_dev, int up_n, float up_host, const rnntOptions& opt) {
  const rnntStatus_t rc = check(acts, labels, ylen, xlen, workspace, V, B, opt);
  if (rc != RNNT_STATUS_SUCCESS) return rc;
  if (!grads || ((uintptr_t)grads & 15) || (up_n != 0 && up_n != 1 && up_n != B) || (up_n != 0 && !up_dev))
    return RNNT_STATUS_INVALID_VALUE;
  const bool delay = delay_values != nullptr;
  hipStream_t st = (hipStream_t)opt.stream;
  Dims d{B, opt.maxT, opt.maxU, V, opt.blank_label, opt.maxT + opt.maxU - 1};
  Work w = carve(workspace, B, opt.maxT, opt.maxU, delay);
  const long rows = (long)B * opt.maxT * opt.maxU;
  const unsigned row_blocks = (unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  if (delay)
    hipLaunchKernelGGL(grad_kernel<true>, dim3(row_blocks), dim3(WAVE * ROWS_PER_BLOCK), 0, st, acts, grads, labels, xlen,
                       ylen, delay_values, w, d, delay_scale, smooth, flags & 1, up_dev, up_n, up_host);
  else
    hipLaunchKernelGGL(grad_kernel<false>, dim3(row_blocks), dim3(WAVE * ROWS_PER_BLOCK), 0, st, acts, grads, labels, xlen,
                       ylen, delay_values, w, d, 0.f, smooth, 0, up_dev, up_n, up_host);
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
  if (minibatch <= 0 || maxT <= 0 || maxU <= 0 || !size_bytes || dtype_size != sizeof(float) || !gpu)
    return RNNT_STATUS_INVALID_VALUE;
  *size_bytes = work_floats(minibatch, maxT, maxU, false) * sizeof(float);
  return RNNT_STATUS_SUCCESS;
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
                                     float* costs_dev, void* workspace, float delay_scale, rnntOptions options) {
  if (!costs_dev) return RNNT_STATUS_INVALID_VALUE;
  return run_fwd(activations, flat_labels, label_lengths, input_lengths, delay_values, alphabet_size, minibatch, costs_dev,
                 workspace, delay_scale, options);
}

rnntStatus_t w2vs_rnnt_backward_async(const float* activations, float* gradients, const int* flat_labels,
                                      const int* label_lengths, const int* input_lengths, const float* delay_values,
                                      int alphabet_size, int minibatch, void* workspace, float delay_scale, float smooth,
                                      int flags, const float* grad_scale_dev, int grad_scale_n, float grad_scale_host,
                                      rnntOptions options) {
  return run_bwd(activations, gradients, flat_labels, label_lengths, input_lengths, delay_values, alphabet_size, minibatch,
                 workspace, delay_scale, smooth, flags, grad_scale_dev, grad_scale_n, grad_scale_host, options);
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
