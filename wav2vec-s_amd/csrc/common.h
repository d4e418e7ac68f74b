// Device-side helpers shared by every kernel of libw2vs (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace w2vs {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe

// ---- buffer resources: hardware bounds checking gives zero-fill loads and dropped stores ----
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}

// ---- wave / block reductions -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based RNG for dropout / gumbel noise ----------------------------------------
// One 32-bit hash per (seed, element index).  The same (seed, index) is re-evaluated in
// the backward kernels, so no mask is ever stored.
__device__ __forceinline__ uint32_t hash32(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)(idx >> 32) * 0x85EBCA77u;
  x ^= (uint32_t)seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  x += (uint32_t)(seed >> 32);
  x ^= x >> 15; x *= 0x2C1B3C6Du;
  x ^= x >> 12; x *= 0x297A2D39u;
  x ^= x >> 15;
  return x;
}
// keep-mask threshold for drop probability p: keep iff hash >= thr
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}
// Chunk-granular dropout for the row kernels: the 8 consecutive elements of chunk `cidx` (= element index / 8)
// take their keep decisions from four hash words, two 16-bit fields each (p_eff = round-down(p * 65536) / 65536,
// e.g. 0.099991 for p = 0.1; inv_keep is 1 / (1 - p_eff), so E[mask * inv_keep] = 1 exactly).  Two 32-bit
// multiplies per word instead of hash32's six per element: the full-rate-equivalent VALU cost of the
// mask drops ~5x, which is what bounds the LayerNorm kernels (v_mul_lo_u32 issues at quarter rate).
struct Drop {
  uint32_t s0, s1, thr16;
  float inv_keep;
};
__host__ __device__ __forceinline__ Drop make_drop(float p, uint64_t seed) {
  Drop d;
  d.s0 = (uint32_t)seed; d.s1 = (uint32_t)(seed >> 32);
  d.thr16 = drop_threshold(p) >> 16;
  d.inv_keep = d.thr16 > 0 ? 65536.f / (65536.f - (float)d.thr16) : 1.f;
  return d;
}
__device__ __forceinline__ void drop8(const Drop& d, uint32_t cidx, float (&m)[8]) {
  const uint32_t base = cidx * (4u * 0x9E3779B1u);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t x = (base + (uint32_t)j * 0x9E3779B1u) ^ d.s0;
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x ^= d.s1; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    m[2 * j] = (x & 0xFFFFu) >= d.thr16 ? d.inv_keep : 0.f;
    m[2 * j + 1] = (x >> 16) >= d.thr16 ? d.inv_keep : 0.f;
  }
}
__device__ __forceinline__ float u01(uint32_t h) {  // (0,1]
  return ((float)(h >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// GELU(x) = x Phi(x) (exact-erf form, fs/modules/gelu.py:24-25 = torch.nn.functional.gelu) through the normal TAIL
//   q(a) = 1 - Phi(a) = 0.5 erfc(a / sqrt 2),  a = |x|:   Phi(x) = x >= 0 ? 1 - q : q,   gelu(x) = max(x, 0) - a q(a).
// q is ONE v_exp_f32 of a degree-6 polynomial in a (minimax fit of log2 q on [0, 6], constant term -1 = log2 q(0)): its error
// is RELATIVE (< 5e-5 of q), so the small negative-side values x q keep their leading digits, and gelu is within 7.4e-6
// absolute / 5e-5 relative of the erf form everywhere, gelu' within 2e-5 (bf16 resolves 4e-3) - tools/fit_gelu_tail.py
// re-derives the coefficients and these bounds in float32 arithmetic.  Beyond a = 6, q < 1e-9 and a is clamped.
// Round 5: this replaces Abramowitz-Stegun 7.1.26 (v_rcp + v_exp + a 5-term chain): hundreds of millions of GELUs per step
// sit in conv / GEMM epilogues that in-kernel stamps show VALU-bound (profiles/round5_nt_anatomy_probe.txt); the forward-only
// form drops from 15 vector instructions with two transcendentals to 10 with one.
__device__ __forceinline__ float normal_tail(float ax) {   // ax = |x| >= 0
  const float a = fminf(ax, 6.0f);
  float p = fmaf(2.3433251092e-05f, a, -6.1973935318e-04f);
  p = fmaf(p, a, 7.2603524696e-03f);
  p = fmaf(p, a, -5.1418609203e-02f);
  p = fmaf(p, a, -4.6086354093e-01f);
  p = fmaf(p, a, -1.1504803413f);
  return __builtin_amdgcn_exp2f(fmaf(p, a, -1.0f));
}
__device__ __forceinline__ float gelu_exact(float x) {
  const float ax = fabsf(x);
  return fmaf(-ax, normal_tail(ax), fmaxf(x, 0.f));
}
// Phi(x) and the density term exp(-x^2 / 2) (one more v_exp_f32)
__device__ __forceinline__ float normal_cdf(float x, float& ex2) {
  const float q = normal_tail(fabsf(x));
  ex2 = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  return x >= 0.f ? 1.0f - q : q;
}
// Two values at a time, the same arithmetic operation for operation (results are bit-identical to the scalar forms): written
// on 2-vectors so that the compiler emits v_pk_fma_f32 / v_pk_mul_f32 - one issue slot per PAIR.  Left to itself hipcc keeps the
// Horner chain scalar (v_fmaak_f32 with its literal).  Round 5: conv layer 0's forward uses the pairs (110 -> 105 us alone, 103.7 ->
// 100.8 in the step); in the GEMM epilogues they cut the fc1 forward's epilogue from 15.7 to 12.4 us in the probe and NOTHING in
// the step (the packed constants cost the 8-phase kernel 33 more spilled SGPRs: 53.9 -> 54.8 us per launch) - not used there.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 normal_tail2(f32x2 ax) {
  const f32x2 a = {fminf(ax.x, 6.0f), fminf(ax.y, 6.0f)};
  f32x2 p = pk_fma(splat2(2.3433251092e-05f), a, splat2(-6.1973935318e-04f));
  p = pk_fma(p, a, splat2(7.2603524696e-03f));
  p = pk_fma(p, a, splat2(-5.1418609203e-02f));
  p = pk_fma(p, a, splat2(-4.6086354093e-01f));
  p = pk_fma(p, a, splat2(-1.1504803413f));
  p = pk_fma(p, a, splat2(-1.0f));
  return f32x2{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
}
__device__ __forceinline__ f32x2 gelu_exact2(f32x2 x) {
  const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
  return pk_fma(-ax, normal_tail2(ax), f32x2{fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)});
}
// gelu(x) and gelu'(x) of a pair from one evaluation of the tail each
__device__ __forceinline__ void gelu_pair2(f32x2 x, f32x2& gv, f32x2& dv) {
  const f32x2 q = normal_tail2(f32x2{fabsf(x.x), fabsf(x.y)});
  const f32x2 t = (splat2(-0.72134752044448170368f) * x) * x;
  const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  const f32x2 cdf = {x.x >= 0.f ? 1.0f - q.x : q.x, x.y >= 0.f ? 1.0f - q.y : q.y};
  gv = x * cdf;
  dv = pk_fma(x * splat2(0.3989422804014327f), e, cdf);
}
__device__ __forceinline__ float gelu_grad(float x) {  // Phi(x) + x phi(x)
  float e;
  const float cdf = normal_cdf(x, e);
  return fmaf(x * 0.3989422804014327f, e, cdf);
}

}  // namespace w2vs
