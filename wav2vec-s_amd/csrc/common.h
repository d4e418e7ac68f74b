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

// erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution): one v_rcp, one
// v_exp and a 5-term Horner chain instead of libm's branchy erff.  Hundreds of millions of GELUs per
// step sit in GEMM / conv epilogues, so this is a first-order cost.  Returns erf(x) and exp(-x^2).
__device__ __forceinline__ float fast_erf(float x, float& ex2) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  ex2 = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float r = 1.0f - poly * t * ex2;
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_exact(float x) {  // 0.5 x (1 + erf(x / sqrt 2))
  float e;
  return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float gelu_grad(float x) {  // Phi(x) + x phi(x); exp(-x^2/2) comes with the erf
  float e;
  const float er = fast_erf(x * 0.70710678118654752440f, e);
  return 0.5f * (1.0f + er) + x * 0.3989422804014327f * e;
}

}  // namespace w2vs
