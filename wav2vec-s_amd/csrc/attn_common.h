// Device helpers shared by the attention kernels (attention.hip: v1, one workgroup = 128 queries walking the keys in
// lockstep; attention2.hip: v2, one workgroup = 32 queries / 32 keys with the reduction dimension split over its waves).
#pragma once
#include <type_traits>
#include "common.h"
#include "w2vs_internal.h"

namespace w2vs {

constexpr int HD = 64;         // head dim
constexpr int QB = 128;        // queries per block (4 waves x 32)
constexpr int KT = 64;         // keys per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnP {
  const bf16* q; const bf16* k; const bf16* v;  // [B, N, ld] with the head at column h*64
  bf16* o;                                      // [B, N, ldo]
  float* lse;                                   // [B, H, N] natural-log LSE of the scaled scores
  const uint8_t* kpad;                          // [B, N] 1 = padded key (may be null)
  const bf16* dout; const float* delta;         // backward
  bf16* dq; bf16* dk; bf16* dv;                 // [B, N, ld] same layout as q/k/v
  long ld, ldo, sb, sbo;                        // row stride, batch stride (elements)
  int B, H, N, Tp, m, r;
  int Nq;                                       // queries are positions 0..Nq-1 (Nq == N, or Nq <= Tp: main frames only)
  float scale; float p_drop; uint64_t seed;
  // cross mode (mq > 0; attention2.hip only; r == 0, N == Tp): the Nq queries live in their own buffer (row stride ldq,
  // batch stride sbq), query q belongs to block q / mq and sees the keys < min((q / mq + 1) * m, N) - the group-prefix
  // attention of the CAAT joiner (rain/layers/attention_transducer.py:642-715, 810-824).  lse / delta are [B, H, Ns].
  int mq; long ldq, sbq; int Ns;
  // optional keep-mask store (attention2.hip): the forward writes the attention-dropout decisions of every visible 32 x 32
  // (query tile, key tile) block as 32 dwords - dword 2i + w = the 32 query bits of key row (i&3) + 8(i>>2) + 4w, i.e. the
  // pair (2i, 2i+1) is the 64-lane mask of accumulator element i - and the two backward passes read them back instead of
  // re-hashing (the hash was 60 % of their VALU work).  [B*H][nQT][nKT][32] uint32.
  uint32_t* drop_bits; int nQT, nKT;
  // multiply-high magics of the divisors m, r, mq, H (div_magic; attention2.hip divides without a division)
  uint32_t mg_m, mg_r, mg_mq, mg_H;
  uint32_t thr16;                               // drop_threshold(p_drop) >> 16, made on the host
};

// exact x / d for 0 <= x < 65536 and 1 <= d < 65536 as one multiply-high: magic = ceil(2^32 / d), and 0 stands for d == 1.
// (error x * (magic - 2^32/d) / 2^32 < 2^-16 <= 1/d never carries the quotient over an integer.)  An integer division
// costs ~35 VALU instructions; the kernels of attention2.hip ran ten of them per workgroup before their first load.
__device__ __host__ inline uint32_t div_magic(int d) {
  return d > 1 ? (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d) : 0u;
}
__device__ __forceinline__ int fdiv(int x, uint32_t mg) { return mg ? (int)__umulhi((uint32_t)x, mg) : x; }
// Pull the launch parameters a kernel uses into SGPRs in its entry block and keep them there.  Left alone the compiler
// fetches each field of the by-value argument where it is first needed - one s_load_dword + s_waitcnt lgkmcnt(0) round
// trip per field, 13 of them in a row before the first K / V load of attention2.hip's forward (about 4 k cycles per
// workgroup, measured with s_memtime).  The "+s" makes each value opaque: a plain kernarg load is rematerialisable and the
// register allocator happily re-issues it (with its wait) further down instead of keeping the register.
#define W2VS_PIN_S(x) asm volatile("" : "+s"(x))
// pointers: the asm hides that they came from the kernel arguments (= global memory) and every access would turn into a
// flat_load - which also counts on lgkmcnt, so each LDS wait would drain the K / V prefetch.  Say it again.
#define W2VS_PIN_P(x)                                                                                                   \
  do {                                                                                                                  \
    __attribute__((address_space(1))) std::remove_pointer_t<decltype(x)>* g_;                                           \
    asm volatile("" : "=s"(g_) : "0"(x));     /* the opaque value is born a global pointer */                           \
    x = (decltype(x))g_;                                                                                                \
  } while (0)
#define W2VS_PIN_ATTNP(p)                                                                                              \
  W2VS_PIN_P((p).q); W2VS_PIN_P((p).k); W2VS_PIN_P((p).v); W2VS_PIN_P((p).o); W2VS_PIN_P((p).lse); W2VS_PIN_P((p).kpad); \
  W2VS_PIN_S((p).ld); W2VS_PIN_S((p).ldo); W2VS_PIN_S((p).sb); W2VS_PIN_S((p).sbo); W2VS_PIN_S((p).H); W2VS_PIN_S((p).N); \
  W2VS_PIN_S((p).Tp); W2VS_PIN_S((p).m); W2VS_PIN_S((p).r); W2VS_PIN_S((p).Nq); W2VS_PIN_S((p).scale);                  \
  W2VS_PIN_S((p).seed); W2VS_PIN_S((p).mq); W2VS_PIN_S((p).ldq); W2VS_PIN_S((p).sbq); W2VS_PIN_S((p).Ns);               \
  W2VS_PIN_S((p).mg_m); W2VS_PIN_S((p).mg_r); W2VS_PIN_S((p).mg_mq); W2VS_PIN_S((p).mg_H); W2VS_PIN_S((p).thr16);       \
  W2VS_PIN_P((p).drop_bits); W2VS_PIN_S((p).nQT); W2VS_PIN_S((p).nKT)
#define W2VS_PIN_ATTNP_BWD(p)                                                                                          \
  W2VS_PIN_P((p).dout); W2VS_PIN_P((p).delta); W2VS_PIN_P((p).dq); W2VS_PIN_P((p).dk); W2VS_PIN_P((p).dv)

// LDS image of a [rows][64] bf16 tile read by rows (16-B chunks): XOR swizzle as in gemm.hip
__device__ __forceinline__ int kswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
// LDS image of a [rows][64] bf16 tile read through tr reads (8-B pieces): flip the 64-B half on
// rows 2,3 (mod 4) so the four rows of a tr block sit on distinct banks
__device__ __forceinline__ int vswz(int row, int col) { return row * 64 + (col ^ (((row >> 1) & 1) << 5)); }
// row * ld as a 24 x 24-bit product (attn2_ok: rows and strides below 2^24, products below 2^32): one full-rate multiply
// and a 64-bit add per row address instead of the five-instruction 64-bit multiply the `long` arithmetic compiles to -
// the loops form five such addresses per sub-tile
__device__ __forceinline__ const bf16* row_at(const bf16* base, int row, uint32_t ld) { return base + __umul24((uint32_t)row, ld); }
// The same tile when it is WRITTEN from MFMA operand registers (lane = row, one 16-B chunk per instruction: 16 rows of one
// chunk column per LDS pass) and read through tr reads.  vswz serves the tr reads but puts those 16 rows on 4 bank groups
// (PMC: 40-47 % of the LDS cycles of the two backward kernels were bank conflicts).  XOR the chunk with a value that is a
// bijection of (row >> 1) & 7 - conflict-free row writes - whose bit 2 separates rows 2,3 (mod 4) from rows 0,1 - the
// four rows of a tr block stay on distinct banks.
__device__ __forceinline__ int uswz(int row, int col) {
  const int s = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  return row * 64 + ((((col >> 3) ^ s) << 3) | (col & 7));
}

__device__ __forceinline__ s16x4 ds_tr16(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ bf16x8 tr_pair(const bf16* lo, const bf16* hi) {
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = ds_tr16(lo);
  u.h[1] = ds_tr16(hi);
  return u.v;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f2bf(a[8 * s + j]);
  return r;
}
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// Scaled + masked scores of one 32-key sub-tile, branch free: the 16 key biases of this lane's accumulator rows
// come in as four 16-byte LDS reads, the mask compares compile-time row offsets against lane-relative limits.
// (A per-element `ok ? S*c + kb[kl] : -inf` made the compiler branch around sixteen dependent ds_read_b32.)
__device__ __forceinline__ void masked_scores(f32x16& S, float c, const float* kb32, int hh, int lim_r, int clo_r, int chi_r) {
  f32x4 kbv[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) kbv[g4] = *(const f32x4*)(kb32 + 8 * g4 + 4 * hh);
  lim_r -= 4 * hh; clo_r -= 4 * hh; chi_r -= 4 * hh;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rc = (i & 3) + 8 * (i >> 2);
    const bool ok = (rc < lim_r) | ((rc >= clo_r) & (rc < chi_r));
    const float sv = fmaf(S[i], c, kbv[i >> 2][i & 3]);
    S[i] = ok ? sv : -INFINITY;
  }
}

struct QLimits { int lim, clo, chi; };
__device__ __forceinline__ QLimits q_limits(int q, int Tp, int m, int r, int N, int mq = 0) {
  QLimits L;
  if (mq > 0) { L.lim = min((q / mq + 1) * m, Tp); L.clo = N; L.chi = N; return L; }
  int bq = (q < Tp) ? q / m : (r > 0 ? (q - Tp) / r : 0);
  L.lim = min((bq + 1) * m, Tp);
  L.clo = r > 0 ? Tp + bq * r : N;
  L.chi = r > 0 ? min(Tp + (bq + 1) * r, N) : N;
  return L;
}
// the same through the magics (q < 65536)
__device__ __forceinline__ QLimits q_limits_mg(int q, const AttnP& p) {
  QLimits L;
  if (p.mq > 0) { L.lim = min((fdiv(q, p.mg_mq) + 1) * p.m, p.Tp); L.clo = p.N; L.chi = p.N; return L; }
  const int bq = (q < p.Tp) ? fdiv(q, p.mg_m) : (p.r > 0 ? fdiv(q - p.Tp, p.mg_r) : 0);
  L.lim = min((bq + 1) * p.m, p.Tp);
  L.clo = p.r > 0 ? p.Tp + bq * p.r : p.N;
  L.chi = p.r > 0 ? min(p.Tp + (bq + 1) * p.r, p.N) : p.N;
  return L;
}
// key range a set of queries [q0, q1] can touch: main keys [0, mlim), copies [clo, chi)
__device__ __forceinline__ void tile_ranges(int q0, int q1, int Tp, int m, int r, int N, int& mlim, int& clo, int& chi) {
  int bmin, bmax;
  if (q1 < Tp) { bmin = q0 / m; bmax = q1 / m; }
  else if (q0 >= Tp) { bmin = r > 0 ? (q0 - Tp) / r : 0; bmax = r > 0 ? (q1 - Tp) / r : 0; }
  else { bmin = 0; bmax = (Tp - 1) / m; }
  mlim = min((bmax + 1) * m, Tp);
  clo = r > 0 ? Tp + bmin * r : N;
  chi = r > 0 ? min(Tp + (bmax + 1) * r, N) : N;
}
// attention-dropout keep decision: 32-bit element index ((b*H+h)*N + q)*N + key, two-round
// multiply-xorshift mix keyed by the 64-bit seed.  Cheaper than common.h's hash32 because it runs
// once per score inside three kernels (fwd, dQ pass, dK/dV pass) that must agree bit for bit.
__device__ __forceinline__ float keep_scale(uint32_t s0, uint32_t s1, uint32_t idx, uint32_t thr, float inv_keep) {
  uint32_t x = idx * 0x9E3779B1u ^ s0;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= s1;
  x ^= x >> 16;
  return x >= thr ? inv_keep : 0.f;
}
// Two keep decisions per hash word: element (q, key) uses word ((b*H+h)*N + q)*ceil(N/2) + key/2 and
// its low (key even) or high (key odd) 16 bits, compared against a 16-bit threshold
// (p_eff = round(p*65536)/65536, e.g. 0.099991 for p = 0.1; inv_keep uses p_eff).
// The word is a Weyl step (index * golden ratio + seed) through one xorshift-multiply-xorshift round; callers
// pass the premultiplied index so that neighbouring words cost an add, not a quarter-rate v_mul_lo_u32.
// (Three multiply rounds per word made the dropout a quarter of the attention kernels' VALU work; on 669 k
// decisions this form shows the same keep rate, adjacent-element and field-to-field correlations < 0.004.)
constexpr uint32_t HASH_K = 0x9E3779B1u;
// Round 3: no 32-bit multiply at all.  v_mul_lo_u32 issues at a quarter of the VALU rate (4 of this word's 10 issue slots);
// the 24-bit v_mad_u32_u24 is full rate.  x0 = idx * K + seed is already well spread (K = 2^32 / phi, odd); its high half is
// folded into the low 24 bits, those are multiplied by an odd 24-bit constant and x0 itself is added back (the word stays a
// function of all 32 bits of x0: no systematic collisions), one xor-shift finishes it: 6 full-rate instructions per word of
// two decisions.  Checked on 2 x 419 k decisions per seed (numpy mirror, three seeds): keep rate within 6e-4 of 1 - p;
// field-to-field, adjacent-column / -row / -diagonal and two-apart correlations all < 0.005 (the noise level of the
// sample, as for the 32-bit form it replaces); 64-bin chi-square of the fields 65 - 84 (expected 63 +- 11).
// Round 4: the second seed word enters NON-linearly.  Round 3 folded the whole 64-bit seed into the additive offset of the
// index (x0 = idx * K + f(s0, s1)): with K odd, the masks of two seeds were the same 2^32-long sequence read at two offsets,
// and the windows of two layers (2e7 words per layer and step) overlapped in about half of the steps - shifted IDENTICAL
// attention-dropout masks.  Now s0 alone shifts the Weyl sequence and a mix of s1 is XOR-ed into the word before the 24-bit
// multiply, whose carries make the result depend on it non-linearly: two seeds that land on the same x0 still draw different
// words.  One more full-rate op per word (v_xor3 where the compiler fuses it).  tests/test_host_cpu.py mirrors this function
// in numpy and checks keep rate, neighbour correlations AND cross-seed correlations at aligned offsets;
// tests/test_kernels_gpu.py ties the mirror to the kernels' own decisions (drop_bits).
__device__ __forceinline__ uint32_t seed_mix(uint32_t s1) {              // loop-invariant scalar work
  uint32_t m = (s1 ^ (s1 >> 15)) * 0x85EBCA77u;
  return m ^ (m >> 13);
}
__device__ __forceinline__ uint32_t pair_hash_pm(uint32_t s0, uint32_t s1, uint32_t idx_times_k) {
  const uint32_t x0 = idx_times_k + s0;
  const uint32_t x = x0 ^ (x0 >> 16) ^ seed_mix(s1);
  uint32_t h = __umul24(x, 0xB5352Du) + x0;                           // v_mad_u32_u24
  h ^= h >> 15;
  return h;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// attention2.hip: the same contract with 32-row workgroups and the long dimension split over their waves
bool attn2_ok(const AttnP& p);
int attn2_fwd(const AttnP& p, hipStream_t st);
int attn2_bwd(const AttnP& p, hipStream_t st);

}  // namespace w2vs
