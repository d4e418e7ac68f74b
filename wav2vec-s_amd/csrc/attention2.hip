// Block-causal streaming attention, second decomposition (gfx950, head_dim 64).  Same mathematics, masks, dropout words
// and argument contract as attention.hip; what changes is who does what.
//
// Why: under gen_block_attn_mask (fs/models/wav2vec/wav2vec_S.py:444-489) a query of block b sees (b+1)*m main keys, so
// the work of a 32-query tile grows linearly along the sequence.  In attention.hip one workgroup owns 128 queries and
// walks the visible keys in lockstep: every launch lasts as long as its longest workgroup (measured on MI355X:
// 61 us for the 236 k visible pairs of the cfgB shape - and 62 us for the 669 k pairs of the same shape unmasked).
//
// Here the unit is small and the long dimension is split:
//   forward / dQ pass : workgroup = 32 queries of one (batch, head); its 4 waves take the visible 32-key sub-tiles
//                       round-robin, each with its own online-softmax state, and merge (m, l, O) / sum dQ through LDS;
//   dK/dV pass        : workgroup = 32 keys; its 4 waves take the visible 32-query sub-tiles round-robin, dK / dV summed
//                       through LDS.
// Workgroups are launched longest-first (host-made record table, which also carries each tile's sub-tile list), so the tail
// of the launch is made of the short ones.  A wave never waits for another inside its loop - one barrier after the
// workgroup's shared operand tile is parked, one before the final merge.
//
// Data path (the second thing this file is about): EVERY operand row travels global -> registers -> an LDS tile -> MFMA
// operand, and every global load is row-contiguous (8 lanes x 16 B per row, 8 rows per instruction).  The first version
// loaded K / V (Q / dO) fragments straight into the MFMA operand layout - one row per lane, 32 rows x 32 B per instruction -
// which costs the vector-memory address path four lines per quad of lanes instead of one; that, not VALU work, occupancy or
// prologue length, was what those kernels waited on (a timing-only build with row-contiguous addresses ran the
// backward 13 % faster, the reworked kernels 127 -> 100 us).  Per-workgroup tiles (Q; Q, dO, O; K, V) are fetched ONCE by the
// four waves together; per-sub-tile operands go through wave-private 4 KB tiles whose swizzle (uswz) is conflict-free for
// the row writes, the ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads.  One register set: the next
// sub-tile's loads are issued as soon as the current tiles sit in LDS.
#include <algorithm>
#include <vector>
#include "attn_common.h"

// Timing-only ablations of the forward kernel (make ablation ABL="-DW2VS_ATTN_ABL=n"; never in libw2vs.so; results WRONG):
//   bit 0: no K / V global loads inside the loop (the first sub-tile's registers are reused)
//   bit 1: no MFMAs            bit 2: no LDS round trip of K / V (fragments are read from whatever the tiles hold)
//   bit 3: the loop body never runs (prologue + merge only)
#if defined(W2VS_ABLATION) && defined(W2VS_ATTN_ABL)
#define W2VS_AABL W2VS_ATTN_ABL
#else
#define W2VS_AABL 0
#endif

namespace w2vs {
namespace {

constexpr int MAXT2 = 256;   // 32-row tiles per sequence (N <= 8192; longer sequences take attention.hip)

// One record per workgroup row of the launch, longest first, made on the host (make_table): the tile and its list of
// visible sub-tiles.  In-kernel this was ~10 integer divisions per workgroup before the first load could be issued -
// measured with s_memtime at the cfgB shape: 4.7 k of a workgroup's 18 k cycles, on SIMDs shared with other waves' loops.
//   bits 0-9 tile | 10-19 nT | 20-29 nM | 30-39 rc0 | 40-49 aux (fwd/dq: key tiles every query sees in full; dkv: m_lo)
struct Attn2P {
  AttnP a;
  int ntiles;                 // query tiles (fwd, dq) or key tiles (dkv)
#ifdef W2VS_ABLATION
  // tuning build only: in-kernel time stamps (s_memrealtime, 100 MHz), 8 words per wave: 0 entry, 1 loop entry, 2 loop exit,
  // 3 behind the merge barrier, 4 end, 5 list length, 6 HW_ID, 7 XCC_ID (tools/attn_anatomy_probe.py); null = off
  unsigned long long* stamps;
#endif
  uint64_t rec[MAXT2];
};
#ifdef W2VS_ABLATION
#define W2VS_ASTAMP_V(slot, v)                                                                                         \
  do {                                                                                                                 \
    if (pp.stamps && lane == 0) pp.stamps[(((long)blockIdx.y * gridDim.x + blockIdx.x) * NW + wid) * 8 + (slot)] = (v);   \
  } while (0)
#define W2VS_ASTAMP(slot) W2VS_ASTAMP_V(slot, __builtin_amdgcn_s_memrealtime())
#define W2VS_ASTAMP_IDS(n)                                                                                             \
  do {                                                                                                                 \
    W2VS_ASTAMP_V(5, (unsigned long long)(n));                                                                         \
    W2VS_ASTAMP_V(6, (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));   /* HW_ID */    \
    W2VS_ASTAMP_V(7, (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));  /* XCC_ID */   \
  } while (0)
#else
#define W2VS_ASTAMP(slot) do { } while (0)
#define W2VS_ASTAMP_IDS(n) do { } while (0)
#endif

struct SubList { int nM, rc0, nT; };   // sub-tiles [0, nM) then rc0, rc0 + 1, ... : nT in all

// 32-key sub-tiles the queries [q0, q1] can see
__device__ __host__ inline void key_ranges_h(int q0, int q1, int Tp, int m, int r, int N, int mq, int& mlim, int& clo, int& chi, int& full) {
  int bmin, bmax;
  bool mixed = false;
  if (mq > 0) { bmin = q0 / mq; bmax = q1 / mq; r = 0; }
  else if (q1 < Tp) { bmin = q0 / m; bmax = q1 / m; }
  else if (q0 >= Tp) { bmin = r > 0 ? (q0 - Tp) / r : 0; bmax = r > 0 ? (q1 - Tp) / r : 0; }
  else { bmin = 0; bmax = (Tp - 1) / m; mixed = true; }
  mlim = std::min((bmax + 1) * m, Tp);
  clo = r > 0 ? Tp + bmin * r : N;
  chi = r > 0 ? std::min(Tp + (bmax + 1) * r, N) : N;
  full = std::min((bmin + 1) * m, Tp);      // every query of the range sees the main keys below this
  (void)mixed;
}
__device__ __host__ inline SubList key_list(int q0, int q1, int Tp, int m, int r, int N, int mq, int& full) {
  int mlim, clo, chi;
  key_ranges_h(q0, q1, Tp, m, r, N, mq, mlim, clo, chi, full);
  SubList t;
  t.nM = (mlim + 31) >> 5;
  int lo = std::max(clo >> 5, t.nM), hi = (chi + 31) >> 5;
  if (chi <= clo || hi < lo) hi = lo;
  t.rc0 = lo;
  t.nT = t.nM + (hi - lo);
  return t;
}
// 32-query sub-tiles that can see the keys [k0, k1]; queries are 0..Nq-1 (Nq == N or Nq <= Tp)
__device__ __host__ inline SubList query_list(int k0, int k1, int Tp, int m, int r, int N, int Nq, int mq) {
  int mq0, mq1, rq0, rq1;
  if (mq > 0) { mq0 = (k0 / m) * mq; mq1 = Nq; rq0 = rq1 = Nq; }
  else if (k1 < Tp) { const int bmin = k0 / m; mq0 = bmin * m; mq1 = Tp; rq0 = r > 0 ? Tp + bmin * r : N; rq1 = N; }
  else if (k0 >= Tp) {
    const int bmin = (k0 - Tp) / std::max(r, 1), bmax = (k1 - Tp) / std::max(r, 1);
    mq0 = bmin * m; mq1 = std::min((bmax + 1) * m, Tp); rq0 = Tp + bmin * r; rq1 = std::min(Tp + (bmax + 1) * r, N);
  } else { mq0 = 0; mq1 = Tp; rq0 = Tp; rq1 = N; }
  mq1 = std::min(mq1, Nq);
  rq1 = std::min(rq1, Nq);
  SubList t;
  const int m_lo = mq0 >> 5, m_hi = mq1 > mq0 ? (mq1 + 31) >> 5 : m_lo;
  t.nM = m_hi - m_lo;                      // main sub-tiles m_lo .. m_hi-1: encoded through rc0 below
  int lo = std::max(rq0 >> 5, m_hi), hi = (rq1 + 31) >> 5;
  if (rq1 <= rq0 || hi < lo) hi = lo;
  t.rc0 = lo;
  t.nT = t.nM + (hi - lo);
  // main part starts at m_lo: callers map position p -> (p < nM ? m_lo + p : rc0 + p - nM); m_lo travels in the sign-free
  // upper half of nM to keep the struct three ints
  t.nM |= m_lo << 16;
  return t;
}

__device__ __forceinline__ float other_half(float v) {   // the value held by lane ^ 32
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  const unsigned x = __builtin_bit_cast(unsigned, v);
  const u2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  // one of the two results is this lane's own value, the other its partner's - whichever operand the swap moved
  return __builtin_bit_cast(float, r[0] != x ? r[0] : r[1]);
}
// max over the two lane halves.  NOT fmaxf(r[0], r[1]) of one swap: when the compiler hands the instruction the same
// register twice, both results are the partner's value and the lane's own maximum is lost (seen on gfx950: a query whose
// visible keys of a tile all sat in one half lost them).
__device__ __forceinline__ float max_halves(float v) { return fmaxf(v, other_half(v)); }

// mask one 32 x 32 score tile (keys on accumulator rows, queries on lanes): invisible or padded keys -> -inf
__device__ __forceinline__ void mask_keys(f32x16& S, int hh, int lim_r, int clo_r, int chi_r, uint32_t padbits) {
  lim_r -= 4 * hh; clo_r -= 4 * hh; chi_r -= 4 * hh;
  const uint32_t pb = padbits >> (4 * hh);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int rc = (i & 3) + 8 * (i >> 2);
    const bool ok = ((rc < lim_r) | ((rc >= clo_r) & (rc < chi_r))) & !((pb >> rc) & 1u);
    S[i] = ok ? S[i] : -INFINITY;
  }
}
// 32 padding bits of the keys k0 .. k0+31 (bit j = key k0+j is padded or beyond N), wave-uniform
__device__ __forceinline__ uint32_t pad_bits(const uint8_t* kp, int k0, int N, int r32) {
  const int key = k0 + r32;
  const bool bad = key >= N || (kp && kp[key]);
  return (uint32_t)__ballot(bad);          // lanes 0..31 and 32..63 vote alike: the low word is the tile's mask
}


struct KVRegs { bf16x8 k[4]; u32x4 v[4]; };

// v = mask[lane] ? v : 0 with the 64-bit lane mask in an SGPR pair (one VALU op; the mask comes from a stored ballot)
__device__ __forceinline__ float keep_by_mask(uint64_t mask, float v) {
  float r;
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask));
  return r;
}
// lanes L .. L+3 of the result take four wave-uniform values, the other lanes keep `old` (v_writelane_b32).  The values are
// fresh compare masks: a VALU write of an SGPR / VCC needs wait states before v_writelane may read it, and hipcc does not
// insert them around inline asm (seen: the records of every second key pair came out stale) - hence the s_nop.
template <int L>
__device__ __forceinline__ uint32_t write_lanes4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t old) {
  asm("s_nop 4\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8"
      : "+v"(old) : "s"(a), "s"(b), "s"(c), "s"(d), "n"(L), "n"(L + 1), "n"(L + 2), "n"(L + 3));
  return old;
}
// the 32 dwords of a keep-mask record as scalars (two s_load_dwordx16 from a wave-uniform address, waited for here)
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void sload_record(const uint32_t* ptr, u32x16& a, u32x16& b) {
  const uint64_t up = (uint64_t)ptr;
  // readfirstlane returns a SIGNED int: without the uint32_t casts a low word with bit 31 set sign-extends into the high
  // word and the scalar load faults (it did, address-dependently)
  const uint64_t sp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(up >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)up);
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b) : "s"(sp) : "memory");
}
__device__ __forceinline__ uint32_t* bits_block(const AttnP& p, int bh, int qt, int kt) {
  return p.drop_bits + ((((long)bh * p.nQT + qt) * p.nKT + kt) << 5);
}

// =================================================================================================
// forward
// =================================================================================================
template <int DM, int NW>   // DM dropout mode: 0 none, 1 hashed keep decisions, 2 keep-mask records (AttnP::drop_bits); NW waves
__global__ __launch_bounds__(NW * 64, 3) void attn2_fwd_kernel(Attn2P pp) {
  AttnP p = pp.a;
  W2VS_PIN_ATTNP(p);
  // loop: a V and a K tile per wave (4 KB each) + the workgroup's Q tile; afterwards the first 25.5 KB carry (O0, O1, m, l) of waves 1..3.
  // 36 KB per workgroup and ~150 registers: three workgroups per CU, so one workgroup's prologue / merge (dependent global
  // loads, two barriers) is covered by the loops of the others - with ~3 sub-tiles per wave those ends are not small.
  __shared__ __attribute__((aligned(16))) float smem[NW * 2 * 32 * HD / 2 + 32 * HD / 2];   // 32 KB: a K and a V tile per wave (the merge needs 25.5) + 4 KB: the Q tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hh = lane >> 5;
  W2VS_ASTAMP(0);
  const uint64_t rec = pp.rec[blockIdx.y];       // grid (B*H, tiles): x runs fastest, rows are dispatched longest first
  const int qt = (int)(rec & 1023), bh = blockIdx.x;
  const int b = fdiv(bh, p.mg_H), h = bh - b * p.H;
  const int N = p.N, Nq = p.Nq;
  const int q0 = qt * 32, q = q0 + r32, qc = min(q, Nq - 1);
  const bf16* Q = p.q + (long)b * p.sbq + h * HD;
  const bf16* K = p.k + (long)b * p.sb + h * HD;
  const bf16* V = p.v + (long)b * p.sb + h * HD;
  const uint8_t* kp = p.kpad ? p.kpad + (long)b * N : nullptr;
  const uint32_t ld24 = (uint32_t)p.ld;

  // the workgroup's Q tile: fetched once, row-contiguous (wave w brings rows 8w .. 8w+7), parked in LDS behind the waves'
  // K / V tiles; every wave takes its B fragments from there after the barrier below
  bf16* Qs = (bf16*)(smem + NW * 2 * 32 * HD / 2);
  {
    const int tch = (lane & 7) * 8;
#pragma unroll
    for (int j = 0; j < 4 / NW; ++j) {
      const int trow = (32 / NW) * wid + 8 * j + (lane >> 3);
      const u32x4 qv = *(const u32x4*)(Q + __umul24((uint32_t)min(q0 + trow, Nq - 1), (uint32_t)p.ldq) + tch);
      *(u32x4*)(Qs + uswz(trow, tch)) = qv;
    }
  }
  const QLimits L = q_limits_mg(qc, p);
  SubList tl;
  tl.nT = (int)(rec >> 10) & 1023; tl.nM = (int)(rec >> 20) & 1023; tl.rc0 = (int)(rec >> 30) & 1023;
  const int wfull = kp ? 0 : ((int)(rec >> 40) & 1023) * 32;     // keys [0, wfull) are visible to all 32 queries

  const float c = p.scale * LOG2E;
  const uint32_t thr = DM ? p.thr16 : 0u;
  const uint32_t s0 = (uint32_t)p.seed, s1 = (uint32_t)(p.seed >> 32);
  const uint32_t Nh = (uint32_t)(N + 1) >> 1;
  const uint32_t drow = ((uint32_t)(b * p.H + h) * (uint32_t)p.Ns + (uint32_t)qc) * Nh;

  f32x16 O0, O1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { O0[i] = 0.f; O1[i] = 0.f; }
  float mrun = -INFINITY, lrun = 0.f;       // mrun: scaled (log2) units, shared by both lane halves of a query
  bf16* Vw = (bf16*)smem + wid * (2 * 32 * HD);
  bf16* Kw = Vw + 32 * HD;
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int vrow = lane >> 3, vch = lane & 7;

  auto tile_of = [&](int pos) { return pos < tl.nM ? pos : tl.rc0 + (pos - tl.nM); };
  // K and V rows alike: row-contiguous global loads (8 lanes x 16 B per row) -> the wave's LDS tiles -> MFMA operands
  u32x4 kr[4], vr[4];
  auto load_kv = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t off = __umul24((uint32_t)min(t * 32 + vrow + 8 * j, N - 1), ld24) + vch * 8;
      kr[j] = *(const u32x4*)(K + off);
      vr[j] = *(const u32x4*)(V + off);
    }
  };
  int pos = wid;
  if (pos < tl.nT) load_kv(tile_of(pos));
  __syncthreads();
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(Qs + uswz(r32, (2 * s + hh) * 8));
  W2VS_ASTAMP(1);
  W2VS_ASTAMP_IDS(tl.nT);
  while (pos < tl.nT && !(W2VS_AABL & 8)) {
    const int k0 = tile_of(pos) * 32;
    const int nxt = pos + NW;
    // one register set: the V registers are free again once they sit in LDS, the K registers once S is issued - the next
    // sub-tile's loads go out right there and have the softmax and the P.V product to land
    asm volatile("" ::: "memory");
    if (!(W2VS_AABL & 4)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *(u32x4*)(Kw + uswz(vrow + 8 * j, vch * 8)) = kr[j];
        *(u32x4*)(Vw + vswz(vrow + 8 * j, vch * 8)) = vr[j];
      }
    }
    if (!(W2VS_AABL & 1) && nxt < tl.nT) load_kv(tile_of(nxt));
    f32x16 S;
#pragma unroll
    for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 kfr = *(const bf16x8*)(Kw + uswz(r32, (2 * s + hh) * 8));
      if (!(W2VS_AABL & 2)) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[s], S, 0, 0, 0);
      else S[s] += bf2f(kfr[0]);
    }
    if (!(k0 + 32 <= wfull)) {          // wave-uniform: boundary / right-context / padded tiles only
      const uint32_t pb = (kp || k0 + 32 > N) ? pad_bits(kp, k0, N, r32) : 0u;
      mask_keys(S, hh, L.lim - k0, L.clo - k0, L.chi - k0, pb);
    }
    float mloc = fmaxf(S[0], S[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mloc = fmaxf(fmaxf(mloc, S[i]), S[i + 1]);
    mloc = max_halves(mloc) * c;
    float muse;
    {
      const bool keep = __all(mloc <= mrun + 6.0f);
      // lazy rescale (attention.hip): keep the reference maximum while no query's maximum grew by more than 2^6.  Round 3:
      // the rescale sits under a wave-uniform branch (taken on the first sub-tile and then rarely) and the accumulators are
      // pinned to ONE register set behind it by an empty asm - without the pin the compiler carried them in two sets and paid
      // 40 v_mov_b64 per sub-tile, which is why round 2 multiplied by alpha = 1 unconditionally: 59 of the common path's 247
      // instructions (the 32 multiplies and their bookkeeping) are gone
      if (!keep) {
        const float mnew = fmaxf(mrun, mloc);
        const float mu = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = fast_exp2(mrun - mu);
        mrun = mnew;
        lrun *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { O0[i] *= alpha; O1[i] *= alpha; }
      }
      asm volatile("" : "+v"(O0), "+v"(O1));
      muse = (mrun == -INFINITY) ? 0.f : mrun;
    }
    float ls = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = fast_exp2(fmaf(S[i], c, -muse)); ls += S[i]; }
    lrun += ls;
    if (DM) {           // keep decisions: two per hash word (attn_common.h); 1/(1-p) is applied once, at the end
      const uint32_t wbase = (drow + (uint32_t)((k0 >> 1) + 2 * hh)) * HASH_K;
      uint32_t packed = 0;   // lane j < 32 collects dword j of this block's keep-mask record (AttnP::drop_bits)
#define W2VS_DROP_PAIR(i)                                                                                              \
      {                                                                                                                  \
        const uint32_t hw = pair_hash_pm(s0, s1, wbase + (uint32_t)((((i) & 3) >> 1) + 4 * ((i) >> 2)) * HASH_K);         \
        const bool ka = (hw & 0xFFFFu) >= thr, kb = (hw >> 16) >= thr;                                                    \
        S[i] = ka ? S[i] : 0.f;                                                                                           \
        S[(i) + 1] = kb ? S[(i) + 1] : 0.f;                                                                               \
        if (DM == 2) { /* the compare results ARE 64-lane masks: park them, one dword per lane */                         \
          const uint64_t ma = __ballot(ka), mb = __ballot(kb);                                                            \
          packed = write_lanes4<2 * (i)>((uint32_t)ma, (uint32_t)(ma >> 32), (uint32_t)mb, (uint32_t)(mb >> 32), packed);     \
        }                                                                                                                 \
      }
      W2VS_DROP_PAIR(0) W2VS_DROP_PAIR(2) W2VS_DROP_PAIR(4) W2VS_DROP_PAIR(6)
      W2VS_DROP_PAIR(8) W2VS_DROP_PAIR(10) W2VS_DROP_PAIR(12) W2VS_DROP_PAIR(14)
#undef W2VS_DROP_PAIR
      if (DM == 2 && lane < 32) bits_block(p, bh, qt, k0 >> 5)[lane] = packed;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pb = pack8(S, s2);
      const int krow = 16 * s2 + 4 * (g >> 1) + tq;
      const int dcol = (g & 1) * 16 + 4 * tp;
      const bf16x8 a0 = tr_pair(Vw + vswz(krow, dcol), Vw + vswz(krow + 8, dcol));
      const bf16x8 a1 = tr_pair(Vw + vswz(krow, 32 + dcol), Vw + vswz(krow + 8, 32 + dcol));
      if (!(W2VS_AABL & 2)) {
        O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pb, O0, 0, 0, 0);
        O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pb, O1, 0, 0, 0);
      } else { O0[s2] += bf2f(a0[0]) * bf2f(pb[0]); O1[s2] += bf2f(a1[0]); }
    }
    pos = nxt;
  }
  // ---- merge the four partial softmaxes ----
  W2VS_ASTAMP(2);
  float ltot = lrun + other_half(lrun);
  __syncthreads();                       // every wave is done with its V tile: the memory becomes the merge buffer
  if (wid > 0) {
    float* rw = smem + (wid - 1) * 34 * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) { rw[i * 64 + lane] = O0[i]; rw[(16 + i) * 64 + lane] = O1[i]; }
    rw[32 * 64 + lane] = mrun;
    rw[33 * 64 + lane] = ltot;
  }
  __syncthreads();
  W2VS_ASTAMP(3);
  if (wid != 0) return;
  float mw[NW - 1], mall = mrun;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w) { mw[w] = smem[(w * 34 + 32) * 64 + lane]; mall = fmaxf(mall, mw[w]); }
  const float mref = (mall == -INFINITY) ? 0.f : mall;
  const float a0s = fast_exp2(mrun - mref);
  ltot *= a0s;
#pragma unroll
  for (int i = 0; i < 16; ++i) { O0[i] *= a0s; O1[i] *= a0s; }
#pragma unroll
  for (int w = 0; w < NW - 1; ++w) {
    const float* rw = smem + w * 34 * 64;
    const float aw = fast_exp2(mw[w] - mref);
    ltot = fmaf(rw[33 * 64 + lane], aw, ltot);
#pragma unroll
    for (int i = 0; i < 16; ++i) { O0[i] = fmaf(rw[i * 64 + lane], aw, O0[i]); O1[i] = fmaf(rw[(16 + i) * 64 + lane], aw, O1[i]); }
  }
  const float inv_keep = DM ? 65536.f / (65536.f - (float)thr) : 1.f;
  const float inv = ltot > 0.f ? inv_keep / ltot : 0.f;
  if (q < Nq) {
    bf16* orow = p.o + (long)b * p.sbo + (long)q * p.ldo + h * HD;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      bf16x4 v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = f2bf(O0[4 * gq + e] * inv); v1[e] = f2bf(O1[4 * gq + e] * inv); }
      *(bf16x4*)(orow + 8 * gq + 4 * hh) = v0;
      *(bf16x4*)(orow + 32 + 8 * gq + 4 * hh) = v1;
    }
    if (hh == 0 && p.lse) p.lse[((long)(b * p.H + h)) * p.Ns + q] = ltot > 0.f ? (mall + log2f(ltot)) * LN2 : INFINITY;
  }
  W2VS_ASTAMP(4);
}

// =================================================================================================
// backward, dQ pass (also writes delta = dO . O for the dK/dV pass)
//   S^T = K Q^T ; P^T = exp(S^T - lse) ; dP^T = V dO^T ; dS^T = P^T o (dP^T o drop - delta)
//   dQ^T[d][q] += K^T[d][key] dS^T[key][q]           (scale applied once at the end)
// =================================================================================================
template <int DM, int NW>
__global__ __launch_bounds__(NW * 64, 3) void attn2_dq_kernel(Attn2P pp) {
  AttnP p = pp.a;
  W2VS_PIN_ATTNP(p);
  W2VS_PIN_ATTNP_BWD(p);
  // loop: a K and a V tile per wave + the workgroup's Q / dO / O tiles; afterwards the partial dQ of waves 1..3 (24 KB)
  __shared__ __attribute__((aligned(16))) float red_mem[NW * 2 * 32 * HD / 2 + 3 * 32 * HD / 2];   // 32 KB: a K and a V tile per wave (the merge needs 24) + 12 KB: the Q, dO, O tiles
  float (*red)[32][64] = (float (*)[32][64])red_mem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hh = lane >> 5;
  W2VS_ASTAMP(0);
  const uint64_t rec = pp.rec[blockIdx.y];       // grid (B*H, tiles): x runs fastest, rows are dispatched longest first
  const int qt = (int)(rec & 1023), bh = blockIdx.x;
  const int b = fdiv(bh, p.mg_H), h = bh - b * p.H;
  const int N = p.N, Nq = p.Nq;
  const int q0 = qt * 32, q = q0 + r32, qc = min(q, Nq - 1);
  const bf16* Q = p.q + (long)b * p.sbq + h * HD;
  const bf16* K = p.k + (long)b * p.sb + h * HD;
  const bf16* V = p.v + (long)b * p.sb + h * HD;
  const bf16* dO = p.dout + (long)b * p.sbo + h * HD;
  const uint8_t* kp = p.kpad ? p.kpad + (long)b * N : nullptr;
  const uint32_t ld24 = (uint32_t)p.ld;
  // the workgroup's Q, dO and O tiles: fetched once, row-contiguous (wave w brings rows 8w .. 8w+7 of each), parked in LDS
  // behind the waves' K / V tiles; every wave takes its fragments from there (before: four waves x twelve loads in the
  // one-row-per-lane operand layout - twice the address-unit time of the loop's loads)
  bf16* QDO = (bf16*)(red_mem + NW * 2 * 32 * HD / 2);
  {
    const int tch = (lane & 7) * 8;
#pragma unroll
    for (int j = 0; j < 4 / NW; ++j) {
      const int trow = (32 / NW) * wid + 8 * j + (lane >> 3);
      const uint32_t row = (uint32_t)min(q0 + trow, Nq - 1);
      const u32x4 qv = *(const u32x4*)(Q + __umul24(row, (uint32_t)p.ldq) + tch);
      const u32x4 dv = *(const u32x4*)(dO + __umul24(row, (uint32_t)p.ldo) + tch);
      const u32x4 ov = *(const u32x4*)(p.o + (long)b * p.sbo + h * HD + __umul24(row, (uint32_t)p.ldo) + tch);
      *(u32x4*)(QDO + uswz(trow, tch)) = qv;
      *(u32x4*)(QDO + 32 * HD + uswz(trow, tch)) = dv;
      *(u32x4*)(QDO + 2 * 32 * HD + uswz(trow, tch)) = ov;
    }
  }
  // the first sub-tile's K / V loads go out before anything waits: the delta below needs a round trip of its own (O rows)
  SubList tl;
  tl.nT = (int)(rec >> 10) & 1023; tl.nM = (int)(rec >> 20) & 1023; tl.rc0 = (int)(rec >> 30) & 1023;
  auto tile_of = [&](int pos) { return pos < tl.nM ? pos : tl.rc0 + (pos - tl.nM); };
  // K / V rows: row-contiguous global loads (8 lanes x 16 B per row) -> the wave's two LDS tiles -> MFMA operands (as in the
  // dK/dV pass: loads in the operand layout, one row per lane, cost 13 % of the backward)
  u32x4 kr[4], vr[4];
  const int crow = lane >> 3, cch = lane & 7;
  auto load_kv = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t off = __umul24((uint32_t)min(t * 32 + crow + 8 * j, N - 1), ld24) + cch * 8;
      kr[j] = *(const u32x4*)(K + off);
      vr[j] = *(const u32x4*)(V + off);
    }
  };
  int pos = wid;
  if (pos < tl.nT) load_kv(tile_of(pos));
  const long sidx = ((long)(b * p.H + h)) * p.Ns + qc;
  const float lse2 = p.lse[sidx] * LOG2E;
  __syncthreads();
  bf16x8 qf[4], dof[4];
  float delta = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *(const bf16x8*)(QDO + uswz(r32, (2 * s + hh) * 8));
    dof[s] = *(const bf16x8*)(QDO + 32 * HD + uswz(r32, (2 * s + hh) * 8));
    const bf16x8 ov = *(const bf16x8*)(QDO + 2 * 32 * HD + uswz(r32, (2 * s + hh) * 8));
#pragma unroll
    for (int e = 0; e < 8; ++e) delta = fmaf(bf2f(ov[e]), bf2f(dof[s][e]), delta);
  }
  delta += other_half(delta);
  if (wid == 0 && q < Nq && hh == 0) const_cast<float*>(p.delta)[sidx] = delta;
  const QLimits L = q_limits_mg(qc, p);
  const int wfull = kp ? 0 : ((int)(rec >> 40) & 1023) * 32;     // keys [0, wfull) are visible to all 32 queries
  const float c = p.scale * LOG2E;
  const uint32_t thr = DM ? p.thr16 : 0u;
  const float inv_keep = DM ? 65536.f / (65536.f - (float)thr) : 1.f;
  const uint32_t s0 = (uint32_t)p.seed, s1 = (uint32_t)(p.seed >> 32);
  const uint32_t Nh = (uint32_t)(N + 1) >> 1;
  const uint32_t drow = ((uint32_t)(b * p.H + h) * (uint32_t)p.Ns + (uint32_t)qc) * Nh;
  f32x16 D0, D1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { D0[i] = 0.f; D1[i] = 0.f; }
  bf16* Kw = (bf16*)red_mem + wid * (2 * 32 * HD);
  bf16* Vw = Kw + 32 * HD;
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  W2VS_ASTAMP(1);
  W2VS_ASTAMP_IDS(tl.nT);
  while (pos < tl.nT) {
    const int k0 = tile_of(pos) * 32;
    const int nxt = pos + NW;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(u32x4*)(Kw + uswz(crow + 8 * j, cch * 8)) = kr[j];
      *(u32x4*)(Vw + uswz(crow + 8 * j, cch * 8)) = vr[j];
    }
    if (nxt < tl.nT) load_kv(tile_of(nxt));     // one register set: free once the tiles sit in LDS
    f32x16 S, dP;
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 kfr = *(const bf16x8*)(Kw + uswz(r32, (2 * s + hh) * 8));
      const bf16x8 vfr = *(const bf16x8*)(Vw + uswz(r32, (2 * s + hh) * 8));
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[s], S, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, dof[s], dP, 0, 0, 0);
    }
    if (!(k0 + 32 <= wfull)) {
      const uint32_t pb = (kp || k0 + 32 > N) ? pad_bits(kp, k0, N, r32) : 0u;
      mask_keys(S, hh, L.lim - k0, L.clo - k0, L.chi - k0, pb);
    }
    if (DM) {
      if (DM == 2) {       // the forward's decisions, one scalar pair + one v_cndmask per element (no hash)
        u32x16 ra, rb;
        sload_record(bits_block(p, bh, qt, k0 >> 5), ra, rb);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          dP[i] = keep_by_mask(((uint64_t)ra[2 * i + 1] << 32) | (uint64_t)ra[2 * i], dP[i]);
          dP[8 + i] = keep_by_mask(((uint64_t)rb[2 * i + 1] << 32) | (uint64_t)rb[2 * i], dP[8 + i]);
        }
      } else {
        const uint32_t wbase = (drow + (uint32_t)((k0 >> 1) + 2 * hh)) * HASH_K;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const uint32_t hw = pair_hash_pm(s0, s1, wbase + (uint32_t)(((i & 3) >> 1) + 4 * (i >> 2)) * HASH_K);
          dP[i] = (hw & 0xFFFFu) >= thr ? dP[i] : 0.f;
          dP[i + 1] = (hw >> 16) >= thr ? dP[i + 1] : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) S[i] = fast_exp2(fmaf(S[i], c, -lse2)) * fmaf(dP[i], inv_keep, -delta);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 db = pack8(S, s2);
      const int krow = 16 * s2 + 4 * (g >> 1) + tq;
      const int dcol = (g & 1) * 16 + 4 * tp;
      const bf16x8 a0 = tr_pair(Kw + uswz(krow, dcol), Kw + uswz(krow + 8, dcol));
      const bf16x8 a1 = tr_pair(Kw + uswz(krow, 32 + dcol), Kw + uswz(krow + 8, 32 + dcol));
      D0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, db, D0, 0, 0, 0);
      D1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, db, D1, 0, 0, 0);
    }
    pos = nxt;
  }
  W2VS_ASTAMP(2);
  __syncthreads();
  if (wid > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { red[wid - 1][i][lane] = D0[i]; red[wid - 1][16 + i][lane] = D1[i]; }
  }
  __syncthreads();
  W2VS_ASTAMP(3);
  if (wid != 0) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w)
#pragma unroll
    for (int i = 0; i < 16; ++i) { D0[i] += red[w][i][lane]; D1[i] += red[w][16 + i][lane]; }
  if (q < Nq) {
    bf16* drow_p = p.dq + (long)b * p.sbq + (long)q * p.ldq + h * HD;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      bf16x4 v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = f2bf(D0[4 * gq + e] * p.scale); v1[e] = f2bf(D1[4 * gq + e] * p.scale); }
      *(bf16x4*)(drow_p + 8 * gq + 4 * hh) = v0;
      *(bf16x4*)(drow_p + 32 + 8 * gq + 4 * hh) = v1;
    }
  }
  W2VS_ASTAMP(4);
}

// =================================================================================================
// backward, dK / dV pass.  S (not transposed): queries on accumulator rows, keys on lanes.
//   S = Q K^T ; P = exp(S - lse[q]) ; dP = dO V^T ; dS = P o (dP o drop - delta[q])        (scale at the end, dK only)
//   dV^T[d][key] += dO^T[d][q] (P o drop)[q][key]      dK^T[d][key] += Q^T[d][q] dS[q][key]
// =================================================================================================
template <int DM, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn2_dkv_kernel(Attn2P pp) {
  AttnP p = pp.a;
  W2VS_PIN_ATTNP(p);
  W2VS_PIN_ATTNP_BWD(p);
  // during the loop: per wave a Q tile and a dO tile (tr-read images, 4 KB each) and five 32-entry query vectors;
  // afterwards the same memory carries the partial dK / dV of waves 1..3
  constexpr int TILES_F = NW * 2 * 32 * HD / 2, KVS_F = TILES_F + ((NW * 160 + 255) / 256) * 256, LOOP_F = KVS_F + 2 * 32 * HD / 2;
  constexpr int MERGE_F = (NW - 1) * 64 * 64;
  __shared__ __attribute__((aligned(16))) float smem[LOOP_F > MERGE_F ? LOOP_F : MERGE_F];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hh = lane >> 5;
  W2VS_ASTAMP(0);
  const uint64_t rec = pp.rec[blockIdx.y];
  const int kt = (int)(rec & 1023), bh = blockIdx.x;
  const int b = fdiv(bh, p.mg_H), h = bh - b * p.H;
  const int N = p.N, Nq = p.Nq;
  const int kb0 = kt * 32, key = kb0 + r32, keyc = min(key, N - 1);
  bf16* Qw = (bf16*)smem + wid * (2 * 32 * HD);
  bf16* Dw = Qw + 32 * HD;
  float* qs = smem + (NW * 2 * 32 * HD) / 2 + wid * 160;       // lse, delta, lim, clo, chi of the sub-tile's 32 queries
  const bf16* Q = p.q + (long)b * p.sbq + h * HD;
  const bf16* K = p.k + (long)b * p.sb + h * HD;
  const bf16* V = p.v + (long)b * p.sb + h * HD;
  const bf16* dO = p.dout + (long)b * p.sbo + h * HD;
  const float* lse_bh = p.lse + (long)bh * p.Ns;
  const float* delta_bh = p.delta + (long)bh * p.Ns;
  const bool key_ok = key < N && !(p.kpad && p.kpad[(long)b * N + key]);
  // the workgroup's K / V tile: fetched ONCE, row-contiguous (wave w brings rows 8w .. 8w+7), parked in LDS, and every wave
  // takes its MFMA fragments from there (before: four waves x eight loads in the one-row-per-lane operand layout - as much
  // address-unit time as the whole loop's loads)
  bf16* KVs = (bf16*)(smem + KVS_F);      // behind the waves' tiles and query vectors: K tile, then V tile (4 KB each)
  {
    const int tch = (lane & 7) * 8;
#pragma unroll
    for (int j = 0; j < 4 / NW; ++j) {
      const int trow = (32 / NW) * wid + 8 * j + (lane >> 3);
      const uint32_t off = __umul24((uint32_t)min(kb0 + trow, N - 1), (uint32_t)p.ld) + tch;
      const u32x4 kv = *(const u32x4*)(K + off), vv = *(const u32x4*)(V + off);
      *(u32x4*)(KVs + uswz(trow, tch)) = kv;
      *(u32x4*)(KVs + 32 * HD + uswz(trow, tch)) = vv;
    }
  }
  const float c = p.scale * LOG2E;
  const uint32_t thr = DM ? p.thr16 : 0u;
  const float inv_keep = DM ? 65536.f / (65536.f - (float)thr) : 1.f;
  const uint32_t s0 = (uint32_t)p.seed, s1 = (uint32_t)(p.seed >> 32);
  const uint32_t Nh = (uint32_t)(N + 1) >> 1;
  const uint32_t dbase = (uint32_t)(b * p.H + h) * (uint32_t)p.Ns;
  const uint32_t khalf = (uint32_t)keyc >> 1, kodd = (uint32_t)keyc & 1u;
  const uint32_t stepK = Nh * HASH_K;
  const uint32_t lane_rows = (uint32_t)(lane & 1) * 2u;
  const bool keys_clean = __all(key_ok);
  SubList ql;
  ql.nT = (int)(rec >> 10) & 1023; ql.nM = (int)(rec >> 20) & 1023; ql.rc0 = (int)(rec >> 30) & 1023;
  const int qm_lo = (int)(rec >> 40) & 1023;
  f32x16 dV0, dV1, dK0, dK1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dV0[i] = dV1[i] = dK0[i] = dK1[i] = 0.f; }
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  // Q / dO rows travel global -> registers -> the wave's LDS tiles -> MFMA operands.  The global loads are ROW-CONTIGUOUS
  // (8 lanes x 16 B per row, 8 rows per instruction): loading straight into the MFMA operand layout (lane = row) touched 32
  // rows x 32 B per instruction and ran the backward 13 % slower (timing-only build).
  struct Regs { u32x4 q[4], d[4]; float sc; uint32_t bits; };   // sc: lse (lanes < 32) / delta (lanes >= 32) of query q0 + r32
  auto tile_of = [&](int pos) { return pos < ql.nM ? qm_lo + pos : ql.rc0 + (pos - ql.nM); };
  const int crow = lane >> 3, cch = lane & 7;
  Regs R;
  auto gload = [&](int t) {
    const int qq = min(t * 32 + r32, Nq - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(t * 32 + crow + 8 * j, Nq - 1);
      R.q[j] = *(const u32x4*)(row_at(Q, row, (uint32_t)p.ldq) + cch * 8);
      R.d[j] = *(const u32x4*)(row_at(dO, row, (uint32_t)p.ldo) + cch * 8);
    }
    R.sc = hh ? delta_bh[qq] : lse_bh[qq] * LOG2E;
    // this lane's key row of the block's keep-mask record: dword 2i + w with key = (i&3) + 8(i>>2) + 4w
    R.bits = DM == 2 ? bits_block(p, bh, t, kt)[2 * (r32 & 3) + 8 * (r32 >> 3) + ((r32 >> 2) & 1)] : 0u;
  };
  int pos = wid;
  if (pos < ql.nT) gload(tile_of(pos));
  __syncthreads();
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = *(const bf16x8*)(KVs + uswz(r32, (2 * s + hh) * 8));
    vf[s] = *(const bf16x8*)(KVs + 32 * HD + uswz(r32, (2 * s + hh) * 8));
  }
  W2VS_ASTAMP(1);
  W2VS_ASTAMP_IDS(ql.nT);
  while (pos < ql.nT) {
    const int q0 = tile_of(pos) * 32;
    const int nxt = pos + NW;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(u32x4*)(Qw + uswz(crow + 8 * j, cch * 8)) = R.q[j];
      *(u32x4*)(Dw + uswz(crow + 8 * j, cch * 8)) = R.d[j];
    }
    // per-query scalars: rows past Nq get lse = +inf -> P = 0
    const bool qvalid = q0 + r32 < Nq;
    qs[lane] = qvalid ? R.sc : (hh ? 0.f : INFINITY);
    const uint32_t bits_now = R.bits;
    if (nxt < ql.nT) gload(tile_of(nxt));      // one register set: everything of this sub-tile sits in LDS by now
    // visibility limits of the sub-tile's queries; "full" = every query sees every key of this workgroup, none padded
    int minlim = 0;
    {
      const int qs1 = min(q0 + 32, Nq) - 1;
      if (p.mq > 0) minlim = min((fdiv(q0, p.mg_mq) + 1) * p.m, p.Tp);
      else if (qs1 < p.Tp) minlim = min((fdiv(q0, p.mg_m) + 1) * p.m, p.Tp);
      else if (q0 >= p.Tp && p.r > 0) minlim = min((fdiv(q0 - p.Tp, p.mg_r) + 1) * p.m, p.Tp);
    }
    const bool full = keys_clean && q0 + 32 <= Nq && kb0 + 32 <= minlim;
    if (!full && hh == 0) {
      const QLimits L = q_limits_mg(min(q0 + r32, Nq - 1), p);
      ((int*)qs)[64 + r32] = L.lim; ((int*)qs)[96 + r32] = L.clo; ((int*)qs)[128 + r32] = L.chi;
    }
    f32x16 S, dP;
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 qfr = *(const bf16x8*)(Qw + uswz(r32, (2 * s + hh) * 8));
      const bf16x8 dfr = *(const bf16x8*)(Dw + uswz(r32, (2 * s + hh) * 8));
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr, kf[s], S, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfr, vf[s], dP, 0, 0, 0);
    }
    asm volatile("" ::: "memory");
    const uint32_t wbits = bits_now >> (4 * hh);      // bit (i&3) + 8(i>>2) = query row of accumulator element i
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const uint32_t wsub = ((dbase + (uint32_t)(q0 + 4 * hh)) * Nh + khalf) * HASH_K;
    const uint32_t wsub_l = wsub + lane_rows * stepK;      // this lane's share of the hashing starts at row 0 or 2 of each group
    f32x16 Pd;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int rb = 8 * g4 + 4 * hh;
      const f32x4 lse4 = *(const f32x4*)(qs + rb);
      const f32x4 del4 = *(const f32x4*)(qs + 32 + rb);
      float pe[4];
      if (full) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pe[e] = fast_exp2(fmaf(S[4 * g4 + e], c, -lse4[e]));
      } else {
        const i32x4 lim4 = *(const i32x4*)((const int*)qs + 64 + rb), clo4 = *(const i32x4*)((const int*)qs + 96 + rb),
                    chi4 = *(const i32x4*)((const int*)qs + 128 + rb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = key_ok & ((key < lim4[e]) | ((key >= clo4[e]) & (key < chi4[e])));
          const float v = fast_exp2(fmaf(S[4 * g4 + e], c, -lse4[e]));
          pe[e] = ok ? v : 0.f;
        }
      }
      // The two lanes of a key pair (keys 2j, 2j+1: lanes l, l^1) need the SAME words - a word covers the pair - and only
      // test different halves.  Each hashes half of them (even lane: rows e = 0, 1 of the group of four; odd lane: e = 2, 3)
      // and reads the other half from its neighbour through DPP: 8 hashes per sub-tile instead of 16.
      uint32_t hw4[4] = {0u, 0u, 0u, 0u};
      if (DM == 1) {
        const uint32_t h0 = pair_hash_pm(s0, s1, wsub_l + (uint32_t)(8 * g4) * stepK);
        const uint32_t h1 = pair_hash_pm(s0, s1, wsub_l + (uint32_t)(8 * g4 + 1) * stepK);
        hw4[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0xA0, 0xF, 0xF, false);   // quad_perm [0,0,2,2]: the even lane's
        hw4[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0xA0, 0xF, 0xF, false);
        hw4[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h0, 0xF5, 0xF, 0xF, false);   // quad_perm [1,1,3,3]: the odd lane's
        hw4[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0xF5, 0xF, 0xF, false);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * g4 + e;
        int km = -1;                                   // non-zero = keep
        if (DM) {
          if (DM == 2) km = (int)(wbits & (1u << ((i & 3) + 8 * (i >> 2))));
          else {           // word ((dbase + query) * Nh + key / 2): consecutive rows are stepK apart
            km = ((kodd ? (hw4[e] >> 16) : (hw4[e] & 0xFFFFu)) >= thr) ? -1 : 0;
          }
        }
        // 1/(1-p) multiplies dV once at the end; inside dS it rides in the fma
        const bool kp = km != 0;       // (selects, not integer ANDs on the float bits: that form miscompiled dS on ROCm 7.2)
        Pd[i] = kp ? pe[e] : 0.f;
        S[i] = pe[e] * fmaf(kp ? dP[i] : 0.f, inv_keep, -del4[e]);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pb = pack8(Pd, s2), sb = pack8(S, s2);
      const int qrow = 16 * s2 + 4 * (g >> 1) + tq;
      const int dcol = (g & 1) * 16 + 4 * tp;
      const bf16x8 d0 = tr_pair(Dw + uswz(qrow, dcol), Dw + uswz(qrow + 8, dcol));
      const bf16x8 d1 = tr_pair(Dw + uswz(qrow, 32 + dcol), Dw + uswz(qrow + 8, 32 + dcol));
      dV0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, pb, dV0, 0, 0, 0);
      dV1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, pb, dV1, 0, 0, 0);
      const bf16x8 q0f = tr_pair(Qw + uswz(qrow, dcol), Qw + uswz(qrow + 8, dcol));
      const bf16x8 q1f = tr_pair(Qw + uswz(qrow, 32 + dcol), Qw + uswz(qrow + 8, 32 + dcol));
      dK0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0f, sb, dK0, 0, 0, 0);
      dK1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1f, sb, dK1, 0, 0, 0);
    }
    pos = nxt;
  }
  W2VS_ASTAMP(2);
  __syncthreads();                       // every wave is done with its tiles: the memory becomes the reduction buffer
  if (wid > 0) {
    float* rw = smem + (wid - 1) * 64 * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      rw[i * 64 + lane] = dK0[i]; rw[(16 + i) * 64 + lane] = dK1[i];
      rw[(32 + i) * 64 + lane] = dV0[i]; rw[(48 + i) * 64 + lane] = dV1[i];
    }
  }
  __syncthreads();
  W2VS_ASTAMP(3);
  if (wid != 0) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w) {
    const float* rw = smem + w * 64 * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dK0[i] += rw[i * 64 + lane]; dK1[i] += rw[(16 + i) * 64 + lane];
      dV0[i] += rw[(32 + i) * 64 + lane]; dV1[i] += rw[(48 + i) * 64 + lane];
    }
  }
  if (key < N) {
    bf16* dkr = p.dk + (long)b * p.sb + (long)key * p.ld + h * HD;
    bf16* dvr = p.dv + (long)b * p.sb + (long)key * p.ld + h * HD;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      bf16x4 a0, a1, b0, b1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] = f2bf(dK0[4 * gq + e] * p.scale); a1[e] = f2bf(dK1[4 * gq + e] * p.scale);
        b0[e] = f2bf(dV0[4 * gq + e] * inv_keep); b1[e] = f2bf(dV1[4 * gq + e] * inv_keep);
      }
      *(bf16x4*)(dkr + 8 * gq + 4 * hh) = a0;
      *(bf16x4*)(dkr + 32 + 8 * gq + 4 * hh) = a1;
      *(bf16x4*)(dvr + 8 * gq + 4 * hh) = b0;
      *(bf16x4*)(dvr + 32 + 8 * gq + 4 * hh) = b1;
    }
  }
  W2VS_ASTAMP(4);
}

// the launch's record table (host): one record per 32-row tile with its visible sub-tile list, longest list first - the
// tail of the launch is then made of the short workgroups
template <class F>
void make_table(Attn2P& pp, int first, int ntiles, F list) {   // list(t, aux) -> SubList of tile t (nM without packed extras)
  struct E { int nT, t, nM, rc0, aux; };                        // tiles [first, first + ntiles): one launch holds <= MAXT2 records
  if (ntiles > MAXT2) ntiles = MAXT2;                           // (callers chunk; never write past the kernel argument)
  std::vector<E> v(ntiles);
  for (int i = 0; i < ntiles; ++i) {
    int aux = 0;
    const SubList l = list(first + i, aux);
    v[i] = {l.nT, first + i, l.nM, l.rc0, aux};
  }
  std::stable_sort(v.begin(), v.end(), [](const E& a, const E& b) { return a.nT > b.nT; });
  pp.ntiles = ntiles;
  for (int i = 0; i < ntiles; ++i)
    pp.rec[i] = (uint64_t)v[i].t | ((uint64_t)v[i].nT << 10) | ((uint64_t)v[i].nM << 20) | ((uint64_t)v[i].rc0 << 30) |
                ((uint64_t)v[i].aux << 40);
}
static void query_tile_table(Attn2P& pp, const AttnP& p, int first, int nqt) {
  make_table(pp, first, nqt, [&](int t, int& aux) {
    int full;
    const SubList l = key_list(t * 32, std::min(t * 32 + 32, p.Nq) - 1, p.Tp, p.m, p.r, p.N, p.mq, full);
    aux = full >> 5;
    return l;
  });
}
static void key_tile_table(Attn2P& pp, const AttnP& p, int nkt) {
  make_table(pp, 0, nkt, [&](int t, int& aux) {
    SubList l = query_list(t * 32, std::min(t * 32 + 32, p.N) - 1, p.Tp, p.m, p.r, p.N, p.Nq, p.mq);
    aux = l.nM >> 16;
    l.nM &= 0xFFFF;
    return l;
  });
}

}  // namespace

// dropout mode of a launch: the kernels are compiled once per mode, so the no-dropout loop carries neither the hash nor
// the branches around it
static inline int drop_mode(const AttnP& p) { return p.thr16 == 0 ? 0 : (p.drop_bits ? 2 : 1); }
// forward / dQ pass: waves per workgroup.  Both kernels are bound by instruction issue, and a workgroup's prologue + merge are
// about a third of its instructions: two waves instead of four halve that share - as long as the longest tile's sub-tile
// list, now split two ways only, does not become the tail of the launch.  Measured (p = 0.1): N = 818 (longest list 26)
// forward 33.4 -> 31.0 us, dQ pass -2 us; N = 1496 (longest 47) forward 35.3 -> 38.2 us.  W2VS_ATTN_NW=2|4 forces one.
static const int g_attn_nw_env = W2VS_ENV_INT("W2VS_ATTN_NW", 0);
#define W2VS_LAUNCH_DM_NW(kern, tiles)                                                                              \
 {const int longest_ = (int)(pp.rec[0] >> 10) & 1023;     /* the table is sorted longest first */                   \
  const int nw_ = g_attn_nw_env == 2 || g_attn_nw_env == 4 ? g_attn_nw_env : (longest_ <= 32 ? 2 : 4);               \
  if (nw_ == 2) {                                                                                              \
    switch (drop_mode(p)) {                                                                                          \
      case 0: hipLaunchKernelGGL((kern<0, 2>), dim3(p.B * p.H, tiles), dim3(128), 0, st, pp); break;                 \
      case 1: hipLaunchKernelGGL((kern<1, 2>), dim3(p.B * p.H, tiles), dim3(128), 0, st, pp); break;                 \
      default: hipLaunchKernelGGL((kern<2, 2>), dim3(p.B * p.H, tiles), dim3(128), 0, st, pp); break;                \
    }                                                                                                                \
  } else {                                                                                                           \
    switch (drop_mode(p)) {                                                                                          \
      case 0: hipLaunchKernelGGL((kern<0, 4>), dim3(p.B * p.H, tiles), dim3(256), 0, st, pp); break;                 \
      case 1: hipLaunchKernelGGL((kern<1, 4>), dim3(p.B * p.H, tiles), dim3(256), 0, st, pp); break;                 \
      default: hipLaunchKernelGGL((kern<2, 4>), dim3(p.B * p.H, tiles), dim3(256), 0, st, pp); break;                \
    }                                                                                                                \
  }}

// (the multiply-high divisions are exact below 65536: positions, and the (batch, head) index;
// and the 24-bit row-address products need strides below 2^24 and N * stride below 2^32)
bool attn2_ok(const AttnP& p) {
  const long ldmax = std::max(std::max(p.ld, p.ldo), p.ldq);
  return (p.N + 31) / 32 <= MAXT2 && (long)p.B * p.H < 65536 && p.m < 65536 && p.r < 65536 && p.mq < 65536 &&
         ldmax < (1l << 24) && (long)std::max(p.N, p.Nq) * ldmax < (1l << 32);
}

#ifdef W2VS_ABLATION
static unsigned long long* g_attn_stamps = nullptr;
static int g_attn_stamp_pass = 0;      // which launch writes: 0 forward, 1 dQ pass, 2 dK/dV pass
extern "C" void w2vs_dbg_attn_stamps(void* buf, int pass) { g_attn_stamps = (unsigned long long*)buf; g_attn_stamp_pass = pass; }
#define W2VS_SET_STAMPS(pp, pass) (pp).stamps = g_attn_stamp_pass == (pass) ? g_attn_stamps : nullptr
#else
#define W2VS_SET_STAMPS(pp, pass) do { } while (0)
#endif

int attn2_fwd(const AttnP& p, hipStream_t st) {
  Attn2P pp;
  pp.a = p;
  W2VS_SET_STAMPS(pp, 0);
  pp.a.nQT = (p.Nq + 31) / 32; pp.a.nKT = (p.N + 31) / 32;
  const int nqt = (p.Nq + 31) / 32;
  // a launch carries at most MAXT2 tile records in its kernel argument: the cross mode (queries = G x U joiner rows, up to
  // 512 tiles) goes out in chunks of query tiles; records hold absolute tile numbers, so the kernels need no offset
  for (int t0 = 0; t0 < nqt; t0 += MAXT2) {
    const int cnt = std::min(MAXT2, nqt - t0);
    query_tile_table(pp, p, t0, cnt);
    W2VS_LAUNCH_DM_NW(attn2_fwd_kernel, cnt)
  }
  return hip_check(hipGetLastError(), "attn_fwd");
}

int attn2_bwd(const AttnP& p, hipStream_t st) {
  Attn2P pp;
  pp.a = p;
  pp.a.nQT = (p.Nq + 31) / 32; pp.a.nKT = (p.N + 31) / 32;
  const int nqt = (p.Nq + 31) / 32, nkt = (p.N + 31) / 32;
  for (int t0 = 0; t0 < nqt; t0 += MAXT2) {
    const int cnt = std::min(MAXT2, nqt - t0);
    query_tile_table(pp, p, t0, cnt);
    W2VS_SET_STAMPS(pp, 1);
    W2VS_LAUNCH_DM_NW(attn2_dq_kernel, cnt)     // dq rows >= Nq are not written
  }
  key_tile_table(pp, p, nkt);
  W2VS_SET_STAMPS(pp, 2);
  {
    static const int dkv_nw_env = W2VS_ENV_INT("W2VS_ATTN_DKV_NW", 0);
    const int longest_k = (int)(pp.rec[0] >> 10) & 1023;
    // two waves per workgroup: half the K / V staging, hand-over and merge instructions per key tile (measured: dQ + dK/dV
    // 96.6 -> 89.8 us at N = 818, 99.6 -> 92.2 us at N = 1496); four once a key tile's query list gets long enough to be the tail
    const int nwk = dkv_nw_env == 2 || dkv_nw_env == 4 ? dkv_nw_env : (longest_k <= 128 ? 2 : 4);
    if (nwk == 2) {
      switch (drop_mode(p)) {
        case 0: hipLaunchKernelGGL((attn2_dkv_kernel<0, 2>), dim3(p.B * p.H, nkt), dim3(128), 0, st, pp); break;
        case 1: hipLaunchKernelGGL((attn2_dkv_kernel<1, 2>), dim3(p.B * p.H, nkt), dim3(128), 0, st, pp); break;
        default: hipLaunchKernelGGL((attn2_dkv_kernel<2, 2>), dim3(p.B * p.H, nkt), dim3(128), 0, st, pp); break;
      }
    } else {
      switch (drop_mode(p)) {
        case 0: hipLaunchKernelGGL((attn2_dkv_kernel<0, 4>), dim3(p.B * p.H, nkt), dim3(256), 0, st, pp); break;
        case 1: hipLaunchKernelGGL((attn2_dkv_kernel<1, 4>), dim3(p.B * p.H, nkt), dim3(256), 0, st, pp); break;
        default: hipLaunchKernelGGL((attn2_dkv_kernel<2, 4>), dim3(p.B * p.H, nkt), dim3(256), 0, st, pp); break;
      }
    }
  }
  return hip_check(hipGetLastError(), "attn_bwd");
}

}  // namespace w2vs
