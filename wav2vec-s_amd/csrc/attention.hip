// Block-causal streaming attention with per-block right-context copies (gfx950, head_dim 64).
//
// Replaces gen_block_attn_mask + F.multi_head_attention_forward's core
// (fs/models/wav2vec/wav2vec_S.py:444-489, fs/modules/multihead_attention.py:161-193).
// The reference materialises an N x N float mask (0 / -1e4) and a B x N key-padding mask and
// runs dense attention; here the mask is never built.  Token n < Tp is main frame n of block
// n/m; token Tp+c is right-context copy c of block c/r.  Query q (block bq) may attend key k
//     k <  Tp : iff k < min((bq+1)*m, Tp)
//     k >= Tp : iff Tp + bq*r <= k < Tp + (bq+1)*r
// and never a padded key.  exp(-1e4 + s - max) underflows to exactly 0 in fp32, so dropping
// masked keys equals the reference's additive -1e4 for every row that has an allowed key
// (always true: a query's own block is allowed).  Whole 64-key tiles outside a query
// tile's allowed set are skipped.
//
// Flash-style: scores never touch HBM.  MFMA 32x32x16 bf16; S is computed TRANSPOSED
// (keys on accumulator rows, queries on lanes) so that softmax statistics are per-lane scalars
// and the bf16-converted accumulator is directly the B operand of the P.V product; V^T / K^T /
// Q^T / dO^T operands come out of row-major LDS tiles through ds_read_b64_tr_b16.
#include <stdlib.h>
#include "attn_common.h"

namespace w2vs {

// cooperative load of a [64][64] tile (rows row0.., zero beyond nrows) into a swizzled LDS image
template <bool TR>
__device__ __forceinline__ void load_tile(bf16* dst, const bf16* src, long ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int c = tid + 256 * j, row = c >> 3, ch = c & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row0 + row < nrows) v = *(const u32x4*)(src + (long)(row0 + row) * ld + ch * 8);
    int off = TR ? vswz(row, ch * 8) : kswz(row, ch);
    *(u32x4*)(dst + off) = v;
  }
}

// =================================================================================================
// forward
// =================================================================================================
// 3 waves per SIMD (168 VGPRs, 9 dwords of spill): +5 % over 2 waves at 176 VGPRs
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnP p) {
  // double-buffered K / V tiles: the next needed tile is prefetched into registers while the current
  // one is consumed from LDS, then written to the other buffer (one barrier per tile)
  __shared__ __attribute__((aligned(16))) bf16 Ks[2][KT * HD];
  __shared__ __attribute__((aligned(16))) bf16 Vs[2][KT * HD];
  __shared__ __attribute__((aligned(16))) float kbias[2][KT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qblk0 = (gridDim.x - 1 - blockIdx.x) * QB;  // longest query tiles first: shorter tail
  const int q = qblk0 + wid * 32 + r32;
  const int N = p.N, Nq = p.Nq;
  const bf16* Q = p.q + (long)b * p.sb + h * HD;
  const bf16* K = p.k + (long)b * p.sb + h * HD;
  const bf16* V = p.v + (long)b * p.sb + h * HD;
  const uint8_t* kp = p.kpad ? p.kpad + (long)b * N : nullptr;

  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (q < Nq) qf[s] = *(const bf16x8*)(Q + (long)q * p.ld + 16 * s + 8 * hh);
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = f2bf(0.f);
  }
  const QLimits L = q_limits(min(q, Nq - 1), p.Tp, p.m, p.r, N);
  int mlim, bclo, bchi;
  tile_ranges(qblk0, min(qblk0 + QB, Nq) - 1, p.Tp, p.m, p.r, N, mlim, bclo, bchi);
  // per-wave key ranges: sub-tiles no query of this wave can see are skipped, sub-tiles every
  // query sees completely (and that hold no padded key) skip the mask arithmetic
  const int wmlim = wave_max_i(L.lim), wfull = wave_min_i(L.lim), wclo = wave_min_i(L.clo), wchi = wave_max_i(L.chi);

  const float c = p.scale * LOG2E;
  const uint32_t thr = drop_threshold(p.p_drop) >> 16;           // 16-bit threshold
  const float inv_keep = thr > 0 ? 65536.f / (65536.f - (float)thr) : 1.f;
  const uint32_t s0 = (uint32_t)p.seed, s1 = (uint32_t)(p.seed >> 32);
  const uint32_t Nh = (uint32_t)(N + 1) >> 1;
  const uint32_t drow = ((uint32_t)(b * p.H + h) * (uint32_t)N + (uint32_t)min(q, Nq - 1)) * Nh;

  f32x16 O0, O1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { O0[i] = 0.f; O1[i] = 0.f; }
  float mrun = -INFINITY, lrun = 0.f;

  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  auto next_tile = [&](int kt0) {  // first needed 64-key tile at or after kt0 (block-uniform)
    while (kt0 < N && !((kt0 < mlim) || (kt0 < bchi && kt0 + KT > bclo))) kt0 += KT;
    return kt0;
  };
  // Register-staged prefetch TWO tiles ahead (two register sets): an ablation showed the loop bound by the
  // latency of the global loads, which one tile of compute (~1.5 us) does not cover.
  struct TileRegs { u32x4 k[2], v[2]; float bias; };
  auto gload = [&](int kt0, TileRegs& r) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int cidx = tid + 256 * j, row = cidx >> 3, ch = cidx & 7;
      r.k[j] = u32x4{0u, 0u, 0u, 0u};
      r.v[j] = u32x4{0u, 0u, 0u, 0u};
      if (kt0 + row < N) {
        r.k[j] = *(const u32x4*)(K + (long)(kt0 + row) * p.ld + ch * 8);
        r.v[j] = *(const u32x4*)(V + (long)(kt0 + row) * p.ld + ch * 8);
      }
    }
    r.bias = 0.f;
    if (tid < KT) {
      int key = kt0 + tid;
      r.bias = (key < N && !(kp && kp[key])) ? 0.f : -INFINITY;
    }
  };
  auto lstore = [&](int buf, const TileRegs& r) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int cidx = tid + 256 * j, row = cidx >> 3, ch = cidx & 7;
      *(u32x4*)(Ks[buf] + kswz(row, ch)) = r.k[j];
      *(u32x4*)(Vs[buf] + vswz(row, ch * 8)) = r.v[j];
    }
    if (tid < KT) kbias[buf][tid] = r.bias;
  };
  auto compute = [&](int kt0, int buf) {
    const bf16* Kb = Ks[buf];
    const bf16* Vb = Vs[buf];
    const float* kb = kbias[buf];
    const bool tile_clean = !__any(kb[lane] != 0.f);  // no padded / out-of-range key in this tile
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int k0 = kt0 + sub * 32;
      if (!((k0 < wmlim) || (k0 < wchi && k0 + 32 > wclo))) continue;  // wave-uniform
      const bool full = tile_clean && (k0 + 32 <= wfull);
      f32x16 S;
#pragma unroll
      for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a = *(const bf16x8*)(Kb + kswz(sub * 32 + r32, 2 * s + hh));
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], S, 0, 0, 0);
      }
      float mloc = -INFINITY;
      if (full) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { S[i] *= c; mloc = fmaxf(mloc, S[i]); }
      } else {
        masked_scores(S, c, kb + sub * 32, hh, L.lim - k0, L.clo - k0, L.chi - k0);
#pragma unroll
        for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, S[i]);
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      // Lazy rescale: keep exponentiating against the reference max `mrun` while no row's maximum grew by
      // more than 2^6; p then lies in (0, 64], which fp32 sums and bf16 P (floating point) carry without
      // loss.  Moving the 32 O accumulators AGPR->VGPR->AGPR costs ~80 instructions, so skip it when idle.
      float muse;
      if (__all(mloc <= mrun + 6.0f)) {
        muse = (mrun == -INFINITY) ? 0.f : mrun;
      } else {
        const float mnew = fmaxf(mrun, mloc);
        muse = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = fast_exp2(mrun - muse);
        mrun = mnew;
        lrun *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { O0[i] *= alpha; O1[i] *= alpha; }
      }
      float ls = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = fast_exp2(S[i] - muse); ls += S[i]; }
      if (thr > 0) {
        const uint32_t wbase = (drow + (uint32_t)((k0 >> 1) + 2 * hh)) * HASH_K;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {   // rows i, i+1 are keys 2j, 2j+1: one hash word
          // word index drow + (k0 + row_i)/2 with row_i = (i&3) + 8(i>>2) + 4hh, i even: linear in a compile-time part
          const uint32_t hw = pair_hash_pm(s0, s1, wbase + (uint32_t)(((i & 3) >> 1) + 4 * (i >> 2)) * HASH_K);
          S[i] *= (hw & 0xFFFFu) >= thr ? inv_keep : 0.f;
          S[i + 1] *= (hw >> 16) >= thr ? inv_keep : 0.f;
        }
      }
      lrun += ls;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pb = pack8(S, s2);
        const int krow = sub * 32 + 16 * s2 + 4 * (g >> 1) + tq;
        const int dcol = (g & 1) * 16 + 4 * tp;
        bf16x8 a0 = tr_pair(Vb + vswz(krow, dcol), Vb + vswz(krow + 8, dcol));
        bf16x8 a1 = tr_pair(Vb + vswz(krow, 32 + dcol), Vb + vswz(krow + 8, 32 + dcol));
        O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pb, O0, 0, 0, 0);
        O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pb, O1, 0, 0, 0);
      }
    }
  };

  TileRegs R0, R1;
  int t0 = next_tile(0);
  int t1 = t0 < N ? next_tile(t0 + KT) : N;
  if (t0 < N) gload(t0, R0);
  if (t1 < N) gload(t1, R1);
  if (t0 < N) lstore(0, R0);
  __syncthreads();
  while (t0 < N) {
    // even step: tile t0 in LDS[0], tile t1 in R1; fetch t2 into R0
    int t2 = t1 < N ? next_tile(t1 + KT) : N;
    if (t2 < N) gload(t2, R0);
    compute(t0, 0);
    if (t1 < N) lstore(1, R1);
    __syncthreads();
    if (t1 >= N) break;
    // odd step: tile t1 in LDS[1], tile t2 in R0; fetch t3 into R1
    int t3 = t2 < N ? next_tile(t2 + KT) : N;
    if (t3 < N) gload(t3, R1);
    compute(t1, 1);
    if (t2 < N) lstore(0, R0);
    __syncthreads();
    t0 = t2; t1 = t3;
  }
  const float ltot = lrun + __shfl_xor(lrun, 32, 64);
  const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
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
    if (hh == 0 && p.lse) p.lse[((long)(b * p.H + h)) * N + q] = ltot > 0.f ? (mrun + log2f(ltot)) * LN2 : INFINITY;
  }
}

// =================================================================================================
// backward, pass 1: dQ.  Same loop structure as the forward (one block = 128 queries).
//   S^T = K Q^T ; P^T = exp(S^T - lse) ; dP^T = V dO^T ; dS^T = P^T o (dP^T o drop - delta)
//   dQ^T[d][q] += K^T[d][key] dS^T[key][q]
// =================================================================================================
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) bf16 Ks[2][KT * HD];   // row reads (S^T)
  __shared__ __attribute__((aligned(16))) bf16 Kt[2][KT * HD];   // tr reads (K^T operand of dQ^T)
  __shared__ __attribute__((aligned(16))) bf16 Vs[2][KT * HD];   // row reads (dP^T)
  __shared__ __attribute__((aligned(16))) float kbias[2][KT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qblk0 = (gridDim.x - 1 - blockIdx.x) * QB;
  const int q = qblk0 + wid * 32 + r32;
  const int N = p.N, Nq = p.Nq;
  const int qc = min(q, Nq - 1);
  const bf16* Q = p.q + (long)b * p.sb + h * HD;
  const bf16* K = p.k + (long)b * p.sb + h * HD;
  const bf16* V = p.v + (long)b * p.sb + h * HD;
  const bf16* dO = p.dout + (long)b * p.sbo + h * HD;
  const uint8_t* kp = p.kpad ? p.kpad + (long)b * N : nullptr;
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (q < Nq) {
      qf[s] = *(const bf16x8*)(Q + (long)q * p.ld + 16 * s + 8 * hh);
      dof[s] = *(const bf16x8*)(dO + (long)q * p.ldo + 16 * s + 8 * hh);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { qf[s][j] = f2bf(0.f); dof[s][j] = f2bf(0.f); }
    }
  }
  const long sidx = ((long)(b * p.H + h)) * N + qc;
  const float lse2 = p.lse[sidx] * LOG2E;
  // delta[q] = dO[q] . O[q]: the lane already holds half of dO[q] (the other half sits in lane ^ 32); computed
  // here and written out for the dK/dV pass instead of a separate launch
  float delta = 0.f;
  if (q < Nq) {
    const bf16* Orow = p.o + (long)b * p.sbo + (long)q * p.ldo + h * HD;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ov = *(const bf16x8*)(Orow + 16 * s + 8 * hh);
#pragma unroll
      for (int e = 0; e < 8; ++e) delta = fmaf(bf2f(ov[e]), bf2f(dof[s][e]), delta);
    }
  }
  delta += __shfl_xor(delta, 32, 64);
  if (q < Nq && hh == 0) const_cast<float*>(p.delta)[sidx] = delta;
  const QLimits L = q_limits(qc, p.Tp, p.m, p.r, N);
  int mlim, bclo, bchi;
  tile_ranges(qblk0, min(qblk0 + QB, Nq) - 1, p.Tp, p.m, p.r, N, mlim, bclo, bchi);
  const int wmlim = wave_max_i(L.lim), wfull = wave_min_i(L.lim), wclo = wave_min_i(L.clo), wchi = wave_max_i(L.chi);
  const float c = p.scale * LOG2E;
  const uint32_t thr = drop_threshold(p.p_drop) >> 16;
  const float inv_keep = thr > 0 ? 65536.f / (65536.f - (float)thr) : 1.f;
  const uint32_t s0 = (uint32_t)p.seed, s1 = (uint32_t)(p.seed >> 32);
  const uint32_t Nh = (uint32_t)(N + 1) >> 1;
  const uint32_t drow = ((uint32_t)(b * p.H + h) * (uint32_t)N + (uint32_t)qc) * Nh;
  f32x16 D0, D1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { D0[i] = 0.f; D1[i] = 0.f; }
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  auto next_tile = [&](int kt0) {
    while (kt0 < N && !((kt0 < mlim) || (kt0 < bchi && kt0 + KT > bclo))) kt0 += KT;
    return kt0;
  };
  u32x4 rk[2], rv[2];
  float rbias = 0.f;
  auto gload = [&](int kt0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int cidx = tid + 256 * j, row = cidx >> 3, ch = cidx & 7;
      rk[j] = u32x4{0u, 0u, 0u, 0u};
      rv[j] = u32x4{0u, 0u, 0u, 0u};
      if (kt0 + row < N) {
        rk[j] = *(const u32x4*)(K + (long)(kt0 + row) * p.ld + ch * 8);
        rv[j] = *(const u32x4*)(V + (long)(kt0 + row) * p.ld + ch * 8);
      }
    }
    if (tid < KT) {
      int key = kt0 + tid;
      rbias = (key < N && !(kp && kp[key])) ? 0.f : -INFINITY;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int cidx = tid + 256 * j, row = cidx >> 3, ch = cidx & 7;
      *(u32x4*)(Ks[buf] + kswz(row, ch)) = rk[j];
      *(u32x4*)(Kt[buf] + vswz(row, ch * 8)) = rk[j];
      *(u32x4*)(Vs[buf] + kswz(row, ch)) = rv[j];
    }
    if (tid < KT) kbias[buf][tid] = rbias;
  };
  int cur = next_tile(0), buf = 0;
  if (cur < N) { gload(cur); lstore(0); }
  __syncthreads();
  while (cur < N) {
    const int kt0 = cur;
    const int nxt = next_tile(cur + KT);
    if (nxt < N) gload(nxt);
    const bf16* Kb = Ks[buf];
    const bf16* Ktb = Kt[buf];
    const bf16* Vb = Vs[buf];
    const float* kb = kbias[buf];
    const bool tile_clean = !__any(kb[lane] != 0.f);  // no padded / out-of-range key in this tile
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int k0 = kt0 + sub * 32;
      if (!((k0 < wmlim) || (k0 < wchi && k0 + 32 > wclo))) continue;  // wave-uniform
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a = *(const bf16x8*)(Kb + kswz(sub * 32 + r32, 2 * s + hh));
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], S, 0, 0, 0);
        bf16x8 av = *(const bf16x8*)(Vb + kswz(sub * 32 + r32, 2 * s + hh));
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dof[s], dP, 0, 0, 0);
      }
      if (thr > 0) {
        const uint32_t wbase = (drow + (uint32_t)((k0 >> 1) + 2 * hh)) * HASH_K;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          // word index drow + (k0 + row_i)/2 with row_i = (i&3) + 8(i>>2) + 4hh, i even: linear in a compile-time part
          const uint32_t hw = pair_hash_pm(s0, s1, wbase + (uint32_t)(((i & 3) >> 1) + 4 * (i >> 2)) * HASH_K);
          dP[i] *= (hw & 0xFFFFu) >= thr ? inv_keep : 0.f;
          dP[i + 1] *= (hw >> 16) >= thr ? inv_keep : 0.f;
        }
      }
      if (tile_clean && (k0 + 32 <= wfull)) {   // every query of the wave sees all 32 keys, none padded
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] *= c;
      } else {
        masked_scores(S, c, kb + sub * 32, hh, L.lim - k0, L.clo - k0, L.chi - k0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float pe = fast_exp2(S[i] - lse2);
        S[i] = pe * (dP[i] - delta) * p.scale;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 db = pack8(S, s2);
        const int krow = sub * 32 + 16 * s2 + 4 * (g >> 1) + tq;
        const int dcol = (g & 1) * 16 + 4 * tp;
        bf16x8 a0 = tr_pair(Ktb + vswz(krow, dcol), Ktb + vswz(krow + 8, dcol));
        bf16x8 a1 = tr_pair(Ktb + vswz(krow, 32 + dcol), Ktb + vswz(krow + 8, 32 + dcol));
        D0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, db, D0, 0, 0, 0);
        D1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, db, D1, 0, 0, 0);
      }
    }
    if (nxt < N) lstore(buf ^ 1);
    __syncthreads();
    cur = nxt;
    buf ^= 1;
  }
  if (q < Nq) {
    bf16* drow_p = p.dq + (long)b * p.sb + (long)q * p.ld + h * HD;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      bf16x4 v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = f2bf(D0[4 * gq + e]); v1[e] = f2bf(D1[4 * gq + e]); }
      *(bf16x4*)(drow_p + 8 * gq + 4 * hh) = v0;
      *(bf16x4*)(drow_p + 32 + 8 * gq + 4 * hh) = v1;
    }
  }
}

// =================================================================================================
// backward, pass 2: dK, dV.  One block = 128 keys (4 waves x 32), loops over the query tiles
// that can see them.  S (not transposed): queries on accumulator rows, keys on lanes.
//   S = Q K^T ; P = exp(S - lse[q]) ; dP = dO V^T ; dS = P o (dP o drop - delta[q]) * scale
//   dV^T[d][key] += dO^T[d][q] (P o drop)[q][key]      dK^T[d][key] += Q^T[d][q] dS[q][key]
// =================================================================================================
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnP p) {
  constexpr int QT = 64;  // queries per LDS tile
  __shared__ __attribute__((aligned(16))) bf16 Qs[2][QT * HD];   // row reads
  __shared__ __attribute__((aligned(16))) bf16 Qt[2][QT * HD];   // tr reads
  __shared__ __attribute__((aligned(16))) bf16 Ds[2][QT * HD];   // dO row reads
  __shared__ __attribute__((aligned(16))) bf16 Dt[2][QT * HD];   // dO tr reads
  __shared__ __attribute__((aligned(16))) float lse_s[2][QT];
  __shared__ __attribute__((aligned(16))) float del_s[2][QT];
  __shared__ __attribute__((aligned(16))) int qlim_s[2][QT], qclo_s[2][QT], qchi_s[2][QT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kblk0 = blockIdx.x * QB;
  const int key = kblk0 + wid * 32 + r32;
  const int N = p.N, Nq = p.Nq;
  const bf16* Q = p.q + (long)b * p.sb + h * HD;
  const bf16* K = p.k + (long)b * p.sb + h * HD;
  const bf16* V = p.v + (long)b * p.sb + h * HD;
  const bf16* dO = p.dout + (long)b * p.sbo + h * HD;
  const bool key_ok = key < N && !(p.kpad && p.kpad[(long)b * N + key]);
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (key < N) {
      kf[s] = *(const bf16x8*)(K + (long)key * p.ld + 16 * s + 8 * hh);
      vf[s] = *(const bf16x8*)(V + (long)key * p.ld + 16 * s + 8 * hh);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { kf[s][j] = f2bf(0.f); vf[s][j] = f2bf(0.f); }
    }
  }
  const float c = p.scale * LOG2E;
  const uint32_t thr = drop_threshold(p.p_drop) >> 16;
  const float inv_keep = thr > 0 ? 65536.f / (65536.f - (float)thr) : 1.f;
  const uint32_t s0 = (uint32_t)p.seed, s1 = (uint32_t)(p.seed >> 32);
  const uint32_t Nh = (uint32_t)(N + 1) >> 1;
  const uint32_t dbase = (uint32_t)(b * p.H + h) * (uint32_t)N;
  const uint32_t khalf = (uint32_t)min(key, N - 1) >> 1, kodd = (uint32_t)min(key, N - 1) & 1u;
  const int klo = kblk0, khi = min(kblk0 + QB, N);  // this block's keys [klo, khi)
  const int wk0 = kblk0 + wid * 32, wk1 = min(wk0 + 32, N);  // this wave's keys [wk0, wk1)
  const bool wave_keys_ok = __all(key_ok);                     // none of them padded or past the end
  f32x16 dV0, dV1, dK0, dK1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dV0[i] = dV1[i] = dK0[i] = dK1[i] = 0.f; }
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  auto next_tile = [&](int q0) {  // first query tile at or after q0 that sees any key of this block
    while (q0 < Nq) {
      int mlim, bclo, bchi;
      tile_ranges(q0, min(q0 + QT, Nq) - 1, p.Tp, p.m, p.r, N, mlim, bclo, bchi);
      if ((klo < mlim) || (klo < bchi && khi > bclo)) break;
      q0 += QT;
    }
    return min(q0, N) >= Nq ? N : q0;   // past the last query tile: the loops below test against N
  };
  u32x4 rq[2], rd[2];
  float rlse = 0.f, rdel = 0.f;
  int rlim = 0, rclo = 0, rchi = 0;
  auto gload = [&](int q0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int cidx = tid + 256 * j, row = cidx >> 3, ch = cidx & 7;
      rq[j] = u32x4{0u, 0u, 0u, 0u};
      rd[j] = u32x4{0u, 0u, 0u, 0u};
      if (q0 + row < Nq) {
        rq[j] = *(const u32x4*)(Q + (long)(q0 + row) * p.ld + ch * 8);
        rd[j] = *(const u32x4*)(dO + (long)(q0 + row) * p.ldo + ch * 8);
      }
    }
    if (tid < QT) {
      int qq = q0 + tid;
      int qcl = min(qq, Nq - 1);
      rlse = (qq < Nq) ? p.lse[(long)(b * p.H + h) * N + qq] * LOG2E : INFINITY;  // +inf -> P = 0
      rdel = (qq < Nq) ? p.delta[(long)(b * p.H + h) * N + qq] : 0.f;
      QLimits L = q_limits(qcl, p.Tp, p.m, p.r, N);
      rlim = L.lim; rclo = L.clo; rchi = L.chi;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int cidx = tid + 256 * j, row = cidx >> 3, ch = cidx & 7;
      *(u32x4*)(Qs[buf] + kswz(row, ch)) = rq[j];
      *(u32x4*)(Qt[buf] + vswz(row, ch * 8)) = rq[j];
      *(u32x4*)(Ds[buf] + kswz(row, ch)) = rd[j];
      *(u32x4*)(Dt[buf] + vswz(row, ch * 8)) = rd[j];
    }
    if (tid < QT) {
      lse_s[buf][tid] = rlse; del_s[buf][tid] = rdel;
      qlim_s[buf][tid] = rlim; qclo_s[buf][tid] = rclo; qchi_s[buf][tid] = rchi;
    }
  };
  int cur = next_tile(0), buf = 0;
  if (cur < N) { gload(cur); lstore(0); }
  __syncthreads();
  while (cur < N) {
    const int q0 = cur;
    const int nxt = next_tile(cur + QT);
    if (nxt < N) gload(nxt);
    const bf16* Qb = Qs[buf]; const bf16* Qtb = Qt[buf]; const bf16* Db = Ds[buf]; const bf16* Dtb = Dt[buf];
    const float* lse_b = lse_s[buf]; const float* del_b = del_s[buf];
    const int* qlim_b = qlim_s[buf]; const int* qclo_b = qclo_s[buf]; const int* qchi_b = qchi_s[buf];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      bool sub_full = false;
      {  // can any query of this 32-row sub-tile see any key of this wave?  (wave-uniform)
        const int qs0 = q0 + sub * 32;
        if (qs0 >= Nq || wk0 >= N) continue;
        const int qs1 = min(qs0 + 32, Nq) - 1;
        int smlim, sclo, schi;
        tile_ranges(qs0, qs1, p.Tp, p.m, p.r, N, smlim, sclo, schi);
        if (!((wk0 < smlim) || (wk0 < schi && wk1 > sclo))) continue;
        // smallest main-key limit over the sub-tile's queries (both query kinds are ordered by block)
        int minlim = 0;
        if (qs1 < p.Tp) minlim = min((qs0 / p.m + 1) * p.m, p.Tp);
        else if (qs0 >= p.Tp && p.r > 0) minlim = min(((qs0 - p.Tp) / p.r + 1) * p.m, p.Tp);
        sub_full = wave_keys_ok && qs0 + 32 <= Nq && wk0 + 32 <= minlim;
      }
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 a = *(const bf16x8*)(Qb + kswz(sub * 32 + r32, 2 * s + hh));
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[s], S, 0, 0, 0);
        bf16x8 ad = *(const bf16x8*)(Db + kswz(sub * 32 + r32, 2 * s + hh));
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[s], dP, 0, 0, 0);
      }
      // per-query scalars of this lane's accumulator rows come as 16-byte LDS reads, four consecutive rows each
      // (the per-element form branched around five dependent ds_read_b32 per score)
      typedef __attribute__((ext_vector_type(4))) int i32x4;
      const uint32_t stepK = Nh * HASH_K;
      const uint32_t wsub = ((dbase + (uint32_t)(q0 + sub * 32 + 4 * hh)) * Nh + khalf) * HASH_K;
      f32x16 Pd;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int rb = sub * 32 + 8 * g4 + 4 * hh;
        const f32x4 lse4 = *(const f32x4*)(lse_b + rb);
        const f32x4 del4 = *(const f32x4*)(del_b + rb);
        float pe[4];
        if (sub_full) {   // every query of the sub-tile sees every key of this wave, no padded key: no mask at all
#pragma unroll
          for (int e = 0; e < 4; ++e) pe[e] = fast_exp2(fmaf(S[4 * g4 + e], c, -lse4[e]));
        } else {
          const i32x4 lim4 = *(const i32x4*)(qlim_b + rb), clo4 = *(const i32x4*)(qclo_b + rb), chi4 = *(const i32x4*)(qchi_b + rb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = key_ok & ((key < lim4[e]) | ((key >= clo4[e]) & (key < chi4[e])));
            const float v = fast_exp2(fmaf(S[4 * g4 + e], c, -lse4[e]));
            pe[e] = ok ? v : 0.f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g4 + e;
          float ks = 1.f;
          if (thr > 0) {   // word ((dbase + query) * Nh + key/2): consecutive rows are stepK apart (rows past N carry P = 0)
            const uint32_t hw = pair_hash_pm(s0, s1, wsub + (uint32_t)(8 * g4 + e) * stepK);
            ks = ((kodd ? (hw >> 16) : (hw & 0xFFFFu)) >= thr) ? inv_keep : 0.f;
          }
          Pd[i] = pe[e] * ks;
          S[i] = pe[e] * (dP[i] * ks - del4[e]) * p.scale;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pb = pack8(Pd, s2), sb = pack8(S, s2);
        const int qrow = sub * 32 + 16 * s2 + 4 * (g >> 1) + tq;
        const int dcol = (g & 1) * 16 + 4 * tp;
        bf16x8 d0 = tr_pair(Dtb + vswz(qrow, dcol), Dtb + vswz(qrow + 8, dcol));
        bf16x8 d1 = tr_pair(Dtb + vswz(qrow, 32 + dcol), Dtb + vswz(qrow + 8, 32 + dcol));
        dV0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, pb, dV0, 0, 0, 0);
        dV1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, pb, dV1, 0, 0, 0);
        bf16x8 q0f = tr_pair(Qtb + vswz(qrow, dcol), Qtb + vswz(qrow + 8, dcol));
        bf16x8 q1f = tr_pair(Qtb + vswz(qrow, 32 + dcol), Qtb + vswz(qrow + 8, 32 + dcol));
        dK0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0f, sb, dK0, 0, 0, 0);
        dK1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1f, sb, dK1, 0, 0, 0);
      }
    }
    if (nxt < N) lstore(buf ^ 1);
    __syncthreads();
    cur = nxt;
    buf ^= 1;
  }
  if (key < N) {
    bf16* dkr = p.dk + (long)b * p.sb + (long)key * p.ld + h * HD;
    bf16* dvr = p.dv + (long)b * p.sb + (long)key * p.ld + h * HD;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      bf16x4 a0, a1, b0, b1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] = f2bf(dK0[4 * gq + e]); a1[e] = f2bf(dK1[4 * gq + e]);
        b0[e] = f2bf(dV0[4 * gq + e]); b1[e] = f2bf(dV1[4 * gq + e]);
      }
      *(bf16x4*)(dkr + 8 * gq + 4 * hh) = a0;
      *(bf16x4*)(dkr + 32 + 8 * gq + 4 * hh) = a1;
      *(bf16x4*)(dvr + 8 * gq + 4 * hh) = b0;
      *(bf16x4*)(dvr + 32 + 8 * gq + 4 * hh) = b1;
    }
  }
}

// -------------------------------------------------------------------------------------------------
static int attn_fill(const AttnDesc& d, AttnP& p) {
  p.q = (const bf16*)d.q; p.k = (const bf16*)d.k; p.v = (const bf16*)d.v; p.o = (bf16*)d.o; p.lse = d.lse; p.kpad = d.kpad;
  p.dout = (const bf16*)d.dout; p.delta = d.delta; p.dq = (bf16*)d.dq; p.dk = (bf16*)d.dk; p.dv = (bf16*)d.dv;
  p.ld = d.ld; p.ldo = d.ldo; p.sb = d.sb; p.sbo = d.sbo; p.B = d.B; p.H = d.H; p.N = d.N; p.Tp = d.Tp; p.m = d.m; p.r = d.r;
  p.Nq = d.Nq > 0 ? d.Nq : d.N;
  p.scale = d.scale; p.p_drop = d.p_drop; p.seed = d.seed;
  p.drop_bits = (uint32_t*)d.drop_bits;
  p.mq = d.mq; p.ldq = d.mq > 0 ? d.ldq : d.ld; p.sbq = d.mq > 0 ? d.sbq : d.sb; p.Ns = d.mq > 0 ? p.Nq : d.N;
  p.mg_m = div_magic(d.m); p.mg_r = div_magic(d.r); p.mg_mq = div_magic(d.mq); p.mg_H = div_magic(d.H);
  p.thr16 = drop_threshold(d.p_drop) >> 16;
  if (d.mq < 0) return set_error("attention: mq must be >= 0");
  if (d.mq > 0) {
    if (d.r != 0 || d.Tp != d.N) return set_error("attention (cross mode): needs r == 0 and Tp == N");
    if (d.Nq <= 0 || (p.ldq % 8) || (p.sbq % 8)) return set_error("attention (cross mode): Nq > 0, ldq and sbq multiples of 8");
    if (!attn2_ok(p) || (p.Nq + 31) / 32 > 512) return set_error("attention (cross mode): too many tiles");
    if ((long)p.B * p.H * p.Ns * (long)((p.N + 1) / 2) >= (1L << 32)) return set_error("attention: B*H*Nq*N/2 must be < 2^32 (dropout index)");
  }
  if (!p.q || !p.k || !p.v || !p.o || !p.lse) return set_error("attention: null pointer");
  if (d.head_dim != HD) return set_error("attention: only head_dim 64 is built");
  if (p.B <= 0 || p.H <= 0 || p.N <= 0) return set_error("attention: bad B/H/N");
  if (p.m <= 0 || p.r < 0 || p.Tp <= 0 || p.Tp > p.N) return set_error("attention: bad block structure (m>0, r>=0, 0<Tp<=N)");
  if (p.N != p.Tp + (p.Tp / p.m) * p.r) return set_error("attention: N must equal Tp + (Tp/m)*r");
  if (p.mq == 0 && p.Nq != p.N && p.Nq > p.Tp) return set_error("attention: Nq must be N (all queries) or <= Tp (main frames only)");
  if ((p.ld % 8) || (p.ldo % 8) || (p.sb % 8) || (p.sbo % 8)) return set_error("attention: strides must be multiples of 8 elements");
  if (p.p_drop < 0.f || p.p_drop >= 1.f) return set_error("attention: dropout must be in [0,1)");
  if ((long)p.B * p.H * p.N * (long)((p.N + 1) / 2) >= (1L << 32)) return set_error("attention: B*H*N*N/2 must be < 2^32 (dropout index)");
  return 0;
}

// W2VS_ATTN_V1=1 selects the round-1 kernels of this file (128-query workgroups) for A/B runs; default = attention2.hip
static int g_attn_variant = -1;      // -1: environment / default, 1: attention.hip kernels, 2: attention2.hip kernels
void attn_tune(int variant) { g_attn_variant = variant; }
static bool use_v2(const AttnP& p) {
  static const bool v1 = W2VS_ENV_INT("W2VS_ATTN_V1", 0) != 0;
  const bool want_v1 = g_attn_variant == 1 || (g_attn_variant < 0 && v1);
  return p.mq > 0 || (!want_v1 && attn2_ok(p));
}

int attn_fwd(const AttnDesc& d, hipStream_t st) {
  AttnP p{};
  if (int e = attn_fill(d, p)) return e;
  if (use_v2(p)) return attn2_fwd(p, st);
  dim3 grid((p.Nq + QB - 1) / QB, p.H, p.B);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "attn_fwd");
}

int attn_bwd(const AttnDesc& d, hipStream_t st) {
  AttnP p{};
  if (int e = attn_fill(d, p)) return e;
  if (!p.dout || !p.delta || !p.dq || !p.dk || !p.dv) return set_error("attn_bwd: null pointer");
  if (use_v2(p)) return attn2_bwd(p, st);
  dim3 gridq((p.Nq + QB - 1) / QB, p.H, p.B), gridk((p.N + QB - 1) / QB, p.H, p.B);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, gridq, dim3(256), 0, st, p);     // dq rows >= Nq are not written
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, gridk, dim3(256), 0, st, p);
  return hip_check(hipGetLastError(), "attn_bwd");
}

}  // namespace w2vs
