// bf16 MFMA GEMMs for the wav2vec-S hot path (gfx950).
//
//   NT:  C[M,N] = epi( A[M,K] . B[N,K]^T )        linear fwd, dgrad (with pre-transposed W),
//                                                  conv1-6 fwd/dgrad as strided-A GEMM
//   TN:  C[M,N] (+)= A[K,M]^T . B[K,N]            weight gradients (contraction over tokens)
//
// Layout notes
//  * A rows may OVERLAP: row m starts at element (a_off + m*lda).  With channel-last
//    activations [L, C] a 1-D conv (kernel k, stride s, no padding) is exactly the NT
//    GEMM with lda = s*C and K = k*C - no im2col buffer exists anywhere.
//  * All global reads go through buffer descriptors sized to the valid extent, so tile
//    overhang (row >= M, k >= K, and the "row -1" of the k3/s2 dgrad form) reads zeros
//    in hardware instead of branching.
//  * 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
//    LDS rows are 128 B; 16-B chunk c of row r lives at chunk (c ^ ((r>>1)&7)) which makes
//    the ds_read_b128 fragment reads and the ds_write_b128 staging writes conflict free.
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <type_traits>
#include <vector>
#include "common.h"
#include "w2vs_internal.h"

namespace w2vs {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_ELEMS = BM * BK;  // == BN*BK

// Tile order.  (1) Workgroups are dealt round-robin over the 8 XCDs (ids equal mod 8 share an L2), so each
// XCD is given a CONTIGUOUS run of tile ids.  (2) Inside that run, ids walk GM M-tiles x all N-tiles
// "group-M" fashion (m fastest within a group of GM), so the ~64-96 blocks in flight on an XCD touch
// only ~GM A-panels and a few B-panels: their footprint stays inside the 4 MiB L2 instead of streaming
// the whole B operand once per M-tile (measured with FETCH_SIZE: 3.5-5x the algorithmic bytes before).
__device__ __forceinline__ void tile_order(int orig, int ntm, int ntn, int GM, int& tm, int& tn) {
  const int nwg = ntm * ntn;
  const int qd = nwg >> 3, rm = nwg & 7, xcd = orig & 7;
  const int id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
  const int per_group = GM * ntn;
  const int group = id / per_group, first_m = group * GM;
  const int gsz = min(ntm - first_m, GM);
  const int in_group = id - group * per_group;
  tm = first_m + in_group % gsz;
  tn = in_group / gsz;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

struct GemmP {
  const bf16* A; const bf16* B; bf16* C; bf16* C2; float* Cf;
  const bf16* bias; const bf16* aux;
  int M, N, K;
  long lda, ldb, ldc;
  long a_off;                 // element offset of row 0 (may be negative)
  long sA, sB, sC;            // batch strides (elements), blockIdx.z
  uint32_t a_bytes, b_bytes;  // valid bytes per batch for the A / B descriptors
  long c_elems;               // valid output elements per batch
  int k_split, n_split;       // TN only: K range per split, splits per batch
  float* colsum;              // TN only: optional out[m] += sum_k A[k, m]  (bias gradient)
  float* slab;                // TN loader/consumer: partial tiles [grid.z][M][N] instead of atomics into Cf
  float alpha;
  int overwrite;              // TN single-writer kernels: Cf = alpha * acc instead of Cf += (the first writer of a zero-free arena)
  int zk_col, zk_kt;          // NT persistent kernels: tiles with n0 >= zk_col start at K tile zk_kt (B is zero before it); 0 = off
#ifdef W2VS_ABLATION
  // tuning build only: in-kernel time stamps of the 8-phase NT kernel (s_memrealtime, 100 MHz), 8 words per wave group and
  // workgroup: entry, first tile landed, end of each tile's K loop / epilogue issue (first two tiles), all stores done
  unsigned long long* stamps;
  int epi_dbg;                // timing-only epilogue ablations: bit 0 no GELU arithmetic, bit 1 no second (gelu') store, bit 2 no store at all
#endif
};
#ifdef W2VS_ABLATION
static unsigned long long* g_nt_stamps = nullptr;
static int g_epi_dbg = 0;
extern "C" void w2vs_dbg_nt_stamps(void* buf, int epi_dbg) { g_nt_stamps = (unsigned long long*)buf; g_epi_dbg = epi_dbg; }
#define W2VS_STAMP(p, slot)                                                                                       \
  do {                                                                                                            \
    if ((p).stamps && lane == 0 && wc == 0) (p).stamps[((long)blockIdx.x * 2 + wr) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define W2VS_STAMP(p, slot) do { } while (0)
#endif

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_GELU_SAVE = 3, EPI_DGELU = 4, EPI_F32 = 5, EPI_ADD = 6,
       EPI_BIAS_GELU_SAVEG = 7,   // C = gelu(pre), C2 = gelu'(pre): the backward then needs one multiply, not an erf
       EPI_MUL = 8 };             // C = acc * aux (aux = the saved gelu')
constexpr bool epi_has_bias(int e) { return e == EPI_BIAS || e == EPI_BIAS_GELU || e == EPI_BIAS_GELU_SAVE || e == EPI_BIAS_GELU_SAVEG; }
constexpr bool epi_is_gelu(int e) { return e == EPI_BIAS_GELU || e == EPI_BIAS_GELU_SAVE || e == EPI_BIAS_GELU_SAVEG; }
constexpr bool epi_is_save(int e) { return e == EPI_BIAS_GELU_SAVE || e == EPI_BIAS_GELU_SAVEG; }
constexpr bool epi_is_dact(int e) { return e == EPI_DGELU || e == EPI_MUL; }
// gelu(x) and gelu'(x) from one evaluation of the normal tail (common.h)
__device__ __forceinline__ void gelu_pair(float x, float& gv, float& dv) {
  float e;
  const float cdf = normal_cdf(x, e);
  gv = x * cdf;
  dv = fmaf(x * 0.3989422804014327f, e, cdf);
}

// ---- register-direct epilogue of the DMA kernels (operands swapped: the accumulators hold C^T fragments) --------------
// Lane (fr, fq) owns, per 16-row block b of its wave's rows, row row0 + rel_row(b) and - from the two adjacent 16-column
// tiles of column group u - the EIGHT consecutive columns col_of(u) .. +7: one 16-byte access per block and operand.
// Everything goes through buffer instructions; a lane outside the matrix (row >= M, col >= N, or past c_elems) gets an
// out-of-range offset, so its loads return zeros and its stores are dropped.  No exec branches: the side operand of EVERY
// block (saved gelu' / residual, 16 B per lane and block) is requested before the first one is used.  With a branch per
// block hipcc emitted load -> s_waitcnt vmcnt(0) -> store twenty times in a row: +10 us on an fc2 dgrad launch, +30-60 us on
// the conv dgrads.  The accumulators are packed to bf16 first (the reference rounds the GEMM result before the pointwise
// op as well), which frees the registers the outstanding loads land in.
//   val(u, b, e): fp32 accumulator e (0..7) of column group u, block b.   Requires ldc <= 2^21 (host check).
template <int EPI, int NU, int NB, bool BIAS_VEC, class ColOf, class RowBlk, class Val>
__device__ __forceinline__ void epi_direct(const GemmP& p, int bz, int row0_, int fr_, ColOf col_of, RowBlk row_blk, Val val) {
  static_assert(EPI != EPI_F32, "fp32 outputs keep their own loop");
  // rows of block b: row0 + row_blk(b) + fr, row_blk(b) wave-uniform.  fr goes through an opaque asm so that nothing below can
  // be hoisted in front of the K loop as a per-lane value: the 320-row 8-phase kernel has no register to spare there.
  int fr = fr_;
  asm volatile("" : "+v"(fr));
  const int row0 = __builtin_amdgcn_readfirstlane(row0_);
  const int ldc = (int)p.ldc;
  const long base = (long)bz * p.sC + (long)row0 * p.ldc;               // wave-uniform element offset of (row0, 0)
  long room = p.c_elems - (long)row0 * p.ldc;                          // valid elements of this batch from row0 on
  room = room < 0 ? 0 : (room > 0x3FFFFFF8L ? 0x3FFFFFF8L : room);
  const int lim = (int)room;
  const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C + base, (uint32_t)lim * 2u);
  const int rows_left = p.M - row0;
  // byte offset of block (u, b) for this lane, or an out-of-range one; recomputed where it is used (3 VALU) rather than kept
  auto offs = [&](int frx, int col, int b) -> uint32_t {
    const int rb = row_blk(b);                                         // scalar
    const int o = frx * ldc + col + rb * ldc;
    const bool ok = frx < rows_left - rb && col < p.N && o + 8 <= lim;
    return ok ? (uint32_t)o * 2u : 0x80000000u;
  };
  union V8 { bf16x8 h; u32x4 w; };
  if constexpr (epi_is_dact(EPI) || EPI == EPI_ADD) {
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.aux + base, (uint32_t)lim * 2u);
    V8 pk[NU][NB], ax[NU][NB];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[u][b].h[e] = f2bf(val(u, b, e));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NB; ++b) ax[u][b].w = __builtin_amdgcn_raw_buffer_load_b128(ra, offs(fr, col_of(u), b), 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    int fr2 = fr_;
    asm volatile("" : "+v"(fr2));                                      // the store offsets are not kept from the load phase
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        V8 o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = bf2f(pk[u][b].h[e]), a = bf2f(ax[u][b].h[e]);
          o8.h[e] = f2bf(EPI == EPI_ADD ? g + a : g * (EPI == EPI_MUL ? a : gelu_grad(a)));
        }
        __builtin_amdgcn_raw_buffer_store_b128(o8.w, rc, offs(fr2, col_of(u), b), 0, 0);
      }
  } else {
    __amdgpu_buffer_rsrc_t rc2 = rc;
    if constexpr (epi_is_save(EPI)) rc2 = make_rsrc(p.C2 + base, (uint32_t)lim * 2u);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int col = col_of(u);
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if constexpr (epi_has_bias(EPI)) {
        if (p.bias != nullptr && col < p.N) {
          if constexpr (BIAS_VEC) {
            const bf16x8 b8 = *(const bf16x8*)(p.bias + col);          // N % 8 == 0, bias 16-byte aligned (host check)
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = bf2f(b8[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = bf2f(p.bias[col + e]); // N % 8 == 0 (host check)
          }
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        V8 o8;
        if constexpr (epi_is_gelu(EPI)) {
          V8 pre;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = val(u, b, e) + bv[e];
            pre.h[e] = f2bf(v);
            float gv, dv;
#ifdef W2VS_ABLATION
            if (p.epi_dbg & 1) { gv = v; dv = v + 1.f; } else
#endif
            gelu_pair(epi_is_save(EPI) ? bf2f(pre.h[e]) : v, gv, dv);  // the activation of the value that is saved
            o8.h[e] = f2bf(gv);
            if (EPI == EPI_BIAS_GELU_SAVEG) pre.h[e] = f2bf(dv);
          }
#ifdef W2VS_ABLATION
          if (p.epi_dbg & 6) { if (p.epi_dbg & 4) asm volatile("" :: "v"(pre.w), "v"(o8.w)); else { asm volatile("" :: "v"(pre.w)); __builtin_amdgcn_raw_buffer_store_b128(o8.w, rc, offs(fr, col, b), 0, 0); } continue; }
#endif
          if constexpr (epi_is_save(EPI)) __builtin_amdgcn_raw_buffer_store_b128(pre.w, rc2, offs(fr, col, b), 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o8.h[e] = f2bf(val(u, b, e) + bv[e]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(o8.w, rc, offs(fr, col, b), 0, 0);
      }
    }
  }
}

template <int NU, int NB, class ColOf, class RowBlk, class Val>
__device__ __forceinline__ void epi_direct_f32(const GemmP& p, int bz, int row0, int fr, ColOf col_of, RowBlk row_blk, Val val) {
  float* Cf = p.Cf + (long)bz * p.sC;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int col = col_of(u);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int row = row0 + row_blk(b) + fr;
      if (row >= p.M || col >= p.N) continue;
      float* dst = Cf + (long)row * p.ldc + col;
      *(f32x4*)dst = f32x4{val(u, b, 0) * p.alpha, val(u, b, 1) * p.alpha, val(u, b, 2) * p.alpha, val(u, b, 3) * p.alpha};
      *(f32x4*)(dst + 4) = f32x4{val(u, b, 4) * p.alpha, val(u, b, 5) * p.alpha, val(u, b, 6) * p.alpha, val(u, b, 7) * p.alpha};
    }
  }
}

// SB = single LDS buffer: 34 KiB per block instead of 68, which lets THREE blocks share a CU (12 waves);
// the tile for step kt+1 waits in registers while step kt computes, at the price of a second barrier.
//   MODE 0: register-staged, double LDS buffer (2 blocks/CU)
//   MODE 1: register-staged, single LDS buffer (3 blocks/CU)
//   MODE 2: LDS-DMA (buffer_load ... lds): global -> LDS without VGPRs or ds_write; PMC showed the
//           register-staged forms LDS-bound (ds_write_b128 costs ~13 LDS cycles against 4 for a
//           ds_read_b128: 12 waves x (16 reads + 8 writes) per K-tile > the MFMA time of that K-tile)
template <int EPI, int MODE>
__global__ __launch_bounds__(256, MODE == 1 ? 3 : 2) void gemm_nt_kernel(GemmP p) {
  constexpr bool SB = MODE == 1;
  constexpr bool DMA = MODE == 2;
  // A0 A1 B0 B1 during the K loop (64 KiB; SB: A B, 32 KiB); padded [128][136] output tile(s) in the epilogue
  __shared__ __attribute__((aligned(16))) bf16 lds[MODE == 0 ? 2 * 128 * 136 : (MODE == 1 ? 128 * 136 : 4 * TILE_ELEMS)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (id % 8 share an L2), so give
  // every XCD a contiguous run of tiles - the N-tiles of one M-tile then share that XCD's L2 copy of A.
  int tm_, tn_;
  tile_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.y, gridDim.x, 8, tm_, tn_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const int bz = blockIdx.z;

  const bf16* Ab = p.A + (long)bz * p.sA;
  const bf16* Bb = p.B + (long)bz * p.sB;
  __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, p.a_bytes);
  __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb, p.b_bytes);

  // staging assignment: thread t owns chunks t + 256*j (row = c>>3, kc = c&7), j = 0..3
  // Byte offsets are kept modulo 2^32: a row that starts before the buffer (the "row -1" of
  // the k3/s2 dgrad form) wraps to a huge offset = out of range = zeros, and walks back
  // into range exactly where its window re-enters the buffer.
  uint32_t a_goff[4], b_goff[4];
  int l_off[4];
  bool a_ok[4], b_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = tid + 256 * j, row = c >> 3, kc = c & 7;
    long ae = p.a_off + (long)(m0 + row) * p.lda + kc * 8;
    a_ok[j] = (m0 + row < p.M);  // rows >= M must read zero even if still inside the descriptor
    a_goff[j] = (uint32_t)(ae * 2);
    long be = (long)(n0 + row) * p.ldb + kc * 8;
    b_ok[j] = (n0 + row < p.N);
    b_goff[j] = (uint32_t)(be * 2);
    l_off[j] = swz(row, kc);
  }
  const int nk = (p.K + BK - 1) / BK;
  // the k tail (K % 64 != 0) is handled by clamping per-chunk: chunk kc of tile kt is valid iff kt*64+kc*8 < K
  u32x4 ra_reg[4], rb_reg[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int kc = (tid + 256 * j) & 7;
      bool kok = (kt * BK + kc * 8) < p.K;
      uint32_t ao = (kok && a_ok[j]) ? a_goff[j] + (uint32_t)(kt * BK * 2) : 0xFFFFFFF0u;
      uint32_t bo = (kok && b_ok[j]) ? b_goff[j] + (uint32_t)(kt * BK * 2) : 0xFFFFFFF0u;
      ra_reg[j] = buf_load16(ra, ao);
      rb_reg[j] = buf_load16(rb, bo);
    }
  };
  auto lstore = [&](int buf) {
    bf16* sa = lds + buf * TILE_ELEMS;
    bf16* sb = lds + ((SB ? 1 : 2) + buf) * TILE_ELEMS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(u32x4*)(sa + l_off[j]) = ra_reg[j];
      *(u32x4*)(sb + l_off[j]) = rb_reg[j];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- LDS-DMA staging: wave w fills rows [32w, 32w+32) of the A and of the B tile, 8 rows (1 KiB) per
  // instruction.  The LDS image is written linearly (lane l -> row l>>3, physical chunk l&7), so the
  // XOR swizzle goes on the SOURCE: that lane fetches logical chunk (l&7) ^ ((row>>1)&7).
  uint32_t da_off[4], db_off[4];
  bool da_ok[4], db_ok[4];
  if (DMA) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wid * 32 + j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      // B rows are stored PERMUTED (LDS row w*64 + jj*16 + f <- B row w*64 + (jj>>1)*32 + (f>>2)*8 + (jj&1)*4 + (f&3)):
      // with the MFMA operands swapped a lane then owns 8 consecutive output columns (tiles 2u, 2u+1) of one row and
      // the epilogue stores 16 bytes straight from the registers - no LDS staging, no barriers
      const int f = row & 15, jj = (row >> 4) & 3;
      const int brow = (row & 64) + (jj >> 1) * 32 + (f >> 2) * 8 + (jj & 1) * 4 + (f & 3);
      da_ok[j] = (m0 + row < p.M) && (c * 8 < p.K);
      db_ok[j] = (n0 + brow < p.N) && (c * 8 < p.K);
      da_off[j] = (uint32_t)((p.a_off + (long)(m0 + row) * p.lda + c * 8) * 2);
      db_off[j] = (uint32_t)(((long)(n0 + brow) * p.ldb + c * 8) * 2);
    }
  }
  auto dma_issue = [&](int kt, int buf) {
    bf16* sa = lds + buf * TILE_ELEMS + wid * 32 * 64;
    bf16* sb = lds + (2 + buf) * TILE_ELEMS + wid * 32 * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = (lane & 7) ^ (((wid * 32 + j * 8 + (lane >> 3)) >> 1) & 7);
      const bool kok = (kt * BK + c * 8) < p.K;
      const uint32_t ao = (kok && da_ok[j]) ? da_off[j] + (uint32_t)(kt * BK * 2) : 0xFFFFFFF0u;
      const uint32_t bo = (kok && db_ok[j]) ? db_off[j] + (uint32_t)(kt * BK * 2) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (void __attribute__((address_space(3)))*)(sa + j * 8 * 64), 16, ao, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (void __attribute__((address_space(3)))*)(sb + j * 8 * 64), 16, bo, 0, 0, 0);
    }
  };
  if (DMA) {
    dma_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    gload(0);
    if (!SB) {
      lstore(0);
      __syncthreads();
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    if (DMA) {
      const int buf = kt & 1;
      if (kt + 1 < nk) dma_issue(kt + 1, buf ^ 1);   // lands in the other buffer while this one is consumed
      const bf16* sa = lds + buf * TILE_ELEMS;
      const bf16* sb = lds + (2 + buf) * TILE_ELEMS;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sa + swz(wm * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + swz(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)   // operands swapped: acc holds C^T (4 consecutive columns per lane)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (this wave's part)
      __builtin_amdgcn_s_barrier();                       // ... everyone's part; and everyone is done reading `buf`
      continue;
    }
    const int buf = SB ? 0 : (kt & 1);
    if (SB) {
      if (kt > 0) __syncthreads();   // every wave is done reading the previous tile
      lstore(0);
      __syncthreads();
    }
    if (kt + 1 < nk) gload(kt + 1);
    const bf16* sa = lds + buf * TILE_ELEMS;
    const bf16* sb = lds + ((SB ? 1 : 2) + buf) * TILE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sa + swz(wm * 64 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + swz(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (!SB) {
      if (kt + 1 < nk) lstore(buf ^ 1);
      __syncthreads();
    }
  }
  if (SB) __syncthreads();

  if (DMA) {
    // ---- epilogue straight from the registers: lane (fr, fq) holds row i*16+fr and, from tiles 2u / 2u+1, the eight
    // consecutive columns u*32 + fq*8 .. +7 of its wave's 64-column strip
    auto col_of = [&](int u) { return n0 + wn * 64 + u * 32 + fq * 8; };
    auto row_blk = [&](int b) { return b * 16; };
    auto val = [&](int u, int b, int e) { return acc[b][2 * u + (e >> 2)][e & 3]; };
    if constexpr (EPI == EPI_F32) epi_direct_f32<2, 4>(p, bz, m0 + wm * 64, fr, col_of, row_blk, val);
    else epi_direct<EPI, 2, 4, false>(p, bz, m0 + wm * 64, fr, col_of, row_blk, val);
    return;
  }
  // ---- epilogue: registers -> (fp32 math) -> LDS bf16 tile -> 16-B row-contiguous stores ----
  bf16* Cb = p.C + (long)bz * p.sC;
  if (EPI == EPI_F32) {
    float* Cf = p.Cf + (long)bz * p.sC;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = m0 + wm * 64 + i * 16 + fq * 4 + r, col = n0 + wn * 64 + j * 16 + fr;
          if (row < p.M && col < p.N) Cf[(long)row * p.ldc + col] = acc[i][j][r] * p.alpha;
        }
    return;
  }
  // staging tile [128][128] bf16 with a 16-B pad per row (row pitch 272 B) to spread banks
  constexpr int CP = 136;
  bf16* st = lds;                          // 128*136*2 = 34816 B
  bf16* st2 = (SB || DMA) ? lds : lds + 128 * CP;   // pre-activation tile (SB/DMA: same buffer, emitted in a first pass)
  constexpr bool TWO_PASS = (SB || DMA) && epi_is_save(EPI);
#pragma unroll
  for (int pass = 0; pass < (TWO_PASS ? 2 : 1); ++pass) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int col = n0 + wn * 64 + j * 16 + fr;
      float bv = 0.f;
      if (epi_has_bias(EPI))
        if (p.bias != nullptr && col < p.N) bv = bf2f(p.bias[col]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int lr = wm * 64 + i * 16 + fq * 4 + r, lc = wn * 64 + j * 16 + fr;
          float v = acc[i][j][r] + bv;
          if (epi_is_gelu(EPI)) {
            bf16 pre = f2bf(v);
            if (epi_is_save(EPI)) v = bf2f(pre);  // the activation is taken of the value that is actually saved
            float gv, dv;
            gelu_pair(v, gv, dv);
            if (EPI == EPI_BIAS_GELU_SAVEG) pre = f2bf(dv);
            if (epi_is_save(EPI) && (!TWO_PASS || pass == 0)) st2[lr * CP + lc] = pre;
            v = gv;
          }
          if (!TWO_PASS || pass == 1) st[lr * CP + lc] = f2bf(v);
        }
    }
    __syncthreads();
    // 128 rows x 16 chunks of 16 B; thread t handles chunks t + 256*j, j=0..7
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int c = tid + 256 * j, lr = c >> 4, cc = c & 15;
      int row = m0 + lr, col = n0 + cc * 8;
      long o = (long)row * p.ldc + col;
      if (row < p.M && col < p.N && o + 8 <= p.c_elems) {
        if (epi_is_dact(EPI)) {
          bf16x8 g = *(const bf16x8*)(st + lr * CP + cc * 8);
          bf16x8 a = *(const bf16x8*)(p.aux + (long)bz * p.sC + o);
          bf16x8 outv;
#pragma unroll
          for (int e = 0; e < 8; ++e) outv[e] = f2bf(bf2f(g[e]) * (EPI == EPI_MUL ? bf2f(a[e]) : gelu_grad(bf2f(a[e]))));
          *(bf16x8*)(Cb + o) = outv;
        } else if (EPI == EPI_ADD) {
          bf16x8 g = *(const bf16x8*)(st + lr * CP + cc * 8);
          bf16x8 a = *(const bf16x8*)(p.aux + (long)bz * p.sC + o);
          bf16x8 outv;
#pragma unroll
          for (int e = 0; e < 8; ++e) outv[e] = f2bf(bf2f(g[e]) + bf2f(a[e]));
          *(bf16x8*)(Cb + o) = outv;
        } else if (TWO_PASS) {
          if (pass == 0) *(u32x4*)(p.C2 + (long)bz * p.sC + o) = *(const u32x4*)(st2 + lr * CP + cc * 8);
          else *(u32x4*)(Cb + o) = *(const u32x4*)(st + lr * CP + cc * 8);
        } else {
          *(u32x4*)(Cb + o) = *(const u32x4*)(st + lr * CP + cc * 8);
          if (epi_is_save(EPI)) *(u32x4*)(p.C2 + (long)bz * p.sC + o) = *(const u32x4*)(st2 + lr * CP + cc * 8);
        }
      }
    }
    if (TWO_PASS && pass == 0) __syncthreads();
  }
}

template <int N> __device__ __forceinline__ void wait_vm() {   // s_waitcnt vmcnt(N) needs an immediate
  if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else { static_assert(N == 0, "add the immediate to wait_vm"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}
// ---------------------------------------------------------------------------------------------
// NT with dedicated loader waves: 256 x 128 x 64 tiles, 12 waves per workgroup = 8 consumers (64x64 each, the
// fragment code of the kernel above) + 4 loaders that do nothing but issue LDS-DMA, one workgroup per CU,
// three LDS stages.  Measured on the 4-wave kernel and on an 8-wave variant without loaders: the MFMAs of a K
// tile take 0.25 / 0.5 us, its LDS-DMA alone 0.46 / 0.7 us (~70 GB/s per CU, the issue rate of
// buffer_load...lds: the issuing wave is stuck ~100+ cycles per 1-KiB piece), and the two hardly overlap while
// the SAME waves do both.  A loader wave can sit in the issue queue all the time; the consumers never do.
// ---------------------------------------------------------------------------------------------
// WN = 4 (round 2): 160 x 256 tiles for the N = 3072 outputs (fc1 forward, fc2 dgrad): 41 x 12 = 492 workgroups = 1.92 rounds
// of the 256 CUs where 128 x 128 tiles need 2.44 -> 3, at 98 instead of 64 FLOP per staged byte; three stages of 52 KiB use
// the whole 160 KiB of LDS, so the bf16 epilogue goes out in two 128-column halves.
template <int EPI, int WM, int MI, int WN = 2>   // consumers: WM x WN waves of (16 MI) x 64; tile (WM * MI * 16) x (WN * 64)
__global__ __launch_bounds__((WN * WM + 4) * 64) void gemm_nt_lc_kernel(GemmP p) {
  constexpr int NLOAD = 4, NC = WN * WM;
  constexpr int TBM = WM * MI * 16, TBN = WN * 64, NST = 3;
  constexpr int A_EL = TBM * BK, B_EL = TBN * BK, STAGE_EL = A_EL + B_EL;   // 48 KiB per stage
  constexpr int NTHR = (NC + NLOAD) * 64;
  constexpr int APIECES = TBM / 8, BPIECES = TBN / 8, LP = (APIECES + BPIECES) / NLOAD;   // pieces per loader wave and stage
  static_assert((APIECES + BPIECES) % NLOAD == 0, "pieces must divide over the loaders");
  static_assert(NST * STAGE_EL * 2 <= 160 * 1024, "three stages must fit the 160 KiB of LDS");
  __shared__ __attribute__((aligned(16))) bf16 lds[NST * STAGE_EL];         // 144 KiB (156 KiB at WN = 4); the epilogue reuses it
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool loader = wid >= NC;
  const int wm = (wid / WN) % WM, wn = wid % WN;
  int tm_, tn_;
  tile_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.y, gridDim.x, 4, tm_, tn_);
  const int m0 = tm_ * TBM, n0 = tn_ * TBN;
  const int bz = blockIdx.z;
  const int nk = (p.K + BK - 1) / BK;
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;

  if (loader) {
    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (long)bz * p.sA, p.a_bytes);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + (long)bz * p.sB, p.b_bytes);
    // the stage image is 48 pieces of 8 rows x 128 B (A rows 0..255, then B rows 0..127); loader l takes pieces
    // l, l+NLOAD, ...  Linear LDS image, XOR swizzle on the SOURCE chunk as in the 4-wave kernel.
    const int lw = (wid - NC) & 3;   // the mask tells the compiler the range: which pieces are A / B becomes static
    uint32_t d_off[LP];
    int d_c8[LP];
    bool d_ok[LP];
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      const int piece = lw + NLOAD * j;              // 0..47
      const bool isA = piece < APIECES;
      const int row = (isA ? piece : piece - APIECES) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      d_c8[j] = c * 8;
      if (isA) {
        d_ok[j] = m0 + row < p.M;
        d_off[j] = (uint32_t)((p.a_off + (long)(m0 + row) * p.lda + c * 8) * 2);
      } else {
        d_ok[j] = n0 + row < p.N;
        d_off[j] = (uint32_t)(((long)(n0 + row) * p.ldb + c * 8) * 2);
      }
    }
    auto dma_issue = [&](int kt, int stage) {   // kt >= nk: out-of-range fetches (zeros), same instruction count
      bf16* sbase = lds + stage * STAGE_EL;
#pragma unroll
      for (int j = 0; j < LP; ++j) {
        const int piece = lw + NLOAD * j;
        const bool kok = kt < nk && (kt * BK + d_c8[j]) < p.K;
        const uint32_t o = (kok && d_ok[j]) ? d_off[j] + (uint32_t)(kt * BK * 2) : 0xFFFFFFF0u;
        bf16* dst = sbase + piece * 8 * 64;          // the A pieces fill [0, A_EL), the B pieces follow
        if (piece < APIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
      }
    };
    dma_issue(0, 0);
    dma_issue(1, 1);
    wait_vm<LP>();   // tile 0 landed
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int pre = stage == 0 ? 2 : stage - 1;        // (stage + 2) % 3: free since the last barrier
      dma_issue(kt + 2, pre);
      wait_vm<LP>();  // tile kt+1 landed; kt+2 in flight
      __builtin_amdgcn_s_barrier();
      stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing zero-fill prefetches still target the LDS
  } else {
    __builtin_amdgcn_s_barrier();                         // tile 0 is in
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bf16* sa = lds + stage * STAGE_EL;
      const bf16* sb = sa + A_EL;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[MI], bfr[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + swz(wm * (MI * 16) + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + swz(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_barrier();                       // done reading `stage`; the next tile is in
      stage = stage == 2 ? 0 : stage + 1;
    }
  }
  __builtin_amdgcn_s_barrier();                           // every DMA has landed, every read is done

  // ---- epilogue (consumer waves): registers -> (fp32 math) -> LDS bf16 tile(s) [256][136] -> 16-B row stores ----
  bf16* Cb = p.C + (long)bz * p.sC;
  if (EPI == EPI_F32) {
    if (loader) return;
    float* Cf = p.Cf + (long)bz * p.sC;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = m0 + wm * (MI * 16) + i * 16 + fq * 4 + r, col = n0 + wn * 64 + j * 16 + fr;
          if (row < p.M && col < p.N) Cf[(long)row * p.ldc + col] = acc[i][j][r] * p.alpha;
        }
    return;
  }
  constexpr int CP = 136;
  bf16* st = lds;                 // TBM * 136 * 2 bytes (69632 for 256 rows)
  bf16* st2 = lds + TBM * CP;     // pre-activation tile
#pragma unroll 1
  for (int half = 0; half < WN / 2; ++half) {      // 128 output columns per pass through the staging tile(s)
    if (half > 0) __syncthreads();
    if (!loader && (wn >> 1) == half) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int col = n0 + wn * 64 + j * 16 + fr;
        float bv = 0.f;
        if (epi_has_bias(EPI))
          if (p.bias != nullptr && col < p.N) bv = bf2f(p.bias[col]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int lr = wm * (MI * 16) + i * 16 + fq * 4 + r, lc = (wn & 1) * 64 + j * 16 + fr;
            float v = acc[i][j][r] + bv;
            if (epi_is_gelu(EPI)) {
              bf16 pre = f2bf(v);
              if (epi_is_save(EPI)) v = bf2f(pre);  // the activation is taken of the value that is actually saved
              float gv, dv;
              gelu_pair(v, gv, dv);
              if (EPI == EPI_BIAS_GELU_SAVEG) pre = f2bf(dv);
              if (epi_is_save(EPI)) st2[lr * CP + lc] = pre;
              v = gv;
            }
            st[lr * CP + lc] = f2bf(v);
          }
      }
    }
    __syncthreads();
    // TBM rows x 16 chunks of 16 B over all threads
    for (int c = tid; c < TBM * 16; c += NTHR) {
      int lr = c >> 4, cc = c & 15;
      int row = m0 + lr, col = n0 + half * 128 + cc * 8;
      long o = (long)row * p.ldc + col;
      if (row < p.M && col < p.N && o + 8 <= p.c_elems) {
        if (epi_is_dact(EPI)) {
          bf16x8 g = *(const bf16x8*)(st + lr * CP + cc * 8);
          bf16x8 a = *(const bf16x8*)(p.aux + (long)bz * p.sC + o);
          bf16x8 outv;
#pragma unroll
          for (int e = 0; e < 8; ++e) outv[e] = f2bf(bf2f(g[e]) * (EPI == EPI_MUL ? bf2f(a[e]) : gelu_grad(bf2f(a[e]))));
          *(bf16x8*)(Cb + o) = outv;
        } else if (EPI == EPI_ADD) {
          bf16x8 g = *(const bf16x8*)(st + lr * CP + cc * 8);
          bf16x8 a = *(const bf16x8*)(p.aux + (long)bz * p.sC + o);
          bf16x8 outv;
#pragma unroll
          for (int e = 0; e < 8; ++e) outv[e] = f2bf(bf2f(g[e]) + bf2f(a[e]));
          *(bf16x8*)(Cb + o) = outv;
        } else {
          *(u32x4*)(Cb + o) = *(const u32x4*)(st + lr * CP + cc * 8);
          if (epi_is_save(EPI)) *(u32x4*)(p.C2 + (long)bz * p.sC + o) = *(const u32x4*)(st2 + lr * CP + cc * 8);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NT, PERSISTENT loader/consumer: the kernel above as a loop over output tiles.  One workgroup per CU walks
// tiles g, g+G, g+2G, ...; the LDS-DMA ring simply keeps running across tile boundaries, so while the consumers
// write a finished tile out the loaders already fetch the first K tiles of the next one, and the per-tile
// fixed cost (launch, two K tiles of cold-start latency, an LDS-staged epilogue: 7.5 us above) shrinks to the
// store time of the epilogue.  The epilogue needs no LDS (the loaders own it all the time): the MFMA operands are
// swapped, acc[i][j] = B_frag x A_frag, so a lane holds FOUR CONSECUTIVE output columns of one row and stores
// them as 8 bytes; the four j tiles of a wave complete a row's 128-byte line.
// ---------------------------------------------------------------------------------------------
struct TileIter {   // tiles of one workgroup, in XCD-aware order: round r covers ids [r*G, (r+1)*G), the 32 workgroups
  int G, slot, total, ntm, ntn;   // of an XCD take a contiguous chunk of it, ids walk group-M inside a batch plane
  // zk skip: cheap and full column tiles must alternate per workgroup, but 256 % (4 ntn) == 0 hands a workgroup the same
  // column tile every round.  rot > 0: the column tile is rotated by (row group / rot + batch plane) - a function of the
  // TILE alone (a bijection of the tile set: rotating by the round is not, it duplicates tiles where a row group straddles two
  // rounds), and with rot = G / (4 ntn) it advances by one per round.
  int rot = 0;
  __device__ __forceinline__ bool get(int r, int& bz, int& tm, int& tn) const {
    const long id = (long)r * G + slot;
    if (id >= total) return false;
    const int plane = ntm * ntn;
    bz = (int)(id / plane);
    const int t = (int)(id - (long)bz * plane), GM = 4;
    const int per_group = GM * ntn, group = t / per_group, first_m = group * GM;
    const int gsz = min(ntm - first_m, GM), in_group = t - group * per_group;
    // workgroup-uniform by construction; say so, or every LDS-DMA gets a waterfall loop around its descriptor
    bz = __builtin_amdgcn_readfirstlane(bz);
    tm = __builtin_amdgcn_readfirstlane(first_m + in_group % gsz);
    tn = in_group / gsz;
    if (rot) tn = (tn + group / rot + bz) % ntn;
    tn = __builtin_amdgcn_readfirstlane(tn);
    return true;
  }
};

struct TileIter32 {   // the same order in 32-bit arithmetic (total tiles < 2^31): the 8-phase kernel calls it from its staging cursor
  int G, slot, total, ntm, ntn;
  int rot = 0;
  __device__ __forceinline__ bool get(int r, int& bz, int& tm, int& tn) const {
    const unsigned id = (unsigned)r * (unsigned)G + (unsigned)slot;
    if (id >= (unsigned)total) return false;
    const unsigned plane = (unsigned)(ntm * ntn), b = id / plane, t = id - b * plane, GM = 4;
    const unsigned per_group = GM * (unsigned)ntn, group = t / per_group, first_m = group * GM;
    const unsigned gsz = min((unsigned)ntm - first_m, GM), in_group = t - group * per_group, q = in_group / gsz;
    bz = __builtin_amdgcn_readfirstlane((int)b);
    tm = __builtin_amdgcn_readfirstlane((int)(first_m + in_group - q * gsz));
    unsigned tq = q;
    if (rot) tq = (q + group / (unsigned)rot + b) % (unsigned)ntn;      // see TileIter::rot
    tn = __builtin_amdgcn_readfirstlane((int)tq);
    return true;
  }
};

template <int EPI, int WM, int MI, int WN = 2>     // WN = 4: (WM * MI * 16) x 256 tiles (round 2, the N = 3072 / 2304 encoder outputs)
__global__ __launch_bounds__((WN * WM + 4) * 64) void gemm_nt_p_kernel(GemmP p, int ntm, int ntn, int total_tiles) {
  constexpr int NLOAD = 4, NC = WN * WM;
  constexpr int TBM = WM * MI * 16, TBN = WN * 64, NST = 3;
  constexpr int A_EL = TBM * BK, B_EL = TBN * BK, STAGE_EL = A_EL + B_EL;
  constexpr int APIECES = TBM / 8, BPIECES = TBN / 8, LP = (APIECES + BPIECES) / NLOAD;
  static_assert((APIECES + BPIECES) % NLOAD == 0, "pieces must divide over the loaders");
  static_assert(NST * STAGE_EL * 2 <= 160 * 1024, "three stages must fit the 160 KiB of LDS");
  __shared__ __attribute__((aligned(16))) bf16 lds[NST * STAGE_EL];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool loader = wid >= NC;
  const int wm = (wid / WN) % WM, wn = wid % WN;
  const int nk = (p.K + BK - 1) / BK;
  TileIter it;
  it.G = gridDim.x; it.total = total_tiles; it.ntm = ntm; it.ntn = ntn; it.rot = p.zk_kt > 0 ? max(1, (int)gridDim.x / (4 * ntn)) : 0;
  {
    const int g = blockIdx.x, G = gridDim.x, qd = G >> 3, rm = G & 7, xcd = g & 7;
    it.slot = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (g >> 3);
  }
  if (loader) {
    const int lw = (wid - NC) & 3;   // the mask tells the compiler the range: which pieces are A / B becomes static
    // Issues run two K tiles ahead of the consumers: issue I0, then for every further issue {issue; wait until the
    // previous one has landed; barrier}.  After the last real K tile two out-of-range issues (zeros) keep the
    // instruction count per step constant for the counted vmcnt.  Plain nested loops over (tile, K tile): the
    // compiler must see that descriptors and offsets are wave-uniform, or it wraps every DMA in a waterfall loop.
    int my_tiles = 0;
    { int b_, m_, n_; while (it.get(my_tiles, b_, m_, n_)) ++my_tiles; }
    int stage = 0;
    bool first = true;
    for (int r = 0; r <= my_tiles; ++r) {            // r == my_tiles: the two trailing dummy issues
      int bz = 0, tm = 0, tn = 0;
      const bool live = r < my_tiles && it.get(r, bz, tm, tn);
      __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (long)bz * p.sA, p.a_bytes);
      __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + (long)bz * p.sB, p.b_bytes);
      const int m0 = tm * TBM, n0 = tn * TBN;
      uint32_t d_off[LP];
      int d_c8[LP];
      bool d_ok[LP];
#pragma unroll
      for (int j = 0; j < LP; ++j) {
        const int piece = lw + NLOAD * j;
        const bool isA = piece < APIECES;
        const int row = (isA ? piece : piece - APIECES) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        d_c8[j] = c * 8;
        if (isA) {
          d_ok[j] = live && m0 + row < p.M;
          d_off[j] = (uint32_t)((p.a_off + (long)(m0 + row) * p.lda + c * 8) * 2);
        } else {
          // B rows are stored PERMUTED: LDS row (w*64 + jj*16 + f) <- B row w*64 + (jj>>1)*32 + (f>>2)*8 + (jj&1)*4 + (f&3),
          // so that a consumer lane ends up with 8 consecutive output columns in the accumulators of tiles 2u, 2u+1
          const int f = row & 15, jj = (row >> 4) & 3;
          const int srow = (row & ~63) + (jj >> 1) * 32 + (f >> 2) * 8 + (jj & 1) * 4 + (f & 3);
          d_ok[j] = live && n0 + srow < p.N;
          d_off[j] = (uint32_t)(((long)(n0 + srow) * p.ldb + c * 8) * 2);
        }
      }
      const int kts = live ? nk : 2;
      const int kb = (live && p.zk_kt > 0 && n0 >= p.zk_col) ? p.zk_kt : 0;     // B is zero before this K tile: skipped
      for (int kt = kb; kt < kts; ++kt) {
        bf16* sbase = lds + stage * STAGE_EL;
#pragma unroll
        for (int j = 0; j < LP; ++j) {
          const int piece = lw + NLOAD * j;
          const bool kok = (kt * BK + d_c8[j]) < p.K;
          const uint32_t o = (kok && d_ok[j]) ? d_off[j] + (uint32_t)(kt * BK * 2) : 0xFFFFFFF0u;
          bf16* dst = sbase + piece * 8 * 64;
          if (piece < APIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
        }
        if (!first) {
          wait_vm<LP>();                              // the previous issue has landed, this one stays in flight
          __builtin_amdgcn_s_barrier();
        }
        first = false;
        stage = stage == 2 ? 0 : stage + 1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // ---- consumers
  const int fr = lane & 15, fq = lane >> 4;
  __builtin_amdgcn_s_barrier();
  int stage = 0;
  for (int r = 0;; ++r) {
    int bz, tm, tn;
    if (!it.get(r, bz, tm, tn)) break;
    const int m0 = tm * TBM, n0 = tn * TBN;
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = (p.zk_kt > 0 && n0 >= p.zk_col) ? p.zk_kt : 0; kt < nk; ++kt) {
      const bf16* sa = lds + stage * STAGE_EL;
      const bf16* sb = sa + A_EL;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[MI], bfr[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + swz(wm * (MI * 16) + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j)   // LDS row j*16 + fr holds output column (j>>1)*32 + (fr>>2)*8 + (j&1)*4 + (fr&3): the loaders permute
          bfr[j] = *(const bf16x8*)(sb + swz(wn * 64 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)   // operands swapped: the accumulator holds C^T, i.e. 4 consecutive columns per lane
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_barrier();
      stage = stage == 2 ? 0 : stage + 1;
    }
    // ---- epilogue straight from the registers: lane (fr, fq) holds row i*16+fr and, from tiles 2u / 2u+1, the
    // EIGHT consecutive columns u*32 + fq*8 .. +7 (one 16-byte store; dwordx2 stores are issue-bound)
    auto col_of = [&](int u) { return n0 + wn * 64 + u * 32 + fq * 8; };
    auto row_blk = [&](int b) { return b * 16; };
    auto val = [&](int u, int b, int e) { return acc[b][2 * u + (e >> 2)][e & 3]; };
    if constexpr (EPI == EPI_F32) epi_direct_f32<2, MI>(p, bz, m0 + wm * (MI * 16), fr, col_of, row_blk, val);
    else epi_direct<EPI, 2, MI, false>(p, bz, m0 + wm * (MI * 16), fr, col_of, row_blk, val);
  }
}

// ---------------------------------------------------------------------------------------------
// NT, EIGHT-PHASE (round 3): (64 MH) x 256 x 64 tiles (256 rows at MH = 4, 320 at MH = 5, 192 at MH = 3), 8 waves =
// 2 (M) x 4 (N), one persistent workgroup per CU.  Every wave both stages (LDS-DMA) and computes; a wave owns
// (32 MH) x 64 outputs = 2 x 2 quadrants of (16 MH) x 32, one quadrant per PHASE, four phases per K tile:
//     { fragment reads of the quadrant | LDS-DMA of ONE half-tile two K tiles ahead | s_barrier | 4 MH MFMAs | s_barrier }
// The two wave groups (wr = 0 / 1, SIMD partners) run staggered by one barrier, so on every SIMD one wave is in its MFMA
// section while its partner reads / stages.  A K tile lives in four half-tile images:  A_h = the m-half h rows of both wave
// groups, B_h = the n-half h columns (32) of all four wave columns - a half-tile is dead as soon as ITS quadrants are done,
// which is what lets three half-tiles stay in flight behind a counted vmcnt with only two LDS buffers:
//     phase 1: read B_h0 (first) + A_h0, stage A_h1 of K tile s+1 -> other buffer | lgkmcnt retires the B reads | q(0,0)
//     phase 2: read B_h1,                stage B_h0 of K tile s+2 -> this buffer  | q(0,1)
//     phase 3: read A_h1,                stage A_h0 of s+2                        | q(1,1)
//     phase 4:                           stage B_h1 of s+2, s_waitcnt vmcnt(the 3 youngest half-tiles) | q(1,0)
// Hazards (guide, "8-phase template"): RAW - the wait of phase 4 (step s+1) retires every piece of K tile s+2, whose first
// read is phase 1 of step s+2, one phase and (for both groups) at least one barrier later.  WAR - a half-tile is restaged two
// phases after its last read, or one phase after when those reads were retired before the reading phase's first barrier
// (B_h0).  The staging cursor runs over the workgroup's whole (tile, K tile) stream, so the first K tiles of the next output
// tile are already in flight during an epilogue.  The epilogue is register-direct (operands swapped + permuted B rows, as
// in the persistent kernel above).  Needs K % 64 == 0 (no k tail: garbage beyond M / N only reaches masked outputs).
// ---------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_lgkm() {
  if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int N> __device__ __forceinline__ void wait_vm8() {
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, int MH>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(GemmP p, int ntm, int ntn, int total_tiles) {
  constexpr int HROWS = 16 * MH, WROWS = 2 * HROWS, TBM = 2 * WROWS;
  constexpr int HA_ROWS = 2 * HROWS;                    // an A half-tile: m-half h of both wave groups
  constexpr int HA_EL = HA_ROWS * 64, HB_EL = 128 * 64;
  constexpr int BUF_EL = 2 * HA_EL + 2 * HB_EL;
  constexpr int APC = HA_ROWS / 8;                       // 1-KiB pieces (8 rows x 128 B) of an A half-tile
  constexpr int NPA = (APC + 7) / 8;                     // LDS-DMA instructions per wave and A half-tile
  constexpr bool PADA = (APC % 8) != 0;                  // some waves issue one dummy piece (zeros into a scratch KiB)
  constexpr int VMW = NPA + 4;                           // DMAs of the three youngest half-tiles: B_h0, A_h0, B_h1
  static_assert(VMW == 6 || VMW == 7, "add the immediate to wait_vm8");
  static_assert((2 * BUF_EL + (PADA ? 8 * 512 : 0)) * 2 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) bf16 lds[2 * BUF_EL + (PADA ? 8 * 512 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3, fr = lane & 15, fq = lane >> 4, l3 = lane >> 3, l7 = lane & 7;
  const int nk = p.K / BK, nk2 = (nk + 1) & ~1;          // K tiles per output tile, padded to a pair (pad = zeros)
  TileIter32 it;
  it.G = gridDim.x; it.total = total_tiles; it.ntm = ntm; it.ntn = ntn; it.rot = p.zk_kt > 0 ? max(1, (int)gridDim.x / (4 * ntn)) : 0;
  {
    const int g = blockIdx.x, G = gridDim.x, qd = G >> 3, rm = G & 7, xcd = g & 7;
    it.slot = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (g >> 3);
  }
  // ---- staging geometry.  Piece pc = j * 8 + wid of a half-tile = LDS rows pc*8 .. pc*8+7 (lane>>3 = row, lane&7 = physical
  // chunk); the XOR swizzle of the fragment reads, chunk ^ ((row >> 1) & 7), goes on the SOURCE chunk.
  const int csrc = l7 ^ (((wid & 1) * 4 + (l3 >> 1)) & 7);
  const uint32_t a_thr = (uint32_t)(((long)l3 * p.lda + csrc * 8) * 2);
  const uint32_t b_thr = (uint32_t)(((long)((l3 >> 2) * 8 + (l3 & 3)) * p.ldb + csrc * 8) * 2);
  uint32_t arow[NPA][2], brow[2][2];
  bool a_dummy[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int pc = j * 8 + wid;
    a_dummy[j] = PADA && pc >= APC;
    const int g = pc / (2 * MH), rr0 = (pc % (2 * MH)) * 8;     // 2 MH pieces per wave group
#pragma unroll
    for (int h = 0; h < 2; ++h) arow[j][h] = (uint32_t)((long)(g * WROWS + h * HROWS + rr0) * p.lda * 2);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    // LDS row (wc*32 + j1*16 + f) of B_h  <-  B row wc*64 + h*32 + (f>>2)*8 + j1*4 + (f&3): with the MFMA operands swapped a
    // lane then holds 8 consecutive output columns in the accumulators of the j-tiles (2h, 2h+1)
    const int pc = j * 8 + wid;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      brow[j][h] = (uint32_t)((long)((pc >> 2) * 64 + h * 32 + (pc & 1) * 16 + ((pc >> 1) & 1) * 4) * p.ldb * 2);
  }
  // ---- fragment read offsets (elements): row = 16-aligned base + fr, so (row >> 1) & 7 = fr >> 1
  const int a_rd0 = (wr * HROWS + fr) * 64 + ((fq ^ (fr >> 1)) << 3), a_rd1 = (wr * HROWS + fr) * 64 + (((4 + fq) ^ (fr >> 1)) << 3);
  const int b_rd0 = (wc * 32 + fr) * 64 + ((fq ^ (fr >> 1)) << 3), b_rd1 = (wc * 32 + fr) * 64 + (((4 + fq) ^ (fr >> 1)) << 3);

  // ---- staging cursor over the (tile, K tile) stream of this workgroup
  int r_s = 0, kt_s = 0;
  bool live_s = false, ok_cur = false, ok_prev = false;
  uint32_t a_cur = 0, b_cur = 0, a_prev = 0;
  __amdgpu_buffer_rsrc_t ra_cur = make_rsrc(p.A, p.a_bytes), rb_cur = make_rsrc(p.B, p.b_bytes), ra_prev = ra_cur;
  auto set_tile = [&](int r) {
    int bz, tm, tn;
    live_s = it.get(r, bz, tm, tn);
    ok_cur = live_s;
    kt_s = 0;
    if (live_s) {
      // zk skip (host: zk_kt even): a tile in the zero-block columns starts its K loop - cursor and compute loop alike - at zk_kt
      kt_s = (p.zk_kt > 0 && tn * 256 >= p.zk_col) ? p.zk_kt : 0;
      a_cur = a_thr + (uint32_t)((p.a_off + (long)tm * TBM * p.lda + (long)kt_s * BK) * 2);
      b_cur = b_thr + (uint32_t)(((long)tn * 256 * p.ldb + (long)kt_s * BK) * 2);
      ra_cur = make_rsrc(p.A + (long)bz * p.sA, p.a_bytes);
      rb_cur = make_rsrc(p.B + (long)bz * p.sB, p.b_bytes);
    }
  };
  auto advance = [&]() {
    a_prev = a_cur; ok_prev = ok_cur; ra_prev = ra_cur;
    ++kt_s;
    if (kt_s < nk2) { a_cur += 2 * BK; b_cur += 2 * BK; ok_cur = live_s && kt_s < nk; }
    else { ++r_s; set_tile(r_s); }
  };
  auto stage_A = [&](int buf, int h, __amdgpu_buffer_rsrc_t rs, uint32_t base, bool ok) {
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const uint32_t o = (ok && !a_dummy[j]) ? base + arow[j][h] : 0xFFFFFFF0u;
      bf16* dst = a_dummy[j] ? lds + 2 * BUF_EL + wid * 512 : lds + buf * BUF_EL + h * HA_EL + (j * 8 + wid) * 512;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
    }
  };
  auto stage_B = [&](int buf, int h, __amdgpu_buffer_rsrc_t rs, uint32_t base, bool ok) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t o = ok ? base + brow[j][h] : 0xFFFFFFF0u;
      bf16* dst = lds + buf * BUF_EL + 2 * HA_EL + h * HB_EL + (j * 8 + wid) * 512;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
    }
  };
  W2VS_STAMP(p, 0);
  set_tile(0);
  if (!live_s) return;                                   // workgroup-uniform: more workgroups than tiles
  // ---- prologue: K tile 0 whole, K tile 1 without its A_h1 (phase 1 of step 0 brings it)
  stage_B(0, 0, rb_cur, b_cur, ok_cur); stage_A(0, 0, ra_cur, a_cur, ok_cur);
  stage_B(0, 1, rb_cur, b_cur, ok_cur); stage_A(0, 1, ra_cur, a_cur, ok_cur);
  advance();
  stage_B(1, 0, rb_cur, b_cur, ok_cur); stage_A(1, 0, ra_cur, a_cur, ok_cur); stage_B(1, 1, rb_cur, b_cur, ok_cur);
  advance();
  wait_vm8<VMW>();
  __builtin_amdgcn_s_barrier();
  W2VS_STAMP(p, 1);
  if (wr == 1) __builtin_amdgcn_s_barrier();             // the stagger: group 1 runs one barrier behind group 0

  f32x4 acc[2][MH][4];
  bf16x8 af[MH][2], b0[2][2], b1[2][2];
  auto mfma_q = [&](int mh, int nh, bf16x8 (&bb)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j1 = 0; j1 < 2; ++j1)   // operands swapped: the accumulator holds C^T, 4 consecutive columns per lane
          acc[mh][i][nh * 2 + j1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j1][ks], af[i][ks], acc[mh][i][nh * 2 + j1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto ktile = [&](auto ebuf) {
    constexpr int EB = decltype(ebuf)::value;
    const bf16* sA0 = lds + EB * BUF_EL;
    const bf16* sA1 = sA0 + HA_EL;
    const bf16* sB0 = sA0 + 2 * HA_EL;
    const bf16* sB1 = sB0 + HB_EL;
    // ---- phase 1
#pragma unroll
    for (int j1 = 0; j1 < 2; ++j1) {
      b0[j1][0] = *(const bf16x8*)(sB0 + b_rd0 + j1 * 1024);
      b0[j1][1] = *(const bf16x8*)(sB0 + b_rd1 + j1 * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MH; ++i) {
      af[i][0] = *(const bf16x8*)(sA0 + a_rd0 + i * 1024);
      af[i][1] = *(const bf16x8*)(sA0 + a_rd1 + i * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_A(EB ^ 1, 1, ra_prev, a_prev, ok_prev);
    wait_lgkm<2 * MH>();                                 // the four B_h0 reads (issued first) are done: phase 2 may restage it
    __builtin_amdgcn_s_barrier();
    mfma_q(0, 0, b0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2
#pragma unroll
    for (int j1 = 0; j1 < 2; ++j1) {
      b1[j1][0] = *(const bf16x8*)(sB1 + b_rd0 + j1 * 1024);
      b1[j1][1] = *(const bf16x8*)(sB1 + b_rd1 + j1 * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_B(EB, 0, rb_cur, b_cur, ok_cur);
    __builtin_amdgcn_s_barrier();
    mfma_q(0, 1, b1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3
#pragma unroll
    for (int i = 0; i < MH; ++i) {
      af[i][0] = *(const bf16x8*)(sA1 + a_rd0 + i * 1024);
      af[i][1] = *(const bf16x8*)(sA1 + a_rd1 + i * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_A(EB, 0, ra_cur, a_cur, ok_cur);
    __builtin_amdgcn_s_barrier();
    mfma_q(1, 1, b1);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4
    stage_B(EB, 1, rb_cur, b_cur, ok_cur);
    wait_vm8<VMW>();                                     // everything but the three youngest half-tiles has landed
    __builtin_amdgcn_s_barrier();
    mfma_q(1, 0, b0);
    __builtin_amdgcn_s_barrier();
    advance();
  };

  for (int r = 0;; ++r) {
    int bz, tm, tn;
    if (!it.get(r, bz, tm, tn)) break;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mh][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kp = (p.zk_kt > 0 && tn * 256 >= p.zk_col) ? p.zk_kt : 0; kp < nk2; kp += 2) {
      ktile(std::integral_constant<int, 0>{});
      ktile(std::integral_constant<int, 1>{});
    }
    // ---- epilogue straight from the registers: lane (fr, fq) holds row .. + i*16 + fr and, from the j-tiles 2u / 2u+1,
    // the EIGHT consecutive columns wc*64 + u*32 + fq*8 .. +7.  The stagger is suspended around it (group 0 waits one barrier
    // for group 1's last MFMA phase, group 1 re-opens the gap afterwards): staggered, the two groups' epilogues run one after
    // the other, each with one wave per SIMD - half the VALU / store issue rate of the CU.
    if (r < 2) W2VS_STAMP(p, 2 + 2 * r);
    if (wr == 0) __builtin_amdgcn_s_barrier();
    const int m0 = tm * TBM, n0 = tn * 256;
    auto col_of = [&](int u) { return n0 + wc * 64 + u * 32 + fq * 8; };
    auto row_blk = [&](int b) { return (b / MH) * HROWS + (b % MH) * 16; };
    auto val = [&](int u, int b, int e) { return acc[b / MH][b % MH][2 * u + (e >> 2)][e & 3]; };
    if constexpr (EPI == EPI_F32) epi_direct_f32<2, 2 * MH>(p, bz, m0 + wr * WROWS, fr, col_of, row_blk, val);
    else epi_direct<EPI, 2, 2 * MH, true>(p, bz, m0 + wr * WROWS, fr, col_of, row_blk, val);
    if (r < 2) W2VS_STAMP(p, 3 + 2 * r);
    if (wr == 1) __builtin_amdgcn_s_barrier();           // group 1 falls one barrier behind again
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();             // pairs with group 1's extra barrier in front of the tile loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing (out-of-range) prefetches still target the LDS
  W2VS_STAMP(p, 6);
}

// ---------------------------------------------------------------------------------------------
// TN: C[M,N] += alpha * sum_k A[k,M] * B[k,N];  A,B row-major with the contraction index as the
// ROW index (tokens).  grid.z splits K; partial sums are combined with fp32 atomics into Cf
// (which the caller zeroes), so a step's dW for every layer lands in one fp32 gradient arena.
// LDS tiles keep the global [k][m] orientation; MFMA fragments come out of LDS through
// ds_read_b64_tr_b16 (hardware transpose), two reads per 8-deep k group.
// ---------------------------------------------------------------------------------------------
constexpr int TK = 64;  // k rows per tile
// tile [64 k][128 m] bf16, unpadded 256-B rows.  A 16-lane tr-read group touches 4 rows x 32 B and the
// two groups of a 32-lane half sit 8 rows apart: XOR-ing the 32-B-window index with
// w = (row & 3) + 4 * ((row >> 3) & 1) puts those 8 (row, window) pieces on 8 disjoint bank windows.
constexpr int TP = 128;
__device__ __forceinline__ int tswz(int row, int col) {
  return row * TP + (col ^ ((((row & 3) | (((row >> 3) & 1) << 2))) << 4));
}

__device__ __forceinline__ s16x4 ds_tr(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) bf16 lds[4 * TK * TP];  // A0 A1 B0 B1 : 4*64*128*2 = 65536 B
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  int tm_, tn_;
  tile_order(blockIdx.y * gridDim.x + blockIdx.x, gridDim.y, gridDim.x, 4, tm_, tn_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const int bz = blockIdx.z / p.n_split, sp = blockIdx.z % p.n_split;
  const int k_begin = sp * p.k_split;
  const int k_end = min(p.K, k_begin + p.k_split);
  if (k_begin >= k_end) return;
  __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (long)bz * p.sA, p.a_bytes);
  __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + (long)bz * p.sB, p.b_bytes);
  // staging: tile = 64 rows x 16 chunks (16 B); thread t owns chunks t + 256*j, j = 0..3
  int l_off[4];
  uint32_t a_col[4], b_col[4];
  int krow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = tid + 256 * j, row = c >> 4, cc = c & 15;
    krow[j] = row;
    l_off[j] = tswz(row, cc * 8);
    a_col[j] = (m0 + cc * 8 < p.M) ? (uint32_t)((m0 + cc * 8) * 2) : 0xFFFFFFF0u;
    b_col[j] = (n0 + cc * 8 < p.N) ? (uint32_t)((n0 + cc * 8) * 2) : 0xFFFFFFF0u;
  }
  u32x4 ra_reg[4], rb_reg[4];
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = p.colsum != nullptr && tn_ == 0;
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int k = k0 + krow[j];
      bool ok = k < k_end;
      uint32_t ao = (ok && a_col[j] != 0xFFFFFFF0u) ? (uint32_t)(((long)p.a_off + (long)k * p.lda) * 2) + a_col[j] : 0xFFFFFFF0u;
      uint32_t bo = (ok && b_col[j] != 0xFFFFFFF0u) ? (uint32_t)((long)k * p.ldb * 2) + b_col[j] : 0xFFFFFFF0u;
      ra_reg[j] = buf_load16(ra, ao);
      rb_reg[j] = buf_load16(rb, bo);
    }
  };
  auto lstore = [&](int buf) {
    bf16* sa = lds + buf * TK * TP;
    bf16* sb = lds + (2 + buf) * TK * TP;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(u32x4*)(sa + l_off[j]) = ra_reg[j];
      *(u32x4*)(sb + l_off[j]) = rb_reg[j];
    }
    if (do_colsum) {  // the A tile passes through these registers exactly once: add it up on the way
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        union { u32x4 u; bf16x8 h; } cv; cv.u = ra_reg[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += bf2f(cv.h[e]);
      }
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (k_end - k_begin + TK - 1) / TK;
  gload(k_begin);
  lstore(0);
  __syncthreads();
  // tr-read addressing: 16-lane group g = lane>>4 owns k rows 8g..8g+7 of a 32-deep k step;
  // inside the group lane i = 4q+pp supplies row q, columns 4pp..4pp+3 and receives column i.
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(k_begin + (kt + 1) * TK);
    const bf16* sa = lds + buf * TK * TP;
    const bf16* sb = lds + (2 + buf) * TK * TP;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
      const int kr = ks * 32 + g * 8 + q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = wm * 64 + i * 16 + pp * 4;
        s16x4 lo = ds_tr(sa + tswz(kr, col)), hi = ds_tr(sa + tswz(kr + 4, col));
        union { bf16x8 v; s16x4 h[2]; } u; u.h[0] = lo; u.h[1] = hi; af[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wn * 64 + j * 16 + pp * 4;
        s16x4 lo = ds_tr(sb + tswz(kr, col)), hi = ds_tr(sb + tswz(kr + 4, col));
        union { bf16x8 v; s16x4 h[2]; } u; u.h[0] = lo; u.h[1] = hi; bfr[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = m0 + wm * 64 + i * 16 + fq * 4 + r, col = n0 + wn * 64 + j * 16 + fr;
        if (row < p.M && col < p.N) atomicAdd(&p.Cf[(long)bz * p.sC + (long)row * p.ldc + col], acc[i][j][r] * p.alpha);
      }
  if (do_colsum) {
    // 16 threads (tid>>4) share a column chunk cc = tid&15: each parks its 8 sums in its own LDS row, 128 threads
    // then fold the 16 rows (no LDS float atomics: ds_add_f32 runs at ~130 cycles per wave-instruction)
    float* red = (float*)lds;
    __syncthreads();
    {
      float* mine = red + (tid >> 4) * BM + (tid & 15) * 8;
      *(f32x4*)(mine) = f32x4{csum[0], csum[1], csum[2], csum[3]};
      *(f32x4*)(mine + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
    }
    __syncthreads();
    if (tid < BM) {
      float sum = 0.f;
#pragma unroll
      for (int gq = 0; gq < 16; ++gq) sum += red[gq * BM + tid];
      red[tid] = sum;   // row 0 is only read by its own thread before this store
    }
    if (tid < BM && m0 + tid < p.M) atomicAdd(&p.colsum[m0 + tid], red[tid] * p.alpha);
  }
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// TN with dedicated loader waves (weight gradients): 256(m) x 128(n) output tile, K tiles of 64 token rows,
// 8 consumer waves (64x64 each, operands through ds_read_b64_tr_b16 as above) + 4 loader waves, three LDS
// stages, one workgroup per CU.  A stage is three [64 k][128] bf16 images (A columns 0..127, A columns
// 128..255, B), each with the tswz swizzle applied on the DMA source side.  grid.z splits K; partial sums are
// added with fp32 atomics.  The optional column sum of A (bias gradient) rides on the matrix cores: consumers
// with wn == 0 of the tn == 0 blocks multiply their A fragments with an all-ones B fragment.
// ---------------------------------------------------------------------------------------------
// MODE 0: partial sums added with fp32 atomics; 1: partial tiles to the slab (summing launch follows); 2: the ONLY writer
// of its tile (no K split): Cf tile += acc with plain 16-byte loads / stores
template <int MODE>
__device__ __forceinline__ void tn_lc_body(const GemmP& p, const int tm_, const int tn_, const int zz) {
  constexpr bool SLAB = MODE == 1;
  constexpr int IMG = TK * TP;                       // one [64][128] image, 16 KiB
  constexpr int STAGE_EL = 3 * IMG, NST = 3, LP = 12;
  __shared__ __attribute__((aligned(16))) bf16 lds[NST * STAGE_EL];   // 144 KiB
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool loader = wid >= 8;
  const int wm = (wid >> 1) & 3, wn = wid & 1;
  const int m0 = tm_ * 256, n0 = tn_ * 128;
  const int bz = zz / p.n_split, sp = zz % p.n_split;
  const int k_begin = sp * p.k_split;
  const int k_end = min(p.K, k_begin + p.k_split);
  if (!SLAB && k_begin >= k_end) return;             // block-uniform (the host sizes splits so that none is empty)
  const int nk = k_end > k_begin ? (k_end - k_begin + TK - 1) / TK : 0;
  const bool do_colsum = p.colsum != nullptr && tn_ == 0 && wn == 0 && !loader;
  f32x4 acc[4][4], cs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (loader) {
    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (long)bz * p.sA, p.a_bytes);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B + (long)bz * p.sB, p.b_bytes);
    // 48 pieces per stage: 16 per image, a piece = 4 k-rows x 256 B; loader l takes pieces l, l+4, ...
    const int lw = (wid - 8) & 3;
    uint32_t d_col[LP];     // byte offset of this lane's 16-B chunk inside its row, or ~0 if the column is out of range
    int d_row[LP];          // k row inside the tile
#pragma unroll
    for (int j = 0; j < LP; ++j) {
      const int piece = lw + 4 * j, img = piece >> 4, r = (piece & 15) * 4 + (lane >> 4);
      const int wsw = (r & 3) | (((r >> 3) & 1) << 2);
      const int lc = (lane & 15) ^ (wsw << 1);          // logical 16-B chunk stored at physical chunk lane&15
      d_row[j] = r;
      if (img < 2) d_col[j] = (m0 + img * 128 + lc * 8 < p.M) ? (uint32_t)((m0 + img * 128 + lc * 8) * 2) : 0xFFFFFFF0u;
      else d_col[j] = (n0 + lc * 8 < p.N) ? (uint32_t)((n0 + lc * 8) * 2) : 0xFFFFFFF0u;
    }
    auto dma_issue = [&](int kt, int stage) {
      bf16* sbase = lds + stage * STAGE_EL;
#pragma unroll
      for (int j = 0; j < LP; ++j) {
        const int piece = lw + 4 * j, img = piece >> 4;
        const int k = k_begin + kt * TK + d_row[j];
        const bool ok = kt < nk && k < k_end && d_col[j] != 0xFFFFFFF0u;
        bf16* dst = sbase + piece * 4 * TP;              // pieces are consecutive 1-KiB blocks of the stage
        if (img < 2) {
          const uint32_t o = ok ? (uint32_t)(((long)p.a_off + (long)k * p.lda) * 2) + d_col[j] : 0xFFFFFFF0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
        } else {
          const uint32_t o = ok ? (uint32_t)((long)k * p.ldb * 2) + d_col[j] : 0xFFFFFFF0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
        }
      }
    };
    dma_issue(0, 0);
    dma_issue(1, 1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int pre = stage == 0 ? 2 : stage - 1;
      dma_issue(kt + 2, pre);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;                                              // no barrier follows for the consumers either
  }
  // ---- consumers
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  union { bf16x8 v; uint32_t u[4]; } ones;
  ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3F803F80u;
  __builtin_amdgcn_s_barrier();
  // Two copies of the K loop, with and without the column-sum MFMAs: a wave-uniform `if` around them inside ONE loop
  // is a basic-block boundary after every k step, across which hipcc does not hoist the next step's fragment reads -
  // the LDS latency then sits in front of every 16 MFMAs (measured: 1.0 us per K tile instead of 0.5)
  auto kloop = [&](auto cs_tag) {
    constexpr bool CS = decltype(cs_tag)::value;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bf16* sa = lds + stage * STAGE_EL + (wm >> 1) * IMG;
      const bf16* sb = lds + stage * STAGE_EL + 2 * IMG;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bfr[4];
        const int kr = ks * 32 + g * 8 + q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int col = (wm & 1) * 64 + i * 16 + pp * 4;
          s16x4 lo = ds_tr(sa + tswz(kr, col)), hi = ds_tr(sa + tswz(kr + 4, col));
          union { bf16x8 v; s16x4 h[2]; } u; u.h[0] = lo; u.h[1] = hi; af[i] = u.v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = wn * 64 + j * 16 + pp * 4;
          s16x4 lo = ds_tr(sb + tswz(kr, col)), hi = ds_tr(sb + tswz(kr + 4, col));
          union { bf16x8 v; s16x4 h[2]; } u; u.h[0] = lo; u.h[1] = hi; bfr[j] = u.v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)   // operands swapped: acc holds the transposed tile, 4 consecutive output columns per lane
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if (CS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, af[i], cs[i], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_barrier();
      stage = stage == 2 ? 0 : stage + 1;
    }
  };
  if (do_colsum) kloop(std::true_type{});
  else kloop(std::false_type{});
  // lane (fr, fq) of tile (i, j) holds output row i*16 + fr and the four consecutive columns j*16 + fq*4 .. +3
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = m0 + wm * 64 + i * 16 + fr, col = n0 + wn * 64 + j * 16 + fq * 4;
      if (row >= p.M || col >= p.N) continue;      // N % 8 == 0: the four columns are in or out together
      // a workgroup's 128 KiB of atomics drain at ~one wave-instruction per 50 ns per CU (25 us, as long as the
      // whole K loop of an encoder wgrad); plain 16-byte stores of the partial tile + a summing launch cost a third
      if (SLAB) {
        *(f32x4*)(p.slab + ((long)zz * p.M + row) * p.N + col) =
            f32x4{acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
      } else if (MODE == 2) {
        f32x4* dst = (f32x4*)(p.Cf + (long)row * p.ldc + col);
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!p.overwrite) o = *dst;            // workgroup-uniform
        o[0] += acc[i][j][0] * p.alpha; o[1] += acc[i][j][1] * p.alpha; o[2] += acc[i][j][2] * p.alpha; o[3] += acc[i][j][3] * p.alpha;
        *dst = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&p.Cf[(long)row * p.ldc + col + r], acc[i][j][r] * p.alpha);
      }
    }
  if (do_colsum && fq == 0) {   // every accumulator row of cs holds the same sums: lane fr owns output row i*16 + fr
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + fr;
      if (row < p.M) atomicAdd(&p.colsum[row], cs[i][0] * p.alpha);
    }
  }
}

template <bool SLAB>
__global__ __launch_bounds__(768) void gemm_tn_lc_kernel(GemmP p) {
  // XCD-aware order over the WHOLE 3-D grid: workgroups are dealt to the 8 XCDs round-robin in dispatch order
  // (x fastest, then y, then z); every XCD gets one contiguous run of (K-split, tile) pairs, K-split major and
  // tiles in group-M order, i.e. a few M tiles x all N tiles of one token range.  With the per-plane order the
  // three planes of an encoder wgrad put three different token ranges on every XCD and the kernel fetched 3.2x
  // its algorithmic bytes (rocprofv3 FETCH_SIZE: 199 MB per launch, ~5 TB/s of fabric traffic).
  int tm_, tn_, zz;
  {
    const int plane = gridDim.x * gridDim.y, nwg = plane * gridDim.z;
    const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = lin & 7;
    const int id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (lin >> 3);
    zz = id / plane;
    const int t = id - zz * plane, ntm = gridDim.y, ntn = gridDim.x, GM = 4;
    const int per_group = GM * ntn, group = t / per_group, first_m = group * GM;
    const int gsz = min(ntm - first_m, GM), in_group = t - group * per_group;
    tm_ = first_m + in_group % gsz;
    tn_ = in_group / gsz;
  }
  tn_lc_body<SLAB ? 1 : 0>(p, tm_, tn_, zz);
}

// Several weight-gradient GEMMs that share the reduction dimension (the four of one encoder layer) as ONE launch, no K
// split: 54 + 18 + 72 + 72 tiles fill 216 CUs with full-length K loops, every tile has a single writer (no partial-tile
// slab, no summing launch, no atomics), and one ramp-up / tail instead of four.
struct TnGroupP { GemmP p[4]; int first[5]; };
__global__ __launch_bounds__(768) void gemm_tn_group_kernel(TnGroupP g) {
  const int nwg = gridDim.x, lin = blockIdx.x;
  const int qd = nwg >> 3, rm = nwg & 7, xcd = lin & 7;
  const int id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (lin >> 3);   // contiguous run per XCD
  int pi = 0;
  if (id >= g.first[1]) pi = 1;
  if (id >= g.first[2]) pi = 2;
  if (id >= g.first[3]) pi = 3;
  pi = __builtin_amdgcn_readfirstlane(pi);
  const GemmP& p = g.p[pi];
  const int t = id - g.first[pi];
  const int ntm = (p.M + 255) / 256, ntn = (p.N + 127) / 128, GM = 4;
  const int per_group = GM * ntn, group = t / per_group, first_m = group * GM;
  const int gsz = min(ntm - first_m, GM), in_group = t - group * per_group;
  tn_lc_body<2>(p, first_m + in_group % gsz, in_group / gsz, 0);
}

// ---------------------------------------------------------------------------------------------
// TN, EIGHT-PHASE (round 3): the weight gradients of an encoder layer on 256 x 256 output tiles - 1.5 x the MFMA work per
// staged byte of the 256 x 128 loader/consumer tiles above, whose operand stream (not their MFMAs) set the pace - with the
// phase structure of gemm_nt8_kernel: 8 waves = 2 (m) x 4 (n), every wave stages and computes, three half-tiles in flight
// behind a counted vmcnt, wave groups staggered by one barrier.  A K tile is 64 token rows; its four half-tile images are
// [64 k][128] bf16 (A_h: the m-half h columns of both wave groups, B_h: the n-half h columns of all four wave columns) in
// the k-major orientation of global memory; fragments come out through ds_read_b64_tr_b16 (tswz swizzle on the DMA source).
// 256 x 256 tiles of a base layer number 108, so the token dimension is split S = 2 ways over workgroup PAIRS (adjacent
// ids: same XCD) that exchange half a partial tile each at the end (see the tail of the kernel): every output element
// still has a single writer per launch - no atomics (but the bias column sums), no summing launch.
// Measured while building it (timing-only ablations, cfgB layer, 216 workgroups x 52 K tiles): the loop runs 1.45 us per
// K tile with or without its memory traffic (MFMA floor 1.05-1.15 us; fragment reads + barriers alone 1.06 us), i.e. it is
// issue-bound, not operand-bound like the 256 x 128 kernel; a one-sided hand-off (256 KB slab, plain stores + release
// fence, last arriver combines) cost ~30 us per launch and erased the gain, hence the symmetric write-through form.
// ---------------------------------------------------------------------------------------------
constexpr int TN8_MAXP = 12;   // problems per grouped launch: the caller packs whole GEMMs of up to four layers into one round of the chip
struct Tn8GroupP { GemmP p[TN8_MAXP]; int first[TN8_MAXP + 1]; int S; float* slab; long slab_bytes; unsigned* cnt; int dbg; };   // dbg: timing-only ablations
// the tr read as inline asm: behind the builtin hipcc puts an s_waitcnt vmcnt(0) in front of the first read of every loop
// iteration (an LDS read without a memory operand "may alias" every LDS-DMA in flight), which drains the staging pipeline.
// The wave waits for these reads itself: s_waitcnt lgkmcnt(0) + sched_barrier behind the phase's first barrier.
template <int OFF> __device__ __forceinline__ s16x4 ds_tr_asm(uint32_t lds_byte_addr) {   // OFF: immediate byte offset (< 64 KiB)
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF> __device__ __forceinline__ bf16x8 tr_frag(uint32_t lds_byte_addr) {      // k rows q and q + 4 of a 32-deep step
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = ds_tr_asm<OFF>(lds_byte_addr);
  u.h[1] = ds_tr_asm<OFF + 4 * TP * 2>(lds_byte_addr);
  return u.v;
}

__global__ __launch_bounds__(512) void gemm_tn8_group_kernel(Tn8GroupP g) {
  constexpr int IMG = TK * TP;                    // [64 k][128] bf16 = 16 KiB
  constexpr int BUF_EL = 4 * IMG;                 // A_h0 A_h1 B_h0 B_h1
  __shared__ __attribute__((aligned(16))) bf16 lds[2 * BUF_EL];     // 128 KiB
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  int tile, sp;
  {
    const int nwg = gridDim.x, lin = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = lin & 7;
    const int id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (lin >> 3);   // contiguous run per XCD
    tile = id / g.S; sp = id - tile * g.S;     // (split-major ids - pairs on different XCDs, more shared panels per XCD - measured the same)
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < TN8_MAXP; ++i)
    if (tile >= g.first[i]) pi = i;
  pi = __builtin_amdgcn_readfirstlane(pi);
  const GemmP& p = g.p[pi];
  const int t = tile - g.first[pi];
  const int ntn = (p.N + 255) / 256;
  const int tm = t / ntn, tn = t - tm * ntn;
  const int m0 = tm * 256, n0 = tn * 256;
  const int units = (p.K + TK - 1) / TK;
  const int u0 = (int)((long)sp * units / g.S), u1 = (int)((long)(sp + 1) * units / g.S);
  const int nk = u1 - u0, nk2 = (nk + 1) & ~1;
  __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes), rb = make_rsrc(p.B, p.b_bytes);
  // ---- staging: piece pc = j * 8 + wid of an image = k rows pc*4 .. pc*4+3 (lane >> 4 = row, lane & 15 = physical chunk)
  const int l4 = lane >> 4;
  const int wsw = l4 | (((wid >> 1) & 1) << 2);                 // tswz window of row (j*32 + wid*4 + l4): independent of j
  const int lc = (lane & 15) ^ (wsw << 1);                      // logical 16-byte chunk (8 columns) this lane fetches
  const uint32_t a_thr = (uint32_t)((p.a_off + (long)(u0 * TK + wid * 4 + l4) * p.lda + m0 + (lc >> 3) * 128 + (lc & 7) * 8) * 2);
  const uint32_t b_thr = (uint32_t)(((long)(u0 * TK + wid * 4 + l4) * p.ldb + n0 + (lc >> 2) * 64 + (lc & 3) * 8) * 2);
  const uint32_t a_j = (uint32_t)(32 * p.lda * 2), b_j = (uint32_t)(32 * p.ldb * 2);      // piece j = 1: 32 rows further
  const uint32_t a_kt = (uint32_t)(TK * p.lda * 2), b_kt = (uint32_t)(TK * p.ldb * 2);    // next K tile
  int kt_s = 0;
#ifdef W2VS_ABLATION
  const bool live = !(g.dbg & 1);                 // dbg bit 0: every DMA out of range (zero fill, no memory traffic)
#else
  constexpr bool live = true;
#endif
  bool ok_cur = nk > 0 && live, ok_prev = false;
  uint32_t a_cur = a_thr, b_cur = b_thr, a_prev = 0;
  auto advance = [&]() {
    a_prev = a_cur; ok_prev = ok_cur;
    ++kt_s;
    a_cur += a_kt; b_cur += b_kt;
    ok_cur = kt_s < nk && live;
  };
  auto stage_A = [&](int buf, int h, uint32_t base, bool ok) {    // h: + 64 columns
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t o = ok ? base + (j ? a_j : 0u) + (uint32_t)(h * 128) : 0xFFFFFFF0u;
      bf16* dst = lds + buf * BUF_EL + h * IMG + (j * 8 + wid) * 512;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
    }
  };
  auto stage_B = [&](int buf, int h, uint32_t base, bool ok) {    // h: + 32 columns
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t o = ok ? base + (j ? b_j : 0u) + (uint32_t)(h * 64) : 0xFFFFFFF0u;
      bf16* dst = lds + buf * BUF_EL + (2 + h) * IMG + (j * 8 + wid) * 512;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (void __attribute__((address_space(3)))*)dst, 16, o, 0, 0, 0);
    }
  };
  // ---- fragment addresses (elements): 16-lane group gq owns k rows 8 gq .. 8 gq + 7 of a 32-deep step; inside it lane
  // 4 q + pp supplies row q (and q + 4), columns 4 pp .. 4 pp + 3, and receives column (lane & 15)
  const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int rsw = (q | ((gq & 1) << 2)) << 4;                   // tswz of rows (ks*32 + gq*8 + q) and (.. + 4): the same window
  // (byte addresses inside the LDS, one set per K-tile buffer: buffer 1 lies 64 KiB up, beyond an immediate offset)
  uint32_t a_tr[2][4], b_tr[2][2];
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
    // fragment slot i of wave column wc holds i-tile (i + wc) & 3 of the m-half: the wave's share of the bias column sums
    // (one i-tile per wave column) is then always slot 0 - a wave-dependent CHOICE among the slots would make them a
    // runtime-indexed array, which hipcc keeps in scratch memory
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a_tr[eb][i] = lds0 + (uint32_t)(eb * BUF_EL + (gq * 8 + q) * TP + ((wr * 64 + ((i + wc) & 3) * 16 + pp * 4) ^ rsw)) * 2;
#pragma unroll
      for (int j1 = 0; j1 < 2; ++j1)
        b_tr[eb][j1] = lds0 + (uint32_t)(eb * BUF_EL + 2 * IMG + (gq * 8 + q) * TP + ((wc * 32 + j1 * 16 + pp * 4) ^ rsw)) * 2;
    }
  }
  constexpr int KS1 = 32 * TP * 2, IMGB = IMG * 2;      // byte offsets: second 32-deep k step; next half-tile image
  // bias gradient (column sums of A) on the matrix cores, spread over the four wave columns of the tn == 0 tiles: wave
  // column wc takes the i-tile wc of each m-half against an all-ones fragment
  const bool do_cs = p.colsum != nullptr && tn == 0;
  union { bf16x8 v; uint32_t u[4]; } ones;
  ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3F803F80u;

  // ---- prologue
  stage_B(0, 0, b_cur, ok_cur); stage_A(0, 0, a_cur, ok_cur); stage_B(0, 1, b_cur, ok_cur); stage_A(0, 1, a_cur, ok_cur);
  advance();
  stage_B(1, 0, b_cur, ok_cur); stage_A(1, 0, a_cur, ok_cur); stage_B(1, 1, b_cur, ok_cur);
  advance();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  f32x4 acc[2][4][4], cs[2];
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
    cs[mh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mh][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 af[4][2], b0[2][2], b1[2][2];
  auto mfma_q = [&](int mh, int nh, bf16x8 (&bb)[2][2], bool with_cs) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this phase's (inline-asm) fragment reads
    __builtin_amdgcn_sched_barrier(0);                     // ... and no MFMA may move above the wait
#ifdef W2VS_ABLATION
    if (g.dbg & 2) return;                                 // dbg bit 1: no MFMAs
#endif
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j1 = 0; j1 < 2; ++j1)   // operands swapped: lane (fr, fq) holds output row i*16 + fr, columns j*16 + 4 fq .. + 3
          acc[mh][i][nh * 2 + j1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j1][ks], af[i][ks], acc[mh][i][nh * 2 + j1], 0, 0, 0);
    if (with_cs) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) cs[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, af[0][ks], cs[mh], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto ktile = [&](auto ebuf) {
    constexpr int EB = decltype(ebuf)::value;
    // ---- phase 1: B_h0 first; lgkmcnt counts to 15 only, so the wait that retires those 4 fragments (8 reads) sits
    // after 7 of the 16 A reads
#pragma unroll
    for (int j1 = 0; j1 < 2; ++j1) { b0[j1][0] = tr_frag<0>(b_tr[EB][j1]); b0[j1][1] = tr_frag<KS1>(b_tr[EB][j1]); }
    af[0][0] = tr_frag<0>(a_tr[EB][0]); af[0][1] = tr_frag<KS1>(a_tr[EB][0]); af[1][0] = tr_frag<0>(a_tr[EB][1]);
    {
      union { bf16x8 v; s16x4 h[2]; } u;
      u.h[0] = ds_tr_asm<KS1>(a_tr[EB][1]);
      asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
      u.h[1] = ds_tr_asm<KS1 + 4 * TP * 2>(a_tr[EB][1]);
      af[1][1] = u.v;
    }
    af[2][0] = tr_frag<0>(a_tr[EB][2]); af[2][1] = tr_frag<KS1>(a_tr[EB][2]);
    af[3][0] = tr_frag<0>(a_tr[EB][3]); af[3][1] = tr_frag<KS1>(a_tr[EB][3]);
    stage_A(EB ^ 1, 1, a_prev, ok_prev);
    __builtin_amdgcn_s_barrier();
    mfma_q(0, 0, b0, do_cs);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2
#pragma unroll
    for (int j1 = 0; j1 < 2; ++j1) { b1[j1][0] = tr_frag<IMGB>(b_tr[EB][j1]); b1[j1][1] = tr_frag<IMGB + KS1>(b_tr[EB][j1]); }
    stage_B(EB, 0, b_cur, ok_cur);
    __builtin_amdgcn_s_barrier();
    mfma_q(0, 1, b1, false);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i][0] = tr_frag<IMGB>(a_tr[EB][i]); af[i][1] = tr_frag<IMGB + KS1>(a_tr[EB][i]); }
    stage_A(EB, 0, a_cur, ok_cur);
    __builtin_amdgcn_s_barrier();
    mfma_q(1, 1, b1, do_cs);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4
    stage_B(EB, 1, b_cur, ok_cur);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    mfma_q(1, 0, b0, false);
    __builtin_amdgcn_s_barrier();
    advance();
  };
  for (int kp = 0; kp < nk2; kp += 2) {
    ktile(std::integral_constant<int, 0>{});
    ktile(std::integral_constant<int, 1>{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();             // pairs with group 1's extra barrier: both groups in step again
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trailing (out-of-range) prefetches have landed: the LDS is free
  __builtin_amdgcn_s_barrier();

  const int fr = lane & 15, fq = lane >> 4;
  // ---- split K over a PAIR of workgroups (S == 2), symmetric exchange: workgroup sp keeps the column half sp of the tile
  // (wave columns 2 sp, 2 sp + 1) and gives the other half away.  The four giving waves park their accumulators in the slab
  // with write-through (sc1) 16-byte stores, lane-linear (the partner's waves have the same lane -> element map), drain,
  // and after the workgroup barrier one lane raises flag[tile][sp].  Then one lane polls the PARTNER's flag, lowers it
  // again (the flags are zero between launches), and after another barrier the four keeping waves add the partner's half
  // (sc1 loads: no acquire needed, guide Guideline 16 / MI355X_MICROARCH "Valid forms") and update Cf with plain 16-byte
  // loads / stores.  Every output element has one writer per launch; both workgroups work during the exchange.
  bool keep = true;
  if (g.S == 2) {
    keep = (wc >> 1) == sp;
    const int ws = wr * 2 + (wc & 1);                       // wave slot inside a half: 0 .. 3
    __amdgpu_buffer_rsrc_t rs = make_rsrc(g.slab, (uint32_t)g.slab_bytes);
    unsigned* flags = g.cnt + tile * 2;
    if (!keep) {
      const uint32_t o = (uint32_t)(((tile * 2 + sp) * 4 + ws) * 32) * 1024u + (uint32_t)lane * 16u;   // 32 x 1 KiB per wave
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            union { f32x4 f; u32x4 u; } cv; cv.f = acc[mh][i][j];
            __builtin_amdgcn_raw_buffer_store_b128(cv.u, rs, o + (uint32_t)(((mh * 4 + i) * 4 + j) * 1024), 0, 16);   // aux 16 = sc1
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains before the flag
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(flags + sp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(flags + (sp ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(2);
      __hip_atomic_store(flags + (sp ^ 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
    __syncthreads();
    if (keep) {
      const uint32_t o = (uint32_t)(((tile * 2 + (sp ^ 1)) * 4 + ws) * 32) * 1024u + (uint32_t)lane * 16u;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            union { f32x4 f; u32x4 u; } cv;
            cv.u = __builtin_amdgcn_raw_buffer_load_b128(rs, o + (uint32_t)(((mh * 4 + i) * 4 + j) * 1024), 0, 16);
            acc[mh][i][j][0] += cv.f[0]; acc[mh][i][j][1] += cv.f[1]; acc[mh][i][j][2] += cv.f[2]; acc[mh][i][j][3] += cv.f[3];
          }
    }
  }
  if (keep && !(g.dbg & 4)) {                              // dbg bit 2: the output tile is not stored
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = m0 + wr * 128 + mh * 64 + ((i + wc) & 3) * 16 + fr, col = n0 + wc * 64 + j * 16 + fq * 4;
          if (row >= p.M || col >= p.N) continue;      // N % 8 == 0: the four columns are in or out together
          f32x4* dst = (f32x4*)(p.Cf + (long)row * p.ldc + col);
          f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
          if (!p.overwrite) o = *dst;          // workgroup-uniform: the first writer of a step skips the read (and nobody zeroed it)
          o[0] += acc[mh][i][j][0] * p.alpha; o[1] += acc[mh][i][j][1] * p.alpha;
          o[2] += acc[mh][i][j][2] * p.alpha; o[3] += acc[mh][i][j][3] * p.alpha;
          *dst = o;
        }
  }
  if (do_cs && fq == 0 && !(g.dbg & 8)) {   // (dbg bit 3: no bias column sums)  every accumulator row of cs holds the same sums: lane fr owns output row (i-tile wc = slot 0) * 16 + fr
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      const int row = m0 + wr * 128 + mh * 64 + wc * 16 + fr;
      if (row < p.M) atomicAdd(&p.colsum[row], cs[mh][0] * p.alpha);
    }
  }
}

// Cf[m][n] += sum over parts of slab[part][m][n]   (M x N fp32, N % 4 == 0; Cf rows are ldc apart)
__global__ __launch_bounds__(256) void tn_slab_reduce_kernel(const float* slab, int parts, int M, int N, long ldc, float* Cf) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;      // one float4 per thread
  const long total4 = (long)M * N / 4;
  if (i4 >= total4) return;
  const long MN = (long)M * N;
  f32x4 s4 = *(const f32x4*)(slab + i4 * 4);
  for (int q = 1; q < parts; ++q) {
    const f32x4 v = *(const f32x4*)(slab + q * MN + i4 * 4);
    s4[0] += v[0]; s4[1] += v[1]; s4[2] += v[2]; s4[3] += v[3];
  }
  const long e = i4 * 4, row = e / N, col = e - row * N;
  float* dst = Cf + row * ldc + col;
  f32x4 o = *(f32x4*)dst;
  o[0] += s4[0]; o[1] += s4[1]; o[2] += s4[2]; o[3] += s4[3];
  *(f32x4*)dst = o;
}

// Launch-level profiling hooks (bench.py "roofline"): every stride-th GEMM launch is bracketed by a
// pair of HIP events ON THE STREAM IT IS LAUNCHED ON; totals are read back after the timed region.
struct ProfSample { hipEvent_t e0, e1; int id; double flops; };
static std::vector<ProfSample> g_prof;
static int g_prof_stride = 0;
static long g_prof_count = 0;
static long g_prof_launches[64] = {0};      // every launch per id since prof_enable (the sampled ones are a subset)
static double g_prof_flops_all[64] = {0};   // algorithmic FLOPs of every launch per id (scales the sample to the whole family)
void prof_enable(int stride) {
  for (auto& s : g_prof) { (void)hipEventDestroy(s.e0); (void)hipEventDestroy(s.e1); }
  g_prof.clear();
  g_prof_stride = stride;
  g_prof_count = 0;
  for (long& c : g_prof_launches) c = 0;
  for (double& f : g_prof_flops_all) f = 0;
}
static inline hipEvent_t prof_begin(hipStream_t st, int own_stride = 0) {
  if (g_prof_stride <= 0) return nullptr;
  static long own_count = 0;                       // kernels launched a few times per step keep their own, denser sample
  if (own_stride > 0 ? (own_count++ % own_stride) != 0 : (g_prof_count++ % g_prof_stride) != 0) return nullptr;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  (void)hipEventRecord(e, st);
  return e;
}
static inline void prof_end(hipEvent_t e0, int id, double flops, hipStream_t st) {
  if (g_prof_stride > 0 && id >= 0 && id < 64) { ++g_prof_launches[id]; g_prof_flops_all[id] += flops; }
  if (!e0) return;
  hipEvent_t e1;
  if (hipEventCreate(&e1) != hipSuccess) return;
  (void)hipEventRecord(e1, st);
  g_prof.push_back({e0, e1, id, flops});
}
long prof_launches(int id) { return (id >= 0 && id < 64) ? g_prof_launches[id] : 0; }
double prof_flops_all(int id) { return (id >= 0 && id < 64) ? g_prof_flops_all[id] : 0.0; }
// Totals over the timed samples of a family.  An event pair also times any gap in which the stream waited for the HOST to
// enqueue the launch (a late launch thread: one 0.8 ms sample among seven once made a 45 us kernel look like 150 us), so samples
// whose time per FLOP exceeds 3 x the family's median are left out - with fewer than four samples nothing is.
static int prof_read_impl(int id, bool filter, double* total_ms, double* total_flops, int* launches);
int prof_read(int id, double* total_ms, double* total_flops, int* launches) { return prof_read_impl(id, true, total_ms, total_flops, launches); }
// the same totals with NOTHING left out (bench.py prints both and the number of samples the filter dropped)
int prof_read_raw(int id, double* total_ms, double* total_flops, int* launches) { return prof_read_impl(id, false, total_ms, total_flops, launches); }
static int prof_read_impl(int id, bool filter, double* total_ms, double* total_flops, int* launches) {
  std::vector<std::pair<double, double>> v;      // (ms, flops)
  for (auto& s : g_prof) {
    if (s.id != id) continue;
    if (hipEventSynchronize(s.e1) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.e0, s.e1) != hipSuccess) continue;
    v.emplace_back((double)t, s.flops);
  }
  double cut = 1e300;
  if (filter && v.size() >= 4) {
    std::vector<double> r;
    for (auto& x : v) r.push_back(x.first / std::max(x.second, 1.0));
    std::nth_element(r.begin(), r.begin() + r.size() / 2, r.end());
    cut = 3.0 * r[r.size() / 2];
  }
  double ms = 0, fl = 0;
  int n = 0;
  for (auto& x : v) {
    if (x.first / std::max(x.second, 1.0) > cut) continue;
    ms += x.first; fl += x.second; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  return 0;
}

static int check_common(const GemmDesc& d) {
  if (!d.A || !d.B) return set_error("gemm: null operand");
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return set_error("gemm: non-positive dimension");
  if ((d.lda % 8) || (d.ldb % 8) || (d.a_off % 8)) return set_error("gemm: lda/ldb/a_off must be multiples of 8 elements (16-B loads)");
  if (((uintptr_t)d.A % 16) || ((uintptr_t)d.B % 16)) return set_error("gemm: operands must be 16-B aligned");
  return 0;
}

// Variant overrides for tests and tuning (process-wide, not thread safe): -1 / 0 = automatic choice.
static int g_force_nt_mode = -1, g_force_lc_h = 0, g_force_tn_lc = -1;
void gemm_tune(int nt_mode, int lc_height, int tn_lc) { g_force_nt_mode = nt_mode; g_force_lc_h = lc_height; g_force_tn_lc = tn_lc; }

int gemm_nt(const GemmDesc& d, hipStream_t s) {
  if (int e = check_common(d)) return e;
  if (d.epi == EPI_F32 ? !d.Cf : !d.C) return set_error("gemm_nt: null output");
  if ((d.ldc % 8) || (d.N % 8)) return set_error("gemm_nt: N and ldc must be multiples of 8");
  if (d.K % 8) return set_error("gemm_nt: K must be a multiple of 8");
  if (d.ldc > (1L << 21)) return set_error("gemm_nt: ldc must be <= 2^21 (32-bit tile-relative output offsets)");
  if ((epi_is_dact(d.epi) || d.epi == EPI_ADD) && !d.aux) return set_error("gemm_nt: DGELU/MUL/ADD need aux");
  if (epi_is_save(d.epi) && !d.C2) return set_error("gemm_nt: GELU_SAVE / GELU_SAVEG need C2");
  GemmP p{};
  p.A = (const bf16*)d.A; p.B = (const bf16*)d.B; p.C = (bf16*)d.C; p.C2 = (bf16*)d.C2; p.Cf = d.Cf;
  p.bias = (const bf16*)d.bias; p.aux = (const bf16*)d.aux;
  p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.a_off = d.a_off;
  p.sA = d.sA; p.sB = d.sB; p.sC = d.sC; p.alpha = d.alpha;
  p.c_elems = d.c_elems ? d.c_elems : (long)(d.M - 1) * d.ldc + d.N;
  long a_ext = d.a_bytes ? d.a_bytes : ((long)d.a_off + (long)(d.M - 1) * d.lda + d.K) * 2;
  long b_ext = d.b_bytes ? d.b_bytes : ((long)(d.N - 1) * d.ldb + d.K) * 2;
  if (a_ext <= 0 || a_ext >= 0x7FFFFFF0L || b_ext >= 0x7FFFFFF0L) return set_error("gemm_nt: operand extent must be < 2 GiB per batch");
  p.a_bytes = (uint32_t)a_ext; p.b_bytes = (uint32_t)b_ext;
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.batch > 0 ? d.batch : 1), block(256);
  // single-buffer / 3-blocks-per-CU form when the grid has enough tiles to fill it (measured +6..10 % on
  // the QKV / fc1 / conv shapes), double-buffer / 2-per-CU for short grids with a long K (fc2, out_proj)
  hipEvent_t pe = prof_begin(s);
  static const int mode_env = W2VS_ENV_INT("W2VS_GEMM_MODE", -1);
  const long ntiles = (long)grid.x * grid.y * grid.z;
  // measured on MI355X: LDS-DMA staging wins on encoder-sized grids (+8..15 % on the N=768 and QKV shapes),
  // the register-staged 3-blocks-per-CU form on the very large conv grids (+5..8 %)
  int mode = ntiles >= 1500 ? 1 : 2;
  // Loader/consumer kernels (one workgroup per CU) against the 128x128 kernels (two / three per CU); time models
  // fitted to rocprofv3 timings:  128^2: rounds x (K tiles x 0.47 us + 2.8 us);  loader/consumer, tile height
  // 256 / 192 / 160: rounds x (K tiles x {0.74, 0.71, 0.62} us + 7.5 us) - the K tile is LDS-DMA bound at ~70 GB/s
  // per CU, and with more CUs busy the shared L2 paces them, so the time is not proportional to the tile bytes.
  // The lower heights put 210 / 246 instead of 156 workgroups on the 256 CUs for the N = 768 outputs.  On the
  // huge conv grids the PERSISTENT form (tile loop inside the kernel, 256-row tiles) replaces the register-staged
  // 3-per-CU kernel; on the encoder shapes it measured 1 % slower end to end than separate workgroups.
  int lc_h = 256;
  const bool p_ok = (d.N % 8) == 0 && (d.ldc % 8) == 0 && d.epi != EPI_F32 && ((uintptr_t)d.C % 16) == 0 &&
                    ((uintptr_t)d.C2 % 16) == 0 && ((uintptr_t)d.aux % 16) == 0 && (d.sC % 8) == 0;
  static const int conv_p_env = W2VS_ENV_INT("W2VS_CONV_PERSIST", 1);
  bool wide_auto = false;
  static const int conv_model_env = W2VS_ENV_INT("W2VS_CONV_TILE_MODEL", 1);   // 0: 256 x 128 on the conv grids (A/B)
  if (mode == 1 && p_ok && conv_p_env && (d.N < 256 || !conv_model_env)) {
    mode = 5;
  } else if ((mode == 1 || mode == 2) && p_ok && conv_p_env && ntiles >= 256 && d.N >= 256) {
    // (the huge conv grids too: 160 x 256 tiles run conv1 forward 267 -> 205 us, with GELU + saved gelu' 319 -> 255, conv2
    // forward 158 -> 135, conv1 dgrad 372 -> 345 against the 256 x 128 tiles they used to get unconditionally)
    // Round 2: at least a chip's worth of output -> the PERSISTENT loader/consumer kernel (register-direct epilogue, the DMA ring
    // keeps running across tile boundaries) with the tile shape that minimises rounds x (rows + columns) - the staged bytes per
    // K step set the step time (~36 GB/s per CU with every CU streaming), the rounds of 256 workgroups the number of K loops.
    // Measured at R = 6544 (tools/gemm_probe.py, us, bias / GELU+save / x aux):
    //   N 3072, K 768: 160x256 tiles 36 / 44 / 44   against 43 / 61 / 56 for the round-1 choice (128^2, 2 per CU)
    //   N 2304, K 768: 256x128 tiles 30 / 38 / 34   against 33 / 41 / 36
    //   N  768, K 3072: 160x128 tiles 33 / 36 / 34  against 33 / 39 / 35
    const long nbz = d.batch > 0 ? d.batch : 1;
    const int hs[4] = {256, 192, 160, 160}, ws[4] = {128, 128, 128, 256};
    long best = -1;
    for (int c = 0; c < 4; ++c) {
      const long t8 = (long)((d.N + ws[c] - 1) / ws[c]) * ((d.M + hs[c] - 1) / hs[c]) * nbz;
      const long cost = ((t8 + 255) / 256) * (hs[c] + ws[c]);
      if (best < 0 || cost < best) { best = cost; lc_h = hs[c]; wide_auto = ws[c] == 256; }
    }
    mode = 5;
    // Round 3: the 8-phase kernel (256 x 256 / 320 x 256 tiles, every wave stages and computes).  Per staged byte it does 1.3x
    // the work of a 160 x 256 tile, but its three half-tiles in flight fill the LDS at ~45 GB/s per CU against ~52 for the
    // dedicated loader waves, and its tile epilogue is longer: measured time ~ 1.35 x the rounds x (rows + columns) proxy
    // relative to the kernels above (tools/gemm_probe.py on QKV / fc1 / conv1 dgrad / conv2 / 4096^3: predicted 0.90 - 0.93,
    // measured 0.92 - 0.95; conv1 forward predicted 1.07, measured 1.04).
    static const int nt8_env = W2VS_ENV_INT("W2VS_NT8", 1);
    if (nt8_env && (d.K % 64) == 0 && ((uintptr_t)d.bias % 16) == 0) {
      const int h8[2] = {256, 320};
      for (int c = 0; c < 2; ++c) {
        const long t8 = (long)((d.N + 255) / 256) * ((d.M + h8[c] - 1) / h8[c]) * nbz;
        const long cost = (((t8 + 255) / 256) * (h8[c] + 256) * 135 + 99) / 100;
        if (cost < best) { best = cost; lc_h = h8[c]; mode = 8; wide_auto = false; }
      }
    }
  } else if (mode == 2) {
    const double nkt = (d.K + BK - 1) / BK;
    const long nbz = d.batch > 0 ? d.batch : 1;
    // Below one workgroup per CU the 128^2 kernel has no co-resident partner to overlap its loads with (0.8 us per K tile
    // measured on 102 tiles x 48 K tiles, against 0.47 with two per CU), while the loader/consumer kernels speed up with
    // fewer CUs streaming (0.40 us per K tile for 78 workgroups of 160 rows): the last layer's pruned fc2 forward /
    // fc1 dgrad (2080 x 768 x 3072) ran 39 us on the former and 26.5 us on the latter.
    double best = 0.95 * (double)((ntiles + 255) / 256) * (nkt * (ntiles <= 256 ? 0.8 : 0.47) + 2.8);
    const int hs[3] = {256, 192, 160};
    const double tk[3] = {0.74, 0.71, 0.62};
    for (int c = 0; c < 3; ++c) {
      const long t8 = (long)((d.N + 127) / 128) * ((d.M + hs[c] - 1) / hs[c]) * nbz;
      const double fill = t8 >= 256 ? 1.0 : std::max(0.6, (double)t8 / 256.0);
      const double t = (double)((t8 + 255) / 256) * (nkt * tk[c] * fill + 7.5 + (epi_is_save(d.epi) ? 4.0 : 0.0));
      if (t < best) { best = t; mode = 3; lc_h = hs[c]; }
    }
    // Round 4: 64 x 128 tiles (2 consumer waves + 4 loaders, 72 KiB of LDS: two workgroups per CU) for SHORT outputs - the
    // streaming encoder's M = B x N' of a few hundred rows, where 160-row tiles leave three quarters of the chip idle behind a
    // 48-K-tile loop (fc2 at a 10 s prefix: 30 tiles, 32 us).  Taken only below one chip's worth of 64-row tiles x 2.
    static const int lc64_env = W2VS_ENV_INT("W2VS_LC64", 1);
    {
      const long t64 = (long)((d.N + 127) / 128) * ((d.M + 63) / 64) * nbz;
      if (lc64_env && t64 <= 512) {
        const double fill = std::max(0.5, std::min(1.0, (double)t64 / 512.0));
        const double t = nkt * 0.42 * fill + 6.0 + (epi_is_save(d.epi) ? 2.0 : 0.0);
        if (t < best) { best = t; mode = 3; lc_h = 64; }
      }
    }
  }
  if (mode_env >= 0) mode = mode_env;
  if (g_force_nt_mode >= 0) mode = g_force_nt_mode;
  {
    // W2VS_NT_FORCE="N:K:epi=mode:height,..." (tuning): the kernel form / tile height for every launch of that output width,
    // depth and epilogue - how tile choices are A/B-ed IN THE STEP, where operands are cold and probe rankings do not hold
    struct Force { int N, K, epi, mode, h; };
    static const std::vector<Force> forces = [] {
      std::vector<Force> v;
      const char* e = W2VS_ENV_STR("W2VS_NT_FORCE");
      while (e && *e) {
        Force f{};
        if (sscanf(e, "%d:%d:%d=%d:%d", &f.N, &f.K, &f.epi, &f.mode, &f.h) == 5) v.push_back(f);
        e = strchr(e, ',');
        if (e) ++e;
      }
      return v;
    }();
    for (const Force& f : forces)
      if (f.N == d.N && f.K == d.K && f.epi == d.epi && d.M >= 2048) { mode = f.mode; lc_h = f.h; wide_auto = false; }
  }
  if ((mode == 5 || mode == 6) && !p_ok) return set_error("gemm_nt: the persistent kernel needs N % 8 == 0, ldc % 8 == 0, 16-byte aligned outputs");
  static const int lc_env = W2VS_ENV_INT("W2VS_LC_H", 0);
  if (lc_env > 0) lc_h = lc_env;
  if (g_force_lc_h > 0) lc_h = g_force_lc_h;
  // lc_h = 1160 selects the 160 x 256 loader/consumer tile (WN = 4); tuning build: 2160 = the same tile on FOUR consumer waves
  // (one per SIMD, 160 x 64 outputs each) - the price of a lone consumer wave per SIMD, what a ping-pong schedule would run on
  bool lone4 = false;
#ifdef W2VS_TUNING
  if (lc_h == 2160) { lone4 = true; lc_h = 1160; }
#endif
  const bool wide = (lc_h == 1160) || (wide_auto && lc_h == 160 && lc_env <= 0 && g_force_lc_h <= 0 && mode_env < 0 && g_force_nt_mode < 0);
  if (wide) { lc_h = 160; if (mode != 3 && mode != 5 && mode != 6) return set_error("gemm_nt: the 160 x 256 tile exists for the loader/consumer kernels only"); }
  if (mode != 8 && lc_h == 320) lc_h = 256;                  // a forced mode after the model picked the 8-phase tile
  if (mode == 8 && lc_h != 320) lc_h = 256;
  if (mode == 8) {
    if (!p_ok || (d.K % 64) || ((uintptr_t)d.bias % 16)) return set_error("gemm_nt: the 8-phase kernel needs K % 64 == 0, N % 8 == 0, ldc % 8 == 0, 16-byte aligned outputs / bias");
    if (lc_h != 256 && lc_h != 320) return set_error("gemm_nt: 8-phase tile height must be 256 or 320");
  } else if (lc_h == 64 && mode != 3) { lc_h = 160;        // the 64-row tile exists for the one-tile-per-workgroup kernel only
  } else if (lc_h != 256 && lc_h != 192 && lc_h != 160 && lc_h != 64) return set_error("gemm_nt: tile height must be 256, 192, 160 or 64");
  static const bool nt_log = W2VS_ENV_SET("W2VS_GEMM_LOG");      // shapes and the form chosen for them, one line per launch
  if (nt_log) fprintf(stderr, "gemm_nt M %d N %d K %d batch %d epi %d lda %ld -> mode %d tile %dx%d\n", d.M, d.N, d.K, (int)d.batch, d.epi,
                      (long)d.lda, mode, mode >= 3 ? lc_h : 128, (wide || mode == 8) ? 256 : 128);
  const dim3 grid8((d.N + (wide ? 255 : 127)) / (wide ? 256 : 128), (d.M + lc_h - 1) / lc_h, d.batch > 0 ? d.batch : 1);
  // structural zero block of B (w2vs_gemm_desc.zk_*): only the persistent kernels skip it, and only when their column tiles do
  // not straddle zk_col and the skipped K range is a whole, EVEN number of K tiles (the 8-phase loop walks K tiles in pairs)
  p.zk_col = 0; p.zk_kt = 0;
  if (d.zk_k > 0 && d.zk_col > 0 && (mode == 8 || mode == 5 || mode == 6)) {
    const int tw = (wide || mode == 8) ? 256 : 128;
    if (d.zk_k % (2 * BK) == 0 && d.zk_k < d.K && d.zk_col % tw == 0 && d.zk_col < d.N) { p.zk_col = d.zk_col; p.zk_kt = d.zk_k / BK; }
  }
#ifdef W2VS_ABLATION
  p.stamps = g_nt_stamps; p.epi_dbg = g_epi_dbg;
#endif
#ifdef W2VS_TUNING
#define W2VS_LONE4_LAUNCH(E) hipLaunchKernelGGL((gemm_nt_p_kernel<E, 1, 10, 4>), gp, dim3(512), 0, s, p, ntm_, ntn_, tot_)
#else
#define W2VS_LONE4_LAUNCH(E) (void)0
#endif
#define NT_LAUNCH(E)                                                                          \
  do {                                                                                        \
    if (mode == 8) {                /* 8-phase, persistent: (lc_h) x 256 tiles */              \
      const int ntm_ = (d.M + lc_h - 1) / lc_h, ntn_ = (d.N + 255) / 256;                     \
      const int tot_ = ntm_ * ntn_ * (d.batch > 0 ? d.batch : 1);                              \
      const dim3 gp(std::min(tot_, 256));                                                     \
      if (lc_h == 320) hipLaunchKernelGGL((gemm_nt8_kernel<E, 5>), gp, dim3(512), 0, s, p, ntm_, ntn_, tot_); \
      else hipLaunchKernelGGL((gemm_nt8_kernel<E, 4>), gp, dim3(512), 0, s, p, ntm_, ntn_, tot_);             \
    } else if (mode == 5 || mode == 6) {   /* 6: the same kernel with one workgroup per tile */ \
      const int ntm_ = grid8.y, ntn_ = grid8.x, tot_ = ntm_ * ntn_ * (int)grid8.z;            \
      const dim3 gp(mode == 6 ? tot_ : std::min(tot_, 256));                                  \
      if (lone4) { W2VS_LONE4_LAUNCH(E); }                                                    \
      else if (wide) hipLaunchKernelGGL((gemm_nt_p_kernel<E, 2, 5, 4>), gp, dim3(768), 0, s, p, ntm_, ntn_, tot_);     \
      else if (lc_h == 256) hipLaunchKernelGGL((gemm_nt_p_kernel<E, 4, 4>), gp, dim3(768), 0, s, p, ntm_, ntn_, tot_); \
      else if (lc_h == 192) hipLaunchKernelGGL((gemm_nt_p_kernel<E, 3, 4>), gp, dim3(640), 0, s, p, ntm_, ntn_, tot_); \
      else hipLaunchKernelGGL((gemm_nt_p_kernel<E, 2, 5>), gp, dim3(512), 0, s, p, ntm_, ntn_, tot_);                  \
    } else if (mode == 3 && wide) hipLaunchKernelGGL((gemm_nt_lc_kernel<E, 2, 5, 4>), grid8, dim3(768), 0, s, p); \
    else if (mode == 3 && lc_h == 256) hipLaunchKernelGGL((gemm_nt_lc_kernel<E, 4, 4>), grid8, dim3(768), 0, s, p); \
    else if (mode == 3 && lc_h == 192) hipLaunchKernelGGL((gemm_nt_lc_kernel<E, 3, 4>), grid8, dim3(640), 0, s, p); \
    else if (mode == 3 && lc_h == 64) hipLaunchKernelGGL((gemm_nt_lc_kernel<E, 1, 4>), grid8, dim3(384), 0, s, p); \
    else if (mode == 3) hipLaunchKernelGGL((gemm_nt_lc_kernel<E, 2, 5>), grid8, dim3(512), 0, s, p); \
    else if (mode == 2) hipLaunchKernelGGL((gemm_nt_kernel<E, 2>), grid, block, 0, s, p);     \
    else if (mode == 1) hipLaunchKernelGGL((gemm_nt_kernel<E, 1>), grid, block, 0, s, p);     \
    else hipLaunchKernelGGL((gemm_nt_kernel<E, 0>), grid, block, 0, s, p);                    \
  } while (0)
  switch (d.epi) {
    case EPI_NONE: NT_LAUNCH(EPI_NONE); break;
    case EPI_BIAS: NT_LAUNCH(EPI_BIAS); break;
    case EPI_BIAS_GELU: NT_LAUNCH(EPI_BIAS_GELU); break;
    case EPI_BIAS_GELU_SAVE: NT_LAUNCH(EPI_BIAS_GELU_SAVE); break;
    case EPI_DGELU: NT_LAUNCH(EPI_DGELU); break;
    case EPI_F32: NT_LAUNCH(EPI_F32); break;
    case EPI_ADD: NT_LAUNCH(EPI_ADD); break;
    case EPI_BIAS_GELU_SAVEG: NT_LAUNCH(EPI_BIAS_GELU_SAVEG); break;
    case EPI_MUL: NT_LAUNCH(EPI_MUL); break;
    default: return set_error("gemm_nt: unknown epilogue");
  }
#undef NT_LAUNCH
#undef W2VS_LONE4_LAUNCH
  // profiling id = one kernel symbol family: 16 * form + epilogue (form 0: gemm_nt_kernel, 1: gemm_nt_lc_kernel, 2: gemm_nt_p_kernel);
  // the weight-gradient kernels use 10..12 (form 0 epilogues stop at 8)
  const int form = mode == 8 ? 3 : (mode == 5 || mode == 6) ? 2 : (mode == 3 ? 1 : 0);
  prof_end(pe, 16 * form + d.epi, 2.0 * d.M * d.N * d.K * (d.batch > 0 ? d.batch : 1), s);
  return hip_check(hipGetLastError(), "gemm_nt launch");
}

int gemm_tn(const GemmDesc& d, int num_cu_hint, hipStream_t s) {
  if (int e = check_common(d)) return e;
  if (!d.Cf) return set_error("gemm_tn: needs an fp32 accumulation target");
  if ((d.M % 8) || (d.N % 8)) return set_error("gemm_tn: M and N must be multiples of 8");
  if (d.overwrite) {       // the general kernels accumulate (atomics / slab sums): an overwriting call clears its target first
    if (d.ldc != d.N || d.sC != 0) return set_error("gemm_tn: overwrite needs a contiguous, unbatched target (ldc == N)");
    if (hipMemsetAsync(d.Cf, 0, (size_t)d.M * d.N * sizeof(float), s) != hipSuccess) return set_error("gemm_tn: memset failed");
  }
  GemmP p{};
  p.A = (const bf16*)d.A; p.B = (const bf16*)d.B; p.Cf = d.Cf;
  p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.a_off = d.a_off; p.alpha = d.alpha;
  p.colsum = d.colsum;
  long a_ext = d.a_bytes ? d.a_bytes : ((long)d.a_off + (long)(d.K - 1) * d.lda + d.M) * 2;
  long b_ext = d.b_bytes ? d.b_bytes : ((long)(d.K - 1) * d.ldb + d.N) * 2;
  if (a_ext <= 0 || a_ext >= 0x7FFFFFF0L || b_ext >= 0x7FFFFFF0L) return set_error("gemm_tn: operand extent must be < 2 GiB per batch");
  p.a_bytes = (uint32_t)a_ext; p.b_bytes = (uint32_t)b_ext;
  const int nb = d.batch > 0 ? d.batch : 1;
  p.sA = d.sA; p.sB = d.sB; p.sC = d.sC;   // sC: batch stride of Cf (0 = every batch adds into the same matrix)
  // loader/consumer form (256x128 tiles, one workgroup per CU) when its grid can fill the chip with at most ~1
  // workgroup per CU and each keeps a long K loop; small outputs (out_proj) stay on the 128x128 kernel
  static const int tn_lc_env = W2VS_ENV_INT("W2VS_TN_LC", -1);
  {
    const int tiles8 = ((d.N + 127) / 128) * ((d.M + 255) / 256) * nb;
    const int ncu = num_cu_hint > 0 ? num_cu_hint : 256;
    bool use_lc = tiles8 >= 16 && tiles8 <= ncu && d.sC == 0;
    if (tn_lc_env >= 0) use_lc = tn_lc_env != 0;
    if (g_force_tn_lc >= 0) use_lc = g_force_tn_lc != 0 && d.sC == 0;
    static const bool tn_log = W2VS_ENV_SET("W2VS_GEMM_LOG");
    if (tn_log) fprintf(stderr, "gemm_tn M %d N %d K %d batch %d -> %s (%d tiles of 256x128)\n", d.M, d.N, d.K, nb, use_lc ? "loader/consumer" : "128^2 atomics", tiles8);
    if (use_lc) {
      int splits = std::max(1, ncu / tiles8);
      const int max_splits = (d.K + TK - 1) / TK;
      splits = std::min(splits, max_splits);
      int ks = (d.K + splits - 1) / splits;
      ks = ((ks + TK - 1) / TK) * TK;
      splits = (d.K + ks - 1) / ks;
      p.k_split = ks; p.n_split = splits;
      dim3 grid((d.N + 127) / 128, (d.M + 255) / 256, splits * nb);
      const int parts = splits * nb;
      const bool slab = d.ws && d.ws_bytes >= (int64_t)parts * d.M * d.N * 4 && parts > 1 && (d.ldc % 4) == 0 &&
                        ((uintptr_t)d.ws % 16) == 0 && ((uintptr_t)d.Cf % 16) == 0;
      hipEvent_t pe = prof_begin(s);
      if (slab) {
        p.slab = (float*)d.ws;
        hipLaunchKernelGGL(gemm_tn_lc_kernel<true>, grid, dim3(768), 0, s, p);
        const long total4 = (long)d.M * d.N / 4;
        hipLaunchKernelGGL(tn_slab_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, p.slab, parts, d.M,
                           d.N, (long)d.ldc, d.Cf);
      } else {
        hipLaunchKernelGGL(gemm_tn_lc_kernel<false>, grid, dim3(768), 0, s, p);
      }
      prof_end(pe, 11, 2.0 * d.M * d.N * d.K * nb, s);  // id 11: loader/consumer form (+ its summing launch); NT launches use ids 0..8 = their epilogue
      return hip_check(hipGetLastError(), "gemm_tn launch");
    }
  }
  int tiles = ((d.N + BN - 1) / BN) * ((d.M + BM - 1) / BM) * nb;
  static const int tn_target = W2VS_ENV_INT("W2VS_TN_TARGET", 0);
  int target = tn_target > 0 ? tn_target : (num_cu_hint > 0 ? num_cu_hint : 256) * 3 / 2;  // 1.5 blocks per CU: measured best trade between fill and atomic traffic
  int splits = (target + tiles - 1) / tiles;
  int max_splits = (d.K + TK - 1) / TK;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int ks = (d.K + splits - 1) / splits;
  ks = ((ks + TK - 1) / TK) * TK;
  splits = (d.K + ks - 1) / ks;
  p.k_split = ks; p.n_split = splits;
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, splits * nb), block(256);
  hipEvent_t pe = prof_begin(s);
  hipLaunchKernelGGL(gemm_tn_kernel, grid, block, 0, s, p);
  prof_end(pe, 10, 2.0 * d.M * d.N * d.K * nb, s);  // id 10: 128x128 atomics form
  return hip_check(hipGetLastError(), "gemm_tn launch");
}

// What the device can hold, asked once per device: CU count and how many gemm_tn8_group_kernel workgroups can be resident
// at once.  The S = 2 form of that kernel makes the two workgroups of a pair wait on each other's flag: it is only launched
// when EVERY workgroup of the grid fits on the chip together (and never while something else - the gradient all-reduce - may
// hold CUs: gemm_tn8_max_split, set by the training step when an exchange is active).
struct DevFacts { int cus = 0; int tn8_resident = 0; unsigned* flags = nullptr; int next = 0; };
static DevFacts* dev_facts() {
  static DevFacts facts[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  DevFacts& f = facts[dev];
  if (f.cus == 0) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return nullptr;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_tn8_group_kernel, 512, 0) != hipSuccess) per_cu = 0;
    f.tn8_resident = per_cu * pr.multiProcessorCount;
    f.cus = pr.multiProcessorCount;
  }
  return &f;
}
static int g_tn8_max_split = 2;
static int g_last_group_form = 0;      // 0: one launch per problem, 12: 256x128 single-writer group, 13: 8-phase S = 1, 14: 8-phase S = 2
void gemm_tn8_max_split(int s) { g_tn8_max_split = s < 1 ? 1 : (s > 2 ? 2 : s); }
int gemm_last_group_form() { return g_last_group_form; }

// n <= 4 weight-gradient GEMMs in one launch (see gemm_tn_group_kernel).  Falls back to one gemm_tn per problem when the
// group does not qualify (alignment, batching, more tiles than CUs, tiny K).
int gemm_tn_group(const GemmDesc* ds, int n, int num_cu_hint, hipStream_t s) {
  if (!ds || n < 1 || n > TN8_MAXP) return set_error("gemm_tn_group: 1..12 problems");
  DevFacts* df = dev_facts();
  if (!df) return set_error("gemm_tn_group: cannot query the device");
  // the caller's hint can only LOWER the count (tests, a partitioned device); the real CU count caps it
  const int ncu = std::min(num_cu_hint > 0 ? num_cu_hint : df->cus, df->cus);
  g_last_group_form = 0;
  if (n > 4) {
    // more than four problems (two layers' weight gradients): only the 8-phase kernel takes them as ONE launch, and only when
    // their 256^2 tiles fit one workgroup per CU WITHOUT a K split; otherwise two groups, as before
    static const int tn8_env2 = W2VS_ENV_INT("W2VS_TN8", 1);
    int t8 = 0;
    bool ok8 = tn8_env2 != 0;
    for (int i = 0; i < n; ++i) {
      const GemmDesc& d = ds[i];
      t8 += ((d.N + 255) / 256) * ((d.M + 255) / 256);
      ok8 = ok8 && d.Cf && (d.M % 8) == 0 && (d.N % 8) == 0 && (d.batch <= 1) && d.sC == 0 && (d.ldc % 4) == 0 &&
            ((uintptr_t)d.Cf % 16) == 0 && d.K >= 8 * TK;
    }
    if (!ok8 || t8 > ncu || t8 * 8 < ncu * 5) {
      for (int o = 0; o < n; o += 4)
        if (int e = gemm_tn_group(ds + o, std::min(4, n - o), num_cu_hint, s)) return e;
      return 0;
    }
  }
  TnGroupP g{};
  GemmP gp[TN8_MAXP] = {};
  int tiles = 0;
  double flops = 0;
  bool ok = true;
  for (int i = 0; i < n; ++i) {
    const GemmDesc& d = ds[i];
    if (int e = check_common(d)) return e;
    if (!d.Cf) return set_error("gemm_tn: needs an fp32 accumulation target");
    if ((d.M % 8) || (d.N % 8)) return set_error("gemm_tn: M and N must be multiples of 8");
    GemmP& p = gp[i];
    p.A = (const bf16*)d.A; p.B = (const bf16*)d.B; p.Cf = d.Cf;
    p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.a_off = d.a_off; p.alpha = d.alpha;
    p.colsum = d.colsum; p.overwrite = d.overwrite;      // the single-writer group kernels honour it; the fallbacks memset (gemm_tn)
    const long a_ext = d.a_bytes ? d.a_bytes : ((long)d.a_off + (long)(d.K - 1) * d.lda + d.M) * 2;
    const long b_ext = d.b_bytes ? d.b_bytes : ((long)(d.K - 1) * d.ldb + d.N) * 2;
    if (a_ext <= 0 || a_ext >= 0x7FFFFFF0L || b_ext >= 0x7FFFFFF0L) return set_error("gemm_tn: operand extent must be < 2 GiB per batch");
    p.a_bytes = (uint32_t)a_ext; p.b_bytes = (uint32_t)b_ext;
    p.k_split = ((d.K + TK - 1) / TK) * TK; p.n_split = 1;
    if (i < 4) { g.p[i] = p; g.first[i] = tiles; }      // the single-writer kernel's table holds four (a larger group never reaches it)
    tiles += ((d.N + 127) / 128) * ((d.M + 255) / 256);
    flops += 2.0 * d.M * d.N * d.K;
    ok = ok && (d.batch <= 1) && d.sC == 0 && (d.ldc % 4) == 0 && ((uintptr_t)d.Cf % 16) == 0 && d.K >= 8 * TK;
  }
  for (int i = std::min(n, 4); i <= 4; ++i) g.first[i] = tiles;
  // Round 3: 256 x 256 tiles + split K over workgroup pairs (gemm_tn8_group_kernel) when that fills the chip better
  static const int tn8_env = W2VS_ENV_INT("W2VS_TN8", 1);
  if (ok && tn8_env) {
    Tn8GroupP g8{};
    int t8 = 0;
    for (int i = 0; i < n; ++i) { g8.p[i] = gp[i]; g8.first[i] = t8; t8 += ((ds[i].N + 255) / 256) * ((ds[i].M + 255) / 256); }
    for (int i = n; i <= TN8_MAXP; ++i) g8.first[i] = t8;
    int minK = ds[0].K;
    for (int i = 1; i < n; ++i) minK = std::min(minK, ds[i].K);
    int S = std::max(1, std::min(g_tn8_max_split, ncu / std::max(1, t8)));
    // a pair spins on its partner: S = 2 needs the WHOLE grid co-resident (occupancy x CUs of THIS device, not a constant)
    while (S > 1 && (t8 * S > std::min(ncu, df->tn8_resident) || minK < 8 * TK * S || !ds[0].ws ||
                     ds[0].ws_bytes < (int64_t)t8 * (S - 1) * 65536 * 4 || ((uintptr_t)ds[0].ws % 16))) --S;
    // worth it when the 256^2 grid keeps at least as many CUs busy as the 256 x 128 grid would (a base layer: 108 tiles x 2)
    if (t8 * S <= ncu && t8 * S * 8 >= ncu * 5 && t8 <= 1024) {
      // 16 regions x 1024 tiles x 2 exchange flags, used round-robin (launches in flight on different streams do not share
      // one); zero between launches: whoever polls a flag lowers it again
      // one pool PER DEVICE, zeroed synchronously when it is made (visible to every stream that launches later)
      if (!df->flags) {
        if (hipMalloc((void**)&df->flags, 16 * 2048 * sizeof(unsigned)) != hipSuccess) return set_error("gemm_tn_group: flag allocation failed");
        if (hipMemset(df->flags, 0, 16 * 2048 * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
          return set_error("gemm_tn_group: flag memset failed");
      }
      static const int s_env = W2VS_ENV_INT("W2VS_TN8_S", 0);
      if (s_env > 0) S = std::min(S, s_env);
#ifdef W2VS_ABLATION      // timing-only ablations (results are WRONG by design): never in the product build
      static const int dbg_env = W2VS_ENV_INT("W2VS_TN8_DBG", 0);
      g8.dbg = dbg_env;
#endif
      g8.S = S; g8.slab = (float*)ds[0].ws; g8.slab_bytes = std::min<int64_t>(ds[0].ws_bytes, 0x7FFFFFF0L);
      g8.cnt = df->flags + 2048 * (df->next++ & 15);
      g_last_group_form = S == 2 ? 14 : 13;
      static const bool tn_log8 = W2VS_ENV_SET("W2VS_GEMM_LOG");
      if (tn_log8) fprintf(stderr, "gemm_tn_group -> 8-phase, %d tiles of 256x256, split K %d\n", t8, S);
      hipEvent_t pe = prof_begin(s, 5);
      hipLaunchKernelGGL(gemm_tn8_group_kernel, dim3(t8 * S), dim3(512), 0, s, g8);
      prof_end(pe, 13, flops, s);                  // id 13: the 8-phase grouped weight-gradient launch
      return hip_check(hipGetLastError(), "gemm_tn8_group launch");
    }
  }
  if (n > 4) {                                     // (the pre-check above makes this unreachable; never run the 4-entry kernel on more)
    for (int o = 0; o < n; o += 4)
      if (int e = gemm_tn_group(ds + o, std::min(4, n - o), num_cu_hint, s)) return e;
    return 0;
  }
  static const int grp_env = W2VS_ENV_INT("W2VS_TN_GROUP", 1);
  // grouped only when it fills >= 3/4 of the chip AND leaves some slack: with exactly one workgroup per CU a single CU that is
  // not free at dispatch costs a whole extra round of full-length K loops (measured on the large model's 128 + 128 tiles)
  if (!ok || tiles > ncu - ncu / 16 || tiles * 4 < ncu * 3 || !grp_env) {
    for (int i = 0; i < n; ++i)
      if (int e = gemm_tn(ds[i], num_cu_hint, s)) return e;
    return 0;
  }
  g_last_group_form = 12;
  hipEvent_t pe = prof_begin(s, 5);                // ~10 launches per step: every 5th is timed (the global 1-in-29 would see 7 in a run)
  hipLaunchKernelGGL(gemm_tn_group_kernel, dim3(tiles), dim3(768), 0, s, g);
  prof_end(pe, 12, flops, s);                      // id 12: the grouped weight-gradient launch
  return hip_check(hipGetLastError(), "gemm_tn_group launch");
}

}  // namespace w2vs
