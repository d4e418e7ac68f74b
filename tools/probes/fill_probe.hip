// L2 -> LDS fill rate per CU as a function of the bytes in flight: every workgroup (8 waves) streams 16 KiB chunks of an
// L2-resident region into a ring of LDS slots with LDS-DMA (buffer_load_dwordx4 ... lds), D chunks in flight, optionally one
// workgroup barrier per chunk (as a GEMM main loop has).  No MFMA, no LDS reads: the ceiling of the staging path alone.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/fill_probe.hip -o tools/probes/fill_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

// D = chunks in flight (ring of D + 1 slots); BAR = one s_barrier per chunk
template <int D, int BAR, int NW = 8>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const char* src, uint32_t region_bytes, int chunks, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups of one XCD (ids equal mod 8) share one region: L2 hits after the first pass
  const char* base = src + (size_t)(blockIdx.x & 7) * region_bytes;
  __amdgpu_buffer_rsrc_t r = make_rsrc(base, region_bytes);
  const uint32_t wg_off = (uint32_t)(blockIdx.x >> 3) * 65536u;     // neighbours start 64 KiB apart: overlapping panels
  constexpr int PER = 16 / NW;                                        // 1 KiB pieces per wave and chunk
  const uint32_t lane_off = (uint32_t)wid * (PER * 1024u) + (uint32_t)lane * 16u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  int slot = 0;
  for (int c = 0; c < chunks; ++c) {
    const uint32_t off = (wg_off + (uint32_t)c * 16384u + lane_off) % region_bytes;
    char* dst = lds + slot * 16384 + wid * (PER * 1024);
#pragma unroll
    for (int q = 0; q < PER; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(dst + q * 1024), 16, (off + q * 1024u) % region_bytes & ~15u, 0, 0, 0);
    wait_vm<PER * (D - 1)>();        // at most D - 1 older chunks still in flight behind this one... i.e. D in flight
    if (BAR) __builtin_amdgcn_s_barrier();
    slot = slot == D ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int D, int BAR, int NW = 8>
static int launch(const void* src, uint32_t region_bytes, int chunks, unsigned long long* cycles, int grid, hipStream_t s) {
  const size_t lds = (size_t)(D + 1) * 16384;
  if (hipFuncSetAttribute((const void*)fill_kernel<D, BAR, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  hipLaunchKernelGGL((fill_kernel<D, BAR, NW>), dim3(grid), dim3(NW * 64), lds, s, (const char*)src, region_bytes, chunks, cycles);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
// depth 3 (48 KiB in flight), no barrier, NW issuing waves
extern "C" int fill_probe_waves(const void* src, uint32_t region_bytes, int chunks, int waves, int bar, unsigned long long* cycles, int grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (waves * 2 + (bar ? 1 : 0)) {
    case 2: return launch<3, 0, 1>(src, region_bytes, chunks, cycles, grid, s);
    case 4: return launch<3, 0, 2>(src, region_bytes, chunks, cycles, grid, s);
    case 8: return launch<3, 0, 4>(src, region_bytes, chunks, cycles, grid, s);
    case 16: return launch<3, 0, 8>(src, region_bytes, chunks, cycles, grid, s);
    case 3: return launch<3, 1, 1>(src, region_bytes, chunks, cycles, grid, s);
    case 5: return launch<3, 1, 2>(src, region_bytes, chunks, cycles, grid, s);
    case 9: return launch<3, 1, 4>(src, region_bytes, chunks, cycles, grid, s);
    case 17: return launch<3, 1, 8>(src, region_bytes, chunks, cycles, grid, s);
  }
  return 3;
}

extern "C" int fill_probe(const void* src, uint32_t region_bytes, int chunks, int depth, int bar, unsigned long long* cycles, int grid,
                          void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define CASE(DD) case DD: return bar ? launch<DD, 1>(src, region_bytes, chunks, cycles, grid, s) : launch<DD, 0>(src, region_bytes, chunks, cycles, grid, s);
  switch (depth) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(8) CASE(9)
    default: return 3;
  }
#undef CASE
}

// ---- LDS read rate per CU: NW waves of a workgroup read the same 64 KiB of LDS over and over, 8 independent reads in flight per
// wave.  MODE 0: ds_read_b128 (lane-linear, conflict-free), 1: ds_read_b64, 2: ds_read_b64_tr_b16, 3: ds_read_b128 through the GEMM
// kernels' XOR swizzle (row = lane & 15 of a 128-byte-row tile, chunk (lane >> 4) ^ ((row >> 1) & 7)).
template <int MODE>
__global__ __launch_bounds__(512) void lds_read_kernel(int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 16384; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t base;
  if (MODE == 0) base = (uint32_t)(wid * 8192 + lane * 16);
  else if (MODE == 1 || MODE == 2) base = (uint32_t)(wid * 8192 + lane * 8);
  else { const int row = lane & 15, ch = (lane >> 4) ^ ((row >> 1) & 7); base = (uint32_t)(wid * 8192 + row * 128 + ch * 16); }
  base += (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {
      u4 r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                   "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(base) : "memory");
      acc ^= r0[0] ^ r1[1] ^ r2[2] ^ r3[3] ^ r4[0] ^ r5[1] ^ r6[2] ^ r7[3];
    } else {
      u2 r0, r1, r2, r3, r4, r5, r6, r7;
      if (MODE == 1)
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                     "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(base) : "memory");
      else
        asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:512\n ds_read_b64_tr_b16 %2, %8 offset:1024\n ds_read_b64_tr_b16 %3, %8 offset:1536\n"
                     "ds_read_b64_tr_b16 %4, %8 offset:2048\n ds_read_b64_tr_b16 %5, %8 offset:2560\n ds_read_b64_tr_b16 %6, %8 offset:3072\n ds_read_b64_tr_b16 %7, %8 offset:3584\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(base) : "memory");
      acc ^= r0[0] ^ r1[1] ^ r2[0] ^ r3[1] ^ r4[0] ^ r5[1] ^ r6[0] ^ r7[1];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int lds_read_probe(int mode, int waves, int iters, unsigned* sink, int grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = 65536;
#define LCASE(M) case M: hipFuncSetAttribute((const void*)lds_read_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((lds_read_kernel<M>), dim3(grid), dim3(waves * 64), lds, s, iters, sink); break;
  switch (mode) { LCASE(0) LCASE(1) LCASE(2) LCASE(3) default: return 3; }
#undef LCASE
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
