// Input side of the hot path.
//   batch_by_size : host restatement of the reference's Cython batcher (fs/data/data_utils_fast.pyx:19-98)
//   collate       : RawAudioDataset.collater (fs/data/audio/raw_audio_dataset.py:123-192) + per-utterance
//                   normalisation (:60-72) as two HBM-bound kernels (statistics partials, then crop / pad / cast)
#include "common.h"
#include "w2vs_internal.h"

namespace w2vs {

// ------------------------------------------------------------------------------------------------ batcher (host)
// State: the last complete batch [batch_start, ends[count]) and a tail [ends[count], pos].  A sample joins the tail;
// batch + tail is committed whenever its size satisfies bsz_mult; on overflow of max_tokens / max_sentences the running
// batch is finalised and the tail starts the next one (and if the tail alone overflows, it is finalised without the
// current sample, which becomes the new tail).  Cost of a batch = sentences x longest sample (padded size).
int batch_by_size(const int64_t* num_tokens, int64_t n, int64_t max_tokens, int64_t max_sentences, int32_t bsz_mult,
                  int32_t* ends, int32_t* n_batches) {
  if (!n_batches || (n > 0 && (!num_tokens || !ends))) return set_error("batch_by_size: null pointer");
  if (n < 0 || n > 0x7fffffff) return set_error("batch_by_size: n out of range");
  if (bsz_mult < 1) return set_error("batch_by_size: bsz_mult must be >= 1");
  *n_batches = 0;
  if (n == 0) return 0;
  if (max_tokens > 0)
    for (int64_t i = 0; i < n; ++i)
      if (num_tokens[i] > max_tokens) return set_error("batch_by_size: a sample is longer than max_tokens");
  for (int64_t i = 0; i < n; ++i) ends[i] = 0;
  int32_t count = 0, batch_start = 0;
  int64_t tail_max = 0, batch_max = 0;
  for (int32_t pos = 0; pos < (int32_t)n; ++pos) {
    tail_max = std::max(tail_max, num_tokens[pos]);
    const int32_t new_end = pos + 1;
    int64_t new_max = std::max(batch_max, tail_max);
    const int32_t sentences = new_end - batch_start;
    const int64_t tokens = (int64_t)sentences * new_max;
    const bool overflow = (max_sentences > 0 && sentences > max_sentences) || (max_tokens > 0 && tokens > max_tokens);
    const bool mult_ok = sentences < bsz_mult || sentences % bsz_mult == 0;
    if (overflow) {
      const int64_t tail_tokens = tail_max * (int64_t)(new_end - ends[count]);
      if (max_tokens > 0 && tail_tokens > max_tokens) {   // the tail alone does not fit: close it before this sample
        ++count;
        ends[count] = pos;
        tail_max = num_tokens[pos];
      }
      batch_start = ends[count];
      ++count;
      new_max = tail_max;
    }
    if (overflow || mult_ok) {
      ends[count] = new_end;
      batch_max = new_max;
      tail_max = 0;
    }
  }
  if (ends[count] != (int32_t)n) ++count;
  // the reference returns np.split(indices, ends[:count]): count split points = count + 1 batches, the last one up to n
  ends[count] = (int32_t)n;
  *n_batches = count + 1;
  return 0;
}

// ------------------------------------------------------------------------------------------------ collate (device)
constexpr int CH = 8192;          // samples per statistics chunk (one 256-thread block, 32 per thread)

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// partial[(b * nch + c) * 2 + {0,1}] = sum x, sum x^2 over chunk c of utterance b (fp64: exact enough that the
// result does not depend on the chunking)
__global__ __launch_bounds__(256) void collate_stats_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offset,
                                                            const int32_t* __restrict__ size, double* __restrict__ partial,
                                                            int nch) {
  __shared__ double sh[4];
  const int b = blockIdx.y, c = blockIdx.x;
  const int n = size[b];
  const float* x = flat + offset[b];
  const int lo = c * CH, hi = min(lo + CH, n);
  double s = 0, q = 0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = (double)x[i];
    s += v;
    q += v * v;
  }
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  if (threadIdx.x == 0) {
    partial[((long)b * nch + c) * 2] = s;
    partial[((long)b * nch + c) * 2 + 1] = q;
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void collate_rows_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offset,
                                                           const int32_t* __restrict__ size, const int32_t* __restrict__ crop,
                                                           const double* __restrict__ partial, void* __restrict__ out,
                                                           uint8_t* __restrict__ pmask, int target, int width, int nch, int normalize) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  const int n = size[b];
  float mean = 0.f, rstd = 1.f;
  if (normalize) {
    double s = 0, q = 0;
    const int used = (n + CH - 1) / CH;
    for (int c = threadIdx.x; c < used; c += 256) {
      s += partial[((long)b * nch + c) * 2];
      q += partial[((long)b * nch + c) * 2 + 1];
    }
    s = block_sum(s, sh);
    q = block_sum(q, sh);
    const double m = n > 0 ? s / n : 0.0;
    const double var = n > 0 ? fmax(q / n - m * m, 0.0) : 0.0;     // biased variance, as layer_norm
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + 1e-5));
  }
  const int start = n > target ? crop[b] : 0;
  const float* x = flat + offset[b] + start;
  const int valid = min(n, target);
  const int base = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (base >= width) return;
  float v[8];
  uint8_t pm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = base + e;
    const bool in = i < valid;
    v[e] = in ? (x[i] - mean) * rstd : 0.f;
    pm[e] = in ? 0 : 1;
  }
  const long o = (long)b * width + base;
  const int cnt = min(8, width - base);
  if (F32) {
    float* dst = (float*)out + o;
    for (int e = 0; e < cnt; ++e) dst[e] = v[e];
  } else {
    bf16* dst = (bf16*)out + o;
    for (int e = 0; e < cnt; ++e) dst[e] = f2bf(v[e]);
  }
  if (pmask)
    for (int e = 0; e < cnt; ++e) pmask[o + e] = pm[e];
}

int collate_chunks(int max_size) { return max_size <= 0 ? 1 : (max_size + CH - 1) / CH; }

int collate(const w2vs_collate_desc& d, hipStream_t st) {
  if (d.B <= 0 || d.target <= 0 || d.width < d.target) return set_error("collate: need B > 0 and 0 < target <= width");
  if (!d.flat || !d.offset || !d.size || !d.crop_start || !d.out) return set_error("collate: null pointer");
  if (d.normalize && (!d.partial || d.max_size <= 0)) return set_error("collate: normalize needs partial scratch and max_size");
  const int nch = collate_chunks(d.max_size);
  if (d.normalize) {
    hipLaunchKernelGGL(collate_stats_kernel, dim3(nch, d.B), dim3(256), 0, st, d.flat, d.offset, d.size, d.partial, nch);
  }
  const dim3 grid((d.width + 2047) / 2048, d.B);
  if (d.out_f32)
    hipLaunchKernelGGL(collate_rows_kernel<true>, grid, dim3(256), 0, st, d.flat, d.offset, d.size, d.crop_start, d.partial,
                       d.out, d.padding_mask, d.target, d.width, nch, d.normalize);
  else
    hipLaunchKernelGGL(collate_rows_kernel<false>, grid, dim3(256), 0, st, d.flat, d.offset, d.size, d.crop_start, d.partial,
                       d.out, d.padding_mask, d.target, d.width, nch, d.normalize);
  return hip_check(hipGetLastError(), "collate");
}

}  // namespace w2vs
