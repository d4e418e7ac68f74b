/* w2vs_rnnt.h - transducer-loss entry points of libw2vs.so.
 *
 * The reference ships this loss as native code with a C interface of its own:
 * /root/reference/warp_transducer/include/rnnt.h:13-168 (implemented in src/rnnt_entrypoint.cu and
 * src/attent_entrypoint.cu, bound for PyTorch in pytorch_binding/src/binding.cpp:100-210).  libw2vs.so exports THE
 * SAME SYMBOLS with the same signatures and argument meaning, so the reference's binding links against it unchanged;
 * the declarations below restate that interface for gfx950 (the CUDA stream type becomes an opaque stream pointer
 * that receives a hipStream_t).
 *
 * Differences a caller can observe:
 *  - loc = RNNT_CPU is rejected with RNNT_STATUS_EXECUTION_FAILED: this build has no CPU path.
 *  - workspace sizes differ from the reference's (the lattice is kept anti-diagonal-major); always ask
 *    get_workspace_size / get_delay_workspace_size of THIS library.
 *  - flat_labels / label_lengths / input_lengths / delay_values are DEVICE pointers for loc = RNNT_GPU (the reference's
 *    kernels read them on the device too, gpu_rnnt_kernel.h:17-19; its header comment "always in CPU memory"
 *    describes the CPU path).  costs is HOST memory; the call synchronises the stream before returning, as the
 *    reference does (gpu_rnnt.h:208-210, delay_transducer.h:366-368).
 *  - w2vs_rnnt_forward_async / _backward_async are additions: costs stay on the device and nothing synchronises.
 */
#ifndef W2VS_RNNT_H
#define W2VS_RNNT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#else
#include <stdbool.h>
#endif

typedef struct CUstream_st* CUstream;      /* receives a hipStream_t */

typedef enum {
  RNNT_STATUS_SUCCESS = 0,
  RNNT_STATUS_MEMOPS_FAILED = 1,
  RNNT_STATUS_INVALID_VALUE = 2,
  RNNT_STATUS_EXECUTION_FAILED = 3,
  RNNT_STATUS_UNKNOWN_ERROR = 4
} rnntStatus_t;

typedef enum { RNNT_CPU = 0, RNNT_GPU = 1 } rnntComputeLocation;

/* rnnt.h:44-66 - same field order and types (the struct is passed BY VALUE) */
struct rnntOptions {
  rnntComputeLocation loc;     /* must be RNNT_GPU */
  unsigned int num_threads;    /* ignored */
  CUstream stream;             /* hipStream_t the kernels are enqueued on */
  int blank_label;
  int maxT;                    /* activations are [B, maxT, maxU, V] */
  int maxU;                    /* = longest label sequence + 1 */
  bool batch_first;            /* ignored: the GPU layout is always batch first */
};
#ifndef __cplusplus
typedef struct rnntOptions rnntOptions;
#endif

int get_warprnnt_version(void);                              /* rnnt.h:25 -> 1 */
const char* rnntGetStatusString(rnntStatus_t status);        /* rnnt.h:31 */

/* rnnt.h:108-117.  activations: raw (un-normalised) network outputs [B, maxT, maxU, V] fp32; the log-softmax is fused.
 * gradients: NULL, or [B, maxT, maxU, V] fp32 receiving d(cost)/d(activations) (zero outside each sample's T x U).
 * flat_labels [B, maxU-1], label_lengths [B], input_lengths [B]: int32.  costs [B] (host): negative log-likelihoods. */
rnntStatus_t compute_rnnt_loss(const float* const activations, float* gradients, const int* const flat_labels,
                               const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                               int minibatch, float* costs, void* workspace, struct rnntOptions options);

/* rnnt.h:115-124 (called for double tensors by pytorch_binding/src/binding.cpp:69, :141).  Same contract as
 * compute_rnnt_loss with fp64 activations / gradients (device) and fp64 costs (host).  A CONVERTING WRAPPER: the
 * lattice runs in fp32, so results carry fp32 ACCURACY - the reference instantiates GpuRNNT<double> here, which exists
 * for its gradient checks: a finite-difference check through this entry needs fp32-sized steps and tolerances.  With a
 * gradient buffer nothing is allocated (that buffer doubles as the fp32 staging area and is widened in place); a
 * costs-only call (gradients == NULL) allocates the narrowed activations for the duration of the call. */
rnntStatus_t compute_rnnt_loss_fp64(const double* const activations, double* gradients, const int* const flat_labels,
                                    const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                                    int minibatch, double* costs, void* workspace, struct rnntOptions options);

/* rnnt.h:144-148 (dtype_size 4, or 8 for the fp64 entry; gpu must be true) */
rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, bool gpu, size_t* size_bytes
#ifdef __cplusplus
                                , size_t dtype_size = sizeof(float)
#else
                                , size_t dtype_size
#endif
);

/* rnnt.h:150-162 - the delay transducer (src/attent_entrypoint.cu:11-60, include/detail/delay_transducer.h).
 * delay_values [B, maxT, maxU] fp32: cost of emitting label u at frame t.  costs [3, B] (host): NLL, expected delay,
 * NLL + delay_scale * expected delay.  smooth = the Python front end's "temperature" (exponent on the occupancy terms
 * of the likelihood gradient, gpu_rnnt_kernel.h:399-423). */
rnntStatus_t compute_rnnt_delay_loss(const float* const activations, float* gradients, const int* const flat_labels,
                                     const int* const label_lengths, const int* const input_lengths,
                                     const float* delay_values, int alphabet_size, int minibatch, float* costs,
                                     void* workspace, float delay_scale, float smooth, struct rnntOptions options);

rnntStatus_t get_delay_workspace_size(int maxT, int maxU, int minibatch, bool gpu, size_t* size_bytes
#ifdef __cplusplus
                                      , size_t dtype_size = sizeof(float)
#else
                                      , size_t dtype_size
#endif
);

/* ---- additions (no counterpart in the reference) --------------------------------------------------------------- */
/* The same computation split the way an autograd node needs it, with no host round trip and no synchronisation.
 * forward: log-softmax denominators + both lattice recursions; costs_dev [3, B] on the DEVICE (NLL, expected delay,
 * NLL + delay_scale * expected delay; delay_values may be NULL = plain RNN-T, rows 1 = 0 and 2 = row 0).
 * backward: the gradient rows from the workspace the forward call filled (same arguments), multiplied in the same pass
 * by d(loss)/d(cost): grad_scale_host * grad_scale_dev[0] (grad_scale_n = 1), * grad_scale_dev[b] (= minibatch) or 1
 * (= 0) - the reference multiplies the finished gradient tensor once more in Python (delay_transducer.py:86-90).
 * flags bit 0: read the emission cost in the gradient as delay_values[b, t, u] instead of the reference's
 * delay_values[b * maxT + t] (gpu_rnnt_kernel.h:409 indexes the B x T x U array with a B x T index; clear = reproduce).
 * flags bit 1: write the gradients as bf16 (what the GEMMs of the output projection's backward consume).
 * cell_index / n_cells (both calls): NULL / 0 = dense activations [B, maxT, maxU, V].  Otherwise activations and gradients
 * hold ONLY the n_cells lattice cells cell_index[i] = (b * maxT + t) * maxU + u (device, int32; every cell with t < T_b and
 * u <= U_b must be listed): a ragged batch then costs the projection GEMMs and these kernels nothing for its padding. */
rnntStatus_t w2vs_rnnt_forward_async(const float* activations, const int* flat_labels, const int* label_lengths,
                                     const int* input_lengths, const float* delay_values, int alphabet_size, int minibatch,
                                     float* costs_dev, void* workspace, float delay_scale, struct rnntOptions options,
                                     const int* cell_index, int64_t n_cells);
rnntStatus_t w2vs_rnnt_backward_async(const float* activations, void* gradients, const int* flat_labels,
                                      const int* label_lengths, const int* input_lengths, const float* delay_values,
                                      int alphabet_size, int minibatch, void* workspace, float delay_scale, float smooth,
                                      int flags, const float* grad_scale_dev, int grad_scale_n, float grad_scale_host,
                                      struct rnntOptions options, const int* cell_index, int64_t n_cells);
/* Label-smoothed cross-entropy rows: fairseq's label_smoothed_nll_loss (fs/criterions/label_smoothed_cross_entropy.py:33-50)
 * on log_softmax(logits [rows, V] fp32), summed over rows whose target != pad, as TransducerOut.cross_entropy uses it
 * (rain/layers/attention_transducer.py:339-360).  sums2[0] += loss, sums2[1] += nll (device, caller zeroes); grads: NULL
 * or [rows, V] (fp32, or bf16 with grads_bf16) = grad_scale * d loss / d logits. */
rnntStatus_t w2vs_ls_ce_rows(const float* logits, const int* target, void* grads, float* sums2, int64_t rows, int V, int pad,
                             float epsilon, float grad_scale, int grads_bf16, void* stream);
/* delay_values builders of pytorch_binding/warprnnt_pytorch/delay_transducer.py:96-134 as one kernel.
 * kind 0 "zero": s / src_len; 1 "diagonal": |(s+1) * tgt/src - (u+1)| / tgt; 2 "diag_positive": max(.., 0) / tgt. */
rnntStatus_t w2vs_rnnt_delay_values(int kind, const int* src_lens, const int* tgt_lens, float* out, int minibatch,
                                    int maxT, int maxU, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* W2VS_RNNT_H */
