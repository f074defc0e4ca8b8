/*
 * dpp_hip.h -- C ABI of libdpp_hip.so, the MI355X (gfx950) kernels behind the DeepPrior++ hot path.
 *
 * The reference (moberweger/deep-prior-pp) has NO FFI / plugin / operator interface for this path:
 * every device op is generated implicitly by Theano 0.9 from the Python layer classes (SURVEY.md
 * section 8(b)).  This ABI is therefore new; each entry point names the reference call site whose
 * arithmetic it replaces.  The Python packages net/ and trainer/ (the drop-in boundary proper) bind it
 * with ctypes (deep-prior-pp_amd/hipdp/lib.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm allocations in the product);
 *     nothing is allocated, freed or retained across calls;
 *   - all tensors are float32; activations are NHWC ("pixel-major": row m = (n*H + y)*W + x holds the C
 *     channels of one pixel); a single-channel NCHW depth crop is bit-identical to its NHWC form;
 *   - every call is asynchronous on `stream`; return value 0 = launched, otherwise a hipError_t or a
 *     DPP_E_* code; no exceptions cross the boundary; thread-compatible (one stream per process/GPU);
 *   - convolution weights are kept in "kernel layout" Wk[Cout][taps][Cin] with the Theano true-
 *     convolution flip already applied: Wk[o][(dy+ph)*kw + (dx+pw)][c] = W_ref[o][c][kh-1-(dy+ph)]
 *     [kw-1-(dx+pw)], so the kernels compute a plain correlation.
 */
#ifndef DPP_HIP_H
#define DPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dpp_stream_t; /* a hipStream_t */

#define DPP_OK 0
#define DPP_E_BADARG 10001
#define DPP_E_UNSUPPORTED 10002

#define DPP_ABI_VERSION 1
int dpp_abi_version(void);

/* Pixel row map: row m of a compact (N,Ho,Wo) map -> row of a (N,Hi,Wi) map sampled with stride s.
 * s == 1 is the identity.  Used for Theano's `subsample` (convlayer.py:230-235) and its gradient. */
typedef struct {
    int s, Wo, HoWo, Wi, HiWi;
} dpp_rowmap;

/* Operand prologue applied while a tile is staged: the BatchNorm + ReLU that precede a conv / FC in the
 * pre-activation blocks (batchnormlayer.py:192, theano_helpers.py:61-69), evaluated as
 *   v = (x - mean[c]) * scale[c] + beta[c]   (mode & 2),   v = max(v, 0)   (mode & 1)
 * with c = (index along the operand's contiguous dimension) % cmod. */
typedef struct {
    const float* mean;
    const float* scale;
    const float* beta;
    int mode;
    int cmod;
} dpp_act;

/*
 * Generic f32 MFMA GEMM  C[M x N] = A_op[M x K] * B_op[K x N]  (v_mfma_f32_16x16x4_f32, LDS-staged).
 *   a_kc = 1: A_op(i,k) = A[mapA(i)*lda + k]      a_kc = 0: A_op(i,k) = A[mapA(k)*lda + i]
 *   b_kc = 1: B_op(k,j) = B[j*ldb + k]            b_kc = 0: B_op(k,j) = B[mapB(k)*ldb + j]
 * splitk == 1: C[mapC(i)*ldc + j] = acc + bias[j] + residual[mapC(i)*ldc + j]   (residual may alias C)
 * splitk  > 1: partial[z][i*N + j] = acc over K-slice z (reduce with dpp_reduce_partials).
 * Replaces, with the operand flags shown in deep-prior-pp_amd/hipdp/ops.py:
 *   1x1 ConvLayer forward                conv2d           /root/reference/src/net/convlayer.py:230-240
 *   its data / filter gradients          T.grad           /root/reference/src/trainer/poseregnettrainer.py:110-111
 *   HiddenLayer forward x.W + b          T.dot            /root/reference/src/net/hiddenlayer.py:136-139
 *   its gradients                        T.grad           poseregnettrainer.py:110-111
 *   residual add                         inputVar + conv  /root/reference/src/net/resnet.py:379,414
 */
typedef struct {
    const float* A; int lda; int a_kc; dpp_rowmap mapA; dpp_act actA;
    const float* B; int ldb; int b_kc; dpp_rowmap mapB; dpp_act actB;
    float* C; int ldc; dpp_rowmap mapC;
    const float* bias;
    const float* residual;
    int M, N, K;
    int splitk;
    float* partial;
    int bm, bn, wm; /* tile: rows, cols, waves along M (4 or 1); 0 = choose */
} dpp_gemm_desc;
int dpp_gemm(const dpp_gemm_desc* d, dpp_stream_t stream);

/* out[i] = sum_z partial[z*n + i] (+ bias[i % nbias] if bias) -- fixed summation order (deterministic). */
int dpp_reduce_partials(const float* partial, int nz, int n, const float* bias, int nbias, float* out,
                        dpp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPP_HIP_H */
