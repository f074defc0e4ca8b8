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
 *   - all tensors are float32 (exception: the bf16 STORAGE mode below); activations are NHWC ("pixel-major": row m = (n*H + y)*W + x holds the C
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

#define DPP_ABI_VERSION 11
int dpp_abi_version(void);

/* bf16 STORAGE of activation tensors (ABI v9; BASELINE config 5 "bf16 MFMA, 256x256 input stress").  The [pixels][channels] tensors the
 * convolutions write -- and only those: weights, gradients, FC outputs and every vector stay float32 -- may be held as bfloat16: the
 * producer's epilogue rounds the f32 value it formed (accumulator + bias + residual) to nearest-even on the store, the fused BatchNorm
 * statistics are still those of the UNROUNDED values, and every reader widens on the load (exact).  A `store` bit mask on the entry
 * points that can see such a tensor says which of its pointers address bf16 elements (same element offsets / leading dimensions as
 * f32; 8-byte alignment instead of 16).  0 everywhere = the float32 layout of ABI v8.
 * The GRADIENTS of those tensors -- the masked gradient G a data-gradient epilogue writes, the dX dpp_bn_bwd_apply writes -- may be
 * bf16-stored in the same way (rounded on the store, partial sums from the unrounded values): they take the A / C roles of the calls
 * that read / write them. */
#define DPP_ST_A 1    /* operand A / the input map X */
#define DPP_ST_B 2    /* operand B (the forward activations in a filter-gradient call) */
#define DPP_ST_C 4    /* the output C / Y and, when given, the residual */
#define DPP_ST_BNX 8  /* dpp_epilogue.bn_x (the BatchNorm input a data-gradient epilogue reads) */

/* Pixel row map: row m of a compact (N,Ho,Wo) map -> row of a (N,Hi,Wi) map sampled with stride s.
 * s == 1 is the identity.  Used for Theano's `subsample` (convlayer.py:230-235) and its gradient. */
typedef struct {
    int s, Wo, HoWo, Wi, HiWi;
} dpp_rowmap;

/* Operand prologue applied while a tile is staged: the BatchNorm + ReLU that precede a conv / FC in the
 * pre-activation blocks (batchnormlayer.py:192, theano_helpers.py:61-69), evaluated as
 *   v = (x - mean[c]) * scale[c] + beta[c]   (mode & 2),   v = max(v, 0)   (mode & 1)
 * with c = (index along the operand's contiguous dimension) % cmod.
 * mode == 4 (operand A of dpp_gemm only): the operand is the gradient through a BatchNorm's batch statistics,
 *   v = scale[c] * g - aux[c] * (x2 - mean[c]) - beta[c]
 * evaluated from the masked gradient g (the operand itself) and the BatchNorm input x2 (same shape and leading dimension
 * as the operand) instead of being materialised by dpp_bn_bwd_apply: with scale = gamma*inv_std, aux = scale*inv_std*c2 and
 * beta = scale*c1 (written by dpp_bn_bwd_finalize) this is dX = scale * (G - c1 - xhat*c2), batchnormlayer.py:119-194 under
 * T.grad.  If `out` is set (K-contiguous operand only) the values formed are also written there, in the operand's layout,
 * by the workgroups of the first column block: the data-gradient GEMM materialises dX on the way for the kernels that
 * want it as a plain tensor (the filter gradient on the other stream). */
typedef struct {
    const float* mean;
    const float* scale;
    const float* beta;
    int mode;
    int cmod;
    const float* x2;     /* mode 4 only */
    const float* aux;    /* mode 4 only */
    float* out;          /* mode 4 only, may be NULL */
} dpp_act;

/* Optional fused epilogue work on the output tile (both NULL = plain epilogue).
 *   stats      [2][N][row_blocks] (block index fastest, see the BatchNorm section): per-workgroup-row-block (mean, M2) of
 *              every output column, i.e. the partial BatchNorm statistics of the tensor being written
 *              (batchnormlayer.py:154-155) -- saves the separate statistics pass; combine with
 *              dpp_bn_finalize(partial = stats, nb = row_blocks, rows_per_block = tile rows).
 *   bn_x ...   BatchNorm-backward fusion for a data-gradient call: the value written is G = acc * [ (x-mean)*scale+beta >= 0 ]
 *              (bn_relu != 0) with x = bn_x at the output's index, and bn_partial [2][N][row_blocks] receives the per-block
 *              (sum G, sum G*xhat), xhat = (x-mean)*inv_std: exactly what dpp_bn_bwd_reduce would produce. */
typedef struct {
    float* stats;
    const float* bn_x;
    const float* bn_mean;
    const float* bn_inv_std;
    const float* bn_scale;
    const float* bn_beta;
    int bn_relu;
    int pad_;
    float* bn_partial;
} dpp_epilogue;

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
    int variant;    /* 0: LDS-tiled kernel (any layout, split-K); 1: row-streaming kernel for skinny conv GEMMs (a_kc = 1,
                       splitk = 1, bm in {64,128}, bn in {16,32,64}): A fragments straight from global memory, no barrier
                       in the K loop;
                       2: K-split kernel (a_kc = 1, splitk = 1, identity row maps, no actB; K = 256 with N % 64 == 0 or
                       K = 128 with N % 32 == 0, M % 32 == 0): a workgroup owns 32 rows x all columns x the whole K, every
                       load issued up front, the four waves split K; statistics blocks of 32 rows;
                       3: barrier-free row stream (a_kc = 1, splitk = 1, identity row maps, no actB; K = 64 -> N = 16
                       with M % 128 == 0, statistics blocks of 128 rows, or K = 16 -> N = 64 with M % 64 == 0, blocks of
                       64 rows): a wave owns whole 16-row tiles, epilogue in the MFMA D layout;
                       4: wave-autonomous strips for the channel-EXPANDING shapes (a_kc = 1, splitk = 1, identity row maps,
                       no actB; K in {16, 32, 64}, N % 64 == 0, M % bm == 0 with bm = rows per wave = rows per statistics
                       block, a multiple of 32, 0 = choose): a wave owns bm rows x 64 columns x the whole K, columns dealt
                       interleaved to the accumulator tiles so that the D layout gives 16-byte accesses; no LDS, no
                       barrier.  b_kc = 1 with the modes 0-3 prologue (forward), b_kc = 0 with a plain or mode-4
                       operand and the BatchNorm-backward epilogue (data gradient).
                       Variants 2-4 return DPP_E_UNSUPPORTED for anything else (bn / wm are ignored). */
    dpp_epilogue epi; /* fused statistics / BatchNorm-backward epilogue (requires splitk == 1) */
    int store;        /* DPP_ST_* mask: which of A, B, C (+ residual), epi.bn_x hold bf16 elements (0 = all float32).  Variant 4 and the
                         generic tiles with 16-byte-aligned whole-quad shapes take it; DPP_E_UNSUPPORTED otherwise. */
    int precision;    /* 0: f32 MFMA (exact); 1: both operands rounded to bf16 (RNE, A after its prologue), f32 accumulation on
                         v_mfma_f32_16x16x32_bf16 -- BASELINE config 5.  Variant 4 with K = 32 / 64 only (dpp_gemm_variant_rows says
                         0 otherwise); dpp_fc_gemm takes its precision as an argument and ignores this field. */
} dpp_gemm_desc;
int dpp_gemm(const dpp_gemm_desc* d, dpp_stream_t stream);
/* Would dpp_gemm run `d` on the kernel d->variant asks for (2, 3 or 4)?  Returns that kernel's rows per workgroup (the row-block count of
 * the fused-epilogue partials is M / rows), or 0: alignment, prologue mode or the epilogue rule it out and the caller should describe
 * the problem with variant 0 and a generic tile instead (dpp_gemm itself returns DPP_E_UNSUPPORTED rather than change the partial
 * layout behind the caller's back).  Launches nothing. */
int dpp_gemm_variant_rows(const dpp_gemm_desc* d);

/* Filter gradient of a 1x1 ConvLayer as a barrier-free row stream (csrc/wgrad.hip):
 *   partial[s][o][c] = sum over the pixel rows m of slice s of  dY[m][o] * act(X)[mapX(m)][c]
 * dY [M][Co] and X [rows][Ci] pixel-major (NHWC), mapX the stride row map of the layer (NULL = identity), actX the BatchNorm +
 * ReLU prologue of the layer's input (NULL = none; modes 0-3).  rows_per_wave (a multiple of 4) pixel rows go to one wave; the
 * number of slices written is dpp_wgrad_stream_slices(Co, Ci, M, rows_per_wave) -- 0 when (Co, Ci) is not one of the shapes of the
 * ResNet's 1x1 layers (16x64, 64x16, 16x32, 64x32, 32x64, 32x128, 128x32, 128x64, 64x128, 64x256, 256x64, 256x128), for which
 * dpp_gemm's generic filter-gradient layout (a_kc = b_kc = 0, splitk) remains.  Sum the slices with dpp_reduce_multi /
 * dpp_reduce_partials (fixed order).  T.grad of convlayer.py:230-240 (poseregnettrainer.py:110-111). */
int dpp_wgrad_stream_slices(int Co, int Ci, int M, int rows_per_wave);
int dpp_wgrad_stream(const float* dY, int Co, const float* X, int Ci, const dpp_rowmap* mapX, const dpp_act* actX, int M,
                     int rows_per_wave, float* partial, int store /* DPP_ST_B: X, DPP_ST_A: dY hold bf16 */, dpp_stream_t stream);
/* (ABI v11) the same with both operands on bf16 MFMA (the activation rounded after its prologue); the stage-1 shapes only
 * (dpp_wgrad_stream_bf16_ok: Co in {16, 64} with Ci in {16, 32, 64}), DPP_E_UNSUPPORTED otherwise.  Same slices / partial layout. */
int dpp_wgrad_stream_bf16_ok(int Co, int Ci);
int dpp_wgrad_stream_bf16(const float* dY, int Co, const float* X, int Ci, const dpp_rowmap* mapX, const dpp_act* actX, int M,
                     int rows_per_wave, float* partial, int store /* DPP_ST_B: X, DPP_ST_A: dY hold bf16 */, dpp_stream_t stream);

/* Filter gradient of a 3x3 'half'-padded, stride-1 ConvLayer on the same kind of stream:
 *   partial[s][o][t][c] = sum over the pixels p of slice s of  dY[p][o] * act(X)[p + (dy, dx)][c],  t = 3 (dy + 1) + (dx + 1),
 * taps outside the image contribute zero.  X [N][H][W][Ci], dY [N][H][W][Co], W % 4 == 0 (a step is four neighbouring pixels of a
 * row).  Shapes: Co == Ci in {16, 32, 64} (the bottleneck convolutions of resnet.py:300-420); dpp_wgrad3_stream_slices returns 0
 * for anything else and dpp_conv3x3_wgrad (the LDS-tiled kernel, same partial layout per block) remains.  Same T.grad. */
/* Filter gradient of a HiddenLayer whose reduction is only the batch (FC1: dW[k][n] = sum_b act(X)[b][k] * dY[b][n]; X [Nb][K],
 * dY [Nb][N], dW [K][N], all row-major; actX = the BatchNorm + ReLU prologue of the flattened map, channel = column % cmod) on the
 * row stream: a wave owns a 64 x 64 block of dW over all rows, no LDS, no partials.  K and N multiples of 128
 * (dpp_fc_wgrad_stream_ok), otherwise dpp_fc_gemm's layout remains.  T.grad of hiddenlayer.py:136-139. */
int dpp_fc_wgrad_stream_ok(int Nb, int K, int N);
int dpp_fc_wgrad_stream(const float* X, const float* dY, float* dW, int Nb, int K, int N, const dpp_act* actX,
                        int store /* DPP_ST_B: X holds bf16 */, dpp_stream_t stream);
int dpp_wgrad3_stream_slices(int Co, int Ci, int N, int H, int W, int rows_per_wave);
int dpp_wgrad3_stream(const float* dY, int Co, const float* X, int Ci, int N, int H, int W, const dpp_act* actX, int rows_per_wave,
                      float* partial, int store /* DPP_ST_B: X, DPP_ST_A: dY hold bf16 */, dpp_stream_t stream);

/* The same contract on the weight-streaming kernel for the HiddenLayer behind the last convolution map (FC1: 16 384 x 1 024
 * weights at 128x128 input, 65 536 x 1 024 at 256x256; hiddenlayer.py:136-139 and its T.grad): tile 128 x 64, both operands
 * K-contiguous in LDS (memory-MN-contiguous operands are transposed while staged), double-buffered LDS.
 * precision 0: exact f32 (v_mfma_f32_16x16x4_f32; equals dpp_gemm up to summation order).
 * precision 1: operands rounded to bf16 (RNE, after the prologue), f32 accumulation (v_mfma_f32_16x16x32_bf16) -- BASELINE
 *              config 5; not for the 1e-3 mm parity path.
 * kchunk: 32 | 64 (0 = 64).  d->bm / bn / wm / variant are ignored; no fused statistics / BatchNorm-backward epilogue. */
int dpp_fc_gemm(const dpp_gemm_desc* d, int precision, int kchunk, dpp_stream_t stream);

/* out[i] = sum_z partial[z*n + i] (+ bias[i % nbias] if bias) -- fixed summation order (deterministic). */
int dpp_reduce_partials(const float* partial, int nz, int n, const float* bias, int nbias, float* out,
                        dpp_stream_t stream);


/* Batched form: jobs_dev = device array of { const float* partial; float* out; int nz, n, block0, pad; } sorted by block0,
 * job j owning workgroups [block0_j, block0_j + ceil(n_j / dpp_reduce_multi_block_cols())); total_blocks = sum.  One launch
 * reduces every filter / bias gradient partial of a backward pass. */
size_t dpp_reduce_job_bytes(void);
int dpp_reduce_multi_block_cols(void);
int dpp_reduce_multi(const void* jobs_dev, int njobs, int total_blocks, dpp_stream_t stream);

/* ---- 3x3 'half' stride-1 ConvLayer on NHWC maps (implicit GEMM, halo tile in LDS) --------------------------
 * Y[n,y,x,o] = sum_{tap,c} act(X)[n, y+dy, x+dx, c] * Wk[o][tap][c] + bias[o] + residual[n,y,x,o], zero padding applied
 * after `act`.  conv2d 3x3 of res_block, /root/reference/src/net/resnet.py:365-368,394-397 via
 * /root/reference/src/net/convlayer.py:230-240.  The data gradient is the same call on dY with the weights from
 * dpp_conv3x3_wtrans.  bm = 64 | 128 rows per workgroup (0 = choose).  Ci, Co multiples of 16. */
int dpp_conv3x3(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* Wk, int Co,
                const float* bias, const float* residual, float* Y, int bm, const dpp_epilogue* epi,
                int store /* DPP_ST_A: X, DPP_ST_C: Y + residual, DPP_ST_BNX: epi->bn_x hold bf16 */, dpp_stream_t stream);
/* The same with the activated input and the weights rounded to bf16 (RNE) as they are staged in LDS, f32 accumulation
 * (v_mfma_f32_16x16x32_bf16): BASELINE config 5, not for the 1e-3 mm parity path.  Same tiling, epilogues and results layout. */
int dpp_conv3x3_bf16(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* Wk, int Co,
                     const float* bias, const float* residual, float* Y, int bm, const dpp_epilogue* epi, int store, dpp_stream_t stream);
/* The same convolution for the narrow square layers (C = Ci = Co in {16, 32}, W % 16 == 0, float32 tensors, no residual) as a
 * wave-autonomous stream: a wave owns 16-pixel pieces of image rows, keeps the 9 x C x C filter in registers and loads every tap
 * straight from global memory -- no halo tile, no barrier before the column reductions.  dpp_conv3x3_stream_rows = pixels per
 * workgroup (= per block of epi->stats / epi->bn_partial), or 0 when the shape is not taken (then dpp_conv3x3_stream returns
 * DPP_E_UNSUPPORTED and dpp_conv3x3 remains).  Same arithmetic per output element as dpp_conv3x3 up to the summation order. */
int dpp_conv3x3_stream_rows(int N, int H, int W, int C);
int dpp_conv3x3_stream(const float* X, int N, int H, int W, int C, const dpp_act* act, const float* Wk, const float* bias, float* Y,
                       const dpp_epilogue* epi, dpp_stream_t stream);
/* tile geometry chosen for (N,H,W,bm): returns the number of workgroup row blocks, writes tile height / width / images */
int dpp_conv3x3_tiling(int N, int H, int W, int bm, int* th, int* tw, int* img);
/* Wd[c][8-tap][o] = Wk[o][tap][c] (mirrored taps, channels swapped): weights of the data-gradient correlation. */
int dpp_conv3x3_wtrans(const float* Wk, int Co, int Ci, float* Wd, dpp_stream_t stream);
/* Batched form: jobs_dev = device array of { const float* Wk; float* Wd; int Co, Ci, block0, pad; } sorted by block0, job j owning
 * workgroups [block0_j, block0_j + ceil(Co*9*Ci / 256)); one launch mirrors the weights of every 3x3 layer of a net. */
size_t dpp_wtrans_job_bytes(void);
int dpp_conv3x3_wtrans_multi(const void* jobs_dev, int njobs, int total_blocks, dpp_stream_t stream);
/* Filter gradient partials: partial[blk][o][tap][c] = sum over the workgroup's pixels of dY[.,o] * act(X)[.+tap, c];
 * blk < dpp_conv3x3_wgrad_blocks(N,H,W,Ci,Co,bm); sum over blk with dpp_reduce_partials.  (T.grad, poseregnettrainer.py:110-111) */
int dpp_conv3x3_wgrad_blocks(int N, int H, int W, int Ci, int Co, int bm);
int dpp_conv3x3_wgrad(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, int Co,
                      float* partial, int bm, int store /* DPP_ST_A: X, DPP_ST_B: dY hold bf16 */, dpp_stream_t stream);
/* (ABI v11) the same with bf16 MFMA operands (BASELINE config 5; `T.grad` of /root/reference/src/net/convlayer.py:230-240 evaluated with
 * both operands rounded to bfloat16, f32 accumulation): act(X) is rounded to nearest-even AFTER the prologue, dY is rounded likewise (exact
 * when it is bf16-stored).  Layers dpp_conv3x3_wgrad_bf16_ok() accepts only (16 or 32 channels in and out, maps >= 12 wide);
 * DPP_E_UNSUPPORTED otherwise.  Same partial layout and slice count as dpp_conv3x3_wgrad. */
int dpp_conv3x3_wgrad_bf16_ok(int N, int H, int W, int Ci, int Co);
int dpp_conv3x3_wgrad_bf16(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, int Co,
                           float* partial, int bm, int store /* DPP_ST_A: X, DPP_ST_B: dY hold bf16 */, dpp_stream_t stream);

/* ---- ResNet stem: ConvPoolLayer 5x5 'half' 1 -> Co (<= 32), 2x2 max-pool, bias AFTER the pool ----------------
 * /root/reference/src/net/convpoollayer.py:251-282 as built at /root/reference/src/net/resnet.py:128-133.
 * X: [N][H][W] single-channel crops (NCHW == NHWC); Wk: [Co][25]; Y: [N][H/2][W/2][Co]; argmax: the TIE MASK of each
 * 2x2 window (bit j set = window element j, row-major, equals the maximum), kept for the filter gradient: Theano's
 * MaxPoolGrad gives the gradient to every element equal to the maximum, and the constant background of a depth crop
 * makes whole windows tie.
 * stats (may be NULL; needs H % 16 == W % 16 == 0): [2][Co][N * H/16 * W/16] per-tile (mean, M2) of the 64 pooled outputs a
 * workgroup writes -- the BatchNorm statistics partial of Y, combined by dpp_bn_finalize(rows_per_block = 64). */
int dpp_stem_fwd(const float* X, int N, int H, int W, const float* Wk, const float* bias, int Co, float* Y, uint8_t* argmax,
                 float* stats, int store /* DPP_ST_C: Y holds bf16 */, dpp_stream_t stream);
int dpp_stem_wgrad_blocks(int N, int H, int W, int tiles_per_block);
int dpp_stem_wgrad(const float* X, int N, int H, int W, const float* dY, const uint8_t* argmax, int Co, float* partial,
                   int tiles_per_block, dpp_stream_t stream);

/* ---- generic ConvPoolLayer: kh x kw conv ('valid': pad 0, 'half': pad k/2), pool x pool max-pool (ignore_border), bias
 * AFTER the pool; the activation is the consumer's operand prologue.  /root/reference/src/net/convpoollayer.py:251-282
 * as built by PoseRegNet, /root/reference/src/net/poseregnet.py:62-78 (5x5/pool 4, 5x5/pool 2, 3x3/no pool, 8 filters).
 * X: [N][H][W][Ci] with the optional ReLU / BN prologue `act`; Wk: [Co][kh*kw][Ci] (flipped, layout.py);
 * Y: [N][Hp][Wp][Co], Hp = (H + 2 pad - kh + 1) / pool; ties: [N][Hp][Wp][Co] uint16 tie masks of the pool windows
 * (bit j = window element j, row-major, equals the maximum; NULL allowed when pool == 1 or in inference).
 * wgrad: partial[blk][Co][kh*kw][Ci], blk < dpp_convpool_wgrad_blocks(N, Hp, Wp); sum with dpp_reduce_multi.
 * dgrad: dX[N][H][W][Ci] = gradient w.r.t. the (activated) input operand.  Limits: Co <= 32, Ci <= 32, pool <= 4, k <= 7.
 * These layers are 12.8 MFLOP / sample in total: VALU kernels, one thread per output pixel. */
int dpp_convpool_fwd(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* Wk, int kh, int kw, int pad,
                     int Co, int pool, const float* bias, float* Y, uint16_t* ties, dpp_stream_t stream);
int dpp_convpool_wgrad_blocks(int N, int Hp, int Wp);
int dpp_convpool_wgrad(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, const uint16_t* ties,
                       int kh, int kw, int pad, int Co, int pool, float* partial, dpp_stream_t stream);
int dpp_convpool_dgrad(const float* dY, const uint16_t* ties, int N, int H, int W, int Ci, const float* Wk, int kh, int kw,
                       int pad, int Co, int pool, float* dX, dpp_stream_t stream);

/* ---- BatchNormLayer, /root/reference/src/net/batchnormlayer.py:119-194 (tensors pixel-major [M][C], C % 4 == 0) ----
 * stats_partial: per row-chunk (mean_b, M2_b) -> partial[2][C][nb], nb = ceil(M / rows_per_block); the block index is the
 *                fastest one so that the finalize (one wave per channel, lanes over blocks) reads contiguous memory
 * finalize:      batch mean, inv_std = 1/sqrt(var_biased + eps), scale = gamma*inv_std; running mean / inv_std EMA
 *                (alpha = 0 or run_mean == NULL: no update).  `partial` holds nseg consecutive [2][C][nb] segments (nseg = 1,
 *                or the ranks of a synchronised-BatchNorm all-gather) covering M rows in total, M / nseg per segment
 * eval_coeffs:   deterministic mode: mean = run_mean, inv_std = run_inv_std, scale = gamma*run_inv_std
 * bwd_reduce:    G = dA * [ (x-mean)*scale+beta >= 0 ] (relu != 0) or dA; partial[2][C][nb] = sum G, sum G*xhat
 * bwd_finalize:  dbeta, dgamma and c1 = dbeta/M, c2 = dgamma/M; if q != NULL also q = scale*c1 and p = scale*inv_std*c2, the
 *                per-channel constants of the mode-4 operand prologue (dpp_act)
 * bwd_apply:     dX = scale * (G - c1 - xhat*c2) + add      (gradient through the batch statistics) */
int dpp_bn_stats_partial(const float* X, int M, int C, int rows_per_block, float* partial, int store /* DPP_ST_A: X holds bf16 */,
                         dpp_stream_t stream);
int dpp_bn_finalize(const float* partial, int nb, int nseg, int M, int rows_per_block, int C, const float* gamma, float eps,
                    float* mean, float* inv_std, float* scale, float* run_mean, float* run_inv_std, float alpha,
                    dpp_stream_t stream);
int dpp_bn_eval_coeffs(const float* gamma, const float* run_mean, const float* run_inv_std, int C, float* mean,
                       float* inv_std, float* scale, dpp_stream_t stream);
/* eval_coeffs for EVERY BatchNorm of a net in ONE launch (ABI v10): the test-time forward pass issued one 4.7 us launch per
 * BatchNorm -- 61 of its 131 launches.  jobs (device): dpp_bn_eval_job_bytes() bytes each = { const float* gamma, * run_mean, * run_inv_std;
 * float* mean, * inv_std, * scale; int C; int block0 } with block0 = the first 256-thread block of the job, total_blocks their sum. */
size_t dpp_bn_eval_job_bytes(void);
int dpp_bn_eval_coeffs_multi(const void* jobs, int njobs, int total_blocks, dpp_stream_t stream);
int dpp_bn_bwd_reduce(const float* dA, const float* X, int M, int C, const float* mean, const float* inv_std,
                      const float* scale, const float* beta, int relu, float* G, int rows_per_block, float* partial,
                      int store /* DPP_ST_BNX: X, DPP_ST_A: the incoming gradient (dA / G), DPP_ST_C: the outgoing one (G / dX + add) hold bf16 */, dpp_stream_t stream);
int dpp_bn_bwd_finalize(const float* partial, int nb, int nseg, int M, int C, float* dbeta, float* dgamma, float* c1, float* c2,
                        const float* inv_std, const float* scale, float* q, float* p, dpp_stream_t stream);
int dpp_bn_bwd_apply(const float* G, const float* X, int M, int C, const float* mean, const float* inv_std,
                     const float* scale, const float* c1, const float* c2, const float* add, float* dX,
                     int rows_per_block, float* colsum_partial /* [nb][C] column sums of dX, or NULL */,
                     int store /* DPP_ST_BNX: X, DPP_ST_A: the incoming gradient (dA / G), DPP_ST_C: the outgoing one (G / dX + add) hold bf16 */, dpp_stream_t stream);
/* dpp_bn_bwd_finalize (nseg = 1, no q / p) + dpp_bn_bwd_apply in one launch, for BatchNorms whose per-block sums are few (the small
 * maps of the late stages): every workgroup -- grid (ceil(M / rows_per_block), C / 32) -- reduces the `nb` blocks of `partial`
 * ([2][C][nb], the layout dpp_bn_bwd_reduce and the fused data-gradient epilogues write) for its 32 channels itself, then applies.
 * Writes dbeta / dgamma like the finalize kernel; c1 / c2 never reach memory.  dpp_bn_bwd_finalize_apply_ok: C % 32 == 0 and
 * nb <= 256, else DPP_E_UNSUPPORTED (the two launches remain). */
int dpp_bn_bwd_finalize_apply_ok(int M, int C, int nb);
int dpp_bn_bwd_finalize_apply(const float* G, const float* X, int M, int C, const float* mean, const float* inv_std,
                              const float* scale, const float* partial, int nb, const float* add, float* dX, int rows_per_block,
                              float* colsum_partial, float* dbeta, float* dgamma, int store, dpp_stream_t stream);

/* ---- loss / optimiser / small elementwise ------------------------------------------------------------------ */
/* partial[b][c] = sum of rows of chunk b (bias gradients; reduce with dpp_reduce_partials) */
int dpp_colsum_partial(const float* X, int M, int C, int rows_per_block, float* partial, dpp_stream_t stream);
/* cost = (1/denom) sum (out-y)^2, dout = (2/denom)(out-y): /root/reference/src/trainer/poseregnettrainer.py:92-99
 * (denom = batch for the embedding loss, batch*numJoints for the joint loss); dout may be NULL */
int dpp_loss_sse(const float* out, const float* y, int rows, int d, int denom, float* cost, float* dout, dpp_stream_t stream);
/* (ABI v11) dpp_reduce_partials (out[rows][d] = sum of nz split-K slices + bias[d], the last HiddenLayer of the net,
 * /root/reference/src/net/hiddenlayer.py:136-139) and dpp_loss_sse on that output in ONE launch; rows * d <= 65536. */
int dpp_reduce_partials_loss(const float* partial, int nz, int rows, int d, const float* bias, float* out, const float* y, int denom,
                             float* cost, float* dout, dpp_stream_t stream);
/* The scalar-target case (numJoints == nDims == 1, poseregnettrainer.py:84-85, 92-93): Theano broadcasts the (B, 1) output against
 * the VECTOR y, so the cost is mean_i mean_j (out_i - y_j)^2 and dout_i = (2 / B)(out_i - mean(y)); n = B.  err (may be NULL):
 * the monitor of :115 under the same broadcast, err[0] = mean, err[1] = max over all pairs of |out_i - y_j|. */
int dpp_loss_sse_bcast(const float* out, const float* y, int n, float* cost, float* dout, float* err, dpp_stream_t stream);
/* err[0] = mean_rows sqrt(sum_d (out-y)^2), err[1] = max_rows: poseregnettrainer.py:114-129 (errors, errors_avg, errors_max) */
int dpp_error_l2(const float* out, const float* y, int rows, int d, float* err, dpp_stream_t stream);
/* The reference's ADAM (/root/reference/src/trainer/optimizer.py:58-90) over a flat parameter buffer.
 * state (device, 8 floats): lr, t, beta1, beta2, epsilon, gamma, 0, 0 -- the bias-correction terms are evaluated on the
 * device in float32; dpp_adam_tick performs `t <- t + 1` (optimizer.py:88) after the update. */
int dpp_adam(float* w, const float* g, float* m, float* v, size_t n, const float* state, dpp_stream_t stream);
/* (ABI v11) dpp_adam followed by dpp_adam_tick in ONE launch: the last workgroup to finish advances t (hyper[1]); hyper[7] is the
 * launch's ticket counter (an unsigned, zero between launches).  For an update that is a single launch over the whole flat buffer. */
int dpp_adam_ticked(float* w, const float* g, float* m, float* v, size_t n, float* hyper, dpp_stream_t stream);
int dpp_adam_tick(float* state, dpp_stream_t stream);
int dpp_axpy(float* y, const float* x, float alpha, size_t n, dpp_stream_t stream);              /* y += alpha x */
int dpp_sumsq(const float* x, size_t n, float alpha, float* out, int accumulate, dpp_stream_t stream);
/* The two above over nseg segments of ONE flat buffer in one launch: seg (DEVICE memory) holds (offset, length) pairs in elements
 * -- the weights of all layers inside the flat parameter / gradient buffers (cost += weightreg_factor * sum(W^2) and its gradient
 * 2 * weightreg_factor * W, poseregnettrainer.py:101-107).  Deterministic: fixed shares per block, f64 partials summed in order.
 * workspace: dpp_sumsq_multi_workspace_bytes() bytes of device memory. */
size_t dpp_sumsq_multi_workspace_bytes(void);
int dpp_sumsq_multi(const float* base, const long long* seg, int nseg, float alpha, void* workspace, float* out, int accumulate,
                    dpp_stream_t stream);
int dpp_axpy_multi(float* ybase, const float* xbase, const long long* seg, int nseg, float alpha, dpp_stream_t stream);
/* y = mask ? mask*relu?(x) : a*relu?(x): DropoutLayer, /root/reference/src/net/dropoutlayer.py:98-104 */
int dpp_scale(const float* x, const float* mask, float a, int relu, float* y, size_t n, dpp_stream_t stream);
/* g = (mask ? mask : a) * dy * [pre >= 0] */
int dpp_relu_bwd(const float* dy, const float* pre, const float* mask, float a, float* g, size_t n, dpp_stream_t stream);

int dpp_fill_zero(void* p, size_t nbytes, dpp_stream_t stream);
/* dst[r][c] = relu ? max(src[r][c], 0) : src[r][c] over rows x cols with row strides lds / ldd: the column-wise
 * concatenation of ScaleNet's flattened tower outputs (/root/reference/src/net/scalenet.py:167-171) and its gradient split */
int dpp_copy2d(const float* src, int lds, float* dst, int ldd, int rows, int cols, int relu, dpp_stream_t stream);
/* out[r][c] = x[r][c] * (s[r*sld + scol] * factor): normalised labels back to mm (label * cube_z / 2), the augmentation's input
 * when the labels are the joints themselves (/root/reference/src/trainer/poseregnettrainer.py:228-240) */
int dpp_rowscale(const float* x, const float* s, int sld, int scol, float factor, float* out, int rows, int cols, dpp_stream_t stream);
/* centre h x w window of each [H][W] image: ScaleNet's 1/2 and 1/4 inputs
 * (/root/reference/src/trainer/scalenettrainer.py:239-251, /root/reference/src/util/handdetector.py:654-666) */
int dpp_crop_center(const float* src, int B, int H, int W, float* dst, int h, int w, dpp_stream_t stream);
/* mask[i] = 1 with probability keep (counter-based generator keyed by seed, counter + *counter_dev, i): DropoutLayer
 * masks; counter_dev (may be NULL) is a device-resident step counter so that a recorded launch draws a new mask per step */
int dpp_bernoulli_mask(float* mask, size_t n, float keep, unsigned long long seed, unsigned long long counter,
                       const unsigned long long* counter_dev, dpp_stream_t stream);

/* ---- online crop augmentation, NetTrainer.augmentCrop (/root/reference/src/trainer/nettrainer.py:919-997) -------
 * prepare: per crop, the geometry of HandDetector.moveCoM / rotateHand / scaleHand
 *          (/root/reference/src/util/handdetector.py:678-780): new CoM, crop transform, inverse warp matrix,
 *          z-thresholds, augmented joint labels, PCA-prior projection (poseregnettrainer.py:262) -> records, out_y.
 *          norm_zero_one: bit 0: the crops are normalised to [0, 1] (normZeroOne, nettrainer.py:948, 982-988) instead of [-1, 1];
 *          bit 1: binarizeImage (poseregnettrainer.py:255-257): the augmented crop is thresholded, < 0.5 -> 0, >= 0.5 -> 1.
 *          mode codes: 0 none, 1 com, 2 rot, 3 sc.  mode == NULL: (mode, off, rot, sc) are drawn on the device from
 *          Philox(seed, counter, sample) with mode = mode_table[u % n_modes].
 * warp:    per pixel, cv2.warpAffine / warpPerspective (NEAREST, constant 0) + recropHand's z-clamp
 *          (handdetector.py:782-803) + the far-plane fill / clamp / re-normalisation of nettrainer.py:982-995. */
size_t dpp_augment_record_bytes(void);
int dpp_augment_prepare(const float* img, const float* com3d, const float* cube, const float* Mcrop, const float* gt3d,
                        int B, int J, int dsz, const int* mode, const double* off, const double* rot, const double* sc,
                        const int* mode_table, int n_modes, unsigned long long seed, unsigned long long counter,
                        double sigma_com, double sigma_sc, double rot_range, double fx, double fy, double ux, double uy,
                        int flip_y, int norm_zero_one, const float* pca_mean, const float* pca_comp, int E, void* records, float* out_y,
                        int* out_mode, const unsigned long long* counter_dev, dpp_stream_t stream);
/* *counter += inc (device-resident draw counter added to `counter` when counter_dev != NULL, so replayed graphs draw fresh
 * augmentation parameters every step) */
int dpp_counter_add(unsigned long long* counter, unsigned long long inc, dpp_stream_t stream);
int dpp_augment_warp(const float* img, const void* records, int B, int dsz, float* out, dpp_stream_t stream);
/* Both stages as ONE launch (what the trainer and bench.py use): prepare + warp of B crops into out_x [B][dsz][dsz] and
 * out_y, `records` optional (NULL: not written).  Device draws (mode == NULL) are keyed by
 *   (seed, (counter + *counter_dev) * global_batch + sample0 + b)
 * i.e. by (seed, step, GLOBAL sample index): a data-parallel rank passes sample0 = rank * B and the global minibatch size, so
 * what a sample draws does not depend on the number of GPUs (global_batch = 0 means B).  With ticket != NULL (a device
 * word, zero before the first call) the launch also advances *counter_dev by one after every workgroup has read it, so a
 * recorded plan / replayed graph draws fresh parameters each step without a dpp_counter_add launch.
 * splits: workgroups per crop (a power of two dividing dsz*dsz/4; 0 = choose so that the grid fills the chip). */
int dpp_augment(const float* img, const float* com3d, const float* cube, const float* Mcrop, const float* gt3d,
                int B, int J, int dsz, const int* mode, const double* off, const double* rot, const double* sc,
                const int* mode_table, int n_modes, unsigned long long seed, unsigned long long counter,
                double sigma_com, double sigma_sc, double rot_range, double fx, double fy, double ux, double uy,
                int flip_y, int norm_zero_one, const float* pca_mean, const float* pca_comp, int E, void* records,
                float* out_x, float* out_y, int* out_mode, unsigned long long* counter_dev, unsigned* ticket,
                unsigned long long sample0, unsigned long long global_batch, int splits, dpp_stream_t stream);

/* ---- initial crop: HandDetector.cropArea3D (docom = False) fused with Dataset.imgStackDepthOnly -------------------------
 * /root/reference/src/util/handdetector.py:53-68, 204-226, 260-296, 382-490; /root/reference/src/data/dataset.py:97-103.
 * frames: [B][H][W] raw depth in mm (0 = not defined); com: [B][3] crop centre in IMAGE coordinates (u, v, d mm);
 * cube: [B][3] metric crop size in mm.  prepare: per frame the detector's valid depth range [max(10, min), min(1500, max)],
 * the crop window (comToBounds), the aspect-preserving nearest-neighbour resize geometry (cv2.resize INTER_NEAREST) and the
 * crop transform M_out [B][9] (= comToTransform; may be NULL).  warp: out [B][dsz][dsz]: the crop in mm with background
 * nd_value (normalize == 0, what cropArea3D returns) or normalised like the training stacks: 0 -> com_z + cube_z/2, then
 * (d - com_z) / (cube_z / 2) (normalize != 0).  stretch != 0: the window is resized to dsz x dsz as it is (no aspect-preserving
 * size, no centred paste): `resizeCrop(cropped, dsize)`, the image cropArea3D shows its refinement net (handdetector.py:430). */
size_t dpp_crop_record_bytes(void);
int dpp_crop_prepare(const float* frames, int B, int H, int W, const float* com, const float* cube, double fx, double fy,
                     int dsz, int stretch, void* records, float* M_out, dpp_stream_t stream);
int dpp_crop_warp(const float* frames, const void* records, int B, int H, int W, int dsz, int normalize, float nd_value,
                  float* out, dpp_stream_t stream);
/* docom = True (handdetector.py:413-427): the centre of mass (calculateCoM, :91-108) of the crop window described by
 * `records`, in image coordinates -> com_out [B][3]; re-run dpp_crop_prepare with it, then dpp_crop_warp. */
size_t dpp_crop_com_workspace_bytes(int B);       /* device scratch of dpp_crop_com: per-band partial sums, summed in a fixed order */
int dpp_crop_com(const float* frames, const void* records, int B, int H, int W, void* workspace, float* com_out, dpp_stream_t stream);
/* docom = True WITH a refinement net (handdetector.py:429-440 and refineCoM, :634-676), batched: net_out [B][3] is the net's
 * output for the re-centred crops (normalised offset of the hand centre), com_in [B][3] the centre those crops were cut around
 * (dpp_crop_com's output), records the crop records of that second dpp_crop_prepare.
 *   com_out [B][3]  = joint3DToImg(net_out * cube_z / 2 + jointImgTo3D(com_in)); an all-zero result takes the depth of the crop
 *                     window's centre pixel.  Re-run dpp_crop_prepare / dpp_crop_warp with it (any dsz) for the final crop.
 * Optional, for a refine -> re-crop -> regress cascade that never leaves the device (importers.py:388-392, dataset.py:103,
 * poseregnettrainer.py:262): gt3d_orig [B][J][3] -> gt3d_crop = gt3d_orig - jointImgTo3D(com_out) (may be NULL), com3d_out
 * [B][3] (may be NULL) and out_y = gt3d_crop / (cube_z / 2), projected onto the PCA prior when pca_comp [E][J*3] / pca_mean are
 * given ([B][E]) and raw otherwise ([B][J*3]).  (fx, fy, ux, uy, flip_y): the importer's camera, signs included. */
int dpp_crop_refine(const float* frames, const void* records, int B, int H, int W, const float* com_in, const float* cube,
                    const float* net_out, double fx, double fy, double ux, double uy, int flip_y, const float* gt3d_orig, int J,
                    const float* pca_mean, const float* pca_comp, int E, float* com_out, float* com3d_out, float* gt3d_crop,
                    float* out_y, dpp_stream_t stream);

/* ---- PCA prior set-up and evaluation on the device --------------------------------------------
 * pose_sample: HandDetector.sampleRandomPoses (/root/reference/src/util/handdetector.py:805-909) for n samples: sample i augments
 *   base pose ridx[i] in label space with mode[i] (0 none, 1 com, 2 rot, 3 sc, 4 rot+com, 5 rot+com+sc) and the draws off[i][3],
 *   sc[i], rot[i] (degrees) -- the host draws them from the script's RandomState in the reference's order, the arithmetic (f64
 *   projections, f32 storage) runs here.  out_poses [n][J][3] normalised by the (new) cube_z / 2; out_com / out_cube may be NULL.
 * pca_fit: sklearn PCA.fit on X [N][D] f32 (main_nyu_posereg_embedding.py:86-92): mean[D], the eigenvalues of the covariance
 *   (1 / (N - 1)) in descending order and the eigenvectors as ROWS (components_, largest-magnitude entry of each row positive),
 *   all f64; D <= 80 (26 joints: the scatter tile and the Jacobi matrices are LDS-resident; DPP_E_BADARG beyond).  The Jacobi sweeps
 *   run until the off-diagonal mass is at f64 rounding level of the diagonal's (at most 30).  workspace: dpp_pca_workspace_bytes(N, D) bytes of device memory.
 * pose_eval: HandposeEvaluation's numeric methods (/root/reference/src/util/handpose_evaluation.py:92-228) on gt / pred [N][J][3]:
 *   err [N][J] Euclidean errors, frame [N][4] = per-frame (nanmean, nanmax, count, nanstd) over joints, and out (4 + 3J + 2T
 *   doubles): mean error, max error, mean of the frame stds, frames counted; per-joint nanmean / nanstd / nanmax; for each of the T
 *   thresholds the frames whose max (then: mean) error is <= it. */
int dpp_pose_sample(const float* base_poses, const float* base_com, const float* base_cube, int n_base, int J, const int* mode,
                    const int* ridx, const double* off, const double* sc, const double* rot, long n, double fx, double fy,
                    double ux, double uy, int flip_y, float* out_poses, float* out_com, float* out_cube, dpp_stream_t stream);
/* (ABI v11) the same with rot3D=True (handdetector.py:870, 891, 903): the rotation modes turn the pose in 3-D about the (new) centre --
 * rotatePoints3D, /root/reference/src/data/transformations.py:105-155 -- by the sample's matrix rot3[i] (3x3 row-major f64: getRotationMatrix of
 * the three drawn angles, formed on the host like the draws themselves) instead of in the image plane. */
int dpp_pose_sample_rot3d(const float* base_poses, const float* base_com, const float* base_cube, int n_base, int J, const int* mode,
                          const int* ridx, const double* off, const double* sc, const double* rot3, long n, double fx, double fy,
                          double ux, double uy, int flip_y, float* out_poses, float* out_com, float* out_cube, dpp_stream_t stream);
size_t dpp_pca_workspace_bytes(long N, int D);
int dpp_pca_fit(const float* X, long N, int D, void* workspace, double* mean, double* evals, double* components, dpp_stream_t stream);
int dpp_pose_eval(const float* gt, const float* pred, int N, int J, const double* thresholds, int T, double* err, double* frame,
                  double* out, dpp_stream_t stream);

/* ---- a whole bottleneck block of the deterministic forward pass as ONE launch (ABI v10) ----------------------------------------
 * res_block of /root/reference/src/net/resnet.py:349-414 with every BatchNormLayer in deterministic mode
 * (/root/reference/src/net/batchnormlayer.py:158-159: stored running mean / inv_std), as netbase.py:257-310 (computeOutput) runs it:
 *     h = relu(bn0(x));  c1 = conv1x1_s(h) + b1;  c2 = conv3x3(relu(bn1(c1))) + b2;  c3 = conv1x1(relu(bn2(c2))) + b3
 *     Y = x + c3  (identity block: Wsc == NULL, Cin == Cout, stride 1)     Y = c3 + conv1x1_s(h) + bsc  (projection block)
 * bn(v) = (v - mean) * (gamma * inv_std) + beta.  X [N][H][W][Cin], Y [N][Ho][Wo][Cout] (Ho = ceil(H / stride)), weights in kernel
 * layout: W1 [Nb][Cin], W2 [Nb][9][Nb], W3 [Cout][Nb], Wsc [Cout][Cin].  float32 tensors, f32 MFMA.  The 16- / 32- / 64-channel
 * intermediates stay in LDS (csrc/resblock.hip).  dpp_resblock_eval_ok: 1 when the kernel takes the shape, else the caller issues the
 * block layer by layer (dpp_gemm / dpp_conv3x3 with dpp_act prologues). */
typedef struct {
    const float* mean;
    const float* inv_std;
    const float* gamma;
    const float* beta;
} dpp_bn_eval;
typedef struct {
    const float* X; int N, H, W, Cin;
    int stride, Ho, Wo, Cout, Nb;
    dpp_bn_eval bn0, bn1, bn2;
    const float* W1; const float* b1;
    const float* W2; const float* b2;
    const float* W3; const float* b3;
    const float* Wsc; const float* bsc;
    float* Y;
    int store;      /* (ABI v11) DPP_ST_A: X holds bf16 elements, DPP_ST_C: Y does.  DPP_ST_C selects the bf16 MODE of the block (BASELINE config 5):
                     * what the layer-by-layer bf16 path stores is rounded to bfloat16 where it would have been stored, what it multiplies on bf16
                     * MFMA operands is rounded as its kernels round it.  DPP_ST_A alone is refused. */
} dpp_resblock_desc;
int dpp_resblock_eval_ok(int Cin, int Cout, int Nb, int stride, int projection);
int dpp_resblock_eval(const dpp_resblock_desc* d, dpp_stream_t stream);
/* (ABI v11) the status dpp_resblock_eval would return for this descriptor, without launching (alignment, LDS size, offsets, storage) */
int dpp_resblock_eval_check(const dpp_resblock_desc* d);

/* ---- launch plans: a whole train / inference step as ONE call --------------------------------------------------------
 * The reference runs `train_model(index, lr)` as one compiled device function (theano.function,
 * /root/reference/src/trainer/poseregnettrainer.py:146-170, called at /root/reference/src/trainer/nettrainer.py:840).
 * A plan is the equivalent here: the ordered kernel launches of a step, recorded once and re-issued from C++ without any
 * host-language work per launch.  While a plan is recording on the calling thread (dpp_plan_record_begin ..
 * dpp_plan_record_end) every dpp_* launch call of that thread appends its fully resolved launch (kernel, grid, arguments)
 * to the plan instead of launching, its `stream` argument being ignored.  A plan has two lanes: lane 0 (main) and lane 1 (side,
 * the parameter-gradient branch).  dpp_plan_fork makes the side lane wait for everything recorded so far on the main lane,
 * dpp_plan_join makes the main lane wait for everything recorded so far on the side lane.
 *   dpp_plan_run          issues the launches on two HIP streams (events for fork / join); side == NULL or == main: one stream
 *   dpp_plan_graph_build  builds an explicit hipGraph (hipGraphAddKernelNode + dependency edges: lanes stay parallel
 *                         branches); two_lanes == 0 chains everything in recorded order
 *   dpp_plan_graph_launch replays it on `stream`
 * Device pointers recorded in a plan must stay valid for its lifetime (the caller owns them, as everywhere in this ABI). */
/* Profiling build only (make -C csrc prof): kernels compiled with phase stamps write 16 x uint64 of 100 MHz ticks per workgroup
 * to `buf` (tools/phase_profile.py); in the product build the stamps are compiled out and this only stores the pointer. */
int dpp_prof_set(void* buf);
typedef struct dpp_plan dpp_plan;
int dpp_plan_create(dpp_plan** out);
int dpp_plan_destroy(dpp_plan* plan);
int dpp_plan_record_begin(dpp_plan* plan);
int dpp_plan_record_lane(dpp_plan* plan, int lane);
int dpp_plan_record_end(dpp_plan* plan);
int dpp_plan_fork(dpp_plan* plan);
int dpp_plan_join(dpp_plan* plan);
int dpp_plan_count(const dpp_plan* plan, int* launches, int* forks, int* joins);
int dpp_plan_run(dpp_plan* plan, dpp_stream_t main_stream, dpp_stream_t side_stream);
int dpp_plan_graph_build(dpp_plan* plan, int two_lanes);
int dpp_plan_graph_launch(dpp_plan* plan, dpp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPP_HIP_H */
