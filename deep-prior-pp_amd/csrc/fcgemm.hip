// fcgemm.hip -- the weight-streaming GEMMs of the HiddenLayer that follows the last convolution map (FC1: 16 384 x 1 024 at
// 128x128 input, 65 536 x 1 024 at 256x256), in two precisions: exact f32 (v_mfma_f32_16x16x4_f32) and bf16 operands with
// f32 accumulation (v_mfma_f32_16x16x32_bf16, BASELINE config 5).
//
// Reference arithmetic: HiddenLayer x.W + b, /root/reference/src/net/hiddenlayer.py:136-139, and its T.grad
// (/root/reference/src/trainer/poseregnettrainer.py:110-111): forward C = X.W, data gradient dX = dY.W^T, weight gradient
// dW = X^T.dY -- 4.3 GFLOP each at batch 128 over a 67 MB weight matrix.
//
// Same contract as dpp_gemm (same descriptor, layouts, prologue, bias, split-K partials); what differs is the machinery:
//   * BOTH operands sit in LDS K-contiguous ([row][chunk]): an operand that is MN-contiguous in memory (W in the forward pass,
//     both operands of the weight gradient) is transposed on its way into LDS -- a thread loads 4 (f32) or 8 (bf16) k-rows of
//     a 4-wide column quad and writes four 16-byte k-runs -- so every fragment read is ONE ds_read_b128 (dpp_gemm reads such
//     operands 4 bytes at a time);
//   * the LDS image is double-buffered: chunk c+1 is written while other waves still multiply chunk c, one barrier per chunk;
//   * bf16: operands are rounded (RNE) after the prologue when they are written to LDS, halving LDS bytes per flop twice over
//     (half the bytes, 8 k per read); the matrix pipe is 16x faster, so these GEMMs become pure HBM streams.
// Tile: 128 x 64 per workgroup (4 waves stacked along M, each 32 x 64 = 2 x 4 MFMA tiles), K chunks of 64.
#include <stdlib.h>
#include "dpp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int PREC> struct Prec;
template <> struct Prec<0> {                     // f32
    typedef float elem;
    static constexpr int EPT = 4;                // elements per 16-byte LDS run
    static constexpr int PAD = 4;                // row padding (elements): 16 rows x 16 B land on distinct bank quads
};
template <> struct Prec<1> {                     // bf16
    typedef __bf16 elem;
    static constexpr int EPT = 8;
    static constexpr int PAD = 8;
};

struct FcArgs {
    dpp_gemm_desc d;
    int vecA, vecB;
    int Kper;
};

__device__ __forceinline__ float4 fc_load4(const float* p, int idx0, int limit, bool vec, bool b16 = false) {
    if (b16) {                                   // `p` addresses bf16 elements (DPP_ST_A: the flattened activation map)
        const dpp_bf16* q = reinterpret_cast<const dpp_bf16*>(p);
        if (vec && idx0 + 3 < limit) return dpp_ld4(q);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx0 + 0 < limit) v.x = (float)q[0];
        if (idx0 + 1 < limit) v.y = (float)q[1];
        if (idx0 + 2 < limit) v.z = (float)q[2];
        if (idx0 + 3 < limit) v.w = (float)q[3];
        return v;
    }
    if (vec && idx0 + 3 < limit) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx0 + 0 < limit) v.x = p[0];
    if (idx0 + 1 < limit) v.y = p[1];
    if (idx0 + 2 < limit) v.z = p[2];
    if (idx0 + 3 < limit) v.w = p[3];
    return v;
}

__device__ __forceinline__ float4 fc_act4(float4 v, const dpp_act& a, int c0, int limit) {
    if (a.mode == 0) return v;
    if ((a.cmod & 3) == 0 && c0 + 3 < limit) return dpp_act4(v, a, c0 % a.cmod);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 + 0 < limit) o.x = dpp_act1(v.x, a, (c0 + 0) % a.cmod);
    if (c0 + 1 < limit) o.y = dpp_act1(v.y, a, (c0 + 1) % a.cmod);
    if (c0 + 2 < limit) o.z = dpp_act1(v.z, a, (c0 + 2) % a.cmod);
    if (c0 + 3 < limit) o.w = dpp_act1(v.w, a, (c0 + 3) % a.cmod);
    return o;
}

// 16-byte k-run stores into the LDS image
__device__ __forceinline__ void fc_store_run(float* dst, const float (&v)[4]) { *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void fc_store_run(__bf16* dst, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(dst) = o;
}

// Staging of one operand tile of ROWS (m or n) x KC (k) into its K-contiguous LDS image.
//   KCONT = true : memory is [row][k]   -> slot = (row, k-run): EPT/4 float4 loads, one 16-byte store
//   KCONT = false: memory is [k][row]   -> slot = (k-run, row quad): EPT float4 loads (one per k), four 16-byte stores
template <int PREC, int ROWS, int KC, bool KCONT>
struct Stage {
    typedef typename Prec<PREC>::elem elem;
    static constexpr int EPT = Prec<PREC>::EPT;
    static constexpr int LD = KC + Prec<PREC>::PAD;
    static constexpr int NSLOT = KCONT ? ROWS * (KC / EPT) : (KC / EPT) * (ROWS / 4);
    static constexpr int SLOTS = (NSLOT + DPP_THREADS - 1) / DPP_THREADS;
    static constexpr int NLD = KCONT ? EPT / 4 : EPT;            // float4 loads per slot
    float4 r[SLOTS][NLD];

    // base: operand pointer; ld: leading dimension; map: row map of the NON-contiguous index; r0: first row (m / n) of the tile;
    // rlim: number of rows of the problem; kc .. k_end: the chunk; act: prologue (channel = contiguous index % cmod)
    // sh = 1: the operand's elements are bf16 (whole-quad geometry: every element offset is a multiple of 4 and is halved on the cursor)
    __device__ __forceinline__ void fetch(const float* base, int ld, const dpp_rowmap& map, int r0, int rlim, int kc, int k_end,
                                          const dpp_act& act, bool vec, int sh = 0) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = tid + s * DPP_THREADS;
#pragma unroll
            for (int j = 0; j < NLD; ++j) r[s][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot >= NSLOT) continue;
            if (KCONT) {
                const int row = r0 + slot / (KC / EPT), k = kc + (slot % (KC / EPT)) * EPT;
                if (row < rlim) {
                    const float* p = base + (((size_t)dpp_map_row(map, row) * ld + k) >> sh);
#pragma unroll
                    for (int j = 0; j < NLD; ++j)
                        if (k + 4 * j < k_end) r[s][j] = fc_act4(fc_load4(p + ((4 * j) >> sh), k + 4 * j, k_end, vec, sh != 0), act, k + 4 * j, k_end);
                }
            } else {
                const int q = slot % (ROWS / 4), kr = slot / (ROWS / 4);
                const int row = r0 + q * 4, k = kc + kr * EPT;
                if (row < rlim) {
#pragma unroll
                    for (int j = 0; j < NLD; ++j)
                        if (k + j < k_end)
                            r[s][j] = fc_act4(fc_load4(base + (((size_t)dpp_map_row(map, k + j) * ld + row) >> sh), row, rlim, vec, sh != 0), act, row, rlim);
                }
            }
        }
    }

    __device__ __forceinline__ void commit(elem* img) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int slot = tid + s * DPP_THREADS;
            if (slot >= NSLOT) continue;
            if (KCONT) {
                const int row = slot / (KC / EPT), k = (slot % (KC / EPT)) * EPT;
                float v[EPT];
#pragma unroll
                for (int j = 0; j < NLD; ++j) { v[4 * j] = r[s][j].x; v[4 * j + 1] = r[s][j].y; v[4 * j + 2] = r[s][j].z; v[4 * j + 3] = r[s][j].w; }
                fc_store_run(img + row * LD + k, v);
            } else {
                const int q = slot % (ROWS / 4), kr = slot / (ROWS / 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v[EPT];
#pragma unroll
                    for (int j = 0; j < EPT; ++j) v[j] = dpp_f4_get(r[s][j], e);
                    fc_store_run(img + (q * 4 + e) * LD + kr * EPT, v);
                }
            }
        }
    }
};

template <int PREC, int RM, int CN, int KC, bool AKC, bool BKC>
__global__ __launch_bounds__(DPP_THREADS) void fc_gemm_kernel(FcArgs ga) {
    dpp_kernarg_warm<sizeof(FcArgs)>();
    const dpp_gemm_desc& d = ga.d;
    typedef typename Prec<PREC>::elem elem;
    constexpr int EPT = Prec<PREC>::EPT;
    constexpr int BM = 64 * RM, BN = 16 * CN;
    typedef Stage<PREC, BM, KC, AKC> SA;
    typedef Stage<PREC, BN, KC, BKC> SB;
    constexpr int LD = KC + Prec<PREC>::PAD;
    constexpr int SZA = BM * LD, SZB = BN * LD;                                  // elements per buffer
    constexpr int OPB = 2 * (SZA + SZB) * (int)sizeof(elem);                     // both buffers, bytes
    constexpr int EPB = (BM * (BN + 4) + 16 * BN) * 4;                           // wide epilogue image, bytes
    HIP_DYNAMIC_SHARED(float4, smem4)
    static_assert(OPB % 16 == 0, "buffer alignment");
    elem* const img = reinterpret_cast<elem*>(smem4);
    (void)EPB;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    const int M = d.M, N = d.N;
    const int k_begin = blockIdx.z * ga.Kper;
    const int k_end = (k_begin + ga.Kper < d.K) ? (k_begin + ga.Kper) : d.K;
    const int nchunks = (k_end > k_begin) ? (k_end - k_begin + KC - 1) / KC : 0;

    f32x4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dpp_wide_coef wco;
    wco.load<BN>(col0, N, d.splitk == 1 ? d.bias : nullptr, d.epi, d.splitk > 1 ? d.partial : d.C);

    SA sa;
    SB sb;
    const dpp_rowmap ident = {1, 0, 0, 0, 0};
    auto fetch = [&](int c) __attribute__((always_inline)) {
        const int kc = k_begin + c * KC;
        sa.fetch(d.A, d.lda, d.mapA, row0, M, kc, k_end, d.actA, ga.vecA, (d.store & DPP_ST_A) ? 1 : 0);
        sb.fetch(d.B, d.ldb, BKC ? ident : d.mapB, col0, N, kc, k_end, d.actB, ga.vecB);
    };
    if (nchunks > 0) fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        elem* As = img + (c & 1) * (SZA + SZB);
        elem* Bs = As + SZA;
        sa.commit(As);
        sb.commit(Bs);
        if (c + 1 < nchunks) fetch(c + 1);            // in flight under this chunk's MFMAs
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += 4 * EPT) {    // one 16-byte run per lane covers k0 + kq*EPT .. + EPT-1
            if constexpr (PREC == 0) {
                float af[RM][4], bf[CN][4];
#pragma unroll
                for (int rt = 0; rt < RM; ++rt) {
                    const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(As) + (wave * (BM / 4) + rt * 16 + l15) * LD + k0 + kq * 4);
                    af[rt][0] = v.x; af[rt][1] = v.y; af[rt][2] = v.z; af[rt][3] = v.w;
                }
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Bs) + (ct * 16 + l15) * LD + k0 + kq * 4);
                    bf[ct][0] = v.x; bf[ct][1] = v.y; bf[ct][2] = v.z; bf[ct][3] = v.w;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CN; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rt][t], bf[ct][t], acc[rt][ct], 0, 0, 0);
            } else {
                bf16x8 af[RM], bf[CN];
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
                    af[rt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(As) + (wave * (BM / 4) + rt * 16 + l15) * LD + k0 + kq * 8);
#pragma unroll
                for (int ct = 0; ct < CN; ++ct)
                    bf[ct] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(Bs) + (ct * 16 + l15) * LD + k0 + kq * 8);
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CN; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt], bf[ct], acc[rt][ct], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // the operand images are dead: the epilogue reuses the memory

    const int nvalid = (M - row0 < BM) ? (M - row0) : BM;
    float* smem = reinterpret_cast<float*>(smem4);
    if (d.splitk > 1) {
        dpp_epilogue ep0 = {};
        float* P = d.partial + (size_t)blockIdx.z * M * N;
        dpp_epilogue_wide<RM, CN, 4, 1, BM, BN>(acc, smem, col0, N, wco, nullptr, P, ep0, nvalid, wave, 0, l15, kq, [&](int rl) {
            const int row = row0 + rl;
            return row < M ? (long)row * N : -1L;
        });
    } else {
        dpp_epilogue_wide<RM, CN, 4, 1, BM, BN>(acc, smem, col0, N, wco, d.residual, d.C, d.epi, nvalid, wave, 0, l15, kq, [&](int rl) {
            const int row = row0 + rl;
            return row < M ? (long)dpp_map_row(d.mapC, row) * d.ldc : -1L;
        });
    }
}


// ---- f32, three LDS stages (fc_stream_kernel) ----------------------------------------------------------------------------
// The same GEMMs when the whole tile is inside the problem (M % 128 == 0, N % BN == 0, K slices of whole 32-deep chunks,
// 16-byte aligned operands): tile 128 x BN (BN = 128 or 64), 2 x 2 waves of 64 x BN/2 (4 x BN/32 MFMA tiles each), and a
// three-deep pipeline -- while the waves multiply chunk c out of LDS buffer c % 3, chunk c+1 has just been written to the next
// buffer and the global loads of chunk c+2 are in flight, with ONE barrier per chunk.  At batch 128 the three FC1 GEMMs are 4.3
// GFLOP each on a 67 MB weight matrix: 27 us at the f32 matrix-core peak, 10 us of HBM.  dpp_gemm's 64 x 32 / 128 x 64 tiles move
// 390 / 200 MB through the CUs for them and wait on every chunk (102 / 104 us); here a workgroup of the forward pass owns
// 128 x 128 outputs of one 512-deep K slice (131 MB through the CUs in total, 256 workgroups = one per CU), staging is one
// unchecked 16-byte load per slot, and a chunk's 128..256 MFMAs per wave cover the next chunk's loads and LDS writes.
struct FsArgs {
    dpp_gemm_desc d;
    int Kper;
};

// (DPP_WAVES_PER_EU: 83-110 KB of dynamic LDS allow one workgroup per CU, which the compiler cannot see -- aiming at four waves per
// SIMD it kept the kernel under 128 VGPRs by spilling the chunk in flight (ra[]) to scratch memory and back in every iteration)
// PREC = 1 (BASELINE config 5): the same pipeline with bf16 LDS images -- operands rounded RNE after the prologue when the chunk is
// committed, one v_mfma_f32_16x16x32_bf16 per accumulator tile and 32-deep chunk (16 instead of 128 MFMAs per wave and chunk: the
// kernel is then bound by the weight stream alone).
template <int PREC, int BN, bool AKC, bool BKC>
__global__ __launch_bounds__(DPP_THREADS) DPP_WAVES_PER_EU(1, 2) void fc_stream_kernel(FsArgs ga) {
    dpp_kernarg_warm<sizeof(FsArgs)>();
    const dpp_gemm_desc& d = ga.d;
    typedef typename Prec<PREC>::elem elem;
    constexpr int BM = 128, KC = 32, LD = KC + Prec<PREC>::PAD, NST = 3, WM = 2, WN = 2;
    constexpr int RM = BM / (16 * WM), CN = BN / (16 * WN);
    constexpr int SZA = BM * LD, SZB = BN * LD, SZ = SZA + SZB;
    HIP_DYNAMIC_SHARED(float4, smem4)
    elem* const img = reinterpret_cast<elem*>(smem4);
    // one k-run of four values into the image: 16 bytes (f32) or 8 bytes (bf16, rounded to nearest even)
    auto put4 = [](elem* dst, float x, float y, float z, float w) __attribute__((always_inline)) {
        if constexpr (PREC == 0) *reinterpret_cast<float4*>(dst) = make_float4(x, y, z, w);
        else {
            bf16x4 o;
            o[0] = (__bf16)x; o[1] = (__bf16)y; o[2] = (__bf16)z; o[3] = (__bf16)w;
            *reinterpret_cast<bf16x4*>(dst) = o;
        }
    };
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    const int M = d.M, N = d.N;
    const int k_begin = blockIdx.z * ga.Kper;
    const int nchunks = ga.Kper / KC;

    // ---- staging slots -------------------------------------------------------------------------------------------------
    // K-contiguous operand of R rows: thread (r = tid / 8, quad = tid % 8) owns rows r, r + 32, ...: one 16-byte load and one
    // 16-byte LDS store per slot.  MN-contiguous operand [k][R]: thread (kg = tid / (R/4), cq = tid % (R/4)) loads the four k-rows
    // 4kg .. 4kg+3 of column quad cq and stores four k-runs (one per column): the transposition happens in registers.
    constexpr int SA = AKC ? BM / 32 : 4, SB = BKC ? BN / 32 : 4;
    const int shA = (d.store & DPP_ST_A) ? 1 : 0;            // bf16-stored operand A (the flattened activation map): halved cursors, 8-byte loads
    const float* pa[SA];
    const float* pb[SB];
    int la, lb;                                   // LDS offset of slot 0 (floats)
    bool bvalid = true;
    if (AKC) {
        const int r = tid >> 3, quad = (tid & 7) * 4;
#pragma unroll
        for (int s = 0; s < SA; ++s) pa[s] = d.A + (((size_t)(row0 + r + 32 * s) * d.lda + k_begin + quad) >> shA);
        la = r * LD + quad;
    } else {
        const int cq = tid % (BM / 4), kg = tid / (BM / 4);
#pragma unroll
        for (int s = 0; s < SA; ++s) pa[s] = d.A + (((size_t)(k_begin + 4 * kg + s) * d.lda + row0 + 4 * cq) >> shA);
        la = (4 * cq) * LD + 4 * kg;
    }
    if (BKC) {
        const int r = tid >> 3, quad = (tid & 7) * 4;
#pragma unroll
        for (int s = 0; s < SB; ++s) pb[s] = d.B + (size_t)(col0 + r + 32 * s) * d.ldb + k_begin + quad;
        lb = r * LD + quad;
    } else {
        const int cq = tid % (BN / 4), kg = tid / (BN / 4);
        bvalid = kg < KC / 4;                     // BN = 64: the chunk has 128 slots, half the threads idle
#pragma unroll
        for (int s = 0; s < SB; ++s) pb[s] = d.B + (size_t)(k_begin + 4 * (bvalid ? kg : 0) + s) * d.ldb + col0 + 4 * cq;
        lb = (4 * cq) * LD + 4 * kg;
    }
    // operand-A prologue (BatchNorm + ReLU of the map that was flattened): channel = contiguous index % cmod
    const int modeA = d.actA.mode;
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), sc = mu, be = mu;
    int cidx = 0;
    if (modeA & 2) {
        cidx = AKC ? (k_begin + (tid & 7) * 4) % d.actA.cmod : (row0 + 4 * (tid % (BM / 4))) % d.actA.cmod;
        if (!AKC) {
            mu = *reinterpret_cast<const float4*>(d.actA.mean + cidx);
            sc = *reinterpret_cast<const float4*>(d.actA.scale + cidx);
            be = *reinterpret_cast<const float4*>(d.actA.beta + cidx);
        }
    }
    // rb as a native vector type: float4 is a struct, and a value that is only copied (global -> register -> LDS, no arithmetic) was
    // lowered to memcpy intrinsics through a private-memory array -- scratch stores and loads in every K iteration, and a drain of
    // all outstanding loads right where they were issued
    float4 ra[SA];
    f32x4 rb[SB];
    const size_t stepA = AKC ? (size_t)KC : (size_t)KC * d.lda, stepB = BKC ? (size_t)KC : (size_t)KC * d.ldb;
    auto fetch = [&](int c) __attribute__((always_inline)) {
        if (AKC && (modeA & 2)) {
            mu = *reinterpret_cast<const float4*>(d.actA.mean + cidx);
            sc = *reinterpret_cast<const float4*>(d.actA.scale + cidx);
            be = *reinterpret_cast<const float4*>(d.actA.beta + cidx);
            cidx += KC;
            if (cidx >= d.actA.cmod) cidx -= d.actA.cmod;
        }
        if (shA) {
#pragma unroll
            for (int s = 0; s < SA; ++s) ra[s] = dpp_raw8(pa[s] + ((c * stepA) >> 1));
        } else {
#pragma unroll
            for (int s = 0; s < SA; ++s) ra[s] = *reinterpret_cast<const float4*>(pa[s] + c * stepA);
        }
        if (BKC || bvalid) {
#pragma unroll
            for (int s = 0; s < SB; ++s) rb[s] = *reinterpret_cast<const f32x4*>(pb[s] + c * stepB);
        }
    };
    auto act = [&](float4 v) __attribute__((always_inline)) {
        if (modeA & 2) {
            v.x = dpp_fma(v.x - mu.x, sc.x, be.x); v.y = dpp_fma(v.y - mu.y, sc.y, be.y);
            v.z = dpp_fma(v.z - mu.z, sc.z, be.z); v.w = dpp_fma(v.w - mu.w, sc.w, be.w);
        }
        if (modeA & 1) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        return v;
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        elem* As = img + buf * SZ;
        elem* Bs = As + SZA;
        if constexpr (AKC) {
#pragma unroll
            for (int s = 0; s < SA; ++s) {
                const float4 w = shA ? dpp_widen4(ra[s]) : ra[s];
                const float4 v = modeA ? act(w) : w;
                put4(&As[la + 32 * s * LD], v.x, v.y, v.z, v.w);
            }
        } else {
            float4 v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { const float4 w = shA ? dpp_widen4(ra[s]) : ra[s]; v[s] = modeA ? act(w) : w; }
            put4(&As[la + 0 * LD], v[0].x, v[1].x, v[2].x, v[3].x);
            put4(&As[la + 1 * LD], v[0].y, v[1].y, v[2].y, v[3].y);
            put4(&As[la + 2 * LD], v[0].z, v[1].z, v[2].z, v[3].z);
            put4(&As[la + 3 * LD], v[0].w, v[1].w, v[2].w, v[3].w);
        }
        if constexpr (BKC) {
#pragma unroll
            for (int s = 0; s < SB; ++s) {
                if constexpr (PREC == 0) *reinterpret_cast<f32x4*>(&Bs[lb + 32 * s * LD]) = rb[s];
                else put4(&Bs[lb + 32 * s * LD], rb[s][0], rb[s][1], rb[s][2], rb[s][3]);
            }
        } else if (bvalid) {
            put4(&Bs[lb + 0 * LD], rb[0][0], rb[1][0], rb[2][0], rb[3][0]);
            put4(&Bs[lb + 1 * LD], rb[0][1], rb[1][1], rb[2][1], rb[3][1]);
            put4(&Bs[lb + 2 * LD], rb[0][2], rb[1][2], rb[2][2], rb[3][2]);
            put4(&Bs[lb + 3 * LD], rb[0][3], rb[1][3], rb[2][3], rb[3][3]);
        }
    };

    f32x4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dpp_wide_coef wco;
    wco.load<BN>(col0, N, d.splitk == 1 ? d.bias : nullptr, d.epi, d.splitk > 1 ? d.partial : d.C);

    fetch(0);
    commit(0);
    if (nchunks > 1) fetch(1);
    __syncthreads();
    int cur = 0;
    constexpr int EPT = Prec<PREC>::EPT;      // k-values of a lane's 16-byte fragment read
    const int aoff = (wm * (BM / WM) + l15) * LD + kq * EPT, boff = (wn * (BN / WN) + l15) * LD + kq * EPT;
    for (int c = 0; c < nchunks; ++c) {
        const int nxt = cur == NST - 1 ? 0 : cur + 1;
        if (c + 1 < nchunks) commit(nxt);                 // its loads were issued one whole MFMA block ago
        if (c + 2 < nchunks) fetch(c + 2);                // in flight under this chunk's MFMAs
        const elem* As = img + cur * SZ;
        const elem* Bs = As + SZA;
        if constexpr (PREC == 0) {
#pragma unroll
            for (int k0 = 0; k0 < KC; k0 += 16) {
                float4 af[RM], bf[CN];
#pragma unroll
                for (int rt = 0; rt < RM; ++rt) af[rt] = *reinterpret_cast<const float4*>(&As[aoff + rt * 16 * LD + k0]);
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) bf[ct] = *reinterpret_cast<const float4*>(&Bs[boff + ct * 16 * LD + k0]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CN; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(af[rt], t), dpp_f4_get(bf[ct], t), acc[rt][ct], 0, 0, 0);
            }
        } else {
            bf16x8 af[RM], bf[CN];                        // lane (i, kq) holds k = 8 kq .. 8 kq + 7 of row i: the whole chunk in one step
#pragma unroll
            for (int rt = 0; rt < RM; ++rt) af[rt] = *reinterpret_cast<const bf16x8*>(&As[aoff + rt * 16 * LD]);
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) bf[ct] = *reinterpret_cast<const bf16x8*>(&Bs[boff + ct * 16 * LD]);
#pragma unroll
            for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                for (int ct = 0; ct < CN; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt], bf[ct], acc[rt][ct], 0, 0, 0);
        }
        __syncthreads();
        cur = nxt;
    }

    float* smem = reinterpret_cast<float*>(smem4);
    if (d.splitk > 1) {
        dpp_epilogue ep0 = {};
        float* P = d.partial + (size_t)blockIdx.z * M * N;
        dpp_epilogue_wide<RM, CN, WM, WN, BM, BN>(acc, smem, col0, N, wco, nullptr, P, ep0, BM, wm, wn, l15, kq,
                                                  [&](int rl) { return (long)(row0 + rl) * N; });
    } else {
        dpp_epilogue_wide<RM, CN, WM, WN, BM, BN>(acc, smem, col0, N, wco, d.residual, d.C, d.epi, BM, wm, wn, l15, kq,
                                                  [&](int rl) { return (long)dpp_map_row(d.mapC, row0 + rl) * d.ldc; });
    }
}

template <int PREC, int BN, bool AKC, bool BKC>
int fs_launch(const FsArgs& ga, dim3 grid, hipStream_t st) {
    constexpr size_t opb = 3 * (size_t)(128 + BN) * (32 + Prec<PREC>::PAD) * sizeof(typename Prec<PREC>::elem);
    constexpr size_t epb = ((size_t)128 * (BN + 4) + 16 * BN) * 4;
    const size_t lds = opb > epb ? opb : epb;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_stream_kernel<PREC, BN, AKC, BKC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DPP_LAUNCH((fc_stream_kernel<PREC, BN, AKC, BKC>), grid, dim3(DPP_THREADS), lds, st, ga);
    return dpp_launch_status();
}

template <int PREC, int BN>
int fs_dispatch(const FsArgs& ga, dim3 grid, hipStream_t st) {
    const dpp_gemm_desc& d = ga.d;
    if (d.a_kc && d.b_kc) return fs_launch<PREC, BN, true, true>(ga, grid, st);
    if (d.a_kc && !d.b_kc) return fs_launch<PREC, BN, true, false>(ga, grid, st);
    if (!d.a_kc && !d.b_kc) return fs_launch<PREC, BN, false, false>(ga, grid, st);
    return fs_launch<PREC, BN, false, true>(ga, grid, st);
}

// whether fc_stream_kernel takes the call; bn receives its column tile
bool fs_accepts(const dpp_gemm_desc& d, int& bn) {
    static const bool on = []() { const char* e = getenv("DPP_FC_STREAM3"); return !(e && e[0] == '0'); }();
    if (!on) return false;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (d.M % 128 || d.K % d.splitk || (d.K / d.splitk) % 32) return false;
    if (!al(d.A) || !al(d.B) || (d.lda & 3) || (d.ldb & 3)) return false;
    if (d.mapA.s != 1 || d.mapB.s != 1 || d.actB.mode != 0) return false;
    if (d.actA.mode & 2) {
        if ((d.actA.cmod & 3) || d.actA.cmod < 32) return false;
        if (!al(d.actA.mean) || !al(d.actA.scale) || !al(d.actA.beta)) return false;
    }
    // 128-wide column tiles when they still give every CU a workgroup
    bn = (d.N % 128 == 0 && (long)(d.M / 128) * (d.N / 128) * d.splitk >= 256) ? 128 : 64;
    return d.N % bn == 0;
}

template <int PREC, int RM, int CN, int KC, bool AKC, bool BKC>
int fc_launch(const FcArgs& ga, dim3 grid, hipStream_t st) {
    typedef typename Prec<PREC>::elem elem;
    constexpr int BM = 64 * RM, BN = 16 * CN, LD = KC + Prec<PREC>::PAD;
    constexpr size_t opb = 2 * (size_t)(BM + BN) * LD * sizeof(elem);
    constexpr size_t epb = ((size_t)BM * (BN + 4) + 16 * BN) * 4;
    const size_t lds = opb > epb ? opb : epb;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_kernel<PREC, RM, CN, KC, AKC, BKC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DPP_LAUNCH((fc_gemm_kernel<PREC, RM, CN, KC, AKC, BKC>), grid, dim3(DPP_THREADS), lds, st, ga);
    return dpp_launch_status();
}

template <int PREC, int KC>
int fc_dispatch(const FcArgs& ga, dim3 grid, hipStream_t st) {
    const dpp_gemm_desc& d = ga.d;
    if (d.a_kc && d.b_kc) return fc_launch<PREC, 2, 4, KC, true, true>(ga, grid, st);
    if (d.a_kc && !d.b_kc) return fc_launch<PREC, 2, 4, KC, true, false>(ga, grid, st);
    if (!d.a_kc && !d.b_kc) return fc_launch<PREC, 2, 4, KC, false, false>(ga, grid, st);
    return fc_launch<PREC, 2, 4, KC, false, true>(ga, grid, st);
}

}  // namespace

// dpp_fc_gemm: dpp_gemm's contract (same descriptor) on the weight-streaming kernel above.  precision: 0 = f32 (exact, the
// result equals dpp_gemm's up to summation order), 1 = bf16 operands / f32 accumulation.  Tile 128 x 64, d.bm / d.bn / d.wm /
// d.variant are ignored; kchunk = 32 or 64 (0 = 64).  Supported: actA / actB modes 0-3, bias, residual, mapA / mapB / mapC,
// split-K partials; the fused statistics / BatchNorm-backward epilogues of dpp_gemm are not (the FC layers have no BatchNorm).
extern "C" int dpp_fc_gemm(const dpp_gemm_desc* dp, int precision, int kchunk, dpp_stream_t stream) {
    if (!dp) return DPP_E_BADARG;
    const dpp_gemm_desc& d = *dp;
    if (!d.A || !d.B || d.M < 1 || d.N < 1 || d.K < 1 || d.splitk < 1) return DPP_E_BADARG;
    if (d.splitk > 1 ? !d.partial : !d.C) return DPP_E_BADARG;
    if (precision != 0 && precision != 1) return DPP_E_BADARG;
    if (d.actA.mode > 3 || d.actB.mode > 3 || d.epi.stats || d.epi.bn_x) return DPP_E_UNSUPPORTED;
    if (d.N % 4) return DPP_E_UNSUPPORTED;                                  // the 16-byte epilogue
    if (d.store & ~DPP_ST_A) return DPP_E_UNSUPPORTED;                      // only operand A (the flattened activation map) may be bf16-stored
    if ((d.store & DPP_ST_A) && ((reinterpret_cast<uintptr_t>(d.A) & 15) || (d.lda & 3))) return DPP_E_UNSUPPORTED;
    const int ldc = d.splitk > 1 ? d.N : d.ldc;
    if (ldc % 4 || (reinterpret_cast<uintptr_t>(d.splitk > 1 ? d.partial : d.C) & 15)) return DPP_E_UNSUPPORTED;
    if (d.residual && (reinterpret_cast<uintptr_t>(d.residual) & 15)) return DPP_E_UNSUPPORTED;
    if (d.splitk == 1 && d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15)) return DPP_E_UNSUPPORTED;     // one 16-byte load per column quad
    const int KC = kchunk == 0 ? 64 : kchunk;
    if (KC != 32 && KC != 64) return DPP_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int bn3 = 0;
    static const bool fs_bf16 = []() { const char* e = getenv("DPP_FC_STREAM3_BF16"); return !(e && e[0] == '0'); }();
    if ((precision == 0 || fs_bf16) && fs_accepts(d, bn3)) {
        FsArgs fa;
        fa.d = d;
        fa.Kper = d.K / d.splitk;
        dim3 g3(d.M / 128, d.N / bn3, d.splitk);
        if (precision == 0) return bn3 == 128 ? fs_dispatch<0, 128>(fa, g3, st) : fs_dispatch<0, 64>(fa, g3, st);
        return bn3 == 128 ? fs_dispatch<1, 128>(fa, g3, st) : fs_dispatch<1, 64>(fa, g3, st);
    }
    FcArgs ga;
    ga.d = d;
    auto aligned = [](const float* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; };
    ga.vecA = aligned(d.A, d.lda);
    ga.vecB = aligned(d.B, d.ldb);
    int per = (d.K + d.splitk - 1) / d.splitk;
    per = (per + KC - 1) / KC * KC;
    ga.Kper = per;
    dim3 grid(dpp_cdiv(d.M, 128), dpp_cdiv(d.N, 64), d.splitk);
    if (precision == 0) return KC == 64 ? fc_dispatch<0, 64>(ga, grid, st) : fc_dispatch<0, 32>(ga, grid, st);
    return KC == 64 ? fc_dispatch<1, 64>(ga, grid, st) : fc_dispatch<1, 32>(ga, grid, st);
}
