// gemm.hip -- generic f32 MFMA GEMM for gfx950 (see dpp_gemm in include/dpp_hip.h).
//
// One kernel family covers every GEMM-shaped op of the hot path: 1x1 ConvLayer forward / data
// gradient / filter gradient (pixel rows, optional stride-2 row maps) and HiddenLayer forward / data
// gradient / weight gradient, with the pre-activation BatchNorm+ReLU fused into the operand staging and
// bias + residual fused into the epilogue.
//
// Structure (CDNA4): 256 threads = 4 wave64; block tile BM x BN, K walked in chunks of 16 through LDS;
// each wave owns RM x CN tiles of 16x16 and issues v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma chain).
//   * K-contiguous operand  -> LDS image [row][16+4]; a lane fetches its 4 k-values with ONE ds_read_b128
//     (lane (i, kq) owns k = 4*kq + t, t = 0..3; the +4 pad makes the 16 rows of a lane group hit 16
//     distinct 16-byte slots of the 256-byte bank row);
//   * MN-contiguous operand -> LDS image [k][rows+4]; a lane fetches with ds_read_b32 (rows+4 == 16 mod 32
//     dwords apart for kq = 0/1, so the two halves of a 32-lane group use disjoint banks).
// Global loads are float4 (16 B/lane) along the contiguous dimension whenever alignment allows.
// f32 MFMA runs at the f32 vector rate (157 TF), so these layers are HBM/L2-bound for the small-channel
// stages; the kernel keeps LDS small (<= 15 KB) to run 8 blocks per CU and hide load latency with TLP.
#include "dpp_common.h"

namespace {


struct GemmArgs {
    dpp_gemm_desc d;
    int vecA, vecB;   // float4 loads legal for the operand
    int Kper;         // K-slice length per blockIdx.z (multiple of the chunk depth)
    int bk;           // chunk depth for K-contiguous A: 16 or 32 (both-MN-contiguous layout always uses 64)
};

// Load 4 consecutive floats p[0..3] where element e is valid iff (idx0 + e) < limit.
__device__ __forceinline__ float4 load4(const float* p, int idx0, int limit, bool vec) {
    if (vec && idx0 + 3 < limit) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx0 + 0 < limit) v.x = p[0];
    if (idx0 + 1 < limit) v.y = p[1];
    if (idx0 + 2 < limit) v.z = p[2];
    if (idx0 + 3 < limit) v.w = p[3];
    return v;
}

// Prologue on 4 consecutive elements along the contiguous dim starting at contiguous index c0;
// elements at or beyond `limit` are forced to zero AFTER the activation.
__device__ __forceinline__ float4 act4_masked(float4 v, const dpp_act& a, int c0, int limit) {
    if (a.mode == 0) return v;
    if ((a.cmod & 3) == 0 && c0 + 3 < limit) return dpp_act4(v, a, c0 % a.cmod);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 + 0 < limit) o.x = dpp_act1(v.x, a, (c0 + 0) % a.cmod);
    if (c0 + 1 < limit) o.y = dpp_act1(v.y, a, (c0 + 1) % a.cmod);
    if (c0 + 2 < limit) o.z = dpp_act1(v.z, a, (c0 + 2) % a.cmod);
    if (c0 + 3 < limit) o.w = dpp_act1(v.w, a, (c0 + 3) % a.cmod);
    return o;
}

// Global -> register fetch of one float4 staging slot (with the operand prologue applied), and its LDS address.
// K-contiguous operand: slot = (row r, k-quad c4); MN-contiguous operand: slot = (k row rk, mn-quad c4).
template <int ROWS, int BKT, bool KC>
struct Stager {
    static constexpr int QK = BKT / 4;                       // float4 per row (KC)
    static constexpr int SLOTS = (ROWS * QK + DPP_THREADS - 1) / DPP_THREADS;
    static constexpr int LD = KC ? (BKT + 4) : (ROWS + 4);
};

template <int BM, int BN, int WM, int BKT, bool AKC, bool BKC>
__global__ __launch_bounds__(DPP_THREADS) void gemm_kernel(GemmArgs ga) {
    const dpp_gemm_desc& d = ga.d;
    constexpr int WN = 4 / WM;
    constexpr int RM = BM / (16 * WM);
    constexpr int CN = BN / (16 * WN);
    using SA = Stager<BM, BKT, AKC>;
    using SB = Stager<BN, BKT, BKC>;
    constexpr int LDA_ = SA::LD, LDB_ = SB::LD;
    constexpr int KL = BKT / 4;                              // k-values owned by one lane per chunk: kq*KL + e
    __shared__ __attribute__((aligned(16))) float As[AKC ? BM * LDA_ : BKT * LDA_];
    __shared__ __attribute__((aligned(16))) float Bs[BKC ? BN * LDB_ : BKT * LDB_];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    const int M = d.M, N = d.N;
    const int k_begin = blockIdx.z * ga.Kper;
    const int k_end = (k_begin + ga.Kper < d.K) ? (k_begin + ga.Kper) : d.K;

    // ---- per-thread staging slots: base pointers that do not depend on the chunk -----------------------
    const float* a_base[SA::SLOTS];
    const float* b_base[SB::SLOTS];
#pragma unroll
    for (int s = 0; s < SA::SLOTS; ++s) {
        int slot = tid + s * DPP_THREADS;
        a_base[s] = nullptr;
        if (AKC) {
            int gi = row0 + slot / SA::QK;
            if (slot < BM * SA::QK && gi < M) a_base[s] = d.A + (size_t)dpp_map_row(d.mapA, gi) * d.lda;
        } else {
            int gi = row0 + (slot % (BM / 4)) * 4;
            if (slot < BKT * (BM / 4) && gi < M) a_base[s] = d.A + gi;
        }
    }
#pragma unroll
    for (int s = 0; s < SB::SLOTS; ++s) {
        int slot = tid + s * DPP_THREADS;
        b_base[s] = nullptr;
        if (BKC) {
            int gj = col0 + slot / SB::QK;
            if (slot < BN * SB::QK && gj < N) b_base[s] = d.B + (size_t)gj * d.ldb;
        } else {
            int gj = col0 + (slot % (BN / 4)) * 4;
            if (slot < BKT * (BN / 4) && gj < N) b_base[s] = d.B + gj;
        }
    }

    float4 ra[SA::SLOTS], rb[SB::SLOTS];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int s = 0; s < SA::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_base[s] != nullptr) {
                if (AKC) {
                    int k = kc + (slot % SA::QK) * 4;
                    if (k < k_end) v = act4_masked(load4(a_base[s] + k, k, k_end, ga.vecA), d.actA, k, k_end);
                } else {
                    int k = kc + slot / (BM / 4), gi = row0 + (slot % (BM / 4)) * 4;
                    if (k < k_end)
                        v = act4_masked(load4(a_base[s] + (size_t)dpp_map_row(d.mapA, k) * d.lda, gi, M, ga.vecA), d.actA, gi, M);
                }
            }
            ra[s] = v;
        }
#pragma unroll
        for (int s = 0; s < SB::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_base[s] != nullptr) {
                if (BKC) {
                    int k = kc + (slot % SB::QK) * 4;
                    if (k < k_end) v = act4_masked(load4(b_base[s] + k, k, k_end, ga.vecB), d.actB, k, k_end);
                } else {
                    int k = kc + slot / (BN / 4), gj = col0 + (slot % (BN / 4)) * 4;
                    if (k < k_end)
                        v = act4_masked(load4(b_base[s] + (size_t)dpp_map_row(d.mapB, k) * d.ldb, gj, N, ga.vecB), d.actB, gj, N);
                }
            }
            rb[s] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < SA::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            if (AKC) { if (slot < BM * SA::QK) *reinterpret_cast<float4*>(&As[(slot / SA::QK) * LDA_ + (slot % SA::QK) * 4]) = ra[s]; }
            else { if (slot < BKT * (BM / 4)) *reinterpret_cast<float4*>(&As[(slot / (BM / 4)) * LDA_ + (slot % (BM / 4)) * 4]) = ra[s]; }
        }
#pragma unroll
        for (int s = 0; s < SB::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            if (BKC) { if (slot < BN * SB::QK) *reinterpret_cast<float4*>(&Bs[(slot / SB::QK) * LDB_ + (slot % SB::QK) * 4]) = rb[s]; }
            else { if (slot < BKT * (BN / 4)) *reinterpret_cast<float4*>(&Bs[(slot / (BN / 4)) * LDB_ + (slot % (BN / 4)) * 4]) = rb[s]; }
        }
    };

    f32x4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline: the global loads of chunk k+1 are issued right after the barrier that publishes chunk k and
    // stay in flight under chunk k's MFMAs; their LDS write happens after the next barrier.
    if (k_begin < k_end) fetch(k_begin);
    for (int kc = k_begin; kc < k_end; kc += BKT) {
        commit();
        __syncthreads();
        if (kc + BKT < k_end) fetch(kc + BKT);
#pragma unroll
        for (int e4 = 0; e4 < KL; e4 += 4) {
            float af[RM][4], bf[CN][4];
#pragma unroll
            for (int rt = 0; rt < RM; ++rt) {
                int r = wm * (BM / WM) + rt * 16 + l15;
                if (AKC) {
                    float4 v = *reinterpret_cast<const float4*>(&As[r * LDA_ + kq * KL + e4]);
                    af[rt][0] = v.x; af[rt][1] = v.y; af[rt][2] = v.z; af[rt][3] = v.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) af[rt][t] = As[(kq * KL + e4 + t) * LDA_ + r];
                }
            }
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                int c = wn * (BN / WN) + ct * 16 + l15;
                if (BKC) {
                    float4 v = *reinterpret_cast<const float4*>(&Bs[c * LDB_ + kq * KL + e4]);
                    bf[ct][0] = v.x; bf[ct][1] = v.y; bf[ct][2] = v.z; bf[ct][3] = v.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) bf[ct][t] = Bs[(kq * KL + e4 + t) * LDB_ + c];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CN; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rt][t], bf[ct][t], acc[rt][ct], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: D layout col = lane&15, row = (lane>>4)*4 + r --------------------------------
#pragma unroll
    for (int rt = 0; rt < RM; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = row0 + wm * (BM / WM) + rt * 16 + kq * 4 + r;
            if (row >= M) continue;
            if (d.splitk > 1) {
                float* prow = d.partial + ((size_t)blockIdx.z * M + row) * N;
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    int col = col0 + wn * (BN / WN) + ct * 16 + l15;
                    if (col < N) prow[col] = acc[rt][ct][r];
                }
            } else {
                size_t o = (size_t)dpp_map_row(d.mapC, row) * d.ldc;
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    int col = col0 + wn * (BN / WN) + ct * 16 + l15;
                    if (col < N) {
                        float v = acc[rt][ct][r];
                        if (d.bias) v += d.bias[col];
                        if (d.residual) v += d.residual[o + col];
                        d.C[o + col] = v;
                    }
                }
            }
        }
    }
}

// out[i] = sum_z partial[z][i] (+ bias).  Threads are laid out as CB columns x ZL z-lanes: lane zl sums z = zl, zl+ZL, ...
// (a fixed order), then the ZL partial sums are combined through LDS in a fixed order: deterministic, and parallel in z
// when there are many slices of a small output (filter gradients: hundreds of slices of a few thousand elements).
__global__ __launch_bounds__(DPP_THREADS) void reduce_partials_kernel(const float* __restrict__ partial, int nz, int n,
                                                                      const float* __restrict__ bias, int nbias,
                                                                      float* __restrict__ out, int ZL) {
    __shared__ float red[DPP_THREADS];
    const int CB = DPP_THREADS / ZL;
    const int col = threadIdx.x % CB, zl = threadIdx.x / CB;
    for (int i0 = blockIdx.x * CB; i0 < n; i0 += gridDim.x * CB) {
        int i = i0 + col;
        float s = 0.0f;
        if (i < n)
            for (int z = zl; z < nz; z += ZL) s += partial[(size_t)z * n + i];
        if (ZL > 1) {
            red[threadIdx.x] = s;
            __syncthreads();
            if (zl == 0) {
                for (int j = 1; j < ZL; ++j) s += red[j * CB + col];
            }
        }
        if (zl == 0 && i < n) {
            if (bias) s += bias[i % nbias];
            out[i] = s;
        }
        if (ZL > 1) __syncthreads();
    }
}

template <int BM, int BN, int WM>
int launch_layout(const GemmArgs& ga, hipStream_t st) {
    const dpp_gemm_desc& d = ga.d;
    dim3 grid(dpp_cdiv(d.M, BM), dpp_cdiv(d.N, BN), d.splitk);
    const bool k32 = ga.bk == 32;
    if (d.a_kc && d.b_kc) {
        if (k32) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, 32, true, true>), grid, dim3(DPP_THREADS), 0, st, ga);
        else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, 16, true, true>), grid, dim3(DPP_THREADS), 0, st, ga);
    } else if (d.a_kc && !d.b_kc) {
        if (k32) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, 32, true, false>), grid, dim3(DPP_THREADS), 0, st, ga);
        else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, 16, true, false>), grid, dim3(DPP_THREADS), 0, st, ga);
    } else if (!d.a_kc && !d.b_kc) {
        // reduction over pixels / samples: long K, both operands [k][mn] -> 64-deep chunks keep 20+ KB per workgroup in flight
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, 64, false, false>), grid, dim3(DPP_THREADS), 0, st, ga);
    } else
        return DPP_E_UNSUPPORTED;
    return dpp_launch_status();
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dpp_abi_version(void) { return DPP_ABI_VERSION; }

extern "C" int dpp_gemm(const dpp_gemm_desc* dp, dpp_stream_t stream) {
    if (!dp || !dp->A || !dp->B || dp->M <= 0 || dp->N <= 0 || dp->K <= 0) return DPP_E_BADARG;
    GemmArgs ga;
    ga.d = *dp;
    dpp_gemm_desc& d = ga.d;
    if (d.splitk < 1) d.splitk = 1;
    if (d.splitk > 1 && !d.partial) return DPP_E_BADARG;
    if (d.splitk == 1 && !d.C) return DPP_E_BADARG;
    if (d.actA.mode && d.actA.cmod <= 0) return DPP_E_BADARG;
    if (d.actB.mode && d.actB.cmod <= 0) return DPP_E_BADARG;
    ga.vecA = aligned16(d.A) && (d.lda % 4 == 0);
    ga.vecB = aligned16(d.B) && (d.ldb % 4 == 0);
    if ((d.actA.mode & 2) && !(aligned16(d.actA.mean) && aligned16(d.actA.scale) && aligned16(d.actA.beta))) return DPP_E_BADARG;
    if ((d.actB.mode & 2) && !(aligned16(d.actB.mean) && aligned16(d.actB.scale) && aligned16(d.actB.beta))) return DPP_E_BADARG;
    const bool red_layout = !d.a_kc && !d.b_kc;
    ga.bk = (d.K > 16) ? 32 : 16;
    const int chunk = red_layout ? 64 : ga.bk;
    int kper = dpp_cdiv(d.K, d.splitk);
    ga.Kper = dpp_cdiv(kper, chunk) * chunk;
    // every requested slice is written (slices beyond K hold zeros), so the caller's reduce over `splitk` slices is exact
    int bm = d.bm, bn = d.bn, wm = d.wm;
    if (bm == 0) {
        if (d.M <= 16) { bm = 16; bn = 64; wm = 1; }
        else if (d.M <= 32) { bm = 32; bn = 64; wm = 1; }
        else {
            wm = 4;
            bn = d.N > 32 ? 64 : (d.N > 16 ? 32 : 16);
            long blocks128 = (long)dpp_cdiv(d.M, 128) * dpp_cdiv(d.N, bn) * d.splitk;
            bm = blocks128 >= 512 ? 128 : 64;
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_TILE(BM_, BN_, WM_) if (bm == BM_ && bn == BN_ && wm == WM_) return launch_layout<BM_, BN_, WM_>(ga, st);
    DPP_TILE(128, 64, 4)
    DPP_TILE(128, 32, 4)
    DPP_TILE(128, 16, 4)
    DPP_TILE(64, 64, 4)
    DPP_TILE(64, 32, 4)
    DPP_TILE(64, 16, 4)
    DPP_TILE(16, 64, 1)
    DPP_TILE(32, 64, 1)
    DPP_TILE(16, 128, 1)
#undef DPP_TILE
    return DPP_E_UNSUPPORTED;
}

extern "C" int dpp_reduce_partials(const float* partial, int nz, int n, const float* bias, int nbias, float* out,
                                   dpp_stream_t stream) {
    if (!partial || !out || nz < 1 || n < 1) return DPP_E_BADARG;
    int ZL = 1;                                   // z-lanes: trade column parallelism for slice parallelism on small outputs
    while (ZL < 16 && ZL * 2 <= nz && dpp_cdiv(n, DPP_THREADS / ZL) < 512) ZL *= 2;
    int blocks = dpp_cdiv(n, DPP_THREADS / ZL);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
                       partial, nz, n, bias, nbias > 0 ? nbias : 1, out, ZL);
    return dpp_launch_status();
}
