// gemm_kernels.h -- the LDS-tiled, K-split, row-stream and 16-column-stream kernels of dpp_gemm with their launchers (see gemm.hip for the
// overview).  Included by gemm.hip (float32 instantiations, entry points) and gemm_st.hip (instantiations for bf16-stored tensors).
#pragma once
#include <stdlib.h>
#include "gemm_args.h"

namespace {




// Load 4 consecutive floats p[0..3] where element e is valid iff (idx0 + e) < limit.
// (b16: `p` addresses bf16 elements -- 8-byte vector load or 2-byte element loads, widened exactly)
__device__ __forceinline__ float4 load4(const float* p, int idx0, int limit, bool vec, bool b16 = false) {
    if (b16) {
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

// Operand A in mode 4: the gradient through a BatchNorm's batch statistics, v = scale*g - aux*(x - mean) - beta, from the masked
// gradient g and the BatchNorm input x (4 consecutive elements each); elements at or beyond `limit` are zero.
__device__ __forceinline__ float4 bnbwd4_masked(float4 g, float4 x, const dpp_act& a, int c0, int limit) {
    if ((a.cmod & 3) == 0 && c0 + 3 < limit) {
        const int c = c0 % a.cmod;
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + c), ax = *reinterpret_cast<const float4*>(a.aux + c);
        const float4 mu = *reinterpret_cast<const float4*>(a.mean + c), be = *reinterpret_cast<const float4*>(a.beta + c);
        return make_float4(sc.x * g.x - ax.x * (x.x - mu.x) - be.x, sc.y * g.y - ax.y * (x.y - mu.y) - be.y,
                           sc.z * g.z - ax.z * (x.z - mu.z) - be.z, sc.w * g.w - ax.w * (x.w - mu.w) - be.w);
    }
    const float gv[4] = {g.x, g.y, g.z, g.w}, xv[4] = {x.x, x.y, x.z, x.w};
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (c0 + j < limit) {
            const int c = (c0 + j) % a.cmod;
            o[j] = a.scale[c] * gv[j] - a.aux[c] * (xv[j] - a.mean[c]) - a.beta[c];
        }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

// Shared epilogue of the GEMM kernels: bias / residual / fused BatchNorm-backward mask, stores (or split-K partial stores),
// and the fused column statistics.  acc is in the MFMA D layout; `red` is LDS scratch of >= WM*BN floats.
template <int RM, int CN, int WM, int WN, int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[RM][CN], const dpp_gemm_desc& d, int row0, int col0, int wm, int wn,
                                              int l15, int kq, float* As) {
    const int M = d.M, N = d.N;
    // ---- epilogue: D layout col = lane&15, row = (lane>>4)*4 + r --------------------------------
    const dpp_epilogue& ep = d.epi;
    const bool fused = d.splitk == 1 && (ep.stats != nullptr || ep.bn_x != nullptr);
    float sx[CN], sy[CN];            // fused BatchNorm-backward sums (sum G, sum G*xhat)
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) { sx[ct] = 0.0f; sy[ct] = 0.0f; }
    // per-column vectors are read once (the stores to C below could alias them as far as the compiler knows)
    float cbias[CN], cmean[CN], cscale[CN], cbeta[CN], cistd[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) {
        const int col = col0 + wn * (BN / WN) + ct * 16 + l15;
        const bool in = col < N && d.splitk == 1;
        cbias[ct] = (in && d.bias) ? d.bias[col] : 0.0f;
        const bool bn = in && ep.bn_x != nullptr;
        cmean[ct] = bn ? ep.bn_mean[col] : 0.0f;
        cscale[ct] = bn ? ep.bn_scale[col] : 0.0f;
        cbeta[ct] = bn ? ep.bn_beta[col] : 0.0f;
        cistd[ct] = bn ? ep.bn_inv_std[col] : 0.0f;
    }
#pragma unroll
    for (int rt = 0; rt < RM; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = row0 + wm * (BM / WM) + rt * 16 + kq * 4 + r;
            if (row >= M) {
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) acc[rt][ct][r] = 0.0f;
                continue;
            }
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
                    float v = 0.0f;
                    if (col < N) {
                        v = acc[rt][ct][r] + cbias[ct];
                        if (d.residual) v += d.residual[o + col];
                        if (ep.bn_x != nullptr) {
                            float dx = ep.bn_x[o + col] - cmean[ct];
                            if (ep.bn_relu && dx * cscale[ct] + cbeta[ct] < 0.0f) v = 0.0f;
                            sx[ct] += v;
                            sy[ct] += v * (dx * cistd[ct]);
                        }
                        d.C[o + col] = v;
                    }
                    acc[rt][ct][r] = v;
                }
            }
        }
    }
    if (fused) {
        float* red = As;                                   // the operand tiles are dead after the last barrier of the K loop
        const int cbase = col0 + wn * (BN / WN) + l15;
        if (ep.bn_x != nullptr && ep.bn_partial != nullptr) {
            dpp_tile_colsum<CN, WM, WN, BN>(sx, red, wm, wn, l15, kq);
            dpp_tile_colsum<CN, WM, WN, BN>(sy, red, wm, wn, l15, kq);
            if (kq == 0 && wm == 0) {
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    int col = cbase + ct * 16;
                    if (col < N) {
                        ep.bn_partial[dpp_partial_index(0, col, blockIdx.x, N, gridDim.x)] = sx[ct];
                        ep.bn_partial[dpp_partial_index(1, col, blockIdx.x, N, gridDim.x)] = sy[ct];
                    }
                }
            }
        }
        if (ep.stats != nullptr) {
            // two passes over the registers: block mean first, then M2 about it (no cancellation when |mean| >> std)
            const int nvalid = (M - row0 < BM) ? (M - row0) : BM;
            float sm[CN];
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                float t = 0.0f;
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t += acc[rt][ct][r];       // invalid rows were zeroed above
                sm[ct] = t;
            }
            dpp_tile_colsum<CN, WM, WN, BN>(sm, red, wm, wn, l15, kq);
            float m2[CN];
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                sm[ct] = sm[ct] / (float)nvalid;
                float t = 0.0f;
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row = row0 + wm * (BM / WM) + rt * 16 + kq * 4 + r;
                        float dv = acc[rt][ct][r] - sm[ct];
                        if (row < M) t += dv * dv;
                    }
                m2[ct] = t;
            }
            dpp_tile_colsum<CN, WM, WN, BN>(m2, red, wm, wn, l15, kq);
            if (kq == 0 && wm == 0) {
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    int col = cbase + ct * 16;
                    if (col < N) {
                        ep.stats[dpp_partial_index(0, col, blockIdx.x, N, gridDim.x)] = sm[ct];
                        ep.stats[dpp_partial_index(1, col, blockIdx.x, N, gridDim.x)] = m2[ct];
                    }
                }
            }
        }
    }
}

// Global -> register fetch of one float4 staging slot (with the operand prologue applied), and its LDS address.
// K-contiguous operand: slot = (row r, k-quad c4); MN-contiguous operand: slot = (k row rk, mn-quad c4).
template <int ROWS, int BKT, bool KC>
struct Stager {
    static constexpr int QK = BKT / 4;                       // float4 per row (KC)
    static constexpr int SLOTS = (ROWS * QK + DPP_THREADS - 1) / DPP_THREADS;
    static constexpr int LD = KC ? (BKT + 4) : (ROWS + 4);
};

// ST: the instantiation may meet bf16-stored operands / outputs (DPP_ST_*); the float32 instantiations carry none of that code
// PB (dpp_gemm_desc.precision = 1, BASELINE config 5, round 6): both operands are rounded to bfloat16 (RNE; the activation after its
// prologue) when the fragments are read from the float32 LDS images, eight k-values of a lane become ONE v_mfma_f32_16x16x32_bf16
// operand (lane (i, kq) owns k = kq * KL + e .. + 7 on both sides), accumulation stays float32.  Chunks of 32 / 64 only (KL >= 8).
template <int BM, int BN, int WM, int BKT, bool AKC, bool BKC, int DEPTH, bool LAZY = false, bool WIDE = true, bool ST = false, bool PB = false>
__global__ __launch_bounds__(DPP_THREADS) void gemm_kernel(GemmArgs ga) {
    static_assert(!PB || (BKT % 32 == 0 && !LAZY), "bf16 MFMA operands: 8 k-values per lane and chunk");
    dpp_kernarg_warm<sizeof(GemmArgs)>();
    const dpp_gemm_desc& d = ga.d;
    constexpr int WN = 4 / WM;
    constexpr int RM = BM / (16 * WM);
    constexpr int CN = BN / (16 * WN);
    using SA = Stager<BM, BKT, AKC>;
    using SB = Stager<BN, BKT, BKC>;
    constexpr int LDA_ = SA::LD, LDB_ = SB::LD;
    constexpr int KL = BKT / 4;                              // k-values owned by one lane per chunk: kq*KL + e
    constexpr int SZA = AKC ? BM * LDA_ : BKT * LDA_;
    constexpr int SZB = BKC ? BN * LDB_ : BKT * LDB_;
    constexpr int SZE = BM * (BN + 4) + 16 * BN;             // tile image + reduction scratch of the wide epilogue
    __shared__ __attribute__((aligned(16))) float smem[(SZA + SZB > SZE) ? (SZA + SZB) : SZE];
    float* const As = smem;
    float* const Bs = smem + SZA;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    const int M = d.M, N = d.N;
    const int k_begin = blockIdx.z * ga.Kper;
    const int k_end = (k_begin + ga.Kper < d.K) ? (k_begin + ga.Kper) : d.K;
    // bf16-stored operands (DPP_ST_A / DPP_ST_B; the host only sets them with 16-byte-quad geometry, so every element offset below is a
    // multiple of 4): the `const float*` cursors advance by half the element offset, the loads fetch 8 instead of 16 bytes
    const int shA = ST ? ga.shA : 0, shB = ST ? ga.shB : 0;
    dpp_stamp(ga.prof, 0);

    // ---- per-thread staging slots: base pointers that do not depend on the chunk -----------------------
    const float* a_base[SA::SLOTS];
    const float* b_base[SB::SLOTS];
#pragma unroll
    for (int s = 0; s < SA::SLOTS; ++s) {
        int slot = tid + s * DPP_THREADS;
        a_base[s] = nullptr;
        if (AKC) {
            int gi = row0 + slot / SA::QK;
            if (slot < BM * SA::QK && gi < M) a_base[s] = d.A + (((size_t)dpp_map_row(d.mapA, gi) * d.lda) >> shA);
        } else {
            int gi = row0 + (slot % (BM / 4)) * 4;
            if (slot < BKT * (BM / 4) && gi < M) a_base[s] = d.A + (gi >> shA);
        }
    }
#pragma unroll
    for (int s = 0; s < SB::SLOTS; ++s) {
        int slot = tid + s * DPP_THREADS;
        b_base[s] = nullptr;
        if (BKC) {
            int gj = col0 + slot / SB::QK;
            if (slot < BN * SB::QK && gj < N) b_base[s] = d.B + (((size_t)gj * d.ldb) >> shB);
        } else {
            int gj = col0 + (slot % (BN / 4)) * 4;
            if (slot < BKT * (BN / 4) && gj < N) b_base[s] = d.B + (gj >> shB);
        }
    }

    // ---- fast path (block-uniform): the whole tile inside the problem, whole K chunks, 16-byte loads, prologue channels in quads.
    // The checked path below spends 3-4 instructions per bounds test and a runtime modulo per prologue quad; one wave of a
    // 64x16 K=256 forward tile issued 2360 instructions for its 64 MFMAs, and two such waves per SIMD at ~4 cycles per
    // instruction are 8 of the kernel's 14 us (tools/inst_summary.py, profiles/r02_instruction_mix.txt).  On the fast path a
    // staging slot is one load and, at commit, 12 VALU + one ds_write; the BatchNorm coefficients of a thread's quad are loaded
    // once per chunk (K-contiguous operand: all slots of a thread share the k-quad) or once per kernel (MN-contiguous operand:
    // the quad is the thread's column) together with the operand loads, not after them.
    constexpr int QA = AKC ? SA::QK : BM / 4, QB = BKC ? SB::QK : BN / 4;      // quads per staged row
    constexpr bool A_EXACT = ((AKC ? BM * SA::QK : BKT * (BM / 4)) % DPP_THREADS) == 0 && DPP_THREADS % QA == 0;
    constexpr bool B_EXACT = ((BKC ? BN * SB::QK : BKT * (BN / 4)) % DPP_THREADS) == 0 && DPP_THREADS % QB == 0;
    const int klen = k_end - k_begin;
    const bool kfull = klen > 0 && (klen % BKT) == 0;
    const int modeA = d.actA.mode, modeB = d.actB.mode;
    const bool fastA = !LAZY && A_EXACT && ga.vecA && kfull && row0 + BM <= M && (modeA == 0 || (d.actA.cmod & 3) == 0);
    const bool fastB = B_EXACT && ga.vecB && kfull && col0 + BN <= N && (modeB == 0 || (d.actB.cmod & 3) == 0);
    const int qa = (tid % QA) * 4, qb = (tid % QB) * 4;                          // this thread's quad along the contiguous dimension
    const int ldsA0 = (tid / QA) * LDA_ + qa, ldsB0 = (tid / QB) * LDB_ + qb;    // its first LDS slot; slot s is s*(THREADS/Q) rows below
    struct Co4 { float4 mu, sc, be; };
    auto load_co = [](const dpp_act& a, int c0) {
        const int c = c0 < a.cmod ? c0 : c0 % a.cmod;
        Co4 o;
        o.mu = *reinterpret_cast<const float4*>(a.mean + c);
        o.sc = *reinterpret_cast<const float4*>(a.scale + c);
        o.be = *reinterpret_cast<const float4*>(a.beta + c);
        return o;
    };
    auto apply_co = [](float4 v, const Co4& co, int mode) {
        if (mode & 2) {
            v.x = dpp_fma(v.x - co.mu.x, co.sc.x, co.be.x);
            v.y = dpp_fma(v.y - co.mu.y, co.sc.y, co.be.y);
            v.z = dpp_fma(v.z - co.mu.z, co.sc.z, co.be.z);
            v.w = dpp_fma(v.w - co.mu.w, co.sc.w, co.be.w);
        }
        if (mode & 1) {
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
        }
        return v;
    };
    Co4 coA[DEPTH], coB[DEPTH];
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) {
        coA[dd].mu = coA[dd].sc = coA[dd].be = make_float4(0.f, 0.f, 0.f, 0.f);
        coB[dd] = coA[dd];
    }
    if (!AKC && fastA && (modeA & 2)) coA[0] = load_co(d.actA, row0 + qa);
    if (!BKC && fastB && (modeB & 2)) coB[0] = load_co(d.actB, col0 + qb);

    // DEPTH chunks are kept in flight per workgroup (a ring of register sets): what bounds these skinny GEMMs is the
    // latency of the dependent load -> barrier -> MFMA chain, not bandwidth or the MFMA rate.
    float4 ra[DEPTH][SA::SLOTS], rb[DEPTH][SB::SLOTS];
    // LAZY: operand A = gradient through a BatchNorm, built from (g, x2) (dpp_act mode 4); its own instantiation, so that the
    // second load and the extra per-channel vectors cost the ordinary GEMMs nothing
    const ptrdiff_t a2off = LAZY ? (d.actA.x2 - d.A) : 0;
    auto fetch = [&](int dd, int kc) {
        float4* const ra_ = ra[dd];
        float4* const rb_ = rb[dd];
        if (fastA) {
            if (AKC) {
                if (modeA & 2) coA[dd] = load_co(d.actA, kc + qa);
                if (shA) {
#pragma unroll
                    for (int s = 0; s < SA::SLOTS; ++s) ra_[s] = dpp_raw8(a_base[s] + ((kc + qa) >> 1));
                } else {
#pragma unroll
                    for (int s = 0; s < SA::SLOTS; ++s) ra_[s] = *reinterpret_cast<const float4*>(a_base[s] + kc + qa);
                }
            } else if (shA) {
#pragma unroll
                for (int s = 0; s < SA::SLOTS; ++s)
                    ra_[s] = dpp_raw8(a_base[s] + (((size_t)dpp_map_row(d.mapA, kc + tid / QA + s * (DPP_THREADS / QA)) * d.lda) >> 1));
            } else {
#pragma unroll
                for (int s = 0; s < SA::SLOTS; ++s)
                    ra_[s] = *reinterpret_cast<const float4*>(a_base[s] + (size_t)dpp_map_row(d.mapA, kc + tid / QA + s * (DPP_THREADS / QA)) * d.lda);
            }
        } else
#pragma unroll
        for (int s = 0; s < SA::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_base[s] != nullptr) {
                if (AKC) {
                    int k = kc + (slot % SA::QK) * 4;
                    if (k < k_end) {
                        const float* pa = a_base[s] + (k >> shA);
                        const float4 g = load4(pa, k, k_end, ga.vecA, shA != 0);
                        if constexpr (LAZY) {
                            v = bnbwd4_masked(g, load4(pa + a2off, k, k_end, ga.vecA), d.actA, k, k_end);
                            if (d.actA.out != nullptr && blockIdx.y == 0) {
                                // every operand element is staged exactly once by the first column block: it leaves a copy
                                float* po = d.actA.out + (pa - d.A);
                                if (ga.vecA && k + 3 < k_end) *reinterpret_cast<float4*>(po) = v;
                                else {
                                    const float vv[4] = {v.x, v.y, v.z, v.w};
                                    for (int j = 0; j < 4; ++j)
                                        if (k + j < k_end) po[j] = vv[j];
                                }
                            }
                        } else v = g;                  // the BN + ReLU prologue is applied when the chunk is written to LDS (commit)
                    }
                } else {
                    int k = kc + slot / (BM / 4), gi = row0 + (slot % (BM / 4)) * 4;
                    if (k < k_end) {
                        const float* pa = a_base[s] + (((size_t)dpp_map_row(d.mapA, k) * d.lda) >> shA);
                        const float4 g = load4(pa, gi, M, ga.vecA, shA != 0);
                        if constexpr (LAZY) v = bnbwd4_masked(g, load4(pa + a2off, gi, M, ga.vecA), d.actA, gi, M);
                        else v = g;
                    }
                }
            }
            ra_[s] = v;
        }
        if (fastB) {
            if (BKC) {
                if (modeB & 2) coB[dd] = load_co(d.actB, kc + qb);
                if (shB) {
#pragma unroll
                    for (int s = 0; s < SB::SLOTS; ++s) rb_[s] = dpp_raw8(b_base[s] + ((kc + qb) >> 1));
                } else {
#pragma unroll
                    for (int s = 0; s < SB::SLOTS; ++s) rb_[s] = *reinterpret_cast<const float4*>(b_base[s] + kc + qb);
                }
            } else if (shB) {
#pragma unroll
                for (int s = 0; s < SB::SLOTS; ++s)
                    rb_[s] = dpp_raw8(b_base[s] + (((size_t)dpp_map_row(d.mapB, kc + tid / QB + s * (DPP_THREADS / QB)) * d.ldb) >> 1));
            } else {
#pragma unroll
                for (int s = 0; s < SB::SLOTS; ++s)
                    rb_[s] = *reinterpret_cast<const float4*>(b_base[s] + (size_t)dpp_map_row(d.mapB, kc + tid / QB + s * (DPP_THREADS / QB)) * d.ldb);
            }
        } else
#pragma unroll
        for (int s = 0; s < SB::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_base[s] != nullptr) {
                if (BKC) {
                    int k = kc + (slot % SB::QK) * 4;
                    if (k < k_end) v = load4(b_base[s] + (k >> shB), k, k_end, ga.vecB, shB != 0);
                } else {
                    int k = kc + slot / (BN / 4), gj = col0 + (slot % (BN / 4)) * 4;
                    if (k < k_end) v = load4(b_base[s] + (((size_t)dpp_map_row(d.mapB, k) * d.ldb) >> shB), gj, N, ga.vecB, shB != 0);
                }
            }
            rb_[s] = v;
        }
    };
    // commit: apply the BN + ReLU prologue to the raw registers of chunk `kc` and write them to LDS.  Doing it HERE, not in
    // fetch, is what lets the loads of the next chunk stay in flight under this chunk's MFMAs: a prologue in fetch makes the
    // wave wait for its loads right where they are issued (measured with tools/phase_profile.py: 2-3 us per 64-deep chunk of the
    // K = 256 layers, all of it exposed latency).
    auto commit = [&](int dd, int kc) {
        const float4* const ra_ = ra[dd];
        const float4* const rb_ = rb[dd];
        if (fastA) {
#pragma unroll
            for (int s = 0; s < SA::SLOTS; ++s) {
                float4 v = shA ? dpp_widen4(ra_[s]) : ra_[s];
                if (modeA != 0) v = apply_co(v, coA[AKC ? dd : 0], modeA);
                *reinterpret_cast<float4*>(&As[ldsA0 + s * (DPP_THREADS / QA) * LDA_]) = v;
            }
        } else
#pragma unroll
        for (int s = 0; s < SA::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            float4 v = ra_[s];
            if (!LAZY && d.actA.mode != 0 && a_base[s] != nullptr) {
                if (AKC) { const int k = kc + (slot % SA::QK) * 4; if (k < k_end) v = act4_masked(v, d.actA, k, k_end); }
                else { const int k = kc + slot / (BM / 4), gi = row0 + (slot % (BM / 4)) * 4; if (k < k_end) v = act4_masked(v, d.actA, gi, M); }
            }
            if (AKC) { if (slot < BM * SA::QK) *reinterpret_cast<float4*>(&As[(slot / SA::QK) * LDA_ + (slot % SA::QK) * 4]) = v; }
            else { if (slot < BKT * (BM / 4)) *reinterpret_cast<float4*>(&As[(slot / (BM / 4)) * LDA_ + (slot % (BM / 4)) * 4]) = v; }
        }
        if (fastB) {
#pragma unroll
            for (int s = 0; s < SB::SLOTS; ++s) {
                float4 v = shB ? dpp_widen4(rb_[s]) : rb_[s];
                if (modeB != 0) v = apply_co(v, coB[BKC ? dd : 0], modeB);
                *reinterpret_cast<float4*>(&Bs[ldsB0 + s * (DPP_THREADS / QB) * LDB_]) = v;
            }
        } else
#pragma unroll
        for (int s = 0; s < SB::SLOTS; ++s) {
            int slot = tid + s * DPP_THREADS;
            float4 v = rb_[s];
            if (d.actB.mode != 0 && b_base[s] != nullptr) {
                if (BKC) { const int k = kc + (slot % SB::QK) * 4; if (k < k_end) v = act4_masked(v, d.actB, k, k_end); }
                else { const int k = kc + slot / (BN / 4), gj = col0 + (slot % (BN / 4)) * 4; if (k < k_end) v = act4_masked(v, d.actB, gj, N); }
            }
            if (BKC) { if (slot < BN * SB::QK) *reinterpret_cast<float4*>(&Bs[(slot / SB::QK) * LDB_ + (slot % SB::QK) * 4]) = v; }
            else { if (slot < BKT * (BN / 4)) *reinterpret_cast<float4*>(&Bs[(slot / (BN / 4)) * LDB_ + (slot % (BN / 4)) * 4]) = v; }
        }
    };

    f32x4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dpp_wide_coef wco;
    if (ga.wide) wco.load<BN>(col0, N, d.bias, d.epi, d.C);

    // Software pipeline: DEPTH chunks are fetched ahead; chunk c is written to LDS from ring slot c % DEPTH, and as soon
    // as the barrier publishes it the slot is refilled with chunk c + DEPTH, whose loads stay in flight under the MFMAs.
    const int nchunks = (k_end > k_begin) ? (k_end - k_begin + BKT - 1) / BKT : 0;
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd)
        if (dd < nchunks) fetch(dd, k_begin + dd * BKT);
    dpp_stamp(ga.prof, 1);
    for (int c0 = 0; c0 < nchunks; c0 += DEPTH) {
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) {
            const int c = c0 + dd;
            if (c < nchunks) {
                commit(dd, k_begin + c * BKT);
                __syncthreads();
                if (c == 0) dpp_stamp(ga.prof, 2);
                if (c + DEPTH < nchunks) fetch(dd, k_begin + (c + DEPTH) * BKT);
                if constexpr (PB) {
#pragma unroll
                    for (int e8 = 0; e8 < KL; e8 += 8) {
                        dpp_bf16x8 af[RM], bf[CN];
#pragma unroll
                        for (int rt = 0; rt < RM; ++rt) {
                            const int r = wm * (BM / WM) + rt * 16 + l15;
                            float f[8];
                            if (AKC) {
                                const float4 v0 = *reinterpret_cast<const float4*>(&As[r * LDA_ + kq * KL + e8]);
                                const float4 v1 = *reinterpret_cast<const float4*>(&As[r * LDA_ + kq * KL + e8 + 4]);
                                f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
                            } else {
#pragma unroll
                                for (int t = 0; t < 8; ++t) f[t] = As[(kq * KL + e8 + t) * LDA_ + r];
                            }
#pragma unroll
                            for (int t = 0; t < 8; ++t) af[rt][t] = (dpp_bf16)f[t];
                        }
#pragma unroll
                        for (int ct = 0; ct < CN; ++ct) {
                            const int cc = wn * (BN / WN) + ct * 16 + l15;
                            float f[8];
                            if (BKC) {
                                const float4 v0 = *reinterpret_cast<const float4*>(&Bs[cc * LDB_ + kq * KL + e8]);
                                const float4 v1 = *reinterpret_cast<const float4*>(&Bs[cc * LDB_ + kq * KL + e8 + 4]);
                                f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
                            } else {
#pragma unroll
                                for (int t = 0; t < 8; ++t) f[t] = Bs[(kq * KL + e8 + t) * LDB_ + cc];
                            }
#pragma unroll
                            for (int t = 0; t < 8; ++t) bf[ct][t] = (dpp_bf16)f[t];
                        }
#pragma unroll
                        for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                            for (int ct = 0; ct < CN; ++ct)
                                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt], bf[ct], acc[rt][ct], 0, 0, 0);
                    }
                } else
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
                        int cc = wn * (BN / WN) + ct * 16 + l15;
                        if (BKC) {
                            float4 v = *reinterpret_cast<const float4*>(&Bs[cc * LDB_ + kq * KL + e4]);
                            bf[ct][0] = v.x; bf[ct][1] = v.y; bf[ct][2] = v.z; bf[ct][3] = v.w;
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) bf[ct][t] = Bs[(kq * KL + e4 + t) * LDB_ + cc];
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
        }
    }

    dpp_stamp(ga.prof, 3);
    // WIDE = false: instantiations that never take the LDS-image epilogue (split-K filter gradients) do not carry its registers
    if (WIDE && ga.wide) {
        const int nvalid = (M - row0 < BM) ? (M - row0) : BM;
        dpp_epilogue_wide<RM, CN, WM, WN, BM, BN, 1, ST>(acc, smem, col0, N, wco, d.residual, d.C, d.epi, nvalid, wm, wn, l15, kq, [&](int rl) {
            const int row = row0 + rl;
            return row < M ? (long)dpp_map_row(d.mapC, row) * d.ldc : -1L;
        }, 0, d.store);
    }
    else gemm_epilogue<RM, CN, WM, WN, BM, BN>(acc, d, row0, col0, wm, wn, l15, kq, As);
    dpp_stamp(ga.prof, 4);
}

// ---- row-streaming variant for the skinny conv GEMMs (M = pixels >> K, N) -------------------------------------------------
// Each wave owns 16*RM output rows and ALL BN columns of the workgroup: its A fragments are loaded straight from global
// memory into registers (lane (i, kq) reads 16 B of row i; the four kq lanes of a row cover one 64-B segment), the small B
// slice (weights, K x BN) is staged in LDS ONCE, and there is no barrier in the K loop at all -- waves never wait for each
// other, so the CU hides HBM latency purely with its 8-16 resident waves.  Requires a K-contiguous A, splitk == 1.
template <int RM, int CN, bool BKC>
__global__ __launch_bounds__(DPP_THREADS) void gemm_rowstream_kernel(GemmArgs ga) {
    const dpp_gemm_desc& d = ga.d;
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* Bs = reinterpret_cast<float*>(smem4);
    constexpr int BM = 64 * RM, BN = 16 * CN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    const int M = d.M, N = d.N, K = d.K;
    const int K16 = (K + 15) & ~15;
    const int LDB = BKC ? (K16 + 4) : (BN + 4);
    // ---- stage the B slice (zero padded to K16 x BN) ----
    if (BKC) {
        const int q = K16 >> 2;
        for (int s = tid; s < BN * q; s += DPP_THREADS) {
            int j = s / q, c4 = s - j * q, k = c4 * 4, gj = col0 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gj < N && k < K) v = act4_masked(load4(d.B + (size_t)gj * d.ldb + k, k, K, ga.vecB), d.actB, k, K);
            *reinterpret_cast<float4*>(&Bs[j * LDB + k]) = v;
        }
    } else {
        constexpr int q = BN / 4;
        for (int s = tid; s < K16 * q; s += DPP_THREADS) {
            int k = s / q, c4 = s - k * q, gj = col0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K && gj < N) v = act4_masked(load4(d.B + (size_t)dpp_map_row(d.mapB, k) * d.ldb + gj, gj, N, ga.vecB), d.actB, gj, N);
            *reinterpret_cast<float4*>(&Bs[k * LDB + c4 * 4]) = v;
        }
    }
    const float* arow[RM];
#pragma unroll
    for (int rt = 0; rt < RM; ++rt) {
        int gi = row0 + wave * (16 * RM) + rt * 16 + l15;
        arow[rt] = gi < M ? d.A + (size_t)dpp_map_row(d.mapA, gi) * d.lda : nullptr;
    }
    f32x4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int kc = 0; kc < K16; kc += 16) {
        const int k = kc + kq * 4;
        float4 av[RM];
#pragma unroll
        for (int rt = 0; rt < RM; ++rt) {
            av[rt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow[rt] != nullptr && k < K) av[rt] = act4_masked(load4(arow[rt] + k, k, K, ga.vecA), d.actA, k, K);
        }
        float bf[CN][4];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            if (BKC) {
                float4 v = *reinterpret_cast<const float4*>(&Bs[(ct * 16 + l15) * LDB + k]);
                bf[ct][0] = v.x; bf[ct][1] = v.y; bf[ct][2] = v.z; bf[ct][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bf[ct][t] = Bs[(k + t) * LDB + ct * 16 + l15];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                for (int ct = 0; ct < CN; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), bf[ct][t], acc[rt][ct], 0, 0, 0);
    }
    __syncthreads();                        // the B slice is dead: the epilogue reuses LDS for its column reductions
    gemm_epilogue<RM, CN, 4, 1, BM, BN>(acc, d, row0, col0, wave, 0, l15, kq, Bs);
}

// ---- K-split variant for the long-K / narrow-N 1x1 convolutions (variant 2) ------------------------------------------------
// The stage-3/4 bottleneck entries (256 -> 64 channels over 8 192 pixels) and the data gradients of the bottleneck exits (same
// shape) are four dependent 64-deep global -> LDS -> MFMA round trips in gemm_kernel, on 64 x 16 tiles that pull every A row
// through the CUs four times (tools/phase_profile.py: 6.2 of the kernel's 12.8 us are the K loop, 2.4 us its entry code).  Here a
// workgroup owns 32 rows x ALL BN columns and the WHOLE K: every load of the kernel is issued up front (one memory round trip),
// one barrier publishes the operands, then wave w multiplies its quarter of K (KT/4 deep, no barrier, 8 accumulator tiles) and
// the four partial tiles meet in the LDS images of the wide epilogue.  Compile-time K and BN, whole tiles only: the entry code
// is a few dozen instructions.  A is K-contiguous [M][KT] with the BatchNorm + ReLU prologue, B either K-contiguous [BN][KT]
// (forward: the filters) or [KT][BN] (data gradient).
// Measured: alone the kernel is no faster than the 64 x 16 tiles (9.8 vs 9.3 us: a CU takes in ~25 KB/us whatever the access
// pattern, and 96 KB per workgroup is 3.8 us of that), in the step it is (3.96 -> 3.85 ms: 24 instead of 40 MB through the
// L2s per launch).  Running row tile t on the XCD that the 64-row producers put rows 64 (t/2).. on (their L2 keeps what it wrote)
// changed nothing (3.84-3.86 ms either way): not kept.
// (amdgpu_waves_per_eu: with the default occupancy goal the scheduler sinks every load down to its LDS write to save registers --
// "load, wait, write" twenty-four times over; one workgroup per CU is all the 100 KB LDS footprint allows anyway)
template <int KT, int BN, bool BKC, bool ST = false, bool PB = false>
__global__ __launch_bounds__(DPP_THREADS) DPP_WAVES_PER_EU(1, KT >= 256 ? 1 : 4) void gemm_ksplit_kernel(GemmArgs ga) {
    dpp_kernarg_warm<sizeof(GemmArgs)>();
    const dpp_gemm_desc& d = ga.d;
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int BM = 32, RM = 2, CN = BN / 16;
    constexpr int LDA_ = KT + 4, LDB_ = BKC ? KT + 4 : BN + 4;
    constexpr int QK = KT / 4;                               // float4 per K-contiguous row
    constexpr int SA = BM * QK / DPP_THREADS;                // staging slots per thread
    constexpr int SB = BN * QK / DPP_THREADS;
    constexpr int KS = KT / 4;                               // K slice of a wave
    static_assert(DPP_THREADS % QK == 0 && (BM * QK) % DPP_THREADS == 0 && (BN * QK) % DPP_THREADS == 0 && KS % 16 == 0, "tile");
    // LDS: [A tile | the epilogue's four partial-tile images (they take A's place, and more)] [B: resident for the whole walk]
    constexpr int EPI_FLOATS = 4 * BM * (BN + 4) + 16 * BN, A_REGION = (BM * LDA_ > EPI_FLOATS ? BM * LDA_ : EPI_FLOATS);
    float* const As = smem;
    float* const Bs = smem + A_REGION;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int col0 = blockIdx.y * BN, ntiles = d.M / BM;
    dpp_stamp(ga.prof, 0);

    // A workgroup WALKS row tiles blockIdx.x, + gridDim.x, ... (the launch is as many workgroups as are resident at once -- one per CU at
    // 100 KB of LDS): the B slice -- 64 of the 96 KB a 32-row tile takes in -- is staged ONCE, and the next tile's A rows are requested right
    // after the current tile's commit, so they travel under its products and epilogue.  With one tile per workgroup the 1 024-tile launches
    // of the 256 x 256 net were four rounds on a CU, each re-reading B (22 us against 9 us for 256 tiles).
    const int modeA = d.actA.mode;
    const int ka = (tid % QK) * 4;                           // this thread's k-quad (the same for all its A slots)
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), sc = mu, be = mu;
    if (modeA & 2) {
        mu = *reinterpret_cast<const float4*>(d.actA.mean + ka);
        sc = *reinterpret_cast<const float4*>(d.actA.scale + ka);
        be = *reinterpret_cast<const float4*>(d.actA.beta + ka);
    }
    float4 ra[SA], rb[SB];
    constexpr int QN = BN / 4;
    auto fetch_a = [&](int tile) {
        const int row0 = tile * BM;
        if (ST && ga.shA) {                                 // bf16-stored activations: 8-byte loads, widened at the commit
            const float* pa = d.A + (((size_t)(row0 + tid / QK) * d.lda + ka) >> 1);
#pragma unroll
            for (int s = 0; s < SA; ++s) ra[s] = dpp_raw8(pa + (((size_t)s * (DPP_THREADS / QK) * d.lda) >> 1));
        } else {
            const float* pa = d.A + (size_t)(row0 + tid / QK) * d.lda + ka;
#pragma unroll
            for (int s = 0; s < SA; ++s) ra[s] = *reinterpret_cast<const float4*>(pa + (size_t)s * (DPP_THREADS / QK) * d.lda);
        }
    };
    int tile = blockIdx.x;
    {   // B first (its commit is nothing but LDS writes), then the first A tile: loads return in issue order
        if (BKC) {
            const float* pb = d.B + (size_t)(col0 + tid / QK) * d.ldb + ka;
#pragma unroll
            for (int s = 0; s < SB; ++s) rb[s] = *reinterpret_cast<const float4*>(pb + (size_t)s * (DPP_THREADS / QK) * d.ldb);
        } else {
            const float* pb = d.B + (size_t)(tid / QN) * d.ldb + col0 + (tid % QN) * 4;
#pragma unroll
            for (int s = 0; s < SB; ++s) rb[s] = *reinterpret_cast<const float4*>(pb + (size_t)s * (DPP_THREADS / QN) * d.ldb);
        }
        fetch_a(tile);
    }
    dpp_wide_coef wco;
    wco.load<BN>(col0, d.N, d.bias, d.epi, d.C);
    dpp_stamp(ga.prof, 1);
    {
        if (BKC) {
            float* lb = Bs + (tid / QK) * LDB_ + ka;
#pragma unroll
            for (int s = 0; s < SB; ++s) *reinterpret_cast<float4*>(lb + s * (DPP_THREADS / QK) * LDB_) = rb[s];
        } else {
            float* lb = Bs + (tid / QN) * LDB_ + (tid % QN) * 4;
#pragma unroll
            for (int s = 0; s < SB; ++s) *reinterpret_cast<float4*>(lb + s * (DPP_THREADS / QN) * LDB_) = rb[s];
        }
    }
    for (; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * BM;
    if (tile != (int)blockIdx.x) __syncthreads();            // the previous tile's epilogue is done with the images that share A's place
    // ---- commit: this tile's A rows into LDS with the prologue ----
    {
        float* la = As + (tid / QK) * LDA_ + ka;
#pragma unroll
        for (int s = 0; s < SA; ++s) {
            float4 v = (ST && ga.shA) ? dpp_widen4(ra[s]) : ra[s];
            if (modeA & 2) {
                v.x = dpp_fma(v.x - mu.x, sc.x, be.x); v.y = dpp_fma(v.y - mu.y, sc.y, be.y);
                v.z = dpp_fma(v.z - mu.z, sc.z, be.z); v.w = dpp_fma(v.w - mu.w, sc.w, be.w);
            }
            if (modeA & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(la + s * (DPP_THREADS / QK) * LDA_) = v;
        }
    }
    __syncthreads();
    if (tile == (int)blockIdx.x) dpp_stamp(ga.prof, 2);
    if (tile + (int)gridDim.x < ntiles) fetch_a(tile + gridDim.x);

    // ---- wave `wave` multiplies k in [wave*KS, wave*KS + KS): lane (i, kq) owns k = base + 16 g + 4 kq + t ----
    f32x4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow = As + l15 * LDA_ + wave * KS + kq * 4;
    const float* brow = BKC ? Bs + l15 * LDB_ + wave * KS + kq * 4 : Bs + (wave * KS + kq * 4) * LDB_ + l15;
    if constexpr (PB) {
        // bf16 operands: the lane's k-values of two 16-deep groups (k = base + 16 g + 4 kq + t, g = 2 h, 2 h + 1) form one 32-deep step
        static_assert(KS % 32 == 0, "whole 32-deep steps per wave");
#pragma unroll
        for (int h = 0; h < KS / 32; ++h) {
            dpp_bf16x8 af[RM], bf[CN];
#pragma unroll
            for (int rt = 0; rt < RM; ++rt) {
                const float4 v0 = *reinterpret_cast<const float4*>(arow + rt * 16 * LDA_ + h * 32);
                const float4 v1 = *reinterpret_cast<const float4*>(arow + rt * 16 * LDA_ + h * 32 + 16);
                af[rt][0] = (dpp_bf16)v0.x; af[rt][1] = (dpp_bf16)v0.y; af[rt][2] = (dpp_bf16)v0.z; af[rt][3] = (dpp_bf16)v0.w;
                af[rt][4] = (dpp_bf16)v1.x; af[rt][5] = (dpp_bf16)v1.y; af[rt][6] = (dpp_bf16)v1.z; af[rt][7] = (dpp_bf16)v1.w;
            }
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                if (BKC) {
                    const float4 v0 = *reinterpret_cast<const float4*>(brow + ct * 16 * LDB_ + h * 32);
                    const float4 v1 = *reinterpret_cast<const float4*>(brow + ct * 16 * LDB_ + h * 32 + 16);
                    bf[ct][0] = (dpp_bf16)v0.x; bf[ct][1] = (dpp_bf16)v0.y; bf[ct][2] = (dpp_bf16)v0.z; bf[ct][3] = (dpp_bf16)v0.w;
                    bf[ct][4] = (dpp_bf16)v1.x; bf[ct][5] = (dpp_bf16)v1.y; bf[ct][6] = (dpp_bf16)v1.z; bf[ct][7] = (dpp_bf16)v1.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        bf[ct][t] = (dpp_bf16)brow[(h * 32 + t) * LDB_ + ct * 16];
                        bf[ct][4 + t] = (dpp_bf16)brow[(h * 32 + 16 + t) * LDB_ + ct * 16];
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                for (int ct = 0; ct < CN; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt], bf[ct], acc[rt][ct], 0, 0, 0);
        }
    } else
#pragma unroll
    for (int g = 0; g < KS / 16; ++g) {
        float af[RM][4], bf[CN][4];
#pragma unroll
        for (int rt = 0; rt < RM; ++rt) {
            const float4 v = *reinterpret_cast<const float4*>(arow + rt * 16 * LDA_ + g * 16);
            af[rt][0] = v.x; af[rt][1] = v.y; af[rt][2] = v.z; af[rt][3] = v.w;
        }
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            if (BKC) {
                const float4 v = *reinterpret_cast<const float4*>(brow + ct * 16 * LDB_ + g * 16);
                bf[ct][0] = v.x; bf[ct][1] = v.y; bf[ct][2] = v.z; bf[ct][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bf[ct][t] = brow[(g * 16 + t) * LDB_ + ct * 16];
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
    __syncthreads();                         // the A image is dead: the epilogue's tile images take its place
    if (tile == (int)blockIdx.x) dpp_stamp(ga.prof, 3);
    dpp_epilogue_wide<RM, CN, 1, 1, BM, BN, 4, ST>(acc, smem, col0, d.N, wco, d.residual, d.C, d.epi, BM, 0, 0, l15, kq,
                                               [&](int rl) { return (long)(row0 + rl) * d.ldc; }, wave, d.store, tile, ntiles);
    }
    dpp_stamp(ga.prof, 4);
}

// ---- row-streaming variant for the stage-1 shapes with 16 output columns and K = 64 (variant 3) ----------------------------
// 131 072 pixel rows x (64 -> 16 channels): 41 MB of traffic and 0.27 GFLOP, i.e. an elementwise-like stream with a tiny matrix
// product inside.  On 128 x 16 LDS tiles every workgroup runs entry code, one load round trip, four barriers and the LDS-image
// epilogue in lockstep with all the others (tools/phase_profile.py: 2.9 + 5.0 + 2.8 us per workgroup, nothing overlapping).  Here a
// WAVE owns TPW tiles of 16 rows: its A fragments come straight from memory (lane (i, kq) reads 16 bytes of row i, all tiles'
// loads issued before the first use), the 64 x 16 filter lives in registers, and the epilogue works in the MFMA D layout --
// with the 16 rows of a tile fed in 4 x 4-transposed order, so that one store instruction covers four CONSECUTIVE 64-byte rows
// (256 contiguous bytes); no LDS, no barrier until the column reductions at the very end.
// ACT: operand prologue present (forward); EPI: residual / BatchNorm-backward epilogue present -- register diets.  KT = K (16 or 64),
// CN = column tiles of 16 (N = 16 or 64), TPW = 16-row tiles per wave (rows per workgroup = 64 * TPW = one BatchNorm partial block).
template <int KT, int CN, bool BKC, int TPW, bool ACT, bool EPI, bool ST = false, bool PB = false>
__global__ __launch_bounds__(DPP_THREADS) void gemm_stream16_kernel(GemmArgs ga) {
    dpp_kernarg_warm<sizeof(GemmArgs)>();
    const dpp_gemm_desc& d = ga.d;
    constexpr int G = KT / 16, N = 16 * CN;
    __shared__ float red[4 * N];
    constexpr int ROWS = 64 * TPW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wrow0 = blockIdx.x * ROWS + wave * 16 * TPW;
    const int mrow = (l15 & 3) * 4 + (l15 >> 2);             // memory row (within a tile) that MFMA row l15 carries
    const dpp_epilogue& ep = d.epi;
    const bool bn = EPI && ep.bn_x != nullptr;
    const int modeA = ACT ? d.actA.mode : 0;

    // ---- every load up front ----
    float4 bw[G][CN];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            if (BKC) bw[g][ct] = *reinterpret_cast<const float4*>(d.B + (size_t)(ct * 16 + l15) * d.ldb + g * 16 + kq * 4);
            else {
                const float* pb = d.B + (size_t)(g * 16 + kq * 4) * d.ldb + ct * 16 + l15;
                bw[g][ct] = make_float4(pb[0], pb[d.ldb], pb[2 * (size_t)d.ldb], pb[3 * (size_t)d.ldb]);
            }
        }
    float4 ra[TPW][G];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (ST && ga.shA) {                                 // bf16-stored activations: 8-byte loads, widened where they are consumed
            const float* pa = d.A + (((size_t)(wrow0 + t * 16 + mrow) * d.lda + kq * 4) >> 1);
#pragma unroll
            for (int g = 0; g < G; ++g) ra[t][g] = dpp_raw8(pa + g * 8);
        } else {
            const float* pa = d.A + (size_t)(wrow0 + t * 16 + mrow) * d.lda + kq * 4;
#pragma unroll
            for (int g = 0; g < G; ++g) ra[t][g] = *reinterpret_cast<const float4*>(pa + g * 16);
        }
    }
    const bool c16 = ST && (d.store & DPP_ST_C) != 0, x16 = ST && (d.store & DPP_ST_BNX) != 0;
    float4 mu[G], sc[G], be[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        mu[g] = sc[g] = be[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (modeA & 2) {
            mu[g] = *reinterpret_cast<const float4*>(d.actA.mean + g * 16 + kq * 4);
            sc[g] = *reinterpret_cast<const float4*>(d.actA.scale + g * 16 + kq * 4);
            be[g] = *reinterpret_cast<const float4*>(d.actA.beta + g * 16 + kq * 4);
        }
    }
    float cb[CN], cmean[CN], cscale[CN], cbeta[CN], cistd[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) {
        const int col = ct * 16 + l15;
        cb[ct] = d.bias ? d.bias[col] : 0.0f;
        cmean[ct] = bn ? ep.bn_mean[col] : 0.0f; cscale[ct] = bn ? ep.bn_scale[col] : 0.0f;
        cbeta[ct] = bn ? ep.bn_beta[col] : 0.0f; cistd[ct] = bn ? ep.bn_inv_std[col] : 0.0f;
    }
    float xr[TPW][CN][4], rr[TPW][CN][4];                    // bn_x / residual at this lane's output elements
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int ct = 0; ct < CN; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(wrow0 + t * 16 + r * 4 + kq) * d.ldc + ct * 16 + l15;
                xr[t][ct][r] = bn ? dpp_ld1_rt(ep.bn_x, o, x16) : 0.0f;
                rr[t][ct][r] = (EPI && d.residual) ? dpp_ld1_rt(d.residual, o, c16) : 0.0f;
            }

    float vals[TPW][CN][4];
    float sx[CN], sy[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) { sx[ct] = 0.0f; sy[ct] = 0.0f; }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        f32x4 acc[CN];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 va[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4 v = (ST && ga.shA) ? dpp_widen4(ra[t][g]) : ra[t][g];
            if (modeA & 2) {
                v.x = dpp_fma(v.x - mu[g].x, sc[g].x, be[g].x); v.y = dpp_fma(v.y - mu[g].y, sc[g].y, be[g].y);
                v.z = dpp_fma(v.z - mu[g].z, sc[g].z, be[g].z); v.w = dpp_fma(v.w - mu[g].w, sc[g].w, be[g].w);
            }
            if (modeA & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            va[g] = v;
        }
        if constexpr (PB) {
            // bf16 operands: the lane's k-quads of two 16-deep groups make one 32-deep step (K = 16: the upper half is zero)
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < (G + 1) / 2; ++h) {
                const float4 a0 = va[2 * h], a1 = (2 * h + 1 < G) ? va[2 * h + 1] : z4;
                dpp_bf16x8 af;
                af[0] = (dpp_bf16)a0.x; af[1] = (dpp_bf16)a0.y; af[2] = (dpp_bf16)a0.z; af[3] = (dpp_bf16)a0.w;
                af[4] = (dpp_bf16)a1.x; af[5] = (dpp_bf16)a1.y; af[6] = (dpp_bf16)a1.z; af[7] = (dpp_bf16)a1.w;
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    const float4 b0 = bw[2 * h][ct], b1 = (2 * h + 1 < G) ? bw[(2 * h + 1 < G) ? 2 * h + 1 : 0][ct] : z4;
                    dpp_bf16x8 bf;
                    bf[0] = (dpp_bf16)b0.x; bf[1] = (dpp_bf16)b0.y; bf[2] = (dpp_bf16)b0.z; bf[3] = (dpp_bf16)b0.w;
                    bf[4] = (dpp_bf16)b1.x; bf[5] = (dpp_bf16)b1.y; bf[6] = (dpp_bf16)b1.z; bf[7] = (dpp_bf16)b1.w;
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[ct], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 v = va[g];
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, bw[g][ct].x, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, bw[g][ct].y, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, bw[g][ct].z, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, bw[g][ct].w, acc[ct], 0, 0, 0);
            }
        }
        // D layout: this lane holds column ct*16 + l15 of MFMA rows 4 kq + r, i.e. memory rows 4 r + kq of the tile
#pragma unroll
        for (int ct = 0; ct < CN; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(wrow0 + t * 16 + r * 4 + kq) * d.ldc + ct * 16 + l15;
                float v = acc[ct][r] + cb[ct] + rr[t][ct][r];
                if (bn) {
                    const float dx = xr[t][ct][r] - cmean[ct];
                    if (ep.bn_relu && dx * cscale[ct] + cbeta[ct] < 0.0f) v = 0.0f;
                    if (c16) v = dpp_bf16_round(v);          // bf16-stored gradient: sums of the values as stored (dpp_epilogue_wide)
                    sx[ct] += v;
                    sy[ct] += v * (dx * cistd[ct]);
                }
                if (c16) reinterpret_cast<dpp_bf16*>(d.C)[o] = (dpp_bf16)v; else d.C[o] = v;
                vals[t][ct][r] = v;
            }
    }
    // ---- column reductions over the workgroup's rows (the only barriers of the kernel) ----
    if (bn && ep.bn_partial != nullptr) {
        dpp_tile_colsum<CN, 4, 1, N>(sx, red, wave, 0, l15, kq);
        dpp_tile_colsum<CN, 4, 1, N>(sy, red, wave, 0, l15, kq);
        if (wave == 0 && kq == 0) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                ep.bn_partial[dpp_partial_index(0, ct * 16 + l15, blockIdx.x, N, gridDim.x)] = sx[ct];
                ep.bn_partial[dpp_partial_index(1, ct * 16 + l15, blockIdx.x, N, gridDim.x)] = sy[ct];
            }
        }
    }
    if (ep.stats != nullptr) {
        float sm[CN], m2[CN];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            sm[ct] = 0.0f;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sm[ct] += vals[t][ct][r];
        }
        dpp_tile_colsum<CN, 4, 1, N>(sm, red, wave, 0, l15, kq);
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            sm[ct] *= 1.0f / (float)ROWS;
            m2[ct] = 0.0f;
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dv = vals[t][ct][r] - sm[ct]; m2[ct] += dv * dv; }
        }
        dpp_tile_colsum<CN, 4, 1, N>(m2, red, wave, 0, l15, kq);
        if (wave == 0 && kq == 0) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                ep.stats[dpp_partial_index(0, ct * 16 + l15, blockIdx.x, N, gridDim.x)] = sm[ct];
                ep.stats[dpp_partial_index(1, ct * 16 + l15, blockIdx.x, N, gridDim.x)] = m2[ct];
            }
        }
    }
}

// ---- the same stream with the product TRANSPOSED in the accumulators (round 6) ----------------------------------------------------
// gemm_stream16_kernel keeps the MFMA D layout of a [pixels][channels] product: a lane holds ONE channel of four pixel rows, so its
// epilogue touches memory 4 bytes per lane -- 2 bytes on bf16-stored tensors: a store instruction of the wave moves 128 bytes.  At
// 128 x 128 input the tensors sit in the Infinity Cache and that is hidden (5 TB/s); at 256 x 256 (BASELINE config 5) the kernel runs at
// 2.1 TB/s of its algorithmic bytes, bound by the number of memory instructions, not by bytes.
// Here the operands swap places: D^T[channel][pixel] = sum_k W[channel][k] * act(X)[pixel][k] -- the filter fragment is the A operand,
// the pixel fragment (the same 16-byte load of row `l15`) the B operand -- and the D layout gives lane (pixel l15, kq) the FOUR
// CONSECUTIVE channels 4 kq .. 4 kq + 3 of its pixel: bias, residual, BatchNorm input and output are 16-byte (f32) / 8-byte (bf16)
// accesses, 16 pixels x 64 (32) bytes contiguous per instruction, with no LDS image and no 4 x 4 feeding order.  Column sums: over
// the tiles in registers, over the 16 pixel lanes by xor shuffles, over the waves through LDS (fixed order).
template <int KT, int CN, bool BKC, int TPW, bool ACT, bool EPI, bool ST = false, bool PB = false>
__global__ __launch_bounds__(DPP_THREADS) void gemm_stream16t_kernel(GemmArgs ga) {
    dpp_kernarg_warm<sizeof(GemmArgs)>();
    const dpp_gemm_desc& d = ga.d;
    constexpr int G = KT / 16, N = 16 * CN;
    __shared__ __attribute__((aligned(16))) float red[4 * N];
    constexpr int ROWS = 64 * TPW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const dpp_epilogue& ep = d.epi;
    const bool bn = EPI && ep.bn_x != nullptr;
    const bool has_res = EPI && d.residual != nullptr;
    const int modeA = ACT ? d.actA.mode : 0;
    const bool c16 = ST && (d.store & DPP_ST_C) != 0, x16 = ST && (d.store & DPP_ST_BNX) != 0;
    const int nblocks = d.M / ROWS;

    // A workgroup WALKS row blocks blockIdx.x, + gridDim.x, ... (the launch is what is resident at once): the filter fragments and per-channel
    // vectors are loaded once, and the next block's rows (operand, residual, BatchNorm input) are requested as soon as the current block's
    // products are issued, so they travel under its epilogue and column sums.  One block per workgroup, the 4 096-block launches of the
    // 256 x 256 net were four rounds of entry code + one exposed round trip each.
    // ---- loaded once: filter fragments (lane (i, kq): filter row = output channel ct * 16 + i, k = 16 g + 4 kq ..), per-channel vectors ----
    float4 bw[G][CN];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            if (BKC) bw[g][ct] = *reinterpret_cast<const float4*>(d.B + (size_t)(ct * 16 + l15) * d.ldb + g * 16 + kq * 4);
            else {
                const float* pb = d.B + (size_t)(g * 16 + kq * 4) * d.ldb + ct * 16 + l15;
                bw[g][ct] = make_float4(pb[0], pb[d.ldb], pb[2 * (size_t)d.ldb], pb[3 * (size_t)d.ldb]);
            }
        }
    // RAW loads (16 bytes, or 8 of a bf16-stored tensor), widened where they are used: `bn ? dpp_ld4_rt(..) : zero` converts inside the branch, and the
    // branch then ends on s_waitcnt vmcnt(0) -- the data-gradient instance on bf16-stored tensors waited out four round trips one after the other here
    float4 ra[TPW][G];
    float4 xr32[TPW][CN], rr32[TPW][CN];
    uint2 xr16[TPW][CN], rr16[TPW][CN];
    auto fetch = [&](int block) {
        const int wrow0 = block * ROWS + wave * 16 * TPW;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (ST && ga.shA) {
                const float* pa = d.A + (((size_t)(wrow0 + t * 16 + l15) * d.lda + kq * 4) >> 1);
#pragma unroll
                for (int g = 0; g < G; ++g) ra[t][g] = dpp_raw8(pa + g * 8);
            } else {
                const float* pa = d.A + (size_t)(wrow0 + t * 16 + l15) * d.lda + kq * 4;
#pragma unroll
                for (int g = 0; g < G; ++g) ra[t][g] = *reinterpret_cast<const float4*>(pa + g * 16);
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                const size_t o = (size_t)(wrow0 + t * 16 + l15) * d.ldc + ct * 16 + kq * 4;
                if (bn) {
                    if (x16) xr16[t][ct] = *reinterpret_cast<const uint2*>(reinterpret_cast<const dpp_bf16*>(ep.bn_x) + o);
                    else xr32[t][ct] = *reinterpret_cast<const float4*>(ep.bn_x + o);
                }
                if (has_res) {
                    if (c16) rr16[t][ct] = *reinterpret_cast<const uint2*>(reinterpret_cast<const dpp_bf16*>(d.residual) + o);
                    else rr32[t][ct] = *reinterpret_cast<const float4*>(d.residual + o);
                }
            }
    };
    int block = blockIdx.x;
    fetch(block);
    float4 mu[G], sc[G], be[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        mu[g] = sc[g] = be[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (modeA & 2) {
            mu[g] = *reinterpret_cast<const float4*>(d.actA.mean + g * 16 + kq * 4);
            sc[g] = *reinterpret_cast<const float4*>(d.actA.scale + g * 16 + kq * 4);
            be[g] = *reinterpret_cast<const float4*>(d.actA.beta + g * 16 + kq * 4);
        }
    }
    // per-channel vectors of this lane's quad (channels ct * 16 + 4 kq ..)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 cb[CN], cmean[CN], cscale[CN], cbeta[CN], cistd[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) {
        const int c0 = ct * 16 + kq * 4;
        cb[ct] = d.bias ? *reinterpret_cast<const float4*>(d.bias + c0) : z4;
        cmean[ct] = bn ? *reinterpret_cast<const float4*>(ep.bn_mean + c0) : z4;
        cscale[ct] = bn ? *reinterpret_cast<const float4*>(ep.bn_scale + c0) : z4;
        cbeta[ct] = bn ? *reinterpret_cast<const float4*>(ep.bn_beta + c0) : z4;
        cistd[ct] = bn ? *reinterpret_cast<const float4*>(ep.bn_inv_std + c0) : z4;
    }
    auto widen8 = [](const uint2& r) {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    };
    // ---- column sums: the 16 pixel lanes of a quad by xor shuffles, the four waves through LDS in wave order ----
    auto colsum = [&](float4 (&s)[CN]) {
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s[ct].x += __shfl_xor(s[ct].x, o); s[ct].y += __shfl_xor(s[ct].y, o);
                s[ct].z += __shfl_xor(s[ct].z, o); s[ct].w += __shfl_xor(s[ct].w, o);
            }
        }
        __syncthreads();
        if (l15 == 0) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) *reinterpret_cast<float4*>(&red[wave * N + ct * 16 + kq * 4]) = s[ct];
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            float4 t = *reinterpret_cast<const float4*>(&red[ct * 16 + kq * 4]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 u = *reinterpret_cast<const float4*>(&red[w * N + ct * 16 + kq * 4]);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            s[ct] = t;
        }
    };

    for (; block < nblocks; block += gridDim.x) {
        const int wrow0 = block * ROWS + wave * 16 * TPW;
        // ---- this block's epilogue operands out of the fetch registers (the next fetch reuses them) ----
        float4 xr[TPW][CN], rr[TPW][CN];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                xr[t][ct] = bn ? (x16 ? widen8(xr16[t][ct]) : xr32[t][ct]) : z4;
                rr[t][ct] = has_res ? (c16 ? widen8(rr16[t][ct]) : rr32[t][ct]) : z4;
            }
        // ---- products of all the wave's tiles ----
        f32x4 acc[TPW][CN];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) acc[t][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float4 va[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float4 v = (ST && ga.shA) ? dpp_widen4(ra[t][g]) : ra[t][g];
                if (modeA & 2) {
                    v.x = dpp_fma(v.x - mu[g].x, sc[g].x, be[g].x); v.y = dpp_fma(v.y - mu[g].y, sc[g].y, be[g].y);
                    v.z = dpp_fma(v.z - mu[g].z, sc[g].z, be[g].z); v.w = dpp_fma(v.w - mu[g].w, sc[g].w, be[g].w);
                }
                if (modeA & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                va[g] = v;
            }
            if constexpr (PB) {
#pragma unroll
                for (int h = 0; h < (G + 1) / 2; ++h) {
                    const float4 a0 = va[2 * h], a1 = (2 * h + 1 < G) ? va[(2 * h + 1 < G) ? 2 * h + 1 : 0] : z4;
                    dpp_bf16x8 af;
                    af[0] = (dpp_bf16)a0.x; af[1] = (dpp_bf16)a0.y; af[2] = (dpp_bf16)a0.z; af[3] = (dpp_bf16)a0.w;
                    af[4] = (dpp_bf16)a1.x; af[5] = (dpp_bf16)a1.y; af[6] = (dpp_bf16)a1.z; af[7] = (dpp_bf16)a1.w;
#pragma unroll
                    for (int ct = 0; ct < CN; ++ct) {
                        const float4 b0 = bw[2 * h][ct], b1 = (2 * h + 1 < G) ? bw[(2 * h + 1 < G) ? 2 * h + 1 : 0][ct] : z4;
                        dpp_bf16x8 bf;
                        bf[0] = (dpp_bf16)b0.x; bf[1] = (dpp_bf16)b0.y; bf[2] = (dpp_bf16)b0.z; bf[3] = (dpp_bf16)b0.w;
                        bf[4] = (dpp_bf16)b1.x; bf[5] = (dpp_bf16)b1.y; bf[6] = (dpp_bf16)b1.z; bf[7] = (dpp_bf16)b1.w;
                        acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af, acc[t][ct], 0, 0, 0);      // filter rows x pixel columns
                    }
                }
            } else
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float4 v = va[g];
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][ct].x, v.x, acc[t][ct], 0, 0, 0);
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][ct].y, v.y, acc[t][ct], 0, 0, 0);
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][ct].z, v.z, acc[t][ct], 0, 0, 0);
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g][ct].w, v.w, acc[t][ct], 0, 0, 0);
                }
            }
        }
        // ---- the next block's rows: in flight under this block's epilogue and column sums ----
        if (block + (int)gridDim.x < nblocks) fetch(block + gridDim.x);

        float4 vals[TPW][CN];
        float4 sx[CN], sy[CN];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) { sx[ct] = z4; sy[ct] = z4; }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            // D^T layout: this lane holds channels ct * 16 + 4 kq + r (r = 0 .. 3) of pixel row wrow0 + 16 t + l15
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                const size_t o = (size_t)(wrow0 + t * 16 + l15) * d.ldc + ct * 16 + kq * 4;
                float v[4] = {acc[t][ct][0] + cb[ct].x + rr[t][ct].x, acc[t][ct][1] + cb[ct].y + rr[t][ct].y, acc[t][ct][2] + cb[ct].z + rr[t][ct].z,
                              acc[t][ct][3] + cb[ct].w + rr[t][ct].w};
                if (bn) {
                    const float dx[4] = {xr[t][ct].x - cmean[ct].x, xr[t][ct].y - cmean[ct].y, xr[t][ct].z - cmean[ct].z, xr[t][ct].w - cmean[ct].w};
                    const float cs[4] = {cscale[ct].x, cscale[ct].y, cscale[ct].z, cscale[ct].w}, cbt[4] = {cbeta[ct].x, cbeta[ct].y, cbeta[ct].z, cbeta[ct].w};
                    const float ci[4] = {cistd[ct].x, cistd[ct].y, cistd[ct].z, cistd[ct].w};
                    float sxa[4] = {sx[ct].x, sx[ct].y, sx[ct].z, sx[ct].w}, sya[4] = {sy[ct].x, sy[ct].y, sy[ct].z, sy[ct].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ep.bn_relu && dx[r] * cs[r] + cbt[r] < 0.0f) v[r] = 0.0f;
                        if (c16) v[r] = dpp_bf16_round(v[r]);            // bf16-stored gradient: sums of the values as stored
                        sxa[r] += v[r];
                        sya[r] += v[r] * (dx[r] * ci[r]);
                    }
                    sx[ct] = make_float4(sxa[0], sxa[1], sxa[2], sxa[3]);
                    sy[ct] = make_float4(sya[0], sya[1], sya[2], sya[3]);
                }
                const float4 v4 = make_float4(v[0], v[1], v[2], v[3]);
                if (c16) dpp_st4(reinterpret_cast<dpp_bf16*>(d.C) + o, v4); else *reinterpret_cast<float4*>(d.C + o) = v4;
                vals[t][ct] = v4;
            }
        }
        auto put = [&](float* dst, int which, const float4 (&s)[CN]) {
            if (wave == 0 && l15 == 0) {
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    const int c0 = ct * 16 + kq * 4;
                    dst[dpp_partial_index(which, c0 + 0, block, N, nblocks)] = s[ct].x;
                    dst[dpp_partial_index(which, c0 + 1, block, N, nblocks)] = s[ct].y;
                    dst[dpp_partial_index(which, c0 + 2, block, N, nblocks)] = s[ct].z;
                    dst[dpp_partial_index(which, c0 + 3, block, N, nblocks)] = s[ct].w;
                }
            }
        };
        if (bn && ep.bn_partial != nullptr) {
            colsum(sx);
            colsum(sy);
            put(ep.bn_partial, 0, sx);
            put(ep.bn_partial, 1, sy);
        }
        if (ep.stats != nullptr) {
            float4 sm[CN], m2[CN];
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                sm[ct] = z4;
#pragma unroll
                for (int t = 0; t < TPW; ++t) { sm[ct].x += vals[t][ct].x; sm[ct].y += vals[t][ct].y; sm[ct].z += vals[t][ct].z; sm[ct].w += vals[t][ct].w; }
            }
            colsum(sm);
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                const float inv = 1.0f / (float)ROWS;
                sm[ct].x *= inv; sm[ct].y *= inv; sm[ct].z *= inv; sm[ct].w *= inv;
                m2[ct] = z4;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const float a0 = vals[t][ct].x - sm[ct].x, a1 = vals[t][ct].y - sm[ct].y, a2 = vals[t][ct].z - sm[ct].z, a3 = vals[t][ct].w - sm[ct].w;
                    m2[ct].x += a0 * a0; m2[ct].y += a1 * a1; m2[ct].z += a2 * a2; m2[ct].w += a3 * a3;
                }
            }
            colsum(m2);
            put(ep.stats, 0, sm);
            put(ep.stats, 1, m2);
        }
    }
}

// dpp_gemm variant 3: rows per workgroup of gemm_stream16_kernel for this problem (K = 64 -> 16 columns: 128; K = 16 -> 64 columns:
// 64), or 0 when the kernel does not take it
static int stream16_rows(const dpp_gemm_desc& d, const GemmArgs& ga) {
    const bool narrow = d.N == 16 && d.K == 64, wide = d.N == 64 && d.K == 16;
    if (d.store & DPP_ST_B) return 0;
    if (!d.a_kc || d.splitk != 1 || !(narrow || wide) || !ga.vecA || !d.C || d.M % (narrow ? 128 : 64)) return 0;
    if (d.mapA.s != 1 || d.mapB.s != 1 || d.mapC.s != 1 || d.actB.mode != 0 || (d.actA.mode & ~3)) return 0;
    if ((d.actA.mode & 2) && d.actA.cmod < d.K) return 0;        // (dpp_gemm has checked that the prologue vectors are 16-byte aligned)
    if (d.b_kc && !ga.vecB) return 0;
    if (d.epi.stats && d.epi.bn_x) return 0;
    return narrow ? 128 : 64;
}

// dpp_gemm variant 2: can this problem run on gemm_ksplit_kernel, and with which tile?
static int ksplit_bn(const dpp_gemm_desc& d, const GemmArgs& ga) {
    if (d.store & DPP_ST_B) return 0;
    if (!d.a_kc || d.splitk != 1 || !ga.wide || !ga.vecA || !ga.vecB || d.M % 32 || (d.K != 256 && d.K != 128)) return 0;
    if (d.mapA.s != 1 || d.mapB.s != 1 || d.mapC.s != 1 || d.actB.mode != 0 || (d.actA.mode & ~3)) return 0;
    if ((d.actA.mode & 2) && d.actA.cmod < d.K) return 0;
    const int bn = d.K == 256 ? 64 : 32;
    return d.N % bn == 0 ? bn : 0;
}

template <int KT, int BN, bool BKC, bool ST, bool PB = false>
static int launch_ksplit(const GemmArgs& ga, hipStream_t st) {
    const dpp_gemm_desc& d = ga.d;
    constexpr int epil = 4 * 32 * (BN + 4) + 16 * BN, aimg = 32 * (KT + 4);
    constexpr size_t lds = sizeof(float) * ((aimg > epil ? aimg : epil) + (BKC ? BN * (KT + 4) : KT * (BN + 4)));
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ksplit_kernel<KT, BN, BKC, ST, PB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // as many workgroups as are resident at once (LDS: one per CU for K = 256, four for K = 128), walking the row tiles; DPP_KSPLIT_WALK=0: one tile each
    static const bool walk = []() { const char* e = getenv("DPP_KSPLIT_WALK"); return !(e && e[0] == '0'); }();
    const int ntiles = d.M / 32, ncol = d.N / BN, cap = (int)((160 * 1024 / lds) * 256 / ncol);
    DPP_LAUNCH((gemm_ksplit_kernel<KT, BN, BKC, ST, PB>), dim3((walk && ntiles > cap && cap > 0) ? cap : ntiles, ncol), dim3(DPP_THREADS), lds, st, ga);
    return dpp_launch_status();
}


template <int BM, int BN, int WM, bool ST, bool PB = false>
int launch_layout(const GemmArgs& ga, hipStream_t st) {
    const dpp_gemm_desc& d = ga.d;
    dim3 grid(dpp_cdiv(d.M, BM), dpp_cdiv(d.N, BN), d.splitk);
    const bool k32 = ga.bk == 32;
    if (PB && (d.actA.mode == 4 || ga.bk < 32)) return DPP_E_UNSUPPORTED;       // (dpp_gemm refuses these before it gets here)
    if (d.actA.mode == 4) {
        if (ST) return DPP_E_UNSUPPORTED;            // (gemm_prepare refuses the combination; keeps the LAZY kernels float32-only)
        // data gradient (A [pixels][C] with B = W [K][N]) and filter gradient (both operands [k][mn]) of a 1x1 convolution
        if (d.a_kc && !d.b_kc) {
            if (ga.bk == 64) DPP_LAUNCH((gemm_kernel<BM, BN, WM, 64, true, false, 1, true, true, ST>), grid, dim3(DPP_THREADS), 0, st, ga);
            else if (k32) DPP_LAUNCH((gemm_kernel<BM, BN, WM, 32, true, false, 1, true, true, ST>), grid, dim3(DPP_THREADS), 0, st, ga);
            else DPP_LAUNCH((gemm_kernel<BM, BN, WM, 16, true, false, 1, true, true, ST>), grid, dim3(DPP_THREADS), 0, st, ga);
        } else if (!d.a_kc && !d.b_kc)
            DPP_LAUNCH((gemm_kernel<BM, BN, WM, 64, false, false, 1, true, false, ST>), grid, dim3(DPP_THREADS), 0, st, ga);
        else
            return DPP_E_UNSUPPORTED;
        return dpp_launch_status();
    }
    if (ga.bk == 64 && d.a_kc) {
        // long K-contiguous reductions (stage-2..4 1x1 convolutions and FC layers, K >= 128; measured 4.80 -> 4.74 ms per step): half as many global -> LDS round trips
        if (d.b_kc) DPP_LAUNCH((gemm_kernel<BM, BN, WM, 64, true, true, 1, false, true, ST, PB>), grid, dim3(DPP_THREADS), 0, st, ga);
        else DPP_LAUNCH((gemm_kernel<BM, BN, WM, 64, true, false, 1, false, true, ST, PB>), grid, dim3(DPP_THREADS), 0, st, ga);
    } else if (d.a_kc && d.b_kc) {
        if (k32) DPP_LAUNCH((gemm_kernel<BM, BN, WM, 32, true, true, 1, false, true, ST, PB>), grid, dim3(DPP_THREADS), 0, st, ga);
        else DPP_LAUNCH((gemm_kernel<BM, BN, WM, 16, true, true, 1, false, true, ST, false>), grid, dim3(DPP_THREADS), 0, st, ga);
    } else if (d.a_kc && !d.b_kc) {
        if (k32) DPP_LAUNCH((gemm_kernel<BM, BN, WM, 32, true, false, 1, false, true, ST, PB>), grid, dim3(DPP_THREADS), 0, st, ga);
        else DPP_LAUNCH((gemm_kernel<BM, BN, WM, 16, true, false, 1, false, true, ST, false>), grid, dim3(DPP_THREADS), 0, st, ga);
    } else if (!d.a_kc && !d.b_kc) {
        // reduction over pixels / samples: long K, both operands [k][mn] -> 64-deep chunks keep 20+ KB per workgroup in flight
        if (ga.wide) DPP_LAUNCH((gemm_kernel<BM, BN, WM, 64, false, false, 1, false, true, ST, PB>), grid, dim3(DPP_THREADS), 0, st, ga);
        else DPP_LAUNCH((gemm_kernel<BM, BN, WM, 64, false, false, 1, false, false, ST, PB>), grid, dim3(DPP_THREADS), 0, st, ga);
    } else
        return DPP_E_UNSUPPORTED;
    return dpp_launch_status();
}


// variants 3 / 2 / 1 / 0 of dpp_gemm for a prepared problem.  ST = false: every tensor float32 (gemm.hip); ST = true: some operand /
// output is bf16-stored (gemm_st.hip, its own translation unit: the two sets of instantiations compile side by side)
template <bool ST, bool PB = false>
int gemm_dispatch(GemmArgs& ga, int bm, int bn, int wm, hipStream_t st) {
    dpp_gemm_desc& d = ga.d;
    if (d.variant == 3) {
        const int rows = stream16_rows(d, ga);
        if (!rows) return DPP_E_UNSUPPORTED;
        const bool act = d.actA.mode != 0, epi = d.residual != nullptr || d.epi.bn_x != nullptr;
        // the transposed-accumulator form (16- / 8-byte epilogue accesses) needs aligned quads everywhere; DPP_STREAM16T=0: the round-2 kernel
        static const bool s16t_on = []() { const char* e = getenv("DPP_STREAM16T"); return !(e && e[0] == '0'); }();
        // workgroups of a stream16t launch: two per CU (the walking form is allocated 160-172 registers at K = 64: two or three fit), walking the row blocks; DPP_STREAM16T_WGS overrides, 0: one block each.  bf16 256 x 256 step, same box: one block each 6.55 / 6.57 ms, 512 workgroups 6.47 / 6.46, 768 6.52 / 6.50, 1 024 6.51 / 6.54
        static const int s16t_cap = []() { const char* e = getenv("DPP_STREAM16T_WGS"); const int v = e ? atoi(e) : 512; return v > 0 ? v : (1 << 30); }();
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        // ... and pays on bf16-stored tensors only: same-box A/B of the float32 128 x 128 step 3.462 / 3.465 (transposed) against 3.449 /
        // 3.452 ms (its 4-byte accesses were not the bound there), bf16 256 x 256 7.452 / 7.535 against 7.550 / 7.561 (profiles/r06_ab.txt)
        const bool s16t = s16t_on && ST && (d.store & (DPP_ST_A | DPP_ST_C | DPP_ST_BNX)) != 0 && d.ldc % 4 == 0 && al16(d.C) && al16(d.residual) && al16(d.bias) && al16(d.epi.bn_x) && al16(d.epi.bn_mean) &&
                          al16(d.epi.bn_scale) && al16(d.epi.bn_beta) && al16(d.epi.bn_inv_std);
#define DPP_S16(K_, CN_, T_, B_, A_, E_) do { if (s16t) DPP_LAUNCH((gemm_stream16t_kernel<K_, CN_, B_, T_, A_, E_, ST, PB>), dim3(d.M / rows > s16t_cap ? s16t_cap : d.M / rows), dim3(DPP_THREADS), 0, st, ga); \
            else DPP_LAUNCH((gemm_stream16_kernel<K_, CN_, B_, T_, A_, E_, ST, PB>), dim3(d.M / rows), dim3(DPP_THREADS), 0, st, ga); } while (0)
#define DPP_S16_ALL(K_, CN_, T_) \
        if (d.b_kc) { if (act) { if (epi) DPP_S16(K_, CN_, T_, true, true, true); else DPP_S16(K_, CN_, T_, true, true, false); } \
                      else { if (epi) DPP_S16(K_, CN_, T_, true, false, true); else DPP_S16(K_, CN_, T_, true, false, false); } } \
        else { if (act) { if (epi) DPP_S16(K_, CN_, T_, false, true, true); else DPP_S16(K_, CN_, T_, false, true, false); } \
               else { if (epi) DPP_S16(K_, CN_, T_, false, false, true); else DPP_S16(K_, CN_, T_, false, false, false); } }
        if (rows == 128) { DPP_S16_ALL(64, 1, 2) } else { DPP_S16_ALL(16, 4, 1) }
#undef DPP_S16_ALL
#undef DPP_S16
        return dpp_launch_status();
    }
    if (d.variant == 2) {
        const int kbn = ksplit_bn(d, ga);
        if (!kbn) return DPP_E_UNSUPPORTED;
        if (d.K == 256) return d.b_kc ? launch_ksplit<256, 64, true, ST, PB>(ga, st) : launch_ksplit<256, 64, false, ST, PB>(ga, st);
        return d.b_kc ? launch_ksplit<128, 32, true, ST, PB>(ga, st) : launch_ksplit<128, 32, false, ST, PB>(ga, st);
    }
    if (d.variant == 1) {
        // row-streaming kernel: bm in {64, 128} rows per workgroup, bn in {16, 32, 64} columns, whole K staged for B
        if (!d.a_kc || d.splitk != 1 || ST || PB) return DPP_E_UNSUPPORTED;
        if (bm != 64 && bm != 128) bm = 64;
        if (bn != 16 && bn != 32 && bn != 64) bn = d.N > 32 ? 64 : (d.N > 16 ? 32 : 16);
        const int K16 = (d.K + 15) & ~15;
        size_t lds = (d.b_kc ? (size_t)bn * (K16 + 4) : (size_t)K16 * (bn + 4)) * sizeof(float);
        if (lds < (size_t)4 * bn * sizeof(float)) lds = (size_t)4 * bn * sizeof(float);
        if (lds > 64 * 1024) return DPP_E_UNSUPPORTED;
        dim3 grid(dpp_cdiv(d.M, bm), dpp_cdiv(d.N, bn), 1);
#define DPP_RS(RM_, CN_) if (bm == 64 * RM_ && bn == 16 * CN_) { \
            if (d.b_kc) DPP_LAUNCH((gemm_rowstream_kernel<RM_, CN_, true>), grid, dim3(DPP_THREADS), lds, st, ga); \
            else DPP_LAUNCH((gemm_rowstream_kernel<RM_, CN_, false>), grid, dim3(DPP_THREADS), lds, st, ga); \
            return dpp_launch_status(); }
        DPP_RS(1, 1) DPP_RS(1, 2) DPP_RS(1, 4) DPP_RS(2, 1) DPP_RS(2, 2) DPP_RS(2, 4)
#undef DPP_RS
        return DPP_E_UNSUPPORTED;
    }
#define DPP_TILE(BM_, BN_, WM_) if (bm == BM_ && bn == BN_ && wm == WM_) return launch_layout<BM_, BN_, WM_, ST, PB>(ga, st);
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

}  // namespace
