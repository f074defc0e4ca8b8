// gemm_expand.hip -- dpp_gemm variant 4: the wave-autonomous kernel for the channel-expanding 1x1 convolutions (see include/dpp_hip.h,
// dpp_gemm_desc.variant; the entry points and the other kernels are in gemm.hip).
#include "gemm_args.h"

namespace {

// ---- wave-autonomous variant for the channel-EXPANDING 1x1 convolutions and the data gradients of the reducing ones (variant 4) ----
// K = 16 / 32 / 64 -> N = 64 / 128 / 256 columns over 131 072 / 32 768 / 8 192 pixel rows: the bottleneck exits of resnet.py:369-379,
// 398-414 (bias + residual + the statistics of the tensor they write) and, backwards, the data gradients of the bottleneck entries
// (BatchNorm-backward mask and sums).  These are OUTPUT-bound: 4 x as many bytes leave (and come in as residual / BatchNorm input) as
// the operand brings, the matrix product is 16..64 deep.  On the LDS-tiled kernel a workgroup stages, synchronises four times and
// transposes its accumulators through LDS to reach 16-byte accesses (14.8 us for the 18 MB of a stage-3 exit).  Here a WAVE owns
// `rpw` rows x 64 columns and the whole K, nothing is shared, so there is no LDS and no barrier at all:
//   * lane (l15, kq) owns k = kq*K/4 .. +K/4-1: its A fragment is K/16 consecutive 16-byte loads of pixel row l15 and the BatchNorm +
//     ReLU prologue happens in registers; its B fragment -- the wave's whole 64-column slice of the filter, K/4 x 4 registers -- is
//     loaded once and stays in registers for all rows of the wave;
//   * the 64 columns are dealt INTERLEAVED to the four accumulator tiles: tile ct holds columns 4*l15 + ct.  In the MFMA D layout a
//     lane then owns FOUR ADJACENT columns of each of its rows, one per tile: residual, BatchNorm input and output move as 16-byte
//     accesses straight from / to memory (16 lanes = one 256-byte row segment), and a [K][N] filter row is one 16-byte load per lane;
//   * every load of an iteration (32 rows) -- operand, residual, BatchNorm input -- is issued before the first MFMA; waves with more
//     than 32 rows (stages 1-2) walk them in iterations, the next iteration's loads in flight under the current one;
//   * the column statistics (mean, M2) of an iteration are formed in two passes over the registers and merged into the wave's running
//     (n, mean, M2) by Chan's update in a fixed order; one partial row per wave (block index = the wave's row block).
//   * LZ (data gradient of a convolution whose output feeds a BatchNorm and nothing else): the operand is the gradient through that
//     BatchNorm's batch statistics, formed in registers from the masked gradient G and the BatchNorm input x (dpp_act mode 4, the
//     arithmetic of bnbwd4_masked) -- bn_bwd_apply on the fly; the column-group-0 wave of a row block leaves it in actA.out for
//     the filter gradient.
//   * PB (dpp_gemm_desc.precision = 1, BASELINE config 5; K = 32 / 64): both operands are rounded to bfloat16 (RNE, the activation after its
//     prologue) and multiplied on v_mfma_f32_16x16x32_bf16 with f32 accumulation: a lane's 8 / 16 k values are one / two 8-element
//     operands, so the 128 f32 MFMAs of a 32-row iteration at K = 64 become 16.  The filter slice is packed once per wave.
// KT = K; BKC: B is [N][K] (forward) or [K][N] (data gradient); IT2: two register sets (iterations > 1); ACT / RES / BNB / LZ: register diets.
// ST: the instantiation may meet bf16-stored tensors (the float32 ones carry none of the run-time type tests).
template <int KT, bool BKC, bool IT2, bool ACT, bool RES, bool BNB, bool LZ = false, bool PB = false, bool ST = false>
__global__ __launch_bounds__(DPP_THREADS) DPP_WAVES_PER_EU(1, 2) void gemm_expand_kernel(GemmArgs ga, int rpw) {
    static_assert(!PB || ((KT % 32 == 0 || KT == 16) && !LZ), "bf16 MFMA operands: whole 32-deep steps, or K = 16 with a zero upper half");
    dpp_kernarg_warm<sizeof(GemmArgs)>();
    const dpp_gemm_desc& d = ga.d;
    constexpr int KL = KT / 4, KV = KL / 4;                  // k values / 16-byte vectors of a lane
    constexpr int RT = 2, RI = 16 * RT;                      // row tiles / rows per iteration
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int NG = d.N >> 6, nblk = d.M / rpw;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= NG * nblk) return;
    const int cg = gw % NG, rblk = gw / NG;
    const int n0 = cg * 64 + 4 * l15;                        // this lane's four output columns
    const int wrow0 = rblk * rpw, iters = rpw / RI;
    const dpp_epilogue& ep = d.epi;
    const int modeA = ACT ? d.actA.mode : 0;
    const bool relu_mask = BNB && ep.bn_relu != 0;
    // bf16-stored tensors (DPP_ST_*): A (forward: the activations), C + residual (forward: the output), epi.bn_x (data gradient)
    const bool a16 = ST && !LZ && ga.shA != 0, c16 = ST && (d.store & DPP_ST_C) != 0, x16 = ST && (d.store & DPP_ST_BNX) != 0;
    dpp_stamp(ga.prof, 0);

    // ---- the filter slice, the prologue coefficients of this lane's k range, the per-column vectors of its four columns ----
    float bw[4][KL];
    if (BKC) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int j = 0; j < KV; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(d.B + (size_t)(n0 + ct) * d.ldb + kq * KL + 4 * j);
                bw[ct][4 * j] = t.x; bw[ct][4 * j + 1] = t.y; bw[ct][4 * j + 2] = t.z; bw[ct][4 * j + 3] = t.w;
            }
    } else {
#pragma unroll
        for (int e = 0; e < KL; ++e) {
            const float4 t = *reinterpret_cast<const float4*>(d.B + (size_t)(kq * KL + e) * d.ldb + n0);
            bw[0][e] = t.x; bw[1][e] = t.y; bw[2][e] = t.z; bw[3][e] = t.w;
        }
    }
    constexpr int KS = PB ? (KL + 7) / 8 : 1;                // 8-element bf16 operands per lane and accumulator tile (K = 16: 4 values + 4 zeros)
    dpp_bf16x8 bwp[4][KS];
    if (PB) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int ss = 0; ss < KS; ++ss)
#pragma unroll
                for (int q = 0; q < 8; ++q) bwp[ct][ss][q] = (dpp_bf16)(ss * 8 + q < KL ? bw[ct][ss * 8 + q < KL ? ss * 8 + q : 0] : 0.0f);
    }
    struct Regs { float4 a[RT][KV]; float4 a2[RT][LZ ? KV : 1]; float4 res[RT][4]; float4 bx[RT][4]; };
    auto fetch = [&](Regs& g, int it) {
        const int r0 = wrow0 + it * RI;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const size_t oa = (size_t)(r0 + rt * 16 + l15) * d.lda + kq * KL;
            if (a16) {                                      // bf16-stored activations: 8-byte loads, widened where they are consumed
#pragma unroll
                for (int j = 0; j < KV; ++j) g.a[rt][j] = dpp_raw8(d.A + ((oa + 4 * j) >> 1));
            } else {
#pragma unroll
                for (int j = 0; j < KV; ++j) {
                    g.a[rt][j] = *reinterpret_cast<const float4*>(d.A + oa + 4 * j);
                    if (LZ) g.a2[rt][j] = *reinterpret_cast<const float4*>(d.actA.x2 + oa + 4 * j);
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(r0 + rt * 16 + kq * 4 + r) * d.ldc + n0;
                if (RES) g.res[rt][r] = c16 ? dpp_raw8(d.residual + (o >> 1)) : *reinterpret_cast<const float4*>(d.residual + o);
                if (BNB) g.bx[rt][r] = x16 ? dpp_raw8(ep.bn_x + (o >> 1)) : *reinterpret_cast<const float4*>(ep.bn_x + o);
            }
    };
    Regs g0, g1;
    fetch(g0, 0);
    float4 mu[KV], sc[KV], be[KV], ax[LZ ? KV : 1];
#pragma unroll
    for (int j = 0; j < KV; ++j) {
        mu[j] = sc[j] = be[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LZ || (modeA & 2)) {
            mu[j] = *reinterpret_cast<const float4*>(d.actA.mean + kq * KL + 4 * j);
            sc[j] = *reinterpret_cast<const float4*>(d.actA.scale + kq * KL + 4 * j);
            be[j] = *reinterpret_cast<const float4*>(d.actA.beta + kq * KL + 4 * j);
        }
        if (LZ) ax[j] = *reinterpret_cast<const float4*>(d.actA.aux + kq * KL + 4 * j);
    }
    const bool lz_store = LZ && d.actA.out != nullptr && cg == 0;
    float4 cbias = make_float4(0.f, 0.f, 0.f, 0.f), cmean = cbias, cscale = cbias, cbeta = cbias, cistd = cbias;
    if (d.bias) cbias = *reinterpret_cast<const float4*>(d.bias + n0);
    if (BNB) {
        cmean = *reinterpret_cast<const float4*>(ep.bn_mean + n0); cscale = *reinterpret_cast<const float4*>(ep.bn_scale + n0);
        cbeta = *reinterpret_cast<const float4*>(ep.bn_beta + n0); cistd = *reinterpret_cast<const float4*>(ep.bn_inv_std + n0);
    }
    dpp_stamp(ga.prof, 1);

    float sx[4] = {0.f, 0.f, 0.f, 0.f}, sy[4] = {0.f, 0.f, 0.f, 0.f};     // BatchNorm-backward sums of the wave's rows
    float rmean[4] = {0.f, 0.f, 0.f, 0.f}, rm2[4] = {0.f, 0.f, 0.f, 0.f};   // running statistics of the rows done so far
    const bool want_stats = ep.stats != nullptr;
    auto colsum = [&](float (&s)[4]) {                       // over the 4 lane groups (rows); every lane gets the total
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += __shfl_xor(s[j], 16); s[j] += __shfl_xor(s[j], 32); }
    };

    auto step = [&](Regs& g, int it) {
        const int r0 = wrow0 + it * RI;
        f32x4 acc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dpp_bf16x8 ap[RT][KS];
        if (PB && KL < 8) {                                  // K = 16: the upper half of the 32-deep step is zero
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int q = 0; q < 8; ++q) ap[rt][0][q] = (dpp_bf16)0.0f;
        }
#pragma unroll
        for (int j = 0; j < KV; ++j) {
            float av[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float4 v = a16 ? dpp_widen4(g.a[rt][j]) : g.a[rt][j];
                if (LZ) {
                    const float4 x = g.a2[rt][j];
                    v = make_float4(sc[j].x * v.x - ax[j].x * (x.x - mu[j].x) - be[j].x, sc[j].y * v.y - ax[j].y * (x.y - mu[j].y) - be[j].y,
                                    sc[j].z * v.z - ax[j].z * (x.z - mu[j].z) - be[j].z, sc[j].w * v.w - ax[j].w * (x.w - mu[j].w) - be[j].w);
                    if (lz_store) *reinterpret_cast<float4*>(d.actA.out + (size_t)(r0 + rt * 16 + l15) * d.lda + kq * KL + 4 * j) = v;
                }
                if (modeA & 2) {
                    v.x = dpp_fma(v.x - mu[j].x, sc[j].x, be[j].x); v.y = dpp_fma(v.y - mu[j].y, sc[j].y, be[j].y);
                    v.z = dpp_fma(v.z - mu[j].z, sc[j].z, be[j].z); v.w = dpp_fma(v.w - mu[j].w, sc[j].w, be[j].w);
                }
                if (modeA & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                av[rt][0] = v.x; av[rt][1] = v.y; av[rt][2] = v.z; av[rt][3] = v.w;
                if (PB) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) ap[rt][j / 2][(j & 1) * 4 + t] = (dpp_bf16)av[rt][t];
                }
            }
            if (PB) {
                if ((j & 1) || KL < 8) {                     // the 8 k values 8 (j/2) .. +7 of every row tile are complete: one 32-deep step
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[rt][j / 2], bwp[ct][j / 2], acc[rt][ct], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][t], bw[ct][4 * j + t], acc[rt][ct], 0, 0, 0);
            }
        }
        // D layout: register r of tile (rt, ct) = row rt*16 + kq*4 + r, column n0 + ct: four adjacent columns per (rt, r)
        float vals[RT][4][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[4] = {acc[rt][0][r] + cbias.x, acc[rt][1][r] + cbias.y, acc[rt][2][r] + cbias.z, acc[rt][3][r] + cbias.w};
                if (RES) {
                    const float4 rr = c16 ? dpp_widen4(g.res[rt][r]) : g.res[rt][r];
                    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                }
                if (BNB) {
                    const float4 bxv = x16 ? dpp_widen4(g.bx[rt][r]) : g.bx[rt][r];
                    const float x[4] = {bxv.x, bxv.y, bxv.z, bxv.w};
                    const float cm[4] = {cmean.x, cmean.y, cmean.z, cmean.w}, cs[4] = {cscale.x, cscale.y, cscale.z, cscale.w};
                    const float cb[4] = {cbeta.x, cbeta.y, cbeta.z, cbeta.w}, ci[4] = {cistd.x, cistd.y, cistd.z, cistd.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dx = x[j] - cm[j];
                        if (relu_mask && dx * cs[j] + cb[j] < 0.0f) v[j] = 0.0f;
                        if (c16) v[j] = dpp_bf16_round(v[j]);   // bf16-stored gradient: sums of the values as stored (dpp_epilogue_wide)
                        sx[j] += v[j];
                        sy[j] += v[j] * (dx * ci[j]);
                    }
                }
                {
                    const size_t oc = (size_t)(r0 + rt * 16 + kq * 4 + r) * d.ldc + n0;
                    if (c16) dpp_st4(reinterpret_cast<dpp_bf16*>(d.C) + oc, make_float4(v[0], v[1], v[2], v[3]));      // rounded on the store;
                    else *reinterpret_cast<float4*>(d.C + oc) = make_float4(v[0], v[1], v[2], v[3]);                      // statistics below from the f32 values
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) vals[rt][r][j] = v[j];
            }
        if (want_stats) {
            // this iteration's (mean, M2) in two passes over the registers, then Chan's update of the running pair (fixed order)
            float sm[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sm[j] += vals[rt][r][j];
            colsum(sm);
#pragma unroll
            for (int j = 0; j < 4; ++j) sm[j] *= 1.0f / (float)RI;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float dv = vals[rt][r][j] - sm[j]; m2[j] += dv * dv; }
            colsum(m2);
            if (it == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { rmean[j] = sm[j]; rm2[j] = m2[j]; }
            } else {
                const float na = (float)(it * RI), nb = (float)RI, nn = na + nb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dl = sm[j] - rmean[j];
                    rmean[j] += dl * (nb / nn);
                    rm2[j] += m2[j] + dl * dl * (na * nb / nn);
                }
            }
        }
    };

    if (IT2) {
        for (int it = 0; it < iters; it += 2) {
            if (it + 1 < iters) fetch(g1, it + 1);
            DPP_SCHED_FENCE();
            step(g0, it);
            if (it + 1 < iters) {
                if (it + 2 < iters) fetch(g0, it + 2);
                DPP_SCHED_FENCE();
                step(g1, it + 1);
            }
        }
    } else {
        DPP_SCHED_FENCE();
        step(g0, 0);
    }
    dpp_stamp(ga.prof, 3);
    if (BNB && ep.bn_partial != nullptr) {
        // every lane group holds partial sums over ITS rows: total over the four groups
        float tx[4] = {sx[0], sx[1], sx[2], sx[3]}, ty[4] = {sy[0], sy[1], sy[2], sy[3]};
        colsum(tx);
        colsum(ty);
        if (kq == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ep.bn_partial[dpp_partial_index(0, n0 + j, rblk, d.N, nblk)] = tx[j];
                ep.bn_partial[dpp_partial_index(1, n0 + j, rblk, d.N, nblk)] = ty[j];
            }
        }
    }
    if (want_stats && kq == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ep.stats[dpp_partial_index(0, n0 + j, rblk, d.N, nblk)] = rmean[j];
            ep.stats[dpp_partial_index(1, n0 + j, rblk, d.N, nblk)] = rm2[j];
        }
    }
    dpp_stamp(ga.prof, 4);
}

// dpp_gemm variant 4: rows per wave (= rows per statistics block) of gemm_expand_kernel for this problem, or 0.  d.bm carries the
// caller's wish (a multiple of 32 that divides M), 0 = choose by the row count.
}  // namespace

int dpp_gemm_expand_rows(const dpp_gemm_desc& d, const GemmArgs& ga) {
    if ((d.store & DPP_ST_B) || (d.store && (d.actA.mode & 4))) return 0;
    if (!d.a_kc || d.splitk != 1 || !d.C || !ga.vecA || !ga.vecB || !ga.wide) return 0;
    if (d.K != 16 && d.K != 32 && d.K != 64) return 0;
    if (d.N % 64 || d.M % 32) return 0;
    if (d.mapA.s != 1 || d.mapB.s != 1 || d.mapC.s != 1 || d.actB.mode != 0) return 0;
    if (d.actA.mode != 0 && d.actA.cmod < d.K) return 0;
    if (d.epi.stats && d.epi.bn_x) return 0;
    // instantiated: [N][K] filters (forward) with the modes 0-3 prologue and no BatchNorm-backward epilogue; [K][N] filters (data
    // gradient) with a plain or a mode-4 operand (gemm_prepare has checked mode 4's vectors and alignment)
    if (d.b_kc ? (d.epi.bn_x != nullptr || (d.actA.mode & 4)) : (d.actA.mode != 0 && d.actA.mode != 4)) return 0;
    if (d.actA.mode == 4 && d.actA.out && d.lda != d.K) return 0;
    if (d.precision != 0 && (d.precision != 1 || (d.actA.mode & 4))) return 0;      // bf16 MFMA operands (round 6: K = 16 too, zero upper half): no mode-4 operand
    int rpw = d.bm;
    if (rpw <= 0) rpw = d.M >= 65536 ? 128 : (d.M >= 16384 ? 64 : 32);
    if (rpw % 32 || d.M % rpw) return 0;
    return rpw;
}

namespace {

template <int KT>
int launch_expand(const GemmArgs& ga, int rpw, hipStream_t st) {
    const dpp_gemm_desc& d = ga.d;
    const int waves = (d.M / rpw) * (d.N / 64);
    const dim3 grid(dpp_cdiv(waves, 4));
    const bool it2 = rpw > 32, act = d.actA.mode != 0, res = d.residual != nullptr, bnb = d.epi.bn_x != nullptr, lz = d.actA.mode == 4;
    const bool pb = d.precision == 1;
    const bool stg = d.store != 0;
#define DPP_EXS(K_, I_, A_, R_, B_, L_, P_, S_) DPP_LAUNCH((gemm_expand_kernel<KT, K_, I_, A_, R_, B_, L_, P_, S_>), grid, dim3(DPP_THREADS), 0, st, ga, rpw)
#define DPP_EX(K_, I_, A_, R_, B_, L_, P_) do { if (!(L_) && stg) DPP_EXS(K_, I_, A_, R_, B_, L_, P_, !(L_)); else DPP_EXS(K_, I_, A_, R_, B_, L_, P_, false); } while (0)
#define DPP_EX_P(K_, I_, A_, R_, B_) do { if (pb) { DPP_EX(K_, I_, A_, R_, B_, false, true); break; } \
                                          DPP_EX(K_, I_, A_, R_, B_, false, false); } while (0)
#define DPP_EX_I(K_, A_, R_, B_, L_) do { if (L_) { if (it2) DPP_EX(K_, true, A_, R_, B_, true, false); else DPP_EX(K_, false, A_, R_, B_, true, false); } \
                                          else if (it2) DPP_EX_P(K_, true, A_, R_, B_); else DPP_EX_P(K_, false, A_, R_, B_); } while (0)
#define DPP_EX_R(K_, A_, B_, L_) do { if (res) DPP_EX_I(K_, A_, true, B_, L_); else DPP_EX_I(K_, A_, false, B_, L_); } while (0)
    if (d.b_kc) {
        if (act) DPP_EX_R(true, true, false, false); else DPP_EX_R(true, false, false, false);
    } else if (lz) {
        if (bnb) DPP_EX_R(false, false, true, true); else DPP_EX_R(false, false, false, true);
    } else {
        if (bnb) DPP_EX_R(false, false, true, false); else DPP_EX_R(false, false, false, false);
    }
#undef DPP_EX_P
#undef DPP_EXS
#undef DPP_EX_R
#undef DPP_EX_I
#undef DPP_EX
    return dpp_launch_status();
}

}  // namespace

int dpp_gemm_expand_launch(const GemmArgs& ga, int rpw, hipStream_t st) {
    const dpp_gemm_desc& d = ga.d;
    return d.K == 64 ? launch_expand<64>(ga, rpw, st) : (d.K == 32 ? launch_expand<32>(ga, rpw, st) : launch_expand<16>(ga, rpw, st));
}
