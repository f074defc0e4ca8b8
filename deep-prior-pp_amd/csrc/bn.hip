// bn.hip -- BatchNormLayer statistics, coefficients and backward for gfx950 (HBM-bound kernels).
//
// Reference arithmetic: /root/reference/src/net/batchnormlayer.py:119-194
//   train: mean / BIASED variance over (N,H,W); inv_std = 1/sqrt(var + 1e-4); y = (x-mean)*(gamma*inv_std)+beta;
//          running mean / running INV-STD EMA with alpha = 0.1;   eval: stored mean / inv_std.
// The normalisation itself is never materialised: consumers apply it as an operand prologue (dpp_act).
//
// All tensors are pixel-major [M][C] (C % 4 == 0), so a wave reads whole 64..1024-byte rows with float4
// lanes (coalesced).  Statistics are two-level: per-thread shifted sums -> per-block (mean, M2) combined
// with Chan's parallel formula in f64 -> a finalize kernel that walks the per-block partials in a FIXED
// order (deterministic, no atomics).  This keeps the variance accurate when |mean| >> std.
#include "dpp_common.h"

namespace {

constexpr int MAXQ = 256;   // float4 lanes per row => C <= 1024

// TX: element type of the activation tensor (float, or dpp_bf16 in the bf16 storage mode)
template <class TX>
__global__ __launch_bounds__(DPP_THREADS) void bn_stats_partial_kernel(const TX* __restrict__ X, int M, int C,
                                                                       int rpb, float* __restrict__ partial) {
    __shared__ float s_mean[DPP_THREADS * 4];
    __shared__ float s_m2[DPP_THREADS * 4];
    __shared__ float s_n[DPP_THREADS];
    const int Q = C >> 2;                    // float4 lanes per row
    const int RP = DPP_THREADS / Q;          // rows per pass
    const int tid = threadIdx.x;
    const int q = tid % Q, rr = tid / Q;
    const int r_begin = blockIdx.x * rpb;
    const int r_end = (r_begin + rpb < M) ? r_begin + rpb : M;
    float4 K = make_float4(0.f, 0.f, 0.f, 0.f), s1 = K, s2 = K;
    int n = 0;
    if (rr < RP) {
        for (int r = r_begin + rr; r < r_end; r += RP) {
            float4 v = dpp_ld4(X + (size_t)r * C + q * 4);
            if (n == 0) K = v;
            float dx = v.x - K.x, dy = v.y - K.y, dz = v.z - K.z, dw = v.w - K.w;
            s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
            s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
            ++n;
        }
    }
    float inv = n > 0 ? 1.0f / (float)n : 0.0f;
    s_n[tid] = (float)n;
    s_mean[tid * 4 + 0] = K.x + s1.x * inv; s_m2[tid * 4 + 0] = s2.x - s1.x * s1.x * inv;
    s_mean[tid * 4 + 1] = K.y + s1.y * inv; s_m2[tid * 4 + 1] = s2.y - s1.y * s1.y * inv;
    s_mean[tid * 4 + 2] = K.z + s1.z * inv; s_m2[tid * 4 + 2] = s2.z - s1.z * s1.z * inv;
    s_mean[tid * 4 + 3] = K.w + s1.w * inv; s_m2[tid * 4 + 3] = s2.w - s1.w * s1.w * inv;
    __syncthreads();
    for (int c = tid; c < C; c += DPP_THREADS) {
        int cq = c >> 2, ce = c & 3;
        double cn = 0.0, mean = 0.0, m2 = 0.0;
        for (int j = 0; j < RP; ++j) {
            int t = j * Q + cq;
            double nb = (double)s_n[t];
            if (nb == 0.0) continue;
            double mb = (double)s_mean[t * 4 + ce], m2b = (double)s_m2[t * 4 + ce];
            double delta = mb - mean, tot = cn + nb;
            mean += delta * nb / tot;
            m2 += m2b + delta * delta * cn * nb / tot;
            cn = tot;
        }
        partial[dpp_partial_index(0, c, blockIdx.x, C, gridDim.x)] = (float)mean;
        partial[dpp_partial_index(1, c, blockIdx.x, C, gridDim.x)] = (float)m2;
    }
}

// One wave per channel, lane l owns partial blocks l, l+64, ... (contiguous in memory, see dpp_partial_index).  The per-block
// (mean, M2) pairs are combined instead of chained through pairwise Chan updates (a dependent f64 division per partial is what the
// finalize of a stage-1 layer -- 2048 partial blocks, 8192 for the stem -- spent its time on once the loads were coalesced):
//   * single rank, all blocks in one sweep (every training launch of the nets here): ONE pass about a pivot K = block 0's mean,
//         mean = K + S1 / M,   M2 = S2 - S1^2 / M,   S1 = sum_b n_b (mean_b - K),   S2 = sum_b [M2_b + n_b (mean_b - K)^2]
//     The subtraction S2 - S1^2 / M is there again (round 5), but what cancels is the spread of the BLOCK means about K -- not
//     |mean| >> std --, in f64: relative loss ~1e-16 (spread / std)^2, i.e. nothing until block 0 lies ~1e4 standard deviations away
//     from the others (tests/test_kernels.py: an outlier block 0);
//   * otherwise (sync-BN segments, more blocks than one sweep): two passes, mean = sum_b n_b mean_b / M first, then
//         M2 = sum_b [M2_b + n_b (mean_b - mean)^2]  -- deviations from the final mean, no cancellation at all.
// All sums are f64 in a fixed order; lanes meet in an xor butterfly (every lane ends with the same value).
__device__ __forceinline__ double wave_allsum(double v) {
    for (int o = 1; o < DPP_WAVE; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over the WPC waves that share a channel (WPC = 1: the wave; WPC = 4: the whole workgroup, through LDS in wave order).
// Every thread of the workgroup must call it; `slot` is a distinct LDS array per call site.
template <int WPC>
__device__ __forceinline__ double group_allsum(double v, double* slot) {
    v = wave_allsum(v);
    if (WPC == 1) return v;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) slot[wave] = v;
    __syncthreads();
    return ((slot[0] + slot[1]) + slot[2]) + slot[3];
}

template <int U, int WPC>
__global__ __launch_bounds__(DPP_THREADS) void bn_finalize_kernel(const float* __restrict__ partial, int nbs, int nseg, int M, int rpb,
                                                                  int C, const float* __restrict__ gamma, float eps,
                                                                  float* __restrict__ mean_o, float* __restrict__ inv_std_o,
                                                                  float* __restrict__ scale_o, float* __restrict__ run_mean,
                                                                  float* __restrict__ run_inv_std, float alpha) {
    // WPC waves per channel (4 = the whole workgroup, used when there are many partial blocks); `lane` is the thread's
    // index among the WPC * 64 threads of its channel
    dpp_kernarg_warm<96>();
    __shared__ double slot_a[4], slot_b[4];
    constexpr int GT = DPP_WAVE * WPC;
    const int lane = threadIdx.x % GT;
    const int c = blockIdx.x * (DPP_THREADS / GT) + threadIdx.x / GT;
    const int cc = c < C ? c : C - 1;                  // keep every lane in the shuffles
    // what the last lines need from memory is fetched NOW, with the partials: loaded where it is used it is one more dependent
    // round trip at the end of a kernel that consists of nothing else
    const bool upd = alpha > 0.0f && run_mean != nullptr;
    const float gam = gamma[cc];
    const float rm_old = (upd ? run_mean : gamma)[cc], ri_old = (upd ? run_inv_std : gamma)[cc];
    // `partial` holds nseg segments (the ranks of a sync-BN all-gather) of nbs blocks each; a lane walks blocks lane, lane+64,
    // ... of every segment (no integer division per partial: this kernel is a few microseconds of pure latency)
    const int Mseg = M / nseg;
    auto rows_of = [&](int bi) { return (bi * rpb + rpb <= Mseg) ? rpb : (Mseg - bi * rpb); };
    // One sweep covers all blocks in every single-rank case (nbs <= GT * U by the host's choice of U / WPC): (mean_b, M2_b) are then
    // loaded ONCE, both in the first round trip, and the second pass runs from registers -- this kernel is nothing but dependent
    // memory latency, so a saved round trip is a fifth of its run time (tools/gemm_micro.py floor).
    const bool one_sweep = nseg == 1 && nbs <= GT * U;
    float pm0[U], pq0[U];
    if (one_sweep) {
        // Round 5: ONE pass.  With the block means taken about a pivot K (block 0's mean, a broadcast load in the same round trip),
        //     mean = K + S1 / M,  M2 = S2 - S1^2 / M,   S1 = sum_b n_b d_b,  S2 = sum_b [M2_b + n_b d_b^2],  d_b = mean_b - K
        // the two sums are INDEPENDENT: their cross-lane butterflies interleave instead of the second one (deviations from the final
        // mean) waiting for the first -- the forward finalize took 4.1-4.9 us against the backward finalize's 3.0-3.5 us for the same
        // volume (profiles/r04_instruction_mix.txt), and this dependent reduction was the difference.  The subtraction cancels only
        // the spread of the BLOCK means about K (not |mean| >> std), in f64: what is lost is ~1e-16 (spread / std)^2.
        const float* pm_row = partial + dpp_partial_index(0, cc, 0, C, nbs);
        const float* pq_row = partial + dpp_partial_index(1, cc, 0, C, nbs);
        const float K = pm_row[0];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = lane + u * GT;
            const int bb = b < nbs ? b : nbs - 1;
            pm0[u] = pm_row[bb];
            pq0[u] = pq_row[bb];
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = lane + u * GT;
            if (b < nbs) {
                const double d = (double)pm0[u] - (double)K, nb_ = (double)rows_of(b);
                s1 += nb_ * d;
                s2 += (double)pq0[u] + nb_ * d * d;
            }
        }
        s1 = group_allsum<WPC>(s1, slot_a);
        s2 = group_allsum<WPC>(s2, slot_b);
        if (lane != 0 || c >= C) return;
        const double mean1 = (double)K + s1 / (double)M;
        const double var1 = (s2 - s1 * s1 / (double)M) / (double)M;       // biased, T.var
        const float meanf = (float)mean1;
        const float inv_std = (float)(1.0 / sqrt((var1 > 0.0 ? var1 : 0.0) + (double)eps));
        mean_o[c] = meanf;
        inv_std_o[c] = inv_std;
        scale_o[c] = gam * inv_std;
        if (upd) {
            const float oma = 1.0f - alpha;                 // (1. - alpha) in floatX, batchnormlayer.py:165-172
            run_mean[c] = oma * rm_old + alpha * meanf;
            run_inv_std[c] = oma * ri_old + alpha * inv_std;
        }
        return;
    }
    double snm = 0.0;
    for (int seg = 0; seg < nseg; ++seg) {
        const float* pm_row = partial + (size_t)seg * 2 * C * nbs + dpp_partial_index(0, cc, 0, C, nbs);
        const float* pq_row = partial + (size_t)seg * 2 * C * nbs + dpp_partial_index(1, cc, 0, C, nbs);
        for (int b0 = lane; b0 < nbs; b0 += GT * U) {
            float pm[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + u * GT;
                const int bb = b < nbs ? b : nbs - 1;
                pm[u] = pm_row[bb];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + u * GT;
                if (b < nbs) snm += (double)rows_of(b) * (double)pm[u];
            }
        }
    }
    const double mean = group_allsum<WPC>(snm, slot_a) / (double)M;  // sum_b n_b = M
    double q = 0.0;
    {
        for (int seg = 0; seg < nseg; ++seg) {
            const float* pm_row = partial + (size_t)seg * 2 * C * nbs + dpp_partial_index(0, cc, 0, C, nbs);
            const float* pq_row = partial + (size_t)seg * 2 * C * nbs + dpp_partial_index(1, cc, 0, C, nbs);
            for (int b0 = lane; b0 < nbs; b0 += GT * U) {
                float pm[U], pq[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int b = b0 + u * GT;
                    const int bb = b < nbs ? b : nbs - 1;
                    pm[u] = pm_row[bb];
                    pq[u] = pq_row[bb];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int b = b0 + u * GT;
                    if (b < nbs) {
                        const double d = (double)pm[u] - mean;
                        q += (double)pq[u] + (double)rows_of(b) * d * d;
                    }
                }
            }
        }
    }
    const double m2 = group_allsum<WPC>(q, slot_b);
    if (lane != 0 || c >= C) return;
    double var = m2 / (double)M;                       // biased, T.var
    float meanf = (float)mean;
    float inv_std = (float)(1.0 / sqrt(var + (double)eps));
    mean_o[c] = meanf;
    inv_std_o[c] = inv_std;
    scale_o[c] = gam * inv_std;
    if (upd) {
        float oma = 1.0f - alpha;                        // (1. - alpha) in floatX, batchnormlayer.py:165-172
        run_mean[c] = oma * rm_old + alpha * meanf;
        run_inv_std[c] = oma * ri_old + alpha * inv_std;
    }
}

__global__ __launch_bounds__(DPP_THREADS) void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ run_mean,
                                                                     const float* __restrict__ run_inv_std, int C,
                                                                     float* __restrict__ mean_o, float* __restrict__ inv_std_o,
                                                                     float* __restrict__ scale_o) {
    int c = blockIdx.x * DPP_THREADS + threadIdx.x;
    if (c >= C) return;
    mean_o[c] = run_mean[c];
    inv_std_o[c] = run_inv_std[c];
    scale_o[c] = gamma[c] * run_inv_std[c];
}

// the same for every BatchNorm of a net in one launch: job j covers blocks [block0_j, block0_{j+1})
struct BnEvalJob { const float* gamma; const float* run_mean; const float* run_inv_std; float* mean; float* inv_std; float* scale; int C, block0; };
__global__ __launch_bounds__(DPP_THREADS) void bn_eval_coeffs_multi_kernel(const BnEvalJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                       // the last job whose block0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= jobs[mid].block0) lo = mid; else hi = mid - 1;
    }
    const BnEvalJob jb = jobs[lo];
    const int c = ((int)blockIdx.x - jb.block0) * DPP_THREADS + threadIdx.x;
    if (c >= jb.C) return;
    jb.mean[c] = jb.run_mean[c];
    jb.inv_std[c] = jb.run_inv_std[c];
    jb.scale[c] = jb.gamma[c] * jb.run_inv_std[c];
}

// G = dA * [bn(x) >= 0]  (Theano's Maximum.grad passes the gradient where out == x, i.e. v >= 0) and the
// per-block partial sums of  sum(G), sum(G * xhat)  needed by the BN backward.
// TG: element type of the gradient tensors dA / G (float, or dpp_bf16 when the gradients of the activation tensors are bf16-stored too:
// G is rounded on the store, the sums are those of the unrounded values)
template <class TX, class TG>
__global__ __launch_bounds__(DPP_THREADS) void bn_bwd_reduce_kernel(const TG* dA, const TX* __restrict__ X, int M, int C,
                                                                    const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                                    const float* __restrict__ scale, const float* __restrict__ beta,
                                                                    int relu, TG* G, int rpb, float* __restrict__ partial) {
    dpp_kernarg_warm<128>();
    __shared__ float s_a[DPP_THREADS * 4];
    __shared__ float s_b[DPP_THREADS * 4];
    const int Q = C >> 2, RP = DPP_THREADS / Q;
    const int tid = threadIdx.x, q = tid % Q, rr = tid / Q;
    const int r_begin = blockIdx.x * rpb;
    const int r_end = (r_begin + rpb < M) ? r_begin + rpb : M;
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    if (rr < RP) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + q * 4);
        const float4 is = *reinterpret_cast<const float4*>(inv_std + q * 4);
        const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4);
        const float4 be = *reinterpret_cast<const float4*>(beta + q * 4);
        for (int r = r_begin + rr; r < r_end; r += RP) {
            size_t o = (size_t)r * C + q * 4;
            float4 x = dpp_ld4(X + o);
            float4 g = dpp_ld4(dA + o);
            float dx = x.x - mu.x, dy = x.y - mu.y, dz = x.z - mu.z, dw = x.w - mu.w;
            if (relu) {
                if (dx * sc.x + be.x < 0.0f) g.x = 0.0f;
                if (dy * sc.y + be.y < 0.0f) g.y = 0.0f;
                if (dz * sc.z + be.z < 0.0f) g.z = 0.0f;
                if (dw * sc.w + be.w < 0.0f) g.w = 0.0f;
            }
            dpp_st4(G + o, g);
            if (sizeof(TG) == 2) {               // bf16-stored gradient: the sums are those of the values as stored (dpp_epilogue_wide)
                g.x = dpp_bf16_round(g.x); g.y = dpp_bf16_round(g.y); g.z = dpp_bf16_round(g.z); g.w = dpp_bf16_round(g.w);
            }
            sa.x += g.x; sa.y += g.y; sa.z += g.z; sa.w += g.w;
            sb.x += g.x * (dx * is.x); sb.y += g.y * (dy * is.y); sb.z += g.z * (dz * is.z); sb.w += g.w * (dw * is.w);
        }
    }
    s_a[tid * 4 + 0] = sa.x; s_a[tid * 4 + 1] = sa.y; s_a[tid * 4 + 2] = sa.z; s_a[tid * 4 + 3] = sa.w;
    s_b[tid * 4 + 0] = sb.x; s_b[tid * 4 + 1] = sb.y; s_b[tid * 4 + 2] = sb.z; s_b[tid * 4 + 3] = sb.w;
    __syncthreads();
    for (int c = tid; c < C; c += DPP_THREADS) {
        int cq = c >> 2, ce = c & 3;
        double a = 0.0, b = 0.0;
        for (int j = 0; j < RP; ++j) {
            int t = j * Q + cq;
            a += (double)s_a[t * 4 + ce];
            b += (double)s_b[t * 4 + ce];
        }
        partial[dpp_partial_index(0, c, blockIdx.x, C, gridDim.x)] = (float)a;
        partial[dpp_partial_index(1, c, blockIdx.x, C, gridDim.x)] = (float)b;
    }
}

template <int U, int WPC>
__global__ __launch_bounds__(DPP_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nbs, int nseg, int M, int C,
                                                                      float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                                      float* __restrict__ c1, float* __restrict__ c2,
                                                                      const float* __restrict__ inv_std, const float* __restrict__ scale,
                                                                      float* __restrict__ q, float* __restrict__ p) {
    dpp_kernarg_warm<96>();
    __shared__ double slot_a[4], slot_b[4];
    constexpr int GT = DPP_WAVE * WPC;
    const int lane = threadIdx.x % GT;
    const int c = blockIdx.x * (DPP_THREADS / GT) + threadIdx.x / GT;
    const int cc = c < C ? c : C - 1;
    double a = 0.0, b = 0.0;
    for (int seg = 0; seg < nseg; ++seg) {
        const float* pa_row = partial + (size_t)seg * 2 * C * nbs + dpp_partial_index(0, cc, 0, C, nbs);
        const float* pb_row = partial + (size_t)seg * 2 * C * nbs + dpp_partial_index(1, cc, 0, C, nbs);
        for (int k0 = lane; k0 < nbs; k0 += GT * U) {
            float pa[U], pb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * GT;
                const int kk = k < nbs ? k : nbs - 1;
                pa[u] = pa_row[kk];
                pb[u] = pb_row[kk];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + u * GT < nbs) { a += (double)pa[u]; b += (double)pb[u]; }
        }
    }
    a = group_allsum<WPC>(a, slot_a);
    b = group_allsum<WPC>(b, slot_b);
    if (lane != 0 || c >= C) return;
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
    const float c1f = (float)(a / (double)M), c2f = (float)(b / (double)M);
    c1[c] = c1f;
    c2[c] = c2f;
    if (q != nullptr) {                                 // constants of the mode-4 operand prologue (dpp_act)
        q[c] = scale[c] * c1f;
        p[c] = scale[c] * inv_std[c] * c2f;
    }
}

// dX = scale * (G - c1 - xhat * c2) (+ add): gradient through the batch statistics.  Row-chunked like the reductions so
// that the column sums of dX -- the bias gradient of the conv that produced X (T.grad of `+ b.dimshuffle`,
// convlayer.py:238) -- can be emitted as per-block partials in the same pass (colsum != nullptr).
template <class TX, class TG, class TO>
__global__ __launch_bounds__(DPP_THREADS) void bn_bwd_apply_kernel(const TG* __restrict__ G, const TX* __restrict__ X,
                                                                   int M, int C, const float* __restrict__ mean,
                                                                   const float* __restrict__ inv_std, const float* __restrict__ scale,
                                                                   const float* __restrict__ c1, const float* __restrict__ c2,
                                                                   const TO* add, TO* dX, int rpb, float* __restrict__ colsum) {
    dpp_kernarg_warm<128>();
    __shared__ float s_a[DPP_THREADS * 4];
    const int Q = C >> 2, RP = DPP_THREADS / Q;
    const int tid = threadIdx.x, q = tid % Q, rr = tid / Q;
    const int r_begin = blockIdx.x * rpb;
    const int r_end = (r_begin + rpb < M) ? r_begin + rpb : M;
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rr < RP) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + q * 4);
        const float4 is = *reinterpret_cast<const float4*>(inv_std + q * 4);
        const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4);
        const float4 a1 = *reinterpret_cast<const float4*>(c1 + q * 4);
        const float4 a2 = *reinterpret_cast<const float4*>(c2 + q * 4);
        auto one = [&](const float4& g, const float4& x, const float4& r4) {
            float4 o;
            o.x = sc.x * (g.x - a1.x - (x.x - mu.x) * is.x * a2.x);
            o.y = sc.y * (g.y - a1.y - (x.y - mu.y) * is.y * a2.y);
            o.z = sc.z * (g.z - a1.z - (x.z - mu.z) * is.z * a2.z);
            o.w = sc.w * (g.w - a1.w - (x.w - mu.w) * is.w * a2.w);
            if (add) { o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w; }
            return o;
        };
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int r = r_begin + rr;
        // four rows in flight per thread: the loads of a row do not depend on the previous one
        for (; r + 3 * RP < r_end; r += 4 * RP) {
            float4 g[4], x[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = (size_t)(r + u * RP) * Q + q;
                g[u] = dpp_ld4(G + 4 * i);
                x[u] = dpp_ld4(X + 4 * i);
                o[u] = add ? dpp_ld4(add + 4 * i) : zero4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = (size_t)(r + u * RP) * Q + q;
                o[u] = one(g[u], x[u], o[u]);
                dpp_st4(dX + 4 * i, o[u]);
                if (sizeof(TO) == 2) {                           // bf16-stored dX: the column sums (a bias gradient) are those of the values as stored
                    o[u].x = dpp_bf16_round(o[u].x); o[u].y = dpp_bf16_round(o[u].y); o[u].z = dpp_bf16_round(o[u].z); o[u].w = dpp_bf16_round(o[u].w);
                }
                sa.x += o[u].x; sa.y += o[u].y; sa.z += o[u].z; sa.w += o[u].w;
            }
        }
        for (; r < r_end; r += RP) {
            const size_t i = (size_t)r * Q + q;
            const float4 g = dpp_ld4(G + 4 * i);
            const float4 x = dpp_ld4(X + 4 * i);
            float4 o = one(g, x, add ? dpp_ld4(add + 4 * i) : zero4);
            dpp_st4(dX + 4 * i, o);
            if (sizeof(TO) == 2) { o.x = dpp_bf16_round(o.x); o.y = dpp_bf16_round(o.y); o.z = dpp_bf16_round(o.z); o.w = dpp_bf16_round(o.w); }
            sa.x += o.x; sa.y += o.y; sa.z += o.z; sa.w += o.w;
        }
    }
    if (colsum == nullptr) return;
    s_a[tid * 4 + 0] = sa.x; s_a[tid * 4 + 1] = sa.y; s_a[tid * 4 + 2] = sa.z; s_a[tid * 4 + 3] = sa.w;
    __syncthreads();
    for (int c = tid; c < C; c += DPP_THREADS) {
        int cq = c >> 2, ce = c & 3;
        double a = 0.0;
        for (int j = 0; j < RP; ++j) a += (double)s_a[(j * Q + cq) * 4 + ce];
        colsum[(size_t)blockIdx.x * C + c] = (float)a;
    }
}

// bn_bwd_finalize + bn_bwd_apply in ONE launch, for the small maps of the late stages: when the per-block sums of a BatchNorm are few
// (nb <= 128 blocks), every workgroup of the apply pass can afford to reduce them itself -- the 2 x 32 x nb floats of ITS 32 channels,
// 8 KB..32 KB out of L2, the same order in every workgroup, so all of them form the same (c1, c2) -- instead of waiting for a
// finalize kernel whose whole duration is a launch and one memory round trip (4.8 us of a 10 us pair).  Grid (row blocks, C / 32);
// the first four rows of a thread are requested BEFORE the reduction, so the two round trips overlap.  Workgroup (0, slice) also
// writes dbeta / dgamma (the parameter gradients the finalize kernel writes).  Arithmetic of bn_bwd_finalize_kernel (f64 sums of the
// f32 partials, one rounding) and of bn_bwd_apply_kernel.
template <class TX, class TG, class TO>
__global__ __launch_bounds__(DPP_THREADS) void bn_bwd_finalize_apply_kernel(const TG* __restrict__ G, const TX* __restrict__ X, int M, int C,
                                                                            const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                                            const float* __restrict__ scale, const float* __restrict__ partial,
                                                                            int nb, const TO* add, TO* dX, int rpb, float* __restrict__ colsum,
                                                                            float* __restrict__ dbeta, float* __restrict__ dgamma) {
    dpp_kernarg_warm<128>();
    constexpr int CS = 32, Q = CS / 4, RP = DPP_THREADS / Q;
    __shared__ double s_sum[2 * CS];
    __shared__ float s_a[DPP_THREADS * 4];
    const int tid = threadIdx.x, q = tid % Q, rr = tid / Q;
    const int c0 = blockIdx.y * CS;
    const int r_begin = blockIdx.x * rpb;
    const int r_end = (r_begin + rpb < M) ? r_begin + rpb : M;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g[4], x[4], o[4];
    auto fetch = [&](int r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ru = r + u * RP;
            const size_t i = (size_t)(ru < r_end ? ru : r_end - 1) * C + c0 + q * 4;
            g[u] = dpp_ld4(G + i);
            x[u] = dpp_ld4(X + i);
            o[u] = add ? dpp_ld4(add + i) : zero4;
        }
    };
    int r = r_begin + rr;
    fetch(r);
    // ---- the sums of this slice's channels over the nb blocks: thread (pair = (s, c), quarter) ----
    {
        const int pair = tid >> 2, quarter = tid & 3, s = pair / CS, c = pair % CS;
        const float* row = partial + dpp_partial_index(s, c0 + c, 0, C, nb);
        double acc = 0.0;
        if ((nb & 3) == 0) {
            for (int b = quarter * 4; b < nb; b += 16) {
                const float4 v = *reinterpret_cast<const float4*>(row + b);
                acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            }
        } else {
            for (int b = quarter; b < nb; b += 4) acc += (double)row[b];
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (quarter == 0) s_sum[pair] = acc;
    }
    const float4 mu = *reinterpret_cast<const float4*>(mean + c0 + q * 4);
    const float4 is = *reinterpret_cast<const float4*>(inv_std + c0 + q * 4);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c0 + q * 4);
    __syncthreads();
    if (blockIdx.x == 0 && tid < CS) {
        dbeta[c0 + tid] = (float)s_sum[tid];
        dgamma[c0 + tid] = (float)s_sum[CS + tid];
    }
    float4 a1, a2;
    a1.x = (float)(s_sum[q * 4 + 0] / (double)M); a1.y = (float)(s_sum[q * 4 + 1] / (double)M);
    a1.z = (float)(s_sum[q * 4 + 2] / (double)M); a1.w = (float)(s_sum[q * 4 + 3] / (double)M);
    a2.x = (float)(s_sum[CS + q * 4 + 0] / (double)M); a2.y = (float)(s_sum[CS + q * 4 + 1] / (double)M);
    a2.z = (float)(s_sum[CS + q * 4 + 2] / (double)M); a2.w = (float)(s_sum[CS + q * 4 + 3] / (double)M);
    float4 sa = zero4;
    for (; r < r_end; r += 4 * RP) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ru = r + u * RP;
            float4 v;
            v.x = sc.x * (g[u].x - a1.x - (x[u].x - mu.x) * is.x * a2.x);
            v.y = sc.y * (g[u].y - a1.y - (x[u].y - mu.y) * is.y * a2.y);
            v.z = sc.z * (g[u].z - a1.z - (x[u].z - mu.z) * is.z * a2.z);
            v.w = sc.w * (g[u].w - a1.w - (x[u].w - mu.w) * is.w * a2.w);
            if (add) { v.x += o[u].x; v.y += o[u].y; v.z += o[u].z; v.w += o[u].w; }
            if (ru < r_end) {
                dpp_st4(dX + (size_t)ru * C + c0 + q * 4, v);
                if (sizeof(TO) == 2) { v.x = dpp_bf16_round(v.x); v.y = dpp_bf16_round(v.y); v.z = dpp_bf16_round(v.z); v.w = dpp_bf16_round(v.w); }
                sa.x += v.x; sa.y += v.y; sa.z += v.z; sa.w += v.w;
            }
        }
        if (r + 4 * RP < r_end) fetch(r + 4 * RP);
    }
    if (colsum == nullptr) return;
    s_a[tid * 4 + 0] = sa.x; s_a[tid * 4 + 1] = sa.y; s_a[tid * 4 + 2] = sa.z; s_a[tid * 4 + 3] = sa.w;
    __syncthreads();
    if (tid < CS) {
        const int cq = tid >> 2, ce = tid & 3;
        double a = 0.0;
        for (int j = 0; j < RP; ++j) a += (double)s_a[(j * Q + cq) * 4 + ce];
        colsum[(size_t)blockIdx.x * C + c0 + tid] = (float)a;
    }
}

bool ok_c(int C) { return C >= 4 && (C & 3) == 0 && (C >> 2) <= MAXQ && (DPP_THREADS % (C >> 2) == 0 || (C >> 2) > DPP_THREADS); }

}  // namespace

extern "C" int dpp_bn_stats_partial(const float* X, int M, int C, int rows_per_block, float* partial, int store, dpp_stream_t stream) {
    if (!X || !partial || M < 1 || rows_per_block < 1 || !ok_c(C) || (C >> 2) > DPP_THREADS || (store & ~DPP_ST_A)) return DPP_E_BADARG;
    int nb = dpp_cdiv(M, rows_per_block);
    if (store & DPP_ST_A)
        DPP_LAUNCH(bn_stats_partial_kernel<dpp_bf16>, dim3(nb), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
                   reinterpret_cast<const dpp_bf16*>(X), M, C, rows_per_block, partial);
    else
        DPP_LAUNCH(bn_stats_partial_kernel<float>, dim3(nb), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), X, M, C, rows_per_block, partial);
    return dpp_launch_status();
}

extern "C" int dpp_bn_finalize(const float* partial, int nb, int nseg, int M, int rows_per_block, int C, const float* gamma, float eps,
                               float* mean, float* inv_std, float* scale, float* run_mean, float* run_inv_std, float alpha,
                               dpp_stream_t stream) {
    if (!partial || !gamma || !mean || !inv_std || !scale || nseg < 1 || M % nseg || nb != dpp_cdiv(M / nseg, rows_per_block))
        return DPP_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // one wave per channel, or the whole workgroup per channel when there are many partial blocks; loads batched so that a
    // thread needs one or two round trips
#define DPP_BNF(U_, W_) DPP_LAUNCH((bn_finalize_kernel<U_, W_>), dim3(dpp_cdiv(C, 4 / W_)), dim3(DPP_THREADS), 0, st, partial, nb, nseg, \
                                           M, rows_per_block, C, gamma, eps, mean, inv_std, scale, run_mean, run_inv_std, alpha)
    if (nb <= 2 * DPP_WAVE) DPP_BNF(2, 1); else if (nb <= 8 * DPP_WAVE) DPP_BNF(8, 1); else if (nb <= 32 * DPP_WAVE) DPP_BNF(8, 4); else DPP_BNF(32, 4);
#undef DPP_BNF
    return dpp_launch_status();
}

extern "C" int dpp_bn_eval_coeffs(const float* gamma, const float* run_mean, const float* run_inv_std, int C, float* mean,
                                  float* inv_std, float* scale, dpp_stream_t stream) {
    if (!gamma || !run_mean || !run_inv_std || !mean || !inv_std || !scale || C < 1) return DPP_E_BADARG;
    DPP_LAUNCH(bn_eval_coeffs_kernel, dim3(dpp_cdiv(C, DPP_THREADS)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
                       gamma, run_mean, run_inv_std, C, mean, inv_std, scale);
    return dpp_launch_status();
}

extern "C" size_t dpp_bn_eval_job_bytes(void) { return sizeof(BnEvalJob); }

extern "C" int dpp_bn_eval_coeffs_multi(const void* jobs, int njobs, int total_blocks, dpp_stream_t stream) {
    if (!jobs || njobs < 1 || total_blocks < 1) return DPP_E_BADARG;
    DPP_LAUNCH(bn_eval_coeffs_multi_kernel, dim3(total_blocks), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
               static_cast<const BnEvalJob*>(jobs), njobs);
    return dpp_launch_status();
}

extern "C" int dpp_bn_bwd_reduce(const float* dA, const float* X, int M, int C, const float* mean, const float* inv_std,
                                 const float* scale, const float* beta, int relu, float* G, int rows_per_block, float* partial,
                                 int store, dpp_stream_t stream) {
    if (!dA || !X || !G || !partial || M < 1 || rows_per_block < 1 || !ok_c(C) || (C >> 2) > DPP_THREADS || (store & ~(DPP_ST_BNX | DPP_ST_A | DPP_ST_C)))
        return DPP_E_BADARG;
    const bool x16 = (store & DPP_ST_BNX) != 0, g16 = (store & DPP_ST_A) != 0;
    if (g16 != ((store & DPP_ST_C) != 0)) return DPP_E_UNSUPPORTED;            // dA and G are stored alike (G usually overwrites dA)
    int nb = dpp_cdiv(M, rows_per_block);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_BR(TX_, TG_) DPP_LAUNCH((bn_bwd_reduce_kernel<TX_, TG_>), dim3(nb), dim3(DPP_THREADS), 0, st, reinterpret_cast<const TG_*>(dA), \
                                    reinterpret_cast<const TX_*>(X), M, C, mean, inv_std, scale, beta, relu, reinterpret_cast<TG_*>(G), rows_per_block, partial)
    if (x16) { if (g16) DPP_BR(dpp_bf16, dpp_bf16); else DPP_BR(dpp_bf16, float); }
    else { if (g16) DPP_BR(float, dpp_bf16); else DPP_BR(float, float); }
#undef DPP_BR
    return dpp_launch_status();
}

extern "C" int dpp_bn_bwd_finalize(const float* partial, int nb, int nseg, int M, int C, float* dbeta, float* dgamma, float* c1,
                                   float* c2, const float* inv_std, const float* scale, float* q, float* p, dpp_stream_t stream) {
    if (!partial || !dbeta || !dgamma || !c1 || !c2 || nb < 1 || nseg < 1) return DPP_E_BADARG;
    if (q && !(p && inv_std && scale)) return DPP_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_BNBF(U_, W_) DPP_LAUNCH((bn_bwd_finalize_kernel<U_, W_>), dim3(dpp_cdiv(C, 4 / W_)), dim3(DPP_THREADS), 0, st, partial, nb, \
                                            nseg, M, C, dbeta, dgamma, c1, c2, inv_std, scale, q, p)
    if (nb <= 2 * DPP_WAVE) DPP_BNBF(2, 1); else if (nb <= 8 * DPP_WAVE) DPP_BNBF(8, 1); else if (nb <= 32 * DPP_WAVE) DPP_BNBF(8, 4); else DPP_BNBF(32, 4);
#undef DPP_BNBF
    return dpp_launch_status();
}

extern "C" int dpp_bn_bwd_apply(const float* G, const float* X, int M, int C, const float* mean, const float* inv_std,
                                const float* scale, const float* c1, const float* c2, const float* add, float* dX,
                                int rows_per_block, float* colsum_partial, int store, dpp_stream_t stream) {
    if (!G || !X || !dX || M < 1 || rows_per_block < 1 || !ok_c(C) || (C >> 2) > DPP_THREADS || (store & ~(DPP_ST_BNX | DPP_ST_A | DPP_ST_C)))
        return DPP_E_BADARG;
    const bool x16 = (store & DPP_ST_BNX) != 0, g16 = (store & DPP_ST_A) != 0, o16 = (store & DPP_ST_C) != 0;
    const dim3 grid(dpp_cdiv(M, rows_per_block));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_BA(TX_, TG_, TO_) DPP_LAUNCH((bn_bwd_apply_kernel<TX_, TG_, TO_>), grid, dim3(DPP_THREADS), 0, st, reinterpret_cast<const TG_*>(G), \
                                         reinterpret_cast<const TX_*>(X), M, C, mean, inv_std, scale, c1, c2, reinterpret_cast<const TO_*>(add), \
                                         reinterpret_cast<TO_*>(dX), rows_per_block, colsum_partial)
#define DPP_BA_O(TX_, TG_) do { if (o16) DPP_BA(TX_, TG_, dpp_bf16); else DPP_BA(TX_, TG_, float); } while (0)
    if (x16) { if (g16) DPP_BA_O(dpp_bf16, dpp_bf16); else DPP_BA_O(dpp_bf16, float); }
    else { if (g16) DPP_BA_O(float, dpp_bf16); else DPP_BA_O(float, float); }
#undef DPP_BA_O
#undef DPP_BA
    return dpp_launch_status();
}

// Whether dpp_bn_bwd_finalize_apply takes the shape: 32-channel slices, and few enough blocks of sums that every workgroup reduces
// its slice's share itself (2 x 32 x nb floats).
extern "C" int dpp_bn_bwd_finalize_apply_ok(int M, int C, int nb) { return M >= 1 && C >= 32 && (C & 31) == 0 && nb >= 1 && nb <= 256; }

extern "C" int dpp_bn_bwd_finalize_apply(const float* G, const float* X, int M, int C, const float* mean, const float* inv_std,
                                         const float* scale, const float* partial, int nb, const float* add, float* dX,
                                         int rows_per_block, float* colsum_partial, float* dbeta, float* dgamma, int store,
                                         dpp_stream_t stream) {
    if (!G || !X || !dX || !partial || !dbeta || !dgamma || rows_per_block < 1 || (store & ~(DPP_ST_BNX | DPP_ST_A | DPP_ST_C))) return DPP_E_BADARG;
    if (!dpp_bn_bwd_finalize_apply_ok(M, C, nb)) return DPP_E_UNSUPPORTED;
    const bool x16 = (store & DPP_ST_BNX) != 0, g16 = (store & DPP_ST_A) != 0, o16 = (store & DPP_ST_C) != 0;
    const dim3 grid(dpp_cdiv(M, rows_per_block), C / 32);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_BFA(TX_, TG_, TO_) DPP_LAUNCH((bn_bwd_finalize_apply_kernel<TX_, TG_, TO_>), grid, dim3(DPP_THREADS), 0, st, reinterpret_cast<const TG_*>(G), \
                                          reinterpret_cast<const TX_*>(X), M, C, mean, inv_std, scale, partial, nb, reinterpret_cast<const TO_*>(add), \
                                          reinterpret_cast<TO_*>(dX), rows_per_block, colsum_partial, dbeta, dgamma)
#define DPP_BFA_O(TX_, TG_) do { if (o16) DPP_BFA(TX_, TG_, dpp_bf16); else DPP_BFA(TX_, TG_, float); } while (0)
    if (x16) { if (g16) DPP_BFA_O(dpp_bf16, dpp_bf16); else DPP_BFA_O(dpp_bf16, float); }
    else { if (g16) DPP_BFA_O(float, dpp_bf16); else DPP_BFA_O(float, float); }
#undef DPP_BFA_O
#undef DPP_BFA
    return dpp_launch_status();
}
