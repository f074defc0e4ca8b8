// wgrad.hip -- filter gradient of the 1x1 convolutions as a barrier-free row stream (gfx950, v_mfma_f32_16x16x4_f32).
//
//   dW[o][c] = sum_m dY[m][o] * act(X)[map(m)][c]            T.grad of /root/reference/src/net/convlayer.py:230-240
//                                                             (/root/reference/src/trainer/poseregnettrainer.py:110-111)
// M = pixels (8 192 .. 131 072 at batch 128), Co x Ci = 16 x 64 .. 256 x 64: a reduction over a long, thin pair of tensors into a
// small matrix -- an HBM stream with a tiny product inside.  The LDS-tiled GEMM of gemm.hip ran these as <1, 1, 256> grids, one
// workgroup per CU, one 64-row chunk in flight each: 0.93 TB/s on 40 MB (profiles/r02_kernel_stats_single_stream_by_grid.txt), and
// the 44 launches were most of the 1.9 ms gradient branch.  Here:
//   * no LDS, no barrier: a WAVE owns a range of pixel rows and a block of the output; its operands go from global memory straight
//     into MFMA fragments.  The MFMA wants lane (i, kq) to hold A[i][k = kq]: the lane loads VA CONTIGUOUS floats of pixel row
//     m0 + kq at channel VA * i (16 lanes x 16 bytes = one 256-byte row for 64 channels) and component e of that vector is the A
//     operand of the MFMAs that produce output rows o = VA * i + e -- the channel permutation lives in the epilogue's addresses,
//     nothing is transposed.  The same on the X side (VB contiguous floats, columns c = VB * j + f), with the BatchNorm + ReLU
//     prologue applied in registers (the lane's VB channels never change: coefficients are loaded once).
//   * a ring of U stages (4 pixel rows each) keeps U steps of loads in flight: a stage is refilled the moment it is consumed;
//   * wide outputs are split over the waves of a workgroup by channel blocks (64 x 256: four waves x 64 columns), narrow ones by
//     rows; every (workgroup, row split) writes its own partial slice [Co][Ci], summed in fixed order by dpp_reduce_multi.
#include "dpp_common.h"

namespace {

struct WgradArgs {
    const float* dY; const float* X; float* partial;
    dpp_rowmap mapX; dpp_act actX;
    int M, Co, Ci, rpw;                  // rpw: pixel rows per wave
};

// The X operand (the forward activations) may be a bf16-stored tensor (DPP_ST_B): V consecutive bf16 elements as ONE 2 / 4 / 8 / 16
// byte load, widened exactly.
template <int V>
__device__ __forceinline__ void load_vec(const dpp_bf16* p, float (&v)[V]) {
    if (V == 1) v[0] = (float)p[0];
    else if (V == 2) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const b2 t = *reinterpret_cast<const b2*>(p);
        v[0] = (float)t[0]; v[1] = (float)t[1];
    } else if (V == 4) {
        const dpp_bf16x4 t = *reinterpret_cast<const dpp_bf16x4*>(p);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (float)t[q];
    } else {
#pragma unroll
        for (int h = 0; h < V / 8; ++h) {
            const dpp_bf16x8 t = *reinterpret_cast<const dpp_bf16x8*>(p + h * 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[h * 8 + q] = (float)t[q];
        }
    }
}

template <int V>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[V]) {
    if (V == 1) v[0] = p[0];
    else if (V == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    else {
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(p + q * 4);
            v[q * 4 + 0] = t.x; v[q * 4 + 1] = t.y; v[q * 4 + 2] = t.z; v[q * 4 + 3] = t.w;
        }
    }
}

// TA = Co / 16, TB = Ci / 16 (vector lengths of a full row per lane); the 4 waves split A channels WA ways, B channels WB ways and
// rows WR = 4 / (WA * WB) ways.  VA = TA / WA, VB = TB / WB floats per lane and operand; VA * VB accumulator tiles per wave.
// PB (round 6, dpp_wgrad_stream_bf16): both operands rounded to bfloat16 (the activation after its prologue), the eight stages of a ring
// round -- lane (i, kq) holds pixel rows m0 + 4 u + kq, u = 0 .. 7 -- packed into ONE v_mfma_f32_16x16x32_bf16 operand per output tile
// (the same pixel for element u on the dY and on the X side, which is all the reduction needs); the ring, the loads and the epilogue are
// the float32 kernel's.
template <int TA, int TB, int WA, int WB, int U, bool STRIDED, class TX = float, class TY = float, bool PB = false>
__global__ __launch_bounds__(DPP_THREADS) void wgrad_stream_kernel(WgradArgs a) {
    static_assert(!PB || U == 8, "bf16 operands: eight stages per MFMA");
    constexpr int WR = 4 / (WA * WB), VA = TA / WA, VB = TB / WB;
    static_assert(WA * WB * WR == 4 && VA >= 1 && VB >= 1 && VA <= 4 && VB <= 8, "wave split");
    dpp_kernarg_warm<sizeof(WgradArgs)>();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wa = wave % WA, wb = (wave / WA) % WB, wr = wave / (WA * WB);
    const int slice = blockIdx.x * WR + wr;
    const int row_begin = slice * a.rpw, row_end = (row_begin + a.rpw < a.M) ? row_begin + a.rpw : a.M;
    const int ca = wa * 16 * VA + VA * l15;              // first dY channel of this lane
    const int cb = wb * 16 * VB + VB * l15;              // first X channel of this lane
    float mu[VB], sc[VB], be[VB];
    const int mode = a.actX.mode;
#pragma unroll
    for (int f = 0; f < VB; ++f) { mu[f] = 0.f; sc[f] = 1.f; be[f] = 0.f; }
    if (mode & 2) { load_vec<VB>(a.actX.mean + cb, mu); load_vec<VB>(a.actX.scale + cb, sc); load_vec<VB>(a.actX.beta + cb, be); }
    f32x4 acc[VA][VB];
#pragma unroll
    for (int e = 0; e < VA; ++e)
#pragma unroll
        for (int f = 0; f < VB; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A ring of U stages, one step (4 pixel rows) each: stage s is consumed and at once refilled with the step U ahead, so U steps
    // of loads are in flight at any time (vmcnt retires loads in order: consuming a stage waits for ITS loads only).  With one wave
    // per SIMD on the wide layers nothing else hides the memory round trip: loads consumed in the iteration that issued them made
    // the stage-3 / 4 launches 30 us for 3.4 us of MFMA work.  The loop body is BRANCH-FREE (the refill past the end re-reads the
    // last row, whose dY operand is zeroed when the stage is consumed; the row map is a template flag): any branch between issue and
    // use makes the compiler wait for all outstanding loads (s_waitcnt vmcnt(0)) at the merge point.
    float av[U][VA], bv[U][VB];
    bool okv[U];
    dpp_bf16x8 pa[PB ? VA : 1], pb[PB ? VB : 1];
    const int last = row_end - 1;
    auto fetch = [&](int u, int m0) {
        const int m = m0 + kq;
        okv[u] = m < row_end;
        const int mm = m < row_end ? m : last;
        load_vec<VA>(reinterpret_cast<const TY*>(a.dY) + (size_t)mm * a.Co + ca, av[u]);
        const int xr = STRIDED ? dpp_map_row(a.mapX, mm) : mm;
        load_vec<VB>(reinterpret_cast<const TX*>(a.X) + (size_t)xr * a.Ci + cb, bv[u]);
    };
    if (row_begin < row_end) {
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(u, row_begin + 4 * u);
        DPP_SCHED_FENCE();
        for (int m0 = row_begin; m0 < row_end; m0 += 4 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float bq[VB], aq[VA];
#pragma unroll
                for (int f = 0; f < VB; ++f) {
                    float v = bv[u][f];
                    if (mode & 2) v = dpp_fma(v - mu[f], sc[f], be[f]);
                    if (mode & 1) v = fmaxf(v, 0.0f);
                    bq[f] = v;
                }
#pragma unroll
                for (int e = 0; e < VA; ++e) aq[e] = okv[u] ? av[u][e] : 0.0f;
                DPP_SCHED_FENCE();
                fetch(u, m0 + 4 * (U + u));                               // refill: the step U ahead
                DPP_SCHED_FENCE();
                if constexpr (PB) {
#pragma unroll
                    for (int e = 0; e < VA; ++e) pa[e][u] = (dpp_bf16)aq[e];
#pragma unroll
                    for (int f = 0; f < VB; ++f) pb[f][u] = (dpp_bf16)bq[f];
                    if (u == U - 1) {
#pragma unroll
                        for (int e = 0; e < VA; ++e)
#pragma unroll
                            for (int f = 0; f < VB; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[e], pb[f], acc[e][f], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VA; ++e)
#pragma unroll
                        for (int f = 0; f < VB; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[e], bq[f], acc[e][f], 0, 0, 0);
                }
            }
        }
    }
    // D layout: lane (j = l15, kq) holds rows i = 4 kq + r of tile (e, f): output element (o = wa*16*VA + VA*i + e, c = cb + f)
    float* out = a.partial + (size_t)slice * a.Co * a.Ci;
#pragma unroll
    for (int e = 0; e < VA; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = wa * 16 * VA + VA * (kq * 4 + r) + e;
            float* po = out + (size_t)o * a.Ci + cb;
            if (VB == 1) po[0] = acc[e][0][r];
            else if (VB == 2) *reinterpret_cast<float2*>(po) = make_float2(acc[e][0][r], acc[e][1][r]);
            else {
#pragma unroll
                for (int q = 0; q < VB / 4; ++q)
                    *reinterpret_cast<float4*>(po + q * 4) = make_float4(acc[e][q * 4][r], acc[e][q * 4 + 1][r], acc[e][q * 4 + 2][r], acc[e][q * 4 + 3][r]);
            }
        }
}

struct Shape { int Co, Ci, WR; };

// (Co, Ci) -> rows split WR of the instantiation that handles it, or 0
int shape_wr(int Co, int Ci) {
    static const Shape table[] = {{16, 64, 4}, {64, 16, 4}, {16, 32, 4}, {64, 32, 4}, {32, 64, 4}, {32, 128, 2}, {128, 32, 2},
                                  {128, 64, 2}, {64, 128, 2}, {64, 256, 1}, {256, 64, 1}, {256, 128, 1}};
    for (const Shape& s : table)
        if (s.Co == Co && s.Ci == Ci) return s.WR;
    return 0;
}

// ---- 3x3 filter gradient as the same kind of stream ---------------------------------------------------------------------------------
//   dW[o][t][c] = sum_p dY[p][o] * act(X)[p + (dy, dx)][c],   t = 3 (dy + 1) + (dx + 1), zero outside the image
// The LDS-tiled conv3x3_wgrad_kernel stages a halo tile and the dY tile in LDS and spends 17 VALU instructions per MFMA on LDS
// addressing and the prologue (profiles/r02_instruction_mix.txt): 18 TF/s, 0.67 ms of the gradient branch.  A 3x3 tap is only a
// shifted row of the same tensor, so the row stream above takes it as it is: a step is 4 CONSECUTIVE pixels of one image row (W % 4
// == 0), lane (l15, kq) loads its VA channels of dY at pixel p0 + kq and, per tap, its VB channels of X at the shifted pixel (the
// centre pixel's address and a zero operand where the tap leaves the image); TG taps x VA x VB accumulator tiles per wave.  The nine
// shifted reads of a pixel row hit the same cache lines (neighbouring taps, neighbouring steps): HBM sees each tensor about once
// per tap group.
struct Wgrad3sArgs {
    const float* dY; const float* X; float* partial;
    dpp_act actX;
    int M, H, W, Co, Ci, rpw;
};

template <int V, class T = float>
__device__ __forceinline__ void load_vec_at(const char* base, unsigned byte_off, float (&v)[V]) {     // uniform base + 32-bit lane offset
    load_vec<V>(reinterpret_cast<const T*>(base + byte_off), v);
}

template <int TA, int TB, int WA, int WB, int TG, int U, class TX = float, class TY = float>
__global__ __launch_bounds__(DPP_THREADS) void wgrad3_stream_kernel(Wgrad3sArgs a) {
    constexpr int WR = 4 / (WA * WB), VA = TA / WA, VB = TB / WB;
    static_assert(WA * WB * WR == 4 && VA >= 1 && VB >= 1 && VA <= 4 && VB <= 4 && (TG == 3 || TG == 9), "wave split");
    dpp_kernarg_warm<sizeof(Wgrad3sArgs)>();
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);              // row ranges and cursors below stay in scalar registers
    const int wa = wave % WA, wb = (wave / WA) % WB, wr = wave / (WA * WB);
    const int slice = blockIdx.x * WR + wr;
    const int tap0 = blockIdx.y * TG;                    // TG == 3: one filter row (dy = blockIdx.y - 1) per blockIdx.y
    const int row_begin = slice * a.rpw, row_end = (row_begin + a.rpw < a.M) ? row_begin + a.rpw : a.M;
    const int H = a.H, W = a.W, Co = a.Co, Ci = a.Ci;
    const int ca = wa * 16 * VA + VA * l15, cb = wb * 16 * VB + VB * l15;
    // the prologue without selects: (v - 0) * 1 + 0 and max(v, -inf) are exact identities
    float mu[VB], sc[VB], be[VB];
    const int mode = a.actX.mode;
#pragma unroll
    for (int f = 0; f < VB; ++f) { mu[f] = 0.f; sc[f] = 1.f; be[f] = 0.f; }
    if (mode & 2) { load_vec<VB>(a.actX.mean + cb, mu); load_vec<VB>(a.actX.scale + cb, sc); load_vec<VB>(a.actX.beta + cb, be); }
    const float lo = (mode & 1) ? 0.0f : -__builtin_inff();
    f32x4 acc[TG][VA][VB];
#pragma unroll
    for (int j = 0; j < TG; ++j)
#pragma unroll
        for (int e = 0; e < VA; ++e)
#pragma unroll
            for (int f = 0; f < VB; ++f) acc[j][e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av[U][VA], bv[U][TG][VB];
    unsigned okv[U];                                     // bit j: tap j of the stage is inside the image; bit 16: the pixel row exists
    const int last = row_end - 1;
    const char* baseA = reinterpret_cast<const char*>(a.dY);
    const char* baseB = reinterpret_cast<const char*>(a.X);
    const unsigned strideA = (unsigned)Co * (unsigned)sizeof(TY), strideB = (unsigned)Ci * (unsigned)sizeof(TX), offA = (unsigned)ca * (unsigned)sizeof(TY),
                   offB = (unsigned)cb * (unsigned)sizeof(TX);
    int fp = row_begin, fx = row_begin % W, fy = (row_begin / W) % H;         // the fetch cursor: pixel, its column and image row
    auto fetch = [&](int u) {
        const int p = fp + kq, x = fx + kq;
        const bool ok = p < row_end;
        const int pp = ok ? p : last;
        unsigned mask = ok ? 0x10000u : 0u;
        load_vec_at<VA, TY>(baseA, (unsigned)pp * strideA + offA, av[u]);
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            const int dy = (TG == 9 ? j / 3 : (int)blockIdx.y) - 1, dx = (TG == 9 ? j % 3 : j) - 1;
            const bool v = ok & ((unsigned)(fy + dy) < (unsigned)H) & ((unsigned)(x + dx) < (unsigned)W);    // no short-circuit: no branches
            const int q = v ? pp + dy * W + dx : pp;
            mask |= (v ? 1u : 0u) << j;
            load_vec_at<VB, TX>(baseB, (unsigned)q * strideB + offB, bv[u][j]);
        }
        okv[u] = mask;
        fp += 4; fx += 4;
        const bool wrap = fx >= W;
        fx = wrap ? 0 : fx;
        fy += wrap ? 1 : 0;
        fy = fy >= H ? 0 : fy;
    };
    if (row_begin < row_end) {
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(u);
        DPP_SCHED_FENCE();
        for (int m0 = row_begin; m0 < row_end; m0 += 4 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float bq[TG][VB], aq[VA];
                const unsigned mask = okv[u];
#pragma unroll
                for (int j = 0; j < TG; ++j) {
                    const bool in = (mask >> j) & 1u;
#pragma unroll
                    for (int f = 0; f < VB; ++f) bq[j][f] = in ? fmaxf(dpp_fma(bv[u][j][f] - mu[f], sc[f], be[f]), lo) : 0.0f;
                }
#pragma unroll
                for (int e = 0; e < VA; ++e) aq[e] = (mask & 0x10000u) ? av[u][e] : 0.0f;
                DPP_SCHED_FENCE();
                fetch(u);                                                  // refill: the step U ahead
                DPP_SCHED_FENCE();
#pragma unroll
                for (int j = 0; j < TG; ++j)
#pragma unroll
                    for (int e = 0; e < VA; ++e)
#pragma unroll
                        for (int f = 0; f < VB; ++f) acc[j][e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[e], bq[j][f], acc[j][e][f], 0, 0, 0);
            }
        }
    }
    float* out = a.partial + (size_t)slice * Co * 9 * Ci;
#pragma unroll
    for (int e = 0; e < VA; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = wa * 16 * VA + VA * (kq * 4 + r) + e;
#pragma unroll
            for (int j = 0; j < TG; ++j) {
                float* po = out + ((size_t)o * 9 + tap0 + j) * Ci + cb;
                if (VB == 1) po[0] = acc[j][e][0][r];
                else if (VB == 2) *reinterpret_cast<float2*>(po) = make_float2(acc[j][e][0][r], acc[j][e][1][r]);
                else *reinterpret_cast<float4*>(po) = make_float4(acc[j][e][0][r], acc[j][e][1][r], acc[j][e][2][r], acc[j][e][3][r]);
            }
        }
}

// ---- the filter gradient of a HiddenLayer with a SHORT reduction (FC1: dW[k][n] = sum_b act(X)[b][k] * dY[b][n], b < 128) ---------------
// 16 384 x 1 024 outputs, 67 MB, from a reduction over only the batch: the LDS-tiled kernels (fc_stream_kernel: 110 KB of LDS, one
// workgroup per CU, a three-stage pipeline for FOUR chunks) are all prologue and epilogue here -- 84 us for 4.3 GFLOP.  The operands are
// contiguous along the OUTPUT dimensions (X[b][k], dY[b][n]), which is exactly the fragment order of the row stream above: lane
// (l15, kq) loads 4 contiguous floats of row b0 + kq from each operand, a wave owns a 64 x 64 block of dW over ALL rows (16 MFMAs per
// pair of 16-byte loads, accumulators in registers, no LDS, no barrier, no partials: each output is written once by its owner).
// A = X (its BatchNorm + ReLU prologue applied in registers: channel = column % cmod) gives the output rows k, B = dY the columns n.
struct FcWgradArgs {
    const float* X; const float* dY; float* dW;
    dpp_act actX;
    int Nb, K, N;                        // rows (samples), columns of X (= rows of dW), columns of dY
};

template <int U, class TX = float>
__global__ __launch_bounds__(DPP_THREADS) void fc_wgrad_stream_kernel(FcWgradArgs a) {
    constexpr int V = 4;                 // 4 x 4 tiles of 16 x 16 per wave
    dpp_kernarg_warm<sizeof(FcWgradArgs)>();
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k0 = blockIdx.x * 128 + (wave >> 1) * 64, n0 = blockIdx.y * 128 + (wave & 1) * 64;      // this wave's 64 x 64 block
    const int ka = k0 + V * l15, nb = n0 + V * l15;              // first X column / dY column of this lane
    float mu[V], sc[V], be[V];
    const int mode = a.actX.mode;
#pragma unroll
    for (int f = 0; f < V; ++f) { mu[f] = 0.f; sc[f] = 1.f; be[f] = 0.f; }
    if (mode & 2) {
        const int c = ka % a.actX.cmod;                         // cmod % 4 == 0: the four columns share a quad of channels
        load_vec<V>(a.actX.mean + c, mu); load_vec<V>(a.actX.scale + c, sc); load_vec<V>(a.actX.beta + c, be);
    }
    const float lo = (mode & 1) ? 0.0f : -__builtin_inff();
    f32x4 acc[V][V];
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int f = 0; f < V; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av[U][V], bv[U][V];
    const char* baseA = reinterpret_cast<const char*>(a.X);
    const char* baseB = reinterpret_cast<const char*>(a.dY);
    const unsigned strideA = (unsigned)a.K * (unsigned)sizeof(TX), strideB = (unsigned)a.N * 4u, offA = (unsigned)ka * (unsigned)sizeof(TX), offB = (unsigned)nb * 4u;
    const int last = a.Nb - 1;
    int fb = 0;                                                  // the fetch cursor (row of lane group kq = 0)
    auto fetch = [&](int u) {
        const int b = fb + kq;
        const int bb = b < a.Nb ? b : last;                      // rows past the end re-read the last row; their A operand is zeroed
        load_vec_at<V, TX>(baseA, (unsigned)bb * strideA + offA, av[u]);
        load_vec_at<V>(baseB, (unsigned)bb * strideB + offB, bv[u]);
        fb += 4;
    };
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(u);
    DPP_SCHED_FENCE();
    for (int b0 = 0; b0 < a.Nb; b0 += 4 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float aq[V], bq[V];
            const bool ok = b0 + 4 * u + kq < a.Nb;
#pragma unroll
            for (int e = 0; e < V; ++e) aq[e] = ok ? fmaxf(dpp_fma(av[u][e] - mu[e], sc[e], be[e]), lo) : 0.0f;
#pragma unroll
            for (int f = 0; f < V; ++f) bq[f] = bv[u][f];
            DPP_SCHED_FENCE();
            fetch(u);                                            // refill: the step U ahead
            DPP_SCHED_FENCE();
#pragma unroll
            for (int e = 0; e < V; ++e)
#pragma unroll
                for (int f = 0; f < V; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[e], bq[f], acc[e][f], 0, 0, 0);
        }
    }
    // D layout: lane (j = l15, kq) holds rows i = 4 kq + r of tile (e, f): dW[k0 + V i + e][n0 + V l15 + f]
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + V * (kq * 4 + r) + e;
            *reinterpret_cast<float4*>(a.dW + (size_t)k * a.N + nb) = make_float4(acc[e][0][r], acc[e][1][r], acc[e][2][r], acc[e][3][r]);
        }
}

// channels C = Co = Ci -> (rows split WR, taps per blockIdx.y TG) of the instantiation, or WR = 0
void shape3(int Co, int Ci, int& WR, int& TG) {
    WR = 0; TG = 9;
    if (Co != Ci) return;
    if (Co == 16) { WR = 4; TG = 9; }
    else if (Co == 32) { WR = 4; TG = 3; }
    else if (Co == 64) { WR = 1; TG = 3; }
}

}  // namespace

extern "C" int dpp_wgrad_stream_slices(int Co, int Ci, int M, int rows_per_wave) {
    const int WR = shape_wr(Co, Ci);
    if (!WR || M < 1 || rows_per_wave < 4 || (rows_per_wave & 3)) return 0;
    return dpp_cdiv(M, rows_per_wave * WR) * WR;
}

static int wgrad_stream_launch(const float* dY, int Co, const float* X, int Ci, const dpp_rowmap* mapX, const dpp_act* actX, int M,
                               int rows_per_wave, float* partial, int store, int precision, dpp_stream_t stream);

extern "C" int dpp_wgrad_stream(const float* dY, int Co, const float* X, int Ci, const dpp_rowmap* mapX, const dpp_act* actX, int M,
                                int rows_per_wave, float* partial, int store, dpp_stream_t stream) {
    return wgrad_stream_launch(dY, Co, X, Ci, mapX, actX, M, rows_per_wave, partial, store, 0, stream);
}

// bf16 MFMA operands: the stage-1 shapes (16 / 64 output x 16 / 32 / 64 input channels), the layers the engine puts on this kernel
extern "C" int dpp_wgrad_stream_bf16_ok(int Co, int Ci) {
    return ((Co == 16 && (Ci == 64 || Ci == 32)) || (Co == 64 && (Ci == 16 || Ci == 32))) ? 1 : 0;
}

extern "C" int dpp_wgrad_stream_bf16(const float* dY, int Co, const float* X, int Ci, const dpp_rowmap* mapX, const dpp_act* actX, int M,
                                     int rows_per_wave, float* partial, int store, dpp_stream_t stream) {
    if (!dpp_wgrad_stream_bf16_ok(Co, Ci)) return DPP_E_UNSUPPORTED;
    return wgrad_stream_launch(dY, Co, X, Ci, mapX, actX, M, rows_per_wave, partial, store, 1, stream);
}

static int wgrad_stream_launch(const float* dY, int Co, const float* X, int Ci, const dpp_rowmap* mapX, const dpp_act* actX, int M,
                               int rows_per_wave, float* partial, int store, int precision, dpp_stream_t stream) {
    if (!dY || !X || !partial || (store & ~(DPP_ST_A | DPP_ST_B))) return DPP_E_BADARG;
    const bool x16 = (store & DPP_ST_B) != 0, y16 = (store & DPP_ST_A) != 0;
    const int nsl = dpp_wgrad_stream_slices(Co, Ci, M, rows_per_wave);
    if (!nsl) return DPP_E_UNSUPPORTED;
    WgradArgs a;
    a.dY = dY; a.X = X; a.partial = partial; a.M = M; a.Co = Co; a.Ci = Ci; a.rpw = rows_per_wave;
    a.mapX.s = 1; a.mapX.Wo = a.mapX.HoWo = a.mapX.Wi = a.mapX.HiWi = 0;
    if (mapX) a.mapX = *mapX;
    a.actX.mean = a.actX.scale = a.actX.beta = nullptr; a.actX.mode = 0; a.actX.cmod = Ci; a.actX.x2 = a.actX.aux = nullptr; a.actX.out = nullptr;
    if (actX) a.actX = *actX;
    if (a.actX.mode & ~3) return DPP_E_UNSUPPORTED;
    if ((a.actX.mode & 2) && !(a.actX.mean && a.actX.scale && a.actX.beta && a.actX.cmod == Ci)) return DPP_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(partial)) & 15) return DPP_E_BADARG;
    const int WR = shape_wr(Co, Ci);
    const dim3 grid(nsl / WR), block(DPP_THREADS);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_WGK(CO_, CI_, WA_, WB_, U_, S_, TX_, TY_) DPP_LAUNCH((wgrad_stream_kernel<CO_ / 16, CI_ / 16, WA_, WB_, U_, S_, TX_, TY_>), grid, block, 0, st, a)
#define DPP_WGKB(CO_, CI_, S_, TX_, TY_) DPP_LAUNCH((wgrad_stream_kernel<CO_ / 16, CI_ / 16, 1, 1, 8, S_, TX_, TY_, true>), grid, block, 0, st, a)
#define DPP_WGSB(CO_, CI_, TX_, TY_) do { if (a.mapX.s != 1) DPP_WGKB(CO_, CI_, true, TX_, TY_); else DPP_WGKB(CO_, CI_, false, TX_, TY_); } while (0)
#define DPP_WGB(CO_, CI_) if (precision && Co == CO_ && Ci == CI_) { \
        if (x16) { if (y16) DPP_WGSB(CO_, CI_, dpp_bf16, dpp_bf16); else DPP_WGSB(CO_, CI_, dpp_bf16, float); } \
        else { if (y16) DPP_WGSB(CO_, CI_, float, dpp_bf16); else DPP_WGSB(CO_, CI_, float, float); } \
        return dpp_launch_status(); }
    DPP_WGB(16, 64) DPP_WGB(64, 16) DPP_WGB(16, 32) DPP_WGB(64, 32)
    if (precision) return DPP_E_UNSUPPORTED;
#define DPP_WGS(CO_, CI_, WA_, WB_, U_, TX_, TY_) do { if (a.mapX.s != 1) DPP_WGK(CO_, CI_, WA_, WB_, U_, true, TX_, TY_); \
                                                       else DPP_WGK(CO_, CI_, WA_, WB_, U_, false, TX_, TY_); } while (0)
#define DPP_WG(CO_, CI_, WA_, WB_, U_) if (Co == CO_ && Ci == CI_) { \
        if (x16) { if (y16) DPP_WGS(CO_, CI_, WA_, WB_, U_, dpp_bf16, dpp_bf16); else DPP_WGS(CO_, CI_, WA_, WB_, U_, dpp_bf16, float); } \
        else { if (y16) DPP_WGS(CO_, CI_, WA_, WB_, U_, float, dpp_bf16); else DPP_WGS(CO_, CI_, WA_, WB_, U_, float, float); } \
        return dpp_launch_status(); }
    DPP_WG(16, 64, 1, 1, 8)
    DPP_WG(64, 16, 1, 1, 8)
    DPP_WG(16, 32, 1, 1, 8)
    DPP_WG(64, 32, 1, 1, 8)
    DPP_WG(32, 64, 1, 1, 8)
    DPP_WG(32, 128, 1, 2, 8)
    DPP_WG(128, 32, 2, 1, 8)
    DPP_WG(128, 64, 2, 1, 8)
    DPP_WG(64, 128, 1, 2, 8)
    DPP_WG(64, 256, 1, 4, 8)
    DPP_WG(256, 64, 4, 1, 8)
    DPP_WG(256, 128, 4, 1, 4)
#undef DPP_WG
#undef DPP_WGS
#undef DPP_WGK
#undef DPP_WGB
#undef DPP_WGSB
#undef DPP_WGKB
    return DPP_E_UNSUPPORTED;
}

extern "C" int dpp_wgrad3_stream_slices(int Co, int Ci, int N, int H, int W, int rows_per_wave) {
    int WR, TG;
    shape3(Co, Ci, WR, TG);
    if (!WR || N < 1 || H < 1 || W < 4 || (W & 3) || rows_per_wave < 4 || (rows_per_wave & 3)) return 0;
    if ((long long)N * H * W * (Co > Ci ? Co : Ci) * 4 > 0x7fffffffLL) return 0;       // 32-bit byte offsets
    return dpp_cdiv(N * H * W, rows_per_wave * WR) * WR;
}

extern "C" int dpp_wgrad3_stream(const float* dY, int Co, const float* X, int Ci, int N, int H, int W, const dpp_act* actX,
                                 int rows_per_wave, float* partial, int store, dpp_stream_t stream) {
    if (!dY || !X || !partial || (store & ~(DPP_ST_A | DPP_ST_B))) return DPP_E_BADARG;
    const bool x16 = (store & DPP_ST_B) != 0, y16 = (store & DPP_ST_A) != 0;
    const int nsl = dpp_wgrad3_stream_slices(Co, Ci, N, H, W, rows_per_wave);
    if (!nsl) return DPP_E_UNSUPPORTED;
    Wgrad3sArgs a;
    a.dY = dY; a.X = X; a.partial = partial; a.M = N * H * W; a.H = H; a.W = W; a.Co = Co; a.Ci = Ci; a.rpw = rows_per_wave;
    a.actX.mean = a.actX.scale = a.actX.beta = nullptr; a.actX.mode = 0; a.actX.cmod = Ci; a.actX.x2 = a.actX.aux = nullptr; a.actX.out = nullptr;
    if (actX) a.actX = *actX;
    if (a.actX.mode & ~3) return DPP_E_UNSUPPORTED;
    if ((a.actX.mode & 2) && !(a.actX.mean && a.actX.scale && a.actX.beta && a.actX.cmod == Ci)) return DPP_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(partial)) & 15) return DPP_E_BADARG;
    int WR, TG;
    shape3(Co, Ci, WR, TG);
    const dim3 grid(nsl / WR, 9 / TG), block(DPP_THREADS);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_W3S(TX_, TY_) do { \
        if (Co == 16) DPP_LAUNCH((wgrad3_stream_kernel<1, 1, 1, 1, 9, 4, TX_, TY_>), grid, block, 0, st, a); \
        else if (Co == 32) DPP_LAUNCH((wgrad3_stream_kernel<2, 2, 1, 1, 3, 8, TX_, TY_>), grid, block, 0, st, a); \
        else DPP_LAUNCH((wgrad3_stream_kernel<4, 4, 2, 2, 3, 8, TX_, TY_>), grid, block, 0, st, a); } while (0)
    if (x16) { if (y16) DPP_W3S(dpp_bf16, dpp_bf16); else DPP_W3S(dpp_bf16, float); }
    else { if (y16) DPP_W3S(float, dpp_bf16); else DPP_W3S(float, float); }
#undef DPP_W3S
    return dpp_launch_status();
}

// 1 when dpp_fc_wgrad_stream takes the shape (whole 128 x 128 output blocks, 32-bit byte offsets), else 0
extern "C" int dpp_fc_wgrad_stream_ok(int Nb, int K, int N) {
    if (Nb < 1 || K < 128 || N < 128 || (K & 127) || (N & 127)) return 0;
    if ((long long)Nb * (K > N ? K : N) * 4 > 0x7fffffffLL) return 0;
    return 1;
}

extern "C" int dpp_fc_wgrad_stream(const float* X, const float* dY, float* dW, int Nb, int K, int N, const dpp_act* actX, int store,
                                   dpp_stream_t stream) {
    if (!X || !dY || !dW || (store & ~DPP_ST_B)) return DPP_E_BADARG;
    if (!dpp_fc_wgrad_stream_ok(Nb, K, N)) return DPP_E_UNSUPPORTED;
    FcWgradArgs a;
    a.X = X; a.dY = dY; a.dW = dW; a.Nb = Nb; a.K = K; a.N = N;
    a.actX.mean = a.actX.scale = a.actX.beta = nullptr; a.actX.mode = 0; a.actX.cmod = K; a.actX.x2 = a.actX.aux = nullptr; a.actX.out = nullptr;
    if (actX) a.actX = *actX;
    if (a.actX.mode & ~3) return DPP_E_UNSUPPORTED;
    if ((a.actX.mode & 2) && !(a.actX.mean && a.actX.scale && a.actX.beta && a.actX.cmod >= 4 && (a.actX.cmod & 3) == 0)) return DPP_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(dW)) & 15) return DPP_E_BADARG;
    if (store & DPP_ST_B) DPP_LAUNCH((fc_wgrad_stream_kernel<8, dpp_bf16>), dim3(K / 128, N / 128), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), a);
    else DPP_LAUNCH((fc_wgrad_stream_kernel<8>), dim3(K / 128, N / 128), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return dpp_launch_status();
}
