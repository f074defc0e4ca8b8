// conv3x3.hip -- 3x3 'half' stride-1 ConvLayer on NHWC maps for gfx950: implicit GEMM on f32 MFMA.
//
// Reference arithmetic: conv2d(border_mode='half') + bias, /root/reference/src/net/convlayer.py:230-240 (the
// bottleneck 3x3 of res_block, /root/reference/src/net/resnet.py:365-368, 394-397), its data gradient and its
// filter gradient (T.grad, /root/reference/src/trainer/poseregnettrainer.py:110-111).
//
// Forward / data gradient (conv3x3_kernel): a workgroup owns IMG x TH x TW output pixels (BM = 64 or 128
// rows).  The input halo (TH+2)x(TW+2) is staged ONCE into LDS with the BatchNorm+ReLU prologue applied (zero
// padding is applied AFTER the activation, as conv2d pads its already-activated input); the 9 taps then read
// shifted rows of that tile, so every input element is fetched from HBM/L2 once per workgroup instead of 9x.
// Per tap the [BN][Cin] weight slice is staged, and each wave issues v_mfma_f32_16x16x4_f32 with one
// ds_read_b128 per operand per 4 k-steps (lane (i,kq) owns channels 16*chunk + 4*kq + t).
// The data gradient is the same kernel run on dY with the mirrored/transposed weights produced by
// conv3x3_wtrans_kernel.
//
// Filter gradient (conv3x3_wgrad_kernel): same halo tile + the dY tile in LDS; the reduction runs over the
// tile's pixels, out tile = [Cout][Cin] per tap, (tap, 16x16 tile) pairs are dealt round-robin to the 4
// waves; per-workgroup partials are summed by dpp_reduce_partials in a fixed order.
#include <stdio.h>
#include <stdlib.h>
#include "dpp_common.h"
#include "conv3x3_wgrad_t.h"

namespace {

// Division constants of the staging loops.  A runtime integer division is ~25 instructions on this hardware and the staging
// loops had three per 16-byte slot; a wave issues about one instruction per 4 cycles, so they cost more than the loads they
// index (tools/inst_summary.py).  Halo positions are split with n / d = umulhi(n, floor(2^32 / d) + 1), exact for n * d < 2^32;
// channel quads are dealt to threads in power-of-two groups (QP >= quads per position; the lanes beyond the image idle).
struct C3Stage {
    unsigned m_hw2, m_tw2;   // magic of (TH+2)*(TW+2) and of TW+2
    int lqp;                 // log2(QP)
};

__host__ __device__ inline unsigned c3_magic(int d) { return (unsigned)(0x100000000ull / (unsigned)d) + 1u; }      // d >= 2

struct Conv3Args {
    const float* X;       // [N][H][W][Ci]
    int N, H, W, Ci, Co;
    dpp_act act;
    const float* Wk;      // [Co][9][Ci]
    int allw;             // all nine weight slices staged in LDS up front
    int wide;             // 16-byte epilogue through an LDS image of the tile (dpp_epilogue_wide)
    const float* bias;    // [Co] or null
    const float* residual;
    float* Y;             // [N][H][W][Co]
    dpp_epilogue epi;     // fused BatchNorm statistics / BatchNorm-backward epilogue
    int store;            // DPP_ST_* mask: X (A), Y + residual (C), epi.bn_x (BNX) hold bf16 elements
    int lth, ltw;         // log2 of tile height / width
    int img;              // images per workgroup
    int tiles_x, tiles_y;
    C3Stage sg;           // staging geometry (host-computed division constants)
    unsigned long long* prof;   // phase stamps (profiling build only, see dpp_stamp)
    int ntiles, woff;     // conv3x3_p_kernel: tiles in total (a workgroup walks several), byte offset of the weight images in LDS
};

typedef __bf16 c3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 c3_bf16x4 __attribute__((ext_vector_type(4)));

// LDS element type of the operand images: f32 (exact path) or bf16 (BASELINE config 5: operands rounded RNE after the
// prologue, f32 accumulation).  KP = channels per tap in the image: Ci, or at least 32 for bf16 (one 16x16x32 step; the
// 16-channel layers zero-fill the upper half).
template <int PREC> struct C3Prec;
template <> struct C3Prec<0> { typedef float elem; static constexpr int PAD = 4; __device__ static int kp(int Ci) { return Ci; } };
template <> struct C3Prec<1> { typedef __bf16 elem; static constexpr int PAD = 8; __device__ static int kp(int Ci) { return Ci < 32 ? 32 : Ci; } };

__device__ __forceinline__ void c3_store4(float* dst, float4 v) { *reinterpret_cast<float4*>(dst) = v; }
__device__ __forceinline__ void c3_store4(__bf16* dst, float4 v) {
    c3_bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<c3_bf16x4*>(dst) = o;
}

__device__ __forceinline__ void tile_origin(const Conv3Args& a, int bid, int& n0, int& y0, int& x0) {
    int bx = bid % a.tiles_x;
    int t = bid / a.tiles_x;
    int by = t % a.tiles_y;
    int bn = t / a.tiles_y;
    n0 = bn * a.img;
    y0 = by << a.lth;
    x0 = bx << a.ltw;
}

// Stage the activated input halo of the tile into LDS: Ah[(img*(TH+2)+hy)*(TW+2)+hx][KP+pad] (KP >= Ci: channels Ci..KP-1 zero).
// A thread owns ONE channel quad (its BatchNorm coefficients are loaded once) and walks halo positions; the loads of up to 8
// positions are issued before the first is used, so the usual tile (<= 8 slots per thread) costs one memory round trip.
// TX: element type of X in memory (float, or dpp_bf16 in the bf16 storage mode: 8-byte loads, widened exactly).
template <int UB, class E, class TX>
__device__ __forceinline__ void stage_halo(const TX* __restrict__ X, int N, int H, int W, int Ci, const dpp_act& act, int n0,
                                           int y0, int x0, int TH, int TW, int IMG, E* Ah, int LDA, int KP, const C3Stage& sg) {
    if (KP < Ci) KP = Ci;
    const int HW2 = (TH + 2) * (TW + 2), TW2 = TW + 2;
    const int HP = IMG * HW2;
    const int c0 = ((int)threadIdx.x & ((1 << sg.lqp) - 1)) * 4;
    const int hstep = DPP_THREADS >> sg.lqp;
    if (c0 >= KP) return;                            // idle lanes of the power-of-two quad group
    const bool cld = c0 < Ci;                        // bf16 images pad narrow layers to 32 channels with zeros
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), sc = mu, be = mu;
    if (cld && (act.mode & 2)) {
        mu = *reinterpret_cast<const float4*>(act.mean + c0);
        sc = *reinterpret_cast<const float4*>(act.scale + c0);
        be = *reinterpret_cast<const float4*>(act.beta + c0);
    }
    for (int hb = (int)threadIdx.x >> sg.lqp; hb < HP; hb += UB * hstep) {
        float4 v[UB];
        bool in[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int hp = hb + u * hstep;
            const int im = (int)__umulhi((unsigned)hp, sg.m_hw2), rem = hp - im * HW2;
            const int hy = (int)__umulhi((unsigned)rem, sg.m_tw2), hx = rem - hy * TW2;
            const int n = n0 + im, y = y0 + hy - 1, x = x0 + hx - 1;
            in[u] = hp < HP && cld && n < N && y >= 0 && y < H && x >= 0 && x < W;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in[u]) v[u] = dpp_ld4(X + (((size_t)n * H + y) * W + x) * Ci + c0);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int hp = hb + u * hstep;
            if (hp < HP) {
                float4 t = v[u];
                if (in[u]) {                         // zero padding is applied AFTER the activation
                    if (act.mode & 2) {
                        t.x = dpp_fma(t.x - mu.x, sc.x, be.x); t.y = dpp_fma(t.y - mu.y, sc.y, be.y);
                        t.z = dpp_fma(t.z - mu.z, sc.z, be.z); t.w = dpp_fma(t.w - mu.w, sc.w, be.w);
                    }
                    if (act.mode & 1) { t.x = fmaxf(t.x, 0.0f); t.y = fmaxf(t.y, 0.0f); t.z = fmaxf(t.z, 0.0f); t.w = fmaxf(t.w, 0.0f); }
                }
                c3_store4(&Ah[hp * LDA + c0], t);
            }
        }
    }
}

// EST: the epilogue may meet bf16-stored tensors (Y / residual / bn_x); TX: element type of X
template <int BM, int BN, int PREC = 0, class TX = float, bool EST = false>
__global__ __launch_bounds__(DPP_THREADS) void conv3x3_kernel(Conv3Args a) {
    dpp_kernarg_warm<sizeof(Conv3Args)>();
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* smem = reinterpret_cast<float*>(smem4);
    typedef typename C3Prec<PREC>::elem elem;
    constexpr int RM = BM / 64;
    constexpr int CN = BN / 16;
    constexpr int KSTEP = PREC ? 32 : 16;       // channels consumed per fragment read (one b128 per operand)
    const int TH = 1 << a.lth, TW = 1 << a.ltw;
    const int Ci = a.Ci, KP = C3Prec<PREC>::kp(Ci), LDA = KP + C3Prec<PREC>::PAD;
    const int HP = a.img * (TH + 2) * (TW + 2);
    elem* Ah = reinterpret_cast<elem*>(smem);
    elem* Bs = Ah + HP * LDA;                    // [BN][KP+pad] per slice
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    int n0, y0, x0;
    tile_origin(a, blockIdx.x, n0, y0, x0);
    const int col0 = blockIdx.y * BN;

    dpp_stamp(a.prof, 0);
    dpp_wide_coef wco;
    if (a.wide) wco.load<BN>(col0, a.Co, a.bias, a.epi, a.Y);
    // halo slots per thread: 3 for the 16- and 32-channel layers, 7 for the 64-channel ones; unrolling 8 for 3 wastes 5 slots of index arithmetic
    if (((HP + (DPP_THREADS >> a.sg.lqp) - 1) >> (8 - a.sg.lqp)) <= 4)
        stage_halo<4>(reinterpret_cast<const TX*>(a.X), a.N, a.H, a.W, Ci, a.act, n0, y0, x0, TH, TW, a.img, Ah, LDA, KP, a.sg);
    else
        stage_halo<8>(reinterpret_cast<const TX*>(a.X), a.N, a.H, a.W, Ci, a.act, n0, y0, x0, TH, TW, a.img, Ah, LDA, KP, a.sg);
    dpp_stamp(a.prof, 1);

    // halo index of this lane's A rows (centre tap)
    int hbase[RM];
#pragma unroll
    for (int rt = 0; rt < RM; ++rt) {
        int row = wave * (BM / 4) + rt * 16 + l15;
        int im = row >> (a.lth + a.ltw);
        int ty = (row >> a.ltw) & (TH - 1);
        int tx = row & (TW - 1);
        hbase[rt] = (im * (TH + 2) + ty + 1) * (TW + 2) + tx + 1;
    }

    f32x4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weight slices are double-buffered in LDS: the slice of tap t+1 is fetched into registers before the MFMAs of tap t
    // and written to the other buffer after them, so one barrier per tap suffices and the fetch latency is hidden.
    // Slot s of a thread is (output column j, channel quad c4) of the weight slice, dealt in the same power-of-two quad groups
    // as the halo: the pointer and the LDS offset are formed once, a tap is one load (+ tap*Ci) and one store per slot.
    constexpr int WSLOTS = (BN * 16 + DPP_THREADS - 1) / DPP_THREADS;      // KP <= 64 -> at most 16 quads per column
    float4 wreg[WSLOTS];
    const float* wp[WSLOTS];
    int wo[WSLOTS];
#pragma unroll
    for (int s = 0; s < WSLOTS; ++s) {
        const int slot = tid + s * DPP_THREADS;
        const int j = slot >> a.sg.lqp, c0 = (slot & ((1 << a.sg.lqp) - 1)) * 4;
        wo[s] = (j < BN && c0 < KP) ? j * LDA + c0 : -1;
        wp[s] = (j < BN && c0 < Ci && col0 + j < a.Co) ? a.Wk + (size_t)(col0 + j) * 9 * Ci + c0 : nullptr;
    }
    auto wfetch = [&](int tap) {
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            wreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wp[s] != nullptr) wreg[s] = *reinterpret_cast<const float4*>(wp[s] + tap * Ci);
        }
    };
    auto wcommit = [&](elem* dst) {
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s)
            if (wo[s] >= 0) c3_store4(&dst[wo[s]], wreg[s]);
    };
    // Narrow layers (stages 1-2: Ci = 16 / 32) have all nine weight slices staged up front (11-21 KB): one barrier for the
    // whole workgroup instead of one per tap, which is what bounds these short K loops.
    const bool allw = a.allw != 0;
    if (allw) {
#pragma unroll
        for (int t3 = 0; t3 < 9; t3 += 3) {          // three taps' loads in flight at a time
            float4 w3[3][WSLOTS];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int s = 0; s < WSLOTS; ++s) {
                    w3[u][s] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (wp[s] != nullptr) w3[u][s] = *reinterpret_cast<const float4*>(wp[s] + (t3 + u) * Ci);
                }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int s = 0; s < WSLOTS; ++s)
                    if (wo[s] >= 0) c3_store4(&Bs[(t3 + u) * BN * LDA + wo[s]], w3[u][s]);
        }
    } else {
        wfetch(0);
        wcommit(Bs);
    }
    __syncthreads();                // halo + first slice (or all slices) visible
    dpp_stamp(a.prof, 2);
    for (int tap = 0; tap < 9; ++tap) {
        const elem* Bcur = allw ? Bs + tap * BN * LDA : Bs + (tap & 1) * BN * LDA;
        if (!allw && tap + 1 < 9) wfetch(tap + 1);
        const int toff = (tap / 3 - 1) * (TW + 2) + (tap % 3 - 1);
        for (int kc = 0; kc < KP; kc += KSTEP) {
            if constexpr (PREC == 0) {
                float4 av[RM], bv[CN];
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
                    av[rt] = *reinterpret_cast<const float4*>(&Ah[(hbase[rt] + toff) * LDA + kc + kq * 4]);
#pragma unroll
                for (int ct = 0; ct < CN; ++ct)
                    bv[ct] = *reinterpret_cast<const float4*>(&Bcur[(ct * 16 + l15) * LDA + kc + kq * 4]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CN; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), dpp_f4_get(bv[ct], t),
                                                                               acc[rt][ct], 0, 0, 0);
            } else {
                c3_bf16x8 av[RM], bv[CN];
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
                    av[rt] = *reinterpret_cast<const c3_bf16x8*>(&Ah[(hbase[rt] + toff) * LDA + kc + kq * 8]);
#pragma unroll
                for (int ct = 0; ct < CN; ++ct)
                    bv[ct] = *reinterpret_cast<const c3_bf16x8*>(&Bcur[(ct * 16 + l15) * LDA + kc + kq * 8]);
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CN; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rt], bv[ct], acc[rt][ct], 0, 0, 0);
            }
        }
        if (!allw) {
            if (tap + 1 < 9) wcommit(Bs + ((tap + 1) & 1) * BN * LDA);
            __syncthreads();
        }
    }
    if (allw) __syncthreads();      // the epilogue reuses the operand images as scratch
    dpp_stamp(a.prof, 3);

    if (a.wide) {
        const int vi = (a.N - n0 < a.img) ? (a.N - n0) : a.img;
        const int vy = (a.H - y0 < TH) ? (a.H - y0) : TH;
        const int vx = (a.W - x0 < TW) ? (a.W - x0) : TW;
        dpp_epilogue_wide<RM, CN, 4, 1, BM, BN, 1, EST>(acc, smem, col0, a.Co, wco, a.residual, a.Y, a.epi, vi * vy * vx, wave, 0, l15, kq,
                                                [&](int rl) {
            const int im = rl >> (a.lth + a.ltw);
            const int ty = (rl >> a.ltw) & (TH - 1);
            const int tx = rl & (TW - 1);
            const int n = n0 + im, y = y0 + ty, x = x0 + tx;
            const bool ok = !(im >= a.img || n >= a.N || y >= a.H || x >= a.W);
            return ok ? (long)((((size_t)n * a.H + y) * a.W + x) * a.Co) : -1L;
        }, 0, a.store);
        dpp_stamp(a.prof, 4);
        return;
    }

    const dpp_epilogue& ep = a.epi;
    const bool fused = ep.stats != nullptr || ep.bn_x != nullptr;
    float sx[CN], sy[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) { sx[ct] = 0.0f; sy[ct] = 0.0f; }
    // per-column vectors are read once (the stores to Y below could alias them as far as the compiler knows)
    float cbias[CN], cmean[CN], cscale[CN], cbeta[CN], cistd[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) {
        const int col = col0 + ct * 16 + l15;
        const bool in = col < a.Co;
        cbias[ct] = (in && a.bias) ? a.bias[col] : 0.0f;
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
            int row = wave * (BM / 4) + rt * 16 + kq * 4 + r;
            int im = row >> (a.lth + a.ltw);
            int ty = (row >> a.ltw) & (TH - 1);
            int tx = row & (TW - 1);
            int n = n0 + im, y = y0 + ty, x = x0 + tx;
            bool ok = !(im >= a.img || n >= a.N || y >= a.H || x >= a.W);
            size_t o = ok ? (((size_t)n * a.H + y) * a.W + x) * a.Co : 0;
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                int col = col0 + ct * 16 + l15;
                float v = 0.0f;
                if (ok && col < a.Co) {
                    v = acc[rt][ct][r] + cbias[ct];
                    if (a.residual) v += a.residual[o + col];
                    if (ep.bn_x != nullptr) {
                        float dx = ep.bn_x[o + col] - cmean[ct];
                        if (ep.bn_relu && dx * cscale[ct] + cbeta[ct] < 0.0f) v = 0.0f;
                        sx[ct] += v;
                        sy[ct] += v * (dx * cistd[ct]);
                    }
                    a.Y[o + col] = v;
                }
                acc[rt][ct][r] = v;
            }
        }
    }
    if (fused) {
        float* red = smem;                             // the operand images are dead after the last tap's barrier
        const int cbase = col0 + l15;
        if (ep.bn_x != nullptr && ep.bn_partial != nullptr) {
            dpp_tile_colsum<CN, 4, 1, BN>(sx, red, wave, 0, l15, kq);
            dpp_tile_colsum<CN, 4, 1, BN>(sy, red, wave, 0, l15, kq);
            if (kq == 0 && wave == 0) {
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    int col = cbase + ct * 16;
                    if (col < a.Co) {
                        ep.bn_partial[dpp_partial_index(0, col, blockIdx.x, a.Co, gridDim.x)] = sx[ct];
                        ep.bn_partial[dpp_partial_index(1, col, blockIdx.x, a.Co, gridDim.x)] = sy[ct];
                    }
                }
            }
        }
        if (ep.stats != nullptr) {
            const int vi = (a.N - n0 < a.img) ? (a.N - n0) : a.img;
            const int vy = (a.H - y0 < TH) ? (a.H - y0) : TH;
            const int vx = (a.W - x0 < TW) ? (a.W - x0) : TW;
            const float nvalid = (float)(vi * vy * vx);
            float sm[CN], m2[CN];
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                float t = 0.0f;
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t += acc[rt][ct][r];
                sm[ct] = t;
            }
            dpp_tile_colsum<CN, 4, 1, BN>(sm, red, wave, 0, l15, kq);
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                sm[ct] = sm[ct] / nvalid;
                float t = 0.0f;
#pragma unroll
                for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row = wave * (BM / 4) + rt * 16 + kq * 4 + r;
                        int im = row >> (a.lth + a.ltw);
                        int ty = (row >> a.ltw) & (TH - 1);
                        int tx = row & (TW - 1);
                        bool ok = !(im >= a.img || n0 + im >= a.N || y0 + ty >= a.H || x0 + tx >= a.W);
                        float dv = acc[rt][ct][r] - sm[ct];
                        if (ok) t += dv * dv;
                    }
                m2[ct] = t;
            }
            dpp_tile_colsum<CN, 4, 1, BN>(m2, red, wave, 0, l15, kq);
            if (kq == 0 && wave == 0) {
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    int col = cbase + ct * 16;
                    if (col < a.Co) {
                        ep.stats[dpp_partial_index(0, col, blockIdx.x, a.Co, gridDim.x)] = sm[ct];
                        ep.stats[dpp_partial_index(1, col, blockIdx.x, a.Co, gridDim.x)] = m2[ct];
                    }
                }
            }
        }
    }
}

// ---- the narrow layers as a tile-WALKING kernel (round 6) -------------------------------------------------------------------------
// conv3x3_kernel gives every 128-pixel tile a workgroup of its own: entry code, the halo's round trip to memory, the nine weight
// slices' round trip, 18-72 MFMAs per wave, the epilogue's round trip for residual / BatchNorm input -- three dependent memory
// latencies around a microsecond of arithmetic, hidden only by how many such workgroups a CU holds (16-channel layers at 256 x 256
// input: 4 096 workgroups, 48 us for 58 MB = 1.2 TB/s).  Here a workgroup stages the nine weight slices ONCE and
// walks tiles blockIdx.x, + gridDim.x, ...: the next tile's halo is requested (unconditional, clamped loads) right after the barrier that
// publishes the current one and travels under the current tile's products and epilogue; it is committed to LDS -- prologue applied
// there -- at the top of the next round.  Everything else is conv3x3_kernel's: same images, same tap loop, same 16-byte epilogue
// through the LDS tile image, and the statistics partials keep ONE block per tile (dpp_epilogue_wide(blk = tile)), so callers size and
// finalize them exactly as before.  Narrow layers only (all nine weight slices resident), the wide epilogue only.
// Everything about the geometry is a compile-time constant here: CI input channels, BN output columns per workgroup (blockIdx.y picks the
// column tile), and the tile -- GEOM 0: 8 x 16 pixels of one image (power-of-two tile counts per image), GEOM 1: two whole 8 x 8 images (the
// 64-channel layers of stages 3-4 at 128 x 128 input).  The generic kernel's run-time strides, division constants and slot tables cost it
// 150-160 registers (3 workgroups per CU).
template <int CI, int BN, int PREC, class TX, bool EST, int GEOM>
__global__ __launch_bounds__(DPP_THREADS) DPP_WAVES_PER_EU((CI * (PREC ? 2 : 4) * (9 * BN + 200) > 40 * 1024) ? 1 : 4, 8) void conv3x3_p_kernel(Conv3Args a) {
    dpp_kernarg_warm<sizeof(Conv3Args)>();
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* smem = reinterpret_cast<float*>(smem4);
    typedef typename C3Prec<PREC>::elem elem;
    constexpr int BM = 128, RM = 2, CN = BN / 16, TH = 8, TW = GEOM ? 8 : 16, LTH = 3, LTW = GEOM ? 3 : 4, IMG = GEOM ? 2 : 1, TW2 = TW + 2, HW2 = (TH + 2) * TW2,
                  HP = IMG * HW2;
    constexpr int Ci = CI, KP = PREC ? (CI < 32 ? 32 : CI) : CI, LDA = KP + C3Prec<PREC>::PAD, KSTEP = PREC ? 32 : 16;
    constexpr int QP = KP / 4, HSTEP = DPP_THREADS / QP, SLOTS = (HP + HSTEP - 1) / HSTEP;     // (QP is a power of two: 4, 8 or 16)
    constexpr int CW = 64;                                                                  // floats per per-channel vector in LDS
    elem* Ah = reinterpret_cast<elem*>(smem);
    elem* Bs = reinterpret_cast<elem*>(reinterpret_cast<char*>(smem) + a.woff);          // [9][BN][KP+pad], behind the halo / epilogue image
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const TX* const Xg = reinterpret_cast<const TX*>(a.X);
    const int H = a.H, W = a.W;
    constexpr bool SPLIT = GEOM != 0 || BN != CI;                                // column tiles over blockIdx.y (else: a square layer's whole column range)
    const int col0 = SPLIT ? (int)blockIdx.y * BN : 0, Co = SPLIT ? a.Co : BN;

    // Per-channel vectors (epilogue, this workgroup's BN columns: bias, BatchNorm-backward mean / scale / beta / inv_std; prologue, the CI input
    // channels: mean / scale / beta) live in LDS, staged once: in registers they are 32 values held across the whole walk.  Cf[v][CW]; an absent
    // vector is never read.
    float* Cf = reinterpret_cast<float*>(reinterpret_cast<char*>(Bs) + 9 * BN * LDA * sizeof(elem));
    const bool on_bn = a.epi.bn_x != nullptr;

    // ---- halo slots of this thread: (position hp0 + u * HSTEP, channel quad c0) ----
    const int c0 = (tid & (QP - 1)) * 4, hp0 = tid / QP;
    const bool cld = c0 < Ci;                          // bf16 images pad the 16-channel layers to 32 channels with zeros: those lanes only store
    typedef typename std::conditional<std::is_same<TX, float>::value, float4, uint2>::type raw_t;
    raw_t hr[SLOTS];
    unsigned hin = 0;
    auto origin = [&](int tile, int& n0, int& y0, int& x0) {
        if (GEOM) { n0 = tile * IMG; y0 = 0; x0 = 0; return; }
        x0 = (tile & (a.tiles_x - 1)) << LTW;          // tiles_x, tiles_y are powers of two (a.lth / a.ltw carry their logs here)
        y0 = ((tile >> a.ltw) & (a.tiles_y - 1)) << LTH;
        n0 = tile >> (a.ltw + a.lth);
    };
    auto fetch = [&](int tile) {
        int n0, y0, x0;
        origin(tile, n0, y0, x0);
        const int nimg = GEOM ? ((a.N - n0 < IMG) ? a.N - n0 : IMG) : 1;
        const TX* const Xn = Xg + (size_t)n0 * H * W * Ci + c0;
        hin = 0;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int hp = hp0 + u * HSTEP, im = GEOM ? hp / HW2 : 0, rem = hp - im * HW2, hy = rem / TW2, hx = rem - hy * TW2;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            const bool in = hp < HP && cld && im < nimg && y >= 0 && y < H && x >= 0 && x < W;
            hin |= in ? (1u << u) : 0u;
            hr[u] = *reinterpret_cast<const raw_t*>(Xn + (in ? ((im * H + y) * W + x) * Ci : 0));       // unconditional, clamped
        }
    };
    auto commit = [&]() {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), sc = mu, be = mu;
        if (cld && (a.act.mode & 2)) {
            mu = *reinterpret_cast<const float4*>(&Cf[5 * CW + c0]);
            sc = *reinterpret_cast<const float4*>(&Cf[6 * CW + c0]);
            be = *reinterpret_cast<const float4*>(&Cf[7 * CW + c0]);
        }
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int hp = hp0 + u * HSTEP;
            if (hp >= HP) continue;
            float4 t;
            if constexpr (std::is_same<TX, float>::value) t = hr[u];
            else t = make_float4(__uint_as_float(hr[u].x << 16), __uint_as_float(hr[u].x & 0xffff0000u), __uint_as_float(hr[u].y << 16),
                                 __uint_as_float(hr[u].y & 0xffff0000u));
            if ((hin >> u) & 1u) {                       // zero padding is applied AFTER the activation
                if (a.act.mode & 2) {
                    t.x = dpp_fma(t.x - mu.x, sc.x, be.x); t.y = dpp_fma(t.y - mu.y, sc.y, be.y);
                    t.z = dpp_fma(t.z - mu.z, sc.z, be.z); t.w = dpp_fma(t.w - mu.w, sc.w, be.w);
                }
                if (a.act.mode & 1) { t.x = fmaxf(t.x, 0.0f); t.y = fmaxf(t.y, 0.0f); t.z = fmaxf(t.z, 0.0f); t.w = fmaxf(t.w, 0.0f); }
            } else t = make_float4(0.f, 0.f, 0.f, 0.f);
            c3_store4(&Ah[hp * LDA + c0], t);
        }
    };

    int tile = blockIdx.x;
    if (tile < a.ntiles) fetch(tile);
    {   // the per-channel vectors: thread (v, quad) copies one 16-byte piece (an absent vector: the weights' first bytes, never read back)
        constexpr int QC = CW / 4;
        const int v = tid / QC, q4 = (tid % QC) * 4;
        const float* src = nullptr;
        if (v == 0) src = a.bias;
        if (on_bn) { if (v == 1) src = a.epi.bn_mean; if (v == 2) src = a.epi.bn_scale; if (v == 3) src = a.epi.bn_beta; if (v == 4) src = a.epi.bn_inv_std; }
        if (src != nullptr) src += col0;               // the epilogue's vectors: this workgroup's columns
        if (a.act.mode & 2) { if (v == 5) src = a.act.mean; if (v == 6) src = a.act.scale; if (v == 7) src = a.act.beta; }
        const bool have = src != nullptr && q4 < (v < 5 ? BN : Ci);
        const float4 cv = *reinterpret_cast<const float4*>(have ? src + q4 : a.Wk);
        if (v < 8) *reinterpret_cast<float4*>(&Cf[v * CW + q4]) = cv;
    }
    // ---- all nine weight slices, once: every load is UNCONDITIONAL (a slot outside the slice reads the first weights and is zeroed at the
    // commit) and issued behind the first halo's, so the entry code is ONE memory round trip.  With `ld ? load : zero` the compiler put
    // each load in a branch of its own that ends on s_waitcnt vmcnt(0): nine serialized round trips ahead of the first tile.
    {
        constexpr int WSLOTS = (BN * QP + DPP_THREADS - 1) / DPP_THREADS;
        constexpr int WB = WSLOTS == 1 ? 9 : (WSLOTS == 2 ? 5 : 3);          // taps in flight at a time (36 / 40 / 48 registers)
#pragma unroll
        for (int t0 = 0; t0 < 9; t0 += WB) {
            float4 wv[WB][WSLOTS];
#pragma unroll
            for (int s = 0; s < WSLOTS; ++s) {
                const int slot = tid + s * DPP_THREADS;
                const int j = slot / QP, cc = (slot & (QP - 1)) * 4;
                const bool wl = j < BN && cc < Ci && col0 + j < Co;
                const float* wp = a.Wk + (wl ? (size_t)(col0 + j) * 9 * Ci + cc : (size_t)0);
#pragma unroll
                for (int u = 0; u < WB; ++u)
                    if (t0 + u < 9) wv[u][s] = *reinterpret_cast<const float4*>(wp + (wl ? (t0 + u) * Ci : 0));
            }
#pragma unroll
            for (int u = 0; u < WB; ++u)
#pragma unroll
                for (int s = 0; s < WSLOTS; ++s) {
                    const int slot = tid + s * DPP_THREADS;
                    const int j = slot / QP, cc = (slot & (QP - 1)) * 4;
                    if (t0 + u < 9 && j < BN)
                        c3_store4(&Bs[(t0 + u) * BN * LDA + j * LDA + cc], (cc < Ci && col0 + j < Co) ? wv[u][s] : make_float4(0.f, 0.f, 0.f, 0.f));
                }
        }
    }
    int hbase[RM];
#pragma unroll
    for (int rt = 0; rt < RM; ++rt) {
        const int row = wave * (BM / 4) + rt * 16 + l15;
        hbase[rt] = ((row >> (LTH + LTW)) * (TH + 2) + ((row >> LTW) & (TH - 1)) + 1) * TW2 + (row & (TW - 1)) + 1;
    }
    for (; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();                               // the previous tile's epilogue is done with the image that shares the halo's space
        commit();
        __syncthreads();                               // halo (and, first round, the weight slices) visible
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
        f32x4 acc[RM][CN];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < CN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const elem* Bcur = Bs + tap * BN * LDA;
            const int toff = (tap / 3 - 1) * TW2 + (tap % 3 - 1);
#pragma unroll
            for (int kc = 0; kc < KP; kc += KSTEP) {
                if constexpr (PREC == 0) {
                    float4 av[RM], bv[CN];
#pragma unroll
                    for (int rt = 0; rt < RM; ++rt) av[rt] = *reinterpret_cast<const float4*>(&Ah[(hbase[rt] + toff) * LDA + kc + kq * 4]);
#pragma unroll
                    for (int ct = 0; ct < CN; ++ct) bv[ct] = *reinterpret_cast<const float4*>(&Bcur[(ct * 16 + l15) * LDA + kc + kq * 4]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                            for (int ct = 0; ct < CN; ++ct)
                                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), dpp_f4_get(bv[ct], t), acc[rt][ct], 0, 0, 0);
                } else {
                    c3_bf16x8 av[RM], bv[CN];
#pragma unroll
                    for (int rt = 0; rt < RM; ++rt) av[rt] = *reinterpret_cast<const c3_bf16x8*>(&Ah[(hbase[rt] + toff) * LDA + kc + kq * 8]);
#pragma unroll
                    for (int ct = 0; ct < CN; ++ct) bv[ct] = *reinterpret_cast<const c3_bf16x8*>(&Bcur[(ct * 16 + l15) * LDA + kc + kq * 8]);
#pragma unroll
                    for (int rt = 0; rt < RM; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CN; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rt], bv[ct], acc[rt][ct], 0, 0, 0);
                }
            }
        }
        __syncthreads();                               // the epilogue's image takes the halo's place
        int n0, y0, x0;
        origin(tile, n0, y0, x0);
        const long obase = (((long)n0 * H + y0) * W + x0) * Co;
        const int nimg = (a.N - n0 < IMG) ? a.N - n0 : IMG;            // (GEOM 1: an odd batch leaves the last tile one image)
        dpp_wide_coef wco;
        wco.on_bias = a.bias != nullptr; wco.on_bn = on_bn;
        {
            const int cq4 = (tid % (BN / 4)) * 4;
#pragma unroll
            for (int v = 0; v < 5; ++v) wco.raw[v] = *reinterpret_cast<const float4*>(&Cf[v * CW + cq4]);
        }
        dpp_epilogue_wide<RM, CN, 4, 1, BM, BN, 1, EST>(acc, smem, col0, Co, wco, a.residual, a.Y, a.epi, nimg * TH * TW, wave, 0, l15, kq,
                                                [&](int rl) {
            const int im = rl >> (LTH + LTW), ty = (rl >> LTW) & (TH - 1), tx = rl & (TW - 1);
            return im < nimg ? obase + (long)(((im * H + ty) * W + tx) * Co) : -1L;
        }, 0, a.store, tile, a.ntiles);
    }
}

// Wd[c][8 - tap][o] = Wk[o][tap][c]: the weights of the data-gradient correlation (mirrored taps, channels swapped)
__global__ __launch_bounds__(DPP_THREADS) void conv3x3_wtrans_kernel(const float* __restrict__ Wk, int Co, int Ci, float* __restrict__ Wd) {
    int n = Co * 9 * Ci;
    for (int i = blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += gridDim.x * DPP_THREADS) {
        int o = i % Co;
        int t = (i / Co) % 9;
        int c = i / (Co * 9);
        Wd[i] = Wk[((size_t)o * 9 + (8 - t)) * Ci + c];      // i indexes Wd[c][t][o]
    }
}

// The same for every 3x3 layer of a net in ONE launch (the engine prepares all mirrored weight sets at the start of a step, on the
// side stream under the forward pass: twenty tiny launches there were worth 0.045 ms of interference with the forward chain).
struct WtransJob { const float* Wk; float* Wd; int Co, Ci, block0, pad; };
__global__ __launch_bounds__(DPP_THREADS) void conv3x3_wtrans_multi_kernel(const WtransJob* __restrict__ jobs, int njobs) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].block0) ++j;
    const WtransJob jb = jobs[j];
    const int i = ((int)blockIdx.x - jb.block0) * DPP_THREADS + threadIdx.x;
    if (i >= jb.Co * 9 * jb.Ci) return;
    const int o = i % jb.Co, t = (i / jb.Co) % 9, c = i / (jb.Co * 9);
    jb.Wd[i] = jb.Wk[((size_t)o * 9 + (8 - t)) * jb.Ci + c];
}

struct Wgrad3Args {
    const float* X;       // [N][H][W][Ci] forward input of the conv (pre-activation source)
    int N, H, W, Ci, Co;
    dpp_act act;
    const float* dY;      // [N][H][W][Co]
    float* partial;       // [nblk][Co][9][Ci]
    int lth, ltw, img, tiles_x, tiles_y;
    int ntiles;           // spatial tiles in total (a workgroup walks several)
    int taps_pb;          // taps per blockIdx.y
    C3Stage sg;           // staging geometry of the halo (host-computed division constants)
    int lqy;              // log2 of the power-of-two quad group of the dY rows (>= Co/4 quads per row)
};

template <int BM, int MAXACC, class TX = float, class TY = float>
__global__ __launch_bounds__(DPP_THREADS) void conv3x3_wgrad_kernel(Wgrad3Args a) {
    dpp_kernarg_warm<sizeof(Wgrad3Args)>();
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* smem = reinterpret_cast<float*>(smem4);
    const int TH = 1 << a.lth, TW = 1 << a.ltw;
    const int Ci = a.Ci, Co = a.Co, LDA = Ci + 4, LDY = Co + 4;
    const int HP = a.img * (TH + 2) * (TW + 2);
    float* Ah = smem;                 // [HP][Ci+4]
    float* Ys = smem + HP * LDA;      // [BM][Co+4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    Conv3Args ta;                      // reuse tile_origin
    ta.tiles_x = a.tiles_x; ta.tiles_y = a.tiles_y; ta.img = a.img; ta.lth = a.lth; ta.ltw = a.ltw;
    const int nto = Co >> 4, ntc = Ci >> 4, NT = nto * ntc;
    const int tap0 = blockIdx.y * a.taps_pb;
    const int pairs = a.taps_pb * NT;
    f32x4 acc[MAXACC];
#pragma unroll
    for (int i = 0; i < MAXACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (tap, 16x16 tile) of accumulator i and this lane's operand offsets, formed ONCE: inside the reduction loop the divisions
    // that split the pair index cost 20x the MFMAs they fed (4 100-4 900 VALU instructions per wave, tools/inst_summary.py)
    int yoff[MAXACC], xoff[MAXACC];
#pragma unroll
    for (int i = 0; i < MAXACC; ++i) {
        const int p = wave + 4 * i;
        yoff[i] = -1; xoff[i] = 0;
        if (p < pairs) {
            const int tap = tap0 + p / NT, tile = p % NT;
            const int to = tile / ntc, tc = tile - to * ntc;
            yoff[i] = to * 16 + l15;
            xoff[i] = ((tap / 3 - 1) * (TW + 2) + (tap % 3 - 1)) * LDA + tc * 16 + l15;
        }
    }

    // a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... keeping its accumulators in registers, so the
    // number of partial slices (and the traffic of the reduce) is bounded by the grid size, not by the tile count
    for (int tile_id = blockIdx.x; tile_id < a.ntiles; tile_id += gridDim.x) {
        int n0, y0, x0;
        tile_origin(ta, tile_id, n0, y0, x0);
        __syncthreads();               // previous tile's fragments are consumed before LDS is overwritten
        stage_halo<4>(reinterpret_cast<const TX*>(a.X), a.N, a.H, a.W, Ci, a.act, n0, y0, x0, TH, TW, a.img, Ah, LDA, Ci, a.sg);
        {   // stage dY rows of the tile (zeros for out-of-range rows): a thread owns one channel quad and walks rows
            const int c0 = (tid & ((1 << a.lqy) - 1)) * 4, rstep = DPP_THREADS >> a.lqy;
            if (c0 < Co) {
                for (int row = tid >> a.lqy; row < BM; row += rstep) {
                    int im = row >> (a.lth + a.ltw);
                    int ty = (row >> a.ltw) & (TH - 1);
                    int tx = row & (TW - 1);
                    int n = n0 + im, y = y0 + ty, x = x0 + tx;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (im < a.img && n < a.N && y < a.H && x < a.W)
                        v = dpp_ld4(reinterpret_cast<const TY*>(a.dY) + (((size_t)n * a.H + y) * a.W + x) * Co + c0);     // (TY: dY may be bf16-stored, DPP_ST_B)
                    *reinterpret_cast<float4*>(&Ys[row * LDY + c0]) = v;
                }
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int rc = 0; rc < BM; rc += 16) {
            // halo index of the 4 reduction rows this lane feeds: r = rc + 4*kq + t
            int hb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int row = rc + kq * 4 + t;
                int im = row >> (a.lth + a.ltw);
                int ty = (row >> a.ltw) & (TH - 1);
                int tx = row & (TW - 1);
                hb[t] = ((im * (TH + 2) + ty + 1) * (TW + 2) + tx + 1) * LDA;
            }
            const float* yrow = Ys + (rc + kq * 4) * LDY;
#pragma unroll
            for (int i = 0; i < MAXACC; ++i) {
                if (yoff[i] >= 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float av = yrow[t * LDY + yoff[i]];
                        float bv = Ah[hb[t] + xoff[i]];
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }

    float* out = a.partial + (size_t)blockIdx.x * Co * 9 * Ci;
#pragma unroll
    for (int i = 0; i < MAXACC; ++i) {
        int p = wave + 4 * i;
        if (p < pairs) {
            int tap = tap0 + p / NT, tile = p % NT;
            int to = tile / ntc, tc = tile - to * ntc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int o = to * 16 + kq * 4 + r, c = tc * 16 + l15;
                out[((size_t)o * 9 + tap) * Ci + c] = acc[i][r];
            }
        }
    }
}

constexpr int WGRAD_MAX_BLOCKS = 256;   // partial slices of the 3x3 filter gradient

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// tile geometry: TH x TW x IMG = BM output pixels per workgroup
void pick_tile(int N, int H, int W, int bm, int& lth, int& ltw, int& img) {
    int tw = 1; while (tw * 2 <= W && tw < 16) tw *= 2;
    int th = 1; while (th * 2 <= H && th < 8 && th * tw * 2 <= bm) th *= 2;
    while (th * tw > bm) tw /= 2;
    img = bm / (th * tw);
    lth = ilog2(th); ltw = ilog2(tw);
}

}  // namespace

extern "C" int dpp_conv3x3_tiling(int N, int H, int W, int bm, int* th, int* tw, int* img) {
    int lth, ltw, im;
    if (bm != 64 && bm != 128) return -1;
    pick_tile(N, H, W, bm, lth, ltw, im);
    if (th) *th = 1 << lth;
    if (tw) *tw = 1 << ltw;
    if (img) *img = im;
    return dpp_cdiv(W, 1 << ltw) * dpp_cdiv(H, 1 << lth) * dpp_cdiv(N, im);
}

static int conv3x3_launch(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* Wk, int Co,
                          const float* bias, const float* residual, float* Y, int bm, const dpp_epilogue* epi, int precision, int store,
                          dpp_stream_t stream) {
    if (!X || !Wk || !Y || N < 1 || H < 1 || W < 1 || Ci < 16 || (Ci & 15) || Co < 16 || (Co & 15)) return DPP_E_BADARG;
    if (store & ~(DPP_ST_A | DPP_ST_C | DPP_ST_BNX)) return DPP_E_BADARG;
    Conv3Args a;
    a.store = store;
    a.X = X; a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
    a.prof = dpp_prof_buffer;
    if (act && (act->mode & 4)) return DPP_E_UNSUPPORTED;       // the two-tensor BatchNorm-backward operand is a dpp_gemm feature
    if (act) a.act = *act; else { a.act.mode = 0; a.act.cmod = 1; a.act.mean = a.act.scale = a.act.beta = nullptr; }
    a.Wk = Wk; a.bias = bias; a.residual = residual; a.Y = Y;
    if (epi) a.epi = *epi; else a.epi = dpp_epilogue{};      // (all pointers null: the alignment test of the wide epilogue below reads every one)
    if (a.epi.bn_x && !(a.epi.bn_mean && a.epi.bn_inv_std && a.epi.bn_scale && a.epi.bn_beta && a.epi.bn_partial)) return DPP_E_BADARG;
    long pixels = (long)N * H * W;
    if (bm == 0) bm = (pixels / 128) * dpp_cdiv(Co, 64) >= 512 ? 128 : 64;
    pick_tile(N, H, W, bm, a.lth, a.ltw, a.img);
    int TH = 1 << a.lth, TW = 1 << a.ltw;
    a.tiles_x = dpp_cdiv(W, TW); a.tiles_y = dpp_cdiv(H, TH);
    int nblk = a.tiles_x * a.tiles_y * dpp_cdiv(N, a.img);
    if (Ci > 64) return DPP_E_UNSUPPORTED;          // weight-slice register staging is sized for Ci <= 64
    a.sg.m_hw2 = c3_magic((TH + 2) * (TW + 2));
    a.sg.m_tw2 = c3_magic(TW + 2);
    a.sg.lqp = ilog2(((precision && Ci < 32) ? 32 : Ci) / 4);
    int bn = Co >= 64 ? 64 : (Co >= 32 ? 32 : 16);
    // The column tile is halved while the grid is below ~2 workgroups per CU.  Round 5, same-box A/B of the bs128 step over 200 steps
    // (tools/ab_r05.sh, profiles/r05_ab.txt): 1024 (rounds 1-4) 3.463 / 3.462 ms, 512 3.440, 256 3.548 -- with 512 the 32-channel layers of
    // stage 2 (512 pixel tiles) keep their 32 columns together (one halo staging + BatchNorm prologue per tile instead of two), the
    // 64-channel layers of stages 3-4 (128 tiles) still go to four workgroups per tile; with 256 those serialise (the VERDICT r4 proposal:
    // measured slower).  256x256 bf16: 8.214 -> 8.160 ms.  DPP_C3_MIN_WGS overrides (experiments).
    static const long min_wgs = []() { const char* e = getenv("DPP_C3_MIN_WGS"); return (e && atol(e) > 0) ? atol(e) : 512L; }();
    while (bn > 16 && (long)nblk * dpp_cdiv(Co, bn) < min_wgs) bn >>= 1;
    size_t halo = (size_t)a.img * (TH + 2) * (TW + 2);
    // bytes of one image row: f32 [Ci + 4], bf16 [max(Ci, 32) + 8]
    const size_t rowb = precision ? (size_t)((Ci < 32 ? 32 : Ci) + 8) * 2 : (size_t)(Ci + 4) * sizeof(float);
    // All nine weight slices up front for the narrow layers only.  Measured for the 64-channel layers as well (76 KB limit,
    // tools/phase_profile.py): the barrier-free tap loop is then at its f32 MFMA floor (3.8 us), but staging 36 KB of weights ahead
    // of the first MFMA costs more than the nine overlapped per-tap fetches did (13.6 vs 12.2 us per launch).
    a.allw = (halo + 9 * bn) * rowb <= 48 * 1024;
    size_t lds = (halo + (a.allw ? 9 : 2) * bn) * rowb;
    static const bool wide_ok = []() { const char* e = getenv("DPP_GEMM_WIDE_EPILOGUE"); return !(e && e[0] == '0'); }();
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    a.wide = wide_ok && al16(Y) && al16(residual) && al16(a.epi.bn_x) && al16(bias) && al16(a.epi.bn_mean) && al16(a.epi.bn_scale) &&
             al16(a.epi.bn_beta) && al16(a.epi.bn_inv_std);
    if ((store & (DPP_ST_C | DPP_ST_BNX)) && !a.wide) return DPP_E_UNSUPPORTED;      // bf16-stored tensors go through the 16-byte epilogue only
    size_t need = a.wide ? ((size_t)bm * (bn + 4) + 16 * bn) * sizeof(float) : (size_t)4 * bn * sizeof(float);
    if (lds < need) lds = need;
    if (lds > 160 * 1024) return DPP_E_UNSUPPORTED;
    dim3 grid(nblk, dpp_cdiv(Co, bn));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // The narrow square layers (16 / 32 channels, all nine weight slices resident, the 16-byte epilogue, 8 x 16 tiles that divide the maps) run on
    // conv3x3_p_kernel: DPP_C3_PERSIST workgroups (0: off) walk the tiles, the next halo in flight under the current tile.  The default is what a
    // chip holds at once -- five workgroups per CU at the kernel's 96 registers (conv3x3_kernel: 112 -> four; the generic round-6 form of the walk:
    // 152-164 -> three, which is why it gained so little).  tools/conv3_micro.py, us per launch, forward / data-gradient form, one-tile kernel -> walk:
    // 256 x 256 bf16 16 ch (4 096 tiles) 45.5 / 46.2 -> 27.0 / 25.9 (1 024 workgroups: 27.5 / 27.5, 1 536: 27.0 / 27.5); 128 x 128 float32 16 ch (1 024 tiles,
    // one each) 14.7 / 14.8 -> 12.3 / 12.6; 256 x 256 bf16 32 ch (1 024 tiles) 16.6 / 17.2 -> 13.8 / 15.7; 256 x 256 float32 16 ch 48.4 / 47.9 -> 39.2 / 37.9.
    const char* pe = getenv("DPP_C3_PERSIST");          // (read per call: the tests switch it)
    const int persist = pe ? atoi(pe) : 1024;
    static const int p_min = []() { const char* e = getenv("DPP_C3_P_MIN_TILES"); return e ? atoi(e) : 0; }();      // (experiments: the walk only above this many tiles)
    const bool p_always = nblk > p_min;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    // (64 channels on 8 x 16 tiles: bf16 operands only -- the 16 x 16 maps of the 256 x 256 net, 256 tiles.  DPP_C3_P_64W = the column tile, 0: off.
    // us per launch alone, forward / data gradient: 512 one-tile workgroups of 64 x 64 16.8 / 16.7, the walk with 64 columns (124 KB of LDS, one
    // workgroup per CU) 12.5 / 12.9, with 32 columns (70 KB, two per CU) 11.6 / 11.3; the bf16 256 x 256 step, same box: 6.83 / 6.75 -> 6.67 -> 6.57 ms)
    const char* g0e = getenv("DPP_C3_P_64W");
    const int bn64 = g0e ? atoi(g0e) : 32;
    const bool geom0 = (Ci == 16 || Ci == 32 || (Ci == 64 && precision && (bn64 == 64 || bn64 == 32))) && a.img == 1 && TH == 8 && TW == 16 && H % TH == 0 && W % TW == 0 &&
                       pow2(a.tiles_x) && pow2(a.tiles_y);
    // ... and the 64-channel layers of 8 x 8 maps (stages 3-4 at 128 x 128 input): two whole images per tile, 16 of the 64 output columns per
    // workgroup (blockIdx.y), 94 KB of LDS -- 256 workgroups, one per CU, instead of 512 that met at a barrier after every tap
    // bf16 operands only by default (52 KB per workgroup): the float32 form (94 KB, one workgroup per CU) is 12.3 / 12.1 us per launch against 12.8 / 12.4
    // alone, but the step is SLOWER with it (3.371 -> 3.400 ms, same box: a CU full of its LDS takes no gradient-branch workgroup beside it);
    // bf16 128 x 128: 2.998 -> 2.928 ms.  DPP_C3_P_64 = 0: off, 2: float32 as well.
    const char* g1e = getenv("DPP_C3_P_64");            // (read per call: the tests switch it)
    const int geom1_on = g1e ? atoi(g1e) : 1;
    const bool geom1 = (geom1_on >= 2 || (geom1_on == 1 && precision)) && Ci == 64 && H == 8 && W == 8 && a.img == 2 && TH == 8 && TW == 8;
    if (persist > 0 && a.wide && Ci == Co && bm == 128 && p_always && (geom0 || geom1) && (long)N * H * W * Ci < (1L << 31)) {
        // (GEOM 0: whole 16 / 32 column tiles, whatever the column split of the one-tile kernel would have been)
        const int pbn = geom1 ? 16 : (Ci == 64 ? bn64 : Co);
        const size_t pneed = ((size_t)bm * (pbn + 4) + 16 * pbn) * sizeof(float);
        const size_t region0 = (halo * rowb > pneed ? halo * rowb : pneed);
        const size_t ldsp = ((region0 + 15) & ~(size_t)15) + 9 * pbn * rowb + 8 * 64 * sizeof(float);      // halo / epilogue image | nine weight slices | per-channel vectors
        const long pwgs = (long)(persist < nblk ? persist : nblk) * (Co / pbn);
        // float32 images of a 32-channel layer are 68 KB: two workgroups per CU, taken where the launch is no more than that (128 x 128 input, stage 2:
        // 256 tiles -- one workgroup per CU, its tap loop barrier-free at the float32 MFMA rate); the 64-channel form: 94 KB, one per CU
        if (ldsp <= 64 * 1024 || (ldsp <= 80 * 1024 && pwgs <= 512) || (ldsp <= 160 * 1024 && pwgs <= 256)) {
            a.woff = (int)((region0 + 15) & ~(size_t)15);
            a.ntiles = nblk;
            if (geom0) { a.ltw = ilog2(a.tiles_x); a.lth = ilog2(a.tiles_y); }      // (the tile itself is compile-time 8 x 16 there: the fields carry the tile COUNTS' logs)
            const dim3 gp(persist < nblk ? persist : nblk, Co / pbn);
            if (getenv("DPP_C3_PERSIST_VERBOSE")) fprintf(stderr, "conv3x3_p_kernel: %d tiles on %d x %d workgroups, C %d, prec %d, store %d, %zu bytes of LDS\n", nblk, (int)gp.x, (int)gp.y, Ci, precision, store, ldsp);
#define DPP_C3P(CI_, BN_, P_, T_, E_, G_) do { \
            if (ldsp > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p_kernel<CI_, BN_, P_, T_, E_, G_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp); \
            DPP_LAUNCH((conv3x3_p_kernel<CI_, BN_, P_, T_, E_, G_>), gp, dim3(DPP_THREADS), ldsp, st, a); return dpp_launch_status(); } while (0)
#define DPP_C3PE(CI_, BN_, P_, T_, G_) do { if (store & (DPP_ST_C | DPP_ST_BNX)) DPP_C3P(CI_, BN_, P_, T_, true, G_); else DPP_C3P(CI_, BN_, P_, T_, false, G_); } while (0)
#define DPP_C3PX(CI_, BN_, G_) if (Ci == CI_) { \
            if (precision) { if (store & DPP_ST_A) DPP_C3PE(CI_, BN_, 1, dpp_bf16, G_); else DPP_C3PE(CI_, BN_, 1, float, G_); } \
            if (store & DPP_ST_A) DPP_C3PE(CI_, BN_, 0, dpp_bf16, G_); else DPP_C3PE(CI_, BN_, 0, float, G_); }
            DPP_C3PX(16, 16, 0) DPP_C3PX(32, 32, 0)
            if (geom1) { DPP_C3PX(64, 16, 1) }
            if (Ci == 64 && precision && pbn == 64) { if (store & DPP_ST_A) DPP_C3PE(64, 64, 1, dpp_bf16, 0); else DPP_C3PE(64, 64, 1, float, 0); }
            if (Ci == 64 && precision && pbn == 32) { if (store & DPP_ST_A) DPP_C3PE(64, 32, 1, dpp_bf16, 0); else DPP_C3PE(64, 32, 1, float, 0); }
#undef DPP_C3PX
#undef DPP_C3PE
#undef DPP_C3P
        }
    }
    // gfx950 has 160 KiB of LDS per CU; requests above the default 64 KiB window need the opt-in attribute
#define DPP_C3K(BM_, BN_, P_, T_, E_) do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<BM_, BN_, P_, T_, E_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        DPP_LAUNCH((conv3x3_kernel<BM_, BN_, P_, T_, E_>), grid, dim3(DPP_THREADS), lds, st, a); return dpp_launch_status(); } while (0)
#define DPP_C3E(BM_, BN_, P_, T_) do { if (store & (DPP_ST_C | DPP_ST_BNX)) DPP_C3K(BM_, BN_, P_, T_, true); else DPP_C3K(BM_, BN_, P_, T_, false); } while (0)
#define DPP_C3(BM_, BN_) if (bm == BM_ && bn == BN_) { \
        if (precision) { if (store & DPP_ST_A) DPP_C3E(BM_, BN_, 1, dpp_bf16); else DPP_C3E(BM_, BN_, 1, float); } \
        if (store & DPP_ST_A) DPP_C3E(BM_, BN_, 0, dpp_bf16); else DPP_C3E(BM_, BN_, 0, float); }
    DPP_C3(128, 64) DPP_C3(128, 32) DPP_C3(128, 16) DPP_C3(64, 64) DPP_C3(64, 32) DPP_C3(64, 16)
#undef DPP_C3
#undef DPP_C3E
#undef DPP_C3K
    return DPP_E_UNSUPPORTED;
}

extern "C" int dpp_conv3x3(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* Wk, int Co,
                           const float* bias, const float* residual, float* Y, int bm, const dpp_epilogue* epi, int store, dpp_stream_t stream) {
    return conv3x3_launch(X, N, H, W, Ci, act, Wk, Co, bias, residual, Y, bm, epi, 0, store, stream);
}

extern "C" int dpp_conv3x3_bf16(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* Wk, int Co,
                                const float* bias, const float* residual, float* Y, int bm, const dpp_epilogue* epi, int store, dpp_stream_t stream) {
    return conv3x3_launch(X, N, H, W, Ci, act, Wk, Co, bias, residual, Y, bm, epi, 1, store, stream);
}

extern "C" int dpp_conv3x3_wtrans(const float* Wk, int Co, int Ci, float* Wd, dpp_stream_t stream) {
    if (!Wk || !Wd || Co < 1 || Ci < 1) return DPP_E_BADARG;
    int n = Co * 9 * Ci;
    DPP_LAUNCH(conv3x3_wtrans_kernel, dim3(dpp_cdiv(n, DPP_THREADS)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), Wk, Co, Ci, Wd);
    return dpp_launch_status();
}

extern "C" size_t dpp_wtrans_job_bytes(void) { return sizeof(WtransJob); }

extern "C" int dpp_conv3x3_wtrans_multi(const void* jobs_dev, int njobs, int total_blocks, dpp_stream_t stream) {
    if (!jobs_dev || njobs < 1 || total_blocks < 1) return DPP_E_BADARG;
    DPP_LAUNCH(conv3x3_wtrans_multi_kernel, dim3(total_blocks), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
               static_cast<const WtransJob*>(jobs_dev), njobs);
    return dpp_launch_status();
}

// Geometry of the filter-gradient launch: `nblk` partial slices [Co][9][Ci] (one per blockIdx.x, each workgroup walks tiles
// blockIdx.x, +nblk, ...) and `taps_pb` taps per blockIdx.y.  The slice count is bounded by the BYTES the slices cost: the
// 64-channel layers of a batch-128 step wrote 128 slices of 147 KB each (189 MB per step for ten layers, re-read by
// dpp_reduce_multi at the end of the data-gradient chain); 6 MB of slices per layer is 32 of them, and the taps are dealt to
// more workgroups instead so that >= 256 workgroups stay in flight.
static void wgrad_geometry(int N, int H, int W, int Ci, int Co, int bm, int& nblk, int& taps_pb) {
    int lth, ltw, img;
    pick_tile(N, H, W, bm, lth, ltw, img);
    const int ntiles = dpp_cdiv(W, 1 << ltw) * dpp_cdiv(H, 1 << lth) * dpp_cdiv(N, img);
    const int NT = (Co >> 4) * (Ci >> 4);
    taps_pb = NT <= 4 ? 9 : (NT <= 16 ? 3 : 1);        // (tap, tile) pairs per workgroup <= 4 waves * MAXACC accumulators
    int cap = WGRAD_MAX_BLOCKS;
    while (cap > 32 && (long)cap * Co * 9 * Ci * 4 > 6L * 1024 * 1024) cap >>= 1;
    // 64 channels on the transposed-image kernel (maps >= 12 wide: the 16-wide maps of the 256 x 256 net): four workgroups per slice (its
    // output channels over blockIdx.y), so 64 slices fill the chip; their 9.4 MB of partials are a tenth of what the layer reads
    if (Ci == 64 && Co == 64 && dpp_conv3x3_wgrad_t_ok(N, H, W, Ci, Co, nullptr) && cap < 64) cap = 64;
    // 32 channels with >= 1 024 tiles (the 256 x 256 net): 256 slices, so that a workgroup takes all nine taps in ONE pass over its tiles
    // instead of three workgroups staging every tile for three taps each
    if (Ci == 32 && Co == 32 && ntiles >= 1024 && dpp_conv3x3_wgrad_t_ok(N, H, W, Ci, Co, nullptr) && cap < 256) cap = 256;
    nblk = ntiles < cap ? ntiles : cap;
    while (taps_pb > 1 && nblk * (9 / taps_pb) < 256) taps_pb = taps_pb == 9 ? 3 : 1;
}

extern "C" int dpp_conv3x3_wgrad_blocks(int N, int H, int W, int Ci, int Co, int bm) {
    if ((bm != 64 && bm != 128) || Ci < 16 || Co < 16) return -1;
    int nblk, taps_pb;
    wgrad_geometry(N, H, W, Ci, Co, bm, nblk, taps_pb);
    return nblk;
}

static int conv3x3_wgrad_launch(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, int Co,
                                float* partial, int bm, int store, int precision, dpp_stream_t stream);

extern "C" int dpp_conv3x3_wgrad(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, int Co,
                                 float* partial, int bm, int store, dpp_stream_t stream) {
    return conv3x3_wgrad_launch(X, N, H, W, Ci, act, dY, Co, partial, bm, store, 0, stream);
}

extern "C" int dpp_conv3x3_wgrad_bf16_ok(int N, int H, int W, int Ci, int Co) { return dpp_conv3x3_wgrad_t_ok(N, H, W, Ci, Co, nullptr) ? 1 : 0; }

extern "C" int dpp_conv3x3_wgrad_bf16(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, int Co,
                                      float* partial, int bm, int store, dpp_stream_t stream) {
    return conv3x3_wgrad_launch(X, N, H, W, Ci, act, dY, Co, partial, bm, store, 1, stream);
}

static int conv3x3_wgrad_launch(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, int Co,
                                float* partial, int bm, int store, int precision, dpp_stream_t stream) {
    if (!X || !dY || !partial || N < 1 || Ci < 16 || (Ci & 15) || Co < 16 || (Co & 15) || (bm != 64 && bm != 128) || (store & ~(DPP_ST_A | DPP_ST_B))) return DPP_E_BADARG;
    const int y16 = (store & DPP_ST_B) ? 1 : 0;
    Wgrad3Args a;
    a.X = X; a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.dY = dY; a.partial = partial;
    if (act && (act->mode & 4)) return DPP_E_UNSUPPORTED;       // the two-tensor BatchNorm-backward operand is a dpp_gemm feature
    if (act) a.act = *act; else { a.act.mode = 0; a.act.cmod = 1; a.act.mean = a.act.scale = a.act.beta = nullptr; }
    pick_tile(N, H, W, bm, a.lth, a.ltw, a.img);
    int TH = 1 << a.lth, TW = 1 << a.ltw;
    a.tiles_x = dpp_cdiv(W, TW); a.tiles_y = dpp_cdiv(H, TH);
    a.ntiles = a.tiles_x * a.tiles_y * dpp_cdiv(N, a.img);
    int nblk, taps_pb;
    wgrad_geometry(N, H, W, Ci, Co, bm, nblk, taps_pb);
    // the 16- / 32-channel layers: channel-major LDS images, 16-byte operand reads (conv3x3_wgrad_t.hip, round 6)
    if (dpp_conv3x3_wgrad_t_ok(N, H, W, Ci, Co, act))
        return dpp_conv3x3_wgrad_t_launch(X, N, H, W, Ci, act, dY, partial, nblk, taps_pb, store, precision, static_cast<hipStream_t>(stream));
    if (precision) return DPP_E_UNSUPPORTED;          // bf16 MFMA operands: the transposed-image kernel only
    int NT = (Co >> 4) * (Ci >> 4);
    int pairs = taps_pb * NT;
    int maxacc = dpp_cdiv(pairs, 4);
    a.taps_pb = taps_pb;
    a.sg.m_hw2 = c3_magic((TH + 2) * (TW + 2));
    a.sg.m_tw2 = c3_magic(TW + 2);
    a.sg.lqp = ilog2(Ci / 4);
    a.lqy = ilog2(Co / 4);
    size_t lds = ((size_t)a.img * (TH + 2) * (TW + 2) * (Ci + 4) + (size_t)bm * (Co + 4)) * sizeof(float);
    if (lds > 160 * 1024) return DPP_E_UNSUPPORTED;
    dim3 grid(nblk, 9 / taps_pb);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_W3K(BM_, MA_, T_, Y_) do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<BM_, MA_, T_, Y_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        DPP_LAUNCH((conv3x3_wgrad_kernel<BM_, MA_, T_, Y_>), grid, dim3(DPP_THREADS), lds, st, a); return dpp_launch_status(); } while (0)
#define DPP_W3(BM_, MA_) if (bm == BM_ && maxacc <= MA_) { \
        if (store & DPP_ST_A) { if (y16) DPP_W3K(BM_, MA_, dpp_bf16, dpp_bf16); else DPP_W3K(BM_, MA_, dpp_bf16, float); } \
        else { if (y16) DPP_W3K(BM_, MA_, float, dpp_bf16); else DPP_W3K(BM_, MA_, float, float); } }
    DPP_W3(128, 3) DPP_W3(128, 4) DPP_W3(128, 9) DPP_W3(128, 12) DPP_W3(128, 16)
    DPP_W3(64, 3) DPP_W3(64, 4) DPP_W3(64, 9) DPP_W3(64, 12) DPP_W3(64, 16)
#undef DPP_W3
    return DPP_E_UNSUPPORTED;
}
