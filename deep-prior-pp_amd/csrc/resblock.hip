// resblock.hip -- a whole pre-activation bottleneck block of the deterministic (test-time) forward pass as ONE kernel for gfx950.
//
// Reference arithmetic: res_block, /root/reference/src/net/resnet.py:349-414, with every BatchNormLayer in deterministic mode
// (/root/reference/src/net/batchnormlayer.py:158-159: the stored running mean / inv_std, a per-channel affine):
//     h  = relu(bn0(x))                    [P][Cin]
//     c1 = conv1x1(h) + b1                 [P'][NB]      (stride s in a projection block)
//     c2 = conv3x3(relu(bn1(c1))) + b2     [P'][NB]      'half' padding, zeros AFTER the activation
//     c3 = conv1x1(relu(bn2(c2))) + b3     [P'][Cout]
//     out = x + c3                         (identity block, Cin == Cout, s = 1)
//     out = c3 + conv1x1_s(h) + bsc        (projection block)
// which netbase.py:257-310 (`computeOutput`) runs for every test batch.  In training mode the batch statistics of bn1 / bn2 make a
// grid-wide dependency between the three convolutions; in deterministic mode nothing does, so the
// 16- / 32- / 64-channel intermediates never leave the CU: 20 launches per forward pass instead of ~120, and the block's input
// and output are the only tensors that touch HBM.
//
// A workgroup (4 waves) owns a TH x TW tile of output pixels of one image:
//   phase A   the activated input halo (TH+2) x (TW+2) x Cin is staged through LDS in K-chunks of 64 channels (bn0 + ReLU applied on
//             the way) and multiplied with W1 on v_mfma_f32_16x16x4_f32: c1 for the tile AND its one-pixel border (the 3x3 needs
//             it; recomputed by the neighbouring tiles -- 1.4-1.9x of the cheapest of the three products); bias + bn1 + ReLU, zero
//             outside the image, into the LDS image A1 [halo][NB]
//   phase B   the nine taps read shifted rows of A1 (as conv3x3.hip does), bias + bn2 + ReLU into A2 [tile][NB]
//   phase C   A2 . W3 in passes of 64 output channels (+ in a projection block the shortcut's product on the centre pixels of the
//             activated input, which then stays whole in LDS), through an LDS image of the pass so that bias, residual and the
//             store are 16-byte accesses.
// The weight slices (W1 K-chunks, W2 taps, W3 / Wsc column passes) form ONE stream through two LDS buffers: the next slice is
// fetched into registers before the MFMAs of the current one and committed after them, one barrier per slice.
// f32 in, f32 MFMA, f32 accumulate (the 1e-3 mm path).  The summation order of a pixel does not depend on the batch size or on
// where the pixel's tile lies, so a frame's joints do not depend on the batch it is evaluated in (tests/test_full_size.py).
#include <stdlib.h>
#include "dpp_common.h"

namespace {

struct RBArgs {
    dpp_resblock_desc d;
    int lth, ltw, tiles_x, tiles_y, ntiles;
    unsigned m_tw2;          // floor(2^32 / (TW+2)) + 1
    int xcd_chunk;           // ntiles / 8 when the tile -> XCD swizzle applies, else 0
    unsigned long long* prof;   // phase stamps (profiling build only, see dpp_stamp)
};

// Raw loads of four consecutive elements of a float32 or bf16-stored tensor (16 / 8 bytes): kept UNCONVERTED in registers and widened
// where they are consumed -- a conversion at the load would make the wave wait for the data right there.
template <class T> struct RBRaw;
template <> struct RBRaw<float> {
    typedef float4 type;
    __device__ static __forceinline__ type ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ float4 widen(const type& r) { return r; }
    __device__ static __forceinline__ type zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct RBRaw<dpp_bf16> {
    typedef uint2 type;
    __device__ static __forceinline__ type ld(const dpp_bf16* p) { return *reinterpret_cast<const uint2*>(p); }
    __device__ static __forceinline__ float4 widen(const type& r) {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    }
    __device__ static __forceinline__ type zero() { type z; z.x = 0u; z.y = 0u; return z; }
};
// bf16 mode (Q): a value as a bf16 MFMA operand / a bf16-stored tensor element sees it (round to nearest even, back in float32)
template <bool Q> __device__ __forceinline__ float rb_q(float v) { return Q ? dpp_bf16_round(v) : v; }
template <bool Q> __device__ __forceinline__ float4 rb_q4(const float4& v) {
    return Q ? make_float4(dpp_bf16_round(v.x), dpp_bf16_round(v.y), dpp_bf16_round(v.z), dpp_bf16_round(v.w)) : v;
}

__device__ __forceinline__ float4 rb_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void rb_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// v = relu((x - mean) * (gamma * inv_std) + beta), the arithmetic of dpp_act1 with dpp_bn_eval_coeffs' scale
__device__ __forceinline__ float rb_bnrelu(float x, float mu, float sc, float be) { return fmaxf((x - mu) * sc + be, 0.0f); }

// QW: the weight fragments are rounded to bfloat16 first (bf16 mode: the operand of a bf16 MFMA product; the activations were rounded
// when they were written to LDS)
template <int RT, int CT, bool QW = false>
__device__ __forceinline__ void rb_mfma16(f32x4 (&acc)[RT][CT], const float4 (&av)[RT], const float4 (&bw)[CT]) {
    float4 bv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bv[ct] = rb_q4<QW>(bw[ct]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), dpp_f4_get(bv[ct], t), acc[rt][ct], 0, 0, 0);
}

// NB: bottleneck width (16 / 32 / 64); the block's output has 4 NB channels, its input 4 NB (identity block) or 2 NB (PROJ: the projection
// blocks that open a stage, strided input + shortcut convolution).  BM = TH * TW output pixels per workgroup; RT1: 16-row tiles of the
// halo per wave (halo padded to RT1 * 64 rows).
//
// Round-5 second version.  The first one streamed the weight slices through two LDS buffers with one barrier per slice (K chunk / tap /
// column pass): a slice is 0.4-0.9 us of MFMA work per wave but a 1.5-2 us global -> register -> LDS round trip, so the kernel ran at the
// round-trip rate (30.8 us per stage-3/4 block against 9 us of MFMA time; profiles/r05_forward_kernels_fused_v1.txt).  Now
//   * the whole activated input halo goes to LDS in ONE round trip (all loads of the workgroup in flight at once), one barrier;
//   * weights never touch LDS: every wave loads its B fragments straight from global memory / L2 in MFMA fragment order (the kernel
//     layouts [out][k] are K-contiguous: lane (l15 = column, kq) reads 16 bytes of row `column` at k = 16 step + 4 kq) through a register
//     ring several k-steps deep, requested before the barrier that precedes their phase -- no barrier inside a phase;
//   * the output image of a column pass is private to the wave (wave-level ordering only).
// Three workgroup barriers per block: input halo, A1, A2.
//
// bf16 mode (round 6; XT / YT = the storage of the block's input / output, Q = YT is bfloat16: BASELINE config 5's deterministic forward,
// which fell back to ~6 launches per block): the block computes what the layer-by-layer bf16 path computes -- every tensor that path
// MATERIALISES is rounded to bfloat16 where it would have been stored (c1, c2, the shortcut's output, the block's output), every operand
// of a product that path runs on bf16 MFMA is rounded as its kernel rounds it (the activated operands after their prologue, the filters)
// -- with float32 MFMA on the rounded values (bf16 x bf16 products
// are exact in float32: the same sums in another order of additions).
template <int NB, int BM, int RT1, bool PROJ, class XT = float, class YT = float>
__global__ __launch_bounds__(DPP_THREADS) void resblock_eval_kernel(RBArgs a) {
    constexpr bool Q = !std::is_same<YT, float>::value;
    constexpr bool Q3 = Q;                             // the exit convolution (K = NB) too (K = 16: a 32-deep step with a zero upper half)
    dpp_kernarg_warm<sizeof(RBArgs)>();
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int CIN = PROJ ? 2 * NB : 4 * NB, COUT = 4 * NB;
    constexpr int CT1 = NB / 16;                     // column tiles of the bottleneck width
    constexpr int HPP = RT1 * 64;                    // padded halo rows
    constexpr int LDA = CIN + 4;                     // row stride of the activated input halo
    constexpr int LD1 = NB + 4;                      // row stride of A1 / A2
    constexpr int RTB = BM / 16;                     // row tiles of the output tile
    // phases B / C: the waves split the COLUMNS first (a B fragment comes from global memory and feeds RM2 row tiles; an A fragment is an
    // LDS read): 1 x 4 waves (64 channels), 2 x 2 (32), 4 x 1 (16) -- two row tiles and one column tile per wave in phase B everywhere
    constexpr int WN2 = CT1 < 4 ? CT1 : 4, WM2 = 4 / WN2, RM2 = RTB / WM2;
    constexpr int CN2 = CT1 / WN2;                   // phase B: column tiles per wave
    constexpr int CN3 = 4 / WN2;                     // phase C: column tiles per wave of a 64-column pass
    constexpr int QR = CIN / 4;                      // channel quads per halo row
    constexpr int RS = DPP_THREADS / QR;             // halo rows per staging sweep
    constexpr int ASLOTS = HPP / RS;                 // halo float4 slots per thread
    constexpr int KSA = CIN / 16, RDA = KSA < 8 ? KSA : 8;             // phase A: k-steps, B-fragment ring depth
    constexpr int KSB = 9 * (NB / 16), RDB = KSB < 12 ? KSB : 12;      // phase B
    constexpr int KSC = NB / 16;                                       // phase C: k-steps of a pass
    constexpr int NPASS = COUT / 64;
    constexpr int WROWS = RM2 * 16, WCOLS = CN3 * 16, LDI = WCOLS + 4; // a wave's share of a pass and its private image
    constexpr int QW = WCOLS / 4, RSW = 64 / QW, SWEEPS = WROWS / RSW; // the wave's 16-byte store sweep
    static_assert(CN2 >= 1 && RM2 >= 1 && HPP % RS == 0 && WROWS % RSW == 0, "tile shape");
    const dpp_resblock_desc& d = a.d;
    const int TH = 1 << a.lth, TW = 1 << a.ltw, TW2 = TW + 2, HP = (TH + 2) * TW2;
    const int S = d.stride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wm = wave % WM2, wn = wave / WM2;

    // LDS: A0 [HPP][LDA] | A2 [BM][LD1] | (PROJ: A1 [HPP][LD1] | images) ; identity: A1 and the images alias A0 (dead after phase A)
    float* A0 = smem;
    float* A2 = A0 + HPP * LDA;
    float* A1 = PROJ ? A2 + BM * LD1 : A0;
    float* IMG = (PROJ ? A1 + HPP * LD1 : A0 + HPP * LD1) + wave * (WROWS * LDI);

    int bid = blockIdx.x;
    if (a.xcd_chunk) bid = (bid & 7) * a.xcd_chunk + (bid >> 3);      // neighbouring tiles (shared halo rows) on one XCD's L2
    const int bx = bid % a.tiles_x;
    const int tq = bid / a.tiles_x;
    const int by = tq % a.tiles_y, n = tq / a.tiles_y;
    const int y0 = by << a.lth, x0 = bx << a.ltw;
    const int Ho = d.Ho, Wo = d.Wo;
    const XT* Xn = reinterpret_cast<const XT*>(d.X) + (size_t)n * d.H * d.W * CIN;

    dpp_stamp(a.prof, 0);
    // ---- the activated input halo: slot u of a thread is halo position hp = tid / QR + u * RS, channel quad qa.  Every load is
    //      UNCONDITIONAL (a pixel outside the image reads the clamped one and is zeroed afterwards): a branch between a load and its use
    //      makes the compiler drain all outstanding loads, and these are the longest pole of the kernel -- requested first ----
    const int qa = (tid % QR) * 4;
    typename RBRaw<XT>::type areg[ASLOTS];
    bool ain[ASLOTS];
#pragma unroll
    for (int u = 0; u < ASLOTS; ++u) {
        const int hp = tid / QR + u * RS;
        const int hy = (int)__umulhi((unsigned)hp, a.m_tw2), hx = hp - hy * TW2;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        ain[u] = hp < HP && y >= 0 && y < Ho && x >= 0 && x < Wo;
        const int yc = y < 0 ? 0 : (y >= Ho ? Ho - 1 : y), xc = x < 0 ? 0 : (x >= Wo ? Wo - 1 : x);
        areg[u] = RBRaw<XT>::ld(Xn + ((size_t)(yc * S) * d.W + xc * S) * CIN + qa);
    }
    const float4 mu = rb_ld4(d.bn0.mean + qa), g0 = rb_ld4(d.bn0.gamma + qa), i0 = rb_ld4(d.bn0.inv_std + qa), be = rb_ld4(d.bn0.beta + qa);
    // ---- phase A's first B fragments (in flight while the halo is committed) ----
    const float* w1p[CT1];
#pragma unroll
    for (int ct = 0; ct < CT1; ++ct) w1p[ct] = d.W1 + (size_t)(ct * 16 + l15) * CIN + kq * 4;
    float4 bA[RDA][CT1];
#pragma unroll
    for (int s = 0; s < RDA; ++s)
#pragma unroll
        for (int ct = 0; ct < CT1; ++ct) bA[s][ct] = rb_ld4(w1p[ct] + s * 16);
    DPP_SCHED_FENCE();
    {
        const float4 sc = make_float4(g0.x * i0.x, g0.y * i0.y, g0.z * i0.z, g0.w * i0.w);
#pragma unroll
        for (int u = 0; u < ASLOTS; ++u) {
            const int hp = tid / QR + u * RS;
            const float4 v = RBRaw<XT>::widen(areg[u]);
            float4 t = rb_q4<Q>(make_float4(rb_bnrelu(v.x, mu.x, sc.x, be.x), rb_bnrelu(v.y, mu.y, sc.y, be.y),
                                            rb_bnrelu(v.z, mu.z, sc.z, be.z), rb_bnrelu(v.w, mu.w, sc.w, be.w)));
            if (!ain[u]) t = make_float4(0.f, 0.f, 0.f, 0.f);       // zero padding is applied AFTER the activation
            rb_st4(&A0[hp * LDA + qa], t);
        }
    }
    dpp_stamp(a.prof, 1);
    // per-lane column constants of the first epilogue
    float e1b[CT1], e1m[CT1], e1s[CT1], e1t[CT1];
#pragma unroll
    for (int ct = 0; ct < CT1; ++ct) {
        const int c = ct * 16 + l15;
        e1b[ct] = d.b1[c]; e1m[ct] = d.bn1.mean[c]; e1s[ct] = d.bn1.gamma[c] * d.bn1.inv_std[c]; e1t[ct] = d.bn1.beta[c];
    }
    __syncthreads();
    dpp_stamp(a.prof, 2);

    // ================================ phase A: c1 over the halo ================================
    f32x4 acc1[RT1][CT1];
#pragma unroll
    for (int i = 0; i < RT1; ++i)
#pragma unroll
        for (int j = 0; j < CT1; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        // (the A fragment of step s + 1 is read from LDS before the products of step s: its latency hides under 16 MFMAs)
        float4 av[2][RT1];
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt) av[0][rt] = rb_ld4(&A0[((wave * RT1 + rt) * 16 + l15) * LDA + kq * 4]);
#pragma unroll
        for (int s = 0; s < KSA; ++s) {
            if (s + 1 < KSA) {
#pragma unroll
                for (int rt = 0; rt < RT1; ++rt) av[(s + 1) & 1][rt] = rb_ld4(&A0[((wave * RT1 + rt) * 16 + l15) * LDA + (s + 1) * 16 + kq * 4]);
            }
            rb_mfma16<RT1, CT1, Q>(acc1, av[s & 1], bA[s % RDA]);
            if (s + RDA < KSA) {
#pragma unroll
                for (int ct = 0; ct < CT1; ++ct) bA[s % RDA][ct] = rb_ld4(w1p[ct] + (s + RDA) * 16);
            }
            DPP_SCHED_FENCE();
        }
    }
    dpp_stamp(a.prof, 3);
    // phase B's first B fragments (W2 [NB][9][NB]: the (tap, channel) pairs of a row are contiguous, step s is offset 16 s)
    const float* w2p[CN2];
#pragma unroll
    for (int ct = 0; ct < CN2; ++ct) w2p[ct] = d.W2 + (size_t)((wn * CN2 + ct) * 16 + l15) * 9 * NB + kq * 4;
    float4 bB[RDB][CN2];
#pragma unroll
    for (int s = 0; s < RDB; ++s)
#pragma unroll
        for (int ct = 0; ct < CN2; ++ct) bB[s][ct] = rb_ld4(w2p[ct] + s * 16);
    float e2b[CN2], e2m[CN2], e2s[CN2], e2t[CN2];
#pragma unroll
    for (int ct = 0; ct < CN2; ++ct) {
        const int c = (wn * CN2 + ct) * 16 + l15;
        e2b[ct] = d.b2[c]; e2m[ct] = d.bn2.mean[c]; e2s[ct] = d.bn2.gamma[c] * d.bn2.inv_std[c]; e2t[ct] = d.bn2.beta[c];
    }
    if (!PROJ) __syncthreads();                         // identity: A1 overwrites the halo, which every wave has to be done reading
    // c1 + b1 -> bn1 -> ReLU -> A1 (zero outside the image: the 3x3 pads its ACTIVATED input)
#pragma unroll
    for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hp = (wave * RT1 + rt) * 16 + kq * 4 + r;
            const int hy = (int)__umulhi((unsigned)hp, a.m_tw2), hx = hp - hy * TW2;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            const bool in = hp < HP && y >= 0 && y < Ho && x >= 0 && x < Wo;
#pragma unroll
            for (int ct = 0; ct < CT1; ++ct) {
                // (bf16 mode: c1 as stored, then the 3x3's operand as its kernel rounds it)
                const float v = rb_q<Q>(rb_bnrelu(rb_q<Q>(acc1[rt][ct][r] + e1b[ct]), e1m[ct], e1s[ct], e1t[ct]));
                A1[hp * LD1 + ct * 16 + l15] = in ? v : 0.0f;
            }
        }
    __syncthreads();

    // ================================ phase B: the 3x3 over A1 ================================
    int hbase[RM2];
#pragma unroll
    for (int rt = 0; rt < RM2; ++rt) {
        const int row = (wm * RM2 + rt) * 16 + l15;
        hbase[rt] = ((row >> a.ltw) + 1) * TW2 + (row & (TW - 1)) + 1;
    }
    f32x4 acc2[RM2][CN2];
#pragma unroll
    for (int i = 0; i < RM2; ++i)
#pragma unroll
        for (int j = 0; j < CN2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dpp_stamp(a.prof, 4);
    {
        constexpr int KPT = NB / 16;                    // k-steps per tap
        auto afrag = [&](int s, float4 (&av)[RM2]) {
            const int tap = s / KPT, kk = s % KPT;
            const int toff = (tap / 3 - 1) * TW2 + (tap % 3 - 1);
#pragma unroll
            for (int rt = 0; rt < RM2; ++rt) av[rt] = rb_ld4(&A1[(hbase[rt] + toff) * LD1 + kk * 16 + kq * 4]);
        };
        float4 av[2][RM2];
        afrag(0, av[0]);
#pragma unroll
        for (int s = 0; s < KSB; ++s) {
            if (s + 1 < KSB) afrag(s + 1, av[(s + 1) & 1]);
            rb_mfma16<RM2, CN2, Q>(acc2, av[s & 1], bB[s % RDB]);
            if (s + RDB < KSB) {
#pragma unroll
                for (int ct = 0; ct < CN2; ++ct) bB[s % RDB][ct] = rb_ld4(w2p[ct] + (s + RDB) * 16);
            }
            DPP_SCHED_FENCE();
        }
    }
    dpp_stamp(a.prof, 5);
    // phase C: the sweep geometry of this wave's stores, and EVERYTHING the passes need from memory -- W3 fragments, bias, residual rows
    // (identity block) -- requested here, before the A2 barrier: a pass is 0.4 us of products, a round trip 1-2 us
    const int cq = lane % QW, rq = lane / QW;           // column quad / first row of this lane in the wave's sweep
    const int colw = wn * WCOLS + cq * 4;               // its first column inside a 64-column pass
    int ooff[SWEEPS];                                   // pixel index of the row in Y (and, identity block, in X), -1 outside
#pragma unroll
    for (int it = 0; it < SWEEPS; ++it) {
        const int row = wm * WROWS + rq + it * RSW;
        const int y = y0 + (row >> a.ltw), x = x0 + (row & (TW - 1));
        ooff[it] = (y < Ho && x < Wo) ? ((n * Ho + y) * Wo + x) : -1;
    }
    const float* w3p[CN3];
#pragma unroll
    for (int ct = 0; ct < CN3; ++ct) w3p[ct] = d.W3 + (size_t)((wn * CN3 + ct) * 16 + l15) * NB + kq * 4;
    float4 bC[NPASS][KSC][CN3], bq[NPASS], bs[NPASS];
    typename RBRaw<XT>::type res[NPASS][SWEEPS];
    const XT* const Xr = reinterpret_cast<const XT*>(d.X);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
        for (int sk = 0; sk < KSC; ++sk)
#pragma unroll
            for (int ct = 0; ct < CN3; ++ct) bC[pass][sk][ct] = rb_ld4(w3p[ct] + (size_t)pass * 64 * NB + sk * 16);
        bq[pass] = rb_ld4(d.b3 + pass * 64 + colw);
#pragma unroll
        for (int it = 0; it < SWEEPS; ++it) {
            // (unconditional: a row outside the image reads pixel 0 of the tensor and is never stored)
            res[pass][it] = PROJ ? RBRaw<XT>::zero() : RBRaw<XT>::ld(Xr + (size_t)(ooff[it] >= 0 ? ooff[it] : 0) * CIN + pass * 64 + colw);
        }
        bs[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PROJ) {
            const float4 t = rb_ld4(d.bsc + pass * 64 + colw);
            if (Q) bs[pass] = t;                             // bf16 mode: the shortcut's output is a stored tensor of its own (rounded first)
            else { bq[pass].x += t.x; bq[pass].y += t.y; bq[pass].z += t.z; bq[pass].w += t.w; }
        }
    }
    DPP_SCHED_FENCE();
#pragma unroll
    for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (wm * RM2 + rt) * 16 + kq * 4 + r;
#pragma unroll
            for (int ct = 0; ct < CN2; ++ct)
                A2[row * LD1 + (wn * CN2 + ct) * 16 + l15] = rb_q<Q3>(rb_bnrelu(rb_q<Q>(acc2[rt][ct][r] + e2b[ct]), e2m[ct], e2s[ct], e2t[ct]));
        }
    __syncthreads();
    dpp_stamp(a.prof, 6);

    // ================================ phase C: c3 (+ shortcut) in passes of 64 output channels ================================
    int hc[RM2];                                        // (projection) halo row of this lane's A rows, centre tap
#pragma unroll
    for (int rt = 0; rt < RM2; ++rt) {
        const int row = (wm * RM2 + rt) * 16 + l15;
        hc[rt] = ((row >> a.ltw) + 1) * TW2 + (row & (TW - 1)) + 1;
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const float* wsp[CN3];
        float4 bS[RDA][CN3];
        if (PROJ) {
#pragma unroll
            for (int ct = 0; ct < CN3; ++ct) wsp[ct] = d.Wsc + (size_t)(pass * 64 + (wn * CN3 + ct) * 16 + l15) * CIN + kq * 4;
#pragma unroll
            for (int s = 0; s < RDA; ++s)
#pragma unroll
                for (int ct = 0; ct < CN3; ++ct) bS[s][ct] = rb_ld4(wsp[ct] + s * 16);
        }
        f32x4 acc3[RM2][CN3], accS[RM2][CN3];
#pragma unroll
        for (int i = 0; i < RM2; ++i)
#pragma unroll
            for (int j = 0; j < CN3; ++j) { acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accS[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < KSC; ++s) {
            float4 av[RM2];
#pragma unroll
            for (int rt = 0; rt < RM2; ++rt) av[rt] = rb_ld4(&A2[((wm * RM2 + rt) * 16 + l15) * LD1 + s * 16 + kq * 4]);
            rb_mfma16<RM2, CN3, Q3>(acc3, av, bC[pass][s]);
        }
        if (PROJ) {
            // the shortcut's product on the centre pixels of the activated input (still whole in LDS); bf16 mode: its own accumulators
#pragma unroll
            for (int s = 0; s < KSA; ++s) {
                float4 av[RM2];
#pragma unroll
                for (int rt = 0; rt < RM2; ++rt) av[rt] = rb_ld4(&A0[hc[rt] * LDA + s * 16 + kq * 4]);
                if (Q) rb_mfma16<RM2, CN3, true>(accS, av, bS[s % RDA]);
                else rb_mfma16<RM2, CN3, false>(acc3, av, bS[s % RDA]);
                if (s + RDA < KSA) {
#pragma unroll
                    for (int ct = 0; ct < CN3; ++ct) bS[s % RDA][ct] = rb_ld4(wsp[ct] + (s + RDA) * 16);
                }
                DPP_SCHED_FENCE();
            }
        }
        // the wave's [WROWS][WCOLS] share through its private image: 16-byte bias / residual / store
        float4 scv[SWEEPS];
        if (PROJ && Q) {
            // bf16 mode: the shortcut's output as the tensor the layer-by-layer path stores (rounded), one more trip through the image
            DPP_WAVE_SYNC();
#pragma unroll
            for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
                for (int ct = 0; ct < CN3; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) IMG[(rt * 16 + kq * 4 + r) * LDI + ct * 16 + l15] = accS[rt][ct][r];
            DPP_WAVE_SYNC();
#pragma unroll
            for (int it = 0; it < SWEEPS; ++it) {
                const float4 v = rb_ld4(&IMG[(rq + it * RSW) * LDI + cq * 4]);
                scv[it] = rb_q4<true>(make_float4(v.x + bs[pass].x, v.y + bs[pass].y, v.z + bs[pass].z, v.w + bs[pass].w));
            }
        }
        DPP_WAVE_SYNC();                                // (the previous reads of the image are done)
#pragma unroll
        for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CN3; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) IMG[(rt * 16 + kq * 4 + r) * LDI + ct * 16 + l15] = acc3[rt][ct][r];
        DPP_WAVE_SYNC();
#pragma unroll
        for (int it = 0; it < SWEEPS; ++it) {
            if (ooff[it] >= 0) {
                const float4 v = rb_ld4(&IMG[(rq + it * RSW) * LDI + cq * 4]);
                const float4 bb = bq[pass];
                const float4 rr = (PROJ && Q) ? scv[it] : RBRaw<XT>::widen(res[pass][it]);
                dpp_st4(reinterpret_cast<YT*>(d.Y) + (size_t)ooff[it] * COUT + pass * 64 + colw,
                        make_float4(v.x + bb.x + rr.x, v.y + bb.y + rr.y, v.z + bb.z + rr.z, v.w + bb.w + rr.w));
            }
        }
    }
    dpp_stamp(a.prof, 7);
}

int rb_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

bool rb_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dpp_resblock_eval_ok(int Cin, int Cout, int Nb, int stride, int projection) {
    if (Nb != 16 && Nb != 32 && Nb != 64) return 0;
    if (Cout != 4 * Nb) return 0;
    if (!projection) return stride == 1 && Cin == Cout;                    // identity block
    return (stride == 1 || stride == 2) && Cin == 2 * Nb;                  // the block that opens a stage
}

// argument checks, tile geometry and LDS size of a block (shared by the launch and by dpp_resblock_eval_check)
static int rb_prepare(const dpp_resblock_desc* desc, RBArgs& a, size_t& lds, bool& proj) {
    if (!desc) return DPP_E_BADARG;
    const dpp_resblock_desc& d = *desc;
    if (!d.X || !d.Y || !d.W1 || !d.b1 || !d.W2 || !d.b2 || !d.W3 || !d.b3 || d.N < 1 || d.H < 1 || d.W < 1) return DPP_E_BADARG;
    const dpp_bn_eval* bns[3] = {&d.bn0, &d.bn1, &d.bn2};
    for (const dpp_bn_eval* b : bns)
        if (!b->mean || !b->inv_std || !b->gamma || !b->beta) return DPP_E_BADARG;
    proj = d.Wsc != nullptr;
    if (proj && !d.bsc) return DPP_E_BADARG;
    if (!dpp_resblock_eval_ok(d.Cin, d.Cout, d.Nb, d.stride, proj ? 1 : 0)) return DPP_E_UNSUPPORTED;
    if (d.Ho != (d.H + d.stride - 1) / d.stride || d.Wo != (d.W + d.stride - 1) / d.stride) return DPP_E_BADARG;
    const void* al[] = {d.X, d.Y, d.W1, d.W2, d.W3, d.b3, d.bn0.mean, d.bn0.inv_std, d.bn0.gamma, d.bn0.beta, d.Wsc, d.bsc};
    for (const void* p : al)
        if (!rb_al16(p)) return DPP_E_UNSUPPORTED;
    if ((long)d.N * d.H * d.W * (long)(d.Cin > d.Cout ? d.Cin : d.Cout) >= (1L << 31)) return DPP_E_UNSUPPORTED;     // 32-bit element offsets
    if (d.store & ~(DPP_ST_A | DPP_ST_C)) return DPP_E_BADARG;
    if ((d.store & DPP_ST_A) && !(d.store & DPP_ST_C)) return DPP_E_UNSUPPORTED;
    a.d = d;
    a.prof = dpp_prof_buffer;
    // the tile follows from the bottleneck width alone (never from the batch): 8 x 8 (16 and 32 channels), 4 x 8 (64).  (16 channels: 8 x 16
    // tiles -- 1.4x instead of 1.56x halo recomputation, but 62 KB of LDS = two workgroups per CU -- measured 13 us per forward pass slower
    // than 8 x 8 with four per CU: these blocks are bound by memory-level parallelism.  32 channels on 4 x 8 tiles: 30 us slower.)
    const int th = d.Nb == 64 ? 4 : 8, tw = 8;
    a.lth = rb_ilog2(th); a.ltw = rb_ilog2(tw);
    a.tiles_x = dpp_cdiv(d.Wo, tw); a.tiles_y = dpp_cdiv(d.Ho, th);
    a.ntiles = a.tiles_x * a.tiles_y * d.N;
    a.m_tw2 = (unsigned)(0x100000000ull / (unsigned)(tw + 2)) + 1u;
    const int HP = (th + 2) * (tw + 2), HPP = (HP + 63) / 64 * 64, BM = th * tw, LD1 = d.Nb + 4, LDA = d.Cin + 4;
    // LDS: halo | A2 | (projection: A1 | images); identity: A1 + images alias the halo
    const int wn2 = d.Nb / 16 < 4 ? d.Nb / 16 : 4, wm2 = 4 / wn2, wrows = BM / wm2, wcols = 64 / wn2;      // a wave's share of a 64-column pass
    const int img = 4 * wrows * (wcols + 4);                                                   // four wave-private images
    const int r0a = HPP * LDA, r0b = HPP * LD1 + img;
    lds = ((size_t)(proj ? r0a + r0b : (r0a > r0b ? r0a : r0b)) + (size_t)BM * LD1) * sizeof(float);
    if (lds > 160 * 1024) return DPP_E_UNSUPPORTED;
    a.xcd_chunk = (a.ntiles % 8 == 0 && a.ntiles >= 64) ? a.ntiles / 8 : 0;
    return DPP_OK;
}

// would dpp_resblock_eval take this descriptor?  (DPP_OK, or the status it would return: alignment, LDS size, 32-bit offsets, storage
// combination -- the engine asks while it COMPILES a net, so that a block the kernel refuses is lowered layer by layer, not a failed run)
extern "C" int dpp_resblock_eval_check(const dpp_resblock_desc* desc) {
    RBArgs a;
    size_t lds;
    bool proj;
    return rb_prepare(desc, a, lds, proj);
}

extern "C" int dpp_resblock_eval(const dpp_resblock_desc* desc, dpp_stream_t stream) {
    RBArgs a;
    size_t lds;
    bool proj;
    const int prep = rb_prepare(desc, a, lds, proj);
    if (prep != DPP_OK) return prep;
    const dpp_resblock_desc& d = a.d;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // storage: float32 in and out; bf16 mode: bf16 out with float32 in (the block behind the stem, whose pooled map stays float32) or bf16 in
    const bool x16 = (d.store & DPP_ST_A) != 0, y16 = (d.store & DPP_ST_C) != 0;
#define DPP_RBT(NB_, BM_, RT_, P_, XT_, YT_) do { \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_eval_kernel<NB_, BM_, RT_, P_, XT_, YT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DPP_E_UNSUPPORTED; \
        DPP_LAUNCH((resblock_eval_kernel<NB_, BM_, RT_, P_, XT_, YT_>), dim3(a.ntiles), dim3(DPP_THREADS), lds, st, a); return dpp_launch_status(); } while (0)
#define DPP_RBK(NB_, BM_, RT_, P_) do { if (x16) DPP_RBT(NB_, BM_, RT_, P_, dpp_bf16, dpp_bf16); else if (y16) DPP_RBT(NB_, BM_, RT_, P_, float, dpp_bf16); else DPP_RBT(NB_, BM_, RT_, P_, float, float); } while (0)
#define DPP_RB(NB_, BM_, RT_) do { if (proj) DPP_RBK(NB_, BM_, RT_, true); else DPP_RBK(NB_, BM_, RT_, false); } while (0)
    if (d.Nb == 16) DPP_RB(16, 64, 2);
    if (d.Nb == 32) DPP_RB(32, 64, 2);
    if (d.Nb == 64) DPP_RB(64, 32, 1);
#undef DPP_RB
#undef DPP_RBK
#undef DPP_RBT
    return DPP_E_UNSUPPORTED;
}
