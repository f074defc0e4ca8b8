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
    int chunks, KC, lqa;     // phase-A K chunks, channels per chunk, log2(KC / 4)
    int nbufA;               // LDS buffers of the input halo: 1, 2, or `chunks` (projection: the whole activated halo stays)
    int r0_floats, wbuf_floats;
    int xcd_chunk;           // ntiles / 8 when the tile -> XCD swizzle applies, else 0
};

__device__ __forceinline__ float4 rb_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void rb_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// v = relu((x - mean) * (gamma * inv_std) + beta), the arithmetic of dpp_act1 with dpp_bn_eval_coeffs' scale
__device__ __forceinline__ float rb_bnrelu(float x, float mu, float sc, float be) { return fmaxf((x - mu) * sc + be, 0.0f); }

// NB: bottleneck width (16 / 32 / 64); BM = TH * TW output pixels per workgroup; RT1: 16-row tiles of the halo per wave (halo padded to
// RT1 * 64 rows); PROJ: projection block (strided input, shortcut convolution) instead of the identity block.
template <int NB, int BM, int RT1, bool PROJ>
__global__ __launch_bounds__(DPP_THREADS) void resblock_eval_kernel(RBArgs a) {
    dpp_kernarg_warm<sizeof(RBArgs)>();
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int CT1 = NB / 16;                     // column tiles of the bottleneck width
    constexpr int HPP = RT1 * 64;                    // padded halo rows
    constexpr int LD1 = NB + 4;                      // row stride of A1 / A2 and of the W2 / W3 slices
    constexpr int RTB = BM / 16;                     // row tiles of the output tile
    constexpr int WM2 = RTB < 4 ? RTB : 4, RM2 = RTB / WM2, WN2 = 4 / WM2;
    constexpr int CN2 = CT1 / WN2;                   // phase B: column tiles per wave
    constexpr int CN3 = 4 / WN2;                     // phase C: column tiles per wave of a 64-column pass
    constexpr int ASLOTS = HPP / 16;                 // halo float4 slots per thread and chunk (KC = 64; fewer rows per sweep for KC = 32)
    constexpr int WSLOTS = CT1;                      // weight-slice float4 slots per thread
    constexpr int LDI = 64 + 4;                      // row stride of the output image of a pass
    static_assert(CN2 >= 1 && RM2 >= 1, "tile shape");
    const dpp_resblock_desc& d = a.d;
    const int TH = 1 << a.lth, TW = 1 << a.ltw, TW2 = TW + 2, HP = (TH + 2) * TW2;
    const int KC = a.KC, LDA = KC + 4, Cin = d.Cin, Cout = d.Cout, S = d.stride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wm = wave % WM2, wn = wave / WM2;

    float* R0 = smem;                                 // input-halo buffers; later A1 [HPP][LD1] + the output image [BM][LDI]
    float* Wb = smem + a.r0_floats;                   // two weight-slice buffers
    float* A2 = Wb + 2 * a.wbuf_floats;               // [BM][LD1]
    float* A1 = PROJ ? A2 + BM * LD1 : R0;            // (projection: the activated halo stays alive for the shortcut)
    float* IMG = PROJ ? A1 + HPP * LD1 : R0 + HPP * LD1;

    int bid = blockIdx.x;
    if (a.xcd_chunk) bid = (bid & 7) * a.xcd_chunk + (bid >> 3);      // neighbouring tiles (shared halo rows) on one XCD's L2
    const int bx = bid % a.tiles_x;
    const int tq = bid / a.tiles_x;
    const int by = tq % a.tiles_y, n = tq / a.tiles_y;
    const int y0 = by << a.lth, x0 = bx << a.ltw;
    const int Ho = d.Ho, Wo = d.Wo;
    const float* Xn = d.X + (size_t)n * d.H * d.W * Cin;

    // ---- per-lane column constants of the three epilogues (requested now, used after the first products) ----
    float e1b[CT1], e1m[CT1], e1s[CT1], e1t[CT1];
#pragma unroll
    for (int ct = 0; ct < CT1; ++ct) {
        const int c = ct * 16 + l15;
        e1b[ct] = d.b1[c]; e1m[ct] = d.bn1.mean[c]; e1s[ct] = d.bn1.gamma[c] * d.bn1.inv_std[c]; e1t[ct] = d.bn1.beta[c];
    }
    float e2b[CN2], e2m[CN2], e2s[CN2], e2t[CN2];
#pragma unroll
    for (int ct = 0; ct < CN2; ++ct) {
        const int c = (wn * CN2 + ct) * 16 + l15;
        e2b[ct] = d.b2[c]; e2m[ct] = d.bn2.mean[c]; e2s[ct] = d.bn2.gamma[c] * d.bn2.inv_std[c]; e2t[ct] = d.bn2.beta[c];
    }

    // ---- halo geometry of this thread's staging slots: slot u is halo position hp = (tid >> lqa) + u * hstep ----
    const int qa = (tid & ((1 << a.lqa) - 1)) * 4;     // channel quad inside a chunk
    const int hstep = DPP_THREADS >> a.lqa;
    const int nslots = HPP / hstep;                     // <= ASLOTS (KC = 64), 2x fewer rows per sweep would exceed it: host keeps KC >= 32 with HPP * KC <= ASLOTS * 1024
    int xoff[ASLOTS];                                   // element offset of the halo pixel in this image, -1 outside
#pragma unroll
    for (int u = 0; u < ASLOTS; ++u) {
        xoff[u] = -1;
        if (u < nslots) {
            const int hp = (tid >> a.lqa) + u * hstep;
            const int hy = (int)__umulhi((unsigned)hp, a.m_tw2), hx = hp - hy * TW2;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            if (hp < HP && y >= 0 && y < Ho && x >= 0 && x < Wo) xoff[u] = ((y * S) * d.W + x * S) * Cin;
        }
    }
    float4 areg[ASLOTS];
    float4 amu, asc, abe;
    auto fetchA = [&](int chunk) {
        const int c0 = chunk * KC + qa;
        amu = rb_ld4(d.bn0.mean + c0);
        const float4 g = rb_ld4(d.bn0.gamma + c0), is = rb_ld4(d.bn0.inv_std + c0);
        asc = make_float4(g.x * is.x, g.y * is.y, g.z * is.z, g.w * is.w);
        abe = rb_ld4(d.bn0.beta + c0);
#pragma unroll
        for (int u = 0; u < ASLOTS; ++u) {
            areg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < nslots && xoff[u] >= 0) areg[u] = rb_ld4(Xn + xoff[u] + c0);
        }
    };
    auto commitA = [&](float* buf) {
#pragma unroll
        for (int u = 0; u < ASLOTS; ++u) {
            if (u < nslots) {
                const int hp = (tid >> a.lqa) + u * hstep;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);          // zero padding is applied AFTER the activation
                if (xoff[u] >= 0) {
                    const float4 v = areg[u];
                    t = make_float4(rb_bnrelu(v.x, amu.x, asc.x, abe.x), rb_bnrelu(v.y, amu.y, asc.y, abe.y),
                                    rb_bnrelu(v.z, amu.z, asc.z, abe.z), rb_bnrelu(v.w, amu.w, asc.w, abe.w));
                }
                rb_st4(&buf[hp * LDA + qa], t);
            }
        }
    };

    // ---- the weight-slice stream ----
    float4 wreg[WSLOTS];
    auto fetchW1 = [&](int chunk) {                     // [NB][KC] of W1 [NB][Cin]
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const int slot = tid + s * DPP_THREADS, j = slot >> a.lqa;
            wreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < NB) wreg[s] = rb_ld4(d.W1 + (size_t)j * Cin + chunk * KC + qa);
        }
    };
    auto commitW1 = [&](float* buf) {
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const int slot = tid + s * DPP_THREADS, j = slot >> a.lqa;
            if (j < NB) rb_st4(&buf[j * LDA + qa], wreg[s]);
        }
    };
    constexpr int Q2 = NB / 4;                          // quads per row of the NB-deep slices
    auto fetchW2 = [&](int tap) {                       // [NB][NB] of W2 [NB][9][NB]
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const int slot = tid + s * DPP_THREADS, j = slot / Q2, c0 = (slot % Q2) * 4;
            wreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < NB) wreg[s] = rb_ld4(d.W2 + ((size_t)j * 9 + tap) * NB + c0);
        }
    };
    auto fetchW3 = [&](int pass) {                      // [64][NB] of W3 [Cout][NB]
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const int slot = tid + s * DPP_THREADS, j = slot / Q2, c0 = (slot % Q2) * 4;
            wreg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < 64) wreg[s] = rb_ld4(d.W3 + ((size_t)pass * 64 + j) * NB + c0);
        }
    };
    auto commitWn = [&](float* buf, int rows) {         // NB-deep slices (W2 taps: rows = NB, W3 passes: rows = 64)
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) {
            const int slot = tid + s * DPP_THREADS, j = slot / Q2, c0 = (slot % Q2) * 4;
            if (j < rows) rb_st4(&buf[j * LD1 + c0], wreg[s]);
        }
    };
    auto fetchWsc = [&](int pass, int chunk) {          // projection shortcut: [64][KC] of Wsc [Cout][Cin]
#pragma unroll
        for (int s = 0; s < 4; ++s) {                   // (64 rows x KC / 4 quads: up to 4 slots; uses the halo registers, free by then)
            const int slot = tid + s * DPP_THREADS, j = slot >> a.lqa;
            areg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < 64) areg[s] = rb_ld4(d.Wsc + ((size_t)pass * 64 + j) * Cin + chunk * KC + qa);
        }
    };
    auto commitWsc = [&](float* buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int slot = tid + s * DPP_THREADS, j = slot >> a.lqa;
            if (j < 64) rb_st4(&buf[j * LDA + qa], areg[s]);
        }
    };

    // ================================ phase A: c1 over the halo ================================
    f32x4 acc1[RT1][CT1];
#pragma unroll
    for (int i = 0; i < RT1; ++i)
#pragma unroll
        for (int j = 0; j < CT1; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int chunks = a.chunks;
    fetchA(0);
    fetchW1(0);
    commitA(R0);
    commitW1(Wb);
    __syncthreads();
    int ws = 0;                                         // index of the weight slice in Wb[ws & 1]
    for (int c = 0; c < chunks; ++c, ++ws) {
        const float* Ab = R0 + (a.nbufA == 1 ? 0 : (a.nbufA == 2 ? (c & 1) : c)) * HPP * LDA;
        const float* Wc = Wb + (ws & 1) * a.wbuf_floats;
        const bool more = c + 1 < chunks;
        if (more) { fetchA(c + 1); fetchW1(c + 1); } else fetchW2(0);
        for (int kc = 0; kc < KC; kc += 16) {
            float4 av[RT1], bv[CT1];
#pragma unroll
            for (int rt = 0; rt < RT1; ++rt) av[rt] = rb_ld4(&Ab[((wave * RT1 + rt) * 16 + l15) * LDA + kc + kq * 4]);
#pragma unroll
            for (int ct = 0; ct < CT1; ++ct) bv[ct] = rb_ld4(&Wc[(ct * 16 + l15) * LDA + kc + kq * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT1; ++ct)
                        acc1[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), dpp_f4_get(bv[ct], t), acc1[rt][ct], 0, 0, 0);
        }
        float* Wn = Wb + ((ws + 1) & 1) * a.wbuf_floats;
        if (more) {
            if (a.nbufA == 1) __syncthreads();          // single halo buffer: everybody has read chunk c
            float* An = R0 + (a.nbufA == 1 ? 0 : (a.nbufA == 2 ? ((c + 1) & 1) : (c + 1)) * HPP * LDA);
            commitA(An);
            commitW1(Wn);
        } else {
            commitWn(Wn, NB);
        }
        __syncthreads();
    }
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
                const float v = rb_bnrelu(acc1[rt][ct][r] + e1b[ct], e1m[ct], e1s[ct], e1t[ct]);
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
    for (int tap = 0; tap < 9; ++tap, ++ws) {
        const float* Wc = Wb + (ws & 1) * a.wbuf_floats;
        if (tap + 1 < 9) fetchW2(tap + 1); else fetchW3(0);
        const int toff = (tap / 3 - 1) * TW2 + (tap % 3 - 1);
#pragma unroll
        for (int kc = 0; kc < NB; kc += 16) {
            float4 av[RM2], bv[CN2];
#pragma unroll
            for (int rt = 0; rt < RM2; ++rt) av[rt] = rb_ld4(&A1[(hbase[rt] + toff) * LD1 + kc + kq * 4]);
#pragma unroll
            for (int ct = 0; ct < CN2; ++ct) bv[ct] = rb_ld4(&Wc[((wn * CN2 + ct) * 16 + l15) * LD1 + kc + kq * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CN2; ++ct)
                        acc2[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), dpp_f4_get(bv[ct], t), acc2[rt][ct], 0, 0, 0);
        }
        commitWn(Wb + ((ws + 1) & 1) * a.wbuf_floats, tap + 1 < 9 ? NB : 64);
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (wm * RM2 + rt) * 16 + kq * 4 + r;
#pragma unroll
            for (int ct = 0; ct < CN2; ++ct)
                A2[row * LD1 + (wn * CN2 + ct) * 16 + l15] = rb_bnrelu(acc2[rt][ct][r] + e2b[ct], e2m[ct], e2s[ct], e2t[ct]);
        }
    __syncthreads();

    // ================================ phase C: c3 (+ shortcut) in passes of 64 output channels ================================
    // a thread of the store sweep owns one column quad of the pass and rows rb, rb + 16, ...
    const int cq = tid & 15, rb = tid >> 4;
    constexpr int RSW = BM / 16;                        // rows per thread in the sweep
    int ooff[RSW];                                      // element offset of the row's pixel in Y (and, identity block, in X), -1 outside
#pragma unroll
    for (int it = 0; it < RSW; ++it) {
        const int row = rb + it * 16;
        const int y = y0 + (row >> a.ltw), x = x0 + (row & (TW - 1));
        ooff[it] = (y < Ho && x < Wo) ? ((n * Ho + y) * Wo + x) : -1;
    }
    const int npass = Cout >> 6;
    auto product_c3 = [&](f32x4 (&acc3)[RM2][CN3], const float* Wc) {
#pragma unroll
        for (int kc = 0; kc < NB; kc += 16) {
            float4 av[RM2], bv[CN3];
#pragma unroll
            for (int rt = 0; rt < RM2; ++rt) av[rt] = rb_ld4(&A2[((wm * RM2 + rt) * 16 + l15) * LD1 + kc + kq * 4]);
#pragma unroll
            for (int ct = 0; ct < CN3; ++ct) bv[ct] = rb_ld4(&Wc[((wn * CN3 + ct) * 16 + l15) * LD1 + kc + kq * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CN3; ++ct)
                        acc3[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), dpp_f4_get(bv[ct], t), acc3[rt][ct], 0, 0, 0);
        }
    };
    auto store_pass = [&](f32x4 (&acc3)[RM2][CN3], int pass, const float4& bq, const float4 (&res)[RSW]) {
        if (pass > 0) __syncthreads();                  // the previous pass's image has been read by everybody
#pragma unroll
        for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CN3; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    IMG[((wm * RM2 + rt) * 16 + kq * 4 + r) * LDI + (wn * CN3 + ct) * 16 + l15] = acc3[rt][ct][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < RSW; ++it) {
            if (ooff[it] >= 0) {
                const float4 v = rb_ld4(&IMG[(rb + it * 16) * LDI + cq * 4]);
                rb_st4(d.Y + (size_t)ooff[it] * Cout + pass * 64 + cq * 4,
                       make_float4(v.x + bq.x + res[it].x, v.y + bq.y + res[it].y, v.z + bq.z + res[it].z, v.w + bq.w + res[it].w));
            }
        }
    };
    if constexpr (!PROJ) {
        for (int pass = 0; pass < npass; ++pass, ++ws) {
            const float* Wc = Wb + (ws & 1) * a.wbuf_floats;
            if (pass + 1 < npass) fetchW3(pass + 1);
            // the residual rows of this pass, requested before the products
            float4 res[RSW];
            const float4 bq = rb_ld4(d.b3 + pass * 64 + cq * 4);
#pragma unroll
            for (int it = 0; it < RSW; ++it) {
                res[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ooff[it] >= 0) res[it] = rb_ld4(d.X + (size_t)ooff[it] * Cin + pass * 64 + cq * 4);
            }
            f32x4 acc3[RM2][CN3];
#pragma unroll
            for (int i = 0; i < RM2; ++i)
#pragma unroll
                for (int j = 0; j < CN3; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            product_c3(acc3, Wc);
            if (pass + 1 < npass) commitWn(Wb + ((ws + 1) & 1) * a.wbuf_floats, 64);
            store_pass(acc3, pass, bq, res);
        }
    } else {
        // Projection block: buffer U holds the W3 slice of the pass, buffer V the [64][KC] slices of the shortcut's weights, whose product
        // runs on the centre pixels of the activated input (all K chunks are still alive in R0).  Three of the twenty blocks take this
        // path: plain fetch / barrier / commit / barrier sequencing, no pipelining across slices.
        float* U = Wb + (ws & 1) * a.wbuf_floats;
        float* V = Wb + ((ws + 1) & 1) * a.wbuf_floats;
        int hc[RM2];                                    // halo row of this lane's A rows (centre tap)
#pragma unroll
        for (int rt = 0; rt < RM2; ++rt) {
            const int row = (wm * RM2 + rt) * 16 + l15;
            hc[rt] = ((row >> a.ltw) + 1) * TW2 + (row & (TW - 1)) + 1;
        }
        float4 res[RSW];
#pragma unroll
        for (int it = 0; it < RSW; ++it) res[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int pass = 0; pass < npass; ++pass) {
            if (pass > 0) {
                fetchW3(pass);
                __syncthreads();
                commitWn(U, 64);
                __syncthreads();
            }
            fetchWsc(pass, 0);
            float4 bq = rb_ld4(d.b3 + pass * 64 + cq * 4);
            { const float4 t = rb_ld4(d.bsc + pass * 64 + cq * 4); bq.x += t.x; bq.y += t.y; bq.z += t.z; bq.w += t.w; }
            f32x4 acc3[RM2][CN3];
#pragma unroll
            for (int i = 0; i < RM2; ++i)
#pragma unroll
                for (int j = 0; j < CN3; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            product_c3(acc3, U);
            for (int c = 0; c < chunks; ++c) {
                if (c > 0) fetchWsc(pass, c);
                __syncthreads();                         // V has been read by everybody
                commitWsc(V);
                __syncthreads();
                const float* Ab = R0 + c * HPP * LDA;
                for (int kc = 0; kc < KC; kc += 16) {
                    float4 av[RM2], bv[CN3];
#pragma unroll
                    for (int rt = 0; rt < RM2; ++rt) av[rt] = rb_ld4(&Ab[hc[rt] * LDA + kc + kq * 4]);
#pragma unroll
                    for (int ct = 0; ct < CN3; ++ct) bv[ct] = rb_ld4(&V[((wn * CN3 + ct) * 16 + l15) * LDA + kc + kq * 4]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int rt = 0; rt < RM2; ++rt)
#pragma unroll
                            for (int ct = 0; ct < CN3; ++ct)
                                acc3[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpp_f4_get(av[rt], t), dpp_f4_get(bv[ct], t), acc3[rt][ct], 0, 0, 0);
                }
            }
            store_pass(acc3, pass, bq, res);
        }
    }
}

int rb_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

bool rb_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dpp_resblock_eval_ok(int Cin, int Cout, int Nb, int stride, int projection) {
    if (Nb != 16 && Nb != 32 && Nb != 64) return 0;
    if (Cout < 64 || (Cout & 63)) return 0;
    if (!projection) return stride == 1 && Cin == Cout;                    // identity block
    return (stride == 1 || stride == 2) && (Cin == 32 || (Cin >= 64 && Cin % 64 == 0 && Cin <= 256));
}

extern "C" int dpp_resblock_eval(const dpp_resblock_desc* desc, dpp_stream_t stream) {
    if (!desc) return DPP_E_BADARG;
    const dpp_resblock_desc& d = *desc;
    if (!d.X || !d.Y || !d.W1 || !d.b1 || !d.W2 || !d.b2 || !d.W3 || !d.b3 || d.N < 1 || d.H < 1 || d.W < 1) return DPP_E_BADARG;
    const dpp_bn_eval* bns[3] = {&d.bn0, &d.bn1, &d.bn2};
    for (const dpp_bn_eval* b : bns)
        if (!b->mean || !b->inv_std || !b->gamma || !b->beta) return DPP_E_BADARG;
    const bool proj = d.Wsc != nullptr;
    if (proj && !d.bsc) return DPP_E_BADARG;
    if (!dpp_resblock_eval_ok(d.Cin, d.Cout, d.Nb, d.stride, proj ? 1 : 0)) return DPP_E_UNSUPPORTED;
    if (d.Ho != (d.H + d.stride - 1) / d.stride || d.Wo != (d.W + d.stride - 1) / d.stride) return DPP_E_BADARG;
    const void* al[] = {d.X, d.Y, d.W1, d.W2, d.W3, d.b3, d.bn0.mean, d.bn0.inv_std, d.bn0.gamma, d.bn0.beta, d.Wsc, d.bsc};
    for (const void* p : al)
        if (!rb_al16(p)) return DPP_E_UNSUPPORTED;
    if ((long)d.N * d.H * d.W * (long)(d.Cin > d.Cout ? d.Cin : d.Cout) >= (1L << 31)) return DPP_E_UNSUPPORTED;     // 32-bit element offsets
    RBArgs a;
    a.d = d;
    // the tile follows from the bottleneck width alone (never from the batch): 8 x 16 (16 channels), 8 x 8 (32), 4 x 8 (64)
    const int th = d.Nb == 64 ? 4 : 8, tw = d.Nb == 16 ? 16 : 8;
    a.lth = rb_ilog2(th); a.ltw = rb_ilog2(tw);
    a.tiles_x = dpp_cdiv(d.Wo, tw); a.tiles_y = dpp_cdiv(d.Ho, th);
    a.ntiles = a.tiles_x * a.tiles_y * d.N;
    a.m_tw2 = (unsigned)(0x100000000ull / (unsigned)(tw + 2)) + 1u;
    a.KC = d.Cin < 64 ? d.Cin : 64;
    if (a.KC != 32 && a.KC != 64) return DPP_E_UNSUPPORTED;
    a.chunks = d.Cin / a.KC;
    a.lqa = rb_ilog2(a.KC / 4);
    const int HP = (th + 2) * (tw + 2), HPP = (HP + 63) / 64 * 64, BM = th * tw, LD1 = d.Nb + 4, LDA = a.KC + 4;
    a.nbufA = proj ? a.chunks : (a.chunks == 1 ? 1 : (HPP == 64 ? 2 : 1));
    const int r0a = a.nbufA * HPP * LDA, r0b = HPP * LD1 + BM * 68;
    a.r0_floats = proj ? r0a : (r0a > r0b ? r0a : r0b);      // (projection: A1 and the output image live behind A2 instead)
    const int w1 = d.Nb * LDA, w3 = 64 * LD1, wsc = proj ? 64 * LDA : 0;
    a.wbuf_floats = w1 > w3 ? w1 : w3;
    if (wsc > a.wbuf_floats) a.wbuf_floats = wsc;
    const size_t lds = ((size_t)a.r0_floats + 2 * (size_t)a.wbuf_floats + (size_t)BM * LD1 + (proj ? (size_t)r0b : 0)) * sizeof(float);
    if (lds > 160 * 1024) return DPP_E_UNSUPPORTED;
    a.xcd_chunk = (a.ntiles % 8 == 0 && a.ntiles >= 64) ? a.ntiles / 8 : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DPP_RBK(NB_, BM_, RT_, P_) do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_eval_kernel<NB_, BM_, RT_, P_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        DPP_LAUNCH((resblock_eval_kernel<NB_, BM_, RT_, P_>), dim3(a.ntiles), dim3(DPP_THREADS), lds, st, a); return dpp_launch_status(); } while (0)
#define DPP_RB(NB_, BM_, RT_) do { if (proj) DPP_RBK(NB_, BM_, RT_, true); else DPP_RBK(NB_, BM_, RT_, false); } while (0)
    if (d.Nb == 16) DPP_RB(16, 128, 3);
    if (d.Nb == 32) DPP_RB(32, 64, 2);
    if (d.Nb == 64) DPP_RB(64, 32, 1);
#undef DPP_RB
#undef DPP_RBK
    return DPP_E_UNSUPPORTED;
}
