// stem.hip -- the ResNet stem ConvPoolLayer on single-channel depth crops for gfx950.
//
// Reference arithmetic: /root/reference/src/net/convpoollayer.py:251-282 as instantiated at
// /root/reference/src/net/resnet.py:128-133: conv2d 5x5 'half' 1 -> Co, then 2x2 max-pool (ignore_border),
// then the bias AFTER pooling, no activation.
//
// Forward (stem_fwd_kernel): implicit GEMM on v_mfma_f32_16x16x4_f32 with K = 25 taps (padded to 28).
// A workgroup owns a 16x16 patch of conv pixels (8x8 pooled outputs); the 20x20 input halo and the weights
// live in LDS.  The 16 rows of an MFMA tile are ordered as 4 pooling windows x 4 window pixels, so the
// four D registers of a lane are exactly one pooling window: the max-pool and its tie mask (bit j set = window
// element j equals the maximum; Theano's MaxPoolGrad gives the gradient to every tied element, and the constant
// background of a depth crop does tie) happen in registers, the 128x128x32 conv map never touches memory.  The single input channel makes NCHW == NHWC: loads are the
// reference's own NCHW depth patches, coalesced along x.
//
// Filter gradient (stem_wgrad_kernel): dW[o][tap] = sum over pooled outputs p and tied window elements j of
// dY[p][o] * x[pixel(p, j) + tap] (the max-pool routes each gradient to conv pixels that differ per channel, so this is
// not a GEMM);
// VALU kernel, thread = (channel, pixel group), per-workgroup partials, fixed-order reduce.
#include "dpp_common.h"

namespace {

constexpr int KS = 5, PAD = 2, NTAP = 25, KPAD = 28;
constexpr int TC = 16;             // conv pixels per tile side
constexpr int LX = TC + 2 * PAD;   // 20: halo side
constexpr int LXP = LX + 1;        // padded row

template <int CN>   // CN = Co / 16 column tiles
__global__ __launch_bounds__(DPP_THREADS) void stem_fwd_kernel(const float* __restrict__ X, int N, int H, int W, const float* __restrict__ Wk,
                                                               const float* __restrict__ bias, int Co, float* __restrict__ Y,
                                                               uint8_t* __restrict__ arg, int tiles_x, int tiles_y,
                                                               float* __restrict__ stats, int y16) {
    dpp_kernarg_warm<128>();
    __shared__ float xs[LX * LXP];
    __shared__ float Ws[KPAD * CN * 16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int bx = blockIdx.x % tiles_x, by = (blockIdx.x / tiles_x) % tiles_y, n = blockIdx.x / (tiles_x * tiles_y);
    const int cy0 = by * TC, cx0 = bx * TC;
    const int Hp = H >> 1, Wp = W >> 1;
    const float* img = X + (size_t)n * H * W;
    for (int s = tid; s < LX * LX; s += DPP_THREADS) {
        int hy = s / LX, hx = s - hy * LX;
        int y = cy0 + hy - PAD, x = cx0 + hx - PAD;
        xs[hy * LXP + hx] = (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.0f;
    }
    for (int s = tid; s < KPAD * CN * 16; s += DPP_THREADS) {
        int k = s / (CN * 16), o = s - k * (CN * 16);
        Ws[s] = (k < NTAP && o < Co) ? Wk[(size_t)o * NTAP + k] : 0.0f;
    }
    __syncthreads();

    // B fragments for all 7 k-steps stay in registers: k = 4*t + kq
    float bf[7][CN];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) bf[t][ct] = Ws[(4 * t + kq) * (CN * 16) + ct * 16 + l15];
    // halo offsets of the taps this lane feeds
    int koff[7];
    bool kval[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        int k = 4 * t + kq;
        kval[t] = k < NTAP;
        int kk = kval[t] ? k : 0;
        koff[t] = (kk / KS) * LXP + (kk % KS);
    }

    float vals[4][CN];               // this lane's pooled outputs (kept for the fused BatchNorm statistics)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) vals[i][j] = 0.0f;
    // wave w owns pooled rows 2w, 2w+1 of the 8x8 pooled tile: 4 row-tiles of (4 windows x 4 pixels)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int ply = 2 * wave + (rt >> 1);
        // A row i = 4*q + p : window q (pooled x = (rt&1)*4 + q), window pixel p = (py, px) = (p>>1, p&1)
        const int qa = l15 >> 2, pa = l15 & 3;
        const int ay = 2 * ply + (pa >> 1), ax = 2 * (((rt & 1) << 2) + qa) + (pa & 1);   // conv pixel in tile
        const int abase = ay * LXP + ax;     // halo origin is conv pixel (-PAD,-PAD) => tap (dy,dx) at +dy*LXP+dx
        f32x4 acc[CN];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            float av = kval[t] ? xs[abase + koff[t]] : 0.0f;
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[t][ct], acc[ct], 0, 0, 0);
        }
        // D: lane (col = l15, window q = kq) holds the 4 window pixels in acc[ct][0..3] -> max-pool in registers
        const int py = (cy0 >> 1) + ply, px = (cx0 >> 1) + ((rt & 1) << 2) + kq;
        if (py < Hp && px < Wp) {
            size_t o = (((size_t)n * Hp + py) * Wp + px) * Co;
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                int col = ct * 16 + l15;
                if (col < Co) {
                    float best = fmaxf(fmaxf(acc[ct][0], acc[ct][1]), fmaxf(acc[ct][2], acc[ct][3]));
                    int ties = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ties |= (acc[ct][r] == best) ? (1 << r) : 0;
                    if (y16) reinterpret_cast<dpp_bf16*>(Y)[o + col] = (dpp_bf16)(best + bias[col]);      // bf16 storage: rounded on the store,
                    else Y[o + col] = best + bias[col];                                                    // statistics from the f32 value
                    vals[rt][ct] = best + bias[col];
                    if (arg) arg[o + col] = (uint8_t)ties;
                }
            }
        }
    }
    if (stats != nullptr) {
        // per-tile (mean, M2) of the 64 pooled outputs of every channel: the BatchNorm statistics partial of the tensor just
        // written (the host only asks for it when all tiles are full), combined by dpp_bn_finalize with rows_per_block = 64
        float sm[CN], m2[CN];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) sm[ct] = vals[0][ct] + vals[1][ct] + vals[2][ct] + vals[3][ct];
        dpp_tile_colsum<CN, 4, 1, CN * 16>(sm, xs, wave, 0, l15, kq);
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            sm[ct] *= (1.0f / 64.0f);
            float t = 0.0f;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) { float dv = vals[rt][ct] - sm[ct]; t += dv * dv; }
            m2[ct] = t;
        }
        dpp_tile_colsum<CN, 4, 1, CN * 16>(m2, xs, wave, 0, l15, kq);
        if (kq == 0 && wave == 0) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                int col = ct * 16 + l15;
                if (col < Co) {
                    stats[dpp_partial_index(0, col, blockIdx.x, Co, gridDim.x)] = sm[ct];
                    stats[dpp_partial_index(1, col, blockIdx.x, Co, gridDim.x)] = m2[ct];
                }
            }
        }
    }
}

// Filter gradient on the matrix cores.  With the max-pool routing folded into the A operand,
//     dW[o][tap] = sum over conv pixels pix of  dYc[pix][o] * x[pix + tap],   dYc[pix][o] = dY[pool(pix)][o] * tie(pix, o),
// is a GEMM with M = o (<= 32), N = tap (25, padded to 32) and K = pixels.  One MFMA k-step (4 k-values) is one pooling
// window, k = the window pixel a.  A workgroup walks 16x16-pixel tiles (8x8 windows): per tile it stages the input halo and
// the MASKED gradients gm[a][window][o] = dY[window][o] if tie bit a of ties[window][o] else 0 in LDS (16-byte global loads,
// the next tile's loads in flight under this tile's MFMAs); its four waves take 16 windows each, lane (o, a) reads gm as A,
// lane (tap, a) reads x[window pixel a + tap] from the halo as B, and the whole 32x32 product stays in 4 accumulators per
// wave until the waves meet in LDS at the end.  (A VALU version spent 93 us issuing one FMA per tap, window pixel and
// channel -- the constant background of a depth crop ties all four pixels -- alone at the end of the step.)
constexpr int LXW = LX + 2;              // 22: row pitch of the halo image in this kernel
constexpr int GM_A = 64 * 32 + 16;       // floats between the four window-pixel planes of gm (the +16 spreads them over the banks)

__global__ __launch_bounds__(DPP_THREADS) void stem_wgrad_kernel(const float* __restrict__ X, int N, int H, int W, const float* __restrict__ dY,
                                                                 const uint8_t* __restrict__ arg, int Co, float* __restrict__ partial,
                                                                 int tiles_x, int tiles_y, int tiles_per_block, int total_tiles) {
    dpp_kernarg_warm<128>();
    __shared__ __attribute__((aligned(16))) float gm[4 * GM_A];     // 33 KB; re-used for the final reduction (4 x 32 x 32 floats)
    __shared__ float xs[LX * LXW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int Hp = H >> 1, Wp = W >> 1;
    const bool vec = (Co & 3) == 0 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(arg) & 3) == 0;
    // B operand: this lane's tap(s) and window pixel give a constant offset into the halo image
    int offB[2];
    bool okB[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int tap = ct * 16 + l15;
        okB[ct] = tap < NTAP;
        const int tt = okB[ct] ? tap : 0;
        offB[ct] = ((kq >> 1) + tt / KS) * LXW + (kq & 1) + tt % KS;
    }
    // staging registers of one tile: two (window, channel quad) items and two halo pixels per thread
    float4 rg[2];
    unsigned rt_[2];
    float rh[2];
    auto load_tile = [&](int tile) {
        const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int cy0 = by * TC, cx0 = bx * TC;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = tid + i * DPP_THREADS, p = item >> 3, oq = item & 7;
            const int py = (cy0 >> 1) + (p >> 3), px = (cx0 >> 1) + (p & 7);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned t = 0;
            if (py < Hp && px < Wp && oq * 4 < Co) {
                const size_t idx = (((size_t)n * Hp + py) * Wp + px) * Co + oq * 4;
                if (vec) {
                    g = *reinterpret_cast<const float4*>(dY + idx);
                    t = *reinterpret_cast<const unsigned*>(arg + idx);
                } else {
                    float gv[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int j = 0; j < 4; ++j)
                        if (oq * 4 + j < Co) { gv[j] = dY[idx + j]; t |= (unsigned)arg[idx + j] << (8 * j); }
                    g = make_float4(gv[0], gv[1], gv[2], gv[3]);
                }
            }
            rg[i] = g;
            rt_[i] = t;
        }
        const float* img = X + (size_t)n * H * W;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s2 = tid + i * DPP_THREADS;
            float v = 0.0f;
            if (s2 < LX * LX) {
                const int hy = s2 / LX, hx = s2 - hy * LX;
                const int y = cy0 + hy - PAD, x = cx0 + hx - PAD;
                if (y >= 0 && y < H && x >= 0 && x < W) v = img[(size_t)y * W + x];
            }
            rh[i] = v;
        }
    };
    auto commit_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int item = tid + i * DPP_THREADS, p = item >> 3, oq = item & 7;
            const float4 g = rg[i];
            const unsigned t = rt_[i];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float4 m;
                m.x = ((t >> a) & 1u) ? g.x : 0.0f;
                m.y = ((t >> (8 + a)) & 1u) ? g.y : 0.0f;
                m.z = ((t >> (16 + a)) & 1u) ? g.z : 0.0f;
                m.w = ((t >> (24 + a)) & 1u) ? g.w : 0.0f;
                *reinterpret_cast<float4*>(&gm[a * GM_A + p * 32 + oq * 4]) = m;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s2 = tid + i * DPP_THREADS;
            if (s2 < LX * LX) xs[(s2 / LX) * LXW + s2 % LX] = rh[i];
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tile0 = blockIdx.x * tiles_per_block;
    int ntiles = total_tiles - tile0;
    if (ntiles > tiles_per_block) ntiles = tiles_per_block;
    if (ntiles > 0) load_tile(tile0);
    for (int ti = 0; ti < ntiles; ++ti) {
        commit_tile();
        __syncthreads();
        if (ti + 1 < ntiles) load_tile(tile0 + ti + 1);          // in flight under the MFMAs below
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            const int p = wave * 16 + sidx;
            const int base = (2 * (p >> 3)) * LXW + 2 * (p & 7);
            const float a0 = gm[kq * GM_A + p * 32 + l15], a1 = gm[kq * GM_A + p * 32 + 16 + l15];
            const float b0 = okB[0] ? xs[base + offB[0]] : 0.0f;
            const float b1 = okB[1] ? xs[base + offB[1]] : 0.0f;
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: col (tap) = lane & 15, row (o) = (lane >> 4) * 4 + r
    float* red = gm;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * 1024 + (rt * 16 + kq * 4 + r) * 32 + ct * 16 + l15] = acc[rt][ct][r];
    __syncthreads();
    for (int s2 = tid; s2 < Co * NTAP; s2 += DPP_THREADS) {
        const int oo = s2 / NTAP, k = s2 - oo * NTAP;
        const int at = oo * 32 + k;
        partial[(size_t)blockIdx.x * Co * NTAP + s2] = ((red[at] + red[1024 + at]) + red[2048 + at]) + red[3072 + at];
    }
}

}  // namespace

extern "C" int dpp_stem_fwd(const float* X, int N, int H, int W, const float* Wk, const float* bias, int Co, float* Y, uint8_t* argmax,
                            float* stats, int store,
                            dpp_stream_t stream) {
    if (!X || !Wk || !bias || !Y || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || Co < 1 || Co > 32 || (store & ~DPP_ST_C)) return DPP_E_BADARG;
    const int y16 = (store & DPP_ST_C) ? 1 : 0;
    if (stats && ((H % TC) || (W % TC))) return DPP_E_BADARG;     // per-tile statistics assume full 16x16 conv tiles
    int tiles_x = dpp_cdiv(W, TC), tiles_y = dpp_cdiv(H, TC);
    dim3 grid(tiles_x * tiles_y * N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Co <= 16)
        DPP_LAUNCH((stem_fwd_kernel<1>), grid, dim3(DPP_THREADS), 0, st, X, N, H, W, Wk, bias, Co, Y, argmax, tiles_x, tiles_y, stats, y16);
    else
        DPP_LAUNCH((stem_fwd_kernel<2>), grid, dim3(DPP_THREADS), 0, st, X, N, H, W, Wk, bias, Co, Y, argmax, tiles_x, tiles_y, stats, y16);
    return dpp_launch_status();
}

extern "C" int dpp_stem_wgrad_blocks(int N, int H, int W, int tiles_per_block) {
    int total = dpp_cdiv(W, TC) * dpp_cdiv(H, TC) * N;
    return dpp_cdiv(total, tiles_per_block > 0 ? tiles_per_block : 1);
}

extern "C" int dpp_stem_wgrad(const float* X, int N, int H, int W, const float* dY, const uint8_t* argmax, int Co, float* partial,
                              int tiles_per_block, dpp_stream_t stream) {
    if (!X || !dY || !argmax || !partial || N < 1 || (H & 1) || (W & 1) || Co < 1 || Co > 32 || tiles_per_block < 1)
        return DPP_E_BADARG;
    int tiles_x = dpp_cdiv(W, TC), tiles_y = dpp_cdiv(H, TC);
    int total = tiles_x * tiles_y * N;
    DPP_LAUNCH(stem_wgrad_kernel, dim3(dpp_cdiv(total, tiles_per_block)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), X, N,
                       H, W, dY, argmax, Co, partial, tiles_x, tiles_y, tiles_per_block, total);
    return dpp_launch_status();
}
