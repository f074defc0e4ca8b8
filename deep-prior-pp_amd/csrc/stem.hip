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
                                                               float* __restrict__ stats) {
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
                    Y[o + col] = best + bias[col];
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

// thread = (o = tid % Co', pixel group); accumulates the 25 taps in registers.  The kernel is bound by instruction issue (one
// LDS read per FMA when every tap is fetched separately, and the constant background of a depth crop ties all four window
// pixels, i.e. four passes over the taps), so the 6x6 input patch under a pooling window is read ONCE into registers with
// 8-byte LDS reads (18 instead of up to 100) and the tied window pixels take their taps from there.
constexpr int LXW = LX + 2;        // 22: even row pitch, so a window's patch rows start 8-byte aligned

__global__ __launch_bounds__(DPP_THREADS) void stem_wgrad_kernel(const float* __restrict__ X, int N, int H, int W, const float* __restrict__ dY,
                                                                 const uint8_t* __restrict__ arg, int Co, float* __restrict__ partial,
                                                                 int tiles_x, int tiles_y, int tiles_per_block, int total_tiles) {
    __shared__ __attribute__((aligned(16))) float xs[LX * LXW];
    __shared__ float red[DPP_THREADS * NTAP];     // 25.6 KB
    const int tid = threadIdx.x;
    const int o = tid % Co, pg = tid / Co, npg = DPP_THREADS / Co;
    const int Hp = H >> 1, Wp = W >> 1;
    float acc[NTAP];
#pragma unroll
    for (int k = 0; k < NTAP; ++k) acc[k] = 0.0f;
    for (int ti = 0; ti < tiles_per_block; ++ti) {
        int tile = blockIdx.x * tiles_per_block + ti;
        if (tile >= total_tiles) break;
        const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int cy0 = by * TC, cx0 = bx * TC;
        const float* img = X + (size_t)n * H * W;
        // pooled gradient and tie mask of pooled output p (0 when p is outside the tile / the map)
        auto fetch = [&](int p, float& g, int& ties) {
            const int py = (cy0 >> 1) + (p >> 3), px = (cx0 >> 1) + (p & 7);
            const bool ok = p < 64 && py < Hp && px < Wp;
            const size_t idx = ok ? (((size_t)n * Hp + py) * Wp + px) * Co + o : 0;
            g = ok ? dY[idx] : 0.0f;
            ties = ok ? (int)arg[idx] : 0;
        };
        float g_next = 0.0f;
        int t_next = 0;
        if (pg < npg) fetch(pg, g_next, t_next);
        __syncthreads();
        for (int s = tid; s < LX * LX; s += DPP_THREADS) {
            int hy = s / LX, hx = s - hy * LX;
            int y = cy0 + hy - PAD, x = cx0 + hx - PAD;
            xs[hy * LXW + hx] = (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.0f;
        }
        __syncthreads();
        if (pg < npg) {
#pragma unroll 1
            for (int p = pg; p < 64; p += npg) {          // 8x8 pooled outputs of the tile
                const float g = g_next;
                const int ties = t_next;
                fetch(p + npg, g_next, t_next);           // the next output's loads fly under this one's FMAs
                if (ties == 0) continue;
                const int ply = p >> 3, plx = p & 7;
                float patch[6][6];
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c2 = 0; c2 < 3; ++c2) {
                        const float2 v = *reinterpret_cast<const float2*>(&xs[(2 * ply + r) * LXW + 2 * plx + 2 * c2]);
                        patch[r][2 * c2] = v.x;
                        patch[r][2 * c2 + 1] = v.y;
                    }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (!((ties >> a) & 1)) continue;
#pragma unroll
                    for (int k = 0; k < NTAP; ++k) acc[k] += g * patch[(a >> 1) + k / KS][(a & 1) + k % KS];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NTAP; ++k) red[tid * NTAP + k] = acc[k];
    __syncthreads();
    for (int s = tid; s < Co * NTAP; s += DPP_THREADS) {
        int oo = s / NTAP, k = s - oo * NTAP;
        float sum = 0.0f;
        for (int j = 0; j < npg; ++j) sum += red[(j * Co + oo) * NTAP + k];
        partial[(size_t)blockIdx.x * Co * NTAP + s] = sum;
    }
}

}  // namespace

extern "C" int dpp_stem_fwd(const float* X, int N, int H, int W, const float* Wk, const float* bias, int Co, float* Y, uint8_t* argmax,
                            float* stats,
                            dpp_stream_t stream) {
    if (!X || !Wk || !bias || !Y || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || Co < 1 || Co > 32) return DPP_E_BADARG;
    if (stats && ((H % TC) || (W % TC))) return DPP_E_BADARG;     // per-tile statistics assume full 16x16 conv tiles
    int tiles_x = dpp_cdiv(W, TC), tiles_y = dpp_cdiv(H, TC);
    dim3 grid(tiles_x * tiles_y * N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Co <= 16)
        hipLaunchKernelGGL((stem_fwd_kernel<1>), grid, dim3(DPP_THREADS), 0, st, X, N, H, W, Wk, bias, Co, Y, argmax, tiles_x, tiles_y, stats);
    else
        hipLaunchKernelGGL((stem_fwd_kernel<2>), grid, dim3(DPP_THREADS), 0, st, X, N, H, W, Wk, bias, Co, Y, argmax, tiles_x, tiles_y, stats);
    return dpp_launch_status();
}

extern "C" int dpp_stem_wgrad_blocks(int N, int H, int W, int tiles_per_block) {
    int total = dpp_cdiv(W, TC) * dpp_cdiv(H, TC) * N;
    return dpp_cdiv(total, tiles_per_block > 0 ? tiles_per_block : 1);
}

extern "C" int dpp_stem_wgrad(const float* X, int N, int H, int W, const float* dY, const uint8_t* argmax, int Co, float* partial,
                              int tiles_per_block, dpp_stream_t stream) {
    if (!X || !dY || !argmax || !partial || N < 1 || (H & 1) || (W & 1) || Co < 1 || Co > 32 || (DPP_THREADS % Co) || tiles_per_block < 1)
        return DPP_E_BADARG;
    int tiles_x = dpp_cdiv(W, TC), tiles_y = dpp_cdiv(H, TC);
    int total = tiles_x * tiles_y * N;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(dpp_cdiv(total, tiles_per_block)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), X, N,
                       H, W, dY, argmax, Co, partial, tiles_x, tiles_y, tiles_per_block, total);
    return dpp_launch_status();
}
