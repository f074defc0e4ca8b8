// conv3x3_stream.hip -- the 3x3 'half' convolutions of the NARROW bottlenecks (16 -> 16 and 32 -> 32 channels: stages 1 and 2 of the
// ResNet, resnet.py:365-368 / 394-397 via convlayer.py:230-240) and their data gradients, as a wave-autonomous stream.
//
// 131 072 pixels x (16 -> 16 channels x 9 taps): 8 MB in, 8 MB out, 0.6 GFLOP -- 3-4 us of bandwidth or of f32 MFMA time.  The
// LDS-tiled kernel (conv3x3.hip) takes 15.6 us for it: every workgroup stages a halo tile and nine weight slices, synchronises per
// tap and transposes its accumulators through LDS, all in lockstep with the other 1 023.  Here a WAVE owns TPW tiles of 16
// consecutive pixels of an image row and all CO output channels:
//   * a tap is the same pixel row shifted: lane (i, kq) loads the 16 bytes of channels 4 kq .. 4 kq + 3 (+ 16 g for the wider layer)
//     of pixel i + (dy, dx) -- nine (eighteen) independent 16-byte loads per tile, all issued before the first use, neighbouring
//     taps hitting L1; pixels outside the image read the centre pixel and are zeroed AFTER the BatchNorm + ReLU prologue (zero
//     padding of the activated map), decided by branch-free masks;
//   * the filter -- 9 x C x C floats, 9 KB / 36 KB -- lives in registers for the whole wave (36 / 144 VGPRs): lane (o, kq) holds
//     Wk[o][tap][4 kq .. 4 kq + 3];
//   * MFMA row i carries memory pixel (i & 3) * 4 + (i >> 2) of the tile, so that in the D layout (lane (o, kq), register r = row
//     4 kq + r) one store instruction covers FOUR consecutive pixels x 16 channels = 256 contiguous bytes; bias, the
//     BatchNorm-backward mask and sums (data gradient) and the statistics of the written tensor (forward) work in that layout;
//   * no LDS and no barrier until the column reductions of the workgroup's 64 TPW rows at the very end.
// The data gradient is the same kernel on dY with the mirrored weights of dpp_conv3x3_wtrans.
#include "dpp_common.h"

namespace {

struct C3sArgs {
    const float* X; const float* Wk; const float* bias; float* Y;
    dpp_act act;
    dpp_epilogue epi;
    int N, H, W;
};

// CN = C / 16 (input and output channel tiles), TPW = 16-pixel tiles per wave
template <int CN, int TPW, bool ACT, bool BNB>
__global__ __launch_bounds__(DPP_THREADS) void conv3x3_stream_kernel(C3sArgs a) {
    dpp_kernarg_warm<sizeof(C3sArgs)>();
    constexpr int C = 16 * CN, ROWS = 64 * TPW;
    __shared__ float red[4 * C];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int mrow = (l15 & 3) * 4 + (l15 >> 2);             // memory pixel (within a tile) that MFMA row l15 carries
    const int H = a.H, W = a.W, TX = W >> 4;                  // tiles per image row
    const int tile0 = (blockIdx.x * 4 + wave) * TPW;
    const dpp_epilogue& ep = a.epi;
    const int mode = ACT ? a.act.mode : 0;

    // ---- the filter: lane (o = l15 (+ 16 ct), kq) holds Wk[o][tap][16 g + 4 kq .. + 3] ----
    float4 bw[CN][9][CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < CN; ++g)
                bw[ct][t][g] = *reinterpret_cast<const float4*>(a.Wk + ((size_t)(ct * 16 + l15) * 9 + t) * C + g * 16 + kq * 4);
    // ---- every operand load of the wave's tiles ----
    float4 xa[TPW][9][CN];
    unsigned okm[TPW];                                        // bit t: tap t of this lane's pixel lies inside the image
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        const int tile = tile0 + tl;
        const int x0 = (tile % TX) << 4, row = tile / TX;     // row = n * H + y
        const int y = row % H, x = x0 + mrow;
        const size_t centre = ((size_t)row * W + x) * C + kq * 4;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = ((unsigned)(y + dy) < (unsigned)H) & ((unsigned)(x + dx) < (unsigned)W);
            m |= (ok ? 1u : 0u) << t;
            const size_t o = ok ? centre + (ptrdiff_t)(dy * W + dx) * C : centre;
#pragma unroll
            for (int g = 0; g < CN; ++g) xa[tl][t][g] = *reinterpret_cast<const float4*>(a.X + o + g * 16);
        }
        okm[tl] = m;
    }
    float4 mu[CN], sc[CN], be[CN];
#pragma unroll
    for (int g = 0; g < CN; ++g) {
        mu[g] = sc[g] = be[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mode & 2) {
            mu[g] = *reinterpret_cast<const float4*>(a.act.mean + g * 16 + kq * 4);
            sc[g] = *reinterpret_cast<const float4*>(a.act.scale + g * 16 + kq * 4);
            be[g] = *reinterpret_cast<const float4*>(a.act.beta + g * 16 + kq * 4);
        }
    }
    float cb[CN], cmean[CN], cscale[CN], cbeta[CN], cistd[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) {
        const int col = ct * 16 + l15;
        cb[ct] = a.bias ? a.bias[col] : 0.0f;
        cmean[ct] = BNB ? ep.bn_mean[col] : 0.0f; cscale[ct] = BNB ? ep.bn_scale[col] : 0.0f;
        cbeta[ct] = BNB ? ep.bn_beta[col] : 0.0f; cistd[ct] = BNB ? ep.bn_inv_std[col] : 0.0f;
    }
    float xr[TPW][CN][4];                                     // bn_x at this lane's output elements (data gradient)
    if (BNB) {
#pragma unroll
        for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
            for (int ct = 0; ct < CN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[tl][ct][r] = ep.bn_x[((size_t)(tile0 + tl) * 16 + r * 4 + kq) * C + ct * 16 + l15];
    }
    DPP_SCHED_FENCE();

    float vals[TPW][CN][4];
    float sx[CN], sy[CN];
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) { sx[ct] = 0.0f; sy[ct] = 0.0f; }
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl) {
        f32x4 acc[CN];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const bool in = (okm[tl] >> t) & 1u;
#pragma unroll
            for (int g = 0; g < CN; ++g) {
                float4 v = xa[tl][t][g];
                if (mode & 2) {
                    v.x = dpp_fma(v.x - mu[g].x, sc[g].x, be[g].x); v.y = dpp_fma(v.y - mu[g].y, sc[g].y, be[g].y);
                    v.z = dpp_fma(v.z - mu[g].z, sc[g].z, be[g].z); v.w = dpp_fma(v.w - mu[g].w, sc[g].w, be[g].w);
                }
                if (mode & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                v.x = in ? v.x : 0.0f; v.y = in ? v.y : 0.0f; v.z = in ? v.z : 0.0f; v.w = in ? v.w : 0.0f;      // zero padding AFTER the activation
#pragma unroll
                for (int ct = 0; ct < CN; ++ct) {
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, bw[ct][t][g].x, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, bw[ct][t][g].y, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, bw[ct][t][g].z, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, bw[ct][t][g].w, acc[ct], 0, 0, 0);
                }
            }
        }
        // D layout: this lane holds column ct*16 + l15 of MFMA rows 4 kq + r, i.e. memory pixels 4 r + kq of the tile
#pragma unroll
        for (int ct = 0; ct < CN; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = ((size_t)(tile0 + tl) * 16 + r * 4 + kq) * C + ct * 16 + l15;
                float v = acc[ct][r] + cb[ct];
                if (BNB) {
                    const float dx = xr[tl][ct][r] - cmean[ct];
                    if (ep.bn_relu && dx * cscale[ct] + cbeta[ct] < 0.0f) v = 0.0f;
                    sx[ct] += v;
                    sy[ct] += v * (dx * cistd[ct]);
                }
                a.Y[o] = v;
                vals[tl][ct][r] = v;
            }
    }
    // ---- column reductions over the workgroup's rows (the only barriers of the kernel) ----
    if (BNB && ep.bn_partial != nullptr) {
        dpp_tile_colsum<CN, 4, 1, C>(sx, red, wave, 0, l15, kq);
        dpp_tile_colsum<CN, 4, 1, C>(sy, red, wave, 0, l15, kq);
        if (wave == 0 && kq == 0) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                ep.bn_partial[dpp_partial_index(0, ct * 16 + l15, blockIdx.x, C, gridDim.x)] = sx[ct];
                ep.bn_partial[dpp_partial_index(1, ct * 16 + l15, blockIdx.x, C, gridDim.x)] = sy[ct];
            }
        }
    }
    if (ep.stats != nullptr) {
        float sm[CN], m2[CN];
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            sm[ct] = 0.0f;
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
                for (int r = 0; r < 4; ++r) sm[ct] += vals[tl][ct][r];
        }
        dpp_tile_colsum<CN, 4, 1, C>(sm, red, wave, 0, l15, kq);
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            sm[ct] *= 1.0f / (float)ROWS;
            m2[ct] = 0.0f;
#pragma unroll
            for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dv = vals[tl][ct][r] - sm[ct]; m2[ct] += dv * dv; }
        }
        dpp_tile_colsum<CN, 4, 1, C>(m2, red, wave, 0, l15, kq);
        if (wave == 0 && kq == 0) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) {
                ep.stats[dpp_partial_index(0, ct * 16 + l15, blockIdx.x, C, gridDim.x)] = sm[ct];
                ep.stats[dpp_partial_index(1, ct * 16 + l15, blockIdx.x, C, gridDim.x)] = m2[ct];
            }
        }
    }
}

// (tiles per wave 1 | 2 | 4 measured alike on stage 1: 15.5 | 14.8 | 15.9 us, profiles/r04_conv3x3_stream.txt)
int tiles_per_wave(int C, long pixels) { return (C == 16 && pixels >= 65536) ? 2 : 1; }

}  // namespace

// Rows (pixels) per workgroup = per statistics block of dpp_conv3x3_stream for this layer, or 0: the kernel takes C == Ci == Co in
// {16, 32}, W % 16 == 0 and a pixel count that fills whole workgroups; the LDS-tiled dpp_conv3x3 remains for everything else.
extern "C" int dpp_conv3x3_stream_rows(int N, int H, int W, int C) {
    if ((C != 16 && C != 32) || N < 1 || H < 1 || W < 16 || (W & 15)) return 0;
    const long px = (long)N * H * W;
    const int rows = 64 * tiles_per_wave(C, px);
    if (px % rows || px * C * 4 > 0x7fffffffL * 4L) return 0;
    return rows;
}

extern "C" int dpp_conv3x3_stream(const float* X, int N, int H, int W, int C, const dpp_act* act, const float* Wk, const float* bias,
                                  float* Y, const dpp_epilogue* epi, dpp_stream_t stream) {
    if (!X || !Wk || !Y) return DPP_E_BADARG;
    const int rows = dpp_conv3x3_stream_rows(N, H, W, C);
    if (!rows) return DPP_E_UNSUPPORTED;
    C3sArgs a;
    a.X = X; a.Wk = Wk; a.bias = bias; a.Y = Y; a.N = N; a.H = H; a.W = W;
    a.act.mean = a.act.scale = a.act.beta = nullptr; a.act.mode = 0; a.act.cmod = C; a.act.x2 = a.act.aux = nullptr; a.act.out = nullptr;
    if (act) a.act = *act;
    if (a.act.mode & ~3) return DPP_E_UNSUPPORTED;
    if ((a.act.mode & 2) && !(a.act.mean && a.act.scale && a.act.beta && a.act.cmod == C)) return DPP_E_BADARG;
    a.epi = epi ? *epi : dpp_epilogue{};
    if (a.epi.bn_x && !(a.epi.bn_mean && a.epi.bn_inv_std && a.epi.bn_scale && a.epi.bn_beta && a.epi.bn_partial)) return DPP_E_BADARG;
    if (a.epi.bn_x && a.epi.stats) return DPP_E_UNSUPPORTED;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(X) || !al16(Wk) || !al16(a.act.mean) || !al16(a.act.scale) || !al16(a.act.beta)) return DPP_E_UNSUPPORTED;
    const long px = (long)N * H * W;
    const dim3 grid((unsigned)(px / rows)), block(DPP_THREADS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool actv = a.act.mode != 0, bnb = a.epi.bn_x != nullptr;
#define DPP_C3S(CN_, T_) do { \
        if (actv) { if (bnb) DPP_LAUNCH((conv3x3_stream_kernel<CN_, T_, true, true>), grid, block, 0, st, a); \
                    else DPP_LAUNCH((conv3x3_stream_kernel<CN_, T_, true, false>), grid, block, 0, st, a); } \
        else { if (bnb) DPP_LAUNCH((conv3x3_stream_kernel<CN_, T_, false, true>), grid, block, 0, st, a); \
               else DPP_LAUNCH((conv3x3_stream_kernel<CN_, T_, false, false>), grid, block, 0, st, a); } } while (0)
    if (C == 16) { if (rows == 128) DPP_C3S(1, 2); else DPP_C3S(1, 1); }
    else DPP_C3S(2, 1);
#undef DPP_C3S
    return dpp_launch_status();
}
