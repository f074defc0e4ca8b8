// conv3x3_wgrad_t.hip -- filter gradient of the 3x3 'half' convolutions of the 16- / 32-channel layers (stages 1-2), round 6.
//
// Reference arithmetic: T.grad of conv2d(border_mode='half') with respect to the filters, /root/reference/src/net/convlayer.py:230-240
// through /root/reference/src/trainer/poseregnettrainer.py:110-111:   dW[o][tap][c] = sum_p dY[p][o] * act(X)[p + tap][c]
// (act = the BatchNorm + ReLU in front of the convolution, zero padding applied after it).
//
// What was wrong with conv3x3_wgrad_kernel on these layers (37 / 25 us for 0.6 GFLOP = 0.10 of the f32 MFMA roof,
// 155 / 102 us at 256x256): its LDS images are pixel-major like memory, [pixel][channel], so a lane of the MFMA -- which needs ONE
// channel of FOUR pixels per operand -- fetched every operand of every k-step with its own ds_read_b32 and its own address
// arithmetic: 17 VALU instructions per MFMA, one LDS instruction per operand.
//
// Here the images are TRANSPOSED on the way into LDS: Xt[channel][halo row][x] and Yt[channel][row][x], pixel-contiguous.  The
// reduction of one MFMA group runs over the 16 pixels of an image row of the tile; lane (i, kq) owns channel i and the four
// neighbouring pixels x = 4 kq .. 4 kq + 3, component e of its 16-byte read is the operand of k-step e -- the same assignment of
// pixels to (lane, k-step) on the dY and on the X side, which is all an MFMA reduction needs.  A tap is the same read at a shifted
// position: dy moves a whole row (pitch 24 floats: aligned), dx = -1 / +1 is the aligned read rotated by one component plus ONE
// extra dword from the neighbouring run (x = 4 kq - 1 or 4 kq + 4).  Per 16 pixels a wave issues 1 + 3 ds_read_b128 and 6
// ds_read_b32 for 36 MFMAs (the old kernel: 72 ds_read_b32), and no address arithmetic inside the tap loop.
//
// A workgroup walks tiles of 8 x 16 pixels keeping its accumulators in registers (one partial slice per workgroup, as before); the
// next tile's global loads are in flight under the MFMAs of the current one and are transposed into LDS with the prologue when
// they are committed.  16 channels: the four waves split the rows of a tile and meet once, at the end, in an LDS tree; 32 channels:
// wave = one of the 2 x 2 (o, c) tiles.  Partials layout and grid are those of conv3x3_wgrad_kernel (conv3x3.hip: wgrad_geometry):
// [slice][Co][9][Ci], blockIdx.y = tap group.  Sums run in a fixed order: bit-identical replays.
//
// PB (bf16 MFMA operands, BASELINE config 5): the images hold bfloat16 (dY exact when it is bf16-stored, act(X) rounded RNE after
// the prologue), a group is TWO image rows = 32 pixels, lane (i, kq) owns the 8 pixels x = 8 (kq & 1) .. + 7 of row 2 g + (kq >> 1)
// as one 16-byte read feeding v_mfma_f32_16x16x32_bf16; dx = -1 / +1 is a funnel shift of that register by one element with the
// neighbouring element shifted in.
#include <stdlib.h>
#include "dpp_common.h"
#include "conv3x3_wgrad_t.h"

namespace {

template <class T> struct W3Raw;
template <> struct W3Raw<float> {
    typedef float4 type;
    __device__ static __forceinline__ type ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ float4 widen(const type& r) { return r; }
};
template <> struct W3Raw<dpp_bf16> {
    typedef uint2 type;
    __device__ static __forceinline__ type ld(const dpp_bf16* p) { return *reinterpret_cast<const uint2*>(p); }
    __device__ static __forceinline__ float4 widen(const type& r) {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
};

struct Wgrad3TArgs {
    const void* X;        // [N][H][W][C] forward input of the convolution (pre-activation source), float or bf16
    const void* dY;       // [N][H][W][C]
    float* partial;       // [nblk][C][9][C]
    int N, H, W;
    dpp_act act;
    int tiles_x, tiles_y, ntiles;
};

constexpr int W3_TH = 8, W3_TW = 16, W3_HR = W3_TH + 2, W3_HC = W3_TW + 2;
constexpr int W3_RP = 24;                         // floats per halo row of a channel image: x = -4 .. 19, x = 0 at column 4 (16-byte aligned)
constexpr int W3_CPX = W3_HR * W3_RP + 4;         // 244: channel pitch of Xt (== 20 mod 32: the 16 lanes of a ds_read_b128 quarter hit 8 x 4 banks twice)
constexpr int W3_CPY = W3_TH * W3_TW + 4;         // 132: channel pitch of Yt (== 4 mod 32)
// bf16 images: element pitches (a row of 16 pixels is 32 bytes; x = 0 at element 8 so that the aligned 8-pixel runs are 16-byte aligned)
constexpr int W3_RPB = 32;                        // bf16 elements per halo row: x = -8 .. 23
constexpr int W3_CPXB = W3_HR * W3_RPB + 8;       // 328 elements = 164 dwords (== 4 mod 32 dwords)
constexpr int W3_CPYB = W3_TH * W3_TW + 8;        // 136 elements = 68 dwords (== 4 mod 32)

__device__ __forceinline__ float w3_get(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// the operand of k-step e for the tap at dx = dxi - 1, from the aligned run b0 (x = 4 kq .. 4 kq + 3) and its two neighbours
__device__ __forceinline__ float w3_tapval(const float4& b0, float bm1, float bp4, int dxi, int e) {
    if (dxi == 1) return w3_get(b0, e);
    if (dxi == 0) return e == 0 ? bm1 : w3_get(b0, e - 1);
    return e == 3 ? bp4 : w3_get(b0, e + 1);
}

typedef unsigned w3_u32x4 __attribute__((ext_vector_type(4)));

// 8 bf16 elements (x = x0 .. x0 + 7, element 0 in the low half of .x) shifted by one pixel: dxi = 0 -> x0 - 1 .. x0 + 6 with `lo` (the
// element at x0 - 1, in the low half of its dword) shifted in; dxi = 2 -> x0 + 1 .. x0 + 8 with `hi` (the element at x0 + 8, low half).
__device__ __forceinline__ w3_u32x4 w3_shift8(const w3_u32x4& b, unsigned lo, unsigned hi, int dxi) {
    if (dxi == 1) return b;
    w3_u32x4 o;
    if (dxi == 0) {
        o[0] = (b[0] << 16) | (lo & 0xffffu);
        o[1] = (b[1] << 16) | (b[0] >> 16);
        o[2] = (b[2] << 16) | (b[1] >> 16);
        o[3] = (b[3] << 16) | (b[2] >> 16);
    } else {
        o[0] = (b[0] >> 16) | (b[1] << 16);
        o[1] = (b[1] >> 16) | (b[2] << 16);
        o[2] = (b[2] >> 16) | (b[3] << 16);
        o[3] = (b[3] >> 16) | (hi << 16);
    }
    return o;
}

__device__ __forceinline__ unsigned short w3_bf16_bits(float v) {
    const dpp_bf16 h = (dpp_bf16)v;                    // round to nearest even
    return *reinterpret_cast<const unsigned short*>(&h);
}

// OS (64 channels): the OUTPUT channels are split over blockIdx.y in groups of 16 -- a workgroup owns all nine taps of (its 16 dY channels)
// x (all 64 X channels), wave = the X channel tile: nine accumulator tiles per wave, the full X halo but only a quarter of dY staged per
// tile, and one pass over the tiles instead of nine.  (The first 64-channel form gave every tap its own workgroup: nine workgroups staged
// the same 39 KB per tile for 16 MFMAs each -- 59 us per launch at 256 x 256, no faster than the row stream it replaced.)
template <int C, int TPB, class TX, class TY, bool PB, bool OS = false>
__global__ __launch_bounds__(DPP_THREADS) void conv3x3_wgrad_t_kernel(Wgrad3TArgs a) {
    static_assert(!OS || (C == 64 && TPB == 9), "output-channel split: 64 channels, all taps");
    dpp_kernarg_warm<sizeof(Wgrad3TArgs)>();
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* const smem = reinterpret_cast<float*>(smem4);
    constexpr int Q = C / 4;                                  // channel quads per pixel
    constexpr int CY = OS ? 16 : C, QY = CY / 4;              // dY channels / quads a workgroup stages
    constexpr int NXS = W3_HR * W3_HC * Q, SX = (NXS + DPP_THREADS - 1) / DPP_THREADS;      // halo staging slots (16 bytes of f32 each)
    constexpr int NYS = W3_TH * W3_TW * QY, SY = NYS / DPP_THREADS;
    static_assert(DPP_THREADS % Q == 0 && NYS % DPP_THREADS == 0, "slot geometry");
    // f32 images: floats; bf16 images: the same buffer addressed in 2-byte elements
    float* const Xt = smem;
    float* const Yt = smem + C * W3_CPX;
    unsigned short* const Xb = reinterpret_cast<unsigned short*>(smem);
    unsigned short* const Yb = Xb + C * W3_CPXB;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W;
    const TX* const Xg = reinterpret_cast<const TX*>(a.X);
    const TY* const Yg = reinterpret_cast<const TY*>(a.dY);
    const int q4 = (tid % Q) * 4;                             // this thread's channel quad, the same in every slot (256 % Q == 0)
    const int q4y = (tid % QY) * 4, ych0 = OS ? 16 * (int)blockIdx.y : 0;      // ... of dY, and the first dY channel of this workgroup

    // ---- slot geometry: independent of the tile ----
    int xl[SX], xg[SX], xyx[SX];
#pragma unroll
    for (int s = 0; s < SX; ++s) {
        const int slot = tid + s * DPP_THREADS;
        const int hp = slot / Q, hy = hp / W3_HC, hx = hp - hy * W3_HC;
        xl[s] = PB ? (q4 * W3_CPXB + hy * W3_RPB + hx + 7) : (q4 * W3_CPX + hy * W3_RP + hx + 3);
        xg[s] = ((hy - 1) * W + (hx - 1)) * C + q4;
        xyx[s] = slot < NXS ? ((hy << 8) | hx) : -1;
    }
    int yl[SY], yg[SY], yyx[SY];
#pragma unroll
    for (int s = 0; s < SY; ++s) {
        const int slot = tid + s * DPP_THREADS;
        const int p = slot / QY, ty = p / W3_TW, tx = p - ty * W3_TW;
        yl[s] = PB ? (q4y * W3_CPYB + ty * W3_TW + tx) : (q4y * W3_CPY + ty * W3_TW + tx);
        yg[s] = (ty * W + tx) * C + ych0 + q4y;
        yyx[s] = (ty << 8) | tx;
    }
    const int mode = a.act.mode;
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), sc = mu, be = mu;
    if (mode & 2) {
        const int c0 = q4 % a.act.cmod;
        mu = *reinterpret_cast<const float4*>(a.act.mean + c0);
        sc = *reinterpret_cast<const float4*>(a.act.scale + c0);
        be = *reinterpret_cast<const float4*>(a.act.beta + c0);
    }

    typename W3Raw<TX>::type rx[SX];
    typename W3Raw<TY>::type ry[SY];
    unsigned inx = 0, iny = 0;                                // which slots of the tile in flight lie inside the image

    // every load of a tile, unconditionally (a branch between a load and its use drains all loads in flight): slots outside the
    // image read the tensor's first elements and are zeroed at the commit
    auto fetch = [&](int tile_id) {
        const int bx = tile_id % a.tiles_x, t = tile_id / a.tiles_x, by = t % a.tiles_y, n = t / a.tiles_y;
        const int y0 = by * W3_TH, x0 = bx * W3_TW;
        const size_t org = (((size_t)n * H + y0) * W + x0) * C;
        inx = 0; iny = 0;
#pragma unroll
        for (int s = 0; s < SX; ++s) {
            const int y = y0 + (xyx[s] >> 8) - 1, x = x0 + (xyx[s] & 255) - 1;
            const bool in = xyx[s] >= 0 && y >= 0 && y < H && x >= 0 && x < W;
            inx |= in ? (1u << s) : 0u;
            rx[s] = W3Raw<TX>::ld(Xg + (in ? (ptrdiff_t)org + xg[s] : (ptrdiff_t)q4));
        }
#pragma unroll
        for (int s = 0; s < SY; ++s) {
            const int y = y0 + (yyx[s] >> 8), x = x0 + (yyx[s] & 255);
            const bool in = y < H && x < W;
            iny |= in ? (1u << s) : 0u;
            ry[s] = W3Raw<TY>::ld(Yg + (in ? (ptrdiff_t)org + yg[s] : (ptrdiff_t)q4y));
        }
    };
    // registers -> transposed LDS images, the prologue applied here (not at the load: the loads stay in flight under the MFMAs)
    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < SX; ++s) {
            float4 v = W3Raw<TX>::widen(rx[s]);
            if (mode & 2) {
                v.x = dpp_fma(v.x - mu.x, sc.x, be.x); v.y = dpp_fma(v.y - mu.y, sc.y, be.y);
                v.z = dpp_fma(v.z - mu.z, sc.z, be.z); v.w = dpp_fma(v.w - mu.w, sc.w, be.w);
            }
            if (mode & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!((inx >> s) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);        // zero padding AFTER the activation
            if (xyx[s] >= 0) {
                if (PB) {
                    unsigned short* d = Xb + xl[s];
                    d[0] = w3_bf16_bits(v.x); d[W3_CPXB] = w3_bf16_bits(v.y); d[2 * W3_CPXB] = w3_bf16_bits(v.z); d[3 * W3_CPXB] = w3_bf16_bits(v.w);
                } else {
                    float* d = Xt + xl[s];
                    d[0] = v.x; d[W3_CPX] = v.y; d[2 * W3_CPX] = v.z; d[3 * W3_CPX] = v.w;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < SY; ++s) {
            float4 v = W3Raw<TY>::widen(ry[s]);
            if (!((iny >> s) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (PB) {
                unsigned short* d = Yb + yl[s];
                d[0] = w3_bf16_bits(v.x); d[W3_CPYB] = w3_bf16_bits(v.y); d[2 * W3_CPYB] = w3_bf16_bits(v.z); d[3 * W3_CPYB] = w3_bf16_bits(v.w);
            } else {
                float* d = Yt + yl[s];
                d[0] = v.x; d[W3_CPY] = v.y; d[2 * W3_CPY] = v.z; d[3 * W3_CPY] = v.w;
            }
        }
    };

    // 16 channels: the waves split the row groups of a tile (and meet at the end); 32 channels: a wave owns one (o, c) tile; 64 channels
    // (the 16-wide maps of stages 3-4 at 256 x 256 input): a wave owns the four (o = wave, c) tiles
    constexpr int NTC = (C == 64 && !OS) ? 4 : 1;
    f32x4 acc[TPB][NTC];
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int n = 0; n < NTC; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // `to`: the 16-channel tile of dY inside the STAGED image (OS: the image holds the workgroup's 16 channels only)
    const int to = (C == 16 || OS) ? 0 : (C == 32 ? (wave >> 1) : wave), tc0 = C == 32 ? (wave & 1) : (OS ? wave : 0);
    const int tap0 = OS ? 0 : blockIdx.y * TPB;
    const int dy0 = TPB == 9 ? 0 : tap0 / 3;                  // first halo row offset of the taps this workgroup owns
    const int dx0 = TPB == 1 ? tap0 % 3 : 0;

    int tile = blockIdx.x;
    if (tile < a.ntiles) fetch(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();                                      // the previous tile's images have been read
        commit();
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
        if (!PB) {
            constexpr int NG = C == 16 ? W3_TH / 4 : W3_TH;
#pragma unroll 2
            for (int g = 0; g < NG; ++g) {
                const int r = C == 16 ? wave + 4 * g : g;     // tile row of this group
                const float4 av = *reinterpret_cast<const float4*>(Yt + (to * 16 + l15) * W3_CPY + r * W3_TW + 4 * kq);
#pragma unroll
                for (int n = 0; n < NTC; ++n) {
                const float* xb = Xt + ((tc0 + n) * 16 + l15) * W3_CPX + (r + dy0) * W3_RP + 4 + 4 * kq;
#pragma unroll
                for (int dyi = 0; dyi < (TPB == 9 ? 3 : 1); ++dyi) {
                    const float* b = xb + dyi * W3_RP;
                    const float4 b0 = *reinterpret_cast<const float4*>(b);
                    const float bm1 = b[-1], bp4 = b[4];
                    if (TPB == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float bv = dx0 == 1 ? w3_get(b0, e) : (dx0 == 0 ? w3_tapval(b0, bm1, bp4, 0, e) : w3_tapval(b0, bm1, bp4, 2, e));
                            acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3_get(av, e), bv, acc[0][n], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[dyi * 3 + dxi][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3_get(av, e), w3_tapval(b0, bm1, bp4, dxi, e),
                                                                                             acc[dyi * 3 + dxi][n], 0, 0, 0);
                    }
                }
                }
            }
        } else {
            // bf16 operands: a group = two tile rows (32 pixels); lane (i, kq) owns x = 8 (kq & 1) .. + 7 of row 2 g + (kq >> 1)
            constexpr int NG = C == 16 ? 1 : W3_TH / 2;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int r = (C == 16 ? 2 * wave : 2 * g) + (kq >> 1), xo = 8 * (kq & 1);
                const w3_u32x4 av = *reinterpret_cast<const w3_u32x4*>(Yb + (to * 16 + l15) * W3_CPYB + r * W3_TW + xo);
#pragma unroll
                for (int n = 0; n < NTC; ++n) {
                const unsigned short* xb = Xb + ((tc0 + n) * 16 + l15) * W3_CPXB + (r + dy0) * W3_RPB + 8 + xo;
#pragma unroll
                for (int dyi = 0; dyi < (TPB == 9 ? 3 : 1); ++dyi) {
                    const unsigned short* b = xb + dyi * W3_RPB;
                    const w3_u32x4 b0 = *reinterpret_cast<const w3_u32x4*>(b);
                    const unsigned lo = b[-1], hi = b[8];
                    if (TPB == 1) {
                        const w3_u32x4 bv = dx0 == 1 ? b0 : (dx0 == 0 ? w3_shift8(b0, lo, hi, 0) : w3_shift8(b0, lo, hi, 2));
                        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dpp_bf16x8, av), __builtin_bit_cast(dpp_bf16x8, bv), acc[0][n], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int dxi = 0; dxi < 3; ++dxi) {
                            const w3_u32x4 bv = w3_shift8(b0, lo, hi, dxi);
                            acc[dyi * 3 + dxi][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dpp_bf16x8, av), __builtin_bit_cast(dpp_bf16x8, bv),
                                                                                            acc[dyi * 3 + dxi][n], 0, 0, 0);
                        }
                    }
                }
                }
            }
        }
    }

    // ---- 16 channels: the four waves' row shares meet in an LDS tree (fixed order) ----
    if (C == 16) {
        f32x4* const red = reinterpret_cast<f32x4*>(smem);                 // [2][TPB][64] accumulator quads = 18 KB at TPB = 9
        __syncthreads();
        if (wave >= 2) {
#pragma unroll
            for (int t = 0; t < TPB; ++t) red[((wave - 2) * TPB + t) * 64 + lane] = acc[t][0];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int t = 0; t < TPB; ++t) acc[t][0] += red[(wave * TPB + t) * 64 + lane];
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int t = 0; t < TPB; ++t) red[t * 64 + lane] = acc[t][0];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < TPB; ++t) acc[t][0] += red[t * 64 + lane];
        }
    }
    if (C != 16 || wave == 0) {
        float* out = a.partial + (size_t)blockIdx.x * C * 9 * C;
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
            const int tap = tap0 + t;
#pragma unroll
            for (int n = 0; n < NTC; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (OS ? (int)blockIdx.y : to) * 16 + kq * 4 + r, c = (tc0 + n) * 16 + l15;
                    out[((size_t)o * 9 + tap) * C + c] = acc[t][n][r];
                }
        }
    }
}

template <int C, int TPB, bool PB>
int w3t_launch_typed(const Wgrad3TArgs& a, dim3 grid, size_t lds, bool x16, bool y16, hipStream_t st) {
#define DPP_W3T(TX_, TY_) do { DPP_LAUNCH((conv3x3_wgrad_t_kernel<C, TPB, TX_, TY_, PB>), grid, dim3(DPP_THREADS), lds, st, a); return dpp_launch_status(); } while (0)
    if (x16) { if (y16) DPP_W3T(dpp_bf16, dpp_bf16); else DPP_W3T(dpp_bf16, float); }
    if (y16) DPP_W3T(float, dpp_bf16);
    DPP_W3T(float, float);
#undef DPP_W3T
}

template <int C, bool PB>
int w3t_launch_taps(const Wgrad3TArgs& a, int taps_pb, dim3 grid, size_t lds, bool x16, bool y16, hipStream_t st) {
    if (taps_pb == 9) return w3t_launch_typed<C, 9, PB>(a, grid, lds, x16, y16, st);
    if (taps_pb == 3) return w3t_launch_typed<C, 3, PB>(a, grid, lds, x16, y16, st);
    return w3t_launch_typed<C, 1, PB>(a, grid, lds, x16, y16, st);
}

// 64 channels: output channels split over blockIdx.y, all nine taps per workgroup; the float32 images are 71 KB (above the default 64 KB
// window: opt in per instantiation)
template <bool PB>
int w3t_launch_c64(const Wgrad3TArgs& a, dim3 grid, size_t lds, bool x16, bool y16, hipStream_t st) {
#define DPP_W3T64(TX_, TY_) do { \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_t_kernel<64, 9, TX_, TY_, PB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        DPP_LAUNCH((conv3x3_wgrad_t_kernel<64, 9, TX_, TY_, PB, true>), grid, dim3(DPP_THREADS), lds, st, a); return dpp_launch_status(); } while (0)
    if (x16) { if (y16) DPP_W3T64(dpp_bf16, dpp_bf16); else DPP_W3T64(dpp_bf16, float); }
    if (y16) DPP_W3T64(float, dpp_bf16);
    DPP_W3T64(float, float);
#undef DPP_W3T64
}

}  // namespace

bool dpp_conv3x3_wgrad_t_ok(int N, int H, int W, int Ci, int Co, const dpp_act* act) {
    static const bool on = []() { const char* e = getenv("DPP_WGRAD3_T"); return !(e && e[0] == '0'); }();
    if (!on || Ci != Co || (Ci != 16 && Ci != 32 && Ci != 64) || W < 12 || H < 4) return false;
    if ((long)N * H * W * Ci >= (1L << 31)) return false;                      // 32-bit element offsets inside the kernel
    if (act && ((act->mode & 4) || (act->mode && (act->cmod & 3)))) return false;
    return true;
}

int dpp_conv3x3_wgrad_t_launch(const float* X, int N, int H, int W, int C, const dpp_act* act, const float* dY, float* partial, int nblk,
                               int taps_pb, int store, int precision, hipStream_t st) {
    Wgrad3TArgs a;
    a.X = X; a.dY = dY; a.partial = partial; a.N = N; a.H = H; a.W = W;
    if (act) a.act = *act; else { a.act = dpp_act{}; a.act.mode = 0; a.act.cmod = 4; }
    a.tiles_x = dpp_cdiv(W, W3_TW); a.tiles_y = dpp_cdiv(H, W3_TH);
    a.ntiles = a.tiles_x * a.tiles_y * N;
    const bool x16 = (store & DPP_ST_A) != 0, y16 = (store & DPP_ST_B) != 0;
    const dim3 grid(nblk, 9 / taps_pb);
    size_t lds = precision ? (size_t)C * (W3_CPXB + W3_CPYB) * 2 : (size_t)C * (W3_CPX + W3_CPY) * sizeof(float);
    const size_t red = C == 16 ? (size_t)2 * taps_pb * 64 * 16 : 0;
    if (lds < red) lds = red;
    if (C == 64) {
        // all nine taps per workgroup, the 64 output channels in four groups over blockIdx.y (whatever tap split the geometry suggests)
        const dim3 g64(nblk, 4);
        const size_t l64 = precision ? (size_t)(64 * W3_CPXB + 16 * W3_CPYB) * 2 : (size_t)(64 * W3_CPX + 16 * W3_CPY) * sizeof(float);
        return precision ? w3t_launch_c64<true>(a, g64, l64, x16, y16, st) : w3t_launch_c64<false>(a, g64, l64, x16, y16, st);
    }
    if (precision) return C == 16 ? w3t_launch_taps<16, true>(a, taps_pb, grid, lds, x16, y16, st) : w3t_launch_taps<32, true>(a, taps_pb, grid, lds, x16, y16, st);
    return C == 16 ? w3t_launch_taps<16, false>(a, taps_pb, grid, lds, x16, y16, st) : w3t_launch_taps<32, false>(a, taps_pb, grid, lds, x16, y16, st);
}
