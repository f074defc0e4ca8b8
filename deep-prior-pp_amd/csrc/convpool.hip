// convpool.hip -- generic ConvPoolLayer kernels for gfx950 (the DeepPose-style PoseRegNet front end).
//
// Reference arithmetic: /root/reference/src/net/convpoollayer.py:251-282 -- conv2d ('valid' or 'half'), non-overlapping
// max-pool with ignore_border, bias AFTER pooling; the activation belongs to the consumer's operand prologue.
// /root/reference/src/net/poseregnet.py:62-78 builds 5x5 valid / pool 4, 5x5 valid / pool 2, 3x3 valid / no pool with 8
// filters each: 12.8 MFLOP per sample in total, of which none is GEMM-shaped enough for the matrix cores (N = 8), so
// these are VALU kernels: a workgroup owns an 8x8 patch of pooled outputs, stages the activated input patch once in
// LDS, and every wave computes one group of output channels so that all weight reads are wave-uniform scalar loads.
//
// The pooling tie mask (bit j = window element j equals the maximum) follows Theano's MaxPoolGrad, which gives the
// gradient to EVERY element equal to the maximum -- whole windows tie on the constant far-plane background of a crop.
#include "dpp_common.h"

namespace {

constexpr int TP = 8;                   // pooled outputs per tile side
constexpr int WG_WAVES = DPP_THREADS / DPP_WAVE;

struct cp_geom {
    int N, H, W, Ci, kh, kw, pad, Co, pool, Hp, Wp, tiles_x, tiles_y, sy, sx;
};

__host__ inline bool cp_make_geom(cp_geom& g, int N, int H, int W, int Ci, int kh, int kw, int pad, int Co, int pool) {
    if (N < 1 || H < 1 || W < 1 || Ci < 1 || Ci > 32 || Co < 1 || Co > 32 || kh < 1 || kw < 1 || kh > 7 || kw > 7 || pad < 0 ||
        pad > 3 || pool < 1 || pool > 4)
        return false;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.kh = kh; g.kw = kw; g.pad = pad; g.Co = Co; g.pool = pool;
    int Hc = H + 2 * pad - kh + 1, Wc = W + 2 * pad - kw + 1;
    if (Hc < pool || Wc < pool) return false;
    g.Hp = Hc / pool; g.Wp = Wc / pool;
    g.tiles_y = dpp_cdiv(g.Hp, TP); g.tiles_x = dpp_cdiv(g.Wp, TP);
    g.sy = TP * pool + kh - 1; g.sx = TP * pool + kw - 1;
    return true;
}

// Stage the activated input patch of tile (n, ty, tx): xs[y][x][ci], zero outside the image (the conv's zero padding).
__device__ __forceinline__ void cp_stage(float* xs, const float* __restrict__ X, const dpp_act& act, const cp_geom& g, int n, int ty,
                                         int tx) {
    const int y0 = ty * TP * g.pool - g.pad, x0 = tx * TP * g.pool - g.pad;
    const int total = g.sy * g.sx * g.Ci;
    for (int s = threadIdx.x; s < total; s += DPP_THREADS) {
        int ci = s % g.Ci, r = s / g.Ci;
        int hx = r % g.sx, hy = r / g.sx;
        int y = y0 + hy, x = x0 + hx;
        float v = 0.0f;
        if (y >= 0 && y < g.H && x >= 0 && x < g.W) v = dpp_act1(X[(((size_t)n * g.H + y) * g.W + x) * g.Ci + ci], act, ci);
        xs[s] = v;
    }
}

template <int CPT>   // output channels per wave
__global__ __launch_bounds__(DPP_THREADS) void convpool_fwd_kernel(const float* __restrict__ X, dpp_act act, const float* __restrict__ Wk,
                                                                   const float* __restrict__ bias, float* __restrict__ Y,
                                                                   uint16_t* __restrict__ ties, cp_geom g) {
    HIP_DYNAMIC_SHARED(float, xs)
    const int tile = blockIdx.x;
    const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, n = tile / (g.tiles_x * g.tiles_y);
    cp_stage(xs, X, act, g, n, ty, tx);
    __syncthreads();
    const int q = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ply = q >> 3, plx = q & 7;
    const int py = ty * TP + ply, px = tx * TP + plx;
    const int taps = g.kh * g.kw;
    for (int c0 = wave * CPT; c0 < g.Co; c0 += WG_WAVES * CPT) {       // wave-uniform channel group
        float best[CPT];
        int mask[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) { best[c] = -INFINITY; mask[c] = 0; }
        for (int j = 0; j < g.pool * g.pool; ++j) {
            const int wy = j / g.pool, wx = j - wy * g.pool;
            const float* xp = xs + ((size_t)(ply * g.pool + wy) * g.sx + (plx * g.pool + wx)) * g.Ci;
            float acc[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) acc[c] = 0.0f;
            for (int t = 0; t < taps; ++t) {
                const int dy = t / g.kw, dx = t - dy * g.kw;
                const float* xr = xp + ((size_t)dy * g.sx + dx) * g.Ci;
                for (int ci = 0; ci < g.Ci; ++ci) {
                    const float xv = xr[ci];
#pragma unroll
                    for (int c = 0; c < CPT; ++c) {
                        const int co = c0 + c < g.Co ? c0 + c : g.Co - 1;
                        acc[c] = fmaf(xv, Wk[((size_t)co * taps + t) * g.Ci + ci], acc[c]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (acc[c] > best[c]) { best[c] = acc[c]; mask[c] = 1 << j; }
                else if (acc[c] == best[c]) mask[c] |= 1 << j;
            }
        }
        if (py < g.Hp && px < g.Wp) {
            const size_t o = (((size_t)n * g.Hp + py) * g.Wp + px) * g.Co;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                if (c0 + c < g.Co) {
                    Y[o + c0 + c] = best[c] + bias[c0 + c];
                    if (ties) ties[o + c0 + c] = (uint16_t)mask[c];
                }
            }
        }
    }
}

// The same layer for the shapes the nets use (square K x K filter, pool P, at most 2 output channels per wave: PoseRegNet / ScaleNet,
// 8 filters), compile-time K and P.  The generic kernel above re-reads every input value from LDS once per (window position, tap)
// and fetches its weights with scalar loads INSIDE the innermost loop (one s_waitcnt per multiply-add group: 100 us for the 0.28 GFLOP
// second ScaleNet layer at batch 128).  Here a lane keeps the (P + K - 1)^2 input patch of its pooling window in registers per input
// channel and all P * P window positions accumulate from it: (P + K - 1)^2 + 2 K^2 LDS reads for 2 P^2 K^2 multiply-adds (64 + 50
// for 800 at K = 5, P = 4); weights are staged in LDS once per workgroup.  Sums run over (ci, dy, dx) -- ci outermost.
template <int K, int P, int CPT>
__global__ __launch_bounds__(DPP_THREADS) void convpool_fwd_fast_kernel(const float* __restrict__ X, dpp_act act, const float* __restrict__ Wk,
                                                                        const float* __restrict__ bias, float* __restrict__ Y,
                                                                        uint16_t* __restrict__ ties, cp_geom g) {
    HIP_DYNAMIC_SHARED(float, sm)
    constexpr int R = P + K - 1, taps = K * K;
    const int plane = g.sy * g.sx;
    float* xs = sm;                         // [Ci][sy][sx]: channel planes, a patch row is contiguous
    float* ws = sm + g.Ci * plane;          // [Co][taps][Ci]
    const int tile = blockIdx.x;
    const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, n = tile / (g.tiles_x * g.tiles_y);
    {
        const int y0 = ty * TP * P - g.pad, x0 = tx * TP * P - g.pad;
        const int total = plane * g.Ci;
        for (int s = threadIdx.x; s < total; s += DPP_THREADS) {            // global order (pixel-major), planar in LDS
            const int ci = s % g.Ci, r = s / g.Ci;
            const int hx = r % g.sx, hy = r / g.sx;
            const int y = y0 + hy, x = x0 + hx;
            float v = 0.0f;
            if (y >= 0 && y < g.H && x >= 0 && x < g.W) v = dpp_act1(X[(((size_t)n * g.H + y) * g.W + x) * g.Ci + ci], act, ci);
            xs[ci * plane + r] = v;
        }
        for (int s = threadIdx.x; s < g.Co * taps * g.Ci; s += DPP_THREADS) ws[s] = Wk[s];
    }
    __syncthreads();
    const int q = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ply = q >> 3, plx = q & 7;
    const int py = ty * TP + ply, px = tx * TP + plx;
    for (int c0 = wave * CPT; c0 < g.Co; c0 += WG_WAVES * CPT) {       // wave-uniform channel group
        float acc[P * P][CPT];
#pragma unroll
        for (int j = 0; j < P * P; ++j)
#pragma unroll
            for (int c = 0; c < CPT; ++c) acc[j][c] = 0.0f;
        int wbase[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) wbase[c] = (c0 + c < g.Co ? c0 + c : g.Co - 1) * taps * g.Ci;
        for (int ci = 0; ci < g.Ci; ++ci) {
            float patch[R][R];
            const float* xp = xs + ci * plane + (ply * P) * g.sx + plx * P;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < R; ++c) patch[r][c] = xp[r * g.sx + c];
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    float w[CPT];
#pragma unroll
                    for (int c = 0; c < CPT; ++c) w[c] = ws[wbase[c] + (dy * K + dx) * g.Ci + ci];
#pragma unroll
                    for (int wy = 0; wy < P; ++wy)
#pragma unroll
                        for (int wx = 0; wx < P; ++wx)
#pragma unroll
                            for (int c = 0; c < CPT; ++c) acc[wy * P + wx][c] = fmaf(patch[wy + dy][wx + dx], w[c], acc[wy * P + wx][c]);
                }
        }
        if (py < g.Hp && px < g.Wp) {
            const size_t o = (((size_t)n * g.Hp + py) * g.Wp + px) * g.Co;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                float best = -INFINITY;
                int mask = 0;
#pragma unroll
                for (int j = 0; j < P * P; ++j) {
                    if (acc[j][c] > best) { best = acc[j][c]; mask = 1 << j; }
                    else if (acc[j][c] == best) mask |= 1 << j;
                }
                if (c0 + c < g.Co) {
                    Y[o + c0 + c] = best + bias[c0 + c];
                    if (ties) ties[o + c0 + c] = (uint16_t)mask;
                }
            }
        }
    }
}

// Filter gradient: thread = a few (co, tap, ci) elements; workgroups walk tiles keeping their sums in registers and
// write one partial each (fixed order, deterministic).
constexpr int EPT = 8;     // weight elements per thread per pass

__global__ __launch_bounds__(DPP_THREADS) void convpool_wgrad_kernel(const float* __restrict__ X, dpp_act act, const float* __restrict__ dY,
                                                                     const uint16_t* __restrict__ ties, float* __restrict__ partial,
                                                                     cp_geom g, int total_tiles) {
    HIP_DYNAMIC_SHARED(float, xs)
    const int taps = g.kh * g.kw, nW = g.Co * taps * g.Ci;
    const int xs_floats = g.sy * g.sx * g.Ci;
    float* gs = xs + xs_floats;                                  // [64][Co] gradients of the tile
    int* ms = reinterpret_cast<int*>(gs + 64 * g.Co);            // [64][Co] tie masks
    for (int e0 = 0; e0 < nW; e0 += DPP_THREADS * EPT) {
        float acc[EPT];
        int eco[EPT], eoff[EPT];
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            acc[i] = 0.0f;
            int e = e0 + i * DPP_THREADS + threadIdx.x;
            int ee = e < nW ? e : 0;
            int ci = ee % g.Ci, r = ee / g.Ci;
            int t = r % taps;
            eco[i] = e < nW ? r / taps : -1;
            eoff[i] = ((t / g.kw) * g.sx + (t % g.kw)) * g.Ci + ci;
        }
        for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
            const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, n = tile / (g.tiles_x * g.tiles_y);
            __syncthreads();
            cp_stage(xs, X, act, g, n, ty, tx);
            for (int s = threadIdx.x; s < 64 * g.Co; s += DPP_THREADS) {
                int co = s % g.Co, q = s / g.Co;
                int py = ty * TP + (q >> 3), px = tx * TP + (q & 7);
                float gv = 0.0f;
                int m = 0;
                if (py < g.Hp && px < g.Wp) {
                    size_t o = (((size_t)n * g.Hp + py) * g.Wp + px) * g.Co + co;
                    gv = dY[o];
                    m = ties ? (int)ties[o] : 1;
                }
                gs[s] = gv;
                ms[s] = m;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                if (eco[i] < 0) continue;
                float a = acc[i];
                for (int q = 0; q < 64; ++q) {
                    const float gv = gs[q * g.Co + eco[i]];
                    int m = ms[q * g.Co + eco[i]];
                    if (gv == 0.0f || m == 0) continue;
                    const int base = (((q >> 3) * g.pool) * g.sx + (q & 7) * g.pool) * g.Ci + eoff[i];
                    float xsum = 0.0f;
                    for (int j = 0; m != 0; ++j, m >>= 1) {
                        if (m & 1) {
                            int wy = j / g.pool, wx = j - wy * g.pool;
                            xsum += xs[base + (wy * g.sx + wx) * g.Ci];
                        }
                    }
                    a = fmaf(gv, xsum, a);
                }
                acc[i] = a;
            }
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            int e = e0 + i * DPP_THREADS + threadIdx.x;
            if (e < nW) partial[(size_t)blockIdx.x * nW + e] = acc[i];
        }
    }
}

// Data gradient: thread = (input pixel, group of 8 input channels); gathers over taps and output channels, adding only
// where the tie mask routes the pooled gradient to that conv pixel.
__global__ __launch_bounds__(DPP_THREADS) void convpool_dgrad_kernel(const float* __restrict__ dY, const uint16_t* __restrict__ ties,
                                                                     const float* __restrict__ Wk, float* __restrict__ dX, cp_geom g) {
    const size_t pix = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x;
    const size_t npix = (size_t)g.N * g.H * g.W;
    const int c0 = blockIdx.y * 8;
    if (pix >= npix) return;
    const int ix = (int)(pix % g.W), iy = (int)((pix / g.W) % g.H), n = (int)(pix / ((size_t)g.W * g.H));
    const int taps = g.kh * g.kw;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
    for (int t = 0; t < taps; ++t) {
        const int dy = t / g.kw, dx = t - dy * g.kw;
        const int cy = iy + g.pad - dy, cx = ix + g.pad - dx;
        if (cy < 0 || cx < 0 || cy >= g.Hp * g.pool || cx >= g.Wp * g.pool) continue;
        const int py = cy / g.pool, px = cx / g.pool;
        const int j = (cy - py * g.pool) * g.pool + (cx - px * g.pool);
        const size_t o = (((size_t)n * g.Hp + py) * g.Wp + px) * g.Co;
        for (int co = 0; co < g.Co; ++co) {
            const int m = ties ? (int)ties[o + co] : 1;
            if (!((m >> j) & 1)) continue;
            const float gv = dY[o + co];
            const float* w = Wk + ((size_t)co * taps + t) * g.Ci + c0;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c0 + c < g.Ci) acc[c] = fmaf(gv, w[c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c0 + c < g.Ci) dX[pix * g.Ci + c0 + c] = acc[c];
}

inline dpp_act cp_act(const dpp_act* a) {
    dpp_act r = {nullptr, nullptr, nullptr, 0, 1};
    return a ? *a : r;
}

constexpr int CP_WGRAD_MAX_BLOCKS = 512;

}  // namespace

extern "C" int dpp_convpool_fwd(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* Wk, int kh, int kw, int pad,
                                int Co, int pool, const float* bias, float* Y, uint16_t* ties, dpp_stream_t stream) {
    if (act && (act->mode & 4)) return DPP_E_UNSUPPORTED;       // the two-tensor BatchNorm-backward operand is a dpp_gemm feature
    cp_geom g;
    if (!X || !Wk || !bias || !Y || !cp_make_geom(g, N, H, W, Ci, kh, kw, pad, Co, pool)) return DPP_E_BADARG;
    size_t lds = (size_t)g.sy * g.sx * Ci * sizeof(float);
    if (lds > 64 * 1024) return DPP_E_UNSUPPORTED;
    dim3 grid(N * g.tiles_x * g.tiles_y), block(DPP_THREADS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int cpt = dpp_cdiv(Co, WG_WAVES);
    const size_t lds_fast = lds + (size_t)Co * kh * kw * Ci * sizeof(float);
    if (kh == kw && cpt <= 2 && lds_fast <= 64 * 1024) {
#define DPP_CPF(K_, P_) if (kh == K_ && pool == P_) { \
            DPP_LAUNCH((convpool_fwd_fast_kernel<K_, P_, 2>), grid, block, lds_fast, st, X, cp_act(act), Wk, bias, Y, ties, g); \
            return dpp_launch_status(); }
        DPP_CPF(5, 4) DPP_CPF(5, 2) DPP_CPF(5, 1) DPP_CPF(3, 1) DPP_CPF(3, 2)
#undef DPP_CPF
    }
    if (cpt <= 2)
        DPP_LAUNCH((convpool_fwd_kernel<2>), grid, block, lds, st, X, cp_act(act), Wk, bias, Y, ties, g);
    else if (cpt <= 4)
        DPP_LAUNCH((convpool_fwd_kernel<4>), grid, block, lds, st, X, cp_act(act), Wk, bias, Y, ties, g);
    else
        DPP_LAUNCH((convpool_fwd_kernel<8>), grid, block, lds, st, X, cp_act(act), Wk, bias, Y, ties, g);
    return dpp_launch_status();
}

extern "C" int dpp_convpool_wgrad_blocks(int N, int Hp, int Wp) {
    if (N < 1 || Hp < 1 || Wp < 1) return 0;
    long total = (long)N * dpp_cdiv(Hp, TP) * dpp_cdiv(Wp, TP);
    return (int)(total < CP_WGRAD_MAX_BLOCKS ? total : CP_WGRAD_MAX_BLOCKS);
}

extern "C" int dpp_convpool_wgrad(const float* X, int N, int H, int W, int Ci, const dpp_act* act, const float* dY, const uint16_t* ties,
                                  int kh, int kw, int pad, int Co, int pool, float* partial, dpp_stream_t stream) {
    if (act && (act->mode & 4)) return DPP_E_UNSUPPORTED;       // the two-tensor BatchNorm-backward operand is a dpp_gemm feature
    cp_geom g;
    if (!X || !dY || !partial || !cp_make_geom(g, N, H, W, Ci, kh, kw, pad, Co, pool) || (pool > 1 && !ties)) return DPP_E_BADARG;
    size_t lds = ((size_t)g.sy * g.sx * Ci + 2 * 64 * Co) * sizeof(float);
    if (lds > 64 * 1024) return DPP_E_UNSUPPORTED;
    int total = N * g.tiles_x * g.tiles_y;
    int blocks = dpp_convpool_wgrad_blocks(N, g.Hp, g.Wp);
    DPP_LAUNCH(convpool_wgrad_kernel, dim3(blocks), dim3(DPP_THREADS), lds, static_cast<hipStream_t>(stream), X, cp_act(act), dY,
                       ties, partial, g, total);
    return dpp_launch_status();
}

extern "C" int dpp_convpool_dgrad(const float* dY, const uint16_t* ties, int N, int H, int W, int Ci, const float* Wk, int kh, int kw,
                                  int pad, int Co, int pool, float* dX, dpp_stream_t stream) {
    cp_geom g;
    if (!dY || !Wk || !dX || !cp_make_geom(g, N, H, W, Ci, kh, kw, pad, Co, pool) || (pool > 1 && !ties)) return DPP_E_BADARG;
    size_t npix = (size_t)N * H * W;
    dim3 grid((unsigned)((npix + DPP_THREADS - 1) / DPP_THREADS), dpp_cdiv(Ci, 8));
    DPP_LAUNCH(convpool_dgrad_kernel, grid, dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), dY, ties, Wk, dX, g);
    return dpp_launch_status();
}
