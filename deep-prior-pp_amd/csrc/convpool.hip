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
        constexpr int SU = 4;                   // loads of SU elements are requested before the first is used: a load behind a
        for (int s0 = threadIdx.x; s0 < total; s0 += DPP_THREADS * SU) {     // branch per element is one memory round trip EACH
            float raw[SU];
            bool ok[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {      // global order (pixel-major), planar in LDS
                const int s = s0 + u * DPP_THREADS;
                const int ci = s % g.Ci, r = s / g.Ci;
                const int hx = r % g.sx, hy = r / g.sx;
                const int y = y0 + hy, x = x0 + hx;
                ok[u] = (s < total) & (y >= 0) & (y < g.H) & (x >= 0) & (x < g.W);
                raw[u] = X[ok[u] ? (((size_t)n * g.H + y) * g.W + x) * g.Ci + ci : 0];
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int s = s0 + u * DPP_THREADS;
                const int ci = s % g.Ci, r = s / g.Ci;
                if (s < total) xs[ci * plane + r] = ok[u] ? dpp_act1(raw[u], act, ci) : 0.0f;
            }
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

// The filter gradient as a dense correlation over the UNPOOLED gradient (see convpool_dgrad_fast_kernel): per tile of T x T conv
// pixels the activated input halo xs[ci][T + k - 1][T + k - 1] and gc[pixel][8 output channels] are staged in LDS; a thread owns one
// (tap, ci) and the 8 output channels of blockIdx.y, walks the tile's pixels -- one input value, two 16-byte gradient broadcasts, eight
// multiply-adds -- and keeps its sums in registers across the tiles its workgroup walks.  With fewer than 128 (tap, ci) pairs the
// threads form several groups that split the pixels and combine through LDS in group order.  The kernel above spends its time in
// per-thread loops over tie-mask bits (36-39 us for PoseRegNet's layers at batch 128).
__global__ __launch_bounds__(DPP_THREADS) void convpool_wgrad_dense_kernel(const float* __restrict__ X, dpp_act act, const float* __restrict__ dY,
                                                                           const uint16_t* __restrict__ ties, float* __restrict__ partial,
                                                                           cp_geom g, int T, int tiles_x, int tiles_y, int total_tiles) {
    HIP_DYNAMIC_SHARED(float, sm)
    const int taps = g.kh * g.kw, TC = taps * g.Ci, nW = g.Co * TC;
    const int S = T + g.kh - 1, SX = T + g.kw - 1;
    float* xs = sm;                               // [Ci][S][SX]
    float* gcs = sm + g.Ci * S * SX;              // [T * T][8]
    const int groups = DPP_THREADS / TC;          // >= 1 (the host checks)
    const int grp = threadIdx.x / TC, e = threadIdx.x - grp * TC;
    const bool active = grp < groups;
    const int t = e / g.Ci, ci = e - t * g.Ci;
    const int dy = t / g.kw, dx = t - dy * g.kw;
    const int co0 = blockIdx.y * 8;
    const int Hc = g.Hp * g.pool, Wc = g.Wp * g.pool;
    const float* xbase = xs + (ci * S + dy) * SX + dx;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int cy0 = ty * T, cx0 = tx * T;
        __syncthreads();
        constexpr int SU = 4;                   // SU elements' loads in flight at once, no branch per element (see the forward kernel)
        for (int s0 = threadIdx.x; s0 < S * SX * g.Ci; s0 += DPP_THREADS * SU) {   // global order (pixel-major), planar in LDS
            float raw[SU];
            bool ok[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int s = s0 + u * DPP_THREADS;
                const int c = s % g.Ci, r = s / g.Ci;
                const int hx = r % SX, hy = r / SX;
                const int y = cy0 + hy - g.pad, x = cx0 + hx - g.pad;
                ok[u] = (s < S * SX * g.Ci) & (y >= 0) & (y < g.H) & (x >= 0) & (x < g.W);
                raw[u] = X[ok[u] ? (((size_t)n * g.H + y) * g.W + x) * g.Ci + c : 0];
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int s = s0 + u * DPP_THREADS;
                const int c = s % g.Ci, r = s / g.Ci;
                if (s < S * SX * g.Ci) xs[c * S * SX + r] = ok[u] ? dpp_act1(raw[u], act, c) : 0.0f;
            }
        }
        for (int s0 = threadIdx.x; s0 < T * T * 8; s0 += DPP_THREADS * SU) {
            float d[SU];
            int m[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int s = s0 + u * DPP_THREADS;
                const int c = s & 7, p = s >> 3;
                const int cy = cy0 + p / T, cx = cx0 + p % T;
                const bool ok = (s < T * T * 8) & (cy < Hc) & (cx < Wc) & (co0 + c < g.Co);
                const int py = ok ? cy / g.pool : 0, px = ok ? cx / g.pool : 0;
                const int j = ok ? (cy - py * g.pool) * g.pool + (cx - px * g.pool) : 0;
                const size_t o = (((size_t)n * g.Hp + py) * g.Wp + px) * g.Co + (ok ? co0 + c : 0);
                d[u] = dY[o];
                m[u] = ok ? ((ties ? (int)ties[o] : 1) >> j) & 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int s = s0 + u * DPP_THREADS;
                if (s < T * T * 8) gcs[s] = m[u] ? d[u] : 0.0f;
            }
        }
        __syncthreads();
        if (active) {
            const int lt = (T == 32) ? 5 : 4;                             // T is 16 or 32
            for (int p = grp; p < T * T; p += groups) {
                const int r = p >> lt, c = p & (T - 1);
                const float xv = xbase[r * SX + c];
                const float4* gp = reinterpret_cast<const float4*>(gcs + p * 8);
                const float4 g0 = gp[0], g1 = gp[1];
                acc[0] = fmaf(xv, g0.x, acc[0]); acc[1] = fmaf(xv, g0.y, acc[1]); acc[2] = fmaf(xv, g0.z, acc[2]); acc[3] = fmaf(xv, g0.w, acc[3]);
                acc[4] = fmaf(xv, g1.x, acc[4]); acc[5] = fmaf(xv, g1.y, acc[5]); acc[6] = fmaf(xv, g1.z, acc[6]); acc[7] = fmaf(xv, g1.w, acc[7]);
            }
        }
    }
    // combine the groups in order, then one partial per workgroup
    __syncthreads();
    float* red = gcs;                             // [groups][TC][8]  (<= 256 * 8 floats <= the gradient tile)
    if (active) {
#pragma unroll
        for (int c = 0; c < 8; ++c) red[(grp * TC + e) * 8 + c] = acc[c];
    }
    __syncthreads();
    if (threadIdx.x < TC) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = red[e * 8 + c];
            for (int k = 1; k < groups; ++k) v += red[(k * TC + e) * 8 + c];
            if (co0 + c < g.Co) partial[(size_t)blockIdx.x * nW + (size_t)(co0 + c) * TC + e] = v;
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

// The data gradient for the shapes the nets build (square K x K filter, pool P), compile-time K, P and PX x PX input pixels per thread.
// The generic kernel above walks (tap, output channel) per input pixel with a tie-mask load, a gradient load and a divergent branch
// in the innermost loop (86 us for the 0.28 GFLOP second PoseRegNet layer at batch 128).  Here the pooled gradient is first UNPOOLED
// into LDS -- gc[co][cy][cx] = dY[co][cy / P][cx / P] where the tie mask routes it to that conv pixel, else 0 -- and the data gradient
// is then a plain correlation over that map: a thread keeps the (PX + K - 1)^2 patch of gc around its pixels in registers per output
// channel, weights [co][tap][ci] come from LDS as two 16-byte broadcasts per (co, tap): (PX + K - 1)^2 + 2 K^2 LDS reads for
// 8 PX^2 K^2 multiply-adds.  Sums run over (co, dy, dx), co outermost.
template <int K, int P, int PX>
__global__ __launch_bounds__(DPP_THREADS) void convpool_dgrad_fast_kernel(const float* __restrict__ dY, const uint16_t* __restrict__ ties,
                                                                          const float* __restrict__ Wk, float* __restrict__ dX, cp_geom g,
                                                                          int tiles_x, int tiles_y) {
    HIP_DYNAMIC_SHARED(float, sm)
    constexpr int T = 16 * PX, S = T + K - 1, R = PX + K - 1, taps = K * K;
    float* gcs = sm;                          // [Co][S][S]
    float* ws = sm + g.Co * S * S;            // [Co][taps][8] : the 8 input channels of this blockIdx.y
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int c0 = blockIdx.y * 8;
    const int iy0 = ty * T, ix0 = tx * T;
    const int cy0 = iy0 + g.pad - (K - 1), cx0 = ix0 + g.pad - (K - 1);
    const int Hc = g.Hp * P, Wc = g.Wp * P;   // conv pixels that reach a pooled output
    constexpr int SU = 4;                       // see the forward kernel: SU elements' loads in flight at once, no branch per element
    for (int s0 = threadIdx.x; s0 < S * S * g.Co; s0 += DPP_THREADS * SU) {
        float d[SU];
        int m[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int s = s0 + u * DPP_THREADS;
            const int co = s % g.Co, rc = s / g.Co;
            const int c = rc % S, r = rc / S;
            const int cy = cy0 + r, cx = cx0 + c;
            const bool ok = (s < S * S * g.Co) & (cy >= 0) & (cx >= 0) & (cy < Hc) & (cx < Wc);
            const int py = ok ? cy / P : 0, px = ok ? cx / P : 0;
            const int j = ok ? (cy - py * P) * P + (cx - px * P) : 0;
            const size_t o = (((size_t)n * g.Hp + py) * g.Wp + px) * g.Co + (ok ? co : 0);
            d[u] = dY[o];
            m[u] = ok ? ((ties ? (int)ties[o] : 1) >> j) & 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int s = s0 + u * DPP_THREADS;
            const int co = s % g.Co, rc = s / g.Co;
            if (s < S * S * g.Co) gcs[(co * S + rc / S) * S + rc % S] = m[u] ? d[u] : 0.0f;
        }
    }
    for (int s = threadIdx.x; s < g.Co * taps * 8; s += DPP_THREADS) {
        const int ci = s & 7, ct = s >> 3;
        ws[s] = (c0 + ci < g.Ci) ? Wk[(size_t)ct * g.Ci + c0 + ci] : 0.0f;
    }
    __syncthreads();
    const int ly0 = (threadIdx.x >> 4) * PX, lx0 = (threadIdx.x & 15) * PX;
    float acc[PX * PX][8];
#pragma unroll
    for (int q = 0; q < PX * PX; ++q)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[q][c] = 0.0f;
    for (int co = 0; co < g.Co; ++co) {
        float patch[R][R];
        const float* gp = gcs + (co * S + ly0) * S + lx0;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < R; ++c) patch[r][c] = gp[r * S + c];
#pragma unroll
        for (int dy = 0; dy < K; ++dy)
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const float4* wp = reinterpret_cast<const float4*>(ws + (co * taps + dy * K + dx) * 8);
                const float4 w0 = wp[0], w1 = wp[1];
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int qy = 0; qy < PX; ++qy)
#pragma unroll
                    for (int qx = 0; qx < PX; ++qx) {
                        const float gv = patch[qy + K - 1 - dy][qx + K - 1 - dx];
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[qy * PX + qx][c] = fmaf(gv, w[c], acc[qy * PX + qx][c]);
                    }
            }
    }
#pragma unroll
    for (int qy = 0; qy < PX; ++qy)
#pragma unroll
        for (int qx = 0; qx < PX; ++qx) {
            const int iy = iy0 + ly0 + qy, ix = ix0 + lx0 + qx;
            if (iy < g.H && ix < g.W) {
                float* o = dX + (((size_t)n * g.H + iy) * g.W + ix) * g.Ci + c0;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c0 + c < g.Ci) o[c] = acc[qy * PX + qx][c];
            }
        }
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
    if (kh * kw * Ci <= DPP_THREADS && pool <= 2) {              // (pool >= 3: the unpooled map is mostly zeros, the kernel above skips them)
        const int T = (Ci <= 2) ? 32 : 16;                      // conv pixels per tile side
        const size_t need = ((size_t)Ci * (T + kh - 1) * (T + kw - 1) + (size_t)T * T * 8) * sizeof(float);
        if (need <= 64 * 1024 && (size_t)DPP_THREADS * 8 <= (size_t)T * T * 8) {
            const int tiles_x = dpp_cdiv(g.Wp * pool, T), tiles_y = dpp_cdiv(g.Hp * pool, T);
            DPP_LAUNCH(convpool_wgrad_dense_kernel, dim3(blocks, dpp_cdiv(Co, 8)), dim3(DPP_THREADS), need, static_cast<hipStream_t>(stream), X,
                       cp_act(act), dY, ties, partial, g, T, tiles_x, tiles_y, N * tiles_x * tiles_y);
            return dpp_launch_status();
        }
    }
    DPP_LAUNCH(convpool_wgrad_kernel, dim3(blocks), dim3(DPP_THREADS), lds, static_cast<hipStream_t>(stream), X, cp_act(act), dY,
                       ties, partial, g, total);
    return dpp_launch_status();
}

extern "C" int dpp_convpool_dgrad(const float* dY, const uint16_t* ties, int N, int H, int W, int Ci, const float* Wk, int kh, int kw,
                                  int pad, int Co, int pool, float* dX, dpp_stream_t stream) {
    cp_geom g;
    if (!dY || !Wk || !dX || !cp_make_geom(g, N, H, W, Ci, kh, kw, pad, Co, pool) || (pool > 1 && !ties)) return DPP_E_BADARG;
    if (kh == kw) {
        // 2 x 2 pixels per thread only when that still leaves several workgroups per CU (a 31 x 31 map at batch 128: 128 tiles of
        // 32 x 32 leave half the chip idle, 512 tiles of 16 x 16 do not)
        const int px = ((long)N * dpp_cdiv(H, 32) * dpp_cdiv(W, 32) >= 1024 && (H > 16 || W > 16)) ? 2 : 1, T = 16 * px, S = T + kh - 1;
        const size_t lds = ((size_t)Co * S * S + (size_t)Co * kh * kw * 8) * sizeof(float);
        const int tiles_x = dpp_cdiv(W, T), tiles_y = dpp_cdiv(H, T);
        const dim3 fgrid(N * tiles_x * tiles_y, dpp_cdiv(Ci, 8));
        if (lds <= 64 * 1024) {
#define DPP_CPD(K_, P_, PX_) if (kh == K_ && pool == P_ && px == PX_) { \
                DPP_LAUNCH((convpool_dgrad_fast_kernel<K_, P_, PX_>), fgrid, dim3(DPP_THREADS), lds, static_cast<hipStream_t>(stream), dY, ties, \
                           Wk, dX, g, tiles_x, tiles_y); \
                return dpp_launch_status(); }
            DPP_CPD(5, 2, 2) DPP_CPD(5, 2, 1) DPP_CPD(5, 1, 2) DPP_CPD(5, 1, 1) DPP_CPD(3, 1, 2) DPP_CPD(3, 1, 1) DPP_CPD(3, 2, 2) DPP_CPD(3, 2, 1)
            DPP_CPD(5, 4, 2) DPP_CPD(5, 4, 1)
#undef DPP_CPD
        }
    }
    size_t npix = (size_t)N * H * W;
    dim3 grid((unsigned)((npix + DPP_THREADS - 1) / DPP_THREADS), dpp_cdiv(Ci, 8));
    DPP_LAUNCH(convpool_dgrad_kernel, grid, dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), dY, ties, Wk, dX, g);
    return dpp_launch_status();
}
