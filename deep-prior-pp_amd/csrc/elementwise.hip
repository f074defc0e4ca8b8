// elementwise.hip -- loss, ADAM, column sums and small helpers for gfx950 (all HBM-bound, float4 lanes).
#include "dpp_common.h"

namespace {

// ---- column sums: bias gradients db[c] = sum_m dY[m][c] (T.grad of `+ b.dimshuffle`, convlayer.py:238) ----
__global__ __launch_bounds__(DPP_THREADS) void colsum_partial_kernel(const float* __restrict__ X, int M, int C, int rpb,
                                                                     float* __restrict__ partial) {
    __shared__ float s[DPP_THREADS];
    const int tid = threadIdx.x;
    const int RP = DPP_THREADS / C > 0 ? DPP_THREADS / C : 1;   // rows per pass when C <= 256
    const int r_begin = blockIdx.x * rpb;
    const int r_end = (r_begin + rpb < M) ? r_begin + rpb : M;
    for (int c0 = 0; c0 < C; c0 += DPP_THREADS) {               // C > 256: several column passes
        int cw = (C - c0 < DPP_THREADS) ? C - c0 : DPP_THREADS;
        int rp = DPP_THREADS / cw;
        int c = tid % cw, rr = tid / cw;
        float acc = 0.0f;
        if (rr < rp)
            for (int r = r_begin + rr; r < r_end; r += rp) acc += X[(size_t)r * C + c0 + c];
        s[tid] = acc;
        __syncthreads();
        if (tid < cw) {
            double a = 0.0;
            for (int j = 0; j < rp; ++j) a += (double)s[j * cw + tid];
            partial[(size_t)blockIdx.x * C + c0 + tid] = (float)a;
        }
        __syncthreads();
    }
    (void)RP;
}

// ---- loss: cost = (1/denom) * sum_n sum_d (out - y)^2 ; dout = (2/denom) * (out - y) -----------------------
// embedding case denom = B, joints case denom = B*J (poseregnettrainer.py:92-99).  One block.
__global__ __launch_bounds__(DPP_THREADS) void loss_sse_kernel(const float* __restrict__ out, const float* __restrict__ y, int n,
                                                               float inv_denom, float* __restrict__ cost,
                                                               float* __restrict__ dout) {
    __shared__ double s[DPP_THREADS];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += DPP_THREADS) {
        float d = out[i] - y[i];
        acc += (double)d * (double)d;
        if (dout) dout[i] = 2.0f * inv_denom * d;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = DPP_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[0] = (float)(s[0] * (double)inv_denom);
}

// The last HiddenLayer's split-K reduction and the cost in ONE launch (round 6: the loss was a 5 us launch of its own between the forward
// and the backward chain): out[i] = sum_z partial[z][i] + bias[i % nbias] in the slice order of reduce_partials_kernel's one-lane form,
// then loss_sse_kernel's arithmetic on the values just written.  One workgroup; n = batch x output dimension (3 840 for the 30-D
// embedding at batch 128).
constexpr int RPL_THREADS = 1024;
__global__ __launch_bounds__(RPL_THREADS) void reduce_partials_loss_kernel(const float* __restrict__ partial, int nz, int n,
                                                                           const float* __restrict__ bias, int nbias, float* __restrict__ out,
                                                                           const float* __restrict__ y, float inv_denom,
                                                                           float* __restrict__ cost, float* __restrict__ dout) {
    // 1024 threads, one element quad each (3 840 outputs = 960 quads): every slice's load of a thread is issued before the first is
    // used -- ONE memory round trip for the whole reduction (the first version walked 15 elements x 8 slices per thread of a 256-thread
    // block, a dependent round trip per element: 0.02 ms slower per step than the two launches it replaced, profiles/r06_ab.txt)
    __shared__ double s[RPL_THREADS / DPP_WAVE];
    double acc = 0.0;
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(y) |
                                       reinterpret_cast<uintptr_t>(dout)) & 15) == 0;
    if (vec) {
        const int n4 = n >> 2;
        for (int q = threadIdx.x; q < n4; q += RPL_THREADS) {
            float4 v[16];
            const int nzc = nz < 16 ? nz : 16;
#pragma unroll
            for (int z = 0; z < 16; ++z) v[z] = reinterpret_cast<const float4*>(partial + (size_t)(z < nzc ? z : 0) * n)[q];
            const float4 yv = reinterpret_cast<const float4*>(y)[q];
            // (the output dimension -- 30 for the embedding -- need not be a multiple of four: the bias element by element)
            const int i0 = q << 2;
            const float4 bv = bias ? make_float4(bias[i0 % nbias], bias[(i0 + 1) % nbias], bias[(i0 + 2) % nbias], bias[(i0 + 3) % nbias])
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 t = v[0];
#pragma unroll
            for (int z = 1; z < 16; ++z)
                if (z < nzc) { t.x += v[z].x; t.y += v[z].y; t.z += v[z].z; t.w += v[z].w; }
            for (int z = 16; z < nz; ++z) {
                const float4 u = reinterpret_cast<const float4*>(partial + (size_t)z * n)[q];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            t.x += bv.x; t.y += bv.y; t.z += bv.z; t.w += bv.w;
            reinterpret_cast<float4*>(out)[q] = t;
            const float d0 = t.x - yv.x, d1 = t.y - yv.y, d2 = t.z - yv.z, d3 = t.w - yv.w;
            acc += (double)d0 * (double)d0 + (double)d1 * (double)d1 + (double)d2 * (double)d2 + (double)d3 * (double)d3;
            if (dout) reinterpret_cast<float4*>(dout)[q] = make_float4(2.0f * inv_denom * d0, 2.0f * inv_denom * d1, 2.0f * inv_denom * d2, 2.0f * inv_denom * d3);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += RPL_THREADS) {
            float v = 0.0f;
            for (int z = 0; z < nz; ++z) v += partial[(size_t)z * n + i];
            if (bias) v += bias[i % nbias];
            out[i] = v;
            const float d = v - y[i];
            acc += (double)d * (double)d;
            if (dout) dout[i] = 2.0f * inv_denom * d;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
        for (int w = 0; w < RPL_THREADS / DPP_WAVE; ++w) tsum += s[w];
        cost[0] = (float)(tsum * (double)inv_denom);
    }
}

// The scalar-target cost of poseregnettrainer.py:84-85, 92-93 (numJoints == nDims == 1): the net output reshaped to (B, 1) -- a
// broadcastable column in Theano -- minus the VECTOR y (B,) is the (B, B) matrix o_i - y_j, so
//   cost = mean_i mean_j (o_i - y_j)^2 ,   d cost / d o_i = (2 / B) (o_i - mean(y)).
// (Whatever the authors meant, this is what the graph computes.)  One block; sums in f64.
__global__ __launch_bounds__(DPP_THREADS) void loss_sse_bcast_kernel(const float* __restrict__ out, const float* __restrict__ y, int n,
                                                                     float* __restrict__ cost, float* __restrict__ dout,
                                                                     float* __restrict__ err) {
    __shared__ double s[4][DPP_THREADS];
    double so = 0.0, soo = 0.0, sy = 0.0, syy = 0.0;
    for (int i = threadIdx.x; i < n; i += DPP_THREADS) {
        const double o = out[i], t = y[i];
        so += o; soo += o * o; sy += t; syy += t * t;
    }
    s[0][threadIdx.x] = so; s[1][threadIdx.x] = soo; s[2][threadIdx.x] = sy; s[3][threadIdx.x] = syy;
    __syncthreads();
    for (int w = DPP_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 4; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + w];
        __syncthreads();
    }
    const double B = (double)n, ybar = s[2][0] / B;
    if (threadIdx.x == 0) cost[0] = (float)((B * s[1][0] - 2.0 * s[0][0] * s[2][0] + B * s[3][0]) / (B * B));
    if (dout)
        for (int i = threadIdx.x; i < n; i += DPP_THREADS) dout[i] = (float)(2.0 / B * ((double)out[i] - ybar));
    if (err == nullptr) return;
    // the monitor of poseregnettrainer.py:115 under the same broadcast: mean (and max) over all pairs of |o_i - y_j|
    __syncthreads();
    double acc = 0.0, mx = 0.0;
    for (int p = threadIdx.x; p < n * n; p += DPP_THREADS) {
        const double e = fabs((double)out[p / n] - (double)y[p % n]);
        acc += e;
        mx = e > mx ? e : mx;
    }
    s[0][threadIdx.x] = acc; s[1][threadIdx.x] = mx;
    __syncthreads();
    for (int w = DPP_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s[0][threadIdx.x] += s[0][threadIdx.x + w];
            s[1][threadIdx.x] = s[1][threadIdx.x + w] > s[1][threadIdx.x] ? s[1][threadIdx.x + w] : s[1][threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { err[0] = (float)(s[0][0] / (B * B)); err[1] = (float)s[1][0]; }
}

// err[0] = mean_rows sqrt(sum_d (out-y)^2), err[1] = max_rows of the same: the monitors of poseregnettrainer.py:114-129
// (errors / errors_avg / errors_max).
__global__ __launch_bounds__(DPP_THREADS) void error_l2_kernel(const float* __restrict__ out, const float* __restrict__ y, int rows,
                                                               int d, float* __restrict__ err) {
    __shared__ double s[DPP_THREADS];
    __shared__ double smax[DPP_THREADS];
    double acc = 0.0, mx = 0.0;
    for (int r = threadIdx.x; r < rows; r += DPP_THREADS) {
        double q = 0.0;
        for (int j = 0; j < d; ++j) {
            double e = (double)out[(size_t)r * d + j] - (double)y[(size_t)r * d + j];
            q += e * e;
        }
        q = sqrt(q);
        acc += q;
        mx = q > mx ? q : mx;
    }
    s[threadIdx.x] = acc;
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int w = DPP_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s[threadIdx.x] += s[threadIdx.x + w];
            smax[threadIdx.x] = smax[threadIdx.x + w] > smax[threadIdx.x] ? smax[threadIdx.x + w] : smax[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { err[0] = (float)(s[0] / (double)rows); err[1] = (float)smax[0]; }
}

// ---- ADAM (optimizer.py:58-90), one launch over the flat parameter buffer -------------------------------
// state (device, 8 floats): lr, t, beta1, beta2, epsilon, gamma, -, -.  The scalar terms of optimizer.py:69-84 are evaluated
// here in float32 (beta1_t = beta1*gamma^(t-1), 1-beta1^t, 1-beta2^t) so that a captured step can be replayed without
// any host upload; adam_tick_kernel advances t after the update (the reference's `t <- t + 1`).
// tick (round 6, dpp_adam_ticked): the launch advances t itself -- every workgroup takes a ticket (state[7], an unsigned counter) when it
// is done, the LAST one resets the counter and bumps t: all the others have read t by then (a workgroup reads the state before its
// first element), so the 5 us adam_tick launch at the end of every step is gone.
__device__ __forceinline__ void adam_take_ticket(float* state) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* ctr = reinterpret_cast<unsigned*>(state + 7);
        if (atomicAdd(ctr, 1u) == gridDim.x - 1) { *ctr = 0u; state[1] += 1.0f; }
    }
}

__global__ __launch_bounds__(DPP_THREADS) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, size_t n, float* state, int tick) {
    const float lr = state[0], t = state[1], beta1 = state[2], b2 = state[3], eps = state[4], gamma = state[5];
    if (state[6] != 0.0f) {
        // RMSProp (optimizer.py:92-116): msg = decay * msg + (1 - decay) * g^2;  w += -lr * g / max(sqrt(msg), epsilon).
        // state[3] = decay, state[4] = epsilon (the clip); msg lives in v, m is not used.
        const float od = 1.0f - b2;
        for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) {
            const float gg = g[i];
            const float ms = b2 * v[i] + od * (gg * gg);
            w[i] = w[i] + (-lr * gg) / fmaxf(sqrtf(ms), eps);
            v[i] = ms;
        }
        if (tick) adam_take_ticket(state);
        return;
    }
    const float b1 = beta1 * powf(gamma, t - 1.0f);
    const float ob1 = 1.0f - b1, ob2 = 1.0f - b2;
    const float c1 = 1.0f - powf(beta1, t), c2 = 1.0f - powf(b2, t);
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * DPP_THREADS) {
        float4 W = reinterpret_cast<float4*>(w)[i], G = reinterpret_cast<const float4*>(g)[i];
        float4 Mv = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
#define DPP_ADAM1(f)                                            \
        Mv.f = b1 * Mv.f + ob1 * G.f;                           \
        V.f = b2 * V.f + ob2 * (G.f * G.f);                     \
        W.f = W.f - (lr * (Mv.f / c1)) / (sqrtf(V.f / c2) + eps);
        DPP_ADAM1(x) DPP_ADAM1(y) DPP_ADAM1(z) DPP_ADAM1(w)
        reinterpret_cast<float4*>(w)[i] = W;
        reinterpret_cast<float4*>(m)[i] = Mv;
        reinterpret_cast<float4*>(v)[i] = V;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t i = (n4 << 2) + threadIdx.x;
        float mm = b1 * m[i] + ob1 * g[i];
        float vv = b2 * v[i] + ob2 * (g[i] * g[i]);
        w[i] = w[i] - (lr * (mm / c1)) / (sqrtf(vv / c2) + eps);
        m[i] = mm;
        v[i] = vv;
    }
#undef DPP_ADAM1
    if (tick) adam_take_ticket(state);
}

__global__ void adam_tick_kernel(float* state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += 1.0f;
}

// y[i] += alpha * x[i]   (L2 weight-decay gradient 2*wd*W, poseregnettrainer.py:101-107)
__global__ __launch_bounds__(DPP_THREADS) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, size_t n) {
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) y[i] += alpha * x[i];
}

// out[0] (+)= alpha * sum x^2 -- single block, deterministic (the wd * sum(W^2) term of the cost)
__global__ __launch_bounds__(DPP_THREADS) void sumsq_kernel(const float* __restrict__ x, size_t n, float alpha, float* out, int accumulate) {
    __shared__ double s[DPP_THREADS];
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += DPP_THREADS) acc += (double)x[i] * (double)x[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = DPP_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + (float)(s[0] * (double)alpha);
}

// The same two over SEGMENTS of one flat buffer in a single launch (the weights of all layers inside the flat parameter buffer,
// seg[2 s] = offset, seg[2 s + 1] = length in elements): the trainers' default weightreg_factor is 0.001 (nettrainer.py:52) and the
// 67 per-layer launches of the ResNet -- the single-block sum over FC1's 16.8 M weights alone took 17 ms -- made a step with the
// regulariser six times the step without.  Every block takes the same 1 / gridDim share of every segment, so the block partials
// (f64) and their fixed-order sum do not depend on timing.
constexpr int SEG_BLOCKS = 1024;

__global__ __launch_bounds__(DPP_THREADS) void sumsq_multi_kernel(const float* __restrict__ base, const long long* __restrict__ seg, int nseg,
                                                                  double* __restrict__ partial) {
    __shared__ double s[DPP_THREADS];
    double acc = 0.0;
    for (int k = 0; k < nseg; ++k) {
        const long long off = seg[2 * k], len = seg[2 * k + 1];
        const long long per = (len + gridDim.x - 1) / gridDim.x;
        const long long i0 = (long long)blockIdx.x * per, i1 = (i0 + per < len) ? i0 + per : len;
        const float* x = base + off;
        for (long long i = i0 + threadIdx.x; i < i1; i += DPP_THREADS) acc += (double)x[i] * (double)x[i];
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = DPP_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(DPP_THREADS) void sumsq_finish_kernel(const double* __restrict__ partial, int n, float alpha, float* out, int accumulate) {
    __shared__ double s[DPP_THREADS];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += DPP_THREADS) acc += partial[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int w = DPP_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + (float)(s[0] * (double)alpha);
}

__global__ __launch_bounds__(DPP_THREADS) void axpy_multi_kernel(float* __restrict__ ybase, const float* __restrict__ xbase,
                                                                 const long long* __restrict__ seg, int nseg, float alpha) {
    for (int k = 0; k < nseg; ++k) {
        const long long off = seg[2 * k], len = seg[2 * k + 1];
        const long long per = (len + gridDim.x - 1) / gridDim.x;
        const long long i0 = (long long)blockIdx.x * per, i1 = (i0 + per < len) ? i0 + per : len;
        float* y = ybase + off;
        const float* x = xbase + off;
        for (long long i = i0 + threadIdx.x; i < i1; i += DPP_THREADS) y[i] += alpha * x[i];
    }
}

// y = a * x  elementwise with optional relu first: the deterministic DropoutLayer (prob_keep * x,
// dropoutlayer.py:104) applied to relu(pre) ; mask variant: y = mask * relu?(x)
__global__ __launch_bounds__(DPP_THREADS) void scale_kernel(const float* __restrict__ x, const float* __restrict__ mask, float a, int relu,
                                                            float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) {
        float v = x[i];
        if (relu) v = fmaxf(v, 0.0f);
        y[i] = mask ? mask[i] * v : a * v;
    }
}

// g = dy * [pre >= 0]  (ReLU backward on a stored pre-activation), optional extra factor (dropout mask or keep prob)
__global__ __launch_bounds__(DPP_THREADS) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                                               const float* __restrict__ mask, float a, float* __restrict__ g, size_t n) {
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) {
        float v = dy[i] * (mask ? mask[i] : a);
        g[i] = (pre[i] >= 0.0f) ? v : 0.0f;
    }
}

// Bernoulli(keep) mask from a counter-based generator (splitmix64 of (seed, counter, index)): the dropout mask of
// dropoutlayer.py:98-103 (the reference's MRG31k3p stream is not reproduced bit for bit)
__global__ __launch_bounds__(DPP_THREADS) void bernoulli_mask_kernel(float* __restrict__ mask, size_t n, float keep, unsigned long long seed,
                                                                     unsigned long long counter,
                                                                     const unsigned long long* __restrict__ counter_dev) {
    if (counter_dev) counter += *counter_dev;
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) {
        unsigned long long z = seed * 0x9E3779B97F4A7C15ull + counter * 0xD1B54A32D192ED03ull + i + 0x632BE59BD9B4E019ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        float u = (float)(z >> 40) * (1.0f / 16777216.0f);
        mask[i] = u < keep ? 1.0f : 0.0f;
    }
}

// dst[r][c] = relu?(src[r][c]) for a rows x cols block with independent row strides: packs the flattened tower outputs of
// ScaleNet side by side (T.concatenate(..., axis=1), scalenet.py:167-171) and splits the gradient again.
__global__ __launch_bounds__(DPP_THREADS) void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                                             int rows, int cols, int relu) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) {
        const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
        float v = src[(size_t)r * lds + c];
        dst[(size_t)r * ldd + c] = relu ? fmaxf(v, 0.0f) : v;
    }
}

// out[r][c] = x[r][c] * (s[r * sld + scol] * factor): normalised joint labels back to millimetres, label * cube_z / 2, as the
// augmentation's input when no separate mm-space copy is kept (poseregnettrainer.py:228-240, scalenettrainer.py:226-229).
// The product s * factor is formed first, like the reference's `(cube[2] / 2.)`.
__global__ __launch_bounds__(DPP_THREADS) void rowscale_kernel(const float* __restrict__ x, const float* __restrict__ s, int sld, int scol,
                                                               float factor, float* __restrict__ out, int rows, int cols) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) {
        const int r = (int)(i / cols);
        out[i] = x[i] * (s[(size_t)r * sld + scol] * factor);
    }
}

// centre h x w window of every [H][W] image of a batch (scalenettrainer.py:239-251, handdetector.py:654-666)
__global__ __launch_bounds__(DPP_THREADS) void crop_center_kernel(const float* __restrict__ src, int B, int H, int W, float* __restrict__ dst,
                                                                  int h, int w, int y0, int x0) {
    const size_t n = (size_t)B * h * w;
    for (size_t i = (size_t)blockIdx.x * DPP_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * DPP_THREADS) {
        const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((size_t)w * h));
        dst[i] = src[((size_t)b * H + y0 + y) * W + x0 + x];
    }
}

int grid_for(size_t n) {
    size_t b = (n + DPP_THREADS - 1) / DPP_THREADS;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int dpp_colsum_partial(const float* X, int M, int C, int rows_per_block, float* partial, dpp_stream_t stream) {
    if (!X || !partial || M < 1 || C < 1 || rows_per_block < 1) return DPP_E_BADARG;
    DPP_LAUNCH(colsum_partial_kernel, dim3(dpp_cdiv(M, rows_per_block)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream),
                       X, M, C, rows_per_block, partial);
    return dpp_launch_status();
}

extern "C" int dpp_loss_sse(const float* out, const float* y, int rows, int d, int denom, float* cost, float* dout,
                            dpp_stream_t stream) {
    if (!out || !y || !cost || rows < 1 || d < 1 || denom < 1) return DPP_E_BADARG;
    DPP_LAUNCH(loss_sse_kernel, dim3(1), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), out, y, rows * d,
                       1.0f / (float)denom, cost, dout);
    return dpp_launch_status();
}

extern "C" int dpp_reduce_partials_loss(const float* partial, int nz, int rows, int d, const float* bias, float* out, const float* y, int denom,
                                        float* cost, float* dout, dpp_stream_t stream) {
    if (!partial || !out || !y || !cost || nz < 1 || rows < 1 || d < 1 || denom < 1 || (long)rows * d > 65536) return DPP_E_BADARG;
    DPP_LAUNCH(reduce_partials_loss_kernel, dim3(1), dim3(RPL_THREADS), 0, static_cast<hipStream_t>(stream), partial, nz, rows * d, bias, d, out, y,
               1.0f / (float)denom, cost, dout);
    return dpp_launch_status();
}

extern "C" int dpp_loss_sse_bcast(const float* out, const float* y, int n, float* cost, float* dout, float* err, dpp_stream_t stream) {
    if (!out || !y || !cost || n < 1 || n > 32768) return DPP_E_BADARG;
    DPP_LAUNCH(loss_sse_bcast_kernel, dim3(1), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), out, y, n, cost, dout, err);
    return dpp_launch_status();
}

extern "C" int dpp_error_l2(const float* out, const float* y, int rows, int d, float* err, dpp_stream_t stream) {
    if (!out || !y || !err || rows < 1 || d < 1) return DPP_E_BADARG;
    DPP_LAUNCH(error_l2_kernel, dim3(1), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), out, y, rows, d, err);
    return dpp_launch_status();
}

extern "C" int dpp_adam_tick(float* state, dpp_stream_t stream) {
    if (!state) return DPP_E_BADARG;
    DPP_LAUNCH(adam_tick_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state);
    return dpp_launch_status();
}

static int adam_launch(float* w, const float* g, float* m, float* v, size_t n, float* hyper, int tick, dpp_stream_t stream) {
    if (!w || !g || !m || !v || !hyper || n < 1) return DPP_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
        return DPP_E_BADARG;
    DPP_LAUNCH(adam_kernel, dim3(grid_for(n >> 2)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), w, g, m, v, n, hyper, tick);
    return dpp_launch_status();
}

extern "C" int dpp_adam(float* w, const float* g, float* m, float* v, size_t n, const float* hyper, dpp_stream_t stream) {
    return adam_launch(w, g, m, v, n, const_cast<float*>(hyper), 0, stream);
}

extern "C" int dpp_adam_ticked(float* w, const float* g, float* m, float* v, size_t n, float* hyper, dpp_stream_t stream) {
    return adam_launch(w, g, m, v, n, hyper, 1, stream);
}

extern "C" int dpp_axpy(float* y, const float* x, float alpha, size_t n, dpp_stream_t stream) {
    if (!y || !x || n < 1) return DPP_E_BADARG;
    DPP_LAUNCH(axpy_kernel, dim3(grid_for(n)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), y, x, alpha, n);
    return dpp_launch_status();
}

extern "C" int dpp_sumsq(const float* x, size_t n, float alpha, float* out, int accumulate, dpp_stream_t stream) {
    if (!x || !out || n < 1) return DPP_E_BADARG;
    DPP_LAUNCH(sumsq_kernel, dim3(1), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), x, n, alpha, out, accumulate);
    return dpp_launch_status();
}

extern "C" size_t dpp_sumsq_multi_workspace_bytes(void) { return SEG_BLOCKS * sizeof(double); }

extern "C" int dpp_sumsq_multi(const float* base, const long long* seg, int nseg, float alpha, void* workspace, float* out, int accumulate,
                               dpp_stream_t stream) {
    if (!base || !seg || !workspace || !out || nseg < 1) return DPP_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DPP_LAUNCH(sumsq_multi_kernel, dim3(SEG_BLOCKS), dim3(DPP_THREADS), 0, st, base, seg, nseg, static_cast<double*>(workspace));
    DPP_LAUNCH(sumsq_finish_kernel, dim3(1), dim3(DPP_THREADS), 0, st, static_cast<const double*>(workspace), SEG_BLOCKS, alpha, out, accumulate);
    return dpp_launch_status();
}

extern "C" int dpp_axpy_multi(float* ybase, const float* xbase, const long long* seg, int nseg, float alpha, dpp_stream_t stream) {
    if (!ybase || !xbase || !seg || nseg < 1) return DPP_E_BADARG;
    DPP_LAUNCH(axpy_multi_kernel, dim3(SEG_BLOCKS), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), ybase, xbase, seg, nseg, alpha);
    return dpp_launch_status();
}

extern "C" int dpp_scale(const float* x, const float* mask, float a, int relu, float* y, size_t n, dpp_stream_t stream) {
    if (!x || !y || n < 1) return DPP_E_BADARG;
    DPP_LAUNCH(scale_kernel, dim3(grid_for(n)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), x, mask, a, relu, y, n);
    return dpp_launch_status();
}

extern "C" int dpp_relu_bwd(const float* dy, const float* pre, const float* mask, float a, float* g, size_t n, dpp_stream_t stream) {
    if (!dy || !pre || !g || n < 1) return DPP_E_BADARG;
    DPP_LAUNCH(relu_bwd_kernel, dim3(grid_for(n)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), dy, pre, mask, a, g, n);
    return dpp_launch_status();
}

extern "C" int dpp_fill_zero(void* p, size_t nbytes, dpp_stream_t stream) {
    if (!p) return DPP_E_BADARG;
    if (dpp_tls_plan != nullptr) {
        dpp_plan_node n;
        n.kind = 1;
        n.ptr = p;
        n.nbytes = nbytes;
        dpp_plan_append(dpp_tls_plan, std::move(n));
        return DPP_OK;
    }
    hipError_t e = hipMemsetAsync(p, 0, nbytes, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? DPP_OK : (int)e;
}

extern "C" int dpp_bernoulli_mask(float* mask, size_t n, float keep, unsigned long long seed, unsigned long long counter,
                                  const unsigned long long* counter_dev, dpp_stream_t stream) {
    if (!mask || n < 1) return DPP_E_BADARG;
    DPP_LAUNCH(bernoulli_mask_kernel, dim3(grid_for(n)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), mask, n, keep, seed,
                       counter, counter_dev);
    return dpp_launch_status();
}

extern "C" int dpp_copy2d(const float* src, int lds, float* dst, int ldd, int rows, int cols, int relu, dpp_stream_t stream) {
    if (!src || !dst || rows < 1 || cols < 1 || lds < cols || ldd < cols) return DPP_E_BADARG;
    DPP_LAUNCH(copy2d_kernel, dim3(grid_for((size_t)rows * cols)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), src, lds,
                       dst, ldd, rows, cols, relu);
    return dpp_launch_status();
}

extern "C" int dpp_rowscale(const float* x, const float* s, int sld, int scol, float factor, float* out, int rows, int cols,
                            dpp_stream_t stream) {
    if (!x || !s || !out || rows < 1 || cols < 1 || sld < 1 || scol < 0 || scol >= sld) return DPP_E_BADARG;
    DPP_LAUNCH(rowscale_kernel, dim3(grid_for((size_t)rows * cols)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), x, s, sld, scol,
               factor, out, rows, cols);
    return dpp_launch_status();
}

extern "C" int dpp_crop_center(const float* src, int B, int H, int W, float* dst, int h, int w, dpp_stream_t stream) {
    if (!src || !dst || B < 1 || h < 1 || w < 1 || h > H || w > W) return DPP_E_BADARG;
    // the reference's index arithmetic: start = int(size / 2 - dsize / 2) along each axis
    int y0 = (int)(H / 2.0 - h / 2.0), x0 = (int)(W / 2.0 - w / 2.0);
    DPP_LAUNCH(crop_center_kernel, dim3(grid_for((size_t)B * h * w)), dim3(DPP_THREADS), 0, static_cast<hipStream_t>(stream), src, B,
                       H, W, dst, h, w, y0, x0);
    return dpp_launch_status();
}
