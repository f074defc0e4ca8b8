// dpp_common.h -- shared device helpers for the gfx950 kernels of the DeepPrior++ hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dpp_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DPP_WAVE 64
#define DPP_THREADS 256

static inline int dpp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DPP_OK : (int)e;
}

static inline int dpp_cdiv(int a, int b) { return (a + b - 1) / b; }

// Row map of a compact (N,Ho,Wo) pixel index onto a (N,Hi,Wi) map sampled with stride s.
__device__ __forceinline__ int dpp_map_row(const dpp_rowmap& m, int r) {
    if (m.s == 1) return r;
    int n = r / m.HoWo;
    int q = r - n * m.HoWo;
    int y = q / m.Wo;
    int x = q - y * m.Wo;
    return n * m.HiWi + (y * m.s) * m.Wi + x * m.s;
}

// The fused BatchNorm(+ReLU) operand prologue: v = (x - mean) * scale + beta ; relu.
// Written as (x - mean) * scale + beta (not x*s + t) so that no bits are lost when |mean| >> std.
__device__ __forceinline__ float dpp_act1(float x, const dpp_act& a, int c) {
    float v = x;
    if (a.mode & 2) v = (x - a.mean[c]) * a.scale[c] + a.beta[c];
    if (a.mode & 1) v = fmaxf(v, 0.0f);
    return v;
}

__device__ __forceinline__ float4 dpp_act4(float4 x, const dpp_act& a, int c) {
    float4 v = x;
    if (a.mode & 2) {
        const float4 mu = *reinterpret_cast<const float4*>(a.mean + c);
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
        const float4 be = *reinterpret_cast<const float4*>(a.beta + c);
        v.x = (x.x - mu.x) * sc.x + be.x;
        v.y = (x.y - mu.y) * sc.y + be.y;
        v.z = (x.z - mu.z) * sc.z + be.z;
        v.w = (x.w - mu.w) * sc.w + be.w;
    }
    if (a.mode & 1) {
        v.x = fmaxf(v.x, 0.0f);
        v.y = fmaxf(v.y, 0.0f);
        v.z = fmaxf(v.z, 0.0f);
        v.w = fmaxf(v.w, 0.0f);
    }
    return v;
}

__device__ __forceinline__ float dpp_f4_get(const float4& v, int i) {
    return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// ---- column reductions of an MFMA output tile in the D layout ----------------------------------------------------------
// A wave holds RM x CN tiles of 16x16: lane (l15 = column, kq = row quad), register r -> row kq*4 + r.  `s[ct]` enters as
// this lane's partial sum for column tile ct and leaves as the sum over ALL rows of the workgroup tile (every lane of the
// column gets it).  WM waves stack along rows (they share columns), WN waves sit side by side.  `red` is LDS scratch of at
// least WM*BN floats that nobody else uses during the call.
template <int CN, int WM, int WN, int BN>
__device__ __forceinline__ void dpp_tile_colsum(float (&s)[CN], float* red, int wm, int wn, int l15, int kq) {
#pragma unroll
    for (int ct = 0; ct < CN; ++ct) {
        s[ct] += __shfl_xor(s[ct], 16);
        s[ct] += __shfl_xor(s[ct], 32);
    }
    if (WM > 1) {
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int ct = 0; ct < CN; ++ct) red[wm * BN + wn * (BN / WN) + ct * 16 + l15] = s[ct];
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < CN; ++ct) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[w * BN + wn * (BN / WN) + ct * 16 + l15];
            s[ct] = t;
        }
    }
}
