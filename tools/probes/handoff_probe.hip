// handoff_probe.hip -- what is a BatchNorm finalize worth as (a) its own launch between producer and consumer (what the step does
// today), (b) an in-kernel hand-off inside the CONSUMER launch: the first C/4 workgroups combine the producer's per-block partials
// (one wave per channel), publish (mean, scale) with write-through (sc1) stores and bump an arrival counter; every workgroup issues
// its operand loads first, one lane polls the counter (relaxed, agent scope), then the coefficients are read with sc1 loads (no
// fences on either side: MI355X_MICROARCH.md, "sc1 stores and loads both sides"), (c) not at all (lower bound, wrong numbers),
// (d) round 4, the two-level form of VERDICT r3 task 1(a), made placement-independent: the PRODUCER's workgroups are grouped by
// block id (b % 8 -- the XCD the dispatcher is observed to give block b, used for speed only), every workgroup releases its partials
// (agent scope) and takes a ticket on its group's counter, the group's last arriver acquires, folds the group's partials
// ([2][C][8][NB/8], contiguous per group) and publishes ONE (sum, sumsq) row per channel; the CONSUMER -- the next launch, so the
// kernel boundary orders it -- combines the 8 rows per channel in its prologue.  No finalize launch, no polling.
// A chain of L "layers": y = relu((x - mean) * scale) over [M][C] f32, each layer also emits the per-row-block (sum, sumsq) partials
// of what it writes -- an elementwise stand-in for a 1x1 convolution with the same memory behaviour and the same dependency chain.
//   hipcc --offload-arch=gfx950 -O3 [-DHANDOFF_NOFENCE] tools/probes/handoff_probe.hip -o /tmp/handoff_probe && /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define THREADS 256
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ double wave_allsum(double v) {
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// one wave per channel: (sum, sumsq) partials [2][C][NB] -> mean, scale = 1/sqrt(var + eps)
__device__ __forceinline__ void finalize_channel(const float* __restrict__ partial, int c, int C, int NB, int M, float* mean, float* scale,
                                                 bool sc1) {
    const int lane = threadIdx.x & 63;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < NB; b += 64) {
        s += (double)partial[(size_t)c * NB + b];
        q += (double)partial[((size_t)C + c) * NB + b];
    }
    s = wave_allsum(s);
    q = wave_allsum(q);
    if (lane == 0) {
        const double mu = s / M, var = q / M - mu * mu;
        const float m = (float)mu, sc = (float)(1.0 / sqrt(fabs(var) + 1e-4));
        if (sc1) { st_sc1(mean + c, m); st_sc1(scale + c, sc); } else { mean[c] = m; scale[c] = sc; }
    }
}

__global__ __launch_bounds__(THREADS) void finalize_kernel(const float* __restrict__ partial, int C, int NB, int M, float* mean, float* scale) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c < C) finalize_channel(partial, c, C, NB, M, mean, scale, false);
}

// MODE 0: coefficients are final (plain loads); MODE 1: in-kernel hand-off; ROWS rows per workgroup
template <int MODE, int C, int ROWS>
__global__ __launch_bounds__(THREADS) void layer_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ pin,
                                                        float* mean, float* scale, float* __restrict__ pout, int M, unsigned* counter,
                                                        const unsigned* __restrict__ step_word, const float* __restrict__ rows_in = nullptr,
                                                        float* rows_out = nullptr) {
    constexpr int Q = C / 4, RSTEP = THREADS / Q, IT = ROWS / RSTEP;
    __shared__ float red[2][4][C];
    const int tid = threadIdx.x, cq = tid % Q, rb = tid / Q, NB = gridDim.x;
    const size_t row0 = (size_t)blockIdx.x * ROWS;
    unsigned target = 0;
    if (MODE == 1) {
        const int npub = C / 4;
        target = (*step_word + 1u) * (unsigned)npub;
        if ((int)blockIdx.x < npub) {
            finalize_channel(pin, blockIdx.x * 4 + (tid >> 6), C, NB, M, mean, scale, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) v[i] = *reinterpret_cast<const float4*>(x + (row0 + rb + i * RSTEP) * C + cq * 4);
    float mu[4], sc[4];
    if (MODE == 1) {
        if (tid == 0) {
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = ld_sc1(mean + cq * 4 + j); sc[j] = ld_sc1(scale + cq * 4 + j); }
    } else if (MODE == 2) {
        // eight (sum, sumsq) rows per channel, combined in group order: 16 independent 16-byte loads per thread
        float4 rs[8], rq[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            rs[g] = *reinterpret_cast<const float4*>(rows_in + (size_t)g * C + cq * 4);
            rq[g] = *reinterpret_cast<const float4*>(rows_in + (size_t)(8 + g) * C + cq * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double ss = 0.0, qq = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                ss += (double)(j == 0 ? rs[g].x : j == 1 ? rs[g].y : j == 2 ? rs[g].z : rs[g].w);
                qq += (double)(j == 0 ? rq[g].x : j == 1 ? rq[g].y : j == 2 ? rq[g].z : rq[g].w);
            }
            const double m_ = ss / M, var = qq / M - m_ * m_;
            mu[j] = (float)m_;
            sc[j] = (float)(1.0 / sqrt(fabs(var) + 1e-4));
        }
    } else {
        const float4 a = *reinterpret_cast<const float4*>(mean + cq * 4), b = *reinterpret_cast<const float4*>(scale + cq * 4);
        mu[0] = a.x; mu[1] = a.y; mu[2] = a.z; mu[3] = a.w;
        sc[0] = b.x; sc[1] = b.y; sc[2] = b.z; sc[3] = b.w;
    }
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = fmaxf((o[j] - mu[j]) * sc[j] + 0.25f, 0.0f) + 0.1f * o[j];
            s[j] += o[j];
            q[j] += o[j] * o[j];
        }
        *reinterpret_cast<float4*>(y + (row0 + rb + i * RSTEP) * C + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    // column sums over the workgroup: lanes that share a quad, then the four waves through LDS
#pragma unroll
    for (int off = Q; off < 64; off <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[j] += __shfl_xor(s[j], off); q[j] += __shfl_xor(q[j], off); }
    const int lane = tid & 63, wave = tid >> 6;
    if (Q <= 64) {
        if (lane < Q)
#pragma unroll
            for (int j = 0; j < 4; ++j) { red[0][wave][cq * 4 + j] = s[j]; red[1][wave][cq * 4 + j] = q[j]; }
        __syncthreads();
        if (tid < C) {
            float a = 0, b = 0;
            const int nw = Q <= 64 ? (THREADS / 64) : 1;
            for (int w = 0; w < nw; ++w) { a += red[0][w][tid]; b += red[1][w][tid]; }
            // Q < 64: waves hold different rows of the same quads; Q == 64: every wave holds all quads of its rows
            if (MODE == 2) {
                const int NBG = NB >> 3, g = blockIdx.x & 7, bg = blockIdx.x >> 3;
                pout[((size_t)tid * 8 + g) * NBG + bg] = a;
                pout[(((size_t)C + tid) * 8 + g) * NBG + bg] = b;
#ifdef HANDOFF_NOFENCE
                // the fence-free form the proposal hoped for ("same XCD, same L2"): only correct while block b really runs on XCD b % 8,
                // which the platform does not promise -- compiled with -DHANDOFF_NOFENCE to price it, never a design
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // the storing threads release their partials
#endif
            } else {
                pout[(size_t)tid * NB + blockIdx.x] = a;
                pout[((size_t)C + tid) * NB + blockIdx.x] = b;
            }
        }
    }
    if (MODE == 2) {
        __shared__ int s_last;
        const int NBG = NB >> 3, g = blockIdx.x & 7;
        __syncthreads();
        if (tid == 0) {
            const unsigned want = (*step_word + 1u) * (unsigned)NBG;
            s_last = (__hip_atomic_fetch_add(counter + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == want);
        }
        __syncthreads();
        if (s_last) {
#ifndef HANDOFF_NOFENCE
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            // 16 lanes per (statistic, channel) row of NBG contiguous partials; 16 rows per pass
            const int l16 = tid & 15, r0 = tid >> 4;
            for (int r = r0; r < 2 * C; r += THREADS / 16) {
                const float* row = pout + ((size_t)r * 8 + g) * NBG;
                double acc = 0.0;
#ifdef HANDOFF_NOFENCE
                for (int b = l16; b < NBG; b += 16) acc += (double)ld_sc1(row + b);      // past this CU's L1
#else
                for (int b = l16; b < NBG; b += 16) acc += (double)row[b];
#endif
                acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 8);
                if (l16 == 0) st_sc1(rows_out + (size_t)((r / C) * 8 + g) * C + (r % C), (float)acc);
            }
        }
    }
}

__global__ void tick_kernel(unsigned* w) { *w += 1; }

template <int C, int ROWS>
void run(int M, int L) {
    const int NB = M / ROWS;
    float *x[2], *part[2], *mean, *scale, *rows;
    unsigned *counters, *step;
    for (int i = 0; i < 2; ++i) { hipMalloc(&x[i], (size_t)M * C * 4); hipMalloc(&part[i], (size_t)2 * C * NB * 4); }
    hipMalloc(&mean, (size_t)L * C * 4); hipMalloc(&scale, (size_t)L * C * 4);
    hipMalloc(&counters, L * 64 * 4); hipMalloc(&step, 4);
    hipMalloc(&rows, (size_t)(L + 1) * 16 * C * 4); hipMemset(rows, 0, (size_t)(L + 1) * 16 * C * 4);
    std::vector<float> h((size_t)M * C);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.0f - 1.0f;
    hipMemcpy(x[0], h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(part[0], 0, (size_t)2 * C * NB * 4); hipMemset(part[1], 0, (size_t)2 * C * NB * 4);
    hipMemset(mean, 0, (size_t)L * C * 4); hipMemset(scale, 0, (size_t)L * C * 4);
    hipMemset(counters, 0, L * 64 * 4); hipMemset(step, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto chain = [&](int mode) {     // 0: finalize launches, 1: hand-off, 2: no finalize at all, 3: two-level last arriver in the producer
        for (int l = 0; l < L; ++l) {
            const int a = l & 1, b = a ^ 1;
            if (mode == 0) hipLaunchKernelGGL(finalize_kernel, dim3(C / 4), dim3(THREADS), 0, 0, part[a], C, NB, M, mean + l * C, scale + l * C);
            if (mode == 1)
                hipLaunchKernelGGL((layer_kernel<1, C, ROWS>), dim3(NB), dim3(THREADS), 0, 0, x[a], x[b], part[a], mean + l * C, scale + l * C, part[b], M,
                                   counters + l * 16, step);
            else if (mode == 3)
                hipLaunchKernelGGL((layer_kernel<2, C, ROWS>), dim3(NB), dim3(THREADS), 0, 0, x[a], x[b], part[a], mean + l * C, scale + l * C, part[b], M,
                                   counters + l * 16, step, rows + (size_t)l * 16 * C, rows + (size_t)(l + 1) * 16 * C);
            else
                hipLaunchKernelGGL((layer_kernel<0, C, ROWS>), dim3(NB), dim3(THREADS), 0, 0, x[a], x[b], part[a], mean + l * C, scale + l * C, part[b], M,
                                   counters + l * 16, step);
        }
        if (mode == 1 || mode == 3) hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(1), 0, 0, step);
    };
    float res[4];
    double sums[4];
    for (int mode = 0; mode < 4; ++mode) {
        hipMemcpy(x[0], h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemset(counters, 0, L * 64 * 4); hipMemset(step, 0, 4);
        for (int i = 0; i < 3; ++i) chain(mode);
        hipDeviceSynchronize();
        const int reps = 10;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) chain(mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        res[mode] = ms * 1e3f / (reps * L);
        std::vector<float> m(C);
        sums[mode] = 0;
        if (mode == 3) {                       // the published group rows of the last layer: sum over groups / M = the mean
            std::vector<float> r8((size_t)8 * C);
            hipMemcpy(r8.data(), rows + (size_t)(L - 1) * 16 * C, (size_t)8 * C * 4, hipMemcpyDeviceToHost);
            for (int c = 0; c < C; ++c) { double t = 0; for (int g = 0; g < 8; ++g) t += r8[(size_t)g * C + c]; sums[mode] += (float)(t / M); }
        } else {
            hipMemcpy(m.data(), mean + (L - 1) * C, C * 4, hipMemcpyDeviceToHost);
            for (int c = 0; c < C; ++c) sums[mode] += m[c];
        }
    }
    // the hand-off must give the same coefficients as the finalize launches (same arithmetic, same order): checksum of the last layer's means
    printf("M %7d C %3d rows/wg %3d (%4d wgs, %5.1f MB/tensor): finalize launch %6.2f us/layer | in-consumer hand-off %6.2f | no finalize %6.2f | "
           "two-level last arriver %6.2f   [mean checksum %.6f vs %.6f (hand-off) vs %.6f (two-level)]\n", M, C, ROWS, NB, M * C * 4 / 1e6, res[0], res[1],
           res[2], res[3], sums[0], sums[1], sums[3]);
    for (int i = 0; i < 2; ++i) { hipFree(x[i]); hipFree(part[i]); }
    hipFree(mean); hipFree(scale); hipFree(counters); hipFree(step); hipFree(rows);
}

int main() {
    const int L = 60;
    run<64, 64>(131072, L);      // stage 1 exit tensors: 2048 workgroups, 33.5 MB
    run<16, 128>(131072, L);     // stage 1 bottleneck tensors: 1024 workgroups, 8.4 MB
    run<128, 64>(32768, L);      // stage 2
    run<64, 32>(8192, L);        // stages 3-4 bottleneck: 256 workgroups, 2 MB
    run<256, 64>(8192, L);       // stages 3-4 exit: 128 workgroups, 8.4 MB
    return 0;
}
