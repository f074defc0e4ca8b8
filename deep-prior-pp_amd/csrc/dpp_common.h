// dpp_common.h -- shared device helpers for the gfx950 kernels of the DeepPrior++ hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dpp_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- bf16 STORAGE of activation tensors (BASELINE config 5, include/dpp_hip.h DPP_ST_*) ---------------------------------------------
// The [pixels][channels] tensors the convolutions write may be held as bfloat16 (round-to-nearest-even of the f32 value the epilogue
// formed; BatchNorm statistics still come from the f32 values): half the bytes on every pass over them.  Kernels address such a
// tensor with the SAME element offsets as an f32 one and go through dpp_ld4 / dpp_st4: four consecutive elements as 16 (f32) or 8
// (bf16) bytes.
typedef __bf16 dpp_bf16;
typedef __bf16 dpp_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 dpp_bf16x8 __attribute__((ext_vector_type(8)));

template <class T> __device__ __forceinline__ float4 dpp_ld4(const T* p);
template <> __device__ __forceinline__ float4 dpp_ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 dpp_ld4<dpp_bf16>(const dpp_bf16* p) {
    const dpp_bf16x4 v = *reinterpret_cast<const dpp_bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <class T> __device__ __forceinline__ float dpp_ld1(const T* p) { return (float)p[0]; }
template <class T> __device__ __forceinline__ void dpp_st4(T* p, float4 v);
template <> __device__ __forceinline__ void dpp_st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void dpp_st4<dpp_bf16>(dpp_bf16* p, float4 v) {
    dpp_bf16x4 o;
    o[0] = (dpp_bf16)v.x; o[1] = (dpp_bf16)v.y; o[2] = (dpp_bf16)v.z; o[3] = (dpp_bf16)v.w;
    *reinterpret_cast<dpp_bf16x4*>(p) = o;
}
// the value a bf16 store leaves behind, as f32 (what every later reader sees)
__device__ __forceinline__ float dpp_bf16_round(float v) { return (float)(dpp_bf16)v; }
// Raw / widen pair for software-pipelined operand staging: the 8-byte load of four bf16 elements lands in .x / .y of a float4 register
// set UNCONVERTED (a conversion at the load would make the wave wait for the data right where the load was issued), and is widened
// when the chunk is committed to LDS.  Element 0 sits in the low half of .x (little endian).
__device__ __forceinline__ float4 dpp_raw8(const void* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), 0.0f, 0.0f);
}
__device__ __forceinline__ float4 dpp_widen4(const float4& raw) {
    const unsigned x = __float_as_uint(raw.x), y = __float_as_uint(raw.y);
    return make_float4(__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u), __uint_as_float(y << 16), __uint_as_float(y & 0xffff0000u));
}
// run-time typed forms for the epilogues (one uniform branch per access; not for inner loops): `p` addresses f32 or bf16 elements
__device__ __forceinline__ float4 dpp_ld4_rt(const float* p, size_t off, bool b16) {
    return b16 ? dpp_ld4(reinterpret_cast<const dpp_bf16*>(p) + off) : dpp_ld4(p + off);
}
__device__ __forceinline__ float dpp_ld1_rt(const float* p, size_t off, bool b16) {
    return b16 ? (float)reinterpret_cast<const dpp_bf16*>(p)[off] : p[off];
}

#define DPP_WAVE 64
#define DPP_THREADS 256

// ---- launch recording (dpp_plan_*, include/dpp_hip.h) ------------------------------------------------------------------
// Every kernel launch of the library goes through DPP_LAUNCH.  Normally that is a plain launch on `stream`; while the calling
// thread records a plan (plan.hip) the fully resolved launch -- kernel, grid, block, LDS bytes and a private copy of the
// arguments -- is appended to the plan instead, to be re-issued later from C++ (or as a hipGraph kernel node).
#include <functional>
#include <memory>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

struct dpp_plan_node {
    int kind = 0;                                          // 0 kernel, 1 memset, 2 fork, 3 join
    int lane = 0;                                          // 0 main, 1 side
    std::function<hipError_t(hipStream_t)> issue;          // re-issues the launch on a stream
    const void* func = nullptr;                            // raw form for hipGraphAddKernelNode
    dim3 grid, block;
    size_t shmem = 0;
    std::shared_ptr<void> argstore;
    std::vector<void*> argptrs;
    void* ptr = nullptr;                                   // memset target
    size_t nbytes = 0;
};
struct dpp_plan;
extern thread_local dpp_plan* dpp_tls_plan;                // the plan this thread is recording into (plan.hip)
void dpp_plan_append(dpp_plan* plan, dpp_plan_node&& node);

template <class Tuple, size_t... I>
static inline void dpp_arg_pointers(Tuple& t, std::vector<void*>& out, std::index_sequence<I...>) {
    (out.push_back(const_cast<void*>(static_cast<const void*>(&std::get<I>(t)))), ...);
}

template <class... KA, class... A>
static inline void dpp_launch(void (*kernel)(KA...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, A&&... args) {
    static_assert(sizeof...(KA) == sizeof...(A), "kernel / argument count mismatch");
    if (dpp_tls_plan == nullptr) {
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, static_cast<KA>(args)...);
        return;
    }
    using Tup = std::tuple<std::remove_cv_t<KA>...>;
    auto tup = std::make_shared<Tup>(static_cast<KA>(args)...);
    dpp_plan_node n;
    n.func = reinterpret_cast<const void*>(kernel);
    n.grid = grid;
    n.block = block;
    n.shmem = shmem;
    dpp_arg_pointers(*tup, n.argptrs, std::index_sequence_for<KA...>{});
    n.issue = [kernel, grid, block, shmem, tup](hipStream_t s) -> hipError_t {
        std::apply([&](auto&... a) { hipLaunchKernelGGL(kernel, grid, block, shmem, s, a...); }, *tup);
        return hipGetLastError();
    };
    n.argstore = tup;
    dpp_plan_append(dpp_tls_plan, std::move(n));
}
#define DPP_LAUNCH(kernel, grid, block, shmem, stream, ...) dpp_launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, __VA_ARGS__)

static inline int dpp_launch_status() {
    if (dpp_tls_plan != nullptr) return DPP_OK;            // recorded, not launched
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DPP_OK : (int)e;
}

static inline int dpp_cdiv(int a, int b) { return (a + b - 1) / b; }

// Phase stamps for tools/phase_profile.py: compiled in only in the profiling build of the library (-DDPP_KERNEL_PROF, a separate
// .so that the product never loads); in the product build the call is empty.  slot s of workgroup w lands in buf[w * 16 + s] as a
// 100 MHz wall-clock tick (s_memrealtime: comparable across CUs).
#ifdef DPP_KERNEL_PROF
__device__ __forceinline__ void dpp_stamp(unsigned long long* buf, int slot) {
    if (buf != nullptr && threadIdx.x == 0) {
        const size_t w = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        buf[w * 16 + slot] = __builtin_amdgcn_s_memrealtime();
    }
}
#else
__device__ __forceinline__ void dpp_stamp(unsigned long long*, int) {}
#endif
extern unsigned long long* dpp_prof_buffer;                // host-side: where instrumented launches stamp (NULL = nowhere); plan.hip

// Kernel arguments are passed by value (descriptors of 300-450 bytes = 5-8 cache lines of the kernarg segment).  The compiler
// fetches the fields where they are first used -- dozens of s_load / s_waitcnt pairs spread over the entry code -- and the first
// wave of a workgroup on a CU finds none of the lines in the scalar cache: a dispatch's kernarg segment is fresh memory, every line
// is a miss to HBM taken one after the other.  This touches one dword per 64-byte line with back-to-back scalar loads, so the misses
// overlap (one latency instead of one per line) and the field loads that follow hit the scalar cache.
// occupancy goal of a kernel for the register allocator / scheduler (the host build against the SIMT emulator has no such thing)
#ifdef DPP_HIP_EMU
#define DPP_WAVES_PER_EU(lo, hi)
#else
#define DPP_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

// Nothing may be scheduled across this point: the LLVM machine scheduler otherwise sinks loads that were issued early ON PURPOSE
// (a software prefetch ring) down to their first use to save registers, which serialises the memory round trips again.
#ifdef DPP_HIP_EMU
#define DPP_SCHED_FENCE()
#else
#define DPP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Instruction-issue priority of the calling wave among the waves of its SIMD (0 .. 3; hardware hint, nothing for the emulator).
#ifdef DPP_HIP_EMU
#define DPP_SETPRIO(p)
#else
#define DPP_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#endif

// All lanes of a wave have executed what precedes before any executes what follows (LDS traffic between the lanes of ONE wave: the
// hardware runs a wave's LDS instructions in order, so this only has to stop the compiler; the emulator's fibers meet in a shuffle).
#ifdef DPP_HIP_EMU
#define DPP_WAVE_SYNC() ((void)__shfl_xor(0, 1))
#else
#define DPP_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

template <int NBYTES>
__device__ __forceinline__ void dpp_kernarg_warm() {
#ifndef DPP_HIP_EMU
    typedef __attribute__((address_space(4))) const int kint;
    kint* ka = (kint*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int L = (NBYTES + 63) / 64;
    int t[L];
#pragma unroll
    for (int i = 0; i < L; ++i) t[i] = ka[i * 16];
#pragma unroll
    for (int i = 0; i < L; ++i) asm volatile("" ::"s"(t[i]));
#endif
}

// Row map of a compact (N,Ho,Wo) pixel index onto a (N,Hi,Wi) map sampled with stride s.
__device__ __forceinline__ int dpp_map_row(const dpp_rowmap& m, int r) {
    if (m.s == 1) return r;
    int n = r / m.HoWo;
    int q = r - n * m.HoWo;
    int y = q / m.Wo;
    int x = q - y * m.Wo;
    return n * m.HiWi + (y * m.s) * m.Wi + x * m.s;
}

// (x - mean) * scale + beta of the operand prologues as ONE fused multiply-add, spelled out: the device compiler contracts the expression
// anyway (so this changes nothing there), the host build against the SIMT emulator does not -- and tests/pinning.py rebuilds the kernels'
// bf16-rounded operands with exactly this arithmetic (an unfused evaluation flips the bfloat16 rounding of an element in 1e4).
__device__ __forceinline__ float dpp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// The fused BatchNorm(+ReLU) operand prologue: v = (x - mean) * scale + beta ; relu.
// Written as (x - mean) * scale + beta (not x*s + t) so that no bits are lost when |mean| >> std.
__device__ __forceinline__ float dpp_act1(float x, const dpp_act& a, int c) {
    float v = x;
    if (a.mode & 2) v = dpp_fma(x - a.mean[c], a.scale[c], a.beta[c]);
    if (a.mode & 1) v = fmaxf(v, 0.0f);
    return v;
}

__device__ __forceinline__ float4 dpp_act4(float4 x, const dpp_act& a, int c) {
    float4 v = x;
    if (a.mode & 2) {
        const float4 mu = *reinterpret_cast<const float4*>(a.mean + c);
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
        const float4 be = *reinterpret_cast<const float4*>(a.beta + c);
        v.x = dpp_fma(x.x - mu.x, sc.x, be.x);
        v.y = dpp_fma(x.y - mu.y, sc.y, be.y);
        v.z = dpp_fma(x.z - mu.z, sc.z, be.z);
        v.w = dpp_fma(x.w - mu.w, sc.w, be.w);
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

// Layout of the per-workgroup BatchNorm partials (forward statistics (mean, M2) and backward sums (sum G, sum G*xhat)):
// partial[s][c][b], s in {0, 1}, c the channel, b the row block (= blockIdx.x of the producer, NB = gridDim.x of them).
// Block-fastest, so that the finalize kernels -- one wave per channel, lanes over blocks -- read 256 contiguous bytes per
// load; with [b][s][c] every lane of such a load touched its own cache line and a 2048-block finalize took 10-20 us.
__device__ __forceinline__ size_t dpp_partial_index(int s, int c, int b, int C, int NB) { return ((size_t)s * C + c) * NB + b; }


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

// The MFMA-tile epilogue (bias / residual / fused BatchNorm-backward mask / column statistics) with 16-byte global accesses.  In the MFMA D layout a lane owns single floats of 4 different rows, so
// a direct epilogue reads residual / bn_x and writes C 4 bytes per lane -- about half the bandwidth the memory
// pipeline gives 16-byte accesses, which is what bounds the GEMMs with wide outputs (expanding 1x1 convolutions and their
// data gradients move 3-5 bytes through the epilogue for every byte of operand).  Here the waves drop their accumulators
// into an LDS image of the BM x BN tile, and then every thread owns ONE column quad and walks down the rows: float4 in,
// float4 out, the per-column coefficients in registers, and the column sums reduce over lanes first (xor shuffles between
// the lanes that share a quad) and over the four waves through LDS.  Requires N % 4 == 0, row offsets that are multiples of 4
// and 16-byte aligned C / residual / bn_x.  `smem` must hold BM*(BN+4) + 16*BN floats.
// `rowoff(rl)` gives the element offset of tile row rl in C / residual / bn_x, or -1 for a row outside the problem; `nvalid` is
// the number of rows it accepts (the divisor of the block mean).
// The per-column coefficients of a thread's quad.  Kernels load them BEFORE their K loop: the loads are independent of everything
// else, and issued late they put one more memory latency on the critical path of a short workgroup.
struct dpp_wide_coef {
    float cbias[4], cmean[4], cscale[4], cbeta[4], cistd[4];
    float4 raw[5];
    bool on_bias, on_bn;
    // `safe`: any valid 16-byte aligned device address (the kernel's output pointer).  Every vector is fetched with ONE unconditional
    // 16-byte load -- from `safe` when the vector is absent or the quad lies outside the problem -- and masked by finish(), which
    // the epilogue calls: nothing between the loads and the K loop touches the loaded registers, so the wave does not wait for
    // them in its entry code.  The earlier form (aligned 16-byte load OR four scalar loads, chosen at run time, values used on the
    // spot) made the compiler drain all outstanding loads in front of each vector (both paths write the same registers): up to five
    // serialized memory round trips at the start of every kernel with the BatchNorm-backward epilogue, one in the others.  The host
    // only takes the wide path with 16-byte aligned vectors.
    template <int BN>
    __device__ __forceinline__ void load(int col0, int N, const float* bias, const dpp_epilogue& ep, const float* safe) {
        const int col = col0 + ((int)threadIdx.x % (BN / 4)) * 4;
        const bool cin = col < N;                            // N % 4 == 0 on this path: a quad is inside or outside as a whole
        on_bias = cin && bias != nullptr;
        on_bn = cin && ep.bn_x != nullptr;
        raw[0] = *reinterpret_cast<const float4*>(on_bias ? bias + col : safe);
        raw[1] = *reinterpret_cast<const float4*>(on_bn ? ep.bn_mean + col : safe);
        raw[2] = *reinterpret_cast<const float4*>(on_bn ? ep.bn_scale + col : safe);
        raw[3] = *reinterpret_cast<const float4*>(on_bn ? ep.bn_beta + col : safe);
        raw[4] = *reinterpret_cast<const float4*>(on_bn ? ep.bn_inv_std + col : safe);
    }
    __device__ __forceinline__ void finish() {
        auto quad = [](const float4& v, bool on, float (&dst)[4]) {
            dst[0] = on ? v.x : 0.0f; dst[1] = on ? v.y : 0.0f; dst[2] = on ? v.z : 0.0f; dst[3] = on ? v.w : 0.0f;
        };
        quad(raw[0], on_bias, cbias);
        quad(raw[1], on_bn, cmean);
        quad(raw[2], on_bn, cscale);
        quad(raw[3], on_bn, cbeta);
        quad(raw[4], on_bn, cistd);
    }
};

// NIMG > 1 (the K-split kernel): the tile is the sum of NIMG images -- every wave holds a partial sum of the WHOLE tile over its K
// slice (WM = WN = 1) and drops it into image `img`; the images are added in a fixed order when the tile is read back.  `smem` must
// then hold NIMG*BM*(BN+4) + 16*BN floats.
// `store` (DPP_ST_C: C and residual, DPP_ST_BNX: bn_x) marks the tensors held as bf16: the value written is rounded on the store, the
// statistics are formed from the UNROUNDED f32 values (see oracle/torch_ref.py: stats of v, normalisation of round(v)).
// ST (compile time): the instantiation may meet bf16-stored tensors at all.  The float32 instantiations (ST = false) carry none of the
// run-time type tests: round 4 measured them at +4 % of the whole fp32 step when every kernel had them (3.50 -> 3.65 ms).
template <int RM, int CN, int WM, int WN, int BM, int BN, int NIMG = 1, bool ST = false, class RowOff>
__device__ __forceinline__ void dpp_epilogue_wide(f32x4 (&acc)[RM][CN], float* smem, int col0, int N, dpp_wide_coef& co,
                                                  const float* residual, float* C, const dpp_epilogue& ep, int nvalid, int wm,
                                                  int wn, int l15, int kq, RowOff rowoff, int img = 0, int store = 0, int blk = -1, int nblk = 0) {
    // (blk, nblk): the row block this tile is in the statistics partials -- by default the workgroup's own index; a workgroup that walks
    // several tiles (conv3x3_p_kernel) passes the tile's
    const int pblk = blk >= 0 ? blk : (int)blockIdx.x, pnblk = blk >= 0 ? nblk : (int)gridDim.x;
    const bool c16 = ST && (store & DPP_ST_C) != 0, x16 = ST && (store & DPP_ST_BNX) != 0;
    co.finish();
    constexpr int LDT = BN + 4;
    constexpr int Q = BN / 4;                                // column quads per tile row
    constexpr int RSTEP = DPP_THREADS / Q;                   // rows covered by one sweep of the workgroup
    constexpr int ITERS = (BM + RSTEP - 1) / RSTEP;
    float* Ts = smem;
    float* red = smem + NIMG * BM * LDT;                     // 4 regions of 4*BN floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cq = tid % Q, rb = tid / Q;
    const int col = col0 + cq * 4;
    const bool cin = col < N;                                // N % 4 == 0: the quad is inside or outside as a whole
    const bool bn = ep.bn_x != nullptr;
    // What the row sweep reads from memory -- residual and BatchNorm input of a row -- is requested TWO rows ahead: the first two rows' here, ahead
    // of the accumulators' trip through LDS, row it + 2's behind row it's store.  Unconditional loads (a row outside the problem reads element 0),
    // raw 16 / 8 bytes, converted where they are used.  With the load at the top of its own row, each one sat behind the previous row's store to C
    // (which may alias it as far as the compiler knows) and was waited for on the spot: ITERS serialized memory round trips at the end of every
    // data-gradient launch (2 for 16-column tiles, 4 for 32 / 64 columns).
    long rofs[ITERS];
    float4 res32[ITERS], bnx32[ITERS];
    uint2 res16[ITERS], bnx16[ITERS];
    auto request = [&](int it) {
        const int rl = rb + it * RSTEP;
        rofs[it] = (rl < BM && cin) ? rowoff(rl) : -1L;
        const size_t o = rofs[it] >= 0 ? (size_t)rofs[it] + col : (size_t)0;
        if (residual) {
            if (c16) res16[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const dpp_bf16*>(residual) + o);
            else res32[it] = *reinterpret_cast<const float4*>(residual + o);
        }
        if (bn) {
            if (x16) bnx16[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const dpp_bf16*>(ep.bn_x) + o);
            else bnx32[it] = *reinterpret_cast<const float4*>(ep.bn_x + o);
        }
    };
    request(0);
    if constexpr (ITERS > 1) request(1);
    auto widen = [](const uint2& r) {                        // four bf16 -> four f32, exact
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    };
    // (the K loop ended on a barrier, so nobody reads the operand tiles any more)
    {
        float* Ti = Ts + (NIMG > 1 ? img * BM * LDT : 0);
#pragma unroll
        for (int rt = 0; rt < RM; ++rt)
#pragma unroll
            for (int ct = 0; ct < CN; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ti[(wm * (BM / WM) + rt * 16 + kq * 4 + r) * LDT + wn * (BN / WN) + ct * 16 + l15] = acc[rt][ct][r];
    }
    __syncthreads();

    float vals[ITERS][4];
    bool valid[ITERS];
    float sx[4] = {0.f, 0.f, 0.f, 0.f}, sy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int rl = rb + it * RSTEP;
        const long ro = rofs[it];
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (ro >= 0) {
            float4 t = *reinterpret_cast<const float4*>(&Ts[rl * LDT + cq * 4]);
#pragma unroll
            for (int im = 1; im < NIMG; ++im) {
                const float4 u = *reinterpret_cast<const float4*>(&Ts[(im * BM + rl) * LDT + cq * 4]);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            v[0] = t.x + co.cbias[0]; v[1] = t.y + co.cbias[1]; v[2] = t.z + co.cbias[2]; v[3] = t.w + co.cbias[3];
            const size_t o = (size_t)ro + col;
            if (residual) {
                const float4 rr = c16 ? widen(res16[it]) : res32[it];
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
            if (bn) {
                const float4 xx = x16 ? widen(bnx16[it]) : bnx32[it];
                const float x[4] = {xx.x, xx.y, xx.z, xx.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float dx = x[j] - co.cmean[j];
                    if (ep.bn_relu && dx * co.cscale[j] + co.cbeta[j] < 0.0f) v[j] = 0.0f;
                    // a bf16-stored gradient: the sums are those of the values AS STORED, so that the c1 / c2 bn_bwd_apply gets are
                    // exactly the means of the G it reads (masked elements have dX = -scale (c1 + xhat c2): they see nothing else)
                    if (c16) v[j] = dpp_bf16_round(v[j]);
                    sx[j] += v[j];
                    sy[j] += v[j] * (dx * co.cistd[j]);
                }
            }
            if (c16) dpp_st4(reinterpret_cast<dpp_bf16*>(C) + o, make_float4(v[0], v[1], v[2], v[3]));
            else *reinterpret_cast<float4*>(C + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
        valid[it] = ro >= 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) vals[it][j] = v[j];
        if (it + 2 < ITERS) request(it + 2);
    }
    const bool do_bn = bn && ep.bn_partial != nullptr;
    // lanes of a wave that share a quad differ in the bits >= log2(Q)
    auto lanesum = [&](float (&s)[4]) {
#pragma unroll
        for (int off = Q; off < 64; off <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], off);
    };
    auto publish = [&](const float (&s)[4], float* region) {
        if (lane < Q) *reinterpret_cast<float4*>(&region[wave * BN + cq * 4]) = make_float4(s[0], s[1], s[2], s[3]);
    };
    auto collect = [&](float (&s)[4], const float* region) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(&region[w * BN + cq * 4]);
            s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
        }
    };
    if (do_bn) {
        lanesum(sx);
        lanesum(sy);
        publish(sx, red);
        publish(sy, red + 4 * BN);
        __syncthreads();
        collect(sx, red);
        collect(sy, red + 4 * BN);
        if (tid < Q && cin) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ep.bn_partial[dpp_partial_index(0, col + j, pblk, N, pnblk)] = sx[j];
                ep.bn_partial[dpp_partial_index(1, col + j, pblk, N, pnblk)] = sy[j];
            }
        }
    }
    if (ep.stats != nullptr) {
        // block mean first, then M2 about it (no cancellation when |mean| >> std).  A one-pass version about a shared pivot (the
        // column's value in tile row 0) saves a reduction round but measured no time and cost accuracy: the bs256 training-mode
        // forward moved from < 1e-5 to 1.1e-5 of the output scale against the float64 oracle.
        float sm[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < ITERS; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) sm[j] += vals[it][j];
        lanesum(sm);
        publish(sm, red + 8 * BN);
        __syncthreads();
        collect(sm, red + 8 * BN);
        const float inv_n = 1.0f / (float)nvalid;          // exact for the power-of-two tiles; one division instead of four
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[j] = sm[j] * inv_n;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (valid[it]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float dv = vals[it][j] - sm[j];
                    m2[j] += dv * dv;
                }
            }
        }
        lanesum(m2);
        publish(m2, red + 12 * BN);
        __syncthreads();
        collect(m2, red + 12 * BN);
        if (tid < Q && cin) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ep.stats[dpp_partial_index(0, col + j, pblk, N, pnblk)] = sm[j];
                ep.stats[dpp_partial_index(1, col + j, pblk, N, pnblk)] = m2[j];
            }
        }
    }
}
