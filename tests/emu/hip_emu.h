// tests/emu/hip_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A host-side SIMT emulator used to run the UNMODIFIED kernel sources of
// deep-prior-pp_amd/csrc/*.hip on the CPU of the build container (which has no
// GPU).  The sources are compiled as plain C++ with
//     clang++ -x c++ -include tests/emu/hip_emu.h -Itests/emu/stub ...
// and linked into tests/emu/_build/libdpp_emu.so, which exports the same C ABI
// as the product library.  Only the `-m "not gpu"` kernel-logic tests load it;
// the product never does (it fails loudly when the HIP library is missing).
//
// Model: every thread of a workgroup is a ucontext fiber; fibers run round-robin
// and switch only at __syncthreads() and at wave collectives (shuffles, MFMA).
// A wavefront is 64 lanes.  MFMA lane layouts follow cdna_hip_programming.md s.3:
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15, row=(l>>4)*4+r
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,
//                 row=(r&3)+8*(r>>2)+4*(l>>5)
// with a k-ordered fmaf chain (bitwise what the hardware computes).
#pragma once
#include <setjmp.h>
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu::g_dyn_smem);

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

namespace emu {
struct Fiber {
    ucontext_t ctx;
    jmp_buf jb;
    char* stack = nullptr;
    bool done = false;
    bool started = false;
    dim3 tid;
    int lin = 0;
};
struct Wave {
    alignas(16) unsigned char buf[64][160];
    int count = 0, gen = 0, alive = 0;
};
inline ucontext_t g_sched;
inline jmp_buf g_sched_jb;
inline Fiber* g_cur = nullptr;
inline std::vector<Fiber> g_fibers;
inline std::vector<Wave> g_waves;
inline std::function<void()>* g_body = nullptr;
inline dim3 g_bid, g_bdim, g_gdim;
inline int g_alive = 0, g_bar_count = 0, g_bar_gen = 0;
inline unsigned long g_progress = 0;
inline unsigned char* g_dyn_smem = nullptr;
inline int g_last_error = 0;
constexpr size_t kStack = 96 * 1024;

// Fibers are entered once through makecontext/setcontext; every later switch is a _setjmp/_longjmp pair (no signal-mask
// system call), which is what makes barrier-heavy kernels emulate in reasonable time.
inline void yield() { if (_setjmp(g_cur->jb) == 0) _longjmp(g_sched_jb, 1); }
inline void resume(Fiber* f) {
    if (_setjmp(g_sched_jb) == 0) {
        if (!f->started) { f->started = true; setcontext(&f->ctx); }
        else _longjmp(f->jb, 1);
    }
}

inline void fiber_entry() {
    Fiber* f = g_cur;
    (*g_body)();
    f->done = true;
    g_alive--;
    g_waves[f->lin >> 6].alive--;
    g_progress++;
    if (g_bar_count > 0 && g_bar_count == g_alive) { g_bar_count = 0; g_bar_gen++; }
    _longjmp(g_sched_jb, 1);
}

inline void syncthreads() {
    int gen = g_bar_gen;
    if (++g_bar_count == g_alive) { g_bar_count = 0; g_bar_gen++; g_progress++; }
    else while (g_bar_gen == gen) yield();
}

inline void wave_barrier(Wave& w) {
    int gen = w.gen;
    if (++w.count == w.alive) { w.count = 0; w.gen++; g_progress++; }
    else while (w.gen == gen) yield();
}

// Deposit `n` bytes for this lane, wait for the whole wave, let `reader` look at all lanes, wait again.
template <class Reader>
inline void wave_exchange(const void* in, size_t n, Reader reader) {
    Fiber* f = g_cur;
    Wave& w = g_waves[f->lin >> 6];
    int lane = f->lin & 63;
    if (n > sizeof(w.buf[0])) { fprintf(stderr, "emu: exchange too large\n"); abort(); }
    memcpy(w.buf[lane], in, n);
    wave_barrier(w);
    reader(w, lane);
    wave_barrier(w);
}

inline void run_block(std::function<void()>& body, dim3 bid, dim3 bdim, dim3 gdim) {
    int n = bdim.x * bdim.y * bdim.z;
    if ((int)g_fibers.size() < n) {
        size_t old = g_fibers.size();
        g_fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) g_fibers[i].stack = (char*)malloc(kStack);
    }
    g_waves.assign((n + 63) / 64, Wave());
    g_body = &body; g_bid = bid; g_bdim = bdim; g_gdim = gdim;
    g_alive = n; g_bar_count = 0; g_bar_gen = 0;
    for (int i = 0; i < n; ++i) {
        Fiber& f = g_fibers[i];
        f.done = false; f.started = false; f.lin = i;
        f.tid = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
        g_waves[i >> 6].alive++;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &g_sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int stalled = 0;
    while (g_alive > 0) {
        unsigned long before = g_progress;
        for (int i = 0; i < n; ++i) {
            if (g_fibers[i].done) continue;
            g_cur = &g_fibers[i];
            resume(&g_fibers[i]);
        }
        if (g_progress == before) {
            if (++stalled > 4) { fprintf(stderr, "emu: deadlock (divergent barrier / collective) in block (%u,%u,%u)\n", bid.x, bid.y, bid.z); abort(); }
        } else stalled = 0;
    }
    g_cur = nullptr;
}

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
    static std::vector<unsigned char> dyn;
    if (dyn.size() < shmem + 64) dyn.resize(shmem + 64);
    g_dyn_smem = (unsigned char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
    if (shmem > 160 * 1024) { fprintf(stderr, "emu: LDS request %zu > 160 KiB\n", shmem); g_last_error = hipErrorLaunchFailure; return; }
    if (block.x * block.y * block.z > 1024 || grid.x == 0 || grid.y == 0 || grid.z == 0) { g_last_error = hipErrorInvalidValue; return; }
    std::function<void()> body = [=]() { kernel(static_cast<KArgs>(args)...); };
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) run_block(body, dim3(x, y, z), block, grid);
}
}  // namespace emu

// events / graphs: the emulator executes every launch synchronously, so events are no-ops; there is no graph executor
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum { hipEventDisableTiming = 2, hipEventReleaseToDevice = 0x40000000, hipErrorNotSupported = 801 };
struct hipKernelNodeParams { void* func; dim3 gridDim, blockDim; unsigned sharedMemBytes; void** kernelParams; void** extra; };
struct hipMemsetParams { void* dst; unsigned value, elementSize; size_t width, height, pitch; };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipGraphCreate(hipGraph_t*, unsigned) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphAddKernelNode(hipGraphNode_t*, hipGraph_t, const hipGraphNode_t*, size_t, const hipKernelNodeParams*) { return hipErrorNotSupported; }
static inline hipError_t hipGraphAddMemsetNode(hipGraphNode_t*, hipGraph_t, const hipGraphNode_t*, size_t, const hipMemsetParams*) { return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipExtLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t, hipEvent_t, int) { return hipErrorNotSupported; }
#define DPP_HIP_EMU 1

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_bid)
#define blockDim (emu::g_bdim)
#define gridDim (emu::g_gdim)
#define __syncthreads() emu::syncthreads()
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)

static inline hipError_t hipGetLastError() { int e = emu::g_last_error; emu::g_last_error = 0; return e; }
static inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess" : "emu error"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {       // distinct handles; everything runs in issue order
    static char handles[64];
    static int n = 0;
    *s = &handles[(n++) & 63];
    return hipSuccess;
}

// ---- wave collectives -------------------------------------------------------------------------
template <class T> static inline T emu_shfl_idx(T v, int srcLaneFn(int, int, int), int arg, int width) {
    T out = v;
    emu::wave_exchange(&v, sizeof(T), [&](emu::Wave& w, int lane) {
        int src = srcLaneFn(lane, arg, width);
        if (src >= 0 && src < 64) memcpy(&out, w.buf[src], sizeof(T));
    });
    return out;
}
static inline int emu_src_xor(int lane, int m, int width) { int s = lane ^ m; return (s / width == lane / width) ? s : lane; }
static inline int emu_src_down(int lane, int d, int width) { int s = lane + d; return (s / width == lane / width) ? s : lane; }
static inline int emu_src_up(int lane, int d, int width) { int s = lane - d; return (s >= 0 && s / width == lane / width) ? s : lane; }
static inline int emu_src_abs(int lane, int j, int width) { return (lane / width) * width + (j % width); }
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) { return emu_shfl_idx(v, emu_src_xor, m, width); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) { return emu_shfl_idx(v, emu_src_down, d, width); }
template <class T> static inline T __shfl_up(T v, int d, int width = 64) { return emu_shfl_idx(v, emu_src_up, d, width); }
template <class T> static inline T __shfl(T v, int j, int width = 64) { return emu_shfl_idx(v, emu_src_abs, j, width); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }

typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));

static inline emu_f32x4 emu_mfma_16x16x4(float a, float b, emu_f32x4 c, int, int, int) {
    struct { float a, b; } in = {a, b};
    emu_f32x4 d = c;
    emu::wave_exchange(&in, sizeof(in), [&](emu::Wave& w, int lane) {
        int col = lane & 15;
        for (int r = 0; r < 4; ++r) {
            int row = (lane >> 4) * 4 + r;
            float acc = c[r];
            for (int k = 0; k < 4; ++k) {
                float av, bv;
                memcpy(&av, w.buf[row + 16 * k], 4);
                memcpy(&bv, w.buf[col + 16 * k] + 4, 4);
                acc = fmaf(av, bv, acc);
            }
            d[r] = acc;
        }
    });
    return d;
}
static inline emu_f32x16 emu_mfma_32x32x2(float a, float b, emu_f32x16 c, int, int, int) {
    struct { float a, b; } in = {a, b};
    emu_f32x16 d = c;
    emu::wave_exchange(&in, sizeof(in), [&](emu::Wave& w, int lane) {
        int col = lane & 31;
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float acc = c[r];
            for (int k = 0; k < 2; ++k) {
                float av, bv;
                memcpy(&av, w.buf[row + 32 * k], 4);
                memcpy(&bv, w.buf[col + 32 * k] + 4, 4);
                acc = fmaf(av, bv, acc);
            }
            d[r] = acc;
        }
    });
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_16x16x4
// 16x16x32 bf16: lane l holds A[i = l & 15][k = 8 * (l >> 4) + j] and B[k = 8 * (l >> 4) + j][col = l & 15], j = 0..7; the
// products are exact in f32 and the hardware accumulates them in f32 (summation order inside one instruction is not
// architecturally defined; k-ordered here)
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x4 emu_mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    struct { float a[8], b[8]; } in;
    for (int j = 0; j < 8; ++j) { in.a[j] = (float)a[j]; in.b[j] = (float)b[j]; }
    emu_f32x4 d = c;
    emu::wave_exchange(&in, sizeof(in), [&](emu::Wave& w, int lane) {
        int col = lane & 15;
        for (int r = 0; r < 4; ++r) {
            int row = (lane >> 4) * 4 + r;
            float acc = c[r];
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < 8; ++j) {
                    float av, bv;
                    memcpy(&av, w.buf[row + 16 * g] + 4 * j, 4);
                    memcpy(&bv, w.buf[col + 16 * g] + 32 + 4 * j, 4);
                    acc = fmaf(av, bv, acc);
                }
            d[r] = acc;
        }
    });
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_32x32x2

// ---- atomics / misc device functions ------------------------------------------------------------
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicInc(unsigned* p, unsigned lim) { unsigned o = *p; *p = (o >= lim) ? 0 : o + 1; return o; }
static inline void __threadfence() {}
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __fdividef(float a, float b) { return a / b; }
template <class T> static inline T __ldg(const T* p) { return *p; }
static inline int __float2int_rn(float x) { return (int)lrintf(x); }
static inline int __double2int_rn(double x) { return (int)lrint(x); }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
