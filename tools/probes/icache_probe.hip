// icache_probe.hip -- what does it cost a workgroup to run code it has never run before?  The conv / GEMM kernels of the train step are
// 10..75 KB of machine code each, every workgroup walks its path through them ONCE, and consecutive launches are different kernels:
// if instruction fetch from a cold instruction cache is slow, it is part of the ~5 us fixed cost per launch.
//   straight<KB>: KB*256 four-byte scalar instructions in a straight line (KB kilobytes of code, each executed once);
//   looped:       the same instruction count as a 1 KB body executed KB times (hot after the first trip).
// Timed GPU-side (s_memrealtime, 100 MHz) from the first instruction of the first wave to the last instruction of the last wave.
// "same kernel again": the launch before was the same kernel; "after a different kernel": the launch before was another 64 KB kernel
// (what the train step does).  Built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/icache_probe.hip -o /tmp/icache_probe && /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define BODY(N) asm volatile(".rept " #N "\n s_add_u32 %0, %0, 1\n .endr" : "+s"(v) : : "scc")

__device__ __forceinline__ void stamp(unsigned long long* t, int slot) {
    if (threadIdx.x == 0) t[blockIdx.x * 2 + slot] = __builtin_amdgcn_s_memrealtime();
}

template <int KB, int ID> __global__ void straight(unsigned long long* t, unsigned* out) {
    stamp(t, 0);
    unsigned v = ID;
    if (KB >= 1) BODY(256);
    if (KB >= 2) BODY(256);
    if (KB >= 4) BODY(512);
    if (KB >= 8) BODY(1024);
    if (KB >= 16) BODY(2048);
    if (KB >= 32) BODY(4096);
    if (KB >= 64) BODY(8192);
    if (v == 0xdeadbeefu) out[0] = v;
    stamp(t, 1);
}

__global__ void looped(unsigned long long* t, unsigned* out, int trips) {
    stamp(t, 0);
    unsigned v = 7;
#pragma unroll 1
    for (int i = 0; i < trips; ++i) BODY(256);
    if (v == 0xdeadbeefu) out[0] = v;
    stamp(t, 1);
}

static unsigned long long* dT;
static unsigned* dOut;
static std::vector<unsigned long long> hT;

struct Res { double span, med; };
template <class F, class G> Res measure(F target, G before, int wgs) {
    std::vector<double> spans, meds;
    for (int rep = 0; rep < 12; ++rep) {
        before();
        target();
        hipDeviceSynchronize();
        hipMemcpy(hT.data(), dT, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        std::vector<double> d;
        for (int w = 0; w < wgs; ++w) {
            lo = std::min(lo, hT[2 * w]); hi = std::max(hi, hT[2 * w + 1]);
            d.push_back((hT[2 * w + 1] - hT[2 * w]) * 0.01);
        }
        std::sort(d.begin(), d.end());
        if (rep >= 2) { spans.push_back((hi - lo) * 0.01); meds.push_back(d[d.size() / 2]); }
    }
    std::sort(spans.begin(), spans.end()); std::sort(meds.begin(), meds.end());
    return {spans[spans.size() / 2], meds[meds.size() / 2]};
}

template <int KB> void run_case(int wgs, int threads) {
    auto self = [&] { hipLaunchKernelGGL((straight<KB, 1>), dim3(wgs), dim3(threads), 0, 0, dT, dOut); };
    auto other = [&] { hipLaunchKernelGGL((straight<64, 2>), dim3(wgs), dim3(threads), 0, 0, dT, dOut); };
    auto loop = [&] { hipLaunchKernelGGL(looped, dim3(wgs), dim3(threads), 0, 0, dT, dOut, KB); };
    Res hot = measure(loop, loop, wgs);
    Res again = measure(self, self, wgs);
    Res cold = measure(self, other, wgs);
    printf("%5d %7d %4d KB | looped %7.2f (%6.2f) | straight, same kernel again %7.2f (%6.2f) | straight, after a different kernel %7.2f (%6.2f)\n",
           wgs, threads, KB, hot.span, hot.med, again.span, again.med, cold.span, cold.med);
}

int main() {
    hipMalloc(&dT, sizeof(unsigned long long) * 2 * 4096);
    hipMalloc(&dOut, 64);
    hT.resize(2 * 4096);
    printf("us: span first start -> last end over all workgroups (median per-workgroup duration)\n");
    printf("%5s %7s %7s\n", "WGs", "threads", "code");
    for (int threads : {64, 256})
        for (int wgs : {256, 512, 2048}) {
            run_case<1>(wgs, threads);
            run_case<4>(wgs, threads);
            run_case<16>(wgs, threads);
            run_case<32>(wgs, threads);
            run_case<64>(wgs, threads);
        }
    return 0;
}
