// atomic_probe.hip -- what do device-scope f64 atomics cost as a replacement for the per-workgroup BatchNorm partials + finalize
// launch?  G workgroups each add 2*C doubles (sum, sum of squares per channel) to ONE [2][C] accumulator, against G workgroups
// storing 2*C floats to their own slice of a [G][2][C] array (what the kernels do today).  Built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_store(float* partial, int C, int rounds) {
    for (int r = 0; r < rounds; ++r)
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) partial[((size_t)blockIdx.x * rounds + r) * 2 * C + c] = (float)(c + r);
}
__global__ void k_atomic(double* acc, int C, int rounds, int slots) {
    double* a = acc + (size_t)(blockIdx.x % slots) * 2 * C;
    for (int r = 0; r < rounds; ++r)
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) unsafeAtomicAdd(&a[c], (double)(c + r));
}
__global__ void k_atomic_f32(float* acc, int C, int rounds, int slots) {
    float* a = acc + (size_t)(blockIdx.x % slots) * 2 * C;
    for (int r = 0; r < rounds; ++r)
        for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) unsafeAtomicAdd(&a[c], (float)(c + r));
}

template <class F> float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int Gs[] = {128, 512, 2048}, Cs[] = {16, 64, 256};
    float* partial; double* acc; float* accf;
    hipMalloc(&partial, (size_t)2048 * 64 * 2 * 256 * 4);
    hipMalloc(&acc, 64 * 2 * 256 * 8); hipMalloc(&accf, 64 * 2 * 256 * 4);
    hipMemset(acc, 0, 64 * 2 * 256 * 8); hipMemset(accf, 0, 64 * 2 * 256 * 4);
    printf("us per launch (200 back-to-back launches; the launch floor is in every column)\n");
    printf("%6s %5s %7s | %9s %12s %12s %12s %12s\n", "G", "C", "rounds", "store", "atomic f64", "f64 8 slots", "atomic f32", "f32 8 slots");
    for (int G : Gs) for (int C : Cs) for (int rounds : {1, 16, 64}) {
        float a = time_us([&] { hipLaunchKernelGGL(k_store, dim3(G), dim3(256), 0, 0, partial, C, rounds); }, 200);
        float b = time_us([&] { hipLaunchKernelGGL(k_atomic, dim3(G), dim3(256), 0, 0, acc, C, rounds, 1); }, 200);
        float c = time_us([&] { hipLaunchKernelGGL(k_atomic, dim3(G), dim3(256), 0, 0, acc, C, rounds, 8); }, 200);
        float d = time_us([&] { hipLaunchKernelGGL(k_atomic_f32, dim3(G), dim3(256), 0, 0, accf, C, rounds, 1); }, 200);
        float e = time_us([&] { hipLaunchKernelGGL(k_atomic_f32, dim3(G), dim3(256), 0, 0, accf, C, rounds, 8); }, 200);
        printf("%6d %5d %7d | %9.2f %12.2f %12.2f %12.2f %12.2f   (%.0f atomics per launch)\n", G, C, rounds, a, b, c, d, e, (double)G * 2 * C * rounds);
    }
    return 0;
}
