// What does a kernel's CODE cost when it is fetched cold?  The train step replays nine different kernels of 20-86 KB
// of code each, one after the other: whatever a CU's instruction cache holds from the previous step has been evicted
// by the time the same kernel comes round again.  A single workgroup of straight-line ALU code (8 independent
// accumulators: issue-bound, ~1 instruction per cycle and wave), N instructions long:
//   warm   the same kernel launched back to back (its code stays in the instruction cache)
//   cold   alternating with another kernel of the same size (two of them exceed the cache)
// Reported: time per launch (events around 200 launches), warm and cold, per code size.
//   hipcc --offload-arch=gfx950 -O3 tools/icacheprobe.hip -o tools/debug/icacheprobe && tools/debug/icacheprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N, int SALT>
__global__ __launch_bounds__(256) void straight(float* out, float x) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = x + (float)(j + SALT);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(a[j], 1.0001f + (float)SALT * 1e-6f, a[(j + 1) & 7]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[threadIdx.x] = s;
}

template <int N>
static void run(float* out) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    float ms_warm, ms_cold;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((straight<N, 0>), dim3(1), dim3(256), 0, s, out, 1.f);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((straight<N, 0>), dim3(1), dim3(256), 0, s, out, 1.f);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms_warm, e0, e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps / 2; ++i) {
        hipLaunchKernelGGL((straight<N, 0>), dim3(1), dim3(256), 0, s, out, 1.f);
        hipLaunchKernelGGL((straight<N, 1>), dim3(1), dim3(256), 0, s, out, 1.f);
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms_cold, e0, e1));
    printf("%6d instructions (~%3d KB): warm %6.2f us / launch, alternating %6.2f us / launch -> cold fetch %5.2f us\n", N, N * 8 / 1024,
           ms_warm * 1e3 / reps, ms_cold * 1e3 / reps, (ms_cold - ms_warm) * 1e3 / reps);
}

int main() {
    float* out;
    CK(hipMalloc(&out, 1024));
    run<512>(out);
    run<2048>(out);
    run<4096>(out);
    run<8192>(out);
    run<16384>(out);
    return 0;
}
