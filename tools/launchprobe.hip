// What does a dependent kernel boundary cost inside a replayed hipGraph on this box, and does it depend on the
// clock / power state?  A chain of K trivial dependent kernels (one workgroup each, one load + one store) is captured
// and replayed; per-kernel cost = replay time / K (HIP events on the launch stream).  Variants:
//   idle        nothing else on the device
//   spin W      a persistent kernel of W workgroups spinning on FMAs on a second stream (keeps SCLK up?)
//   back2back   graph launches issued without host gaps
// In-kernel effective shader clock: clock64() (s_memtime, shader cycles) against wall_clock64() (100 MHz) around a
// dependent FMA chain, sampled by the LAST kernel of the chain.
//   hipcc --offload-arch=gfx950 -O3 tools/launchprobe.hip -o tools/launchprobe && tools/launchprobe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void tiny(float* p) { p[threadIdx.x] += 1.0f; }

__global__ void tiny_wide(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.0f;
}

__global__ void clk(unsigned long long* out) {
    float x = 1.0f + threadIdx.x, a = 1.0001f, b = 0.5f;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
#pragma unroll
    for (int i = 0; i < 2048; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (unsigned long long)x; }
}

// bounded: spins on FMAs for `ticks` of the 100 MHz wall clock, then ends by itself
__global__ void spin(unsigned long long ticks, float* sink) {
    float x = 1.0f + threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 256; ++i) x = x * 1.0001f + 0.5f;
        if (x > 1e30f) x = 1.0f;
    }
    if (x == 12345.f) sink[0] = x;
}

static double replay_us(hipGraphExec_t exec, hipStream_t s, int reps, bool sync_each) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) {
        CK(hipGraphLaunch(exec, s));
        if (sync_each) CK(hipStreamSynchronize(s));
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}

int main(int argc, char** argv) {
    const int K = 64;
    float* buf; CK(hipMalloc(&buf, 1 << 22)); CK(hipMemset(buf, 0, 1 << 22));
    unsigned long long* cbuf; CK(hipMalloc(&cbuf, 64));
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* sink; CK(hipMalloc(&sink, 4));
    hipStream_t s, side;
    CK(hipStreamCreate(&s)); CK(hipStreamCreate(&side));

    auto capture = [&](int wgs) {
        hipGraph_t g; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) {
            if (wgs == 1) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, buf);
            else hipLaunchKernelGGL(tiny_wide, dim3(wgs), dim3(256), 0, s, buf, wgs * 256);
        }
        hipLaunchKernelGGL(clk, dim3(1), dim3(64), 0, s, cbuf);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        return exec;
    };
    auto report = [&](const char* tag, hipGraphExec_t exec, bool sync_each) {
        double us = replay_us(exec, s, 200, sync_each);
        unsigned long long h[3];
        CK(hipMemcpy(h, cbuf, 24, hipMemcpyDeviceToHost));
        double mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0;   // shader cycles per 10 ns tick
        printf("%-34s %7.2f us per replay  %5.2f us per kernel   fma chain %5.2f us, shader clock %4.0f MHz\n", tag, us,
               us / (K + 1), h[1] / 100.0, mhz);
    };

    for (int wgs : {1, 256}) {
        hipGraphExec_t exec = capture(wgs);
        char tag[96];
        snprintf(tag, sizeof tag, "idle, %d-WG kernels, back-to-back", wgs);
        report(tag, exec, false);
        snprintf(tag, sizeof tag, "idle, %d-WG kernels, sync each", wgs);
        report(tag, exec, true);
        for (int W : {1, 8, 32, 128}) {
            hipLaunchKernelGGL(spin, dim3(W), dim3(256), 0, side, 100ull * 400000ull /* 0.4 s */, sink);
            snprintf(tag, sizeof tag, "spin %d WGs, %d-WG kernels, b2b", W, wgs);
            report(tag, exec, false);
            snprintf(tag, sizeof tag, "spin %d WGs, %d-WG kernels, sync", W, wgs);
            report(tag, exec, true);
            auto t0 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(side));
            printf("      (spinner still ran %.0f ms after the measurement)\n",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    }
    // eager stream launches (no graph), same chain
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < 20 * K; ++k) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, buf);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("eager stream, 1-WG kernels: %.2f us per kernel\n", 1e3 * ms / (20 * K));
        }
    }
    return 0;
}
