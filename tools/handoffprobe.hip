// What does a producer -> consumers hand-off INSIDE one launch cost on this box (8 XCDs, one L2 each)?  Decides
// whether a launch whose workgroups need another workgroup's output (the window gather needs the sampler's ids) can
// ride in the producer's launch, spinning on a flag, instead of paying a dependent kernel boundary.
//   workgroup 0   spends `work` us (a stand-in for the sampler), writes `bytes` of payload, releases a flag
//   workgroups 1..W  spin on the flag (bounded), then read the payload and check it
// Variants of the producer's release / the consumers' acquire:
//   fence     __threadfence() + relaxed flag store   |  relaxed spin + __threadfence() before reading
//   scoped    flag store with release / load with acquire at agent scope (what the compiler makes of the C++ model)
//   bypass    payload written / read with agent-scope relaxed atomics (sc1: past the non-coherent caches), relaxed flag
// Reported: kernel time - producer work = what the hand-off adds; compare with a dependent launch (1.8 us + the
// consumer kernel's own entry, tools/launchprobe.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/handoffprobe.hip -o tools/debug/handoffprobe && tools/debug/handoffprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kSpinLimit = 1 << 22;

template <int MODE>
__global__ __launch_bounds__(256) void handoff(int* flag, int* arrivals, int* payload, int words, int epoch,
                                               unsigned long long work_ticks, int* bad, unsigned long long* stamps) {
    if (blockIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < work_ticks) {}
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            if (MODE == 2) __hip_atomic_store(&payload[i], epoch * 7 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else payload[i] = epoch * 7 + i;
        }
        if (MODE == 0) __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 1) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stamps[0] = wall_clock64();
        }
        return;
    }
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int spins = 0, seen = 0;
        while (spins++ < kSpinLimit) {
            seen = MODE == 1 ? __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                             : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen == epoch) break;
            __builtin_amdgcn_s_sleep(2);
        }
        ok = seen == epoch;
        if (MODE == 0) __threadfence();
    }
    __syncthreads();
    if (!ok) {
        if (threadIdx.x == 0) atomicAdd(bad, 1000000);
        return;
    }
    int wrong = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const int v = MODE == 2 ? __hip_atomic_load(&payload[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : payload[i];
        wrong += v != epoch * 7 + i;
    }
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) stamps[1] = wall_clock64();
}

template <int MODE>
static void run(const char* name, int consumers, int words, double work_us) {
    int *flag, *arrivals, *payload, *bad;
    unsigned long long* stamps;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&arrivals, 4)); CK(hipMalloc(&payload, words * 4)); CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&stamps, 16));
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(payload, 0, words * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned long long ticks = (unsigned long long)(work_us * 100.0);
    const int reps = 200;
    double tot = 0, after = 0;
    for (int r = 1; r <= reps + 5; ++r) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(handoff<MODE>, dim3(1 + consumers), dim3(256), 0, s, flag, arrivals, payload, words, r, ticks, bad, stamps);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[2];
        CK(hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost));
        if (r > 5) { tot += ms * 1e3; after += (double)(st[1] - st[0]) / 100.0; }
    }
    int h_bad = 0;
    CK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost));
    printf("%-7s consumers %3d payload %6d B work %.1f us: kernel %.2f us (event pair), last consumer done %.2f us after the "
           "flag, wrong words %d\n", name, consumers, words * 4, work_us, tot / reps, after / reps, h_bad);
    CK(hipFree(flag)); CK(hipFree(arrivals)); CK(hipFree(payload)); CK(hipFree(bad)); CK(hipFree(stamps));
}

int main() {
    for (int consumers : {16, 64}) {
        for (int words : {512, 16384}) {
            run<0>("fence", consumers, words, 6.0);
            run<1>("scoped", consumers, words, 6.0);
            run<2>("bypass", consumers, words, 6.0);
        }
    }
    return 0;
}
