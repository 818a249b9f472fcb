// Fixed-order sums of per-workgroup partial gradients, several jobs per launch.
//
// Every backward kernel of the library leaves one slab of parameter-gradient partials per workgroup and a small second launch
// adds the slabs in workgroup order (no float atomics: `k_mlp_reduce_partials(_sliced)`, `k_attn_sum_partials`).  The
// representation of BASELINE configs[4] is walked once per gated loss (reference sac_base.py:1607-1631 via `_train_rpm`
// 1798-1839): twelve of those second launches per step, ~5 us each, none of whose results is read before all walks are done.
// `asac_sum_partials_multi` runs them as ONE launch — per job the same slices, the same order of additions and therefore the same
// bits as the launch it stands for (slices = 16: the sliced kernels; slices = 1: the sequential one).
#include "asac_common.h"

namespace asac {

constexpr int kSumSlices = 16;

struct SumJobs {
    asac_partial_sum_t job[ASAC_SUM_PARTIALS_MAX_JOBS];
    int32_t first_block[ASAC_SUM_PARTIALS_MAX_JOBS];
    int32_t n;
};

// 64 elements a workgroup, one wave per slice of slabs: eight loads in flight, added in slab order; the slice sums in order
__global__ __launch_bounds__(64 * kSumSlices) void k_sum_partials_multi(const SumJobs J) {
    __shared__ float part[kSumSlices][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    int k = 0;
#pragma unroll
    for (int q = 1; q < ASAC_SUM_PARTIALS_MAX_JOBS; ++q)
        if (q < J.n && (int)blockIdx.x >= J.first_block[q]) k = q;
    const asac_partial_sum_t& job = J.job[k];
    const int64_t i = (int64_t)((int)blockIdx.x - J.first_block[k]) * 64 + lane;
    const int per = (job.slabs + job.slices - 1) / job.slices;
    const int lo = sl * per, hi = sl < job.slices ? min(lo + per, job.slabs) : lo;
    float s = 0.f;
    if (i < job.n) {
        const float* p = job.partial + i;
        int t = lo;
        for (; t + 8 <= hi; t += 8) {
            float v[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) v[w] = p[(int64_t)(t + w) * job.slab_stride];
#pragma unroll
            for (int w = 0; w < 8; ++w) s += v[w];
        }
        for (; t < hi; ++t) s += p[(int64_t)t * job.slab_stride];
    }
    part[sl][lane] = s;
    __syncthreads();
    if (sl != 0 || i >= job.n) return;
    s = 0.f;
#pragma unroll
    for (int w = 0; w < kSumSlices; ++w) s += part[w][lane];
    job.out[i] = job.accumulate ? job.out[i] + s : s;
}

}  // namespace asac

using namespace asac;

extern "C" {

int asac_sum_partials_multi(int n_jobs, const asac_partial_sum_t* jobs_host, void* stream) {
    if (n_jobs < 1 || n_jobs > ASAC_SUM_PARTIALS_MAX_JOBS || !jobs_host) return bad_arg("asac_sum_partials_multi");
    SumJobs J{};
    int blocks = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const asac_partial_sum_t& j = jobs_host[k];
        if (!j.partial || !j.out || j.slabs < 1 || j.n < 1 || (j.slices != 1 && j.slices != kSumSlices) || j.slab_stride < j.n)
            return bad_arg("asac_sum_partials_multi: job");
        J.job[k] = j;
        J.first_block[k] = blocks;
        blocks += (int)((j.n + 63) / 64);
    }
    J.n = n_jobs;
    // launched once (not under the repeat knob: a job may accumulate)
    hipLaunchKernelGGL(k_sum_partials_multi, dim3((unsigned)blocks), dim3(64 * kSumSlices), 0, as_stream(stream), J);
    return finish_launch("asac_sum_partials_multi");
}

}  // extern "C"
