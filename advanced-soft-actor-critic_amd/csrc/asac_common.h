// Shared helpers for the libasac_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asac_hip.h"

namespace asac {

constexpr int kWave = 64;

void set_error(hipError_t e, const char* where);

// Measurement knob (asac_set_launch_repeat): every kernel launch of an entry point is issued this
// many times back-to-back, so HIP events around ONE call resolve per-launch device time even though
// a single launch (3-10 us) is shorter than the host cost of issuing it.  1 in normal operation.
extern int g_launch_repeat;
#define ASAC_LAUNCH(...)                                                      \
    for (int asac_rep_ = 0; asac_rep_ < ::asac::g_launch_repeat; ++asac_rep_) \
    hipLaunchKernelGGL(__VA_ARGS__)

inline int finish_launch(const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) set_error(e, where);
    return (int)e;
}

inline int bad_arg(const char* where) {
    set_error(hipErrorInvalidValue, where);
    return (int)hipErrorInvalidValue;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// out (+)= the sum over `blocks` partials [m * n | m] in a fixed order (csrc/xty.hip)
void xty_reduce_launch(const float* part, int blocks, int mn, int m, float* out, float* colsum, int accumulate, hipStream_t stream);

inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// ring slot of a (possibly negative) id, NumPy `%` semantics
__device__ __forceinline__ int ring_slot(int64_t id, int capacity) {
    int64_t m = id % capacity;
    return (int)(m < 0 ? m + capacity : m);
}

// K5 Polyak, reference algorithm/sac_base.py:761-764:  t.copy_(t * (1 - tau) + p * tau)
__device__ __forceinline__ float polyak1(float t, float p, float one_m_tau, float tau) {
    return t * one_m_tau + p * tau;   // two roundings + add (-ffp-contract=off)
}

// lane `tid` of `stride` lanes updates its share of target[0..n) (16-byte vectors when aligned)
__device__ __forceinline__ void polyak_span(float* __restrict__ target, const float* __restrict__ source, int64_t n,
                                            float one_m_tau, float tau, int64_t tid, int64_t stride) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(source)) & 15) == 0;
    int64_t done = 0;
    if (aligned) {
        const int64_t n4 = n / 4;
        float4* t4 = reinterpret_cast<float4*>(target);
        const float4* s4 = reinterpret_cast<const float4*>(source);
        for (int64_t i = tid; i < n4; i += stride) {
            float4 t = t4[i];
            const float4 s = s4[i];
            t.x = polyak1(t.x, s.x, one_m_tau, tau);
            t.y = polyak1(t.y, s.y, one_m_tau, tau);
            t.z = polyak1(t.z, s.z, one_m_tau, tau);
            t.w = polyak1(t.w, s.w, one_m_tau, tau);
            t4[i] = t;
        }
        done = n4 * 4;
    }
    for (int64_t i = done + tid; i < n; i += stride) target[i] = polyak1(target[i], source[i], one_m_tau, tau);
}

// Clipped double-Q loss of one (member, row) pair (reference sac_base.py:1539-1561):
//   l = max((t + clamp(q - t, +-eps) - y)^2, (q - y)^2) * w;  returns l, *grad = d l / d q.
// d max(la, lb)/dq: the lb branch always depends on q, the la branch only while the clamp is inactive;
// torch.maximum splits ties half/half.
__device__ __forceinline__ float clipped_q_loss_row(float qv, float tv, float yv, float wv, float clip_eps,
                                                    float* grad) {
    const float diff = qv - tv;
    const float clipped = tv + fminf(fmaxf(diff, -clip_eps), clip_eps);
    const float da = clipped - yv, db = qv - yv;
    const float la = da * da, lb = db * db;
    const bool clamp_open = (diff >= -clip_eps) && (diff <= clip_eps);
    float g;
    if (lb > la) g = 2.f * db;
    else if (lb < la) g = clamp_open ? 2.f * da : 0.f;
    else g = 0.5f * (2.f * db) + (clamp_open ? 0.5f * (2.f * da) : 0.f);
    *grad = wv * g;
    return fmaxf(la, lb) * wv;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Kernel arguments read WHERE THEY ARE USED.  A by-value struct parameter is loaded whole in the kernel's entry block:
// with argument blocks of 0.5 - 2 KB the kernel starts with dozens of scalar loads in several dependent wait groups and
// parks what does not fit the 102 SGPRs in VGPR lanes (`v_writelane`), all before its first vector load is issued.  A
// reference into the kernarg segment (address space 4) makes every field an ordinary scalar load at its use: the staging
// phase fetches what it needs in one batch, everything else arrives while the kernel is busy.  Usage:
//     __global__ void k(const Args by_value) { const ASAC_KARG Args& a = *static_cast<const ASAC_KARG Args*>(kernarg_base()); ...
// (`by_value` must be the FIRST parameter: offset 0 of the segment; it is never named again.)
#define ASAC_KARG __attribute__((address_space(4)))
__device__ __forceinline__ const ASAC_KARG void* kernarg_base() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (const ASAC_KARG void*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;
#endif
}

// IS weight of one sampled row (reference replay_buffer.py:352-354 under NumPy 2 promotion rules)
__device__ __forceinline__ float is_weight(float p, float total, float min_ratio, double beta) {
    const float ratio = p / total;                   // float32, like NumPy
    const float rel = ratio / min_ratio;
    return (float)pow((double)rel, -beta);           // float64 power, then astype(float32)
}

// a by-value copy of (a part of) the kernel arguments, fetched where the copy is made
template <typename T>
__device__ __forceinline__ T karg_copy(const ASAC_KARG T* p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}

}  // namespace asac
