// Random draws + the independent work every captured train step begins with (shared by noise.hip and the fused
// prologue + sample launch of sumtree.hip).
#pragma once
#include "asac_common.h"

#include <cmath>

namespace asac {

struct Philox {
    uint32_t c[4];
};

__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox{{c0, c1, c2, c3}};
}

struct PrologueArgs {
    uint64_t seed;
    const int64_t* step;
    double* u;
    int64_t n_u;
    float* normal;
    int64_t n_normal;
    int32_t* subsets;
    int32_t n_subsets, E_sample, E;
    int32_t polyak_blocks;
    float* target;
    const float* source;
    int64_t n_polyak;
    float one_m_tau, tau;
    int32_t zero_blocks;
    float* zero_out;
    int64_t n_zero;
};

// the f64 uniform number `t` of the step's block of draws: what lane (normal lanes + t / 2) of the fill produces
__device__ __forceinline__ double prologue_uniform(uint64_t seed, uint64_t s, int64_t n_normal, int64_t t) {
    const int64_t i = (n_normal + 3) / 4 + (t >> 1);
    const Philox x = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)s, (uint32_t)(s >> 32),
                                   (uint32_t)seed, (uint32_t)(seed >> 32));
    // (selects, not x.c[2 * h]: a run-time index sends the four words through scratch memory — a store / load round trip
    // in front of the sampler's descent, the first thing the step's first launch does)
    const bool h = (t & 1) != 0;
    const uint64_t bits = ((uint64_t)(h ? x.c[2] : x.c[0]) << 32) | (h ? x.c[3] : x.c[1]);
    return (double)(bits >> 11) * 1.1102230246251565e-16;   // 2^-53: [0, 1)
}

// lanes [0, ceil(n_normal/4)): four N(0,1) each (Box-Muller);  the next ceil(n_u/2) lanes: two U[0,1) f64 each;
// the next n_subsets lanes: one random ensemble subset each (the first E_sample entries of a uniformly
// random permutation of range(E): partial Fisher-Yates, reference `torch.randperm(E)[:E_sample]`,
// sac_base.py:1434)
// The leading `polyak_blocks` workgroups (if any) apply the step's Polyak update instead, the next `zero_blocks`
// clear the step's gradient buffer: the independent launches every captured step begins with, as one.
// `block`: index among the launch's prologue workgroups (256 threads each); `skip_uniforms`: the uniforms are drawn by
// their consumer (the fused sampler), not stored by these lanes
__device__ __forceinline__ void prologue_block(const PrologueArgs& a, int block, bool skip_uniforms) {
    if (block < a.polyak_blocks) {
        polyak_span(a.target, a.source, a.n_polyak, a.one_m_tau, a.tau, (int64_t)block * 256 + threadIdx.x,
                    (int64_t)a.polyak_blocks * 256);
        return;
    }
    if (block < a.polyak_blocks + a.zero_blocks) {
        const int64_t first = (int64_t)(block - a.polyak_blocks) * 256 + threadIdx.x;
        const int64_t stride = (int64_t)a.zero_blocks * 256;
        if ((reinterpret_cast<uintptr_t>(a.zero_out) & 15) == 0) {
            float4* z4 = reinterpret_cast<float4*>(a.zero_out);
            for (int64_t k = first; k < a.n_zero / 4; k += stride) z4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t k = (a.n_zero & ~(int64_t)3) + first; k < a.n_zero; k += stride) a.zero_out[k] = 0.f;
        } else {
            for (int64_t k = first; k < a.n_zero; k += stride) a.zero_out[k] = 0.f;
        }
        return;
    }
    const int64_t i = (int64_t)(block - a.polyak_blocks - a.zero_blocks) * 256 + threadIdx.x;
    const int64_t normal_lanes = (a.n_normal + 3) / 4, u_lanes = (a.n_u + 1) / 2;
    if (i >= normal_lanes + u_lanes + a.n_subsets) return;
    const uint64_t s = (uint64_t)*a.step;
    const uint64_t seed = a.seed;
    if (i >= normal_lanes + u_lanes) {
        const int k = (int)(i - normal_lanes - u_lanes);
        int perm[ASAC_MAX_ENSEMBLE];
        for (int e = 0; e < a.E; ++e) perm[e] = e;
        Philox x{};
        for (int e = 0; e < a.E_sample; ++e) {
            if ((e & 3) == 0)      // a distinct counter block per subset lane: bit 63 of the lane index set
                x = philox4x32_10((uint32_t)k, 0x80000000u | (uint32_t)(e >> 2), (uint32_t)s, (uint32_t)(s >> 32),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
            // (a select chain, not `x.c[e & 3]`: the dynamic index put the four words into scratch memory — 20 bytes per lane
            // of scratch for every workgroup of the launches this is inlined into)
            const int sel = e & 3;
            const uint32_t word = sel == 0 ? x.c[0] : sel == 1 ? x.c[1] : sel == 2 ? x.c[2] : x.c[3];
            const int j = e + (int)(word % (uint32_t)(a.E - e));
            const int tmp = perm[e];
            perm[e] = perm[j];
            perm[j] = tmp;
            a.subsets[k * a.E_sample + e] = perm[e];
        }
        return;
    }
    if (i >= normal_lanes && skip_uniforms) return;
    const Philox x = philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)s, (uint32_t)(s >> 32),
                                   (uint32_t)seed, (uint32_t)(seed >> 32));
    if (i < normal_lanes) {
        float out[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)x.c[2 * h] + 1.f) * 2.3283064365386963e-10f;     // (0, 1]
            const float u2 = (float)x.c[2 * h + 1] * 2.3283064365386963e-10f;         // [0, 1]
            const float r = sqrtf(-2.f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            out[2 * h] = r * cs;
            out[2 * h + 1] = r * sn;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * i + k < a.n_normal) a.normal[4 * i + k] = out[k];
    } else {
        const int64_t j = i - normal_lanes;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint64_t bits = ((uint64_t)x.c[2 * h] << 32) | x.c[2 * h + 1];
            if (2 * j + h < a.n_u) a.u[2 * j + h] = (double)(bits >> 11) * 1.1102230246251565e-16;   // 2^-53: [0, 1)
        }
    }
}

inline int64_t prologue_span_blocks(int64_t n) {
    int64_t nb = (n / 4 + 255) / 256;
    return n == 0 ? (int64_t)0 : (nb < 1 ? (int64_t)1 : (nb > 2048 ? (int64_t)2048 : nb));
}

}  // namespace asac
