// Tanh-squashed Gaussian sampling and stored-action probabilities (reference algorithm/utils/operators.py:12-31,
// sac_base.py:1346-1351, 1430, 1183-1187, 1452), shared by the elementwise launches (returns.hip) and the fused
// policy -> sample -> critics forward (mlp.hip).  Evaluation order of the reference's eager ops; -ffp-contract=off.
#pragma once
#include "asac_common.h"

#include <cmath>

namespace asac {

constexpr float kSquashFloor = 1e-2f;
constexpr float kLogSqrt2Pi = 0.91893853320467274178f;   // math.log(math.sqrt(2*math.pi))

// torch.distributions.Normal.log_prob:  -((x-loc)^2)/(2*scale^2) - log(scale) - log(sqrt(2pi))
__device__ __forceinline__ float normal_log_prob(float x, float loc, float scale) {
    const float d = x - loc;
    const float var = scale * scale;
    return -(d * d) / (2.f * var) - logf(scale) - kLogSqrt2Pi;
}

__device__ __forceinline__ float squash_jac(float x) {
    const float t = tanhf(x);
    return fmaxf(1.f - t * t, kSquashFloor);
}

// ------------------------------------------------------------------------------------------------
// rsample + tanh + squash-corrected log-prob, optionally fused with the probability of the STORED
// actions under the same Gaussian.  One lane per row (A is small: 1..64).  loc / scale rows are
// `ls` floats apart, so they may be the two halves of the fused policy network's [rows, 2A] output.
// ------------------------------------------------------------------------------------------------
struct StoredProb {
    const float* action;     // [samples, T, >= a_off + A] view; NULL = not requested
    int32_t T;
    int64_t a_sb, a_st;
    int32_t a_off;
    float* out;              // same addressing
    int64_t p_sb, p_st;
    int32_t p_off;
};

// prob_d = exp(N(x_d).log_prob) / prod_e max(1 - tanh(x_e)^2, 1e-2), x = atanh(clamp(a, +-0.999))
__device__ __forceinline__ void stored_action_prob(const float* __restrict__ loc, const float* __restrict__ scale,
                                                   const StoredProb& sp, int64_t r, int A) {
    const int64_t sb = r / sp.T;
    const int64_t st = r - sb * sp.T;
    const float* a = sp.action + sb * sp.a_sb + st * sp.a_st + sp.a_off;
    float jac = 1.f;
    for (int d = 0; d < A; ++d) {
        const float x = atanhf(fminf(fmaxf(a[d], -0.999f), 0.999f));
        jac *= squash_jac(x);
    }
    float* out = sp.out + sb * sp.p_sb + st * sp.p_st + sp.p_off;
    for (int d = 0; d < A; ++d) {
        const float x = atanhf(fminf(fmaxf(a[d], -0.999f), 0.999f));
        out[d] = expf(normal_log_prob(x, loc[d], scale[d])) / jac;
    }
}

// one row: loc / scale / eps / a_out (/ x_out) point at the row's A values, logp_out at its scalar
__device__ __forceinline__ void squash_sample_at(const float* __restrict__ lrow, const float* __restrict__ srow,
                                                 const float* __restrict__ erow, int A, float* __restrict__ a_row,
                                                 float* __restrict__ logp_out, float* __restrict__ x_row) {
    float corr = 0.f;    // sum_e log(max(1 - tanh(x_e)^2, 1e-2))
    for (int d = 0; d < A; ++d) {
        const float x = lrow[d] + erow[d] * srow[d];
        const float t = tanhf(x);
        corr += logf(fmaxf(1.f - t * t, kSquashFloor));
        a_row[d] = t;
        if (x_row) x_row[d] = x;
    }
    float lp = 0.f;
    for (int d = 0; d < A; ++d) {
        const float l = lrow[d], s = srow[d];
        const float x = l + erow[d] * s;
        float v = normal_log_prob(x, l, s) - corr;      // correction broadcast to every component
        if (v == INFINITY) v = 0.f;                     // sum_log_prob's inf mask
        lp += v;
    }
    *logp_out = lp;
}

__device__ __forceinline__ void squash_sample_row(const float* __restrict__ loc, const float* __restrict__ scale,
                                                  int64_t ls, const float* __restrict__ eps, int64_t r, int A,
                                                  float* __restrict__ a_out, float* __restrict__ logp_out,
                                                  float* __restrict__ x_out, const StoredProb& sp) {
    const int64_t base = r * A;
    const float* lrow = loc + r * ls;
    const float* srow = scale + r * ls;
    squash_sample_at(lrow, srow, eps + base, A, a_out + base, logp_out + r, x_out ? x_out + base : nullptr);
    if (sp.action) stored_action_prob(lrow, srow, sp, r, A);
}

}  // namespace asac
