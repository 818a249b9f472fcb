// Per-step and per-row pieces of the n-step V-trace return (reference sac_base.py:1244-1295, 1423-1464),
// shared by the stand-alone return kernel (returns.hip), the Q backward that forms its own target
// (mlp.hip) and the priority update that forms its own TD error (sumtree.hip).
#pragma once
#include "asac_common.h"

#include <cmath>

namespace asac {

// ensemble member e of a subset; the subset lives in DEVICE memory because it changes every step
// while the launch itself may be frozen inside a hipGraph (NULL = members 0..E_sample-1)
__device__ __forceinline__ int member(const int32_t* subset, int e) { return subset ? subset[e] : e; }

// operators.py:27-31 on a length-A vector read with stride 1
__device__ __forceinline__ float masked_prod(const float* p, int A) {
    float out = 1.f;
    for (int d = 0; d < A; ++d) {
        float v = p[d];
        if (isinf(v)) v = 1.f;
        out *= v;
    }
    if (isinf(out) || isnan(out)) out = 1.f;
    return out;
}

// Everything of step t of row b that does not depend on the running product:
//   V(s_t)   = min_{e in subset_n}    Q_e(s_t, a_t)     - alpha logpi_t      (returned)
//   V(s_t+1) = min_{e in subset_next} Q_e(s_t+1, a_t+1) - alpha logpi_t+1
//   *d = rho_t * lambda^t * gamma^t * (r_t + gamma (1 - done_t) V(s_t+1) - V(s_t)) * ~(last | pad)
//   *c = min(pi/mu, c_bar)   (1 without importance sampling)
// in two halves: everything that is read from memory (independent of the temperature), then the arithmetic — a
// launch that first has to work the temperature out (alpha_adam_preview) has its loads in flight meanwhile.
struct VtraceStepRaw {
    float m0, m1, lp0, lp1, g, reward, gamma_ratio, lambda_ratio, ratio, keep;
};

__device__ __forceinline__ VtraceStepRaw vtrace_step_load(const asac_vtrace_args_t& a, int b, int t) {
    VtraceStepRaw r;
    const int n = a.n;
    const float* q0 = a.q + (int64_t)b * a.q_stride_b + (int64_t)t * a.q_stride_t;
    const float* q1 = q0 + a.q_stride_t;
    float m0 = q0[(int64_t)member(a.subset_n, 0) * a.q_stride_e];
    float m1 = q1[(int64_t)member(a.subset_next, 0) * a.q_stride_e];
    for (int e = 1; e < a.E_sample; ++e) {
        m0 = fminf(m0, q0[(int64_t)member(a.subset_n, e) * a.q_stride_e]);
        m1 = fminf(m1, q1[(int64_t)member(a.subset_next, e) * a.q_stride_e]);
    }
    r.m0 = m0, r.m1 = m1;
    const float* lp = a.logp + (int64_t)b * (n + 1) + t;
    r.lp0 = lp[0], r.lp1 = lp[1];
    const int64_t mi = (int64_t)b * a.mask_stride + t;
    r.g = a.done[mi] ? 0.f : a.gamma;                                  // gamma * ~done
    r.reward = a.reward[(int64_t)b * a.reward_stride + t];
    r.gamma_ratio = a.gamma_ratio[t];
    r.lambda_ratio = 1.f, r.ratio = 1.f;
    if (a.use_n_step_is) {
        r.lambda_ratio = a.lambda_ratio[t];
        const float pi = masked_prod(a.pi_prob + (int64_t)b * a.pi_stride_b + (int64_t)t * a.pi_stride_t, a.A);
        const float mu = masked_prod(a.mu_prob + (int64_t)b * a.mu_stride_b + (int64_t)t * a.mu_stride_t + a.mu_offset, a.A);
        r.ratio = pi / fmaxf(mu, 1e-8f);
    }
    r.keep = (a.last_mask[mi] | a.padding_mask[mi]) ? 0.f : 1.f;       // ~(last | pad)
    return r;
}

__device__ __forceinline__ float vtrace_step_finish(const asac_vtrace_args_t& a, const VtraceStepRaw& r, float alpha,
                                                    float* d, float* c) {
    const float v_t = r.m0 - alpha * r.lp0, v_next = r.m1 - alpha * r.lp1;
    float td = r.reward + r.g * v_next - v_t;
    td = r.gamma_ratio * td;
    float cc = 1.f;
    if (a.use_n_step_is) {
        td = r.lambda_ratio * td;
        td = fminf(r.ratio, a.v_rho) * td;
        cc = fminf(r.ratio, a.v_c);
    }
    *d = td * r.keep;
    *c = cc;
    return v_t;
}

__device__ __forceinline__ float vtrace_step_terms(const asac_vtrace_args_t& a, int b, int t, float alpha,
                                                   float* d, float* c) {
    const VtraceStepRaw r = vtrace_step_load(a, b, t);
    return vtrace_step_finish(a, r, alpha, d, c);
}

// y of row b by ONE lane (short windows: a handful of independent loads per step, then n fused
// multiply-adds): y = V(s_0) + sum_t (prod_{s<t} c_s) d_t
__device__ __forceinline__ float vtrace_row_y(const asac_vtrace_args_t& a, int b) {
    const float alpha = expf(*a.log_alpha);
    float S = 0.f, P = 1.f, v0 = 0.f;
    for (int t = 0; t < a.n; ++t) {
        float d, c;
        const float v_t = vtrace_step_terms(a, b, t, alpha, &d, &c);
        if (t == 0) v0 = v_t;
        S += P * d;
        P *= c;
    }
    return v0 + S;
}

}  // namespace asac
