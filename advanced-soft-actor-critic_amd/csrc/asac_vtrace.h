// Per-step and per-row pieces of the n-step V-trace return (reference sac_base.py:1244-1295, 1423-1464),
// shared by the stand-alone return kernels and the priority update that forms its own TD errors (returns.hip:
// k_vtrace_return_min(_sc), k_td_update).
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

// masked_prod over values already in registers (the first A of v[0..4))
__device__ __forceinline__ float masked_prod4(const float (&v)[4], int A) {
    float out = 1.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d < A) {
            float x = v[d];
            if (isinf(x)) x = 1.f;
            out *= x;
        }
    }
    if (isinf(out) || isnan(out)) out = 1.f;
    return out;
}

__device__ __forceinline__ VtraceStepRaw vtrace_step_load(const asac_vtrace_args_t& a, int b, int t) {
    VtraceStepRaw r;
    const int n = a.n;
    const float* q0 = a.q + (int64_t)b * a.q_stride_b + (int64_t)t * a.q_stride_t;
    const float* q1 = q0 + a.q_stride_t;
    const float* lp = a.logp + (int64_t)b * (n + 1) + t;
    const int64_t mi = (int64_t)b * a.mask_stride + t;
    if (a.E_sample <= 4 && (!a.use_n_step_is || a.A <= 4)) {
        // Small ensembles and action vectors (the usual ones): every load of the step is issued before the first use —
        // ONE round trip.  The loops below (run-time trip counts) wait for each member / each action dimension in turn:
        // five or six round trips of ~0.6 us on a kernel that does little else.  Members / dimensions beyond the real
        // ones re-read the last real one (a valid address, the value is not used); same operations in the same order.
        const int E = a.E_sample;
        float qa[4], qb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = min(j, E - 1);
            qa[j] = q0[(int64_t)member(a.subset_n, e) * a.q_stride_e];
            qb[j] = q1[(int64_t)member(a.subset_next, e) * a.q_stride_e];
        }
        r.lp0 = lp[0], r.lp1 = lp[1];
        const bool done = a.done[mi];
        r.reward = a.reward[(int64_t)b * a.reward_stride + t];
        r.gamma_ratio = a.gamma_ratio[t];
        const bool gone = a.last_mask[mi] | a.padding_mask[mi];
        float pv[4] = {1.f, 1.f, 1.f, 1.f}, mv[4] = {1.f, 1.f, 1.f, 1.f};
        r.lambda_ratio = 1.f, r.ratio = 1.f;
        if (a.use_n_step_is) {
            r.lambda_ratio = a.lambda_ratio[t];
            const float* pi = a.pi_prob + (int64_t)b * a.pi_stride_b + (int64_t)t * a.pi_stride_t;
            const float* mu = a.mu_prob + (int64_t)b * a.mu_stride_b + (int64_t)t * a.mu_stride_t + a.mu_offset;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = min(j, a.A - 1);
                pv[j] = pi[d], mv[j] = mu[d];
            }
        }
        float m0 = qa[0], m1 = qb[0];
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (j < E) m0 = fminf(m0, qa[j]), m1 = fminf(m1, qb[j]);
        r.m0 = m0, r.m1 = m1;
        r.g = done ? 0.f : a.gamma;                                        // gamma * ~done
        if (a.use_n_step_is) r.ratio = masked_prod4(pv, a.A) / fmaxf(masked_prod4(mv, a.A), 1e-8f);
        r.keep = gone ? 0.f : 1.f;                                         // ~(last | pad)
        return r;
    }
    float m0 = q0[(int64_t)member(a.subset_n, 0) * a.q_stride_e];
    float m1 = q1[(int64_t)member(a.subset_next, 0) * a.q_stride_e];
    for (int e = 1; e < a.E_sample; ++e) {
        m0 = fminf(m0, q0[(int64_t)member(a.subset_n, e) * a.q_stride_e]);
        m1 = fminf(m1, q1[(int64_t)member(a.subset_next, e) * a.q_stride_e]);
    }
    r.m0 = m0, r.m1 = m1;
    r.lp0 = lp[0], r.lp1 = lp[1];
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

// sum_t (prod_{s<t} c_s) d_t of one row by ONE lane, in the association order of the return kernel's scan (returns.hip:
// `seg` lanes per row, each a contiguous segment, segments combined pairwise S_k + P_k * S_k+1): bit-identical to it
__device__ __forceinline__ float vtrace_scan_row(const float* d, const float* c, int n, int seg) {
    float S[4] = {0.f, 0.f, 0.f, 0.f}, P[4] = {1.f, 1.f, 1.f, 1.f};
    const int len = (n + seg - 1) / seg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= seg) break;
        const int t0 = min(n, k * len), t1 = min(n, t0 + len);
        for (int t = t0; t < t1; ++t) {
            S[k] += P[k] * d[t];
            P[k] *= c[t];
        }
    }
    if (seg == 1) return S[0];
    // off = 1: lanes 0 and 2 take their right neighbour; off = 2: lane 0 takes lane 2
    const float S01 = S[0] + P[0] * S[1], P01 = P[0] * P[1];
    const float S23 = S[2] + P[2] * S[3];
    return S01 + P01 * S23;
}
__host__ __device__ __forceinline__ int vtrace_scan_lanes(int64_t B, int n) {
    return ((int64_t)B * n >= (1 << 21) && n <= 8) ? 1 : 4;
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
