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

// ------------------------------------------------------------------------------------------------
// The same arithmetic as an EPILOGUE of the policy's forward launch (mlp.hip, `asac_mlp_forward_multi_sampled`): the lanes
// that have just formed a row's (loc | scale) head values in MFMA layout — lane (g, col) of a wave holds column `col` of the
// row of lane group g, locations in columns [0, A), scales in [A, 2A), 2A <= 16 — sample from it, score the stored
// action and draw the optional second sample of window position t2 without the values leaving the wave: per-element
// terms on the A lanes of a row, the sums / products over the action dimension by shuffles in d order — the same values
// in the same order as squash_rows_block (returns.hip), bit for bit.  Called by ALL 64 lanes of a wave (shuffles).
// ------------------------------------------------------------------------------------------------
struct SampleEpi {
    const float *eps, *eps2;           // main sample over every row [N][A] (or NULL); second sample [samples][A] (or NULL)
    float *a_out, *logp_out, *a2_out, *logp2_out;
    const float* action;               // stored actions (or NULL): element (sb, st, d) at action[sb * a_sb + st * a_st + a_off + d]
    float* prob_out;                   // ... their probabilities, same addressing with p_*
    int32_t a_sb, a_st, a_off, p_sb, p_st, p_off;
    int32_t A, T, t2, on;              // T: rows per sample (window length) for the stored actions / the second sample
};

__device__ __forceinline__ float epi_row_logp(float s0, float s1, int base, int A) {
    float corr = 0.f;
    for (int dd = 0; dd < A; ++dd) corr += __shfl(s0, base + dd);
    float lp = 0.f;
    for (int dd = 0; dd < A; ++dd) {
        float v = __shfl(s1, base + dd) - corr;      // correction broadcast to every component
        if (v == INFINITY) v = 0.f;                  // sum_log_prob's inf mask
        lp += v;
    }
    return lp;
}

// what a lane reads from memory for its (row, col): requested at the START of the tile (`sample_epilogue_fetch`) so that it
// travels under the tile's layers — requested in the head phase each of the (up to three) loads was a cold round trip of
// 1-2 us on the tile's critical path (20 736 rows: 26.9 us instead of 19.8 + 5.0 for the two launches)
struct SampleEpiIn {
    float ev, av, ev2;
    int32_t row, sb, st;
    bool live, at;
};

__device__ __forceinline__ SampleEpiIn sample_epilogue_fetch(const SampleEpi& s, int64_t row64, int64_t N, int col) {
    SampleEpiIn in;
    const int A = s.A;
    in.live = row64 < N && col < A;
    in.row = (int)(row64 < N ? row64 : 0);
    in.sb = in.st = 0;
    if (s.action || s.eps2) {
        in.sb = (int)((unsigned)in.row / (unsigned)s.T);
        in.st = in.row - in.sb * s.T;
    }
    in.at = in.live && s.eps2 && in.st == s.t2;
    const int c = col < A ? col : 0;               // (every lane loads from a valid address; dead lanes' values are unused)
    in.ev = s.eps ? s.eps[in.row * A + c] : 0.f;
    in.av = s.action ? s.action[in.sb * s.a_sb + in.st * s.a_st + s.a_off + c] : 0.f;
    in.ev2 = s.eps2 ? s.eps2[in.sb * A + c] : 0.f;
    return in;
}

__device__ __forceinline__ void sample_epilogue(const SampleEpi& s, const SampleEpiIn& in, int col, float hv, int lane) {
    const int A = s.A, base = lane & 48;
    const bool live = in.live;
    const int row = in.row, sb = in.sb, st = in.st;
    const float l = hv;                                             // this lane's location (col < A)
    const float sc = __shfl(hv, base + ((A + col) & 15));           // ... and its scale, from the lane A columns on
    if (s.eps) {
        const float x = l + in.ev * sc;
        const float t = tanhf(x);
        const float s0 = logf(fmaxf(1.f - t * t, kSquashFloor));
        const float s1 = normal_log_prob(x, l, sc);
        if (live) s.a_out[row * A + col] = t;
        const float lp = epi_row_logp(s0, s1, base, A);
        if (live && col == 0) s.logp_out[row] = lp;
    }
    if (s.action) {
        const float x = atanhf(fminf(fmaxf(in.av, -0.999f), 0.999f));
        const float j = squash_jac(x);
        const float pr = expf(normal_log_prob(x, l, sc));
        float jac = 1.f;
        for (int dd = 0; dd < A; ++dd) jac *= __shfl(j, base + dd);
        if (live) s.prob_out[sb * s.p_sb + st * s.p_st + s.p_off + col] = pr / jac;
    }
    if (s.eps2) {
        const bool at = in.at;
        const float x = l + in.ev2 * sc;
        const float t = tanhf(x);
        const float s0 = logf(fmaxf(1.f - t * t, kSquashFloor));
        const float s1 = normal_log_prob(x, l, sc);
        if (at) s.a2_out[sb * A + col] = t;
        const float lp = epi_row_logp(s0, s1, base, A);
        if (at && col == 0) s.logp2_out[sb] = lp;
    }
}

}  // namespace asac
