// Return / target kernels for gfx950: tanh-squashed Gaussian sampling and probabilities, the
// fused ensemble-min + V + V-trace n-step return (K4), and the clipped double-Q loss with its
// gradient.  C ABI in include/asac_hip.h.  All f32, evaluation order of the reference's eager ops
// (algorithm/sac_base.py:1244-1295, 1423-1464, 1539-1561; algorithm/utils/operators.py:12-31);
// built with -ffp-contract=off.
#include "asac_common.h"
#include "asac_sidecar.h"
#include "asac_squash.h"
#include "asac_common.h"
namespace asac {
// phase time stamps of the workgroup for tools/td_phases.py (100 MHz wall clock); compiled out of the library
#ifdef ASAC_TD_STAMPS
__device__ unsigned long long g_td_stamps[16];
#define TD_STAMP(i)                                              \
    do {                                                         \
        if (threadIdx.x == 0) g_td_stamps[i] = wall_clock64();   \
    } while (0)
#else
#define TD_STAMP(i)
#endif
}  // namespace asac
#include "asac_tree_update.h"
#include "asac_vtrace.h"

#include <cmath>

namespace asac {


// Several independent sampling / probability jobs in ONE launch (a train step issues up to three
// back-to-back on outputs of the same policy forward; at 256..1280 rows each a launch is pure latency).
struct SquashJobDev {
    const float *loc, *scale, *eps;      // eps == NULL: stored-action probabilities only
    int64_t ls, rows;
    int32_t A, first_block;
    float *a_out, *logp_out, *x_out;
    StoredProb sp;
};
struct SquashJobsDev {
    SquashJobDev j[ASAC_SQUASH_MAX_JOBS];
    int32_t n, blocks;
};

// One workgroup (256 threads) of a sampling / probability job.  The rows' transcendental chains (tanh, log, atanh, exp:
// ~4 us when one lane walks a row's A dimensions, whatever the batch) are spread over lanes: a workgroup owns
// R = 256 / A rows, lane (row, d) evaluates the per-element terms of squash_sample_at / stored_action_prob
// (asac_squash.h) and parks them in LDS, then the sums and products over the action dimension are formed in d order —
// the same values in the same order as the one-lane-per-row form, bit for bit.
inline int squash_rows_per_block(int A) { return 256 / A; }

__device__ __forceinline__ void squash_rows_block(const SquashJobDev& job, int64_t local_block, float* lds /* 512 floats */) {
    const int A = job.A, R = 256 / A;
    const int lrow = threadIdx.x / A, d = threadIdx.x - lrow * A;
    const int64_t r = local_block * R + lrow;
    const bool live = lrow < R && r < job.rows;
    float* s0 = lds;
    float* s1 = lds + 256;
    const float* lrow_p = job.loc + r * job.ls;
    const float* srow_p = job.scale + r * job.ls;
    float l = 0.f, sc = 1.f;
    if (live) l = lrow_p[d], sc = srow_p[d];
    if (job.eps) {
        if (live) {
            const float x = l + job.eps[r * A + d] * sc;
            const float t = tanhf(x);
            s0[threadIdx.x] = logf(fmaxf(1.f - t * t, kSquashFloor));
            s1[threadIdx.x] = normal_log_prob(x, l, sc);
            job.a_out[r * A + d] = t;
            if (job.x_out) job.x_out[r * A + d] = x;
        }
        __syncthreads();
        if (live && d == 0) {
            float corr = 0.f;
            for (int dd = 0; dd < A; ++dd) corr += s0[lrow * A + dd];
            float lp = 0.f;
            for (int dd = 0; dd < A; ++dd) {
                float v = s1[lrow * A + dd] - corr;      // correction broadcast to every component
                if (v == INFINITY) v = 0.f;              // sum_log_prob's inf mask
                lp += v;
            }
            job.logp_out[r] = lp;
        }
        if (!job.sp.action) return;
        __syncthreads();
    }
    // probabilities of the stored actions under the same Gaussian
    const StoredProb& sp = job.sp;
    int64_t sb = 0, st = 0;
    if (live) {
        sb = r / sp.T;
        st = r - sb * sp.T;
        const float x = atanhf(fminf(fmaxf(sp.action[sb * sp.a_sb + st * sp.a_st + sp.a_off + d], -0.999f), 0.999f));
        s0[threadIdx.x] = squash_jac(x);
        s1[threadIdx.x] = expf(normal_log_prob(x, l, sc));
    }
    __syncthreads();
    if (live) {
        float jac = 1.f;
        for (int dd = 0; dd < A; ++dd) jac *= s0[lrow * A + dd];
        sp.out[sb * sp.p_sb + st * sp.p_st + sp.p_off + d] = s1[threadIdx.x] / jac;
    }
}

__global__ __launch_bounds__(256) void k_squash_multi(const SquashJobsDev js, const SidecarsDev sc) {
    __shared__ float lds[512];
    if ((int)blockIdx.x >= js.blocks) {         // sidecar workgroups (asac_sidecar.h)
        sidecar_run(sc, (int)blockIdx.x - js.blocks, lds);
        return;
    }
    int k = 0;
#pragma unroll
    for (int q = 1; q < ASAC_SQUASH_MAX_JOBS; ++q)
        if (q < js.n && (int)blockIdx.x >= js.j[q].first_block) k = q;
    const SquashJobDev& job = js.j[k];
    squash_rows_block(job, (int)blockIdx.x - job.first_block, lds);
}

// d logp / d x_d  = A * 2 tanh(x_d) [1 - tanh^2 > floor]   (+ the Normal part cancels between the
// direct and the via-x path);  d logp / d scale_d (direct) = -1/scale_d.
__global__ __launch_bounds__(256) void k_squash_sample_bwd(
    const float* __restrict__ loc, const float* __restrict__ scale, int64_t ls, const float* __restrict__ eps,
    const float* __restrict__ grad_a, int grad_a_members, int64_t grad_a_member_stride,
    const float* __restrict__ grad_logp, const float* __restrict__ log_alpha, int64_t rows, int A,
    float* __restrict__ grad_loc, float* __restrict__ grad_scale, int64_t gs) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t base = r * A;
    // dL/dlogp: given per row, or the policy objective's constant alpha / rows (sac_base.py:1896)
    const float gl = grad_logp ? grad_logp[r] : (log_alpha ? expf(*log_alpha) * (1.f / (float)rows) : 0.f);
    for (int d = 0; d < A; ++d) {
        const float s = scale[r * ls + d], e = eps[base + d];
        const float x = loc[r * ls + d] + e * s;
        const float t = tanhf(x);
        const float one_m = 1.f - t * t;
        float ga = 0.f;     // the action feeds every ensemble member: sum their gradients in member order
        if (grad_a)
            for (int m = 0; m < grad_a_members; ++m) ga += grad_a[m * grad_a_member_stride + base + d];
        float gx = ga * one_m;
        if (one_m > kSquashFloor) gx += gl * ((float)A * 2.f * t);
        grad_loc[r * gs + d] = gx;
        grad_scale[r * gs + d] = gx * e - gl / s;
    }
}

// a single job (asac_squash_sample_fwd, asac_squash_prob)
__global__ __launch_bounds__(256) void k_squash_job(const SquashJobDev job) {
    __shared__ float lds[512];
    squash_rows_block(job, blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------------
// K4.  A workgroup owns R consecutive batch rows (as many as two LDS slabs allow).  Phase 1 (all
// 256 lanes, coalesced over (row, t), one round of global loads): V(s_t) = min_{e in subset_n} Q_e -
// alpha*logpi, V'(s_t+1) with subset_next, and from them everything of step t that does not depend on
// the running product: d_t = rho_t lambda^t gamma^t delta_t * mask and c_t = min(pi/mu, c_bar), into
// LDS slabs with an odd pitch.
// Phase 2: y = V(s_0) + sum_t (prod_{s<t} c_s) d_t with 256/R lanes per row, each scanning a
// contiguous segment; the (sum, product) pairs of the segments combine associatively by shuffles.
// ------------------------------------------------------------------------------------------------
struct VtraceDev {
    asac_vtrace_args_t a;
    int32_t R, pitch, seg, blocks;
};
template <int NSC>
struct VtraceExtra {                 // what the launches with sidecars / a pending temperature step carry on top
    SidecarsT<NSC> sc;
    int32_t has_pending, pad_;       // the temperature step is still pending: use the value it will produce
    AlphaAdamArgs pending;
};

__device__ __forceinline__ void vtrace_return_min_wg(const VtraceDev& v, float* lds, const AlphaAdamArgs* pending) {
    const asac_vtrace_args_t& a = v.a;
    const int n = a.n, R = v.R, pitch = v.pitch, SEG = v.seg;
    float* s_d = lds;                        // [R][pitch]  per-step term d_t
    float* s_c = s_d + R * pitch;            // [R][pitch]  trace-cutting factor c_t = min(pi/mu, c_bar)
    float* s_v0 = s_c + R * pitch;           // [R]         V(s_0)
    const int row0 = blockIdx.x * R;
    // phase 1 (all lanes, coalesced over (row, t), ONE round of global loads): everything of step t that
    // does not depend on the running product
    //   V(s_t)   = min_{e in subset_n}    Q_e(s_t, a_t)     - alpha logpi_t
    //   V(s_t+1) = min_{e in subset_next} Q_e(s_t+1, a_t+1) - alpha logpi_t+1
    //   d_t = rho_t * lambda^t * gamma^t * (r_t + gamma (1 - done_t) V(s_t+1) - V(s_t)) * ~(last | pad)
    // The first item's loads are issued before the temperature is worked out (a pending temperature step costs a
    // reduction and two f64 powers: they run while the loads travel).
    const int f0 = threadIdx.x;
    const bool have0 = f0 < R * n && row0 + f0 / n < a.B;
    VtraceStepRaw raw0{};
    if (have0) raw0 = vtrace_step_load(a, row0 + f0 / n, f0 - (f0 / n) * n);
    // ... and so are the online critics' values of the row this lane will finish (TD error variant)
    float q_on[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const int r = threadIdx.x / SEG, b = row0 + r;
        if (a.td_error_out && r < R && b < a.B && threadIdx.x == r * SEG) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q_on[j] = a.q_online[(int64_t)min(j, a.E_online - 1) * a.B + b];
        }
    }
    float alpha = a.q ? expf(*a.log_alpha) : 0.f;
    if (pending) alpha = expf(alpha_adam_preview(*pending, lds));
    for (int f = threadIdx.x; f < R * n; f += blockDim.x) {
        const int r = f / n, t = f - r * n;
        const int b = row0 + r;
        if (b >= a.B) continue;
        float d, c;
        const float v_t = f == f0 ? vtrace_step_finish(a, raw0, alpha, &d, &c) : vtrace_step_terms(a, b, t, alpha, &d, &c);
        if (t == 0) s_v0[r] = v_t;
        s_d[r * pitch + t] = d;
        s_c[r * pitch + t] = c;
    }
    __syncthreads();

    // phase 2: SEG lanes per row, each scans a contiguous segment of the n steps:
    //   S = sum_t (prod_{s<t in segment} c_s) d_t,  P = prod c;  segments combine as S_k + P_k * (rest)
    const int r = threadIdx.x / SEG, k = threadIdx.x - r * SEG;
    const int b = row0 + r;
    const bool valid = r < R && b < a.B;
    const int len = (n + SEG - 1) / SEG;
    const int t0 = min(n, k * len), t1 = min(n, t0 + len);
    float S = 0.f, P = 1.f;
    if (valid) {
        const float* d = s_d + r * pitch;
        const float* c = s_c + r * pitch;
        for (int t = t0; t < t1; ++t) {
            S += P * d[t];
            P *= c[t];
        }
    }
    for (int off = 1; off < SEG; off <<= 1) {
        const float S_hi = __shfl_down(S, off, 64), P_hi = __shfl_down(P, off, 64);
        S += P * S_hi;
        P *= P_hi;
    }
    if (!valid || k != 0) return;
    const float y = s_v0[r] + S;
    a.y_out[b] = y;
    if (a.td_error_out) {
        float s = 0.f;
        if (a.E_online <= 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < a.E_online) s += fabsf(q_on[e] - y);
        } else {
            for (int e = 0; e < a.E_online; ++e) s += fabsf(a.q_online[(int64_t)e * a.B + b] - y);
        }
        a.td_error_out[b] = s / (float)a.E_online;
    }
}

__global__ __launch_bounds__(256) void k_vtrace_return_min(const VtraceDev v) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    vtrace_return_min_wg(v, lds, nullptr);
}

template <int NSC>
__global__ __launch_bounds__(256) void k_vtrace_return_min_sc(const VtraceDev v, const VtraceExtra<NSC> x) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x >= v.blocks) {           // sidecar workgroups (asac_sidecar.h)
        sidecar_run(x.sc, (int)blockIdx.x - v.blocks, lds);
        return;
    }
    vtrace_return_min_wg(v, lds, x.has_pending ? &x.pending : nullptr);
}

// ------------------------------------------------------------------------------------------------
// The step's last two launches as ONE: the TD errors' return (K4 with mean_e|q_e - y|) and the priority update (K6)
// that consumes them.  The update is a single workgroup anyway (the tree climb is a chain of dependent rounds), so that
// workgroup first forms the returns of all B rows itself — one round of loads for its B n items, the scan of a row by one
// lane in the return kernel's association order (bit-identical) — and goes straight on to the election and the climb:
// no launch boundary, no trip of the TD errors through memory before their first use.  A pending temperature step
// (asac_sidecar.h ALPHA_ADAM) is RUN by this workgroup first (the return needs its result; nobody else reads the
// temperature in this launch); sidecar jobs are workgroups 1...  The ids, the id map's entries and the online
// critics' values of the rows are requested at kernel entry, under the return's loads; the last writer of a leaf is
// elected in LDS (asac_tree_update.h sumtree_update_wg_own): the `winner` scratch is not touched.
// LDS: [B][pitch] d_t | [B][pitch] c_t | [B] V(s_0) | [B + 4] leaves, pitch = (n + 1) | 1, each part 16-byte aligned.
// ------------------------------------------------------------------------------------------------
struct TdUpdateArgs {
    asac_vtrace_args_t a;
    float* tree;
    const int64_t* ids;
    const int64_t* slot_ids;
    int32_t* winner;
    int32_t* nan_flag;
    int32_t capacity, levels, seg, has_alpha;
    float alpha_pow, td_min, td_max;
    AlphaAdamArgs alpha;
};

template <int NSC>
__global__ __launch_bounds__(kUpdateBlock) void k_td_update(const TdUpdateArgs u, const SidecarsT<NSC> sc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (blockIdx.x > 0) {
        sidecar_run(sc, (int)blockIdx.x - 1, lds);
        return;
    }
    const asac_vtrace_args_t& a = u.a;
    const int n = a.n, B = a.B, pitch = (n + 1) | 1;
    TD_STAMP(0);
    float* s_d = lds;
    float* s_c = s_d + B * pitch;
    float* s_v0 = lds + ((2 * B * pitch + 3) & ~3);                     // (16-byte aligned: so are the arrays behind it)
    int* s_leaf = reinterpret_cast<int*>(s_v0 + ((B + 63) & ~63));      // [whole waves + 4]
    // everything that does not wait for the TD errors is requested now: this thread's row id (-> its ring slot, the id
    // map's entry: is the row still resident?), its item of the return; the temperature step's loads follow
    const int row = threadIdx.x;
    int64_t id = 0;
    if (row < B) id = u.ids[row];
    const int f0 = threadIdx.x;
    const bool have0 = f0 < B * n;
    VtraceStepRaw raw0{};
    if (have0) raw0 = vtrace_step_load(a, f0 / n, f0 - (f0 / n) * n);
    float q_on[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < B) {
#pragma unroll
        for (int j = 0; j < 4; ++j) q_on[j] = a.q_online[(int64_t)min(j, a.E_online - 1) * B + row];
    }
    // (nothing above is USED yet: the id -> ring slot -> id-map lookup is a dependent round trip, and taken here it held
    // the whole workgroup 2 us in front of the temperature step, whose own loads were not even issued — tools/td_phases.py.
    // The step's loads go out now, beside the row's; the id map is asked once the step is through and answers under
    // the return's phases)
    float log_alpha;
    TD_STAMP(1);
    if (u.has_alpha) log_alpha = alpha_adam_block(u.alpha, lds);
    else log_alpha = *a.log_alpha;
    __syncthreads();
    TD_STAMP(2);
    int slot = 0;
    int64_t resident = 0;
    if (row < B) {
        slot = ring_slot(id, u.capacity);
        resident = u.slot_ids ? u.slot_ids[slot] : id;
    }
    const float alpha = expf(log_alpha);
    for (int f = threadIdx.x; f < B * n; f += blockDim.x) {
        const int r = f / n, t = f - r * n;
        float d, c;
        const float v_t = f == f0 ? vtrace_step_finish(a, raw0, alpha, &d, &c) : vtrace_step_terms(a, r, t, alpha, &d, &c);
        if (t == 0) s_v0[r] = v_t;
        s_d[r * pitch + t] = d;
        s_c[r * pitch + t] = c;
    }
    __syncthreads();
    TD_STAMP(3);
    // what follows has one lane per ROW (B <= blockDim.x): waves without a row leave (a finished wave no longer counts
    // at the barriers of the climb: 4 waves instead of 16 at each of them for a batch of 256).  This leans on the CDNA
    // rule for S_BARRIER — "if some waves of the workgroup have already terminated, the barrier waits for the surviving
    // waves only" (CDNA3 / CDNA4 ISA guide, S_BARRIER) — not on the HIP programming model, which wants every thread at a
    // __syncthreads(): the condition is wave-uniform by construction (whole waves leave), and the file refuses to build
    // for any other target (below).  tests/test_kernels_gpu.py::test_td_error_and_priority_update_in_one_launch covers
    // batches that are not a multiple of the wave size and checks the tree invariant afterwards.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_td_update retires whole waves ahead of workgroup barriers: valid on gfx950 (CDNA S_BARRIER semantics) only"
#endif
    if ((int)(threadIdx.x & ~63u) >= B) return;
    float y = 0.f, td = 0.f;
    if (row < B) {
        y = s_v0[row] + vtrace_scan_row(s_d + row * pitch, s_c + row * pitch, n, u.seg);
        float s = 0.f;
        if (a.E_online <= 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < a.E_online) s += fabsf(q_on[e] - y);
        } else {
            for (int e = 0; e < a.E_online; ++e) s += fabsf(a.q_online[(int64_t)e * B + row] - y);
        }
        td = s / (float)a.E_online;
    }
    const int leaf1 = (row < B && resident == id) ? slot + u.capacity : 0;
    // (the outputs are stored with the leaves, behind the election)
    TD_STAMP(4);
    const bool fine = sumtree_update_wg_own(u.tree, u.levels, B, leaf1, td, u.alpha_pow, u.td_min, u.td_max, u.nan_flag, s_leaf,
                                            min((int)blockDim.x, (B + 63) & ~63),
                                            [&] { if (row < B) a.y_out[row] = y, a.td_error_out[row] = td; });
    if (!fine && row < B) a.y_out[row] = y, a.td_error_out[row] = td;
    TD_STAMP(7);
}

#ifdef ASAC_TD_STAMPS
extern "C" int asac_debug_td_stamps(unsigned long long* out_host) {
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_td_stamps), sizeof(g_td_stamps));
}
#endif

// precomputed-V variant of phase 1a (discrete / hybrid branches hand V in directly)
__global__ __launch_bounds__(256) void k_vtrace_direct(const VtraceDev v, const float* __restrict__ v_n,
                                                       const float* __restrict__ v_next,
                                                       const float* __restrict__ pi_prod,
                                                       const float* __restrict__ mu_prod) {
    const asac_vtrace_args_t& a = v.a;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int n = a.n;
    float cum_c = 1.f, acc = 0.f;
    for (int t = 0; t < n; ++t) {
        const int64_t mi = (int64_t)b * a.mask_stride + t;
        const float g = a.done[mi] ? 0.f : a.gamma;
        float td = a.reward[(int64_t)b * a.reward_stride + t] + g * v_next[(int64_t)b * n + t] - v_n[(int64_t)b * n + t];
        td = a.gamma_ratio[t] * td;
        if (a.use_n_step_is) {
            td = a.lambda_ratio[t] * td;
            const float ratio = pi_prod[(int64_t)b * n + t] / fmaxf(mu_prod[(int64_t)b * n + t], 1e-8f);
            td = (cum_c * fminf(ratio, a.v_rho)) * td;
            cum_c = cum_c * fminf(ratio, a.v_c);
        }
        td = td * ((a.last_mask[mi] | a.padding_mask[mi]) ? 0.f : 1.f);
        acc += td;
    }
    a.y_out[b] = v_n[(int64_t)b * n] + acc;
}

// ------------------------------------------------------------------------------------------------
// Clipped double-Q loss: one workgroup per ensemble member, fixed-order tree reduce (deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_q_loss(const float* __restrict__ q, const float* __restrict__ tq,
                                                const float* __restrict__ y, const float* __restrict__ w,
                                                int B, float clip_eps, float* __restrict__ loss_out,
                                                float* __restrict__ grad_out) {
    const int e = blockIdx.x;
    const float inv_b = 1.f / (float)B;
    float part = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float g;
        part += clipped_q_loss_row(q[(int64_t)e * B + b], tq[(int64_t)e * B + b], y[b], w ? w[b] : 1.f, clip_eps, &g);
        grad_out[(int64_t)e * B + b] = inv_b * g;
    }
    __shared__ float red[256];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[e] = red[0] * inv_b;
}

// Policy objective of the continuous head (reference sac_base.py:1882-1903), value + gradients in one
// single-workgroup launch:  L = mean_b( alpha * logp_b - min_{e in subset} q[e][b] )
//   dL/dlogp_b = alpha / B;   dL/dq[e][b] = -1/B at the (first) arg-min member, else 0
// plus the logging statistic mean_b sum_d (log scale + 1/2 + 1/2 log 2pi) (1910-1911).
__global__ __launch_bounds__(256) void k_policy_loss(const float* __restrict__ logp, const float* __restrict__ q,
                                                     const int32_t* __restrict__ subset, int E, int Es, int B,
                                                     const float* __restrict__ log_alpha,
                                                     const float* __restrict__ scale, int64_t scale_rs, int A,
                                                     float* __restrict__ loss_out, float* __restrict__ grad_logp,
                                                     float* __restrict__ grad_q, float* __restrict__ entropy_out) {
    const float alpha = expf(*log_alpha);
    const float inv_b = 1.f / (float)B;
    float part = 0.f, ent = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        int best = subset ? subset[0] : 0;
        float m = q[(int64_t)best * B + b];
        for (int k = 1; k < Es; ++k) {
            const int e = subset ? subset[k] : k;
            const float v = q[(int64_t)e * B + b];
            if (v < m) {
                m = v;
                best = e;
            }
        }
        for (int e = 0; e < E; ++e) grad_q[(int64_t)e * B + b] = (e == best) ? -inv_b : 0.f;
        grad_logp[b] = alpha * inv_b;
        part += alpha * logp[b] - m;
        if (scale)
            for (int d = 0; d < A; ++d) ent += logf(scale[(int64_t)b * scale_rs + d]) + 1.4189385332046727f;
    }
    __shared__ float red[2][256];
    red[0][threadIdx.x] = part;
    red[1][threadIdx.x] = ent;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss_out = red[0][0] * inv_b;
        if (entropy_out) *entropy_out = red[1][0] * inv_b;
    }
}

// Temperature objective, continuous head (reference sac_base.py:1931-1944):
//   L = mean_b( log_alpha * (-logp_b - target) ),  dL/dlog_alpha = mean_b(-logp_b) - target
// written straight into the flat gradient slot of log_c_alpha.
__global__ __launch_bounds__(256) void k_alpha_grad(const float* __restrict__ logp, int B, float target,
                                                    float* __restrict__ grad_slot) {
    float part = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) part += -logp[b] - target;
    __shared__ float red[256];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *grad_slot = red[0] / (float)B;
}

// Gaussian head bounding of the stock policy (reference nn_models/policy.py:170-172)
__global__ __launch_bounds__(256) void k_gauss_head_fwd(const float* __restrict__ raw, int64_t n, int A,
                                                        float* __restrict__ loc, float* __restrict__ scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / A;
    const int d = (int)(i - r * A);
    const float mean = raw[r * 2 * A + d], logstd = raw[r * 2 * A + A + d];
    loc[i] = tanhf(mean / 5.f) * 5.f;
    scale[i] = expf(fminf(fmaxf(logstd, -20.f), 0.5f));
}

__global__ __launch_bounds__(256) void k_gauss_head_bwd(const float* __restrict__ raw, const float* __restrict__ gloc,
                                                        const float* __restrict__ gscale, int64_t n, int A,
                                                        float* __restrict__ graw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / A;
    const int d = (int)(i - r * A);
    const float mean = raw[r * 2 * A + d], logstd = raw[r * 2 * A + A + d];
    const float t = tanhf(mean / 5.f);
    graw[r * 2 * A + d] = gloc ? gloc[i] * (1.f - t * t) : 0.f;          // d(5 tanh(m/5))/dm
    const bool open = logstd >= -20.f && logstd <= 0.5f;
    graw[r * 2 * A + A + d] = (gscale && open) ? gscale[i] * expf(logstd) : 0.f;
}

}  // namespace asac

using namespace asac;

static int set_lds_limit_fn(const void* fn, size_t bytes, bool& done, const char* where) {
    if (done) return 0;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err != hipSuccess) {
        set_error(err, where);
        return (int)err;
    }
    done = true;
    return 0;
}

extern "C" {

int asac_squash_sample_fwd(const float* loc, const float* scale, int64_t ls_row_stride, const float* eps,
                           int64_t rows, int A, float* a_tanh_out, float* logp_out, float* x_out,
                           const float* action, int T, int64_t action_stride_b, int64_t action_stride_t,
                           int action_offset, float* prob_out, int64_t prob_stride_b, int64_t prob_stride_t,
                           int prob_offset, void* stream) {
    if (rows <= 0 || A <= 0 || A > ASAC_MAX_ACTION || (action && (T <= 0 || !prob_out)))
        return bad_arg("asac_squash_sample_fwd");
    if (!eps || !a_tanh_out || !logp_out) return bad_arg("asac_squash_sample_fwd: outputs");
    SquashJobDev job{};
    job.loc = loc; job.scale = scale; job.eps = eps;
    job.ls = ls_row_stride; job.rows = rows; job.A = A;
    job.a_out = a_tanh_out; job.logp_out = logp_out; job.x_out = x_out;
    job.sp = StoredProb{action, T, action_stride_b, action_stride_t, action_offset,
                        prob_out, prob_stride_b, prob_stride_t, prob_offset};
    const int R = squash_rows_per_block(A);
    ASAC_LAUNCH(k_squash_job, dim3((unsigned)((rows + R - 1) / R)), dim3(256), 0, as_stream(stream), job);
    return finish_launch("asac_squash_sample_fwd");
}

int asac_squash_sample_bwd(const float* loc, const float* scale, int64_t ls_row_stride, const float* eps,
                           const float* grad_a, int grad_a_members, int64_t grad_a_member_stride,
                           const float* grad_logp, const float* log_alpha, int64_t rows, int A,
                           float* grad_loc, float* grad_scale, int64_t grad_row_stride, void* stream) {
    if (rows <= 0 || A <= 0 || A > ASAC_MAX_ACTION || (grad_a && grad_a_members < 1))
        return bad_arg("asac_squash_sample_bwd");
    ASAC_LAUNCH(k_squash_sample_bwd, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                as_stream(stream), loc, scale, ls_row_stride, eps, grad_a, grad_a_members, grad_a_member_stride,
                grad_logp, log_alpha, rows, A, grad_loc, grad_scale, grad_row_stride);
    return finish_launch("asac_squash_sample_bwd");
}

int asac_squash_multi(const asac_squash_job_t* jobs_host, int n_jobs, const asac_sidecar_t* sidecars_host,
                      int n_sidecars, void* stream) {
    if (!jobs_host || n_jobs < 1 || n_jobs > ASAC_SQUASH_MAX_JOBS) return bad_arg("asac_squash_multi");
    SidecarsDev sc{}, none{};
    if (sidecars_prepare(sidecars_host, n_sidecars, sc)) return bad_arg("asac_squash_multi: sidecar");
    SquashJobsDev js{};
    js.n = n_jobs;
    int blocks = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const asac_squash_job_t& h = jobs_host[k];
        if (h.rows <= 0 || h.A <= 0 || h.A > ASAC_MAX_ACTION || !h.loc || !h.scale) return bad_arg("asac_squash_multi: job");
        if (h.eps && (!h.a_tanh_out || !h.logp_out)) return bad_arg("asac_squash_multi: sample outputs");
        if (!h.eps && (!h.action || !h.prob_out)) return bad_arg("asac_squash_multi: empty job");
        if (h.action && (!h.prob_out || h.T <= 0)) return bad_arg("asac_squash_multi: stored-action outputs");
        SquashJobDev& d = js.j[k];
        d.loc = h.loc; d.scale = h.scale; d.eps = h.eps;
        d.ls = h.ls_row_stride; d.rows = h.rows; d.A = h.A;
        d.first_block = blocks;
        d.a_out = h.a_tanh_out; d.logp_out = h.logp_out; d.x_out = h.x_out;
        d.sp = StoredProb{h.action, h.T, h.action_stride_b, h.action_stride_t, h.action_offset,
                          h.prob_out, h.prob_stride_b, h.prob_stride_t, h.prob_offset};
        const int R = squash_rows_per_block(h.A);
        blocks += (int)((h.rows + R - 1) / R);
    }
    js.blocks = blocks;
    // (under the measurement repeat knob only the last repetition carries the sidecars)
    for (int rep = 0; rep < g_launch_repeat; ++rep) {
        const bool last = rep == g_launch_repeat - 1;
        hipLaunchKernelGGL(k_squash_multi, dim3((unsigned)(blocks + (last ? sc.blocks : 0))), dim3(256), 0,
                           as_stream(stream), js, last ? sc : none);
    }
    return finish_launch("asac_squash_multi");
}

int asac_squash_prob(const float* loc, const float* scale, int64_t ls_row_stride, const float* action, int T,
                     int64_t action_stride_b, int64_t action_stride_t, int action_offset,
                     int64_t rows, int A, float* prob_out, int64_t prob_stride_b,
                     int64_t prob_stride_t, int prob_offset, void* stream) {
    if (rows <= 0 || A <= 0 || A > ASAC_MAX_ACTION || T <= 0 || !action || !prob_out)
        return bad_arg("asac_squash_prob");
    SquashJobDev job{};
    job.loc = loc; job.scale = scale;
    job.ls = ls_row_stride; job.rows = rows; job.A = A;
    job.sp = StoredProb{action, T, action_stride_b, action_stride_t, action_offset,
                        prob_out, prob_stride_b, prob_stride_t, prob_offset};
    const int R = squash_rows_per_block(A);
    ASAC_LAUNCH(k_squash_job, dim3((unsigned)((rows + R - 1) / R)), dim3(256), 0, as_stream(stream), job);
    return finish_launch("asac_squash_prob");
}

int asac_vtrace_return_min(const asac_vtrace_args_t* args_host, void* stream) {
    return asac_vtrace_return_min_sc(args_host, nullptr, 0, nullptr, stream);
}

int asac_vtrace_return_min_sc(const asac_vtrace_args_t* args_host, const asac_sidecar_t* sidecars_host, int n_sidecars,
                              const asac_sidecar_t* pending_alpha, void* stream) {
    const asac_vtrace_args_t& h = *args_host;
    SidecarsDev sc{}, none{}, pend{};
    if (sidecars_prepare(sidecars_host, n_sidecars, sc)) return bad_arg("asac_vtrace_return_min: sidecar");
    if (pending_alpha) {
        if (pending_alpha->kind != ASAC_SIDECAR_ALPHA_ADAM || sidecars_prepare(pending_alpha, 1, pend) ||
            h.log_alpha != pending_alpha->param + pending_alpha->slot)
            return bad_arg("asac_vtrace_return_min: pending temperature step");
    }
    if (h.B <= 0 || h.n <= 0 || !h.y_out || !h.q || h.E_sample <= 0 || h.E_sample > ASAC_MAX_ENSEMBLE)
        return bad_arg("asac_vtrace_return_min");
    if (h.use_n_step_is && (!h.mu_prob || !h.pi_prob || h.A <= 0)) return bad_arg("asac_vtrace_return_min: is");
    VtraceDev v{};
    v.a = h;
    v.pitch = (h.n + 1) | 1;                      // odd pitch: conflict-free row-per-lane reads
    // Rows per workgroup.  A lane's items in phase 1 are sequential global round trips, so small batches
    // get few rows per workgroup (about two (row, t) items per lane) and spread over many workgroups;
    // larger ones use 64 rows x 4 lanes per row in the scan phase; very large short-window batches 256
    // rows with one lane per row (fewer and fatter workgroups).
    const int64_t items = (int64_t)h.B * h.n;
    const bool huge = items >= (1 << 21) && h.n <= 8;
    int R = huge ? 256 : 64;
    if (items < (1 << 17))
        while (R > 1 && R * h.n > 512) R >>= 1;
    while (R > 1 && (size_t)(2 * R * v.pitch + R) * sizeof(float) > 64 * 1024) R >>= 1;
    v.R = R;
    v.seg = vtrace_scan_lanes(h.B, h.n);
    size_t lds = (size_t)(2 * R * v.pitch + R) * sizeof(float);
    if (lds > 64 * 1024) return bad_arg("asac_vtrace_return_min: n too large");
    if (lds < 256 * sizeof(float)) lds = 256 * sizeof(float);        // (a sidecar workgroup's reduction scratch)
    const int blocks = (h.B + R - 1) / R;
    v.blocks = blocks;
    if (sc.n == 0 && !pending_alpha) {
        ASAC_LAUNCH(k_vtrace_return_min, dim3((unsigned)blocks), dim3(256), lds, as_stream(stream), v);
        return finish_launch("asac_vtrace_return_min");
    }
    for (int rep = 0; rep < g_launch_repeat; ++rep) {      // (repeat knob: only the last repetition carries the sidecars)
        const bool last = rep == g_launch_repeat - 1;
        const unsigned grid = (unsigned)(blocks + (last ? sc.blocks : 0));
        if (sc.n <= 1) {
            VtraceExtra<1> x{sidecars_first<1>(last ? sc : none), pending_alpha ? 1 : 0, 0, pend.j[0].alpha};
            hipLaunchKernelGGL(k_vtrace_return_min_sc<1>, dim3(grid), dim3(256), lds, as_stream(stream), v, x);
        } else {
            VtraceExtra<ASAC_MAX_SIDECARS> x{last ? sc : none, pending_alpha ? 1 : 0, 0, pend.j[0].alpha};
            hipLaunchKernelGGL(k_vtrace_return_min_sc<ASAC_MAX_SIDECARS>, dim3(grid), dim3(256), lds, as_stream(stream), v, x);
        }
    }
    return finish_launch("asac_vtrace_return_min");
}

int asac_td_update(const asac_vtrace_args_t* args_host, float* tree, int capacity, const int64_t* ids,
                   const int64_t* slot_ids, float alpha, float td_min, float td_max, int32_t* winner, int32_t* nan_flag,
                   const asac_sidecar_t* sidecars_host, int n_sidecars, const asac_sidecar_t* alpha_step, void* stream) {
    if (!args_host) return bad_arg("asac_td_update");
    const asac_vtrace_args_t& h = *args_host;
    if (h.B <= 0 || h.B > kUpdateBlock || h.n <= 0 || !h.y_out || !h.q || h.E_sample <= 0 || h.E_sample > ASAC_MAX_ENSEMBLE ||
        !h.q_online || h.E_online <= 0 || !h.td_error_out || (h.use_n_step_is && (!h.mu_prob || !h.pi_prob || h.A <= 0)))
        return bad_arg("asac_td_update: return arguments");
    if (!tree || capacity <= 0 || (capacity & (capacity - 1)) || !ids || !winner || !nan_flag)
        return bad_arg("asac_td_update: tree arguments");
    SidecarsDev sc{}, none{}, al{};
    if (sidecars_prepare(sidecars_host, n_sidecars, sc)) return bad_arg("asac_td_update: sidecar");
    if (alpha_step && (alpha_step->kind != ASAC_SIDECAR_ALPHA_ADAM || sidecars_prepare(alpha_step, 1, al) ||
                       h.log_alpha != alpha_step->param + alpha_step->slot))
        return bad_arg("asac_td_update: temperature step");
    const int pitch = (h.n + 1) | 1;
    // the return's tiles, V(s_0), the leaves of the election (a whole number of waves)
    size_t lds = (size_t)(((2 * h.B * pitch + 3) & ~3) + 2 * ((h.B + 63) & ~63) + 4) * sizeof(float);
    if (lds > 128 * 1024) return bad_arg("asac_td_update: window too long for one workgroup");
    if (lds < 256 * sizeof(float)) lds = 256 * sizeof(float);
    TdUpdateArgs u{};
    const int threads = (int64_t)h.B * h.n <= 256 ? 256 : kUpdateBlock;
    u.a = h;
    u.tree = tree; u.ids = ids; u.slot_ids = slot_ids; u.winner = winner; u.nan_flag = nan_flag;
    u.capacity = capacity; u.levels = ilog2(capacity); u.seg = vtrace_scan_lanes(h.B, h.n);
    u.alpha_pow = alpha; u.td_min = td_min; u.td_max = td_max;
    u.has_alpha = alpha_step ? 1 : 0;
    if (alpha_step) u.alpha = al.j[0].alpha;
    static bool attr1 = false, attr4 = false;
    for (int rep = 0; rep < g_launch_repeat; ++rep) {      // (repeat knob: sidecars and the temperature step ride once)
        const bool last = rep == g_launch_repeat - 1;
        TdUpdateArgs ur = u;
        if (!last) ur.has_alpha = 0;
        const dim3 grid(1u + (unsigned)(last ? sc.blocks : 0));
        if (sc.n <= 1) {
            if (int rc = set_lds_limit_fn(reinterpret_cast<const void*>(k_td_update<1>), 128 * 1024, attr1, "asac_td_update")) return rc;
            hipLaunchKernelGGL(k_td_update<1>, grid, dim3(threads), lds, as_stream(stream), ur, sidecars_first<1>(last ? sc : none));
        } else {
            if (int rc = set_lds_limit_fn(reinterpret_cast<const void*>(k_td_update<ASAC_MAX_SIDECARS>), 128 * 1024, attr4, "asac_td_update")) return rc;
            hipLaunchKernelGGL(k_td_update<ASAC_MAX_SIDECARS>, grid, dim3(threads), lds, as_stream(stream), ur, last ? sc : none);
        }
    }
    return finish_launch("asac_td_update");
}

int asac_vtrace_return_direct(const asac_vtrace_args_t* args_host, const float* v_n,
                               const float* v_next, const float* pi_prod, const float* mu_prod,
                               void* stream) {
    const asac_vtrace_args_t& h = *args_host;
    if (h.B <= 0 || h.n <= 0 || !h.y_out || !v_n || !v_next) return bad_arg("asac_vtrace_return_direct");
    if (h.use_n_step_is && (!pi_prod || !mu_prod)) return bad_arg("asac_vtrace_return_direct: is");
    VtraceDev v{};
    v.a = h;
    v.R = v.pitch = v.seg = 0;
    ASAC_LAUNCH(k_vtrace_direct, dim3((h.B + 255) / 256), dim3(256), 0, as_stream(stream), v,
                       v_n, v_next, pi_prod, mu_prod);
    return finish_launch("asac_vtrace_return_direct");
}

int asac_policy_loss_fwd_bwd(const float* logp, const float* q, const int32_t* subset, int E, int E_sample,
                             int B, const float* log_alpha, const float* scale, int64_t scale_row_stride, int A,
                             float* loss_out, float* grad_logp, float* grad_q, float* entropy_out, void* stream) {
    if (E <= 0 || E_sample <= 0 || E_sample > E || B <= 0 || !loss_out || !grad_logp || !grad_q)
        return bad_arg("asac_policy_loss_fwd_bwd");
    ASAC_LAUNCH(k_policy_loss, dim3(1), dim3(256), 0, as_stream(stream), logp, q, subset, E, E_sample, B,
                log_alpha, scale, scale_row_stride, A, loss_out, grad_logp, grad_q, entropy_out);
    return finish_launch("asac_policy_loss_fwd_bwd");
}

int asac_alpha_grad(const float* logp, int B, float target, float* grad_slot, void* stream) {
    if (B <= 0 || !grad_slot) return bad_arg("asac_alpha_grad");
    ASAC_LAUNCH(k_alpha_grad, dim3(1), dim3(256), 0, as_stream(stream), logp, B, target, grad_slot);
    return finish_launch("asac_alpha_grad");
}

int asac_gauss_head_fwd(const float* raw, int64_t rows, int A, float* loc, float* scale, void* stream) {
    if (rows <= 0 || A <= 0) return bad_arg("asac_gauss_head_fwd");
    const int64_t n = rows * A;
    ASAC_LAUNCH(k_gauss_head_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), raw, n, A,
                loc, scale);
    return finish_launch("asac_gauss_head_fwd");
}

int asac_gauss_head_bwd(const float* raw, const float* grad_loc, const float* grad_scale,
                        int64_t rows, int A, float* grad_raw, void* stream) {
    if (rows <= 0 || A <= 0) return bad_arg("asac_gauss_head_bwd");
    const int64_t n = rows * A;
    ASAC_LAUNCH(k_gauss_head_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), raw,
                grad_loc, grad_scale, n, A, grad_raw);
    return finish_launch("asac_gauss_head_bwd");
}

int asac_q_loss_fwd_bwd(const float* q, const float* tq, const float* y, const float* w, int E,
                        int B, float clip_eps, float* loss_out, float* grad_q_out, void* stream) {
    if (E <= 0 || B <= 0 || clip_eps <= 0.f) return bad_arg("asac_q_loss_fwd_bwd");
    ASAC_LAUNCH(k_q_loss, dim3(E), dim3(256), 0, as_stream(stream), q, tq, y, w, B, clip_eps,
                       loss_out, grad_q_out);
    return finish_launch("asac_q_loss_fwd_bwd");
}

}  // extern "C"
