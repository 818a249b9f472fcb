// Parameter-update kernels for gfx950 over flat f32 buffers: Polyak soft update (K5) and Adam.
// C ABI in include/asac_hip.h.  Pure streaming kernels: 12 B/param (Polyak), 28 B/param (Adam);
// 16-byte loads/stores, grid capped at 2048 workgroups with a grid-stride loop.
#include "asac_common.h"
#include "asac_gelu.h"
#include "asac_sidecar.h"

#include <cmath>

namespace asac {

__global__ __launch_bounds__(256) void k_polyak(float* __restrict__ target, const float* __restrict__ source,
                                                int64_t n, float one_m_tau, float tau) {
    polyak_span(target, source, n, one_m_tau, tau, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                (int64_t)gridDim.x * blockDim.x);
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ param, const float* __restrict__ grad,
                                              float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                              int64_t n, AdamScalars c, const int64_t* __restrict__ steps_done) {
    float step_size, bc2_sqrt;
    adam_bias_terms(c, *steps_done, &step_size, &bc2_sqrt);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool aligned = ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                           reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
    int64_t done = 0;
    if (aligned) {
        const int64_t n4 = n / 4;
        for (int64_t i = tid; i < n4; i += stride) {
            float4 p = reinterpret_cast<float4*>(param)[i];
            const float4 g = reinterpret_cast<const float4*>(grad)[i];
            float4 m = reinterpret_cast<float4*>(exp_avg)[i];
            float4 v = reinterpret_cast<float4*>(exp_avg_sq)[i];
            adam1(p.x, g.x, m.x, v.x, c, step_size, bc2_sqrt);
            adam1(p.y, g.y, m.y, v.y, c, step_size, bc2_sqrt);
            adam1(p.z, g.z, m.z, v.z, c, step_size, bc2_sqrt);
            adam1(p.w, g.w, m.w, v.w, c, step_size, bc2_sqrt);
            reinterpret_cast<float4*>(param)[i] = p;
            reinterpret_cast<float4*>(exp_avg)[i] = m;
            reinterpret_cast<float4*>(exp_avg_sq)[i] = v;
        }
        done = n4 * 4;
    }
    for (int64_t i = done + tid; i < n; i += stride)
        adam1(param[i], grad[i], exp_avg[i], exp_avg_sq[i], c, step_size, bc2_sqrt);
}

// Temperature step in one launch: asac_sidecar.h `alpha_adam_block`
__global__ __launch_bounds__(256) void k_alpha_adam(const AlphaAdamArgs a) {
    __shared__ float red[256];
    alpha_adam_block(a, red);
}

// Adam over the E member segments of a stock network whose parameter gradients are still per-tile
// partial sums (asac_mlp_backward* with ASAC_MLP_REDUCE_DEFER): the fixed-order tile sum (the same
// order k_mlp_reduce_partials uses), the optional accumulation into grad, and the update in one pass.
// sum over the tiles of one parameter's partial gradients in tile order (the order k_mlp_reduce_partials uses), the
// loads issued sixteen at a time: a plain `for (t) s += partial[t]` compiles to load -> wait -> add per tile, i.e. one
// dependent L2 round trip per tile (sixteen of them at batch 256: most of this launch); sixteen = the tiles of a
// batch-256 backward, so the step's Adam launches wait for ONE round trip (eight at a time made it two)
__device__ __forceinline__ float sum_tiles(const float* __restrict__ partial, int64_t tile_stride, int tiles) {
    float s = 0.f;
    for (int t0 = 0; t0 < tiles; t0 += 16) {
        float v[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int t = t0 + w < tiles ? t0 + w : tiles - 1;         // clamped: every load is issued, none branches
            v[w] = partial[(int64_t)t * tile_stride];
        }
#pragma unroll
        for (int w = 0; w < 16; ++w) s += t0 + w < tiles ? v[w] : 0.f;
    }
    return s;
}

__global__ __launch_bounds__(256) void k_adam_partials(float* __restrict__ param, float* __restrict__ grad,
                                                       float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                       int64_t n, AdamScalars c, const int64_t* __restrict__ steps_done,
                                                       const float* __restrict__ partial, int tiles, int E,
                                                       int64_t member_stride, int64_t used, int accumulate,
                                                       const float* __restrict__ loss_partial,
                                                       float* __restrict__ loss_out, float inv_n) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tile_stride = (int64_t)E * member_stride;
    // this thread's first parameter: its loads are in flight while the (double precision) bias corrections are
    // formed; launches of this kernel have one parameter per thread
    float g = 0.f, p = 0.f, m = 0.f, v = 0.f, s = 0.f;
    bool summed = false;
    if (tid < n) {
        const int64_t e = tid / member_stride, local = tid - e * member_stride;
        g = grad[tid], p = param[tid], m = exp_avg[tid], v = exp_avg_sq[tid];
        summed = local < used;
        if (summed) s = sum_tiles(partial + e * member_stride + local, tile_stride, tiles);
    }
    float l = 0.f;
    if (loss_partial && tid < E) l = sum_tiles(loss_partial + tid, E, tiles);
    float step_size, bc2_sqrt;
    adam_bias_terms(c, *steps_done, &step_size, &bc2_sqrt);
    if (loss_partial && tid < E) loss_out[tid] = l * inv_n;
    if (tid < n) {
        if (summed) {
            g = accumulate ? g + s : s;
            grad[tid] = g;
        }
        adam1(p, g, m, v, c, step_size, bc2_sqrt);
        param[tid] = p, exp_avg[tid] = m, exp_avg_sq[tid] = v;
    }
    for (int64_t i = tid + stride; i < n; i += stride) {
        const int64_t e = i / member_stride, local = i - e * member_stride;
        float gi = grad[i];
        if (local < used) {
            const float si = sum_tiles(partial + e * member_stride + local, tile_stride, tiles);
            gi = accumulate ? gi + si : si;
            grad[i] = gi;
        }
        adam1(param[i], gi, exp_avg[i], exp_avg_sq[i], c, step_size, bc2_sqrt);
    }
}

// Curiosity (reference sac_base.py:1333-1343): the sampled reward window is augmented in place by
//   strength * 0.5 * sum_k (approx[b][t][k] - actual[b][t][k])^2
// `approx` is the dynamics model's dense output, `actual` a strided view of the window (the next states for the
// FORWARD model, the stored actions for the INVERSE one).  One lane per (b, t): one launch instead of ATen's five.
__global__ __launch_bounds__(256) void k_curiosity_bonus(const float* __restrict__ approx, const float* __restrict__ actual,
                                                         int64_t actual_sb, int64_t actual_st, float* __restrict__ reward,
                                                         int64_t reward_sb, int B, int T, int K, float strength) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;
    const float* p = approx + (int64_t)i * K;
    const float* q = actual + b * actual_sb + t * actual_st;
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const float d = p[k] - q[k];
        s += d * d;
    }
    float* r = reward + b * reward_sb + t;
    *r = *r + strength * (s * 0.5f);
}

// Loss of the curiosity model (reference sac_base.py:1951-1976): mean over ALL N = B T K elements of the squared
// error with padded rows zeroed, and its gradient with respect to the prediction:
//   d = (pred - target) * !mask[b][t];  loss = sum d^2 / N;  grad = d * 2 / N
// Eight elements per lane; the workgroups' sums are added in workgroup order by the last one to arrive.
constexpr int kMseThreads = 256, kMsePerLane = 8;
__global__ __launch_bounds__(kMseThreads) void k_masked_mse(const float* __restrict__ pred, const float* __restrict__ target,
                                                            int64_t target_sb, int64_t target_st,
                                                            const uint8_t* __restrict__ mask, int64_t mask_sb, int B, int T,
                                                            int K, float* __restrict__ grad, float* __restrict__ loss,
                                                            float* __restrict__ partial, unsigned int* __restrict__ counter) {
    __shared__ float red[kMseThreads];
    __shared__ bool last;
    const int N = B * T * K;                         // (N <= 2^20: 32-bit index arithmetic)
    const float scale = 2.f / (float)N;
    const int base = blockIdx.x * kMseThreads * kMsePerLane;
    // eight elements per lane, their loads requested together (a rolled loop waits for every pair in turn)
    float pv[kMsePerLane], qv[kMsePerLane], s = 0.f;
    bool pad[kMsePerLane];
#pragma unroll
    for (int u = 0; u < kMsePerLane; ++u) {
        const int i = min(base + u * kMseThreads + (int)threadIdx.x, N - 1);
        const int row = i / K, k = i - row * K, b = row / T, t = row - b * T;
        pv[u] = pred[i];
        qv[u] = target[b * target_sb + t * target_st + k];
        pad[u] = mask && mask[b * mask_sb + t];
    }
#pragma unroll
    for (int u = 0; u < kMsePerLane; ++u) {
        const int i = base + u * kMseThreads + (int)threadIdx.x;
        if (i < N) {
            const float d = pad[u] ? 0.f : pv[u] - qv[u];
            s += d * d;
            grad[i] = d * scale;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = kMseThreads / 2; w >= 64; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 64) {                          // the last strides inside wave 0
        float v = red[threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (threadIdx.x == 0) partial[blockIdx.x] = v;
    }
    // the last workgroup to arrive adds the workgroups' sums in workgroup order (deterministic)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x == 0) {
        float v = 0.f;
        for (unsigned int w = 0; w < gridDim.x; ++w) v += partial[w];
        *loss = v / (float)N;
        *counter = 0u;                               // ready for the next launch
    }
}

// The same loss over MILLIONS of elements — a plugin's observation loss `mse_loss(decoded frames, frames)`, 11 M floats at
// BASELINE configs[4] (sac_base.py:1817 through `ModelObservation.get_loss`) — for which ATen runs a 44 MB elementwise
// pass, a split reduction and, backwards, a fill and another elementwise pass: one launch, loss and gradient (the scaled
// difference) from ONE read of both operands.  Rows of K % 4 == 0 floats, 16-byte loads, a bounded grid of persistent
// workgroups; the workgroups' sums are added in a fixed order by the last to arrive.  The exchange uses relaxed
// agent-scope atomics only: an agent-scope release would write back every dirty line of the XCD's L2, i.e. the
// gradient this very launch is streaming out (sumtree.hip, SampleSync).
constexpr int kMseBigGrid = 2048;
__global__ __launch_bounds__(kMseThreads) void k_mse_big(const float* __restrict__ pred, const float* __restrict__ target,
                                                         int64_t target_sb, int64_t target_st, int T, int K4,
                                                         unsigned int n4, float inv_n, float grad_scale,
                                                         float* __restrict__ grad, float* __restrict__ loss, float* partial,
                                                         unsigned int* counter) {
    __shared__ float red[kMseThreads];
    __shared__ bool last;
    const float scale = 2.f * inv_n * grad_scale;
    const float4* p4 = reinterpret_cast<const float4*>(pred);
    float4* g4 = reinterpret_cast<float4*>(grad);
    constexpr int U = 4;
    float s = 0.f;
    auto term = [&](const float4& a, const float4& b, float4* out) {
        const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
        s += ((d.x * d.x + d.y * d.y) + d.z * d.z) + d.w * d.w;
        *out = make_float4(d.x * scale, d.y * scale, d.z * scale, d.w * scale);
    };
    if (target_st == (int64_t)4 * K4) {
        // a sample's T rows are consecutive in the target too (a slice x[:, b:] of the window batch): a workgroup walks
        // whole samples — no index arithmetic per element (two 32-bit divisions per float4 made the general form
        // 40 us for 133 MB)
        const unsigned int len4 = (unsigned)T * (unsigned)K4, B = n4 / len4;
        for (unsigned int smp = blockIdx.x; smp < B; smp += gridDim.x) {
            const float4* pc = p4 + (int64_t)smp * len4;
            const float4* tc = reinterpret_cast<const float4*>(target + (int64_t)smp * target_sb);
            float4* gc = g4 + (int64_t)smp * len4;
            for (unsigned int base = threadIdx.x; base < len4; base += kMseThreads * U) {
                float4 pv[U], qv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned int i = min(base + u * kMseThreads, len4 - 1);
                    pv[u] = pc[i], qv[u] = tc[i];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base + u * kMseThreads < len4) term(pv[u], qv[u], gc + base + u * kMseThreads);
            }
        }
    } else {
        for (unsigned int base = blockIdx.x * (kMseThreads * U); base < n4; base += gridDim.x * (kMseThreads * U)) {
            float4 pv[U], qv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned int i = min(base + u * kMseThreads + threadIdx.x, n4 - 1);
                const unsigned int row = i / (unsigned)K4, k4 = i - row * (unsigned)K4, b = row / (unsigned)T, t = row - b * (unsigned)T;
                pv[u] = p4[i];
                qv[u] = reinterpret_cast<const float4*>(target + (int64_t)b * target_sb + (int64_t)t * target_st)[k4];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned int i = base + u * kMseThreads + threadIdx.x;
                if (i < n4) term(pv[u], qv[u], g4 + i);
            }
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = kMseThreads / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    // fixed order: lane t adds partials [8 t, 8 t + 8) in turn, then the tree above
    // (all eight requested before the first is added: one by one each waited for its own trip to memory)
    constexpr int PER = kMseBigGrid / kMseThreads;
    float t[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const unsigned int w = min(threadIdx.x * PER + j, gridDim.x - 1);
        t[j] = __hip_atomic_load(partial + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) v += threadIdx.x * PER + j < gridDim.x ? t[j] : 0.f;
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = kMseThreads / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss = red[0] * inv_n;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
}

// Transition model's loss (SAC_Base._train_rpm, reference sac_base.py:1798-1816): for the model's Normal(loc, scale)
// over the next state and the target representation's state x,
//   loss = -mean(log N(x; loc, scale)) + w * mean(KL(N(loc, scale) || N(0, 1))),  entropy = mean(H(N(loc, scale)))
// with torch.distributions' formulas, and d loss / d loc, d loss / d scale — the forward AND the backward of ATen's
// ~25 + ~35 elementwise launches over [B, n, S] tensors as one launch.  Sums over several workgroups, added in workgroup
// order by the last to arrive.
__global__ __launch_bounds__(kMseThreads) void k_normal_nll_kl(const float* __restrict__ loc, int64_t loc_sb, int64_t loc_st,
                                                               const float* __restrict__ scale, int64_t scale_sb,
                                                               int64_t scale_st, const float* __restrict__ target,
                                                               int64_t target_sb, int64_t target_st, int B, int T, int K,
                                                               float w, float* __restrict__ grad_loc,
                                                               float* __restrict__ grad_scale, float* __restrict__ out,
                                                               float* __restrict__ partial, unsigned int* __restrict__ counter,
                                                               int log_scale, float scale_min, float scale_max,
                                                               int64_t grad_pitch) {
    __shared__ float red[3][kMseThreads];
    __shared__ bool last;
    const int N = B * T * K;
    const float inv_n = 1.f / (float)N;
    const float c_lp = 0.918938533204672742f;        // log(sqrt(2 pi))
    const float c_ent = 1.418938533204672742f;       // 0.5 + 0.5 log(2 pi)
    const int base = blockIdx.x * kMseThreads * kMsePerLane;
    float mu[kMsePerLane], sg[kMsePerLane], xv[kMsePerLane], s_lp = 0.f, s_kl = 0.f, s_ent = 0.f;
#pragma unroll
    for (int u = 0; u < kMsePerLane; ++u) {
        const int i = min(base + u * kMseThreads + (int)threadIdx.x, N - 1);
        const int row = i / K, k = i - row * K, b = row / T, t = row - b * T;
        mu[u] = loc[b * loc_sb + t * loc_st + k];
        sg[u] = scale[b * scale_sb + t * scale_st + k];
        xv[u] = target[b * target_sb + t * target_st + k];
    }
    // log_scale: `scale` holds log-std values; the distribution's scale is clamp(exp(.), scale_min, scale_max)
    // (ModelTransition.forward, nn_models/predictions.py) and grad_scale becomes the gradient of the log-std: exp's
    // backward times the clamp's pass-through test (min <= exp <= max, both inclusive as ATen's clamp_backward)
    float e_raw[kMsePerLane];
#pragma unroll
    for (int u = 0; u < kMsePerLane; ++u) {
        e_raw[u] = 1.f;
        if (log_scale) {
            e_raw[u] = expf(sg[u]);
            sg[u] = fminf(fmaxf(e_raw[u], scale_min), scale_max);
        }
    }
#pragma unroll
    for (int u = 0; u < kMsePerLane; ++u) {
        const int i = base + u * kMseThreads + (int)threadIdx.x;
        if (i < N) {
            const float d = xv[u] - mu[u], var = sg[u] * sg[u], ls = logf(sg[u]), inv_s = 1.f / sg[u];
            s_lp += -(d * d) / (2.f * var) - ls - c_lp;
            s_kl += 0.5f * (var + mu[u] * mu[u] - 1.f - logf(var));
            s_ent += c_ent + ls;
            const int64_t at = (int64_t)(i / K) * grad_pitch + (i % K);
            grad_loc[at] = (-(d / var) + w * mu[u]) * inv_n;
            float gs = (-((d * d) / (var * sg[u])) + inv_s + w * (sg[u] - inv_s)) * inv_n;
            if (log_scale) gs = (e_raw[u] >= scale_min && e_raw[u] <= scale_max) ? gs * e_raw[u] : 0.f;
            grad_scale[at] = gs;
        }
    }
    red[0][threadIdx.x] = s_lp, red[1][threadIdx.x] = s_kl, red[2][threadIdx.x] = s_ent;
    __syncthreads();
    for (int h = kMseThreads / 2; h >= 64; h >>= 1) {
        if ((int)threadIdx.x < h) {
#pragma unroll
            for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + h];
        }
        __syncthreads();
    }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v = red[q][threadIdx.x];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (threadIdx.x == 0) partial[blockIdx.x * 3 + q] = v;
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x == 0) {
        float a = 0.f, b2 = 0.f, e = 0.f;
        for (unsigned int g = 0; g < gridDim.x; ++g) a += partial[g * 3], b2 += partial[g * 3 + 1], e += partial[g * 3 + 2];
        out[0] = -(a * inv_n) + w * (b2 * inv_n);
        out[1] = e * inv_n;
        *counter = 0u;
    }
}

inline int stream_grid(int64_t n_vec) {
    int64_t b = (n_vec + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}


// Cosine-sign gating of auxiliary gradients (reference sac_base.py:1619-1631, after the autograd calls):
//   gate_k = clamp(sign(cos(main, aux_k)), min = 0);  grad += gate_k * aux_k, loss by loss.
// The cosine's sign is the sign of the dot product (the norms' product is positive, cosine_similarity clamps it at
// eps from below); a NaN anywhere makes the gate NaN as torch.sign does.  ONE workgroup: the K dots in fixed order
// (lane-strided partial sums, wave shuffles, waves in order), then the gated additions in the reference's order.
constexpr int kGateThreads = 1024;
struct GateArgs {
    const float* main;
    const float* aux[ASAC_GATE_MAX_LOSSES];
    float* grad;
    float* gates_out;
    int n, K;
};

__global__ void __launch_bounds__(kGateThreads) k_cosine_gate_add(const GateArgs a) {
    __shared__ float s_part[ASAC_GATE_MAX_LOSSES][kGateThreads / 64];
    __shared__ float s_gate[ASAC_GATE_MAX_LOSSES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float dot[ASAC_GATE_MAX_LOSSES];
#pragma unroll
    for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k) dot[k] = 0.f;
    // eight elements per lane in flight (the loop is one dependent round trip per iteration otherwise); the partial sums
    // stay lane-strided in index order
    for (int i0 = tid; i0 < a.n; i0 += 8 * kGateThreads) {
        float m[8], v[ASAC_GATE_MAX_LOSSES][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kGateThreads;
            m[u] = i < a.n ? a.main[i] : 0.f;
#pragma unroll
            for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k) v[k][u] = (k < a.K && i < a.n) ? a.aux[k][i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k)
                if (k < a.K) dot[k] += m[u] * v[k][u];
    }
#pragma unroll
    for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k) {
        const float t = wave_sum(dot[k]);
        if (lane == 0) s_part[k][wave] = t;
    }
    __syncthreads();
    if (tid < a.K) {
        float t = 0.f;
        for (int w = 0; w < kGateThreads / 64; ++w) t += s_part[tid][w];
        const float gate = t > 0.f ? 1.f : (t <= 0.f ? 0.f : t);     // (NaN stays NaN)
        s_gate[tid] = gate;
        if (a.gates_out) a.gates_out[tid] = gate;
    }
    __syncthreads();
    float gate[ASAC_GATE_MAX_LOSSES];
#pragma unroll
    for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k) gate[k] = k < a.K ? s_gate[k] : 0.f;
    for (int i0 = tid; i0 < a.n; i0 += 8 * kGateThreads) {
        float g[8], v[ASAC_GATE_MAX_LOSSES][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kGateThreads;
            g[u] = i < a.n ? a.grad[i] : 0.f;
#pragma unroll
            for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k) v[k][u] = (k < a.K && i < a.n) ? a.aux[k][i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kGateThreads;
#pragma unroll
            for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k)
                if (k < a.K) g[u] += gate[k] * v[k][u];
            if (i < a.n) a.grad[i] = g[u];
        }
    }
}

}  // namespace asac

using namespace asac;

// The activation of every fused MLP / convolution kernel, element by element (asac_gelu.h), so that its distance from
// torch.nn.functional.gelu can be measured through the C ABI (tests/test_kernels_gpu.py::test_gelu_against_torch).
__global__ __launch_bounds__(256) void k_gelu_eval(const float* __restrict__ z, float* __restrict__ value,
                                                   float* __restrict__ deriv, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v, d;
        gelu_parts(z[i], v, d);
        value[i] = v;
        deriv[i] = d;
    }
}

extern "C" {

int asac_curiosity_bonus(const float* approx, const float* actual, int64_t actual_stride_b, int64_t actual_stride_t,
                         float* reward, int64_t reward_stride_b, int B, int T, int K, float strength, void* stream) {
    if (B <= 0 || T <= 0 || K <= 0 || !approx || !actual || !reward) return bad_arg("asac_curiosity_bonus");
    ASAC_LAUNCH(k_curiosity_bonus, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, as_stream(stream), approx, actual,
                actual_stride_b, actual_stride_t, reward, reward_stride_b, B, T, K, strength);
    return finish_launch("asac_curiosity_bonus");
}

int64_t asac_masked_mse_workspace(int64_t n) {
    if (n <= 0 || n > ASAC_MASKED_MSE_MAX) return -1;
    return (n + kMseThreads * kMsePerLane - 1) / (kMseThreads * kMsePerLane) + 1;      // workgroup sums + arrival counter
}

int asac_masked_mse(const float* pred, const float* target, int64_t target_stride_b, int64_t target_stride_t,
                    const uint8_t* padding_mask, int64_t mask_stride_b, int B, int T, int K, float* grad_out,
                    float* loss_out, float* workspace, void* stream) {
    const int64_t n = (int64_t)B * T * K;
    if (B <= 0 || T <= 0 || K <= 0 || !pred || !target || !grad_out || !loss_out || !workspace || n > ASAC_MASKED_MSE_MAX)
        return bad_arg("asac_masked_mse");
    const int64_t blocks = asac_masked_mse_workspace(n) - 1;
    ASAC_LAUNCH(k_masked_mse, dim3((unsigned)blocks), dim3(kMseThreads), 0, as_stream(stream), pred, target, target_stride_b,
                target_stride_t, padding_mask, mask_stride_b, B, T, K, grad_out, loss_out, workspace,
                reinterpret_cast<unsigned int*>(workspace + blocks));
    return finish_launch("asac_masked_mse");
}

int64_t asac_mse_mean_grad_workspace(void) { return kMseBigGrid + 1; }

int asac_mse_mean_grad(const float* pred, const float* target, int64_t target_stride_b, int64_t target_stride_t, int B, int T,
                       int K, float grad_scale, float* grad_out, float* loss_out, float* workspace, void* stream) {
    const int64_t n = (int64_t)B * T * K;
    if (B <= 0 || T <= 0 || K <= 0 || (K & 3) || !pred || !target || !grad_out || !loss_out || !workspace ||
        n / 4 >= 0x7fffffffll || (target_stride_b & 3) || (target_stride_t & 3) ||
        ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad_out)) & 15))
        return bad_arg("asac_mse_mean_grad");
    const unsigned int n4 = (unsigned int)(n / 4);
    const int64_t want = ((int64_t)n4 + kMseThreads * 4 - 1) / (kMseThreads * 4);
    const unsigned blocks = (unsigned)(want < kMseBigGrid ? want : kMseBigGrid);
    ASAC_LAUNCH(k_mse_big, dim3(blocks), dim3(kMseThreads), 0, as_stream(stream), pred, target, target_stride_b,
                target_stride_t, T, K / 4, n4, 1.f / (float)n, grad_scale, grad_out, loss_out, workspace,
                reinterpret_cast<unsigned int*>(workspace + kMseBigGrid));
    return finish_launch("asac_mse_mean_grad");
}

int64_t asac_normal_nll_kl_workspace(int64_t n) {
    if (n <= 0 || n > ASAC_MASKED_MSE_MAX) return -1;
    return 3 * ((n + kMseThreads * kMsePerLane - 1) / (kMseThreads * kMsePerLane)) + 1;
}

int asac_normal_nll_kl(const float* loc, int64_t loc_stride_b, int64_t loc_stride_t, const float* scale,
                       int64_t scale_stride_b, int64_t scale_stride_t, const float* target, int64_t target_stride_b,
                       int64_t target_stride_t, int B, int T, int K, float kl_weight, float* grad_loc, float* grad_scale,
                       float* loss_entropy_out, float* workspace, void* stream) {
    const int64_t n = (int64_t)B * T * K;
    if (B <= 0 || T <= 0 || K <= 0 || !loc || !scale || !target || !grad_loc || !grad_scale || !loss_entropy_out ||
        !workspace || n > ASAC_MASKED_MSE_MAX)
        return bad_arg("asac_normal_nll_kl");
    const int64_t blocks = (asac_normal_nll_kl_workspace(n) - 1) / 3;
    ASAC_LAUNCH(k_normal_nll_kl, dim3((unsigned)blocks), dim3(kMseThreads), 0, as_stream(stream), loc, loc_stride_b,
                loc_stride_t, scale, scale_stride_b, scale_stride_t, target, target_stride_b, target_stride_t, B, T, K,
                kl_weight, grad_loc, grad_scale, loss_entropy_out, workspace,
                reinterpret_cast<unsigned int*>(workspace + 3 * blocks), 0, 0.f, 0.f, (int64_t)K);
    return finish_launch("asac_normal_nll_kl");
}

int asac_normal_nll_kl_logstd(const float* mean_logstd, int64_t stride_b, int64_t stride_t, float scale_min, float scale_max,
                              const float* target, int64_t target_stride_b, int64_t target_stride_t, int B, int T, int K,
                              float kl_weight, float* grad_mean_logstd, float* loss_entropy_out, float* workspace,
                              void* stream) {
    const int64_t n = (int64_t)B * T * K;
    if (B <= 0 || T <= 0 || K <= 0 || !mean_logstd || !target || !grad_mean_logstd || !loss_entropy_out || !workspace ||
        n > ASAC_MASKED_MSE_MAX || !(scale_min > 0.f) || !(scale_max >= scale_min))
        return bad_arg("asac_normal_nll_kl_logstd");
    const int64_t blocks = (asac_normal_nll_kl_workspace(n) - 1) / 3;
    ASAC_LAUNCH(k_normal_nll_kl, dim3((unsigned)blocks), dim3(kMseThreads), 0, as_stream(stream), mean_logstd, stride_b,
                stride_t, mean_logstd + K, stride_b, stride_t, target, target_stride_b, target_stride_t, B, T, K, kl_weight,
                grad_mean_logstd, grad_mean_logstd + K, loss_entropy_out, workspace,
                reinterpret_cast<unsigned int*>(workspace + 3 * blocks), 1, scale_min, scale_max, (int64_t)2 * K);
    return finish_launch("asac_normal_nll_kl_logstd");
}

int asac_cosine_gate_add(const float* main, const float* const* aux_host, int K, int64_t n, float* grad, float* gates_out,
                         void* stream) {
    if (!main || !aux_host || !grad || K <= 0 || K > ASAC_GATE_MAX_LOSSES || n <= 0 || n > ASAC_GATE_MAX_N)
        return bad_arg("asac_cosine_gate_add");
    GateArgs a;
    a.main = main, a.grad = grad, a.gates_out = gates_out, a.n = (int)n, a.K = K;
    for (int k = 0; k < ASAC_GATE_MAX_LOSSES; ++k) a.aux[k] = k < K ? aux_host[k] : nullptr;
    for (int k = 0; k < K; ++k)
        if (!a.aux[k]) return bad_arg("asac_cosine_gate_add");
    ASAC_LAUNCH(k_cosine_gate_add, dim3(1), dim3(kGateThreads), 0, as_stream(stream), a);
    return finish_launch("asac_cosine_gate_add");
}


int asac_gelu_eval(const float* z, float* value, float* deriv, int64_t n, void* stream) {
    if (n <= 0 || !z || !value || !deriv) return bad_arg("asac_gelu_eval");
    ASAC_LAUNCH(k_gelu_eval, dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), z, value, deriv, n);
    return finish_launch("asac_gelu_eval");
}

int asac_polyak(float* target, const float* source, int64_t n, float tau, void* stream) {
    if (n <= 0) return bad_arg("asac_polyak");
    const float one_m_tau = (float)(1.0 - (double)tau);   // python: (1. - tau) in double, cast by ATen
    ASAC_LAUNCH(k_polyak, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, as_stream(stream), target,
                       source, n, one_m_tau, tau);
    return finish_launch("asac_polyak");
}

int asac_alpha_adam_step(const float* logp, int B, float target, int slot, float* param, float* grad,
                         float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1, float beta2,
                         float eps, int64_t* steps_done, int advance_counter, void* stream) {
    if (B <= 0 || n <= 0 || slot < 0 || slot >= n || !logp || !steps_done) return bad_arg("asac_alpha_adam_step");
    // under the measurement repeat knob only the last repetition advances the counter
    for (int rep = 0; rep < g_launch_repeat; ++rep)
        hipLaunchKernelGGL(k_alpha_adam, dim3(1), dim3(256), 0, as_stream(stream),
                           AlphaAdamArgs{logp, B, target, slot, param, grad, exp_avg, exp_avg_sq, n,
                                         adam_scalars(lr, beta1, beta2, eps), steps_done,
                                         (advance_counter && rep == g_launch_repeat - 1) ? 1 : 0});
    return finish_launch("asac_alpha_adam_step");
}

int asac_adam_step_partials(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                            float beta2, float eps, const int64_t* steps_done, const float* workspace,
                            int64_t tiles, int E, int64_t member_stride, int64_t used, int accumulate,
                            float* loss_out, int64_t loss_rows, void* stream) {
    if (E <= 0 || member_stride <= 0 || tiles <= 0 || used <= 0 || used > member_stride || !steps_done || !workspace)
        return bad_arg("asac_adam_step_partials");
    const int64_t n = (int64_t)E * member_stride;
    const float* loss_partial = loss_out ? workspace + tiles * E * member_stride : nullptr;
    ASAC_LAUNCH(k_adam_partials, dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), param, grad, exp_avg,
                exp_avg_sq, n, adam_scalars(lr, beta1, beta2, eps), steps_done, workspace, (int)tiles, E,
                member_stride, used, accumulate, loss_partial, loss_out,
                loss_rows > 0 ? 1.f / (float)loss_rows : 0.f);
    return finish_launch("asac_adam_step_partials");
}

int asac_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, const int64_t* steps_done,
                   void* stream) {
    if (n <= 0 || !steps_done) return bad_arg("asac_adam_step");
    const AdamScalars c = adam_scalars(lr, beta1, beta2, eps);
    ASAC_LAUNCH(k_adam, dim3(stream_grid(n / 4 + 1)), dim3(256), 0, as_stream(stream), param, grad,
                       exp_avg, exp_avg_sq, n, c, steps_done);
    return finish_launch("asac_adam_step");
}

}  // extern "C"
