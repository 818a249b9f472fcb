// Sidecar jobs: small launches of the train step that do not sit on its critical path ride as EXTRA WORKGROUPS of a
// launch that does (asac_squash_multi, asac_mlp_forward_multi) instead of paying a dependent kernel boundary of their
// own (1.8 us for an empty kernel inside a replayed graph on MI355X, tools/launchprobe.hip; 3.7-4.9 us for these with
// their own argument / first-load latency).  A sidecar only needs to be ordered after the launches before its host and
// before the launches after it, and must not touch anything the host launch itself reads or writes.
//   ALPHA_ADAM      the temperature step (k_alpha_adam), one workgroup
//   SCATTER_ELECT   pass 1 of asac_scatter_rows_if_id_match (last row-major writer per ring slot)
//   SCATTER_WRITE   pass 2 (the elected rows copy their payload and hand the slot back); one lane per row for rows
//                   of <= 32 bytes, one wave per row above
#pragma once
#include "asac_common.h"
#include "asac_gather.h"

#include <cmath>

namespace asac {

// torch.optim.Adam (single-tensor form, torch/optim/adam.py):
//   m.lerp_(g, 1-b1);  v.mul_(b2).addcmul_(g, g, value=1-b2)
//   denom = v.sqrt() / sqrt(1-b2^t) + eps;  p.addcdiv_(m, denom, value=-(lr / (1-b1^t)))
struct AdamScalars {
    float w1, b2, one_m_b2, eps;
    double lr, b1, b2d;
};

inline AdamScalars adam_scalars(float lr, float beta1, float beta2, float eps) {
    AdamScalars c;
    c.w1 = (float)(1.0 - (double)beta1);
    c.b2 = beta2;
    c.one_m_b2 = (float)(1.0 - (double)beta2);
    c.eps = eps;
    c.lr = (double)lr;
    c.b1 = (double)beta1;
    c.b2d = (double)beta2;
    return c;
}

// The step's two bias corrections, lr / (1 - b1^t) and sqrt(1 - b2^t) with t = done + 1, in double precision and rounded
// to float as torch's single-tensor Adam has them.  b^t by binary powering (t is an integer): <= 2 log2 t dependent
// multiplies per base, the two bases interleaved — a library pow(double, double) is some 150 instructions each and was
// the long pole of the one-workgroup temperature step (tools/td_phases.py: 2.55 -> 2.02 us, the same as no pow at
// all).  The powering's rounding (<= log2 t ulp of double) disappears in the cast to float.
__device__ __forceinline__ void adam_bias_terms(const AdamScalars& c, int64_t done, float* step_size, float* bc2_sqrt) {
    double x1 = c.b1, x2 = c.b2d, r1 = 1.0, r2 = 1.0;
    for (int64_t e = done + 1; e > 0; e >>= 1) {
        if (e & 1) r1 *= x1, r2 *= x2;
        x1 *= x1, x2 *= x2;
    }
    *step_size = (float)(c.lr / (1.0 - r1));
    *bc2_sqrt = (float)sqrt(1.0 - r2);
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamScalars& c,
                                      float step_size, float bc2_sqrt) {
    m = m + c.w1 * (g - m);                         // lerp, weight < 0.5
    v = v * c.b2 + c.one_m_b2 * (g * g);            // addcmul: v + value * (g*g)
    const float denom = sqrtf(v) / bc2_sqrt + c.eps;
    p = p + (-step_size) * (m / denom);             // addcdiv: p + value * (m / denom)
}

struct AlphaAdamArgs {
    const float* logp;
    int32_t B;
    float target;
    int32_t slot;
    float *param, *grad, *exp_avg, *exp_avg_sq;
    int32_t n;
    AdamScalars c;
    int64_t* steps_done;
    int32_t advance;
};

// Temperature step (reference `_train_alpha`, sac_base.py:1913-1949, continuous head):
//   dL/dlog_alpha = mean_b(-logp_b) - target   into grad[slot], then Adam over the n (= 2) temperature
// parameters [log_d_alpha, log_c_alpha] exactly as k_adam would.  Called by EVERY thread of a workgroup of >= 256
// threads; the first 256 do the work in a fixed order (the result does not depend on the host kernel's size).
// -> the new value of parameter `slot` (in every thread).
// red[0] <- the sum of red[0..256) in the order of the halving tree (stride 128, 64, ... 1: red[i] += red[i + stride]).
// The strides below a wave's width stay inside wave 0: lane exchanges, no barrier rounds — the same additions.
// Called by every thread of the workgroup, red[] written and a barrier passed; returns after a barrier.
__device__ __forceinline__ void alpha_reduce_256(float* red) {
    const int tid = threadIdx.x;
    if (tid < 128) red[tid] += red[tid + 128];
    __syncthreads();
    if (tid < 64) {
        float v = red[tid] + red[tid + 64];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
        if (tid == 0) red[0] = v;
    }
    __syncthreads();
}

__device__ __forceinline__ float alpha_adam_block(const AlphaAdamArgs& a, float* red /* LDS, 256 floats */) {
    const int tid = threadIdx.x;
    // everything the step reads is requested before the reduction's barriers (one round of loads, not three); every
    // thread also follows the parameter `slot` in registers: the new temperature is RETURNED, nobody re-reads it
    const int64_t done = *a.steps_done;
    float part = 0.f;
    if (tid < 256)
        for (int b = tid; b < a.B; b += 256) part += -a.logp[b] - a.target;
    const bool mine = tid < 256 && tid < a.n;
    float p = 0.f, g = 0.f, m = 0.f, v = 0.f;
    if (mine) p = a.param[tid], g = a.grad[tid], m = a.exp_avg[tid], v = a.exp_avg_sq[tid];
    float ps = a.param[a.slot], ms = a.exp_avg[a.slot], vs = a.exp_avg_sq[a.slot];
    // the bias corrections (double precision) by ONE wave — the second, idle once
    // the reduction's first stage is through — while the first finishes the sum; a 1024-thread host workgroup would
    // otherwise run them in all of its 16 waves, four deep on every SIMD
    float step_size = 0.f, bc2_sqrt = 0.f;
    if ((tid >> 6) == 1) adam_bias_terms(a.c, done, &step_size, &bc2_sqrt);
    if (tid < 256) red[tid] = part;
    __syncthreads();
    if (tid < 128) red[tid] += red[tid + 128];
    __syncthreads();
    if (tid < 64) {                  // (the halving tree's last six strides stay inside wave 0: alpha_reduce_256)
        float s = red[tid] + red[tid + 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (tid == 0) red[0] = s;
    }
    if (tid == 64) red[192] = step_size, red[193] = bc2_sqrt;
    __syncthreads();
    const float g_slot = red[0] / (float)a.B;
    step_size = red[192], bc2_sqrt = red[193];
    if (tid == 0) a.grad[a.slot] = g_slot;
    if (a.advance && tid == 0) *a.steps_done = done + 1;
    if (mine) {
        adam1(p, tid == a.slot ? g_slot : g, m, v, a.c, step_size, bc2_sqrt);
        a.param[tid] = p, a.exp_avg[tid] = m, a.exp_avg_sq[tid] = v;
    }
    if (tid < 256)
        for (int i = tid + 256; i < a.n; i += 256)
            adam1(a.param[i], i == a.slot ? g_slot : a.grad[i], a.exp_avg[i], a.exp_avg_sq[i], a.c, step_size, bc2_sqrt);
    adam1(ps, g_slot, ms, vs, a.c, step_size, bc2_sqrt);
    return ps;
}

// The value the temperature parameter `slot` WILL have after alpha_adam_block(a), computed without writing anything:
// a launch that must already see the updated temperature (the TD error's return, sac_base.py:2182-2245 after 1913-1949)
// evaluates this per workgroup while the step itself rides as a sidecar of a LATER launch — nothing waits for a
// one-workgroup kernel.  Same reduction order and arithmetic as alpha_adam_block: bit-identical.  Called by every thread
// of a 256-thread workgroup.
__device__ __forceinline__ float alpha_adam_preview(const AlphaAdamArgs& a, float* red /* LDS, 256 floats */) {
    const int tid = threadIdx.x;
    float part = 0.f;
    for (int b = tid; b < a.B; b += 256) part += -a.logp[b] - a.target;
    const int64_t done = *a.steps_done;
    float p = a.param[a.slot], m = a.exp_avg[a.slot], v = a.exp_avg_sq[a.slot];
    float step_size, bc2_sqrt;
    adam_bias_terms(a.c, done, &step_size, &bc2_sqrt);                      // (overlaps the loads above)
    red[tid] = part;
    __syncthreads();
    alpha_reduce_256(red);
    const float g_slot = red[0] / (float)a.B;
    __syncthreads();                                    // (the caller reuses `red`)
    adam1(p, g_slot, m, v, a.c, step_size, bc2_sqrt);
    return p;
}

// ------------------------------------------------------------------------------------------------
// K7: two passes over the [batch, count] targets.  Pass 1 elects, per ring slot, the LAST row (in
// row-major order) that is unpadded and whose id still lives in the slot; pass 2 lets only the
// elected row copy its payload and hand the slot back (-1).
// ------------------------------------------------------------------------------------------------
struct ScatterArgs {
    uint8_t* ring;
    int32_t row_bytes, capacity;
    const int64_t* ids;
    int32_t batch, first_off, count;
    const int64_t* slot_ids;
    const uint8_t* padding_mask;
    int32_t mask_sample_stride;
    const uint8_t* rows;
    int64_t rows_sample_stride, rows_row_stride;
    int32_t* winner;
};

__device__ __forceinline__ bool scatter_target(const ScatterArgs& a, int flat, int* slot_out) {
    const int s = flat / a.count;
    const int j = flat - s * a.count;
    if (a.padding_mask && a.padding_mask[(int64_t)s * a.mask_sample_stride + j]) return false;
    // a negative sample id marks a sample another replay shard owns (sharded "parity" sampling, algorithm/parallel.py):
    // none of its window rows is this shard's to write, whatever id + offset comes to
    if (a.ids[s] < 0) return false;
    const int64_t tid = a.ids[s] + a.first_off + j;
    const int slot = ring_slot(tid, a.capacity);
    *slot_out = slot;
    return a.slot_ids[slot] == tid;
}

__device__ __forceinline__ void scatter_elect_row(const ScatterArgs& a, int flat) {
    if (flat >= a.batch * a.count) return;
    int slot;
    if (scatter_target(a, flat, &slot)) atomicMax(&a.winner[slot], flat);
}

// the elected row `flat` copies its payload with `lanes` lanes (lane index `lane`), then hands the slot back
__device__ __forceinline__ void scatter_write_row(const ScatterArgs& a, int flat, int lane, int lanes) {
    if (flat >= a.batch * a.count) return;
    int slot;
    if (!scatter_target(a, flat, &slot)) return;
    if (__hip_atomic_load(&a.winner[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != flat) return;
    const int s = flat / a.count;
    const int j = flat - s * a.count;
    const uint8_t* src = a.rows + (int64_t)s * a.rows_sample_stride + (int64_t)j * a.rows_row_stride;
    uint8_t* dst = a.ring + (int64_t)slot * a.row_bytes;
    if (((a.row_bytes | (int)(reinterpret_cast<uintptr_t>(src) & 3) |
          (int)(reinterpret_cast<uintptr_t>(dst) & 3)) & 3) == 0) {
        for (int w = lane; w < a.row_bytes / 4; w += lanes)
            reinterpret_cast<uint32_t*>(dst)[w] = reinterpret_cast<const uint32_t*>(src)[w];
    } else {
        for (int w = lane; w < a.row_bytes; w += lanes) dst[w] = src[w];
    }
    if (lane == 0) __hip_atomic_store(&a.winner[slot], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
struct SidecarDev {
    int32_t kind, first_block;        // first workgroup of this job among the launch's sidecar workgroups
    AlphaAdamArgs alpha;
    ScatterArgs scatter;
    const void* gather;               // WINDOW_GATHER: a GatherLaunch<ASAC_MAX_GATHER_KEYS> in device memory
};
// (kernel arguments are copied per launch: hosts that usually carry no or one sidecar have variants taking a shorter
// list — SidecarsT<1> is 250 bytes, the full list 1 KB)
template <int NSC>
struct SidecarsT {
    SidecarDev j[NSC];
    int32_t n, blocks;                // jobs, sidecar workgroups in all
};
using SidecarsDev = SidecarsT<ASAC_MAX_SIDECARS>;

template <int NSC>
inline SidecarsT<NSC> sidecars_first(const SidecarsDev& full) {
    SidecarsT<NSC> out{};
    for (int k = 0; k < NSC && k < full.n; ++k) out.j[k] = full.j[k];
    out.n = full.n < NSC ? full.n : NSC;
    out.blocks = full.blocks;
    return out;
}

constexpr int kSidecarRowsPerWg = 256;        // SCATTER_ELECT, and SCATTER_WRITE of short rows: one lane per row
inline int scatter_write_rows_per_wg(int row_bytes) { return row_bytes <= 32 ? kSidecarRowsPerWg : 4; }

// host: device-side job list + workgroup count from the C structs; 0 = ok
inline int sidecars_prepare(const asac_sidecar_t* jobs, int n, SidecarsDev& out) {
    out.n = 0;
    out.blocks = 0;
    if (n < 0 || n > ASAC_MAX_SIDECARS || (n > 0 && !jobs)) return 1;
    for (int k = 0; k < n; ++k) {
        const asac_sidecar_t& h = jobs[k];
        SidecarDev& d = out.j[k];
        d.kind = h.kind;
        d.first_block = out.blocks;
        if (h.kind == ASAC_SIDECAR_ALPHA_ADAM) {
            if (h.B <= 0 || h.n <= 0 || h.slot < 0 || h.slot >= h.n || !h.logp || !h.steps_done) return 1;
            d.alpha = AlphaAdamArgs{h.logp, h.B, h.target, h.slot, h.param, h.grad, h.exp_avg, h.exp_avg_sq, h.n,
                                    adam_scalars(h.lr, h.beta1, h.beta2, h.eps), h.steps_done, h.advance_counter};
            out.blocks += 1;
        } else if (h.kind == ASAC_SIDECAR_SCATTER_ELECT || h.kind == ASAC_SIDECAR_SCATTER_WRITE) {
            if (h.row_bytes <= 0 || h.capacity <= 0 || h.batch <= 0 || h.count <= 0 || !h.winner || !h.slot_ids) return 1;
            d.scatter = ScatterArgs{static_cast<uint8_t*>(h.ring), h.row_bytes, h.capacity, h.ids, h.batch, h.first_off,
                                    h.count, h.slot_ids, h.padding_mask, h.mask_sample_stride,
                                    static_cast<const uint8_t*>(h.rows), h.rows_sample_stride_bytes,
                                    h.rows_row_stride_bytes, h.winner};
            const int total = h.batch * h.count;
            const int per = h.kind == ASAC_SIDECAR_SCATTER_ELECT ? kSidecarRowsPerWg : scatter_write_rows_per_wg(h.row_bytes);
            out.blocks += (total + per - 1) / per;
        } else if (h.kind == ASAC_SIDECAR_WINDOW_GATHER) {
            if (!h.gather_plan || h.gather_blocks <= 0) return 1;
            d.gather = h.gather_plan;
            out.blocks += h.gather_blocks;
        } else {
            return 1;
        }
    }
    out.n = n;
    return 0;
}

// device: workgroup `block` (0-based among the sidecar workgroups) of a host kernel with >= 256 threads
// (GATHER: hosts that accept ASAC_SIDECAR_WINDOW_GATHER jobs — the others do not carry the gather's code)
template <int NSC, bool GATHER = false>
__device__ __forceinline__ void sidecar_run(const SidecarsT<NSC>& sc, int block, float* lds256) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < NSC; ++q)
        if (q < sc.n && block >= sc.j[q].first_block) k = q;
    const SidecarDev& job = sc.j[k];
    const int local = block - job.first_block;
    if (job.kind == ASAC_SIDECAR_ALPHA_ADAM) {
        alpha_adam_block(job.alpha, lds256);
    } else if (GATHER && job.kind == ASAC_SIDECAR_WINDOW_GATHER) {
        if (threadIdx.x < kGatherBlock)
            gather_block<ASAC_MAX_GATHER_KEYS, 1>(*static_cast<const GatherLaunch<ASAC_MAX_GATHER_KEYS>*>(job.gather),
                                                  (unsigned)local);
    } else if (threadIdx.x < 256) {
        if (job.kind == ASAC_SIDECAR_SCATTER_ELECT) {
            scatter_elect_row(job.scatter, local * kSidecarRowsPerWg + (int)threadIdx.x);
        } else if (job.scatter.row_bytes <= 32) {
            scatter_write_row(job.scatter, local * kSidecarRowsPerWg + (int)threadIdx.x, 0, 1);
        } else {
            scatter_write_row(job.scatter, local * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), kWave);
        }
    }
}

}  // namespace asac
