// Fused residual-MLP forward / backward for gfx950 (MFMA f32): the stock Q and policy networks of
// the reference (`LinearLayers` stacks: ResBlock = GELU(Linear(x)) (+x), then Linear head(s);
// algorithm/nn_models/layers/linear_layers.py:24-119, q.py:34-91, policy.py:116-174) evaluated in
// ONE launch per (ensemble of) network(s) instead of ~12 (forward) / ~25 (backward) eager kernels.
// C ABI in include/asac_hip.h.
//
// Shape of the problem: N <= a few thousand rows, widths <= 64 — 10..30 MFLOP per pass.  The pass is
// launch/latency-bound, not FLOP-bound, so the design goal is "one launch, shortest serial chain":
//   * a workgroup of 8 waves owns a 32-row tile of one ensemble member.  Every layer's weights are
//     staged into LDS ONCE at kernel entry (16-byte loads, all issued before the first store: one L2
//     round trip), activations ping-pong between two LDS tiles ([32][66] f32; pitch 66 makes the
//     16x4 MFMA fragment reads bank-conflict-free), so a layer costs one barrier
//   * a layer is C[32 x 64] = X[32 x K] * W^T with v_mfma_f32_16x16x4_f32 (exact f32: bitwise an
//     fmaf chain): wave w owns the 16x16 output tile (row tile w&1, column tile w>>1), two
//     independent accumulators over even / odd k-steps keep the MFMA pipe at its issue rate
//   * GELU (erf form) and its derivative come from one exp (Abramowitz-Stegun erf, |err| <= 1.5e-7)
//   * backward recomputes the forward (pre-activations stay in registers in the MFMA C layout,
//     block inputs in LDS), then walks the layers in reverse: delta = g * gelu'(z); dX = delta * W;
//     dW = delta^T * X_prev (MFMA with the 32 rows as the reduction dim); per-tile partial parameter
//     gradients go to a scratch slab and are summed in fixed tile order by a second kernel
//     (deterministic, no float atomics)
#include "asac_common.h"
#include "asac_gelu.h"
#include "asac_sidecar.h"
#include "asac_squash.h"
#include "asac_vtrace.h"

#include <cmath>

namespace asac {

// Rows per workgroup tile: TM = 32 (8 waves: 2 row tiles x 4 column tiles) or TM = 16 (4 waves, one per column tile).
// The f32 MFMA rate of ONE CU is what a layer costs (32 x 64 x 64: 1024 cycles, 0.43 us at 2.4 GHz), so a pass that
// has few rows (the batch-256 launches of the train step) takes 16-row tiles and twice the workgroups — half the MFMA
// time per layer on twice as many of the 256 CUs — while long row sets keep 32-row tiles (one resident round).
constexpr int kP = 66;           // LDS pitch (floats)
constexpr int kMaxW = 64;        // max layer width / input width
constexpr int kMaxB = ASAC_MLP_MAX_BLOCKS;
constexpr int kHeadPad = 16;     // head output columns are padded to one MFMA tile
// waves of k_policy_step (A/B builds: -DASAC_PS_WAVES=8): the compute phases have work for eight (critics: 2 members x 4
// column tiles) resp. four waves; the others share the staging of the three networks and the weight-gradient tiles
#ifndef ASAC_PS_WAVES
#define ASAC_PS_WAVES 16
#endif
// ... and of k_pi_sample_q's fused job (its plain forward jobs and sidecars run on four)
#ifndef ASAC_PIQ_WAVES
#define ASAC_PIQ_WAVES 8      // (sixteen: 120 VGPRs, no spills, and 0.9 % slower on cfg2 — A/B on one box, round 3)
#endif
constexpr int kPiQThreads = 64 * ASAC_PIQ_WAVES;
template <int TM> constexpr int threads_of() { return TM * 16; }    // one thread per (row, 4-column group)

// 16-row tiles while they still fit one resident round of workgroups
inline int mlp_tile_rows(int64_t N, int E) { return ((N + 15) / 16) * (int64_t)(E > 0 ? E : 1) <= 256 ? 16 : 32; }

using f32x4 = __attribute__((ext_vector_type(4))) float;

// phase time stamps of workgroup (0, 0) for tools/mlp_phases.hip (100 MHz wall clock); compiled out of the library
#ifdef ASAC_MLP_STAMPS
__device__ unsigned long long g_mlp_stamps[32];
#define MLP_STAMP(i)                                                                             \
    do {                                                                                         \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_mlp_stamps[i] = wall_clock64(); \
    } while (0)
#else
#define MLP_STAMP(i)
#endif

__device__ __forceinline__ int round4(int v) { return (v + 3) & ~3; }

// stage a row-major [rows][cols] global matrix into LDS [64][kP], zero padded to 64 x 64
template <int THREADS>
__device__ __forceinline__ void stage_matrix(float* dst, const float* __restrict__ src, int rows, int cols) {
    if (rows == kMaxW && cols == kMaxW && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        constexpr int U = 1024 / THREADS;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s4[threadIdx.x + u * THREADS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = threadIdx.x + u * THREADS;   // float4 index: row q/16, col 4*(q%16)
            float2* d = reinterpret_cast<float2*>(dst + (q >> 4) * kP + ((q & 15) << 2));
            d[0] = make_float2(v[u].x, v[u].y);
            d[1] = make_float2(v[u].z, v[u].w);
        }
        return;
    }
    constexpr int U = 4096 / THREADS;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;
        const int r = i >> 6, c = i & 63;
        v[u] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;
        dst[(i >> 6) * kP + (i & 63)] = v[u];
    }
}

// columns [c0, c0 + ncols) of a row-major [rows][cols] global matrix into LDS [64][kP], zero padded to 64 x 64
// (the two halves of a first layer wider than 64 inputs)
template <int THREADS>
__device__ __forceinline__ void stage_matrix_part(float* dst, const float* __restrict__ src, int rows, int cols, int c0,
                                                  int ncols) {
    constexpr int U = 4096 / THREADS;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;
        const int r = i >> 6, c = i & 63;
        v[u] = (r < rows && c < ncols) ? src[r * cols + c0 + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;
        dst[(i >> 6) * kP + (i & 63)] = v[u];
    }
}

// the (up to two) head Linear layers as one zero-padded [16][64] matrix + bias vector
template <int THREADS>
__device__ __forceinline__ void stage_heads(const asac_mlp_desc_t& d, const float* __restrict__ P, int K,
                                            float* head, float* head_bias) {
    constexpr int U = 1024 / THREADS;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;       // 1024 = 16 x 64
        const int o = i >> 6, c = i & 63;
        float x = 0.f;
        if (c < K) {
            if (o < d.head_cols[0]) x = P[d.head_w_off[0] + o * K + c];
            else if (o < d.head_cols[0] + d.head_cols[1]) x = P[d.head_w_off[1] + (o - d.head_cols[0]) * K + c];
        }
        v[u] = x;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;
        head[(i >> 6) * kP + (i & 63)] = v[u];
    }
    if (head_bias && threadIdx.x < kHeadPad) {
        const int o = threadIdx.x;
        float x = 0.f;
        if (o < d.head_cols[0]) x = P[d.head_b_off[0] + o];
        else if (o < d.head_cols[0] + d.head_cols[1]) x = P[d.head_b_off[1] + o - d.head_cols[0]];
        head_bias[o] = x;
    }
}

// A whole network (every block's weight and bias, the heads) global -> registers -> LDS with ALL global loads issued
// before the first LDS store: one memory round trip for the kernel's staging phase (each stage_* call on its own
// waits for its loads before it stores — five dependent round trips of ~0.6 us at the head of every launch).
template <int THREADS>
struct StagedNet {
    float w[kMaxB][4096 / THREADS];
    float head[1024 / THREADS];
    float bias[kMaxB];
    float head_bias;
};

__device__ __forceinline__ bool fast_tile(const float* src, int rows, int cols) {
    return rows == kMaxW && cols == kMaxW && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
}

// `skip_first`: block 0 is staged by the caller (a first layer wider than 64 inputs)
template <int THREADS>
__device__ __forceinline__ void net_fetch(const asac_mlp_desc_t& d, const float* __restrict__ P, int K0, bool skip_first,
                                          StagedNet<THREADS>& r) {
    constexpr int U = 4096 / THREADS;
    int K = K0;
#pragma unroll
    for (int l = 0; l < kMaxB; ++l) {
        if (l < d.n_blocks) {
            const int W = d.width[l];
            const float* src = P + d.w_off[l];
            if (!(l == 0 && skip_first)) {
                if (fast_tile(src, W, K)) {
                    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
                    for (int u = 0; u < U / 4; ++u) {
                        const float4 t = s4[threadIdx.x + u * THREADS];
                        r.w[l][4 * u] = t.x, r.w[l][4 * u + 1] = t.y, r.w[l][4 * u + 2] = t.z, r.w[l][4 * u + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = threadIdx.x + u * THREADS;
                        const int rr = i >> 6, c = i & 63;
                        r.w[l][u] = (rr < W && c < K) ? src[rr * K + c] : 0.f;
                    }
                }
            }
            r.bias[l] = ((int)threadIdx.x < W && threadIdx.x < kMaxW) ? P[d.b_off[l] + threadIdx.x] : 0.f;
            K = W;
        }
    }
#pragma unroll
    for (int u = 0; u < 1024 / THREADS; ++u) {
        const int i = threadIdx.x + u * THREADS;       // 1024 = 16 x 64
        const int o = i >> 6, c = i & 63;
        float x = 0.f;
        if (c < K) {
            if (o < d.head_cols[0]) x = P[d.head_w_off[0] + o * K + c];
            else if (o < d.head_cols[0] + d.head_cols[1]) x = P[d.head_w_off[1] + (o - d.head_cols[0]) * K + c];
        }
        r.head[u] = x;
    }
    r.head_bias = 0.f;
    if (threadIdx.x < kHeadPad) {
        const int o = threadIdx.x;
        if (o < d.head_cols[0]) r.head_bias = P[d.head_b_off[0] + o];
        else if (o < d.head_cols[0] + d.head_cols[1]) r.head_bias = P[d.head_b_off[1] + o - d.head_cols[0]];
    }
}

template <int THREADS, typename LDS>
__device__ __forceinline__ void net_put(const asac_mlp_desc_t& d, const float* __restrict__ P, int K0, bool skip_first,
                                        const StagedNet<THREADS>& r, LDS& L) {
    constexpr int U = 4096 / THREADS;
    int K = K0;
#pragma unroll
    for (int l = 0; l < kMaxB; ++l) {
        if (l < d.n_blocks) {
            const int W = d.width[l];
            float* dst = L.w[l];
            if (!(l == 0 && skip_first)) {
                if (fast_tile(P + d.w_off[l], W, K)) {
#pragma unroll
                    for (int u = 0; u < U / 4; ++u) {
                        const int q = threadIdx.x + u * THREADS;   // float4 index: row q/16, col 4*(q%16)
                        float2* o2 = reinterpret_cast<float2*>(dst + (q >> 4) * kP + ((q & 15) << 2));
                        o2[0] = make_float2(r.w[l][4 * u], r.w[l][4 * u + 1]);
                        o2[1] = make_float2(r.w[l][4 * u + 2], r.w[l][4 * u + 3]);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = threadIdx.x + u * THREADS;
                        dst[(i >> 6) * kP + (i & 63)] = r.w[l][u];
                    }
                }
            }
            if (threadIdx.x < kMaxW) L.bias[l][threadIdx.x] = r.bias[l];
            K = W;
        }
    }
#pragma unroll
    for (int u = 0; u < 1024 / THREADS; ++u) {
        const int i = threadIdx.x + u * THREADS;
        L.head[(i >> 6) * kP + (i & 63)] = r.head[u];
    }
    if (threadIdx.x < kHeadPad) L.head_bias[threadIdx.x] = r.head_bias;
}

// D[16 x 16] += A[rt*16.., 0..K) * B^T, B = lds [out col][k]: this wave's tile (rt, ct)
__device__ __forceinline__ f32x4 gemm_tile(const float* __restrict__ A, const float* __restrict__ B, int K4,
                                           int rt, int ct) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const float* a_ptr = A + (rt * 16 + lr) * kP + lk;
    const float* b_ptr = B + (ct * 16 + lr) * kP + lk;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (K4 == 64) {
        float av[16], bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            av[i] = a_ptr[4 * i];
            bv[i] = b_ptr[4 * i];
        }
        // every LDS read in flight before the first MFMA (left alone the scheduler pairs each MFMA step with its
        // own reads: eight dependent LDS round trips per layer)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1], bv[i + 1], acc1, 0, 0, 0);
        }
    } else {
        for (int k0 = 0; k0 < K4; k0 += 4)
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ptr[k0], b_ptr[k0], acc0, 0, 0, 0);
    }
    return acc0 + acc1;
}

// D[16 x 16] += A[rows][0..J) * Wm, Wm = lds [j][k] used as B[kk = j][jj = k]: dX = delta * W
__device__ __forceinline__ f32x4 gemm_tile_nt(const float* __restrict__ A, const float* __restrict__ Wm, int J4,
                                              int rt, int ct) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const float* a_ptr = A + (rt * 16 + lr) * kP + lk;
    const float* b_ptr = Wm + lk * kP + ct * 16 + lr;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (J4 == 64) {
        float av[16], bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            av[i] = a_ptr[4 * i];
            bv[i] = b_ptr[4 * i * kP];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1], bv[i + 1], acc1, 0, 0, 0);
        }
    } else {
        for (int j0 = 0; j0 < J4; j0 += 4)
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ptr[j0], b_ptr[j0 * kP], acc0, 0, 0, 0);
    }
    return acc0 + acc1;
}

// Gaussian policy head (reference nn_models/policy.py:170-172): columns of head 0 are means ->
// 5*tanh(x/5), columns of head 1 are log-stds -> exp(clamp(x, -20, 0.5)).
template <typename DESC>
__device__ __forceinline__ float head_value(const DESC& d, int col, float raw) {
    if (d.head_transform != 1) return raw;
    if (col < d.head_cols[0]) return tanhf(raw / 5.f) * 5.f;
    return expf(fminf(fmaxf(raw, -20.f), 0.5f));
}
template <typename DESC>
__device__ __forceinline__ float head_deriv(const DESC& d, int col, float raw) {
    if (d.head_transform != 1) return 1.f;
    if (col < d.head_cols[0]) {
        const float t = tanhf(raw / 5.f);
        return 1.f - t * t;
    }
    return (raw >= -20.f && raw <= 0.5f) ? expf(raw) : 0.f;
}

// (forward jobs carry only this part: kernel arguments are fetched on the launch's critical path, ~0.5-0.9 us per KB)
struct MlpFwdArgs {
    asac_mlp_desc_t d;
    const float* params;
    int64_t member_stride;
    const float* x0;
    int64_t x0_rs, x0_ms;
    const float* x1;
    int64_t x1_rs, x1_ms;
    // optional window addressing of x0 (forward jobs): row = s * x0_T + t lives at x0 + s * x0_sb + t * x0_rs
    // (a [samples, T, in0] view of a larger window, e.g. states[:, b:]); x0_T == 0: flat rows
    int64_t x0_sb;
    int32_t x0_T, pad_;
    int64_t N;
    float* out;          // [E][N][head columns]
};
struct MlpArgs : MlpFwdArgs {
    // backward only
    const float* gout;   // [E][N][head columns]
    float* gx0;          // [E][N][in0] or NULL
    float* gx1;          // [E][N][in1] or NULL
    float* partial;      // [tiles][E][member_stride] or NULL (no parameter gradients)
    // backward in Q-loss mode (gout == NULL): the gradient of the clipped double-Q loss w.r.t. the single
    // head output is formed on chip from the recomputed forward
    const float* tq;     // [E][N] target-network value of the same (state, action) rows
    const float* y;      // [N] return target
    const float* w;      // [N] importance weights or NULL
    float clip_eps;
    float* loss_partial; // [tiles][E] per-tile sums of the (unnormalised) loss
    // backward in policy mode (gout == NULL, tq == NULL): d(mean_b -min_{e in subset} q_e)/dq_e from the
    // ensemble's value table
    const float* q_table;      // [E][N]
    const int32_t* subset;     // device, E_sample members (NULL: 0..E_sample-1)
    int32_t E_sample;
    // backward in policy-sample mode (gout == NULL, eps != NULL; Gaussian-head policy): the gradient w.r.t.
    // (loc | scale) of the rsample / tanh / log-prob chain is formed on chip (asac_squash_sample_bwd's math)
    const float* eps;          // [N][A]
    const float* grad_a;       // [grad_a_members][N][A]  d objective / d tanh-action, summed over members
    int32_t grad_a_members;
    const float* log_alpha;    // dL/dlogp = exp(*log_alpha) / N
};

// a TM x 64 input tile, 4 slots per thread: global -> registers (fetch) and registers -> LDS (put), so a tile
// loop can have the next tile's rows in flight while the current one computes
template <int THREADS, bool WINDOW = false>
__device__ __forceinline__ void fetch_input_tile(const MlpFwdArgs& a, int e, int64_t row0, float (&v)[4], int c0 = 0) {
    // TM x 64 slots / (16 TM) threads = 4 each; columns >= in0+in1 are zero; c0 = 64: the second half of a wide input
    const int in0 = a.d.in0, in1 = a.d.in1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + u * THREADS;
        const int r = i >> 6, c = c0 + (i & 63);
        const int64_t row = row0 + r;
        float x = 0.f;
        if (row < a.N) {
            if (c < in0) {
                int64_t off = row * a.x0_rs;
                if (WINDOW) {       // rows of a [samples, T, in0] window view: (sample, t) addressing
                    const int64_t smp = row / a.x0_T;
                    off = smp * a.x0_sb + (row - smp * a.x0_T) * a.x0_rs;
                }
                x = a.x0[e * a.x0_ms + off + c];
            } else if (c < in0 + in1) {
                x = a.x1[e * a.x1_ms + row * a.x1_rs + (c - in0)];
            }
        }
        v[u] = x;
    }
}

template <int THREADS, int SLOTS = 4>
__device__ __forceinline__ void put_input_tile(const float (&v)[SLOTS], float* xs) {
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int i = threadIdx.x + u * THREADS;
        xs[(i >> 6) * kP + (i & 63)] = v[u];
    }
}

template <int THREADS, bool WINDOW = false>
__device__ __forceinline__ void load_input_tile(const MlpFwdArgs& a, int e, int64_t row0, float* xs, int c0 = 0) {
    float v[4];
    fetch_input_tile<THREADS, WINDOW>(a, e, row0, v, c0);
    put_input_tile<THREADS>(v, xs);
}

// ---- the fixed-shape path (NB > 0: exactly NB blocks, every block 64 wide, first layer <= 64 inputs, parameters
// 16-byte aligned — the reference's stock Q / policy networks): staging without a single branch.  Every slot loads
// from a clamped (always valid) address and selects, so the compiler can issue every scalar and vector load of the
// phase up front and wait once; the generic path's predicated loads compile to a branch per load and a wait per layer.
// every scalar the staging phase needs, read from the kernel arguments in ONE batch at kernel entry and pinned in
// SGPRs (otherwise each is fetched where it is first used: a scalar-memory round trip per use site)
struct StageScalars {
    const float *P, *x0, *x1;
    int64_t N, x0_rs, x0_ms, x1_rs, x1_ms, x0_sb;
    int32_t in0, in1, x0_T, h0, h1, hw0, hw1, hb0, hb1;
    int32_t w_off[kMaxB], b_off[kMaxB];
};
#define ASAC_PIN(x) asm volatile("" : "+s"(x))

template <int NB, typename ARGS>
__device__ __forceinline__ StageScalars stage_scalars(const ARGS& a, int e) {
    StageScalars q;
    q.P = a.params + e * a.member_stride;
    q.x0 = a.x0, q.x1 = a.d.in1 > 0 ? a.x1 : a.x0;
    q.N = a.N, q.x0_rs = a.x0_rs, q.x0_ms = a.x0_ms, q.x1_rs = a.x1_rs, q.x1_ms = a.x1_ms, q.x0_sb = a.x0_sb;
    q.in0 = a.d.in0, q.in1 = a.d.in1, q.x0_T = a.x0_T;
    q.h0 = a.d.head_cols[0], q.h1 = a.d.head_cols[1];
    q.hw0 = a.d.head_w_off[0], q.hw1 = a.d.head_w_off[1], q.hb0 = a.d.head_b_off[0], q.hb1 = a.d.head_b_off[1];
#pragma unroll
    for (int l = 0; l < NB; ++l) q.w_off[l] = a.d.w_off[l], q.b_off[l] = a.d.b_off[l];
    ASAC_PIN(q.P); ASAC_PIN(q.x0); ASAC_PIN(q.x1);
    ASAC_PIN(q.N); ASAC_PIN(q.x0_rs); ASAC_PIN(q.x0_ms); ASAC_PIN(q.x1_rs); ASAC_PIN(q.x1_ms); ASAC_PIN(q.x0_sb);
    ASAC_PIN(q.in0); ASAC_PIN(q.in1); ASAC_PIN(q.x0_T); ASAC_PIN(q.h0); ASAC_PIN(q.h1);
    ASAC_PIN(q.hw0); ASAC_PIN(q.hw1); ASAC_PIN(q.hb0); ASAC_PIN(q.hb1);
#pragma unroll
    for (int l = 0; l < NB; ++l) { ASAC_PIN(q.w_off[l]); ASAC_PIN(q.b_off[l]); }
    return q;
}

// Buffer loads (raw buffer resource, byte offsets): an offset beyond the resource's size returns 0 from the hardware's
// range check, so "this slot is padding" is an offset, not a branch — the whole staging phase is straight-line code.
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned kOob = 0x80000000u;       // beyond every resource below (sizes <= 0x7fffffff bytes)
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes = 0x7fffffffu) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int buf_ld(rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
}

template <int THREADS, bool WINDOW, int SLOTS = 4>
__device__ __forceinline__ void fetch_input_tile_fixed(const StageScalars& q, int e, int64_t row0, float (&v)[SLOTS]) {
    const int in0 = q.in0, in1 = q.in1;
    const rsrc_t r0 = make_rsrc(q.x0 + e * q.x0_ms);
    const rsrc_t r1 = make_rsrc(q.x1 + e * q.x1_ms, in1 > 0 ? 0x7fffffffu : 0u);
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int i = threadIdx.x + u * THREADS;
        const int r = i >> 6, c = i & 63;
        const uint32_t row = (uint32_t)row0 + (uint32_t)r;
        const bool live = (int64_t)row0 + r < q.N;
        uint32_t off = row * (uint32_t)q.x0_rs;
        if (WINDOW) {       // rows of a [samples, T, in0] window view: (sample, t) addressing
            const uint32_t smp = row / (uint32_t)q.x0_T;
            off = smp * (uint32_t)q.x0_sb + (row - smp * (uint32_t)q.x0_T) * (uint32_t)q.x0_rs;
        }
        const unsigned o0 = (live && c < in0) ? (off + (uint32_t)c) * 4u : kOob;
        const unsigned o1 = (live && c >= in0 && c < in0 + in1) ? (row * (uint32_t)q.x1_rs + (uint32_t)(c - in0)) * 4u : kOob;
        v[u] = __builtin_bit_cast(float, buf_ld(r0, o0) | buf_ld(r1, o1));       // (one of the two is the zero pattern)
    }
}

template <int THREADS, int NB>
__device__ __forceinline__ void net_fetch_fixed(const StageScalars& q, StagedNet<THREADS>& r) {
    constexpr int U = 4096 / THREADS;
    const rsrc_t rp = make_rsrc(q.P);
    const int K0 = q.in0 + q.in1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;
        const int rr = i >> 6, c = i & 63;
        r.w[0][u] = __builtin_bit_cast(float, buf_ld(rp, c < K0 ? (unsigned)(q.w_off[0] + rr * K0 + c) * 4u : kOob));
    }
#pragma unroll
    for (int l = 1; l < NB; ++l) {
#pragma unroll
        for (int u = 0; u < U / 4; ++u) {
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rp, (q.w_off[l] + 4 * (int)(threadIdx.x + u * THREADS)) * 4, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.w[l][4 * u + k] = __builtin_bit_cast(float, (int)t[k]);
        }
    }
#pragma unroll
    for (int l = 0; l < NB; ++l) r.bias[l] = __builtin_bit_cast(float, buf_ld(rp, (unsigned)(q.b_off[l] + (threadIdx.x & 63)) * 4u));
    const int h0 = q.h0, h1 = q.h1;
#pragma unroll
    for (int u = 0; u < 1024 / THREADS; ++u) {
        const int i = threadIdx.x + u * THREADS;       // 1024 = 16 x 64
        const int o = i >> 6, c = i & 63;
        const bool first = o < h0, second = !first && o < h0 + h1;
        const int idx = second ? q.hw1 + (o - h0) * kMaxW + c : q.hw0 + o * kMaxW + c;
        r.head[u] = __builtin_bit_cast(float, buf_ld(rp, (first || second) ? (unsigned)idx * 4u : kOob));
    }
    {
        const int o = threadIdx.x & 15;
        const bool first = o < h0, second = !first && o < h0 + h1;
        const int idx = second ? q.hb1 + o - h0 : q.hb0 + o;
        r.head_bias = __builtin_bit_cast(float, buf_ld(rp, (first || second) ? (unsigned)idx * 4u : kOob));
    }
}

template <int THREADS, int NB, typename LDS>
__device__ __forceinline__ void net_put_fixed(const StagedNet<THREADS>& r, LDS& L) {
    constexpr int U = 4096 / THREADS;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * THREADS;
        L.w[0][(i >> 6) * kP + (i & 63)] = r.w[0][u];
    }
#pragma unroll
    for (int l = 1; l < NB; ++l) {
#pragma unroll
        for (int u = 0; u < U / 4; ++u) {
            const int q = threadIdx.x + u * THREADS;   // float4 index: row q/16, col 4*(q%16)
            float2* o2 = reinterpret_cast<float2*>(L.w[l] + (q >> 4) * kP + ((q & 15) << 2));
            o2[0] = make_float2(r.w[l][4 * u], r.w[l][4 * u + 1]);
            o2[1] = make_float2(r.w[l][4 * u + 2], r.w[l][4 * u + 3]);
        }
    }
    if (threadIdx.x < kMaxW) {
#pragma unroll
        for (int l = 0; l < NB; ++l) L.bias[l][threadIdx.x] = r.bias[l];
    }
#pragma unroll
    for (int u = 0; u < 1024 / THREADS; ++u) {
        const int i = threadIdx.x + u * THREADS;
        L.head[(i >> 6) * kP + (i & 63)] = r.head[u];
    }
    if (threadIdx.x < kHeadPad) L.head_bias[threadIdx.x] = r.head_bias;
}

template <int TM>
struct MlpLds {
    float head[kHeadPad * kP];
    float bias[kMaxB][kMaxW];
    float head_bias[kHeadPad];
    float xs[2][TM * kP];            // activation ping-pong
    float w[kMaxB][kMaxW * kP];      // every block's weight [out j][in k], zero padded to 64 x 64; LAST: a launch
};                                   // only allocates the blocks its networks have (mlp_fwd_lds_bytes)

// 3-block networks (the stock Q / policy): 72.8 KB, two workgroups per CU — one's MFMA phase overlaps the other's
// activation phase on the long window launches
// A first layer with more than 64 inputs (up to 128) runs as two 64-column halves: its second weight half takes
// the tile after the network's blocks, the second input half one more activation tile behind that.
inline size_t mlp_fwd_lds_bytes(int n_blocks, bool wide = false, int TM = 32) {
    const size_t head = TM == 16 ? offsetof(MlpLds<16>, w) : offsetof(MlpLds<32>, w);
    return head + (size_t)(n_blocks + (wide ? 1 : 0)) * kMaxW * kP * sizeof(float) +
           (wide ? (size_t)TM * kP * sizeof(float) : 0);
}

// ------------------------------------------------------------------------------------------------
// One workgroup evaluates member e on the row tiles tile0, tile0 + tile_stride, ...: every layer's weights
// are staged into LDS ONCE and reused by all of them (for window-sized inputs — tens of thousands of rows —
// re-staging 36 KB of weights per tile would be most of the traffic).
// EPI: a Gaussian-head policy job whose rows are sampled from (and whose stored actions are scored) by the lanes that form
// the head (asac_squash.h `sample_epilogue`: asac_squash_multi's jobs without a launch of their own)
template <int TM, bool WINDOW, bool WIDE = false, int NB = 0, bool EPI = false>
__device__ __forceinline__ void mlp_fwd_tiles(const MlpFwdArgs& a, const int e, const int tile0, const int tile_stride,
                                              MlpLds<TM>& L, const SampleEpi* epi = nullptr) {
    constexpr int THREADS = threads_of<TM>();
    constexpr int RT = TM / 16;                               // row tiles of a workgroup tile
    constexpr bool fixed = NB > 0;                            // NB blocks of 64 (see net_fetch_fixed)
    static_assert(!(fixed && WIDE), "the fixed-shape path has a first layer of <= 64 inputs");
    const float* P = a.params + e * a.member_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rt = wave % RT, ct = wave / RT;
    const int nb = fixed ? NB : a.d.n_blocks;
    const int K0 = a.d.in0 + a.d.in1;
    const int n_tiles = (int)((a.N + TM - 1) / TM);

    // one staging phase (with the first input tile), one barrier
    constexpr bool wide = WIDE;                              // K0 > 64 (the stock networks' instantiation carries none of it)
    float* w_hi = &L.w[0][0] + nb * kMaxW * kP;              // second half of a wide first layer
    float* x_hi = w_hi + kMaxW * kP;                         // ... and of its input tile
    StageScalars q;
    if constexpr (fixed) {
        q = stage_scalars<NB>(a, e);
        float in_lo[4];
        fetch_input_tile_fixed<THREADS, WINDOW>(q, e, (int64_t)tile0 * TM, in_lo);
        StagedNet<THREADS> regs;
        net_fetch_fixed<THREADS, NB>(q, regs);
        put_input_tile<THREADS>(in_lo, L.xs[0]);
        net_put_fixed<THREADS, NB>(regs, L);
    } else {
        float in_lo[4], in_hi[4];
        fetch_input_tile<THREADS, WINDOW>(a, e, (int64_t)tile0 * TM, in_lo);
        if (wide) fetch_input_tile<THREADS, WINDOW>(a, e, (int64_t)tile0 * TM, in_hi, kMaxW);
        StagedNet<THREADS> regs;
        net_fetch<THREADS>(a.d, P, K0, wide, regs);
        put_input_tile<THREADS>(in_lo, L.xs[0]);
        if (wide) {
            put_input_tile<THREADS>(in_hi, x_hi);
            stage_matrix_part<THREADS>(L.w[0], P + a.d.w_off[0], a.d.width[0], K0, 0, kMaxW);
            stage_matrix_part<THREADS>(w_hi, P + a.d.w_off[0], a.d.width[0], K0, kMaxW, K0 - kMaxW);
        }
        net_put<THREADS>(a.d, P, K0, wide, regs, L);
    }
    const int K_last = fixed ? kMaxW : a.d.width[nb - 1];
    __syncthreads();

    const int O = a.d.head_cols[0] + a.d.head_cols[1];
    int cur = 0;                   // the buffer holding the current tile's input
    for (int tile = tile0; tile < n_tiles; tile += tile_stride) {
        const int64_t row0 = (int64_t)tile * TM;
        // the next tile's rows travel while this tile computes; they land in the buffer the head phase leaves free
        const bool more = tile + tile_stride < n_tiles;
        float nxt[4], nxt_hi[4];
        if (more) {
            if constexpr (fixed) fetch_input_tile_fixed<THREADS, WINDOW>(q, e, (int64_t)(tile + tile_stride) * TM, nxt);
            else fetch_input_tile<THREADS, WINDOW>(a, e, (int64_t)(tile + tile_stride) * TM, nxt);
        }
        if (more && wide) fetch_input_tile<THREADS, WINDOW>(a, e, (int64_t)(tile + tile_stride) * TM, nxt_hi, kMaxW);
        // the sampling epilogue's operands of this lane's head row travel under the layers (asac_squash.h)
        SampleEpiIn epi_in{};
        bool epi_on = false;
        if constexpr (EPI) {
            epi_on = epi->on != 0;                            // (uniform)
            if (epi_on)
                epi_in = sample_epilogue_fetch(*epi, row0 + (wave % RT) * 16 + 4 * (lane >> 4) + wave / RT, a.N, lane & 15);
        }
        int K = K0;
#pragma unroll
        for (int l = 0; l < (fixed ? NB : kMaxB); ++l) {
            if (l >= nb) break;
            const int W = fixed ? kMaxW : a.d.width[l];
            const float* xin = L.xs[cur];
            float* xout = L.xs[cur ^ 1];
            f32x4 acc = (fixed && l > 0) ? gemm_tile(xin, L.w[l], kMaxW, rt, ct)
                                         : gemm_tile(xin, L.w[l], (l == 0 && wide) ? kMaxW : round4(K), rt, ct);
            if (l == 0 && wide) acc += gemm_tile(x_hi, w_hi, round4(K0 - kMaxW), rt, ct);
            const int col = ct * 16 + (lane & 15);
            const float bias = L.bias[l][col];
            const bool res = a.d.residual[l] != 0;
            f32x2_g ya, yb, unused;       // the four elements as two packed pairs
            gelu_parts2((f32x2_g){acc[0] + bias, acc[1] + bias}, ya, unused);
            gelu_parts2((f32x2_g){acc[2] + bias, acc[3] + bias}, yb, unused);
            const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + 4 * (lane >> 4) + r;
                float y = yv[r];
                if (res) y += xin[row * kP + col];
                xout[row * kP + col] = col < W ? y : 0.f;
            }
            __syncthreads();
            cur ^= 1;
            K = W;
        }
        if (more) put_input_tile<THREADS>(nxt, L.xs[cur ^ 1]);
        if (more && wide) put_input_tile<THREADS>(nxt_hi, x_hi);
        // heads: one padded column tile per row tile.  Transformed heads (tanh / exp per element, both branches taken by
        // every wave) of short launches: all 4 RT waves form their row tile's head (the same MFMA chain: the same values)
        // and each finishes ONE of the four rows a lane holds; otherwise the first wave of every row tile
        // (with an epilogue every wave takes the all-waves form: the rows' transcendental chains are what the head phase costs)
        if (a.d.head_transform == 1 && (n_tiles <= 256 || epi_on)) {      // (launches of at most a tile per CU: latency is what counts)
            const int hrt = wave % RT, hr = wave / RT;
            const f32x4 acc = gemm_tile(L.xs[cur], L.head, round4(K_last), hrt, 0);
            const int col = lane & 15;
            const float mine = hr == 0 ? acc[0] : hr == 1 ? acc[1] : hr == 2 ? acc[2] : acc[3];
            const int64_t row = row0 + hrt * 16 + 4 * (lane >> 4) + hr;
            const float hv = head_value(a.d, col, mine + L.head_bias[col]);
            if (row < a.N && col < O) a.out[((int64_t)e * a.N + row) * O + col] = hv;
            if constexpr (EPI) {
                if (epi_on) sample_epilogue(*epi, epi_in, col, hv, lane);
            }
        } else if (wave < RT) {
            const f32x4 acc = gemm_tile(L.xs[cur], L.head, round4(K_last), wave, 0);
            const int col = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + wave * 16 + 4 * (lane >> 4) + r;
                if (row < a.N && col < O)
                    a.out[((int64_t)e * a.N + row) * O + col] = head_value(a.d, col, acc[r] + L.head_bias[col]);
            }
        }
        if (more) __syncthreads();     // the head readers are done with this tile, the next tile's input is in place
        cur ^= 1;
    }
}

template <int TM, bool WIDE, int NB>
__global__ __launch_bounds__(TM * 16) void k_mlp_fwd(const MlpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    mlp_fwd_tiles<TM, false, WIDE, NB>(a, blockIdx.y, blockIdx.x, gridDim.x, *reinterpret_cast<MlpLds<TM>*>(smem_raw));
}

// Several independent forward passes (different networks / inputs) in ONE launch: blocks are dealt to the
// jobs in order, a job's blocks to (tile, member) pairs.
struct MlpMultiArgs {
    MlpFwdArgs job[ASAC_MLP_MAX_JOBS];
    int32_t E[ASAC_MLP_MAX_JOBS], first_block[ASAC_MLP_MAX_JOBS], tile_stride[ASAC_MLP_MAX_JOBS];
    int32_t n, blocks;      // jobs; workgroups of the jobs (sidecar workgroups follow)
};

// workgroups along the row-tile axis: one per tile while that keeps the whole grid within about one
// resident wave of workgroups (one per CU: the LDS footprint), else a fixed number that loop over tiles
inline int mlp_tile_groups(int64_t N, int E, int per_cu, int TM) {
    const int tiles = (int)((N + TM - 1) / TM);
    const int cap = 256 * per_cu / (E > 0 ? E : 1);
    return tiles <= (cap > 1 ? cap : 1) ? tiles : (cap > 1 ? cap : 1);
}

template <int TM, int NB>
__global__ __launch_bounds__(TM * 16) void k_mlp_fwd_multi(const MlpMultiArgs m, const SidecarsDev sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if ((int)blockIdx.x >= m.blocks) {          // sidecar workgroups (asac_sidecar.h)
        sidecar_run(sc, (int)blockIdx.x - m.blocks, reinterpret_cast<float*>(smem_raw));
        return;
    }
    int k = 0;
#pragma unroll
    for (int q = 1; q < ASAC_MLP_MAX_JOBS; ++q)
        if (q < m.n && (int)blockIdx.x >= m.first_block[q]) k = q;
    const int local = (int)blockIdx.x - m.first_block[k];
    const int E = m.E[k];
    MlpLds<TM>& L = *reinterpret_cast<MlpLds<TM>*>(smem_raw);
    if (m.job[k].x0_T > 0)
        mlp_fwd_tiles<TM, true, false, NB>(m.job[k], local % E, local / E, m.tile_stride[k], L);
    else
        mlp_fwd_tiles<TM, false, false, NB>(m.job[k], local % E, local / E, m.tile_stride[k], L);
}

// ... with a sampling epilogue per job (asac_mlp_forward_multi_sampled)
struct SampleEpis {
    SampleEpi e[ASAC_MLP_MAX_JOBS];
};
template <int TM, int NB>
__global__ __launch_bounds__(TM * 16) void k_mlp_fwd_multi_sampled(const MlpMultiArgs m, const SampleEpis epis, const SidecarsDev sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if ((int)blockIdx.x >= m.blocks) {          // sidecar workgroups (asac_sidecar.h)
        sidecar_run(sc, (int)blockIdx.x - m.blocks, reinterpret_cast<float*>(smem_raw));
        return;
    }
    int k = 0;
#pragma unroll
    for (int q = 1; q < ASAC_MLP_MAX_JOBS; ++q)
        if (q < m.n && (int)blockIdx.x >= m.first_block[q]) k = q;
    const int local = (int)blockIdx.x - m.first_block[k];
    const int E = m.E[k];
    MlpLds<TM>& L = *reinterpret_cast<MlpLds<TM>*>(smem_raw);
    if (m.job[k].x0_T > 0)
        mlp_fwd_tiles<TM, true, false, NB, true>(m.job[k], local % E, local / E, m.tile_stride[k], L, &epis.e[k]);
    else
        mlp_fwd_tiles<TM, false, false, NB, true>(m.job[k], local % E, local / E, m.tile_stride[k], L, &epis.e[k]);
}

// ------------------------------------------------------------------------------------------------
// Backward.  LDS: block inputs x[0..nb] (x_0 = network input, x_l = output of block l), every
// weight, one delta tile.  Registers: this wave's fragment of every pre-activation z_l.
// ------------------------------------------------------------------------------------------------
template <int TM>
struct MlpBwdLds {
    float w[kMaxB][kMaxW * kP];
    float head[kHeadPad * kP];
    float x[kMaxB + 1][TM * kP];
    float delta[TM * kP];
    float bias[kMaxB][kMaxW];
    float head_bias[kHeadPad];
};

// partial dW[j][k] = sum_rows delta[row][jbase + j] * xprev[row][k]  -> out[j*K + k]
// 8 waves (TM = 32): wave w owns j tile w>>1, k tiles 2*(w&1) and 2*(w&1)+1; 4 waves (TM = 16): j tile w, all k tiles.
// The MFMA computes the TRANSPOSED tile (A = xprev^T, B = delta): a lane then holds four consecutive k of one row j,
// i.e. one 16-byte store per tile instead of four scattered 4-byte ones (the stores were 2/3 of this phase).
template <int TM>
__device__ __forceinline__ void grad_weight(const float* __restrict__ delta, int jbase,
                                            const float* __restrict__ xprev, int J, int K,
                                            float* __restrict__ out, int out_stride = 0) {
    constexpr int RT = TM / 16, KT = 4 / RT, STEPS = TM / 4;
    if (out_stride == 0) out_stride = K;           // (a 64-column half of a wider matrix passes the full width)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int jt = wave / RT;
    if (jt * 16 >= J) return;
    const bool vec = (out_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    float dv[STEPS];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) dv[i] = delta[(4 * i + lk) * kP + jbase + jt * 16 + lr];   // B[kk = row][jj = j]
#pragma unroll
    for (int h = 0; h < KT; ++h) {
        const int kt = KT * (wave % RT) + h;
        if (kt * 16 >= K) continue;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float xv[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) xv[i] = xprev[(4 * i + lk) * kP + kt * 16 + lr];       // A[i = k][kk = row]
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEPS; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i], dv[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i + 1], dv[i + 1], acc1, 0, 0, 0);
        }
        const f32x4 acc = acc0 + acc1;      // acc[r] = dW[j = jt*16 + lr][k = kt*16 + 4*lk + r]
        const int j = jt * 16 + lr, k0 = kt * 16 + 4 * lk;
        if (j >= J) continue;
        float* o = out + j * out_stride + k0;
        if (vec && k0 + 3 < K) {
            *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k0 + r < K) o[r] = acc[r];
        }
    }
}

// partial db[j] = sum_rows delta[row][jbase + j]: wave w < 4 owns columns 16w..16w+15, a lane TM/4 rows of one
// column, the four row groups meet through two shuffles
template <int TM>
__device__ __forceinline__ void grad_bias(const float* __restrict__ delta, int jbase, int J,
                                          float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4 || wave * 16 >= J) return;
    const int c = wave * 16 + (lane & 15), r0 = (lane >> 4) * (TM / 4);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < TM / 4; ++r) s += delta[(r0 + r) * kP + jbase + c];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16 && c < J) out[c] = s;
}

// grad_weight<16> with the output's (J/16) x (K/16) MFMA tiles dealt over 8 waves (same chain per tile)
// ... by `workers` waves, this one being number `worker` of them (a wave with worker < 0 has no part in it)
__device__ __forceinline__ void grad_weight_tiles(const float* __restrict__ delta, int jbase, const float* __restrict__ xprev,
                                                  int J, int K, float* __restrict__ out, int worker, int workers) {
    if (worker < 0) return;
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int njt = (J + 15) >> 4, nkt = (K + 15) >> 4;
    const bool vec = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int t = worker; t < njt * nkt; t += workers) {
        const int jt = t / nkt, kt = t - jt * nkt;
        float dv[4], xv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dv[i] = delta[(4 * i + lk) * kP + jbase + jt * 16 + lr];
            xv[i] = xprev[(4 * i + lk) * kP + kt * 16 + lr];
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[0], dv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[1], dv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[2], dv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[3], dv[3], acc1, 0, 0, 0);
        const f32x4 acc = acc0 + acc1;
        const int j = jt * 16 + lr, k0 = kt * 16 + 4 * lk;
        if (j >= J) continue;
        float* o = out + j * K + k0;
        if (vec && k0 + 3 < K) {
            *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k0 + r < K) o[r] = acc[r];
        }
    }
}
template <int NW = ASAC_PS_WAVES>
__device__ __forceinline__ void ps_grad_weight(const float* __restrict__ delta, int jbase, const float* __restrict__ xprev,
                                               int J, int K, float* __restrict__ out, int first_wave = 0) {
    grad_weight_tiles(delta, jbase, xprev, J, K, out, ((threadIdx.x >> 6) + NW - first_wave) & (NW - 1), NW);
}

// RET: the Q-loss backward forms its own return target (asac_mlp_backward_qloss_return): every workgroup evaluates
// the n-step V-trace return of ITS tile's rows (asac_vtrace.h: the return kernel's per-step terms and its scan
// association, bit-identical) instead of reading y from a launch in front.  The (row, step) loads are issued right
// behind the weight staging's — one round, travelling under it — so the tile's y is in LDS long before the loss phase
// wants it: the return launch (K4) and its boundary leave the step's chain.  LDS behind the tile struct:
// [TM][pitch] d_t | [TM][pitch] c_t | [TM] V(s_0) | [TM] y, pitch = (n + 1) | 1.
template <bool RET>
struct RetIn {};
template <>
struct RetIn<true> {
    asac_vtrace_args_t v;
    int32_t seg;              // vtrace_scan_lanes(B, n): the stand-alone kernel's scan association
};

// W8 (16-row tiles of the stock networks only): EIGHT or SIXTEEN waves instead of four.  The forward recompute and the
// dX chains have four column tiles, i.e. work for four waves — the others share the staging (its loads, LDS writes and
// address arithmetic are dealt over all 512 / 1024 threads) and the weight-gradient tiles (ps_grad_weight: two / one
// 16 x 16 tiles per wave and layer instead of four; the same MFMA chain per tile: the same partial sums).
template <int TM, bool WIDE, int NB, bool RET = false, int W8 = 0>      // W8: 0, or the number of waves (8 / 16)
__global__ __launch_bounds__(W8 ? 64 * W8 : TM * 16) void k_mlp_bwd(const MlpArgs a, const RetIn<RET> rv) {
    static_assert(W8 == 0 || ((W8 == 8 || W8 == 16) && TM == 16 && NB == 3 && !WIDE), "8 / 16 waves: 16-row tiles of the stock networks");
    constexpr int THREADS = W8 ? 64 * W8 : threads_of<TM>();
    constexpr int RT = TM / 16;
    constexpr bool fixed = NB > 0;                            // NB blocks of 64 (see net_fetch_fixed)
    static_assert(!(fixed && WIDE), "the fixed-shape path has a first layer of <= 64 inputs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpBwdLds<TM>& L = *reinterpret_cast<MlpBwdLds<TM>*>(smem_raw);
    const int e = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * TM;
    const float* P = a.params + e * a.member_stride;
    float* part = a.partial ? a.partial + ((int64_t)blockIdx.x * gridDim.y + e) * a.member_stride : nullptr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool comp = !W8 || wave < 4;          // this wave owns a column tile of the forward / dX chains
    const int rt = wave % RT, ct = comp ? wave / RT : 0;
    const int nb = fixed ? NB : a.d.n_blocks;
    const int K0 = a.d.in0 + a.d.in1;
    const int O = a.d.head_cols[0] + a.d.head_cols[1];
    const int col = ct * 16 + (lane & 15);

    MLP_STAMP(0);
    // ---- staging: input tile, every weight, the incoming gradient tile (padded to 16 columns) -------
    // (a first layer wider than 64 inputs: second halves in the spare tiles x[kMaxB] / w[nb], desc_ok keeps nb < kMaxB)
    constexpr bool wide = WIDE;              // K0 > 64
    float* x_hi = L.x[kMaxB];
    float* w_hi = L.w[nb < kMaxB ? nb : kMaxB - 1];
    if constexpr (fixed) {
        const StageScalars q = stage_scalars<NB>(a, e);
        constexpr int SLOTS = TM * 64 / THREADS;                  // of the TM x 64 input tile per thread (4; W8: 2)
        float in_lo[SLOTS];
        fetch_input_tile_fixed<THREADS, false, SLOTS>(q, e, row0, in_lo);
        const int r = threadIdx.x >> 4, c = threadIdx.x & 15;     // TM x 16 slots (W8: the first 256 threads)
        const int64_t row = row0 + r;
        float gin = 0.f;
        if (a.gout && r < TM) {       // (a.gout: uniform)
            const float t = a.gout[((int64_t)e * a.N + (row < a.N ? row : a.N - 1)) * O + (c < O ? c : O - 1)];
            gin = (row < a.N && c < O) ? t : 0.f;
        }
        StagedNet<THREADS> regs;
        net_fetch_fixed<THREADS, NB>(q, regs);
        if constexpr (RET) {
            const asac_vtrace_args_t& v = rv.v;
            const int n = v.n, pitch = (n + 1) | 1;
            float* s_d = reinterpret_cast<float*>(smem_raw + sizeof(MlpBwdLds<TM>));
            float* s_c = s_d + TM * pitch;
            float* s_v0 = s_c + TM * pitch;
            const int f = threadIdx.x, fr = f / n, ft = f - fr * n;
            const bool have = f < TM * n && row0 + fr < a.N;
            VtraceStepRaw raw{};
            float log_alpha = 0.f;
            if (have) {
                raw = vtrace_step_load(v, (int)(row0 + fr), ft);
                log_alpha = *v.log_alpha;
            }
            put_input_tile<THREADS, SLOTS>(in_lo, L.x[0]);
            net_put_fixed<THREADS, NB>(regs, L);
            if (have) {
                float d, cc;
                const float v_t = vtrace_step_finish(v, raw, expf(log_alpha), &d, &cc);
                if (ft == 0) s_v0[fr] = v_t;
                s_d[fr * pitch + ft] = d;
                s_c[fr * pitch + ft] = cc;
            }
        } else {
            put_input_tile<THREADS, SLOTS>(in_lo, L.x[0]);
            net_put_fixed<THREADS, NB>(regs, L);
        }
        if (r < TM) L.delta[r * kP + c] = gin;
    } else {
        float in_lo[4], in_hi[4];
        fetch_input_tile<THREADS>(a, e, row0, in_lo);
        if (wide) fetch_input_tile<THREADS>(a, e, row0, in_hi, kMaxW);
        const int r = threadIdx.x >> 4, c = threadIdx.x & 15;     // TM x 16 = THREADS slots
        const int64_t row = row0 + r;
        const float gin = (a.gout && row < a.N && c < O) ? a.gout[((int64_t)e * a.N + row) * O + c] : 0.f;
        StagedNet<THREADS> regs;
        net_fetch<THREADS>(a.d, P, K0, wide, regs);
        put_input_tile<THREADS>(in_lo, L.x[0]);
        if (wide) {
            put_input_tile<THREADS>(in_hi, x_hi);
            stage_matrix_part<THREADS>(L.w[0], P + a.d.w_off[0], a.d.width[0], K0, 0, kMaxW);
            stage_matrix_part<THREADS>(w_hi, P + a.d.w_off[0], a.d.width[0], K0, kMaxW, K0 - kMaxW);
        }
        net_put<THREADS>(a.d, P, K0, wide, regs, L);
        L.delta[r * kP + c] = gin;
    }
    __syncthreads();
    MLP_STAMP(1);
    if constexpr (RET) {     // one lane per row of the tile: y = V(s_0) + scan; first read in the loss phase (barriers between)
        const asac_vtrace_args_t& v = rv.v;
        const int n = v.n, pitch = (n + 1) | 1;
        float* s_d = reinterpret_cast<float*>(smem_raw + sizeof(MlpBwdLds<TM>));
        float* s_c = s_d + TM * pitch;
        float* s_v0 = s_c + TM * pitch;
        float* s_y = s_v0 + TM;
        if ((int)threadIdx.x < TM && row0 + threadIdx.x < a.N) {
            const int r = threadIdx.x;
            const float y = s_v0[r] + vtrace_scan_row(s_d + r * pitch, s_c + r * pitch, n, rv.seg);
            s_y[r] = y;
            if (e == 0) v.y_out[row0 + r] = y;
        }
    }

    // ---- forward recompute ----------------------------------------------------------------------------
    f32x4 z[kMaxB];          // gelu'(pre-activation) of this wave's fragment, per block
    int K = K0;
#pragma unroll
    for (int l = 0; l < kMaxB; ++l) {
        if (l < nb) {
            const int W = fixed ? kMaxW : a.d.width[l];
            const float* xin = L.x[l];
            float* xout = L.x[l + 1];
            if (l == 1) MLP_STAMP(20);
            if (!comp) {              // (W8: waves 4..7 have no column tile)
                __syncthreads();
                K = W;
                continue;
            }
            f32x4 acc = (fixed && l > 0) ? gemm_tile(xin, L.w[l], kMaxW, rt, ct)
                                         : gemm_tile(xin, L.w[l], (l == 0 && wide) ? kMaxW : round4(K), rt, ct);
            if (l == 0 && wide) acc += gemm_tile(x_hi, w_hi, round4(K0 - kMaxW), rt, ct);
            if (l == 1) MLP_STAMP(21);
            const float bias = L.bias[l][col];
            const bool res = a.d.residual[l] != 0;
            // value and derivative of the four elements as two packed pairs; the derivative replaces the
            // pre-activation in the registers the reverse pass reads
            f32x2_g ya, yb, da, db;
            gelu_parts2((f32x2_g){acc[0] + bias, acc[1] + bias}, ya, da);
            gelu_parts2((f32x2_g){acc[2] + bias, acc[3] + bias}, yb, db);
            const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
            z[l] = f32x4{da.x, da.y, db.x, db.y};
            if (l == 1) MLP_STAMP(22);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + 4 * (lane >> 4) + r;
                float y = yv[r];
                if (res) y += xin[row * kP + col];
                xout[row * kP + col] = col < W ? y : 0.f;
            }
            if (l == 1) MLP_STAMP(23);
            __syncthreads();
            if (l == 1) MLP_STAMP(24);
            K = W;
        }
    }
    const int H = K;   // width of the last hidden layer
    MLP_STAMP(2);

    // policy-sample mode (Gaussian head): raw head values -> (loc, scale) -> gradient of the sampled action /
    // log-prob chain -> chain rule through the head transform, all on this tile
    if (!a.gout && a.eps) {
        const int A = a.d.head_cols[0];
        if (wave < RT) {
            const f32x4 raw = gemm_tile(L.x[nb], L.head, round4(H), wave, 0);
            const int hc = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) L.delta[(wave * 16 + 4 * (lane >> 4) + r) * kP + hc] = raw[r] + L.head_bias[hc];
        }
        __syncthreads();
        if ((int)threadIdx.x < TM * A) {
            const int lrow = threadIdx.x / A, d = threadIdx.x - lrow * A;
            const int64_t row = row0 + lrow;
            float g_loc = 0.f, g_scale = 0.f;
            const float raw_l = L.delta[lrow * kP + d], raw_s = L.delta[lrow * kP + A + d];
            if (row < a.N) {
                const float loc = head_value(a.d, d, raw_l), sc = head_value(a.d, A + d, raw_s);
                const float ev = a.eps[row * A + d];
                const float t = tanhf(loc + ev * sc);
                const float one_m = 1.f - t * t;
                float ga = 0.f;
                for (int m = 0; m < a.grad_a_members; ++m) ga += a.grad_a[((int64_t)m * a.N + row) * A + d];
                const float gl = expf(*a.log_alpha) * (1.f / (float)a.N);
                float gx = ga * one_m;
                if (one_m > 1e-2f) gx += gl * ((float)A * 2.f * t);     // squash-correction floor, operators.py:12-14
                g_loc = gx * head_deriv(a.d, d, raw_l);
                g_scale = (gx * ev - gl / sc) * head_deriv(a.d, A + d, raw_s);
            }
            L.delta[lrow * kP + d] = g_loc;
            L.delta[lrow * kP + A + d] = g_scale;
        }
        __syncthreads();
    }
    // policy mode: the gradient of mean_b(-min_{e in subset} q_e) w.r.t. this member's q: -1/N on the rows
    // where it is the (first) arg-min of the subset, else 0 (reference sac_base.py:1896-1903)
    if (!a.gout && !a.eps && a.q_table) {
        if (threadIdx.x < TM) {
            const int64_t row = row0 + threadIdx.x;
            float g = 0.f;
            if (row < a.N) {
                int best = a.subset ? a.subset[0] : 0;
                float m = a.q_table[(int64_t)best * a.N + row];
                for (int k = 1; k < a.E_sample; ++k) {
                    const int ee = a.subset ? a.subset[k] : k;
                    const float vq = a.q_table[(int64_t)ee * a.N + row];
                    if (vq < m) {
                        m = vq;
                        best = ee;
                    }
                }
                if (best == e) g = -1.f / (float)a.N;
            }
            L.delta[threadIdx.x * kP] = g;
        }
        __syncthreads();
    }
    // Q-loss mode: q = head(x) for this tile, delta[:, 0] = d(mean_b l)/dq, per-tile loss sum
    if (!a.gout && !a.eps && !a.q_table) {
        __shared__ float loss_red[4 * RT];
        if (wave < RT) {
            const f32x4 raw = gemm_tile(L.x[nb], L.head, round4(H), wave, 0);
            float part = 0.f;
            if ((lane & 15) == 0) {
                const float inv_n = 1.f / (float)a.N;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lrow = wave * 16 + 4 * (lane >> 4) + r;
                    const int64_t row = row0 + lrow;
                    float g = 0.f;
                    if (row < a.N) {
                        float y_row;
                        if constexpr (RET) {
                            const int pitch = (rv.v.n + 1) | 1;
                            y_row = reinterpret_cast<const float*>(smem_raw + sizeof(MlpBwdLds<TM>))[(2 * pitch + 1) * TM + lrow];
                        } else {
                            y_row = a.y[row];
                        }
                        part += clipped_q_loss_row(raw[r] + L.head_bias[0], a.tq[(int64_t)e * a.N + row], y_row,
                                                   a.w ? a.w[row] : 1.f, a.clip_eps, &g);
                    }
                    L.delta[lrow * kP] = inv_n * g;
                }
                loss_red[wave * 4 + (lane >> 4)] = part;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int i = 0; i < 4 * RT; ++i) s += loss_red[i];
            a.loss_partial[(int64_t)blockIdx.x * gridDim.y + e] = s;
        }
    }

    // transformed head: the incoming gradient is w.r.t. the transformed outputs; recompute the raw
    // head values for this tile and apply the chain rule in place on the delta tile
    if (a.d.head_transform != 0 && a.gout) {
        if (wave < RT) {
            const f32x4 raw = gemm_tile(L.x[nb], L.head, round4(H), wave, 0);
            const int hc = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + 4 * (lane >> 4) + r;
                L.delta[row * kP + hc] *= head_deriv(a.d, hc, raw[r] + L.head_bias[hc]);
            }
        }
        __syncthreads();
    }

    MLP_STAMP(3);
    // ---- head: parameter grads, then g = gout * Wh -------------------------------------------------------
    if (part) {
        int jb = 0;
        for (int h = 0; h < 2; ++h) {
            if (a.d.head_cols[h] > 0) {
                if constexpr (W8 != 0) grad_weight_tiles(L.delta, jb, L.x[nb], a.d.head_cols[h], H, part + a.d.head_w_off[h], wave - 4, W8 - 4);
                else grad_weight<TM>(L.delta, jb, L.x[nb], a.d.head_cols[h], H, part + a.d.head_w_off[h]);
                grad_bias<TM>(L.delta, jb, a.d.head_cols[h], part + a.d.head_b_off[h]);
            }
            jb += a.d.head_cols[h];
        }
    }
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (comp) g = gemm_tile_nt(L.delta, L.head, kHeadPad, rt, ct);      // g[row][k], k over H
    MLP_STAMP(4);

    // ---- blocks in reverse ---------------------------------------------------------------------------------
    f32x4 g_hi = {0.f, 0.f, 0.f, 0.f};       // input gradient of columns 64.. (wide first layer)
#pragma unroll
    for (int l = kMaxB - 1; l >= 0; --l) {
        if (l < nb) {
            const int W = fixed ? kMaxW : a.d.width[l];
            const int Kin = (l == 0) ? K0 : (fixed ? kMaxW : a.d.width[l - 1]);
            __syncthreads();   // readers of the previous delta tile are done
            if (comp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + 4 * (lane >> 4) + r;
                    L.delta[row * kP + col] = col < W ? g[r] * z[l][r] : 0.f;     // z holds gelu'(pre-activation)
                }
            }
            __syncthreads();
            if (l == 2) MLP_STAMP(16);
            if (part) {
                if (l == 0 && wide) {
                    grad_weight<TM>(L.delta, 0, L.x[0], W, kMaxW, part + a.d.w_off[0], K0);
                    grad_weight<TM>(L.delta, 0, x_hi, W, K0 - kMaxW, part + a.d.w_off[0] + kMaxW, K0);
                } else if constexpr (W8 != 0) {
                    // (the waves without a column tile take the weight-gradient tiles; waves 0..3 go straight on to the
                    // bias gradients and the dX chain: the two run side by side)
                    grad_weight_tiles(L.delta, 0, L.x[l], W, Kin, part + a.d.w_off[l], wave - 4, W8 - 4);
                } else {
                    grad_weight<TM>(L.delta, 0, L.x[l], W, Kin, part + a.d.w_off[l]);
                }
                if (l == 2) MLP_STAMP(17);
                grad_bias<TM>(L.delta, 0, W, part + a.d.b_off[l]);
                if (l == 2) MLP_STAMP(18);
            }
            if (l == 0 && wide && (a.gx0 || a.gx1)) g_hi = gemm_tile_nt(L.delta, w_hi, round4(W), rt, ct);
            if (comp) {
                f32x4 gin = gemm_tile_nt(L.delta, L.w[l], fixed ? kMaxW : round4(W), rt, ct);   // d x_{l-1}[row][k]
                if (a.d.residual[l]) gin += g;
                g = gin;
            }
            MLP_STAMP(5 + (kMaxB - 1 - l));
        }
    }
    // ---- input gradients --------------------------------------------------------------------------------------
    if (comp && (a.gx0 || a.gx1)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + rt * 16 + 4 * (lane >> 4) + r;
            if (row >= a.N) continue;
            if (col < a.d.in0) {
                if (a.gx0) a.gx0[((int64_t)e * a.N + row) * a.d.in0 + col] = g[r];
            } else if (col < K0) {
                if (a.gx1) a.gx1[((int64_t)e * a.N + row) * a.d.in1 + (col - a.d.in0)] = g[r];
            }
            const int ch = col + kMaxW;            // the second half of a wide input
            if (wide && ch < K0) {
                if (ch < a.d.in0) {
                    if (a.gx0) a.gx0[((int64_t)e * a.N + row) * a.d.in0 + ch] = g_hi[r];
                } else if (a.gx1) {
                    a.gx1[((int64_t)e * a.N + row) * a.d.in1 + (ch - a.d.in0)] = g_hi[r];
                }
            }
        }
    }
    MLP_STAMP(10);
}

// ------------------------------------------------------------------------------------------------
// The whole policy step of the stock networks in ONE launch (reference sac_base.py:1883-1906), two critics:
//   Q_0, Q_1 forward on (s, a~pi)  ->  d(mean_b -min_e q_e)/dq  ->  both critics' backward to the ACTION only
//   ->  rsample / tanh / log-prob backward  ->  policy backward with parameter gradients
// i.e. asac_mlp_forward + asac_mlp_backward_policy_q + asac_mlp_backward_policy_sample, whose three launches
// each stage 36-72 KB of weights and hand 16 KB of intermediate gradients to the next through L2.  Every step is
// row-local, so a workgroup owns a 16-row tile from the critics' inputs to the policy's parameter-gradient
// partials.  8 waves: in the critic phase waves 0-3 run member 0 and waves 4-7 member 1 side by side; the policy's
// weights travel in registers meanwhile and take over the critics' LDS once they are done with it; in the policy
// phase the 16 (heads: 8) independent weight-gradient tiles of a layer are dealt over all 8 waves.  Every MFMA
// chain has the operand order of the three separate kernels: results are bit-identical to theirs.
constexpr int kPsWaves = ASAC_PS_WAVES;
constexpr int kPsThreads = 64 * kPsWaves;
constexpr int kPsSlots = 1024 / kPsThreads;         // of the 16 x 64 input tile per thread
constexpr int kPsRows = 16;

struct PsQLds {       // one critic: weights, scalar head, bias, two activation tiles (forward ping-pong; then delta)
    float w[3][kMaxW * kP];
    float head[kHeadPad * kP];
    float bias[3][kMaxW];
    float head_bias[kHeadPad];
    float xs[2][kPsRows * kP];
};
struct PsPiLds {      // the policy's backward: weights, heads, block inputs x_0..x_3, one delta tile
    float w[3][kMaxW * kP];
    float head[kHeadPad * kP];
    float bias[3][kMaxW];
    float head_bias[kHeadPad];
    float x[4][kPsRows * kP];
    float delta[kPsRows * kP];
};
struct PsLds {
    PsQLds q[2];                           // the policy phase's PsPiLds overlays this
    float ga[2][kPsRows * kHeadPad];       // the members' action gradients of the tile
    float qv[2][kPsRows];                  // the members' values of the tile's rows
    float ls[kPsRows * 2 * kHeadPad];      // sampling on chip: (loc | scale) of the tile, row pitch 32
    float act[kPsRows * kHeadPad];         // ... and the sampled actions
};
static_assert(sizeof(PsPiLds) <= 2 * sizeof(PsQLds), "the policy phase reuses the critics' LDS");

struct PolicyStepArgs {
    MlpArgs q, pi;          // q: x0 = states, x1 = sampled actions, subset = the TWO members the objective samples (device
                            // indices, NULL = members 0 and 1); pi: x0 = states, eps, log_alpha, partial
    float* q_out;           // [E][N] the value table (statistics; the two sampled members' rows are written) or NULL
    // q.x1 == NULL: the action is SAMPLED HERE, tanh(loc + eps * scale) from a first run of the policy on the tile (its
    // weights are in registers anyway and go to LDS twice), instead of arriving from a policy-forward and a sampling
    // launch of its own; outputs for the statistics / the caller:
    float* a_out;           // [N][A]
    float* logp_out;        // [N]
    float* ls_out;          // [N][2A] (loc | scale) or NULL
};

// a 16-row input tile: kPsSlots of the 16 x 64 slots per thread
__device__ __forceinline__ void ps_fetch_tile(const StageScalars& q, int64_t row0, float (&v)[kPsSlots]) {
    const int in0 = q.in0, in1 = q.in1;
    const rsrc_t r0 = make_rsrc(q.x0);
    const rsrc_t r1 = make_rsrc(q.x1, in1 > 0 ? 0x7fffffffu : 0u);
#pragma unroll
    for (int u = 0; u < kPsSlots; ++u) {
        const int i = threadIdx.x + u * kPsThreads;
        const int r = i >> 6, c = i & 63;
        const uint32_t row = (uint32_t)row0 + (uint32_t)r;
        const bool live = (int64_t)row0 + r < q.N;
        const unsigned o0 = (live && c < in0) ? (row * (uint32_t)q.x0_rs + (uint32_t)c) * 4u : kOob;
        const unsigned o1 = (live && c >= in0 && c < in0 + in1) ? (row * (uint32_t)q.x1_rs + (uint32_t)(c - in0)) * 4u : kOob;
        v[u] = __builtin_bit_cast(float, buf_ld(r0, o0) | buf_ld(r1, o1));
    }
}
__device__ __forceinline__ void ps_put_tile(const float (&v)[kPsSlots], float* xs) {
#pragma unroll
    for (int u = 0; u < kPsSlots; ++u) {
        const int i = threadIdx.x + u * kPsThreads;
        xs[(i >> 6) * kP + (i & 63)] = v[u];
    }
}

__global__ __launch_bounds__(kPsThreads) void k_policy_step(const PolicyStepArgs a_by_value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const ASAC_KARG PolicyStepArgs& a = *static_cast<const ASAC_KARG PolicyStepArgs*>(kernarg_base());   // (asac_common.h)
    PsLds& L = *reinterpret_cast<PsLds*>(smem_raw);
    PsPiLds& P = *reinterpret_cast<PsPiLds*>(smem_raw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = wave >> 2, ct = wave & 3;           // critic phase: member, column tile
    const int col = ct * 16 + (lane & 15);
    const int64_t row0 = (int64_t)blockIdx.x * kPsRows;
    const int64_t N = a.pi.N;
    const int A = a.pi.d.head_cols[0];
    const int S = a.q.d.in0, K0q = a.q.d.in0 + a.q.d.in1, K0p = a.pi.d.in0;
    MLP_STAMP(0);

    // ---- staging: both critics -> LDS, the policy's parameters and this tile's noise -> registers ---------------
    // the two critics of the objective's subset (an ensemble of more members: the others get no gradient from
    // mean_b -min_{e in subset} q_e, so they are not evaluated at all); the order of the subset breaks ties
    const int e0 = a.q.subset ? a.q.subset[0] : 0, e1 = a.q.subset ? a.q.subset[1] : 1;
    const StageScalars s0 = stage_scalars<3>(a.q, e0), s1 = stage_scalars<3>(a.q, e1), sp = stage_scalars<3>(a.pi, 0);
    const bool sample_here = a.q.x1 == nullptr;
    float in_q[kPsSlots] = {}, in_pi[kPsSlots];
    if (!sample_here) ps_fetch_tile(s0, row0, in_q);
    ps_fetch_tile(sp, row0, in_pi);
    StagedNet<kPsThreads> r0, r1, rp;
    net_fetch_fixed<kPsThreads, 3>(s0, r0);
    net_fetch_fixed<kPsThreads, 3>(s1, r1);
    net_fetch_fixed<kPsThreads, 3>(sp, rp);
    float ev = 0.f, gl = 0.f;       // the sampling backward's per-(row, action dim) inputs
    {
        const int lrow = threadIdx.x / (A > 0 ? A : 1), d = threadIdx.x - lrow * A;
        if ((int)threadIdx.x < kPsRows * A && row0 + lrow < N) ev = a.pi.eps[(row0 + lrow) * A + d];
        gl = expf(*a.pi.log_alpha) * (1.f / (float)N);
    }
    if (sample_here) {
        // ---- the policy on the tile, once, for the action: weights -> LDS (they return from the registers later) ------
        net_put_fixed<kPsThreads, 3>(rp, P);
        ps_put_tile(in_pi, P.x[0]);
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (wave < 4) {
                const float* xin = P.x[l & 1];
                float* xout = P.x[(l & 1) ^ 1];
                const f32x4 acc = l > 0 ? gemm_tile(xin, P.w[l], kMaxW, 0, wave) : gemm_tile(xin, P.w[l], round4(K0p), 0, wave);
                const int pc = wave * 16 + (lane & 15);
                const float bias = P.bias[l][pc];
                const bool res = a.pi.d.residual[l] != 0;
                f32x2_g ya, yb, unused;
                gelu_parts2((f32x2_g){acc[0] + bias, acc[1] + bias}, ya, unused);
                gelu_parts2((f32x2_g){acc[2] + bias, acc[3] + bias}, yb, unused);
                const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r;
                    float y = yv[r];
                    if (res) y += xin[row * kP + pc];
                    xout[row * kP + pc] = y;
                }
            }
            __syncthreads();
        }
        if (wave < 4) {        // three layers: the last activations sit in P.x[1]; one of a lane's four rows per wave
            const f32x4 acc = gemm_tile(P.x[1], P.head, kMaxW, 0, 0);
            const int hc = lane & 15;
            const float mine = wave == 0 ? acc[0] : wave == 1 ? acc[1] : wave == 2 ? acc[2] : acc[3];
            const int lrow = 4 * (lane >> 4) + wave;
            const float v = head_value(a.pi.d, hc, mine + P.head_bias[hc]);
            L.ls[lrow * 2 * kHeadPad + hc] = v;
            if (a.ls_out && row0 + lrow < N && hc < 2 * A) a.ls_out[(row0 + lrow) * (2 * A) + hc] = v;
        }
        __syncthreads();
        if (threadIdx.x < kPsRows) {      // one lane per row: the sampling launch's device function (asac_squash.h)
            const int lrow = threadIdx.x;
            const int64_t r = row0 + lrow;
            float* arow = L.act + lrow * kHeadPad;
            if (r < N) {
                float lp;
                squash_sample_at(L.ls + lrow * 2 * kHeadPad, L.ls + lrow * 2 * kHeadPad + A, a.pi.eps + r * A, A, arow, &lp, nullptr);
                for (int d = 0; d < A; ++d) a.a_out[r * A + d] = arow[d];
                a.logp_out[r] = lp;
            } else {
                for (int d = 0; d < A; ++d) arow[d] = 0.f;
            }
        }
        __syncthreads();       // the policy is done with the LDS the critics' weights go to
        ps_put_tile(in_pi, L.q[0].xs[0]);
        ps_put_tile(in_pi, L.q[1].xs[0]);
    } else {
        ps_put_tile(in_q, L.q[0].xs[0]);
        ps_put_tile(in_q, L.q[1].xs[0]);
    }
    net_put_fixed<kPsThreads, 3>(r0, L.q[0]);
    net_put_fixed<kPsThreads, 3>(r1, L.q[1]);
    __syncthreads();
    if (sample_here) {         // the sampled actions into the critics' input tiles
        if ((int)threadIdx.x < kPsRows * A) {
            const int lrow = threadIdx.x / A, d = threadIdx.x - lrow * A;
            const float v = L.act[lrow * kHeadPad + d];
            L.q[0].xs[0][lrow * kP + S + d] = v;
            L.q[1].xs[0][lrow * kP + S + d] = v;
        }
        __syncthreads();
    }
    MLP_STAMP(1);

    // ---- critics forward (derivatives of the activations stay in registers) ------------------------------------
    const bool qwave = wave < 8;                 // (waves beyond the eight of the critic phase keep the barriers company)
    PsQLds& Q = L.q[qwave ? m : 0];
    f32x4 z[3];
    int cur = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (qwave) {
        const float* xin = Q.xs[cur];
        float* xout = Q.xs[cur ^ 1];
        const f32x4 acc = l > 0 ? gemm_tile(xin, Q.w[l], kMaxW, 0, ct) : gemm_tile(xin, Q.w[l], round4(K0q), 0, ct);
        const float bias = Q.bias[l][col];
        const bool res = a.q.d.residual[l] != 0;
        f32x2_g ya, yb, da, db;
        gelu_parts2((f32x2_g){acc[0] + bias, acc[1] + bias}, ya, da);
        gelu_parts2((f32x2_g){acc[2] + bias, acc[3] + bias}, yb, db);
        const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
        z[l] = f32x4{da.x, da.y, db.x, db.y};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r;
            float y = yv[r];
            if (res) y += xin[row * kP + col];
            xout[row * kP + col] = y;
        }
        }
        __syncthreads();
        cur ^= 1;
    }
    MLP_STAMP(2);
    if (qwave && ct == 0) {          // the scalar head: one wave per member
        const f32x4 raw = gemm_tile(Q.xs[cur], Q.head, kMaxW, 0, 0);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lrow = 4 * (lane >> 4) + r;
                const float qv = raw[r] + Q.head_bias[0];
                L.qv[m][lrow] = qv;
                if (a.q_out && row0 + lrow < N) a.q_out[(int64_t)(m ? e1 : e0) * N + row0 + lrow] = qv;
            }
        }
    }
    __syncthreads();
    MLP_STAMP(3);
    // d(mean_b -min_e q_e)/dq_m: -1/N on the rows where member m is the (first) arg-min (sac_base.py:1896-1903)
    float* qdelta = Q.xs[cur ^ 1];           // (the forward is done with both tiles; x_3 itself is not needed again)
    if (qwave) {
        const int t = threadIdx.x & 255;     // the member's 256 threads clear its 16 x 16 delta tile
        const int r = t >> 4, c = t & 15;
        float g = 0.f;
        if (c == 0 && row0 + r < N) {
            const int best = L.qv[1][r] < L.qv[0][r] ? 1 : 0;
            if (best == m) g = -1.f / (float)N;
        }
        qdelta[r * kP + c] = g;
    }
    __syncthreads();
    MLP_STAMP(4);
    // ---- critics backward to the action ---------------------------------------------------------------------------
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (qwave) g = gemm_tile_nt(qdelta, Q.head, kHeadPad, 0, ct);
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        __syncthreads();
        if (qwave) {
#pragma unroll
            for (int r = 0; r < 4; ++r) qdelta[(4 * (lane >> 4) + r) * kP + col] = g[r] * z[l][r];
        }
        __syncthreads();
        if (qwave) {
            f32x4 gin = gemm_tile_nt(qdelta, Q.w[l], kMaxW, 0, ct);
            if (a.q.d.residual[l]) gin += g;
            g = gin;
        }
    }
    if (qwave && col >= S && col < K0q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) L.ga[m][(4 * (lane >> 4) + r) * kHeadPad + (col - S)] = g[r];
    }
    __syncthreads();        // the critics are done with their LDS
    MLP_STAMP(5);

    // ---- the policy takes over: parameters registers -> LDS, forward recompute ------------------------------------
    net_put_fixed<kPsThreads, 3>(rp, P);
    ps_put_tile(in_pi, P.x[0]);
    if (threadIdx.x < kPsRows * kHeadPad) P.delta[(threadIdx.x >> 4) * kP + (threadIdx.x & 15)] = 0.f;
    __syncthreads();
    MLP_STAMP(6);
    f32x4 zp[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (wave < 4) {
            const float* xin = P.x[l];
            float* xout = P.x[l + 1];
            const f32x4 acc = l > 0 ? gemm_tile(xin, P.w[l], kMaxW, 0, wave) : gemm_tile(xin, P.w[l], round4(K0p), 0, wave);
            const int pc = wave * 16 + (lane & 15);
            const float bias = P.bias[l][pc];
            const bool res = a.pi.d.residual[l] != 0;
            f32x2_g ya, yb, da, db;
            gelu_parts2((f32x2_g){acc[0] + bias, acc[1] + bias}, ya, da);
            gelu_parts2((f32x2_g){acc[2] + bias, acc[3] + bias}, yb, db);
            const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
            zp[l] = f32x4{da.x, da.y, db.x, db.y};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                float y = yv[r];
                if (res) y += xin[row * kP + pc];
                xout[row * kP + pc] = y;
            }
        }
        __syncthreads();
    }
    MLP_STAMP(7);
    // raw head values -> (loc, scale) -> gradient of the rsample / tanh / log-prob chain (k_mlp_bwd's policy-sample mode)
    if (wave == 0) {
        const f32x4 raw = gemm_tile(P.x[3], P.head, kMaxW, 0, 0);
        const int hc = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) P.delta[(4 * (lane >> 4) + r) * kP + hc] = raw[r] + P.head_bias[hc];
    }
    __syncthreads();
    MLP_STAMP(8);
    if ((int)threadIdx.x < kPsRows * A) {
        const int lrow = threadIdx.x / A, d = threadIdx.x - lrow * A;
        const int64_t row = row0 + lrow;
        float g_loc = 0.f, g_scale = 0.f;
        const float raw_l = P.delta[lrow * kP + d], raw_s = P.delta[lrow * kP + A + d];
        if (row < N) {
            const float loc = head_value(a.pi.d, d, raw_l), sc = head_value(a.pi.d, A + d, raw_s);
            const float t = tanhf(loc + ev * sc);
            const float one_m = 1.f - t * t;
            float ga = 0.f;
            ga += L.ga[0][lrow * kHeadPad + d];
            ga += L.ga[1][lrow * kHeadPad + d];
            float gx = ga * one_m;
            if (one_m > 1e-2f) gx += gl * ((float)A * 2.f * t);
            g_loc = gx * head_deriv(a.pi.d, d, raw_l);
            g_scale = (gx * ev - gl / sc) * head_deriv(a.pi.d, A + d, raw_s);
        }
        P.delta[lrow * kP + d] = g_loc;
        P.delta[lrow * kP + A + d] = g_scale;
    }
    __syncthreads();
    MLP_STAMP(9);
    // ---- policy backward: parameter-gradient partials of this tile ------------------------------------------------
    float* part = a.pi.partial + (int64_t)blockIdx.x * a.pi.member_stride;
    ps_grad_weight(P.delta, 0, P.x[3], A, kMaxW, part + a.pi.d.head_w_off[0]);
    ps_grad_weight(P.delta, A, P.x[3], A, kMaxW, part + a.pi.d.head_w_off[1], 4);
    if (wave == 7) {        // the two head biases: 16 columns x 16 rows
        const int c = lane & 15, rq = (lane >> 4) * 4;
        float sb = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sb += P.delta[(rq + r) * kP + c];
        sb += __shfl_xor(sb, 16, 64);
        sb += __shfl_xor(sb, 32, 64);
        if (lane < 16) {
            if (c < A) part[a.pi.d.head_b_off[0] + c] = sb;
            else if (c < 2 * A) part[a.pi.d.head_b_off[1] + c - A] = sb;
        }
    }
    f32x4 gp = {0.f, 0.f, 0.f, 0.f};
    if (wave < 4) gp = gemm_tile_nt(P.delta, P.head, kHeadPad, 0, wave);
    MLP_STAMP(10);
#pragma unroll
    for (int l = 2; l >= 0; --l) {
        const int Kin = l == 0 ? K0p : kMaxW;
        __syncthreads();
        if (wave < 4) {
            const int pc = wave * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) P.delta[(4 * (lane >> 4) + r) * kP + pc] = gp[r] * zp[l][r];
        }
        __syncthreads();
        ps_grad_weight(P.delta, 0, P.x[l], kMaxW, Kin, part + a.pi.d.w_off[l]);
        grad_bias<kPsRows>(P.delta, 0, kMaxW, part + a.pi.d.b_off[l]);
        if (l > 0 && wave < 4) {
            f32x4 gin = gemm_tile_nt(P.delta, P.w[l], kMaxW, 0, wave);
            if (a.pi.d.residual[l]) gin += gp;
            gp = gin;
        }
        MLP_STAMP(11 + (2 - l));
    }
}

// ------------------------------------------------------------------------------------------------
// Policy forward -> sampling -> critic ensemble forward over the same rows in ONE launch: what the train step issues
// as asac_mlp_forward(_multi)[policy] -> asac_squash_multi -> asac_mlp_forward(_multi)[critics] for the return target
// (sac_base.py:1297-1466) and again for the TD error / new behaviour probabilities (2182-2245, 1159-1189).  The
// chain is row-local: a workgroup (16-row tile, member e) evaluates the policy on its tile, samples on chip, runs
// critic e on (state, sampled action).  Both networks' weights are staged together (one L2 round trip instead of
// two launches' worth), (loc | scale) and the sampled actions never travel through L2 between launches.  The policy
// part is repeated by the E workgroups of a tile (they run side by side on different CUs); member 0's workgroup
// writes the shared outputs.  Same MFMA chains and the very device functions of the separate launches: bit-identical.
// Kernel-argument form of a forward job on a STOCK network (three 64-wide blocks on <= 64 inputs: what the
// fixed-shape instantiations read): 32-bit offsets and strides, none of asac_mlp_desc_t's tables — 128 bytes instead of
// MlpFwdArgs' 272.  A launch's argument block is fetched on its critical path and every field that stays live costs a
// scalar register: with four MlpFwdArgs (1.9 KB with the sidecar list) k_pi_sample_q spilled 258 SGPRs and kept nine
// dwords of scratch per lane; in this form it does neither.  The kernel expands a job back into the MlpFwdArgs view the
// device functions are written against (the constants — three blocks, width 64 — fold at compile time).
struct StockJobArg {
    const float *P, *x0, *x1;
    float* out;
    int32_t member_stride, N, x0_rs, x0_ms, x1_rs, x1_ms, x0_sb, x0_T;
    int32_t in0, in1, h0, h1, hw0, hw1, hb0, hb1;
    int32_t w_off[3], b_off[3];
    int32_t residual_bits, head_transform;
};
static_assert(sizeof(StockJobArg) == 128, "StockJobArg");

template <typename JOB>
__device__ __forceinline__ MlpFwdArgs expand_stock(const JOB& j) {
    MlpFwdArgs a;
    a.d.in0 = j.in0, a.d.in1 = j.in1, a.d.n_blocks = 3;
#pragma unroll
    for (int l = 0; l < kMaxB; ++l) {
        a.d.width[l] = l < 3 ? kMaxW : 0;
        a.d.residual[l] = l < 3 ? (j.residual_bits >> l) & 1 : 0;
        a.d.w_off[l] = l < 3 ? j.w_off[l] : 0;
        a.d.b_off[l] = l < 3 ? j.b_off[l] : 0;
    }
    a.d.head_cols[0] = j.h0, a.d.head_cols[1] = j.h1;
    a.d.head_w_off[0] = j.hw0, a.d.head_w_off[1] = j.hw1;
    a.d.head_b_off[0] = j.hb0, a.d.head_b_off[1] = j.hb1;
    a.d.head_transform = j.head_transform, a.d.reserved_ = 0;
    a.params = j.P, a.member_stride = j.member_stride;
    a.x0 = j.x0, a.x0_rs = j.x0_rs, a.x0_ms = j.x0_ms;
    a.x1 = j.x1, a.x1_rs = j.x1_rs, a.x1_ms = j.x1_ms;
    a.x0_sb = j.x0_sb, a.x0_T = j.x0_T, a.pad_ = 0;
    a.N = j.N, a.out = j.out;
    return a;
}

struct PiQLds {
    float head[kHeadPad * kP];            // --- the layout of MlpLds<16> up to `w`: the extra plain forward jobs of the
    float bias[kMaxB][kMaxW];             //     launch run mlp_fwd_tiles<16> on the same memory
    float head_bias[kHeadPad];
    float xs[2][16 * kP];
    float w[3][kMaxW * kP];               // the policy
    float qw[3][kMaxW * kP];              // critic e
    float qhead[kHeadPad * kP];
    float qbias[3][kMaxW];
    float qhead_bias[kHeadPad];
    float ls[16 * 2 * kHeadPad];          // (loc | scale) of the tile, row pitch 32
    float act[16 * kHeadPad];             // sampled actions of the tile
};
static_assert(offsetof(PiQLds, w) == offsetof(MlpLds<16>, w), "PiQLds starts like MlpLds<16>");

struct PiQCritic {       // a view with the member names net_put_fixed expects
    float (&w)[3][kMaxW * kP];
    float (&head)[kHeadPad * kP];
    float (&bias)[3][kMaxW];
    float (&head_bias)[kHeadPad];
};

// (32-bit strides: pi_q_job_ok bounds every offset of the launch below 2^29 floats)
struct StoredProbArg {
    const float* action;
    float* out;
    int32_t T, a_sb, a_st, a_off, p_sb, p_st, p_off, pad_;
};
struct PiQLaunch {
    StockJobArg pi, q;                      // the fused job: pi.out [N][2A] or NULL, q.out [E][N]
    const float *eps, *eps2;
    float *a_out, *logp_out, *a2_out, *logp2_out;
    StoredProbArg sp;
    int32_t E, tile_groups, blocks, T, t2;
    // plain forward jobs riding in the same launch (their workgroups follow the fused job's)
    int32_t x_n, x_blocks;
    int32_t x_E[ASAC_MLP_MAX_JOBS], x_first_block[ASAC_MLP_MAX_JOBS], x_tile_stride[ASAC_MLP_MAX_JOBS];
    StockJobArg x_job[ASAC_MLP_MAX_JOBS];
};


// Kernel arguments are read WHERE THEY ARE USED, through the kernarg segment's own address space: a by-value struct
// parameter is loaded whole in the kernel's entry block (k_pi_sample_q: ~130 scalar registers parked in VGPR lanes by
// `v_writelane` before the first weight load was issued); a reference into the segment makes each field an ordinary
// scalar load at its use — the staging phase fetches what it needs in one batch (stage_scalars_stock), the sampling
// pointers arrive while the policy runs.
__device__ __forceinline__ StageScalars stage_scalars_stock(const ASAC_KARG StockJobArg& j, int e) {
    StageScalars q;
    q.P = j.P + e * j.member_stride;
    q.x0 = j.x0, q.x1 = j.in1 > 0 ? j.x1 : j.x0;
    q.N = j.N, q.x0_rs = j.x0_rs, q.x0_ms = j.x0_ms, q.x1_rs = j.x1_rs, q.x1_ms = j.x1_ms, q.x0_sb = j.x0_sb;
    q.in0 = j.in0, q.in1 = j.in1, q.x0_T = j.x0_T;
    q.h0 = j.h0, q.h1 = j.h1;
    q.hw0 = j.hw0, q.hw1 = j.hw1, q.hb0 = j.hb0, q.hb1 = j.hb1;
#pragma unroll
    for (int l = 0; l < 3; ++l) q.w_off[l] = j.w_off[l], q.b_off[l] = j.b_off[l];
    ASAC_PIN(q.P); ASAC_PIN(q.x0); ASAC_PIN(q.x1);
    ASAC_PIN(q.N); ASAC_PIN(q.x0_rs); ASAC_PIN(q.x0_ms); ASAC_PIN(q.x1_rs); ASAC_PIN(q.x1_ms); ASAC_PIN(q.x0_sb);
    ASAC_PIN(q.in0); ASAC_PIN(q.in1); ASAC_PIN(q.x0_T); ASAC_PIN(q.h0); ASAC_PIN(q.h1);
    ASAC_PIN(q.hw0); ASAC_PIN(q.hw1); ASAC_PIN(q.hb0); ASAC_PIN(q.hb1);
#pragma unroll
    for (int l = 0; l < 3; ++l) { ASAC_PIN(q.w_off[l]); ASAC_PIN(q.b_off[l]); }
    return q;
}

// ... of a second network on the SAME rows as `rows` (x0 addressing shared; x1 is never fetched: it is formed on chip)
__device__ __forceinline__ StageScalars stage_scalars_stock_like(const ASAC_KARG StockJobArg& j, int e, const StageScalars& rows) {
    StageScalars q = rows;
    q.P = j.P + e * j.member_stride;
    q.in1 = j.in1;
    q.h0 = j.h0, q.h1 = j.h1;
    q.hw0 = j.hw0, q.hw1 = j.hw1, q.hb0 = j.hb0, q.hb1 = j.hb1;
#pragma unroll
    for (int l = 0; l < 3; ++l) q.w_off[l] = j.w_off[l], q.b_off[l] = j.b_off[l];
    ASAC_PIN(q.P); ASAC_PIN(q.in1); ASAC_PIN(q.h0); ASAC_PIN(q.h1);
    ASAC_PIN(q.hw0); ASAC_PIN(q.hw1); ASAC_PIN(q.hb0); ASAC_PIN(q.hb1);
#pragma unroll
    for (int l = 0; l < 3; ++l) { ASAC_PIN(q.w_off[l]); ASAC_PIN(q.b_off[l]); }
    return q;
}

// the Gaussian policy head (policy.py:170-172): location columns 5 tanh(x / 5), scale columns exp(clamp(x, -20, 0.5))
__device__ __forceinline__ float gauss_head_value(bool location, float raw) {
    return location ? tanhf(raw / 5.f) * 5.f : expf(fminf(fmaxf(raw, -20.f), 0.5f));
}

template <bool WINDOW>
__device__ __forceinline__ void pi_q_tiles(const ASAC_KARG PiQLaunch& a, PiQLds& L) {
    // eight waves: the staging (loads and LDS writes of two networks and the input tile) is dealt over 512 threads —
    // 1.5 us instead of 2.5; the forward chains have four column tiles: waves 4..7 only keep the barriers company there
    constexpr int THREADS = kPiQThreads;
    constexpr int SLOTS = 1024 / kPiQThreads;      // of the 16 x 64 input tile per thread
    const bool comp = (threadIdx.x >> 6) < 4;
    const int e = (int)blockIdx.x % a.E, group = (int)blockIdx.x / a.E;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = wave * 16 + (lane & 15);
    const int A = a.pi.h0, S = a.q.in0;
    const int K0p = a.pi.in0, K0q = a.q.in0 + a.q.in1;
    const int N = a.pi.N;                 // (rows, strides and offsets of this launch are below 2^29: 32-bit index arithmetic)
    const int n_tiles = (int)((N + 15) / 16);
    const bool writer = e == 0;
    MLP_STAMP(0);

    // ---- staging: the policy and critic e, one round trip ----------------------------------------------------------
    // (the critics read the policy's rows — pi_q_job_ok: their row addressing shares the policy's registers)
    const StageScalars sp = stage_scalars_stock(a.pi, 0);
    const StageScalars sq = stage_scalars_stock_like(a.q, e, sp);
    float in_lo[SLOTS];
    fetch_input_tile_fixed<THREADS, WINDOW, SLOTS>(sp, 0, (int64_t)group * 16, in_lo);
    StagedNet<THREADS> rp, rq;
    net_fetch_fixed<THREADS, 3>(sp, rp);
    net_fetch_fixed<THREADS, 3>(sq, rq);
    put_input_tile<THREADS, SLOTS>(in_lo, L.xs[0]);
    net_put_fixed<THREADS, 3>(rp, L);
    // (critic e's weights stay in registers until the policy has run: its 36 KB are still travelling when the policy's
    // have landed — memory returns in order — and nothing needs them before the first critic layer)
    PiQCritic C{L.qw, L.qhead, L.qbias, L.qhead_bias};
    __syncthreads();
    MLP_STAMP(1);

    for (int tile = group; tile < n_tiles; tile += a.tile_groups) {
        const int row0 = tile * 16;
        if (tile != group) {       // (later tiles of a looping workgroup: the first one arrived with the weights)
            fetch_input_tile_fixed<THREADS, WINDOW, SLOTS>(sp, 0, row0, in_lo);
            __syncthreads();
            put_input_tile<THREADS, SLOTS>(in_lo, L.xs[0]);
            __syncthreads();
        }
        // ---- policy forward ----------------------------------------------------------------------------------------
        int cur = 0;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (comp) {
                const float* xin = L.xs[cur];
                float* xout = L.xs[cur ^ 1];
                const f32x4 acc = l > 0 ? gemm_tile(xin, L.w[l], kMaxW, 0, wave) : gemm_tile(xin, L.w[l], round4(K0p), 0, wave);
                const float bias = L.bias[l][col];
                const bool res = (a.pi.residual_bits >> l) & 1;
                f32x2_g ya, yb, unused;
                gelu_parts2((f32x2_g){acc[0] + bias, acc[1] + bias}, ya, unused);
                gelu_parts2((f32x2_g){acc[2] + bias, acc[3] + bias}, yb, unused);
                const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r;
                    float y = yv[r];
                    if (res) y += xin[row * kP + col];
                    xout[row * kP + col] = y;
                }
            }
            __syncthreads();
            cur ^= 1;
        }
        MLP_STAMP(2);
        if (comp) {
            // the head tile is 16 x 16: every wave forms it (the same MFMA chain: the same values) and finishes ONE of
            // the four rows a lane holds — the head transform (tanh on the location columns, exp on the scale columns,
            // both branches taken by every wave) runs once per lane instead of four times on a single wave
            const f32x4 acc = gemm_tile(L.xs[cur], L.head, kMaxW, 0, 0);
            const int hc = lane & 15;
            const float mine = wave == 0 ? acc[0] : wave == 1 ? acc[1] : wave == 2 ? acc[2] : acc[3];
            const int lrow = 4 * (lane >> 4) + wave;
            const float v = gauss_head_value(hc < A, mine + L.head_bias[hc]);
            L.ls[lrow * 2 * kHeadPad + hc] = v;
            if (writer && a.pi.out && row0 + lrow < N && hc < 2 * A) a.pi.out[(row0 + lrow) * (2 * A) + hc] = v;
        }
        __syncthreads();
        MLP_STAMP(3);
        // ---- sampling: the elementwise launches' arithmetic (asac_squash.h), spread over lanes ---------------------------
        // Their one-lane-per-row loops cost ~4 us of dependent tanh / log / atanh / exp chains here (16 rows = 16 lanes).
        // Same values, same summation order: the per-(row, d) transcendentals run on 16 A lanes — main sample on wave
        // 0, stored-action probabilities on wave 1, the second sample on wave 2 — and land in LDS; one lane per row
        // then forms the sums / products over d in d order.
        {
            float* scr = L.xs[1];                  // (the policy's last activations are dead; xs[0] is being refilled)
            const int job = wave;                  // 0 main, 1 stored-action probabilities, 2 second sample
            const bool job_on = job == 0 || (writer && ((job == 1 && a.sp.action) || (job == 2 && a.eps2)));
            float* s0 = scr + job * 256;           // [16][8] per-element term 0 | [16][8] term 1
            float* s1 = s0 + 128;
            if (job < 3 && job_on) {
                for (int it = lane; it < 16 * A; it += 64) {
                    const int lrow = it / A, d = it - lrow * A;
                    const int r = row0 + lrow;
                    if (r >= N) continue;
                    const float l = L.ls[lrow * 2 * kHeadPad + d], sc = L.ls[lrow * 2 * kHeadPad + A + d];
                    if (job == 1) {
                        const int spT = a.sp.T;
                        const int sb = (int)((unsigned)r / (unsigned)spT), st = r - sb * spT;
                        const float av = a.sp.action[sb * a.sp.a_sb + st * a.sp.a_st + a.sp.a_off + d];
                        const float x = atanhf(fminf(fmaxf(av, -0.999f), 0.999f));
                        s0[lrow * 8 + d] = squash_jac(x);
                        s1[lrow * 8 + d] = expf(normal_log_prob(x, l, sc));
                    } else {
                        float ev;
                        const int smp = (int)((unsigned)r / (unsigned)a.T);
                        if (job == 0) {
                            ev = a.eps[r * A + d];
                        } else {
                            if (r - smp * a.T != a.t2) continue;
                            ev = a.eps2[smp * A + d];
                        }
                        const float x = l + ev * sc;
                        const float t = tanhf(x);
                        s0[lrow * 8 + d] = logf(fmaxf(1.f - t * t, kSquashFloor));
                        s1[lrow * 8 + d] = normal_log_prob(x, l, sc);
                        if (job == 0) {
                            L.act[lrow * kHeadPad + d] = t;
                            if (writer) a.a_out[r * A + d] = t;
                        } else {
                            a.a2_out[smp * A + d] = t;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 16) {
                    const int lrow = lane;
                    const int r = row0 + lrow;
                    if (r < N) {
                        if (job == 1) {
                            float jac = 1.f;
                            for (int d = 0; d < A; ++d) jac *= s0[lrow * 8 + d];
                            const int spT = a.sp.T;
                            const int sb = (int)((unsigned)r / (unsigned)spT), st = r - sb * spT;
                            float* out = a.sp.out + (sb * a.sp.p_sb + st * a.sp.p_st + a.sp.p_off);
                            for (int d = 0; d < A; ++d) out[d] = s1[lrow * 8 + d] / jac;
                        } else {
                            const int smp = (int)((unsigned)r / (unsigned)a.T);
                            if (job == 0 || r - smp * a.T == a.t2) {
                                float corr = 0.f;
                                for (int d = 0; d < A; ++d) corr += s0[lrow * 8 + d];
                                float lp = 0.f;
                                for (int d = 0; d < A; ++d) {
                                    float v = s1[lrow * 8 + d] - corr;
                                    if (v == INFINITY) v = 0.f;
                                    lp += v;
                                }
                                if (job == 2) a.logp2_out[smp] = lp;
                                else if (writer) a.logp_out[r] = lp;
                            }
                        }
                    }
                }
            }
            if (job == 0 && lane < 16 && row0 + lane >= N)
                for (int d = 0; d < A; ++d) L.act[lane * kHeadPad + d] = 0.f;
        }
        // ---- critic e on (state, sampled action) ---------------------------------------------------------------------
        if (tile == group) net_put_fixed<THREADS, 3>(rq, C);
        put_input_tile<THREADS, SLOTS>(in_lo, L.xs[0]);         // the states again (columns >= S are zero)
        __syncthreads();
        MLP_STAMP(4);
        if ((int)threadIdx.x < 16 * A) {
            const int lrow = threadIdx.x / A, d = threadIdx.x - lrow * A;
            L.xs[0][lrow * kP + S + d] = L.act[lrow * kHeadPad + d];
        }
        __syncthreads();
        MLP_STAMP(5);
        cur = 0;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (comp) {
                const float* xin = L.xs[cur];
                float* xout = L.xs[cur ^ 1];
                const f32x4 acc = l > 0 ? gemm_tile(xin, L.qw[l], kMaxW, 0, wave) : gemm_tile(xin, L.qw[l], round4(K0q), 0, wave);
                const float bias = L.qbias[l][col];
                const bool res = (a.q.residual_bits >> l) & 1;
                f32x2_g ya, yb, unused;
                gelu_parts2((f32x2_g){acc[0] + bias, acc[1] + bias}, ya, unused);
                gelu_parts2((f32x2_g){acc[2] + bias, acc[3] + bias}, yb, unused);
                const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r;
                    float y = yv[r];
                    if (res) y += xin[row * kP + col];
                    xout[row * kP + col] = y;
                }
            }
            __syncthreads();
            cur ^= 1;
        }
        MLP_STAMP(6);
        if (wave == 0) {
            const f32x4 acc = gemm_tile(L.xs[cur], L.qhead, kMaxW, 0, 0);
            if ((lane & 15) == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 4 * (lane >> 4) + r;
                    if (row < N) a.q.out[(int64_t)e * N + row] = acc[r] + L.qhead_bias[0];
                }
            }
        }
        MLP_STAMP(7);
    }
}

static_assert(sizeof(PiQLaunch) <= 672, "kernel arguments of k_pi_sample_q (was 1 240 bytes as four MlpFwdArgs)");
static_assert(sizeof(PiQLaunch) + sizeof(SidecarsDev) <= 4096, "kernel arguments of k_pi_sample_q");

template <int NSC>
__global__ __launch_bounds__(kPiQThreads) void k_pi_sample_q(const PiQLaunch m_by_value, const SidecarsT<NSC> sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // (`m_by_value` sits at offset 0 of the kernarg segment; it is read through `m`, field by field, where used)
    const ASAC_KARG PiQLaunch& m = *static_cast<const ASAC_KARG PiQLaunch*>(kernarg_base());
    const int blk = (int)blockIdx.x;
    const int fused_blocks = m.blocks;
    // (the fused job's workgroups run eight waves; the plain forward jobs and the sidecars riding along are written
    // for four: their workgroups let the other four go — a finished wave no longer counts at the barriers)
    if (blk >= fused_blocks && threadIdx.x >= 256) return;
    if (blk < fused_blocks) {
        PiQLds& L = *reinterpret_cast<PiQLds*>(smem_raw);
        if (m.pi.x0_T > 0) pi_q_tiles<true>(m, L);
        else pi_q_tiles<false>(m, L);
        return;
    }
    const int xb = blk - fused_blocks;
    if (xb >= m.x_blocks) {
        sidecar_run<NSC, true>(sc, xb - m.x_blocks, reinterpret_cast<float*>(smem_raw));
        return;
    }
    int k = 0;
#pragma unroll
    for (int q = 1; q < ASAC_MLP_MAX_JOBS; ++q)
        if (q < m.x_n && xb >= m.x_first_block[q]) k = q;
    const int local = xb - m.x_first_block[k];
    const int E = m.x_E[k];
    MlpLds<16>& L = *reinterpret_cast<MlpLds<16>*>(smem_raw);
    const MlpFwdArgs job = expand_stock(m.x_job[k]);
    if (job.x0_T > 0)
        mlp_fwd_tiles<16, true, false, 3>(job, local % E, local / E, m.x_tile_stride[k], L);
    else
        mlp_fwd_tiles<16, false, false, 3>(job, local % E, local / E, m.x_tile_stride[k], L);
}

// grad[e*stride + i] (+)= sum_tiles partial[tile][e][i]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_mlp_reduce_partials(const float* __restrict__ partial, int tiles, int E,
                                                             int64_t member_stride, int64_t used,
                                                             float* __restrict__ grad, int accumulate,
                                                             const float* __restrict__ loss_partial,
                                                             float* __restrict__ loss_out, float inv_n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (loss_partial && i == 0) {
        float l = 0.f;
        for (int t = 0; t < tiles; ++t) l += loss_partial[(int64_t)t * E + e];
        loss_out[e] = l * inv_n;
    }
    if (i >= used) return;
    // tile order, loads issued eight at a time (see optim.hip sum_tiles: the plain loop waits per tile)
    float s = 0.f;
    for (int t0 = 0; t0 < tiles; t0 += 8) {
        float v[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int t = t0 + w < tiles ? t0 + w : tiles - 1;
            v[w] = partial[((int64_t)t * E + e) * member_stride + i];
        }
#pragma unroll
        for (int w = 0; w < 8; ++w) s += t0 + w < tiles ? v[w] : 0.f;
    }
    grad[e * member_stride + i] = accumulate ? grad[e * member_stride + i] + s : s;
}

// the same for many tiles (long row sets, e.g. an encoder head over every frame of the sampled windows): 64
// parameters per workgroup, 16 waves each summing a contiguous slice of the tiles with 8 loads in flight, then the
// slice sums in order — fixed order for a given launch shape
constexpr int kReduceSlices = 16;
constexpr int kSlicedFrom = 64;     // tiles; below, the sequential kernel (whose order asac_adam_step_partials shares)
__global__ __launch_bounds__(64 * kReduceSlices) void k_mlp_reduce_partials_sliced(
    const float* __restrict__ partial, int tiles, int E, int64_t member_stride, int64_t used, float* __restrict__ grad,
    int accumulate, const float* __restrict__ loss_partial, float* __restrict__ loss_out, float inv_n) {
    __shared__ float part[kReduceSlices][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6, e = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    if (loss_partial && blockIdx.x == 0 && threadIdx.x == 0) {
        float l = 0.f;
        for (int t = 0; t < tiles; ++t) l += loss_partial[(int64_t)t * E + e];
        loss_out[e] = l * inv_n;
    }
    const int per = (tiles + kReduceSlices - 1) / kReduceSlices;
    const int lo = sl * per, hi = min(lo + per, tiles);
    float s = 0.f;
    if (i < used) {
        int t = lo;
        for (; t + 8 <= hi; t += 8) {
            float v[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) v[w] = partial[((int64_t)(t + w) * E + e) * member_stride + i];
#pragma unroll
            for (int w = 0; w < 8; ++w) s += v[w];
        }
        for (; t < hi; ++t) s += partial[((int64_t)t * E + e) * member_stride + i];
    }
    part[sl][lane] = s;
    __syncthreads();
    if (sl != 0 || i >= used) return;
    s = 0.f;
#pragma unroll
    for (int w = 0; w < kReduceSlices; ++w) s += part[w][lane];
    grad[e * member_stride + i] = accumulate ? grad[e * member_stride + i] + s : s;
}

static bool desc_ok(const asac_mlp_desc_t& d) {
    if (d.n_blocks < 1 || d.n_blocks > kMaxB) return false;
    const int K0 = d.in0 + d.in1;
    if (d.in0 <= 0 || d.in1 < 0 || K0 > 2 * kMaxW) return false;
    if (K0 > kMaxW && d.n_blocks >= kMaxB) return false;      // the wide first layer borrows the spare tiles
    int prev = K0;
    for (int l = 0; l < d.n_blocks; ++l) {
        if (d.width[l] <= 0 || d.width[l] > kMaxW) return false;
        if (d.residual[l] && d.width[l] != prev) return false;
        prev = d.width[l];
    }
    const int O = d.head_cols[0] + d.head_cols[1];
    return d.head_cols[0] > 0 && d.head_cols[1] >= 0 && O <= kHeadPad;
}

static int set_lds_limit(const void* fn, size_t bytes, bool& done, const char* where) {
    if (done) return 0;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err != hipSuccess) {
        set_error(err, where);
        return (int)err;
    }
    done = true;
    return 0;
}

static MlpArgs make_args(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride,
                         const float* x0, int64_t x0_rs, int64_t x0_ms, const float* x1, int64_t x1_rs,
                         int64_t x1_ms, int64_t N) {
    MlpArgs a{};
    a.d = *desc;
    a.params = params;
    a.member_stride = member_stride;
    a.x0 = x0; a.x0_rs = x0_rs; a.x0_ms = x0_ms;
    a.x1 = x1; a.x1_rs = x1_rs; a.x1_ms = x1_ms;
    a.N = N;
    return a;
}

}  // namespace asac

using namespace asac;

// the fixed-shape instantiation (NB = 3: the reference's stock networks) applies when the description is exactly
// three 64-wide blocks on <= 64 inputs and every 64 x 64 weight sits on a 16-byte boundary
static bool stock3(const asac_mlp_desc_t& d, const float* params, int64_t member_stride) {
    if (d.n_blocks != 3 || d.in0 + d.in1 > kMaxW) return false;
    for (int l = 0; l < 3; ++l)
        if (d.width[l] != kMaxW) return false;
    if ((reinterpret_cast<uintptr_t>(params) & 15) || (member_stride & 3) || (d.w_off[1] & 3) || (d.w_off[2] & 3)) return false;
    return true;
}
// ... and its 32-bit byte offsets need every row of the inputs within 2 GiB of the base
static bool offsets32(const MlpFwdArgs& a) {
    const int64_t lim = 0x7fffffffLL / 4;
    const int64_t span0 = a.x0_T > 0 ? (a.N / a.x0_T + 1) * a.x0_sb : a.N * a.x0_rs;
    return span0 + kMaxW < lim && a.N * a.x1_rs + kMaxW < lim && a.member_stride < lim;
}

template <int TM>
static int launch_forward(const asac_mlp_desc_t* desc, const MlpFwdArgs& a, int E, int64_t N, hipStream_t s) {
    const bool wide = desc->in0 + desc->in1 > kMaxW;
    const size_t lds = mlp_fwd_lds_bytes(desc->n_blocks, wide, TM);
    const dim3 grid((unsigned)mlp_tile_groups(N, E, lds <= 80 * 1024 ? 2 : 1, TM), (unsigned)E);
    static bool attr_done = false, attr_wide = false, attr_stock = false;
    if (wide) {
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(k_mlp_fwd<TM, true, 0>),
                                   mlp_fwd_lds_bytes(kMaxB - 1, true, TM), attr_wide, "asac_mlp_forward: hipFuncSetAttribute"))
            return rc;
        ASAC_LAUNCH((k_mlp_fwd<TM, true, 0>), grid, dim3(threads_of<TM>()), lds, s, a);
    } else if (stock3(*desc, a.params, a.member_stride) && offsets32(a)) {
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(k_mlp_fwd<TM, false, 3>), sizeof(MlpLds<TM>), attr_stock,
                                   "asac_mlp_forward: hipFuncSetAttribute"))
            return rc;
        ASAC_LAUNCH((k_mlp_fwd<TM, false, 3>), grid, dim3(threads_of<TM>()), lds, s, a);
    } else {
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(k_mlp_fwd<TM, false, 0>), sizeof(MlpLds<TM>), attr_done,
                                   "asac_mlp_forward: hipFuncSetAttribute"))
            return rc;
        ASAC_LAUNCH((k_mlp_fwd<TM, false, 0>), grid, dim3(threads_of<TM>()), lds, s, a);
    }
    return finish_launch("asac_mlp_forward");
}

template <int TM, int NB>
static int launch_forward_multi(const asac_mlp_job_t* jobs, int n_jobs, const SidecarsDev& sc, hipStream_t s,
                                const SampleEpis* epis = nullptr) {
    static bool attr_done = false, attr_done_epi = false;
    if (int rc = epis ? set_lds_limit(reinterpret_cast<const void*>(k_mlp_fwd_multi_sampled<TM, NB>), sizeof(MlpLds<TM>),
                                      attr_done_epi, "asac_mlp_forward_multi_sampled: hipFuncSetAttribute")
                      : set_lds_limit(reinterpret_cast<const void*>(k_mlp_fwd_multi<TM, NB>), sizeof(MlpLds<TM>), attr_done,
                                      "asac_mlp_forward_multi: hipFuncSetAttribute"))
        return rc;
    MlpMultiArgs m{};
    m.n = n_jobs;
    int blocks = 0;
    size_t lds = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const size_t need = mlp_fwd_lds_bytes(jobs[k].desc->n_blocks, false, TM);
        lds = need > lds ? need : lds;
    }
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    for (int k = 0; k < n_jobs; ++k) {
        const asac_mlp_job_t& j = jobs[k];
        m.job[k] = make_args(j.desc, j.params, j.member_stride, j.x0, j.x0_row_stride, j.x0_member_stride, j.x1,
                             j.x1_row_stride, j.x1_member_stride, j.N);
        m.job[k].x0_T = j.x0_window_T;
        m.job[k].x0_sb = j.x0_sample_stride;
        m.job[k].out = j.out;
        m.E[k] = j.E;
        m.first_block[k] = blocks;
        m.tile_stride[k] = mlp_tile_groups(j.N, j.E, per_cu, TM);
        blocks += m.tile_stride[k] * j.E;
    }
    m.blocks = blocks;
    const SidecarsDev none{};
    for (int rep = 0; rep < g_launch_repeat; ++rep) {      // (repeat knob: only the last repetition carries the sidecars)
        const bool last = rep == g_launch_repeat - 1;
        if (epis)
            hipLaunchKernelGGL((k_mlp_fwd_multi_sampled<TM, NB>), dim3((unsigned)(blocks + (last ? sc.blocks : 0))),
                               dim3(threads_of<TM>()), lds, s, m, *epis, last ? sc : none);
        else
            hipLaunchKernelGGL((k_mlp_fwd_multi<TM, NB>), dim3((unsigned)(blocks + (last ? sc.blocks : 0))),
                               dim3(threads_of<TM>()), lds, s, m, last ? sc : none);
    }
    return finish_launch(epis ? "asac_mlp_forward_multi_sampled" : "asac_mlp_forward_multi");
}

// waves of k_mlp_bwd on the stock networks' 16-row tiles (A/B builds: -DASAC_BWD_WAVES=8; measured on cfg2, same box,
// alternating libraries: 4 -> 8 waves +1.7 %, 8 -> 16 waves +0.9 %)
#ifndef ASAC_BWD_WAVES
#define ASAC_BWD_WAVES 16
#endif
template <int TM>
static int launch_backward(const char* where, const asac_mlp_desc_t* desc, MlpArgs& a, int E, int tiles, hipStream_t s,
                           const RetIn<true>* ret = nullptr) {
    static bool attr_done = false, attr_wide = false, attr_stock = false, attr_ret = false;
    const bool wide = desc->in0 + desc->in1 > kMaxW;
    const bool stock = !wide && stock3(*desc, a.params, a.member_stride) && offsets32(a);
    if (ret) {             // (asac_mlp_backward_qloss_return_ok has said yes: stock network, the tile's steps fit)
        const size_t lds = sizeof(MlpBwdLds<TM>) + (size_t)(2 * ((ret->v.n + 1) | 1) + 2) * TM * sizeof(float);
        constexpr int w8 = TM == 16 ? ASAC_BWD_WAVES : 0;            // (16-row tiles of the stock networks: see k_mlp_bwd)
        // one thread per (row, step) of the tile: the launch's own thread count bounds n (16 waves: n <= 64)
        if (!stock || TM * ret->v.n > (w8 ? 64 * w8 : threads_of<TM>()) || lds > 128 * 1024) return bad_arg(where);
        if (int rc = set_lds_limit(reinterpret_cast<const void*>(k_mlp_bwd<TM, false, 3, true, w8>), 128 * 1024, attr_ret, where))
            return rc;
        ASAC_LAUNCH((k_mlp_bwd<TM, false, 3, true, w8>), dim3(tiles, E), dim3(w8 ? 64 * w8 : threads_of<TM>()), lds, s, a, *ret);
        return 0;
    }
    if (int rc = wide    ? set_lds_limit(reinterpret_cast<const void*>(k_mlp_bwd<TM, true, 0>), sizeof(MlpBwdLds<TM>), attr_wide, where)
                 : stock ? set_lds_limit(reinterpret_cast<const void*>(k_mlp_bwd<TM, false, 3, false, TM == 16 ? ASAC_BWD_WAVES : 0>), sizeof(MlpBwdLds<TM>), attr_stock, where)
                         : set_lds_limit(reinterpret_cast<const void*>(k_mlp_bwd<TM, false, 0>), sizeof(MlpBwdLds<TM>), attr_done, where))
        return rc;
    if (wide)
        ASAC_LAUNCH((k_mlp_bwd<TM, true, 0>), dim3(tiles, E), dim3(threads_of<TM>()), sizeof(MlpBwdLds<TM>), s, a, RetIn<false>{});
    else if (stock)
        ASAC_LAUNCH((k_mlp_bwd<TM, false, 3, false, TM == 16 ? ASAC_BWD_WAVES : 0>), dim3(tiles, E),
                    dim3(TM == 16 ? 64 * ASAC_BWD_WAVES : threads_of<TM>()),
                    sizeof(MlpBwdLds<TM>), s, a, RetIn<false>{});
    else
        ASAC_LAUNCH((k_mlp_bwd<TM, false, 0>), dim3(tiles, E), dim3(threads_of<TM>()), sizeof(MlpBwdLds<TM>), s, a, RetIn<false>{});
    return 0;
}

static int mlp_backward_common(const char* where, const asac_mlp_desc_t* desc, MlpArgs& a, int E, int64_t N,
                               int64_t member_stride, float* grad_params, float* workspace, int reduce_mode,
                               float* loss_out, hipStream_t s, const RetIn<true>* ret = nullptr) {
    const int tiles = (int)asac_mlp_backward_tiles(N, E);
    a.partial = grad_params ? workspace : nullptr;
    a.loss_partial = workspace ? workspace + (int64_t)tiles * E * member_stride : nullptr;
    if (int rc = mlp_tile_rows(N, E) == 16 ? launch_backward<16>(where, desc, a, E, tiles, s, ret)
                                           : launch_backward<32>(where, desc, a, E, tiles, s, ret))
        return rc;
    if (grad_params && reduce_mode != ASAC_MLP_REDUCE_DEFER) {
        const int64_t used = asac_mlp_param_extent(desc);
        // launched once (not under the repeat knob: it may accumulate)
        if (tiles >= kSlicedFrom)
            hipLaunchKernelGGL(k_mlp_reduce_partials_sliced, dim3((unsigned)((used + 63) / 64), (unsigned)E),
                               dim3(64 * kReduceSlices), 0, s, workspace, tiles, E, member_stride, used, grad_params,
                               reduce_mode == ASAC_MLP_REDUCE_ACCUMULATE ? 1 : 0, loss_out ? a.loss_partial : nullptr,
                               loss_out, 1.f / (float)N);
        else
            hipLaunchKernelGGL(k_mlp_reduce_partials, dim3((unsigned)((used + 255) / 256), (unsigned)E), dim3(256), 0, s,
                               workspace, tiles, E, member_stride, used, grad_params,
                               reduce_mode == ASAC_MLP_REDUCE_ACCUMULATE ? 1 : 0, loss_out ? a.loss_partial : nullptr,
                               loss_out, 1.f / (float)N);
    }
    return finish_launch(where);
}

extern "C" {

int asac_mlp_forward(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride, int E,
                     const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                     const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                     float* out, void* stream) {
    if (!desc || !desc_ok(*desc) || E <= 0 || N <= 0 || !x0 || (desc->in1 > 0 && !x1) || !out)
        return bad_arg("asac_mlp_forward");
    MlpArgs a = make_args(desc, params, member_stride, x0, x0_row_stride, x0_member_stride, x1, x1_row_stride,
                          x1_member_stride, N);
    a.out = out;
    return mlp_tile_rows(N, E) == 16 ? launch_forward<16>(desc, a, E, N, as_stream(stream))
                                     : launch_forward<32>(desc, a, E, N, as_stream(stream));
}

int asac_mlp_forward_multi(const asac_mlp_job_t* jobs, int n_jobs, const asac_sidecar_t* sidecars_host, int n_sidecars,
                           void* stream) {
    if (!jobs || n_jobs < 1 || n_jobs > ASAC_MLP_MAX_JOBS) return bad_arg("asac_mlp_forward_multi");
    SidecarsDev sc{};
    if (sidecars_prepare(sidecars_host, n_sidecars, sc)) return bad_arg("asac_mlp_forward_multi: sidecar");
    int64_t groups16 = 0;       // workgroups of the whole launch with 16-row tiles
    bool all_stock = true, any_wide = false;
    for (int k = 0; k < n_jobs; ++k)
        if (jobs[k].desc && desc_ok(*jobs[k].desc) && jobs[k].desc->in0 + jobs[k].desc->in1 > kMaxW) any_wide = true;
    if (any_wide) {
        // a first layer wider than 64 inputs (critics on a 64-wide state + the action): the jobs go one launch each through
        // the wide instantiation of the single-network forward; no window addressing, no sidecars there
        if (n_sidecars > 0) return bad_arg("asac_mlp_forward_multi: sidecars beside a wide job");
        for (int k = 0; k < n_jobs; ++k) {
            const asac_mlp_job_t& j = jobs[k];
            if (j.desc && desc_ok(*j.desc) && j.desc->in0 + j.desc->in1 <= kMaxW) {       // a narrow job beside a wide one
                if (int rc = asac_mlp_forward_multi(&j, 1, nullptr, 0, stream)) return rc;
                continue;
            }
            if (!j.desc || !desc_ok(*j.desc) || j.E <= 0 || j.N <= 0 || !j.x0 || (j.desc->in1 > 0 && !j.x1) || !j.out ||
                j.x0_window_T != 0)
                return bad_arg("asac_mlp_forward_multi: wide job");
            if (int rc = asac_mlp_forward(j.desc, static_cast<const float*>(j.params), j.member_stride, j.E,
                                          static_cast<const float*>(j.x0), j.x0_row_stride, j.x0_member_stride,
                                          static_cast<const float*>(j.x1), j.x1_row_stride, j.x1_member_stride, j.N,
                                          static_cast<float*>(j.out), stream))
                return rc;
        }
        return 0;
    }
    for (int k = 0; k < n_jobs; ++k) {
        const asac_mlp_job_t& j = jobs[k];
        if (!j.desc || !desc_ok(*j.desc) || j.desc->in0 + j.desc->in1 > kMaxW)
            return bad_arg("asac_mlp_forward_multi: job");
        if (j.E <= 0 || j.N <= 0 || !j.x0 || (j.desc->in1 > 0 && !j.x1) || !j.out || j.x0_window_T < 0)
            return bad_arg("asac_mlp_forward_multi: job");
        groups16 += ((j.N + 15) / 16) * j.E;
        all_stock = all_stock && stock3(*j.desc, j.params, j.member_stride) && j.N * (j.x0_row_stride + j.x1_row_stride + 1) < 0x1fffffffLL &&
                    (j.x0_window_T == 0 || (j.N / j.x0_window_T + 1) * j.x0_sample_stride < 0x1fffffffLL);
    }
    hipStream_t s = as_stream(stream);
    if (groups16 <= 256)
        return all_stock ? launch_forward_multi<16, 3>(jobs, n_jobs, sc, s) : launch_forward_multi<16, 0>(jobs, n_jobs, sc, s);
    return all_stock ? launch_forward_multi<32, 3>(jobs, n_jobs, sc, s) : launch_forward_multi<32, 0>(jobs, n_jobs, sc, s);
}

static bool sample_epilogue_ok(const asac_mlp_job_t& j, const asac_mlp_sample_epilogue_t& h) {
    const asac_squash_job_t& q = h.sample;
    if (!j.desc || j.desc->head_transform != 1 || j.desc->head_cols[0] != j.desc->head_cols[1] || j.E != 1) return false;
    const int A = j.desc->head_cols[0];
    if (A < 1 || 2 * A > 16 || q.A != A || q.rows != j.N || j.N * (int64_t)A >= 0x7fffffffLL) return false;
    if (!q.eps && !q.action && !h.eps2) return false;
    if (q.eps && (!q.a_tanh_out || !q.logp_out)) return false;
    if (q.x_out) return false;
    if ((q.action || h.eps2) && (q.T <= 0 || j.N % q.T != 0)) return false;
    if (q.action) {
        const int64_t sb = j.N / q.T;
        if (!q.prob_out || sb * q.action_stride_b + (int64_t)q.T * q.action_stride_t + q.action_offset + A >= 0x7fffffffLL ||
            sb * q.prob_stride_b + (int64_t)q.T * q.prob_stride_t + q.prob_offset + A >= 0x7fffffffLL ||
            q.action_stride_b < 0 || q.action_stride_t < 0 || q.prob_stride_b < 0 || q.prob_stride_t < 0)
            return false;
    }
    if (h.eps2 && (!h.a2_out || !h.logp2_out || h.t2 < 0 || h.t2 >= q.T)) return false;
    return true;
}

int asac_mlp_forward_multi_sampled_ok(const asac_mlp_job_t* jobs, int n_jobs, const asac_mlp_sample_epilogue_t* epilogues) {
    if (!jobs || !epilogues || n_jobs < 1 || n_jobs > ASAC_MLP_MAX_JOBS) return 0;
    bool any = false;
    for (int k = 0; k < n_jobs; ++k) {
        const asac_mlp_job_t& j = jobs[k];
        if (!j.desc || !desc_ok(*j.desc) || j.desc->in0 + j.desc->in1 > kMaxW || j.E <= 0 || j.N <= 0) return 0;
        const asac_mlp_sample_epilogue_t& h = epilogues[k];
        const bool on = h.sample.eps || h.sample.action || h.eps2;
        if (on && !sample_epilogue_ok(j, h)) return 0;
        // a workgroup pays 2-3 us of transcendental chains per tile it finishes (2 A of a wave's 64 lanes hold a row's
        // values): worth it while every tile has a workgroup of its own — the elementwise launch it replaces costs ~5 us —,
        // not where workgroups loop over tiles (cfg3's 20 736-row window: 25.8 us against 19.8 + 5.0 for the two launches)
        if (on && (j.N + 31) / 32 > 512) return 0;
        any = any || on;
    }
    return any ? 1 : 0;
}

int asac_mlp_forward_multi_sampled(const asac_mlp_job_t* jobs, int n_jobs, const asac_mlp_sample_epilogue_t* epilogues,
                                   const asac_sidecar_t* sidecars_host, int n_sidecars, void* stream) {
    if (!asac_mlp_forward_multi_sampled_ok(jobs, n_jobs, epilogues)) return bad_arg("asac_mlp_forward_multi_sampled");
    SidecarsDev sc{};
    if (sidecars_prepare(sidecars_host, n_sidecars, sc)) return bad_arg("asac_mlp_forward_multi_sampled: sidecar");
    int64_t groups16 = 0;
    bool all_stock = true;
    SampleEpis epis{};
    for (int k = 0; k < n_jobs; ++k) {
        const asac_mlp_job_t& j = jobs[k];
        if (!j.x0 || (j.desc->in1 > 0 && !j.x1) || !j.out || j.x0_window_T < 0) return bad_arg("asac_mlp_forward_multi_sampled: job");
        groups16 += ((j.N + 15) / 16) * j.E;
        all_stock = all_stock && stock3(*j.desc, j.params, j.member_stride) && j.N * (j.x0_row_stride + j.x1_row_stride + 1) < 0x1fffffffLL &&
                    (j.x0_window_T == 0 || (j.N / j.x0_window_T + 1) * j.x0_sample_stride < 0x1fffffffLL);
        const asac_mlp_sample_epilogue_t& h = epilogues[k];
        const asac_squash_job_t& q = h.sample;
        SampleEpi& d = epis.e[k];
        d.on = (q.eps || q.action || h.eps2) ? 1 : 0;
        if (!d.on) continue;
        d.eps = q.eps, d.eps2 = h.eps2, d.a_out = q.a_tanh_out, d.logp_out = q.logp_out, d.a2_out = h.a2_out, d.logp2_out = h.logp2_out;
        d.action = q.action, d.prob_out = q.prob_out;
        d.a_sb = (int32_t)q.action_stride_b, d.a_st = (int32_t)q.action_stride_t, d.a_off = q.action_offset;
        d.p_sb = (int32_t)q.prob_stride_b, d.p_st = (int32_t)q.prob_stride_t, d.p_off = q.prob_offset;
        d.A = q.A, d.T = q.T > 0 ? q.T : 1, d.t2 = h.t2;
    }
    hipStream_t s = as_stream(stream);
    if (groups16 <= 256)
        return all_stock ? launch_forward_multi<16, 3>(jobs, n_jobs, sc, s, &epis) : launch_forward_multi<16, 0>(jobs, n_jobs, sc, s, &epis);
    return all_stock ? launch_forward_multi<32, 3>(jobs, n_jobs, sc, s, &epis) : launch_forward_multi<32, 0>(jobs, n_jobs, sc, s, &epis);
}

/* row tiles (= workgroups along the row axis, = per-tile partial slabs) the backward of this shape uses */
int64_t asac_mlp_backward_tiles(int64_t N, int E) {
    const int TM = mlp_tile_rows(N, E);
    return (N + TM - 1) / TM;
}

int64_t asac_mlp_backward_workspace(int64_t member_stride, int E, int64_t N) {
    const int64_t tiles = asac_mlp_backward_tiles(N, E);
    return tiles * (int64_t)E * member_stride + tiles * E;   // floats: parameter partials | loss partials
}

// extent of this network's parameters inside a member segment
int64_t asac_mlp_param_extent(const asac_mlp_desc_t* desc) {
    if (!desc || !desc_ok(*desc)) return -1;
    int64_t used = 0;
    const int H = desc->width[desc->n_blocks - 1];
    for (int l = 0; l < desc->n_blocks; ++l) {
        const int Kin = l == 0 ? desc->in0 + desc->in1 : desc->width[l - 1];
        const int64_t we = desc->w_off[l] + (int64_t)desc->width[l] * Kin;
        const int64_t be = desc->b_off[l] + desc->width[l];
        used = we > used ? we : used;
        used = be > used ? be : used;
    }
    for (int h = 0; h < 2; ++h) {
        if (desc->head_cols[h] <= 0) continue;
        const int64_t we = desc->head_w_off[h] + (int64_t)desc->head_cols[h] * H;
        const int64_t be = desc->head_b_off[h] + desc->head_cols[h];
        used = we > used ? we : used;
        used = be > used ? be : used;
    }
    return used;
}

int asac_mlp_backward(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride, int E,
                      const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                      const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                      const float* grad_out, float* grad_x0, float* grad_x1, float* grad_params,
                      float* workspace, int reduce_mode, void* stream) {
    if (!desc || !desc_ok(*desc) || E <= 0 || N <= 0 || !x0 || (desc->in1 > 0 && !x1) || !grad_out)
        return bad_arg("asac_mlp_backward");
    if (grad_params && !workspace) return bad_arg("asac_mlp_backward: workspace");
    MlpArgs a = make_args(desc, params, member_stride, x0, x0_row_stride, x0_member_stride, x1, x1_row_stride,
                          x1_member_stride, N);
    a.gout = grad_out;
    a.gx0 = grad_x0;
    a.gx1 = grad_x1;
    return mlp_backward_common("asac_mlp_backward", desc, a, E, N, member_stride, grad_params, workspace,
                               reduce_mode, nullptr, as_stream(stream));
}

int asac_mlp_backward_policy_q(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride, int E,
                               const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                               const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                               const float* q_table, const int32_t* subset, int E_sample, float* grad_x1,
                               void* stream) {
    if (!desc || !desc_ok(*desc) || E <= 0 || N <= 0 || !x0 || desc->in1 <= 0 || !x1 || !q_table || !grad_x1 ||
        E_sample < 1 || E_sample > E)
        return bad_arg("asac_mlp_backward_policy_q");
    if (desc->head_cols[0] != 1 || desc->head_cols[1] != 0 || desc->head_transform != 0)
        return bad_arg("asac_mlp_backward_policy_q: not a scalar-head network");
    MlpArgs a = make_args(desc, params, member_stride, x0, x0_row_stride, x0_member_stride, x1, x1_row_stride,
                          x1_member_stride, N);
    a.q_table = q_table;
    a.subset = subset;
    a.E_sample = E_sample;
    a.gx1 = grad_x1;
    return mlp_backward_common("asac_mlp_backward_policy_q", desc, a, E, N, member_stride, nullptr, nullptr,
                               ASAC_MLP_REDUCE_OVERWRITE, nullptr, as_stream(stream));
}

int asac_mlp_backward_policy_sample(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride,
                                    const float* x0, int64_t x0_row_stride, int64_t N, const float* eps,
                                    const float* grad_a, int grad_a_members, const float* log_alpha,
                                    float* grad_params, float* workspace, int reduce_mode, void* stream) {
    if (!desc || !desc_ok(*desc) || N <= 0 || !x0 || desc->in1 != 0 || !eps || !grad_a || grad_a_members < 1 ||
        !log_alpha || !grad_params || !workspace)
        return bad_arg("asac_mlp_backward_policy_sample");
    if (desc->head_transform != 1 || desc->head_cols[0] != desc->head_cols[1] || desc->head_cols[0] > 16)
        return bad_arg("asac_mlp_backward_policy_sample: not a Gaussian-head policy");
    MlpArgs a = make_args(desc, params, member_stride, x0, x0_row_stride, 0, nullptr, 0, 0, N);
    a.eps = eps;
    a.grad_a = grad_a;
    a.grad_a_members = grad_a_members;
    a.log_alpha = log_alpha;
    return mlp_backward_common("asac_mlp_backward_policy_sample", desc, a, 1, N, member_stride, grad_params,
                               workspace, reduce_mode, nullptr, as_stream(stream));
}

int asac_policy_step_fused_ok(const asac_mlp_desc_t* q_desc, const float* q_params, int64_t q_member_stride,
                              const asac_mlp_desc_t* pi_desc, const float* pi_params, int64_t pi_member_stride, int64_t N) {
    if (!q_desc || !pi_desc || !desc_ok(*q_desc) || !desc_ok(*pi_desc) || N <= 0) return 0;
    if (!stock3(*q_desc, q_params, q_member_stride) || !stock3(*pi_desc, pi_params, pi_member_stride)) return 0;
    if (q_desc->head_cols[0] != 1 || q_desc->head_cols[1] != 0 || q_desc->head_transform != 0) return 0;
    if (pi_desc->head_transform != 1 || pi_desc->head_cols[0] != pi_desc->head_cols[1] || pi_desc->in1 != 0) return 0;
    if (q_desc->in0 != pi_desc->in0 || q_desc->in1 != pi_desc->head_cols[0] || 2 * pi_desc->head_cols[0] > kHeadPad) return 0;
    return mlp_tile_rows(N, 1) == kPsRows ? 1 : 0;        // the partials' tile count is asac_mlp_backward_tiles(N, 1)
}

int asac_policy_step_fused(const asac_mlp_desc_t* q_desc, const float* q_params, int64_t q_member_stride,
                           const asac_mlp_desc_t* pi_desc, const float* pi_params, int64_t pi_member_stride,
                           const float* x, int64_t x_row_stride, int64_t N, const float* action, const float* eps,
                           const float* log_alpha, const int32_t* subset, float* q_out, float* a_tanh_out,
                           float* logp_out, float* ls_out, float* pi_grad_params, float* workspace, int reduce_mode,
                           void* stream) {
    if (!asac_policy_step_fused_ok(q_desc, q_params, q_member_stride, pi_desc, pi_params, pi_member_stride, N) ||
        !x || !eps || !log_alpha || !pi_grad_params || !workspace || (!action && (!a_tanh_out || !logp_out)))
        return bad_arg("asac_policy_step_fused");
    PolicyStepArgs a{};
    const int A = pi_desc->head_cols[0];
    a.q = make_args(q_desc, q_params, q_member_stride, x, x_row_stride, 0, action, A, 0, N);
    a.pi = make_args(pi_desc, pi_params, pi_member_stride, x, x_row_stride, 0, nullptr, 0, 0, N);
    if (!offsets32(a.q) || !offsets32(a.pi)) return bad_arg("asac_policy_step_fused: offsets");
    a.q.subset = subset;
    a.a_out = a_tanh_out;
    a.logp_out = logp_out;
    a.ls_out = ls_out;
    a.pi.eps = eps;
    a.pi.log_alpha = log_alpha;
    a.pi.partial = workspace;
    a.q_out = q_out;
    static bool attr_done = false;
    if (int rc = set_lds_limit(reinterpret_cast<const void*>(k_policy_step), sizeof(PsLds), attr_done,
                               "asac_policy_step_fused: hipFuncSetAttribute"))
        return rc;
    const int tiles = (int)((N + kPsRows - 1) / kPsRows);
    hipStream_t s = as_stream(stream);
    ASAC_LAUNCH(k_policy_step, dim3((unsigned)tiles), dim3(kPsThreads), sizeof(PsLds), s, a);
    if (reduce_mode != ASAC_MLP_REDUCE_DEFER) {
        const int64_t used = asac_mlp_param_extent(pi_desc);
        if (tiles >= kSlicedFrom)
            hipLaunchKernelGGL(k_mlp_reduce_partials_sliced, dim3((unsigned)((used + 63) / 64), 1u), dim3(64 * kReduceSlices),
                               0, s, workspace, tiles, 1, pi_member_stride, used, pi_grad_params,
                               reduce_mode == ASAC_MLP_REDUCE_ACCUMULATE ? 1 : 0, nullptr, nullptr, 0.f);
        else
            hipLaunchKernelGGL(k_mlp_reduce_partials, dim3((unsigned)((used + 255) / 256), 1u), dim3(256), 0, s, workspace,
                               tiles, 1, pi_member_stride, used, pi_grad_params,
                               reduce_mode == ASAC_MLP_REDUCE_ACCUMULATE ? 1 : 0, nullptr, nullptr, 0.f);
    }
    return finish_launch("asac_policy_step_fused");
}

static bool fits32(int64_t v) { return v >= 0 && v < 0x1fffffffLL; }

// asac_mlp_desc_t + addressing -> the compact kernel-argument form (callers have checked stock3 and the 32-bit ranges)
static StockJobArg stock_job_arg(const asac_mlp_desc_t* d, const float* params, int64_t member_stride, const float* x0,
                                 int64_t x0_rs, int64_t x0_ms, const float* x1, int64_t x1_rs, int64_t x1_ms, int64_t N,
                                 int64_t x0_T, int64_t x0_sb, float* out) {
    StockJobArg j{};
    j.P = params, j.x0 = x0, j.x1 = x1, j.out = out;
    j.member_stride = (int32_t)member_stride, j.N = (int32_t)N;
    j.x0_rs = (int32_t)x0_rs, j.x0_ms = (int32_t)x0_ms, j.x1_rs = (int32_t)x1_rs, j.x1_ms = (int32_t)x1_ms;
    j.x0_sb = (int32_t)x0_sb, j.x0_T = (int32_t)x0_T;
    j.in0 = d->in0, j.in1 = d->in1, j.h0 = d->head_cols[0], j.h1 = d->head_cols[1];
    j.hw0 = (int32_t)d->head_w_off[0], j.hw1 = (int32_t)d->head_w_off[1];
    j.hb0 = (int32_t)d->head_b_off[0], j.hb1 = (int32_t)d->head_b_off[1];
    for (int l = 0; l < 3; ++l) {
        j.w_off[l] = (int32_t)d->w_off[l], j.b_off[l] = (int32_t)d->b_off[l];
        j.residual_bits |= (d->residual[l] ? 1 : 0) << l;
    }
    j.head_transform = d->head_transform;
    return j;
}

static bool pi_q_job_ok(const asac_pi_q_job_t& j) {
    const asac_mlp_job_t &p = j.pi, &q = j.q;
    if (!p.desc || !q.desc || !desc_ok(*p.desc) || !desc_ok(*q.desc) || p.N <= 0 || q.N != p.N || p.E != 1 || q.E < 1) return false;
    if (!stock3(*p.desc, p.params, p.member_stride) || !stock3(*q.desc, q.params, q.member_stride)) return false;
    if (q.desc->head_cols[0] != 1 || q.desc->head_cols[1] != 0 || q.desc->head_transform != 0) return false;
    if (p.desc->head_transform != 1 || p.desc->head_cols[0] != p.desc->head_cols[1] || p.desc->in1 != 0) return false;
    const int A = p.desc->head_cols[0];
    if (q.desc->in0 != p.desc->in0 || q.desc->in1 != A || 2 * A > kHeadPad || A > 8) return false;
    if (!p.x0 || q.x0 != p.x0 || q.x0_row_stride != p.x0_row_stride || q.x0_window_T != p.x0_window_T ||
        q.x0_sample_stride != p.x0_sample_stride || p.x0_member_stride != 0 || q.x0_member_stride != 0 || !q.out)
        return false;        // the critics read the policy's rows
    const asac_squash_job_t& s = j.sample;
    if (!s.eps || !s.a_tanh_out || !s.logp_out || s.A != A || s.rows != p.N) return false;
    if (s.action && (!s.prob_out || s.T <= 0)) return false;
    if (j.eps2 && (!j.a2_out || !j.logp2_out || s.T <= 0 || j.t2 < 0 || j.t2 >= s.T)) return false;
    if (!fits32(p.member_stride) || !fits32(q.member_stride * q.E)) return false;
    if (s.action && (!fits32((p.N / s.T + 1) * s.action_stride_b) || !fits32((p.N / s.T + 1) * s.prob_stride_b) ||
                     !fits32(s.T * s.action_stride_t) || !fits32(s.T * s.prob_stride_t)))
        return false;
    return p.N * (p.x0_row_stride + 1) < 0x1fffffffLL &&
           (p.x0_window_T == 0 || (p.N / p.x0_window_T + 1) * p.x0_sample_stride < 0x1fffffffLL);
}

int asac_policy_sample_q_forward_ok(const asac_pi_q_job_t* job) { return job && pi_q_job_ok(*job) ? 1 : 0; }

int asac_policy_sample_q_forward(const asac_pi_q_job_t* job, const asac_mlp_job_t* extra_jobs, int n_extra,
                                 const asac_sidecar_t* sidecars_host, int n_sidecars, void* stream) {
    if (!job || !pi_q_job_ok(*job) || n_extra < 0 || n_extra > ASAC_MLP_MAX_JOBS || (n_extra > 0 && !extra_jobs))
        return bad_arg("asac_policy_sample_q_forward");
    SidecarsDev sc{};
    if (sidecars_prepare(sidecars_host, n_sidecars, sc)) return bad_arg("asac_policy_sample_q_forward: sidecar");
    PiQLaunch m{};
    const asac_mlp_job_t &p = job->pi, &q = job->q;
    m.pi = stock_job_arg(p.desc, p.params, p.member_stride, p.x0, p.x0_row_stride, 0, nullptr, 0, 0, p.N, p.x0_window_T,
                         p.x0_sample_stride, p.out);
    m.q = stock_job_arg(q.desc, q.params, q.member_stride, p.x0, p.x0_row_stride, 0, p.x0, 0, 0, p.N, p.x0_window_T,
                        p.x0_sample_stride, q.out);            // (x1: the sampled actions, on chip)
    m.E = q.E;
    const int tiles = (int)((p.N + 15) / 16);
    const int cap = 256 / q.E > 0 ? 256 / q.E : 1;
    m.tile_groups = tiles <= cap ? tiles : cap;
    m.blocks = m.tile_groups * q.E;
    const asac_squash_job_t& sj = job->sample;
    m.eps = sj.eps;
    m.a_out = sj.a_tanh_out;
    m.logp_out = sj.logp_out;
    m.sp = StoredProbArg{sj.action, sj.prob_out, sj.T, (int32_t)sj.action_stride_b, (int32_t)sj.action_stride_t, sj.action_offset,
                         (int32_t)sj.prob_stride_b, (int32_t)sj.prob_stride_t, sj.prob_offset, 0};
    m.eps2 = job->eps2;
    m.T = sj.T > 0 ? sj.T : 1;
    m.t2 = job->t2;
    m.a2_out = job->a2_out;
    m.logp2_out = job->logp2_out;
    m.x_n = n_extra;
    int blocks = 0;
    for (int k = 0; k < n_extra; ++k) {
        const asac_mlp_job_t& j = extra_jobs[k];
        if (!j.desc || !desc_ok(*j.desc) || !stock3(*j.desc, j.params, j.member_stride) || j.E <= 0 || j.N <= 0 || !j.x0 ||
            (j.desc->in1 > 0 && !j.x1) || !j.out || j.x0_window_T < 0 ||
            j.N * (j.x0_row_stride + j.x1_row_stride + 1) >= 0x1fffffffLL ||
            (j.x0_window_T > 0 && (j.N / j.x0_window_T + 1) * j.x0_sample_stride >= 0x1fffffffLL) ||
            !fits32(j.member_stride) || !fits32(j.x0_member_stride) || !fits32(j.x1_member_stride))
            return bad_arg("asac_policy_sample_q_forward: extra job");
        m.x_job[k] = stock_job_arg(j.desc, j.params, j.member_stride, j.x0, j.x0_row_stride, j.x0_member_stride, j.x1,
                                   j.x1_row_stride, j.x1_member_stride, j.N, j.x0_window_T, j.x0_sample_stride, j.out);
        m.x_E[k] = j.E;
        m.x_first_block[k] = blocks;
        m.x_tile_stride[k] = mlp_tile_groups(j.N, j.E, 1, 16);
        blocks += m.x_tile_stride[k] * j.E;
    }
    m.x_blocks = blocks;
    static bool attr_done = false;
    static bool attr_done1 = false;
    if (int rc = set_lds_limit(reinterpret_cast<const void*>(k_pi_sample_q<ASAC_MAX_SIDECARS>), sizeof(PiQLds), attr_done,
                               "asac_policy_sample_q_forward: hipFuncSetAttribute"))
        return rc;
    if (int rc = set_lds_limit(reinterpret_cast<const void*>(k_pi_sample_q<1>), sizeof(PiQLds), attr_done1,
                               "asac_policy_sample_q_forward: hipFuncSetAttribute"))
        return rc;
    const SidecarsDev none{};
    for (int rep = 0; rep < g_launch_repeat; ++rep) {      // (repeat knob: only the last repetition carries the sidecars)
        const bool last = rep == g_launch_repeat - 1;
        const dim3 grid((unsigned)(m.blocks + blocks + (last ? sc.blocks : 0)));
        if (sc.n <= 1)
            hipLaunchKernelGGL(k_pi_sample_q<1>, grid, dim3(kPiQThreads), sizeof(PiQLds), as_stream(stream), m,
                               sidecars_first<1>(last ? sc : none));
        else
            hipLaunchKernelGGL(k_pi_sample_q<ASAC_MAX_SIDECARS>, grid, dim3(kPiQThreads), sizeof(PiQLds), as_stream(stream), m,
                               last ? sc : none);
    }
    return finish_launch("asac_policy_sample_q_forward");
}

int asac_mlp_backward_qloss(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride, int E,
                            const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                            const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                            const float* target_q, const float* y, const float* weights, float clip_eps,
                            float* loss_out, float* grad_params, float* workspace, int reduce_mode,
                            void* stream) {
    return asac_mlp_backward_qloss_gx(desc, params, member_stride, E, x0, x0_row_stride, x0_member_stride, x1, x1_row_stride,
                                      x1_member_stride, N, target_q, y, weights, clip_eps, loss_out, nullptr, grad_params,
                                      workspace, reduce_mode, stream);
}

int asac_mlp_backward_qloss_return_ok(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride, int E,
                                      int64_t N, const asac_vtrace_args_t* ret) {
    if (!desc || !desc_ok(*desc) || !ret || E <= 0 || N <= 0 || ret->B != N || ret->n <= 0 || !ret->q || !ret->y_out ||
        ret->E_sample <= 0 || ret->E_sample > ASAC_MAX_ENSEMBLE || ret->td_error_out)
        return 0;
    if (desc->in0 + desc->in1 > kMaxW || !stock3(*desc, params, member_stride)) return 0;
    const int tm = mlp_tile_rows(N, E);
    const size_t lds = (tm == 16 ? sizeof(MlpBwdLds<16>) : sizeof(MlpBwdLds<32>)) +
                       (size_t)(2 * ((ret->n + 1) | 1) + 2) * tm * sizeof(float);
    // one thread per (row, step) of a tile: 16-row tiles run ASAC_BWD_WAVES waves (n <= 64 with 16), 32-row tiles 512 threads
    const int threads = tm == 16 ? 64 * ASAC_BWD_WAVES : 32 * 16;
    return tm * ret->n <= threads && lds <= 128 * 1024;
}

int asac_mlp_backward_qloss_return(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride, int E,
                                   const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                                   const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                                   const float* target_q, const asac_vtrace_args_t* ret, const float* weights,
                                   float clip_eps, float* loss_out, float* grad_x0, float* grad_params, float* workspace,
                                   int reduce_mode, void* stream) {
    if (!desc || !desc_ok(*desc) || E <= 0 || N <= 0 || !x0 || (desc->in1 > 0 && !x1) || !target_q || !ret ||
        !grad_params || !workspace || clip_eps <= 0.f)
        return bad_arg("asac_mlp_backward_qloss_return");
    if (desc->head_cols[0] != 1 || desc->head_cols[1] != 0 || desc->head_transform != 0)
        return bad_arg("asac_mlp_backward_qloss_return: not a scalar-head network");
    if (reduce_mode != ASAC_MLP_REDUCE_DEFER && !loss_out) return bad_arg("asac_mlp_backward_qloss_return: loss_out");
    if (!asac_mlp_backward_qloss_return_ok(desc, params, member_stride, E, N, ret) ||
        (ret->use_n_step_is && (!ret->mu_prob || !ret->pi_prob || ret->A <= 0)))
        return bad_arg("asac_mlp_backward_qloss_return: return arguments");
    MlpArgs a = make_args(desc, params, member_stride, x0, x0_row_stride, x0_member_stride, x1, x1_row_stride,
                          x1_member_stride, N);
    if (!offsets32(a)) return bad_arg("asac_mlp_backward_qloss_return: offsets");
    a.tq = target_q;
    a.y = ret->y_out;
    a.w = weights;
    a.clip_eps = clip_eps;
    a.gx0 = grad_x0;
    RetIn<true> rv{*ret, vtrace_scan_lanes(ret->B, ret->n)};
    return mlp_backward_common("asac_mlp_backward_qloss_return", desc, a, E, N, member_stride, grad_params, workspace,
                               reduce_mode, loss_out, as_stream(stream), &rv);
}

int asac_mlp_backward_qloss_gx(const asac_mlp_desc_t* desc, const float* params, int64_t member_stride, int E,
                               const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                               const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                               const float* target_q, const float* y, const float* weights, float clip_eps,
                               float* loss_out, float* grad_x0, float* grad_params, float* workspace, int reduce_mode,
                               void* stream) {
    if (!desc || !desc_ok(*desc) || E <= 0 || N <= 0 || !x0 || (desc->in1 > 0 && !x1) || !target_q || !y ||
        !grad_params || !workspace || clip_eps <= 0.f)
        return bad_arg("asac_mlp_backward_qloss");
    if (desc->head_cols[0] != 1 || desc->head_cols[1] != 0 || desc->head_transform != 0)
        return bad_arg("asac_mlp_backward_qloss: not a scalar-head network");
    if (reduce_mode != ASAC_MLP_REDUCE_DEFER && !loss_out) return bad_arg("asac_mlp_backward_qloss: loss_out");
    MlpArgs a = make_args(desc, params, member_stride, x0, x0_row_stride, x0_member_stride, x1, x1_row_stride,
                          x1_member_stride, N);
    a.tq = target_q;
    a.y = y;
    a.w = weights;
    a.clip_eps = clip_eps;
    a.gx0 = grad_x0;
    return mlp_backward_common("asac_mlp_backward_qloss", desc, a, E, N, member_stride, grad_params, workspace,
                               reduce_mode, loss_out, as_stream(stream));
}

}  // extern "C"
