// y = tanh(x W^T + b) over the rows of a sampled window, forward and backward: the state head the reference's
// representation plugins put after their encoders (`nn.Sequential(nn.Linear(n, 8), nn.Tanh())`,
// tests/nn_conv_vanilla.py:11-14, tests/nn_conv_attn.py, envs/test/nn_rnn.py).  At [B*L, <= 64] -> <= 16 the
// library GEMMs are pure latency (the weight-gradient GEMM reduces 4 608 rows into 8 x 18 numbers: 18 us, plus a
// bias reduction, the tanh launches and two gradient accumulations); here a pass is one launch.
//
// A workgroup (4 waves) owns 64 rows: staged in LDS as they lie in memory (coalesced, whatever the row strides and the
// two-part input), then every product on 16x16x4 f32 MFMA with the rows as the N dimension — forward x W^T, backward g W
// (input gradient) and G^T (X | 1) (per-workgroup partial parameter gradients).  The partials are summed in workgroup order
// (deterministic; no float atomics): by the last workgroup to finish when they are few, else by csrc/xty.hip's slice reduction
// as a second launch.  (Rounds 1-4 ran one lane per row with the weights broadcast from LDS: at K = 64 the 64-step scalar
// loops of a lane made the backward 35 us for 9 216 rows; the MFMA form is 9.)
#include "asac_common.h"

namespace asac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLinMaxK = ASAC_LINEAR_TANH_MAX_IN;      // 64
constexpr int kLinMaxO = ASAC_LINEAR_TANH_MAX_OUT;     // 16
constexpr int kLinRows = 64, kLinThreads = 256;        // rows per workgroup (4 waves: one 16-row tile each)
#ifndef ASAC_LIN_TAIL_MAX
#define ASAC_LIN_TAIL_MAX 12288
#endif
constexpr int64_t kLinTailMax = ASAC_LIN_TAIL_MAX;
constexpr int64_t kLinTailBlocks = 48;                 // workgroups the last one still sums faster than a second launch would                 // partial floats (workgroups x O (K + 1)) the last workgroup sums itself

struct LinArgs {
    const float* x;         // the input rows: x [N][K0] | x1 [N][K - K0] side by side (x1 NULL: K0 == K) — the
    int64_t x_stride;       // concatenation `torch.cat([a, b], -1)` the reference's plugins feed their heads, read in place
    const float* x1;
    int64_t x1_stride;
    int K0;
    int x_T;                // > 0: x row r lies at (r / x_T) * x_sb + (r % x_T) * x_stride — a
    int64_t x_sb;           // [samples][T][K0] slice of the sampled windows (obs[:, b:]) read in place
    const float* w;         // [O][K]
    const float* b;         // [O]
    int64_t N;
    int K, O;
    float* y;               // [N][O]
    const float* gy;        // [members][N / gy_window][O]: the output gradient of row r is sum_e gy[e][r / gy_window] when
    int gy_members;         // r % gy_window == gy_position and zero otherwise (members 1, window 1: a dense [N][O])
    int gy_window, gy_position;
    float* gx;              // [N][K0] or null
    float* gx1;             // [N][K - K0] or null
    float* gp;              // W | b gradients, O*K + O
    int accumulate;
    float* partial;         // [blocks][O*(K+1)]
    unsigned int* counter;
};

// ---- shared pieces: 64 rows a workgroup of 4 waves; the rows staged in LDS as they lie in memory (coalesced), the products on
// 16x16x4 f32 MFMA with the ROWS as the N dimension ---------------------------------------------------------------------------
constexpr int kWtPitch = 20;       // W^T rows at a pitch of 20 floats: the 16-byte reads of 16 lanes (k = lane) hit distinct banks
constexpr int kXsPitch = kLinMaxK + 1;

__device__ __forceinline__ void stage_wt(const LinArgs& a, float* wt) {
    for (int i = threadIdx.x; i < a.K * kLinMaxO; i += kLinThreads) {
        const int k = i / kLinMaxO, o = i - k * kLinMaxO;
        wt[k * kWtPitch + o] = o < a.O ? a.w[o * a.K + k] : 0.f;
    }
}

// where the two parts of a row start (the second indexed by k like the first)
__device__ __forceinline__ void stage_offsets(const LinArgs& a, int64_t r0, int rows, int64_t* roff, int64_t* roff1) {
    if (threadIdx.x < kLinRows) {
        const int64_t rs = (int)threadIdx.x < rows ? r0 + threadIdx.x : r0;
        roff[threadIdx.x] = a.x_T ? (rs / a.x_T) * a.x_sb + (rs % a.x_T) * a.x_stride : rs * a.x_stride;
        roff1[threadIdx.x] = a.x1 ? rs * a.x1_stride - a.K0 : 0;
    }
}

// xs[row][k] <- the workgroup's rows (zero beyond `rows`): thread t starts at element t of the [64][K] block and steps by 256
// elements; eight loads in flight, then their eight LDS stores
__device__ __forceinline__ void stage_rows(const LinArgs& a, int rows, const int64_t* roff, const int64_t* roff1, float* xs) {
    const int K = a.K, dr = kLinThreads / K, dk = kLinThreads - dr * K;
    int row = threadIdx.x / K, k = threadIdx.x - row * K;
    while (row < kLinRows) {
        float v[8];
        int at[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = 0.f;
            at[u] = row < kLinRows ? row * kXsPitch + k : -1;
            if (row < rows) v[u] = k < a.K0 ? a.x[roff[row] + k] : a.x1[roff1[row] + k];
            row += dr, k += dk;
            if (k >= K) k -= K, ++row;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (at[u] >= 0) xs[at[u]] = v[u];
    }
}

// y^T[o][row] = sum_k W[o][k] x[row][k]: A lane (i = o, slot q) = W^T[k = 4 s + q][o], B lane (j = row, slot q) = x[row][4 s + q]
__global__ __launch_bounds__(kLinThreads) void k_linear_tanh_fwd(const LinArgs a) {
    __shared__ __attribute__((aligned(16))) float wt[kLinMaxK * kWtPitch];
    __shared__ float bias[kLinMaxO];
    __shared__ float xs[kLinRows * kXsPitch];
    __shared__ int64_t roff[kLinRows], roff1[kLinRows];
    const int64_t r0 = (int64_t)blockIdx.x * kLinRows;
    const int rows = (int)min((int64_t)kLinRows, a.N - r0);
    stage_wt(a, wt);
    if (threadIdx.x < kLinMaxO) bias[threadIdx.x] = (int)threadIdx.x < a.O ? a.b[threadIdx.x] : 0.f;
    stage_offsets(a, r0, rows, roff, roff1);
    __syncthreads();
    stage_rows(a, rows, roff, roff1, xs);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const float* xrow = xs + (16 * wave + i) * kXsPitch;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int steps = (a.K + 3) >> 2;
    for (int s = 0; s < steps; s += 2) {
        const int k0 = 4 * s + q, k1 = k0 + 4;
        const float a0 = k0 < a.K ? wt[k0 * kWtPitch + i] : 0.f, b0 = k0 < a.K ? xrow[k0] : 0.f;
        const float a1 = k1 < a.K ? wt[k1 * kWtPitch + i] : 0.f, b1 = k1 < a.K ? xrow[k1] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    const f32x4 acc = acc0 + acc1;                       // acc[r] = pre-activation of output 4 q + r, row 16 wave + i
    const int64_t r = r0 + 16 * wave + i;
    if (r >= a.N) return;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int o = 4 * q + rr;
        if (o < a.O) a.y[r * a.O + o] = tanhf(acc[rr] + bias[o]);
    }
}

// TAIL: the last workgroup to arrive sums the partials (few, small partials: one launch); else the caller reduces them with
// csrc/xty.hip's slice reduction
template <bool TAIL>
__global__ __launch_bounds__(kLinThreads) void k_linear_tanh_bwd(const LinArgs a) {
    __shared__ __attribute__((aligned(16))) float wt[kLinMaxK * kWtPitch];
    __shared__ __attribute__((aligned(16))) float gs[kLinRows * kLinMaxO];          // g = gy * (1 - y^2), [row][16]
    __shared__ float xs[kLinRows * kXsPitch];          // the workgroup's input rows and a column of ones, [row][K + 1]
    __shared__ int64_t roff[kLinRows], roff1[kLinRows];
    __shared__ bool last;
    const int64_t r0 = (int64_t)blockIdx.x * kLinRows;
    const int rows = (int)min((int64_t)kLinRows, a.N - r0);
    stage_wt(a, wt);
    stage_offsets(a, r0, rows, roff, roff1);
    {
        // g: thread t -> row t / 4, outputs 4 (t % 4) ... + 3
        const int row = threadIdx.x >> 2, o0 = 4 * (threadIdx.x & 3);
        const int64_t r = r0 + row;
        const bool live = row < rows;
        const int64_t gy_row = live ? r / a.gy_window : 0;
        const bool gy_on = live && (a.gy_window == 1 || (int)(r - gy_row * a.gy_window) == a.gy_position);
        const int64_t gy_rows = a.N / a.gy_window;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o = o0 + rr;
            float v = 0.f;
            if (gy_on && o < a.O) {
                const float yv = a.y[r * a.O + o];
                float gsum = a.gy[gy_row * a.O + o];
                for (int e = 1; e < a.gy_members; ++e) gsum += a.gy[(e * gy_rows + gy_row) * a.O + o];     // member order
                v = gsum * (1.f - yv * yv);
            }
            gs[row * kLinMaxO + o] = v;
        }
    }
    __syncthreads();
    stage_rows(a, rows, roff, roff1, xs);
    if (threadIdx.x < kLinRows) xs[threadIdx.x * kXsPitch + a.K] = 1.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    if (a.gx || a.gx1) {
        // gx^T[k][row] = sum_o W[o][k] g[row][o]: A lane (i = k, slot q) = W^T[k][4 q + s], B lane (j = row, slot q) = g[row][4 q + s]
        // for the four steps s; the result holds four consecutive k of a row
        const int K1 = a.K - a.K0;
        const int64_t r = r0 + 16 * wave + i;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gs + (16 * wave + i) * kLinMaxO + 4 * q);
        for (int kt = 0; 16 * kt < a.K; ++kt) {
            const int kw = 16 * kt + i;
            f32x4 wv = {0.f, 0.f, 0.f, 0.f};
            if (kw < a.K) wv = *reinterpret_cast<const f32x4*>(wt + kw * kWtPitch + 4 * q);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], gv[s], acc, 0, 0, 0);
            if (r >= a.N) continue;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int k = 16 * kt + 4 * q + rr;
                if (k < a.K0) {
                    if (a.gx) a.gx[r * a.K0 + k] = acc[rr];
                } else if (k < a.K && a.gx1) {
                    a.gx1[r * K1 + (k - a.K0)] = acc[rr];
                }
            }
        }
    }
    __syncthreads();
    // partial parameter gradients of this workgroup, G^T [O x rows] times (X | 1) [rows x (K + 1)] on the matrix
    // cores, column K (the ones) being the bias; stored as [O][K] | [O] (the layout of the gradient and of csrc/xty.hip's
    // partials).  A lane (i, kk) = g[row 4 step + kk][o = i], B lane (j, kk) = x[row 4 step + kk][column 16 tile + j]; a
    // wave takes every fourth column tile.  Dead rows carry g = 0.
    const int P = a.O * (a.K + 1);
    float* mine = a.partial + (int64_t)blockIdx.x * P;
    {
        const int kk = q;
        const int tiles = (a.K + 1 + 15) / 16;
        for (int tile = wave; tile < tiles; tile += kLinThreads / 64) {
            const int col = tile * 16 + i, colc = min(col, a.K);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const bool real = col <= a.K;
            for (int s0 = 0; s0 < kLinRows / 4; s0 += 8) {        // eight steps' operands in flight, two chains
                float av[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = 4 * (s0 + u) + kk;
                    av[u] = gs[row * kLinMaxO + i];
                    xv[u] = xs[row * kXsPitch + colc];
                }
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], real ? xv[u] : 0.f, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], real ? xv[u + 1] : 0.f, acc1, 0, 0, 0);
                }
            }
            const f32x4 acc = acc0 + acc1;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o = 4 * kk + rr;
                if (o < a.O && col <= a.K) {
                    float* dst = mine + (col < a.K ? o * a.K + col : a.O * a.K + o);
                    // (TAIL: written through to where every workgroup sees it — no release fence, which would write back
                    // this XCD's whole L2)
                    if (TAIL) __hip_atomic_store(dst, acc[rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *dst = acc[rr];
                }
            }
        }
    }
    if (!TAIL) return;
    // the last workgroup to arrive sums the partials in workgroup order (relaxed device-scope accesses ordered by the wait
    // for this wave's stores and the barrier)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    // sixteen partials requested at a time, then added in workgroup order
    const int nb = (int)gridDim.x;
    for (int t = threadIdx.x; t < P; t += kLinThreads) {
        float* col = a.partial + t;
        float s = 0.f;
        for (int b0 = 0; b0 < nb; b0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = __hip_atomic_load(col + (int64_t)min(b0 + u, nb - 1) * P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 16; ++u) s += b0 + u < nb ? v[u] : 0.f;
        }
        float* dst = a.gp + t;
        *dst = a.accumulate ? *dst + s : s;
    }
    if (threadIdx.x == 0) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
}

static bool lin_dims_ok(int64_t N, int K, int O) { return N > 0 && K > 0 && K <= kLinMaxK && O > 0 && O <= kLinMaxO; }

}  // namespace asac

using namespace asac;

extern "C" {

int64_t asac_linear_tanh_workspace(int64_t N, int K, int O) {
    if (!lin_dims_ok(N, K, O)) return -1;
    return ((N + kLinRows - 1) / kLinRows) * (int64_t)O * (K + 1) + 1;      // partials + the arrival counter
}

int asac_linear_tanh_forward2(const float* x0, int64_t x0_row_stride, int K0, const float* x1, int64_t x1_row_stride,
                              int K1, const float* weight, const float* bias, int64_t N, int O, float* y, void* stream) {
    return asac_linear_tanh_forward2w(x0, x0_row_stride, 0, 0, K0, x1, x1_row_stride, K1, weight, bias, N, O, y, stream);
}

int asac_linear_tanh_forward2w(const float* x0, int64_t x0_row_stride, int x0_window_T, int64_t x0_sample_stride, int K0,
                               const float* x1, int64_t x1_row_stride, int K1, const float* weight, const float* bias,
                               int64_t N, int O, float* y, void* stream) {
    const int K = K0 + (x1 ? K1 : 0);
    if (!lin_dims_ok(N, K, O) || !x0 || K0 <= 0 || !weight || !bias || !y || x0_row_stride < K0 ||
        (x1 && (K1 <= 0 || x1_row_stride < K1)) || x0_window_T < 0 || (x0_window_T && N % x0_window_T != 0))
        return bad_arg("asac_linear_tanh_forward");
    LinArgs a{};
    a.x_T = x0_window_T, a.x_sb = x0_sample_stride;
    a.x = x0, a.x_stride = x0_row_stride, a.K0 = K0, a.x1 = x1, a.x1_stride = x1_row_stride;
    a.w = weight, a.b = bias, a.N = N, a.K = K, a.O = O, a.y = y;
    ASAC_LAUNCH(k_linear_tanh_fwd, dim3((unsigned)((N + kLinRows - 1) / kLinRows)), dim3(kLinThreads), 0, as_stream(stream), a);
    return finish_launch("asac_linear_tanh_forward");
}

int asac_linear_tanh_forward(const float* x, int64_t x_row_stride, const float* weight, const float* bias, int64_t N,
                             int K, int O, float* y, void* stream) {
    return asac_linear_tanh_forward2(x, x_row_stride, K, nullptr, 0, 0, weight, bias, N, O, y, stream);
}

int asac_linear_tanh_backward2(const float* x0, int64_t x0_row_stride, int K0, const float* x1, int64_t x1_row_stride,
                               int K1, const float* weight, const float* y, const float* grad_y, int grad_members,
                               int grad_window, int grad_position, int64_t N, int O, float* grad_x0, float* grad_x1,
                               float* grad_params, int accumulate, float* workspace, void* stream) {
    return asac_linear_tanh_backward2w(x0, x0_row_stride, 0, 0, K0, x1, x1_row_stride, K1, weight, y, grad_y, grad_members,
                                       grad_window, grad_position, N, O, grad_x0, grad_x1, grad_params, accumulate,
                                       workspace, stream);
}

int asac_linear_tanh_backward2w(const float* x0, int64_t x0_row_stride, int x0_window_T, int64_t x0_sample_stride, int K0,
                                const float* x1, int64_t x1_row_stride, int K1, const float* weight, const float* y,
                                const float* grad_y, int grad_members, int grad_window, int grad_position, int64_t N, int O,
                                float* grad_x0, float* grad_x1, float* grad_params, int accumulate, float* workspace,
                                void* stream) {
    const int K = K0 + (x1 ? K1 : 0);
    if (x0_window_T < 0 || (x0_window_T && N % x0_window_T != 0)) return bad_arg("asac_linear_tanh_backward");
    if (!lin_dims_ok(N, K, O) || !x0 || K0 <= 0 || !weight || !y || !grad_y || !grad_params || !workspace ||
        x0_row_stride < K0 || (x1 && (K1 <= 0 || x1_row_stride < K1)) || (!x1 && grad_x1) || grad_members <= 0 ||
        grad_window <= 0 || N % grad_window != 0 || grad_position < 0 || grad_position >= grad_window)
        return bad_arg("asac_linear_tanh_backward");
    const int64_t blocks = (N + kLinRows - 1) / kLinRows;
    LinArgs a{};
    a.x_T = x0_window_T, a.x_sb = x0_sample_stride;
    a.x = x0, a.x_stride = x0_row_stride, a.K0 = K0, a.x1 = x1, a.x1_stride = x1_row_stride;
    a.w = weight, a.N = N, a.K = K, a.O = O;
    a.y = const_cast<float*>(y), a.gy = grad_y, a.gy_members = grad_members, a.gy_window = grad_window;
    a.gy_position = grad_position, a.gx = grad_x0, a.gx1 = grad_x1, a.gp = grad_params, a.accumulate = accumulate;
    a.partial = workspace;
    a.counter = reinterpret_cast<unsigned int*>(workspace + blocks * (int64_t)O * (K + 1));
    // the last workgroup sums the partials itself while there are few of them (sixteen agent-scope loads a round: 144
    // workgroups' tail took 6.4 us, more than the reduction launch it saves); ASAC_LINEAR_SUM_DEFER: the sums are left to
    // asac_sum_partials_multi
    if (accumulate != ASAC_LINEAR_SUM_DEFER && blocks <= kLinTailBlocks && blocks * O * (K + 1) <= kLinTailMax) {
        ASAC_LAUNCH(k_linear_tanh_bwd<true>, dim3((unsigned)blocks), dim3(kLinThreads), 0, as_stream(stream), a);
    } else {
        ASAC_LAUNCH(k_linear_tanh_bwd<false>, dim3((unsigned)blocks), dim3(kLinThreads), 0, as_stream(stream), a);
        if (accumulate != ASAC_LINEAR_SUM_DEFER)
            xty_reduce_launch(workspace, (int)blocks, O * K, O, grad_params, grad_params + O * K, accumulate, as_stream(stream));
    }
    return finish_launch("asac_linear_tanh_backward");
}

int asac_linear_tanh_backward(const float* x, int64_t x_row_stride, const float* weight, const float* y,
                              const float* grad_y, int64_t N, int K, int O, float* grad_x, float* grad_params,
                              int accumulate, float* workspace, void* stream) {
    return asac_linear_tanh_backward2(x, x_row_stride, K, nullptr, 0, 0, weight, y, grad_y, 1, 1, 0, N, O, grad_x, nullptr,
                                      grad_params, accumulate, workspace, stream);
}

}  // extern "C"
