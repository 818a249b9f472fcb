// y = tanh(x W^T + b) over the rows of a sampled window, forward and backward: the state head the reference's
// representation plugins put after their encoders (`nn.Sequential(nn.Linear(n, 8), nn.Tanh())`,
// tests/nn_conv_vanilla.py:11-14, tests/nn_conv_attn.py, envs/test/nn_rnn.py).  At [B*L, <= 64] -> <= 16 the
// library GEMMs are pure latency (the weight-gradient GEMM reduces 4 608 rows into 8 x 18 numbers: 18 us, plus a
// bias reduction, the tanh launches and two gradient accumulations); here a pass is one launch.
//
// HBM-bound by construction (a row is read once, <= 2 K flop per row): one lane per row, the weights broadcast
// from LDS.  Backward: per-workgroup partial parameter gradients, summed in workgroup order by the last
// workgroup to finish (deterministic; no float atomics).
#include "asac_common.h"

namespace asac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLinMaxK = ASAC_LINEAR_TANH_MAX_IN;      // 64
constexpr int kLinMaxO = ASAC_LINEAR_TANH_MAX_OUT;     // 16
constexpr int kLinRows = 128;                          // rows (= lanes) per workgroup

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

// W^T padded to 16 columns: wt[k][o] (zero beyond O), so a lane's 16 accumulators read four float4 per k
__device__ __forceinline__ void stage_wt(const LinArgs& a, float* wt) {
    for (int i = threadIdx.x; i < a.K * kLinMaxO; i += blockDim.x) {
        const int k = i / kLinMaxO, o = i - k * kLinMaxO;
        wt[i] = o < a.O ? a.w[o * a.K + k] : 0.f;
    }
}

__global__ __launch_bounds__(kLinRows) void k_linear_tanh_fwd(const LinArgs a) {
    __shared__ __attribute__((aligned(16))) float wt[kLinMaxK * kLinMaxO];
    __shared__ float bias[kLinMaxO];
    stage_wt(a, wt);
    if (threadIdx.x < kLinMaxO) bias[threadIdx.x] = (int)threadIdx.x < a.O ? a.b[threadIdx.x] : 0.f;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * kLinRows + threadIdx.x;
    if (r >= a.N) return;
    float acc[kLinMaxO];
#pragma unroll
    for (int o = 0; o < kLinMaxO; ++o) acc[o] = 0.f;
    const float* xr = a.x + (a.x_T ? (r / a.x_T) * a.x_sb + (r % a.x_T) * a.x_stride : r * a.x_stride);
    const float* xr1 = a.x1 ? a.x1 + r * a.x1_stride - a.K0 : xr;      // (indexed by k like the first part)
    for (int k = 0; k < a.K; ++k) {
        const float xv = k < a.K0 ? xr[k] : xr1[k];
        const float4* w4 = reinterpret_cast<const float4*>(wt + k * kLinMaxO);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = w4[q];
            acc[4 * q + 0] += xv * w.x;
            acc[4 * q + 1] += xv * w.y;
            acc[4 * q + 2] += xv * w.z;
            acc[4 * q + 3] += xv * w.w;
        }
    }
    float* yr = a.y + r * a.O;
#pragma unroll
    for (int o = 0; o < kLinMaxO; ++o)
        if (o < a.O) yr[o] = tanhf(acc[o] + bias[o]);
}

__global__ __launch_bounds__(kLinRows) void k_linear_tanh_bwd(const LinArgs a) {
    __shared__ __attribute__((aligned(16))) float wt[kLinMaxK * kLinMaxO];
    __shared__ float gs[kLinRows * kLinMaxO];          // g = gy * (1 - y^2), [row][16]
    __shared__ float xs[kLinRows * (kLinMaxK + 1)];    // the workgroup's input rows and a column of ones, [row][K + 1]
    __shared__ bool last;
    stage_wt(a, wt);
    const int64_t r0 = (int64_t)blockIdx.x * kLinRows;
    const int rows = (int)min((int64_t)kLinRows, a.N - r0);
    const int64_t r = r0 + threadIdx.x;
    const bool live = (int)threadIdx.x < rows;
    float g[kLinMaxO];
    const int64_t gy_row = live ? r / a.gy_window : 0;
    const bool gy_on = live && (a.gy_window == 1 || (int)(r - gy_row * a.gy_window) == a.gy_position);
    const int64_t gy_rows = a.N / a.gy_window;
#pragma unroll
    for (int o = 0; o < kLinMaxO; ++o) {
        float v = 0.f;
        if (gy_on && o < a.O) {
            const float yv = a.y[r * a.O + o];
            float gsum = a.gy[gy_row * a.O + o];
            for (int e = 1; e < a.gy_members; ++e) gsum += a.gy[(e * gy_rows + gy_row) * a.O + o];     // member order
            v = gsum * (1.f - yv * yv);
        }
        g[o] = v;
        gs[threadIdx.x * kLinMaxO + o] = v;
    }
    const int XS = a.K + 1;
    {
        const int64_t rs = live ? r : r0;
        const float* xr = a.x + (a.x_T ? (rs / a.x_T) * a.x_sb + (rs % a.x_T) * a.x_stride : rs * a.x_stride);
        const float* xr1 = a.x1 ? a.x1 + rs * a.x1_stride - a.K0 : xr;
        for (int k = 0; k < a.K; ++k) xs[threadIdx.x * XS + k] = live ? (k < a.K0 ? xr[k] : xr1[k]) : 0.f;
        xs[threadIdx.x * XS + a.K] = 1.f;
    }
    __syncthreads();
    if ((a.gx || a.gx1) && live) {
        float* gxr = a.gx ? a.gx + r * a.K0 : nullptr;
        float* gxr1 = a.gx1 ? a.gx1 + r * (a.K - a.K0) - a.K0 : nullptr;
        for (int k = a.gx ? 0 : a.K0; k < (a.gx1 ? a.K : a.K0); ++k) {
            const float4* w4 = reinterpret_cast<const float4*>(wt + k * kLinMaxO);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = w4[q];
                s += g[4 * q + 0] * w.x;
                s += g[4 * q + 1] * w.y;
                s += g[4 * q + 2] * w.z;
                s += g[4 * q + 3] * w.w;
            }
            if (k < a.K0) gxr[k] = s;
            else gxr1[k] = s;
        }
    }
    // partial parameter gradients of this workgroup, G^T [O x rows] times (X | 1) [rows x (K + 1)] on the matrix
    // cores: element t = o * (K + 1) + k, column K (the ones) being the bias.  16x16x4 f32 MFMA, A lane (i, kk) =
    // g[row 4 step + kk][o = i], B lane (j, kk) = x[row 4 step + kk][column 16 tile + j]; a wave takes every
    // other column tile.  Dead rows carry g = 0.
    const int P = a.O * (a.K + 1);
    float* mine = a.partial + (int64_t)blockIdx.x * P;
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
        const int tiles = (a.K + 1 + 15) / 16;
        for (int tile = wave; tile < tiles; tile += kLinRows / 64) {
            const int col = tile * 16 + i, colc = min(col, a.K);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const bool real = col <= a.K;
            for (int s0 = 0; s0 < kLinRows / 4; s0 += 8) {        // eight steps' operands in flight, two chains
                float av[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = 4 * (s0 + u) + kk;
                    av[u] = gs[row * kLinMaxO + i];
                    xv[u] = xs[row * XS + colc];
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
                if (o < a.O && col <= a.K) mine[o * (a.K + 1) + col] = acc[rr];
            }
        }
    }
    // the last workgroup to arrive sums the partials in workgroup order
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // (plain loads: this workgroup has not touched the other workgroups' partials before, and the fences order them
    // after the arrival count; sixteen are requested at a time, then added in workgroup order)
    const int nb = (int)gridDim.x;
    for (int t = threadIdx.x; t < P; t += kLinRows) {
        const int o = t / (a.K + 1), k = t - o * (a.K + 1);
        const float* col = a.partial + t;
        float s = 0.f;
        for (int b0 = 0; b0 < nb; b0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = col[(int64_t)min(b0 + u, nb - 1) * P];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += b0 + u < nb ? v[u] : 0.f;
        }
        float* dst = a.gp + (k < a.K ? o * a.K + k : a.O * a.K + o);
        *dst = a.accumulate ? *dst + s : s;
    }
    if (threadIdx.x == 0) *a.counter = 0u;       // ready for the next launch
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
    ASAC_LAUNCH(k_linear_tanh_fwd, dim3((unsigned)((N + kLinRows - 1) / kLinRows)), dim3(kLinRows), 0, as_stream(stream), a);
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
    ASAC_LAUNCH(k_linear_tanh_bwd, dim3((unsigned)blocks), dim3(kLinRows), 0, as_stream(stream), a);
    return finish_launch("asac_linear_tanh_backward");
}

int asac_linear_tanh_backward(const float* x, int64_t x_row_stride, const float* weight, const float* y,
                              const float* grad_y, int64_t N, int K, int O, float* grad_x, float* grad_params,
                              int accumulate, float* workspace, void* stream) {
    return asac_linear_tanh_backward2(x, x_row_stride, K, nullptr, 0, 0, weight, y, grad_y, 1, 1, 0, N, O, grad_x, nullptr,
                                      grad_params, accumulate, workspace, stream);
}

}  // extern "C"
