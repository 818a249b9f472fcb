// A Linear layer with a WIDE input on f32 MFMA:  y = act(x W^T + b),  x [R][K] with K in the thousands, W [N][K], N <= 128 —
// the first layer of the dense head behind a flattened convolution map: `ConvLayers(84, 84, C, 'simple')` hands 32 x 9 x 9 =
// 2 592 features per frame to `LinearLayers(2592, 64, ...)` (reference nn_models/layers/image_layers.py:188-227 with
// linear_layers.py:24-119; every 84 x 84 plugin under envs/), over the 1 024 ... 2 304 frames of a train step's windows.
// The fused dense-stack kernels (mlp.hip) stop at 128 inputs; through the library this layer was a GEMM + a GELU launch
// forward and two GEMMs, a split reduction for the bias and an elementwise launch backward, each followed by `add`s of
// autograd (BASELINE-like configuration cfg4_84: 6 GEMMs, 9 adds, 3 reductions a step).
//
//   forward   split over K: workgroup (32-row block, K chunk) -> partial [chunk][R][N]; a second launch sums the chunks in
//             order, adds the bias, applies GELU (saving the pre-activation when training).  Operands are read where they
//             lie, 16 bytes per lane: lane (m, g) of a wave reads x[row m][k0 + 4 g .. + 3] and W[n][k0 + 4 g .. + 3], step
//             t of a quad of four MFMAs uses component t (the reduction index a (g, t) stands for is k0 + 4 g + t on both sides)
//   d input   dx [R][K] = dpre W with dpre = g * gelu'(pre) formed once per 16-row tile in LDS (and written out for the
//             parameter gradients); a workgroup keeps a 64-column block of W in registers and walks row tiles
//   d params  dW [N][K] += dpre^T x over row slices -> partials, summed in slice order; db = column sums of dpre
// Every sum has a fixed order: deterministic.  C ABI in include/asac_hip.h.
#include "asac_common.h"
#include "asac_gelu.h"

namespace asac {
namespace wide {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kThreads = 256;
constexpr int kMaxN = 128;

template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* gptr(const T* p) {
    return (const __attribute__((address_space(1))) T*)p;
}

struct FwdArgs {
    const float* x; int64_t xs;
    const float* w; const float* b;
    float* part;            // [splits][R][N]
    float* y; float* pre;
    int64_t R;
    int32_t K, N, quads_per_split, splits, act;
};

// ---- forward, one K chunk of a 32-row block -------------------------------------------------------------------------
// wave w: row tile w & 1, column tiles (w >> 1) * NTW .. + NTW - 1   (NTW = N / 32: 2 for N = 64)
template <int NTW>
__global__ __launch_bounds__(kThreads) void k_wide_fwd(const FwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int rt = wave & 1, ct0 = (wave >> 1) * NTW;
    const int64_t row0 = (int64_t)blockIdx.x * 32 + rt * 16;
    const int64_t row = min(row0 + m, a.R - 1);
    const int q0 = blockIdx.y * a.quads_per_split, q1 = min(q0 + a.quads_per_split, a.K / 16);
    const auto* xp = gptr(reinterpret_cast<const f32x4*>(a.x + row * a.xs) + g);
    const __attribute__((address_space(1))) f32x4* wp[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int n = min((ct0 + t) * 16 + m, a.N - 1);
        wp[t] = gptr(reinterpret_cast<const f32x4*>(a.w + (int64_t)n * a.K) + g);
    }
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // WIDE_DEPTH quads in flight: slot d holds quad q + d; its successor q + d + DEPTH is requested as soon as the slot's
    // MFMAs are issued (slots beyond the chunk hold zeros: no branch around an MFMA)
#ifndef WIDE_DEPTH
#define WIDE_DEPTH 2
#endif
    constexpr int D = WIDE_DEPTH;
    f32x4 xa[D], wb[D][NTW];
    auto fetch = [&](int d, int q) {
        const bool on = q < q1;
        const int qc = min(q, q1 - 1);
        const f32x4 xv = xp[qc * 4];
        xa[d] = on ? xv : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NTW; ++t) wb[d][t] = wp[t][qc * 4];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, q0 + d);
    for (int q = q0; q < q1; q += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[d][j], wb[d][t][j], acc[t], 0, 0, 0);
            fetch(d, q + d + D);
        }
    }
    // D[row 4 g + r][column m] of each column tile -> the chunk's partial slab
    float* out = a.part + (int64_t)blockIdx.y * a.R * a.N;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int n = (ct0 + t) * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t rr = row0 + 4 * g + r;
            if (rr < a.R && n < a.N) out[rr * a.N + n] = acc[t][r];
        }
    }
}

// ---- forward, second launch: chunks summed in order, bias, activation ---------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_wide_finish(const FwdArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x, total = a.R * a.N;
    if (i >= total) return;
    float s = 0.f;
    for (int c = 0; c < a.splits; ++c) s += a.part[(int64_t)c * total + i];
    const float z = s + a.b[i % a.N];
    if (a.pre) a.pre[i] = z;
    a.y[i] = a.act ? gelu_f(z) : z;
}

// ---- backward, input gradient (and dpre) -----------------------------------------------------------------------------
struct DxArgs {
    const float* g; const float* pre;        // [R][N]
    const float* w;                          // [N][K]
    float* dpre;                             // [R][N] out
    float* dx; int64_t dxs;                  // [R][K] out (or NULL)
    int64_t R, tiles_per_block;
    int32_t K, N, act;
};

// workgroup (64-column block of K, slice of row tiles): W[:, block] in registers (wave w: 16 columns), per 16-row tile
// dpre = g * act'(pre) formed once in LDS, A operand = 16 bytes of a dpre row (reduction index 16 j + 4 g + t <-> step 4 j + t)
template <int NQ>       // N / 16 quads of the reduction over the features
__global__ __launch_bounds__(kThreads) void k_wide_bwd_dx(const DxArgs a) {
    __shared__ __attribute__((aligned(16))) float s_dp[16 * (kMaxN + 4)];       // row pitch N + 4: the 16 rows' reads on distinct banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
    const int kc = (int)blockIdx.x * 64 + wave * 16 + m;          // this lane's column of W / dx
    const bool k_on = kc < a.K && a.dx != nullptr;
    float wr[NQ][4];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) wr[j][t] = k_on ? gptr(a.w)[(int64_t)(16 * j + 4 * g + t) * a.K + kc] : 0.f;
    const int64_t n_tiles = (a.R + 15) / 16;
    const int64_t t0 = (int64_t)blockIdx.y * a.tiles_per_block, t1 = min(t0 + a.tiles_per_block, n_tiles);
    const int N = a.N, P = a.N + 4;
    for (int64_t tile = t0; tile < t1; ++tile) {
        const int64_t row0 = tile * 16;
        // dpre tile: 16 rows x N, four consecutive features per thread step
        for (int e = threadIdx.x * 4; e < 16 * N; e += kThreads * 4) {
            const int r = e / N, c = e - r * N;
            const int64_t row = row0 + r;
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (row < a.R) {
                const f32x4 gv = *gptr(reinterpret_cast<const f32x4*>(a.g + row * N + c));
                if (a.act) {
                    const f32x4 pv = *gptr(reinterpret_cast<const f32x4*>(a.pre + row * N + c));
                    d = f32x4{gv.x * gelu_grad(pv.x), gv.y * gelu_grad(pv.y), gv.z * gelu_grad(pv.z), gv.w * gelu_grad(pv.w)};
                } else {
                    d = gv;
                }
                if (blockIdx.x == 0) *reinterpret_cast<f32x4*>(a.dpre + row * N + c) = d;
            }
            *reinterpret_cast<f32x4*>(s_dp + r * P + c) = d;
        }
        __syncthreads();
        if (a.dx) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(s_dp + m * P + 16 * j + 4 * g);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], wr[j][t], acc, 0, 0, 0);
            }
            if (kc < a.K) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + 4 * g + r;
                    if (row < a.R) a.dx[row * a.dxs + kc] = acc[r];
                }
            }
        }
        __syncthreads();
    }
}

// ---- backward, parameter gradients -----------------------------------------------------------------------------------
struct DwArgs {
    const float* dpre;                       // [R][N]
    const float* x; int64_t xs;              // [R][K]
    float* part;                             // [slices][N][K]
    float* dw; float* db;
    int64_t R, rows_per_slice;
    int32_t K, N, slices, accumulate;
};

// workgroup (128-column block of K, row slice): wave w: columns 32 w .. + 31 (two tiles) against all NT feature tiles;
// per 4 rows a lane loads one dpre element per feature tile and one x element per column tile
template <int NT>
__global__ __launch_bounds__(kThreads) void k_wide_bwd_dw(const DwArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int kc0 = (int)blockIdx.x * 128 + wave * 32;
    const int64_t r0 = (int64_t)blockIdx.y * a.rows_per_slice, r1 = min(a.R, r0 + a.rows_per_slice);
    f32x4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool on0 = kc0 + c < a.K, on1 = kc0 + 16 + c < a.K;
    const auto* xg = gptr(a.x);
    const auto* dg = gptr(a.dpre);
    for (int64_t row = r0; row < r1; row += 16) {
        float av[4][NT], bv[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rr = row + 4 * s + q;
            const bool on = rr < r1;
            const int64_t rc = on ? rr : r1 - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float v = dg[rc * a.N + 16 * t + c];
                av[s][t] = on ? v : 0.f;
            }
            const float b0 = xg[rc * a.xs + (on0 ? kc0 + c : 0)], b1 = xg[rc * a.xs + (on1 ? kc0 + 16 + c : 0)];
            bv[s][0] = (on && on0) ? b0 : 0.f, bv[s][1] = (on && on1) ? b1 : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[s][0], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[s][1], acc[t][1], 0, 0, 0);
            }
    }
    // D[feature 4 q + r][column c] of tile (t, u)
    float* out = a.part + (int64_t)blockIdx.y * a.N * a.K;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kc = kc0 + 16 * u + c;
            if (kc < a.K) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(int64_t)(16 * t + 4 * q + r) * a.K + kc] = acc[t][u][r];
            }
        }
}

// slices summed in order into dW; the last blocks form db = column sums of dpre (64 row slices per feature, then the slices)
__global__ __launch_bounds__(kThreads) void k_wide_dw_reduce(const DwArgs a) {
    const int64_t total = (int64_t)a.N * a.K;
    const int64_t w_blocks = (total + kThreads - 1) / kThreads;
    if ((int64_t)blockIdx.x < w_blocks) {
        const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
        if (i >= total) return;
        float s = 0.f;
        for (int k = 0; k < a.slices; ++k) s += a.part[(int64_t)k * total + i];
        a.dw[i] = a.accumulate ? a.dw[i] + s : s;
        return;
    }
    if (!a.db) return;
    __shared__ float red[kThreads];
    const int blk = (int)(blockIdx.x - w_blocks);           // four features per block, 64 row lanes each
    const int n = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float s = 0.f;
    if (n < a.N)
        for (int64_t r = lane; r < a.R; r += 64) s += a.dpre[r * a.N + n];
    red[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && n < a.N) {
        float t = 0.f;
        for (int k = 0; k < 64; ++k) t += red[(threadIdx.x & ~63) + k];
        a.db[n] = a.accumulate ? a.db[n] + t : t;
    }
}

inline int fwd_splits(int64_t R, int K) {
    const int64_t row_blocks = (R + 31) / 32;
    const int quads = K / 16;
    int64_t s = (512 + row_blocks - 1) / row_blocks;
    s = s < 1 ? 1 : s;
    s = s > 16 ? 16 : s;
    s = s > quads ? quads : s;
    return (int)s;
}
inline int dw_slices(int64_t R, int K) {
    const int col_blocks = (K + 127) / 128;
    int64_t s = (384 + col_blocks - 1) / col_blocks;
    const int64_t most = (R + 31) / 32;           // at least 32 rows a slice
    s = s > most ? most : s;
    s = s > 16 ? 16 : s;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace wide
}  // namespace asac

using namespace asac;
using namespace asac::wide;

extern "C" {

int asac_rows_wide_supported(int64_t R, int K, int N) {
    return R >= 1 && R < (1LL << 30) && K >= 16 && K <= 16384 && (K & 15) == 0 && (N == 32 || N == 64 || N == 128);
}

// floats of scratch the forward (split-K partials) and the parameter-gradient launch (row-slice partials) need: the larger
int64_t asac_rows_wide_workspace(int64_t R, int K, int N) {
    if (!asac_rows_wide_supported(R, K, N)) return -1;
    const int64_t f = (int64_t)fwd_splits(R, K) * R * N, w = (int64_t)dw_slices(R, K) * N * K;
    return f > w ? f : w;
}

int asac_rows_wide_forward(const float* x, int64_t x_row_stride, int64_t R, int K, const float* w, const float* b, int N,
                           int act, float* y, float* pre, float* workspace, void* stream) {
    if (!x || !w || !b || !y || !workspace || !asac_rows_wide_supported(R, K, N) || x_row_stride < K || (x_row_stride & 3) ||
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15))
        return bad_arg("asac_rows_wide_forward");
    FwdArgs a{};
    a.x = x, a.xs = x_row_stride, a.w = w, a.b = b, a.part = workspace, a.y = y, a.pre = pre, a.R = R, a.K = K, a.N = N, a.act = act;
    a.splits = fwd_splits(R, K);
    a.quads_per_split = (K / 16 + a.splits - 1) / a.splits;
    a.splits = (K / 16 + a.quads_per_split - 1) / a.quads_per_split;       // (no empty chunk)
    const dim3 grid((unsigned)((R + 31) / 32), (unsigned)a.splits);
    hipStream_t s = as_stream(stream);
    if (N == 32) ASAC_LAUNCH(k_wide_fwd<1>, grid, dim3(kThreads), 0, s, a);
    else if (N == 64) ASAC_LAUNCH(k_wide_fwd<2>, grid, dim3(kThreads), 0, s, a);
    else ASAC_LAUNCH(k_wide_fwd<4>, grid, dim3(kThreads), 0, s, a);
    hipLaunchKernelGGL(k_wide_finish, dim3((unsigned)((R * N + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a);
    return finish_launch("asac_rows_wide_forward");
}

int asac_rows_wide_backward_input(const float* grad_y, const float* pre, int act, int64_t R, int K, const float* w, int N,
                                  float* dpre_out, float* dx, int64_t dx_row_stride, void* stream) {
    if (!grad_y || (act && !pre) || !w || !dpre_out || !asac_rows_wide_supported(R, K, N) || (dx && dx_row_stride < K) ||
        ((reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(dpre_out)) & 15))
        return bad_arg("asac_rows_wide_backward_input");
    DxArgs a{};
    a.g = grad_y, a.pre = pre, a.w = w, a.dpre = dpre_out, a.dx = dx, a.dxs = dx_row_stride, a.R = R, a.K = K, a.N = N, a.act = act;
    const int64_t tiles = (R + 15) / 16;
    const int col_blocks = dx ? (K + 63) / 64 : 1;
    int64_t slices = (512 + col_blocks - 1) / col_blocks;
    slices = slices > tiles ? tiles : slices;
    a.tiles_per_block = (tiles + slices - 1) / slices;
    slices = (tiles + a.tiles_per_block - 1) / a.tiles_per_block;
    const dim3 grid((unsigned)col_blocks, (unsigned)slices);
    hipStream_t s = as_stream(stream);
    if (N == 32) ASAC_LAUNCH(k_wide_bwd_dx<2>, grid, dim3(kThreads), 0, s, a);
    else if (N == 64) ASAC_LAUNCH(k_wide_bwd_dx<4>, grid, dim3(kThreads), 0, s, a);
    else ASAC_LAUNCH(k_wide_bwd_dx<8>, grid, dim3(kThreads), 0, s, a);
    return finish_launch("asac_rows_wide_backward_input");
}

int asac_rows_wide_backward_params(const float* dpre, const float* x, int64_t x_row_stride, int64_t R, int K, int N, float* dw,
                                   float* db, int accumulate, float* workspace, void* stream) {
    if (!dpre || !x || !dw || !workspace || !asac_rows_wide_supported(R, K, N) || x_row_stride < K)
        return bad_arg("asac_rows_wide_backward_params");
    DwArgs a{};
    a.dpre = dpre, a.x = x, a.xs = x_row_stride, a.part = workspace, a.dw = dw, a.db = db, a.R = R, a.K = K, a.N = N;
    a.accumulate = accumulate;
    a.slices = dw_slices(R, K);
    a.rows_per_slice = ((R + a.slices - 1) / a.slices + 15) / 16 * 16;
    a.slices = (int)((R + a.rows_per_slice - 1) / a.rows_per_slice);
    const dim3 grid((unsigned)((K + 127) / 128), (unsigned)a.slices);
    hipStream_t s = as_stream(stream);
    if (N == 32) ASAC_LAUNCH(k_wide_bwd_dw<2>, grid, dim3(kThreads), 0, s, a);
    else if (N == 64) ASAC_LAUNCH(k_wide_bwd_dw<4>, grid, dim3(kThreads), 0, s, a);
    else ASAC_LAUNCH(k_wide_bwd_dw<8>, grid, dim3(kThreads), 0, s, a);
    const int64_t w_blocks = ((int64_t)N * K + kThreads - 1) / kThreads;
    // launched once (not under the repeat knob: it may accumulate)
    hipLaunchKernelGGL(k_wide_dw_reduce, dim3((unsigned)(w_blocks + (db ? (N + 3) / 4 : 0))), dim3(kThreads), 0, s, a);
    return finish_launch("asac_rows_wide_backward_params");
}

}  // extern "C"
