// The Linear layers AROUND the multi-head attention core (csrc/attn_mh.hip), over the rows of a batch of sampled windows, on
// f32 MFMA — `MultiheadAttention.forward` (reference algorithm/nn_models/layers/seq_layers.py:239-333) at the widths of the
// reference's environments (`EpisodeMultiheadAttention(64, …)`: envs/gym/toy_queue/nn_attn.py:28-45):
//   * q / k / v projections of one input as ONE launch (three nn.Linear(E, E): three library GEMMs of 9 216 x 64 x 64, each
//     at the launch floor), the query projected for the last `tail` positions of a window only (the episode blocks' cut query);
//     backward: the input gradient of the three as one launch (three GEMMs and two accumulations);
//   * the output ResBlock  y = (x + gelu(x W^T + b)) * row_scale  (LinearLayers(E, E, depth 1) + the dead-row / padded-row
//     factor: a GEMM and three elementwise launches), backward to grad_x and the pre-activation gradient (whose products
//     over the rows — weight and bias gradients — are csrc/xty.hip's).
//
// A workgroup (4 waves) owns 16 rows; wave w the output feature tiles w, w + 4, ...  The ROWS are the N dimension of the
// MFMA (B operand = 16 bytes of a row's features, as they lie in memory), the weights the A operand: forward a weight row's
// 16 bytes, backward four strided words of a weight column; the result tile holds four consecutive features of a row per
// lane -> 16-byte stores.  Everything is L2-resident (weights 16 KB a matrix); the launches are latency, not bandwidth.
#include "asac_common.h"
#include "asac_gelu.h"

namespace asac {
namespace rowsp {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kWaves = 4, kThreads = 64 * kWaves;
constexpr int kMaxJobs = 3;

#define RP_MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
    c = RP_MF(a[0], b[0], c);
    c = RP_MF(a[1], b[1], c);
    c = RP_MF(a[2], b[2], c);
    c = RP_MF(a[3], b[3], c);
    return c;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

struct ProjArgs {
    const float* x; int64_t xs_b, xs_t;      // input rows x[b][t][E] (feature stride 1)
    int32_t B, L, J;
    const float* w[kMaxJobs];                // [E][E] (nn.Linear layout: out x in)
    const float* b[kMaxJobs];                // [E]
    int32_t tail[kMaxJobs];                  // job j covers positions t >= L - tail[j]; its rows are dense [B][tail[j]][E]
    float* y[kMaxJobs];
    const float* g[kMaxJobs];                // backward: gradients of y
    float* gx;                               // [B][L][E] dense
};

struct ResArgs {
    const float* x; int64_t xs;              // [rows][E], row stride xs
    const float* w; const float* b;
    const float* scale;                      // [rows] or NULL
    int64_t rows;
    float* y; float* pre;                    // [rows][E]
    const float* gy; float* gx; float* gpre;
};

// four strided words of a weight column block: W[n0 + s][k] for s < 4
template <int E>
__device__ __forceinline__ f32x4 wcol4(const float* w, int n0, int k) {
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = w[(n0 + s) * E + k];
    return v;
}

template <int EC>
__global__ void __launch_bounds__(kThreads) k_rows_proj_fwd(const ProjArgs a) {
    constexpr int E = 16 * EC;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    const int64_t rows = (int64_t)a.B * a.L, row = (int64_t)blockIdx.x * 16 + x;
    const bool live = row < rows;
    const int64_t rc = live ? row : rows - 1;
    const int b = (int)(rc / a.L), t = (int)(rc - (int64_t)b * a.L);
    const float* xp = a.x + b * a.xs_b + t * a.xs_t + 4 * q;
    f32x4 xv[EC];
#pragma unroll
    for (int c = 0; c < EC; ++c) xv[c] = live ? ld4(xp + 16 * c) : zero4();
    // one job per workgroup (blockIdx.y): the launch is a dependent chain per wave (rows -> weights -> products -> store), not
    // bandwidth — three times the waves, a third of the chain each (10 -> 6 us at 9 216 rows x 64)
    for (int nt = wv; nt < EC; nt += kWaves) {
#pragma unroll
        for (int j = 0; j < kMaxJobs; ++j) {
            if (j != (int)blockIdx.y) continue;
            const float* wp = a.w[j] + (16 * nt + x) * E + 4 * q;
            f32x4 wa[EC];
#pragma unroll
            for (int c = 0; c < EC; ++c) wa[c] = ld4(wp + 16 * c);
            const f32x4 bias = ld4(a.b[j] + 16 * nt + 4 * q);
            f32x4 acc = zero4();
#pragma unroll
            for (int c = 0; c < EC; ++c) acc = mfma4(wa[c], xv[c], acc);      // acc[r] = y[row x][16 nt + 4 q + r]
            acc += bias;
            const int skip = a.L - a.tail[j];
            if (live && t >= skip) st4(a.y[j] + ((int64_t)b * a.tail[j] + (t - skip)) * E + 16 * nt + 4 * q, acc);
        }
    }
}

// gx[row][k] = sum_j sum_n g_j[row][n] W_j[n][k]   (jobs in order, features in order)
template <int EC>
__global__ void __launch_bounds__(kThreads) k_rows_proj_bwd(const ProjArgs a) {
    constexpr int E = 16 * EC;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    const int64_t rows = (int64_t)a.B * a.L, row = (int64_t)blockIdx.x * 16 + x;
    const bool live = row < rows;
    const int64_t rc = live ? row : rows - 1;
    const int b = (int)(rc / a.L), t = (int)(rc - (int64_t)b * a.L);
    f32x4 gv[kMaxJobs][EC];
#pragma unroll
    for (int j = 0; j < kMaxJobs; ++j) {
        if (j >= a.J) continue;
        const int skip = a.L - a.tail[j];
        const bool on = live && t >= skip;
        const float* gp = a.g[j] + ((int64_t)b * a.tail[j] + (on ? t - skip : 0)) * E + 4 * q;
#pragma unroll
        for (int c = 0; c < EC; ++c) gv[j][c] = on ? ld4(gp + 16 * c) : zero4();
    }
    for (int kt = wv; kt < EC; kt += kWaves) {
        f32x4 acc = zero4();
#pragma unroll
        for (int j = 0; j < kMaxJobs; ++j) {
            if (j >= a.J) continue;
#pragma unroll
            for (int c = 0; c < EC; ++c) acc = mfma4(wcol4<E>(a.w[j], 16 * c + 4 * q, 16 * kt + x), gv[j][c], acc);
        }
        if (live) st4(a.gx + row * E + 16 * kt + 4 * q, acc);
    }
}

template <int EC>
__global__ void __launch_bounds__(kThreads) k_rows_res_fwd(const ResArgs a) {
    constexpr int E = 16 * EC;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + x;
    const bool live = row < a.rows;
    const int64_t rc = live ? row : a.rows - 1;
    const float* xp = a.x + rc * a.xs + 4 * q;
    f32x4 xv[EC];
#pragma unroll
    for (int c = 0; c < EC; ++c) xv[c] = ld4(xp + 16 * c);
    const float s = a.scale ? a.scale[rc] : 1.f;
    for (int nt = wv; nt < EC; nt += kWaves) {
        const float* wp = a.w + (16 * nt + x) * E + 4 * q;
        f32x4 wa[EC];
#pragma unroll
        for (int c = 0; c < EC; ++c) wa[c] = ld4(wp + 16 * c);
        const f32x4 bias = ld4(a.b + 16 * nt + 4 * q), res = ld4(xp + 16 * nt);
        f32x4 acc = zero4();
#pragma unroll
        for (int c = 0; c < EC; ++c) acc = mfma4(wa[c], xv[c], acc);
        acc += bias;
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (gelu_f(acc[r]) + res[r]) * s;
        if (live) {
            st4(a.y + row * E + 16 * nt + 4 * q, y);
            st4(a.pre + row * E + 16 * nt + 4 * q, acc);
        }
    }
}

// g = gy * scale;  gpre = g * gelu'(pre);  gx = g + gpre W
template <int EC>
__global__ void __launch_bounds__(kThreads) k_rows_res_bwd(const ResArgs a) {
    constexpr int E = 16 * EC;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + x;
    const bool live = row < a.rows;
    const int64_t rc = live ? row : a.rows - 1;
    const float s = a.scale ? a.scale[rc] : 1.f;
    const float* gyp = a.gy + rc * E + 4 * q;
    const float* prp = a.pre + rc * E + 4 * q;
    f32x4 gp[EC];
#pragma unroll
    for (int c = 0; c < EC; ++c) {
        const f32x4 g = ld4(gyp + 16 * c) * s, z = ld4(prp + 16 * c);
#pragma unroll
        for (int r = 0; r < 4; ++r) gp[c][r] = live ? g[r] * gelu_grad(z[r]) : 0.f;
        if (live && (c & (kWaves - 1)) == wv) st4(a.gpre + row * E + 16 * c + 4 * q, gp[c]);
    }
    for (int kt = wv; kt < EC; kt += kWaves) {
        f32x4 acc = zero4();
#pragma unroll
        for (int c = 0; c < EC; ++c) acc = mfma4(wcol4<E>(a.w, 16 * c + 4 * q, 16 * kt + x), gp[c], acc);
        if (live) st4(a.gx + row * E + 16 * kt + 4 * q, ld4(gyp + 16 * kt) * s + acc);
    }
}

struct AffineArgs {
    const float* x; int64_t xs;              // [rows][K], row stride xs
    const float* w; const float* b;          // [N][K], [N]
    int64_t rows;
    int32_t K, N;
    float* y;                                // [rows][N] dense
    float* pre;                              // GELU form: the pre-activation, saved for the backward
};

// y = x W^T + b for a NARROW input (K <= 64: observation ++ action in front of a recurrent layer — the input products
// `x W_ih^T + b_ih` of all steps, reference seq_layers.py:14-114 through nn.GRU) and a wide output (N = 3 H): the library
// GEMM takes 13-14 us for 20 736 x 8 -> 192; the work is writing 16 MB.  A workgroup owns 16 rows, its waves the output
// tiles w, w + 4, ...; the x tile (K / 4 scalars a lane) is the B operand of every tile.
template <int STEPS, bool GELU>      // k-steps of four: K <= 4 STEPS;  GELU: y = gelu(x W^T + b), the sum itself to `pre`
__global__ void __launch_bounds__(kThreads) k_rows_affine(const AffineArgs a) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + x;
    const bool live = row < a.rows;
    const float* xp = a.x + (live ? row : a.rows - 1) * a.xs;
    float xv[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int k = 4 * s + q;
        const float got = xp[min(k, a.K - 1)];
        xv[s] = k < a.K ? got : 0.f;
    }
    const int NT = a.N >> 4;
    // TB of the wave's tiles at a time: their weight operands requested together with the rows (one round trip), then the
    // products and the stores
    constexpr int TB = STEPS <= 4 ? 4 : 1;
    for (int nt0 = wv; nt0 < NT; nt0 += TB * kWaves) {
        float wq[TB][STEPS];
        f32x4 bias[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int nt = min(nt0 + j * kWaves, NT - 1);
            const float* wp = a.w + (int64_t)(16 * nt + x) * a.K;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int k = 4 * s + q;
                const float got = wp[min(k, a.K - 1)];
                wq[j][s] = k < a.K ? got : 0.f;
            }
            bias[j] = ld4(a.b + 16 * nt + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int nt = nt0 + j * kWaves;
            f32x4 acc = zero4();
#pragma unroll
            for (int s = 0; s < STEPS; ++s) acc = RP_MF(wq[j][s], xv[s], acc);
            acc += bias[j];
            if (live && nt < NT) {
                if (GELU) {
                    st4(a.pre + row * a.N + 16 * nt + 4 * q, acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = gelu_f(acc[r]);
                }
                st4(a.y + row * a.N + 16 * nt + 4 * q, acc);
            }
        }
    }
}

inline bool width_ok(int E) { return E == 32 || E == 64 || E == 128; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace rowsp
}  // namespace asac

using namespace asac;
using namespace asac::rowsp;

#define RP_LAUNCH(kernel, E, grid, stream, args)                                                    \
    do {                                                                                            \
        if ((E) == 32) ASAC_LAUNCH(kernel<2>, grid, dim3(kThreads), 0, stream, args);               \
        else if ((E) == 64) ASAC_LAUNCH(kernel<4>, grid, dim3(kThreads), 0, stream, args);          \
        else ASAC_LAUNCH(kernel<8>, grid, dim3(kThreads), 0, stream, args);                         \
    } while (0)

extern "C" {

int asac_rows_proj_supported(int width) { return width_ok(width); }

int asac_rows_proj_forward(const float* x, int64_t x_stride_b, int64_t x_stride_t, int batch, int window, int width, int n_jobs,
                           const float* const* weights, const float* const* biases, const int* tails, float* const* outs,
                           void* stream) {
    if (!x || !weights || !biases || !tails || !outs || batch <= 0 || window <= 0 || !width_ok(width) || n_jobs < 1 ||
        n_jobs > kMaxJobs || !aligned16(x) || (x_stride_b & 3) || (x_stride_t & 3) || x_stride_t < width)
        return bad_arg("asac_rows_proj_forward");
    ProjArgs a{};
    a.x = x, a.xs_b = x_stride_b, a.xs_t = x_stride_t, a.B = batch, a.L = window, a.J = n_jobs;
    for (int j = 0; j < n_jobs; ++j) {
        if (!weights[j] || !biases[j] || !outs[j] || tails[j] < 1 || tails[j] > window || !aligned16(weights[j]) ||
            !aligned16(biases[j]) || !aligned16(outs[j]))
            return bad_arg("asac_rows_proj_forward: job");
        a.w[j] = weights[j], a.b[j] = biases[j], a.tail[j] = tails[j], a.y[j] = outs[j];
    }
    const dim3 grid((unsigned)(((int64_t)batch * window + 15) / 16), (unsigned)n_jobs);
    RP_LAUNCH(k_rows_proj_fwd, width, grid, as_stream(stream), a);
    return finish_launch("asac_rows_proj_forward");
}

int asac_rows_proj_backward(const float* const* grads, const int* tails, int n_jobs, const float* const* weights, int batch,
                            int window, int width, float* grad_x, void* stream) {
    if (!grads || !tails || !weights || !grad_x || batch <= 0 || window <= 0 || !width_ok(width) || n_jobs < 1 ||
        n_jobs > kMaxJobs || !aligned16(grad_x))
        return bad_arg("asac_rows_proj_backward");
    ProjArgs a{};
    a.B = batch, a.L = window, a.J = n_jobs, a.gx = grad_x;
    for (int j = 0; j < n_jobs; ++j) {
        if (!weights[j] || !grads[j] || tails[j] < 1 || tails[j] > window || !aligned16(grads[j]))
            return bad_arg("asac_rows_proj_backward: job");
        a.w[j] = weights[j], a.g[j] = grads[j], a.tail[j] = tails[j];
    }
    const dim3 grid((unsigned)(((int64_t)batch * window + 15) / 16));
    RP_LAUNCH(k_rows_proj_bwd, width, grid, as_stream(stream), a);
    return finish_launch("asac_rows_proj_backward");
}

int asac_rows_resblock_forward(const float* x, int64_t x_row_stride, const float* weight, const float* bias,
                               const float* row_scale, int64_t rows, int width, float* y, float* pre, void* stream) {
    if (!x || !weight || !bias || !y || !pre || rows <= 0 || !width_ok(width) || x_row_stride < width || (x_row_stride & 3) ||
        !aligned16(x) || !aligned16(weight) || !aligned16(bias) || !aligned16(y) || !aligned16(pre))
        return bad_arg("asac_rows_resblock_forward");
    ResArgs a{};
    a.x = x, a.xs = x_row_stride, a.w = weight, a.b = bias, a.scale = row_scale, a.rows = rows, a.y = y, a.pre = pre;
    const dim3 grid((unsigned)((rows + 15) / 16));
    RP_LAUNCH(k_rows_res_fwd, width, grid, as_stream(stream), a);
    return finish_launch("asac_rows_resblock_forward");
}

int asac_rows_resblock_backward(const float* grad_y, const float* pre, const float* weight, const float* row_scale,
                                int64_t rows, int width, float* grad_x, float* grad_pre, void* stream) {
    if (!grad_y || !pre || !weight || !grad_x || !grad_pre || rows <= 0 || !width_ok(width) || !aligned16(grad_y) ||
        !aligned16(pre) || !aligned16(grad_x) || !aligned16(grad_pre))
        return bad_arg("asac_rows_resblock_backward");
    ResArgs a{};
    a.gy = grad_y, a.pre = const_cast<float*>(pre), a.w = weight, a.scale = row_scale, a.rows = rows, a.gx = grad_x, a.gpre = grad_pre;
    const dim3 grid((unsigned)((rows + 15) / 16));
    RP_LAUNCH(k_rows_res_bwd, width, grid, as_stream(stream), a);
    return finish_launch("asac_rows_resblock_backward");
}

int asac_rows_affine_supported(int K, int N) { return K >= 1 && K <= 64 && N >= 16 && N <= 1024 && (N & 15) == 0; }

static int affine_launch(const char* what, const float* x, int64_t x_row_stride, int K, const float* weight, const float* bias,
                         int64_t rows, int N, float* y, float* pre, void* stream) {
    if (!x || !weight || !bias || !y || rows <= 0 || !asac_rows_affine_supported(K, N) || x_row_stride < K || !aligned16(bias) ||
        !aligned16(y) || (pre && !aligned16(pre)))
        return bad_arg(what);
    AffineArgs a{};
    a.x = x, a.xs = x_row_stride, a.w = weight, a.b = bias, a.rows = rows, a.K = K, a.N = N, a.y = y, a.pre = pre;
    const dim3 grid((unsigned)((rows + 15) / 16));
    hipStream_t s = as_stream(stream);
    if (pre) {
        if (K <= 8) ASAC_LAUNCH((k_rows_affine<2, true>), grid, dim3(kThreads), 0, s, a);
        else if (K <= 16) ASAC_LAUNCH((k_rows_affine<4, true>), grid, dim3(kThreads), 0, s, a);
        else ASAC_LAUNCH((k_rows_affine<16, true>), grid, dim3(kThreads), 0, s, a);
    } else {
        if (K <= 8) ASAC_LAUNCH((k_rows_affine<2, false>), grid, dim3(kThreads), 0, s, a);
        else if (K <= 16) ASAC_LAUNCH((k_rows_affine<4, false>), grid, dim3(kThreads), 0, s, a);
        else ASAC_LAUNCH((k_rows_affine<16, false>), grid, dim3(kThreads), 0, s, a);
    }
    return finish_launch(what);
}

int asac_rows_affine_forward(const float* x, int64_t x_row_stride, int K, const float* weight, const float* bias, int64_t rows,
                             int N, float* y, void* stream) {
    return affine_launch("asac_rows_affine_forward", x, x_row_stride, K, weight, bias, rows, N, y, nullptr, stream);
}

int asac_rows_affine_gelu_forward(const float* x, int64_t x_row_stride, int K, const float* weight, const float* bias,
                                  int64_t rows, int N, float* y, float* pre, void* stream) {
    if (!pre) return bad_arg("asac_rows_affine_gelu_forward");
    return affine_launch("asac_rows_affine_gelu_forward", x, x_row_stride, K, weight, bias, rows, N, y, pre, stream);
}

}  // extern "C"
