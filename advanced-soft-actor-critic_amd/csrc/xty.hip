// Products over the ROWS of two tall matrices on f32 MFMA:  out[m][n] = sum_r x[r][m] * y[r][n]  (+ the column sums of x) —
// the weight / bias gradient of a Linear layer applied to thousands of rows (`torch.nn.Linear` backward: grad_out^T input and
// grad_out.sum(0), reference nn_models/layers/linear_layers.py:24-119 under autograd), the input / recurrent weight gradients
// of the wide GRU (seq_layers.py:14-114) and of the attention projections (seq_layers.py:239-333) at the widths of the
// reference's environments (64 ... 128 features, 9 216 ... 20 736 rows).  The library GEMM picks a 32 x 6-style tile for
// these shapes (M, N <= 128, K = rows) and takes 40-76 us; this is HBM-bound work of a few MB.
//
// A workgroup (4 waves) owns a range of rows and a block of 64 x-columns (one 16-column tile per wave) against all y-columns
// (<= 8 tiles): per 4 rows a lane loads one x element and one y element per tile — the operands as they lie in memory —
// and issues one MFMA per tile; 16 rows in flight.  Per-workgroup partials, summed in workgroup order by a second launch.
#include "asac_common.h"

namespace asac {
namespace xty {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kThreads = 256;
constexpr int kMaxN = 128, kMaxM = 512;
constexpr int kRowsPerIter = 16;

struct Args {
    const float* x; int64_t xs; int32_t M;
    const float* y; int64_t ys; int32_t N;
    int64_t rows, rows_per_block;
    float* part;          // [blocks][M * N + M]
    int32_t blocks;
};

// several products in one launch pair (blockIdx.z = the job): the parameter gradients of the Linear layers of one attention
// block — q / k / v projections and the output block — are four products over the same rows
constexpr int kMaxJobs = 4;
struct MultiArgs {
    Args job[kMaxJobs];
    float* out[kMaxJobs];
    float* colsum[kMaxJobs];
    int32_t accumulate;
};

template <int NT>
__device__ __forceinline__ void xty_body(const Args& a) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, c = l & 15;
    const int m0 = ((int)blockIdx.y * 4 + w) * 16;
    if (m0 >= a.M || (int)blockIdx.x >= a.blocks) return;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block, r1 = min(a.rows, r0 + a.rows_per_block);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    const bool m_on = m0 + c < a.M;
    const float* xp = a.x + m0 + c;
    for (int64_t row = r0; row < r1; row += kRowsPerIter) {
        float av[4], bv[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rr = row + 4 * s + q;
            const bool on = rr < r1;
            av[s] = (on && m_on) ? xp[rr * a.xs] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[s][t] = (on && 16 * t + c < a.N) ? a.y[rr * a.ys + 16 * t + c] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            csum += av[s];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][t], acc[t], 0, 0, 0);
        }
    }
    float* part = a.part + (int64_t)blockIdx.x * ((int64_t)a.M * a.N + a.M);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * q + r, n = 16 * t + c;
            if (m < a.M && n < a.N) part[(int64_t)m * a.N + n] = acc[t][r];
        }
    csum += __shfl_xor(csum, 16, 64);
    csum += __shfl_xor(csum, 32, 64);
    if (q == 0 && m_on) part[(int64_t)a.M * a.N + m0 + c] = csum;
}

template <int NT>
__global__ void __launch_bounds__(kThreads) k_xty(const Args a) {
    xty_body<NT>(a);
}

template <int NT>
__global__ void __launch_bounds__(kThreads) k_xty_multi(const MultiArgs by_value) {
    const ASAC_KARG MultiArgs& A = *static_cast<const ASAC_KARG MultiArgs*>(kernarg_base());
    const Args a = karg_copy(&A.job[blockIdx.z]);
    xty_body<NT>(a);
}

// out[i] (+)= sum over the workgroups' partials in a fixed order: 16 slices of the workgroup list summed side by side (slice
// s takes workgroups s, s + 16, ...: the loads of a thread are independent), then the slices in order; the first M * N entries
// go to `out`, the M after them to `colsum` (skipped when NULL)
constexpr int kRedSlices = 16;
__device__ __forceinline__ void reduce_body(const float* __restrict__ part, int blocks, int mn, int m, float* __restrict__ out,
                                            float* __restrict__ colsum, int accumulate) {
    __shared__ float red[kRedSlices][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < mn + m)
#pragma unroll 8
        for (int g = sl; g < blocks; g += kRedSlices) s += part[(int64_t)g * (mn + m) + i];
    red[sl][o] = s;
    __syncthreads();
    if (sl != 0 || i >= mn + m) return;
    float* dst = i < mn ? out + i : (colsum ? colsum + (i - mn) : nullptr);
    if (!dst) return;
    float t = red[0][o];
#pragma unroll
    for (int q = 1; q < kRedSlices; ++q) t += red[q][o];
    *dst = accumulate ? *dst + t : t;
}

__global__ void __launch_bounds__(64 * kRedSlices) k_xty_reduce(const float* __restrict__ part, int blocks, int mn, int m,
                                                               float* __restrict__ out, float* __restrict__ colsum,
                                                               int accumulate) {
    reduce_body(part, blocks, mn, m, out, colsum, accumulate);
}

__global__ void __launch_bounds__(64 * kRedSlices) k_xty_reduce_multi(const MultiArgs by_value) {
    const ASAC_KARG MultiArgs& A = *static_cast<const ASAC_KARG MultiArgs*>(kernarg_base());
    const int j = blockIdx.y;
    const int M = A.job[j].M, N = A.job[j].N;
    if ((int)blockIdx.x * 64 >= M * N + M) return;
    reduce_body(A.job[j].part, A.job[j].blocks, M * N, M, A.out[j], A.colsum[j], A.accumulate);
}

inline int row_blocks(int64_t rows) {
    int64_t b = (rows + 63) / 64;            // >= 64 rows a workgroup
    return (int)(b < 1 ? 1 : (b > 128 ? 128 : b));
}

}  // namespace xty
}  // namespace asac

namespace asac {
// csrc/linear.hip's partial parameter gradients share the layout: summed by the same slice reduction
void xty_reduce_launch(const float* part, int blocks, int mn, int m, float* out, float* colsum, int accumulate, hipStream_t stream) {
    hipLaunchKernelGGL(xty::k_xty_reduce, dim3((unsigned)((mn + m + 63) / 64)), dim3(64 * xty::kRedSlices), 0, stream, part, blocks, mn, m,
                       out, colsum, accumulate);
}
}  // namespace asac

using namespace asac;
using namespace asac::xty;

extern "C" {

int asac_xty_supported(int64_t rows, int M, int N) { return rows >= 1 && M >= 1 && M <= kMaxM && N >= 1 && N <= kMaxN; }

int64_t asac_xty_workspace(int64_t rows, int M, int N) {
    if (!asac_xty_supported(rows, M, N)) return -1;
    return (int64_t)row_blocks(rows) * ((int64_t)M * N + M);
}

int asac_xty(const float* x, int64_t x_row_stride, int M, const float* y, int64_t y_row_stride, int N, int64_t rows, float* out,
             float* colsum_x, int accumulate, float* workspace, void* stream) {
    if (!x || !y || !out || !workspace || !asac_xty_supported(rows, M, N) || x_row_stride < M || y_row_stride < N)
        return bad_arg("asac_xty");
    Args a{};
    a.x = x, a.xs = x_row_stride, a.M = M, a.y = y, a.ys = y_row_stride, a.N = N, a.rows = rows, a.part = workspace;
    const int blocks = row_blocks(rows);
    a.blocks = blocks;
    a.rows_per_block = ((rows + blocks - 1) / blocks + kRowsPerIter - 1) / kRowsPerIter * kRowsPerIter;
    const dim3 grid((unsigned)blocks, (unsigned)((M + 63) / 64));
    hipStream_t s = as_stream(stream);
    const int nt = (N + 15) / 16;
    if (nt <= 1) ASAC_LAUNCH(k_xty<1>, grid, dim3(kThreads), 0, s, a);
    else if (nt <= 2) ASAC_LAUNCH(k_xty<2>, grid, dim3(kThreads), 0, s, a);
    else if (nt <= 4) ASAC_LAUNCH(k_xty<4>, grid, dim3(kThreads), 0, s, a);
    else ASAC_LAUNCH(k_xty<8>, grid, dim3(kThreads), 0, s, a);
    const int total = M * N + M;
    // launched once (not under the measurement repeat knob: it may accumulate)
    hipLaunchKernelGGL(k_xty_reduce, dim3((unsigned)((total + 63) / 64)), dim3(64 * kRedSlices), 0, s, workspace, blocks, M * N, M, out,
                       colsum_x, accumulate);
    return finish_launch("asac_xty");
}

int64_t asac_xty_multi_workspace(int n_jobs, const int64_t* rows, const int* M, const int* N) {
    if (n_jobs < 1 || n_jobs > kMaxJobs || !rows || !M || !N) return -1;
    int64_t total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int64_t w = asac_xty_workspace(rows[j], M[j], N[j]);
        if (w < 0) return -1;
        total += (w + 3) / 4 * 4;
    }
    return total;
}

int asac_xty_multi(int n_jobs, const float* const* x, const int64_t* x_row_strides, const int* M, const float* const* y,
                   const int64_t* y_row_strides, const int* N, const int64_t* rows, float* const* outs, float* const* colsums,
                   int accumulate, float* workspace, void* stream) {
    if (n_jobs < 1 || n_jobs > kMaxJobs || !x || !x_row_strides || !M || !y || !y_row_strides || !N || !rows || !outs || !workspace)
        return bad_arg("asac_xty_multi");
    MultiArgs A{};
    A.accumulate = accumulate;
    int max_blocks = 0, max_mt = 0, max_nt = 0, max_total = 0;
    float* ws = workspace;
    for (int j = 0; j < n_jobs; ++j) {
        if (!x[j] || !y[j] || !outs[j] || !asac_xty_supported(rows[j], M[j], N[j]) || x_row_strides[j] < M[j] || y_row_strides[j] < N[j])
            return bad_arg("asac_xty_multi: job");
        for (int i = 0; i < j; ++i)
            if (outs[i] == outs[j] || (colsums && colsums[j] && colsums[i] == colsums[j])) return bad_arg("asac_xty_multi: one output twice");
        Args& a = A.job[j];
        a.x = x[j], a.xs = x_row_strides[j], a.M = M[j], a.y = y[j], a.ys = y_row_strides[j], a.N = N[j], a.rows = rows[j], a.part = ws;
        a.blocks = row_blocks(rows[j]);
        a.rows_per_block = ((rows[j] + a.blocks - 1) / a.blocks + kRowsPerIter - 1) / kRowsPerIter * kRowsPerIter;
        A.out[j] = outs[j], A.colsum[j] = colsums ? colsums[j] : nullptr;
        ws += (asac_xty_workspace(rows[j], M[j], N[j]) + 3) / 4 * 4;
        max_blocks = a.blocks > max_blocks ? a.blocks : max_blocks;
        max_mt = (M[j] + 63) / 64 > max_mt ? (M[j] + 63) / 64 : max_mt;
        max_nt = (N[j] + 15) / 16 > max_nt ? (N[j] + 15) / 16 : max_nt;
        max_total = M[j] * N[j] + M[j] > max_total ? M[j] * N[j] + M[j] : max_total;
    }
    const dim3 grid((unsigned)max_blocks, (unsigned)max_mt, (unsigned)n_jobs);
    hipStream_t s = as_stream(stream);
    if (max_nt <= 1) ASAC_LAUNCH(k_xty_multi<1>, grid, dim3(kThreads), 0, s, A);
    else if (max_nt <= 2) ASAC_LAUNCH(k_xty_multi<2>, grid, dim3(kThreads), 0, s, A);
    else if (max_nt <= 4) ASAC_LAUNCH(k_xty_multi<4>, grid, dim3(kThreads), 0, s, A);
    else ASAC_LAUNCH(k_xty_multi<8>, grid, dim3(kThreads), 0, s, A);
    hipLaunchKernelGGL(k_xty_reduce_multi, dim3((unsigned)((max_total + 63) / 64), (unsigned)n_jobs), dim3(64 * kRedSlices), 0, s, A);
    return finish_launch("asac_xty_multi");
}

}  // extern "C"
