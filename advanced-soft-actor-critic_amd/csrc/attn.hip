// Attention core for short windows (gfx950): scores, mask, softmax and the weighted sum of values of
// `MultiheadAttention.forward` (reference algorithm/nn_models/layers/seq_layers.py:239-333, the episode attention
// of `get_l_states`, sac_base.py:1117-1146) for key / query windows of at most 32 positions and heads of at most
// 16 channels — the episodic representation attends over a handful of window positions, so a (batch, query) row is
// a few hundred multiply-adds: one lane per row, no matrix tiles.  One launch per pass instead of ~10 / ~12
// elementwise, batched-GEMM and softmax launches.  C ABI in include/asac_hip.h.
//   forward   s_j = (q_i / sqrt(D)) . k_j;  blocked keys -> -inf unless EVERY key of the row is blocked (such a
//             "dead" row attends unmasked and is zeroed by the caller, as in the reference);  w = softmax(s);
//             out_i = sum_j w_j v_j;  keep_i = row not dead;  the returned weights are w * keep
//   backward  gw_j = g_w_j + g_out . v_j;  gs = w * (gw - sum_l w_l gw_l);  g_q = (sum_j gs_j k_j) / sqrt(D);
//             g_k_j = sum_i gs_ij q_i / sqrt(D);  g_v_j = sum_i w_ij g_out_i  (a workgroup owns whole batch entries,
//             so the sums over the queries of an entry stay inside it: LDS, fixed order)
#include "asac_common.h"
#include "asac_gelu.h"

#include <cmath>

namespace asac {

constexpr int kAttnThreads = 64;      // one wave: a launch has few thousand rows, spread them over the CUs
constexpr int kAttnMaxL = ASAC_ATTN_MAX_LEN;    // 32
constexpr int kAttnMaxD = ASAC_ATTN_MAX_DIM;    // 16
constexpr int kAttnPitch = kAttnMaxL + 1;

struct AttnArgs {
    const float* q; const float* k; const float* v;       // [B][Lq][D], [B][Lk][D], [B][Lk][D]
    const uint8_t* mask;                                    // element (b, i, j) at b*sb + i*si + j*sj, or NULL
    int64_t mask_sb, mask_si, mask_sj;
    int32_t B, Lq, Lk, D;
    float* out; float* w; float* keep;                      // [B][Lq][D], [B][Lq][Lk], [B][Lq]
    const float* g_out; const float* g_w;                   // backward: [B][Lq][D], [B][Lq][Lk] or NULL
    float* g_q; float* g_k; float* g_v;
};

// A workgroup (one wave) owns EPB whole batch entries: their keys / values (and, backward, queries and output
// gradients) are staged into LDS with coalesced loads — a lane's inner loops then run on LDS latency instead of one
// dependent global round trip per key.
constexpr int kAttnStage = 2048;                // floats per staged array: EPB * L * D <= this

__host__ __device__ inline int attn_entries_per_block(int Lq, int Lk, int D, int stage = kAttnStage) {
    const int P = Lq > Lk ? Lq : Lk;
    int e = kAttnThreads / P;
    const int cap = stage / (P * D);
    e = e < cap ? e : cap;
    return e < 1 ? 1 : e;
}

__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, int count) {
    for (int i = threadIdx.x; i < count; i += kAttnThreads) dst[i] = src[i];
}

__global__ __launch_bounds__(kAttnThreads) void k_attn_fwd(const AttnArgs a, int EPB) {
    __shared__ float s_l[kAttnThreads * kAttnPitch];        // this lane's scores (odd pitch: conflict-free)
    __shared__ float kL[kAttnStage], vL[kAttnStage];
    const int b0 = blockIdx.x * EPB;
    const int nb = min(EPB, a.B - b0);
    stage_rows(kL, a.k + (int64_t)b0 * a.Lk * a.D, nb * a.Lk * a.D);
    stage_rows(vL, a.v + (int64_t)b0 * a.Lk * a.D, nb * a.Lk * a.D);
    const int bl = threadIdx.x / a.Lq, i = threadIdx.x - bl * a.Lq;
    const bool on = bl < nb;
    const int b = b0 + (on ? bl : 0);
    const int64_t row = (int64_t)b * a.Lq + i;
    float* s = s_l + threadIdx.x * kAttnPitch;
    float qv[kAttnMaxD];
#pragma unroll
    for (int d = 0; d < kAttnMaxD; ++d) qv[d] = (on && d < a.D) ? a.q[row * a.D + d] / sqrtf((float)a.D) : 0.f;
    // the row's mask bits (up to 32 keys) in one register
    unsigned blocked = 0u;
    if (on && a.mask) {
        const uint8_t* mrow = a.mask + (int64_t)b * a.mask_sb + (int64_t)i * a.mask_si;
#pragma unroll 8
        for (int j = 0; j < a.Lk; ++j) blocked |= (mrow[(int64_t)j * a.mask_sj] ? 1u : 0u) << j;
    }
    const unsigned all = a.Lk >= 32 ? 0xffffffffu : ((1u << a.Lk) - 1u);
    const bool dead = a.mask && blocked == all;
    if (dead) blocked = 0u;                                 // a dead row attends unmasked (reference: its bias row is 0)
    __syncthreads();
    if (!on) return;
    const float* kb = kL + bl * a.Lk * a.D;
    const float* vb = vL + bl * a.Lk * a.D;
    float m = -INFINITY;
    for (int j = 0; j < a.Lk; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < kAttnMaxD; ++d)
            if (d < a.D) acc = fmaf(qv[d], kb[j * a.D + d], acc);
        acc = ((blocked >> j) & 1u) ? -INFINITY : acc;
        s[j] = acc;
        m = fmaxf(m, acc);
    }
    float sum = 0.f;
    for (int j = 0; j < a.Lk; ++j) {
        const float e = expf(s[j] - m);
        s[j] = e;
        sum += e;
    }
    const float rs = 1.f / sum;
    float ov[kAttnMaxD];
#pragma unroll
    for (int d = 0; d < kAttnMaxD; ++d) ov[d] = 0.f;
    const float kp = dead ? 0.f : 1.f;
    for (int j = 0; j < a.Lk; ++j) {
        const float w = s[j] * rs;
        a.w[row * a.Lk + j] = w * kp;
#pragma unroll
        for (int d = 0; d < kAttnMaxD; ++d)
            if (d < a.D) ov[d] = fmaf(w, vb[j * a.D + d], ov[d]);
    }
#pragma unroll
    for (int d = 0; d < kAttnMaxD; ++d)
        if (d < a.D) a.out[row * a.D + d] = ov[d];
    a.keep[row] = kp;
}

// lane (entry bl, r): r = query row in phase 1, key row in phase 2; P = lanes per entry (max(Lq, Lk))
__global__ __launch_bounds__(kAttnThreads) void k_attn_bwd(const AttnArgs a, int P, int EPB) {
    __shared__ float gs_l[kAttnThreads * kAttnPitch];       // gs[bl][i][j] at (bl * P + i) * pitch + j
    __shared__ float w_l[kAttnThreads * kAttnPitch];        // saved weights, same layout
    __shared__ float kL[kAttnStage], vL[kAttnStage], qL[kAttnStage], goL[kAttnStage];
    const int b0 = blockIdx.x * EPB;
    const int nb = min(EPB, a.B - b0);
    stage_rows(kL, a.k + (int64_t)b0 * a.Lk * a.D, nb * a.Lk * a.D);
    stage_rows(vL, a.v + (int64_t)b0 * a.Lk * a.D, nb * a.Lk * a.D);
    stage_rows(qL, a.q + (int64_t)b0 * a.Lq * a.D, nb * a.Lq * a.D);
    stage_rows(goL, a.g_out + (int64_t)b0 * a.Lq * a.D, nb * a.Lq * a.D);
    for (int f = threadIdx.x; f < nb * a.Lq * a.Lk; f += kAttnThreads) {
        const int e = f / (a.Lq * a.Lk), rem = f - e * a.Lq * a.Lk, i = rem / a.Lk, j = rem - i * a.Lk;
        w_l[(e * P + i) * kAttnPitch + j] = a.w[(int64_t)b0 * a.Lq * a.Lk + f];
    }
    __syncthreads();
    const int bl = threadIdx.x / P, r = threadIdx.x - bl * P;
    const bool on = bl < nb;
    const int b = b0 + (on ? bl : 0);
    const float* kb = kL + bl * a.Lk * a.D;
    const float* vb = vL + bl * a.Lk * a.D;
    const float rsd = 1.f / sqrtf((float)a.D);
    // phase 1: query row i = r
    if (on && r < a.Lq) {
        const int64_t row = (int64_t)b * a.Lq + r;
        float* gs = gs_l + (bl * P + r) * kAttnPitch;
        const float* w = w_l + (bl * P + r) * kAttnPitch;   // zero on dead rows: their gradients vanish
        float go[kAttnMaxD];
#pragma unroll
        for (int d = 0; d < kAttnMaxD; ++d) go[d] = d < a.D ? goL[(bl * a.Lq + r) * a.D + d] : 0.f;
        float dot = 0.f;
        for (int j = 0; j < a.Lk; ++j) {
            float gw = a.g_w ? a.g_w[row * a.Lk + j] : 0.f;
#pragma unroll
            for (int d = 0; d < kAttnMaxD; ++d)
                if (d < a.D) gw = fmaf(go[d], vb[j * a.D + d], gw);
            gs[j] = gw;
            dot = fmaf(w[j], gw, dot);
        }
        float gq[kAttnMaxD];
#pragma unroll
        for (int d = 0; d < kAttnMaxD; ++d) gq[d] = 0.f;
        for (int j = 0; j < a.Lk; ++j) {
            const float g = w[j] * (gs[j] - dot);
            gs[j] = g;
#pragma unroll
            for (int d = 0; d < kAttnMaxD; ++d)
                if (d < a.D) gq[d] = fmaf(g, kb[j * a.D + d], gq[d]);
        }
#pragma unroll
        for (int d = 0; d < kAttnMaxD; ++d)
            if (d < a.D) a.g_q[row * a.D + d] = gq[d] / sqrtf((float)a.D);
    }
    __syncthreads();
    // phase 2: key row j = r: sums over the entry's queries (fixed order)
    if (on && r < a.Lk) {
        float gk[kAttnMaxD], gv[kAttnMaxD];
#pragma unroll
        for (int d = 0; d < kAttnMaxD; ++d) gk[d] = gv[d] = 0.f;
        for (int i = 0; i < a.Lq; ++i) {
            const float g = gs_l[(bl * P + i) * kAttnPitch + r];
            const float w = w_l[(bl * P + i) * kAttnPitch + r];
#pragma unroll
            for (int d = 0; d < kAttnMaxD; ++d)
                if (d < a.D) {
                    gk[d] = fmaf(g, qL[(bl * a.Lq + i) * a.D + d] / sqrtf((float)a.D), gk[d]);
                    gv[d] = fmaf(w, goL[(bl * a.Lq + i) * a.D + d], gv[d]);
                }
        }
        (void)rsd;
        const int64_t kr = ((int64_t)b * a.Lk + r) * a.D;
#pragma unroll
        for (int d = 0; d < kAttnMaxD; ++d)
            if (d < a.D) {
                a.g_k[kr + d] = gk[d];
                a.g_v[kr + d] = gv[d];
            }
    }
}


// ------------------------------------------------------------------------------------------------
// The same with the three input projections on chip:  q = Wq x_q + bq,  k = Wk x_k + bk,  v = Wv x_k + bv
// (one head, E = embed dim <= 16; the reference's q_proj / k_proj / v_proj are plain Linear layers when
// qkv_dense_depth = 0, seq_layers.py:268-276).  x_q / x_k are read with (batch, row) strides so the query slice of
// the key window needs no copy.  The backward recomputes the projections, returns the gradients of x_q / x_k and
// leaves per-workgroup partial parameter gradients [Wq | bq | Wk | bk | Wv | bv] for a fixed-order reduction.
// ------------------------------------------------------------------------------------------------
struct AttnProjArgs {
    const float* xq; const float* xk;
    int64_t xq_sb, xq_sr, xk_sb, xk_sr;                     // strides in floats: batch entry, row
    const float* wq; const float* bq; const float* wk; const float* bk; const float* wv; const float* bv;
    const float* wo; const float* bo;                       // optional output ResBlock y = GELU(Wo o + bo) + o (NULL: none)
    float* attn_out;                                        // with it: the attention output o [B][Lq][E] (saved / read back)
    const uint8_t* row_zero;                                // with it, optional: [B][Lq] (strides rz_sb, rz_si) rows whose
    int64_t rz_sb, rz_si;                                   //   output is zeroed (padded positions)
    const uint8_t* mask;
    int64_t mask_sb, mask_si, mask_sj;
    int32_t B, Lq, Lk, E;
    float* out; float* w; float* keep;
    const float* g_out; const float* g_w;
    int64_t go_sb, go_sr;                                   // projection backward: strides of g_out (floats: entry, row)
    float* g_xq; float* g_xk;                               // dense [B][Lq][E], [B][Lk][E]
    float* partial;                                         // [blocks][(3 or 4) * (E*E + E)]
};

constexpr int kProjStage = 768;                 // floats per staged array of the projection kernels (eleven arrays)

// The projection kernels are compiled for a padded width EM (8 or 16 >= E): weights, rows and gradients sit in LDS /
// registers zero-padded to EM, so the inner loops carry no width predicates (padding contributes exact zeros).
//   wL: 4 blocks of [EM*EM weights (out, in) | EM biases]
template <int EM>
__device__ __forceinline__ void stage_weights(const AttnProjArgs& a, float* wL) {
    constexpr int blk = EM * EM + EM, total = 4 * blk, IT = (total + kAttnThreads - 1) / kAttnThreads;
    const int E = a.E;
    // every element's load is issued before the first LDS store (a loop of load -> wait -> store was one global round trip
    // per 64 elements: five in a row for E <= 8 at the head of every launch of the block, forward and backward): padding
    // slots load a valid address and are zeroed afterwards
    float v[IT];
    bool on[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = min((int)threadIdx.x + k * kAttnThreads, total - 1);
        const int m = i / blk, r = i - m * blk;
        const float* W = m == 0 ? a.wq : (m == 1 ? a.wk : (m == 2 ? a.wv : a.wo));
        const float* bb = m == 0 ? a.bq : (m == 1 ? a.bk : (m == 2 ? a.bv : a.bo));
        const float* src = a.wq;
        on[k] = false;
        if (W) {
            if (r < EM * EM) {
                const int o = r / EM, c = r - o * EM;
                if (o < E && c < E) src = W + o * E + c, on[k] = true;
            } else if (r - EM * EM < E) {
                src = bb + (r - EM * EM), on[k] = true;
            }
        }
        v[k] = *src;
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) asm volatile("" : "+v"(v[k]));     // (pinned behind the LAST load: no load sinks into a branch)
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = (int)threadIdx.x + k * kAttnThreads;
        if (i < total) wL[i] = on[k] ? v[k] : 0.f;
    }
}

// y = W x + b for one row (block of wL)
template <int EM>
__device__ __forceinline__ void project(const float* w, const float (&x)[EM], float (&y)[EM]) {
#pragma unroll
    for (int o = 0; o < EM; ++o) {
        float acc = w[EM * EM + o];
#pragma unroll
        for (int c = 0; c < EM; ++c) acc = fmaf(w[o * EM + c], x[c], acc);
        y[o] = acc;
    }
}

// y = W^T g (no bias): gradient of a projection's input
template <int EM>
__device__ __forceinline__ void project_t(const float* w, const float (&g)[EM], float (&y)[EM]) {
#pragma unroll
    for (int c = 0; c < EM; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < EM; ++o) acc = fmaf(w[o * EM + c], g[o], acc);
        y[c] = acc;
    }
}

// one element of W^T g at a time (the sixteen-channel backward: with all sixteen outputs of two such products unrolled the
// compiler hoisted 2 x 256 weight reads — 256 VGPRs + 256 AGPRs + 228 bytes of scratch per lane)
template <int EM>
__device__ __forceinline__ float project_t_at(const float* w, const float (&g)[EM], int c) {
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < EM; ++o) acc = fmaf(w[o * EM + c], g[o], acc);
    return acc;
}

template <int EM>
__device__ __forceinline__ void load_row(const float* src, int E, float (&x)[EM]) {
#pragma unroll
    for (int c = 0; c < EM; ++c) x[c] = c < E ? src[c] : 0.f;
}
template <int EM>
__device__ __forceinline__ void put_row(float* dst, const float (&x)[EM]) {
#pragma unroll
    for (int c = 0; c < EM; ++c) dst[c] = x[c];
}
template <int EM>
__device__ __forceinline__ void get_row(const float* src, float (&x)[EM]) {
#pragma unroll
    for (int c = 0; c < EM; ++c) x[c] = src[c];
}

__device__ __forceinline__ unsigned mask_bits(const uint8_t* mask, int64_t sb, int64_t si, int64_t sj, int b, int i,
                                              int Lk) {
    unsigned blocked = 0u;
    const uint8_t* mrow = mask + (int64_t)b * sb + (int64_t)i * si;
#pragma unroll 8
    for (int j = 0; j < Lk; ++j) blocked |= (mrow[(int64_t)j * sj] ? 1u : 0u) << j;
    return blocked;
}

template <int EM>
__global__ __launch_bounds__(kAttnThreads) void k_attn_proj_fwd(const AttnProjArgs a, int P, int EPB) {
    __shared__ float s_l[kAttnThreads * kAttnPitch];
    __shared__ float kL[kProjStage], vL[kProjStage], wL[4 * (EM * EM + EM)];
    constexpr int blk = EM * EM + EM;
    const int E = a.E;
    const int b0 = blockIdx.x * EPB, nb = min(EPB, a.B - b0);
    const int bl = threadIdx.x / P, r = threadIdx.x - bl * P;
    const bool on = bl < nb;
    const int b = b0 + (on ? bl : 0);
    const bool row_on = on && r < a.Lq;
    // inputs travel while the weights are being staged (requested first: the staging waits for its own loads)
    float xk[EM], xq[EM];
    load_row<EM>(a.xk + (int64_t)b * a.xk_sb + (int64_t)min(r, a.Lk - 1) * a.xk_sr, E, xk);
    load_row<EM>(a.xq + (int64_t)b * a.xq_sb + (int64_t)min(r, a.Lq - 1) * a.xq_sr, E, xq);
    stage_weights<EM>(a, wL);
    unsigned blocked = 0u;
    if (row_on && a.mask) blocked = mask_bits(a.mask, a.mask_sb, a.mask_si, a.mask_sj, b, r, a.Lk);
    __syncthreads();
    if (on && r < a.Lk) {          // key / value rows of this entry: lane (bl, j)
        float y[EM];
        project<EM>(wL + blk, xk, y);
        put_row<EM>(kL + (bl * a.Lk + r) * EM, y);
        project<EM>(wL + 2 * blk, xk, y);
        put_row<EM>(vL + (bl * a.Lk + r) * EM, y);
    }
    float qv[EM];
    project<EM>(wL, xq, qv);
    const float rsd = sqrtf((float)E);
#pragma unroll
    for (int c = 0; c < EM; ++c) qv[c] = qv[c] / rsd;
    __syncthreads();
    if (!row_on) return;
    const int64_t row = (int64_t)b * a.Lq + r;
    const unsigned all = a.Lk >= 32 ? 0xffffffffu : ((1u << a.Lk) - 1u);
    const bool dead = a.mask && blocked == all;
    if (dead) blocked = 0u;
    float* s = s_l + threadIdx.x * kAttnPitch;
    const float* kb = kL + bl * a.Lk * EM;
    const float* vb = vL + bl * a.Lk * EM;
    float m = -INFINITY;
    for (int j = 0; j < a.Lk; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < EM; ++d) acc = fmaf(qv[d], kb[j * EM + d], acc);
        acc = ((blocked >> j) & 1u) ? -INFINITY : acc;
        s[j] = acc;
        m = fmaxf(m, acc);
    }
    float sum = 0.f;
    for (int j = 0; j < a.Lk; ++j) {
        const float e = expf(s[j] - m);
        s[j] = e;
        sum += e;
    }
    const float rs = 1.f / sum, kp = dead ? 0.f : 1.f;
    float ov[EM];
#pragma unroll
    for (int d = 0; d < EM; ++d) ov[d] = 0.f;
    for (int j = 0; j < a.Lk; ++j) {
        const float w = s[j] * rs;
        a.w[row * a.Lk + j] = w * kp;
#pragma unroll
        for (int d = 0; d < EM; ++d) ov[d] = fmaf(w, vb[j * EM + d], ov[d]);
    }
    if (a.wo) {                    // output ResBlock on the row, then the dead-row rule: y = (GELU(Wo o + bo) + o) * keep
        float z[EM];
        project<EM>(wL + 3 * blk, ov, z);
        const float ko = (a.row_zero && a.row_zero[(int64_t)b * a.rz_sb + (int64_t)r * a.rz_si]) ? 0.f : kp;
#pragma unroll
        for (int d = 0; d < EM; ++d)
            if (d < E) {
                a.attn_out[row * E + d] = ov[d];
                a.out[row * E + d] = (gelu_f(z[d]) + ov[d]) * ko;
            }
    } else {
#pragma unroll
        for (int d = 0; d < EM; ++d)
            if (d < E) a.out[row * E + d] = ov[d];
    }
    a.keep[row] = kp;
}

// phase clocks of workgroup 0 (tools/debug/attn_bwd_phases.py); compiled out of the library
#ifdef ASAC_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[8];
#define ATTN_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_attn_stamps[k] = __builtin_readcyclecounter(); } while (0)
extern "C" int asac_debug_attn_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps));
}
#else
#define ATTN_STAMP(k)
#endif

template <int EM>
__global__ __launch_bounds__(kAttnThreads) void k_attn_proj_bwd(const AttnProjArgs a, int P, int EPB) {
    __shared__ float gs_l[kAttnThreads * kAttnPitch], w_l[kAttnThreads * kAttnPitch];
    __shared__ float kL[kProjStage], vL[kProjStage], qL[kProjStage], xqL[kProjStage], xkL[kProjStage];
    __shared__ float gqL[kProjStage], gkL[kProjStage], gvL[kProjStage], wL[4 * (EM * EM + EM)];
    __shared__ float goL[kProjStage], gzL[kProjStage], ovL[kProjStage];   // d/d attention output; output-block terms
    __shared__ float gxqL[kProjStage];      // tail form: the query rows' input gradients on their way into the key rows'
    constexpr int blk = EM * EM + EM;
    const int E = a.E;
    const bool tail = a.g_xq == nullptr;     // the queries ARE the last Lq key rows: one gradient serves both
    const int b0 = blockIdx.x * EPB, nb = min(EPB, a.B - b0);
    ATTN_STAMP(0);
    // this lane's rows first: their loads travel under the staging of the weights and of the attention weights (issued
    // behind it each of the three waited for a global round trip of its own)
    const int bl = threadIdx.x / P, r = threadIdx.x - bl * P;
    const bool on = bl < nb;
    const int b = b0 + (on ? bl : 0);
    const bool q_on = on && r < a.Lq, k_on = on && r < a.Lk;
    const int64_t row = (int64_t)b * a.Lq + min(r, a.Lq - 1);
    float xk[EM], xq[EM], go[EM], o[EM];
    load_row<EM>(a.xk + (int64_t)b * a.xk_sb + (int64_t)min(r, a.Lk - 1) * a.xk_sr, E, xk);
    load_row<EM>(a.xq + (int64_t)b * a.xq_sb + (int64_t)min(r, a.Lq - 1) * a.xq_sr, E, xq);
    load_row<EM>(a.g_out + (int64_t)b * a.go_sb + (int64_t)min(r, a.Lq - 1) * a.go_sr, E, go);
    stage_weights<EM>(a, wL);
    // (the weights' gradient is staged with the weights: read inside phase 1's loop over the keys it was one global round
    // trip per key in front of every row's chain)
    for (int f = threadIdx.x; f < nb * a.Lq * a.Lk; f += kAttnThreads) {
        const int e = f / (a.Lq * a.Lk), rem = f - e * a.Lq * a.Lk, i = rem / a.Lk, j = rem - i * a.Lk;
        const float wv = a.w[(int64_t)b0 * a.Lq * a.Lk + f];
        const float gv = a.g_w ? a.g_w[(int64_t)b0 * a.Lq * a.Lk + f] : 0.f;
        w_l[(e * P + i) * kAttnPitch + j] = wv;
        gs_l[(e * P + i) * kAttnPitch + j] = gv;
    }
    float kp = 1.f;
    if (a.wo) {
        load_row<EM>(a.attn_out + row * E, E, o);
        kp = a.keep[row];
        if (a.row_zero && a.row_zero[(int64_t)b * a.rz_sb + (int64_t)min(r, a.Lq - 1) * a.rz_si]) kp = 0.f;
    }
    __syncthreads();
    ATTN_STAMP(1);
    // recompute the projections of this entry (lane (bl, j): k, v; lane (bl, i): q), keep the inputs for the
    // parameter gradients
    if (k_on) {
        float y[EM];
        project<EM>(wL + blk, xk, y);
        put_row<EM>(kL + (bl * a.Lk + r) * EM, y);
        put_row<EM>(xkL + (bl * a.Lk + r) * EM, xk);
        project<EM>(wL + 2 * blk, xk, y);
        put_row<EM>(vL + (bl * a.Lk + r) * EM, y);
    }
    const float rsd = sqrtf((float)E);
    if (q_on) {
        float y[EM];
        project<EM>(wL, xq, y);
#pragma unroll
        for (int c = 0; c < EM; ++c) y[c] = y[c] / rsd;
        put_row<EM>(qL + (bl * a.Lq + r) * EM, y);
        put_row<EM>(xqL + (bl * a.Lq + r) * EM, xq);
        if constexpr (EM > 8) {
            // (the same products and sums, one output at a time through the row's LDS stages: see project_t_at)
            float* gor = goL + (bl * a.Lq + r) * EM;
            if (a.wo) {
                float* gzr = gzL + (bl * a.Lq + r) * EM;
                const float* wo = wL + 3 * blk;
#pragma unroll
                for (int d = 0; d < EM; ++d) go[d] *= kp;
                put_row<EM>(gor, go);
                put_row<EM>(ovL + (bl * a.Lq + r) * EM, o);
#pragma unroll 1
                for (int d = 0; d < EM; ++d) {
                    float z = wo[EM * EM + d];
#pragma unroll
                    for (int c = 0; c < EM; ++c) z = fmaf(wo[d * EM + c], o[c], z);
                    gzr[d] = gor[d] * gelu_grad(z);
                }
#pragma unroll 1
                for (int c = 0; c < EM; ++c) {
                    float acc = 0.f;
#pragma unroll
                    for (int d = 0; d < EM; ++d) acc = fmaf(wo[d * EM + c], gzr[d], acc);
                    gor[c] += acc;
                }
#pragma unroll
                for (int d = 0; d < EM; ++d) go[d] = gor[d];
            } else {
                put_row<EM>(gor, go);
            }
        } else {
        if (a.wo) {                // back through y = (GELU(z) + o) * keep, z = Wo o + bo
            float z[EM], gz[EM], back[EM];
            project<EM>(wL + 3 * blk, o, z);
#pragma unroll
            for (int d = 0; d < EM; ++d) {
                go[d] *= kp;
                gz[d] = go[d] * gelu_grad(z[d]);
            }
            put_row<EM>(gzL + (bl * a.Lq + r) * EM, gz);
            put_row<EM>(ovL + (bl * a.Lq + r) * EM, o);
            project_t<EM>(wL + 3 * blk, gz, back);
#pragma unroll
            for (int d = 0; d < EM; ++d) go[d] += back[d];
        }
        put_row<EM>(goL + (bl * a.Lq + r) * EM, go);
        }
    }
    __syncthreads();
    ATTN_STAMP(2);
    const float* kb = kL + bl * a.Lk * EM;
    const float* vb = vL + bl * a.Lk * EM;
    // phase 1: query row i = r -> gs (LDS), g_q, gradient of x_q
    if (q_on) {
        float* gs = gs_l + (bl * P + r) * kAttnPitch;
        const float* w = w_l + (bl * P + r) * kAttnPitch;
        float dot = 0.f;
        for (int j = 0; j < a.Lk; ++j) {
            float gw = gs[j];                                      // (d loss / d weight, staged above; 0: none)
#pragma unroll
            for (int d = 0; d < EM; ++d) gw = fmaf(go[d], vb[j * EM + d], gw);
            gs[j] = gw;
            dot = fmaf(w[j], gw, dot);
        }
        float gq[EM], gx[EM];
#pragma unroll
        for (int d = 0; d < EM; ++d) gq[d] = 0.f;
        for (int j = 0; j < a.Lk; ++j) {
            const float g = w[j] * (gs[j] - dot);
            gs[j] = g;
#pragma unroll
            for (int d = 0; d < EM; ++d) gq[d] = fmaf(g, kb[j * EM + d], gq[d]);
        }
#pragma unroll
        for (int d = 0; d < EM; ++d) gq[d] = gq[d] / rsd;      // gradient of the unscaled projection output
        put_row<EM>(gqL + (bl * a.Lq + r) * EM, gq);
        if constexpr (EM > 8) {                                   // (same sums, one output at a time: see project_t_at)
#pragma unroll 1
            for (int c = 0; c < EM; ++c) {
                const float v = project_t_at<EM>(wL, gq, c);
                if (tail) gxqL[(bl * a.Lq + r) * EM + c] = v;
                else if (c < E) a.g_xq[row * E + c] = v;
            }
        } else {
        project_t<EM>(wL, gq, gx);                                // gradient of x_q: Wq^T g_q
        if (tail) {
            put_row<EM>(gxqL + (bl * a.Lq + r) * EM, gx);
        } else {
#pragma unroll
            for (int c = 0; c < EM; ++c)
                if (c < E) a.g_xq[row * E + c] = gx[c];
        }
        }
    }
    __syncthreads();
    ATTN_STAMP(3);
    // phase 2: key row j = r -> g_k, g_v (sums over the entry's queries), gradient of x_k
    if (k_on) {
        float gk[EM], gv[EM], gx[EM], gx2[EM];
#pragma unroll
        for (int d = 0; d < EM; ++d) gk[d] = gv[d] = 0.f;
        for (int i = 0; i < a.Lq; ++i) {
            const float g = gs_l[(bl * P + i) * kAttnPitch + r];
            const float w = w_l[(bl * P + i) * kAttnPitch + r];
#pragma unroll
            for (int d = 0; d < EM; ++d) {
                gk[d] = fmaf(g, qL[(bl * a.Lq + i) * EM + d], gk[d]);
                gv[d] = fmaf(w, goL[(bl * a.Lq + i) * EM + d], gv[d]);
            }
        }
        put_row<EM>(gkL + (bl * a.Lk + r) * EM, gk);
        put_row<EM>(gvL + (bl * a.Lk + r) * EM, gv);
        const int64_t kr = ((int64_t)b * a.Lk + r) * E;
        const int qi = r - (a.Lk - a.Lq);                         // tail form: this key row is query row qi
        if constexpr (EM > 8) {
#pragma unroll 1
            for (int c = 0; c < E; ++c) {
                float v = project_t_at<EM>(wL + blk, gk, c) + project_t_at<EM>(wL + 2 * blk, gv, c);
                if (tail && qi >= 0) v += gxqL[(bl * a.Lq + qi) * EM + c];
                a.g_xk[kr + c] = v;
            }
        } else {
        project_t<EM>(wL + blk, gk, gx);
        project_t<EM>(wL + 2 * blk, gv, gx2);
#pragma unroll
        for (int c = 0; c < EM; ++c)
            if (c < E) {
                float v = gx[c] + gx2[c];
                if (tail && qi >= 0) v += gxqL[(bl * a.Lq + qi) * EM + c];     // (as `g_xk[:, -Lq:] += g_xq`)
                a.g_xk[kr + c] = v;
            }
        }
    }
    __syncthreads();
    ATTN_STAMP(4);
    // phase 3: this workgroup's partial parameter gradients, packed for width E.  dW_m[o][c] = sum over the rows t of
    // g_m[t][o] x_m[t][c] on the matrix pipe (16x16x4 f32: A[m = o][k = t], B[k = t][n = c], four rows a step; E <= 8: two
    // matrices share a tile — o / c of the second in rows / columns 8..15, the mixed blocks are not stored): as a loop of
    // one product per (o, c) and row with its two LDS reads this phase was 46 % of the launch (17.8 of 38.6 k clocks at
    // 1 024 x 9 x 8, tools/debug/attn_bwd_phases.py).  Biases: plain sums in row order.
    const int nmat = a.wo ? 4 : 3, pblk = E * E + E;
    float* part = a.partial + (int64_t)blockIdx.x * nmat * pblk;
    {
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const int lm = threadIdx.x & 15, lq = threadIdx.x >> 4;
        constexpr bool PAIR = EM == 8;
        auto outer = [&](const float* gA, const float* xA, const float* gB, const float* xB, int rows) -> f32x4 {
            const bool second = PAIR && lm >= 8;
            const float* gp = second ? gB + (lm - 8) : gA + lm;
            const float* xp = second ? xB + (lm - 8) : xA + lm;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            constexpr int SB = 4;                      // steps whose operands are read before the first of their products
            for (int t0 = 0; t0 < rows; t0 += 4 * SB) {
                float av[SB], bv[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int t = t0 + 4 * u + lq, tc = min(t, rows - 1);
                    av[u] = gp[tc * EM];
                    bv[u] = xp[tc * EM];
                    av[u] = t < rows ? av[u] : 0.f;    // (rows beyond the workgroup's: no contribution)
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            }
            return acc;
        };
        auto store = [&](const f32x4& acc, int mA, int mB) {       // D[4 lq + r][lm] -> the matrices' blocks of the slab
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mrow = 4 * lq + r;
                if (PAIR) {
                    const bool second = lm >= 8;
                    if ((mrow >= 8) != second) continue;            // (a mixed block: g of one matrix with x of the other)
                    const int o = mrow & 7, c = lm & 7, mi = second ? mB : mA;
                    if (mi >= 0 && o < E && c < E) part[mi * pblk + o * E + c] = acc[r];
                } else if (mrow < E && lm < E) {
                    part[mA * pblk + mrow * E + lm] = acc[r];
                }
            }
        };
        if (PAIR) {
            store(outer(gqL, xqL, gzL, ovL, nb * a.Lq), 0, a.wo ? 3 : -1);
            store(outer(gkL, xkL, gvL, xkL, nb * a.Lk), 1, 2);
        } else {
            store(outer(gqL, xqL, gqL, xqL, nb * a.Lq), 0, -1);
            store(outer(gkL, xkL, gkL, xkL, nb * a.Lk), 1, -1);
            store(outer(gvL, xkL, gvL, xkL, nb * a.Lk), 2, -1);
            if (a.wo) store(outer(gzL, ovL, gzL, ovL, nb * a.Lq), 3, -1);
        }
    }
    for (int idx = threadIdx.x; idx < nmat * EM; idx += kAttnThreads) {
        const int m = idx / EM, oo = idx - m * EM;
        if (oo >= E) continue;
        const float* g = m == 0 ? gqL : (m == 1 ? gkL : (m == 2 ? gvL : gzL));
        const int rows = nb * ((m == 0 || m == 3) ? a.Lq : a.Lk);
        constexpr int UB = 6;
        float acc = 0.f;
        for (int t0 = 0; t0 < rows; t0 += UB) {
            float gv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) gv[u] = g[min(t0 + u, rows - 1) * EM + oo];
#pragma unroll
            for (int u = 0; u < UB; ++u) acc = t0 + u < rows ? acc + gv[u] : acc;
        }
        part[m * pblk + E * E + oo] = acc;
    }
    ATTN_STAMP(5);
}

// grad[i] (+)= sum over workgroups of partial[block][i]: 64 parameters per workgroup, 16 slices of blocks
__global__ __launch_bounds__(64 * 16) void k_attn_sum_partials(const float* __restrict__ partial, int blocks, int n,
                                                               float* __restrict__ out, int accumulate) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int per = (blocks + 15) / 16, lo = sl * per, hi = min(lo + per, blocks);
    float s = 0.f;
    if (i < n) {
        int bk = lo;
        for (; bk + 8 <= hi; bk += 8) {
            float v[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) v[w] = partial[(int64_t)(bk + w) * n + i];
#pragma unroll
            for (int w = 0; w < 8; ++w) s += v[w];
        }
        for (; bk < hi; ++bk) s += partial[(int64_t)bk * n + i];
    }
    part[sl][lane] = s;
    __syncthreads();
    if (sl != 0 || i >= n) return;
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += part[w][lane];
    out[i] = accumulate ? out[i] + s : s;
}

// the backward's last phase sums outer products over the workgroup's rows: few entries per workgroup keep that loop
// short and the launch wide (the partial slabs are summed by a second kernel either way)
static int attn_proj_bwd_entries(int Lq, int Lk, int E) {
    const int e = attn_entries_per_block(Lq, Lk, E <= 8 ? 8 : 16, kProjStage);
    return e < 2 ? e : 2;
}

static bool attn_ok(int B, int Lq, int Lk, int D) {
    return B > 0 && Lq >= 1 && Lq <= kAttnMaxL && Lk >= 1 && Lk <= kAttnMaxL && D >= 1 && D <= kAttnMaxD;
}

}  // namespace asac

using namespace asac;

extern "C" {

int asac_attention_supported(int Lq, int Lk, int D) { return attn_ok(1, Lq, Lk, D) ? 1 : 0; }

int asac_attention_forward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                           int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int D, float* out,
                           float* weights, float* keep, void* stream) {
    if (!attn_ok(B, Lq, Lk, D) || !q || !k || !v || !out || !weights || !keep) return bad_arg("asac_attention_forward");
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v;
    a.mask = mask; a.mask_sb = mask_stride_b; a.mask_si = mask_stride_q; a.mask_sj = mask_stride_k;
    a.B = B; a.Lq = Lq; a.Lk = Lk; a.D = D;
    a.out = out; a.w = weights; a.keep = keep;
    const int EPB = attn_entries_per_block(Lq, Lk, D);
    ASAC_LAUNCH(k_attn_fwd, dim3((unsigned)((B + EPB - 1) / EPB)), dim3(kAttnThreads), 0, as_stream(stream), a, EPB);
    return finish_launch("asac_attention_forward");
}

int asac_attention_backward(const float* q, const float* k, const float* v, const float* weights, const float* grad_out,
                            const float* grad_weights, int B, int Lq, int Lk, int D, float* grad_q, float* grad_k,
                            float* grad_v, void* stream) {
    if (!attn_ok(B, Lq, Lk, D) || !q || !k || !v || !weights || !grad_out || !grad_q || !grad_k || !grad_v)
        return bad_arg("asac_attention_backward");
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v;
    a.B = B; a.Lq = Lq; a.Lk = Lk; a.D = D;
    a.w = const_cast<float*>(weights);
    a.g_out = grad_out; a.g_w = grad_weights;
    a.g_q = grad_q; a.g_k = grad_k; a.g_v = grad_v;
    const int P = Lq > Lk ? Lq : Lk, EPB = attn_entries_per_block(Lq, Lk, D);
    ASAC_LAUNCH(k_attn_bwd, dim3((unsigned)((B + EPB - 1) / EPB)), dim3(kAttnThreads), 0, as_stream(stream), a, P, EPB);
    return finish_launch("asac_attention_backward");
}

int64_t asac_attention_proj_workspace(int B, int Lq, int Lk, int E) {
    if (!attn_ok(B, Lq, Lk, E)) return -1;
    const int EPB = attn_proj_bwd_entries(Lq, Lk, E);
    return (int64_t)((B + EPB - 1) / EPB) * 4 * (E * E + E);
}

static void fill_proj(AttnProjArgs& a, const float* xq, int64_t xq_sb, int64_t xq_sr, const float* xk, int64_t xk_sb,
                      int64_t xk_sr, const float* const* params, int B, int Lq, int Lk, int E) {
    a.xq = xq; a.xq_sb = xq_sb; a.xq_sr = xq_sr;
    a.xk = xk; a.xk_sb = xk_sb; a.xk_sr = xk_sr;
    a.wq = params[0]; a.bq = params[1]; a.wk = params[2]; a.bk = params[3]; a.wv = params[4]; a.bv = params[5];
    a.wo = params[6]; a.bo = params[7];       // both NULL: no output block
    a.B = B; a.Lq = Lq; a.Lk = Lk; a.E = E;
}

int asac_attention_proj_forward(const float* xq, int64_t xq_stride_b, int64_t xq_stride_r, const float* xk,
                                int64_t xk_stride_b, int64_t xk_stride_r, const float* const* params,
                                const uint8_t* mask, int64_t mask_stride_b, int64_t mask_stride_q,
                                int64_t mask_stride_k, int B, int Lq, int Lk, int E, float* out, float* weights,
                                float* keep, float* attn_out, const uint8_t* row_zero, int64_t row_zero_stride_b,
                                int64_t row_zero_stride_q, void* stream) {
    if (!attn_ok(B, Lq, Lk, E) || !xq || !xk || !params || !out || !weights || !keep)
        return bad_arg("asac_attention_proj_forward");
    for (int i = 0; i < 6; ++i)
        if (!params[i]) return bad_arg("asac_attention_proj_forward: params");
    if ((!params[6] != !params[7]) || (params[6] && !attn_out)) return bad_arg("asac_attention_proj_forward: output block");
    AttnProjArgs a{};
    a.attn_out = attn_out;
    a.row_zero = params[6] ? row_zero : nullptr; a.rz_sb = row_zero_stride_b; a.rz_si = row_zero_stride_q;
    fill_proj(a, xq, xq_stride_b, xq_stride_r, xk, xk_stride_b, xk_stride_r, params, B, Lq, Lk, E);
    a.mask = mask; a.mask_sb = mask_stride_b; a.mask_si = mask_stride_q; a.mask_sj = mask_stride_k;
    a.out = out; a.w = weights; a.keep = keep;
    const int P = Lq > Lk ? Lq : Lk, EPB = attn_entries_per_block(Lq, Lk, E <= 8 ? 8 : 16, kProjStage);
    const dim3 grid((unsigned)((B + EPB - 1) / EPB));
    if (E <= 8)
        ASAC_LAUNCH(k_attn_proj_fwd<8>, grid, dim3(kAttnThreads), 0, as_stream(stream), a, P, EPB);
    else
        ASAC_LAUNCH(k_attn_proj_fwd<16>, grid, dim3(kAttnThreads), 0, as_stream(stream), a, P, EPB);
    return finish_launch("asac_attention_proj_forward");
}

int asac_attention_proj_backward(const float* xq, int64_t xq_stride_b, int64_t xq_stride_r, const float* xk,
                                 int64_t xk_stride_b, int64_t xk_stride_r, const float* const* params,
                                 const float* weights, const float* keep, const float* attn_out, const float* grad_out,
                                 int64_t grad_out_stride_b, int64_t grad_out_stride_r,
                                 const float* grad_weights, const uint8_t* row_zero, int64_t row_zero_stride_b,
                                 int64_t row_zero_stride_q, int B, int Lq, int Lk, int E, float* grad_xq, float* grad_xk,
                                 float* grad_params, int accumulate, float* workspace, void* stream) {
    if (!attn_ok(B, Lq, Lk, E) || !xq || !xk || !params || !weights || !grad_out || !grad_xk || !grad_params || !workspace ||
        (!grad_xq && Lq > Lk))
        return bad_arg("asac_attention_proj_backward");
    for (int i = 0; i < 6; ++i)
        if (!params[i]) return bad_arg("asac_attention_proj_backward: params");
    if ((!params[6] != !params[7]) || (params[6] && (!attn_out || !keep)))
        return bad_arg("asac_attention_proj_backward: output block");
    AttnProjArgs a{};
    a.attn_out = const_cast<float*>(attn_out);
    a.keep = const_cast<float*>(keep);
    a.row_zero = params[6] ? row_zero : nullptr; a.rz_sb = row_zero_stride_b; a.rz_si = row_zero_stride_q;
    fill_proj(a, xq, xq_stride_b, xq_stride_r, xk, xk_stride_b, xk_stride_r, params, B, Lq, Lk, E);
    a.w = const_cast<float*>(weights);
    a.g_out = grad_out; a.g_w = grad_weights;
    a.go_sb = grad_out_stride_b; a.go_sr = grad_out_stride_r;
    a.g_xq = grad_xq; a.g_xk = grad_xk;
    a.partial = workspace;
    const int P = Lq > Lk ? Lq : Lk, EPB = attn_proj_bwd_entries(Lq, Lk, E);
    const int blocks = (B + EPB - 1) / EPB, n = (params[6] ? 4 : 3) * (E * E + E);
    hipStream_t s = as_stream(stream);
    if (E <= 8)
        ASAC_LAUNCH(k_attn_proj_bwd<8>, dim3((unsigned)blocks), dim3(kAttnThreads), 0, s, a, P, EPB);
    else
        ASAC_LAUNCH(k_attn_proj_bwd<16>, dim3((unsigned)blocks), dim3(kAttnThreads), 0, s, a, P, EPB);
    // launched once (not under the repeat knob: it may accumulate)
    if (accumulate != ASAC_ATTN_SUM_DEFER)
        hipLaunchKernelGGL(k_attn_sum_partials, dim3((unsigned)((n + 63) / 64)), dim3(64 * 16), 0, s, workspace, blocks, n,
                           grad_params, accumulate);
    return finish_launch("asac_attention_proj_backward");
}

}  // extern "C"
