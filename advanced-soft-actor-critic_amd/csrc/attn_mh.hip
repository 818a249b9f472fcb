// Multi-head attention core for short windows on f32 MFMA: scores, mask, softmax, weighted sum of values and the head-averaged
// weights of `MultiheadAttention.forward` (reference algorithm/nn_models/layers/seq_layers.py:239-333) for the widths the
// reference's environments use (`EpisodeMultiheadAttention(64, …, num_heads 2 … 8)`: envs/square/obstacle/nn_attn.py:16,
// envs/gym/toy_queue/nn_attn.py:28-45) — csrc/attn.hip (one lane per (batch, query) row) covers one head of <= 16 channels.
// The q / k / v / output projections around it are plain Linears: library GEMMs on the host side (seq_layers.py here).
//
// One workgroup (4 waves) owns a batch entry; wave w walks heads w, w + 4, ... (the head average of the weights: per-wave
// register sums, added in wave order through LDS).
// Per head, with windows of <= 32 positions as <= 2 x 2 tiles of 16 and the QUERIES as the N dimension:
//   S[key][query]  = K Q^T       A = a key row's 16 bytes of channels, B = a query row's (x 1/sqrt(d))
//   P = softmax over the keys: the keys of a query live in 4 registers x 4 lanes x <= 2 tiles -> two shuffles
//   O[c][query]    = V^T P       B = the P tile as it stands (keys are its k-slots), A = 4 strided words of V
// Backward: dP = V dO^T (same shape as S), dS = P (dP - sum_keys P dP), dQ = K^T dS, and the two products over the QUERIES
// (dV = P dO, dK = dS Q) with P / dS turned through LDS.  "Dead" rows (every key blocked) attend unmasked and report
// keep = 0, as csrc/attn.hip and the reference do.
#include "asac_common.h"
#include "asac_gelu.h"

#include <cmath>

namespace asac {
namespace amh {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kWaves = 4, kThreads = 64 * kWaves;      // the waves of a workgroup = the heads of one batch entry, dealt round robin
constexpr int kMaxL = 32;
constexpr int kTP = 20;             // LDS pitch of a turned tile's rows

#define AM_MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
    c = AM_MF(a[0], b[0], c);
    c = AM_MF(a[1], b[1], c);
    c = AM_MF(a[2], b[2], c);
    c = AM_MF(a[3], b[3], c);
    return c;
}
// reductions over the lanes that share x (a query's keys are spread over the four lane quarters)
__device__ __forceinline__ float max_over_q(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float sum_over_q(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ int tpos(int f) { return ((f & 3) << 2) | (f >> 2); }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Args {
    const float* q; const float* k; const float* v;      // [B][Lq][E], [B][Lk][E], [B][Lk][E]  (E = heads * d)
    const uint8_t* mask; int64_t m_sb, m_si, m_sj;       // blocked (b, query, key), shared by the heads; or NULL
    int32_t B, Lq, Lk, H, d;
    float* out;            // [B][Lq][E]
    float* w_avg;          // [B][Lq][Lk]  mean over heads of softmax, * keep
    float* keep;           // [B][Lq]
    const uint8_t* row_zero;   // [B][Lq] (batch stride rz_sb) or NULL: the caller's padded positions (rows whose OUTPUT it zeroes)
    int64_t rz_sb;
    float* keep_rows;      // [B][Lq] or NULL: keep * !row_zero — the one factor the caller multiplies the output with
    float* p_heads;        // [B][H][Lq][Lk] per-head softmax (saved for the backward) or NULL
    // backward
    const float* g_out;    // [B][Lq][E]
    const float* g_w;      // [B][Lq][Lk] or NULL
    float* g_q; float* g_k; float* g_v;
    // forward with the projections inside (PEC > 0): q / k / v above are then OUTPUTS (saved for the backward)
    const float* x; int64_t xs_b, xs_t;       // the block's input rows [B][Lk][E]; the queries are its last Lq positions
    const float* pw[3]; const float* pb[3];   // q / k / v projection weights [E][E] and biases [E]
    // ... and the output ResBlock behind the core (ow non-NULL): y = (o + gelu(o ow^T + ob)) * keep_rows, pre = o ow^T + ob
    const float* ow; const float* ob;
    float* y; float* pre;                     // [B][Lq][E]
    // block backward (BWD with PEC > 0): the ResBlock's backward in front of the core's, the projections' input gradient behind
    const float* gy; const float* scale_in;   // gradient of y [B][Lq][E] (strides gy_sb, gy_st); the forward's keep_rows [B][Lq] or NULL
    int64_t gy_sb, gy_st;
    float* gpre; float* gx;                   // gradient of `pre` [B][Lq][E] and of the block's input [B][Lk][E]
};

// 16 bytes of channels [c0, c0 + 4) of a row, zero beyond d (and for rows beyond the window)
__device__ __forceinline__ f32x4 row4(const float* base, bool live, int c0, int d) {
    f32x4 v = zero4();
    if (live && c0 + 3 < d) return *reinterpret_cast<const f32x4*>(base + c0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (live && c0 + r < d) v[r] = base[c0 + r];
    return v;
}

// PEC > 0 (forward, one tile a side, E = 16 PEC): the q / k / v projections of the block's input run in front of the scores — the
// input tile is the B operand of 3 E / 16 weight tiles dealt over the waves (the arithmetic of csrc/rows_proj.hip), the results
// go to LDS for this launch and to q / k / v for the backward: one launch less per block pass (10 of ~25 us).
template <bool BWD, int NT, int PEC = 0>
__global__ void __launch_bounds__(kThreads) k_attn_mh(const Args a) {
    // per wave: P and dS tiles turned for the products over the queries; rows of 16 at a pitch of kTP = 20 floats: the four
    // lane quarters of a store then fall into four different 16-bank groups, rows stay 16-byte aligned for the b128 reads
    __shared__ float s_t[kWaves][2 * NT * NT][16 * kTP];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, qq = l >> 4, x = l & 15;
    const int b = blockIdx.x;
    const int Lq = a.Lq, Lk = a.Lk, H = a.H, d = a.d, E = H * d;
    const int QT = (Lq + 15) >> 4, KT = (Lk + 15) >> 4, CT = (d + 15) >> 4;
    const float scale = 1.f / sqrtf((float)d);
    constexpr int kPP = 16 * (PEC > 0 ? PEC : 1) + 4;      // LDS row pitch of the projected rows (conflict-free 16-byte reads)
    constexpr bool kF = PEC > 0 && !BWD, kB = PEC > 0 && BWD;
    __shared__ __attribute__((aligned(16))) float s_qkv[kF ? 3 : 1][kF ? 16 : 1][kF ? kPP : 4];
    __shared__ __attribute__((aligned(16))) float s_o[kF ? 16 : 1][kF ? kPP : 4];      // the core's output rows (for the ResBlock)
    __shared__ __attribute__((aligned(16))) float s_go[kB ? 16 : 1][kB ? kPP : 4];     // backward: gradient of the core's output
    __shared__ __attribute__((aligned(16))) float s_g3[kB ? 3 : 1][kB ? 16 : 1][kB ? kPP : 4];      // ... and of q / k / v
    const float* qb = a.q + (int64_t)b * Lq * E;
    const float* kb = a.k + (int64_t)b * Lk * E;
    const float* vb = a.v + (int64_t)b * Lk * E;
    int RP = E;                                             // row pitch of q / k / v as the products below read them
    if constexpr (kF) {
        constexpr int EE = 16 * PEC;
        const bool live = x < Lk;
        const float* xp = a.x + (int64_t)b * a.xs_b + (int64_t)min(x, Lk - 1) * a.xs_t + 4 * qq;
        f32x4 xv[PEC];
#pragma unroll
        for (int c = 0; c < PEC; ++c) xv[c] = live ? *reinterpret_cast<const f32x4*>(xp + 16 * c) : zero4();
        const int skip = Lk - Lq;
        for (int item = wv; item < 3 * PEC; item += kWaves) {
            const int j = item / PEC, nt = item - j * PEC;
            const float* wj = j == 0 ? a.pw[0] : (j == 1 ? a.pw[1] : a.pw[2]);
            const float* bj = j == 0 ? a.pb[0] : (j == 1 ? a.pb[1] : a.pb[2]);
            const float* wp = wj + (16 * nt + x) * EE + 4 * qq;
            f32x4 acc = zero4();
#pragma unroll
            for (int c = 0; c < PEC; ++c) acc = mfma4(*reinterpret_cast<const f32x4*>(wp + 16 * c), xv[c], acc);
            acc += *reinterpret_cast<const f32x4*>(bj + 16 * nt + 4 * qq);      // acc[r] = y_j[row x][16 nt + 4 qq + r]
            *reinterpret_cast<f32x4*>(&s_qkv[j][x][16 * nt + 4 * qq]) = acc;
            if (live) {
                if (j == 0) {
                    if (x >= skip) *reinterpret_cast<f32x4*>(const_cast<float*>(qb) + (int64_t)(x - skip) * EE + 16 * nt + 4 * qq) = acc;
                } else {
                    float* dst = const_cast<float*>(j == 1 ? kb : vb);
                    *reinterpret_cast<f32x4*>(dst + (int64_t)x * EE + 16 * nt + 4 * qq) = acc;
                }
            }
        }
        __syncthreads();
        qb = &s_qkv[0][skip][0], kb = &s_qkv[1][0][0], vb = &s_qkv[2][0][0];
        RP = kPP;
    }
    // mask of this entry: lane (qq, x) of tile (km, qn) holds keys 16 km + 4 qq + r of query 16 qn + x
    bool blocked[NT][NT][4];
    bool dead[NT];
#pragma unroll
    for (int qn = 0; qn < NT; ++qn) {
        const int qi = 16 * qn + x;
        bool all_blocked = true;
#pragma unroll
        for (int km = 0; km < NT; ++km)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kj = 16 * km + 4 * qq + r;
                bool bl = false;
                if (a.mask && qi < Lq && kj < Lk) bl = a.mask[(int64_t)b * a.m_sb + (int64_t)qi * a.m_si + (int64_t)kj * a.m_sj] != 0;
                blocked[km][qn][r] = bl;
                if (kj < Lk && !bl) all_blocked = false;
            }
        // every key of the row blocked <=> all four quarters say so
        const int votes = (int)sum_over_q(all_blocked ? 1.f : 0.f);
        dead[qn] = a.mask != nullptr && votes == 4;
    }
    f32x4 wsum[NT][NT];
#pragma unroll
    for (int km = 0; km < NT; ++km)
#pragma unroll
        for (int qn = 0; qn < NT; ++qn) wsum[km][qn] = zero4();
    float* tbuf = &s_t[wv][0][0];
    const float* gob = BWD ? a.g_out + (int64_t)b * Lq * E : nullptr;
    int GP = E;                                             // row pitch of the core's output gradient
    // four strided words of a weight column block: W[n0 + s][k], s < 4 (E x E)
    auto wcol4 = [&](const float* w, int n0, int k) {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = w[(n0 + s) * E + k];
        return v;
    };
    if constexpr (kB) {
        // the output ResBlock's backward over the entry's <= 16 query rows (the arithmetic of k_rows_res_bwd): g = gy * scale,
        // gpre = g * gelu'(pre) -> memory (its product over the rows is the block's weight gradient), g + gpre W -> LDS
        constexpr int EE = 16 * PEC;
        const bool live = x < Lq;
        const int64_t rr = ((int64_t)b * Lq + min(x, Lq - 1)) * EE + 4 * qq;
        const float* gyr = a.gy + (int64_t)b * a.gy_sb + (int64_t)min(x, Lq - 1) * a.gy_st + 4 * qq;      // (a slice is read in place)
        const float sc = a.scale_in ? a.scale_in[(int64_t)b * Lq + min(x, Lq - 1)] : 1.f;
        f32x4 gp[PEC];
#pragma unroll
        for (int c = 0; c < PEC; ++c) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gyr + 16 * c) * sc, z = *reinterpret_cast<const f32x4*>(a.pre + rr + 16 * c);
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[c][r] = live ? g[r] * gelu_grad(z[r]) : 0.f;
            if (live && (c & (kWaves - 1)) == wv) *reinterpret_cast<f32x4*>(a.gpre + rr + 16 * c) = gp[c];
        }
        for (int kt = wv; kt < PEC; kt += kWaves) {
            f32x4 acc = zero4();
#pragma unroll
            for (int c = 0; c < PEC; ++c) acc = mfma4(wcol4(a.ow, 16 * c + 4 * qq, 16 * kt + x), gp[c], acc);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gyr + 16 * kt) * sc + acc;
            *reinterpret_cast<f32x4*>(&s_go[x][16 * kt + 4 * qq]) = live ? g : zero4();
        }
        __syncthreads();
        gob = &s_go[0][0];
        GP = kPP;
    }

    for (int h = wv; h < H; h += kWaves) {
        const int hc = h * d;
        f32x4 P[NT][NT];
        if (!BWD) {
            // ---- scores ---------------------------------------------------------------------------------------------
            f32x4 S[NT][NT];
#pragma unroll
            for (int km = 0; km < NT; ++km)
#pragma unroll
                for (int qn = 0; qn < NT; ++qn) S[km][qn] = zero4();
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 ka[NT], qv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int kj = 16 * t + x, qi = 16 * t + x;
                    ka[t] = row4(kb + (int64_t)min(kj, Lk - 1) * RP + hc, t < KT && kj < Lk, 16 * ct + 4 * qq, d);
                    qv[t] = row4(qb + (int64_t)min(qi, Lq - 1) * RP + hc, t < QT && qi < Lq, 16 * ct + 4 * qq, d) * scale;
                }
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int qn = 0; qn < NT; ++qn)
                        if (km < KT && qn < QT) S[km][qn] = mfma4(ka[km], qv[qn], S[km][qn]);
            }
            // ---- mask + softmax over the keys of each query ---------------------------------------------------------
#pragma unroll
            for (int qn = 0; qn < NT; ++qn) {
                float mx = -INFINITY;
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kj = 16 * km + 4 * qq + r;
                        const bool off = kj >= Lk || (blocked[km][qn][r] && !dead[qn]);
                        S[km][qn][r] = off ? -INFINITY : S[km][qn][r];
                        mx = fmaxf(mx, S[km][qn][r]);
                    }
                mx = max_over_q(mx);
                float sum = 0.f;
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = S[km][qn][r] == -INFINITY ? 0.f : expf(S[km][qn][r] - mx);
                        P[km][qn][r] = e;
                        sum += e;
                    }
                sum = sum_over_q(sum);
                const float inv = 1.f / sum;
#pragma unroll
                for (int km = 0; km < NT; ++km) {
                    P[km][qn] *= inv;
                    wsum[km][qn] += P[km][qn];
                }
                if (a.p_heads) {
                    const int qi = 16 * qn + x;
#pragma unroll
                    for (int km = 0; km < NT; ++km)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kj = 16 * km + 4 * qq + r;
                            if (qi < Lq && kj < Lk) a.p_heads[(((int64_t)b * H + h) * Lq + qi) * Lk + kj] = P[km][qn][r];
                        }
                }
            }
            // ---- weighted sum of the values: O[c][query] = sum_keys V[key][c] P[key][query] ----------------------------
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 16 * ct + x;
                f32x4 va[NT];
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kj = 16 * km + 4 * qq + r;
                        va[km][r] = (c < d && kj < Lk) ? vb[(int64_t)kj * RP + hc + c] : 0.f;
                    }
#pragma unroll
                for (int qn = 0; qn < NT; ++qn) {
                    if (qn >= QT) continue;
                    f32x4 o = zero4();
#pragma unroll
                    for (int km = 0; km < NT; ++km)
                        if (km < KT) o = mfma4(va[km], P[km][qn], o);
                    const int qi = 16 * qn + x, c0 = 16 * ct + 4 * qq;
                    if (qi < Lq) {
                        float* dst = a.out + ((int64_t)b * Lq + qi) * E + hc + c0;
                        if constexpr (kF) {
                            if (a.ow && c0 < d) *reinterpret_cast<f32x4*>(&s_o[qi][hc + c0]) = o;      // (head_dim % 4 == 0 here)
                        }
                        if (c0 + 3 < d) *reinterpret_cast<f32x4*>(dst) = o;
                        else
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (c0 + r < d) dst[r] = o[r];
                    }
                }
            }
        } else {
            // ---- backward of head h ---------------------------------------------------------------------------------
#pragma unroll
            for (int qn = 0; qn < NT; ++qn) {
                const int qi = 16 * qn + x;
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kj = 16 * km + 4 * qq + r;
                        P[km][qn][r] = (qi < Lq && kj < Lk) ? a.p_heads[(((int64_t)b * H + h) * Lq + qi) * Lk + kj] : 0.f;
                    }
            }
            // dP[key][query] = sum_c V[key][c] dO[query][c]  (+ the head's share of the averaged weights' gradient)
            f32x4 dP[NT][NT];
#pragma unroll
            for (int km = 0; km < NT; ++km)
#pragma unroll
                for (int qn = 0; qn < NT; ++qn) dP[km][qn] = zero4();
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 va[NT], go[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int kj = 16 * t + x, qi = 16 * t + x;
                    va[t] = row4(vb + (int64_t)min(kj, Lk - 1) * E + hc, t < KT && kj < Lk, 16 * ct + 4 * qq, d);
                    go[t] = row4(gob + (int64_t)min(qi, Lq - 1) * GP + hc, t < QT && qi < Lq, 16 * ct + 4 * qq, d);
                }
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int qn = 0; qn < NT; ++qn)
                        if (km < KT && qn < QT) dP[km][qn] = mfma4(va[km], go[qn], dP[km][qn]);
            }
            f32x4 dS[NT][NT];
#pragma unroll
            for (int qn = 0; qn < NT; ++qn) {
                const int qi = 16 * qn + x;
                const float kp = dead[qn] ? 0.f : 1.f / (float)H;       // weights returned = mean_h(P) * keep
                float dot = 0.f;
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kj = 16 * km + 4 * qq + r;
                        if (a.g_w && qi < Lq && kj < Lk) dP[km][qn][r] += kp * a.g_w[((int64_t)b * Lq + qi) * Lk + kj];
                        dot += P[km][qn][r] * dP[km][qn][r];
                    }
                dot = sum_over_q(dot);
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dS[km][qn][r] = P[km][qn][r] * (dP[km][qn][r] - dot);
            }
            // dQ[query][c] = scale * sum_keys K[key][c] dS[key][query]
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 16 * ct + x;
                f32x4 ka[NT];
#pragma unroll
                for (int km = 0; km < NT; ++km)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kj = 16 * km + 4 * qq + r;
                        ka[km][r] = (c < d && kj < Lk) ? kb[(int64_t)kj * E + hc + c] : 0.f;
                    }
#pragma unroll
                for (int qn = 0; qn < NT; ++qn) {
                    if (qn >= QT) continue;
                    f32x4 o = zero4();
#pragma unroll
                    for (int km = 0; km < NT; ++km)
                        if (km < KT) o = mfma4(ka[km], dS[km][qn], o);
                    const int qi = 16 * qn + x, c0 = 16 * ct + 4 * qq;
                    if (qi < Lq) {
                        float* dst = a.g_q + ((int64_t)b * Lq + qi) * E + hc + c0;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c0 + r < d) dst[r] = o[r] * scale;
                        if constexpr (kB) {
                            if (c0 < d) *reinterpret_cast<f32x4*>(&s_g3[0][qi][hc + c0]) = o * scale;      // (head_dim % 4 == 0 here)
                        }
                    }
                }
            }
            // the products over the QUERIES: P and dS tiles turned (lane (qq, x = key) then holds queries 4 s + qq)
            wave_sync();
#pragma unroll
            for (int km = 0; km < NT; ++km)
#pragma unroll
                for (int qn = 0; qn < NT; ++qn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tbuf[((km * NT + qn) * 16 * kTP) + (4 * qq + r) * kTP + tpos(x)] = P[km][qn][r];
                        tbuf[((NT * NT + km * NT + qn) * 16 * kTP) + (4 * qq + r) * kTP + tpos(x)] = dS[km][qn][r];
                    }
            wave_sync();
            // dV[key][c] = sum_queries P[key][query] dO[query][c];  dK[key][c] = scale * sum_queries dS[key][query] Q[query][c]
            for (int ct = 0; ct < CT; ++ct) {
                const int c = 16 * ct + x;
                f32x4 ga[NT], qa[NT];       // A[i = c][k-slot (qq, s) <-> query 16 qn + 4 s + qq]
#pragma unroll
                for (int qn = 0; qn < NT; ++qn)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int qi = 16 * qn + 4 * s + qq;
                        const bool ok = c < d && qi < Lq;
                        ga[qn][s] = ok ? gob[(int64_t)qi * GP + hc + c] : 0.f;
                        qa[qn][s] = ok ? qb[(int64_t)qi * E + hc + c] : 0.f;
                    }
#pragma unroll
                for (int km = 0; km < NT; ++km) {
                    if (km >= KT) continue;
                    f32x4 dv = zero4(), dk = zero4();
#pragma unroll
                    for (int qn = 0; qn < NT; ++qn) {
                        if (qn >= QT) continue;
                        const f32x4 pt = *reinterpret_cast<const f32x4*>(tbuf + (km * NT + qn) * 16 * kTP + x * kTP + 4 * qq);
                        const f32x4 st = *reinterpret_cast<const f32x4*>(tbuf + (NT * NT + km * NT + qn) * 16 * kTP + x * kTP + 4 * qq);
                        dv = mfma4(ga[qn], pt, dv);
                        dk = mfma4(qa[qn], st, dk);
                    }
                    const int kj = 16 * km + x, c0 = 16 * ct + 4 * qq;
                    if (kj < Lk) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c0 + r < d) {
                                a.g_v[((int64_t)b * Lk + kj) * E + hc + c0 + r] = dv[r];
                                a.g_k[((int64_t)b * Lk + kj) * E + hc + c0 + r] = dk[r] * scale;
                            }
                        if constexpr (kB) {
                            if (c0 < d) {
                                *reinterpret_cast<f32x4*>(&s_g3[2][kj][hc + c0]) = dv;
                                *reinterpret_cast<f32x4*>(&s_g3[1][kj][hc + c0]) = dk * scale;
                            }
                        }
                    }
                }
            }
        }
    }
    if constexpr (kB) {
        // the projections' input gradient (the arithmetic of k_rows_proj_bwd): gx = g_q W_q + g_k W_k + g_v W_v, the queries
        // being the entry's last Lq positions
        constexpr int EE = 16 * PEC;
        __syncthreads();
        const bool live = x < Lk;
        const int skip = Lk - Lq;
        f32x4 gv[3][PEC];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int rj = j == 0 ? x - skip : x;
            const bool on = live && rj >= 0;
#pragma unroll
            for (int c = 0; c < PEC; ++c) gv[j][c] = on ? *reinterpret_cast<const f32x4*>(&s_g3[j][max(rj, 0)][16 * c + 4 * qq]) : zero4();
        }
        for (int kt = wv; kt < PEC; kt += kWaves) {
            f32x4 acc = zero4();
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < PEC; ++c) acc = mfma4(wcol4(a.pw[j], 16 * c + 4 * qq, 16 * kt + x), gv[j][c], acc);
            if (live) *reinterpret_cast<f32x4*>(a.gx + ((int64_t)b * Lk + x) * EE + 16 * kt + 4 * qq) = acc;
        }
    }
    if (!BWD) {
        // the waves' sums over their heads, added in wave order by wave 0 (s_t is free in the forward)
        f32x4* ws = reinterpret_cast<f32x4*>(&s_t[0][0][0]);
#pragma unroll
        for (int km = 0; km < NT; ++km)
#pragma unroll
            for (int qn = 0; qn < NT; ++qn) ws[(wv * NT * NT + km * NT + qn) * 64 + l] = wsum[km][qn];
        __syncthreads();
        if constexpr (kF) {
            if (a.ow) {
                // the output ResBlock over the entry's <= 16 query rows (the arithmetic of csrc/rows_proj.hip's k_rows_res_fwd)
                constexpr int EE = 16 * PEC;
                const bool live = x < Lq;
                f32x4 ov[PEC];
#pragma unroll
                for (int c = 0; c < PEC; ++c) ov[c] = live ? *reinterpret_cast<const f32x4*>(&s_o[x][16 * c + 4 * qq]) : zero4();
                float sc = dead[0] ? 0.f : 1.f;
                if (a.row_zero && live && a.row_zero[(int64_t)b * a.rz_sb + x]) sc = 0.f;
                for (int nt = wv; nt < PEC; nt += kWaves) {
                    const float* wp = a.ow + (16 * nt + x) * EE + 4 * qq;
                    f32x4 acc = zero4();
#pragma unroll
                    for (int c = 0; c < PEC; ++c) acc = mfma4(*reinterpret_cast<const f32x4*>(wp + 16 * c), ov[c], acc);
                    acc += *reinterpret_cast<const f32x4*>(a.ob + 16 * nt + 4 * qq);
                    if (live) {
                        const f32x4 res = *reinterpret_cast<const f32x4*>(&s_o[x][16 * nt + 4 * qq]);
                        f32x4 yv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) yv[r] = (gelu_f(acc[r]) + res[r]) * sc;
                        const int64_t at = ((int64_t)b * Lq + x) * EE + 16 * nt + 4 * qq;
                        *reinterpret_cast<f32x4*>(a.y + at) = yv;
                        *reinterpret_cast<f32x4*>(a.pre + at) = acc;
                    }
                }
            }
        }
        if (wv != 0) return;
        const float invH = 1.f / (float)H;
#pragma unroll
        for (int qn = 0; qn < NT; ++qn) {
            const int qi = 16 * qn + x;
            if (qi >= Lq) continue;
            const float kp = dead[qn] ? 0.f : 1.f;
            if (qq == 0) {
                a.keep[(int64_t)b * Lq + qi] = kp;
                if (a.keep_rows)
                    a.keep_rows[(int64_t)b * Lq + qi] = (a.row_zero && a.row_zero[(int64_t)b * a.rz_sb + qi]) ? 0.f : kp;
            }
#pragma unroll
            for (int km = 0; km < NT; ++km) {
                f32x4 t = wsum[km][qn];
#pragma unroll
                for (int w2 = 1; w2 < kWaves; ++w2) t += ws[(w2 * NT * NT + km * NT + qn) * 64 + l];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kj = 16 * km + 4 * qq + r;
                    if (kj < Lk) a.w_avg[((int64_t)b * Lq + qi) * Lk + kj] = t[r] * invH * kp;
                }
            }
        }
    }
}

}  // namespace amh
}  // namespace asac

using namespace asac;
using namespace asac::amh;

// windows of <= 16 positions are one tile a side: a quarter of the tile registers and of the LDS of the 2 x 2 form, so every
// entry of a batch of 1024 is resident at once (56 / 92 VGPRs against 121 / 164; 9.8 / 14.0 us against 20 / 27 at window 9)
template <bool BWD>
static void launch(const Args& a, hipStream_t stream) {
    const dim3 grid((unsigned)a.B);
    if (a.Lq <= 16 && a.Lk <= 16) ASAC_LAUNCH((k_attn_mh<BWD, 1>), grid, dim3(kThreads), 0, stream, a);
    else ASAC_LAUNCH((k_attn_mh<BWD, 2>), grid, dim3(kThreads), 0, stream, a);
}

extern "C" {

int asac_attention_mh_supported(int Lq, int Lk, int heads, int head_dim) {
    return Lq >= 1 && Lk >= 1 && Lq <= kMaxL && Lk <= kMaxL && heads >= 1 && heads <= 16 && head_dim >= 1 && head_dim <= 64;
}

int asac_attention_mh_forward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                              int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                              float* out, float* weights, float* keep, float* p_heads, const uint8_t* row_zero,
                              float* keep_rows, void* stream) {
    if (!q || !k || !v || !out || !weights || !keep || B <= 0 || !asac_attention_mh_supported(Lq, Lk, heads, head_dim))
        return bad_arg("asac_attention_mh_forward");
    Args a{};
    a.q = q, a.k = k, a.v = v, a.mask = mask, a.m_sb = mask_stride_b, a.m_si = mask_stride_q, a.m_sj = mask_stride_k;
    a.B = B, a.Lq = Lq, a.Lk = Lk, a.H = heads, a.d = head_dim, a.out = out, a.w_avg = weights, a.keep = keep, a.p_heads = p_heads;
    a.row_zero = row_zero, a.rz_sb = Lq, a.keep_rows = keep_rows;
    launch<false>(a, as_stream(stream));
    return finish_launch("asac_attention_mh_forward");
}

int asac_attention_mh_proj_supported(int Lq, int Lk, int heads, int head_dim) {
    const int E = heads * head_dim;
    return asac_attention_mh_supported(Lq, Lk, heads, head_dim) && Lk <= 16 && Lq <= Lk && (E == 32 || E == 64 || E == 128) &&
           head_dim % 4 == 0;
}

int asac_attention_mh_proj_forward(const float* x, int64_t x_stride_b, int64_t x_stride_t, const float* const* weights,
                                   const float* const* biases, const uint8_t* mask, int64_t mask_stride_b,
                                   int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                                   float* q, float* k, float* v, float* out, float* attn_weights, float* keep, float* p_heads,
                                   const uint8_t* row_zero, int64_t row_zero_stride_b, float* keep_rows, const float* out_weight,
                                   const float* out_bias, float* y, float* pre, void* stream) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!x || !weights || !biases || !q || !k || !v || !out || !attn_weights || !keep || B <= 0 ||
        !asac_attention_mh_proj_supported(Lq, Lk, heads, head_dim) || !al(x) || (x_stride_b & 3) || (x_stride_t & 3) || !al(q) ||
        !al(k) || !al(v) || (row_zero && row_zero_stride_b < Lq) || (out_weight && (!out_bias || !y || !pre || !al(out_weight) || !al(out_bias) || !al(y) || !al(pre))))
        return bad_arg("asac_attention_mh_proj_forward");
    Args a{};
    a.q = q, a.k = k, a.v = v, a.mask = mask, a.m_sb = mask_stride_b, a.m_si = mask_stride_q, a.m_sj = mask_stride_k;
    a.B = B, a.Lq = Lq, a.Lk = Lk, a.H = heads, a.d = head_dim, a.out = out, a.w_avg = attn_weights, a.keep = keep, a.p_heads = p_heads;
    a.row_zero = row_zero, a.rz_sb = row_zero_stride_b, a.keep_rows = keep_rows;
    a.x = x, a.xs_b = x_stride_b, a.xs_t = x_stride_t;
    a.ow = out_weight, a.ob = out_bias, a.y = y, a.pre = pre;
    for (int j = 0; j < 3; ++j) {
        if (!weights[j] || !biases[j] || !al(weights[j]) || !al(biases[j])) return bad_arg("asac_attention_mh_proj_forward: job");
        a.pw[j] = weights[j], a.pb[j] = biases[j];
    }
    const dim3 grid((unsigned)B);
    const int E = heads * head_dim;
    hipStream_t s = as_stream(stream);
    if (E == 32) ASAC_LAUNCH((k_attn_mh<false, 1, 2>), grid, dim3(kThreads), 0, s, a);
    else if (E == 64) ASAC_LAUNCH((k_attn_mh<false, 1, 4>), grid, dim3(kThreads), 0, s, a);
    else ASAC_LAUNCH((k_attn_mh<false, 1, 8>), grid, dim3(kThreads), 0, s, a);
    return finish_launch("asac_attention_mh_proj_forward");
}

int asac_attention_mh_block_backward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                                     int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                                     const float* p_heads, const float* grad_y, int64_t grad_y_stride_b, int64_t grad_y_stride_t,
                                     const float* pre, const float* row_scale, const float* out_weight, const float* grad_weights,
                                     const float* const* proj_weights, float* grad_q, float* grad_k, float* grad_v, float* grad_pre,
                                     float* grad_x, void* stream) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!q || !k || !v || !p_heads || !grad_y || !pre || !out_weight || !proj_weights || !grad_q || !grad_k || !grad_v || !grad_pre ||
        !grad_x || B <= 0 || !asac_attention_mh_proj_supported(Lq, Lk, heads, head_dim) || !al(grad_y) || !al(pre) || !al(grad_pre) ||
        !al(grad_x) || (grad_y_stride_b & 3) || (grad_y_stride_t & 3) || grad_y_stride_t < heads * head_dim)
        return bad_arg("asac_attention_mh_block_backward");
    Args a{};
    a.q = q, a.k = k, a.v = v, a.mask = mask, a.m_sb = mask_stride_b, a.m_si = mask_stride_q, a.m_sj = mask_stride_k;
    a.B = B, a.Lq = Lq, a.Lk = Lk, a.H = heads, a.d = head_dim;
    a.p_heads = const_cast<float*>(p_heads), a.g_w = grad_weights, a.g_q = grad_q, a.g_k = grad_k, a.g_v = grad_v;
    a.gy = grad_y, a.gy_sb = grad_y_stride_b, a.gy_st = grad_y_stride_t, a.pre = const_cast<float*>(pre), a.scale_in = row_scale, a.ow = out_weight, a.gpre = grad_pre, a.gx = grad_x;
    for (int j = 0; j < 3; ++j) {
        if (!proj_weights[j]) return bad_arg("asac_attention_mh_block_backward: weights");
        a.pw[j] = proj_weights[j];
    }
    const dim3 grid((unsigned)B);
    const int E = heads * head_dim;
    hipStream_t s = as_stream(stream);
    if (E == 32) ASAC_LAUNCH((k_attn_mh<true, 1, 2>), grid, dim3(kThreads), 0, s, a);
    else if (E == 64) ASAC_LAUNCH((k_attn_mh<true, 1, 4>), grid, dim3(kThreads), 0, s, a);
    else ASAC_LAUNCH((k_attn_mh<true, 1, 8>), grid, dim3(kThreads), 0, s, a);
    return finish_launch("asac_attention_mh_block_backward");
}

int asac_attention_mh_backward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                               int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                               const float* p_heads, const float* grad_out, const float* grad_weights, float* grad_q,
                               float* grad_k, float* grad_v, void* stream) {
    if (!q || !k || !v || !p_heads || !grad_out || !grad_q || !grad_k || !grad_v || B <= 0 ||
        !asac_attention_mh_supported(Lq, Lk, heads, head_dim))
        return bad_arg("asac_attention_mh_backward");
    Args a{};
    a.q = q, a.k = k, a.v = v, a.mask = mask, a.m_sb = mask_stride_b, a.m_si = mask_stride_q, a.m_sj = mask_stride_k;
    a.B = B, a.Lq = Lq, a.Lk = Lk, a.H = heads, a.d = head_dim;
    a.p_heads = const_cast<float*>(p_heads), a.g_out = grad_out, a.g_w = grad_weights, a.g_q = grad_q, a.g_k = grad_k, a.g_v = grad_v;
    launch<true>(a, as_stream(stream));
    return finish_launch("asac_attention_mh_backward");
}

}  // extern "C"
