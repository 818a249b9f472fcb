// GRU recurrence for the hidden sizes the reference's environments use (32 / 64 / 128: `m.GRU(…, 64, 1)`
// envs/square/memory_corridor/nn.py:19, `m.GRU(…, 128, 1)` envs/uav/uav_hole/nn.py:22; layer: reference
// algorithm/nn_models/layers/seq_layers.py:14-114) on f32 MFMA.  csrc/gru.hip (one lane per hidden unit) stops at 16.
//
// What is a plain GEMM stays a library GEMM (the host side, algorithm/fused_gru_wide.py): the input projections of ALL steps
// gi = x W_ih^T + b_ih before the recurrence, and after the backward recurrence dW_ih = dgi^T x, dW_hh = dgh^T h_prev,
// dx = dgi W_ih and the bias sums.  Hand-written here: the two time loops — L dependent steps that MIOpen runs as one launch
// per step and layer.
//
// A workgroup owns 16 rows of the batch for the whole window; rows are the N dimension of v_mfma_f32_16x16x4_f32 (the tile
// scheme of csrc/decoder.hip: lane (q, x) of an accumulator holds units 4 q + r of a 16-unit block for row x, and that float4
// IS the B operand of the next product's four k-steps).  The recurrent weights live in REGISTERS for the whole loop, read where
// they lie (a lane's four k are 16 contiguous bytes of a W_hh row; the backward takes W_hh^T): wave w owns the unit blocks
// w, w + WAVES, … with their r / z / n rows, so the gate arithmetic of a unit happens in one lane (four waves; eight for hidden
// 128: one unit block a wave, as hidden 64 has — with two blocks a wave the loops below had no registers for their second input
// set and the deferred stores: forward 331 -> 322 / 291 -> 272 us, backward 351 -> 298 us at 256 x 81).  Per step: state tiles from LDS
// (double-buffered, one barrier), 3 H / 16 x H / 4 MFMAs over the workgroup, the gates, stores nobody waits on.
// Padding as csrc/gru.hip: steps before a row's first unpadded one are skipped (state held), the output of a padded step is 0.
#include "asac_common.h"

#include <cmath>
#include <type_traits>

namespace asac {
namespace gruw {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#define GW_MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
    c = GW_MF(a[0], b[0], c);
    c = GW_MF(a[1], b[1], c);
    c = GW_MF(a[2], b[2], c);
    c = GW_MF(a[3], b[3], c);
    return c;
}
// sigmoid / tanh on the hardware exp2 and reciprocal, as csrc/gru.hip (absolute error ~1e-7; on the serial chain of a step)
__device__ __forceinline__ float sigmoidf_(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
}
__device__ __forceinline__ float tanhf_(float v) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177793f * v));
}

// workgroup barrier that orders LDS traffic only: `__syncthreads()` is a full fence and would wait, every step, for the
// step's stores and the next step's prefetched loads
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FwdArgs {
    const float* gi; int64_t gi_sb, gi_st;      // [B][L][3H]: x W_ih^T + b_ih
    const float* w_hh;                          // [3H][H]
    const float* b_hh;                          // [3H]
    const float* h0; int64_t h0_sb;             // [B][H] (row stride h0_sb) or NULL
    const uint8_t* pad; int64_t pad_sb;         // [B][L] or NULL
    float* out; int64_t out_sb, out_st;         // the layer's masked output (may be a slice of hn [B][L][layers][H])
    float* hraw;                                // [B][L][H] the unmasked state after every step, or NULL
    float* gates;                               // [B][L][4H]: r | z | n | W_hn h + b_hn, or NULL
    int32_t B, L;
    // twin form (twin_B > 0): rows [twin_B, B) are the SAME windows under a second network's recurrent weights (the target
    // representation beside the online one): gi / out hold 2 x twin_B rows, h0 / pad / hraw / gates twin_B rows
    const float* w_hh2; const float* b_hh2;
    int32_t twin_B;
};

// first unpadded step of a row (0 without a mask or when the whole row is padded)
__device__ __forceinline__ int lead_of(const uint8_t* pad, int64_t pad_sb, int64_t row, int L) {
    if (!pad) return 0;
    const uint8_t* p = pad + row * pad_sb;
    for (int t = 0; t < L; ++t)
        if (!p[t]) return t;
    return 0;
}

// The time loop is written so that the compiler can wait for a step's inputs BY COUNT (`s_waitcnt vmcnt(n)`, the later stores
// still in flight) instead of draining the queue every step (a store round trip a step: 0.4 of 1.65 us with the saves):
//   * two input register sets, the loop unrolled by two — no register copies that would pin a wait to the loop's back edge;
//   * the same vector-memory instructions on every path: lanes beyond the batch replicate the last row (same inputs, same
//     arithmetic) and rewrite its values; with SAVE (h_raw and gates given) the twin's second network, which saves nothing,
//     stores its output in their place; the request after the last step re-reads it; without a mask the byte comes from w_hh.
template <int HB, bool SAVE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_gruw_fwd(const FwdArgs a) {
    constexpr int H = 16 * HB, NBW = (HB + WAVES - 1) / WAVES;
    __shared__ f32x4 s_h[2][HB][64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    __builtin_assume(w >= 0 && w < WAVES);
    const int64_t row = min((int64_t)blockIdx.x * 16 + x, (int64_t)a.B - 1);
    const bool second = a.twin_B > 0 && (int64_t)blockIdx.x * 16 >= a.twin_B;      // (workgroup-uniform: twin_B % 16 == 0)
    const float* w_hh = second ? a.w_hh2 : a.w_hh;
    const float* b_hh = second ? a.b_hh2 : a.b_hh;
    const int64_t srow = second ? row - a.twin_B : row;      // the row of h0 / pad / the saves (shared by the two networks)
    f32x4 wr[NBW][3][HB], bh[NBW][3], h[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int ub = w + WAVES * i;
        if (ub >= HB) continue;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) {
#pragma unroll
            for (int kt = 0; kt < HB; ++kt)
                wr[i][gate][kt] = *reinterpret_cast<const f32x4*>(w_hh + (int64_t)(gate * H + 16 * ub + x) * H + 16 * kt + 4 * q);
            bh[i][gate] = *reinterpret_cast<const f32x4*>(b_hh + gate * H + 16 * ub + 4 * q);
        }
        h[i] = a.h0 ? *reinterpret_cast<const f32x4*>(a.h0 + srow * a.h0_sb + 16 * ub + 4 * q) : zero4();
        s_h[0][ub][l] = h[i];
    }
    const int lead = lead_of(a.pad, a.pad_sb, srow, a.L);
    struct StepIn {
        f32x4 gi[NBW][3];
        uint8_t pad;
    };
    const uint8_t* const pad_row = a.pad ? a.pad + srow * a.pad_sb : reinterpret_cast<const uint8_t*>(a.w_hh);
    const int pad_step = a.pad ? 1 : 0;
    auto request = [&](int t, StepIn& in) {
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int ub = w + WAVES * i;
            if (ub >= HB) continue;
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
                in.gi[i][gate] = *reinterpret_cast<const f32x4*>(a.gi + row * a.gi_sb + t * a.gi_st + gate * H + 16 * ub + 4 * q);
        }
        in.pad = pad_row[t * pad_step];
    };
    // a step's results leave in the NEXT step, between its products (one store per few tiles): issued together at the end of
    // their own step, the four waves' 24 KB queued at the CU's one vector-memory port (64 B a cycle) and every wave stood at the
    // barrier until its last store had been taken (0.4 of the step's 1.6 us)
    constexpr int NS = SAVE ? 6 : 1;
    f32x4 late[NBW][NS];        // ov | hv, rg, zg, ng, hn of the step before
    auto store_late = [&](int t, int i, int sidx) {
        const int col = 16 * (w + WAVES * i) + 4 * q;
        float* const op = a.out + row * a.out_sb + t * a.out_st + col;
        float* dst = op;
        if (sidx == 1) dst = a.hraw + (srow * a.L + t) * H + col;
        if (sidx >= 2) dst = a.gates + (srow * a.L + t) * (4 * H) + (sidx - 2) * H + col;
        *reinterpret_cast<f32x4*>(second ? op : dst) = late[i][second ? 0 : sidx];
    };
    auto step = [&](int t, const StepIn& in, auto has_late) {
        const int cur = t & 1;
        const bool padded = a.pad && in.pad != 0, active = t >= lead;
        f32x4 hb[HB];
#pragma unroll
        for (int kt = 0; kt < HB; ++kt) hb[kt] = s_h[cur][kt][l];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int ub = w + WAVES * i;
            if (ub >= HB) continue;
            f32x4 ar = zero4(), az = zero4(), an = zero4();
#pragma unroll
            for (int kt = 0; kt < HB; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ar = GW_MF(wr[i][0][kt][r], hb[kt][r], ar);
                    az = GW_MF(wr[i][1][kt][r], hb[kt][r], az);
                    an = GW_MF(wr[i][2][kt][r], hb[kt][r], an);
                }
                if (decltype(has_late)::value) {
#pragma unroll
                    for (int sidx = 0; sidx < NS; ++sidx)
                        if (sidx * HB / NS == kt) store_late(t - 1, i, sidx);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            f32x4 rg, zg, ng, hn, hv, ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rg[r] = sigmoidf_(in.gi[i][0][r] + (ar[r] + bh[i][0][r]));
                zg[r] = sigmoidf_(in.gi[i][1][r] + (az[r] + bh[i][1][r]));
                hn[r] = an[r] + bh[i][2][r];
                ng[r] = tanhf_(in.gi[i][2][r] + rg[r] * hn[r]);
                const float hnew = (1.f - zg[r]) * ng[r] + zg[r] * h[i][r];
                hv[r] = active ? hnew : h[i][r];
                ov[r] = padded ? 0.f : hv[r];
            }
            h[i] = hv;
            s_h[cur ^ 1][ub][l] = hv;
            late[i][0] = ov;
            if (SAVE) late[i][1 % NS] = hv, late[i][2 % NS] = rg, late[i][3 % NS] = zg, late[i][4 % NS] = ng, late[i][5 % NS] = hn;
        }
        lds_barrier();
    };
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    StepIn in_a, in_b;
    request(0, in_a);
    __syncthreads();
    // (the first pair outside the loop: the loop is then entered in the state its back edge leaves — requested inputs followed
    // by a step's stores — and the count holds on both paths)
    int t = 0;
    if (a.L >= 2) {
        request(1, in_b);
        step(0, in_a, no);
        request(min(2, a.L - 1), in_a);
        step(1, in_b, yes);
        for (t = 2; t + 1 < a.L; t += 2) {
            request(t + 1, in_b);
            step(t, in_a, yes);
            request(min(t + 2, a.L - 1), in_a);
            step(t + 1, in_b, yes);
        }
        if (t < a.L) step(t, in_a, yes);      // (odd L: in_a holds step L - 1)
    } else {
        step(0, in_a, no);
    }
    // the last step's results
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        if (w + WAVES * i >= HB) continue;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) store_late(a.L - 1, i, sidx);
    }
}

struct BwdArgs {
    const float* gout; int64_t go_sb, go_st;    // [B][L][H] gradient of the layer's masked output
    const float* w_hh_t;                        // [H][3H] = W_hh^T
    const float* gates;                         // [B][L][4H]
    const float* hraw;                          // [B][L][H]
    const float* h0; int64_t h0_sb;
    const uint8_t* pad; int64_t pad_sb;
    float* dgi;                                 // [B][L][3H] gradient of gi (zero where a step did not run)
    float* dgh;                                 // [B][L][3H] gradient of W_hh h + b_hh
    float* dh0;                                 // [B][H] or NULL
    int32_t B, L;
};

// (the time loop is shaped like k_gruw_fwd's: two input register sets, the same vector-memory instructions on every path)
template <int HB, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_gruw_bwd(const BwdArgs a) {
    constexpr int H = 16 * HB, NBW = (HB + WAVES - 1) / WAVES, KT = 3 * HB;
    __shared__ f32x4 s_g[2][KT][64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    __builtin_assume(w >= 0 && w < WAVES);
    const int64_t row = min((int64_t)blockIdx.x * 16 + x, (int64_t)a.B - 1);      // (lanes beyond the batch replicate the last row)
    const bool live = (int64_t)blockIdx.x * 16 + x < a.B;
    f32x4 wt[NBW][KT], dh[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int ub = w + WAVES * i;
        dh[i] = zero4();
        if (ub >= HB) continue;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            wt[i][kt] = *reinterpret_cast<const f32x4*>(a.w_hh_t + (int64_t)(16 * ub + x) * (3 * H) + 16 * kt + 4 * q);
    }
    const int lead = lead_of(a.pad, a.pad_sb, row, a.L);
    struct StepIn {
        f32x4 g[NBW], rg[NBW], zg[NBW], ng[NBW], hn[NBW], hp[NBW];
        uint8_t pad;
    };
    const uint8_t* const pad_row = a.pad ? a.pad + row * a.pad_sb : reinterpret_cast<const uint8_t*>(a.w_hh_t);
    const int pad_step = a.pad ? 1 : 0;
    auto request = [&](int t, StepIn& in) {
        in.pad = pad_row[t * pad_step];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int ub = w + WAVES * i;
            if (ub >= HB) continue;
            const int col = 16 * ub + 4 * q;
            in.g[i] = *reinterpret_cast<const f32x4*>(a.gout + row * a.go_sb + t * a.go_st + col);
            const float* gp = a.gates + (row * a.L + t) * (4 * H) + col;
            in.rg[i] = *reinterpret_cast<const f32x4*>(gp), in.zg[i] = *reinterpret_cast<const f32x4*>(gp + H);
            in.ng[i] = *reinterpret_cast<const f32x4*>(gp + 2 * H), in.hn[i] = *reinterpret_cast<const f32x4*>(gp + 3 * H);
            // the state before the step: h_raw of the step before, h0 before step 0 — or zero: then any address, the value is dropped
            const float* hp_at = t > 0 ? a.hraw + (row * a.L + t - 1) * H + col
                                       : (a.h0 ? a.h0 + row * a.h0_sb + col : a.hraw + row * a.L * H + col);
            in.hp[i] = *reinterpret_cast<const f32x4*>(hp_at);      // (dropped in the step when there is no such state)
        }
    };
    auto step = [&](int t, const StepIn& in) {
        const int cur = t & 1;
        const bool padded = a.pad && in.pad != 0;
        const bool active = t >= lead;
        f32x4 direct[NBW], st_v[NBW][4];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int ub = w + WAVES * i;
            if (ub >= HB) continue;
            const f32x4 g = padded ? zero4() : in.g[i];
            const f32x4 rg = in.rg[i], zg = in.zg[i], ng = in.ng[i], hn = in.hn[i];
            const f32x4 hp = (t > 0 || a.h0) ? in.hp[i] : zero4();
            f32x4 d_r, d_z, d_n, d_nh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dht = dh[i][r] + g[r];
                const float dn = dht * (1.f - zg[r]), dz = dht * (hp[r] - ng[r]);
                const float dn_pre = dn * (1.f - ng[r] * ng[r]);
                const float dz_pre = dz * zg[r] * (1.f - zg[r]);
                const float dr_pre = dn_pre * hn[r] * rg[r] * (1.f - rg[r]);
                d_r[r] = active ? dr_pre : 0.f;
                d_z[r] = active ? dz_pre : 0.f;
                d_n[r] = active ? dn_pre : 0.f;
                d_nh[r] = active ? dn_pre * rg[r] : 0.f;
                direct[i][r] = active ? dht * zg[r] : dht;
            }
            st_v[i][0] = d_r, st_v[i][1] = d_z, st_v[i][2] = d_n, st_v[i][3] = d_nh;
            s_g[cur][0 * HB + ub][l] = d_r;
            s_g[cur][1 * HB + ub][l] = d_z;
            s_g[cur][2 * HB + ub][l] = d_nh;
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int ub = w + WAVES * i;
            if (ub >= HB) continue;
            f32x4 acc0 = zero4(), acc1 = zero4();      // two chains: the products are the whole step
            // the step's six stores go out BETWEEN the products, one per pair of tiles: issued together before the barrier, the
            // four waves' 24 KB queued at the CU's one vector-memory port (64 B a cycle) and every wave stood at the barrier
            // until its last store had been taken (0.7 of the step's 2.1 us)
            const int col = 16 * ub + 4 * q;
            float* const o = a.dgi + (row * a.L + t) * (3 * H) + col;
            float* const o2 = a.dgh + (row * a.L + t) * (3 * H) + col;
#pragma unroll
            for (int kt = 0; kt < KT; kt += 2) {
                acc0 = mfma4(wt[i][kt], s_g[cur][kt][l], acc0);
                if (kt + 1 < KT) acc1 = mfma4(wt[i][kt + 1], s_g[cur][kt + 1][l], acc1);
                constexpr int kChunks = (KT + 1) / 2;
                const int c = kt / 2;
#pragma unroll
                for (int sidx = 0; sidx < 6; ++sidx) {
                    if (sidx * kChunks / 6 != c) continue;
                    float* const dst = (sidx < 3 ? o : o2) + (sidx % 3) * H;
                    *reinterpret_cast<f32x4*>(dst) = st_v[i][sidx == 5 ? 3 : (sidx % 3 == 2 ? 2 : sidx % 3)];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            dh[i] = direct[i] + (acc0 + acc1);
        }
    };
    // steps L - 1 ... 0, in pairs; the first pair outside the loop (see k_gruw_fwd)
    StepIn in_a, in_b;
    int t = a.L - 1;
    request(t, in_a);
    if (a.L >= 2) {
        request(t - 1, in_b);
        step(t, in_a);
        request(max(t - 2, 0), in_a);
        step(t - 1, in_b);
        for (t -= 2; t >= 1; t -= 2) {
            request(t - 1, in_b);
            step(t, in_a);
            request(max(t - 2, 0), in_a);
            step(t - 1, in_b);
        }
    }
    if (t == 0) step(0, in_a);      // (odd L, or L == 1: in_a holds step 0)
    if (a.dh0 && live) {
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int ub = w + WAVES * i;
            if (ub >= HB) continue;
            *reinterpret_cast<f32x4*>(a.dh0 + row * H + 16 * ub + 4 * q) = dh[i];
        }
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// hidden 128: eight waves (one unit block each, as hidden 64 has with four) — with four, two unit blocks a wave left no registers
// for the second input set and the deferred stores
static void launch_fwd(int hidden, bool save, dim3 grid, hipStream_t s, const FwdArgs& a) {
    if (save) {
        if (hidden == 32) ASAC_LAUNCH((k_gruw_fwd<2, true, 4>), grid, dim3(256), 0, s, a);
        else if (hidden == 64) ASAC_LAUNCH((k_gruw_fwd<4, true, 4>), grid, dim3(256), 0, s, a);
        else ASAC_LAUNCH((k_gruw_fwd<8, true, 8>), grid, dim3(512), 0, s, a);
    } else {
        if (hidden == 32) ASAC_LAUNCH((k_gruw_fwd<2, false, 4>), grid, dim3(256), 0, s, a);
        else if (hidden == 64) ASAC_LAUNCH((k_gruw_fwd<4, false, 4>), grid, dim3(256), 0, s, a);
        else ASAC_LAUNCH((k_gruw_fwd<8, false, 8>), grid, dim3(512), 0, s, a);
    }
}

}  // namespace gruw
}  // namespace asac

using namespace asac;
using namespace asac::gruw;

extern "C" {

int asac_gru_wide_supported(int hidden) { return hidden == 32 || hidden == 64 || hidden == 128; }

int asac_gru_wide_forward(const float* gi, int64_t gi_stride_b, int64_t gi_stride_t, const float* w_hh, const float* b_hh,
                          const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask, int64_t mask_stride_b, int B,
                          int L, int hidden, float* out, int64_t out_stride_b, int64_t out_stride_t, float* h_raw,
                          float* gates, void* stream) {
    if (!gi || !w_hh || !b_hh || !out || B <= 0 || L <= 0 || !asac_gru_wide_supported(hidden) || !aligned16(gi) ||
        !aligned16(w_hh) || !aligned16(b_hh) || !aligned16(out) || (h0 && !aligned16(h0)) || (gi_stride_b & 3) ||
        (gi_stride_t & 3) || (out_stride_b & 3) || (out_stride_t & 3) || (h0_stride_b & 3) || (h_raw && !aligned16(h_raw)) ||
        (gates && !aligned16(gates)) || (h_raw == nullptr) != (gates == nullptr))
        return bad_arg("asac_gru_wide_forward");
    FwdArgs a;
    a.gi = gi, a.gi_sb = gi_stride_b, a.gi_st = gi_stride_t, a.w_hh = w_hh, a.b_hh = b_hh, a.h0 = h0, a.h0_sb = h0_stride_b;
    a.pad = padding_mask, a.pad_sb = mask_stride_b, a.out = out, a.out_sb = out_stride_b, a.out_st = out_stride_t;
    a.hraw = h_raw, a.gates = gates, a.B = B, a.L = L;
    a.w_hh2 = a.b_hh2 = nullptr, a.twin_B = 0;
    const dim3 grid((unsigned)((B + 15) / 16));
    hipStream_t s = as_stream(stream);
    launch_fwd(hidden, h_raw != nullptr, grid, s, a);
    return finish_launch("asac_gru_wide_forward");
}

int asac_gru_wide_forward_twin(const float* gi, int64_t gi_stride_b, int64_t gi_stride_t, const float* w_hh, const float* b_hh,
                               const float* w_hh_twin, const float* b_hh_twin, const float* h0, int64_t h0_stride_b,
                               const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L, int hidden, float* out,
                               int64_t out_stride_b, int64_t out_stride_t, float* h_raw, float* gates, void* stream) {
    if (!gi || !w_hh || !b_hh || !w_hh_twin || !b_hh_twin || !out || B <= 0 || (B & 15) || L <= 0 ||
        !asac_gru_wide_supported(hidden) || !aligned16(gi) || !aligned16(w_hh) || !aligned16(b_hh) || !aligned16(w_hh_twin) ||
        !aligned16(b_hh_twin) || !aligned16(out) || (h0 && !aligned16(h0)) || (gi_stride_b & 3) || (gi_stride_t & 3) ||
        (out_stride_b & 3) || (out_stride_t & 3) || (h0_stride_b & 3) || (h_raw && !aligned16(h_raw)) ||
        (gates && !aligned16(gates)) || (h_raw == nullptr) != (gates == nullptr))
        return bad_arg("asac_gru_wide_forward_twin");
    FwdArgs a;
    a.gi = gi, a.gi_sb = gi_stride_b, a.gi_st = gi_stride_t, a.w_hh = w_hh, a.b_hh = b_hh, a.h0 = h0, a.h0_sb = h0_stride_b;
    a.pad = padding_mask, a.pad_sb = mask_stride_b, a.out = out, a.out_sb = out_stride_b, a.out_st = out_stride_t;
    a.hraw = h_raw, a.gates = gates, a.B = 2 * B, a.L = L;
    a.w_hh2 = w_hh_twin, a.b_hh2 = b_hh_twin, a.twin_B = B;
    const dim3 grid((unsigned)(2 * B / 16));
    hipStream_t s = as_stream(stream);
    launch_fwd(hidden, h_raw != nullptr, grid, s, a);
    return finish_launch("asac_gru_wide_forward_twin");
}

int asac_gru_wide_backward(const float* grad_out, int64_t go_stride_b, int64_t go_stride_t, const float* w_hh_t,
                           const float* gates, const float* h_raw, const float* h0, int64_t h0_stride_b,
                           const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L, int hidden, float* grad_gi,
                           float* grad_gh, float* grad_h0, void* stream) {
    if (!grad_out || !w_hh_t || !gates || !h_raw || !grad_gi || !grad_gh || B <= 0 || L <= 0 ||
        !asac_gru_wide_supported(hidden) || !aligned16(grad_out) || !aligned16(w_hh_t) || !aligned16(gates) ||
        !aligned16(h_raw) || !aligned16(grad_gi) || !aligned16(grad_gh) || (h0 && !aligned16(h0)) ||
        (grad_h0 && !aligned16(grad_h0)) || (go_stride_b & 3) || (go_stride_t & 3) || (h0_stride_b & 3))
        return bad_arg("asac_gru_wide_backward");
    BwdArgs a;
    a.gout = grad_out, a.go_sb = go_stride_b, a.go_st = go_stride_t, a.w_hh_t = w_hh_t, a.gates = gates, a.hraw = h_raw;
    a.h0 = h0, a.h0_sb = h0_stride_b, a.pad = padding_mask, a.pad_sb = mask_stride_b;
    a.dgi = grad_gi, a.dgh = grad_gh, a.dh0 = grad_h0, a.B = B, a.L = L;
    const dim3 grid((unsigned)((B + 15) / 16));
    hipStream_t s = as_stream(stream);
    if (hidden == 32) ASAC_LAUNCH((k_gruw_bwd<2, 4>), grid, dim3(256), 0, s, a);
    else if (hidden == 64) ASAC_LAUNCH((k_gruw_bwd<4, 4>), grid, dim3(256), 0, s, a);
    else ASAC_LAUNCH((k_gruw_bwd<8, 8>), grid, dim3(512), 0, s, a);
    return finish_launch("asac_gru_wide_backward");
}

}  // extern "C"
