// Fused multi-layer GRU over a padded window, forward and backward, for gfx950.
//
// This is the burn-in of the R2D2-style configurations (reference `algorithm/sac_base.py:1117-1146`
// `get_l_states` -> `nn_models/layers/seq_layers.py:14-114` `GRU`): a stack of GRU cells of hidden
// size H run over every row's [L] window three times per train step.  Through MIOpen that is a
// chain of per-time-step kernels (81 steps x 2 layers, forward and backward, at burn_in 40 +
// n_step 40); here a whole pass is ONE launch.
//
// Semantics = the padding-aware stack of `nn.GRU(batch_first=True)` cells of the plugin layer:
//   r = sig(W_ir x + b_ir + W_hr h + b_hr);  z = sig(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn));  h' = (1 - z) * n + z * h
// Padding follows the plugin layer exactly: `lead` = the first step whose padding_mask byte is 0
// (0 when the whole row is padded); steps before `lead` are skipped (state untouched), steps from
// `lead` on run the cells normally, and the OUTPUT of every padded step is zero — the same values as
// left-aligning the valid block, running the cells and shifting the result back.
//
// The recurrence is a latency chain (L x layers dependent steps of a few hundred cycles), so the
// design removes everything that is not on that chain:
//   * one COMPUTE wave per workgroup, one lane per (row, hidden unit): 64/HP rows per workgroup (HP =
//     H rounded up to a power of two).  The state exchange between the lanes of a row goes through LDS, which a
//     single wave executes in order — no s_barrier and no memory-counter drain in the time loop;
//   * forward keeps the lane's three gate rows of W_ih / W_hh in registers; backward keeps the
//     per-lane weight-gradient accumulators in registers and reads transposed weights from LDS with
//     128-bit reads;
//   * inputs (x, and for backward the saved gate activations) reach the compute wave through LDS, a
//     chunk of time steps at a time, staged by PRODUCER waves of the same workgroup into the other
//     half of a double buffer while the compute wave works on the current chunk (one s_barrier per
//     chunk) — global-memory latency is off the chain entirely; outputs are plain stores nobody
//     waits on;
//   * weight gradients: rows of a wave are summed through LDS in row order, waves by a second kernel
//     in block order (deterministic, no float atomics).
// Sizes: input, hidden <= 16, layers <= 2 (register budget); larger cells use the generic path.
#include "asac_common.h"
#include "asac_sidecar.h"

#include <cmath>

namespace asac {

constexpr int kGruWave = 64;
constexpr int kGruMaxLayers = ASAC_GRU_MAX_LAYERS;
constexpr int kGruMaxDim = ASAC_GRU_MAX_DIM;
constexpr int kFwdChunk = 16;   // time steps of x staged per chunk (forward)
constexpr int kBwdChunk = 8;    // time steps of saved activations staged per chunk (backward)
constexpr int kCopyBatch = 8;   // loads a lane keeps in flight while staging a chunk
constexpr int kFwdThreads = 2 * kGruWave;   // compute wave + 1 producer wave
constexpr int kBwdThreads = 4 * kGruWave;   // compute wave + 3 producer waves

struct GruArgs {
    asac_gru_desc_t d;
    const float* w_ih[kGruMaxLayers];
    const float* w_hh[kGruMaxLayers];
    const float* b_ih[kGruMaxLayers];
    const float* b_hh[kGruMaxLayers];
    const float* x;           // [B, L, I]
    int64_t x_sb, x_st;       // strides in floats
    const float* h0;          // [B, layers, H] (batch stride h0_sb floats) or NULL
    int64_t h0_sb;
    float* out_top;           // [B, L, H] the top layer's (masked) output again, dense, or NULL
    const uint8_t* pad;       // [B, L] (stride pad_sb) or NULL
    int64_t pad_sb;
    int32_t B, L;
    float* hn;                // [B, L, layers, H]  every layer's (masked) output at every step
    float* gates;             // [B, L, layers, 5H] (r, z, n, a_hn, raw state) or NULL (inference)
    // backward
    const float* g_hn;        // [B, L, layers, H] or NULL
    const float* g_top;       // [B, L, H] gradient of out_top or NULL (added to the top layer's)
    float* g_x;               // [B, L, I] or NULL
    float* g_h0;              // [B, layers, H] or NULL
    float* partial;           // [blocks][param_count]
    int64_t param_count;
    // backward from ONE window position (asac_gru_backward_at): the top layer's output gradient at step L_run - 1 is
    // sum_e g_top_m[e][b][:], every other output gradient is zero, so the recursion starts there — the steps behind
    // it would only push zeros around (L_run == L otherwise)
    const float* g_top_m;     // [E][B][H] or NULL
    int32_t g_top_E, L_run;
};

// sigmoid / tanh on the hardware exp2 and reciprocal (each ~1 ulp): absolute error ~1e-7, far inside
// the f32 tolerance of the recurrence and ~4x fewer instructions on the serial chain than expf/tanhf
__device__ __forceinline__ float sigmoidf_(float v) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
}
__device__ __forceinline__ float tanhf_(float v) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177793f * v));
}

// Orders this wave's LDS traffic for the compiler; the hardware already executes one wave's LDS
// instructions in program order, so no waitcnt / s_barrier is needed between lanes of a wave.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__host__ __device__ inline int gru_layer_params(const asac_gru_desc_t& d, int l) {
    const int H = d.hidden, I = l == 0 ? d.input : H;
    return 3 * H * I + 3 * H * H + 6 * H;
}

// first unpadded step of row b (0 without a mask or when the whole row is padded), found by the HP
// lanes of the row together: independent byte loads, then a min over the row's lanes
__device__ __forceinline__ int gru_lead(const GruArgs& a, int b, bool row_ok, int j, int HP) {
    int lead = a.L;
    if (row_ok && a.pad) {
        const uint8_t* m = a.pad + (int64_t)b * a.pad_sb;
        for (int t = j; t < a.L; t += HP)
            if (!m[t]) { lead = t; break; }
    }
    for (int off = HP >> 1; off > 0; off >>= 1) lead = min(lead, __shfl_xor(lead, off, kGruWave));
    return (a.pad && lead < a.L) ? lead : 0;
}

// pins a value: it is computed / loaded unconditionally right here (the compiler may not sink it into the
// branch of a later predicated use, where it would serialise behind that branch's own waits)
__device__ __forceinline__ float pinned(float v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int MAXD>
__device__ __forceinline__ void read_vec(const float* src, float (&v)[MAXD]) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int q = 0; q < MAXD / 4; ++q) {
        const float4 f = s4[q];
        v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
    }
}

// dst[0..count) (LDS) <- src[0..count) (global), by the HP lanes (index j) of one row; every lane of
// the wave runs the same trip count, NB loads are issued before the first LDS write
template <int NB>
__device__ __forceinline__ void row_copy(float* dst, const float* src, int count, bool on, int j, int HP) {
    if (count <= 0) return;
    if ((count & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const int c4 = count >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int base = 0; base < c4; base += HP * NB) {
            float4 v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = s4[min(base + u * HP + j, c4 - 1)];
            // (pinned behind the LAST load: left alone the compiler sinks each load into the branch of its predicated store
            // — a load, a full wait and a store per element, NB dependent round trips where NB loads were meant to be in
            // flight; round 6)
#pragma unroll
            for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (on && base + u * HP + j < c4) d4[base + u * HP + j] = v[u];
        }
    } else {
        for (int base = 0; base < count; base += HP * NB) {
            float v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = src[min(base + u * HP + j, count - 1)];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = pinned(v[u]);
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (on && base + u * HP + j < count) dst[base + u * HP + j] = v[u];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Forward: wave 0 computes, wave 1 stages.
// LDS: state [rows][layers][MAXD] | 2 x (x chunk [rows][kFwdChunk][MAXD]) | 2 x mask chunk
// ------------------------------------------------------------------------------------------------
struct GruFwdPlan { int state, xc, mc, total; };
__host__ __device__ inline GruFwdPlan gru_fwd_plan(int rows, int maxd) {
    GruFwdPlan p;
    int off = 0;
    p.state = off; off += rows * kGruMaxLayers * maxd;
    p.xc = off; off += 2 * rows * kFwdChunk * maxd;   // double-buffered
    p.mc = off; off += 2 * rows * kFwdChunk;
    p.total = off;
    return p;
}

// broadcast of lane K of every aligned group of four lanes (DPP quad_perm: a full-rate VALU move)
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xF, 0xF, false));
}

// Forward lane mapping: FOUR lanes per (row, hidden unit), one per pre-activation
//   part 0: a_r = W_ir x + W_hr h + b_ir + b_hr     part 1: a_z likewise
//   part 2: a_n = W_in x + b_in                     part 3: a_hn = W_hn h + b_hn
// so a lane carries 16 of the step's 48 multiply-adds and one of its two sigmoid evaluations; the four meet
// through quad broadcasts (r, z, a_n, a_hn), every lane then forms n and h' redundantly, and the stores of
// the step are shared out (part q stores gate q).  64 / (4 HP) rows per compute wave.
// blockIdx.y picks the parameter set: 0 = the caller's, 1 = the twin (the target network's copy run over the same
// window for inference, asac_gru_forward_twin)
struct GruFwdJobs { GruArgs job[2]; };

template <int MAXD, bool TWO>
__global__ __launch_bounds__(kFwdThreads) void k_gru_fwd(const GruFwdJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GruArgs& a = jobs.job[blockIdx.y];
    const int H = a.d.hidden, HP = a.d.hidden_pow2, layers = a.d.layers, I0 = a.d.input;
    const int GP = 4 * HP;                         // lanes per row
    const int rows = kGruWave / GP;
    const GruFwdPlan p = gru_fwd_plan(rows, MAXD);
    const int lane = threadIdx.x & (kGruWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kGruWave);
    const int r = lane / GP, u = lane % GP;
    const int j = u >> 2, q = u & 3;
    const int b = blockIdx.x * rows + r;
    const bool row_ok = b < a.B;
    const bool live = row_ok && j < H;
    const int n_chunks = (a.L + kFwdChunk - 1) / kFwdChunk;

    for (int i = threadIdx.x; i < p.total; i += kFwdThreads) lds[i] = 0.f;
    __syncthreads();

    // x[t0 .. t0+n) and the mask bytes of this workgroup's rows -> half `buf`: the GP lanes of a row
    // copy that row, kCopyBatch loads in flight per lane before the first LDS write
    if (wave == 1) {               // the producer wave: one chunk ahead of the compute wave
        for (int c = 0; c <= n_chunks; ++c) {
            if (c < n_chunks) {
                const int t0 = c * kFwdChunk, n = min(kFwdChunk, a.L - t0), buf = c & 1;
                const int count = row_ok ? n * I0 : 0;
                const float* xs = a.x + (int64_t)(row_ok ? b : 0) * a.x_sb + (int64_t)t0 * a.x_st;
                float* xd = lds + p.xc + (buf * rows + r) * kFwdChunk * MAXD;
                for (int base = 0; base < n * I0; base += GP * kCopyBatch) {
                    float v[kCopyBatch];
#pragma unroll
                    for (int w = 0; w < kCopyBatch; ++w) {
                        const int e = min(base + w * GP + u, n * I0 - 1), tt = e / I0, k = e - tt * I0;
                        v[w] = xs[(int64_t)tt * a.x_st + k];
                    }
#pragma unroll
                    for (int w = 0; w < kCopyBatch; ++w) {
                        const int e = base + w * GP + u, tt = e / I0, k = e - tt * I0;
                        if (e < count) xd[tt * MAXD + k] = v[w];
                    }
                }
                if (a.pad) {
                    float mv[kFwdChunk];
#pragma unroll
                    for (int w = 0; w < kFwdChunk; ++w)
                        mv[w] = a.pad[(int64_t)(row_ok ? b : 0) * a.pad_sb + min(t0 + w, a.L - 1)] ? 1.f : 0.f;
                    if (u == 0 && row_ok) {
#pragma unroll
                        for (int w = 0; w < kFwdChunk; ++w) lds[p.mc + (buf * rows + r) * kFwdChunk + w] = mv[w];
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    // this lane's row of W_ih and / or W_hh (zero where the part has no such term) and its bias, per layer
    float wx[kGruMaxLayers][MAXD], wh[kGruMaxLayers][MAXD], bias[kGruMaxLayers];
#pragma unroll
    for (int l = 0; l < kGruMaxLayers; ++l) {
        const bool on = l < layers && j < H;
        const int ls = l < layers ? l : 0, js = j < H ? j : H - 1;    // clamped: every load is in range,
        const int I = ls == 0 ? I0 : H;                                // out-of-range lanes select zero
        const int g = q < 2 ? q : 2;                                   // gate row block: r, z, n, n
        const bool use_x = q != 3, use_h = q != 2;
#pragma unroll
        for (int k = 0; k < MAXD; ++k) {
            const float vi = a.w_ih[ls][(g * H + js) * I + min(k, I - 1)];
            const float vh = a.w_hh[ls][(g * H + js) * H + min(k, H - 1)];
            wx[l][k] = (on && use_x && k < I) ? vi : 0.f;
            wh[l][k] = (on && use_h && k < H) ? vh : 0.f;
        }
        const float vbi = a.b_ih[ls][g * H + js], vbh = a.b_hh[ls][g * H + js];
        bias[l] = on ? ((use_x ? vbi : 0.f) + (use_h ? vbh : 0.f)) : 0.f;
    }
    // first unpadded step of the row (the GP lanes of the row look at the mask together)
    int lead = a.L;
    if (row_ok && a.pad) {
        const uint8_t* m = a.pad + (int64_t)b * a.pad_sb;
        for (int t = u; t < a.L; t += GP)
            if (!m[t]) { lead = t; break; }
    }
    for (int off = GP >> 1; off > 0; off >>= 1) lead = min(lead, __shfl_xor(lead, off, kGruWave));
    lead = (a.pad && lead < a.L) ? lead : 0;
    wave_sync();
    float* state = lds + p.state + r * kGruMaxLayers * MAXD;
    if (live && q == 0 && a.h0)
        for (int l = 0; l < layers; ++l) state[l * MAXD + j] = a.h0[(int64_t)b * a.h0_sb + l * H + j];

    // Stores of a tick, shared out over the four lanes of a unit so that every lane issues the same one or two
    // store instructions per layer with loop-invariant predicates and pointers that only advance:
    //   store A (training only): part q writes gate q (r, z, n, a_hn) of its unit into the saved activations
    //   store B: part 0 writes the layer's (masked) output, part 1 the unmasked state beside the gates,
    //            part 2 the separate top-layer output
    const int64_t bq = row_ok ? b : 0, jq = j < H ? j : 0;
    const bool train = a.gates != nullptr;
    const int hn_step = layers * H, gate_step = layers * 5 * H;
    float* pa = train ? a.gates + (bq * a.L) * gate_step + q * H + jq : nullptr;             // + l * 5H per layer
    // parts without a store B of their own (part 3; part 1 at inference; part 2 without a separate top output or,
    // for layer 0 of a two-layer stack, always) repeat part 0's store: same address, same value, no predicate
    int kind = q;                                   // 0: layer output, 1: raw state, 2: top output
    if (q == 3 || (q == 1 && !train) || (q == 2 && !a.out_top)) kind = 0;
    float* pb = kind == 0 ? a.hn + (bq * a.L) * hn_step + jq
              : kind == 1 ? a.gates + (bq * a.L) * gate_step + 4 * H + jq
                          : a.out_top + (bq * a.L) * H + jq;
    float* pb0 = (kind == 2 && layers == 2) ? a.hn + (bq * a.L) * hn_step + jq : pb;    // layer 0's store B
    const int pb_layer = kind == 0 ? H : (kind == 1 ? 5 * H : 0), pb_step = kind == 0 ? hn_step : (kind == 1 ? gate_step : H);
    const int pb0_step = (kind == 2 && layers == 2) ? hn_step : pb_step;
    const bool raw_b = kind == 1, raw_b0 = raw_b;
    constexpr bool two = TWO;               // layers == 2

    // One tick = layer 0 at step k TOGETHER WITH layer 1 at step k-1: the two steps are independent (layer 1
    // consumes the state layer 0 had BEFORE this tick), so their dependent chains — LDS round trip, multiply-adds,
    // two transcendental chains, state write — overlap inside the one instruction stream.
    auto tick = [&](int k, const float* xrow, bool padded0, bool padded1) {
        const bool on0 = k < a.L, on1 = two && k >= 1;
        const bool skip0 = k < lead, skip1 = k - 1 < lead;
        float xin[MAXD], hs0[MAXD], hs1[MAXD];
        read_vec<MAXD>(xrow, xin);
        read_vec<MAXD>(state, hs0);                       // also layer 1's input: layer 0's output of step k-1
        if (two) read_vec<MAXD>(state + MAXD, hs1);
        const float h_old0 = state[j], h_old1 = two ? state[MAXD + j] : 0.f;
        // packed multiply-adds (v_pk_fma_f32): even / odd columns in the two halves, four independent sums
        f32x2 s0x = {bias[0], 0.f}, s0h = {0.f, 0.f}, s1x = {bias[1], 0.f}, s1h = {0.f, 0.f};
#pragma unroll
        for (int w = 0; w < MAXD; w += 2) {
            const f32x2 xv = {xin[w], xin[w + 1]}, h0v = {hs0[w], hs0[w + 1]};
            s0x = __builtin_elementwise_fma((f32x2){wx[0][w], wx[0][w + 1]}, xv, s0x);
            s0h = __builtin_elementwise_fma((f32x2){wh[0][w], wh[0][w + 1]}, h0v, s0h);
            if (two) {
                const f32x2 h1v = {hs1[w], hs1[w + 1]};
                s1x = __builtin_elementwise_fma((f32x2){wx[1][w], wx[1][w + 1]}, h0v, s1x);
                s1h = __builtin_elementwise_fma((f32x2){wh[1][w], wh[1][w + 1]}, h1v, s1h);
            }
        }
        const float acc0 = (s0x.x + s0x.y) + (s0h.x + s0h.y), acc1 = (s1x.x + s1x.y) + (s1h.x + s1h.y);
        const float sg0 = sigmoidf_(acc0), sg1 = sigmoidf_(acc1);
        const float r0 = quad_bcast<0>(sg0), z0 = quad_bcast<1>(sg0), an0 = quad_bcast<2>(acc0), ahn0 = quad_bcast<3>(acc0);
        const float r1 = quad_bcast<0>(sg1), z1 = quad_bcast<1>(sg1), an1 = quad_bcast<2>(acc1), ahn1 = quad_bcast<3>(acc1);
        const float n0 = tanhf_(fmaf(r0, ahn0, an0)), n1 = tanhf_(fmaf(r1, ahn1, an1));
        const float h_new0 = skip0 ? h_old0 : (1.f - z0) * n0 + z0 * h_old0;
        const float h_new1 = skip1 ? h_old1 : (1.f - z1) * n1 + z1 * h_old1;
        // part q's gate value, and what its store B carries (parts 0 / 2: the masked output, part 1: the raw state)
        const float ga0 = q == 0 ? sg0 : (q == 1 ? sg0 : (q == 2 ? n0 : acc0));
        const float ga1 = q == 0 ? sg1 : (q == 1 ? sg1 : (q == 2 ? n1 : acc1));
        const float vb0 = (!raw_b0 && padded0) ? 0.f : h_new0, vb1 = (!raw_b && padded1) ? 0.f : h_new1;
        const float s0 = pinned(h_new0), s1 = pinned(h_new1);     // both layers' chains finish before the stores
        wave_sync();                      // every lane of the row has read the old states
        if (live) {                       // the four lanes of a unit write the same state value: no part predicate
            if (on0) {
                state[j] = s0;
                if (train) pa[0] = ga0;
                pb0[0] = vb0;
            }
            if (on1) {                    // layer 1's outputs belong to the previous time step
                state[MAXD + j] = s1;
                if (train) pa[5 * H - gate_step] = ga1;
                pb[pb_layer - pb_step] = vb1;
            }
        }
        if (train) pa += gate_step;
        pb += pb_step;
        pb0 += pb0_step;
        wave_sync();
    };

    __syncthreads();               // chunk 0 is staged
    bool padded_prev = false;
    for (int c = 0; c < n_chunks; ++c) {
        const int t0 = c * kFwdChunk, n = min(kFwdChunk, a.L - t0);
        const float* xc = lds + p.xc + ((c & 1) * rows + r) * kFwdChunk * MAXD;
        const float* mc = lds + p.mc + ((c & 1) * rows + r) * kFwdChunk;
        for (int tt = 0; tt < n; ++tt) {
            const bool padded = mc[tt] != 0.f;
            tick(t0 + tt, xc + tt * MAXD, padded, padded_prev);
            padded_prev = padded;
        }
        __syncthreads();           // chunk c+1 is staged; half c&1 may be overwritten
    }
    if (two) tick(a.L, lds + p.xc, false, padded_prev);      // layer 1's last step (the x row read is unused)
}

// ------------------------------------------------------------------------------------------------
// Backward through time: wave 0 computes; waves 1..3 stage the saved activations, the output
// gradients and the inputs + mask of the next chunk.
// LDS: transposed weights | activation chunk | exchange vectors | reduction slab
//   tih [layers][MAXD(k)][3][MAXD(jj)]   tih[l][k][g][jj] = W_ih[l][g*H+jj][k]
//   thh [layers][MAXD(j)][3][MAXD(jj)]   thh[l][j][g][jj] = W_hh[l][g*H+jj][j]
//   G   2 x [rows][kBwdChunk+1][layers*5H]   saved activations of steps t0-1 .. t0+n-1
//   GH  2 x [rows][kBwdChunk][layers*H]      incoming gradient of the per-step outputs
//   X   2 x [rows][kBwdChunk][I]  M 2 x [rows][kBwdChunk]
//   hv/xv [rows][MAXD], dg [rows][4][MAXD]   this row's h_{t-1}, layer input and gate deltas
// ------------------------------------------------------------------------------------------------
struct GruBwdPlan { int tih, thh, G, GH, X, M, gsz, hsz, xsz, msz, hv, xv, dg, slab, total; };
__host__ __device__ inline GruBwdPlan gru_bwd_plan(const asac_gru_desc_t& d, int rows, int maxd) {
    GruBwdPlan p;
    int off = 0, max_layer = 0;
    for (int l = 0; l < d.layers; ++l) {
        const int n = gru_layer_params(d, l);
        max_layer = n > max_layer ? n : max_layer;
    }
    auto pad4 = [](int v) { return (v + 3) & ~3; };
    p.tih = off; off += kGruMaxLayers * maxd * 3 * maxd;
    p.thh = off; off += kGruMaxLayers * maxd * 3 * maxd;
    p.hv = off; off += rows * 2 * maxd;         // exchange vectors: one set per layer of a tick
    p.xv = off; off += rows * 2 * maxd;
    p.dg = off; off += rows * 2 * 4 * maxd;
    p.gsz = pad4(rows * (kBwdChunk + 1) * d.layers * 5 * d.hidden);
    p.hsz = pad4(rows * kBwdChunk * d.layers * d.hidden);
    p.xsz = pad4(rows * kBwdChunk * d.input);
    p.msz = pad4(rows * kBwdChunk);
    p.G = off; off += 2 * p.gsz;       // every staged stream is double-buffered
    p.GH = off; off += 2 * p.hsz;
    p.X = off; off += 2 * p.xsz;
    p.M = off; off += 2 * p.msz;
    p.slab = off; off += pad4(max_layer);
    p.total = off;
    return p;
}

// sum over the four lanes of a unit (two DPP quad permutes), result in every lane
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));   // [2,3,0,1]
    return v;
}

// Backward lane mapping: FOUR lanes per (row, hidden unit), like the forward.  Every lane forms the unit's gate
// deltas (cheap), then part q owns one slice of the heavy work:
//   weight gradients   part 0: the r rows of W_ih and W_hh, part 1: the z rows, part 2: the n row of W_ih,
//                      part 3: the n row of W_hh   (16 accumulating multiply-adds per lane instead of 48)
//   back-propagation   parts 0..2: gate g = q's term of  dh_{t-1}[j] = sum_jj W_h*[jj][j] delta_*[jj]  and of
//                      dx[k] = sum_jj W_i*[jj][k] delta_*[jj];  the three terms meet in a quad sum
template <int MAXD, bool TWO>
__global__ __launch_bounds__(kBwdThreads) void k_gru_bwd(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.d.hidden, HP = a.d.hidden_pow2, layers = a.d.layers, I0 = a.d.input;
    const int GP = 4 * HP;
    const int rows = kGruWave / GP;
    const GruBwdPlan p = gru_bwd_plan(a.d, rows, MAXD);
    const int lane = threadIdx.x & (kGruWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kGruWave);
    const int r = lane / GP, u = lane % GP;
    const int j = u >> 2, q = u & 3;
    const int b = blockIdx.x * rows + r;
    const bool row_ok = b < a.B;
    const bool live = row_ok && j < H;
    const int GS = layers * 5 * H, HS = layers * H;
    const int Lr = a.L_run;                 // steps the recursion covers (strides stay those of the L-step window)
    const int n_chunks = (Lr + kBwdChunk - 1) / kBwdChunk;

    for (int i = threadIdx.x; i < p.total; i += kBwdThreads) lds[i] = 0.f;
    __syncthreads();
    for (int l = 0; l < layers; ++l) {
        const int I = l == 0 ? I0 : H;
        for (int i = threadIdx.x; i < 3 * H * I; i += kBwdThreads) {      // W_ih[l][g*H+jj][k]
            const int row = i / I, k = i - row * I, g = row / H, jj = row - g * H;
            lds[p.tih + ((l * MAXD + k) * 3 + g) * MAXD + jj] = a.w_ih[l][i];
        }
        for (int i = threadIdx.x; i < 3 * H * H; i += kBwdThreads) {      // W_hh[l][g*H+jj][k]
            const int row = i / H, k = i - row * H, g = row / H, jj = row - g * H;
            lds[p.thh + ((l * MAXD + k) * 3 + g) * MAXD + jj] = a.w_hh[l][i];
        }
    }

    // Producer waves.  Chunk c covers steps t0 .. t0+n-1; each stream is contiguous per row and copied
    // by the row's GP lanes into half c&1 of its double buffer:
    //   wave 1: saved activations of steps t0-1 .. t0+n-1 (slot 0 = step t0-1)
    //   wave 2: gradients of the per-step outputs          wave 3: inputs and mask
    if (wave != 0) {               // producers run one chunk ahead of the compute wave
        for (int c = n_chunks - 1; c >= -1; --c) {
            if (c >= 0) {
                const int t0 = c * kBwdChunk, n = min(kBwdChunk, Lr - t0), buf = c & 1;
                const int64_t bs = row_ok ? b : 0;
                if (wave == 1) {
                    const int skip0 = t0 == 0 ? GS : 0;     // there is no step -1
                    row_copy<kCopyBatch>(lds + p.G + buf * p.gsz + r * (kBwdChunk + 1) * GS + skip0,
                                         a.gates + (bs * a.L + t0 - 1) * GS + skip0, (n + 1) * GS - skip0, row_ok, u, GP);
                } else if (wave == 2) {
                    float* gh = lds + p.GH + buf * p.hsz + r * kBwdChunk * HS;
                    if (a.g_hn) {
                        row_copy<kCopyBatch>(gh, a.g_hn + (bs * a.L + t0) * HS, n * HS, row_ok, u, GP);
                    } else {
                        for (int e = u; e < n * HS; e += GP) gh[e] = 0.f;
                    }
                    if (a.g_top) {           // the separately returned top-layer output's gradient
                        wave_sync();
                        const float* gt = a.g_top + (bs * a.L + t0) * H;
                        for (int e = u; e < n * H; e += GP) {
                            const int tt = e / H, k = e - tt * H;
                            if (row_ok) gh[tt * HS + (layers - 1) * H + k] += gt[e];
                        }
                    }
                    if (a.g_top_m && t0 + n == Lr) {        // the one position with a gradient: members summed in order
                        wave_sync();
                        for (int k = u; k < H; k += GP) {
                            float sum = 0.f;
                            for (int e = 0; e < a.g_top_E; ++e) {
                                const float v = a.g_top_m[((int64_t)e * a.B + bs) * H + k];
                                sum = e == 0 ? v : sum + v;
                            }
                            if (row_ok) gh[(n - 1) * HS + (layers - 1) * H + k] += sum;
                        }
                    }
                } else {
                    const float* xs = a.x + bs * a.x_sb + (int64_t)t0 * a.x_st;
                    float* xd = lds + p.X + buf * p.xsz + r * kBwdChunk * I0;
                    if (a.x_st == I0) {
                        row_copy<kCopyBatch>(xd, xs, n * I0, row_ok, u, GP);
                    } else {
                        for (int e = u; e < n * I0; e += GP) {
                            const int tt = e / I0, k = e - tt * I0;
                            if (row_ok) xd[tt * I0 + k] = xs[(int64_t)tt * a.x_st + k];
                        }
                    }
                    if (a.pad) {
                        float mv[kBwdChunk];
#pragma unroll
                        for (int w = 0; w < kBwdChunk; ++w) mv[w] = a.pad[bs * a.pad_sb + min(t0 + w, a.L - 1)] ? 1.f : 0.f;
                        if (u == 0 && row_ok) {
#pragma unroll
                            for (int w = 0; w < kBwdChunk; ++w) lds[p.M + buf * p.msz + r * kBwdChunk + w] = mv[w];
                        }
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    // per-lane accumulators: this part's row of W_ih (gx) and / or W_hh (gh) gradients and their biases
    float gx[kGruMaxLayers][MAXD], gh[kGruMaxLayers][MAXD], gbx[kGruMaxLayers], gbh[kGruMaxLayers];
    float dh[kGruMaxLayers], h0v[kGruMaxLayers];
#pragma unroll
    for (int l = 0; l < kGruMaxLayers; ++l) {
        dh[l] = gbx[l] = gbh[l] = 0.f;
        h0v[l] = (live && a.h0 && l < layers) ? a.h0[(int64_t)b * a.h0_sb + l * H + j] : 0.f;
#pragma unroll
        for (int k = 0; k < MAXD; ++k) gx[l][k] = gh[l][k] = 0.f;
    }
    int lead = a.L;                // first unpadded step of the row (the GP lanes look at the mask together)
    if (row_ok && a.pad) {
        const uint8_t* m = a.pad + (int64_t)b * a.pad_sb;
        for (int t = u; t < a.L; t += GP)
            if (!m[t]) { lead = t; break; }
    }
    for (int off = GP >> 1; off > 0; off >>= 1) lead = min(lead, __shfl_xor(lead, off, kGruWave));
    lead = (a.pad && lead < a.L) ? lead : 0;
    // exchange vectors of the tick's two layer steps: set 0 = the top layer's, set 1 = layer 0's under a second layer
    float* hvA = lds + p.hv + (r * 2) * MAXD;
    float* xvA = lds + p.xv + (r * 2) * MAXD;
    float* dgA = lds + p.dg + (r * 2) * 4 * MAXD;
    float* hvB = hvA + MAXD;
    float* xvB = xvA + MAXD;
    float* dgB = dgA + 4 * MAXD;
    const int g = q < 3 ? q : 2;            // the gate whose back-propagation term this part carries (q == 3: none)
    const int gh_row = g == 2 ? 3 : g;      // ... and the delta vector its hidden-side term multiplies
    const float dot_on = q < 3 ? 1.f : 0.f;
    const int jq = min(j, H - 1);           // clamped unit index: every lane's LDS reads are in range
    constexpr int XP = MAXD / 4;            // input elements a lane may have to publish (GP >= 4)

    // One tick = the top layer at step k TOGETHER WITH (two layers) layer 0 at step k+1: layer 0 needs the
    // input gradient the top layer produced for ITS step one tick earlier (`below`), so the two are independent
    // and their LDS round trips and multiply-add chains overlap in the one instruction stream.  Layer 0's saved
    // values are fetched into registers a tick ahead (while its step's chunk is certainly still staged).
    float below = 0.f;                       // d(top layer's input) of the previous tick = d(layer 0 output) at k+1
    float b_rg = 0.f, b_zg = 0.f, b_ng = 0.f, b_ahn = 0.f, b_hp = 0.f, b_ghn = 0.f, b_x[XP];
#pragma unroll
    for (int w = 0; w < XP; ++w) b_x[w] = 0.f;

    __syncthreads();               // the last chunk is staged (and the transposed weights are in place)
    int c = n_chunks - 1;
    for (int k = Lr - 1; k >= (TWO ? -1 : 0); --k) {
        const bool onA = k >= 0, onB = TWO && k + 1 < Lr;
        const int t0 = c * kBwdChunk, tt = onA ? k - t0 : 0;
        const float* Gc = lds + p.G + (c & 1) * p.gsz + r * (kBwdChunk + 1) * GS;
        const float* GHc = lds + p.GH + (c & 1) * p.hsz + r * kBwdChunk * HS;
        const float* Xc = lds + p.X + (c & 1) * p.xsz + r * kBwdChunk * I0;
        const float* Mc = lds + p.M + (c & 1) * p.msz + r * kBwdChunk;
        const float* Gt = Gc + (tt + 1) * GS;                                 // step k
        const float* Gp = Gt - GS;                                            // step k-1
        constexpr int lt = TWO ? 1 : 0;                                       // the top layer
        const bool skipA = k < lead, skipB = k + 1 < lead;

        // Every LDS read below is unconditional on a clamped index and predicates are applied with selects
        // afterwards: the reads of a phase then issue back to back and the wave waits once per phase instead of
        // once per divergent region (one wave per SIMD: every wait is a full LDS round trip).
        // ---- top layer, step k: saved values, deltas, publish ------------------------------------
        const float* gpA = Gt + lt * 5 * H + jq;
        const float rg = gpA[0], zg = gpA[H], ng = gpA[2 * H], a_hn = gpA[3 * H];
        float hp_saved = Gp[lt * 5 * H + 4 * H + jq], ghn_saved = GHc[tt * HS + lt * H + jq];
        float xa[XP];
#pragma unroll
        for (int w = 0; w < XP; ++w) {
            const int kk = u + w * GP;
            xa[w] = TWO ? Gt[4 * H + min(kk, H - 1)]                         // layer 0's state of step k
                        : Xc[tt * I0 + min(kk, I0 - 1)];
        }
        const bool paddedA = Mc[tt] != 0.f;
        hp_saved = pinned(hp_saved);
        ghn_saved = pinned(ghn_saved);
#pragma unroll
        for (int w = 0; w < XP; ++w) xa[w] = pinned(xa[w]);
        const bool actA = live && onA && !skipA;
        const float hp = (live && onA) ? (k - 1 >= lead ? hp_saved : h0v[lt]) : 0.f;
        const float ghn = paddedA ? 0.f : ghn_saved;
        float dhtA = dh[lt] + ghn;
        float dnA = dhtA * (1.f - zg) * (1.f - ng * ng);
        float dzA = dhtA * (hp - ng) * zg * (1.f - zg);
        float dhnA = dnA * rg;
        float drA = dnA * a_hn * rg * (1.f - rg);
        dhtA = actA ? dhtA : 0.f; dnA = actA ? dnA : 0.f; dzA = actA ? dzA : 0.f;
        dhnA = actA ? dhnA : 0.f; drA = actA ? drA : 0.f;
        hvA[j] = hp;                       // the four lanes of a unit write the same value
#pragma unroll
        for (int w = 0; w < XP; ++w) {
            const int kk = u + w * GP;
            if (kk < MAXD) xvA[kk] = (row_ok && onA && kk < (TWO ? H : I0)) ? xa[w] : 0.f;
        }
        // part q publishes delta q (r, z, n, hn) and accumulates with (dx, dh) = q0 (d_r, d_r), q1 (d_z, d_z),
        // q2 (d_n, 0), q3 (0, d_hn)
        dgA[q * MAXD + j] = q == 0 ? drA : (q == 1 ? dzA : (q == 2 ? dnA : dhnA));
        const float dxqA = q == 0 ? drA : (q == 1 ? dzA : (q == 2 ? dnA : 0.f));
        const float dhqA = q == 0 ? drA : (q == 1 ? dzA : (q == 3 ? dhnA : 0.f));

        // ---- layer 0, step k+1 (operands fetched last tick) --------------------------------------
        float dhtB = 0.f, drB = 0.f, dzB = 0.f, dnB = 0.f, dhnB = 0.f;
        const bool actB = TWO && live && onB && !skipB;
        if (TWO) {
            dhtB = dh[0] + b_ghn + below;
            dnB = dhtB * (1.f - b_zg) * (1.f - b_ng * b_ng);
            dzB = dhtB * (b_hp - b_ng) * b_zg * (1.f - b_zg);
            dhnB = dnB * b_rg;
            drB = dnB * b_ahn * b_rg * (1.f - b_rg);
            dhtB = actB ? dhtB : 0.f; dnB = actB ? dnB : 0.f; dzB = actB ? dzB : 0.f;
            dhnB = actB ? dhnB : 0.f; drB = actB ? drB : 0.f;
            hvB[j] = b_hp;
#pragma unroll
            for (int w = 0; w < XP; ++w) {
                const int kk = u + w * GP;
                if (kk < MAXD) xvB[kk] = b_x[w];
            }
            dgB[q * MAXD + j] = q == 0 ? drB : (q == 1 ? dzB : (q == 2 ? dnB : dhnB));
        }
        const float dxqB = q == 0 ? drB : (q == 1 ? dzB : (q == 2 ? dnB : 0.f));
        const float dhqB = q == 0 ? drB : (q == 1 ? dzB : (q == 3 ? dhnB : 0.f));
        const float zgB = b_zg;
        wave_sync();

        // ---- next tick's layer 0 operands: its step k, read from this (still staged) chunk; the reads go out with
        //      the vector reads below and are consumed at the end of the tick -------------------------------
        float n_rg = 0.f, n_zg = 0.f, n_ng = 0.f, n_ahn = 0.f, n_hp = 0.f, n_ghn = 0.f, n_x[XP];
        if (TWO) {
            n_rg = Gt[jq]; n_zg = Gt[H + jq]; n_ng = Gt[2 * H + jq]; n_ahn = Gt[3 * H + jq];
            n_hp = Gp[4 * H + jq]; n_ghn = GHc[tt * HS + jq];
#pragma unroll
            for (int w = 0; w < XP; ++w) n_x[w] = Xc[tt * I0 + min(u + w * GP, I0 - 1)];
        }

        // ---- accumulate the weight gradients, back-propagate into h_{t-1} and the layer input ------------
        float hvecA[MAXD], xvecA[MAXD], vxA[MAXD], vhA[MAXD], whA[MAXD];
        read_vec<MAXD>(hvA, hvecA);
        read_vec<MAXD>(xvA, xvecA);
        read_vec<MAXD>(dgA + g * MAXD, vxA);                         // delta_r / delta_z / delta_n
        read_vec<MAXD>(dgA + gh_row * MAXD, vhA);                    // delta_r / delta_z / delta_hn
        read_vec<MAXD>(lds + p.thh + ((lt * MAXD + j) * 3 + g) * MAXD, whA);
        float wx0[MAXD];           // layer 0's W_ih column block of input j (j < MAXD: in range, zero beyond I)
        read_vec<MAXD>(lds + p.tih + (j * 3 + g) * MAXD, wx0);
        float hvecB[MAXD], xvecB[MAXD], vxB[MAXD], vhB[MAXD], whB[MAXD], wxA[MAXD];
        if (TWO) {
            read_vec<MAXD>(hvB, hvecB);
            read_vec<MAXD>(xvB, xvecB);
            read_vec<MAXD>(dgB + g * MAXD, vxB);
            read_vec<MAXD>(dgB + gh_row * MAXD, vhB);
            read_vec<MAXD>(lds + p.thh + (j * 3 + g) * MAXD, whB);
            read_vec<MAXD>(lds + p.tih + ((MAXD + j) * 3 + g) * MAXD, wxA);
        }
#pragma unroll
        for (int kk = 0; kk < MAXD; ++kk) {
            gx[lt][kk] = fmaf(dxqA, xvecA[kk], gx[lt][kk]);
            gh[lt][kk] = fmaf(dhqA, hvecA[kk], gh[lt][kk]);
        }
        gbx[lt] += dxqA;
        gbh[lt] += dhqA;
        {   // d h_{t-1}[j] = dht * z + sum over gates of sum_jj W_h(gate)[jj][j] * delta[jj]
            float acc = 0.f;
#pragma unroll
            for (int jj = 0; jj < MAXD; ++jj) acc = fmaf(whA[jj], vhA[jj], acc);
            acc = quad_sum(acc * dot_on);
            dh[lt] = actA ? dhtA * zg + acc : dh[lt];
        }
        // d input[kk] = sum over gates of sum_jj W_i(gate)[jj][kk] * delta[jj]
        if (TWO) {                 // the input is layer 0's output: unit j's gradient, consumed next tick
            float acc = 0.f;
#pragma unroll
            for (int jj = 0; jj < MAXD; ++jj) acc = fmaf(wxA[jj], vxA[jj], acc);
            below = quad_sum(acc * dot_on);
#pragma unroll
            for (int kk = 0; kk < MAXD; ++kk) {
                gx[0][kk] = fmaf(dxqB, xvecB[kk], gx[0][kk]);
                gh[0][kk] = fmaf(dhqB, hvecB[kk], gh[0][kk]);
            }
            gbx[0] += dxqB;
            gbh[0] += dhqB;
            acc = 0.f;
#pragma unroll
            for (int jj = 0; jj < MAXD; ++jj) acc = fmaf(whB[jj], vhB[jj], acc);
            acc = quad_sum(acc * dot_on);
            dh[0] = actB ? dhtB * zgB + acc : dh[0];
        }
        {   // the network input's gradient (layer 0): step k for a single layer, step k+1 under a second layer
            const float (&vx0)[MAXD] = TWO ? vxB : vxA;
            const bool on0 = TWO ? onB : onA;
            const int64_t t_out = TWO ? k + 1 : k;
            float acc = 0.f;
#pragma unroll
            for (int jj = 0; jj < MAXD; ++jj) acc = fmaf(wx0[jj], vx0[jj], acc);
            acc = quad_sum(acc * dot_on);
            if (a.g_x && row_ok && on0 && q == 0 && j < I0) a.g_x[((int64_t)b * a.L + t_out) * I0 + j] = acc;
            for (int base = HP; base < I0; base += HP) {      // inputs wider than the lane group: uniform trips
                const int kk = base + j;
                float w0[MAXD];
                read_vec<MAXD>(lds + p.tih + (min(kk, MAXD - 1) * 3 + g) * MAXD, w0);
                acc = 0.f;
#pragma unroll
                for (int jj = 0; jj < MAXD; ++jj) acc = fmaf(w0[jj], vx0[jj], acc);
                acc = quad_sum(acc * dot_on);
                if (a.g_x && row_ok && on0 && q == 0 && kk < I0) a.g_x[((int64_t)b * a.L + t_out) * I0 + kk] = acc;
            }
        }
        if (TWO) {                 // hand the fetched operands to the next tick
            const bool on = live && onA;
            n_hp = pinned(n_hp);
            n_ghn = pinned(n_ghn);
            b_rg = on ? n_rg : 0.f; b_zg = on ? n_zg : 0.f; b_ng = on ? n_ng : 0.f; b_ahn = on ? n_ahn : 0.f;
            b_hp = on ? (k - 1 >= lead ? n_hp : h0v[0]) : 0.f;
            b_ghn = (on && !paddedA) ? n_ghn : 0.f;
#pragma unroll
            for (int w = 0; w < XP; ++w) {
                const float v = pinned(n_x[w]);
                b_x[w] = (row_ok && onA && u + w * GP < I0) ? v : 0.f;
            }
        }
        wave_sync();
        if (onA && tt == 0) {      // chunk c-1 is staged; half c&1 may be overwritten
            __syncthreads();
            --c;
        }
    }
    if (live && q == 0 && a.g_h0) {
#pragma unroll
        for (int l = 0; l < kGruMaxLayers; ++l)
            if (l < layers) a.g_h0[((int64_t)b * layers + l) * H + j] = dh[l];
    }

    // ---- sum the rows of this block in row order through an LDS slab, layer by layer -----------------
    float* slab = lds + p.slab;
    float* part = a.partial + (int64_t)blockIdx.x * a.param_count;
    int64_t layer_base = 0;
#pragma unroll
    for (int l = 0; l < kGruMaxLayers; ++l) {
        if (l >= layers) continue;
        const int I = l == 0 ? I0 : H;
        const int n = gru_layer_params(a.d, l);
        const int o_whh = 3 * H * I, o_bih = o_whh + 3 * H * H, o_bhh = o_bih + 3 * H;
        wave_sync();
        for (int i = lane; i < n; i += kGruWave) slab[i] = 0.f;
        wave_sync();
        for (int rr = 0; rr < rows; ++rr) {
            if (r == rr && live) {
                if (q < 3) {           // W_ih / b_ih row of gate q
#pragma unroll
                    for (int k = 0; k < MAXD; ++k)
                        if (k < I) slab[(q * H + j) * I + k] += gx[l][k];
                    slab[o_bih + q * H + j] += gbx[l];
                }
                if (q != 2) {          // W_hh / b_hh row of gate r, z, n for parts 0, 1, 3
                    const int gg = q == 3 ? 2 : q;
#pragma unroll
                    for (int k = 0; k < MAXD; ++k)
                        if (k < H) slab[o_whh + (gg * H + j) * H + k] += gh[l][k];
                    slab[o_bhh + gg * H + j] += gbh[l];
                }
            }
            wave_sync();
        }
        for (int i = lane; i < n; i += kGruWave) part[layer_base + i] = slab[i];
        layer_base += n;
    }
}

// sum_blocks partial[block][i] in fixed order -> the packed gradient buffer, or straight into (onto) the
// parameters' own gradient tensors
struct GruGradDst {
    float* packed;                         // [n] or NULL
    float* dst[4 * kGruMaxLayers];         // per tensor: w_ih, w_hh, b_ih, b_hh of layer 0, 1, ...
    int64_t start[4 * kGruMaxLayers + 1];  // packed offsets of the tensors
    int32_t n_dst, accumulate;
    // optimizer epilogue (asac_adam_epilogue_t): the finished gradient's parameter is stepped by the lane that summed it
    int32_t adam_on;
    float* adam_param;
    const float* adam_grad;
    float* adam_m;
    float* adam_v;
    AdamScalars adam_c;
    const int64_t* adam_steps;
};

// 64 parameters per workgroup; wave s of the 16 sums a contiguous slice of the workgroups' partials (coalesced
// rows, the loads of a slice in flight together), then wave 0 adds the 16 slice sums in order: fixed order for a
// given launch shape, a few load latencies deep instead of `blocks` of them
constexpr int kReduceSlices = 16;
__global__ __launch_bounds__(64 * kReduceSlices) void k_gru_reduce(const float* __restrict__ partial, int blocks,
                                                                   int64_t n, const GruGradDst g) {
    __shared__ float part[kReduceSlices][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const int per = (blocks + kReduceSlices - 1) / kReduceSlices;
    const int lo = sl * per, hi = min(lo + per, blocks);
    // (epilogue: this lane's destination, its parameter and moments are requested with the partials, not behind them)
    const bool owner = sl == 0 && i < n && !g.packed;
    float* d = nullptr;
    float d_old = 0.f, pv = 0.f, mv = 0.f, vv = 0.f;
    int64_t off = 0, done = 0;
    if (owner) {
        int k = 0;
        while (k + 1 < g.n_dst && i >= g.start[k + 1]) ++k;
        d = g.dst[k] + (i - g.start[k]);
        if (g.accumulate) d_old = *d;
        if (g.adam_on) {
            off = d - g.adam_grad;
            pv = g.adam_param[off], mv = g.adam_m[off], vv = g.adam_v[off];
            done = *g.adam_steps;
        }
    }
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
    for (int w = 0; w < kReduceSlices; ++w) s += part[w][lane];
    if (g.packed) {
        g.packed[i] = s;
        return;
    }
    s = g.accumulate ? d_old + s : s;
    *d = s;
    if (g.adam_on) {
        float step_size, bc2_sqrt;
        adam_bias_terms(g.adam_c, done, &step_size, &bc2_sqrt);
        adam1(pv, s, mv, vv, g.adam_c, step_size, bc2_sqrt);
        g.adam_param[off] = pv, g.adam_m[off] = mv, g.adam_v[off] = vv;
    }
}

constexpr size_t kGruLdsLimit = 128 * 1024;     // of the CU's 160 KB; one workgroup per CU is plenty here

static int gru_lds_limit(const void* fn, bool& done, const char* where) {
    if (done) return 0;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGruLdsLimit);
    if (err != hipSuccess) {
        set_error(err, where);
        return (int)err;
    }
    done = true;
    return 0;
}

static int gru_maxd(const asac_gru_desc_t& d) { return (d.input > 8 || d.hidden > 8) ? 16 : 8; }

static bool gru_desc_ok(const asac_gru_desc_t& d) {
    if (d.layers < 1 || d.layers > kGruMaxLayers || d.hidden < 1 || d.hidden > kGruMaxDim) return false;
    if (d.input < 1 || d.input > kGruMaxDim) return false;
    int hp = 1;
    while (hp < d.hidden) hp <<= 1;
    if (hp != d.hidden_pow2) return false;
    const int rows = kGruWave / (4 * hp), maxd = gru_maxd(d);
    return (size_t)gru_bwd_plan(d, rows, maxd).total * sizeof(float) <= kGruLdsLimit &&
           (size_t)gru_fwd_plan(kGruWave / (4 * hp), maxd).total * sizeof(float) <= kGruLdsLimit;
}

static void gru_fill_ptrs(GruArgs& a, const float* const* w_ih, const float* const* w_hh,
                          const float* const* b_ih, const float* const* b_hh) {
    for (int l = 0; l < a.d.layers; ++l) {
        a.w_ih[l] = w_ih[l];
        a.w_hh[l] = w_hh[l];
        a.b_ih[l] = b_ih[l];
        a.b_hh[l] = b_hh[l];
    }
}

}  // namespace asac

using namespace asac;

extern "C" {

int64_t asac_gru_param_count(const asac_gru_desc_t* desc) {
    if (!desc || !gru_desc_ok(*desc)) return -1;
    int64_t n = 0;
    for (int l = 0; l < desc->layers; ++l) n += gru_layer_params(*desc, l);
    return n;
}

int64_t asac_gru_backward_workspace(const asac_gru_desc_t* desc, int B) {
    if (!desc || !gru_desc_ok(*desc) || B <= 0) return -1;
    const int rows = kGruWave / (4 * desc->hidden_pow2);      // four lanes per unit
    return (int64_t)((B + rows - 1) / rows) * asac_gru_param_count(desc);
}

static int gru_forward_launch(const char* where, const asac_gru_desc_t* desc, const float* const* w_ih,
                              const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                              const float* const* t_w_ih, const float* const* t_w_hh, const float* const* t_b_ih,
                              const float* const* t_b_hh, const float* x, int64_t x_stride_b, int64_t x_stride_t,
                              const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask, int64_t mask_stride_b,
                              int B, int L, float* hn_out, float* out_top, float* gates_out, float* t_hn_out,
                              float* t_out_top, void* stream) {
    const bool twin = t_w_ih != nullptr;
    if (!desc || !gru_desc_ok(*desc) || B <= 0 || L <= 0 || !x || !hn_out) return bad_arg(where);
    if (twin && (!t_w_hh || !t_b_ih || !t_b_hh || !t_hn_out)) return bad_arg(where);
    GruFwdJobs jobs{};
    GruArgs& a = jobs.job[0];
    a.d = *desc;
    gru_fill_ptrs(a, w_ih, w_hh, b_ih, b_hh);
    a.x = x; a.x_sb = x_stride_b; a.x_st = x_stride_t;
    a.h0 = h0; a.h0_sb = h0_stride_b;
    a.pad = padding_mask; a.pad_sb = mask_stride_b;
    a.B = B; a.L = L;
    a.hn = hn_out;
    a.out_top = out_top;
    a.gates = gates_out;
    if (twin) {
        GruArgs& t = jobs.job[1];
        t = a;
        gru_fill_ptrs(t, t_w_ih, t_w_hh, t_b_ih, t_b_hh);
        t.hn = t_hn_out;
        t.out_top = t_out_top;
        t.gates = nullptr;
    }
    const int rows = kGruWave / (4 * desc->hidden_pow2), blocks = (B + rows - 1) / rows;   // 4 lanes per unit
    const dim3 grid(blocks, twin ? 2 : 1);
    hipStream_t s = as_stream(stream);
    static bool attr[4] = {false, false, false, false};
    const int maxd = gru_maxd(*desc);
    const size_t lds = (size_t)gru_fwd_plan(rows, maxd).total * sizeof(float);
#define ASAC_GRU_FWD(MAXD, TWO, SLOT)                                                                          \
    do {                                                                                                       \
        if (int rc = gru_lds_limit(reinterpret_cast<const void*>(k_gru_fwd<MAXD, TWO>), attr[SLOT], where)) return rc; \
        ASAC_LAUNCH((k_gru_fwd<MAXD, TWO>), grid, dim3(kFwdThreads), lds, s, jobs);                            \
    } while (0)
    if (maxd == 8 && desc->layers == 1) ASAC_GRU_FWD(8, false, 0);
    else if (maxd == 8) ASAC_GRU_FWD(8, true, 1);
    else if (desc->layers == 1) ASAC_GRU_FWD(16, false, 2);
    else ASAC_GRU_FWD(16, true, 3);
#undef ASAC_GRU_FWD
    return finish_launch(where);
}

int asac_gru_forward(const asac_gru_desc_t* desc, const float* const* w_ih, const float* const* w_hh,
                     const float* const* b_ih, const float* const* b_hh, const float* x, int64_t x_stride_b,
                     int64_t x_stride_t, const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask,
                     int64_t mask_stride_b, int B, int L, float* hn_out, float* out_top, float* gates_out,
                     void* stream) {
    return gru_forward_launch("asac_gru_forward", desc, w_ih, w_hh, b_ih, b_hh, nullptr, nullptr, nullptr, nullptr, x,
                              x_stride_b, x_stride_t, h0, h0_stride_b, padding_mask, mask_stride_b, B, L, hn_out,
                              out_top, gates_out, nullptr, nullptr, stream);
}

int asac_gru_forward_twin(const asac_gru_desc_t* desc, const float* const* w_ih, const float* const* w_hh,
                          const float* const* b_ih, const float* const* b_hh, const float* const* twin_w_ih,
                          const float* const* twin_w_hh, const float* const* twin_b_ih,
                          const float* const* twin_b_hh, const float* x, int64_t x_stride_b, int64_t x_stride_t,
                          const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask, int64_t mask_stride_b,
                          int B, int L, float* hn_out, float* out_top, float* gates_out, float* twin_hn_out,
                          float* twin_out_top, void* stream) {
    if (!twin_w_ih) return bad_arg("asac_gru_forward_twin");
    return gru_forward_launch("asac_gru_forward_twin", desc, w_ih, w_hh, b_ih, b_hh, twin_w_ih, twin_w_hh, twin_b_ih,
                              twin_b_hh, x, x_stride_b, x_stride_t, h0, h0_stride_b, padding_mask, mask_stride_b, B, L,
                              hn_out, out_top, gates_out, twin_hn_out, twin_out_top, stream);
}

static int gru_backward_launch(const char* where, const asac_gru_desc_t* desc, const float* const* w_ih,
                               const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                               const float* x, int64_t x_stride_b, int64_t x_stride_t, const float* h0,
                               int64_t h0_stride_b, const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L,
                               const float* hn, const float* gates, const float* grad_hn, const float* grad_top,
                               const float* grad_top_members, int members, int position, float* grad_x,
                               float* grad_h0, float* grad_params, float* const* grad_param_tensors, int accumulate,
                               const asac_adam_epilogue_t* adam, float* workspace, void* stream) {
    if (!desc || !gru_desc_ok(*desc) || B <= 0 || L <= 0 || !x || !hn || !gates || !workspace ||
        (!grad_params == !grad_param_tensors) || (adam && !grad_param_tensors))
        return bad_arg(where);
    if (adam && !(adam->param_base && adam->grad_base && adam->exp_avg_base && adam->exp_avg_sq_base && adam->steps_done))
        return bad_arg(where);
    GruArgs a{};
    a.d = *desc;
    gru_fill_ptrs(a, w_ih, w_hh, b_ih, b_hh);
    a.x = x; a.x_sb = x_stride_b; a.x_st = x_stride_t;
    a.h0 = h0; a.h0_sb = h0_stride_b;
    a.pad = padding_mask; a.pad_sb = mask_stride_b;
    a.B = B; a.L = L;
    a.hn = const_cast<float*>(hn);
    a.gates = const_cast<float*>(gates);
    a.g_hn = grad_hn;
    a.g_top = grad_top;
    a.g_top_m = grad_top_members;
    a.g_top_E = members;
    a.L_run = grad_top_members ? position + 1 : L;
    a.g_x = grad_x;
    a.g_h0 = grad_h0;
    a.partial = workspace;
    a.param_count = asac_gru_param_count(desc);
    const int rows = kGruWave / (4 * desc->hidden_pow2), blocks = (B + rows - 1) / rows;
    hipStream_t s = as_stream(stream);
    if (grad_x && a.L_run < L) {        // the inputs behind the position receive no gradient: the kernel does not visit them
        if (hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)B * L * desc->input, s) != hipSuccess) return finish_launch(where);
    }
    static bool attr[4] = {false, false, false, false};
    const int maxd = gru_maxd(*desc);
    const size_t lds = (size_t)gru_bwd_plan(*desc, rows, maxd).total * sizeof(float);
#define ASAC_GRU_BWD(MAXD, TWO, SLOT)                                                                              \
    do {                                                                                                           \
        if (int rc = gru_lds_limit(reinterpret_cast<const void*>(k_gru_bwd<MAXD, TWO>), attr[SLOT], where))        \
            return rc;                                                                                             \
        ASAC_LAUNCH((k_gru_bwd<MAXD, TWO>), dim3(blocks), dim3(kBwdThreads), lds, s, a);                           \
    } while (0)
    if (maxd == 8 && desc->layers == 1) ASAC_GRU_BWD(8, false, 0);
    else if (maxd == 8) ASAC_GRU_BWD(8, true, 1);
    else if (desc->layers == 1) ASAC_GRU_BWD(16, false, 2);
    else ASAC_GRU_BWD(16, true, 3);
#undef ASAC_GRU_BWD
    GruGradDst g{};
    g.packed = grad_params;
    g.accumulate = accumulate;
    if (grad_param_tensors) {
        int64_t off = 0;
        for (int l = 0; l < desc->layers; ++l) {
            const int H = desc->hidden, I = l == 0 ? desc->input : H;
            const int64_t sizes[4] = {3LL * H * I, 3LL * H * H, 3LL * H, 3LL * H};
            for (int q = 0; q < 4; ++q) {
                g.dst[g.n_dst] = grad_param_tensors[4 * l + q];
                g.start[g.n_dst++] = off;
                off += sizes[q];
            }
        }
        g.start[g.n_dst] = off;
    }
    if (adam) {
        g.adam_on = 1;
        g.adam_param = adam->param_base;
        g.adam_grad = adam->grad_base;
        g.adam_m = adam->exp_avg_base;
        g.adam_v = adam->exp_avg_sq_base;
        g.adam_c = adam_scalars(adam->lr, adam->beta1, adam->beta2, adam->eps);
        g.adam_steps = adam->steps_done;
    }
    // launched once (not under the repeat knob: it may accumulate)
    hipLaunchKernelGGL(k_gru_reduce, dim3((unsigned)((a.param_count + 63) / 64)), dim3(64 * kReduceSlices), 0, s, workspace,
                       blocks, a.param_count, g);
    return finish_launch(where);
}

int asac_gru_backward(const asac_gru_desc_t* desc, const float* const* w_ih, const float* const* w_hh,
                      const float* const* b_ih, const float* const* b_hh, const float* x, int64_t x_stride_b,
                      int64_t x_stride_t, const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask,
                      int64_t mask_stride_b, int B, int L, const float* hn, const float* gates,
                      const float* grad_hn, const float* grad_top, float* grad_x, float* grad_h0,
                      float* grad_params, float* const* grad_param_tensors, int accumulate, float* workspace,
                      void* stream) {
    return gru_backward_launch("asac_gru_backward", desc, w_ih, w_hh, b_ih, b_hh, x, x_stride_b, x_stride_t, h0,
                               h0_stride_b, padding_mask, mask_stride_b, B, L, hn, gates, grad_hn, grad_top, nullptr, 0,
                               0, grad_x, grad_h0, grad_params, grad_param_tensors, accumulate, nullptr, workspace, stream);
}

int asac_gru_backward_at(const asac_gru_desc_t* desc, const float* const* w_ih, const float* const* w_hh,
                         const float* const* b_ih, const float* const* b_hh, const float* x, int64_t x_stride_b,
                         int64_t x_stride_t, const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask,
                         int64_t mask_stride_b, int B, int L, const float* hn, const float* gates,
                         const float* grad_top_members, int members, int position, float* grad_x, float* grad_h0,
                         float* grad_params, float* const* grad_param_tensors, int accumulate,
                         const asac_adam_epilogue_t* adam, float* workspace, void* stream) {
    if (!grad_top_members || members <= 0 || position < 0 || position >= L) return bad_arg("asac_gru_backward_at");
    return gru_backward_launch("asac_gru_backward_at", desc, w_ih, w_hh, b_ih, b_hh, x, x_stride_b, x_stride_t, h0,
                               h0_stride_b, padding_mask, mask_stride_b, B, L, hn, gates, nullptr, nullptr,
                               grad_top_members, members, position, grad_x, grad_h0, grad_params, grad_param_tensors,
                               accumulate, adam, workspace, stream);
}

}  // extern "C"
