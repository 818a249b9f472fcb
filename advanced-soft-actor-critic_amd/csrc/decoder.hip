// Observation decoder of the recurrent prediction models (`use_prediction`), forward and backward on f32 MFMA.
//
// Replaces, inside `_train_rpm` (reference algorithm/sac_base.py:1798-1839), the transposed-convolution decoder every
// reference plugin with an image observation model builds (envs/roller/nn_visual_hard_attn.py:64-96,
// envs/roller/nn_visual_hard.py:47-57, envs/pyramid/nn_visual.py:50-60) through
// `ConvTransposeLayers` (algorithm/nn_models/layers/image_layers.py:231-253):
//
//   state [N, S<=16] -> Linear(S, 64) GELU -> Linear(64, 128) -> [32, 2, 2]
//         -> ConvTranspose2d(32, 32, 4, 2) LeakyReLU -> [32, 6, 6]
//         -> ConvTranspose2d(32, 16, 8, 4) LeakyReLU -> [16, 28, 28]
//         -> ConvTranspose2d(16, 3, 3, 1)  LeakyReLU -> [3, 30, 30]
//
// Formulation.  Every layer is a channel-mixing GEMM per (pixel, tap) with the STATES as the N dimension of
// `v_mfma_f32_16x16x4_f32`: a workgroup owns a group of 16 states end to end, D[out channel][state] +=
// W[out channel][in channel] X[in channel][state].  With the weights as the A operand the accumulator of one layer
//   lane (q = lane >> 4, x = lane & 15) holds  D[4 q + r][state x],  r = 0..3
// IS the B operand of the next layer's four k-steps (step r contracts channels {4 q' + r : q' = 0..3}); a 16-channel x
// 16-state tile is therefore kept as one float4 per lane ("T16", 1 KB, stored / loaded as one coalesced 16-byte access
// per lane) everywhere: in LDS between the layers of a kernel and in HBM between kernels.  Weights are re-packed into that
// operand order once per call by `k_dec_pack` (one float4 per lane and K-tile).  The contractions over the states
// (parameter gradients) need the transposed tile ("TT16": lane (q, x = channel) holds states 4 s + q); tiles are turned
// through LDS.
//
// Kernels (grid = groups of 16 states, 256 threads = 4 waves):
//   k_dec_fwd12   dense head, ConvTranspose 1 and 2.  The 64 taps of layer 2 live in REGISTERS, dealt over the waves by
//                 output row phase (wave w: ky in {w, w + 4}), layer-1 activations in LDS; a wave forms the four output
//                 pixels of a 4-pixel block from the <= 4 input pixels under it.  Saves z1, h0, act1, act2.
//   k_dec_fwd3    ConvTranspose 3 in scatter form: per input pixel one [27 (oc, tap) x 16 ic] x [16 ic x 16 states]
//                 product, its rows added (read - add - write in program order inside ONE wave: deterministic) into a five-row ring
//                 of output rows in LDS; waves own output row ranges and recompute a two-row halo so that no wave waits
//                 for another; bias + LeakyReLU -> frames [N, 3, 30, 30].
//   k_dec_bwd3    d frames -> d z3 (ring in LDS) -> d act2 (gather form) -> d z2 (stored), weight / bias gradient of layer 3.
//   k_dec_bwd2dx  d z2 -> d act1 (taps in registers by wave, cross-wave sum through LDS) -> d z1 -> layer 1 and dense
//                 head backward -> d state; their parameter gradients.
//   k_dec_bwd2dw  weight / bias gradient of layer 2 (64 taps x [32 x 16] accumulators in registers, dealt by row phase).
//   k_dec_reduce  per-group partial parameter gradients summed in group order (deterministic) into the gradients.
#include <hip/hip_runtime.h>

#include "asac_common.h"
#include "asac_gelu.h"

namespace asac {
namespace dec {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 256;
constexpr float kSlope = 0.01f;   // nn.LeakyReLU default

// ---- tiles per group in the saved-activation buffers (256 floats each) ------------------------------------------
constexpr int kZ1Tiles = 4;      // dense-1 pre-activations [64]
constexpr int kH0Tiles = 8;      // dense-2 output = layer-1 input: (pixel p, channel tile t)
constexpr int kA1Tiles = 72;     // layer-1 output (36 pixels x 2 channel tiles), post LeakyReLU
constexpr int kA2Tiles = 784;    // layer-2 output (28 x 28 pixels, 16 channels), post LeakyReLU

// ---- packed operand buffer (floats) -------------------------------------------------------------------------------
constexpr int OFF_PA_D1 = 0;          // [mt 4][64][4]
constexpr int OFF_PA_D2 = 1024;       // [p 4][t 2][kt 4][64][4]
constexpr int OFF_PA_C1 = 9216;       // [tap 16][mt 2][kt 2][64][4]
constexpr int OFF_PA_C2 = 25600;      // [tap 64][kt 2][64][4]
constexpr int OFF_PA_C3 = 58368;      // [mt 2][64][4]
constexpr int OFF_PAT_C3 = 58880;     // [s 8][64]   (s = 7 unused)
constexpr int OFF_PAT_C2 = 59392;     // [tap 64][mt 2][64][4]
constexpr int OFF_PAT_C1 = 92160;     // [tap 16][mt 2][kt 2][64][4]
constexpr int OFF_PAT_D2 = 108544;    // [mt 4][p 4][t 2][64][4]
constexpr int OFF_PAT_D1 = 116736;    // [kt 4][64][4]
constexpr int OFF_PA_C3X = 117760;    // [mt 3][64][4]   rows m' = 4 (oc, ky) + kx (kx = 3: zero)
constexpr int kPackedFloats = 118528;

// ---- per-group partial gradients (floats) ---------------------------------------------------------------------------
constexpr int POFF_W2 = 0;            // [w 4][dy 2][kx 8][mt 2][64][4]
constexpr int POFF_W1 = 32768;        // [tap 16][mt 2][nt 2][64][4]
constexpr int POFF_WD2 = 49152;       // [p 4][t 2][nt 4][64][4]
constexpr int POFF_WD1 = 57344;       // [mt 4][64][4]
constexpr int POFF_W3 = 58368;        // [nt 2][64][4]
constexpr int POFF_B2 = 58880;        // [w 4][16]
constexpr int POFF_B1 = 58944;        // [32]
constexpr int POFF_BD2 = 58976;       // [128]
constexpr int POFF_BD1 = 59104;       // [64]
constexpr int POFF_B3 = 59168;        // [w 4][4]
constexpr int kPartialFloats = 59184;


struct Params {            // the ten parameter tensors (or their gradients), natural PyTorch layouts
    float* wd1;   // [64, S]
    float* bd1;   // [64]
    float* wd2;   // [128, 64]
    float* bd2;   // [128]
    float* w1;    // [32, 32, 4, 4]   (in, out, kh, kw)
    float* b1;    // [32]
    float* w2;    // [32, 16, 8, 8]
    float* b2;    // [16]
    float* w3;    // [16, 3, 3, 3]
    float* b3;    // [3]
};

__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }

#define ASAC_MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// the four k-steps of one 16-channel K-tile
__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
    c = ASAC_MF(a[0], b[0], c);
    c = ASAC_MF(a[1], b[1], c);
    c = ASAC_MF(a[2], b[2], c);
    c = ASAC_MF(a[3], b[3], c);
    return c;
}

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * kSlope; }
__device__ __forceinline__ float leaky_grad_from_out(float out) { return out > 0.f ? 1.f : kSlope; }

// position of a state inside a 16-float row of a transposed tile: states 4 s + q (s = 0..3) are consecutive
__device__ __forceinline__ int tpos(int state) { return ((state & 3) << 2) | (state >> 2); }

// lanes of ONE wave exchange data through LDS: order the accesses for the compiler (the LDS itself is in order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// T16 tile (registers) -> TT16 (registers) through a 256-float scratch of the wave
__device__ __forceinline__ f32x4 transpose_tile(float* scratch, const f32x4 v, int q, int x) {
    const int px = tpos(x);
#pragma unroll
    for (int r = 0; r < 4; ++r) scratch[(4 * q + r) * 16 + px] = v[r];
    wave_sync();
    const f32x4 u = *reinterpret_cast<const f32x4*>(scratch + x * 16 + 4 * q);
    wave_sync();
    return u;
}

// sum over the 16 lanes that share q (the states of a T16 tile); every lane ends with the total
__device__ __forceinline__ float sum_over_x(float v) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// weights -> operand order
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dec_pack(const Params P, int S, float* __restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kPackedFloats) return;
    float v = 0.f;
    if (i >= OFF_PAT_C3 && i < OFF_PAT_C2) {      // [s][64]
        const int j = i - OFF_PAT_C3, s = j >> 6, l = j & 63, q = l >> 4, x = l & 15, m = 4 * s + q;
        if (s < 7 && m < 27) v = P.w3[x * 27 + m];
        packed[i] = v;
        return;
    }
    if (i >= OFF_PA_C3X) {
        const int j0 = i - OFF_PA_C3X, r = j0 & 3, l = (j0 >> 2) & 63, mt = j0 >> 8, q = l >> 4, x = l & 15;
        const int mrow = 16 * mt + x, grp = mrow >> 2, kx = mrow & 3;
        packed[i] = (grp < 9 && kx < 3) ? P.w3[(4 * q + r) * 27 + grp * 3 + kx] : 0.f;
        return;
    }
    int sec_off;
    if (i < OFF_PA_D2) sec_off = OFF_PA_D1;
    else if (i < OFF_PA_C1) sec_off = OFF_PA_D2;
    else if (i < OFF_PA_C2) sec_off = OFF_PA_C1;
    else if (i < OFF_PA_C3) sec_off = OFF_PA_C2;
    else if (i < OFF_PAT_C3) sec_off = OFF_PA_C3;
    else if (i < OFF_PAT_C1) sec_off = OFF_PAT_C2;
    else if (i < OFF_PAT_D2) sec_off = OFF_PAT_C1;
    else if (i < OFF_PAT_D1) sec_off = OFF_PAT_D2;
    else sec_off = OFF_PAT_D1;
    const int j = i - sec_off, r = j & 3, l = (j >> 2) & 63, tile = j >> 8, q = l >> 4, x = l & 15, c = 4 * q + r;
    switch (sec_off) {
        case OFF_PA_D1: v = c < S ? P.wd1[(16 * tile + x) * S + c] : 0.f; break;
        case OFF_PA_D2: {
            const int kt = tile & 3, t = (tile >> 2) & 1, p = tile >> 3;
            v = P.wd2[((16 * t + x) * 4 + p) * 64 + 16 * kt + c];
        } break;
        case OFF_PA_C1: {
            const int kt = tile & 1, mt = (tile >> 1) & 1, tap = tile >> 2;
            v = P.w1[((16 * kt + c) * 32 + 16 * mt + x) * 16 + tap];
        } break;
        case OFF_PA_C2: {
            const int kt = tile & 1, tap = tile >> 1;
            v = P.w2[((16 * kt + c) * 16 + x) * 64 + tap];
        } break;
        case OFF_PA_C3: {
            const int m = 16 * tile + x;
            v = m < 27 ? P.w3[c * 27 + m] : 0.f;
        } break;
        case OFF_PAT_C2: {
            const int mt = tile & 1, tap = tile >> 1;
            v = P.w2[((16 * mt + x) * 16 + c) * 64 + tap];
        } break;
        case OFF_PAT_C1: {
            const int kt = tile & 1, mt = (tile >> 1) & 1, tap = tile >> 2;
            v = P.w1[((16 * mt + x) * 32 + 16 * kt + c) * 16 + tap];
        } break;
        case OFF_PAT_D2: {
            const int t = tile & 1, p = (tile >> 1) & 3, mt = tile >> 3;
            v = P.wd2[((16 * t + c) * 4 + p) * 64 + 16 * mt + x];
        } break;
        default:   // OFF_PAT_D1
            v = x < S ? P.wd1[(16 * tile + c) * S + x] : 0.f;
            break;
    }
    packed[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: dense head + ConvTranspose 1 + ConvTranspose 2
// ---------------------------------------------------------------------------------------------------------------------
struct Fwd12Args {
    const float* x;          // [N, S]
    int64_t x_stride;        // floats between states
    int N, S;
    const float* packed;
    const float* bd1;
    const float* bd2;
    const float* b1;
    const float* b2;
    f32x4* z1;               // [G][4][64]
    f32x4* h0;               // [G][8][64]
    f32x4* act1;             // [G][72][64]
    f32x4* act2;             // [G][784][64]
};

__global__ void __launch_bounds__(kThreads) k_dec_fwd12(const Fwd12Args a) {
    __shared__ f32x4 s_act1[kA1Tiles * 64];   // 72 KB
    __shared__ f32x4 s_h1[4 * 64];
    __shared__ f32x4 s_h0[kH0Tiles * 64];
    const int g = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    const f32x4* __restrict__ pk = reinterpret_cast<const f32x4*>(a.packed);

    // layer-2 taps of this wave: ky in {w, w + 4}, all kx -> registers (in flight under the head)
    f32x4 wr[2][2][4][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const int tap = (w + 4 * dy) * 8 + px + 4 * dx;
                    wr[dy][dx][px][kt] = pk[OFF_PA_C2 / 4 + (tap * 2 + kt) * 64 + l];
                }

    // ---- dense 1: wave w forms rows 16 w .. 16 w + 15 ---------------------------------------------------------------
    {
        const int64_t sidx = min((int64_t)g * 16 + x, (int64_t)a.N - 1);
        f32x4 xv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * q + r;
            xv[r] = c < a.S ? a.x[sidx * a.x_stride + c] : 0.f;
        }
        f32x4 acc = mfma4(pk[OFF_PA_D1 / 4 + w * 64 + l], xv, zero4());
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] += a.bd1[16 * w + 4 * q + r];
            h[r] = gelu_f(acc[r]);
        }
        a.z1[((int64_t)g * kZ1Tiles + w) * 64 + l] = acc;
        s_h1[w * 64 + l] = h;
    }
    __syncthreads();
    // ---- dense 2: wave w forms pixel p = w (both channel tiles) ----------------------------------------------------
    {
        f32x4 hb[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) hb[kt] = s_h1[kt * 64 + l];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc = zero4();
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) acc = mfma4(pk[OFF_PA_D2 / 4 + ((w * 2 + t) * 4 + kt) * 64 + l], hb[kt], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += a.bd2[(16 * t + 4 * q + r) * 4 + w];
            s_h0[(w * 2 + t) * 64 + l] = acc;
            a.h0[((int64_t)g * kH0Tiles + w * 2 + t) * 64 + l] = acc;
        }
    }
    __syncthreads();
    // ---- ConvTranspose 1 (2x2 -> 6x6, k 4, s 2): wave w owns the output phase (py, px) = (w >> 1, w & 1) -----------
    {
        const int py = w >> 1, px = w & 1;
        for (int by = 0; by < 3; ++by)
            for (int bx = 0; bx < 3; ++bx) {
                f32x4 acc[2] = {zero4(), zero4()};
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int iy = by - dy;
                    if (iy < 0 || iy > 1) continue;
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int ix = bx - dx;
                        if (ix < 0 || ix > 1) continue;
                        const int p = iy * 2 + ix, tap = (py + 2 * dy) * 4 + px + 2 * dx;
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt) {
                            const f32x4 b = s_h0[(p * 2 + kt) * 64 + l];
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                acc[mt] = mfma4(pk[OFF_PA_C1 / 4 + ((tap * 2 + mt) * 2 + kt) * 64 + l], b, acc[mt]);
                        }
                    }
                }
                const int opix = (2 * by + py) * 6 + 2 * bx + px;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][r] = leaky(acc[mt][r] + a.b1[16 * mt + 4 * q + r]);
                    s_act1[(opix * 2 + mt) * 64 + l] = acc[mt];
                    a.act1[((int64_t)g * kA1Tiles + opix * 2 + mt) * 64 + l] = acc[mt];
                }
            }
    }
    __syncthreads();
    // ---- ConvTranspose 2 (6x6 -> 28x28, k 8, s 4): wave w owns output rows 4 by + w ------------------------------
    f32x4 bias2;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias2[r] = a.b2[4 * q + r];
    f32x4* __restrict__ out = a.act2 + (int64_t)g * kA2Tiles * 64 + l;
    for (int by = 0; by < 7; ++by)
        for (int bx = 0; bx < 7; ++bx) {
            f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int iy = by - dy;
                if (iy < 0 || iy > 5) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int ix = bx - dx;
                    if (ix < 0 || ix > 5) continue;
                    const int p = iy * 6 + ix;
                    const f32x4 b0 = s_act1[(p * 2 + 0) * 64 + l], b1 = s_act1[(p * 2 + 1) * 64 + l];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int px = 0; px < 4; ++px) acc[px] = ASAC_MF(wr[dy][dx][px][0][r], b0[r], acc[px]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int px = 0; px < 4; ++px) acc[px] = ASAC_MF(wr[dy][dx][px][1][r], b1[r], acc[px]);
                }
            }
            const int opix0 = (4 * by + w) * 28 + 4 * bx;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[px][r] = leaky(acc[px][r] + bias2[r]);
                out[(int64_t)(opix0 + px) * 64] = acc[px];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// ConvTranspose 3: ring of output rows in LDS, one per wave
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kRingRows = 5, kPitch = 20;
constexpr int kRowFloats = 3 * 30 * kPitch;              // 1800
constexpr int kRingFloats = kRingRows * kRowFloats;      // 9000 (36 KB) per wave
constexpr int kTailLds = (4 * kRingFloats + 4 * 4 * 256) * 4;   // rings + four transposition tiles per wave: 156.6 KB

struct Fwd3Args {
    const float* packed;
    const float* b3;
    const f32x4* act2;
    float* out;        // [N, 3, 30, 30]
    int N;
};

__global__ void __launch_bounds__(kThreads) k_dec_fwd3(const Fwd3Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int g = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    float* ring = lds + w * kRingFloats;
    const f32x4* __restrict__ pk = reinterpret_cast<const f32x4*>(a.packed);
    // operand rows m' = 4 j + kx, j = (oc, ky): a lane's accumulator registers r = 0..2 ARE the three kx of ITS group
    // j = 4 mt + q, so the sum over kx (output pixel X takes kx from input pixel X - kx) is a running sum in registers
    // along the row and only the sum over ky goes through the ring
    f32x4 pa[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) pa[mt] = pk[OFF_PA_C3X / 4 + mt * 64 + l];
    // output rows owned by the waves: [0, 9) [9, 16) [16, 23) [23, 30); input rows two above
    const int R0 = w == 0 ? 0 : 2 + 7 * w, R1 = w == 3 ? 30 : 9 + 7 * w;
    const int in0 = max(R0 - 2, 0), in1 = min(R1, 28);
    for (int i = l; i < kRingFloats / 4; i += 64) reinterpret_cast<f32x4*>(ring)[i] = zero4();
    int cst[3], kyv[3];
    bool valid[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int j = 4 * mt + q, jj = min(j, 8);
        valid[mt] = j < 9;
        kyv[mt] = jj % 3;
        cst[mt] = (jj / 3) * 30 * kPitch + tpos(x);
    }
    const float b3v[3] = {a.b3[0], a.b3[1], a.b3[2]};
    const f32x4* __restrict__ src = a.act2 + (int64_t)g * kA2Tiles * 64 + l;
    wave_sync();
    f32x4 vt[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) vt[j] = src[(int64_t)(in0 * 28 + j) * 64];
    for (int y = in0; y < in1; ++y) {
        int base[3];
        {
            const int rb0 = (y % kRingRows) * kRowFloats, rb1 = ((y + 1) % kRingRows) * kRowFloats,
                      rb2 = ((y + 2) % kRingRows) * kRowFloats;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) base[mt] = cst[mt] + (kyv[mt] == 0 ? rb0 : kyv[mt] == 1 ? rb1 : rb2);
        }
        // ky = 2 opens an output row (plain store: what the slot held is done), ky = 1 / 0 add to it
        auto emit = [&](int mt, int X, float v) {
            if (valid[mt]) {
                float* dst = ring + base[mt] + X * kPitch;
                float cur = 0.f;
                if (kyv[mt] != 2) cur = *dst;
                *dst = cur + v;
            }
        };
        float p1[3] = {0.f, 0.f, 0.f}, p2[3] = {0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < 28; c0 += 7) {
            f32x4 acc[7][3];
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) acc[j][mt] = zero4();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 7; ++j)
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) acc[j][mt] = ASAC_MF(pa[mt][r], vt[j][r], acc[j][mt]);
            // the next chunk's tiles travel under this chunk's sums and ring traffic
            {
                const int nc = c0 + 7 < 28 ? c0 + 7 : 0, ny = c0 + 7 < 28 ? y : min(y + 1, 27);
#pragma unroll
                for (int j = 0; j < 7; ++j) vt[j] = src[(int64_t)(ny * 28 + nc + j) * 64];
            }
            float e[7][3];
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    e[j][mt] = acc[j][mt][0] + p1[mt];
                    p1[mt] = acc[j][mt][1] + p2[mt];
                    p2[mt] = acc[j][mt][2];
                }
            // every output pixel of the row is touched once per input row: the chunk's read - add - writes are independent
            float cur[7][3];
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
                    cur[j][mt] = (valid[mt] && kyv[mt] != 2) ? ring[base[mt] + (c0 + j) * kPitch] : 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
                    if (valid[mt]) ring[base[mt] + (c0 + j) * kPitch] = cur[j][mt] + e[j][mt];
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            emit(mt, 28, p1[mt]);
            emit(mt, 29, p2[mt]);
        }
        wave_sync();
        // rows that are complete now
        const int f0 = y, f1 = (w == 3 && y == 27) ? 30 : y + 1;
        for (int yy = f0; yy < f1; ++yy) {
            if (yy < R0) continue;
            const float* row = ring + (yy % kRingRows) * kRowFloats;
            const int xx = l & 31;
#pragma unroll
            for (int oc = 0; oc < 3; ++oc)
#pragma unroll 4
                for (int i8 = 0; i8 < 8; ++i8) {
                    const int st = 2 * i8 + (l >> 5);
                    const int64_t state = (int64_t)g * 16 + st;
                    if (xx < 30 && state < a.N) {
                        const float v = row[(oc * 30 + xx) * kPitch + tpos(st)] + b3v[oc];
                        a.out[(state * 3 + oc) * 900 + yy * 30 + xx] = leaky(v);
                    }
                }
        }
        wave_sync();
    }
}

struct Bwd3Args {
    const float* packed;
    const f32x4* act2;
    const float* out;      // [N, 3, 30, 30] forward result (LeakyReLU sign)
    const float* gout;     // [N, 3, 30, 30]
    f32x4* dz2;            // [G][784][64]
    float* partial;        // [G][kPartialFloats]
    int N;
};

__global__ void __launch_bounds__(kThreads) k_dec_bwd3(const Bwd3Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int g = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    float* ring = lds + w * kRingFloats;
    float* scratch = lds + 4 * kRingFloats + w * (4 * 256);      // four tiles turned at a time
    const float* __restrict__ pk = a.packed;
    float pat[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) pat[s] = pk[OFF_PAT_C3 + s * 64 + l];
    const int Q0 = 7 * w, Q1 = Q0 + 7;
    // gather constants.  d act2: k-slot (s, q) <-> m = 4 s + q;  weight gradient: column x of tile nt <-> m = 16 nt + x
    int cx[7], kx_[7], cw[2], kw_[2];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int m = min(4 * s + q, 26), oc = m / 9, tap = m % 9;
        kx_[s] = tap / 3;
        cx[s] = (oc * 30 + tap % 3) * kPitch + tpos(x);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int m = min(16 * nt + x, 26), oc = m / 9, tap = m % 9;
        kw_[nt] = tap / 3;
        cw[nt] = (oc * 30 + tap % 3) * kPitch + 4 * q;
    }
    float b3acc[3] = {0.f, 0.f, 0.f};
    // d z3 of one output row: 24 (oc, state pair) items per lane, requested as one batch and put into the ring later
    const int xx = l & 31, sth = l >> 5;
    float rg[24], ro[24];
    auto row_request = [&](int yy) {
#pragma unroll
        for (int oc = 0; oc < 3; ++oc)
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) {
                const int64_t state = (int64_t)g * 16 + 2 * i8 + sth;
                const bool live = xx < 30 && state < a.N;
                const int64_t idx = live ? (state * 3 + oc) * 900 + yy * 30 + xx : 0;
                // (the loads themselves are unconditional — index 0 for the lanes without an item: a load under a lane
                // condition becomes a branch of its own with its own wait, 48 serial round trips per row instead of one batch)
                const float gv = a.gout[idx], ov = a.out[idx];
                rg[oc * 8 + i8] = live ? gv : 0.f;
                ro[oc * 8 + i8] = live ? ov : 1.f;
            }
    };
    auto row_store = [&](int yy) {
        float* row = ring + (yy % kRingRows) * kRowFloats;
        const bool own = w == 3 || yy < Q1;
#pragma unroll
        for (int oc = 0; oc < 3; ++oc)
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) {
                if (xx < 30) {
                    const float d = rg[oc * 8 + i8] * leaky_grad_from_out(ro[oc * 8 + i8]);
                    row[(oc * 30 + xx) * kPitch + tpos(2 * i8 + sth)] = d;
                    if (own) b3acc[oc] += d;
                }
            }
    };
    row_request(Q0);
    row_store(Q0);
    row_request(Q0 + 1);
    row_store(Q0 + 1);
    row_request(Q0 + 2);
    row_store(Q0 + 2);
    f32x4 acc3[2] = {zero4(), zero4()};
    const f32x4* __restrict__ src = a.act2 + (int64_t)g * kA2Tiles * 64 + l;
    f32x4* __restrict__ dst = a.dz2 + (int64_t)g * kA2Tiles * 64 + l;
    f32x4 vt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vt[j] = src[(int64_t)(Q0 * 28 + j) * 64];
    wave_sync();
    for (int y = Q0; y < Q1; ++y) {
        const bool more = y + 1 < Q1;          // row y + 3 is needed by the next iteration
        if (more) row_request(y + 3);
        int bx[7], bw[2];
        {
            const int rb0 = (y % kRingRows) * kRowFloats, rb1 = ((y + 1) % kRingRows) * kRowFloats,
                      rb2 = ((y + 2) % kRingRows) * kRowFloats;
#pragma unroll
            for (int s = 0; s < 7; ++s) bx[s] = cx[s] + (kx_[s] == 0 ? rb0 : kx_[s] == 1 ? rb1 : rb2);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bw[nt] = cw[nt] + (kw_[nt] == 0 ? rb0 : kw_[nt] == 1 ? rb1 : rb2);
        }
        for (int c0 = 0; c0 < 28; c0 += 4) {
            f32x4 cur[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = vt[j];
            {   // next chunk's tiles (next row's first at the end of a row)
                const int nc = c0 + 4 < 28 ? c0 + 4 : 0, ny = c0 + 4 < 28 ? y : min(y + 1, 27);
#pragma unroll
                for (int j = 0; j < 4; ++j) vt[j] = src[(int64_t)(ny * 28 + nc + j) * 64];
            }
            // d act2 of four pixels: four independent accumulator chains
            float bv[7][4];
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[s][j] = ring[bx[s] + (c0 + j) * kPitch];
            f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = ASAC_MF(pat[s], bv[s][j], acc[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 dz;
#pragma unroll
                for (int r = 0; r < 4; ++r) dz[r] = acc[j][r] * leaky_grad_from_out(cur[j][r]);
                dst[(int64_t)(y * 28 + c0 + j) * 64] = dz;
            }
            // layer-3 weight gradient: the four tiles turned through LDS together
            const int px = tpos(x);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) scratch[j * 256 + (4 * q + r) * 16 + px] = cur[j][r];
            wave_sync();
            f32x4 u[4], wv[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const f32x4*>(scratch + j * 256 + x * 16 + 4 * q);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wv[nt][j] = *reinterpret_cast<const f32x4*>(ring + bw[nt] + (c0 + j) * kPitch);
            wave_sync();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc3[0] = ASAC_MF(u[j][r], wv[0][j][r], acc3[0]);
                    acc3[1] = ASAC_MF(u[j][r], wv[1][j][r], acc3[1]);
                }
        }
        wave_sync();
        if (more) row_store(y + 3);
        wave_sync();
    }
    float* part = a.partial + (int64_t)g * kPartialFloats;
    // layer-3 weight gradient is a sum over the four waves' rows: through LDS, in wave order
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds);      // the rings are done
    red[(w * 2 + 0) * 64 + l] = acc3[0];
    red[(w * 2 + 1) * 64 + l] = acc3[1];
    __syncthreads();
    if (w < 2) {
        f32x4 s = red[(0 * 2 + w) * 64 + l];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += red[(ww * 2 + w) * 64 + l];
        reinterpret_cast<f32x4*>(part + POFF_W3)[w * 64 + l] = s;
    }
#pragma unroll
    for (int oc = 0; oc < 3; ++oc) {
        const float t = wave_sum(b3acc[oc]);
        if (l == 0) part[POFF_B3 + w * 4 + oc] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward through ConvTranspose 2 (input gradient), ConvTranspose 1 and the dense head
// ---------------------------------------------------------------------------------------------------------------------
struct Bwd2dxArgs {
    const float* x;
    int64_t x_stride;
    int N, S;
    const float* packed;
    const f32x4* z1;
    const f32x4* h0;
    const f32x4* act1;
    const f32x4* dz2;
    float* gx;             // [N, S] or NULL
    float* partial;
};

constexpr int kDxLds = (kA1Tiles * 256 + 2 * 8 * 256 + kH0Tiles * 256 * 2 + 4 * 256 * 3 + 4 * 256) * 4;

__global__ void __launch_bounds__(kThreads) k_dec_bwd2dx(const Bwd2dxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* s_dz1 = reinterpret_cast<f32x4*>(lds);                    // [72][64]   T16
    f32x4* s_red = s_dz1 + kA1Tiles * 64;                            // [2][4 waves][2 mt][64]
    f32x4* s_dh0 = s_red + 2 * 8 * 64;                               // [8][64]    T16
    float* s_h0T = reinterpret_cast<float*>(s_dh0 + kH0Tiles * 64);  // [8][256]   TT16
    f32x4* s_h1 = reinterpret_cast<f32x4*>(s_h0T + kH0Tiles * 256);  // [4][64]    T16 (gelu(z1))
    float* s_h1T = reinterpret_cast<float*>(s_h1 + 4 * 64);          // [4][256]   TT16
    f32x4* s_dzd1 = reinterpret_cast<f32x4*>(s_h1T + 4 * 256);       // [4][64]    T16
    float* scratch = reinterpret_cast<float*>(s_dzd1 + 4 * 64) + (threadIdx.x >> 6) * 256;
    const int g = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    const f32x4* __restrict__ pk = reinterpret_cast<const f32x4*>(a.packed);
    float* part = a.partial + (int64_t)g * kPartialFloats;

    // this wave's taps of layer 2 (ky in {w, w + 4}, all kx), transposed operand
    f32x4 wt[2][8][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int kx = 0; kx < 8; ++kx)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                wt[dy][kx][mt] = pk[OFF_PAT_C2 / 4 + (((w + 4 * dy) * 8 + kx) * 2 + mt) * 64 + l];
    // head activations for the end of the kernel: h0 (TT16), h1 = gelu(z1) (T16 + TT16), gelu'(z1) kept per wave
    f32x4 gd1;     // gelu'(z1) of tile w
    {
        const f32x4 z = a.z1[((int64_t)g * kZ1Tiles + w) * 64 + l];
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float hv, dv;
            gelu_parts(z[r], hv, dv);
            h[r] = hv, gd1[r] = dv;
        }
        s_h1[w * 64 + l] = h;
#pragma unroll
        for (int r = 0; r < 4; ++r) s_h1T[w * 256 + (4 * q + r) * 16 + tpos(x)] = h[r];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 v = a.h0[((int64_t)g * kH0Tiles + w * 2 + t) * 64 + l];
#pragma unroll
            for (int r = 0; r < 4; ++r) s_h0T[(w * 2 + t) * 256 + (4 * q + r) * 16 + tpos(x)] = v[r];
        }
    }
    // ---- d act1 = ConvTranspose2^T d z2: per input pixel, every wave its 16 taps, summed in wave order -------------
    const f32x4* __restrict__ dz2 = a.dz2 + (int64_t)g * kA2Tiles * 64 + l;
    f32x4 b1acc = zero4();
    f32x4 bt[2][8];      // the 16 tiles under the wave's taps of the NEXT input pixel: requested a pixel ahead
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) bt[dy][kx] = dz2[(int64_t)((w + 4 * dy) * 28 + kx) * 64];
    for (int p = 0; p < 36; ++p) {
        f32x4 cur[2][8];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) cur[dy][kx] = bt[dy][kx];
        {
            const int pn = min(p + 1, 35), iy = pn / 6, ix = pn - iy * 6;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int kx = 0; kx < 8; ++kx) bt[dy][kx] = dz2[(int64_t)((4 * iy + w + 4 * dy) * 28 + 4 * ix + kx) * 64];
        }
        f32x4 acc[2] = {zero4(), zero4()};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0] = ASAC_MF(wt[dy][kx][0][r], cur[dy][kx][r], acc[0]);
                    acc[1] = ASAC_MF(wt[dy][kx][1][r], cur[dy][kx][r], acc[1]);
                }
        f32x4* red = s_red + (p & 1) * 8 * 64;
        red[(w * 2 + 0) * 64 + l] = acc[0];
        red[(w * 2 + 1) * 64 + l] = acc[1];
        __syncthreads();
        if (w < 2) {
            f32x4 s = red[(0 * 2 + w) * 64 + l];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) s += red[(ww * 2 + w) * 64 + l];
            const f32x4 a1 = a.act1[((int64_t)g * kA1Tiles + p * 2 + w) * 64 + l];
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] *= leaky_grad_from_out(a1[r]);
            s_dz1[(p * 2 + w) * 64 + l] = s;
            b1acc += s;
        }
    }
    if (w < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = sum_over_x(b1acc[r]);
            if (x == 0) part[POFF_B1 + 16 * w + 4 * q + r] = t;
        }
    }
    __syncthreads();
    // ---- ConvTranspose 1 backward --------------------------------------------------------------------------------
    {   // d h0 of input pixel w
        const int iy = w >> 1, ix = w & 1;
        f32x4 acc[2] = {zero4(), zero4()};
        for (int tap = 0; tap < 16; ++tap) {
            const int opix = (2 * iy + (tap >> 2)) * 6 + 2 * ix + (tap & 3);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const f32x4 b = s_dz1[(opix * 2 + kt) * 64 + l];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = mfma4(pk[OFF_PAT_C1 / 4 + ((tap * 2 + mt) * 2 + kt) * 64 + l], b, acc[mt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            s_dh0[(w * 2 + mt) * 64 + l] = acc[mt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = sum_over_x(acc[mt][r]);
                if (x == 0) part[POFF_BD2 + (16 * mt + 4 * q + r) * 4 + w] = t;
            }
        }
    }
    {   // weight gradient of layer 1: wave w owns taps 4 w .. 4 w + 3
        for (int tt = 0; tt < 4; ++tt) {
            const int tap = 4 * w + tt;
            f32x4 acc[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};
            for (int p = 0; p < 4; ++p) {
                const int opix = (2 * (p >> 1) + (tap >> 2)) * 6 + 2 * (p & 1) + (tap & 3);
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(s_h0T + (p * 2 + 0) * 256 + x * 16 + 4 * q);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(s_h0T + (p * 2 + 1) * 256 + x * 16 + 4 * q);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 u = transpose_tile(scratch, s_dz1[(opix * 2 + nt) * 64 + l], q, x);
                    acc[0][nt] = mfma4(a0, u, acc[0][nt]);
                    acc[1][nt] = mfma4(a1, u, acc[1][nt]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    reinterpret_cast<f32x4*>(part + POFF_W1)[((tap * 2 + mt) * 2 + nt) * 64 + l] = acc[mt][nt];
        }
    }
    __syncthreads();
    // ---- dense 2 backward ------------------------------------------------------------------------------------------
    {   // weight gradient rows of pixel p = w
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 u = transpose_tile(scratch, s_dh0[(w * 2 + t) * 64 + l], q, x);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(s_h1T + nt * 256 + x * 16 + 4 * q);
                reinterpret_cast<f32x4*>(part + POFF_WD2)[((w * 2 + t) * 4 + nt) * 64 + l] = mfma4(u, b, zero4());
            }
        }
        // d h1 tile w, then through GELU
        f32x4 acc = zero4();
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) acc = mfma4(pk[OFF_PAT_D2 / 4 + (w * 8 + pt) * 64 + l], s_dh0[pt * 64 + l], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= gd1[r];
        s_dzd1[w * 64 + l] = acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = sum_over_x(acc[r]);
            if (x == 0) part[POFF_BD1 + 16 * w + 4 * q + r] = t;
        }
        // dense 1 weight gradient rows 16 w ..: A = d z^T, B = x^T (lane (q, c): states 4 s + q)
        const f32x4 u = transpose_tile(scratch, acc, q, x);
        f32x4 xb;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t state = (int64_t)g * 16 + 4 * s + q;
            xb[s] = (x < a.S && state < a.N) ? a.x[state * a.x_stride + x] : 0.f;
        }
        reinterpret_cast<f32x4*>(part + POFF_WD1)[w * 64 + l] = mfma4(u, xb, zero4());
    }
    __syncthreads();
    if (w == 0 && a.gx != nullptr) {
        f32x4 acc = zero4();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc = mfma4(pk[OFF_PAT_D1 / 4 + kt * 64 + l], s_dzd1[kt * 64 + l], acc);
        const int64_t state = (int64_t)g * 16 + x;
        if (state < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < a.S) a.gx[state * a.S + 4 * q + r] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight / bias gradient of ConvTranspose 2
// ---------------------------------------------------------------------------------------------------------------------
struct Bwd2dwArgs {
    const f32x4* act1;
    const f32x4* dz2;
    float* partial;
};
constexpr int kDwLds = (kA1Tiles * 256 + 4 * 4 * 256) * 4;

__global__ void __launch_bounds__(kThreads) k_dec_bwd2dw(const Bwd2dwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_a1T = lds;                                    // [72][256] TT16
    float* scratch = lds + kA1Tiles * 256 + (threadIdx.x >> 6) * (4 * 256);
    const int g = blockIdx.x, l = threadIdx.x & 63, w = threadIdx.x >> 6, q = l >> 4, x = l & 15;
    for (int i = w; i < kA1Tiles; i += 4) {
        const f32x4 v = a.act1[((int64_t)g * kA1Tiles + i) * 64 + l];
#pragma unroll
        for (int r = 0; r < 4; ++r) s_a1T[i * 256 + (4 * q + r) * 16 + tpos(x)] = v[r];
    }
    __syncthreads();
    f32x4 acc[2][8][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) acc[dy][kx][0] = acc[dy][kx][1] = zero4();
    f32x4 bsum = zero4();
    const f32x4* __restrict__ dz2 = a.dz2 + (int64_t)g * kA2Tiles * 64 + l;
    f32x4 vt[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) vt[px] = dz2[(int64_t)(w * 28 + px) * 64];
    const int tp = tpos(x);
    for (int blk = 0; blk < 49; ++blk) {
        const int by = blk / 7, bx = blk - by * 7;
        // the block's four tiles turned through LDS together; the next block's requested meanwhile
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int r = 0; r < 4; ++r) scratch[px * 256 + (4 * q + r) * 16 + tp] = vt[px][r];
        {
            const int nb = min(blk + 1, 48), nby = nb / 7, nbx = nb - nby * 7;
#pragma unroll
            for (int px = 0; px < 4; ++px) vt[px] = dz2[(int64_t)((4 * nby + w) * 28 + 4 * nbx + px) * 64];
        }
        wave_sync();
        f32x4 u[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            u[px] = *reinterpret_cast<const f32x4*>(scratch + px * 256 + x * 16 + 4 * q);
            bsum += u[px];
        }
        wave_sync();
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int iy = by - dy;
            if (iy < 0 || iy > 5) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ix = bx - dx;
                if (ix < 0 || ix > 5) continue;
                const int p = iy * 6 + ix;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(s_a1T + (p * 2 + 0) * 256 + x * 16 + 4 * q);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(s_a1T + (p * 2 + 1) * 256 + x * 16 + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        acc[dy][px + 4 * dx][0] = ASAC_MF(a0[r], u[px][r], acc[dy][px + 4 * dx][0]);
                        acc[dy][px + 4 * dx][1] = ASAC_MF(a1[r], u[px][r], acc[dy][px + 4 * dx][1]);
                    }
            }
        }
    }
    float* part = a.partial + (int64_t)g * kPartialFloats;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int kx = 0; kx < 8; ++kx)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                reinterpret_cast<f32x4*>(part + POFF_W2)[(((w * 2 + dy) * 8 + kx) * 2 + mt) * 64 + l] = acc[dy][kx][mt];
    float t = bsum[0] + bsum[1] + bsum[2] + bsum[3];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (q == 0) part[POFF_B2 + w * 16 + x] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// partial gradients -> parameter gradients (group order)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dec_reduce(const float* __restrict__ partial, int G, int S, const Params dst,
                                                    int accumulate) {
    // 64 entries per workgroup, four threads (group slices g = slice mod 4) per entry: the loads of an entry's 256 partials
    // in flight four ways, the four slice sums added in slice order
    __shared__ float s_sum[4][64];
    const int slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    float* out = nullptr;
    int64_t nat = 0;
    int terms = 1, term_stride = 0;     // bias sections hold one partial per wave
    if (i < POFF_B2) {
        const int sec = i < POFF_W1 ? POFF_W2 : i < POFF_WD2 ? POFF_W1 : i < POFF_WD1 ? POFF_WD2 : i < POFF_W3 ? POFF_WD1 : POFF_W3;
        const int j = i - sec, r = j & 3, l = (j >> 2) & 63, tile = j >> 8, q = l >> 4, x = l & 15, c = 4 * q + r;
        if (sec == POFF_W2) {
            const int mt = tile & 1, kx = (tile >> 1) & 7, dy = (tile >> 4) & 1, w = tile >> 5;
            out = dst.w2;
            nat = ((16 * mt + c) * 16 + x) * 64 + (w + 4 * dy) * 8 + kx;
        } else if (sec == POFF_W1) {
            const int nt = tile & 1, mt = (tile >> 1) & 1, tap = tile >> 2;
            out = dst.w1;
            nat = ((16 * mt + c) * 32 + 16 * nt + x) * 16 + tap;
        } else if (sec == POFF_WD2) {
            const int nt = tile & 3, t = (tile >> 2) & 1, p = tile >> 3;
            out = dst.wd2;
            nat = ((16 * t + c) * 4 + p) * 64 + 16 * nt + x;
        } else if (sec == POFF_WD1) {
            if (x >= S) out = nullptr; else {
            out = dst.wd1;
            nat = (16 * tile + c) * S + x; }
        } else {
            const int m = 16 * tile + x;
            if (m < 27) {
                out = dst.w3;
                nat = c * 27 + m;
            }
        }
    } else if (i < POFF_B1) {
        const int j = i - POFF_B2;
        if (j < 16) out = dst.b2, nat = j, terms = 4, term_stride = 16;
    } else if (i < POFF_BD2) {
        out = dst.b1, nat = i - POFF_B1;
    } else if (i < POFF_BD1) {
        out = dst.bd2, nat = i - POFF_BD2;
    } else if (i < POFF_B3) {
        out = dst.bd1, nat = i - POFF_BD1;
    } else {
        const int j = i - POFF_B3;
        if (j < 3) out = dst.b3, nat = j, terms = 4, term_stride = 4;
    }
    float s = 0.f;
    if (out != nullptr) {
        const float* __restrict__ src = partial + i;
        for (int g0 = slice; g0 < G; g0 += 32) {
            float v[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    v[k][t] = (g0 + 4 * k < G && t < terms) ? src[(int64_t)(g0 + 4 * k) * kPartialFloats + t * term_stride] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < terms) s += v[k][t];
        }
    }
    s_sum[slice][threadIdx.x & 63] = s;
    __syncthreads();
    if (slice == 0 && out != nullptr) {
        const int e = threadIdx.x & 63;
        const float total = ((s_sum[0][e] + s_sum[1][e]) + s_sum[2][e]) + s_sum[3][e];
        out[nat] = accumulate ? out[nat] + total : total;
    }
}

static int set_lds(const void* fn, int bytes, bool& done, const char* where) {
    if (done) return 0;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (err != hipSuccess) {
        set_error(err, where);
        return (int)err;
    }
    done = true;
    return 0;
}

}  // namespace dec
}  // namespace asac

using namespace asac;
using namespace asac::dec;

extern "C" {

int64_t asac_obs_decoder_packed_floats(void) { return kPackedFloats; }

int64_t asac_obs_decoder_saved_floats(int64_t N) {
    const int64_t G = (N + 15) / 16;
    return G * (kZ1Tiles + kH0Tiles + kA1Tiles + kA2Tiles) * 256;
}

int64_t asac_obs_decoder_workspace_floats(int64_t N) {
    const int64_t G = (N + 15) / 16;
    return G * ((int64_t)kA2Tiles * 256 + kPartialFloats);
}

static Params as_params(const asac_obs_decoder_params_t* p) {
    Params P;
    P.wd1 = p->dense1_w, P.bd1 = p->dense1_b, P.wd2 = p->dense2_w, P.bd2 = p->dense2_b;
    P.w1 = p->ct1_w, P.b1 = p->ct1_b, P.w2 = p->ct2_w, P.b2 = p->ct2_b, P.w3 = p->ct3_w, P.b3 = p->ct3_b;
    return P;
}

struct SavedView {
    f32x4 *z1, *h0, *act1, *act2;
};
static SavedView saved_view(float* saved, int64_t G) {
    SavedView v;
    v.z1 = reinterpret_cast<f32x4*>(saved);
    v.h0 = v.z1 + G * kZ1Tiles * 64;
    v.act1 = v.h0 + G * kH0Tiles * 64;
    v.act2 = v.act1 + G * kA1Tiles * 64;
    return v;
}

int asac_obs_decoder_forward(const float* state, int64_t state_stride, int64_t N, int state_size,
                             const asac_obs_decoder_params_t* params_host, float* packed, float* saved, float* frames,
                             void* stream) {
    if (!state || !params_host || !packed || !saved || !frames || N <= 0 || state_size <= 0 || state_size > 16 ||
        N > (int64_t)1 << 24)
        return bad_arg("asac_obs_decoder_forward");
    static bool lds_done = false;
    if (set_lds((const void*)k_dec_fwd3, kTailLds, lds_done, "asac_obs_decoder_forward: hipFuncSetAttribute")) return 1;
    const int64_t G = (N + 15) / 16;
    const Params P = as_params(params_host);
    hipStream_t st = as_stream(stream);
    ASAC_LAUNCH(k_dec_pack, dim3((kPackedFloats + 255) / 256), dim3(256), 0, st, P, state_size, packed);
    const SavedView sv = saved_view(saved, G);
    Fwd12Args a;
    a.x = state, a.x_stride = state_stride, a.N = (int)N, a.S = state_size, a.packed = packed;
    a.bd1 = P.bd1, a.bd2 = P.bd2, a.b1 = P.b1, a.b2 = P.b2;
    a.z1 = sv.z1, a.h0 = sv.h0, a.act1 = sv.act1, a.act2 = sv.act2;
    ASAC_LAUNCH(k_dec_fwd12, dim3((unsigned)G), dim3(kThreads), 0, st, a);
    Fwd3Args b;
    b.packed = packed, b.b3 = P.b3, b.act2 = sv.act2, b.out = frames, b.N = (int)N;
    ASAC_LAUNCH(k_dec_fwd3, dim3((unsigned)G), dim3(kThreads), kTailLds, st, b);
    return finish_launch("asac_obs_decoder_forward");
}

int asac_obs_decoder_backward(const float* state, int64_t state_stride, int64_t N, int state_size, const float* packed,
                              const float* saved, const float* frames, const float* grad_frames, float* grad_state,
                              const asac_obs_decoder_params_t* grad_params_host, int accumulate, float* workspace,
                              void* stream) {
    if (!state || !packed || !saved || !frames || !grad_frames || !grad_params_host || !workspace || N <= 0 ||
        state_size <= 0 || state_size > 16 || N > (int64_t)1 << 24)
        return bad_arg("asac_obs_decoder_backward");
    static bool lds3 = false, ldsx = false, ldsw = false;
    if (set_lds((const void*)k_dec_bwd3, kTailLds, lds3, "asac_obs_decoder_backward: hipFuncSetAttribute")) return 1;
    if (set_lds((const void*)k_dec_bwd2dx, kDxLds, ldsx, "asac_obs_decoder_backward: hipFuncSetAttribute")) return 1;
    if (set_lds((const void*)k_dec_bwd2dw, kDwLds, ldsw, "asac_obs_decoder_backward: hipFuncSetAttribute")) return 1;
    const int64_t G = (N + 15) / 16;
    hipStream_t st = as_stream(stream);
    const SavedView sv = saved_view(const_cast<float*>(saved), G);
    f32x4* dz2 = reinterpret_cast<f32x4*>(workspace);
    float* partial = workspace + G * kA2Tiles * 256;
    Bwd3Args a;
    a.packed = packed, a.act2 = sv.act2, a.out = frames, a.gout = grad_frames, a.dz2 = dz2, a.partial = partial, a.N = (int)N;
    ASAC_LAUNCH(k_dec_bwd3, dim3((unsigned)G), dim3(kThreads), kTailLds, st, a);
    Bwd2dxArgs b;
    b.x = state, b.x_stride = state_stride, b.N = (int)N, b.S = state_size, b.packed = packed;
    b.z1 = sv.z1, b.h0 = sv.h0, b.act1 = sv.act1, b.dz2 = dz2, b.gx = grad_state, b.partial = partial;
    ASAC_LAUNCH(k_dec_bwd2dx, dim3((unsigned)G), dim3(kThreads), kDxLds, st, b);
    Bwd2dwArgs c;
    c.act1 = sv.act1, c.dz2 = dz2, c.partial = partial;
    ASAC_LAUNCH(k_dec_bwd2dw, dim3((unsigned)G), dim3(kThreads), kDwLds, st, c);
    const Params D = as_params(grad_params_host);
    ASAC_LAUNCH(k_dec_reduce, dim3((kPartialFloats + 63) / 64), dim3(256), 0, st, partial, (int)G, state_size, D,
                accumulate);
    return finish_launch("asac_obs_decoder_backward");
}

}  // extern "C"
