// Fused two-layer convolution stack for gfx950 (MFMA f32): Conv2d + GELU + Conv2d + GELU over small images —
// the `simple` visual encoder of the reference (`algorithm/nn_models/layers/image_layers.py:70-80`:
// Conv2d(C,16,8,4) GELU Conv2d(16,32,4,2) GELU) that `SAC_Base.get_l_states` (sac_base.py:1117-1146) runs three
// times per train step over all B x L frames of the sampled windows (B*L = 4 608 / 9 216 frames of 3x30x30 at
// BASELINE configs 4 / 5).  One launch per pass instead of ~12 MIOpen / elementwise launches with three layout
// transposes; C ABI in include/asac_hip.h.
//
// Work decomposition: a workgroup of 4 waves owns a GROUP of G = 16 / (H2*W2) frames at a time (4 for 30x30:
// the second layer's G*H2*W2 = 16 output positions fill exactly one 16-row MFMA tile) and loops over groups.
//   stage   the group's frames into LDS by LDS-DMA (global_load_lds_dwordx4, contiguous KiB pieces: the only HBM
//           traffic, 4*C*H*W bytes a frame); the next group's frames are requested as soon as this group's are
//           consumed (forward) or into a second buffer while this group computes (backward)
//   layer 1 implicit GEMM [G*H1*W1 positions] x [C*k1*k1] x [O1 <= 16] with v_mfma_f32_16x16x4_f32; the patch
//           element of (position, k) is read straight from the staged frame through a per-k offset table, so no
//           im2col buffer exists.  Whole 16-row tiles are dealt to the waves; the last (<4) tiles are split 4 ways
//           along k and summed through LDS so that the four SIMDs finish together
//   layer 2 the same on the layer-1 activations kept in LDS: one row tile, O2 <= 32 = two column tiles, each
//           split in two along k (one (column tile, k half) per wave), weights in registers
// Training saves the two pre-activations (position-major, so rows are contiguous); the backward recomputes
// nothing but GELU and produces the parameter gradients only (frames are data, not activations):
//   dz2 = g * gelu'(z2);  dW2 += dz2^T patches(a1);  dz1 = col2im(dz2 W2) * gelu'(z1)  (a gather, no da1 buffer);
//   dW1 += dz1^T patches(x)
// with the two weight-gradient GEMMs accumulating in MFMA registers across all groups of a workgroup, written
// once as per-workgroup partials and summed in fixed order by a second kernel (no float atomics).
//
// Two things measured on MI355X shaped the code (each was worth 2-9x on a phase):
//   * integer division by a run-time divisor is ~40 instructions: every index decomposition is tabulated once
//     per workgroup in LDS (one entry per thread), and the operand constants of the MFMA loops sit in compact LDS
//     tables laid out for one 16-byte read per four steps instead of in per-lane register arrays built by long
//     unrolled setup code
//   * a branch around an MFMA makes the compiler move the accumulator between the two register files on every
//     step: MFMAs of padding tiles run unconditionally on clamped operands and are simply never stored
#include "asac_common.h"
#include "asac_gelu.h"

#include <type_traits>

namespace asac {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// a pointer into device memory as such: a load through a pointer the compiler cannot place (one selected between an LDS
// and a global address, or derived through clamps from a by-value argument struct) becomes a FLAT load — it counts
// against both wait counters, so the LDS-only barriers of these kernels wait for it, and `s_waitcnt vmcnt(0)` in front of
// its use also drains an LDS-DMA prefetch issued before it (k_conv2_bwd's first phase: a third of the launch, round 6)
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* as_global(const T* p) {
    return (const __attribute__((address_space(1))) T*)p;
}

constexpr int kConvThreads = 256;              // 4 waves
constexpr int kConvMaxK = ASAC_CONV2_MAX_K;     // patch length of either layer (C*k1*k1, O1*k2*k2)
constexpr int kConvBwdGroupsCap = 256;          // workgroups of the backward per resident workgroup of a CU (each writes one partial slab)

struct ConvDims {
    int C, H, W, CHW;
    int O1, k1, s1, H1, W1, M1, K1;
    int O2, k2, s2, H2, W2, M2, K2;
    int G, rows1, RT1;                          // frames per group, layer-1 positions per group, their 16-row tiles
    // TILED mode — frames whose second-layer map has more than 16 positions (84 x 84 frames of the reference's
    // environments: 20 x 20 -> 9 x 9): the unit of work is a "virtual frame", the crop of a frame that ONE block of
    // bh x bw <= 16 second-layer positions needs (C, H, W, H1 ... M2 above then describe the crop and its maps, G = 1),
    // `tiles` of them per frame.  Blocks tile the map; a last block that would stick out is moved back inside and its
    // rows / columns already covered by its neighbour are masked (`skip`), so every position is produced exactly once —
    // the crops overlap (layer 1 is recomputed on the overlap: 1.44x at 84 x 84 with 3 x 3 blocks), the gradients are
    // sums over positions and stay exact.  tiles == 1: the frame itself (everything below equals the fields above).
    int tiles, nbx, bh, bw;
    int FH, FW, FHW, FCHW, FH2, FW2, FM2;       // the real frame, its plane / frame strides, its second-layer map
    int crop4, cw4;                             // float4s of a crop, of a crop row
    // FORWARD only, block-row mode (conv_block_rows): the unit of work is a whole ROW of blocks — one crop of full-width
    // frame rows (contiguous in memory), layer 1 once over the union of the blocks' first-layer regions (H1 x the frame's
    // whole first-layer width: C, H, W, W1, M1, rows1, RT1 above describe that union, tiles = block rows per frame,
    // nbx = 1), then layer 2 and the epilogue block by block (`sub` blocks of width bw; M2, H2, W2 stay the block's).
    // sub == 1: one block per unit.  The first-layer pre-activations are then saved per UNIT (slab (frame tiles + block row)
    // of H1 x zW1 positions); the backward, which stays block by block, picks its block's columns out of the slab
    // (zW1 != 0 in ITS dims says so: the union's first-layer width).
    int sub, zW1;
};

struct ConvArgs {
    ConvDims d;
    const float* x;                             // [N][C][H][W]
    const float* w1; const float* b1;           // [O1][C][k1][k1], [O1]
    const float* w2; const float* b2;           // [O2][O1][k2][k2], [O2]
    float* y;                                   // [N][O2*M2]  (channel-major like a flattened NCHW map)
    float* z1;                                  // [N][M1][O1] pre-activations (position-major) or NULL
    float* z2;                                  // [N][O2*M2] pre-activations or NULL
    const float* gy;                            // backward: gradient of y
    const float* gy_more[3];                    // ... of cotangents 1 .. NC - 1 (k_conv2_bwd<., NC>: several backward walks of ONE
                                                //     forward pass as one launch; slab c of a block's partials behind slab c - 1)
    float* partial;                             // backward: [blocks][param_count]
    int64_t N, n_groups;
    // forward over a SLICE of sampled windows (x[:, b:] of [B][L][C][H][W], read in place): a sample's frames are
    // consecutive, samples x_sample_stride floats apart, x_sample_groups whole groups per sample (0: dense [N])
    int64_t x_sample_stride;
    int32_t x_sample_groups;
};

// first frame of group g in memory (tiled mode: g = the frame's index, G = 1)
__device__ __forceinline__ const float* group_frames(const ConvArgs& a, int64_t g) {
    if (a.x_sample_groups == 0) return a.x + g * a.d.G * a.d.FCHW;
    const int64_t s = g / a.x_sample_groups;
    return a.x + s * a.x_sample_stride + (g - s * a.x_sample_groups) * a.d.G * a.d.FCHW;
}

// where group g's outputs live: first frame, the block's origin in the frame's second-layer map and the rows / columns of
// the block that a neighbour already produced (tiled mode; otherwise frame g * G, origin (0, 0), nothing skipped)
struct TileAt {
    int64_t frame;
    int r0, c0, skip_y, skip_x, by;             // (by: the block's row of blocks)
};
template <bool TILED>
__device__ __forceinline__ TileAt tile_at(const ConvDims& d, int64_t g) {
    TileAt t;
    if (!TILED) {
        t.frame = g * d.G, t.r0 = t.c0 = t.skip_y = t.skip_x = t.by = 0;
        return t;
    }
    t.frame = g / d.tiles;
    const int tile = (int)(g - t.frame * d.tiles), by = tile / d.nbx, bx = tile - by * d.nbx;
    t.r0 = min(by * d.bh, d.FH2 - d.bh), t.c0 = min(bx * d.bw, d.FW2 - d.bw);
    t.skip_y = by * d.bh - t.r0, t.skip_x = bx * d.bw - t.c0, t.by = by;
    return t;
}

// LDS plan (floats).  fwd: frames | a1 | index tables | reduction slabs
struct ConvFwdPlan { int img, a1, ktab, koff2, rowx, rowa, ktq, w1q, red, total; };
__host__ __device__ inline ConvFwdPlan conv_fwd_plan(const ConvDims& d) {
    ConvFwdPlan p;
    int off = 0;
    auto take = [&](int n) { const int o = off; off += (n + 3) & ~3; return o; };
    p.ktab = take(kConvMaxK);
    p.koff2 = take(kConvMaxK);
    p.rowx = take(d.RT1 * 16);
    p.rowa = take(d.RT1 * 16);
    p.ktq = take(d.K1);
    p.w1q = take(d.K1 * 16);
    off = (off + 255) & ~255;
    p.img = take((d.G * d.CHW + 255) & ~255);   // whole KiB: the DMA path writes 1 KiB pieces
    p.a1 = take(d.G * d.O1 * d.M1);
    const int rem = d.RT1 % 4;
    p.red = take(4 * 256 * (rem > 1 ? rem : 1));
    // at kernel start all filters pass through the work area (from `img` on) on their way into registers
    const int params = p.img + d.O1 * d.K1 + d.O2 * d.K2;
    p.total = off > params ? off : params;
    return p;
}

// patch offset of reduction index k of a [Cin][kk][kk] filter inside a [Cin][plane_h][plane_w] map
__device__ __forceinline__ int patch_offset(int k, int kk, int plane, int plane_w) {
    const int c = k / (kk * kk), rem = k - c * kk * kk, ky = rem / kk, kx = rem - ky * kk;
    return c * plane + ky * plane_w + kx;
}

// the group's frames -> LDS (zero beyond the last frame of the batch)
__device__ __forceinline__ void stage_frames(const ConvArgs& a, int64_t g, float* img) {
    const ConvDims& d = a.d;
    const int64_t first = g * d.G;
    const int n_img = (int)min((int64_t)d.G, a.N - first);
    const int count = n_img * d.CHW, total = d.G * d.CHW;
    const float* src = group_frames(a, g);
    if ((d.CHW & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.x_sample_stride & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(img);
        const int c4 = count >> 2, t4 = total >> 2;
        constexpr int NB = 6;
        for (int base = 0; base < t4; base += kConvThreads * NB) {
            float4 v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int i = base + u * kConvThreads + (int)threadIdx.x;
                v[u] = i < c4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int i = base + u * kConvThreads + (int)threadIdx.x;
                if (i < t4) d4[i] = v[u];
            }
        }
    } else {
        for (int i = threadIdx.x; i < total; i += kConvThreads) img[i] = i < count ? src[i] : 0.f;
    }
}

// LDS-DMA copy (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave instruction, no staging registers, the
// wave keeps running): `chunks` KiB from src (16-byte aligned, `n4` float4 readable; the tail clamps) to dst
__device__ __forceinline__ void async_copy_kib(const float* src, float* dst, int n4, int chunks, int wave, int lane) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int c = wave; c < chunks; c += kConvThreads / 64) {
        const int i4 = min(c * 64 + lane, n4 - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s4 + i4),
                                         (__attribute__((address_space(3))) void*)(dst + c * 256), 16, 0, 0);
    }
}

// Marks a value as produced here: the wait for the global load behind it is paid once at this point.  Without it
// the compiler, which cannot see that the setup loads completed long ago, guards every later use inside the
// group loop with `s_waitcnt vmcnt(0)` — which also drains the stores and the DMA prefetch then in flight.
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void settle(int& v) { asm volatile("" : "+v"(v)); }

// tiled mode: float4 i of a crop [C][H][W] sits `off` floats behind the crop's first pixel in the frame.  A thread moves the
// same float4s of every crop (piece j of wave w: i = (w + 4 j) 64 + lane), so its offsets live in registers: a table in LDS
// cost the backward its second workgroup per CU (87 KB with it)
constexpr int kCropPieces = 8;                                  // per wave: a block's crop up to 4 x 8 KiB (conv_dims checks)
constexpr int kConvZPieces = 2;                                 // per wave: a block's z1 rows up to 4 x 2 KiB (conv_block_rows checks)
constexpr int kCropPiecesRow = 10;                              // ... the forward's crop of a row of blocks up to 4 x 10 KiB
constexpr int kCropMaxFloats = kCropPieces * (kConvThreads / 64) * 256;
constexpr int kCropMaxFloatsRow = kCropPiecesRow * (kConvThreads / 64) * 256;
template <int P>
struct CropOffsets { int off[P]; };
template <int P>
__device__ __forceinline__ CropOffsets<P> crop_offsets(const ConvDims& d, int wave, int lane) {
    CropOffsets<P> t;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int i = min((wave + (kConvThreads / 64) * j) * 64 + lane, max(d.crop4 - 1, 0));
        const int hw4 = max(d.H * d.cw4, 1), cw4 = max(d.cw4, 1);
        const int c = i / hw4, rem = i - c * hw4, y = rem / cw4, x4 = rem - y * cw4;
        t.off[j] = c * d.FHW + y * d.FW + 4 * x4;
        settle(t.off[j]);
    }
    return t;
}

// frames of group g -> LDS through the DMA path; frames beyond the batch re-read the last real one (their results
// are never stored and their gradients are zero)
template <bool TILED = false, int P = kCropPieces>
__device__ __forceinline__ void async_frames(const ConvArgs& a, int64_t g, float* img, int wave, int lane,
                                             const CropOffsets<P>* crop = nullptr, const TileAt* at = nullptr) {
    const ConvDims& d = a.d;
    if (TILED) {
        // the crop of frame g / tiles that block g % tiles needs: a lane's 16 bytes come from wherever its offsets say
        const TileAt t = at ? *at : tile_at<true>(d, g);
        const float* src = group_frames(a, t.frame) + (int64_t)(t.r0 * d.s2 * d.s1) * d.FW + t.c0 * d.s2 * d.s1;
        const int chunks = (d.CHW + 255) >> 8;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int c = wave + (kConvThreads / 64) * j;
            if (c < chunks)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + crop->off[j]),
                                                 (__attribute__((address_space(3))) void*)(img + c * 256), 16, 0, 0);
        }
        return;
    }
    const int64_t first = g * d.G;
    const int n_img = (int)min((int64_t)d.G, a.N - first);
    async_copy_kib(group_frames(a, g), img, (n_img * d.CHW) >> 2, (d.G * d.CHW + 255) >> 8, wave, lane);
}

// Two parameter arrays global -> LDS (dst, then dst + nA) by the whole workgroup: each element is fetched once per
// workgroup, coalesced, with ALL loads of both arrays in flight before the first LDS store (one memory round
// trip), and only then spread into the lanes' registers from LDS — per-lane strided reads of the same few KB by
// every wave of the grid pile up on a handful of L2 channels.
__device__ __forceinline__ void coop_copy2(const float* __restrict__ srcA, int nA, const float* __restrict__ srcB,
                                           int nB, float* dst) {
    const bool vec = ((nA | nB) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(srcA) | reinterpret_cast<uintptr_t>(srcB)) & 15) == 0;
    if (vec) {
        const float4* a4 = reinterpret_cast<const float4*>(srcA);
        const float4* b4 = reinterpret_cast<const float4*>(srcB);
        float4* d4 = reinterpret_cast<float4*>(dst);
        const int na = nA >> 2, n = (nA + nB) >> 2;
        constexpr int NB = 12;
        for (int base = 0; base < n; base += NB * kConvThreads) {
            float4 v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int i = base + u * kConvThreads + (int)threadIdx.x;
                v[u] = i < na ? a4[i] : (i < n ? b4[i - na] : make_float4(0.f, 0.f, 0.f, 0.f));
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int i = base + u * kConvThreads + (int)threadIdx.x;
                if (i < n) d4[i] = v[u];
            }
        }
    } else {
        for (int i = threadIdx.x; i < nA; i += kConvThreads) dst[i] = srcA[i];
        for (int i = threadIdx.x; i < nB; i += kConvThreads) dst[nA + i] = srcB[i];
    }
}

// workgroup barrier that only drains this wave's LDS traffic (an LDS-DMA prefetch stays in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and the one that makes the prefetched data visible: every wave's DMA landed
__device__ __forceinline__ void dma_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Integer division by a run-time divisor costs ~40 instructions on this ISA: every index decomposition
// (reduction index -> (channel, ky, kx), position row -> (frame, oy, ox)) is done ONCE per workgroup into small
// LDS tables (one entry per thread), never per element.

// layer-1 tile: 16 positions (this lane's A row starts at `base`) over the reduction quads [q0, q1) — a quad is four
// MFMA steps = 16 reduction indices.  The lane's operand constants come from two LDS tables laid out so that one
// 16-byte read serves a quad:  ktq[quad][lk][4] patch offsets,  w1q[quad][lane][4] weights (B operand).
// The next quad's constants are requested before this quad's MFMAs are issued.
__device__ __forceinline__ f32x4 conv1_tile(const float* base, const int* ktq, const float* w1q, int q0, int q1) {
    const int lane = threadIdx.x & 63, lk = lane >> 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int4* kt = reinterpret_cast<const int4*>(ktq) + lk;
    const float4* wt = reinterpret_cast<const float4*>(w1q) + lane;
    if (q0 >= q1) return acc0;
    int4 ko = kt[q0 * 4];
    float4 w = wt[q0 * 64];
    for (int q = q0; q < q1; ++q) {
        const float a0 = base[ko.x], a1v = base[ko.y], a2 = base[ko.z], a3 = base[ko.w];
        const float4 wc = w;
        const int qn = min(q + 1, q1 - 1);         // the next quad's constants travel under this quad's MFMAs
        ko = kt[qn * 4];
        w = wt[qn * 64];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wc.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, wc.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, wc.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, wc.w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// ... with the operand constants of ALL quads in registers (Q1C compile-time quads: 8 VGPRs each): per quad the LDS then
// serves the four patch reads only — the two 16-byte table reads were two thirds of the layer's LDS bytes
template <int Q1C>
__device__ __forceinline__ f32x4 conv1_tile_reg(const float* base, const int4 (&ko)[Q1C], const float4 (&w)[Q1C]) {
    // (requesting the patch elements two quads ahead of their MFMAs, pinned with scheduling barriers, changed nothing —
    // 9.69 -> 9.56 k clocks a group, round 6: two workgroups a CU share each SIMD's matrix pipe, whose 108 + 32 MFMAs per
    // wave and group are ~60 % of the group's time; the reads already travel under the other workgroup's MFMAs)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < Q1C; ++q) {
        const float a0 = base[ko[q].x], a1v = base[ko[q].y], a2 = base[ko[q].z], a3 = base[ko[q].w];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, w[q].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w[q].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w[q].w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// ... and a wave's quarter of the reduction of a tail tile (quads [Q1C wave / 4, Q1C (wave + 1) / 4), the accumulator
// pattern of conv1_tile: the same bits) from the same registers — through the LDS tables each of its quads was two
// dependent round trips (constants, then patch elements) in front of four MFMAs
template <int Q1C, int QA, int QB>
__device__ __forceinline__ f32x4 conv1_quads_reg(const float* base, const int4 (&ko)[Q1C], const float4 (&w)[Q1C]) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = QA; q < QB; ++q) {
        const float a0 = base[ko[q].x], a1v = base[ko[q].y], a2 = base[ko[q].z], a3 = base[ko[q].w];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v, w[q].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, w[q].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, w[q].w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}
template <int Q1C>
__device__ __forceinline__ f32x4 conv1_tail_reg(const float* base, const int4 (&ko)[Q1C], const float4 (&w)[Q1C], int wave) {
    switch (wave) {                                 // (uniform: a scalar branch)
        case 0: return conv1_quads_reg<Q1C, 0, Q1C / 4>(base, ko, w);
        case 1: return conv1_quads_reg<Q1C, Q1C / 4, Q1C / 2>(base, ko, w);
        case 2: return conv1_quads_reg<Q1C, Q1C / 2, 3 * Q1C / 4>(base, ko, w);
        default: return conv1_quads_reg<Q1C, 3 * Q1C / 4, Q1C>(base, ko, w);
    }
}

// bias + GELU of one layer-1 element; keeps the activation in LDS (channel-major, what layer 2's patches index)
// and, when training, the pre-activation in HBM (position-major: frame*M1 + pos == group row).  `abase` = the
// row's a1 offset frame*O1*M1 + pos (negative beyond the group's positions), `z_rows` = rows backed by real frames
__device__ __forceinline__ void conv1_store(const ConvArgs& a, int64_t g, int row, int abase, int z_rows, int oc, float z,
                                            float act, float* a1) {
    const ConvDims& d = a.d;
    if (abase < 0 || oc >= d.O1) return;
    a1[abase + oc * d.M1] = act;
    if (a.z1 && row < z_rows) a.z1[(g * d.rows1 + row) * d.O1 + oc] = z;
}

__device__ __forceinline__ void conv1_finish(const ConvArgs& a, int64_t g, int row, int abase, int z_rows, int oc,
                                             float sum, float bias, float* a1) {
    const float z = sum + bias;
    conv1_store(a, g, row, abase, z_rows, oc, z, gelu_f(z), a1);
}

// phase clocks of workgroup 0 (thread 0) summed over its groups, for tools/debug/conv_bwd_phases.py; compiled out of the library
#ifdef ASAC_CONV_STAMPS
__device__ unsigned long long g_conv_stamps[16];
#define CONV_STAMP_INIT unsigned long long st_last = __builtin_readcyclecounter(), st_acc[10] = {}
#define CONV_STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[k] += t_ - st_last; st_last = t_; } while (0)
#define CONV_STAMP_FLUSH do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int k_ = 0; k_ < 10; ++k_) g_conv_stamps[k_] = st_acc[k_]; } while (0)
extern "C" int asac_debug_conv_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamps), sizeof(unsigned long long) * 16);
}
#else
#define CONV_STAMP_INIT
#define CONV_STAMP(k)
#define CONV_STAMP_FLUSH
#endif

// (TILED: a second instantiation — the whole-frame form keeps the code it had before the tiled mode existed: with the
// block addressing compiled in, its launches were 0.5-2.8 us longer at cfg4's sizes)
template <bool TILED, int Q1C = 0>
__global__ __launch_bounds__(kConvThreads) void k_conv2_fwd(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvDims& d = a.d;
    const ConvFwdPlan p = conv_fwd_plan(d);
    float* img = lds + p.img;
    float* a1 = lds + p.a1;
    int* ktab = reinterpret_cast<int*>(lds + p.ktab);
    int* koff2 = reinterpret_cast<int*>(lds + p.koff2);
    int* rowx = reinterpret_cast<int*>(lds + p.rowx);
    int* rowa = reinterpret_cast<int*>(lds + p.rowa);
    int* ktq = reinterpret_cast<int*>(lds + p.ktq);
    float* w1q = lds + p.w1q;
    float* red = lds + p.red;
    const int n_sub = TILED ? d.sub : 1;                       // (block-row mode: the blocks of a row share crop and layer 1)
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    CropOffsets<kCropPiecesRow> crop_r{};
    if (TILED) crop_r = crop_offsets<kCropPiecesRow>(d, wave, lane);
    const CropOffsets<kCropPiecesRow>* crop = &crop_r;
    const int rows_pad = d.RT1 * 16;
    // biases are requested first: their latency hides under the table builds instead of in front of the first DMA
    const float b1v = lr < d.O1 ? a.b1[lr] : 0.f;
    float b1e = (int)(threadIdx.x & 15) < d.O1 ? a.b1[threadIdx.x & 15] : 0.f;     // tail-tile element's bias
    float e_bias[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int oc = (threadIdx.x + q * kConvThreads) & 31;
        e_bias[q] = oc < d.O2 ? a.b2[oc] : 0.f;
    }

    // DIRECT (16-byte aligned filters): every lane reads the weights of its MFMA B operands where they lie, 16 bytes at a
    // time — the reduction index a (quad q, lane group lk, step u) of layer 1 stands for is 16 q + 4 lk + u, i.e. four
    // CONSECUTIVE filter taps per lane (the order of a reduction is free as long as both operands agree on it), and
    // likewise k = 128 kh + 16 j + 4 lk + t for layer 2 — so a wave's load covers 64-byte pieces of 16 filter rows, nothing
    // passes through the frame buffer, and the first group's frames are requested at kernel entry: all of it travels under
    // the index-table builds.  (Round 3 / 5 read single taps per lane, 768 bytes apart: the grid's waves queued on a
    // handful of L2 channels, +3.5 us; staged through the frame buffer the set-up ended 10.3 us after entry at 4 608
    // frames.)  Unaligned filters (a plugin with an odd parameter in front of them) take the staged path.
    const bool direct = ((reinterpret_cast<uintptr_t>(a.w1) | reinterpret_cast<uintptr_t>(a.w2)) & 15) == 0;
    const bool dma = TILED ||
                     ((d.CHW & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (a.x_sample_stride & 3) == 0);
    const int Q1 = d.K1 / 16;
    // layer-2 weights of this wave's (column tile, k half): B operand element (k = 4 step + lk, column lr)
    const int ct = wave >> 1, kh = wave & 1;
    constexpr int S2H = kConvMaxK / 8;           // steps per k half
    float w2r[S2H];
    const int oc2 = ct * 16 + lr;
    constexpr int QR = Q1C > 0 ? Q1C : 1;
    int4 ko_r[QR];
    float4 w_r[QR];
    if (direct) {
        if (dma && (int64_t)blockIdx.x < a.n_groups) async_frames<TILED, kCropPiecesRow>(a, blockIdx.x, img, wave, lane, crop);
        const auto* w2p = as_global(reinterpret_cast<const f32x4*>(a.w2 + (int64_t)min(oc2, d.O2 - 1) * d.K2));
#pragma unroll
        for (int j = 0; j < S2H / 4; ++j) {
            const int k = kh * (4 * S2H) + 16 * j + 4 * lk;
            const f32x4 v = w2p[min(k, d.K2 - 4) >> 2];
            const bool on = k < d.K2 && oc2 < d.O2;
            w2r[4 * j] = on ? v.x : 0.f, w2r[4 * j + 1] = on ? v.y : 0.f, w2r[4 * j + 2] = on ? v.z : 0.f, w2r[4 * j + 3] = on ? v.w : 0.f;
        }
        const auto* w1p = as_global(reinterpret_cast<const f32x4*>(a.w1 + (int64_t)min(lr, d.O1 - 1) * d.K1) + lk);
        auto load4 = [&](int i, bool on) {
            const f32x4 v = w1p[i];
            return on ? make_float4(v.x, v.y, v.z, v.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        if (Q1C > 0) {
#pragma unroll
            for (int q = 0; q < QR; ++q) w_r[q] = load4(q * 4, lr < d.O1);
        }
        // (the LDS table of the tail tiles / of the quad counts without a register form: quad q by wave q % 4)
        float4 w1t[kConvMaxK / 64];
#pragma unroll
        for (int i = 0; i < kConvMaxK / 64; ++i) {
            const int q = wave + 4 * i;
            w1t[i] = load4(min(q, Q1 - 1) * 4, q < Q1 && lr < d.O1);
        }
        for (int k = threadIdx.x; k < kConvMaxK; k += kConvThreads) {
            ktab[k] = k < d.K1 ? patch_offset(k, d.k1, d.H * d.W, d.W) : 0;
            // (entry 4 (kh S2H + s) + lk, what layer 2 reads for step s, stands for k = 128 kh + 16 (s / 4) + 4 lk + s % 4)
            const int s2 = (k >> 2) & (S2H - 1), k2 = (k >> 7) * (4 * S2H) + 16 * (s2 >> 2) + 4 * (k & 3) + (s2 & 3);
            koff2[k] = k2 < d.K2 ? patch_offset(k2, d.k2, d.M1, d.W1) : 0;
        }
        for (int row = threadIdx.x; row < rows_pad; row += kConvThreads) {
            const int rr = min(row, d.rows1 - 1);          // the tail tile re-reads a valid position
            const int im = rr / d.M1, pos = rr - im * d.M1, oy = pos / d.W1, ox = pos - oy * d.W1;
            rowx[row] = im * d.CHW + d.s1 * oy * d.W + d.s1 * ox;
            rowa[row] = row < d.rows1 ? im * d.O1 * d.M1 + pos : -1;
        }
#pragma unroll
        for (int i = 0; i < kConvMaxK / 64; ++i) {
            const int q = wave + 4 * i;
            if (q < Q1) reinterpret_cast<float4*>(w1q)[q * 64 + lane] = w1t[i];
        }
        __syncthreads();           // ktab is complete
        for (int i = threadIdx.x; i < Q1 * 16; i += kConvThreads) ktq[i] = ktab[i];     // (q, lk, u) <-> k = 16 q + 4 lk + u
    } else {
    for (int k = threadIdx.x; k < kConvMaxK; k += kConvThreads) {
        ktab[k] = k < d.K1 ? patch_offset(k, d.k1, d.H * d.W, d.W) : 0;
        koff2[k] = k < d.K2 ? patch_offset(k, d.k2, d.M1, d.W1) : 0;
    }
    for (int row = threadIdx.x; row < rows_pad; row += kConvThreads) {
        const int rr = min(row, d.rows1 - 1);          // the tail tile re-reads a valid position
        const int im = rr / d.M1, pos = rr - im * d.M1, oy = pos / d.W1, ox = pos - oy * d.W1;
        rowx[row] = im * d.CHW + d.s1 * oy * d.W + d.s1 * ox;
        rowa[row] = row < d.rows1 ? im * d.O1 * d.M1 + pos : -1;
    }
    // parameters pass through the (still free) work area: W1 | W2
    coop_copy2(a.w1, d.O1 * d.K1, a.w2, d.O2 * d.K2, img);
    __syncthreads();
    // layer-1 operand tables (see conv1_tile): element (quad, lane, u) <-> reduction index k = 16 quad + 4 u + lk
    for (int i = threadIdx.x; i < Q1 * 256; i += kConvThreads) {
        const int u = i & 3, ln = (i >> 2) & 63, q = i >> 8;
        const int k = 16 * q + 4 * u + (ln >> 4), oc = ln & 15;
        w1q[i] = oc < d.O1 ? img[oc * d.K1 + k] : 0.f;
    }
    for (int i = threadIdx.x; i < Q1 * 16; i += kConvThreads) {
        const int u = i & 3, lkk = (i >> 2) & 3, q = i >> 4;
        ktq[i] = ktab[16 * q + 4 * u + lkk];
    }
    {
        const float* w2l = img + d.O1 * d.K1 + min(oc2, d.O2 - 1) * d.K2;
#pragma unroll
        for (int s = 0; s < S2H; ++s) {
            const int k = 4 * (kh * S2H + s) + lk;
            const float v = w2l[min(k, d.K2 - 1)];
            w2r[s] = (k < d.K2 && oc2 < d.O2) ? v : 0.f;
        }
    }
    }
    __syncthreads();               // the work area is free again
    const int full = d.RT1 & ~3, rem = d.RT1 & 3;
    // layer 2: this lane's A row (position lr of the 16) and the two output elements this thread finishes
    int base2;
    {
        const int lrc = TILED ? min(lr, d.G * d.M2 - 1) : lr;      // (rows beyond a block's positions repeat the last)
        const int im = lrc / d.M2, pos = lrc - im * d.M2, oy = pos / d.W2, ox = pos - oy * d.W2;
        base2 = im * d.O1 * d.M1 + d.s2 * oy * d.W1 + d.s2 * ox;
    }
    // the two output elements this thread finishes: frame within the group, channel offset (-1: none), position (y, x)
    // inside the group's block of the map (the whole map unless tiled)
    int e_im[2], e_out[2], e_py[2], e_px[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = threadIdx.x + q * kConvThreads, row = e >> 5, oc = e & 31;
        e_im[q] = row / d.M2;
        const int pos = row - e_im[q] * d.M2;
        e_py[q] = pos / d.W2, e_px[q] = pos - e_py[q] * d.W2;
        e_out[q] = oc < d.O2 ? (TILED ? oc * d.FM2 : oc * d.M2 + pos) : -1;
    }
    float b1s = b1v;
    const int qa = (Q1 * wave) / 4, qb = (Q1 * (wave + 1)) / 4;    // this wave's share of a split tail tile
#pragma unroll
    for (int s = 0; s < S2H; ++s) settle(w2r[s]);
    settle(e_bias[0]); settle(e_bias[1]); settle(b1e); settle(b1s);
    // (Q1C > 0: the full tiles' operand constants, read from the tables once; the weights came straight from memory where
    // the filters are aligned)
    if (Q1C > 0) {
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            ko_r[q] = (reinterpret_cast<const int4*>(ktq) + lk)[q * 4];
            if (!direct) w_r[q] = (reinterpret_cast<const float4*>(w1q) + lane)[q * 64];
            settle(w_r[q].x); settle(w_r[q].y); settle(w_r[q].z); settle(w_r[q].w);
        }
    }

    // frames travel by LDS-DMA when they are 16-byte granular: the next group's are requested as soon as layer 1 has
    // consumed this group's, and land while layer 2 and the epilogues run
    // (tiled mode: the host has checked the 16-byte granularity the crops need)
    if (!direct && dma && (int64_t)blockIdx.x < a.n_groups) async_frames<TILED, kCropPiecesRow>(a, blockIdx.x, img, wave, lane, crop);
    CONV_STAMP_INIT;
    for (int64_t g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
        CONV_STAMP(9);
        const int n_img = (int)min((int64_t)d.G, a.N - g * d.G);
        const TileAt at = tile_at<TILED>(d, g);
        const int z_rows = n_img * d.M1;
        if (dma) {
            dma_barrier();
        } else {
            stage_frames(a, g, img);
            __syncthreads();
        }
        CONV_STAMP(0);
        // ---- layer 1 ------------------------------------------------------------------------------------
        f32x4 tail[3];
#pragma unroll
        for (int u = 0; u < 3; ++u)     // the last tiles: every wave takes a quarter of the reduction of each
            if (u < rem)
                tail[u] = Q1C > 0 ? conv1_tail_reg<QR>(img + rowx[(full + u) * 16 + lr], ko_r, w_r, wave)
                                  : conv1_tile(img + rowx[(full + u) * 16 + lr], ktq, w1q, qa, qb);
        for (int t = wave; t < full; t += 4) {
            const f32x4 acc = Q1C > 0 ? conv1_tile_reg<QR>(img + rowx[t * 16 + lr], ko_r, w_r)
                                      : conv1_tile(img + rowx[t * 16 + lr], ktq, w1q, 0, Q1);
            f32x2_g ya, yb, unused;       // the fragment's four elements as two packed pairs
            gelu_parts2((f32x2_g){acc[0] + b1s, acc[1] + b1s}, ya, unused);
            gelu_parts2((f32x2_g){acc[2] + b1s, acc[3] + b1s}, yb, unused);
            const float yv[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + 4 * lk + r;
                conv1_store(a, g, row, rowa[row], z_rows, lr, acc[r] + b1s, yv[r], a1);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u < rem) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(u * 4 + wave) * 256 + (4 * lk + r) * 16 + lr] = tail[u][r];
            }
        CONV_STAMP(7);
        lds_barrier();             // the frames are consumed
        CONV_STAMP(1);
        // (a crop's six requests per wave take ~1 700 clocks to issue — the CU's 64 B / clock address path shared with the
        // other workgroup's, whose MFMAs run meanwhile; spreading them over the loop's later phases only moved the stall:
        // 103 -> 108 us at 1 024 frames of 84 x 84, round 6)
        if (dma && g + gridDim.x < a.n_groups) async_frames<TILED, kCropPiecesRow>(a, g + gridDim.x, img, wave, lane, crop);
        CONV_STAMP(5);
        for (int u = 0; u < rem; ++u) {
            const float* ru = red + u * 4 * 256;
            const int e = threadIdx.x;                 // element (row e/16, channel e%16) of the tile
            const float sum = ((ru[e] + ru[256 + e]) + ru[512 + e]) + ru[768 + e];
            const int row = (full + u) * 16 + (e >> 4);
            conv1_finish(a, g, row, rowa[row], z_rows, e & 15, sum, b1e, a1);
        }
        CONV_STAMP(6);
        lds_barrier();
        CONV_STAMP(2);

        // ---- layer 2: 16 positions (G frames x M2), wave = (column tile, k half) ---------------------------
        // (block-row mode: once per block of the row, on the shared activations: the block's first column moves the window)
        for (int sb = 0; sb < n_sub; ++sb) {
            const int sb_c0 = TILED && n_sub > 1 ? min(sb * d.bw, d.FW2 - d.bw) : 0;      // the block's first output column
            {
                const float* base = a1 + base2 + sb_c0 * d.s2;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                float av[S2H];
#pragma unroll
                for (int s = 0; s < S2H; ++s) av[s] = base[koff2[4 * (kh * S2H + s) + lk]];
#pragma unroll
                for (int s = 0; s < S2H; s += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], w2r[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s + 1], w2r[s + 1], acc1, 0, 0, 0);
                }
                const f32x4 acc = acc0 + acc1;
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * lk + r) * 16 + lr] = acc[r];
            }
            CONV_STAMP(8);
            lds_barrier();
            CONV_STAMP(3);
            const int at_c0 = TILED && n_sub > 1 ? sb_c0 : at.c0;
            const int at_skip_x = TILED && n_sub > 1 ? sb * d.bw - sb_c0 : at.skip_x;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                       // (position row, output channel) = e / 32, e % 32
                const int e = threadIdx.x + q * kConvThreads, row = e >> 5, oc = e & 31, c2 = oc >> 4;
                if (e_out[q] >= 0 && e_im[q] < n_img && (!TILED || (e_py[q] >= at.skip_y && e_px[q] >= at_skip_x))) {
                    const int i = row * 16 + (oc & 15);
                    const float z = (red[(2 * c2) * 256 + i] + red[(2 * c2 + 1) * 256 + i]) + e_bias[q];
                    const int64_t o = TILED ? (at.frame + e_im[q]) * ((int64_t)d.O2 * d.FM2) + e_out[q] +
                                                  (at.r0 + e_py[q]) * d.FW2 + at_c0 + e_px[q]
                                            : (g * d.G + e_im[q]) * (d.O2 * d.M2) + e_out[q];
                    a.y[o] = gelu_f(z);
                    if (a.z2) a.z2[o] = z;
                }
            }
            if (sb + 1 < n_sub) lds_barrier();                 // (the slabs are read: the next block's sums may land)
        }
        // (the next iteration's first barrier orders these reads before the slabs / activations are rewritten)
        CONV_STAMP(4);
    }
    CONV_STAMP_FLUSH;
}

// ------------------------------------------------------------------------------------------------
// Backward: parameter gradients only.
// LDS: frames (two buffers) | raw z1 rows | gelu'(z1) -> dz1 (position-major) | a1 (channel-major), later the product tile
//      dz2 W2 | dz1 of further cotangents | dz2 [16][32] | position offsets, col2im contributions.  Per-lane constants (patch
//      offsets of the lane's gradient columns, the W2 elements of its column tiles) live in registers; 128 KB for 30 x 30
//      frames (one workgroup per CU), 79 KB for the crops of 84 x 84 frames (two)
// ------------------------------------------------------------------------------------------------
struct ConvBwdPlan { int img, img_size, bufs, zraw, g1, a1, dp_pitch, dz1x, dz1_size, dz2, rowoff1, rowa, cont, n_cont, ktab, red, total; };
// positions of the second layer's map that one first-layer position feeds, per axis: ceil(k2 / s2)
__host__ __device__ inline int conv_reach(const ConvDims& d) { return (d.k2 + d.s2 - 1) / d.s2; }
// nc cotangents (k_conv2_bwd<., NC>): one dz2 buffer each; dz1 = col2im(dz2 W2) * gelu'(z1) is formed position-major
// ([row][16 channels]: conflict-free A operand reads) — cotangent 0 in place of gelu'(z1), which every cotangent needs
// and which therefore goes last, the others in buffers of their own (30 x 30: 128 KB + 11.2 KB per further cotangent).
// The product tile dz2 W2 [16 positions][K2] passes through LDS as [position][ky k2 + kx][17: channel] (pitch dp_pitch,
// rows 4 apart 16 banks apart: the gather's reads — a channel per lane — and, for 4 x 4 patches, the fragments' stores at
// the minimum of two lanes per bank) in the place of a1, which is dead by then; the setup's patch-offset table lives
// there too.
__host__ __device__ inline ConvBwdPlan conv_bwd_layout(const ConvDims& d, int nc, int bufs) {
    ConvBwdPlan p;
    int off = 0;
    auto take = [&](int n) { const int o = off; off += (n + 3) & ~3; return o; };
    const int rows_pad = d.RT1 * 16;
    p.bufs = bufs;
    p.img_size = (d.G * d.CHW + 255) & ~255;     // whole KiB pieces (LDS-DMA); two buffers: the next group's frames
    p.img = take(bufs * p.img_size);             // travel while this group computes
    p.g1 = take(rows_pad * 16);                  // gelu'(z1), then dz1 of cotangent 0: [position row][16 channels]
    // the group's saved z1 rows as they sit in HBM (LDS-DMA target).  One frame buffer: they land where gelu'(z1) goes
    // (16 channels: the same [row][16] layout, every thread of the first phase rewrites the element it read)
    p.zraw = bufs == 1 ? p.g1 : take((d.rows1 * d.O1 + 255) & ~255);
    p.dp_pitch = 17 * d.k2 * d.k2;
    p.dp_pitch += (12 - (p.dp_pitch & 7)) & 7;   // = 4 mod 8
    int shared = d.G * d.O1 * d.M1;              // a1 | the product tile | the setup's patch offsets
    if (shared < 16 * p.dp_pitch) shared = 16 * p.dp_pitch;
    if (shared < 2 * kConvMaxK) shared = 2 * kConvMaxK;
    p.a1 = take(shared);
    p.ktab = p.a1;
    p.dz1_size = rows_pad * 16;
    p.dz1x = take((nc - 1) * p.dz1_size);
    p.dz2 = take(nc * 16 * 32);
    p.rowoff1 = take(rows_pad);
    p.rowa = take(rows_pad);
    p.n_cont = (conv_reach(d) * conv_reach(d) + 3) & ~3;
    p.cont = take(rows_pad * p.n_cont);          // per first-layer row: where its col2im contributions sit in the product tile
    p.red = take(kConvThreads);
    p.total = off;
    return p;
}
// Two frame buffers (the next group's frames and z1 rows travel while this group computes) — unless ONE buffer, with the raw
// z1 rows landing in gelu'(z1)'s place, is what lets a second workgroup share the CU (80 KB each): whole 30 x 30 frames in
// groups of four are 128 KB with two buffers and 78 KB with one; the other workgroup's arithmetic then covers a group's
// transfers, and every phase's LDS / MFMA latencies besides (the launch is latency-bound: 19 % of its clocks are MFMAs).
__host__ __device__ inline ConvBwdPlan conv_bwd_plan(const ConvDims& d, int nc = 1) {
    const ConvBwdPlan two = conv_bwd_layout(d, nc, 2);
    if (nc != 1 || d.tiles > 1 || d.O1 != 16 || (size_t)two.total * sizeof(float) <= 80 * 1024) return two;
    const ConvBwdPlan one = conv_bwd_layout(d, nc, 1);
    return (size_t)one.total * sizeof(float) <= 80 * 1024 ? one : two;
}

// workgroups of a backward launch (each writes one partial slab per cotangent): one per CU, two where the launch's LDS
// leaves room for two
inline int64_t conv_bwd_cap(const ConvDims& d, int nc = 1) {
    return kConvBwdGroupsCap * ((size_t)conv_bwd_plan(d, nc).total * sizeof(float) <= 80 * 1024 ? 2 : 1);
}

// packed parameter gradients: w1 | b1 | w2 | b2
__host__ __device__ inline int conv_param_count(const ConvDims& d) { return d.O1 * d.K1 + d.O1 + d.O2 * d.K2 + d.O2; }

constexpr int kNT1 = kConvMaxK / 16 / 4;        // layer-1 weight-gradient column tiles per wave (k1 index / 16)
constexpr int kNT2 = kConvMaxK / 16 / 4;        // layer-2 weight-gradient column tiles per wave, per row tile

// NC > 1: the backward walks of NC cotangents through ONE forward pass (the per-loss gradients of `calculate_adaptive_weights`,
// reference sac_base.py:1607-1631: the representation's graph differentiated once per gated loss) as one launch.  Everything that
// does not depend on the cotangent is done once per group of frames — the frames' and z1 rows' DMA, a1 and gelu'(z1), and
// above all the B operands of the two weight-gradient products, the patch elements gathered from LDS (the launch is bound
// by those gathers, not by its MFMAs) — and feeds NC MFMAs where it fed one.  Per cotangent the same operations in the same
// order as the single form: bit-identical gradients.
// KT1: layer-1 weight-gradient column tiles per wave — 3 for filters of <= 192 taps (8 x 8 over RGB frames: 12 tiles; a fourth
// per wave ran on clamped operands and was never stored: a quarter of the phase's gathers and MFMAs), else 4
template <bool TILED, int NC = 1, int KT1 = kNT1>
__global__ __launch_bounds__(kConvThreads) void k_conv2_bwd(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvDims& d = a.d;
    const ConvBwdPlan p = conv_bwd_plan(d, NC);
    float* zraw = lds + p.zraw;
    float* g1 = lds + p.g1;
    float* a1 = lds + p.a1;
    float* dpb = lds + p.a1;               // the product tile dz2 W2 of one cotangent (a1 is dead by then)
    float* dz1x = lds + p.dz1x;            // [NC - 1][dz1_size]: dz1 of cotangents 1 .. NC - 1
    int* cont = reinterpret_cast<int*>(lds + p.cont);
    float* dz2 = lds + p.dz2;              // [NC][16 * 32]
    int* rowoff1 = reinterpret_cast<int*>(lds + p.rowoff1);
    int* rowa = reinterpret_cast<int*>(lds + p.rowa);
    int* ktab = reinterpret_cast<int*>(lds + p.ktab);
    float* red = lds + p.red;
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    CropOffsets<kCropPieces> crop_r{};
    if (TILED) crop_r = crop_offsets<kCropPieces>(d, wave, lane);
    const CropOffsets<kCropPieces>* crop = &crop_r;
    const int rows_pad = d.RT1 * 16;
    const int NT1 = d.K1 / 16, NT2 = d.K2 / 16;

    // index tables, one entry per thread (see the note on integer division above)
    for (int row = threadIdx.x; row < rows_pad; row += kConvThreads) {
        const int rr = min(row, d.rows1 - 1);
        const int im = rr / d.M1, pos = rr - im * d.M1, oy = pos / d.W1, ox = pos - oy * d.W1;
        rowoff1[row] = im * d.CHW + d.s1 * oy * d.W + d.s1 * ox;
        rowa[row] = row < d.rows1 ? im * d.O1 * d.M1 + pos : -1;
    }
    // col2im as a gather: first-layer position (y, x) of a frame receives element (ky, kx) = (y - s2 oy, x - s2 ox) of the
    // patch gradient of every second-layer position (oy, ox) whose patch covers it — at most reach^2 of them, listed here
    // as the offset of that element's 17 channel slots in the product tile, -1: none; ascending (oy, ox): the order of the sum
    for (int row = threadIdx.x; row < rows_pad; row += kConvThreads) {
        const int im = row / d.M1, pos = row - im * d.M1, y = pos / d.W1, x = pos - y * d.W1;
        int n = 0;
        if (row < d.rows1) {
            for (int oy = max(0, (y - d.k2 + d.s2) / d.s2); oy <= min(y / d.s2, d.H2 - 1); ++oy)
                for (int ox = max(0, (x - d.k2 + d.s2) / d.s2); ox <= min(x / d.s2, d.W2 - 1); ++ox)
                    cont[row * p.n_cont + n++] =
                        (im * d.M2 + oy * d.W2 + ox) * p.dp_pitch + 17 * ((y - d.s2 * oy) * d.k2 + (x - d.s2 * ox));
        }
        for (; n < p.n_cont; ++n) cont[row * p.n_cont + n] = -1;
    }
    for (int k = threadIdx.x; k < kConvMaxK; k += kConvThreads) {
        ktab[k] = k < d.K1 ? patch_offset(k, d.k1, d.H * d.W, d.W) : 0;
        ktab[kConvMaxK + k] = k < d.K2 ? patch_offset(k, d.k2, d.M1, d.W1) : 0;
    }
    __syncthreads();
    // this lane's gradient columns: k index 16 c + lr of column tile c = wave + 4 i
    int k1off[KT1], k2off[kNT2], dpoff[kNT2];      // (dpoff: where k2 index 16 c + lr = (channel, ky, kx) sits in a product-tile row)
    float w2b[kNT2][8];             // W2[4 s + lk][16 c + lr]: B operand of the da1 GEMM (reduction over out2)
#pragma unroll
    for (int i = 0; i < KT1; ++i) {
        const int c = wave + 4 * i;
        k1off[i] = c < NT1 ? ktab[c * 16 + lr] : 0;
    }
#pragma unroll
    for (int i = 0; i < kNT2; ++i) {
        const int c = wave + 4 * i;
        k2off[i] = c < NT2 ? ktab[kConvMaxK + c * 16 + lr] : 0;
        dpoff[i] = 17 * ((c * 16 + lr) % (d.k2 * d.k2)) + (c * 16 + lr) / (d.k2 * d.k2);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int oc = 4 * s + lk;
            w2b[i][s] = (c < NT2 && oc < d.O2) ? a.w2[oc * d.K2 + c * 16 + lr] : 0.f;
        }
    }
    __syncthreads();                // (the offsets' table sits where the first group's a1 goes)
    // accumulators: dW1 [O1 <= 16][K1]: column tiles c = wave + 4 i;  dW2 [O2 <= 32][K2]: row tiles 0/1, same columns
    f32x4 dw1[NC][KT1], dw2[NC][2][kNT2];
    float db1[NC], db2[NC];                      // thread (channel = tid % 16 | tid % 32, row slice) partial bias sums
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int i = 0; i < KT1; ++i) dw1[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kNT2; ++i) dw2[c][0][i] = dw2[c][1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db1[c] = db2[c] = 0.f;
    }
    // layer-2 positions: row = frame*M2 + pos of the group's 16; this lane's B rows (4 step + lk) and accumulator
    // rows (4 lk + r), and the two dz2 elements (e / 32, e % 32) this thread forms
    // (rows beyond the group's G * M2 positions — 7 of the 16 with a 3 x 3 block — carry dz2 = 0: their operand
    // addresses repeat the last real row's, their col2im contributions are dropped)
    const int last_row = TILED ? d.G * d.M2 - 1 : 15;
    int rowbase[4], e_im[2], e_out[2], e_py[2], e_px[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = min(4 * s + lk, last_row), im = row / d.M2, pos = row - im * d.M2, oy = pos / d.W2, ox = pos - oy * d.W2;
        rowbase[s] = im * d.O1 * d.M1 + d.s2 * oy * d.W1 + d.s2 * ox;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = threadIdx.x + q * kConvThreads, row = e >> 5, oc = e & 31;
        e_im[q] = row / d.M2;
        const int pos = row - e_im[q] * d.M2;
        e_py[q] = pos / d.W2, e_px[q] = pos - e_py[q] * d.W2;
        e_out[q] = oc < d.O2 ? (TILED ? oc * d.FM2 : oc * d.M2 + pos) : -1;
    }
    // frames and saved z1 rows travel by LDS-DMA when 16-byte granular; the output-side operands (two elements of gy
    // and z2 per thread) are fetched into registers one group ahead
    const bool dma = TILED ||
                     ((d.CHW & 3) == 0 && ((d.M1 * d.O1 * d.G) & 3) == 0 && (a.x_sample_stride & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.z1)) & 15) == 0);
    // (tiled, the forward saved z1 per ROW of blocks, H1 x zW1 positions (ConvDims::zW1): this block's columns are picked out
    // of the slab, 16 bytes per lane, into the dense [rows1][O1] form the first phase reads — where a lane's pieces sit inside
    // a slab does not depend on the block: kConvZPieces offsets in registers)
    constexpr int kZPieces = kConvZPieces;
    int zoff[kZPieces];
    const int z_chunks = (d.rows1 * d.O1 + 255) >> 8;
    if (TILED && d.zW1) {
        const int per = d.O1 >> 2, n4 = d.rows1 * per;
#pragma unroll
        for (int j = 0; j < kZPieces; ++j) {
            const int i4 = min((wave + (kConvThreads / 64) * j) * 64 + lane, n4 - 1);
            const int pos = i4 / per, part = i4 - pos * per, y = pos / d.W1, xl = pos - y * d.W1;
            zoff[j] = (y * d.zW1 + xl) * d.O1 + 4 * part;
            settle(zoff[j]);
        }
    }
    const int z_slab = d.H1 * d.zW1 * d.O1, z_nby = TILED && d.zW1 ? d.tiles / d.nbx : 1;
    // the next group's frames (and raw z1 rows: the current ones are consumed) by LDS-DMA
    auto request = [&](int64_t g, int buf) {
        if (TILED) {
            const TileAt t = tile_at<true>(d, g);
            async_frames<true>(a, g, lds + p.img + buf * p.img_size, wave, lane, crop, &t);
            if (d.zW1) {
                const float* src = a.z1 + (t.frame * z_nby + t.by) * (int64_t)z_slab + t.c0 * d.s2 * d.O1;
#pragma unroll
                for (int j = 0; j < kZPieces; ++j) {
                    const int c = wave + (kConvThreads / 64) * j;
                    if (c < z_chunks)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + zoff[j]),
                                                         (__attribute__((address_space(3))) void*)(zraw + c * 256), 16, 0, 0);
                }
                return;
            }
        } else {
            const int64_t first = g * d.G;
            const int n_img = (int)min((int64_t)d.G, a.N - first);
            async_copy_kib(group_frames(a, g), lds + p.img + buf * p.img_size, (n_img * d.CHW) >> 2, p.img_size >> 8, wave,
                           lane);
        }
        const int64_t first = g * d.G;
        const int n_img = (int)min((int64_t)d.G, a.N - first);
        async_copy_kib(a.z1 + first * d.M1 * d.O1, zraw, (n_img * d.M1 * d.O1) >> 2, (d.rows1 * d.O1 + 255) >> 8, wave,
                       lane);
    };
    float gy_n[NC][2], z2_n[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) gy_n[c][0] = gy_n[c][1] = 0.f;
    auto fetch_out = [&](int64_t g) {
        const int64_t first = g * d.G;
        const int n_img = (int)min((int64_t)d.G, a.N - first);
        const TileAt at = tile_at<TILED>(d, g);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            z2_n[q] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) gy_n[c][q] = 0.f;
            if (e_out[q] >= 0 && e_im[q] < n_img && (!TILED || (e_py[q] >= at.skip_y && e_px[q] >= at.skip_x))) {
                const int64_t o = TILED ? (at.frame + e_im[q]) * ((int64_t)d.O2 * d.FM2) + e_out[q] +
                                              (at.r0 + e_py[q]) * d.FW2 + at.c0 + e_px[q]
                                        : (first + e_im[q]) * (d.O2 * d.M2) + e_out[q];
                gy_n[0][q] = a.gy[o];
#pragma unroll
                for (int c = 1; c < NC; ++c) gy_n[c][q] = a.gy_more[c - 1][o];
                z2_n[q] = a.z2[o];
            }
        }
    };
    if ((int64_t)blockIdx.x < a.n_groups) {
        if (dma) {
            request(blockIdx.x, 0);
        }
        fetch_out(blockIdx.x);
    }
    int buf = 0;
    CONV_STAMP_INIT;
    const bool one_buf = p.bufs == 1;         // (the next group's transfers wait for the end of this group: see conv_bwd_plan)
    for (int64_t g = blockIdx.x; g < a.n_groups; g += gridDim.x, buf = one_buf ? 0 : buf ^ 1) {
        CONV_STAMP(9);
        const int64_t first = g * d.G;
        const int n_img = (int)min((int64_t)d.G, a.N - first);
        float* img = lds + p.img + buf * p.img_size;
        const bool more = g + gridDim.x < a.n_groups;
        float gy_c[NC][2];
#pragma unroll
        for (int c = 0; c < NC; ++c) gy_c[c][0] = gy_n[c][0], gy_c[c][1] = gy_n[c][1];
        const float z2_c[2] = {z2_n[0], z2_n[1]};
        if (dma) {
            dma_barrier();             // this group's frames and z1 rows have landed (all waves' pieces)
            CONV_STAMP(0);
        } else {
            stage_frames(a, g, img);
        }
        // (the next group's frames / z1 rows / output-side operands are requested BEHIND the first phase, not here: that
        // phase reads LDS the DMA path writes (zraw) and registers earlier loads filled (gy, z2), and in front of such a
        // use the compiler waits for EVERY outstanding memory operation — `s_waitcnt vmcnt(0)` — i.e. for the prefetch just
        // issued: the launch had no overlap of a group's transfers with the previous group's arithmetic, round 6)
        // z1 (position-major rows, contiguous for the group) -> a1 (channel-major) and gelu'(z1)
        {
            const float* src = a.z1 + first * d.M1 * d.O1;
            const int count = n_img * d.M1 * d.O1;
            // (rows_pad * 16 = RT1 * 256 elements: thread (row slice tid / 16, channel tid % 16) takes RT1 of them, three at a
            // time with every LDS read of the three issued before the first use — one by one each element waited for two
            // dependent round trips (its row's offset, then its value): a third of the launch, NOTES round 6)
            const int oc = threadIdx.x & 15, r0 = threadIdx.x >> 4;
            constexpr int UB = 3;
            // (two instantiations: `dma ? zraw[j] : src[j]` selects between an LDS and a global POINTER — a flat load whose
            // wait, `vmcnt(0)`, also drained the next group's frame DMA requested just above)
            auto rows = [&](auto from_lds) {
                constexpr bool LDS = decltype(from_lds)::value;
                const auto* gsrc = as_global(src);
                for (int it0 = 0; it0 < d.RT1; it0 += UB) {
                    int ab[UB], jj[UB];
                    float zz[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int row = r0 + 16 * min(it0 + u, d.RT1 - 1);
                        ab[u] = rowa[row];
                        jj[u] = row * d.O1 + oc;
                        const int jc = min(jj[u], count - 1);
                        if constexpr (LDS) zz[u] = zraw[jc];
                        else zz[u] = gsrc[jc];
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        if (it0 + u < d.RT1) {
                            const int row = r0 + 16 * (it0 + u);
                            const bool on = ab[u] >= 0 && oc < d.O1, have = jj[u] < count;
                            float v, dv;
                            gelu_parts(have ? zz[u] : 0.f, v, dv);
                            if (on) a1[ab[u] + oc * d.M1] = v;
                            g1[row * 16 + oc] = (on && have) ? dv : 0.f;
                        }
                    }
                }
            };
            if (dma) rows(std::true_type{});
            else rows(std::false_type{});
            // dz2 [row = frame*M2 + pos][32 channels] = gy * gelu'(z2)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float gg = gelu_grad(z2_c[q]);
#pragma unroll
                for (int c = 0; c < NC; ++c) dz2[c * 512 + threadIdx.x + q * kConvThreads] = gy_c[c][q] * gg;
            }
        }
        lds_barrier();
        CONV_STAMP(1);
        if (more) {
            if (dma) {
                if (!one_buf) request(g + gridDim.x, buf ^ 1);        // (the raw rows are consumed)
            }
            fetch_out(g + gridDim.x);
        }
        // ---- bias 2, dW2 += dz2^T patches(a1), da1 patches = dz2 W2 --------------------------------------
        if (threadIdx.x < 32) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float s = 0.f;
                for (int row = 0; row < 16; ++row) s += dz2[c * 512 + row * 32 + threadIdx.x];
                db2[c] += s;
            }
        }
        {
            // (column tiles beyond K2 / 16 run on clamped operands and are never stored: a branch around an MFMA makes
            // the compiler shuttle its accumulator between register files on every step)
            // (the A operands — dz2 of the lane's rows — do not depend on the column tile: read once, not once per tile)
            float za[NC][4], zb[NC][4];
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    za[c][s] = dz2[c * 512 + (4 * s + lk) * 32 + lr], zb[c][s] = dz2[c * 512 + (4 * s + lk) * 32 + 16 + lr];
#pragma unroll
            for (int i = 0; i < kNT2; ++i) {               // column tile c = wave + 4 i: k2 = 16 c + lr
                const int ko = k2off[i];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float bv = a1[rowbase[s] + ko];          // (the gathered patch element: once for all cotangents)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        dw2[c][0][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[c][s], bv, dw2[c][0][i], 0, 0, 0);
                        dw2[c][1][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zb[c][s], bv, dw2[c][1][i], 0, 0, 0);
                    }
                }
            }
            CONV_STAMP(2);
            // da1 patch tile [16 rows][16 k2 of column tile c] = dz2 [16][32] * W2[32][k2]
            f32x4 dp[NC][kNT2];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float zr[8];                               // dz2[row lr][channel 4 s + lk]: the same for every column tile
#pragma unroll
                for (int s = 0; s < 8; ++s) zr[s] = dz2[c * 512 + lr * 32 + 4 * s + lk];
#pragma unroll
                for (int i = 0; i < kNT2; ++i) {
                    dp[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 8; ++s)                // reduction over the 32 output channels
                        dp[c][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zr[s], w2b[i][s], dp[c][i], 0, 0, 0);
                }
            }
            CONV_STAMP(3);
            // ---- da1 = col2im(dz2 W2) as a gather, dz1 = da1 * gelu'(z1), bias 1 ------------------------------------
            // A cotangent's product tile goes to LDS as the fragments hold it; element (position row, channel) of dz1 then
            // sums its <= reach^2 contributions and is scaled — no read-modify-write passes with a barrier each (one per
            // position, or per colour class of positions whose patches cannot overlap: a third of the tiled launch and a
            // seventh of the whole-frame one, with the da1 buffer they needed), no da1 at all.
            lds_barrier();                                     // dW2's reads of a1 are done: the tile takes its place
            const int oc = threadIdx.x & 15, r0 = threadIdx.x >> 4, nq = p.n_cont >> 2;
            const bool oc_on = oc < d.O1;
#pragma unroll
            for (int c = NC - 1; c >= 0; --c) {                // (cotangent 0 last: it overwrites gelu'(z1))
#pragma unroll
                for (int i = 0; i < kNT2; ++i) {
                    if (wave + 4 * i < NT2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dpb[(4 * lk + r) * p.dp_pitch + dpoff[i]] = dp[c][i][r];
                    }
                }
                // (a chunk of rows per step; a chunk's table entries and gelu' values — they do not depend on the tile — are
                // requested one step ahead, the first chunk's in front of the barrier: one LDS round trip per step, not three)
                constexpr int UB = 3;
                float* out = c == 0 ? g1 : dz1x + (c - 1) * p.dz1_size;
                auto fetch = [&](int it0, int4 (&cn_)[UB], float (&gp_)[UB]) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int row = r0 + 16 * min(it0 + u, d.RT1 - 1);
                        cn_[u] = reinterpret_cast<const int4*>(cont)[row * nq];
                        gp_[u] = g1[row * 16 + oc];
                    }
                };
                float s = 0.f;
                auto finish = [&](int it0, const int4 (&cn_)[UB], const float (&gp_)[UB]) {
                    float sum[UB], t[UB][4];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int e[4] = {cn_[u].x, cn_[u].y, cn_[u].z, cn_[u].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[u][j] = dpb[max(e[j], 0) + oc];
                    }
                    // (every read is issued, then masked: left to itself the compiler branches around each read of an absent
                    // contribution and waits for each one that is present by itself)
#pragma unroll
                    for (int u = 0; u < UB; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) settle(t[u][j]);
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int e[4] = {cn_[u].x, cn_[u].y, cn_[u].z, cn_[u].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[u][j] = (e[j] >= 0 && oc_on) ? t[u][j] : 0.f;
                        sum[u] = ((t[u][0] + t[u][1]) + t[u][2]) + t[u][3];
                    }
                    for (int q = 1; q < nq; ++q) {             // (patches reaching further than two positions per axis)
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int4 cq = reinterpret_cast<const int4*>(cont)[(r0 + 16 * min(it0 + u, d.RT1 - 1)) * nq + q];
                            const int e[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float v = dpb[max(e[j], 0) + oc];
                                sum[u] += (e[j] >= 0 && oc_on) ? v : 0.f;
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        if (it0 + u < d.RT1) {
                            const float v = sum[u] * gp_[u];
                            out[(r0 + 16 * (it0 + u)) * 16 + oc] = v;
                            s += v;
                        }
                    }
                };
                int4 cnA[UB], cnB[UB];
                float gpA[UB], gpB[UB];
                fetch(0, cnA, gpA);
                lds_barrier();
                CONV_STAMP(7);
                for (int it0 = 0; it0 < d.RT1; it0 += 2 * UB) {
                    fetch(it0 + UB, cnB, gpB);                 // (beyond the end: the last row's again, not used)
                    __builtin_amdgcn_sched_barrier(0);
                    finish(it0, cnA, gpA);
                    if (it0 + UB < d.RT1) {
                        fetch(it0 + 2 * UB, cnA, gpA);
                        __builtin_amdgcn_sched_barrier(0);
                        finish(it0 + UB, cnB, gpB);
                    }
                }
                db1[c] += s;
                if (c > 0) lds_barrier();                      // the tile is consumed: the next cotangent's replaces it
            }
        }
        CONV_STAMP(4);
        lds_barrier();
        CONV_STAMP(5);
        // ---- dW1 += dz1^T patches(x): reduction over the group's positions, 4 per step -------------------
        // One iteration = 4 reduction steps (16 positions): row offsets -> gathered patch elements are two dependent LDS
        // round trips in front of 16 NC MFMAs.  Pipelined by hand: the gathers of iteration k + 1 (and the row offsets of
        // k + 2) are requested before the MFMAs of k are issued and wait behind them; the loop runs two iterations per trip
        // so that the two operand sets alternate without register copies.
        {
            const int n_it = rows_pad / 16;                    // = RT1
            auto offsets = [&](int it, int (&ro_)[4]) {
                const int s0 = 4 * min(it, n_it - 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) ro_[u] = rowoff1[4 * (s0 + u) + lk];
            };
            auto operands = [&](int it, const int (&ro_)[4], float (&av_)[NC][4], float (&bv_)[4][KT1]) {
                const int s0 = 4 * min(it, n_it - 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = 4 * (s0 + u) + lk;
#pragma unroll
                    for (int c = 0; c < NC; ++c)                   // A[m = channel lr][k = row]
                        av_[c][u] = (c == 0 ? g1 : dz1x + (c - 1) * p.dz1_size)[row * 16 + lr];
#pragma unroll
                    for (int i = 0; i < KT1; ++i) bv_[u][i] = img[ro_[u] + k1off[i]];     // (gathered once for all cotangents)
                }
            };
            auto products = [&](const float (&av_)[NC][4], const float (&bv_)[4][KT1]) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int i = 0; i < KT1; ++i)
                            dw1[c][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[c][u], bv_[u][i], dw1[c][i], 0, 0, 0);
            };
            int roA[4], roB[4];
            float avA[NC][4], bvA[4][KT1], avB[NC][4], bvB[4][KT1];
            offsets(0, roA);
            offsets(1, roB);
            operands(0, roA, avA, bvA);
            for (int it = 0; it < n_it; it += 2) {
                operands(it + 1, roB, avB, bvB);               // (beyond the end: the last iteration's again, not used)
                offsets(it + 2, roA);
                __builtin_amdgcn_sched_barrier(0);             // (left alone the scheduler sinks every read to its use)
                products(avA, bvA);
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < n_it) {
                    operands(it + 2, roA, avA, bvA);
                    offsets(it + 3, roB);
                    __builtin_amdgcn_sched_barrier(0);
                    products(avB, bvB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        lds_barrier();             // everything of this group is consumed
        if (one_buf && dma && more) request(g + gridDim.x, 0);
        CONV_STAMP(6);
    }
    CONV_STAMP_FLUSH;

    // ---- this workgroup's partial gradients -> its slabs (one per cotangent): w1 | b1 | w2 | b2 -----------------------
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float* out = a.partial + ((int64_t)blockIdx.x * NC + c) * conv_param_count(d);
        float* o_b1 = out + d.O1 * d.K1;
        float* o_w2 = o_b1 + d.O1;
        float* o_b2 = o_w2 + d.O2 * d.K2;
#pragma unroll
        for (int i = 0; i < KT1; ++i) {
            const int ct = wave + 4 * i;
            if (ct < NT1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int oc = 4 * lk + r;
                    if (oc < d.O1) out[oc * d.K1 + ct * 16 + lr] = dw1[c][i][r];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kNT2; ++i) {
            const int ct = wave + 4 * i;
            if (ct < NT2) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oc = rt * 16 + 4 * lk + r;
                        if (oc < d.O2) o_w2[oc * d.K2 + ct * 16 + lr] = dw2[c][rt][i][r];
                    }
            }
        }
        // bias sums: threads of the same channel (tid % 16) hold row slices
        if (c > 0) __syncthreads();
        red[threadIdx.x] = db1[c];
        __syncthreads();
        if (threadIdx.x < 16 && (int)threadIdx.x < d.O1) {
            float sum = 0.f;
            for (int q = 0; q < kConvThreads / 16; ++q) sum += red[q * 16 + threadIdx.x];
            o_b1[threadIdx.x] = sum;
        }
        if (threadIdx.x < 32 && (int)threadIdx.x < d.O2) o_b2[threadIdx.x] = db2[c];
    }
}

// sum of the workgroups' slabs in fixed order (64 parameters per workgroup, 16 slices of slabs, then the slices)
constexpr int kSumSlices = 16;
__global__ __launch_bounds__(64 * kSumSlices) void k_conv_sum_partials(const float* __restrict__ partial, int blocks,
                                                                       int n, float* __restrict__ out, int accumulate) {
    __shared__ float part[kSumSlices][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int per = (blocks + kSumSlices - 1) / kSumSlices;
    const int lo = sl * per, hi = min(lo + per, blocks);
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
    for (int w = 0; w < kSumSlices; ++w) s += part[w][lane];
    out[i] = accumulate ? out[i] + s : s;
}

constexpr size_t kConvLdsLimit = 160 * 1024;

static bool conv_block_rows(const ConvDims& d, ConvDims& u);
static bool conv_dims(const asac_conv2_desc_t& c, ConvDims& d) {
    if (c.channels < 1 || c.height < 1 || c.width < 1 || c.out1 < 1 || c.out2 < 1 || c.kernel1 < 1 || c.kernel2 < 1 ||
        c.stride1 < 1 || c.stride2 < 1)
        return false;
    d = ConvDims{};
    d.C = c.channels; d.H = c.height; d.W = c.width; d.CHW = d.C * d.H * d.W;
    d.O1 = c.out1; d.k1 = c.kernel1; d.s1 = c.stride1;
    d.O2 = c.out2; d.k2 = c.kernel2; d.s2 = c.stride2;
    if (d.H < d.k1 || d.W < d.k1) return false;
    d.H1 = (d.H - d.k1) / d.s1 + 1; d.W1 = (d.W - d.k1) / d.s1 + 1; d.M1 = d.H1 * d.W1; d.K1 = d.C * d.k1 * d.k1;
    if (d.H1 < d.k2 || d.W1 < d.k2) return false;
    d.H2 = (d.H1 - d.k2) / d.s2 + 1; d.W2 = (d.W1 - d.k2) / d.s2 + 1; d.M2 = d.H2 * d.W2; d.K2 = d.O1 * d.k2 * d.k2;
    if (d.O1 > 16 || d.O2 > 32 || d.K1 > kConvMaxK || d.K2 > kConvMaxK || (d.K1 & 15) || (d.K2 & 15)) return false;
    d.tiles = 1, d.nbx = 1, d.bh = d.H2, d.bw = d.W2;
    d.FH = d.H, d.FW = d.W, d.FHW = d.H * d.W, d.FCHW = d.CHW, d.FH2 = d.H2, d.FW2 = d.W2, d.FM2 = d.M2;
    d.crop4 = 0, d.cw4 = 0;
    d.sub = 1, d.zW1 = 0;
    auto fits = [&]() {
        return (size_t)conv_fwd_plan(d).total * sizeof(float) <= kConvLdsLimit &&
               (size_t)conv_bwd_plan(d).total * sizeof(float) <= kConvLdsLimit;
    };
    if (d.M2 <= 16 && (16 % d.M2) == 0) {
        d.G = 16 / d.M2;
        d.rows1 = d.G * d.M1;
        d.RT1 = (d.rows1 + 15) / 16;
        if (fits()) return true;
    }
    // tiled: blocks of bh x bw <= 16 second-layer positions; the block shape that recomputes the fewest layer-1 positions
    // among those whose crops are 16-byte granular (LDS-DMA of crop rows) and fit the LDS
    if (d.M2 <= 4 || (d.FW & 3) || (d.FHW & 3) || ((d.s1 * d.s2) & 3)) return false;
    const ConvDims full = d;
    int64_t best = -1;
    ConvDims pick{};
    for (int bh = 1; bh <= 16 && bh <= full.H2; ++bh)
        for (int bw = 1; bh * bw <= 16 && bw <= full.W2; ++bw) {
            if (bh * bw <= 4) continue;           // (never the cheapest cover; not exercised)
            ConvDims t = full;
            t.bh = bh, t.bw = bw;
            t.H2 = bh, t.W2 = bw, t.M2 = bh * bw;
            t.H1 = (bh - 1) * t.s2 + t.k2, t.W1 = (bw - 1) * t.s2 + t.k2, t.M1 = t.H1 * t.W1;
            t.H = (t.H1 - 1) * t.s1 + t.k1, t.W = (t.W1 - 1) * t.s1 + t.k1, t.CHW = t.C * t.H * t.W;
            if ((t.W & 3) || t.CHW > kCropMaxFloats) continue;
            const int nby = (full.H2 + bh - 1) / bh;
            t.nbx = (full.W2 + bw - 1) / bw;
            t.tiles = nby * t.nbx;
            t.crop4 = t.CHW / 4, t.cw4 = t.W / 4;
            t.G = 1, t.rows1 = t.M1, t.RT1 = (t.rows1 + 15) / 16;
            d = t;
            if (!fits()) continue;
            // cost: layer-1 positions computed per frame, a whole 16-row tile at a time, plus the layer-2 tiles
            const int64_t cost = (int64_t)t.tiles * (t.RT1 * 16 * (int64_t)t.K1 * 16 + 16 * (int64_t)t.K2 * 32);
            if (best < 0 || cost < best) best = cost, pick = t;
        }
    if (best < 0) return false;
    d = pick;
    d.sub = 1, d.zW1 = 0;
    ConvDims rows;
    if (conv_block_rows(d, rows)) d.zW1 = rows.W1;       // (the forward works by block rows: z1 is saved per row of blocks)
    return true;
}

// The forward's block-row form of a tiled geometry (ConvDims::sub): the blocks of one block row as ONE unit of work — their
// crops are the same frame rows (full width: contiguous, 1.25x instead of 1.6x the frame's bytes at 84 x 84) and their
// first-layer regions overlap (8 x 20 positions once instead of two times 8 x 12).  Taken when there are at least two
// blocks a row and the unit still fits two workgroups per CU; false: the caller keeps one block per unit.
static bool conv_block_rows(const ConvDims& d, ConvDims& u) {
    if (d.tiles <= 1 || d.nbx < 2 || (d.O1 & 3)) return false;       // (O1 % 4: the backward picks 16-byte pieces of a position)
    u = d;
    u.W = d.FW, u.CHW = d.C * d.H * d.FW;
    u.W1 = (d.FW - d.k1) / d.s1 + 1, u.M1 = d.H1 * u.W1;
    u.rows1 = u.M1, u.RT1 = (u.rows1 + 15) / 16;
    u.crop4 = u.CHW / 4, u.cw4 = u.W / 4;
    u.sub = d.nbx, u.zW1 = u.W1;
    u.tiles = d.tiles / d.nbx, u.nbx = 1;
    if ((u.W & 3) || u.CHW > kCropMaxFloatsRow || d.rows1 * d.O1 > kConvZPieces * (kConvThreads / 64) * 256) return false;
    return (size_t)conv_fwd_plan(u).total * sizeof(float) <= 80 * 1024;
}

static int conv_lds_limit(const void* fn, bool& done, const char* where) {
    if (done) return 0;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConvLdsLimit);
    if (err != hipSuccess) {
        set_error(err, where);
        return (int)err;
    }
    done = true;
    return 0;
}

}  // namespace asac

using namespace asac;

extern "C" {

int asac_conv2_supported(const asac_conv2_desc_t* desc) {
    ConvDims d;
    return desc && conv_dims(*desc, d) ? 1 : 0;
}

int asac_conv2_group_frames(const asac_conv2_desc_t* desc) {
    ConvDims d;
    return desc && conv_dims(*desc, d) ? d.G : -1;
}

int64_t asac_conv2_param_count(const asac_conv2_desc_t* desc) {
    ConvDims d;
    if (!desc || !conv_dims(*desc, d)) return -1;
    return conv_param_count(d);
}

int64_t asac_conv2_backward_workspace(const asac_conv2_desc_t* desc, int64_t N) {
    ConvDims d;
    if (!desc || !conv_dims(*desc, d) || N <= 0) return -1;
    const int64_t groups = (N * d.tiles + d.G - 1) / d.G;
    return (groups < conv_bwd_cap(d) ? groups : conv_bwd_cap(d)) * conv_param_count(d);
}

int64_t asac_conv2_z1_floats(const asac_conv2_desc_t* desc, int64_t N) {
    ConvDims d;
    if (!desc || !conv_dims(*desc, d) || N <= 0) return -1;
    if (d.zW1) return N * (d.tiles / d.nbx) * d.H1 * d.zW1 * d.O1;      // (one slab per row of blocks: see ConvDims::sub)
    return N * d.tiles * d.M1 * d.O1;
}

int asac_conv2_tiles(const asac_conv2_desc_t* desc) {
    ConvDims d;
    return desc && conv_dims(*desc, d) ? d.tiles : -1;
}

int asac_conv2_forward(const asac_conv2_desc_t* desc, const float* x, int64_t N, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* y, float* z1_out, float* z2_out, void* stream) {
    return asac_conv2_forward_windows(desc, x, N, 0, 0, w1, b1, w2, b2, y, z1_out, z2_out, stream);
}

int asac_conv2_forward_windows(const asac_conv2_desc_t* desc, const float* x, int64_t N, int frames_per_sample,
                               int64_t sample_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* y, float* z1_out, float* z2_out, void* stream) {
    ConvArgs a{};
    if (!desc || !conv_dims(*desc, a.d) || N <= 0 || !x || !w1 || !b1 || !w2 || !b2 || !y || (!z1_out != !z2_out))
        return bad_arg("asac_conv2_forward");
    if (frames_per_sample) {
        if (frames_per_sample < 0 || frames_per_sample % a.d.G != 0 || N % frames_per_sample != 0 ||
            sample_stride < (int64_t)frames_per_sample * a.d.FCHW)
            return bad_arg("asac_conv2_forward_windows");
        a.x_sample_groups = frames_per_sample / a.d.G;
        a.x_sample_stride = sample_stride;
    }
    if (a.d.tiles > 1 && ((reinterpret_cast<uintptr_t>(x) & 15) || (sample_stride & 3) || (a.d.FCHW & 3)))
        return bad_arg("asac_conv2_forward: tiled frames need 16-byte aligned rows");
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.y = y; a.z1 = z1_out; a.z2 = z2_out;
    if (a.d.zW1) {                                           // (tiled: a row of blocks per unit where that fits)
        ConvDims rows;
        conv_block_rows(a.d, rows);
        a.d = rows;
    }
    a.N = N * a.d.tiles;                                     // (tiled: virtual frames, one group each)
    a.n_groups = (a.N + a.d.G - 1) / a.d.G;
    const size_t lds = (size_t)conv_fwd_plan(a.d).total * sizeof(float);
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int64_t cap = 256 * per_cu;
    const unsigned blocks = (unsigned)(a.n_groups < cap ? a.n_groups : cap);
    // the layer-1 operand constants of the full tiles in registers where the quad count is one of the usual ones (8 x 8
    // filters over 1 / 3 / 4 input channels; cfg4 +0.9 %, cfg4_84 +2 % A/B); any other count: the LDS tables
    static bool attr[2][4] = {};
    auto launch = [&](auto kernel, bool& done) -> int {
        if (int rc = conv_lds_limit(reinterpret_cast<const void*>(kernel), done, "asac_conv2_forward")) return rc;
        ASAC_LAUNCH(kernel, dim3(blocks), dim3(kConvThreads), lds, as_stream(stream), a);
        return 0;
    };
    const bool tiled = a.d.tiles > 1;
    int rc = 0;
    switch (a.d.K1 / 16) {
    case 4: rc = tiled ? launch(k_conv2_fwd<true, 4>, attr[1][1]) : launch(k_conv2_fwd<false, 4>, attr[0][1]); break;
    case 12: rc = tiled ? launch(k_conv2_fwd<true, 12>, attr[1][2]) : launch(k_conv2_fwd<false, 12>, attr[0][2]); break;
    case 16: rc = tiled ? launch(k_conv2_fwd<true, 16>, attr[1][3]) : launch(k_conv2_fwd<false, 16>, attr[0][3]); break;
    default: rc = tiled ? launch(k_conv2_fwd<true, 0>, attr[1][0]) : launch(k_conv2_fwd<false, 0>, attr[0][0]); break;
    }
    if (rc) return rc;
    return finish_launch("asac_conv2_forward");
}

int asac_conv2_backward(const asac_conv2_desc_t* desc, const float* x, int64_t N, const float* w2, const float* z1,
                        const float* z2, const float* grad_y, float* grad_params, int accumulate, float* workspace,
                        void* stream) {
    return asac_conv2_backward_windows(desc, x, N, 0, 0, w2, z1, z2, grad_y, grad_params, accumulate, workspace, stream);
}

int asac_conv2_backward_windows(const asac_conv2_desc_t* desc, const float* x, int64_t N, int frames_per_sample,
                                int64_t sample_stride, const float* w2, const float* z1, const float* z2,
                                const float* grad_y, float* grad_params, int accumulate, float* workspace, void* stream) {
    ConvArgs a{};
    if (!desc || !conv_dims(*desc, a.d) || N <= 0 || !x || !w2 || !z1 || !z2 || !grad_y || !grad_params || !workspace)
        return bad_arg("asac_conv2_backward");
    if (frames_per_sample) {
        if (frames_per_sample < 0 || frames_per_sample % a.d.G != 0 || N % frames_per_sample != 0 ||
            sample_stride < (int64_t)frames_per_sample * a.d.FCHW)
            return bad_arg("asac_conv2_backward_windows");
        a.x_sample_groups = frames_per_sample / a.d.G;
        a.x_sample_stride = sample_stride;
    }
    if (a.d.tiles > 1 && ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(z1) & 15) ||
                          (sample_stride & 3) || (a.d.FCHW & 3) || ((a.d.M1 * a.d.O1) & 3)))
        return bad_arg("asac_conv2_backward: tiled frames need 16-byte aligned rows");
    a.x = x; a.w2 = w2;
    a.z1 = const_cast<float*>(z1); a.z2 = const_cast<float*>(z2);
    a.gy = grad_y;
    a.partial = workspace;
    a.N = N * a.d.tiles;
    a.n_groups = (a.N + a.d.G - 1) / a.d.G;
    static bool attr = false, attr_t = false;
    const size_t lds = (size_t)conv_bwd_plan(a.d).total * sizeof(float);
    const unsigned blocks = (unsigned)(a.n_groups < conv_bwd_cap(a.d) ? a.n_groups : conv_bwd_cap(a.d));
    hipStream_t s = as_stream(stream);
    static bool attr3 = false, attr3_t = false;
    const bool kt3 = a.d.K1 / 16 <= 12;            // (three column tiles per wave cover the filter: see k_conv2_bwd)
    auto launch = [&](auto kernel, bool& done) -> int {
        if (int rc = conv_lds_limit(reinterpret_cast<const void*>(kernel), done, "asac_conv2_backward")) return rc;
        ASAC_LAUNCH(kernel, dim3(blocks), dim3(kConvThreads), lds, s, a);
        return 0;
    };
    if (int rc = a.d.tiles > 1 ? (kt3 ? launch(k_conv2_bwd<true, 1, 3>, attr3_t) : launch(k_conv2_bwd<true, 1, 4>, attr_t))
                               : (kt3 ? launch(k_conv2_bwd<false, 1, 3>, attr3) : launch(k_conv2_bwd<false, 1, 4>, attr)))
        return rc;
    const int n = conv_param_count(a.d);
    // launched once (not under the repeat knob: it may accumulate); ASAC_CONV_SUM_DEFER: left to asac_sum_partials_multi
    if (accumulate != ASAC_CONV_SUM_DEFER)
        hipLaunchKernelGGL(k_conv_sum_partials, dim3((unsigned)((n + 63) / 64)), dim3(64 * kSumSlices), 0, s, workspace,
                           (int)blocks, n, grad_params, accumulate);
    return finish_launch("asac_conv2_backward");
}

// partial slabs (= workgroups) a backward launch over N frames with n_cot cotangents leaves in its workspace: what a deferred
// reduction (asac_sum_partials_multi) sums
int asac_conv2_backward_slabs(const asac_conv2_desc_t* desc, int64_t N, int n_cot) {
    ConvDims d;
    if (!desc || !conv_dims(*desc, d) || N <= 0 || n_cot < 1 || n_cot > asac_conv2_backward_multi_max(desc)) return -1;
    const int64_t groups = (N * d.tiles + d.G - 1) / d.G, cap = conv_bwd_cap(d, n_cot);
    return (int)(groups < cap ? groups : cap);
}

// the largest number of cotangents (1..4) whose buffers fit the LDS beside the double-buffered frames: what ONE launch of
// asac_conv2_backward_multi takes (more are issued in launches of at most this many)
int asac_conv2_backward_multi_max(const asac_conv2_desc_t* desc) {
    ConvDims d;
    if (!desc || !conv_dims(*desc, d)) return -1;
    int best = 1;
    for (int nc = 2; nc <= 3; ++nc)          // (a fourth set of accumulators leaves no registers for the operand pipeline)
        if ((size_t)conv_bwd_plan(d, nc).total * sizeof(float) <= kConvLdsLimit) best = nc;
    return best;
}

// Several backward walks of ONE forward pass (n_cot cotangents grad_ys[c], each [N][out2 * M2]) as one launch:
// grads_out [n_cot][param_count], cotangent c's packed gradients (w1 | b1 | w2 | b2) in slab c; workspace:
// n_cot x asac_conv2_backward_workspace floats.  Bit-identical to n_cot asac_conv2_backward(_windows) calls.
int asac_conv2_backward_multi(const asac_conv2_desc_t* desc, const float* x, int64_t N, int frames_per_sample,
                              int64_t sample_stride, const float* w2, const float* z1, const float* z2,
                              const float* const* grad_ys, int n_cot, float* grads_out, int accumulate, float* workspace,
                              void* stream) {
    ConvArgs a{};
    if (!desc || !conv_dims(*desc, a.d) || N <= 0 || !x || !w2 || !z1 || !z2 || !grad_ys || n_cot < 1 || n_cot > 4 ||
        !grads_out || !workspace)
        return bad_arg("asac_conv2_backward_multi");
    for (int c = 0; c < n_cot; ++c)
        if (!grad_ys[c]) return bad_arg("asac_conv2_backward_multi: cotangent");
    if (n_cot == 1)
        return asac_conv2_backward_windows(desc, x, N, frames_per_sample, sample_stride, w2, z1, z2, grad_ys[0], grads_out,
                                           accumulate, workspace, stream);
    if (frames_per_sample) {
        if (frames_per_sample < 0 || frames_per_sample % a.d.G != 0 || N % frames_per_sample != 0 ||
            sample_stride < (int64_t)frames_per_sample * a.d.FCHW)
            return bad_arg("asac_conv2_backward_multi: windows");
        a.x_sample_groups = frames_per_sample / a.d.G;
        a.x_sample_stride = sample_stride;
    }
    if (a.d.tiles > 1 && ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(z1) & 15) ||
                          (sample_stride & 3) || (a.d.FCHW & 3) || ((a.d.M1 * a.d.O1) & 3)))
        return bad_arg("asac_conv2_backward_multi: tiled frames need 16-byte aligned rows");
    const size_t lds = (size_t)conv_bwd_plan(a.d, n_cot).total * sizeof(float);
    if (n_cot > asac_conv2_backward_multi_max(desc)) {         // (more buffer sets than fit: launches of as many as do)
        if (accumulate == ASAC_CONV_SUM_DEFER)                 // (... which share the workspace: their partials cannot wait)
            return bad_arg("asac_conv2_backward_multi: deferred sums take at most asac_conv2_backward_multi_max cotangents");
        const int64_t n = conv_param_count(a.d);
        const int most = asac_conv2_backward_multi_max(desc);
        for (int c = 0; c < n_cot; c += most)
            if (int rc = asac_conv2_backward_multi(desc, x, N, frames_per_sample, sample_stride, w2, z1, z2, grad_ys + c,
                                                   n_cot - c < most ? n_cot - c : most, grads_out + c * n, accumulate,
                                                   workspace, stream))
                return rc;
        return 0;
    }
    a.x = x; a.w2 = w2;
    a.z1 = const_cast<float*>(z1); a.z2 = const_cast<float*>(z2);
    a.gy = grad_ys[0];
    for (int c = 1; c < n_cot; ++c) a.gy_more[c - 1] = grad_ys[c];
    a.partial = workspace;
    a.N = N * a.d.tiles;
    a.n_groups = (a.N + a.d.G - 1) / a.d.G;
    static bool attr[2][2][2] = {};
    const unsigned blocks = (unsigned)(a.n_groups < conv_bwd_cap(a.d, n_cot) ? a.n_groups : conv_bwd_cap(a.d, n_cot));
    hipStream_t s = as_stream(stream);
    const bool tiled = a.d.tiles > 1, kt3 = a.d.K1 / 16 <= 12;
    auto launch = [&](auto kernel, bool& done) -> int {
        if (int rc = conv_lds_limit(reinterpret_cast<const void*>(kernel), done, "asac_conv2_backward_multi")) return rc;
        ASAC_LAUNCH(kernel, dim3(blocks), dim3(kConvThreads), lds, s, a);
        return 0;
    };
    int rc = 0;
    if (n_cot == 2)
        rc = tiled ? (kt3 ? launch(k_conv2_bwd<true, 2, 3>, attr[1][0][0]) : launch(k_conv2_bwd<true, 2, 4>, attr[1][0][1]))
                   : (kt3 ? launch(k_conv2_bwd<false, 2, 3>, attr[0][0][0]) : launch(k_conv2_bwd<false, 2, 4>, attr[0][0][1]));
    else
        rc = tiled ? (kt3 ? launch(k_conv2_bwd<true, 3, 3>, attr[1][1][0]) : launch(k_conv2_bwd<true, 3, 4>, attr[1][1][1]))
                   : (kt3 ? launch(k_conv2_bwd<false, 3, 3>, attr[0][1][0]) : launch(k_conv2_bwd<false, 3, 4>, attr[0][1][1]));
    if (rc) return rc;
    const int n = conv_param_count(a.d) * n_cot;       // (a block's n_cot slabs are consecutive: one reduction over all of them)
    if (accumulate != ASAC_CONV_SUM_DEFER)
        hipLaunchKernelGGL(k_conv_sum_partials, dim3((unsigned)((n + 63) / 64)), dim3(64 * kSumSlices), 0, s, workspace,
                           (int)blocks, n, grads_out, accumulate);
    return finish_launch("asac_conv2_backward_multi");
}

}  // extern "C"
