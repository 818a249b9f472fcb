// Window gather + episode-continuity padding (K3): the device code shared by the stand-alone launch (gather.hip) and the
// sidecar form (asac_sidecar.h: ASAC_SIDECAR_WINDOW_GATHER — the NEXT batch's gather of the lookahead schedule riding in a
// launch of the current step).  See gather.hip for the layout.
#pragma once
#include "asac_common.h"

namespace asac {

constexpr int kGatherBlock = 256;
// units per thread (template parameter UNROLL): a gather of a few thousand units (the headline batch: 8 keys x 1 280 rows)
// wants ONE unit per thread — with more, each thread walks several dependent id -> index ring -> row chains and the launch
// is a handful of workgroups (cfg2: +4 % steps/s, A/B on one box).  A gather of megabytes wants a few 16-byte loads in
// flight per lane: TWO (round 5; it had been four since round 1).  `tools/k14_probe.py`, fresh ids every launch, 100 / 200
// MB on a 711 MB ring: one unit 21.2 / 36.7 us, two 20.7 / 35.2, three 21.4 / 36.5, four 21.4 / 37.3, six 24.6 / 38.2;
// in the step (A/B on one box): cfg4 2 895 vs 2 874 steps/s, cfg5 438.9 vs 437.8; the kernel in situ 22.8 / 40.1 us
// against 23.7-24.8 / 41.6-45.
#ifndef ASAC_GATHER_UNROLL
#define ASAC_GATHER_UNROLL 2
#endif
constexpr int kUnrollLarge = ASAC_GATHER_UNROLL;
constexpr int64_t kSmallGatherBlocks = 8192;      // up to this many one-unit workgroups: UNROLL = 1 (cfg3: 1 100, +0.9 %)

struct GatherKeyDev {
    const uint8_t* src;
    uint8_t* dst;
    const uint8_t* pad_row;
    int32_t row_bytes;
    int32_t pad_mode;
    uint32_t pad_word;
    int32_t convert;
    int32_t unit_log2;       // log2 of the SOURCE unit size in bytes (0, 2 or 4)
    int32_t units_per_row;
    uint32_t first_block;    // prefix sum of blocks over the keys
    int32_t dst_pitch;       // bytes between destination rows (dense: the output row's size)
    int32_t derive;          // ASAC_DERIVE_*: the window row a destination row is taken from
};

struct GatherArgs {          // what every key's blocks share
    int32_t n_keys;
    const int64_t* ids;
    const int32_t* index_ring;
    int32_t batch, prev_n, L, capacity;
};
// (sized key tables: a launch's argument block is fetched on its critical path, ~0.5-0.9 us per KB; the usual batches
// have at most eight keys — 448 bytes less than the full table)
template <int NK>
struct GatherLaunch {
    GatherKeyDev key[NK];
    GatherArgs c;
};

__device__ __forceinline__ int64_t load_id(const int64_t* ids, int s) { return ids[s]; }

__device__ __forceinline__ bool row_valid(const GatherArgs& a, int64_t id, int j) {
    if (j == a.prev_n) return true;
    const int idx_j = a.index_ring[ring_slot(id + (j - a.prev_n), a.capacity)];
    const int idx_c = a.index_ring[ring_slot(id, a.capacity)];
    return (idx_j - idx_c) == (j - a.prev_n);
}

template <typename Unit>
__device__ __forceinline__ Unit pad_value(const GatherKeyDev& k, int w);

template <>
__device__ __forceinline__ uint4 pad_value<uint4>(const GatherKeyDev& k, int w) {
    if (k.pad_mode == ASAC_PAD_ROW) return reinterpret_cast<const uint4*>(k.pad_row)[w];
    uint32_t x = k.pad_word;
    if (k.pad_mode == ASAC_PAD_BYTE) x = (x & 0xff) * 0x01010101u;
    return make_uint4(x, x, x, x);
}
template <>
__device__ __forceinline__ uint32_t pad_value<uint32_t>(const GatherKeyDev& k, int w) {
    if (k.pad_mode == ASAC_PAD_ROW) return reinterpret_cast<const uint32_t*>(k.pad_row)[w];
    uint32_t x = k.pad_word;
    if (k.pad_mode == ASAC_PAD_BYTE) x = (x & 0xff) * 0x01010101u;
    return x;
}
template <>
__device__ __forceinline__ uint8_t pad_value<uint8_t>(const GatherKeyDev& k, int w) {
    if (k.pad_mode == ASAC_PAD_ROW) return k.pad_row[w];
    if (k.pad_mode == ASAC_PAD_WORD) return (uint8_t)(k.pad_word >> (8 * (w & 3)));
    return (uint8_t)(k.pad_word & 0xff);
}

// Derived keys (ASAC_DERIVE_*): destination row j of the window is the key's padded row `src_row(j)`; the first row of a
// PREVIOUS key is zeros, the last row of a HOLD_LAST_NEXT key (an i32 column: the step index) counts one further unless
// it is the padding value -1 — SAC_Base.get_bnx_data (sac_base.py:1090-1115) formed inside the gather.
__device__ __forceinline__ int derived_row(int derive, int j, int L) {
    if (derive == ASAC_DERIVE_PREVIOUS) return j > 0 ? j - 1 : 0;
    if (derive >= ASAC_DERIVE_HOLD_LAST) return min(j, L - 2);
    return j;
}
template <typename Unit> __device__ __forceinline__ Unit zero_unit() { return Unit(0); }
template <> __device__ __forceinline__ uint4 zero_unit<uint4>() { return make_uint4(0, 0, 0, 0); }
template <typename Unit> __device__ __forceinline__ Unit next_index(Unit v) { return v; }
template <> __device__ __forceinline__ uint32_t next_index<uint32_t>(uint32_t v) { return v + (v != 0xffffffffu ? 1u : 0u); }

template <typename Unit, int kUnroll>
__device__ __forceinline__ void copy_units(const GatherArgs& a, const GatherKeyDev& k, int64_t g0,
                                           int64_t total_units) {
    // g indexes units of the dense destination [B, L, units_per_row]
    int64_t g[kUnroll], at[kUnroll];
    Unit val[kUnroll];
    bool live[kUnroll];
#pragma unroll
    for (int r = 0; r < kUnroll; ++r) {
        g[r] = g0 + (int64_t)r * kGatherBlock;
        live[r] = g[r] < total_units;
        if (!live[r]) continue;
        const int64_t row = g[r] / k.units_per_row;
        const int w = (int)(g[r] - row * k.units_per_row);
        at[r] = row * k.dst_pitch + (int64_t)w * (int)sizeof(Unit);
        const int sample = (int)(row / a.L);
        const int j = (int)(row - (int64_t)sample * a.L);
        const int64_t id = load_id(a.ids, sample);
        // the row is read whether or not it turns out to belong to the centre row's episode (the slot is always a
        // valid address): the validity test's own loads — random reads of the index ring — travel WITH the data
        // instead of in front of it
        const int js = derived_row(k.derive, j, a.L);
        const int slot = ring_slot(id + (js - a.prev_n), a.capacity);
        const Unit data = reinterpret_cast<const Unit*>(k.src + (int64_t)slot * k.row_bytes)[w];
        const bool valid = (k.pad_mode == ASAC_PAD_KEEP) || row_valid(a, id, js);
        val[r] = valid ? data : pad_value<Unit>(k, w);
        if (k.derive == ASAC_DERIVE_PREVIOUS && j == 0) val[r] = zero_unit<Unit>();
        if (k.derive == ASAC_DERIVE_HOLD_LAST_NEXT && j == a.L - 1) val[r] = next_index<Unit>(val[r]);
    }
#pragma unroll
    for (int r = 0; r < kUnroll; ++r)
        if (live[r]) *reinterpret_cast<Unit*>(k.dst + at[r]) = val[r];
}

// conversion path: 4 source bytes -> 4 floats (uint8/255 or bool)
template <int kUnroll>
__device__ __forceinline__ void convert_units(const GatherArgs& a, const GatherKeyDev& k, int64_t g0,
                                              int64_t total_units) {
#pragma unroll
    for (int r = 0; r < kUnroll; ++r) {
        const int64_t g = g0 + (int64_t)r * kGatherBlock;
        if (g >= total_units) continue;
        const int64_t row = g / k.units_per_row;
        const int w = (int)(g - row * k.units_per_row);
        const int sample = (int)(row / a.L);
        const int j = (int)(row - (int64_t)sample * a.L);
        const int64_t id = load_id(a.ids, sample);
        const int slot = ring_slot(id + (j - a.prev_n), a.capacity);
        const uint8_t* srow = k.src + (int64_t)slot * k.row_bytes;
        float* drow = reinterpret_cast<float*>(k.dst + row * k.dst_pitch);
        if (k.unit_log2 == 2) {
            const uint32_t x = reinterpret_cast<const uint32_t*>(srow)[w];
            float4 o;
            if (k.convert == ASAC_CVT_U8_TO_F32_UNIT) {
                o = make_float4((float)(x & 0xff) / 255.f, (float)((x >> 8) & 0xff) / 255.f,
                                (float)((x >> 16) & 0xff) / 255.f, (float)(x >> 24) / 255.f);
            } else {
                o = make_float4((x & 0xff) ? 1.f : 0.f, ((x >> 8) & 0xff) ? 1.f : 0.f,
                                ((x >> 16) & 0xff) ? 1.f : 0.f, (x >> 24) ? 1.f : 0.f);
            }
            reinterpret_cast<float4*>(drow)[w] = o;
        } else {
            const uint8_t x = srow[w];
            drow[w] = (k.convert == ASAC_CVT_U8_TO_F32_UNIT) ? (float)x / 255.f : (x ? 1.f : 0.f);
        }
    }
}

// workgroup `block` of a gather launch: its key `k` and the shared arguments `a`
template <int kUnroll>
__device__ __forceinline__ void gather_work(const GatherArgs& a, const GatherKeyDev& k, unsigned block) {
    const int64_t rows = (int64_t)a.batch * a.L;
    const int64_t total_units = rows * k.units_per_row;
    const int64_t g0 = (int64_t)(block - k.first_block) * (kGatherBlock * kUnroll) + threadIdx.x;

    if (k.pad_mode == ASAC_PAD_EMIT_MASK) {
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t g = g0 + (int64_t)r * kGatherBlock;
            if (g >= rows) continue;
            const int sample = (int)(g / a.L);
            const int j = (int)(g - (int64_t)sample * a.L);
            k.dst[g] = row_valid(a, load_id(a.ids, sample), derived_row(k.derive, j, a.L)) ? 0 : 1;
        }
        return;
    }
    if (k.convert != ASAC_CVT_NONE) {
        convert_units<kUnroll>(a, k, g0, total_units);
        return;
    }
    if (k.unit_log2 == 4) copy_units<uint4, kUnroll>(a, k, g0, total_units);
    else if (k.unit_log2 == 2) copy_units<uint32_t, kUnroll>(a, k, g0, total_units);
    else copy_units<uint8_t, kUnroll>(a, k, g0, total_units);
}

// workgroup `block` of a gather launch described by `m` (read from the kernel arguments or from device memory)
template <int NK, int kUnroll>
__device__ __forceinline__ void gather_block(const GatherLaunch<NK>& m, unsigned block) {
    // which key does this block belong to?  (<= 16 entries, wave-uniform scan)
    int ki = 0;
#pragma unroll 1
    for (int q = 1; q < m.c.n_keys; ++q)
        if (block >= m.key[q].first_block) ki = q;
    gather_work<kUnroll>(m.c, m.key[ki], block);
}

// host: the launch description of a window gather (key table, shared arguments, workgroups per key); force_unroll 0:
// chosen by size (gather.hip)
int gather_fill(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int batch, int prev_n, int post_n,
                int capacity, const int32_t* index_ring, int force_unroll, GatherLaunch<ASAC_MAX_GATHER_KEYS>& m,
                uint64_t* blocks_out, int* unroll_out);

}  // namespace asac
