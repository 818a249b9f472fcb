// Ring-storage kernels for gfx950: fused window gather + episode-continuity padding (K3) and the
// predicated, last-writer-wins row scatter of the write-backs (K7).  C ABI in include/asac_hip.h.
//
// HBM layout: one ring per stored key, [C, row_bytes] row-major, plus the id map i64[C] and the
// 'index' column i32[C].  A sampled window is L = prev_n+1+post_n consecutive ring slots, i.e. one
// contiguous span of L*row_bytes bytes (two spans at the ring seam), copied into a dense
// [B, L, row_bytes] batch tensor.  The work is flattened over (key, sample, row, unit) with
// unit = 16 / 4 / 1 bytes chosen per key from its row size, so consecutive lanes touch consecutive
// addresses whatever the row width (4-byte rewards ... 10.8 KB images).  HBM-bound: algorithmic
// traffic = 2 * B * L * row_bytes per key (+ 8 B/sample of ids).
#include "asac_common.h"
#include "asac_sidecar.h"

namespace asac {

constexpr int kGatherBlock = 256;
// units per thread (template parameter UNROLL): 4 keeps >= 4 x 16 B loads in flight per lane — what a gather of
// megabytes wants (70-75 % of the HBM peak); a gather of a few thousand units (the headline batch: 8 keys x 1 280 rows)
// is a handful of workgroups that way, each thread walking four dependent id -> index ring -> row chains: with ONE
// unit per thread and four times the workgroups the same launch is 4 us shorter (cfg2: +4 % steps/s, A/B on one box)
constexpr int kUnrollLarge = 4;
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
        const int64_t id = a.ids[sample];
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
        const int64_t id = a.ids[sample];
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

template <int NK, int kUnroll>
__global__ __launch_bounds__(kGatherBlock) void k_window_gather_pad(const GatherLaunch<NK> m) {
    const GatherArgs& a = m.c;
    // which key does this block belong to?  (<= 16 entries, wave-uniform scan)
    int ki = 0;
#pragma unroll 1
    for (int q = 1; q < a.n_keys; ++q)
        if (blockIdx.x >= m.key[q].first_block) ki = q;
    const GatherKeyDev& k = m.key[ki];
    const int64_t rows = (int64_t)a.batch * a.L;
    const int64_t total_units = rows * k.units_per_row;
    const int64_t g0 = (int64_t)(blockIdx.x - k.first_block) * (kGatherBlock * kUnroll) + threadIdx.x;

    if (k.pad_mode == ASAC_PAD_EMIT_MASK) {
#pragma unroll
        for (int r = 0; r < kUnroll; ++r) {
            const int64_t g = g0 + (int64_t)r * kGatherBlock;
            if (g >= rows) continue;
            const int sample = (int)(g / a.L);
            const int j = (int)(g - (int64_t)sample * a.L);
            k.dst[g] = row_valid(a, a.ids[sample], derived_row(k.derive, j, a.L)) ? 0 : 1;
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

// K7 (asac_sidecar.h: ScatterArgs, scatter_elect_row, scatter_write_row)
__global__ __launch_bounds__(256) void k_scatter_elect(const ScatterArgs a) {
    scatter_elect_row(a, blockIdx.x * blockDim.x + threadIdx.x);
}

// one wave per target row; lanes stride over the payload in 4-byte (or 1-byte) units
__global__ __launch_bounds__(256) void k_scatter_write(const ScatterArgs a) {
    scatter_write_row(a, blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave, threadIdx.x & (kWave - 1), kWave);
}

// The representation's window inputs that are pure functions of the sampled window (reference
// SAC_Base.get_bnx_data, sac_base.py:1090-1115 + utils/operators.py gen_n_pre_actions), one lane per (b, t):
//   index_x[b][t]      = t < L-1 ? index[b][t] : index[b][L-2] + (index[b][L-2] != -1)
//   pad_x[b][t]        = t < L-1 ? pad[b][t]   : pad[b][L-2]
//   pre_action[b][t][:] = t == 0 ? 0 : action[b][t-1][:]
__global__ __launch_bounds__(256) void k_window_aux(const int32_t* __restrict__ index, int64_t index_sb,
                                                    const uint8_t* __restrict__ pad, int64_t pad_sb,
                                                    const float* __restrict__ action, int64_t action_sb,
                                                    int64_t action_st, int B, int L, int A,
                                                    int32_t* __restrict__ index_x, uint8_t* __restrict__ pad_x,
                                                    float* __restrict__ pre_action, int64_t pre_action_st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, t = i - b * L;
    const int ts = t < L - 1 ? t : L - 2;
    const int32_t iv = index[(int64_t)b * index_sb + ts];
    index_x[i] = t < L - 1 ? iv : iv + (iv != -1 ? 1 : 0);
    pad_x[i] = pad[(int64_t)b * pad_sb + ts];
    float* dst = pre_action + (int64_t)i * pre_action_st;
    const float* src = action + (int64_t)b * action_sb + (int64_t)(t - 1) * action_st;
    for (int d = 0; d < A; ++d) dst[d] = t == 0 ? 0.f : src[d];
}

}  // namespace asac

using namespace asac;

extern "C" {

int asac_window_aux(const int32_t* index, int64_t index_stride_b, const uint8_t* padding_mask,
                    int64_t mask_stride_b, const float* action, int64_t action_stride_b, int64_t action_stride_t,
                    int B, int L, int A, int32_t* index_x_out, uint8_t* padding_mask_x_out,
                    float* pre_action_out, int64_t pre_action_stride_t, void* stream) {
    if (pre_action_stride_t == 0) pre_action_stride_t = A;
    if (B <= 0 || L < 2 || A <= 0 || !index || !padding_mask || !action || !index_x_out || !padding_mask_x_out ||
        !pre_action_out || pre_action_stride_t < A)
        return bad_arg("asac_window_aux");
    ASAC_LAUNCH(k_window_aux, dim3((unsigned)((B * L + 255) / 256)), dim3(256), 0, as_stream(stream), index,
                index_stride_b, padding_mask, mask_stride_b, action, action_stride_b, action_stride_t, B, L, A,
                index_x_out, padding_mask_x_out, pre_action_out, pre_action_stride_t);
    return finish_launch("asac_window_aux");
}

int asac_window_gather_pad(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids,
                           int batch, int prev_n, int post_n, int capacity,
                           const int32_t* index_ring, void* stream) {
    if (n_keys <= 0 || n_keys > ASAC_MAX_GATHER_KEYS || batch <= 0 || prev_n < 0 || post_n < 0 ||
        capacity <= 0)
        return bad_arg("asac_window_gather_pad");
    GatherLaunch<ASAC_MAX_GATHER_KEYS> m{};
    GatherArgs& a = m.c;
    a.n_keys = n_keys;
    a.ids = ids;
    a.index_ring = index_ring;
    a.batch = batch;
    a.prev_n = prev_n;
    a.L = prev_n + 1 + post_n;
    a.capacity = capacity;
    const int64_t rows = (int64_t)batch * a.L;
    uint64_t blocks = 0;
    for (int q = 0; q < n_keys; ++q) {
        const asac_gather_key_t& h = keys_host[q];
        GatherKeyDev& d = m.key[q];
        d.src = static_cast<const uint8_t*>(h.src);
        d.dst = static_cast<uint8_t*>(h.dst);
        d.pad_row = static_cast<const uint8_t*>(h.pad_row);
        d.row_bytes = h.row_bytes;
        d.pad_mode = h.pad_mode;
        d.pad_word = h.pad_word;
        d.convert = h.convert;
        const int out_row_bytes = h.convert != ASAC_CVT_NONE ? 4 * h.row_bytes : h.row_bytes;
        d.dst_pitch = h.dst_row_pitch ? h.dst_row_pitch : out_row_bytes;
        d.derive = h.derive;
        if (h.derive < ASAC_DERIVE_NONE || h.derive > ASAC_DERIVE_HOLD_LAST_NEXT ||
            (h.derive != ASAC_DERIVE_NONE && (h.convert != ASAC_CVT_NONE || a.L < 2)) ||
            (h.derive == ASAC_DERIVE_HOLD_LAST_NEXT && (h.row_bytes != 4 || h.pad_mode == ASAC_PAD_EMIT_MASK)))
            return bad_arg("asac_window_gather_pad: derived key");
        if (h.pad_mode == ASAC_PAD_EMIT_MASK) {
            if (h.dst_row_pitch) return bad_arg("asac_window_gather_pad: the mask is dense");
            d.unit_log2 = 0;
            d.units_per_row = 1;
        } else {
            if (h.row_bytes <= 0 || !h.src || !h.dst || d.dst_pitch < out_row_bytes) return bad_arg("asac_window_gather_pad: key");
            const uintptr_t al = reinterpret_cast<uintptr_t>(h.src) | reinterpret_cast<uintptr_t>(h.dst) |
                                 (uintptr_t)(uint32_t)d.dst_pitch |
                                 (h.pad_mode == ASAC_PAD_ROW ? reinterpret_cast<uintptr_t>(h.pad_row) : 0);
            int ul = 0;
            if (h.convert != ASAC_CVT_NONE) {
                // source rows of 4-byte groups when possible; destination is 4x wider
                ul = (h.row_bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(h.src) % 4 == 0) &&
                      (reinterpret_cast<uintptr_t>(h.dst) % 16 == 0) && d.dst_pitch % 16 == 0) ? 2 : 0;
                if (h.pad_mode != ASAC_PAD_KEEP) return bad_arg("asac_window_gather_pad: convert+pad");
            } else if (h.row_bytes % 16 == 0 && al % 16 == 0) {
                ul = 4;
            } else if (h.row_bytes % 4 == 0 && al % 4 == 0) {
                ul = 2;
            }
            d.unit_log2 = ul;
            d.units_per_row = h.row_bytes >> ul;
        }
    }
    // one unit per thread while that keeps the launch within a few workgroups per CU, else four (see kUnrollLarge)
    int64_t small_blocks = 0;
    for (int q = 0; q < n_keys; ++q) small_blocks += (rows * m.key[q].units_per_row + kGatherBlock - 1) / kGatherBlock;
    const int unroll = small_blocks <= kSmallGatherBlocks ? 1 : kUnrollLarge;
    for (int q = 0; q < n_keys; ++q) {
        m.key[q].first_block = (uint32_t)blocks;
        const int64_t units = rows * m.key[q].units_per_row;
        blocks += (uint64_t)((units + kGatherBlock * unroll - 1) / (kGatherBlock * unroll));
    }
    if (blocks == 0 || blocks > 0x7fffffffull) return bad_arg("asac_window_gather_pad: grid");
    if (n_keys <= 8) {
        GatherLaunch<8> m8{};
        for (int q = 0; q < n_keys; ++q) m8.key[q] = m.key[q];
        m8.c = a;
        if (unroll == 1)
            ASAC_LAUNCH((k_window_gather_pad<8, 1>), dim3((unsigned)blocks), dim3(kGatherBlock), 0, as_stream(stream), m8);
        else
            ASAC_LAUNCH((k_window_gather_pad<8, kUnrollLarge>), dim3((unsigned)blocks), dim3(kGatherBlock), 0, as_stream(stream), m8);
    } else if (unroll == 1) {
        ASAC_LAUNCH((k_window_gather_pad<ASAC_MAX_GATHER_KEYS, 1>), dim3((unsigned)blocks), dim3(kGatherBlock), 0,
                    as_stream(stream), m);
    } else {
        ASAC_LAUNCH((k_window_gather_pad<ASAC_MAX_GATHER_KEYS, kUnrollLarge>), dim3((unsigned)blocks), dim3(kGatherBlock), 0,
                    as_stream(stream), m);
    }
    return finish_launch("asac_window_gather_pad");
}

int asac_gather_rows(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int n_rows, int capacity,
                     void* stream) {
    if (n_keys <= 0 || n_keys > ASAC_MAX_GATHER_KEYS || !keys_host) return bad_arg("asac_gather_rows");
    for (int q = 0; q < n_keys; ++q)        // plain rows: no window, no padding, no widening
        if (keys_host[q].pad_mode != ASAC_PAD_KEEP || keys_host[q].convert != ASAC_CVT_NONE)
            return bad_arg("asac_gather_rows: key");
    return asac_window_gather_pad(keys_host, n_keys, ids, n_rows, 0, 0, capacity, nullptr, stream);
}

int asac_scatter_rows_if_id_match(void* ring, int row_bytes, int capacity, const int64_t* ids,
                                  int batch, int first_off, int count, const int64_t* slot_ids,
                                  const uint8_t* padding_mask, int mask_sample_stride,
                                  const void* rows, int64_t rows_sample_stride_bytes,
                                  int64_t rows_row_stride_bytes, int32_t* winner, void* stream) {
    if (row_bytes <= 0 || capacity <= 0 || batch <= 0 || count <= 0 || !winner || !slot_ids)
        return bad_arg("asac_scatter_rows_if_id_match");
    ScatterArgs a{static_cast<uint8_t*>(ring), row_bytes, capacity, ids, batch, first_off, count,
                  slot_ids, padding_mask, mask_sample_stride, static_cast<const uint8_t*>(rows),
                  rows_sample_stride_bytes, rows_row_stride_bytes, winner};
    const int total = batch * count;
    hipStream_t s = as_stream(stream);
    ASAC_LAUNCH(k_scatter_elect, dim3((total + 255) / 256), dim3(256), 0, s, a);
    ASAC_LAUNCH(k_scatter_write, dim3((total + 3) / 4), dim3(256), 0, s, a);
    return finish_launch("asac_scatter_rows_if_id_match");
}

}  // extern "C"
