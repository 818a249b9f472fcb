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
#include "asac_gather.h"
#include "asac_sidecar.h"

namespace asac {

template <int NK, int kUnroll>
__global__ __launch_bounds__(kGatherBlock) void k_window_gather_pad(const GatherLaunch<NK> m) {
    gather_block<NK, kUnroll>(m, blockIdx.x);
}

// ... with the batch's IS weights (K2) formed by ONE extra workgroup beside the gather's: the sampler of the launch in
// front (asac_step_prologue_sample_partial: a workgroup per 256 samples) left its workgroups' minima in min_p_out[2..]; here
// they are combined, beta advances, and the weights of all <= 1 024 rows are written — under a gather of tens of
// microseconds instead of behind a cross-workgroup exchange inside the sampler (~1.5 us of the step's first launch).
struct WeightsJob {
    const float* p;            // [batch] leaf priorities of the sampled rows
    const float* tree;         // tree[0] = total
    double* beta_state;
    double beta_increment;
    float* w_out;              // [batch]
    float* min_p_out;          // [0] <- min p; [2 .. 2 + parts) the sampler workgroups' minima
    int32_t batch, parts;
};

__device__ __forceinline__ void weights_job(const WeightsJob& j) {
    float bm = j.min_p_out[2];
    for (int k = 1; k < j.parts; ++k) bm = fminf(bm, j.min_p_out[2 + k]);
    const float root = j.tree[0];
    const double b = fmin(1.0, *j.beta_state + j.beta_increment);
    const float min_ratio = bm / root;
    __syncthreads();               // every lane has read the old beta
    for (int i = threadIdx.x; i < j.batch; i += kGatherBlock) j.w_out[i] = is_weight(j.p[i], root, min_ratio, b);
    if (threadIdx.x == 0) {
        *j.beta_state = b;
        j.min_p_out[0] = bm;
    }
}

template <int NK, int kUnroll>
__global__ __launch_bounds__(kGatherBlock) void k_window_gather_pad_w(const GatherLaunch<NK> m, unsigned gather_blocks,
                                                                      const WeightsJob j) {
    if (blockIdx.x == 0) {         // (first in the grid: dispatched at once — as the LAST workgroup it started when the
        weights_job(j);            //  gather was nearly through and its f64 powers stuck out of the launch by ~1.2 us)
        return;
    }
    gather_block<NK, kUnroll>(m, blockIdx.x - 1);
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

// the launch description of a window gather (key table, shared arguments, workgroups per key); force_unroll 0: chosen by size
int gather_fill(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int batch, int prev_n, int post_n,
                       int capacity, const int32_t* index_ring, int force_unroll, GatherLaunch<ASAC_MAX_GATHER_KEYS>& m,
                       uint64_t* blocks_out, int* unroll_out) {
    if (n_keys <= 0 || n_keys > ASAC_MAX_GATHER_KEYS || batch <= 0 || prev_n < 0 || post_n < 0 ||
        capacity <= 0)
        return bad_arg("asac_window_gather_pad");
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
    const int unroll = force_unroll ? force_unroll : (small_blocks <= kSmallGatherBlocks ? 1 : kUnrollLarge);
    // Workgroups are dispatched in block order: the keys with the FEWEST units go first.  Their workgroups are the slow ones
    // — narrow rows, every unit behind its own chain of dependent loads (id -> index ring -> row) — and at the end of the
    // grid (the order the columns happen to have) they started when the frames' copy was through and stuck out of it (in situ 23.9 -> 23.1 / 40.9 -> 39.7 us; the step: neutral).
    if (n_keys > 1) {
        for (int i = 1; i < n_keys; ++i)                       // (insertion sort: <= 16 entries; stable)
            for (int j = i; j > 0 && rows * m.key[j].units_per_row < rows * m.key[j - 1].units_per_row; --j) {
                const GatherKeyDev t = m.key[j];
                m.key[j] = m.key[j - 1];
                m.key[j - 1] = t;
            }
    }
    for (int q = 0; q < n_keys; ++q) {
        GatherKeyDev& d = m.key[q];
        d.first_block = (uint32_t)blocks;
        const int64_t units = rows * d.units_per_row;
        blocks += (uint64_t)((units + kGatherBlock * unroll - 1) / (kGatherBlock * unroll));
    }
    if (blocks == 0 || blocks > 0x7fffffffull) return bad_arg("asac_window_gather_pad: grid");
    *blocks_out = blocks;
    *unroll_out = unroll;
    return 0;
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
    GatherLaunch<ASAC_MAX_GATHER_KEYS> m{};
    uint64_t blocks = 0;
    int unroll = 0;
    if (const int rc = gather_fill(keys_host, n_keys, ids, batch, prev_n, post_n, capacity, index_ring, 0, m, &blocks, &unroll))
        return rc;
    if (n_keys <= 8) {
        GatherLaunch<8> m8{};
        for (int q = 0; q < n_keys; ++q) m8.key[q] = m.key[q];
        m8.c = m.c;
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

int asac_window_gather_pad_w(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int batch, int prev_n,
                             int post_n, int capacity, const int32_t* index_ring, const float* p, const float* tree,
                             double* beta_state, double beta_increment, float* is_weights_out, float* min_p_out,
                             void* stream) {
    if (!p || !tree || !beta_state || !is_weights_out || !min_p_out || batch <= 256 || batch > 1024)
        return bad_arg("asac_window_gather_pad_w");
    GatherLaunch<ASAC_MAX_GATHER_KEYS> m{};
    uint64_t blocks = 0;
    int unroll = 0;
    if (const int rc = gather_fill(keys_host, n_keys, ids, batch, prev_n, post_n, capacity, index_ring, 0, m, &blocks, &unroll))
        return rc;
    const dim3 grid((unsigned)blocks + 1u);
    // (under the measurement repeat knob beta advances in the first repetition only)
    for (int rep = 0; rep < g_launch_repeat; ++rep) {
        const WeightsJob j{p, tree, beta_state, rep == 0 ? beta_increment : 0.0, is_weights_out, min_p_out, batch,
                           (batch + 255) / 256};
        if (n_keys <= 8) {
            GatherLaunch<8> m8{};
            for (int q = 0; q < n_keys; ++q) m8.key[q] = m.key[q];
            m8.c = m.c;
            if (unroll == 1)
                hipLaunchKernelGGL((k_window_gather_pad_w<8, 1>), grid, dim3(kGatherBlock), 0, as_stream(stream), m8,
                                   (unsigned)blocks, j);
            else
                hipLaunchKernelGGL((k_window_gather_pad_w<8, kUnrollLarge>), grid, dim3(kGatherBlock), 0, as_stream(stream),
                                   m8, (unsigned)blocks, j);
        } else if (unroll == 1) {
            hipLaunchKernelGGL((k_window_gather_pad_w<ASAC_MAX_GATHER_KEYS, 1>), grid, dim3(kGatherBlock), 0,
                               as_stream(stream), m, (unsigned)blocks, j);
        } else {
            hipLaunchKernelGGL((k_window_gather_pad_w<ASAC_MAX_GATHER_KEYS, kUnrollLarge>), grid, dim3(kGatherBlock), 0,
                               as_stream(stream), m, (unsigned)blocks, j);
        }
    }
    return finish_launch("asac_window_gather_pad_w");
}

int64_t asac_window_gather_plan_bytes(void) { return (int64_t)sizeof(GatherLaunch<ASAC_MAX_GATHER_KEYS>); }

int asac_window_gather_plan(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int batch, int prev_n,
                            int post_n, int capacity, const int32_t* index_ring, void* plan_dev, int* blocks_out) {
    if (!plan_dev || !blocks_out) return bad_arg("asac_window_gather_plan");
    GatherLaunch<ASAC_MAX_GATHER_KEYS> m{};
    uint64_t blocks = 0;
    int unroll = 0;
    // (one unit per thread: a sidecar's workgroups share the machine with their host's)
    if (const int rc = gather_fill(keys_host, n_keys, ids, batch, prev_n, post_n, capacity, index_ring, 1, m, &blocks, &unroll))
        return rc;
    const hipError_t e = hipMemcpy(plan_dev, &m, sizeof(m), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error(e, "asac_window_gather_plan");
        return (int)e;
    }
    *blocks_out = (int)blocks;
    return 0;
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
