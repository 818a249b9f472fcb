// Episode slabs for gfx950: the agent-side episode assembly (reference algorithm/agent.py) kept in HBM.
//
// The reference keeps, per live agent, a dict of NumPy arrays [max_episode_length, *shape] and moves one
// transition at a time through Python (`Agent.set_tmp_obs_action` 87-97, `_add_transition` 191-235,
// `get_episode_trans` 258-316), then stacks the agents' pending values for `choose_action`
// (`AgentManager._get_merged_*` 474-485) — a dozen small host copies per agent and environment step, and a
// host -> device copy of the finished episode in `put_episode`.  Here every key is one slab
// [slots, rows, row_bytes] in HBM plus per-slot pending rows [slots, row_bytes]; the three movements of an
// environment step are ONE launch each over all agents and all keys:
//   commit   pending rows (+ reward / done / index scalars of this step) -> slab[slot, cursor]
//   collect  pending action / hidden state of the listed agents         -> dense batch for the policy
//   stage    the policy's outputs and the new observations              -> pending rows
// and the episode window of the attention agents (`AgentManager.get_action` 536-560) is the same kernel
// reading slab rows with a per-key row offset and zero / -1 padding in front of short episodes.
//
// The kernel is a row mover: item i of key k copies row_bytes from
//   src_k + a*src_stride0 + b*src_stride1      to      dst_k + c*dst_stride0 + d*dst_stride1
// with (a, b) / (c, d) chosen by the key's addressing mode from (i, slot[i], src_row[i], dst_row[i]).
// Work is flattened over (key, item, unit), unit = 16 / 4 / 1 bytes per key, like the replay gather (K3).
#include "asac_common.h"

namespace asac {

constexpr int kMoveBlock = 256;
constexpr int kMoveUnroll = 4;

struct RowMoveDev {
    const uint8_t* src;
    uint8_t* dst;
    int64_t src_stride0, src_stride1, dst_stride0, dst_stride1;
    int32_t row_bytes, src_mode, dst_mode, row_offset;
    uint32_t pad_word;
    int32_t unit_log2, units_per_row;
    uint32_t first_block;
};

struct RowMoveArgs {
    RowMoveDev key[ASAC_MAX_GATHER_KEYS];
    int32_t n_keys, n_items;
    const int32_t* slot;
    const int32_t* src_row;
    const int32_t* dst_row;
};

// byte offset of item i under an addressing mode; *neg := the row index is negative (source: emit padding)
__device__ __forceinline__ int64_t row_address(int mode, int64_t s0, int64_t s1, int i, const int32_t* slot,
                                               const int32_t* row, int row_offset, bool* neg) {
    *neg = false;
    switch (mode) {
        case ASAC_ROW_ITEM: return (int64_t)i * s0;
        case ASAC_ROW_SLOT: return (int64_t)slot[i] * s0;
        case ASAC_ROW_SLOT_ROW: {
            const int r = row[i] + row_offset;
            *neg = r < 0;
            return (int64_t)slot[i] * s0 + (int64_t)(r < 0 ? 0 : r) * s1;
        }
        default: return 0;   // ASAC_ROW_BROADCAST
    }
}

template <typename Unit>
__device__ __forceinline__ Unit move_pad(uint32_t w);
template <>
__device__ __forceinline__ uint4 move_pad<uint4>(uint32_t w) { return make_uint4(w, w, w, w); }
template <>
__device__ __forceinline__ uint32_t move_pad<uint32_t>(uint32_t w) { return w; }
template <>
__device__ __forceinline__ uint8_t move_pad<uint8_t>(uint32_t w) { return (uint8_t)(w & 0xff); }

template <typename Unit>
__device__ __forceinline__ void move_units(const RowMoveArgs& a, const RowMoveDev& k, int64_t g0, int64_t total) {
    Unit val[kMoveUnroll];
    int64_t doff[kMoveUnroll];
    bool live[kMoveUnroll];
#pragma unroll
    for (int r = 0; r < kMoveUnroll; ++r) {
        const int64_t g = g0 + (int64_t)r * kMoveBlock;
        live[r] = g < total;
        if (!live[r]) continue;
        const int i = (int)(g / k.units_per_row);
        const int w = (int)(g - (int64_t)i * k.units_per_row);
        bool neg, dneg;
        const int64_t so = row_address(k.src_mode, k.src_stride0, k.src_stride1, i, a.slot, a.src_row, k.row_offset, &neg);
        doff[r] = row_address(k.dst_mode, k.dst_stride0, k.dst_stride1, i, a.slot, a.dst_row, 0, &dneg) +
                  (int64_t)w * (int64_t)sizeof(Unit);
        live[r] = !dneg;
        val[r] = neg ? move_pad<Unit>(k.pad_word) : reinterpret_cast<const Unit*>(k.src + so)[w];
    }
#pragma unroll
    for (int r = 0; r < kMoveUnroll; ++r)
        if (live[r]) *reinterpret_cast<Unit*>(k.dst + doff[r]) = val[r];
}

__global__ __launch_bounds__(kMoveBlock) void k_rows_move(const RowMoveArgs a) {
    int ki = 0;
#pragma unroll 1
    for (int q = 1; q < a.n_keys; ++q)
        if (blockIdx.x >= a.key[q].first_block) ki = q;
    const RowMoveDev& k = a.key[ki];
    const int64_t total = (int64_t)a.n_items * k.units_per_row;
    const int64_t g0 = (int64_t)(blockIdx.x - k.first_block) * (kMoveBlock * kMoveUnroll) + threadIdx.x;
    if (k.unit_log2 == 4) move_units<uint4>(a, k, g0, total);
    else if (k.unit_log2 == 2) move_units<uint32_t>(a, k, g0, total);
    else move_units<uint8_t>(a, k, g0, total);
}

}  // namespace asac

using namespace asac;

extern "C" {

int asac_rows_move(const asac_row_move_t* keys_host, int n_keys, const int32_t* slot, const int32_t* src_row,
                   const int32_t* dst_row, int n_items, void* stream) {
    if (n_keys <= 0 || n_keys > ASAC_MAX_GATHER_KEYS || !keys_host || n_items < 0) return bad_arg("asac_rows_move");
    if (n_items == 0) return 0;
    RowMoveArgs a;
    a.n_keys = n_keys;
    a.n_items = n_items;
    a.slot = slot;
    a.src_row = src_row;
    a.dst_row = dst_row;
    uint64_t blocks = 0;
    for (int q = 0; q < n_keys; ++q) {
        const asac_row_move_t& h = keys_host[q];
        RowMoveDev& d = a.key[q];
        if (!h.src || !h.dst || h.row_bytes <= 0) return bad_arg("asac_rows_move: key");
        for (int side = 0; side < 2; ++side) {
            const int mode = side ? h.dst_mode : h.src_mode;
            if (mode < ASAC_ROW_ITEM || mode > ASAC_ROW_BROADCAST) return bad_arg("asac_rows_move: mode");
            if ((mode == ASAC_ROW_SLOT || mode == ASAC_ROW_SLOT_ROW) && !slot) return bad_arg("asac_rows_move: slot");
            if (mode == ASAC_ROW_SLOT_ROW && !(side ? dst_row : src_row)) return bad_arg("asac_rows_move: row");
        }
        if (h.dst_mode == ASAC_ROW_BROADCAST) return bad_arg("asac_rows_move: broadcast destination");
        d.src = static_cast<const uint8_t*>(h.src);
        d.dst = static_cast<uint8_t*>(h.dst);
        d.src_stride0 = h.src_stride0;
        d.src_stride1 = h.src_stride1;
        d.dst_stride0 = h.dst_stride0;
        d.dst_stride1 = h.dst_stride1;
        d.row_bytes = h.row_bytes;
        d.src_mode = h.src_mode;
        d.dst_mode = h.dst_mode;
        d.row_offset = h.src_row_offset;
        const uint64_t al = reinterpret_cast<uintptr_t>(h.src) | reinterpret_cast<uintptr_t>(h.dst) |
                            (uint64_t)h.src_stride0 | (uint64_t)h.src_stride1 | (uint64_t)h.dst_stride0 |
                            (uint64_t)h.dst_stride1 | (uint64_t)h.row_bytes;
        d.unit_log2 = (al % 16 == 0) ? 4 : (al % 4 == 0) ? 2 : 0;
        d.units_per_row = h.row_bytes >> d.unit_log2;
        d.pad_word = d.unit_log2 == 0 ? (h.pad_word & 0xff) : h.pad_word;
        d.first_block = (uint32_t)blocks;
        const int64_t units = (int64_t)n_items * d.units_per_row;
        blocks += (uint64_t)((units + kMoveBlock * kMoveUnroll - 1) / (kMoveBlock * kMoveUnroll));
    }
    if (blocks == 0 || blocks > 0x7fffffffull) return bad_arg("asac_rows_move: grid");
    ASAC_LAUNCH(k_rows_move, dim3((unsigned)blocks), dim3(kMoveBlock), 0, as_stream(stream), a);
    return finish_launch("asac_rows_move");
}

}  // extern "C"
