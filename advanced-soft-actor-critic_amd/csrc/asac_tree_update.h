// The single-workgroup priority update (K6) as device code: shared by its own launches (sumtree.hip) and the launch
// that forms the TD errors and updates the priorities in one go (returns.hip).
#pragma once
#include "asac_common.h"

#include <cmath>

namespace asac {

// ------------------------------------------------------------------------------------------------
// K6 / add: single workgroup.  Leaves first (duplicates resolved to the LAST writer through the
// `winner` scratch), then ancestors level by level: parent = left + right in f32.  Lanes sharing a
// parent store identical values, so no sort/unique is needed.  __syncthreads() orders the levels
// (one CU, one L1: workgroup scope is enough).
// ------------------------------------------------------------------------------------------------
constexpr int kUpdateBlock = 1024;

// Three levels per round trip: once the nodes `s` levels above the leaves are final, the item's ancestor A three
// levels further up has its eight descendants of that level contiguous in the heap — one round of loads, seven adds
// (each parent = left + right of the values just formed: the very sums the level-by-level walk stores), and the three
// ancestors on the item's own path are written.  Nodes beside the path are recomputed in registers only: they equal
// what the tree holds (untouched: tree[i] == tree[2i+1] + tree[2i+2] is the tree's invariant) or what the item that
// owns them writes (same operands).  19 levels: 7 rounds of [loads -> barrier] instead of 19.
__device__ __forceinline__ float pick4(const float (&v)[4], int i) {
    return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3];
}

// every thread of the workgroup calls this (leaf1 = leaf index + 1, 0 = nothing to do); the leaves are final
__device__ __forceinline__ void propagate_leaf(float* tree, int levels, int leaf1) {
    {
        int s = 0;
        for (; s + 3 <= levels; s += 3) {
            if (leaf1) {
                const int top1 = leaf1 >> (s + 3);             // 1-based index of A
                const float* d = tree + (top1 << 3) - 1;       // its descendants three levels down
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = d[j];
                const int me = (leaf1 >> s) & 7;
                const float p1[4] = {v[0] + v[1], v[2] + v[3], v[4] + v[5], v[6] + v[7]};
                const float p2[2] = {p1[0] + p1[1], p1[2] + p1[3]};
                tree[(top1 << 2) - 1 + (me >> 1)] = pick4(p1, me >> 1);
                tree[(top1 << 1) - 1 + (me >> 2)] = (me >> 2) ? p2[1] : p2[0];
                tree[top1 - 1] = p2[0] + p2[1];
            }
            __syncthreads();
        }
        for (; s < levels; ++s) {
            if (leaf1) {
                const int node = (leaf1 >> (s + 1)) - 1;
                tree[node] = tree[2 * node + 1] + tree[2 * node + 2];
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void propagate_chunks(float* tree, int levels, int k,
                                                 const int* __restrict__ leaf1_of_item,
                                                 int own_leaf1_first) {
    // leaf1 = leaf index + 1 (0 = dead item).  Ancestor after s steps = (leaf1 >> s) - 1.
    for (int base = 0; base < k; base += blockDim.x) {
        const int i = base + threadIdx.x;
        int leaf1 = 0;
        if (i < k) leaf1 = (base == 0) ? own_leaf1_first : leaf1_of_item[i];
        propagate_leaf(tree, levels, leaf1);
    }
}

__device__ __forceinline__ void sumtree_update_wg(
    float* tree, int capacity, int levels, int k, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_ids, const float* __restrict__ td, float alpha, float td_min,
    float td_max, int mode, int32_t* winner, int32_t* nan_flag, int32_t* item_scratch) {
    // pass 0: NaN screen (the reference raises before touching the tree)
    int bad = 0;
    for (int i = threadIdx.x; i < k; i += blockDim.x) bad |= (td[i] != td[i]);
    if (__syncthreads_or(bad)) {
        if (threadIdx.x == 0) *nan_flag = 1;
        return;
    }
    // pass 1: claim slots (last item wins)
    int own_leaf1 = 0;
    float own_p = 0.f;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const int64_t id = ids[i];
        const int slot = ring_slot(id, capacity);
        const bool live = (slot_ids == nullptr) || (slot_ids[slot] == id);
        float p = td[i];
        if (mode == 0) {
            p = fminf(fmaxf(p, td_min), td_max);                 // np.clip
            p = (float)pow((double)p, (double)alpha);            // np.power(f32, 0.9) -> f32
        }
        const int leaf1 = live ? slot + capacity : 0;            // (slot + C - 1) + 1
        if (live) atomicMax(&winner[slot], i);
        if (i < (int)blockDim.x) {
            own_leaf1 = leaf1;
            own_p = p;
        } else {
            item_scratch[i] = leaf1;
            reinterpret_cast<float*>(item_scratch)[k + i] = p;
        }
    }
    __syncthreads();
    // pass 2: winners write their leaf and release the slot
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const int leaf1 = (i < (int)blockDim.x) ? own_leaf1 : item_scratch[i];
        if (!leaf1) continue;
        const int slot = leaf1 - capacity;
        if (__hip_atomic_load(&winner[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i) {
            tree[leaf1 - 1] = (i < (int)blockDim.x) ? own_p : reinterpret_cast<float*>(item_scratch)[k + i];
            __hip_atomic_store(&winner[slot], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    propagate_chunks(tree, levels, k, item_scratch, own_leaf1);
}


// ------------------------------------------------------------------------------------------------
// The same update for a batch of at most ONE item per thread whose (leaf, TD error) the caller already holds in
// registers (returns.hip k_td_update: the workgroup has just formed the TD errors, and asked for the ids and the id
// map while it was waiting for the return's inputs).  Against sumtree_update_wg this saves the round trips in front of
// the climb: no TD errors / ids / id map to fetch, and the last-writer election runs in LDS instead of an atomicMax
// round and a read-back round on the `winner` scratch (which stays untouched: all -1):
//   * leaves in non-decreasing order over the batch (what the stratified sampler hands out: leaf order = order of
//     the prefix sums): duplicates are neighbours, an item is the last of its leaf iff its successor's differs;
//   * any other order: every item compares its leaf with all the later items' (broadcast reads, four per read).
// `leaves`: LDS, nthreads + 4 ints.  Same leaves, same climb: bit-identical trees.
// `with_leaves()`: the caller's own stores, issued beside the leaves' (one acknowledgement wait for both).
// -> false: a NaN TD error, nothing written (nan_flag raised)
// ------------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ bool sumtree_update_wg_own(float* tree, int levels, int k, int leaf1, float td, float alpha,
                                                      float td_min, float td_max, int32_t* nan_flag, int* leaves,
                                                      int nthreads, F with_leaves) {
    const int i = threadIdx.x;
    // (0 = no leaf: a row whose slot has been overwritten, or no row.  For the order test such an entry stands for
    // "the same as my predecessor": it neither breaks a run of duplicates nor the order)
    leaves[i] = leaf1;
    if (i < 4) leaves[nthreads + i] = 0;
    if (__syncthreads_or(i < k && td != td)) {                 // NaN screen (the reference raises before touching the tree)
        if (i == 0) *nan_flag = 1;
        return false;
    }
    float p = fminf(fmaxf(td, td_min), td_max);                // np.clip
    p = (float)pow((double)p, (double)alpha);                  // np.power(f32, 0.9) -> f32
    // the next item that has a leaf at all (a handful of reads: dead rows are rare) — 0 beyond the batch
    int next = 0;
    for (int j = i + 1; j < nthreads && (next = leaves[j]) == 0; ++j) {}
    const bool unordered = leaf1 != 0 && next != 0 && next < leaf1;
    bool last = leaf1 != 0 && next != leaf1;
    if (__syncthreads_or(unordered)) {
        last = leaf1 != 0;
        if (leaf1) {
            const int4* four = reinterpret_cast<const int4*>(leaves);
#pragma unroll 8
            for (int j = 0; j < nthreads / 4; ++j) {
                const int4 v = four[j];
                const int j0 = 4 * j;
                const bool later = (v.x == leaf1 && j0 > i) | (v.y == leaf1 && j0 + 1 > i) | (v.z == leaf1 && j0 + 2 > i) |
                                   (v.w == leaf1 && j0 + 3 > i);
                last = last & !later;
            }
        }
    }
#ifdef TD_STAMP
    TD_STAMP(5);
#endif
    if (last) tree[leaf1 - 1] = p;
    with_leaves();
    __syncthreads();
#ifdef TD_STAMP
    TD_STAMP(6);
#endif
    propagate_leaf(tree, levels, leaf1);
    return true;
}

}  // namespace asac
