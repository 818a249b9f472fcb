// Sum-tree kernels for gfx950: stratified sample (K1+K2), priority update (K6), episode add,
// leaf max (K8), invariant check.  C ABI in include/asac_hip.h.
//
// Layout in HBM: the reference's array heap, f32[2C-1], root 0, leaves [C-1, 2C-1)
// (reference algorithm/replay_buffer.py:145-167).  At C = 2^19 the tree is 4 MiB: it lives in the
// XCD L2s / Infinity Cache; the top 12 levels (4095 nodes, 16 KiB) are staged into LDS once per
// workgroup for the descent.
#include "asac_common.h"
#include "asac_gather.h"
#include "asac_noise.h"
#include "asac_sidecar.h"
#include "asac_tree_update.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace asac {

int g_launch_repeat = 1;
static char g_err[256] = "";
static std::mutex g_err_mu;

void set_error(hipError_t e, const char* where) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
}

// ------------------------------------------------------------------------------------------------
// K1: one lane per sample.  Binary descent in the reference's exact order: f64 `v` against f32
// nodes promoted to f64; go left when v <= left or right == 0; subtract left when going right.
// ------------------------------------------------------------------------------------------------
constexpr int kSampleBlock = 256;
constexpr int kLdsLevels = 12;                       // nodes of levels 0..11 -> 4095 floats
constexpr int kLdsNodes = (1 << kLdsLevels) - 1;

// STAGE_TOP: the top kLdsNodes nodes of the heap are staged into LDS first (one coalesced round trip,
// then ~12 of the 19 levels cost an LDS read).  That pays for the BASELINE batch (1 workgroup); with
// thousands of workgroups the 16 KB per workgroup would dwarf the useful traffic, and because the
// stratified samples are sorted along the leaves the lanes of a wave walk nearly the same path, so the
// plain loads hit L1 / coalesce anyway.
constexpr int kFusedSampleMax = 1024;   // batches up to here: ONE workgroup samples, reduces and weights

// `DRAW`: the uniforms are not read from `u` but drawn here (prologue_uniform: the numbers the step's noise fill would
// have stored there) and written to `u` for whoever inspects the step's draws
struct SampleDraw {
    uint64_t seed;
    const int64_t* step;
    int64_t n_normal;
};

// `MULTI` (the step prologue's sampler at batches of 257 .. 1024): one workgroup per 256 samples instead of one
// workgroup walking up to four samples per lane — the descent, the Philox draws and above all the f64 power of the
// weights are per-lane serial work, and a second / third / fourth sample per lane cost ~3.9 us each inside a replayed
// step (cfg4: 9.97 us, cfg5: 17.72 us against 6.3 us at 256).  The batch minimum the weights need crosses the (<= 4)
// workgroups through `SampleSync` in device memory: every workgroup publishes its minimum, arrives, and waits until all
// have arrived (they are the first workgroups of the grid: resident before anything else of the launch is, and what
// they wait for is finite work of each other); the last one to LEAVE clears the counters for the next launch and
// publishes beta (nobody reads the old value after that).  Per sample: the same comparisons in the same order.
struct SampleSync {            // min_p_out[2..]: floats [2..5] partial minima, [6] arrivals, [7] departures;
    float part[4];             // all counters 0 between launches
    unsigned int arrived, left;
};
// The exchange uses RELAXED agent-scope atomics only (sc1: performed at the device's coherence point, past the XCD's L2)
// and orders them by waiting for their completion (`s_waitcnt 0`).  An agent-scope RELEASE would do it too, but on this
// part it writes back every dirty line of the XCD's L2 (`buffer_wbl2`) — with the launch's own gather writing 50-100 MB
// at that moment a single release took tens of microseconds, and every gather workgroup issued one (round 5: 81 us
// instead of 30).  Everything the other workgroups read is itself read and written by such atomics.
__device__ __forceinline__ void sync_complete() { __builtin_amdgcn_s_waitcnt(0); }
__device__ __forceinline__ void sync_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sync_load(float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool FUSE_WEIGHTS, bool STAGE_TOP, bool DRAW = false, bool EXPLICIT = false, int TRIPS = kFusedSampleMax / kSampleBlock,
          bool MULTI = false, int LDS_LEVELS = kLdsLevels, bool PARTIAL = false>
__device__ __forceinline__ void sumtree_sample_body(
    const float* __restrict__ tree, int capacity, int levels, int batch,
    double* __restrict__ u, const int64_t* __restrict__ slot_ids, double* beta_state,
    double beta_increment, int32_t* __restrict__ leaf_out, float* __restrict__ p_out,
    int64_t* __restrict__ ids_out, float* __restrict__ w_out, float* min_p_out, const SampleDraw draw = SampleDraw{},
    const int32_t* __restrict__ owner = nullptr, int rank = 0) {
    constexpr int kLdsNodes = (1 << LDS_LEVELS) - 1;       // (shadows the default: this instantiation's staged levels)
    __shared__ __attribute__((aligned(16))) float top[kLdsNodes + 1];
    __shared__ float red[kSampleBlock / kWave];
    constexpr int kTrips = FUSE_WEIGHTS ? TRIPS : 1;
    const int first = blockIdx.x * kSampleBlock + threadIdx.x;

    const int n_lds = STAGE_TOP ? min(kLdsNodes, 2 * capacity - 1) : 0;
    // the tree's top travels to LDS as four 16-byte loads per thread; the lane's uniforms (ten Philox rounds each) are
    // drawn while they are in flight
    const bool top_vec = STAGE_TOP && n_lds == kLdsNodes && 2 * capacity - 1 > kLdsNodes &&
                         (reinterpret_cast<uintptr_t>(tree) & 15) == 0;
    typedef float st4 __attribute__((ext_vector_type(4)));
    constexpr int kStage = (kLdsNodes + 1) / 4 / kSampleBlock;
    st4 st[kStage];
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
        st[k] = st4{0.f, 0.f, 0.f, 0.f};
        if (top_vec) st[k] = reinterpret_cast<const st4*>(tree)[threadIdx.x + k * kSampleBlock];
    }
    // (requested at entry: read after the batch-minimum barrier it was a dependent round trip in front of the weights)
    const double beta_old = (FUSE_WEIGHTS && w_out) ? *beta_state : 0.0;
    const double beta_old_any = beta_old;
    double ui_[kTrips];
#pragma unroll
    for (int trip = 0; trip < kTrips; ++trip) {
        const int i = first + trip * kSampleBlock;
        ui_[trip] = 0.0;
        if (i >= batch) continue;
        if (DRAW) {
            ui_[trip] = prologue_uniform(draw.seed, (uint64_t)*draw.step, draw.n_normal, i);
            u[i] = ui_[trip];
        } else {
            ui_[trip] = u[i];
        }
    }
    if (STAGE_TOP) {
        if (top_vec) {
#pragma unroll
            for (int k = 0; k < kStage; ++k) reinterpret_cast<st4*>(top)[threadIdx.x + k * kSampleBlock] = st[k];
        } else {
            for (int i = threadIdx.x; i < n_lds; i += kSampleBlock) top[i] = tree[i];
        }
        __syncthreads();
    }

    const float root = STAGE_TOP ? top[0] : tree[0];
    // the fused single-workgroup form strides over the batch (up to kFusedSampleMax samples: TRIPS samples per lane), the
    // multi-workgroup form handles one sample per lane.  A lane's samples descend IN LOCKSTEP, level by level: their
    // chains of dependent reads are independent of each other, so the loads of all of them travel together and a batch
    // of 512 / 1024 costs the round trips of a batch of 256 (walked one after the other, each sample's `s_waitcnt` also
    // drained its neighbours' loads).  Per sample the comparisons and their order are unchanged.
    float pmin = INFINITY;
    float p_reg[kTrips];
    double v_[kTrips];
    int node_[kTrips];
    bool live_[kTrips];
#pragma unroll
    for (int trip = 0; trip < kTrips; ++trip) {
        const int i = first + trip * kSampleBlock;
        p_reg[trip] = root;
        node_[trip] = 0;
        v_[trip] = 0.0;
        live_[trip] = i < batch;
        if (!live_[trip]) continue;
        const float seg = root / (float)batch;                 // np.float32(root / B)
        const double lo = (double)i * (double)seg;             // int64 * float32 -> float64
        const double hi = (double)(i + 1) * (double)seg;
        v_[trip] = lo + (hi - lo) * ui_[trip];                 // np.random.uniform(lo, hi)
        if (EXPLICIT) v_[trip] = ui_[trip];                    // the caller's own values (asac_sumtree_descend)
        if (EXPLICIT && owner && owner[i] != rank) {           // another shard's sample: nothing of it lives here
            leaf_out[i] = -1, p_out[i] = 0.f, ids_out[i] = -1;
            live_[trip] = false;
        }
    }
    // one level of the reference's descent: left/right sums a, b of the current node's children
    auto step = [&](int t, float a, float b) -> int {
        const bool go_left = (v_[t] <= (double)a) || (b == 0.0f);
        if (!go_left) v_[t] -= (double)a;
        p_reg[t] = go_left ? a : b;
        return go_left ? 0 : 1;
    };
    int l = 0;
    // levels held in LDS (whole levels are staged: the test is the same for every node of a level)
    for (; l < levels; ++l) {
        if (!(STAGE_TOP && (4 << l) - 2 < n_lds)) break;      // the largest right child of level l is 2^(l+2) - 2
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int left = 2 * node_[t] + 1;
            node_[t] = left + step(t, top[left], top[left + 1]);
        }
    }
    // levels in memory, three per round trip: in the array heap the children (2), grandchildren (4) and
    // great-grandchildren (8) of a node are each contiguous, so all 14 loads are issued together and the
    // three decisions run on registers — same comparisons, a third of the dependent latencies
    for (; l + 3 <= levels; l += 3) {
        // (named vector registers, not arrays: a select chain over an array element is turned back into an indexed load
        // of a scratch copy)
        typedef float v2f __attribute__((ext_vector_type(2)));
        typedef float v4f __attribute__((ext_vector_type(4)));
        v2f a1[kTrips];
        v4f a2[kTrips], a3l[kTrips], a3h[kTrips];
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int64_t node = live_[t] ? node_[t] : 0;      // (a lane without a sample re-reads the top: in range)
            const float* c1 = tree + 2 * node + 1;
            const float* c2 = tree + 4 * node + 3;
            const float* c3 = tree + 8 * node + 7;
            a1[t] = v2f{c1[0], c1[1]};
            a2[t] = v4f{c2[0], c2[1], c2[2], c2[3]};
            a3l[t] = v4f{c3[0], c3[1], c3[2], c3[3]};
            a3h[t] = v4f{c3[4], c3[5], c3[6], c3[7]};
        }
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int i1 = step(t, a1[t].x, a1[t].y);
            const int i2 = 2 * i1 + step(t, i1 ? a2[t].z : a2[t].x, i1 ? a2[t].w : a2[t].y);
            const v4f q4 = (i2 & 2) ? a3h[t] : a3l[t];
            const float l3 = (i2 & 1) ? q4.z : q4.x;
            const float r3 = (i2 & 1) ? q4.w : q4.y;
            node_[t] = 8 * node_[t] + 7 + 2 * i2 + step(t, l3, r3);
        }
    }
    // two levels left: children and grandchildren in one round trip
    for (; l + 2 <= levels; l += 2) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        typedef float v4f __attribute__((ext_vector_type(4)));
        v2f a1[kTrips];
        v4f a2[kTrips];
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int64_t node = live_[t] ? node_[t] : 0;
            const float* c1 = tree + 2 * node + 1;
            const float* c2 = tree + 4 * node + 3;
            a1[t] = v2f{c1[0], c1[1]};
            a2[t] = v4f{c2[0], c2[1], c2[2], c2[3]};
        }
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int i1 = step(t, a1[t].x, a1[t].y);
            node_[t] = 4 * node_[t] + 3 + 2 * i1 + step(t, i1 ? a2[t].z : a2[t].x, i1 ? a2[t].w : a2[t].y);
        }
    }
    for (; l < levels; ++l) {
        float ab[kTrips][2];
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int left = live_[t] ? 2 * node_[t] + 1 : 1;
            ab[t][0] = tree[left], ab[t][1] = tree[left + 1];
        }
#pragma unroll
        for (int t = 0; t < kTrips; ++t) node_[t] = 2 * node_[t] + 1 + step(t, ab[t][0], ab[t][1]);
    }
    int64_t id_reg[kTrips];
#pragma unroll
    for (int trip = 0; trip < kTrips; ++trip) {
        const int i = first + trip * kSampleBlock;
        id_reg[trip] = -1;
        if (!live_[trip]) {
            p_reg[trip] = 0.f;
            continue;
        }
        const float p = levels == 0 ? tree[0] : p_reg[trip];
        leaf_out[i] = node_[trip];
        p_out[i] = p;
        id_reg[trip] = slot_ids[node_[trip] - (capacity - 1)];     // (stored at the end: the lookup travels under the
        pmin = fminf(pmin, p);                                     //  reduction and the weights)
        p_reg[trip] = p;
    }

    // batch minimum of p (per block -> global)
    float m = wave_min(pmin);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
    __syncthreads();
    SampleSync* const sync = MULTI ? reinterpret_cast<SampleSync*>(min_p_out + 2) : nullptr;
    const unsigned n_wg = MULTI ? (unsigned)((batch + kSampleBlock - 1) / kSampleBlock) : 1u;
    if (PARTIAL) {
        // the weights are formed by a workgroup of the NEXT launch (asac_window_gather_pad_w, beside the gather): this
        // workgroup's minimum is all that is left to hand over — no exchange, no power, no beta here
        if (threadIdx.x == 0) {
            float bm = red[0];
            for (int w = 1; w < kSampleBlock / kWave; ++w) bm = fminf(bm, red[w]);
            sync->part[blockIdx.x] = bm;
        }
#pragma unroll
        for (int trip = 0; trip < kTrips; ++trip) {
            const int i = first + trip * kSampleBlock;
            if (live_[trip]) ids_out[i] = id_reg[trip];
        }
        return;
    }
    if (MULTI) {
        if (threadIdx.x == 0) {
            float bm = red[0];
            for (int w = 1; w < kSampleBlock / kWave; ++w) bm = fminf(bm, red[w]);
            sync_store(&sync->part[blockIdx.x], bm);
            sync_complete();
            __hip_atomic_fetch_add(&sync->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&sync->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_wg)
                __builtin_amdgcn_s_sleep(1);
            bm = sync_load(&sync->part[0]);
            for (unsigned w = 1; w < n_wg; ++w) bm = fminf(bm, sync_load(&sync->part[w]));
            red[0] = bm;
        }
    } else if (threadIdx.x == 0) {
        float bm = red[0];
        for (int w = 1; w < kSampleBlock / kWave; ++w) bm = fminf(bm, red[w]);
        red[0] = bm;
        if (FUSE_WEIGHTS && !w_out) {
            // weights deferred (sharded replay: they are normalised by the minimum sampling ratio over ALL ranks'
            // shards): min p and this shard's ratio out, beta untouched — asac_per_is_weights finishes after the MIN
            min_p_out[0] = bm;
            min_p_out[1] = bm / root;
        } else if (FUSE_WEIGHTS) {
            *beta_state = fmin(1.0, beta_old + beta_increment);
            *min_p_out = bm;
        } else if (w_out) {
            // two-pass weights: the weight buffer doubles as the per-workgroup minimum scratch until the weight
            // kernel overwrites it (thousands of workgroups hammering one atomic serialise at the L2)
            w_out[blockIdx.x] = bm;
        } else {
            // priorities are >= 0, so the unsigned bit pattern orders like the float
            atomicMin(reinterpret_cast<unsigned int*>(min_p_out), __float_as_uint(bm));
        }
    }
    if (MULTI) {
        __syncthreads();
        // every workgroup holds the batch minimum now; the last one to leave publishes it (and beta) and clears the counters
        if (threadIdx.x == 0) {
            const float bm = red[0];
            const unsigned gone = __hip_atomic_fetch_add(&sync->left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gone == n_wg - 1) {
                min_p_out[0] = bm;
                if (w_out) *beta_state = fmin(1.0, beta_old_any + beta_increment);
                else min_p_out[1] = bm / root;
                __hip_atomic_store(&sync->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync->left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (FUSE_WEIGHTS && w_out) {
        if (!MULTI) __syncthreads();
        const float min_ratio = red[0] / root;                 // min(p/total) == min(p)/total
        const double b = fmin(1.0, beta_old + beta_increment);
#pragma unroll
        for (int trip = 0; trip < kTrips; ++trip) {
            const int i = first + trip * kSampleBlock;
            if (i < batch) w_out[i] = is_weight(p_reg[trip], root, min_ratio, b);
        }
    }
#pragma unroll
    for (int trip = 0; trip < kTrips; ++trip) {
        const int i = first + trip * kSampleBlock;
        if (live_[trip]) ids_out[i] = id_reg[trip];
    }
}

template <bool FUSE_WEIGHTS, bool STAGE_TOP>
__global__ __launch_bounds__(kSampleBlock) void k_sumtree_sample(
    const float* __restrict__ tree, int capacity, int levels, int batch,
    const double* __restrict__ u, const int64_t* __restrict__ slot_ids, double* beta_state,
    double beta_increment, int32_t* __restrict__ leaf_out, float* __restrict__ p_out,
    int64_t* __restrict__ ids_out, float* __restrict__ w_out, float* min_p_out) {
    sumtree_sample_body<FUSE_WEIGHTS, STAGE_TOP>(tree, capacity, levels, batch, const_cast<double*>(u), slot_ids, beta_state,
                                                 beta_increment, leaf_out, p_out, ids_out, w_out, min_p_out);
}

// The first launch of a captured train step: workgroup 0 is the fused single-workgroup sampler (drawing its own
// stratified uniforms), the others are the step prologue's workers (Polyak, gradient memset, Gaussian draws, ensemble
// subsets): nothing the sampler reads is written by them.
template <bool PARTIAL>
__global__ __launch_bounds__(kSampleBlock) void k_prologue_sample(
    const PrologueArgs pa, const float* __restrict__ tree, int capacity, int levels, int batch,
    const int64_t* __restrict__ slot_ids, double* beta_state, double beta_increment, int32_t* __restrict__ leaf_out,
    float* __restrict__ p_out, int64_t* __restrict__ ids_out, float* __restrict__ w_out, float* min_p_out) {
    // workgroups [0, n_s): the sampler, 256 samples each (n_s = 1 for the BASELINE batch); the others: the step
    // prologue's workers
    const int n_s = (batch + kSampleBlock - 1) / kSampleBlock;
    if ((int)blockIdx.x < n_s) {
        const SampleDraw dr{pa.seed, pa.step, pa.n_normal};
        // (thirteen staged levels, 32 KB: trees of 2^16 / 2^19 leaves keep 3 / 6 levels for memory — one / two round trips
        // of three levels each, one less than with twelve; the launch has a handful of other workgroups, LDS is free)
        if (n_s == 1)
            sumtree_sample_body<true, true, true, false, 1, false, 13>(
                tree, capacity, levels, batch, pa.u, slot_ids, beta_state, beta_increment, leaf_out, p_out, ids_out, w_out,
                min_p_out, dr);
        else
            sumtree_sample_body<true, true, true, false, 1, true, 13, PARTIAL>(
                tree, capacity, levels, batch, pa.u, slot_ids, beta_state, beta_increment, leaf_out, p_out, ids_out, w_out,
                min_p_out, dr);
    } else {
        prologue_block(pa, (int)blockIdx.x - n_s, true);
    }
}

// the descent alone for explicit f64 values (sharded "parity" sampling: the residual values of the top-level walk)
__global__ __launch_bounds__(kSampleBlock) void k_sumtree_descend(
    const float* __restrict__ tree, int capacity, int levels, int n, const double* __restrict__ v,
    const int64_t* __restrict__ slot_ids, int32_t* __restrict__ leaf_out, float* __restrict__ p_out,
    int64_t* __restrict__ ids_out, float* scratch_min) {
    sumtree_sample_body<false, false, false, true>(tree, capacity, levels, n, const_cast<double*>(v), slot_ids, nullptr, 0.0,
                                                   leaf_out, p_out, ids_out, nullptr, scratch_min);
}

// ... of the samples this shard owns only (the others get leaf -1, priority 0, id -1)
__global__ __launch_bounds__(kSampleBlock) void k_sumtree_descend_owned(
    const float* __restrict__ tree, int capacity, int levels, int n, const double* __restrict__ v,
    const int32_t* __restrict__ owner, int rank, const int64_t* __restrict__ slot_ids, int32_t* __restrict__ leaf_out,
    float* __restrict__ p_out, int64_t* __restrict__ ids_out, float* scratch_min) {
    sumtree_sample_body<false, false, false, true>(tree, capacity, levels, n, const_cast<double*>(v), slot_ids, nullptr, 0.0,
                                                   leaf_out, p_out, ids_out, nullptr, scratch_min, SampleDraw{}, owner, rank);
}

// The TOP of a sharded sum tree (SURVEY.md section 8e "parity"): the G = 2^k shard roots are the leaves of a heap whose
// parents are formed like the reference's (left + right in f32, replay_buffer.py:172-183); every sample of the GLOBAL
// batch draws its stratified value over the global root and walks these k levels with the reference's comparisons
// (f64 value against f32 sums, 196-205) -> the owning shard and the residual value its own tree continues with.
// Every rank runs this on the same roots and uniforms and gets the same plan.
constexpr int kMaxShards = 64;
__global__ __launch_bounds__(kSampleBlock) void k_sumtree_plan_top(const float* __restrict__ roots, int G, int levels_top,
                                                                   int batch, const double* __restrict__ u,
                                                                   int32_t* __restrict__ owner_out, double* __restrict__ v_out,
                                                                   float* __restrict__ total_out) {
    __shared__ float heap[2 * kMaxShards];
    if ((int)threadIdx.x < G) heap[G - 1 + threadIdx.x] = roots[threadIdx.x];
    __syncthreads();
    for (int l = levels_top - 1; l >= 0; --l) {          // parents of level l, bottom up
        const int first = (1 << l) - 1, count = 1 << l;
        if ((int)threadIdx.x < count) {
            const int node = first + threadIdx.x;
            heap[node] = heap[2 * node + 1] + heap[2 * node + 2];
        }
        __syncthreads();
    }
    const float root = heap[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) *total_out = root;
    const int i = blockIdx.x * kSampleBlock + threadIdx.x;
    if (i >= batch) return;
    const float seg = root / (float)batch;                 // np.float32(root / B)
    const double lo = (double)i * (double)seg, hi = (double)(i + 1) * (double)seg;
    double v = lo + (hi - lo) * u[i];
    int node = 0;
    for (int l = 0; l < levels_top; ++l) {
        const int left = 2 * node + 1;
        const float a = heap[left], b = heap[left + 1];
        const bool go_left = (v <= (double)a) || (b == 0.0f);
        if (!go_left) v -= (double)a;
        node = go_left ? left : left + 1;
    }
    owner_out[i] = node - (G - 1);
    v_out[i] = v;
}

// IS weights of the rows [lo, lo + per) of a GLOBAL batch whose leaf priorities p_all[n_all] and total are at hand (the
// sharded parity sampling after its all-reduce): the minimum runs over the whole batch, beta advances first
// (replay_buffer.py:352-354).  One workgroup.
__global__ __launch_bounds__(256) void k_is_weights_slice(const float* __restrict__ p_all, int n_all, int lo, int per,
                                                          const float* __restrict__ total, double* beta_state,
                                                          double beta_increment, float* __restrict__ w_out) {
    __shared__ float red[4];
    float m = INFINITY;
    for (int i = threadIdx.x; i < n_all; i += blockDim.x) m = fminf(m, p_all[i]);
    m = wave_min(m);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
    const double b = fmin(1.0, *beta_state + beta_increment);
    const float tot = *total;
    __syncthreads();
    const float min_ratio = fminf(fminf(red[0], red[1]), fminf(red[2], red[3])) / tot;
    if (threadIdx.x == 0) *beta_state = b;
    for (int i = threadIdx.x; i < per; i += blockDim.x) w_out[i] = is_weight(p_all[lo + i], tot, min_ratio, b);
}

__global__ void k_fill_u32(unsigned int* p, unsigned int v) { *p = v; }

__global__ void k_is_weights(const float* __restrict__ p, int batch, const float* total,
                             const float* min_ratio, double* beta_state, double beta_increment,
                             float* __restrict__ w_out, int advance_beta) {
    // every block recomputes the advanced beta from the OLD value; block 0 publishes it last
    const double b = advance_beta ? fmin(1.0, *beta_state + beta_increment) : *beta_state;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch) w_out[i] = is_weight(p[i], *total, *min_ratio, b);
}

// ... the same for batches of one workgroup, beta advanced by the launch itself (every thread has read the old value)
__global__ __launch_bounds__(256) void k_is_weights_advance(const float* __restrict__ p, int batch, const float* total,
                                                            const float* min_ratio, double* beta_state,
                                                            double beta_increment, float* __restrict__ w_out) {
    const double b = fmin(1.0, *beta_state + beta_increment);
    const float tot = *total, mr = *min_ratio;
    __syncthreads();
    if (threadIdx.x == 0) *beta_state = b;
    for (int i = threadIdx.x; i < batch; i += blockDim.x) w_out[i] = is_weight(p[i], tot, mr, b);
}

__global__ void k_advance_beta(double* beta_state, double beta_increment) {
    *beta_state = fmin(1.0, *beta_state + beta_increment);
}

// min_ratio from min_p / total, for the two-pass single-GPU path
__global__ void k_ratio(const float* min_p, const float* total, float* out) { *out = *min_p / *total; }

// min over the per-workgroup minima -> out[0] = min p, out[1] = min p / total
__global__ __launch_bounds__(256) void k_ratio_partials(const float* __restrict__ partial, int n,
                                                        const float* __restrict__ total, float* __restrict__ out) {
    float m = INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fminf(m, partial[i]);
    m = wave_min(m);
    __shared__ float red[4];
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
        out[0] = m;
        out[1] = m / *total;
    }
}

// K6 / add: the single-workgroup update is device code in asac_tree_update.h (shared with returns.hip)

__global__ __launch_bounds__(kUpdateBlock) void k_sumtree_update(
    float* tree, int capacity, int levels, int k, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_ids, const float* __restrict__ td, float alpha, float td_min,
    float td_max, int mode, int32_t* winner, int32_t* nan_flag, int32_t* item_scratch) {
    sumtree_update_wg(tree, capacity, levels, k, ids, slot_ids, td, alpha, td_min, td_max, mode, winner, nan_flag, item_scratch);
}

// ... with sidecar jobs (asac_sidecar.h) as workgroups 1.. of the launch
template <int NSC>
__global__ __launch_bounds__(kUpdateBlock) void k_sumtree_update_sc(
    float* tree, int capacity, int levels, int k, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_ids, const float* __restrict__ td, float alpha, float td_min,
    float td_max, int mode, int32_t* winner, int32_t* nan_flag, int32_t* item_scratch, const SidecarsT<NSC> sc) {
    if (blockIdx.x > 0) {
        __shared__ float sc_red[256];
        sidecar_run(sc, (int)blockIdx.x - 1, sc_red);
        return;
    }
    sumtree_update_wg(tree, capacity, levels, k, ids, slot_ids, td, alpha, td_min, td_max, mode, winner, nan_flag, item_scratch);
}

__global__ __launch_bounds__(kUpdateBlock) void k_per_add(
    float* tree, int capacity, int levels, int64_t first_id, int count, int ignore_size,
    const float* max_p_dev, float max_p_host, int64_t* slot_ids) {
    const float max_p = max_p_dev ? *max_p_dev : max_p_host;
    const int64_t max_id = 10ll * capacity;
    // when an episode is longer than the ring only its last C rows survive (later rows overwrite)
    const int first_live = count > capacity ? count - capacity : 0;
    for (int j = first_live + threadIdx.x; j < count; j += blockDim.x) {
        const int64_t id = (first_id + j) % max_id;
        const int slot = (int)(id % capacity);
        float p = max_p;
        if (j >= count - ignore_size || slot >= capacity - ignore_size) p = 0.f;
        slot_ids[slot] = id;
        tree[slot + capacity - 1] = p;
    }
    __syncthreads();
    for (int base = first_live; base < count; base += blockDim.x) {
        const int j = base + threadIdx.x;
        int leaf1 = 0;
        if (j < count) leaf1 = (int)(((first_id + j) % max_id) % capacity) + capacity;
        propagate_leaf(tree, levels, leaf1);
    }
}

// ------------------------------------------------------------------------------------------------
// K8: max over the leaves; 16-byte loads, wave reduce, one atomic per block.  Priorities >= 0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_leaf_max(const float* __restrict__ leaves, int n,
                                                  unsigned int* out) {
    float m = 0.f;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = gridDim.x * blockDim.x;
    // leaves start at element C-1 of the tree: peel to a 16-byte boundary
    const uintptr_t addr = reinterpret_cast<uintptr_t>(leaves);
    int head = (int)(((16 - (addr & 15)) & 15) / 4);
    if (head > n) head = n;
    if (tid < head) m = fmaxf(m, leaves[tid]);
    const float4* v = reinterpret_cast<const float4*>(leaves + head);
    const int n4 = (n - head) / 4;
    for (int i = tid; i < n4; i += stride) {
        const float4 x = v[i];
        m = fmaxf(fmaxf(m, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
    }
    const int tail0 = head + n4 * 4;
    if (tid < n - tail0) m = fmaxf(m, leaves[tail0 + tid]);
    m = wave_max(m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(out, __float_as_uint(m));
    }
}

__global__ void k_tree_check(const float* __restrict__ tree, int capacity, int32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < capacity - 1) {
        if (tree[i] != tree[2 * i + 1] + tree[2 * i + 2]) atomicAdd(out, 1);
    }
}

}  // namespace asac

using namespace asac;

extern "C" {

int asac_version(void) { return ASAC_ABI_VERSION; }

// sizeof of the by-value structs of the ABI as this library was compiled (bindings check their mirrors against it)
int64_t asac_struct_size(const char* name) {
#define ASAC_SZ(T) if (!strcmp(name, #T)) return (int64_t)sizeof(T)
    ASAC_SZ(asac_gather_key_t);
    ASAC_SZ(asac_partial_sum_t);
    ASAC_SZ(asac_row_move_t);
    ASAC_SZ(asac_sidecar_t);
    ASAC_SZ(asac_squash_job_t);
    ASAC_SZ(asac_vtrace_args_t);
    ASAC_SZ(asac_mlp_desc_t);
    ASAC_SZ(asac_mlp_job_t);
    ASAC_SZ(asac_pi_q_job_t);
    ASAC_SZ(asac_mlp_sample_epilogue_t);
    ASAC_SZ(asac_gru_desc_t);
    ASAC_SZ(asac_conv2_desc_t);
    ASAC_SZ(asac_obs_decoder_params_t);
#undef ASAC_SZ
    return -1;
}

const char* asac_last_error(void) { return g_err; }

int asac_set_launch_repeat(int repeat) {
    const int old = g_launch_repeat;
    g_launch_repeat = repeat < 1 ? 1 : repeat;
    return old;
}

int asac_sumtree_sample(const float* tree, int capacity, int batch, const double* u,
                        const int64_t* slot_ids, double* beta_state, double beta_increment,
                        int32_t* leaf_out, float* p_out, int64_t* ids_out, float* is_weights_out,
                        float* min_p_out, void* stream) {
    if (capacity <= 0 || (capacity & (capacity - 1)) || batch <= 0 || !min_p_out)
        return bad_arg("asac_sumtree_sample");
    hipStream_t s = as_stream(stream);
    const int levels = ilog2(capacity);
    const int blocks = (batch + kSampleBlock - 1) / kSampleBlock;
    if (batch <= kFusedSampleMax && is_weights_out) {
        ASAC_LAUNCH((k_sumtree_sample<true, true>), dim3(1), dim3(kSampleBlock), 0, s, tree, capacity,
                           levels, batch, u, slot_ids, beta_state, beta_increment, leaf_out, p_out,
                           ids_out, is_weights_out, min_p_out);
        return finish_launch("asac_sumtree_sample");
    }
    if (!is_weights_out)
        ASAC_LAUNCH(k_fill_u32, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned int*>(min_p_out),
                    0x7f800000u /* +inf */);
    if (blocks <= 32) {
        ASAC_LAUNCH((k_sumtree_sample<false, true>), dim3(blocks), dim3(kSampleBlock), 0, s, tree, capacity,
                    levels, batch, u, slot_ids, beta_state, beta_increment, leaf_out, p_out, ids_out,
                    is_weights_out, min_p_out);
    } else {
        ASAC_LAUNCH((k_sumtree_sample<false, false>), dim3(blocks), dim3(kSampleBlock), 0, s, tree, capacity,
                    levels, batch, u, slot_ids, beta_state, beta_increment, leaf_out, p_out, ids_out,
                    is_weights_out, min_p_out);
    }
    if (is_weights_out) {
        // two-pass weights: min_p_out[1] := min_p / root, then the stand-alone weight kernel
        ASAC_LAUNCH(k_ratio_partials, dim3(1), dim3(256), 0, s, is_weights_out, blocks, tree, min_p_out);
        ASAC_LAUNCH(k_is_weights, dim3(blocks), dim3(kSampleBlock), 0, s, p_out, batch, tree,
                           min_p_out + 1, beta_state, beta_increment, is_weights_out, 1);
        ASAC_LAUNCH(k_advance_beta, dim3(1), dim3(1), 0, s, beta_state, beta_increment);
    }
    return finish_launch("asac_sumtree_sample");
}

int asac_sumtree_descend(const float* tree, int capacity, int n, const double* values, const int64_t* slot_ids,
                         int32_t* leaf_out, float* p_out, int64_t* ids_out, void* stream) {
    if (capacity <= 0 || (capacity & (capacity - 1)) || n <= 0 || !tree || !values || !slot_ids || !leaf_out || !p_out ||
        !ids_out)
        return bad_arg("asac_sumtree_descend");
    static float* scratch = nullptr;      // the body's per-launch minimum lands here (unused)
    if (!scratch && hipMalloc(reinterpret_cast<void**>(&scratch), sizeof(float)) != hipSuccess)
        return bad_arg("asac_sumtree_descend: scratch");
    ASAC_LAUNCH(k_sumtree_descend, dim3((n + kSampleBlock - 1) / kSampleBlock), dim3(kSampleBlock), 0, as_stream(stream),
                tree, capacity, ilog2(capacity), n, values, slot_ids, leaf_out, p_out, ids_out, scratch);
    return finish_launch("asac_sumtree_descend");
}

int asac_sumtree_descend_owned(const float* tree, int capacity, int n, const double* values, const int32_t* owner,
                               int rank, const int64_t* slot_ids, int32_t* leaf_out, float* p_out, int64_t* ids_out,
                               void* stream) {
    if (capacity <= 0 || (capacity & (capacity - 1)) || n <= 0 || !tree || !values || !owner || !slot_ids || !leaf_out ||
        !p_out || !ids_out)
        return bad_arg("asac_sumtree_descend_owned");
    static float* scratch = nullptr;      // the body's per-launch minimum lands here (unused)
    if (!scratch && hipMalloc(reinterpret_cast<void**>(&scratch), sizeof(float)) != hipSuccess)
        return bad_arg("asac_sumtree_descend_owned: scratch");
    ASAC_LAUNCH(k_sumtree_descend_owned, dim3((n + kSampleBlock - 1) / kSampleBlock), dim3(kSampleBlock), 0,
                as_stream(stream), tree, capacity, ilog2(capacity), n, values, owner, rank, slot_ids, leaf_out, p_out,
                ids_out, scratch);
    return finish_launch("asac_sumtree_descend_owned");
}

int asac_sumtree_plan_top(const float* shard_roots, int n_shards, int batch, const double* u, int32_t* owner_out,
                          double* value_out, float* total_out, void* stream) {
    if (n_shards <= 0 || n_shards > kMaxShards || (n_shards & (n_shards - 1)) || batch <= 0 || !shard_roots || !u ||
        !owner_out || !value_out || !total_out)
        return bad_arg("asac_sumtree_plan_top");
    ASAC_LAUNCH(k_sumtree_plan_top, dim3((batch + kSampleBlock - 1) / kSampleBlock), dim3(kSampleBlock), 0,
                as_stream(stream), shard_roots, n_shards, ilog2(n_shards), batch, u, owner_out, value_out, total_out);
    return finish_launch("asac_sumtree_plan_top");
}

int asac_per_is_weights_slice(const float* p_all, int n_all, int first, int count, const float* total, double* beta_state,
                              double beta_increment, float* is_weights_out, void* stream) {
    if (n_all <= 0 || first < 0 || count <= 0 || first + count > n_all || !p_all || !total || !beta_state || !is_weights_out)
        return bad_arg("asac_per_is_weights_slice");
    // (not idempotent: under the measurement repeat knob beta advances in the first repetition only)
    for (int rep = 0; rep < g_launch_repeat; ++rep)
        hipLaunchKernelGGL(k_is_weights_slice, dim3(1), dim3(256), 0, as_stream(stream), p_all, n_all, first, count, total,
                           beta_state, rep == 0 ? beta_increment : 0.0, is_weights_out);
    return finish_launch("asac_per_is_weights_slice");
}

static int prologue_sample_launch(float* target, const float* source, int64_t n_polyak, float tau, float* zero_out,
                                  int64_t n_zero, uint64_t seed, const int64_t* step_counter, double* uniform_out,
                                  float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets, int E_sample,
                                  int E, const float* tree, int capacity, int batch, const int64_t* slot_ids,
                                  double* beta_state, double beta_increment, int32_t* leaf_out, float* p_out,
                                  int64_t* ids_out, float* is_weights_out, float* min_p_out, bool partial, void* stream) {
    if (!step_counter || n_normal < 0 || n_subsets < 0 || n_polyak < 0 || n_zero < 0 || !uniform_out ||
        (n_polyak > 0 && (!target || !source)) || (n_zero > 0 && !zero_out) || (n_normal > 0 && !normal_out))
        return bad_arg("asac_step_prologue_sample");
    if (n_subsets > 0 && (!subsets_out || E_sample < 1 || E_sample > E || E > ASAC_MAX_ENSEMBLE))
        return bad_arg("asac_step_prologue_sample: subsets");
    if (capacity <= 0 || (capacity & (capacity - 1)) || batch <= 0 || batch > kFusedSampleMax ||
        !min_p_out || !tree || !slot_ids || !beta_state)
        return bad_arg("asac_step_prologue_sample: sampler");
    const int64_t lanes = (n_normal + 3) / 4 + (batch + 1) / 2 + n_subsets;
    const int64_t pb = prologue_span_blocks(n_polyak), zb = prologue_span_blocks(n_zero);
    const float one_m_tau = (float)(1.0 - (double)tau);   // python: (1. - tau) in double, cast by ATen
    // under the measurement repeat knob Polyak and the beta advance (not idempotent) run in the first repetition only
    for (int rep = 0; rep < g_launch_repeat; ++rep) {
        const int blocks_p = rep == 0 ? (int)pb : 0;
        const PrologueArgs a{seed, step_counter, uniform_out, batch, normal_out, n_normal, subsets_out, n_subsets, E_sample,
                             E, blocks_p, target, source, n_polyak, one_m_tau, tau, (int)zb, zero_out, n_zero};
        const dim3 grid((unsigned)((batch + kSampleBlock - 1) / kSampleBlock + blocks_p + zb + (lanes + 255) / 256));
        if (partial)
            hipLaunchKernelGGL(k_prologue_sample<true>, grid, dim3(kSampleBlock), 0, as_stream(stream), a, tree, capacity,
                               ilog2(capacity), batch, slot_ids, beta_state, 0.0, leaf_out, p_out, ids_out, is_weights_out,
                               min_p_out);
        else
            hipLaunchKernelGGL(k_prologue_sample<false>, grid, dim3(kSampleBlock), 0, as_stream(stream), a, tree, capacity,
                               ilog2(capacity), batch, slot_ids, beta_state, rep == 0 ? beta_increment : 0.0, leaf_out, p_out,
                               ids_out, is_weights_out, min_p_out);
    }
    return finish_launch("asac_step_prologue_sample");
}

int asac_step_prologue_sample(float* target, const float* source, int64_t n_polyak, float tau, float* zero_out,
                              int64_t n_zero, uint64_t seed, const int64_t* step_counter, double* uniform_out,
                              float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets, int E_sample,
                              int E, const float* tree, int capacity, int batch, const int64_t* slot_ids,
                              double* beta_state, double beta_increment, int32_t* leaf_out, float* p_out,
                              int64_t* ids_out, float* is_weights_out, float* min_p_out, void* stream) {
    return prologue_sample_launch(target, source, n_polyak, tau, zero_out, n_zero, seed, step_counter, uniform_out, normal_out,
                                  n_normal, subsets_out, n_subsets, E_sample, E, tree, capacity, batch, slot_ids, beta_state,
                                  beta_increment, leaf_out, p_out, ids_out, is_weights_out, min_p_out, false, stream);
}

int asac_step_prologue_sample_partial(float* target, const float* source, int64_t n_polyak, float tau, float* zero_out,
                                      int64_t n_zero, uint64_t seed, const int64_t* step_counter, double* uniform_out,
                                      float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets, int E_sample,
                                      int E, const float* tree, int capacity, int batch, const int64_t* slot_ids,
                                      int32_t* leaf_out, float* p_out, int64_t* ids_out, float* min_p_out, void* stream) {
    if (batch <= kSampleBlock) return bad_arg("asac_step_prologue_sample_partial: one workgroup samples such a batch whole");
    static double unused_beta = 0.0;      // (never dereferenced in the partial form)
    return prologue_sample_launch(target, source, n_polyak, tau, zero_out, n_zero, seed, step_counter, uniform_out, normal_out,
                                  n_normal, subsets_out, n_subsets, E_sample, E, tree, capacity, batch, slot_ids, &unused_beta,
                                  0.0, leaf_out, p_out, ids_out, nullptr, min_p_out, true, stream);
}

int asac_per_is_weights(const float* p, int batch, const float* total, const float* min_ratio,
                        double* beta_state, double beta_increment, float* is_weights_out,
                        void* stream) {
    if (batch <= 0) return bad_arg("asac_per_is_weights");
    hipStream_t s = as_stream(stream);
    const int blocks = (batch + 255) / 256;
    if (batch <= kFusedSampleMax && g_launch_repeat == 1) {
        ASAC_LAUNCH(k_is_weights_advance, dim3(1), dim3(256), 0, s, p, batch, total, min_ratio, beta_state, beta_increment,
                    is_weights_out);
        return finish_launch("asac_per_is_weights");
    }
    ASAC_LAUNCH(k_is_weights, dim3(blocks), dim3(256), 0, s, p, batch, total, min_ratio,
                       beta_state, beta_increment, is_weights_out, 1);
    ASAC_LAUNCH(k_advance_beta, dim3(1), dim3(1), 0, s, beta_state, beta_increment);
    return finish_launch("asac_per_is_weights");
}

int asac_sumtree_update(float* tree, int capacity, int k, const int64_t* ids,
                        const int64_t* slot_ids, const float* td_error, float alpha, float td_min,
                        float td_max, int mode, int32_t* winner, int32_t* nan_flag, void* stream) {
    return asac_sumtree_update_sc(tree, capacity, k, ids, slot_ids, td_error, alpha, td_min, td_max, mode, winner, nan_flag,
                                  nullptr, 0, stream);
}

int asac_sumtree_update_sc(float* tree, int capacity, int k, const int64_t* ids,
                           const int64_t* slot_ids, const float* td_error, float alpha, float td_min,
                           float td_max, int mode, int32_t* winner, int32_t* nan_flag,
                           const asac_sidecar_t* sidecars_host, int n_sidecars, void* stream) {
    if (capacity <= 0 || (capacity & (capacity - 1)) || k <= 0 || !winner || !nan_flag)
        return bad_arg("asac_sumtree_update");
    SidecarsDev sc{}, none{};
    if (sidecars_prepare(sidecars_host, n_sidecars, sc)) return bad_arg("asac_sumtree_update: sidecar");
    // items beyond the first kUpdateBlock keep their (leaf, p) in the tail of the winner scratch?
    // No: winner is [C] and must stay -1.  They are spilled behind it by contract: callers with
    // k > 1024 must provide winner of size C + 2k.
    const int threads = k <= 256 ? 256 : kUpdateBlock;
    int32_t* item_scratch = winner + capacity;
    if (sc.n == 0) {
        ASAC_LAUNCH(k_sumtree_update, dim3(1), dim3(threads), 0, as_stream(stream), tree, capacity, ilog2(capacity), k,
                    ids, slot_ids, td_error, alpha, td_min, td_max, mode, winner, nan_flag, item_scratch);
        return finish_launch("asac_sumtree_update");
    }
    for (int rep = 0; rep < g_launch_repeat; ++rep) {      // (repeat knob: only the last repetition carries the sidecars)
        const bool last = rep == g_launch_repeat - 1;
        const dim3 grid(1u + (unsigned)(last ? sc.blocks : 0));
        if (sc.n == 1)
            hipLaunchKernelGGL(k_sumtree_update_sc<1>, grid, dim3(threads), 0, as_stream(stream), tree, capacity,
                               ilog2(capacity), k, ids, slot_ids, td_error, alpha, td_min, td_max, mode, winner, nan_flag,
                               item_scratch, sidecars_first<1>(last ? sc : none));
        else
            hipLaunchKernelGGL(k_sumtree_update_sc<ASAC_MAX_SIDECARS>, grid, dim3(threads), 0, as_stream(stream), tree,
                               capacity, ilog2(capacity), k, ids, slot_ids, td_error, alpha, td_min, td_max, mode, winner,
                               nan_flag, item_scratch, last ? sc : none);
    }
    return finish_launch("asac_sumtree_update");
}

int asac_per_add(float* tree, int capacity, int64_t first_id, int count, int ignore_size,
                 const float* max_p_dev, float max_p_host, int64_t* slot_ids, void* stream) {
    if (capacity <= 0 || (capacity & (capacity - 1)) || count <= 0) return bad_arg("asac_per_add");
    const int threads = count <= 256 ? 256 : kUpdateBlock;
    ASAC_LAUNCH(k_per_add, dim3(1), dim3(threads), 0, as_stream(stream), tree, capacity,
                       ilog2(capacity), first_id, count, ignore_size, max_p_dev, max_p_host, slot_ids);
    return finish_launch("asac_per_add");
}

int asac_sumtree_leaf_max(const float* tree, int capacity, float* out, void* stream) {
    if (capacity <= 0) return bad_arg("asac_sumtree_leaf_max");
    hipStream_t s = as_stream(stream);
    ASAC_LAUNCH(k_fill_u32, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned int*>(out), 0u);
    int blocks = (capacity / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    ASAC_LAUNCH(k_leaf_max, dim3(blocks), dim3(256), 0, s, tree + (capacity - 1), capacity,
                       reinterpret_cast<unsigned int*>(out));
    return finish_launch("asac_sumtree_leaf_max");
}

int asac_sumtree_check(const float* tree, int capacity, int32_t* out, void* stream) {
    hipStream_t s = as_stream(stream);
    ASAC_LAUNCH(k_fill_u32, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned int*>(out), 0u);
    if (capacity > 1)
        ASAC_LAUNCH(k_tree_check, dim3((capacity - 1 + 255) / 256), dim3(256), 0, s, tree,
                           capacity, out);
    return finish_launch("asac_sumtree_check");
}

}  // extern "C"
