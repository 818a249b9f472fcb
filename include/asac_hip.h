/*
 * asac_hip.h — C ABI of libasac_hip.so: the MI355X (gfx950) kernels behind the SAC training step
 * of BlueFisher/Advanced-Soft-Actor-Critic (`SAC_Base.train()`).
 *
 * The reference has no FFI: its "operator API" for this path is two Python classes
 * (SURVEY.md §8b).  Each entry point below names the reference code it replaces (file:line under
 * the reference root) — that is the binding a maintainer would route through ctypes, see
 * INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`; no ownership transfer,
 *     no allocation, no host synchronisation; re-entrant per stream (safe to capture in a hipGraph)
 *   - `stream` is a hipStream_t passed as void*
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch / argument check
 *     (asac_last_error() gives text)
 *   - f32 sums, products and scans follow the reference's evaluation order; the library is built with
 *     -ffp-contract=off so no multiply-add is fused behind the reference's back.  Transcendentals are the device's
 *     (`expf`, `logf`, `tanhf`, `powf`: within a few ulp of the host's).  ONE function is an approximation by design:
 *     GELU inside the fused MLP / convolution kernels is x * Phi(x) with erf from Abramowitz-Stegun 7.1.26 (one
 *     `__expf`, one `v_rcp_f32`; csrc/asac_gelu.h) instead of ATen's erff.  Bound, against float64 GELU on [-12, 12]:
 *     |gelu - gelu64| <= 6e-7 + 3e-7 |gelu64|, |gelu' - gelu64'| <= 6e-7 (observed on MI355X: 4.6e-7 and 2.8e-7;
 *     ATen's own f32 GELU is 4.5e-7 from the same float64 values) — asserted by
 *     tests/test_kernels_gpu.py::test_gelu_against_torch through asac_gelu_eval() below
 */
#ifndef ASAC_HIP_H
#define ASAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASAC_ABI_VERSION 81
#define ASAC_MAX_GATHER_KEYS 16
#define ASAC_MAX_ENSEMBLE 16
#define ASAC_MAX_ACTION 64
#define ASAC_MLP_MAX_BLOCKS 4

int asac_version(void);
/* sizeof of a by-value struct of this header ("asac_mlp_job_t", ...) as the library was compiled, -1 for an unknown name:
 * bindings in other languages check their mirrors of the structs against it */
int64_t asac_struct_size(const char* name);
const char* asac_last_error(void);

/* Measurement knob: every kernel launch inside an entry point is issued `repeat` times
 * back-to-back (default 1).  bench.py sets it during its profile pass so that HIP events around one
 * call resolve per-launch device time.  Returns the previous value.  Not for production use. */
int asac_set_launch_repeat(int repeat);

/* ---------------------------------------------------------------------------------------------
 * Sum tree (HBM-resident segment tree).  `tree` is the reference's array heap: f32[2C-1], root
 * at 0, leaves at [C-1, 2C-1), C a power of two.  replay_buffer.py:145-167.
 * ------------------------------------------------------------------------------------------- */

/* K1(+K2) stratified inverse-CDF sample, fused with leaf->slot->id lookup and (optionally) the
 * importance-sampling weights.
 * Replaces SumTree.sample (replay_buffer.py:185-205) + _prefetch_loop lines 347-354.
 *   u            f64[batch] uniforms in [0,1): v_i = lo_i + (hi_i-lo_i)*u_i exactly as
 *                np.random.uniform does; parity mode feeds recorded draws, fast mode device RNG
 *   slot_ids     i64[C]   DataStorage._id (id currently stored in each ring slot)
 *   beta_state   f64[1]   importance exponent; updated in place to min(1, beta+beta_increment)
 *                BEFORE use (line 353).  May be NULL together with is_weights_out.
 *   leaf_out     i32[batch] tree index of the sampled leaf        (bit-exact vs reference)
 *   p_out        f32[batch] leaf priority
 *   ids_out      i64[batch] data ids (what PrioritizedReplayBuffer.sample returns first)
 *   is_weights_out f32[batch] ((p/total)/min(p/total))^-beta, or NULL to skip (multi-GPU: use
 *                asac_per_is_weights after the cross-rank reduction)
 *   min_p_out    f32[2]  [0] = min over the batch of p (always written); [1] scratch (min ratio)
 *                of the multi-workgroup path (batch > 256)
 */
int asac_sumtree_sample(const float* tree, int capacity, int batch, const double* u,
                        const int64_t* slot_ids, double* beta_state, double beta_increment,
                        int32_t* leaf_out, float* p_out, int64_t* ids_out, float* is_weights_out,
                        float* min_p_out, void* stream);

/* The binary descent of asac_sumtree_sample alone, for n explicit f64 values (each in [0, root]): leaf index, leaf
 * priority and stored id per value; same comparisons (f64 value against f32 node sums, replay_buffer.py:196-205).
 * Used by the sharded "parity" sampling (algorithm/parallel.py): the G shard trees are the subtrees of one tree whose
 * top levels every rank walks on the host; the owner finishes the walk here with the residual values. */
int asac_sumtree_descend(const float* tree, int capacity, int n, const double* values, const int64_t* slot_ids,
                         int32_t* leaf_out, float* p_out, int64_t* ids_out, void* stream);

/* Sharded replay, "parity" sampling (no reference counterpart beyond the single tree it
 * reproduces, replay_buffer.py:172-205): the G = 2^k shard trees are the subtrees of ONE sum tree.
 *   asac_sumtree_plan_top       the k top levels: parents = left + right (f32) over the G shard roots; every sample of the
 *                               GLOBAL batch draws v = lo + (hi - lo) u over the global root and walks them with the
 *                               reference's comparisons -> owner_out i32[batch] (the shard that holds the sample),
 *                               value_out f64[batch] (the residual value its tree continues with), total_out f32[1]
 *   asac_sumtree_descend_owned  asac_sumtree_descend for the samples with owner[i] == rank; the others get leaf -1,
 *                               priority 0, id -1 (so that a SUM all-reduce of p_out over the ranks is the batch's
 *                               priorities, and update / scatter launches on ids_out skip them as stale)
 *   asac_per_is_weights_slice   IS weights of rows [first, first + count) of the global batch: minimum over all n_all
 *                               priorities, beta advanced first (352-354)
 * Everything stays on the device: with the roots all-gathered and the windows exchanged by fixed-size collectives the
 * sharded step has no host synchronisation and is captured like the plain one. */
int asac_sumtree_plan_top(const float* shard_roots, int n_shards, int batch, const double* u, int32_t* owner_out,
                          double* value_out, float* total_out, void* stream);
int asac_sumtree_descend_owned(const float* tree, int capacity, int n, const double* values, const int32_t* owner,
                               int rank, const int64_t* slot_ids, int32_t* leaf_out, float* p_out, int64_t* ids_out,
                               void* stream);
int asac_per_is_weights_slice(const float* p_all, int n_all, int first, int count, const float* total, double* beta_state,
                              double beta_increment, float* is_weights_out, void* stream);

/* K2 stand-alone: w_i = ((p_i/total)/(min_ratio))^-beta, beta_state advanced first.
 * total / min_ratio are device scalars so a cross-rank all-reduce can produce them without a
 * host round trip.  replay_buffer.py:352-354. */
int asac_per_is_weights(const float* p, int batch, const float* total, const float* min_ratio,
                        double* beta_state, double beta_increment, float* is_weights_out,
                        void* stream);

/* K6 priority update: p = clip(td, td_min, td_max)^alpha, rows whose ring slot was overwritten
 * since the sample are dropped (slot_ids[id % C] != id), duplicate ids: last one wins, then every
 * ancestor is recomputed as left+right, level by level.
 * Replaces PrioritizedReplayBuffer.update (replay_buffer.py:412-427) + SumTree.update (172-183).
 *   mode 0: td_error -> priority as above;  mode 1: `td_error` already holds priorities
 *   winner   i32[C] scratch, plus 2k more entries when k > 1024; the first C entries are all -1 on entry and are
 *            restored to -1 on exit (the tail spills per-item state when k > 1024)
 *   nan_flag i32[1] set to 1 (and nothing is written) when a td_error is NaN (lines 418-420)
 */
int asac_sumtree_update(float* tree, int capacity, int k, const int64_t* ids,
                        const int64_t* slot_ids, const float* td_error, float alpha, float td_min,
                        float td_max, int mode, int32_t* winner, int32_t* nan_flag, void* stream);

/* Episode ingress into the tree: rows [first_id, first_id+count) (ids mod 10*C) get priority
 * max_p, except the episode's last `ignore_size` rows and ring slots >= C-ignore_size which get
 * 0; slot_ids is updated; ancestors recomputed.  Replaces PrioritizedReplayBuffer.add
 * (replay_buffer.py:293-307) minus the row copy (done by the caller with async copies).
 *   max_p_dev  f32[1] device scalar (result of asac_sumtree_leaf_max) or NULL -> max_p_host */
int asac_per_add(float* tree, int capacity, int64_t first_id, int count, int ignore_size,
                 const float* max_p_dev, float max_p_host, int64_t* slot_ids, void* stream);

/* K8 max over the C leaves -> out[0].  Replaces SumTree.max (replay_buffer.py:237-239). */
int asac_sumtree_leaf_max(const float* tree, int capacity, float* out, void* stream);

/* Debug invariant: counts internal nodes with tree[i] != tree[2i+1]+tree[2i+2] into out[0]. */
int asac_sumtree_check(const float* tree, int capacity, int32_t* out, void* stream);

/* Measurement: value[i] = gelu(z[i]), deriv[i] = gelu'(z[i]) exactly as the fused MLP / convolution kernels evaluate
 * them (replaces nothing: the reference's activation is ATen's `nn.GELU()`, linear_layers.py:24-119; this exposes
 * the library's evaluation so that its distance from ATen's can be bounded by a test). */
int asac_gelu_eval(const float* z, float* value, float* deriv, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Ring storage: window gather with episode-continuity padding, and predicated row scatter.
 * ------------------------------------------------------------------------------------------- */
enum {
    ASAC_PAD_KEEP = 0,      /* copy rows as stored (observations, last_mask)                     */
    ASAC_PAD_WORD = 1,      /* invalid rows: every 4-byte word := pad_word (index -1, reward 0,
                               mu_prob 1.0f, hidden 0)                                             */
    ASAC_PAD_BYTE = 2,      /* invalid rows: every byte := pad_word & 0xff (done := 1)            */
    ASAC_PAD_ROW = 3,       /* invalid rows := pad_row[0:row_bytes] (action := padding action)    */
    ASAC_PAD_EMIT_MASK = 4  /* no source: dst byte := row is invalid (the padding_mask itself)    */
};
enum {
    ASAC_CVT_NONE = 0,
    ASAC_CVT_U8_TO_F32_UNIT = 1, /* uint8 -> f32 / 255  (sac_base.py:783-786) */
    ASAC_CVT_BOOL_TO_F32 = 2     /* bool  -> f32        (sac_base.py:787-788) */
};
/* Derived keys: a sequence representation is fed the window's step indexes, padding mask and PREVIOUS actions
 * extended / shifted by one row (SAC_Base.get_bnx_data, sac_base.py:1090-1115; utils/operators.py gen_n_pre_actions);
 * a key with `derive` set delivers that form straight from the ring — the padded window row it shows is
 *   PREVIOUS         row j-1, zeros for j = 0                                  (pre_action from the action column)
 *   HOLD_LAST        row min(j, L-2)                                           (the padding mask, ASAC_PAD_EMIT_MASK)
 *   HOLD_LAST_NEXT   row min(j, L-2); at j = L-1 the i32 value + (value != -1)  (the step index)
 * — the values asac_window_aux forms from the gathered window, without its launch.  L >= 2, no conversion. */
enum { ASAC_DERIVE_NONE = 0, ASAC_DERIVE_PREVIOUS = 1, ASAC_DERIVE_HOLD_LAST = 2, ASAC_DERIVE_HOLD_LAST_NEXT = 3 };
typedef struct {
    const void* src;     /* ring [C, row_bytes]                                                   */
    void* dst;           /* [batch, L, out_row_bytes]; out_row_bytes = row_bytes (x4 if converting) */
    const void* pad_row; /* ASAC_PAD_ROW only                                                     */
    int32_t row_bytes;
    int32_t pad_mode;
    uint32_t pad_word;
    int32_t convert;
    int32_t dst_row_pitch; /* bytes between destination rows; 0 = dense (out_row_bytes).  A wider pitch lets a key
                              land as a column block of a wider [batch, L, *] tensor: the vector observation beside
                              the previous action, the concatenation a recurrent representation starts with */
    int32_t derive;        /* ASAC_DERIVE_*: which window row a destination row shows (0: its own)                */
} asac_gather_key_t;

/* K3: for every sampled id gather the rows id-prev_n .. id+post_n of every key (ring slot =
 * id mod C, negative ids wrap like NumPy's %), fused with the validity test
 *   valid(j) := j == prev_n  or  index[j] - index[prev_n] == j - prev_n
 * and the padding values.  Replaces DataStorage.get over the window ids
 * (replay_buffer.py:356-364, 64-75) + SAC_Base._sample_from_replay_buffer padding
 * (sac_base.py:2435-2453) + _process_torch_obs_list (783-788).
 *   keys_host  HOST array of n_keys descriptors (copied into the kernel argument buffer)
 *   index_ring i32[C]  the stored 'index' column */
int asac_window_gather_pad(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids,
                           int batch, int prev_n, int post_n, int capacity,
                           const int32_t* index_ring, void* stream);

/* Random reads: dst_key[r] = ring_key[ids[r] mod capacity] for every key (pad_mode ASAC_PAD_KEEP, convert
 * ASAC_CVT_NONE), any ids, no residency check, ONE launch for all keys — the reference's
 * `DataStorage.get(ids)` behind `PrioritizedReplayBuffer.get_storage_data` (replay_buffer.py:64-75, 401-406), which
 * the option-critic variant calls per key-transition hop (oc/option_selector_base.py:2205, 2223). */
/* The same gather as a PLAN in device memory (asac_window_gather_plan_bytes() bytes at plan_dev, written with a blocking
 * copy: build time, not step time) for an ASAC_SIDECAR_WINDOW_GATHER job: the windows of the batch the lookahead schedule
 * (`hip_config['lookahead']`) draws one step ahead are gathered by extra workgroups of a launch of the current step
 * (replay_buffer.py:377-396 `sample`'s `get_storage_data` part).  blocks_out: the workgroups the job needs. */
int64_t asac_window_gather_plan_bytes(void);
int asac_window_gather_plan(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int batch, int prev_n,
                            int post_n, int capacity, const int32_t* index_ring, void* plan_dev, int* blocks_out);

int asac_gather_rows(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int n_rows, int capacity,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Episode slabs: the agent-side episode assembly (reference algorithm/agent.py) in HBM.
 * One row mover serves every movement of an environment step, over all agents and all keys in ONE launch:
 *   commit   pending rows + this step's scalars -> slab[slot, cursor]   (Agent._add_transition, agent.py:191-235)
 *   collect  pending action / hidden state of the listed agents -> dense batch
 *                                                 (AgentManager._get_merged_action/_seq_hidden_state, 481-485)
 *   stage    policy outputs + new observations -> pending rows          (Agent.set_tmp_obs_action, 87-97)
 *   window   the last rows of the agents' running episodes, left-padded (Agent.get_episode_trans(force_length)
 *            258-316 + the concatenations of AgentManager.get_action 536-552)
 * Item i of a key copies row_bytes from  src + a*src_stride0 + b*src_stride1  to  dst + c*dst_stride0 +
 * d*dst_stride1  with (a, b) resp. (c, d) given by the side's addressing mode:
 *   ASAC_ROW_ITEM (i, 0)   ASAC_ROW_SLOT (slot[i], 0)   ASAC_ROW_SLOT_ROW (slot[i], row[i] (+ src_row_offset))
 *   ASAC_ROW_BROADCAST (0, 0; sources only)
 * A SLOT_ROW source whose row index is negative yields padding: every 4-byte word := pad_word (byte-wide keys:
 * its low byte); a SLOT_ROW destination with a negative row is skipped.  Strides in bytes. */
enum { ASAC_ROW_ITEM = 0, ASAC_ROW_SLOT = 1, ASAC_ROW_SLOT_ROW = 2, ASAC_ROW_BROADCAST = 3 };
typedef struct {
    const void* src;
    void* dst;
    int64_t src_stride0, src_stride1;
    int64_t dst_stride0, dst_stride1;
    int32_t row_bytes;
    int32_t src_mode, dst_mode;
    int32_t src_row_offset;
    uint32_t pad_word;
    int32_t reserved_;
} asac_row_move_t;

/*   keys_host  HOST array of n_keys (<= ASAC_MAX_GATHER_KEYS) descriptors (copied into the kernel arguments)
 *   slot, src_row, dst_row   i32[n_items] device arrays (each may be NULL when no key's mode reads it) */
int asac_rows_move(const asac_row_move_t* keys_host, int n_keys, const int32_t* slot, const int32_t* src_row,
                   const int32_t* dst_row, int n_items, void* stream);

/* The representation's window inputs derived from the sampled window, in one launch (SAC_Base.get_bnx_data,
 * sac_base.py:1090-1115; utils/operators.py gen_n_pre_actions with keep_last_action): for the L-1 leading
 * rows `bn` of every sampled window
 *   index_x      [B][L]    = bn indexes, then last + (last != -1)
 *   pad_x        [B][L]    = bn padding mask, then its last entry again
 *   pre_action   [B][L][A] = zeros, then the bn actions (rows pre_action_stride_t floats apart; 0 = A, dense)
 * index / padding_mask / action are the window tensors ([B][>=L-1] with the given strides in elements). */
int asac_window_aux(const int32_t* index, int64_t index_stride_b, const uint8_t* padding_mask,
                    int64_t mask_stride_b, const float* action, int64_t action_stride_b, int64_t action_stride_t,
                    int B, int L, int A, int32_t* index_x_out, uint8_t* padding_mask_x_out,
                    float* pre_action_out, int64_t pre_action_stride_t, void* stream);

/* K7: rows[s, j] -> ring[(ids[s] + first_off + j) mod C] for j in [0, count), only where
 * padding_mask[s, j] == 0 and the slot still holds that id; when several rows target one slot the
 * last in row-major (s, j) order wins (NumPy fancy-assignment order).
 * Replaces PrioritizedReplayBuffer.update_transitions (replay_buffer.py:429-434) + the target-id
 * construction in SAC_Base.train (sac_base.py:2589-2605).
 *   rows            [batch, count, row_bytes] with row stride rows_row_stride_bytes between j's
 *                   and rows_sample_stride_bytes between samples
 *   padding_mask    u8 [batch, count], strides in bytes (mask_sample_stride)
 *   winner          i32[C] scratch, all -1 on entry and on exit */
int asac_scatter_rows_if_id_match(void* ring, int row_bytes, int capacity, const int64_t* ids,
                                  int batch, int first_off, int count, const int64_t* slot_ids,
                                  const uint8_t* padding_mask, int mask_sample_stride,
                                  const void* rows, int64_t rows_sample_stride_bytes,
                                  int64_t rows_row_stride_bytes, int32_t* winner, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Return / target kernels (the non-GEMM arithmetic of _get_y, _v_trace, the Q loss).
 * ------------------------------------------------------------------------------------------- */

/* x = loc + eps*scale (Normal.rsample), a = tanh(x), and the tanh-squash-corrected log-prob
 *   logp = sum_d [ N(x_d; loc_d, scale_d).log_prob - sum_e log(max(1 - tanh(x_e)^2, 1e-2)) ]
 * (the correction is summed over the action dim and broadcast back before the final sum, as
 * operators.py:12-14,22-24 do).  Replaces sac_base.py:1346,1351(tanh),1430 and 1883-1890.
 *   loc, scale   row r at loc + r*ls_row_stride (A for dense [rows, A]; 2A when they are the two
 *                halves of the fused policy network's [rows, 2A] output)
 *   eps, a_tanh_out  dense [rows, A]; logp_out [rows]; x_out may be NULL
 *   optional, same launch: the per-dimension probability of STORED actions under the same Gaussian
 *   (see asac_squash_prob for the addressing); action == NULL skips it.  sac_base.py:1452. */
int asac_squash_sample_fwd(const float* loc, const float* scale, int64_t ls_row_stride, const float* eps,
                           int64_t rows, int A, float* a_tanh_out, float* logp_out, float* x_out,
                           const float* action, int T, int64_t action_stride_b, int64_t action_stride_t,
                           int action_offset, float* prob_out, int64_t prob_stride_b, int64_t prob_stride_t,
                           int prob_offset, void* stream);

/* Backward of the sampling part for the policy update: given dL/da_tanh [rows, A] (may be NULL) and
 * dL/dlogp [rows] (may be NULL; then, with log_alpha != NULL, the policy objective's constant
 * exp(*log_alpha) / rows is used) produce dL/dloc, dL/dscale (rows grad_row_stride floats apart, so
 * they can be the two halves of one [rows, 2A] gradient for the fused policy network). */
int asac_squash_sample_bwd(const float* loc, const float* scale, int64_t ls_row_stride, const float* eps,
                           const float* grad_a, int grad_a_members, int64_t grad_a_member_stride,
                           const float* grad_logp, const float* log_alpha, int64_t rows, int A,
                           float* grad_loc, float* grad_scale, int64_t grad_row_stride, void* stream);

/* Sidecar jobs: small, off-critical-path launches of the train step executed by EXTRA workgroups of a hosting
 * launch (asac_squash_multi, asac_mlp_forward_multi) instead of paying a dependent kernel boundary of their own.
 * A sidecar is ordered after everything launched before its host and before everything launched after it; it must
 * not touch what the host launch itself reads or writes.  Fields as the arguments of the stand-alone entry points:
 *   ASAC_SIDECAR_ALPHA_ADAM      asac_alpha_adam_step (reference sac_base.py:1913-1949)
 *   ASAC_SIDECAR_SCATTER_ELECT   pass 1 of asac_scatter_rows_if_id_match (replay_buffer.py:429-434)
 *   ASAC_SIDECAR_SCATTER_WRITE   pass 2; needs a launch boundary after the ELECT job of the same scatter
 *   ASAC_SIDECAR_WINDOW_GATHER   asac_window_gather_pad of a plan made by asac_window_gather_plan (the NEXT batch's
 *                                windows of the lookahead schedule); hosted by asac_policy_sample_q_forward only */
#define ASAC_SIDECAR_ALPHA_ADAM 1
#define ASAC_SIDECAR_SCATTER_ELECT 2
#define ASAC_SIDECAR_SCATTER_WRITE 3
#define ASAC_SIDECAR_WINDOW_GATHER 4
#define ASAC_MAX_SIDECARS 4
typedef struct {
    int32_t kind;
    /* ALPHA_ADAM */
    const float* logp;
    int32_t B;
    float target;
    int32_t slot;
    float *param, *grad, *exp_avg, *exp_avg_sq;
    int32_t n;
    float lr, beta1, beta2, eps;
    int64_t* steps_done;
    int32_t advance_counter;
    /* SCATTER_ELECT / SCATTER_WRITE */
    void* ring;
    int32_t row_bytes, capacity;
    const int64_t* ids;
    int32_t batch, first_off, count;
    const int64_t* slot_ids;
    const uint8_t* padding_mask;
    int32_t mask_sample_stride;
    const void* rows;
    int64_t rows_sample_stride_bytes, rows_row_stride_bytes;
    int32_t* winner;
    /* WINDOW_GATHER */
    const void* gather_plan;
    int32_t gather_blocks;
} asac_sidecar_t;

/* Up to ASAC_SQUASH_MAX_JOBS independent jobs of the two kinds above / below in ONE launch (a train
 * step samples for the target, for the policy step, for the temperature step and for the TD error, and
 * scores the stored actions, on outputs of at most two policy forwards).  Fields as the arguments of
 * asac_squash_sample_fwd; eps == NULL makes the job a probability-only job (asac_squash_prob). */
#define ASAC_SQUASH_MAX_JOBS 4
typedef struct {
    const float* loc;
    const float* scale;
    int64_t ls_row_stride;
    const float* eps;
    int64_t rows;
    int32_t A;
    int32_t T;
    float* a_tanh_out;
    float* logp_out;
    float* x_out;
    const float* action;
    int64_t action_stride_b, action_stride_t;
    float* prob_out;
    int64_t prob_stride_b, prob_stride_t;
    int32_t action_offset, prob_offset;
} asac_squash_job_t;
int asac_squash_multi(const asac_squash_job_t* jobs_host, int n_jobs, const asac_sidecar_t* sidecars_host,
                      int n_sidecars, void* stream);

/* Per-dimension tanh-squashed policy probability of STORED actions:
 *   x = atanh(clamp(a, -0.999, 0.999)); prob_d = exp(N.log_prob(x_d)) / prod_e max(1-tanh(x_e)^2, 1e-2)
 * Replaces sac_base.py:1183-1187 (get_l_probs) and 1452 (pi for V-trace); operators.py:17-19.
 *   rows = (#samples) * T;  action element (s, t, d) at
 *   s*action_stride_b + t*action_stride_t + action_offset + d (floats); prob_out likewise */
int asac_squash_prob(const float* loc, const float* scale, int64_t ls_row_stride, const float* action, int T,
                     int64_t action_stride_b, int64_t action_stride_t, int action_offset,
                     int64_t rows, int A, float* prob_out, int64_t prob_stride_b,
                     int64_t prob_stride_t, int prob_offset, void* stream);

typedef struct {
    /* target-Q table q[e][b][t], t in [0, n]: strides in floats */
    const float* q;
    int64_t q_stride_e, q_stride_b, q_stride_t;
    const int32_t* subset_n;    /* DEVICE i32[E_sample] members used for V(s_t) (randperm draw 1); NULL = 0..E_sample-1 */
    const int32_t* subset_next; /* DEVICE i32[E_sample] members used for V(s_t+1) (randperm draw 2)                     */
    int32_t E_sample;
    const float* logp;          /* [B, n+1] log pi(a'|s), contiguous                              */
    const float* log_alpha;     /* f32[1]                                                         */
    const float* reward;        /* [B, n]  row stride reward_stride (floats)                      */
    int64_t reward_stride;
    const uint8_t* done;        /* [B, n]  row stride mask_stride (bytes), same for last / pad    */
    const uint8_t* last_mask;
    const uint8_t* padding_mask;
    int64_t mask_stride;
    const float* mu_prob;       /* [B, n, A_total] behaviour probs; NULL when !use_n_step_is       */
    int64_t mu_stride_b, mu_stride_t;
    int32_t mu_offset;          /* first continuous component                                     */
    const float* pi_prob;       /* [B, >=n, A] current-policy per-dim probs (asac_squash_prob)     */
    int64_t pi_stride_b, pi_stride_t;
    int32_t A;                  /* continuous action size                                         */
    const float* gamma_ratio;   /* f32[n]  gamma^t   (torch.logspace, computed by the host once)   */
    const float* lambda_ratio;  /* f32[n]                                                         */
    float gamma, v_rho, v_c;
    int32_t use_n_step_is;
    int32_t B, n;
    /* optional fused TD error: td[b] = mean_e |q_online[e][b] - y[b]|  (sac_base.py:2233-2244)   */
    const float* q_online;      /* [E_online, B] contiguous or NULL                               */
    int32_t E_online;
    float* td_error_out;        /* f32[B] or NULL                                                 */
    float* y_out;               /* f32[B]                                                         */
} asac_vtrace_args_t;

/* K4: ensemble subset + min over E, V = minQ - alpha*logpi, pi/mu products with the reference's
 * inf/nan masking (operators.py:27-31), rho/c clipping, cumprod of c, gamma^t lambda^t, masks,
 * sum -> y[B].  One lane per batch row; the [rows x n] input slabs of a 64-row tile are staged
 * through LDS with coalesced loads.  Replaces sac_base.py:1434-1445 + 1450-1464 + _v_trace
 * (1244-1295). */
int asac_vtrace_return_min(const asac_vtrace_args_t* args_host, void* stream);
/* ... with sidecar jobs riding as extra workgroups of the launch (e.g. the write pass of the behaviour-probability
 * write-back beside the TD error's return) */
/* pending_alpha (optional): an ASAC_SIDECAR_ALPHA_ADAM job that has NOT run yet but precedes this return in the
 * reference's order (the TD error sees the updated temperature): the launch uses the value that job will write
 * (args.log_alpha must be its parameter slot); the job itself rides in a later launch (asac_sumtree_update_sc). */
int asac_vtrace_return_min_sc(const asac_vtrace_args_t* args_host, const asac_sidecar_t* sidecars_host,
                              int n_sidecars, const asac_sidecar_t* pending_alpha, void* stream);
/* The TD errors' return and the priority update that consumes them as ONE launch (asac_vtrace_return_min with
 * q_online/td_error_out set, then asac_sumtree_update mode 0 over ids[0..B) with those TD errors): the update's single
 * workgroup forms the B returns itself (bit-identical to the return kernel's) and goes on to the tree without a launch
 * boundary.  `alpha_step` (optional): an ASAC_SIDECAR_ALPHA_ADAM job that precedes the return in the reference's order
 * (sac_base.py:1913-1949 before 2182-2245) and has not run yet — the workgroup RUNS it first (args.log_alpha must be
 * its parameter slot).  Sidecar jobs ride as workgroups 1...  Replaces sac_base.py:2182-2245 + replay_buffer.py:412-427
 * for one batch; B <= 1024 and 2 B (n+2) + 2 B floats of LDS <= 128 KB.  The `winner` scratch is not touched (the
 * last-writer election among the batch's rows runs on chip). */
int asac_td_update(const asac_vtrace_args_t* args_host, float* tree, int capacity, const int64_t* ids,
                   const int64_t* slot_ids, float alpha, float td_min, float td_max, int32_t* winner, int32_t* nan_flag,
                   const asac_sidecar_t* sidecars_host, int n_sidecars, const asac_sidecar_t* alpha_step, void* stream);
/* asac_sumtree_update (declared above) with sidecar jobs riding as extra workgroups */
int asac_sumtree_update_sc(float* tree, int capacity, int k, const int64_t* ids, const int64_t* slot_ids,
                           const float* td_error, float alpha, float td_min, float td_max, int mode, int32_t* winner,
                           int32_t* nan_flag, const asac_sidecar_t* sidecars_host, int n_sidecars, void* stream);

/* Same scan with V(s_t), V(s_t+1) [B, n] and the pi / mu products [B, n] handed in directly (the
 * discrete-action branch computes its V from categorical probabilities, sac_base.py:1387-1421;
 * only reward / masks / ratios / gamma fields of `args_host` are read).  _v_trace, 1244-1295. */
int asac_vtrace_return_direct(const asac_vtrace_args_t* args_host, const float* v_n,
                              const float* v_next, const float* pi_prod, const float* mu_prod,
                              void* stream);

/* Clipped double-Q loss, forward value and gradient in one pass (sac_base.py:1539-1561):
 *   l_e = mean_b w_b * max( (tq+clamp(q-tq,-eps,eps) - y)^2, (q - y)^2 )    (eps <= 0: 2x plain MSE
 *   is NOT reproduced here; callers fall back to eager ops for clip_epsilon <= 0)
 *   q, tq: [E, B] contiguous; y, w: [B] (w may be NULL)
 *   loss_out f32[E] per-ensemble means; grad_q_out [E, B] = d(sum_e l_e)/dq */
int asac_q_loss_fwd_bwd(const float* q, const float* tq, const float* y, const float* w, int E,
                        int B, float clip_eps, float* loss_out, float* grad_q_out, void* stream);

/* Policy objective of the continuous head, value + gradients in one launch (sac_base.py:1882-1903,
 * 1910-1911):  L = mean_b(alpha*logp_b - min_{e in subset} q[e][b]);  grad_logp[b] = alpha/B;
 * grad_q[e][b] = -1/B at the first arg-min member of the subset, 0 elsewhere;  entropy_out (may be
 * NULL, needs scale [B, A], rows scale_row_stride floats apart) = mean_b sum_d (log scale + 1/2 + 1/2 log 2pi).
 *   q [E, B] contiguous; subset DEVICE i32[E_sample] or NULL (= members 0..E_sample-1) */
int asac_policy_loss_fwd_bwd(const float* logp, const float* q, const int32_t* subset, int E, int E_sample,
                             int B, const float* log_alpha, const float* scale, int64_t scale_row_stride, int A,
                             float* loss_out, float* grad_logp, float* grad_q, float* entropy_out, void* stream);

/* Temperature gradient of the continuous head (sac_base.py:1931-1944):
 * *grad_slot = mean_b(-logp_b) - target, where target = target_c_alpha * (-A).  grad_slot is the
 * flat-gradient element of log_c_alpha. */
int asac_alpha_grad(const float* logp, int B, float target, float* grad_slot, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused residual MLP (MFMA f32): the stock Q / policy networks as ONE launch per pass.
 * Network: x = [x0 | x1] -> n_blocks x { z = W_l x + b_l; x = GELU(z) (+ x if residual[l]) } -> up to
 * two Linear heads whose outputs are concatenated (Q: one head of 1 column; policy: mean | logstd).
 * This is reference `LinearLayers` (nn_models/layers/linear_layers.py:24-119) as composed by the
 * stock `ModelQ.c_dense` (q.py:67-91) and `ModelPolicy.c_dense / mean_dense / logstd_dense`
 * (policy.py:147-174), and as the ResBlock head of the visual encoders (`ConvLayers.dense`,
 * image_layers.py:178-216).  Block widths <= 64, total head columns <= 16; input size in0 + in1 <= 64, or
 * <= 128 for networks of at most ASAC_MLP_MAX_BLOCKS - 1 blocks whose first block is not residual (the first
 * layer then runs as two 64-column halves).  An ensemble of E
 * structurally identical networks whose parameter segments lie `member_stride` floats apart in one
 * flat buffer is evaluated by the same launch (grid.y = E).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t in0, in1;                        /* widths of the two concatenated inputs (in1 may be 0) */
    int32_t n_blocks;                        /* 1..ASAC_MLP_MAX_BLOCKS                               */
    int32_t width[ASAC_MLP_MAX_BLOCKS];      /* output width of block l                              */
    int32_t residual[ASAC_MLP_MAX_BLOCKS];   /* block adds its input (requires equal widths)         */
    int32_t head_cols[2];                    /* output columns of head 0 / head 1 (head 1 may be 0)  */
    int64_t w_off[ASAC_MLP_MAX_BLOCKS];      /* float offsets inside ONE member's parameter segment: */
    int64_t b_off[ASAC_MLP_MAX_BLOCKS];      /*   block weight [width][in] row-major, bias [width]    */
    int64_t head_w_off[2], head_b_off[2];    /*   head weight [cols][width_last], bias [cols]         */
    int32_t head_transform;                  /* 0: raw outputs; 1: Gaussian policy head — head 0 ->   */
                                             /*    5*tanh(x/5), head 1 -> exp(clamp(x, -20, 0.5))     */
                                             /*    (policy.py:170-172); backward applies the chain    */
    int32_t reserved_;
} asac_mlp_desc_t;

/* out[e][row][0:head_cols0+head_cols1] for e < E, row < N.
 *   x0 element (e, row, c) at x0 + e*x0_member_stride + row*x0_row_stride + c (member stride 0 =
 *   the same input for every ensemble member); same for x1. */
int asac_mlp_forward(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride,
                     int E, const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                     const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                     float* out, void* stream);

/* floats of scratch `asac_mlp_backward` needs when parameter gradients are requested */
/* Up to ASAC_MLP_MAX_JOBS independent forward passes (different networks and / or inputs) in ONE launch:
 * e.g. the target-Q of the stored (s, a) pairs beside the policy's forward over the window, or the online
 * and the target ensemble of the TD error.  Fields as the arguments of asac_mlp_forward. */
#define ASAC_MLP_MAX_JOBS 2
typedef struct {
    const asac_mlp_desc_t* desc;
    const float* params;
    int64_t member_stride;
    const float* x0;
    int64_t x0_row_stride, x0_member_stride;
    const float* x1;
    int64_t x1_row_stride, x1_member_stride;
    int64_t N;
    float* out;
    int32_t E;
    /* window addressing of x0: with x0_window_T > 0, row = s * T + t is read at
     * x0 + s * x0_sample_stride + t * x0_row_stride — a [samples, T, in0] view of a larger replay window
     * (states[:, b:]) without a staging copy; 0: flat rows */
    int32_t x0_window_T;
    int64_t x0_sample_stride;
} asac_mlp_job_t;
int asac_mlp_forward_multi(const asac_mlp_job_t* jobs_host, int n_jobs, const asac_sidecar_t* sidecars_host,
                           int n_sidecars, void* stream);

/* asac_mlp_forward_multi whose Gaussian-head policy jobs also run asac_squash_multi's jobs on the rows they have just
 * formed — the chain  policy forward -> [sample a ~ pi(s), log pi(a); pi(stored action); second sample at window position
 * t2]  of sac_base.py:1346-1351, 1430, 1183-1187, 1452, 2182-2245 as ONE launch: the lanes that hold a row's
 * (loc | scale) head values in MFMA layout do the elementwise work (csrc/asac_squash.h `sample_epilogue`), bit for bit
 * what asac_squash_multi writes.  epilogues[k] belongs to jobs[k]; a job without one has sample.eps == sample.action ==
 * eps2 == NULL.  In `sample`: eps / a_tanh_out / logp_out (main sample over every row), action / prob_out with their strides
 * and T (stored actions), rows (= the job's N) and A; loc / scale / ls_row_stride / x_out are unused.  A job with an epilogue:
 * E == 1, head_transform 1, 2 A <= 16.  _ok: 1 when the launch qualifies (at least one epilogue, all of them valid). */
typedef struct {
    asac_squash_job_t sample;
    const float* eps2;          /* [samples][A] or NULL: second sample from row (sample, t2) of every window */
    int32_t t2;
    int32_t reserved_;
    float* a2_out;              /* [samples][A] */
    float* logp2_out;           /* [samples] */
} asac_mlp_sample_epilogue_t;
int asac_mlp_forward_multi_sampled_ok(const asac_mlp_job_t* jobs_host, int n_jobs, const asac_mlp_sample_epilogue_t* epilogues);
int asac_mlp_forward_multi_sampled(const asac_mlp_job_t* jobs_host, int n_jobs, const asac_mlp_sample_epilogue_t* epilogues,
                                   const asac_sidecar_t* sidecars_host, int n_sidecars, void* stream);

/* Policy forward -> sampling -> critic ensemble forward over the same rows in ONE launch: the chain
 * asac_mlp_forward(_multi)[policy] -> asac_squash_multi -> asac_mlp_forward(_multi)[critics] of the return target
 * (sac_base.py:1297-1466: a' ~ pi(s_t), Q'_e(s_t, a')) and of the TD error / new behaviour probabilities (2182-2245,
 * 1159-1189), bit for bit.  A workgroup owns (16-row tile, critic e) end to end; the policy's output and the sampled
 * actions stay on chip between the three stages.
 *   pi      the policy (E = 1) over N rows (flat or window addressing); out: [N][2A] (loc | scale) or NULL
 *   sample  the main sample over every row: eps [N][A] -> a_tanh_out [N][A], logp_out [N]; with `action` also the
 *           stored actions' probabilities (fields as asac_squash_job_t; loc / scale / ls_row_stride / x_out unused)
 *   eps2    optional second sample from the same (loc | scale) at window position t2 of every sample (rows are
 *           [samples][sample.T]): eps2 [samples][A] -> a2_out [samples][A], logp2_out [samples]
 *   q       the critics (E members) on (the policy's rows, the main sample): x0 fields equal pi's, x1 ignored,
 *           out [E][N]
 * extra_jobs: up to ASAC_MLP_MAX_JOBS plain forward passes (three 64-wide blocks) riding as further workgroups;
 * sidecars as asac_mlp_forward_multi.  asac_policy_sample_q_forward_ok: 1 when `job` qualifies (both networks three
 * 64-wide blocks on <= 64 inputs, 16-byte aligned weights, scalar-head critics on (in0 | A), Gaussian-head policy). */
typedef struct {
    asac_mlp_job_t pi;
    asac_squash_job_t sample;
    const float* eps2;
    int32_t t2;
    int32_t reserved_;
    float* a2_out;
    float* logp2_out;
    asac_mlp_job_t q;
} asac_pi_q_job_t;
int asac_policy_sample_q_forward_ok(const asac_pi_q_job_t* job);
int asac_policy_sample_q_forward(const asac_pi_q_job_t* job, const asac_mlp_job_t* extra_jobs, int n_extra,
                                 const asac_sidecar_t* sidecars_host, int n_sidecars, void* stream);

#define ASAC_MLP_REDUCE_OVERWRITE 0
#define ASAC_MLP_REDUCE_ACCUMULATE 1
#define ASAC_MLP_REDUCE_DEFER 2
int64_t asac_mlp_backward_workspace(int64_t member_stride, int E, int64_t N);
/* Row tiles the backward of an [E][N] pass is cut into (16-row tiles while E * ceil(N / 16) workgroups fit one
 * resident round on the 256 CUs, else 32-row tiles): the number of per-tile partial slabs in `workspace`, i.e. the
 * `tiles` argument asac_adam_step_partials needs after an ASAC_MLP_REDUCE_DEFER backward. */
int64_t asac_mlp_backward_tiles(int64_t N, int E);

/* Backward of the above (the forward is recomputed on chip; nothing is saved between the two).
 *   grad_out     [E][N][head columns]
 *   grad_x0/x1   [E][N][in0] / [E][N][in1], written (not accumulated); either may be NULL
 *   grad_params  flat gradient buffer with the SAME layout as `params`: the tiles' partial sums are
 *                combined in a fixed order (deterministic) and, per reduce_mode, written over it
 *                (ASAC_MLP_REDUCE_OVERWRITE), added to it (ASAC_MLP_REDUCE_ACCUMULATE) or left in the
 *                workspace for asac_adam_step_partials (ASAC_MLP_REDUCE_DEFER); NULL = input
 *                gradients only
 *   workspace    asac_mlp_backward_workspace() floats (only with grad_params) */
int asac_mlp_backward(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride,
                      int E, const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                      const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                      const float* grad_out, float* grad_x0, float* grad_x1, float* grad_params,
                      float* workspace, int reduce_mode, void* stream);

/* The Q step's loss and backward in ONE launch (sac_base.py:1539-1570 for the stock ModelQ ensemble):
 * the forward is recomputed on chip anyway, so q = Q_e(x0, x1) is formed there, the clipped double-Q
 * loss  l = max((tq + clamp(q - tq, +-clip_eps) - y)^2, (q - y)^2) * w  and d(mean_b l)/dq replace
 * grad_out, and back-propagation continues as in asac_mlp_backward (parameter gradients only).
 *   target_q [E][N], y [N], weights [N] or NULL;  loss_out [E] = mean_b l  (written by the reducing
 *   launch: this call, or asac_adam_step_partials with ASAC_MLP_REDUCE_DEFER) */
int asac_mlp_backward_qloss(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride,
                            int E, const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                            const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                            const float* target_q, const float* y, const float* weights, float clip_eps,
                            float* loss_out, float* grad_params, float* workspace, int reduce_mode,
                            void* stream);
/* ... also returning the gradient of the summed member losses w.r.t. the STATE input, grad_x0 [E][N][in0] (one per
 * member; the caller sums them): a trainable representation's Q step needs no separate critic forward / loss launch
 * (sac_base.py:1539-1570 with the representation in the graph). */
int asac_mlp_backward_qloss_gx(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride, int E,
                               const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                               const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                               const float* target_q, const float* y, const float* weights, float clip_eps,
                               float* loss_out, float* grad_x0, float* grad_params, float* workspace, int reduce_mode,
                               void* stream);

/* asac_mlp_backward_qloss forming its own return target (sac_base.py:1423-1464 + 1244-1295 inside 1539-1570): every
 * workgroup evaluates the n-step V-trace return `ret` describes for ITS tile's rows — the loads travel under the weight
 * staging, the per-step terms and the scan association are asac_vtrace_return_min's (bit-identical y) — so no return
 * launch precedes the Q step.  ret->y_out [N] is written as well (member 0's workgroups); ret->td_error_out must be
 * NULL, ret->B == N; grad_x0 as in asac_mlp_backward_qloss_gx (or NULL).  `_ok`: stock three-block network, n <= 16, the tile's steps fit LDS (otherwise: the two launches). */
int asac_mlp_backward_qloss_return_ok(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride, int E,
                                      int64_t N, const asac_vtrace_args_t* ret);
int asac_mlp_backward_qloss_return(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride, int E,
                                   const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                                   const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                                   const float* target_q, const asac_vtrace_args_t* ret, const float* weights,
                                   float clip_eps, float* loss_out, float* grad_x0, float* grad_params, float* workspace,
                                   int reduce_mode, void* stream);

/* The policy step's Q backward (sac_base.py:1896-1903): the gradient of mean_b(-min_{e in subset} q_e)
 * w.r.t. the ensemble outputs is formed on chip from the value table q_table [E][N] the preceding
 * asac_mlp_forward produced (-1/N at the first arg-min member of the subset, else 0) and pushed back to
 * the ACTION input only: grad_x1 [E][N][in1] (one gradient per member; asac_squash_sample_bwd sums them).
 * subset: device i32[E_sample] or NULL (= members 0..E_sample-1). */
int asac_mlp_backward_policy_q(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride,
                               int E, const float* x0, int64_t x0_row_stride, int64_t x0_member_stride,
                               const float* x1, int64_t x1_row_stride, int64_t x1_member_stride, int64_t N,
                               const float* q_table, const int32_t* subset, int E_sample, float* grad_x1,
                               void* stream);

/* The WHOLE policy step of the stock networks with two critics in one launch (sac_base.py:1883-1906): the critics'
 * forward on (x, action), d(mean_b -min_e q_e)/dq, both critics' backward to the action, the rsample / tanh /
 * log-prob backward (dL/dlogp = exp(*log_alpha) / N) and the policy's backward — what asac_mlp_forward +
 * asac_mlp_backward_policy_q (E = E_sample = 2) + asac_mlp_backward_policy_sample compute in three launches,
 * bit for bit (same MFMA chains).  An ensemble of more than two critics with E_sample = 2 qualifies too: `subset`
 * (device i32[2], NULL = members 0, 1) names the two the objective samples — the others get no gradient from it and
 * are not evaluated; q_out is then [E][N] with those two rows written.
 * A workgroup of 8 waves owns a 16-row tile end to end; nothing but the policy's
 * per-tile parameter-gradient partials (workspace, tiles = asac_mlp_backward_tiles(N, 1); reduce_mode as above)
 * and the value table q_out [2][N] (optional) leaves the chip.
 *   x [N] rows of in0 floats (row stride x_row_stride), action [N][A] = tanh(loc + eps * scale), eps [N][A].
 *   action == NULL: the action is sampled HERE from a first run of the policy on the tile (what asac_mlp_forward +
 *   asac_squash_sample_fwd would have produced, bit for bit) and stored in a_tanh_out [N][A] with its log-probability
 *   logp_out [N] and, optionally, the policy's output ls_out [N][2A] (loc | scale).
 * asac_policy_step_fused_ok: 1 when the shapes qualify (both networks three 64-wide blocks on <= 64 inputs with
 * 16-byte aligned weights, scalar-head critics on (in0 | A), Gaussian-head policy on in0, N <= 4096). */
int asac_policy_step_fused_ok(const asac_mlp_desc_t* q_desc, const float* q_params, int64_t q_member_stride,
                              const asac_mlp_desc_t* pi_desc, const float* pi_params, int64_t pi_member_stride, int64_t N);
int asac_policy_step_fused(const asac_mlp_desc_t* q_desc, const float* q_params, int64_t q_member_stride,
                           const asac_mlp_desc_t* pi_desc, const float* pi_params, int64_t pi_member_stride,
                           const float* x, int64_t x_row_stride, int64_t N, const float* action, const float* eps,
                           const float* log_alpha, const int32_t* subset, float* q_out, float* a_tanh_out,
                           float* logp_out, float* ls_out, float* pi_grad_params, float* workspace, int reduce_mode,
                           void* stream);

/* The policy step's policy backward (sac_base.py:1883-1906, stock Gaussian-head ModelPolicy): the
 * gradient of the objective w.r.t. (loc | scale) — asac_squash_sample_bwd's math with dL/dlogp =
 * exp(*log_alpha) / N and dL/da = sum over grad_a_members of grad_a [m][N][A] — is formed on chip from the
 * forward the backward recomputes, then back-propagated (parameter gradients only; reduce_mode as above). */
int asac_mlp_backward_policy_sample(const asac_mlp_desc_t* desc_host, const float* params, int64_t member_stride,
                                    const float* x0, int64_t x0_row_stride, int64_t N, const float* eps,
                                    const float* grad_a, int grad_a_members, const float* log_alpha,
                                    float* grad_params, float* workspace, int reduce_mode, void* stream);

/* floats of one member's parameter block that the network actually uses (<= member_stride) */
int64_t asac_mlp_param_extent(const asac_mlp_desc_t* desc_host);

/* Gaussian policy head bounding (policy.py:170-172): loc = 5*tanh(mean/5),
 * scale = exp(clamp(logstd, -20, 0.5)); raw = [rows][2A] (mean | logstd).  Backward: graw from
 * gloc / gscale (either may be NULL). */
int asac_gauss_head_fwd(const float* raw, int64_t rows, int A, float* loc, float* scale, void* stream);
int asac_gauss_head_bwd(const float* raw, const float* grad_loc, const float* grad_scale,
                        int64_t rows, int A, float* grad_raw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused GRU stack over a padded window (the RNN burn-in, sac_base.py:1117-1146 `get_l_states`
 * through nn_models/layers/seq_layers.py:14-114 `GRU`): nn.GRU(batch_first=True) cell semantics,
 * gate order (r, z, n).  Padding as in that layer: steps before the first unpadded step are skipped
 * (state untouched), later steps all run, and the output of every padded step is zero.  One launch
 * per pass instead of one MIOpen launch per time step.
 *   w_ih/w_hh/b_ih/b_hh  HOST arrays of `layers` device pointers: [3H][I_l], [3H][H], [3H], [3H]
 *   x             [B][L][input] with strides (floats) x_stride_b / x_stride_t
 *   h0            [B][layers][H] with batch stride h0_stride_b floats, or NULL (zeros)
 *   padding_mask  [B][L] bytes, row stride mask_stride_b, or NULL
 *   hn_out        [B][L][layers][H]   every layer's output at every step (top layer = the output;
 *                                     the state after the last valid step = next hidden state)
 *   out_top       [B][L][H]           the top layer's output once more, dense (may be NULL)
 *   gates_out     [B][L][layers][5H]  (r, z, n, W_hn h + b_hn, unmasked state) saved for backward;
 *                                     NULL = inference
 * Limits: input, hidden <= ASAC_GRU_MAX_DIM, layers <= ASAC_GRU_MAX_LAYERS, hidden_pow2 = hidden
 * rounded up to a power of two; anything else returns ASAC_ERR_BAD_ARG (callers keep their own
 * generic path for such cells).
 * ------------------------------------------------------------------------------------------- */
#define ASAC_GRU_MAX_LAYERS 2
#define ASAC_GRU_MAX_DIM 16
typedef struct {
    int32_t input, hidden, hidden_pow2, layers;
} asac_gru_desc_t;

/* floats of the packed parameter-gradient buffer: per layer w_ih | w_hh | b_ih | b_hh */
int64_t asac_gru_param_count(const asac_gru_desc_t* desc_host);
int64_t asac_gru_backward_workspace(const asac_gru_desc_t* desc_host, int B);

int asac_gru_forward(const asac_gru_desc_t* desc_host, const float* const* w_ih, const float* const* w_hh,
                     const float* const* b_ih, const float* const* b_hh, const float* x,
                     int64_t x_stride_b, int64_t x_stride_t, const float* h0, int64_t h0_stride_b,
                     const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L, float* hn_out,
                     float* out_top, float* gates_out, void* stream);

/* The same launch with a SECOND parameter set run over the same window for inference: the target
 * representation's pass beside the online one's (sac_base.py:2066-2079 calls `get_l_states` for `model_rep`
 * and then for `model_target_rep` on identical inputs).  twin_hn_out [B][L][layers][H], twin_out_top [B][L][H]
 * (may be NULL); the twin saves no activations.  Values are those of two asac_gru_forward calls. */
int asac_gru_forward_twin(const asac_gru_desc_t* desc_host, const float* const* w_ih, const float* const* w_hh,
                          const float* const* b_ih, const float* const* b_hh, const float* const* twin_w_ih,
                          const float* const* twin_w_hh, const float* const* twin_b_ih,
                          const float* const* twin_b_hh, const float* x, int64_t x_stride_b, int64_t x_stride_t,
                          const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask,
                          int64_t mask_stride_b, int B, int L, float* hn_out, float* out_top, float* gates_out,
                          float* twin_hn_out, float* twin_out_top, void* stream);

/* BPTT of the above.  grad_hn [B][L][layers][H] (gradient w.r.t. hn_out) and grad_top [B][L][H]
 * (gradient w.r.t. out_top) may each be NULL; grad_x [B][L][input] and grad_h0 [B][layers][H] are
 * written (either may be NULL).  Parameter gradients, summed in a fixed order: either WRITTEN packed into
 * grad_params (asac_gru_param_count floats), or written / added (accumulate != 0) straight into the 4*layers
 * tensors grad_param_tensors points at (HOST array of device pointers: w_ih, w_hh, b_ih, b_hh per layer) —
 * exactly one of the two is non-NULL. */
int asac_gru_backward(const asac_gru_desc_t* desc_host, const float* const* w_ih, const float* const* w_hh,
                      const float* const* b_ih, const float* const* b_hh, const float* x,
                      int64_t x_stride_b, int64_t x_stride_t, const float* h0, int64_t h0_stride_b,
                      const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L, const float* hn,
                      const float* gates, const float* grad_hn, const float* grad_top, float* grad_x,
                      float* grad_h0, float* grad_params, float* const* grad_param_tensors, int accumulate,
                      float* workspace, void* stream);

/* BPTT from ONE window position: the loss reads the representation's state at a single step of the window
 * (SAC_Base._train_rep_q takes `m_states[:, burn_in_step]`, sac_base.py:2104-2110, everything behind it only feeds
 * detached targets), and E critics each hand back d loss / d state there.  The gradient of out_top is
 * sum_e grad_top_members[e][b][:] (summed in member order) at `position` and zero elsewhere, grad_hn is zero — the
 * values asac_gru_backward returns for that dense gradient, bit for bit, but the recursion starts AT the position
 * (the steps behind it carry zeros) and no [B][L][H] gradient tensor is formed or read.
 *   grad_top_members [members][B][H];  0 <= position < L;  grad_x (when asked for) is zero behind the position.
 *   adam (may be NULL; grad_param_tensors form only): the launch that finishes the parameter gradients also takes the
 *   optimizer step of those parameters — the representation's `optimizer_rep.step()` (sac_base.py:1601-1603) without a
 *   launch of its own. */
typedef struct {
    float* param_base;          /* the parameter whose gradient is grad_base[k] is param_base[k] ...             */
    const float* grad_base;     /* ... (the learner's flat buffers: every grad_param_tensors entry points inside) */
    float* exp_avg_base;        /* Adam moments at the same offsets                                              */
    float* exp_avg_sq_base;
    float lr, beta1, beta2, eps;
    const int64_t* steps_done;  /* device counter; the update uses step *steps_done + 1 and does not advance it  */
} asac_adam_epilogue_t;
int asac_gru_backward_at(const asac_gru_desc_t* desc_host, const float* const* w_ih, const float* const* w_hh,
                         const float* const* b_ih, const float* const* b_hh, const float* x,
                         int64_t x_stride_b, int64_t x_stride_t, const float* h0, int64_t h0_stride_b,
                         const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L, const float* hn,
                         const float* gates, const float* grad_top_members, int members, int position,
                         float* grad_x, float* grad_h0, float* grad_params, float* const* grad_param_tensors,
                         int accumulate, const asac_adam_epilogue_t* adam, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GRU recurrence for hidden sizes 32 / 64 / 128 on f32 MFMA (csrc/gru_wide.hip): the time loops of ONE layer of the
 * plugin layer `GRU` (nn_models/layers/seq_layers.py:14-114; `m.GRU(…, 64, 1)` envs/square/memory_corridor/nn.py:19,
 * `m.GRU(…, 128, 1)` envs/uav/uav_hole/nn.py:22) under `get_l_states` (sac_base.py:1117-1146).  The batched products
 * around them are library GEMMs on the host side (algorithm/fused_gru_wide.py): gi = x W_ih^T + b_ih before, dW_ih =
 * dgi^T x, dW_hh = dgh^T h_prev, dx = dgi W_ih and the bias sums after.
 *   forward : gi [B][L][3H] (strides in floats), w_hh [3H][H], b_hh [3H], h0 [B][H] (row stride) or NULL, padding_mask
 *             [B][L] bytes or NULL -> out (the masked output, strided: may be a layer's slice of hn [B][L][layers][H]),
 *             h_raw [B][L][H] (the unmasked state after every step) and gates [B][L][4H] (r | z | n | W_hn h + b_hn): both
 *             saved for the backward, both or neither NULL.  Padding as asac_gru_forward: steps before a row's first
 *             unpadded one are skipped (state held), the output of a padded step is 0.
 *   backward: grad_out [B][L][H] (gradient of the masked output, strided), w_hh_t [H][3H] = W_hh^T -> grad_gi, grad_gh
 *             [B][L][3H] (gradients of gi and of W_hh h + b_hh; zero where a step did not run), grad_h0 [B][H] or NULL.
 * Every pointer 16-byte aligned, every stride a multiple of 4 floats.  Deterministic (no atomics).
 * ------------------------------------------------------------------------------------------- */
int asac_gru_wide_supported(int hidden);
int asac_gru_wide_forward(const float* gi, int64_t gi_stride_b, int64_t gi_stride_t, const float* w_hh, const float* b_hh,
                          const float* h0, int64_t h0_stride_b, const uint8_t* padding_mask, int64_t mask_stride_b, int B,
                          int L, int hidden, float* out, int64_t out_stride_b, int64_t out_stride_t, float* h_raw,
                          float* gates, void* stream);
/* Twin form: the same B windows under TWO networks' recurrent weights in one launch (the online representation and its target
 * copy over the sampled windows, reference sac_base.py:2066-2079): gi / out hold 2B rows (network 1 then network 2),
 * h0 / padding_mask B rows (shared), h_raw / gates B rows (network 1 only: the target pass is not differentiated).  B % 16 == 0. */
int asac_gru_wide_forward_twin(const float* gi, int64_t gi_stride_b, int64_t gi_stride_t, const float* w_hh, const float* b_hh,
                               const float* w_hh_twin, const float* b_hh_twin, const float* h0, int64_t h0_stride_b,
                               const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L, int hidden, float* out,
                               int64_t out_stride_b, int64_t out_stride_t, float* h_raw, float* gates, void* stream);
int asac_gru_wide_backward(const float* grad_out, int64_t go_stride_b, int64_t go_stride_t, const float* w_hh_t,
                           const float* gates, const float* h_raw, const float* h0, int64_t h0_stride_b,
                           const uint8_t* padding_mask, int64_t mask_stride_b, int B, int L, int hidden, float* grad_gi,
                           float* grad_gh, float* grad_h0, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused two-layer convolution stack: Conv2d(C->out1, kernel1, stride1) GELU Conv2d(out1->out2, kernel2,
 * stride2) GELU (no padding, square kernels) over N small frames — the `simple` visual encoder
 * (nn_models/layers/image_layers.py:70-80) that `get_l_states` (sac_base.py:1117-1146) applies to every
 * frame of the sampled windows, three times per train step.  One launch per pass.
 *   x  [N][C][H][W];  w1 [out1][C][k1][k1], b1 [out1];  w2 [out2][out1][k2][k2], b2 [out2]  (nn.Conv2d layout)
 *   y  [N][out2*H2*W2]  = the second activation map flattened channel-major (what `.reshape(N, -1)` of the NCHW
 *      map gives)
 *   z1_out [N][H1*W1][out1], z2_out [N][out2*H2*W2]: pre-activations saved for the backward (both or neither)
 * Backward: gradients of the four parameter tensors only (frames are data), packed w1 | b1 | w2 | b2 into
 * grad_params (asac_conv2_param_count floats; written, or added with accumulate != 0 — the layout of the
 * parameters' gradient views inside the learner's flat buffer), summed in a fixed order; workspace of
 * asac_conv2_backward_workspace floats.
 * Limits (asac_conv2_supported): out1 <= 16, out2 <= 32, C*k1*k1 and out1*k2*k2 multiples of 16 and
 * <= ASAC_CONV2_MAX_K, H2*W2 a divisor of 16, a group of 16/(H2*W2) frames within the LDS budget; anything else
 * returns ASAC_ERR_BAD_ARG and callers keep their generic path.
 * ------------------------------------------------------------------------------------------- */
#define ASAC_CONV2_MAX_K 256
typedef struct {
    int32_t channels, height, width;
    int32_t out1, kernel1, stride1;
    int32_t out2, kernel2, stride2;
} asac_conv2_desc_t;

int asac_conv2_supported(const asac_conv2_desc_t* desc_host);
/* Frames whose second-layer map has more than 16 positions (the 84 x 84 frames of the reference's environments:
 * ConvLayers(84, 84, C, 'simple'), 20 x 20 -> 9 x 9, envs/roller/nn_visual_hard_attn.py) run TILED: asac_conv2_tiles(desc)
 * blocks of <= 16 positions per frame, each formed from the crop of the frame it needs (csrc/conv.hip).  Same results;
 * z1_out then holds asac_conv2_z1_floats(desc, N) floats (the crops' first-layer pre-activations, not the frame's map:
 * per block, or — where the forward works by rows of blocks — per row of blocks; an opaque buffer between the two passes). */
int asac_conv2_tiles(const asac_conv2_desc_t* desc_host);
int64_t asac_conv2_z1_floats(const asac_conv2_desc_t* desc_host, int64_t N);
int64_t asac_conv2_param_count(const asac_conv2_desc_t* desc_host);
int64_t asac_conv2_backward_workspace(const asac_conv2_desc_t* desc_host, int64_t N);
/* partial slabs (one per workgroup, n_cot x the packed parameter count floats each) a backward launch over N frames with
 * n_cot cotangents leaves in its workspace — what asac_sum_partials_multi sums after accumulate == ASAC_CONV_SUM_DEFER */
int asac_conv2_backward_slabs(const asac_conv2_desc_t* desc_host, int64_t N, int n_cot);
int asac_conv2_forward(const asac_conv2_desc_t* desc_host, const float* x, int64_t N, const float* w1,
                       const float* b1, const float* w2, const float* b2, float* y, float* z1_out, float* z2_out,
                       void* stream);
/* The same forward over a SLICE of the sampled windows read where it lies: x points at frame [0][b] of a
 * [B][L][C][H][W] batch, a sample's frames_per_sample frames are consecutive, samples sample_stride floats apart
 * (N = B * frames_per_sample).  A representation without a sequence encoder only needs the states of the positions
 * behind the burn-in before its update (sac_base.py:2066-2103: `m_states[:, burn_in_step:]`), so those two passes skip
 * the burn-in frames without a copy.  frames_per_sample must be a multiple of asac_conv2_group_frames(desc) (a
 * workgroup's frames then never straddle two samples); anything else returns ASAC_ERR_BAD_ARG (callers copy). */
int asac_conv2_group_frames(const asac_conv2_desc_t* desc_host);
int asac_conv2_forward_windows(const asac_conv2_desc_t* desc_host, const float* x, int64_t N, int frames_per_sample,
                               int64_t sample_stride, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* y, float* z1_out, float* z2_out, void* stream);
/* ... and the backward of such a pass: x addressed the same way, z1 / z2 / grad_y dense over the N frames. */
int asac_conv2_backward_windows(const asac_conv2_desc_t* desc_host, const float* x, int64_t N, int frames_per_sample,
                                int64_t sample_stride, const float* w2, const float* z1, const float* z2,
                                const float* grad_y, float* grad_params, int accumulate, float* workspace, void* stream);

/* Several backward walks of ONE forward pass as one launch: the reference differentiates the representation's graph once per
 * gated auxiliary loss (`calculate_adaptive_weights`, sac_base.py:1607-1631, called from 1798-1839) — the same frames, the same
 * saved pre-activations, n_cot (1..4) different output gradients grad_ys[c] [N][out2 * M2].  What does not depend on the
 * cotangent (the frames' staging, the activations, the gathered patch operands of both weight-gradient products) is done once
 * per group of frames.  grads_out [n_cot][asac_conv2_param_count]: cotangent c's packed gradients w1 | b1 | w2 | b2, bit-identical
 * to asac_conv2_backward_windows(..., grad_ys[c], ...); workspace: n_cot x asac_conv2_backward_workspace floats.  More
 * cotangents than asac_conv2_backward_multi_max (their LDS buffers beside the double-buffered frames: 3 for 30 x 30 frames)
 * go in launches of at most that many. */
int asac_conv2_backward_multi_max(const asac_conv2_desc_t* desc);     /* cotangents ONE launch takes for these frames (1..3) */
int asac_conv2_backward_multi(const asac_conv2_desc_t* desc, const float* x, int64_t N, int frames_per_sample,
                              int64_t sample_stride, const float* w2, const float* z1, const float* z2,
                              const float* const* grad_ys, int n_cot, float* grads_out, int accumulate, float* workspace,
                              void* stream);
int asac_conv2_backward(const asac_conv2_desc_t* desc_host, const float* x, int64_t N, const float* w2,
                        const float* z1, const float* z2, const float* grad_y, float* grad_params, int accumulate,
                        float* workspace, void* stream);

/* Cosine-sign gating of auxiliary gradients: `calculate_adaptive_weights` (sac_base.py:1607-1631) after its autograd
 * calls.  main / aux_k / grad: flat f32[n] (the representation's gradient segment and K <= 4 auxiliary gradients of the
 * same layout; aux_host = host array of K device pointers):  gate_k = clamp(sign(cos(main, aux_k)), min = 0)  — the sign
 * of the dot product —, then grad += gate_k * aux_k for k = 0 .. K-1 in that order.  gates_out (f32[K], or NULL) receives
 * the gates.  One workgroup, fixed summation order (deterministic); n <= ASAC_GATE_MAX_N. */
#define ASAC_GATE_MAX_LOSSES 4
#define ASAC_GATE_MAX_N (1 << 20)
int asac_cosine_gate_add(const float* main, const float* const* aux_host, int K, int64_t n, float* grad, float* gates_out,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Observation decoder of the recurrent prediction models (`use_prediction`): `ConvTransposeLayers`
 * (nn_models/layers/image_layers.py:231-253) with the decoder every reference plugin with an image observation
 * model builds (envs/roller/nn_visual_hard_attn.py:64-96, envs/roller/nn_visual_hard.py:47-57,
 * envs/pyramid/nn_visual.py:50-60), called from `_train_rpm` (sac_base.py:1798-1839) through the plugin's
 * `ModelObservation.get_loss`:
 *   state [N, S <= 16] -> Linear(S, 64) GELU -> Linear(64, 128) -> [32, 2, 2] -> ConvTranspose2d(32, 32, 4, 2) LeakyReLU
 *   -> ConvTranspose2d(32, 16, 8, 4) LeakyReLU -> ConvTranspose2d(16, 3, 3, 1) LeakyReLU -> frames [N, 3, 30, 30]
 * as f32 MFMA products with the states as the N dimension (csrc/decoder.hip).  Parameters in PyTorch's layouts
 * (Linear [out, in]; ConvTranspose2d [in, out, kh, kw]).
 *   forward : packs the weights into operand order (`packed`, asac_obs_decoder_packed_floats floats), keeps the
 *             layers' activations in `saved` (asac_obs_decoder_saved_floats(N) floats) for the backward, writes `frames`
 *   backward: grad_frames [N, 3, 30, 30] -> grad_state [N, S] (or NULL) and the ten parameter gradients (written, or
 *             added with accumulate != 0; per-group partial sums added in group order: deterministic);
 *             `workspace` asac_obs_decoder_workspace_floats(N) floats; `packed` / `saved` / `frames` as the forward left them
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float* dense1_w;  /* [64, S]  */
    float* dense1_b;  /* [64]     */
    float* dense2_w;  /* [128, 64] */
    float* dense2_b;  /* [128]    */
    float* ct1_w;     /* [32, 32, 4, 4] */
    float* ct1_b;     /* [32]     */
    float* ct2_w;     /* [32, 16, 8, 8] */
    float* ct2_b;     /* [16]     */
    float* ct3_w;     /* [16, 3, 3, 3]  */
    float* ct3_b;     /* [3]      */
} asac_obs_decoder_params_t;

int64_t asac_obs_decoder_packed_floats(void);
int64_t asac_obs_decoder_saved_floats(int64_t N);
int64_t asac_obs_decoder_workspace_floats(int64_t N);
int asac_obs_decoder_forward(const float* state, int64_t state_stride, int64_t N, int state_size,
                             const asac_obs_decoder_params_t* params_host, float* packed, float* saved, float* frames,
                             void* stream);
int asac_obs_decoder_backward(const float* state, int64_t state_stride, int64_t N, int state_size, const float* packed,
                              const float* saved, const float* frames, const float* grad_frames, float* grad_state,
                              const asac_obs_decoder_params_t* grad_params_host, int accumulate, float* workspace,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention core for short windows on f32 MFMA (csrc/attn_mh.hip): the scores / mask / softmax / weighted-sum
 * part of `MultiheadAttention.forward` (nn_models/layers/seq_layers.py:239-333) for num_heads >= 1 heads of head_dim <= 64
 * channels and windows of <= 32 positions — the widths of the reference's environments (`EpisodeMultiheadAttention(64, …)`
 * with 2 - 8 heads: envs/square/obstacle/nn_attn.py:16, envs/gym/toy_queue/nn_attn.py:28-45).  The q / k / v / output
 * projections around it stay Linears (library GEMMs) on the host side.
 *   q [B][Lq][E], k / v [B][Lk][E], E = heads * head_dim, head h = channels [h * head_dim, (h + 1) * head_dim);
 *   mask as asac_attention_forward (shared by the heads).  out [B][Lq][E] (heads concatenated);  weights [B][Lq][Lk] = the
 *   mean over the heads of softmax(s) * keep;  keep [B][Lq] = 1 - dead;  p_heads [B][heads][Lq][Lk] (or NULL): every head's
 *   softmax, saved for the backward.  row_zero u8 [B][Lq] (or NULL) = the caller's padded positions and keep_rows [B][Lq]
 *   (or NULL) <- keep * !row_zero: the ONE factor the caller multiplies the projected output with (`out * keep` and
 *   `out * ~out_row_mask` of seq_layers.py:320-333 as one product).
 * Backward: grad_out [B][Lq][E], grad_weights [B][Lq][Lk] (w.r.t. the returned weights) or NULL -> grad_q / grad_k / grad_v.
 * One workgroup per batch entry, its heads dealt over four waves; sums in a fixed order: deterministic.
 * ------------------------------------------------------------------------------------------- */
int asac_attention_mh_supported(int Lq, int Lk, int heads, int head_dim);
int asac_attention_mh_forward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                              int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                              float* out, float* weights, float* keep, float* p_heads, const uint8_t* row_zero, float* keep_rows, void* stream);
int asac_attention_mh_backward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                               int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                               const float* p_heads, const float* grad_out, const float* grad_weights, float* grad_q,
                               float* grad_k, float* grad_v, void* stream);

/* asac_attention_mh_forward with the q / k / v projections of the block's input in front of the scores, in the same launch
 * (`self.q_proj(query), self.k_proj(key), self.v_proj(value)` with value = key and query = the last Lq positions of key,
 * seq_layers.py:239-333): x [B][Lk][E] with strides in floats (multiples of 4, 16-byte aligned), weights / biases = HOST arrays
 * of the three [E][E] / [E] device pointers (q, k, v order); q [B][Lq][E], k / v [B][Lk][E] are OUTPUTS (dense; what the
 * backward reads); row_zero [B][Lq] with a batch stride in bytes (a slice of a wider mask is read in place).  Windows Lq <= Lk <= 16, E = heads * head_dim in {32, 64, 128}, head_dim a multiple of 4.  out_weight non-NULL:
 * the output ResBlock of asac_rows_resblock_forward runs behind the core in the same launch — y / pre [B][Lq][E] as there, with
 * row_scale = keep * !row_zero (`out` still receives the core's own output: the ResBlock's input).  The backward is
 * asac_rows_resblock_backward (if fused), asac_attention_mh_backward, asac_rows_proj_backward. */
int asac_attention_mh_proj_supported(int Lq, int Lk, int heads, int head_dim);
int asac_attention_mh_proj_forward(const float* x, int64_t x_stride_b, int64_t x_stride_t, const float* const* weights,
                                   const float* const* biases, const uint8_t* mask, int64_t mask_stride_b,
                                   int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                                   float* q, float* k, float* v, float* out, float* attn_weights, float* keep, float* p_heads,
                                   const uint8_t* row_zero, int64_t row_zero_stride_b, float* keep_rows, const float* out_weight,
                                   const float* out_bias, float* y, float* pre, void* stream);

/* The backward of asac_attention_mh_proj_forward WITH its output block as one launch: asac_rows_resblock_backward in front of the
 * core's backward (grad_y [B][Lq][E] with strides in floats — multiples of 4, feature stride 1 —, pre, row_scale = the forward's keep_rows or NULL, out_weight -> grad_pre [B][Lq][E], and the gradient
 * of the core's output, which stays on chip), asac_attention_mh_backward (-> grad_q / grad_k / grad_v, written for the parameter
 * gradient products), asac_rows_proj_backward behind it (proj_weights = HOST array of the three [E][E] device pointers ->
 * grad_x [B][Lk][E]).  Same values as the three launches. */
int asac_attention_mh_block_backward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                                     int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int heads, int head_dim,
                                     const float* p_heads, const float* grad_y, int64_t grad_y_stride_b, int64_t grad_y_stride_t,
                                     const float* pre, const float* row_scale, const float* out_weight, const float* grad_weights,
                                     const float* const* proj_weights, float* grad_q, float* grad_k, float* grad_v, float* grad_pre,
                                     float* grad_x, void* stream);

/* The Linear layers around that core over the rows of a batch of windows, one launch each (csrc/rows_proj.hip) — replaces, in
 * `MultiheadAttention.forward` (nn_models/layers/seq_layers.py:239-333): `self.q_proj(query), self.k_proj(key),
 * self.v_proj(value)` (three nn.Linear(E, E) = three library GEMMs at the launch floor; backward three GEMMs and two
 * accumulations) and `self.out_proj` when it is ONE residual GELU ResBlock (LinearLayers(E, E, dense_depth 1),
 * linear_layers.py:24-119) followed by the dead-row / padded-row factor (a GEMM and three elementwise launches).
 * width = E in {32, 64, 128}; all pointers 16-byte aligned; weights [E][E] (out x in), biases [E].
 *
 * asac_rows_proj_forward: x [batch][window][E] with strides in floats (feature stride 1; multiples of 4); job j (n_jobs <= 3)
 * writes outs[j] [batch][tails[j]][E] = x[:, window - tails[j]:] weights[j]^T + biases[j]  (tails[j] in 1..window: the cut
 * query of the episode blocks projects the newest positions only).  The pointer arrays are HOST arrays read at the call.
 * asac_rows_proj_backward: grad_x [batch][window][E] (dense, overwritten) = sum_j grads[j] weights[j], job j reaching the
 * newest tails[j] positions only; summation order: jobs in order, features in order.
 * asac_rows_resblock_forward: y = (x + gelu(x W^T + b)) * row_scale[row] (row_scale NULL: 1), pre = x W^T + b (saved for the
 * backward); x [rows][E] with row stride x_row_stride (multiple of 4), y / pre dense.
 * asac_rows_resblock_backward: g = grad_y * row_scale; grad_pre = g * gelu'(pre) (dense: its products over the rows with x
 * are the parameter gradients, asac_xty); grad_x = g + grad_pre W. */
int asac_rows_proj_supported(int width);
int asac_rows_proj_forward(const float* x, int64_t x_stride_b, int64_t x_stride_t, int batch, int window, int width, int n_jobs,
                           const float* const* weights, const float* const* biases, const int* tails, float* const* outs,
                           void* stream);
int asac_rows_proj_backward(const float* const* grads, const int* tails, int n_jobs, const float* const* weights, int batch,
                            int window, int width, float* grad_x, void* stream);
int asac_rows_resblock_forward(const float* x, int64_t x_row_stride, const float* weight, const float* bias,
                               const float* row_scale, int64_t rows, int width, float* y, float* pre, void* stream);
int asac_rows_resblock_backward(const float* grad_y, const float* pre, const float* weight, const float* row_scale,
                                int64_t rows, int width, float* grad_x, float* grad_pre, void* stream);
/* y [rows][N] = x [rows][K] weight[N][K]^T + bias[N] for a narrow input (K <= 64) and a wide output (N a multiple of 16, <= 1024):
 * the input products `x W_ih^T + b_ih` of every step in front of a recurrence (reference seq_layers.py:14-114 through nn.GRU;
 * observation ++ action -> 3 x hidden); x with a row stride in floats, y dense and 16-byte aligned. */
int asac_rows_affine_supported(int K, int N);
int asac_rows_affine_forward(const float* x, int64_t x_row_stride, int K, const float* weight, const float* bias, int64_t rows,
                             int N, float* y, void* stream);
/* ... with the GELU of a `ResBlock` without a residual path (widths differ: the embedding `LinearLayers(n_in, 64, dense_depth 1)`
 * in front of a sequence encoder, linear_layers.py:24-119): y = gelu(x W^T + b), pre = x W^T + b (dense, saved for the backward). */
int asac_rows_affine_gelu_forward(const float* x, int64_t x_row_stride, int K, const float* weight, const float* bias,
                                  int64_t rows, int N, float* y, float* pre, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention core for short windows: the scores / mask / softmax / weighted-sum part of
 * `MultiheadAttention.forward` (nn_models/layers/seq_layers.py:239-333) under the episode attention of
 * `get_l_states` (sac_base.py:1117-1146).  One head per batch entry (callers fold heads into the batch).
 *   q [B][Lq][D], k / v [B][Lk][D] dense;  mask element (b, i, j) at mask + b*stride_b + i*stride_q + j*stride_k
 *   bytes, nonzero = blocked, or NULL (strides of 0 broadcast a [Lq][Lk] or [B][1][Lk] mask)
 *   scores s_ij = (q_i / sqrt(D)) . k_j; blocked -> -inf, except on rows whose every key is blocked ("dead" rows:
 *   they attend unmasked and are zeroed by the caller after its output projection, like the reference)
 *   out [B][Lq][D] = softmax(s) v;  weights [B][Lq][Lk] = softmax(s) * keep;  keep [B][Lq] = 1 - dead
 * Backward: grad_weights (w.r.t. the returned weights) may be NULL; writes grad_q / grad_k / grad_v.
 * Limits: Lq, Lk <= ASAC_ATTN_MAX_LEN, D <= ASAC_ATTN_MAX_DIM (asac_attention_supported).
 * ------------------------------------------------------------------------------------------- */
#define ASAC_ATTN_MAX_LEN 32
#define ASAC_ATTN_MAX_DIM 16
int asac_attention_supported(int Lq, int Lk, int D);
int asac_attention_forward(const float* q, const float* k, const float* v, const uint8_t* mask, int64_t mask_stride_b,
                           int64_t mask_stride_q, int64_t mask_stride_k, int B, int Lq, int Lk, int D, float* out,
                           float* weights, float* keep, void* stream);
int asac_attention_backward(const float* q, const float* k, const float* v, const float* weights,
                            const float* grad_out, const float* grad_weights, int B, int Lq, int Lk, int D,
                            float* grad_q, float* grad_k, float* grad_v, void* stream);

/* The same with the three input projections on chip (seq_layers.py:268-276 with qkv_dense_depth = 0: q_proj,
 * k_proj, v_proj are Linear(E, E)):  q = Wq x_q + bq, k = Wk x_k + bk, v = Wv x_k + bv, one head of E <= 16 channels.
 *   x_q element (b, i, c) at x_q + b*stride_b + i*stride_r + c (floats; the query slice of the key window needs no
 *   copy), x_k likewise;  params = HOST array of 8 device pointers Wq [E][E], bq [E], Wk, bk, Wv, bv, Wo, bo
 *   Wo / bo (both or neither; NULL = none): the output ResBlock of out_dense_depth = 1 applied to the attention
 *   output o on chip, with the dead-row rule:  out = (GELU(Wo o + bo) + o) * keep;  attn_out [B][Lq][E] then
 *   receives o (the backward reads it back together with keep);  row_zero (optional, with the output block):
 *   bytes [B][Lq] with the given strides, nonzero = the row's output is zeroed as well (the episode block's
 *   padded positions, seq_layers.py:445-447)
 * Backward recomputes the projections; grad_out element (b, i, c) at grad_out + b*stride_b + i*stride_r + c (a slice of a
 * longer gradient needs no copy); grad_xq [B][Lq][E] and grad_xk [B][Lk][E] are written dense — grad_xq == NULL: the
 * queries are the LAST Lq key rows (the episode blocks' cut query, seq_layers.py:600-610) and their input gradient is added
 * to those rows of grad_xk inside the launch; the parameter
 * gradients, packed Wq | bq | Wk | bk | Wv | bv (| Wo | bo) (3 or 4 times E*E+E floats), are written or (accumulate == 1) added to
 * grad_params after a fixed-order reduction over workgroups; accumulate == ASAC_ATTN_SUM_DEFER (2): the reduction is left to
 * asac_sum_partials_multi (the partials stay in the workspace); workspace of asac_attention_proj_workspace floats. */
int64_t asac_attention_proj_workspace(int B, int Lq, int Lk, int E);
int asac_attention_proj_forward(const float* xq, int64_t xq_stride_b, int64_t xq_stride_r, const float* xk,
                                int64_t xk_stride_b, int64_t xk_stride_r, const float* const* params_host,
                                const uint8_t* mask, int64_t mask_stride_b, int64_t mask_stride_q,
                                int64_t mask_stride_k, int B, int Lq, int Lk, int E, float* out, float* weights,
                                float* keep, float* attn_out, const uint8_t* row_zero, int64_t row_zero_stride_b,
                                int64_t row_zero_stride_q, void* stream);
int asac_attention_proj_backward(const float* xq, int64_t xq_stride_b, int64_t xq_stride_r, const float* xk,
                                 int64_t xk_stride_b, int64_t xk_stride_r, const float* const* params_host,
                                 const float* weights, const float* keep, const float* attn_out,
                                 const float* grad_out, int64_t grad_out_stride_b, int64_t grad_out_stride_r,
                                 const float* grad_weights, const uint8_t* row_zero,
                                 int64_t row_zero_stride_b, int64_t row_zero_stride_q, int B, int Lq, int Lk, int E,
                                 float* grad_xq, float* grad_xk, float* grad_params, int accumulate,
                                 float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Parameter updates over flat f32 buffers.
 * ------------------------------------------------------------------------------------------- */

/* K5 Polyak: target = target*(1-tau) + source*tau  (two multiplies then an add, like
 * sac_base.py:761-764).  n floats, 16-byte vector loads when aligned. */
int asac_polyak(float* target, const float* source, int64_t n, float tau, void* stream);

/* Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) over a flat segment.
 * `steps_done` is a device counter holding the number of optimizer steps ALREADY taken; the kernel
 * uses t = *steps_done + 1 for the bias corrections and does not modify it (the caller advances it
 * once per train step, so every optimizer of the step shares one counter).
 * sac_base.py:296-300 (optimizer construction), 1589-1603, 1906-1908, 1942-1944. */
int asac_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, const int64_t* steps_done,
                   void* stream);

/* The same update over the E member blocks (E * member_stride floats from param / grad / moments) of a
 * stock network whose gradients were left as per-tile partial sums by asac_mlp_backward* with
 * ASAC_MLP_REDUCE_DEFER: tile sum (same fixed order), optional accumulation, grad write-out and Adam in
 * one launch.  workspace / tiles as that call used them; used = asac_mlp_param_extent().  loss_out [E]
 * (may be NULL) receives the mean-over-loss_rows loss of the qloss variant. */
int asac_adam_step_partials(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                            float beta1, float beta2, float eps, const int64_t* steps_done,
                            const float* workspace, int64_t tiles, int E, int64_t member_stride,
                            int64_t used, int accumulate, float* loss_out, int64_t loss_rows, void* stream);

/* Every random draw of one train step in one launch: n_uniform f64 in [0, 1) (the stratified PER
 * sample's uniforms, replay_buffer.py:196) and n_normal f32 N(0, 1) (the rsample noise, sac_base.py:1346,
 * 1883, 1927).  Philox4x32-10 keyed by `seed`, counter = (lane, *step_counter): the counter lives in
 * device memory so a launch frozen inside a hipGraph still draws fresh numbers every step.
 * subsets_out [n_subsets][E_sample] i32 (may be NULL / 0): independent uniformly random E_sample-subsets of
 * range(E) in random order — the ensemble members each target / objective uses (`torch.randperm(E)[:Es]`,
 * sac_base.py:1434). */
int asac_noise_fill(uint64_t seed, const int64_t* step_counter, double* uniform_out, int64_t n_uniform,
                    float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets, int E_sample,
                    int E, void* stream);

/* asac_polyak + a gradient-buffer memset + asac_noise_fill as ONE launch: the independent launches a train
 * step begins with (sac_base.py:2512-2514 target update, the optimizers' zero_grad 1589-1603, the step's random
 * draws).  Either of the two leading parts may be absent (n_polyak == 0 / n_zero == 0), not both. */
int asac_step_prologue(float* target, const float* source, int64_t n_polyak, float tau, float* zero_out,
                       int64_t n_zero, uint64_t seed, const int64_t* step_counter, double* uniform_out,
                       int64_t n_uniform, float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets,
                       int E_sample, int E, void* stream);

/* asac_step_prologue and the fused form of asac_sumtree_sample (batch <= 1024, IS weights in the launch) as ONE
 * launch, the first of a captured train step: the first ceil(batch / 256) workgroups sample 256 rows each, drawing their
 * stratified uniforms themselves (the very numbers asac_step_prologue would have stored in uniform_out, which they
 * also fill), the other workgroups are the prologue's.  replay_buffer.py:185-205, 347-354 + sac_base.py:745-764.
 * min_p_out is f32[528], 64-byte aligned, HERE: [0] min p, [1] min ratio, [2..7] the workgroups' exchange (partial minima,
 * two counters) — which must be ZERO before the first launch and are left zero by every launch.
 * is_weights_out == NULL defers the weights (sharded replay: they are normalised by the minimum sampling ratio over
 * all ranks' shards): min_p_out[0] = min p, min_p_out[1] = min p / total, beta untouched; after the MIN all-reduce of
 * min_p_out[1], asac_per_is_weights advances beta and writes the weights. */
int asac_step_prologue_sample(float* target, const float* source, int64_t n_polyak, float tau, float* zero_out,
                              int64_t n_zero, uint64_t seed, const int64_t* step_counter, double* uniform_out,
                              float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets, int E_sample,
                              int E, const float* tree, int capacity, int batch, const int64_t* slot_ids,
                              double* beta_state, double beta_increment, int32_t* leaf_out, float* p_out,
                              int64_t* ids_out, float* is_weights_out, float* min_p_out, void* stream);

/* The pair for batches of 257 .. 1 024 rows on one GPU (hip_config['defer_is_weights']): the sampler's workgroups (256
 * samples each) only leave their minima in min_p_out[2 ..] — no exchange between them, no weights, beta untouched —, and
 * asac_window_gather_pad_w, the NEXT launch, forms the weights in one extra workgroup beside the gather's (minimum over the
 * workgroups' minima, beta advanced first, f64 power: replay_buffer.py:352-354), min_p_out[0] = min p.  Same numbers as
 * asac_step_prologue_sample followed by asac_window_gather_pad, bit for bit. */
int asac_step_prologue_sample_partial(float* target, const float* source, int64_t n_polyak, float tau, float* zero_out,
                                      int64_t n_zero, uint64_t seed, const int64_t* step_counter, double* uniform_out,
                                      float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets, int E_sample,
                                      int E, const float* tree, int capacity, int batch, const int64_t* slot_ids,
                                      int32_t* leaf_out, float* p_out, int64_t* ids_out, float* min_p_out, void* stream);
int asac_window_gather_pad_w(const asac_gather_key_t* keys_host, int n_keys, const int64_t* ids, int batch, int prev_n,
                             int post_n, int capacity, const int32_t* index_ring, const float* p, const float* tree,
                             double* beta_state, double beta_increment, float* is_weights_out, float* min_p_out,
                             void* stream);

/* hipGraphLaunch of an instantiated graph (the captured train step) on `stream`. */
int asac_graph_launch(void* graph_exec, void* stream);

/* Fix-up pass over the captured train step (a hipGraph_t, before it is instantiated): every 1-D memset node is replaced by a
 * kernel node with the same edges (csrc/graph_fix.hip).  On ROCm 7.2 / gfx950 a captured hipMemsetAsync of >= 16 bytes takes
 * effect on the first launch of the instantiated graph only; ATen's split reductions zero their semaphores with one, so a
 * captured bias gradient (`x.sum(0)` of an nn.Linear backward inside `SAC_Base._train_rep_q`, sac_base.py:1916-2010) is wrong
 * from the second replay on.  n_replaced / n_kept (2-D memsets, left alone) may be NULL. */
int asac_graph_replace_memset_nodes(void* graph, int* n_replaced, int* n_kept);

/* Temperature step in one launch: dL/dlog_alpha = mean_b(-logp_b) - target into grad[slot], then the
 * same Adam update as asac_adam_step over the n temperature parameters (param / grad / moments point
 * at the alpha segment).  sac_base.py:1913-1949 (continuous head) + 1942-1944.  Single-GPU form; with
 * a gradient all-reduce between the two halves use asac_alpha_grad + asac_adam_step.  This is the last
 * optimizer launch of a train step: with advance_counter != 0 it also advances *steps_done by one
 * (after reading it), which saves the step a counter-increment launch. */
int asac_alpha_adam_step(const float* logp, int B, float target, int slot, float* param, float* grad,
                         float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1, float beta2,
                         float eps, int64_t* steps_done, int advance_counter, void* stream);

/* Curiosity (SAC_Base._get_y, sac_base.py:1333-1343): reward[b][t] += strength * 0.5 * sum_k (approx - actual)^2 over
 * the sampled window, in place.  approx [B][T][K] dense (the dynamics model's output); actual [B][T][K] with strides in
 * floats (the next states of the window for the FORWARD model, the stored actions for the INVERSE one); reward [B][T]
 * with batch stride reward_stride_b.  One launch for ATen's subtract, square, sum, scale and add. */
int asac_curiosity_bonus(const float* approx, const float* actual, int64_t actual_stride_b, int64_t actual_stride_t,
                         float* reward, int64_t reward_stride_b, int B, int T, int K, float strength, void* stream);
/* Loss of the curiosity model and its gradient (SAC_Base._train_curiosity, sac_base.py:1951-1976):
 *   d = (pred - target) * !padding_mask[b][t];  *loss_out = sum d^2 / N;  grad_out = d * 2 / N,  N = B T K
 * pred / grad_out [B][T][K] dense, target strided like `actual` above, padding_mask u8 [B][T] or NULL.  The
 * workgroups' sums are added in workgroup order by the last workgroup to arrive (fixed summation order).
 * workspace: asac_masked_mse_workspace(N) floats, ZERO before its first use (the launch leaves its arrival counter at
 * zero again); N <= ASAC_MASKED_MSE_MAX, larger problems return ASAC_ERR_BAD_ARG and keep ATen's chain. */
#define ASAC_MASKED_MSE_MAX (1 << 20)
int64_t asac_masked_mse_workspace(int64_t n);
int asac_masked_mse(const float* pred, const float* target, int64_t target_stride_b, int64_t target_stride_t,
                    const uint8_t* padding_mask, int64_t mask_stride_b, int B, int T, int K, float* grad_out,
                    float* loss_out, float* workspace, void* stream);

/* The plain mean squared error over millions of elements and its gradient, one launch, one read of both operands (a
 * plugin's observation loss `mse_loss(decoded frames, frames)` under sac_base.py:1817 / 1798-1839:
 * *loss_out = sum (pred - target)^2 / N, grad_out = grad_scale * (pred - target) * 2 / N, N = B T K; grad_scale: the factor
 * the caller is going to multiply the loss with — 1 / n_step in `_train_rpm` — so that the gradient needs no second pass
 * over its 44 MB).  pred / grad_out [B][T][K]
 * dense, target strided (floats) like asac_masked_mse's; K and the strides multiples of 4, 16-byte aligned pointers;
 * N < 2^33.  Fixed summation order.  workspace: f32[asac_mse_mean_grad_workspace()], ZERO before its first use (left
 * zero by every launch). */
int64_t asac_mse_mean_grad_workspace(void);
int asac_mse_mean_grad(const float* pred, const float* target, int64_t target_stride_b, int64_t target_stride_t, int B, int T,
                       int K, float grad_scale, float* grad_out, float* loss_out, float* workspace, void* stream);

/* Loss of the recurrent prediction model's transition head and its gradient (SAC_Base._train_rpm,
 * sac_base.py:1798-1816; torch/distributions/normal.py log_prob / entropy, kl.py _kl_normal_normal):
 *   out[0] = -mean(log N(target; loc, scale)) + kl_weight * mean(KL(N(loc, scale) || N(0, 1)))
 *   out[1] = mean entropy of N(loc, scale)
 *   grad_loc / grad_scale [B][T][K] dense = d out[0] / d loc, / d scale
 * loc / scale / target [B][T][K] views with strides in floats (the two halves of the model's output, a slice of the
 * target representation's window).  One launch for ~60 elementwise ATen launches forward and backward; sums in
 * workgroup order (last to arrive); workspace asac_normal_nll_kl_workspace(N) floats, zero before first use. */
int64_t asac_normal_nll_kl_workspace(int64_t n);
int asac_normal_nll_kl(const float* loc, int64_t loc_stride_b, int64_t loc_stride_t, const float* scale,
                       int64_t scale_stride_b, int64_t scale_stride_t, const float* target, int64_t target_stride_b,
                       int64_t target_stride_t, int B, int T, int K, float kl_weight, float* grad_loc, float* grad_scale,
                       float* loss_entropy_out, float* workspace, void* stream);

/* The same with the head's activation inside: mean_logstd [B][T][2K] (strides in floats) is the transition model's raw
 * output (mean | logstd, `torch.chunk` of nn_models/predictions.py:39-44), the distribution's scale is
 * clamp(exp(logstd), scale_min, scale_max); grad_mean_logstd [B][T][2K] dense = d out[0] / d (mean | logstd) — exp's and
 * clamp's backward (pass-through where scale_min <= exp <= scale_max) applied in the launch. */
int asac_normal_nll_kl_logstd(const float* mean_logstd, int64_t stride_b, int64_t stride_t, float scale_min, float scale_max,
                              const float* target, int64_t target_stride_b, int64_t target_stride_t, int B, int T, int K,
                              float kl_weight, float* grad_mean_logstd, float* loss_entropy_out, float* workspace,
                              void* stream);

/* Products over the ROWS of two tall matrices (csrc/xty.hip, f32 MFMA):  out [M][N] = sum_r x[r][m] y[r][n]  and, optionally,
 * colsum_x [M] = sum_r x[r][m] — the weight and bias gradients of a Linear over thousands of rows (x = grad_out, y = input:
 * `nn.Linear` backward inside `LinearLayers` / `ResBlock`, nn_models/layers/linear_layers.py:24-119; the attention
 * projections, seq_layers.py:239-333) and the input / recurrent weight gradients of the GRU (seq_layers.py:14-114) at the widths
 * of the reference's environments (M <= 512, N <= 128).  Row strides in floats; accumulate != 0 adds to out / colsum_x;
 * fixed summation order (per-workgroup partials in `workspace`, asac_xty_workspace floats, summed in workgroup order). */
int asac_xty_supported(int64_t rows, int M, int N);
int64_t asac_xty_workspace(int64_t rows, int M, int N);
int asac_xty(const float* x, int64_t x_row_stride, int M, const float* y, int64_t y_row_stride, int N, int64_t rows, float* out,
             float* colsum_x, int accumulate, float* workspace, void* stream);
/* n_jobs <= 4 such products as ONE launch pair (the parameter gradients of the Linear layers of one attention block: q / k / v
 * projections and the output block).  Arrays of n_jobs entries (HOST arrays, read at the call); colsums NULL or with NULL
 * entries: no column sums; no output may appear twice; workspace: asac_xty_multi_workspace floats.  Same values as n_jobs
 * calls of asac_xty (same partition of the rows, same summation order). */
int64_t asac_xty_multi_workspace(int n_jobs, const int64_t* rows, const int* M, const int* N);
int asac_xty_multi(int n_jobs, const float* const* x, const int64_t* x_row_strides, const int* M, const float* const* y,
                   const int64_t* y_row_strides, const int* N, const int64_t* rows, float* const* outs, float* const* colsums,
                   int accumulate, float* workspace, void* stream);

/* State head of a representation plugin: y = tanh(x W^T + b) over the N = batch * window rows of an encoder
 * output, one launch per pass (the reference's test plugins end their representations with
 * `nn.Sequential(nn.Linear(n, 8), nn.Tanh())`: tests/nn_conv_vanilla.py:11-14, tests/nn_conv_attn.py:15-17,
 * envs/test/nn_rnn.py; stock torch runs a GEMM and a tanh forward, two GEMMs, a bias reduction, a tanh backward and
 * two gradient accumulations backward).  x [N][K] with row stride x_row_stride >= K floats, weight [O][K], bias [O]
 * (torch.nn.Linear layout), y [N][O].  K <= ASAC_LINEAR_TANH_MAX_IN, O <= ASAC_LINEAR_TANH_MAX_OUT.
 * Backward: grad_x [N][K] (may be NULL), grad_params = weight gradient [O][K] | bias gradient [O], overwritten or
 * (accumulate != 0) added to; per-workgroup partial sums are combined in workgroup order by the last workgroup
 * to finish (deterministic).  workspace: asac_linear_tanh_workspace(N, K, O) floats, ZERO before its first use
 * (the launch leaves its arrival counter at zero again); one workspace per concurrently running launch. */
#define ASAC_LINEAR_TANH_MAX_IN 64
#define ASAC_LINEAR_TANH_MAX_OUT 16
int64_t asac_linear_tanh_workspace(int64_t N, int K, int O);
int asac_linear_tanh_forward(const float* x, int64_t x_row_stride, const float* weight, const float* bias, int64_t N,
                             int K, int O, float* y, void* stream);
int asac_linear_tanh_backward(const float* x, int64_t x_row_stride, const float* weight, const float* y,
                              const float* grad_y, int64_t N, int K, int O, float* grad_x, float* grad_params,
                              int accumulate, float* workspace, void* stream);
/* The same head over an input that is the concatenation of two row blocks, x0 [N][K0] | x1 [N][K1] (x1 NULL: x0
 * alone), read in place — `self.dense(torch.cat([vec, self.conv(img)], dim=-1))`, tests/nn_conv_vanilla.py:16-19 —
 * with K = K0 + K1 <= ASAC_LINEAR_TANH_MAX_IN (weight [O][K]).  Backward: grad_x0 [N][K0], grad_x1 [N][K1] (each may be
 * NULL); the output gradient is given as grad_y [grad_members][N / grad_window][O]: row r receives
 * sum_e grad_y[e][r / grad_window] (member order) when r % grad_window == grad_position and zero otherwise — the
 * critics' state gradients at the one window position the Q loss reads (sac_base.py:2104-2110), summed in the launch
 * (grad_members = 1, grad_window = 1: a dense [N][O] gradient).  Values of asac_linear_tanh_forward / _backward on
 * the materialised concatenation / gradient, bit for bit. */
int asac_linear_tanh_forward2(const float* x0, int64_t x0_row_stride, int K0, const float* x1, int64_t x1_row_stride,
                              int K1, const float* weight, const float* bias, int64_t N, int O, float* y, void* stream);
/* ... forward / backward (`2w`) with x0 a [samples][x0_window_T][K0] slice of the sampled windows (`vec[:, burn_in_step:]`) read in place:
 * row r lies at (r / x0_window_T) * x0_sample_stride + (r % x0_window_T) * x0_row_stride (x0_window_T = 0: uniform rows) */
int asac_linear_tanh_forward2w(const float* x0, int64_t x0_row_stride, int x0_window_T, int64_t x0_sample_stride, int K0,
                               const float* x1, int64_t x1_row_stride, int K1, const float* weight, const float* bias,
                               int64_t N, int O, float* y, void* stream);
int asac_linear_tanh_backward2w(const float* x0, int64_t x0_row_stride, int x0_window_T, int64_t x0_sample_stride, int K0,
                                const float* x1, int64_t x1_row_stride, int K1, const float* weight, const float* y,
                                const float* grad_y, int grad_members, int grad_window, int grad_position, int64_t N, int O,
                                float* grad_x0, float* grad_x1, float* grad_params, int accumulate, float* workspace,
                                void* stream);
int asac_linear_tanh_backward2(const float* x0, int64_t x0_row_stride, int K0, const float* x1, int64_t x1_row_stride,
                               int K1, const float* weight, const float* y, const float* grad_y, int grad_members,
                               int grad_window, int grad_position, int64_t N, int O, float* grad_x0, float* grad_x1,
                               float* grad_params, int accumulate, float* workspace, void* stream);

/* ---- a Linear layer with a WIDE input (csrc/wide.hip) ----------------------------------------------------------------
 * y = act(x W^T + b) for x [R][K] (rows x_row_stride floats apart), W [N][K] row-major, N in {32, 64, 128}, K a multiple of
 * 16 up to 16 384; act: 0 identity, 1 GELU (exact).  The first layer of the dense head behind a flattened convolution map:
 * `ConvLayers.dense = LinearLayers(h * w * out_c, out_dense_n, ...)` with h * w * out_c = 2 592 for 84 x 84 frames
 * (nn_models/layers/image_layers.py:188-227, linear_layers.py:24-119: `ResBlock` = Linear + GELU without a residual path when
 * the widths differ), under autograd the GEMM / GELU / bias-reduction launches of `nn.Linear`.
 *   forward          split-K partials in `workspace`, summed in order with the bias; `pre` (or NULL): the pre-activations
 *   backward_input   dpre_out [R][N] = grad_y * act'(pre), dx [R][K] = dpre W (dx NULL: dpre only)
 *   backward_params  dw [N][K] (+)= dpre^T x, db [N] (+)= column sums of dpre (db may be NULL); fixed-order partial sums
 * workspace: asac_rows_wide_workspace(R, K, N) floats (-1: unsupported shape), for either launch. */
int asac_rows_wide_supported(int64_t R, int K, int N);
int64_t asac_rows_wide_workspace(int64_t R, int K, int N);
int asac_rows_wide_forward(const float* x, int64_t x_row_stride, int64_t R, int K, const float* w, const float* b, int N,
                           int act, float* y, float* pre, float* workspace, void* stream);
int asac_rows_wide_backward_input(const float* grad_y, const float* pre, int act, int64_t R, int K, const float* w, int N,
                                  float* dpre_out, float* dx, int64_t dx_row_stride, void* stream);
int asac_rows_wide_backward_params(const float* dpre, const float* x, int64_t x_row_stride, int64_t R, int K, int N, float* dw,
                                   float* db, int accumulate, float* workspace, void* stream);

/* ---- several fixed-order partial sums as one launch (csrc/reduce.hip) ---------------------------------------------------
 * The second launch of a backward — the per-workgroup parameter-gradient partials added in workgroup order — for up to
 * ASAC_SUM_PARTIALS_MAX_JOBS backwards at once: the walks of the representation's graph, one per gated loss, of
 * `calculate_adaptive_weights` (reference sac_base.py:1607-1631, entered from `_train_rpm` 1798-1839) read none of their
 * parameter gradients before all walks are done.  A job: out[i] (+)= sum over slabs t of partial[t * slab_stride + i], i < n,
 * in the order of the launch it replaces — slices = 16: sixteen contiguous slices of the slabs, each summed in order, then
 * the slice sums in order (asac_mlp_backward with >= 64 tiles, asac_attention_proj_backward); slices = 1: slab order
 * (asac_mlp_backward with fewer tiles).  The partials are what those entry points leave in their workspace under
 * ASAC_MLP_REDUCE_DEFER / accumulate == ASAC_ATTN_SUM_DEFER:
 *   asac_mlp_backward             [tiles = asac_mlp_backward_tiles(N, E)][E][member_stride], n = asac_mlp_param_extent(desc)
 *   asac_attention_proj_backward  [blocks][n], n = (3 | 4) (E E + E), blocks = asac_attention_proj_workspace / (4 (E E + E))
 *   asac_linear_tanh_backward*    (accumulate == ASAC_LINEAR_SUM_DEFER)  [blocks = ceil(N / 64)][n = O K + O]; 16 slices
 *   asac_conv2_backward(_windows / _multi)  (accumulate == ASAC_CONV_SUM_DEFER)  [blocks][n], n = n_cot x the packed parameter
 *                                 count, blocks = asac_conv2_backward_slabs(desc, N, n_cot); 16 slices
 * jobs_host: a HOST array. */
#define ASAC_SUM_PARTIALS_MAX_JOBS 16
#define ASAC_ATTN_SUM_DEFER 2
#define ASAC_CONV_SUM_DEFER 2
#define ASAC_LINEAR_SUM_DEFER 2
typedef struct {
    const float* partial;
    float* out;
    int64_t slab_stride;
    int64_t n;
    int32_t slabs, slices, accumulate, pad_;
} asac_partial_sum_t;
int asac_sum_partials_multi(int n_jobs, const asac_partial_sum_t* jobs_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASAC_HIP_H */
