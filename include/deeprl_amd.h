/* deeprl_amd.h -- C ABI of libdeeprl_amd.so: the MI355X (gfx950) replacement for the
 * rollout -> replay -> update hot path of ShangtongZhang/DeepRL.
 *
 * The reference has no FFI of its own (it is pure Python over ATen); the boundary below is what a
 * binding for this path would call, one group per reference component (file:line into the
 * reference tree).  Conventions:
 *   - plain `extern "C"`, raw pointers + sizes, no torch / C++ types;
 *   - every pointer named *_dev or documented "device" is a DEVICE pointer (HBM), outputs are
 *     CALLER-allocated; handles own only their own HBM;
 *   - the last argument is a hipStream_t (as void*); every call is asynchronous on it, never
 *     synchronises the device, allocates nothing (except *_create) and is hipGraph-capturable;
 *   - return 0 on success, a positive hipError_t or a negative errno-style code otherwise;
 *     nothing throws or aborts across the boundary;
 *   - host RNG (numpy legacy RandomState / python `random`) stays with the caller so that index
 *     streams are the reference's own; only indices / uniforms cross the boundary.
 */
#ifndef DEEPRL_AMD_H
#define DEEPRL_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DRA_ACT_NONE 0
#define DRA_ACT_RELU 1
#define DRA_ACT_TANH 2
#define DRA_MAX_Z 4 /* independent (input, weights) sets per batched forward launch */

/* ---- replay ring: deep_rl/component/replay.py:57-149 (UniformReplay storage + construct_transition) */
typedef struct dra_ring dra_ring;
/* replay.py:60-67.  frame_bytes = bytes of ONE stored observation (84*84 for Atari), action_bytes = bytes of one
 * action record (8 for int64). */
int dra_ring_create(dra_ring** out, int64_t capacity, int64_t frame_bytes, int64_t action_bytes, int history,
                    int n_step, double discount);
int dra_ring_shape(dra_ring* ring, int* history, int* n_step);   /* the history_length / n_step it was created with */
int dra_ring_discount(dra_ring* ring, double* discount);           /* ... and the discount of its n-step fold */
int dra_ring_destroy(dra_ring* ring);
int dra_ring_pointers(dra_ring* ring, void** frames, void** actions, void** rewards, void** masks);
/* replay.py:75-90 (feed): write `count` consecutive slots from DEVICE-ACCESSIBLE sources (device or pinned host).
 * NULL action/reward/mask sources take the by-value scalars (count == 1 use). */
int dra_ring_put(dra_ring* ring, int64_t slot0, int64_t count, const void* frame_src, const void* action_src,
                 int64_t action_val, const double* reward_src, double reward_val, const int32_t* mask_src,
                 int32_t mask_val, void* stream);
/* replay.py:75-90 from pageable HOST memory (staged through the handle's pinned buffer). */
int dra_ring_put_host(dra_ring* ring, int64_t slot, const void* frame_host, const void* action_host, double reward,
                      int32_t mask, void* stream);
/* synthetic transitions: frame k = splitmix64(seed, counter) bytes, action/reward/mask hashed. */
int dra_ring_fill_synthetic(dra_ring* ring, int64_t slot0, int64_t count, int64_t counter0, uint64_t seed,
                            int n_actions, int done_period, void* stream);
/* replay.py:112-140 (construct_transition) for a batch of validated indices idx_dev[batch] (int64, device):
 * out_state / out_next_state [batch][history][frame_bytes], out_action [batch][action_bytes], out_reward f64[batch]
 * (n-step return, fp64, reference association), out_mask i32[batch]; optional f32 copies of reward / mask
 * (what torch_utils.py:20-25 `tensor()` would produce).  Any output may be NULL. */
int dra_ring_gather(dra_ring* ring, const int64_t* idx_dev, int batch, void* out_state, void* out_next_state,
                    void* out_action, double* out_reward, int32_t* out_mask, float* out_reward_f32,
                    float* out_mask_f32, void* stream);
/* the same transition batch with state / next_state as two VIEWS of one block: out_block [batch][history + n_step][frame_bytes]
 * holds every frame of a sample's run once -- state = frames [0, history), next_state = frames [n_step, history + n_step) (the
 * reference stacks the history - n_step shared frames twice, replay.py:126-133; 1.13 instead of 1.81 MB written per DQN minibatch) */
int dra_ring_gather_block(dra_ring* ring, const int64_t* idx_dev, int batch, void* out_block, void* out_action, double* out_reward,
                          int32_t* out_mask, float* out_reward_f32, float* out_mask_f32, void* stream);
/* deep_rl/utils/normalizer.py:58-66 + torch_utils.py:23: out[i] = lut[in[i]], lut = f32(f64(v) * coef). */
int dra_u8_to_f32_lut(const void* in_u8, float* out, int64_t n, const float* lut256_dev, void* stream);
/* ... for n_rows rows of row_elems bytes that are in_row_stride bytes apart (such views); out dense; row_elems % 16 == 0 */
int dra_u8_to_f32_lut_rows(const void* in_u8, float* out, int64_t n_rows, int64_t row_elems, int64_t in_row_stride,
                           const float* lut256_dev, void* stream);

/* ---- sum tree: deep_rl/utils/sum_tree.py:6-66 as driven by replay.py:152-196 (PrioritizedReplay) */
typedef struct dra_sumtree dra_sumtree;
int dra_sumtree_create(dra_sumtree** out, int64_t capacity);            /* sum_tree.py:8-13 */
int dra_sumtree_destroy(dra_sumtree* tree);
int dra_sumtree_pointer(dra_sumtree* tree, void** tree_dev, int64_t* n_nodes);
/* sum_tree.py:54-60 + 16-20 for n (<= 1024) UNIQUE leaves (tree indices, int64 device) and fp64 priorities.
 * ordered != 0 replays the reference's incremental `+= change` walk update by update (slow, bit-exact always);
 * otherwise affected ancestors are recomputed level by level (bit-identical for fp32-valued priorities). */
int dra_sumtree_update(dra_sumtree* tree, const int64_t* leaf_idx_dev, const double* prio_dev, int n, int ordered,
                       void* stream);
int dra_sumtree_set(dra_sumtree* tree, int64_t leaf_idx, double prio, void* stream); /* sum_tree.py:39-51 (add) */
int dra_sumtree_set_from(dra_sumtree* tree, int64_t leaf_idx, const double* prio_dev, void* stream); /* same, *prio_dev */
/* n (<= 64) consecutive adds at write cursor write0, write0+1, ... (mod capacity), all at *prio_dev */
int dra_sumtree_set_many_from(dra_sumtree* tree, int64_t write0, int n, const double* prio_dev, void* stream);
/* replay.py:193-196 with the new priorities still on the device: leaf_idx_dev[i] <- f64(prio_f32_dev[pos_dev[i]]), i < n
 * (the host picks the pending, first-occurrence entries); stat_dev = {max_priority, smallest priority offered} is kept
 * over all `batch` offered values.  Falls back by itself to the reference's ordered walk when the level-parallel update
 * would not be exact in fp64 (capacity * max / ulp_f32(min) > 2^53), or when force_ordered != 0. */
int dra_sumtree_commit_f32(dra_sumtree* tree, const int64_t* leaf_idx_dev, const int32_t* pos_dev, int n,
                           const float* prio_f32_dev, int batch, double* stat_dev, int force_ordered, void* stream);
#define DRA_PER_CHAIN_MAX 1024
/* The whole of PrioritizedReplay.sample() on the device (round 3), so that the HOST is not between an update's priorities and
 * the next update: ONE kernel body = write-back of the update that just produced its loss vector (as dra_sumtree_commit_f32,
 * incl. its ordered-walk fallback) -> add_n adds at the write cursor at max_priority (as dra_sumtree_set_many_from) ->
 * stratified descent of the NEXT draw (as dra_sumtree_sample).  Every per-step input is read from one PINNED HOST block, so
 * the launch has constant arguments and is captured into the learner's update graph.  The kernel
 *   - gates the commit itself (sum_tree.py:54-60: first occurrence of a leaf in the minibatch wins; every sampled leaf is
 *     pending by construction, the sampled leaves are kept in `dev` from one launch to the next),
 *   - draws its uniforms from python's `random` stream: `rng_words` is a pinned ring of DRA_PER_RNG_WORDS raw Mersenne-Twister
 *     outputs the host generates ahead (random.getrandbits), `dev` keeps the cursor; random.random() = two words
 *     ((w0 >> 5) * 2^26 + (w1 >> 6)) * 2^-53, random.uniform(a, b) = a + (b - a) * random(),
 *   - applies replay.py:122-127's valid_index to every draw, drops the invalid ones and pads with random.choice of what
 *     has been picked so far (replay.py:176-186; _randbelow: k = n.bit_length(), words >> (32 - k) until one is < n),
 *   - writes the minibatch's ring indices to `idx_out_dev` (int64[next_batch], what the next update's kernels read) and
 *     f32(p / total) + the importance exponent to `samp_prob_dev` (f32[next_batch + 1]),
 *   - computes the priorities itself from the update's per-sample loss vector ((|loss| + eps)^alpha, DQN_agent.py:121-123;
 *     also left in prio_out_dev) and the NEXT update's importance weights ((P * B + 1e-6)^-beta / max, :124-126) into
 *     weights_out_dev: the update needs no batch-wide reduction of its own any more,
 *   - and leaves everything the host's (lagging) bookkeeping wants in the pinned block, out_seq last. */
#define DRA_PER_RNG_WORDS 65536
typedef struct dra_per_chain2_io {
  int32_t add_n, batch, next_batch, force_ordered;      /* inputs (batch = next_batch = the launch's `batch` argument) */
  int32_t history, n_step;
  int64_t add_write0;                  /* tree write cursor before the adds */
  int64_t memory_size;                 /* ring slots = leaves */
  int64_t pos_after, size_after;       /* replay.pos / size() after those adds (valid_index of the next draw) */
  uint64_t rng_produced;               /* words the host has generated so far: the cursor must not pass it */
  float beta_next;                     /* importance exponent of the next update (DQN_agent.py:124) */
  int32_t reserved;
  int64_t out_raw_idx[DRA_PER_CHAIN_MAX];   /* outputs: the leaf of every descent, in segment order (all become pending) */
  int64_t out_idx[DRA_PER_CHAIN_MAX];       /* the minibatch's leaves after the validity filter and the padding */
  double out_p[DRA_PER_CHAIN_MAX];          /* their priorities */
  double out_total;
  int32_t out_n_valid;
  int32_t out_flags;                   /* 1: the word ring ran dry, 2: no valid draw at all -- the results are invalid */
  uint64_t out_rng_cursor;             /* words consumed so far */
  uint64_t out_seq;                    /* launches completed, written LAST with system scope */
} dra_per_chain2_io;
/* dev: device state of *dra_sumtree_per_chain2_state_bytes bytes ({cursor, seq, leaves of the current minibatch}). */
int dra_sumtree_per_chain2_state_bytes(int64_t* bytes);
int dra_sumtree_per_chain2_state_set(void* dev_state, uint64_t rng_cursor, uint64_t seq, const int64_t* tree_idx_host, int n);
int dra_sumtree_per_chain2(dra_sumtree* tree, dra_per_chain2_io* io_pinned, const float* loss_vec_dev, float replay_eps,
                           float replay_alpha, float* prio_out_dev, double* stat_dev, void* dev_state,
                           const uint32_t* rng_words_pinned, int64_t* idx_out_dev, float* samp_prob_dev,
                           float* weights_out_dev, int batch, void* stream);
/* replay.py:168-175 + sum_tree.py:23-33,63-66: u_dev[batch] are raw python random.random() draws; lane i samples
 * s = a + (b-a)*u_i on segment i of total/batch and descends; outputs tree index, leaf priority, and the total. */
int dra_sumtree_sample(dra_sumtree* tree, const double* u_dev, int batch, int64_t* out_tree_idx, double* out_p,
                       double* out_total, void* stream);
int dra_sumtree_rebuild(dra_sumtree* tree, void* stream);

/* ---- fused losses (forward + backward) */
/* deep_rl/agent/DQN_agent.py:78-99 (+ PER :120-127).  q, q_next_* [batch][n_actions] f32; action int64[batch] or
 * f32[batch]; reward/mask f32[batch].  sampling_prob NULL = uniform replay.  out_dq = d(reduced loss)/dq.
 * beta < 0 (here and in dra_per_weights): the importance exponent is read from sampling_prob[batch] on the device, so that
 * the launch has no per-update argument and can be replayed from a captured graph. */
int dra_td_loss(const float* q, const float* q_next_target, const float* q_next_online, const void* action,
                int action_is_i64, const float* reward, const float* mask, int batch, int n_actions, float gamma_n,
                const float* sampling_prob, float beta, float replay_eps, float replay_alpha, float* out_loss,
                float* out_dq, float* out_delta, float* out_prio, float* out_weights, void* stream);
/* deep_rl/agent/CategoricalDQN_agent.py:60-89 on LOGITS [batch][n_actions][n_atoms] (softmax folded in). */
int dra_c51_loss(const float* logits, const float* logits_next_target, const float* logits_next_online,
                 const void* action, int action_is_i64, const float* reward, const float* mask, int batch,
                 int n_actions, int n_atoms, float gamma_n, float v_min, float v_max, const float* atoms,
                 float* out_kl, float* out_dlogits, const float* weights, void* stream);
/* deep_rl/agent/QuantileRegressionDQN_agent.py:55-77 + utils/torch_utils.py:47-48; workspace f32[batch*n_q]. */
int dra_qr_loss(const float* theta, const float* theta_next_target, const void* action, int action_is_i64,
                const float* reward, const float* mask, int batch, int n_actions, int n_quantiles, float gamma_n,
                float* workspace, float* out_loss_vec, float* out_loss, float* out_dtheta, void* stream);
/* DQN_agent.py:121-127 applied to any loss vector. */
int dra_per_weights(const float* loss_vec, const float* sampling_prob, int batch, float beta, float replay_eps,
                    float replay_alpha, float* out_prio, float* out_weights, void* stream);
int dra_weighted_mean(const float* x, const float* w, int n, float* out, void* stream);
/* deep_rl/network/network_heads.py:240-255 (CategoricalActorCriticNet: Categorical(logits) -> sample / log_prob / entropy).
 * action_in NULL: actions are sampled by inverse CDF from uniform[batch] into action_out.  n_actions <= 64. */
int dra_categorical_fwd(const float* logits, int batch, int n_actions, const int64_t* action_in, const float* uniform,
                        int64_t* action_out, float* log_pi_a, float* entropy, void* stream);
int dra_categorical_bwd(const float* logits, int batch, int n_actions, const int64_t* action, const float* g_log_pi_a,
                        const float* g_entropy, float* out_dlogits, void* stream);
/* rank-invariant Categorical(logits).sample() for the data-parallel on-policy agents: action[i] = argmax_a
 * (logits[i][a] + Gumbel noise hashed from (seed, *step_dev, lo + i, a)), lo = first GLOBAL environment of this rank; the
 * kernel advances *step_dev (device int64), so the launch replays from a captured rollout graph.  n_actions <= 64. */
int dra_gumbel_sample(const float* logits, int n_local, int n_actions, uint64_t seed, int64_t* step_dev, int64_t lo,
                      int64_t* action_out, void* stream);
/* deep_rl/agent/PPO_agent.py:77-86: out3 = {policy_loss, value_loss, approx_kl}. */
int dra_ppo_loss(const float* log_pi_a, const float* entropy, const float* v, const float* old_log_pi_a,
                 const float* adv, const float* ret, int m, float ratio_clip, float entropy_weight, float* out3,
                 float* g_log_pi_a, float* g_entropy, float* g_v, void* stream);
/* deep_rl/agent/A2C_agent.py:55-62: out4 = {total, policy, value, entropy}. */
int dra_a2c_loss(const float* log_pi_a, const float* entropy, const float* v, const float* adv, const float* ret,
                 int m, float entropy_weight, float value_loss_weight, float* out4, float* g_log_pi_a,
                 float* g_entropy, float* g_v, void* stream);

/* ---- return / advantage recurrences: PPO_agent.py:51-61, A2C_agent.py:43-53, NStepDQN_agent.py:56-60 */
/* reward, mask [t_len][n_env]; value [t_len+1][n_env]; outputs [t_len][n_env]. */
int dra_gae(const float* reward, const float* mask, const float* value, int t_len, int n_env, float gamma, float tau,
            int use_gae, float* out_adv, float* out_ret, void* stream);
int dra_adv_normalize(float* adv, int64_t n, void* stream); /* PPO_agent.py:66 */

/* ---- network contractions: deep_rl/network/network_bodies.py:10-33,50-73 + network_heads.py heads */
/* layer 1..3 of NatureConvBody; nz batched (input, weights) sets per launch. */
int dra_conv_fwd(int layer, int nz, const void* const* x, const float* const* w, const float* const* bias,
                 float* const* y, int batch, int x_is_u8, double u8_coef, int act, void* stream);
int dra_conv_bwd_w(int layer, const float* dy, const void* x, float* dw, float* db, int64_t slab_stride, int ksplit,
                   int batch, int x_is_u8, double u8_coef, void* stream);
int dra_conv_bwd_x(int layer, const float* dy, const float* w, const float* xact, float* dx, int batch, int act,
                   void* stream);
/* KOC weight layout ([K=(c,kh,kw)][OC]; conv_v2.hip): one-round-trip forward, and the matching gradients. */
int dra_conv_fwd_koc(int layer, int nz, const void* const* x, const float* const* wt, const float* const* bias,
                     float* const* y, int batch, int x_is_u8, double u8_coef, int act, void* stream);
/* conv1 of a batch-1 forward whose 4-frame uint8 stack is read straight from the ring (newest slot on device);
 * stack_age_dev (optional, device int32): channel c reads slot newest - min(3 - c, *stack_age_dev) (episode start:
 * the first frame repeated, dra_dqn_step_params.stack_age); null = the last 4 ring frames */
int dra_conv1_fwd_koc_ring(const void* frames, const int64_t* newest_slot_dev, const int32_t* stack_age_dev, int64_t capacity, const float* wt,
                           const float* bias, float* y, double u8_coef, int act, void* stream);
/* same, the slot taken from entry (*seq_dev mod n_entries) of an array of parameter blocks stride_bytes apart;
 * newest_frame (optional, device uint8[84*84]): the newest channel comes from this not-yet-committed observation */
int dra_conv1_fwd_koc_ring_seq(const void* frames, const int64_t* slot_field_dev, const int32_t* stack_age_field_dev, const unsigned* seq_dev, int n_entries,
                               int64_t stride_bytes, int64_t capacity, const void* newest_frame, const float* wt,
                               const float* bias, float* y, double u8_coef, int act, void* stream);
int dra_conv_bwd_w_koc(int layer, const float* dy, const void* x, float* dw, float* db, int64_t slab_stride, int ksplit,
                       int batch, int x_is_u8, double u8_coef, void* stream);
int dra_conv_bwd_x_koc(int layer, const float* dy, const float* wt, const float* xact, float* dx, int batch, int act,
                       void* stream);
int dra_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);
int dra_act_bwd(const float* dy, const float* y, float* dpre, int64_t n, int act, void* stream);
int dra_linear_fwd(int nz, const float* const* x, const float* const* w, const float* const* bias, float* const* y,
                   int batch, int in_features, int out_features, int act, float* workspace, int64_t workspace_floats,
                   void* stream);
/* two heads of different width on the same features in one launch (network_heads.py:241-243: fc_action / fc_critic of the
 * actor-critic nets on phi), one wave per input row; in_features <= 512; same per-output arithmetic as dra_linear_fwd */
int dra_linear_fwd_pair(const float* x, const float* w0, const float* b0, float* y0, int out0, const float* w1, const float* b1,
                        float* y1, int out1, int batch, int in_features, int act, void* stream);
/* a rollout step's policy head (network_heads.py:240-255 under no_grad, action = None) in one launch: logits = x W0^T + b0
 * [batch, n_actions <= 64], v = x w1^T + b1 [batch], then Categorical(logits): action by inverse CDF from uniform[b], log_pi_a,
 * entropy -- dra_linear_fwd_pair + dra_categorical_fwd, bit for bit; out_logits may be NULL */
int dra_policy_heads_sample(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                            const float* uniform, int batch, int in_features, int n_actions, int64_t* out_action,
                            float* out_log_pi_a, float* out_entropy, float* out_v, float* out_logits, void* stream);
/* the same head for GIVEN actions (the update's forward): log_pi_a / entropy of action[b], v, and the logits [batch, n_actions] */
int dra_policy_heads_given(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                           const int64_t* action, int batch, int in_features, int n_actions, float* out_log_pi_a,
                           float* out_entropy, float* out_v, float* out_logits, void* stream);
/* ... with the finish of the 512-feature layer below in front: features = relu(fold_bias + sum of the n_slabs (8 or 14) K-slice
 * partial sums slabs [n_slabs][batch][512] of dra_linear_fwd_slabs_one, slab 0 first), written to out_phi [batch][512] */
int dra_policy_heads_given_fold(const float* slabs, int n_slabs, const float* fold_bias, const float* w0, const float* b0,
                                const float* w1, const float* b1, const int64_t* action, int batch, int n_actions,
                                float* out_log_pi_a, float* out_entropy, float* out_v, float* out_logits, float* out_phi, void* stream);
/* a rollout step's head with the finish of fc4's 28-slice one-pass forward in front: features = relu(fold_bias + sum of slabs
 * [28][batch][512], slab 0 first); one workgroup per row; then as dra_policy_heads_sample */
int dra_policy_heads_sample_fold28(const float* slabs, const float* fold_bias, const float* w0, const float* b0, const float* w1,
                                   const float* b1, const float* uniform, int batch, int n_actions, int64_t* out_action,
                                   float* out_log_pi_a, float* out_entropy, float* out_v, void* stream);
/* its backward in one launch (dra_categorical_bwd + dra_linear_bwd_pair [+ dra_act_bwd], same sums in the same order): from the
 * gradients of log_pi_a / entropy / v [batch] (any may be NULL = zero) -> dx [batch, in_features] (optional; relu_mask != 0:
 * times [x > 0], x being a fused-ReLU output), dW0 [n_actions, in_features], db0, dW1 [1, in_features], db1; batch <= 8192 */
int dra_policy_heads_bwd(const float* logits, const int64_t* action, const float* g_log_pi_a, const float* g_entropy,
                         const float* g_v, const float* x, const float* w0, const float* w1, float* dx, float* dw0, float* db0,
                         float* dw1, float* db1, int batch, int in_features, int n_actions, int relu_mask, void* stream);
/* backward of that pair in one launch: dx [B, K] (optional) = g0 W0 + g1 W1, dW_h = g_h^T x, db_h = column sums of g_h */
int dra_linear_bwd_pair(const float* g0, const float* g1, const float* x, const float* w0, const float* w1, float* dx, float* dw0,
                        float* db0, float* dw1, float* db1, int batch, int in_features, int out0, int out1, void* stream);
/* both gradients of a 512-output linear layer (fc4 of NatureConvBody) in one launch: dx [batch, in_features] = dy W (times
 * [x > 0] when x is a fused-ReLU output), dW [512, in_features] = dy^T x, db [512] (optional) = column sums of dy; in_features >= 1024 */
int dra_linear_bwd_xw_one512(const float* dy, const float* w, const float* x, int x_is_relu_output, float* dx, float* dw, float* db,
                             int batch, int in_features, void* stream);
/* raw split-K partial sums [nz][ksplit][batch][out] (no bias / activation): the consumer reduces them. */
int dra_linear_fwd_slabs(int nz, const float* const* x, const float* const* w, int batch, int in_features,
                         int out_features, int ksplit, float* slabs, void* stream);
int dra_linear_bwd_w(const float* dy, const float* x, float* dw, float* db, int batch, int in_features,
                     int out_features, void* stream);
int dra_linear_bwd_x(const float* dy, const float* w, const float* xact, float* dx, int batch, int in_features,
                     int out_features, int act, void* stream);

/* ---- Atari frame preprocessing (csrc/preproc.hip): deep_rl/component/envs.py:39-47 -> baselines' MaxAndSkipEnv (max of the
 * last two raw frames) + WarpFrame (cv2 RGB2GRAY, cv2.resize INTER_AREA to 84x84), restated from OpenCV's published
 * algorithms; parity unpinned by the reference (cv2 / baselines are not in the image).  dra_resize_area_tab builds one
 * axis' (source index, weight) table on the HOST (offs has dsize + 1 entries; returns the entry count); the kernel takes
 * device copies of the two tables.  raw: [n_env][2][height][width][3] uint8, out: [n_env][out_h][out_w] uint8. */
int dra_resize_area_tab(int ssize, int dsize, int* si, float* alpha, int* offs, int max_entries);
int dra_atari_preprocess(const uint8_t* raw, int n_env, int height, int width, int out_h, int out_w, const int* x_si,
                         const float* x_alpha, const int* x_off, const int* y_si, const float* y_alpha, const int* y_off,
                         uint8_t* out, void* stream);

/* ---- horizontally fused backward launches + one-pass contractions (csrc/fused.hip, csrc/oneshot.h): the autograd
 * backward behind DQN_agent.py:129 for VanillaNet(NatureConvBody).  `variant` / tuning bits: */
#define DRA_VAR_FUSED_BWD 1      /* learner: a layer's weight- and input-gradient kernels share one launch */
#define DRA_VAR_ONESHOT_DGRAD 2  /* one-pass input gradients (conv2, conv3, fc4) */
#define DRA_VAR_ONESHOT_FWD 4    /* one-pass fc4 forward partial sums */
#define DRA_VAR_ONESHOT_WGRAD 8  /* one-pass conv weight gradients, one slab per (sample, row chunk) */
#define DRA_VAR_PINNED_IDX 16    /* learner: the gather reads minibatch indices from pinned host memory */
#define DRA_VAR_ACTOR_V2 32      /* learner: 5-kernel actor step (ring-direct conv1, GEMV fc4, head + env) */
#define DRA_VAR_ACTOR_PARAMS 64  /* learner: async actor reads a double-buffered parameter copy */
#define DRA_VAR_ACTOR_V3 512     /* learner: the actor graph's first kernel reads its parameter block from a pinned ring
                                    (no copy command in front of the graph) */
#define DRA_VAR_ACTOR_FUSED_HEAD 1024 /* with ACTOR_V3: fc4 + head + env step as one kernel (last-workgroup ticket) */
#define DRA_VAR_GATHER_IN_GRAPH 2048 /* learner, async: a call carries transitions AND indices of the same step; actor
                                    transitions + gather are one graph, the update issued is the previous step's */
#define DRA_VAR_ACTOR_RING 4096  /* learner, async pipelined: the actor's parameter blocks are pushed K steps ahead into a
                                    device ring (dra_dqn_learner_actor_ring_push); no per-step copy command, the last actor
                                    kernel of a step produces the next step's first frame */
#define DRA_VAR_ACTOR_FUSED_CONV1 8192 /* with ACTOR_RING: the head of env step e-1 and the environment step run in front of
                                          conv1 of step e in one launch (4 launches per env step instead of 5) */
#define DRA_VAR_GATHER_ON_UPDATE 16384 /* learner, async pipelined: the gather runs on the UPDATE stream (the actor chain is the
                                       * longer one); the rare step whose minibatch touches the ring slots the next actor
                                       * graph overwrites makes that graph wait for the update (decided on the host) */
#define DRA_VAR_RING_DIRECT 32768  /* learner, async pipelined (with GATHER_ON_UPDATE's host-decided waits): NO gather -- conv1
                                    * (forward of every net, weight gradient) reads the uint8 frames of the sampled
                                    * transitions straight from the replay ring and the head kernel their action / n-step
                                    * reward / mask: one launch, 1.8 MB of writes and 1.8 MB of re-reads less per update */
/* (bit 65536 was DRA_VAR_COOP_OPT, the one-launch fold + norm + optimiser behind a grid barrier: measured 14 % slower than the
 * two launches, removed in round 4; the bit is ignored) */
#define DRA_VAR_IDX_PREFETCH 131072 /* learner (with RING_DIRECT): a step-tagged copy of the minibatch indices goes to the device by
                                    * an unordered async copy when the step is enqueued; conv1 takes an element from there when
                                    * its tag is this update's (no PCIe read in front of its frame loads), else from pinned memory */
#define DRA_VAR_DGRAD_SCATTER 262144 /* dra_conv_bwd_fused, layers 2 / 3, batch >= 256 (with ONESHOT_DGRAD + ONESHOT_WGRAD): the
                                    * input gradient in scatter form -- the contraction runs over the OUTPUT positions (no padded
                                    * (pixel, tap) columns: conv3 1.96x -> 1.14x, conv2 1.58x -> 1.19x issued MFMAs), col2im
                                    * by in-order read-add-write into an LDS image of dX; deterministic (csrc/dgrad_scatter.h).  Below 256 samples
                                    * the bit is ignored.  (Until round 4 this bit was DRA_VAR_WGRAD_ACC, a removed experiment.)
                                    * network_bodies.py:10-33 */
#define DRA_VAR_LATE_FOLD 524288  /* learner (with ONESHOT_WGRAD + FUSED_BWD): no gradient-norm launch -- sums of squares come
                                   * from the kernels that write each gradient, conv3 / conv2 slabs are folded by spare
                                   * workgroups of the NEXT layer's backward launch, conv1's by the first workgroups of the
                                   * optimizer launch (dra_clip_step_late) */
#define DRA_VAR_ACTOR_MEGA 1048576 /* learner (with ACTOR_RING + ACTOR_FUSED_CONV1): an env step of the device actor -- head of the
                                   * previous step, environment step, conv1, conv2, conv3, fc4 -- is ONE launch whose workgroups
                                   * hand their outputs over through arrival counters (4 launches per agent step + the tail
                                   * kernel instead of 17): weights are prefetched before a layer's input exists */
#define DRA_VAR_MEASURE_DGRAD_ONLY 2097152 /* measurement aid (tools/conv_big_bwd.py): dra_conv_bwd_fused launches ONLY the
                                          * input-gradient role -- the weight-gradient slabs are NOT written */
#define DRA_VAR_MEASURE_WGRAD_ONLY 4194304 /* ... ONLY the weight-gradient role -- dx is NOT written */
#define DRA_VAR_DEFER_FC4 8388608 /* learner (RING_DIRECT + LATE_FOLD + ACTOR_MEGA, VanillaNet + RMSprop): the optimizer launch of the
                                   * pipelined graphs steps everything but fc4's weights (95 % of the parameters, 51 of its 54 MB);
                                   * that segment is stepped by rider workgroups in the NEXT update's conv1 / conv2 forward
                                   * launches -- nothing reads it before that graph's fc4 forward, the actor's copy of it is guarded
                                   * by a device word -- with the same arithmetic (same bits).  dra_dqn_learner_flush steps a
                                   * pending segment at once; every entry that reads parameters outside those graphs does so itself */
#define DRA_VAR_ACTOR_PERSIST 16777216 /* learner (with ACTOR_RING + ACTOR_FUSED_CONV1 + ACTOR_MEGA, VanillaNet): ALL env steps of an agent
                                    * step of the device actor as ONE launch of 32 co-resident workgroups -- activations cross
                                    * workgroups as 8-byte {value, tag} words (no launch boundary, no arrival counter), fc4's and the
                                    * convolutions' weights stay in registers / LDS for the whole agent step, conv1's workgroups keep
                                    * their rows of the frame stack in LDS.  Same arithmetic as the multi-launch env step: bit-identical
                                    * actions, action values and ring contents.  Needs >= 32 CUs on the actor's stream
                                    * (dra_dqn_learner_set_actor_cus), else the multi-launch form runs.  DQN_agent.py:24-45 */
#define DRA_VAR_FWD_CHAIN 33554432 /* learner (RING_DIRECT, VanillaNet, two nets, batch 17..32): conv1 + conv2 + conv3 of the update's
                                    * forward pass as ONE launch in dependency order -- a workgroup of layer L + 1 waits (arrival
                                    * counter per (net, sample), weights requested first) for the workgroups of ITS sample in layer L
                                    * instead of for the whole layer and a launch boundary.  Same arithmetic: bit-identical.  Carries
                                    * no riders: DRA_VAR_DEFER_FC4 is off with it.  DQN_agent.py:81-99, network_bodies.py:10-33 */
#define DRA_VAR_BWD_CHAIN 67108864 /* learner (as FWD_CHAIN, with LATE_FOLD): conv3's, conv2's and conv1's backward launches as ONE launch
                                    * in dependency order -- a workgroup of layer L - 1 waits for the input-gradient workgroups of ITS
                                    * sample in layer L, the slab folds for the weight-gradient workgroups of their layer.  Same
                                    * arithmetic: bit-identical gradients.  DQN_agent.py:129-134 */
#define DRA_VAR_FLAG_SYNC 134217728 /* learner (RING_DIRECT + ACTOR_RING + ACTOR_PERSIST + FWD_CHAIN, plain uniform replay): the steady-state
                                    * pipelined step records no event and waits for none.  The update graph's first launch counts
                                    * itself in a device word when it STARTS (= the previous update is complete and written back);
                                    * the actor launch polls that word for the number the host left for it in a pinned ring; the
                                    * host paces itself (three calls ahead) on a pinned count the actor launch publishes.  An event
                                    * record behind a graph replay costs 4.9 us of the update stream, the other stream's wait on it
                                    * 9.4 us more (tools/ubench/graph_gap.hip).  Every other entry point drains both streams first.
                                    * Same launches on the same data: bit-identical.  DQN_agent.py:101-138, BaseAgent.py:108-182 */
#define DRA_VAR_LANE_EAGER 268435456 /* learner (with FLAG_SYNC): in the event-free lane the update is issued as its six plain launches
                                    * instead of one graph replay (a replay costs 6 us before its first kernel, a plain dependent
                                    * launch 1.3 us; the host pays the launches instead).  Same launches: bit-identical */
#define DRA_VAR_TARGET_AHEAD 536870912 /* learner (with FLAG_SYNC + LANE_EAGER + BWD_CHAIN, an ahead stream set): target(next_states) of
                                    * update t + 1 (DQN_agent.py:85-88: conv1-3 + fc4 of the target net, which depend on the target
                                    * parameters and the replay ring only) is issued one call early on its own stream and runs UNDER
                                    * update t; the update's forward chain then carries the online net alone and its head kernel folds
                                    * the target's fc4 partial sums from a stash.  Needs the next minibatch's indices one call early
                                    * (dra_dqn_learner_stage_next_indices); an update without a stash computes its target in line.
                                    * Same kernels on the same data: bit-identical.  DQN_agent.py:114-127 */
#define DRA_VAR_BWD_CHAIN_FC 1073741824 /* learner (with BWD_CHAIN): fc4's and the head's backward (input gradient, fc4 weight gradient,
                                    * head weight gradient) lead the chained backward launch; conv3's roles wait for the
                                    * input-gradient workgroups instead of for a launch boundary.  Same arithmetic: bit-identical
                                    * gradients.  OPT-IN, measured SLOWER on MI355X (92.4 -> 97.5 us per step): a launch's register
                                    * and LDS footprint is the maximum over its roles -- fc4's input gradient (152 + 32 registers,
                                    * 65.7 KB of LDS) takes the whole launch from three workgroups per CU to two.
                                    * DQN_agent.py:129-134 */
#define DRA_VAR_HEAD_CHAIN 65536 /* learner (with FWD_CHAIN, VanillaNet head, uniform replay, batch <= 32): the head launch (fc4 fold, head,
                                    * TD error, dq, dh4) and fc4's + the head's backward launch as ONE launch in dependency order: the
                                    * backward roles request fc4's weights / conv3's activations first, then wait for the head role's
                                    * workgroups on one arrival counter.  Same arithmetic: bit-identical.  DQN_agent.py:85-99,131 */
#define DRA_VAR_CU_PARTITION 256 /* host: actor stream and update stream own disjoint CU sets (dra_stream_create_masked) */
#define DRA_VAR_PIPE_GATHER 128  /* learner, async: gather on the actor stream into a double-buffered minibatch,
                                    body + optimizer as one graph -- no cross-stream wait on either chain */
/* process-wide default variant mask used by learners created afterwards */
int dra_set_tuning(int mask);
int dra_get_tuning(int* mask);
int dra_conv_wgrad_slabs(int layer, int batch, int ksplit, int variant, int* n_slabs);
int dra_conv_bwd_fused(int layer, const float* dy, const void* x, const float* wt, const float* xact, float* dw,
                       float* db, int64_t slab_stride, int ksplit, float* dx, int batch, int x_is_u8, double u8_coef,
                       int act, int variant, void* stream);
int dra_fc_bwd_fused(const float* dq, const float* h4, const float* dh4, const float* x3, const float* w4, float* dwh,
                     float* dbh, float* dw4, float* db4, float* dx3, int batch, int n_actions, int in_features, int act,
                     int variant, void* stream);
int dra_linear_fwd_slabs_one(int nz, const float* const* x, const float* const* w, int batch, int in_features,
                             int out_features, int ksplit, float* slabs, void* stream);

/* ---- clip + optimiser: DQN_agent.py:130-134 with the optimisers of examples.py:67-68,139,204,370,508-509,534 */
int dra_norm_partials(void); /* doubles needed per dra_grad_sqnorm call */
int dra_grad_sqnorm(float* grad, int64_t n, const float* slabs, int n_slabs, int64_t slab_stride, double* partials,
                    void* stream);
/* segmented norm + fold (one launch): each segment of the flat gradient is the fixed-order sum of its own
 * n_slabs split-K slabs (slab s of element i at slabs[s*slab_stride + (i - begin)]); the rest of grad is read as is. */
#define DRA_MAX_FOLD_SEGS 4
typedef struct dra_fold_seg {
  int64_t begin, count;     /* floats, multiples of 4 */
  const float* slabs;       /* 16-byte aligned */
  int64_t slab_stride;      /* floats, multiple of 4 */
  int32_t n_slabs, reserved;
} dra_fold_seg;
int dra_norm_partials_max(void);
int dra_grad_sqnorm_segs(float* grad, int64_t n, const dra_fold_seg* segs, int n_segs, double* partials,
                         int* n_partials, void* stream);
int dra_grad_sqnorm_segs_blocks(int64_t n, const dra_fold_seg* segs, int n_segs, int* blocks); /* its workgroups = partials (host only) */
/* the late-fold form (DRA_VAR_LATE_FOLD): every tensor's sum of squares was left in partials[0, n_prior) by the kernels
 * that produced its gradient, except ONE segment (starting at element 0, n_slabs <= 256: conv1, whose weight gradient is the
 * last kernel of the backward) that is still in slabs.  The launch's first dra_clip_step_late_blocks() (<= 256) workgroups
 * fold it and publish their sums of squares into partials[n_prior ...], which must hold -1.0 at launch: the value IS the
 * arrival flag.  Every workgroup loads its operands and the earlier partials, waits for those slots to turn non-negative
 * (bounded: timeout_flag as above), reduces all partials in the fixed order and applies the step.  No gradient-norm launch,
 * no ticket counter. */
int dra_clip_step_late_blocks(const dra_fold_seg* seg, int* fold_blocks);
int dra_clip_step_late(float* param, float* grad, float* state1, float* state2, int64_t n, const dra_fold_seg* seg,
                       double* partials, int n_prior, int* timeout_flag, int optimizer, float max_norm, const float* hyper,
                       int centered, const int64_t* step_dev, float* out_norm, float* param_copy, void* stream);
int dra_rmsprop_step(float* param, const float* grad, float* square_avg, float* grad_avg, int64_t n,
                     const double* partials, int n_partials, float max_norm, float lr, float alpha, float eps,
                     int centered, float* out_norm, void* stream);
int dra_rmsprop_step_copy(float* param, const float* grad, float* square_avg, float* grad_avg, int64_t n,
                          const double* partials, int n_partials, float max_norm, float lr, float alpha, float eps,
                          int centered, float* out_norm, float* param_copy, void* stream);
int dra_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  const double* partials, int n_partials, float max_norm, float lr, float beta1, float beta2, float eps,
                  int64_t step, float* out_norm, void* stream);
/* graph-replayable Adam: the step-dependent scalars {lr/(1-b1^t), 1/sqrt(1-b2^t)} (dra_adam_hyper, host) are read
 * from device memory, every kernel argument is constant across steps */
int dra_adam_hyper(float lr, float beta1, float beta2, int64_t step, float* out2);
int dra_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const double* partials, int n_partials, float max_norm, float beta1, float beta2, float eps,
                      const float* hyper_dev, float* out_norm, void* stream);
/* Adam whose 1-based step count is read from DEVICE memory (a learner graph bumps it), with an optional mirror of the
 * updated parameters (the async actor's copy) */
int dra_adam_step_counter(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                          const double* partials, int n_partials, float max_norm, float lr, float beta1, float beta2,
                          float eps, const int64_t* step_dev, float* out_norm, float* param_copy, void* stream);
int dra_copy_f32(float* dst, const float* src, int64_t n, void* stream); /* DQN_agent.py:136-138 */
/* polyak averaging of the target network (DDPG_agent.py:26-30, TD3_agent.py:28-32) over the two networks' flat parameter
 * buffers (16-byte aligned): target[i] = target[i] * keep + src[i] * mix with keep = f32(1 - mix); each product is rounded
 * before the add, as the reference's `target_param * (1.0 - mix) + param * mix` is.  One launch. */
int dra_soft_update(float* target, const float* src, int64_t n, float keep, float mix, void* stream);

/* ---- minibatch rows of an on-policy rollout (PPO_agent.py:77-80: entries[batch_indices] over state / action / log_pi_a / ret /
 * advantage): dst[q][r] = src[q][idx[r]] for n_fields <= DRA_GATHER_MAX_FIELDS row-major device arrays of row_bytes[q] bytes per
 * row, one launch for all of them; idx_dev int64 [n_rows] (negative = from the end, as torch indexes), n_src_rows rows per source */
#define DRA_GATHER_MAX_FIELDS 8
int dra_gather_rows(int n_fields, const void* const* src, void* const* dst, const int64_t* row_bytes, const int64_t* idx_dev,
                    int n_rows, int64_t n_src_rows, void* stream);

/* ---- device-resident synthetic vector environment (on-policy agents; A2C_agent.py:26-34, PPO_agent.py:33-47): the uint8
 * [n_env][history][84*84] observations of one rollout step from per-environment frame counters / episode ages / stream
 * seeds (device arrays of n_env).  Frames are the counter-hash frames of dra_ring_fill_synthetic. */
int dra_synth_stacks(const int64_t* counter_dev, const int32_t* age_dev, const int64_t* seed_dev, int n_env, int history,
                     void* out_u8, void* stream);

/* ---- an A2C / PPO rollout step over NatureConvBody at 8-32 device-resident environments as four launches (agents._PixelRollout;
 * A2C_agent.py:26-34 / PPO_agent.py:33-47 under no_grad): [conv1 of step t | the policy head of step t-1], conv2, conv3
 * (dra_conv_fwd_koc), fc4 (dra_linear_fwd).  Same arithmetic as the separate launches, bit for bit. */
/* conv1 (+ ReLU) of uint8 frames [batch][4][84][84] -> y1 [batch][32][20][20]; and, when phi_prev != NULL, the policy head of the
 * previous step (workgroups of their own in the same launch: the observations do not depend on the previous step's actions):
 * fold_bias == NULL: dra_policy_heads_sample of the features phi_prev [batch][512]; fold_bias != NULL: phi_prev is the
 * [28][batch][512] K-slice partial sums of fc4 (dra_linear_fwd_slabs_one, ksplit 28) and the head folds them first
 * (dra_policy_heads_sample_fold28) */
int dra_rollout_conv1_heads(const void* frames_u8, const float* wt1, const float* b1, float* y1, int batch, double u8_coef,
                            const float* phi_prev, const float* fold_bias, const float* w_a, const float* b_a, const float* w_v,
                            const float* b_v, const float* uniform, int n_actions, int64_t* out_action, float* out_log_pi_a,
                            float* out_entropy, float* out_v, void* stream);
/* the same launch; out_phi != NULL (needs fold_bias): the head's workgroups also store the previous step's folded features
 * relu(sum of the 28 slices + bias) [batch][512] -- fc4's forward output, kept for an A2C update that backpropagates through the
 * rollout's own activations (A2C_agent.py:29-64) */
int dra_rollout_conv1_heads_phi(const void* frames_u8, const float* wt1, const float* b1, float* y1, int batch, double u8_coef,
                                const float* phi_prev, const float* fold_bias, const float* w_a, const float* b_a, const float* w_v,
                                const float* b_v, const float* uniform, int n_actions, int64_t* out_action, float* out_log_pi_a,
                                float* out_entropy, float* out_v, float* out_phi, void* stream);

/* ---- fused DQN learner + device-resident actor: DQN_agent.py:24-45 (actor step), :114-138 (update) for
 * VanillaNet(NatureConvBody).  All five flat buffers are caller-owned f32[n_params] with the tensor order
 * conv1.w, conv1.b, conv2.w, conv2.b, conv3.w, conv3.b, fc4.w, fc4.b, head.w, head.b at `offset[]` (16-byte
 * aligned); [0, conv_end) is the conv segment whose split-K slabs are folded in the norm pass. */
typedef struct dra_dqn_config {
  int32_t batch, n_actions, double_q, ksplit, centered, env_done_period;
  float gamma_n, gradient_clip, lr, alpha, eps, replay_eps, replay_alpha;
  int32_t variant;          /* DRA_VAR_* mask, or < 0 = the process default (dra_set_tuning) */
  double u8_coef;
  int64_t n_params, conv_end, ring_capacity;
  uint64_t env_seed;        /* synthetic frame source (dra_ring_fill_synthetic stream) used by the device actor */
  int64_t offset[10];
  /* head on top of NatureConvBody's 512 features (network_heads.py): DRA_HEAD_VANILLA = VanillaNet (n_actions outputs,
   * MSE TD loss, DQN_agent.py:81-99); DRA_HEAD_CATEGORICAL = CategoricalNet (n_actions x n_atoms logits over
   * linspace(v_min, v_max, n_atoms), CategoricalDQN_agent.py:60-89); DRA_HEAD_QUANTILE = QuantileNet (n_actions x n_atoms
   * quantiles, QuantileRegressionDQN_agent.py:55-77).  head.w is [n_actions * n_atoms][512], action-major. */
  int32_t head_kind, n_atoms;
  float v_min, v_max;
  /* DRA_OPT_RMSPROP (lr, alpha, eps, centered above) or DRA_OPT_ADAM (lr, beta1, beta2, eps; state1 = exp_avg,
   * state2 = exp_avg_sq; examples.py:139,204) */
  int32_t optimizer;
  float beta1, beta2;
  int32_t reserved;
} dra_dqn_config;
#define DRA_HEAD_VANILLA 0
#define DRA_HEAD_CATEGORICAL 1
#define DRA_HEAD_QUANTILE 2
#define DRA_OPT_RMSPROP 0
#define DRA_OPT_ADAM 1
/* per-agent-step arguments: the device actor's kernels read them from a device copy, so the 4 env steps of a
 * DQN agent step replay as one captured graph.  All randomness is drawn by the HOST in the reference's order
 * (torch_utils.py:51-58: randint(A) then rand()), so the np.random stream is the reference's. */
typedef struct dra_dqn_step_params {
  int64_t slot[8];           /* ring slot of each env transition (frame / action / reward / mask live there) */
  int64_t counter[8];        /* >= 0: synthesise frame `counter` (+ hashed reward / mask) into the slot; < 0: frame already there */
  int64_t rcounter[8];       /* counter whose hashes give the reward / mask stored WITH the slot: the transition that leaves
                                this observation (envs.py:140-141: the counter of the NEXT frame); = counter for a plain stream */
  int32_t random_action[8];  /* np.random.randint(A) drawn by the host */
  int32_t store_action[8];   /* write the chosen action into the slot's action record */
  float dice[8];             /* np.random.rand() drawn by the host */
  float epsilon[8];
  int32_t stack_age[8];      /* observations of the same episode before this one, capped at history-1: channel c of the
                                actor's frame stack is ring slot  slot - min(history-1-c, stack_age)  (after a reset the
                                first frame is repeated, envs.py FrameStack.reset); history-1 = plain "last 4 ring frames" */
  int32_t n_env, reserved;
  int64_t idx[1024];         /* minibatch indices of this step's update (first `batch` used) */
} dra_dqn_step_params;
typedef struct dra_dqn_learner dra_dqn_learner;
int dra_dqn_learner_create(dra_dqn_learner** out, dra_ring* ring, const dra_dqn_config* cfg, float* params,
                           float* target, float* grad, float* state1, float* state2);
int dra_dqn_learner_destroy(dra_dqn_learner* learner);
int dra_dqn_learner_buffers(dra_dqn_learner* learner, void** idx, void** sampling_prob, void** loss, void** norm,
                            void** q, void** delta, void** prio, void** actor_q);
/* one gradient update on the int64[batch] indices in the learner's idx buffer; use_graph replays a captured
 * hipGraph (stream must not be the NULL stream); per != 0 adds the PER branch (DQN_agent.py:120-127). */
int dra_dqn_learner_update(dra_dqn_learner* learner, int use_graph, int per, float beta, void* stream);
/* PER for the in-order dra_dqn_learner_step (stream_actor == NULL): importance weights from the learner's sampling_prob
 * buffer with exponent beta, new priorities into its prio buffer (DQN_agent.py:120-127) */
int dra_dqn_learner_set_per(dra_dqn_learner* learner, int per, float beta);
/* PER: sampling probabilities (host f64[batch], converted to f32) + importance exponent -> the learner's sampling_prob
 * buffer on `stream`, through the learner's own pinned staging (DQN_agent.py:120-127's tensor(sampling_prob)) */
int dra_dqn_learner_upload_sampling_prob(dra_dqn_learner* learner, const double* prob_host, int n, float beta, void* stream);
/* minibatch indices (host int64[batch], replay.py:92-103's sampled_indices) -> the learner's idx buffer on `stream`, through
 * the learner's own pinned staging (n must equal the learner's batch) */
int dra_dqn_learner_upload_indices(dra_dqn_learner* learner, const int64_t* idx_host, int n, void* stream);
/* DRA_VAR_RING_DIRECT: also gather the minibatch into the learner's buffers (dra_dqn_learner_last_minibatch) -- for
 * checkers; the update itself keeps reading the ring */
int dra_dqn_learner_keep_minibatch(dra_dqn_learner* learner, int keep);
/* PER in the pipelined step: the update is issued as [forward passes + loss] [backward + optimizer]; `stream` waits for the
 * first half of the update issued last, i.e. until its TD errors / new priorities exist (the write-back to the sum tree and
 * the next prioritized draw then run under the backward pass) */
int dra_dqn_learner_wait_loss(dra_dqn_learner* learner, void* stream);
/* measurement aid: one eager update with a HIP event in front of every kernel group; out_ms[k] = milliseconds of group k
 * (dra_dqn_learner_kernel_count groups, names from _kernel_name).  With n_out > count, out_ms[count] = the same event pair
 * with NOTHING in between (the bracket's own cost, to be subtracted).  Synchronises. */
int dra_dqn_learner_profile(dra_dqn_learner* learner, float* out_ms, int n_out, void* stream);
/* how many compute units the stream the actor launches are issued on may use (a CU-masked stream: DRA_VAR_CU_PARTITION; 0 = the
 * whole device, the default).  DRA_VAR_ACTOR_PERSIST's launch needs 32 co-resident workgroups of one per CU: with fewer CUs the
 * learner keeps the multi-launch env step.  Call before the first step (the choice is baked into the captured actor graphs). */
int dra_dqn_learner_set_actor_cus(dra_dqn_learner* learner, int n_cus);
/* DRA_VAR_DEFER_FC4: step a pending fc4 segment of the optimizer step now, on `stream` (ordered behind the update that left it).
 * Call before reading parameters / optimizer state / actor copies from outside the library (a synchronise alone does not
 * complete the step); a no-op when nothing is pending.  DQN_agent.py:133 (optimizer.step() is ONE call in the reference). */
int dra_dqn_learner_flush(dra_dqn_learner* learner, void* stream);
/* measurement aid: kernel group `kernel` (index as in _kernel_name) of the update ALONE, `reps` dependent launches in ONE
 * captured graph between two events: out_us[0] = microseconds per launch (kernel + one in-graph launch boundary), out_us[1] =
 * the same for an empty kernel (the boundary alone).  Their difference is the kernel's own duration -- the quantity rocprofv3
 * reports -- measured live.  Runs on the workspaces of the last update; refuses the optimizer group.  Synchronises. */
int dra_dqn_learner_kernel_replay(dra_dqn_learner* learner, int kernel, int reps, float* out_us, void* stream);
/* measurement aid: the chained launches the timed pipeline runs under DRA_VAR_FWD_CHAIN / DRA_VAR_BWD_CHAIN, ALONE -- which = 0:
 * conv1 + conv2 + conv3 forward of both nets (network_bodies.py:10-33, online(states) and target(next_states) of
 * DQN_agent.py:114-127); which = 1: conv3 / conv2 / conv1 backward + the two slab folds (DQN_agent.py:131).  `reps` x [the launch,
 * a one-thread launch advancing the chains' epoch word as the update's head kernel does] in ONE captured graph between two
 * events: out_us[0] = microseconds per repetition, out_us[1] = the same with an empty kernel in the launch's place; the
 * difference is the chained kernel's own duration (rocprofv3's figure for it, less the deferred fc4 optimizer segment's riders:
 * a replay never steps parameters, a pending segment is flushed first).  Runs on the workspaces of the last update.  Synchronises. */
int dra_dqn_learner_chain_replay(dra_dqn_learner* learner, int which, int reps, float* out_us, void* stream);
/* the minibatch the most recently issued update consumed (device pointers into the learner's buffers: u8 states /
 * next states [B][4][84][84], int64 actions [B], f32 rewards / masks [B]); for checkers, after a synchronise */
int dra_dqn_learner_last_minibatch(dra_dqn_learner* learner, void** state, void** next_state, void** action, void** reward,
                                   void** mask);
int dra_dqn_learner_kernel_name(int k, char* out, int n);
int dra_dqn_learner_kernel_count(void);
int dra_dqn_learner_sync_target(dra_dqn_learner* learner, void* stream); /* DQN_agent.py:136-138 */
/* the parameters were written from outside (checkpoint load, BaseAgent.py:29-33): reseed the async actor's copies */
int dra_dqn_learner_invalidate_actor_copy(dra_dqn_learner* learner);
/* DQNActor._transition on device for prm->n_env transitions (graph replay when use_graph). */
int dra_dqn_learner_act(dra_dqn_learner* learner, const dra_dqn_step_params* prm, int use_graph, void* stream);
/* DQNActor._transition's forward (DQN_agent.py:29-33) for a HOST environment: state_host = uint8 [4][84][84]
 * observation, q_host = float[n_actions] out.  The observation is read in place from the learner's mapped host staging, the
 * batch-1 forward of the online parameters is one captured graph whose last kernel publishes q and a completion word to
 * mapped host memory; returns when that word arrives (like the reference's to_np(q)); DRA_ETIMEDOUT after 10 s. */
int dra_dqn_learner_q_host(dra_dqn_learner* learner, const uint8_t* state_host, float* q_host, void* stream);
/* PrioritizedReplay.sample() on the device (dra_sumtree_per_chain2; needs the ring-direct pipeline): _set_per_chain2 once
 * before the first prioritized update; _per_chain2_seed hands the NEXT update's minibatch over from the host (first update,
 * resume); _per_chain2_wait spins until rotation slot `slot`'s block shows launch number >= seq (DRA_ETIMEDOUT after
 * timeout_us).  dra_dqn_learner_step_update / _step_actor are dra_dqn_learner_step (async mode) in two calls: the update of
 * this agent step on device-resident indices (prm: n_env only), then the NEXT step's actor transitions, prm->idx = the
 * indices of the update just issued (read back from the chain kernel's block) for the ring-slot hazard check. */
int dra_dqn_learner_set_per_chain2(dra_dqn_learner* l, dra_sumtree* tree, double* stat_dev, dra_per_chain2_io* io0,
                                   dra_per_chain2_io* io1, dra_per_chain2_io* io2, dra_per_chain2_io* io3,
                                   const uint32_t* rng_words_pinned);
int dra_dqn_learner_per_chain2_seed(dra_dqn_learner* l, const int64_t* tree_idx, const int64_t* data_idx, const double* prob,
                                    float beta, uint64_t rng_cursor, uint64_t seq, void* stream);
int dra_dqn_learner_per_chain2_wait(dra_dqn_learner* l, int slot, uint64_t seq, int64_t timeout_us);
int dra_dqn_learner_step_update(dra_dqn_learner* learner, const dra_dqn_step_params* prm, void* stream_update, void* stream_actor);
int dra_dqn_learner_step_actor(dra_dqn_learner* learner, const dra_dqn_step_params* prm, void* stream_update, void* stream_actor);
int dra_dqn_learner_next_slot(dra_dqn_learner* l, int* slot);
int dra_dqn_learner_sync_loss(dra_dqn_learner* l);
/* Async actor over a HOST environment (BaseAgent.py:142-162 with a real emulator; needs DRA_VAR_ACTOR_PARAMS):
 * _update_async = dra_dqn_learner_update whose optimizer also mirrors the new parameters into actor copy (t mod 2);
 * _q_host_async = dra_dqn_learner_q_host on `stream_actor`, reading the copy the update BEFORE the most recent one wrote --
 * the forward for agent step t+1 overlaps update t and never races with its optimizer. */
int dra_dqn_learner_update_async(dra_dqn_learner* l, int use_graph, int per, float beta, void* stream_update);
int dra_dqn_learner_q_host_async(dra_dqn_learner* l, const uint8_t* state_host, float* q_host, void* stream_actor,
                                 void* stream_update);
/* True resume (optimizer + ring + RNG state; the reference's save() keeps weights only, BaseAgent.py:24-33):
 * the learner-internal state a bit-exact continuation needs beyond what the host owns.  _resume_buffer enumerates device
 * buffers (index 0 .. _resume_buffer_count() - 1; *ptr null when this configuration has no such buffer); _resume_counters reads
 * (restore = 0) or installs (restore = 1, into a FRESH learner of the same configuration before its first step) the
 * pipelines' host-side counters (n >= 16).  Call with the learner synchronised, between agent steps. */
int dra_dqn_learner_resume_buffer_count(void);
int dra_dqn_learner_resume_buffer(dra_dqn_learner* l, int index, void** ptr, int64_t* bytes, char* name, int name_len);
int dra_dqn_learner_resume_counters(dra_dqn_learner* l, int64_t* io, int n, int restore);
/* DRA_VAR_ACTOR_RING: upload the parameter blocks of the next `n` agent steps (consumed in order, one per actor launch;
 * at most 32 may be pending).  With the ring, dra_dqn_learner_step / _act take the transitions from it and use only
 * n_env and idx of the block passed to them. */
int dra_dqn_learner_actor_ring_push(dra_dqn_learner* learner, const dra_dqn_step_params* blocks, int n, void* stream);
/* DQNAgent.step: prm->n_env actor transitions + one update on prm->idx.  stream_actor == NULL: in-order on
 * stream_update (async_actor=False semantics).  stream_actor != NULL: this call's transitions belong to the NEXT
 * step and overlap this step's update (async_actor=True; config.lock becomes HIP events). */
int dra_dqn_learner_step(dra_dqn_learner* learner, const dra_dqn_step_params* prm, int do_update, void* stream_update,
                         void* stream_actor);

/* host-side accounting of dra_dqn_learner_step since the last reset: out[0] calls, out[1] seconds in the call,
 * out[2] seconds of that blocked on a pinned staging slot (GPU back-pressure). */
int dra_dqn_learner_host_stats(dra_dqn_learner* learner, double* out, int reset);

/* DRA_VAR_TARGET_AHEAD: the stream the ahead sequences run on (on the update's CU partition; not owned by the learner; null
 * = switch the variant off).  The learner must have been created with the variant bit (workspaces).  Leaves the lane. */
int dra_dqn_learner_set_ahead_stream(dra_dqn_learner* learner, void* stream);
/* DRA_VAR_TARGET_AHEAD: the indices UniformReplay.sample (replay.py:92-110) will return for the update AFTER the coming
 * dra_dqn_learner_step call (n = batch).  The coming call issues target_network(next_states) of that minibatch
 * (DQN_agent.py:85-88) under its own update; the call after it uses the result if prm->idx equals these indices and
 * computes the target in line otherwise.  A no-op without the variant. */
int dra_dqn_learner_stage_next_indices(dra_dqn_learner* learner, const int64_t* idx_next, int n);
/* out[5]: updates whose target came from the stash, updates that computed it in line, ahead sequences skipped for a ring-slot
 * hazard, stashes dropped for an index mismatch, 1 if the variant is active. */
int dra_dqn_learner_ahead_stats(dra_dqn_learner* learner, int64_t* out);
/* DRA_VAR_FLAG_SYNC accounting since creation, out[12]: out[0] steps issued in the event-free lane, out[1] times the lane was entered,
 * out[2] steps whose actor launch waited for one more count (it overwrites slots the step's own minibatch reads), out[3] steps
 * whose update made the host wait for the actor stream (the minibatch reads slots an unfinished actor launch writes);
 * out[4..8] host nanoseconds inside those calls: pacing wait + hazard checks, index staging (+ the tagged copy command), the update's
 * launches, the actor launch, the whole call; out[9..11] reserved (0). */
int dra_dqn_learner_lane_stats(dra_dqn_learner* learner, int64_t* out);

/* HIP stream restricted to the compute units whose bit is set in cu_mask (n_words x 32 bits): the async agent step
 * gives the actor chain and the update chain disjoint CU partitions (DRA_VAR_CU_PARTITION, host side). */
int dra_stream_create_masked(void** out_stream, const uint32_t* cu_mask, int n_words);
int dra_stream_destroy(void* stream);
/* placement probe: out[2*wg] = XCC (XCD) id, out[2*wg+1] = HW_ID register of workgroup wg (device uint32[2*n]) */
int dra_probe_hw_id(uint32_t* out, int n_workgroups, void* stream);

/* timeline of the pipelined async step without a profiler: arms 5 timing events per step for the next n_steps
 * steps (actor stream: before gather, after gather, after the actor graph; update stream: before / after the
 * update graph); trace_read returns milliseconds relative to the first event, out[step*5 + slot]. */
int dra_dqn_learner_trace(dra_dqn_learner* learner, int n_steps);
int dra_dqn_learner_trace_read(dra_dqn_learner* learner, float* out_ms, int max_steps, int* n_steps);

/* ---- comm: RCCL gradient all-reduce for the data-parallel on-policy agents (SURVEY.md 8b "comm", 8e).  The reference
 * has no collective; A2C / PPO (A2C_agent.py:55-64, PPO_agent.py:77-99) shard their environments over the GPUs of a node
 * and exchange ONE flat fp32 gradient per optimizer step.  One process per GPU; rank 0 creates the id, the host ships
 * its DRA_COMM_ID_BYTES to the other ranks, every rank calls init_rank (collective). */
#define DRA_COMM_ID_BYTES 256
typedef struct dra_comm dra_comm;
int dra_comm_unique_id(void* id_bytes);
int dra_comm_init_rank(dra_comm** out, int n_ranks, int rank, const void* id_bytes);
int dra_comm_destroy(dra_comm* comm);
int dra_comm_info(dra_comm* comm, int* n_ranks, int* rank); /* as RCCL reports them (ncclCommCount / ncclCommUserRank) */
/* flat_grad <- sum over ranks of (this rank's scale * this rank's flat_grad), in place, asynchronous on stream: the scale is
 * applied BEFORE the sum, so ranks may pass different scales (PPO: rows here / rows of the global minibatch); 1/n_ranks on
 * every rank gives the gradient of the global mean over equal shards */
int dra_allreduce_grads(float* flat_grad, int64_t count, float scale, dra_comm* comm, void* stream);
/* a few fp64 scalars summed over ranks (PPO's global advantage statistics, PPO_agent.py:66) */
int dra_allreduce_f64(double* values, int count, dra_comm* comm, void* stream);

/* ---- ppo_mlp: BASELINE configs[2] (PPO on HalfCheetah shapes, examples.py:497-523) end to end on the device.
 * GaussianActorCriticNet (network_heads.py:173-214) with two separate tanh FCBody(state_dim, (H, H)) networks, H in
 * {16, 32, 64}, state_dim <= 64, action_dim <= 16, separate Adam optimisers (examples.py:508-509), mini_batch_size <= 64. */
/* deep_rl/utils/normalizer.py:28-51 (MeanStdNormalizer over baselines' RunningMeanStd, restated in
 * deeprl_amd/normalizers.py): x f64 [n][d] (device); mean / var f64 [d], count f64 [1] (device, updated when update != 0:
 * batch moments over axis 0, Chan merge, the host class's operation order); outputs clip((x - mean) / sqrt(var + epsilon),
 * +-clip) as f32 and / or f64 (either may be NULL). */
int dra_rms_normalize(const double* x, int n, int d, double* mean, double* var, double* count, int update, double epsilon,
                      double clip, float* out_f32, double* out_f64, void* stream);
/* network_heads.py:205-206 (dist.sample() = mean + scale * standard normal) with counter-hash normals: row r of mean [n][a_dim]
 * is GLOBAL environment env0 + r of n_global; *step_dev (device int64) is the sampler's position, advanced by one per call. */
int dra_gauss_sample(const float* mean, const float* scale, int n, int a_dim, uint64_t noise_seed, int64_t* step_dev,
                     int64_t n_global, int64_t env0, float* out_action, void* stream);
/* one step of n synthetic continuous environments (deeprl_amd/envs.py SyntheticContinuous, csrc/cont_env.h), auto reset on
 * done (envs.py:126-150): state f64 [n][s_dim] and counter i64 [n] are updated in place; action f32 [n][a_dim] is clipped to
 * [-1, 1] (envs.py:186-189); out_reward f64 [n], out_done i32 [n]. */
int dra_cont_env_step(double* state, int64_t* counter, const int64_t* seed, const float* action, int n, int s_dim, int a_dim,
                      int64_t horizon, double* out_reward, int32_t* out_done, void* stream);
typedef struct dra_ppo_mlp_net {     /* one of the two networks with its Adam optimiser (optim.FusedOptimizer's buffers) */
  float* param;                      /* flat parameters (device) */
  float* exp_avg;                    /* Adam first / second moments, same layout */
  float* exp_avg_sq;
  int64_t* step_dev;                 /* device: optimizer steps taken so far; advanced by every step the kernel applies */
  int32_t off_w1, off_b1, off_w2, off_b2, off_w3, off_b3, off_std, reserved;  /* float offsets; off_std < 0: no std (critic) */
  float lr, beta1, beta2, eps;
} dra_ppo_mlp_net;
typedef struct dra_ppo_mlp_cfg {
  int32_t state_dim, action_dim, hidden, mini_batch;
  float ratio_clip, entropy_weight;
  double kl_limit;                   /* the actor steps while approx_kl <= kl_limit = 1.5 * target_kl (PPO_agent.py:88) */
} dra_ppo_mlp_cfg;
int dra_ppo_mlp_supported(int state_dim, int action_dim, int hidden1, int hidden2, int mini_batch);   /* 0 = yes */
/* PPO_agent.py:72-76: the rows of every minibatch of every epoch, gathered once.  state [n][S], action [n][A], log_pi_a /
 * advantage / ret [n] (f32 device), perm i64 [epochs][n] (the np.random permutations, device) -> out_packed: one image per
 * minibatch (epochs x ceil(n / mini_batch) of them, *floats in total) in the layout the update kernel keeps in LDS:
 * 64 x (16 ceil(S / 16) + 4) observations, zero padded, then 64 x 20 = action | log_pi_a, advantage, ret at columns 16..18. */
int dra_ppo_mlp_packed_floats(int n, int epochs, int mini_batch, int s_dim, int64_t* floats);
int dra_ppo_mlp_pack(const float* state, const float* action, const float* log_pi_a, const float* advantage, const float* ret,
                     const int64_t* perm, int n, int epochs, int mini_batch, int s_dim, int a_dim, float* out_packed, void* stream);
/* PPO_agent.py:71-99 for shared_repr = False: epochs x ceil(n / mini_batch) minibatch updates in ONE launch of two persistent
 * workgroups (actor: forward, clipped-ratio loss, approx-KL gate, backward, Adam; critic: forward, value loss, backward, Adam),
 * weights and Adam moments resident in registers / LDS from the first minibatch to the last.  out3 (device f32 [3]) = policy
 * loss, value loss, approx_kl of the LAST minibatch; out_counts (device i64 [2]) = actor / critic steps applied by this launch.
 * dbg: NULL, or device f32 [DRA_PPO_MLP_DBG_FLOATS] receiving the first minibatch's intermediates (tests). */
#define DRA_PPO_MLP_DBG_FLOATS 65536
int dra_ppo_mlp_update(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                       const float* packed, int n, int epochs, float* out3, int64_t* out_counts, float* dbg, void* stream);
/* measurement aid: the same launch (hidden = 64) with shader-clock cycles per phase of the minibatch loop accumulated by thread 0
 * of each workgroup into cycles (device i64 [2][16]: actor row, critic row); tools/prof_ppo_mlp.py names the phases. */
int dra_ppo_mlp_update_profile(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                               const float* packed, int n, int epochs, float* out3, int64_t* out_counts, int64_t* cycles,
                               void* stream);
/* PPO_agent.py:32-49 over device-resident synthetic environments: t_len x [store normalised observation, no-grad forward of
 * both networks, action = mean + softplus(std) * noise, log-probability, environment step, reward / mask, running
 * observation statistics + normalisation] and the bootstrap forward, in ONE launch of one persistent workgroup. */
typedef struct dra_ppo_mlp_rollout_io {
  double* env_state;        /* [n_env][S] raw observations (in/out) */
  int64_t* env_counter;     /* [n_env] (in/out) */
  const int64_t* env_seed;  /* [n_env] */
  double* rms;              /* mean [S], var [S], count [1] (in/out; not updated when rms_update == 0) */
  float* cur_state;         /* [n_env][S] normalised observation the rollout starts from (in) / the next one starts from (out) */
  int64_t* sampler_step;    /* device int64: position of the action-noise stream (advanced by t_len + 1) */
  float* out_state;         /* [t_len][n_env][S] */
  float* out_action;        /* [t_len][n_env][A] */
  float* out_log_pi_a;      /* [t_len][n_env] */
  float* out_v;             /* [t_len + 1][n_env] */
  float* out_reward;        /* [t_len][n_env]  f32(reward * reward_coef) */
  float* out_mask;          /* [t_len][n_env]  1 - done */
  int64_t env0, n_global;   /* first GLOBAL environment index of this rank / global environment count (noise stream) */
  uint64_t noise_seed;
  int64_t horizon;
  double reward_coef, rms_epsilon, rms_clip;
  int32_t rms_update, t_len, n_env, reserved;
} dra_ppo_mlp_rollout_io;
int dra_ppo_mlp_rollout(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                        const dra_ppo_mlp_rollout_io* io, void* stream);
/* measurement aid: the same launch (hidden = 64) with thread 0's shader-clock cycles per phase of the step loop (cycles: device
 * i64 [8]: F1, F2, heads, environment, statistics, normalisation, trailing barrier) */
int dra_ppo_mlp_rollout_profile(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                                const dra_ppo_mlp_rollout_io* io, int64_t* cycles, void* stream);

#ifdef __cplusplus
}
#endif
#endif
