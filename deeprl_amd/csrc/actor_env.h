// Device-resident synthetic environment + the actor's parameter-block ring, shared by learner.hip (actor head /
// environment kernels) and conv_v2.hip (conv1 of the actor with the previous step's head and the environment
// step fused in front of it).
//
// The environment (SURVEY.md 8d "C2"): frame k of stream `seed` is a counter hash (the bytes
// dra_ring_fill_synthetic writes), reward in {-1, 0, 1} with p = .1 / .8 / .1 and done w.p. 1/done_period are hashes of
// the same counter.  obs_{t+1} = env_step(state_t, action_t): every frame generator below takes the action of the
// transition that produced the frame, so that the kernels keep the data dependence a real environment has (forward ->
// action -> environment step -> next observation -> next forward) even though THIS environment's observations do
// not depend on it.
#pragma once
#include "common.h"
#include <stddef.h>

constexpr int kMaxEnvSteps = 8;  // env transitions per agent step (sgd_update_frequency)
constexpr size_t kPrmHeadBytes = offsetof(dra_dqn_step_params, idx);  // the part the actor kernels read
constexpr int kAringSlots = 64;
constexpr size_t kAprmStride = 512;
static_assert(kPrmHeadBytes <= kAprmStride && kPrmHeadBytes % 4 == 0, "parameter ring entry");

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// 8-byte word w (0..881) of the 84x84 frame with counter `counter`; `action` = the action whose environment step
// produced this observation (unused by the synthetic source)
__device__ __forceinline__ uint64_t synth_frame_word(uint64_t seed, int64_t counter, int64_t action, int w) {
  (void)action;
  return mix64(seed * 0x9E3779B97F4A7C15ull + (uint64_t)counter * 882ull + (uint64_t)w);
}
__device__ __forceinline__ double synth_reward(uint64_t seed, int64_t counter) {
  const uint64_t hh = mix64((seed + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)counter);
  const uint32_t u = (uint32_t)(hh >> 32) % 10u;
  return (u == 0) ? -1.0 : ((u == 9) ? 1.0 : 0.0);
}
__device__ __forceinline__ int32_t synth_mask(uint64_t seed, int64_t counter, int done_period) {
  const uint64_t h2 = mix64((seed + 2) * 0x9E3779B97F4A7C15ull + (uint64_t)counter);
  return ((h2 % (uint64_t)done_period) == 0) ? 0 : 1;
}

// entry (seq mod kAringSlots) of the device parameter ring = the head of a dra_dqn_step_params
__device__ __forceinline__ const dra_dqn_step_params* aring_entry(const uint8_t* ring, unsigned seq) {
  return reinterpret_cast<const dra_dqn_step_params*>(ring + (size_t)(seq % kAringSlots) * kAprmStride);
}

// epsilon-greedy on q[0..A) with HOST-drawn randomness (torch_utils.py:51-58: np.argmax = first maximum)
__device__ __forceinline__ int64_t eps_greedy_action(const float* q, int A, const dra_dqn_step_params* prm, int e) {
  int best = 0;
  float bv = q[0];
  for (int a = 1; a < A; ++a) if (q[a] > bv) { bv = q[a]; best = a; }
  return (prm->dice[e] < prm->epsilon[e]) ? (int64_t)prm->random_action[e] : (int64_t)best;
}

// Action value of ONE action from its N head outputs x[0..N) (a distributional head at batch 1), computed by one wave:
// categorical (kind 1): sum_n softmax(x)_n * atoms[n] (CategoricalDQN_agent.py:21-24); quantile (kind 2): mean_n x[n]
// (QuantileRegressionDQN_agent.py:17-20).  Every lane returns the value.  Shared by the actor's head kernels and the fused
// conv1 launch: one summation order, bit-identical action values.
__device__ __forceinline__ float dist_action_value(const float* __restrict__ x, int N, int kind, const float* __restrict__ atoms,
                                                   int lane) {
  if (kind == 1) {
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, x[n]);
    m = wave_max(m);
    float se = 0.f;
    for (int n = lane; n < N; n += 64) se += expf(x[n] - m);
    se = wave_sum(se);
    float acc = 0.f;
    for (int n = lane; n < N; n += 64) acc += (expf(x[n] - m) / se) * atoms[n];
    return wave_sum(acc);
  }
  float acc = 0.f;
  for (int n = lane; n < N; n += 64) acc += x[n];
  return wave_sum(acc) / (float)N;
}

// What conv1 of env step e >= 1 of the ring actor needs to perform the head of step e-1 and the environment step
// in front of its own work (launch = conv1's workgroups + ONE environment workgroup, the last one):
//   every workgroup : q = head(h4) -> action of step e-1 -> the rows of observation e it convolves (generated, not read);
//   environment wg  : stores that action into ring slot[e-1], writes observation e (frame / reward / mask) to ring slot[e].
// mode 1 (env step 0): conv1 reads its newest frame from the pending buffer as before and the environment
// workgroup commits the pending observation to ring slot[0] (the feed of DQN_agent.py:104-112 happens after the
// minibatch of the previous agent step was gathered).
struct ActorFuse {
  int mode;                    // 0 = plain conv1, 1 = commit pending (env step 0), 2 = head(e-1) + env step -> e
  int e, n_actions, done_period;
  const float* h4;             // [512] fc4 output of env step e-1
  const float* wh;             // [A][512]
  const float* bh;             // [A]
  const uint8_t* aring;        // device parameter ring
  const unsigned* seq;         // agent steps completed by the actor
  uint8_t* frames; uint8_t* actions; double* rewards; int32_t* masks;   // replay ring arrays
  float* q_out;                // [A] or null
  const uint8_t* pend_frame; const double* pend_reward; const int32_t* pend_mask;
  uint64_t seed;
  // distributional heads (head_kind 1 = categorical, 2 = quantile; 0 = VanillaNet): the A * n_atoms head outputs of env
  // step e-1 were formed by actor_dist_gemv_kernel (learner.hip) into `pre`; the workgroups reduce them to action values
  int head_kind, n_atoms;
  const float* atoms;
  const float* pre;
};

// optim.hip (library-internal; DRA_VAR_DEFER_FC4, common.h DraFc4Rider): dra_clip_step_late with floats [skip_begin, + skip_count)
// left for the riders: the launch writes the clip coefficient to *defer_coef, raises *defer_pending, clears *defer_valid
int dra_clip_step_late_defer(float* param, float* grad, float* state1, float* state2, int64_t n, const dra_fold_seg* seg,
                             double* partials, int n_prior, int* timeout_flag, float max_norm, const float* hyper, int centered,
                             float* out_norm, float* param_copy, int64_t skip_begin, int64_t skip_count, float* defer_coef,
                             int* defer_pending, int* defer_valid, void* stream);
// conv_v2.hip (library-internal): what the NEXT batched conv forward launch issued from this thread carries beside its own work
// (consumed by that launch, then cleared): rider workgroups [first, first + count) of the deferred fc4 segment as extra
// z-slices, and / or the words the first workgroup sets once the riders' launches are over (pending <- 0, valid <- 1)
void dra_conv_attach_rider(const DraFc4Rider* rider, int first, int count, int* done_pending, int* done_valid);

// conv_v2.hip (library-internal)
int dra_conv1_fwd_actor_fused(const void* frames, const int64_t* slot_field_dev, const int32_t* stack_age_field_dev, const unsigned* seq_dev, int n_entries,
                              int64_t stride_bytes, int64_t capacity, const void* newest_frame, const float* wt,
                              const float* bias, float* y, double u8_coef, int act, const ActorFuse* f, void* stream);

// conv_v2.hip (library-internal): batch-1 conv2 / conv3 with the reduction split over two workgroups per output tile;
// the two partial planes y[2][OC][P] (plane 0 includes the bias) are summed and passed through the ReLU by the consumer
int dra_conv_b1_split(int layer, const float* x0, const float* x1, const float* wt, const float* bias, float* y_planes,
                      void* stream);

// conv_v2.hip / fused.hip (library-internal): conv1 of the update reading its uint8 minibatch straight from the replay ring
// (sample b = the 4 frames ending at slot idx[b] + newest_off): forward for nz nets, and the weight gradient
// (idx may be pinned host memory; idx_copy, optional: device copy of it written on the way, for the update's later kernels;
// idx_tagged + update_seq, optional: the step-tagged device copy a previous update prefetched, ConvV2Args::sample_idx_tagged)
int dra_conv1_fwd_koc_ringbatch(const void* frames, const int64_t* idx, int64_t* idx_copy, const int64_t* idx_tagged,
                                const unsigned long long* update_seq, const int64_t* newest_off, int nz,
                                const float* const* wt, const float* const* bias, float* const* y, int batch, double u8_coef, int act,
                                void* stream);
// fused.hip (library-internal): y_z = x_z W_z^T + bias_z for in_features = 512 in one pass (the distributional heads' update)
int dra_head_fwd_one(int nz, const float* const* x, const float* const* w, const float* const* bias, float* const* y, int batch,
                     int out_features, void* stream);
int dra_conv1_wgrad_ringbatch(const float* dy, const void* frames, const int64_t* idx, float* dw_slabs, float* db_slabs,
                              int64_t slab_stride, int batch, double u8_coef, int variant, void* stream);

// fused.hip (library-internal), DRA_VAR_LATE_FOLD: the backward launches with (a) the linear layers' workgroups leaving
// the sums of squares of what they store, (b) a FoldRole riding along that folds the PREVIOUS launch's weight-gradient
// slabs (`fold`, at most 32) into the flat gradient `grad` and leaves its workgroups' sums of squares; reset_slots[0, n_reset)
// (optional) are set to -1.0 by the fold's first workgroup: the arrival slots of the late-fold optimizer launch
// conv_v2.hip: conv3's forward with extra workgroups that prefetch fc4's forward weights into the L2 that will read them
int dra_conv3_fwd_koc_pf(int nz, const void* const* x, const float* const* wt, const float* const* bias, float* const* y, int batch,
                         int act, const float* const* pf_w, int pf_nz, void* stream);
int dra_fc_bwd_fused_sq_partials(int batch, int n_actions, int in_features);   // partials the call below writes
int dra_fc_bwd_fused_sq(const float* dq, const float* h4, const float* dh4, const float* x3, const float* w4, float* dwh,
                        float* dbh, float* dw4, float* db4, float* dx3, int batch, int n_actions, int in_features, int act,
                        int variant, double* sq_partials, int* n_sq_partials, const int64_t* head_action, int head_group,
                        void* stream);   // head_action / head_group (optional): the head's gradient is zero outside the
                                         // taken action's `head_group` outputs (distributional heads): those samples are skipped
int dra_conv_bwd_fused_fold(int layer, const float* dy, const void* x, const float* wt, const float* xact, float* dw, float* db,
                            int64_t slab_stride, float* dx, int batch, int act, int variant, const dra_fold_seg* fold,
                            float* grad, double* fold_partials, int* n_fold_partials, double* reset_slots, int n_reset,
                            void* stream);
// conv3's backward + the device-side prioritized draw as a riding role (fused.hip; chain = per_chain2.h PerChain2Args)
struct PerChain2Args;
int dra_conv3_bwd_fused_chain(const float* dy, const void* x, const float* wt, const float* xact, float* dw, float* db,
                              int64_t slab_stride, float* dx, int batch, int act, int variant, const PerChain2Args* chain,
                              int first_half_only, void* stream);
int dra_conv1_wgrad_fold(const float* dy, const void* x, const int64_t* idx, float* dw_slabs, float* db_slabs, int64_t slab_stride,
                         int batch, double u8_coef, int variant, const dra_fold_seg* fold, float* grad, double* fold_partials,
                         int* n_fold_partials, double* reset_slots, int n_reset, const PerChain2Args* chain, void* stream);

// ... and its default form: conv1 (fused head / environment step) and conv2 keep their launches, conv3 + fc4 share one
// (flags[2] = conv3's arrival counter)
// (w4_valid, optional: a device word the fc4 role waits to become non-zero before it requests w4 -- DRA_VAR_DEFER_FC4: the copy's
// fc4 segment is completed by riders of the update that runs beside this actor graph)
int dra_actor_c3fc4_valid(const float* y2_planes, const float* w3, const float* b3, const float* w4, const float* b4, float* y3_planes,
                          float* h4, unsigned* flags, int* timeout_flag, const int* w4_valid, void* stream);
int dra_actor_c3fc4(const float* y2_planes, const float* w3, const float* b3, const float* w4, const float* b4, float* y3_planes,
                    float* h4, unsigned* flags, int* timeout_flag, void* stream);

// conv_v2.hip (library-internal), DRA_VAR_ACTOR_PERSIST: every env step of an agent step of the ring actor as ONE launch
// (actor_persist.h; VanillaNet head; 64 co-resident workgroups of 512 threads: needs >= 32 CUs on the stream)
typedef unsigned long long ll_t;

struct ActorPersistArgs {
  const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4, *wh, *bh;
  uint8_t* frames; uint8_t* actions; double* rewards; int32_t* masks;     // replay ring arrays
  int64_t ring_cap;
  const uint8_t* aring; unsigned* seq;                                     // device parameter-block ring + step counter
  uint8_t* pend_frame; double* pend_reward; int32_t* pend_mask;
  float* q_out;                // [A] action values of the last head evaluated
  float* h4_plain;             // [512] plain copy of the last env step's features (observers)
  ll_t *y1, *y2p, *y3p, *h4;   // {value, tag} words: [32 * 400], [2 * 64 * 81], [2 * 64 * 49], [2][512]
  int* abort_word;             // device: set by the first wait that gives up, read by the others
  uint64_t seed;
  double coef;
  int done_period, n_actions, n_env;
  int* timeout_flag;           // pinned host
  const int* w4_valid;         // optional (DRA_VAR_DEFER_FC4)
  // DRA_VAR_FLAG_SYNC (all optional): instead of a stream wait on the update stream's event, every workgroup waits until
  // *fs_count (device: update launches STARTED, conv_fwd_chain_kernel) has reached fs_need[agent step mod kAringSlots] (pinned
  // host ring, written by the host before it issues this launch; 0 = nothing to wait for) BEFORE it reads anything the update
  // stream produced; the environment workgroup publishes the agent steps completed to *fs_done_host (pinned host) at its end
  const unsigned long long* fs_count;
  const unsigned long long* fs_need;
  unsigned long long* fs_done_host;
};

constexpr size_t kPersistY1 = 32 * 400, kPersistY2 = 2 * 64 * 81, kPersistY3 = 2 * 64 * 49, kPersistH4 = 2 * 512;
constexpr size_t kPersistLLWords = kPersistY1 + kPersistY2 + kPersistY3 + kPersistH4;
constexpr int kPersistWgs = 32;                 // one per CU of the actor's partition
int dra_actor_persist(const ActorPersistArgs* a, void* stream);

// conv_v2.hip (library-internal), DRA_VAR_FWD_CHAIN: conv1 (ring-direct) + conv2 + conv3 of the update's forward pass as one launch;
// done_counters = kFwdChainCounters unsigned (never reset), *epoch = chains completed (bumped by a later launch of the update);
// rider (optional, DRA_VAR_DEFER_FC4): the deferred fc4 segment as trailing workgroups of the launch, rider_count a zeroed device word
constexpr int kChainPad = 32;                  // unsigned per counter: one 128-byte line each
constexpr int kFwdChainCounters = 2 * DRA_MAX_Z * 32 * kChainPad;
void dra_conv_chain_attach_announce(unsigned long long* count);   // the next dra_conv_fwd_chain launch counts itself in *count
void dra_conv_chain_attach_zero(unsigned* word);   // DRA_VAR_HEAD_CHAIN: the next dra_conv_fwd_chain launch zeroes *word when it starts
int dra_conv_fwd_chain(const void* frames, const int64_t* idx, int64_t* idx_copy, const int64_t* idx_tagged,
                       const unsigned long long* update_seq, const int64_t* newest_off, int nz, const float* const* w1,
                       const float* const* b1, float* const* y1, const float* const* w2, const float* const* b2, float* const* y2,
                       const float* const* w3, const float* const* b3, float* const* y3, int batch, double u8_coef,
                       unsigned* done_counters, const unsigned* epoch, int* timeout_flag, const DraFc4Rider* rider,
                       unsigned* rider_count, int* rider_pending, int* rider_valid, void* stream);

// fused.hip (library-internal), DRA_VAR_BWD_CHAIN: conv3's, conv2's and conv1's backward launches (one-pass roles, late fold) of a
// minibatch of at most 32 as ONE launch in dependency order; counters = dra_bwd_chain_counters() zeroed unsigned, never reset
int dra_bwd_chain_counters(void);
// DRA_VAR_BWD_CHAIN_FC: fc4's + the head's backward (the arguments of dra_fc_bwd_fused_sq: input gradient | fc4 weight gradient |
// head weight gradient) as the LEADING roles of the same launch; conv3's roles wait for the input-gradient workgroups (one counter)
struct DraBwdChainFc {
  const float *dq, *h4, *dh4, *x3, *w4;
  float *dwh, *dbh, *dw4, *db4;
  int n_actions, in_features;
  double* sq_partials;
  int* n_sq_partials;
  const int64_t* head_action;
  int head_group;
};
int dra_conv_bwd_chain(const float* dy3, const float* y2, const float* wt3, float* dw3, float* db3, int64_t stride3, float* dy2,
                       const float* y1, const float* wt2, float* dw2, float* db2, int64_t stride2, float* dy1, const void* frames,
                       const int64_t* idx, float* dw1, float* db1, int64_t stride1, int batch, double u8_coef, int act,
                       const dra_fold_seg* fold3, const dra_fold_seg* fold2, float* grad, double* partials3, int* n_partials3,
                       double* partials2, int* n_partials2, double* reset_slots, int n_reset, unsigned* counters,
                       const unsigned* epoch, int* timeout_flag, const DraBwdChainFc* fc, void* stream);
