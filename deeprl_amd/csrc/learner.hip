// Fused DQN learner + device-resident actor: one C-ABI call per agent step.
// Replaces deep_rl/agent/DQN_agent.py:101-138 (DQNAgent.step: actor transitions, replay feed,
// sample, compute_loss, backward, clip, optimizer.step, target sync) and DQNActor._transition
// (DQN_agent.py:24-45) for VanillaNet(NatureConvBody).
//
// MI355X design (the reference issues ~640 ATen ops per update plus 4 batch-1 forwards with a
// device->host sync each, spread over three processes):
//   * the update is a fixed chain of hand-written kernels over persistent HBM workspaces, captured
//     ONCE into a hipGraph and replayed; online(states) / target(next_states) forwards share
//     launches (blockIdx.z); weight- and input-gradient kernels of a layer run on forked graph
//     branches; split-K slabs of the conv weight gradients are folded inside the norm pass;
//   * fc4's split-K reduction, bias, ReLU, both heads, the TD error, dL/dq and dL/dh4 are ONE kernel;
//   * the actor's env steps are a second captured graph whose kernels read their per-step
//     arguments (slot, epsilon, host-drawn random action / dice) from a device parameter block,
//     so a whole agent step costs two small async memcpys and three graph/kernel launches and
//     never synchronises with the host;
//   * async mode runs the actor graph of step t+1 on its own stream under the update of step t.
//     The reference's process-level overlap (BaseActor, async_actor=True) and its config.lock
//     around optimizer.step() / the actor's forward (DQN_agent.py:30,133) become HIP events: the
//     optimizer kernel is the only section exclusive with the actor's weight reads, and the
//     actor's ring writes wait for the update's gather.
#include "oneshot_lin.h"     // (LinDgradOne / LinWgradOne / HeadWgradRole: DRA_VAR_HEAD_CHAIN launches them next to the head role)
#include "actor_env.h"
#include "per_chain2.h"
#include <new>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

enum { K_GATHER, K_CONV1_F, K_CONV2_F, K_CONV3_F, K_FC4_F, K_HEAD, K_HEAD_BW, K_FC4_BW, K_FC4_BX, K_CONV3_BW, K_CONV3_BX,
       K_CONV2_BW, K_CONV2_BX, K_CONV1_BW, K_NORM, K_STEP, K_COUNT };

static const char* kKernelNames[K_COUNT] = {
  "gather", "conv1_fwd", "conv2_fwd", "conv3_fwd", "fc4_fwd", "head_loss", "head_bwd_w", "fc4_bwd_w", "fc4_bwd_x",
  "conv3_bwd_w", "conv3_bwd_x", "conv2_bwd_w", "conv2_bwd_x", "conv1_bwd_w", "grad_norm", "rmsprop_step"};

// parameter tensor order inside the flat buffers (conv segment first, 16-byte aligned offsets);
// the three conv weights are stored in the KOC layout [(c,kh,kw)][oc] (conv_v2.hip)
enum { P_W1, P_B1, P_W2, P_B2, P_W3, P_B3, P_W4, P_B4, P_WH, P_BH, P_COUNT };

constexpr int kFc4Split = 8;     // split-K of the 3136 -> 512 layer (slabs reduced inside head_fused_kernel)
constexpr int kFc4SplitWide = 28; // ... of the one-pass forward when DRA_FC4_KS=28: 448 workgroups of 112-wide K slices
                                  // instead of 128 of 392 (the layer streams 12.8 MB of weights: more CUs pulling)
constexpr int kFc4SplitMid = 14;  // ... DRA_FC4_KS=14: 224 workgroups of 224-wide slices for two nets -- one per CU of the 224-CU
                                  // update partition (a CU's share of HBM is ~32 GB/s: 128 workgroups leave 96 CUs idle)
static int fc4_ks(const struct dra_dqn_learner* l);
constexpr int kAprmSlots = 16;

struct dra_dqn_learner {
  dra_dqn_config c;
  dra_ring* ring;
  float *p, *pt, *g, *s1, *s2;
  // workspaces
  // minibatch (gather outputs), double-buffered: buffer `gb` is the one the gather / update body kernels
  // currently issued refer to (pipelined async mode alternates; every other path uses buffer 0)
  uint8_t *state_[2], *next_state_[2], *act_state;
  int64_t *action_[2], *idx;
  float *reward_[2], *mask_[2];
  int gb;
  int last_gb;                      // buffer the most recently issued gather filled
  float *y1[3], *y2[3], *y3[3], *h4, *q[3];
  float *ay1, *ay2, *ay3, *aq;  // actor (batch 1)
  float *ay2p, *ay3p;           // actor conv2 / conv3 as two partial planes each (dra_conv_b1_split)
  float *dq, *dh4, *dy3, *dy2, *dy1, *delta, *prio, *weights, *samp_prob;
  // distributional heads (c.head_kind != 0): h4 holds the features of every net ([3][B][512], z = 0 first), q[z] the
  // head outputs [B][n_out] (logits / quantiles), delta the per-sample loss vector (KL / quantile-Huber)
  int n_out;                        // head outputs per sample: A, or A * n_atoms
  float *atoms, *qr_ws, *alog;      // linspace(v_min, v_max, n_atoms); [B * n_atoms] workspace; actor head outputs [n_out]
  int64_t* opt_step;                // device: optimizer steps issued (1-based after the bump; Adam's bias corrections)
  float *slabs, *fc4_slabs, *afc4_slabs, *lin_ws;
  int64_t lin_ws_floats, slab_stride;
  int variant;                      // DRA_VAR_* mask fixed at creation
  float* lslabs[3];                 // per-layer slab buffers (one-pass weight gradients): [n_slabs][w | b]
  int64_t lstride[3];
  int lnslabs[3];
  int n_partials;                   // doubles the norm pass of this configuration writes
  double* partials;
  float *loss, *norm;
  dra_dqn_step_params* prm_dev;     // device copy read by the actor kernels
  dra_dqn_step_params* prm_stage;   // pinned staging ring
  int64_t* idx_stage;               // pinned staging ring for the minibatch indices
  int stage_k;
  hipEvent_t stage_ev[8];
  bool stage_used[8];
  hipGraphExec_t g_update;
  bool g_update_ready;
  hipGraphExec_t g_update_per;      // the same chain with PER importance weights (beta from sampling_prob[B])
  bool g_update_per_ready;
  // pipelined async mode (DRA_VAR_PIPE_GATHER): per step parity, body + optimizer in one graph
  hipGraphExec_t g_pipe[4];         // (indexed by parity in the two-copy pipelines, by step mod 4 in step_pipelined3)
  bool g_pipe_ready[4];
  hipGraphExec_t g_pipe_per[4];     // ... with PER importance weights (exponent from sampling_prob[B]): forward + loss ...
  bool g_pipe_per_ready[4];
  hipGraphExec_t g_pipe_per_b[4];   // ... and backward + optimizer, with ev_loss recorded in between
  hipEvent_t ev_loss;               // PER pipeline: the update's TD errors / new priorities exist
  hipEvent_t ev_mb_ready[2], ev_mb_free[2];   // gather of buffer b finished / update body done with buffer b
  bool mb_used[2];
  int64_t step_no;                  // async steps with an update issued so far
  // DRA_VAR_GATHER_IN_GRAPH: actor transitions + the gather of the SAME step as one graph per step parity
  hipGraphExec_t g_ag[2];
  bool g_ag_ready[2];
  int g_ag_nenv[2];
  bool ag_have_prev;                // a gathered minibatch is waiting for its update
  int ag_prev_par;
  // optional timeline of the pipelined async step (measurement aid): per traced step 5 timing events --
  // actor stream before gather / after gather / after the actor graph, update stream before / after the graph
  hipEvent_t* tr_ev;
  int tr_cap, tr_n;
  double host_wait_s, host_call_s;  // dra_dqn_learner_step: seconds blocked on a staging slot / seconds in the call
  int64_t host_calls;
  // actor graphs are keyed by (n_env, parameter block they read): online params in in-order mode, one of the
  // two actor copies in async mode (DRA_VAR_ACTOR_PARAMS)
  struct { hipGraphExec_t exec; const float* params; int n_env; bool ready; } g_actor[5];
  float* pa[4];                     // parameter copies the async actor reads: two in the pipelines whose update waits for
                                    // the previous actor graph every step, four (rotating) in step_pipelined3, which does not
  hipEvent_t pa_reader[4];          // step_pipelined3: recorded after the last actor launch that reads copy i
  int pa_cur;                       // pa[pa_cur] holds the newest completed parameters
  bool pa_valid;
  float* ah4;                       // actor fc4 output (v2)
  // q(state) for a HOST environment (dra_dqn_learner_q_host): pinned staging + one captured graph
  uint8_t* qs_stage;                // pinned, device-mapped, coherent [4 x 7056]: the host actor's observation, read by conv1 in place
  float* q_stage;                   // pinned, device-mapped, coherent [64 q values | sequence word]: written by head_q_kernel itself
  unsigned* q_seq_dev;              // device: forwards completed (head_q_kernel publishes it behind the q values)
  unsigned q_seq_host;              // host: forwards issued
  hipGraphExec_t g_q;
  bool g_q_ready;
  // actor v3: parameter blocks are read by the graph's first kernel straight from a pinned ring (no copy
  // command in front of the graph); the device counter selects the ring entry, in lockstep with aprm_seq
  uint8_t* aprm_ring;               // pinned, kAprmSlots x kAprmStride bytes
  unsigned* aprm_seq_dev;           // actor launches executed so far (device)
  unsigned* fc4_ticket;             // last-workgroup ticket of the fused fc4 + head kernel
  uint64_t aprm_seq;                // actor launches issued so far (host)
  // DRA_VAR_ACTOR_RING: parameter blocks of the next agent steps live in a DEVICE ring filled ahead of time
  // (dra_dqn_learner_actor_ring_push); the actor kernels read entry (*aring_seq mod kAringSlots), the last kernel of
  // an agent step produces the first frame of the NEXT step and advances the counter -- no per-step copy command
  // and no separate frame kernel in front of the actor graph
  uint8_t* aring_dev;               // kAringSlots x kAprmStride bytes
  uint8_t* aring_stage;             // pinned mirror
  unsigned* aring_seq;              // agent steps completed by the actor (device)
  uint8_t* pend_frame;              // the observation the actor acts on next, not yet fed to the replay ring
  double* pend_reward;              //   (DQN_agent.py:104-112 feeds a transition only after the env step)
  int32_t* pend_mask;
  uint64_t aring_pushed, aring_issued;
  bool aring_primed;
  hipGraphExec_t g_aring[4];
  bool g_aring_ready[4];
  int g_aring_nenv[4];
  hipEvent_t aprm_ev[16];
  bool aprm_used[16];
  hipStream_t side;                 // fork stream for graph branches
  hipEvent_t ev_fork, ev_join[4];
  hipEvent_t ev_actor_done, ev_gather_done, ev_step_done;
  hipEvent_t last_done;             // the event recorded after the most recent optimizer launch (ev_step_done, or the
                                    // pipelined step's per-parity event: ONE record per step on the update stream)
  bool actor_pending;               // async mode: an actor graph has been issued and not yet consumed
  int step_per;                     // dra_dqn_learner_set_per: the in-order agent step applies PER importance weights
  float step_beta;
  // DRA_VAR_RING_DIRECT: minibatch indices in four fixed pinned buffers (captured graphs bake the address; the host may
  // run up to four steps ahead), the graph / event of the same rotation, and the flag for an additional gather (checkers)
  int64_t* idx_pin[4];
  hipGraphExec_t g_rd[4], g_rd_per[4], g_rd_per_b[4];
  bool g_rd_ready[4], g_rd_per_ready[4];
  hipEvent_t ev_upd[4];
  bool upd_used[4];
  int rd_slot;                      // >= 0 while run_body is being captured / run for the ring-direct pipeline
  int keep_minibatch;
  // actor launches issued and not yet known to be complete (oldest first; at most 8: a staging slot is reused only after
  // its event was synchronised): the event recorded behind each and the ring slots it writes (n < 0: unknown = all)
  struct { hipEvent_t ev; int n; int64_t slots[8]; } arec[8];
  int arec_count;
  hipEvent_t actor_last = nullptr;  // GATHER_ON_UPDATE: recorded after the most recent actor launch (a staging-slot event)
  hipEvent_t ev[K_COUNT + 1];
  bool profiling;
  int only_kernel;                  // >= 0: run_body issues this kernel group alone (dra_dqn_learner_kernel_replay); -1 otherwise
  int only_chain;                   // 1 / 2: run_body issues the chained forward / backward launch alone (dra_dqn_learner_chain_replay)
  unsigned* hchain_dev;             // DRA_VAR_HEAD_CHAIN: arrivals of the head role's workgroups (zeroed by the next update's forward chain)
  // DRA_VAR_DEFER_FC4 (common.h DraFc4Rider): the ring-direct pipelined graphs leave fc4's segment of the optimizer step to rider
  // workgroups in the NEXT graph's conv1 / conv2 forward launches
  bool defer;                       // active for this learner (decided at creation)
  float* defer_dev;                 // device words: [0] clip coefficient (float), [1] pending (int), [2 + q] parameter copy q valid (int)
  bool defer_host;                  // an issued optimizer launch left the segment pending and no rider graph / flush has been issued since
  int defer_q;                      // ... and the rotation slot (actor copy) it belongs to
  int rider_q;                      // >= 0 while run_body captures a graph whose forward launches carry the riders for copy rider_q
  int64_t defer_begin, defer_count; // the segment (floats)
  hipEvent_t ev_flush;              // recorded behind a flush (becomes last_done)
  int* timeout_flag;                // pinned host: set by a workgroup whose bounded device-side wait gave up (late_step's arrival
                                    // slots, the actor's in-launch hand-over): every later step / update returns DRA_ETIMEDOUT
  // DRA_VAR_IDX_PREFETCH (ring-direct pipeline): step-tagged copies of the minibatch indices -- pinned (written by the host
  // with the indices), device (an unordered async copy on the side stream), and the device count of completed updates
  int64_t* idx_tag_pin[4];
  int64_t* idx_tag_dev;             // [4][1024]
  unsigned long long* rd_seq_dev;   // ring-direct updates completed (bumped by each one's head kernel)
  uint64_t rd_issued;               // ring-direct updates issued (host)
  float* sp_stage;                  // pinned [8][1025]: staging of dra_dqn_learner_upload_sampling_prob
  hipEvent_t sp_ev[8];
  bool sp_used[8];
  int sp_k;
  int64_t* ui_stage;                // pinned [8][1024]: staging of dra_dqn_learner_upload_indices
  hipEvent_t ui_ev[8];
  bool ui_used[8];
  int ui_k;
  // DRA_VAR_LATE_FOLD: no gradient-norm launch (optim.hip late_step_kernel): sums of squares from the producing kernels,
  // conv3 / conv2 folds riding in the next backward launch, conv1's fold in front of the optimizer launch
  // host-environment async actor (dra_dqn_learner_update_async / _q_host_async): update t mirrors its parameters into copy
  // t mod 2, the batch-1 forwards of agent step t+1 read the copy update t-1 wrote, on the actor stream
  hipGraphExec_t g_qa[2];
  bool g_qa_ready[2];
  hipEvent_t ev_hq[2];
  int64_t hq_updates;               // async updates issued
  bool hq_seeded;                   // copy (hq_updates - 1) mod 2 holds valid parameters
  // the prioritized draw inside the update (dra_dqn_learner_set_per_chain2): the replay's tree, its {max, min} pair and one pinned
  // io block per rotation slot; the whole draw on the device; the next update's minibatch indices arrive in
  // per2_idx[slot] (device), the sampling probabilities in samp_prob, without the host in between
  dra_sumtree* per_tree;
  double* per_stat;
  dra_per_chain2_io* per2_io[4];
  void* per2_dev;                   // sumtree.hip PerChain2Dev
  int64_t* per2_idx;                // [4][1024]
  const uint32_t* per2_words;       // pinned ring of Mersenne-Twister words (host-generated ahead)
  int split_q;                      // step_pipelined3 in two calls: state carried from the update half to the actor half
  hipEvent_t split_opt_prev;
  bool split_seed, split_open;
  bool per2_active;                 // capturing the one-graph prioritized update: weights precomputed, priorities by the chain kernel
  bool per2_ride;                   // ... and the chain kernel rides in conv3's backward launch (per2_args) instead of its own
  bool per2_split;                  // ... its second half in conv1's weight-gradient launch (late-fold backward only)
  PerChain2Args per2_args;
  unsigned* fchain_dev;             // DRA_VAR_FWD_CHAIN: [kFwdChainCounters] arrival counters (never reset) + [1] chains completed
  bool fchain;                      // the update's conv forwards run as one chained launch
  unsigned* bchain_dev;             // DRA_VAR_BWD_CHAIN: arrival counters of the chained backward launch (never reset)
  bool bchain;                      // the update's conv backward launches run as one chained launch
  unsigned long long* all_dev;      // DRA_VAR_ACTOR_PERSIST: {value, tag} hand-over arrays (y1 | y2p | y3p | h4[2]) + the abort word
  int actor_cus;                    // CUs of the stream the actor launches run on (0 = unknown: the whole device)
  unsigned* aflags;                 // DRA_VAR_ACTOR_MEGA: [kMaxEnvSteps][4] arrival counters of the one-launch env steps (zeroed by
                                    // the agent step's tail kernel)
  // DRA_VAR_FLAG_SYNC (step_lane): the steady-state pipelined step without events.  fs = the learner can run it (decided at
  // creation), fs_on = the streams currently carry lane work that no event covers (every other entry point leaves the lane first)
  bool fs, fs_on;
  unsigned long long* fs_count;     // device (own 256 bytes): ring-direct update graphs STARTED (conv_fwd_chain_kernel's first workgroup)
  uint64_t fs_issued;               // host: ... issued (every launch of a graph captured with the announce, + fs_bump launches)
  unsigned long long* fs_host;      // pinned: [0] agent steps completed by the persistent actor, [8 + i] `need` of agent step i mod kAringSlots
  bool fs_graph[4];                 // g_rd[q] was captured with the announce
  bool fs_agraph[4];                // g_aring[k] holds the persistent actor launch (the one that polls fs_count)
  bool fs_persist_taken;            // set by run_actor_steps_ring_fused when it issues the persistent launch
  bool fs_capturing;                // run_body is capturing such a graph
  uint64_t fs_need_next;            // `need` of the next actor launch issue_actor_ring issues (0 = none; consumed there)
  uint64_t fs_reader[4];            // agent-step count (aring_issued + 1) of the last actor launch that reads copy q (0: none pending)
  struct { uint64_t done_at; int n; int64_t slots[8]; } fs_arec[4];   // the last four lane actor launches: complete once fs_host[0] >= done_at
  int fs_arec_n;
  hipStream_t fs_su, fs_sa;         // the streams the lane runs on (while fs_on)
  int64_t fs_stat[12];              // lane steps, lane entries, hazard bumps, host waits for the actor stream; [4..8] host nanoseconds in the
                                    // lane call: pacing wait, index staging (+ tag copy), update launches, actor launch, whole call
  hipEvent_t ev_fs;                 // recorded on the actor stream when the lane is left (actor_last of the event paths)
  // DRA_VAR_TARGET_AHEAD (step_lane + rd_eager): target(next_states) of update t + 1 -- conv1 + conv2 + conv3 + fc4's split-K partial
  // sums, which depend on the target parameters and the ring only -- runs on its own stream UNDER update t (gated on update t's
  // start), the update's forward chain carries the online net alone and its head kernel reads the target's partial sums from a
  // stash.  ah = the learner can (decided at creation + dra_dqn_learner_set_ahead_stream); everything per parity of the update
  // number: ahead(t + 1) works on set (t + 1) & 1 while update t may still read / fall back on set t & 1.
  bool ah;
  hipStream_t ah_stream;            // (not owned: a stream on the update's CU partition)
  float *ah_y1[2], *ah_y2[2], *ah_y3[2];   // the target net's activations (nothing reads them after fc4)
  float* ah_slabs[2];               // [ks][B][512]: what the head kernel folds for z = 1
  unsigned* ah_chain[2];            // arrival counters + epoch word of the target chain's launches (never reset: each use adds one epoch)
  int64_t* ah_idx_pin[8];           // pinned: the indices ahead(t + 1) reads (rotation of 8: the host runs up to three calls ahead of
                                    // the device, an ahead sequence is only known complete at its update's head kernel)
  int64_t* ah_idx_copy;             // device scratch (conv1 leaves a copy of what it read)
  unsigned long long* ah_done;      // device: number (+ 1) of the newest update whose ahead launches are complete
  int64_t ah_next_idx[1024];        // host: dra_dqn_learner_stage_next_indices
  bool ah_next_set;
  int64_t ah_idx[1024];             // host: the indices the stash in flight was computed for ...
  uint64_t ah_for_step;             // ... and the update (step_no) it belongs to
  bool ah_valid;
  int ah_mode;                      // run_body: 0 = both nets in the update's chain, 1 = z = 1 from the stash, 2 = z = 1 by the target
                                    // chain on the update stream first (no stash for this update)
  int64_t ah_stat[4];               // updates served from the stash, computed in line, ahead launches skipped for a slot hazard, index mismatches
  bool late;
  int late_nprior;                  // partials written before the optimizer launch
  int late_nfold;                   // fold workgroups of the optimizer launch (their partial slots double as arrival flags)
  bool captured;                    // some graph has been captured (the decision above is baked into it)
};

struct HeadSpec;
static HeadSpec head_spec(const dra_dqn_learner* l);
static int fs_leave_any(dra_dqn_learner* l);   // DRA_VAR_FLAG_SYNC: leave the event-free lane (both streams drained) -- defined with step_lane

// K slices of the update's fc4 forward (one-pass kernel only).  Default: 14 for two nets (224 workgroups: one per CU of the
// update partition; fc4_fwd 10.4 -> 8.2 us, +1.8 % updates/s same box, profiles/r02zu_*), 8 with the third net of double-Q
// (192 workgroups; 336 would need a second round).  DRA_FC4_KS = 8 / 14 / 28 overrides.
static int fc4_ks(const dra_dqn_learner* l) {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DRA_FC4_KS");
    const int want = e ? atoi(e) : 0;
    v = (want == kFc4SplitWide || want == kFc4SplitMid || want == kFc4Split) ? want : 0;
  }
  if (!(l->variant & DRA_VAR_ONESHOT_FWD)) return kFc4Split;
  if (v) return v;
  return l->c.double_q ? kFc4Split : kFc4SplitMid;
}

// HIP stream restricted to a set of compute units (bit i of cu_mask = CU i enabled).  The async agent step runs
// two latency-bound kernel chains concurrently; without a partition every small actor kernel queues behind
// whatever workgroups of the update currently fill the chip (measured: actor chain 116 us alone, 165 us under
// the update).  A disjoint CU partition gives each chain its own workgroup slots.
DRA_API int dra_stream_create_masked(void** out, const uint32_t* cu_mask, int n_words) {
  if (!out || !cu_mask || n_words < 1) return DRA_EINVAL;
  hipStream_t s;
  DRA_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask));
  *out = (void*)s;
  return DRA_OK;
}

// Where did a workgroup land?  out[2*wg] = XCC id, out[2*wg+1] = HW_ID (cu_id bits 11:8, sh 12, se 15:13).
// Used by tools/probe_cu_mask.py to map CU-mask bits to XCDs.
__global__ void hw_id_kernel(uint32_t* __restrict__ out) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

DRA_API int dra_probe_hw_id(uint32_t* out, int n_workgroups, void* stream) {
  if (!out || n_workgroups < 1) return DRA_EINVAL;
  hipLaunchKernelGGL(hw_id_kernel, dim3(n_workgroups), dim3(64), 0, dra_stream(stream), out);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_stream_destroy(void* stream) {
  if (!stream) return DRA_OK;
  DRA_HIP(hipStreamDestroy(dra_stream(stream)));
  return DRA_OK;
}

constexpr int kMaxHeadOut = 4096;   // n_actions * n_atoms the batch-1 actor head keeps in LDS

static int alloc_f(float** p, int64_t n) { return (int)hipMalloc(p, (size_t)n * sizeof(float)); }

DRA_API int dra_dqn_learner_create(dra_dqn_learner** out, dra_ring* ring, const dra_dqn_config* cfg, float* params,
                                   float* target, float* grad, float* state1, float* state2) {
  if (!out || !ring || !cfg || !params || !target || !grad || !state1 || !state2) return DRA_EINVAL;
  if (cfg->batch < 1 || cfg->batch > 1024 || cfg->n_actions < 1 || cfg->n_actions > 64 || cfg->ksplit < 1 ||
      cfg->ksplit > 64)
    return DRA_EINVAL;
  if (cfg->head_kind < DRA_HEAD_VANILLA || cfg->head_kind > DRA_HEAD_QUANTILE || cfg->optimizer < DRA_OPT_RMSPROP ||
      cfg->optimizer > DRA_OPT_ADAM)
    return DRA_EINVAL;
  if (cfg->head_kind != DRA_HEAD_VANILLA &&
      (cfg->n_atoms < 2 || cfg->n_atoms > 1024 || (int64_t)cfg->n_actions * cfg->n_atoms > kMaxHeadOut ||
       (cfg->head_kind == DRA_HEAD_CATEGORICAL && !(cfg->v_max > cfg->v_min))))
    return DRA_EINVAL;
  dra_dqn_learner* l = new (std::nothrow) dra_dqn_learner();
  if (!l) return DRA_ENOMEM;
  memset(l, 0, sizeof(*l));
  l->c = *cfg; l->ring = ring; l->p = params; l->pt = target; l->g = grad; l->s1 = state1; l->s2 = state2;
  l->rd_slot = -1;
  l->only_kernel = -1;
  if (cfg->head_kind == DRA_HEAD_QUANTILE) l->c.double_q = 0;   // QuantileRegressionDQN_agent.py:58-60: target network only
  const int B = cfg->batch, A = cfg->n_actions;
  const int nz = l->c.double_q ? 3 : 2;
  const int NO = cfg->head_kind == DRA_HEAD_VANILLA ? A : A * cfg->n_atoms;
  l->n_out = NO;
  int rc = 0;
  for (int g = 0; g < 2; ++g) {
    rc |= (int)hipMalloc(&l->state_[g], (size_t)B * 4 * 7056);
    rc |= (int)hipMalloc(&l->next_state_[g], (size_t)B * 4 * 7056);
    rc |= (int)hipMalloc(&l->action_[g], (size_t)B * 8);
    rc |= alloc_f(&l->reward_[g], B); rc |= alloc_f(&l->mask_[g], B);
  }
  rc |= (int)hipMalloc(&l->act_state, (size_t)4 * 7056);
  rc |= (int)hipMalloc(&l->idx, (size_t)B * 8);
  for (int z = 0; z < nz; ++z) {
    rc |= alloc_f(&l->y1[z], (int64_t)B * 32 * 400); rc |= alloc_f(&l->y2[z], (int64_t)B * 64 * 81);
    rc |= alloc_f(&l->y3[z], (int64_t)B * 64 * 49); rc |= alloc_f(&l->q[z], (int64_t)B * NO);
  }
  rc |= alloc_f(&l->h4, (int64_t)3 * B * 512);
  rc |= (int)hipMalloc(&l->opt_step, sizeof(int64_t));
  if (!rc) rc |= (int)hipMemset(l->opt_step, 0, sizeof(int64_t));
  if (cfg->head_kind != DRA_HEAD_VANILLA) {
    const int N = cfg->n_atoms;
    rc |= alloc_f(&l->atoms, N); rc |= alloc_f(&l->qr_ws, (int64_t)B * N); rc |= alloc_f(&l->alog, NO);
    if (!rc) {   // np.linspace(v_min, v_max, N) in fp64 (start + i * step, the end point exact), then fp32 (tensor())
      float host_atoms[1024];
      const double step = ((double)cfg->v_max - (double)cfg->v_min) / (double)(N - 1);
      for (int i = 0; i < N; ++i) host_atoms[i] = (float)((double)cfg->v_min + (double)i * step);
      host_atoms[N - 1] = cfg->v_max;
      rc |= (int)hipMemcpy(l->atoms, host_atoms, (size_t)N * sizeof(float), hipMemcpyHostToDevice);
    }
  }
  rc |= alloc_f(&l->ay1, 32 * 400); rc |= alloc_f(&l->ay2, 64 * 81); rc |= alloc_f(&l->ay3, 64 * 49);
  rc |= alloc_f(&l->aq, A);
  rc |= alloc_f(&l->ay2p, 2 * 64 * 81); rc |= alloc_f(&l->ay3p, 2 * 64 * 49);
  rc |= alloc_f(&l->dq, (int64_t)B * NO); rc |= alloc_f(&l->dh4, (int64_t)B * 512);
  rc |= alloc_f(&l->dy3, (int64_t)B * 64 * 49); rc |= alloc_f(&l->dy2, (int64_t)B * 64 * 81);
  rc |= alloc_f(&l->dy1, (int64_t)B * 32 * 400);
  // (the quantile head's loss vector has one entry per target quantile: QuantileRegressionDQN_agent.py:75-77)
  rc |= alloc_f(&l->delta, (cfg->head_kind == DRA_HEAD_QUANTILE && cfg->n_atoms > B) ? cfg->n_atoms : B);
  rc |= alloc_f(&l->prio, B); rc |= alloc_f(&l->weights, B);
  rc |= alloc_f(&l->samp_prob, B + 1);   // [B] = the PER exponent beta of the update (graph-replayable PER launches)
  l->slab_stride = cfg->conv_end;  // conv segment occupies [0, conv_end) of the flat layout
  rc |= alloc_f(&l->slabs, (int64_t)cfg->ksplit * l->slab_stride);
  if (cfg->variant >= 0) l->variant = cfg->variant;
  else rc |= dra_get_tuning(&l->variant);
  // ring-direct needs the one-launch-per-layer backward (its conv1 weight gradient) and the host-decided cross-stream waits
  // of the gather-on-update pipeline (the distributional heads get their transition scalars from fc4_reduce_kernel, which
  // folds them from the ring as head_fused_kernel does for VanillaNet)
  if ((l->variant & DRA_VAR_RING_DIRECT) &&
      (!(l->variant & DRA_VAR_ONESHOT_WGRAD) || !(l->variant & DRA_VAR_GATHER_ON_UPDATE) || !(l->variant & DRA_VAR_PINNED_IDX)))
    l->variant &= ~DRA_VAR_RING_DIRECT;
  if (cfg->head_kind != DRA_HEAD_VANILLA) {
    // the distributional heads exist in the second-generation actor (own head kernel per env step, in order or from the
    // parameter ring); the launches that fold the VanillaNet head into a neighbouring kernel do not apply
    l->variant |= DRA_VAR_ACTOR_V2;
    l->variant &= ~(DRA_VAR_ACTOR_V3 | DRA_VAR_ACTOR_FUSED_HEAD | DRA_VAR_GATHER_IN_GRAPH);
    // (the ring actor's fused conv1 launch reduces the head outputs of actor_dist_gemv_kernel to action values; the six-launch
    // env step with its own head kernel was the A/B partner until round 6: DRA_ACTOR_DIST_FUSED, retired)
  }
  if (l->variant & DRA_VAR_ONESHOT_WGRAD) {
    // layer L's segment of the flat gradient is [offset(W_L), offset(b_L) + OC): weight then bias, contiguous
    const int wi[3] = {P_W1, P_W2, P_W3};
    const int64_t seg_end[3] = {cfg->offset[P_W2], cfg->offset[P_W3], cfg->conv_end};
    for (int k = 0; k < 3; ++k) {
      l->lstride[k] = seg_end[k] - cfg->offset[wi[k]];
      rc |= dra_conv_wgrad_slabs(k + 1, B, cfg->ksplit, l->variant, &l->lnslabs[k]);
      if (rc) break;
      rc |= alloc_f(&l->lslabs[k], (int64_t)l->lnslabs[k] * l->lstride[k]);
      if (!rc) rc |= (int)hipMemset(l->lslabs[k], 0, (size_t)l->lnslabs[k] * l->lstride[k] * sizeof(float));
    }
  }
  l->n_partials = 2 * dra_norm_partials();
  if ((l->variant & DRA_VAR_LATE_FOLD) && (l->variant & DRA_VAR_ONESHOT_WGRAD) && (l->variant & DRA_VAR_ONESHOT_DGRAD) &&
      (l->variant & DRA_VAR_FUSED_BWD) && !rc && cfg->offset[P_W1] == 0 && l->lnslabs[1] <= 32 && l->lnslabs[2] <= 32) {
    // partials: fc4's weight-gradient workgroups + head workgroups + one per 256 folded floats of conv3 / conv2 + conv1's
    // fold workgroups in the optimizer launch
    dra_fold_seg s0;
    memset(&s0, 0, sizeof(s0));
    s0.begin = 0; s0.count = l->lstride[0]; s0.slabs = l->lslabs[0]; s0.slab_stride = l->lstride[0]; s0.n_slabs = l->lnslabs[0];
    if (dra_clip_step_late_blocks(&s0, &l->late_nfold) == DRA_OK) {
      const int64_t np = dra_fc_bwd_fused_sq_partials(cfg->batch, NO, 3136) + (l->lstride[2] / 4 + 63) / 64 + (l->lstride[1] / 4 + 63) / 64 + l->late_nfold;
      l->late = np <= dra_norm_partials_max();
    }
  }
  rc |= alloc_f(&l->ah4, 512);
  l->rider_q = -1;
  l->defer_begin = cfg->offset[P_W4];
  l->defer_count = cfg->offset[P_B4] - cfg->offset[P_W4];
  {
    const int need = DRA_VAR_DEFER_FC4 | DRA_VAR_RING_DIRECT | DRA_VAR_GATHER_ON_UPDATE | DRA_VAR_ACTOR_PARAMS | DRA_VAR_ACTOR_RING |
                     DRA_VAR_ACTOR_FUSED_CONV1 | DRA_VAR_ACTOR_MEGA | DRA_VAR_ONESHOT_FWD;
    l->defer = (l->variant & need) == need && l->late && cfg->head_kind == DRA_HEAD_VANILLA && cfg->optimizer == DRA_OPT_RMSPROP &&
               !cfg->double_q && cfg->batch > 16 && cfg->batch < 128 &&            // (the forwards' four-wave latency shape)
               (l->defer_begin & 3) == 0 && (l->defer_count & 3) == 0 && l->defer_count == (int64_t)512 * 3136 &&
               l->defer_begin >= l->lstride[0];
  }
  // DRA_VAR_FWD_CHAIN: VanillaNet, ring-direct, two nets, the four-wave latency shape
  l->fchain = (l->variant & DRA_VAR_FWD_CHAIN) && (l->variant & DRA_VAR_RING_DIRECT) && cfg->head_kind == DRA_HEAD_VANILLA &&
              !cfg->double_q && cfg->batch > 16 && cfg->batch <= 32;
  // DRA_VAR_BWD_CHAIN: the one-pass backward roles with the late fold, VanillaNet, ring-direct, batch 17..32
  l->bchain = (l->variant & DRA_VAR_BWD_CHAIN) && (l->variant & DRA_VAR_RING_DIRECT) && l->late && cfg->head_kind == DRA_HEAD_VANILLA &&
              !cfg->double_q && cfg->batch > 16 && cfg->batch <= 32;
  rc |= (int)hipMalloc(&l->bchain_dev, (size_t)dra_bwd_chain_counters() * sizeof(unsigned));
  if (!rc) rc |= (int)hipMemset(l->bchain_dev, 0, (size_t)dra_bwd_chain_counters() * sizeof(unsigned));
  rc |= (int)hipMalloc(&l->defer_dev, 8 * sizeof(float));
  if (!rc) {
    const int init[8] = {0, 0, 1, 1, 1, 1, 0, 0};      // coefficient 0.0f, nothing pending, every copy valid
    rc |= (int)hipMemcpy(l->defer_dev, init, sizeof(init), hipMemcpyHostToDevice);
  }
  // DRA_VAR_FLAG_SYNC: the forward chain announces, the persistent actor polls and publishes, the step is ring-direct with the
  // parameter-block ring and four rotating copies
  {
    const int need = DRA_VAR_FLAG_SYNC | DRA_VAR_RING_DIRECT | DRA_VAR_GATHER_ON_UPDATE | DRA_VAR_PIPE_GATHER | DRA_VAR_ACTOR_PARAMS |
                     DRA_VAR_ACTOR_RING | DRA_VAR_ACTOR_FUSED_CONV1 | DRA_VAR_ACTOR_MEGA | DRA_VAR_ACTOR_PERSIST;
    l->fs = (l->variant & need) == need && l->fchain && cfg->head_kind == DRA_HEAD_VANILLA &&
            !(l->variant & (DRA_VAR_GATHER_IN_GRAPH | DRA_VAR_ACTOR_V3));
  }
  rc |= (int)hipMalloc(&l->fs_count, 256);
  if (!rc) rc |= (int)hipMemset(l->fs_count, 0, 256);
  rc |= (int)hipHostMalloc(&l->fs_host, (size_t)(8 + kAringSlots) * sizeof(unsigned long long), hipHostMallocDefault);
  if (!rc) memset(l->fs_host, 0, (size_t)(8 + kAringSlots) * sizeof(unsigned long long));
  rc |= (int)hipEventCreateWithFlags(&l->ev_fs, hipEventDisableTiming);
  rc |= (int)hipMalloc(&l->aflags, (size_t)kMaxEnvSteps * 4 * sizeof(unsigned));
  rc |= (int)hipMalloc(&l->hchain_dev, 256);
  if (!rc) rc |= (int)hipMemset(l->hchain_dev, 0, 256);
  rc |= (int)hipMalloc(&l->fchain_dev, (size_t)(kFwdChainCounters + 2) * sizeof(unsigned));
  if (!rc) rc |= (int)hipMemset(l->fchain_dev, 0, (size_t)(kFwdChainCounters + 2) * sizeof(unsigned));
  // DRA_VAR_TARGET_AHEAD: workspaces (the stream comes later: dra_dqn_learner_set_ahead_stream)
  l->ah = false;
  if ((l->variant & DRA_VAR_TARGET_AHEAD) && l->fs && (l->variant & DRA_VAR_LANE_EAGER) && l->bchain && !cfg->double_q && B <= 32 &&
      (l->variant & DRA_VAR_ONESHOT_FWD)) {
    for (int k = 0; k < 2; ++k) {
      rc |= alloc_f(&l->ah_y1[k], (int64_t)B * 32 * 400); rc |= alloc_f(&l->ah_y2[k], (int64_t)B * 64 * 81);
      rc |= alloc_f(&l->ah_y3[k], (int64_t)B * 64 * 49); rc |= alloc_f(&l->ah_slabs[k], (int64_t)kFc4SplitWide * B * 512);
      rc |= (int)hipMalloc(&l->ah_chain[k], (size_t)(kFwdChainCounters + 2) * sizeof(unsigned));
      if (!rc) rc |= (int)hipMemset(l->ah_chain[k], 0, (size_t)(kFwdChainCounters + 2) * sizeof(unsigned));
    }
    for (int k = 0; k < 8; ++k) rc |= (int)hipHostMalloc(&l->ah_idx_pin[k], 1024 * sizeof(int64_t), hipHostMallocDefault);
    rc |= (int)hipMalloc(&l->ah_idx_copy, 1024 * sizeof(int64_t));
    rc |= (int)hipMalloc(&l->ah_done, 256);
    if (!rc) rc |= (int)hipMemset(l->ah_done, 0, 256);
  }
  {
    const size_t words = kPersistLLWords + 1;
    rc |= (int)hipMalloc(&l->all_dev, words * sizeof(unsigned long long));
    if (!rc) rc |= (int)hipMemset(l->all_dev, 0, words * sizeof(unsigned long long));   // tag 0 is never used
  }
  if (!rc) rc |= (int)hipMemset(l->aflags, 0, (size_t)kMaxEnvSteps * 4 * sizeof(unsigned));
  if (l->variant & DRA_VAR_ACTOR_PARAMS) { rc |= alloc_f(&l->pa[0], cfg->n_params); rc |= alloc_f(&l->pa[1], cfg->n_params); }
  if ((l->variant & DRA_VAR_ACTOR_PARAMS) && (l->variant & DRA_VAR_GATHER_ON_UPDATE)) {
    rc |= alloc_f(&l->pa[2], cfg->n_params); rc |= alloc_f(&l->pa[3], cfg->n_params);
  }
  if (l->variant & DRA_VAR_ACTOR_V3) {
    rc |= (int)hipHostMalloc(&l->aprm_ring, kAprmSlots * kAprmStride, hipHostMallocDefault);
    rc |= (int)hipMalloc(&l->aprm_seq_dev, sizeof(unsigned));
    rc |= (int)hipMalloc(&l->fc4_ticket, sizeof(unsigned));
    if (!rc) rc |= (int)hipMemset(l->aprm_seq_dev, 0, sizeof(unsigned));
    if (!rc) rc |= (int)hipMemset(l->fc4_ticket, 0, sizeof(unsigned));
    for (int k = 0; k < kAprmSlots; ++k) rc |= (int)hipEventCreateWithFlags(&l->aprm_ev[k], hipEventDisableTiming);
  }
  if (l->variant & DRA_VAR_ACTOR_RING) {
    rc |= (int)hipMalloc(&l->aring_dev, kAringSlots * kAprmStride);
    rc |= (int)hipHostMalloc(&l->aring_stage, kAringSlots * kAprmStride, hipHostMallocDefault);
    rc |= (int)hipMalloc(&l->aring_seq, sizeof(unsigned));
    rc |= (int)hipMalloc(&l->pend_frame, 7056);
    rc |= (int)hipMalloc(&l->pend_reward, sizeof(double));
    rc |= (int)hipMalloc(&l->pend_mask, sizeof(int32_t));
    if (!rc) rc |= (int)hipMemset(l->aring_seq, 0, sizeof(unsigned));
    if (!rc) rc |= (int)hipMemset(l->aring_dev, 0, kAringSlots * kAprmStride);
  }
  rc |= alloc_f(&l->fc4_slabs, (int64_t)3 * kFc4SplitWide * B * 512);
  rc |= alloc_f(&l->afc4_slabs, (int64_t)kFc4Split * 512);
  l->lin_ws_floats = (int64_t)3 * 32 * B * 512;
  rc |= alloc_f(&l->lin_ws, l->lin_ws_floats);
  rc |= (int)hipMalloc(&l->partials, (size_t)dra_norm_partials_max() * sizeof(double));
  rc |= alloc_f(&l->loss, 1); rc |= alloc_f(&l->norm, 1);
  if (l->variant & DRA_VAR_IDX_PREFETCH) {
    if (!(l->variant & DRA_VAR_RING_DIRECT) || cfg->ring_capacity >= (1ll << 40)) l->variant &= ~DRA_VAR_IDX_PREFETCH;
  }
  if (l->variant & DRA_VAR_IDX_PREFETCH) {
    for (int k = 0; k < 4; ++k) {
      rc |= (int)hipHostMalloc(&l->idx_tag_pin[k], (size_t)1024 * sizeof(int64_t), hipHostMallocDefault);
      if (!rc) memset(l->idx_tag_pin[k], 0, (size_t)1024 * sizeof(int64_t));
    }
    rc |= (int)hipMalloc(&l->idx_tag_dev, (size_t)4 * 1024 * sizeof(int64_t));
    if (!rc) rc |= (int)hipMemset(l->idx_tag_dev, 0, (size_t)4 * 1024 * sizeof(int64_t));
    rc |= (int)hipMalloc(&l->rd_seq_dev, sizeof(unsigned long long));
    if (!rc) rc |= (int)hipMemset(l->rd_seq_dev, 0, sizeof(unsigned long long));
  }
  rc |= (int)hipHostMalloc(&l->sp_stage, (size_t)8 * 1025 * sizeof(float), hipHostMallocDefault);
  for (int k = 0; k < 8; ++k) rc |= (int)hipEventCreateWithFlags(&l->sp_ev[k], hipEventDisableTiming);
  rc |= (int)hipHostMalloc(&l->ui_stage, (size_t)8 * 1024 * sizeof(int64_t), hipHostMallocDefault);
  for (int k = 0; k < 8; ++k) rc |= (int)hipEventCreateWithFlags(&l->ui_ev[k], hipEventDisableTiming);
  rc |= (int)hipHostMalloc(&l->timeout_flag, sizeof(int), hipHostMallocDefault);
  if (!rc) *l->timeout_flag = 0;
  rc |= (int)hipMalloc(&l->prm_dev, sizeof(dra_dqn_step_params));
  rc |= (int)hipHostMalloc(&l->prm_stage, 8 * sizeof(dra_dqn_step_params), hipHostMallocDefault);
  rc |= (int)hipHostMalloc(&l->idx_stage, (size_t)8 * 1024 * sizeof(int64_t), hipHostMallocDefault);
  for (int k = 0; k < 4; ++k) {
    rc |= (int)hipHostMalloc(&l->idx_pin[k], (size_t)1024 * sizeof(int64_t), hipHostMallocDefault);
    rc |= (int)hipEventCreateWithFlags(&l->ev_upd[k], hipEventDisableTiming);
  }
  // (coherent = fine-grained: the device reads the observation over the fabric uncached, the host sees the kernel's stores
  // while the stream is still busy)
  rc |= (int)hipHostMalloc(&l->qs_stage, (size_t)4 * 7056, hipHostMallocMapped | hipHostMallocCoherent);
  rc |= (int)hipHostMalloc(&l->q_stage, 80 * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent);
  rc |= (int)hipMalloc(&l->q_seq_dev, sizeof(unsigned));
  if (!rc) {
    memset(l->q_stage, 0, 80 * sizeof(float));
    rc |= (int)hipMemset(l->q_seq_dev, 0, sizeof(unsigned));
  }
  if (rc) { delete l; return rc; }
  // slab gaps (alignment padding between tensors) are never written: keep them zero
  rc |= (int)hipMemset(l->slabs, 0, (size_t)cfg->ksplit * l->slab_stride * sizeof(float));
  rc |= (int)hipMemset(l->partials, 0, (size_t)dra_norm_partials_max() * sizeof(double));
  rc |= (int)hipMemset(l->prm_dev, 0, sizeof(dra_dqn_step_params));
  for (int k = 0; k <= K_COUNT; ++k) rc |= (int)hipEventCreate(&l->ev[k]);
  for (int k = 0; k < 8; ++k) rc |= (int)hipEventCreateWithFlags(&l->stage_ev[k], hipEventDisableTiming);
  for (int k = 0; k < 2; ++k) rc |= (int)hipEventCreateWithFlags(&l->ev_hq[k], hipEventDisableTiming);
  rc |= (int)hipStreamCreateWithFlags(&l->side, hipStreamNonBlocking);
  rc |= (int)hipEventCreateWithFlags(&l->ev_fork, hipEventDisableTiming);
  for (int k = 0; k < 4; ++k) rc |= (int)hipEventCreateWithFlags(&l->ev_join[k], hipEventDisableTiming);
  rc |= (int)hipEventCreateWithFlags(&l->ev_actor_done, hipEventDisableTiming);
  rc |= (int)hipEventCreateWithFlags(&l->ev_loss, hipEventDisableTiming);
  rc |= (int)hipEventCreateWithFlags(&l->ev_gather_done, hipEventDisableTiming);
  rc |= (int)hipEventCreateWithFlags(&l->ev_step_done, hipEventDisableTiming);
  rc |= (int)hipEventCreateWithFlags(&l->ev_flush, hipEventDisableTiming);
  for (int g = 0; g < 2; ++g) {
    rc |= (int)hipEventCreateWithFlags(&l->ev_mb_ready[g], hipEventDisableTiming);
    rc |= (int)hipEventCreateWithFlags(&l->ev_mb_free[g], hipEventDisableTiming);
  }
  if (rc) { delete l; return rc; }
  *out = l;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_destroy(dra_dqn_learner* l) {
  if (!l) return DRA_OK;
  (void)hipDeviceSynchronize();
  if (l->g_update_ready) (void)hipGraphExecDestroy(l->g_update);
  if (l->g_update_per_ready) (void)hipGraphExecDestroy(l->g_update_per);
  if (l->tr_ev) {
    for (int i = 0; i < l->tr_cap * 5; ++i) (void)hipEventDestroy(l->tr_ev[i]);
    delete[] l->tr_ev;
  }
  for (int g = 0; g < 2; ++g) {
    if (l->g_ag_ready[g]) (void)hipGraphExecDestroy(l->g_ag[g]);
    for (int h = g; h < 4; h += 2) {
      if (l->g_pipe_ready[h]) (void)hipGraphExecDestroy(l->g_pipe[h]);
      if (l->g_pipe_per_ready[h]) {
        (void)hipGraphExecDestroy(l->g_pipe_per[h]);
        if (l->g_pipe_per_b[h]) (void)hipGraphExecDestroy(l->g_pipe_per_b[h]);
      }
    }
    (void)hipEventDestroy(l->ev_mb_ready[g]); (void)hipEventDestroy(l->ev_mb_free[g]);
    void* mb[] = {l->state_[g], l->next_state_[g], l->action_[g], l->reward_[g], l->mask_[g]};
    for (void* b : mb) if (b) (void)hipFree(b);
  }
  for (auto& ga : l->g_actor) if (ga.ready) (void)hipGraphExecDestroy(ga.exec);
  for (int k = 0; k < 4; ++k) if (l->pa[k]) (void)hipFree(l->pa[k]);
  if (l->ah4) (void)hipFree(l->ah4);
  if (l->defer_dev) (void)hipFree(l->defer_dev);
  if (l->aflags) (void)hipFree(l->aflags);
  if (l->all_dev) (void)hipFree(l->all_dev);
  if (l->fchain_dev) (void)hipFree(l->fchain_dev);
  if (l->hchain_dev) (void)hipFree(l->hchain_dev);
  for (int k = 0; k < 2; ++k) {
    void* ab[] = {l->ah_y1[k], l->ah_y2[k], l->ah_y3[k], l->ah_slabs[k], l->ah_chain[k]};
    for (void* b : ab) if (b) (void)hipFree(b);
  }
  for (int k = 0; k < 8; ++k) if (l->ah_idx_pin[k]) (void)hipHostFree(l->ah_idx_pin[k]);
  if (l->ah_idx_copy) (void)hipFree(l->ah_idx_copy);
  if (l->ah_done) (void)hipFree(l->ah_done);
  if (l->bchain_dev) (void)hipFree(l->bchain_dev);
  if (l->fs_count) (void)hipFree(l->fs_count);
  if (l->fs_host) (void)hipHostFree(l->fs_host);
  if (l->ev_fs) (void)hipEventDestroy(l->ev_fs);
  if (l->aring_dev) {
    (void)hipFree(l->aring_dev); (void)hipHostFree(l->aring_stage); (void)hipFree(l->aring_seq);
    (void)hipFree(l->pend_frame); (void)hipFree(l->pend_reward); (void)hipFree(l->pend_mask);
  }
  for (int g = 0; g < 4; ++g) if (l->g_aring_ready[g]) (void)hipGraphExecDestroy(l->g_aring[g]);
  if (l->aprm_ring) {
    (void)hipHostFree(l->aprm_ring); (void)hipFree(l->aprm_seq_dev); (void)hipFree(l->fc4_ticket);
    for (int k = 0; k < kAprmSlots; ++k) (void)hipEventDestroy(l->aprm_ev[k]);
  }
  void* bufs[] = {l->act_state, l->idx, l->h4, l->ay1, l->ay2,
                  l->ay3, l->aq, l->dq, l->dh4, l->dy3, l->dy2, l->dy1, l->delta, l->prio, l->weights, l->samp_prob,
                  l->slabs, l->fc4_slabs, l->afc4_slabs, l->lin_ws, l->partials, l->loss, l->norm, l->prm_dev};
  for (void* b : bufs) if (b) (void)hipFree(b);
  void* hb[] = {l->atoms, l->qr_ws, l->alog, l->opt_step, l->ay2p, l->ay3p};
  for (void* b : hb) if (b) (void)hipFree(b);
  for (int k = 0; k < 3; ++k) if (l->lslabs[k]) (void)hipFree(l->lslabs[k]);
  for (int z = 0; z < 3; ++z) {
    if (l->y1[z]) (void)hipFree(l->y1[z]);
    if (l->y2[z]) (void)hipFree(l->y2[z]);
    if (l->y3[z]) (void)hipFree(l->y3[z]);
    if (l->q[z]) (void)hipFree(l->q[z]);
  }
  (void)hipHostFree(l->prm_stage);
  (void)hipHostFree(l->idx_stage);
  for (int k = 0; k < 4; ++k) {
    if (l->idx_pin[k]) (void)hipHostFree(l->idx_pin[k]);
    if (l->g_rd_ready[k]) (void)hipGraphExecDestroy(l->g_rd[k]);
    if (l->g_rd_per_ready[k]) {
      (void)hipGraphExecDestroy(l->g_rd_per[k]);
      if (l->g_rd_per_b[k]) (void)hipGraphExecDestroy(l->g_rd_per_b[k]);   // (null with the one-graph prioritized update)
    }
    (void)hipEventDestroy(l->ev_upd[k]);
  }
  for (int k = 0; k < 4; ++k) if (l->idx_tag_pin[k]) (void)hipHostFree(l->idx_tag_pin[k]);
  if (l->idx_tag_dev) (void)hipFree(l->idx_tag_dev);
  if (l->per2_dev) (void)hipFree(l->per2_dev);
  if (l->per2_idx) (void)hipFree(l->per2_idx);
  if (l->rd_seq_dev) (void)hipFree(l->rd_seq_dev);
  if (l->sp_stage) (void)hipHostFree(l->sp_stage);
  for (int k = 0; k < 8; ++k) if (l->sp_ev[k]) (void)hipEventDestroy(l->sp_ev[k]);
  if (l->ui_stage) (void)hipHostFree(l->ui_stage);
  for (int k = 0; k < 8; ++k) if (l->ui_ev[k]) (void)hipEventDestroy(l->ui_ev[k]);
  if (l->timeout_flag) (void)hipHostFree(l->timeout_flag);
  if (l->qs_stage) (void)hipHostFree(l->qs_stage);
  if (l->q_stage) (void)hipHostFree(l->q_stage);
  if (l->q_seq_dev) (void)hipFree(l->q_seq_dev);
  if (l->g_q_ready) (void)hipGraphExecDestroy(l->g_q);
  for (int k = 0; k < 2; ++k) {
    if (l->g_qa_ready[k]) (void)hipGraphExecDestroy(l->g_qa[k]);
    if (l->ev_hq[k]) (void)hipEventDestroy(l->ev_hq[k]);
  }
  for (int k = 0; k <= K_COUNT; ++k) (void)hipEventDestroy(l->ev[k]);
  for (int k = 0; k < 8; ++k) (void)hipEventDestroy(l->stage_ev[k]);
  (void)hipStreamDestroy(l->side);
  (void)hipEventDestroy(l->ev_fork);
  for (int k = 0; k < 4; ++k) (void)hipEventDestroy(l->ev_join[k]);
  (void)hipEventDestroy(l->ev_actor_done); (void)hipEventDestroy(l->ev_gather_done); (void)hipEventDestroy(l->ev_step_done); (void)hipEventDestroy(l->ev_flush);
  (void)hipEventDestroy(l->ev_loss);
  delete l;
  return DRA_OK;
}

// ---- true resume (BaseAgent.py:24-33 saves weights only): the learner-internal state a bit-exact
// continuation needs on top of what the host owns (parameters, target, optimizer state, the replay ring).  Call with the
// learner synchronised, at a step boundary.
//   dra_dqn_learner_resume_buffer(i)   enumerates the device buffers: the optimizer step count, the actor's rotating
//                                      parameter copies, the actor parameter-block ring + its device counter, the pending
//                                      observation, the ring-direct update counter (DRA_EINVAL past the last one);
//   dra_dqn_learner_resume_counters    the host-side counters of the pipelines (restore != 0: into a FRESH learner of the
//                                      same configuration, before its first step; all of its events are then unrecorded,
//                                      which is correct -- nothing is in flight after a load).
static const struct { const char* name; } kResumeNames[] = {{"opt_step"}, {"pa0"}, {"pa1"}, {"pa2"}, {"pa3"}, {"aring_dev"}, {"aring_seq"},
                                                            {"pend_frame"}, {"pend_reward"}, {"pend_mask"}, {"rd_seq_dev"}, {"prm_dev"}};
DRA_API int dra_dqn_learner_resume_buffer_count(void) { return (int)(sizeof(kResumeNames) / sizeof(kResumeNames[0])); }

DRA_API int dra_dqn_learner_resume_buffer(dra_dqn_learner* l, int index, void** ptr, int64_t* bytes, char* name, int name_len) {
  if (!l || !ptr || !bytes || index < 0 || index >= (int)(sizeof(kResumeNames) / sizeof(kResumeNames[0]))) return DRA_EINVAL;
  if (int rcl = fs_leave_any(l)) return rcl;
  if (l->defer_host) return DRA_EINVAL;   // (DRA_VAR_DEFER_FC4: dra_dqn_learner_flush + a synchronise first -- a pending fc4 segment is not a state to save)
  void* p = nullptr;
  int64_t n = 0;
  const int64_t pbytes = (int64_t)l->c.n_params * (int64_t)sizeof(float);
  switch (index) {
    case 0: p = l->opt_step; n = sizeof(int64_t); break;
    case 1: case 2: case 3: case 4: p = l->pa[index - 1]; n = pbytes; break;
    case 5: p = l->aring_dev; n = (int64_t)kAringSlots * (int64_t)kAprmStride; break;
    case 6: p = l->aring_seq; n = sizeof(unsigned); break;
    case 7: p = l->pend_frame; n = 7056; break;
    case 8: p = l->pend_reward; n = sizeof(double); break;
    case 9: p = l->pend_mask; n = sizeof(int32_t); break;
    case 10: p = l->rd_seq_dev; n = sizeof(unsigned long long); break;
    case 11: p = l->prm_dev; n = sizeof(dra_dqn_step_params); break;
  }
  *ptr = p;
  *bytes = p ? n : 0;      // a buffer this configuration does not have: null / 0
  if (name && name_len > 0) { strncpy(name, kResumeNames[index].name, (size_t)name_len - 1); name[name_len - 1] = 0; }
  return DRA_OK;
}

DRA_API int dra_dqn_learner_resume_counters(dra_dqn_learner* l, int64_t* io, int n, int restore) {
  if (!l || !io || n < 16) return DRA_EINVAL;
  if (int rcl = fs_leave_any(l)) return rcl;
  if (l->defer_host) return DRA_EINVAL;   // (DRA_VAR_DEFER_FC4: dra_dqn_learner_flush + a synchronise first -- a pending fc4 segment is not a state to save)
  if (!restore) {
    memset(io, 0, (size_t)n * sizeof(int64_t));
    io[0] = l->step_no; io[1] = l->pa_cur; io[2] = l->pa_valid ? 1 : 0; io[3] = (int64_t)l->aring_pushed;
    io[4] = (int64_t)l->aring_issued; io[5] = l->aring_primed ? 1 : 0; io[6] = (int64_t)l->rd_issued; io[7] = l->actor_pending ? 1 : 0;
    io[8] = l->stage_k; io[9] = l->gb; io[10] = l->last_gb; io[11] = (int64_t)l->aprm_seq; io[12] = l->variant;
    io[13] = l->c.n_params; io[14] = l->ag_have_prev ? 1 : 0; io[15] = l->ag_prev_par;
    return DRA_OK;
  }
  if (io[12] != l->variant || io[13] != l->c.n_params) return DRA_EINVAL;   // another pipeline / another network
  if (l->step_no != 0 || l->aring_pushed != 0 || l->captured) return DRA_EINVAL;   // only into a fresh learner
  l->step_no = io[0]; l->pa_cur = (int)io[1]; l->pa_valid = io[2] != 0; l->aring_pushed = (uint64_t)io[3];
  l->aring_issued = (uint64_t)io[4]; l->aring_primed = io[5] != 0; l->rd_issued = (uint64_t)io[6]; l->actor_pending = false;
  l->stage_k = (int)io[8]; l->gb = (int)io[9]; l->last_gb = (int)io[10]; l->aprm_seq = (uint64_t)io[11];
  l->ag_have_prev = io[14] != 0; l->ag_prev_par = (int)io[15];
  // the host mirror of the actor parameter-block ring: the pipelined step reads the slots of the blocks it issues from it
  // (which ring slots an actor launch writes decides the cross-stream waits); the device ring was restored by the caller
  if (l->aring_dev && l->aring_stage)
    DRA_HIP(hipMemcpy(l->aring_stage, l->aring_dev, (size_t)kAringSlots * kAprmStride, hipMemcpyDeviceToHost));
  return DRA_OK;
}

// Device pointers the host fills / reads: idx (int64[B], written before every update),
// sampling_prob (f32[B], PER only), and read-only results.
DRA_API int dra_dqn_learner_buffers(dra_dqn_learner* l, void** idx, void** sampling_prob, void** loss, void** norm,
                                    void** q, void** delta, void** prio, void** actor_q) {
  if (!l) return DRA_EINVAL;
  if (idx) *idx = l->idx;
  if (sampling_prob) *sampling_prob = l->samp_prob;
  if (loss) *loss = l->loss;
  if (norm) *norm = l->norm;
  if (q) *q = l->q[0];
  if (delta) *delta = l->delta;
  if (prio) *prio = l->prio;
  if (actor_q) *actor_q = l->aq;
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------------
// Batch-1 action values of the distributional heads (the actor side of CategoricalDQN_agent.py:21-24 and
// QuantileRegressionDQN_agent.py:17-20).  out[o] = bh[o] + <h4, Wh[o]> for the A*N head outputs, one wave per output
// (a 2 KB weight row is one coalesced read per lane group; 4 outputs in flight per wave), kept in LDS; then one wave
// per action:  categorical  q[a] = sum_n softmax(out[a])_n * atoms[n],   quantile  q[a] = mean_n out[a][n].
// where head_fused_kernel finds the transition scalars when the update reads the replay ring directly (idx == null:
// from the gathered minibatch)
struct RingScalars {
  const int64_t* idx;          // [B] sampled slots (device-visible)
  const uint8_t* actions;      // ring arrays
  const double* rewards;
  const int32_t* masks;
  int n_step;
  double discount;
  int64_t* out_action; float* out_reward; float* out_mask;   // the learner's minibatch scalar buffers, filled on the way
  // DRA_VAR_IDX_PREFETCH: workgroup 0 counts this update as done (conv1 of the next update compares the tags of its
  // prefetched indices with the count: ConvV2Args::sample_idx_tagged)
  unsigned long long* seq;
  // DRA_VAR_FWD_CHAIN: workgroup 0 counts the forward chain of this update as done (conv_v2.hip FwdChainArgs::epoch)
  unsigned* chain_epoch;
  // DRA_VAR_TARGET_AHEAD: the target net's split-K partial sums [KS][B][512] come from this stash (written by launches on another
  // stream: complete once *z1_done >= z1_want; read with agent-scope loads) instead of slabs' z = 1 block.  Null: both from `slabs`.
  const float* z1_slabs;
  const unsigned long long* z1_done;
  unsigned long long z1_want;
  int* z1_timeout;
};

// action / n-step reward / mask of sampled transition b straight from the replay ring, folded as ring_gather_kernel does
// (replay.py:133-139, fp64, the reference's association), then f32 as tensor() would
__device__ __forceinline__ void ring_scalars_of(const RingScalars& rs, int b, int64_t* ab, float* rew_b, float* mask_b) {
  const int64_t i = rs.idx[b];
  *ab = *reinterpret_cast<const int64_t*>(rs.actions + i * 8);
  double cum_r = 0.0;
  int32_t cum_m = 1;
  for (int k = rs.n_step - 1; k >= 0; --k) {
    const int32_t m = rs.masks[i + k];
    cum_r = __dadd_rn(rs.rewards[i + k], __dmul_rn(__dmul_rn((double)m, rs.discount), cum_r));
    cum_m = cum_m ? m : cum_m;
  }
  *rew_b = (float)cum_r;
  *mask_b = (float)cum_m;
}

struct HeadSpec {
  int kind, n_atoms;
  const float* atoms;
  float* out;          // optional global copy of the A*N head outputs (tests)
  const float* pre;    // optional: the A*N head outputs, already computed (actor_dist_gemv_kernel)
};

static HeadSpec head_spec(const dra_dqn_learner* l) {
  HeadSpec hs;
  hs.kind = l->c.head_kind; hs.n_atoms = l->c.n_atoms; hs.atoms = l->atoms; hs.out = l->alog; hs.pre = nullptr;
  return hs;
}

// out[o] = bh[o] + <h4, Wh[o]> for the A*N outputs of a distributional head at batch 1: one wave per output (its 2 KB weight
// row = 8 floats per lane, all requested at once), same per-lane products and butterfly as dist_head_q: bit-identical.
__global__ void __launch_bounds__(256)
actor_dist_gemv_kernel(const float* __restrict__ h4, const float* __restrict__ wh, const float* __restrict__ bh, int NO,
                       float* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + wave;
  const float* row = wh + (int64_t)min(o, NO - 1) * 512;
  float hv[8], wv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { wv[i] = row[lane + 64 * i]; hv[i] = h4[lane + 64 * i]; }
  const float bias = bh[min(o, NO - 1)];
  float part = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) part += hv[i] * wv[i];
  part = wave_sum(part);
  if (lane == 0 && o < NO) out[o] = part + bias;
}

// hs.pre != null: the A*N head outputs were already formed by actor_dist_gemv_kernel (one wave per output over many CUs: the
// in-workgroup loop below is 4 / 13 dependent passes for C51 / QR-DQN at 16 waves -- 12 / 34 us of the actor's env step,
// profiles/r02zu_kernel_stats_*); they are only copied in.
__device__ __forceinline__ void dist_head_q(const float* __restrict__ h4, const float* __restrict__ wh,
                                            const float* __restrict__ bh, int A, const HeadSpec hs,
                                            float* __restrict__ s_out, float* __restrict__ s_q) {
  const int nw = (int)(blockDim.x >> 6), wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = hs.n_atoms, NO = A * N;
  if (hs.pre) {
    for (int o = threadIdx.x; o < NO; o += blockDim.x) s_out[o] = hs.pre[o];
    __syncthreads();
  }
  float hv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) hv[i] = h4[lane + 64 * i];
  for (int o0 = wave * 4; o0 < NO && !hs.pre; o0 += nw * 4) {
    float wv[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* row = wh + (int64_t)min(o0 + u, NO - 1) * 512;
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[u][i] = row[lane + 64 * i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part += hv[i] * wv[u][i];
      part = wave_sum(part);
      if (lane == 0 && o0 + u < NO) s_out[o0 + u] = part + bh[o0 + u];
    }
  }
  __syncthreads();
  if (hs.out) for (int o = threadIdx.x; o < NO; o += blockDim.x) hs.out[o] = s_out[o];
  for (int a = wave; a < A; a += nw) {
    const float q = dist_action_value(s_out + a * N, N, hs.kind, hs.atoms, lane);
    if (lane == 0) s_q[a] = q;
  }
  __syncthreads();
}

// Input gradient of a distributional head (CategoricalNet / QuantileNet over 512 features) at the update's batch:
//   dh4[b][i] = relu'(h4[b][i]) * sum_n dq[b][a_b * N + n] * Wh[a_b * N + n][i]
// The loss differentiates the taken action's N outputs only (dq is zero elsewhere: the loss kernels write the zeros), so the
// [B, A*N] x [A*N, 512] contraction the K-chunked GEMM performed (17 us for C51, 24 us for QR-DQN at 4 actions) is a GATHERED
// matrix-vector product per sample.  Workgroup = (sample, 128-column block): threads (r, c) = (t >> 7, t & 127) walk the rows
// n = r, r + 2, ... of the sample's block as coalesced 512-byte row segments, eight in flight; the two row classes meet in
// LDS as (even rows) + (odd rows).  One multiply and one add per term (no contraction).
__global__ void __launch_bounds__(256)
dist_head_dgrad_kernel(const float* __restrict__ dq, const float* __restrict__ wh, const float* __restrict__ h4,
                       const int64_t* __restrict__ action, int N, int NO, float* __restrict__ dh4) {
  __shared__ float s_part[128];
  const int b = blockIdx.x, t = threadIdx.x, r = t >> 7, col = blockIdx.y * 128 + (t & 127);
  const int64_t act_b = action[b];
  const int64_t base = (act_b < 0 ? 0 : (act_b >= NO / N ? NO / N - 1 : act_b)) * (int64_t)N;   // (clamped like head_fused_kernel's read)
  const float* __restrict__ d = dq + (int64_t)b * NO + base;
  const float* __restrict__ w = wh + base * 512 + col;
  const float x = h4[(int64_t)b * 512 + col];
  float acc = 0.f;
  int n = r;
  for (; n + 14 < N; n += 16) {
    float dv[8], wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { dv[u] = d[n + 2 * u]; wv[u] = w[(int64_t)(n + 2 * u) * 512]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += dv[u] * wv[u];
  }
  for (; n < N; n += 2) acc += d[n] * w[(int64_t)n * 512];
  if (r == 1) s_part[t & 127] = acc;
  __syncthreads();
  if (r == 0) dh4[(int64_t)b * 512 + col] = x > 0.f ? acc + s_part[t] : 0.f;     // (ReLU: h4 is the post-activation value)
}

// h4[z][b][:] = relu(b4_z + sum_s fc4_slab[z][s][b][:]) for every net of the update (z = 0 online(states), 1 target(next),
// 2 online(next)): the split-K reduction head_fused_kernel performs for the VanillaNet head, on its own for the
// distributional heads (their outputs are a [B,512] x [512, A*N] contraction -> linear kernel).  grid (B, nz).
template <int KS>
__global__ void __launch_bounds__(256)
fc4_reduce_kernel(const float* __restrict__ slabs, int B, const float* __restrict__ b4_on, const float* __restrict__ b4_tg,
                  float* __restrict__ h4, int64_t* __restrict__ opt_step, const RingScalars rs) {
  const int b = blockIdx.x, z = blockIdx.y, tid = threadIdx.x;
  if (rs.idx && z == 0 && tid == 0) {
    // DRA_VAR_RING_DIRECT with a distributional head: the loss kernel (later in this stream) reads the minibatch's
    // action / n-step reward / mask buffers -- filled here from the ring, as head_fused_kernel does for VanillaNet
    int64_t ab;
    float rew_b, mask_b;
    ring_scalars_of(rs, b, &ab, &rew_b, &mask_b);
    rs.out_action[b] = ab; rs.out_reward[b] = rew_b; rs.out_mask[b] = mask_b;
    if (rs.seq && b == 0) *rs.seq += 1ull;
  }
  const float* bias = (z == 1) ? b4_tg : b4_on;
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int k = tid + 256 * rep;
    const float* s = slabs + ((int64_t)z * KS * B + b) * 512 + k;
    float part[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) part[i] = s[(int64_t)i * B * 512];
    float v = part[0];
#pragma unroll
    for (int i = 1; i < KS; ++i) v += part[i];
    v += bias[k];
    h4[((int64_t)z * B + b) * 512 + k] = v > 0.f ? v : 0.f;
  }
  if (opt_step && b == 0 && z == 0 && tid == 0) *opt_step += 1;   // one optimizer step per update (Adam's t)
}

// ------------------------------------------------------------------------------------------------
// head_fused_kernel: one workgroup per sample.
//   h4[z][b][:] = relu(b4_z + sum_s fc4_slab[z][s][b][:])          (split-K reduction of fc4, all nets)
//   q[z][b][a]  = bh_z[a] + <h4[z][b], Wh_z[a]>                     (VanillaNet head, network_heads.py:18-21)
//   delta_b     = (r + (g*qnext)*m) - q[0][b][a_b]                  (DQN_agent.py:85-99; uniform replay)
//   dq[b][:]    = -delta_b / B at a_b, 0 elsewhere                  (d mean(0.5 delta^2) / dq)
//   dh4[b][k]   = dq[b][a_b] * Wh_0[a_b][k] * (h4[0][b][k] > 0)     (gradient w.r.t. fc4's pre-activation)
// Per-sample work only: the batch-mean loss is recovered from `delta` on demand, and the PER
// variant (needs max over the batch) keeps the separate td_loss kernel.
// (the body: one workgroup = one sample b.  COUT (DRA_VAR_HEAD_CHAIN): h4 / dq / dh4 go to workgroups of the SAME launch -- fc4's and
// the head's backward roles -- agent-scope stores, then the workgroup counts itself on `done`)
template <int KS, bool COUT>
__device__ __forceinline__ void head_fused_body(const float* __restrict__ slabs, int nz, int B, int A, const float* __restrict__ b4_on,
                  const float* __restrict__ b4_tg, const float* __restrict__ wh_on, const float* __restrict__ wh_tg,
                  const float* __restrict__ bh_on, const float* __restrict__ bh_tg, const int64_t* __restrict__ action,
                  const float* __restrict__ reward, const float* __restrict__ mask, float gamma_n, int double_q,
                  float* __restrict__ h4_out, float* __restrict__ q_on, float* __restrict__ q_tg, float* __restrict__ q_on2,
                  float* __restrict__ delta, float* __restrict__ dq, float* __restrict__ dh4, int64_t* __restrict__ opt_step,
                  const RingScalars rs, const float* __restrict__ per_w, const int b, float (*s_h)[512], float (*s_q)[64], unsigned* done) {
  const int tid = threadIdx.x;
  DRA_STAMP(TR_HEAD, 0);
  // everything this workgroup reads is requested up front: the head weights of the (net, action) pairs this wave owns, the
  // transition scalars, and then the split-K partials -- ONE exposed memory latency instead of three (phase trace r02a:
  // 1.9 us partials, 1.9 us head, 1.4 us epilogue)
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int MAXP = 4;                                   // pairs per wave held in registers (nz * A <= 16)
  float whr[MAXP][8], bhr[MAXP];
#pragma unroll
  for (int u = 0; u < MAXP; ++u) {
    const int pair = min(wave + 4 * u, nz * A - 1);
    const int z = pair / A, a = pair - z * A;
    const float* wh = ((z == 1) ? wh_tg : wh_on) + a * 512;
#pragma unroll
    for (int i = 0; i < 8; ++i) whr[u][i] = wh[lane + 64 * i];
    bhr[u] = ((z == 1) ? bh_tg : bh_on)[a];
  }
  int64_t ab;
  float rew_b, mask_b;
  if (rs.idx) {
    // DRA_VAR_RING_DIRECT: the transition scalars straight from the replay ring
    ring_scalars_of(rs, b, &ab, &rew_b, &mask_b);
    if (tid == 0) { rs.out_action[b] = ab; rs.out_reward[b] = rew_b; rs.out_mask[b] = mask_b; }   // (checkers, PER loss kernel)
  } else {
    ab = action[b];
    rew_b = reward[b];
    mask_b = mask[b];
  }
  float dwh[2];                                             // wh_on[ab][k] for this thread's two k (dL/dh4 below)
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) dwh[rep] = wh_on[(int)min(max(ab, (int64_t)0), (int64_t)(A - 1)) * 512 + tid + 256 * rep];
  if (rs.z1_slabs) {
    // (expected to hold already: the target's launches were issued one update ago and started when the previous update did)
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(rs.z1_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < rs.z1_want) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > kMegaWaitTicks) {
          if (rs.z1_timeout) __hip_atomic_store(rs.z1_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    __syncthreads();
  }
  for (int z = 0; z < nz; ++z) {
    const float* bias = (z == 1) ? b4_tg : b4_on;
    const bool stash = z == 1 && rs.z1_slabs;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int k = tid + 256 * rep;
      const float* s = stash ? rs.z1_slabs + (int64_t)b * 512 + k : slabs + ((int64_t)z * KS * B + b) * 512 + k;
      float part[KS];
      if (stash) {
#pragma unroll
        for (int i = 0; i < KS; ++i) part[i] = mega_ld<true>(s + (int64_t)i * B * 512);
      } else {
#pragma unroll
        for (int i = 0; i < KS; ++i) part[i] = s[(int64_t)i * B * 512];  // all split-K partials in flight at once
      }
      float v = part[0];
#pragma unroll
      for (int i = 1; i < KS; ++i) v += part[i];
      v += bias[k];
      v = v > 0.f ? v : 0.f;
      s_h[z][k] = v;
      if (z == 0) mega_st<COUT>(&h4_out[(int64_t)b * 512 + k], v);
    }
  }
  __syncthreads();
  DRA_STAMP(TR_HEAD, 2);
  // heads: wave w owns the (net, action) pairs w, w+4, ... -- one wave-level dot product each, no
  // workgroup barrier per output
  if (nz * A <= 4 * MAXP) {
#pragma unroll
    for (int u = 0; u < MAXP; ++u) {
      const int pair = wave + 4 * u;
      if (pair < nz * A) {
        const int z = pair / A, a = pair - z * A;
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) part += s_h[z][lane + 64 * i] * whr[u][i];
        part = wave_sum(part);
        if (lane == 0) s_q[z][a] = part + bhr[u];
      }
    }
  } else {
    for (int pair = wave; pair < nz * A; pair += 4) {
      const int z = pair / A, a = pair - z * A;
      const float* wh = ((z == 1) ? wh_tg : wh_on) + a * 512;
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part += s_h[z][lane + 64 * i] * wh[lane + 64 * i];
      part = wave_sum(part);
      if (lane == 0) s_q[z][a] = part + ((z == 1) ? bh_tg : bh_on)[a];
    }
  }
  __syncthreads();
  DRA_STAMP(TR_HEAD, 4);
  float dqa;
  {
    float qn;
    if (double_q) {
      int best = 0;
      float bv = s_q[2][0];
      for (int k = 1; k < A; ++k) if (s_q[2][k] > bv) { bv = s_q[2][k]; best = k; }
      qn = s_q[1][best];
    } else {
      qn = s_q[1][0];
      for (int k = 1; k < A; ++k) qn = fmaxf(qn, s_q[1][k]);
    }
    const float target = rew_b + (gamma_n * qn) * mask_b;
    const float d = target - s_q[0][ab];
    if (per_w) {
      // PrioritizedReplay with the importance weights already known (the previous update's chain kernel normalised them:
      // sumtree.hip dra_sumtree_per_chain2): d mean(0.5 (delta w)^2) / dq, in td_loss_kernel's order of operations
      const float w = per_w[b];
      const float lw = d * w;
      dqa = -(lw * w) / (float)B;
    } else {
      dqa = -d / (float)B;
    }
    if (tid == 0) delta[b] = d;
  }
  if (tid < A) {
    q_on[(int64_t)b * A + tid] = s_q[0][tid];
    q_tg[(int64_t)b * A + tid] = s_q[1][tid];
    if (nz > 2) q_on2[(int64_t)b * A + tid] = s_q[2][tid];
    mega_st<COUT>(&dq[(int64_t)b * A + tid], (tid == ab) ? dqa : 0.f);
  }
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int k = tid + 256 * rep;
    mega_st<COUT>(&dh4[(int64_t)b * 512 + k], s_h[0][k] > 0.f ? dqa * dwh[rep] : 0.f);
  }
  if (opt_step && b == 0 && tid == 0) *opt_step += 1;   // one optimizer step per update (Adam's t)
  if (rs.seq && b == 0 && tid == 0) *rs.seq += 1ull;
  if (rs.chain_epoch && b == 0 && tid == 0) *rs.chain_epoch += 1u;
  DRA_STAMP(TR_HEAD, 5);
  DRA_STAMP_END(TR_HEAD);
  if constexpr (COUT) {
    MegaSync ms;
    ms.done = done;
    mega_publish(ms);
  }
}

template <int KS>
__global__ void __launch_bounds__(256)
head_fused_kernel(const float* __restrict__ slabs, int nz, int B, int A, const float* __restrict__ b4_on,
                  const float* __restrict__ b4_tg, const float* __restrict__ wh_on, const float* __restrict__ wh_tg,
                  const float* __restrict__ bh_on, const float* __restrict__ bh_tg, const int64_t* __restrict__ action,
                  const float* __restrict__ reward, const float* __restrict__ mask, float gamma_n, int double_q,
                  float* __restrict__ h4_out, float* __restrict__ q_on, float* __restrict__ q_tg, float* __restrict__ q_on2,
                  float* __restrict__ delta, float* __restrict__ dq, float* __restrict__ dh4, int64_t* __restrict__ opt_step,
                  const RingScalars rs, const float* __restrict__ per_w, const float* __restrict__ pf_w4) {
  __shared__ float s_h[3][512];
  __shared__ float s_q[3][64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= B) {
    // Spare workgroups of this 32-workgroup launch prefetch what the NEXT launch's input-gradient role will stream: workgroup
    // B + p pulls W4[0:512][32p : 32p + 32] (512 row segments of 128 B) -- the operand of LinDgradOne workgroup p, which runs on
    // the same XCD (workgroups are dealt round-robin over the 8 XCDs and B is a multiple of 8) -- into that XCD's L2.
    const int pcol = (b - B) * 32;
    float4 v[16];
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) {
      const int e = tid + 256 * qq, row = e >> 3, c4 = e & 7;
      v[qq] = *reinterpret_cast<const float4*>(pf_w4 + (int64_t)row * 3136 + pcol + 4 * c4);
    }
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) asm volatile("" :: "v"(v[qq].x), "v"(v[qq].y), "v"(v[qq].z), "v"(v[qq].w));
    return;
  }
  head_fused_body<KS, false>(slabs, nz, B, A, b4_on, b4_tg, wh_on, wh_tg, bh_on, bh_tg, action, reward, mask, gamma_n, double_q, h4_out, q_on, q_tg, q_on2, delta, dq, dh4, opt_step, rs, per_w, b, s_h, s_q, nullptr);
}

// DRA_VAR_HEAD_CHAIN: the head launch and fc4's / the head's backward launch as ONE launch in dependency order --
//   [head role: B workgroups] [fc4 input gradient] [fc4 weight gradient] [head weight gradient]
// the backward roles request what does not depend on the head (fc4's weights, conv3's activations) FIRST, wait on one arrival
// counter for the B head workgroups, then read dh4 / dq / h4 with agent-scope loads.  The counter is zeroed by the first
// workgroup of the NEXT update's forward chain.  Same arithmetic in the same order as the two launches: bit-identical.
// DQN_agent.py:85-99 (loss) + the first two nodes of loss.backward() (DQN_agent.py:131).
struct HeadArgs {
  const float* slabs; int nz, B, A;
  const float *b4_on, *b4_tg, *wh_on, *wh_tg, *bh_on, *bh_tg;
  const int64_t* action; const float *reward, *mask; float gamma_n; int double_q;
  float *h4_out, *q_on, *q_tg, *q_on2, *delta, *dq, *dh4; int64_t* opt_step;
  RingScalars rs; const float* per_w; unsigned* done;
};
template <int KS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
head_fc_bwd_kernel(const HeadArgs ha, const LinDgradOne<512> rd, const LinWgradOne<8> rl, const HeadWgradRole rh, const int nd, const int nw) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  int b = blockIdx.x;
  if (b < ha.B) {
    float (*s_h)[512] = reinterpret_cast<float (*)[512]>(dyn_lds);
    float (*s_q)[64] = reinterpret_cast<float (*)[64]>(dyn_lds + 3 * 512);
    head_fused_body<KS, true>(ha.slabs, ha.nz, ha.B, ha.A, ha.b4_on, ha.b4_tg, ha.wh_on, ha.wh_tg, ha.bh_on, ha.bh_tg, ha.action, ha.reward,
                              ha.mask, ha.gamma_n, ha.double_q, ha.h4_out, ha.q_on, ha.q_tg, ha.q_on2, ha.delta, ha.dq, ha.dh4, ha.opt_step,
                              ha.rs, ha.per_w, b, s_h, s_q, ha.done);
    return;
  }
  b -= ha.B;
  if (b < nd) { rd.run_<false, true>(b, dyn_lds); return; }
  b -= nd;
  if (b < nw) { rl.run_<true>(b, dyn_lds); return; }
  rh.run_<true>(b - nw, dyn_lds);
}

// dWh[a][k] = sum_b dq[b][a] * h4[b][k] ;  dbh[a] = sum_b dq[b][a]      (grid = A, 512 threads)
__global__ void __launch_bounds__(64)
head_wgrad_kernel(const float* __restrict__ dq, const float* __restrict__ h4, int B, int A, float* __restrict__ dwh,
                  float* __restrict__ dbh) {
  const int a = blockIdx.x, k = blockIdx.y * 64 + threadIdx.x;  // grid (A, 8) x 64 threads
  float acc = 0.f, accb = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    float d[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = dq[(int64_t)(b + i) * A + a]; h[i] = h4[(int64_t)(b + i) * 512 + k]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc += d[i] * h[i]; accb += d[i]; }
  }
  for (; b < B; ++b) {
    const float d = dq[(int64_t)b * A + a];
    acc += d * h4[(int64_t)b * 512 + k];
    accb += d;
  }
  dwh[a * 512 + k] = acc;
  if (k == 0) dbh[a] = accb;
}

#define STEP(kid, expr)                                                        \
  do {                                                                         \
    if (l->only_kernel >= 0 && l->only_kernel != (kid)) break;                 \
    if (l->profiling) DRA_HIP(hipEventRecord(l->ev[kid], st));                 \
    int _rc = (expr);                                                          \
    if (_rc != DRA_OK) return _rc;                                             \
  } while (0)

// `idx`: device buffer, or (DRA_VAR_PINNED_IDX) the pinned staging slot itself -- the 160 gather workgroups
// read their 8-byte index over the host link (~1-2 us, inside the kernel) instead of waiting for a
// separate 4-5 us copy command on the update's critical path.
static int launch_gather(dra_dqn_learner* l, hipStream_t st, const int64_t* idx = nullptr) {
  l->last_gb = l->gb;
  return dra_ring_gather(l->ring, idx ? idx : l->idx, l->c.batch, l->state_[l->gb], l->next_state_[l->gb], l->action_[l->gb], nullptr, nullptr,
                         l->reward_[l->gb], l->mask_[l->gb], (void*)st);
}

// the three conv layers' slab segments of the flat gradient (one-pass weight gradients)
static void conv_fold_segs(const dra_dqn_learner* l, dra_fold_seg segs[3]) {
  const int wi[3] = {P_W1, P_W2, P_W3};
  for (int k = 0; k < 3; ++k) {
    segs[k].begin = l->c.offset[wi[k]]; segs[k].count = l->lstride[k]; segs[k].slabs = l->lslabs[k];
    segs[k].slab_stride = l->lstride[k]; segs[k].n_slabs = l->lnslabs[k]; segs[k].reserved = 0;
  }
}

// share of the rider blocks in conv1's forward launch (the rest rides in conv2's): conv1 is the longer launch
#ifndef DRA_EXP_RIDER_CONV1_PCT
#define DRA_EXP_RIDER_CONV1_PCT 57
#endif
constexpr int kRiderConv1Pct = DRA_EXP_RIDER_CONV1_PCT;
static int* defer_pending_word(const dra_dqn_learner* l) { return reinterpret_cast<int*>(l->defer_dev) + 1; }
static int* defer_valid_word(const dra_dqn_learner* l, int q) { return reinterpret_cast<int*>(l->defer_dev) + 2 + (q & 3); }

// defer_q >= 0 (DRA_VAR_DEFER_FC4): fc4's segment is left to the riders of the next graph; the launch marks copy defer_q incomplete
static int launch_optimizer(dra_dqn_learner* l, hipStream_t st, float* p_copy = nullptr, int defer_q = -1) {
  const dra_dqn_config& c = l->c;
  if (l->late) {
    dra_fold_seg segs[3];
    conv_fold_segs(l, segs);
    const bool adam = c.optimizer == DRA_OPT_ADAM;
    const float hyper[4] = {c.lr, adam ? c.beta1 : c.alpha, c.eps, adam ? c.beta2 : 0.f};
    if (defer_q >= 0)
      return dra_clip_step_late_defer(l->p, l->g, l->s1, l->s2, c.n_params, &segs[0], l->partials, l->late_nprior, l->timeout_flag,
                                      c.gradient_clip, hyper, c.centered, l->norm, p_copy, l->defer_begin, l->defer_count,
                                      l->defer_dev, defer_pending_word(l), defer_valid_word(l, defer_q), (void*)st);
    return dra_clip_step_late(l->p, l->g, l->s1, l->s2, c.n_params, &segs[0], l->partials, l->late_nprior,
                              l->timeout_flag, c.optimizer, c.gradient_clip, hyper, c.centered, l->opt_step, l->norm, p_copy,
                              (void*)st);
  }
  if (c.optimizer == DRA_OPT_ADAM)
    return dra_adam_step_counter(l->p, l->g, l->s1, l->s2, c.n_params, l->partials, l->n_partials, c.gradient_clip, c.lr,
                                 c.beta1, c.beta2, c.eps, l->opt_step, l->norm, p_copy, (void*)st);
  return dra_rmsprop_step_copy(l->p, l->g, l->s1, l->s2, c.n_params, l->partials, l->n_partials, c.gradient_clip,
                               c.lr, c.alpha, c.eps, c.centered, l->norm, p_copy, (void*)st);
}

static DraFc4Rider fc4_rider(const dra_dqn_learner* l, float* p_copy) {
  DraFc4Rider r;
  memset(&r, 0, sizeof(r));
  r.p = l->p; r.g = l->g; r.s1 = l->s1; r.s2 = l->s2; r.p_copy = p_copy;
  r.begin4 = l->defer_begin >> 2; r.count4 = l->defer_count >> 2;
  r.coef = l->defer_dev; r.pending = defer_pending_word(l);
  r.lr = l->c.lr; r.alpha = l->c.alpha; r.eps = l->c.eps; r.centered = l->c.centered;
  return r;
}

__global__ void __launch_bounds__(256) fc4_flush_kernel(const DraFc4Rider r) { fc4_rider_run(r, (int)blockIdx.x); }
__global__ void fc4_flush_done_kernel(int* pending, int* valid) {
  __hip_atomic_store(pending, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(valid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The deferred fc4 segment stepped NOW, on `st` (ordered behind the optimizer launch that left it: `st` waits for last_done):
// in front of everything that reads the parameters, the optimizer state or an actor copy outside the pipelined ring-direct graphs.
static int flush_fc4(dra_dqn_learner* l, hipStream_t st) {
  if (int rcl = fs_leave_any(l)) return rcl;     // (DRA_VAR_FLAG_SYNC: the event paths never meet un-recorded lane work)
  if (!l->defer_host) return DRA_OK;
  if (l->last_done) DRA_HIP(hipStreamWaitEvent(st, l->last_done, 0));
  const DraFc4Rider r = fc4_rider(l, l->pa[l->defer_q & 3]);
  hipLaunchKernelGGL(fc4_flush_kernel, dim3(fc4_rider_blocks(r.count4)), dim3(256), 0, st, r);
  DRA_LAUNCH_CHECK();
  hipLaunchKernelGGL(fc4_flush_done_kernel, dim3(1), dim3(1), 0, st, defer_pending_word(l), defer_valid_word(l, l->defer_q));
  DRA_LAUNCH_CHECK();
  DRA_HIP(hipEventRecord(l->ev_flush, st));     // whoever waits for "the last optimizer step" now waits for this
  l->last_done = l->ev_flush;
  l->defer_host = false;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_flush(dra_dqn_learner* l, void* stream) {
  if (!l) return DRA_EINVAL;
  return flush_fc4(l, dra_stream(stream));
}

__global__ void chain_epoch_bump_kernel(unsigned* epoch) { *epoch += 1u; }
__global__ void ah_done_kernel(unsigned long long* done, unsigned long long value) {
  __hip_atomic_store(done, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// first launch of an ahead sequence: holds its stream until the update stream has STARTED launch number `need` (fs_count, counted by
// the forward chain's first workgroup) -- the previous update is then complete: nothing reads this parity's stash, scratch or
// counters any more
__global__ void ah_gate_kernel(const unsigned long long* count, unsigned long long need, int* timeout_flag) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > 4 * kMegaWaitTicks) {
      __hip_atomic_store(timeout_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}

// DRA_VAR_TARGET_AHEAD: target(next_states) of ONE minibatch on `st` with parity set `par` -- conv1 + conv2 + conv3 as the chained
// launch over the target net alone (own counters; the one-thread launch behind it advances their epoch), and with_fc4: fc4's split-K
// partial sums into the stash + the completion word.  Same kernels, same per-sample arithmetic as net z = 1 of the update's own
// chain and fc4 launch: the stash holds the same bits.  DQN_agent.py:85-88 (q_next = target_network(next_states).detach()).
static int ah_target_launches(dra_dqn_learner* l, hipStream_t st, int par, const int64_t* idx_pinned, bool with_fc4,
                              unsigned long long done_value) {
  const dra_dqn_config& c = l->c;
  void *ring_frames = nullptr, *ra = nullptr, *rr = nullptr, *rm = nullptr;
  int ring_h = 4, ring_n = 1;
  int rc = dra_ring_pointers(l->ring, &ring_frames, &ra, &rr, &rm);
  if (!rc) rc = dra_ring_shape(l->ring, &ring_h, &ring_n);
  if (rc) return rc;
  if (ring_h != 4) return DRA_EINVAL;
  const float* T = l->pt;
  const int64_t* o = c.offset;
  const int64_t off[1] = {ring_n};
  const float* w1[1] = {T + o[P_W1]}; const float* b1[1] = {T + o[P_B1]};
  const float* w2[1] = {T + o[P_W2]}; const float* b2[1] = {T + o[P_B2]};
  const float* w3[1] = {T + o[P_W3]}; const float* b3[1] = {T + o[P_B3]};
  float* y1[1] = {l->ah_y1[par]}; float* y2[1] = {l->ah_y2[par]}; float* y3[1] = {l->ah_y3[par]};
  unsigned* cnt = l->ah_chain[par];
  rc = dra_conv_fwd_chain(ring_frames, idx_pinned, l->ah_idx_copy, nullptr, nullptr, off, 1, w1, b1, y1, w2, b2, y2, w3, b3, y3,
                          c.batch, c.u8_coef, cnt, cnt + kFwdChainCounters, l->timeout_flag, nullptr, nullptr, nullptr, nullptr, (void*)st);
  if (rc) return rc;
  hipLaunchKernelGGL(chain_epoch_bump_kernel, dim3(1), dim3(1), 0, st, cnt + kFwdChainCounters);
  DRA_LAUNCH_CHECK();
  if (with_fc4) {
    const float* x4[1] = {l->ah_y3[par]};
    const float* w4[1] = {T + o[P_W4]};
    rc = dra_linear_fwd_slabs_one(1, x4, w4, c.batch, 3136, 512, fc4_ks(l), l->ah_slabs[par], (void*)st);
    if (rc) return rc;
    hipLaunchKernelGGL(ah_done_kernel, dim3(1), dim3(1), 0, st, l->ah_done, done_value);
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

// Head + loss + head input-gradient for the distributional heads (everything head_fused_kernel does for VanillaNet):
//   h4[z] <- fc4 partial sums ; out[z] = h4[z] Wh_z^T + bh_z (q[z], [B][A*N]) ; fused loss kernel -> per-sample loss
//   vector (delta) and d(reduced loss)/d out (dq) ; dh4 = (dq Wh) * relu'(h4[0]).
static int run_dist_head(dra_dqn_learner* l, hipStream_t st, int per, float beta, const RingScalars& rs) {
  const dra_dqn_config& c = l->c;
  const int B = c.batch, A = c.n_actions, N = c.n_atoms, NO = l->n_out;
  const int nz = c.double_q ? 3 : 2;
  void* s = (void*)st;
  const float* P = l->p;
  const float* T = l->pt;
  const int64_t* o = c.offset;
  if (fc4_ks(l) == kFc4SplitWide)
    hipLaunchKernelGGL(fc4_reduce_kernel<kFc4SplitWide>, dim3(B, nz), dim3(256), 0, st, (const float*)l->fc4_slabs, B, P + o[P_B4],
                       T + o[P_B4], l->h4, l->opt_step, rs);
  else if (fc4_ks(l) == kFc4SplitMid)
    hipLaunchKernelGGL(fc4_reduce_kernel<kFc4SplitMid>, dim3(B, nz), dim3(256), 0, st, (const float*)l->fc4_slabs, B, P + o[P_B4],
                       T + o[P_B4], l->h4, l->opt_step, rs);
  else
    hipLaunchKernelGGL(fc4_reduce_kernel<kFc4Split>, dim3(B, nz), dim3(256), 0, st, (const float*)l->fc4_slabs, B, P + o[P_B4],
                       T + o[P_B4], l->h4, l->opt_step, rs);
  DRA_LAUNCH_CHECK();
  // the head contraction as the one-pass MFMA kernel (fused.hip dra_head_fwd_one); the wave-per-output GEMV and the K-chunked
  // GEMM it replaced were A/B partners behind DRA_HEAD_GEMV until round 6
  int rc = DRA_OK;
  {
    const float* hx[3] = {l->h4, l->h4 + (int64_t)B * 512, l->h4 + (int64_t)2 * B * 512};
    const float* hw[3] = {P + o[P_WH], T + o[P_WH], P + o[P_WH]};
    const float* hb[3] = {P + o[P_BH], T + o[P_BH], P + o[P_BH]};
    if ((rc = dra_head_fwd_one(nz, hx, hw, hb, l->q, B, NO, s))) return rc;
  }
  if (c.head_kind == DRA_HEAD_CATEGORICAL) {
    const float* weights = nullptr;
    if (per) {   // DQN_agent.py:124-126: importance weights scale the per-sample loss before the mean
      // (device-side prioritized draw: the previous update's chain kernel left them, and computes the priorities itself)
      if (!l->per2_active &&
          (rc = dra_per_weights(nullptr, l->samp_prob, B, beta, c.replay_eps, c.replay_alpha, nullptr, l->weights, s))) return rc;
      weights = l->weights;
    }
    rc = dra_c51_loss(l->q[0], l->q[1], c.double_q ? l->q[2] : nullptr, l->action_[l->gb], 1, l->reward_[l->gb], l->mask_[l->gb],
                      B, A, N, c.gamma_n, c.v_min, c.v_max, l->atoms, l->delta, l->dq, weights, s);
    if (rc) return rc;
    if (per && !l->per2_active &&
        (rc = dra_per_weights(l->delta, l->samp_prob, B, beta, c.replay_eps, c.replay_alpha, l->prio, nullptr, s))) return rc;
  } else {
    if (per) return DRA_EINVAL;   // not a valid reference configuration (QuantileRegressionDQN_agent.py: uniform replay only)
    rc = dra_qr_loss(l->q[0], l->q[1], l->action_[l->gb], 1, l->reward_[l->gb], l->mask_[l->gb], B, A, N, c.gamma_n, l->qr_ws,
                     l->delta, l->loss, l->dq, s);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(dist_head_dgrad_kernel, dim3(B, 4), dim3(256), 0, st, (const float*)l->dq, P + o[P_WH], (const float*)l->h4,
                     (const int64_t*)l->action_[l->gb], N, NO, l->dh4);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// forward + loss + backward + gradient norm (everything between the gather and the optimizer).
// `fork` != 0 places each layer's weight-gradient kernel on the side stream (captured as a parallel
// graph branch) while the input-gradient chain continues on `st`.
// `part`: 0 = everything; 1 = forward passes + loss only (ends when the TD errors / priorities exist); 2 = the rest
// (backward, gradient norm).  The PER pipeline captures the two halves as separate graphs with an event in between, so
// that the priority write-back and the next prioritized draw start under the backward pass.
static int run_body(dra_dqn_learner* l, hipStream_t st, int per, float beta, int fork, int part = 0) {
  const dra_dqn_config& c = l->c;
  l->captured = true;   // (the optimizer form is now fixed: dra_dqn_learner_set_update_cus)
  const int B = c.batch, A = c.n_actions;
  const int nz = c.double_q ? 3 : 2;
  void* s = (void*)st;
  const float* P = l->p;
  const float* T = l->pt;
  const int64_t* o = c.offset;
  // DRA_VAR_RING_DIRECT (l->rd_slot >= 0): the uint8 frames come straight from the replay ring, sample b of net z = the 4
  // slots ending at idx[b] (+ n_step for the next-state nets); no gathered copy exists
  const bool rd = l->rd_slot >= 0;
  // DRA_VAR_BWD_CHAIN: conv3 / conv2 / conv1 backward as one chained launch (fused.hip bwd_chain_kernel)
  const bool bchain = l->bchain && rd && !l->profiling && l->only_kernel < 0 && part == 0 && !per && !(l->per2_active && l->per2_ride);
  bool head_chain = false;   // DRA_VAR_HEAD_CHAIN: the head launch below also carried fc4's / the head's backward roles
  void *ring_frames = nullptr, *ring_actions = nullptr, *ring_rewards = nullptr, *ring_masks = nullptr;
  int ring_h = 4, ring_n = 1;
  double ring_discount = 1.0;
  if (rd) {
    int rc0 = dra_ring_pointers(l->ring, &ring_frames, &ring_actions, &ring_rewards, &ring_masks);
    if (!rc0) rc0 = dra_ring_shape(l->ring, &ring_h, &ring_n);
    if (!rc0) rc0 = dra_ring_discount(l->ring, &ring_discount);
    if (rc0) return rc0;
    if (ring_h != 4) return DRA_EINVAL;
  }
  if (part != 2 && l->only_chain != 2) {
  // z = 0: online(states)   z = 1: target(next_states)   z = 2: online(next_states) [double-Q]
  const void* x1[3] = {l->state_[l->gb], l->next_state_[l->gb], l->next_state_[l->gb]};
  const float* w1[3] = {P + o[P_W1], T + o[P_W1], P + o[P_W1]};
  const float* b1[3] = {P + o[P_B1], T + o[P_B1], P + o[P_B1]};
  bool chain = false;   // DRA_VAR_FWD_CHAIN: conv1 + conv2 + conv3 as one launch (conv_v2.hip conv_fwd_chain_kernel)
  if (rd) {
    const int64_t off[3] = {0, ring_n, ring_n};
    // (the indices sit in pinned host memory: conv1's workgroups pay the one PCIe read and leave a device copy in l->idx
    // for the head and the weight-gradient kernels of this update)
    // (PER drawn on the device: the previous update's chain kernel left them in device memory)
    const bool dev_idx = per && l->per2_dev;
    const bool pf = (l->variant & DRA_VAR_IDX_PREFETCH) && !dev_idx;
    chain = l->fchain && !l->profiling && l->only_kernel < 0 && nz == 2;
    if (!chain && l->rider_q >= 0 && l->only_kernel < 0) {   // first half of the deferred fc4 segment rides here (common.h DraFc4Rider)
      const DraFc4Rider r = fc4_rider(l, l->pa[l->rider_q]);
      const int nb = fc4_rider_blocks(r.count4);
      dra_conv_attach_rider(&r, 0, (nb * kRiderConv1Pct) / 100, nullptr, nullptr);
    }
    if (chain) {
      const float* w2c[3] = {P + o[P_W2], T + o[P_W2], P + o[P_W2]};
      const float* b2c[3] = {P + o[P_B2], T + o[P_B2], P + o[P_B2]};
      const float* w3c[3] = {P + o[P_W3], T + o[P_W3], P + o[P_W3]};
      const float* b3c[3] = {P + o[P_B3], T + o[P_B3], P + o[P_B3]};
      const bool riding = l->rider_q >= 0;      // the deferred fc4 segment as trailing workgroups of the chained launch
      DraFc4Rider rdr;
      if (riding) rdr = fc4_rider(l, l->pa[l->rider_q]);
      if (l->fs_capturing) dra_conv_chain_attach_announce(l->fs_count);
      if ((l->variant & DRA_VAR_HEAD_CHAIN) && l->hchain_dev) dra_conv_chain_attach_zero(l->hchain_dev);
      // DRA_VAR_TARGET_AHEAD (rd_eager): the update's chain carries the online net alone
      int rcc = dra_conv_fwd_chain(ring_frames, dev_idx ? l->per2_idx + (size_t)l->rd_slot * 1024 : l->idx_pin[l->rd_slot], l->idx,
                                   pf ? l->idx_tag_dev + (size_t)l->rd_slot * 1024 : nullptr, pf ? l->rd_seq_dev : nullptr, off,
                                   l->ah_mode ? 1 : nz,
                                   w1, b1, l->y1, w2c, b2c, l->y2, w3c, b3c, l->y3, B, c.u8_coef, l->fchain_dev,
                                   l->fchain_dev + kFwdChainCounters, l->timeout_flag, riding ? &rdr : nullptr,
                                   l->fchain_dev + kFwdChainCounters + 1, defer_pending_word(l),
                                   defer_valid_word(l, l->rider_q >= 0 ? l->rider_q : 0), s);
      if (rcc) return rcc;
      if (l->only_chain == 1) return DRA_OK;
      // ... and with no stash for this update the target net follows in line (its own chained launch; fc4 below takes both)
      if (l->ah_mode == 2)
        if (int rct = ah_target_launches(l, st, (int)(l->step_no & 1), l->idx_pin[l->rd_slot], false, 0)) return rct;
    } else
    STEP(K_CONV1_F, dra_conv1_fwd_koc_ringbatch(ring_frames, dev_idx ? l->per2_idx + (size_t)l->rd_slot * 1024 : l->idx_pin[l->rd_slot], l->idx,
                                                pf ? l->idx_tag_dev + (size_t)l->rd_slot * 1024 : nullptr, pf ? l->rd_seq_dev : nullptr,
                                                off, nz, w1, b1, l->y1, B, c.u8_coef, DRA_ACT_RELU, s));
  } else {
    STEP(K_CONV1_F, dra_conv_fwd_koc(1, nz, x1, w1, b1, l->y1, B, 1, c.u8_coef, DRA_ACT_RELU, s));
  }
  const void* x2[3] = {l->y1[0], l->y1[1], l->y1[2]};
  const float* w2[3] = {P + o[P_W2], T + o[P_W2], P + o[P_W2]};
  const float* b2[3] = {P + o[P_B2], T + o[P_B2], P + o[P_B2]};
  if (!chain && rd && l->rider_q >= 0 && l->only_kernel < 0) {   // ... the second half here
    const DraFc4Rider r = fc4_rider(l, l->pa[l->rider_q]);
    const int nb = fc4_rider_blocks(r.count4);
    dra_conv_attach_rider(&r, (nb * kRiderConv1Pct) / 100, nb - (nb * kRiderConv1Pct) / 100, nullptr, nullptr);
  }
  if (!chain)
  STEP(K_CONV2_F, dra_conv_fwd_koc(2, nz, x2, w2, b2, l->y2, B, 0, 1.0, DRA_ACT_RELU, s));
  const void* x3[3] = {l->y2[0], l->y2[1], l->y2[2]};
  const float* w3[3] = {P + o[P_W3], T + o[P_W3], P + o[P_W3]};
  const float* b3[3] = {P + o[P_B3], T + o[P_B3], P + o[P_B3]};
  const float* x4[3] = {l->y3[0], l->y3[1], l->y3[2]};
  const float* w4[3] = {P + o[P_W4], T + o[P_W4], P + o[P_W4]};
  const int ks4 = fc4_ks(l);
  // fc4's forward weights are prefetched into the L2 of the XCD that will stream them, by spare workgroups of conv3's forward
  // launch (conv_v2.hip fc4_weight_prefetch; same box: fc4_fwd 10.9 -> 9.95 us, conv3_fwd +1.0 us, +0.5-0.8 % updates/s)
  // ... and the launch after the riders' lowers `pending` and marks the actor copy they completed valid
  if (!chain && rd && l->rider_q >= 0 && l->only_kernel < 0)
    dra_conv_attach_rider(nullptr, 0, 0, defer_pending_word(l), defer_valid_word(l, l->rider_q));
  if (chain) {
    // (conv3 ran inside the chained launch)
  } else if ((l->variant & DRA_VAR_ONESHOT_FWD) && ks4 == kFc4SplitMid && nz == 2 && B <= 32)
    STEP(K_CONV3_F, dra_conv3_fwd_koc_pf(nz, x3, w3, b3, l->y3, B, DRA_ACT_RELU, w4, nz, s));
  else
  STEP(K_CONV3_F, dra_conv_fwd_koc(3, nz, x3, w3, b3, l->y3, B, 0, 1.0, DRA_ACT_RELU, s));
  if (chain && l->ah_mode == 2) x4[1] = l->ah_y3[l->step_no & 1];      // (the in-line target chain's planes)
  if (l->variant & DRA_VAR_ONESHOT_FWD) STEP(K_FC4_F, dra_linear_fwd_slabs_one((chain && l->ah_mode == 1) ? 1 : nz, x4, w4, B, 3136, 512, ks4, l->fc4_slabs, s));
  else STEP(K_FC4_F, dra_linear_fwd_slabs(nz, x4, w4, B, 3136, 512, kFc4Split, l->fc4_slabs, s));
  if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_HEAD], st));
  RingScalars rs;
  memset(&rs, 0, sizeof(rs));
  if (rd) {
    rs.idx = l->idx; rs.actions = (const uint8_t*)ring_actions; rs.rewards = (const double*)ring_rewards;
    rs.masks = (const int32_t*)ring_masks; rs.n_step = ring_n; rs.discount = ring_discount;
    rs.out_action = l->action_[l->gb]; rs.out_reward = l->reward_[l->gb]; rs.out_mask = l->mask_[l->gb];
    if ((l->variant & DRA_VAR_IDX_PREFETCH) && l->only_kernel < 0) rs.seq = l->rd_seq_dev;   // (a replay is not an update)
    if (chain || bchain) rs.chain_epoch = l->fchain_dev + kFwdChainCounters;
    if (chain && l->ah_mode == 1) {     // the target net's partial sums: ahead(this update) left them in the stash
      rs.z1_slabs = l->ah_slabs[l->step_no & 1]; rs.z1_done = l->ah_done; rs.z1_want = l->step_no + 1ull; rs.z1_timeout = l->timeout_flag;
    }
  }
  if (l->only_kernel >= 0 && l->only_kernel != K_HEAD) {
    // (single-kernel replay of another group: no head launch)
  } else if (c.head_kind != DRA_HEAD_VANILLA) {
    int rc = run_dist_head(l, st, per, beta, rs);
    if (rc) return rc;
  } else {
    // weights known before the update (device-side prioritized draw): the fused head applies them, no batch-wide reduction
    const float* per_w = (per && l->per2_active) ? l->weights : nullptr;
    head_chain = (l->variant & DRA_VAR_HEAD_CHAIN) && chain && !per && ks4 == kFc4SplitMid && B <= 32 && nz == 2 && l->late && part == 0 &&
                 !(l->variant & DRA_VAR_BWD_CHAIN_FC) && (l->variant & DRA_VAR_ONESHOT_DGRAD) && l->hchain_dev && !l->profiling;
    // fc4's input-gradient weights prefetched by spare workgroups of the head launch (see head_fused_kernel; same box: the fc
    // backward launch 11.5 -> 10.8 us, 9 163 -> 9 183 updates/s, profiles/r04v_ab_pf_fc4_bwd.jsonl)
    const bool pf = (l->variant & DRA_VAR_ONESHOT_DGRAD) && B % 8 == 0 && B <= 32 && dra_xcd_order_enabled();
    const float* pf_w4 = pf ? P + o[P_W4] : nullptr;
    const int hb = B + (pf ? 3136 / 32 : 0);
    if (head_chain) {
      HeadArgs ha;
      ha.slabs = l->fc4_slabs; ha.nz = nz; ha.B = B; ha.A = A; ha.b4_on = P + o[P_B4]; ha.b4_tg = T + o[P_B4]; ha.wh_on = P + o[P_WH];
      ha.wh_tg = T + o[P_WH]; ha.bh_on = P + o[P_BH]; ha.bh_tg = T + o[P_BH]; ha.action = (const int64_t*)l->action_[l->gb];
      ha.reward = (const float*)l->reward_[l->gb]; ha.mask = (const float*)l->mask_[l->gb]; ha.gamma_n = c.gamma_n; ha.double_q = c.double_q;
      ha.h4_out = l->h4; ha.q_on = l->q[0]; ha.q_tg = l->q[1]; ha.q_on2 = l->q[2]; ha.delta = l->delta; ha.dq = l->dq; ha.dh4 = l->dh4;
      ha.opt_step = l->opt_step; ha.rs = rs; ha.per_w = nullptr; ha.done = l->hchain_dev;
      ChainHook hk;
      hk.wait = l->hchain_dev; hk.wait_target = (unsigned)B; hk.timeout_flag = l->timeout_flag;
      // (the roles of dra_fc_bwd_fused_sq, argument for argument)
      LinDgradOne<512> rd = {};
      rd.dy = l->dh4; rd.w = P + o[P_W4]; rd.xact = l->y3[0]; rd.dx = l->dy3; rd.B = B; rd.I = 3136; rd.act = DRA_ACT_RELU; rd.tiles_n = 3136 / 32;
      rd.hook = hk;
      LinWgradOne<8> rl = {};
      float* Gh = l->g;
      rl.dy = l->dh4; rl.x = l->y3[0]; rl.dw = Gh + o[P_W4]; rl.db = Gh + o[P_B4]; rl.partials = l->partials; rl.B = B; rl.O = 512; rl.I = 3136;
      rl.tiles_o = 512 / 32; rl.groups_i = (rd.tiles_n + 8 - 1) / 8;
      rl.hook = hk;
      HeadWgradRole rh;
      rh.dq = l->dq; rh.h4 = l->h4; rh.dwh = Gh + o[P_WH]; rh.dbh = Gh + o[P_BH]; rh.B = B; rh.A = A;
      const int nd = rd.tiles_n, nw = rl.blocks();
      rh.partials = l->partials + nw;
      rh.hook = hk;
      if (nw + 2 * A != dra_fc_bwd_fused_sq_partials(B, A, 3136)) return DRA_EINVAL;
      constexpr size_t hbytes = (size_t)LinDgradOne<512>::LDS_FLOATS * sizeof(float);
      static DraLdsAttr lds_attr;
      if (int rcl = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&head_fc_bwd_kernel<kFc4SplitMid>), hbytes)) return rcl;
      hipLaunchKernelGGL(head_fc_bwd_kernel<kFc4SplitMid>, dim3(B + nd + nw + 2 * A), dim3(256), hbytes, st, ha, rd, rl, rh, nd, nw);
    } else
    if (ks4 == kFc4SplitWide)
      hipLaunchKernelGGL(head_fused_kernel<kFc4SplitWide>, dim3(hb), dim3(256), 0, st, (const float*)l->fc4_slabs, nz, B, A,
                         P + o[P_B4], T + o[P_B4], P + o[P_WH], T + o[P_WH], P + o[P_BH], T + o[P_BH],
                         (const int64_t*)l->action_[l->gb], (const float*)l->reward_[l->gb], (const float*)l->mask_[l->gb], c.gamma_n,
                         c.double_q,
                         l->h4, l->q[0], l->q[1], l->q[2], l->delta, l->dq, l->dh4, l->opt_step, rs, per_w, pf_w4);
    else if (ks4 == kFc4SplitMid)
      hipLaunchKernelGGL(head_fused_kernel<kFc4SplitMid>, dim3(hb), dim3(256), 0, st, (const float*)l->fc4_slabs, nz, B, A,
                         P + o[P_B4], T + o[P_B4], P + o[P_WH], T + o[P_WH], P + o[P_BH], T + o[P_BH],
                         (const int64_t*)l->action_[l->gb], (const float*)l->reward_[l->gb], (const float*)l->mask_[l->gb], c.gamma_n,
                         c.double_q,
                         l->h4, l->q[0], l->q[1], l->q[2], l->delta, l->dq, l->dh4, l->opt_step, rs, per_w, pf_w4);
    else
      hipLaunchKernelGGL(head_fused_kernel<kFc4Split>, dim3(hb), dim3(256), 0, st, (const float*)l->fc4_slabs, nz, B, A,
                         P + o[P_B4], T + o[P_B4], P + o[P_WH], T + o[P_WH], P + o[P_BH], T + o[P_BH],
                         (const int64_t*)l->action_[l->gb], (const float*)l->reward_[l->gb], (const float*)l->mask_[l->gb], c.gamma_n,
                         c.double_q,
                         l->h4, l->q[0], l->q[1], l->q[2], l->delta, l->dq, l->dh4, l->opt_step, rs, per_w, pf_w4);
    DRA_LAUNCH_CHECK();
    if (per && !per_w) {  // PER needs the batch-wide max of the importance weights: separate kernel recomputes dq, then dh4
      int rc = dra_td_loss(l->q[0], l->q[1], c.double_q ? l->q[2] : nullptr, l->action_[l->gb], 1, l->reward_[l->gb],
                           l->mask_[l->gb], B, A, c.gamma_n,
                           l->samp_prob, beta, c.replay_eps, c.replay_alpha, l->loss, l->dq, l->delta, l->prio, l->weights, s);
      if (rc) return rc;
      rc = dra_linear_bwd_x(l->dq, P + o[P_WH], l->h4, l->dh4, B, 512, A, DRA_ACT_RELU, s);
      if (rc) return rc;
    }
  }
  }   // part != 2
  if (part == 1) return DRA_OK;
  const int NO = l->n_out;      // head outputs the backward sees: A, or A * n_atoms
  float* G = l->g;
  float* S = l->slabs;
  if (l->variant & (DRA_VAR_FUSED_BWD | DRA_VAR_ONESHOT_WGRAD)) {
    // one launch per layer: {head wgrad, fc4 wgrad, fc4 dgrad} {conv3 wgrad, dgrad} {conv2 wgrad, dgrad} {conv1 wgrad};
    // the profile charges each launch to its input-gradient group (the sibling groups read ~0)
    const int var = l->variant;
    const bool own = var & DRA_VAR_ONESHOT_WGRAD;  // per-layer slab buffers
    float* dw[3]; float* dbs[3]; int64_t stride[3];
    const int wi[3] = {P_W1, P_W2, P_W3}, bi_[3] = {P_B1, P_B2, P_B3};
    for (int k = 0; k < 3; ++k) {
      dw[k] = own ? l->lslabs[k] : S + o[wi[k]];
      dbs[k] = own ? l->lslabs[k] + (o[bi_[k]] - o[wi[k]]) : S + o[bi_[k]];
      stride[k] = own ? l->lstride[k] : l->slab_stride;
    }
    if (l->profiling) { DRA_HIP(hipEventRecord(l->ev[K_HEAD_BW], st)); DRA_HIP(hipEventRecord(l->ev[K_FC4_BW], st)); }
    if (l->late) {
      // no norm launch: partials = [fc4 + head workgroups][conv3 fold, riding in conv2's launch][conv2 fold, riding in
      // conv1's launch]; conv1's own slabs are folded by the first workgroups of the optimizer launch
      dra_fold_seg segs[3];
      conv_fold_segs(l, segs);
      const int nfc_expect = dra_fc_bwd_fused_sq_partials(B, NO, 3136), n3_expect = (int)((l->lstride[2] / 4 + 63) / 64), n2_expect = (int)((l->lstride[1] / 4 + 63) / 64);
      // (single-kernel replay: the skipped launches leave their partial counts at the expected values)
      const bool fc_in_chain = bchain && (l->variant & DRA_VAR_BWD_CHAIN_FC) && B <= 32 && !head_chain;
      int nfc = (l->only_kernel >= 0 || (l->only_chain == 2 && !fc_in_chain) || head_chain) ? nfc_expect : 0, n3 = l->only_kernel >= 0 ? n3_expect : 0, n2 = l->only_kernel >= 0 ? n2_expect : 0;
      if (l->only_chain != 2 && !fc_in_chain && !head_chain)
      STEP(K_FC4_BX, dra_fc_bwd_fused_sq(l->dq, l->h4, l->dh4, l->y3[0], P + o[P_W4], G + o[P_WH], G + o[P_BH], G + o[P_W4],
                                         G + o[P_B4], l->dy3, B, NO, 3136, DRA_ACT_RELU, var, l->partials, &nfc,
                                         c.head_kind != DRA_HEAD_VANILLA ? l->action_[l->gb] : nullptr, c.n_atoms, s));
      if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_CONV3_BW], st));
      if (bchain) {
        DraBwdChainFc fcb;
        if (fc_in_chain) {      // DRA_VAR_BWD_CHAIN_FC: the arguments of the dra_fc_bwd_fused_sq launch skipped above
          fcb.dq = l->dq; fcb.h4 = l->h4; fcb.dh4 = l->dh4; fcb.x3 = l->y3[0]; fcb.w4 = P + o[P_W4]; fcb.dwh = G + o[P_WH];
          fcb.dbh = G + o[P_BH]; fcb.dw4 = G + o[P_W4]; fcb.db4 = G + o[P_B4]; fcb.n_actions = NO; fcb.in_features = 3136;
          fcb.sq_partials = l->partials; fcb.n_sq_partials = &nfc;
          fcb.head_action = c.head_kind != DRA_HEAD_VANILLA ? l->action_[l->gb] : nullptr; fcb.head_group = c.n_atoms;
        }
        int rcb = dra_conv_bwd_chain(l->dy3, l->y2[0], P + o[P_W3], dw[2], dbs[2], stride[2], l->dy2, l->y1[0], P + o[P_W2], dw[1],
                                     dbs[1], stride[1], l->dy1, ring_frames, l->idx, dw[0], dbs[0], stride[0], B, c.u8_coef,
                                     DRA_ACT_RELU, &segs[2], &segs[1], G, l->partials + nfc_expect, &n3,
                                     l->partials + nfc_expect + n3_expect, &n2,
                                     l->partials + nfc_expect + n3_expect + n2_expect, l->late_nfold, l->bchain_dev,
                                     l->fchain_dev + kFwdChainCounters, l->timeout_flag, fc_in_chain ? &fcb : nullptr, s);
        if (rcb) return rcb;
        if (nfc != nfc_expect || n3 != n3_expect || n2 != n2_expect) return DRA_EINVAL;
        l->late_nprior = nfc + n3 + n2;
        return DRA_OK;
      }
      if (l->per2_active && l->per2_ride)
        STEP(K_CONV3_BX, dra_conv3_bwd_fused_chain(l->dy3, l->y2[0], P + o[P_W3], l->y2[0], dw[2], dbs[2], stride[2], l->dy2, B,
                                                   DRA_ACT_RELU, var, &l->per2_args, l->per2_split ? 1 : 0, s));
      else
      STEP(K_CONV3_BX, dra_conv_bwd_fused(3, l->dy3, l->y2[0], P + o[P_W3], l->y2[0], dw[2], dbs[2], stride[2], c.ksplit,
                                          l->dy2, B, 0, 1.0, DRA_ACT_RELU, var, s));
      if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_CONV2_BW], st));
      STEP(K_CONV2_BX, dra_conv_bwd_fused_fold(2, l->dy2, l->y1[0], P + o[P_W2], l->y1[0], dw[1], dbs[1], stride[1], l->dy1, B,
                                               DRA_ACT_RELU, var, &segs[2], G, l->partials + nfc, &n3, nullptr, 0, s));
      // (the fold riding in conv1's launch also resets the arrival slots of the optimizer launch that follows)
      STEP(K_CONV1_BW, dra_conv1_wgrad_fold(l->dy1, rd ? ring_frames : (const void*)l->state_[l->gb], rd ? l->idx : nullptr, dw[0],
                                            dbs[0], stride[0], B, c.u8_coef, var, &segs[1], G, l->partials + nfc + n3, &n2,
                                            l->partials + nfc_expect + n3_expect + n2_expect, l->late_nfold,
                                            (l->per2_active && l->per2_ride && l->per2_split) ? &l->per2_args : nullptr, s));
      if (nfc != nfc_expect || n3 != n3_expect || n2 != n2_expect) return DRA_EINVAL;
      l->late_nprior = nfc + n3 + n2;
      if (l->profiling) { DRA_HIP(hipEventRecord(l->ev[K_NORM], st)); DRA_HIP(hipEventRecord(l->ev[K_STEP], st)); }
      return DRA_OK;
    }
    STEP(K_FC4_BX, dra_fc_bwd_fused(l->dq, l->h4, l->dh4, l->y3[0], P + o[P_W4], G + o[P_WH], G + o[P_BH], G + o[P_W4],
                                    G + o[P_B4], l->dy3, B, NO, 3136, DRA_ACT_RELU, var, s));
    if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_CONV3_BW], st));
    if (l->per2_active && l->per2_ride)
      STEP(K_CONV3_BX, dra_conv3_bwd_fused_chain(l->dy3, l->y2[0], P + o[P_W3], l->y2[0], dw[2], dbs[2], stride[2], l->dy2, B,
                                                 DRA_ACT_RELU, var, &l->per2_args, 0, s));
    else
    STEP(K_CONV3_BX, dra_conv_bwd_fused(3, l->dy3, l->y2[0], P + o[P_W3], l->y2[0], dw[2], dbs[2], stride[2], c.ksplit,
                                        l->dy2, B, 0, 1.0, DRA_ACT_RELU, var, s));
    if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_CONV2_BW], st));
    STEP(K_CONV2_BX, dra_conv_bwd_fused(2, l->dy2, l->y1[0], P + o[P_W2], l->y1[0], dw[1], dbs[1], stride[1], c.ksplit,
                                        l->dy1, B, 0, 1.0, DRA_ACT_RELU, var, s));
    if (rd) {
      STEP(K_CONV1_BW, dra_conv1_wgrad_ringbatch(l->dy1, ring_frames, l->idx, dw[0], dbs[0], stride[0], B, c.u8_coef, var, s));
    } else {
      STEP(K_CONV1_BW, dra_conv_bwd_fused(1, l->dy1, l->state_[l->gb], nullptr, nullptr, dw[0], dbs[0], stride[0], c.ksplit, nullptr,
                                          B, 1, c.u8_coef, DRA_ACT_RELU, var, s));
    }
    if (own) {
      dra_fold_seg segs[3];
      conv_fold_segs(l, segs);
      int np_out = 0;
      STEP(K_NORM, dra_grad_sqnorm_segs(G, c.n_params, segs, 3, l->partials, &np_out, s));
      l->n_partials = np_out;
      if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_STEP], st));
      return DRA_OK;
    }
    const int np = dra_norm_partials();
    STEP(K_NORM, dra_grad_sqnorm(G, c.conv_end, S, c.ksplit, l->slab_stride, l->partials, s));
    if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_STEP], st));
    return dra_grad_sqnorm(G + c.conv_end, c.n_params - c.conv_end, nullptr, 0, 0, l->partials + np, s);
  }
  hipStream_t sd = fork ? l->side : st;
  void* sds = (void*)sd;
  if (fork) { DRA_HIP(hipEventRecord(l->ev_fork, st)); DRA_HIP(hipStreamWaitEvent(sd, l->ev_fork, 0)); }
  // side branch: head and fc4 weight gradients need only dq / dh4 / stored activations
  if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_HEAD_BW], st));
  hipLaunchKernelGGL(head_wgrad_kernel, dim3(NO, 8), dim3(64), 0, sd, (const float*)l->dq, (const float*)l->h4, B, NO,
                     G + o[P_WH], G + o[P_BH]);
  DRA_LAUNCH_CHECK();
  STEP(K_FC4_BW, dra_linear_bwd_w(l->dh4, l->y3[0], G + o[P_W4], G + o[P_B4], B, 3136, 512, sds));
  STEP(K_FC4_BX, dra_linear_bwd_x(l->dh4, P + o[P_W4], l->y3[0], l->dy3, B, 3136, 512, DRA_ACT_RELU, s));
  if (fork) { DRA_HIP(hipEventRecord(l->ev_join[0], st)); DRA_HIP(hipStreamWaitEvent(sd, l->ev_join[0], 0)); }
  STEP(K_CONV3_BW, dra_conv_bwd_w_koc(3, l->dy3, l->y2[0], S + o[P_W3], S + o[P_B3], l->slab_stride, c.ksplit, B, 0, 1.0, sds));
  STEP(K_CONV3_BX, dra_conv_bwd_x_koc(3, l->dy3, P + o[P_W3], l->y2[0], l->dy2, B, DRA_ACT_RELU, s));
  if (fork) { DRA_HIP(hipEventRecord(l->ev_join[1], st)); DRA_HIP(hipStreamWaitEvent(sd, l->ev_join[1], 0)); }
  STEP(K_CONV2_BW, dra_conv_bwd_w_koc(2, l->dy2, l->y1[0], S + o[P_W2], S + o[P_B2], l->slab_stride, c.ksplit, B, 0, 1.0, sds));
  STEP(K_CONV2_BX, dra_conv_bwd_x_koc(2, l->dy2, P + o[P_W2], l->y1[0], l->dy1, B, DRA_ACT_RELU, s));
  STEP(K_CONV1_BW, dra_conv_bwd_w_koc(1, l->dy1, l->state_[l->gb], S + o[P_W1], S + o[P_B1], l->slab_stride, c.ksplit, B, 1, c.u8_coef, s));
  if (fork) { DRA_HIP(hipEventRecord(l->ev_join[2], sd)); DRA_HIP(hipStreamWaitEvent(st, l->ev_join[2], 0)); }
  const int np = dra_norm_partials();
  STEP(K_NORM, dra_grad_sqnorm(G, c.conv_end, S, c.ksplit, l->slab_stride, l->partials, s));  // folds the conv slabs
  if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_STEP], st));  // second norm launch is charged to the norm group
  int rc = dra_grad_sqnorm(G + c.conv_end, c.n_params - c.conv_end, nullptr, 0, 0, l->partials + np, s);
  return rc;
}

// PER variant of body_graph: the importance exponent comes from device memory (sampling_prob[B], uploaded with the
// probabilities), so the captured chain has no per-update argument.
static int body_graph_per(dra_dqn_learner* l, hipStream_t st) {
  if (!l->g_update_per_ready) {
    hipGraph_t graph;
    DRA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = run_body(l, st, 1, -1.f, 0);
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&l->g_update_per, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    l->g_update_per_ready = true;
  }
  DRA_HIP(hipGraphLaunch(l->g_update_per, st));
  return DRA_OK;
}

static int body_graph(dra_dqn_learner* l, hipStream_t st) {
  if (!l->g_update_ready) {
    hipGraph_t graph;
    DRA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    // single-queue chain: forked branches end up on other HW queues and every cross-queue join was
    // measured at 10-13 us of idle gap (profiles/r01_timeline_async_forked.txt); same-queue gaps are 0
    int rc = run_body(l, st, 0, 0.f, 0);
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&l->g_update, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    l->g_update_ready = true;
  }
  DRA_HIP(hipGraphLaunch(l->g_update, st));
  return DRA_OK;
}

// Pipelined async mode: update body + optimizer of one step parity as ONE graph (no launch gap in front of
// the optimizer).  The optimizer's second output is the actor parameter copy of the same parity.
// Captures (once) and launches the update of rotation slot q as a graph: uniform replay = ONE graph (body + optimizer);
// PER = two graphs, [forward passes + loss] and [backward + norm + optimizer], with ev_loss recorded between them -- the
// priority write-back and the next prioritized draw (tree stream + one host round trip) then run under the backward pass.
// rd: the ring-direct body (idx_pin[q]); otherwise the gathered minibatch of parity q & 1.
static int capture_part(dra_dqn_learner* l, hipStream_t st, int q, bool rd, int per, int part, bool with_optimizer,
                        hipGraphExec_t* exec) {
  hipGraph_t graph;
  l->gb = q & 1;
  l->rd_slot = rd ? q : -1;
  hipError_t b = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (b != hipSuccess) { l->gb = 0; l->rd_slot = -1; return (int)b; }
  // DRA_VAR_DEFER_FC4: this graph's forwards carry the riders that finish update q - 1's optimizer step (its copy: slot q + 3),
  // and its own optimizer launch leaves fc4's segment to the next graph
  const bool defer = rd && l->defer && !per && part == 0 && with_optimizer;
  l->rider_q = defer ? ((q + 3) & 3) : -1;
  // DRA_VAR_FLAG_SYNC: the plain ring-direct graph's first launch counts itself in fs_count (update_graph counts the replays)
  l->fs_capturing = l->fs && rd && !per && part == 0 && with_optimizer;
  int rc = run_body(l, st, per, per ? -1.f : 0.f, 0, part);     // PER: the exponent is read from sampling_prob[B]
  if (l->fs_capturing) { l->fs_graph[q & 3] = rc == DRA_OK; l->fs_capturing = false; }
  l->rider_q = -1;
  if (rc == DRA_OK && with_optimizer) rc = launch_optimizer(l, st, l->pa[q], defer ? q : -1);
  hipError_t e = hipStreamEndCapture(st, &graph);
  l->gb = 0;
  l->rd_slot = -1;
  if (rc != DRA_OK) return rc;
  if (e != hipSuccess) return (int)e;
  DRA_HIP(hipGraphInstantiate(exec, graph, nullptr, nullptr, 0));
  (void)hipGraphDestroy(graph);
  return DRA_OK;
}

// The prioritized update with the draw on the device (dra_sumtree_per_chain2) as ONE graph: [forward + loss], the draw
// (priorities -> tree, adds, next draw), [backward + norm + optimizer].  Nothing in the backward pass touches what the draw
// reads or writes (prio / tree / sampling_prob / the NEXT slot's indices), which is what lets it ride inside backward launches.
static int capture_per2(dra_dqn_learner* l, hipStream_t st, int q, bool rd, hipGraphExec_t* exec) {
  hipGraph_t graph;
  l->gb = q & 1;
  l->rd_slot = rd ? q : -1;
  hipError_t b = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (b != hipSuccess) { l->gb = 0; l->rd_slot = -1; return (int)b; }
  l->per2_active = true;
  int rc = run_body(l, st, 1, -1.f, 0, 1);
  // The draw rides as roles of two backward launches -- priorities / commits / adds in conv3's, descent / filter / hand-over in
  // conv1's weight gradient (late-fold backward; without it: whole in conv3's) -- for minibatches up to 256 (a role has the
  // launch's 256 threads) and the one-pass backward kernels; otherwise it is a launch of its own between the loss and the
  // backward pass.  (Measured and removed in round 4: the kernel as a parallel BRANCH of the graph -- the forked graph no
  // longer overlapped with the actor stream's, 4.5 k vs 6.0 k updates/s -- and whole-in-conv3's as a choice: 27 instead of
  // 12 us for that launch; DESIGN.md section 4, profiles/r03h_*, r03i_*.)
  hipStream_t sd = st;
  const int both = DRA_VAR_ONESHOT_WGRAD | DRA_VAR_ONESHOT_DGRAD;
  l->per2_ride = l->c.batch <= 256 && (l->variant & both) == both;
  l->per2_split = l->per2_ride && l->late;
  if (rc == DRA_OK && l->per2_ride)
    rc = dra_sumtree_per_chain2_args(l->per_tree, l->per2_io[q & 3], l->delta, l->c.replay_eps, l->c.replay_alpha, l->prio,
                                     l->per_stat, l->per2_dev, l->per2_words, l->per2_idx + (size_t)((q + 1) & 3) * 1024,
                                     l->samp_prob, l->weights, l->c.batch, &l->per2_args);
  else if (rc == DRA_OK)
    rc = dra_sumtree_per_chain2(l->per_tree, l->per2_io[q & 3], l->delta, l->c.replay_eps, l->c.replay_alpha, l->prio, l->per_stat,
                                l->per2_dev, l->per2_words, l->per2_idx + (size_t)((q + 1) & 3) * 1024, l->samp_prob, l->weights,
                                l->c.batch, (void*)sd);
  if (rc == DRA_OK) rc = run_body(l, st, 1, -1.f, 0, 2);
  l->per2_active = false;
  l->per2_ride = l->per2_split = false;
  if (rc == DRA_OK) rc = launch_optimizer(l, st, l->pa[q]);
  hipError_t e = hipStreamEndCapture(st, &graph);
  l->gb = 0;
  l->rd_slot = -1;
  if (rc != DRA_OK) return rc;
  if (e != hipSuccess) return (int)e;
  DRA_HIP(hipGraphInstantiate(exec, graph, nullptr, nullptr, 0));
  (void)hipGraphDestroy(graph);
  return DRA_OK;
}

static int update_graph(dra_dqn_learner* l, hipStream_t st, int q, bool rd, int per) {
  hipGraphExec_t* exec = rd ? (per ? &l->g_rd_per[q] : &l->g_rd[q]) : (per ? &l->g_pipe_per[q] : &l->g_pipe[q]);
  hipGraphExec_t* exec_b = rd ? &l->g_rd_per_b[q] : &l->g_pipe_per_b[q];
  bool* ready = rd ? (per ? &l->g_rd_per_ready[q] : &l->g_rd_ready[q]) : (per ? &l->g_pipe_per_ready[q] : &l->g_pipe_ready[q]);
  if (!*ready) {
    int rc;
    if (per && l->per2_dev && l->per_tree) {
      if ((rc = capture_per2(l, st, q, rd, exec))) return rc;
    } else if (per) {
      if ((rc = capture_part(l, st, q, rd, 1, 1, false, exec))) return rc;
      if ((rc = capture_part(l, st, q, rd, 1, 2, true, exec_b))) return rc;
    } else if ((rc = capture_part(l, st, q, rd, 0, 0, true, exec))) {
      return rc;
    }
    *ready = true;
  }
  // DRA_VAR_DEFER_FC4: a pending fc4 segment is completed by THIS graph's riders only if it is the previous rotation slot's and
  // this is a riding graph; anything else steps it first
  const bool riding = rd && l->defer && !per;
  if (l->defer_host && !(riding && l->defer_q == ((q + 3) & 3))) {
    int rcf = flush_fc4(l, st);
    if (rcf) return rcf;
  }
  DRA_HIP(hipGraphLaunch(*exec, st));
  if (rd && !per && l->fs_graph[q & 3]) l->fs_issued++;     // (its first launch bumps fs_count when it starts)
  if (riding) { l->defer_host = true; l->defer_q = q; }
  if (per && !(l->per2_dev && l->per_tree)) {
    DRA_HIP(hipEventRecord(l->ev_loss, st));
    DRA_HIP(hipGraphLaunch(*exec_b, st));
  }
  return DRA_OK;
}

static int pipe_graph(dra_dqn_learner* l, hipStream_t st, int par, int per = 0) {   // par: 0 / 1, or step mod 4
  return update_graph(l, st, par, false, per);
}

static int rd_graph(dra_dqn_learner* l, hipStream_t st, int q, int per = 0) { return update_graph(l, st, q, true, per); }

// PrioritizedReplay.sample() inside the update (sumtree.hip dra_sumtree_per_chain2): io0..3 = pinned dra_per_chain2_io blocks, rng_words = pinned ring of
// DRA_PER_RNG_WORDS Mersenne-Twister outputs.  From then on every prioritized update of the ring-direct pipeline reads its
// minibatch indices from device memory (slot step_no & 3) and its sampling probabilities / exponent from the learner's
// sampling_prob buffer, both written by the PREVIOUS update's chain kernel -- or by _per_chain2_seed for the first one.
DRA_API int dra_dqn_learner_set_per_chain2(dra_dqn_learner* l, dra_sumtree* tree, double* stat_dev, dra_per_chain2_io* io0,
                                           dra_per_chain2_io* io1, dra_per_chain2_io* io2, dra_per_chain2_io* io3,
                                           const uint32_t* rng_words_pinned) {
  if (!l || !tree || !stat_dev || !io0 || !io1 || !io2 || !io3 || !rng_words_pinned) return DRA_EINVAL;
  if (!(l->variant & DRA_VAR_RING_DIRECT) || !(l->variant & DRA_VAR_GATHER_ON_UPDATE)) return DRA_EINVAL;
  if (l->c.batch > DRA_PER_CHAIN_MAX) return DRA_EINVAL;
  for (int k = 0; k < 4; ++k)
    if (l->g_rd_per_ready[k] || l->g_pipe_per_ready[k]) return DRA_EINVAL;     // the graphs bake the choice
  if (!l->per2_dev) {
    int64_t sb = 0;
    (void)dra_sumtree_per_chain2_state_bytes(&sb);
    DRA_HIP(hipMalloc(&l->per2_dev, (size_t)sb));
    DRA_HIP(hipMemset(l->per2_dev, 0, (size_t)sb));
    DRA_HIP(hipMalloc(&l->per2_idx, (size_t)4 * 1024 * sizeof(int64_t)));
    DRA_HIP(hipMemset(l->per2_idx, 0, (size_t)4 * 1024 * sizeof(int64_t)));
  }
  l->per_tree = tree; l->per_stat = stat_dev;
  l->per2_io[0] = io0; l->per2_io[1] = io1; l->per2_io[2] = io2; l->per2_io[3] = io3;
  l->per2_words = rng_words_pinned;
  return DRA_OK;
}
// The minibatch of the NEXT update, from the host (the first prioritized update's classic draw, or a resumed run): leaves,
// ring indices, sampling probabilities (f64, cast as tensor() would), its importance exponent, and the word-ring cursor /
// launch count the chain kernel continues from.  Ordered on `stream` (the update stream).
DRA_API int dra_dqn_learner_per_chain2_seed(dra_dqn_learner* l, const int64_t* tree_idx, const int64_t* data_idx,
                                            const double* prob, float beta, uint64_t rng_cursor, uint64_t seq, void* stream) {
  if (!l || !l->per2_dev || !tree_idx || !data_idx || !prob) return DRA_EINVAL;
  const int B = l->c.batch;
  hipStream_t st = dra_stream(stream);
  DRA_HIP(hipStreamSynchronize(st));
  float sp[DRA_PER_CHAIN_MAX + 1];
  for (int i = 0; i < B; ++i) sp[i] = (float)prob[i];
  sp[B] = beta;
  DRA_HIP(hipMemcpy(l->samp_prob, sp, (size_t)(B + 1) * sizeof(float), hipMemcpyHostToDevice));
  int rc = dra_sumtree_per_chain2_state_set(l->per2_dev, rng_cursor, seq, tree_idx, B);
  if (rc) return rc;
  // the importance weights of that update, by the kernel every other path uses (exponent read from samp_prob[B])
  if ((rc = dra_per_weights(nullptr, l->samp_prob, B, -1.f, l->c.replay_eps, l->c.replay_alpha, nullptr, l->weights, stream))) return rc;
  DRA_HIP(hipStreamSynchronize(st));
  DRA_HIP(hipMemcpy(l->per2_idx + (size_t)(l->step_no & 3) * 1024, data_idx, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice));
  return DRA_OK;
}
// Spins (no driver call) until the chain kernel of rotation slot `slot` has published launch number `seq`; DRA_ETIMEDOUT
// after timeout_us.
DRA_API int dra_dqn_learner_per_chain2_wait(dra_dqn_learner* l, int slot, uint64_t seq, int64_t timeout_us) {
  if (!l || !l->per2_dev || slot < 0 || slot > 3) return DRA_EINVAL;
  const volatile uint64_t* p = &l->per2_io[slot]->out_seq;
  if (__atomic_load_n(p, __ATOMIC_ACQUIRE) >= seq) return DRA_OK;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    for (int i = 0; i < 256; ++i) {
      if (__atomic_load_n(p, __ATOMIC_ACQUIRE) >= seq) {
        l->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return DRA_OK;
      }
      __builtin_ia32_pause();
    }
    if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > (double)timeout_us) return DRA_ETIMEDOUT;
  }
}
DRA_API int dra_dqn_learner_next_slot(dra_dqn_learner* l, int* slot) {
  if (!l || !slot) return DRA_EINVAL;
  *slot = (int)(l->step_no & 3);
  return DRA_OK;
}
DRA_API int dra_dqn_learner_sync_loss(dra_dqn_learner* l) {
  if (!l) return DRA_EINVAL;
  DRA_HIP(hipEventSynchronize(l->ev_loss));
  return DRA_OK;
}

// The tree stream of a PER pipeline waits here for the TD errors / new priorities of the update issued last (they exist
// after the loss kernel: the backward pass and the optimizer are still to run).
DRA_API int dra_dqn_learner_wait_loss(dra_dqn_learner* l, void* stream) {
  if (!l || !stream) return DRA_EINVAL;
  DRA_HIP(hipStreamWaitEvent(dra_stream(stream), l->ev_loss, 0));
  return DRA_OK;
}

// Checkers (bench.py's parity check, tests): with DRA_VAR_RING_DIRECT no gathered minibatch exists; keep != 0 makes the
// pipelined step ALSO run the gather into the learner's minibatch buffers (dra_dqn_learner_last_minibatch) -- the update
// itself still reads the ring.
DRA_API int dra_dqn_learner_keep_minibatch(dra_dqn_learner* l, int keep) {
  if (!l) return DRA_EINVAL;
  l->keep_minibatch = keep != 0;
  return DRA_OK;
}

// One gradient update on the indices currently in the learner's idx buffer (all on `stream`).
// use_graph != 0 replays the captured forward/backward graph (uniform replay only: the PER beta
// is a kernel argument that changes per update); the stream must not be the NULL stream.
DRA_API int dra_dqn_learner_update(dra_dqn_learner* l, int use_graph, int per, float beta, void* stream) {
  if (!l) return DRA_EINVAL;
  if (*l->timeout_flag) return DRA_ETIMEDOUT;
  if (int rcf = flush_fc4(l, dra_stream(stream))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  hipStream_t st = dra_stream(stream);
  l->profiling = false;
  l->pa_valid = false;
  int rc = launch_gather(l, st);
  if (rc) return rc;
  rc = (use_graph && !per) ? body_graph(l, st) : ((use_graph && per && beta < 0.f) ? body_graph_per(l, st) : run_body(l, st, per, beta, 0));
  if (rc) return rc;
  if ((rc = launch_optimizer(l, st))) return rc;
  DRA_HIP(hipEventRecord(l->ev_step_done, st));
  l->last_done = l->ev_step_done;
  return DRA_OK;
}

// Eager update with a HIP event before every kernel group (on the launch stream); returns the
// per-group milliseconds of THIS update.  Synchronises: measurement aid, not the hot path.
DRA_API int dra_dqn_learner_profile(dra_dqn_learner* l, float* out_ms, int n_out, void* stream) {
  if (!l || !out_ms || n_out < K_COUNT) return DRA_EINVAL;
  if (int rcf = flush_fc4(l, dra_stream(stream))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  hipStream_t st = dra_stream(stream);
  l->profiling = true;
  l->pa_valid = false;   // the parameters change behind the async actor's copies
  DRA_HIP(hipEventRecord(l->ev[K_GATHER], st));
  int rc = launch_gather(l, st);
  if (!rc) rc = run_body(l, st, 0, 0.f, 0);
  // run_body recorded ev[K_STEP] before the second norm launch; re-record it right before the optimizer
  if (!rc) rc = (int)hipEventRecord(l->ev[K_STEP], st);
  if (!rc) rc = launch_optimizer(l, st);
  if (!rc) rc = (int)hipEventRecord(l->ev[K_COUNT], st);
  l->profiling = false;
  if (rc != DRA_OK) return rc;
  DRA_HIP(hipEventSynchronize(l->ev[K_COUNT]));
  for (int k = 0; k < K_COUNT; ++k) {
    float ms = 0.f;
    DRA_HIP(hipEventElapsedTime(&ms, l->ev[k], l->ev[k + 1]));
    out_ms[k] = ms;
  }
  if (n_out > K_COUNT) {
    // the bracket itself: two event records with nothing in between, on the same stream.  A kernel's bracket reads
    // (this) + (the kernel's duration); callers subtract it (rocprofv3's per-kernel durations have no such term)
    DRA_HIP(hipEventRecord(l->ev[0], st));
    DRA_HIP(hipEventRecord(l->ev[1], st));
    DRA_HIP(hipEventSynchronize(l->ev[1]));
    float ms = 0.f;
    DRA_HIP(hipEventElapsedTime(&ms, l->ev[0], l->ev[1]));
    out_ms[K_COUNT] = ms;
  }
  return DRA_OK;
}

__global__ void __launch_bounds__(256) empty_probe_kernel(const int* p) { if (p == reinterpret_cast<const int*>(1)) __builtin_trap(); }

// Measurement aid: kernel group `kernel` of the update ALONE, `reps` dependent launches captured into one graph and replayed
// between two events on `stream` -- out_us[0] = microseconds per launch (kernel + one in-graph dependent-launch boundary);
// out_us[1] = the same for an EMPTY kernel (the boundary alone); the kernel's own duration is their difference, which is what
// rocprofv3's per-kernel duration measures.  The kernel runs on the workspaces the last update left (same shapes, same
// grids as the timed graphs; results are overwritten by the next update).  Groups that change state a replay must not change
// (the optimizer) and groups the variant does not launch are refused.  Synchronises.
DRA_API int dra_dqn_learner_kernel_replay(dra_dqn_learner* l, int kernel, int reps, float* out_us, void* stream) {
  if (!l || !out_us || reps < 1 || reps > 4096) return DRA_EINVAL;
  if (int rcf = flush_fc4(l, dra_stream(stream))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  if (kernel <= K_GATHER || kernel >= K_NORM) return DRA_EINVAL;
  if (!(l->variant & (DRA_VAR_FUSED_BWD | DRA_VAR_ONESHOT_WGRAD)) || !l->late) return DRA_EINVAL;   // (the default chain's groups)
  if (kernel == K_HEAD_BW || kernel == K_FC4_BW || kernel == K_CONV3_BW || kernel == K_CONV2_BW) return DRA_EINVAL;  // ride in *_BX
  hipStream_t st = dra_stream(stream);
  DRA_HIP(hipStreamSynchronize(st));
  float us[2] = {0.f, 0.f};
  for (int pass = 0; pass < 2; ++pass) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    const bool rd = l->variant & DRA_VAR_RING_DIRECT;
    l->gb = 0;
    l->rd_slot = rd ? 0 : -1;
    l->only_kernel = kernel;
    hipError_t b = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    int rc = b == hipSuccess ? DRA_OK : (int)b;
    for (int r = 0; r < reps && rc == DRA_OK && b == hipSuccess; ++r) {
      if (pass == 0) rc = run_body(l, st, 0, 0.f, 0);
      else hipLaunchKernelGGL(empty_probe_kernel, dim3(224), dim3(256), 0, st, (const int*)nullptr);
    }
    l->only_kernel = -1;
    l->rd_slot = -1;
    hipError_t e = b == hipSuccess ? hipStreamEndCapture(st, &graph) : b;
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    DRA_HIP(hipGraphLaunch(exec, st));                   // warm: code objects / TLB / clocks
    DRA_HIP(hipStreamSynchronize(st));
    float best = 0.f;
    for (int t = 0; t < 3; ++t) {                        // the fastest of three replays (a host hiccup only ever adds time)
      DRA_HIP(hipEventRecord(l->ev[0], st));
      DRA_HIP(hipGraphLaunch(exec, st));
      DRA_HIP(hipEventRecord(l->ev[1], st));
      DRA_HIP(hipEventSynchronize(l->ev[1]));
      float ms = 0.f;
      DRA_HIP(hipEventElapsedTime(&ms, l->ev[0], l->ev[1]));
      if (t == 0 || ms < best) best = ms;
    }
    us[pass] = best * 1e3f / (float)reps;
    (void)hipGraphExecDestroy(exec);
  }
  out_us[0] = us[0];
  out_us[1] = us[1];
  return DRA_OK;
}

// Measurement aid: the chained forward (which = 0: conv1 + conv2 + conv3 of both nets, conv_fwd_chain_kernel) or backward
// (which = 1: conv3 / conv2 / conv1 backward + the two slab folds, bwd_chain_kernel) launch of the update ALONE -- the launches the
// timed pipeline runs under DRA_VAR_FWD_CHAIN / DRA_VAR_BWD_CHAIN, which dra_dqn_learner_kernel_replay's per-layer groups are not.
// `reps` x [chain launch, a one-thread launch that advances the chains' epoch word as the update's head kernel does] in one
// captured graph between two events: out_us[0] = microseconds per repetition, out_us[1] = the same with an EMPTY kernel in the
// chain launch's place; their difference is the chained kernel's own duration beyond an empty launch (what rocprofv3 reports
// for it, minus the deferred fc4 segment's riders: a replay must not step parameters, the pending segment is flushed first).
// Same workspaces, indices and grids as the last update.  Synchronises.
DRA_API int dra_dqn_learner_chain_replay(dra_dqn_learner* l, int which, int reps, float* out_us, void* stream) {
  if (!l || !out_us || reps < 1 || reps > 4096 || which < 0 || which > 1) return DRA_EINVAL;
  if (!(l->variant & DRA_VAR_RING_DIRECT) || !l->late || (which == 0 ? !l->fchain : !l->bchain) || l->c.double_q) return DRA_EINVAL;
  if (int rcl = fs_leave_any(l)) return rcl;
  hipStream_t st = dra_stream(stream);
  if (int rcf = flush_fc4(l, st)) return rcf;
  DRA_HIP(hipStreamSynchronize(st));
  unsigned* epoch = l->fchain_dev + kFwdChainCounters;
  float us[2] = {0.f, 0.f};
  for (int pass = 0; pass < 2; ++pass) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    l->gb = 0;
    l->rd_slot = 0;
    l->only_chain = which + 1;
    hipError_t b = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    int rc = b == hipSuccess ? DRA_OK : (int)b;
    for (int r = 0; r < reps && rc == DRA_OK && b == hipSuccess; ++r) {
      // (the forward chain waits for (epoch + 1) x arrivals: the bump follows it; the backward chain for epoch x arrivals, its
      // own update's head included: the bump precedes it)
      if (which == 1) hipLaunchKernelGGL(chain_epoch_bump_kernel, dim3(1), dim3(1), 0, st, epoch);
      if (pass == 0) rc = run_body(l, st, 0, 0.f, 0);
      else hipLaunchKernelGGL(empty_probe_kernel, dim3(224), dim3(256), 0, st, (const int*)nullptr);
      if (which == 0) hipLaunchKernelGGL(chain_epoch_bump_kernel, dim3(1), dim3(1), 0, st, epoch);
    }
    l->only_chain = 0;
    l->rd_slot = -1;
    hipError_t e = b == hipSuccess ? hipStreamEndCapture(st, &graph) : b;
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    float best = 0.f;
    for (int t = 0; t < 4; ++t) {                        // one warm replay, then the fastest of three
      DRA_HIP(hipEventRecord(l->ev[0], st));
      DRA_HIP(hipGraphLaunch(exec, st));
      DRA_HIP(hipEventRecord(l->ev[1], st));
      DRA_HIP(hipEventSynchronize(l->ev[1]));
      float ms = 0.f;
      DRA_HIP(hipEventElapsedTime(&ms, l->ev[0], l->ev[1]));
      if (t == 1 || (t > 1 && ms < best)) best = ms;
    }
    // the replays advanced the epoch word for ONE chain's counters (the empty pass for none): both chains are brought level
    // again -- every arrival counter holds epoch x (arrivals per epoch) between updates, so all zero is a level state
    DRA_HIP(hipMemsetAsync(l->fchain_dev, 0, (size_t)(kFwdChainCounters + 1) * sizeof(unsigned), st));
    DRA_HIP(hipMemsetAsync(l->bchain_dev, 0, (size_t)dra_bwd_chain_counters() * sizeof(unsigned), st));
    DRA_HIP(hipStreamSynchronize(st));
    us[pass] = best * 1e3f / (float)reps;
    (void)hipGraphExecDestroy(exec);
  }
  out_us[0] = us[0];
  out_us[1] = us[1];
  return *l->timeout_flag ? DRA_ETIMEDOUT : DRA_OK;   // (a chain workgroup's bounded wait gave up: the reading is void)
}

// The minibatch the most recently issued update consumed (device pointers into the learner's own buffers: uint8
// [B][4][84][84] states / next states, int64 [B] actions, f32 [B] n-step rewards and masks).  Valid until two more
// updates have been issued; callers synchronise first.  For checkers (bench.py's parity_check replays the update on
// the CPU oracle), not for the hot path.
DRA_API int dra_dqn_learner_last_minibatch(dra_dqn_learner* l, void** state, void** next_state, void** action, void** reward,
                                           void** mask) {
  if (!l) return DRA_EINVAL;
  const int g = l->last_gb;
  if (state) *state = l->state_[g];
  if (next_state) *next_state = l->next_state_[g];
  if (action) *action = l->action_[g];
  if (reward) *reward = l->reward_[g];
  if (mask) *mask = l->mask_[g];
  return DRA_OK;
}

// The parameters were written from outside the learner (checkpoint load): the async actor's double-buffered copies
// are reseeded from them by the next step.
DRA_API int dra_dqn_learner_invalidate_actor_copy(dra_dqn_learner* l) {
  if (!l) return DRA_EINVAL;
  l->pa_valid = false;
  // the host-emulator async actor (dra_dqn_learner_q_host_async) never looks at pa_valid: it reads the copy update
  // (hq_updates - 2) wrote.  Start its rotation over, so that the next call re-copies the online parameters as they are now
  // (the caller has synchronised: DQNAgent.load).
  l->hq_updates = 0;
  l->hq_seeded = false;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_kernel_name(int k, char* out, int n) {
  if (k < 0 || k >= K_COUNT || !out || n < 1) return DRA_EINVAL;
  strncpy(out, kKernelNames[k], (size_t)n - 1);
  out[n - 1] = 0;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_kernel_count(void) { return K_COUNT; }

DRA_API int dra_dqn_learner_sync_target(dra_dqn_learner* l, void* stream) {
  if (!l) return DRA_EINVAL;
  if (int rcf = flush_fc4(l, dra_stream(stream))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  return dra_copy_f32(l->pt, l->p, l->c.n_params, stream);
}

// ------------------------------------------------------------------------------------------------
// Actor step (DQN_agent.py:24-45 + torch_utils.py:51-58) on device.  Per-step arguments come from
// the device parameter block `prm` (entry e), so the kernels can live in a captured graph.
//   env_stack_kernel : 4 workgroups.  Workgroup j < 3 copies ring frame (slot-3+j, modular) into the
//                      actor's stack; workgroup 3 produces the NEW observation frame of `slot` --
//                      synthetic counter-hash frame (counter >= 0) written to the ring together with
//                      its hashed reward / mask, or the frame already in the ring (counter < 0) --
//                      and copies it to the stack.
//   actor_head_kernel: fc4 split-K reduction + bias + ReLU + head + epsilon-greedy with the
//                      HOST-drawn (random_action, dice) -> action record of `slot`.
__global__ void __launch_bounds__(256)
env_stack_kernel(const dra_dqn_step_params* __restrict__ prm, int e, uint8_t* __restrict__ frames,
                 double* __restrict__ rewards, int32_t* __restrict__ masks, int64_t capacity, uint64_t seed,
                 int done_period, uint8_t* __restrict__ stack) {
  const int j = blockIdx.x;  // 0 = oldest of the 4-frame stack
  const int64_t newest = prm->slot[e];
  const int64_t counter = prm->counter[e];
  int64_t slot = newest - 3 + j;
  if (slot < 0) slot += capacity;
  uint64_t* dst = reinterpret_cast<uint64_t*>(stack + (int64_t)j * 7056);
  uint64_t* src = reinterpret_cast<uint64_t*>(frames + slot * 7056);
  if (j == 3 && counter >= 0) {
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull + (uint64_t)counter * 882ull;
    for (int w = threadIdx.x; w < 882; w += blockDim.x) {
      const uint64_t v = mix64(base + (uint64_t)w);
      src[w] = v;
      dst[w] = v;
    }
    if (threadIdx.x == 0) {
      rewards[newest] = synth_reward(seed, prm->rcounter[e]);
      masks[newest] = synth_mask(seed, prm->rcounter[e], done_period);
    }
  } else {
    for (int w = threadIdx.x; w < 882; w += blockDim.x) dst[w] = src[w];
  }
}

__global__ void __launch_bounds__(512)
actor_head_kernel(const dra_dqn_step_params* __restrict__ prm, int e, const float* __restrict__ slabs, int ks,
                  const float* __restrict__ b4, const float* __restrict__ wh, const float* __restrict__ bh, int A,
                  uint8_t* __restrict__ ring_actions, float* __restrict__ q_out, int64_t* __restrict__ out_action) {
  __shared__ float s_h[512];
  __shared__ float s_red[8];
  __shared__ float s_q[64];
  const int k = threadIdx.x;
  float part[kFc4Split];
#pragma unroll
  for (int i = 0; i < kFc4Split; ++i) part[i] = slabs[i * 512 + k];
  float v = part[0];
#pragma unroll
  for (int i = 1; i < kFc4Split; ++i) v += part[i];
  v += b4[k];
  s_h[k] = v > 0.f ? v : 0.f;
  __syncthreads();
  for (int a = 0; a < A; ++a) {
    float part = wave_sum(s_h[k] * wh[a * 512 + k]);
    if ((k & 63) == 0) s_red[k >> 6] = part;
    __syncthreads();
    if (k == 0) {
      float t = 0.f;
      for (int i = 0; i < 8; ++i) t += s_red[i];
      s_q[a] = t + bh[a];
    }
    __syncthreads();
  }
  if (k < A && q_out) q_out[k] = s_q[k];
  if (k == 0) {
    int best = 0;
    float bv = s_q[0];
    for (int a = 1; a < A; ++a) if (s_q[a] > bv) { bv = s_q[a]; best = a; }  // np.argmax: first max
    const int64_t act = (prm->dice[e] < prm->epsilon[e]) ? (int64_t)prm->random_action[e] : (int64_t)best;
    const int64_t slot = prm->slot[e];
    if (prm->store_action[e]) *reinterpret_cast<int64_t*>(ring_actions + slot * 8) = act;
    if (out_action) out_action[e] = act;
  }
}

// ---- actor v2 (DRA_VAR_ACTOR_V2): 5 kernels per env step instead of 6, none of them K-chunked:
//   conv1 reads its 4-frame stack straight from the ring (dra_conv1_fwd_koc_ring), conv2, conv3,
//   actor_fc4_kernel  : batch-1 fc4 as a GEMV -- one wave per output row, the 12.5 KB row and the input as
//                       float4 loads all in flight, wave reduction, bias + ReLU;
//   actor_head_env_kernel: head + epsilon-greedy + action record of env step e, then the ENVIRONMENT step:
//                       the synthetic frame / reward / mask of transition e+1 (what env.step(action)
//                       returns, envs.py:140-141).  env_frame_kernel produces the frame of e = 0.
__device__ __forceinline__ void synth_transition(const dra_dqn_step_params* __restrict__ prm, int e,
                                                 uint8_t* __restrict__ frames, double* __restrict__ rewards,
                                                 int32_t* __restrict__ masks, uint64_t seed, int done_period) {
  const int64_t counter = prm->counter[e];
  if (counter < 0) return;  // frame already in the ring
  const int64_t slot = prm->slot[e];
  uint64_t* dst = reinterpret_cast<uint64_t*>(frames + slot * 7056);
  const uint64_t base = seed * 0x9E3779B97F4A7C15ull + (uint64_t)counter * 882ull;
  for (int w = threadIdx.x; w < 882; w += blockDim.x) dst[w] = mix64(base + (uint64_t)w);
  if (threadIdx.x == 0) {
    rewards[slot] = synth_reward(seed, prm->rcounter[e]);
    masks[slot] = synth_mask(seed, prm->rcounter[e], done_period);
  }
}

__global__ void __launch_bounds__(256)
env_frame_kernel(const dra_dqn_step_params* __restrict__ prm, int e, uint8_t* __restrict__ frames,
                 double* __restrict__ rewards, int32_t* __restrict__ masks, uint64_t seed, int done_period) {
  synth_transition(prm, e, frames, rewards, masks, seed, done_period);
}

__global__ void __launch_bounds__(256)
actor_fc4_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                 float* __restrict__ h4, int in_features) {
  constexpr int R = 13;  // float4 per lane: 3136 / 4 / 64 = 12.25
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  const int nv = in_features >> 2;
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w + (int64_t)row * in_features);
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  DRA_STAMP(TR_A_FC4, 0);
  float4 wv[R], xv[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int i = min(lane + 64 * q, nv - 1);
    wv[q] = w4[i];
    xv[q] = x4[i];
  }
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    float4 a = wv[q], b = xv[q];
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));  // loads stay unconditional and batched
    if (lane + 64 * q < nv) acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float v = acc + bias[row];
    h4[row] = v > 0.f ? v : 0.f;
  }
  DRA_STAMP(TR_A_FC4, 5);
  DRA_STAMP_END(TR_A_FC4);
}

// actor_fc4_kernel whose input is conv3's two partial planes (dra_conv_b1_split; plane 0 carries the bias), x = relu(x0 + x1)
// formed ONCE per workgroup and staged in LDS (12.5 KB) instead of both planes in every lane's registers: 13 float4 of weights + a few transient registers per lane, so that a CU keeps >= 4 of these workgroups
// resident -- the 128 workgroups are ONE round on the actor's 32 CUs (the register-resident form ran 64 + 32 + 32: phase
// trace profiles/r02zj_phase_async_acu32.json, 7.7 us per env step for 3.3 us workgroups).  Same products, same order.
__global__ void __launch_bounds__(256)
actor_fc4_planes_lds_kernel(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ w,
                            const float* __restrict__ bias, float* __restrict__ h4, int in_features) {
  constexpr int R = 13, NVMAX = 784;  // 3136 / 4 float4; per lane 3136 / 4 / 64 = 12.25
  __shared__ float4 s_x[NVMAX];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  const int nv = in_features >> 2;
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w + (int64_t)row * in_features);
  const float4* __restrict__ a4 = reinterpret_cast<const float4*>(x0);
  const float4* __restrict__ c4 = reinterpret_cast<const float4*>(x1);
  DRA_STAMP(TR_A_FC4, 0);
  float4 wv[R];
#pragma unroll
  for (int q = 0; q < R; ++q) wv[q] = w4[min(lane + 64 * q, nv - 1)];
  constexpr int XQ = (NVMAX + 255) / 256;
  float4 av[XQ], cv[XQ];
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int i = min((int)threadIdx.x + 256 * q, nv - 1);
    av[q] = a4[i];
    cv[q] = c4[i];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int i = (int)threadIdx.x + 256 * q;
    float4 b;
    b.x = fmaxf(av[q].x + cv[q].x, 0.f); b.y = fmaxf(av[q].y + cv[q].y, 0.f);
    b.z = fmaxf(av[q].z + cv[q].z, 0.f); b.w = fmaxf(av[q].w + cv[q].w, 0.f);
    if (i < nv) s_x[i] = b;
  }
  __syncthreads();
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    float4 a = wv[q];
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));  // loads stay unconditional and batched
    const float4 b = s_x[min(lane + 64 * q, nv - 1)];
    if (lane + 64 * q < nv) acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float v = acc + bias[row];
    h4[row] = v > 0.f ? v : 0.f;
  }
  DRA_STAMP(TR_A_FC4, 5);
  DRA_STAMP_END(TR_A_FC4);
}

__global__ void __launch_bounds__(1024)
actor_head_env_kernel(const dra_dqn_step_params* __restrict__ prm, int e, const float* __restrict__ h4,
                      const float* __restrict__ wh, const float* __restrict__ bh, int A,
                      uint8_t* __restrict__ ring_actions, float* __restrict__ q_out, uint8_t* __restrict__ frames,
                      double* __restrict__ rewards, int32_t* __restrict__ masks, uint64_t seed, int done_period,
                      const HeadSpec hs) {
  __shared__ float s_q[64];
  __shared__ float s_out[kMaxHeadOut];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (hs.kind != DRA_HEAD_VANILLA) {
    dist_head_q(h4, wh, bh, A, hs, s_out, s_q);
  } else {
    for (int a = wave; a < A; a += (int)(blockDim.x >> 6)) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part += h4[lane + 64 * i] * wh[a * 512 + lane + 64 * i];
      part = wave_sum(part);
      if (lane == 0) s_q[a] = part + bh[a];
    }
    __syncthreads();
  }
  if (threadIdx.x < A && q_out) q_out[threadIdx.x] = s_q[threadIdx.x];
  if (threadIdx.x == 0) {
    int best = 0;
    float bv = s_q[0];
    for (int a = 1; a < A; ++a) if (s_q[a] > bv) { bv = s_q[a]; best = a; }  // np.argmax: first max
    const int64_t act = (prm->dice[e] < prm->epsilon[e]) ? (int64_t)prm->random_action[e] : (int64_t)best;
    if (prm->store_action[e]) *reinterpret_cast<int64_t*>(ring_actions + prm->slot[e] * 8) = act;
  }
  // env.step(action): the next observation (synthetic source: independent of the action taken)
  if (e + 1 < prm->n_env) synth_transition(prm, e + 1, frames, rewards, masks, seed, done_period);
}

// ---- actor v3 (DRA_VAR_ACTOR_V3): 4 kernels per env step and no copy command in front of the graph.
//   env_frame_v3_kernel        : first kernel of the graph.  Copies this launch's parameter block from the pinned
//                                ring entry (*seq mod kAprmSlots) to the device block the later kernels read, bumps
//                                *seq, then produces the frame of env step 0.
//   actor_fc4_head_env_kernel  : fc4 GEMV as in v2; every workgroup publishes its 4 rows of h4, fences and takes a
//                                ticket; the LAST workgroup to arrive (all of h4 is then visible to it) computes the
//                                head, the epsilon-greedy action and the environment step -- one dependent launch
//                                less per env step.
__global__ void __launch_bounds__(256)
env_frame_v3_kernel(const uint8_t* __restrict__ ring_pinned, unsigned* __restrict__ seq, dra_dqn_step_params* __restrict__ prm_dev,
                    uint8_t* __restrict__ frames, double* __restrict__ rewards, int32_t* __restrict__ masks, uint64_t seed,
                    int done_period) {
  __shared__ uint32_t s_prm[kAprmStride / 4];
  constexpr int NW = (int)(kPrmHeadBytes / 4);
  const unsigned n = *seq;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(ring_pinned + (size_t)(n % kAprmSlots) * kAprmStride);
  if (threadIdx.x < NW) {
    const uint32_t v = src[threadIdx.x];
    s_prm[threadIdx.x] = v;
    reinterpret_cast<uint32_t*>(prm_dev)[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) *seq = n + 1;
  synth_transition(reinterpret_cast<const dra_dqn_step_params*>(s_prm), 0, frames, rewards, masks, seed, done_period);
}

__global__ void __launch_bounds__(256)
actor_fc4_head_env_kernel(const dra_dqn_step_params* __restrict__ prm, int e, const float* __restrict__ x,
                          const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ h4,
                          int in_features, unsigned* __restrict__ ticket, const float* __restrict__ wh,
                          const float* __restrict__ bh, int A, uint8_t* __restrict__ ring_actions, float* __restrict__ q_out,
                          uint8_t* __restrict__ frames, double* __restrict__ rewards, int32_t* __restrict__ masks,
                          uint64_t seed, int done_period) {
  constexpr int R = 13;  // float4 per lane: 3136 / 4 / 64 = 12.25
  __shared__ float s_q[64];
  __shared__ unsigned s_last;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    const int row = blockIdx.x * 4 + wave;
    const int nv = in_features >> 2;
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w + (int64_t)row * in_features);
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float4 wv[R], xv[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int i = min(lane + 64 * q, nv - 1);
      wv[q] = w4[i];
      xv[q] = x4[i];
    }
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      float4 a = wv[q], b = xv[q];
      asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));  // loads stay unconditional and batched
      if (lane + 64 * q < nv) acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = acc + bias[row];
      // agent-scope store: written through to memory, no cache-wide writeback needed to publish it
      __hip_atomic_store(h4 + row, v > 0.f ? v : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // publish, then take a ticket: the workgroup that draws the last one sees every other workgroup's rows.
  // (A __threadfence() here costs a full L2 writeback + invalidate per workgroup: measured +9 us per env step;
  // the rows are published with agent-scope stores and completed with s_waitcnt before the ticket instead.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  asm volatile("" ::: "memory");
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch starts from zero
  for (int a = wave; a < A; a += 4) {
    float hv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = __hip_atomic_load(h4 + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) part += hv[i] * wh[a * 512 + lane + 64 * i];
    part = wave_sum(part);
    if (lane == 0) s_q[a] = part + bh[a];
  }
  __syncthreads();
  if (threadIdx.x < A && q_out) q_out[threadIdx.x] = s_q[threadIdx.x];
  if (threadIdx.x == 0) {
    int best = 0;
    float bv = s_q[0];
    for (int a = 1; a < A; ++a) if (s_q[a] > bv) { bv = s_q[a]; best = a; }  // np.argmax: first max
    const int64_t act = (prm->dice[e] < prm->epsilon[e]) ? (int64_t)prm->random_action[e] : (int64_t)best;
    if (prm->store_action[e]) *reinterpret_cast<int64_t*>(ring_actions + prm->slot[e] * 8) = act;
  }
  if (e + 1 < prm->n_env) synth_transition(prm, e + 1, frames, rewards, masks, seed, done_period);
}

static int run_actor_steps_v3(dra_dqn_learner* l, int n_env, const float* P, hipStream_t st) {
  const dra_dqn_config& c = l->c;
  void *frames, *actions, *rewards, *masks;
  int rc = dra_ring_pointers(l->ring, &frames, &actions, &rewards, &masks);
  if (rc) return rc;
  const int64_t* o = c.offset;
  void* s = (void*)st;
  hipLaunchKernelGGL(env_frame_v3_kernel, dim3(1), dim3(256), 0, st, (const uint8_t*)l->aprm_ring, l->aprm_seq_dev,
                     l->prm_dev, (uint8_t*)frames, (double*)rewards, (int32_t*)masks, (uint64_t)c.env_seed,
                     (int)c.env_done_period);
  DRA_LAUNCH_CHECK();
  for (int e = 0; e < n_env; ++e) {
    if ((rc = dra_conv1_fwd_koc_ring(frames, &l->prm_dev->slot[e], &l->prm_dev->stack_age[e], c.ring_capacity, P + o[P_W1], P + o[P_B1], l->ay1,
                                     c.u8_coef, DRA_ACT_RELU, s)))
      return rc;
    const void* x2[1] = {l->ay1}; const float* w2[1] = {P + o[P_W2]}; const float* b2[1] = {P + o[P_B2]};
    float* y2[1] = {l->ay2};
    if ((rc = dra_conv_fwd_koc(2, 1, x2, w2, b2, y2, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    const void* x3[1] = {l->ay2}; const float* w3[1] = {P + o[P_W3]}; const float* b3[1] = {P + o[P_B3]};
    float* y3[1] = {l->ay3};
    if ((rc = dra_conv_fwd_koc(3, 1, x3, w3, b3, y3, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    if (l->variant & DRA_VAR_ACTOR_FUSED_HEAD) {
      hipLaunchKernelGGL(actor_fc4_head_env_kernel, dim3(128), dim3(256), 0, st, (const dra_dqn_step_params*)l->prm_dev, e,
                         (const float*)l->ay3, P + o[P_W4], P + o[P_B4], l->ah4, 3136, l->fc4_ticket, P + o[P_WH],
                         P + o[P_BH], c.n_actions, (uint8_t*)actions, l->aq, (uint8_t*)frames, (double*)rewards,
                         (int32_t*)masks, (uint64_t)c.env_seed, (int)c.env_done_period);
    } else {
      hipLaunchKernelGGL(actor_fc4_kernel, dim3(128), dim3(256), 0, st, (const float*)l->ay3, P + o[P_W4], P + o[P_B4],
                         l->ah4, 3136);
      hipLaunchKernelGGL(actor_head_env_kernel, dim3(1), dim3(c.head_kind != DRA_HEAD_VANILLA ? 1024 : 256), 0, st, (const dra_dqn_step_params*)l->prm_dev, e,
                         (const float*)l->ah4, P + o[P_WH], P + o[P_BH], c.n_actions, (uint8_t*)actions, l->aq,
                         (uint8_t*)frames, (double*)rewards, (int32_t*)masks, (uint64_t)c.env_seed, (int)c.env_done_period,
                         head_spec(l));
    }
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

// v3: the parameter block travels through the pinned ring (read by the graph's first kernel) instead of a copy
// command; entry = launch sequence number mod kAprmSlots, free again once the launch that read it has finished.
static int stage_actor_params(dra_dqn_learner* l, const dra_dqn_step_params* prm, hipStream_t st, bool before_launch) {
  const int k = (int)(l->aprm_seq % kAprmSlots);
  if (before_launch) {
    if (l->aprm_used[k]) DRA_HIP(hipEventSynchronize(l->aprm_ev[k]));
    memcpy(l->aprm_ring + (size_t)k * kAprmStride, prm, kPrmHeadBytes);
  } else {
    DRA_HIP(hipEventRecord(l->aprm_ev[k], st));
    l->aprm_used[k] = true;
    l->aprm_seq++;
  }
  return DRA_OK;
}

// q[a] of the head at batch 1 (one wave per action for VanillaNet; dist_head_q for the distributional heads).
// mail != null (an environment on the HOST: dra_dqn_learner_q_host / _q_host_async): the q values are ALSO stored into the
// host-mapped block mail[0..A) and then, behind a system-scope release, the number of forwards completed so far into mail[64]
// -- the host polls that word instead of waiting for a device-to-host copy and an event (two engine hand-overs, ~15 us).
__global__ void __launch_bounds__(1024)
head_q_kernel(const float* __restrict__ h4, const float* __restrict__ wh, const float* __restrict__ bh, int A,
              float* __restrict__ q_out, const HeadSpec hs, unsigned* __restrict__ flags_reset, int n_flags,
              float* __restrict__ mail, unsigned* __restrict__ seq_dev) {
  __shared__ float s_out[kMaxHeadOut];
  __shared__ float s_q[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (the arrival counters of the fused conv3 + fc4 launch in front of this kernel are done with: zero them for the next one)
  if (flags_reset)
    for (int i = threadIdx.x; i < n_flags; i += blockDim.x) __hip_atomic_store(flags_reset + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (hs.kind != DRA_HEAD_VANILLA) {
    dist_head_q(h4, wh, bh, A, hs, s_out, s_q);
  } else {
    for (int a = wave; a < A; a += (int)(blockDim.x >> 6)) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part += h4[lane + 64 * i] * wh[a * 512 + lane + 64 * i];
      part = wave_sum(part);
      if (lane == 0) s_q[a] = part + bh[a];
    }
    __syncthreads();
  }
  if (threadIdx.x < A) {            // (A <= 64: wave 0)
    const float q = s_q[threadIdx.x];
    q_out[threadIdx.x] = q;
    if (mail) __hip_atomic_store(mail + threadIdx.x, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (mail && wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the q stores of this wave have left
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const unsigned n = *seq_dev + 1u;
      *seq_dev = n;
      __hip_atomic_store(reinterpret_cast<unsigned*>(mail) + 64, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Host side of the mailbox: wait for forward number `want` (spin on the mapped word: the host IS the critical path of a
// host-environment actor, four dependent round trips per agent step), then copy the q values out.
static int q_mail_wait(dra_dqn_learner* l, unsigned want, float* q_host) {
  volatile unsigned* word = reinterpret_cast<volatile unsigned*>(l->q_stage) + 64;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint64_t spin = 1;; ++spin) {
    if (*word == want) break;
    if ((spin & 0xffff) == 0) {
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 10.0) return DRA_ETIMEDOUT;
      const hipError_t e = hipGetLastError();           // a failed launch would never publish
      if (e != hipSuccess) return (int)e;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const volatile float* q = l->q_stage;
  for (int a = 0; a < l->c.n_actions; ++a) q_host[a] = q[a];
  return DRA_OK;
}

// DQNActor._transition's forward (DQN_agent.py:29-33) for an environment that lives on the HOST: the caller's
// uint8 [4][84][84] observation is copied into mapped host memory that conv1 reads in place, the batch-1 forward of the
// ONLINE parameters runs as one captured graph (5 kernels) whose last kernel stores q[0..A) and a completion word into
// mapped host memory; the call returns when the word arrives (the reference's to_np(q) waits too: the action must reach the
// host emulator before it can step).  Round 3's form -- H2D copy, graph, D2H copy, event poll -- was 58 us per call against
// 34 us (profiles/r04g_prof_host_async_before.txt, r04h_prof_host_async.txt; the agent: 2.5 k -> 3.3 k updates/s).
DRA_API int dra_dqn_learner_q_host(dra_dqn_learner* l, const uint8_t* state_host, float* q_host, void* stream) {
  if (!l || !state_host || !q_host) return DRA_EINVAL;
  if (int rcf = flush_fc4(l, dra_stream(stream))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  hipStream_t st = dra_stream(stream);
  const dra_dqn_config& c = l->c;
  memcpy(l->qs_stage, state_host, (size_t)4 * 7056);         // (mapped, coherent: conv1 reads it in place)
  if (!l->g_q_ready) {
    const int64_t* o = c.offset;
    const float* P = l->p;
    void* s = (void*)st;
    hipGraph_t graph;
    DRA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = DRA_OK;
    {
      const void* x1[1] = {l->qs_stage};
      const float* w1[1] = {P + o[P_W1]}; const float* b1[1] = {P + o[P_B1]};
      float* y1[1] = {l->ay1};
      rc = dra_conv_fwd_koc(1, 1, x1, w1, b1, y1, 1, 1, c.u8_coef, DRA_ACT_RELU, s);
      const void* x2[1] = {l->ay1}; const float* w2[1] = {P + o[P_W2]}; const float* b2[1] = {P + o[P_B2]};
      float* y2[1] = {l->ay2};
      if (!rc) rc = dra_conv_fwd_koc(2, 1, x2, w2, b2, y2, 1, 0, 1.0, DRA_ACT_RELU, s);
      const void* x3[1] = {l->ay2}; const float* w3[1] = {P + o[P_W3]}; const float* b3[1] = {P + o[P_B3]};
      float* y3[1] = {l->ay3};
      if (!rc) rc = dra_conv_fwd_koc(3, 1, x3, w3, b3, y3, 1, 0, 1.0, DRA_ACT_RELU, s);
      if (!rc) {
        hipLaunchKernelGGL(actor_fc4_kernel, dim3(128), dim3(256), 0, st, (const float*)l->ay3, P + o[P_W4], P + o[P_B4],
                           l->ah4, 3136);
        hipLaunchKernelGGL(head_q_kernel, dim3(1), dim3(c.head_kind != DRA_HEAD_VANILLA ? 1024 : 256), 0, st, (const float*)l->ah4,
                           P + o[P_WH], P + o[P_BH], c.n_actions, l->aq, head_spec(l), (unsigned*)nullptr, 0,
                           l->q_stage, l->q_seq_dev);
      }
    }
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&l->g_q, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    l->g_q_ready = true;
  }
  DRA_HIP(hipGraphLaunch(l->g_q, st));
  return q_mail_wait(l, ++l->q_seq_host, q_host);
}

// ---- async actor over a HOST environment (BaseAgent.py:142-162 with a real emulator: the actor's forward for agent step
// t+1 runs while the learner trains on step t).  The update mirrors its new parameters into copy (t mod 2) of the actor's
// parameter copies; dra_dqn_learner_q_host_async runs the batch-1 forward of the copy update t-1 wrote on `stream_actor`
// (the actor's CU partition), so it neither waits for update t nor races with its optimizer.  Needs DRA_VAR_ACTOR_PARAMS.
DRA_API int dra_dqn_learner_update_async(dra_dqn_learner* l, int use_graph, int per, float beta, void* stream_update) {
  if (!l || !l->pa[0] || !l->pa[1]) return DRA_EINVAL;
  if (*l->timeout_flag) return DRA_ETIMEDOUT;
  if (int rcf = flush_fc4(l, dra_stream(stream_update))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  hipStream_t st = dra_stream(stream_update);
  l->profiling = false;
  l->pa_valid = false;
  const int k = (int)(l->hq_updates & 1);
  int rc = launch_gather(l, st);
  if (rc) return rc;
  rc = (use_graph && !per) ? body_graph(l, st) : ((use_graph && per && beta < 0.f) ? body_graph_per(l, st) : run_body(l, st, per, beta, 0));
  if (rc) return rc;
  if ((rc = launch_optimizer(l, st, l->pa[k]))) return rc;
  DRA_HIP(hipEventRecord(l->ev_hq[k], st));
  DRA_HIP(hipEventRecord(l->ev_step_done, st));
  l->last_done = l->ev_step_done;
  l->hq_updates++;
  l->hq_seeded = true;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_q_host_async(dra_dqn_learner* l, const uint8_t* state_host, float* q_host, void* stream_actor,
                                         void* stream_update) {
  if (!l || !state_host || !q_host || !l->pa[0] || !l->pa[1]) return DRA_EINVAL;
  if (int rcf = flush_fc4(l, dra_stream(stream_update))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  hipStream_t st = dra_stream(stream_actor);
  const dra_dqn_config& c = l->c;
  // copy the newest COMPLETED-BY-ORDER update wrote: update (hq_updates - 2) while update (hq_updates - 1) may still run
  int k;
  if (l->hq_updates >= 2) {
    k = (int)((l->hq_updates - 2) & 1);
    DRA_HIP(hipStreamWaitEvent(st, l->ev_hq[k], 0));
  } else {
    // before the second update: the online parameters as of now (after update 0 if it was issued), copied once per call site
    k = (int)(l->hq_updates & 1) ^ 1;     // the copy the NEXT update does not write
    hipStream_t su = dra_stream(stream_update);
    DRA_HIP(hipEventRecord(l->ev_join[3], su));
    DRA_HIP(hipStreamWaitEvent(st, l->ev_join[3], 0));
    if (l->hq_updates == 0) {
      DRA_HIP(hipMemcpyAsync(l->pa[k], l->p, (size_t)c.n_params * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
      k = 0;                              // exactly one update issued: it wrote copy 0; wait for it (no older copy exists)
      DRA_HIP(hipStreamWaitEvent(st, l->ev_hq[0], 0));
    }
  }
  memcpy(l->qs_stage, state_host, (size_t)4 * 7056);         // (mapped, coherent: conv1 reads it in place)
  if (!l->g_qa_ready[k]) {
    const int64_t* o = c.offset;
    const float* P = l->pa[k];
    void* s = (void*)st;
    hipGraph_t graph;
    DRA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = DRA_OK;
    {
      const void* x1[1] = {l->qs_stage};
      const float* w1[1] = {P + o[P_W1]}; const float* b1[1] = {P + o[P_B1]};
      float* y1[1] = {l->ay1};
      rc = dra_conv_fwd_koc(1, 1, x1, w1, b1, y1, 1, 1, c.u8_coef, DRA_ACT_RELU, s);
      // conv2 with its reduction split over two workgroups per tile, conv3 + fc4 as one launch (the device actor's kernels)
      if (!rc) rc = dra_conv_b1_split(2, l->ay1, nullptr, P + o[P_W2], P + o[P_B2], l->ay2p, s);
      if (!rc) rc = dra_actor_c3fc4(l->ay2p, P + o[P_W3], P + o[P_B3], P + o[P_W4], P + o[P_B4], l->ay3p, l->ah4,
                                    l->aflags + 4 * (kMaxEnvSteps - 1), l->timeout_flag, s);
      if (!rc) {
        hipLaunchKernelGGL(head_q_kernel, dim3(1), dim3(c.head_kind != DRA_HEAD_VANILLA ? 1024 : 256), 0, st, (const float*)l->ah4,
                           P + o[P_WH], P + o[P_BH], c.n_actions, l->aq, head_spec(l), l->aflags + 4 * (kMaxEnvSteps - 1), 4,
                           l->q_stage, l->q_seq_dev);
      }
    }
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&l->g_qa[k], graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    l->g_qa_ready[k] = true;
  }
  DRA_HIP(hipGraphLaunch(l->g_qa[k], st));
  return q_mail_wait(l, ++l->q_seq_host, q_host);
}

// ---- actor parameter ring (DRA_VAR_ACTOR_RING).  entry(seq) = ring + (seq mod kAringSlots) * kAprmStride holds the
// head of a dra_dqn_step_params.  actor_head_env_ring_kernel = actor_head_env_kernel reading its block through the
// device step counter; for the LAST env step of an agent step it also performs what the next agent step would
// start with -- the first observation of the next block (env.step's return value, envs.py:140-141) -- and then
// advances the counter.  env_frame_ring_kernel primes the very first block.
// synthetic observation `e` of block `prm` into a PENDING buffer (frame, reward, mask of the transition it starts)
__device__ __forceinline__ void synth_pending(const dra_dqn_step_params* __restrict__ prm, int e, uint8_t* __restrict__ frame,
                                              double* __restrict__ reward, int32_t* __restrict__ mask, uint64_t seed,
                                              int done_period) {
  const int64_t counter = prm->counter[e];
  if (counter < 0) return;
  uint64_t* dst = reinterpret_cast<uint64_t*>(frame);
  const uint64_t base = seed * 0x9E3779B97F4A7C15ull + (uint64_t)counter * 882ull;
  for (int w = threadIdx.x; w < 882; w += blockDim.x) dst[w] = mix64(base + (uint64_t)w);
  if (threadIdx.x == 0) {
    *reward = synth_reward(seed, prm->rcounter[e]);
    *mask = synth_mask(seed, prm->rcounter[e], done_period);
  }
}

__global__ void __launch_bounds__(256)
env_frame_ring_kernel(const uint8_t* __restrict__ ring, const unsigned* __restrict__ seq, uint8_t* __restrict__ pend_frame,
                      double* __restrict__ pend_reward, int32_t* __restrict__ pend_mask, uint64_t seed, int done_period) {
  synth_pending(aring_entry(ring, *seq), 0, pend_frame, pend_reward, pend_mask, seed, done_period);
}

// (launched with 1024 threads: the 882-word frame generation is the long pole of this one-workgroup kernel)
__global__ void __launch_bounds__(1024)
actor_head_env_ring_kernel(const uint8_t* __restrict__ ring, unsigned* __restrict__ seq, int e, int last, int commit,
                           const float* __restrict__ h4, const float* __restrict__ wh, const float* __restrict__ bh, int A,
                           uint8_t* __restrict__ ring_actions, float* __restrict__ q_out, uint8_t* __restrict__ frames,
                           double* __restrict__ rewards, int32_t* __restrict__ masks, uint8_t* __restrict__ pend_frame,
                           double* __restrict__ pend_reward, int32_t* __restrict__ pend_mask, uint64_t seed,
                           int done_period, const HeadSpec hs, unsigned* __restrict__ flags_reset, int n_flags) {
  __shared__ float s_q[64];
  __shared__ float s_out[kMaxHeadOut];
  DRA_STAMP(TR_A_HEAD, 0);
  // DRA_VAR_ACTOR_MEGA: the arrival counters of this agent step's one-launch env steps are done with; zero them for the next
  if (flags_reset)
    for (int i = threadIdx.x; i < n_flags; i += blockDim.x) __hip_atomic_store(flags_reset + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned sq = *seq;
  const dra_dqn_step_params* __restrict__ prm = aring_entry(ring, sq);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (commit && e == 0 && prm->counter[0] >= 0) {
    // feed: the observation this step started from becomes ring slot[0] now (not when it was produced: a
    // minibatch gathered in between must still see the slot's previous contents)
    const int64_t slot = prm->slot[0];
    const uint64_t* src = reinterpret_cast<const uint64_t*>(pend_frame);
    uint64_t* dst = reinterpret_cast<uint64_t*>(frames + slot * 7056);
    for (int w = threadIdx.x; w < 882; w += blockDim.x) dst[w] = src[w];
    if (threadIdx.x == 0) { rewards[slot] = *pend_reward; masks[slot] = *pend_mask; }
  }
  if (hs.kind != DRA_HEAD_VANILLA) {
    dist_head_q(h4, wh, bh, A, hs, s_out, s_q);
  } else {
    for (int a = wave; a < A; a += (int)(blockDim.x >> 6)) {
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part += h4[lane + 64 * i] * wh[a * 512 + lane + 64 * i];
      part = wave_sum(part);
      if (lane == 0) s_q[a] = part + bh[a];
    }
  }
  __syncthreads();   // (also: every thread is done reading the pending frame)
  DRA_STAMP(TR_A_HEAD, 2);
  if (threadIdx.x < A && q_out) q_out[threadIdx.x] = s_q[threadIdx.x];
  if (threadIdx.x == 0) {
    int best = 0;
    float bv = s_q[0];
    for (int a = 1; a < A; ++a) if (s_q[a] > bv) { bv = s_q[a]; best = a; }  // np.argmax: first max
    const int64_t act = (prm->dice[e] < prm->epsilon[e]) ? (int64_t)prm->random_action[e] : (int64_t)best;
    if (prm->store_action[e]) *reinterpret_cast<int64_t*>(ring_actions + prm->slot[e] * 8) = act;
  }
  if (!last) {
    synth_transition(prm, e + 1, frames, rewards, masks, seed, done_period);
  } else {
    synth_pending(aring_entry(ring, sq + 1), 0, pend_frame, pend_reward, pend_mask, seed, done_period);
    if (threadIdx.x == 0) *seq = sq + 1;   // every thread read *seq before the barrier above
  }
  DRA_STAMP(TR_A_HEAD, 5);
  DRA_STAMP_END(TR_A_HEAD);
}

// DRA_VAR_ACTOR_FUSED_CONV1: 4 launches per env step instead of 5.  The head of env step e-1 (action values,
// epsilon-greedy, action record) and the environment step that produces observation e run in front of conv1 of step e
// inside ONE launch (conv_v2.hip, ActorFuse): every conv1 workgroup derives the action and generates the rows of the new
// observation it convolves, one extra workgroup writes the action and the whole observation to the replay ring.  Env
// step 0's launch commits the pending observation instead; the head of the LAST env step keeps its own kernel (it also
// produces the next agent step's first observation and advances the step counter).  Same arithmetic, same order:
// bit-identical action values, actions and ring contents.
static constexpr int actor_ksplit() { return 1; }   // (conv2 / conv3 of the ring actor split along K over two workgroups; the
                                                    // DRA_ACTOR_KSPLIT=0 A/B switch of round 2 is retired)

static int run_actor_steps_ring_fused(dra_dqn_learner* l, int n_env, const float* P, hipStream_t st) {
  const dra_dqn_config& c = l->c;
  void *frames, *actions, *rewards, *masks;
  int rc = dra_ring_pointers(l->ring, &frames, &actions, &rewards, &masks);
  if (rc) return rc;
  const int64_t* o = c.offset;
  void* s = (void*)st;
  ActorFuse f;
  memset(&f, 0, sizeof(f));
  f.n_actions = c.n_actions; f.done_period = (int)c.env_done_period; f.h4 = l->ah4; f.wh = P + o[P_WH]; f.bh = P + o[P_BH];
  f.aring = l->aring_dev; f.seq = l->aring_seq; f.frames = (uint8_t*)frames; f.actions = (uint8_t*)actions;
  f.rewards = (double*)rewards; f.masks = (int32_t*)masks; f.q_out = l->aq; f.pend_frame = l->pend_frame;
  f.pend_reward = l->pend_reward; f.pend_mask = l->pend_mask; f.seed = (uint64_t)c.env_seed;
  const bool dist = c.head_kind != DRA_HEAD_VANILLA;
  f.head_kind = c.head_kind; f.n_atoms = c.n_atoms; f.atoms = l->atoms; f.pre = l->alog;
  // DRA_VAR_ACTOR_MEGA: conv3 + fc4 of an env step as ONE launch with an in-kernel hand-over (conv_v2.hip actor_c3fc4_kernel)
  const bool mega = (l->variant & DRA_VAR_ACTOR_MEGA) && l->aflags;   // (the K-split batch-1 convolutions: always since round 6)
  // DRA_VAR_ACTOR_PERSIST: the whole agent step as ONE launch (conv_v2.hip actor_persist.h).  Needs its 32 workgroups co-resident
  // (one per CU: a stream restricted to fewer CUs keeps the multi-launch form), the VanillaNet head and consecutive ring slots
  // (the host's feed order: replay.py:70-80; checked when the blocks are pushed)
  if (mega && (l->variant & DRA_VAR_ACTOR_PERSIST) && !dist && l->all_dev && n_env <= kMaxEnvSteps &&
      (l->actor_cus == 0 || l->actor_cus >= kPersistWgs)) {
    ActorPersistArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.w1 = P + o[P_W1]; pa.b1 = P + o[P_B1]; pa.w2 = P + o[P_W2]; pa.b2 = P + o[P_B2]; pa.w3 = P + o[P_W3]; pa.b3 = P + o[P_B3];
    pa.w4 = P + o[P_W4]; pa.b4 = P + o[P_B4]; pa.wh = P + o[P_WH]; pa.bh = P + o[P_BH];
    pa.frames = (uint8_t*)frames; pa.actions = (uint8_t*)actions; pa.rewards = (double*)rewards; pa.masks = (int32_t*)masks;
    pa.ring_cap = c.ring_capacity; pa.aring = l->aring_dev; pa.seq = l->aring_seq;
    pa.pend_frame = l->pend_frame; pa.pend_reward = l->pend_reward; pa.pend_mask = l->pend_mask;
    pa.q_out = l->aq; pa.h4_plain = l->ah4;
    pa.y1 = l->all_dev; pa.y2p = pa.y1 + kPersistY1; pa.y3p = pa.y2p + kPersistY2; pa.h4 = pa.y3p + kPersistY3;
    pa.abort_word = reinterpret_cast<int*>(l->all_dev + kPersistLLWords);
    pa.seed = (uint64_t)c.env_seed; pa.coef = c.u8_coef; pa.done_period = (int)c.env_done_period; pa.n_actions = c.n_actions;
    pa.n_env = n_env; pa.timeout_flag = l->timeout_flag;
    if (l->fs) { pa.fs_count = l->fs_count; pa.fs_need = l->fs_host + 8; pa.fs_done_host = l->fs_host; l->fs_persist_taken = true; }
    if (l->defer)
      for (int k = 0; k < 4; ++k)
        if (P == l->pa[k] && l->pa[k]) pa.w4_valid = defer_valid_word(l, k);
    return dra_actor_persist(&pa, s);
  }
  for (int e = 0; e < n_env; ++e) {
    const int64_t* slot_field = reinterpret_cast<const int64_t*>(l->aring_dev + offsetof(dra_dqn_step_params, slot)) + e;
    const int32_t* age_field = reinterpret_cast<const int32_t*>(l->aring_dev + offsetof(dra_dqn_step_params, stack_age)) + e;
    f.mode = e == 0 ? 1 : 2;
    f.e = e;
    if (mega) {
      if ((rc = dra_conv1_fwd_actor_fused(frames, slot_field, age_field, l->aring_seq, kAringSlots, (int64_t)kAprmStride, c.ring_capacity,
                                          e == 0 ? l->pend_frame : nullptr, P + o[P_W1], P + o[P_B1], l->ay1, c.u8_coef,
                                          DRA_ACT_RELU, &f, s)))
        return rc;
      if ((rc = dra_conv_b1_split(2, l->ay1, nullptr, P + o[P_W2], P + o[P_B2], l->ay2p, s))) return rc;
      // (DRA_VAR_DEFER_FC4: the first env step's fc4 waits for the word that says copy P's fc4 segment is complete)
      const int* w4_valid = nullptr;
      if (e == 0 && l->defer)
        for (int k = 0; k < 4; ++k)
          if (P == l->pa[k] && l->pa[k]) w4_valid = defer_valid_word(l, k);
      if ((rc = dra_actor_c3fc4_valid(l->ay2p, P + o[P_W3], P + o[P_B3], P + o[P_W4], P + o[P_B4], l->ay3p, l->ah4, l->aflags + 4 * e,
                                      l->timeout_flag, w4_valid, s)))
        return rc;
      if (dist) {
        hipLaunchKernelGGL(actor_dist_gemv_kernel, dim3((l->n_out + 3) / 4), dim3(256), 0, st, (const float*)l->ah4, P + o[P_WH],
                           P + o[P_BH], l->n_out, l->alog);
        DRA_LAUNCH_CHECK();
      }
      continue;
    }
    if ((rc = dra_conv1_fwd_actor_fused(frames, slot_field, age_field, l->aring_seq, kAringSlots, (int64_t)kAprmStride, c.ring_capacity,
                                        e == 0 ? l->pend_frame : nullptr, P + o[P_W1], P + o[P_B1], l->ay1, c.u8_coef,
                                        DRA_ACT_RELU, &f, s)))
      return rc;
    if (actor_ksplit()) {
      // conv2 / conv3 with their reduction halved over two workgroups per output tile; the partial planes are summed (+ bias,
      // ReLU) by the consumer's staging (conv_v2.hip conv_b1_split_kernel)
      if ((rc = dra_conv_b1_split(2, l->ay1, nullptr, P + o[P_W2], P + o[P_B2], l->ay2p, s))) return rc;
      if ((rc = dra_conv_b1_split(3, l->ay2p, l->ay2p + 64 * 81, P + o[P_W3], P + o[P_B3], l->ay3p, s))) return rc;
      // the input staged through LDS: one round of workgroups on the actor's CUs, same-box A/B 8 575 / 8 559 vs 8 409 / 8 242
      // updates/s against the register-resident form (profiles/r02zt_ab.json; that form and its switch were retired in round 6)
      hipLaunchKernelGGL(actor_fc4_planes_lds_kernel, dim3(128), dim3(256), 0, st, (const float*)l->ay3p,
                         (const float*)(l->ay3p + 64 * 49), P + o[P_W4], P + o[P_B4], l->ah4, 3136);
      DRA_LAUNCH_CHECK();
      if (dist) {   // the head's A*N outputs of this env step (consumed by the next launch: fused conv1 of e+1, or the tail kernel)
        hipLaunchKernelGGL(actor_dist_gemv_kernel, dim3((l->n_out + 3) / 4), dim3(256), 0, st, (const float*)l->ah4, P + o[P_WH],
                           P + o[P_BH], l->n_out, l->alog);
        DRA_LAUNCH_CHECK();
      }
      continue;
    }
    const void* x2[1] = {l->ay1}; const float* w2[1] = {P + o[P_W2]}; const float* b2[1] = {P + o[P_B2]};
    float* y2[1] = {l->ay2};
    if ((rc = dra_conv_fwd_koc(2, 1, x2, w2, b2, y2, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    const void* x3[1] = {l->ay2}; const float* w3[1] = {P + o[P_W3]}; const float* b3[1] = {P + o[P_B3]};
    float* y3[1] = {l->ay3};
    if ((rc = dra_conv_fwd_koc(3, 1, x3, w3, b3, y3, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    hipLaunchKernelGGL(actor_fc4_kernel, dim3(128), dim3(256), 0, st, (const float*)l->ay3, P + o[P_W4], P + o[P_B4],
                       l->ah4, 3136);
    DRA_LAUNCH_CHECK();
    if (dist) {
      hipLaunchKernelGGL(actor_dist_gemv_kernel, dim3((l->n_out + 3) / 4), dim3(256), 0, st, (const float*)l->ah4, P + o[P_WH],
                         P + o[P_BH], l->n_out, l->alog);
      DRA_LAUNCH_CHECK();
    }
  }
  HeadSpec hs_tail = head_spec(l);
  if (dist) { hs_tail.pre = l->alog; hs_tail.out = nullptr; }
  hipLaunchKernelGGL(actor_head_env_ring_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)l->aring_dev, l->aring_seq,
                     n_env - 1, 1, 0, (const float*)l->ah4, P + o[P_WH], P + o[P_BH], c.n_actions, (uint8_t*)actions, l->aq,
                     (uint8_t*)frames, (double*)rewards, (int32_t*)masks, l->pend_frame, l->pend_reward, l->pend_mask,
                     (uint64_t)c.env_seed, (int)c.env_done_period, hs_tail, mega ? l->aflags : (unsigned*)nullptr, kMaxEnvSteps * 4);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

static int run_actor_steps_ring(dra_dqn_learner* l, int n_env, const float* P, hipStream_t st) {
  if (l->variant & DRA_VAR_ACTOR_FUSED_CONV1) return run_actor_steps_ring_fused(l, n_env, P, st);
  const dra_dqn_config& c = l->c;
  void *frames, *actions, *rewards, *masks;
  int rc = dra_ring_pointers(l->ring, &frames, &actions, &rewards, &masks);
  if (rc) return rc;
  const int64_t* o = c.offset;
  void* s = (void*)st;
  for (int e = 0; e < n_env; ++e) {
    const int64_t* slot_field = reinterpret_cast<const int64_t*>(l->aring_dev + offsetof(dra_dqn_step_params, slot)) + e;
    const int32_t* age_field = reinterpret_cast<const int32_t*>(l->aring_dev + offsetof(dra_dqn_step_params, stack_age)) + e;
    if ((rc = dra_conv1_fwd_koc_ring_seq(frames, slot_field, age_field, l->aring_seq, kAringSlots, (int64_t)kAprmStride, c.ring_capacity,
                                         e == 0 ? l->pend_frame : nullptr, P + o[P_W1], P + o[P_B1], l->ay1, c.u8_coef,
                                         DRA_ACT_RELU, s)))
      return rc;
    const void* x2[1] = {l->ay1}; const float* w2[1] = {P + o[P_W2]}; const float* b2[1] = {P + o[P_B2]};
    float* y2[1] = {l->ay2};
    if ((rc = dra_conv_fwd_koc(2, 1, x2, w2, b2, y2, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    const void* x3[1] = {l->ay2}; const float* w3[1] = {P + o[P_W3]}; const float* b3[1] = {P + o[P_B3]};
    float* y3[1] = {l->ay3};
    if ((rc = dra_conv_fwd_koc(3, 1, x3, w3, b3, y3, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    hipLaunchKernelGGL(actor_fc4_kernel, dim3(128), dim3(256), 0, st, (const float*)l->ay3, P + o[P_W4], P + o[P_B4],
                       l->ah4, 3136);
    HeadSpec hs = head_spec(l);
    if (c.head_kind != DRA_HEAD_VANILLA) {
      hipLaunchKernelGGL(actor_dist_gemv_kernel, dim3((l->n_out + 3) / 4), dim3(256), 0, st, (const float*)l->ah4, P + o[P_WH],
                         P + o[P_BH], l->n_out, l->alog);
      hs.pre = l->alog;
      hs.out = nullptr;
    }
    hipLaunchKernelGGL(actor_head_env_ring_kernel, dim3(1), dim3(1024), 0, st, (const uint8_t*)l->aring_dev, l->aring_seq, e,
                       (int)(e == n_env - 1), 1, (const float*)l->ah4, P + o[P_WH], P + o[P_BH], c.n_actions, (uint8_t*)actions,
                       l->aq, (uint8_t*)frames, (double*)rewards, (int32_t*)masks, l->pend_frame, l->pend_reward, l->pend_mask,
                       (uint64_t)c.env_seed, (int)c.env_done_period, hs, (unsigned*)nullptr, 0);
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

DRA_API int dra_dqn_learner_set_actor_cus(dra_dqn_learner* l, int n_cus) {
  if (!l || n_cus < 0) return DRA_EINVAL;
  if (l->captured || l->step_no != 0 || l->aring_issued != 0) return DRA_EINVAL;   // the actor graphs bake the choice in
  l->actor_cus = n_cus;
  return DRA_OK;
}

// Uploads `n` parameter blocks (agent steps pushed, pushed+1, ...) into the device ring on `stream` (the actor
// stream: in order with the actor graphs).  At most kAringSlots / 2 blocks may be pending (pushed - issued).
DRA_API int dra_dqn_learner_actor_ring_push(dra_dqn_learner* l, const dra_dqn_step_params* blocks, int n, void* stream) {
  if (!l || !blocks || n < 1 || !l->aring_dev) return DRA_EINVAL;
  if ((int64_t)(l->aring_pushed - l->aring_issued) + n > kAringSlots / 2) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  for (int i = 0; i < n; ++i) {
    if (blocks[i].n_env < 1 || blocks[i].n_env > kMaxEnvSteps) return DRA_EINVAL;
    for (int e = 0; e < blocks[i].n_env; ++e)
      if (blocks[i].counter[e] < 0) return DRA_EINVAL;   // the ring actor owns the (device-resident) environment
    // (DRA_VAR_ACTOR_PERSIST keeps the frame stack in LDS across env steps: observation e is the frame of slot[e], the one
    // before it slot[e] - 1 -- the feed order of replay.py:70-80)
    if (l->variant & DRA_VAR_ACTOR_PERSIST)
      for (int e = 1; e < blocks[i].n_env; ++e)
        if (blocks[i].slot[e] != (blocks[i].slot[e - 1] + 1) % l->c.ring_capacity) return DRA_EINVAL;
    const size_t k = (size_t)((l->aring_pushed + i) % kAringSlots);
    memcpy(l->aring_stage + k * kAprmStride, &blocks[i], kPrmHeadBytes);
  }
  const size_t k0 = (size_t)(l->aring_pushed % kAringSlots);
  const size_t first = (k0 + n <= kAringSlots) ? (size_t)n : kAringSlots - k0;
  DRA_HIP(hipMemcpyAsync(l->aring_dev + k0 * kAprmStride, l->aring_stage + k0 * kAprmStride, first * kAprmStride,
                         hipMemcpyHostToDevice, st));
  if (first < (size_t)n)
    DRA_HIP(hipMemcpyAsync(l->aring_dev, l->aring_stage, ((size_t)n - first) * kAprmStride, hipMemcpyHostToDevice, st));
  l->aring_pushed += n;
  return DRA_OK;
}

// one agent step of actor transitions from the ring on `st` (graph per parameter-copy parity, or eager for P = online)
static int issue_actor_ring(dra_dqn_learner* l, int n_env, const float* P, int par, hipStream_t st, bool use_graph) {
  if (!l->aring_dev || l->aring_issued >= l->aring_pushed) return DRA_EINVAL;   // no block pushed for this step
  if (!l->aring_primed) {     // very first step: its first observation has no previous step to come from
    void *frames, *actions, *rewards, *masks;
    int rc0 = dra_ring_pointers(l->ring, &frames, &actions, &rewards, &masks);
    if (rc0) return rc0;
    hipLaunchKernelGGL(env_frame_ring_kernel, dim3(1), dim3(256), 0, st, (const uint8_t*)l->aring_dev,
                       (const unsigned*)l->aring_seq, l->pend_frame, l->pend_reward, l->pend_mask,
                       (uint64_t)l->c.env_seed, (int)l->c.env_done_period);
    DRA_LAUNCH_CHECK();
    l->aring_primed = true;
  }
  // DRA_VAR_FLAG_SYNC: what this agent step's launch waits for on the device (0: nothing -- the caller ordered the streams)
  if (l->fs_host) {
    __atomic_store_n(&l->fs_host[8 + (size_t)(l->aring_issued % kAringSlots)], (unsigned long long)l->fs_need_next, __ATOMIC_RELEASE);
    l->fs_need_next = 0;
  }
  int rc;
  if (!use_graph) {
    rc = run_actor_steps_ring(l, n_env, P, st);
  } else {
    if (l->g_aring_ready[par] && l->g_aring_nenv[par] != n_env) {
      (void)hipGraphExecDestroy(l->g_aring[par]);
      l->g_aring_ready[par] = false;
    }
    if (!l->g_aring_ready[par]) {
      hipGraph_t graph;
      DRA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      l->fs_persist_taken = false;
      rc = run_actor_steps_ring(l, n_env, P, st);
      l->fs_agraph[par] = l->fs_persist_taken && rc == DRA_OK;
      hipError_t e = hipStreamEndCapture(st, &graph);
      if (rc != DRA_OK) return rc;
      if (e != hipSuccess) return (int)e;
      DRA_HIP(hipGraphInstantiate(&l->g_aring[par], graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      l->g_aring_ready[par] = true;
      l->g_aring_nenv[par] = n_env;
    }
    DRA_HIP(hipGraphLaunch(l->g_aring[par], st));
    rc = DRA_OK;
  }
  if (rc == DRA_OK) l->aring_issued++;
  return rc;
}

static int run_actor_steps_v2(dra_dqn_learner* l, int n_env, const float* P, hipStream_t st) {
  const dra_dqn_config& c = l->c;
  void *frames, *actions, *rewards, *masks;
  int rc = dra_ring_pointers(l->ring, &frames, &actions, &rewards, &masks);
  if (rc) return rc;
  const int64_t* o = c.offset;
  void* s = (void*)st;
  hipLaunchKernelGGL(env_frame_kernel, dim3(1), dim3(256), 0, st, (const dra_dqn_step_params*)l->prm_dev, 0,
                     (uint8_t*)frames, (double*)rewards, (int32_t*)masks, (uint64_t)c.env_seed, (int)c.env_done_period);
  DRA_LAUNCH_CHECK();
  for (int e = 0; e < n_env; ++e) {
    if ((rc = dra_conv1_fwd_koc_ring(frames, &l->prm_dev->slot[e], &l->prm_dev->stack_age[e], c.ring_capacity, P + o[P_W1], P + o[P_B1], l->ay1,
                                     c.u8_coef, DRA_ACT_RELU, s)))
      return rc;
    const void* x2[1] = {l->ay1}; const float* w2[1] = {P + o[P_W2]}; const float* b2[1] = {P + o[P_B2]};
    float* y2[1] = {l->ay2};
    if ((rc = dra_conv_fwd_koc(2, 1, x2, w2, b2, y2, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    const void* x3[1] = {l->ay2}; const float* w3[1] = {P + o[P_W3]}; const float* b3[1] = {P + o[P_B3]};
    float* y3[1] = {l->ay3};
    if ((rc = dra_conv_fwd_koc(3, 1, x3, w3, b3, y3, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    hipLaunchKernelGGL(actor_fc4_kernel, dim3(128), dim3(256), 0, st, (const float*)l->ay3, P + o[P_W4], P + o[P_B4],
                       l->ah4, 3136);
    DRA_LAUNCH_CHECK();
    hipLaunchKernelGGL(actor_head_env_kernel, dim3(1), dim3(c.head_kind != DRA_HEAD_VANILLA ? 1024 : 256), 0, st, (const dra_dqn_step_params*)l->prm_dev, e,
                       (const float*)l->ah4, P + o[P_WH], P + o[P_BH], c.n_actions, (uint8_t*)actions, l->aq,
                       (uint8_t*)frames, (double*)rewards, (int32_t*)masks, (uint64_t)c.env_seed, (int)c.env_done_period,
                       head_spec(l));
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

static int run_actor_steps(dra_dqn_learner* l, int n_env, const float* P, hipStream_t st) {
  if (l->variant & DRA_VAR_ACTOR_V3) return run_actor_steps_v3(l, n_env, P, st);
  if (l->variant & DRA_VAR_ACTOR_V2) return run_actor_steps_v2(l, n_env, P, st);
  const dra_dqn_config& c = l->c;
  void *frames, *actions, *rewards, *masks;
  int rc = dra_ring_pointers(l->ring, &frames, &actions, &rewards, &masks);
  if (rc) return rc;
  const int64_t* o = c.offset;
  void* s = (void*)st;
  for (int e = 0; e < n_env; ++e) {
    hipLaunchKernelGGL(env_stack_kernel, dim3(4), dim3(256), 0, st, (const dra_dqn_step_params*)l->prm_dev, e,
                       (uint8_t*)frames, (double*)rewards, (int32_t*)masks, c.ring_capacity, (uint64_t)c.env_seed,
                       (int)c.env_done_period, l->act_state);
    DRA_LAUNCH_CHECK();
    const void* x1[1] = {l->act_state}; const float* w1[1] = {P + o[P_W1]}; const float* b1[1] = {P + o[P_B1]};
    float* y1[1] = {l->ay1};
    if ((rc = dra_conv_fwd_koc(1, 1, x1, w1, b1, y1, 1, 1, c.u8_coef, DRA_ACT_RELU, s))) return rc;
    const void* x2[1] = {l->ay1}; const float* w2[1] = {P + o[P_W2]}; const float* b2[1] = {P + o[P_B2]};
    float* y2[1] = {l->ay2};
    if ((rc = dra_conv_fwd_koc(2, 1, x2, w2, b2, y2, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    const void* x3[1] = {l->ay2}; const float* w3[1] = {P + o[P_W3]}; const float* b3[1] = {P + o[P_B3]};
    float* y3[1] = {l->ay3};
    if ((rc = dra_conv_fwd_koc(3, 1, x3, w3, b3, y3, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
    const float* x4[1] = {l->ay3}; const float* w4[1] = {P + o[P_W4]};
    if ((rc = dra_linear_fwd_slabs(1, x4, w4, 1, 3136, 512, kFc4Split, l->afc4_slabs, s))) return rc;
    hipLaunchKernelGGL(actor_head_kernel, dim3(1), dim3(512), 0, st, (const dra_dqn_step_params*)l->prm_dev, e,
                       (const float*)l->afc4_slabs, kFc4Split, P + o[P_B4], P + o[P_WH], P + o[P_BH], c.n_actions,
                       (uint8_t*)actions, l->aq, (int64_t*)nullptr);
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

static int actor_graph(dra_dqn_learner* l, int n_env, const float* P, hipStream_t st) {
  int slot = -1;
  constexpr int NS = 5;
  for (int k = 0; k < NS; ++k) if (l->g_actor[k].ready && l->g_actor[k].params == P && l->g_actor[k].n_env == n_env) slot = k;
  if (slot < 0) {
    for (int k = 0; k < NS && slot < 0; ++k) if (!l->g_actor[k].ready) slot = k;
    if (slot < 0) {  // evict an entry of the same parameter block (n_env changed), else entry 0
      slot = 0;
      for (int k = 0; k < NS; ++k) if (l->g_actor[k].params == P) slot = k;
      (void)hipGraphExecDestroy(l->g_actor[slot].exec);
      l->g_actor[slot].ready = false;
    }
    hipGraph_t graph;
    DRA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = run_actor_steps(l, n_env, P, st);
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&l->g_actor[slot].exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    l->g_actor[slot].ready = true;
    l->g_actor[slot].params = P;
    l->g_actor[slot].n_env = n_env;
  }
  DRA_HIP(hipGraphLaunch(l->g_actor[slot].exec, st));
  return DRA_OK;
}

static int actor_graph(dra_dqn_learner* l, int n_env, const float* P, hipStream_t st);
static int issue_actor(dra_dqn_learner* l, const dra_dqn_step_params* prm, int k, const float* P, hipStream_t st, bool use_graph);

#define TRACE(slot, stream)                                                                        \
  do {                                                                                             \
    if (l->tr_ev && l->tr_n < l->tr_cap) DRA_HIP(hipEventRecord(l->tr_ev[l->tr_n * 5 + (slot)], (stream))); \
  } while (0)

// Arms the timeline for the next `n_steps` pipelined async steps (n_steps = 0 frees it).
DRA_API int dra_dqn_learner_trace(dra_dqn_learner* l, int n_steps) {
  if (!l || n_steps < 0 || n_steps > 4096) return DRA_EINVAL;
  if (l->tr_ev) {
    for (int i = 0; i < l->tr_cap * 5; ++i) (void)hipEventDestroy(l->tr_ev[i]);
    delete[] l->tr_ev;
    l->tr_ev = nullptr;
  }
  l->tr_cap = l->tr_n = 0;
  if (n_steps == 0) return DRA_OK;
  l->tr_ev = new (std::nothrow) hipEvent_t[(size_t)n_steps * 5];
  if (!l->tr_ev) return DRA_ENOMEM;
  for (int i = 0; i < n_steps * 5; ++i) DRA_HIP(hipEventCreate(&l->tr_ev[i]));
  l->tr_cap = n_steps;
  return DRA_OK;
}

// Milliseconds of every traced event relative to the first one: out[step * 5 + slot].  Synchronises.
DRA_API int dra_dqn_learner_trace_read(dra_dqn_learner* l, float* out_ms, int max_steps, int* n_steps) {
  if (!l || !out_ms || !n_steps) return DRA_EINVAL;
  const int n = l->tr_n < max_steps ? l->tr_n : max_steps;
  *n_steps = n;
  if (n == 0) return DRA_OK;
  DRA_HIP(hipDeviceSynchronize());
  for (int i = 0; i < n * 5; ++i) DRA_HIP(hipEventElapsedTime(&out_ms[i], l->tr_ev[0], l->tr_ev[i]));
  return DRA_OK;
}

// async mode, DRA_VAR_PIPE_GATHER (needs DRA_VAR_ACTOR_PARAMS).  Two independent chains per agent step t:
//   actor stream  : gather(t) -> minibatch buffer t%2 ; actor graph of step t+1 (reads parameter copy (t+1)%2)
//   update stream : body(t) + optimizer(t) as one graph on minibatch buffer t%2 ; the optimizer also writes
//                   parameter copy t%2 (read by the actor graph of step t+2)
// The gather sits between the actor graph that wrote this step's transitions and the one that overwrites the
// oldest ring slots, in stream order -- no event on either chain's critical path: the waits below (minibatch
// buffer free again, optimizer of step t-1 done) refer to work issued a whole step earlier.
static int step_pipelined(dra_dqn_learner* l, const dra_dqn_step_params* prm, int do_update, hipStream_t su,
                          hipStream_t sa, int k) {
  const int B = l->c.batch;
  const int par = (int)(l->step_no & 1);
  int rc;
  if (do_update) {
    if (l->mb_used[par]) DRA_HIP(hipStreamWaitEvent(sa, l->ev_mb_free[par], 0));
    const int64_t* pinned = (l->variant & DRA_VAR_PINNED_IDX) ? l->idx_stage + (size_t)k * 1024 : nullptr;
    if (!pinned)
      DRA_HIP(hipMemcpyAsync(l->idx, l->idx_stage + (size_t)k * 1024, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, sa));
    TRACE(0, sa);
    l->gb = par;
    rc = launch_gather(l, sa, pinned);
    l->gb = 0;
    if (rc) return rc;
    DRA_HIP(hipEventRecord(l->ev_mb_ready[par], sa));
    TRACE(1, sa);
  }
  bool seeded = false;
  if (prm->n_env > 0) {
    if (l->pa_valid && l->pa_cur != (par ^ 1)) l->pa_valid = false;   // copies out of phase with the step parity
    if (l->last_done) DRA_HIP(hipStreamWaitEvent(sa, l->last_done, 0));   // the optimizer that produced the copy read below
    if (!l->pa_valid) {
      l->pa_cur = par ^ 1;
      DRA_HIP(hipMemcpyAsync(l->pa[l->pa_cur], l->p, (size_t)l->c.n_params * sizeof(float), hipMemcpyDeviceToDevice, sa));
      DRA_HIP(hipEventRecord(l->ev_join[0], sa));
      l->pa_valid = true;
      seeded = true;
    }
    if (l->variant & DRA_VAR_ACTOR_RING) rc = issue_actor_ring(l, prm->n_env, l->pa[l->pa_cur], l->pa_cur, sa, true);
    else rc = issue_actor(l, prm, k, l->pa[l->pa_cur], sa, true);
    if (rc) return rc;
    l->actor_pending = true;     // (nothing waits for an 'actor done' event on this path: stream order + stage_ev[k] below)
    TRACE(2, sa);
  }
  DRA_HIP(hipEventRecord(l->stage_ev[k], sa));  // staging slot k: parameter block copy and the gather's pinned index reads
  if (prm->n_env > 0) l->actor_last = l->stage_ev[k];
  if (do_update) {
    DRA_HIP(hipStreamWaitEvent(su, l->ev_mb_ready[par], 0));
    if (seeded) DRA_HIP(hipStreamWaitEvent(su, l->ev_join[0], 0));    // the seed copy read the parameters this step overwrites
    TRACE(3, su);
    if ((rc = pipe_graph(l, su, par))) return rc;
    TRACE(4, su);
    if (l->tr_ev && l->tr_n < l->tr_cap) l->tr_n++;
    // ONE event per step on the update stream (each stream-level event costs ~3.5 us of queue time between two graph
    // launches: profiles/r02d_*): it means both 'minibatch buffer par is free' and 'optimizer of this step is done'
    DRA_HIP(hipEventRecord(l->ev_mb_free[par], su));
    l->last_done = l->ev_mb_free[par];
    l->mb_used[par] = true;
    if (l->pa_valid) l->pa_cur = par;   // the graph's optimizer wrote copy `par`
    l->step_no++;
  }
  return DRA_OK;
}

// async mode, DRA_VAR_GATHER_ON_UPDATE (on top of PIPE_GATHER + ACTOR_PARAMS): the gather moves to the UPDATE stream.
// The phase traces (profiles/r02x_phase_async.json) show the actor chain -- 4 env steps of 4 launches + tail + gather and
// three stream-level event records -- is the longer one (update stream idle ~18 us per step); here the actor stream
// carries only [wait optimizer t-1] [actor graph t+1] [one record], the update stream
// [wait actor graph t] [gather t] [update graph t] [one record].
// The write-after-read hazard the actor-stream gather excluded by stream order (actor graph t+1 overwrites the OLDEST
// ring slots, which minibatch t may still sample once the ring is full) is decided on the host, which knows both the
// indices and the slots: in that (rare: ~1e-4 of the steps at 10^6 slots) case the actor graph also waits for this
// step's update.  Results are bit-identical to step_pipelined.
static bool gather_reads_slots(const dra_dqn_learner* l, const int64_t* idx, const int64_t* slots, int n_slots) {
  int hh = 4, nn = 1;
  (void)dra_ring_shape(l->ring, &hh, &nn);
  const int64_t h = hh, n = nn;
  for (int e = 0; e < n_slots; ++e) {
    const int64_t s = slots[e];
    for (int b = 0; b < l->c.batch; ++b)
      if (s >= idx[b] - h + 1 && s <= idx[b] + n) return true;
  }
  return false;
}

// Bookkeeping of the actor launches an update may have to wait for.  The update stream no longer waits for "the previous
// actor graph" every step, so the host must know, for every actor launch that may still be running, which ring slots it
// writes: a minibatch that touches the slots of launch i waits for launch i (which, the actor stream being in order,
// covers every older one).  During the exploration phase the host issues actor-only calls far ahead of the device; the
// first updates then meet several unfinished launches, not just the last one.
static void arec_push(dra_dqn_learner* l, hipEvent_t ev, const int64_t* slots, int n) {
  int w = 0;
  for (int i = 0; i < l->arec_count; ++i)        // the staging slot's previous user is complete (stage_acquire synchronised it)
    if (l->arec[i].ev != ev) l->arec[w++] = l->arec[i];
  l->arec_count = w;
  if (l->arec_count == 8) {                       // (cannot happen with 8 staging slots; keep the newest 7)
    for (int i = 1; i < 8; ++i) l->arec[i - 1] = l->arec[i];
    l->arec_count = 7;
  }
  auto& r = l->arec[l->arec_count++];
  r.ev = ev;
  r.n = n;
  for (int e = 0; e < n && e < 8; ++e) r.slots[e] = slots[e];
}

// the newest unfinished actor launch whose slots the minibatch `idx` touches (null: none)
static hipEvent_t arec_needed(dra_dqn_learner* l, const int64_t* idx) {
  int first = 0;
  while (first < l->arec_count && hipEventQuery(l->arec[first].ev) == hipSuccess) ++first;   // finished launches, oldest first
  if (first > 0) {
    for (int i = first; i < l->arec_count; ++i) l->arec[i - first] = l->arec[i];
    l->arec_count -= first;
  }
  for (int i = l->arec_count - 1; i >= 0; --i)
    if (l->arec[i].n < 0 || gather_reads_slots(l, idx, l->arec[i].slots, l->arec[i].n)) return l->arec[i].ev;
  return nullptr;
}

// phase 0: the whole step.  phase 1 / 2: the same step in two calls for the device-drawn prioritized minibatch
// (dra_dqn_learner_step_update / _step_actor): 1 = everything of the update (its indices are in device memory, so the host
// cannot test them: it waits for the newest actor launch unconditionally), 2 = the actor launch + the step's bookkeeping,
// prm->idx = the indices of the update issued by phase 1 (known to the host by then) for the slot hazard check.
static int step_pipelined3(dra_dqn_learner* l, const dra_dqn_step_params* prm, int do_update, hipStream_t su,
                           hipStream_t sa, int k, int phase = 0) {
  const int B = l->c.batch;
  // Parameter copies rotate by FOUR here: update t writes copy t mod 4, the actor graph of step t+1 (issued with update t)
  // reads the copy update t-1 wrote.  The copy update t overwrites was last read by the actor graph issued three calls ago;
  // nothing on the device orders the update after that graph any more (the per-step wait is gone), so the HOST makes sure it
  // is done before issuing the update (pa_reader: normally long complete -- the host runs at most four steps ahead).
  const int q = (int)(l->step_no & 3), qr = (q + 3) & 3;
  const int par = q & 1;           // minibatch buffers alternate
  int rc;
  // DRA_VAR_DEFER_FC4: a pending fc4 segment stays pending only if this call issues the riding graph that completes it (a plain
  // ring-direct update, the actor copies in phase with the rotation); every other call steps it first -- before `opt_prev` is
  // read: the actor launch below then waits for the flush
  if (l->defer_host && phase != 2) {
    const bool riding = do_update && phase == 0 && l->defer && !l->step_per && (l->variant & DRA_VAR_RING_DIRECT) &&
                        l->defer_q == qr && (prm->n_env <= 0 || (l->pa_valid && l->pa_cur == qr));
    if (!riding && (rc = flush_fc4(l, su))) return rc;
  }
  hipEvent_t opt_prev = phase == 2 ? l->split_opt_prev : l->last_done;   // optimizer of step t-1: produced the copy the actor graph below reads
  // the block the actor graph issued by THIS call consumes: with the parameter ring that is the next un-issued ring entry
  // (pushed up to 16 agent steps ahead; `prm` then only carries n_env and the minibatch indices)
  const dra_dqn_step_params* ablk = prm;
  if ((l->variant & DRA_VAR_ACTOR_RING) && prm->n_env > 0) {
    if (!l->aring_stage || l->aring_issued >= l->aring_pushed) return DRA_EINVAL;
    ablk = reinterpret_cast<const dra_dqn_step_params*>(l->aring_stage + (size_t)(l->aring_issued % kAringSlots) * kAprmStride);
  }
  const bool hazard = phase != 1 && do_update && prm->n_env > 0 && gather_reads_slots(l, prm->idx, ablk->slot, prm->n_env);
  // ... and the other direction: the update reads the transitions an unfinished actor launch is writing only if the
  // minibatch touches its slots (same order of probability); otherwise the two chains do not meet in this step at all
  const hipEvent_t needs_actor = !do_update || phase == 2 ? nullptr : (phase == 1 ? (l->actor_pending ? l->actor_last : nullptr)
                                                                                  : arec_needed(l, prm->idx));
  bool seed = phase == 2 ? l->split_seed : false;
  if (phase == 1) { l->split_opt_prev = opt_prev; l->split_q = q; }
  if (phase != 2 && prm->n_env > 0) {
    if (l->pa_valid && l->pa_cur != qr) l->pa_valid = false;   // copies out of phase with the step rotation
    seed = !l->pa_valid;
  }
  if (seed && phase != 2) {   // (re)seed the actor copy from the online parameters BEFORE this step's optimizer overwrites them
    if (opt_prev) DRA_HIP(hipStreamWaitEvent(sa, opt_prev, 0));
    l->pa_cur = qr;
    DRA_HIP(hipMemcpyAsync(l->pa[l->pa_cur], l->p, (size_t)l->c.n_params * sizeof(float), hipMemcpyDeviceToDevice, sa));
    DRA_HIP(hipEventRecord(l->ev_join[0], sa));
    l->pa_valid = true;
  }
  if (phase == 1) l->split_seed = seed;
  if (do_update && phase != 2) {
    if (l->pa_reader[q]) {           // the optimizer of this update overwrites copy q: its last reader must be done
      const auto t0 = std::chrono::steady_clock::now();
      DRA_HIP(hipEventSynchronize(l->pa_reader[q]));
      l->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      l->pa_reader[q] = nullptr;
    }
    if (needs_actor) DRA_HIP(hipStreamWaitEvent(su, needs_actor, 0));   // the sampled transitions are in the ring
    if (seed) DRA_HIP(hipStreamWaitEvent(su, l->ev_join[0], 0));
    if (l->variant & DRA_VAR_RING_DIRECT) {
      // no gather: the update's own kernels read the ring through idx_pin[q] (four rotating pinned buffers: a captured graph
      // bakes the address; buffer q is free again once update t-4 is done)
      if (l->upd_used[q]) {
        const auto t0 = std::chrono::steady_clock::now();
        DRA_HIP(hipEventSynchronize(l->ev_upd[q]));
        l->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      if (phase == 0) memcpy(l->idx_pin[q], prm->idx, (size_t)B * sizeof(int64_t));
      if (phase == 0 && (l->variant & DRA_VAR_IDX_PREFETCH)) {
        // a step-tagged copy of the indices travels to the device on the side stream, ordered with NOTHING: conv1 uses an
        // element only if its tag is this update's (then the copy landed in time, the normal case -- the host runs ahead
        // of the device), else it reads the pinned indices as before.  (A first version let the previous update's head
        // kernel fetch them over PCIe: conv1 -1.2 us, but the in-order vmcnt made the head kernel wait for that read:
        // +1.9 us, profiles/r02zt_*.)
        const uint64_t tag = ((l->rd_issued + 1ull) & 0xffffffull) << 40;
        volatile int64_t* dst = l->idx_tag_pin[q];
        for (int b = 0; b < B; ++b) dst[b] = (int64_t)((uint64_t)prm->idx[b] | tag);
        DRA_HIP(hipMemcpyAsync(l->idx_tag_dev + (size_t)q * 1024, l->idx_tag_pin[q], (size_t)B * sizeof(int64_t),
                               hipMemcpyHostToDevice, l->side));
      }
      l->rd_issued++;
      TRACE(0, su);
      if (l->keep_minibatch) {
        l->gb = par;
        rc = launch_gather(l, su, phase == 1 ? l->per2_idx + (size_t)q * 1024 : l->idx_pin[q]);
        l->gb = 0;
        if (rc) return rc;
      } else {
        l->last_gb = par;
      }
      TRACE(1, su);
      TRACE(3, su);
      if ((rc = rd_graph(l, su, q, l->step_per))) return rc;
      TRACE(4, su);
      DRA_HIP(hipEventRecord(l->ev_upd[q], su));         // the ONE record of the update stream: optimizer t done
      l->last_done = l->ev_upd[q];
      l->upd_used[q] = true;
    } else {
      const int64_t* pinned = (l->variant & DRA_VAR_PINNED_IDX) ? l->idx_stage + (size_t)k * 1024 : nullptr;
      if (!pinned)
        DRA_HIP(hipMemcpyAsync(l->idx, l->idx_stage + (size_t)k * 1024, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, su));
      TRACE(0, su);
      l->gb = par;
      rc = launch_gather(l, su, pinned);
      l->gb = 0;
      if (rc) return rc;
      TRACE(1, su);
      TRACE(3, su);
      if ((rc = pipe_graph(l, su, q, l->step_per))) return rc;
      TRACE(4, su);
      DRA_HIP(hipEventRecord(l->ev_mb_free[par], su));   // the ONE record of the update stream: optimizer t done
      l->last_done = l->ev_mb_free[par];
      l->mb_used[par] = true;
    }
  }
  if (phase == 1) return DRA_OK;
  if (prm->n_env > 0) {
    if (!seed && opt_prev) DRA_HIP(hipStreamWaitEvent(sa, opt_prev, 0));
    if (hazard) DRA_HIP(hipStreamWaitEvent(sa, l->last_done, 0));      // the gather reads slots this graph overwrites
    const int cur = l->pa_cur;
    if (l->variant & DRA_VAR_ACTOR_RING) rc = issue_actor_ring(l, prm->n_env, l->pa[cur], cur, sa, true);
    else rc = issue_actor(l, prm, k, l->pa[cur], sa, true);
    if (rc) return rc;
    l->actor_pending = true;
    TRACE(2, sa);
  }
  DRA_HIP(hipEventRecord(l->stage_ev[k], sa));   // the ONE record of the actor stream: graph done, staging slot k free
  if (prm->n_env > 0) {
    l->pa_reader[l->pa_cur] = l->stage_ev[k];
    l->actor_last = l->stage_ev[k];
    arec_push(l, l->stage_ev[k], ablk->slot, prm->n_env);
  }
  if (do_update) {
    if (l->tr_ev && l->tr_n < l->tr_cap) l->tr_n++;
    if (l->pa_valid) l->pa_cur = q;   // the graph's optimizer wrote copy q
    l->step_no++;
  }
  return DRA_OK;
}

// Actor transitions of `prm` on `st`: parameter block (copy command, or the pinned ring of actor v3) + the
// captured graph (or the eager kernels).
static int issue_actor(dra_dqn_learner* l, const dra_dqn_step_params* prm, int k, const float* P, hipStream_t st, bool use_graph) {
  int rc;
  if (!(l->variant & (DRA_VAR_ACTOR_V2 | DRA_VAR_ACTOR_V3)))     // the first-generation actor copies the last 4 ring frames
    for (int e = 0; e < prm->n_env; ++e) if (prm->stack_age[e] != 3) return DRA_EINVAL;
  const bool v3 = l->variant & DRA_VAR_ACTOR_V3;
  if (v3) { if ((rc = stage_actor_params(l, prm, st, true))) return rc; }
  else DRA_HIP(hipMemcpyAsync(l->prm_dev, &l->prm_stage[k], kPrmHeadBytes, hipMemcpyHostToDevice, st));
  rc = use_graph ? actor_graph(l, prm->n_env, P, st) : run_actor_steps(l, prm->n_env, P, st);
  if (rc) return rc;
  return v3 ? stage_actor_params(l, prm, st, false) : DRA_OK;
}

// async mode, DRA_VAR_GATHER_IN_GRAPH (on top of PIPE_GATHER + ACTOR_PARAMS).  Call k carries the transitions of
// step k AND the minibatch indices of step k (the host draws them in the reference's order: actor randomness of a
// step, then its sample).  Per call:
//   actor stream  : ONE copy (parameter block + indices, contiguous in dra_dqn_step_params) and ONE graph
//                   [actor transitions of step k ... gather of step k -> minibatch buffer k%2]
//   update stream : body + optimizer of step k-1 (one graph) on the minibatch gathered by the previous call
// Same dependencies as step_pipelined (gather(k) after actor(k), before actor(k+1); actor(k) on the parameters of
// optimizer k-2; optimizer k-1 after actor(k-1)), hence bit-identical results; one launch boundary and one event
// less per step on the actor chain.  A call with n_env == 0 flushes: it only issues the pending update.
static int ag_graph(dra_dqn_learner* l, int n_env, int par, hipStream_t st) {
  if (l->g_ag_ready[par] && l->g_ag_nenv[par] != n_env) {
    (void)hipGraphExecDestroy(l->g_ag[par]);
    l->g_ag_ready[par] = false;
  }
  if (!l->g_ag_ready[par]) {
    hipGraph_t graph;
    hipError_t b = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (b != hipSuccess) return (int)b;
    int rc = run_actor_steps(l, n_env, l->pa[par], st);
    if (rc == DRA_OK) {
      l->gb = par;
      rc = launch_gather(l, st, l->prm_dev->idx);
      l->gb = 0;
    }
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&l->g_ag[par], graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    l->g_ag_ready[par] = true;
    l->g_ag_nenv[par] = n_env;
  }
  DRA_HIP(hipGraphLaunch(l->g_ag[par], st));
  return DRA_OK;
}

static int step_pipelined2(dra_dqn_learner* l, const dra_dqn_step_params* prm, int do_update, hipStream_t su,
                           hipStream_t sa, int k) {
  const int B = l->c.batch;
  int rc;
  bool seeded = false;
  const int par = (int)(l->step_no & 1);
  if (prm->n_env > 0) {
    if (l->mb_used[par]) DRA_HIP(hipStreamWaitEvent(sa, l->ev_mb_free[par], 0));   // update k-2 is done with buffer par
    memcpy(l->prm_stage[k].idx, prm->idx, (size_t)B * sizeof(int64_t));
    DRA_HIP(hipMemcpyAsync(l->prm_dev, &l->prm_stage[k], kPrmHeadBytes + (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, sa));
    TRACE(0, sa);
    if (l->pa_valid && l->pa_cur != par) l->pa_valid = false;
    if (l->last_done) DRA_HIP(hipStreamWaitEvent(sa, l->last_done, 0));   // optimizer k-2 produced the copy this graph reads
    if (!l->pa_valid) {
      l->pa_cur = par;
      DRA_HIP(hipMemcpyAsync(l->pa[par], l->p, (size_t)l->c.n_params * sizeof(float), hipMemcpyDeviceToDevice, sa));
      DRA_HIP(hipEventRecord(l->ev_join[0], sa));
      l->pa_valid = true;
      seeded = true;
    }
    if ((rc = ag_graph(l, prm->n_env, par, sa))) return rc;
    DRA_HIP(hipEventRecord(l->ev_mb_ready[par], sa));
    DRA_HIP(hipEventRecord(l->ev_actor_done, sa));
    l->actor_pending = true;
    TRACE(1, sa);
    TRACE(2, sa);
  }
  DRA_HIP(hipEventRecord(l->stage_ev[k], sa));
  if (do_update && l->ag_have_prev) {
    const int q = l->ag_prev_par;
    DRA_HIP(hipStreamWaitEvent(su, l->ev_mb_ready[q], 0));
    if (seeded) DRA_HIP(hipStreamWaitEvent(su, l->ev_join[0], 0));
    TRACE(3, su);
    if ((rc = pipe_graph(l, su, q))) return rc;
    TRACE(4, su);
    if (l->tr_ev && l->tr_n < l->tr_cap) l->tr_n++;
    DRA_HIP(hipEventRecord(l->ev_mb_free[q], su));   // one record: minibatch buffer q free AND optimizer done
    l->last_done = l->ev_mb_free[q];
    l->mb_used[q] = true;
    if (l->pa_valid) l->pa_cur = q;     // the optimizer wrote copy q = the one the NEXT call's graph reads
    l->ag_have_prev = false;
  }
  if (prm->n_env > 0) {
    l->ag_have_prev = true;
    l->ag_prev_par = par;
    l->step_no++;
  }
  return DRA_OK;
}

// ---- DRA_VAR_FLAG_SYNC: the steady-state pipelined step without events ------------------------------------------------------------
// tools/ubench/graph_gap.hip (profiles/r06zi_graph_gap.jsonl): with the update as one graph per step, the event record behind it
// costs 4.9 us of the update stream's time and the actor stream's wait on that event another 9.4 us -- per step, on the chain that
// bounds the rate.  In the lane neither stream records or waits for anything:
//   * update t's first launch counts itself in fs_count when it STARTS -- which says that everything the update stream ran before
//     it, update t-1's optimizer included, is complete and written back (a launch boundary);
//   * the actor launch issued with update t reads the copy update t-1 wrote: its workgroups poll fs_count >= `need` (the host leaves
//     the number of update t in a pinned ring indexed by the agent step) instead of the stream waiting for an event;
//   * the host never runs more than three calls ahead: before it issues update t it polls the pinned count of agent steps the
//     actor has completed until the launch issued with update t-3 is done -- that launch read the copy update t's optimizer
//     overwrites, and it started only after update t-4 was complete, which frees the pinned index buffer of rotation slot q;
//   * the two rare slot hazards (DQN_agent.py:101-127 fed and sampled in one step) keep their meaning: a minibatch that reads slots
//     an unfinished actor launch writes makes the HOST wait for the actor stream; an actor launch that overwrites slots this
//     call's minibatch reads waits for one more count, which a one-thread launch behind the update graph provides.
// Every other entry point of the learner leaves the lane first (fs_leave: both streams drained, the event paths' bookkeeping reset
// to "nothing pending"), so the event paths never meet un-recorded work.
__global__ void fs_bump_kernel(unsigned long long* count) { __hip_atomic_fetch_add(count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static int fs_leave(dra_dqn_learner* l, hipStream_t su, hipStream_t sa);
static int fs_leave_any(dra_dqn_learner* l) { return l->fs_on ? fs_leave(l, l->fs_su, l->fs_sa) : DRA_OK; }
static int fs_leave(dra_dqn_learner* l, hipStream_t su, hipStream_t sa) {
  if (!l->fs_on) return DRA_OK;
  l->fs_on = false;
  DRA_HIP(hipStreamSynchronize(su));     // (every count an actor launch polls for comes from work already issued on `su`)
  DRA_HIP(hipStreamSynchronize(sa));
  if (l->ah && l->ah_stream) {
    // DRA_VAR_TARGET_AHEAD: an ahead sequence still in flight completes (its gate's count was issued on `su`); a stash nobody
    // consumed is dropped (the next lane step computes its target in line).  The lane's chains carried the online net alone: the
    // update's chain counters are brought level for the event paths' two-net launches (every counter holds epoch x arrivals
    // between updates: all zero is a level state)
    DRA_HIP(hipStreamSynchronize(l->ah_stream));
    l->ah_valid = false;
    l->ah_next_set = false;
    DRA_HIP(hipMemsetAsync(l->fchain_dev, 0, (size_t)(kFwdChainCounters + 1) * sizeof(unsigned), su));
    DRA_HIP(hipMemsetAsync(l->bchain_dev, 0, (size_t)dra_bwd_chain_counters() * sizeof(unsigned), su));
    DRA_HIP(hipStreamSynchronize(su));
  }
  l->last_done = nullptr;
  for (int q = 0; q < 4; ++q) { l->pa_reader[q] = nullptr; l->upd_used[q] = false; l->fs_reader[q] = 0; }
  for (int k = 0; k < 8; ++k) l->stage_used[k] = false;
  l->arec_count = 0;
  l->fs_arec_n = 0;
  DRA_HIP(hipEventRecord(l->ev_fs, sa));
  l->actor_last = l->ev_fs;
  return DRA_OK;
}

static inline uint64_t fs_actor_done(const dra_dqn_learner* l) { return __atomic_load_n(&l->fs_host[0], __ATOMIC_ACQUIRE); }

// the 32-bit device count against a 64-bit host target (the device counter wraps, the distance never exceeds a few steps)
static inline bool fs_reached(uint64_t done32, uint64_t target) { return (uint32_t)((uint32_t)done32 - (uint32_t)target) < 0x80000000u; }

static int fs_wait_actor(dra_dqn_learner* l, uint64_t target) {
  if (fs_reached(fs_actor_done(l), target)) return DRA_OK;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    for (int i = 0; i < 64; ++i) {
      if (fs_reached(fs_actor_done(l), target)) {
        l->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return DRA_OK;
      }
      __builtin_ia32_pause();
    }
    if (*l->timeout_flag) return DRA_ETIMEDOUT;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) return DRA_ETIMEDOUT;
  }
}

// can THIS call run in the lane?  (everything that would make step_pipelined3 do more than [update graph, actor graph])
static bool fs_eligible(const dra_dqn_learner* l, const dra_dqn_step_params* prm, int do_update, void* stream_actor) {
  if (!l->fs || !stream_actor || !do_update || prm->n_env < 1 || l->step_per || l->keep_minibatch || l->tr_ev) return false;
  const int q = (int)(l->step_no & 3), qr = (q + 3) & 3;
  if (!l->pa_valid || l->pa_cur != qr) return false;                       // (a seed copy: event path)
  if (!l->g_rd_ready[q] || !l->fs_graph[q]) return false;                  // (the first four updates capture on the event path)
  if (!l->g_aring_ready[qr] || !l->fs_agraph[qr] || l->g_aring_nenv[qr] != prm->n_env) return false;
  if (!l->aring_primed || l->aring_issued >= l->aring_pushed) return false;
  if (l->defer ? (l->defer_host && l->defer_q != qr) : l->defer_host) return false;   // (a pending fc4 segment this graph's riders do not finish)
  return true;
}

// DRA_VAR_LANE_EAGER: update t as its six plain launches instead of one graph replay -- exactly what capture_part records for the
// plain ring-direct graph of rotation slot q (riders for copy q - 1, announce, optimizer into copy q, fc4's segment deferred), with
// update_graph's bookkeeping.  A graph replay costs 6 us before its first kernel, a plain dependent launch 1.3 us
// (profiles/r06zi_graph_gap.jsonl); the host pays ~4 us per launch instead, which it has (it paces itself three calls ahead).
static int rd_eager(dra_dqn_learner* l, hipStream_t st, int q) {
  l->gb = q & 1;
  l->rd_slot = q;
  l->rider_q = l->defer ? ((q + 3) & 3) : -1;
  l->fs_capturing = true;
  int rc = run_body(l, st, 0, 0.f, 0, 0);
  l->fs_capturing = false;
  l->rider_q = -1;
  if (rc == DRA_OK) rc = launch_optimizer(l, st, l->pa[q], l->defer ? q : -1);
  l->gb = 0;
  l->rd_slot = -1;
  if (rc != DRA_OK) return rc;
  l->fs_issued++;
  if (l->defer) { l->defer_host = true; l->defer_q = q; }
  return DRA_OK;
}

static int step_lane(dra_dqn_learner* l, const dra_dqn_step_params* prm, hipStream_t su, hipStream_t sa) {
  const int B = l->c.batch;
  const int q = (int)(l->step_no & 3), cur = (q + 3) & 3;
  int rc;
  if (!l->fs_on) {
    // entering: whatever the event paths left on the streams completes first, then the counts agree by construction
    DRA_HIP(hipStreamSynchronize(su));
    DRA_HIP(hipStreamSynchronize(sa));
    __atomic_store_n(&l->fs_host[0], (unsigned long long)(uint32_t)l->aring_issued, __ATOMIC_RELEASE);   // (non-persistent launches do not publish)
    for (int i = 0; i < 4; ++i) l->fs_reader[i] = 0;
    l->fs_arec_n = 0;
    l->fs_on = true;
    l->fs_stat[1]++;
  }
  const dra_dqn_step_params* ablk =
      reinterpret_cast<const dra_dqn_step_params*>(l->aring_stage + (size_t)(l->aring_issued % kAringSlots) * kAprmStride);
  using clk = std::chrono::steady_clock;
  auto ns = [](clk::time_point a, clk::time_point b) { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
  const auto tp0 = clk::now();
  // the copy update t overwrites (and, transitively, index buffer q): its last reader must be done
  if (l->fs_reader[q]) {
    if ((rc = fs_wait_actor(l, l->fs_reader[q]))) return rc;
    l->fs_reader[q] = 0;
  }
  // minibatch reads slots an unfinished actor launch writes: the host waits for the actor stream (a real completion: the ring
  // writes must be visible to the update's first launch)
  for (int i = l->fs_arec_n - 1; i >= 0; --i) {
    const auto& r = l->fs_arec[i];
    if (fs_reached(fs_actor_done(l), r.done_at)) continue;
    if (gather_reads_slots(l, prm->idx, r.slots, r.n)) {
      const auto t0 = std::chrono::steady_clock::now();
      DRA_HIP(hipStreamSynchronize(sa));
      l->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      l->fs_stat[3]++;
      break;
    }
  }
  const bool hazard = gather_reads_slots(l, prm->idx, ablk->slot, prm->n_env);
  const auto tp1 = clk::now();
  memcpy(l->idx_pin[q], prm->idx, (size_t)B * sizeof(int64_t));
  if (l->variant & DRA_VAR_IDX_PREFETCH) {
    const uint64_t tag = ((l->rd_issued + 1ull) & 0xffffffull) << 40;
    volatile int64_t* dst = l->idx_tag_pin[q];
    for (int b = 0; b < B; ++b) dst[b] = (int64_t)((uint64_t)prm->idx[b] | tag);
    DRA_HIP(hipMemcpyAsync(l->idx_tag_dev + (size_t)q * 1024, l->idx_tag_pin[q], (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, l->side));
  }
  l->rd_issued++;
  l->last_gb = q & 1;
  const auto tp2 = clk::now();
  // DRA_VAR_TARGET_AHEAD: is the target side of THIS update in the stash (issued one call ago for exactly these indices)?
  l->ah_mode = 0;
  if (l->ah && l->ah_stream) {
    if (l->ah_valid && l->ah_for_step == l->step_no && memcmp(l->ah_idx, prm->idx, (size_t)B * sizeof(int64_t)) == 0) {
      l->ah_mode = 1;
      l->ah_stat[0]++;
    } else {
      if (l->ah_valid) {                                   // a stash for other indices / another update: its launches finish before
        DRA_HIP(hipStreamSynchronize(l->ah_stream));       // this update's in-line target chain touches the same parity set
        l->ah_stat[3]++;
      }
      l->ah_mode = 2;
      l->ah_stat[1]++;
    }
    l->ah_valid = false;
  }
  rc = (l->variant & DRA_VAR_LANE_EAGER) ? rd_eager(l, su, q) : rd_graph(l, su, q, 0);   // [update t] -- counts itself in fs_count at its start
  l->ah_mode = 0;
  if (rc) return rc;
  const uint64_t ah_need = l->fs_issued;                  // (the count update t's first workgroup raises fs_count to)
  if (!l->fs_on) return DRA_EINVAL;                       // (rd_graph never flushes here: fs_eligible checked the pending segment)
  if (hazard) {                                           // the actor launch below must not start before update t has read the ring
    hipLaunchKernelGGL(fs_bump_kernel, dim3(1), dim3(1), 0, su, l->fs_count);
    DRA_LAUNCH_CHECK();
    l->fs_issued++;
    l->fs_stat[2]++;
  }
  l->fs_need_next = l->fs_issued;
  const auto tp3 = clk::now();
  if ((rc = issue_actor_ring(l, prm->n_env, l->pa[cur], cur, sa, true))) return rc;   // [actor t+1] -- polls fs_count >= need
  const auto tp4 = clk::now();
  l->fs_stat[4] += ns(tp0, tp1); l->fs_stat[5] += ns(tp1, tp2); l->fs_stat[6] += ns(tp2, tp3); l->fs_stat[7] += ns(tp3, tp4);
  l->actor_pending = true;
  const uint64_t done_at = l->aring_issued;               // (issue_actor_ring advanced it: agent steps completed once this launch is)
  l->fs_reader[cur] = done_at;
  if (l->fs_arec_n == 4) { for (int i = 1; i < 4; ++i) l->fs_arec[i - 1] = l->fs_arec[i]; l->fs_arec_n = 3; }
  auto& r = l->fs_arec[l->fs_arec_n++];
  r.done_at = done_at; r.n = prm->n_env;
  for (int e = 0; e < prm->n_env && e < 8; ++e) r.slots[e] = ablk->slot[e];
  l->pa_cur = q;                                          // the graph's optimizer writes copy q
  // DRA_VAR_TARGET_AHEAD: [target(next_states) of update t + 1] on its own stream, gated on update t's start.  Skipped (the next
  // update then computes its target in line) when one of its frames is a slot an unfinished actor launch writes -- the launch just
  // issued included -- or the NEXT actor launch will overwrite (that launch starts with update t + 1, the ahead sequence is only
  // known to be complete at that update's head kernel).
  if (l->ah && l->ah_stream && l->ah_next_set) {
    l->ah_next_set = false;
    bool ok = l->aring_issued < l->aring_pushed;          // (the next launch's block must be known)
    for (int i = 0; ok && i < l->fs_arec_n; ++i) {
      const auto& ar = l->fs_arec[i];
      if (!fs_reached(fs_actor_done(l), ar.done_at) && gather_reads_slots(l, l->ah_next_idx, ar.slots, ar.n)) ok = false;
    }
    if (ok) {
      const dra_dqn_step_params* nblk =
          reinterpret_cast<const dra_dqn_step_params*>(l->aring_stage + (size_t)(l->aring_issued % kAringSlots) * kAprmStride);
      if (gather_reads_slots(l, l->ah_next_idx, nblk->slot, nblk->n_env)) ok = false;
    }
    if (ok) {
      const uint64_t nxt = l->step_no + 1;
      int64_t* pin = l->ah_idx_pin[nxt & 7];
      memcpy(pin, l->ah_next_idx, (size_t)B * sizeof(int64_t));
      // (gated on the update's START.  Gating on its head kernel instead -- the sequence then runs beside the small launches and the
      // backward chain -- and confining the ahead stream to 64 / 112 CUs measured no better: profiles/r06zx_ab_ahead_placement.jsonl)
      hipLaunchKernelGGL(ah_gate_kernel, dim3(1), dim3(1), 0, l->ah_stream, (const unsigned long long*)l->fs_count,
                         (unsigned long long)ah_need, l->timeout_flag);
      DRA_LAUNCH_CHECK();
      if ((rc = ah_target_launches(l, l->ah_stream, (int)(nxt & 1), pin, true, nxt + 1ull))) return rc;
      memcpy(l->ah_idx, l->ah_next_idx, (size_t)B * sizeof(int64_t));
      l->ah_for_step = nxt;
      l->ah_valid = true;
    } else {
      l->ah_stat[2]++;
    }
  }
  l->step_no++;
  l->fs_stat[0]++;
  l->fs_stat[8] += ns(tp0, clk::now());
  return DRA_OK;
}

// DRA_VAR_TARGET_AHEAD: the stream the ahead sequences run on (a stream on the update's CU partition; not owned).  Null switches
// the variant off again.  Leaves the lane.
DRA_API int dra_dqn_learner_set_ahead_stream(dra_dqn_learner* l, void* stream) {
  if (!l) return DRA_EINVAL;
  if (int rcl = fs_leave_any(l)) return rcl;
  if (!stream || !l->ah_done) { l->ah = false; l->ah_stream = nullptr; return DRA_OK; }   // (no workspaces: this learner's configuration
                                                                                              // cannot run the variant -- dra_dqn_learner_ahead_stats says so)
  l->ah_stream = dra_stream(stream);
  l->ah = true;
  return DRA_OK;
}

// DRA_VAR_TARGET_AHEAD: the minibatch indices of the NEXT update (the call after the coming dra_dqn_learner_step), known one call
// early.  The coming call issues that update's target(next_states) under its own update; the next call uses it if it is handed
// exactly these indices, and computes the target in line otherwise.  n = the learner's batch.
DRA_API int dra_dqn_learner_stage_next_indices(dra_dqn_learner* l, const int64_t* idx_next, int n) {
  if (!l || !idx_next || n != l->c.batch || n > 1024) return DRA_EINVAL;
  if (!l->ah) return DRA_OK;          // (nothing to do: every update computes both nets)
  memcpy(l->ah_next_idx, idx_next, (size_t)n * sizeof(int64_t));
  l->ah_next_set = true;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_ahead_stats(dra_dqn_learner* l, int64_t* out) {
  if (!l || !out) return DRA_EINVAL;
  for (int i = 0; i < 4; ++i) out[i] = l->ah_stat[i];
  out[4] = l->ah ? 1 : 0;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_lane_stats(dra_dqn_learner* l, int64_t* out) {
  if (!l || !out) return DRA_EINVAL;
  for (int i = 0; i < 12; ++i) out[i] = l->fs_stat[i];
  return DRA_OK;
}

// pinned staging slot k is free again once the copies issued from it have completed
static int stage_acquire(dra_dqn_learner* l, int* k_out) {
  const int k = l->stage_k;
  l->stage_k = (k + 1) % 8;
  if (l->stage_used[k]) {
    const auto t0 = std::chrono::steady_clock::now();
    DRA_HIP(hipEventSynchronize(l->stage_ev[k]));
    // GATHER_ON_UPDATE: the gather of that call read its indices from slot k on the UPDATE stream; the actor graph of the
    // call after it waited for that update, so its event (7 calls old) implies the gather is done
    if ((l->variant & DRA_VAR_GATHER_ON_UPDATE) && l->stage_used[(k + 1) % 8]) DRA_HIP(hipEventSynchronize(l->stage_ev[(k + 1) % 8]));
    l->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  l->stage_used[k] = true;
  *k_out = k;
  return DRA_OK;
}

// Actor only: runs prm->n_env environment transitions on `stream` (graph replay).
DRA_API int dra_dqn_learner_act(dra_dqn_learner* l, const dra_dqn_step_params* prm, int use_graph, void* stream) {
  if (!l || !prm || prm->n_env < 1 || prm->n_env > kMaxEnvSteps) return DRA_EINVAL;
  if (int rcf = flush_fc4(l, dra_stream(stream))) return rcf;   // (DRA_VAR_DEFER_FC4: nothing outside the riding graphs sees a half-stepped fc4)
  hipStream_t st = dra_stream(stream);
  int k;
  int rc = stage_acquire(l, &k);
  if (rc) return rc;
  memcpy(&l->prm_stage[k], prm, kPrmHeadBytes);
  if (l->variant & DRA_VAR_ACTOR_RING) rc = issue_actor_ring(l, prm->n_env, l->p, 0, st, false);
  else rc = issue_actor(l, prm, k, l->p, st, use_graph != 0);
  DRA_HIP(hipEventRecord(l->stage_ev[k], st));
  l->actor_last = l->stage_ev[k];
  arec_push(l, l->stage_ev[k], nullptr, -1);   // (slots not tracked: an update issued while it runs waits unconditionally)
  return rc;
}

// One whole agent step (DQN_agent.py:101-138 minus logging): prm->n_env actor transitions, then one
// gradient update on prm->idx.
//   sync mode  (stream_actor == NULL): actor then update, in order, on stream_update -- the
//              reference's async_actor=False ordering; this is the parity mode.
//   async mode (stream_actor != NULL): the transitions issued by THIS call are those of the NEXT
//              step; they run on stream_actor under this step's update (which consumes the
//              transitions issued by the previous call).  Call dra_dqn_learner_act once before
//              the first step.  Exclusions, as HIP events: actor ring writes wait for this
//              update's gather; the optimizer kernel waits for the actor's forwards; the next
//              actor waits for the optimizer.
static int learner_step_impl(dra_dqn_learner* l, const dra_dqn_step_params* prm, int do_update, void* stream_update,
                             void* stream_actor);

DRA_API int dra_dqn_learner_step(dra_dqn_learner* l, const dra_dqn_step_params* prm, int do_update, void* stream_update,
                                 void* stream_actor) {
  if (!l || !prm || prm->n_env < 0 || prm->n_env > kMaxEnvSteps) return DRA_EINVAL;
  if (*l->timeout_flag) return DRA_ETIMEDOUT;   // a bounded device-side wait gave up: results are invalid
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = learner_step_impl(l, prm, do_update, stream_update, stream_actor);
  l->host_call_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  l->host_calls++;
  return rc;
}

// dra_dqn_learner_step in two calls (device-drawn prioritized minibatch; include/deeprl_amd.h)
static int step_split_check(dra_dqn_learner* l, const dra_dqn_step_params* prm, void* su, void* sa) {
  if (!l || !prm || !su || !sa || prm->n_env < 1 || prm->n_env > kMaxEnvSteps || !l->per2_dev) return DRA_EINVAL;
  if (!l->step_per || l->step_beta >= 0.f) return DRA_EINVAL;
  if (int rcl = fs_leave_any(l)) return rcl;
  const int need = DRA_VAR_PIPE_GATHER | DRA_VAR_ACTOR_PARAMS | DRA_VAR_GATHER_ON_UPDATE | DRA_VAR_RING_DIRECT | DRA_VAR_ACTOR_RING;
  if ((l->variant & need) != need) return DRA_EINVAL;
  if (*l->timeout_flag) return DRA_ETIMEDOUT;
  return DRA_OK;
}
DRA_API int dra_dqn_learner_step_update(dra_dqn_learner* l, const dra_dqn_step_params* prm, void* stream_update, void* stream_actor) {
  int rc = step_split_check(l, prm, stream_update, stream_actor);
  if (rc) return rc;
  if (l->split_open) return DRA_EINVAL;
  const auto t0 = std::chrono::steady_clock::now();
  l->profiling = false;
  rc = step_pipelined3(l, prm, 1, dra_stream(stream_update), dra_stream(stream_actor), -1, 1);
  if (rc == DRA_OK) {
    l->split_open = true;
    (void)hipStreamQuery(dra_stream(stream_update));     // on its way to the device before the host turns to the chain block
  }
  l->host_call_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
DRA_API int dra_dqn_learner_step_actor(dra_dqn_learner* l, const dra_dqn_step_params* prm, void* stream_update, void* stream_actor) {
  int rc = step_split_check(l, prm, stream_update, stream_actor);
  if (rc) return rc;
  if (!l->split_open) return DRA_EINVAL;
  const auto t0 = std::chrono::steady_clock::now();
  int k;
  if ((rc = stage_acquire(l, &k))) return rc;
  memcpy(&l->prm_stage[k], prm, kPrmHeadBytes);
  memcpy(l->idx_stage + (size_t)k * 1024, prm->idx, (size_t)l->c.batch * sizeof(int64_t));
  rc = step_pipelined3(l, prm, 1, dra_stream(stream_update), dra_stream(stream_actor), k, 2);
  if (rc == DRA_OK) {
    l->split_open = false;
    (void)hipStreamQuery(dra_stream(stream_actor));
  }
  l->host_call_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  l->host_calls++;
  return rc;
}

// PER for the in-order agent step (dra_dqn_learner_step with stream_actor == NULL): per != 0 makes its update apply the
// importance weights of the sampling probabilities currently in the learner's sampling_prob buffer with exponent beta
// and emit the new priorities (DQN_agent.py:120-127); the pipelined step is uniform-replay only.
DRA_API int dra_dqn_learner_set_per(dra_dqn_learner* l, int per, float beta) {
  if (!l) return DRA_EINVAL;
  l->step_per = per != 0;
  l->step_beta = beta;
  return DRA_OK;
}

// PER: the sampling probabilities of the minibatch (f64 on the host, f32 as tensor() would make them) and the importance
// exponent beta go to the learner's sampling_prob buffer ([batch] probabilities, then beta) on `stream`, through the
// learner's own rotating pinned staging -- one call instead of a pinned tensor copy, a device copy and an event in python.
DRA_API int dra_dqn_learner_upload_sampling_prob(dra_dqn_learner* l, const double* prob_host, int n, float beta, void* stream) {
  if (!l || !prob_host || n != l->c.batch) return DRA_EINVAL;
  const int k = l->sp_k;
  l->sp_k = (k + 1) % 8;
  if (l->sp_used[k]) DRA_HIP(hipEventSynchronize(l->sp_ev[k]));
  float* dst = l->sp_stage + (size_t)k * 1025;
  for (int i = 0; i < n; ++i) dst[i] = (float)prob_host[i];
  dst[n] = beta;
  DRA_HIP(hipMemcpyAsync(l->samp_prob, dst, (size_t)(n + 1) * sizeof(float), hipMemcpyHostToDevice, dra_stream(stream)));
  DRA_HIP(hipEventRecord(l->sp_ev[k], dra_stream(stream)));
  l->sp_used[k] = true;
  return DRA_OK;
}

// Minibatch indices (host int64[batch]) -> the learner's idx buffer on `stream`, through the learner's own pinned staging
// (eight rotating slots).  What tensor(idx) + copy_ did from python, without torch's stream / device context managers and
// per-call Event objects on the host's critical path (~30 us per agent step of a host-environment run).
DRA_API int dra_dqn_learner_upload_indices(dra_dqn_learner* l, const int64_t* idx_host, int n, void* stream) {
  if (!l || !idx_host || n != l->c.batch) return DRA_EINVAL;
  const int k = l->ui_k;
  l->ui_k = (k + 1) % 8;
  if (l->ui_used[k]) DRA_HIP(hipEventSynchronize(l->ui_ev[k]));
  int64_t* dst = l->ui_stage + (size_t)k * 1024;
  memcpy(dst, idx_host, (size_t)n * sizeof(int64_t));
  DRA_HIP(hipMemcpyAsync(l->idx, dst, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, dra_stream(stream)));
  DRA_HIP(hipEventRecord(l->ui_ev[k], dra_stream(stream)));
  l->ui_used[k] = true;
  return DRA_OK;
}

// Host-side accounting of dra_dqn_learner_step since the last reset: out[0] = calls, out[1] = seconds inside
// the call, out[2] = seconds of that blocked on a pinned staging slot (i.e. waiting for the GPU: back-pressure).
DRA_API int dra_dqn_learner_host_stats(dra_dqn_learner* l, double* out, int reset) {
  if (!l || !out) return DRA_EINVAL;
  out[0] = (double)l->host_calls; out[1] = l->host_call_s; out[2] = l->host_wait_s;
  if (reset) { l->host_calls = 0; l->host_call_s = 0; l->host_wait_s = 0; }
  return DRA_OK;
}

static int learner_step_impl(dra_dqn_learner* l, const dra_dqn_step_params* prm, int do_update, void* stream_update,
                             void* stream_actor) {
  hipStream_t su = dra_stream(stream_update);
  hipStream_t sa = dra_stream(stream_actor);
  const int B = l->c.batch;
  l->profiling = false;
  int k, rc;
  if (fs_eligible(l, prm, do_update, stream_actor)) {
    if (l->fs_on && (su != l->fs_su || sa != l->fs_sa)) { if ((rc = fs_leave_any(l))) return rc; }
    l->fs_su = su; l->fs_sa = sa;
    return step_lane(l, prm, su, sa);
  }
  if ((rc = fs_leave_any(l))) return rc;
  {   // DRA_VAR_DEFER_FC4: only step_pipelined3 may leave a segment pending across calls (it decides for itself)
    const bool p3 = stream_actor && (l->variant & DRA_VAR_PIPE_GATHER) && (l->variant & DRA_VAR_ACTOR_PARAMS) &&
                    (l->variant & DRA_VAR_GATHER_ON_UPDATE) && !((l->variant & DRA_VAR_GATHER_IN_GRAPH) && !(l->variant & DRA_VAR_ACTOR_V3));
    if (!p3 && (rc = flush_fc4(l, su))) return rc;
  }
  if ((rc = stage_acquire(l, &k))) return rc;
  memcpy(&l->prm_stage[k], prm, kPrmHeadBytes);
  memcpy(l->idx_stage + (size_t)k * 1024, prm->idx, (size_t)B * sizeof(int64_t));
  if (!stream_actor) {  // ---- sync mode
    if (prm->n_env > 0) {
      if ((rc = issue_actor(l, prm, k, l->p, su, true))) return rc;
    }
    if (do_update) l->pa_valid = false;
    if (do_update) {
      const int64_t* pinned = (l->variant & DRA_VAR_PINNED_IDX) ? l->idx_stage + (size_t)k * 1024 : nullptr;
      if (!pinned)
        DRA_HIP(hipMemcpyAsync(l->idx, l->idx_stage + (size_t)k * 1024, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, su));
      if ((rc = launch_gather(l, su, pinned))) return rc;
      // PER: beta >= 0 is a kernel argument that changes every update (eager chain); beta < 0 = read it from
      // sampling_prob[B] on the device (captured graph)
      if ((rc = l->step_per ? (l->step_beta < 0.f ? body_graph_per(l, su) : run_body(l, su, 1, l->step_beta, 0)) : body_graph(l, su)))
        return rc;
      if ((rc = launch_optimizer(l, su))) return rc;
      DRA_HIP(hipEventRecord(l->ev_step_done, su));
      l->last_done = l->ev_step_done;
    }
    DRA_HIP(hipEventRecord(l->stage_ev[k], su));
    return DRA_OK;
  }
  // ---- async mode
  // PER in the pipelined step: the gather-on-update pipeline only (its update graphs have a PER variant; the host performs the
  // prioritized draw of step t after the write-back of update t-1, so the two chains still overlap inside a step)
  if (l->step_per && do_update && (!(l->variant & DRA_VAR_GATHER_ON_UPDATE) || l->step_beta >= 0.f)) return DRA_EINVAL;
  if ((l->variant & DRA_VAR_PIPE_GATHER) && (l->variant & DRA_VAR_ACTOR_PARAMS)) {
    if ((l->variant & DRA_VAR_GATHER_IN_GRAPH) && !(l->variant & DRA_VAR_ACTOR_V3)) return step_pipelined2(l, prm, do_update, su, sa, k);
    if (l->variant & DRA_VAR_GATHER_ON_UPDATE) return step_pipelined3(l, prm, do_update, su, sa, k);
    return step_pipelined(l, prm, do_update, su, sa, k);
  }
  if (do_update) {
    if (l->actor_pending) DRA_HIP(hipStreamWaitEvent(su, l->ev_actor_done, 0));  // transitions of this step are in the ring
    const int64_t* pinned = (l->variant & DRA_VAR_PINNED_IDX) ? l->idx_stage + (size_t)k * 1024 : nullptr;
    if (!pinned)
      DRA_HIP(hipMemcpyAsync(l->idx, l->idx_stage + (size_t)k * 1024, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, su));
    if ((rc = launch_gather(l, su, pinned))) return rc;
    DRA_HIP(hipEventRecord(l->ev_gather_done, su));
  }
  const bool dbuf = l->variant & DRA_VAR_ACTOR_PARAMS;
  bool seeded = false;
  if (prm->n_env > 0) {  // issued before the update body so that it starts as soon as the gather is done
    if (do_update) DRA_HIP(hipStreamWaitEvent(sa, l->ev_gather_done, 0));  // do not overwrite slots the gather reads
    const float* pact = l->p;
    if (dbuf) {
      if (!l->pa_valid) {  // (re)seed the actor copy: first async step, or the parameters changed behind it
        if (l->last_done) DRA_HIP(hipStreamWaitEvent(sa, l->last_done, 0));   // after the last optimiser step ...
        DRA_HIP(hipMemcpyAsync(l->pa[l->pa_cur], l->p, (size_t)l->c.n_params * sizeof(float), hipMemcpyDeviceToDevice, sa));
        DRA_HIP(hipEventRecord(l->ev_join[0], sa));            // ... and before this step's (see below)
        l->pa_valid = true;
        seeded = true;
      }
      pact = l->pa[l->pa_cur];
    }
    if ((rc = issue_actor(l, prm, k, pact, sa, true))) return rc;
    DRA_HIP(hipEventRecord(l->ev_actor_done, sa));
    l->actor_pending = true;
  }
  if (do_update) {
    if ((rc = body_graph(l, su))) return rc;
    if (dbuf && l->pa_valid) {
      // the optimiser writes the parameters twice: in place, and into the actor copy the NEXT actor graph will
      // read; the copy the current actor reads is untouched, so neither side waits for the other
      if (seeded) DRA_HIP(hipStreamWaitEvent(su, l->ev_join[0], 0));  // the seed copy read the old parameters
      if ((rc = launch_optimizer(l, su, l->pa[l->pa_cur ^ 1]))) return rc;
      l->pa_cur ^= 1;
      DRA_HIP(hipEventRecord(l->ev_step_done, su));
      l->last_done = l->ev_step_done;
    } else {
      if (prm->n_env > 0) DRA_HIP(hipStreamWaitEvent(su, l->ev_actor_done, 0));  // config.lock: optimizer excludes actor reads
      if ((rc = launch_optimizer(l, su))) return rc;
      DRA_HIP(hipEventRecord(l->ev_step_done, su));
      l->last_done = l->ev_step_done;
      DRA_HIP(hipStreamWaitEvent(sa, l->ev_step_done, 0));  // the next actor sees whole optimizer steps only
    }
  }
  DRA_HIP(hipEventRecord(l->stage_ev[k], do_update ? su : sa));
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------------
// Observations of N device-resident synthetic Atari environments (the vectorised environments of the on-policy agents,
// A2C_agent.py:26-34 / PPO_agent.py:33-47: `states` of one rollout step).  The environment is a pure function of its frame
// counter, so an observation needs no state on the device: frame j of env e's stack is counter[e] - min(H-1-j, age[e])
// of stream seed[e] (age = earlier observations of the same episode, capped at H-1: after a reset the first frame is
// repeated, envs.py FrameStack.reset); the host shadow (SyntheticEpisodeStream) supplies counters / ages for a whole
// rollout ahead of time together with the rewards and terminals.  grid (N, H) x 256 threads, out u8 [N][H][84*84].
__global__ void __launch_bounds__(256)
synth_stacks_kernel(const int64_t* __restrict__ counter, const int32_t* __restrict__ age, const int64_t* __restrict__ seed,
                    int history, uint8_t* __restrict__ out) {
  const int e = blockIdx.x, j = blockIdx.y;
  const int back = min(history - 1 - j, (int)age[e]);
  const int64_t c = counter[e] - back;
  uint64_t* dst = reinterpret_cast<uint64_t*>(out + ((int64_t)e * history + j) * 7056);
  const uint64_t sd = (uint64_t)seed[e];
  for (int w = threadIdx.x; w < 882; w += 256) dst[w] = synth_frame_word(sd, c, 0, w);
}

DRA_API int dra_synth_stacks(const int64_t* counter_dev, const int32_t* age_dev, const int64_t* seed_dev, int n_env, int history,
                             void* out_u8, void* stream) {
  if (!counter_dev || !age_dev || !seed_dev || !out_u8 || n_env < 1 || history < 1 || history > 16) return DRA_EINVAL;
  if (((uintptr_t)out_u8) & 7) return DRA_EINVAL;
  hipLaunchKernelGGL(synth_stacks_kernel, dim3(n_env, history), dim3(256), 0, dra_stream(stream), counter_dev, age_dev, seed_dev,
                     history, (uint8_t*)out_u8);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

#ifdef DRA_TRACE
// Measurement build only (libdeeprl_amd_trace.so): points every translation unit's phase-trace pointer at `buf`
// (TR_REGIONS x kTraceWgs x 8 u64, device memory; null switches the stamps off).  tools/phase_trace.py.
extern "C" int dra_trace_set_conv_v2(void*);
extern "C" int dra_trace_set_ring(void*);
extern "C" int dra_trace_set_optim(void*);
extern "C" int dra_trace_set_fused(void*);
DRA_API int dra_trace_set(void* buf) {
  int rc = dra_trace_set_local(buf);
  rc |= dra_trace_set_conv_v2(buf); rc |= dra_trace_set_ring(buf); rc |= dra_trace_set_optim(buf); rc |= dra_trace_set_fused(buf);
  return rc;
}
DRA_API int dra_trace_layout(int* regions, int* wgs) { *regions = TR_REGIONS; *wgs = kTraceWgs; return DRA_OK; }
#endif
