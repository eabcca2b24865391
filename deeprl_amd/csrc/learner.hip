// Fused DQN learner: one C-ABI call per gradient update / per environment step.
// Replaces the body of deep_rl/agent/DQN_agent.py:114-138 (sample -> compute_loss -> backward ->
// clip -> optimizer.step -> target sync) and the per-step inference of DQNActor._transition
// (DQN_agent.py:24-45) for VanillaNet(NatureConvBody).
//
// MI355X design: the reference issues ~640 ATen ops per update; here the update is a fixed chain
// of ~20 hand-written kernels over persistent HBM workspaces, captured ONCE into a hipGraph and
// replayed -- launch-bound inner loops belong in graphs (no tracing compiler involved).  Online
// (states) and target (next_states) forwards share launches (blockIdx.z), split-K slabs of the
// conv weight gradients are folded inside the gradient-norm pass, and nothing returns to the host:
// the only per-update host->device traffic is the 256-byte index vector.
#include "common.h"
#include <new>
#include <string.h>

enum { K_GATHER, K_CONV1_F, K_CONV2_F, K_CONV3_F, K_FC4_F, K_HEAD_F, K_LOSS, K_HEAD_B, K_FC4_BW, K_FC4_BX, K_CONV3_BW,
       K_CONV3_BX, K_CONV2_BW, K_CONV2_BX, K_CONV1_BW, K_NORM, K_STEP, K_COUNT };

static const char* kKernelNames[K_COUNT] = {
  "gather", "conv1_fwd", "conv2_fwd", "conv3_fwd", "fc4_fwd", "head_fwd", "td_loss", "head_bwd", "fc4_bwd_w", "fc4_bwd_x",
  "conv3_bwd_w", "conv3_bwd_x", "conv2_bwd_w", "conv2_bwd_x", "conv1_bwd_w", "grad_norm", "rmsprop_step"};

// parameter tensor order inside the flat buffers (conv segment first, 16-byte aligned offsets);
// the three conv weights are stored in the KOC layout [(c,kh,kw)][oc] (conv_v2.hip)
enum { P_W1, P_B1, P_W2, P_B2, P_W3, P_B3, P_W4, P_B4, P_WH, P_BH, P_COUNT };

struct dra_dqn_learner {
  dra_dqn_config c;
  dra_ring* ring;
  float *p, *pt, *g, *s1, *s2;
  // workspaces
  uint8_t *state, *next_state, *act_state;
  int64_t *action, *idx;
  float *reward, *mask;
  float *y1[3], *y2[3], *y3[3], *h4[3], *q[3];
  float *ay1, *ay2, *ay3, *ah4, *aq;  // actor (batch 1)
  float *dq, *dh4, *dy3, *dy2, *dy1, *delta, *prio, *weights, *samp_prob;
  float *slabs, *lin_ws;
  int64_t lin_ws_floats, slab_stride;
  double* partials;
  float *loss, *norm;
  hipGraphExec_t graph;
  bool graph_ready;
  hipEvent_t ev[K_COUNT + 1];
  bool profiling;
};

static int alloc_f(float** p, int64_t n) { return (int)hipMalloc(p, (size_t)n * sizeof(float)); }

DRA_API int dra_dqn_learner_create(dra_dqn_learner** out, dra_ring* ring, const dra_dqn_config* cfg, float* params,
                                   float* target, float* grad, float* state1, float* state2) {
  if (!out || !ring || !cfg || !params || !target || !grad || !state1 || !state2) return DRA_EINVAL;
  if (cfg->batch < 1 || cfg->batch > 1024 || cfg->n_actions < 1 || cfg->ksplit < 1 || cfg->ksplit > 64) return DRA_EINVAL;
  dra_dqn_learner* l = new (std::nothrow) dra_dqn_learner();
  if (!l) return DRA_ENOMEM;
  memset(l, 0, sizeof(*l));
  l->c = *cfg; l->ring = ring; l->p = params; l->pt = target; l->g = grad; l->s1 = state1; l->s2 = state2;
  const int B = cfg->batch, A = cfg->n_actions;
  const int nz = cfg->double_q ? 3 : 2;
  int rc = 0;
  rc |= (int)hipMalloc(&l->state, (size_t)B * 4 * 7056);
  rc |= (int)hipMalloc(&l->next_state, (size_t)B * 4 * 7056);
  rc |= (int)hipMalloc(&l->act_state, (size_t)4 * 7056);
  rc |= (int)hipMalloc(&l->action, (size_t)B * 8);
  rc |= (int)hipMalloc(&l->idx, (size_t)B * 8);
  rc |= alloc_f(&l->reward, B); rc |= alloc_f(&l->mask, B);
  for (int z = 0; z < nz; ++z) {
    rc |= alloc_f(&l->y1[z], (int64_t)B * 32 * 400); rc |= alloc_f(&l->y2[z], (int64_t)B * 64 * 81);
    rc |= alloc_f(&l->y3[z], (int64_t)B * 64 * 49); rc |= alloc_f(&l->h4[z], (int64_t)B * 512);
    rc |= alloc_f(&l->q[z], (int64_t)B * A);
  }
  rc |= alloc_f(&l->ay1, 32 * 400); rc |= alloc_f(&l->ay2, 64 * 81); rc |= alloc_f(&l->ay3, 64 * 49);
  rc |= alloc_f(&l->ah4, 512); rc |= alloc_f(&l->aq, A);
  rc |= alloc_f(&l->dq, (int64_t)B * A); rc |= alloc_f(&l->dh4, (int64_t)B * 512);
  rc |= alloc_f(&l->dy3, (int64_t)B * 64 * 49); rc |= alloc_f(&l->dy2, (int64_t)B * 64 * 81);
  rc |= alloc_f(&l->dy1, (int64_t)B * 32 * 400);
  rc |= alloc_f(&l->delta, B); rc |= alloc_f(&l->prio, B); rc |= alloc_f(&l->weights, B); rc |= alloc_f(&l->samp_prob, B);
  l->slab_stride = cfg->conv_end;  // conv segment occupies [0, conv_end) of the flat layout
  rc |= alloc_f(&l->slabs, (int64_t)cfg->ksplit * l->slab_stride);
  l->lin_ws_floats = (int64_t)3 * 32 * B * 512;
  rc |= alloc_f(&l->lin_ws, l->lin_ws_floats);
  rc |= (int)hipMalloc(&l->partials, (size_t)2 * dra_norm_partials() * sizeof(double));
  rc |= alloc_f(&l->loss, 1); rc |= alloc_f(&l->norm, 1);
  if (rc) { delete l; return rc; }
  // slab gaps (alignment padding between tensors) are never written: keep them zero
  rc |= (int)hipMemset(l->slabs, 0, (size_t)cfg->ksplit * l->slab_stride * sizeof(float));
  rc |= (int)hipMemset(l->partials, 0, (size_t)2 * dra_norm_partials() * sizeof(double));
  for (int k = 0; k <= K_COUNT; ++k) rc |= (int)hipEventCreate(&l->ev[k]);
  if (rc) { delete l; return rc; }
  *out = l;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_destroy(dra_dqn_learner* l) {
  if (!l) return DRA_OK;
  if (l->graph_ready) (void)hipGraphExecDestroy(l->graph);
  void* bufs[] = {l->state, l->next_state, l->act_state, l->action, l->idx, l->reward, l->mask, l->ay1, l->ay2, l->ay3,
                  l->ah4, l->aq, l->dq, l->dh4, l->dy3, l->dy2, l->dy1, l->delta, l->prio, l->weights, l->samp_prob,
                  l->slabs, l->lin_ws, l->partials, l->loss, l->norm};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (int z = 0; z < 3; ++z) {
    if (l->y1[z]) (void)hipFree(l->y1[z]);
    if (l->y2[z]) (void)hipFree(l->y2[z]);
    if (l->y3[z]) (void)hipFree(l->y3[z]);
    if (l->h4[z]) (void)hipFree(l->h4[z]);
    if (l->q[z]) (void)hipFree(l->q[z]);
  }
  for (int k = 0; k <= K_COUNT; ++k) (void)hipEventDestroy(l->ev[k]);
  delete l;
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

#define STEP(kid, expr)                                                        \
  do {                                                                         \
    if (l->profiling) DRA_HIP(hipEventRecord(l->ev[kid], st));                 \
    int _rc = (expr);                                                          \
    if (_rc != DRA_OK) return _rc;                                             \
  } while (0)

static int run_update(dra_dqn_learner* l, hipStream_t st, int per, float beta) {
  const dra_dqn_config& c = l->c;
  const int B = c.batch, A = c.n_actions;
  const int nz = c.double_q ? 3 : 2;
  void* s = (void*)st;
  const float* P = l->p;
  const float* T = l->pt;
  const int64_t* o = c.offset;
  STEP(K_GATHER, dra_ring_gather(l->ring, l->idx, B, l->state, l->next_state, l->action, nullptr, nullptr, l->reward,
                                 l->mask, s));
  // z = 0: online(states)   z = 1: target(next_states)   z = 2: online(next_states) [double-Q]
  const void* x1[3] = {l->state, l->next_state, l->next_state};
  const float* w1[3] = {P + o[P_W1], T + o[P_W1], P + o[P_W1]};
  const float* b1[3] = {P + o[P_B1], T + o[P_B1], P + o[P_B1]};
  STEP(K_CONV1_F, dra_conv_fwd_koc(1, nz, x1, w1, b1, l->y1, B, 1, c.u8_coef, DRA_ACT_RELU, s));
  const void* x2[3] = {l->y1[0], l->y1[1], l->y1[2]};
  const float* w2[3] = {P + o[P_W2], T + o[P_W2], P + o[P_W2]};
  const float* b2[3] = {P + o[P_B2], T + o[P_B2], P + o[P_B2]};
  STEP(K_CONV2_F, dra_conv_fwd_koc(2, nz, x2, w2, b2, l->y2, B, 0, 1.0, DRA_ACT_RELU, s));
  const void* x3[3] = {l->y2[0], l->y2[1], l->y2[2]};
  const float* w3[3] = {P + o[P_W3], T + o[P_W3], P + o[P_W3]};
  const float* b3[3] = {P + o[P_B3], T + o[P_B3], P + o[P_B3]};
  STEP(K_CONV3_F, dra_conv_fwd_koc(3, nz, x3, w3, b3, l->y3, B, 0, 1.0, DRA_ACT_RELU, s));
  const float* x4[3] = {l->y3[0], l->y3[1], l->y3[2]};
  const float* w4[3] = {P + o[P_W4], T + o[P_W4], P + o[P_W4]};
  const float* b4[3] = {P + o[P_B4], T + o[P_B4], P + o[P_B4]};
  STEP(K_FC4_F, dra_linear_fwd(nz, x4, w4, b4, l->h4, B, 3136, 512, DRA_ACT_RELU, l->lin_ws, l->lin_ws_floats, s));
  const float* xh[3] = {l->h4[0], l->h4[1], l->h4[2]};
  const float* wh[3] = {P + o[P_WH], T + o[P_WH], P + o[P_WH]};
  const float* bh[3] = {P + o[P_BH], T + o[P_BH], P + o[P_BH]};
  STEP(K_HEAD_F, dra_linear_fwd(nz, xh, wh, bh, l->q, B, 512, A, DRA_ACT_NONE, l->lin_ws, l->lin_ws_floats, s));
  STEP(K_LOSS, dra_td_loss(l->q[0], l->q[1], c.double_q ? l->q[2] : nullptr, l->action, 1, l->reward, l->mask, B, A,
                           c.gamma_n, per ? l->samp_prob : nullptr, beta, c.replay_eps, c.replay_alpha, l->loss, l->dq,
                           l->delta, per ? l->prio : nullptr, per ? l->weights : nullptr, s));
  float* G = l->g;
  STEP(K_HEAD_B, dra_linear_bwd_w(l->dq, l->h4[0], G + o[P_WH], G + o[P_BH], B, 512, A, s));
  STEP(K_HEAD_B, dra_linear_bwd_x(l->dq, P + o[P_WH], l->h4[0], l->dh4, B, 512, A, DRA_ACT_RELU, s));
  STEP(K_FC4_BW, dra_linear_bwd_w(l->dh4, l->y3[0], G + o[P_W4], G + o[P_B4], B, 3136, 512, s));
  STEP(K_FC4_BX, dra_linear_bwd_x(l->dh4, P + o[P_W4], l->y3[0], l->dy3, B, 3136, 512, DRA_ACT_RELU, s));
  float* S = l->slabs;
  STEP(K_CONV3_BW, dra_conv_bwd_w_koc(3, l->dy3, l->y2[0], S + o[P_W3], S + o[P_B3], l->slab_stride, c.ksplit, B, 0, 1.0, s));
  STEP(K_CONV3_BX, dra_conv_bwd_x_koc(3, l->dy3, P + o[P_W3], l->y2[0], l->dy2, B, DRA_ACT_RELU, s));
  STEP(K_CONV2_BW, dra_conv_bwd_w_koc(2, l->dy2, l->y1[0], S + o[P_W2], S + o[P_B2], l->slab_stride, c.ksplit, B, 0, 1.0, s));
  STEP(K_CONV2_BX, dra_conv_bwd_x_koc(2, l->dy2, P + o[P_W2], l->y1[0], l->dy1, B, DRA_ACT_RELU, s));
  STEP(K_CONV1_BW, dra_conv_bwd_w_koc(1, l->dy1, l->state, S + o[P_W1], S + o[P_B1], l->slab_stride, c.ksplit, B, 1, c.u8_coef, s));
  const int np = dra_norm_partials();
  STEP(K_NORM, dra_grad_sqnorm(G, c.conv_end, S, c.ksplit, l->slab_stride, l->partials, s));  // folds the conv slabs
  STEP(K_NORM, dra_grad_sqnorm(G + c.conv_end, c.n_params - c.conv_end, nullptr, 0, 0, l->partials + np, s));
  STEP(K_STEP, dra_rmsprop_step(l->p, G, l->s1, l->s2, c.n_params, l->partials, 2 * np, c.gradient_clip, c.lr, c.alpha,
                                c.eps, c.centered, l->norm, s));
  if (l->profiling) DRA_HIP(hipEventRecord(l->ev[K_COUNT], st));
  return DRA_OK;
}

// One gradient update on the indices currently in the learner's idx buffer.  use_graph != 0
// captures the chain on first use and replays it afterwards (uniform replay only: the PER beta is
// a kernel argument that changes per update).
DRA_API int dra_dqn_learner_update(dra_dqn_learner* l, int use_graph, int per, float beta, void* stream) {
  if (!l) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  l->profiling = false;
  if (!use_graph || per) return run_update(l, st, per, beta);
  if (!l->graph_ready) {
    hipGraph_t graph;
    DRA_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = run_update(l, st, 0, 0.f);
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc != DRA_OK) return rc;
    if (e != hipSuccess) return (int)e;
    DRA_HIP(hipGraphInstantiate(&l->graph, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    l->graph_ready = true;
  }
  DRA_HIP(hipGraphLaunch(l->graph, st));
  return DRA_OK;
}

// Eager update with a HIP event before every kernel group (on the launch stream); returns the
// per-group milliseconds of THIS update.  Synchronises: measurement aid, not the hot path.
DRA_API int dra_dqn_learner_profile(dra_dqn_learner* l, float* out_ms, int n_out, void* stream) {
  if (!l || !out_ms || n_out < K_COUNT) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  l->profiling = true;
  int rc = run_update(l, st, 0, 0.f);
  l->profiling = false;
  if (rc != DRA_OK) return rc;
  DRA_HIP(hipEventSynchronize(l->ev[K_COUNT]));
  // groups with two launches record their event twice; the later record wins, so a group's time
  // is measured from its last record to the next group's record -- use single-launch groups for
  // roofline figures (conv / fc / gather all are).
  for (int k = 0; k < K_COUNT; ++k) {
    float ms = 0.f;
    DRA_HIP(hipEventElapsedTime(&ms, l->ev[k], l->ev[k + 1]));
    out_ms[k] = ms;
  }
  return DRA_OK;
}

DRA_API int dra_dqn_learner_kernel_name(int k, char* out, int n) {
  if (k < 0 || k >= K_COUNT || !out || n < 1) return DRA_EINVAL;
  strncpy(out, kKernelNames[k], (size_t)n - 1);
  out[n - 1] = 0;
  return DRA_OK;
}

DRA_API int dra_dqn_learner_sync_target(dra_dqn_learner* l, void* stream) {
  if (!l) return DRA_EINVAL;
  return dra_copy_f32(l->pt, l->p, l->c.n_params, stream);
}

// ------------------------------------------------------------------------------------------------
// Actor step (DQN_agent.py:24-45 + torch_utils.py:51-58) entirely on device: stack the last H
// frames ending at `newest_slot` (modular, so the stack may wrap the ring end), batch-1 forward
// of the ONLINE net, epsilon-greedy with HOST-drawn randomness (so the np.random stream is the
// reference's: randint(A) then rand()), action written into the ring's action record of
// `store_slot` and to out_action (device int64, may be NULL).  No device->host sync.
__global__ void __launch_bounds__(256)
actor_stack_kernel(const uint8_t* __restrict__ frames, int64_t capacity, int64_t frame_bytes, int64_t newest, int H,
                   uint8_t* __restrict__ out) {
  const int j = blockIdx.x;  // 0 = oldest
  int64_t slot = newest - (H - 1) + j;
  if (slot < 0) slot += capacity;
  const uint4* s = reinterpret_cast<const uint4*>(frames + slot * frame_bytes);
  uint4* d = reinterpret_cast<uint4*>(out + (int64_t)j * frame_bytes);
  for (int64_t t = threadIdx.x; t < (frame_bytes >> 4); t += blockDim.x) d[t] = s[t];
}

__global__ void eps_greedy_kernel(const float* __restrict__ q, int A, float epsilon, int random_action, float dice,
                                  uint8_t* __restrict__ ring_actions, int64_t store_slot, int64_t action_bytes,
                                  int64_t* __restrict__ out_action) {
  if (threadIdx.x != 0) return;
  int best = 0;
  float bv = q[0];
  for (int k = 1; k < A; ++k) if (q[k] > bv) { bv = q[k]; best = k; }  // np.argmax: first max
  const int64_t a = (dice < epsilon) ? random_action : best;
  if (ring_actions) *reinterpret_cast<int64_t*>(ring_actions + store_slot * action_bytes) = a;
  if (out_action) *out_action = a;
}

DRA_API int dra_dqn_learner_act(dra_dqn_learner* l, int64_t newest_slot, float epsilon, int random_action, float dice,
                                int64_t store_slot, int64_t* out_action_dev, void* stream) {
  if (!l) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  void *frames, *actions;
  int rc = dra_ring_pointers(l->ring, &frames, &actions, nullptr, nullptr);
  if (rc) return rc;
  const dra_dqn_config& c = l->c;
  if (newest_slot < 0 || newest_slot >= c.ring_capacity || store_slot >= c.ring_capacity) return DRA_EINVAL;
  hipLaunchKernelGGL(actor_stack_kernel, dim3(4), dim3(256), 0, st, (const uint8_t*)frames, c.ring_capacity,
                     (int64_t)7056, newest_slot, 4, l->act_state);
  DRA_LAUNCH_CHECK();
  const float* P = l->p;
  const int64_t* o = c.offset;
  void* s = (void*)st;
  const void* x1[1] = {l->act_state}; const float* w1[1] = {P + o[P_W1]}; const float* b1[1] = {P + o[P_B1]};
  float* y1[1] = {l->ay1};
  if ((rc = dra_conv_fwd_koc(1, 1, x1, w1, b1, y1, 1, 1, c.u8_coef, DRA_ACT_RELU, s))) return rc;
  const void* x2[1] = {l->ay1}; const float* w2[1] = {P + o[P_W2]}; const float* b2[1] = {P + o[P_B2]};
  float* y2[1] = {l->ay2};
  if ((rc = dra_conv_fwd_koc(2, 1, x2, w2, b2, y2, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
  const void* x3[1] = {l->ay2}; const float* w3[1] = {P + o[P_W3]}; const float* b3[1] = {P + o[P_B3]};
  float* y3[1] = {l->ay3};
  if ((rc = dra_conv_fwd_koc(3, 1, x3, w3, b3, y3, 1, 0, 1.0, DRA_ACT_RELU, s))) return rc;
  const float* x4[1] = {l->ay3}; const float* w4[1] = {P + o[P_W4]}; const float* b4[1] = {P + o[P_B4]};
  float* h4[1] = {l->ah4};
  if ((rc = dra_linear_fwd(1, x4, w4, b4, h4, 1, 3136, 512, DRA_ACT_RELU, l->lin_ws, l->lin_ws_floats, s))) return rc;
  const float* xh[1] = {l->ah4}; const float* wh[1] = {P + o[P_WH]}; const float* bh[1] = {P + o[P_BH]};
  float* q[1] = {l->aq};
  if ((rc = dra_linear_fwd(1, xh, wh, bh, q, 1, 512, c.n_actions, DRA_ACT_NONE, l->lin_ws, l->lin_ws_floats, s))) return rc;
  hipLaunchKernelGGL(eps_greedy_kernel, dim3(1), dim3(64), 0, st, (const float*)l->aq, c.n_actions, epsilon, random_action,
                     dice, store_slot >= 0 ? (uint8_t*)actions : nullptr, store_slot, (int64_t)8, out_action_dev);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
