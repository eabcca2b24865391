// K2/K3 exports: NatureConvBody / FCBody / head contractions (templates in igemm.h).
#include "igemm.h"
#include "rollout_roles.h"
#include <stdlib.h>

template <class G, int BM, int BN, int BK, bool U8>
static int conv_fwd_t(int nz, const void* const* x, const float* const* w, const float* const* bias, float* const* y,
                      int batch, double coef, int act, hipStream_t st) {
  ConvFwd<G, BM, BN, BK, U8> p;
  p.M = G::OC; p.N = batch * G::P; p.K = G::K;
  for (int z = 0; z < nz; ++z) { p.q.x[z] = x[z]; p.q.w[z] = w[z]; p.q.bias[z] = bias[z]; p.q.y[z] = y[z]; }
  p.act = act; p.coef = coef;
  return launch_igemm(p, nz, 1, st);
}

// layer: 1, 2, 3 = the NatureConvBody convolutions.  x/w/bias/y are arrays of nz (<= 4) device
// pointers: nz independent (input, weights) pairs -- e.g. online net on `states` and target net
// on `next_states` -- run in ONE launch (blockIdx.z), doubling the workgroups in flight.
DRA_API int dra_conv_fwd(int layer, int nz, const void* const* x, const float* const* w, const float* const* bias,
                         float* const* y, int batch, int x_is_u8, double u8_coef, int act, void* stream) {
  if (nz < 1 || nz > kMaxZ || batch < 1 || !x || !w || !bias || !y) return DRA_EINVAL;
  for (int z = 0; z < nz; ++z) if (!x[z] || !w[z] || !bias[z] || !y[z]) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  switch (layer) {
    case 1:
      return x_is_u8 ? conv_fwd_t<G1, 32, 64, 32, true>(nz, x, w, bias, y, batch, u8_coef, act, st)
                     : conv_fwd_t<G1, 32, 64, 32, false>(nz, x, w, bias, y, batch, 1.0, act, st);
    case 2: return x_is_u8 ? DRA_EINVAL : conv_fwd_t<G2, 32, 32, 64, false>(nz, x, w, bias, y, batch, 1.0, act, st);
    case 3: return x_is_u8 ? DRA_EINVAL : conv_fwd_t<G3, 32, 32, 64, false>(nz, x, w, bias, y, batch, 1.0, act, st);
  }
  return DRA_EINVAL;
}

template <class G, int BM, int BN, int BK, bool U8>
static int conv_wgrad_t(const float* dy, const void* x, float* dw, float* db, int64_t slab_stride, int ksplit, int batch,
                        double coef, hipStream_t st) {
  ConvWgrad<G, BM, BN, BK, U8> p;
  p.M = G::OC; p.N = G::K + 1; p.K = batch * G::P;
  p.dy = dy; p.x = x; p.dw = dw; p.db = db; p.slab_stride = slab_stride; p.coef = coef;
  return launch_igemm(p, 1, ksplit, st);
}

// dw/db point at slab 0; slab s of each lives `slab_stride` floats further (s < ksplit).  Every
// slab element is written (zero where a split is empty), so slabs need no pre-clearing.
DRA_API int dra_conv_bwd_w(int layer, const float* dy, const void* x, float* dw, float* db, int64_t slab_stride,
                           int ksplit, int batch, int x_is_u8, double u8_coef, void* stream) {
  if (!dy || !x || !dw || !db || batch < 1 || ksplit < 1) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  switch (layer) {
    case 1:
      return x_is_u8 ? conv_wgrad_t<G1, 32, 32, 64, true>(dy, x, dw, db, slab_stride, ksplit, batch, u8_coef, st)
                     : conv_wgrad_t<G1, 32, 32, 64, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0, st);
    case 2: return x_is_u8 ? DRA_EINVAL : conv_wgrad_t<G2, 32, 32, 64, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0, st);
    case 3: return x_is_u8 ? DRA_EINVAL : conv_wgrad_t<G3, 32, 32, 64, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0, st);
  }
  return DRA_EINVAL;
}

template <class G, int BM, int BN, int BK>
static int conv_dgrad_t(const float* dy, const float* w, const float* xact, float* dx, int batch, int act, hipStream_t st) {
  using PT = ConvDgrad<G, BM, BN, BK>;
  PT p;
  p.M = G::C; p.N = batch * PT::PP; p.K = G::OC * PT::KPP;
  p.dy = dy; p.w = w; p.xact = xact; p.dx = dx; p.act = act;
  return launch_igemm(p, G::S * G::S, 1, st);
}

// dx = gradient w.r.t. the PRE-activation of the layer below (xact = that layer's output, `act`
// its activation).  layer 2 or 3 (conv1 needs no input gradient).
DRA_API int dra_conv_bwd_x(int layer, const float* dy, const float* w, const float* xact, float* dx, int batch, int act,
                           void* stream) {
  if (!dy || !w || !dx || batch < 1) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  switch (layer) {
    case 2: return conv_dgrad_t<G2, 32, 32, 64>(dy, w, xact, dx, batch, act, st);
    case 3: return conv_dgrad_t<G3, 32, 32, 64>(dy, w, xact, dx, batch, act, st);
  }
  return DRA_EINVAL;
}


template <class G, int BM, int BN, int BK, bool U8>
static int conv_wgrad_koc_t(const float* dy, const void* x, float* dw, float* db, int64_t slab_stride, int ksplit,
                            int batch, double coef, hipStream_t st) {
  ConvWgradKoc<G, BM, BN, BK, U8> p;
  p.M = G::K + 1; p.N = G::OC; p.K = batch * G::P;
  p.dy = dy; p.x = x; p.dw = dw; p.db = db; p.slab_stride = slab_stride; p.coef = coef;
  return launch_igemm(p, 1, ksplit, st);
}

DRA_API int dra_conv_bwd_w_koc(int layer, const float* dy, const void* x, float* dw, float* db, int64_t slab_stride,
                               int ksplit, int batch, int x_is_u8, double u8_coef, void* stream) {
  if (!dy || !x || !dw || !db || batch < 1 || ksplit < 1) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  switch (layer) {
    case 1:
      return x_is_u8 ? conv_wgrad_koc_t<G1, 32, 32, 64, true>(dy, x, dw, db, slab_stride, ksplit, batch, u8_coef, st)
                     : conv_wgrad_koc_t<G1, 32, 32, 64, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0, st);
    case 2: return x_is_u8 ? DRA_EINVAL : conv_wgrad_koc_t<G2, 32, 32, 64, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0, st);
    case 3: return x_is_u8 ? DRA_EINVAL : conv_wgrad_koc_t<G3, 32, 32, 64, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0, st);
  }
  return DRA_EINVAL;
}

template <class G, int BM, int BN, int BK>
static int conv_dgrad_koc_t(const float* dy, const float* wt, const float* xact, float* dx, int batch, int act,
                            hipStream_t st) {
  using PT = ConvDgradKoc<G, BM, BN, BK>;
  PT p;
  p.M = G::C; p.N = batch * PT::PP; p.K = G::OC * PT::KPP;
  p.dy = dy; p.wt = wt; p.xact = xact; p.dx = dx; p.act = act;
  return launch_igemm(p, G::S * G::S, 1, st);
}

DRA_API int dra_conv_bwd_x_koc(int layer, const float* dy, const float* wt, const float* xact, float* dx, int batch,
                               int act, void* stream) {
  if (!dy || !wt || !dx || batch < 1) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  switch (layer) {
    case 2: return conv_dgrad_koc_t<G2, 32, 32, 64>(dy, wt, xact, dx, batch, act, st);
    case 3: return conv_dgrad_koc_t<G3, 32, 32, 64>(dy, wt, xact, dx, batch, act, st);
  }
  return DRA_EINVAL;
}

// =============================================================================================
// Linear layers  y[b][o] = act(bias[o] + sum_i x[b][i] * W[o][i])   (runtime sizes)
__global__ void __launch_bounds__(256)
linear_finish_kernel(LinPtrs q, const float* __restrict__ slabs, int ksplit, int B, int O, int act) {
  const int z = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * O) return;
  const float* s = slabs + (int64_t)z * ksplit * B * O + i;
  float v = s[0];
  for (int k = 1; k < ksplit; ++k) v += s[(int64_t)k * B * O];
  const int o = i % O;
  q.y[z][i] = act_apply(v + (q.bias[z] ? q.bias[z][o] : 0.f), act);
}

// Small layers (in_features <= 512: the actor-critic / Q heads on 512 features, every layer of the FCBody nets):
// the K-chunked implicit GEMM pays ~2 us per dependent 64-wide chunk and lands at 8-16 us for a few MFLOP.  Here a
// workgroup stages up to 32 input rows in LDS once (<= 64 KB) and each of its 4 waves owns one output: the weight row
// sits in registers (<= 8 floats per lane, coalesced 2 KB reads), one wave-level dot product per input row.
// grid (ceil(O / 4), nz, ceil(B / 32)).  Summation order per output: lane-strided partial sums, then the wave butterfly.
template <int KV>   // float per lane: ceil(K / 64)
__global__ void __launch_bounds__(256)
linear_gemv_kernel(LinPtrs q, int B, int K, int O, int act) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];   // [rows][K]
  const int z = blockIdx.y, b0 = blockIdx.z * 32, nb = min(32, B - b0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + wave;
  const float* __restrict__ wrow = q.w[z] + (int64_t)min(o, O - 1) * K;
  float w[KV];
#pragma unroll
  for (int i = 0; i < KV; ++i) w[i] = (lane + 64 * i < K) ? wrow[lane + 64 * i] : 0.f;
  const float bias = q.bias[z] ? q.bias[z][min(o, O - 1)] : 0.f;
  const float* __restrict__ src = q.x[z] + (int64_t)b0 * K;
  if ((K & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    // every load of the staging requested before the first LDS write: the plain load -> store loop below is one memory round
    // trip per trip (32 of them for 16 rows of 512: the A2C / PPO heads took 12 us per launch, profiles/r03k_kernel_stats_*)
    constexpr int NV = KV * 64 * 32 / 4 / 256;               // float4 per thread at 32 rows of KV * 64
    const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
    float4* s_x4 = reinterpret_cast<float4*>(s_x);
    const int n4 = nb * K / 4;
    float4 r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = src4[min((int)threadIdx.x + 256 * i, n4 - 1)];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((int)threadIdx.x + 256 * i < n4) s_x4[threadIdx.x + 256 * i] = r[i];
  } else {
    for (int i = threadIdx.x; i < nb * K; i += 256) s_x[i] = src[i];
  }
  __syncthreads();
  if (o >= O) return;
  float* __restrict__ out = q.y[z];
  // eight input rows per round: their wave butterflies are independent chains of six cross-lane moves each (~100 cycles apiece),
  // interleaved by offset -- one row at a time the kernel spent ~0.45 us per input row waiting for them (rocprofv3: 7.2 us at 8
  // rows, 15.5 us at 16, profiles/r04o_kernel_stats_ppo_pixel_8.txt, r04z_kernel_stats_a2c_pixel_16.txt).  Same sums, same order.
  for (int bb = 0; bb < nb; bb += 8) {
    float part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = min(bb + u, nb - 1);
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < KV; ++i) p += (lane + 64 * i < K) ? s_x[b * K + lane + 64 * i] * w[i] : 0.f;
      part[u] = p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] += __shfl_xor(part[u], off, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (bb + u < nb) out[(int64_t)(b0 + bb + u) * O + o] = act_apply(part[u] + bias, act);
    }
  }
}

// Wide layers at rollout batch sizes (fc4 of NatureConvBody, 3136 -> 512, for the 8 / 16 environments of one A2C / PPO rollout
// step): the K-chunked GEMM + its split-K finish were 13.2 + 5.4 us per rollout step (profiles/r02zw_kernel_stats_ppo_pixel_8.txt)
// for 6.4 MB of weights and 26 MFLOP.  Round 2's form -- one wave per output row, eight input rows staged in 100 KB of LDS and
// shared by a workgroup's four rows -- kept one workgroup per CU on 128 CUs, every workgroup re-reading all the input rows:
// 11.6 us = 0.55 TB/s (profiles/r04o_kernel_stats_ppo_pixel_8.txt); removed in round 4 for the form below (ppo_pixel +1.8 %,
// a2c_pixel +0.3 %, profiles/r04u_ab_gemv_rows.txt).
// The input rows are read straight from L2 (they are 100-200 KB in all).  A workgroup of EIGHT waves owns two output rows, each row's reduction split over four waves (K quarters); a lane keeps its
// R float4 of the weight row in registers and, per round, the matching float4 of up to eight input rows -- all requested before
// the first is used; per sample: products, a wave sum, the four quarter sums met in LDS as (q0 + q1) + (q2 + q3).  No barrier
// before the last step, 256 workgroups for 512 outputs.  Per-sample arithmetic does not depend on the batch.
// grid (ceil(O / 2), nz, ceil(B / 32)).  K % 4 == 0, K <= 4096 * ... R * 64 * 4 * 4.
template <int R, int RB, int WPR>
__global__ void __launch_bounds__(512)
linear_gemv_rows_kernel(LinPtrs q, int B, int K, int O, int act) {     // body: rollout_roles.h gemv_rows_body
  __shared__ float s_part[32][8];
  const int z = blockIdx.y;
  gemv_rows_body<R, RB, WPR>(q.x[z], q.w[z], q.bias[z], q.y[z], blockIdx.x, blockIdx.z * 32, B, K, O, act, s_part);
}

template <int R>
static int launch_gemv_rows(const LinPtrs& q, int nz, int batch, int in_features, int out_features, int act, hipStream_t st) {
  hipLaunchKernelGGL((linear_gemv_rows_kernel<R, 8, 4>), dim3((out_features + 1) / 2, nz, (batch + 31) / 32), dim3(512), 0, st, q, batch,
                     in_features, out_features, act);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

extern "C" int dra_linear_fwd_slabs_one(int nz, const float* const* x, const float* const* w, int batch, int in_features,
                                        int out_features, int ksplit, float* slabs, void* stream);   // fused.hip

extern "C" int dra_linear_bwd_x_one512(const float* dy, const float* w, const float* xact, float* dx, int batch, int in_features,
                                       int act, void* stream);   // fused.hip

static constexpr int gemv_enabled() { return 1; }   // (the DRA_LINEAR_GEMV=0 A/B switch is retired: small-batch linears are GEMVs)

// Two linear heads of different width on the SAME input (CategoricalActorCriticNet's fc_action / fc_critic on phi,
// network_heads.py:241-243) in one launch: y0 = x W0^T + b0 [B, O0], y1 = x W1^T + b1 [B, O1].  in_features <= 512.
// A rollout step's heads are a few thousand multiply-adds: everything is latency.  One WAVE per input row: its 512 features sit in
// 8 registers per lane (k = lane + 64 i, the small-layer kernel's own assignment), the weight rows of up to eight outputs are
// requested together with them -- one memory round trip, no LDS, no barrier -- then the eight dot products and their wave
// butterflies run interleaved.  Per-output arithmetic (lane-strided partial sums in i order, then the butterfly) is exactly
// linear_gemv_kernel's: the update's forward through the two Linear modules reproduces these values bit for bit.
// (The staging form took 12.7 us for 16 rows x 5 outputs, 15 % of an A2C agent step: profiles/r04ab_kernel_stats_a2c_pixel_16.txt.)
__global__ void __launch_bounds__(256)
linear_heads_rows_kernel(const float* __restrict__ x, const float* __restrict__ w0, const float* __restrict__ b0,
                         float* __restrict__ y0, int O0, const float* __restrict__ w1, const float* __restrict__ b1,
                         float* __restrict__ y1, int O1, int B, int K, int act) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  heads_row_outputs(x, w0, b0, O0, w1, b1, O1, b, K, act, lane, [&](int o, float v) {
    if (o < O0) y0[(int64_t)b * O0 + o] = v;
    else y1[(int64_t)b * O1 + (o - O0)] = v;
  });
}

// A rollout step's whole policy head (network_heads.py:240-255 under no_grad, action = None) in ONE launch: the two heads above,
// then Categorical(logits) of the row -- inverse-CDF sample from uniform[b], log_pi_a, entropy (common.h categorical_row: the
// statement categorical_fwd_kernel runs) -- on the lane that holds the row's outputs.  Three launches of ~4.5 us each (heads,
// torch.rand, categorical_fwd) in a rollout step of eight before (profiles/r05q_kernel_stats_a2c_pixel_16.txt); the uniforms now
// come from one draw per rollout (nets.RolloutSlots).  Bit-identical with the separate launches for the same uniforms.
template <int KS>
__global__ void __launch_bounds__(256)
policy_heads_sample_kernel(const PolicyHeadArgs h) {       // body: rollout_roles.h policy_head_row
  __shared__ float s_out[4][68];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= h.B) return;
  policy_head_row<KS>(h, b, lane, s_out[wave]);
}

// one workgroup per row, the row's features folded from KS K-slice partial sums first (rollout_roles.h policy_head_row_fold_wg)
template <int KS>
__global__ void __launch_bounds__(256)
policy_heads_foldwg_kernel(const PolicyHeadArgs h) {
  __shared__ float s_phi[512];
  __shared__ float s_out[68];
  policy_head_row_fold_wg<KS>(h, blockIdx.x, s_phi, s_out);
}

static PolicyHeadArgs head_args(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                                const float* uniform, const int64_t* action_in, int batch, int in_features, int n_actions,
                                int64_t* out_action, float* out_lp, float* out_ent, float* out_v, float* out_logits) {
  PolicyHeadArgs h;
  h.x = x; h.w0 = w0; h.b0 = b0; h.w1 = w1; h.b1 = b1; h.uniform = uniform; h.action_in = action_in; h.out_action = out_action;
  h.out_lp = out_lp; h.out_ent = out_ent; h.out_v = out_v; h.out_logits = out_logits; h.B = batch; h.K = in_features; h.A = n_actions;
  h.slabs = nullptr; h.fold_bias = nullptr; h.out_x = nullptr;
  return h;
}

DRA_API int dra_policy_heads_sample(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                                    const float* uniform, int batch, int in_features, int n_actions, int64_t* out_action,
                                    float* out_log_pi_a, float* out_entropy, float* out_v, float* out_logits, void* stream) {
  if (!x || !w0 || !w1 || !uniform || !out_action || !out_log_pi_a || !out_entropy || !out_v || batch < 1 || batch > 65536 ||
      in_features < 1 || in_features > 512 || n_actions < 1 || n_actions > 64)
    return DRA_EINVAL;
  hipLaunchKernelGGL(policy_heads_sample_kernel<0>, dim3((batch + 3) / 4), dim3(256), 0, dra_stream(stream),
                     head_args(x, w0, b0, w1, b1, uniform, nullptr, batch, in_features, n_actions, out_action, out_log_pi_a,
                               out_entropy, out_v, out_logits));
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// The same launch for GIVEN actions (the update's forward, network_heads.py:249-254 with action != None): log_pi_a / entropy of
// the stored actions, v, and the logits the backward needs.
DRA_API int dra_policy_heads_given(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                                   const int64_t* action, int batch, int in_features, int n_actions, float* out_log_pi_a,
                                   float* out_entropy, float* out_v, float* out_logits, void* stream) {
  if (!x || !w0 || !w1 || !action || !out_log_pi_a || !out_entropy || !out_v || !out_logits || batch < 1 || batch > 65536 ||
      in_features < 1 || in_features > 512 || n_actions < 1 || n_actions > 64)
    return DRA_EINVAL;
  hipLaunchKernelGGL(policy_heads_sample_kernel<0>, dim3((batch + 3) / 4), dim3(256), 0, dra_stream(stream),
                     head_args(x, w0, b0, w1, b1, nullptr, action, batch, in_features, n_actions, nullptr, out_log_pi_a, out_entropy,
                               out_v, out_logits));
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ... and with fc4's finish in front (the update's forward of CategoricalActorCriticNet(NatureConvBody)): the features are folded
// from the 8 / 14 K-slice partial sums of dra_linear_fwd_slabs_one inside the head launch (+ bias, ReLU; written to out_phi for
// the backward pass; 8 slices: 8.9 / 17.3 us at 80 / 256 rows against 10.9 / 21.2 with 14, tools/fc4_small_probe.py) -- linear_finish_kernel was a launch of 5.6 us for 0.16 MB (profiles/r05z2_kernel_stats_a2c_pixel_16.txt).
DRA_API int dra_policy_heads_given_fold(const float* slabs, int n_slabs, const float* fold_bias, const float* w0, const float* b0,
                                        const float* w1, const float* b1, const int64_t* action, int batch, int n_actions,
                                        float* out_log_pi_a, float* out_entropy, float* out_v, float* out_logits, float* out_phi,
                                        void* stream) {
  if (!slabs || (n_slabs != 8 && n_slabs != 14) || !fold_bias || !w0 || !w1 || !action || !out_log_pi_a || !out_entropy || !out_v ||
      !out_logits || !out_phi || batch < 1 || batch > 65536 || n_actions < 1 || n_actions > 64)
    return DRA_EINVAL;
  PolicyHeadArgs h = head_args(nullptr, w0, b0, w1, b1, nullptr, action, batch, 512, n_actions, nullptr, out_log_pi_a, out_entropy,
                               out_v, out_logits);
  h.slabs = slabs; h.fold_bias = fold_bias; h.out_x = out_phi;
  // one workgroup per row (the fold spread over four waves): the wave-per-row form took 10.2 us at 256 rows -- 64 workgroups for
  // 4 MB of slabs (profiles/r05fb_kernel_stats_ppo_pixel_8.txt); same sums in the same order
  if (n_slabs == 8) hipLaunchKernelGGL(policy_heads_foldwg_kernel<8>, dim3(batch), dim3(256), 0, dra_stream(stream), h);
  else hipLaunchKernelGGL(policy_heads_foldwg_kernel<14>, dim3(batch), dim3(256), 0, dra_stream(stream), h);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// A rollout step's fc4 finish + policy head (rollout_roles.h policy_head_row_fold_wg): one workgroup per row, the features
// folded from the 28 K-slice partial sums of dra_linear_fwd_slabs_one(ksplit = 28).  K = 512.

DRA_API int dra_policy_heads_sample_fold28(const float* slabs, const float* fold_bias, const float* w0, const float* b0,
                                           const float* w1, const float* b1, const float* uniform, int batch, int n_actions,
                                           int64_t* out_action, float* out_log_pi_a, float* out_entropy, float* out_v,
                                           void* stream) {
  if (!slabs || !fold_bias || !w0 || !w1 || !uniform || !out_action || !out_log_pi_a || !out_entropy || !out_v || batch < 1 ||
      batch > 65536 || n_actions < 1 || n_actions > 64)
    return DRA_EINVAL;
  PolicyHeadArgs h = head_args(nullptr, w0, b0, w1, b1, uniform, nullptr, batch, 512, n_actions, out_action, out_log_pi_a, out_entropy,
                               out_v, nullptr);
  h.slabs = slabs; h.fold_bias = fold_bias;
  hipLaunchKernelGGL(policy_heads_foldwg_kernel<28>, dim3(batch), dim3(256), 0, dra_stream(stream), h);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Backward of that head in ONE launch: categorical_bwd_kernel (dlogits from g_log_pi_a / g_entropy), linear_pair_bwd_kernel
// (d phi, both layers' weight / bias gradients) and -- when phi is the output of a fused ReLU -- act_bwd_kernel's mask were three
// launches of ~5-12 us on the update's dependent chain (profiles/r05r_kernel_stats_*).  Same sums in the same order:
//   workgroups [0, B): input row b.  Lanes a < A of wave 0 form dlogits[b][a] (each walks the row's A logits itself: the
//     forward's ascending sums), then thread t owns k = t, t + 256:  dx[b][k] = sum_a dlogits[a] W0[a][k] + g_v[b] W1[0][k],
//     times [x[b][k] > 0] with relu_mask
//   workgroups [B, B + 2 (A + 1)): (output row, half of K).  The row's gradient column over all samples is formed in LDS first
//     (thread t: samples t, t + 256, ...), then dW[o][k] = sum_b g[b][o] x[b][k], db[o] = sum_b g[b][o] in ascending b.
constexpr int kHeadsBwdMaxBatch = 8192;    // the gradient column in LDS (32 KB)
__global__ void __launch_bounds__(256)
policy_heads_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ action, const float* __restrict__ g_lp,
                        const float* __restrict__ g_ent, const float* __restrict__ g_v, const float* __restrict__ x,
                        const float* __restrict__ w0, const float* __restrict__ w1, float* __restrict__ dx,
                        float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dw1, float* __restrict__ db1,
                        int B, int K, int A, int relu_mask) {
  extern __shared__ float s_g[];      // dx workgroups: [A]; weight-gradient workgroups: [B]
  const int t = threadIdx.x;
  if ((int)blockIdx.x < B) {
    if (!dx) return;
    const int b = blockIdx.x;
    if (t < A) {
      const float* xr = logits + (int64_t)b * A;
      float lse, ent;
      categorical_row_stats(xr, A, &lse, &ent);
      s_g[t] = categorical_dlogit(xr[t], lse, ent, (int64_t)t == action[b], g_lp ? g_lp[b] : 0.f, g_ent ? g_ent[b] : 0.f);
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
    const int k0 = t, k1 = t + 256;
    for (int o = 0; o < A; ++o) {
      const float g = s_g[o];
      if (k0 < K) a0 += g * w0[(int64_t)o * K + k0];
      if (k1 < K) a1 += g * w0[(int64_t)o * K + k1];
    }
    float c0 = 0.f, c1 = 0.f;
    {
      const float g = g_v ? g_v[b] : 0.f;
      if (k0 < K) c0 += g * w1[k0];
      if (k1 < K) c1 += g * w1[k1];
    }
    if (k0 < K) { const float v = a0 + c0; dx[(int64_t)b * K + k0] = (!relu_mask || x[(int64_t)b * K + k0] > 0.f) ? v : 0.f; }
    if (k1 < K) { const float v = a1 + c1; dx[(int64_t)b * K + k1] = (!relu_mask || x[(int64_t)b * K + k1] > 0.f) ? v : 0.f; }
    return;
  }
  const int r = blockIdx.x - B, row = r >> 1, k = (r & 1) * 256 + t;
  const bool first = row < A;
  const int o = first ? row : 0;
  for (int b = t; b < B; b += 256) {
    float g;
    if (first) {
      const float* xr = logits + (int64_t)b * A;
      float lse, ent;
      categorical_row_stats(xr, A, &lse, &ent);
      g = categorical_dlogit(xr[o], lse, ent, (int64_t)o == action[b], g_lp ? g_lp[b] : 0.f, g_ent ? g_ent[b] : 0.f);
    } else {
      g = g_v ? g_v[b] : 0.f;
    }
    s_g[b] = g;
  }
  __syncthreads();
  float acc = 0.f, accb = 0.f;
  int b = 0;
  for (; b + 32 <= B; b += 32) {        // linear_pair_bwd_kernel's rounds: 32 samples' loads in flight, ascending b
    float h[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) h[i] = k < K ? x[(int64_t)(b + i) * K + k] : 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float d = s_g[b + i]; acc += d * h[i]; accb += d; }
  }
  for (; b + 8 <= B; b += 8) {
    float h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = k < K ? x[(int64_t)(b + i) * K + k] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = s_g[b + i]; acc += d * h[i]; accb += d; }
  }
  for (; b < B; ++b) {
    const float d = s_g[b];
    acc += d * (k < K ? x[(int64_t)b * K + k] : 0.f);
    accb += d;
  }
  float* __restrict__ dw = first ? dw0 : dw1;
  float* __restrict__ db = first ? db0 : db1;
  if (k < K) dw[(int64_t)o * K + k] = acc;
  if (k == 0) db[o] = accb;
}

DRA_API int dra_policy_heads_bwd(const float* logits, const int64_t* action, const float* g_log_pi_a, const float* g_entropy,
                                 const float* g_v, const float* x, const float* w0, const float* w1, float* dx, float* dw0,
                                 float* db0, float* dw1, float* db1, int batch, int in_features, int n_actions, int relu_mask,
                                 void* stream) {
  if (!logits || !action || !x || !w0 || !w1 || !dw0 || !db0 || !dw1 || !db1 || batch < 1 || batch > kHeadsBwdMaxBatch ||
      in_features < 1 || in_features > 512 || n_actions < 1 || n_actions > 64)
    return DRA_EINVAL;
  const size_t lds = sizeof(float) * (size_t)(batch > 64 ? batch : 64);
  hipLaunchKernelGGL(policy_heads_bwd_kernel, dim3(batch + 2 * (n_actions + 1)), dim3(256), lds, dra_stream(stream), logits, action,
                     g_log_pi_a, g_entropy, g_v, x, w0, w1, dx, dw0, db0, dw1, db1, batch, in_features, n_actions, relu_mask);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_linear_fwd_pair(const float* x, const float* w0, const float* b0, float* y0, int out0, const float* w1,
                                const float* b1, float* y1, int out1, int batch, int in_features, int act, void* stream) {
  if (!x || !w0 || !w1 || !y0 || !y1 || batch < 1 || batch > 65536 || in_features < 1 || in_features > 512 || out0 < 1 || out1 < 1)
    return DRA_EINVAL;
  hipLaunchKernelGGL(linear_heads_rows_kernel, dim3((batch + 3) / 4), dim3(256), 0, dra_stream(stream), x, w0, b0, y0, out0, w1, b1,
                     y1, out1, batch, in_features, act);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Backward of the two heads above in one launch (autograd hands both output gradients over together):
//   dx[b][k] = sum_o g0[b][o] W0[o][k] + sum_o g1[b][o] W1[o][k]        workgroups [0, B): one input row each, thread t owns k = t, t + 256
//   dW_h[o][k] = sum_b g_h[b][o] x[b][k],  db_h[o] = sum_b g_h[b][o]     workgroups [B, B + 2 (O0 + O1)): (output row, half of K)
// Five launches (two input gradients, their sum, two weight gradients) before: 31 us of an A2C update for ~1 MFLOP.  Sums in
// ascending o / b, one multiply and one add per term.
__global__ void __launch_bounds__(256)
linear_pair_bwd_kernel(const float* __restrict__ g0, const float* __restrict__ g1, const float* __restrict__ x,
                       const float* __restrict__ w0, const float* __restrict__ w1, float* __restrict__ dx,
                       float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dw1, float* __restrict__ db1,
                       int B, int K, int O0, int O1) {
  const int t = threadIdx.x;
  if ((int)blockIdx.x < B) {
    if (!dx) return;
    const int b = blockIdx.x;
    float a0 = 0.f, a1 = 0.f;
    const int k0 = t, k1 = t + 256;
    for (int o = 0; o < O0; ++o) {
      const float g = g0[(int64_t)b * O0 + o];
      if (k0 < K) a0 += g * w0[(int64_t)o * K + k0];
      if (k1 < K) a1 += g * w0[(int64_t)o * K + k1];
    }
    float c0 = 0.f, c1 = 0.f;
    for (int o = 0; o < O1; ++o) {
      const float g = g1[(int64_t)b * O1 + o];
      if (k0 < K) c0 += g * w1[(int64_t)o * K + k0];
      if (k1 < K) c1 += g * w1[(int64_t)o * K + k1];
    }
    if (k0 < K) dx[(int64_t)b * K + k0] = a0 + c0;
    if (k1 < K) dx[(int64_t)b * K + k1] = a1 + c1;
    return;
  }
  const int r = blockIdx.x - B, row = r >> 1, k = (r & 1) * 256 + t;
  const bool first = row < O0;
  const int o = first ? row : row - O0, O = first ? O0 : O1;
  const float* __restrict__ g = first ? g0 : g1;
  float acc = 0.f, accb = 0.f;
  int b = 0;
  // 32 samples per round (their 64 loads in flight together), then 8, then the tail: a PPO minibatch of 256 walked eight samples
  // at a time was 32 dependent memory round trips, 18 us (profiles/r04ap_kernel_stats_ppo_pixel_8.txt).  Ascending b throughout.
  for (; b + 32 <= B; b += 32) {
    float d[32], h[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { d[i] = g[(int64_t)(b + i) * O + o]; h[i] = k < K ? x[(int64_t)(b + i) * K + k] : 0.f; }
#pragma unroll
    for (int i = 0; i < 32; ++i) { acc += d[i] * h[i]; accb += d[i]; }
  }
  for (; b + 8 <= B; b += 8) {
    float d[8], h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = g[(int64_t)(b + i) * O + o]; h[i] = k < K ? x[(int64_t)(b + i) * K + k] : 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc += d[i] * h[i]; accb += d[i]; }
  }
  for (; b < B; ++b) {
    const float d = g[(int64_t)b * O + o];
    acc += d * (k < K ? x[(int64_t)b * K + k] : 0.f);
    accb += d;
  }
  float* __restrict__ dw = first ? dw0 : dw1;
  float* __restrict__ db = first ? db0 : db1;
  if (k < K) dw[(int64_t)o * K + k] = acc;
  if (k == 0) db[o] = accb;
}

DRA_API int dra_linear_bwd_pair(const float* g0, const float* g1, const float* x, const float* w0, const float* w1, float* dx,
                                float* dw0, float* db0, float* dw1, float* db1, int batch, int in_features, int out0, int out1,
                                void* stream) {
  if (!g0 || !g1 || !x || !w0 || !w1 || !dw0 || !db0 || !dw1 || !db1 || batch < 1 || in_features < 1 || in_features > 512 ||
      out0 < 1 || out1 < 1)
    return DRA_EINVAL;
  hipLaunchKernelGGL(linear_pair_bwd_kernel, dim3(batch + 2 * (out0 + out1)), dim3(256), 0, dra_stream(stream), g0, g1, x, w0, w1, dx,
                     dw0, db0, dw1, db1, batch, in_features, out0, out1);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_linear_fwd(int nz, const float* const* x, const float* const* w, const float* const* bias,
                           float* const* y, int batch, int in_features, int out_features, int act, float* workspace,
                           int64_t workspace_floats, void* stream) {
  if (nz < 1 || nz > kMaxZ || batch < 1 || in_features < 1 || out_features < 1 || !x || !w || !y) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  // latency regime: few rows and a reduction that fits a wave's registers; large batches amortise the GEMM's chunks
  if (in_features <= 512 && batch <= 128 && (int64_t)batch * out_features <= 65536 && gemv_enabled()) {
    LinPtrs q;
    for (int z = 0; z < nz; ++z) {
      if (!x[z] || !w[z] || !y[z]) return DRA_EINVAL;
      q.x[z] = x[z]; q.w[z] = w[z]; q.bias[z] = bias ? bias[z] : nullptr; q.y[z] = y[z];
    }
    const dim3 grid((out_features + 3) / 4, nz, (batch + 31) / 32);
    const size_t lds = (size_t)(batch < 32 ? batch : 32) * in_features * sizeof(float);
    const int kv = (in_features + 63) / 64;
    if (kv <= 1) hipLaunchKernelGGL(linear_gemv_kernel<1>, grid, dim3(256), lds, st, q, batch, in_features, out_features, act);
    else if (kv <= 2) hipLaunchKernelGGL(linear_gemv_kernel<2>, grid, dim3(256), lds, st, q, batch, in_features, out_features, act);
    else if (kv <= 4) hipLaunchKernelGGL(linear_gemv_kernel<4>, grid, dim3(256), lds, st, q, batch, in_features, out_features, act);
    else hipLaunchKernelGGL(linear_gemv_kernel<8>, grid, dim3(256), lds, st, q, batch, in_features, out_features, act);
    DRA_LAUNCH_CHECK();
    return DRA_OK;
  }
  if (in_features > 512 && in_features <= 4096 && (in_features & 3) == 0 && batch <= 32 && gemv_enabled()) {
    LinPtrs q;
    bool aligned = true;
    for (int z = 0; z < nz; ++z) {
      if (!x[z] || !w[z] || !y[z]) return DRA_EINVAL;
      q.x[z] = x[z]; q.w[z] = w[z]; q.bias[z] = bias ? bias[z] : nullptr; q.y[z] = y[z];
      aligned = aligned && !((((uintptr_t)x[z]) | ((uintptr_t)w[z])) & 15);
    }
    if (aligned) {
      const int rq = ((((in_features >> 2) + 3) >> 2) + 63) / 64;     // float4 per lane of a K quarter
      // (sixteen samples per round with a row's K in eighths -- gemv_rows_body<2, 16, 8>, one round for A2C's 16 environments
      // instead of two -- measured SLOWER: a2c_pixel 215 k against 224 k env-steps/s, profiles/r05y_bench_agents_gemv16.jsonl; and
      // its summation tree differs from the eight-sample form the fused DQN learner's actor reproduces.  Not dispatched.)
      if (rq <= 2) return launch_gemv_rows<2>(q, nz, batch, in_features, out_features, act, st);
      return launch_gemv_rows<4>(q, nz, batch, in_features, out_features, act, st);
    }
  }
  if (in_features == 3136 && batch > 32 && batch <= 4096 && workspace &&
      (int64_t)nz * 8 * batch * out_features <= workspace_floats && gemv_enabled()) {
    // fc4 of NatureConvBody at update batch sizes (A2C 80, a PPO minibatch of 256): the learner's one-pass K-slice kernel (both
    // operands through LDS once, K slices) + the slab finish, instead of the K-chunked GEMM (13.9 + 5.1 us
    // at batch 80, 30.8 + 5 us at 256: profiles/r04ag_kernel_stats_a2c_pixel_16.txt, r04o_kernel_stats_ppo_pixel_8.txt)
    bool aligned = true;
    LinPtrs q;
    for (int z = 0; z < nz; ++z) {
      if (!x[z] || !w[z] || !y[z]) return DRA_EINVAL;
      q.x[z] = x[z]; q.w[z] = w[z]; q.bias[z] = bias ? bias[z] : nullptr; q.y[z] = y[z];
      aligned = aligned && !((((uintptr_t)x[z]) | ((uintptr_t)w[z])) & 15);
    }
    if (aligned) {
      // (8 slices from round 5 on: 8.9 / 17.3 us at 80 / 256 rows against 10.9 / 21.2 with 14, tools/fc4_small_probe.py -- the same
      // count nets._Fc4PolicyHeadFn folds inside its head launch, so both forms give the same bits)
      int rc = dra_linear_fwd_slabs_one(nz, x, w, batch, in_features, out_features, 8, workspace, stream);
      if (rc != DRA_OK) return rc;
      hipLaunchKernelGGL(linear_finish_kernel, dim3((batch * out_features + 255) / 256, nz), dim3(256), 0, st, q,
                         (const float*)workspace, 8, batch, out_features, act);
      DRA_LAUNCH_CHECK();
      return DRA_OK;
    }
  }
  LinFwd<32, 32, 64> p;
  p.M = out_features; p.N = batch; p.K = in_features;
  for (int z = 0; z < nz; ++z) {
    if (!x[z] || !w[z] || !y[z]) return DRA_EINVAL;
    p.q.x[z] = x[z]; p.q.w[z] = w[z]; p.q.bias[z] = bias ? bias[z] : nullptr; p.q.y[z] = y[z];
  }
  // enough workgroups to cover the chip: split K when the output alone gives too few tiles
  const int tiles = ((out_features + 31) / 32) * ((batch + 31) / 32) * nz;
  const int chunks = (in_features + 63) / 64;
  int ksplit = 1;
  if (tiles < 128 && chunks >= 16) {  // K >= 1024: heads (K = 512) stay single-launch
    ksplit = (256 + tiles - 1) / tiles;
    if (ksplit > chunks / 2) ksplit = chunks / 2;
    if (ksplit > 32) ksplit = 32;
    while (ksplit > 1 && (int64_t)nz * ksplit * batch * out_features > workspace_floats) --ksplit;
    if (!workspace) ksplit = 1;
  }
  // every split must own >= 1 chunk so that all slab elements are written
  while (ksplit > 1 && (ksplit - 1) * ((chunks + ksplit - 1) / ksplit) >= chunks) --ksplit;
  p.slabs = workspace; p.ksplit = ksplit; p.act = act;
  int rc = launch_igemm(p, nz, ksplit, st);
  if (rc != DRA_OK) return rc;
  if (ksplit > 1) {
    hipLaunchKernelGGL(linear_finish_kernel, dim3((batch * out_features + 255) / 256, nz), dim3(256), 0, st, p.q,
                       (const float*)workspace, ksplit, batch, out_features, act);
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

DRA_API int dra_linear_fwd_slabs(int nz, const float* const* x, const float* const* w, int batch, int in_features,
                                 int out_features, int ksplit, float* slabs, void* stream) {
  if (nz < 1 || nz > kMaxZ || batch < 1 || in_features < 1 || out_features < 1 || ksplit < 2 || !x || !w || !slabs)
    return DRA_EINVAL;
  LinFwd<32, 32, 64> p;
  p.M = out_features; p.N = batch; p.K = in_features;
  for (int z = 0; z < nz; ++z) {
    if (!x[z] || !w[z]) return DRA_EINVAL;
    p.q.x[z] = x[z]; p.q.w[z] = w[z]; p.q.bias[z] = nullptr; p.q.y[z] = nullptr;
  }
  p.slabs = slabs; p.ksplit = ksplit; p.act = ACT_NONE;
  return launch_igemm(p, nz, ksplit, dra_stream(stream));
}

DRA_API int dra_linear_bwd_w(const float* dy, const float* x, float* dw, float* db, int batch, int in_features,
                             int out_features, void* stream) {
  if (!dy || !x || !dw || batch < 1 || in_features < 1 || out_features < 1) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  if (out_features >= 64) {
    LinWgrad<64, 64, 32> p;
    p.M = out_features; p.N = in_features + 1; p.K = batch; p.I = in_features; p.dy = dy; p.x = x; p.dw = dw; p.db = db;
    return launch_igemm(p, 1, 1, st);
  }
  LinWgrad<32, 32, 64> p;
  p.M = out_features; p.N = in_features + 1; p.K = batch; p.I = in_features; p.dy = dy; p.x = x; p.dw = dw; p.db = db;
  return launch_igemm(p, 1, 1, st);
}

DRA_API int dra_linear_bwd_x(const float* dy, const float* w, const float* xact, float* dx, int batch, int in_features,
                             int out_features, int act, void* stream) {
  if (!dy || !w || !dx || batch < 1 || in_features < 1 || out_features < 1) return DRA_EINVAL;
  if (out_features == 512 && in_features >= 1024 && gemv_enabled())     // fc4-shaped: the one-pass role (fused.hip)
    return dra_linear_bwd_x_one512(dy, w, xact, dx, batch, in_features, act, stream);
  LinDgrad<32, 32, 64> p;
  p.M = batch; p.N = in_features; p.K = out_features; p.dy = dy; p.w = w; p.xact = xact; p.dx = dx; p.act = act;
  return launch_igemm(p, 1, 1, dra_stream(stream));
}

// dpre = dy * act'(y): used by the stand-alone autograd wrappers (the fused learner folds this
// into the dgrad epilogues instead).
__global__ void __launch_bounds__(256)
act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out, int64_t n, int act) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = dy[i] * act_grad(y[i], act);
}

DRA_API int dra_act_bwd(const float* dy, const float* y, float* dpre, int64_t n, int act, void* stream) {
  if (!dy || !y || !dpre || n < 0) return DRA_EINVAL;
  if (n == 0) return DRA_OK;
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)b), dim3(256), 0, dra_stream(stream), dy, y, dpre, n, act);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
