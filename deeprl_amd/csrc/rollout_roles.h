// Device-side pieces shared by the stand-alone kernels of igemm.hip and the fused rollout launches of conv_v2.hip (an A2C / PPO
// rollout step over NatureConvBody as four launches: [conv1 | policy head of the previous step], conv2, conv3, fc4).
// One statement of each piece, so that a fused launch and the separate kernels give the same bits.
#pragma once
#include "common.h"

__device__ __forceinline__ float rr_act(float v, int act) {      // DRA_ACT_*: igemm.h act_apply / conv_v2.hip v2_act
  if (act == DRA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DRA_ACT_TANH) return tanhf(v);
  return v;
}

// Two linear heads on the same features: the wave's work for input row b (in_features K <= 512: 8 registers per lane, k = lane +
// 64 i); the weight rows of up to eight outputs are requested together with them -- one memory round trip, no LDS, no barrier --
// then the eight dot products and their wave butterflies run interleaved.  sink(o, value) runs on lane 0 for every output o of
// [0, O0 + O1).  Per-output arithmetic (lane-strided partial sums in i order, then the butterfly) is linear_gemv_kernel's.
// load_x(k): feature k of the row (a plain row of x, or -- the update's fc4 handing its K-slice partial sums over -- their fold).
template <class XLoad, class Sink>
__device__ __forceinline__ void heads_row_outputs_from(XLoad load_x, const float* __restrict__ w0, const float* __restrict__ b0, int O0,
                                                       const float* __restrict__ w1, const float* __restrict__ b1, int O1, int K,
                                                       int act, int lane, Sink sink) {
  float xv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xv[i] = (lane + 64 * i < K) ? load_x(lane + 64 * i) : 0.f;
  const int OT = O0 + O1;
  for (int oc = 0; oc < OT; oc += 8) {
    float wv[8][8], bias[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int o = min(oc + u, OT - 1);
      const float* __restrict__ row = o < O0 ? w0 + (int64_t)o * K : w1 + (int64_t)(o - O0) * K;
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[u][i] = (lane + 64 * i < K) ? row[lane + 64 * i] : 0.f;
      const float* __restrict__ bp = o < O0 ? b0 : b1;
      bias[u] = bp ? bp[o < O0 ? o : o - O0] : 0.f;
    }
    float part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) p += (lane + 64 * i < K) ? xv[i] * wv[u][i] : 0.f;
      part[u] = p;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] += __shfl_xor(part[u], off, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = oc + u;
        if (o < OT) sink(o, rr_act(part[u] + bias[u], act));
      }
    }
  }
}

template <class Sink>
__device__ __forceinline__ void heads_row_outputs(const float* __restrict__ x, const float* __restrict__ w0,
                                                  const float* __restrict__ b0, int O0, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, int O1, int b, int K, int act, int lane, Sink sink) {
  heads_row_outputs_from([&](int k) { return x[(int64_t)b * K + k]; }, w0, b0, O0, w1, b1, O1, K, act, lane, sink);
}

// A categorical actor-critic's whole policy head for input row b (network_heads.py:240-255): logits = x W0^T + b0 [A <= 64],
// v = x w1^T + b1, then Categorical(logits) of the row on the lane that holds the outputs -- inverse-CDF sample from uniform[b]
// (action_in == nullptr) or the given action, log_pi_a, entropy (common.h categorical_row).  so: >= A + 1 floats of LDS owned by
// the calling wave.
// KS > 0 (template argument of policy_head_row): the features are the fold of KS K-slice partial sums slabs[s][b][k] of the layer
// below (fc4's one-pass forward) + fold_bias[k], through a ReLU -- linear_finish_kernel's sum, slab 0 first -- and are written to
// out_x [B][K] for the backward pass; KS == 0: x [B][K] as is.
struct PolicyHeadArgs {
  const float *x, *w0, *b0, *w1, *b1, *uniform;
  const int64_t* action_in;
  int64_t* out_action;
  float *out_lp, *out_ent, *out_v, *out_logits;
  int B, K, A;
  const float *slabs, *fold_bias;
  float* out_x;
};
template <int KS = 0>
__device__ __forceinline__ void policy_head_row(const PolicyHeadArgs& h, int b, int lane, float* so) {
  auto sink = [&](int o, float v) { so[o] = v; };
  if constexpr (KS > 0) {
    heads_row_outputs_from([&](int k) {
      const float* sl = h.slabs + (int64_t)b * h.K + k;
      float part[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) part[s] = sl[(int64_t)s * h.B * h.K];
      float v = part[0];
#pragma unroll
      for (int s = 1; s < KS; ++s) v += part[s];
      v = rr_act(v + h.fold_bias[k], DRA_ACT_RELU);
      h.out_x[(int64_t)b * h.K + k] = v;
      return v;
    }, h.w0, h.b0, h.A, h.w1, h.b1, 1, h.K, /*act=*/0, lane, sink);
  } else {
    heads_row_outputs(h.x, h.w0, h.b0, h.A, h.w1, h.b1, 1, b, h.K, /*act=*/0, lane, sink);
  }
  if (lane == 0) {      // (the same lane wrote so[]: program order, no barrier)
    int64_t act;
    float lp, ent;
    categorical_row(so, h.A, h.action_in != nullptr, h.action_in ? h.action_in[b] : 0, h.uniform ? h.uniform[b] : 0.f, &act, &lp,
                    &ent);
    if (h.out_action) h.out_action[b] = act;
    h.out_lp[b] = lp;
    h.out_ent[b] = ent;
    h.out_v[b] = so[h.A];
    if (h.out_logits)
      for (int a = 0; a < h.A; ++a) h.out_logits[(int64_t)b * h.A + a] = so[a];
  }
}

// The same head for ONE row per WORKGROUP of four waves, the row's features folded from KS K-slice partial sums first (fc4 of a
// rollout step through the one-pass K-slice kernel with 28 slices: 4.3 us at 8-32 samples against the eight-wave GEMV's 5.6 / 8.1 /
// 13.2 us, tools/fc4_small_probe.py -- its finish happens here, and in a rollout this head rides in the NEXT step's conv1 launch,
// where the fold's loads cost nothing on the chain): thread t folds features t and t + 256 (2 KS loads in flight, slab 0 first, +
// bias, ReLU: linear_finish_kernel's sum) into LDS, then wave 0 runs the head on them.  s_phi: 512 floats, so: >= A + 1 floats.
template <int KS>
__device__ __forceinline__ void policy_head_row_fold_wg(const PolicyHeadArgs& h, int b, float* s_phi, float* so) {
  const int t = threadIdx.x;
  float part[2][KS];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float* sl = h.slabs + (int64_t)b * h.K + t + 256 * j;
#pragma unroll
    for (int s = 0; s < KS; ++s) part[j][s] = sl[(int64_t)s * h.B * h.K];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = t + 256 * j;
    float v = part[j][0];
#pragma unroll
    for (int s = 1; s < KS; ++s) v += part[j][s];
    v = rr_act(v + h.fold_bias[k], DRA_ACT_RELU);
    s_phi[k] = v;
    if (h.out_x) h.out_x[(int64_t)b * h.K + k] = v;
  }
  __syncthreads();
  if (t >= 64) return;
  const int lane = t;
  heads_row_outputs_from([&](int k) { return s_phi[k]; }, h.w0, h.b0, h.A, h.w1, h.b1, 1, h.K, /*act=*/0, lane,
                         [&](int o, float v) { so[o] = v; });
  if (lane == 0) {
    int64_t act;
    float lp, ent;
    categorical_row(so, h.A, h.action_in != nullptr, h.action_in ? h.action_in[b] : 0, h.uniform ? h.uniform[b] : 0.f, &act, &lp,
                    &ent);
    if (h.out_action) h.out_action[b] = act;
    h.out_lp[b] = lp;
    h.out_ent[b] = ent;
    h.out_v[b] = so[h.A];
    if (h.out_logits)
      for (int a = 0; a < h.A; ++a) h.out_logits[(int64_t)b * h.A + a] = so[a];
  }
}

// A wide linear layer at rollout batch sizes (fc4 of NatureConvBody, 3136 -> 512, for the 8 / 16 environments of one rollout
// step): a workgroup of EIGHT waves owns 8 / WPR output rows (o2 = index of the group), each row's reduction split over WPR waves
// (K parts); a lane keeps its R float4 of the weight row in registers and, per round, the matching float4 of up to RB input
// rows -- all requested before the first is used; per sample: products, a wave sum, the WPR part sums met in LDS as a fixed
// pairwise tree.  Per-sample arithmetic depends on (R, WPR) only.  K % 4 == 0, K <= 4 * 64 * WPR * R.
//   <4, 8, 4>: two rows per workgroup, eight samples per round (<= 8 samples: one round; 16 samples were two dependent rounds,
//              9.4 us against 6.3: profiles/r05t_kernel_stats_a2c_pixel_16.txt)
//   <2, 16, 8>: one row per workgroup, sixteen samples per round (the same 128 registers of inputs in flight)
// s_part: [32][8] floats of LDS.
template <int R, int RB, int WPR>
__device__ __forceinline__ void gemv_rows_body(const float* __restrict__ x, const float* __restrict__ w,
                                               const float* __restrict__ bias, float* __restrict__ y, int o2, int b0, int B, int K,
                                               int O, int act, float (*s_part)[8]) {
  static_assert(WPR == 4 || WPR == 8, "a row's K parts: 4 or 8 waves");
  constexpr int RPW = 8 / WPR;                             // output rows per workgroup
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = wave / WPR, part = wave % WPR;
  const int o = o2 * RPW + row;
  const int nv = K >> 2, nvq = (nv + WPR - 1) / WPR;       // float4 per row / per part
  const int v0 = part * nvq, v1 = min(nv, v0 + nvq);
  const int nb = min(32, B - b0);
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w + (int64_t)min(o, O - 1) * K);
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + (int64_t)b0 * K);
  float4 wv[R];
  int vi[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int v = v0 + lane + 64 * i;
    vi[i] = v < v1 ? v : -1;
    wv[i] = w4[v < v1 ? v : (v1 > v0 ? v1 - 1 : 0)];
  }
  for (int bb = 0; bb < nb; bb += RB) {
    float4 xv[RB][R];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int b = min(bb + u, nb - 1);
#pragma unroll
      for (int i = 0; i < R; ++i) xv[u][i] = x4[(int64_t)b * nv + (vi[i] >= 0 ? vi[i] : 0)];
    }
    float acc[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      float a_ = 0.f;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const float4 a = wv[i], xx = xv[u][i];
        if (vi[i] >= 0) a_ += (a.x * xx.x + a.y * xx.y) + (a.z * xx.z + a.w * xx.w);
      }
      acc[u] = a_;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {      // RB independent butterflies, interleaved by offset
#pragma unroll
      for (int u = 0; u < RB; ++u) acc[u] += __shfl_xor(acc[u], off, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < RB; ++u)
        if (bb + u < nb) s_part[bb + u][wave] = acc[u];
    }
  }
  __syncthreads();
  // thread t < RPW * nb: (row, sample)
  const int t = threadIdx.x;
  if (t < RPW * nb) {
    const int r = t / nb, b = t - r * nb, oo = o2 * RPW + r;
    if (oo < O) {
      const float* sp = &s_part[b][WPR * r];
      float v = (sp[0] + sp[1]) + (sp[2] + sp[3]);
      if constexpr (WPR == 8) v = v + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
      y[(int64_t)(b0 + b) * O + oo] = rr_act(v + (bias ? bias[oo] : 0.f), act);
    }
  }
}

// (Measured and removed, round 5 -- the eight-wave form above stayed the fastest at 8 AND 16 samples:
//   gemv_rows_body<2, 16, 8>  sixteen samples per round, K in eighths             a2c_pixel 215 k against 224-226 k env-steps/s
//   eight-sample groups on separate workgroups (two per output pair at 16)      221 k
//   two output rows per wave, four waves per workgroup (half the input reads)   222 k; ppo_pixel 116.1 k against 117.6 k
//  profiles/r05y_bench_agents_gemv16.jsonl, r05z3_bench_agents_gemv_split.jsonl, r05z4_bench_agents_gemv_pair.jsonl (second pair
//  of lines: the eight-wave form in the same call): the launch is a latency chain -- weights, inputs,
//  butterflies, LDS exchange -- that eight waves per workgroup hide best, not a traffic problem.)
