// Fused per-batch loss kernels, forward + backward in one launch each.
//   K4  dra_td_loss   deep_rl/agent/DQN_agent.py:78-99 (+ PER branch :120-127)
//   K5  dra_c51_loss  deep_rl/agent/CategoricalDQN_agent.py:60-89
//   K6  dra_qr_loss   deep_rl/agent/QuantileRegressionDQN_agent.py:55-77 (+ utils/torch_utils.py:47-48)
//   K10 dra_ppo_loss  deep_rl/agent/PPO_agent.py:77-86 ; dra_a2c_loss deep_rl/agent/A2C_agent.py:55-62
// All are bandwidth/latency-bound on a few KB; each is one launch (two for QR) instead of the
// reference's 15-30 ATen ops, and never materialises the [B,N,N] / [N,B,N] intermediates.
// Actions arrive as the ring's raw int64 records or as f32 (what `tensor()` makes of them).
#include "common.h"

// (clamped to [0, n_actions): an action record that was never written -- a sampling bug upstream -- must show up as a wrong
// number, not turn into a wild pointer and a GPU fault)
__device__ __forceinline__ int64_t load_action(const void* a, int is_i64, int b, int n_actions) {
  const int64_t v = is_i64 ? reinterpret_cast<const int64_t*>(a)[b] : (int64_t)reinterpret_cast<const float*>(a)[b];
  return v < 0 ? 0 : (v >= n_actions ? (int64_t)n_actions - 1 : v);
}

// block-wide reductions for blockDim <= 1024 (<=16 waves); every thread gets the result.  A single-wave workgroup (the C51
// loss at 51 atoms: ~20 reductions per sample) needs neither LDS nor barriers: the butterfly already leaves the result in
// every lane (and 0.f + v, what the general path computes for one wave, is v).
__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum(v);
  if (blockDim.x <= 64) return v;
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += s_red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* s_red) {
  v = wave_max(v);
  if (blockDim.x <= 64) return v;
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[w] = v;
  __syncthreads();
  float t = s_red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, s_red[i]);
  return t;
}

// ------------------------------------------------------------------------------------------ K4
// One workgroup, one thread per sample (B <= 1024).
__global__ void __launch_bounds__(1024)
td_loss_kernel(const float* __restrict__ q, const float* __restrict__ qn_t, const float* __restrict__ qn_o,
               const void* __restrict__ action, int action_i64, const float* __restrict__ reward,
               const float* __restrict__ mask, int B, int A, float gamma_n, const float* __restrict__ samp_prob,
               float beta, float eps, float alpha, float* __restrict__ out_loss, float* __restrict__ out_dq,
               float* __restrict__ out_delta, float* __restrict__ out_prio, float* __restrict__ out_w) {
  __shared__ float s_red[16];
  const int b = threadIdx.x;
  const bool on = b < B;
  float delta = 0.f, w = 1.f;
  int64_t a = 0;
  if (on) {
    const float* t = qn_t + (int64_t)b * A;
    float qn;
    if (qn_o) {  // double-Q: target value at the online argmax (first max wins, as torch.argmax)
      const float* o = qn_o + (int64_t)b * A;
      int best = 0;
      float bv = o[0];
      for (int k = 1; k < A; ++k) if (o[k] > bv) { bv = o[k]; best = k; }
      qn = t[best];
    } else {
      qn = t[0];
      for (int k = 1; k < A; ++k) qn = fmaxf(qn, t[k]);
    }
    a = load_action(action, action_i64, b, A);
    // rewards + gamma^n * q_next * masks  ->  r + ((g*qn)*m), unfused
    const float target = __fadd_rn(reward[b], __fmul_rn(__fmul_rn(gamma_n, qn), mask[b]));
    delta = __fsub_rn(target, q[(int64_t)b * A + a]);
    if (out_delta) out_delta[b] = delta;
  }
  if (samp_prob) {  // PER: priorities from the PRE-weight vector; weights use the batch size
    float wraw = 0.f;
    if (on) {
      const float ad = fabsf(delta) + eps;
      if (out_prio) out_prio[b] = (alpha == 0.5f) ? sqrtf(ad) : powf(ad, alpha);
      // beta < 0: the exponent is the float behind the probabilities (sampling_prob[B]) -- every kernel argument is then
      // constant across updates and the launch can be replayed from a captured graph
      wraw = powf(samp_prob[b] * (float)B + 1e-6f, -(beta < 0.f ? samp_prob[B] : beta));
    }
    const float wmax = block_max(on ? wraw : -INFINITY, s_red);
    w = wraw / wmax;
    if (on && out_w) out_w[b] = w;
  }
  const float lw = delta * w;
  const float tot = block_sum(on ? 0.5f * lw * lw : 0.f, s_red);
  if (b == 0) *out_loss = tot / (float)B;
  if (on && out_dq) {
    float* g = out_dq + (int64_t)b * A;
    for (int k = 0; k < A; ++k) g[k] = 0.f;
    g[a] = -(lw * w) / (float)B;
  }
}

DRA_API int dra_td_loss(const float* q, const float* q_next_target, const float* q_next_online, const void* action,
                        int action_is_i64, const float* reward, const float* mask, int batch, int n_actions,
                        float gamma_n, const float* sampling_prob, float beta, float replay_eps, float replay_alpha,
                        float* out_loss, float* out_dq, float* out_delta, float* out_prio, float* out_weights,
                        void* stream) {
  if (!q || !q_next_target || !action || !reward || !mask || !out_loss || batch < 1 || batch > 1024 || n_actions < 1)
    return DRA_EINVAL;
  const int threads = ((batch + 63) / 64) * 64;
  hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), q, q_next_target, q_next_online,
                     action, action_is_i64, reward, mask, batch, n_actions, gamma_n, sampling_prob, beta, replay_eps,
                     replay_alpha, out_loss, out_dq, out_delta, out_prio, out_weights);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------ K5
// One workgroup per sample, one thread per atom (N <= 256).  Inputs are LOGITS [B,A,N]; the
// softmax / log-softmax of CategoricalNet (network_heads.py:51-53) is folded in, and the gradient
// comes out w.r.t. the online logits:  d/dlogit_i = (softmax_i * sum_j m_j - m_i) / B  on row a.
__global__ void __launch_bounds__(256)
c51_loss_kernel(const float* __restrict__ logits, const float* __restrict__ logits_t, const float* __restrict__ logits_o,
                const void* __restrict__ action, int action_i64, const float* __restrict__ reward,
                const float* __restrict__ mask, int B, int A, int N, float gamma_n, float v_min, float v_max,
                const float* __restrict__ atoms, float* __restrict__ out_kl, float* __restrict__ out_dlogits,
                const float* __restrict__ weights) {
  extern __shared__ float smem[];  // [N] p_next | [N] atoms
  __shared__ float s_red[16];
  __shared__ int s_anext;
  float* s_p = smem;
  float* s_z = smem + N;
  const int b = blockIdx.x, j = threadIdx.x;
  const bool on = j < N;
  // delta_atom is a python float in the reference (CategoricalDQN_agent.py:49), divided into an f32 tensor
  const float delta_atom = (float)(((double)v_max - (double)v_min) / (double)(N - 1));
  // atoms = tensor(np.linspace(v_min, v_max, N)) is passed in as f32[N]
  const float zj = on ? atoms[j] : 0.f;
  if (on) s_z[j] = zj;

  // every load that does not depend on the greedy action is requested up front: stored action -> its online row, the
  // transition scalars, and the selector rows four at a time (the first version paid one memory round trip per action,
  // one for the target row and two for the online row: 12.6 us at B=32, profiles/r02zu_kernel_stats_c51_*)
  const int64_t a = load_action(action, action_i64, b, A);
  const float r = reward[b], gm = __fmul_rn(gamma_n, mask[b]);
  const float x_on = on ? logits[((int64_t)b * A + a) * N + j] : -INFINITY;
  // greedy next action under the selector net (online for double-Q, else target)
  const float* sel = (logits_o ? logits_o : logits_t) + (int64_t)b * A * N;
  float best_q = -INFINITY, p_best = 0.f;   // p_best: softmax(selector row of the greedy action)_j
  int best_a = 0;
  for (int a0 = 0; a0 < A; a0 += 4) {
    float xs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xs[u] = (on && a0 + u < A) ? sel[(a0 + u) * N + j] : -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (a0 + u < A) {
        const float mx = block_max(xs[u], s_red);
        const float e = on ? expf(xs[u] - mx) : 0.f;
        const float den = block_sum(e, s_red);
        const float qa = block_sum(on ? (e / den) * zj : 0.f, s_red);
        if (qa > best_q) { best_q = qa; best_a = a0 + u; p_best = e / den; }
      }
    }
  }
  if (logits_o) {  // double-Q: p_next = softmax(TARGET logits[b, a_next, :]), the selector was the online net
    if (j == 0) s_anext = best_a;
    __syncthreads();
    const int a_next = s_anext;
    const float x = on ? logits_t[((int64_t)b * A + a_next) * N + j] : -INFINITY;
    const float mx = block_max(x, s_red);
    const float e = on ? expf(x - mx) : 0.f;
    const float den = block_sum(e, s_red);
    if (on) s_p[j] = e / den;
  } else if (on) {
    s_p[j] = p_best;                       // the same exp / sum / divide the selection loop performed on that row
  }
  __syncthreads();
  // projected target m_j = sum_i clamp(1 - |Tz_i - z_j| / dz, 0, 1) * p_i
  float m = 0.f;
  if (on) {
#pragma unroll 8
    for (int i = 0; i < N; ++i) {
      const float zi = s_z[i];
      float tz = __fadd_rn(r, __fmul_rn(gm, zi));
      tz = fminf(fmaxf(tz, v_min), v_max);
      float c = 1.f - fabsf(tz - zj) / delta_atom;
      c = fminf(fmaxf(c, 0.f), 1.f);
      m += c * s_p[i];
    }
  }
  // online log-softmax on row a
  const float x = x_on;
  const float mx = block_max(x, s_red);
  const float e = on ? expf(x - mx) : 0.f;
  const float den = block_sum(e, s_red);
  const float lp = on ? (x - mx) - logf(den) : 0.f;
  const float kl = block_sum(on ? m * logf(m + 1e-5f) - m * lp : 0.f, s_red);
  const float msum = block_sum(m, s_red);
  const float wb = weights ? weights[b] : 1.f;
  if (j == 0) out_kl[b] = kl;
  if (out_dlogits) {
    float* g = out_dlogits + (int64_t)b * A * N;
    for (int k = j; k < A * N; k += blockDim.x) g[k] = 0.f;
    __syncthreads();
    if (on) g[a * N + j] = wb * ((e / den) * msum - m) / (float)B;
  }
}

DRA_API int dra_c51_loss(const float* logits, const float* logits_next_target, const float* logits_next_online,
                         const void* action, int action_is_i64, const float* reward, const float* mask, int batch,
                         int n_actions, int n_atoms, float gamma_n, float v_min, float v_max, const float* atoms,
                         float* out_kl, float* out_dlogits, const float* weights, void* stream) {
  if (!logits || !logits_next_target || !action || !reward || !mask || !atoms || !out_kl || batch < 1 || n_actions < 1 ||
      n_atoms < 2 || n_atoms > 256)
    return DRA_EINVAL;
  const int threads = ((n_atoms + 63) / 64) * 64;
  hipLaunchKernelGGL(c51_loss_kernel, dim3(batch), dim3(threads), 2 * n_atoms * sizeof(float), dra_stream(stream), logits,
                     logits_next_target, logits_next_online, action, action_is_i64, reward, mask, batch, n_actions, n_atoms,
                     gamma_n, v_min, v_max, atoms, out_kl, out_dlogits, weights);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------ K6
// One workgroup per sample, one thread per quantile (N <= 1024).  Pass 1 (thread = online
// quantile i) accumulates d loss / d theta_i over all target quantiles j; pass 2 (thread = target
// quantile j) accumulates the per-j loss over i.  The [N,B,N] tensor (5 MB at N=200) never exists.
__device__ __forceinline__ float huber1(float d) {
  const float ad = fabsf(d);
  return ad < 1.f ? 0.5f * d * d : ad - 0.5f;
}
__device__ __forceinline__ float huber1_grad(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }

// grid (B, 2): blockIdx.y = 0 computes the gradient (thread = online quantile i, sum over target quantiles j),
// blockIdx.y = 1 the per-target-quantile loss (thread = target quantile j, sum over i) -- the two O(N^2) passes of a sample
// run on different CUs.  Every global load of a workgroup is requested before the first reduction (the target rows of ALL
// actions, the online row of the stored action as soon as the action is known); tau_i = (2i+1)/(2N) is formed once per
// thread (fp64, then f32 as the reference's tensor() does) and kept in LDS -- the first version divided in fp64 inside
// the inner loop and loaded the rows one action at a time: 34 us at B=32, N=200 (profiles/r02zu_kernel_stats_qr_*).
// Same arithmetic, same summation order.
__global__ void __launch_bounds__(1024)
qr_loss_kernel(const float* __restrict__ theta, const float* __restrict__ theta_t, const void* __restrict__ action,
               int action_i64, const float* __restrict__ reward, const float* __restrict__ mask, int B, int A, int N,
               float gamma_n, float* __restrict__ out_partial /*[B][N]*/, float* __restrict__ out_dtheta) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NP] T theta | [NP] theta_a | [NP] tau, NP = N rounded up to 4
  __shared__ float s_red[16];
  const int NP = (N + 3) & ~3;         // 16-byte aligned segments: the O(N^2) loops read them four elements per LDS access
  float* s_t = smem;
  float* s_th = smem + NP;
  float* s_tau = smem + 2 * NP;
  const int b = blockIdx.x, i = threadIdx.x, pass = blockIdx.y;
  const bool on = i < N;
  if (pass == 0 && !out_dtheta) return;
  const float* tt = theta_t + (int64_t)b * A * N;
  const int64_t a = load_action(action, action_i64, b, A);
  const float rb = reward[b], gm = __fmul_rn(gamma_n, mask[b]);
  const float th_i = on ? theta[((int64_t)b * A + a) * N + i] : 0.f;
  if (on) s_tau[i] = (float)((2.0 * (double)i + 1.0) / (2.0 * (double)N));
  float best = -INFINITY, x_best = 0.f;   // x_best: this thread's element of the greedy action's target row
  for (int a0 = 0; a0 < A; a0 += 4) {  // a* = argmax_a sum_q theta_target: four rows in flight
    float x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = (on && a0 + u < A) ? tt[(a0 + u) * N + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (a0 + u < A) {
        const float sm = block_sum(x[u], s_red);
        if (sm > best) { best = sm; x_best = x[u]; }
      }
    }
  }
  if (on) {
    // rewards + gamma^n * masks * quantiles_next -> r + ((g*m)*theta')
    s_t[i] = __fadd_rn(rb, __fmul_rn(gm, x_best));
    s_th[i] = th_i;
  }
  __syncthreads();
  if (pass == 0) {
    float* g = out_dtheta + (int64_t)b * A * N;
    for (int k = i; k < A * N; k += blockDim.x) g[k] = 0.f;
    __syncthreads();
    if (on) {
      const float tau = s_tau[i];
      float acc = 0.f;
      const float4* s_t4 = reinterpret_cast<const float4*>(s_t);
      const int n4 = N >> 2;
#pragma unroll 4
      for (int j4 = 0; j4 < n4; ++j4) {     // same terms in the same order, one ds_read_b128 per four of them
        const float4 v = s_t4[j4];
        const float tj[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = tj[u] - th_i;
          acc += huber1_grad(d) * fabsf(tau - (d < 0.f ? 1.f : 0.f));
        }
      }
      for (int j = n4 << 2; j < N; ++j) {
        const float d = s_t[j] - th_i;
        acc += huber1_grad(d) * fabsf(tau - (d < 0.f ? 1.f : 0.f));
      }
      g[a * N + i] = -acc / ((float)N * (float)B);  // loss = mean_j mean_b sum_i rho ; d(d)/d(theta) = -1
    }
    return;
  }
  if (on) {
    const float tj = s_t[i];  // thread plays target quantile j = i
    float l = 0.f;
    const float4* s_th4 = reinterpret_cast<const float4*>(s_th);
    const float4* s_tau4 = reinterpret_cast<const float4*>(s_tau);
    const int n4 = N >> 2;
#pragma unroll 4
    for (int k4 = 0; k4 < n4; ++k4) {
      const float4 hv = s_th4[k4], tv = s_tau4[k4];
      const float th[4] = {hv.x, hv.y, hv.z, hv.w}, ta[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = tj - th[u];
        l += huber1(d) * fabsf(ta[u] - (d < 0.f ? 1.f : 0.f));
      }
    }
    for (int k = n4 << 2; k < N; ++k) {
      const float d = tj - s_th[k];
      l += huber1(d) * fabsf(s_tau[k] - (d < 0.f ? 1.f : 0.f));
    }
    out_partial[(int64_t)b * N + i] = l;
  }
}

__global__ void __launch_bounds__(1024)
qr_finalize_kernel(const float* __restrict__ partial, int B, int N, float* __restrict__ out_loss_vec,
                   float* __restrict__ out_loss) {
  __shared__ float s_red[16];
  const int j = threadIdx.x;
  float l = 0.f;
  if (j < N) {
    int b = 0;
    for (; b + 8 <= B; b += 8) {   // eight samples in flight, added in sample order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u) * N + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) l += v[u];
    }
    for (; b < B; ++b) l += partial[(int64_t)b * N + j];
    l /= (float)B;
    if (out_loss_vec) out_loss_vec[j] = l;
  }
  const float tot = block_sum(j < N ? l : 0.f, s_red);
  if (j == 0) *out_loss = tot / (float)N;
}

DRA_API int dra_qr_loss(const float* theta, const float* theta_next_target, const void* action, int action_is_i64,
                        const float* reward, const float* mask, int batch, int n_actions, int n_quantiles, float gamma_n,
                        float* workspace /*[B*N]*/, float* out_loss_vec, float* out_loss, float* out_dtheta, void* stream) {
  if (!theta || !theta_next_target || !action || !reward || !mask || !workspace || !out_loss || batch < 1 ||
      n_actions < 1 || n_quantiles < 1 || n_quantiles > 1024)
    return DRA_EINVAL;
  const int threads = ((n_quantiles + 63) / 64) * 64;
  hipLaunchKernelGGL(qr_loss_kernel, dim3(batch, 2), dim3(threads), 3 * ((n_quantiles + 3) & ~3) * sizeof(float), dra_stream(stream),
                     theta, theta_next_target, action, action_is_i64, reward, mask, batch, n_actions, n_quantiles, gamma_n,
                     workspace, out_dtheta);
  DRA_LAUNCH_CHECK();
  hipLaunchKernelGGL(qr_finalize_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), (const float*)workspace, batch,
                     n_quantiles, out_loss_vec, out_loss);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------ K10
// out[0..2] = policy_loss, value_loss, approx_kl.  Gradients are of (policy_loss + value_loss);
// the three inputs are disjoint so they also serve the separate actor / critic backward passes.
__global__ void __launch_bounds__(1024)
ppo_loss_kernel(const float* __restrict__ lp, const float* __restrict__ ent, const float* __restrict__ v,
                const float* __restrict__ old_lp, const float* __restrict__ adv, const float* __restrict__ ret, int M,
                float clip, float entropy_weight, float* __restrict__ out, float* __restrict__ g_lp,
                float* __restrict__ g_ent, float* __restrict__ g_v) {
  __shared__ float s_red[16];
  float s_obj = 0.f, s_ent = 0.f, s_v = 0.f, s_kl = 0.f;
  const float inv_m = 1.f / (float)M;
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    const float ratio = expf(lp[i] - old_lp[i]);
    const float a = adv[i];
    const float obj = ratio * a;
    const float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    const float objc = rc * a;
    s_obj += fminf(obj, objc);
    s_ent += ent[i];
    const float dv = ret[i] - v[i];
    s_v += dv * dv;
    s_kl += old_lp[i] - lp[i];
    // torch.min splits ties evenly and clamp passes gradient inside [1-c, 1+c] (inclusive):
    // inside the range both branches carry ratio*adv; outside only `obj` does, when it is the min.
    const bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
    const float gate = inside ? 1.f : (obj < objc ? 1.f : (obj == objc ? 0.5f : 0.f));
    if (g_lp) g_lp[i] = -gate * obj * inv_m;
    if (g_ent) g_ent[i] = -entropy_weight * inv_m;
    if (g_v) g_v[i] = -dv * inv_m;
  }
  s_obj = block_sum(s_obj, s_red);
  s_ent = block_sum(s_ent, s_red);
  s_v = block_sum(s_v, s_red);
  s_kl = block_sum(s_kl, s_red);
  if (threadIdx.x == 0) {
    out[0] = -s_obj * inv_m - entropy_weight * (s_ent * inv_m);
    out[1] = 0.5f * (s_v * inv_m);
    out[2] = s_kl * inv_m;
  }
}

DRA_API int dra_ppo_loss(const float* log_pi_a, const float* entropy, const float* v, const float* old_log_pi_a,
                         const float* adv, const float* ret, int m, float ratio_clip, float entropy_weight, float* out3,
                         float* g_log_pi_a, float* g_entropy, float* g_v, void* stream) {
  if (!log_pi_a || !entropy || !v || !old_log_pi_a || !adv || !ret || !out3 || m < 1) return DRA_EINVAL;
  int threads = ((m + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), log_pi_a, entropy, v, old_log_pi_a,
                     adv, ret, m, ratio_clip, entropy_weight, out3, g_log_pi_a, g_entropy, g_v);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

__global__ void __launch_bounds__(1024)
a2c_loss_kernel(const float* __restrict__ lp, const float* __restrict__ ent, const float* __restrict__ v,
                const float* __restrict__ adv, const float* __restrict__ ret, int M, float entropy_weight,
                float value_weight, float* __restrict__ out, float* __restrict__ g_lp, float* __restrict__ g_ent,
                float* __restrict__ g_v) {
  __shared__ float s_red[16];
  float s_pg = 0.f, s_ent = 0.f, s_v = 0.f;
  const float inv_m = 1.f / (float)M;
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    s_pg += lp[i] * adv[i];
    s_ent += ent[i];
    const float dv = ret[i] - v[i];
    s_v += dv * dv;
    if (g_lp) g_lp[i] = -adv[i] * inv_m;
    if (g_ent) g_ent[i] = -entropy_weight * inv_m;
    if (g_v) g_v[i] = -value_weight * dv * inv_m;
  }
  s_pg = block_sum(s_pg, s_red);
  s_ent = block_sum(s_ent, s_red);
  s_v = block_sum(s_v, s_red);
  if (threadIdx.x == 0) {
    const float policy = -s_pg * inv_m, value = 0.5f * (s_v * inv_m), entropy = s_ent * inv_m;
    out[0] = policy - entropy_weight * entropy + value_weight * value;
    out[1] = policy; out[2] = value; out[3] = entropy;
  }
}

DRA_API int dra_a2c_loss(const float* log_pi_a, const float* entropy, const float* v, const float* adv, const float* ret,
                         int m, float entropy_weight, float value_loss_weight, float* out4, float* g_log_pi_a,
                         float* g_entropy, float* g_v, void* stream) {
  if (!log_pi_a || !entropy || !v || !adv || !ret || !out4 || m < 1) return DRA_EINVAL;
  int threads = ((m + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(a2c_loss_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), log_pi_a, entropy, v, adv, ret, m,
                     entropy_weight, value_loss_weight, out4, g_log_pi_a, g_entropy, g_v);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------
// K12: categorical policy head of the actor-critic nets (network_heads.py:240-255: Categorical(logits), sample,
// log_prob, entropy) as ONE kernel per direction instead of ~16 / ~10 elementwise, reduction and indexing kernels --
// at Atari action counts (4..18) each of those is a launch for a few hundred bytes, and a rollout forward is
// latency-bound on exactly these launches (rocprofv3: 60 % of the kernels of an A2C step, profiles/r02z9_*).
//   logp[a] = x[a] - m - log sum exp(x - m)          (logits normalised as torch.distributions.Categorical does)
//   action  = given, or sampled by inverse CDF from the uniform u[b] (first a with cumsum(p)[a] > u; the last action
//             absorbs rounding), log_pi_a = logp[action], entropy = -sum p logp
//   backward: dlogits[a] = g_lp (1[a == action] - p[a]) - g_ent p[a] (logp[a] + entropy)
// One thread per sample (A <= 64).
__global__ void __launch_bounds__(256)
categorical_fwd_kernel(const float* __restrict__ logits, int B, int A, const int64_t* __restrict__ action_in,
                       const float* __restrict__ u, int64_t* __restrict__ action_out, float* __restrict__ log_pi_a,
                       float* __restrict__ entropy) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t act;
  float lp, ent;
  categorical_row(logits + (int64_t)b * A, A, action_in != nullptr, action_in ? action_in[b] : 0, u ? u[b] : 0.f, &act, &lp, &ent);
  if (action_out) action_out[b] = act;
  log_pi_a[b] = lp;
  entropy[b] = ent;
}

__global__ void __launch_bounds__(256)
categorical_bwd_kernel(const float* __restrict__ logits, int B, int A, const int64_t* __restrict__ action,
                       const float* __restrict__ g_lp, const float* __restrict__ g_ent, float* __restrict__ dlogits) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* x = logits + (int64_t)b * A;
  float lse, ent;
  categorical_row_stats(x, A, &lse, &ent);
  const float gl = g_lp ? g_lp[b] : 0.f, ge = g_ent ? g_ent[b] : 0.f;
  const int64_t act = action[b];
  for (int a = 0; a < A; ++a) dlogits[(int64_t)b * A + a] = categorical_dlogit(x[a], lse, ent, a == act, gl, ge);
}

DRA_API int dra_categorical_fwd(const float* logits, int batch, int n_actions, const int64_t* action_in, const float* uniform,
                                int64_t* action_out, float* log_pi_a, float* entropy, void* stream) {
  if (!logits || !log_pi_a || !entropy || batch < 1 || n_actions < 1 || n_actions > 64) return DRA_EINVAL;
  if (!action_in && (!uniform || !action_out)) return DRA_EINVAL;
  hipLaunchKernelGGL(categorical_fwd_kernel, dim3((batch + 255) / 256), dim3(256), 0, dra_stream(stream), logits, batch, n_actions,
                     action_in, uniform, action_out, log_pi_a, entropy);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_categorical_bwd(const float* logits, int batch, int n_actions, const int64_t* action, const float* g_log_pi_a,
                                const float* g_entropy, float* out_dlogits, void* stream) {
  if (!logits || !action || !out_dlogits || batch < 1 || n_actions < 1 || n_actions > 64) return DRA_EINVAL;
  hipLaunchKernelGGL(categorical_bwd_kernel, dim3((batch + 255) / 256), dim3(256), 0, dra_stream(stream), logits, batch, n_actions,
                     action, g_log_pi_a, g_entropy, out_dlogits);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------
// PER helpers for the distributional agents (DQN_agent.py:121-127 applied to a KL / QR vector):
// priorities from a loss vector, importance weights from the sampling probabilities, and the
// weighted mean used by reduce_loss.
__global__ void __launch_bounds__(1024)
per_kernel(const float* __restrict__ loss_vec, const float* __restrict__ samp_prob, int B, float beta, float eps,
           float alpha, float* __restrict__ out_prio, float* __restrict__ out_w) {
  __shared__ float s_red[16];
  const int b = threadIdx.x;
  const bool on = b < B;
  if (on && loss_vec && out_prio) {
    const float ad = fabsf(loss_vec[b]) + eps;
    out_prio[b] = (alpha == 0.5f) ? sqrtf(ad) : powf(ad, alpha);
  }
  if (samp_prob && out_w) {
    const float wraw = on ? powf(samp_prob[b] * (float)B + 1e-6f, -(beta < 0.f ? samp_prob[B] : beta)) : -INFINITY;
    const float wmax = block_max(wraw, s_red);
    if (on) out_w[b] = wraw / wmax;
  }
}

DRA_API int dra_per_weights(const float* loss_vec, const float* sampling_prob, int batch, float beta, float replay_eps,
                            float replay_alpha, float* out_prio, float* out_weights, void* stream) {
  if (batch < 1 || batch > 1024) return DRA_EINVAL;
  const int threads = ((batch + 63) / 64) * 64;
  hipLaunchKernelGGL(per_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), loss_vec, sampling_prob, batch, beta,
                     replay_eps, replay_alpha, out_prio, out_weights);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

__global__ void __launch_bounds__(1024)
weighted_mean_kernel(const float* __restrict__ x, const float* __restrict__ w, int n, float* __restrict__ out) {
  __shared__ float s_red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += w ? x[i] * w[i] : x[i];
  s = block_sum(s, s_red);
  if (threadIdx.x == 0) *out = s / (float)n;
}

DRA_API int dra_weighted_mean(const float* x, const float* w, int n, float* out, void* stream) {
  if (!x || !out || n < 1) return DRA_EINVAL;
  int threads = ((n + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  hipLaunchKernelGGL(weighted_mean_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), x, w, n, out);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ---- rank-invariant categorical sampling (data-parallel A2C / PPO, dist.py) -----------------------------------------------
// action[i] = argmax_a (logits[i][a] + Gumbel noise), the noise a counter hash of (seed, rollout step, GLOBAL environment
// index lo + i, a): G ranks x N/G environments draw exactly what 1 rank x N would (Categorical(logits).sample() of
// network_heads.py:249-252 in distribution).  The rollout step lives in DEVICE memory and is advanced by the kernel itself,
// so the launch has constant arguments and replays from a captured rollout graph (round 2 reseeded a host generator per
// step, which kept every data-parallel rollout on the eager path).  One workgroup; rows strided over its threads.
__device__ __forceinline__ uint64_t gs_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256)
gumbel_sample_kernel(const float* __restrict__ logits, int n_local, int n_actions, uint64_t seed, int64_t* __restrict__ step_dev,
                     int64_t lo, int64_t* __restrict__ action_out) {
  const int64_t step = *step_dev;
  __syncthreads();                    // every thread has read the step before thread 0 advances it
  const uint64_t base = gs_mix64(seed * 0x9E3779B97F4A7C15ull + (uint64_t)step);
  for (int i = threadIdx.x; i < n_local; i += blockDim.x) {
    float best = -INFINITY;
    int arg = 0;
    for (int a = 0; a < n_actions; ++a) {
      const uint64_t h = gs_mix64(base + (uint64_t)(lo + i) * 64ull + (uint64_t)a);
      // 23 bits: k + 0.5 is exact in fp32 for every k < 2^23, so u lies STRICTLY inside (0, 1) (with 24 bits the largest k
      // rounded up to 2^24 and u == 1 made -log(-log u) = +inf: that action won whatever the logits, 2^-24 per draw)
      const float u = ((float)(h >> 41) + 0.5f) * (1.0f / 8388608.0f);
      const float v = logits[(int64_t)i * n_actions + a] - logf(-logf(u));
      if (v > best) { best = v; arg = a; }
    }
    action_out[i] = arg;
  }
  if (threadIdx.x == 0) *step_dev = step + 1;
}

DRA_API int dra_gumbel_sample(const float* logits, int n_local, int n_actions, uint64_t seed, int64_t* step_dev, int64_t lo,
                              int64_t* action_out, void* stream) {
  if (!logits || !step_dev || !action_out || n_local < 1 || n_actions < 1 || n_actions > 64 || lo < 0) return DRA_EINVAL;
  hipLaunchKernelGGL(gumbel_sample_kernel, dim3(1), dim3(256), 0, dra_stream(stream), logits, n_local, n_actions, seed, step_dev, lo,
                     action_out);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
