// ppo_mlp: BASELINE configs[2] -- PPO over GaussianActorCriticNet with two small tanh MLPs (examples.py:497-523: HalfCheetah
// shapes 17 -> 64 -> 64 -> {6, 1}, 16 workers, rollout 2048, 10 epochs x 512 minibatches of 64, two Adam optimisers) -- as
// persistent kernels.  Replaces, per rollout, 2049 host round trips of PPO_agent.py:32-49 and the 5120 x (forward + loss +
// backward + 2 optimizer steps) launch chains of PPO_agent.py:71-99 with four launches:
//
//   ppo_mlp_rollout_kernel   one workgroup walks the whole rollout: observation statistics (fp64 RunningMeanStd), both forwards,
//                            action = mean + softplus(std) * noise, log-probability, environment step (csrc/cont_env.h)
//   (dra_gae, dra_adv_normalize: scan.hip / losses.hip, unchanged)
//   ppo_pack_kernel          the rows of every minibatch of every epoch gathered once (PPO_agent.py:72-76)
//   ppo_mlp_update_kernel    TWO workgroups -- the actor's and the critic's networks never meet when phi_body is the identity
//                            (network_heads.py:181-183) -- each keeping its weights and Adam moments in registers (one MFMA
//                            operand layout serves the forward AND receives the weight gradient, see below) from the first
//                            minibatch to the last; the approx-KL gate (PPO_agent.py:88) is evaluated on the device.
//
// The work is a chain of 5120 dependent 64-row updates of an 5.7 k-parameter network: latency-bound by construction (SURVEY.md
// 8d), nothing to spread over 256 CUs.  What can be removed is everything between the dependent steps: launches, HBM round
// trips of weights / activations / optimizer state, host decisions.  Contractions run on v_mfma_f32_16x16x4_f32 (exact fp32).
//
// Operand layout (lane l of a wave: c16 = l & 15, g = l >> 4).  For D = A x B the instruction takes A[i = c16][k = g],
// B[k = g][j = c16] and returns D[i = 4g + reg][j = c16].  The K index of a step may be ANY 4 values as long as A and B agree,
// so a hidden layer y[row][n] = sum_k x[row][k] W[n][k] walks k as 16 tk + 4 g + r (step (tk, r)): a lane's B operands
// W[16 nt + c16][16 tk + 4 g + r] are then exactly the elements the weight-gradient contraction dW^T[k][n] = sum_row x[row][k]
// dz[row][n] leaves in that lane's accumulators (D row 4 g + r of tile tk, column c16).  Wave nt therefore OWNS units
// [16 nt, 16 nt + 16): it alone reads those weights in the forward, receives their gradient and applies Adam in registers;
// only the transposed use (dh = dz W) needs a copy in LDS.
#include "common.h"
#include "cont_env.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kRows = 64;              // rows a workgroup holds at once: 4 M tiles
constexpr int kMaxS = 64, kMaxA = 16;
constexpr int kLdX = kMaxS + 4;        // allocation stride of the observation tile (the used stride is 16 KT1 + 4)
constexpr int kLd3 = 20;               // stride of the [rows][16] head-side arrays
constexpr int kAuxLp = 16, kAuxAdv = 17, kAuxRet = 18;
constexpr float kLogSqrt2Pi = 0.91893853320467274178f;
constexpr float kEntConst = 1.4189385332046727f;   // 0.5 + 0.5 log(2 pi)   (torch.distributions.Normal.entropy)

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the transcendental unit (v_exp_f32 / v_rcp_f32, 1 ulp each): absolute error < 2e-7 on
// the whole line, saturating to +-1 exactly.  36 evaluations per lane and minibatch: the library routine would cost more
// than the layer's MFMAs.
__device__ __forceinline__ float fast_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
// F.softplus (beta 1, threshold 20) and its derivative as autograd forms it (z / (z + 1), z = exp(x))
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float softplus_grad(float x) {
  if (x > 20.f) return 1.f;
  const float z = expf(x);
  return z / (z + 1.f);
}
// batch row of contraction step s for lane group g in the weight-gradient contractions: a bijection (s < 4 MT, g < 4) -> rows
// whose four rows per step lie 4 apart, so the 4 x 16-float reads of one ds_read_b32 fall on 4 distinct bank groups
__device__ __forceinline__ int row_of(int s, int g) { return ((s >> 2) << 4) + (g << 2) + (s & 3); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {     // lane permutation inside a row of 16 lanes, on the VALU (no LDS crossbar)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group16_sum(float v) {   // over the 16 lanes that share g; every lane receives the sum
  v += dpp_mov<0xB1>(v);     // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);    // row_half_mirror: quad 0 <-> quad 1, quad 2 <-> quad 3
  v += dpp_mov<0x140>(v);    // row_mirror: lower half <-> upper half
  return v;
}
__device__ __forceinline__ float over_g_sum(float v) {    // over the 4 lanes that share c16
  v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}

struct AdamScalars { float step_size, inv_sqrt_bc2, beta1, beta2, omb1, omb2, eps; };
__device__ __forceinline__ void adam_elem(float& p, float gk, float& m, float& v, const AdamScalars& a) {
  // optim.hip adam_step_kernel's element formula (torch.optim.Adam, no amsgrad / weight decay), with the square root and the
  // division on the transcendental unit (v_sqrt_f32 / v_rcp_f32, 1 ulp each: a relative 2e-7 on a step of lr x O(1)) -- the
  // IEEE sequences cost ~20 instructions per parameter, 37 parameters per lane and minibatch (profiles/r05c_prof_ppo_mlp.json)
  m = m * a.beta1 + a.omb1 * gk;
  v = v * a.beta2 + a.omb2 * gk * gk;
  p = p - (a.step_size * m) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * a.inv_sqrt_bc2 + a.eps);
}

constexpr size_t update_lds_floats(int H, int S) {
  // 2 images | h1, h2 [row][unit] | h1T, h2T, dz2T [unit][row] | W2T | W3 | dz3 | dz3T | b3, std, reduction slots
  return 2 * (size_t)kRows * (16 * ((S + 15) / 16) + 4 + kLd3) + 2 * (size_t)kRows * (H + 4) + 3 * (size_t)H * (kRows + 4) +
         (size_t)H * (H + 4) + 16 * (size_t)(H + 4) + (size_t)kRows * kLd3 + 16 * (size_t)(kRows + 4) + 16 + 16 + 32 + 256;
}
constexpr size_t kLdsFloatsMax = 160 * 1024 / 4;

// debug dump layout (floats, per role; the critic's region starts at kDbgRole)
constexpr int kDbgRole = 32768, kDbgH1 = 0, kDbgH2 = 4096, kDbgHead = 8192, kDbgLp = 9216, kDbgGl = 9280, kDbgScal = 9344,
              kDbgDz3 = 10240, kDbgDz2 = 11264, kDbgDz1 = 15360, kDbgW1 = 19456, kDbgW2 = 23552, kDbgW3 = 27648,
              kDbgB1 = 28672, kDbgB2 = 28736, kDbgB3 = 28800, kDbgStd = 28816;
static_assert(2 * kDbgRole <= DRA_PPO_MLP_DBG_FLOATS, "debug buffer");

// ------------------------------------------------------------------------------------------------ update
// One workgroup = one network.  Per minibatch (phases separated by workgroup barriers; "own": wave w < H / 16 owns hidden units
// [16 w, 16 w + 16) of both hidden layers -- their weights, the Adam moments and the matching slice of the head's weights):
//   F1   h1 = tanh(x W1^T + b1)                       tile by tile, a tile's tanh / LDS stores under the next tile's MFMAs
//   F2   h2 = tanh(h1 W2^T + b2)
//   F3   head + loss: wave mt takes rows [16 mt, 16 mt + 16); policy mean / value, log-probability, clipped-ratio terms, dz3
//   gate approx-KL against the limit (PPO_agent.py:88); the next minibatch's image is committed to the OTHER LDS buffer
//   B3   dz2 = (dz3 W3)(1 - h2^2);  dW3^T, db3, dstd
//   B2   dW2^T with Adam(W3, b3, std) in the MFMA shadow; dz1 = (dz2 W2)(1 - h1^2) with Adam(W2, b2) in the MFMA shadow
//   B1   dW1^T; Adam(W1, b1)
// The time of a minibatch is the sum of dependent latencies, not of work: what matters is how little sits between the MFMAs
// (profiles/r05*_prof_ppo_mlp.json: cycles per phase).
template <int H, bool ACTOR, int MODE, int SC>
__device__ __forceinline__ void ppo_update_role(const dra_ppo_mlp_cfg& cfg, const dra_ppo_mlp_net& net, const float* __restrict__ packed,
                                                const int n, const int epochs, float* __restrict__ out3,
                                                int64_t* __restrict__ out_counts, float* __restrict__ dbg_all, float* lds) {
  constexpr int NT = H / 16, LD = H + 4;
  constexpr int KTC = (SC + 15) / 16;           // SC: state_dim at compile time (0: any, read from the configuration)
  constexpr int KTM = SC ? KTC : 4;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c16 = l & 15, g = l >> 4;
  const int S = SC ? SC : cfg.state_dim, A = ACTOR ? cfg.action_dim : 1, MB = cfg.mini_batch;
  const int KT1 = SC ? KTC : (S + 15) >> 4;
  const int LDX = 16 * KT1 + 4;
  const int MT = (MB + 15) >> 4;      // M tiles that can hold rows (the contractions over rows stop there)
  const int per_epoch = (n + MB - 1) / MB, total = per_epoch * epochs;
  const bool own = w < NT;
  const int ncol = 16 * w + c16;
  constexpr bool DUMP = MODE == 1, PROF = MODE == 2;   // 1: dump of minibatch 0 (tests); 2: cycle counts per phase (tools)
  float* dbg = (DUMP && dbg_all) ? dbg_all + (ACTOR ? 0 : kDbgRole) : nullptr;
  long long prof[16], prof_last = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) prof[i] = 0;
  auto stamp = [&](int k) {
    if (PROF && tid == 0) {
      const long long t = clock64();
      prof[k] += t - prof_last;
      prof_last = t;
    }
  };

  // LDS: two minibatch images ([kRows][LDX] observations | [kRows][kLd3] action, log_pi_a, advantage, ret -- the layout
  // ppo_pack_kernel writes), then activations / gradients row-major AND transposed (every MFMA operand is then one 16-byte read:
  // contractions over units read [row][unit], contractions over rows read [unit][row]) and the transposed-use weight copies
  constexpr int LDT = kRows + 4;
  const int img = kRows * (LDX + kLd3);
  float* sImg = lds;
  float* sH1 = lds + 2 * img;         // h1 [row][unit]
  float* sH1T = sH1 + kRows * LD;     // h1 [unit][row]
  float* sH2 = sH1T + H * LDT;        // h2 [row][unit], overwritten in place by dz2
  float* sH2T = sH2 + kRows * LD;     // h2 [unit][row], later dz1 [unit][row]
  float* sDZ2T = sH2T + H * LDT;      // dz2 [unit][row]
  float* sW2T = sDZ2T + H * LDT;      // W2 [in][out]
  float* sW3 = sW2T + H * LD;         // W3 [out (16)][in]
  float* sDZ3 = sW3 + 16 * LD;        // dz3 [row][out]
  float* sDZ3T = sDZ3 + kRows * kLd3; // dz3 [out][row]
  float* sB3 = sDZ3T + 16 * LDT;
  float* sStd = sB3 + 16;
  float* sRed = sStd + 16;            // [4 waves][4 g][2]
  float* sPart = sRed + 32;           // [4 waves][4 g][16]
  for (int i = tid; i < (int)update_lds_floats(H, S); i += 256) lds[i] = 0.f;
  __syncthreads();

  // ---- masters: parameters + Adam moments of this lane's share, in the forward's B-operand layout
  float w1p[KTM][4], w1m[KTM][4], w1v[KTM][4];
  float w2p[NT][4], w2m[NT][4], w2v[NT][4];
  float w3p[4], w3m[4], w3v[4];
  float b1p = 0.f, b1m = 0.f, b1v = 0.f, b2p = 0.f, b2m = 0.f, b2v = 0.f, b3p = 0.f, b3m = 0.f, b3v = 0.f;
  float sdp = 0.f, sdm = 0.f, sdv = 0.f;
#pragma unroll
  for (int tk = 0; tk < KTM; ++tk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * tk + 4 * g + r;
      const bool ok = own && k < S;
      const int idx = net.off_w1 + ncol * S + k;
      w1p[tk][r] = ok ? net.param[idx] : 0.f;
      w1m[tk][r] = ok ? net.exp_avg[idx] : 0.f;
      w1v[tk][r] = ok ? net.exp_avg_sq[idx] : 0.f;
    }
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = net.off_w2 + ncol * H + 16 * tk + 4 * g + r;
      w2p[tk][r] = own ? net.param[idx] : 0.f;
      w2m[tk][r] = own ? net.exp_avg[idx] : 0.f;
      w2v[tk][r] = own ? net.exp_avg_sq[idx] : 0.f;
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = own && c16 < A;
    const int idx = net.off_w3 + c16 * H + 16 * w + 4 * g + r;
    w3p[r] = ok ? net.param[idx] : 0.f;
    w3m[r] = ok ? net.exp_avg[idx] : 0.f;
    w3v[r] = ok ? net.exp_avg_sq[idx] : 0.f;
  }
  if (own) {
    b1p = net.param[net.off_b1 + ncol]; b1m = net.exp_avg[net.off_b1 + ncol]; b1v = net.exp_avg_sq[net.off_b1 + ncol];
    b2p = net.param[net.off_b2 + ncol]; b2m = net.exp_avg[net.off_b2 + ncol]; b2v = net.exp_avg_sq[net.off_b2 + ncol];
  }
  if (w == 0 && c16 < A) {
    b3p = net.param[net.off_b3 + c16]; b3m = net.exp_avg[net.off_b3 + c16]; b3v = net.exp_avg_sq[net.off_b3 + c16];
    if (ACTOR) { sdp = net.param[net.off_std + c16]; sdm = net.exp_avg[net.off_std + c16]; sdv = net.exp_avg_sq[net.off_std + c16]; }
  }
  // the LDS copies the transposed contractions (dh = dz W) and the head read
  auto publish_w2 = [&]() {      // W2 [in][out]: what dh1 = dz2 W2 reads (lane: out = 16 tn + 4 g + r contiguous, in = its unit)
    if (own) {
#pragma unroll
      for (int tk = 0; tk < NT; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r) sW2T[(16 * tk + 4 * g + r) * LD + ncol] = w2p[tk][r];
    }
  };
  auto publish_head = [&]() {
    if (own) {
      f32x4 v3 = {w3p[0], w3p[1], w3p[2], w3p[3]};
      *reinterpret_cast<f32x4*>(&sW3[c16 * LD + 16 * w + 4 * g]) = v3;
    }
    if (w == 0 && g == 0) { sB3[c16] = b3p; sStd[c16] = sdp; }
  };
  publish_w2();
  publish_head();

  // ---- Adam's bias corrections: beta^t as a running product from pow(beta, t0) (within 1e-12 of pow(beta, t))
  const int64_t steps0 = *net.step_dev;
  double pw1 = pow((double)net.beta1, (double)steps0), pw2 = pow((double)net.beta2, (double)steps0);
  AdamScalars ad;
  ad.beta1 = net.beta1; ad.beta2 = net.beta2; ad.omb1 = 1.f - net.beta1; ad.omb2 = 1.f - net.beta2; ad.eps = net.eps;
  ad.step_size = 0.f; ad.inv_sqrt_bc2 = 0.f;

  // ---- minibatch images: prefetched one minibatch ahead into registers, committed to the other LDS buffer after the gate
  constexpr int NJ = (kRows * ((SC ? 16 * KTC + 4 : kLdX) + kLd3) / 4 + 255) / 256;
  f32x4 pf[NJ];
  const int img4 = img / 4;
#pragma unroll
  for (int j = 0; j < NJ; ++j) pf[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int q) {
    const f32x4* src = reinterpret_cast<const f32x4*>(packed) + (int64_t)q * img4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (tid + 256 * j < img4) pf[j] = src[tid + 256 * j];
  };
  auto commit = [&](int buf) {
    f32x4* dst = reinterpret_cast<f32x4*>(sImg + buf * img);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (tid + 256 * j < img4) dst[tid + 256 * j] = pf[j];
  };
  if (total > 0) { issue(0); commit(0); }
  __syncthreads();

  int64_t applied = 0;
  float sd = 1.f, log_sd = 0.f, std_raw = 0.f, inv_sd = 1.f, inv_var = 1.f;
  bool std_dirty = true;
  for (int q = 0; q < total; ++q) {
    const int kq = q % per_epoch;
    const int rows = min(MB, n - kq * MB);
    const float inv_m = 1.f / (float)rows;
    const bool dump = DUMP && dbg && q == 0;
    const float* sX = sImg + (q & 1) * img;
    const float* sAux = sX + kRows * LDX;
    if (PROF && tid == 0 && q == 0) prof_last = clock64();
    if (q + 1 < total) issue(q + 1);
    stamp(0);

    // ---- F1: h1 = tanh(x W1^T + b1).  Tile by tile: the next tile's operands are requested before this tile's MFMAs (two
    // accumulator chains), the previous tile's tanh + stores ride in their shadow.
    if (own) {
      f32x4 avc[KTM], avn[KTM], prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < KTM; ++tk) {
        avc[tk] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tk < KT1) avc[tk] = *reinterpret_cast<const f32x4*>(&sX[c16 * LDX + 16 * tk + 4 * g]);
      }
#pragma unroll
      for (int mt = 0; mt <= 4; ++mt) {
        if (mt < 3) {
#pragma unroll
          for (int tk = 0; tk < KTM; ++tk)
            if (tk < KT1) avn[tk] = *reinterpret_cast<const f32x4*>(&sX[(16 * (mt + 1) + c16) * LDX + 16 * tk + 4 * g]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        if (mt < 4) {
#pragma unroll
          for (int tk = 0; tk < KTM; ++tk)
            if (tk < KT1) {
              // (a step whose four k = 16 tk + 4 g + r all lie past the observation is all zeros: skipped when the observation
              // size is a compile-time constant -- a run-time test per MFMA costs more than the MFMA)
              if (!SC || 16 * tk + 0 < SC) a0 = MFMA16(avc[tk][0], w1p[tk][0], a0);
              if (!SC || 16 * tk + 1 < SC) a1 = MFMA16(avc[tk][1], w1p[tk][1], a1);
              if (!SC || 16 * tk + 2 < SC) a0 = MFMA16(avc[tk][2], w1p[tk][2], a0);
              if (!SC || 16 * tk + 3 < SC) a1 = MFMA16(avc[tk][3], w1p[tk][3], a1);
            }
        }
        if (mt > 0) {
          f32x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * (mt - 1) + 4 * g + r;
            h[r] = fast_tanh(prev[r] + b1p);
            sH1[row * LD + ncol] = h[r];
            if (dump) dbg[kDbgH1 + row * 64 + ncol] = h[r];
          }
          *reinterpret_cast<f32x4*>(&sH1T[ncol * LDT + 16 * (mt - 1) + 4 * g]) = h;
        }
        prev = a0 + a1;
#pragma unroll
        for (int tk = 0; tk < KTM; ++tk) avc[tk] = avn[tk];
      }
    }
    stamp(1);
    __syncthreads();
    stamp(2);
    // ---- F2: h2 = tanh(h1 W2^T + b2)
    if (own) {
      f32x4 avc[NT], avn[NT], prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < NT; ++tk) avc[tk] = *reinterpret_cast<const f32x4*>(&sH1[c16 * LD + 16 * tk + 4 * g]);
#pragma unroll
      for (int mt = 0; mt <= 4; ++mt) {
        if (mt < 3) {
#pragma unroll
          for (int tk = 0; tk < NT; ++tk)
            avn[tk] = *reinterpret_cast<const f32x4*>(&sH1[(16 * (mt + 1) + c16) * LD + 16 * tk + 4 * g]);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        if (mt < 4) {
#pragma unroll
          for (int tk = 0; tk < NT; ++tk) {
            a0 = MFMA16(avc[tk][0], w2p[tk][0], a0);
            a1 = MFMA16(avc[tk][1], w2p[tk][1], a1);
            a0 = MFMA16(avc[tk][2], w2p[tk][2], a0);
            a1 = MFMA16(avc[tk][3], w2p[tk][3], a1);
          }
        }
        if (mt > 0) {
          f32x4 h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * (mt - 1) + 4 * g + r;
            h[r] = fast_tanh(prev[r] + b2p);
            sH2[row * LD + ncol] = h[r];
            if (dump) dbg[kDbgH2 + row * 64 + ncol] = h[r];
          }
          *reinterpret_cast<f32x4*>(&sH2T[ncol * LDT + 16 * (mt - 1) + 4 * g]) = h;
        }
        prev = a0 + a1;
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) avc[tk] = avn[tk];
      }
    }
    stamp(3);
    __syncthreads();
    stamp(4);
    // ---- F3 + loss: lane (c16 = head output, g, reg) <-> row 16 w + 4 g + reg
    {
      f32x4 av[NT], bv[NT];
#pragma unroll
      for (int tk = 0; tk < NT; ++tk) {
        av[tk] = *reinterpret_cast<const f32x4*>(&sH2[(16 * w + c16) * LD + 16 * tk + 4 * g]);
        bv[tk] = *reinterpret_cast<const f32x4*>(&sW3[c16 * LD + 16 * tk + 4 * g]);
      }
      f32x4 aux;      // this lane's four rows of the image: its own action column, and the row scalars
      float lp_old[4], adv4[4], ret4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * g + r;
        aux[r] = sAux[row * kLd3 + c16];
        if (ACTOR) { lp_old[r] = sAux[row * kLd3 + kAuxLp]; adv4[r] = sAux[row * kLd3 + kAuxAdv]; }
        else ret4[r] = sAux[row * kLd3 + kAuxRet];
      }
      const float b3 = sB3[c16];
      if (ACTOR && std_dirty) {      // scale = softplus(std) and what the row loop needs of it, once per actor step
        std_raw = sStd[c16];
        sd = std_raw > 20.f ? std_raw : fast_log(1.f + fast_exp(std_raw));
        log_sd = fast_log(sd);
        inv_sd = __builtin_amdgcn_rcpf(sd);
        inv_var = inv_sd * inv_sd;
        std_dirty = false;
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};     // two chains: a dependent MFMA waits 40 cycles
#pragma unroll
      for (int tk = 0; tk < NT; ++tk) {
        acc = MFMA16(av[tk][0], bv[tk][0], acc);
        acc_b = MFMA16(av[tk][1], bv[tk][1], acc_b);
        acc = MFMA16(av[tk][2], bv[tk][2], acc);
        acc_b = MFMA16(av[tk][3], bv[tk][3], acc_b);
      }
      acc += acc_b;
      const bool col_ok = c16 < A;
      float s0 = 0.f, s1 = 0.f, gsd = 0.f;
      f32x4 dzv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * g + r;
        const bool row_ok = row < rows;
        float dz = 0.f;
        if (ACTOR) {
          const float mean = fast_tanh(acc[r] + b3);
          const float diff = aux[r] - mean;
          const float lpe = col_ok ? (-(diff * diff) * (0.5f * inv_var) - log_sd - kLogSqrt2Pi) : 0.f;
          const float lp = group16_sum(lpe);
          // losses.hip ppo_loss_kernel's arithmetic (PPO_agent.py:78-86), row by row
          const float ratio = fast_exp(lp - lp_old[r]);
          const float obj = ratio * adv4[r];
          const float rc = fminf(fmaxf(ratio, 1.f - cfg.ratio_clip), 1.f + cfg.ratio_clip);
          const float objc = rc * adv4[r];
          const bool inside = (ratio >= 1.f - cfg.ratio_clip) && (ratio <= 1.f + cfg.ratio_clip);
          const float gate = inside ? 1.f : (obj < objc ? 1.f : (obj == objc ? 0.5f : 0.f));
          const float g_lp = row_ok ? -gate * obj * inv_m : 0.f;
          if (row_ok) { s0 += fminf(obj, objc); s1 += lp_old[r] - lp; }
          if (col_ok) {
            dz = (g_lp * (diff * inv_var)) * (1.f - mean * mean);
            gsd += g_lp * ((diff * diff) * (inv_var * inv_sd) - inv_sd);
          }
          if (dump) {
            dbg[kDbgHead + row * 16 + c16] = mean;
            if (c16 == 0) { dbg[kDbgLp + row] = lp; dbg[kDbgGl + row] = g_lp; }
          }
        } else {
          const float v = acc[r] + b3;
          const float dv = ret4[r] - v;
          if (row_ok && c16 == 0) { s0 += dv * dv; dz = -dv * inv_m; }
          if (dump) {
            dbg[kDbgHead + row * 16 + c16] = v;
            if (c16 == 0) { dbg[kDbgLp + row] = v; dbg[kDbgGl + row] = dz; }
          }
        }
        dzv[r] = dz;
        sDZ3[row * kLd3 + c16] = dz;
        if (dump) dbg[kDbgDz3 + row * 16 + c16] = dz;
      }
      *reinterpret_cast<f32x4*>(&sDZ3T[c16 * LDT + 16 * w + 4 * g]) = dzv;
      // per (wave, lane group) partial sums: reduced after the barrier by whoever needs them, in a fixed order
      if (c16 == 0) { sRed[2 * (4 * w + g)] = s0; sRed[2 * (4 * w + g) + 1] = s1; }
      if (ACTOR) sPart[16 * (4 * w + g) + c16] = gsd;
    }
    stamp(5);
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { t0 += sRed[2 * i]; t1 += sRed[2 * i + 1]; }
    bool open = true;
    if (ACTOR) {
      const float kl = t1 * inv_m;
      open = (double)kl <= cfg.kl_limit;
      if (tid == 0 && (q == total - 1 || dump)) {
        float ent = 0.f;
        for (int a = 0; a < A; ++a) {
          const float x = sStd[a];
          ent += kEntConst + fast_log(x > 20.f ? x : fast_log(1.f + fast_exp(x)));
        }
        const float pl = -t0 * inv_m - cfg.entropy_weight * ent;
        if (q == total - 1) { out3[0] = pl; out3[2] = kl; }
        if (dump) { dbg[kDbgScal] = pl; dbg[kDbgScal + 2] = kl; dbg[kDbgScal + 3] = open ? 1.f : 0.f; dbg[kDbgScal + 4] = ent; }
      }
    } else if (tid == 0 && (q == total - 1 || dump)) {
      const float vl = 0.5f * (t0 * inv_m);
      if (q == total - 1) out3[1] = vl;
      if (dump) dbg[kDbgScal + 1] = vl;
    }
    if (q + 1 < total) commit((q + 1) & 1);      // (that buffer's last readers finished before the previous iteration's last barrier)
    stamp(6);

    if (open) {
      float gw3[4] = {0.f, 0.f, 0.f, 0.f}, gb3 = 0.f, gstd = 0.f;
      // ---- B3: dz2 = (dz3 W3) (1 - h2^2);  dW3^T tile tk = w;  db3, dstd
      if (own) {
        // operands: dh2 = dz3 W3 contracts the (<= 16) head outputs in steps of four; dW3^T contracts the rows: both operands of
        // that one come [unit][row] / [out][row] (and h2 [unit][row] is also what the dz2 epilogue multiplies by)
        float b3w[4];
        float a3d[4][4];
        f32x4 hT[4], dT[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          b3w[s2] = 0.f;
          if (4 * s2 < A) {
            b3w[s2] = sW3[(4 * s2 + g) * LD + ncol];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a3d[s2][mt] = sDZ3[(16 * mt + c16) * kLd3 + 4 * s2 + g];
          }
        }
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          hT[sg] = *reinterpret_cast<const f32x4*>(&sH2T[ncol * LDT + 16 * sg + 4 * g]);
          dT[sg] = *reinterpret_cast<const f32x4*>(&sDZ3T[c16 * LDT + 16 * sg + 4 * g]);
        }
        pw1 *= (double)net.beta1;
        pw2 *= (double)net.beta2;
        ad.step_size = (float)((double)net.lr / (1.0 - pw1));
        ad.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pw2));
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
          if (4 * s2 < A) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(a3d[s2][mt], b3w[s2], acc[mt]);
          }
        // dW3^T (two chains over the rows) with the dz2 epilogue of the four dh2 tiles in its shadow
        f32x4 a3 = {0.f, 0.f, 0.f, 0.f}, a3b = {0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          if (sg < MT) {
            bsum += (dT[sg][0] + dT[sg][1]) + (dT[sg][2] + dT[sg][3]);
            a3 = MFMA16(hT[sg][0], dT[sg][0], a3);
            a3b = MFMA16(hT[sg][1], dT[sg][1], a3b);
            a3 = MFMA16(hT[sg][2], dT[sg][2], a3);
            a3b = MFMA16(hT[sg][3], dT[sg][3], a3b);
          }
          {
            const int mt = sg;
            f32x4 dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * mt + 4 * g + r;
              dz[r] = acc[mt][r] * (1.f - hT[mt][r] * hT[mt][r]);
              sH2[row * LD + ncol] = dz[r];       // in place: this lane read h2 [row][its unit] from the transposed copy
              if (dump) dbg[kDbgDz2 + row * 64 + ncol] = dz[r];
            }
            *reinterpret_cast<f32x4*>(&sDZ2T[ncol * LDT + 16 * mt + 4 * g]) = dz;
          }
        }
        a3 += a3b;
#pragma unroll
        for (int r = 0; r < 4; ++r) gw3[r] = a3[r];
        gb3 = over_g_sum(bsum);
        if (ACTOR && w == 0) {
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) t += sPart[16 * i + c16];
          // d(-entropy_weight * mean(entropy)) / d scale = -entropy_weight / scale;  d softplus = z / (z + 1), z = exp(std)
          const float z = fast_exp(std_raw);
          const float sg = std_raw > 20.f ? 1.f : z * __builtin_amdgcn_rcpf(z + 1.f);
          gstd = (t - cfg.entropy_weight * inv_sd) * sg;
        }
        if (dump) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dbg[kDbgW3 + c16 * 64 + 16 * w + 4 * g + r] = gw3[r];
          if (w == 0 && g == 0) { dbg[kDbgB3 + c16] = gb3; dbg[kDbgStd + c16] = gstd; }
        }
      }
      stamp(7);
      __syncthreads();
      stamp(8);
      // ---- B2: dW2^T tiles (this wave's units x all inputs) with Adam(W3, b3, std) in the MFMA shadow;
      //          dz1 = (dz2 W2) (1 - h1^2), kept [unit][row], with Adam(W2, b2) in the MFMA shadow
      if (own) {
        f32x4 bw[4], awc[NT], awn[NT], bd[NT];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) bw[sg] = *reinterpret_cast<const f32x4*>(&sDZ2T[ncol * LDT + 16 * sg + 4 * g]);
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) awc[tk] = *reinterpret_cast<const f32x4*>(&sH1T[(16 * tk + c16) * LDT + 4 * g]);
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) bd[tn] = *reinterpret_cast<const f32x4*>(&sW2T[ncol * LD + 16 * tn + 4 * g]);
        f32x4 acc[NT];
#pragma unroll
        for (int tk = 0; tk < NT; ++tk) acc[tk] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          if (sg < 3) {
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
              awn[tk] = *reinterpret_cast<const f32x4*>(&sH1T[(16 * tk + c16) * LDT + 16 * (sg + 1) + 4 * g]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (sg < MT) {
            bsum += (bw[sg][0] + bw[sg][1]) + (bw[sg][2] + bw[sg][3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int tk = 0; tk < NT; ++tk) acc[tk] = MFMA16(awc[tk][j], bw[sg][j], acc[tk]);
          }
          adam_elem(w3p[sg], gw3[sg], w3m[sg], w3v[sg], ad);
          if (sg == 3 && w == 0 && c16 < A) {
            adam_elem(b3p, gb3, b3m, b3v, ad);
            if (ACTOR) adam_elem(sdp, gstd, sdm, sdv, ad);
          }
#pragma unroll
          for (int tk = 0; tk < NT; ++tk) awc[tk] = awn[tk];
        }
        const float gb2 = over_g_sum(bsum);
        publish_head();           // (sW3 / sB3 / sStd were last read before the B3 | B2 barrier)
        if (dump) {
#pragma unroll
          for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) dbg[kDbgW2 + ncol * 64 + 16 * tk + 4 * g + r] = acc[tk][r];
          if (g == 0) dbg[kDbgB2 + ncol] = gb2;
        }
        // dh1 tile by tile: the next tile's operands requested first, the previous tile's epilogue and one W2 Adam element per
        // k step in the MFMA shadow
        f32x4 adc[NT], adn[NT], h1c, h1n, prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) adc[tn] = *reinterpret_cast<const f32x4*>(&sH2[c16 * LD + 16 * tn + 4 * g]);
        h1c = *reinterpret_cast<const f32x4*>(&sH1T[ncol * LDT + 4 * g]);
        h1n = h1c;
#pragma unroll
        for (int mt = 0; mt <= 4; ++mt) {
          if (mt < 3) {
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
              adn[tn] = *reinterpret_cast<const f32x4*>(&sH2[(16 * (mt + 1) + c16) * LD + 16 * tn + 4 * g]);
          }
          if (mt >= 1 && mt < 4) h1n = *reinterpret_cast<const f32x4*>(&sH1T[ncol * LDT + 16 * mt + 4 * g]);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
          if (mt < 4) {
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
              a0 = MFMA16(adc[tn][0], bd[tn][0], a0);
              a1 = MFMA16(adc[tn][1], bd[tn][1], a1);
              a0 = MFMA16(adc[tn][2], bd[tn][2], a0);
              a1 = MFMA16(adc[tn][3], bd[tn][3], a1);
              // Adam on the master registers (the forward's operands); the transposed reads went to the LDS copy, which keeps the
              // OLD weights until publish_w2() in B1
              adam_elem(w2p[tn][mt], acc[tn][mt], w2m[tn][mt], w2v[tn][mt], ad);
            }
          }
          if (mt > 0) {
            f32x4 dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dz[r] = prev[r] * (1.f - h1c[r] * h1c[r]);
              if (dump) dbg[kDbgDz1 + (16 * (mt - 1) + 4 * g + r) * 64 + ncol] = dz[r];
            }
            *reinterpret_cast<f32x4*>(&sH2T[ncol * LDT + 16 * (mt - 1) + 4 * g]) = dz;
          }
          prev = a0 + a1;
          if (mt >= 1) h1c = h1n;
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) adc[tn] = adn[tn];
        }
        adam_elem(b2p, gb2, b2m, b2v, ad);
      }
      stamp(9);
      __syncthreads();
      stamp(10);
      // ---- B1: dW1^T tiles;  Adam(W1, b1)
      if (own) {
        publish_w2();             // (the LDS copy was last read before the B2 | B1 barrier)
        f32x4 bx[4];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) bx[sg] = *reinterpret_cast<const f32x4*>(&sH2T[ncol * LDT + 16 * sg + 4 * g]);
        float axc[4][KTM], axn[4][KTM];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tk = 0; tk < KTM; ++tk) {
            axc[j][tk] = 0.f;
            if (tk < KT1) axc[j][tk] = sX[(4 * g + j) * LDX + 16 * tk + c16];
          }
        f32x4 acc[KTM];
#pragma unroll
        for (int tk = 0; tk < KTM; ++tk) acc[tk] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          if (sg < 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int tk = 0; tk < KTM; ++tk)
                if (tk < KT1) axn[j][tk] = sX[(16 * (sg + 1) + 4 * g + j) * LDX + 16 * tk + c16];
          }
          __builtin_amdgcn_sched_barrier(0);
          if (sg < MT) {
            bsum += (bx[sg][0] + bx[sg][1]) + (bx[sg][2] + bx[sg][3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int tk = 0; tk < KTM; ++tk)
                if (tk < KT1) acc[tk] = MFMA16(axc[j][tk], bx[sg][j], acc[tk]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tk = 0; tk < KTM; ++tk) axc[j][tk] = axn[j][tk];
        }
        const float gb1 = over_g_sum(bsum);
        if (dump) {
#pragma unroll
          for (int tk = 0; tk < KTM; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * tk + 4 * g + r < S) dbg[kDbgW1 + ncol * 64 + 16 * tk + 4 * g + r] = acc[tk][r];
          if (g == 0) dbg[kDbgB1 + ncol] = gb1;
        }
        stamp(11);
#pragma unroll
        for (int tk = 0; tk < KTM; ++tk)
          if (tk < KT1)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!SC || 16 * tk + r < SC) adam_elem(w1p[tk][r], acc[tk][r], w1m[tk][r], w1v[tk][r], ad);
        adam_elem(b1p, gb1, b1m, b1v, ad);
      }
      ++applied;
      std_dirty = true;
    }
    stamp(12);
    __syncthreads();          // the next image, the published weights and every read of this minibatch's buffers are complete
    stamp(13);
  }
  if (PROF && tid == 0 && dbg_all) {
    long long* out = reinterpret_cast<long long*>(dbg_all) + (ACTOR ? 0 : 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = prof[i];
  }

  // ---- write the resident state back
  if (own) {
#pragma unroll
    for (int tk = 0; tk < KTM; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * tk + 4 * g + r;
        if (k < S) {
          const int idx = net.off_w1 + ncol * S + k;
          net.param[idx] = w1p[tk][r]; net.exp_avg[idx] = w1m[tk][r]; net.exp_avg_sq[idx] = w1v[tk][r];
        }
      }
#pragma unroll
    for (int tk = 0; tk < NT; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = net.off_w2 + ncol * H + 16 * tk + 4 * g + r;
        net.param[idx] = w2p[tk][r]; net.exp_avg[idx] = w2m[tk][r]; net.exp_avg_sq[idx] = w2v[tk][r];
      }
    if (c16 < A)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = net.off_w3 + c16 * H + 16 * w + 4 * g + r;
        net.param[idx] = w3p[r]; net.exp_avg[idx] = w3m[r]; net.exp_avg_sq[idx] = w3v[r];
      }
    if (g == 0) {
      net.param[net.off_b1 + ncol] = b1p; net.exp_avg[net.off_b1 + ncol] = b1m; net.exp_avg_sq[net.off_b1 + ncol] = b1v;
      net.param[net.off_b2 + ncol] = b2p; net.exp_avg[net.off_b2 + ncol] = b2m; net.exp_avg_sq[net.off_b2 + ncol] = b2v;
      if (w == 0 && c16 < A) {
        net.param[net.off_b3 + c16] = b3p; net.exp_avg[net.off_b3 + c16] = b3m; net.exp_avg_sq[net.off_b3 + c16] = b3v;
        if (ACTOR) { net.param[net.off_std + c16] = sdp; net.exp_avg[net.off_std + c16] = sdm; net.exp_avg_sq[net.off_std + c16] = sdv; }
      }
    }
  }
  if (tid == 0) {
    *net.step_dev = steps0 + applied;
    if (out_counts) out_counts[ACTOR ? 0 : 1] = applied;
  }
}

template <int H, int MODE, int SC>
__global__ void __launch_bounds__(256)
ppo_mlp_update_kernel(dra_ppo_mlp_cfg cfg, dra_ppo_mlp_net actor, dra_ppo_mlp_net critic, const float* __restrict__ packed, int n,
                      int epochs, float* __restrict__ out3, int64_t* __restrict__ out_counts, float* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (blockIdx.x == 0) ppo_update_role<H, true, MODE, SC>(cfg, actor, packed, n, epochs, out3, out_counts, dbg, lds);
  else ppo_update_role<H, false, MODE, SC>(cfg, critic, packed, n, epochs, out3, out_counts, dbg, lds);
}

// ------------------------------------------------------------------------------------------------ pack
// PPO_agent.py:72-76: minibatch k of epoch e holds rows perm[e][k MB .. k MB + MB) of the rollout.  One LDS image per minibatch:
// [kRows][LDX] observations (columns >= S and rows past the minibatch zero) then [kRows][kLd3] = action [A] | 0 .. | log_pi_a,
// advantage, ret at columns 16, 17, 18.
__global__ void __launch_bounds__(256)
ppo_pack_kernel(const float* __restrict__ state, const float* __restrict__ action, const float* __restrict__ lp,
                const float* __restrict__ adv, const float* __restrict__ ret, const int64_t* __restrict__ perm, int n, int epochs,
                int mb, int S, int A, float* __restrict__ out) {
  const int LDX = 16 * ((S + 15) >> 4) + 4, img = kRows * (LDX + kLd3), per_epoch = (n + mb - 1) / mb;
  const int64_t total = (int64_t)epochs * per_epoch * img, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t q = i / img;
    const int o = (int)(i - q * img);
    const int e = (int)(q / per_epoch), k = (int)(q - (int64_t)e * per_epoch);
    const int rows = min(mb, n - k * mb);
    float v = 0.f;
    if (o < kRows * LDX) {
      const int row = o / LDX, c = o - row * LDX;
      if (row < rows && c < S) v = state[perm[(int64_t)e * n + k * mb + row] * S + c];
    } else {
      const int o2 = o - kRows * LDX;
      const int row = o2 / kLd3, c = o2 - row * kLd3;
      if (row < rows) {
        const int64_t src = perm[(int64_t)e * n + k * mb + row];
        if (c < A) v = action[src * A + c];
        else if (c == kAuxLp) v = lp[src];
        else if (c == kAuxAdv) v = adv[src];
        else if (c == kAuxRet) v = ret[src];
      }
    }
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------ stand-alone pieces
__global__ void __launch_bounds__(256)
rms_normalize_kernel(const double* __restrict__ x, int n, int d, double* __restrict__ mean, double* __restrict__ var,
                     double* __restrict__ count, int update, double epsilon, double clip, float* __restrict__ out_f32,
                     double* __restrict__ out_f64) {
  extern __shared__ __attribute__((aligned(16))) double s_stat[];   // mean [d], var [d]
  const double cnt = *count;
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    double m = mean[j], v = var[j];
    if (update) rms_fold(x + j, d, n, m, v, cnt);
    s_stat[j] = m;
    s_stat[d + j] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n * d; i += blockDim.x) {
    const int j = i % d;
    const double m = s_stat[j], v = s_stat[d + j];
    double z = (x[i] - m) / sqrt(v + epsilon);
    z = z < -clip ? -clip : (z > clip ? clip : z);
    if (out_f64) out_f64[i] = z;
    if (out_f32) out_f32[i] = (float)z;
  }
  if (update) {
    for (int j = threadIdx.x; j < d; j += blockDim.x) { mean[j] = s_stat[j]; var[j] = s_stat[d + j]; }
    __syncthreads();     // every thread has read *count
    if (threadIdx.x == 0) *count = cnt + (double)n;
  }
}

__global__ void __launch_bounds__(256)
gauss_sample_kernel(const float* __restrict__ mean, const float* __restrict__ scale, int n, int a_dim, uint64_t noise_seed,
                    int64_t* __restrict__ step_dev, int64_t n_global, int64_t env0, float* __restrict__ out) {
  const int64_t t = *step_dev;
  for (int i = threadIdx.x; i < n * a_dim; i += blockDim.x) {
    const int r = i / a_dim, d = i - r * a_dim;
    out[i] = gauss_noise(noise_seed, t, n_global, env0 + r, d) * scale[d] + mean[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) *step_dev = t + 1;
}

__global__ void __launch_bounds__(256)
cont_env_step_kernel(double* __restrict__ state, int64_t* __restrict__ counter, const int64_t* __restrict__ seed,
                     const float* __restrict__ action, int n, int S, int A, int64_t horizon, double* __restrict__ out_reward,
                     int32_t* __restrict__ out_done) {
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t c = counter[i] + 1;
    const uint64_t sd = (uint64_t)seed[i];
    const double mean_a = cenv_mean_action(action + (int64_t)i * A, A);
    const bool done = cenv_done(sd, c, horizon);
    for (int j = threadIdx.x; j < S; j += blockDim.x)
      state[(int64_t)i * S + j] = cenv_next_state(sd, c, j, state[(int64_t)i * S + j], mean_a, done);
    __syncthreads();
    if (threadIdx.x == 0) {
      counter[i] = c;
      out_reward[i] = cenv_reward(sd, c);
      out_done[i] = done ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------ rollout
// One workgroup walks the rollout: 8 waves, waves 0-3 the actor's network, 4-7 the critic's (each owning 16 hidden units as in
// the update kernel).  What does NOT depend on the actions is taken off the sequential chain first, in parallel over the whole
// rollout: rewards and terminals (hashes of the step counters: written straight to out_reward / out_mask) and the action noise
// (Box-Muller: parked in out_action, which step t overwrites with the action).  Per step, between workgroup barriers:
//   F1 | F2 | heads (action = mean + scale * noise, log-probability, value) | environment: every observation component |
//   running statistics: one lane per feature, rows in order (normalizer.py:39-41's arithmetic) | normalised observation t + 1
constexpr size_t rollout_lds_floats(int H) {
  // sX | sH1, sH2 (kRows x LD) | sW3 [16][LD] | sAct [kRows][kLd3] | fp64: state [kRows][kMaxS], mean, var, den [kMaxS], count [2]
  return (size_t)kRows * kLdX + 2 * (size_t)kRows * (H + 4) + 16 * (size_t)(H + 4) + (size_t)kRows * kLd3 +
         2 * ((size_t)kRows * kMaxS + 3 * kMaxS + 2) + 64;
}

// the batch x[0..N) of one feature folded into (mean, var, count): rms_fold's arithmetic with the rows fetched eight at a time
// (independent LDS reads) and added in order
__device__ __forceinline__ void rms_fold_lds(const double* x, int stride, int N, double& mean, double& var, double count) {
  double sum = 0.0;
  for (int i0 = 0; i0 < N; i0 += 8) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (i0 + k < N) ? x[(i0 + k) * stride] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k < N) sum += v[k];
  }
  // (N a power of two -- 16 workers in BASELINE configs[2]: x / N == x * (1 / N) to the last bit, and a multiplication
  // instead of a 12-instruction fp64 division sequence on the one-lane-per-feature chain)
  const bool pow2 = (N & (N - 1)) == 0;
  const double inv_n = 1.0 / (double)N;
  const double b_mean = pow2 ? sum * inv_n : sum / (double)N;
  double sq = 0.0;
  for (int i0 = 0; i0 < N; i0 += 8) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (i0 + k < N) ? x[(i0 + k) * stride] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k < N) {
        const double d = v[k] - b_mean;
        sq += d * d;
      }
  }
  const double b_var = pow2 ? sq * inv_n : sq / (double)N;
  const double n = count, b_count = (double)N, total = count + b_count;
  const double delta = b_mean - mean;
  const double m2 = var * n + b_var * b_count + delta * delta * n * b_count / total;
  mean = mean + delta * b_count / total;
  var = m2 / total;
}

template <int H, bool PROF>
__global__ void __launch_bounds__(512)
ppo_mlp_rollout_kernel(dra_ppo_mlp_cfg cfg, dra_ppo_mlp_net actor, dra_ppo_mlp_rollout_io io, long long* __restrict__ cycles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = 0;
  auto stamp = [&](int k) {
    if (PROF && threadIdx.x == 0) {
      const long long t = clock64();
      prof[k] += t - prof_last;
      prof_last = t;
    }
  };
  constexpr int NT = H / 16, LD = H + 4;
  // waves 0-3 carry the policy network (each owning 16 hidden units, as in the update kernel); all 8 waves share the
  // environment / statistics phases.  The value network is NOT on the sequential chain (nothing in the loop reads v_t): it
  // runs afterwards over all (t_len + 1) x n_env observations at once (ppo_mlp_value_kernel).
  const int tid = threadIdx.x, wv = tid >> 6, w = wv & 3, l = tid & 63, c16 = l & 15, g = l >> 4;
  const dra_ppo_mlp_net& net = actor;
  const int S = cfg.state_dim, A = cfg.action_dim;
  const int N = io.n_env, T = io.t_len;
  const int KT1 = (S + 15) >> 4, LDX = 16 * KT1 + 4;
  const int MT = (N + 15) >> 4;
  const bool own = wv < NT;
  const bool head = wv < MT;
  const int ncol = 16 * w + c16;

  float* sX = lds;
  float* sH1 = sX + kRows * kLdX;
  float* sH2 = sH1 + kRows * LD;
  float* sW3 = sH2 + kRows * LD;
  float* sAct = sW3 + 16 * LD;
  double* sState = reinterpret_cast<double*>(sAct + kRows * kLd3);
  double* sMean = sState + kRows * kMaxS;
  double* sVar = sMean + kMaxS;
  double* sDen = sVar + kMaxS;
  double* sCount = sDen + kMaxS;           // [2]
  for (int i = tid; i < (int)rollout_lds_floats(H); i += 512) lds[i] = 0.f;
  __syncthreads();

  // ---- off the chain, over the whole rollout: rewards / terminals and the action noise
  for (int i = tid; i < T * N; i += 512) {
    const int t = i / N, e = i - t * N;
    const int64_t c = io.env_counter[e] + t + 1;
    const uint64_t sdv = (uint64_t)io.env_seed[e];
    io.out_reward[i] = (float)(cenv_reward(sdv, c) * io.reward_coef);
    io.out_mask[i] = cenv_done(sdv, c, io.horizon) ? 0.f : 1.f;
  }
  const int64_t t_noise0 = *io.sampler_step;
  for (int i = tid; i < T * N * A; i += 512) {
    const int t = i / (N * A), r = i - t * (N * A), e = r / A, d = r - e * A;
    io.out_action[i] = gauss_noise(io.noise_seed, t_noise0 + t, io.n_global, io.env0 + e, d);
  }

  // weights as forward B operands (constant over the rollout)
  float w1[4][4], w2[NT][4];
#pragma unroll
  for (int tk = 0; tk < 4; ++tk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * tk + 4 * g + r;
      w1[tk][r] = (own && k < S) ? net.param[net.off_w1 + ncol * S + k] : 0.f;
    }
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int r = 0; r < 4; ++r) w2[tk][r] = own ? net.param[net.off_w2 + ncol * H + 16 * tk + 4 * g + r] : 0.f;
  const float b1 = own ? net.param[net.off_b1 + ncol] : 0.f, b2 = own ? net.param[net.off_b2 + ncol] : 0.f;
  if (own && c16 < A)
#pragma unroll
    for (int r = 0; r < 4; ++r) sW3[c16 * LD + 16 * w + 4 * g + r] = net.param[net.off_w3 + c16 * H + 16 * w + 4 * g + r];
  const float b3 = c16 < A ? net.param[net.off_b3 + c16] : 0.f;
  float sd = 1.f, log_sd = 0.f, inv_2var = 0.5f;
  if (c16 < A) {
    sd = softplus_f(net.param[net.off_std + c16]);
    log_sd = logf(sd);
    inv_2var = 1.f / (2.f * (sd * sd));
  }
  // environment + normaliser state
  for (int i = tid; i < N * S; i += 512) {
    const int e = i / S, j = i - e * S;
    sState[e * kMaxS + j] = io.env_state[i];
    sX[e * LDX + j] = io.cur_state[i];
  }
  for (int i = tid; i < S; i += 512) {
    const double m = io.rms[i], v = io.rms[S + i];
    sMean[i] = m;
    sVar[i] = v;
    sDen[i] = sqrt(v + io.rms_epsilon);
  }
  if (tid == 0) sCount[0] = io.rms[2 * S];
  // the (up to two) observation components this thread steps: environment, component, LDS slots, hashing constants
  int pe[2], pj[2];
  bool pok[2];
  uint64_t pseed[2];
  int64_t pc0[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + 512 * k;
    pok[k] = i < N * S;
    pe[k] = pok[k] ? i / S : 0;
    pj[k] = pok[k] ? i - pe[k] * S : 0;
    pseed[k] = (uint64_t)io.env_seed[pe[k]];
    pc0[k] = io.env_counter[pe[k]];
  }
  __threadfence();
  __syncthreads();

  if (PROF && tid == 0) prof_last = clock64();
  for (int t = 0; t <= T; ++t) {
    // what this step will need from global memory, requested now: the terminal flags of the environment step and the noise
    float mask_e[2] = {1.f, 1.f};
    float eps4[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < T) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (pok[k]) mask_e[k] = io.out_mask[t * N + pe[k]];
      if (head && c16 < A)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 16 * wv + 4 * g + r;
          if (e < N) eps4[r] = io.out_action[(t * N + e) * A + c16];
        }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (pok[k]) io.out_state[t * N * S + tid + 512 * k] = sX[pe[k] * LDX + pj[k]];
      for (int i = tid + 1024; i < N * S; i += 512) io.out_state[t * N * S + i] = sX[(i / S) * LDX + i % S];
    }
    // F1
    if (own && t < T) {
      f32x4 av[4][4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (mt < MT)
#pragma unroll
          for (int tk = 0; tk < 4; ++tk)
            if (tk < KT1) av[mt][tk] = *reinterpret_cast<const f32x4*>(&sX[(16 * mt + c16) * LDX + 16 * tk + 4 * g]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (mt < MT) {
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tk = 0; tk < 4; ++tk)
            if (tk < KT1) {
              a0 = MFMA16(av[mt][tk][0], w1[tk][0], a0);
              a1 = MFMA16(av[mt][tk][1], w1[tk][1], a1);
              a0 = MFMA16(av[mt][tk][2], w1[tk][2], a0);
              a1 = MFMA16(av[mt][tk][3], w1[tk][3], a1);
            }
          a0 += a1;
#pragma unroll
          for (int r = 0; r < 4; ++r) sH1[(16 * mt + 4 * g + r) * LD + ncol] = fast_tanh(a0[r] + b1);
        }
    }
    stamp(0);
    if (t == T) break;       // (the bootstrap observation's value comes from the value kernel; the sampler still counts it)
    __syncthreads();
    if (own) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        if (mt < MT) {
          f32x4 av[NT];
#pragma unroll
          for (int tk = 0; tk < NT; ++tk) av[tk] = *reinterpret_cast<const f32x4*>(&sH1[(16 * mt + c16) * LD + 16 * tk + 4 * g]);
          f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tk = 0; tk < NT; ++tk) {
            a0 = MFMA16(av[tk][0], w2[tk][0], a0);
            a1 = MFMA16(av[tk][1], w2[tk][1], a1);
            a0 = MFMA16(av[tk][2], w2[tk][2], a0);
            a1 = MFMA16(av[tk][3], w2[tk][3], a1);
          }
          a0 += a1;
#pragma unroll
          for (int r = 0; r < 4; ++r) sH2[(16 * mt + 4 * g + r) * LD + ncol] = fast_tanh(a0[r] + b2);
        }
    }
    stamp(1);
    __syncthreads();
    // head: wave mt takes environments [16 mt, 16 mt + 16)
    if (head) {
      f32x4 av[NT], bv[NT];
#pragma unroll
      for (int tk = 0; tk < NT; ++tk) {
        av[tk] = *reinterpret_cast<const f32x4*>(&sH2[(16 * wv + c16) * LD + 16 * tk + 4 * g]);
        bv[tk] = *reinterpret_cast<const f32x4*>(&sW3[c16 * LD + 16 * tk + 4 * g]);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < NT; ++tk) {
        acc = MFMA16(av[tk][0], bv[tk][0], acc);
        acc_b = MFMA16(av[tk][1], bv[tk][1], acc_b);
        acc = MFMA16(av[tk][2], bv[tk][2], acc);
        acc_b = MFMA16(av[tk][3], bv[tk][3], acc_b);
      }
      acc += acc_b;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 16 * wv + 4 * g + r;
        const float mean = fast_tanh(acc[r] + b3);
        float lpe = 0.f;
        if (c16 < A && e < N) {
          // network_heads.py:205-208: action = mean + scale * noise;  log_prob sums the per-dimension Normal log densities
          const float act = eps4[r] * sd + mean;
          const float diff = act - mean;
          lpe = -(diff * diff) * inv_2var - log_sd - kLogSqrt2Pi;
          io.out_action[(t * N + e) * A + c16] = act;
          sAct[e * kLd3 + c16] = act;
        }
        const float lp = group16_sum(lpe);
        if (c16 == 0 && e < N) io.out_log_pi_a[t * N + e] = lp;
      }
    }
    stamp(2);
    __syncthreads();
    // environment step, one thread per observation component (the mean action of its environment recomputed by each)
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (pok[k])
        sState[pe[k] * kMaxS + pj[k]] = cenv_next_state(pseed[k], pc0[k] + t + 1, pj[k], sState[pe[k] * kMaxS + pj[k]],
                                                        cenv_mean_action(sAct + pe[k] * kLd3, A), mask_e[k] == 0.f);
    for (int i = tid + 1024; i < N * S; i += 512) {     // (more than 1024 components: 64 environments x > 16 observations)
      const int e = i / S, j = i - e * S;
      sState[e * kMaxS + j] = cenv_next_state((uint64_t)io.env_seed[e], io.env_counter[e] + t + 1, j, sState[e * kMaxS + j],
                                              cenv_mean_action(sAct + e * kLd3, A), io.out_mask[t * N + e] == 0.f);
    }
    stamp(3);
    __syncthreads();
    // running statistics: one lane per feature (the rows in order), and the divisor of the normalisation
    if (io.rms_update && tid < S) {
      double m = sMean[tid], v = sVar[tid];
      rms_fold_lds(sState + tid, kMaxS, N, m, v, sCount[0]);
      sMean[tid] = m;
      sVar[tid] = v;
      sDen[tid] = sqrt(v + io.rms_epsilon);
    }
    stamp(4);
    __syncthreads();
    if (io.rms_update && tid == 0) sCount[0] = sCount[0] + (double)N;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (pok[k]) {
        double z = (sState[pe[k] * kMaxS + pj[k]] - sMean[pj[k]]) / sDen[pj[k]];
        z = z < -io.rms_clip ? -io.rms_clip : (z > io.rms_clip ? io.rms_clip : z);
        sX[pe[k] * LDX + pj[k]] = (float)z;
      }
    for (int i = tid + 1024; i < N * S; i += 512) {
      const int e = i / S, j = i - e * S;
      double z = (sState[e * kMaxS + j] - sMean[j]) / sDen[j];
      z = z < -io.rms_clip ? -io.rms_clip : (z > io.rms_clip ? io.rms_clip : z);
      sX[e * LDX + j] = (float)z;
    }
    stamp(5);
    __syncthreads();
    stamp(6);
  }
  if (PROF && tid == 0 && cycles)
    for (int i = 0; i < 8; ++i) cycles[i] = prof[i];
  __syncthreads();
  for (int i = tid; i < N * S; i += 512) {
    const int e = i / S, j = i - e * S;
    io.env_state[i] = sState[e * kMaxS + j];
    io.cur_state[i] = sX[e * LDX + j];
  }
  for (int i = tid; i < S; i += 512) { io.rms[i] = sMean[i]; io.rms[S + i] = sVar[i]; }
  __syncthreads();        // (every thread has read the counters the steps were derived from)
  for (int i = tid; i < N; i += 512) io.env_counter[i] = io.env_counter[i] + T;
  if (tid == 0) {
    io.rms[2 * S] = sCount[0];
    *io.sampler_step = t_noise0 + T + 1;
  }
}

// v = critic(observation) for all (t_len + 1) x n_env observations of a rollout at once (PPO_agent.py:35,47: the value of every
// stored observation and of the bootstrap observation): rows [0, rows_a) come from out_state, the rest from cur_state.  One
// workgroup per 64 rows, the update kernel's forward.
template <int H>
__global__ void __launch_bounds__(256)
ppo_mlp_value_kernel(dra_ppo_mlp_cfg cfg, dra_ppo_mlp_net net, const float* __restrict__ rows_a_ptr, int rows_a,
                     const float* __restrict__ rows_b_ptr, int rows_b, float* __restrict__ out_v) {
  __shared__ __attribute__((aligned(16))) float sX[kRows * kLdX];
  __shared__ __attribute__((aligned(16))) float sHa[kRows * (H + 4)];
  __shared__ __attribute__((aligned(16))) float sHb[kRows * (H + 4)];
  constexpr int NT = H / 16, LD = H + 4;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c16 = l & 15, g = l >> 4;
  const int S = cfg.state_dim, KT1 = (S + 15) >> 4, LDX = 16 * KT1 + 4;
  const int row0 = blockIdx.x * kRows, total = rows_a + rows_b;
  const bool own = w < NT;
  const int ncol = 16 * w + c16;
  for (int i = tid; i < kRows * LDX; i += 256) {
    const int r = i / LDX, c = i - r * LDX, row = row0 + r;
    float v = 0.f;
    if (c < S && row < total) v = row < rows_a ? rows_a_ptr[(int64_t)row * S + c] : rows_b_ptr[(int64_t)(row - rows_a) * S + c];
    sX[i] = v;
  }
  float w1[4][4], w2[NT][4], w3[NT][4];
#pragma unroll
  for (int tk = 0; tk < 4; ++tk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * tk + 4 * g + r;
      w1[tk][r] = (own && k < S) ? net.param[net.off_w1 + ncol * S + k] : 0.f;
    }
#pragma unroll
  for (int tk = 0; tk < NT; ++tk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w2[tk][r] = own ? net.param[net.off_w2 + ncol * H + 16 * tk + 4 * g + r] : 0.f;
      w3[tk][r] = c16 == 0 ? net.param[net.off_w3 + 16 * tk + 4 * g + r] : 0.f;     // head output 0 of 16
    }
  const float b1 = own ? net.param[net.off_b1 + ncol] : 0.f, b2 = own ? net.param[net.off_b2 + ncol] : 0.f;
  const float b3 = net.param[net.off_b3];
  __syncthreads();
  if (own) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
        if (tk < KT1) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(&sX[(16 * mt + c16) * LDX + 16 * tk + 4 * g]);
          a0 = MFMA16(av[0], w1[tk][0], a0);
          a1 = MFMA16(av[1], w1[tk][1], a1);
          a0 = MFMA16(av[2], w1[tk][2], a0);
          a1 = MFMA16(av[3], w1[tk][3], a1);
        }
      a0 += a1;
#pragma unroll
      for (int r = 0; r < 4; ++r) sHa[(16 * mt + 4 * g + r) * LD + ncol] = fast_tanh(a0[r] + b1);
    }
  }
  __syncthreads();
  if (own) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tk = 0; tk < NT; ++tk) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(&sHa[(16 * mt + c16) * LD + 16 * tk + 4 * g]);
        a0 = MFMA16(av[0], w2[tk][0], a0);
        a1 = MFMA16(av[1], w2[tk][1], a1);
        a0 = MFMA16(av[2], w2[tk][2], a0);
        a1 = MFMA16(av[3], w2[tk][3], a1);
      }
      a0 += a1;
#pragma unroll
      for (int r = 0; r < 4; ++r) sHb[(16 * mt + 4 * g + r) * LD + ncol] = fast_tanh(a0[r] + b2);
    }
  }
  __syncthreads();
  {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tk = 0; tk < NT; ++tk) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&sHb[(16 * w + c16) * LD + 16 * tk + 4 * g]);
      a0 = MFMA16(av[0], w3[tk][0], a0);
      a1 = MFMA16(av[1], w3[tk][1], a1);
      a0 = MFMA16(av[2], w3[tk][2], a0);
      a1 = MFMA16(av[3], w3[tk][3], a1);
    }
    a0 += a1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 16 * w + 4 * g + r;
      if (c16 == 0 && row < total) out_v[row] = a0[r] + b3;
    }
  }
}

int check_net(const dra_ppo_mlp_net* n, bool actor) {
  if (!n || !n->param || !n->exp_avg || !n->exp_avg_sq || !n->step_dev) return DRA_EINVAL;
  if (n->off_w1 < 0 || n->off_b1 < 0 || n->off_w2 < 0 || n->off_b2 < 0 || n->off_w3 < 0 || n->off_b3 < 0) return DRA_EINVAL;
  if (actor && n->off_std < 0) return DRA_EINVAL;
  if (!(n->eps > 0.f) || !(n->beta1 >= 0.f && n->beta1 < 1.f) || !(n->beta2 >= 0.f && n->beta2 < 1.f)) return DRA_EINVAL;
  return DRA_OK;
}

}  // namespace

DRA_API int dra_ppo_mlp_supported(int state_dim, int action_dim, int hidden1, int hidden2, int mini_batch) {
  if (state_dim < 1 || state_dim > kMaxS || action_dim < 1 || action_dim > kMaxA) return DRA_EINVAL;
  if (hidden1 != hidden2 || (hidden1 != 16 && hidden1 != 32 && hidden1 != 64)) return DRA_EINVAL;
  if (mini_batch < 1 || mini_batch > kRows) return DRA_EINVAL;
  if (update_lds_floats(hidden1, state_dim) > kLdsFloatsMax) return DRA_EINVAL;      // (hidden 64: up to 48 observations)
  return DRA_OK;
}

DRA_API int dra_ppo_mlp_packed_floats(int n, int epochs, int mini_batch, int s_dim, int64_t* floats) {
  if (!floats || n < 1 || epochs < 1 || mini_batch < 1 || mini_batch > kRows || s_dim < 1 || s_dim > kMaxS) return DRA_EINVAL;
  const int LDX = 16 * ((s_dim + 15) >> 4) + 4;
  *floats = (int64_t)epochs * ((n + mini_batch - 1) / mini_batch) * kRows * (LDX + kLd3);
  return DRA_OK;
}

DRA_API int dra_ppo_mlp_pack(const float* state, const float* action, const float* log_pi_a, const float* advantage, const float* ret,
                             const int64_t* perm, int n, int epochs, int mini_batch, int s_dim, int a_dim, float* out_packed,
                             void* stream) {
  if (!state || !action || !log_pi_a || !advantage || !ret || !perm || !out_packed) return DRA_EINVAL;
  int64_t floats = 0;
  if (dra_ppo_mlp_packed_floats(n, epochs, mini_batch, s_dim, &floats) || a_dim < 1 || a_dim > kMaxA) return DRA_EINVAL;
  int64_t blocks = (floats + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(ppo_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, dra_stream(stream), state, action, log_pi_a, advantage,
                     ret, perm, n, epochs, mini_batch, s_dim, a_dim, out_packed);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

template <int H, int MODE, int SC>
static int launch_update(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic, const float* packed,
                         int n, int epochs, float* out3, int64_t* out_counts, float* dbg, void* stream) {
  const size_t bytes = update_lds_floats(H, cfg->state_dim) * sizeof(float);
  static DraLdsAttr lds_attr;
  if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&ppo_mlp_update_kernel<H, MODE, SC>), bytes)) return rc;
  hipLaunchKernelGGL((ppo_mlp_update_kernel<H, MODE, SC>), dim3(2), dim3(256), bytes, dra_stream(stream), *cfg, *actor, *critic, packed,
                     n, epochs, out3, out_counts, dbg);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_ppo_mlp_update(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                               const float* packed, int n, int epochs, float* out3, int64_t* out_counts, float* dbg, void* stream) {
  if (!cfg || !packed || !out3 || n < 1 || epochs < 1) return DRA_EINVAL;
  if (dra_ppo_mlp_supported(cfg->state_dim, cfg->action_dim, cfg->hidden, cfg->hidden, cfg->mini_batch)) return DRA_EINVAL;
  if (check_net(actor, true) || check_net(critic, false)) return DRA_EINVAL;
  // the dump build is a separate instantiation (its stores would otherwise cost the product kernel registers); hidden = 64 with
  // 17 observations (HalfCheetah / Walker2d: examples.py:497-523) or 11 (Hopper / Reacher) has the observation size compiled in
#define DRA_PPO_UPD(HH, SS) (dbg ? launch_update<HH, 1, 0>(cfg, actor, critic, packed, n, epochs, out3, out_counts, dbg, stream) \
                                 : launch_update<HH, 0, SS>(cfg, actor, critic, packed, n, epochs, out3, out_counts, dbg, stream))
  switch (cfg->hidden) {
    case 16: return DRA_PPO_UPD(16, 0);
    case 32: return DRA_PPO_UPD(32, 0);
    default: return cfg->state_dim == 17 ? DRA_PPO_UPD(64, 17) : (cfg->state_dim == 11 ? DRA_PPO_UPD(64, 11) : DRA_PPO_UPD(64, 0));
  }
#undef DRA_PPO_UPD
}

// measurement aid (tools/prof_ppo_mlp.py): the same launch with thread 0 of each workgroup accumulating shader-clock cycles per
// phase of the minibatch loop; cycles (device int64 [2][16]): actor row, critic row.
DRA_API int dra_ppo_mlp_update_profile(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                                       const float* packed, int n, int epochs, float* out3, int64_t* out_counts, int64_t* cycles,
                                       void* stream) {
  if (!cfg || !packed || !out3 || !cycles || n < 1 || epochs < 1 || cfg->hidden != 64) return DRA_EINVAL;
  if (dra_ppo_mlp_supported(cfg->state_dim, cfg->action_dim, cfg->hidden, cfg->hidden, cfg->mini_batch)) return DRA_EINVAL;
  if (check_net(actor, true) || check_net(critic, false)) return DRA_EINVAL;
  if (cfg->state_dim == 17)
    return launch_update<64, 2, 17>(cfg, actor, critic, packed, n, epochs, out3, out_counts, reinterpret_cast<float*>(cycles), stream);
  return launch_update<64, 2, 0>(cfg, actor, critic, packed, n, epochs, out3, out_counts, reinterpret_cast<float*>(cycles), stream);
}

template <int H, bool PROF>
static int launch_rollout(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                          const dra_ppo_mlp_rollout_io* io, int64_t* cycles, void* stream) {
  const size_t bytes = rollout_lds_floats(H) * sizeof(float);
  static DraLdsAttr lds_attr;
  if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&ppo_mlp_rollout_kernel<H, PROF>), bytes)) return rc;
  hipLaunchKernelGGL((ppo_mlp_rollout_kernel<H, PROF>), dim3(1), dim3(512), bytes, dra_stream(stream), *cfg, *actor, *io,
                     reinterpret_cast<long long*>(cycles));
  DRA_LAUNCH_CHECK();
  // the values of the t_len x n_env stored observations and of the bootstrap observation (the one the rollout left in cur_state)
  const int rows_a = io->t_len * io->n_env, rows_b = io->n_env;
  hipLaunchKernelGGL(ppo_mlp_value_kernel<H>, dim3((unsigned)((rows_a + rows_b + kRows - 1) / kRows)), dim3(256), 0, dra_stream(stream),
                     *cfg, *critic, (const float*)io->out_state, rows_a, (const float*)io->cur_state, rows_b, io->out_v);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

static int rollout_checked(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                           const dra_ppo_mlp_rollout_io* io) {
  if (!cfg || !io) return DRA_EINVAL;
  if (dra_ppo_mlp_supported(cfg->state_dim, cfg->action_dim, cfg->hidden, cfg->hidden, 1)) return DRA_EINVAL;
  if (check_net(actor, true) || check_net(critic, false)) return DRA_EINVAL;
  if (io->n_env < 1 || io->n_env > kRows || io->t_len < 1 || io->horizon < 1 || io->n_global < io->n_env || io->env0 < 0) return DRA_EINVAL;
  if ((int64_t)(io->t_len + 1) * io->n_env * (cfg->state_dim > cfg->action_dim ? cfg->state_dim : cfg->action_dim) > 0x7fffffff) return DRA_EINVAL;
  if (!io->env_state || !io->env_counter || !io->env_seed || !io->rms || !io->cur_state || !io->sampler_step || !io->out_state ||
      !io->out_action || !io->out_log_pi_a || !io->out_v || !io->out_reward || !io->out_mask)
    return DRA_EINVAL;
  return DRA_OK;
}

DRA_API int dra_ppo_mlp_rollout(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                                const dra_ppo_mlp_rollout_io* io, void* stream) {
  if (rollout_checked(cfg, actor, critic, io)) return DRA_EINVAL;
  switch (cfg->hidden) {
    case 16: return launch_rollout<16, false>(cfg, actor, critic, io, nullptr, stream);
    case 32: return launch_rollout<32, false>(cfg, actor, critic, io, nullptr, stream);
    default: return launch_rollout<64, false>(cfg, actor, critic, io, nullptr, stream);
  }
}

// measurement aid: the same launch (hidden = 64) with thread 0's shader-clock cycles per phase of the step loop in cycles [8]
DRA_API int dra_ppo_mlp_rollout_profile(const dra_ppo_mlp_cfg* cfg, const dra_ppo_mlp_net* actor, const dra_ppo_mlp_net* critic,
                                        const dra_ppo_mlp_rollout_io* io, int64_t* cycles, void* stream) {
  if (rollout_checked(cfg, actor, critic, io) || !cycles || cfg->hidden != 64) return DRA_EINVAL;
  return launch_rollout<64, true>(cfg, actor, critic, io, cycles, stream);
}

DRA_API int dra_rms_normalize(const double* x, int n, int d, double* mean, double* var, double* count, int update, double epsilon,
                              double clip, float* out_f32, double* out_f64, void* stream) {
  if (!x || !mean || !var || !count || n < 1 || d < 1 || d > 4096) return DRA_EINVAL;
  hipLaunchKernelGGL(rms_normalize_kernel, dim3(1), dim3(256), 2 * (size_t)d * sizeof(double), dra_stream(stream), x, n, d, mean, var,
                     count, update, epsilon, clip, out_f32, out_f64);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_gauss_sample(const float* mean, const float* scale, int n, int a_dim, uint64_t noise_seed, int64_t* step_dev,
                             int64_t n_global, int64_t env0, float* out_action, void* stream) {
  if (!mean || !scale || !step_dev || !out_action || n < 1 || a_dim < 1 || a_dim > 32 || n_global < n || env0 < 0) return DRA_EINVAL;
  hipLaunchKernelGGL(gauss_sample_kernel, dim3(1), dim3(256), 0, dra_stream(stream), mean, scale, n, a_dim, noise_seed, step_dev,
                     n_global, env0, out_action);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_cont_env_step(double* state, int64_t* counter, const int64_t* seed, const float* action, int n, int s_dim, int a_dim,
                              int64_t horizon, double* out_reward, int32_t* out_done, void* stream) {
  if (!state || !counter || !seed || !action || !out_reward || !out_done || n < 1 || s_dim < 1 || s_dim > 64 || a_dim < 1 ||
      horizon < 1)
    return DRA_EINVAL;
  hipLaunchKernelGGL(cont_env_step_kernel, dim3((unsigned)(n < 1024 ? n : 1024)), dim3(64), 0, dra_stream(stream), state, counter, seed,
                     action, n, s_dim, a_dim, horizon, out_reward, out_done);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
