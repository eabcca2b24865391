// Device side of the synthetic CONTINUOUS-control environment, the running observation normaliser and the Gaussian policy's
// action noise -- the three pieces of BASELINE configs[2] (PPO HalfCheetah-shaped, examples.py:497-523) that sat on the host
// until round 5.  Shared by the stand-alone kernels of ppo_mlp.hip (dra_cont_env_step, dra_rms_normalize, dra_gauss_noise:
// what the parity tests call) and by the persistent rollout kernel, so that one arithmetic is tested and used.
//
// Environment (deeprl_amd/envs.py SyntheticContinuous is the host statement of the same function, bit for bit; MuJoCo itself
// is third-party CPU code outside the hot path): every quantity is a hash of (seed, stream, step counter c, j):
//   step(a):  c += 1;  m = (((a0 + a1) + a2) + ...) / A   (fp64, actions clipped to [-1, 1] in fp32 first: envs.py:186-189)
//             s_j = (s_j + 0.01 m) + (-0.02 + 0.04 U(0, c, j))
//             reward = (((U(1,c,0) + U(1,c,1)) + (U(1,c,2) + U(1,c,3))) - 2) * sqrt(3)          (mean 0, variance 1)
//             done = hash(2, c, 0) mod horizon == 0;   done -> s_j = -0.05 + 0.1 U(3, c, j)    (DummyVecEnv's auto reset)
// fp64, one rounding per written operation (the library is built with -ffp-contract=off).
#pragma once
#include "common.h"

__host__ __device__ __forceinline__ uint64_t cenv_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t cenv_hash(uint64_t seed, int stream, int64_t c, int j) {
  return cenv_mix64((seed * 8ull + (uint64_t)stream + 1ull) * 0x9E3779B97F4A7C15ull + (uint64_t)c * 64ull + (uint64_t)j);
}
// uniform in [0, 1): 53 hashed bits, exact in fp64
__host__ __device__ __forceinline__ double cenv_u(uint64_t seed, int stream, int64_t c, int j) {
  return (double)(cenv_hash(seed, stream, c, j) >> 11) * 1.1102230246251565e-16;
}
__host__ __device__ __forceinline__ double cenv_reward(uint64_t seed, int64_t c) {
  const double u = ((cenv_u(seed, 1, c, 0) + cenv_u(seed, 1, c, 1)) + (cenv_u(seed, 1, c, 2) + cenv_u(seed, 1, c, 3))) - 2.0;
  return u * 1.7320508075688772;
}
__host__ __device__ __forceinline__ bool cenv_done(uint64_t seed, int64_t c, int64_t horizon) {
  return (cenv_hash(seed, 2, c, 0) % (uint64_t)horizon) == 0;
}
__host__ __device__ __forceinline__ double cenv_reset_state(uint64_t seed, int64_t c, int j) {
  return -0.05 + 0.1 * cenv_u(seed, 3, c, j);
}
// mean action of one environment: fp32 actions clipped to [-1, 1], summed left to right in fp64
__device__ __forceinline__ double cenv_mean_action(const float* a, int A) {
  double m = 0.0;
  for (int d = 0; d < A; ++d) m += (double)fminf(fmaxf(a[d], -1.f), 1.f);
  return m / (double)A;
}
// component j of the next observation of an environment whose counter has ALREADY been advanced to c
__device__ __forceinline__ double cenv_next_state(uint64_t seed, int64_t c, int j, double s, double mean_a, bool done) {
  if (done) return cenv_reset_state(seed, c, j);
  return (s + 0.01 * mean_a) + (-0.02 + 0.04 * cenv_u(seed, 0, c, j));
}

// ---- Gaussian policy noise: a standard normal per (sampler step t, GLOBAL environment i, action dimension d), Box-Muller over
// two 24-bit hashed uniforms in fp32.  The stream position is (t, i, d) alone, so a rollout spread over ranks, replayed from a
// graph or run inside the persistent kernel draws the same numbers (what dist.DataParallel's invariant sampling needs and what
// lets the device rollout be compared with the step-by-step path).
__device__ __forceinline__ float gauss_noise(uint64_t noise_seed, int64_t t, int64_t n_global, int64_t i, int d) {
  const uint64_t base = (noise_seed * 8ull + 6ull) * 0x9E3779B97F4A7C15ull + (uint64_t)(t * n_global + i) * 64ull + 2ull * (uint64_t)d;
  const uint64_t h1 = cenv_mix64(base), h2 = cenv_mix64(base + 1ull);
  const float u1 = (float)((uint32_t)(h1 >> 40) + 1u) * 5.9604644775390625e-8f;   // (0, 1]
  const float u2 = (float)(uint32_t)(h2 >> 40) * 5.9604644775390625e-8f;          // [0, 1)
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853071795865f * u2);
}

// ---- baselines' RunningMeanStd (normalizer.py:8,39-41; deeprl_amd/normalizers.py RunningMeanStd.update / merge, whose
// operation order this follows so that the statistics agree to the bit): the batch x[0..N) of ONE feature folded into
// (mean, var, count).  np.mean / np.var over axis 0 add the rows in order.
__device__ __forceinline__ void rms_fold(const double* x, int64_t stride, int N, double& mean, double& var, double count) {
  double sum = 0.0;
  for (int i = 0; i < N; ++i) sum += x[(int64_t)i * stride];
  // (N a power of two -- 16 workers in BASELINE configs[2]: x / N == x * (1 / N) to the last bit, and a multiplication
  // instead of a 12-instruction fp64 division sequence on the one-lane-per-feature chain)
  const bool pow2 = (N & (N - 1)) == 0;
  const double inv_n = 1.0 / (double)N;
  const double b_mean = pow2 ? sum * inv_n : sum / (double)N;
  double sq = 0.0;
  for (int i = 0; i < N; ++i) {
    const double d = x[(int64_t)i * stride] - b_mean;
    sq += d * d;
  }
  const double b_var = pow2 ? sq * inv_n : sq / (double)N;
  const double n = count, b_count = (double)N, total = count + b_count;
  const double delta = b_mean - mean;
  const double m2 = var * n + b_var * b_count + delta * delta * n * b_count / total;
  mean = mean + delta * b_count / total;
  var = m2 / total;
}
// clip((x - mean) / sqrt(var + epsilon), +-clip) in fp64, then what tensor() uploads (torch_utils.py:23): float32
__device__ __forceinline__ float rms_apply(double x, double mean, double var, double epsilon, double clip) {
  double z = (x - mean) / sqrt(var + epsilon);
  z = z < -clip ? -clip : (z > clip ? clip : z);
  return (float)z;
}
