// K9: GAE / n-step-return reverse recurrences as an LDS-staged chunked affine scan.
// Replaces the Python reverse loops of deep_rl/agent/PPO_agent.py:51-61, A2C_agent.py:43-53 and
// NStepDQN_agent.py:56-60 (T up to 2048 tiny-tensor op chains) with ONE launch.
//
//   ret_t = r_t + (gamma*m_t) * ret_{t+1}                       ret_T = v_T
//   td_t  = (r_t + (gamma*m_t) * v_{t+1}) - v_t
//   adv_t = ((adv_{t+1}*tau)*gamma)*m_t + td_t                  adv_T = 0      (use_gae)
//   adv_t = ret_t - v_t                                                        (!use_gae)
//
// One wave per environment, lane c owns timesteps [c*L, (c+1)*L), L = ceil(T/64):
//   1. r/m/v rows are staged into LDS with coalesced loads ([env][t] planes, one pad word per
//      32 so lanes L apart hit different banks);
//   2. each lane composes its chunk into an affine map x_lo = A*x_hi + B (serial, L steps);
//   3. a 6-step shuffle suffix-scan composes the maps across lanes, giving every lane the
//      exact value entering its chunk;
//   4. each lane replays its chunk with the reference's own operation order and writes adv/ret
//      back through LDS with coalesced stores.
// Only the 63 chunk-boundary values carry scan rounding (~1e-7 relative); everything inside a
// chunk is evaluated exactly as the reference does.  Algorithmic bytes: (5T+1)*N*4.
#include "common.h"

__device__ __forceinline__ int pad_t(int t) { return t + (t >> 5); }

template <int EPB>
__global__ void __launch_bounds__(EPB * 64)
gae_scan_kernel(const float* __restrict__ r, const float* __restrict__ m, const float* __restrict__ v, int T, int N,
                float gamma, float tau, int use_gae, float* __restrict__ adv, float* __restrict__ ret) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Tp = pad_t(T + 1) + 1;
  float* s_r = lds;
  float* s_m = s_r + EPB * Tp;
  float* s_v = s_m + EPB * Tp;
  const int e0 = blockIdx.x * EPB;
  const int nthreads = EPB * 64;
  for (int i = threadIdx.x; i < (T + 1) * EPB; i += nthreads) {
    const int t = i / EPB, el = i - t * EPB;
    if (e0 + el < N) {
      const int o = el * Tp + pad_t(t);
      s_v[o] = v[(int64_t)t * N + e0 + el];
      if (t < T) {
        s_r[o] = r[(int64_t)t * N + e0 + el];
        s_m[o] = m[(int64_t)t * N + e0 + el];
      }
    }
  }
  __syncthreads();
  const int el = threadIdx.x >> 6, c = threadIdx.x & 63;
  const bool env_on = (e0 + el) < N;
  const int L = (T + 63) / 64;
  const int t_lo = min(T, c * L), t_hi = min(T, c * L + L);
  float* pr = s_r + el * Tp;
  float* pm = s_m + el * Tp;
  const float* pv = s_v + el * Tp;
  const float tg = tau * gamma;
  // phase 2 of the header: compose this lane's chunk, newest timestep first
  float Ar = 1.f, Br = 0.f, Aa = 1.f, Ba = 0.f;
  if (env_on) {
    for (int t = t_hi - 1; t >= t_lo; --t) {
      const int o = pad_t(t);
      const float gm = gamma * pm[o];
      Br = pr[o] + gm * Br;
      Ar = gm * Ar;
      if (use_gae) {
        const float td = (pr[o] + gm * pv[pad_t(t + 1)]) - pv[o];
        const float ga = tg * pm[o];
        Ba = ga * Ba + td;
        Aa = ga * Aa;
      }
    }
  }
  // phase 3: inclusive suffix scan of the affine maps over the 64 lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float Ar2 = __shfl_down(Ar, off, 64), Br2 = __shfl_down(Br, off, 64);
    const float Aa2 = __shfl_down(Aa, off, 64), Ba2 = __shfl_down(Ba, off, 64);
    if (c + off < 64) {
      Br = Ar * Br2 + Br; Ar = Ar * Ar2;
      Ba = Aa * Ba2 + Ba; Aa = Aa * Aa2;
    }
  }
  const float x_r = env_on ? pv[pad_t(T)] : 0.f;  // ret_T = v_T ; adv_T = 0
  const float nAr = __shfl_down(Ar, 1, 64), nBr = __shfl_down(Br, 1, 64), nBa = __shfl_down(Ba, 1, 64);
  float ret_in = (c == 63) ? x_r : nAr * x_r + nBr;
  float adv_in = (c == 63) ? 0.f : nBa;
  // phase 4: replay the chunk in the reference's operation order; outputs overwrite r/m in LDS
  if (env_on) {
    for (int t = t_hi - 1; t >= t_lo; --t) {
      const int o = pad_t(t);
      const float rt = pr[o], mt = pm[o];
      ret_in = __fadd_rn(rt, __fmul_rn(__fmul_rn(gamma, mt), ret_in));
      if (use_gae) {
        const float td = __fsub_rn(__fadd_rn(rt, __fmul_rn(__fmul_rn(gamma, mt), pv[pad_t(t + 1)])), pv[o]);
        adv_in = __fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn(adv_in, tau), gamma), mt), td);
      } else {
        adv_in = __fsub_rn(ret_in, pv[o]);
      }
      pr[o] = adv_in;
      pm[o] = ret_in;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * EPB; i += nthreads) {
    const int t = i / EPB, e = i - t * EPB;
    if (e0 + e < N) {
      const int o = e * Tp + pad_t(t);
      adv[(int64_t)t * N + e0 + e] = s_r[o];
      ret[(int64_t)t * N + e0 + e] = s_m[o];
    }
  }
}

template <int EPB>
static int launch_gae(const float* r, const float* m, const float* v, int T, int N, float gamma, float tau, int use_gae,
                      float* adv, float* ret, hipStream_t st) {
  const int Tp = (T + 1) + ((T + 1) >> 5) + 1;
  const size_t lds = (size_t)3 * EPB * Tp * sizeof(float);
  if (lds > 160 * 1024) return DRA_EINVAL;
  if (lds > 64 * 1024)
    DRA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gae_scan_kernel<EPB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(gae_scan_kernel<EPB>, dim3((N + EPB - 1) / EPB), dim3(EPB * 64), lds, st, r, m, v, T, N, gamma, tau,
                     use_gae, adv, ret);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// r, m: [T,N] f32 ; v: [T+1,N] f32 (row T bootstraps) ; adv, ret: [T,N] f32.
DRA_API int dra_gae(const float* reward, const float* mask, const float* value, int t_len, int n_env, float gamma,
                    float tau, int use_gae, float* out_adv, float* out_ret, void* stream) {
  if (!reward || !mask || !value || !out_adv || !out_ret || t_len < 1 || n_env < 1) return DRA_EINVAL;
  const size_t per_env = (size_t)3 * ((t_len + 1) + ((t_len + 1) >> 5) + 1) * sizeof(float);
  hipStream_t st = dra_stream(stream);
  // as many environments per workgroup as LDS allows, but keep >= 1 workgroup per 4 envs busy
  if (per_env * 4 <= 150 * 1024) return launch_gae<4>(reward, mask, value, t_len, n_env, gamma, tau, use_gae, out_adv, out_ret, st);
  if (per_env * 2 <= 150 * 1024) return launch_gae<2>(reward, mask, value, t_len, n_env, gamma, tau, use_gae, out_adv, out_ret, st);
  return launch_gae<1>(reward, mask, value, t_len, n_env, gamma, tau, use_gae, out_adv, out_ret, st);
}

// Advantage normalisation (PPO_agent.py:66): a <- (a - mean) / std, unbiased std, over all T*N
// entries; one workgroup, fp64 accumulation (two passes over <= a few 100 KB that sit in L2).
__global__ void __launch_bounds__(1024)
adv_normalize_kernel(float* __restrict__ a, int64_t n) {
  __shared__ double s_red[16];
  __shared__ double s_stat[2];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += (double)a[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
    s_stat[0] = t / (double)n;
  }
  __syncthreads();
  const double mean = s_stat[0];
  double q = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { const double d = (double)a[i] - mean; q += d * d; }
  q = wave_sum(q);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
    s_stat[1] = sqrt(t / (double)(n - 1));
  }
  __syncthreads();
  const float fm = (float)mean, fs = (float)s_stat[1];
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a[i] = (a[i] - fm) / fs;
}

DRA_API int dra_adv_normalize(float* adv, int64_t n, void* stream) {
  if (!adv || n < 2) return DRA_EINVAL;
  hipLaunchKernelGGL(adv_normalize_kernel, dim3(1), dim3(1024), 0, dra_stream(stream), adv, n);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
