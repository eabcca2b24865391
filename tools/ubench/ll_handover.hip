// Micro-benchmark (measurement aid): what does ONE cross-workgroup hand-over inside a running kernel cost on this part, in the two
// forms the actor's one-launch env step can use?
//   flag : the producer group stores its values agent-scope (written through), waits for them (s_waitcnt vmcnt(0)), and one
//          thread per workgroup counts itself on an arrival counter; every consumer polls the counter, then reads the values with
//          agent-scope loads (conv_v2.hip MegaSync: what actor_c3fc4_kernel does today);
//   ll   : every value travels as one 8-byte {value, tag} word (the "LL" form of the collective libraries): the producer just
//          stores, the consumer re-reads ITS words until the tag is the stage's -- no wait for the stores, no counter, no poll
//          in front of the data; tags grow monotonically, so nothing is ever reset.
// Two groups of P workgroups (512 threads) play ping-pong for S stages: group (s & 1) reads ALL N values of stage s-1 (every
// workgroup of the group, like conv3 reading conv2's planes or fc4 reading conv3's), then writes its share of stage s
// (value + 1).  Reported: microseconds per stage = one hand-over + a trivial amount of arithmetic.  The final values are checked.
//   hipcc --offload-arch=gfx950 -O3 -o ll_handover tools/ubench/ll_handover.hip ;  ./ll_handover [cu_mask_bits]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NT = 512;

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter(); }

struct Args {
  float* plain[2];                 // flag form: ping-pong value buffers
  unsigned* counters;              // flag form: one arrival counter per stage
  unsigned long long* ll[2];       // LL form: ping-pong {value, tag} buffers
  int n, p, stages;
  unsigned tag0;                   // LL: stage s carries tag0 + s + 1
  int* fail;
};

template <int MAXQ>
__global__ void __launch_bounds__(NT) flag_kernel(const Args a) {
  const int grp = blockIdx.x / a.p, w = blockIdx.x - grp * a.p, tid = threadIdx.x;
  const int share = a.n / a.p;
  float v[MAXQ];
  for (int s = 0; s < a.stages; ++s) {
    if ((s & 1) != grp) continue;
    float sum = 0.f;
    if (s > 0) {
      if (tid == 0) {
        long spins = 0;
        while (__hip_atomic_load(a.counters + (s - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.p) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 20000000) { *a.fail = 1; break; }
        }
      }
      __syncthreads();
      const float* src = a.plain[(s - 1) & 1];
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        const int i = tid + NT * q;
        v[q] = __hip_atomic_load(src + (i < a.n ? i : a.n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) sum += v[q];
    }
    // (a trivial dependency on everything read: the stage's value is the same everywhere)
    __shared__ float s_first;
    if (tid == 0) s_first = s > 0 ? v[0] : 0.f;
    __syncthreads();
    const float out = s_first + 1.f + (sum != sum ? 1.f : 0.f);
    float* dst = a.plain[s & 1];
    for (int i = tid; i < share; i += NT) __hip_atomic_store(dst + w * share + i, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.counters + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MAXQ>
__global__ void __launch_bounds__(NT) ll_kernel(const Args a) {
  const int grp = blockIdx.x / a.p, w = blockIdx.x - grp * a.p, tid = threadIdx.x;
  const int share = a.n / a.p;
  unsigned long long v[MAXQ];
  for (int s = 0; s < a.stages; ++s) {
    if ((s & 1) != grp) continue;
    float sum = 0.f;
    if (s > 0) {
      const unsigned long long* src = a.ll[(s - 1) & 1];
      const unsigned want = a.tag0 + (unsigned)s;      // stage s-1's tag
      long spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
          const int i = tid + NT * q;
          v[q] = __hip_atomic_load(src + (i < a.n ? i : a.n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) ok = ok && (unsigned)(v[q] >> 32) == want;
        if (ok) break;
        if (++spins > 2000000) { *a.fail = 2; break; }
      }
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) sum += __uint_as_float((unsigned)v[q]);
    }
    __shared__ float s_first;
    if (tid == 0) s_first = s > 0 ? __uint_as_float((unsigned)v[0]) : 0.f;
    __syncthreads();
    const float out = s_first + 1.f + (sum != sum ? 1.f : 0.f);
    const unsigned long long word = ((unsigned long long)(a.tag0 + (unsigned)s + 1u) << 32) | __float_as_uint(out);
    unsigned long long* dst = a.ll[s & 1];
    for (int i = tid; i < share; i += NT) __hip_atomic_store(dst + w * share + i, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();   // (s_first is rewritten next round)
  }
}


// Same-XCD forms: only workgroups with blockIdx % 8 == 0 take part (the dispatcher deals workgroups round-robin over the 8 XCDs),
// so both groups share ONE L2.  LOCAL = 0: the agent-scope counter protocol above, unchanged (what the update's chained launches
// use today); LOCAL = 1: plain stores, completed (s_waitcnt), a workgroup-scope atomic (executed in this XCD's L2), and
// non-temporal loads for the poll and the data (lines that are not kept in the CU's L1: every poll is served by the L2).
template <int MAXQ, int LOCAL>
__global__ void __launch_bounds__(NT) flag_xcd_kernel(const Args a) {
  if (blockIdx.x & 7) return;
  const int wg = blockIdx.x >> 3;
  const int grp = wg / a.p, w = wg - grp * a.p, tid = threadIdx.x;
  const int share = a.n / a.p;
  float v[MAXQ];
  for (int s = 0; s < a.stages; ++s) {
    if ((s & 1) != grp) continue;
    float sum = 0.f;
    if (s > 0) {
      if (tid == 0) {
        long spins = 0;
        for (;;) {
          unsigned c;
          if (LOCAL) c = __builtin_nontemporal_load(a.counters + (s - 1));
          else c = __hip_atomic_load(a.counters + (s - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (c >= (unsigned)a.p) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000) { *a.fail = 3 + LOCAL; break; }
        }
      }
      __syncthreads();
      const float* src = a.plain[(s - 1) & 1];
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) {
        const int i = tid + NT * q;
        const float* pp = src + (i < a.n ? i : a.n - 1);
        v[q] = LOCAL ? __builtin_nontemporal_load(pp) : __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int q = 0; q < MAXQ; ++q) sum += v[q];
    }
    __shared__ float s_first;
    if (tid == 0) s_first = s > 0 ? v[0] : 0.f;
    __syncthreads();
    const float out = s_first + 1.f + (sum != sum ? 1.f : 0.f);
    float* dst = a.plain[s & 1];
    for (int i = tid; i < share; i += NT) {
      if (LOCAL) dst[w * share + i] = out;
      else __hip_atomic_store(dst + w * share + i, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (LOCAL) __hip_atomic_fetch_add(a.counters + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(a.counters + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main(int argc, char** argv) {
  const int mask_bits = argc > 1 ? atoi(argv[1]) : 0;   // > 0: run on a stream restricted to the first `mask_bits` CUs
  hipStream_t st;
  if (mask_bits > 0) {
    std::vector<uint32_t> mask(8, 0);
    for (int b = 0; b < mask_bits; ++b) mask[b / 32] |= 1u << (b % 32);
    CK(hipExtStreamCreateWithCUMask(&st, 8, mask.data()));
  } else {
    CK(hipStreamCreate(&st));
  }
  const int S = 200;
  Args a;
  const int NMAX = 8192;
  for (int k = 0; k < 2; ++k) {
    CK(hipMalloc(&a.plain[k], NMAX * sizeof(float)));
    CK(hipMalloc(&a.ll[k], NMAX * sizeof(unsigned long long)));
    CK(hipMemset(a.ll[k], 0, NMAX * sizeof(unsigned long long)));
  }
  CK(hipMalloc(&a.counters, S * sizeof(unsigned)));
  int* fail; CK(hipHostMalloc(&fail, sizeof(int))); *fail = 0;
  a.fail = fail; a.stages = S;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("{\"cu_mask_bits\": %d, \"stages\": %d, \"cases\": [", mask_bits, S);
  bool first = true;
  unsigned tag = 0;
  const int ns[3] = {512, 3072, 6144};
  const int ps[3] = {8, 16, 32};
  for (int ni = 0; ni < 3; ++ni)
    for (int pi = 0; pi < 3; ++pi) {
      a.n = ns[ni]; a.p = ps[pi];
      double us[2] = {0, 0};
      float got[2] = {0, 0};
      for (int form = 0; form < 2; ++form) {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipMemsetAsync(a.counters, 0, S * sizeof(unsigned), st));
          a.tag0 = tag; tag += S + 1;
          CK(hipEventRecord(e0, st));
          if (form == 0) hipLaunchKernelGGL(flag_kernel<12>, dim3(2 * a.p), dim3(NT), 0, st, a);
          else hipLaunchKernelGGL(ll_kernel<12>, dim3(2 * a.p), dim3(NT), 0, st, a);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms * 1e3 / S < best) best = ms * 1e3 / S;
        }
        us[form] = best;
        if (form == 0) { CK(hipMemcpy(&got[0], a.plain[(S - 1) & 1], sizeof(float), hipMemcpyDeviceToHost)); }
        else { unsigned long long wd; CK(hipMemcpy(&wd, a.ll[(S - 1) & 1], 8, hipMemcpyDeviceToHost)); unsigned lo = (unsigned)wd; memcpy(&got[1], &lo, 4); }
      }
      printf("%s{\"n_values\": %d, \"workgroups_per_group\": %d, \"flag_us_per_stage\": %.3f, \"ll_us_per_stage\": %.3f, \"final_flag\": %.0f, \"final_ll\": %.0f}",
             first ? "" : ", ", a.n, a.p, us[0], us[1], got[0], got[1]);
      first = false;
    }
  printf("], \"same_xcd\": [");
  first = true;
  for (int ni = 0; ni < 3; ++ni)
    for (int pi = 0; pi < 2; ++pi) {
      a.n = ns[ni]; a.p = ps[pi] / 2;      // 4 / 8 workgroups per group, all on XCD 0
      double us[2] = {0, 0};
      float got[2] = {0, 0};
      for (int form = 0; form < 2; ++form) {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipMemsetAsync(a.counters, 0, S * sizeof(unsigned), st));
          CK(hipEventRecord(e0, st));
          if (form == 0) hipLaunchKernelGGL((flag_xcd_kernel<12, 0>), dim3(16 * a.p), dim3(NT), 0, st, a);
          else hipLaunchKernelGGL((flag_xcd_kernel<12, 1>), dim3(16 * a.p), dim3(NT), 0, st, a);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms * 1e3 / S < best) best = ms * 1e3 / S;
        }
        us[form] = best;
        CK(hipMemcpy(&got[form], a.plain[(S - 1) & 1], sizeof(float), hipMemcpyDeviceToHost));
      }
      printf("%s{\"n_values\": %d, \"workgroups_per_group\": %d, \"agent_scope_us_per_stage\": %.3f, \"local_us_per_stage\": %.3f, \"final_agent\": %.0f, \"final_local\": %.0f}",
             first ? "" : ", ", a.n, a.p, us[0], us[1], got[0], got[1]);
      first = false;
      fflush(stdout);
    }
  printf("], \"fail\": %d}\n", *fail);
  return 0;
}
