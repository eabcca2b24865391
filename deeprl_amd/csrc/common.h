// Shared helpers for the deeprl_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/deeprl_amd.h"  // exported signatures are checked against the public header

#define DRA_OK 0
#define DRA_EINVAL (-22)
#define DRA_ENOMEM (-12)
#define DRA_ETIMEDOUT (-110)   // a bounded device-side wait gave up (late_step arrival slots, actor hand-over): the results are invalid

// Every export returns 0 or an error code; nothing throws or aborts across the C ABI.
#define DRA_HIP(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return (int)_e;              \
  } while (0)

#define DRA_LAUNCH_CHECK()                             \
  do {                                                 \
    hipError_t _e = hipGetLastError();                 \
    if (_e != hipSuccess) return (int)_e;              \
  } while (0)

#define DRA_API extern "C" __attribute__((visibility("default")))

static inline hipStream_t dra_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- one RMSprop element (torch.optim.RMSprop, centered or not; g already the raw gradient, coef the clip coefficient) -- the
// ONE statement of the arithmetic for optim.hip's step kernels and for the riders below
__device__ __forceinline__ void rmsprop_elem(float& p, float g, float& s, float& a, float coef, float alpha, float oma,
                                             float lr, float eps, int centered) {
  const float gk = g * coef;
  s = s * alpha + oma * gk * gk;
  float avg;
  if (centered) {
    a = a * alpha + oma * gk;
    avg = sqrtf(s - a * a) + eps;
  } else {
    avg = sqrtf(s) + eps;
  }
  p = p - lr * (gk / avg);
}

// ---- DRA_VAR_DEFER_FC4 (round 6): the fc4 segment of the DQN learner's optimizer step, deferred ------------------------------
// fc4's weights are 95 % of the parameters: 51 of the 54 MB the optimizer launch moves, on the update's critical path for ~8 of
// its ~15 us.  Nothing reads them before fc4's forward of the NEXT update (the fourth launch of its graph, ~31 us in) -- and
// the actor reads its copy of them ~16 us after that graph starts.  So the optimizer launch steps everything BUT that segment,
// leaves the clip coefficient in device memory and raises `pending`; the segment is stepped by RIDER workgroups -- extra
// z-slices of the next update's conv1 / conv2 forward launches (conv_v2.hip), memory-bound work beside latency-bound work --
// with the arithmetic above, hence the same bits.  The launch after the riders' lowers `pending` and marks the actor copy the
// riders completed as valid (the actor's fc4 waits for that word before it requests the weights).  Anything that reads the
// parameters outside the pipelined graphs flushes first (learner.hip flush_fc4): the same rider code as a launch of its own.
struct DraFc4Rider {
  float *p, *g, *s1, *s2, *p_copy;    // flat buffers (p_copy: the actor copy the deferred step also has to reach; may be null)
  int64_t begin4, count4;             // the segment, in float4 units
  const float* coef;                  // clip coefficient left by the optimizer launch
  const int* pending;                 // != 0: the segment has not been stepped yet
  float lr, alpha, eps;
  int centered;
};
// (DRA_EXP_RIDER_NT=1: gradient / optimizer-state / copy traffic of the riders as non-temporal accesses -- an A/B build)
#ifndef DRA_EXP_RIDER_NT
#define DRA_EXP_RIDER_NT 0
#endif
constexpr int kRiderNV = 3;           // float4 per thread of a rider workgroup (256 threads)
__host__ __device__ inline int fc4_rider_blocks(int64_t count4) { return (int)((count4 + 256 * kRiderNV - 1) / (256 * kRiderNV)); }

// rider workgroup `rb` of `nrb` (nrb * 256 * kRiderNV >= count4).  COPY_WT: the actor copy is written THROUGH (agent-scope stores) --
// for riders whose completion is announced from inside their own launch (the forward chain: no launch boundary writes the L2 back
// before the actor, on another XCD, is told that the copy is complete)
template <bool COPY_WT = false>
__device__ __forceinline__ void fc4_rider_run(const DraFc4Rider& r, int rb) {
  if (*r.pending == 0) return;
  const int64_t i0 = (int64_t)rb * (256 * kRiderNV) + threadIdx.x;
  float4 P[kRiderNV], G[kRiderNV], S[kRiderNV], A[kRiderNV];
  const float4* p4 = reinterpret_cast<const float4*>(r.p) + r.begin4;
  const float4* g4 = reinterpret_cast<const float4*>(r.g) + r.begin4;
  const float4* s4 = reinterpret_cast<const float4*>(r.s1) + r.begin4;
  const float4* a4 = reinterpret_cast<const float4*>(r.centered ? r.s2 : r.s1) + r.begin4;
#pragma unroll
  for (int v = 0; v < kRiderNV; ++v) {
    const int64_t i = i0 + 256 * v, ic = i < r.count4 ? i : r.count4 - 1;
#if DRA_EXP_RIDER_NT
    P[v] = p4[ic];
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 gq = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(g4 + ic));
    const nt_f4 sq = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(s4 + ic));
    const nt_f4 aq = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(a4 + ic));
    G[v] = make_float4(gq.x, gq.y, gq.z, gq.w); S[v] = make_float4(sq.x, sq.y, sq.z, sq.w); A[v] = make_float4(aq.x, aq.y, aq.z, aq.w);
#else
    P[v] = p4[ic]; G[v] = g4[ic]; S[v] = s4[ic]; A[v] = a4[ic];
#endif
  }
  const float coef = *r.coef, oma = 1.f - r.alpha;
#pragma unroll
  for (int v = 0; v < kRiderNV; ++v) {
    const int64_t i = i0 + 256 * v;
    if (i < r.count4) {
      float* pp = &P[v].x; const float* gg = &G[v].x; float* ss = &S[v].x; float* aa = &A[v].x;
#pragma unroll
      for (int k = 0; k < 4; ++k) rmsprop_elem(pp[k], gg[k], ss[k], aa[k], coef, r.alpha, oma, r.lr, r.eps, r.centered);
      reinterpret_cast<float4*>(r.p)[r.begin4 + i] = P[v];
#if DRA_EXP_RIDER_NT
      typedef float nt_f4 __attribute__((ext_vector_type(4)));
      const nt_f4 pq = {P[v].x, P[v].y, P[v].z, P[v].w}, sq = {S[v].x, S[v].y, S[v].z, S[v].w}, aq = {A[v].x, A[v].y, A[v].z, A[v].w};
      if (r.p_copy) __builtin_nontemporal_store(pq, reinterpret_cast<nt_f4*>(r.p_copy) + r.begin4 + i);
      __builtin_nontemporal_store(sq, reinterpret_cast<nt_f4*>(r.s1) + r.begin4 + i);
      if (r.centered) __builtin_nontemporal_store(aq, reinterpret_cast<nt_f4*>(r.s2) + r.begin4 + i);
#else
      if (r.p_copy) {
        if constexpr (COPY_WT) {
          float* pc = r.p_copy + 4 * (r.begin4 + i);
#pragma unroll
          for (int k = 0; k < 4; ++k) __hip_atomic_store(pc + k, pp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          reinterpret_cast<float4*>(r.p_copy)[r.begin4 + i] = P[v];
        }
      }
      reinterpret_cast<float4*>(r.s1)[r.begin4 + i] = S[v];
      if (r.centered) reinterpret_cast<float4*>(r.s2)[r.begin4 + i] = A[v];
#endif
    }
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: remember, per device ordinal, the largest
// size already granted (a process that launches on a second GPU after a first -- select_device -- must set it there too;
// ADVICE r5).  One object per kernel instantiation (a function-local static at the launch site).
struct DraLdsAttr {
  size_t granted[32] = {};
};
static inline int dra_grant_lds(DraLdsAttr& a, const void* kernel, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
  if (bytes > a.granted[dev]) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    a.granted[dev] = bytes;
  }
  return DRA_OK;
}

constexpr int kWave = 64;  // CDNA4 wavefront

// 64-lane butterfly reductions (wave64: offsets up to 32).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- cross-workgroup hand-over inside ONE launch (the actor's conv3 + fc4 launch, the update's chained launches) ----------------
struct MegaSync {
  unsigned* done = nullptr;        // this role's arrival counter (producers), null = nothing to publish
  const unsigned* wait = nullptr;  // the counter this role waits on (consumers), null = nothing to wait for
  unsigned wait_target = 0;
  int* timeout_flag = nullptr;     // pinned host int
  // optional: counters that are never reset -- the wait is for (*epoch + 1) * wait_target arrivals, `epoch` = launches of this
  // kind completed so far (a LATER launch of the same step bumps it: the update's head kernel for the forward chain)
  const unsigned* epoch = nullptr;
  unsigned epoch_bias = 1;         // (0: the bump has already happened when the waiting launch runs -- the backward chain)
  int slow = 0;                    // != 0: poll every ~0.4 us (hundreds of waiting workgroups on ONE counter: the slab folds)
};
constexpr unsigned long long kMegaWaitTicks = 5000000ull;   // 50 ms of s_memrealtime (100 MHz)

__device__ __forceinline__ void mega_publish(const MegaSync& ms) {
  if (!ms.done) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ms.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mega_wait(const MegaSync& ms) {
  if (!ms.wait) return;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    const unsigned target = ms.epoch ? (*ms.epoch + ms.epoch_bias) * ms.wait_target : ms.wait_target;
    while (__hip_atomic_load(ms.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (ms.slow) __builtin_amdgcn_s_sleep(16);
      else __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > kMegaWaitTicks) {
        if (ms.timeout_flag) __hip_atomic_store(ms.timeout_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
}
template <bool COH> __device__ __forceinline__ float mega_ld(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH> __device__ __forceinline__ void mega_st(float* p, float v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}


// 64- / 128-bit agent-scope loads for float4-shaped staging (two 8-byte relaxed atomic loads: the compiler tracks them)
typedef float dra_f4 __attribute__((ext_vector_type(4)));
template <bool COH> __device__ __forceinline__ dra_f4 mega_ld4(const dra_f4* p) {
  if constexpr (COH) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dra_f4 v;
    v.x = __uint_as_float((unsigned)a); v.y = __uint_as_float((unsigned)(a >> 32));
    v.z = __uint_as_float((unsigned)b); v.w = __uint_as_float((unsigned)(b >> 32));
    return v;
  } else {
    return *p;
  }
}

// What a ROLE of a chained launch (oneshot.h / oneshot_lin.h roles under fused.hip's bwd_chain_kernel) needs to know about its
// place in the chain: the counters its workgroups wait on / count themselves on, one 128-byte line per sample (stride in
// unsigned; 0 = ONE counter for the whole role: the slab folds wait for every weight-gradient workgroup of a layer).
constexpr int kChainLine = 32;    // unsigned per counter line
struct ChainHook {
  const unsigned* wait = nullptr;
  unsigned wait_target = 0;       // arrivals per launch on one counter
  int wait_stride = 0;
  unsigned* done = nullptr;
  int done_stride = 0;
  const unsigned* epoch = nullptr;
  unsigned epoch_bias = 0;
  int slow = 0;
  int* timeout_flag = nullptr;
  __device__ __forceinline__ MegaSync sync(int sample) const {
    MegaSync ms;
    if (wait) { ms.wait = wait + (int64_t)sample * wait_stride; ms.wait_target = wait_target; }
    if (done) ms.done = done + (int64_t)sample * done_stride;
    ms.epoch = epoch; ms.epoch_bias = epoch_bias; ms.timeout_flag = timeout_flag; ms.slow = slow;
    return ms;
  }
};

// ------------------------------------------------------------------------------------------------
// Categorical(logits = x[0..A)) of one sample (network_heads.py:249-254): log-softmax, entropy, and -- when no action is given --
// the inverse-CDF draw from one uniform (first a with cumsum(p)[a] > u; the last action absorbs rounding).  ONE statement shared
// by categorical_fwd_kernel (losses.hip) and the fused rollout head (igemm.hip: policy_heads_sample_kernel), so that both give
// the same bits for the same logits.
__device__ __forceinline__ void categorical_row(const float* x, int A, bool given, int64_t action_in, float ub, int64_t* act_out,
                                                float* log_pi_a, float* entropy) {
  float m = x[0];
  for (int a = 1; a < A; ++a) m = fmaxf(m, x[a]);
  float se = 0.f;
  for (int a = 0; a < A; ++a) se += expf(x[a] - m);
  const float lse = m + logf(se);
  int64_t act = given ? action_in : (int64_t)(A - 1);
  float ent = 0.f, cum = 0.f;
  bool found = given;
  for (int a = 0; a < A; ++a) {
    const float lp = x[a] - lse, p = expf(lp);
    ent -= p * lp;
    cum += p;
    if (!found && cum > ub) { act = a; found = true; }
  }
  if (act < 0) act = 0;
  if (act >= A) act = A - 1;
  *act_out = act;
  *log_pi_a = x[act] - lse;
  *entropy = ent;
}

// Backward of the same: dlogits[a] = g_lp (1[a == action] - p[a]) - g_ent p[a] (logp[a] + entropy).  categorical_row_stats gives
// the row's log-sum-exp and entropy (the sums in ascending a, as the forward forms them); categorical_dlogit one component.
// Shared by categorical_bwd_kernel (losses.hip) and the fused head backward (igemm.hip: policy_heads_bwd_kernel).
__device__ __forceinline__ void categorical_row_stats(const float* x, int A, float* lse_out, float* ent_out) {
  float m = x[0];
  for (int a = 1; a < A; ++a) m = fmaxf(m, x[a]);
  float se = 0.f;
  for (int a = 0; a < A; ++a) se += expf(x[a] - m);
  const float lse = m + logf(se);
  float ent = 0.f;
  for (int a = 0; a < A; ++a) { const float lp = x[a] - lse; ent -= expf(lp) * lp; }
  *lse_out = lse;
  *ent_out = ent;
}
__device__ __forceinline__ float categorical_dlogit(float xa, float lse, float ent, bool is_action, float gl, float ge) {
  const float lp = xa - lse, p = expf(lp);
  return gl * ((is_action ? 1.f : 0.f) - p) - ge * p * (lp + ent);
}

// ------------------------------------------------------------------------------------------------
// XCD-aware workgroup order.  MI355X dispatches the workgroups of a launch round-robin over its 8 XCDs (blockIdx mod 8;
// wgs_per_xcd in every phase trace under profiles/) and every XCD has its own L2: with the natural (sample-major) block
// order the 8-13 workgroups that share one sample's operands land on 8 different XCDs and every L2 fetches that sample
// from the fabric -- conv2's backward launch FETCHED 14.2 MB for 2.4 MB of distinct inputs (profiles/r03e_pmc_traffic_*).
// xcd_order() renumbers the workgroups so that a whole sharing group (all workgroups of one sample / one K slice) runs
// on ONE XCD: workgroup `bid` (first = index of the role's first workgroup in the launch) works on the returned
// virtual index of the natural order.  n_groups groups of per_group workgroups; groups beyond a multiple of 8 keep the
// natural order.  A bijection on [0, n_groups * per_group).
constexpr int kXcds = 8;
__device__ __forceinline__ int xcd_order(int bid, int first, int n_groups, int per_group) {
  const int gpx = n_groups / kXcds;                 // whole groups per XCD
  const int main = gpx * kXcds * per_group;
  if (bid >= main) return bid;
  const int xcd = (bid + first) & (kXcds - 1), j = bid >> 3;      // j < gpx * per_group
  return (xcd * gpx + j / per_group) * per_group + (j % per_group);
}
// (the DRA_XCD_ORDER=0 A/B switch of round 4 is retired: the XCD-aware order is the only one)
static constexpr int dra_xcd_order_enabled() { return 1; }

// ------------------------------------------------------------------------------------------------
// Phase trace (measurement aid, compiled only into libdeeprl_amd_trace.so: `make trace`, -DDRA_TRACE).
// Thread 0 of every workgroup writes a constant-rate (s_memrealtime) time stamp at up to 8 phase
// boundaries into a per-kernel-family region of a trace buffer; tools/phase_trace.py turns the
// stamps into per-phase durations, workgroup start distributions (rounds) and the XCD / CU each
// workgroup ran on.  The product library has no stamps at all (the macros expand to nothing).
enum { TR_GATHER, TR_CONV1_F, TR_CONV2_F, TR_CONV3_F, TR_FC4_F, TR_HEAD, TR_FC_B, TR_CONV3_B, TR_CONV2_B,
       TR_CONV1_B, TR_NORM, TR_STEP, TR_A_CONV1, TR_A_CONV2, TR_A_CONV3, TR_A_FC4, TR_A_HEAD, TR_REGIONS };
constexpr int kTraceWgs = 4096;   // workgroups recorded per region
#ifdef DRA_TRACE
static __device__ unsigned long long* g_dra_trace = nullptr;   // one copy per translation unit
__device__ __forceinline__ void dra_stamp(int region, int k) {
  unsigned long long* p = g_dra_trace;
  if (p && threadIdx.x == 0) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (wg < (unsigned)kTraceWgs) {
      unsigned long long* rec = p + ((size_t)region * kTraceWgs + wg) * 8;
      if (k == 7) {   // last slot also carries where the workgroup ran: [63:48] XCC id, [47:32] HW_ID low bits
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        rec[6] = ((unsigned long long)(xcc & 0xffffu) << 32) | (hw & 0xffffffffu);
      }
      rec[k == 7 ? 7 : k] = t;
    }
  }
}
// every pending vector-memory operation of the calling wave has completed
__device__ __forceinline__ void dra_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
static int dra_trace_set_local(void* p) {
  unsigned long long* v = reinterpret_cast<unsigned long long*>(p);
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_dra_trace), &v, sizeof(v));
  return (int)e;
}
#define DRA_STAMP(region, k) dra_stamp((region), (k))
#define DRA_STAMP_END(region) do { dra_drain(); dra_stamp((region), 7); } while (0)
#else
#define DRA_STAMP(region, k) ((void)0)
#define DRA_STAMP_END(region) ((void)0)
#endif
