// K11: global-norm gradient clipping fused with the optimiser step over ONE flat fp32 buffer.
// Replaces nn.utils.clip_grad_norm_ + torch.optim.{RMSprop,Adam}.step() at
// deep_rl/agent/DQN_agent.py:130-134 (optimisers configured at examples.py:67-68,139,204,370,
// 508-509,534): ~100 tiny ATen ops over 10 tensors become two launches.
//
//   launch 1  dra_grad_sqnorm : per-workgroup partial sums of g^2 (fp32 per lane, fp64 across
//                               lanes) -> partials[nblocks]; optionally folds split-K slabs of
//                               the weight-gradient GEMMs into the final gradient on the way.
//   launch 2  dra_*_step      : every workgroup re-reduces the partials in a fixed order (so
//                               the result is run-to-run deterministic), forms
//                               coef = max_norm / (norm + 1e-6), and applies the update.
// HBM-bound: centered RMSprop reads p,g,sq,ga and writes p,sq,ga (+ the norm read) = 32 B/param.
#include "common.h"
#include <stdlib.h>
#include <string.h>

constexpr int kNormBlocks = 512;  // fixed so graphs replay the same reduction tree

__global__ void __launch_bounds__(256)
grad_sqnorm_kernel(float* __restrict__ grad, int64_t n, const float* __restrict__ slabs, int n_slabs,
                   int64_t slab_stride, double* __restrict__ partials) {
  __shared__ double s_red[4];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  float4* g4 = reinterpret_cast<float4*>(grad);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 g = g4[i];
    if (n_slabs > 0) {  // fold split-K slabs: grad = sum_s slab[s]  (fixed order)
      g = reinterpret_cast<const float4*>(slabs)[i];
      for (int s = 1; s < n_slabs; ++s) {
        const float4 t = reinterpret_cast<const float4*>(slabs + (int64_t)s * slab_stride)[i];
        g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
      }
      g4[i] = g;
    }
    acc += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float g = grad[i];
    if (n_slabs > 0) {
      g = slabs[i];
      for (int s = 1; s < n_slabs; ++s) g += slabs[(int64_t)s * slab_stride + i];
      grad[i] = g;
    }
    acc += g * g;
  }
  double d = wave_sum((double)acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// Each dra_grad_sqnorm call writes dra_norm_partials() doubles.  Several gradient segments ->
// one global norm: give each call its own run of partials and pass the total count to the step.
DRA_API int dra_norm_partials(void) { return kNormBlocks; }

DRA_API int dra_grad_sqnorm(float* grad, int64_t n, const float* slabs, int n_slabs, int64_t slab_stride,
                            double* partials, void* stream) {
  if (!grad || !partials || n < 1 || n_slabs < 0 || (n_slabs > 0 && !slabs)) return DRA_EINVAL;
  if ((((uintptr_t)grad) & 15) || (slabs && ((((uintptr_t)slabs) & 15) || (slab_stride & 3)))) return DRA_EINVAL;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(kNormBlocks), dim3(256), 0, dra_stream(stream), grad, n, slabs, n_slabs,
                     slab_stride, partials);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Fixed-order reduction of the partials (written by an EARLIER launch) by every workgroup; returns the clip coefficient.
__device__ __forceinline__ float clip_coef_from_partials(const double* __restrict__ partials, int n_partials, float max_norm,
                                                         float* __restrict__ out_norm) {
  if (!partials) return 1.f;  // uniform: no clipping requested
  __shared__ double s_part[4];
  __shared__ float s_coef;
  // n_partials <= dra_norm_partials_max() = 4096 = 16 per thread: straight-line loads (a loop here makes the
  // compiler drain every outstanding load of the caller first); same per-thread summation order as a loop
  double d = 0.0;
  {
    constexpr int NPT = 16;
    double v[NPT];
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
      const int i = (int)threadIdx.x + 256 * u;
      v[u] = partials[i < n_partials ? i : n_partials - 1];
    }
#pragma unroll
    for (int u = 0; u < NPT; ++u) d += ((int)threadIdx.x + 256 * u < n_partials) ? v[u] : 0.0;
  }
  d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
    if (out_norm && blockIdx.x == 0) *out_norm = norm;
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      if (coef > 1.f) coef = 1.f;
    }
    s_coef = coef;
  }
  __syncthreads();
  return s_coef;
}

// torch.optim.RMSprop:  sq = a*sq + (1-a)*g*g ; centered: ga = a*ga + (1-a)*g,
// avg = sqrt(sq - ga*ga) + eps  else  avg = sqrt(sq) + eps ;  p -= lr * g / avg.
// NV float4 per thread, every operand of both requested before the clip coefficient is reduced from the partials
// (the operands do not depend on it): one exposed memory latency, and the grid (n / (4 * 256 * NV) workgroups = 824 for
// the DQN learner) fits the update chain's CU partition in ONE round -- with one float4 per thread the 1647 workgroups
// ran as 1280 + 367 on 160 CUs (phase trace, profiles/r02a_phase_async.json: span 11.0 us for 6.0 us workgroups).
constexpr int kStepNV = 2;

// (rmsprop_elem: common.h -- shared with the deferred fc4 segment's riders)

// ---- segmented fold + norm --------------------------------------------------------------------------------------------
// The one-pass conv weight gradients write one slab per (sample, row chunk): 32-160 slabs per layer.  This launch folds
// them and leaves per-workgroup sums of squares in `partials` (the optimiser is a second launch, dra_*_step).
// (Round 2 also ran the optimiser behind a GRID BARRIER in this launch -- every workgroup keeping its elements in registers,
// one ticket counter: bit-identical and 14 % slower, the 796 tickets serialise at ~30 ns each: DESIGN.md section 4,
// profiles/r02zt_*.  Removed in round 4; the late-fold form below is what replaced the second launch.)
// Workgroup kinds (block ranges [fold blocks of segment 0][segment 1]...[plain blocks]); a fold workgroup is 16 slab
// groups x 16 float4 elements per unit, thread (g, el) sums slabs g, g+16, ... in increasing order, the 16 group
// partials meet in LDS and are added in group order: deterministic, and ONE memory round trip per workgroup:
//   narrow (n_slabs <= 32): 4 units per workgroup, 2 slabs per thread and unit (8 float4 in flight)
//   wide   (n_slabs  > 32): 1 unit per workgroup, 10 slabs per thread and pass (160 slabs = one pass)
//   plain  (no slabs)     : 4 float4 per thread
constexpr int kPlainNV = 4;
constexpr int kWideL = 10;
constexpr int kNarrowU = 4;
constexpr unsigned long long kBarrierTicks = 5000000ull;   // 50 ms of s_memrealtime (100 MHz): a device-side wait that cannot
                                                           // complete reports through the timeout flag instead of hanging

struct FoldPlan {
  int64_t begin4[DRA_MAX_FOLD_SEGS];     // first float4 of the segment in `grad`
  int64_t count4[DRA_MAX_FOLD_SEGS];     // float4s
  const float4* slabs[DRA_MAX_FOLD_SEGS];
  int64_t stride4[DRA_MAX_FOLD_SEGS];
  int32_t n_slabs[DRA_MAX_FOLD_SEGS];
  int32_t first_block[DRA_MAX_FOLD_SEGS + 1];  // block range of each segment; [n_segs] = first plain block
  int32_t n_segs, plain_blocks, plain_iters;   // plain_iters > 1: a plain workgroup walks that many strides
  int64_t plain_begin4, plain_count4;    // [plain_begin4, plain_begin4 + plain_count4): no slabs
};

struct StepHyper {
  float max_norm, lr, a, b2, eps;        // a = alpha (RMSprop) / beta1 (Adam); b2 = beta2 (Adam)
  int centered;
  const int64_t* step_dev;               // Adam: 1-based step count in device memory
};

__device__ __forceinline__ float sq4(const float4& v) { return v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
fold_norm_kernel(float* __restrict__ grad, const FoldPlan fp, double* __restrict__ partials) {
  __shared__ float4 s_part[kNarrowU][16][17];
  __shared__ double s_red[4];
  const int tid = threadIdx.x, bid = blockIdx.x;
  float acc = 0.f;
  float4* __restrict__ g4 = reinterpret_cast<float4*>(grad);
  DRA_STAMP(TR_NORM, 0);
  if (bid < fp.first_block[fp.n_segs]) {
    int sg = 0;
    while (sg + 1 < fp.n_segs && bid >= fp.first_block[sg + 1]) ++sg;
    const int b = bid - fp.first_block[sg];
    const int64_t n4 = fp.count4[sg];
    const float4* __restrict__ sl = fp.slabs[sg];
    const int64_t st4 = fp.stride4[sg];
    const int ns = fp.n_slabs[sg];
    const int g = tid >> 4, el = tid & 15;
    if (ns <= 32) {
      // ---- narrow: units b*4 .. b*4+3; the owner of element (u, el) is thread 16 u + el
      const int64_t e0 = (int64_t)b * (16 * kNarrowU);
      const int64_t io = e0 + tid;                               // owner's element (tid < 64)
      float4 t[kNarrowU][2];
      const bool v0 = g < ns, v1 = g + 16 < ns;
      const int64_t r0 = (int64_t)(v0 ? g : 0) * st4, r1 = (int64_t)(v1 ? g + 16 : 0) * st4;
#pragma unroll
      for (int u = 0; u < kNarrowU; ++u) {
        const int64_t i = e0 + 16 * u + el;
        const int64_t ic = i < n4 ? i : n4 - 1;
        t[u][0] = sl[r0 + ic];
        t[u][1] = sl[r1 + ic];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kNarrowU; ++u) {
        float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v0) add4(pp, t[u][0]);
        if (v1) add4(pp, t[u][1]);
        s_part[u][g][el] = pp;
      }
      __syncthreads();
      if (tid < 16 * kNarrowU) {
        const int u = tid >> 4;
        float4 r = s_part[u][0][el];
#pragma unroll
        for (int q = 1; q < 16; ++q) add4(r, s_part[u][q][el]);
        if (io < n4) {
          g4[fp.begin4[sg] + io] = r;
          acc += sq4(r);
        }
      }
    } else {
      // ---- wide: unit b; the owner of element el is thread el
      const int64_t i = (int64_t)b * 16 + el;
      const int64_t ic = i < n4 ? i : n4 - 1;
      float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s0 = g; s0 < ns; s0 += 16 * kWideL) {
        float4 t[kWideL];
#pragma unroll
        for (int u = 0; u < kWideL; ++u) {
          const int s = s0 + 16 * u;
          t[u] = sl[(int64_t)(s < ns ? s : g) * st4 + ic];
        }
#pragma unroll
        for (int u = 0; u < kWideL; ++u)
          if (s0 + 16 * u < ns) add4(pp, t[u]);
      }
      s_part[0][g][el] = pp;
      __syncthreads();
      if (tid < 16) {
        float4 r = s_part[0][0][el];
#pragma unroll
        for (int q = 1; q < 16; ++q) add4(r, s_part[0][q][el]);
        if (i < n4) {
          g4[fp.begin4[sg] + i] = r;
          acc += sq4(r);
        }
      }
    }
  } else {
    // ---- plain: kPlainNV float4 per thread and stride, every operand of a stride requested up front
    const int b = bid - fp.first_block[fp.n_segs];
    for (int it = 0; it < fp.plain_iters; ++it) {
      const int64_t i0 = ((int64_t)it * fp.plain_blocks + b) * (256 * kPlainNV) + tid;
      float4 G[kPlainNV];
#pragma unroll
      for (int v = 0; v < kPlainNV; ++v) {
        const int64_t i = i0 + 256 * v;
        G[v] = g4[fp.plain_begin4 + (i < fp.plain_count4 ? i : fp.plain_count4 - 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < kPlainNV; ++v)
        if (i0 + 256 * v < fp.plain_count4) acc += sq4(G[v]);
    }
  }
  DRA_STAMP(TR_NORM, 4);
  const double d = wave_sum((double)acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = d;
  __syncthreads();
  if (tid == 0) partials[bid] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  DRA_STAMP(TR_NORM, 5);
  DRA_STAMP_END(TR_NORM);
}

DRA_API int dra_norm_partials_max(void) { return 4096; }

static int make_fold_plan(int64_t n, const dra_fold_seg* segs, int n_segs, FoldPlan* out, int* blocks_out) {
  if (n < 4 || (n & 3) || n_segs < 0 || n_segs > DRA_MAX_FOLD_SEGS || (n_segs && !segs)) return DRA_EINVAL;
  FoldPlan fp;
  memset(&fp, 0, sizeof(fp));
  int64_t end = 0;
  int64_t blocks = 0;
  for (int i = 0; i < n_segs; ++i) {
    const dra_fold_seg& sg = segs[i];
    if (sg.begin != end || (sg.begin & 3) || (sg.count & 3) || sg.count < 4 || sg.begin + sg.count > n || !sg.slabs ||
        sg.n_slabs < 1 || (sg.slab_stride & 3) || (((uintptr_t)sg.slabs) & 15))
      return DRA_EINVAL;
    fp.begin4[i] = sg.begin >> 2; fp.count4[i] = sg.count >> 2;
    fp.slabs[i] = reinterpret_cast<const float4*>(sg.slabs); fp.stride4[i] = sg.slab_stride >> 2;
    fp.n_slabs[i] = sg.n_slabs;
    fp.first_block[i] = (int32_t)blocks;
    const int64_t per = sg.n_slabs <= 32 ? 16 * kNarrowU : 16;   // float4 elements per fold workgroup
    blocks += (fp.count4[i] + per - 1) / per;
    end = sg.begin + sg.count;
    if (blocks > 0x7fffffff) return DRA_EINVAL;
  }
  fp.first_block[n_segs] = (int32_t)blocks;
  fp.n_segs = n_segs;
  fp.plain_begin4 = end >> 2; fp.plain_count4 = (n - end) >> 2;
  int64_t pb = (fp.plain_count4 + 256 * kPlainNV - 1) / (256 * kPlainNV);
  fp.plain_iters = 1;
  const int64_t room = (int64_t)dra_norm_partials_max() - blocks;   // every workgroup writes one partial
  if (pb > room) {
    if (room < 1) return DRA_EINVAL;
    fp.plain_iters = (int32_t)((pb + room - 1) / room);
    pb = (pb + fp.plain_iters - 1) / fp.plain_iters;
  }
  fp.plain_blocks = (int32_t)pb;
  blocks += pb;
  if (blocks < 1) return DRA_EINVAL;
  *out = fp;
  *blocks_out = blocks > 0x7fffffff ? 0x7fffffff : (int)blocks;
  return DRA_OK;
}

// grad[0, n): segments (contiguous from 0, 4-float aligned) are folded from their slabs
// (grad[seg] = sum_s slabs[s*stride + i], fixed order) and everything after the last segment is read as
// is; *n_partials doubles are written (<= dra_norm_partials_max()) -- pass that count to dra_*_step.

// Workgroups (= partials written) of dra_grad_sqnorm_segs for this gradient layout: pure host arithmetic.
DRA_API int dra_grad_sqnorm_segs_blocks(int64_t n, const dra_fold_seg* segs, int n_segs, int* blocks) {
  if (!blocks) return DRA_EINVAL;
  FoldPlan fp;
  int rc = make_fold_plan(n, segs, n_segs, &fp, blocks);
  if (rc) return rc;
  return *blocks > dra_norm_partials_max() ? DRA_EINVAL : DRA_OK;
}

DRA_API int dra_grad_sqnorm_segs(float* grad, int64_t n, const dra_fold_seg* segs, int n_segs, double* partials,
                                 int* n_partials, void* stream) {
  if (!grad || !partials || !n_partials || (((uintptr_t)grad) & 15)) return DRA_EINVAL;
  FoldPlan fp;
  int blocks = 0;
  int rc = make_fold_plan(n, segs, n_segs, &fp, &blocks);
  if (rc) return rc;
  if (blocks > dra_norm_partials_max()) return DRA_EINVAL;
  hipLaunchKernelGGL(fold_norm_kernel, dim3(blocks), dim3(256), 0, dra_stream(stream), grad, fp, partials);
  DRA_LAUNCH_CHECK();
  *n_partials = blocks;
  return DRA_OK;
}

// ---- late-fold form (DRA_VAR_LATE_FOLD): the optimizer launch with only the LAST layer's fold in front --------------------
// The gradient-norm pass was a launch of its own on the update's dependent chain (5.8 us + a 1.8 us boundary,
// profiles/r02zzz_phase_async.json) because the conv weight-gradient slabs had to be folded before anything could be
// clipped.  With this form every tensor's sum of squares already exists when the optimizer starts -- the linear layers'
// from the kernels that wrote them (igemm_sumsq, HeadWgradRole), conv3 / conv2 from FoldRole workgroups riding in the next
// layer's backward launch -- EXCEPT the first segment (conv1, whose weight gradient is the last kernel of the backward).
// Its fold_blocks workgroups come first in the grid.  A fold workgroup owns EPB float4 elements: thread (g, el) adds slabs
// g, g + NG, ... in order (every load in flight at once), the NG group sums meet in LDS and are added in group order;
// (EPB, NG) = (64, 4) for <= 64 slabs, (16, 16) for <= 256.  Its sum of squares is PUBLISHED AS THE FLAG: slot
// partials[n_prior + b] holds -1 until the workgroup's one agent-scope store of a (non-negative) sum lands -- no ticket
// counter, no second round trip (the cooperative form's 796 tickets on one counter serialised at ~30 ns each,
// profiles/r02zt_*; round 3's first attempt published + counted, and read ALL partials with agent-scope loads: 19.4 us
// against RMSprop's 12.5, profiles/r03a_*).  EVERY workgroup requests its parameters / optimizer state / gradient and
// the earlier launches' partials (plain cached loads) first; thread t < fold_blocks then polls slot t until it is
// non-negative, the partials are reduced in the fixed order and the step is applied.  Progress: the fold workgroups have
// the lowest block indices, so they are resident before any waiting workgroup; a wait is bounded (50 ms) and reports
// through the pinned timeout flag.  The slots must hold -1 at launch (FoldRole::reset_slots in
// the preceding launch).
struct LatePlan {
  int64_t n;             // floats in the flat buffers (a tail of n % 4 floats is stepped by the last workgroup)
  int64_t n4;            // whole float4s
  int64_t fold_count4;   // the folded segment is [0, fold_count4)
  const float4* slabs;
  int64_t stride4;
  int32_t n_slabs, fold_blocks, n_prior;   // n_prior: partials already written by earlier launches ([0, n_prior))
  // DRA_VAR_DEFER_FC4 (common.h DraFc4Rider): float4s [skip_begin4, skip_begin4 + skip_count4) are NOT stepped by this launch;
  // it leaves the clip coefficient in *defer_coef, raises *defer_pending and clears *defer_valid instead (all null = off)
  int64_t skip_begin4, skip_count4;
  float* defer_coef;
  int* defer_pending;
  int* defer_valid;
};
constexpr int kLateMaxFoldBlocks = 256;    // one polling thread per fold workgroup
constexpr int kLateNV = 3;                 // float4 per thread of a plain workgroup: with the fold workgroups the grid stays
                                           // under the 4-per-CU resident limit of the update's 224 CUs (r03b: 949 workgroups at
                                           // 2 float4 ran 896 + 53 and took 17 us)
template <int NG> struct LateFold { static constexpr int SPT = NG == 4 ? 16 : 10; };   // slabs per thread: <= 64 / <= 160 slabs

// fixed-order reduction: thread t sums prior partials t, t + 256, ... (plain loads, requested by the caller BEFORE the wait:
// `pre`), then the fold workgroups' published sums (thread t < fold_blocks waits for slot t); lanes by butterfly, waves
// (w0 + w1) + (w2 + w3).
__device__ __forceinline__ float late_clip_coef(double pre, double* __restrict__ partials, int n_prior, int fold_blocks,
                                                float max_norm, float* __restrict__ out_norm, int* __restrict__ timeout_flag) {
  __shared__ double s_part[4];
  __shared__ float s_coef;
  double d = pre;
  if ((int)threadIdx.x < fold_blocks) {
    const double* slot = partials + n_prior + threadIdx.x;
    double v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < 0.0) {
      const unsigned long long t0 = wall_clock64();
      do {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wall_clock64() - t0 > kBarrierTicks) {
          __hip_atomic_store(timeout_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          v = 0.0;
        }
      } while (v < 0.0);
    }
    d += v;
  }
  d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
    if (out_norm && blockIdx.x == 0) *out_norm = norm;
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      if (coef > 1.f) coef = 1.f;
    }
    s_coef = coef;
  }
  __syncthreads();
  return s_coef;
}

// OPT: 0 = RMSprop, 1 = Adam;  NG: slab groups of a fold workgroup (4 or 16);  NPT: earlier partials per thread (4: <= 1024)
template <int OPT, int NG, int NPT>
__global__ void __launch_bounds__(256)
late_step_kernel(float* __restrict__ grad, const LatePlan lp, double* __restrict__ partials, float* __restrict__ p,
                 float* __restrict__ s1, float* __restrict__ s2, float* __restrict__ p_copy, const StepHyper hp,
                 float* __restrict__ out_norm, int* __restrict__ timeout_flag) {
  constexpr int EPB = 256 / NG;            // float4 elements per fold workgroup
  constexpr int SPT = LateFold<NG>::SPT;
  constexpr int kStepNV = kLateNV;         // (shadows the two-launch kernels' constant)
  __shared__ float4 s_fold[NG][EPB + 1];
  __shared__ float s_hyper[2];
  const int tid = threadIdx.x, bid = blockIdx.x;
  float4* __restrict__ g4 = reinterpret_cast<float4*>(grad);
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
  const float4* __restrict__ s14 = reinterpret_cast<const float4*>(s1);
  const float4* __restrict__ s24 = reinterpret_cast<const float4*>((OPT == 0 && !hp.centered) ? s1 : s2);
  int64_t gi[kStepNV];
  float4 G[kStepNV], P[kStepNV], S[kStepNV], A[kStepNV];
#pragma unroll
  for (int v = 0; v < kStepNV; ++v) gi[v] = -1;
  DRA_STAMP(TR_STEP, 0);
  // the earlier launches' partials: plain loads, in flight with everything else
  double pv[NPT];
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    const int i = tid + 256 * u;
    pv[u] = partials[i < lp.n_prior ? i : (lp.n_prior > 0 ? lp.n_prior - 1 : 0)];
  }
  if (bid < lp.fold_blocks) {
    const int g = tid / EPB, el = tid % EPB;
    const int64_t i = (int64_t)bid * EPB + el;
    const int64_t ic = i < lp.fold_count4 ? i : lp.fold_count4 - 1;
    if (tid < EPB) { P[0] = p4[ic]; S[0] = s14[ic]; A[0] = s24[ic]; }
    const int ns = lp.n_slabs;
    float4 t[SPT];
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
      const int s = g + NG * u;
      t[u] = lp.slabs[(int64_t)(s < ns ? s : 0) * lp.stride4 + ic];
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < SPT; ++u)
      if (g + NG * u < ns) add4(pp, t[u]);
    s_fold[g][el] = pp;
    __syncthreads();
    float acc = 0.f;
    if (tid < EPB) {
      float4 a = s_fold[0][el];
#pragma unroll
      for (int q = 1; q < NG; ++q) add4(a, s_fold[q][el]);
      if (i < lp.fold_count4) {
        g4[i] = a;
        acc = sq4(a);
        G[0] = a;
        gi[0] = i;
      }
    }
    // (EPB <= 64: the owners are lanes of wave 0; the other waves hold zeros)
    const double d = wave_sum((double)acc);
    if (tid == 0) __hip_atomic_store(partials + lp.n_prior + bid, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const int64_t i0 = lp.fold_count4 + (int64_t)(bid - lp.fold_blocks) * (256 * kStepNV) + tid;
#pragma unroll
    for (int v = 0; v < kStepNV; ++v) {
      int64_t i = i0 + 256 * v;
      if (i >= lp.skip_begin4) i += lp.skip_count4;       // (the deferred segment: skip_count4 = 0 when nothing is deferred)
      const int64_t ic = i < lp.n4 ? i : lp.n4 - 1;
      P[v] = p4[ic]; G[v] = g4[ic]; S[v] = s14[ic]; A[v] = s24[ic];
      if (i < lp.n4) gi[v] = i;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (OPT == 1 && tid == 64) {   // Adam's bias corrections from the device step count, exactly as dra_adam_hyper forms them
    const double t = (double)*hp.step_dev;
    const double bc1 = 1.0 - pow((double)hp.a, t), bc2 = 1.0 - pow((double)hp.b2, t);
    s_hyper[0] = (float)((double)hp.lr / bc1);
    s_hyper[1] = (float)(1.0 / sqrt(bc2));
  }
  double pre = 0.0;
#pragma unroll
  for (int u = 0; u < NPT; ++u) pre += (tid + 256 * u < lp.n_prior) ? pv[u] : 0.0;
  DRA_STAMP(TR_STEP, 2);
  const float coef = late_clip_coef(pre, partials, lp.n_prior, lp.fold_blocks, hp.max_norm, out_norm, timeout_flag);
  if (lp.defer_coef && bid == 0 && tid == 0) {   // hand the deferred segment over: coefficient, pending, the copy it will complete
    *lp.defer_coef = coef;
    __hip_atomic_store(lp.defer_pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(lp.defer_valid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  [[maybe_unused]] const float oma = 1.f - hp.a, omb2 = 1.f - hp.b2;
  [[maybe_unused]] const float step_size = s_hyper[0], inv_sqrt_bc2 = s_hyper[1];
  // (Round 4 measured a SECOND run per plain workgroup whose loads were issued here, behind the wait for the coefficient, to
  // overlap reads with the first run's stores: 16.4 -> 17.6 us.  The loads of the single run already travel under the wait for
  // conv1's fold, which is what bounds the first half of this launch; profiles/r04d_ab_env.jsonl.)
  auto step_run = [&](float4* Pv, const float4* Gv, float4* Sv, float4* Av, const int64_t* giv) {
#pragma unroll
    for (int v = 0; v < kStepNV; ++v) {
      if (giv[v] >= 0) {
        float* pp = &Pv[v].x; const float* gg = &Gv[v].x; float* ss = &Sv[v].x; float* aa = &Av[v].x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (OPT == 0) {
            rmsprop_elem(pp[k], gg[k], ss[k], aa[k], coef, hp.a, oma, hp.lr, hp.eps, hp.centered);
          } else {
            const float gk = gg[k] * coef;
            ss[k] = ss[k] * hp.a + oma * gk;                       // exp_avg
            aa[k] = aa[k] * hp.b2 + omb2 * gk * gk;                // exp_avg_sq
            pp[k] = pp[k] - step_size * (ss[k] / (sqrtf(aa[k]) * inv_sqrt_bc2 + hp.eps));
          }
        }
        reinterpret_cast<float4*>(p)[giv[v]] = Pv[v];
        if (p_copy) reinterpret_cast<float4*>(p_copy)[giv[v]] = Pv[v];
        reinterpret_cast<float4*>(s1)[giv[v]] = Sv[v];
        if (OPT == 1 || hp.centered) reinterpret_cast<float4*>(s2)[giv[v]] = Av[v];
      }
    }
  };
  step_run(P, G, S, A, gi);
  // tail (n not a multiple of 4): the last few floats, by the first threads of the last workgroup
  const int64_t tl = (lp.n4 << 2) + tid;
  if (bid == (int)gridDim.x - 1 && tl < lp.n) {
    float pv1 = p[tl], sv = s1[tl], av = (OPT == 1 || hp.centered) ? s2[tl] : 0.f;
    if (OPT == 0) {
      rmsprop_elem(pv1, grad[tl], sv, av, coef, hp.a, oma, hp.lr, hp.eps, hp.centered);
    } else {
      const float gk = grad[tl] * coef;
      sv = sv * hp.a + oma * gk;
      av = av * hp.b2 + omb2 * gk * gk;
      pv1 = pv1 - step_size * (sv / (sqrtf(av) * inv_sqrt_bc2 + hp.eps));
    }
    p[tl] = pv1;
    if (p_copy) p_copy[tl] = pv1;
    s1[tl] = sv;
    if (OPT == 1 || hp.centered) s2[tl] = av;
  }
  DRA_STAMP(TR_STEP, 5);
  DRA_STAMP_END(TR_STEP);
}

static int late_groups(int n_slabs) { return n_slabs <= 4 * LateFold<4>::SPT ? 4 : 16; }

// Workgroups of the late-fold launch that fold (and publish a partial): 64 float4 each for <= 64 slabs, 16 for <= 160.
DRA_API int dra_clip_step_late_blocks(const dra_fold_seg* seg, int* fold_blocks) {
  if (!seg || !fold_blocks || seg->count < 4 || (seg->count & 3) || seg->n_slabs < 1 || seg->n_slabs > 16 * LateFold<16>::SPT)
    return DRA_EINVAL;
  const int epb = 256 / late_groups(seg->n_slabs);
  const int64_t nb = ((seg->count >> 2) + epb - 1) / epb;
  if (nb > kLateMaxFoldBlocks) return DRA_EINVAL;
  *fold_blocks = (int)nb;
  return DRA_OK;
}

// seg: the ONE segment still in slabs (must start at element 0 of the flat gradient, n_slabs <= 256, at most 256 fold
// workgroups); partials[0, n_prior) were written by earlier launches, partials[n_prior, n_prior + fold_blocks) must hold -1.0
// at launch and receive this launch's published sums; timeout_flag: pinned host int.  optimizer: DRA_OPT_RMSPROP (hyper = {lr, alpha, eps, -}) or DRA_OPT_ADAM ({lr, beta1, eps, beta2}, step count read from step_dev).
static int clip_step_late_impl(float* param, float* grad, float* state1, float* state2, int64_t n, const dra_fold_seg* seg,
                               double* partials, int n_prior, int* timeout_flag, int optimizer, float max_norm,
                               const float* hyper, int centered, const int64_t* step_dev, float* out_norm, float* param_copy,
                               int64_t skip_begin, int64_t skip_count, float* defer_coef, int* defer_pending, int* defer_valid,
                               void* stream) {
  if (!param || !grad || !state1 || !seg || !partials || !timeout_flag || !hyper) return DRA_EINVAL;
  if (optimizer != DRA_OPT_RMSPROP && optimizer != DRA_OPT_ADAM) return DRA_EINVAL;
  if ((optimizer == DRA_OPT_ADAM && (!state2 || !step_dev)) || (optimizer == DRA_OPT_RMSPROP && centered && !state2)) return DRA_EINVAL;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)state1) | ((uintptr_t)state2) | ((uintptr_t)param_copy) |
       ((uintptr_t)seg->slabs)) & 15)
    return DRA_EINVAL;
  if (n < 4 || seg->begin != 0 || seg->count > n || !seg->slabs || (seg->slab_stride & 3) || n_prior < 0) return DRA_EINVAL;
  int fold_blocks = 0;
  int rc = dra_clip_step_late_blocks(seg, &fold_blocks);
  if (rc) return rc;
  LatePlan lp;
  memset(&lp, 0, sizeof(lp));
  lp.n = n; lp.n4 = n >> 2; lp.fold_count4 = seg->count >> 2; lp.slabs = reinterpret_cast<const float4*>(seg->slabs);
  lp.stride4 = seg->slab_stride >> 2; lp.n_slabs = seg->n_slabs; lp.n_prior = n_prior;
  lp.fold_blocks = fold_blocks;
  if (lp.n_prior + lp.fold_blocks > dra_norm_partials_max()) return DRA_EINVAL;
  lp.skip_begin4 = lp.n4; lp.skip_count4 = 0;        // (nothing skipped: no index reaches n4)
  if (skip_count > 0) {
    if ((skip_begin & 3) || (skip_count & 3) || skip_begin < seg->count || skip_begin + skip_count > (n & ~(int64_t)3) || !defer_coef ||
        !defer_pending || !defer_valid || optimizer != DRA_OPT_RMSPROP)
      return DRA_EINVAL;
    lp.skip_begin4 = skip_begin >> 2; lp.skip_count4 = skip_count >> 2;
    lp.defer_coef = defer_coef; lp.defer_pending = defer_pending; lp.defer_valid = defer_valid;
  }
  const int64_t plain = (lp.n4 - lp.fold_count4 - lp.skip_count4 + 256 * kLateNV - 1) / (256 * kLateNV);
  const int64_t blocks = lp.fold_blocks + plain;
  if (blocks > 0x7fffffff) return DRA_EINVAL;
  StepHyper hp;
  memset(&hp, 0, sizeof(hp));
  hp.max_norm = max_norm; hp.lr = hyper[0]; hp.a = hyper[1]; hp.eps = hyper[2]; hp.b2 = hyper[3];
  hp.centered = centered; hp.step_dev = step_dev;
  const bool adam = optimizer == DRA_OPT_ADAM, wide = late_groups(seg->n_slabs) == 16, many = n_prior > 1024;
#define DRA_LATE_LAUNCH(OPT, NG, NPT)                                                                                          \
  hipLaunchKernelGGL((late_step_kernel<OPT, NG, NPT>), dim3((unsigned)blocks), dim3(256), 0, dra_stream(stream), grad, lp, partials, \
                     param, state1, state2, param_copy, hp, out_norm, timeout_flag)
#define DRA_LATE_NG(OPT, NPT) do { if (wide) DRA_LATE_LAUNCH(OPT, 16, NPT); else DRA_LATE_LAUNCH(OPT, 4, NPT); } while (0)
  if (adam) { if (many) DRA_LATE_NG(1, 16); else DRA_LATE_NG(1, 4); }
  else { if (many) DRA_LATE_NG(0, 16); else DRA_LATE_NG(0, 4); }
#undef DRA_LATE_NG
#undef DRA_LATE_LAUNCH
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_clip_step_late(float* param, float* grad, float* state1, float* state2, int64_t n, const dra_fold_seg* seg,
                               double* partials, int n_prior, int* timeout_flag, int optimizer, float max_norm,
                               const float* hyper, int centered, const int64_t* step_dev, float* out_norm, float* param_copy,
                               void* stream) {
  return clip_step_late_impl(param, grad, state1, state2, n, seg, partials, n_prior, timeout_flag, optimizer, max_norm, hyper,
                             centered, step_dev, out_norm, param_copy, 0, 0, nullptr, nullptr, nullptr, stream);
}

// Library-internal (common.h DraFc4Rider): the same launch with floats [skip_begin, skip_begin + skip_count) left for the riders
// (RMSprop only; both multiples of 4, behind the folded segment).
int dra_clip_step_late_defer(float* param, float* grad, float* state1, float* state2, int64_t n, const dra_fold_seg* seg,
                             double* partials, int n_prior, int* timeout_flag, float max_norm, const float* hyper, int centered,
                             float* out_norm, float* param_copy, int64_t skip_begin, int64_t skip_count, float* defer_coef,
                             int* defer_pending, int* defer_valid, void* stream) {
  return clip_step_late_impl(param, grad, state1, state2, n, seg, partials, n_prior, timeout_flag, DRA_OPT_RMSPROP, max_norm, hyper,
                             centered, nullptr, out_norm, param_copy, skip_begin, skip_count, defer_coef, defer_pending, defer_valid,
                             stream);
}

// (Round 2 measured non-temporal loads / stores of the gradient and the optimizer state here, meant to keep the actor's parameter
// copy in L2 / MALL: RMSprop 8.8 -> 11.2 us, actor fc4 unchanged -- DESIGN.md section 4; the switch was removed in round 4.)
__global__ void __launch_bounds__(256)
rmsprop_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, float* __restrict__ ga,
                    int64_t n, const double* __restrict__ partials, int n_partials, float max_norm, float lr,
                    float alpha, float eps, int centered, float* __restrict__ out_norm, float* __restrict__ p_copy) {
  const float oma = 1.f - alpha;
  const int64_t n4 = n >> 2;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * kStepNV) + threadIdx.x;
  DRA_STAMP(TR_STEP, 0);
  // (unconditional loads of a clamped index, n >= 4 is checked by the launcher: a branch or a default value here turns
  // the loaded registers into phi copies and the compiler waits for them on the spot)
  float4 P[kStepNV], G[kStepNV], S[kStepNV], A[kStepNV];
#pragma unroll
  for (int v = 0; v < kStepNV; ++v) {
    const int64_t i = i0 + 256 * v;
    const int64_t ic = i < n4 ? i : n4 - 1;
    P[v] = reinterpret_cast<float4*>(p)[ic];
    G[v] = reinterpret_cast<const float4*>(g)[ic];
    S[v] = reinterpret_cast<const float4*>(sq)[ic];
    A[v] = reinterpret_cast<const float4*>(centered ? ga : sq)[ic];
  }
  __builtin_amdgcn_sched_barrier(0);
  const float coef = clip_coef_from_partials(partials, n_partials, max_norm, out_norm);
  DRA_STAMP(TR_STEP, 2);
#pragma unroll
  for (int v = 0; v < kStepNV; ++v) {
    const int64_t i = i0 + 256 * v;
    if (i < n4) {
      float* pp = &P[v].x; const float* gg = &G[v].x; float* ss = &S[v].x; float* aa = &A[v].x;
#pragma unroll
      for (int k = 0; k < 4; ++k) rmsprop_elem(pp[k], gg[k], ss[k], aa[k], coef, alpha, oma, lr, eps, centered);
      reinterpret_cast<float4*>(p)[i] = P[v];
      if (p_copy) reinterpret_cast<float4*>(p_copy)[i] = P[v];
      reinterpret_cast<float4*>(sq)[i] = S[v];
      if (centered) reinterpret_cast<float4*>(ga)[i] = A[v];
    }
  }
  // tail (n not a multiple of 4): the last few floats, by the first threads of workgroup 0
  const int64_t t = (n4 << 2) + threadIdx.x;
  if (blockIdx.x == 0 && t < n) {
    float pv = p[t], sv = sq[t], av = centered ? ga[t] : 0.f;
    rmsprop_elem(pv, g[t], sv, av, coef, alpha, oma, lr, eps, centered);
    p[t] = pv;
    if (p_copy) p_copy[t] = pv;
    sq[t] = sv;
    if (centered) ga[t] = av;
  }
  DRA_STAMP(TR_STEP, 5);
  DRA_STAMP_END(TR_STEP);
}

static inline int64_t step_blocks(int64_t n) {   // every float4 has its own thread slot: no grid-stride loop
  int64_t b = ((n >> 2) + 256 * kStepNV - 1) / (256 * kStepNV);
  return b < 1 ? 1 : b;
}

// param_copy (optional): the updated parameters are ALSO written there -- the async actor of the fused DQN
// learner reads a double-buffered copy so that the optimiser never has to wait for its forwards
// (DQN_agent.py:30,133: config.lock) -- +4 B/param of write traffic instead of a cross-queue join.
DRA_API int dra_rmsprop_step_copy(float* param, const float* grad, float* square_avg, float* grad_avg, int64_t n,
                                  const double* partials, int n_partials, float max_norm, float lr, float alpha,
                                  float eps, int centered, float* out_norm, float* param_copy, void* stream) {
  if (!param || !grad || !square_avg || (centered && !grad_avg) || n < 1) return DRA_EINVAL;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)square_avg) | ((uintptr_t)grad_avg) | ((uintptr_t)param_copy)) & 15)
    return DRA_EINVAL;
  if (n < 4) return DRA_EINVAL;
  if (partials && (n_partials < 1 || n_partials > dra_norm_partials_max())) return DRA_EINVAL;
  if (step_blocks(n) > 0x7fffffff) return DRA_EINVAL;
  hipLaunchKernelGGL(rmsprop_step_kernel, dim3((unsigned)step_blocks(n)), dim3(256), 0, dra_stream(stream), param, grad,
                     square_avg, grad_avg, n, partials, n_partials, max_norm, lr, alpha, eps, centered, out_norm, param_copy);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_rmsprop_step(float* param, const float* grad, float* square_avg, float* grad_avg, int64_t n,
                             const double* partials, int n_partials, float max_norm, float lr, float alpha, float eps,
                             int centered, float* out_norm, void* stream) {
  return dra_rmsprop_step_copy(param, grad, square_avg, grad_avg, n, partials, n_partials, max_norm, lr, alpha, eps,
                               centered, out_norm, nullptr, stream);
}

// torch.optim.Adam (no amsgrad / weight decay): m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
// p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  `step` is the 1-based count.
__global__ void __launch_bounds__(256)
adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                 int64_t n, const double* __restrict__ partials, int n_partials, float max_norm, float step_size,
                 float beta1, float beta2, float inv_sqrt_bc2, float eps, float* __restrict__ out_norm,
                 const float* __restrict__ hyper) {
  if (hyper) {  // bias-corrected step size / 1/sqrt(1-b2^t) from device memory: the launch is replayable in a graph
    step_size = hyper[0];
    inv_sqrt_bc2 = hyper[1];
  }
  const float coef = clip_coef_from_partials(partials, n_partials, max_norm, out_norm);
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gk = g[i] * coef;
    const float mk = m[i] * beta1 + omb1 * gk;
    const float vk = v[i] * beta2 + omb2 * gk * gk;
    m[i] = mk;
    v[i] = vk;
    p[i] = p[i] - step_size * (mk / (sqrtf(vk) * inv_sqrt_bc2 + eps));
  }
}

// host side of the bias corrections: step_size = lr / (1 - b1^t), inv_sqrt_bc2 = 1 / sqrt(1 - b2^t)
DRA_API int dra_adam_hyper(float lr, float beta1, float beta2, int64_t step, float* out2) {
  if (!out2 || step < 1) return DRA_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  out2[0] = (float)((double)lr / bc1);
  out2[1] = (float)(1.0 / sqrt(bc2));
  return DRA_OK;
}

static int launch_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const double* partials,
                       int n_partials, float max_norm, float step_size, float beta1, float beta2, float inv_sqrt_bc2,
                       float eps, float* out_norm, const float* hyper, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1) return DRA_EINVAL;
  if (partials && (n_partials < 1 || n_partials > dra_norm_partials_max())) return DRA_EINVAL;
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)b), dim3(256), 0, dra_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, n, partials, n_partials, max_norm, step_size, beta1, beta2, inv_sqrt_bc2, eps, out_norm,
                     hyper);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                          const double* partials, int n_partials, float max_norm, float lr, float beta1, float beta2,
                          float eps, int64_t step, float* out_norm, void* stream) {
  float hp[2];
  if (dra_adam_hyper(lr, beta1, beta2, step, hp)) return DRA_EINVAL;
  return launch_adam(param, grad, exp_avg, exp_avg_sq, n, partials, n_partials, max_norm, hp[0], beta1, beta2, hp[1], eps,
                     out_norm, nullptr, stream);
}

// Same step with the two step-dependent scalars read from DEVICE memory (hyper_dev = {step_size, inv_sqrt_bc2},
// filled from dra_adam_hyper): every kernel argument is then constant across steps and the launch can be
// replayed from a captured graph.
DRA_API int dra_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              const double* partials, int n_partials, float max_norm, float beta1, float beta2,
                              float eps, const float* hyper_dev, float* out_norm, void* stream) {
  if (!hyper_dev) return DRA_EINVAL;
  return launch_adam(param, grad, exp_avg, exp_avg_sq, n, partials, n_partials, max_norm, 0.f, beta1, beta2, 0.f, eps,
                     out_norm, hyper_dev, stream);
}

// Adam for a captured learner graph: the 1-based step count lives in DEVICE memory (*step_dev, bumped by an earlier
// kernel of the same graph), the bias corrections are formed from it exactly as dra_adam_hyper does on the host, and
// the updated parameters are optionally mirrored into param_copy (the async actor's double-buffered copy).  Same
// element formula and load shape as rmsprop_step_kernel: 2 float4 per thread, every operand requested before the clip
// coefficient is reduced from the partials.
__global__ void __launch_bounds__(256)
adam_step_ctr_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                     int64_t n, const double* __restrict__ partials, int n_partials, float max_norm, float lr, float beta1,
                     float beta2, float eps, const int64_t* __restrict__ step_dev, float* __restrict__ out_norm,
                     float* __restrict__ p_copy) {
  __shared__ float s_hyper[2];
  const int64_t n4 = n >> 2;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * kStepNV) + threadIdx.x;
  float4 P[kStepNV], G[kStepNV], M[kStepNV], V[kStepNV];
#pragma unroll
  for (int q = 0; q < kStepNV; ++q) {
    const int64_t i = i0 + 256 * q;
    const int64_t ic = i < n4 ? i : n4 - 1;
    P[q] = reinterpret_cast<float4*>(p)[ic];
    G[q] = reinterpret_cast<const float4*>(g)[ic];
    M[q] = reinterpret_cast<float4*>(m)[ic];
    V[q] = reinterpret_cast<float4*>(v)[ic];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x == 0) {
    const double t = (double)*step_dev;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    s_hyper[0] = (float)((double)lr / bc1);
    s_hyper[1] = (float)(1.0 / sqrt(bc2));
  }
  const float coef = clip_coef_from_partials(partials, n_partials, max_norm, out_norm);   // (synchronises the workgroup)
  const float step_size = s_hyper[0], inv_sqrt_bc2 = s_hyper[1];
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  auto elem = [&](float& pp, float gg, float& mm, float& vv) {
    const float gk = gg * coef;
    mm = mm * beta1 + omb1 * gk;
    vv = vv * beta2 + omb2 * gk * gk;
    pp = pp - step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  };
#pragma unroll
  for (int q = 0; q < kStepNV; ++q) {
    const int64_t i = i0 + 256 * q;
    if (i < n4) {
      float* pp = &P[q].x; const float* gg = &G[q].x; float* mm = &M[q].x; float* vv = &V[q].x;
#pragma unroll
      for (int k = 0; k < 4; ++k) elem(pp[k], gg[k], mm[k], vv[k]);
      reinterpret_cast<float4*>(p)[i] = P[q];
      if (p_copy) reinterpret_cast<float4*>(p_copy)[i] = P[q];
      reinterpret_cast<float4*>(m)[i] = M[q];
      reinterpret_cast<float4*>(v)[i] = V[q];
    }
  }
  const int64_t t = (n4 << 2) + threadIdx.x;
  if (blockIdx.x == 0 && t < n) {
    float pv = p[t], mv = m[t], vv = v[t];
    elem(pv, g[t], mv, vv);
    p[t] = pv;
    if (p_copy) p_copy[t] = pv;
    m[t] = mv;
    v[t] = vv;
  }
}

DRA_API int dra_adam_step_counter(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  const double* partials, int n_partials, float max_norm, float lr, float beta1, float beta2,
                                  float eps, const int64_t* step_dev, float* out_norm, float* param_copy, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev || n < 4) return DRA_EINVAL;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq) | ((uintptr_t)param_copy)) & 15)
    return DRA_EINVAL;
  if (partials && (n_partials < 1 || n_partials > dra_norm_partials_max())) return DRA_EINVAL;
  if (step_blocks(n) > 0x7fffffff) return DRA_EINVAL;
  hipLaunchKernelGGL(adam_step_ctr_kernel, dim3((unsigned)step_blocks(n)), dim3(256), 0, dra_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, n, partials, n_partials, max_norm, lr, beta1, beta2, eps, step_dev, out_norm, param_copy);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Device-to-device parameter copy for the target-network sync (DQN_agent.py:136-138).
DRA_API int dra_copy_f32(float* dst, const float* src, int64_t n, void* stream) {
  if (!dst || !src || n < 0) return DRA_EINVAL;
  DRA_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, dra_stream(stream)));
  return DRA_OK;
}

// Polyak averaging of a target network over the two networks' FLAT parameter buffers (DDPG_agent.py:26-30, TD3_agent.py:28-32):
//   target[i] = target[i] * keep + src[i] * mix          keep = f32(1 - mix), both products rounded before the add --
// the reference's `target_param * (1.0 - mix) + param * mix` (fp contraction is off for this library).  One launch, 16
// bytes per lane; 8 B read + 4 B written per parameter: bandwidth-bound.
__global__ void __launch_bounds__(256) soft_update_kernel(float* __restrict__ target, const float* __restrict__ src, int64_t n,
                                                          float keep, float mix) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x, i = 4 * i4;
  if (i + 4 <= n) {
    float4 t = reinterpret_cast<float4*>(target)[i4];
    const float4 s = reinterpret_cast<const float4*>(src)[i4];
    t.x = t.x * keep + s.x * mix;
    t.y = t.y * keep + s.y * mix;
    t.z = t.z * keep + s.z * mix;
    t.w = t.w * keep + s.w * mix;
    reinterpret_cast<float4*>(target)[i4] = t;
  } else {
    for (int64_t j = i; j < n; ++j) target[j] = target[j] * keep + src[j] * mix;
  }
}

DRA_API int dra_soft_update(float* target, const float* src, int64_t n, float keep, float mix, void* stream) {
  if (!target || !src || n < 0) return DRA_EINVAL;
  if ((((uintptr_t)target) | ((uintptr_t)src)) & 15) return DRA_EINVAL;
  if (n == 0) return DRA_OK;
  const int64_t blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 0x7fffffff) return DRA_EINVAL;
  hipLaunchKernelGGL(soft_update_kernel, dim3((unsigned)blocks), dim3(256), 0, dra_stream(stream), target, src, n, keep, mix);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

#ifdef DRA_TRACE
extern "C" int dra_trace_set_optim(void* p) { return dra_trace_set_local(p); }
#endif
