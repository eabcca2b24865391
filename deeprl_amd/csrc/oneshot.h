// One-memory-round-trip contractions of the DQN update at batch 32, written as "roles" that can share
// a launch (multi_kernel).
//
// Why: at batch 32 every contraction of the update is 0.1-0.2 GFLOP on <= 400 workgroups; the K-chunked
// implicit GEMM (igemm.h) pays one dependent global-memory round trip (~2 us) per 64-wide chunk and
// 10-20 integer div/mods per fetched element, so the backward kernels run 11-29 us each at 5-8 % of the
// fp32 MFMA rate (profiles/r01_*).  The kernels below follow conv_v2.hip's recipe instead:
//   * a workgroup owns a 32x32 (or a few 32x32) output tile and ALL of its reduction;
//   * every global load of the workgroup is issued up front (one exposed memory latency);
//   * the operand whose MFMA lane axis is contiguous in memory goes straight to registers (128-byte
//     coalesced rows); the other is staged through LDS once, in a layout where the 32 lanes of an
//     operand read hit 32 different banks and the per-MFMA address is `base + immediate`;
//   * the reduction is split over the 4 waves and folded through LDS in a fixed order
//     ((w0+w1)+(w2+w3)), so results are run-to-run deterministic.
// fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere: an exact fmaf chain, as the 1e-5 parity bar needs.
#pragma once
#include "igemm.h"

// 32x32 MFMA C/D row of accumulator register r for half-wave h
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
  return a;
}

// Fold the 4 waves' partial accumulators of ONE 32x32 tile through `red` (4096 floats): wave w gets
// the sums of accumulator registers 4w .. 4w+3.  Caller guarantees `red` is no longer read as operands.
__device__ __forceinline__ void reduce4(float* __restrict__ red, const f32x16& acc, int wave, int lane, float out[4]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = wave * 4 + q;
    out[q] = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) +
             (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
  }
}

// ------------------------------------------------------------------------------------------------
// Roles: a role is a POD with  LDS_FLOATS,  run(block, lds)  and a host-side block count; multi_kernel
// runs up to three independent roles in one launch (block ranges [0,n1) [n1,n1+n2) [n1+n2, ...)).
struct NoRole {
  static constexpr int LDS_FLOATS = 0;
  __device__ __forceinline__ void run(int, float*, int = 0) const {}
};

template <class P>
struct IgemmRole {
  static constexpr int LDS_FLOATS = igemm_lds_floats<P>();
  P p;
  int tiles, ksplit;
  __device__ __forceinline__ void run(int bid, float* lds, int = 0) const {
    const int bx = bid % tiles, r = bid / tiles;
    const int by = r % ksplit, bz = r / ksplit;
    igemm_body<P>(p, bx, by, bz, ksplit, lds);
  }
};

template <class P>
static IgemmRole<P> make_igemm_role(const P& p, int ksplit) {
  IgemmRole<P> r;
  r.p = p;
  r.tiles = ((p.M + P::BM - 1) / P::BM) * ((p.N + P::BN - 1) / P::BN);
  r.ksplit = ksplit;
  return r;
}

// amdgpu_waves_per_eu(1, 2): these kernels want registers (every load of a workgroup in flight at once),
// not occupancy; without it the scheduler throttles loads to stay under 64 VGPRs.
template <class R1, class R2, class R3>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) multi_kernel(const R1 r1, const R2 r2, const R3 r3, const int n1, const int n2) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  const int b = blockIdx.x;
  // (third argument: index of the role's first workgroup in the launch -- the XCD a workgroup runs on is blockIdx mod 8)
  if (b < n1) r1.run(b, dyn_lds, 0);
  else if (b < n1 + n2) r2.run(b - n1, dyn_lds, n1);
  else r3.run(b - n1 - n2, dyn_lds, n1 + n2);
}

// the same launch for throughput-sized grids (rollout batch sizes: thousands of one-sample workgroups): three waves per SIMD
// instead of two, so that one workgroup's loads / staging / epilogue hide behind two others' MFMAs
template <class R1, class R2, class R3>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) multi_kernel_tp(const R1 r1, const R2 r2, const R3 r3, const int n1, const int n2) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  const int b = blockIdx.x;
  if (b < n1) r1.run(b, dyn_lds, 0);
  else if (b < n1 + n2) r2.run(b - n1, dyn_lds, n1);
  else r3.run(b - n1 - n2, dyn_lds, n1 + n2);
}

template <class R1, class R2, class R3>
static int launch_multi_tp(const R1& r1, int n1, const R2& r2, int n2, const R3& r3, int n3, hipStream_t st) {
  constexpr int f12 = R1::LDS_FLOATS > R2::LDS_FLOATS ? R1::LDS_FLOATS : R2::LDS_FLOATS;
  constexpr int fl = f12 > R3::LDS_FLOATS ? f12 : R3::LDS_FLOATS;
  constexpr size_t bytes = (size_t)fl * sizeof(float);
  static_assert(bytes <= 160 * 1024, "LDS per workgroup");
  if (n1 < 0 || n2 < 0 || n3 < 0 || n1 + n2 + n3 < 1) return DRA_EINVAL;
  static DraLdsAttr lds_attr;     // one per instantiation, per device
  if (bytes > 64 * 1024)
    if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&multi_kernel_tp<R1, R2, R3>), bytes)) return rc;
  hipLaunchKernelGGL((multi_kernel_tp<R1, R2, R3>), dim3(n1 + n2 + n3), dim3(256), bytes, st, r1, r2, r3, n1, n2);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

template <class R1, class R2, class R3>
static int launch_multi(const R1& r1, int n1, const R2& r2, int n2, const R3& r3, int n3, hipStream_t st) {
  constexpr int f12 = R1::LDS_FLOATS > R2::LDS_FLOATS ? R1::LDS_FLOATS : R2::LDS_FLOATS;
  constexpr int fl = f12 > R3::LDS_FLOATS ? f12 : R3::LDS_FLOATS;
  constexpr size_t bytes = (size_t)fl * sizeof(float);
  static_assert(bytes <= 160 * 1024, "LDS per workgroup");
  if (n1 < 0 || n2 < 0 || n3 < 0 || n1 + n2 + n3 < 1) return DRA_EINVAL;
  static DraLdsAttr lds_attr;     // one per instantiation, per device
  if (bytes > 64 * 1024)
    if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&multi_kernel<R1, R2, R3>), bytes)) return rc;
  hipLaunchKernelGGL((multi_kernel<R1, R2, R3>), dim3(n1 + n2 + n3), dim3(256), bytes, st, r1, r2, r3, n1, n2);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------------
// VanillaNet head weight gradient (network_heads.py:18-21 backward):
//   dWh[a][k] = sum_b dq[b][a] * h4[b][k] ;  dbh[a] = sum_b dq[b][a].   blocks = A * 2 (256 k each)
struct HeadWgradRole {
  static constexpr int LDS_FLOATS = 8;
  const float* dq;   // [B][A]
  const float* h4;   // [B][512]
  float* dwh;        // [A][512]
  float* dbh;        // [A]
  int B, A;
  double* partials = nullptr;   // optional [2 * A]: sum of squares of what each workgroup stored (DRA_VAR_LATE_FOLD)
  // Distributional heads (A = n_actions * group outputs): the loss differentiates the TAKEN action's atoms / quantiles only
  // (CategoricalDQN_agent.py:78-82, QuantileRegressionDQN_agent.py:62-66), dq is exactly zero elsewhere (the loss kernels write
  // the zeros), so output row a gets terms from the samples whose action is a / group only.  action != null: the other samples
  // are skipped -- each skipped term is +0 * h4, so the result is the dense sum's bit for bit -- a quarter of the reads at 4 actions.
  const int64_t* action = nullptr;
  int group = 0;
  ChainHook hook;     // DRA_VAR_HEAD_CHAIN (run_<true>: dq / h4 come from the head role of the SAME launch)
  __device__ __forceinline__ void run(int bid, float* lds, int = 0) const { run_<false>(bid, lds); }
  template <bool CIN>
  __device__ __forceinline__ void run_(int bid, float* lds) const {
    const int a = bid >> 1, k = (bid & 1) * 256 + threadIdx.x;
    float acc = 0.f, accb = 0.f;
    int b = 0;
    if constexpr (CIN) mega_wait(hook.sync(0));
    if (!CIN && action) {
      // the matching samples of 64 at a time as a wave-uniform bit mask; up to eight of them per round, their loads in flight
      // together (a scalar loop with a branch per sample was SLOWER than the dense sum: one memory latency per match)
      const int64_t mine = a / group;
      const int lane = threadIdx.x & 63;
      for (; b < B; b += 64) {
        const int bl = b + lane;
        unsigned long long mask = __ballot(bl < B && action[bl < B ? bl : B - 1] == mine);
        while (mask) {
          int idx[8];
          float d[8], h[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            idx[i] = mask ? b + __ffsll((long long)mask) - 1 : -1;
            mask &= mask - 1ull;                  // (0 stays 0)
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int bi = idx[i] < 0 ? 0 : idx[i];
            d[i] = dq[(int64_t)bi * A + a];
            h[i] = h4[(int64_t)bi * 512 + k];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (idx[i] >= 0) { acc += d[i] * h[i]; accb += d[i]; }
        }
      }
      b = B;
    }
    for (; b + 8 <= B; b += 8) {
      float d[8], h[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { d[i] = mega_ld<CIN>(dq + (int64_t)(b + i) * A + a); h[i] = mega_ld<CIN>(h4 + (int64_t)(b + i) * 512 + k); }
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc += d[i] * h[i]; accb += d[i]; }
    }
    for (; b < B; ++b) {
      const float d = mega_ld<CIN>(dq + (int64_t)b * A + a);
      acc += d * mega_ld<CIN>(h4 + (int64_t)b * 512 + k);
      accb += d;
    }
    dwh[a * 512 + k] = acc;
    if (k == 0) dbh[a] = accb;
    if (partials) {
      const double d = wave_sum((double)(acc * acc + (k == 0 ? accb * accb : 0.f)));
      double* dl = reinterpret_cast<double*>(lds);
      if ((threadIdx.x & 63) == 0) dl[threadIdx.x >> 6] = d;
      __syncthreads();
      if (threadIdx.x == 0) partials[bid] = (dl[0] + dl[1]) + (dl[2] + dl[3]);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Slab fold as a ROLE (DRA_VAR_LATE_FOLD): grad[begin + i] = sum_s slabs[s * stride + i] in slab order, for a layer whose
// weight-gradient slabs were written by the PREVIOUS launch, riding in the spare workgroup slots of the next layer's
// backward launch instead of a norm pass on the update's dependent chain.  64 float4 per workgroup: thread (g, el) adds
// slabs g, g + 4, ... (<= 8 each: every load of the workgroup in flight at once, one memory round trip), the four group
// sums meet in LDS and are added in group order; the workgroup's sum of squares goes to partials[bid].
struct FoldRole {
  static constexpr int NG = 4, EPB = 64, SPT = 8;      // n_slabs <= NG * SPT = 32
  static constexpr int LDS_FLOATS = NG * (EPB + 1) * 4;
  float* grad;            // flat gradient
  const float* slabs;     // [n_slabs][stride]
  int64_t begin4, count4, stride4;
  int n_slabs;
  double* partials;       // [blocks()]
  // optional: workgroup 0 resets these slots to -1.0 (the arrival slots of the late-fold optimizer launch that follows)
  double* reset_slots = nullptr;
  int n_reset = 0;
  __host__ int blocks() const { return (int)((count4 + EPB - 1) / EPB); }
  ChainHook hook;     // DRA_VAR_BWD_CHAIN (run_<CIN>: the slabs come from the weight-gradient workgroups of the SAME launch)
  __device__ __forceinline__ void run(int bid, float* lds, int = 0) const { run_<false>(bid, lds); }
  template <bool CIN>
  __device__ __forceinline__ void run_(int bid, float* lds, int = 0) const {
    const int tid = threadIdx.x, g = tid >> 6, el = tid & 63;
    if (reset_slots && bid == 0)
      for (int i = tid; i < n_reset; i += 256) __hip_atomic_store(reset_slots + i, -1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t i = (int64_t)bid * EPB + el;
    const int64_t ic = i < count4 ? i : count4 - 1;
    const float4* __restrict__ sl = reinterpret_cast<const float4*>(slabs) + ic;
    float4 t[SPT];
    if constexpr (CIN) mega_wait(hook.sync(0));
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
      const int s = g + NG * u;
      const dra_f4 v4 = mega_ld4<CIN>(reinterpret_cast<const dra_f4*>(sl + (int64_t)(s < n_slabs ? s : 0) * stride4));
      t[u] = make_float4(v4.x, v4.y, v4.z, v4.w);
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < SPT; ++u)
      if (g + NG * u < n_slabs) { pp.x += t[u].x; pp.y += t[u].y; pp.z += t[u].z; pp.w += t[u].w; }
    float4* sf = reinterpret_cast<float4*>(lds);
    sf[g * (EPB + 1) + el] = pp;
    __syncthreads();
    float sq = 0.f;
    if (tid < EPB) {
      float4 r = sf[el];
#pragma unroll
      for (int q = 1; q < NG; ++q) {
        const float4 o = sf[q * (EPB + 1) + el];
        r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
      }
      if (i < count4) {
        reinterpret_cast<float4*>(grad)[begin4 + i] = r;
        sq = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
      }
    }
    const double d = wave_sum((double)sq);     // the owners are wave 0
    if (tid == 0) partials[bid] = d;
  }
};

// ------------------------------------------------------------------------------------------------
// Linear input gradient, whole reduction in one pass (fc4: O = 512):
//   dx[b][i] = act'(xact[b][i]) * sum_o dy[b][o] * W[o][i]          M = b (32 rows), N = i, K = o
// A = dy rows -> LDS [32][O+1] (coalesced loads, conflict-free lane=row reads);  B = W[o][i0..i0+31]
// straight to registers (128-byte rows).  MFMA slot (j, h) of wave w <-> o = w*O/4 + h*O/8 + j.
template <int O>
struct LinDgradOne {
  static constexpr int KW = O / 4, NJ = KW / 2, LDA = O + 1, RA = 32 * O / 256;
  static constexpr int LDS_FLOATS = 32 * LDA > 4096 ? 32 * LDA : 4096;
  static_assert(O % 8 == 0 && (32 * O) % 256 == 0, "reduction split over 4 waves x 2 half-waves");
  const float* dy;    // [B][O]
  const float* w;     // [O][I]
  const float* xact;  // [B][I] or null
  float* dx;          // [B][I]
  int B, I, act, tiles_n;
  ChainHook hook;     // DRA_VAR_BWD_CHAIN_FC (run_<true>: dx goes to workgroups of the SAME launch -- conv3's backward roles)
  __device__ __forceinline__ void run(int bid, float* __restrict__ lds, int = 0) const { run_<false>(bid, lds); }
  // CIN (DRA_VAR_HEAD_CHAIN): dy comes from the head role of the SAME launch -- the weights and the activation-derivative source
  // are requested first, then the wait, then agent-scope loads of dy
  template <bool COUT, bool CIN = false>
  __device__ __forceinline__ void run_(int bid, float* __restrict__ lds) const {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int bm = bid / tiles_n, bn = bid - bm * tiles_n;
    const int m0 = bm * 32, n0 = bn * 32;
    const int ncol = min(n0 + li, I - 1);
    DRA_STAMP(TR_FC_B, 0);
    const int kb = wave * KW + h * NJ;
    float breg[NJ];
    {
      const float* wp = w + (int64_t)kb * I + ncol;
#pragma unroll
      for (int j = 0; j < NJ; ++j) breg[j] = wp[(int64_t)j * I];
    }
    float araw[RA];
    if constexpr (!CIN) {
#pragma unroll
      for (int q = 0; q < RA; ++q) {
        const int e = tid + 256 * q, row = e / O, col = e - row * O;
        araw[q] = dy[(int64_t)min(m0 + row, B - 1) * O + col];
      }
    }
    float aux[4];
    {
      const float* src = xact ? xact : dx;  // null xact: any mapped address, value ignored
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = min(m0 + mfma_row(wave * 4 + q, h), B - 1);
        aux[q] = src[(int64_t)m * I + ncol];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // every load above is issued before the first LDS write below
    if constexpr (CIN) {
      mega_wait(hook.sync(0));
#pragma unroll
      for (int q = 0; q < RA; ++q) {
        const int e = tid + 256 * q, row = e / O, col = e - row * O;
        araw[q] = mega_ld<true>(dy + (int64_t)min(m0 + row, B - 1) * O + col);
      }
    }
#pragma unroll
    for (int q = 0; q < RA; ++q) {
      const int e = tid + 256 * q, row = e / O, col = e - row * O;
      lds[row * LDA + col] = araw[q];
    }
    DRA_STAMP(TR_FC_B, 1);
    __syncthreads();
    DRA_STAMP(TR_FC_B, 2);
    f32x16 acc = zero16();
    const float* ap = lds + li * LDA + kb;
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[j], breg[j], acc, 0, 0, 0);
    DRA_STAMP(TR_FC_B, 3);
    __syncthreads();
    DRA_STAMP(TR_FC_B, 4);
    float s[4];
    reduce4(lds, acc, wave, lane, s);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + mfma_row(wave * 4 + q, h);
      if (m < B && n0 + li < I) mega_st<COUT>(&dx[(int64_t)m * I + n0 + li], xact ? s[q] * act_grad(aux[q], act) : s[q]);
    }
    DRA_STAMP(TR_FC_B, 5);
    if constexpr (COUT) mega_publish(hook.sync(0));     // (one counter for the whole role: a column tile covers every sample)
    DRA_STAMP_END(TR_FC_B);
  }
};

// ------------------------------------------------------------------------------------------------
// Linear forward partial sums, one K-split per workgroup, whole split in one pass (fc4: I = 3136, KS = 8):
//   slabs[z][s][b][o] = sum_{k in split s} x_z[b][k] * W_z[o][k]      M = b, N = o, K = I / KS
// Both operands are k-contiguous in memory: float4 loads -> LDS [32][KPS+1] each; lane = row reads are
// conflict-free (odd row stride).  The consumer (head_fused_kernel) reduces the KS slabs.
// NT = 32-wide output tiles per workgroup (sharing the staged x rows).  NT = 2 keeps fc4 at 128 workgroups for
// two networks: one round even on the 192-CU partition the update chain owns (at 100-150 KB of LDS a CU holds
// one of these workgroups; with NT = 1 the 256 workgroups ran in two rounds there: 12.4 us vs 8.4 us on 256 CUs).
template <int I, int KS, int NT = 1>
struct LinFwdSlabsOne {
  static constexpr int KPS = I / KS, KW = KPS / 4, NJ = KW / 2, LD = KPS + 1;
  static constexpr int V = KPS / 4, NVX = 32 * V, NVW = 32 * NT * V, RX = (NVX + 255) / 256, RWV = (NVW + 255) / 256;
  static constexpr int LDS_FLOATS = (1 + NT) * 32 * LD > NT * 4096 ? (1 + NT) * 32 * LD : NT * 4096;
  static_assert(I % KS == 0 && KPS % 8 == 0, "K split: float4 rows, even k per half-wave");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS per workgroup");
  const float* x[kMaxZ];  // [B][I]
  const float* w[kMaxZ];  // [O][I]
  float* slabs;           // [nz][KS][B][O]
  int B, O, tiles_n, tiles_m;   // tiles_n counts 32*NT-wide tile groups
  // optional (KS == 1: the whole reduction in one workgroup): finished outputs y_z[b][o] = bias_z[o] + sum instead of slabs
  // (the distributional heads' [B,512] x [512, A*N] contraction of the update)
  const float* bias[kMaxZ] = {};
  float* out[kMaxZ] = {};
  int xcd = 0, n_groups = 0;   // xcd != 0: the tiles_n workgroups that share one (net, K slice, row tile) of x run on one XCD
  __device__ __forceinline__ void run(int bid_, float* __restrict__ lds, int first = 0) const {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int bid = xcd ? xcd_order(bid_, first, n_groups, tiles_n) : bid_;
    const int bn = bid % tiles_n;
    int r = bid / tiles_n;
    const int bm = r % tiles_m;
    r /= tiles_m;
    const int s = r % KS, z = r / KS;
    const int m0 = bm * 32, n0 = bn * 32 * NT, k0 = s * KPS;
    const float* __restrict__ xz = x[z];
    const float* __restrict__ wz = w[z];
    DRA_STAMP(TR_FC4_F, 0);
    float bias_r[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_r[t] = (KS == 1 && out[z]) ? bias[z][min(n0 + 32 * t + li, O - 1)] : 0.f;
    float4 xa[RX], wa[RWV];
#pragma unroll
    for (int q = 0; q < RX; ++q) {
      const int e = min(tid + 256 * q, NVX - 1), row = e / V, c4 = e - row * V;
      xa[q] = *reinterpret_cast<const float4*>(xz + (int64_t)min(m0 + row, B - 1) * I + k0 + 4 * c4);
    }
#pragma unroll
    for (int q = 0; q < RWV; ++q) {
      const int e = min(tid + 256 * q, NVW - 1), row = e / V, c4 = e - row * V;
      wa[q] = *reinterpret_cast<const float4*>(wz + (int64_t)min(n0 + row, O - 1) * I + k0 + 4 * c4);
    }
    float* xs = lds;
    float* ws = lds + 32 * LD;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < RX; ++q) {
      const int e = tid + 256 * q;
      if (e < NVX) {
        const int row = e / V, c4 = e - row * V;
        float* dx_ = xs + row * LD + 4 * c4;
        dx_[0] = xa[q].x; dx_[1] = xa[q].y; dx_[2] = xa[q].z; dx_[3] = xa[q].w;
      }
    }
#pragma unroll
    for (int q = 0; q < RWV; ++q) {
      const int e = tid + 256 * q;
      if (e < NVW) {
        const int row = e / V, c4 = e - row * V;
        float* dw_ = ws + row * LD + 4 * c4;
        dw_[0] = wa[q].x; dw_[1] = wa[q].y; dw_[2] = wa[q].z; dw_[3] = wa[q].w;
      }
    }
    DRA_STAMP(TR_FC4_F, 1);
    __syncthreads();
    DRA_STAMP(TR_FC4_F, 2);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero16();
    const float* ap = xs + li * LD + wave * KW + h * NJ;
    const float* bp = ws + li * LD + wave * KW + h * NJ;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float a = ap[j];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[t * 32 * LD + j], acc[t], 0, 0, 0);
    }
    DRA_STAMP(TR_FC4_F, 3);
    __syncthreads();
    DRA_STAMP(TR_FC4_F, 4);
    const bool finished = KS == 1 && out[z] != nullptr;
    float* outp = finished ? out[z] : slabs + ((int64_t)(z * KS + s) * B) * O;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float sum[4];
      reduce4(lds + t * 4096, acc[t], wave, lane, sum);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + mfma_row(wave * 4 + q, h);
        const int n = n0 + 32 * t + li;
        if (m < B && n < O) outp[(int64_t)m * O + n] = finished ? sum[q] + bias_r[t] : sum[q];
      }
    }
    DRA_STAMP(TR_FC4_F, 5);
    DRA_STAMP_END(TR_FC4_F);
  }
};

// ------------------------------------------------------------------------------------------------
// Convolution weight gradient in the KOC layout, one pass:
//   dWt[k][oc] = sum_(b,p) Xcol[k][(b,p)] * dY[b][oc][p],  k = (c,kh,kw);   db[oc] = sum_(b,p) dY[b][oc][p]
// Workgroup = (sample b, chunk of ROWS output rows, group of MTG 32-row k tiles): M = k (lane li = tap),
// N = oc (lane li = output channel), reduction over the chunk's positions.  The input rows the chunk
// touches are staged to LDS once ([channel][row][RW], uint8 frames normalised f32(f64(v)*coef) on the
// way in); dY of the chunk is staged transposed ([oh][ow][OC+1]) so the B operand is a conflict-free
// lane = oc read.  The two k-slices of an MFMA are the even / odd output columns of the same row, so
// both operand addresses are `lane base + immediate`.  Every (sample, chunk) writes its own slab:
// slab index = b * (OH/ROWS) + chunk; the fold happens in the gradient-norm pass (dra_grad_sqnorm_segs).
// SHARES > 0 (rollout batch sizes, round 5): PERSISTENT over the batch -- a workgroup = (k group, share of the batch) walks its
// samples' chunks with its output tiles resident in the accumulators and writes ONE slab per share (conv1 at batch 1024: 5120
// slabs of 33 KB, 168 MB written and folded back, became SHARES slabs).
template <class G, int ROWS, int MTG, int RW_, int CSPAD, bool U8, int SHARES = 0>
struct ConvWgradOne {
  static constexpr int S = G::S, OH = G::OH, OWP = (OH + 1) & ~1, NPAIR = OWP / 2, NJ = ROWS * NPAIR;
  static constexpr int NCHUNK = OH / ROWS;
  static constexpr int MTILES = G::K / 32, NGRP = MTILES / MTG, NTL = G::OC / 32, TILES = MTG * NTL;
  static constexpr int TPW = (TILES + 3) / 4;
  static constexpr int NR = (ROWS - 1) * S + G::KH;            // input rows a chunk touches
  static constexpr int RW = RW_, CS = NR * RW + CSPAD;          // LDS row / channel strides of the image
  // channels a group of MTG*32 consecutive k can touch (exact when groups start on channel boundaries)
  static constexpr int NCHMAX = ((MTG * 32) % G::KK == 0) ? (MTG * 32) / G::KK : (MTG * 32 + G::KK - 2) / G::KK + 1;
  static constexpr int NCH = NCHMAX < G::C ? NCHMAX : G::C;
  static constexpr int IMG = NCH * CS + RW;                     // + zeroed tail for the odd-column pad slot
  static constexpr int LDB = G::OC + 1, NPOS = ROWS * OWP, DYF = NPOS * LDB;
  static constexpr int LDS_FLOATS = (IMG + DYF + 3) & ~3;       // whole float4s: the zero fill writes 16 bytes at a time
  static_assert(G::K % 32 == 0 && MTILES % MTG == 0 && OH % ROWS == 0, "tiling");
  static_assert(RW >= (OWP - 1) * S + G::KH, "LDS row holds the pad column's taps");
  const float* dy;   // [B][OC][OH][OH]
  const void* x;     // [B][C][H][H] f32 or u8
  float* dw;         // slab 0 of dWt [K][OC]
  float* db;         // slab 0 of db [OC]
  int64_t slab_stride;
  int B;
  double coef;
  // optional (U8): sample bi is the C consecutive ring frames ending at slot sample_idx[bi] of the slot-major frame array x
  // (conv1's weight gradient straight from the replay ring); null = image bi of a plain [B][C][H][H] batch
  const int64_t* sample_idx = nullptr;
  int xcd = 0;        // != 0: all workgroups of a sample on one XCD (xcd_order)
  __host__ __device__ static int spw(int batch) { return SHARES ? (batch + SHARES - 1) / SHARES : 1; }   // samples per workgroup
  __host__ int blocks() const { return SHARES ? ((B + spw(B) - 1) / spw(B)) * NGRP : B * NCHUNK * NGRP; }
  __host__ static int n_slabs(int batch) { return SHARES ? (batch + spw(batch) - 1) / spw(batch) : batch * NCHUNK; }
  ChainHook hook;     // DRA_VAR_BWD_CHAIN (run_<CIN>: dy from workgroups of the same launch; uint8 input, one sample per workgroup)
  __device__ __forceinline__ void run(int bid_, float* __restrict__ lds, int first = 0) const { run_<false>(bid_, lds, first); }
  template <bool CIN>
  __device__ __forceinline__ void run_(int bid_, float* __restrict__ lds, int first = 0) const {
    static_assert(!CIN || (U8 && SHARES == 0), "chained input: conv1's one-slab-per-chunk role");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int bid = (xcd && !SHARES) ? xcd_order(bid_, first, B, NCHUNK * NGRP) : bid_;
    const int grp = bid % NGRP;
    const int r = bid / NGRP;
    // the (sample, chunk) pairs this workgroup contracts: one (r) in the one-slab-per-chunk form, a share's in the persistent form
    const int it0 = SHARES ? r * spw(B) * NCHUNK : r;
    const int it1 = SHARES ? min(B, (r + 1) * spw(B)) * NCHUNK : r + 1;
    const int k0 = grp * MTG * 32;
    const int c_lo = k0 / G::KK;
    const int c_hi = min((k0 + MTG * 32 - 1) / G::KK, G::C - 1);
    const int nch = c_hi - c_lo + 1;                       // <= NCH
    float* img = lds;
    float* dyl = lds + IMG;
    [[maybe_unused]] constexpr int TRR = (G::C == 4) ? TR_CONV1_B : ((G::C == 32) ? TR_CONV2_B : TR_CONV3_B);
    DRA_STAMP(TRR, 0);
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = zero16();
    float sb_acc = 0.f;
    for (int it = it0; it < it1; ++it) {
    const int chunk = it % NCHUNK, bi = it / NCHUNK;
    const int ir0 = chunk * ROWS * S;                      // first input row
    // ---- issue all loads: dY chunk, then the image rows.  Loads walk the SOURCE as float4 (a few per lane, one constant
    // division each) instead of one dword per padded LDS cell with two or three divisions (SQ counters: ~1300 VALU
    // instructions per wave around 90 MFMAs, profiles/r02a_sq_learner_b32.json); padding is zero-filled up front.
    // dY: per output channel the chunk's ROWS*OH gradients are one contiguous run; a whole-sample chunk (ROWS == OH)
    // makes all OC runs one block.
    constexpr bool WHOLE = (ROWS == OH);
    constexpr int RUN = ROWS * OH;                                   // floats per output channel in this chunk
    static_assert(WHOLE ? (G::OC * G::P) % 4 == 0 : RUN % 4 == 0, "float4 dY staging");
    constexpr int NVD = WHOLE ? (G::OC * G::P) / 4 : G::OC * (RUN / 4), RD = (NVD + 255) / 256;
    float4 draw[RD];
    const float* dyb = dy + (int64_t)bi * G::OC * G::P + chunk * ROWS * OH;
    auto request_dy = [&](auto coh) {
#pragma unroll
      for (int q = 0; q < RD; ++q) {
        const int f = min(tid + 256 * q, NVD - 1);
        const float* src;
        if constexpr (WHOLE) {
          src = dyb + 4 * f;
        } else {
          const int oc = f / (RUN / 4), v = f - oc * (RUN / 4);
          src = dyb + oc * G::P + 4 * v;
        }
        const dra_f4 v4 = mega_ld4<decltype(coh)::value>(reinterpret_cast<const dra_f4*>(src));
        draw[q] = make_float4(v4.x, v4.y, v4.z, v4.w);
      }
    };
    if constexpr (!CIN) request_dy(std::false_type{});
    if constexpr (U8) {
      constexpr int WPR = G::H / 4;                         // u32 words per 84-byte row
      constexpr int NW = NCH * NR * WPR, RI = (NW + 255) / 256;
      unsigned iraw[RI];
      const int64_t first = sample_idx ? sample_idx[bi] - (G::C - 1) : (int64_t)bi * G::C;   // first frame of the sample
      const uint8_t* xb = reinterpret_cast<const uint8_t*>(x) + (first + c_lo) * G::HW;
#pragma unroll
      for (int q = 0; q < RI; ++q) {
        const int e = min(tid + 256 * q, NW - 1);
        const int cl = e / (NR * WPR), rem = e - cl * (NR * WPR), rr = rem / WPR, wd = rem - rr * WPR;
        iraw[q] = *reinterpret_cast<const unsigned*>(xb + ((int64_t)min(cl, nch - 1) * G::H + ir0 + rr) * G::H + 4 * wd);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (CIN) {      // (the frame rows above are in flight while this workgroup waits for its sample's gradient)
        mega_wait(hook.sync(bi));
        request_dy(std::true_type{});
      }
      for (int i = tid; i < DYF; i += 256) dyl[i] = 0.f;        // pad columns of the transposed gradient
#pragma unroll
      for (int q = 0; q < RI; ++q) {
        const int e = tid + 256 * q;
        unsigned v = iraw[q];
        asm volatile("" : "+v"(v));
        if (e < NW) {
          const int cl = e / (NR * WPR), rem = e - cl * (NR * WPR), rr = rem / WPR, wd = rem - rr * WPR;
          float* d = img + cl * CS + rr * RW + 4 * wd;
#pragma unroll
          for (int b = 0; b < 4; ++b) d[b] = (float)((double)((v >> (8 * b)) & 0xffu) * coef);
        }
      }
      // pad columns [H, RW) of every row and the tail: zero (finite values for the pad slot's A operand)
      constexpr int PADW = RW - G::H;
      if (PADW > 0) {
        for (int e = tid; e < NCH * NR * PADW; e += 256) {
          const int rowi = e / PADW, pc = e - rowi * PADW, cl = rowi / NR, rr = rowi - cl * NR;
          img[cl * CS + rr * RW + G::H + pc] = 0.f;
        }
      }
      if constexpr (CSPAD > 0) for (int e = tid; e < NCH * CSPAD; e += 256) img[(e / CSPAD) * CS + NR * RW + e % CSPAD] = 0.f;
      for (int e = tid; e < RW; e += 256) img[NCH * CS + e] = 0.f;
      __syncthreads();                                           // zero fill of dyl complete before the gradient lands
    } else {
      // image: per channel the chunk's NR input rows are one contiguous run of NR*H floats (the whole image for conv2 /
      // conv3, where a chunk is a whole sample)
      constexpr int RUNI = NR * G::H;
      constexpr bool V4 = (RUNI % 4 == 0) && (G::HW % 4 == 0) && (G::H % 4 == 0);   // conv3 (9x9 images): dwords
      constexpr int VPC = V4 ? RUNI / 4 : RUNI;                   // loads per channel
      constexpr int NVI = NCH * VPC, RI = (NVI + 255) / 256;
      const float* xf = reinterpret_cast<const float*>(x) + ((int64_t)bi * G::C + c_lo) * G::HW + (int64_t)ir0 * G::H;
      float4 iraw4[V4 ? RI : 1];
      float iraw1[V4 ? 1 : RI];
#pragma unroll
      for (int q = 0; q < RI; ++q) {
        const int f = min(tid + 256 * q, nch * VPC - 1);
        const int cl = f / VPC, v = f - cl * VPC;
        if constexpr (V4) iraw4[q] = *reinterpret_cast<const float4*>(xf + (int64_t)cl * G::HW + 4 * v);
        else iraw1[q] = xf[(int64_t)cl * G::HW + v];
      }
      __builtin_amdgcn_sched_barrier(0);
      // zero everything (image padding + transposed-gradient pad columns) while the loads are in flight
      for (int i = tid; i < (IMG + DYF + 3) / 4; i += 256) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < RI; ++q) {
        const int f = tid + 256 * q;
        if (f < nch * VPC) {
          const int cl = f / VPC, v = f - cl * VPC;
          if constexpr (V4) {
            float4 v4 = iraw4[q];
            const int r0 = 4 * v, rr = r0 / G::H, cc = r0 - rr * G::H;   // H % 4 == 0: a float4 never leaves its image row
            float* d = img + cl * CS + rr * RW + cc;
            d[0] = v4.x; d[1] = v4.y; d[2] = v4.z; d[3] = v4.w;
          } else {
            const int rr = v / G::H, cc = v - rr * G::H;
            img[cl * CS + rr * RW + cc] = iraw1[q];
          }
        }
      }
    }
    // transposed gradient: element (oc, ohl, ow) -> dyl[(ohl*OWP + ow)*LDB + oc]
#pragma unroll
    for (int q = 0; q < RD; ++q) {
      const int f = tid + 256 * q;
      float4 v4 = draw[q];
      asm volatile("" : "+v"(v4.x), "+v"(v4.y), "+v"(v4.z), "+v"(v4.w));
      if (f < NVD) {
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        int oc0, pos0;
        if constexpr (WHOLE) { oc0 = (4 * f) / G::P; pos0 = 4 * f - oc0 * G::P; }
        else { oc0 = f / (RUN / 4); pos0 = 4 * (f - oc0 * (RUN / 4)); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int p = pos0 + i, oc = oc0;
          if (WHOLE && p >= G::P) { p -= G::P; ++oc; }
          const int ohl = p / OH, ow = p - ohl * OH;
          dyl[(ohl * OWP + ow) * LDB + oc] = vv[i];
        }
      }
    }
    DRA_STAMP(TRR, 1);
    __syncthreads();
    DRA_STAMP(TRR, 2);
    // ---- MFMA: wave w owns tiles w, w+4, ...; tile t = (mt, nt), mt = t / NTL
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < TILES) {
        const int mt = tile / NTL, nt = tile - mt * NTL;
        const int k = k0 + mt * 32 + li;
        const int c = k / G::KK, kr = k - c * G::KK, kh = kr / G::KH, kw = kr - kh * G::KH;
        const float* ap = img + (c - c_lo) * CS + kh * RW + kw + h * S;
        const float* bp = dyl + h * LDB + nt * 32 + li;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int ohl = j / NPAIR, jw = j - ohl * NPAIR;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[ohl * S * RW + 2 * jw * S], bp[(ohl * OWP + 2 * jw) * LDB],
                                                       acc[t], 0, 0, 0);
        }
      }
    }
    if (grp == 0 && tid < G::OC) {  // bias gradient of this (sample, chunk): fixed-order column sum
      float sb = 0.f;
#pragma unroll 8
      for (int pos = 0; pos < NPOS; ++pos) sb += dyl[pos * LDB + tid];
      sb_acc += sb;
    }
    if (it + 1 < it1) __syncthreads();       // (persistent form) every wave has read this chunk's images
    }
    DRA_STAMP(TRR, 3);
    // ---- slab stores (rows = k, 32 lanes along oc: 128-byte rows)
    const int64_t slab = SHARES ? (int64_t)r : (int64_t)it0;
    float* dws = dw + slab * slab_stride;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < TILES) {
        const int mt = tile / NTL, nt = tile - mt * NTL;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int k = k0 + mt * 32 + mfma_row(rr, h);
          dws[(int64_t)k * G::OC + nt * 32 + li] = acc[t][rr];
        }
      }
    }
    if (grp == 0 && tid < G::OC) db[slab * slab_stride + tid] = sb_acc;
    DRA_STAMP(TRR, 5);
    DRA_STAMP_END(TRR);
  }
};

