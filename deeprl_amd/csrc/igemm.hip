// K2/K3 exports: NatureConvBody / FCBody / head contractions (templates in igemm.h).
#include "igemm.h"
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
  for (int b = 0; b < nb; ++b) {
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < KV; ++i) part += (lane + 64 * i < K) ? s_x[b * K + lane + 64 * i] * w[i] : 0.f;
    part = wave_sum(part);
    if (lane == 0) out[(int64_t)(b0 + b) * O + o] = act_apply(part + bias, act);
  }
}

// Wide layers at rollout batch sizes (fc4 of NatureConvBody, 3136 -> 512, for the 8 / 16 environments of one A2C / PPO rollout
// step): the K-chunked GEMM + its split-K finish were 13.2 + 5.4 us per rollout step (profiles/r02zw_kernel_stats_ppo_pixel_8.txt)
// for 6.4 MB of weights and 26 MFLOP.  Same shape as the device actor's fc4 GEMV: one wave per output row, the row in registers
// as R float4 per lane (all requested up front), the input rows staged in LDS eight at a time (100 KB at K = 3136) and shared
// by the workgroup's four rows; every load of the workgroup (its weight rows, its eight input rows) is requested before the
// first LDS write (a first version staged the input with a load -> store loop: six dependent round trips, slower than the
// GEMM it replaced).  grid (ceil(O / 4), nz, ceil(B / 8)).  K % 4 == 0.
template <int R>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))   // registers, not occupancy (100 KB of LDS: one workgroup per CU)
linear_gemv_wide_kernel(LinPtrs q, int B, int K, int O, int act) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];   // [<= 8 rows][K]
  float4* __restrict__ s_x4 = reinterpret_cast<float4*>(s_x);
  const int z = blockIdx.y, b0 = blockIdx.z * 8;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + wave;
  const int nv = K >> 2;
  const int nb = min(8, B - b0);
  const int n4 = nb * nv;
  const float4* __restrict__ w4 = reinterpret_cast<const float4*>(q.w[z] + (int64_t)min(o, O - 1) * K);
  const float4* __restrict__ src = reinterpret_cast<const float4*>(q.x[z] + (int64_t)b0 * K);
  // 8 rows x nv float4 over 256 threads: nv / 32 <= 2 R per thread (two arrays of R: one array of 2 R float4 is left in
  // scratch memory by the compiler's alloca promotion limit)
  float4 wv[R], xa[R], xb[R];
#pragma unroll
  for (int i = 0; i < R; ++i) wv[i] = w4[min(lane + 64 * i, nv - 1)];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    xa[i] = src[min((int)threadIdx.x + 256 * i, n4 - 1)];
    xb[i] = src[min((int)threadIdx.x + 256 * (R + i), n4 - 1)];
  }
  const float bias = q.bias[z] ? q.bias[z][min(o, O - 1)] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < R; ++i) {   // (clamped like the loads: a surplus thread rewrites the last element with the value it holds)
    const int e = (int)threadIdx.x + 256 * i;
    float4 va = xa[i], vb = xb[i];
    asm volatile("" : "+v"(va.x), "+v"(va.y), "+v"(va.z), "+v"(va.w));   // (member access: the arrays stay in registers)
    asm volatile("" : "+v"(vb.x), "+v"(vb.y), "+v"(vb.z), "+v"(vb.w));
    s_x4[min(e, n4 - 1)] = va;
    s_x4[min(e + 256 * R, n4 - 1)] = vb;
  }
  __syncthreads();
  if (o >= O) return;
  float* __restrict__ out = q.y[z];
  for (int b = 0; b < nb; ++b) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const float4 a = wv[i];
      const float4 xv = s_x4[b * nv + min(lane + 64 * i, nv - 1)];
      if (lane + 64 * i < nv) acc += (a.x * xv.x + a.y * xv.y) + (a.z * xv.z + a.w * xv.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[(int64_t)(b0 + b) * O + o] = act_apply(acc + bias, act);
  }
}

template <int R>
static int launch_gemv_wide(const LinPtrs& q, int nz, int batch, int in_features, int out_features, int act, hipStream_t st) {
  const size_t lds = (size_t)(batch < 8 ? batch : 8) * in_features * sizeof(float);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    DRA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_gemv_wide_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(linear_gemv_wide_kernel<R>, dim3((out_features + 3) / 4, nz, (batch + 7) / 8), dim3(256), lds, st, q, batch,
                     in_features, out_features, act);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

static int gemv_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DRA_LINEAR_GEMV"); v = e ? atoi(e) : 1; }
  return v;
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
      const int r = ((in_features >> 2) + 63) / 64;
      if (r <= 4) return launch_gemv_wide<4>(q, nz, batch, in_features, out_features, act, st);
      if (r <= 8) return launch_gemv_wide<8>(q, nz, batch, in_features, out_features, act, st);
      if (r <= 13) return launch_gemv_wide<13>(q, nz, batch, in_features, out_features, act, st);
      return launch_gemv_wide<16>(q, nz, batch, in_features, out_features, act, st);
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
