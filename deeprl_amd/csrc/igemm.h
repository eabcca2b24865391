// K2/K3: NatureConvBody / FCBody / head contractions as fp32 MFMA implicit GEMMs.
// Replaces the ATen conv/linear forward+backward behind deep_rl/network/network_bodies.py:10-33,
// :50-73 and the heads of network_heads.py (conv 4->32 k8 s4, 32->64 k4 s2, 64->64 k3 s1,
// fc 3136->512, head 512->A*{1,51,200}).
//
// Why fp32 MFMA: parity is against the reference's fp32 CPU path at 1e-5, so operands stay f32;
// v_mfma_f32_32x32x2_f32 is an exact fmaf chain at the fp32 vector peak (157 TF), 1/16 of bf16 MFMA.
//
// One kernel template computes  C[m][n] = sum_k A(m,k) * B(k,n)  for every layer and pass; a
// "problem" functor supplies A(m,k), B(k,n) (im2col / transposed / gathered on the fly, never
// materialised) and the epilogue store (bias, activation, activation-derivative mask, split-K
// slab).  Orientation is chosen so that the 32 lanes of an MFMA column index run along the
// memory-contiguous axis of the OUTPUT (positions for conv forward/dgrad, taps for wgrad), i.e.
// stores are 128-byte coalesced rows.
//
// Workgroup = 256 threads = 4 waves.  A BMxBN workgroup tile is (BM/32)*(BN/32) MFMA tiles; with
// fewer than 4 MFMA tiles the waves split each BK chunk among themselves (in-workgroup split-K,
// reduced through LDS) because at batch 32 the problem is latency-bound: short dependent MFMA
// chains and >=256 workgroups per launch matter more than tile reuse.  Operand tiles go global ->
// registers (prefetched one chunk ahead) -> LDS [k][m] / [k][n] with an odd row stride, so MFMA
// operand reads are conflict-free ds_read_b32 and the next chunk's loads fly under the MFMAs.
#pragma once
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_TANH) return tanhf(v);
  return v;
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad(float y, int act) {
  if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == ACT_TANH) return 1.f - y * y;
  return 1.f;
}

constexpr int kMaxZ = 4;

// LDS floats one workgroup of igemm_body<P> needs (operand tiles, reused for the in-workgroup fold)
template <class P>
constexpr int igemm_lds_floats() {
  constexpr int TILES = (P::BM / 32) * (P::BN / 32);
  constexpr int KS = TILES >= 4 ? 1 : 4 / TILES;
  constexpr int TILE_FLOATS = P::BK * (P::BM + 1 + P::BN + 1);
  constexpr int RED_FLOATS = (KS > 1) ? TILES * (KS - 1) * 1024 : 0;
  return TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS;
}

// A problem with `static constexpr bool SUMSQ = true` and a `double* partials` member also leaves the sum of squares of
// everything its workgroup stored in partials[bx + tiles * (by + gy * z)] (null = off): the gradient-norm pass of the
// optimizer then has nothing to read for that tensor (DRA_VAR_LATE_FOLD).
template <class P, class = void> struct igemm_sumsq : std::false_type {};
template <class P> struct igemm_sumsq<P, std::void_t<decltype(P::SUMSQ)>> : std::bool_constant<P::SUMSQ> {};

// The body is a device function of (tile, k-split, z) so that several independent problems can share
// ONE launch (multi_kernel below): every dependent launch costs ~4.5 us on this chip regardless of
// its work (profiles/r01_timeline_*), and a weight-gradient GEMM with ~150 workgroups leaves half
// of the CUs idle next to its input-gradient sibling.
template <class P>
__device__ __forceinline__ void igemm_body(const P& p, const int bx, const int by, const int z, const int gy,
                                           float* __restrict__ lds) {
  constexpr int BM = P::BM, BN = P::BN, BK = P::BK;
  constexpr int TM = BM / 32, TN = BN / 32, TILES = TM * TN;
  constexpr int KS = TILES >= 4 ? 1 : 4 / TILES;   // in-workgroup split of each BK chunk
  constexpr int TPW = TILES >= 4 ? TILES / 4 : 1;  // MFMA tiles per wave
  static_assert(TILES == 1 || TILES == 2 || (TILES % 4) == 0, "tile count");
  static_assert((BK / KS) % 2 == 0, "BK per wave must be even (32x32x2 MFMA)");
  static_assert((BM * BK) % 256 == 0 && (BN * BK) % 256 == 0, "tile loads must divide over 256 threads");
  constexpr int LDA = BM + 1, LDB = BN + 1;
  constexpr int RA = BM * BK / 256, RB = BN * BK / 256;
  float* As = lds;
  float* Bs = lds + BK * LDA;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int M = p.M, N = p.N, K = p.K;
  const int tiles_n = (N + BN - 1) / BN;
  const int bm = bx / tiles_n, bn = bx - bm * tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  // inter-workgroup split-K over gy parts, in whole BK chunks
  const int chunks = (K + BK - 1) / BK;
  const int cper = (chunks + gy - 1) / gy;
  const int k_begin = by * cper * BK;
  const int k_end = min(K, k_begin + cper * BK);

  // Operand fetch is split in two so that ALL global loads of a chunk issue back to back and stay
  // in flight under the previous chunk's MFMAs:
  //   fetch(): unconditional loads from clamped (always valid) addresses + a 2-bit code per
  //            element (0 -> 0.0, 1 -> loaded value, 2 -> 1.0) kept in a bit mask;
  //   stash(): the select happens here, behind an empty `asm volatile` that makes the loaded value
  //            opaque -- otherwise LLVM sinks each load under its bounds condition and emits a
  //            branch + s_waitcnt vmcnt(0) per element (serialised L2 round trips, 10x slower).
  float ra[RA], rb[RB];
  unsigned ca = 0, cb = 0;
  static_assert(RA <= 16 && RB <= 16, "2-bit codes live in one 32-bit mask per operand");
  auto fetch = [&](int k0) {
    ca = 0; cb = 0;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int e = tid + 256 * r;
      const int kk = P::A_KFAST ? (e % BK) : (e / BM);
      const int mm = P::A_KFAST ? (e / BK) : (e % BM);
      const int m = m0 + mm, k = k0 + kk;
      int code = 1;
      ra[r] = p.a(z, min(m, M - 1), min(k, K - 1), code);
      code = (m < M && k < k_end) ? code : 0;
      ca |= (unsigned)code << (2 * r);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int e = tid + 256 * r;
      const int kk = P::B_KFAST ? (e % BK) : (e / BN);
      const int nn = P::B_KFAST ? (e / BK) : (e % BN);
      const int n = n0 + nn, k = k0 + kk;
      int code = 1;
      rb[r] = p.b(z, min(k, K - 1), min(n, N - 1), code);
      code = (n < N && k < k_end) ? code : 0;
      cb |= (unsigned)code << (2 * r);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int e = tid + 256 * r;
      const int kk = P::A_KFAST ? (e % BK) : (e / BM);
      const int mm = P::A_KFAST ? (e / BK) : (e % BM);
      float v = ra[r];
      asm volatile("" : "+v"(v));
      const unsigned code = (ca >> (2 * r)) & 3u;
      As[kk * LDA + mm] = code == 1u ? v : (code == 2u ? 1.f : 0.f);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int e = tid + 256 * r;
      const int kk = P::B_KFAST ? (e % BK) : (e / BN);
      const int nn = P::B_KFAST ? (e / BK) : (e % BN);
      float v = rb[r];
      asm volatile("" : "+v"(v));
      v = p.fin_b(v);
      const unsigned code = (cb >> (2 * r)) & 3u;
      Bs[kk * LDB + nn] = code == 1u ? v : (code == 2u ? 1.f : 0.f);
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int tile0 = (KS == 1) ? wave * TPW : (wave % TILES);
  const int kpart = (KS == 1) ? 0 : (wave / TILES);
  constexpr int KW = BK / KS;  // k extent per wave per chunk
  const int lk = lane >> 5, li = lane & 31;

  if (k_begin < k_end) {
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
      stash();
      __syncthreads();
      if (k0 + BK < k_end) fetch(k0 + BK);  // next chunk's global loads fly under the MFMAs
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tile = tile0 + t;
        const int tm = tile / TN, tn = tile - tm * TN;
        const float* ap = As + (kpart * KW + lk) * LDA + tm * 32 + li;
        const float* bp = Bs + (kpart * KW + lk) * LDB + tn * 32 + li;
#pragma unroll
        for (int kk = 0; kk < KW; kk += 2)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * LDA], bp[kk * LDB], acc[t], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  if (KS > 1) {  // fold the in-workgroup k-parts: parts 1.. park in LDS, part 0 adds them in order
    float* red = lds;
    if (kpart > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((tile0 * (KS - 1) + (kpart - 1)) * 16 + r) * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (kpart == 0) {
#pragma unroll
      for (int q = 0; q < KS - 1; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += red[((tile0 * (KS - 1) + q) * 16 + r) * 64 + lane];
    }
  }
  [[maybe_unused]] float sq = 0.f;
  if (kpart == 0) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = tile0 + t;
      const int tm = tile / TN, tn = tile - tm * TN;
      const int n = n0 + tn * 32 + li;
      // per-output side inputs (bias / activation-derivative source) are loaded for all 16 rows
      // first, unconditionally from clamped addresses, so they overlap instead of serialising
      // behind the bounds checks of the stores
      float aux[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;  // 32x32 MFMA C/D row map
        aux[r] = p.aux(z, min(m, M - 1), min(n, N - 1));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(aux[r]));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < M && n < N) {
          p.store(z, by, m, n, acc[t][r], aux[r]);
          if constexpr (igemm_sumsq<P>::value) sq += acc[t][r] * acc[t][r];
        }
      }
    }
  }
  if constexpr (igemm_sumsq<P>::value) {
    if (p.partials) {   // (uniform) fixed-order workgroup sum: lanes by butterfly, waves (w0 + w1) + (w2 + w3)
      const double d = wave_sum((double)sq);
      __syncthreads();
      double* dl = reinterpret_cast<double*>(lds);
      if (lane == 0) dl[wave] = d;
      __syncthreads();
      if (tid == 0) {
        const int tiles = ((M + BM - 1) / BM) * tiles_n;
        p.partials[bx + tiles * (by + gy * z)] = (dl[0] + dl[1]) + (dl[2] + dl[3]);
      }
    }
  }
}

template <class P>
__global__ void __launch_bounds__(256) igemm_kernel(const P p) {
  __shared__ float lds[igemm_lds_floats<P>()];
  igemm_body<P>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, lds);
}

template <class P>
static int launch_igemm(const P& p, int nz, int ksplit, hipStream_t st) {
  const int tiles = ((p.M + P::BM - 1) / P::BM) * ((p.N + P::BN - 1) / P::BN);
  if (tiles < 1 || nz < 1 || ksplit < 1) return DRA_EINVAL;
  hipLaunchKernelGGL(igemm_kernel<P>, dim3(tiles, ksplit, nz), dim3(256), 0, st, p);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// =============================================================================================
// Convolution problems (square images / kernels, no padding), NCHW f32 activations; the first
// layer may read uint8 frames and normalise on the fly: f32(f64(v) * coef) -- bit-identical to
// the reference's `coef * np.asarray(x)` then float32 cast (normalizer.py:58-61, torch_utils.py:23).
template <int C_, int H_, int OC_, int KH_, int S_>
struct ConvGeom {
  static constexpr int C = C_, H = H_, OC = OC_, KH = KH_, S = S_;
  static constexpr int OH = (H - KH) / S + 1, P = OH * OH, KK = KH * KH, K = C * KK, HW = H * H;
};

struct ConvPtrs {
  const void* x[kMaxZ];
  const float* w[kMaxZ];
  const float* bias[kMaxZ];
  float* y[kMaxZ];
};

// Raw element fetch: f32 value, or (U8) the byte's integer bits parked in a float register so that
// the f64 normalisation runs at LDS-stash time, off the load's critical path (see conv_fin).
template <class G, bool U8>
__device__ __forceinline__ float conv_in(const void* x, int bi, int c, int ih, int iw) {
  const int off = ((bi * G::C + c) * G::H + ih) * G::H + iw;
  if (U8) return __uint_as_float((unsigned)reinterpret_cast<const uint8_t*>(x)[off]);
  return reinterpret_cast<const float*>(x)[off];
}
template <bool U8>
__device__ __forceinline__ float conv_fin(float raw, double coef) {
  return U8 ? (float)((double)__float_as_uint(raw) * coef) : raw;
}

// forward: Y[b][oc][p] = act(bias[oc] + sum_k W[oc][k] * Xcol[k][(b,p)])   M=OC, N=B*P, K=C*KH*KH
template <class G, int BM_, int BN_, int BK_, bool U8>
struct ConvFwd {
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int M, N, K;
  ConvPtrs q;
  int act;
  double coef;
  __device__ __forceinline__ float fin_b(float raw) const { return conv_fin<U8>(raw, coef); }
  __device__ __forceinline__ float a(int z, int m, int k, int&) const { return q.w[z][m * G::K + k]; }
  __device__ __forceinline__ float b(int z, int k, int n, int&) const {
    const int bi = n / G::P, pp = n - bi * G::P, oh = pp / G::OH, ow = pp - oh * G::OH;
    const int c = k / G::KK, kr = k - c * G::KK, kh = kr / G::KH, kw = kr - kh * G::KH;
    return conv_in<G, U8>(q.x[z], bi, c, oh * G::S + kh, ow * G::S + kw);
  }
  __device__ __forceinline__ float aux(int z, int m, int) const { return q.bias[z][m]; }
  __device__ __forceinline__ void store(int z, int, int m, int n, float v, float bias) const {
    const int bi = n / G::P, pp = n - bi * G::P;
    q.y[z][(bi * G::OC + m) * G::P + pp] = act_apply(v + bias, act);
  }
};

// weight gradient: dW[oc][k] = sum_(b,p) dY[b][oc][p] * Xcol[k][(b,p)]; column k == K carries the
// bias gradient (Xcol := 1).  M=OC, N=K+1, Kdim=B*P, split over gridDim.y into slabs.
template <class G, int BM_, int BN_, int BK_, bool U8>
struct ConvWgrad {
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int M, N, K;
  const float* dy;   // [B][OC][P] gradient w.r.t. the layer's PRE-activation
  const void* x;     // layer input
  float* dw;         // slab 0 of the weight gradient  [OC][K]
  float* db;         // slab 0 of the bias gradient    [OC]
  int64_t slab_stride;
  double coef;
  __device__ __forceinline__ float fin_b(float raw) const { return conv_fin<U8>(raw, coef); }
  __device__ __forceinline__ float a(int, int m, int k, int&) const {
    const int bi = k / G::P, pp = k - bi * G::P;
    return dy[(bi * G::OC + m) * G::P + pp];
  }
  __device__ __forceinline__ float b(int, int k, int n, int& code) const {
    const int nc = min(n, G::K - 1);
    const int bi = k / G::P, pp = k - bi * G::P, oh = pp / G::OH, ow = pp - oh * G::OH;
    const int c = nc / G::KK, kr = nc - c * G::KK, kh = kr / G::KH, kw = kr - kh * G::KH;
    code = (n == G::K) ? 2 : 1;  // bias column: Xcol := 1
    return conv_in<G, U8>(x, bi, c, oh * G::S + kh, ow * G::S + kw);
  }
  __device__ __forceinline__ float aux(int, int, int) const { return 0.f; }
  __device__ __forceinline__ void store(int, int ks, int m, int n, float v, float) const {
    if (n == G::K) db[ks * slab_stride + m] = v;
    else dw[ks * slab_stride + m * G::K + n] = v;
  }
};

// input gradient, stride-phase decomposed: phase z = (ph, pw) covers the input pixels with
// ih % S == ph, iw % S == pw; only taps kh = S*kh2 + ph, kw = S*kw2 + pw reach them, so every
// phase is a dense correlation (no MFMA work on structural zeros).
//   dXpre[b][c][ih][iw] = act'(X[b][c][ih][iw]) * sum_(oc,kh2,kw2) dY[b][oc][ih2-kh2][iw2-kw2] * W[oc][c][kh][kw]
// M=C, N=B*HP*HP (HP = ceil(H/S) positions per phase axis), Kdim=OC*KP*KP (KP = ceil(KH/S)).
template <class G, int BM_, int BN_, int BK_>
struct ConvDgrad {
  __device__ __forceinline__ float fin_b(float raw) const { return raw; }
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = true, B_KFAST = false;
  static constexpr int HP = (G::H + G::S - 1) / G::S, KP = (G::KH + G::S - 1) / G::S, PP = HP * HP, KPP = KP * KP;
  int M, N, K;
  const float* dy;    // [B][OC][OH][OH] pre-activation gradient of this layer's output
  const float* w;     // [OC][C][KH][KH]
  const float* xact;  // this layer's INPUT activations (post-activation output of the layer below)
  float* dx;          // [B][C][H][H] gradient w.r.t. the pre-activation of the layer below
  int act;
  __device__ __forceinline__ float a(int z, int m, int k, int& code) const {
    const int ph = z / G::S, pw = z - ph * G::S;
    const int oc = k / KPP, kr = k - oc * KPP, kh2 = kr / KP, kw2 = kr - kh2 * KP;
    const int kh = kh2 * G::S + ph, kw = kw2 * G::S + pw;
    code = (kh >= G::KH || kw >= G::KH) ? 0 : 1;
    return w[((oc * G::C + m) * G::KH + min(kh, G::KH - 1)) * G::KH + min(kw, G::KH - 1)];
  }
  __device__ __forceinline__ float b(int z, int k, int n, int& code) const {
    const int bi = n / PP, pp = n - bi * PP, ih2 = pp / HP, iw2 = pp - ih2 * HP;
    const int oc = k / KPP, kr = k - oc * KPP, kh2 = kr / KP, kw2 = kr - kh2 * KP;
    const int oh = ih2 - kh2, ow = iw2 - kw2;
    const int ohc = min(max(oh, 0), G::OH - 1), owc = min(max(ow, 0), G::OH - 1);
    code = (oh < 0 || ow < 0 || oh >= G::OH || ow >= G::OH) ? 0 : 1;  // taps that fall outside the output
    return dy[((bi * G::OC + oc) * G::OH + ohc) * G::OH + owc];
  }
  __device__ __forceinline__ int out_off(int z, int m, int n, bool& inside) const {
    const int ph = z / G::S, pw = z - ph * G::S;
    const int bi = n / PP, pp = n - bi * PP, ih2 = pp / HP, iw2 = pp - ih2 * HP;
    const int ih = ih2 * G::S + ph, iw = iw2 * G::S + pw;
    inside = ih < G::H && iw < G::H;
    return ((bi * G::C + m) * G::H + min(ih, G::H - 1)) * G::H + min(iw, G::H - 1);
  }
  // raw activation value; a null xact reads dx instead (always mapped, value ignored) so the load
  // stays branch-free -- the derivative itself is pure ALU work in store()
  __device__ __forceinline__ float aux(int z, int m, int n) const {
    bool inside;
    const int off = out_off(z, m, n, inside);
    const float* src = xact ? xact : dx;
    return src[off];
  }
  __device__ __forceinline__ void store(int z, int, int m, int n, float v, float y) const {
    bool inside;
    const int off = out_off(z, m, n, inside);
    if (inside) dx[off] = xact ? v * act_grad(y, act) : v;
  }
};

using G1 = ConvGeom<4, 84, 32, 8, 4>;   // 84x84x4  -> 20x20x32
using G2 = ConvGeom<32, 20, 64, 4, 2>;  // 20x20x32 -> 9x9x64
using G3 = ConvGeom<64, 9, 64, 3, 1>;   // 9x9x64   -> 7x7x64

// ---- KOC weight layout ([K=(c,kh,kw)][OC], see conv_v2.hip): gradients and input gradients for the
// fused learner, whose flat parameter buffer keeps the conv weights in that layout.
// weight gradient, KOC: dWt[k][oc] = sum_(b,p) Xcol[k][(b,p)] * dY[b][oc][p]; row k == K carries db.
// M = K+1, N = OC, Kdim = B*P: the 32 MFMA lanes run along oc, so slab stores are coalesced.
template <class G, int BM_, int BN_, int BK_, bool U8>
struct ConvWgradKoc {
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = true, B_KFAST = true;
  int M, N, K;
  const float* dy;
  const void* x;
  float* dw;
  float* db;
  int64_t slab_stride;
  double coef;
  __device__ __forceinline__ float fin_b(float raw) const { return raw; }
  __device__ __forceinline__ float a(int, int m, int k, int& code) const {
    const int mc = min(m, G::K - 1);
    const int bi = k / G::P, pp = k - bi * G::P, oh = pp / G::OH, ow = pp - oh * G::OH;
    const int c = mc / G::KK, kr = mc - c * G::KK, kh = kr / G::KH, kw = kr - kh * G::KH;
    code = (m == G::K) ? 2 : 1;
    // the uint8 normalisation is applied here (A operand has no stash-time hook): still one load
    const float raw = conv_in<G, U8>(x, bi, c, oh * G::S + kh, ow * G::S + kw);
    return conv_fin<U8>(raw, coef);
  }
  __device__ __forceinline__ float b(int, int k, int n, int&) const {
    const int bi = k / G::P, pp = k - bi * G::P;
    return dy[(bi * G::OC + n) * G::P + pp];
  }
  __device__ __forceinline__ float aux(int, int, int) const { return 0.f; }
  __device__ __forceinline__ void store(int, int ks, int m, int n, float v, float) const {
    if (m == G::K) db[ks * slab_stride + n] = v;
    else dw[ks * slab_stride + m * G::OC + n] = v;
  }
};

// input gradient, KOC weights: the reduction index is ordered (kh2, kw2, oc) with oc fastest so that
// consecutive k read consecutive weights.
template <class G, int BM_, int BN_, int BK_>
struct ConvDgradKoc {
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = true, B_KFAST = false;
  static constexpr int HP = (G::H + G::S - 1) / G::S, KP = (G::KH + G::S - 1) / G::S, PP = HP * HP, KPP = KP * KP;
  int M, N, K;
  const float* dy;
  const float* wt;    // [K][OC]
  const float* xact;
  float* dx;
  int act;
  __device__ __forceinline__ float fin_b(float raw) const { return raw; }
  __device__ __forceinline__ float a(int z, int m, int k, int& code) const {
    const int ph = z / G::S, pw = z - ph * G::S;
    const int tap = k / G::OC, oc = k - tap * G::OC, kh2 = tap / KP, kw2 = tap - kh2 * KP;
    const int kh = kh2 * G::S + ph, kw = kw2 * G::S + pw;
    code = (kh >= G::KH || kw >= G::KH) ? 0 : 1;
    return wt[((m * G::KH + min(kh, G::KH - 1)) * G::KH + min(kw, G::KH - 1)) * G::OC + oc];
  }
  __device__ __forceinline__ float b(int z, int k, int n, int& code) const {
    const int bi = n / PP, pp = n - bi * PP, ih2 = pp / HP, iw2 = pp - ih2 * HP;
    const int tap = k / G::OC, oc = k - tap * G::OC, kh2 = tap / KP, kw2 = tap - kh2 * KP;
    const int oh = ih2 - kh2, ow = iw2 - kw2;
    const int ohc = min(max(oh, 0), G::OH - 1), owc = min(max(ow, 0), G::OH - 1);
    code = (oh < 0 || ow < 0 || oh >= G::OH || ow >= G::OH) ? 0 : 1;
    return dy[((bi * G::OC + oc) * G::OH + ohc) * G::OH + owc];
  }
  __device__ __forceinline__ int out_off(int z, int m, int n, bool& inside) const {
    const int ph = z / G::S, pw = z - ph * G::S;
    const int bi = n / PP, pp = n - bi * PP, ih2 = pp / HP, iw2 = pp - ih2 * HP;
    const int ih = ih2 * G::S + ph, iw = iw2 * G::S + pw;
    inside = ih < G::H && iw < G::H;
    return ((bi * G::C + m) * G::H + min(ih, G::H - 1)) * G::H + min(iw, G::H - 1);
  }
  __device__ __forceinline__ float aux(int z, int m, int n) const {
    bool inside;
    const int off = out_off(z, m, n, inside);
    const float* src = xact ? xact : dx;
    return src[off];
  }
  __device__ __forceinline__ void store(int z, int, int m, int n, float v, float y) const {
    bool inside;
    const int off = out_off(z, m, n, inside);
    if (inside) dx[off] = xact ? v * act_grad(y, act) : v;
  }
};


// =============================================================================================
// Linear layers  y[b][o] = act(bias[o] + sum_i x[b][i] * W[o][i])   (runtime sizes)
struct LinPtrs {
  const float* x[kMaxZ];
  const float* w[kMaxZ];
  const float* bias[kMaxZ];
  float* y[kMaxZ];
};

// forward: M=O, N=B, K=I.  ksplit == 1: bias + activation in the epilogue.  ksplit > 1: raw
// partial sums go to slabs [z][ks][B][O] and dra_linear_fwd finishes with linear_finish_kernel.
template <int BM_, int BN_, int BK_>
struct LinFwd {
  __device__ __forceinline__ float fin_b(float raw) const { return raw; }
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = true, B_KFAST = true;
  int M, N, K;
  LinPtrs q;
  float* slabs;
  int ksplit, act;
  __device__ __forceinline__ float a(int z, int m, int k, int&) const { return q.w[z][(int64_t)m * K + k]; }
  __device__ __forceinline__ float b(int z, int k, int n, int&) const { return q.x[z][(int64_t)n * K + k]; }
  __device__ __forceinline__ float aux(int z, int m, int) const {
    const float* src = q.bias[z] ? q.bias[z] : q.w[z];  // null bias: any mapped address, value ignored
    return src[m];
  }
  __device__ __forceinline__ void store(int z, int ks, int m, int n, float v, float bias) const {
    if (ksplit > 1) slabs[((int64_t)(z * ksplit + ks) * N + n) * M + m] = v;
    else q.y[z][(int64_t)n * M + m] = act_apply(q.bias[z] ? v + bias : v, act);
  }
};

// weight gradient: dW[o][i] = sum_b dy[b][o] * x[b][i]; column i == I carries db.  M=O, N=I+1, K=B.
template <int BM_, int BN_, int BK_>
struct LinWgrad {
  __device__ __forceinline__ float fin_b(float raw) const { return raw; }
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = false, B_KFAST = false;
  int M, N, K;
  int I;
  const float* dy;
  const float* x;
  float* dw;
  float* db;
  __device__ __forceinline__ float a(int, int m, int k, int&) const { return dy[(int64_t)k * M + m]; }
  __device__ __forceinline__ float b(int, int k, int n, int& code) const {
    code = (n == I) ? 2 : 1;  // bias column
    return x[(int64_t)k * I + min(n, I - 1)];
  }
  __device__ __forceinline__ float aux(int, int, int) const { return 0.f; }
  __device__ __forceinline__ void store(int, int, int m, int n, float v, float) const {
    if (n == I) { if (db) db[m] = v; }
    else dw[(int64_t)m * I + n] = v;
  }
};

// the same problem, also leaving its workgroups' sums of squares (igemm_sumsq)
template <int BM_, int BN_, int BK_>
struct LinWgradSq : LinWgrad<BM_, BN_, BK_> {
  static constexpr bool SUMSQ = true;
  double* partials;
};

// input gradient: dxpre[b][i] = act'(xact[b][i]) * sum_o dy[b][o] * W[o][i].  M=B, N=I, K=O.
template <int BM_, int BN_, int BK_>
struct LinDgrad {
  __device__ __forceinline__ float fin_b(float raw) const { return raw; }
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool A_KFAST = true, B_KFAST = false;
  int M, N, K;
  const float* dy;
  const float* w;
  const float* xact;
  float* dx;
  int act;
  __device__ __forceinline__ float a(int, int m, int k, int&) const { return dy[(int64_t)m * K + k]; }
  __device__ __forceinline__ float b(int, int k, int n, int&) const { return w[(int64_t)k * N + n]; }
  __device__ __forceinline__ float aux(int, int m, int n) const {
    const float* src = xact ? xact : dx;
    return src[(int64_t)m * N + n];
  }
  __device__ __forceinline__ void store(int, int, int m, int n, float v, float y) const {
    dx[(int64_t)m * N + n] = xact ? v * act_grad(y, act) : v;
  }
};

