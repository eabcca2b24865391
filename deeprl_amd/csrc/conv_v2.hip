// NatureConvBody forward, second generation: ONE memory round trip per workgroup.
//
// The first-generation implicit GEMM (igemm.hip) walks K in chunks and pays a full global-memory
// round trip plus per-element im2col index arithmetic per chunk: 12-30 us per layer at batch 32,
// ~10 % of the fp32 MFMA rate (rocprofv3, profiles/r01_*).  At these sizes (<= 0.4 GFLOP, inputs
// that fit L2) latency, not bandwidth or FLOPs, is the enemy, so this kernel:
//   * assigns a workgroup 32 output positions of ONE sample x 32 output channels, all of K;
//   * stages exactly the input rows those positions touch into LDS once, with coalesced loads
//     (uint8 frames are normalised f32(f64(v)*coef) on the way in -- bit-exact with the reference);
//     columns are stored de-interleaved by (iw mod stride) so that the 32 lanes of an MFMA operand
//     read (consecutive output columns, stride S in the image) hit consecutive LDS banks;
//   * keeps the weights in a [K][OC] layout ("KOC": K = (c,kh,kw) major, output channel minor) and
//     loads each wave's A operands straight into registers, 128-byte coalesced, all issued up front
//     together with the image loads -- a single exposed memory latency per workgroup;
//   * splits K over the 4 waves by channel pairs; the two k-slices of a 32x32x2 MFMA are the SAME
//     tap of two adjacent channels, so every B operand is `ds_read_b32 base + immediate` in a
//     fully unrolled loop (no index arithmetic in the loop at all);
//   * reduces the 4 partial accumulators through LDS, adds bias, applies ReLU, stores coalesced.
// Numerics: fp32 MFMA (exact fmaf chains), summation order differs from igemm.hip / the CPU
// oracle only in the order of the four K-quarters.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C_, int H_, int OC_, int KH_, int S_>
struct V2Geom {
  static constexpr int C = C_, H = H_, OC = OC_, KH = KH_, S = S_;
  static constexpr int OH = (H - KH) / S + 1, P = OH * OH, KK = KH * KH, K = C * KK;
  static constexpr int TPS = (P + 31) / 32;                  // position tiles per sample
  static constexpr int WPH = (H + S - 1) / S;                // columns per stride phase
  static constexpr int RW = S * WPH;                         // LDS row width (>= H)
  // most output rows a 32-position tile can touch, and the input rows they need
  static constexpr int OROWS = (31 + OH - 1) / OH + 1;
  static constexpr int NR_RAW = (OROWS - 1) * S + KH;
  static constexpr int NR = NR_RAW < H ? NR_RAW : H;
  static constexpr int CS = NR * RW;                         // LDS channel stride
  static constexpr int CP = C / 2;                           // channel pairs
  // K split over 4 waves: whole channel pairs when there are >= 4, else pairs x tap halves
  static constexpr int CPW = CP >= 4 ? CP / 4 : 1;           // channel pairs per wave
  static constexpr int TSPLIT = CP >= 4 ? 1 : 4 / CP;        // tap-range splits
  static constexpr int TW = KK / TSPLIT;                     // taps per wave
  static constexpr int NJ = CPW * TW;                        // MFMAs per wave
  static_assert(C % 2 == 0 && (CP >= 4 ? CP % 4 == 0 : (4 % CP == 0 && KK % TSPLIT == 0)), "K split");
  static_assert(TW % KH == 0, "a wave's tap range starts on a kernel-row boundary");
  static_assert(OC % 32 == 0, "OC tiles");
};

struct ConvV2Args {
  const void* x[DRA_MAX_Z];
  const float* wt[DRA_MAX_Z];    // [K][OC]
  const float* bias[DRA_MAX_Z];
  float* y[DRA_MAX_Z];
  int batch, act;
  double coef;
  // ring-direct input (batch 1, uint8 frames): channel c of the stack is ring frame
  // (*ring_slot - (C-1) + c) mod ring_cap of the slot-major frame array x[0]; null = plain NCHW input
  const int64_t* ring_slot;
  int64_t ring_cap;
};

__device__ __forceinline__ float v2_act(float v, int act) {
  if (act == DRA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DRA_ACT_TANH) return tanhf(v);
  return v;
}

template <class G>
__device__ __forceinline__ int lds_col(int iw) { return (iw % G::S) * G::WPH + iw / G::S; }

template <class G, bool U8>
__global__ void __launch_bounds__(256) conv_fwd_v2_kernel(const ConvV2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [C][NR][RW] image, then reused for the reduction
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int z = blockIdx.z;
  const int bi = blockIdx.x / G::TPS, tile = blockIdx.x - bi * G::TPS;
  const int oc0 = blockIdx.y * 32;
  const int p0 = tile * 32;
  const int np = min(32, G::P - p0);
  const int oh0 = p0 / G::OH, oh1 = (p0 + np - 1) / G::OH;
  const int ir0 = oh0 * G::S;
  const int nrows = (oh1 - oh0) * G::S + G::KH;  // <= G::NR

  // ---- issue every global load of this workgroup: weight operands first, then the image rows
  const float* __restrict__ wt = a.wt[z];
  const int cp0 = (G::CP >= 4) ? wave * G::CPW : (wave % G::CP);
  const int t0 = (G::CP >= 4) ? 0 : (wave / G::CP) * G::TW;
  float areg[G::NJ];
  {
    const float* wbase = wt + ((int64_t)(2 * cp0 + h) * G::KK + t0) * G::OC + oc0 + li;
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
      const int cpl = j / G::TW, t = j - cpl * G::TW;
      areg[j] = wbase[(2 * cpl * G::KK + t) * G::OC];
    }
  }
  if (U8) {
    // rows are 84 bytes: the [nrows x 84] block of a channel is contiguous and 4-byte aligned
    constexpr int WPR = G::H / 4;                               // u32 words per row
    constexpr int NWMAX = G::NR * WPR;
    constexpr int LPT = (NWMAX + 63) / 64;                      // words per lane per channel
    constexpr int CPT = (G::C + 3) / 4;                         // channels per wave
    unsigned raw[CPT * LPT];
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(a.x[z]);
    const int nw = nrows * WPR;
    const int64_t newest = a.ring_slot ? *a.ring_slot : 0;
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + 4 * ci;
      int64_t img = (int64_t)bi * G::C + min(c, G::C - 1);       // image index in a plain NCHW batch
      if (a.ring_slot) {
        img = newest - (G::C - 1) + min(c, G::C - 1);
        if (img < 0) img += a.ring_cap;
      }
      const unsigned* src = reinterpret_cast<const unsigned*>(xb + (img * G::H + ir0) * G::H);
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        const int e = lane + 64 * q;
        raw[ci * LPT + q] = src[min(e, nw - 1)];
      }
    }
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + 4 * ci;
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        const int e = lane + 64 * q;
        unsigned v = raw[ci * LPT + q];
        asm volatile("" : "+v"(v));  // keep the loads unconditional and batched (see igemm.hip)
        if (e < nw && c < G::C) {
          const int r = e / WPR, iw = (e - r * WPR) * 4;
          float* dst = lds + c * G::CS + r * G::RW;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            dst[lds_col<G>(iw + b)] = (float)((double)((v >> (8 * b)) & 0xffu) * a.coef);
        }
      }
    }
  } else {
    constexpr int NEMAX = G::NR * G::H;                          // floats per channel block
    constexpr int LPT = (NEMAX + 63) / 64;
    constexpr int CPT = (G::C + 3) / 4;
    float raw[CPT * LPT];
    const float* xf = reinterpret_cast<const float*>(a.x[z]);
    const int ne = nrows * G::H;
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + 4 * ci;
      const float* src = xf + ((int64_t)(bi * G::C + min(c, G::C - 1)) * G::H + ir0) * G::H;
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        const int e = lane + 64 * q;
        raw[ci * LPT + q] = src[min(e, ne - 1)];
      }
    }
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + 4 * ci;
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        const int e = lane + 64 * q;
        float v = raw[ci * LPT + q];
        asm volatile("" : "+v"(v));
        if (e < ne && c < G::C) {
          const int r = e / G::H, iw = e - r * G::H;
          lds[c * G::CS + r * G::RW + lds_col<G>(iw)] = v;
        }
      }
    }
  }
  __syncthreads();

  // ---- MFMA: lane li owns output position p0 + li (clamped), half-wave h the odd channel of a pair
  const int pj = min(li, np - 1);
  const int poh = (p0 + pj) / G::OH, pow_ = (p0 + pj) - poh * G::OH;
  const float* bptr = lds + (2 * cp0 + h) * G::CS + ((poh - oh0) * G::S + t0 / G::KH) * G::RW + pow_;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int j = 0; j < G::NJ; ++j) {
    const int cpl = j / G::TW, t = j - cpl * G::TW;  // tap relative to the wave's base tap t0 (folded into bptr)
    const int kh = t / G::KH, kw = t - kh * G::KH;
    const float b = bptr[2 * cpl * G::CS + kh * G::RW + (kw % G::S) * G::WPH + kw / G::S];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[j], b, acc, 0, 0, 0);
  }
  __syncthreads();  // every wave is done reading the image: reuse LDS for the 4-way reduction

  float* red = lds;  // [4 waves][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  // wave w finalises accumulator registers 4w .. 4w+3 (MFMA C/D rows (r&3) + 8*(r>>2) + 4*h)
  float* __restrict__ y = a.y[z];
  const float* __restrict__ bias = a.bias[z];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = wave * 4 + q;
    const float s = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) +
                    (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const float v = v2_act(s + bias[oc0 + row], a.act);
    if (li < np) y[((int64_t)(bi * G::OC + oc0 + row)) * G::P + p0 + li] = v;
  }
}

using VG1 = V2Geom<4, 84, 32, 8, 4>;
using VG2 = V2Geom<32, 20, 64, 4, 2>;
using VG3 = V2Geom<64, 9, 64, 3, 1>;

template <class G, bool U8>
static int launch_conv_v2(const ConvV2Args& a, int nz, hipStream_t st) {
  constexpr size_t img = (size_t)G::C * G::CS * sizeof(float);
  constexpr size_t red = (size_t)4 * 16 * 64 * sizeof(float);
  constexpr size_t bytes = img > red ? img : red;
  static bool attr_set = false;
  if (bytes > 64 * 1024 && !attr_set) {
    DRA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_v2_kernel<G, U8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_fwd_v2_kernel<G, U8>), dim3(G::TPS * a.batch, G::OC / 32, nz), dim3(256), bytes, st, a);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Same contract as dra_conv_fwd, but the weights are in the KOC layout: wt[(c*KH+kh)*KH+kw][oc].
DRA_API int dra_conv_fwd_koc(int layer, int nz, const void* const* x, const float* const* wt, const float* const* bias,
                             float* const* y, int batch, int x_is_u8, double u8_coef, int act, void* stream) {
  if (nz < 1 || nz > DRA_MAX_Z || batch < 1 || !x || !wt || !bias || !y) return DRA_EINVAL;
  ConvV2Args a;
  for (int z = 0; z < nz; ++z) {
    if (!x[z] || !wt[z] || !bias[z] || !y[z]) return DRA_EINVAL;
    a.x[z] = x[z]; a.wt[z] = wt[z]; a.bias[z] = bias[z]; a.y[z] = y[z];
  }
  a.batch = batch; a.act = act; a.coef = u8_coef; a.ring_slot = nullptr; a.ring_cap = 0;
  hipStream_t st = dra_stream(stream);
  switch (layer) {
    case 1: return x_is_u8 ? launch_conv_v2<VG1, true>(a, nz, st) : launch_conv_v2<VG1, false>(a, nz, st);
    case 2: return x_is_u8 ? DRA_EINVAL : launch_conv_v2<VG2, false>(a, nz, st);
    case 3: return x_is_u8 ? DRA_EINVAL : launch_conv_v2<VG3, false>(a, nz, st);
  }
  return DRA_EINVAL;
}

// conv1 of the actor's batch-1 forward reading its 4-frame stack straight from the replay ring
// (DQN_agent.py:24-33: the state the actor acts on IS the newest `history` frames of the ring):
// frames = slot-major u8 ring, *newest_slot_dev = slot of the newest frame (device int64).
DRA_API int dra_conv1_fwd_koc_ring(const void* frames, const int64_t* newest_slot_dev, int64_t capacity, const float* wt,
                                   const float* bias, float* y, double u8_coef, int act, void* stream) {
  if (!frames || !newest_slot_dev || capacity < VG1::C || !wt || !bias || !y) return DRA_EINVAL;
  ConvV2Args a;
  a.x[0] = frames; a.wt[0] = wt; a.bias[0] = bias; a.y[0] = y;
  a.batch = 1; a.act = act; a.coef = u8_coef; a.ring_slot = newest_slot_dev; a.ring_cap = capacity;
  return launch_conv_v2<VG1, true>(a, 1, dra_stream(stream));
}

// Layout conversion [OC][K] <-> [K][OC] for one layer's weight tensor (tests, generic path, and
// checkpoint interchange; the fused learner keeps KOC as its master layout and never converts).
__global__ void __launch_bounds__(256)
transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  out[(int64_t)c * rows + r] = in[i];
}

DRA_API int dra_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
  if (!in || !out || rows < 1 || cols < 1) return DRA_EINVAL;
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dra_stream(stream), in, out, rows,
                     cols);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
