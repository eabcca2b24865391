// The one-pass backward roles of conv2 / conv3: operands staged into LDS exactly as they lie in memory (round 4).
//
// The round-2 forms of these roles (ConvDgradOne, ConvWgradOne<G2 / G3>; removed) scattered every staged element into a padded
// (input gradient) or transposed (weight gradient) LDS image: a constant division or two, a multiply-add and a ds_write_b32 per
// element -- 730 VALU instructions per wave around 64-90 MFMAs in conv2's backward launch (rocprofv3 SQ_INSTS_VALU /
// SQ_INSTS_MFMA = 10.1, profiles/r04a_sq_learner_b32.json), and a zero fill + an extra barrier in front of it.  Here the
// workgroup's loads are float4, its LDS writes ds_write_b128 of the same float4, no index arithmetic at all, and the geometry
// moves into the per-lane BASE ADDRESS of the MFMA operand reads, computed once per workgroup:
//   * input gradient: the gradient image dY[b] is [OC][P] as in memory.  A lane (position, tap) whose shifted read falls outside
//     the image does not read a zero PADDING cell; its base address points into a small block of zeros instead, so the loop
//     body stays `ds_read_b32 v, base + immediate` with no select;
//   * weight gradient: the reduction index is the output position p in memory order, two consecutive positions per MFMA
//     (k-slices h = 0 / 1).  P is odd (81, 49), so ONE MFMA of a tile carries a pad slot instead of one per output row
//     (41 instead of 45 MFMAs per tile for conv2, 25 instead of 28 for conv3).  dY stays [oc][P]: lane = oc reads have the odd
//     stride P -> conflict-free; the input image stays [c][H][H]: lane = tap reads are conflict-free for conv2's 4x4 taps.
// Same products in the same order as the padded / transposed forms (a pad slot contributes fma(a, 0, acc) = acc): the results
// were bit-identical with them on the device (profiles/r04d_ab_env.jsonl: same parameter sums), and tools/emulate_oneshot.py
// replays both index maps on the CPU.  conv1 keeps oneshot.h's ConvWgradOne (uint8 frames, read straight from the replay ring).
#pragma once
#include "oneshot.h"

// 16-byte staging registers as a NATIVE vector: an array of HIP's float4 (a struct) copied whole was left in scratch memory by
// the compiler (112-176 bytes of private segment per lane, scratch_load_dwordx4 in front of every LDS write)
typedef float lin_f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// Convolution input gradient with KOC weights, one pass (layers 2 and 3 of NatureConvBody).
// Stride-phase decomposition as in ConvDgradKoc: phase (ph, pw) covers input pixels ih = ih2*S + ph, iw = iw2*S + pw and only
// taps kh = kh2*S + ph, kw = kw2*S + pw reach them, so each phase is a dense KPxKP correlation over the output gradient:
//   dXpre[b][c][ih][iw] = act'(X[b][c][ih][iw]) * sum_(oc,kh2,kw2) dY[b][oc][ih2-kh2][iw2-kw2] * Wt[(c,kh,kw)][oc]   (0 outside)
// Workgroup = (sample, phase, PT x 32 positions of the phase, 32 input channels).  MFMA lane li of the B operand is a
// position: every B read is `base(position, tap) + immediate(oc)`.  The A operand (weights, lane li = input channel c) is read
// straight from the KOC tensor: slot (jj, h) of wave w <-> oc = 16w + 8h + jj, i.e. two float4 per tap.
// PT = consecutive 32-position tiles of one (sample, phase) per workgroup: the staged gradient image and the register-resident
// weights are shared by PT independent accumulation chains.  conv2 at batch 32 uses PT = 2: 256 instead of 512 workgroups, so
// that together with the weight-gradient workgroups of the same launch the grid fits the chip's workgroup slots in ONE round.
template <class G, int PT = 1>
struct ConvDgradLin {
  static constexpr int S = G::S, KP = (G::KH + S - 1) / S, NPH = S * S, HP = (G::H + S - 1) / S, PP = HP * HP;
  static constexpr int OH = G::OH, P = G::P;
  static constexpr int TPP = (PP + 31) / 32, TGP = (TPP + PT - 1) / PT;   // tiles / tile groups per phase
  static constexpr int OCW = G::OC / 4, OCH = OCW / 2, NT = KP * KP, NJ = NT * OCH;
  static constexpr int MT = G::C / 32;
  static constexpr int NSRC = G::OC * P;                                 // one sample's gradient image, as in memory
  static constexpr int ZERO = ((OCH - 1) * P + 1 + 3) & ~3;              // zeros read by (lane, tap) pairs outside the image
  static constexpr int IMG = NSRC + ZERO;
  static constexpr int RED = PT * 4096;
  static constexpr int LDS_FLOATS = IMG > RED ? IMG : RED;
  static_assert(G::KH % S == 0, "every stride phase has KP x KP taps");
  static_assert(G::OC % 32 == 0 && OCH % 4 == 0 && G::C % 32 == 0, "float4 weight runs per half-wave");
  static_assert(NSRC % 4 == 0, "float4 staging");
  const float* dy;    // [B][OC][OH][OH] pre-activation gradient of this layer's output
  const float* wt;    // [(c,kh,kw)][OC]
  const float* xact;  // [B][C][H][H] this layer's input (post-activation) or null
  float* dx;          // [B][C][H][H]
  int B, act;
  int xcd = 0;        // != 0: all workgroups of a sample on one XCD (xcd_order)
  ChainHook hook;     // DRA_VAR_BWD_CHAIN: this role's place in the chained backward launch (run_<CIN, COUT>)
  static constexpr int WGS_PER_SAMPLE = NPH * TGP * MT;
  __host__ int blocks() const { return B * NPH * TGP * MT; }
  __device__ __forceinline__ void run(int bid_, float* __restrict__ lds, int first = 0) const { run_<false, false>(bid_, lds, first); }
  // CIN: dy comes from workgroups of the SAME launch (wait for this sample's producers, agent-scope loads; the weights and the
  // activation-derivative source are requested in front of the wait); COUT: dx goes to workgroups of the same launch
  template <bool CIN, bool COUT>
  __device__ __forceinline__ void run_(int bid_, float* __restrict__ lds, int first = 0) const {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int bid = xcd ? xcd_order(bid_, first, B, NPH * TGP * MT) : bid_;
    const int mt = bid % MT;
    int r = bid / MT;
    const int grp = r % TGP;
    r /= TGP;
    const int phi = r % NPH, bi = r / NPH;
    const int ph = phi / S, pw = phi - ph * S;
    const int c0 = mt * 32, p0 = grp * PT * 32;
    const int np = min(32 * PT, PP - p0);
    [[maybe_unused]] constexpr int TRR = (G::C == 32) ? TR_CONV2_B : TR_CONV3_B;
    DRA_STAMP(TRR, 0);
    // ---- weights: lane li <-> input channel c0 + li
    float4 areg[NT][OCH / 4];
    {
      const float* wl = wt + (int64_t)(c0 + li) * G::KK * G::OC + wave * OCW + h * OCH;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int kh = (t / KP) * S + ph, kw = (t % KP) * S + pw;
#pragma unroll
        for (int v = 0; v < OCH / 4; ++v)
          areg[t][v] = *reinterpret_cast<const float4*>(wl + (kh * G::KH + kw) * G::OC + 4 * v);
      }
    }
    // ---- dY[bi] ([OC][P], contiguous, 16-byte aligned) -> registers as float4
    constexpr int NV = NSRC / 4, RV = (NV + 255) / 256;
    lin_f4 rawv[RV];
    const lin_f4* dyb4 = reinterpret_cast<const lin_f4*>(dy + (int64_t)bi * NSRC);
    if constexpr (!CIN) {
#pragma unroll
      for (int q = 0; q < RV; ++q) rawv[q] = dyb4[min(tid + 256 * q, NV - 1)];
    }
    // ---- epilogue side input (activation-derivative source), loaded with everything else; per (tile, tap) operand bases
    int pix[PT];
    bool inside[PT];
    float aux[PT][4];
    int boff[PT][NT];      // LDS float offset of this lane's B operand of tap t, slot jj = 0 (slot jj adds jj * P)
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int pj = min(32 * t + li, np - 1);
      const int ih2 = (p0 + pj) / HP, iw2 = (p0 + pj) - ih2 * HP;
      const int ih = ih2 * S + ph, iw = iw2 * S + pw;
      inside[t] = ih < G::H && iw < G::H;
      pix[t] = min(ih, G::H - 1) * G::H + min(iw, G::H - 1);
      const float* src = xact ? xact : dx;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + mfma_row(wave * 4 + q, h);
        aux[t][q] = src[((int64_t)bi * G::C + c) * G::HW + pix[t]];
      }
#pragma unroll
      for (int tp = 0; tp < NT; ++tp) {
        const int rr = ih2 - tp / KP, cc = iw2 - tp % KP;
        const bool ok = rr >= 0 && rr < OH && cc >= 0 && cc < OH;
        boff[t][tp] = ok ? (wave * OCW + h * OCH) * P + rr * OH + cc : NSRC;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] const MegaSync ms = hook.sync(bi);
    if constexpr (CIN) {
      mega_wait(ms);
#pragma unroll
      for (int q = 0; q < RV; ++q) rawv[q] = mega_ld4<true>(dyb4 + min(tid + 256 * q, NV - 1));
    }
    lin_f4* lds4 = reinterpret_cast<lin_f4*>(lds);
    for (int i = tid; i < ZERO / 4; i += 256) lds4[NSRC / 4 + i] = lin_f4{0.f, 0.f, 0.f, 0.f};
    // (unconditional: a lane past the end re-writes the LAST float4 with the same value it loaded from the clamped index)
#pragma unroll
    for (int q = 0; q < RV; ++q) lds4[min(tid + 256 * q, NV - 1)] = rawv[q];
    DRA_STAMP(TRR, 1);
    __syncthreads();
    DRA_STAMP(TRR, 2);
    // ---- MFMA: slot <-> (oc, tap) map as described above, taps outermost
    f32x16 acc[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[t] = zero16();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int jj = 0; jj < OCH; ++jj) {
        const float4 av = areg[t][jj / 4];
        const float a = (jj % 4 == 0) ? av.x : ((jj % 4 == 1) ? av.y : ((jj % 4 == 2) ? av.z : av.w));
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const float b = lds[boff[pt][t] + jj * P];
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[pt], 0, 0, 0);
        }
      }
    }
    DRA_STAMP(TRR, 3);
    __syncthreads();
    DRA_STAMP(TRR, 4);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      float s[4];
      reduce4(lds + t * 4096, acc[t], wave, lane, s);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + mfma_row(wave * 4 + q, h);
        if (32 * t + li < np && inside[t])
          mega_st<COUT>(&dx[((int64_t)bi * G::C + c) * G::HW + pix[t]], xact ? s[q] * act_grad(aux[t][q], act) : s[q]);
      }
    }
    DRA_STAMP(TRR, 5);
    if constexpr (COUT) mega_publish(ms);
    DRA_STAMP_END(TRR);
  }
};

// ------------------------------------------------------------------------------------------------
// Weight gradient, KOC layout, one pass, ONE SAMPLE per workgroup (conv2 / conv3: a chunk is the whole sample):
//   dWt[k][oc] = sum_p Xcol[k][p] * dY[oc][p],  k = (c,kh,kw);   db[oc] = sum_p dY[oc][p]      slab index = sample
// Workgroup = (sample b, group of MTG 32-row k tiles); M = k (lane li = tap), N = oc (lane li = output channel), reduction over
// the sample's P output positions in memory order, two per MFMA.  The channels the group's taps touch are one contiguous run
// of the input (copied as it lies, from a 16-byte aligned start: `shift` floats of the previous channel come along), the
// gradient is the sample's whole [OC][P] block.
template <class G, int MTG>
struct ConvWgradLin {
  static constexpr int S = G::S, OH = G::OH, P = G::P, H = G::H, HW = G::HW;
  static constexpr int NJ = (P + 1) / 2;                       // MFMAs per tile; the last one carries the pad slot when P is odd
  static constexpr bool ODD = (P & 1) != 0;
  static constexpr int MTILES = G::K / 32, NGRP = MTILES / MTG, NTL = G::OC / 32, TILES = MTG * NTL;
  static constexpr int TPW = (TILES + 3) / 4;
  static constexpr int NCHMAX = ((MTG * 32) % G::KK == 0) ? (MTG * 32) / G::KK : (MTG * 32 + G::KK - 2) / G::KK + 1;
  static constexpr int NCH = NCHMAX < G::C ? NCHMAX : G::C;
  static constexpr int IMGF = (NCH * HW + 3 + 3) & ~3;          // + up to 3 floats of alignment shift, whole float4s
  static constexpr int NSRC = G::OC * P;
  static constexpr int LDS_FLOATS = IMGF + NSRC + 4;            // + one zero float4: the pad slot's B operand
  static_assert(G::K % 32 == 0 && MTILES % MTG == 0, "tiling");
  static_assert(NSRC % 4 == 0 && (G::C * HW) % 4 == 0, "float4 staging; a sample starts on a float4");
  const float* dy;   // [B][OC][OH][OH]
  const void* x;     // [B][C][H][H] f32
  float* dw;         // slab 0 of dWt [K][OC]
  float* db;         // slab 0 of db [OC]
  int64_t slab_stride;
  int B;
  double coef;       // (unused: f32 inputs only; kept so that the role is built like ConvWgradOne)
  const int64_t* sample_idx = nullptr;   // (unused)
  int xcd = 0;        // != 0: all workgroups of a sample on one XCD (xcd_order)
  ChainHook hook;     // DRA_VAR_BWD_CHAIN (run_<CIN, COUT>: dy from / slabs to workgroups of the same launch)
  __host__ int blocks() const { return B * NGRP; }
  __host__ static int n_slabs(int batch) { return batch; }
  // LDS float offset of output position p's top-left input pixel inside a channel: (oh * S) * H + ow * S
  static constexpr int pos_off(int p) { return (p / OH) * S * H + (p % OH) * S; }
  __device__ __forceinline__ void run(int bid_, float* __restrict__ lds, int first = 0) const { run_<false, false>(bid_, lds, first); }
  template <bool CIN, bool COUT>
  __device__ __forceinline__ void run_(int bid_, float* __restrict__ lds, int first = 0) const {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int bid = xcd ? xcd_order(bid_, first, B, NGRP) : bid_;
    const int grp = bid % NGRP, bi = bid / NGRP;
    const int k0 = grp * MTG * 32;
    const int c_lo = k0 / G::KK;
    const int c_hi = min((k0 + MTG * 32 - 1) / G::KK, G::C - 1);
    const int nch = c_hi - c_lo + 1;                       // <= NCH
    float* img = lds;
    float* dyl = lds + IMGF;
    [[maybe_unused]] constexpr int TRR = (G::C == 32) ? TR_CONV2_B : TR_CONV3_B;
    DRA_STAMP(TRR, 0);
    // ---- every load of the workgroup: the sample's gradient block, then the input channels [c_lo, c_hi] as they lie
    constexpr int NVD = NSRC / 4, RD = (NVD + 255) / 256;
    lin_f4 draw[RD];
    const lin_f4* dyb4 = reinterpret_cast<const lin_f4*>(dy + (int64_t)bi * NSRC);
    if constexpr (!CIN) {
#pragma unroll
      for (int q = 0; q < RD; ++q) draw[q] = dyb4[min(tid + 256 * q, NVD - 1)];
    }
    const int64_t xstart = ((int64_t)bi * G::C + c_lo) * HW;          // first float of the run
    const int shift = (int)(xstart & 3);                               // floats between the aligned start and the run
    const int nvi = (nch * HW + shift + 3) >> 2;                        // float4s that cover it
    const int64_t xlast4 = ((int64_t)B * G::C * HW >> 2) - 1;            // last float4 of the tensor
    const lin_f4* x4 = reinterpret_cast<const lin_f4*>(x) + (xstart >> 2);
    constexpr int RI = (IMGF / 4 + 255) / 256;
    lin_f4 iraw[RI];
#pragma unroll
    for (int q = 0; q < RI; ++q) {
      const int64_t f = min((int64_t)(tid + 256 * q), (int64_t)nvi - 1);
      iraw[q] = x4[min(f, xlast4 - (xstart >> 2))];
    }
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] const MegaSync ms = hook.sync(bi);
    if constexpr (CIN) {      // (the input channels above are in flight while this workgroup waits for its sample's gradient)
      mega_wait(ms);
#pragma unroll
      for (int q = 0; q < RD; ++q) draw[q] = mega_ld4<true>(dyb4 + min(tid + 256 * q, NVD - 1));
    }
    lin_f4* img4 = reinterpret_cast<lin_f4*>(img);
    lin_f4* dyl4 = reinterpret_cast<lin_f4*>(dyl);
    // (unconditional stores: lanes past the end re-write the last float4 with the value they loaded from the clamped index)
#pragma unroll
    for (int q = 0; q < RI; ++q) img4[min(tid + 256 * q, nvi - 1)] = iraw[q];
#pragma unroll
    for (int q = 0; q < RD; ++q) dyl4[min(tid + 256 * q, NVD - 1)] = draw[q];
    if (tid == 0) dyl4[NVD] = lin_f4{0.f, 0.f, 0.f, 0.f};
    DRA_STAMP(TRR, 1);
    __syncthreads();
    DRA_STAMP(TRR, 2);
    // ---- MFMA: wave w owns tiles w, w+4, ...; tile t = (mt, nt), mt = t / NTL
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = zero16();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < TILES) {
        const int mt = tile / NTL, nt = tile - mt * NTL;
        const int k = k0 + mt * 32 + li;
        const int c = k / G::KK, kr = k - c * G::KK, kh = kr / G::KH, kw = kr - kh * G::KH;
        const float* abase = img + shift + (c - c_lo) * HW + kh * H + kw;
        // slice h = 1 of MFMA j reads position 2j + 1: the next column of the same output row, or -- when 2j + 1 starts a row --
        // the first column of the next one
        const float* ap_same = abase + h * S;
        const float* ap_wrap = abase + h * (S * H - (OH - 1) * S);
        const float* bp = dyl + (nt * 32 + li) * P + h;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bool last_odd = ODD && j == NJ - 1;
          const bool wrap = ((2 * j + 1) % OH) == 0;
          float a, b;
          if (last_odd) {            // positions (P - 1, pad): both slices read a finite A, slice 1 a zero B
            a = abase[pos_off(2 * j)];
            b = h ? dyl[NSRC] : bp[2 * j - h];
          } else {
            a = wrap ? ap_wrap[pos_off(2 * j)] : ap_same[pos_off(2 * j)];
            b = bp[2 * j];
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    DRA_STAMP(TRR, 3);
    // ---- slab stores (rows = k, 32 lanes along oc: 128-byte rows)
    float* dws = dw + (int64_t)bi * slab_stride;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < TILES) {
        const int mt = tile / NTL, nt = tile - mt * NTL;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int k = k0 + mt * 32 + mfma_row(rr, h);
          mega_st<COUT>(&dws[(int64_t)k * G::OC + nt * 32 + li], acc[t][rr]);
        }
      }
    }
    if (grp == 0 && tid < G::OC) {  // bias gradient of this sample: fixed-order row sum
      float sb = 0.f;
#pragma unroll 8
      for (int pos = 0; pos < P; ++pos) sb += dyl[tid * P + pos];
      mega_st<COUT>(&db[(int64_t)bi * slab_stride + tid], sb);
    }
    DRA_STAMP(TRR, 5);
    if constexpr (COUT) mega_publish(ms);
    DRA_STAMP_END(TRR);
  }
};

// ------------------------------------------------------------------------------------------------
// Linear weight gradient for a minibatch of at most 32 rows, NO LDS at all (fc4 of the DQN update: O = 512, I = 3136):
//   dW[o][i] = sum_b dy[b][o] * x[b][i],   db[o] = sum_b dy[b][o]
// The reduction index is the batch row: 16 MFMAs (32x32x2, slices h = rows 2j / 2j + 1) per 32 x 32 output tile, and both
// operands have their MFMA lane axis contiguous in memory -- lane li <-> output o for dy[b][o0 + li], lane li <-> input i for
// x[b][i0 + li] -- so they go straight from 128-byte coalesced global loads to registers.  The K-chunked implicit GEMM this
// replaces (IgemmRole<LinWgradSq<64, 64, 32>>: 400 workgroups staging 64 x 32 tiles through LDS with ~20 VALU instructions
// per MFMA) was the longest role of the fc backward launch (10.3 us for a contraction that WRITES 6.4 MB and reads 0.5 MB).
// Workgroup = (32 outputs, NI x 32 inputs): wave w owns input tiles w, w + 4, ... of the group; every wave holds the same 16
// dy registers.  With `partials` the workgroup leaves the sum of squares of what it stored (late-fold optimizer).
template <int NI>
struct LinWgradOne {
  static constexpr int LDS_FLOATS = 8;
  static constexpr int TPW = (NI + 3) / 4;
  const float* dy;   // [B][O]
  const float* x;    // [B][I]
  float* dw;         // [O][I]
  float* db;         // [O] or null
  double* partials;  // [blocks()] or null
  int B, O, I, tiles_o, groups_i;
  ChainHook hook;     // DRA_VAR_HEAD_CHAIN (run_<true>: dy comes from the head role of the SAME launch; x is requested before the wait)
  __host__ int blocks() const { return tiles_o * groups_i; }
  __device__ __forceinline__ void run(int bid, float* __restrict__ lds, int = 0) const { run_<false>(bid, lds); }
  template <bool CIN>
  __device__ __forceinline__ void run_(int bid, float* __restrict__ lds) const {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int gi = bid % groups_i, to = bid / groups_i;
    const int o0 = to * 32;
    DRA_STAMP(TR_FC_B, 0);
    // A: dy[2j + h][o0 + li]; rows >= B contribute zeros
    float areg[16];
    const int oc = min(o0 + li, O - 1);
    if constexpr (!CIN) {
#pragma unroll
      for (int j = 0; j < 16; ++j) areg[j] = dy[(int64_t)min(2 * j + h, B - 1) * O + oc];
    }
    float breg[TPW][16];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int i0 = (gi * NI + wave + 4 * t) * 32;
      const int ic = min(i0 + li, I - 1);
#pragma unroll
      for (int j = 0; j < 16; ++j) breg[t][j] = x[(int64_t)min(2 * j + h, B - 1) * I + ic];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (CIN) {
      mega_wait(hook.sync(0));
#pragma unroll
      for (int j = 0; j < 16; ++j) areg[j] = mega_ld<true>(dy + (int64_t)min(2 * j + h, B - 1) * O + oc);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (2 * j + h >= B) areg[j] = 0.f;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int it = wave + 4 * t;
      const int i0 = (gi * NI + it) * 32;
      if (it < NI && i0 < I) {
        f32x16 acc = zero16();
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[j], breg[t][j], acc, 0, 0, 0);
        const int i = i0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + mfma_row(r, h);
          if (o < O && i < I) {
            dw[(int64_t)o * I + i] = acc[r];
            sq += acc[r] * acc[r];
          }
        }
      }
    }
    DRA_STAMP(TR_FC_B, 3);
    // bias gradient: the sequential sum over the batch rows, by the first input group's wave 0 -- from the rows the wave already
    // holds (half h has rows 2j + h: the other half's value comes through a cross-half shuffle; a loop of 32 dependent loads
    // here was a 14 us critical path of its own)
    if (db && gi == 0 && wave == 0) {
      float sb = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float other = __shfl_xor(areg[j], 32, 64);          // (rows >= B are zero in both halves)
        const float even = h ? other : areg[j], odd = h ? areg[j] : other;
        sb += even;
        sb += odd;
      }
      if (h == 0 && o0 + li < O) {
        db[o0 + li] = sb;
        sq += sb * sb;
      }
    }
    if (partials) {   // (uniform) fixed-order workgroup sum: lanes by butterfly, waves (w0 + w1) + (w2 + w3)
      const double d = wave_sum((double)sq);
      double* dl = reinterpret_cast<double*>(lds);
      if (lane == 0) dl[wave] = d;
      __syncthreads();
      if (tid == 0) partials[bid] = (dl[0] + dl[1]) + (dl[2] + dl[3]);
    }
    DRA_STAMP(TR_FC_B, 5);
  }
};

// ------------------------------------------------------------------------------------------------
// Weight gradient of conv2 / conv3 at ROLLOUT batch sizes (PPO minibatches 256, 512, 1024): the same contraction as
// ConvWgradLin -- same staging (operands copied into LDS as they lie in memory), same MFMA loop -- but PERSISTENT over the batch:
// a workgroup = (group of MTG k tiles, share of the batch) keeps its output tiles in its accumulators and walks its samples SPI
// at a time, the next SPI samples' loads in flight (registers) while the current ones are contracted.  One slab per batch SHARE
// instead of one per sample: at batch 1024 conv2 wrote 134 MB of slabs that the fold read straight back (3.3-4.6x the
// algorithmic traffic, profiles/r04w_pmc_traffic.json).  Sizing rules measured in round 5 (profiles/r05k_conv_big_roles.jsonl):
// an iteration needs >= ~8 k cycles of MFMA per wave or the next samples' loads (2-3 us under load) are not back in time
// (conv3 with 75 MFMAs per sample and SPI 1: 0.28 of peak; conv2 with 164: 0.60); the grid is ONE workgroup per CU (288
// workgroups on 256 CUs left 32 CUs with two: conv3 0.37 -> 0.51 at 255); and a launch's LDS / VGPR footprint is the maximum
// over its roles: conv2's 46 KB next to the input-gradient role cost that role its occupancy (fused 205 us, the two launches one
// after the other 155 us at batch 1024), conv3's 20 KB does not (fused 112 us, separate 122 us).
// (A persistent input-gradient role -- weights staged once per workgroup, 16 x 16 x 4 MFMAs, no cross-wave fold -- was built and
// measured in the same round: 96.6 us against ConvDgradLin's 98.4 us for conv2 at batch 1024, 134 us against 75 us for conv3;
// one wave per SIMD cannot hide its own operand reads behind its own MFMAs (tools/ubench/mfma_coissue.hip), and the one-sample
// workgroups of ConvDgradLin overlap each other instead.  Not kept.)
// Bias gradient: the workgroups of k group 0 sum their samples' rows with all four waves (a quarter of the positions each) and
// fold them once at the end.  The DQN update (batch 32) keeps ConvWgradLin: with one sample per workgroup there is nothing to
// walk (and the accumulating form lost there, profiles/r04e_ab_env.jsonl).
template <class G, int MTG, int SHARES, int SPI>
struct ConvWgradPers {
  using L = ConvWgradLin<G, MTG>;
  static constexpr int S = G::S, OH = G::OH, P = G::P, H = G::H, HW = G::HW;
  static constexpr int NJ = L::NJ, NGRP = L::NGRP, NTL = L::NTL, TILES = L::TILES, TPW = L::TPW, IMGF = L::IMGF, NSRC = L::NSRC;
  static constexpr bool ODD = L::ODD;
  static constexpr int BUF = (IMGF + NSRC + 4 + 3) & ~3;       // one staged sample: input channels | gradient block | zero float4
  static constexpr int RED = 4 * G::OC;                       // bias partials of the four waves
  static constexpr int LDS_FLOATS = SPI * BUF + RED;
  const float* dy;   // [B][OC][OH][OH]
  const void* x;     // [B][C][H][H] f32
  float* dw;         // slab 0 of dWt [K][OC]
  float* db;         // slab 0 of db [OC]
  int64_t slab_stride;
  int B;
  __host__ __device__ static int spw(int batch) { return (batch + SHARES - 1) / SHARES; }        // samples per workgroup
  __host__ static int n_slabs(int batch) { return (batch + spw(batch) - 1) / spw(batch); }
  __host__ int blocks() const { return n_slabs(B) * NGRP; }
  static constexpr int pos_off(int p) { return (p / OH) * S * H + (p % OH) * S; }
  __device__ __forceinline__ void run(int bid, float* __restrict__ lds, int = 0) const {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int per = spw(B);
    const int grp = bid % NGRP, share = bid / NGRP;
    const int b0 = share * per, b1 = min(B, b0 + per);
    const int k0 = grp * MTG * 32;
    const int c_lo = k0 / G::KK;
    const int c_hi = min((k0 + MTG * 32 - 1) / G::KK, G::C - 1);
    const int nch = c_hi - c_lo + 1;
    constexpr int NVD = NSRC / 4, RD = (NVD + 255) / 256;
    constexpr int RI = (IMGF / 4 + 255) / 256;
    const int64_t xlast4 = ((int64_t)B * G::C * HW >> 2) - 1;
    lin_f4 draw[SPI][RD], iraw[SPI][RI];
    int shift_nxt[SPI], nvi_nxt[SPI], shift[SPI];
    auto fetch = [&](int bi0) {         // every load of up to SPI samples, into registers (samples past the share: clamped, unused)
#pragma unroll
      for (int u = 0; u < SPI; ++u) {
        const int bi = min(bi0 + u, b1 - 1);
        const lin_f4* dyb4 = reinterpret_cast<const lin_f4*>(dy + (int64_t)bi * NSRC);
#pragma unroll
        for (int q = 0; q < RD; ++q) draw[u][q] = dyb4[min(tid + 256 * q, NVD - 1)];
        const int64_t xstart = ((int64_t)bi * G::C + c_lo) * HW;
        shift_nxt[u] = (int)(xstart & 3);
        nvi_nxt[u] = (nch * HW + shift_nxt[u] + 3) >> 2;
        const lin_f4* x4 = reinterpret_cast<const lin_f4*>(x) + (xstart >> 2);
#pragma unroll
        for (int q = 0; q < RI; ++q) {
          const int64_t f = min((int64_t)(tid + 256 * q), (int64_t)nvi_nxt[u] - 1);
          iraw[u][q] = x4[min(f, xlast4 - (xstart >> 2))];
        }
      }
    };
    auto stage = [&]() {                // registers -> LDS images (unconditional stores, clamped indices: see ConvWgradLin)
#pragma unroll
      for (int u = 0; u < SPI; ++u) {
        lin_f4* img4 = reinterpret_cast<lin_f4*>(lds + u * BUF);
        lin_f4* dyl4 = reinterpret_cast<lin_f4*>(lds + u * BUF + IMGF);
#pragma unroll
        for (int q = 0; q < RI; ++q) img4[min(tid + 256 * q, nvi_nxt[u] - 1)] = iraw[u][q];
#pragma unroll
        for (int q = 0; q < RD; ++q) dyl4[min(tid + 256 * q, NVD - 1)] = draw[u][q];
        if (tid == 0) dyl4[NVD] = lin_f4{0.f, 0.f, 0.f, 0.f};
        shift[u] = shift_nxt[u];
      }
    };
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = zero16();
    float sb = 0.f;                     // bias partial of (oc = tid & (OC - 1), positions = wave, wave + 4, ...)
    fetch(b0);
    stage();
    __syncthreads();
    for (int bi = b0; bi < b1; bi += SPI) {
      if (bi + SPI < b1) fetch(bi + SPI);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < SPI; ++u) {
        if (bi + u < b1) {
          const float* img = lds + u * BUF;
          const float* dyl = img + IMGF;
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const int tile = wave + 4 * t;
            if (tile < TILES) {
              const int mt = tile / NTL, nt = tile - mt * NTL;
              const int k = k0 + mt * 32 + li;
              const int c = k / G::KK, kr = k - c * G::KK, kh = kr / G::KH, kw = kr - kh * G::KH;
              const float* abase = img + shift[u] + (c - c_lo) * HW + kh * H + kw;
              const float* ap_same = abase + h * S;
              const float* ap_wrap = abase + h * (S * H - (OH - 1) * S);
              const float* bp = dyl + (nt * 32 + li) * P + h;
#pragma unroll
              for (int j = 0; j < NJ; ++j) {
                const bool last_odd = ODD && j == NJ - 1;
                const bool wrap = ((2 * j + 1) % OH) == 0;
                float a, b;
                if (last_odd) {
                  a = abase[pos_off(2 * j)];
                  b = h ? dyl[NSRC] : bp[2 * j - h];
                } else {
                  a = wrap ? ap_wrap[pos_off(2 * j)] : ap_same[pos_off(2 * j)];
                  b = bp[2 * j];
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
              }
            }
          }
          if (grp == 0) {
            const int oc = tid & (G::OC - 1);
            for (int pos = wave; pos < P; pos += 4) sb += dyl[oc * P + pos];
          }
        }
      }
      __syncthreads();                  // every wave has read the images
      if (bi + SPI < b1) stage();
      __syncthreads();
    }
    // ---- one slab per batch share (rows = k, 32 lanes along oc: 128-byte rows)
    float* dws = dw + (int64_t)share * slab_stride;
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
    if (grp == 0) {
      static_assert(G::OC == 64, "one bias partial per (wave, output channel)");
      float* red = lds + SPI * BUF;
      red[wave * G::OC + (tid & (G::OC - 1))] = sb;
      __syncthreads();
      if (tid < G::OC) db[(int64_t)share * slab_stride + tid] = (red[tid] + red[G::OC + tid]) + (red[2 * G::OC + tid] + red[3 * G::OC + tid]);
    }
  }
};
