// Convolution input gradient at rollout batch sizes, SCATTER form (round 6; DRA_VAR_DGRAD_SCATTER).
//
// ConvDgradLin (oneshot_lin.h) is the gather form: one MFMA column per INPUT pixel, every tap of it multiplied whether the tap
// reaches an output position or not -- conv3 issues 96 x 9 (position, tap) columns per sample for 49 x 9 useful ones (1.96x),
// conv2 4 phases x 128 x 4 for 81 x 16 (1.58x): profiles/r06zy_conv_big.jsonl `issued_mfma_flop_frac` 1.49 / 1.30 for the whole
// backward launch.  Here the contraction runs over the OUTPUT positions, which carry no padding:
//   T[(c, kh, kw)][q] = sum_oc Wt[(c, kh, kw)][oc] * dY[oc][q]            q = output positions of NS samples, linearised
//   dX[c][oh * S + kh][ow * S + kw] += T[(c, kh, kw)][(oh, ow)]            (col2im, in LDS)
// and the only padding is the last 16-position tile of a workgroup's NS samples (conv3, NS = 2: 112 / 98 = 1.14; conv2, NS = 1:
// 96 / 81 = 1.19).
//
// v_mfma_f32_16x16x4_f32 (same FLOP per cycle as 32x32x2): M = 16 input channels, N = 16 output positions, K = 4 output channels
// per instruction.  A wave owns 16 input channels (x a stride-phase set of taps when the layer has fewer than 64 input channels)
// and keeps ALL of their weights in registers (taps x 16 K-steps: 144 for conv3, 128 for conv2); a tile's gradient operand is 16
// registers loaded straight from global memory (each element is used by one instruction per tap -- nothing to stage) one tile
// ahead.  Each tap's 16 x 16 result is added into the workgroup's dX image in LDS by a plain ds_read / v_add / ds_write sequence:
// within one tap the 64 lanes' addresses are distinct, no two waves ever touch the same address -- waves differ in channel block,
// or (conv2) in the parity of the input row their taps reach -- and one wave's LDS instructions execute in program order, so a
// later tap's read sees an earlier tap's write and every dX element is summed in a fixed order: run-to-run deterministic.
// (ds_add_f32 would do the same in one instruction; it ran at ~230 cycles per wave instruction -- 97 us per workgroup against
// 29, profiles/r06zzg_ab_dgrad_scatter.jsonl.)  The image has the layout of dX itself; the epilogue applies the activation
// derivative and copies it out in float4s.
//
// Measured (profiles/r06zzl_*, r06zzn_conv_big.jsonl): input gradient alone 0.50 of the fp32-MFMA peak at batch 1024 (conv3; gather form
// 0.33), whole backward launch 0.49-0.52 by rocprofv3 (0.43-0.44); a workgroup's life at one per CU = 3.7 us until its 147 KB of
// weights have arrived (all CUs fetch at once: ~10 TB/s chip-wide), 2.9 us per tile (1.9 us of MFMA issue), 2-4 us epilogue.
//
// Arithmetic: per dX element the taps' partial sums (each an fp32 MFMA chain over the 64 output channels) are added in tap
// order -- a different association than the gather form's one chain over (tap, oc), same products; the contraction tests compare
// both against float64 at 1e-5 of the tensor's scale.  network_bodies.py:10-33 (backward of NatureConvBody's conv2 / conv3).
#pragma once
#include "oneshot_lin.h"
#ifndef DRA_SCAT_PIPE
#define DRA_SCAT_PIPE 1     // explicit MFMA / LDS interleave of the tile body (sched_barrier between 8-MFMA chunks); 0 = the compiler's own order
#endif

typedef float scat_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float scat_lds_float;

// LOOP: the workgroup takes several groups of NS samples (`cap` workgroups in all) with its weights fetched once -- for a launch the
// role has to itself; the registers the loop keeps alive across the epilogue (~30) would cost the shared launches their second wave.
template <class G, int NS, bool LOOP = false>
struct ConvDgradScat {
  static constexpr int S = G::S, OH = G::OH, P = G::P, H = G::H, HW = G::HW, OC = G::OC, C = G::C, KH = G::KH;
  static constexpr int NCB = C / 16;           // 16-channel blocks
  static constexpr int NPS = 4 / NCB;          // tap sets per channel block: set ps holds the taps kh = ps + NPS * i (input rows of parity ps)
  static constexpr int KHN = KH / NPS, NTAP = KHN * KH;
  static constexpr int KST = OC / 4;           // MFMA steps per tap; in step j = 4 v + e lane group kq multiplies oc = 16 v + 4 kq + e:
                                               // a weight float4 load (fixed v) reads 64 contiguous bytes per channel
  static constexpr int TG = (NTAP % 3 == 0) ? 3 : ((NTAP % 4 == 0) ? 4 : 1);   // taps whose accumulation chains interleave
#ifndef DRA_SCAT_HWPAD
#define DRA_SCAT_HWPAD 4    // even HW: channel stride HW + 4 (float4 rows for the epilogue; HW + 1 spreads the read-add-write's banks
                            // better -- 2-way instead of 4-way conflicts -- but forces scalar LDS reads + an index division there)
#endif
  static constexpr int HWP = (HW & 1) ? HW : HW + DRA_SCAT_HWPAD;              // channel stride of the LDS image
  static constexpr int IMG = NS * C * HWP;
  static constexpr int ROWF = KH * OC, ROW4 = ROWF / 4;        // one contiguous (channel, kh) weight row: floats, float4s
  static constexpr int RPT = 16 * ROW4 / 64;                   // float4s per lane of one kh row of a wave's 16 channels
  static constexpr int WREG = 16 * (ROWF + 4);                 // a wave's staging region (one kh row)
  static constexpr int DUMP = 64 + 12 * HWP + (KH - 1) * (H + 1) + 4;   // lanes without a position: base IMG + lane, + channel / tap offsets
  static constexpr int LDS_FLOATS = ((IMG + DUMP > 4 * WREG ? IMG + DUMP : 4 * WREG) + 3) & ~3;
  static_assert((16 * ROW4) % 64 == 0, "whole lanes");
  static constexpr int NQ = NS * P, TILES = (NQ + 15) / 16;
  static_assert((TG * KST) % (16 * TG) == 0 && KST == 16 && C % 16 == 0 && NCB * NPS == 4, "four waves = channel blocks x tap sets");
  static_assert(NPS == 1 || (S % NPS == 0 && KH % NPS == 0), "tap sets reach disjoint input rows");
  static_assert(OC == 64 && KST % 4 == 0 && (C * HW) % 4 == 0 && (HWP == HW || HW % 4 == 0), "float4 weight runs / float4 epilogue");
  const float* dy;    // [B][OC][OH][OH] pre-activation gradient of this layer's output
  const float* wt;    // [(c,kh,kw)][OC]
  const float* xact;  // [B][C][H][H] this layer's input (post-activation) or null
  float* dx;          // [B][C][H][H]
  int B, act;
  int cap = 0;        // > 0: at most `cap` workgroups, each looping over groups of NS samples (weights fetched once per workgroup);
                      // set when the role has a launch to itself (two workgroups per CU fit: 2 x the CU count)
  __host__ int groups() const { return (B + NS - 1) / NS; }
  __host__ int blocks() const { return LOOP && cap > 0 && groups() > cap ? cap : groups(); }
  __device__ __forceinline__ void run(int bid, float* __restrict__ lds, int = 0) const {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int cb = wave % NCB, ps = wave / NCB;
    const int n_groups = (B + NS - 1) / NS, n_wg = LOOP && cap > 0 && n_groups > cap ? cap : n_groups;
    int b0 = bid * NS, ns = min(NS, B - b0), nq = ns * P;
    [[maybe_unused]] constexpr int TRR = (G::C == 32) ? TR_CONV2_B : TR_CONV3_B;
    DRA_STAMP(TRR, 0);
    // ---- weights: A row m <-> channel 16 cb + (m >> 2) + 4 (m & 3), so that output register r of lane group kq is channel
    // 16 cb + kq + 4 r: the four lane groups' image addresses differ by one (odd) channel stride.
    // A wave's weights are 16 channels x KHN rows of KH x OC contiguous floats.  Read per lane (16 bytes per request: adjacent
    // lanes are different channels) they arrived in 4.3 us per workgroup, 7.6 with two per CU (profiles/r06zzj); here each wave
    // fetches its rows in whole 1 KB runs (everything requested up front) and redistributes them through its own LDS region,
    // one kh row at a time (channel stride ROWF + 4: 2-way conflicts on the b128 reads); no workgroup barrier involved.
    scat_f4 wraw[KHN][RPT];
#pragma unroll
    for (int khi = 0; khi < KHN; ++khi)
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int f = lane + 64 * i, ci = f / ROW4, col = f - ci * ROW4;
        wraw[khi][i] = *reinterpret_cast<const scat_f4*>(wt + ((int64_t)(16 * cb + ci) * G::KK + (ps + NPS * khi) * KH) * OC + 4 * col);
      }
    const float* dyb = dy + (int64_t)b0 * OC * P;
    int boff = 0, aoff = 0;
    bool ok = false;
    auto place = [&](int tile) {     // this lane's output position of `tile`: gradient offset, image offset
      int q = tile * 16 + n;
      ok = q < nq;
      q = min(q, nq - 1);
      const int s = q / P, p = q - s * P, oh = p / OH, ow = p - oh * OH;
      boff = s * OC * P + 4 * kq * P + p;
      aoff = (s * C + 16 * cb + kq) * HWP + (oh * S + ps) * H + ow * S;
    };
    float bnext[KST];
    place(0);
#pragma unroll
    for (int j = 0; j < KST; ++j) bnext[j] = dyb[boff + (16 * (j >> 2) + (j & 3)) * P];
    float a[NTAP][KST];
    {
      float* wreg = lds + wave * WREG;
#pragma unroll
      for (int khi = 0; khi < KHN; ++khi) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int f = lane + 64 * i, ci = f / ROW4, col = f - ci * ROW4;
          *reinterpret_cast<scat_f4*>(wreg + ci * (ROWF + 4) + 4 * col) = wraw[khi][i];
        }
        const float* wl = wreg + ((n >> 2) + 4 * (n & 3)) * (ROWF + 4) + 4 * kq;
#pragma unroll
        for (int kw = 0; kw < KH; ++kw)
#pragma unroll
          for (int v = 0; v < KST / 4; ++v) {
            const scat_f4 w4 = *reinterpret_cast<const scat_f4*>(wl + kw * OC + 16 * v);
            const int t = khi * KH + kw;
            a[t][4 * v] = w4.x; a[t][4 * v + 1] = w4.y; a[t][4 * v + 2] = w4.z; a[t][4 * v + 3] = w4.w;
          }
      }
    }
    int grp = bid;
    do {
    if (LOOP && grp != bid) {      // (a later group of this workgroup: the first one's operands were requested in front of the weights)
      b0 = grp * NS; ns = min(NS, B - b0); nq = ns * P;
      dyb = dy + (int64_t)b0 * OC * P;
      place(0);
#pragma unroll
      for (int j = 0; j < KST; ++j) bnext[j] = dyb[boff + (16 * (j >> 2) + (j & 3)) * P];
    }
    __syncthreads();       // (the image overlaps every wave's staging region; later groups: the epilogue has read the image)
    // ---- the image starts at zero
    {
      scat_f4* l4 = reinterpret_cast<scat_f4*>(lds);
      for (int i = tid; i < LDS_FLOATS / 4; i += 256) l4[i] = scat_f4{0.f, 0.f, 0.f, 0.f};
    }
#ifdef DRA_TRACE
    dra_drain();          // (trace build only: stamp 1 = weights and the first tile's gradient have arrived)
    DRA_STAMP(TRR, 1);
#endif
    __syncthreads();
    DRA_STAMP(TRR, 2);
    const int tiles = (nq + 15) >> 4;
    // The read-add-write of a tap group runs one group BEHIND the MFMAs (its 3-4 LDS round trips of ~100 cycles each sit between
    // the next group's MFMAs instead of in front of them); lanes past the last position aim at a dump area behind the image
    // (their own address per lane: no exec masking, the tile body is one basic block for the scheduler).
    constexpr int NG = NTAP / TG;
    scat_f4 pacc[TG];
#pragma unroll
    for (int u = 0; u < TG; ++u) pacc[u] = scat_f4{0.f, 0.f, 0.f, 0.f};
    int pao = IMG + lane;
    auto rmw = [&](int t, int base, const scat_f4& v) {
      const int off = NPS * (t / KH) * H + (t % KH);
      // (volatile = in program order; address space kept: ds_read / ds_write, not flat)
      volatile scat_lds_float* pl = (volatile scat_lds_float*)(lds + base + off);
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = pl[4 * r * HWP];
#pragma unroll
      for (int r = 0; r < 4; ++r) pl[4 * r * HWP] = o[r] + v[r];
    };
    for (int tile = 0; tile < tiles; ++tile) {
      float b[KST];
      const bool okc = ok;
      const int ao = okc ? aoff : IMG + lane;
#pragma unroll
      for (int j = 0; j < KST; ++j) b[j] = okc ? bnext[j] : 0.f;
      if (tile + 1 < tiles) {
        place(tile + 1);
#pragma unroll
        for (int j = 0; j < KST; ++j) bnext[j] = dyb[boff + (16 * (j >> 2) + (j & 3)) * P];
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        scat_f4 acc[TG];
#pragma unroll
        for (int u = 0; u < TG; ++u) acc[u] = scat_f4{0.f, 0.f, 0.f, 0.f};
        // 2 TG chunks of 8 MFMAs (the group's TG accumulation chains interleaved); behind chunk 2u the four LDS reads of tap u of
        // the group BEFORE this one (the previous tile's last group when g == 0), behind chunk 2u + 1 its adds and writes.
        // sched_barrier pins that order: each LDS round trip has 256 cycles of MFMA issue to hide under.
        const int lbase = g == 0 ? pao : ao;
        float o[4];
#pragma unroll
        for (int c = 0; c < 2 * TG; ++c) {
#pragma unroll
          for (int i = 8 * c; i < 8 * c + 8; ++i) {
            const int jj = i / TG, u = i - jj * TG;
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g * TG + u][jj], b[jj], acc[u], 0, 0, 0);
          }
          const int t = ((g + NG - 1) % NG) * TG + (c >> 1), off = NPS * (t / KH) * H + (t % KH);
          // (volatile = in program order; address space kept: ds_read / ds_write, not flat)
          volatile scat_lds_float* pl = (volatile scat_lds_float*)(lds + lbase + off);
#if DRA_SCAT_PIPE
          __builtin_amdgcn_sched_barrier(0);
#endif
          if ((c & 1) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = pl[4 * r * HWP];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) pl[4 * r * HWP] = o[r] + pacc[c >> 1][r];
          }
#if DRA_SCAT_PIPE
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) pacc[u] = acc[u];
      }
      pao = ao;
      if (tile == 0) DRA_STAMP(TRR, 3);
    }
#pragma unroll
    for (int u = 0; u < TG; ++u) rmw((NG - 1) * TG + u, pao, pacc[u]);
    DRA_STAMP(TRR, 4);
    __syncthreads();
    DRA_STAMP(TRR, 5);
    // ---- epilogue: image -> dX (the image IS [sample][channel][pixel]), activation derivative from this layer's input
    const int total4 = ns * C * HW / 4;
    const scat_f4* xa4 = reinterpret_cast<const scat_f4*>(xact ? xact + (int64_t)b0 * C * HW : nullptr);
    scat_f4* dx4 = reinterpret_cast<scat_f4*>(dx + (int64_t)b0 * C * HW);
    constexpr int EP = (NS * C * HW / 4 + 255) / 256, EB = 6;      // float4s per thread, in batches of EB with their loads in flight together
#pragma unroll
    for (int i0 = 0; i0 < EP; i0 += EB) {
      scat_f4 y[EB];
#pragma unroll
      for (int i = 0; i < EB; ++i)
        if (i0 + i < EP) y[i] = xact ? xa4[min(tid + 256 * (i0 + i), total4 - 1)] : scat_f4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int i = 0; i < EB; ++i) {
        if (i0 + i >= EP) continue;
        const int e4 = tid + 256 * (i0 + i);
        if (e4 >= total4) continue;
        scat_f4 v;
        if constexpr (HWP == HW) v = reinterpret_cast<const scat_f4*>(lds)[e4];
        else {
          const int e = 4 * e4, ch = e / HW, pix = e - ch * HW;
          const float* src = lds + ch * HWP + pix;
          if constexpr (HWP % 4 == 0) v = *reinterpret_cast<const scat_f4*>(src);
          else v = scat_f4{src[0], src[1], src[2], src[3]};
        }
        if (xact) v = scat_f4{v.x * act_grad(y[i].x, act), v.y * act_grad(y[i].y, act), v.z * act_grad(y[i].z, act), v.w * act_grad(y[i].w, act)};
        dx4[e4] = v;
      }
    }
    } while (LOOP && (grp += n_wg) < n_groups);      // (groups of this workgroup)
    DRA_STAMP_END(TRR);
  }
};
