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
#include "actor_env.h"
#include "rollout_roles.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// conv_fwd_v2_body: image rows requested before the weight operands (see there).  DRA_EXP_IMAGE_FIRST=0 builds the
// round 2-5 order for A/B runs (tools/README.md: `make exp EXPFLAGS=...`).
#ifndef DRA_EXP_IMAGE_FIRST
#define DRA_EXP_IMAGE_FIRST 1
#endif
constexpr bool kImageFirst = DRA_EXP_IMAGE_FIRST != 0;

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
  // the same split over NW waves (NW = 4 gives the members above).  NW = 8 is the batch-1 (actor) shape: the launch
  // has 4-13 workgroups, so a workgroup's latency IS the kernel's -- half the operand loads and half the MFMA chain
  // per wave (conv3: 72 -> 36 dependent 64-cycle MFMAs, 1.9 -> 0.95 us)
  template <int NW>
  struct Split {
    static constexpr int CPW = CP >= NW ? CP / NW : 1;
    static constexpr int TSPLIT = CP >= NW ? 1 : NW / CP;
    static constexpr int TW = KK / TSPLIT;
    static constexpr int NJ = CPW * TW;
    static_assert(CP >= NW ? CP % NW == 0 : (NW % CP == 0 && KK % TSPLIT == 0), "K split");
    static_assert(TW % KH == 0, "a wave's tap range starts on a kernel-row boundary");
  };
};

using VG1_ = V2Geom<4, 84, 32, 8, 4>;      // conv1 (also VG1 below)

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
  // optional, with ring_slot (same indirection): frames of the same episode older than the newest one, capped at C-1;
  // channel c reads slot newest - min(C-1-c, age).  null = C-1 (the last C ring frames)
  const int32_t* stack_age;
  // optional indirection for `ring_slot`: the slot lives in entry (*slot_seq mod slot_entries) of an array of
  // parameter blocks slot_stride bytes apart (the learner's K-steps-ahead actor parameter ring)
  const unsigned* slot_seq;
  int slot_entries;
  int64_t slot_stride;
  // optional: the NEWEST channel of the ring stack comes from this frame instead of ring slot *ring_slot (an
  // observation that has not been committed to the replay ring yet)
  const uint8_t* newest_frame;
  // optional (uint8 BATCHED input, the update's conv1 straight from the replay ring): sample bi of net z is the C
  // consecutive ring frames starting at slot sample_idx[bi] + idx_bias[z] of the slot-major frame array x[z] (a sample's
  // frames never wrap: replay.py:105-110), instead of image bi * C of a plain NCHW batch -- no gathered copy of the
  // minibatch is written or re-read
  const int64_t* sample_idx = nullptr;
  int64_t idx_bias[DRA_MAX_Z] = {};
  // optional (round 4): extra z-slices of the launch whose workgroups only PREFETCH the next launch's weights into the L2 of
  // the XCD that will read them -- fc4's forward (LinFwdSlabsOne<3136, 14, 2>, two nets: 224 workgroups streaming 57 KB of
  // weights each at the ~25 GB/s a CU gets from the fabric, 3.2 us of its 7.5).  Prefetch workgroup p runs on XCD p mod 8
  // (the z-slices in front of it hold a multiple of 8 workgroups), exactly where fc4's workgroup p will run, and issues the
  // same 14 float4 loads per lane; nothing is stored.  pf_nz = nets of that launch (0 = off).
  const float* pf_w[DRA_MAX_Z] = {};
  int pf_nz = 0, pf_first = 0;
  // optional: sample_idx may live in pinned HOST memory (one PCIe read per workgroup, ~1 us inside the operand phase); the
  // first workgroup of every sample leaves a copy here (device memory) for the later kernels of the same update
  int64_t* sample_idx_copy = nullptr;
  // optional (DRA_VAR_IDX_PREFETCH): a device copy of sample_idx left by the PREVIOUS update's head kernel, every element
  // tagged in its bits 63:40 with (number of the update it belongs to + 1) mod 2^24; *sample_seq = updates completed so far.
  // An element whose tag matches is used as is (a device load instead of the PCIe read in front of the frame loads);
  // otherwise (the host had not written the indices yet when the prefetch ran) sample_idx is read as before
  const int64_t* sample_idx_tagged = nullptr;
  const unsigned long long* sample_seq = nullptr;
  // != 0: XCD-aware block order (common.h xcd_order): every workgroup of one (net, sample) runs on one XCD
  int xcd_order = 0;
  // DRA_VAR_DEFER_FC4 (common.h DraFc4Rider; dra_conv_attach_rider): the FIRST rider_z z-slices of the launch are RIDERS -- workgroup k
  // of them steps rider block rider_first + k (< rider_end) of the deferred fc4 segment; nz_real = the nets of the launch itself.
  // rider_done_*: the launch's first workgroup lowers `pending` / marks the completed actor copy valid (the launch AFTER the riders')
  DraFc4Rider rider = {};
  int rider_z = 0, rider_first = 0, rider_end = 0, nz_real = 0;
  int* rider_done_pending = nullptr;
  int* rider_done_valid = nullptr;
};

// the attachment the next batched forward launch consumes (dra_conv_attach_rider)
static thread_local struct { DraFc4Rider rider; int first, count; int* done_pending; int* done_valid; bool armed; } g_rider_next;
void dra_conv_attach_rider(const DraFc4Rider* rider, int first, int count, int* done_pending, int* done_valid) {
  g_rider_next.armed = (rider && count > 0) || done_pending || done_valid;
  if (rider) g_rider_next.rider = *rider;
  g_rider_next.first = first;
  g_rider_next.count = rider ? count : 0;
  g_rider_next.done_pending = done_pending;
  g_rider_next.done_valid = done_valid;
}

__device__ __forceinline__ float v2_act(float v, int act) {
  if (act == DRA_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DRA_ACT_TANH) return tanhf(v);
  return v;
}

template <class G>
__device__ __forceinline__ int lds_col(int iw) { return (iw % G::S) * G::WPH + iw / G::S; }

// PT = consecutive 32-position tiles of ONE sample per workgroup.  PT = 1 is the latency shape (batch 32 and the
// batch-1 actor: most workgroups, shortest dependent chain).  PT > 1 is the throughput shape for large batches
// (A2C / PPO minibatches, the batch-1024 roofline measurement): the register-resident weights and the staged
// rows (adjacent tiles share most of them) are reused by PT independent MFMA accumulation chains per wave --
// rocprofv3 SQ counters at batch 1024, PT = 1: waves spend 71 % of their cycles stalled on MFMA issue behind the
// other resident waves while the pipe itself is only ~45 % busy, because every workgroup pays the full
// load / staging / reduction phases for 32 MFMAs per wave (profiles/r01e_pmc_conv_fwd_b1024_before.json).
template <class G, int PT>
struct V2Tile {
  static constexpr int TPG = (G::TPS + PT - 1) / PT;                       // tile groups per sample
  static constexpr int OROWS = (32 * PT - 1 + G::OH - 1) / G::OH + 1;      // output rows a group can touch
  static constexpr int NR_RAW = (OROWS - 1) * G::S + G::KH;
  static constexpr int NR = NR_RAW < G::H ? NR_RAW : G::H;
  static constexpr int CS = NR * G::RW;                                    // LDS channel stride
};

// q[a] = bh[a] + <h4, Wh[a]> for the VanillaNet head at batch 1, one wave per action (same per-lane order and
// butterfly as actor_head_env_ring_kernel: bit-identical action values), into s_q; caller synchronises.
template <int NW>
__device__ __forceinline__ void fused_head_q(const ActorFuse& f, int wave, int lane, float* __restrict__ s_q) {
  if (f.head_kind != 0) {   // distributional head: its outputs already exist, one wave reduces one action's
    for (int a = wave; a < f.n_actions; a += NW) {
      const float q = dist_action_value(f.pre + a * f.n_atoms, f.n_atoms, f.head_kind, f.atoms, lane);
      if (lane == 0) s_q[a] = q;
    }
    return;
  }
  for (int a = wave; a < f.n_actions; a += NW) {
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) part += f.h4[lane + 64 * i] * f.wh[a * 512 + lane + 64 * i];
    part = wave_sum(part);
    if (lane == 0) s_q[a] = part + f.bh[a];
  }
}

// The environment workgroup of a fused actor launch (ActorFuse, actor_env.h).
template <int NW>
__device__ __forceinline__ void fused_env_workgroup(const ActorFuse& f, float* __restrict__ s_q) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const dra_dqn_step_params* __restrict__ prm = aring_entry(f.aring, *f.seq);
  if (f.mode == 1) {   // env step 0: feed the pending observation into ring slot[0]
    if (prm->counter[0] < 0) return;
    const int64_t slot = prm->slot[0];
    const uint64_t* src = reinterpret_cast<const uint64_t*>(f.pend_frame);
    uint64_t* dst = reinterpret_cast<uint64_t*>(f.frames + slot * 7056);
    for (int w = tid; w < 882; w += 64 * NW) dst[w] = src[w];
    if (tid == 0) { f.rewards[slot] = *f.pend_reward; f.masks[slot] = *f.pend_mask; }
    return;
  }
  const int e = f.e;   // head of env step e-1, then the environment step that produces observation e
  fused_head_q<NW>(f, wave, lane, s_q);
  __syncthreads();
  if (tid < f.n_actions && f.q_out) f.q_out[tid] = s_q[tid];
  const int64_t act = eps_greedy_action(s_q, f.n_actions, prm, e - 1);
  if (tid == 0 && prm->store_action[e - 1]) *reinterpret_cast<int64_t*>(f.actions + prm->slot[e - 1] * 8) = act;
  const int64_t counter = prm->counter[e];
  if (counter < 0) return;
  const int64_t slot = prm->slot[e];
  uint64_t* dst = reinterpret_cast<uint64_t*>(f.frames + slot * 7056);
  for (int w = tid; w < 882; w += 64 * NW) dst[w] = synth_frame_word(f.seed, counter, act, w);
  if (tid == 0) {
    f.rewards[slot] = synth_reward(f.seed, prm->rcounter[e]);
    f.masks[slot] = synth_mask(f.seed, prm->rcounter[e], f.done_period);
  }
}

// ---- the actor's env step as ONE launch (DRA_VAR_ACTOR_MEGA): cross-workgroup hand-over inside a kernel ------------------
// A producer writes its outputs with agent-scope stores (written through: the XCDs' L2s are not coherent with each other),
// completes them (s_waitcnt), and ONE thread counts the workgroup on the layer's arrival counter; a consumer requests
// everything that does not depend on the producer (its weights) first, then one thread polls the counter (bounded: 50 ms,
// then the pinned timeout flag is set and the wait gives up -- results of that launch are invalid and the host reports
// DRA_ETIMEDOUT), and the activations are read with agent-scope loads.  Few arrivals per counter (13 / 12 / 8): the
// ~30 ns an agent-scope atomic costs on this part does not add up (the 796-ticket grid barrier of round 2 did).
// (MegaSync, mega_publish / mega_wait / mega_ld / mega_st live in common.h: the update's chained launches use them too)

// (bx, by, bz) = the workgroup's position in the conv grid (blockIdx of a plain launch; a role offset inside the actor's
// one-launch env step), env_wg = this workgroup is the fused launch's environment workgroup; COH = outputs are handed to
// other workgroups of the SAME launch (agent-scope stores + mega_publish)
// CIN = the input planes come from other workgroups of the SAME launch (the update's forward chain): weights first, then the wait
// for the producers of this workgroup's sample, then agent-scope loads
template <class G, bool U8, int PT, int NW, bool FUSE, bool COH = false, bool CIN = false>
__device__ __forceinline__ void conv_fwd_v2_body(const ConvV2Args& a, const ActorFuse& f, const int bx, const int by, const int bz,
                                                 const bool env_wg, const MegaSync ms = MegaSync()) {
  using T = V2Tile<G, PT>;
  using KS = typename G::template Split<NW>;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  static_assert(!FUSE || (U8 && PT == 1 && G::C == 4), "the fused actor launch is conv1 on uint8 ring frames");
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [C][NR][RW] image, then reused for the reduction
  __shared__ float s_q[FUSE ? 64 : 1];
  if constexpr (FUSE) {
    if (env_wg) {   // the launch's extra workgroup: environment side
      fused_env_workgroup<NW>(f, s_q);
      return;
    }
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int z = bz;
  const int bi = bx / T::TPG, grp = bx - bi * T::TPG;
  const int oc0 = by * 32;
  const int p0 = grp * PT * 32;
  const int np = min(32 * PT, G::P - p0);
  const int oh0 = p0 / G::OH, oh1 = (p0 + np - 1) / G::OH;
  const int ir0 = oh0 * G::S;
  const int nrows = (oh1 - oh0) * G::S + G::KH;  // <= T::NR
  [[maybe_unused]] const int TRR = (a.batch == 1 ? TR_A_CONV1 : TR_CONV1_F) + (G::C == 4 ? 0 : (G::C == 32 ? 1 : 2));
  DRA_STAMP(TRR, 0);

  // ---- fused actor launch: the previous env step's head is requested FIRST (its loads complete first, so the action
  // is known while the weight / frame loads below are still in flight)
  [[maybe_unused]] const dra_dqn_step_params* prm = nullptr;
  if constexpr (FUSE) {
    if (f.mode == 2) {
      prm = aring_entry(f.aring, *f.seq);
      fused_head_q<NW>(f, wave, lane, s_q);
    }
  }
  // ---- issue every global load of this workgroup: weight operands first, then the image rows
  const float* __restrict__ wt = a.wt[z];
  const int cp0 = (G::CP >= NW) ? wave * KS::CPW : (wave % G::CP);
  const int t0 = (G::CP >= NW) ? 0 : (wave / G::CP) * KS::TW;
  float areg[KS::NJ];
  // bias of the 4 output rows this wave finalises: requested with the operands (a load in the epilogue exposes a
  // second memory latency per workgroup: 0.5-0.7 us of the 1.2-2.0 us epilogue in the phase traces, profiles/r02a_*)
  constexpr int RPW = 16 / NW;                                 // accumulator registers a wave finalises
  float bias_r[RPW];
  // Order of the requests (round 6): a wave's loads return in the order they were issued, and only the IMAGE gates the
  // barrier in front of the MFMA loop -- MFMA j needs weight register j alone.  Image rows first, weights behind them:
  // the MFMA loop starts when the image is staged and walks down vmcnt while the rest of its 16 KB of weights per wave
  // is still arriving (weights first: the barrier waited for all of them).  Same values, same order of operations.
  // (CIN: the first half of the weights is requested in front of the wait, the second half once the staged rows have left their
  // registers -- it arrives behind the first half's MFMAs; all of them at once with the 48 row registers of conv2 exceed the 128
  // registers four chain workgroups per CU leave a wave)
  constexpr int NJ0 = CIN ? KS::NJ / 2 : KS::NJ;
  auto request_weights_range = [&](auto lo, auto hi) {
    const float* wbase = wt + ((int64_t)(2 * cp0 + h) * G::KK + t0) * G::OC + oc0 + li;
#pragma unroll
    for (int j = decltype(lo)::value; j < decltype(hi)::value; ++j) {
      const int cpl = j / KS::TW, t = j - cpl * KS::TW;
      areg[j] = wbase[(2 * cpl * G::KK + t) * G::OC];
    }
  };
  auto request_weights = [&]() {
    request_weights_range(std::integral_constant<int, 0>{}, std::integral_constant<int, NJ0>{});
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int r = wave * RPW + q;
      bias_r[q] = a.bias[z][oc0 + (r & 3) + 8 * (r >> 2) + 4 * h];
    }
  };
  static_assert(!CIN || (!U8 && G::H <= 32), "chained input: the fp32 row-shaped staging");
  if (!kImageFirst || CIN) request_weights();
  if constexpr (CIN) {
    __builtin_amdgcn_sched_barrier(0);   // the weight / bias loads above are in flight while this workgroup waits
    mega_wait(ms);
  }
  // Staging maps lanes to (row, column) so that no per-element division is needed: LR lanes walk one image
  // row (the surplus lanes of a row idle), 64 / LR rows per pass; row offsets are compile-time immediates and
  // the de-interleaved LDS column is a per-lane constant.  (The flat "element e of the block" mapping this
  // replaces cost ~40 integer multiplies and 70 shifts / adds per lane -- on CDNA the fp32 MFMA and the vector
  // ALU issue from the same SIMD port, so every VALU cycle in a workgroup is an MFMA cycle lost: SQ counters in
  // profiles/r01e_pmc_conv_fwd_b1024_*.json.)
  if (U8) {
    // uint8 frames (conv1: S = 4, rows of 84 bytes = 21 u32 words): lane (rsub, wd) loads word wd of rows
    // 2q + rsub; pixel b of word wd is column 4 wd + b = stride phase b, phase index wd.  The exact
    // normalisation f32(f64(v) * coef) comes from a 256-entry LDS table built while the loads are in flight.
    static_assert(!U8 || (G::S == 4 && G::H % 4 == 0 && G::H / 4 <= 32), "u8 staging: stride-4 layer, <= 32 words per row");
    constexpr int WPR = G::H / 4;                               // u32 words per row
    // waves = (channel group, row part): RPARTS waves share a channel and interleave its row passes
    constexpr int RPARTS = (NW > G::C) ? NW / G::C : 1;
    constexpr int CW = NW / RPARTS;                             // waves along the channel axis
    constexpr int LPT = ((T::NR + 1) / 2 + RPARTS - 1) / RPARTS; // row passes per lane per channel
    constexpr int CPT = (G::C + CW - 1) / CW;                   // channels per wave
    __shared__ float s_lut[256];
    unsigned raw[CPT * LPT];
    const uint8_t* xb = reinterpret_cast<const uint8_t*>(a.x[z]);
    const int wc = wave % CW, rpart = wave / CW;
    const int rsub = (lane >> 5) + 2 * rpart, wd = lane & 31;   // first row of this lane; passes advance by 2*RPARTS
    const int wdc = min(wd, WPR - 1);
    int64_t newest = 0;
    if (a.ring_slot) {
      const char* sp = reinterpret_cast<const char*>(a.ring_slot);
      if (a.slot_seq) sp += (int64_t)(*a.slot_seq % (unsigned)a.slot_entries) * a.slot_stride;
      newest = *reinterpret_cast<const int64_t*>(sp);
    }
    int age = G::C - 1;
    if (a.ring_slot && a.stack_age) {
      const char* ap = reinterpret_cast<const char*>(a.stack_age);
      if (a.slot_seq) ap += (int64_t)(*a.slot_seq % (unsigned)a.slot_entries) * a.slot_stride;
      age = *reinterpret_cast<const int32_t*>(ap);
    }
    bool generated = false;      // fused mode 2: the newest channel is produced by the environment step below
    if constexpr (FUSE) generated = f.mode == 2;
    int64_t si = 0;              // ring-direct minibatch: the sampled slot of this workgroup's transition
    if (a.sample_idx) {
      if (a.sample_idx_tagged) {
        const int64_t tv = a.sample_idx_tagged[bi];
        const unsigned long long want = (*a.sample_seq + 1ull) & 0xffffffull;
        if ((unsigned long long)tv >> 40 == want) si = tv & ((1ll << 40) - 1);
        else si = a.sample_idx[bi];
      } else {
        si = a.sample_idx[bi];
      }
    }
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wc + CW * ci;
      int64_t img = (int64_t)bi * G::C + min(c, G::C - 1);       // image index in a plain NCHW batch
      if (a.sample_idx) {                                        // ... or a run of ring slots
        img = si + a.idx_bias[z] + min(c, G::C - 1);
        if (a.sample_idx_copy && ci == 0 && tid == 0 && grp == 0 && z == 0 && by == 0) a.sample_idx_copy[bi] = si;
      }
      const int off = min(G::C - 1 - min(c, G::C - 1), age);      // ring stack: frames back from the newest one
      if (a.ring_slot) {
        img = newest - off;
        if (img < 0) img += a.ring_cap;
        if (generated && off == 0) img = newest >= 1 ? newest - 1 : newest + 1;   // any committed slot: value unused
      }
      const unsigned* src = reinterpret_cast<const unsigned*>(xb + (img * G::H + ir0) * G::H) + wdc;
      if (a.ring_slot && a.newest_frame && off == 0)              // the newest observation is not in the ring yet
        src = reinterpret_cast<const unsigned*>(a.newest_frame + (int64_t)ir0 * G::H) + wdc;
#pragma unroll
      for (int q = 0; q < LPT; ++q) raw[ci * LPT + q] = src[min(2 * RPARTS * q + rsub, nrows - 1) * WPR];
    }
    if (kImageFirst) { __builtin_amdgcn_sched_barrier(0); request_weights(); __builtin_amdgcn_sched_barrier(0); }
    if (tid < 256) s_lut[tid] = (float)((double)tid * a.coef);
    __syncthreads();
    if constexpr (FUSE) {
      if (f.mode == 2) {
        // action of env step e-1 (every workgroup, redundantly), then THIS workgroup's rows of the observation that
        // environment step returns: u32 word wd of row r is half ((21 r + wd) & 1) of 8-byte word (21 r + wd) >> 1
        const int64_t act = eps_greedy_action(s_q, f.n_actions, prm, f.e - 1);
        const int64_t counter = prm->counter[f.e];
#pragma unroll
        for (int ci = 0; ci < CPT; ++ci) {
          if (min(G::C - 1 - min(wc + CW * ci, G::C - 1), age) == 0) {   // every channel that shows the newest frame
#pragma unroll
            for (int q = 0; q < LPT; ++q) {
              const int w32 = (ir0 + min(2 * RPARTS * q + rsub, nrows - 1)) * WPR + wdc;
              const uint64_t v = synth_frame_word(f.seed, counter, act, w32 >> 1);
              raw[ci * LPT + q] = (unsigned)(v >> (32 * (w32 & 1)));
            }
          }
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wc + CW * ci;
      float* dst = lds + c * T::CS + rsub * G::RW + wd;
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        unsigned v = raw[ci * LPT + q];
        asm volatile("" : "+v"(v));  // keep the loads unconditional and batched (see igemm.hip)
        if (wd < WPR && 2 * RPARTS * q + rsub < nrows && c < G::C) {
#pragma unroll
          for (int b = 0; b < 4; ++b) dst[2 * RPARTS * q * G::RW + b * G::WPH] = s_lut[(v >> (8 * b)) & 0xffu];
        }
      }
    }
  } else if constexpr (G::H > 32) {
    // wide fp32 rows (conv1 fed with already-normalised floats: tests / generic callers, not a hot path):
    // flat element mapping
    constexpr int NEMAX = T::NR * G::H;                          // floats per channel block
    constexpr int LPT = (NEMAX + 63) / 64;
    constexpr int CPT = (G::C + NW - 1) / NW;
    float raw[CPT * LPT];
    const float* xf = reinterpret_cast<const float*>(a.x[z]);
    const int ne = nrows * G::H;
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + NW * ci;
      const float* src = xf + ((int64_t)(bi * G::C + min(c, G::C - 1)) * G::H + ir0) * G::H;
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        const int e = lane + 64 * q;
        raw[ci * LPT + q] = src[min(e, ne - 1)];
      }
    }
    if (kImageFirst) { __builtin_amdgcn_sched_barrier(0); request_weights(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + NW * ci;
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        const int e = lane + 64 * q;
        float v = raw[ci * LPT + q];
        asm volatile("" : "+v"(v));
        if (e < ne && c < G::C) {
          const int r = e / G::H, iw = e - r * G::H;
          lds[c * T::CS + r * G::RW + lds_col<G>(iw)] = v;
        }
      }
    }
  } else {
    constexpr int LR = G::H > 16 ? 32 : 16;                     // lanes per image row
    constexpr int RP = 64 / LR;                                  // rows per pass
    constexpr int LPT = (T::NR + RP - 1) / RP;
    constexpr int CPT = (G::C + NW - 1) / NW;
    float raw[CPT * LPT];
    const float* xf = reinterpret_cast<const float*>(a.x[z]);
    const int rsub = lane / LR, iw = lane % LR;
    const int iwc = min(iw, G::H - 1);
    const int col = lds_col<G>(iwc);
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + NW * ci;
      const float* src = xf + ((int64_t)(bi * G::C + min(c, G::C - 1)) * G::H + ir0) * G::H + iwc;
#pragma unroll
      for (int q = 0; q < LPT; ++q) raw[ci * LPT + q] = mega_ld<CIN>(src + min(RP * q + rsub, nrows - 1) * G::H);
    }
    if (kImageFirst && !CIN) { __builtin_amdgcn_sched_barrier(0); request_weights(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + NW * ci;
      float* dst = lds + c * T::CS + rsub * G::RW + col;
#pragma unroll
      for (int q = 0; q < LPT; ++q) {
        float v = raw[ci * LPT + q];
        asm volatile("" : "+v"(v));
        if (iw < G::H && RP * q + rsub < nrows && c < G::C) dst[RP * q * G::RW] = v;
      }
    }
    if constexpr (CIN) {
      __builtin_amdgcn_sched_barrier(0);
      request_weights_range(std::integral_constant<int, NJ0>{}, std::integral_constant<int, KS::NJ>{});
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  DRA_STAMP(TRR, 1);   // loads consumed, LDS writes issued
  __syncthreads();
  DRA_STAMP(TRR, 2);   // image staged

  // ---- MFMA: lane li owns output positions p0 + 32 t + li (clamped), half-wave h the odd channel of a pair
  const float* bptr[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int pj = min(32 * t + li, np - 1);
    const int poh = (p0 + pj) / G::OH, pow_ = (p0 + pj) - poh * G::OH;
    bptr[t] = lds + (2 * cp0 + h) * T::CS + ((poh - oh0) * G::S + t0 / G::KH) * G::RW + pow_;
  }
  f32x16 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int j = 0; j < KS::NJ; ++j) {
    const int cpl = j / KS::TW, tp = j - cpl * KS::TW;  // tap relative to the wave's base tap t0 (folded into bptr)
    const int kh = tp / G::KH, kw = tp - kh * G::KH;
    const int off = 2 * cpl * T::CS + kh * G::RW + (kw % G::S) * G::WPH + kw / G::S;
#pragma unroll
    for (int t = 0; t < PT; ++t)  // PT independent accumulation chains share the A operand
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[j], bptr[t][off], acc[t], 0, 0, 0);
  }
  DRA_STAMP(TRR, 3);   // this wave's MFMAs issued
  __syncthreads();  // every wave is done reading the image: reuse LDS for the 4-way reduction
  DRA_STAMP(TRR, 4);

  float* red = lds;  // [PT][NW waves][16][64]
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((t * NW + wave) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  // wave w finalises accumulator registers RPW*w .. RPW*w+RPW-1 (MFMA C/D rows (r&3) + 8*(r>>2) + 4*h); the NW partials are
  // added as a fixed balanced tree: run-to-run deterministic
  float* __restrict__ y = a.y[z];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int r = wave * RPW + q;
      const float* rt = red + (t * NW * 16) * 64;
      float s = (rt[(0 * 16 + r) * 64 + lane] + rt[(1 * 16 + r) * 64 + lane]) +
                (rt[(2 * 16 + r) * 64 + lane] + rt[(3 * 16 + r) * 64 + lane]);
      if constexpr (NW == 8)
        s += (rt[(4 * 16 + r) * 64 + lane] + rt[(5 * 16 + r) * 64 + lane]) +
             (rt[(6 * 16 + r) * 64 + lane] + rt[(7 * 16 + r) * 64 + lane]);
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = v2_act(s + bias_r[q], a.act);
      if (32 * t + li < np) mega_st<COH>(&y[((int64_t)(bi * G::OC + oc0 + row)) * G::P + p0 + 32 * t + li], v);
    }
  }
  DRA_STAMP(TRR, 5);   // reduction folded, stores issued
  if constexpr (COH) mega_publish(ms);
  DRA_STAMP_END(TRR);
}

// The weight slice fc4's forward workgroup with PHYSICAL index p (= blockIdx.x of the launch that follows: oneshot.h
// LinFwdSlabsOne<3136, 14, 2>::run with xcd_order) will stream: 64 rows x 224 floats of net z's [512][3136] weight matrix.
__device__ __forceinline__ void fc4_weight_prefetch(const ConvV2Args& a, const int p) {
  constexpr int I = 3136, KS = 14, KPS = I / KS, V = KPS / 4, TILES_N = 512 / 64, ROWS = 64;
  const int n_groups = KS * a.pf_nz;                  // (net, K slice) groups of TILES_N workgroups, one row tile (batch <= 32)
  if (p >= TILES_N * n_groups) return;
  const int bid = xcd_order(p, 0, n_groups, TILES_N);
  const int bn = bid % TILES_N, r = bid / TILES_N, sl = r % KS, z = r / KS;
  const float* __restrict__ wz = a.pf_w[z] + (int64_t)bn * ROWS * I + sl * KPS;
  constexpr int NVW = ROWS * V, RW = (NVW + 255) / 256;
  float4 t[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int e = min((int)threadIdx.x + 256 * q, NVW - 1), row = e / V, c4 = e - row * V;
    t[q] = *reinterpret_cast<const float4*>(wz + (int64_t)row * I + 4 * c4);
  }
#pragma unroll
  for (int q = 0; q < RW; ++q) asm volatile("" : : "v"(t[q].x), "v"(t[q].y), "v"(t[q].z), "v"(t[q].w));
}

template <class G, bool U8, int PT, int NW = 4>
__global__ void __launch_bounds__(64 * NW) conv_fwd_v2_kernel(const ConvV2Args a) {
  ActorFuse none;
  none.mode = 0;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.rider_z > 0) {
    // riders: the deferred fc4 segment of the previous update's optimizer step, the FIRST rider_z z-slices of the launch -- they are
    // dispatched first and stream while the launch's own workgroups sit in their latency phases
    if ((int)blockIdx.z < a.rider_z) {
      if constexpr (NW == 4) {
        const int k = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
        if (a.rider_first + k < a.rider_end) fc4_rider_run(a.rider, a.rider_first + k);
      }
      return;
    }
    bz -= a.rider_z;
  }
  if (a.rider_done_pending && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    __hip_atomic_store(a.rider_done_pending, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.rider_done_valid) __hip_atomic_store(a.rider_done_valid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (a.pf_nz > 0 && (int)blockIdx.z >= a.pf_first) {
    fc4_weight_prefetch(a, (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * ((int)blockIdx.z - a.pf_first)));
    return;
  }
  if (a.xcd_order) {
    // natural order: x = sample * TPG + tile group (fastest), y = output-channel tile, z = net; the sharing group is one
    // (net, sample): TPG * gridDim.y workgroups staging rows of the same input images
    constexpr int TPG = V2Tile<G, PT>::TPG;
    const int ny = gridDim.y, per = TPG * ny, groups = (a.pf_nz > 0 ? a.pf_first : (a.nz_real > 0 ? a.nz_real : (int)gridDim.z)) * a.batch;
    const int lin = bx + (int)gridDim.x * (by + ny * bz);       // (bz: the slice among the launch's OWN z-slices)
    const int v = xcd_order(lin, 0, groups, per);
    const int g = v / per, w = v - g * per;
    bz = g / a.batch;
    const int bi = g - bz * a.batch, grp = w / ny;
    by = w - grp * ny;
    bx = bi * TPG + grp;
  }
  conv_fwd_v2_body<G, U8, PT, NW, false>(a, none, bx, by, bz, false);
}

// conv1 of the ring actor's env step e with the head of step e-1 and the environment step in front (ActorFuse):
// grid = conv1's workgroups + 1 environment workgroup.
template <class G, int NW>
__global__ void __launch_bounds__(64 * NW) conv1_actor_fused_kernel(const ConvV2Args a, const ActorFuse f) {
  conv_fwd_v2_body<G, true, 1, NW, true>(a, f, blockIdx.x, blockIdx.y, blockIdx.z, blockIdx.x == gridDim.x - 1);
}

// ------------------------------------------------------------------------------------------------
// Persistent, software-pipelined forward for LARGE batches (the throughput regime: A2C / PPO minibatches, the
// batch-1024 MFMA-rate measurement).  The one-shot kernel above pays, per 32*PT positions, a full exposed memory
// latency plus staging and reduction around a short MFMA phase; at occupancy 3-6 that caps the fp32 MFMA pipe at
// ~45 % (SQ counters, profiles/r01e_pmc_conv_fwd_b1024_before.json).  Here a workgroup loads its 32 output
// channels' weights into registers ONCE and walks tile groups g = blockIdx.x, +gridDim.x, ...:
//     global loads of group g+1 -> registers          (in flight during the MFMA phase)
//     MFMA phase of group g from the LDS image
//     barrier; accumulators -> LDS reduction area; registers of group g+1 -> LDS image; barrier
//     4-way fold, bias, activation, store of group g  (overlaps the next group's loads)
// Same per-position arithmetic and summation order as the one-shot kernel: bit-identical outputs.
// Measured (batch 1024, fp32 MFMA fraction, one-shot multi-tile -> persistent): conv1 43.9 -> 43.3 %, conv2
// 45.8 -> 45.0 %, conv3 43.8 -> 48.6 %.  A further variant that flattened (sample, position) into one M axis to
// remove the per-sample tile padding (81 -> 96, 49 -> 64 positions) staged 2-3 whole samples per group at one
// wave per SIMD and LOST (conv2 34 %, conv3 37 %): it was removed.  What remains between ~45 % and the pipe's
// peak is the 16-23 % padding of conv2 / conv3 and the fp32 MFMA sharing its issue port with the staging VALU work.
template <class G, class T, bool U8, int NW = 4>
struct V2Stage {
  static constexpr int CPT = (G::C + NW - 1) / NW;                 // channels per wave
  static constexpr int LR = U8 ? 32 : (G::H > 16 ? 32 : 16);       // lanes per image row (u8: one u32 word per lane)
  static constexpr int RP = 64 / LR;                               // rows per pass
  static constexpr int LPT = (T::NR + RP - 1) / RP;
  static constexpr int N = CPT * LPT;
  static constexpr int COLS = U8 ? G::H / 4 : G::H;                // valid lanes per row
  static_assert(COLS <= LR, "one image row per LR lanes");

  template <class E>
  __device__ static __forceinline__ void load(const ConvV2Args& a, int z, int bi, int ir0, int nrows, int wave, int lane,
                                              E (&raw)[N]) {
    const int rsub = lane / LR, col = min(lane % LR, COLS - 1);
    const E* base = reinterpret_cast<const E*>(a.x[z]);
    constexpr int ROW = U8 ? G::H / 4 : G::H;                      // elements of E per image row
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = min(wave + NW * ci, G::C - 1);
      const E* src = base + ((int64_t)(bi * G::C + c) * G::H + ir0) * ROW + col;
#pragma unroll
      for (int q = 0; q < LPT; ++q) raw[ci * LPT + q] = src[min(RP * q + rsub, nrows - 1) * ROW];
    }
  }

  template <class E>
  __device__ static __forceinline__ void store(E (&raw)[N], float* __restrict__ img, const float* __restrict__ lut, int nrows,
                                               int wave, int lane) {
    const int rsub = lane / LR, cl = lane % LR;
#pragma unroll
    for (int ci = 0; ci < CPT; ++ci) {
      const int c = wave + NW * ci;
      if constexpr (U8) {
        float* dst = img + c * T::CS + rsub * G::RW + cl;
#pragma unroll
        for (int q = 0; q < LPT; ++q) {
          unsigned v = raw[ci * LPT + q];
          asm volatile("" : "+v"(v));
          if (cl < COLS && RP * q + rsub < nrows && c < G::C) {
#pragma unroll
            for (int b = 0; b < 4; ++b) dst[RP * q * G::RW + b * G::WPH] = lut[(v >> (8 * b)) & 0xffu];
          }
        }
      } else {
        float* dst = img + c * T::CS + rsub * G::RW + lds_col<G>(min(cl, G::H - 1));
#pragma unroll
        for (int q = 0; q < LPT; ++q) {
          float v = raw[ci * LPT + q];
          asm volatile("" : "+v"(v));
          if (cl < COLS && RP * q + rsub < nrows && c < G::C) dst[RP * q * G::RW] = v;
        }
      }
    }
  }
};

// Staging of a WHOLE fp32 image of a stride-2 layer (conv2's throughput shape: PT = 3 covers the sample) as 128-bit loads: the
// sample is one contiguous block of C * H * H floats, 13 loads per thread instead of 80 row-shaped dword loads with 20 of 32
// lanes active.  More than 64 loads per thread cannot even be in flight together (vmcnt), and the rest issue only as the
// first ones return -- in front of the MFMA phase they are ordered before: 8.2 us instead of 5.9 us per group
// (tools/phase_conv_big.py).  A float4 never straddles a row (H % 4 == 0); its elements 0 / 2 and 1 / 3 are neighbours in the
// two stride phases of the de-interleaved LDS row: two 64-bit writes.
template <class G, class T, int NT>
struct V2StageWide {
  // (instantiated for every layer, USED only where the kernel's WIDE condition holds: whole image, stride 2, H % 4 == 0)
  static constexpr int TOTAL = G::C * G::H * G::H / 4;               // float4s per sample
  static constexpr int N = (TOTAL + NT - 1) / NT;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));

  __device__ static __forceinline__ void load(const ConvV2Args& a, int z, int bi, int tid, f4 (&raw)[N]) {
    const f4* base = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(a.x[z]) + (int64_t)bi * G::C * G::H * G::H);
#pragma unroll
    for (int i = 0; i < N; ++i) raw[i] = base[min(tid + NT * i, TOTAL - 1)];
  }

  __device__ static __forceinline__ void store(f4 (&raw)[N], float* __restrict__ img, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      f4 v = raw[i];
      asm volatile("" : "+v"(v));
      const int idx = tid + NT * i;
      if (idx < TOTAL) {
        if constexpr (G::S == 1) {
          // stride 1 (conv3): [C][H][RW = H] in LDS IS the sample's layout in memory -- a straight 128-bit copy
          *reinterpret_cast<f4*>(img + 4 * idx) = v;
        } else {
          const int e = 4 * idx, c = e / (G::H * G::H), rem = e - c * (G::H * G::H), row = rem / G::H, col = rem - row * G::H;
          float* dst = img + c * T::CS + row * G::RW + col / 2;
          *reinterpret_cast<f2*>(dst) = f2{v.x, v.z};
          *reinterpret_cast<f2*>(dst + G::WPH) = f2{v.y, v.w};
        }
      }
    }
  }
};

// NW = waves per workgroup = K split.  4: two or three workgroups share a CU where LDS allows (conv1, conv3).  8: conv2, whose
// 51 KB fp32 image + partial-sum exchange leave room for ONE workgroup per CU -- and one wave per SIMD issues a 32x32x2 MFMA
// only every ~90 cycles (7.7 us for 192 MFMAs), two waves per SIMD keep the pipe dense (conv3's two workgroups: 288 MFMAs in
// 7.8 us; tools/phase_conv_big.py, profiles/r03h_phase_conv_big_before.json).
// SEQ: the partial sums of the PT tiles cross LDS one tile at a time through ONE tile's worth of exchange buffer (16 KB instead
// of PT x 16 KB): conv2's 51 KB image + 48 KB of partial sums allowed one workgroup per CU; with 67 KB two fit -- two waves per
// SIMD issue a 32x32x2 MFMA every 27 ns, one wave only every 39 ns (same sums in the same order: bit-identical results;
// starting the second workgroup half a group late was tried and loses 3-6 %).  Batch 1024, same box: 45.1 -> 48.9 %.
// WPE: waves per SIMD the register allocation must leave room for (conv3 with SEQ: 37 KB of LDS, three workgroups per CU).
template <class G, bool U8, int PT, int NW = 4, bool SEQ = false, int WPE = 1>
__global__ void __launch_bounds__(64 * NW, WPE) conv_fwd_v2_persist_kernel(const ConvV2Args a, const int n_groups) {
  using T = V2Tile<G, PT>;
  using ST = V2Stage<G, T, U8, NW>;
  using KS = typename G::template Split<NW>;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  constexpr int RPW = 16 / NW;                // accumulator rows a wave finalises
  using E = typename std::conditional<U8, unsigned, float>::type;
  static_assert(!U8 || G::S == 4, "u8 staging: stride-4 layer");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float s_lut[256];
  float* img = lds;                          // [C][NR][RW]
  float* red = lds + G::C * T::CS;           // [PT][NW waves][16][64]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int z = blockIdx.z;
  const int oc0 = blockIdx.y * 32;
  const float* __restrict__ wt = a.wt[z];
  const int cp0 = (G::CP >= NW) ? wave * KS::CPW : (wave % G::CP);
  const int t0 = (G::CP >= NW) ? 0 : (wave / G::CP) * KS::TW;
  [[maybe_unused]] const int TRR = TR_CONV1_F + (G::C == 4 ? 0 : (G::C == 32 ? 1 : 2));
  DRA_STAMP(TRR, 0);
  float areg[KS::NJ];
  {
    const float* wbase = wt + ((int64_t)(2 * cp0 + h) * G::KK + t0) * G::OC + oc0 + li;
#pragma unroll
    for (int j = 0; j < KS::NJ; ++j) {
      const int cpl = j / KS::TW, t = j - cpl * KS::TW;
      areg[j] = wbase[(2 * cpl * G::KK + t) * G::OC];
    }
  }
  float bias_r[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int r = wave * RPW + q;
    bias_r[q] = a.bias[z][oc0 + (r & 3) + 8 * (r >> 2) + 4 * h];
  }
  if (U8 && tid < 256) s_lut[tid] = (float)((double)tid * a.coef);
  float* __restrict__ y = a.y[z];

  int g = blockIdx.x;
  int bi = g / T::TPG, p0 = (g - bi * T::TPG) * PT * 32;
  int np = min(32 * PT, G::P - p0);
  int oh0 = p0 / G::OH;
  int nrows = ((p0 + np - 1) / G::OH - oh0) * G::S + G::KH;
  constexpr bool WIDE = !U8 && T::NR == G::H &&
                        ((G::S == 2 && G::H % 4 == 0 && G::WPH % 2 == 0 && T::CS % 2 == 0) ||
                         (G::S == 1 && G::RW == G::H && T::CS == G::H * G::H && (G::C * G::H * G::H) % 4 == 0));
  using SW = V2StageWide<G, T, 64 * NW>;
  E raw[WIDE ? 1 : ST::N];
  typename SW::f4 raw4[WIDE ? SW::N : 1];
  if constexpr (WIDE) SW::load(a, z, bi, tid, raw4);
  else ST::load(a, z, bi, oh0 * G::S, nrows, wave, lane, raw);
  __syncthreads();                                   // normalisation table visible
  if constexpr (WIDE) SW::store(raw4, img, tid);
  else ST::store(raw, img, s_lut, nrows, wave, lane);
  __syncthreads();
  DRA_STAMP(TRR, 2);   // prologue done: weights + first group staged
  for (;;) {
    // ---- next group's rows: requested now, consumed after the MFMA phase (always issued -- the last iteration
    // re-reads its own group -- so that no branch sits between the loads and the MFMA loop)
    const int gn = g + gridDim.x;
    const bool more = gn < n_groups;
    const int gl = more ? gn : g;
    const int bi_n = gl / T::TPG, p0_n = (gl - bi_n * T::TPG) * PT * 32;
    const int np_n = min(32 * PT, G::P - p0_n);
    const int oh0_n = p0_n / G::OH;
    const int nrows_n = ((p0_n + np_n - 1) / G::OH - oh0_n) * G::S + G::KH;
    if constexpr (WIDE) SW::load(a, z, bi_n, tid, raw4);
    else ST::load(a, z, bi_n, oh0_n * G::S, nrows_n, wave, lane, raw);
    __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks these loads to the end of the MFMA phase
    // ---- MFMA phase of the current group
    const float* bptr[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int pj = min(32 * t + li, np - 1);
      const int poh = (p0 + pj) / G::OH, pow_ = (p0 + pj) - poh * G::OH;
      bptr[t] = img + (2 * cp0 + h) * T::CS + ((poh - oh0) * G::S + t0 / G::KH) * G::RW + pow_;
    }
    f32x16 acc[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int j = 0; j < KS::NJ; ++j) {
      const int cpl = j / KS::TW, tp = j - cpl * KS::TW;
      const int kh = tp / G::KH, kw = tp - kh * G::KH;
      const int off = 2 * cpl * T::CS + kh * G::RW + (kw % G::S) * G::WPH + kw / G::S;
#pragma unroll
      for (int t = 0; t < PT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[j], bptr[t][off], acc[t], 0, 0, 0);
    }
    if (g == (int)blockIdx.x) DRA_STAMP(TRR, 1);   // (first group only) this wave's MFMAs issued
    __syncthreads();   // image fully consumed
    if (g == (int)blockIdx.x) DRA_STAMP(TRR, 3);   // every wave's MFMAs issued, the next group's rows have arrived
    if constexpr (!SEQ) {
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((t * NW + wave) * 16 + r) * 64 + lane] = acc[t][r];
      if (more) {
        if constexpr (WIDE) SW::store(raw4, img, tid);
        else ST::store(raw, img, s_lut, nrows_n, wave, lane);
      }
      __syncthreads();
      if (g == (int)blockIdx.x) DRA_STAMP(TRR, 4);   // next group staged
    }
    // ---- fold the K parts (four: same order as the one-shot kernel; eight: pairs, then pairs of pairs), bias, activation, store
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      if constexpr (SEQ) {
        if (t > 0) __syncthreads();                  // the previous tile's partial sums have been folded
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
        if (t == 0 && more) {
          if constexpr (WIDE) SW::store(raw4, img, tid);
          else ST::store(raw, img, s_lut, nrows_n, wave, lane);
        }
        __syncthreads();
        if (t == 0 && g == (int)blockIdx.x) DRA_STAMP(TRR, 4);
      }
      const float* rt = red + (SEQ ? 0 : t * NW * 16) * 64;
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int r = wave * RPW + q;
        float s = (rt[(0 * 16 + r) * 64 + lane] + rt[(1 * 16 + r) * 64 + lane]) +
                  (rt[(2 * 16 + r) * 64 + lane] + rt[(3 * 16 + r) * 64 + lane]);
        if constexpr (NW == 8)
          s += (rt[(4 * 16 + r) * 64 + lane] + rt[(5 * 16 + r) * 64 + lane]) +
               (rt[(6 * 16 + r) * 64 + lane] + rt[(7 * 16 + r) * 64 + lane]);
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = v2_act(s + bias_r[q], a.act);
        if (32 * t + li < np) y[((int64_t)(bi * G::OC + oc0 + row)) * G::P + p0 + 32 * t + li] = v;
      }
    }
    if (!more) break;
    g = gn; bi = bi_n; p0 = p0_n; np = np_n; oh0 = oh0_n; nrows = nrows_n;
  }
  DRA_STAMP(TRR, 5);
  DRA_STAMP_END(TRR);
}

// ------------------------------------------------------------------------------------------------
// Batch-1 conv2 / conv3 of the device actor with the reduction split over KZO workgroups per output tile.
// At batch 1 these layers are 6 resp. 4 workgroups of 8 waves whose latency IS the kernel's: 32 / 36 dependent 64-cycle
// MFMAs per wave with two waves sharing each SIMD's pipe (phase traces, profiles/r02x_phase_async.json: 1.9 - 2.1 us of
// MFMA per kernel, 4 x 2 kernels per agent step on the actor chain, which is as long as the update chain).  With the
// channel pairs halved over two workgroups (on different CUs) each wave issues 16 / 18 MFMAs and stages half the image;
// the halves are NOT reduced here: each workgroup stores its partial sums to its own plane (plane 0 carries the bias, no
// activation) and the CONSUMER adds the planes and applies the ReLU while it stages its input (KZI = 2) -- the next
// conv, or the fc4 GEMV.  Fixed order (plane 0 + bias) + plane 1: deterministic.
// grid (position tiles, (OC / 32) * KZO), 512 threads.
// (bx, by) = blockIdx of a plain launch; COH: the input planes come from, and the output planes go to, other workgroups of
// the SAME launch (the actor's one-launch env step): weights first, then the wait, then agent-scope loads / stores
template <class G, int KZI, int KZO, bool COH, bool CIN = COH>
__device__ __forceinline__ void conv_b1_split_body(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ wt,
                                                   const float* __restrict__ bias, float* __restrict__ y, int act, const int bx,
                                                   const int by, const MegaSync ms = MegaSync()) {
  using T = V2Tile<G, 1>;
  constexpr int NW = 8;
  constexpr int CPK = G::CP / KZO;            // channel pairs of this workgroup
  constexpr int CPW = CPK / NW;               // ... of a wave
  constexpr int NJ = CPW * G::KK;
  constexpr int CL = G::C / KZO;              // channels staged by this workgroup
  static_assert(CPK % NW == 0 && CL % NW == 0 && G::H <= 32, "channel split");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [CL][NR][RW] image, then the 8-way reduction
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int kz = by / (G::OC / 32);
  const int oc0 = (by - kz * (G::OC / 32)) * 32;
  const int p0 = bx * 32;
  const int np = min(32, G::P - p0);
  const int oh0 = p0 / G::OH, oh1 = (p0 + np - 1) / G::OH;
  const int ir0 = oh0 * G::S;
  const int nrows = (oh1 - oh0) * G::S + G::KH;
  [[maybe_unused]] const int TRR = TR_A_CONV1 + (G::C == 32 ? 1 : 2);
  DRA_STAMP(TRR, 0);
  // weights of this wave's channel pairs, then the image rows: every load of the workgroup in flight at once
  const int cp0 = kz * CPK + wave * CPW;
  float areg[NJ];
  {
    const float* wbase = wt + ((int64_t)(2 * cp0 + h) * G::KK) * G::OC + oc0 + li;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int cpl = j / G::KK, t = j - cpl * G::KK;
      areg[j] = wbase[(2 * cpl * G::KK + t) * G::OC];
    }
  }
  constexpr int RPW = 16 / NW;
  float bias_r[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int r = wave * RPW + q;
    bias_r[q] = (kz == 0) ? bias[oc0 + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
  }
  constexpr int LR = G::H > 16 ? 32 : 16;
  constexpr int RP = 64 / LR;
  constexpr int LPT = (T::NR + RP - 1) / RP;
  constexpr int CPT = CL / NW;
  float raw0[CPT * LPT];
  [[maybe_unused]] float raw1[CPT * LPT];
  const int rsub = lane / LR, iw = lane % LR;
  const int iwc = min(iw, G::H - 1);
  const int col = lds_col<G>(iwc);
  if constexpr (CIN) {
    __builtin_amdgcn_sched_barrier(0);   // the weight / bias loads above are in flight while this workgroup waits
    mega_wait(ms);
  }
#pragma unroll
  for (int ci = 0; ci < CPT; ++ci) {
    const int c = kz * CL + wave + NW * ci;
    const int64_t o = ((int64_t)c * G::H + ir0) * G::H + iwc;
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      const int64_t oo = o + (int64_t)min(RP * q + rsub, nrows - 1) * G::H;
      raw0[ci * LPT + q] = mega_ld<CIN>(x0 + oo);
      if constexpr (KZI == 2) raw1[ci * LPT + q] = mega_ld<CIN>(x1 + oo);
    }
  }
#pragma unroll
  for (int ci = 0; ci < CPT; ++ci) {
    float* dst = lds + (wave + NW * ci) * T::CS + rsub * G::RW + col;
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      float v = raw0[ci * LPT + q];
      if constexpr (KZI == 2) {
        v = v + raw1[ci * LPT + q];
        v = v > 0.f ? v : 0.f;
      }
      asm volatile("" : "+v"(v));
      if (iw < G::H && RP * q + rsub < nrows) dst[RP * q * G::RW] = v;
    }
  }
  DRA_STAMP(TRR, 1);
  __syncthreads();
  DRA_STAMP(TRR, 2);
  const int pj = min(li, np - 1);
  const int poh = (p0 + pj) / G::OH, pow_ = (p0 + pj) - poh * G::OH;
  const float* bptr = lds + (2 * (wave * CPW) + h) * T::CS + ((poh - oh0) * G::S) * G::RW + pow_;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int cpl = j / G::KK, tp = j - cpl * G::KK;
    const int kh = tp / G::KH, kw = tp - kh * G::KH;
    const int off = 2 * cpl * T::CS + kh * G::RW + (kw % G::S) * G::WPH + kw / G::S;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[j], bptr[off], acc, 0, 0, 0);
  }
  DRA_STAMP(TRR, 3);
  __syncthreads();
  DRA_STAMP(TRR, 4);
  float* red = lds;   // [8 waves][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  float* __restrict__ yo = y + (int64_t)kz * G::OC * G::P;
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int r = wave * RPW + q;
    float s = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) +
              (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
    s += (red[(4 * 16 + r) * 64 + lane] + red[(5 * 16 + r) * 64 + lane]) +
         (red[(6 * 16 + r) * 64 + lane] + red[(7 * 16 + r) * 64 + lane]);
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const float v = (KZO == 1) ? v2_act(s + bias_r[q], act) : s + bias_r[q];
    if (li < np) mega_st<COH>(&yo[(int64_t)(oc0 + row) * G::P + p0 + li], v);
  }
  DRA_STAMP(TRR, 5);
  if constexpr (COH) mega_publish(ms);
  DRA_STAMP_END(TRR);
}

template <class G, int KZI, int KZO>
__global__ void __launch_bounds__(512)
conv_b1_split_kernel(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ wt,
                     const float* __restrict__ bias, float* __restrict__ y, int act) {
  conv_b1_split_body<G, KZI, KZO, false>(x0, x1, wt, bias, y, act, blockIdx.x, blockIdx.y);
}

using VG1 = V2Geom<4, 84, 32, 8, 4>;
using VG2 = V2Geom<32, 20, 64, 4, 2>;
using VG3 = V2Geom<64, 9, 64, 3, 1>;

template <class G, int KZI, int KZO>
static int launch_conv_b1_split(const float* x0, const float* x1, const float* wt, const float* bias, float* y, int act,
                                hipStream_t st) {
  using T = V2Tile<G, 1>;
  constexpr size_t img = (size_t)(G::C / KZO) * T::CS * sizeof(float);
  constexpr size_t red = (size_t)8 * 16 * 64 * sizeof(float);
  constexpr size_t bytes = img > red ? img : red;
  static_assert(bytes <= 64 * 1024, "LDS per workgroup");
  hipLaunchKernelGGL((conv_b1_split_kernel<G, KZI, KZO>), dim3(G::TPS, (G::OC / 32) * KZO), dim3(512), bytes, st, x0, x1, wt, bias, y, act);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Library-internal (actor_env.h): batch-1 conv2 (layer 2) / conv3 (layer 3) writing KZO = 2 partial planes
// y[2][OC][P] (plane 0 includes the bias); layer 3 reads conv2's two planes (x0, x1) and applies the ReLU while staging.
int dra_conv_b1_split(int layer, const float* x0, const float* x1, const float* wt, const float* bias, float* y_planes,
                      void* stream) {
  if (!x0 || !wt || !bias || !y_planes) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  if (layer == 2) return launch_conv_b1_split<VG2, 1, 2>(x0, nullptr, wt, bias, y_planes, DRA_ACT_NONE, st);
  if (layer == 3) {
    if (!x1) return DRA_EINVAL;
    return launch_conv_b1_split<VG3, 2, 2>(x0, x1, wt, bias, y_planes, DRA_ACT_NONE, st);
  }
  return DRA_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// DRA_VAR_ACTOR_MEGA: conv3 + fc4 of one env step of the ring actor as ONE launch (DQN_agent.py:24-45: forward -> epsilon-greedy
// -> env.step -> next forward).  The actor chain is as long as the update chain and every one of its launches is a handful of
// workgroups waiting 1.5-2.7 us for operands after a ~2 us boundary.  conv3's 8 workgroups and fc4's 64 share a launch and hand
// conv3's planes over through a MegaSync counter: fc4's 6.4 MB of weights are in registers before conv3 has finished, what
// remains after the arrival is one agent-scope read of the activations.  Workgroups are dispatched in blockIdx order, so
// the producers are resident before a workgroup that waits for them.  Same arithmetic, same order as separate launches:
// bit-identical (tests/test_gpu_env_switches.py).  (Round 3 also measured ALL FOUR layers as one launch of 98 workgroups:
// -1.6 % -- a hand-over through memory costs what a launch boundary does; removed in round 4, profiles/r03c_ab.jsonl.)
struct ActorMegaArgs {
  ConvV2Args c1;
  ActorFuse f;
  const float *w2, *b2, *w3, *b3, *w4, *b4;
  float *y1, *y2p, *y3p, *h4;
  unsigned* flags;         // [3] arrivals of conv1 / conv2 / conv3 of THIS env step, zero at launch
  int* timeout_flag;
  const int* w4_valid;     // optional (DRA_VAR_DEFER_FC4): non-zero once w4 (a parameter copy's fc4 segment) is complete
};
constexpr int kMegaC3 = VG3::TPS * (VG3::OC / 32) * 2;
// fc4 rows per wave of the fused launch: 1 (64 workgroups of 8 rows).  Round 6 measured 2 (32 workgroups of 16 rows, so that the
// agent-scope read of conv3's planes after the arrival happens half as often per CU): 9.39 against 8.97 us per launch, same
// rate, same bits (profiles/r06p_*) -- 26 instead of 13 float4 of weights per lane in flight did not stream faster and the
// 128-register cap cost conv3's role.  DRA_EXP_MEGA_ROWS=2 builds it.
#ifndef DRA_EXP_MEGA_ROWS
#define DRA_EXP_MEGA_ROWS 1
#endif
constexpr int kMegaRows = DRA_EXP_MEGA_ROWS;
constexpr int kMegaFc = 512 / (8 * kMegaRows);

// fc4 role of the actor's fused launches: h4[row] = relu(b4[row] + <W4[row], relu(x0 + x1)>), one wave per row, the row's
// weights requested BEFORE the wait for conv3's two partial planes (same per-lane products and butterfly as
// actor_fc4_planes_lds_kernel: bit-identical); b = workgroup index within the role (8 rows each)
__device__ __forceinline__ void mega_fc4_role(const ActorMegaArgs& m, const int b, float* __restrict__ lds) {
  constexpr int I = VG3::OC * VG3::P, NV = I / 4, R = (NV + 63) / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (b * 8 + wave) * kMegaRows;
  if (m.w4_valid) {   // the copy's fc4 segment is completed by riders of the update running beside this graph: normally long done
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(m.w4_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > kMegaWaitTicks) {
          if (m.timeout_flag) __hip_atomic_store(m.timeout_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    __syncthreads();
  }
  float4 wv[kMegaRows][R];
  float bias[kMegaRows];
#pragma unroll
  for (int rr = 0; rr < kMegaRows; ++rr) {
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(m.w4 + (int64_t)(row0 + rr) * I);
#pragma unroll
    for (int q = 0; q < R; ++q) wv[rr][q] = w4[min(lane + 64 * q, NV - 1)];
    bias[rr] = m.b4[row0 + rr];
  }
  __builtin_amdgcn_sched_barrier(0);
  MegaSync ms;
  ms.wait = m.flags + 2; ms.wait_target = kMegaC3; ms.timeout_flag = m.timeout_flag;
  mega_wait(ms);
  constexpr int XQ = (I + 511) / 512;
  const float* x0 = m.y3p;
  const float* x1 = m.y3p + I;
  float xa[XQ], xc[XQ];
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int i = min((int)threadIdx.x + 512 * q, I - 1);
    xa[q] = mega_ld<true>(x0 + i);
    xc[q] = mega_ld<true>(x1 + i);
  }
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int i = (int)threadIdx.x + 512 * q;
    if (i < I) lds[i] = fmaxf(xa[q] + xc[q], 0.f);
  }
  __syncthreads();
  const float4* sx = reinterpret_cast<const float4*>(lds);
#pragma unroll
  for (int rr = 0; rr < kMegaRows; ++rr) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      float4 a = wv[rr][q];
      asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
      const float4 x = sx[min(lane + 64 * q, NV - 1)];
      if (lane + 64 * q < NV) acc += (a.x * x.x + a.y * x.y) + (a.z * x.z + a.w * x.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = acc + bias[rr];
      m.h4[row0 + rr] = v > 0.f ? v : 0.f;
    }
  }
}

// conv1 and conv2 keep their own launches -- a hand-over through memory costs about what a launch boundary does (stores
// acknowledged, the arrival count, the poll: ~2.4 us against ~2 us); what pays is fc4's 6.4 MB of weights arriving while
// conv3 computes.  grid = 8 conv3 workgroups + 64 fc4 workgroups.
__global__ void __launch_bounds__(512, kMegaRows == 2 ? 4 : 1) actor_c3fc4_kernel(const ActorMegaArgs m) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int b = blockIdx.x;
  if (b < kMegaC3) {
    MegaSync ms;
    ms.done = m.flags + 2;
    conv_b1_split_body<VG3, 2, 2, true, false>(m.y2p, m.y2p + VG2::OC * VG2::P, m.w3, m.b3, m.y3p, DRA_ACT_NONE, b % VG3::TPS, b / VG3::TPS, ms);
    return;
  }
  mega_fc4_role(m, b - kMegaC3, lds);
}

// Library-internal (actor_env.h): conv3 (from conv2's two partial planes, written by the previous launch) + the fc4 GEMV as
// one launch; flags[2] = conv3's arrival counter, zero at launch.
int dra_actor_c3fc4(const float* y2_planes, const float* w3, const float* b3, const float* w4, const float* b4, float* y3_planes,
                    float* h4, unsigned* flags, int* timeout_flag, void* stream) {
  return dra_actor_c3fc4_valid(y2_planes, w3, b3, w4, b4, y3_planes, h4, flags, timeout_flag, nullptr, stream);
}

int dra_actor_c3fc4_valid(const float* y2_planes, const float* w3, const float* b3, const float* w4, const float* b4, float* y3_planes,
                          float* h4, unsigned* flags, int* timeout_flag, const int* w4_valid, void* stream) {
  if (!y2_planes || !w3 || !b3 || !w4 || !b4 || !y3_planes || !h4 || !flags || !timeout_flag) return DRA_EINVAL;
  ActorMegaArgs m;
  memset(&m, 0, sizeof(m));
  m.w4_valid = w4_valid;
  m.w3 = w3; m.b3 = b3; m.w4 = w4; m.b4 = b4;
  m.y2p = const_cast<float*>(y2_planes); m.y3p = y3_planes; m.h4 = h4; m.flags = flags; m.timeout_flag = timeout_flag;
  constexpr size_t img3 = (size_t)(VG3::C / 2) * V2Tile<VG3, 1>::CS * sizeof(float);
  constexpr size_t red = (size_t)8 * 16 * 64 * sizeof(float);
  constexpr size_t xfc = (size_t)VG3::OC * VG3::P * sizeof(float);
  constexpr size_t m2 = img3 > red ? img3 : red;
  constexpr size_t bytes = m2 > xfc ? m2 : xfc;
  hipLaunchKernelGGL(actor_c3fc4_kernel, dim3(kMegaC3 + kMegaFc), dim3(512), bytes, dra_stream(stream), m);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ---- DRA_VAR_ACTOR_PERSIST: the whole agent step of the ring actor as one launch ({value, tag} hand-overs), actor_persist.h
#include "actor_persist.h"
int dra_actor_persist(const ActorPersistArgs* a, void* stream) {
  if (!a || !a->w1 || !a->w4 || !a->wh || !a->frames || !a->aring || !a->seq || !a->y1 || !a->y2p || !a->y3p || !a->h4 || !a->abort_word ||
      !a->timeout_flag || !a->pend_frame)
    return DRA_EINVAL;
  return launch_actor_persist(*a, dra_stream(stream));
}

// (Round 6 measured a hand-over-free form of this launch -- fc4 split along K into 16 slices x 4 row quarters, every workgroup
// computing its four conv3 planes itself on the vector ALU and the head folding the 16 partial sums -- in four layouts of the
// on-the-fly conv3: 13.8 / 16.6 / 13.2 us per launch against 9.0-9.6 for this one, 8 137 vs 9 190 updates/s
// (profiles/r06e_ab_actor_fly.jsonl, r06f-h_actor_fly_kernel_stats_*).  The reason is not the conv3 arithmetic: fc4's 6.4 MB of
// weights stream at ~30 GB/s per CU on the actor's 32 CUs (6.6 us), which the form above hides completely behind conv3's
// workgroups; any form that makes the weight requests wait behind conv3's inputs puts that stream back on the path.  Removed.)

template <class G, bool U8, int PT, int NW = 4>
static int launch_conv_v2_pt(const ConvV2Args& a, int nz, hipStream_t st) {
  using T = V2Tile<G, PT>;
  constexpr size_t img = (size_t)G::C * T::CS * sizeof(float);
  constexpr size_t red = (size_t)PT * NW * 16 * 64 * sizeof(float);
  constexpr size_t bytes = img > red ? img : red;
  static_assert(bytes <= 160 * 1024, "LDS per workgroup");
  static DraLdsAttr lds_attr;
  if (bytes > 64 * 1024)
    if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&conv_fwd_v2_kernel<G, U8, PT, NW>), bytes)) return rc;
  ConvV2Args ax = a;
  ax.xcd_order = dra_xcd_order_enabled();
  int gz = nz;
  if (ax.pf_nz > 0) {
    const int per_z = T::TPG * a.batch * (G::OC / 32);
    // prefetch workgroup p must run on XCD p mod 8: the slices in front of it hold a multiple of 8 workgroups, 256 threads each
    if (NW != 4 || !ax.xcd_order || (per_z * nz) % 8 != 0) ax.pf_nz = 0;
    else { ax.pf_first = nz; gz = nz + (8 * 14 * ax.pf_nz + per_z - 1) / per_z; }
  }
  if (g_rider_next.armed) {   // (dra_conv_attach_rider: riders behind the launch's own slices, and / or the done words)
    g_rider_next.armed = false;
    ax.rider_done_pending = g_rider_next.done_pending;
    ax.rider_done_valid = g_rider_next.done_valid;
    if (g_rider_next.count > 0) {
      if (NW != 4) return DRA_EINVAL;           // a rider workgroup is 256 threads
      const int per_z = T::TPG * a.batch * (G::OC / 32);
      ax.rider = g_rider_next.rider;
      ax.rider_first = g_rider_next.first;
      ax.rider_end = g_rider_next.first + g_rider_next.count;
      if (ax.pf_nz > 0) return DRA_EINVAL;      // (the prefetch slices count from the launch's own: not combined)
      ax.nz_real = nz;
      ax.rider_z = (g_rider_next.count + per_z - 1) / per_z;
      gz += ax.rider_z;
    }
  }
  hipLaunchKernelGGL((conv_fwd_v2_kernel<G, U8, PT, NW>), dim3(T::TPG * a.batch, G::OC / 32, gz), dim3(64 * NW), bytes, st, ax);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// batch-1 launches (the actor's forward) use 8 waves per workgroup (the DRA_CONV_B1_WAVES=4 A/B switch of round 3 is retired: the
// four-wave form lost on every layer, DESIGN_HISTORY.md section 4)
static constexpr int conv_b1_waves() { return 8; }

template <class G, bool U8, int PT, int NW = 4, bool SEQ = false, int WPE = 1>
static int launch_conv_v2_persist(const ConvV2Args& a, int nz, hipStream_t st) {
  using T = V2Tile<G, PT>;
  constexpr size_t bytes = ((size_t)G::C * T::CS + (size_t)(SEQ ? 1 : PT) * NW * 16 * 64) * sizeof(float);
  static_assert(bytes <= 159 * 1024, "LDS per workgroup (+1 KB normalisation table)");
  static DraLdsAttr lds_attr;
  if (bytes > 64 * 1024)
    if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&conv_fwd_v2_persist_kernel<G, U8, PT, NW, SEQ, WPE>), bytes)) return rc;
  const int n_groups = T::TPG * a.batch;
  // exactly as many workgroups as the chip holds at this kernel's occupancy (registers / LDS: 1-3 per CU), each
  // walking several groups, so that the weight loads and the pipeline prologue are amortised
  static int resident = -1;
  if (resident < 0) {
    int per_cu = 0, dev = 0, n_cu = 0;
    DRA_HIP(hipGetDevice(&dev));
    DRA_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    DRA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&conv_fwd_v2_persist_kernel<G, U8, PT, NW, SEQ, WPE>),
                                                         64 * NW, bytes));
    resident = (per_cu > 0 ? per_cu : 1) * n_cu;
  }
  const int lanes = (G::OC / 32) * nz;
  const int per = resident / lanes > 0 ? resident / lanes : 1;
  const int nwg = n_groups < per ? n_groups : per;
  hipLaunchKernelGGL((conv_fwd_v2_persist_kernel<G, U8, PT, NW, SEQ, WPE>), dim3(nwg, G::OC / 32, nz), dim3(64 * NW), bytes, st, a, n_groups);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------------
// conv1's own throughput shape (plain uint8 NCHW batches >= 128: A2C / PPO minibatches).  The K-split kernels above spend
// 2.6 us of staging + partial-sum exchange per 1.7 us of MFMA work per group on this layer (tools/phase_conv_big.py: 44-48 %
// of the fp32-MFMA peak at batch 1024).  conv1 is the one layer whose WHOLE K fits a wave's registers (K = 256: 128 operand
// registers), so here
//   * every wave keeps all of K and owns whole 32-position tiles: no cross-wave exchange, no reduction buffer, no barrier
//     inside a group;
//   * the four K quarters of the latency shape (channel pair x tap half) stay four separate accumulation chains per tile and
//     are folded (q0 + q1) + (q2 + q3) in registers: the same sums in the same order -- bit-identical with every other
//     shape -- and four independent MFMA chains per wave;
//   * the frames stay UINT8 in LDS (28 KB per sample instead of 113 KB as fp32): one or two whole samples per workgroup
//     iteration, staged by a straight 128-bit copy; an operand is a byte read (4 px = 1 dword per output column: consecutive
//     lanes hit consecutive banks without any de-interleaving) + the exact-normalisation table (f32(f64(v) * coef));
//   * 13 tiles per sample over 4 waves: two samples per iteration (26 tiles: 7 / 7 / 6 / 6) once the launch has two
//     iterations' worth of samples per resident workgroup, else one (4 / 3 / 3 / 3: more workgroups).
//   * small batches (fewer samples than resident workgroups): a sample's 13 tiles are split over `tp` = 2 or 4 workgroups (each
//     stages the whole 28 KB sample), so that the launch still has two workgroups per CU.
__global__ void __launch_bounds__(256, 2) conv1_fwd_u8_tp_kernel(const ConvV2Args a, const int n_groups, const int spg, const int tp) {
  using G = VG1_;
  constexpr int IMG = G::C * G::H * G::H;                    // bytes per sample
  constexpr int TPSAMP = G::TPS;                             // 13 position tiles per sample
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int z = blockIdx.z;
  DRA_STAMP(TR_CONV1_F, 0);
  const float* __restrict__ wt = a.wt[z];
  float areg[4][32];
  if ((reinterpret_cast<uintptr_t>(wt) & 15) == 0) {
    // the 32 KB of weights once per workgroup through LDS (the frame region, before the first frames): every wave loading
    // all of K by itself is 4 x 32 KB of L2 reads per workgroup -- 7 us of prologue at batch 256 (512 workgroups)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* wsrc = reinterpret_cast<const f4*>(wt);
    f4* wl = reinterpret_cast<f4*>(smem);
    f4 wr[G::K * G::OC / 4 / 256];
#pragma unroll
    for (int i = 0; i < G::K * G::OC / 4 / 256; ++i) wr[i] = wsrc[tid + 256 * i];
#pragma unroll
    for (int i = 0; i < G::K * G::OC / 4 / 256; ++i) wl[tid + 256 * i] = wr[i];
    __syncthreads();
    const float* wls = reinterpret_cast<const float*>(smem);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 32; ++j) areg[q][j] = wls[((2 * (q & 1) + h) * G::KK + (q >> 1) * 32 + j) * G::OC + li];
    __syncthreads();
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* wbase = wt + ((int64_t)(2 * (q & 1) + h) * G::KK + (q >> 1) * 32) * G::OC + li;
#pragma unroll
      for (int j = 0; j < 32; ++j) areg[q][j] = wbase[j * G::OC];
    }
  }
  __shared__ float s_bias[G::OC];
  // (one copy of the normalisation table: eight copies spread over the banks by lane measured 86 against 81 us at batch 1024)
  __shared__ float s_lut[256];
  if (tid < G::OC) s_bias[tid] = a.bias[z][tid];
  s_lut[tid] = (float)((double)tid * a.coef);
  const unsigned char* __restrict__ x = reinterpret_cast<const unsigned char*>(a.x[z]);
  float* __restrict__ y = a.y[z];
  DRA_STAMP(TR_CONV1_F, 2);
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int b0 = (g / tp) * spg, ns = min(spg, a.batch - b0), part = g - (g / tp) * tp;
    // tiles of this workgroup: all ns * 13 of them, or (tp > 1, ns == 1) the part-th share of the sample's 13
    const int t_lo = tp > 1 ? part * TPSAMP / tp : 0, t_hi = tp > 1 ? (part + 1) * TPSAMP / tp : ns * TPSAMP;
    {
      // straight copy of ns whole samples (16-byte aligned: 28 224 = 16 x 1764)
      const uint4* src = reinterpret_cast<const uint4*>(x + (int64_t)b0 * IMG);
      uint4* dst = reinterpret_cast<uint4*>(smem);
      const int n16 = ns * (IMG / 16);
      uint4 raw[2 * (IMG / 16 + 255) / 256];
#pragma unroll
      for (int i = 0; i < 2 * (IMG / 16 + 255) / 256; ++i) raw[i] = src[min(tid + 256 * i, n16 - 1)];
#pragma unroll
      for (int i = 0; i < 2 * (IMG / 16 + 255) / 256; ++i)
        if (tid + 256 * i < n16) dst[tid + 256 * i] = raw[i];
    }
    __syncthreads();
    for (int tile = t_lo + wave; tile < t_hi; tile += 4) {
      const int s = tile / TPSAMP, p0 = (tile - s * TPSAMP) * 32;
      const int p = min(p0 + li, G::P - 1), oh = p / G::OH, ow = p - oh * G::OH;
      const unsigned char* bp = smem + s * IMG + h * (G::H * G::H) + (oh * G::S) * G::H + ow * G::S;
      f32x16 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      // operand o = 4 j + q (j-th k-step of quarter q): byte read two chunks ahead, table read one chunk ahead of its MFMA --
      // left alone the scheduler issues [byte read, wait, table read, wait, MFMA] per operand (two exposed LDS latencies each)
      constexpr int CH = 8, NCHK = 128 / CH;
      auto boff = [](int o) {
        const int j = o >> 2, q = o & 3, t = (q >> 1) * 32 + j, kh = t / G::KH, kw = t - kh * G::KH;
        return (q & 1) * 2 * (G::H * G::H) + kh * G::H + kw;
      };
      unsigned u[2][CH];
      float bv[2][CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) { u[0][i] = bp[boff(i)]; u[1][i] = bp[boff(CH + i)]; }
      // (the table, not the vector ALU: v_cvt_f64_u32 + v_mul_f64 + v_cvt_f32_f64 per operand compete with the MFMAs for the
      // SIMD's issue port -- measured 87 against 81 us at batch 1024)
#pragma unroll
      for (int i = 0; i < CH; ++i) bv[0][i] = s_lut[u[0][i]];
#pragma unroll
      for (int c = 0; c < NCHK; ++c) {
        if (c + 1 < NCHK) {
#pragma unroll
          for (int i = 0; i < CH; ++i) bv[(c + 1) & 1][i] = s_lut[u[(c + 1) & 1][i]];
        }
        if (c + 2 < NCHK) {
#pragma unroll
          for (int i = 0; i < CH; ++i) u[c & 1][i] = bp[boff((c + 2) * CH + i)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int o = c * CH + i;
          acc[o & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[o & 3][o >> 2], bv[c & 1][i], acc[o & 3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int bi = b0 + s;
      if (p0 + li < G::P) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sum = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          y[((int64_t)bi * G::OC + row) * G::P + p0 + li] = v2_act(sum + s_bias[row], a.act);
        }
      }
    }
    __syncthreads();     // every wave is done with the frames before the next group is staged
  }
  DRA_STAMP(TR_CONV1_F, 5);
  DRA_STAMP_END(TR_CONV1_F);
}

static int launch_conv1_u8_tp(const ConvV2Args& a, int nz, hipStream_t st) {
  using G = VG1_;
  constexpr int IMG = G::C * G::H * G::H;
  static int resident = -1;
  if (resident < 0) {
    int dev = 0, n_cu = 0;
    DRA_HIP(hipGetDevice(&dev));
    DRA_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    resident = 2 * n_cu;
    static_assert(2 * IMG >= G::K * G::OC * 4, "the weights pass through the frame region");
  }
  static DraLdsAttr lds_attr;
  if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&conv1_fwd_u8_tp_kernel), (size_t)2 * IMG)) return rc;
  const int per = resident / nz > 0 ? resident / nz : 1;
  const int spg = a.batch >= 2 * per ? 2 : 1;
  const int tp = spg == 2 || a.batch >= per ? 1 : (2 * a.batch >= per ? 2 : 4);
  const int n_groups = ((a.batch + spg - 1) / spg) * tp;
  const int nwg = n_groups < per ? n_groups : per;
  const size_t lds_bytes = spg == 2 ? (size_t)2 * IMG : (size_t)(IMG > G::K * G::OC * 4 ? IMG : G::K * G::OC * 4);
  hipLaunchKernelGGL(conv1_fwd_u8_tp_kernel, dim3(nwg, 1, nz), dim3(256), lds_bytes, st, a, n_groups, spg, tp);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------------
// DRA_VAR_FWD_CHAIN (round 6): conv1 -> conv2 -> conv3 of the UPDATE's forward pass (both nets, batch <= 32, ring-direct conv1)
// as ONE launch.  The three launches were 10.8 + 8.8 + 9.1 us of kernel span with 2.4 + 3.2 us between them, each made of
// workgroups that first wait 2-3.8 us for their operands (profiles/r06m_phase_async.json); a workgroup of layer L+1 needs only
// the planes of ITS sample.  Here the 832 + 384 + 256 workgroups are one grid in dependency order (a workgroup waits only for
// workgroups dispatched before it), every (net, sample) has an arrival counter per layer: a producer stores agent-scope,
// completes its stores and counts itself; a consumer requests its weights FIRST, then polls its sample's counter (bounded), then
// reads the planes with agent-scope loads.  The counters are never reset: a wait is for (epoch + 1) x (workgroups per sample)
// arrivals, `epoch` = chains completed, bumped by the update's head kernel (learner.hip).  Same arithmetic in the same order as
// the three launches: bit-identical activations (tests/test_gpu_agents.py::test_forward_chain_is_bit_identical).
struct FwdChainArgs {
  ConvV2Args c1, c2, c3;
  unsigned* done1;           // [nz * batch] x kChainPad: arrivals of conv1's workgroups per (net, sample), one 128-byte line each
                             // (64 counters in two lines polled by 384 workgroups: the chain measured 43 us against 35)
  unsigned* done2;           // ... of conv2's
  const unsigned* epoch;
  int* timeout_flag;
  int n1, n2, n3;            // workgroups per role
  // DRA_VAR_DEFER_FC4: the LAST n_riders workgroups of the grid step the deferred fc4 segment of the previous update (they are
  // dispatched as conv1's workgroups leave, beside the 640 of conv2 / conv3, which are in front of them for the slots); the actor
  // copy is written through, every rider counts itself, the last one lowers `pending` and marks the copy valid
  DraFc4Rider rider;
  int n_riders, rider_after;
  unsigned* rider_count;
  int* rider_pending;
  int* rider_valid;
  // DRA_VAR_FLAG_SYNC: the first workgroup counts this launch as STARTED (a fire-and-forget agent-scope add): everything the
  // stream ran before this launch -- the previous update's optimizer -- is complete and visible, which is what the actor launch
  // on the other stream polls for instead of waiting for an event (learner.hip step_lane)
  unsigned long long* announce;
  // DRA_VAR_HEAD_CHAIN: a word the first workgroup sets to zero -- the arrival counter of the head -> fc4-backward hand-over of the
  // PREVIOUS update's launch (complete when this launch starts; the next such launch of this update counts from zero again)
  unsigned* zero_word;
};

__global__ void __launch_bounds__(256, 4) conv_fwd_chain_kernel(const FwdChainArgs a) {
  ActorFuse none;
  none.mode = 0;
  int b = blockIdx.x;
  const int batch = a.c1.batch;
  if (a.announce && b == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(a.announce, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.zero_word && b == 0 && threadIdx.x == 0) __hip_atomic_store(a.zero_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the riders sit behind `rider_after` convolution workgroups of the grid
  if (b >= a.rider_after && b < a.rider_after + a.n_riders) {
    fc4_rider_run<true>(a.rider, b - a.rider_after);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(a.rider_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)a.n_riders - 1u) {
        __hip_atomic_store(a.rider_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.rider_pending, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.rider_valid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  if (b >= a.rider_after) b -= a.n_riders;
  // (first = index of a role's first workgroup in the GRID: the XCD a workgroup runs on is blockIdx mod 8)
  const int f2 = a.n1 + (a.n1 >= a.rider_after ? a.n_riders : 0), f3 = a.n1 + a.n2 + (a.n1 + a.n2 >= a.rider_after ? a.n_riders : 0);
  const int f1 = 0 >= a.rider_after ? a.n_riders : 0;
  if (b < a.n1) {
    constexpr int TPG = V2Tile<VG1, 1>::TPG, per = TPG * (VG1::OC / 32);
    const int v = a.c1.xcd_order ? xcd_order(b, f1, a.n1 / per, per) : b;
    const int g = v / per, w = v - g * per;
    const int bz = g / batch, bi = g - bz * batch;
    MegaSync ms;
    ms.done = a.done1 + g * kChainPad;
    conv_fwd_v2_body<VG1, true, 1, 4, false, true, false>(a.c1, none, bi * TPG + w, 0, bz, false, ms);
    return;
  }
  b -= a.n1;
  if (b < a.n2) {
    constexpr int TPG = V2Tile<VG2, 1>::TPG, ny = VG2::OC / 32, per = TPG * ny;
    const int v = a.c2.xcd_order ? xcd_order(b, f2, a.n2 / per, per) : b;
    const int g = v / per, w = v - g * per;
    const int bz = g / batch, bi = g - bz * batch, grp = w / ny, by = w - grp * ny;
    MegaSync ms;
    ms.wait = a.done1 + g * kChainPad; ms.wait_target = V2Tile<VG1, 1>::TPG * (VG1::OC / 32); ms.epoch = a.epoch; ms.timeout_flag = a.timeout_flag;
    ms.done = a.done2 + g * kChainPad;
    conv_fwd_v2_body<VG2, false, 1, 4, false, true, true>(a.c2, none, bi * TPG + grp, by, bz, false, ms);
    return;
  }
  b -= a.n2;
  {
    constexpr int TPG = V2Tile<VG3, 1>::TPG, ny = VG3::OC / 32, per = TPG * ny;
    const int v = a.c3.xcd_order ? xcd_order(b, f3, a.n3 / per, per) : b;
    const int g = v / per, w = v - g * per;
    const int bz = g / batch, bi = g - bz * batch, grp = w / ny, by = w - grp * ny;
    MegaSync ms;
    ms.wait = a.done2 + g * kChainPad; ms.wait_target = V2Tile<VG2, 1>::TPG * (VG2::OC / 32); ms.epoch = a.epoch; ms.timeout_flag = a.timeout_flag;
    conv_fwd_v2_body<VG3, false, 1, 4, false, false, true>(a.c3, none, bi * TPG + grp, by, bz, false, ms);
  }
}

// Library-internal (actor_env.h): the NEXT dra_conv_fwd_chain launch counts itself in *count when it starts (DRA_VAR_FLAG_SYNC)
static unsigned long long* g_chain_announce = nullptr;
void dra_conv_chain_attach_announce(unsigned long long* count) { g_chain_announce = count; }
static unsigned* g_chain_zero = nullptr;
void dra_conv_chain_attach_zero(unsigned* word) { g_chain_zero = word; }   // (the NEXT dra_conv_fwd_chain launch zeroes *word when it starts)

// Library-internal (actor_env.h): conv1 (ring-direct, as dra_conv1_fwd_koc_ringbatch) + conv2 + conv3 (as dra_conv_fwd_koc) of nz nets.
int dra_conv_fwd_chain(const void* frames, const int64_t* idx, int64_t* idx_copy, const int64_t* idx_tagged,
                       const unsigned long long* update_seq, const int64_t* newest_off, int nz, const float* const* w1,
                       const float* const* b1, float* const* y1, const float* const* w2, const float* const* b2, float* const* y2,
                       const float* const* w3, const float* const* b3, float* const* y3, int batch, double u8_coef,
                       unsigned* done_counters, const unsigned* epoch, int* timeout_flag, const DraFc4Rider* rider,
                       unsigned* rider_count, int* rider_pending, int* rider_valid, void* stream) {
  if (!frames || !idx || !newest_off || nz < 1 || nz > DRA_MAX_Z || batch < 1 || batch > 32 || !w1 || !b1 || !y1 || !w2 || !b2 || !y2 ||
      !w3 || !b3 || !y3 || !done_counters || !epoch || !timeout_flag)
    return DRA_EINVAL;
  if ((idx_tagged != nullptr) != (update_seq != nullptr)) return DRA_EINVAL;
  if (g_rider_next.armed) { g_rider_next.armed = false; return DRA_EINVAL; }   // (the chain's riders come through `rider`)
  FwdChainArgs a;
  a.announce = g_chain_announce;
  g_chain_announce = nullptr;
  a.zero_word = g_chain_zero;
  g_chain_zero = nullptr;
  for (int z = 0; z < nz; ++z) {
    if (!w1[z] || !b1[z] || !y1[z] || !w2[z] || !b2[z] || !y2[z] || !w3[z] || !b3[z] || !y3[z]) return DRA_EINVAL;
    a.c1.x[z] = frames; a.c1.wt[z] = w1[z]; a.c1.bias[z] = b1[z]; a.c1.y[z] = y1[z];
    a.c1.idx_bias[z] = newest_off[z] - (VG1::C - 1);
    a.c2.x[z] = y1[z]; a.c2.wt[z] = w2[z]; a.c2.bias[z] = b2[z]; a.c2.y[z] = y2[z];
    a.c3.x[z] = y2[z]; a.c3.wt[z] = w3[z]; a.c3.bias[z] = b3[z]; a.c3.y[z] = y3[z];
  }
  a.c1.sample_idx = idx; a.c1.sample_idx_copy = idx_copy; a.c1.sample_idx_tagged = idx_tagged; a.c1.sample_seq = update_seq;
  ConvV2Args* cs[3] = {&a.c1, &a.c2, &a.c3};
  for (ConvV2Args* c : cs) {
    c->batch = batch; c->act = DRA_ACT_RELU; c->coef = 1.0; c->ring_slot = nullptr; c->ring_cap = 0; c->stack_age = nullptr;
    c->slot_seq = nullptr; c->slot_entries = 0; c->slot_stride = 0; c->newest_frame = nullptr;
    c->xcd_order = dra_xcd_order_enabled();
  }
  a.c1.coef = u8_coef;
  a.done1 = done_counters; a.done2 = done_counters + DRA_MAX_Z * 32 * kChainPad; a.epoch = epoch; a.timeout_flag = timeout_flag;
  a.n1 = nz * batch * V2Tile<VG1, 1>::TPG * (VG1::OC / 32);
  a.n2 = nz * batch * V2Tile<VG2, 1>::TPG * (VG2::OC / 32);
  a.n3 = nz * batch * V2Tile<VG3, 1>::TPG * (VG3::OC / 32);
  a.n_riders = 0; a.rider_after = 0;
  if (rider) {
    if (!rider_count || !rider_pending || !rider_valid) return DRA_EINVAL;
    a.rider = *rider; a.n_riders = fc4_rider_blocks(rider->count4);
    a.rider_count = rider_count; a.rider_pending = rider_pending; a.rider_valid = rider_valid;
  }
  // the riders are the LAST workgroups of the grid: dispatched as conv1's workgroups leave, behind conv2's and conv3's for the slots
  // (behind conv2 they measured 8 376, behind conv1 7 557, in front of everything 9 347 against 9 750 updates/s at the end:
  // riders in front of waiting workgroups hold the slots those need -- profiles/r06ze_ab_rider_position.jsonl)
  a.rider_after = a.n1 + a.n2 + a.n3;
  constexpr size_t i1 = (size_t)VG1::C * V2Tile<VG1, 1>::CS, i2 = (size_t)VG2::C * V2Tile<VG2, 1>::CS, i3 = (size_t)VG3::C * V2Tile<VG3, 1>::CS;
  constexpr size_t red = (size_t)4 * 16 * 64;
  constexpr size_t m12 = i1 > i2 ? i1 : i2, m3 = i3 > red ? i3 : red;
  constexpr size_t bytes = (m12 > m3 ? m12 : m3) * sizeof(float);
  static_assert(bytes <= 40 * 1024, "four chain workgroups per CU");
  hipLaunchKernelGGL(conv_fwd_chain_kernel, dim3(a.n1 + a.n2 + a.n3 + a.n_riders), dim3(256), bytes, dra_stream(stream), a);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Tile shape by problem size: one tile per workgroup (latency shape) until the launch has several workgroups
// per CU slot anyway, then PTBIG tiles per workgroup (throughput shape).  DRA_CONV_PT=1 forces the latency shape.
constexpr int kConvNw8MaxBatch = 16;
static int g_conv_pt_threshold = -1;
template <class G, bool U8, int PTBIG>
static int launch_conv_v2(const ConvV2Args& a, int nz, hipStream_t st) {
  if (g_conv_pt_threshold < 0) {
    const char* e = getenv("DRA_CONV_PT_BATCH");   // batch from which the throughput shape is used (0 = never)
    g_conv_pt_threshold = e ? atoi(e) : 128;
  }
  if (PTBIG > 1 && g_conv_pt_threshold > 0 && a.batch >= g_conv_pt_threshold && !a.ring_slot) {
    if (g_rider_next.armed) { g_rider_next.armed = false; return DRA_EINVAL; }   // riders ride in the latency shape only
    // The throughput shapes: persistent pipelined workgroups.  One measured winner per layer is left (the A/B switches
    // DRA_CONV_PERSIST, DRA_CONV1_TP, DRA_CONV1_SEQ, DRA_CONV2_MODE, DRA_CONV3_SEQ of rounds 3-4 are retired; the records are in
    // DESIGN_HISTORY.md section 4 and profiles/r03*_conv_big*, r04*_conv_big*):
    if constexpr (U8 || G::H <= 32) {
      if constexpr (U8 && G::C == 4) {
        // conv1 on uint8 frames from 384 samples per net on: its own throughput kernel (all of K in registers); below, the K-split
        // form's smaller work units fill the chip better -- batch 256: 26.5 against 27.2 us; 512: 48.0 against 44.2; 1024: 91
        // against 81; 2048: 149 us = 57 % of the fp32-MFMA peak
        if (a.batch >= 384 && !a.sample_idx && !a.newest_frame) return launch_conv1_u8_tp(a, nz, st);
      }
      // conv1: 64 MFMAs per wave and group against ~2.6 us of staging / exchange -- the one-tile partial-sum exchange (49 KB of
      // LDS), three workgroups per CU
      if constexpr (G::C == 4) return launch_conv_v2_persist<G, U8, PTBIG, 4, true, 3>(a, nz, st);
      // conv2 (51 KB fp32 image per sample): one tile's partial sums at a time, two workgroups per CU (an eight-wave form and
      // the round-2 four-wave form lost)
      if constexpr (!U8 && G::C == 32) return launch_conv_v2_persist<G, U8, PTBIG, 4, true>(a, nz, st);
      // conv3: the plain persistent form (the one-tile exchange measured no different: 55.1 / 54.3 / 54.5 % at batch 1024)
      return launch_conv_v2_persist<G, U8, PTBIG>(a, nz, st);
    }
    return launch_conv_v2_pt<G, U8, PTBIG>(a, nz, st);
  }
  // the eight-wave latency shape (K split eight ways: half the MFMA chain per wave) up to kConvNw8MaxBatch samples: the batch-1
  // actor's launches since round 3, and from round 5 the 8 / 16-environment rollout steps of the actor-critic agents (a2c_pixel
  // 245.3 k -> 249.0 k, ppo_pixel 134.1 k -> 135.6 k env-steps/s, same-call A/B: profiles/r05g8_bench_agents_nw8.txt)
  if (a.batch <= kConvNw8MaxBatch && !a.pf_nz && !a.sample_idx && conv_b1_waves() == 8) return launch_conv_v2_pt<G, U8, 1, 8>(a, nz, st);
  return launch_conv_v2_pt<G, U8, 1>(a, nz, st);
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
  a.batch = batch; a.act = act; a.coef = u8_coef; a.ring_slot = nullptr; a.ring_cap = 0; a.stack_age = nullptr;
  a.slot_seq = nullptr; a.slot_entries = 0; a.slot_stride = 0; a.newest_frame = nullptr;
  hipStream_t st = dra_stream(stream);
  switch (layer) {
    case 1: return x_is_u8 ? launch_conv_v2<VG1, true, 2>(a, nz, st) : launch_conv_v2<VG1, false, 2>(a, nz, st);
    case 2: return x_is_u8 ? DRA_EINVAL : launch_conv_v2<VG2, false, 3>(a, nz, st);
    case 3: return x_is_u8 ? DRA_EINVAL : launch_conv_v2<VG3, false, 2>(a, nz, st);
  }
  return DRA_EINVAL;
}

// conv3's forward of the update with fc4's weights prefetched for the launch that follows (library-internal, actor_env.h):
// pf_w[z] = the [512][3136] weight matrix net z of dra_linear_fwd_slabs_one(nz = pf_nz, ksplit = 14) will read.
int dra_conv3_fwd_koc_pf(int nz, const void* const* x, const float* const* wt, const float* const* bias, float* const* y, int batch,
                         int act, const float* const* pf_w, int pf_nz, void* stream) {
  if (nz < 1 || nz > DRA_MAX_Z || batch < 1 || batch > 32 || !x || !wt || !bias || !y || !pf_w || pf_nz < 1 || pf_nz > DRA_MAX_Z)
    return DRA_EINVAL;
  ConvV2Args a;
  for (int z = 0; z < nz; ++z) {
    if (!x[z] || !wt[z] || !bias[z] || !y[z]) return DRA_EINVAL;
    a.x[z] = x[z]; a.wt[z] = wt[z]; a.bias[z] = bias[z]; a.y[z] = y[z];
  }
  for (int z = 0; z < pf_nz; ++z) {
    if (!pf_w[z]) return DRA_EINVAL;
    a.pf_w[z] = pf_w[z];
  }
  a.pf_nz = pf_nz;
  a.batch = batch; a.act = act; a.coef = 1.0; a.ring_slot = nullptr; a.ring_cap = 0; a.stack_age = nullptr;
  a.slot_seq = nullptr; a.slot_entries = 0; a.slot_stride = 0; a.newest_frame = nullptr;
  return launch_conv_v2<VG3, false, 2>(a, nz, dra_stream(stream));
}

// conv1 of the UPDATE straight from the replay ring (library-internal, actor_env.h): net z of sample b convolves the 4 ring
// frames ending at slot idx[b] + newest_off[z] (online(states): 0, target / online(next_states): n_step).
int dra_conv1_fwd_koc_ringbatch(const void* frames, const int64_t* idx, int64_t* idx_copy, const int64_t* idx_tagged,
                                const unsigned long long* update_seq, const int64_t* newest_off, int nz,
                                const float* const* wt, const float* const* bias, float* const* y, int batch, double u8_coef, int act,
                                void* stream) {
  if (!frames || !idx || !newest_off || nz < 1 || nz > DRA_MAX_Z || batch < 1 || !wt || !bias || !y) return DRA_EINVAL;
  if ((idx_tagged != nullptr) != (update_seq != nullptr)) return DRA_EINVAL;
  ConvV2Args a;
  for (int z = 0; z < nz; ++z) {
    if (!wt[z] || !bias[z] || !y[z]) return DRA_EINVAL;
    a.x[z] = frames; a.wt[z] = wt[z]; a.bias[z] = bias[z]; a.y[z] = y[z];
    a.idx_bias[z] = newest_off[z] - (VG1::C - 1);
  }
  a.sample_idx = idx;
  a.sample_idx_copy = idx_copy;
  a.sample_idx_tagged = idx_tagged;
  a.sample_seq = update_seq;
  a.batch = batch; a.act = act; a.coef = u8_coef; a.ring_slot = nullptr; a.ring_cap = 0; a.stack_age = nullptr;
  a.slot_seq = nullptr; a.slot_entries = 0; a.slot_stride = 0; a.newest_frame = nullptr;
  // (two 32-position tiles per workgroup -- 448 workgroups in one round instead of 832 in 1.3 -- measured 0.5 us slower,
  // profiles/r04m_ab_conv1_fwd_pt.jsonl)
  return launch_conv_v2_pt<VG1, true, 1>(a, nz, dra_stream(stream));
}

// conv1 of the actor's batch-1 forward reading its 4-frame stack straight from the replay ring
// (DQN_agent.py:24-33: the state the actor acts on IS the newest `history` frames of the ring):
// frames = slot-major u8 ring, *newest_slot_dev = slot of the newest frame (device int64).
DRA_API int dra_conv1_fwd_koc_ring(const void* frames, const int64_t* newest_slot_dev, const int32_t* stack_age_dev, int64_t capacity, const float* wt,
                                   const float* bias, float* y, double u8_coef, int act, void* stream) {
  if (!frames || !newest_slot_dev || capacity < VG1::C || !wt || !bias || !y) return DRA_EINVAL;
  ConvV2Args a;
  a.x[0] = frames; a.wt[0] = wt; a.bias[0] = bias; a.y[0] = y;
  a.batch = 1; a.act = act; a.coef = u8_coef; a.ring_slot = newest_slot_dev; a.ring_cap = capacity; a.stack_age = stack_age_dev;
  a.slot_seq = nullptr; a.slot_entries = 0; a.slot_stride = 0; a.newest_frame = nullptr;
  if (conv_b1_waves() == 8) return launch_conv_v2_pt<VG1, true, 1, 8>(a, 1, dra_stream(stream));
  return launch_conv_v2_pt<VG1, true, 1>(a, 1, dra_stream(stream));
}

// Same, with the slot read from entry (*seq_dev mod n_entries) of an array of parameter blocks: slot_field_dev points at
// the slot field of entry 0, entries are stride_bytes apart (dra_dqn_learner's actor parameter ring).
DRA_API int dra_conv1_fwd_koc_ring_seq(const void* frames, const int64_t* slot_field_dev, const int32_t* stack_age_field_dev, const unsigned* seq_dev,
                                       int n_entries, int64_t stride_bytes, int64_t capacity, const void* newest_frame,
                                       const float* wt, const float* bias, float* y, double u8_coef, int act,
                                       void* stream) {
  if (!frames || !slot_field_dev || !seq_dev || n_entries < 1 || stride_bytes < 8 || capacity < VG1::C || !wt || !bias || !y)
    return DRA_EINVAL;
  ConvV2Args a;
  a.x[0] = frames; a.wt[0] = wt; a.bias[0] = bias; a.y[0] = y;
  a.batch = 1; a.act = act; a.coef = u8_coef; a.ring_slot = slot_field_dev; a.ring_cap = capacity; a.stack_age = stack_age_field_dev;
  a.slot_seq = seq_dev; a.slot_entries = n_entries; a.slot_stride = stride_bytes;
  a.newest_frame = reinterpret_cast<const uint8_t*>(newest_frame);
  if (conv_b1_waves() == 8) return launch_conv_v2_pt<VG1, true, 1, 8>(a, 1, dra_stream(stream));
  return launch_conv_v2_pt<VG1, true, 1>(a, 1, dra_stream(stream));
}

// conv1 of the ring actor's env step with the previous step's head and the environment step fused in (ActorFuse in
// actor_env.h).  Internal to the library (called by learner.hip): not part of the C ABI.
template <int NW>
static int launch_conv1_actor_fused(const ConvV2Args& a, const ActorFuse& f, hipStream_t st) {
  using T = V2Tile<VG1, 1>;
  constexpr size_t img = (size_t)VG1::C * T::CS * sizeof(float);
  constexpr size_t red = (size_t)NW * 16 * 64 * sizeof(float);
  constexpr size_t bytes = img > red ? img : red;
  static_assert(bytes <= 64 * 1024, "default dynamic LDS limit");
  hipLaunchKernelGGL((conv1_actor_fused_kernel<VG1, NW>), dim3(T::TPG + 1, 1, 1), dim3(64 * NW), bytes, st, a, f);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

int dra_conv1_fwd_actor_fused(const void* frames, const int64_t* slot_field_dev, const int32_t* stack_age_field_dev, const unsigned* seq_dev, int n_entries,
                              int64_t stride_bytes, int64_t capacity, const void* newest_frame, const float* wt,
                              const float* bias, float* y, double u8_coef, int act, const ActorFuse* f, void* stream) {
  if (!frames || !slot_field_dev || !seq_dev || n_entries < 1 || stride_bytes < 8 || capacity < VG1::C || !wt || !bias || !y || !f)
    return DRA_EINVAL;
  if (f->mode != 1 && f->mode != 2) return DRA_EINVAL;
  if (f->mode == 2 && (f->e < 1 || f->e >= kMaxEnvSteps || f->n_actions < 1 || f->n_actions > 64 || !f->h4 || !f->wh || !f->bh))
    return DRA_EINVAL;
  ConvV2Args a;
  a.x[0] = frames; a.wt[0] = wt; a.bias[0] = bias; a.y[0] = y;
  a.batch = 1; a.act = act; a.coef = u8_coef; a.ring_slot = slot_field_dev; a.ring_cap = capacity; a.stack_age = stack_age_field_dev;
  a.slot_seq = seq_dev; a.slot_entries = n_entries; a.slot_stride = stride_bytes;
  a.newest_frame = f->mode == 1 ? reinterpret_cast<const uint8_t*>(newest_frame) : nullptr;
  if (conv_b1_waves() == 8) return launch_conv1_actor_fused<8>(a, *f, dra_stream(stream));
  return launch_conv1_actor_fused<4>(a, *f, dra_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// An A2C / PPO rollout step over NatureConvBody at 8-32 environments as FOUR launches instead of five (agents._PixelRollout;
// every launch of such a step is a few dozen workgroups at its latency floor, 5.6-6.6 us each: profiles/r05t_kernel_stats_*):
//   [conv1 of step t | policy head of step t-1]   the synthetic environments' observations do not depend on the actions (they
//       are generated for the whole rollout up front), so step t's conv1 has nothing to wait for in step t-1's head: the head's
//       ceil(B / 4) workgroups ride in front of conv1's 13 B (the DQN actor's conv1_actor_fused_kernel does the same): 5.7 us
//       against 5.6 + 6.0 + a boundary
//   conv2, conv3, fc4                              the plain launches
// Same device functions as the separate launches (conv_fwd_v2_body, rollout_roles.h): bit-identical results.
// (Measured and removed: [conv3 | fc4] as one launch -- fc4's 256 workgroups fetching their 6.4 MB of weights while conv3 runs,
// then waiting on an arrival counter -- took 13.9 us against 6.6 + 6.3 + a 0.7 us boundary: the write-through of conv3's planes,
// the arrival count, the poll and the cold read of the hand-over cost more than the boundary they replace; with an acquire
// fence per wave in front of the reads 27 us.  a2c_pixel 215 k against 224 k env-steps/s, ppo_pixel 115 k against 118 k:
// profiles/r05x_bench_agents_c3fc4_ab.jsonl.)
template <int NW>
__global__ void __launch_bounds__(64 * NW)
rollout_conv1_heads_kernel(const ConvV2Args a, const PolicyHeadArgs h, const int head_wgs) {
  __shared__ float s_out[4][68];
  __shared__ float s_phi[512];
  if ((int)blockIdx.x < head_wgs) {
    if (threadIdx.x >= 256) return;      // the head's workgroups are four waves (an exited wave does not hold a barrier up)
    if (h.slabs) {        // one row per workgroup, its features folded from fc4's 28 K-slice partial sums first
      policy_head_row_fold_wg<28>(h, (int)blockIdx.x, s_phi, s_out[0]);
      return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b < h.B) policy_head_row(h, b, lane, s_out[wave]);
    return;
  }
  ActorFuse none;
  none.mode = 0;
  conv_fwd_v2_body<VG1, true, 1, NW, false>(a, none, (int)blockIdx.x - head_wgs, 0, 0, false);
}

DRA_API int dra_rollout_conv1_heads(const void* frames_u8, const float* wt1, const float* b1, float* y1, int batch, double u8_coef,
                                    const float* phi_prev, const float* fold_bias, const float* w_a, const float* b_a,
                                    const float* w_v, const float* b_v, const float* uniform, int n_actions, int64_t* out_action,
                                    float* out_log_pi_a, float* out_entropy, float* out_v, void* stream) {
  return dra_rollout_conv1_heads_phi(frames_u8, wt1, b1, y1, batch, u8_coef, phi_prev, fold_bias, w_a, b_a, w_v, b_v, uniform, n_actions,
                                     out_action, out_log_pi_a, out_entropy, out_v, nullptr, stream);
}

// The same launch; out_phi != NULL (with fold_bias: phi_prev = fc4's 28 K-slice partial sums): the head's workgroups also leave the
// folded features relu(sum of slices + bias) [batch][512] of the PREVIOUS step there -- what A2C's update needs of fc4's forward
// when it backpropagates through the rollout's own activations (agents._PixelRollout, config.reuse_rollout_activations).
DRA_API int dra_rollout_conv1_heads_phi(const void* frames_u8, const float* wt1, const float* b1, float* y1, int batch, double u8_coef,
                                        const float* phi_prev, const float* fold_bias, const float* w_a, const float* b_a,
                                        const float* w_v, const float* b_v, const float* uniform, int n_actions, int64_t* out_action,
                                        float* out_log_pi_a, float* out_entropy, float* out_v, float* out_phi, void* stream) {
  if (out_phi && !(phi_prev && fold_bias)) return DRA_EINVAL;
  if (!frames_u8 || !wt1 || !b1 || !y1 || batch < 1 || batch > 4096) return DRA_EINVAL;
  if (phi_prev && (!w_a || !w_v || !uniform || !out_action || !out_log_pi_a || !out_entropy || !out_v || n_actions < 1 ||
                   n_actions > 64))
    return DRA_EINVAL;
  using T = V2Tile<VG1, 1>;
  ConvV2Args a;
  a.x[0] = frames_u8; a.wt[0] = wt1; a.bias[0] = b1; a.y[0] = y1;
  a.batch = batch; a.act = DRA_ACT_RELU; a.coef = u8_coef; a.ring_slot = nullptr; a.ring_cap = 0; a.stack_age = nullptr;
  a.slot_seq = nullptr; a.slot_entries = 0; a.slot_stride = 0; a.newest_frame = nullptr;
  PolicyHeadArgs h;
  memset(&h, 0, sizeof(h));
  int head_wgs = 0;
  if (phi_prev) {
    h.x = phi_prev; h.w0 = w_a; h.b0 = b_a; h.w1 = w_v; h.b1 = b_v; h.uniform = uniform; h.action_in = nullptr;
    h.out_action = out_action; h.out_lp = out_log_pi_a; h.out_ent = out_entropy; h.out_v = out_v; h.out_logits = nullptr;
    h.B = batch; h.K = 512; h.A = n_actions;
    h.slabs = nullptr; h.fold_bias = nullptr; h.out_x = nullptr;
    head_wgs = (batch + 3) / 4;
    if (fold_bias) {      // phi_prev is [28][batch][512]: the K-slice partial sums of dra_linear_fwd_slabs_one(ksplit = 28)
      h.x = nullptr; h.slabs = phi_prev; h.fold_bias = fold_bias; h.out_x = out_phi;
      head_wgs = batch;
    }
  }
  constexpr size_t img = (size_t)VG1::C * T::CS * sizeof(float);
  constexpr size_t red8 = (size_t)8 * 16 * 64 * sizeof(float), red4 = (size_t)4 * 16 * 64 * sizeof(float);
  constexpr size_t bytes8 = img > red8 ? img : red8, bytes4 = img > red4 ? img : red4;
  static_assert(bytes8 <= 64 * 1024, "conv1's latency shapes fit the default dynamic LDS limit");
  // the same wave count dra_conv_fwd_koc picks for this batch (launch_conv_v2): bit-identical with the plain launch
  if (batch <= kConvNw8MaxBatch && conv_b1_waves() == 8)
    hipLaunchKernelGGL(rollout_conv1_heads_kernel<8>, dim3(head_wgs + T::TPG * batch), dim3(512), bytes8, dra_stream(stream), a, h, head_wgs);
  else
    hipLaunchKernelGGL(rollout_conv1_heads_kernel<4>, dim3(head_wgs + T::TPG * batch), dim3(256), bytes4, dra_stream(stream), a, h, head_wgs);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
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

#ifdef DRA_TRACE
extern "C" int dra_trace_set_conv_v2(void* p) { return dra_trace_set_local(p); }
#endif
