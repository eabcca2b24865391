// Horizontally fused backward launches and the one-pass contractions of the DQN update (oneshot.h).
// Replaces the autograd backward of deep_rl/network/network_bodies.py:27-33 + network_heads.py:18-21
// as driven by DQN_agent.py:129 (loss.backward()) for VanillaNet(NatureConvBody).
//
// A layer's weight gradient and input gradient are independent given the incoming gradient, so they
// share ONE launch (multi_kernel): one dependent-launch cost instead of two or three, and the two
// half-empty grids fill the chip together.  `variant` selects the kernels behind each role:
//   DRA_VAR_ONESHOT_DGRAD  one-pass input gradients (ConvDgradLin / LinDgradOne) instead of the
//                          K-chunked implicit GEMM;
//   DRA_VAR_ONESHOT_WGRAD  one-pass conv weight gradients (ConvWgradOne / ConvWgradLin): one slab per (sample, row
//                          chunk) -- dra_conv_wgrad_slabs() slabs instead of `ksplit`.
#include "oneshot_lin.h"
#include "dgrad_scatter.h"
#include "actor_env.h"
#include "per_chain2.h"
#include <stdlib.h>
#include <type_traits>

static int g_tuning = 511 | 4096 | 8192 | 16384 | 32768 | 131072 | 524288 | 1048576 | 8388608 | 16777216 | 33554432 | 67108864 | 134217728 | 268435456;
// every bit up to DRA_VAR_CU_PARTITION plus ACTOR_RING, ACTOR_FUSED_CONV1, GATHER_ON_UPDATE and RING_DIRECT measured faster
// on MI355X in same-box A/Bs (profiles/r01b_ab_variants.jsonl, r01d_*, r01f_*, r02y_ab_*, r02zf_ab_*); ACTOR_V3 (512),
// ACTOR_FUSED_HEAD (1024) and GATHER_IN_GRAPH (2048) measured neutral or slower and stay opt-in; IDX_PREFETCH (131072): conv1_fwd
// 11.6 -> 10.3 us, +0.6 % (profiles/r02zu_*); COOP_OPT (65536) measured 14 % SLOWER -- a grid barrier of 796 workgroups on one
// counter costs 24 us on this part (profiles/r02zt_*) -- and stays opt-in as a kept negative result.
// Round 3 (same-box A/Bs, profiles/r03*_ab*.jsonl): LATE_FOLD (524288: no gradient-norm launch) and ACTOR_MEGA (1048576: conv3 + fc4
// of the actor's env step as one launch) together +0.5 %, 14 -> 12 launches on the update + actor chains per env step pair;
// WGRAD_ACC (262144: four samples accumulated per workgroup, a quarter of the slabs) was neutral on conv3 and slower on conv1 /
// conv2; against round 4's linear-map roles it lost on every layer and was removed (the bit is accepted and ignored).

DRA_API int dra_set_tuning(int mask) {
  if (mask < 0) return DRA_EINVAL;
  g_tuning = mask;
  return DRA_OK;
}
DRA_API int dra_get_tuning(int* mask) {
  if (!mask) return DRA_EINVAL;
  static bool env_read = false;
  if (!env_read) {     // DRA_TUNING=<mask>: process default for A/B runs of whole agents (tools/gpu_ab_agents.sh)
    env_read = true;
    const char* e = getenv("DRA_TUNING");
    if (e && atoi(e) >= 0) g_tuning = atoi(e);
  }
  *mask = g_tuning;
  return DRA_OK;
}

// conv1: 4 output rows per chunk, all 8 k tiles per workgroup (uint8 frames, also straight from the replay ring)
using WG1u = ConvWgradOne<G1, 4, 4, 88, 0, true>;
using WG1f = ConvWgradOne<G1, 4, 4, 88, 0, false>;
// rollout batch sizes, persistent over the batch (profiles/r05m_conv_big_roles.jsonl): below 512 samples 10 output rows per chunk,
// 2 k groups x 256 shares (0.41 of peak at 256); from 512 on all 8 k tiles per workgroup, 5 rows per chunk -- the gradient chunk
// is staged once for all of K -- x 512 shares (0.54 at 1024)
using WG1up = ConvWgradOne<G1, 10, 4, 88, 0, true, 256>;
using WG1fp = ConvWgradOne<G1, 10, 4, 88, 0, false, 256>;
using WG1uq = ConvWgradOne<G1, 5, 8, 88, 0, true, 512>;
using WG1fq = ConvWgradOne<G1, 5, 8, 88, 0, false, 512>;
constexpr int kPersistConv1Batch = 128;
// conv2 / conv3 (round 4, oneshot_lin.h): one workgroup = one sample x a group of k tiles, operands staged into LDS as they
// lie in memory.  (The round-2 forms with transposing staging -- ConvWgradOne<G2 / G3>, ConvDgradOne -- and round 3's
// ConvWgradAcc, which accumulated four samples per workgroup to write a quarter of the slabs, were removed in round 4: same
// box, 8633 -> 8922 updates/s for the linear maps, and every layer slower with the accumulating form than without
// (conv1 -4 %, conv1 + conv2 -11 %; profiles/r04e_ab_env.jsonl, r04g_ab_env.jsonl).  DRA_VAR_WGRAD_ACC is accepted and ignored.)
using WG2l = ConvWgradLin<G2, 4>;     // 128 workgroups x 82 MFMAs per wave (2 k-tiles per workgroup: 256 x 41 measured 0.8 us slower)
using WG3l = ConvWgradLin<G3, 2>;     // 288 x 25 (3 k-tiles: 192 x 50 / 25, 0.8 us slower); profiles/r04j_ab_wgrad_tiles_role_order.jsonl

// rollout batch sizes (A2C 80 stays below; PPO minibatches 256 ...): the persistent forms, one slab per batch share
using WG2p = ConvWgradPers<G2, 8, 128, 1>;   // 2 k groups x 128 shares = 256 workgroups; 164 MFMAs per wave and sample
using WG3p = ConvWgradPers<G3, 6, 85, 2>;    // 3 k groups x 85 shares = 255 workgroups (one per CU: 288 left 32 CUs with two); 2 x 75 MFMAs per wave and iteration
// from which batch on: conv3 from 128 (its persistent role shares the launch with the input gradient); conv2 from 768 (two launches:
// at 256 / 512 the one-sample form, one launch, is as fast or faster: 46 vs 49 us, 86 vs 89 us)
// DRA_VAR_DGRAD_SCATTER: samples per workgroup of the scatter-form input gradient (conv3: 2 x 49 positions = 7 tiles of 16, a
// 41.5 KB image; conv2: 81 positions = 6 tiles, 51.3 KB)
constexpr int kScatterFromBatch = 256;      // (at 128 samples the gather form's 512 / 1024 small workgroups win: 22.6 vs 24.5 / 25.8 vs 31.3 us, profiles/r06zzk)
template <class G> struct ScatSamples { static constexpr int NS = 1; };
template <> struct ScatSamples<G3> { static constexpr int NS = 2; };
template <class G> struct PersistFrom { static constexpr int batch = 768; static constexpr bool one_launch = false; };
template <> struct PersistFrom<G3> { static constexpr int batch = 128; static constexpr bool one_launch = true; };

DRA_API int dra_conv_wgrad_slabs(int layer, int batch, int ksplit, int variant, int* n_slabs) {
  if (!n_slabs || batch < 1 || ksplit < 1 || layer < 1 || layer > 3) return DRA_EINVAL;
  if (!(variant & DRA_VAR_ONESHOT_WGRAD)) { *n_slabs = ksplit; return DRA_OK; }
  const bool pers2 = batch >= PersistFrom<G2>::batch && (variant & DRA_VAR_ONESHOT_DGRAD);
  const bool pers3 = batch >= PersistFrom<G3>::batch && (variant & DRA_VAR_ONESHOT_DGRAD);
  switch (layer) {
    case 1: *n_slabs = batch >= 512 ? WG1uq::n_slabs(batch) : (batch >= kPersistConv1Batch ? WG1up::n_slabs(batch) : WG1u::n_slabs(batch)); return DRA_OK;
    case 2: *n_slabs = pers2 ? WG2p::n_slabs(batch) : WG2l::n_slabs(batch); return DRA_OK;
    case 3: *n_slabs = pers3 ? WG3p::n_slabs(batch) : WG3l::n_slabs(batch); return DRA_OK;
  }
  return DRA_EINVAL;
}

template <class W>
static W make_wgrad_one(const float* dy, const void* x, float* dw, float* db, int64_t slab_stride, int batch, double coef) {
  W r;
  r.dy = dy; r.x = x; r.dw = dw; r.db = db; r.slab_stride = slab_stride; r.B = batch; r.coef = coef;
  r.xcd = dra_xcd_order_enabled();
  return r;
}

// tiles per workgroup of the one-pass input gradient: conv2 (4 stride phases x 4 tiles per sample) pairs tiles -- 256 instead of
// 512 workgroups, one round together with the weight-gradient role (1 and 4 tiles per workgroup measured slower in round 2)
template <class G> struct DgradTiles { static constexpr int PT = (G::S == 2) ? 2 : 1; };
template <class G, int PT = DgradTiles<G>::PT>
static ConvDgradLin<G, PT> make_dgrad_one(const float* dy, const float* wt, const float* xact, float* dx, int batch, int act) {
  ConvDgradLin<G, PT> r;
  r.dy = dy; r.wt = wt; r.xact = xact; r.dx = dx; r.B = batch; r.act = act;
  r.xcd = dra_xcd_order_enabled();
  return r;
}

template <class G, bool U8>
static IgemmRole<ConvWgradKoc<G, 32, 32, 64, U8>> make_wgrad_igemm(const float* dy, const void* x, float* dw, float* db,
                                                                 int64_t slab_stride, int ksplit, int batch, double coef) {
  ConvWgradKoc<G, 32, 32, 64, U8> p;
  p.M = G::K + 1; p.N = G::OC; p.K = batch * G::P;
  p.dy = dy; p.x = x; p.dw = dw; p.db = db; p.slab_stride = slab_stride; p.coef = coef;
  return make_igemm_role(p, ksplit);
}

template <class G>
static IgemmRole<ConvDgradKoc<G, 32, 32, 64>> make_dgrad_igemm(const float* dy, const float* wt, const float* xact,
                                                               float* dx, int batch, int act) {
  using PT = ConvDgradKoc<G, 32, 32, 64>;
  PT p;
  p.M = G::C; p.N = batch * PT::PP; p.K = G::OC * PT::KPP;
  p.dy = dy; p.wt = wt; p.xact = xact; p.dx = dx; p.act = act;
  IgemmRole<PT> r = make_igemm_role(p, 1);
  return r;
}

template <class R>
static int igemm_blocks(const R& r, int nz) { return r.tiles * r.ksplit * nz; }

// compute units of the current device (asked once per process: one GPU per process; MI355X's 256 assumed if the question fails)
static int scatter_cu_count() {
  static int n_cu = 0;
  if (n_cu <= 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    n_cu = n;
  }
  return n_cu;
}

// DRA_VAR_DGRAD_SCATTER: weight gradient (persistent form from PersistFrom<G>::batch on, else one sample per workgroup) + the
// scatter-form input gradient with NS samples per workgroup; one launch with DRA_VAR_FUSED_BWD, else two
template <class G, class WOne, class WPers, int NS>
static int conv_bwd_scatter_t(const float* dy, const void* x, const float* wt, const float* xact, float* dw, float* db,
                              int64_t slab_stride, float* dx, int batch, int act, int variant, hipStream_t st) {
  NoRole none;
  using RS = ConvDgradScat<G, NS>;
  RS rs;
  rs.dy = dy; rs.wt = wt; rs.xact = xact; rs.dx = dx; rs.B = batch; rs.act = act;
  ConvDgradScat<G, NS, true> alone;      // a launch to itself: two workgroups per CU (its register footprint), each looping over its groups with the weights kept
  alone.dy = dy; alone.wt = wt; alone.xact = xact; alone.dx = dx; alone.B = batch; alone.act = act;
  alone.cap = 2 * scatter_cu_count();
  const bool only_d = variant & DRA_VAR_MEASURE_DGRAD_ONLY, only_w = variant & DRA_VAR_MEASURE_WGRAD_ONLY;
  if (batch >= PersistFrom<G>::batch) {
    WPers rp;
    rp.dy = dy; rp.x = x; rp.dw = dw; rp.db = db; rp.slab_stride = slab_stride; rp.B = batch;
    if (only_d) return launch_multi(alone, alone.blocks(), none, 0, none, 0, st);
    if (only_w) return launch_multi(rp, rp.blocks(), none, 0, none, 0, st);
    // (conv2: the two roles together need more than 256 registers -- one wave per SIMD -- so they stay two launches)
    if ((variant & DRA_VAR_FUSED_BWD) && PersistFrom<G>::one_launch) return launch_multi(rp, rp.blocks(), rs, rs.blocks(), none, 0, st);
    if (int rc = launch_multi(rp, rp.blocks(), none, 0, none, 0, st)) return rc;
    return launch_multi(alone, alone.blocks(), none, 0, none, 0, st);
  }
  auto rw = make_wgrad_one<WOne>(dy, x, dw, db, slab_stride, batch, 1.0);
  if (only_d) return launch_multi(alone, alone.blocks(), none, 0, none, 0, st);
  if (only_w) return launch_multi_tp(rw, rw.blocks(), none, 0, none, 0, st);
#if defined(DRA_EXP_SCAT_WGRAD_FIRST) && DRA_EXP_SCAT_WGRAD_FIRST
  if (variant & DRA_VAR_FUSED_BWD) return launch_multi(rw, rw.blocks(), rs, rs.blocks(), none, 0, st);     // (A/B build: weight gradient first)
#endif
  if (variant & DRA_VAR_FUSED_BWD) return launch_multi(rs, rs.blocks(), rw, rw.blocks(), none, 0, st);
  if (int rc = launch_multi_tp(rw, rw.blocks(), none, 0, none, 0, st)) return rc;
  return launch_multi(alone, alone.blocks(), none, 0, none, 0, st);
}

// layers 2 / 3: weight gradient + input gradient in one launch
template <class G, class WOne, class WPers = WOne, class R3 = NoRole>
static int conv_bwd_fused_t(const float* dy, const void* x, const float* wt, const float* xact, float* dw, float* db,
                            int64_t slab_stride, int ksplit, float* dx, int batch, int act, int variant, hipStream_t st,
                            const R3& none = R3(), int n3 = 0) {
  const bool ow = variant & DRA_VAR_ONESHOT_WGRAD, od = variant & DRA_VAR_ONESHOT_DGRAD;
  if (n3 > 0 && !(od && ow)) return DRA_EINVAL;   // a riding role exists for the one-pass pair only
  if (od && ow) {
    if constexpr (!std::is_same<WPers, WOne>::value) {
      // DRA_VAR_DGRAD_SCATTER (round 6, dgrad_scatter.h): the input gradient contracted over the output positions
      if ((variant & DRA_VAR_DGRAD_SCATTER) && batch >= kScatterFromBatch && n3 == 0) {
        // (conv3 below 512 samples: one sample per workgroup -- 64 / 49 padding, but twice the workgroups)
        if constexpr (ScatSamples<G>::NS > 1) {
          if (batch < 512) return conv_bwd_scatter_t<G, WOne, WPers, 1>(dy, x, wt, xact, dw, db, slab_stride, dx, batch, act, variant, st);
        }
        return conv_bwd_scatter_t<G, WOne, WPers, ScatSamples<G>::NS>(dy, x, wt, xact, dw, db, slab_stride, dx, batch, act, variant, st);
      }
      if (batch >= PersistFrom<G>::batch && n3 == 0) {
        WPers rp;
        rp.dy = dy; rp.x = x; rp.dw = dw; rp.db = db; rp.slab_stride = slab_stride; rp.B = batch;
        // (input gradient at these sizes: all position tiles of a sample's phase in one workgroup -- the gradient image staged
        // once instead of once per tile)
        const bool only_d = variant & DRA_VAR_MEASURE_DGRAD_ONLY, only_w = variant & DRA_VAR_MEASURE_WGRAD_ONLY;
        if (G::S == 1 && batch < 512) {      // conv3 below 512: one position tile per workgroup keeps the grid large enough
          auto rd1 = make_dgrad_one<G>(dy, wt, xact, dx, batch, act);
          if (only_d || only_w) {
            int rc = DRA_OK;
            if (!only_d) rc = launch_multi(rp, rp.blocks(), none, 0, none, 0, st);
            if (rc == DRA_OK && !only_w) rc = launch_multi_tp(rd1, rd1.blocks(), none, 0, none, 0, st);
            return rc;
          }
          return launch_multi_tp(rp, rp.blocks(), rd1, rd1.blocks(), none, 0, st);
        }
        auto rd = make_dgrad_one<G, (G::S == 2) ? 2 : 3>(dy, wt, xact, dx, batch, act);
        if (!PersistFrom<G>::one_launch || only_d || only_w) {
          // two launches: the persistent role's LDS image would cost the input-gradient workgroups their occupancy (conv2)
          int rc = DRA_OK;
          if (!only_d) rc = launch_multi(rp, rp.blocks(), none, 0, none, 0, st);
          if (rc == DRA_OK && !only_w) rc = launch_multi_tp(rd, rd.blocks(), none, 0, none, 0, st);
          return rc;
        }
        // one launch, the persistent weight-gradient workgroups (its longest) FIRST in the grid (conv3)
        return launch_multi_tp(rp, rp.blocks(), rd, rd.blocks(), none, 0, st);
      }
    }
    auto rw = make_wgrad_one<WOne>(dy, x, dw, db, slab_stride, batch, 1.0);
    auto rd = make_dgrad_one<G>(dy, wt, xact, dx, batch, act);
    if (variant & (DRA_VAR_MEASURE_DGRAD_ONLY | DRA_VAR_MEASURE_WGRAD_ONLY))
      return launch_multi(rd, (variant & DRA_VAR_MEASURE_WGRAD_ONLY) ? 0 : rd.blocks(), rw,
                          (variant & DRA_VAR_MEASURE_DGRAD_ONLY) ? 0 : rw.blocks(), none, n3, st);
    // (the weight-gradient workgroups, the longer ones, FIRST in the launch: no difference, profiles/r04j_ab_wgrad_tiles_role_order.jsonl)
    if (batch >= 128 && n3 == 0) return launch_multi_tp(rd, rd.blocks(), rw, rw.blocks(), none, 0, st);     // three waves per SIMD
    return launch_multi(rd, rd.blocks(), rw, rw.blocks(), none, n3, st);
  }
  if constexpr (!std::is_same<R3, NoRole>::value) return DRA_EINVAL;
  else {
  if (od) {
    auto rd = make_dgrad_one<G>(dy, wt, xact, dx, batch, act);
    auto rw = make_wgrad_igemm<G, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0);
    return launch_multi(rd, rd.blocks(), rw, igemm_blocks(rw, 1), none, 0, st);
  }
  auto rd = make_dgrad_igemm<G>(dy, wt, xact, dx, batch, act);
  const int nd = igemm_blocks(rd, G::S * G::S);
  if (ow) {
    auto rw = make_wgrad_one<WOne>(dy, x, dw, db, slab_stride, batch, 1.0);
    return launch_multi(rd, nd, rw, rw.blocks(), none, 0, st);
  }
  auto rw = make_wgrad_igemm<G, false>(dy, x, dw, db, slab_stride, ksplit, batch, 1.0);
  return launch_multi(rd, nd, rw, igemm_blocks(rw, 1), none, 0, st);
  }
}

// One launch for a conv layer's backward (KOC weights): dWt / db slabs [n_slabs][..] and, for layers 2 and 3,
// dx = gradient w.r.t. the pre-activation of the layer below (xact = that layer's output).  layer 1 has no
// input gradient (dx / wt / xact ignored).  n_slabs = dra_conv_wgrad_slabs(layer, batch, ksplit, variant).
DRA_API int dra_conv_bwd_fused(int layer, const float* dy, const void* x, const float* wt, const float* xact, float* dw,
                               float* db, int64_t slab_stride, int ksplit, float* dx, int batch, int x_is_u8,
                               double u8_coef, int act, int variant, void* stream) {
  if (!dy || !x || !dw || !db || batch < 1 || ksplit < 1) return DRA_EINVAL;
  if (layer != 1 && (!wt || !dx || x_is_u8)) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  NoRole none;
  switch (layer) {
    case 1:
      if (variant & DRA_VAR_ONESHOT_WGRAD) {
        if (batch >= 512) {
          if (x_is_u8) {
            auto rw = make_wgrad_one<WG1uq>(dy, x, dw, db, slab_stride, batch, u8_coef);
            return launch_multi_tp(rw, rw.blocks(), none, 0, none, 0, st);
          }
          auto rw = make_wgrad_one<WG1fq>(dy, x, dw, db, slab_stride, batch, 1.0);
          return launch_multi_tp(rw, rw.blocks(), none, 0, none, 0, st);
        }
        if (batch >= kPersistConv1Batch) {
          if (x_is_u8) {
            auto rw = make_wgrad_one<WG1up>(dy, x, dw, db, slab_stride, batch, u8_coef);
            return launch_multi_tp(rw, rw.blocks(), none, 0, none, 0, st);
          }
          auto rw = make_wgrad_one<WG1fp>(dy, x, dw, db, slab_stride, batch, 1.0);
          return launch_multi_tp(rw, rw.blocks(), none, 0, none, 0, st);
        }
        if (x_is_u8) {
          auto rw = make_wgrad_one<WG1u>(dy, x, dw, db, slab_stride, batch, u8_coef);
          return launch_multi(rw, rw.blocks(), none, 0, none, 0, st);
        }
        auto rw = make_wgrad_one<WG1f>(dy, x, dw, db, slab_stride, batch, 1.0);
        return launch_multi(rw, rw.blocks(), none, 0, none, 0, st);
      }
      return dra_conv_bwd_w_koc(1, dy, x, dw, db, slab_stride, ksplit, batch, x_is_u8, u8_coef, stream);
    case 2:
      return conv_bwd_fused_t<G2, WG2l, WG2p>(dy, x, wt, xact, dw, db, slab_stride, ksplit, dx, batch, act, variant, st);
    case 3:
      return conv_bwd_fused_t<G3, WG3l, WG3p>(dy, x, wt, xact, dw, db, slab_stride, ksplit, dx, batch, act, variant, st);
  }
  return DRA_EINVAL;
}

// conv3's backward with the device-side prioritized draw riding along (library-internal, per_chain2.h / actor_env.h): the
// draw is one workgroup's 16 us latency chain that needs the update's loss vector only; as a third role of this launch
// (10-11 us of MFMA work on every other CU) it disappears from the update's critical path.  Minibatches up to 256
// (the role has the launch's 256 threads), one-pass kernels only.
// PART 0: the whole draw here.  PART 1: its first half (priorities -> tree, adds); the second half (descent, filter, hand-over)
// then rides in conv1's weight-gradient launch (dra_conv1_wgrad_fold): each half is shorter than the launch carrying it.
template <int PART>
struct ChainRole {
  static constexpr int LDS_FLOATS = (per_chain2_lds_bytes<256>() + 3) / 4;
  PerChain2Args a;
  __device__ __forceinline__ void run(int, float* lds, int = 0) const { per_chain2_body<256, PART>(a, reinterpret_cast<char*>(lds)); }
};
template <int PART>
static int conv3_bwd_chain_t(const float* dy, const void* x, const float* wt, const float* xact, float* dw, float* db,
                             int64_t slab_stride, float* dx, int batch, int act, int variant, const PerChain2Args* chain, hipStream_t st) {
  ChainRole<PART> r;
  r.a = *chain;
  return conv_bwd_fused_t<G3, WG3l, WG3l, ChainRole<PART>>(dy, x, wt, xact, dw, db, slab_stride, 1, dx, batch, act, variant, st, r, 1);
}
int dra_conv3_bwd_fused_chain(const float* dy, const void* x, const float* wt, const float* xact, float* dw, float* db,
                              int64_t slab_stride, float* dx, int batch, int act, int variant, const PerChain2Args* chain,
                              int first_half_only, void* stream) {
  if (!dy || !x || !wt || !dx || !dw || !db || batch < 1 || !chain || chain->nb > 256) return DRA_EINVAL;
  if (!(variant & DRA_VAR_ONESHOT_WGRAD) || !(variant & DRA_VAR_ONESHOT_DGRAD)) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  if (first_half_only) return conv3_bwd_chain_t<1>(dy, x, wt, xact, dw, db, slab_stride, dx, batch, act, variant, chain, st);
  return conv3_bwd_chain_t<0>(dy, x, wt, xact, dw, db, slab_stride, dx, batch, act, variant, chain, st);
}

// conv1's weight gradient with the uint8 minibatch read straight from the replay ring (library-internal, actor_env.h):
// sample b = the 4 frames ending at ring slot idx[b].  One-pass kernel only (the learner's variant).
int dra_conv1_wgrad_ringbatch(const float* dy, const void* frames, const int64_t* idx, float* dw_slabs, float* db_slabs,
                              int64_t slab_stride, int batch, double u8_coef, int variant, void* stream) {
  if (!dy || !frames || !idx || !dw_slabs || !db_slabs || batch < 1 || !(variant & DRA_VAR_ONESHOT_WGRAD)) return DRA_EINVAL;
  NoRole none;
  auto rw = make_wgrad_one<WG1u>(dy, frames, dw_slabs, db_slabs, slab_stride, batch, u8_coef);
  rw.sample_idx = idx;
  return launch_multi(rw, rw.blocks(), none, 0, none, 0, dra_stream(stream));
}

// DRA_VAR_LATE_FOLD (library-internal, actor_env.h): the same launches with a FoldRole riding along -- `fold` describes the
// segment of the flat gradient `grad` whose slabs the PREVIOUS backward launch wrote; its workgroups' sums of squares go to
// fold_partials[0, *n_fold_partials); reset_slots[0, n_reset) (optional) are set to -1.0 by the fold's first workgroup (the
// arrival slots of the late-fold optimizer launch, dra_clip_step_late).  At most 32 slabs.
static FoldRole make_fold_role(const dra_fold_seg* fold, float* grad, double* partials, double* reset_slots, int n_reset) {
  FoldRole f;
  f.grad = grad; f.slabs = fold->slabs; f.begin4 = fold->begin >> 2; f.count4 = fold->count >> 2;
  f.stride4 = fold->slab_stride >> 2; f.n_slabs = fold->n_slabs; f.partials = partials;
  f.reset_slots = reset_slots; f.n_reset = reset_slots ? n_reset : 0;
  return f;
}
static bool fold_ok(const dra_fold_seg* fold, const float* grad, const double* partials) {
  return fold && grad && partials && fold->slabs && fold->n_slabs >= 1 && fold->n_slabs <= FoldRole::NG * FoldRole::SPT &&
         fold->count >= 4 && !(fold->begin & 3) &&
         !(fold->count & 3) && !(fold->slab_stride & 3) && !((((uintptr_t)fold->slabs) | ((uintptr_t)grad)) & 15);
}

int dra_conv_bwd_fused_fold(int layer, const float* dy, const void* x, const float* wt, const float* xact, float* dw, float* db,
                            int64_t slab_stride, float* dx, int batch, int act, int variant, const dra_fold_seg* fold,
                            float* grad, double* fold_partials, int* n_fold_partials, double* reset_slots, int n_reset,
                            void* stream) {
  if (!dy || !x || !wt || !dx || !dw || !db || batch < 1 || !n_fold_partials || !fold_ok(fold, grad, fold_partials)) return DRA_EINVAL;
  if (!(variant & DRA_VAR_ONESHOT_WGRAD) || !(variant & DRA_VAR_ONESHOT_DGRAD)) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  const FoldRole f = make_fold_role(fold, grad, fold_partials, reset_slots, n_reset);
  *n_fold_partials = f.blocks();
  if (layer == 2)
    return conv_bwd_fused_t<G2, WG2l, WG2l, FoldRole>(dy, x, wt, xact, dw, db, slab_stride, 1, dx, batch, act, variant, st, f, f.blocks());
  if (layer == 3)
    return conv_bwd_fused_t<G3, WG3l, WG3l, FoldRole>(dy, x, wt, xact, dw, db, slab_stride, 1, dx, batch, act, variant, st, f, f.blocks());
  return DRA_EINVAL;
}

// conv1's weight gradient (uint8 input: a plain [B][4][84][84] batch, or with idx != null the replay ring's frame array read
// through the sampled slots) with a FoldRole riding along
int dra_conv1_wgrad_fold(const float* dy, const void* x, const int64_t* idx, float* dw_slabs, float* db_slabs, int64_t slab_stride,
                         int batch, double u8_coef, int variant, const dra_fold_seg* fold, float* grad, double* fold_partials,
                         int* n_fold_partials, double* reset_slots, int n_reset, const PerChain2Args* chain, void* stream) {
  if (!dy || !x || !dw_slabs || !db_slabs || batch < 1 || !(variant & DRA_VAR_ONESHOT_WGRAD) || !n_fold_partials ||
      !fold_ok(fold, grad, fold_partials))
    return DRA_EINVAL;
  const FoldRole f = make_fold_role(fold, grad, fold_partials, reset_slots, n_reset);
  *n_fold_partials = f.blocks();
  if (chain) {     // the second half of the device-side prioritized draw as third role (see ChainRole)
    if (chain->nb > 256) return DRA_EINVAL;
    ChainRole<2> cr;
    cr.a = *chain;
    auto rw = make_wgrad_one<WG1u>(dy, x, dw_slabs, db_slabs, slab_stride, batch, u8_coef);
    rw.sample_idx = idx;
    return launch_multi(rw, rw.blocks(), f, f.blocks(), cr, 1, dra_stream(stream));
  }
  NoRole none;
  auto rw = make_wgrad_one<WG1u>(dy, x, dw_slabs, db_slabs, slab_stride, batch, u8_coef);
  rw.sample_idx = idx;
  return launch_multi(rw, rw.blocks(), f, f.blocks(), none, 0, dra_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// DRA_VAR_BWD_CHAIN (round 6): the three conv backward launches of the DQN update (batch <= 32, one-pass roles, late fold) as ONE
// launch in dependency order:
//   [conv3 dgrad | conv3 wgrad] [conv2 dgrad | conv2 wgrad] [conv1 wgrad] [fold of conv3's slabs] [fold of conv2's slabs]
// A workgroup of layer L - 1 waits (its weights / forward activations requested first) for the input-gradient workgroups of ITS
// sample in layer L; the folds wait for every weight-gradient workgroup of their layer (one counter, polled slowly: they are the
// last in dispatch order and off the path).  Producers store agent-scope, complete their stores and count themselves; consumers
// read with agent-scope loads.  Counters are never reset (targets = chains completed x arrivals per chain; the update's head
// kernel has bumped the count when this launch runs).  Same arithmetic as the three launches: bit-identical gradients
// (tests/test_gpu_agents.py::test_backward_chain_is_bit_identical).  DQN_agent.py:129-134's loss.backward() through
// network_bodies.py:10-33.
// (fc4's weight gradient of the learner's launch: the register-only role for minibatches up to 32 -- see dra_fc_bwd_fused_sq)
static bool fc_wgrad_one(int batch) { return batch <= 32; }
constexpr int kFcWgradNI = 8;    // 32-wide input tiles per workgroup of LinWgradOne
struct BwdChainN { int fd, fw, fh, d3, w3, d2, w2, w1, f3, f2; };
using BD3 = ConvDgradLin<G3, 1>;
using BD2 = ConvDgradLin<G2, 2>;
using FCD = LinDgradOne<512>;
using FCW = LinWgradOne<8>;
// FC (DRA_VAR_BWD_CHAIN_FC): fc4's input gradient, fc4's weight gradient and the head's weight gradient lead the launch; conv3's roles
// then take dy3 from the input-gradient workgroups of the same launch (one arrival counter: a column tile of dy3 covers every sample)
template <bool FC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
bwd_chain_kernel(const FCD fd, const FCW fw, const HeadWgradRole fh, const BD3 d3, const WG3l w3, const BD2 d2, const WG2l w2,
                 const WG1u w1, const FoldRole f3, const FoldRole f2, const BwdChainN n) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  int b = blockIdx.x, first = 0;
  if constexpr (FC) {
    if (b < n.fd) { fd.run_<true>(b, dyn_lds); return; }
    b -= n.fd; first += n.fd;
    if (b < n.fw) { fw.run(b, dyn_lds, first); return; }
    b -= n.fw; first += n.fw;
    if (b < n.fh) { fh.run(b, dyn_lds, first); return; }
    b -= n.fh; first += n.fh;
  }
  if (b < n.d3) { d3.run_<FC, true>(b, dyn_lds, first); return; }
  b -= n.d3; first += n.d3;
  if (b < n.w3) { w3.run_<FC, true>(b, dyn_lds, first); return; }
  b -= n.w3; first += n.w3;
  if (b < n.d2) { d2.run_<true, true>(b, dyn_lds, first); return; }
  b -= n.d2; first += n.d2;
  if (b < n.w2) { w2.run_<true, true>(b, dyn_lds, first); return; }
  b -= n.w2; first += n.w2;
  if (b < n.w1) { w1.run_<true>(b, dyn_lds, first); return; }
  b -= n.w1;
  if (b < n.f3) { f3.run_<true>(b, dyn_lds); return; }
  b -= n.f3;
  f2.run_<true>(b, dyn_lds);
}

constexpr int kBwdChainCounters = (2 * 32 + 3) * kChainLine;
int dra_bwd_chain_counters(void) { return kBwdChainCounters; }

// Library-internal (actor_env.h).  counters: kBwdChainCounters zeroed unsigned (never reset); *epoch = updates whose head kernel
// has run (so: chains launched, this one included).  The fold arguments are those of dra_conv_bwd_fused_fold (layer 2: conv3's
// segment) and dra_conv1_wgrad_fold (conv2's segment + the optimizer's arrival slots).
int dra_conv_bwd_chain(const float* dy3, const float* y2, const float* wt3, float* dw3, float* db3, int64_t stride3, float* dy2,
                       const float* y1, const float* wt2, float* dw2, float* db2, int64_t stride2, float* dy1, const void* frames,
                       const int64_t* idx, float* dw1, float* db1, int64_t stride1, int batch, double u8_coef, int act,
                       const dra_fold_seg* fold3, const dra_fold_seg* fold2, float* grad, double* partials3, int* n_partials3,
                       double* partials2, int* n_partials2, double* reset_slots, int n_reset, unsigned* counters,
                       const unsigned* epoch, int* timeout_flag, const DraBwdChainFc* fc, void* stream) {
  if (!dy3 || !y2 || !wt3 || !dw3 || !db3 || !dy2 || !y1 || !wt2 || !dw2 || !db2 || !dy1 || !frames || !dw1 || !db1 || batch < 1 ||
      batch > 32 || !n_partials3 || !n_partials2 || !fold_ok(fold3, grad, partials3) || !fold_ok(fold2, grad, partials2) || !counters ||
      !epoch || !timeout_flag)
    return DRA_EINVAL;
  BD3 d3 = make_dgrad_one<G3, 1>(dy3, wt3, y2, dy2, batch, act);
  WG3l w3 = make_wgrad_one<WG3l>(dy3, y2, dw3, db3, stride3, batch, 1.0);
  BD2 d2 = make_dgrad_one<G2, 2>(dy2, wt2, y1, dy1, batch, act);
  WG2l w2 = make_wgrad_one<WG2l>(dy2, y1, dw2, db2, stride2, batch, 1.0);
  WG1u w1 = make_wgrad_one<WG1u>(dy1, frames, dw1, db1, stride1, batch, u8_coef);
  w1.sample_idx = idx;
  FoldRole f3 = make_fold_role(fold3, grad, partials3, nullptr, 0);
  FoldRole f2 = make_fold_role(fold2, grad, partials2, reset_slots, n_reset);
  BwdChainN n;
  n.fd = n.fw = n.fh = 0;
  n.d3 = d3.blocks(); n.w3 = w3.blocks(); n.d2 = d2.blocks(); n.w2 = w2.blocks(); n.w1 = w1.blocks(); n.f3 = f3.blocks(); n.f2 = f2.blocks();
  *n_partials3 = n.f3; *n_partials2 = n.f2;
  unsigned* cA = counters;                       // conv3's input-gradient workgroups, per sample
  unsigned* cB = counters + 32 * kChainLine;     // conv2's
  unsigned* cW3 = counters + 64 * kChainLine;    // conv3's weight-gradient workgroups, all samples
  unsigned* cW2 = counters + 65 * kChainLine;
  unsigned* cF = counters + 66 * kChainLine;     // fc4's input-gradient workgroups (DRA_VAR_BWD_CHAIN_FC)
  ChainHook base;
  base.epoch = epoch; base.epoch_bias = 0; base.timeout_flag = timeout_flag;
  FCD fd = {};
  FCW fw = {};
  HeadWgradRole fh;
  fh.dq = nullptr; fh.h4 = nullptr; fh.dwh = nullptr; fh.dbh = nullptr; fh.B = 0; fh.A = 0;
  if (fc) {
    if (!fc->dq || !fc->h4 || !fc->dh4 || !fc->x3 || !fc->w4 || !fc->dwh || !fc->dbh || !fc->dw4 || !fc->db4 || fc->n_actions < 1 ||
        fc->in_features != G3::OC * G3::P || !fc->sq_partials || !fc->n_sq_partials || !fc_wgrad_one(batch))
      return DRA_EINVAL;
    // (the roles of dra_fc_bwd_fused_sq, argument for argument; the input gradient IS dy3)
    fd.dy = fc->dh4; fd.w = fc->w4; fd.xact = fc->x3; fd.dx = const_cast<float*>(dy3); fd.B = batch; fd.I = fc->in_features; fd.act = act;
    fd.tiles_n = (fc->in_features + 31) / 32;
    fd.hook = base; fd.hook.done = cF;
    fw.dy = fc->dh4; fw.x = fc->x3; fw.dw = fc->dw4; fw.db = fc->db4; fw.partials = fc->sq_partials; fw.B = batch; fw.O = 512;
    fw.I = fc->in_features; fw.tiles_o = 512 / 32; fw.groups_i = (fd.tiles_n + kFcWgradNI - 1) / kFcWgradNI;
    fh.dq = fc->dq; fh.h4 = fc->h4; fh.dwh = fc->dwh; fh.dbh = fc->dbh; fh.B = batch; fh.A = fc->n_actions;
    if (fc->head_action && fc->head_group > 1 && fc->n_actions % fc->head_group == 0) { fh.action = fc->head_action; fh.group = fc->head_group; }
    n.fd = fd.tiles_n * ((batch + 31) / 32); n.fw = fw.blocks(); n.fh = 2 * fc->n_actions;
    fh.partials = fc->sq_partials + n.fw;
    *fc->n_sq_partials = n.fw + 2 * fc->n_actions;
  }
  d3.hook = base; d3.hook.done = cA; d3.hook.done_stride = kChainLine;
  w3.hook = base; w3.hook.done = cW3;
  if (fc) {
    // (480 workgroups on ONE counter: polled slowly, like the folds' -- tight polls on the line starve the producers' adds)
    d3.hook.wait = cF; d3.hook.wait_stride = 0; d3.hook.wait_target = (unsigned)n.fd; d3.hook.slow = 1;
    w3.hook.wait = cF; w3.hook.wait_stride = 0; w3.hook.wait_target = (unsigned)n.fd; w3.hook.slow = 1;
  }
  d2.hook = base; d2.hook.wait = cA; d2.hook.wait_stride = kChainLine; d2.hook.wait_target = BD3::WGS_PER_SAMPLE;
  d2.hook.done = cB; d2.hook.done_stride = kChainLine;
  w2.hook = base; w2.hook.wait = cA; w2.hook.wait_stride = kChainLine; w2.hook.wait_target = BD3::WGS_PER_SAMPLE; w2.hook.done = cW2;
  w1.hook = base; w1.hook.wait = cB; w1.hook.wait_stride = kChainLine; w1.hook.wait_target = BD2::WGS_PER_SAMPLE;
  f3.hook = base; f3.hook.wait = cW3; f3.hook.wait_target = (unsigned)n.w3; f3.hook.slow = 1;
  f2.hook = base; f2.hook.wait = cW2; f2.hook.wait_target = (unsigned)n.w2; f2.hook.slow = 1;
  constexpr int fa = BD3::LDS_FLOATS > WG3l::LDS_FLOATS ? BD3::LDS_FLOATS : WG3l::LDS_FLOATS;
  constexpr int fb = BD2::LDS_FLOATS > WG2l::LDS_FLOATS ? BD2::LDS_FLOATS : WG2l::LDS_FLOATS;
  constexpr int fcl = WG1u::LDS_FLOATS > FoldRole::LDS_FLOATS ? WG1u::LDS_FLOATS : FoldRole::LDS_FLOATS;
  constexpr int fab = fa > fb ? fa : fb, fl = fab > fcl ? fab : fcl;
  constexpr size_t bytes = (size_t)fl * sizeof(float);
  static_assert(bytes <= 64 * 1024, "LDS per workgroup of the chained backward launch");
  const int grid = n.fd + n.fw + n.fh + n.d3 + n.w3 + n.d2 + n.w2 + n.w1 + n.f3 + n.f2;
  if (fc) {
    // fc4's input gradient stages dh4 as [32][513] floats: 65.7 KB -- still two workgroups per CU
    constexpr size_t bytes_fc = (size_t)(FCD::LDS_FLOATS > fl ? FCD::LDS_FLOATS : fl) * sizeof(float);
    static_assert(bytes_fc <= 80 * 1024, "two workgroups of the chained backward launch per CU");
    static DraLdsAttr lds_attr;
    if (bytes_fc > 64 * 1024)
      if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&bwd_chain_kernel<true>), bytes_fc)) return rc;
    hipLaunchKernelGGL(bwd_chain_kernel<true>, dim3(grid), dim3(256), bytes_fc, dra_stream(stream), fd, fw, fh, d3, w3, d2, w2, w1, f3, f2, n);
  } else {
    hipLaunchKernelGGL(bwd_chain_kernel<false>, dim3(grid), dim3(256), bytes, dra_stream(stream), fd, fw, fh, d3, w3, d2, w2, w1, f3, f2, n);
  }
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Backward of head + fc4 in one launch (VanillaNet over NatureConvBody, hidden = 512):
//   dWh / dbh (HeadWgradRole), dW4 / db4 (LinWgrad), dx3 = relu'(x3) * (dh4 . W4) (LinDgrad / LinDgradOne).
// dq [B][A], h4 [B][512] (post-ReLU), dh4 [B][512] (gradient w.r.t. fc4's pre-activation), x3 [B][I].
// sq_partials (optional, DRA_VAR_LATE_FOLD; one-pass input gradient only): the workgroups that write dW4 / db4 and dWh / dbh
// leave their sums of squares in sq_partials[0, *n_sq_partials): fc4's tiles first, then the head's 2 * n_actions
// fc4's weight gradient of the learner's launch: the register-only role for minibatches up to 32 (oneshot_lin.h), else the
// K-chunked implicit GEMM (same box: fc_bwd 13.1 -> 11.2 us, +1.6 % updates/s; profiles/r04e_ab_env.jsonl).
// partials dra_fc_bwd_fused_sq writes for this problem (library-internal, actor_env.h): the learner lays the later launches'
// partials and the optimizer's arrival slots out behind them
int dra_fc_bwd_fused_sq_partials(int batch, int n_actions, int in_features) {
  const int tiles_i = (in_features + 31) / 32;
  const int nw = fc_wgrad_one(batch) ? (512 / 32) * ((tiles_i + kFcWgradNI - 1) / kFcWgradNI) : (512 / 64) * ((in_features + 1 + 63) / 64);
  return nw + 2 * n_actions;
}

int dra_fc_bwd_fused_sq(const float* dq, const float* h4, const float* dh4, const float* x3, const float* w4, float* dwh,
                        float* dbh, float* dw4, float* db4, float* dx3, int batch, int n_actions, int in_features, int act,
                        int variant, double* sq_partials, int* n_sq_partials, const int64_t* head_action, int head_group,
                        void* stream) {
  if (!dq || !h4 || !dh4 || !x3 || !w4 || !dwh || !dbh || !dw4 || !db4 || !dx3 || batch < 1 || n_actions < 1 ||
      in_features < 1 || !sq_partials || !n_sq_partials || !(variant & DRA_VAR_ONESHOT_DGRAD))
    return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  constexpr int O = 512;
  LinDgradOne<O> rd;
  rd.dy = dh4; rd.w = w4; rd.xact = x3; rd.dx = dx3; rd.B = batch; rd.I = in_features; rd.act = act;
  rd.tiles_n = (in_features + 31) / 32;
  HeadWgradRole rh;
  rh.dq = dq; rh.h4 = h4; rh.dwh = dwh; rh.dbh = dbh; rh.B = batch; rh.A = n_actions;
  if (head_action && head_group > 1 && n_actions % head_group == 0) { rh.action = head_action; rh.group = head_group; }
  if (fc_wgrad_one(batch)) {
    constexpr int NI = kFcWgradNI;
    LinWgradOne<NI> rl;
    rl.dy = dh4; rl.x = x3; rl.dw = dw4; rl.db = db4; rl.partials = sq_partials; rl.B = batch; rl.O = O; rl.I = in_features;
    rl.tiles_o = O / 32; rl.groups_i = (rd.tiles_n + NI - 1) / NI;
    const int nw = rl.blocks();
    rh.partials = sq_partials + nw;
    *n_sq_partials = nw + 2 * n_actions;
    return launch_multi(rd, rd.tiles_n * ((batch + 31) / 32), rl, nw, rh, 2 * n_actions, st);
  }
  LinWgradSq<64, 64, 32> pw;
  pw.M = O; pw.N = in_features + 1; pw.K = batch; pw.I = in_features; pw.dy = dh4; pw.x = x3; pw.dw = dw4; pw.db = db4;
  pw.partials = sq_partials;
  auto rw = make_igemm_role(pw, 1);
  const int nw = igemm_blocks(rw, 1);
  rh.partials = sq_partials + nw;
  *n_sq_partials = nw + 2 * n_actions;
  return launch_multi(rd, rd.tiles_n * ((batch + 31) / 32), rw, nw, rh, 2 * n_actions, st);
}

DRA_API int dra_fc_bwd_fused(const float* dq, const float* h4, const float* dh4, const float* x3, const float* w4,
                             float* dwh, float* dbh, float* dw4, float* db4, float* dx3, int batch, int n_actions,
                             int in_features, int act, int variant, void* stream) {
  if (!dq || !h4 || !dh4 || !x3 || !w4 || !dwh || !dbh || !dw4 || !db4 || !dx3 || batch < 1 || n_actions < 1 ||
      in_features < 1)
    return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  constexpr int O = 512;
  HeadWgradRole rh;
  rh.dq = dq; rh.h4 = h4; rh.dwh = dwh; rh.dbh = dbh; rh.B = batch; rh.A = n_actions;
  LinWgrad<64, 64, 32> pw;
  pw.M = O; pw.N = in_features + 1; pw.K = batch; pw.I = in_features; pw.dy = dh4; pw.x = x3; pw.dw = dw4; pw.db = db4;
  auto rw = make_igemm_role(pw, 1);
  if (variant & DRA_VAR_ONESHOT_DGRAD) {
    LinDgradOne<O> rd;
    rd.dy = dh4; rd.w = w4; rd.xact = x3; rd.dx = dx3; rd.B = batch; rd.I = in_features; rd.act = act;
    rd.tiles_n = (in_features + 31) / 32;
    if (fc_wgrad_one(batch)) {     // register-only weight gradient (oneshot_lin.h), as in the learner's launch
      constexpr int NI = kFcWgradNI;
      LinWgradOne<NI> rl;
      rl.dy = dh4; rl.x = x3; rl.dw = dw4; rl.db = db4; rl.partials = nullptr; rl.B = batch; rl.O = O; rl.I = in_features;
      rl.tiles_o = O / 32; rl.groups_i = (rd.tiles_n + NI - 1) / NI;
      return launch_multi(rd, rd.tiles_n * ((batch + 31) / 32), rl, rl.blocks(), rh, 2 * n_actions, st);
    }
    return launch_multi(rd, rd.tiles_n * ((batch + 31) / 32), rw, igemm_blocks(rw, 1), rh, 2 * n_actions, st);
  }
  LinDgrad<32, 32, 64> pd;
  pd.M = batch; pd.N = in_features; pd.K = O; pd.dy = dh4; pd.w = w4; pd.xact = x3; pd.dx = dx3; pd.act = act;
  auto rd = make_igemm_role(pd, 1);
  return launch_multi(rd, igemm_blocks(rd, 1), rw, igemm_blocks(rw, 1), rh, 2 * n_actions, st);
}

// Input gradient of a 512-output linear layer on its own (fc4 of NatureConvBody in the autograd path: dx = act'(xact) * dy W,
// dy [B][512], W [512][I]) through the learner's one-pass role instead of the K-chunked implicit GEMM (24.9 us at a PPO
// minibatch of 256, profiles/r04ap_kernel_stats_ppo_pixel_8.txt).  Library-internal (igemm.hip dra_linear_bwd_x).
extern "C" int dra_linear_bwd_x_one512(const float* dy, const float* w, const float* xact, float* dx, int batch, int in_features,
                                       int act, void* stream) {
  if (!dy || !w || !dx || batch < 1 || in_features < 1) return DRA_EINVAL;
  LinDgradOne<512> rd;
  rd.dy = dy; rd.w = w; rd.xact = xact; rd.dx = dx; rd.B = batch; rd.I = in_features; rd.act = act;
  rd.tiles_n = (in_features + 31) / 32;
  NoRole none;
  return launch_multi(rd, rd.tiles_n * ((batch + 31) / 32), none, 0, none, 0, dra_stream(stream));
}

// Both gradients of such a layer in ONE launch: the input gradient's one-pass role and the weight / bias gradient's implicit GEMM
// (igemm.hip dra_linear_bwd_w's own problem) read the same dy and do not depend on each other -- back to back they were 9.9 + 10.1
// us of an A2C update and 17.5 + 20 us of a PPO minibatch, each launch filling the chip partly
// (profiles/r05z2_kernel_stats_a2c_pixel_16.txt).  Same roles, same arithmetic.
DRA_API int dra_linear_bwd_xw_one512(const float* dy, const float* w, const float* x, int x_is_relu_output, float* dx, float* dw,
                                     float* db, int batch, int in_features, void* stream) {
  if (!dy || !w || !x || !dx || !dw || batch < 1 || in_features < 1024) return DRA_EINVAL;
  LinDgradOne<512> rd;
  rd.dy = dy; rd.w = w; rd.xact = x_is_relu_output ? x : nullptr; rd.dx = dx; rd.B = batch; rd.I = in_features;
  rd.act = x_is_relu_output ? ACT_RELU : ACT_NONE;
  rd.tiles_n = (in_features + 31) / 32;
  LinWgrad<64, 64, 32> p;
  p.M = 512; p.N = in_features + 1; p.K = batch; p.I = in_features; p.dy = dy; p.x = x; p.dw = dw; p.db = db;
  auto rw = make_igemm_role(p, 1);
  NoRole none;
  return launch_multi(rd, rd.tiles_n * ((batch + 31) / 32), rw, rw.tiles, none, 0, dra_stream(stream));
}

// fc4 forward partial sums in one pass per K split (in_features = 3136, ksplit = 8): same contract as
// dra_linear_fwd_slabs.
DRA_API int dra_linear_fwd_slabs_one(int nz, const float* const* x, const float* const* w, int batch, int in_features,
                                     int out_features, int ksplit, float* slabs, void* stream) {
  if (nz < 1 || nz > kMaxZ || batch < 1 || !x || !w || !slabs) return DRA_EINVAL;
  if (in_features != 3136 || (ksplit != 8 && ksplit != 14 && ksplit != 28) || out_features < 1) return DRA_EINVAL;
  for (int z = 0; z < nz; ++z) {
    if (!x[z] || !w[z]) return DRA_EINVAL;
    if ((((uintptr_t)x[z]) | ((uintptr_t)w[z])) & 15) return DRA_EINVAL;
  }
  constexpr int NT = 2;
  NoRole none;
  auto fill = [&](auto& r) {
    for (int z = 0; z < nz; ++z) { r.x[z] = x[z]; r.w[z] = w[z]; }
    r.slabs = slabs; r.B = batch; r.O = out_features;
    r.tiles_n = (out_features + 32 * NT - 1) / (32 * NT); r.tiles_m = (batch + 31) / 32;
    r.xcd = dra_xcd_order_enabled(); r.n_groups = r.tiles_m * ksplit * nz;
  };
  if (ksplit == 28) {   // 112-wide K slices: 3.5x the workgroups (448 at batch 32, two nets), 43 KB of LDS each
    LinFwdSlabsOne<3136, 28, NT> r;
    fill(r);
    return launch_multi(r, r.tiles_n * r.tiles_m * 28 * nz, none, 0, none, 0, dra_stream(stream));
  }
  if (ksplit == 14) {   // 224-wide K slices: 224 workgroups at batch 32 and two nets, 86 KB of LDS each (one per CU)
    LinFwdSlabsOne<3136, 14, NT> r;
    fill(r);
    return launch_multi(r, r.tiles_n * r.tiles_m * 14 * nz, none, 0, none, 0, dra_stream(stream));
  }
  LinFwdSlabsOne<3136, 8, NT> r;
  fill(r);
  return launch_multi(r, r.tiles_n * r.tiles_m * 8 * nz, none, 0, none, 0, dra_stream(stream));
}

// Head contraction of the distributional update in one pass (library-internal, actor_env.h): y_z[b][o] = bias_z[o] +
// <x_z[b], w_z[o]> for in_features = 512, whole reduction per workgroup (32 x 32 output tile, K = 512 over 4 waves x 2
// half-waves of fp32 MFMA, both operands staged through LDS once).  Replaces head_fwd_gemv_kernel's 32 serial wave-level dot
// products per output row (12 us at [32,512] x [512, 204 | 800], profiles/r02zv_kernel_stats_*_after.txt).
int dra_head_fwd_one(int nz, const float* const* x, const float* const* w, const float* const* bias, float* const* y, int batch,
                     int out_features, void* stream) {
  if (nz < 1 || nz > kMaxZ || batch < 1 || out_features < 1 || !x || !w || !bias || !y) return DRA_EINVAL;
  LinFwdSlabsOne<512, 1, 1> r;
  for (int z = 0; z < nz; ++z) {
    if (!x[z] || !w[z] || !bias[z] || !y[z] || ((((uintptr_t)x[z]) | ((uintptr_t)w[z])) & 15)) return DRA_EINVAL;
    r.x[z] = x[z]; r.w[z] = w[z]; r.bias[z] = bias[z]; r.out[z] = y[z];
  }
  r.slabs = nullptr; r.B = batch; r.O = out_features;
  r.tiles_n = (out_features + 31) / 32; r.tiles_m = (batch + 31) / 32;
  NoRole none;
  return launch_multi(r, r.tiles_n * r.tiles_m * nz, none, 0, none, 0, dra_stream(stream));
}

#ifdef DRA_TRACE
extern "C" int dra_trace_set_fused(void* p) { return dra_trace_set_local(p); }
#endif
