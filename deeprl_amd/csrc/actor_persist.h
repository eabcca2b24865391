// DRA_VAR_ACTOR_PERSIST (round 6): ALL env steps of a DQN agent step as ONE launch (DQN_agent.py:24-45, n_env x [forward ->
// epsilon-greedy -> env.step]).  Included by conv_v2.hip (uses its geometry types and the fused head / environment helpers).
//
// Why: the actor chain is 4 x [conv1 (+ head, env step) | conv2 | conv3 + fc4] + a tail = 13 dependent launches, 4 x 25.7 + 6 us,
// as long as the update chain beside it.  A dependent launch costs 1.72 us before its first instruction, and its operands start
// cold; the round-3 one-launch env step (arrival counters) measured no better, because a counter hand-over is four serial trips
// through the fabric (stores acknowledged -> counter -> poll -> loads: 2.2-3.7 us, tools/ubench/ll_handover.hip).  Here every
// activation travels as ONE 8-byte {value, tag} word (the "LL" form of the collective libraries): the producer just stores, the
// consumer re-reads its own words until they carry the stage's tag -- 1.5-1.7 us per hop on this part, nothing to reset (the tag
// is the device step counter x 8 + env step + 1, monotonic), and NO weight is ever fetched twice: the 64 workgroups (two per CU of
// the actor's 32-CU partition, all co-resident) keep fc4's 6.4 MB in registers for the whole agent step and the conv roles
// their operands, conv1's workgroups keep the frame rows they convolve in LDS (a 4-deep history: one new frame per env step).
//
// Roles (512 threads each; every workgroup ALSO owns 8 rows of fc4):
//   [0, 13)  conv1 tile b (+ head of the previous env step, epsilon-greedy, the rows of the new observation it convolves)
//   [13, 25) conv2: 3 position tiles x 2 channel tiles x 2 K halves (partial planes, as conv_b1_split_body)
//   [25, 33) conv3: 2 x 2 x 2 (reads conv2's two planes, ReLU while staging)
//   63       the environment side: commits the pending observation, records actions, writes the generated observations to the
//            replay ring, and after the last env step prepares the next agent step's first observation and advances the counter
// Arithmetic and summation order of every output are those of the multi-launch path (conv_fwd_v2_body<NW = 8>,
// conv_b1_split_body, mega_fc4_role, fused_head_q): bit-identical action values, actions and ring contents
// (tests/test_gpu_agents.py::test_persistent_actor_is_bit_identical).
//
// Ordering argument for the single-buffered hand-over arrays: stage s+1 of env step e reads stage s's output before it writes its
// own, and conv1 of step e+1 starts only when ALL 512 features of step e exist, i.e. when every fc4 workgroup has read all of
// conv3's output, which needs every conv3 workgroup to have read its conv2 planes, ... -- so every reader of step e's arrays is
// done before step e+1's producers write them.  The features are double-buffered by the parity of e because the environment
// workgroup reads them outside that chain.  Ring slots written here are read by nobody inside the launch.
#pragma once

constexpr int kPC1 = 13, kPC2 = 12, kPC3 = 8;   // conv1: [0, 13), conv2: [13, 25), conv3: [0, 8) (conv1's first workgroups, later in the step)
constexpr int kPEnvWg = kPersistWgs - 1;
constexpr int kPR_C1 = 1, kPR_C2 = 2, kPR_C3 = 4, kPR_ENV = 8;   // role bits of a workgroup
constexpr unsigned long long kPersistWaitTicks = 5000000ull;   // 50 ms of the 100 MHz wall clock per wait

__device__ __forceinline__ ll_t ll_load(const ll_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ll_store(ll_t* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((ll_t)tag << 32) | (ll_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool ll_ok(ll_t w, unsigned tag) { return (unsigned)(w >> 32) == tag; }
__device__ __forceinline__ float ll_val(ll_t w) { return __uint_as_float((unsigned)w); }

// bounded waits: a wait that gives up sets the pinned flag (the host reports DRA_ETIMEDOUT: the results are invalid) and the
// device abort word, which every other wait of the launch notices at its next check
struct PersistClock {
  unsigned long long t0;
  int spins;
  int* abort_word;
  int* timeout_flag;
  __device__ __forceinline__ void start() { t0 = wall_clock64(); spins = 0; }
  __device__ __forceinline__ bool expired() {
    if ((++spins & 255) != 0) return false;
    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    if (wall_clock64() - t0 > kPersistWaitTicks) {
      __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (timeout_flag) __hip_atomic_store(timeout_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return true;
    }
    return false;
  }
};

// one thread watches ONE word of a stage further up the chain at a slow pace (so that the workgroup's full-rate polling of its own
// input lasts one stage, not three); everybody leaves together
__device__ __forceinline__ void ll_probe(const ll_t* p, unsigned tag, PersistClock& ck) {
  if (threadIdx.x == 0) {
    ck.start();
    while (!ll_ok(ll_load(p), tag)) {
      __builtin_amdgcn_s_sleep(4);
      if (ck.expired()) break;
    }
  }
  __syncthreads();
}

// N words per thread, re-read until every one carries `tag`
template <int N, class AddrFn>
__device__ __forceinline__ void ll_poll(ll_t (&w)[N], unsigned tag, PersistClock& ck, AddrFn addr) {
  ck.start();
  for (;;) {
#pragma unroll
    for (int q = 0; q < N; ++q) w[q] = ll_load(addr(q));
    bool ok = true;
#pragma unroll
    for (int q = 0; q < N; ++q) ok = ok && ll_ok(w[q], tag);
    if (ok) break;
    if (ck.expired()) break;
  }
}


// per-lane state of one conv role: tile geometry + the weight operands / bias that stay in registers for the whole agent step
template <int NJ>
struct PersistConv {
  float areg[NJ];
  float bias_r[2];
  int p0, np, oh0, ir0, nrows, oc0, kz, pj;
};

// ROLES = the role bits of the workgroup (one instantiation per combination that occurs: the loop-invariant addresses of a role
// the workgroup does not play would otherwise stay live through the env-step loop -- a first single-body version needed 396
// registers).  Every workgroup owns 16 rows of fc4: rows 16 b + wave in registers, rows 16 b + 8 + wave in LDS.
template <int ROLES>
__device__ __forceinline__ void actor_persist_body(const ActorPersistArgs& a, float* __restrict__ lds, float* __restrict__ s_lut,
                                                   float* __restrict__ s_h4, float* __restrict__ s_q, unsigned* __restrict__ s_prm) {
  using T1 = V2Tile<VG1, 1>;
  using T2 = V2Tile<VG2, 1>;
  using T3 = V2Tile<VG3, 1>;
  constexpr bool C1 = (ROLES & kPR_C1) != 0, C2 = (ROLES & kPR_C2) != 0, C3 = (ROLES & kPR_C3) != 0, ENV = (ROLES & kPR_ENV) != 0;
  constexpr int kImgFloats = 4 * T1::CS;                 // conv1's 4-deep frame history
  constexpr int kImg23 = (VG2::C / 2) * T2::CS > (VG3::C / 2) * T3::CS ? (VG2::C / 2) * T2::CS : (VG3::C / 2) * T3::CS;
  constexpr int I4 = VG3::OC * VG3::P, NV4 = I4 / 4, R4 = (NV4 + 63) / 64;
  static_assert(kImg23 <= 8 * 16 * 64 && I4 <= 8 * 16 * 64, "regions that share `red`");
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned sq = *a.seq;
  const unsigned tag0 = sq * 8u;
  const int A = a.n_actions, n_env = a.n_env;
  PersistClock ck;
  ck.abort_word = a.abort_word; ck.timeout_flag = a.timeout_flag;

  // ---- the parameter block of this agent step -> LDS (the head / epsilon-greedy / environment fields are read on the path)
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(aring_entry(a.aring, sq));
    if (tid < (int)(kPrmHeadBytes / 4)) s_prm[tid] = src[tid];
    if (tid < 256) s_lut[tid] = (float)((double)tid * a.coef);
  }
  const dra_dqn_step_params* prm = reinterpret_cast<const dra_dqn_step_params*>(s_prm);

  // ---- DRA_VAR_FLAG_SYNC: the parameter copy read below is complete once the update stream has STARTED launch number `need`
  // (everything it ran before that launch -- the optimizer that wrote the copy -- is then complete and written back); the host left
  // `need` for this agent step in its pinned ring before issuing this launch.  One poller per workgroup, an agent-scope acquire
  // behind it (nothing of the copy can be in this CU's caches from before: the launch boundary invalidated them and nobody here
  // has read the copy since -- the fence is the formal half of that)
  if (a.fs_count) {
    if (tid == 0) {
      const unsigned long long need = __hip_atomic_load(a.fs_need + (sq % kAringSlots), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ck.start();
      while (__hip_atomic_load(a.fs_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (ck.expired()) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }

  // ---- role geometry + operands
  PersistConv<16> c1;
  PersistConv<16> c2;
  PersistConv<18> c3;
  if constexpr (C1) {
    using G = VG1;
    c1.kz = 0; c1.oc0 = 0;
    c1.p0 = b * 32; c1.np = min(32, G::P - c1.p0); c1.oh0 = c1.p0 / G::OH;
    const int oh1 = (c1.p0 + c1.np - 1) / G::OH;
    c1.ir0 = c1.oh0 * G::S; c1.nrows = (oh1 - c1.oh0) * G::S + G::KH;
    c1.pj = min(li, c1.np - 1);
    const int cp0 = wave % G::CP, t0 = (wave / G::CP) * 16;
    const float* wbase = a.w1 + ((int64_t)(2 * cp0 + h) * G::KK + t0) * G::OC + li;
#pragma unroll
    for (int j = 0; j < 16; ++j) c1.areg[j] = wbase[j * G::OC];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int r = wave * 2 + q; c1.bias_r[q] = a.b1[(r & 3) + 8 * (r >> 2) + 4 * h]; }
  }
  if constexpr (C2) {
    using G = VG2;
    const int i2 = b - kPC1, bx = i2 % G::TPS, by = i2 / G::TPS;
    c2.kz = by / (G::OC / 32); c2.oc0 = (by - c2.kz * (G::OC / 32)) * 32;
    c2.p0 = bx * 32; c2.np = min(32, G::P - c2.p0); c2.oh0 = c2.p0 / G::OH;
    const int oh1 = (c2.p0 + c2.np - 1) / G::OH;
    c2.ir0 = c2.oh0 * G::S; c2.nrows = (oh1 - c2.oh0) * G::S + G::KH;
    c2.pj = min(li, c2.np - 1);
    const int cp0 = c2.kz * (G::CP / 2) + wave;
    const float* wbase = a.w2 + ((int64_t)(2 * cp0 + h) * G::KK) * G::OC + c2.oc0 + li;
#pragma unroll
    for (int j = 0; j < 16; ++j) c2.areg[j] = wbase[j * G::OC];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int r = wave * 2 + q; c2.bias_r[q] = c2.kz == 0 ? a.b2[c2.oc0 + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f; }
  }
  if constexpr (C3) {
    using G = VG3;
    const int bx = b % G::TPS, by = b / G::TPS;
    c3.kz = by / (G::OC / 32); c3.oc0 = (by - c3.kz * (G::OC / 32)) * 32;
    c3.p0 = bx * 32; c3.np = min(32, G::P - c3.p0); c3.oh0 = c3.p0 / G::OH;
    const int oh1 = (c3.p0 + c3.np - 1) / G::OH;
    c3.ir0 = c3.oh0 * G::S; c3.nrows = (oh1 - c3.oh0) * G::S + G::KH;
    c3.pj = min(li, c3.np - 1);
    const int cp0 = c3.kz * (G::CP / 2) + wave * 2;
    const float* wbase = a.w3 + ((int64_t)(2 * cp0 + h) * G::KK) * G::OC + c3.oc0 + li;
#pragma unroll
    for (int j = 0; j < 18; ++j) { const int cpl = j / G::KK, t = j - cpl * G::KK; c3.areg[j] = wbase[(2 * cpl * G::KK + t) * G::OC]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int r = wave * 2 + q; c3.bias_r[q] = c3.kz == 0 ? a.b3[c3.oc0 + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f; }
  }
  // head operands (conv1 workgroups and the environment workgroup): action `wave`'s weights in registers
  float whr[8];
  float bhr = 0.f;
  constexpr bool HEADS = C1 || ENV;
  if (HEADS && wave < A) {
#pragma unroll
    for (int i = 0; i < 8; ++i) whr[i] = a.wh[wave * 512 + lane + 64 * i];
    bhr = a.bh[wave];
  }

  // ---- fc4 operands: requested AFTER the workgroup's first conv stage (a wave's loads return in order: in front of it they
  // would hold up its first hand-over)
  float4 wv[R4];
  float bias4a = 0.f, bias4b = 0.f;
  const int row4a = b * 16 + wave, row4b = row4a + 8;
  __syncthreads();   // parameter block + normalisation table staged

  for (int e = 0; e <= n_env; ++e) {
    const unsigned tag = tag0 + (unsigned)e + 1u;
    // the hand-over arrays' addresses are recomputed every env step from these (opaque) bases: hoisted out of the loop they would
    // occupy ~100 registers for the whole agent step
    ll_t *py1 = a.y1, *py2p = a.y2p, *py3p = a.y3p, *ph4 = a.h4;
    const float* pw4 = a.w4;
    asm volatile("" : "+s"(py1), "+s"(py2p), "+s"(py3p), "+s"(ph4), "+s"(pw4));
    // ... and so are the LDS regions (the compiler otherwise keeps ~60 precomputed LDS addresses alive across the loop)
    int oz = 0;
    asm volatile("" : "+s"(oz));
    float* const img1 = lds + oz;                                       // conv1: [4][NR][RW], kept across env steps
    float* const red = lds + kImgFloats + oz;                           // [8 waves][16][64] partial sums / fc4's input vector
    float* const img23 = red;                                           // conv2 / conv3 image (reused for the partial sums)
    float4* const wl4 = reinterpret_cast<float4*>(red + 8 * 16 * 64);   // [8 waves][784] fc4 rows 16 b + 8 + wave
    // q = head(features of env step eh) into s_q (every thread returns after the barrier), as fused_head_q: one wave per action
    auto head_q = [&](int eh, const ll_t* h4base) {
      ll_t w[1];
      const ll_t* src = h4base + (eh & 1) * 512 + tid;
      ll_poll<1>(w, tag0 + (unsigned)eh + 1u, ck, [&](int) { return src; });
      s_h4[tid] = ll_val(w[0]);
      __syncthreads();
      for (int act = wave; act < A; act += 8) {
        float part = 0.f;
        if (act == wave) {
  #pragma unroll
          for (int i = 0; i < 8; ++i) part += s_h4[lane + 64 * i] * whr[i];
        } else {
  #pragma unroll
          for (int i = 0; i < 8; ++i) part += s_h4[lane + 64 * i] * a.wh[act * 512 + lane + 64 * i];
        }
        part = wave_sum(part);
        if (lane == 0) s_q[act] = part + (act == wave ? bhr : a.bh[act]);
      }
      __syncthreads();
    };

    // fold of the 8 waves' partial accumulators (in `red`), bias, optional ReLU, {value, tag} stores -- the epilogue of
    // conv_fwd_v2_body<NW = 8> / conv_b1_split_body
    auto fold_store = [&](const f32x16& acc, const float (&bias_r)[2], int oc0, int p0, int np, ll_t* y, int P, bool relu, unsigned tag) {
  #pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
      __syncthreads();
  #pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = wave * 2 + q;
        float s = (red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) +
                  (red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane]);
        s += (red[(4 * 16 + r) * 64 + lane] + red[(5 * 16 + r) * 64 + lane]) +
             (red[(6 * 16 + r) * 64 + lane] + red[(7 * 16 + r) * 64 + lane]);
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = s + bias_r[q];
        if (relu) v = v > 0.f ? v : 0.f;
        if (li < np) ll_store(y + (int64_t)(oc0 + row) * P + p0 + li, v, tag);
      }
      __syncthreads();   // `red` is free again
    };

    auto request_fc4 = [&]() {
      if (a.w4_valid) {   // the copy's fc4 segment is completed by riders of the update running beside this launch
        if (tid == 0) {
          ck.start();
          while (__hip_atomic_load(a.w4_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(4);
            if (ck.expired()) break;
          }
        }
        __syncthreads();
      }
      const float4* __restrict__ w4a = reinterpret_cast<const float4*>(pw4 + (int64_t)row4a * I4);
      const float4* __restrict__ w4b = reinterpret_cast<const float4*>(pw4 + (int64_t)row4b * I4);
  #pragma unroll
      for (int q = 0; q < R4; ++q) wv[q] = w4a[min(lane + 64 * q, NV4 - 1)];
      bias4a = a.b4[row4a]; bias4b = a.b4[row4b];
      // the LDS row in chunks of 4 requests (all 13 at once would need 52 more registers at the kernel's tightest point)
  #pragma unroll
      for (int q0 = 0; q0 < R4; q0 += 4) {
        float4 tmp[4];
  #pragma unroll
        for (int q = q0; q < q0 + 4 && q < R4; ++q) tmp[q - q0] = w4b[min(lane + 64 * q, NV4 - 1)];
  #pragma unroll
        for (int q = q0; q < q0 + 4 && q < R4; ++q)
          if (lane + 64 * q < NV4) wl4[wave * NV4 + lane + 64 * q] = tmp[q - q0];   // (read back by the same wave only)
        __builtin_amdgcn_sched_barrier(0);
      }
    };


    // ================= environment workgroup: what happens between forward e-1 and forward e =================
    if constexpr (ENV) {
      if (e == 0) {
        if (prm->counter[0] >= 0) {   // the pending observation becomes ring slot[0]
          const int64_t slot = prm->slot[0];
          const uint64_t* src = reinterpret_cast<const uint64_t*>(a.pend_frame);
          uint64_t* dst = reinterpret_cast<uint64_t*>(a.frames + slot * 7056);
          for (int w = tid; w < 882; w += 512) dst[w] = src[w];
          if (tid == 0) { a.rewards[slot] = *a.pend_reward; a.masks[slot] = *a.pend_mask; }
        }
      } else {
        head_q(e - 1, ph4);
        if (tid < A && a.q_out) a.q_out[tid] = s_q[tid];
        const int64_t act = eps_greedy_action(s_q, A, prm, e - 1);
        if (tid == 0 && prm->store_action[e - 1]) *reinterpret_cast<int64_t*>(a.actions + prm->slot[e - 1] * 8) = act;
        if (e < n_env) {
          const int64_t counter = prm->counter[e];
          if (counter >= 0) {
            const int64_t slot = prm->slot[e];
            uint64_t* dst = reinterpret_cast<uint64_t*>(a.frames + slot * 7056);
            for (int w = tid; w < 882; w += 512) dst[w] = synth_frame_word(a.seed, counter, act, w);
            if (tid == 0) {
              a.rewards[slot] = synth_reward(a.seed, prm->rcounter[e]);
              a.masks[slot] = synth_mask(a.seed, prm->rcounter[e], a.done_period);
            }
          }
        } else {
          // after the last env step: the next agent step's first observation (envs.py:140-141), then the step counter
          const dra_dqn_step_params* nxt = aring_entry(a.aring, sq + 1u);
          const int64_t counter = nxt->counter[0];
          if (counter >= 0) {
            uint64_t* dst = reinterpret_cast<uint64_t*>(a.pend_frame);
            for (int w = tid; w < 882; w += 512) dst[w] = synth_frame_word(a.seed, counter, 0, w);
            if (tid == 0) {
              *a.pend_reward = synth_reward(a.seed, nxt->rcounter[0]);
              *a.pend_mask = synth_mask(a.seed, nxt->rcounter[0], a.done_period);
            }
          }
          if (tid == 0) {
            *a.seq = sq + 1u;
            // (DRA_VAR_FLAG_SYNC: every workgroup has read its share of the parameter copy long before the last head; the host
            // uses this count only to decide when a copy / a staging entry may be REUSED, never to read what this launch wrote)
            if (a.fs_done_host) __hip_atomic_store(a.fs_done_host, (unsigned long long)sq + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    }
    if (e == n_env) break;

    // ================= conv1 (+ head of e-1, epsilon-greedy, the new observation's rows) =================
    if constexpr (C1) {
      using G = VG1;
      constexpr int WPR = G::H / 4;
      const int r = tid >> 5, wd = tid & 31, wdc = min(wd, WPR - 1);
      const int rowc = c1.ir0 + min(r, c1.nrows - 1);
      if (e == 0) {
        // frames -3 .. 0 of the stack: ring slots slot[0] - k (k = 1..3) and the pending observation; frame f lives in
        // history slot f & 3
        const int64_t newest = prm->slot[0];
        unsigned raw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          int64_t s = newest - k;
          if (s < 0) s += a.ring_cap;
          const uint8_t* fp = k == 0 ? a.pend_frame : a.frames + s * 7056;
          raw[k] = reinterpret_cast<const unsigned*>(fp)[rowc * WPR + wdc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float* dst = img1 + ((0 - k) & 3) * T1::CS + r * G::RW + wd;
          if (wd < WPR && r < c1.nrows) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) dst[bb * G::WPH] = s_lut[(raw[k] >> (8 * bb)) & 0xffu];
          }
        }
      } else {
        head_q(e - 1, ph4);
        const int64_t act = eps_greedy_action(s_q, A, prm, e - 1);
        const int64_t counter = prm->counter[e];
        const int w32 = rowc * WPR + wdc;
        const uint64_t v64 = synth_frame_word(a.seed, counter, act, w32 >> 1);
        const unsigned word = (unsigned)(v64 >> (32 * (w32 & 1)));
        float* dst = img1 + (e & 3) * T1::CS + r * G::RW + wd;
        if (wd < WPR && r < c1.nrows) {
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) dst[bb * G::WPH] = s_lut[(word >> (8 * bb)) & 0xffu];
        }
      }
      __syncthreads();
      // channel c of the stack is the observation min(3 - c, stack_age) steps back (envs.py FrameStack)
      const int cp0 = wave % G::CP, t0 = (wave / G::CP) * 16;
      const int cl = 2 * cp0 + h;
      const int off = min(G::C - 1 - cl, (int)prm->stack_age[e]);
      const int phys = (e - off) & 3;
      const int poh = (c1.p0 + c1.pj) / G::OH, pow_ = (c1.p0 + c1.pj) - poh * G::OH;
      const float* bptr = img1 + phys * T1::CS + ((poh - c1.oh0) * G::S + t0 / G::KH) * G::RW + pow_;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kh = j / G::KH, kw = j - kh * G::KH;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.areg[j], bptr[kh * G::RW + (kw % G::S) * G::WPH + kw / G::S], acc, 0, 0, 0);
      }
      fold_store(acc, c1.bias_r, 0, c1.p0, c1.np, py1, G::P, true, tag);
    }

    // ================= conv2 =================
    if constexpr (C2) {
      using G = VG2;
      constexpr int LR = 32, RP = 2, LPT = (T2::NR + RP - 1) / RP, CL = G::C / 2, CPT = CL / 8;
      const int rsub = lane / LR, iw = lane % LR, iwc = min(iw, G::H - 1), col = lds_col<G>(iwc);
      if (e > 0) ll_probe(ph4 + ((e - 1) & 1) * 512, tag - 1u, ck);   // the features of step e-1 exist: conv1 of step e is under way
      ll_t w[CPT * LPT];
      ll_poll<CPT * LPT>(w, tag, ck, [&](int k) {
        const int ci = k / LPT, q = k - ci * LPT;
        const int c = c2.kz * CL + wave + 8 * ci;
        return py1 + ((int64_t)c * G::H + c2.ir0 + min(RP * q + rsub, c2.nrows - 1)) * G::H + iwc;
      });
#pragma unroll
      for (int ci = 0; ci < CPT; ++ci) {
        float* dst = img23 + (wave + 8 * ci) * T2::CS + rsub * G::RW + col;
#pragma unroll
        for (int q = 0; q < LPT; ++q)
          if (iw < G::H && RP * q + rsub < c2.nrows) dst[RP * q * G::RW] = ll_val(w[ci * LPT + q]);
      }
      __syncthreads();
      const int poh = (c2.p0 + c2.pj) / G::OH, pow_ = (c2.p0 + c2.pj) - poh * G::OH;
      const float* bptr = img23 + (2 * wave + h) * T2::CS + ((poh - c2.oh0) * G::S) * G::RW + pow_;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kh = j / G::KH, kw = j - kh * G::KH;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c2.areg[j], bptr[kh * G::RW + (kw % G::S) * G::WPH + kw / G::S], acc, 0, 0, 0);
      }
      __syncthreads();   // every wave is done reading the image: its region takes the partial sums
      fold_store(acc, c2.bias_r, c2.oc0, c2.p0, c2.np, py2p + (int64_t)c2.kz * G::OC * G::P, G::P, false, tag);
    }

    // ================= conv3 =================
    if constexpr (C3) {
      using G = VG3;
      constexpr int LR = 16, RP = 4, LPT = (T3::NR + RP - 1) / RP, CL = G::C / 2, CPT = CL / 8;
      const int rsub = lane / LR, iw = lane % LR, iwc = min(iw, G::H - 1), col = lds_col<G>(iwc);
      if constexpr (!C1) ll_probe(py1, tag, ck);                      // conv1 of this step has stored: conv2 is under way
      ll_t w[2 * CPT * LPT];
      ll_poll<2 * CPT * LPT>(w, tag, ck, [&](int k) {
        const int pl = k / (CPT * LPT), kk = k - pl * (CPT * LPT), ci = kk / LPT, q = kk - ci * LPT;
        const int c = c3.kz * CL + wave + 8 * ci;
        return py2p + (int64_t)pl * VG2::OC * VG2::P + ((int64_t)c * G::H + c3.ir0 + min(RP * q + rsub, c3.nrows - 1)) * G::H + iwc;
      });
#pragma unroll
      for (int ci = 0; ci < CPT; ++ci) {
        float* dst = img23 + (wave + 8 * ci) * T3::CS + rsub * G::RW + col;
#pragma unroll
        for (int q = 0; q < LPT; ++q) {
          float v = ll_val(w[ci * LPT + q]) + ll_val(w[CPT * LPT + ci * LPT + q]);
          v = v > 0.f ? v : 0.f;
          if (iw < G::H && RP * q + rsub < c3.nrows) dst[RP * q * G::RW] = v;
        }
      }
      __syncthreads();
      const int poh = (c3.p0 + c3.pj) / G::OH, pow_ = (c3.p0 + c3.pj) - poh * G::OH;
      const float* bptr = img23 + (2 * (wave * 2) + h) * T3::CS + ((poh - c3.oh0) * G::S) * G::RW + pow_;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
      for (int j = 0; j < 18; ++j) {
        const int cpl = j / G::KK, tp = j - cpl * G::KK;
        const int kh = tp / G::KH, kw = tp - kh * G::KH;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c3.areg[j], bptr[2 * cpl * T3::CS + kh * G::RW + (kw % G::S) * G::WPH + kw / G::S], acc, 0, 0, 0);
      }
      __syncthreads();
      fold_store(acc, c3.bias_r, c3.oc0, c3.p0, c3.np, py3p + (int64_t)c3.kz * G::OC * G::P, G::P, false, tag);
    }

    // ================= fc4: 16 rows per workgroup, one wave per row pair (mega_fc4_role's arithmetic per row) =================
    if (e == 0) request_fc4();
    {
      constexpr int XQ = (I4 + 511) / 512;
      if constexpr (!C3) ll_probe(py2p, tag, ck);                     // conv2 of this step has stored: conv3 is under way
      ll_t w[2 * XQ];
      ll_poll<2 * XQ>(w, tag, ck, [&](int k) {
        const int pl = k / XQ, q = k - pl * XQ;
        return py3p + (int64_t)pl * I4 + min(tid + 512 * q, I4 - 1);
      });
      float* xs = red;
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const int i = tid + 512 * q;
        if (i < I4) xs[i] = fmaxf(ll_val(w[q]) + ll_val(w[XQ + q]), 0.f);
      }
      __syncthreads();
      const float4* sx = reinterpret_cast<const float4*>(xs);
      float acca = 0.f, accb = 0.f;
#pragma unroll
      for (int q = 0; q < R4; ++q) {
        float4 wq = wv[q];
        asm volatile("" : "+v"(wq.x), "+v"(wq.y), "+v"(wq.z), "+v"(wq.w));
        const float4 x = sx[min(lane + 64 * q, NV4 - 1)];
        const float4 wb = wl4[wave * NV4 + min(lane + 64 * q, NV4 - 1)];
        if (lane + 64 * q < NV4) {
          acca += (wq.x * x.x + wq.y * x.y) + (wq.z * x.z + wq.w * x.w);
          accb += (wb.x * x.x + wb.y * x.y) + (wb.z * x.z + wb.w * x.w);
        }
        if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (keeps the 26 LDS reads of the loop from being hoisted in front of it)
      }
      acca = wave_sum(acca);
      accb = wave_sum(accb);
      if (lane == 0) {
        float va = acca + bias4a, vb = accb + bias4b;
        va = va > 0.f ? va : 0.f;
        vb = vb > 0.f ? vb : 0.f;
        ll_store(ph4 + (e & 1) * 512 + row4a, va, tag);
        ll_store(ph4 + (e & 1) * 512 + row4b, vb, tag);
        if (a.h4_plain) { a.h4_plain[row4a] = va; a.h4_plain[row4b] = vb; }
      }
      __syncthreads();   // the input vector (`red`) is free again
    }
  }
}

constexpr size_t kPersistLdsBytes = ((size_t)4 * V2Tile<VG1, 1>::CS + 8 * 16 * 64 + 8 * (VG3::OC * VG3::P)) * sizeof(float);

__global__ void __launch_bounds__(512, 2) actor_persist_kernel(const ActorPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float s_lut[256];
  __shared__ float s_h4[512];
  __shared__ float s_q[64];
  __shared__ unsigned s_prm[128];
  const int b = blockIdx.x;
  if (b < kPC3) actor_persist_body<kPR_C1 | kPR_C3>(a, lds, s_lut, s_h4, s_q, s_prm);
  else if (b < kPC1) actor_persist_body<kPR_C1>(a, lds, s_lut, s_h4, s_q, s_prm);
  else if (b < kPC1 + kPC2) actor_persist_body<kPR_C2>(a, lds, s_lut, s_h4, s_q, s_prm);
  else if (b == kPEnvWg) actor_persist_body<kPR_ENV>(a, lds, s_lut, s_h4, s_q, s_prm);
  else actor_persist_body<0>(a, lds, s_lut, s_h4, s_q, s_prm);
}

// Library-internal: one launch = n_env env steps of the ring actor on one parameter copy (VanillaNet head).
static int launch_actor_persist(const ActorPersistArgs& a, hipStream_t st) {
  if (a.n_env < 1 || a.n_env > 8 || a.n_actions < 1 || a.n_actions > 64) return DRA_EINVAL;
  static_assert(kPersistLdsBytes + 4096 <= 160 * 1024, "LDS per workgroup (one workgroup per CU)");
  static DraLdsAttr lds_attr;
  if (int rc = dra_grant_lds(lds_attr, reinterpret_cast<const void*>(&actor_persist_kernel), kPersistLdsBytes)) return rc;
  hipLaunchKernelGGL(actor_persist_kernel, dim3(kPersistWgs), dim3(512), kPersistLdsBytes, st, a);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
