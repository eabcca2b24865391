// HBM-resident replay ring: slot writes, frame-stack + n-step gather, image normaliser.
// Replaces deep_rl/component/replay.py:75-90 (feed), :112-140 (construct_transition)
// and deep_rl/utils/normalizer.py:58-61 + torch_utils.py:20-25 (uint8 -> f32).
//
// Layout (all in HBM, owned by the handle):
//   frames  u8 [capacity][frame_bytes]   slot-major; a sample's (H+n) frames are ONE
//                                        contiguous run (replay.py:105-110 guarantees the
//                                        run never straddles the write head or the end)
//   actions u8 [capacity][action_bytes]  opaque action record (int64 for discrete agents)
//   rewards f64[capacity]                reference keeps python floats (fp64)
//   masks   i32[capacity]                1 - done
// pos/size bookkeeping and the RNG stay on the host (deeprl_amd/component/replay.py) so
// the index stream is the reference's own np.random stream.
#include "common.h"
#include <new>
#include <stdlib.h>
#include <string.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct dra_ring {
  int64_t capacity, frame_bytes, action_bytes;
  int history, n_step;
  double discount;
  uint8_t* frames;
  uint8_t* actions;
  double* rewards;
  int32_t* masks;
  // pinned staging for host-side feeds (device-accessible host memory)
  uint8_t* stage;
  int64_t stage_slot_bytes;
  int stage_slots, stage_next;
  hipEvent_t stage_wrap;
  bool stage_wrap_pending;
};

static constexpr int kStageSlots = 64;

DRA_API int dra_ring_create(dra_ring** out, int64_t capacity, int64_t frame_bytes, int64_t action_bytes,
                            int history, int n_step, double discount) {
  if (!out || capacity <= 0 || frame_bytes <= 0 || action_bytes <= 0 || history < 1 || n_step < 1) return DRA_EINVAL;
  dra_ring* r = new (std::nothrow) dra_ring();
  if (!r) return DRA_ENOMEM;
  memset(r, 0, sizeof(*r));
  r->capacity = capacity; r->frame_bytes = frame_bytes; r->action_bytes = action_bytes;
  r->history = history; r->n_step = n_step; r->discount = discount;
  hipError_t e;
  if ((e = hipMalloc(&r->frames, (size_t)capacity * frame_bytes)) != hipSuccess) { delete r; return (int)e; }
  if ((e = hipMalloc(&r->actions, (size_t)capacity * action_bytes)) != hipSuccess) { hipFree(r->frames); delete r; return (int)e; }
  if ((e = hipMalloc(&r->rewards, (size_t)capacity * sizeof(double))) != hipSuccess) { hipFree(r->frames); hipFree(r->actions); delete r; return (int)e; }
  if ((e = hipMalloc(&r->masks, (size_t)capacity * sizeof(int32_t))) != hipSuccess) { hipFree(r->frames); hipFree(r->actions); hipFree(r->rewards); delete r; return (int)e; }
  // [frame | action | reward f64 | mask i32], padded to 16 B
  r->stage_slot_bytes = ((frame_bytes + 15) / 16) * 16 + ((action_bytes + 15) / 16) * 16 + 16;
  r->stage_slots = kStageSlots;
  if ((e = hipHostMalloc(&r->stage, (size_t)r->stage_slot_bytes * r->stage_slots, hipHostMallocDefault)) != hipSuccess) {
    hipFree(r->frames); hipFree(r->actions); hipFree(r->rewards); hipFree(r->masks); delete r; return (int)e;
  }
  hipEventCreateWithFlags(&r->stage_wrap, hipEventDisableTiming);
  *out = r;
  return DRA_OK;
}

DRA_API int dra_ring_destroy(dra_ring* r) {
  if (!r) return DRA_OK;
  hipFree(r->frames); hipFree(r->actions); hipFree(r->rewards); hipFree(r->masks);
  hipHostFree(r->stage);
  hipEventDestroy(r->stage_wrap);
  delete r;
  return DRA_OK;
}

// Raw device pointers, for zero-copy consumers (fused learner) and tests.
DRA_API int dra_ring_shape(dra_ring* r, int* history, int* n_step) {
  if (!r || !history || !n_step) return DRA_EINVAL;
  *history = r->history;
  *n_step = r->n_step;
  return DRA_OK;
}

DRA_API int dra_ring_discount(dra_ring* r, double* discount) {
  if (!r || !discount) return DRA_EINVAL;
  *discount = r->discount;
  return DRA_OK;
}

DRA_API int dra_ring_pointers(dra_ring* r, void** frames, void** actions, void** rewards, void** masks) {
  if (!r) return DRA_EINVAL;
  if (frames) *frames = r->frames;
  if (actions) *actions = r->actions;
  if (rewards) *rewards = r->rewards;
  if (masks) *masks = r->masks;
  return DRA_OK;
}

// ---------------------------------------------------------------------------------------------
// put: write `count` consecutive slots [slot0, slot0+count) from device-accessible sources.
// One workgroup per slot; 16-byte vector path when sizes/alignment allow, byte path otherwise.
__global__ void __launch_bounds__(256)
ring_put_kernel(uint8_t* __restrict__ frames, uint8_t* __restrict__ actions, double* __restrict__ rewards,
                int32_t* __restrict__ masks, int64_t frame_bytes, int64_t action_bytes, int64_t slot0,
                const uint8_t* __restrict__ fsrc, const uint8_t* __restrict__ asrc, const double* __restrict__ rsrc,
                const int32_t* __restrict__ msrc, int64_t action_val, double reward_val, int32_t mask_val, int vec16) {
  const int64_t k = blockIdx.x;
  const int64_t slot = slot0 + k;
  uint8_t* dst = frames + slot * frame_bytes;
  const uint8_t* src = fsrc + k * frame_bytes;
  if (vec16) {
    const int64_t nv = frame_bytes >> 4;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (int64_t i = threadIdx.x; i < nv; i += blockDim.x) d4[i] = s4[i];
  } else {
    for (int64_t i = threadIdx.x; i < frame_bytes; i += blockDim.x) dst[i] = src[i];
  }
  if (threadIdx.x < action_bytes) {
    uint8_t v;
    if (asrc) v = asrc[k * action_bytes + threadIdx.x];
    else v = (threadIdx.x < 8) ? (uint8_t)((uint64_t)action_val >> (8 * threadIdx.x)) : 0;
    actions[slot * action_bytes + threadIdx.x] = v;
  }
  if (threadIdx.x == 0) {
    rewards[slot] = rsrc ? rsrc[k] : reward_val;
    masks[slot] = msrc ? msrc[k] : mask_val;
  }
}

static inline int aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

DRA_API int dra_ring_put(dra_ring* r, int64_t slot0, int64_t count, const void* frame_src, const void* action_src,
                         int64_t action_val, const double* reward_src, double reward_val, const int32_t* mask_src,
                         int32_t mask_val, void* stream) {
  if (!r || !frame_src || count <= 0 || slot0 < 0 || slot0 + count > r->capacity) return DRA_EINVAL;
  if (!action_src && r->action_bytes > 8) return DRA_EINVAL;
  int vec16 = (r->frame_bytes % 16 == 0) && aligned16(frame_src) && aligned16(r->frames);
  hipLaunchKernelGGL(ring_put_kernel, dim3((unsigned)count), dim3(256), 0, dra_stream(stream), r->frames, r->actions,
                     r->rewards, r->masks, r->frame_bytes, r->action_bytes, slot0, (const uint8_t*)frame_src,
                     (const uint8_t*)action_src, reward_src, mask_src, action_val, reward_val, mask_val, vec16);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Host-side feed (replay.py:75-90 for one env): stage through pinned memory, then the put kernel
// reads the staging slot over the host link.  Never blocks except once per staging wrap.
DRA_API int dra_ring_put_host(dra_ring* r, int64_t slot, const void* frame_host, const void* action_host,
                              double reward, int32_t mask, void* stream) {
  if (!r || !frame_host || !action_host || slot < 0 || slot >= r->capacity) return DRA_EINVAL;
  if (r->stage_next == 0 && r->stage_wrap_pending) {
    DRA_HIP(hipEventSynchronize(r->stage_wrap));
    r->stage_wrap_pending = false;
  }
  uint8_t* s = r->stage + (size_t)r->stage_next * r->stage_slot_bytes;
  const int64_t aoff = ((r->frame_bytes + 15) / 16) * 16;
  const int64_t roff = aoff + ((r->action_bytes + 15) / 16) * 16;
  memcpy(s, frame_host, (size_t)r->frame_bytes);
  memcpy(s + aoff, action_host, (size_t)r->action_bytes);
  memcpy(s + roff, &reward, sizeof(double));
  memcpy(s + roff + 8, &mask, sizeof(int32_t));
  int vec16 = (r->frame_bytes % 16 == 0);
  hipLaunchKernelGGL(ring_put_kernel, dim3(1), dim3(256), 0, dra_stream(stream), r->frames, r->actions, r->rewards,
                     r->masks, r->frame_bytes, r->action_bytes, slot, (const uint8_t*)s, (const uint8_t*)(s + aoff),
                     (const double*)(s + roff), (const int32_t*)(s + roff + 8), (int64_t)0, 0.0, 0, vec16);
  DRA_LAUNCH_CHECK();
  r->stage_next = (r->stage_next + 1) % r->stage_slots;
  if (r->stage_next == 0) {
    DRA_HIP(hipEventRecord(r->stage_wrap, dra_stream(stream)));
    r->stage_wrap_pending = true;
  }
  return DRA_OK;
}

// ---------------------------------------------------------------------------------------------
// Synthetic fill (frame k = counter hash so CPU oracle and GPU agree without a 7 GB
// host array).  splitmix64 finaliser over (seed, global 8-byte word index).
__host__ __device__ __forceinline__ uint64_t dra_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256)
ring_fill_synth_kernel(uint8_t* __restrict__ frames, uint8_t* __restrict__ actions, double* __restrict__ rewards,
                       int32_t* __restrict__ masks, int64_t frame_bytes, int64_t action_bytes, int64_t slot0,
                       int64_t counter0, uint64_t seed, int n_actions, int done_period) {
  const int64_t k = blockIdx.x;
  const int64_t slot = slot0 + k;
  const uint64_t ctr = (uint64_t)(counter0 + k);
  const int64_t words = frame_bytes >> 3;  // frame_bytes % 8 == 0 enforced by the caller
  uint64_t* dst = reinterpret_cast<uint64_t*>(frames + slot * frame_bytes);
  const uint64_t base = seed * 0x9E3779B97F4A7C15ull + ctr * (uint64_t)words;
  for (int64_t w = threadIdx.x; w < words; w += blockDim.x) dst[w] = dra_mix64(base + (uint64_t)w);
  if (threadIdx.x == 0) {
    const uint64_t h = dra_mix64((seed + 1) * 0x9E3779B97F4A7C15ull + ctr);
    const int64_t a = (int64_t)((h & 0xffffffffull) % (uint64_t)n_actions);
    const uint32_t u = (uint32_t)(h >> 32) % 10u;          // reward: -1 (p=.1), 0 (p=.8), +1 (p=.1)
    const double rew = (u == 0) ? -1.0 : ((u == 9) ? 1.0 : 0.0);
    const uint64_t h2 = dra_mix64((seed + 2) * 0x9E3779B97F4A7C15ull + ctr);
    const int32_t m = ((h2 % (uint64_t)done_period) == 0) ? 0 : 1;
    for (int b = 0; b < action_bytes; ++b) actions[slot * action_bytes + b] = (b < 8) ? (uint8_t)((uint64_t)a >> (8 * b)) : 0;
    rewards[slot] = rew;
    masks[slot] = m;
  }
}

DRA_API int dra_ring_fill_synthetic(dra_ring* r, int64_t slot0, int64_t count, int64_t counter0, uint64_t seed,
                                    int n_actions, int done_period, void* stream) {
  if (!r || count <= 0 || slot0 < 0 || slot0 + count > r->capacity || (r->frame_bytes % 8) || n_actions < 1 ||
      done_period < 1)
    return DRA_EINVAL;
  const int64_t chunk = 1 << 20;
  for (int64_t o = 0; o < count; o += chunk) {
    int64_t c = count - o < chunk ? count - o : chunk;
    hipLaunchKernelGGL(ring_fill_synth_kernel, dim3((unsigned)c), dim3(256), 0, dra_stream(stream), r->frames,
                       r->actions, r->rewards, r->masks, r->frame_bytes, r->action_bytes, slot0 + o, counter0 + o, seed,
                       n_actions, done_period);
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

// ---------------------------------------------------------------------------------------------
// gather (K1).  Grid = batch * (H + n) workgroups; workgroup (b, j) loads source frame
// idx[b]-H+1+j ONCE and writes it to state[b][j] (j < H) and next_state[b][j-n] (j >= n): the
// (H-n) frames shared by state and next_state are read once, so HBM reads are the compulsory
// (H+n)*frame_bytes per sample.  Thread 0 of workgroup (b,0) folds the n-step return
// (replay.py:135-139) in fp64 with the reference's association: cum = r + ((m*gamma)*cum).
template <bool VEC16, bool STREAM>
__global__ void __launch_bounds__(256)
ring_gather_kernel(const uint8_t* __restrict__ frames, const uint8_t* __restrict__ actions,
                   const double* __restrict__ rewards, const int32_t* __restrict__ masks,
                   const int64_t* __restrict__ idx, int64_t frame_bytes, int64_t action_bytes, int H, int n,
                   double discount, uint8_t* __restrict__ out_state, uint8_t* __restrict__ out_next,
                   uint8_t* __restrict__ out_action, double* __restrict__ out_reward, int32_t* __restrict__ out_mask,
                   float* __restrict__ out_reward_f32, float* __restrict__ out_mask_f32, const int block) {
  const int span = H + n;
  const int b = blockIdx.x / span;
  const int j = blockIdx.x - b * span;
  DRA_STAMP(TR_GATHER, 0);
  const int64_t i = idx[b];
  const uint8_t* src = frames + (i - H + 1 + j) * frame_bytes;
  // block != 0: out_state is ONE [B][H + n][frame] block, every frame of the run written once (state = frames [0, H), next_state
  // = frames [n, H + n) of the same sample: two views); else the two [B][H][frame] tensors, the H - n shared frames written twice
  uint8_t* d0 = block ? out_state + ((int64_t)b * span + j) * frame_bytes
                      : ((j < H && out_state) ? out_state + ((int64_t)b * H + j) * frame_bytes : nullptr);
  uint8_t* d1 = (!block && j >= n && out_next) ? out_next + ((int64_t)b * H + (j - n)) * frame_bytes : nullptr;
  if (VEC16) {
    // two 16-byte loads per lane in flight before the first store (a 7056-byte frame is 441 vectors: one pass
    // of this loop); STREAM (many-minibatch launches whose output does not fit the caches) also streams the stores
    const int64_t nv = frame_bytes >> 4;
    const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
    u32x4* o0 = reinterpret_cast<u32x4*>(d0);
    u32x4* o1 = reinterpret_cast<u32x4*>(d1);
    for (int64_t t = threadIdx.x; t < nv; t += 2 * blockDim.x) {
      const int64_t t2 = t + blockDim.x;
      const bool second = t2 < nv;
      const u32x4 v = __builtin_nontemporal_load(s4 + t);  // ring frames are streamed once
      const u32x4 w = __builtin_nontemporal_load(s4 + (second ? t2 : t));
      if (STREAM) {
        if (d0) { __builtin_nontemporal_store(v, o0 + t); if (second) __builtin_nontemporal_store(w, o0 + t2); }
        if (d1) { __builtin_nontemporal_store(v, o1 + t); if (second) __builtin_nontemporal_store(w, o1 + t2); }
      } else {
        if (d0) { o0[t] = v; if (second) o0[t2] = w; }
        if (d1) { o1[t] = v; if (second) o1[t2] = w; }
      }
    }
  } else {
    for (int64_t t = threadIdx.x; t < frame_bytes; t += blockDim.x) {
      const uint8_t v = src[t];
      if (d0) d0[t] = v;
      if (d1) d1[t] = v;
    }
  }
  if (j == 0) {
    if (out_action && threadIdx.x < action_bytes)
      out_action[(int64_t)b * action_bytes + threadIdx.x] = actions[i * action_bytes + threadIdx.x];
    if (threadIdx.x == 0) {
      double cum_r = 0.0;
      int32_t cum_m = 1;
      for (int k = n - 1; k >= 0; --k) {
        const int32_t m = masks[i + k];
        // reference: reward[i] + mask[i] * discount * cum_r   (left-assoc, no FMA contraction)
        cum_r = __dadd_rn(rewards[i + k], __dmul_rn(__dmul_rn((double)m, discount), cum_r));
        cum_m = cum_m ? m : cum_m;
      }
      if (out_reward) out_reward[b] = cum_r;
      if (out_mask) out_mask[b] = cum_m;
      if (out_reward_f32) out_reward_f32[b] = (float)cum_r;
      if (out_mask_f32) out_mask_f32[b] = (float)cum_m;
    }
  }
  DRA_STAMP(TR_GATHER, 5);
  DRA_STAMP_END(TR_GATHER);
}

// (A workgroup-per-SAMPLE shape -- one workgroup walking the whole 5-frame run, 3 loads per lane in flight -- was measured
// for many-minibatch launches and LOST to the workgroup-per-frame shape below: 4.90 vs 5.81 TB/s at 1024 minibatches on the
// same box, profiles/r02w_kernel_microbench_gather_ab.json; fewer, longer workgroups leave fewer loads in flight per CU.)
static int ring_gather_launch(dra_ring* r, const int64_t* idx_dev, int batch, void* out_state, void* out_next_state, void* out_action,
                              double* out_reward, int32_t* out_mask, float* out_reward_f32, float* out_mask_f32, int block,
                              void* stream) {
  if (!r || !idx_dev || batch <= 0 || (block && !out_state)) return DRA_EINVAL;
  const int span = r->history + r->n_step;
  const bool vec = (r->frame_bytes % 16 == 0) && aligned16(r->frames) && (!out_state || aligned16(out_state)) &&
                   (!out_next_state || aligned16(out_next_state));
  dim3 grid((unsigned)batch * span), blk(256);
  // an output larger than the 256 MB Infinity Cache cannot stay on die anyway: stream it past the caches
  const bool stream_out = (int64_t)batch * (block ? span : 2 * r->history) * r->frame_bytes >= ((int64_t)256 << 20);
#define DRA_GATHER_LAUNCH(V, S)                                                                                        \
  hipLaunchKernelGGL((ring_gather_kernel<V, S>), grid, blk, 0, dra_stream(stream), r->frames, r->actions, r->rewards,   \
                     r->masks, idx_dev, r->frame_bytes, r->action_bytes, r->history, r->n_step, r->discount,          \
                     (uint8_t*)out_state, (uint8_t*)out_next_state, (uint8_t*)out_action, out_reward, out_mask,       \
                     out_reward_f32, out_mask_f32, block)
  if (vec && stream_out) DRA_GATHER_LAUNCH(true, true);
  else if (vec) DRA_GATHER_LAUNCH(true, false);
  else DRA_GATHER_LAUNCH(false, false);
#undef DRA_GATHER_LAUNCH
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_ring_gather(dra_ring* r, const int64_t* idx_dev, int batch, void* out_state, void* out_next_state,
                            void* out_action, double* out_reward, int32_t* out_mask, float* out_reward_f32,
                            float* out_mask_f32, void* stream) {
  return ring_gather_launch(r, idx_dev, batch, out_state, out_next_state, out_action, out_reward, out_mask, out_reward_f32,
                            out_mask_f32, 0, stream);
}

// state / next_state as two views of ONE [batch][history + n_step][frame] block (replay.py:112-140 stacks the same frames twice:
// the history - n_step shared frames of a sample are written once here -- 1.13 instead of 1.81 MB per DQN minibatch)
DRA_API int dra_ring_gather_block(dra_ring* r, const int64_t* idx_dev, int batch, void* out_block, void* out_action,
                                  double* out_reward, int32_t* out_mask, float* out_reward_f32, float* out_mask_f32, void* stream) {
  return ring_gather_launch(r, idx_dev, batch, out_block, nullptr, out_action, out_reward, out_mask, out_reward_f32, out_mask_f32, 1,
                            stream);
}

// ---------------------------------------------------------------------------------------------
// uint8 -> f32 through a 256-entry table.  The table holds f32(f64(v) * coef) (the reference's
// sync-replay numerics, normalizer.py:58-61 + torch_utils.py:23) so the result is bit-exact.
__global__ void __launch_bounds__(256)
u8_lut_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n, const float* __restrict__ lut) {
  __shared__ float s_lut[256];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const int64_t n16 = n >> 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint4* in4 = reinterpret_cast<const uint4*>(in);
  float4* out4 = reinterpret_cast<float4*>(out);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n16; t += stride) {
    const uint4 v = in4[t];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 o;
      o.x = s_lut[w[q] & 0xff]; o.y = s_lut[(w[q] >> 8) & 0xff];
      o.z = s_lut[(w[q] >> 16) & 0xff]; o.w = s_lut[w[q] >> 24];
      out4[t * 4 + q] = o;
    }
  }
  for (int64_t t = (n16 << 4) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) out[t] = s_lut[in[t]];
}

// The same table for rows that are contiguous inside but in_row_stride bytes apart (the state / next_state views of
// dra_ring_gather_block): out is dense [n_rows][row_elems].  row_elems % 16 == 0, 16-byte aligned rows.
__global__ void __launch_bounds__(256)
u8_lut_rows_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n_rows, int64_t row_elems, int64_t in_row_stride,
                   const float* __restrict__ lut) {
  __shared__ float s_lut[256];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const int64_t v_per_row = row_elems >> 4, total = n_rows * v_per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float4* out4 = reinterpret_cast<float4*>(out);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t row = t / v_per_row, c = t - row * v_per_row;
    const uint4 v = *reinterpret_cast<const uint4*>(in + row * in_row_stride + (c << 4));
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 o;
      o.x = s_lut[w[q] & 0xff]; o.y = s_lut[(w[q] >> 8) & 0xff];
      o.z = s_lut[(w[q] >> 16) & 0xff]; o.w = s_lut[w[q] >> 24];
      out4[t * 4 + q] = o;
    }
  }
}

DRA_API int dra_u8_to_f32_lut_rows(const void* in_u8, float* out, int64_t n_rows, int64_t row_elems, int64_t in_row_stride,
                                   const float* lut256_dev, void* stream) {
  if (!in_u8 || !out || !lut256_dev || n_rows < 1 || row_elems < 16 || (row_elems & 15) || in_row_stride < row_elems ||
      (in_row_stride & 15) || ((((uintptr_t)in_u8) | ((uintptr_t)out)) & 15))
    return DRA_EINVAL;
  int64_t b = (n_rows * (row_elems >> 4) + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(u8_lut_rows_kernel, dim3((unsigned)b), dim3(256), 0, dra_stream(stream), (const uint8_t*)in_u8, out, n_rows,
                     row_elems, in_row_stride, lut256_dev);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// ------------------------------------------------------------------------------------------------
// Minibatch rows of an on-policy rollout (PPO_agent.py:77-80: `entry = entries[batch_indices]` over state / action / log_pi_a /
// ret / advantage): up to DRA_GATHER_MAX_FIELDS row-major arrays gathered by ONE index vector in ONE launch -- ATen's x[idx]
// is a launch per field (four index kernels + one vectorised gather, 24 us of a 330 us ppo_pixel minibatch,
// profiles/r05s_kernel_stats_ppo_pixel_8.txt).  Workgroup (r, c): chunk c of row idx[r] of every field that has one
// (16-byte vectors when the row size and both pointers allow, else bytes); grid (n_rows, max chunks), a chunk = 4 KB.
struct GatherFields {
  const uint8_t* src[DRA_GATHER_MAX_FIELDS];
  uint8_t* dst[DRA_GATHER_MAX_FIELDS];
  int64_t row_bytes[DRA_GATHER_MAX_FIELDS];
  int32_t vec16[DRA_GATHER_MAX_FIELDS];
  int32_t n_fields;
};
constexpr int kGatherChunk = 4096;
__global__ void __launch_bounds__(256)
gather_rows_kernel(const GatherFields f, const int64_t* __restrict__ idx, int64_t n_src_rows) {
  const int r = blockIdx.x;
  const int64_t c0 = (int64_t)blockIdx.y * kGatherChunk;
  int64_t i = idx[r];
  if (i < 0) i += n_src_rows;                     // (torch indexing semantics for negative indices)
  for (int q = 0; q < f.n_fields; ++q) {
    const int64_t rb = f.row_bytes[q];
    if (c0 >= rb) continue;
    const int64_t n = min((int64_t)kGatherChunk, rb - c0);
    const uint8_t* s_ = f.src[q] + i * rb + c0;
    uint8_t* d_ = f.dst[q] + (int64_t)r * rb + c0;
    if (f.vec16[q]) {
      const u32x4* s4 = reinterpret_cast<const u32x4*>(s_);
      u32x4* d4 = reinterpret_cast<u32x4*>(d_);
      const int t = threadIdx.x;                  // a chunk is exactly 256 vectors
      if (16 * (int64_t)t < n) d4[t] = s4[t];
    } else {
      for (int64_t t = threadIdx.x; t < n; t += 256) d_[t] = s_[t];
    }
  }
}

DRA_API int dra_gather_rows(int n_fields, const void* const* src, void* const* dst, const int64_t* row_bytes, const int64_t* idx_dev,
                            int n_rows, int64_t n_src_rows, void* stream) {
  if (n_fields < 1 || n_fields > DRA_GATHER_MAX_FIELDS || !src || !dst || !row_bytes || !idx_dev || n_rows < 1 || n_src_rows < 1)
    return DRA_EINVAL;
  GatherFields f;
  memset(&f, 0, sizeof(f));
  int64_t max_rb = 0;
  for (int q = 0; q < n_fields; ++q) {
    if (!src[q] || !dst[q] || row_bytes[q] < 1) return DRA_EINVAL;
    f.src[q] = (const uint8_t*)src[q]; f.dst[q] = (uint8_t*)dst[q]; f.row_bytes[q] = row_bytes[q];
    f.vec16[q] = ((row_bytes[q] & 15) == 0 && ((((uintptr_t)src[q]) | ((uintptr_t)dst[q])) & 15) == 0) ? 1 : 0;
    if (row_bytes[q] > max_rb) max_rb = row_bytes[q];
  }
  f.n_fields = n_fields;
  const int64_t chunks = (max_rb + kGatherChunk - 1) / kGatherChunk;
  if (chunks > 65535) return DRA_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_rows, (unsigned)chunks), dim3(256), 0, dra_stream(stream), f, idx_dev, n_src_rows);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

DRA_API int dra_u8_to_f32_lut(const void* in_u8, float* out, int64_t n, const float* lut256_dev, void* stream) {
  if (!in_u8 || !out || !lut256_dev || n < 0) return DRA_EINVAL;
  if (n == 0) return DRA_OK;
  if (!aligned16(in_u8) || !aligned16(out)) return DRA_EINVAL;
  int64_t blocks = ((n >> 4) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;  // grid-stride past 256 CUs x 8
  hipLaunchKernelGGL(u8_lut_kernel, dim3((unsigned)blocks), dim3(256), 0, dra_stream(stream), (const uint8_t*)in_u8, out,
                     n, lut256_dev);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

#ifdef DRA_TRACE
extern "C" int dra_trace_set_ring(void* p) { return dra_trace_set_local(p); }
#endif
