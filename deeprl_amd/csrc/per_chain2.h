// PrioritizedReplay.sample() on the device (include/deeprl_amd.h dra_per_chain2_io): the body of dra_sumtree_per_chain2 as a
// device function over NT threads, so that it runs either as its own single-workgroup launch (sumtree.hip, NT = 1024) or as
// a ROLE riding in one of the update's backward launches (fused.hip ChainRole, NT = 256, minibatches up to 256): it is a
// 16 us latency chain of one workgroup that depends on the loss only -- inside the conv3 backward launch it costs the update
// nothing (learner.hip capture_per2).
#pragma once
#include "common.h"

// With the first form the host still sat between an update's priorities and the next update: sync on the loss event,
// validity check, gating, sampling probabilities, indices -> 84 us of host work per step inside the loop, 4.7 k updates/s
// against 8.6 k with uniform replay (tools/diag_per_host.py, profiles/r03h).  Here the kernel does all of it and hands the
// next minibatch to the next update through device memory; the host only generates raw Mersenne-Twister words ahead and
// reads the pinned block one step late (bookkeeping, actor / update hazard check).
struct PerChain2Dev {
  unsigned long long rng_cursor;       // words of the ring consumed so far
  unsigned long long seq;              // launches completed
  int64_t tidx[DRA_PER_CHAIN_MAX];     // leaves of the minibatch the NEXT commit belongs to
  // the launch in two halves (PART 1 / 2 below): what the first half read over PCIe, for the second
  unsigned long long head[9];
  uint32_t w[2 * DRA_PER_CHAIN_MAX];
};

// Stores to the pinned block: system-scope RELAXED atomics (write-through, sc0 sc1) + an explicit wait for their
// acknowledgement.  NOT __threadfence_system() / a system-scope release: on gfx950 those are `buffer_wbl2 sc0 sc1` +
// `buffer_inv sc0 sc1` -- the whole L2 written back and invalidated in the middle of the update (the first form of this
// kernel took 41-66 us that way and slowed the backward pass behind it: profiles/r03h_timeline_per_chain2*.txt).
template <class T> __device__ __forceinline__ void st_sys(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void stores_acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Latency is what this kernel costs (it sits between the loss and the backward pass of every prioritized update): 41 us in its
// first form (three level-by-level walks through L2, 13 header reads and the uniforms over PCIe one after the other:
// profiles/r03h_timeline_per_chain2.txt).  Now:
//   * header (9 quadwords) and uniforms (2 words per lane) are read by parallel lanes up front: one PCIe round trip;
//   * commits and adds are delta propagation with f64 atomics, all ancestors of all leaves in flight together (exact
//     whenever the parallel commit is: `ordered` below; otherwise the level-by-level walks through memory remain).  A climb
//     in registers / LDS with pairwise exchange where two paths meet was tried first: 1.3 us per level, 25 us;
//   * the descent reads the top 11 levels from an LDS copy.
#ifdef DRA_TRACE
// measurement build: thread 0 leaves s_memrealtime stamps (100 MHz) in the unused tail of out_raw_idx (tools/diag_chain2.py)
#define CHAIN2_STAMP(k)                                                                              \
  do {                                                                                               \
    if (threadIdx.x == 0) {                                                                          \
      unsigned long long t_;                                                                         \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");             \
      st_sys(&a.io->out_raw_idx[DRA_PER_CHAIN_MAX - 16 + (k)], (int64_t)t_);                           \
    }                                                                                                \
  } while (0)
#else
#define CHAIN2_STAMP(k) ((void)0)
#endif
constexpr int kTopNodes = 2047;


// smallest j < n with arr[j] == key, or -1.  arr: 16-byte aligned LDS, readable up to the next multiple of 8 entries.  Eight
// entries per trip as four 128-bit reads in flight together: a one-entry-per-trip loop pays a full LDS round trip per entry
// (2.5 us per tree level for 36 entries: tools/diag_chain2.py on the first form of the climb).
__device__ __forceinline__ int lds_find(const int64_t* arr, int n, int64_t key) {
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  int fj = -1;
  for (int j0 = 0; j0 < n; j0 += 8) {
    const ll2* q = reinterpret_cast<const ll2*>(arr + j0);
    const ll2 a = q[0], b = q[1], c = q[2], d = q[3];
    int m = -1;
    m = (d.y == key && j0 + 7 < n) ? j0 + 7 : m;
    m = (d.x == key && j0 + 6 < n) ? j0 + 6 : m;
    m = (c.y == key && j0 + 5 < n) ? j0 + 5 : m;
    m = (c.x == key && j0 + 4 < n) ? j0 + 4 : m;
    m = (b.y == key && j0 + 3 < n) ? j0 + 3 : m;
    m = (b.x == key && j0 + 2 < n) ? j0 + 2 : m;
    m = (a.y == key && j0 + 1 < n) ? j0 + 1 : m;
    m = (a.x == key) ? j0 : m;
    fj = (fj < 0) ? m : fj;
  }
  return fj;
}

// L1-bypassing accessors (as sumtree.hip's node_load / node_store)
__device__ __forceinline__ double chain_node_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_node_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// everything one launch needs (constant across launches of one rotation slot: captured into the update graph)
struct PerChain2Args {
  double* tree;
  int levels, nb;
  int64_t capacity, n_nodes;
  dra_per_chain2_io* io;
  const float* loss_vec;
  float eps, alpha;
  float* prio_out;
  double* stat;
  PerChain2Dev* dev;
  const uint32_t* words;
  int64_t* idx_out;
  float* samp_prob;
  float* weights_out;
};
// library-internal (sumtree.hip): fills `out` for `tree`; DRA_EINVAL as dra_sumtree_per_chain2
int dra_sumtree_per_chain2_args(dra_sumtree* tree, dra_per_chain2_io* io_pinned, const float* loss_vec_dev, float replay_eps,
                                float replay_alpha, float* prio_out_dev, double* stat_dev, void* dev_state,
                                const uint32_t* rng_words_pinned, int64_t* idx_out_dev, float* samp_prob_dev,
                                float* weights_out_dev, int batch, PerChain2Args* out);

template <int NT>
constexpr int per_chain2_lds_bytes() { return (16 + 16 + 12) * 8 + (2 * NT + 8) * 8 + (kTopNodes + 1) * 8 + 16 * 4 + 4 * 4 + NT; }

// PART 0: the whole launch.  PART 1 / 2: its two halves as roles of two DIFFERENT launches of the same update -- 1 = priorities,
// {max, min}, commits and adds (rides in conv3's backward launch), 2 = descent, valid_index filter, padding, hand-over (rides in
// conv1's weight-gradient launch): each half is shorter than the launch that carries it (8-9 us against 10-12), the whole
// (16 us) was not.  The second half reads header and uniforms from the copy the first left in device memory.
template <int NT, int PART = 0>
__device__ __forceinline__ void per_chain2_body(const PerChain2Args& a, char* smem) {
  // ---- LDS carve-out (NT threads, at most NT transitions per minibatch)
  double* s_hi = reinterpret_cast<double*>(smem);
  double* s_lo = s_hi + 16;
  unsigned long long* s_head = reinterpret_cast<unsigned long long*>(s_lo + 16);     // 9 used, 10 reserved
  double& s_max = *reinterpret_cast<double*>(s_head + 10);
  unsigned long long& s_cursor = *(s_head + 11);
  int64_t* s_idx = reinterpret_cast<int64_t*>(s_head + 12);                            // NT + 8, 16-byte aligned
  double* s_p = reinterpret_cast<double*>(s_idx + NT + 8);
  double* s_top = s_p + NT;                                                            // kTopNodes + 1, 16-byte aligned
  float* s_wmax = reinterpret_cast<float*>(s_top + kTopNodes + 1);
  int* s_ints = reinterpret_cast<int*>(s_wmax + 16);
  int& s_ordered = s_ints[0];
  int& s_all_valid = s_ints[1];
  int& s_nvalid = s_ints[2];
  int& s_flags = s_ints[3];
  unsigned char* s_first = reinterpret_cast<unsigned char*>(s_ints + 4);
  double* const tree = a.tree;
  const int levels = a.levels, nb = a.nb;
  const int64_t capacity = a.capacity, n_nodes = a.n_nodes;
  const float* const loss_vec = a.loss_vec;
  const float eps = a.eps, alpha = a.alpha;
  float* const prio_out = a.prio_out;
  double* const stat = a.stat;
  const uint32_t* const words = a.words;
  int64_t* const idx_out = a.idx_out;
  float* const samp_prob = a.samp_prob;
  float* const weights_out = a.weights_out;
  const int tid = threadIdx.x;
  CHAIN2_STAMP(0);
  // ---- everything that crosses PCIe, at once
  const unsigned long long cur0 = a.dev->rng_cursor;
  uint32_t w0 = 0, w1 = 0;
  if constexpr (PART == 2) {
    if (tid < nb) { w0 = a.dev->w[2 * tid]; w1 = a.dev->w[2 * tid + 1]; }
    if (tid < 9) s_head[tid] = a.dev->head[tid];
  } else {
    if (tid < nb) {
      w0 = words[(cur0 + 2ull * tid) & (DRA_PER_RNG_WORDS - 1)];
      w1 = words[(cur0 + 2ull * tid + 1) & (DRA_PER_RNG_WORDS - 1)];
    }
    if (tid < 9) s_head[tid] = reinterpret_cast<const unsigned long long*>(a.io)[tid];
  }
  if (tid == 0) { s_all_valid = 1; s_flags = 0; }
  const int batch = nb;            // (the learner's batch size: every update commits and draws `nb` transitions)
  // ---- commit: {max, min} over every offered priority; a leaf is written by its FIRST occurrence in the minibatch
  // (DQN_agent.py:121-123: priorities = |loss| + eps to the power alpha, from the PRE-weight loss vector; float arithmetic
  // exactly as losses.hip's td_loss_kernel / per_kernel)
  double hi = -INFINITY, lo = INFINITY;
  float prio_f = 0.f;
  if constexpr (PART != 2) {
  if (tid < batch) {
    const float ad = fabsf(loss_vec[tid]) + eps;
    prio_f = (alpha == 0.5f) ? sqrtf(ad) : powf(ad, alpha);
    prio_out[tid] = prio_f;
    const double v = (double)prio_f;
    hi = v;
    lo = v;
    s_idx[tid] = a.dev->tidx[tid];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmax(hi, __shfl_xor(hi, off));
    lo = fmin(lo, __shfl_xor(lo, off));
  }
  if ((tid & 63) == 0) { s_hi[tid >> 6] = hi; s_lo[tid >> 6] = lo; }
  }   // PART != 2
  __syncthreads();
  CHAIN2_STAMP(1);
  const int add_n = (int)(s_head[0] & 0xffffffffull);
  const int force = (int)(s_head[1] >> 32);
  const int hist = (int)(s_head[2] & 0xffffffffull), nstep = (int)(s_head[2] >> 32);
  const int64_t write0 = (int64_t)s_head[3], mem = (int64_t)s_head[4], pos = (int64_t)s_head[5], size = (int64_t)s_head[6];
  const unsigned long long produced = s_head[7];
  if constexpr (PART != 2) {
  if (tid < batch) s_first[tid] = lds_find(s_idx, batch, s_idx[tid]) == tid;
  if (tid == 0) {
    const int nw = (int)(NT >> 6);
    for (int w = 1; w < nw; ++w) { hi = fmax(hi, s_hi[w]); lo = fmin(lo, s_lo[w]); }
    hi = fmax(hi, stat[0]);
    lo = fmin(lo, stat[1]);
    stat[0] = hi;
    stat[1] = lo;
    s_max = hi;
    int ordered = force;
    if (!(lo > 0.0) || !(hi < INFINITY)) ordered = 1;
    else if ((double)capacity * hi > ldexp(1.0, 53 + ilogb(lo) - 23)) ordered = 1;
    s_ordered = ordered;
  }
  __syncthreads();
  CHAIN2_STAMP(2);
  const int n_items = batch + add_n;
  if (!s_ordered && n_items <= NT) {
    // ---- commits and adds as the reference does them -- tree[ancestor] += (new - old) (sum_tree.py:46-60) -- but all at once:
    // one lane per written leaf, one fire-and-forget f64 atomic per ancestor.  `ordered == 0` means every priority is a
    // multiple of one quantum and the total stays below 2^53 quanta: every partial sum is exact, so the additions commute
    // BIT FOR BIT and the result equals the sequential walk's.  (An add on a leaf this minibatch also commits: the add's
    // value stands, as in the reference's order.)  One memory round trip instead of a level-by-level climb (25 us).
    int64_t node = -1;
    double val = 0.0;
    if (tid < batch) {
      if (s_first[tid]) { node = s_idx[tid]; val = (double)prio_f; }
    } else if (tid < n_items) {
      node = (write0 + (tid - batch)) % capacity + capacity - 1;
      val = s_max;
    }
    int64_t* s_add = reinterpret_cast<int64_t*>(s_top);   // (the tree's top is staged later)
    if (tid >= batch && tid < n_items) s_add[tid - batch] = node;
    __syncthreads();
    if (tid < batch && node >= 0 && lds_find(s_add, add_n, node) >= 0) node = -1;
    if (node >= 0) {
      const double delta = __dsub_rn(val, chain_node_load(tree + node));
      chain_node_store(tree + node, val);
      for (int64_t n = node; n > 0;) {
        n = (n - 1) >> 1;
        unsafeAtomicAdd(tree + n, delta);
      }
    }
    CHAIN2_STAMP(3);
  } else {
    if (s_ordered) {
      if (tid == 0) {
        for (int k = 0; k < batch; ++k) {
          if (!s_first[k]) continue;
          int64_t node = s_idx[k];
          const double p = (double)prio_out[k];
          const double change = __dsub_rn(p, chain_node_load(tree + node));
          chain_node_store(tree + node, p);
          while (node > 0) {
            node = (node - 1) >> 1;
            chain_node_store(tree + node, __dadd_rn(chain_node_load(tree + node), change));
          }
        }
      }
      __threadfence_block();
    } else {
      int64_t node = -1;
      if (tid < batch && s_first[tid]) {
        node = s_idx[tid];
        chain_node_store(tree + node, (double)prio_f);
      }
      for (int lv = 0; lv < levels; ++lv) {
        __syncthreads();
        if (node > 0) {
          const int64_t parent = (node - 1) >> 1;
          const double s = __dadd_rn(chain_node_load(tree + 2 * parent + 1), chain_node_load(tree + 2 * parent + 2));
          chain_node_store(tree + parent, s);
          node = parent;
        }
      }
    }
    __syncthreads();
    // adds of the next agent step's transitions at max_priority
    int64_t node = -1;
    if (tid < add_n) {
      node = (write0 + tid) % capacity + capacity - 1;
      chain_node_store(tree + node, s_max);
    }
    for (int lv = 0; lv < levels; ++lv) {
      __syncthreads();
      if (node > 0) {
        const int64_t parent = (node - 1) >> 1;
        const double s = __dadd_rn(chain_node_load(tree + 2 * parent + 1), chain_node_load(tree + 2 * parent + 2));
        chain_node_store(tree + parent, s);
        node = parent;
      }
    }
  }
  }   // PART != 2
  if constexpr (PART == 1) {
    // hand the PCIe reads to the second half
    if (tid < nb) { a.dev->w[2 * tid] = w0; a.dev->w[2 * tid + 1] = w1; }
    if (tid < 9) a.dev->head[tid] = s_head[tid];
    return;
  }
  __syncthreads();
  CHAIN2_STAMP(4);
  // ---- stratified descent of the next draw; the top of the tree from LDS
  const int n_top = (int)(n_nodes < (int64_t)kTopNodes ? n_nodes : (int64_t)kTopNodes);
  for (int i = tid; i < n_top; i += NT) s_top[i] = chain_node_load(tree + i);
  __syncthreads();
  CHAIN2_STAMP(5);
  const double total = s_top[0];
  const bool dry = cur0 + 2ull * (unsigned long long)nb > produced;
  if (tid < nb) {
    double u = 0.0;
    if (!dry) u = ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) * (1.0 / 9007199254740992.0);
    const double seg = __ddiv_rn(total, (double)nb);
    const double sa = __dmul_rn(seg, (double)tid);
    const double sb = __dmul_rn(seg, (double)(tid + 1));
    double s = __dadd_rn(sa, __dmul_rn(__dsub_rn(sb, sa), u));
    int64_t idx = 0;
    while (true) {
      const int64_t left = 2 * idx + 1;
      if (left >= n_nodes) break;
      const double lv = left < n_top ? s_top[left] : chain_node_load(tree + left);
      if (s <= lv) idx = left;
      else { idx = left + 1; s = __dsub_rn(s, lv); }
    }
    s_idx[tid] = idx;
    s_p[tid] = idx < n_top ? s_top[idx] : chain_node_load(tree + idx);
    st_sys(&a.io->out_raw_idx[tid], idx);
    // replay.py:122-127
    const int64_t di = idx - (mem - 1), flo = di - hist + 1, fhi = di + nstep;
    const bool valid = (flo >= 0 && fhi < pos) || (flo >= pos && fhi < size);
    s_first[tid] = valid;
    if (!valid) s_all_valid = 0;
  }
  __syncthreads();
  CHAIN2_STAMP(6);
  if (tid == 0) {
    unsigned long long cur = cur0 + 2ull * (unsigned long long)nb;
    int flags = dry ? 1 : 0;
    int n = nb;
    if (!s_all_valid) {
      // the rare path: drop the invalid draws (order kept), then random.choice over what has been picked so far
      n = 0;
      for (int i = 0; i < nb; ++i)
        if (s_first[i]) { s_idx[n] = s_idx[i]; s_p[n] = s_p[i]; ++n; }
      s_nvalid = n;
      if (n == 0) flags |= 2;
      while (n > 0 && n < nb) {
        const int k = 32 - __clz(n);                 // n.bit_length()
        uint32_t r;
        do {
          if (cur >= produced) { flags |= 1; r = 0; break; }
          r = words[cur & (DRA_PER_RNG_WORDS - 1)] >> (32 - k);
          ++cur;
        } while (r >= (uint32_t)n);
        s_idx[n] = s_idx[r];
        s_p[n] = s_p[r];
        ++n;
      }
      for (; n < nb; ++n) { s_idx[n] = mem - 1 + hist; s_p[n] = 0.0; }   // flags & 2: keep the indices in range
    } else {
      s_nvalid = nb;
    }
    s_cursor = cur;
    s_flags = flags;
  }
  __syncthreads();
  CHAIN2_STAMP(7);
  // ---- hand-over: the next update reads idx_out / samp_prob / weights_out, the next launch of this kernel reads a.dev->tidx
  float beta;
  {
    const unsigned lo32 = (unsigned)(s_head[8] & 0xffffffffull);
    __builtin_memcpy(&beta, &lo32, sizeof(beta));
  }
  // DQN_agent.py:124-126, as per_kernel: weights = (P * B + 1e-6)^-beta / their max
  float wraw = -INFINITY, spf = 0.f;
  if (tid < nb) {
    spf = (float)__ddiv_rn(s_p[tid], total);
    wraw = powf(spf * (float)nb + 1e-6f, -beta);
  }
  {
    const float wm = wave_max(wraw);
    if ((tid & 63) == 0) s_wmax[tid >> 6] = wm;
  }
  __syncthreads();
  if (tid < nb) {
    float wmax = s_wmax[0];
    for (int w = 1; w < (int)(NT >> 6); ++w) wmax = fmaxf(wmax, s_wmax[w]);
    weights_out[tid] = wraw / wmax;
    const int64_t leaf = s_idx[tid];
    const double p = s_p[tid];
    a.dev->tidx[tid] = leaf;
    idx_out[tid] = leaf - (mem - 1);
    samp_prob[tid] = spf;
    st_sys(&a.io->out_idx[tid], leaf);
    st_sys(&a.io->out_p[tid], p);
    stores_acknowledged();
  }
  if (tid == 0) {
    samp_prob[nb] = beta;
    a.dev->rng_cursor = s_cursor;
    st_sys(&a.io->out_total, total);
    st_sys(&a.io->out_n_valid, (int32_t)s_nvalid);
    st_sys(&a.io->out_flags, (int32_t)s_flags);
    st_sys(&a.io->out_rng_cursor, (uint64_t)s_cursor);
    stores_acknowledged();
  }
  __syncthreads();
  CHAIN2_STAMP(8);
  if (tid == 0) {
    const unsigned long long seq = a.dev->seq + 1;
    a.dev->seq = seq;
    st_sys(&a.io->out_seq, (uint64_t)seq);
  }
}
