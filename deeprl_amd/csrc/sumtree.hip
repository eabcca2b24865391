// fp64 binary-heap sum tree in HBM (K7 sample / K8 update).
// Replaces deep_rl/utils/sum_tree.py:6-66 as used by deep_rl/component/replay.py:152-196.
//
// Storage is the reference's own heap array: f64[2*cap-1], root 0, children 2i+1 / 2i+2,
// leaves cap-1 .. 2cap-2 (two depths when cap is not a power of two), so the tree is
// inspectable and comparable node by node with the reference's numpy array.
//
// Update modes:
//   parallel  (default) one lane per updated leaf; leaves are written, then every affected
//             ancestor is recomputed as left+right, one level per barrier.  In the regime the
//             reference runs in (fp32-valued priorities summed in fp64) every
//             node sum is exact, hence identical to the reference's incremental `+= change`.
//   ordered   one lane replays the reference's `tree[parent] += change` walk update by update;
//             bit-identical for ANY fp64 priorities, ~50x slower (dependent L2 round trips).
// pending_idx gating, first-writer-wins de-duplication and max_priority live on the host
// mirror (deeprl_amd/component/replay.py), which passes only the effective updates.
#include "common.h"
#include "per_chain2.h"
#include <new>

struct dra_sumtree {
  int64_t capacity;
  int64_t n_nodes;
  int levels;  // depth of the deepest leaf (root = 0)
  double* tree;
};

DRA_API int dra_sumtree_create(dra_sumtree** out, int64_t capacity) {
  if (!out || capacity < 1) return DRA_EINVAL;
  dra_sumtree* t = new (std::nothrow) dra_sumtree();
  if (!t) return DRA_ENOMEM;
  t->capacity = capacity;
  t->n_nodes = 2 * capacity - 1;
  int lv = 0;
  for (int64_t node = t->n_nodes - 1; node > 0; node = (node - 1) / 2) ++lv;
  t->levels = lv;
  hipError_t e = hipMalloc(&t->tree, (size_t)t->n_nodes * sizeof(double));
  if (e != hipSuccess) { delete t; return (int)e; }
  e = hipMemset(t->tree, 0, (size_t)t->n_nodes * sizeof(double));
  if (e != hipSuccess) { (void)hipFree(t->tree); delete t; return (int)e; }
  *out = t;
  return DRA_OK;
}

DRA_API int dra_sumtree_destroy(dra_sumtree* t) {
  if (!t) return DRA_OK;
  (void)hipFree(t->tree);
  delete t;
  return DRA_OK;
}

DRA_API int dra_sumtree_pointer(dra_sumtree* t, void** tree_dev, int64_t* n_nodes) {
  if (!t) return DRA_EINVAL;
  if (tree_dev) *tree_dev = t->tree;
  if (n_nodes) *n_nodes = t->n_nodes;
  return DRA_OK;
}

// L1-bypassing accessors: nodes written by one lane are read by other lanes of the same
// workgroup a barrier later; sc1 loads/stores are served by L2, never a stale L1 line.
__device__ __forceinline__ double node_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void node_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup, one lane per update (n <= 1024).  Leaves must be unique.
__global__ void __launch_bounds__(1024)
sumtree_update_parallel_kernel(double* __restrict__ tree, int levels, const int64_t* __restrict__ leaf,
                               const double* __restrict__ prio, int n) {
  int64_t node = -1;
  if ((int)threadIdx.x < n) {
    node = leaf[threadIdx.x];
    node_store(tree + node, prio[threadIdx.x]);
  }
  for (int lv = 0; lv < levels; ++lv) {
    __syncthreads();  // previous level's stores are in L2 before anyone reads them
    if (node > 0) {
      const int64_t parent = (node - 1) >> 1;
      const double s = __dadd_rn(node_load(tree + 2 * parent + 1), node_load(tree + 2 * parent + 2));
      node_store(tree + parent, s);
      node = parent;
    }
  }
}

// Reference-order replay: sum_tree.py:54-60 + _propagate :16-20, one update after another.
__global__ void sumtree_update_ordered_kernel(double* __restrict__ tree, const int64_t* __restrict__ leaf,
                                              const double* __restrict__ prio, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int k = 0; k < n; ++k) {
    int64_t node = leaf[k];
    const double p = prio[k];
    const double change = __dsub_rn(p, tree[node]);
    tree[node] = p;
    while (node > 0) {
      node = (node - 1) >> 1;
      tree[node] = __dadd_rn(tree[node], change);
    }
  }
}

DRA_API int dra_sumtree_update(dra_sumtree* t, const int64_t* leaf_idx_dev, const double* prio_dev, int n, int ordered,
                               void* stream) {
  if (!t || !leaf_idx_dev || !prio_dev || n < 0 || n > 1024) return DRA_EINVAL;
  if (n == 0) return DRA_OK;
  if (ordered)
    hipLaunchKernelGGL(sumtree_update_ordered_kernel, dim3(1), dim3(64), 0, dra_stream(stream), t->tree, leaf_idx_dev,
                       prio_dev, n);
  else {
    const int threads = ((n + 63) / 64) * 64;
    hipLaunchKernelGGL(sumtree_update_parallel_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), t->tree, t->levels,
                       leaf_idx_dev, prio_dev, n);
  }
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Single-leaf variant with by-value arguments (PrioritizedReplay.feed -> SumTree.add, sum_tree.py:39-51):
// one wave walks leaf -> root recomputing each ancestor from its children.
__global__ void sumtree_set_kernel(double* __restrict__ tree, int64_t leaf, double prio) {
  if (threadIdx.x != 0) return;
  int64_t node = leaf;
  tree[node] = prio;
  double below = prio;
  while (node > 0) {
    const int64_t parent = (node - 1) >> 1;
    const int64_t sib = (node & 1) ? node + 1 : node - 1;  // odd index = left child
    const double s = (node & 1) ? __dadd_rn(below, tree[sib]) : __dadd_rn(tree[sib], below);
    tree[parent] = s;
    below = s;
    node = parent;
  }
}

DRA_API int dra_sumtree_set(dra_sumtree* t, int64_t leaf_idx, double prio, void* stream) {
  if (!t || leaf_idx < t->capacity - 1 || leaf_idx >= t->n_nodes) return DRA_EINVAL;
  hipLaunchKernelGGL(sumtree_set_kernel, dim3(1), dim3(64), 0, dra_stream(stream), t->tree, leaf_idx, prio);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// feed with the priority taken from DEVICE memory (PrioritizedReplay.feed adds new transitions at max_priority,
// replay.py:161; the running maximum lives in stat_dev[0] once the priorities are written back on device)
__global__ void sumtree_set_from_kernel(double* __restrict__ tree, int64_t leaf, const double* __restrict__ prio) {
  if (threadIdx.x != 0) return;
  int64_t node = leaf;
  const double p = *prio;
  tree[node] = p;
  double below = p;
  while (node > 0) {
    const int64_t parent = (node - 1) >> 1;
    const int64_t sib = (node & 1) ? node + 1 : node - 1;
    const double s = (node & 1) ? __dadd_rn(below, tree[sib]) : __dadd_rn(tree[sib], below);
    tree[parent] = s;
    below = s;
    node = parent;
  }
}

DRA_API int dra_sumtree_set_from(dra_sumtree* t, int64_t leaf_idx, const double* prio_dev, void* stream) {
  if (!t || !prio_dev || leaf_idx < t->capacity - 1 || leaf_idx >= t->n_nodes) return DRA_EINVAL;
  hipLaunchKernelGGL(sumtree_set_from_kernel, dim3(1), dim3(64), 0, dra_stream(stream), t->tree, leaf_idx, prio_dev);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// n consecutive adds (write cursor write0, write0+1, ... mod capacity) at the SAME device-resident priority: what
// PrioritizedReplay.feed does for the n transitions a device producer wrote in one agent step.  One lane per leaf,
// ancestors recomputed level by level as in sumtree_update_parallel_kernel (one walk of `levels` barriers instead of n
// dependent leaf-to-root walks of ~20 L2 round trips each).
__global__ void __launch_bounds__(64)
sumtree_set_many_from_kernel(double* __restrict__ tree, int levels, int64_t capacity, int64_t write0, int n,
                             const double* __restrict__ prio) {
  int64_t node = -1;
  if ((int)threadIdx.x < n) {
    node = (write0 + threadIdx.x) % capacity + capacity - 1;
    node_store(tree + node, *prio);
  }
  for (int lv = 0; lv < levels; ++lv) {
    __syncthreads();
    if (node > 0) {
      const int64_t parent = (node - 1) >> 1;
      const double s = __dadd_rn(node_load(tree + 2 * parent + 1), node_load(tree + 2 * parent + 2));
      node_store(tree + parent, s);
      node = parent;
    }
  }
}

DRA_API int dra_sumtree_set_many_from(dra_sumtree* t, int64_t write0, int n, const double* prio_dev, void* stream) {
  if (!t || !prio_dev || write0 < 0 || write0 >= t->capacity || n < 1 || n > 64 || n > t->capacity) return DRA_EINVAL;
  hipLaunchKernelGGL(sumtree_set_many_from_kernel, dim3(1), dim3(64), 0, dra_stream(stream), t->tree, t->levels, t->capacity,
                     write0, n, prio_dev);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Priority write-back of one minibatch without a host round trip (replay.py:193-196 + sum_tree.py:54-60).  The HOST
// decides WHICH leaves are written (pending_idx gating and first-writer-wins need no priority value): leaf[i] gets
// f64(prio_f32[pos[i]]), i < n.  stat[0] = max(stat[0], every offered priority) (replay.py:195: max_priority tracks all
// of them, gated or not), stat[1] = the smallest priority ever offered.  The level-parallel update is exact -- hence
// identical to the reference's incremental `+= change` -- as long as every leaf is a multiple of u = ulp_f32(stat[1])
// and capacity * stat[0] / u <= 2^53; the kernel checks that bound and replays the reference's walk in order otherwise.
__global__ void __launch_bounds__(1024)
sumtree_commit_kernel(double* __restrict__ tree, int levels, int64_t capacity, const int64_t* __restrict__ leaf,
                      const int32_t* __restrict__ pos, int n, const float* __restrict__ prio, int batch,
                      double* __restrict__ stat, int force_ordered) {
  __shared__ double s_hi[16], s_lo[16];
  __shared__ int s_ordered;
  const int tid = threadIdx.x;
  double hi = -INFINITY, lo = INFINITY;
  for (int b = tid; b < batch; b += blockDim.x) {
    const double v = (double)prio[b];
    hi = fmax(hi, v);
    lo = fmin(lo, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmax(hi, __shfl_xor(hi, off));
    lo = fmin(lo, __shfl_xor(lo, off));
  }
  if ((tid & 63) == 0) { s_hi[tid >> 6] = hi; s_lo[tid >> 6] = lo; }
  __syncthreads();
  if (tid == 0) {
    const int nw = (int)(blockDim.x >> 6);
    for (int w = 1; w < nw; ++w) { hi = fmax(hi, s_hi[w]); lo = fmin(lo, s_lo[w]); }
    hi = fmax(hi, stat[0]);
    lo = fmin(lo, stat[1]);
    stat[0] = hi;
    stat[1] = lo;
    int ordered = force_ordered;
    if (!(lo > 0.0) || !(hi < INFINITY)) ordered = 1;
    else if ((double)capacity * hi > ldexp(1.0, 53 + ilogb(lo) - 23)) ordered = 1;
    s_ordered = ordered;
  }
  __syncthreads();
  if (n <= 0) return;
  if (s_ordered) {
    if (tid != 0) return;
    for (int k = 0; k < n; ++k) {
      int64_t node = leaf[k];
      const double p = (double)prio[pos[k]];
      const double change = __dsub_rn(p, tree[node]);
      tree[node] = p;
      while (node > 0) {
        node = (node - 1) >> 1;
        tree[node] = __dadd_rn(tree[node], change);
      }
    }
    return;
  }
  int64_t node = -1;
  if (tid < n) {
    node = leaf[tid];
    node_store(tree + node, (double)prio[pos[tid]]);
  }
  for (int lv = 0; lv < levels; ++lv) {
    __syncthreads();
    if (node > 0) {
      const int64_t parent = (node - 1) >> 1;
      const double s = __dadd_rn(node_load(tree + 2 * parent + 1), node_load(tree + 2 * parent + 2));
      node_store(tree + parent, s);
      node = parent;
    }
  }
}

DRA_API int dra_sumtree_commit_f32(dra_sumtree* t, const int64_t* leaf_idx_dev, const int32_t* pos_dev, int n,
                                   const float* prio_f32_dev, int batch, double* stat_dev, int force_ordered, void* stream) {
  if (!t || !prio_f32_dev || !stat_dev || n < 0 || n > 1024 || batch < 1 || batch > 1024 || (n > 0 && (!leaf_idx_dev || !pos_dev)))
    return DRA_EINVAL;
  const int m = n > batch ? n : batch;
  const int threads = ((m + 63) / 64) * 64;
  hipLaunchKernelGGL(sumtree_commit_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), t->tree, t->levels, t->capacity,
                     leaf_idx_dev, pos_dev, n, prio_f32_dev, batch, stat_dev, force_ordered);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Stratified sample (replay.py:168-175 + sum_tree.py:23-33): lane i draws
//   s = a + (b - a) * u_i,  a = seg*i, b = seg*(i+1), seg = total / B      (python random.uniform)
// and descends `s <= left ? left : (right, s - left)` until 2i+1 >= n_nodes.  All fp64, no
// contraction, same association as the reference.
__global__ void __launch_bounds__(1024)
sumtree_sample_kernel(const double* __restrict__ tree, int64_t n_nodes, const double* __restrict__ u, int batch,
                      int64_t* __restrict__ out_idx, double* __restrict__ out_p, double* __restrict__ out_total) {
  const int i = threadIdx.x;
  const double total = tree[0];
  if (i == 0 && out_total) *out_total = total;
  if (i >= batch) return;
  const double seg = __ddiv_rn(total, (double)batch);
  const double a = __dmul_rn(seg, (double)i);
  const double b = __dmul_rn(seg, (double)(i + 1));
  double s = __dadd_rn(a, __dmul_rn(__dsub_rn(b, a), u[i]));
  int64_t idx = 0;
  while (true) {
    const int64_t left = 2 * idx + 1;
    if (left >= n_nodes) break;
    const double lv = tree[left];
    if (s <= lv) idx = left;
    else { idx = left + 1; s = __dsub_rn(s, lv); }
  }
  out_idx[i] = idx;
  out_p[i] = tree[idx];
}

DRA_API int dra_sumtree_sample(dra_sumtree* t, const double* u_dev, int batch, int64_t* out_tree_idx, double* out_p,
                               double* out_total, void* stream) {
  if (!t || !u_dev || !out_tree_idx || !out_p || batch < 1 || batch > 1024) return DRA_EINVAL;
  const int threads = ((batch + 63) / 64) * 64;
  hipLaunchKernelGGL(sumtree_sample_kernel, dim3(1), dim3(threads), 0, dra_stream(stream), (const double*)t->tree,
                     t->n_nodes, u_dev, batch, out_tree_idx, out_p, out_total);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}

// Bottom-up rebuild of every internal node from the leaves (state restore / tests): one launch
// per heap level, deepest first.  Level L holds nodes [2^L - 1, 2^(L+1) - 2].
__global__ void __launch_bounds__(256)
sumtree_rebuild_level_kernel(double* __restrict__ tree, int64_t first, int64_t last, int64_t n_nodes) {
  const int64_t node = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node > last) return;
  const int64_t l = 2 * node + 1;
  if (l >= n_nodes) return;  // a leaf living on this level
  tree[node] = __dadd_rn(tree[l], tree[l + 1]);
}

DRA_API int dra_sumtree_rebuild(dra_sumtree* t, void* stream) {
  if (!t) return DRA_EINVAL;
  for (int lv = t->levels - 1; lv >= 0; --lv) {
    const int64_t first = ((int64_t)1 << lv) - 1;
    int64_t last = ((int64_t)1 << (lv + 1)) - 2;
    if (last > t->n_nodes - 1) last = t->n_nodes - 1;
    const int64_t cnt = last - first + 1;
    hipLaunchKernelGGL(sumtree_rebuild_level_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, dra_stream(stream),
                       t->tree, first, last, t->n_nodes);
    DRA_LAUNCH_CHECK();
  }
  return DRA_OK;
}

// ---- PrioritizedReplay.sample() entirely on the device (include/deeprl_amd.h dra_per_chain2_io) ---------------------------
// PrioritizedReplay in the two-stream pipeline cost 229 us per agent step against 117 us with uniform replay while the
// write-back of update t, the adds of step t+1 and the descent of draw t+1 ran on their own stream with the host in between
// (tools/diag_per_host.py).  Round 3's first form put those three steps into ONE kernel behind the loss with the host still
// gating / collecting between two updates (5.5 -> 7.0 k updates/s was the second form's gain over it); it was removed in
// round 4.  What remains: the whole draw on the device, same order as the reference's
// update_priorities(t) -> feed(t+1) x n -> sample(t+1) (replay.py:164-196).
// The body lives in per_chain2.h (it also rides in the update's conv3 backward launch as a role: fused.hip ChainRole).
__global__ void __launch_bounds__(1024) sumtree_per_chain2_kernel(const PerChain2Args a) {
  __shared__ __attribute__((aligned(16))) char smem[per_chain2_lds_bytes<1024>()];
  per_chain2_body<1024>(a, smem);
}

int dra_sumtree_per_chain2_args(dra_sumtree* t, dra_per_chain2_io* io_pinned, const float* loss_vec_dev, float replay_eps,
                                float replay_alpha, float* prio_out_dev, double* stat_dev, void* dev_state,
                                const uint32_t* rng_words_pinned, int64_t* idx_out_dev, float* samp_prob_dev,
                                float* weights_out_dev, int batch, PerChain2Args* out) {
  if (!t || !io_pinned || !loss_vec_dev || !prio_out_dev || !stat_dev || !dev_state || !rng_words_pinned || !idx_out_dev ||
      !samp_prob_dev || !weights_out_dev || batch < 1 || batch > DRA_PER_CHAIN_MAX || !out)
    return DRA_EINVAL;
  out->tree = t->tree; out->levels = t->levels; out->nb = batch; out->capacity = t->capacity; out->n_nodes = t->n_nodes;
  out->io = io_pinned; out->loss_vec = loss_vec_dev; out->eps = replay_eps; out->alpha = replay_alpha; out->prio_out = prio_out_dev;
  out->stat = stat_dev; out->dev = reinterpret_cast<PerChain2Dev*>(dev_state); out->words = rng_words_pinned;
  out->idx_out = idx_out_dev; out->samp_prob = samp_prob_dev; out->weights_out = weights_out_dev;
  return DRA_OK;
}

DRA_API int dra_sumtree_per_chain2_state_bytes(int64_t* bytes) {
  if (!bytes) return DRA_EINVAL;
  *bytes = (int64_t)sizeof(PerChain2Dev);
  return DRA_OK;
}

// installs {cursor, launch count, leaves of the next minibatch} from the host (blocking copies; the device is idle)
DRA_API int dra_sumtree_per_chain2_state_set(void* dev_state, uint64_t rng_cursor, uint64_t seq, const int64_t* tree_idx_host, int n) {
  if (!dev_state || !tree_idx_host || n < 0 || n > DRA_PER_CHAIN_MAX) return DRA_EINVAL;
  PerChain2Dev* d = reinterpret_cast<PerChain2Dev*>(dev_state);
  const unsigned long long head[2] = {(unsigned long long)rng_cursor, (unsigned long long)seq};
  DRA_HIP(hipMemcpy(&d->rng_cursor, &head[0], sizeof(head[0]), hipMemcpyHostToDevice));
  DRA_HIP(hipMemcpy(&d->seq, &head[1], sizeof(head[1]), hipMemcpyHostToDevice));
  DRA_HIP(hipMemcpy(d->tidx, tree_idx_host, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
  return DRA_OK;
}

DRA_API int dra_sumtree_per_chain2(dra_sumtree* t, dra_per_chain2_io* io_pinned, const float* loss_vec_dev, float replay_eps,
                                   float replay_alpha, float* prio_out_dev, double* stat_dev, void* dev_state,
                                   const uint32_t* rng_words_pinned, int64_t* idx_out_dev, float* samp_prob_dev,
                                   float* weights_out_dev, int batch, void* stream) {
  PerChain2Args a;
  const int rc = dra_sumtree_per_chain2_args(t, io_pinned, loss_vec_dev, replay_eps, replay_alpha, prio_out_dev, stat_dev, dev_state,
                                             rng_words_pinned, idx_out_dev, samp_prob_dev, weights_out_dev, batch, &a);
  if (rc) return rc;
  hipLaunchKernelGGL(sumtree_per_chain2_kernel, dim3(1), dim3(1024), 0, dra_stream(stream), a);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
