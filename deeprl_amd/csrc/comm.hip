// comm: gradient all-reduce over RCCL for the data-parallel on-policy agents (SURVEY.md 8b "comm", 8e).
//
// The reference has no collective (its only multi-GPU mode is one job per GPU, docker_batch.sh:2-8).  Data parallelism
// exists here only where the path shards naturally: A2C / PPO partition their vectorised environments over the GPUs
// of a node, every rank holds the full weights, and ONE exchange happens per optimizer step -- the sum of the flat fp32
// gradient buffer (6.75 MB for the Atari actor-critic) over xGMI, scaled by 1/ranks, after which the fused
// clip + optimizer kernels (optim.hip) run identically on every rank (A2C_agent.py:55-64, PPO_agent.py:77-99).
//
// xGMI is point to point (7 links per GPU), so at 6.75 MB RCCL's ring is latency- rather than bandwidth-bound; no bucketing
// (the buffer is already flat).  Round 4: the host issues the exchange in TWO calls -- [fc4 + heads] (6.4 MB, complete as soon
// as fc4's backward has run) on a communication stream while the convolutions are still being differentiated, then the
// 0.3 MB of convolution gradients -- and joins before the clip + optimizer launch (deeprl_amd/dist.py DataParallel.plan_split);
// the per-rank scale runs on the same stream right in front of each call.
//
// One process per GPU; the unique id travels from rank 0 to the others by whatever out-of-band channel the host has
// (deeprl_amd/dist.py uses torch.distributed's store).
#include "common.h"
#include <new>
#include <rccl/rccl.h>
#include <string.h>

struct dra_comm {
  ncclComm_t comm;
  int n_ranks, rank;
};

static_assert(DRA_COMM_ID_BYTES >= sizeof(ncclUniqueId), "unique id buffer");

// rank 0: fills id[DRA_COMM_ID_BYTES]; the host ships the bytes to every other rank.
DRA_API int dra_comm_unique_id(void* id_bytes) {
  if (!id_bytes) return DRA_EINVAL;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return DRA_EINVAL;
  memset(id_bytes, 0, DRA_COMM_ID_BYTES);
  memcpy(id_bytes, &id, sizeof(id));
  return DRA_OK;
}

// Collective: every rank calls it with the same id (blocks until all n_ranks joined).  The calling thread's current
// HIP device is the rank's GPU.
DRA_API int dra_comm_init_rank(dra_comm** out, int n_ranks, int rank, const void* id_bytes) {
  if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks || !id_bytes) return DRA_EINVAL;
  dra_comm* c = new (std::nothrow) dra_comm();
  if (!c) return DRA_ENOMEM;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  if (ncclCommInitRank(&c->comm, n_ranks, id, rank) != ncclSuccess) { delete c; return DRA_EINVAL; }
  c->n_ranks = n_ranks; c->rank = rank;
  *out = c;
  return DRA_OK;
}

// What RCCL itself reports for this communicator (ncclCommCount / ncclCommUserRank): bench.py prints it next to the launcher's
// WORLD_SIZE so that a scaling record says how many ranks the collective really spanned.
DRA_API int dra_comm_info(dra_comm* c, int* n_ranks, int* rank) {
  if (!c) return DRA_EINVAL;
  int n = 0, r = -1;
  if (ncclCommCount(c->comm, &n) != ncclSuccess || ncclCommUserRank(c->comm, &r) != ncclSuccess) return DRA_EINVAL;
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = r;
  return DRA_OK;
}

DRA_API int dra_comm_destroy(dra_comm* c) {
  if (!c) return DRA_OK;
  ncclCommDestroy(c->comm);
  delete c;
  return DRA_OK;
}

__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ x, int64_t n, float s) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  float4* x4 = reinterpret_cast<float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = x4[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    x4[i] = v;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= s;
}

// flat_grad[0, count) <- sum over ranks of (scale_r * flat_grad_r), in place, asynchronous on `stream`: every rank scales ITS
// gradient by ITS `scale` first and the scaled buffers are summed.  scale = 1 / n_ranks on every rank gives the gradient of the
// mean loss over the GLOBAL rollout when the shards are equally sized (A2C); PPO's shuffled minibatches give each rank
// scale_r = rows_r / rows of the global minibatch, which differs per rank -- scaling after the sum (the round-4 order) handed
// every rank a different gradient (dist.py:281).  Same order as the gloo path (mul_ then all_reduce).
DRA_API int dra_allreduce_grads(float* flat_grad, int64_t count, float scale, dra_comm* c, void* stream) {
  if (!flat_grad || count < 1 || !c || (((uintptr_t)flat_grad) & 15)) return DRA_EINVAL;
  hipStream_t st = dra_stream(stream);
  if (scale != 1.f) {
    int64_t blocks = ((count >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, flat_grad, count, scale);
    DRA_LAUNCH_CHECK();
  }
  if (c->n_ranks > 1) {
    if (ncclAllReduce(flat_grad, flat_grad, (size_t)count, ncclFloat, ncclSum, c->comm, st) != ncclSuccess) return DRA_EINVAL;
  }
  return DRA_OK;
}

// (sum, sum of squares, count) style reductions of a few fp64 scalars: PPO's advantage statistics are over the GLOBAL
// rollout (PPO_agent.py:66).
DRA_API int dra_allreduce_f64(double* values, int count, dra_comm* c, void* stream) {
  if (!values || count < 1 || !c) return DRA_EINVAL;
  if (c->n_ranks > 1) {
    if (ncclAllReduce(values, values, (size_t)count, ncclDouble, ncclSum, c->comm, dra_stream(stream)) != ncclSuccess)
      return DRA_EINVAL;
  }
  return DRA_OK;
}
