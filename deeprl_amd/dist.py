"""Multi-GPU: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference has no collective anywhere; what shards naturally is decided per agent:

  * DQN family (off-policy, one learner + one env): REPLICAS ONLY -- one independent
    (seed, ring, learner) per GPU, exactly the reference's one-job-per-GPU mode
    (docker_batch.sh:2-8).  No data-path collective; bench.py --gpus N runs N replicas.
  * A2C / PPO (on-policy): the env axis is partitioned over ranks; every rank holds the full
    weights and its own rollout; ONE exchange step per optimizer step -- a sum all-reduce of the
    flat fp32 gradient (6.75 MB for the Atari actor-critic), scaled by 1/world, after which the
    fused clip + optimizer kernels run identically on every rank.  PPO additionally needs the
    advantage mean / std over the GLOBAL rollout (PPO_agent.py:66): three scalars all-reduced.

`GradAllReduce` plugs into A2CAgent / PPOAgent through their `grad_hook`.
"""
import torch
import torch.distributed as dist


def init(backend=None):
    """Initialises the default process group from the torchrun environment (RANK / WORLD_SIZE /
    MASTER_*).  backend defaults to nccl (RCCL) when a GPU is visible, gloo otherwise."""
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend)


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


class GradAllReduce:
    """grad_hook: averages the flat gradient buffer over ranks in place (one collective per
    optimizer step).  Equal-sized env shards make the averaged gradient equal to the
    single-process mean-loss gradient up to fp32 summation order."""

    def __init__(self, group=None):
        self.group = group
        self.calls = 0

    def __call__(self, flat_grad):
        w = dist.get_world_size(self.group)
        if w == 1:
            return flat_grad
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        flat_grad.mul_(1.0 / w)
        self.calls += 1
        return flat_grad


def shard_envs(num_envs, group=None):
    """Contiguous env shard [lo, hi) of this rank (rank g owns envs g*N/G .. (g+1)*N/G - 1)."""
    w, r = dist.get_world_size(group), dist.get_rank(group)
    if num_envs % w:
        raise ValueError("num_envs (%d) must be divisible by the world size (%d)" % (num_envs, w))
    per = num_envs // w
    return r * per, (r + 1) * per


def global_advantage_normalize_(adv, group=None):
    """PPO_agent.py:66 over the rollouts of ALL ranks: (a - mean) / std with the unbiased std of the
    global T*N entries; all-reduces (sum, sum of squares, count)."""
    a = adv.double()
    stats = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(a.numel()), dtype=torch.float64, device=adv.device)])
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    s, ss, n = stats[0], stats[1], stats[2]
    mean = s / n
    var = (ss - n * mean * mean) / (n - 1)
    adv.sub_(mean.to(adv.dtype)).div_(var.sqrt().to(adv.dtype))
    return adv
