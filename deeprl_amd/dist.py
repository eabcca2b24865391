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

`DataParallel` is what A2CAgent / PPOAgent consult; it is active whenever torch.distributed is initialised with more
than one rank (config.data_parallel = False opts out).  The gradient exchange itself is the C ABI's
dra_allreduce_grads (csrc/comm.hip: one ncclAllReduce over RCCL on the agent's stream) when every rank owns its own
GPU; ranks that share a device (tests on a one-GPU box) and CPU tensors go through torch.distributed (gloo).
"""
import ctypes

import numpy as np
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


# ---- host-side placement of the ranks of one node -------------------------------------------------------------------------
# Every rank is a Python driver that enqueues ~65-100 us of host work per agent step (bench.py `host`); eight of them wandering
# over the cores of both sockets is the contention SURVEY.md 8(e) names as the risk for ">= 6x at 8 GPUs" (the reference runs one
# docker job per GPU and leaves placement to the kernel, docker_batch.sh:2-8).  Each rank pins itself to cores of the NUMA node
# its GPU hangs off, disjoint from the other ranks of the node.

def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def plan_affinity(allowed, node_cpus_of_rank, local_rank, core_of=None):
    """Pure placement rule.  allowed: cpus this process may run on; node_cpus_of_rank[r]: cpus local to the GPU of local rank r
    (None = unknown).  Ranks whose GPUs share a node split that node's allowed cpus into contiguous, disjoint shares in rank
    order; a rank with no topology information gets the share of an even split of `allowed`; with fewer cpus than ranks the
    ranks share round-robin.  core_of (optional): cpu -> id of its physical core; the shares are then cut from the list ordered
    by core, so that the hardware threads of one core go to the SAME rank (a node's cpulist is "64-127,192-255" on the MI355X
    hosts: 192 is the sibling of 64 -- splitting the list in halves would put two ranks on every core).  Returns the sorted cpu
    list of `local_rank` (never empty)."""
    allowed = sorted(set(int(c) for c in allowed))
    order = (lambda c: (core_of.get(c, c), c)) if core_of else (lambda c: c)
    if not allowed:
        raise ValueError("plan_affinity: no cpu allowed")
    n = len(node_cpus_of_rank)
    keys = []
    for r in range(n):
        local = node_cpus_of_rank[r]
        pool = sorted(set(local) & set(allowed), key=order) if local else []
        keys.append(tuple(pool) if pool else None)
    mine = keys[local_rank]
    pool = list(mine) if mine is not None else sorted(allowed, key=order)
    group = [r for r in range(n) if keys[r] == mine]          # the ranks that draw from the same pool, in rank order
    pos, g = group.index(local_rank), len(group)
    if len(pool) < g:
        return [pool[pos % len(pool)]]
    lo, hi = (pos * len(pool)) // g, ((pos + 1) * len(pool)) // g
    return sorted(pool[lo:hi])


def cpu_cores(cpus, sysfs="/sys"):
    """cpu -> smallest cpu number among its hardware-thread siblings (an id of the physical core); cpus without topology files
    map to themselves."""
    import os
    out = {}
    for c in cpus:
        try:
            sib = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c, "topology",
                                                   "thread_siblings_list")).read())
            out[c] = min(sib) if sib else c
        except (OSError, ValueError):
            out[c] = c
    return out


def gpu_local_cpus(device_index, sysfs="/sys"):
    """cpus of the NUMA node GPU `device_index` is attached to, from the PCI device's local_cpulist; None when unknown."""
    import os
    try:
        pr = torch.cuda.get_device_properties(device_index)
        dom, bus, dev = int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id)
    except Exception:
        return None
    for fn in (0, 1):
        path = os.path.join(sysfs, "bus", "pci", "devices", "%04x:%02x:%02x.%d" % (dom, bus, dev, fn), "local_cpulist")
        try:
            cpus = _parse_cpulist(open(path).read())
            if cpus:
                return cpus
        except OSError:
            continue
    return None


def pin_rank(local_rank, local_world, devices=None):
    """Pins the calling process (os.sched_setaffinity) per plan_affinity; devices[r] = GPU index of local rank r (default
    r mod the visible GPUs).  Returns the mapping for the bench line; DRA_NO_PIN=1 or any failure leaves the process
    where it is and says so."""
    import os
    info = {"local_rank": int(local_rank), "local_world": int(local_world), "pinned": False}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if devices is None:
            devices = [(r % n_dev) if n_dev else None for r in range(local_world)]
        node = [gpu_local_cpus(dv) if dv is not None else None for dv in devices]
        cpus = plan_affinity(allowed, node, local_rank, cpu_cores(allowed))
        info.update(device=devices[local_rank], gpu_node_known=node[local_rank] is not None, cpus=cpus, allowed=len(allowed))
        if os.environ.get("DRA_NO_PIN") == "1" or local_world <= 1:
            info["note"] = "not pinned (single rank or DRA_NO_PIN=1)"
            return info
        # every thread the process already has (the HIP runtime's helpers exist as soon as a device was queried); threads
        # created later inherit the caller's mask
        tids = [int(t) for t in os.listdir("/proc/self/task")] if os.path.isdir("/proc/self/task") else [0]
        for tid in tids:
            try:
                os.sched_setaffinity(tid, cpus)
            except OSError:
                pass
        os.sched_setaffinity(0, cpus)
        info["pinned"] = True
        info["threads_pinned"] = len(tids)
    except Exception as e:      # placement is an optimisation: never fail a run over it
        info["error"] = repr(e)
    return info


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


class RcclComm:
    """dra_comm (include/deeprl_amd.h "comm"): an RCCL communicator over the ranks of the default process group; the
    unique id travels through torch.distributed's own broadcast (any backend)."""

    def __init__(self):
        from ._lib import lib
        self.lib = lib
        self.world, self.rank = world(), rank()
        buf = (ctypes.c_ubyte * 256)()
        if self.rank == 0:
            lib.dra_comm_unique_id(buf)
        if self.world > 1:
            on_gpu = dist.get_backend() == "nccl"
            t = torch.tensor(list(buf), dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
            dist.broadcast(t, 0)
            for i, v in enumerate(t.cpu().tolist()):
                buf[i] = v
        self.h = ctypes.c_void_p()
        lib.dra_comm_init_rank(ctypes.byref(self.h), self.world, self.rank, buf)

    def allreduce_grads(self, flat, scale, stream=None):
        """flat <- (sum over ranks of flat) * scale on `stream` (default: the current stream).  `flat` may be a contiguous
        16-byte aligned SEGMENT of the gradient buffer (the early fc4 exchange of DataParallel)."""
        from ._lib import stream_ptr
        self.lib.dra_allreduce_grads(ctypes.c_void_p(flat.data_ptr()), flat.numel(), float(scale), self.h, stream_ptr(stream))
        return flat

    def info(self):
        """(ranks, this rank) as RCCL itself reports them."""
        n, r = ctypes.c_int(0), ctypes.c_int(-1)
        self.lib.dra_comm_info(self.h, ctypes.byref(n), ctypes.byref(r))
        return n.value, r.value

    def close(self):
        if self.h:
            self.lib.dra_comm_destroy(self.h)
            self.h = None


def _distinct_gpus():
    """True when every rank of the default group drives a different GPU (the production layout: one process per GPU)."""
    if not torch.cuda.is_available():
        return False
    mine = torch.tensor([torch.cuda.current_device(), torch.cuda.device_count()], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        return True
    allv = [torch.zeros_like(mine) for _ in range(world())]
    dist.all_gather(allv, mine)
    return len({int(v[0]) for v in allv}) == world() and world() <= int(mine[1])


class DataParallel:
    """Data-parallel state of one on-policy agent.  Rank g owns environments [lo, hi) of the
    config.num_workers GLOBAL environments; the agent builds only its shard (config.num_workers is rewritten to the shard
    size before task_fn runs; config.env_shard = (lo, hi) for task factories that seed per environment), counts GLOBAL
    environment steps, and calls

        sum_grads(flat, weight)        one all-reduce per optimizer step: flat <- sum over ranks of weight * flat
        sum_scalars([...])             fp64 scalars summed over ranks (advantage statistics, KL gate)
        permutation(n)                 the SAME permutation on every rank (one RandomState seeded by rank 0)
        sample(logits)                 categorical actions of the shard's rows by the Gumbel trick, the noise a counter hash of
                                       (noise seed, rollout step, GLOBAL environment, action): G ranks x N/G environments act
                                       exactly like 1 rank x N.  The rollout step is a device counter the kernel advances
                                       (ops.gumbel_sample), so data-parallel rollouts replay from captured graphs
    """

    def __init__(self, config):
        self.world, self.rank = world(), rank()
        self.active = self.world > 1 and getattr(config, "data_parallel", True) is not False
        self.invariant_sampling = self.active or bool(getattr(config, "dp_invariant_sampling", False))
        # the GLOBAL environment count survives in config.global_num_workers: a second agent built from the same Config (an
        # evaluation agent, a restart) shards the global count again instead of the already-sharded one
        if self.active and getattr(config, "global_num_workers", None):
            config.num_workers = config.global_num_workers
        self.global_workers = config.num_workers
        if self.active:
            config.global_num_workers = config.num_workers
        self.lo, self.hi = 0, config.num_workers
        self.comm = None
        self.rs = None
        noise_seed = getattr(config, "dp_noise_seed", None)
        if self.active:
            self.lo, self.hi = shard_envs(config.num_workers)
            config.num_workers = self.hi - self.lo
            if noise_seed is None:          # rank 0 picks the seed of the shared streams
                seed = torch.tensor([np.random.randint(1 << 30)], dtype=torch.int64)
                if dist.get_backend() == "nccl":
                    seed = seed.cuda()
                dist.broadcast(seed, 0)
                noise_seed = int(seed.item())
            if _distinct_gpus():
                self.comm = RcclComm()
        if noise_seed is None:
            # no configured seed: follow the run's torch seed (random_seed() / torch.manual_seed, misc.py:43-46), as the
            # reference's dist.sample() does -- read without consuming any generator, so every other stream stays where the
            # reference has it.  (Round 5 used 0 here: every run, whatever its seed, explored with the same noise.)
            noise_seed = int(torch.initial_seed()) & 0x3fffffff
        self.noise_seed = int(noise_seed)
        if self.invariant_sampling:         # permutations and action noise come from streams every rank can reproduce
            self.rs = np.random.RandomState(self.noise_seed + 12345)
        config.env_shard = (self.lo, self.hi)
        self._gen = None
        self.step_dev = None          # device int64: sampler calls so far (the noise stream's position)
        self._split = None            # plan_split(): float offset where the early (fc4 + heads) segment of the gradient starts
        self._comm_stream = None
        self._early, self._weight, self._hooks = None, 1.0, []
        self.early_exchanges = 0      # tail segments that went out before sum_grads() was called

    def sampler_state(self):
        """(noise seed, sampler calls so far): the position of the counter-hash noise streams, for checkpoints."""
        step = 0 if self.step_dev is None else int(self.step_dev.item())
        return {"noise_seed": int(self.noise_seed), "step": step}

    def load_sampler_state(self, state, device=None):
        self.noise_seed = int(state["noise_seed"])
        if self.invariant_sampling:
            self.rs = np.random.RandomState(self.noise_seed + 12345)
        if self.step_dev is None:
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=device if device is not None else "cpu")
        self.step_dev.fill_(int(state["step"]))

    @property
    def is_main(self):
        """Rank that writes checkpoints / evaluation logs (every rank holds the same parameters)."""
        return not self.active or self.rank == 0

    def sync_state(self, *tensors):
        """Rank 0's values of `tensors` (flat parameter buffer, optimizer state, module buffers) on every rank, in place.
        Ranks launched with different seeds (python -m deeprl_amd.launch under torchrun seeds 1 + RANK so that exploration
        differs) would otherwise build different initial weights and train G different replicas on one averaged gradient."""
        if not self.active:
            return
        nccl = dist.get_backend() == "nccl"
        for t in tensors:
            if t is None:
                continue
            if t.is_cuda and not nccl:
                h = t.detach().cpu()
                dist.broadcast(h, 0)
                t.data.copy_(h)
            elif not t.is_cuda and nccl:
                h = t.detach().cuda()
                dist.broadcast(h, 0)
                t.data.copy_(h.cpu())
            else:
                dist.broadcast(t.data, 0)

    def permutation(self, n):
        return self.rs.permutation(n) if self.rs is not None else np.random.permutation(n)

    def sample(self, logits):
        """Rank-invariant Categorical(logits).sample() for this rank's rows (module docstring)."""
        from . import ops
        if self.step_dev is None:
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=logits.device)
        return ops.gumbel_sample(logits, self.noise_seed, self.step_dev, self.lo)

    def uniforms(self, step, n_global, k, device):
        if self._gen is None:
            self._gen = torch.Generator(device=device)
        self._gen.manual_seed(self.noise_seed * 1000003 + int(step))
        return torch.rand(n_global, k, generator=self._gen, device=device)[self.lo:self.hi]

    # -- the gradient exchange, split so that most of it overlaps the backward pass -----------------
    # The flat gradient of the Atari actor-critic is [conv1 conv2 conv3 | fc4 heads]: 0.3 MB of convolution gradients in front of
    # 6.4 MB that is COMPLETE as soon as fc4's backward has run -- before conv3 / conv2 / conv1 are differentiated.  plan_split()
    # marks that boundary and counts the gradient accumulations of the tail's parameters; when the last of them has fired the
    # tail segment goes out on a communication stream (RCCL; gloo: an asynchronous collective) while the convolutions'
    # backward still runs, and sum_grads() then only exchanges the small head segment and joins.  Element for element the two
    # segment sums are the one-call sum (with two ranks bit for bit: one addition per element either way;
    # tests/test_dist_gloo.py); every rank receives identical values whatever the split.
    def plan_split(self, flat_params, tail_params):
        """flat_params: optim.FlatParams of the network; tail_params: the parameters whose gradients complete first
        (fc4 + heads) -- they must form the END of the flat buffer.  Without a plan sum_grads() is the single exchange."""
        self._split = None
        self._hooks = getattr(self, "_hooks", [])
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if not self.active or not tail_params:
            return
        offs = sorted(flat_params.offset_of(p) for p in tail_params)
        rest = [o for p, o in zip(flat_params.params, flat_params.offsets) if all(p is not q for q in tail_params)]
        if rest and max(rest) > offs[0]:
            return                       # the tail is not contiguous at the end of the buffer: keep the single exchange
        self._split = int(offs[0])
        self._tail_n, self._tail_seen, self._early, self._weight = len(tail_params), 0, None, 1.0
        self._flat_grad = flat_params.grad

        def fired(_p):
            self._tail_seen += 1
            if self._tail_seen == self._tail_n:
                self._begin_tail()
        for p in tail_params:
            self._hooks.append(p.register_post_accumulate_grad_hook(fired))

    def set_weight(self, weight):
        """The factor sum_grads() will be called with, announced BEFORE the backward pass (the early segment needs it)."""
        self._weight = float(weight)

    def _begin_tail(self):
        """Every tail gradient has been accumulated: exchange flat[split:] now, asynchronously."""
        flat, lo = self._flat_grad, self._split
        seg = flat[lo:]
        if flat.is_cuda:
            if self.comm is None or torch.cuda.is_current_stream_capturing():
                return                   # shared-GPU test box (gloo over host memory) / graph capture: exchanged at the tail
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
                self._ev_ready, self._ev_done = torch.cuda.Event(), torch.cuda.Event()
            self._ev_ready.record()                                   # the backward launches that wrote the tail segment
            self._comm_stream.wait_event(self._ev_ready)
            self.comm.allreduce_grads(seg, self._weight, stream=self._comm_stream)
            self._early = "rccl"
        else:
            if self._weight != 1.0:
                seg.mul_(self._weight)
            self._early = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)
        self.early_exchanges += 1

    def sum_grads(self, flat, weight=1.0):
        """flat <- sum over ranks of weight * flat (one exchange per optimizer step; in two segments when plan_split() is in
        effect, the large one possibly already under way)."""
        split = getattr(self, "_split", None)
        if split is not None:
            self._tail_seen = 0
        if not self.active:
            if weight != 1.0:
                flat.mul_(weight)
            return flat
        early, self._early = getattr(self, "_early", None), None
        if early is not None and float(weight) != self._weight:
            raise RuntimeError("DataParallel: the early segment went out with weight %r, sum_grads got %r" % (self._weight, weight))
        if split is None or flat is not self._flat_grad:
            segs = [flat]
        else:
            segs = [flat[:split]] if early is not None else [flat[split:], flat[:split]]     # fc4's 6.4 MB first
        if self.comm is not None and flat.is_cuda:
            if early == "rccl":         # the head segment follows on the communication stream; the step waits for both
                self._ev_ready.record()
                self._comm_stream.wait_event(self._ev_ready)
                for sg in segs:
                    self.comm.allreduce_grads(sg, weight, stream=self._comm_stream)
                self._ev_done.record(self._comm_stream)
                torch.cuda.current_stream().wait_event(self._ev_done)
            else:
                for sg in segs:
                    self.comm.allreduce_grads(sg, weight)
            return flat
        for sg in segs:
            if weight != 1.0:
                sg.mul_(weight)
            dist.all_reduce(sg, op=dist.ReduceOp.SUM)
        if early is not None:
            early.wait()
        return flat

    def sum_scalars(self, values):
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if self.active:
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.cpu()]

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def gumbel_argmax(logits, u):
    """A categorical sample from `logits` [N, A] driven by uniforms u [N, A]: argmax(logits - log(-log u))."""
    g = -torch.log(-torch.log(u.clamp(1e-20, 1.0 - 1e-7)))
    return torch.argmax(logits + g, dim=-1)
