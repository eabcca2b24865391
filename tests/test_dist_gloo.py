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
