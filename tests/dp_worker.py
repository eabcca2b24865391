"""Worker of tests/test_gpu_data_parallel.py (not a test): runs an on-policy agent for a few updates, either as the single
process of a world of 1 or as one rank of a torchrun world, and writes rank 0's parameters to an .npz.

    python tests/dp_worker.py <a2c|ppo> <out.npz> [updates]           (torchrun sets RANK / WORLD_SIZE / MASTER_*)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    kind, out, updates = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import deeprl_amd as d
    import deeprl_amd.agents as agents_mod
    import deeprl_amd.dist as dd

    class Quiet:
        def info(self, *a, **k):
            pass
        add_scalar = add_histogram = info

    agents_mod.get_logger = lambda *a, **k: Quiet()
    gpu = 0
    if world > 1:
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if os.environ.get("DP_WORKER_BACKEND") == "nccl":     # every rank owns a GPU: RCCL carries the gradients (comm.hip)
            gpu = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(gpu)
            dd.init("nccl")
        else:
            dd.init("gloo")                   # ranks share ONE GPU on the test box: gloo moves the bytes
    d.select_device(gpu)
    # identical initial weights on every rank -- or, with DP_WORKER_RANK_SEEDS=1, a different torch seed per rank (what
    # `python -m deeprl_amd.launch` does under torchrun): the agents must then start from rank 0's parameters
    torch.manual_seed(int(os.environ.get("RANK", "0")) if os.environ.get("DP_WORKER_RANK_SEEDS") else 0)
    np.random.seed(0)
    c = d.Config()
    c.merge(dict(game="synthetic-atari", log_level=0, tag="dp", dp_invariant_sampling=True, dp_noise_seed=7))
    c.num_workers = 4
    c.task_fn = lambda: d.Task(c.game, num_envs=c.num_workers, seed=100 + c.env_shard[0], synthetic_done_period=9)
    c.eval_env = d.Task(c.game, seed=1)
    c.network_fn = lambda: d.CategoricalActorCriticNet(c.state_dim, c.action_dim, d.NatureConvBody())
    c.state_normalizer, c.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    c.discount, c.use_gae, c.entropy_weight, c.max_steps = 0.99, True, 0.01, int(1e6)
    if kind == "a2c":       # examples.py:361-381 at a size the test can afford
        c.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-4, alpha=0.99, eps=1e-5)
        c.gae_tau, c.rollout_length, c.gradient_clip = 1.0, 5, 5
        agent = d.A2CAgent(c)
    else:                   # examples.py:525-550
        c.optimizer_fn = lambda p: torch.optim.Adam(p, lr=2.5e-4)
        c.gae_tau, c.rollout_length, c.gradient_clip = 0.95, 8, 0.5
        c.optimization_epochs, c.mini_batch_size, c.ppo_ratio_clip, c.shared_repr = 2, 8, 0.1, True
        agent = d.PPOAgent(c)
    assert agent.dp.active == (world > 1) and agent.dp.global_workers == 4
    for _ in range(updates):
        agent.step()
    torch.cuda.synchronize()
    if world > 1 and os.environ.get("DP_WORKER_RANK_SEEDS"):
        np.savez(out + ".rank%d.npz" % dd.rank(), **{k: v.detach().cpu().numpy() for k, v in agent.network.state_dict().items()})
    if dd.rank() == 0:
        extra = {}
        fused = getattr(agent, "_fused", None)
        if kind == "ppo" and fused is not None and fused.kind == "adam":
            # Adam's second-moment estimate per element (the test's error bound is stated in terms of it) + the step count
            names = {id(p): n for n, p in agent.network.named_parameters()}
            for p, o in zip(fused.flat.params, fused.flat.offsets):
                v = fused.state2[o:o + p.numel()]
                if p.dim() == 4 and not p.data.is_contiguous():          # KOC storage [(c,kh,kw)][oc] (optim.FlatParams)
                    oc, c, kh, kw = p.shape
                    v = v.view(c, kh, kw, oc).permute(3, 0, 1, 2)
                else:
                    v = v.view(p.shape)
                extra["__adam_v." + names[id(p)]] = v.detach().cpu().numpy()
            extra["__adam_steps"] = np.asarray(fused.steps)
        np.savez(out, total_steps=agent.total_steps, **extra,
                 **{k: v.detach().cpu().numpy() for k, v in agent.network.state_dict().items()})
    agent.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
