#!/usr/bin/env python
"""Diagnosis (GPU): the in-order DQN + PrioritizedReplay agent step against the CPU oracle of the same run (which reproduces the
reference's own run: tests/test_oracle_vs_golden.py), update by update: minibatch indices, sampling probabilities, importance
weights, TD errors, priorities, parameter digests.  Prints the first quantity that parts ways."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fake_envs  # noqa: E402
from golden.make_golden_cases import trajectory_digest  # noqa: E402
import deeprl_amd as d  # noqa: E402
import deeprl_amd.agents as agents_mod  # noqa: E402
from oracle.async_schedule_oracle import AsyncPerAgentScheduleOracle  # noqa: E402


class Quiet:
    def info(self, *a, **k):
        pass
    add_scalar = add_histogram = info


def main():
    device_env = len(sys.argv) > 1 and sys.argv[1] == "device"
    agents_mod.get_logger = lambda *a, **k: Quiet()
    d.select_device(0)
    done_period, steps = 8, 24
    cfg = d.Config()
    cfg.merge(dict(game="synthetic-atari", n_step=1, replay_cls=d.PrioritizedReplay, async_replay=False, log_level=0, tag="diag",
                   device_env=device_env))
    cfg.task_fn = lambda: d.Task(cfg.game, seed=7, synthetic_done_period=done_period)
    cfg.eval_env = cfg.task_fn()
    cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
    cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
    cfg.random_action_prob = d.LinearSchedule(1.0, 0.05, 60)
    cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
    kw = dict(memory_size=500, batch_size=32, n_step=1, discount=0.99, history_length=4)
    cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
    cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
    cfg.replay_beta = d.LinearSchedule(0.4, 1.0, 1000)
    cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 3, 40, 4
    cfg.gradient_clip, cfg.double_q, cfg.async_actor, cfg.max_steps = 5, False, False, 1e5
    d.random_seed(3)
    random.seed(3)
    agent = d.DQNAgent(cfg)
    p_np = fake_envs.numpy_params(fake_envs.NATURE_SHAPES + [("fc_head.weight", (4, 512)), ("fc_head.bias", (4,))], 17)
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
    agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
    if agent._learner is not None:
        agent._learner.invalidate_actor_copy()
    recs = []
    for t in range(steps):
        agent.step()
        if agent.total_steps > 40:
            L = agent._learner
            L.synchronize()
            torch.cuda.synchronize()
            recs.append(dict(idx=L.idx.cpu().numpy().copy(), sp=L.sampling_prob.cpu().numpy().copy(), w=d.ops._wrap_device_pointer(
                L._weights_ptr, 32, torch.float32).cpu().numpy().copy() if hasattr(L, "_weights_ptr") else None,
                delta=L.delta.cpu().numpy().copy(), prio=L.prio.cpu().numpy().copy(),
                dig=trajectory_digest(agent.network.state_dict()), norm=float(L.norm.item())))
    agent.close()
    # ---- oracle, in order
    np.random.seed(3)
    np.random.randint(int(1e6))
    random.seed(3)
    sched, n_act = d.LinearSchedule(1.0, 0.05, 60), [0]

    def epsilon():
        e = 1 if n_act[0] < 40 else sched()
        n_act[0] += 1
        return e
    orc = AsyncPerAgentScheduleOracle(p_np, 500, 32, env_seed=7, done_period=done_period, actor_rs=np.random, epsilon_fn=epsilon,
                                      beta_fn=d.LinearSchedule(0.4, 1.0, 1000), exploration_steps=40, target_freq=3, head="vanilla", clip=5.0)
    order = ["fc_head.weight", "fc_head.bias"] + [n for n, _ in fake_envs.NATURE_SHAPES]
    k = 0
    for t in range(steps):
        orc.actor_step(orc.p)
        if orc.report():
            ti, pr, di, batch = orc.draw()
            loss, vec, prio, w = orc.learn(ti, pr, batch)
            r = recs[k]
            dig = trajectory_digest({n: orc.p[n] for n in order})
            print("update %2d: idx equal %s | prob max rel %.2e | beta gpu %.6f | delta max abs %.2e | prio max abs %.2e | digest max abs %.2e | "
                  "margin %.1e" % (k, np.array_equal(di, r["idx"]), float(np.abs(r["sp"][:32] - pr.astype(np.float32)).max() / pr.max()),
                                   float(r["sp"][32]), float(np.abs(r["delta"] - vec).max()), float(np.abs(r["prio"] - prio).max()),
                                   float(np.abs(dig - r["dig"]).max()), orc.relu_margin))
            if not np.array_equal(di, r["idx"]):
                print("   oracle idx", di.tolist())
                print("   gpu    idx", r["idx"].tolist())
            k += 1
        orc.maybe_sync_target()


if __name__ == "__main__":
    main()
