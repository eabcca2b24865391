"""Worker of tests/test_gpu_resume.py (not a test): runs a device-resident DQN-family agent in its own process.

    python tests/resume_worker.py <kind> <out.npz> run   <steps>                     uninterrupted
    python tests/resume_worker.py <kind> <out.npz> save  <steps> <checkpoint prefix>  steps, then save_full
    python tests/resume_worker.py <kind> <out.npz> load  <steps> <checkpoint prefix>  fresh agent, load_full, then steps
kind: dqn_async | dqn_sync | c51_per_async | dqn_per_async | c51_async.  Writes final parameters, ring contents (first 400 slots) and the step count."""
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
sys.path.insert(0, _HERE)
import deeprl_amd as d  # noqa: E402
import deeprl_amd.agents as agents_mod  # noqa: E402


class _Quiet:
    def info(self, *a, **k):
        pass
    add_scalar = add_histogram = info


def build(kind):
    agents_mod.get_logger = lambda *a, **k: _Quiet()
    d.select_device(0)
    d.random_seed(3)
    import random
    random.seed(3)      # PrioritizedReplay draws from python `random` (replay.py:169-172), which random_seed() leaves unseeded
    cfg = d.Config()
    per = "per" in kind
    cfg.merge(dict(game="synthetic-atari", n_step=1, replay_cls=d.PrioritizedReplay if per else d.UniformReplay, async_replay=False,
                   log_level=0, tag="resume", device_env=True))
    cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
    cfg.replay_beta = d.LinearSchedule(0.4, 1.0, 1000)
    cfg.task_fn = lambda: d.Task(cfg.game, seed=9, synthetic_done_period=13)
    cfg.eval_env = cfg.task_fn()
    if kind.startswith("dqn"):
        cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
        cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
        cls = d.DQNAgent
    else:
        cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=0.00025, eps=0.01 / 32)
        cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
        cfg.network_fn = lambda: d.CategoricalNet(cfg.action_dim, cfg.categorical_n_atoms, d.NatureConvBody())
        cls = d.CategoricalDQNAgent
    cfg.random_action_prob = d.LinearSchedule(1.0, 0.05, 60)
    cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
    cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, dict(memory_size=600, batch_size=32, n_step=cfg.n_step, discount=cfg.discount,
                                                                  history_length=4), async_=False)
    cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 7, 40, 4
    cfg.gradient_clip, cfg.double_q, cfg.max_steps, cfg.async_actor = 5, False, int(1e6), "async" in kind
    return cls(cfg)


def main():
    kind, out, mode, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    agent = build(kind)
    assert agent._pipe is not None, "the probe must run the device-resident pipeline"
    if mode == "load":
        agent.load_full(sys.argv[5])
    for _ in range(steps):
        agent.step()
    if mode == "save":
        agent.save_full(sys.argv[5])
    agent._learner.synchronize()
    torch.cuda.synchronize()
    ring = agent._inner_replay()._ring
    frames, actions, rewards, masks = ring.arrays()
    np.savez(out, total_steps=agent.total_steps, frames=frames[:400 * 7056].cpu().numpy(), act=actions[:400 * 8].cpu().numpy(),
             rew=rewards[:400].cpu().numpy(), msk=masks[:400].cpu().numpy(),
             **{"p_" + k: v.detach().cpu().numpy() for k, v in agent.network.state_dict().items()},
             **{"t_" + k: v.detach().cpu().numpy() for k, v in agent.target_network.state_dict().items()})
    agent.close()


if __name__ == "__main__":
    main()
