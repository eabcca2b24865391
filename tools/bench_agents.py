#!/usr/bin/env python
"""Agent-level throughput of the other BASELINE.json configs on one MI355X (parity cases, not the bench line):
every agent is built exactly like the reference's examples.py factory it cites, on the synthetic `Task`
environments (CPU numpy emulators stand-ins, like the reference's DummyVecEnv), and `agent.step()` is timed
after a warm-up.  Prints one JSON object per line: agent steps/s, env steps/s, gradient updates/s.

  config 3  ppo_continuous  examples.py:497-523  (HalfCheetah shapes; 1 worker as in the reference, and 16 workers)
  config 2  dqn_pixel examples.py:55-97 through the agent API with a HOST emulator (fused learner attached by
            DQNAgent; `_generic` = config.fused_learner False, the autograd path the other agents use)
  config 4  dqn_pixel + PrioritizedReplay (examples.py:55-97 with replay_cls=PrioritizedReplay),
            categorical_dqn_pixel examples.py:195-226, quantile_regression_dqn_pixel examples.py:129-160
  config 5  a2c_pixel examples.py:361-381 (16 workers), ppo_pixel examples.py:525-550 (8 workers)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d  # noqa: E402
import deeprl_amd.agents as agents_mod  # noqa: E402


class _Quiet:
    def info(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


def dqn_family(kind, replay_cls, ring=200_000, fused=True, device=False, async_actor=None):
    c = d.Config()
    c.merge(dict(game="synthetic-atari", log_level=0, tag="bench", n_step=1, replay_cls=replay_cls, async_replay=False,
                 fused_learner=fused, device_env=device))     # device=False: HOST emulator; True: device-resident environment
    c.task_fn = lambda: d.Task(c.game, seed=1)
    c.eval_env = c.task_fn()
    if kind == "dqn":
        c.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
        c.network_fn = lambda: d.VanillaNet(c.action_dim, d.NatureConvBody(in_channels=4))
        c.gradient_clip = 5
        agent_cls = d.DQNAgent
    elif kind == "c51":
        c.optimizer_fn = lambda p: torch.optim.Adam(p, lr=0.00025, eps=0.01 / 32)
        c.categorical_v_max, c.categorical_v_min, c.categorical_n_atoms = 10, -10, 51
        c.network_fn = lambda: d.CategoricalNet(c.action_dim, c.categorical_n_atoms, d.NatureConvBody())
        c.gradient_clip = 0.5
        agent_cls = d.CategoricalDQNAgent
    else:
        c.optimizer_fn = lambda p: torch.optim.Adam(p, lr=0.00005, eps=0.01 / 32)
        c.num_quantiles = 200
        c.network_fn = lambda: d.QuantileNet(c.action_dim, c.num_quantiles, d.NatureConvBody())
        c.gradient_clip = 5
        agent_cls = d.QuantileRegressionDQNAgent
    c.random_action_prob = d.LinearSchedule(1.0, 0.01, 1e6)
    c.batch_size = 32
    c.discount = 0.99
    c.history_length = 4
    kw = dict(memory_size=ring, batch_size=c.batch_size, n_step=c.n_step, discount=c.discount, history_length=4)
    c.replay_fn = lambda: d.ReplayWrapper(c.replay_cls, kw, c.async_replay)
    c.replay_eps, c.replay_alpha = 0.01, 0.5
    c.replay_beta = d.LinearSchedule(0.4, 1.0, 2e7)
    c.state_normalizer = d.ImageNormalizer()
    c.reward_normalizer = d.SignNormalizer()
    c.target_network_update_freq = 10000
    c.exploration_steps = 300          # the update phase is what is timed
    c.sgd_update_frequency = 4
    c.double_q = False
    c.async_actor = bool(device) if async_actor is None else bool(async_actor)   # the reference's default for the pixel DQN family
    c.max_steps = int(2e7)
    return agent_cls(c), dict(env_per_step=4, updates_per_step=1)


def a2c_pixel(workers=16, device=True, **switches):
    c = d.Config()
    c.merge(dict(game="BreakoutNoFrameskip-v4", log_level=0, tag="bench", device_env=device, **switches))
    c.num_workers = workers
    c.task_fn = lambda: d.Task(c.game, num_envs=c.num_workers, seed=1)
    c.eval_env = d.Task(c.game, seed=2)
    c.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-4, alpha=0.99, eps=1e-5)
    c.network_fn = lambda: d.CategoricalActorCriticNet(c.state_dim, c.action_dim, d.NatureConvBody())
    c.state_normalizer = d.ImageNormalizer()
    c.reward_normalizer = d.SignNormalizer()
    c.discount, c.use_gae, c.gae_tau, c.entropy_weight, c.rollout_length, c.gradient_clip = 0.99, True, 1.0, 0.01, 5, 5
    c.max_steps = int(2e7)
    return d.A2CAgent(c), dict(env_per_step=5 * workers, updates_per_step=1)


def ppo_continuous(workers=1, device=True, fused=True):
    c = d.Config()
    c.merge(dict(game="synthetic-continuous-HalfCheetah", log_level=0, tag="bench", device_env=device, fused_ppo_mlp=fused))
    c.num_workers = workers
    c.task_fn = lambda: d.Task(c.game, num_envs=c.num_workers, seed=1)
    c.eval_env = d.Task(c.game, seed=2)
    c.network_fn = lambda: d.GaussianActorCriticNet(c.state_dim, c.action_dim, actor_body=d.FCBody(c.state_dim, gate=torch.tanh),
                                                    critic_body=d.FCBody(c.state_dim, gate=torch.tanh))
    c.actor_opt_fn = lambda p: torch.optim.Adam(p, 3e-4)
    c.critic_opt_fn = lambda p: torch.optim.Adam(p, 1e-3)
    c.discount, c.use_gae, c.gae_tau, c.gradient_clip = 0.99, True, 0.95, 0.5
    c.rollout_length, c.optimization_epochs, c.mini_batch_size, c.ppo_ratio_clip = 2048, 10, 64, 0.2
    c.log_interval, c.max_steps, c.target_kl = 2048, 3e6, 0.01
    c.state_normalizer = d.MeanStdNormalizer()
    n_mb = 10 * (2048 * workers // 64)
    return d.PPOAgent(c), dict(env_per_step=2048 * workers, updates_per_step=n_mb)


def ppo_pixel(workers=8, device=True, **switches):
    c = d.Config()
    c.merge(dict(game="BreakoutNoFrameskip-v4", log_level=0, tag="bench", skip=False, device_env=device, **switches))
    c.num_workers = workers
    c.task_fn = lambda: d.Task(c.game, num_envs=c.num_workers, seed=1)
    c.eval_env = d.Task(c.game, seed=2)
    c.optimizer_fn = lambda p: torch.optim.Adam(p, lr=2.5e-4)
    c.network_fn = lambda: d.CategoricalActorCriticNet(c.state_dim, c.action_dim, d.NatureConvBody())
    c.state_normalizer = d.ImageNormalizer()
    c.reward_normalizer = d.SignNormalizer()
    c.discount, c.use_gae, c.gae_tau, c.entropy_weight, c.gradient_clip = 0.99, True, 0.95, 0.01, 0.5
    c.rollout_length, c.optimization_epochs = 128, 4
    c.mini_batch_size = c.rollout_length * c.num_workers // 4
    c.ppo_ratio_clip, c.shared_repr, c.max_steps = 0.1, True, int(2e7)
    c.log_interval = c.rollout_length * c.num_workers
    return d.PPOAgent(c), dict(env_per_step=128 * workers, updates_per_step=16)


CASES = {
    "dqn_pixel_uniform": lambda: dqn_family("dqn", d.UniformReplay),                      # fused learner attached
    "dqn_pixel_uniform_host_async": lambda: dqn_family("dqn", d.UniformReplay, async_actor=True),   # host emulator, async actor
    "dqn_pixel_uniform_generic": lambda: dqn_family("dqn", d.UniformReplay, fused=False),  # autograd path
    "dqn_pixel_per": lambda: dqn_family("dqn", d.PrioritizedReplay),
    "c51_pixel_uniform": lambda: dqn_family("c51", d.UniformReplay),
    "c51_pixel_per": lambda: dqn_family("c51", d.PrioritizedReplay),
    "qr_dqn_pixel_uniform": lambda: dqn_family("qr", d.UniformReplay),
    "dqn_pixel_per_device": lambda: dqn_family("dqn", d.PrioritizedReplay, device=True),
    "dqn_pixel_per_device_sync": lambda: dqn_family("dqn", d.PrioritizedReplay, device=True, async_actor=False),
    "dqn_pixel_uniform_device": lambda: dqn_family("dqn", d.UniformReplay, device=True),
    "c51_pixel_uniform_device": lambda: dqn_family("c51", d.UniformReplay, device=True),
    "c51_pixel_per_device": lambda: dqn_family("c51", d.PrioritizedReplay, device=True),
    "qr_dqn_pixel_uniform_device": lambda: dqn_family("qr", d.UniformReplay, device=True),
    "a2c_pixel_16": lambda: a2c_pixel(16),                      # device-resident environments (the default)
    "a2c_pixel_16_host": lambda: a2c_pixel(16, device=False),   # host emulators
    "a2c_pixel_16_recompute": lambda: a2c_pixel(16, reuse_rollout_activations=False),   # the update recomputes conv1-3 (rounds 2-5)
    "ppo_pixel_8": lambda: ppo_pixel(8),
    "a2c_pixel_16_nofc4head": lambda: a2c_pixel(16, fuse_fc4_head=False),      # fc4 finish / policy head as separate autograd nodes
    "ppo_pixel_8_nofc4head": lambda: ppo_pixel(8, fuse_fc4_head=False),
    "a2c_pixel_16_gemv": lambda: a2c_pixel(16, rollout_fc4_slices=False),       # rollout fc4 as the eight-wave GEMV (module path)
    "ppo_pixel_8_gemv": lambda: ppo_pixel(8, rollout_fc4_slices=False),
    "a2c_pixel_16_modules": lambda: a2c_pixel(16, fused_rollout=False),         # rollout through network.forward (5 launches / step)
    "ppo_pixel_8_modules": lambda: ppo_pixel(8, fused_rollout=False),
    "ppo_pixel_8_host": lambda: ppo_pixel(8, device=False),
    "ppo_continuous_1": lambda: ppo_continuous(1),
    "ppo_continuous_16": lambda: ppo_continuous(16),                                  # device environments + persistent kernels
    "ppo_continuous_16_host": lambda: ppo_continuous(16, device=False),               # host environments, persistent update kernel
    "ppo_continuous_16_generic": lambda: ppo_continuous(16, device=False, fused=False),   # round-4 path
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--seconds", type=float, default=6.0)
    a = ap.parse_args()
    agents_mod.get_logger = lambda *x, **k: _Quiet()
    d.select_device(0)
    d.random_seed(0)
    for name in a.cases.split(","):
        try:
            agent, meta = CASES[name]()
            warm = 120 if "dqn" in name or "c51" in name else 4      # DQN family: past exploration_steps; on-policy: past graph capture
            for _ in range(warm):
                agent.step()
            torch.cuda.synchronize()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < a.seconds or n < 2:
                agent.step()
                n += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"case": name, "agent_steps": n, "seconds": round(dt, 2), "agent_steps_per_s": round(n / dt, 2),
                              "env_steps_per_s": round(n * meta["env_per_step"] / dt, 1),
                              "updates_per_s": round(n * meta["updates_per_step"] / dt, 1)}), flush=True)
            if hasattr(agent, "close"):
                agent.close()
            if os.environ.get("BENCH_AGENTS_SETTLE"):
                del agent
                import gc
                gc.collect()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        except Exception as e:
            import traceback
            print(json.dumps({"case": name, "error": repr(e), "trace": traceback.format_exc()[-600:]}), flush=True)


if __name__ == "__main__":
    main()
