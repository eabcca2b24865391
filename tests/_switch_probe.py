"""Helper of tests/test_gpu_env_switches.py: runs a short device-resident, async-actor DQN-family agent under whatever DRA_* environment
switches the parent set and saves what the run produced (stored actions / rewards, final parameters).  The library reads those
switches once per process, hence one process per setting."""
import os
import random
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
sys.path.insert(0, _HERE)
import deeprl_amd as d  # noqa: E402
import deeprl_amd.agents as agents_mod  # noqa: E402
import fake_envs  # noqa: E402


class _Quiet:
    def info(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


def main(kind, out):
    agents_mod.get_logger = lambda *a, **k: _Quiet()
    d.select_device(0)
    per = kind.endswith("_per")          # "dqn_per" / "c51_per": PrioritizedReplay (the device-side draw and its switches)
    kind = kind.replace("_per", "")
    cfg = d.Config()
    cfg.merge(dict(game="synthetic-atari", n_step=1, replay_cls=d.PrioritizedReplay if per else d.UniformReplay, async_replay=False,
                   log_level=0, tag="probe", device_env=True))
    cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
    cfg.replay_beta = d.LinearSchedule(0.4, 1.0, 1000)
    cfg.task_fn = lambda: d.Task(cfg.game, seed=9, synthetic_done_period=13)
    cfg.eval_env = cfg.task_fn()
    if kind == "dqn":
        cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
        cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
        cls, head = d.DQNAgent, [("fc_head.weight", (4, 512)), ("fc_head.bias", (4,))]
    elif kind == "c51":
        cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=0.00025, eps=0.01 / 32)
        cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
        cfg.network_fn = lambda: d.CategoricalNet(cfg.action_dim, cfg.categorical_n_atoms, d.NatureConvBody())
        cls, head = d.CategoricalDQNAgent, [("fc_categorical.weight", (4 * 51, 512)), ("fc_categorical.bias", (4 * 51,))]
    else:
        cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=0.00005, eps=0.01 / 32)
        cfg.num_quantiles = 200
        cfg.network_fn = lambda: d.QuantileNet(cfg.action_dim, cfg.num_quantiles, d.NatureConvBody())
        cls, head = d.QuantileRegressionDQNAgent, [("fc_quantiles.weight", (4 * 200, 512)), ("fc_quantiles.bias", (4 * 200,))]
    cfg.random_action_prob = d.LinearSchedule(0.05, 0.05, 10)          # mostly greedy: the actor's action values matter
    cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
    kw = dict(memory_size=300, batch_size=32, n_step=1, discount=0.99, history_length=4)
    cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
    cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 5, 40, 4
    cfg.gradient_clip, cfg.double_q, cfg.async_actor, cfg.max_steps = 5, False, True, 1e5
    d.random_seed(3)
    random.seed(3)
    agent = cls(cfg)
    assert agent._pipe is not None and agent._pipe.async_actor
    agent._pipe.rs = np.random.RandomState(77)
    np.random.seed(5)
    p_np = fake_envs.numpy_params(fake_envs.NATURE_SHAPES + head, 17)
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
    agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
    agent._learner.invalidate_actor_copy()
    for _ in range(int(os.environ.get("PROBE_STEPS", "60"))):
        agent.step()
    agent.sync_host()
    agent._learner.synchronize()
    torch.cuda.synchronize()
    rp = agent.replay.replay
    frames, actions, rewards, masks = rp._ring.pointers()
    w = d.ops._wrap_device_pointer
    res = {"act": w(actions, 300, torch.int64).cpu().numpy().copy(), "rew": w(rewards, 300, torch.float64).cpu().numpy().copy()}
    for k, v in agent.network.state_dict().items():
        res["p_" + k] = v.detach().cpu().numpy().copy()
    if per:
        res["tree"] = rp.tree.as_tensor().cpu().numpy().copy()
        res["py_rng"] = np.asarray([random.getrandbits(30) for _ in range(2)], dtype=np.int64)
    np.savez(out, **res)
    agent.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
