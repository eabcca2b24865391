"""deeprl_amd/zoo.py (the example configurations as data) against the reference's own examples.py (CPU, no GPU work):
each reference entry point is executed through deeprl_amd.launch's loader with the agent constructor and run_steps
replaced by a capture, and the Config it built is compared field by field with zoo.config(name).  Also: the launcher's
loader / run_entry on the package's own zoo file.  Skipped where /root/reference is absent (the GPU box)."""
import os

import numpy as np
import pytest
import torch

REF = os.environ.get("DEEPRL_REFERENCE_ROOT", "/root/reference")
GAMES = {"dqn_feature": "CartPole-v0", "dqn_pixel": "BreakoutNoFrameskip-v4",
         "quantile_regression_dqn_pixel": "BreakoutNoFrameskip-v4", "categorical_dqn_pixel": "BreakoutNoFrameskip-v4",
         "a2c_pixel": "BreakoutNoFrameskip-v4", "ppo_pixel": "BreakoutNoFrameskip-v4", "ppo_continuous": "HalfCheetah-v2"}
AGENTS = ["DQNAgent", "CategoricalDQNAgent", "QuantileRegressionDQNAgent", "A2CAgent", "PPOAgent", "NStepDQNAgent",
          "OptionCriticAgent", "DDPGAgent", "TD3Agent"]
PLAIN = (int, float, bool, str, type(None))


def _schedule(s):
    return None if s is None else (type(s).__name__, getattr(s, "inc", None), getattr(s, "current", None), getattr(s, "end", None),
                                   getattr(s, "val", None))


def _describe(cfg):
    """Everything comparable about a Config without a GPU."""
    out = {k: v for k, v in vars(cfg).items() if isinstance(v, PLAIN) and not k.startswith("_")}
    out.pop("tag", None)
    out["eps_schedule"] = _schedule(cfg.random_action_prob)
    out["beta_schedule"] = _schedule(getattr(cfg, "replay_beta", None))
    out["normalizers"] = (type(cfg.state_normalizer).__name__, getattr(cfg.state_normalizer, "coef", None),
                          type(cfg.reward_normalizer).__name__)
    out["replay_cls"] = getattr(getattr(cfg, "replay_cls", None), "__name__", None)
    dummy = [torch.nn.Parameter(torch.zeros(3))]
    for name in ("optimizer_fn", "actor_opt_fn", "critic_opt_fn"):
        fn = getattr(cfg, name, None)
        if fn is not None:
            opt = fn(dummy)
            out[name] = (type(opt).__name__, {k: v for k, v in opt.defaults.items() if isinstance(v, (int, float, bool, tuple))})
    if cfg.replay_fn is not None:
        rw = cfg.replay_fn()
        out["replay"] = (rw.replay_cls.__name__, dict(rw.replay_kwargs), bool(rw.async_))
    net = cfg.network_fn()
    out["network"] = (type(net).__name__, [(k, tuple(v.shape)) for k, v in net.state_dict().items()])
    task = cfg.task_fn()
    out["task"] = (task.name, task.state_dim, task.action_dim, getattr(getattr(task, "env", None), "num_envs", None))
    out["eval_env"] = (cfg.eval_env.name, getattr(getattr(cfg.eval_env, "env", None), "num_envs", None))
    return out


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "examples.py")), reason="reference tree not present")
@pytest.mark.parametrize("name", sorted(GAMES))
def test_zoo_entry_equals_reference_example(name):
    import sys
    import deeprl_amd as d
    from deeprl_amd import launch, zoo
    saved = {k: v for k, v in sys.modules.items() if k == "deep_rl" or k.startswith("deep_rl.")}
    d.select_device(-1)
    try:
        mod = launch.load_examples(os.path.join(REF, "examples.py"), "ref_examples_zoo")
        got = {}
        for a in AGENTS:
            setattr(mod, a, lambda cfg, _a=a: (_a, cfg))
        mod.run_steps = lambda pair: got.update(agent=pair[0], cfg=pair[1])
        np.random.seed(0)
        getattr(mod, name)(game=GAMES[name])
        assert got["agent"] == zoo.ZOO[name]["agent"]
        np.random.seed(0)
        mine = zoo.config(name, game=GAMES[name])
        want, have = _describe(got["cfg"]), _describe(mine)
        have.pop("replay_kwargs", None)
        want.pop("replay_kwargs", None)
        assert set(want) == set(have), (sorted(set(want) ^ set(have)))
        for k in sorted(want):
            assert want[k] == have[k], "%s.%s: reference %r, zoo %r" % (name, k, want[k], have[k])
    finally:
        for k in [k for k in sys.modules if k == "deep_rl" or k.startswith("deep_rl.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_launcher_runs_an_entry_with_capped_steps(tmp_path):
    """load_examples + run_entry on a deep_rl-style file: `async=` is rewritten, deep_rl resolves to the package, and
    max_steps reaches the agent's config before run_steps sees it."""
    import sys
    from deeprl_amd import launch
    saved = {k: v for k, v in sys.modules.items() if k == "deep_rl" or k.startswith("deep_rl.")}
    src = tmp_path / "my_examples.py"
    src.write_text("from deep_rl import *\n"
                   "class FakeAgent:\n"
                   "    def __init__(self, config):\n"
                   "        self.config, self.total_steps, self.closed = config, 0, False\n"
                   "        self.logger = type('L', (), {'info': lambda *a, **k: None})()\n"
                   "    def step(self):\n        self.total_steps += 1\n"
                   "    def switch_task(self):\n        pass\n"
                   "    def close(self):\n        self.closed = True\n"
                   "def entry(**kwargs):\n"
                   "    config = Config()\n    config.merge(kwargs)\n    config.max_steps = int(2e7)\n    config.log_interval = 0\n"
                   "    w = ReplayWrapper(UniformReplay, dict(memory_size=8, batch_size=2), async=True)\n"
                   "    assert w.async_ is True\n"
                   "    run_steps(FakeAgent(config))\n")
    try:
        mod = launch.load_examples(str(src))
        agent = launch.run_entry(mod, "entry", max_steps=37, game="x")
        assert agent.total_steps == 37 and agent.closed and agent.config.game == "x"
    finally:
        for k in [k for k in sys.modules if k == "deep_rl" or k.startswith("deep_rl.")]:
            del sys.modules[k]
        sys.modules.update(saved)
