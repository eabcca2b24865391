"""The drop-in surface END TO END on the GPU: example entry points executed through the INTEGRATION.md launcher
(deeprl_amd.launch) and run_steps (deep_rl/utils/misc.py:19-35) for a few hundred steps.  The entry points come from
deeprl_amd/zoo.py, whose configurations tests/test_zoo_vs_reference.py pins to the reference's examples.py field by
field (the reference tree itself does not exist on the GPU box)."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Log:
    def __init__(self):
        self.lines, self.scalars = [], []

    def info(self, msg, *a, **k):
        self.lines.append(msg)

    def add_scalar(self, tag, value, step=None, log_level=0):
        self.scalars.append((tag, float(value), step))

    def add_histogram(self, *a, **k):
        pass


@pytest.fixture()
def dra(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import deeprl_amd as d
    import deeprl_amd.agents as agents_mod
    d.select_device(0)
    log = _Log()
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: log)
    d._test_log = log
    return d


def _zoo_module():
    from deeprl_amd import launch
    import deeprl_amd.zoo as zoo
    return launch.load_examples(zoo.__file__, "zoo_examples")


@pytest.mark.parametrize("async_actor", [True, False])
def test_launcher_runs_dqn_pixel_through_run_steps(dra, async_actor, monkeypatch):
    """examples.py::dqn_pixel (examples.py:55-97: async_actor=True, 1M-frame replay) through launch.run_entry + run_steps:
    the environment, actor, replay and learner all live on the device (DeviceActorPipeline); async_actor=True is the
    two-stream pipeline, False the in-order mode.  600 agent steps = 2400 environment steps, 100 of them exploration."""
    d = dra
    from deeprl_amd import launch
    from deeprl_amd.envs import synthetic_frame
    mod = _zoo_module()
    d.random_seed(1)
    seen = {}
    real_close = d.DQNAgent.close

    def close(agent):                       # run_steps closes the agent at max_steps (misc.py:31-33): look first
        agent._learner.synchronize()
        rp = agent.replay.replay
        seen.update(size=rp.size(), pos=rp.pos, pipe=agent._pipe, learner=agent._learner is not None,
                    frames=d.ops._wrap_device_pointer(rp._ring.pointers()[0], 8 * 7056, torch.uint8).cpu().numpy().reshape(8, 7056),
                    acts=d.ops._wrap_device_pointer(rp._ring.pointers()[1], 2400, torch.int64).cpu().numpy().copy(),
                    finite=all(bool(torch.isfinite(v).all()) for v in agent.network.state_dict().values()),
                    moved=float((agent.network.state_dict()["fc_head.weight"] - agent.target_network.state_dict()["fc_head.weight"]).abs().max()))
        real_close(agent)

    monkeypatch.setattr(d.DQNAgent, "close", close)
    t0 = time.time()
    agent = launch.run_entry(mod, "dqn_pixel", max_steps=2400, game="synthetic-atari",
                             overrides=dict(exploration_steps=100, target_network_update_freq=10000, async_actor=async_actor,
                                            log_interval=800, save_interval=0))
    dt = time.time() - t0
    assert agent.total_steps == 2400
    assert seen["pipe"] is not None and seen["pipe"].async_actor == async_actor and seen["learner"]
    assert seen["size"] == 2400 and seen["pos"] == 2400 and seen["finite"]
    assert seen["moved"] > 0, "(2400 - 100) / 4 updates moved the online network away from the target network"
    env_seed = seen["pipe"].stream.seed
    for i in range(8):                      # the ring holds the documented counter-hash frames ...
        assert np.array_equal(seen["frames"][i], synthetic_frame(i, env_seed))
    assert ((seen["acts"] >= 0) & (seen["acts"] < 4)).all()      # ... and valid actions
    log = d._test_log
    assert any("steps/s" in ln for ln in log.lines), "run_steps logged its throughput line (misc.py:26-28)"
    assert any(tag == "episodic_return_train" for tag, _, _ in log.scalars), "episode returns were recorded"
    print("dqn_pixel async_actor=%s: %.0f env steps/s through run_steps" % (async_actor, 2400 / dt))


def test_launcher_runs_ppo_pixel_through_run_steps(dra):
    """examples.py::ppo_pixel (examples.py:525-550: 8 workers, rollouts of 128, 4 epochs x 4 minibatches) for two
    rollouts through the launcher + run_steps."""
    d = dra
    from deeprl_amd import launch
    mod = _zoo_module()
    d.random_seed(2)
    agent = launch.run_entry(mod, "ppo_pixel", max_steps=2048, game="synthetic-atari", overrides=dict(save_interval=0))
    assert agent.total_steps == 2048
    assert all(torch.isfinite(v).all() for v in agent.network.state_dict().values())


def test_launcher_runs_a2c_pixel_through_run_steps(dra):
    """examples.py::a2c_pixel (examples.py:361-381: 16 workers, rollouts of 5)."""
    d = dra
    from deeprl_amd import launch
    mod = _zoo_module()
    d.random_seed(3)
    agent = launch.run_entry(mod, "a2c_pixel", max_steps=1600, game="synthetic-atari", overrides=dict(save_interval=0))
    assert agent.total_steps == 1600
    assert all(torch.isfinite(v).all() for v in agent.network.state_dict().values())
