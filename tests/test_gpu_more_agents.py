"""DDPG / TD3 / OptionCritic on the HBM replay and the HIP contractions against runs of the reference's own agents
(tests/golden/make_golden.py gen_ddpg_td3 / gen_option_critic), plus the checkpoint round trip with reference-format
files (BaseAgent.py:24-33)."""
import os
import pickle
import random

import numpy as np
import pytest
import torch

import fake_envs

pytestmark = pytest.mark.gpu


class _Quiet:
    def info(self, *a, **k):
        pass
    add_scalar = add_histogram = info


@pytest.fixture()
def dra(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import deeprl_amd as d
    import deeprl_amd.agents as agents_mod
    d.select_device(0)
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    return d


def _load(module, g, prefix):
    module.load_state_dict({k: torch.from_numpy(g[prefix + k]) for k in module.state_dict().keys()})


@pytest.mark.parametrize("tag", ["ddpg", "td3"])
def test_ddpg_td3_match_reference_run(golden, dra, tag):
    """40 agent steps (10 warm-up steps of action_space.sample(), then policy + OU / Gaussian noise from np.random,
    one sampled minibatch of 8 per step, two Adam optimisers, soft target updates): same np.random consumption, the
    replay holds the same actions / rewards, online and target weights land on the reference's."""
    d = dra
    g = golden("ddpg_td3_agents")
    cfg = d.Config()
    cfg.merge(dict(game="fake", log_level=0, tag=tag))
    cfg.task_fn = lambda: fake_envs.ContinuousTask(seed=13, state_dim=5, action_dim=2, horizon=9)
    cfg.eval_env = cfg.task_fn()
    adam = lambda p: torch.optim.Adam(p, lr=1e-3)
    if tag == "ddpg":      # examples.py:555-583 at test size
        cfg.network_fn = lambda: d.DeterministicActorCriticNet(
            5, 2, actor_body=d.FCBody(5, (16, 16), gate=torch.relu), critic_body=d.FCBody(7, (16, 16), gate=torch.relu),
            actor_opt_fn=adam, critic_opt_fn=adam)
        cfg.replay_fn = lambda: d.UniformReplay(memory_size=200, batch_size=8)
        cfg.random_process_fn = lambda: d.OrnsteinUhlenbeckProcess(size=(2,), std=d.LinearSchedule(0.2))
        cls = d.DDPGAgent
    else:                  # examples.py:587-617
        cfg.network_fn = lambda: d.TD3Net(2, actor_body_fn=lambda: d.FCBody(5, (16, 16), gate=torch.relu),
                                          critic_body_fn=lambda: d.FCBody(7, (16, 16), gate=torch.relu),
                                          actor_opt_fn=adam, critic_opt_fn=adam)
        cfg.replay_fn = lambda: d.ReplayWrapper(d.UniformReplay, dict(memory_size=200, batch_size=8), False)
        cfg.random_process_fn = lambda: d.GaussianProcess(size=(2,), std=d.LinearSchedule(0.1))
        cfg.td3_noise, cfg.td3_noise_clip, cfg.td3_delay = 0.0, 0.5, 2
        cls = d.TD3Agent
    cfg.discount, cfg.warm_up, cfg.target_network_mix, cfg.max_steps = 0.99, 10, 5e-3, 1e5
    torch.manual_seed(7)
    np.random.seed(17)
    random.seed(17)
    agent = cls(cfg)
    k = tag + "_"
    _load(agent.network, g, k + "init_")
    agent.target_network.load_state_dict(agent.network.state_dict())
    for _ in range(40):
        agent.step()
    assert agent.total_steps == int(g[k + "total_steps"])
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g[k + "rng_tail"])
    rp = getattr(agent.replay, "replay", agent.replay)
    n = rp.size()
    acts = d.ops._wrap_device_pointer(rp._ring.pointers()[1], n * 2, torch.float64).cpu().numpy().reshape(n, 2)
    np.testing.assert_allclose(acts, g[k + "replay_action"], rtol=1e-5, atol=1e-6)   # policy outputs: fp32 contractions
    rews = d.ops._wrap_device_pointer(rp._ring.pointers()[2], n, torch.float64).cpu().numpy()
    assert np.array_equal(rews, g[k + "replay_reward"])
    for name, v in agent.network.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g[k + "final_" + name], rtol=2e-4, atol=2e-5, err_msg=name)
    for name, v in agent.target_network.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g[k + "target_" + name], rtol=2e-4, atol=2e-5, err_msg="target " + name)
    agent.close()


def test_option_critic_matches_reference_run(golden, dra, monkeypatch):
    """4 rollouts of 5 steps x 3 workers; the Categorical draws the reference made on torch's CPU generator (options,
    continued options, actions -- in that order per step) are replayed, so both runs take the same decisions."""
    d = dra
    g = golden("option_critic_agent")
    rec = [row for row in g["oc_samples"]]
    dev = d.Config.DEVICE

    def replay_sample(self, sample_shape=torch.Size()):
        return torch.as_tensor(rec.pop(0), device=dev)

    monkeypatch.setattr(torch.distributions.Categorical, "sample", replay_sample)
    cfg = d.Config()
    cfg.merge(dict(game="fake", log_level=0, tag="oc"))
    cfg.num_workers = 3
    cfg.task_fn = lambda: fake_envs.VectorTask(seed=5, state_dim=4, action_dim=2, horizon=7, num_envs=3)
    cfg.eval_env = fake_envs.VectorTask(seed=6, state_dim=4, action_dim=2)
    cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, 0.001)
    cfg.network_fn = lambda: d.OptionCriticNet(d.FCBody(4, hidden_units=(16,)), 2, num_options=2)
    cfg.random_option_prob = d.LinearSchedule(1.0, 0.1, 100)
    cfg.discount, cfg.target_network_update_freq, cfg.rollout_length = 0.99, 4, 5
    cfg.termination_regularizer, cfg.entropy_weight, cfg.gradient_clip = 0.01, 0.01, 5
    torch.manual_seed(9)
    np.random.seed(19)
    agent = d.OptionCriticAgent(cfg)
    _load(agent.network, g, "oc_init_")
    agent._sync_target()
    for _ in range(4):
        agent.step()
    assert not rec, "every recorded draw was consumed, in order"
    assert agent.total_steps == int(g["oc_total_steps"])
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g["oc_rng_tail"])
    for name, v in agent.network.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["oc_final_" + name], rtol=1e-4, atol=1e-5, err_msg=name)
    for name, v in agent.target_network.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["oc_target_" + name], rtol=1e-4, atol=1e-5, err_msg="target " + name)
    agent.close()


def test_checkpoint_round_trip_with_reference_format(golden, dra, tmp_path):
    """BaseAgent.save / load (BaseAgent.py:24-33): '<name>.model' = torch.save(state_dict) with the reference's key names
    and [OC,C,KH,KW] conv tensors, '<name>.stats' = pickled normaliser state.  (1) a file written exactly as the
    reference writes it (plain contiguous CPU tensors under its names -- the nature-update fixture's weights) loads into a
    DQNAgent whose parameters live in the learner's flat KOC buffer; (2) save() after updates writes tensors that are
    contiguous, in the reference's shapes, and load back bit for bit into a fresh agent; (3) with the fused learner
    attached, save() waits for the in-flight update."""
    d = dra
    cfg = d.Config()
    cfg.merge(dict(game="synthetic-atari", log_level=0, tag="ckpt", n_step=1, replay_cls=d.UniformReplay, async_replay=False))
    cfg.task_fn = lambda: d.Task(cfg.game, seed=3)
    cfg.eval_env = cfg.task_fn()
    cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
    cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
    cfg.random_action_prob = d.LinearSchedule(1.0, 0.05, 60)
    cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
    kw = dict(memory_size=400, batch_size=32, n_step=1, discount=0.99, history_length=4)
    cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
    cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 1000, 40, 4
    cfg.gradient_clip, cfg.double_q, cfg.async_actor, cfg.max_steps = 5, False, True, 1e5
    d.random_seed(5)
    agent = d.DQNAgent(cfg)
    assert agent._pipe is not None and agent._learner is not None
    # (1) a reference-format checkpoint
    ref_sd = {k: torch.from_numpy(v.copy()) for k, v in fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(4), 11).items()}
    path = str(tmp_path / "ref_ckpt")
    torch.save(ref_sd, path + ".model")
    with open(path + ".stats", "wb") as f:
        pickle.dump(None, f)                       # RescaleNormalizer.state_dict() is None (normalizer.py:62-66)
    agent.load(path)
    for k, v in agent.network.state_dict().items():
        assert torch.equal(v.cpu(), ref_sd[k]), k
    # the learner computes with the loaded weights: q of the device actor == the module's own forward
    for _ in range(30):
        agent.step()
    # (2) save after updates, reload into a fresh agent
    out = str(tmp_path / "mine")
    agent.save(out)
    saved = torch.load(out + ".model", map_location="cpu")
    assert set(saved.keys()) == set(ref_sd.keys())
    for k, v in saved.items():
        assert v.shape == ref_sd[k].shape and v.is_contiguous(), k
    live = {k: v.detach().cpu().clone() for k, v in agent.network.state_dict().items()}
    for k in live:                                # (3) the file is the state AFTER the last issued update
        assert torch.equal(saved[k], live[k]), k
    assert any(not torch.equal(saved[k], ref_sd[k]) for k in saved), "updates happened"
    d.random_seed(6)
    other = d.DQNAgent(cfg)
    other.load(out)
    for k, v in other.network.state_dict().items():
        assert torch.equal(v.cpu(), saved[k]), k
    other.step()                                    # and keeps running on them
    agent.close()
    other.close()
