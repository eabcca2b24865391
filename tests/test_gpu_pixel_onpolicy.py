"""BASELINE configs[4] shapes at update level against the reference's own A2CAgent.step / PPOAgent.step on
CategoricalActorCriticNet(NatureConvBody) (tests/golden/pixel_onpolicy.npz, generated from the untouched reference by
tests/golden/make_golden.py::gen_pixel_onpolicy).  The rollout is replayed -- the four synthetic Atari emulators are stepped
with the reference's recorded actions, which regenerates its uint8 [N,4,84,84] states -- and every per-step
log-probability / value, the GAE outputs and the parameters after the update(s) must match."""
import os
import sys
from collections import namedtuple

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_envs  # noqa: E402
from golden.make_golden_cases import digest  # noqa: E402

AC_SHAPES = fake_envs.NATURE_SHAPES_PREFIXED("phi_body.") + [
    ("fc_action.weight", (4, 512)), ("fc_action.bias", (4,)), ("fc_critic.weight", (1, 512)), ("fc_critic.bias", (1,))]


class _Quiet:
    def info(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


@pytest.fixture(scope="module")
def dra():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need an MI355X")
    import deeprl_amd as d
    d.select_device(0)
    return d


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pixel_onpolicy.npz"))


def _cmp_digests(net, g, prefix, rtol, atol):
    for name, v in net.state_dict().items():
        got, want = digest(v.detach().cpu().numpy()), g[prefix + name]
        np.testing.assert_allclose(got[2:], want[2:], rtol=rtol, atol=atol, err_msg=name)
        np.testing.assert_allclose(got[:2], want[:2], rtol=1e-4, atol=1e-4, err_msg=name + " (sums)")


def _states(task, actions):
    """The reference's rollout states: reset, then one env step per recorded action row."""
    out = [task.reset()]
    for a in actions:
        out.append(task.step(np.asarray(a).reshape(-1))[0])
    return out


def test_a2c_pixel_update_matches_reference(dra, g):
    d = dra
    dev = d.Config.DEVICE
    k = "a2c_"
    t_len, n_env = 5, 4
    net = d.CategoricalActorCriticNet((4, 84, 84), 4, d.NatureConvBody())
    net.load_state_dict({n: torch.from_numpy(v) for n, v in fake_envs.numpy_params(AC_SHAPES, 23).items()})
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-4, alpha=0.99, eps=1e-5)
    fused = d.optim.FusedOptimizer.adopt(opt)
    norm = d.ImageNormalizer()
    states = _states(fake_envs.PixelVectorTask(seed=5, num_envs=n_env, done_period=9), g[k + "action"])
    storage = d.Storage(t_len)
    for t in range(t_len):
        pred = net(norm(states[t]), torch.from_numpy(g[k + "action"][t]).to(dev))
        np.testing.assert_allclose(pred["log_pi_a"].detach().cpu().numpy(), g[k + "log_pi_a"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pred["v"].detach().cpu().numpy(), g[k + "v"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pred["entropy"].detach().cpu().numpy(), g[k + "entropy"][t], rtol=1e-5, atol=1e-5)
        storage.feed(pred)
        storage.feed({"reward": torch.from_numpy(g[k + "reward"][t]).to(dev), "mask": torch.from_numpy(g[k + "mask"][t]).to(dev)})
    boot = net(norm(states[t_len]))
    storage.feed(boot)
    storage.placeholder()
    cfg = d.Config()
    cfg.rollout_length, cfg.discount, cfg.gae_tau, cfg.use_gae = t_len, 0.99, 1.0, True
    from deeprl_amd.agents import _rollout_scan
    adv, ret = _rollout_scan(storage, cfg, boot["v"])
    np.testing.assert_allclose(adv.cpu().numpy(), g[k + "adv"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), g[k + "ret"], rtol=1e-5, atol=1e-5)
    entries = storage.extract(["log_pi_a", "v", "ret", "advantage", "entropy"])
    out4, (g_lp, g_ent, g_v) = d.ops.a2c_loss(entries.log_pi_a.detach(), entries.entropy.detach(), entries.v.detach(),
                                             entries.advantage, entries.ret, 0.01, 1.0)
    fused.zero_grad()
    torch.autograd.backward([entries.log_pi_a, entries.entropy, entries.v], [g_lp, g_ent, g_v])
    fused.step(5)
    _cmp_digests(net, g, k + "final_", 1e-5, 1e-6)


def test_ppo_pixel_optimize_matches_reference(dra, g, monkeypatch):
    """PPO_agent.py:63-99 with shared_repr=True (one Adam over the whole net): 2 epochs x 4 minibatches of 16."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    dev = d.Config.DEVICE
    k = "ppo_"
    t_len, n_env = 16, 4
    cfg = d.Config()
    cfg.merge(dict(discount=0.99, use_gae=True, gae_tau=0.95, entropy_weight=0.01, rollout_length=t_len, num_workers=n_env,
                   optimization_epochs=2, mini_batch_size=16, ppo_ratio_clip=0.1, target_kl=1e9, shared_repr=True,
                   max_steps=1e6, gradient_clip=0.5))
    cfg.task_fn = lambda: fake_envs.PixelVectorTask(seed=6, num_envs=n_env, done_period=11)
    cfg.network_fn = lambda: d.CategoricalActorCriticNet((4, 84, 84), 4, d.NatureConvBody())
    cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=2.5e-4)
    cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    agent = d.PPOAgent(cfg)
    agent.network.load_state_dict({n: torch.from_numpy(v) for n, v in fake_envs.numpy_params(AC_SHAPES, 23).items()})
    actions = g[k + "ent_action"].reshape(t_len, n_env)
    states = _states(fake_envs.PixelVectorTask(seed=6, num_envs=n_env, done_period=11), actions)
    norm = d.ImageNormalizer()
    # the rollout's values from the replayed states (PPO_agent.py:33-47) and the scan
    with torch.no_grad():
        v = torch.stack([agent.network(norm(states[t]))["v"] for t in range(t_len + 1)])
    np.testing.assert_allclose(v.cpu().numpy(), g[k + "v"], rtol=1e-5, atol=1e-5)
    adv, ret = d.ops.gae(torch.from_numpy(g[k + "reward"]).to(dev).float().squeeze(-1), torch.from_numpy(g[k + "mask"]).to(dev).float().squeeze(-1),
                         v.squeeze(-1).contiguous(), 0.99, 0.95, True)
    np.testing.assert_allclose(adv.cpu().numpy(), g[k + "adv"].squeeze(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), g[k + "ret"].squeeze(-1), rtol=1e-5, atol=1e-5)
    entry_cls = namedtuple("Entry", ["state", "action", "log_pi_a", "ret", "advantage"])
    raw_adv = torch.from_numpy(g[k + "adv"].reshape(-1, 1)).to(dev).contiguous()
    d.ops.adv_normalize_(raw_adv)
    np.testing.assert_allclose(raw_adv.cpu().numpy(), g[k + "ent_adv_normalized"], rtol=1e-5, atol=1e-5)
    state = torch.cat([torch.as_tensor(np.asarray(norm(states[t])), dtype=torch.float32) for t in range(t_len)]).to(dev)
    entries = entry_cls(state, *[torch.from_numpy(g[k + n]).to(dev) for n in ("ent_action", "ent_log_pi_a", "ent_ret")], raw_adv)
    np.random.seed(21)
    # the reference consumed np.random for nothing else before its permutations (torch.manual_seed drives the sampling)
    agent.optimize(entries)
    _cmp_digests(agent.network, g, k + "final_", 2e-4, 2e-6)
    agent.close()
