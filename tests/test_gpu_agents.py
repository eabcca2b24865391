"""GPU parity tests at the agent level: the drop-in surface (Config / Agent.step()) on the HIP
kernels against fixtures produced by the reference's own agents (tests/golden/make_golden.py)."""
import os
import random

import numpy as np
import pytest
import torch

import fake_envs
from parity_log import record_parity as _record_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dra():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need an MI355X")
    import deeprl_amd as d
    d.select_device(0)
    return d


class _Quiet:
    def info(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


def _load(module, arrays, prefix):
    sd = {k: torch.from_numpy(arrays[prefix + k]) for k in module.state_dict().keys()}
    module.load_state_dict(sd)


def _cmp_params(module, g, prefix, rtol, atol):
    for k, v in module.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g[prefix + k], rtol=rtol, atol=atol, err_msg=k)


@pytest.mark.parametrize("tag,per,n_step", [("uniform", False, 1), ("per_n3", True, 3)])
def test_dqn_agent_config1_matches_reference_run(golden, dra, tag, per, n_step, monkeypatch):
    """BASELINE config 1 (examples.py dqn_feature on a CartPole-shaped env, sync replay / actor):
    60 agent.step() calls reproduce the reference's action stream, RNG consumption and weights."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    g = golden("dqn_agent_cartpole")
    cfg = d.Config()
    replay_cls = d.PrioritizedReplay if per else d.UniformReplay
    cfg.merge(dict(game="fake", n_step=n_step, replay_cls=replay_cls, async_replay=False, log_level=0, tag=tag))
    cfg.task_fn = lambda: fake_envs.VectorTask(seed=1, state_dim=4, action_dim=2, horizon=20)
    cfg.eval_env = cfg.task_fn()
    cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, 0.001)
    cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.FCBody(cfg.state_dim))
    cfg.history_length = 1
    cfg.batch_size = 10
    cfg.discount = 0.99
    cfg.max_steps = 1e5
    kw = dict(memory_size=int(1e4), batch_size=cfg.batch_size, n_step=cfg.n_step, discount=cfg.discount,
              history_length=cfg.history_length)
    cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
    cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
    cfg.replay_beta = d.LinearSchedule(0.4, 1.0, cfg.max_steps)
    cfg.random_action_prob = d.LinearSchedule(1.0, 0.1, 100)
    cfg.target_network_update_freq = 5
    cfg.exploration_steps = 40
    cfg.double_q = False
    cfg.sgd_update_frequency = 4
    cfg.gradient_clip = 5
    cfg.async_actor = False
    d.random_seed(0)
    random.seed(0)
    agent = d.DQNAgent(cfg)
    k = tag + "_"
    # same seeds -> same orthogonal init up to LAPACK's QR rounding on this host's CPU (1 ulp between
    # machines); then start both runs from the reference's exact initial weights
    _cmp_params(agent.network, g, k + "init_", 1e-5, 1e-6)
    _load(agent.network, g, k + "init_")
    agent.sync_target()
    for _ in range(60):
        agent.step()
    assert agent.total_steps == int(g[k + "total_steps"])
    rp = agent.replay.replay
    n = rp.size()
    # the replay ring holds the whole action / state history of the run (size < capacity)
    ring_actions = d.ops._wrap_device_pointer(rp._ring.pointers()[1], n, torch.int64).cpu().numpy()
    assert np.array_equal(ring_actions, g[k + "replay_action"])
    ring_states = d.ops._wrap_device_pointer(rp._ring.pointers()[0], n * 4, torch.float64).cpu().numpy().reshape(n, 4)
    assert np.array_equal(ring_states, g[k + "replay_state"])
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g[k + "rng_tail"])
    _cmp_params(agent.network, g, k + "final_", 2e-4, 2e-5)
    _cmp_params(agent.target_network, g, k + "target_", 2e-4, 2e-5)
    agent.close()


def test_dqn_nature_update_matches_reference(golden, dra):
    """Three full DQN updates on VanillaNet(NatureConvBody) (BASELINE config 2 shapes, B=8): q values,
    every gradient, the clipped norm, and the weights after centered RMSprop."""
    d = dra
    g = golden("dqn_nature_update")
    b, a = 8, 4
    rs = np.random.RandomState(int(g["state_seed"]))
    net = d.VanillaNet(a, d.NatureConvBody())
    tgt = d.VanillaNet(a, d.NatureConvBody())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 11).items()})
    tgt.load_state_dict({k: torch.from_numpy(v) for k, v in fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 12).items()})
    dev = d.Config.DEVICE
    state = rs.randint(0, 256, size=(b, 4, 84, 84)).astype(np.uint8)
    next_state = rs.randint(0, 256, size=(b, 4, 84, 84)).astype(np.uint8)
    tr = d.Transition(state=torch.from_numpy(state).to(dev), action=torch.from_numpy(rs.randint(0, a, size=b).astype(np.int64)).to(dev),
                      reward=torch.from_numpy(np.sign(rs.standard_normal(b))).to(dev), next_state=torch.from_numpy(next_state).to(dev),
                      mask=torch.from_numpy((rs.rand(b) > 0.2).astype(np.int32)).to(dev))

    class Agent(d.DQNAgent):
        def __init__(self):  # bypass env / replay construction: only the learner is under test
            cfg = d.Config()
            cfg.discount, cfg.n_step, cfg.double_q, cfg.gradient_clip = 0.99, 1, False, 5
            cfg.state_normalizer = d.ImageNormalizer()
            cfg.lock = agents_lock()
            self.config = cfg
            self.network, self.target_network = net, tgt
            self.optimizer = torch.optim.RMSprop(net.parameters(), lr=0.00025, alpha=0.95, eps=0.01, centered=True)
            self._fused = d.optim.FusedOptimizer.adopt(self.optimizer)

    from deeprl_amd.agents import _NullLock as agents_lock
    import deeprl_amd.optim  # noqa: F401
    agent = Agent()
    with torch.no_grad():
        q0 = net(agent.config.state_normalizer(tr.state))["q"]
    np.testing.assert_allclose(q0.cpu().numpy(), g["q0"], rtol=1e-5, atol=1e-5)
    for it in range(3):
        out = agent._learn(tr)
        want_loss, want_norm = g["loss_gradnorm_traj"][it]
        np.testing.assert_allclose(out["loss"].item(), want_loss, rtol=2e-5)
        np.testing.assert_allclose(agent._fused.norm.item(), want_norm, rtol=5e-5)
        if it == 0:
            for name, p in net.named_parameters():
                got = p.grad.cpu().numpy()
                if name == "body.fc4.weight":
                    np.testing.assert_allclose(got[::37], g["grad_" + name + "_rows"], rtol=1e-4, atol=2e-6)
                else:
                    np.testing.assert_allclose(got, g["grad_" + name], rtol=1e-4, atol=2e-6, err_msg=name)
    for name, v in net.state_dict().items():
        got = v.cpu().numpy()
        want = g["final_" + name + "_rows"] if name == "body.fc4.weight" else g["final_" + name]
        got = got[::37] if name == "body.fc4.weight" else got
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=name)
    with torch.no_grad():
        qf = net(agent.config.state_normalizer(tr.state))["q"]
    np.testing.assert_allclose(qf.cpu().numpy(), g["q_final"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["t5n16", "t5n16gae", "t20n3gae"])
def test_a2c_update_matches_reference(golden, dra, tag):
    """A2C_agent.py:22-64 on the reference's own rollout: states / actions replayed, then the scan,
    the fused loss, backward through the HIP GEMMs, clip and RMSprop must land on the same weights."""
    d = dra
    g = golden("a2c_step")
    k = tag + "_"
    gamma, tau, use_gae, ew, vw, clip, t_len, n_env = g[k + "cfg"]
    t_len, n_env = int(t_len), int(n_env)
    net = d.CategoricalActorCriticNet(6, 3, d.FCBody(6, hidden_units=(32,)))
    _load(net, g, k + "init_")
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, alpha=0.99, eps=1e-5)
    fused = d.optim.FusedOptimizer.adopt(opt)
    dev = d.Config.DEVICE
    storage = d.Storage(t_len)
    for t in range(t_len):
        pred = net(g[k + "states"][t], torch.from_numpy(g[k + "action"][t]).to(dev))
        np.testing.assert_allclose(pred["log_pi_a"].detach().cpu().numpy(), g[k + "log_pi_a"][t], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pred["v"].detach().cpu().numpy(), g[k + "v"][t], rtol=1e-5, atol=1e-5)
        storage.feed(pred)
        storage.feed({"reward": torch.from_numpy(g[k + "reward"][t]).to(dev), "mask": torch.from_numpy(g[k + "mask"][t]).to(dev)})
    boot = net(g[k + "states"][t_len])
    storage.feed(boot)
    storage.placeholder()
    cfg = d.Config()
    cfg.rollout_length, cfg.discount, cfg.gae_tau, cfg.use_gae = t_len, gamma, tau, bool(use_gae)
    from deeprl_amd.agents import _rollout_scan
    adv, ret = _rollout_scan(storage, cfg, boot["v"])
    np.testing.assert_allclose(adv.cpu().numpy(), g[k + "adv"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), g[k + "ret"], rtol=1e-5, atol=1e-5)
    entries = storage.extract(["log_pi_a", "v", "ret", "advantage", "entropy"])
    out4, (g_lp, g_ent, g_v) = d.ops.a2c_loss(entries.log_pi_a.detach(), entries.entropy.detach(), entries.v.detach(),
                                             entries.advantage, entries.ret, ew, vw)
    fused.zero_grad()
    torch.autograd.backward([entries.log_pi_a, entries.entropy, entries.v], [g_lp, g_ent, g_v])
    fused.step(clip)
    _cmp_params(net, g, k + "final_", 1e-5, 1e-6)


@pytest.mark.parametrize("tag", ["t64n2", "t32n4"])
def test_ppo_optimize_matches_reference(golden, dra, tag, monkeypatch):
    """PPO_agent.py:63-99 on the reference's own rollout entries: advantage normalisation, the
    np.random minibatch permutations, the fused clip loss, the approx-KL gate and both Adam steps."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    g = golden("ppo_step")
    k = tag + "_"
    gamma, tau, ew, clip, target_kl, epochs, mb, t_len, n_env = g[k + "cfg"]
    cfg = d.Config()
    cfg.merge(dict(discount=gamma, use_gae=True, gae_tau=tau, entropy_weight=ew, rollout_length=int(t_len),
                   num_workers=int(n_env), optimization_epochs=int(epochs), mini_batch_size=int(mb),
                   ppo_ratio_clip=clip, target_kl=target_kl, shared_repr=False, max_steps=1e6, gradient_clip=0.5))
    cfg.task_fn = lambda: fake_envs.ContinuousTask(seed=9, state_dim=5, action_dim=2, horizon=25, num_envs=int(n_env))
    cfg.network_fn = lambda: d.GaussianActorCriticNet(
        5, 2, actor_body=d.FCBody(5, hidden_units=(16, 16), gate=torch.tanh),
        critic_body=d.FCBody(5, hidden_units=(16, 16), gate=torch.tanh))
    cfg.actor_opt_fn = lambda params: torch.optim.Adam(params, 3e-4)
    cfg.critic_opt_fn = lambda params: torch.optim.Adam(params, 1e-3)
    agent = d.PPOAgent(cfg)
    _load(agent.network, g, k + "init_")
    dev = d.Config.DEVICE
    from collections import namedtuple
    entry_cls = namedtuple("Entry", ["state", "action", "log_pi_a", "ret", "advantage"])
    raw_adv = torch.from_numpy(g[k + "adv"].reshape(-1, 1)).to(dev).contiguous()
    d.ops.adv_normalize_(raw_adv)
    np.testing.assert_allclose(raw_adv.cpu().numpy(), g[k + "ent_adv_normalized"], rtol=1e-5, atol=1e-5)
    entries = entry_cls(*[torch.from_numpy(g[k + n]).to(dev) for n in ("ent_state", "ent_action", "ent_log_pi_a", "ent_ret")],
                        raw_adv)
    np.random.seed(21)
    agent.optimize(entries)
    _cmp_params(agent.network, g, k + "final_", 2e-5, 2e-6)




def _rel(a, b, floor):
    """max |a - b| / max(|b|, floor): the relative error with an absolute floor for values near zero."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


@pytest.mark.parametrize("double_q,variant", [(False, 0), (False, 1), (False, 7), (False, 127), (True, 127), (True, 511),
                                              (False, 511 + 524288), (True, 511 + 524288),   # + LATE_FOLD: no norm launch
                                              (True, 511 + 262144 + 524288)])                  # (bit 262144: ignored since round 4)
def test_fused_learner_matches_oracle(dra, double_q, variant):
    """The captured-graph DQN learner (one C-ABI call per update, zero host round trips) against
    the CPU oracle's full update on identical ring contents, indices and weights: 4 consecutive
    updates (graph capture + 3 replays), B=32, then the device actor step."""
    d = dra
    from deeprl_amd.learner import DQNLearner, draw_uniform_indices
    from oracle import loss_oracle as L, net_oracle as N, numerics_oracle as NUM
    from oracle.replay_oracle import UniformReplayOracle
    from oracle.synth_oracle import synth_transitions
    cap, b, a = 6000, 32, 4
    ring = d.ops.Ring(cap, 7056, 8, 4, 1, 0.99)
    ring.fill_synthetic(0, cap, 0, 9, n_actions=a, done_period=40)
    torch.cuda.synchronize()
    frames, act, rew, msk = synth_transitions(0, cap, 7056, seed=9, n_actions=a, done_period=40)
    orc = UniformReplayOracle(cap, b, 1, 0.99, 4)
    for t in range(cap):
        orc.feed_one(frames[t].reshape(84, 84), act[t], rew[t], msk[t])
    net = d.VanillaNet(a, d.NatureConvBody())
    tgt = d.VanillaNet(a, d.NatureConvBody())
    p_np = fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 21)
    t_np = fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 22)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
    tgt.load_state_dict({k: torch.from_numpy(v) for k, v in t_np.items()})
    learner = DQNLearner(net, tgt, ring, b, a, 0.99, 5.0, 0.00025, 0.95, 0.01, centered=True, double_q=double_q,
                         variant=variant)
    p = {k: torch.tensor(v, requires_grad=True) for k, v in p_np.items()}
    pt = {k: torch.tensor(v) for k, v in t_np.items()}
    names = list(p.keys())
    sq = {k: torch.zeros_like(v) for k, v in p.items()}
    ga = {k: torch.zeros_like(v) for k, v in p.items()}
    np.random.seed(77)
    for it in range(4):
        idx = draw_uniform_indices(orc.size(), orc.pos, b, 4, 1)
        st, ac, rw, ns, mk = orc.gather(idx)
        x, xn = torch.from_numpy(NUM.image_normalize_sync(st)), torch.from_numpy(NUM.image_normalize_sync(ns))
        q = N.vanilla_head(p, N.nature_conv_body(p, x))
        with torch.no_grad():
            qn = N.vanilla_head(pt, N.nature_conv_body(pt, xn))
            qno = N.vanilla_head(p, N.nature_conv_body(p, xn)) if double_q else None   # DQN_agent.py:87-89
        delta = L.dqn_td_error(q, qn, torch.from_numpy(ac), torch.from_numpy(rw.astype(np.float32)),
                               torch.from_numpy(mk.astype(np.float32)), 0.99, q_next_online=qno)
        loss = L.dqn_reduce(delta)
        grads = torch.autograd.grad(loss, [p[k] for k in names])
        # the gradient norm of the ORACLE's fp32 gradients, accumulated in fp64: torch's own fp32 norm of 1.7M elements
        # (clip_grad_norm_, what N.clip_grad_norm restates) carries ~3e-5 of summation noise itself (measured,
        # profiles/r03_parity_errors.json), the device accumulates across lanes in fp64
        norm64 = float(np.sqrt(sum(float((g.double() ** 2).sum()) for g in grads)))
        norm, grads = N.clip_grad_norm(list(grads), 5)
        learner.update(idx, use_graph=True)
        learner.synchronize()
        # north_star's bar (fp32 within 1e-5): action values and TD errors at rtol 1e-5 with an absolute floor of
        # 1e-5 x max|q| (a TD error is a difference of action values of that scale), loss and gradient norm at 1e-5 relative
        # (measured: ~1e-7, profiles/r03_parity_errors.json)
        scale = max(1.0, float(q.detach().abs().max()))
        _record_parity("fused_learner[%s-%d] update %d" % (double_q, variant, it),
                       q=_rel(learner.q.cpu().numpy(), q.detach().numpy(), scale), td=_rel(learner.delta.cpu().numpy(), delta.detach().numpy(), scale),
                       loss=abs(learner.loss.item() - loss.item()) / abs(loss.item()), norm_vs_fp64_sum=abs(learner.norm.item() - norm64) / norm64,
                       norm_vs_torch_fp32=abs(learner.norm.item() - float(norm)) / float(norm), torch_fp32_vs_fp64_sum=abs(float(norm) - norm64) / norm64)
        np.testing.assert_allclose(learner.q.cpu().numpy(), q.detach().numpy(), rtol=1e-5, atol=1e-5 * scale)
        np.testing.assert_allclose(learner.delta.cpu().numpy(), delta.detach().numpy(), rtol=1e-5, atol=1e-5 * scale)
        np.testing.assert_allclose(learner.loss.item(), loss.item(), rtol=1e-5)
        np.testing.assert_allclose(learner.norm.item(), norm64, rtol=1e-5)
        # (a cross-check, not the bar: torch's own fp32 summation of the same 1.7 M gradient elements is 3.3e-5 away from their
        # fp64 sum -- torch_fp32_vs_fp64_sum in the recorded errors -- and its summation order depends on the host's threads)
        np.testing.assert_allclose(learner.norm.item(), float(norm), rtol=1e-4)
        with torch.no_grad():
            for k, g in zip(names, grads):
                newp, sq[k], ga[k] = N.rmsprop_step(p[k], g, sq[k], ga[k], 0.00025, 0.95, 0.01, True)
                p[k].copy_(newp)
    _record_parity("fused_learner[%s-%d] params after 4 updates" % (double_q, variant),
                   params_abs=max(float(np.abs(v.cpu().numpy() - p[k].detach().numpy()).max()) for k, v in net.state_dict().items()))
    for k, v in net.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), p[k].detach().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    # eager (non-graph) path gives the same numbers as the replayed graph
    learner.update(idx, use_graph=False)
    learner.synchronize()
    # device actor: stack ending at a slot that wraps the ring end, greedy and random branches
    for newest, eps, rnd, dice, graph in ((1, 0.5, 3, 0.9, False), (cap - 1, 0.5, 2, 0.1, True), (100, 0.0, 1, 0.0, True)):
        learner.set_env_steps([newest], [-1], [rnd], [dice], [eps])  # counter < 0: frame already in the ring
        learner.act(use_graph=graph)
        learner.synchronize()
        slots = [(newest - 3 + j) % cap for j in range(4)]
        xs = torch.from_numpy(NUM.image_normalize_sync(frames[slots].reshape(1, 4, 84, 84)))
        pp = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}
        q1 = N.vanilla_head(pp, N.nature_conv_body(pp, xs)).numpy()[0]
        np.testing.assert_allclose(learner.actor_q.cpu().numpy(), q1, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(q1).max())))
        want = rnd if dice < eps else int(np.argmax(q1))
        stored = d.ops._wrap_device_pointer(ring.pointers()[1], cap, torch.int64)[newest].item()
        assert stored == want
    learner.close()
    ring.close()


@pytest.mark.parametrize("variant", [0, 127, 1791])
def test_fused_step_sync_equals_act_then_update(dra, variant):
    """dra_dqn_learner_step in in-order mode == explicit actor transitions followed by an update:
    the synthetic frame source, the device epsilon-greedy and the captured graphs change nothing."""
    d = dra
    from deeprl_amd.learner import DQNLearner, draw_uniform_indices
    from oracle.synth_oracle import synth_transitions
    cap, b, a, seed = 3000, 32, 4, 5
    outs = []
    for mode in ("step", "manual"):
        ring = d.ops.Ring(cap, 7056, 8, 4, 1, 0.99)
        ring.fill_synthetic(0, cap, 0, seed, n_actions=a, done_period=800)
        torch.cuda.synchronize()
        net, tgt = d.VanillaNet(a, d.NatureConvBody()), d.VanillaNet(a, d.NatureConvBody())
        p_np = fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 31)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        tgt.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        L = DQNLearner(net, tgt, ring, b, a, 0.99, 5.0, 0.00025, 0.95, 0.01, centered=True, env_seed=seed,
                       variant=variant)
        np.random.seed(3)
        pos, size, counter = 0, cap, cap
        for it in range(6):
            slots = [(pos + e) % cap for e in range(4)]
            ras = [int(np.random.randint(a, size=1)[0]) for _ in range(4)]
            dices = [float(np.random.rand(1)[0]) for _ in range(4)]
            L.set_env_steps(slots, [counter + e for e in range(4)], ras, dices, [0.3] * 4)
            counter += 4
            pos = (pos + 4) % cap
            idx = draw_uniform_indices(size, pos, b, 4, 1)
            if mode == "step":
                L.step(idx, True, False)
            else:
                L.act(use_graph=False)
                L.update(idx, use_graph=False)
        L.synchronize()
        acts = d.ops._wrap_device_pointer(ring.pointers()[1], cap, torch.int64)[:24].cpu().numpy().copy()
        frames24 = d.ops._wrap_device_pointer(ring.pointers()[0], 24 * 7056, torch.uint8).cpu().numpy().copy()
        outs.append((L.flat.flat.cpu().numpy().copy(), acts, frames24))
        L.close()
        ring.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])
    # the device frame source is the documented counter hash
    want_frames, _, _, _ = synth_transitions(cap, 24, 7056, seed=seed)
    assert np.array_equal(outs[0][2].reshape(24, 7056), want_frames)


_ASYNC_RESULTS = {}


@pytest.mark.parametrize("variant", [0, 127, 255, 1023, 2047, 2559, 4607, 12799, 29183, 61951, 61951 + 131072,
                                     61951 + 1048576, 61951 + 131072 + 1048576])   # + ACTOR_MEGA
def test_fused_step_async_pipeline(dra, variant):
    """async_actor=True pipeline (actor one agent step ahead on its own stream, double-buffered actor
    parameters when variant has DRA_VAR_ACTOR_PARAMS): the transitions it feeds are the documented counter-hash
    frames, every stored action is a valid action, the TD errors stay finite, and the run is reproducible
    (two identical runs give bit-identical parameters: no race between the optimiser and the actor's reads)."""
    d = dra
    from deeprl_amd.learner import DQNLearnerBench
    from oracle.synth_oracle import synth_transitions
    outs = []
    for rep in range(2):
        d.random_seed(11)
        torch.manual_seed(5)
        np.random.seed(5)
        bench = DQNLearnerBench(ring_capacity=4000, batch=32, seed=3, actor=True, async_actor=True, variant=variant)
        # DRA_VAR_GATHER_IN_GRAPH (2048) issues the update of the previous call: 26 calls = 26 steps of transitions
        # and 25 updates, what the primed pipelines reach after 25 calls
        for _ in range(26 if variant & 2048 else 25):
            bench.step()
        bench.learner.synchronize()
        L = bench.learner
        assert torch.isfinite(L.delta).all() and torch.isfinite(L.flat.flat).all()
        frames = d.ops._wrap_device_pointer(bench.ring.pointers()[0], 100 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(bench.ring.pointers()[1], 100, torch.int64).cpu().numpy().copy()
        outs.append((L.flat.flat.cpu().numpy().copy(), frames, acts))
        L.close()
        bench.ring.close()
    # 25 agent steps wrote 100 transitions into slots 0..99 (ring was full: pos started at 0), counters 4000..4099
    want_frames, _, _, _ = synth_transitions(4000, 100, 7056, seed=3)
    assert np.array_equal(outs[0][1].reshape(100, 7056), want_frames)
    assert ((outs[0][2] >= 0) & (outs[0][2] < 4)).all()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][2], outs[1][2])
    # DRA_VAR_PIPE_GATHER moves the gather to the actor stream and the optimizer into the graph: same
    # ordering semantics as the event-based pipeline (gather(t) between actor graphs t and t+1, actor t+1 on
    # the parameters of optimizer t-1) -> bit-identical parameters and actions
    _ASYNC_RESULTS[variant] = outs[0]
    # ... and so do the 4-kernel actor step (DRA_VAR_ACTOR_V3: same arithmetic, fused launches) and the CU partition
    # ... and the gather on the update stream (DRA_VAR_GATHER_ON_UPDATE = 16384; ring capacity 4000 with 32 samples per
    # step: the host-decided 'minibatch touches the slots the next actor graph overwrites' wait fires here)
    # ... and the ring-direct update (DRA_VAR_RING_DIRECT = 32768: conv1 and the head read the replay ring, no gather)
    # ... and the prefetched minibatch indices (DRA_VAR_IDX_PREFETCH = 131072: same indices, another route)
    # ... and the actor's env step as ONE launch (DRA_VAR_ACTOR_MEGA = 1048576: the four launches' arithmetic in the same order,
    # outputs handed over through arrival counters inside the launch)
    for other in (255, 1023, 2047, 2559, 4607, 12799, 29183, 61951, 61951 + 131072, 61951 + 1048576, 61951 + 131072 + 1048576):
        if 127 in _ASYNC_RESULTS and other in _ASYNC_RESULTS:
            assert np.array_equal(_ASYNC_RESULTS[127][0], _ASYNC_RESULTS[other][0])
            assert np.array_equal(_ASYNC_RESULTS[127][2], _ASYNC_RESULTS[other][2])


@pytest.mark.parametrize("variant,init,cap", [(-1, "bench", 4000), (-1, "normal", 4000), (4607, "normal", 4000), (12799, "normal", 4000),
                                              (29183, "normal", 4000), (-1, "normal", 160), (12799, "normal", 160),
                                              (61951, "normal", 4000), (61951, "normal", 160), (61951, "bench", 4000),
                                              (193023, "normal", 4000), (193023, "normal", 160), (193023, "bench", 4000),
                                              # round 3: + LATE_FOLD (524288) on the round-2 default 193023
                                              (717311, "normal", 4000), (717311, "normal", 160), (717311, "bench", 4000),
                                              # + ACTOR_MEGA (1048576): conv3 + fc4 of the actor's env step as one launch
                                              (1241599, "normal", 4000), (1241599, "normal", 160), (1241599, "bench", 4000),
                                              (1765887, "normal", 4000), (1765887, "normal", 160), (1765887, "bench", 4000)])
def test_async_pipeline_matches_schedule_oracle(dra, variant, init, cap):
    """THE BENCHMARKED CONFIGURATION against the oracle: DQNLearnerBench(async_actor=True) with the default kernel
    variant (bench.py's: CU partition, pipelined gather, actor parameter ring, fused actor conv1) for 14 agent steps vs
    oracle/async_schedule_oracle.py, the CPU restatement of the pipeline's schedule (actor step t+1 on the parameters
    after update t-1, minibatch t gathered between actor steps t and t+1, the actor's own RandomState).

    Per step: TD errors (rtol 1e-5, atol 1e-5 x max|q|: a TD error is a difference of action values of that scale),
    loss at 1e-5 relative, parameters after the step at rtol 1e-5 / atol 2e-6 (weights are O(0.05)).  Every stored
    action must equal the oracle's (a mismatch is tolerated only where the oracle's own top-2 action values are within
    1e-5: an fp32 near-tie) and the ring frames are compared bit for bit.

    ReLU gates: when some pre-activation of the differentiated forward lies within fp32 summation noise of zero
    (oracle margin < 5e-7), two correct fp32 implementations may gate that unit differently and its whole backward
    contribution differs (measured: parameter errors of 1e-6 .. 5e-5 from ONE such unit, tests/diag_schedule.py).
    Such a step is checked at 100x the tolerances and the oracle then adopts the implementation's state, so that every
    later step starts from a common state again.  init="bench": bench.py's own initialisation (layer_init: orthogonal
    weights, zero biases -- small margins, so the oracle resynchronises after EVERY step and each step is an independent
    check); init="normal": random-normal weights and O(1) biases (tests/fake_envs.numpy_params, as in the in-order
    learner test): no resynchronisation unless a gate is ambiguous -- 14 CHAINED steps at the strict tolerances."""
    d = dra
    from deeprl_amd.learner import DQNLearnerBench
    from oracle.async_schedule_oracle import AsyncDqnScheduleOracle
    # cap = 160: nearly every minibatch touches the 4 slots the actor graph of the same call overwrites, and the 4 the previous
    # one wrote -- the host-decided cross-stream waits of the gather-on-update pipeline (csrc/learner.hip step_pipelined3)
    # are then what keeps the schedule
    b, a, seed, steps = 32, 4, 3, 14
    d.random_seed(11)
    torch.manual_seed(5)
    bench = DQNLearnerBench(ring_capacity=cap, batch=b, seed=seed, actor=True, async_actor=True, variant=variant)
    if variant < 0:
        assert bench._actor_ring and bench.learner.variant & d.ops.VAR_ACTOR_FUSED_CONV1, "default variant = bench.py's"
    if init == "normal":
        p0 = fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 21)
        bench.network.load_state_dict({k: torch.from_numpy(v) for k, v in p0.items()})
        bench.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p0.items()})
    p_np = {k: v.detach().cpu().numpy().copy() for k, v in bench.network.state_dict().items()}
    t_np = {k: v.detach().cpu().numpy().copy() for k, v in bench.target_network.state_dict().items()}
    orc = AsyncDqnScheduleOracle(p_np, t_np, cap, b, seed, n_actions=a, epsilon=bench.epsilon)
    rng_state = np.random.get_state()
    L = bench.learner
    # ---- GPU run: 14 pipelined steps; TD errors and the learner's state read back after each (synchronising changes no result)
    np.random.seed(5)
    gpu_delta, gpu_state = [], []
    for _ in range(steps):
        bench.step()
        L.synchronize()
        gpu_delta.append(L.delta.cpu().numpy().copy())
        gpu_state.append(L.export_state())
    n_tr = 4 * (steps + 1)                       # the actor ran one agent step ahead
    gpu_actions = d.ops._wrap_device_pointer(bench.ring.pointers()[1], n_tr, torch.int64).cpu().numpy().copy()
    gpu_frames = d.ops._wrap_device_pointer(bench.ring.pointers()[0], n_tr * 7056, torch.uint8).cpu().numpy().copy()
    # ---- oracle run of the same schedule (same global np.random stream for the minibatch draws)
    np.random.seed(5)
    near_ties, strict_steps, diag = 0, 0, []

    def check_actions(res, first, who):
        nonlocal near_ties
        for e, (act, gap, rnd) in enumerate(res):
            got = gpu_actions[first + e]
            if act != got:
                assert (not rnd) and gap < 1e-5, "%s env step %d: action %d vs %d, top-2 gap %g" % (who, e, act, got, gap)
                near_ties += 1

    check_actions(orc.actor_step(orc._snapshot(), override_actions=gpu_actions[0:4]), 0, "actor(0)")
    for k in range(steps):
        idx, batch = orc.sample()
        theta = orc._snapshot()                                  # theta_k: what actor(k+1) acts on
        check_actions(orc.actor_step(theta, override_actions=gpu_actions[4 * (k + 1):4 * (k + 2)]), 4 * (k + 1), "actor(%d)" % (k + 1))
        loss, delta, q, norm = orc.update(batch)
        ambiguous = orc.relu_margin < 5e-7
        f = 100.0 if ambiguous else 1.0
        strict_steps += not ambiguous
        scale = max(1.0, float(np.abs(q).max()))
        perr = max(float(np.abs(gpu_state[k]["params"][n].numpy() - orc.p[n].detach().numpy()).max()) for n in orc.names)
        diag.append((k, "%.1e" % orc.relu_margin, "%.1e" % float(np.abs(gpu_delta[k] - delta).max()), "%.1e" % perr))
        msg = "step %d (step, relu margin, max TD err, max param err): %s" % (k, diag)
        _record_parity("schedule_oracle[%d-%s-%d] step %d%s" % (variant, init, cap, k, " (ambiguous ReLU gate)" if ambiguous else ""),
                       td=_rel(gpu_delta[k], delta, scale), loss=abs(0.5 * float(np.mean(gpu_delta[k].astype(np.float64) ** 2)) - loss) / abs(loss),
                       params_abs=perr, relu_margin=orc.relu_margin)
        np.testing.assert_allclose(gpu_delta[k], delta, rtol=1e-5 * f, atol=1e-5 * scale * f, err_msg="TD errors, " + msg)
        np.testing.assert_allclose(0.5 * float(np.mean(gpu_delta[k].astype(np.float64) ** 2)), loss, rtol=1e-5 * f,
                                   err_msg="loss, " + msg)
        for n in orc.names:
            np.testing.assert_allclose(gpu_state[k]["params"][n].numpy(), orc.p[n].detach().numpy(), rtol=1e-5 * f,
                                       atol=2e-6 * f, err_msg=n + ", " + msg)
        if ambiguous or init == "bench":
            orc.load_state(gpu_state[k])
    assert near_ties <= 1, "more than one fp32 near-tie in %d greedy decisions is not plausible" % len(orc.q_gaps)
    assert strict_steps >= (steps // 2 if init == "normal" else 3), "too few unambiguous steps to mean anything: %s" % diag
    assert np.array_equal(gpu_frames, orc.rep.state[:n_tr].reshape(n_tr * 7056))
    np.random.set_state(rng_state)
    L.close()
    bench.ring.close()


@pytest.mark.parametrize("kind,per,n_step,device_env,done_period",
                         [("dqn", False, 1, False, 800), ("dqn", True, 3, False, 800), ("dqn", False, 1, True, 800),
                          ("dqn", False, 1, True, 7), ("dqn", False, 3, True, 11), ("dqn", True, 3, True, 7),
                          ("c51", False, 1, False, 800), ("c51", True, 1, False, 800), ("c51", True, 3, True, 9),
                          ("c51", False, 1, True, 7), ("qr", False, 1, False, 800), ("qr", False, 3, True, 11)])
def test_dqn_agent_fused_fast_path_matches_generic(dra, monkeypatch, kind, per, n_step, device_env, done_period):
    """dqn_pixel configuration (examples.py:55-97 shapes): DQNAgent attaches the fused learner
    (csrc/learner.hip) after the first feed.  20 agent steps give the same action stream and replay
    contents and, to fp32 reassociation, the same parameters as the generic autograd path
    (config.fused_learner = False), which the tests above pin to the reference's fixtures.
    device_env=True: the environment itself is device-resident (DeviceActorPipeline, in-order mode): forward,
    epsilon-greedy, environment step and replay feed never leave the GPU, and the run still equals the generic
    host-emulator path transition for transition -- including episode ends (done_period 7 / 11: frame stacks restart
    with the first frame repeated, the discarded post-terminal frame, rewards / masks of the LEAVING transition).
    kind = c51 / qr: CategoricalDQNAgent / QuantileRegressionDQNAgent with Adam (examples.py:127-158, 192-222: BASELINE
    configs[3]) through the same learner -- distributional head + fused loss kernel + Adam in the update, the
    distributional action values in the device actor; per=True: prioritized draws with the priorities written back to the
    tree on the device (device_env=True then runs the in-order device pipeline)."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for fused in (True, False):
        cfg = d.Config()
        cfg.merge(dict(game="synthetic-atari", n_step=n_step, replay_cls=d.PrioritizedReplay if per else d.UniformReplay,
                       async_replay=False, log_level=0, tag="fast%d" % fused, fused_learner=fused, device_env=device_env))
        cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
        cfg.replay_beta = d.LinearSchedule(0.4, 1.0, 1000)
        cfg.task_fn = lambda: d.Task(cfg.game, seed=7, synthetic_done_period=done_period)
        cfg.eval_env = cfg.task_fn()
        head = [("fc_head.weight", (cfg.action_dim, 512)), ("fc_head.bias", (cfg.action_dim,))]
        cls = d.DQNAgent
        if kind == "dqn":
            cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
            cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
        elif kind == "c51":
            cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=0.00025, eps=0.01 / 32)
            cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
            cfg.network_fn = lambda: d.CategoricalNet(cfg.action_dim, cfg.categorical_n_atoms, d.NatureConvBody())
            cls = d.CategoricalDQNAgent
            head = [("fc_categorical.weight", (cfg.action_dim * 51, 512)), ("fc_categorical.bias", (cfg.action_dim * 51,))]
        else:
            cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=0.00005, eps=0.01 / 32)
            cfg.num_quantiles = 200
            cfg.network_fn = lambda: d.QuantileNet(cfg.action_dim, cfg.num_quantiles, d.NatureConvBody())
            cls = d.QuantileRegressionDQNAgent
            head = [("fc_quantiles.weight", (cfg.action_dim * 200, 512)), ("fc_quantiles.bias", (cfg.action_dim * 200,))]
        cfg.random_action_prob = d.LinearSchedule(1.0, 0.05, 60)
        cfg.batch_size = 32
        cfg.discount = 0.99
        cfg.history_length = 4
        kw = dict(memory_size=500, batch_size=32, n_step=n_step, discount=0.99, history_length=4)
        cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
        cfg.state_normalizer = d.ImageNormalizer()
        cfg.reward_normalizer = d.SignNormalizer()
        cfg.target_network_update_freq = 3
        cfg.exploration_steps = 40
        cfg.sgd_update_frequency = 4
        cfg.gradient_clip = 5
        cfg.double_q = False
        cfg.async_actor = False
        cfg.max_steps = 1e5
        d.random_seed(3)
        random.seed(3)
        agent = cls(cfg)
        p_np = fake_envs.numpy_params(fake_envs.NATURE_SHAPES + head, 17)
        agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        for _ in range(20):
            agent.step()
        assert (agent._learner is not None) == fused
        assert (agent._pipe is not None) == (fused and device_env)
        if fused:
            agent._learner.synchronize()
        torch.cuda.synchronize()
        rp = agent.replay.replay
        n = rp.size()
        acts = d.ops._wrap_device_pointer(rp._ring.pointers()[1], n, torch.int64).cpu().numpy().copy()
        frames = d.ops._wrap_device_pointer(rp._ring.pointers()[0], n * 7056, torch.uint8).cpu().numpy().copy()
        params = {k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()}
        tparams = {k: v.detach().cpu().numpy().copy() for k, v in agent.target_network.state_dict().items()}
        rng_tail = np.concatenate([np.random.randint(0, 1 << 30, size=4), [random.getrandbits(30)]])
        if per:   # the priority tree after 10 updates: same leaves sampled, priorities to fp32 reassociation
            tree = rp.tree.as_tensor().cpu().numpy().copy()
            params["__tree__"] = tree
            tparams["__tree__"] = tree
        outs.append((acts, frames, params, tparams, rng_tail, agent.total_steps))
        agent.close()
    assert outs[0][5] == outs[1][5] == 80
    assert np.array_equal(outs[0][0], outs[1][0])          # action stream
    assert np.array_equal(outs[0][1], outs[1][1])          # frames fed
    assert np.array_equal(outs[0][4], outs[1][4])          # np.random consumption
    for k in outs[0][2]:
        np.testing.assert_allclose(outs[0][2][k], outs[1][2][k], rtol=2e-4, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(outs[0][3][k], outs[1][3][k], rtol=2e-4, atol=2e-6, err_msg="target " + k)


@pytest.mark.parametrize("kind", ["c51", "qr", "dqn_fc"])
def test_graphed_generic_update_is_bit_identical(dra, monkeypatch, kind):
    """The generic DQN-family path replays its update (and, for image observations, the actor forward) from a
    captured hipGraph after two eager updates: same kernels and arguments, so 16 agent steps with
    config.graph_update True / False end in bit-identical parameters, replay contents and RNG state
    (C51 and QR-DQN use Adam: its step-dependent scalars come from device memory in graph mode)."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for graph in (True, False):
        cfg = d.Config()
        pixel = kind != "dqn_fc"
        cfg.merge(dict(game="BreakoutNoFrameskip-v4" if pixel else "CartPole-v0", n_step=1, replay_cls=d.UniformReplay,
                       async_replay=False, log_level=0, tag="g%d" % graph, graph_update=graph, fused_learner=False))
        cfg.task_fn = lambda: d.Task(cfg.game, seed=5)
        cfg.eval_env = cfg.task_fn()
        if kind == "c51":
            cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=0.00025, eps=0.01 / 32)
            cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
            cfg.network_fn = lambda: d.CategoricalNet(cfg.action_dim, cfg.categorical_n_atoms, d.NatureConvBody())
            cls = d.CategoricalDQNAgent
        elif kind == "qr":
            cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=0.00005, eps=0.01 / 32)
            cfg.num_quantiles = 50
            cfg.network_fn = lambda: d.QuantileNet(cfg.action_dim, cfg.num_quantiles, d.NatureConvBody())
            cls = d.QuantileRegressionDQNAgent
        else:
            cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, 0.001)
            cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.FCBody(cfg.state_dim))
            cls = d.DQNAgent
        cfg.random_action_prob = d.LinearSchedule(1.0, 0.05, 60)
        cfg.batch_size = 16
        cfg.discount = 0.99
        cfg.history_length = 4 if pixel else 1
        kw = dict(memory_size=400, batch_size=16, n_step=1, discount=0.99, history_length=cfg.history_length)
        cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
        if pixel:
            cfg.state_normalizer = d.ImageNormalizer()
            cfg.reward_normalizer = d.SignNormalizer()
        cfg.target_network_update_freq = 3
        cfg.exploration_steps = 24
        cfg.sgd_update_frequency = 4
        cfg.gradient_clip = 5
        cfg.double_q = False
        cfg.async_actor = False
        cfg.max_steps = 1e5
        d.random_seed(4)
        random.seed(4)
        torch.manual_seed(4)
        agent = cls(cfg)
        for _ in range(16):
            agent.step()
        torch.cuda.synchronize()
        assert (agent._graphed.graph is not None) == graph
        if pixel:
            assert (agent.actor._graphed_q.graph is not None) == graph
        rp = agent.replay.replay
        n = rp.size()
        acts = d.ops._wrap_device_pointer(rp._ring.pointers()[1], n, torch.int64).cpu().numpy().copy()
        params = {k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()}
        rng_tail = np.random.randint(0, 1 << 30, size=4)
        outs.append((acts, params, rng_tail))
        agent.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][2], outs[1][2])
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


@pytest.mark.parametrize("kind", ["continuous", "continuous_remainder", "pixel"])
def test_graphed_ppo_optimisation_is_bit_identical(dra, monkeypatch, kind):
    """PPO's minibatch loop replayed from captured hipGraphs (config.graph_update, default on) against the eager
    loop: 3 rollouts (the first is eager in both runs) end in bit-identical parameters.  `continuous`: separate
    actor / critic Adam optimisers with the KL gate (examples.py:497-523 shapes); `continuous_remainder`: a
    minibatch size that leaves a remainder batch; `pixel`: shared representation, clipped gradient, lr schedule
    (examples.py:525-550 shapes, 2 workers)."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for graph in (True, False):
        c = d.Config()
        pixel = kind == "pixel"
        c.merge(dict(game="BreakoutNoFrameskip-v4" if pixel else "HalfCheetah-v2", log_level=0, tag="ppo%d" % graph,
                     graph_update=graph, skip=False, fused_ppo_mlp=False))      # (the generic minibatch loop is under test)
        c.num_workers = 2
        c.task_fn = lambda: d.Task(c.game, num_envs=c.num_workers, seed=3)
        c.eval_env = d.Task(c.game, seed=4)
        if pixel:
            c.optimizer_fn = lambda p: torch.optim.Adam(p, lr=2.5e-4)
            c.network_fn = lambda: d.CategoricalActorCriticNet(c.state_dim, c.action_dim, d.NatureConvBody())
            c.state_normalizer, c.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
            c.entropy_weight, c.gradient_clip, c.rollout_length, c.optimization_epochs = 0.01, 0.5, 8, 2
            c.mini_batch_size, c.ppo_ratio_clip, c.shared_repr, c.max_steps = 8, 0.1, True, 1000
        else:
            c.network_fn = lambda: d.GaussianActorCriticNet(c.state_dim, c.action_dim,
                                                            actor_body=d.FCBody(c.state_dim, gate=torch.tanh),
                                                            critic_body=d.FCBody(c.state_dim, gate=torch.tanh))
            c.actor_opt_fn = lambda p: torch.optim.Adam(p, 3e-4)
            c.critic_opt_fn = lambda p: torch.optim.Adam(p, 1e-3)
            c.gradient_clip, c.rollout_length, c.optimization_epochs = 0.5, 32, 2
            c.mini_batch_size = 24 if kind == "continuous_remainder" else 16
            c.ppo_ratio_clip, c.max_steps, c.target_kl = 0.2, 3e6, 0.01
            c.state_normalizer = d.MeanStdNormalizer()
        c.discount, c.use_gae, c.gae_tau = 0.99, True, 0.95
        c.log_interval = 10 ** 9
        d.random_seed(9)
        torch.manual_seed(9)
        torch.cuda.manual_seed_all(9)
        agent = d.PPOAgent(c)
        for _ in range(3):
            agent.step()
        torch.cuda.synchronize()
        assert (agent._graphed.graphs is not None) == graph
        outs.append({k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()})
        agent.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("kind", ["a2c", "ppo"])
def test_onpolicy_device_env_equals_host_emulators(dra, monkeypatch, kind):
    """A2C / PPO on pixels (BASELINE configs[4] shapes): with device-resident environments (device_env.DeviceAtariVec: the
    host lays a rollout out, observations are generated on the device, the whole rollout -- and for A2C the update too --
    replays from one captured graph) the agents end on the SAME parameters, bit for bit, as with the host emulators: same
    frames, same normaliser arithmetic, same kernels, same generator sequence for the action samples."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for device_env in (True, False):
        cfg = d.Config()
        # (reuse_rollout_activations off: bit-identity with the host-emulator agent needs the update to recompute conv1-3 as that one does)
        cfg.merge(dict(game="synthetic-atari", log_level=0, tag="dv%d" % device_env, device_env=device_env, reuse_rollout_activations=False))
        cfg.num_workers = 4
        cfg.task_fn = lambda: d.Task(cfg.game, num_envs=cfg.num_workers, seed=11, synthetic_done_period=13)
        cfg.eval_env = d.Task(cfg.game, seed=12)
        cfg.network_fn = lambda: d.CategoricalActorCriticNet(cfg.state_dim, cfg.action_dim, d.NatureConvBody())
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.discount, cfg.use_gae, cfg.entropy_weight = 0.99, True, 0.01
        if kind == "a2c":
            cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-4, alpha=0.99, eps=1e-5)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 1.0, 5, 5
            cls, n = d.A2CAgent, 6
        else:
            cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=2.5e-4)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 0.95, 16, 0.5
            cfg.optimization_epochs, cfg.mini_batch_size, cfg.ppo_ratio_clip, cfg.shared_repr = 2, 16, 0.1, True
            cfg.max_steps, cfg.log_interval, cfg.target_kl = 1e6, 10 ** 9, 0.01
            cls, n = d.PPOAgent, 5
        d.random_seed(21)
        torch.manual_seed(22)
        agent = cls(cfg)
        assert getattr(agent.task, "on_device", False) == device_env
        for _ in range(n):
            agent.step()
        torch.cuda.synchronize()
        if device_env:
            assert agent._dev_graph.graph is not None, "the rollout must be replayed from the captured graph"
        outs.append(({k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()}, agent.total_steps,
                     np.random.randint(0, 1 << 30, size=2)))
        agent.close()
    assert outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][2], outs[1][2])
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k


@pytest.mark.parametrize("kind", ["a2c", "ppo"])
def test_in_place_parameter_gradients_equal_autograd_accumulation(dra, monkeypatch, kind):
    """nets.direct_param_grads (round 4): inside an A2C update / a PPO minibatch the layer Functions write their weight and bias
    gradients straight into the optimizer's flat gradient buffer and return None for them, instead of twelve AccumulateGrad adds
    onto the zeroed buffer.  0 + g == g: the agents end on bit-identical parameters with the mechanism switched off."""
    d = dra
    import contextlib
    import deeprl_amd.agents as agents_mod
    import deeprl_amd.nets as nets_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for direct in (True, False):
        if not direct:
            monkeypatch.setattr(nets_mod, "direct_param_grads", lambda enable=True, **kw: contextlib.nullcontext())
        cfg = d.Config()
        # (defer_conv_folds off: the deferred form sums the norm's partials in another grouping -- compared on its own below)
        cfg.merge(dict(game="synthetic-atari", num_workers=4, log_level=0, tag="dg%d" % direct, device_env=True,
                       defer_conv_folds=False))
        cfg.task_fn = lambda: d.Task(cfg.game, num_envs=cfg.num_workers, seed=31, synthetic_done_period=9)
        cfg.eval_env = d.Task(cfg.game, seed=3)
        cfg.network_fn = lambda: d.CategoricalActorCriticNet(cfg.state_dim, cfg.action_dim, d.NatureConvBody())
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.discount, cfg.use_gae, cfg.entropy_weight = 0.99, True, 0.01
        if kind == "a2c":
            cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-4, alpha=0.99, eps=1e-5)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 1.0, 5, 5
            cls, n = d.A2CAgent, 5
        else:
            cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=2.5e-4)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 0.95, 16, 0.5
            cfg.optimization_epochs, cfg.mini_batch_size, cfg.ppo_ratio_clip, cfg.shared_repr = 2, 16, 0.1, True
            cfg.max_steps, cfg.log_interval, cfg.target_kl = 1e6, 10 ** 9, 0.01
            cls, n = d.PPOAgent, 4
        d.random_seed(21)
        torch.manual_seed(22)
        agent = cls(cfg)
        for _ in range(n):
            agent.step()
        torch.cuda.synchronize()
        outs.append({k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()})
        agent.close()
    for k in outs[0]:
        assert np.isfinite(outs[0][k]).all()
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("kind,workers", [("a2c", 16), ("ppo", 8), ("a2c", 5)])
def test_fused_rollout_launches_equal_the_module_path(dra, monkeypatch, kind, workers):
    """agents._PixelRollout (round 5): a rollout step as [conv1 | head of the previous step], conv2, conv3, fc4 -- the same device
    functions as the five separate launches of network.forward, the head one step behind in conv1's launch: the agents end on
    the SAME parameters bit for bit and draw the same actions."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for fused in (True, False):
        cfg = d.Config()
        # (reuse_rollout_activations off: with it A2C's update takes the conv outputs of the rollout's eight-wave kernels instead of
        # recomputing them with the four-wave shape -- another fp32 summation order; test_a2c_update_through_the_rollout_activations)
        cfg.merge(dict(game="synthetic-atari", log_level=0, tag="fr%d" % fused, device_env=True, fused_rollout=fused,
                       reuse_rollout_activations=False))
        cfg.num_workers = workers
        cfg.task_fn = lambda: d.Task(cfg.game, num_envs=cfg.num_workers, seed=11, synthetic_done_period=13)
        cfg.eval_env = d.Task(cfg.game, seed=12)
        cfg.network_fn = lambda: d.CategoricalActorCriticNet(cfg.state_dim, cfg.action_dim, d.NatureConvBody())
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.discount, cfg.use_gae, cfg.entropy_weight = 0.99, True, 0.01
        if kind == "a2c":
            cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-3, alpha=0.99, eps=1e-5)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 1.0, 5, 5
            cls, n = d.A2CAgent, 12
        else:
            cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=1e-3)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 0.95, 32, 0.5
            cfg.optimization_epochs, cfg.mini_batch_size, cfg.ppo_ratio_clip, cfg.shared_repr = 2, 64, 0.1, True
            cfg.max_steps, cfg.log_interval, cfg.target_kl = 1e6, 10 ** 9, 0.01
            cls, n = d.PPOAgent, 5
        d.random_seed(21)
        torch.manual_seed(22)
        agent = cls(cfg)
        assert agent._pixel_rollout.eligible() == fused
        acts = []
        for _ in range(n):
            agent.step()
            acts.append(agent.network.rollout_slots.action.clone())
        torch.cuda.synchronize()
        outs.append(({k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()},
                     torch.stack(acts).cpu().numpy(), agent.network.rollout_slots.log_pi_a.cpu().numpy().copy()))
        agent.close()
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert len(np.unique(outs[0][1])) > 1
    for k in outs[0][0]:
        assert np.array_equal(outs[0][0][k], outs[1][0][k]), k


@pytest.mark.parametrize("workers", [16, 5])
def test_a2c_update_through_the_rollout_activations(dra, monkeypatch, workers):
    """config.reuse_rollout_activations (round 6, default): A2C's update backpropagates through the conv outputs the ROLLOUT computed
    (A2C_agent.py:29-64 keeps the rollout's own forward graph) instead of recomputing conv1-3 over the 80 stored observations.
    Same inputs and parameters, but the rollout runs the eight-wave kernel shape and the recomputed forward the four-wave one
    (another fp32 summation tree): from the same start the first update's flat GRADIENT agrees to 1e-5 of its largest magnitude,
    the parameters after it to 1e-4 of each tensor's largest magnitude (RMSprop's first step divides by ~0.1 |g| + eps: an element
    with |g| near eps turns a 1e-9 gradient difference into a 1e-7 step difference -- measured 1.4e-5 of scale on fc_action.weight),
    and the rollout's actions / log-probabilities / stored activations are bit-identical (the rollout's kernels are the same)."""
    d = dra
    import deeprl_amd.agents as agents_mod
    from deeprl_amd import ops
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for reuse in (True, False):
        cfg = d.Config()
        cfg.merge(dict(game="synthetic-atari", log_level=0, tag="ra%d" % reuse, device_env=True, reuse_rollout_activations=reuse))
        cfg.num_workers = workers
        cfg.task_fn = lambda: d.Task(cfg.game, num_envs=cfg.num_workers, seed=11, synthetic_done_period=13)
        cfg.eval_env = d.Task(cfg.game, seed=12)
        cfg.network_fn = lambda: d.CategoricalActorCriticNet(cfg.state_dim, cfg.action_dim, d.NatureConvBody())
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.discount, cfg.use_gae, cfg.entropy_weight = 0.99, True, 0.01
        cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-3, alpha=0.99, eps=1e-5)
        cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 1.0, 5, 5
        d.random_seed(21)
        torch.manual_seed(22)
        agent = d.A2CAgent(cfg)
        assert agent._pixel_rollout.eligible()
        start = {k: v.detach().clone() for k, v in agent.network.state_dict().items()}
        grads = []
        agent.grad_hook = lambda g: grads.append(g.detach().clone())
        agent.step()
        torch.cuda.synchronize()
        agent.grad_hook = None
        bufs = agent._pixel_rollout.bufs
        if reuse:       # the stored activations against a recomputed batched forward with the STARTING parameters
            body = agent.network.phi_body
            for conv in (body.conv1, body.conv2, body.conv3):
                assert "_y_pre" not in conv.__dict__          # consumed by the update's forward
            assert "_phi_pre" not in agent.network.__dict__    # (fc4's output too: the rollout's head launches leave it)
        outs.append(({k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()},
                     agent.network.rollout_slots.action.cpu().numpy().copy(), agent.network.rollout_slots.log_pi_a.cpu().numpy().copy(),
                     {k: bufs[k].detach().cpu().numpy().copy() for k in ("y1", "y2", "y3", "phi")},
                     {k: v.cpu().numpy() for k, v in start.items()}, grads[0].cpu().numpy().astype(np.float64)))
        # a few more steps through the captured graph: finite, and the hand-over is consumed every time
        for _ in range(4):
            agent.step()
        torch.cuda.synchronize()
        assert all(np.isfinite(v.detach().cpu().numpy()).all() for v in agent.network.state_dict().values())
        agent.close()
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    for k in ("y1", "y2", "y3", "phi"):        # the rollout's kernels are the same in both runs
        assert np.array_equal(outs[0][3][k][:5], outs[1][3][k][:5]), k
    assert np.abs(outs[0][3]["phi"][:5]).max() > 0
    moved = 0.0
    for k in outs[0][0]:
        a, b, s0 = outs[0][0][k].astype(np.float64), outs[1][0][k].astype(np.float64), outs[0][4][k].astype(np.float64)
        scale = max(np.abs(b).max(), 1e-3)
        assert np.abs(a - b).max() <= 1e-4 * scale, (k, np.abs(a - b).max() / scale)
        moved = max(moved, np.abs(b - s0).max())
    assert moved > 1e-4          # the update did something
    ga, gb = outs[0][5], outs[1][5]
    assert np.abs(gb).max() > 0 and np.abs(ga - gb).max() <= 1e-5 * np.abs(gb).max(), np.abs(ga - gb).max() / np.abs(gb).max()


@pytest.mark.parametrize("kind", ["a2c", "ppo"])
def test_fc4_and_policy_head_as_one_autograd_node(dra, monkeypatch, kind):
    """nets._Fc4PolicyHeadFn (round 5): in the update's forward fc4's K-slice partial sums are folded inside the policy-head launch
    (no finish launch) and fc4's two gradients share a launch -- the same sums in the same order as Linear + _PolicyHeadFn: the
    agents end on bit-identical parameters with network.fuse_fc4_head switched off."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for fuse in (True, False):
        cfg = d.Config()
        # (reuse_rollout_activations off: with it the fused node takes fc4's output from the rollout's 28-slice fold, the unfused
        # path recomputes it)
        cfg.merge(dict(game="synthetic-atari", log_level=0, tag="fh%d" % fuse, device_env=True, fuse_fc4_head=fuse,
                       reuse_rollout_activations=False))
        cfg.num_workers = 16 if kind == "a2c" else 8
        cfg.task_fn = lambda: d.Task(cfg.game, num_envs=cfg.num_workers, seed=11, synthetic_done_period=13)
        cfg.eval_env = d.Task(cfg.game, seed=12)
        cfg.network_fn = lambda: d.CategoricalActorCriticNet(cfg.state_dim, cfg.action_dim, d.NatureConvBody())
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.discount, cfg.use_gae, cfg.entropy_weight = 0.99, True, 0.01
        if kind == "a2c":
            cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-3, alpha=0.99, eps=1e-5)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 1.0, 5, 5          # update batch 80
            cls, n = d.A2CAgent, 6
        else:
            cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=1e-3)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 0.95, 32, 0.5
            cfg.optimization_epochs, cfg.mini_batch_size, cfg.ppo_ratio_clip, cfg.shared_repr = 2, 64, 0.1, True
            cfg.max_steps, cfg.log_interval, cfg.target_kl = 1e6, 10 ** 9, 0.01
            cls, n = d.PPOAgent, 4
        d.random_seed(21)
        torch.manual_seed(22)
        agent = cls(cfg)
        assert agent.network.fuse_fc4_head == fuse
        for _ in range(n):
            agent.step()
        torch.cuda.synchronize()
        outs.append({k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()})
        agent.close()
    for k in outs[0]:
        assert np.isfinite(outs[0][k]).all()
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_rollout_fc4_through_k_slices_samples_the_same_policy(dra, monkeypatch):
    """config.rollout_fc4_slices (round 5, default on): a rollout step's fc4 runs as the one-pass 28-slice kernel with its finish
    inside the head launch instead of the eight-wave GEMV -- another fp32 summation order for the 3136-term dot products, so the
    rollout's log-probabilities and values agree at 1e-5 (not bit for bit) and the same uniforms pick the same actions except
    where one lands within that distance of a CDF boundary (none in this run's first rollouts)."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for slices in (True, False):
        cfg = d.Config()
        cfg.merge(dict(game="synthetic-atari", log_level=0, tag="rs%d" % slices, device_env=True, rollout_fc4_slices=slices))
        cfg.num_workers = 16
        cfg.task_fn = lambda: d.Task(cfg.game, num_envs=cfg.num_workers, seed=11, synthetic_done_period=13)
        cfg.eval_env = d.Task(cfg.game, seed=12)
        cfg.network_fn = lambda: d.CategoricalActorCriticNet(cfg.state_dim, cfg.action_dim, d.NatureConvBody())
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.discount, cfg.use_gae, cfg.entropy_weight = 0.99, True, 0.01
        cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-4, alpha=0.99, eps=1e-5)
        cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 1.0, 5, 5
        d.random_seed(21)
        torch.manual_seed(22)
        agent = d.A2CAgent(cfg)
        assert agent._pixel_rollout.eligible() == slices
        agent.step()
        torch.cuda.synchronize()
        sl = agent.network.rollout_slots
        outs.append((sl.action.cpu().numpy().copy(), sl.log_pi_a.cpu().numpy().copy(), sl.v.cpu().numpy().copy(),
                     {k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()}))
        agent.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-5)
    # How likely is a flipped action?  The inverse-CDF draw picks another action only when its uniform lies between the two
    # forms' CDF edges.  A probability moves by |dp| <= p * |d log p| <= the measured log-probability difference, an edge of the
    # A-action CDF by at most (A - 1) of those, and there are A - 1 edges: P(flip per draw) <= 2 (A - 1)^2 max|d log pi|.
    # state it and hold it to a bar -- below 1e-4 per draw (measured ~1e-6: a flip every ~10^4 rollouts of 80 draws).
    n_act = 4
    dlog = float(np.abs(outs[0][1] - outs[1][1]).max())
    flip_bound = 2.0 * (n_act - 1) ** 2 * dlog
    assert flip_bound < 1e-4, (dlog, flip_bound)
    for k in outs[0][3]:
        scale = max(1e-3, float(np.abs(outs[1][3][k]).max()))
        assert float(np.abs(outs[0][3][k] - outs[1][3][k]).max()) <= 1e-4 * scale, k


@pytest.mark.parametrize("kind", ["a2c", "ppo"])
def test_deferred_conv_folds_give_the_same_update(dra, monkeypatch, kind):
    """config.defer_conv_folds (round 5, default on): the conv layers leave their weight-gradient slabs unfolded and the
    optimizer's norm launch folds all three (dra_grad_sqnorm_segs over [conv1 | conv2 | conv3 | rest]) -- the folded gradients are
    the same bits, the norm's partial sums are grouped differently (fp64), so the clip coefficient may differ in its last bit:
    parameters after several agent steps agree to 1e-6 of their scale, and the captured graph holds one fold launch."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    outs = []
    for defer in (True, False):
        cfg = d.Config()
        cfg.merge(dict(game="synthetic-atari", num_workers=4, log_level=0, tag="df%d" % defer, device_env=True,
                       defer_conv_folds=defer))
        cfg.task_fn = lambda: d.Task(cfg.game, num_envs=cfg.num_workers, seed=31, synthetic_done_period=9)
        cfg.eval_env = d.Task(cfg.game, seed=3)
        cfg.network_fn = lambda: d.CategoricalActorCriticNet(cfg.state_dim, cfg.action_dim, d.NatureConvBody())
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.discount, cfg.use_gae, cfg.entropy_weight = 0.99, True, 0.01
        if kind == "a2c":
            cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=1e-4, alpha=0.99, eps=1e-5)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 1.0, 5, 5
            cls, n = d.A2CAgent, 5
        else:
            cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=2.5e-4)
            cfg.gae_tau, cfg.rollout_length, cfg.gradient_clip = 0.95, 16, 0.5
            cfg.optimization_epochs, cfg.mini_batch_size, cfg.ppo_ratio_clip, cfg.shared_repr = 2, 16, 0.1, True
            cfg.max_steps, cfg.log_interval, cfg.target_kl = 1e6, 10 ** 9, 0.01
            cls, n = d.PPOAgent, 4
        d.random_seed(21)
        torch.manual_seed(22)
        agent = cls(cfg)
        for _ in range(n):
            agent.step()
        torch.cuda.synchronize()
        assert not agent._fused._pending_folds
        outs.append({k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()})
        agent.close()
    for k in outs[0]:
        scale = max(1e-3, float(np.abs(outs[1][k]).max()))
        assert np.isfinite(outs[0][k]).all()
        assert float(np.abs(outs[0][k] - outs[1][k]).max()) <= 2e-6 * scale + 1e-7, k


@pytest.mark.parametrize("kind,per,chain", [("dqn", True, 2), ("c51", True, 2), ("dqn", True, 0), ("dqn", False, 2), ("c51", False, 2)])
def test_per_async_pipeline_equals_in_order_with_random_actions(dra, monkeypatch, kind, per, chain):
    """PrioritizedReplay inside the two-stream pipeline (config.async_actor=True: actor transitions of step t+1 on their own
    stream under update t; the prioritized draw of step t after the device-side write-back of update t-1; ring-direct update
    for DQN, gathered minibatch for C51).  With epsilon == 1 the actions do not depend on the (one update staler)
    parameters the async actor sees, so the run must equal the in-order pipeline EXACTLY: same transitions, same draws,
    same priority tree, bit-identical parameters -- on a 300-slot ring, where most minibatches touch the slots the
    concurrently running actor overwrites (the host-decided cross-stream waits are what keeps the schedule).
    per=False: the same equality for uniform replay.  The run starts with an exploration phase of actor-only calls, which the
    host issues far ahead of the device: the first updates must wait for every unfinished actor launch whose slots they
    sample, not just the latest one (csrc/learner.hip arec_needed)."""
    d = dra
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    # chain: where the prioritized draw runs in async mode -- 2 = entirely inside the update (device-side filter / padding,
    # replay.DeviceDraw; the default), 0 = the tree-stream form other kernel variants / minibatches above 1024 fall back to
    outs = []
    for async_actor in (True, False):
        cfg = d.Config()
        cfg.merge(dict(game="synthetic-atari", n_step=1, replay_cls=d.PrioritizedReplay if per else d.UniformReplay,
                       async_replay=False, log_level=0, tag="pa%d" % async_actor, device_env=True))
        cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
        cfg.replay_beta = d.LinearSchedule(0.4, 1.0, 1000)
        cfg.task_fn = lambda: d.Task(cfg.game, seed=9, synthetic_done_period=13)
        cfg.eval_env = cfg.task_fn()
        if kind == "dqn":
            cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
            cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
            cls, head = d.DQNAgent, [("fc_head.weight", (4, 512)), ("fc_head.bias", (4,))]
        else:
            cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=0.00025, eps=0.01 / 32)
            cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
            cfg.network_fn = lambda: d.CategoricalNet(cfg.action_dim, cfg.categorical_n_atoms, d.NatureConvBody())
            cls, head = d.CategoricalDQNAgent, [("fc_categorical.weight", (4 * 51, 512)), ("fc_categorical.bias", (4 * 51,))]
        cfg.random_action_prob = d.LinearSchedule(1.0, 1.0, 10)          # epsilon == 1 throughout
        cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
        kw = dict(memory_size=300, batch_size=32, n_step=1, discount=0.99, history_length=4)
        cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
        cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
        cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 5, 40, 4
        cfg.gradient_clip, cfg.double_q, cfg.async_actor, cfg.max_steps = 5, False, async_actor, 1e5
        d.random_seed(3)
        random.seed(3)
        agent = cls(cfg)
        assert agent._pipe is not None and agent._pipe.async_actor == async_actor and agent._pipe.per == per
        if per and async_actor:
            assert agent._pipe.chain == 2
            agent._pipe.chain = chain           # (0: force the fallback form before the first step)
        agent._pipe.rs = np.random.RandomState(77)          # the actor's randint / rand stream, identical in both modes
        np.random.seed(5)                                   # uniform index draws (the async constructor drew its actor seed)
        p_np = fake_envs.numpy_params(fake_envs.NATURE_SHAPES + head, 17)
        agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        agent._learner.invalidate_actor_copy()
        n_steps = 100
        for _ in range(n_steps):
            agent.step()
        agent.sync_host()                                   # (PER drawn on the device: python `random` back on the host)
        agent._learner.synchronize()
        torch.cuda.synchronize()
        rp = agent.replay.replay
        pos_size = (rp.pos, rp.size())
        if per and not async_actor and outs and outs[0]["chain"] == 2:
            # the async run drew INSIDE its update chain, filter and padding included (dra_sumtree_per_chain2): its last update
            # already performed the next agent step's four adds and the whole next draw (valid_index as of those four feeds).
            # Same point for the in-order run:
            rp.advance(4)
            rp.draw()
            torch.cuda.synchronize()
        # the async actor is one agent step ahead: compare the transitions both runs have REPORTED (ring slots of the first
        # n_steps * 4 transitions; with a 300-slot ring that is the whole ring minus the 4 newest slots of the async run)
        total = agent.total_steps
        frames, actions, rewards, masks = rp._ring.pointers()
        w = d.ops._wrap_device_pointer
        outs.append(dict(total=total, pos=pos_size[0], size=pos_size[1], chain=int(getattr(agent._pipe, "chain", 0)),
                         act=w(actions, 300, torch.int64).cpu().numpy().copy(), rew=w(rewards, 300, torch.float64).cpu().numpy().copy(),
                         tree=rp.tree.as_tensor().cpu().numpy().copy() if per else np.zeros(1),
                         maxp=float(rp.max_priority) if per else 0.0, np_rng=np.random.randint(0, 1 << 30, size=2),
                         params={k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()},
                         py_rng=[random.getrandbits(30) for _ in range(2)]))
        agent.close()
    a, b = outs
    assert a["total"] == b["total"] and a["pos"] == b["pos"] and a["size"] == b["size"]
    assert a["py_rng"] == b["py_rng"]                       # same number of prioritized draws / paddings
    assert np.array_equal(a["np_rng"], b["np_rng"])         # ... and of uniform index draws
    newest = [(a["pos"] + k) % 300 for k in range(4)]        # slots the async run's extra actor step has already rewritten
    keep = np.ones(300, dtype=bool)
    keep[newest] = False
    assert np.array_equal(a["act"][keep], b["act"][keep]) and np.array_equal(a["rew"][keep], b["rew"][keep])
    assert np.array_equal(a["tree"], b["tree"]) and a["maxp"] == b["maxp"]
    for k in a["params"]:
        assert np.array_equal(a["params"][k], b["params"][k]), k


@pytest.mark.parametrize("kind,cap,mode", [("dqn", 160, "async"), ("dqn", 4000, "async"), ("c51", 160, "async"),
                                           ("c51", 4000, "async"), ("dqn", 160, "inorder"), ("c51", 4000, "inorder")])
def test_per_pipeline_matches_schedule_oracle(dra, monkeypatch, kind, cap, mode):
    """PrioritizedReplay INSIDE the async two-stream pipeline against an oracle of its schedule (verdict r3 #2/#3: until
    round 4 this combination was pinned to the in-order path with epsilon = 1 only -- stale actor parameters x q-dependent
    actions x the device-side draw one launch late were pinned to nothing).  `agent.step()` of DQNAgent / CategoricalDQNAgent
    (device environment, config.async_actor, the whole `sample()` + `update_priorities()` on the device:
    replay.DeviceDraw + csrc/per_chain2.h) for 60 agent steps with epsilon decaying from 1 to 0.1, against
    oracle/async_schedule_oracle.py::AsyncPerAgentScheduleOracle -- itself pinned, driven in order, to a run of the
    reference's own CategoricalDQNAgent + PrioritizedReplay (tests/test_oracle_vs_golden.py).  Per update:
      * the minibatch's tree indices and sampling probabilities: exact (same tree, same python `random` words);
      * the pre-weight vector compute_loss returns (TD errors / KL) at 1e-5, the device's priorities against the oracle's
        own at 1e-5; the oracle then writes the DEVICE's priorities into its tree, so that
      * the priority tree after the update's commits and the next agent step's adds is compared BIT FOR BIT, every step;
      * parameters at rtol 1e-5 / atol 2e-6 (Adam: 5e-6); every stored action (a mismatch only at an fp32 near-tie).
    At the end: the whole ring (frames, actions, rewards, masks), max_priority and python's generator position.  160 slots:
    the ring wraps 1.5 times and most minibatches touch the slots the concurrently running actor launch overwrites.
    mode "inorder": config.async_actor = False -- the actor acts on the CURRENT parameters, the draw is the host-side one
    (tree descent on the tree stream, validity / padding on the host: PipelinedDqn.step's in-order branch), importance weights
    inside the in-order update, device-side write-back; same oracle driven in order, same per-update checks.  This is the
    GPU-side pin of DQN + PrioritizedReplay in order: an unsynchronised run against the reference's own run is not strict
    for it (tests/test_gpu_pixel_agents.py, GPU_CASES)."""
    import copy
    d = dra
    import deeprl_amd.agents as agents_mod
    from oracle.async_schedule_oracle import AsyncPerAgentScheduleOracle
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    steps, explore, freq, tfreq, done_period, b, A = 60, 40, 4, 5, 13, 32, 4
    cfg = d.Config()
    cfg.merge(dict(game="synthetic-atari", n_step=1, replay_cls=d.PrioritizedReplay, async_replay=False, log_level=0,
                   tag="aper", device_env=True))
    cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
    cfg.replay_beta = d.LinearSchedule(0.4, 1.0, 1000)
    cfg.task_fn = lambda: d.Task(cfg.game, seed=9, synthetic_done_period=done_period)
    cfg.eval_env = cfg.task_fn()
    if kind == "dqn":
        cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
        cfg.network_fn = lambda: d.VanillaNet(cfg.action_dim, d.NatureConvBody(in_channels=4))
        cls, head, okw = d.DQNAgent, [("fc_head.weight", (4, 512)), ("fc_head.bias", (4,))], dict(head="vanilla")
    else:
        cfg.optimizer_fn = lambda params: torch.optim.Adam(params, lr=0.00025, eps=0.01 / 32)
        cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
        cfg.network_fn = lambda: d.CategoricalNet(cfg.action_dim, cfg.categorical_n_atoms, d.NatureConvBody())
        cls, head = d.CategoricalDQNAgent, [("fc_categorical.weight", (4 * 51, 512)), ("fc_categorical.bias", (4 * 51,))]
        okw = dict(head="c51", lr=0.00025, eps=0.01 / 32)
    cfg.random_action_prob = d.LinearSchedule(1.0, 0.1, 40)          # q-dependent actions from the first update on
    cfg.batch_size, cfg.discount, cfg.history_length = b, 0.99, 4
    kw = dict(memory_size=cap, batch_size=b, n_step=1, discount=0.99, history_length=4)
    cfg.replay_fn = lambda: d.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
    cfg.state_normalizer, cfg.reward_normalizer = d.ImageNormalizer(), d.SignNormalizer()
    cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = tfreq, explore, freq
    inorder = mode == "inorder"
    cfg.gradient_clip, cfg.double_q, cfg.async_actor, cfg.max_steps = 5, False, not inorder, 1e5
    py_state = random.getstate()
    d.random_seed(3)
    random.seed(3)
    agent = cls(cfg)
    assert agent._pipe is not None and agent._pipe.async_actor != inorder and agent._pipe.per
    assert inorder or agent._pipe.chain == 2
    agent._pipe.rs = np.random.RandomState(77)
    p_np = fake_envs.numpy_params(fake_envs.NATURE_SHAPES + head, 17)
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
    agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
    agent._learner.invalidate_actor_copy()
    L, rp = agent._learner, agent.replay.replay
    w = d.ops._wrap_device_pointer
    rec = []
    host_draws = []
    if inorder:
        draw_end = rp.draw_end

        def recording_draw_end(pending):
            out = draw_end(pending)
            host_draws.append((np.asarray(out[0]).copy(), np.asarray(out[1], dtype=np.float64).copy()))
            return out

        rp.draw_end = recording_draw_end

    def slots_actions(first_transition):
        _, actions, _, _ = rp._ring.pointers()
        a = w(actions, cap, torch.int64).cpu().numpy()
        return np.asarray([a[(first_transition + e) % cap] for e in range(freq)])

    gpu_actions = {}
    for t in range(steps):
        agent.step()
        L.synchronize()
        torch.cuda.synchronize()
        if inorder:
            gpu_actions[t] = slots_actions(freq * t)
        else:
            if t == 0:
                gpu_actions[0] = slots_actions(0)
            gpu_actions[t + 1] = slots_actions(freq * (t + 1))      # the actor launch this call issued (one step ahead)
        r = dict(updated=agent.total_steps > explore)
        if r["updated"]:
            r.update(vec=L.delta.cpu().numpy().copy(), prio=L.prio.cpu().numpy().copy(), state=L.export_state(),
                     tree=rp.tree.as_tensor().cpu().numpy().copy())
            if inorder:
                r.update(tree_idx=host_draws[-1][0], prob=host_draws[-1][1])
            else:
                dd = agent._pipe._dd
                r.update(tree_idx=np.asarray(dd.next_tree_idx).copy(), prob=np.asarray(dd.next_p, dtype=np.float64) / dd.next_total)
        rec.append(r)
    assert len(host_draws) == (steps - explore // freq if inorder else 0)
    agent.sync_host()
    L.synchronize()
    torch.cuda.synchronize()
    frames, actions, rewards, masks = rp._ring.pointers()
    n = rp.size()
    end = dict(pos=rp.pos, size=n, act=w(actions, cap, torch.int64).cpu().numpy().copy(),
               rew=w(rewards, cap, torch.float64).cpu().numpy().copy(), msk=w(masks, cap, torch.int32).cpu().numpy().copy(),
               frames=w(frames, cap * 7056, torch.uint8).cpu().numpy().reshape(cap, 7056).copy(),
               tree=rp.tree.as_tensor().cpu().numpy().copy(), maxp=float(rp.max_priority),
               py_rng=[random.getrandbits(30) for _ in range(2)], total=agent.total_steps)
    agent.close()
    # ---- the oracle of the schedule
    random.seed(3)
    sched = d.LinearSchedule(1.0, 0.1, 40)
    n_act = [0]

    def epsilon():                       # DQN_agent.py:34-39
        eps = 1 if n_act[0] < explore else sched()
        n_act[0] += 1
        return eps

    orc = AsyncPerAgentScheduleOracle(p_np, cap, b, env_seed=9, done_period=done_period, actor_rs=np.random.RandomState(77),
                                      epsilon_fn=epsilon, beta_fn=d.LinearSchedule(0.4, 1.0, 1000), exploration_steps=explore,
                                      target_freq=tfreq, sgd_update_frequency=freq, n_actions=A, clip=5.0, **okw)
    near_ties, strict, n_upd = 0, 0, 0

    def check_actions(res, k):
        nonlocal near_ties
        for e, (act, gap, rnd) in enumerate(res):
            got = gpu_actions[k][e]
            if act != got:
                assert (not rnd) and gap < 1e-5, "actor(%d) env step %d: action %d vs %d, top-2 gap %g" % (k, e, act, got, gap)
                near_ties += 1

    if not inorder:
        check_actions(orc.actor_step(orc._snapshot(), override_actions=gpu_actions[0]), 0)
    atol_p = 2e-6 if kind == "dqn" else 5e-6
    for t in range(steps):
        r = rec[t]
        if inorder:                                                   # DQN_agent.py:84-127 as written: act, feed, sample, learn
            check_actions(orc.actor_step(orc._snapshot(), override_actions=gpu_actions[t]), t)
        upd = orc.report()
        assert upd == r["updated"]
        if upd:
            tree_idx, prob, data_idx, batch = orc.draw()
            assert np.array_equal(tree_idx, r["tree_idx"]), "update %d: minibatch leaves" % t
            np.testing.assert_allclose(prob, r["prob"], rtol=1e-12, atol=0, err_msg="update %d: sampling probabilities" % t)
        if not inorder:
            theta = orc._snapshot()                               # what actor(t+1) acts on: one update staler than in order
            check_actions(orc.actor_step(theta, override_actions=gpu_actions[t + 1]), t + 1)
        if upd:
            loss, vec, prio, wts = orc.learn(tree_idx, prob, batch, override_priorities=r["prio"])
            ambiguous = orc.relu_margin < 5e-7
            f = 100.0 if ambiguous else 1.0
            strict += not ambiguous
            n_upd += 1
            gvec = r["vec"]                                     # TD errors (DQN) / KL per sample (C51): what compute_loss returns
            scale = max(1e-3, float(np.abs(vec).max())) if kind == "c51" else max(1.0, float(np.abs(vec).max()))
            perr = max(float(np.abs(r["state"]["params"][nm].numpy() - orc.p[nm].detach().numpy()).max()) for nm in orc.names)
            _record_parity("per_schedule_oracle[%s-%d-%s] step %d%s" % (kind, cap, mode, t, " (ambiguous ReLU gate)" if ambiguous else ""),
                           loss_vec=_rel(gvec, vec, scale), prio=float(np.abs(r["prio"] - prio).max() / np.abs(prio).max()),
                           params_abs=perr, relu_margin=orc.relu_margin)
            msg = "step %d: relu margin %.1e, max param err %.1e" % (t, orc.relu_margin, perr)
            np.testing.assert_allclose(gvec, vec, rtol=1e-5 * f, atol=1e-5 * scale * f, err_msg="loss vector, " + msg)
            # priorities = (|vec| + eps)^0.5: the loss vector's tolerance propagated through the square root, d prio = d vec / (2 prio)
            ptol = 1e-5 * scale * f / (2.0 * np.maximum(prio, 1e-3)) + 1e-6
            assert (np.abs(r["prio"] - prio) <= ptol + 1e-5 * f * np.abs(prio)).all(), "priorities, %s: max err %.3g" % (
                msg, float(np.abs(r["prio"] - prio).max()))
            for nm in orc.names:
                np.testing.assert_allclose(r["state"]["params"][nm].numpy(), orc.p[nm].detach().numpy(), rtol=1e-5 * f,
                                           atol=atol_p * f, err_msg=nm + ", " + msg)
            if ambiguous:
                orc.load_state(r["state"])
            # the tree after this update's commits AND the adds of the transitions actor(t+1) just produced (the device
            # performs them inside the same update: per_chain2.h), bit for bit
            ahead = copy.deepcopy(orc.rep)
            for fr, ac, rw, mk in (orc.ahead or []):              # (in order: nothing is ahead)
                ahead.feed_one(fr, ac, rw, mk)
            assert np.array_equal(ahead.tree.tree, r["tree"]), "step %d: priority tree differs in %d nodes" % (
                t, int((ahead.tree.tree != r["tree"]).sum()))
        orc.maybe_sync_target()
    assert near_ties <= 1 and n_upd == steps - explore // freq
    assert strict >= n_upd // 2, "too few unambiguous steps to mean anything"
    # ---- end state (async: the device ran one more actor step, the next step's adds and the next draw)
    if not inorder:
        orc.report()
        orc.draw()
    o = orc.rep
    assert end["total"] == freq * steps == orc.total_steps - (0 if inorder else freq)
    assert end["pos"] == (freq * steps) % cap and end["size"] == min(cap, freq * steps)
    assert np.array_equal(end["act"][:o.size()], o.action[:o.size()].reshape(-1))
    assert np.array_equal(end["rew"][:o.size()], o.reward[:o.size()]) and np.array_equal(end["msk"][:o.size()], o.mask[:o.size()])
    assert np.array_equal(end["frames"][:o.size()], o.state[:o.size()].reshape(o.size(), 7056))
    assert np.array_equal(end["tree"], o.tree.tree) and end["maxp"] == float(o.max_priority)
    assert end["py_rng"] == [random.getrandbits(30) for _ in range(2)]
    random.setstate(py_state)


@pytest.mark.parametrize("head,cap", [("c51", 4000), ("c51", 160), ("qr", 4000)])
def test_async_pipeline_dist_heads_match_schedule_oracle(dra, head, cap):
    """Config 4's heads in ASYNC mode against the oracle of the schedule (verdict r2 #4: until round 3 they were pinned
    through async == in-order with random actions only): DQNLearnerBench(head="c51" | "qr") -- CategoricalNet / QuantileNet
    over NatureConvBody, the fused learner's distributional head + fused loss kernel + Adam, the distributional action values
    in the device actor -- for 10 pipelined agent steps vs AsyncDqnScheduleOracle(head=...): every stored action (a mismatch
    only where the oracle's own top-2 action values are within 1e-5), ring frames bit for bit, per step the loss vector
    (KL per sample / quantile loss per target quantile) at rtol 1e-5 with an absolute floor of 1e-5 x its largest entry, the
    loss at 1e-5 and the parameters at rtol 1e-5 / atol 2e-6 (Adam: atol 5e-6 -- a gradient element that is summation noise
    becomes a step of a fraction of lr through 1 / (sqrt(v) + eps)).  Steps with an ambiguous ReLU gate are judged at 100x
    and the oracle re-adopts the implementation's state, as in the VanillaNet test."""
    d = dra
    from deeprl_amd.learner import DQNLearnerBench
    from oracle.async_schedule_oracle import AsyncDqnScheduleOracle
    b, a, seed, steps = 32, 4, 3, 10
    n = 51 if head == "c51" else 200
    hname = "fc_categorical" if head == "c51" else "fc_quantiles"
    d.random_seed(11)
    torch.manual_seed(5)
    bench = DQNLearnerBench(ring_capacity=cap, batch=b, seed=seed, actor=True, async_actor=True, head=head)
    p0 = fake_envs.numpy_params(fake_envs.NATURE_SHAPES + [(hname + ".weight", (a * n, 512)), (hname + ".bias", (a * n,))], 21)
    bench.network.load_state_dict({k: torch.from_numpy(v) for k, v in p0.items()})
    bench.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p0.items()})
    p_np = {k: v.detach().cpu().numpy().copy() for k, v in bench.network.state_dict().items()}
    kw = dict(clip=0.5, lr=0.00025) if head == "c51" else dict(clip=5.0, lr=0.00005)
    orc = AsyncDqnScheduleOracle(p_np, p_np, cap, b, seed, n_actions=a, epsilon=bench.epsilon, eps=0.01 / 32, head=head, n_atoms=n, **kw)
    rng_state = np.random.get_state()
    L = bench.learner
    np.random.seed(5)
    gpu_vec, gpu_state = [], []
    n_vec = b if head == "c51" else n
    for _ in range(steps):
        bench.step()
        L.synchronize()
        gpu_vec.append(d.ops._wrap_device_pointer(L.delta.data_ptr(), n_vec, torch.float32).cpu().numpy().copy())
        gpu_state.append(L.export_state())
    n_tr = 4 * (steps + 1)
    gpu_actions = d.ops._wrap_device_pointer(bench.ring.pointers()[1], n_tr, torch.int64).cpu().numpy().copy()
    gpu_frames = d.ops._wrap_device_pointer(bench.ring.pointers()[0], n_tr * 7056, torch.uint8).cpu().numpy().copy()
    np.random.seed(5)
    near_ties, strict = 0, 0

    def check_actions(res, first, who):
        nonlocal near_ties
        for e, (act, gap, rnd) in enumerate(res):
            got = gpu_actions[first + e]
            if act != got:
                assert (not rnd) and gap < 1e-5, "%s env step %d: action %d vs %d, top-2 gap %g" % (who, e, act, got, gap)
                near_ties += 1

    check_actions(orc.actor_step(orc._snapshot(), override_actions=gpu_actions[0:4]), 0, "actor(0)")
    for k in range(steps):
        idx, batch = orc.sample()
        theta = orc._snapshot()
        check_actions(orc.actor_step(theta, override_actions=gpu_actions[4 * (k + 1):4 * (k + 2)]), 4 * (k + 1), "actor(%d)" % (k + 1))
        loss, vec, out, norm = orc.update(batch)
        ambiguous = orc.relu_margin < 5e-7
        f = 100.0 if ambiguous else 1.0
        strict += not ambiguous
        scale = max(1e-3, float(np.abs(vec).max()))
        perr = max(float(np.abs(gpu_state[k]["params"][nm].numpy() - orc.p[nm].detach().numpy()).max()) for nm in orc.names)
        _record_parity("schedule_oracle_%s[%d] step %d%s" % (head, cap, k, " (ambiguous ReLU gate)" if ambiguous else ""),
                       loss_vec=_rel(gpu_vec[k], vec, scale), loss=abs(float(np.mean(gpu_vec[k].astype(np.float64))) - loss) / max(abs(loss), scale),
                       params_abs=perr, relu_margin=orc.relu_margin)
        msg = "step %d: relu margin %.1e, max param err %.1e" % (k, orc.relu_margin, perr)
        np.testing.assert_allclose(gpu_vec[k], vec, rtol=1e-5 * f, atol=1e-5 * scale * f, err_msg="loss vector, " + msg)
        # the mean gets the same bar as its components: the categorical loss is a KL divergence, a difference of O(4)
        # cross-entropy terms whose fp32 rounding (~4e-7) is a 1e-5 fraction of a 0.03 mean
        np.testing.assert_allclose(float(np.mean(gpu_vec[k].astype(np.float64))), loss, rtol=1e-5 * f, atol=1e-5 * scale * f,
                                   err_msg="loss, " + msg)
        for nm in orc.names:
            np.testing.assert_allclose(gpu_state[k]["params"][nm].numpy(), orc.p[nm].detach().numpy(), rtol=1e-5 * f,
                                       atol=5e-6 * f, err_msg=nm + ", " + msg)
        if ambiguous:
            orc.load_state(gpu_state[k])
    assert near_ties <= 1
    assert strict >= steps // 2, "too few unambiguous steps to mean anything"
    assert np.array_equal(gpu_frames, orc.rep.state[:n_tr].reshape(n_tr * 7056))
    np.random.set_state(rng_state)
    L.close()
    bench.ring.close()


def test_kernel_replay_measures_without_side_effects(dra):
    """dra_dqn_learner_kernel_replay (round 6: bench.py's live roofline reading) replays ONE kernel group of the update alone, 64
    dependent launches in a captured graph.  It must (i) return a per-launch time above the empty kernel's, (ii) refuse the
    groups a replay must not run (the optimizer, groups that ride in another group's launch), and (iii) leave the learner exactly
    where it was: two benches on the same seeds, one of them replaying every replayable group between its steps, end on the
    same parameters bit for bit."""
    d = dra
    from deeprl_amd._lib import DraError
    from deeprl_amd.learner import DQNLearnerBench
    outs = []
    for replay in (False, True):
        np.random.seed(3)
        torch.manual_seed(4)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=5, actor=True, async_actor=True)
        for _ in range(12):
            b.step()
        if replay:
            for name in ("conv1_fwd", "conv2_fwd", "conv3_fwd", "fc4_fwd", "head_loss", "fc4_bwd_x", "conv3_bwd_x", "conv2_bwd_x",
                         "conv1_bwd_w"):
                b.learner.synchronize()
                per, empty = b.learner.kernel_replay(name, 16)
                assert 0.5 < empty < per < 200.0, (name, per, empty)
            for name in ("rmsprop_step", "grad_norm", "conv2_bwd_w", "gather"):
                with pytest.raises(DraError):
                    b.learner.kernel_replay(name, 4)
            # the chained launches the timed pipeline actually runs (DRA_VAR_FWD_CHAIN / DRA_VAR_BWD_CHAIN): dra_dqn_learner_chain_replay
            for which in ("fwd", "bwd", "fwd"):
                b.learner.synchronize()
                per, empty = b.learner.chain_replay(which, 16)
                assert 1.0 < empty < per < 400.0, (which, per, empty)
        for _ in range(6):
            b.step()
        b.learner.synchronize()
        outs.append(b.learner.flat.flat.detach().cpu().numpy().copy())
        b.learner.close()
        b.ring.close()
    assert np.array_equal(outs[0], outs[1])


def test_deferred_fc4_step_is_bit_identical(dra):
    """DRA_VAR_DEFER_FC4 (round 6): the pipelined graphs' optimizer launch leaves fc4's weights (95 % of the parameters) to rider
    workgroups in the NEXT update's conv1 / conv2 forward launches; the actor's copy of them is guarded by a device word, and
    everything that reads parameters outside those graphs flushes first.  Same arithmetic, hence the same bits: the benchmarked
    pipeline with the bit set and cleared ends on identical parameters, optimizer state, ring contents and target network --
    also when one of the runs is interrupted by synchronise() (a flush), a target sync and kernel replays in the middle."""
    d = dra
    from deeprl_amd import ops
    from deeprl_amd.learner import DQNLearnerBench
    default = ops.get_tuning()
    assert default & ops.VAR_DEFER_FC4, "the library default carries DRA_VAR_DEFER_FC4"
    outs = []
    for variant, interrupt in ((default & ~ops.VAR_DEFER_FC4, False), (default, False), (default, True)):
        np.random.seed(11)
        torch.manual_seed(12)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=13, actor=True, async_actor=True, variant=variant)
        L = b.learner
        for t in range(30):
            b.step()
            if t == 14:
                L.sync_target()                       # DQN_agent.py:136-138 in the middle of the pipeline
            if interrupt and t in (7, 8, 21):
                L.synchronize()                       # a flush between two riding graphs
            if interrupt and t == 19:
                L.kernel_replay("conv2_bwd_x", 4)
        L.synchronize()
        frames = d.ops._wrap_device_pointer(b.ring.pointers()[0], 200 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(b.ring.pointers()[1], 200, torch.int64).cpu().numpy().copy()
        outs.append(dict(p=L.flat.flat.detach().cpu().numpy().copy(), s1=L.state1.detach().cpu().numpy().copy(),
                         s2=L.state2.detach().cpu().numpy().copy(), pt=L.target_flat.flat.detach().cpu().numpy().copy(),
                         frames=frames, acts=acts))
        L.close()
        b.ring.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), ("defer on vs off", k)
        assert np.array_equal(outs[1][k], outs[2][k]), ("interrupted vs not", k)
    assert float(np.abs(outs[0]["p"]).max()) > 0


def test_target_ahead_is_bit_identical(dra):
    """DRA_VAR_TARGET_AHEAD (round 6): target_network(next_states) of update t + 1 (DQN_agent.py:85-88) -- conv1-3 + fc4 of the target
    net, which depend on the target parameters and the ring alone -- is issued one call early on its own stream and runs under
    update t; the update's forward chain carries the online net alone and the head kernel folds the target's partial sums from a
    stash.  Same kernels on the same data: the benchmarked pipeline with the bit set and cleared ends on identical parameters,
    optimizer state, target network, ring contents and actions -- on a 4096-slot ring, where minibatches regularly read slots the
    actor launches around them write (those updates compute their target in line), across synchronise() calls, a target sync and
    kernel / chain replays in the middle (each leaves the lane and drops the stash in flight)."""
    d = dra
    from deeprl_amd import ops
    from deeprl_amd.learner import DQNLearnerBench
    default = ops.get_tuning()
    outs, stats = [], []
    for variant, interrupt in ((default & ~ops.VAR_TARGET_AHEAD, False), (default | ops.VAR_TARGET_AHEAD, False),
                               (default | ops.VAR_TARGET_AHEAD, True)):
        np.random.seed(31)
        torch.manual_seed(32)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=33, actor=True, async_actor=True, variant=variant)
        L = b.learner
        for t in range(400):
            b.step()
            if t == 170:
                L.sync_target()
            if interrupt and t in (50, 51, 290):
                L.synchronize()
            if interrupt and t == 120:
                L.kernel_replay("conv2_bwd_x", 4)
            if interrupt and t == 230:
                L.chain_replay("fwd", 4)
        L.synchronize()
        stats.append(L.ahead_stats())
        frames = d.ops._wrap_device_pointer(b.ring.pointers()[0], 4096 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(b.ring.pointers()[1], 4096, torch.int64).cpu().numpy().copy()
        outs.append(dict(p=L.flat.flat.detach().cpu().numpy().copy(), s1=L.state1.detach().cpu().numpy().copy(),
                         s2=L.state2.detach().cpu().numpy().copy(), pt=L.target_flat.flat.detach().cpu().numpy().copy(),
                         frames=frames, acts=acts, q=L.actor_q.cpu().numpy().copy()))
        L.close()
        b.ring.close()
    assert not stats[0]["active"] and stats[1]["active"] and stats[2]["active"]
    assert stats[1]["from_stash"] > 250, stats          # (the lane's steady state is served from the stash ...)
    assert stats[1]["skipped_slot_hazard"] > 0, stats   # (... except where the 4096-slot ring's hazards forbid it)
    assert stats[1]["index_mismatch"] == 0, stats
    for k in outs[0]:
        for i in (1, 2):
            assert np.array_equal(outs[0][k], outs[i][k]), ("target ahead vs in the update's chain", i, k)
    assert float(np.abs(outs[0]["q"]).max()) > 0


def test_head_chain_is_bit_identical(dra):
    """DRA_VAR_HEAD_CHAIN (round 6): the head launch (fc4 fold, head, TD error, dq, dh4: DQN_agent.py:85-99) and fc4's + the head's
    backward launch as ONE launch in dependency order -- the backward roles request fc4's weights and conv3's activations first,
    then wait on one arrival counter (zeroed by the next update's forward chain) for the head role's workgroups.  Same arithmetic
    in the same order: the benchmarked pipeline with the bit set and cleared ends on identical parameters, optimizer state, target
    network and ring contents -- across synchronise() calls, a target sync, kernel replays (which run the two launches on their own)
    and chain replays in the middle."""
    d = dra
    from deeprl_amd import ops
    from deeprl_amd.learner import DQNLearnerBench
    default = ops.get_tuning()
    outs = []
    for variant, interrupt in ((default & ~ops.VAR_HEAD_CHAIN, False), (default | ops.VAR_HEAD_CHAIN, False),
                               (default | ops.VAR_HEAD_CHAIN, True)):
        np.random.seed(51)
        torch.manual_seed(52)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=53, actor=True, async_actor=True, variant=variant)
        L = b.learner
        for t in range(60):
            b.step()
            if t == 27:
                L.sync_target()
            if interrupt and t in (5, 6, 39):
                L.synchronize()
            if interrupt and t == 11:
                L.kernel_replay("head_loss", 4)
                L.kernel_replay("fc4_bwd_x", 4)
            if interrupt and t == 33:
                L.chain_replay("bwd", 4)
                L.chain_replay("fwd", 4)
        L.synchronize()
        frames = d.ops._wrap_device_pointer(b.ring.pointers()[0], 300 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(b.ring.pointers()[1], 300, torch.int64).cpu().numpy().copy()
        outs.append(dict(p=L.flat.flat.detach().cpu().numpy().copy(), s1=L.state1.detach().cpu().numpy().copy(),
                         s2=L.state2.detach().cpu().numpy().copy(), pt=L.target_flat.flat.detach().cpu().numpy().copy(),
                         frames=frames, acts=acts, q=L.q.detach().cpu().numpy().copy(), delta=L.delta.detach().cpu().numpy().copy()))
        L.close()
        b.ring.close()
    for k in outs[0]:
        for i in (1, 2):
            assert np.array_equal(outs[0][k], outs[i][k]), ("head + fc4 backward as one launch vs two", i, k)
    assert float(np.abs(outs[0]["p"]).max()) > 0 and float(np.abs(outs[0]["delta"]).max()) > 0


def test_persistent_actor_is_bit_identical(dra):
    """DRA_VAR_ACTOR_PERSIST (round 6): the whole agent step of the device actor -- n_env x [forward, epsilon-greedy, env.step]
    (DQN_agent.py:24-45) -- as ONE launch of 32 co-resident workgroups whose activations cross workgroups as {value, tag} words.
    Same arithmetic in the same order as the multi-launch env step, hence the same bits: the benchmarked pipeline with the bit
    set and cleared ends on identical parameters, optimizer state, target network, ring contents (frames and ACTIONS) and action
    values of the last step -- also across synchronise() calls and a target sync in the middle, and with DRA_VAR_DEFER_FC4's
    guarded fc4 segment on either side."""
    d = dra
    from deeprl_amd import ops
    from deeprl_amd.learner import DQNLearnerBench
    default = ops.get_tuning()
    outs = []
    for variant, interrupt in ((default & ~ops.VAR_ACTOR_PERSIST, False), (default | ops.VAR_ACTOR_PERSIST, False),
                               (default | ops.VAR_ACTOR_PERSIST, True),
                               ((default | ops.VAR_ACTOR_PERSIST) & ~ops.VAR_DEFER_FC4, False)):
        np.random.seed(21)
        torch.manual_seed(22)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=23, actor=True, async_actor=True, variant=variant)
        L = b.learner
        for t in range(40):
            b.step()
            if t == 17:
                L.sync_target()
            if interrupt and t in (5, 6, 29):
                L.synchronize()
        L.synchronize()
        frames = d.ops._wrap_device_pointer(b.ring.pointers()[0], 260 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(b.ring.pointers()[1], 260, torch.int64).cpu().numpy().copy()
        outs.append(dict(p=L.flat.flat.detach().cpu().numpy().copy(), s1=L.state1.detach().cpu().numpy().copy(),
                         s2=L.state2.detach().cpu().numpy().copy(), pt=L.target_flat.flat.detach().cpu().numpy().copy(),
                         frames=frames, acts=acts, q=L.actor_q.cpu().numpy().copy()))
        L.close()
        b.ring.close()
    for k in outs[0]:
        for i in (1, 2, 3):
            assert np.array_equal(outs[0][k], outs[i][k]), ("persistent vs multi-launch actor", i, k)
    assert float(np.abs(outs[0]["q"]).max()) > 0
    assert len(set(outs[0]["acts"][:160].tolist())) > 1


def test_forward_chain_is_bit_identical(dra):
    """DRA_VAR_FWD_CHAIN (round 6): conv1 + conv2 + conv3 of the update's forward pass (DQN_agent.py:81-99 through
    network_bodies.py:10-33, both nets) as ONE launch whose workgroups wait for the workgroups of THEIR sample in the layer below
    (arrival counters that are never reset, targets relative to the number of chains completed).  Same arithmetic in the same
    order, hence the same bits: the benchmarked pipeline with the bit set and cleared ends on identical parameters, optimizer
    state, target network and ring contents -- also across synchronise() calls, a target sync, kernel replays (which run the
    three launches on their own) and an eager profile in the middle."""
    d = dra
    from deeprl_amd import ops
    from deeprl_amd.learner import DQNLearnerBench
    default = ops.get_tuning()
    outs = []
    for variant, interrupt in ((default & ~ops.VAR_FWD_CHAIN, False), (default | ops.VAR_FWD_CHAIN, False),
                               (default | ops.VAR_FWD_CHAIN, True)):
        np.random.seed(31)
        torch.manual_seed(32)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=33, actor=True, async_actor=True, variant=variant)
        L = b.learner
        for t in range(40):
            b.step()
            if t == 17:
                L.sync_target()
            if interrupt and t in (5, 6, 29):
                L.synchronize()
            if interrupt and t == 11:
                L.kernel_replay("conv2_fwd", 4)
                L.kernel_replay("conv2_bwd_x", 4)
        L.synchronize()
        frames = d.ops._wrap_device_pointer(b.ring.pointers()[0], 260 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(b.ring.pointers()[1], 260, torch.int64).cpu().numpy().copy()
        outs.append(dict(p=L.flat.flat.detach().cpu().numpy().copy(), s1=L.state1.detach().cpu().numpy().copy(),
                         s2=L.state2.detach().cpu().numpy().copy(), pt=L.target_flat.flat.detach().cpu().numpy().copy(),
                         frames=frames, acts=acts))
        L.close()
        b.ring.close()
    for k in outs[0]:
        for i in (1, 2):
            assert np.array_equal(outs[0][k], outs[i][k]), ("chained vs separate forward launches", i, k)
    assert float(np.abs(outs[0]["p"]).max()) > 0


def test_flag_sync_lane_is_bit_identical(dra):
    """DRA_VAR_FLAG_SYNC (round 6): the steady-state pipelined step (DQN_agent.py:101-138 under BaseAgent.py:108-182's async actor)
    records no event and waits for none -- the update graph's first launch counts itself in a device word, the actor launch
    polls that word for the number the host left in a pinned ring, the host paces itself on a pinned count the actor publishes.
    The same launches on the same data, hence the same bits: with the bit set and cleared the benchmarked pipeline ends on
    identical parameters, optimizer state, target network, ring contents (frames AND actions) and last action values -- on a
    4096-slot ring, where about every fifth step meets one of the two slot hazards (the minibatch reads slots the actor launch
    beside it writes, or an unfinished one wrote), and across the things that LEAVE the lane: synchronise(), a target sync,
    kernel replays, an actor-only call pattern change."""
    d = dra
    from deeprl_amd import ops
    from deeprl_amd.learner import DQNLearnerBench
    default = ops.get_tuning()
    assert default & ops.VAR_FLAG_SYNC, "the library default carries DRA_VAR_FLAG_SYNC"
    outs, stats = [], []
    for variant, interrupt in ((default & ~ops.VAR_FLAG_SYNC, False), (default, False), (default, True),
                               (default & ~ops.VAR_DEFER_FC4, False), (default ^ ops.VAR_LANE_EAGER, True)):
        np.random.seed(41)
        torch.manual_seed(42)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=43, actor=True, async_actor=True, variant=variant)
        L = b.learner
        for t in range(120):
            b.step()
            if t == 57:
                L.sync_target()
            if interrupt and t in (9, 10, 33, 90):
                L.synchronize()
            if interrupt and t == 21:
                L.kernel_replay("conv2_bwd_x", 4)
        L.synchronize()
        stats.append(L.lane_stats())
        frames = d.ops._wrap_device_pointer(b.ring.pointers()[0], 600 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(b.ring.pointers()[1], 600, torch.int64).cpu().numpy().copy()
        outs.append(dict(p=L.flat.flat.detach().cpu().numpy().copy(), s1=L.state1.detach().cpu().numpy().copy(),
                         s2=L.state2.detach().cpu().numpy().copy(), pt=L.target_flat.flat.detach().cpu().numpy().copy(),
                         frames=frames, acts=acts, q=L.actor_q.cpu().numpy().copy()))
        L.close()
        b.ring.close()
    for k in outs[0]:
        for i in (1, 2, 3, 4):
            assert np.array_equal(outs[0][k], outs[i][k]), ("lane vs event path", i, k)
    assert float(np.abs(outs[0]["p"]).max()) > 0
    assert stats[0]["steps"] == 0
    assert stats[1]["steps"] >= 100 and stats[1]["entries"] == 2, stats[1]       # (entered once, left for the target sync, entered again)
    assert stats[1]["hazard_bumps"] + stats[1]["host_waits"] >= 5, stats[1]      # (the small ring did produce hazards)
    assert stats[2]["steps"] >= 80 and stats[2]["entries"] >= 5, stats[2]
    assert stats[3]["steps"] >= 100, stats[3]
    assert stats[4]["steps"] >= 80, stats[4]              # (the update as plain launches / as a graph replay: the other form)


def test_backward_chain_is_bit_identical(dra):
    """DRA_VAR_BWD_CHAIN (round 6): conv1 + conv2 + conv3 of the update's forward pass (DQN_agent.py:81-99 through
    network_bodies.py:10-33, both nets) as ONE launch whose workgroups wait for the workgroups of THEIR sample in the layer below
    (arrival counters that are never reset, targets relative to the number of chains completed).  Same arithmetic in the same
    order, hence the same bits: the benchmarked pipeline with the bit set and cleared ends on identical parameters, optimizer
    state, target network and ring contents -- also across synchronise() calls, a target sync, kernel replays (which run the
    three launches on their own) and an eager profile in the middle."""
    d = dra
    from deeprl_amd import ops
    from deeprl_amd.learner import DQNLearnerBench
    default = ops.get_tuning()
    outs = []
    # (DRA_VAR_BWD_CHAIN_FC: fc4's + the head's backward as the leading roles of the same launch -- cleared and set, the latter
    # also with the chained launch replayed alone in the middle)
    fcbit = ops.VAR_BWD_CHAIN_FC
    for variant, interrupt in (((default & ~ops.VAR_BWD_CHAIN) & ~fcbit, False), ((default | ops.VAR_BWD_CHAIN) & ~fcbit, False),
                               ((default | ops.VAR_BWD_CHAIN) & ~fcbit, True), (default | ops.VAR_BWD_CHAIN | fcbit, False),
                               (default | ops.VAR_BWD_CHAIN | fcbit, True)):
        np.random.seed(41)
        torch.manual_seed(42)
        b = DQNLearnerBench(ring_capacity=4096, batch=32, seed=43, actor=True, async_actor=True, variant=variant)
        L = b.learner
        for t in range(40):
            b.step()
            if t == 17:
                L.sync_target()
            if interrupt and t in (5, 6, 29):
                L.synchronize()
            if interrupt and t == 11:
                L.kernel_replay("conv2_fwd", 4)
                L.kernel_replay("conv2_bwd_x", 4)
            if interrupt and t == 23:
                L.chain_replay("bwd", 4)
        L.synchronize()
        frames = d.ops._wrap_device_pointer(b.ring.pointers()[0], 260 * 7056, torch.uint8).cpu().numpy().copy()
        acts = d.ops._wrap_device_pointer(b.ring.pointers()[1], 260, torch.int64).cpu().numpy().copy()
        outs.append(dict(p=L.flat.flat.detach().cpu().numpy().copy(), s1=L.state1.detach().cpu().numpy().copy(),
                         s2=L.state2.detach().cpu().numpy().copy(), pt=L.target_flat.flat.detach().cpu().numpy().copy(),
                         frames=frames, acts=acts))
        L.close()
        b.ring.close()
    for k in outs[0]:
        for i in (1, 2, 3, 4):
            assert np.array_equal(outs[0][k], outs[i][k]), ("chained vs separate backward launches", i, k)
    assert float(np.abs(outs[0]["p"]).max()) > 0
