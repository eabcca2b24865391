"""Pins the CPU oracle (oracle/) against fixtures generated from the reference's
own code (tests/golden/make_golden.py).  Integer / byte / index / fp64 work is
compared bit-exactly; fp32 losses and gradients at 1e-5 (north_star tolerance)."""
import os
import random

import numpy as np
import pytest
import torch

from golden.make_golden_cases import UNIFORM_CASES, PER_CASES, stream
from oracle import loss_oracle as L
from oracle import net_oracle as N
from oracle import numerics_oracle as NUM
from oracle.replay_oracle import PrioritizedReplayOracle, UniformReplayOracle
from oracle.sumtree_oracle import SumTreeOracle

TOL = dict(rtol=1e-5, atol=1e-5)


def _feed(rep, states, actions, rewards, masks, t):
    rep.feed_one(states[t], actions[t], rewards[t], masks[t])


@pytest.mark.parametrize("case", UNIFORM_CASES, ids=[c[0] for c in UNIFORM_CASES])
def test_uniform_replay(golden, case):
    g = golden("uniform_replay")
    name, mem, b, h, n, disc, shape, kind, t_len, cps = case
    states, actions, rewards, masks = stream(np.random.RandomState(1000 + ord(name)), t_len, shape, kind, 4, 0.1)
    rep = UniformReplayOracle(mem, b, n, disc, h)
    np.random.seed(2000 + ord(name))
    for t in range(t_len):
        _feed(rep, states, actions, rewards, masks, t)
        if t in cps:
            st, ac, rw, ns, mk, idx = rep.sample()
            k = "%s_t%d_" % (name, t)
            assert np.array_equal(idx, g[k + "idx"])
            assert np.array_equal(g[k + "pos_size"], [rep.pos, rep.size()])
            for got, key in ((st, "state"), (ac, "action"), (rw, "reward"), (ns, "next_state"), (mk, "mask")):
                want = g[k + key]
                assert got.shape == want.shape and got.dtype == want.dtype, key
                assert np.array_equal(got, want), key
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g[name + "_rng_tail"])


@pytest.mark.parametrize("case", PER_CASES, ids=[c[0] for c in PER_CASES])
def test_prioritized_replay(golden, case):
    g = golden("prioritized_replay")
    name, mem, b, h, n, disc, shape, kind, t_len, every = case
    rs = np.random.RandomState(3000 + ord(name))
    states, actions, rewards, masks = stream(rs, t_len, shape, kind, 4, 0.1)
    rep = PrioritizedReplayOracle(mem, b, n, disc, h)
    random.seed(4000 + ord(name))
    np.random.seed(4000 + ord(name))
    ks = 0
    for t in range(t_len):
        _feed(rep, states, actions, rewards, masks, t)
        if t >= h + n + 6 and t % every == 0:
            st, ac, rw, ns, mk, prob, tidx = rep.sample()
            k = "%s_s%d_" % (name, ks)
            assert int(g[k + "t"]) == t
            for got, key in ((st, "state"), (ac, "action"), (rw, "reward"), (ns, "next_state"), (mk, "mask"),
                             (prob, "sampling_prob"), (tidx, "idx")):
                assert np.array_equal(got, g[k + key]), (key, ks)
            rs.standard_normal(b)  # the generator drew the fake loss here
            prio = g[k + "prio"]
            assert prio.dtype == np.float32
            rep.update_priorities(zip(tidx, prio))
            assert np.array_equal(rep.tree.tree, g[k + "tree"]), ks
            assert float(rep.max_priority) == float(g[k + "max_priority"])
            ks += 1
    assert ks == int(g[name + "_n_samples"])
    assert np.array_equal(rep.tree.tree, g[name + "_tree_final"])
    assert np.array_equal([random.random() for _ in range(3)], g[name + "_rng_tail"])
    # exactness regime: incremental tree == bottom-up rebuild
    assert np.array_equal(rep.tree.tree, rep.tree.rebuilt())


@pytest.mark.parametrize("cap", [8, 13, 50])
def test_sumtree_ops(golden, cap):
    g = golden("sumtree")
    tree = SumTreeOracle(cap)
    for op, x, p_out, idx in g["cap%d_log" % cap]:
        op, idx = int(op), int(idx)
        if op == 0:
            tree.add(np.float32(x))
        elif op == 1:
            i, p, d = tree.get(x)
            assert i == idx and float(p) == p_out and d == i - cap + 1
        else:
            tree.update(idx, np.float32(x))
    assert np.array_equal(tree.tree, g["cap%d_tree" % cap])
    assert np.array_equal(sorted(tree.pending), g["cap%d_pending" % cap])


def _t(x, grad=False):
    t = torch.tensor(np.asarray(x, dtype=np.float32))
    t.requires_grad_(grad)
    return t


@pytest.mark.parametrize("tag", ["b32a4", "b10a2n3", "b32a4dq", "b7a18"])
def test_dqn_loss(golden, tag):
    g = golden("dqn_loss")
    k = tag + "_"
    gamma, n_step, double_q, eps, alpha, beta = g[k + "cfg"]
    q = _t(g[k + "q"], True)
    delta = L.dqn_td_error(q, _t(g[k + "q_next_t"]), torch.tensor(g[k + "action"]), _t(g[k + "reward"]),
                           _t(g[k + "mask"]), gamma ** int(n_step),
                           _t(g[k + "q_next_o"]) if double_q else None)
    np.testing.assert_allclose(delta.detach().numpy(), g[k + "loss_vec"], **TOL)
    loss = L.dqn_reduce(delta)
    np.testing.assert_allclose(loss.item(), g[k + "loss"], **TOL)
    gq, = torch.autograd.grad(loss, q, retain_graph=True)
    np.testing.assert_allclose(gq.numpy(), g[k + "grad_q"], **TOL)
    prio, w, wl = L.per_priorities_and_weights(delta, _t(g[k + "sampling_prob"]), eps, alpha, beta)
    np.testing.assert_allclose(prio.detach().numpy(), g[k + "prio"], **TOL)
    np.testing.assert_allclose(w.numpy(), g[k + "w"], **TOL)
    lp = L.dqn_reduce(wl)
    np.testing.assert_allclose(lp.item(), g[k + "loss_per"], **TOL)
    gq, = torch.autograd.grad(lp, q)
    np.testing.assert_allclose(gq.numpy(), g[k + "grad_q_per"], **TOL)


@pytest.mark.parametrize("tag", ["b32a4", "b8a3n3dq", "b5a6at21"])
def test_c51_loss(golden, tag):
    g = golden("c51_loss")
    k = tag + "_"
    gamma, n_step, double_q, vmin, vmax, n_atoms = g[k + "cfg"]
    n_atoms = int(n_atoms)
    atoms = _t(np.linspace(vmin, vmax, n_atoms))
    logits = _t(g[k + "logits"], True)
    sm = lambda z: torch.softmax(z, dim=-1)
    kl = L.c51_kl(torch.log_softmax(logits, dim=-1), sm(_t(g[k + "logits_next_t"])), torch.tensor(g[k + "action"]),
                  _t(g[k + "reward"]), _t(g[k + "mask"]), gamma ** int(n_step), atoms, vmin, vmax,
                  sm(_t(g[k + "logits_next_o"])) if double_q else None)
    np.testing.assert_allclose(kl.detach().numpy(), g[k + "kl"], **TOL)
    loss = kl.mean()
    np.testing.assert_allclose(loss.item(), g[k + "loss"], **TOL)
    gl, = torch.autograd.grad(loss, logits)
    np.testing.assert_allclose(gl.numpy(), g[k + "grad_logits"], **TOL)


@pytest.mark.parametrize("tag", ["b32a4", "b6a3q17n3"])
def test_qr_loss(golden, tag):
    g = golden("qr_loss")
    k = tag + "_"
    gamma, n_step, nq = g[k + "cfg"]
    theta = _t(g[k + "theta"], True)
    lv = L.qr_loss(theta, _t(g[k + "theta_next_t"]), torch.tensor(g[k + "action"]), _t(g[k + "reward"]),
                   _t(g[k + "mask"]), gamma ** int(n_step))
    np.testing.assert_allclose(lv.detach().numpy(), g[k + "loss_vec"], rtol=1e-5, atol=1e-5)
    loss = lv.mean()
    np.testing.assert_allclose(loss.item(), g[k + "loss"], **TOL)
    gt, = torch.autograd.grad(loss, theta)
    np.testing.assert_allclose(gt.numpy(), g[k + "grad_theta"], **TOL)


@pytest.mark.parametrize("tag", ["m64", "m256", "m5"])
def test_ppo_loss(golden, tag):
    g = golden("ppo_loss")
    k = tag + "_"
    lp, ent, v = _t(g[k + "lp"], True), _t(g[k + "ent"], True), _t(g[k + "v"], True)
    pl, vl, kl = L.ppo_losses(lp, ent, v, _t(g[k + "old_lp"]), _t(g[k + "adv"]), _t(g[k + "ret"]), 0.2, 0.01)
    np.testing.assert_allclose([pl.item(), vl.item(), kl.item()], g[k + "out"], **TOL)
    gs = torch.autograd.grad(pl + vl, [lp, ent, v])
    for got, key in zip(gs, ("g_lp", "g_ent", "g_v")):
        np.testing.assert_allclose(got.numpy(), g[k + key], **TOL)


@pytest.mark.parametrize("tag", ["t5n16", "t5n16gae", "t20n3gae"])
def test_a2c_gae_and_step(golden, tag):
    g = golden("a2c_step")
    k = tag + "_"
    gamma, tau, use_gae, ew, vw, clip, t_len, n_env = g[k + "cfg"]
    adv, ret = L.gae_reverse(_t(g[k + "reward"]), _t(g[k + "mask"]), _t(g[k + "v"]), gamma, tau, bool(use_gae))
    np.testing.assert_allclose(adv.numpy(), g[k + "adv"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ret.numpy(), g[k + "ret"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["t64n2", "t32n4"])
def test_ppo_gae_and_normalize(golden, tag):
    g = golden("ppo_step")
    k = tag + "_"
    gamma, tau = g[k + "cfg"][:2]
    adv, ret = L.gae_reverse(_t(g[k + "reward"]), _t(g[k + "mask"]), _t(g[k + "v"]), gamma, tau, True)
    np.testing.assert_allclose(adv.numpy(), g[k + "adv"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ret.numpy(), g[k + "ret"], rtol=1e-6, atol=1e-6)
    flat = adv.reshape(-1, 1)
    np.testing.assert_allclose(L.normalize_advantage(flat).numpy(), g[k + "ent_adv_normalized"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.reshape(-1, 1).numpy(), g[k + "ent_ret"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["rmsprop_centered", "rmsprop_plain", "adam", "adam_default"])
@pytest.mark.parametrize("clip", [5.0, 0.5])
def test_optim(golden, name, clip):
    g = golden("optim")
    n_t = 4
    ps = [_t(g["p0_%d" % j]) for j in range(n_t)]
    s1 = [torch.zeros_like(p) for p in ps]
    s2 = [torch.zeros_like(p) for p in ps]
    hp = {"rmsprop_centered": dict(lr=0.00025, alpha=0.95, eps=0.01, centered=True),
          "rmsprop_plain": dict(lr=1e-4, alpha=0.99, eps=1e-5, centered=False),
          "adam": dict(lr=2.5e-4, beta1=0.9, beta2=0.999, eps=0.01 / 32),
          "adam_default": dict(lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8)}[name]
    norms = []
    for i in range(5):
        gs = [_t(g["g%d_%d" % (i, j)]) for j in range(n_t)]
        norm, gs = N.clip_grad_norm(gs, clip)
        norms.append(float(norm))
        for j in range(n_t):
            if name.startswith("rmsprop"):
                ps[j], s1[j], s2[j] = N.rmsprop_step(ps[j], gs[j], s1[j], s2[j], **hp)
            else:
                ps[j], s1[j], s2[j] = N.adam_step(ps[j], gs[j], s1[j], s2[j], i + 1, **hp)
    np.testing.assert_allclose(norms, g["%s_clip%g_norms" % (name, clip)], rtol=1e-6)
    for j in range(n_t):
        np.testing.assert_allclose(ps[j].numpy(), g["%s_clip%g_p%d" % (name, clip, j)], rtol=1e-5, atol=1e-6)


def test_dqn_nature_update(golden):
    """Full reference DQN update on VanillaNet(NatureConvBody): oracle forward/backward
    (F.conv2d chain) + clip + centered RMSprop reproduce the reference trajectory."""
    import fake_envs
    g = golden("dqn_nature_update")
    b, a = 8, 4
    rs = np.random.RandomState(int(g["state_seed"]))
    p = {k: torch.tensor(v, requires_grad=True) for k, v in fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 11).items()}
    pt = {k: torch.tensor(v) for k, v in fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 12).items()}
    state = rs.randint(0, 256, size=(b, 4, 84, 84)).astype(np.uint8)
    next_state = rs.randint(0, 256, size=(b, 4, 84, 84)).astype(np.uint8)
    action = rs.randint(0, a, size=b).astype(np.int64)
    reward = np.sign(rs.standard_normal(b))
    mask = (rs.rand(b) > 0.2).astype(np.int32)
    assert np.array_equal(action, g["action"]) and np.array_equal(reward, g["reward"]) and np.array_equal(mask, g["mask"])
    x = torch.from_numpy(NUM.image_normalize_sync(state))
    xn = torch.from_numpy(NUM.image_normalize_sync(next_state))
    q0 = N.vanilla_head(p, N.nature_conv_body(p, x))
    np.testing.assert_allclose(q0.detach().numpy(), g["q0"], **TOL)
    names = list(p.keys())
    sq = {k: torch.zeros_like(v) for k, v in p.items()}
    ga = {k: torch.zeros_like(v) for k, v in p.items()}
    for it in range(3):
        q = N.vanilla_head(p, N.nature_conv_body(p, x))
        with torch.no_grad():
            qn = N.vanilla_head(pt, N.nature_conv_body(pt, xn))
        delta = L.dqn_td_error(q, qn, torch.tensor(action), _t(reward), _t(mask), 0.99)
        loss = L.dqn_reduce(delta)
        grads = torch.autograd.grad(loss, [p[k] for k in names])
        if it == 0:
            for k, gr in zip(names, grads):
                if k == "body.fc4.weight":
                    np.testing.assert_allclose(gr.numpy()[::37], g["grad_" + k + "_rows"], rtol=1e-4, atol=1e-6)
                    np.testing.assert_allclose(np.sqrt((gr.numpy().astype(np.float64) ** 2).sum()), g["grad_" + k + "_norm"], rtol=1e-5)
                else:
                    np.testing.assert_allclose(gr.numpy(), g["grad_" + k], rtol=1e-4, atol=1e-6)
        norm, grads = N.clip_grad_norm(list(grads), 5)
        np.testing.assert_allclose([loss.item(), float(norm)], g["loss_gradnorm_traj"][it], rtol=5e-5)
        with torch.no_grad():
            for k, gr in zip(names, grads):
                newp, sq[k], ga[k] = N.rmsprop_step(p[k], gr, sq[k], ga[k], 0.00025, 0.95, 0.01, True)
                p[k].copy_(newp)
    for k in names:
        want = g["final_" + k + "_rows"] if k == "body.fc4.weight" else g["final_" + k]
        got = p[k].detach().numpy()[::37] if k == "body.fc4.weight" else p[k].detach().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_dueling_rainbow_heads(golden):
    """oracle/net_oracle.py's DuelingNet / RainbowNet + NoisyLinear restatements against the reference's own modules
    (tests/golden/rainbow_dueling.npz: outputs, gradient digests, and the noise the reference's reset_noise() drew)."""
    import math
    import fake_envs
    from golden.make_golden_cases import DUELING_SHAPES, RAINBOW_SHAPES, NOISY_LAYERS, digest, head_inputs
    g = golden("rainbow_dueling")
    x, wq, wl = head_inputs()
    xn = torch.from_numpy(NUM.image_normalize_sync(x))
    # Dueling
    p = {k: torch.from_numpy(v).requires_grad_() for k, v in fake_envs.numpy_params(DUELING_SHAPES, 31).items()}
    q = N.dueling_head(p, N.nature_conv_body(p, xn))
    np.testing.assert_allclose(q.detach().numpy(), g["dueling_q"], rtol=1e-5, atol=1e-6)
    (q * torch.from_numpy(wq)).sum().backward()
    for n, v in p.items():
        np.testing.assert_allclose(digest(v.grad.numpy()), g["dueling_grad_" + n], rtol=1e-4, atol=1e-5, err_msg=n)
    # Rainbow: mu from the seeded generator, sigma the constructor's constant, epsilon from the reference's noise vectors
    p = {k: torch.from_numpy(v) for k, v in fake_envs.numpy_params(RAINBOW_SHAPES, 33).items()}
    for layer in NOISY_LAYERS:
        fan_in, fan_out = p[layer + ".weight_mu"].shape[1], p[layer + ".weight_mu"].shape[0]
        p[layer + ".weight_sigma"] = torch.full((fan_out, fan_in), 0.4 / math.sqrt(fan_in))
        p[layer + ".bias_sigma"] = torch.full((fan_out,), 0.4 / math.sqrt(fan_out))
        assert p[layer + ".weight_sigma"][0, 0].item() == g["rainbow_%s.weight_sigma0" % layer][0]
        assert p[layer + ".bias_sigma"][0].item() == g["rainbow_%s.bias_sigma0" % layer][0]
        we, be = N.noisy_epsilon(*[torch.from_numpy(g["rainbow_%s.%s" % (layer, b)])
                                     for b in ("noise_in", "noise_out_weight", "noise_out_bias")])
        assert np.array_equal(digest(we.numpy()), g["rainbow_%s.weight_epsilon" % layer])
        p[layer + ".weight_epsilon"], p[layer + ".bias_epsilon"] = we, be
    leaves = [k for k in p if not k.endswith("epsilon")]
    for k in leaves:
        p[k].requires_grad_()
    prob, log_prob = N.rainbow_head(p, N.nature_conv_body_noisy(p, xn), 4, 51)
    np.testing.assert_allclose(prob.detach().numpy(), g["rainbow_prob"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(log_prob.detach().numpy(), g["rainbow_log_prob"], rtol=1e-5, atol=1e-6)
    (log_prob * torch.from_numpy(wl)).sum().backward()
    for k in leaves:
        np.testing.assert_allclose(digest(p[k].grad.numpy()), g["rainbow_grad_" + k], rtol=1e-4, atol=1e-5, err_msg=k)
    with torch.no_grad():
        prob_eval, _ = N.rainbow_head(p, N.nature_conv_body_noisy(p, xn, training=False), 4, 51, training=False)
    np.testing.assert_allclose(prob_eval.numpy(), g["rainbow_prob_eval"], rtol=1e-5, atol=1e-7)


class _LinearSchedule:
    """deep_rl/utils/schedule.py:16-31 (restated here so the oracle test does not lean on the product's copy)."""

    def __init__(self, start, end=None, steps=None):
        if end is None:
            end, steps = start, 1
        self.inc = (end - start) / float(steps)
        self.current, self.end = start, end
        self.bound = min if end > start else max

    def __call__(self, steps=1):
        val = self.current
        self.current = self.bound(self.current + self.inc * steps, self.end)
        return val


@pytest.mark.parametrize("tag", ["c51_per", "dqn_per"])
def test_per_agent_schedule_oracle_in_order_equals_reference_run(golden, tag):
    """oracle/async_schedule_oracle.py::AsyncPerAgentScheduleOracle (the prioritized-replay schedule oracle of round 4) driven
    IN ORDER -- actor(k) on the current parameters, report, draw, learn, target sync -- must reproduce the run of the
    reference's own CategoricalDQNAgent / DQNAgent + PrioritizedReplay (tests/golden/pixel_agents.npz, cases c51_per, dqn_per): every stored
    transition, the priority tree, max_priority, the parameters after every update and the positions of numpy's and python's
    generators.  That pins everything in the class except the one thing the async pipeline defines -- WHICH parameters the
    actor sees (one update staler) -- which is a two-line difference in the driver (tests/test_gpu_agents.py)."""
    import zlib
    import fake_envs
    from golden.make_golden_cases import PIXEL_AGENT_CASES, trajectory_digest
    from oracle.async_schedule_oracle import AsyncPerAgentScheduleOracle
    g = golden("pixel_agents")
    tag, kind, rep, n_step, done_period, steps = [c for c in PIXEL_AGENT_CASES if c[0] == tag][0]
    k = tag + "_"
    np_state, py_state = np.random.get_state(), random.getstate()
    try:
        np.random.seed(3)
        np.random.randint(int(1e6))         # random_seed(3), torch_utils.py:36-38: the torch seed is drawn from the numpy stream
        random.seed(3)
        hname, n_head = ("fc_categorical", 4 * 51) if kind == "c51" else ("fc_head", 4)
        shapes = fake_envs.NATURE_SHAPES + [(hname + ".weight", (n_head, 512)), (hname + ".bias", (n_head,))]
        p_np = fake_envs.numpy_params(shapes, 17)
        okw = dict(head="c51", clip=5.0, lr=0.00025, eps=0.01 / 32) if kind == "c51" else dict(head="vanilla", clip=5.0)
        sched = _LinearSchedule(1.0, 0.05, 60)
        actor_steps = [0]

        def epsilon():                      # DQN_agent.py:34-39
            eps = 1 if actor_steps[0] < 40 else sched()
            actor_steps[0] += 1
            return eps

        orc = AsyncPerAgentScheduleOracle(p_np, 500, 32, env_seed=7, done_period=done_period, actor_rs=np.random,
                                          epsilon_fn=epsilon, beta_fn=_LinearSchedule(0.4, 1.0, 1000), exploration_steps=40,
                                          target_freq=3, **okw)
        # state_dict order of the reference's CategoricalNet / VanillaNet: the head first, then the body
        order = [hname + ".weight", hname + ".bias"] + [n for n, _ in fake_envs.NATURE_SHAPES]
        traj = []
        for t in range(steps):
            orc.actor_step(orc.p)                                    # in order: the CURRENT parameters
            if orc.report():
                tree_idx, prob, data_idx, batch = orc.draw()
                orc.learn(tree_idx, prob, batch)
                traj.append(trajectory_digest({n: orc.p[n] for n in order}))
            orc.maybe_sync_target()
        want_steps = list(g[k + "update_steps"])
        assert len(traj) == len(want_steps) == steps - want_steps[0]
        rp = orc.rep
        n = rp.size()
        assert orc.total_steps == int(g[k + "total_steps"]) and [rp.pos, n] == list(g[k + "pos_size"])
        assert np.array_equal(rp.action[:n].reshape(-1), g[k + "replay_action"])
        assert np.array_equal(rp.reward[:n], g[k + "replay_reward"]) and np.array_equal(rp.mask[:n], g[k + "replay_mask"])
        crc = np.asarray([zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in rp.state[:n]], dtype=np.int64)
        assert np.array_equal(crc, g[k + "replay_frame_crc"])
        assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g[k + "np_rng_tail"])
        assert np.array_equal([random.getrandbits(30) for _ in range(2)], g[k + "py_rng_tail"])
        np.testing.assert_allclose(rp.tree.tree, g[k + "tree"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(float(rp.max_priority), float(g[k + "max_priority"]), rtol=2e-5)
        for i, (a, b) in enumerate(zip(traj, g[k + "update_digests"])):      # the parameters after EVERY update
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6, err_msg="update %d" % i)
    finally:
        np.random.set_state(np_state)
        random.setstate(py_state)


def test_image_lut_matches_reference_numerics():
    lut = NUM.image_lut()
    assert lut.dtype == np.float32 and lut.shape == (256,)
    # f64 multiply then round-to-f32 (sync path), NOT f32*f32 (async path): differs in 126 code points
    async_path = np.arange(256, dtype=np.float32) * np.float32(1.0 / 255)
    assert int((lut != async_path).sum()) == 126


def test_running_mean_std_two_pass():
    rs = np.random.RandomState(0)
    rms = NUM.RunningMeanStdOracle(shape=(1, 3))
    chunks = [rs.randn(5, 3) * 2 + 1 for _ in range(20)]
    for c in chunks:
        rms.update(c)
    allx = np.concatenate(chunks)
    # prior pseudo-count 1e-4 of (mean 0, var 1) is below 1e-5 relative here
    np.testing.assert_allclose(rms.mean[0], allx.mean(0), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rms.var[0], allx.var(0), rtol=1e-4, atol=1e-5)


def test_product_normalizers_equal_oracle():
    """Host-side normalisers of the product (deeprl_amd/normalizers.py, numpy fp64 like the reference's
    normalizer.py:28-71) against the oracle restatement on a random observation stream: MeanStdNormalizer
    (running statistics, clipping, read-only mode, state_dict round trip), RescaleNormalizer / ImageNormalizer on
    numpy input, SignNormalizer."""
    from deeprl_amd.normalizers import ImageNormalizer, MeanStdNormalizer, RescaleNormalizer, SignNormalizer
    rs = np.random.RandomState(7)
    prod, orc = MeanStdNormalizer(), NUM.MeanStdNormalizerOracle()
    for t in range(50):
        x = rs.standard_normal((16, 17)) * (1 + t % 5) + 0.3 * t
        assert np.array_equal(prod(x), orc(x))
    prod.set_read_only()
    orc.read_only = True
    x = rs.standard_normal((16, 17)) * 100
    assert np.array_equal(prod(x), orc(x)) and np.abs(prod(x)).max() <= 10.0
    clone = MeanStdNormalizer(read_only=True)
    clone(x)                                    # creates its statistics object, read-only: no update
    clone.load_state_dict(prod.state_dict())
    assert np.array_equal(clone(x), prod(x))
    img = rs.randint(0, 256, size=(2, 4, 84, 84)).astype(np.uint8)
    assert np.array_equal(np.asarray(ImageNormalizer()(img), dtype=np.float32), NUM.image_normalize_sync(img))
    assert np.array_equal(RescaleNormalizer(0.5)(img), 0.5 * img)
    r = rs.standard_normal(9)
    assert np.array_equal(SignNormalizer()(r), np.sign(r))


@pytest.mark.skipif(not os.path.isdir(os.environ.get("DEEPRL_REFERENCE_ROOT", "/root/reference")),
                    reason="the reference tree is only present in the authoring container")
def test_committed_fixtures_are_the_reference_outputs(tmp_path):
    """Pins tests/golden/*.npz to the reference LIVE: re-runs tests/golden/make_golden.py (which drives the
    untouched /root/reference code through tests/ref_shim.py) into a scratch directory and compares every array
    with the committed fixture bit for bit.  Skipped on the GPU box, where /root/reference does not exist."""
    import glob
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, GOLDEN_OUT=str(tmp_path))
    subprocess.check_call([sys.executable, os.path.join(here, "golden", "make_golden.py")], env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    committed = sorted(glob.glob(os.path.join(here, "golden", "*.npz")))
    assert len(committed) >= 12
    for f in committed:
        a = np.load(f, allow_pickle=True)
        b = np.load(os.path.join(str(tmp_path), os.path.basename(f)), allow_pickle=True)
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            if a[k].dtype == object:
                assert str(a[k]) == str(b[k]), (f, k)
            else:
                assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (f, k)


@pytest.mark.skipif(not os.path.isdir(os.environ.get("DEEPRL_REFERENCE_ROOT", "/root/reference")),
                    reason="the reference tree is only present in the authoring container")
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_replay_oracle_equals_live_reference_on_random_streams(seed):
    """Beyond the fixed fixtures: random feed / sample / update_priorities interleavings (random capacity,
    history, n-step, discount; wrap-around; terminals) through the reference's own UniformReplay /
    PrioritizedReplay (tests/ref_shim.py) and through the oracle -- identical samples, index streams, sampling
    probabilities, tree contents and RNG positions."""
    import ref_shim
    ref = ref_shim.load()
    rs = np.random.RandomState(100 + seed)
    cap = int(rs.randint(40, 200))
    h, n = int(rs.randint(1, 5)), int(rs.randint(1, 4))
    gamma = float(rs.choice([1.0, 0.99, 0.5]))
    b = 8
    for cls_ref, cls_orc, per in ((ref.UniformReplay, UniformReplayOracle, False), (ref.PrioritizedReplay, PrioritizedReplayOracle, True)):
        r = cls_ref(memory_size=cap, batch_size=b, n_step=n, discount=gamma, history_length=h)
        o = cls_orc(cap, b, n, gamma, h)
        np.random.seed(seed)
        random.seed(seed)
        st_np, st_py = np.random.get_state(), random.getstate()
        fed = 0
        for step in range(3 * cap):
            frame = rs.randint(0, 256, size=(3, 3)).astype(np.uint8)
            act, rew, msk = int(rs.randint(0, 4)), float(rs.randint(-1, 2)), int(rs.rand() > 0.2)
            r.feed(dict(state=[frame], action=[act], reward=[rew], mask=[msk]))
            o.feed_one(frame, np.int64(act), rew, msk)
            fed += 1
            if fed > h + n + 4 and step % 7 == 0:
                # same RNG state into both samplers
                np.random.set_state(st_np); random.setstate(st_py)
                got = r.sample()
                a_np, a_py = np.random.get_state(), random.getstate()
                np.random.set_state(st_np); random.setstate(st_py)
                want = o.sample()
                b_np, b_py = np.random.get_state(), random.getstate()
                assert a_py == b_py and all(np.array_equal(x, y) for x, y in zip(a_np[1:3], b_np[1:3]))
                st_np, st_py = a_np, a_py
                assert np.array_equal(np.asarray(got.state), want[0]) and np.array_equal(np.asarray(got.next_state), want[3])
                assert np.array_equal(np.asarray(got.action).reshape(-1), np.asarray(want[1]).reshape(-1))
                assert np.array_equal(np.asarray(got.reward, dtype=np.float64), want[2])
                assert np.array_equal(np.asarray(got.mask), want[4])
                if per:
                    assert np.array_equal(np.asarray(got.sampling_prob), want[5]) and np.array_equal(np.asarray(got.idx), want[6])
                    prio = (rs.rand(b).astype(np.float32) * 2 + 0.1)
                    info = list(zip(np.asarray(got.idx).tolist(), prio.tolist()))
                    r.update_priorities(info)
                    o.update_priorities(info)
                    assert np.array_equal(r.tree.tree, o.tree.tree) and r.max_priority == o.max_priority


def test_atari_preprocess_oracle_vs_float64_area_average():
    """oracle/preproc_oracle.py restates OpenCV's RGB2GRAY + INTER_AREA resize (parity unpinned by the reference: cv2 and
    baselines are not installed).  Checked against an INDEPENDENT formulation: the exact area average of the luminance image
    over each destination cell in float64 (overlap-weighted box filter) -- the fp32 table walk may differ from it by
    rounding only: at most one grey level, and only where the exact average sits on a .5 boundary to ~1e-4."""
    from oracle import preproc_oracle as P
    rs = np.random.RandomState(0)
    raw = rs.randint(0, 256, size=(2, 2, 210, 160, 3)).astype(np.uint8)
    raw[1, :, 50:90] = 255                       # a saturated band and a black band: exact values survive the average
    raw[1, :, 120:160] = 0
    got = P.atari_preprocess(raw)
    assert got.shape == (2, 84, 84) and got.dtype == np.uint8
    mx = np.maximum(raw[:, 0], raw[:, 1]).astype(np.int64)
    gray = ((mx[..., 0] * 4899 + mx[..., 1] * 9617 + mx[..., 2] * 1868 + 8192) >> 14).astype(np.float64)

    def weights(ssize, dsize):
        w = np.zeros((dsize, ssize))
        scale = ssize / dsize
        for d in range(dsize):
            lo, hi = d * scale, (d + 1) * scale
            for s in range(int(np.floor(lo)), min(ssize, int(np.ceil(hi)))):
                w[d, s] = max(0.0, min(hi, s + 1) - max(lo, s))
            w[d] /= w[d].sum()
        return w
    wy, wx = weights(210, 84), weights(160, 84)
    exact = np.einsum("ys,nst,xt->nyx", wy, gray, wx)
    diff = np.abs(got.astype(np.float64) - exact)
    assert diff.max() <= 0.5 + 1e-3, diff.max()
    assert (got[1, 22:34] == 255).all() and (got[1, 50:62] == 0).all()     # rows fully inside the saturated / black bands
    # the table itself: weights of every destination index sum to 1 (fp32) and cover [d*scale, (d+1)*scale)
    for ssize in (210, 160):
        for row in P.resize_area_tab(ssize, 84):
            assert abs(sum(float(a) for _, a in row) - 1.0) < 1e-6
