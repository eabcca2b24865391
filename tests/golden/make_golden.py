#!/usr/bin/env python
"""Generates tests/golden/*.npz by running the REFERENCE's own code (imported
read-only from /root/reference through tests/ref_shim.py) on seeded inputs.

The reference ships no tests or golden vectors, so these
fixtures are what pins the oracle (oracle/) and, through it, the HIP path.
Re-run:  python tests/golden/make_golden.py        (needs /root/reference)
"""
import os
import random
import zlib
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root: deeprl_amd.envs' numpy host emulators
import ref_shim  # noqa: E402
import fake_envs  # noqa: E402

ref = ref_shim.load()
torch.set_num_threads(1)


def save(name, **arrays):
    path = os.path.join(os.environ.get("GOLDEN_OUT", HERE), name + ".npz")   # GOLDEN_OUT: regenerate elsewhere (CI check)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


# --------------------------------------------------------------------------- replay
from golden.make_golden_cases import UNIFORM_CASES, PER_CASES, stream as _stream, trajectory_digest  # noqa: E402


def gen_uniform():
    out = {}
    for name, mem, b, h, n, disc, shape, kind, t_len, cps in UNIFORM_CASES:
        rs = np.random.RandomState(1000 + ord(name))
        states, actions, rewards, masks = _stream(rs, t_len, shape, kind, 4, 0.1)
        replay = ref.UniformReplay(memory_size=mem, batch_size=b, n_step=n, discount=disc, history_length=h)
        accepted = []
        orig = replay.construct_transition

        def logged(index, _orig=orig, _acc=accepted):
            tr = _orig(index)
            if tr is not None:
                _acc.append(index)
            return tr

        replay.construct_transition = logged
        np.random.seed(2000 + ord(name))
        for t in range(t_len):
            replay.feed(dict(state=states[t][None], action=actions[t:t + 1], reward=[rewards[t]],
                             mask=masks[t:t + 1]))
            if t in cps:
                del accepted[:]
                tr = replay.sample()
                k = "%s_t%d_" % (name, t)
                out[k + "idx"] = np.asarray(accepted, dtype=np.int64)
                out[k + "state"], out[k + "action"] = tr.state, tr.action
                out[k + "reward"], out[k + "next_state"], out[k + "mask"] = tr.reward, tr.next_state, tr.mask
                out[k + "pos_size"] = np.asarray([replay.pos, replay.size()], dtype=np.int64)
        out[name + "_rng_tail"] = np.random.randint(0, 1 << 30, size=4)  # RNG stream position check
    save("uniform_replay", **out)


def gen_prioritized():
    out = {}
    for name, mem, b, h, n, disc, shape, kind, t_len, every in PER_CASES:
        rs = np.random.RandomState(3000 + ord(name))
        states, actions, rewards, masks = _stream(rs, t_len, shape, kind, 4, 0.1)
        replay = ref.PrioritizedReplay(memory_size=mem, batch_size=b, n_step=n, discount=disc, history_length=h)
        random.seed(4000 + ord(name))
        np.random.seed(4000 + ord(name))
        k_samples = 0
        for t in range(t_len):
            replay.feed(dict(state=states[t][None], action=actions[t:t + 1], reward=[rewards[t]],
                             mask=masks[t:t + 1]))
            if t >= h + n + 6 and t % every == 0:
                tr = replay.sample()
                k = "%s_s%d_" % (name, k_samples)
                out[k + "t"] = np.asarray(t)
                for f in ("state", "action", "reward", "next_state", "mask", "sampling_prob", "idx"):
                    out[k + f] = getattr(tr, f)
                # agent side (DQN_agent.py:121-123): fp32 priorities, zip(idx, prio)
                loss = torch.from_numpy(rs.standard_normal(b).astype(np.float32)) * 2
                prio = loss.abs().add(0.01).pow(0.5)
                idxs = ref.tensor(tr.idx).long()
                out[k + "prio"] = ref.to_np(prio)
                replay.update_priorities(zip(ref.to_np(idxs), ref.to_np(prio)))
                out[k + "tree"] = replay.tree.tree.copy()
                out[k + "max_priority"] = np.asarray(float(replay.max_priority))
                k_samples += 1
        out[name + "_n_samples"] = np.asarray(k_samples)
        out[name + "_tree_final"] = replay.tree.tree.copy()
        out[name + "_rng_tail"] = np.asarray([random.random() for _ in range(3)])
    save("prioritized_replay", **out)


def gen_sumtree():
    """Raw SumTree ops incl. the pending-idx gating and duplicate handling."""
    out = {}
    for cap in (8, 13, 50):
        rs = np.random.RandomState(cap)
        tree = ref.SumTree(cap)
        log = []
        for step in range(4 * cap):
            op = rs.randint(0, 3)
            if op == 0 or tree.n_entries < 2:
                p = np.float32(rs.uniform(0.1, 4.0))
                tree.add(p, None)
                log.append((0, float(p), 0.0, -1))
            elif op == 1:
                s = rs.uniform(0, tree.total())
                idx, p, didx = tree.get(s)
                log.append((1, float(s), float(p), idx))
            else:
                idx = int(rs.randint(cap - 1, 2 * cap - 1))
                p = np.float32(rs.uniform(0.1, 4.0))
                tree.update(idx, p)
                log.append((2, float(p), 0.0, idx))
        out["cap%d_log" % cap] = np.asarray(log, dtype=np.float64)
        out["cap%d_tree" % cap] = tree.tree.copy()
        out["cap%d_pending" % cap] = np.asarray(sorted(tree.pending_idx), dtype=np.int64)
    save("sumtree", **out)


# --------------------------------------------------------------------------- losses
class _Obj:
    pass


class _FakeNet:
    """Returns fixed (leaf) tensors per call, in order -- lets the reference's
    compute_loss run on hand-made network outputs so grads w.r.t. them are golden."""

    def __init__(self, outs):
        self.outs = list(outs)
        self.k = 0

    def __call__(self, x):
        o = self.outs[self.k % len(self.outs)]
        self.k += 1
        return o


def _cfg(**kw):
    c = ref.Config()
    c.state_normalizer = ref.RescaleNormalizer()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _transitions(rs, b, a, per=False):
    state = rs.standard_normal((b, 3)).astype(np.float32)
    tr = dict(state=state, action=rs.randint(0, a, size=b).astype(np.int64),
              reward=np.sign(rs.standard_normal(b)), next_state=state + 1,
              mask=(rs.rand(b) > 0.2).astype(np.int32))
    if per:
        p = rs.uniform(0.001, 0.2, size=b)
        return ref.PrioritizedTransition(sampling_prob=p, idx=rs.randint(40, 90, size=b).astype(np.int64), **tr)
    return ref.Transition(**tr)


def gen_dqn_loss():
    out = {}
    for tag, b, a, n_step, double_q in (("b32a4", 32, 4, 1, False), ("b10a2n3", 10, 2, 3, False),
                                         ("b32a4dq", 32, 4, 1, True), ("b7a18", 7, 18, 1, True)):
        rs = np.random.RandomState(zlib.crc32(tag.encode()) % 10000)
        tr = _transitions(rs, b, a, per=True)
        q = torch.tensor(rs.standard_normal((b, a)).astype(np.float32), requires_grad=True)
        q_next_t = torch.tensor(rs.standard_normal((b, a)).astype(np.float32))
        q_next_o = torch.tensor(rs.standard_normal((b, a)).astype(np.float32))
        agent = _Obj()
        agent.config = _cfg(discount=0.99, n_step=n_step, double_q=double_q)
        agent.target_network = _FakeNet([dict(q=q_next_t)])
        # online net is called for next_states first (double_q) then for states
        agent.network = _FakeNet([dict(q=q_next_o), dict(q=q)] if double_q else [dict(q=q)])
        loss_vec = ref.DQNAgent.compute_loss(agent, tr)
        # PER branch exactly as DQN_agent.py:120-127
        eps, alpha, beta = 0.01, 0.5, 0.6
        prio = loss_vec.abs().add(eps).pow(alpha)
        sp = ref.tensor(tr.sampling_prob)
        w = sp.mul(sp.size(0)).add(1e-6).pow(-beta)
        w = w / w.max()
        loss_plain = ref.DQNAgent.reduce_loss(agent, loss_vec)
        loss_per = ref.DQNAgent.reduce_loss(agent, loss_vec.mul(w))
        g_plain, = torch.autograd.grad(loss_plain, q, retain_graph=True)
        g_per, = torch.autograd.grad(loss_per, q)
        k = tag + "_"
        out.update({k + "q": q.detach().numpy(), k + "q_next_t": q_next_t.numpy(), k + "q_next_o": q_next_o.numpy(),
                    k + "action": tr.action, k + "reward": tr.reward, k + "mask": tr.mask,
                    k + "sampling_prob": tr.sampling_prob, k + "cfg": np.asarray([0.99, n_step, double_q, eps, alpha, beta]),
                    k + "loss_vec": loss_vec.detach().numpy(), k + "loss": loss_plain.detach().numpy(),
                    k + "grad_q": g_plain.numpy(), k + "prio": prio.detach().numpy(), k + "w": w.numpy(),
                    k + "loss_per": loss_per.detach().numpy(), k + "grad_q_per": g_per.numpy()})
    save("dqn_loss", **out)


def gen_c51_loss():
    out = {}
    for tag, b, a, n_atoms, n_step, double_q in (("b32a4", 32, 4, 51, 1, False), ("b8a3n3dq", 8, 3, 51, 3, True),
                                                 ("b5a6at21", 5, 6, 21, 1, False)):
        rs = np.random.RandomState(zlib.crc32(tag.encode()) % 10000)
        tr = _transitions(rs, b, a)
        logits = torch.tensor(rs.standard_normal((b, a, n_atoms)).astype(np.float32), requires_grad=True)
        lt = torch.tensor(rs.standard_normal((b, a, n_atoms)).astype(np.float32) * 2)
        lo = torch.tensor(rs.standard_normal((b, a, n_atoms)).astype(np.float32) * 2)
        import torch.nn.functional as F
        mk = lambda z: dict(prob=F.softmax(z, dim=-1), log_prob=F.log_softmax(z, dim=-1))
        agent = _Obj()
        vmin, vmax = -10.0, 10.0
        agent.config = _cfg(discount=0.99, n_step=n_step, double_q=double_q, categorical_v_min=vmin,
                            categorical_v_max=vmax, categorical_n_atoms=n_atoms)
        agent.atoms = ref.tensor(np.linspace(vmin, vmax, n_atoms))
        agent.delta_atom = (vmax - vmin) / float(n_atoms - 1)
        agent.batch_indices = ref.range_tensor(b)
        agent.target_network = _FakeNet([mk(lt)])
        agent.network = _FakeNet([mk(lo), mk(logits)] if double_q else [mk(logits)])
        kl = ref.CategoricalDQNAgent.compute_loss(agent, tr)
        loss = ref.CategoricalDQNAgent.reduce_loss(agent, kl)
        g, = torch.autograd.grad(loss, logits)
        k = tag + "_"
        out.update({k + "logits": logits.detach().numpy(), k + "logits_next_t": lt.numpy(), k + "logits_next_o": lo.numpy(),
                    k + "action": tr.action, k + "reward": tr.reward, k + "mask": tr.mask,
                    k + "cfg": np.asarray([0.99, n_step, double_q, vmin, vmax, n_atoms]),
                    k + "kl": kl.detach().numpy(), k + "loss": loss.detach().numpy(), k + "grad_logits": g.numpy()})
    save("c51_loss", **out)


def gen_qr_loss():
    out = {}
    for tag, b, a, nq, n_step in (("b32a4", 32, 4, 200, 1), ("b6a3q17n3", 6, 3, 17, 3)):
        rs = np.random.RandomState(zlib.crc32(tag.encode()) % 10000)
        tr = _transitions(rs, b, a)
        theta = torch.tensor(rs.standard_normal((b, a, nq)).astype(np.float32), requires_grad=True)
        theta_t = torch.tensor(rs.standard_normal((b, a, nq)).astype(np.float32) * 1.5)
        agent = _Obj()
        agent.config = _cfg(discount=0.99, n_step=n_step, num_quantiles=nq)
        agent.batch_indices = ref.range_tensor(b)
        agent.cumulative_density = ref.tensor((2 * np.arange(nq) + 1) / (2.0 * nq)).view(1, -1)
        agent.target_network = _FakeNet([dict(quantile=theta_t)])
        agent.network = _FakeNet([dict(quantile=theta)])
        lv = ref.QuantileRegressionDQNAgent.compute_loss(agent, tr)
        loss = ref.QuantileRegressionDQNAgent.reduce_loss(agent, lv)
        g, = torch.autograd.grad(loss, theta)
        k = tag + "_"
        out.update({k + "theta": theta.detach().numpy(), k + "theta_next_t": theta_t.numpy(),
                    k + "action": tr.action, k + "reward": tr.reward, k + "mask": tr.mask,
                    k + "cfg": np.asarray([0.99, n_step, nq]),
                    k + "loss_vec": lv.detach().numpy(), k + "loss": loss.detach().numpy(), k + "grad_theta": g.numpy()})
    save("qr_loss", **out)


# --------------------------------------------------------------------------- full DQN update on NatureConv
def _load_numpy_params(module, shapes, seed):
    p = fake_envs.numpy_params(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    return p


def gen_dqn_nature_update():
    """One full reference DQN update (DQN_agent.py:114-134) on VanillaNet(NatureConvBody):
    forward q, loss, every gradient, clip, centered RMSprop (examples.py:67-68).
    Weights come from fake_envs.numpy_params so only seeds are stored."""
    out = {}
    b, a = 8, 4
    rs = np.random.RandomState(77)
    net = ref.VanillaNet(a, ref.NatureConvBody())
    tgt = ref.VanillaNet(a, ref.NatureConvBody())
    _load_numpy_params(net, fake_envs.nature_vanilla_shapes(a), seed=11)
    _load_numpy_params(tgt, fake_envs.nature_vanilla_shapes(a), seed=12)
    state = rs.randint(0, 256, size=(b, 4, 84, 84)).astype(np.uint8)
    next_state = rs.randint(0, 256, size=(b, 4, 84, 84)).astype(np.uint8)
    tr = ref.Transition(state=state, action=rs.randint(0, a, size=b).astype(np.int64),
                        reward=np.sign(rs.standard_normal(b)), next_state=next_state,
                        mask=(rs.rand(b) > 0.2).astype(np.int32))
    agent = _Obj()
    agent.config = _cfg(discount=0.99, n_step=1, double_q=False)
    agent.config.state_normalizer = ref.ImageNormalizer()
    agent.network, agent.target_network = net, tgt
    opt = torch.optim.RMSprop(net.parameters(), lr=0.00025, alpha=0.95, eps=0.01, centered=True)
    q0 = net(agent.config.state_normalizer(state))["q"].detach().numpy()
    traj = []
    for it in range(3):
        lv = ref.DQNAgent.compute_loss(agent, tr)
        loss = ref.DQNAgent.reduce_loss(agent, lv)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            grads = {k: p.grad.detach().numpy().copy() for k, p in net.named_parameters()}
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 5)
        opt.step()
        traj.append([float(loss.detach()), float(gn)])
    out["state_seed"] = np.asarray(77)
    out["q0"] = q0
    out["action"], out["reward"], out["mask"] = tr.action, tr.reward, tr.mask
    out["loss_gradnorm_traj"] = np.asarray(traj)
    for k, g in grads.items():
        if k == "body.fc4.weight":
            out["grad_" + k + "_rows"] = g[::37].copy()  # 14 of 512 rows, all 3136 columns
            out["grad_" + k + "_norm"] = np.asarray(np.sqrt((g.astype(np.float64) ** 2).sum()))
        else:
            out["grad_" + k] = g
    fin = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    for k, v in fin.items():
        if k == "body.fc4.weight":
            out["final_" + k + "_rows"] = v[::37].copy()
        else:
            out["final_" + k] = v
    out["q_final"] = net(agent.config.state_normalizer(state))["q"].detach().numpy()
    save("dqn_nature_update", **out)


# --------------------------------------------------------------------------- optimisers
def gen_optim():
    out = {}
    rs = np.random.RandomState(5)
    shapes = [(7, 5), (5,), (3, 2, 2, 2), (130,)]
    p0 = [rs.standard_normal(s).astype(np.float32) for s in shapes]
    grads = [[(rs.standard_normal(s) * (3.0 if i % 2 else 0.3)).astype(np.float32) for s in shapes] for i in range(5)]
    for j, p in enumerate(p0):
        out["p0_%d" % j] = p
    for i, gs in enumerate(grads):
        for j, g in enumerate(gs):
            out["g%d_%d" % (i, j)] = g
    specs = {
        "rmsprop_centered": lambda ps: torch.optim.RMSprop(ps, lr=0.00025, alpha=0.95, eps=0.01, centered=True),
        "rmsprop_plain": lambda ps: torch.optim.RMSprop(ps, lr=1e-4, alpha=0.99, eps=1e-5),
        "adam": lambda ps: torch.optim.Adam(ps, lr=2.5e-4, eps=0.01 / 32),
        "adam_default": lambda ps: torch.optim.Adam(ps, 3e-4),
    }
    for name, mk in specs.items():
        for clip in (5.0, 0.5):
            ps = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in p0]
            opt = mk(ps)
            norms = []
            for gs in grads:
                opt.zero_grad()
                for p, g in zip(ps, gs):
                    p.grad = torch.from_numpy(g.copy())
                norms.append(float(torch.nn.utils.clip_grad_norm_(ps, clip)))
                opt.step()
            for j, p in enumerate(ps):
                out["%s_clip%g_p%d" % (name, clip, j)] = p.detach().numpy()
            out["%s_clip%g_norms" % (name, clip)] = np.asarray(norms)
    save("optim", **out)


# --------------------------------------------------------------------------- on-policy: GAE, PPO, A2C
class _Logger:
    def info(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


def _capture_storage(module_name):
    mod = sys.modules[module_name]
    captured = []
    base = ref.Storage

    class CapturingStorage(base):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            captured.append(self)

        def extract(self, keys):
            e = super().extract(keys)
            self.entries = e
            return e

    mod.Storage = CapturingStorage
    return captured, lambda: setattr(mod, "Storage", base)


def _stack(lst, t_len):
    return np.stack([x.detach().numpy() for x in lst[:t_len]])


def gen_a2c():
    out = {}
    for tag, t_len, n_env, use_gae in (("t5n16", 5, 16, False), ("t5n16gae", 5, 16, True), ("t20n3gae", 20, 3, True)):
        captured, restore = _capture_storage("deep_rl.agent.A2C_agent")
        try:
            cfg = _cfg(discount=0.99, use_gae=use_gae, gae_tau=0.95, entropy_weight=0.01, rollout_length=t_len,
                       gradient_clip=5, num_workers=n_env, value_loss_weight=1.0)
            cfg.reward_normalizer = ref.RescaleNormalizer()
            agent = _Obj()
            agent.config = cfg
            agent.task = fake_envs.VectorTask(seed=5, state_dim=6, action_dim=3, horizon=7, num_envs=n_env)
            torch.manual_seed(3)
            agent.network = ref.CategoricalActorCriticNet(6, 3, ref.FCBody(6, hidden_units=(32,)))
            p_init = {k: v.detach().numpy().copy() for k, v in agent.network.state_dict().items()}
            agent.optimizer = torch.optim.RMSprop(agent.network.parameters(), lr=1e-3, alpha=0.99, eps=1e-5)
            agent.total_steps = 0
            agent.states = agent.task.reset()
            agent.record_online_return = lambda *a, **k: None
            seen_states = [np.asarray(agent.states).copy()]
            raw_step = agent.task.step

            def logging_step(actions, _raw=raw_step, _log=seen_states):
                o = _raw(actions)
                _log.append(np.asarray(o[0]).copy())
                return o

            agent.task.step = logging_step
            torch.manual_seed(33)  # action sampling stream
            ref.A2CAgent.step(agent)
            st = captured[0]
            k = tag + "_"
            out[k + "cfg"] = np.asarray([0.99, 0.95, use_gae, 0.01, 1.0, 5, t_len, n_env])
            out[k + "reward"], out[k + "mask"] = _stack(st.reward, t_len), _stack(st.mask, t_len)
            out[k + "v"] = _stack(st.v, t_len + 1)
            out[k + "log_pi_a"], out[k + "entropy"] = _stack(st.log_pi_a, t_len), _stack(st.entropy, t_len)
            out[k + "action"] = _stack(st.action, t_len)
            out[k + "adv"], out[k + "ret"] = _stack(st.advantage, t_len), _stack(st.ret, t_len)
            out[k + "states"] = np.stack(seen_states)  # [T+1, N, state_dim]
            for n, v in p_init.items():
                out[k + "init_" + n] = v
            for n, v in agent.network.state_dict().items():
                out[k + "final_" + n] = v.detach().numpy()
        finally:
            restore()
    save("a2c_step", **out)


def gen_ppo():
    out = {}
    for tag, t_len, n_env, shared in (("t64n2", 64, 2, False), ("t32n4", 32, 4, False)):
        captured, restore = _capture_storage("deep_rl.agent.PPO_agent")
        try:
            cfg = _cfg(discount=0.99, use_gae=True, gae_tau=0.95, entropy_weight=0.01, rollout_length=t_len,
                       gradient_clip=0.5, num_workers=n_env, optimization_epochs=3, mini_batch_size=32,
                       ppo_ratio_clip=0.2, target_kl=0.01, shared_repr=shared, max_steps=1e6)
            cfg.reward_normalizer = ref.RescaleNormalizer()
            cfg.state_normalizer = ref.RescaleNormalizer()
            agent = _Obj()
            agent.config = cfg
            agent.task = fake_envs.ContinuousTask(seed=9, state_dim=5, action_dim=2, horizon=25, num_envs=n_env)
            torch.manual_seed(4)
            agent.network = ref.GaussianActorCriticNet(
                5, 2, actor_body=ref.FCBody(5, hidden_units=(16, 16), gate=torch.tanh),
                critic_body=ref.FCBody(5, hidden_units=(16, 16), gate=torch.tanh))
            p_init = {k: v.detach().numpy().copy() for k, v in agent.network.state_dict().items()}
            agent.actor_opt = torch.optim.Adam(agent.network.actor_params, 3e-4)
            agent.critic_opt = torch.optim.Adam(agent.network.critic_params, 1e-3)
            agent.total_steps = 0
            agent.states = cfg.state_normalizer(agent.task.reset())
            agent.record_online_return = lambda *a, **k: None
            np.random.seed(21)
            torch.manual_seed(22)  # action sampling stream
            ref.PPOAgent.step(agent)
            st = captured[0]
            k = tag + "_"
            out[k + "cfg"] = np.asarray([0.99, 0.95, 0.01, 0.2, 0.01, 3, 32, t_len, n_env])
            out[k + "reward"], out[k + "mask"] = _stack(st.reward, t_len), _stack(st.mask, t_len)
            out[k + "v"] = _stack(st.v, t_len + 1)
            out[k + "adv"], out[k + "ret"] = _stack(st.advantage, t_len), _stack(st.ret, t_len)
            e = st.entries
            out[k + "ent_state"], out[k + "ent_action"] = e.state.detach().numpy(), e.action.detach().numpy()
            out[k + "ent_log_pi_a"], out[k + "ent_ret"] = e.log_pi_a.detach().numpy(), e.ret.detach().numpy()
            out[k + "ent_adv_normalized"] = e.advantage.detach().numpy()
            for n, v in p_init.items():
                out[k + "init_" + n] = v
            for n, v in agent.network.state_dict().items():
                out[k + "final_" + n] = v.detach().numpy()
        finally:
            restore()
    save("ppo_step", **out)


def gen_ppo_loss():
    """PPO / A2C loss values + grads on hand-made network outputs."""
    out = {}
    rs = np.random.RandomState(31)
    for tag, m in (("m64", 64), ("m256", 256), ("m5", 5)):
        lp = torch.tensor(rs.standard_normal((m, 1)).astype(np.float32) * 0.3 - 1, requires_grad=True)
        ent = torch.tensor(rs.uniform(0.5, 1.5, (m, 1)).astype(np.float32), requires_grad=True)
        v = torch.tensor(rs.standard_normal((m, 1)).astype(np.float32), requires_grad=True)
        old_lp = torch.tensor(lp.detach().numpy() + rs.standard_normal((m, 1)).astype(np.float32) * 0.3)
        adv = torch.tensor(rs.standard_normal((m, 1)).astype(np.float32))
        ret = torch.tensor(rs.standard_normal((m, 1)).astype(np.float32))
        clip, ew = 0.2, 0.01
        ratio = (lp - old_lp).exp()
        obj = ratio * adv
        obj_clipped = ratio.clamp(1.0 - clip, 1.0 + clip) * adv
        policy_loss = -torch.min(obj, obj_clipped).mean() - ew * ent.mean()
        value_loss = 0.5 * (ret - v).pow(2).mean()
        approx_kl = (old_lp - lp).mean()
        g = torch.autograd.grad(policy_loss + value_loss, [lp, ent, v])
        k = tag + "_"
        out.update({k + "lp": lp.detach().numpy(), k + "ent": ent.detach().numpy(), k + "v": v.detach().numpy(),
                    k + "old_lp": old_lp.numpy(), k + "adv": adv.numpy(), k + "ret": ret.numpy(),
                    k + "out": np.asarray([float(policy_loss), float(value_loss), float(approx_kl)]),
                    k + "g_lp": g[0].numpy(), k + "g_ent": g[1].numpy(), k + "g_v": g[2].numpy()})
    save("ppo_loss", **out)


# --------------------------------------------------------------------------- end-to-end DQN agent, config 1
def gen_dqn_agent_cartpole():
    """BASELINE config 1 (examples.py:11-52, dqn_feature) on the reference's own
    DQNAgent/DQNActor/ReplayWrapper classes, sync replay + sync actor, fake CartPole."""
    base_mod = sys.modules["deep_rl.agent.BaseAgent"]
    orig_logger = base_mod.get_logger
    base_mod.get_logger = lambda *a, **k: _Logger()
    try:
        out = {}
        for tag, replay_cls, n_step, n_steps in (("uniform", ref.UniformReplay, 1, 60),
                                                 ("per_n3", ref.PrioritizedReplay, 3, 60)):
            cfg = ref.Config()
            cfg.merge(dict(game="fake", n_step=n_step, replay_cls=replay_cls, async_replay=False, log_level=0, tag=tag))
            cfg.task_fn = lambda: fake_envs.VectorTask(seed=1, state_dim=4, action_dim=2, horizon=20)
            cfg.eval_env = cfg.task_fn()
            cfg.optimizer_fn = lambda params: torch.optim.RMSprop(params, 0.001)
            cfg.network_fn = lambda: ref.VanillaNet(cfg.action_dim, ref.FCBody(cfg.state_dim))
            cfg.history_length = 1
            cfg.batch_size = 10
            cfg.discount = 0.99
            cfg.max_steps = 1e5
            kw = dict(memory_size=int(1e4), batch_size=cfg.batch_size, n_step=cfg.n_step, discount=cfg.discount,
                      history_length=cfg.history_length)
            cfg.replay_fn = lambda: ref.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
            cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
            cfg.replay_beta = ref.LinearSchedule(0.4, 1.0, cfg.max_steps)
            cfg.random_action_prob = ref.LinearSchedule(1.0, 0.1, 100)
            cfg.target_network_update_freq = 5
            cfg.exploration_steps = 40
            cfg.double_q = False
            cfg.sgd_update_frequency = 4
            cfg.gradient_clip = 5
            cfg.async_actor = False
            ref.random_seed(0)
            random.seed(0)
            agent = ref.DQNAgent(cfg)
            init = {k: v.detach().numpy().copy() for k, v in agent.network.state_dict().items()}
            for _ in range(n_steps):
                agent.step()
            k = tag + "_"
            for n, v in init.items():
                out[k + "init_" + n] = v
            for n, v in agent.network.state_dict().items():
                out[k + "final_" + n] = v.detach().numpy()
            for n, v in agent.target_network.state_dict().items():
                out[k + "target_" + n] = v.detach().numpy()
            out[k + "total_steps"] = np.asarray(agent.total_steps)
            rp = agent.replay.replay
            out[k + "replay_action"] = np.asarray(rp.action).reshape(-1)
            out[k + "replay_state"] = np.asarray(rp.state)
            out[k + "rng_tail"] = np.random.randint(0, 1 << 30, size=4)
    finally:
        base_mod.get_logger = orig_logger
    save("dqn_agent_cartpole", **out)


def _quiet_logger():
    base_mod = sys.modules["deep_rl.agent.BaseAgent"]
    orig = base_mod.get_logger
    base_mod.get_logger = lambda *a, **k: _Logger()
    return lambda: setattr(base_mod, "get_logger", orig)


def _dump_agent(out, k, agent, init):
    for n, v in init.items():
        out[k + "init_" + n] = v
    for n, v in agent.network.state_dict().items():
        out[k + "final_" + n] = v.detach().numpy()
    for n, v in agent.target_network.state_dict().items():
        out[k + "target_" + n] = v.detach().numpy()
    out[k + "total_steps"] = np.asarray(agent.total_steps)
    out[k + "rng_tail"] = np.random.randint(0, 1 << 30, size=4)


def gen_ddpg_td3():
    """DDPGAgent.step / TD3Agent.step (DDPG_agent.py:38-100, TD3_agent.py:38-108) for 40 steps on a continuous fake
    task: warm-up with action_space.sample(), exploration noise from np.random, uniform replay of f64 vectors, two Adam
    optimisers, soft target updates.  TD3's target-policy noise is torch's global generator in the reference (not
    reproducible across devices): td3_noise = 0 makes it the zero tensor."""
    out = {}
    restore = _quiet_logger()
    try:
        for tag in ("ddpg", "td3"):
            cfg = ref.Config()
            cfg.merge(dict(game="fake", log_level=0, tag=tag))
            cfg.task_fn = lambda: fake_envs.ContinuousTask(seed=13, state_dim=5, action_dim=2, horizon=9)
            cfg.eval_env = cfg.task_fn()
            if tag == "ddpg":
                cfg.network_fn = lambda: ref.DeterministicActorCriticNet(
                    5, 2, actor_body=ref.FCBody(5, (16, 16), gate=torch.relu), critic_body=ref.FCBody(7, (16, 16), gate=torch.relu),
                    actor_opt_fn=lambda p: torch.optim.Adam(p, lr=1e-3), critic_opt_fn=lambda p: torch.optim.Adam(p, lr=1e-3))
                cfg.replay_fn = lambda: ref.UniformReplay(memory_size=200, batch_size=8)
                cfg.random_process_fn = lambda: ref.OrnsteinUhlenbeckProcess(size=(2,), std=ref.LinearSchedule(0.2))
                cls = ref.DDPGAgent
            else:
                cfg.network_fn = lambda: ref.TD3Net(
                    2, actor_body_fn=lambda: ref.FCBody(5, (16, 16), gate=torch.relu),
                    critic_body_fn=lambda: ref.FCBody(7, (16, 16), gate=torch.relu),
                    actor_opt_fn=lambda p: torch.optim.Adam(p, lr=1e-3), critic_opt_fn=lambda p: torch.optim.Adam(p, lr=1e-3))
                cfg.replay_fn = lambda: ref.ReplayWrapper(ref.UniformReplay, dict(memory_size=200, batch_size=8), False)
                cfg.random_process_fn = lambda: ref.GaussianProcess(size=(2,), std=ref.LinearSchedule(0.1))
                cfg.td3_noise, cfg.td3_noise_clip, cfg.td3_delay = 0.0, 0.5, 2
                cls = ref.TD3Agent
            cfg.discount, cfg.warm_up, cfg.target_network_mix, cfg.max_steps = 0.99, 10, 5e-3, 1e5
            torch.manual_seed(7)
            np.random.seed(17)
            random.seed(17)
            agent = cls(cfg)
            init = {k: v.detach().numpy().copy() for k, v in agent.network.state_dict().items()}
            for _ in range(40):
                agent.step()
            k = tag + "_"
            _dump_agent(out, k, agent, init)
            rp = getattr(agent.replay, "replay", agent.replay)
            out[k + "replay_action"] = np.asarray(rp.action[:rp.size()], dtype=np.float64).reshape(rp.size(), -1)
            out[k + "replay_reward"] = np.asarray(rp.reward[:rp.size()], dtype=np.float64).reshape(-1)
    finally:
        restore()
    save("ddpg_td3_agents", **out)


def gen_option_critic():
    """OptionCriticAgent.step (OptionCritic_agent.py:52-119) for 4 rollouts of 5 steps x 3 workers.  Every
    Categorical.sample() the reference draws (options, options_hat, actions; torch's CPU generator) is recorded so that an
    implementation on another device can replay the same decisions."""
    out = {}
    restore = _quiet_logger()
    samples = []
    orig_sample = torch.distributions.Categorical.sample

    def recording_sample(self, sample_shape=torch.Size()):
        v = orig_sample(self, sample_shape)
        samples.append(v.detach().numpy().copy())
        return v

    torch.distributions.Categorical.sample = recording_sample
    try:
        cfg = ref.Config()
        cfg.merge(dict(game="fake", log_level=0, tag="oc"))
        cfg.num_workers = 3
        cfg.task_fn = lambda: fake_envs.VectorTask(seed=5, state_dim=4, action_dim=2, horizon=7, num_envs=3)
        cfg.eval_env = fake_envs.VectorTask(seed=6, state_dim=4, action_dim=2)
        cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, 0.001)
        cfg.network_fn = lambda: ref.OptionCriticNet(ref.FCBody(4, hidden_units=(16,)), 2, num_options=2)
        cfg.random_option_prob = ref.LinearSchedule(1.0, 0.1, 100)
        cfg.discount, cfg.target_network_update_freq, cfg.rollout_length = 0.99, 4, 5
        cfg.termination_regularizer, cfg.entropy_weight, cfg.gradient_clip = 0.01, 0.01, 5
        torch.manual_seed(9)
        np.random.seed(19)
        agent = ref.OptionCriticAgent(cfg)
        init = {k: v.detach().numpy().copy() for k, v in agent.network.state_dict().items()}
        for _ in range(4):
            agent.step()
        _dump_agent(out, "oc_", agent, init)
        out["oc_samples"] = np.stack(samples).astype(np.int64)      # [4 rollouts * 5 steps * 3 draws, 3 workers]
    finally:
        torch.distributions.Categorical.sample = orig_sample
        restore()
    save("option_critic_agent", **out)


# --------------------------------------------------------------------------- pixel DQN family, agent level (configs 2 / 4)
class _RefPixelTask:
    """The synthetic Atari stream (counter-hash frames; deeprl_amd.envs.SyntheticAtari is a pure numpy host emulator)
    behind the reference's Task surface (envs.py:153-196) with ONE environment and DummyVecEnv's auto-reset
    (envs.py:140-150); observations are the REFERENCE's LazyFrames, so its replay feed stores s[-1] exactly as with the
    real FrameStack wrapper."""

    def __init__(self, seed, done_period, n_actions=4):
        from deeprl_amd.envs import SyntheticAtari
        self.env = SyntheticAtari(seed=seed, history=4, n_actions=n_actions, done_period=done_period)
        self.state_dim, self.action_dim, self.name = (4, 84, 84), n_actions, "synthetic-atari"
        self.action_space = type("Discrete", (), {"n": n_actions})()

    @staticmethod
    def _obs(o):
        return ref.LazyFrames(list(o._frames))

    def reset(self):
        return [self._obs(self.env.reset())]

    def step(self, actions):
        obs, rew, done, info = self.env.step(int(np.asarray(actions).reshape(-1)[0]))
        if done:
            obs = self.env.reset()
        return [self._obs(obs)], np.asarray([rew]), np.asarray([done]), (info,)


from golden.make_golden_cases import PIXEL_AGENT_CASES, digest  # noqa: E402


def gen_pixel_agents():
    """BASELINE configs 2 / 4 at agent level on the reference's own DQNAgent / CategoricalDQNAgent /
    QuantileRegressionDQNAgent (NatureConvBody, batch 32, history 4, sync actor / replay; examples.py:55-97, 127-158,
    192-222) over the synthetic Atari stream: action stream, replay bookkeeping, priority tree, RNG positions and parameter
    digests after `steps` agent steps from seeded normal weights (tests/fake_envs.numpy_params)."""
    restore = _quiet_logger()
    out = {}
    try:
        for tag, kind, rep, n_step, done_period, steps in PIXEL_AGENT_CASES:
            cfg = ref.Config()
            replay_cls = ref.PrioritizedReplay if rep == "per" else ref.UniformReplay
            cfg.merge(dict(game="fake", n_step=n_step, replay_cls=replay_cls, async_replay=False, log_level=0, tag=tag))
            cfg.task_fn = lambda: _RefPixelTask(seed=7, done_period=done_period)
            cfg.eval_env = cfg.task_fn()
            if kind == "dqn":
                cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
                cfg.network_fn = lambda: ref.VanillaNet(cfg.action_dim, ref.NatureConvBody(in_channels=4))
                cls, head, n_head = ref.DQNAgent, "fc_head", 4
            elif kind == "c51":
                cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=0.00025, eps=0.01 / 32)
                cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
                cfg.network_fn = lambda: ref.CategoricalNet(cfg.action_dim, cfg.categorical_n_atoms, ref.NatureConvBody())
                cls, head, n_head = ref.CategoricalDQNAgent, "fc_categorical", 4 * 51
            else:
                cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=0.00005, eps=0.01 / 32)
                cfg.num_quantiles = 200
                cfg.network_fn = lambda: ref.QuantileNet(cfg.action_dim, cfg.num_quantiles, ref.NatureConvBody())
                cls, head, n_head = ref.QuantileRegressionDQNAgent, "fc_quantiles", 4 * 200
            cfg.random_action_prob = ref.LinearSchedule(1.0, 0.05, 60)
            cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
            kw = dict(memory_size=500, batch_size=32, n_step=n_step, discount=0.99, history_length=4)
            cfg.replay_fn = lambda: ref.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
            cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
            cfg.replay_beta = ref.LinearSchedule(0.4, 1.0, 1000)
            cfg.state_normalizer, cfg.reward_normalizer = ref.ImageNormalizer(), ref.SignNormalizer()
            cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 3, 40, 4
            cfg.gradient_clip, cfg.double_q, cfg.async_actor, cfg.max_steps = 5, False, False, 1e5
            ref.random_seed(3)
            random.seed(3)
            agent = cls(cfg)
            shapes = fake_envs.NATURE_SHAPES + [(head + ".weight", (n_head, 512)), (head + ".bias", (n_head,))]
            p_np = fake_envs.numpy_params(shapes, 17)
            agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
            agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
            traj_steps, traj = [], []
            prev = trajectory_digest(agent.network.state_dict())
            for t in range(steps):
                agent.step()
                cur = trajectory_digest(agent.network.state_dict())
                if not np.array_equal(cur, prev):        # this agent step ran an update
                    traj_steps.append(t)
                    traj.append(cur)
                prev = cur
            k = tag + "_"
            rp = agent.replay.replay
            n = rp.size()
            out[k + "update_steps"] = np.asarray(traj_steps, dtype=np.int64)
            out[k + "update_digests"] = np.stack(traj)      # [n_updates, ~420] float32: the parameters after every update
            out[k + "total_steps"] = np.asarray(agent.total_steps)
            out[k + "pos_size"] = np.asarray([rp.pos, n])
            out[k + "replay_action"] = np.asarray(rp.action[:n]).reshape(-1).astype(np.int64)
            out[k + "replay_reward"] = np.asarray(rp.reward[:n], dtype=np.float64).reshape(-1)
            out[k + "replay_mask"] = np.asarray(rp.mask[:n]).reshape(-1).astype(np.int32)
            out[k + "replay_frame_crc"] = np.asarray([zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in rp.state[:n]], dtype=np.int64)
            if rep == "per":
                out[k + "tree"] = np.asarray(rp.tree.tree, dtype=np.float64)
                out[k + "max_priority"] = np.asarray(float(rp.max_priority))
            for name, v in agent.network.state_dict().items():
                out[k + "final_" + name] = digest(v.detach().numpy())
            for name, v in agent.target_network.state_dict().items():
                out[k + "target_" + name] = digest(v.detach().numpy())
            out[k + "np_rng_tail"] = np.random.randint(0, 1 << 30, size=4)
            out[k + "py_rng_tail"] = np.asarray([random.getrandbits(30) for _ in range(2)], dtype=np.int64)
    finally:
        restore()
    save("pixel_agents", **out)


# --------------------------------------------------------------------------- on-policy agents on pixels (config 5 shapes)
AC_SHAPES = fake_envs.NATURE_SHAPES_PREFIXED("phi_body.") + [
    ("fc_action.weight", (4, 512)), ("fc_action.bias", (4,)), ("fc_critic.weight", (1, 512)), ("fc_critic.bias", (1,))]


def gen_pixel_onpolicy():
    """A2CAgent.step / PPOAgent.step of the reference on CategoricalActorCriticNet(NatureConvBody) (examples.py:361-381,
    525-550: BASELINE configs[4] shapes) over 4 synthetic Atari emulators: per-step log-probabilities / values of the
    rollout, actions, GAE outputs and parameter digests after the update(s).  States are regenerated by the test from the
    same emulators (tests/fake_envs.PixelVectorTask); initial weights from fake_envs.numpy_params."""
    out = {}
    p_np = fake_envs.numpy_params(AC_SHAPES, 23)
    # ---- A2C: rollout 5 x 4 envs, RMSprop
    captured, restore = _capture_storage("deep_rl.agent.A2C_agent")
    try:
        cfg = _cfg(discount=0.99, use_gae=True, gae_tau=1.0, entropy_weight=0.01, rollout_length=5, gradient_clip=5,
                   num_workers=4, value_loss_weight=1.0)
        cfg.state_normalizer, cfg.reward_normalizer = ref.ImageNormalizer(), ref.SignNormalizer()
        agent = _Obj()
        agent.config = cfg
        agent.task = fake_envs.PixelVectorTask(seed=5, num_envs=4, done_period=9)
        agent.network = ref.CategoricalActorCriticNet((4, 84, 84), 4, ref.NatureConvBody())
        agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        agent.optimizer = torch.optim.RMSprop(agent.network.parameters(), lr=1e-4, alpha=0.99, eps=1e-5)
        agent.total_steps = 0
        agent.states = agent.task.reset()
        agent.record_online_return = lambda *a, **k: None
        torch.manual_seed(33)
        ref.A2CAgent.step(agent)
        st = captured[0]
        k = "a2c_"
        out[k + "reward"], out[k + "mask"] = _stack(st.reward, 5), _stack(st.mask, 5)
        out[k + "v"] = _stack(st.v, 6)
        out[k + "log_pi_a"], out[k + "entropy"] = _stack(st.log_pi_a, 5), _stack(st.entropy, 5)
        out[k + "action"] = _stack(st.action, 5)
        out[k + "adv"], out[k + "ret"] = _stack(st.advantage, 5), _stack(st.ret, 5)
        for n, v in agent.network.state_dict().items():
            out[k + "final_" + n] = digest(v.detach().numpy())
    finally:
        restore()
    # ---- PPO: rollout 16 x 4 envs, 2 epochs of 4 minibatches, shared body, one Adam
    captured, restore = _capture_storage("deep_rl.agent.PPO_agent")
    try:
        cfg = _cfg(discount=0.99, use_gae=True, gae_tau=0.95, entropy_weight=0.01, rollout_length=16, gradient_clip=0.5,
                   num_workers=4, optimization_epochs=2, mini_batch_size=16, ppo_ratio_clip=0.1, target_kl=1e9,
                   shared_repr=True, max_steps=1e6)
        cfg.state_normalizer, cfg.reward_normalizer = ref.ImageNormalizer(), ref.SignNormalizer()
        agent = _Obj()
        agent.config = cfg
        agent.task = fake_envs.PixelVectorTask(seed=6, num_envs=4, done_period=11)
        agent.network = ref.CategoricalActorCriticNet((4, 84, 84), 4, ref.NatureConvBody())
        agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in p_np.items()})
        agent.opt = torch.optim.Adam(agent.network.parameters(), lr=2.5e-4)
        agent.total_steps = 0
        agent.states = cfg.state_normalizer(agent.task.reset())
        agent.record_online_return = lambda *a, **k: None
        agent.lr_scheduler = type("S", (), {"step": lambda self, *a: None})()
        np.random.seed(21)
        torch.manual_seed(22)
        ref.PPOAgent.step(agent)
        st = captured[0]
        k = "ppo_"
        out[k + "reward"], out[k + "mask"] = _stack(st.reward, 16), _stack(st.mask, 16)
        out[k + "v"] = _stack(st.v, 17)
        out[k + "adv"], out[k + "ret"] = _stack(st.advantage, 16), _stack(st.ret, 16)
        e = st.entries
        out[k + "ent_action"] = e.action.detach().numpy()
        out[k + "ent_log_pi_a"], out[k + "ent_ret"] = e.log_pi_a.detach().numpy(), e.ret.detach().numpy()
        out[k + "ent_adv_normalized"] = e.advantage.detach().numpy()
        for n, v in agent.network.state_dict().items():
            out[k + "final_" + n] = digest(v.detach().numpy())
    finally:
        restore()
    save("pixel_onpolicy", **out)


# --------------------------------------------------------------------------- Dueling / Rainbow heads + NoisyLinear (f4; config 4's
# rainbow_pixel network, examples.py:283-336)
def gen_rainbow_dueling():
    """DuelingNet / RainbowNet(NoisyLinear) of the reference (network_heads.py:24-37,57-86, network_utils.py:31-83) over
    NatureConvBody: (i) module level -- outputs and parameter gradients of a fixed linear functional on a seeded uint8
    batch, the noise vectors the reference's own reset_noise() draws from torch's CPU generator; (ii) agent level --
    CategoricalDQNAgent with rainbow_pixel's settings (noisy layers, PrioritizedReplay, double Q, Adam 6.25e-4 / 1.5e-4,
    clip 10) and DQNAgent over DuelingNet on the synthetic Atari stream: parameters after every update, replay
    bookkeeping, priority tree, RNG positions."""
    from golden.make_golden_cases import (DUELING_SHAPES, RAINBOW_SHAPES, NOISY_LAYERS, NOISE_BUFFERS, HEAD_AGENT_CASES,
                                          head_inputs)
    out = {}
    ref.Config.NOISY_LAYER_STD = 0.5
    x, wq, wl = head_inputs()
    xn = ref.ImageNormalizer()(x)
    # ---- module level: DuelingNet
    net = ref.DuelingNet(4, ref.NatureConvBody())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fake_envs.numpy_params(DUELING_SHAPES, 31).items()})
    q = net(xn)["q"]
    (q * torch.from_numpy(wq)).sum().backward()
    out["dueling_q"] = q.detach().numpy()
    for n, prm in net.named_parameters():
        out["dueling_grad_" + n] = digest(prm.grad.numpy())
    # ---- module level: RainbowNet with noisy layers
    torch.manual_seed(9)
    net = ref.RainbowNet(4, 51, ref.NatureConvBody(noisy_linear=True), noisy_linear=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fake_envs.numpy_params(RAINBOW_SHAPES, 33).items()}, strict=False)
    torch.manual_seed(11)
    net.reset_noise()                         # the reference's own draws (fc_value, fc_advantage, body.fc4: network_heads.py:72-76)
    sd = net.state_dict()
    for layer in NOISY_LAYERS:
        for b in NOISE_BUFFERS:
            out["rainbow_%s.%s" % (layer, b)] = sd["%s.%s" % (layer, b)].numpy().copy()
        out["rainbow_%s.weight_sigma0" % layer] = sd[layer + ".weight_sigma"].numpy().reshape(-1)[:1].copy()
        out["rainbow_%s.bias_sigma0" % layer] = sd[layer + ".bias_sigma"].numpy().reshape(-1)[:1].copy()
        out["rainbow_%s.weight_epsilon" % layer] = digest(sd[layer + ".weight_epsilon"].numpy())
    net.train()
    o = net(xn)
    (o["log_prob"] * torch.from_numpy(wl)).sum().backward()
    out["rainbow_prob"], out["rainbow_log_prob"] = o["prob"].detach().numpy(), o["log_prob"].detach().numpy()
    for n, prm in net.named_parameters():
        out["rainbow_grad_" + n] = digest(prm.grad.numpy())
    net.eval()
    with torch.no_grad():
        out["rainbow_prob_eval"] = net(xn)["prob"].numpy()
    # ---- agent level
    restore = _quiet_logger()
    try:
        for tag, steps in HEAD_AGENT_CASES:
            cfg = ref.Config()
            rainbow = tag == "rainbow"
            replay_cls = ref.PrioritizedReplay if rainbow else ref.UniformReplay
            cfg.merge(dict(game="fake", n_step=1, replay_cls=replay_cls, async_replay=False, log_level=0, tag=tag,
                           noisy_linear=rainbow))
            cfg.task_fn = lambda: _RefPixelTask(seed=7, done_period=8)
            cfg.eval_env = cfg.task_fn()
            if rainbow:
                cfg.optimizer_fn = lambda p: torch.optim.Adam(p, lr=0.000625, eps=1.5e-4)
                cfg.categorical_v_max, cfg.categorical_v_min, cfg.categorical_n_atoms = 10, -10, 51
                cfg.network_fn = lambda: ref.RainbowNet(cfg.action_dim, cfg.categorical_n_atoms,
                                                        ref.NatureConvBody(noisy_linear=True), noisy_linear=True)
                cls, shapes, seed = ref.CategoricalDQNAgent, RAINBOW_SHAPES, 35
            else:
                cfg.optimizer_fn = lambda p: torch.optim.RMSprop(p, lr=0.00025, alpha=0.95, eps=0.01, centered=True)
                cfg.network_fn = lambda: ref.DuelingNet(cfg.action_dim, ref.NatureConvBody())
                cls, shapes, seed = ref.DQNAgent, DUELING_SHAPES, 37
            cfg.random_action_prob = ref.LinearSchedule(1.0, 0.05, 60)
            cfg.batch_size, cfg.discount, cfg.history_length = 32, 0.99, 4
            kw = dict(memory_size=500, batch_size=32, n_step=1, discount=0.99, history_length=4)
            cfg.replay_fn = lambda: ref.ReplayWrapper(cfg.replay_cls, kw, cfg.async_replay)
            cfg.replay_eps, cfg.replay_alpha = 0.01, 0.5
            cfg.replay_beta = ref.LinearSchedule(0.4, 1.0, 1000)
            cfg.state_normalizer, cfg.reward_normalizer = ref.ImageNormalizer(), ref.SignNormalizer()
            cfg.target_network_update_freq, cfg.exploration_steps, cfg.sgd_update_frequency = 3, 40, 4
            cfg.gradient_clip, cfg.double_q, cfg.async_actor, cfg.max_steps = (10 if rainbow else 5), rainbow, False, 1e5
            ref.random_seed(3)
            random.seed(3)
            agent = cls(cfg)
            p_np = {k: torch.from_numpy(v) for k, v in fake_envs.numpy_params(shapes, seed).items()}
            agent.network.load_state_dict(p_np, strict=False)
            agent.target_network.load_state_dict(agent.network.state_dict())
            torch.manual_seed(5)               # from here on torch's CPU generator only feeds reset_noise()
            traj_steps, traj, actions = [], [], []
            prev = trajectory_digest(dict(agent.network.named_parameters()))
            for t in range(steps):
                agent.step()
                cur = trajectory_digest(dict(agent.network.named_parameters()))
                if not np.array_equal(cur, prev):
                    traj_steps.append(t)
                    traj.append(cur)
                prev = cur
            k = tag + "_"
            rp = agent.replay.replay
            n = rp.size()
            out[k + "update_steps"] = np.asarray(traj_steps, dtype=np.int64)
            out[k + "update_digests"] = np.stack(traj)
            out[k + "total_steps"] = np.asarray(agent.total_steps)
            out[k + "pos_size"] = np.asarray([rp.pos, n])
            out[k + "replay_action"] = np.asarray(rp.action[:n]).reshape(-1).astype(np.int64)
            out[k + "replay_reward"] = np.asarray(rp.reward[:n], dtype=np.float64).reshape(-1)
            out[k + "replay_mask"] = np.asarray(rp.mask[:n]).reshape(-1).astype(np.int32)
            if rainbow:
                out[k + "tree"] = np.asarray(rp.tree.tree, dtype=np.float64)
                out[k + "max_priority"] = np.asarray(float(rp.max_priority))
                out[k + "torch_rng_tail"] = torch.randint(0, 1 << 30, (4,)).numpy()
            for name, v in agent.network.named_parameters():
                out[k + "final_" + name] = digest(v.detach().numpy())
            out[k + "np_rng_tail"] = np.random.randint(0, 1 << 30, size=4)
            out[k + "py_rng_tail"] = np.asarray([random.getrandbits(30) for _ in range(2)], dtype=np.int64)
    finally:
        restore()
    save("rainbow_dueling", **out)


GENERATORS = [gen_uniform, gen_prioritized, gen_sumtree, gen_dqn_loss, gen_c51_loss, gen_qr_loss,
              gen_dqn_nature_update, gen_optim, gen_a2c, gen_ppo, gen_ppo_loss, gen_dqn_agent_cartpole, gen_ddpg_td3, gen_option_critic,
              gen_pixel_agents, gen_pixel_onpolicy, gen_rainbow_dueling]

if __name__ == "__main__":
    only = sys.argv[1:]
    for g in GENERATORS:
        if only and g.__name__ not in only:
            continue
        g()
