"""Host-side pieces of the hot path against the LIVE reference (tests/ref_shim.py; skipped where /root/reference is
absent): schedules, epsilon_greedy's RNG order (a4), random_sample's minibatch permutation (a22),
Storage feed / placeholder / extract (a7), Config defaults and merge, generate_tag."""
import os

import numpy as np
import pytest
import torch

REF = os.environ.get("DEEPRL_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the authoring container")


@pytest.fixture(scope="module")
def ref():
    import ref_shim
    return ref_shim.load()


def test_schedules(ref):
    import deeprl_amd as d
    for args in ((1.0, 0.1, 7), (0.4, 1.0, 5), (0.5, None, None), (1.0, 0.01, 1e6)):
        a, b = d.LinearSchedule(*args), ref.LinearSchedule(*args)
        for steps in (1, 1, 3, 1, 10, 1):
            assert a(steps) == b(steps)
    assert d.ConstantSchedule(0.3)(5) == ref.ConstantSchedule(0.3)(5)


def test_epsilon_greedy_consumes_the_same_random_numbers(ref):
    import deeprl_amd as d
    rs = np.random.RandomState(0)
    for shape in ((1, 4), (8, 6), (5,)):
        q = rs.standard_normal(shape)
        for eps in (0.0, 0.3, 1.0):
            np.random.seed(11)
            a = d.epsilon_greedy(eps, q)
            tail_a = np.random.randint(1 << 30)
            np.random.seed(11)
            b = ref.epsilon_greedy(eps, q)
            tail_b = np.random.randint(1 << 30)
            assert np.array_equal(np.asarray(a), np.asarray(b)) and tail_a == tail_b


def test_random_sample_batches(ref):
    import deeprl_amd as d
    for n, b in ((64, 16), (70, 16), (5, 8)):
        np.random.seed(n)
        x = [np.asarray(v).copy() for v in d.random_sample(np.arange(n), b)]
        np.random.seed(n)
        y = [np.asarray(v).copy() for v in ref.random_sample(np.arange(n), b)]
        assert len(x) == len(y) and all(np.array_equal(p, q) for p, q in zip(x, y))


def test_storage_feed_placeholder_extract(ref):
    import deeprl_amd as d
    t_len, n = 5, 3
    a, b = d.Storage(t_len), ref.Storage(t_len)
    rs = np.random.RandomState(3)
    for t in range(t_len + 1):
        row = {"v": torch.tensor(rs.standard_normal((n, 1)).astype(np.float32)),
               "log_pi_a": torch.tensor(rs.standard_normal((n, 1)).astype(np.float32))}
        a.feed(row)
        b.feed(row)
        if t < t_len:
            rm = {"reward": torch.tensor(rs.standard_normal((n, 1)).astype(np.float32)),
                  "mask": torch.tensor((rs.rand(n, 1) > 0.2).astype(np.float32))}
            a.feed(rm)
            b.feed(rm)
    a.placeholder()
    b.placeholder()
    assert a.advantage == b.advantage == [None] * t_len
    ea, eb = a.extract(["v", "log_pi_a", "reward", "mask"]), b.extract(["v", "log_pi_a", "reward", "mask"])
    assert ea._fields == eb._fields
    for x, y in zip(ea, eb):
        assert torch.equal(x, y) and x.shape == (t_len * n, 1)
    with pytest.raises(RuntimeError):
        a.feed({"no_such_key": 1})
    with pytest.raises(RuntimeError):
        b.feed({"no_such_key": 1})


def test_config_defaults_merge_and_tag(ref):
    import deeprl_amd as d
    a, b = d.Config(), ref.Config()
    for k, v in vars(b).items():
        if k.startswith("_Config__") or k == "parser":
            continue
        assert hasattr(a, k), "Config lacks attribute %s" % k
        va = getattr(a, k)
        if isinstance(v, (int, float, str, bool, type(None))):
            assert va == v, (k, va, v)
    for c in (a, b):
        c.merge(dict(game="X-v0", n_step=3, tag="t"))
    assert (a.game, a.n_step, a.tag) == (b.game, b.n_step, b.tag)
    pa, pb = dict(game="G", run=2, lr=0.1, fn=len), dict(game="G", run=2, lr=0.1, fn=len)
    d.generate_tag(pa)
    ref.generate_tag(pb)
    assert pa["tag"] == pb["tag"]


# ---------------------------------------------------------------------------------------------------------------
# loss oracles against the reference's own compute_loss on RANDOM shapes / seeds (the committed fixtures cover
# fixed cases).  The reference agents are driven as unbound methods on a stand-in, as tests/golden/make_golden.py does.
class _Obj:
    pass


class _FakeNet:
    def __init__(self, outs):
        self.outs, self.k = list(outs), 0

    def __call__(self, x):
        o = self.outs[self.k % len(self.outs)]
        self.k += 1
        return o


def _cfg(ref, **kw):
    c = ref.Config()
    c.state_normalizer = ref.RescaleNormalizer()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _tr(ref, rs, b, a):
    state = rs.standard_normal((b, 3)).astype(np.float32)
    return ref.Transition(state=state, action=rs.randint(0, a, size=b).astype(np.int64), reward=np.sign(rs.standard_normal(b)),
                          next_state=state + 1, mask=(rs.rand(b) > 0.3).astype(np.int32))


@pytest.mark.parametrize("seed", range(6))
def test_loss_oracles_equal_live_reference_on_random_cases(ref, seed):
    import torch.nn.functional as F
    from oracle import loss_oracle as L
    rs = np.random.RandomState(500 + seed)
    b, a = int(rs.randint(2, 40)), int(rs.randint(2, 19))
    n_step, double_q = int(rs.randint(1, 4)), bool(rs.randint(0, 2))
    gamma_n = 0.99 ** n_step
    tr = _tr(ref, rs, b, a)
    act, rew, msk = torch.tensor(tr.action), torch.tensor(tr.reward, dtype=torch.float32), torch.tensor(tr.mask, dtype=torch.float32)
    t32 = lambda *s, scale=1.0: torch.tensor((rs.standard_normal(s) * scale).astype(np.float32))
    # --- DQN
    q, qt, qo = t32(b, a).requires_grad_(True), t32(b, a), t32(b, a)
    ag = _Obj()
    ag.config = _cfg(ref, discount=0.99, n_step=n_step, double_q=double_q)
    ag.target_network = _FakeNet([dict(q=qt)])
    ag.network = _FakeNet([dict(q=qo), dict(q=q)] if double_q else [dict(q=q)])
    want = ref.DQNAgent.compute_loss(ag, tr)
    got = L.dqn_td_error(q, qt, act, rew, msk, gamma_n, q_next_online=qo if double_q else None)
    assert torch.equal(got, want)
    gw, = torch.autograd.grad(ref.DQNAgent.reduce_loss(ag, want), q)
    gg, = torch.autograd.grad(L.dqn_reduce(got), q)
    assert torch.equal(gg, gw)
    # --- C51
    n_atoms = int(rs.choice([11, 21, 51]))
    vmin, vmax = -10.0, 10.0
    lg, lt, lo = t32(b, a, n_atoms).requires_grad_(True), t32(b, a, n_atoms, scale=2.0), t32(b, a, n_atoms, scale=2.0)
    mk = lambda z: dict(prob=F.softmax(z, dim=-1), log_prob=F.log_softmax(z, dim=-1))
    ag = _Obj()
    ag.config = _cfg(ref, discount=0.99, n_step=n_step, double_q=double_q, categorical_v_min=vmin, categorical_v_max=vmax,
                     categorical_n_atoms=n_atoms)
    ag.atoms = ref.tensor(np.linspace(vmin, vmax, n_atoms))
    ag.delta_atom = (vmax - vmin) / float(n_atoms - 1)
    ag.batch_indices = ref.range_tensor(b)
    ag.target_network = _FakeNet([mk(lt)])
    ag.network = _FakeNet([mk(lo), mk(lg)] if double_q else [mk(lg)])
    want = ref.CategoricalDQNAgent.compute_loss(ag, tr)
    got = L.c51_kl(F.log_softmax(lg, dim=-1), F.softmax(lt, dim=-1), act, rew, msk, gamma_n, ag.atoms, vmin, vmax,
                   prob_next_online=F.softmax(lo, dim=-1) if double_q else None)
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), rtol=1e-6, atol=1e-6)
    # --- QR-DQN
    nq = int(rs.choice([5, 17, 50]))
    th, tt = t32(b, a, nq).requires_grad_(True), t32(b, a, nq, scale=1.5)
    ag = _Obj()
    ag.config = _cfg(ref, discount=0.99, n_step=n_step, num_quantiles=nq)
    ag.batch_indices = ref.range_tensor(b)
    ag.cumulative_density = ref.tensor((2 * np.arange(nq) + 1) / (2.0 * nq)).view(1, -1)
    ag.target_network = _FakeNet([dict(quantile=tt)])
    ag.network = _FakeNet([dict(quantile=th)])
    want = ref.QuantileRegressionDQNAgent.compute_loss(ag, tr)
    got = L.qr_loss(th, tt, act, rew, msk, gamma_n)
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), rtol=1e-6, atol=1e-6)


def test_network_modules_have_the_reference_names_and_initial_values(ref):
    """Same seed -> same state_dict (names, order, shapes, initial values) as the reference's modules: checkpoints
    interchange key by key.  Covers the heads the on-policy / DQN-family goldens do not construct (Dueling, Rainbow +
    NoisyLinear, OptionCritic, DDPG, TD3).  (Forward passes need the HIP kernels: tests/test_gpu_more_agents.py.)"""
    import deeprl_amd as d
    d.select_device(-1)
    adam = lambda p: torch.optim.Adam(p, 1e-3)
    makers = [
        lambda n: n.NoisyLinear(7, 5),
        lambda n: n.DuelingNet(3, n.FCBody(4, (8,))),
        lambda n: n.RainbowNet(3, 5, n.FCBody(4, (8,), noisy_linear=True), True),
        lambda n: n.OptionCriticNet(n.FCBody(4, (8,)), 2, 3),
        lambda n: n.DeterministicActorCriticNet(4, 2, adam, adam, actor_body=n.FCBody(4, (8,)), critic_body=n.FCBody(6, (8,))),
        lambda n: n.TD3Net(2, lambda: n.FCBody(4, (8,)), lambda: n.FCBody(6, (8,)), adam, adam),
        lambda n: n.VanillaNet(4, n.NatureConvBody()),
        lambda n: n.CategoricalActorCriticNet(4, 3, n.FCBody(4, (8,))),
        lambda n: n.GaussianActorCriticNet(4, 2, actor_body=n.FCBody(4, (8,)), critic_body=n.FCBody(4, (8,))),
    ]
    for mk in makers:
        torch.manual_seed(1)
        mine = mk(d).state_dict()
        torch.manual_seed(1)
        want = mk(ref).state_dict()
        assert list(mine) == list(want)
        for k in want:
            # orthogonal_ goes through LAPACK's QR: identical here (same process, same library)
            assert torch.equal(mine[k].cpu(), want[k]), k


def test_noisy_layers_draw_their_noise_like_the_reference(ref):
    """NoisyLinear.reset_noise / RainbowNet.reset_noise (network_utils.py:73-80, network_heads.py:72-76): same generator
    words, same layer order, same factorised epsilon as the reference's modules, and the generator ends at the same place."""
    import deeprl_amd as d
    d.select_device(-1)
    d.Config.NOISY_LAYER_STD = ref.Config.NOISY_LAYER_STD = 0.5
    tails = []
    dicts = []
    for lib in (d, ref):
        torch.manual_seed(2)
        net = lib.RainbowNet(3, 5, lib.FCBody(4, (8,), noisy_linear=True), True)
        for _ in range(3):
            net.reset_noise()
        dicts.append(net.state_dict())
        tails.append(torch.rand(3))
    assert list(dicts[0]) == list(dicts[1])
    for k in dicts[1]:
        assert torch.equal(dicts[0][k], dicts[1][k]), k
    assert torch.equal(tails[0], tails[1])


def test_random_processes_consume_np_random_like_the_reference(ref):
    import deeprl_amd as d
    for name in ("OrnsteinUhlenbeckProcess", "GaussianProcess"):
        np.random.seed(3)
        a = getattr(d, name)(size=(3,), std=d.LinearSchedule(0.2))
        xs = np.asarray([a.sample().copy() for _ in range(40)])
        a.reset_states()
        tail_a = np.random.rand()
        np.random.seed(3)
        b = getattr(ref, name)(size=(3,), std=ref.LinearSchedule(0.2))
        ys = np.asarray([b.sample().copy() for _ in range(40)])
        b.reset_states()
        assert np.array_equal(xs, ys) and tail_a == np.random.rand()


def test_run_steps_cadence_equals_reference(ref):
    """run_steps (misc.py:19-35): the same sequence of save / log / eval / step / switch_task / close calls for an agent
    whose step() advances total_steps by 4."""
    import deeprl_amd as d

    def trace(run_steps, config_cls):
        calls = []

        class Agent:
            def __init__(self):
                self.config = config_cls()
                self.config.tag, self.config.save_interval, self.config.log_interval = "t", 8, 12
                self.config.eval_interval, self.config.max_steps = 16, 40
                self.total_steps = 0
                self.logger = type("L", (), {"info": lambda s, m: calls.append("log " + m.split(",")[0])})()

            def save(self, name):
                calls.append("save " + name)

            def eval_episodes(self):
                calls.append("eval %d" % self.total_steps)

            def step(self):
                self.total_steps += 4
                calls.append("step")

            def switch_task(self):
                calls.append("switch")

            def close(self):
                calls.append("close")

        run_steps(Agent())
        return calls

    assert trace(d.run_steps, d.Config) == trace(ref.run_steps, ref.Config)
