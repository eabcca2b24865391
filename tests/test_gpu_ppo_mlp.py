"""GPU parity tests of csrc/ppo_mlp.hip (BASELINE configs[2]: PPO, GaussianActorCriticNet over two tanh MLPs) through the C ABI,
against oracle/ppo_mlp_oracle.py (pinned to the reference's own run by tests/test_ppo_mlp_host.py) and against the host classes
the device forms replace.  Bars: bit-exact for the fp64 observation statistics, the environment and every index / counter;
1e-5 (relative, with the stated floors) for fp32 results."""
import ctypes

import numpy as np
import pytest
import torch
from parity_log import record_parity

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
    add_scalar = add_histogram = info


def _rel(a, b, floor):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


# ------------------------------------------------------------------------------------------ stand-alone kernels
@pytest.mark.parametrize("n,d", [(16, 17), (1, 5), (7, 64), (33, 3)])
def test_rms_normalize_equals_host_class_bit_for_bit(dra, n, d):
    """dra_rms_normalize (normalizer.py:28-51 on the device) against normalizers.MeanStdNormalizer stepped on the host: running
    mean / variance / count and the normalised, clipped outputs identical to the last bit over 40 updates, then read-only."""
    from deeprl_amd import ppo_mlp
    from deeprl_amd.normalizers import MeanStdNormalizer
    dev = dra.Config.DEVICE
    rs = np.random.RandomState(n * 100 + d)
    host = MeanStdNormalizer()
    mean = torch.zeros(d, dtype=torch.float64, device=dev)
    var = torch.ones(d, dtype=torch.float64, device=dev)
    count = torch.full((1,), 1e-4, dtype=torch.float64, device=dev)
    for it in range(41):
        x = rs.randn(n, d) * rs.uniform(0.01, 30.0, size=d) + rs.uniform(-5, 5, size=d)
        if it == 40:
            host.set_read_only()
        want = host(x)
        o32, o64 = ppo_mlp.rms_normalize(torch.from_numpy(x).to(dev), mean, var, count, update=it < 40, out_f64=True)
        assert np.array_equal(o64.cpu().numpy(), want), it
        assert np.array_equal(o32.cpu().numpy(), want.astype(np.float32)), it
        assert np.array_equal(mean.cpu().numpy(), host.rms.mean.reshape(-1)), it
        assert np.array_equal(var.cpu().numpy(), host.rms.var.reshape(-1)), it
        assert float(count.cpu()[0]) == host.rms.count, it


def test_mean_std_normalizer_device_call(dra):
    """MeanStdNormalizer called with a float64 DEVICE batch keeps its statistics on the device and returns the float32 tensor
    the host path would have uploaded; state_dict() reads the device statistics back."""
    dev = dra.Config.DEVICE
    rs = np.random.RandomState(2)
    a, b = dra.MeanStdNormalizer(), dra.MeanStdNormalizer()
    for _ in range(5):
        x = rs.randn(16, 17) * 3 + 1
        want = np.asarray(a(x), dtype=np.float32)
        got = b(torch.from_numpy(x).to(dev))
        assert got.is_cuda and got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), want)
    sa, sb = a.state_dict(), b.state_dict()
    assert np.array_equal(sa['mean'], sb['mean']) and np.array_equal(sa['var'], sb['var'])


def test_cont_env_step_equals_host_class_bit_for_bit(dra):
    """dra_cont_env_step against envs.SyntheticContinuous behind DummyVecEnv (auto reset): observations, rewards, terminals and
    counters identical over 300 steps of 6 environments with out-of-range actions (the clip of envs.py:186-189)."""
    from deeprl_amd import ppo_mlp
    from deeprl_amd.envs import DummyVecEnv, SyntheticContinuous
    dev = dra.Config.DEVICE
    n, s_dim, a_dim, horizon = 6, 17, 6, 23
    envs = [SyntheticContinuous(40 + i, s_dim, a_dim, horizon=horizon) for i in range(n)]
    vec = DummyVecEnv(envs)
    state = torch.from_numpy(np.stack(vec.reset())).to(dev)
    counter = torch.zeros(n, dtype=torch.int64, device=dev)
    seed = torch.tensor([e.seed for e in envs], dtype=torch.int64, device=dev)
    rs = np.random.RandomState(5)
    n_done = 0
    for t in range(300):
        act = (rs.randn(n, a_dim) * 1.2).astype(np.float32)
        obs, rew, done, _ = vec.step(np.clip(act, -1.0, 1.0))
        r, dn = ppo_mlp.cont_env_step(state, counter, seed, torch.from_numpy(act).to(dev), horizon)
        assert np.array_equal(state.cpu().numpy(), np.stack(obs)), t
        assert np.array_equal(r.cpu().numpy(), rew) and np.array_equal(dn.cpu().numpy().astype(bool), done), t
        n_done += int(done.sum())
    assert np.array_equal(counter.cpu().numpy(), [e.c for e in envs]) and n_done > 20


def test_gauss_sample_matches_oracle_stream(dra):
    """dra_gauss_sample = mean + scale * hashed standard normal of (seed, sampler step, GLOBAL environment, dimension): equal to
    the oracle's numpy Box-Muller within fp32 library rounding (1e-5 absolute on O(1) normals), advancing its device step
    counter, and rank-invariant (a shard draws its rows of the global matrix)."""
    from deeprl_amd import ppo_mlp
    from oracle.ppo_mlp_oracle import gauss_noise
    dev = dra.Config.DEVICE
    rs = np.random.RandomState(0)
    n, a_dim, seed = 16, 6, 77
    step = torch.full((1,), 5, dtype=torch.int64, device=dev)
    scale = torch.from_numpy(rs.uniform(0.3, 1.5, size=a_dim).astype(np.float32)).to(dev)
    for t in range(5, 9):
        mean = torch.from_numpy(rs.randn(n, a_dim).astype(np.float32)).to(dev)
        got = ppo_mlp.gauss_sample(mean, scale, seed, step).cpu().numpy()
        want = gauss_noise(seed, t, n, np.arange(n), a_dim) * scale.cpu().numpy() + mean.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
        assert int(step.cpu()[0]) == t + 1
    step2 = torch.full((1,), 8, dtype=torch.int64, device=dev)
    shard = ppo_mlp.gauss_sample(mean[4:8], scale, seed, step2, n_global=n, env0=4).cpu().numpy()
    assert np.array_equal(shard, got[4:8])


# ------------------------------------------------------------------------------------------ the update kernel
def _flat_pair(dra, params, lr):
    """A FusedOptimizer over device copies of an oracle parameter dict (order = the dict's)."""
    from deeprl_amd.optim import FusedOptimizer
    dev = dra.Config.DEVICE
    ps = [torch.nn.Parameter(v.detach().clone().to(dev)) for v in params.values()]
    return FusedOptimizer.adopt(torch.optim.Adam(ps, lr)), ps


def _net_struct(fused, ps, step_dev, has_std):
    from deeprl_amd.ppo_mlp import Net
    flat = fused.flat
    b1, b2 = fused.hyper['betas']
    n = Net()
    n.param, n.exp_avg, n.exp_avg_sq, n.step_dev = flat.flat.data_ptr(), fused.state1.data_ptr(), fused.state2.data_ptr(), step_dev.data_ptr()
    offs = [flat.offset_of(p) for p in ps]
    n.off_w1, n.off_b1, n.off_w2, n.off_b2, n.off_w3, n.off_b3 = offs[:6]
    n.off_std = offs[6] if has_std else -1
    n.lr, n.beta1, n.beta2, n.eps = float(fused.hyper['lr']), float(b1), float(b2), float(fused.hyper['eps'])
    return n


def _entries(rs, n, s_dim, a_dim, actor, critic):
    """Rollout-like rows: the behaviour policy is the initial network (so ratios start at 1 and drift as the updates go)."""
    from oracle import ppo_mlp_oracle as O
    state = torch.from_numpy(rs.randn(n, s_dim).astype(np.float32))
    with torch.no_grad():
        pred = O.gaussian_forward(actor, critic, state, noise=torch.from_numpy(rs.randn(n, a_dim).astype(np.float32)))
    adv = torch.from_numpy(rs.randn(n, 1).astype(np.float32))
    ret = pred['v'].detach() + torch.from_numpy(rs.randn(n, 1).astype(np.float32))
    return [state, pred['action'].detach(), pred['log_pi_a'].detach(), ret, adv]


def _run_kernel(dra, actor, critic, entries, perms, mb, clip, ew, target_kl, steps0=(0, 0), dbg=False):
    from deeprl_amd import ppo_mlp
    from deeprl_amd._lib import lib, ptr, stream_ptr
    dev = dra.Config.DEVICE
    fa, pa = _flat_pair(dra, actor, 3e-4)
    fc, pc = _flat_pair(dra, critic, 1e-3)
    s_dim, hidden, a_dim = actor['w1'].shape[1], actor['w1'].shape[0], actor['w3'].shape[0]
    n, epochs = entries[0].shape[0], len(perms)
    steps = torch.tensor(list(steps0), dtype=torch.int64, device=dev)
    cfg = ppo_mlp.Cfg()
    cfg.state_dim, cfg.action_dim, cfg.hidden, cfg.mini_batch = s_dim, a_dim, hidden, mb
    cfg.ratio_clip, cfg.entropy_weight, cfg.kl_limit = clip, ew, 1.5 * target_kl
    na, nc = _net_struct(fa, pa, steps[0:1], True), _net_struct(fc, pc, steps[1:2], False)
    e = [x.to(dev).contiguous() for x in entries]
    perm = torch.from_numpy(np.concatenate([np.asarray(p, dtype=np.int64) for p in perms])).to(dev)
    floats = ctypes.c_int64()
    lib.dra_ppo_mlp_packed_floats(n, epochs, mb, s_dim, ctypes.byref(floats))
    packed = torch.empty(floats.value, dtype=torch.float32, device=dev)
    lib.dra_ppo_mlp_pack(ptr(e[0]), ptr(e[1]), ptr(e[2]), ptr(e[4]), ptr(e[3]), ptr(perm), n, epochs, mb, s_dim, a_dim, ptr(packed),
                         stream_ptr())
    out3 = torch.zeros(3, dtype=torch.float32, device=dev)
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    dbg_t = torch.zeros(ppo_mlp.DBG_FLOATS, dtype=torch.float32, device=dev) if dbg else None
    lib.dra_ppo_mlp_update(ctypes.byref(cfg), ctypes.byref(na), ctypes.byref(nc), ptr(packed), n, epochs, ptr(out3), ptr(counts),
                           ptr(dbg_t), stream_ptr())
    torch.cuda.synchronize()
    return dict(pa=[p.detach().cpu().numpy() for p in pa], pc=[p.detach().cpu().numpy() for p in pc], fa=fa, fc=fc, psa=pa, psc=pc,
                out3=out3.cpu().numpy(), counts=counts.cpu().numpy(), steps=steps.cpu().numpy(),
                dbg=dbg_t.cpu().numpy() if dbg else None)


_DBG = dict(role=32768, h1=0, h2=4096, head=8192, lp=9216, gl=9280, scal=9344, dz3=10240, dz2=11264, dz1=15360, w1=19456, w2=23552,
            w3=27648, b1=28672, b2=28736, b3=28800, std=28816)


@pytest.mark.parametrize("s_dim,a_dim,hidden,mb", [(17, 6, 64, 64), (5, 2, 16, 32), (33, 16, 32, 48), (48, 1, 64, 20)])
def test_update_kernel_first_minibatch_intermediates(dra, s_dim, a_dim, hidden, mb):
    """The debug instantiation's dump of minibatch 0 -- hidden activations, policy mean / value, log-probabilities, loss
    scalars, the loss gradient per row and EVERY parameter gradient of both networks -- against autograd on the reference's
    formulas (oracle.ppo_update): 1e-5 relative with a floor of 1e-5 x the tensor's largest magnitude."""
    from oracle import ppo_mlp_oracle as O
    rs = np.random.RandomState(s_dim * 7 + mb)
    actor, critic = O.init_params(s_dim, a_dim, hidden, seed=s_dim + hidden)
    entries = _entries(rs, mb, s_dim, a_dim, actor, critic)
    # move the policy a little so that ratios differ from 1 and some rows sit outside the clip range
    with torch.no_grad():
        actor['w3'].add_(0.3 * torch.from_numpy(rs.randn(*actor['w3'].shape).astype(np.float32)))
        actor['std'].add_(0.1)
    perms = [np.arange(mb)]
    first = {}
    a0 = {k: v.detach().clone().requires_grad_(True) for k, v in actor.items()}
    c0 = {k: v.detach().clone().requires_grad_(True) for k, v in critic.items()}
    O.ppo_update(a0, c0, entries, perms, mb, 0.2, 0.01, 1e9, first=first)
    k = _run_kernel(dra, actor, critic, entries, perms, mb, 0.2, 0.01, 1e9, dbg=True)
    dbg = k['dbg']

    def cmp(name, got, want, tol=1e-5):
        want = np.asarray(want, dtype=np.float64)
        err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - want)) / max(np.abs(want).max(), 1e-6))
        assert err <= tol, (name, err)

    for role, base, h1, h2, grads in ((0, 0, first['h1a'], first['h2a'], first['actor_grads']),
                                      (1, _DBG['role'], first['h1c'], first['h2c'], first['critic_grads'])):
        d = dbg[base:base + _DBG['role']]
        cmp("h1", d[_DBG['h1']:_DBG['h1'] + 64 * 64].reshape(64, 64)[:mb, :hidden], h1.numpy())
        cmp("h2", d[_DBG['h2']:_DBG['h2'] + 64 * 64].reshape(64, 64)[:mb, :hidden], h2.numpy())
        head = d[_DBG['head']:_DBG['head'] + 64 * 16].reshape(64, 16)
        if role == 0:
            cmp("mean", head[:mb, :a_dim], first['mean'].numpy())
            cmp("log_pi_a", d[_DBG['lp']:_DBG['lp'] + mb], first['log_pi_a'].numpy().reshape(-1))
            cmp("g_log_pi_a", d[_DBG['gl']:_DBG['gl'] + mb], first['g_log_pi_a'].numpy().reshape(-1))
            cmp("policy_loss", d[_DBG['scal']], first['policy_loss'])
            cmp("approx_kl", d[_DBG['scal'] + 2], first['approx_kl'], 1e-4)
        else:
            cmp("v", head[:mb, 0], first['v'].numpy().reshape(-1))
            cmp("g_v", d[_DBG['gl']:_DBG['gl'] + mb], first['g_v'].numpy().reshape(-1))
            cmp("value_loss", d[_DBG['scal'] + 1], first['value_loss'])
        a_out = a_dim if role == 0 else 1
        cmp("dW1", d[_DBG['w1']:_DBG['w1'] + 64 * 64].reshape(64, 64)[:hidden, :s_dim], grads[0].numpy())
        cmp("db1", d[_DBG['b1']:_DBG['b1'] + hidden], grads[1].numpy())
        cmp("dW2", d[_DBG['w2']:_DBG['w2'] + 64 * 64].reshape(64, 64)[:hidden, :hidden], grads[2].numpy())
        cmp("db2", d[_DBG['b2']:_DBG['b2'] + hidden], grads[3].numpy())
        cmp("dW3", d[_DBG['w3']:_DBG['w3'] + 16 * 64].reshape(16, 64)[:a_out, :hidden], grads[4].numpy())
        cmp("db3", d[_DBG['b3']:_DBG['b3'] + a_out], grads[5].numpy())
        if role == 0:
            cmp("dstd", d[_DBG['std']:_DBG['std'] + a_dim], grads[6].numpy())


@pytest.mark.parametrize("s_dim,a_dim,hidden,mb,n,epochs,target_kl", [
    (17, 6, 64, 64, 512, 3, 0.01),       # BASELINE configs[2] shapes (examples.py:497-523), KL gate live
    (17, 6, 64, 64, 200, 2, 1e9),        # remainder minibatch of 8 rows, gate always open
    (5, 2, 16, 32, 128, 3, 0.01),        # the golden fixture's shapes
    (40, 9, 32, 40, 130, 2, 0.002),      # odd everything, tight gate
])
def test_update_kernel_matches_oracle(dra, s_dim, a_dim, hidden, mb, n, epochs, target_kl):
    """dra_ppo_mlp_pack + dra_ppo_mlp_update (all epochs x minibatches in one launch) against the oracle's loop of the
    reference's update (torch autograd + torch.optim.Adam on the CPU): parameters of both networks, Adam moments, step counts
    (the actor's depends on the approx-KL gate of PPO_agent.py:88) and the last minibatch's three loss scalars.  Parameters: 1e-5 relative to the tensor's largest magnitude."""
    from oracle import ppo_mlp_oracle as O
    rs = np.random.RandomState(n + mb)
    actor, critic = O.init_params(s_dim, a_dim, hidden, seed=n)
    entries = _entries(rs, n, s_dim, a_dim, actor, critic)
    perms = [rs.permutation(n) for _ in range(epochs)]
    a0 = {k: v.detach().clone().requires_grad_(True) for k, v in actor.items()}
    c0 = {k: v.detach().clone().requires_grad_(True) for k, v in critic.items()}
    aopt, copt, out3, actor_steps = O.ppo_update(a0, c0, entries, perms, mb, 0.2, 0.01, target_kl)
    k = _run_kernel(dra, actor, critic, entries, perms, mb, 0.2, 0.01, target_kl)
    per_epoch = (n + mb - 1) // mb
    assert int(k['counts'][1]) == per_epoch * epochs and int(k['steps'][1]) == per_epoch * epochs
    assert int(k['counts'][0]) == actor_steps == int(k['steps'][0]), (k['counts'], actor_steps)
    if target_kl == 0.002:
        assert 0 < actor_steps < per_epoch * epochs       # the gate closes once the policy has drifted
    for got, (name, want) in zip(k['pa'], a0.items()):
        w = want.detach().numpy()
        assert np.max(np.abs(got - w)) <= 1e-5 * max(np.abs(w).max(), 1e-2), ("actor", name, np.max(np.abs(got - w)))
    for got, (name, want) in zip(k['pc'], c0.items()):
        w = want.detach().numpy()
        assert np.max(np.abs(got - w)) <= 1e-5 * max(np.abs(w).max(), 1e-2), ("critic", name, np.max(np.abs(got - w)))
    # Adam moments (exp_avg / exp_avg_sq) of every parameter
    worst_m = worst_v = 0.0
    for fused, ps, opt, params in ((k['fa'], k['psa'], aopt, a0), (k['fc'], k['psc'], copt, c0)):
        for p_dev, p_cpu in zip(ps, params.values()):
            st = opt.state[p_cpu]
            m = fused.flat.view(fused.state1, p_dev).cpu().numpy()
            v = fused.flat.view(fused.state2, p_dev).cpu().numpy()
            wm, wv = st['exp_avg'].numpy(), st['exp_avg_sq'].numpy()
            worst_m = max(worst_m, float(np.max(np.abs(m - wm)) / max(np.abs(wm).max(), 1e-6)))
            worst_v = max(worst_v, float(np.max(np.abs(v - wv)) / max(np.abs(wv).max(), 1e-10)))
    want3 = np.asarray(out3, dtype=np.float64)
    worst_s = float(np.max(np.abs(np.asarray(k['out3'], dtype=np.float64) - want3) / np.maximum(np.abs(want3), 0.1)))
    # the measured maxima are written next to the other parity errors (profiles/rNN_parity_errors.json)
    record_parity("ppo_mlp update kernel vs oracle [%d,%d,%d,%d,%d]" % (s_dim, a_dim, hidden, mb, n), adam_exp_avg=worst_m,
                  adam_exp_avg_sq=worst_v, loss_scalars=worst_s)
    # measured on MI355X (round 6, profiles/r06*_parity_errors.json): exp_avg 0.7-2.3e-6, loss scalars 0.8-2.4e-6,
    # exp_avg_sq 1.3-1.6e-5 -- the second moment is QUADRATIC in the gradient, so a gradient inside north_star's 1e-5
    # (test_update_kernel_first_minibatch_matches_autograd holds every parameter gradient to that) is worth 2e-5 there
    assert worst_m <= 1e-5, worst_m
    assert worst_v <= 2e-5, worst_v
    assert worst_s <= 1e-5, (worst_s, k['out3'], out3)


# ------------------------------------------------------------------------------------------ the agent
def _ppo_config(d, fused, device_env, n_env=4, t_len=32, mb=16, hidden=64, seed=3, invariant=True):
    c = d.Config()
    c.merge(dict(game="synthetic-continuous", log_level=0, tag="mlp%d%d" % (fused, device_env), fused_ppo_mlp=fused,
                 device_env=device_env, skip=False, dp_invariant_sampling=invariant, dp_noise_seed=11))
    c.num_workers = n_env
    c.task_fn = lambda: d.Task(c.game, num_envs=c.num_workers, seed=seed, synthetic_done_period=29)
    c.eval_env = d.Task(c.game, seed=seed + 1)
    c.network_fn = lambda: d.GaussianActorCriticNet(
        c.state_dim, c.action_dim, actor_body=d.FCBody(c.state_dim, hidden_units=(hidden, hidden), gate=torch.tanh),
        critic_body=d.FCBody(c.state_dim, hidden_units=(hidden, hidden), gate=torch.tanh))
    c.actor_opt_fn = lambda p: torch.optim.Adam(p, 3e-4)
    c.critic_opt_fn = lambda p: torch.optim.Adam(p, 1e-3)
    c.discount, c.use_gae, c.gae_tau, c.gradient_clip = 0.99, True, 0.95, 0.5
    c.rollout_length, c.optimization_epochs, c.mini_batch_size = t_len, 3, mb
    c.ppo_ratio_clip, c.max_steps, c.target_kl = 0.2, 3e6, 0.01
    c.state_normalizer = d.MeanStdNormalizer()
    c.log_interval = 10 ** 9
    return c


def _run_agent(d, monkeypatch, fused, device_env, rollouts=3, **kw):
    import deeprl_amd.agents as agents_mod
    monkeypatch.setattr(agents_mod, "get_logger", lambda *a, **k: _Quiet())
    c = _ppo_config(d, fused, device_env, **kw)
    d.random_seed(9)
    torch.manual_seed(9)
    torch.cuda.manual_seed_all(9)
    agent = d.PPOAgent(c)
    from deeprl_amd.device_env import DeviceContinuousVec
    assert isinstance(agent.task, DeviceContinuousVec) == (fused and device_env)
    for _ in range(rollouts):
        agent.step()
    torch.cuda.synchronize()
    agent._mlp.sync_counts()
    out = dict(params={k: v.detach().cpu().numpy().copy() for k, v in agent.network.state_dict().items()},
               stats=c.state_normalizer.state_dict(), steps=(agent._fused_actor.steps, agent._fused_critic.steps),
               total=agent.total_steps, launches=agent._mlp.launches, rng=np.random.randint(0, 1 << 30, size=3))
    agent.close()
    return out


def test_agent_persistent_update_equals_generic_path(dra, monkeypatch):
    """PPOAgent.step() over HOST environments with the persistent update kernel (default) against the generic path
    (config.fused_ppo_mlp = False: per-minibatch forward / loss / autograd backward / two optimizer launches): three rollouts of
    4 environments x 32 steps, 3 epochs x 8 minibatches each, same hashed action noise.  Same np.random consumption; parameters
    agree to 2e-4 of each tensor's largest magnitude after 72 Adam steps on each side (per-update agreement is checked at 1e-5
    by test_update_kernel_matches_oracle; Adam turns 1e-7 gradient differences of near-zero gradients into 1e-4 x lr steps)."""
    a = _run_agent(dra, monkeypatch, True, False)
    b = _run_agent(dra, monkeypatch, False, False)
    assert a['launches'] == 3 and b['launches'] == 0
    assert a['total'] == b['total'] and np.array_equal(a['rng'], b['rng'])
    assert a['steps'][1] == b['steps'][1] == 72
    assert abs(a['steps'][0] - b['steps'][0]) <= 2          # a KL within rounding of the gate may fall either way
    for k in a['params']:
        scale = max(np.abs(b['params'][k]).max(), 1e-2)
        assert np.max(np.abs(a['params'][k] - b['params'][k])) <= 2e-4 * scale, k


def test_agent_device_rollout_equals_host_environments(dra, monkeypatch):
    """The device-resident rollout (device_env.DeviceContinuousVec + dra_ppo_mlp_rollout: one launch per 32-step rollout) against
    the same agent stepping envs.SyntheticContinuous + MeanStdNormalizer from python, both with the persistent update kernel and
    the same hashed action noise: observation statistics, parameters and step counts after three rollouts.  The two paths run
    different forward kernels (MFMA in-kernel vs the generic linear launches), so actions differ at 1e-7 and everything downstream
    is compared at 1e-4 of each tensor's largest magnitude; the environment / normaliser arithmetic itself is checked bit for
    bit above."""
    a = _run_agent(dra, monkeypatch, True, True)
    b = _run_agent(dra, monkeypatch, True, False)
    assert a['total'] == b['total'] and np.array_equal(a['rng'], b['rng'])
    assert a['steps'][1] == b['steps'][1] and abs(a['steps'][0] - b['steps'][0]) <= 2
    np.testing.assert_allclose(a['stats']['mean'], b['stats']['mean'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a['stats']['var'], b['stats']['var'], rtol=1e-5, atol=1e-9)
    for k in a['params']:
        scale = max(np.abs(b['params'][k]).max(), 1e-2)
        assert np.max(np.abs(a['params'][k] - b['params'][k])) <= 2e-4 * scale, k


def test_rollout_kernel_matches_oracle(dra):
    """dra_ppo_mlp_rollout against oracle.rollout (torch CPU forwards, oracle environment, oracle normaliser, oracle noise) over
    48 steps of 5 environments with a short horizon: stored observations, actions, log-probabilities, values (1e-5 of the largest
    magnitude), rewards and masks (exact), final counters (exact) and observation statistics (1e-9 relative: the observations
    feeding them depend on fp32 actions)."""
    from deeprl_amd import ppo_mlp
    from deeprl_amd._lib import lib, stream_ptr
    from oracle import ppo_mlp_oracle as O
    from oracle.numerics_oracle import MeanStdNormalizerOracle
    dev = dra.Config.DEVICE
    n, s_dim, a_dim, hidden, t_len, horizon, noise_seed = 5, 17, 6, 64, 48, 19, 4
    actor, critic = O.init_params(s_dim, a_dim, hidden, seed=12)
    envs = [O.ContinuousEnvOracle(70 + i, s_dim, a_dim, horizon) for i in range(n)]
    raw = np.stack([e.reset() for e in envs])
    norm = MeanStdNormalizerOracle()
    cur = np.asarray(norm(raw), dtype=np.float32)
    fa, pa = _flat_pair(dra, actor, 3e-4)
    fc, pc = _flat_pair(dra, critic, 1e-3)
    steps = torch.zeros(2, dtype=torch.int64, device=dev)
    cfg = ppo_mlp.Cfg()
    cfg.state_dim, cfg.action_dim, cfg.hidden, cfg.mini_batch = s_dim, a_dim, hidden, 64
    na, nc = _net_struct(fa, pa, steps[0:1], True), _net_struct(fc, pc, steps[1:2], False)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(dt).to(dev)
    env_state, env_counter = t(raw, torch.float64), torch.zeros(n, dtype=torch.int64, device=dev)
    env_seed = torch.tensor([e.seed for e in envs], dtype=torch.int64, device=dev)
    rms = t(np.concatenate([norm.rms.mean.reshape(-1), norm.rms.var.reshape(-1), [norm.rms.count]]), torch.float64)
    cur_state = t(cur, torch.float32)
    sampler = torch.full((1,), 3, dtype=torch.int64, device=dev)
    f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    o = dict(state=f(t_len, n, s_dim), action=f(t_len, n, a_dim), log_pi_a=f(t_len, n), v=f(t_len + 1, n), reward=f(t_len, n),
             mask=f(t_len, n))
    io = ppo_mlp.RolloutIO()
    io.env_state, io.env_counter, io.env_seed, io.rms = env_state.data_ptr(), env_counter.data_ptr(), env_seed.data_ptr(), rms.data_ptr()
    io.cur_state, io.sampler_step = cur_state.data_ptr(), sampler.data_ptr()
    io.out_state, io.out_action, io.out_log_pi_a = o['state'].data_ptr(), o['action'].data_ptr(), o['log_pi_a'].data_ptr()
    io.out_v, io.out_reward, io.out_mask = o['v'].data_ptr(), o['reward'].data_ptr(), o['mask'].data_ptr()
    io.env0, io.n_global, io.noise_seed, io.horizon = 0, n, noise_seed, horizon
    io.reward_coef, io.rms_epsilon, io.rms_clip, io.rms_update, io.t_len, io.n_env = 1.0, 1e-8, 10.0, 1, t_len, n
    lib.dra_ppo_mlp_rollout(ctypes.byref(cfg), ctypes.byref(na), ctypes.byref(nc), ctypes.byref(io), stream_ptr())
    torch.cuda.synchronize()
    want = O.rollout(actor, critic, envs, raw, norm, cur, t_len, noise_seed, 3)
    assert int(sampler.cpu()[0]) == 3 + t_len + 1
    assert np.array_equal(env_counter.cpu().numpy(), [e.c for e in envs])
    assert np.array_equal(o['mask'].cpu().numpy(), want['mask']) and (want['mask'] == 0).sum() >= 5
    assert np.array_equal(o['reward'].cpu().numpy(), want['reward'])
    for key in ('state', 'action', 'log_pi_a', 'v'):
        got, w = o[key].cpu().numpy(), want[key]
        assert np.max(np.abs(got - w)) <= 1e-5 * max(np.abs(w).max(), 1.0), (key, np.max(np.abs(got - w)))
    np.testing.assert_allclose(cur_state.cpu().numpy(), want['cur_state'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(env_state.cpu().numpy(), want['raw_states'], rtol=1e-6, atol=1e-8)
    h = rms.cpu().numpy()
    np.testing.assert_allclose(h[:s_dim], norm.rms.mean.reshape(-1), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(h[s_dim:2 * s_dim], norm.rms.var.reshape(-1), rtol=1e-7)
    assert h[2 * s_dim] == norm.rms.count
