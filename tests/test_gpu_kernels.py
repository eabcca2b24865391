"""GPU parity tests: every HIP kernel (called through the C-ABI) against the CPU oracle and the
golden fixtures generated from the reference.  Integer / byte / index / fp64 work is compared
bit-exactly; fp32 losses / gradients at 1e-5 (north_star); contractions at 1e-4 relative to the
output scale (different fp32 summation order than the CPU oracle)."""
import random

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden.make_golden_cases import PER_CASES, UNIFORM_CASES, stream  # noqa: E402
from oracle import loss_oracle as L  # noqa: E402
from oracle import net_oracle as NO  # noqa: E402
from oracle import numerics_oracle as NUM  # noqa: E402
from oracle.replay_oracle import UniformReplayOracle  # noqa: E402
from oracle.synth_oracle import synth_transitions  # noqa: E402

TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need an MI355X")
    from deeprl_amd.support import select_device, Config
    select_device(0)
    return Config.DEVICE


def cu(x, dev, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def f32(x, dev):
    return cu(np.asarray(x, dtype=np.float32), dev)


# --------------------------------------------------------------------------------------------- replay
@pytest.mark.parametrize("case", UNIFORM_CASES, ids=[c[0] for c in UNIFORM_CASES])
def test_uniform_replay_golden(golden, dev, case):
    from deeprl_amd.replay import UniformReplay
    g = golden("uniform_replay")
    name, mem, b, h, n, disc, shape, kind, t_len, cps = case
    states, actions, rewards, masks = stream(np.random.RandomState(1000 + ord(name)), t_len, shape, kind, 4, 0.1)
    rep = UniformReplay(memory_size=mem, batch_size=b, n_step=n, discount=disc, history_length=h)
    np.random.seed(2000 + ord(name))
    for t in range(t_len):
        rep.feed(dict(state=states[t][None], action=actions[t:t + 1], reward=[rewards[t]], mask=masks[t:t + 1]))
        if t in cps:
            tr = rep.sample()
            k = "%s_t%d_" % (name, t)
            assert np.array_equal(g[k + "pos_size"], [rep.pos, rep.size()])
            for key in ("state", "action", "reward", "next_state", "mask"):
                got, want = getattr(tr, key).cpu().numpy(), g[k + key]
                assert got.shape == want.shape and got.dtype == want.dtype, (key, got.dtype, want.dtype, got.shape)
                assert np.array_equal(got, want), key
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g[name + "_rng_tail"])
    rep.close()


@pytest.mark.parametrize("case", PER_CASES, ids=[c[0] for c in PER_CASES])
@pytest.mark.parametrize("ordered", [False, True])
def test_prioritized_replay_golden(golden, dev, case, ordered):
    from deeprl_amd.replay import PrioritizedReplay
    g = golden("prioritized_replay")
    name, mem, b, h, n, disc, shape, kind, t_len, every = case
    rs = np.random.RandomState(3000 + ord(name))
    states, actions, rewards, masks = stream(rs, t_len, shape, kind, 4, 0.1)
    rep = PrioritizedReplay(memory_size=mem, batch_size=b, n_step=n, discount=disc, history_length=h)
    rep.ordered_updates = ordered
    random.seed(4000 + ord(name))
    np.random.seed(4000 + ord(name))
    ks = 0
    for t in range(t_len):
        rep.feed(dict(state=states[t][None], action=actions[t:t + 1], reward=[rewards[t]], mask=masks[t:t + 1]))
        if t >= h + n + 6 and t % every == 0:
            tr = rep.sample()
            k = "%s_s%d_" % (name, ks)
            for key in ("state", "action", "reward", "next_state", "mask", "sampling_prob", "idx"):
                got, want = getattr(tr, key).cpu().numpy(), g[k + key]
                assert got.dtype == want.dtype, (key, got.dtype, want.dtype)
                assert np.array_equal(got, want), (key, ks)
            rs.standard_normal(b)
            prio = g[k + "prio"]
            rep.update_priorities(zip(tr.idx.cpu().numpy(), prio))
            assert np.array_equal(rep.tree.as_tensor().cpu().numpy(), g[k + "tree"]), ks
            assert float(rep.max_priority) == float(g[k + "max_priority"])
            ks += 1
    assert ks == int(g[name + "_n_samples"])
    assert np.array_equal([random.random() for _ in range(3)], g[name + "_rng_tail"])
    rep.close()


@pytest.mark.parametrize("case", PER_CASES, ids=[c[0] for c in PER_CASES])
@pytest.mark.parametrize("ordered", [False, True])
def test_prioritized_replay_device_commit_golden(golden, dev, case, ordered):
    """The same reference op log with the priorities handed over as a DEVICE fp32 tensor (what the fused learner's loss
    kernel leaves behind): commit_device applies pending gating / first-writer-wins on the host and writes the leaves,
    max_priority included, without a host round trip (dra_sumtree_commit_f32); new transitions are then fed at the
    device-resident max_priority (dra_sumtree_set_from).  Tree, samples and RNG stream equal the reference's."""
    from deeprl_amd.replay import PrioritizedReplay
    g = golden("prioritized_replay")
    name, mem, b, h, n, disc, shape, kind, t_len, every = case
    rs = np.random.RandomState(3000 + ord(name))
    states, actions, rewards, masks = stream(rs, t_len, shape, kind, 4, 0.1)
    rep = PrioritizedReplay(memory_size=mem, batch_size=b, n_step=n, discount=disc, history_length=h)
    rep.ordered_updates = ordered
    random.seed(4000 + ord(name))
    np.random.seed(4000 + ord(name))
    ks = 0
    for t in range(t_len):
        rep.feed(dict(state=states[t][None], action=actions[t:t + 1], reward=[rewards[t]], mask=masks[t:t + 1]))
        if t >= h + n + 6 and t % every == 0:
            tr = rep.sample()
            k = "%s_s%d_" % (name, ks)
            for key in ("state", "action", "reward", "next_state", "mask", "sampling_prob", "idx"):
                assert np.array_equal(getattr(tr, key).cpu().numpy(), g[k + key]), (key, ks)
            rs.standard_normal(b)
            prio = np.asarray(g[k + "prio"])
            assert prio.dtype == np.float32
            rep.commit_device(tr.idx.cpu().numpy(), torch.from_numpy(prio).to(dev))
            assert np.array_equal(rep.tree.as_tensor().cpu().numpy(), g[k + "tree"]), ks
            assert float(rep.max_priority) == float(g[k + "max_priority"])
            ks += 1
    assert ks == int(g[name + "_n_samples"])
    assert np.array_equal([random.random() for _ in range(3)], g[name + "_rng_tail"])
    rep.close()


def test_sumtree_commit_falls_back_to_ordered_walk(dev):
    """Priorities spanning 2^-40 .. 2^20 on a 4096-leaf tree violate the fp64 exactness bound of the level-parallel
    update (capacity * max / ulp_f32(min) > 2^53): dra_sumtree_commit_f32 notices on the device and replays the
    reference's incremental walk, so the heap equals the numpy restatement of sum_tree.py bit for bit; the host-side
    update_priorities takes the same decision (PrioritizedReplay._exact_parallel)."""
    from deeprl_amd import ops
    from oracle.sumtree_oracle import SumTreeOracle
    cap = 4096
    rs = np.random.RandomState(5)
    tree, orc = ops.SumTree(cap), SumTreeOracle(cap)
    stat = torch.tensor([1.0, 1.0], dtype=torch.float64, device=dev)
    for i in range(cap):
        tree.set(i + cap - 1, 1.0)
        orc.pending.add(i + cap - 1)
        orc.update(i + cap - 1, 1.0)
    for r in range(12):
        li = rs.choice(cap, 32, replace=False).astype(np.int64) + cap - 1
        pr = (np.ldexp(1.0 + rs.rand(32), rs.randint(-40, 21, size=32))).astype(np.float32)
        pos = rs.permutation(32).astype(np.int32)
        tree.commit_f32(cu(li, dev), torch.from_numpy(pos).to(dev), torch.from_numpy(pr).to(dev), stat)
        for k in range(32):
            orc.pending.add(int(li[k]))
            orc.update(int(li[k]), float(pr[pos[k]]))
    torch.cuda.synchronize()
    assert np.array_equal(tree.as_tensor().cpu().numpy(), orc.tree)
    hi, lo = stat.cpu().tolist()
    assert hi >= 2.0 ** 19 and lo < 2.0 ** -38
    tree.close()


def test_adam_step_counter_golden(golden, dev):
    """dra_adam_step_counter (step count in device memory, bias corrections formed in the kernel, mirrored parameter copy)
    against the same torch.optim.Adam trajectory as dra_adam_step."""
    from deeprl_amd import ops
    g = golden("optim")
    shapes = [g["p0_%d" % j].shape for j in range(4)]
    sizes = [int(np.prod(s)) for s in shapes]
    flat = lambda arrs: np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in arrs])
    for name, clip, lr, eps in (("adam", 5.0, 2.5e-4, 0.01 / 32), ("adam", 0.5, 2.5e-4, 0.01 / 32), ("adam_default", 5.0, 3e-4, 1e-8)):
        p = f32(flat([g["p0_%d" % j] for j in range(4)]), dev)
        n_pad = (p.numel() + 3) // 4 * 4
        buf = torch.zeros(4, n_pad, dtype=torch.float32, device=dev)      # 16-byte aligned rows
        buf[0, :p.numel()] = p
        pp, s1, s2, cp = buf[0, :p.numel()], buf[1, :p.numel()], buf[2, :p.numel()], buf[3, :p.numel()]
        npart = ops.norm_partials()
        partials = torch.zeros(npart, dtype=torch.float64, device=dev)
        norm = torch.zeros(1, dtype=torch.float32, device=dev)
        step = torch.zeros(1, dtype=torch.int64, device=dev)
        for i in range(5):
            gr = torch.zeros(n_pad, dtype=torch.float32, device=dev)
            gr[:p.numel()] = f32(flat([g["g%d_%d" % (i, j)] for j in range(4)]), dev)
            ops.grad_sqnorm(gr[:p.numel()], partials)
            step += 1
            ops.adam_step_counter(pp, gr[:p.numel()], s1, s2, partials, npart, clip, lr, 0.9, 0.999, eps, step, norm, cp)
        got = pp.cpu().numpy()
        assert np.array_equal(got, cp.cpu().numpy())
        off = 0
        for j in range(4):
            np.testing.assert_allclose(got[off:off + sizes[j]].reshape(shapes[j]), g["%s_clip%g_p%d" % (name, clip, j)],
                                       rtol=1e-5, atol=1e-6)
            off += sizes[j]


def test_ring_vs_oracle_atari_shapes(dev):
    """84x84 frames (16-byte vector path), H=4, n=3, wrap-around, device-side synthetic fill."""
    from deeprl_amd import ops
    cap, h, n, gamma, fb = 5000, 4, 3, 0.99, 7056
    ring = ops.Ring(cap, fb, 8, h, n, gamma)
    orc = UniformReplayOracle(cap, 32, n, gamma, h)
    total = 7300  # wraps
    frames, act, rew, msk = synth_transitions(0, total, fb, seed=3, n_actions=4, done_period=50)
    pos = 0
    done = 0
    while done < total:  # device fill in runs that do not wrap
        run = min(total - done, cap - pos)
        ring.fill_synthetic(pos, run, done, 3, n_actions=4, done_period=50)
        pos = (pos + run) % cap
        done += run
    for t in range(total):
        orc.feed_one(frames[t].reshape(84, 84), act[t], rew[t], msk[t])
    np.random.seed(5)
    idx = orc.draw_indices(256)
    want = orc.gather(idx)
    got = ring.gather(cu(idx, dev), (84, 84), torch.uint8, torch.int64, want_f32=True)
    torch.cuda.synchronize()
    assert np.array_equal(got["state"].cpu().numpy(), want[0])
    assert np.array_equal(got["action"].cpu().numpy(), want[1])
    assert np.array_equal(got["reward"].cpu().numpy(), want[2])
    assert np.array_equal(got["next_state"].cpu().numpy(), want[3])
    assert np.array_equal(got["mask"].cpu().numpy(), want[4])
    assert np.array_equal(got["reward_f32"].cpu().numpy(), want[2].astype(np.float32))
    assert np.array_equal(got["mask_f32"].cpu().numpy(), want[4].astype(np.float32))
    ring.close()


@pytest.mark.parametrize("h,n", [(4, 1), (4, 3), (1, 1), (2, 5)])
def test_gather_block_views_equal_the_two_tensor_form(dev, h, n):
    """dra_ring_gather_block (replay.py:112-140 with state / next_state as two VIEWS of one [B, history + n_step, ...] block:
    every frame of a sample's run written once) == dra_ring_gather's two stacked tensors, bit for bit, for history / n-step
    combinations incl. history 1 and n_step > history; refilling the returned dict in place keeps the block form; the uint8 ->
    f32 table kernel reads the views in place (dra_u8_to_f32_lut_rows) and gives what it gives on contiguous copies."""
    from deeprl_amd import ops
    from deeprl_amd.normalizers import ImageNormalizer
    cap, fb = 3000, 84 * 84
    ring = ops.Ring(cap, fb, 8, h, n, 0.99)
    ring.fill_synthetic(0, cap, 0, 9, n_actions=6, done_period=37)
    rs = np.random.RandomState(h * 10 + n)
    idx = cu(rs.randint(h + 5, cap - n - 5, size=96).astype(np.int64), dev)
    shape = (84, 84)
    a = ring.gather(idx, shape, torch.uint8, torch.int64, want_f32=True)
    b = ring.gather(idx, shape, torch.uint8, torch.int64, want_f32=True, block=False)
    assert a["block"].shape[1] == h + n and not (h > 1 and a["state"].is_contiguous()) and b["state"].is_contiguous()
    for k in ("state", "next_state", "action", "reward", "mask", "reward_f32", "mask_f32"):
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    idx2 = cu(rs.randint(h + 5, cap - n - 5, size=96).astype(np.int64), dev)
    ptr0 = a["state"].data_ptr()
    a2 = ring.gather(idx2, shape, torch.uint8, torch.int64, want_f32=True, out=a)
    b2 = ring.gather(idx2, shape, torch.uint8, torch.int64, want_f32=True, block=False)
    assert a2 is a and a["state"].data_ptr() == ptr0
    assert torch.equal(a["state"], b2["state"]) and torch.equal(a["next_state"], b2["next_state"]) and torch.equal(a["reward"], b2["reward"])
    norm = ImageNormalizer()
    for k in ("state", "next_state"):
        assert torch.equal(norm(a[k]), norm(b2[k]))
    ring.close()


def test_image_lut_bit_exact(dev):
    from deeprl_amd.normalizers import ImageNormalizer
    rs = np.random.RandomState(0)
    x = rs.randint(0, 256, size=(5, 4, 84, 84)).astype(np.uint8)
    x.reshape(-1)[:256] = np.arange(256)
    got = ImageNormalizer()(cu(x, dev)).cpu().numpy()
    want = NUM.image_normalize_sync(x)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    odd = x.reshape(-1)[:1003].copy()  # tail path (n % 16 != 0)
    got = ImageNormalizer()(cu(odd, dev)).cpu().numpy()
    assert np.array_equal(got, NUM.image_normalize_sync(odd))


@pytest.mark.parametrize("cap", [8, 13, 50])
def test_sumtree_ops_golden(golden, dev, cap):
    """Replays the reference SumTree's op log (add / get / update with pending gating on the host,
    as deeprl_amd.replay does) and compares the device heap array node by node."""
    from deeprl_amd import ops
    g = golden("sumtree")
    tree = ops.SumTree(cap)
    pending, write = set(), 0
    for op, x, p_out, idx in g["cap%d_log" % cap]:
        op, idx = int(op), int(idx)
        if op == 0:  # add: self-mark pending, update, advance the write cursor
            leaf = write + cap - 1
            tree.set(leaf, float(np.float32(x)))
            pending.discard(leaf)
            write = (write + 1) % cap
        elif op == 1:  # get(s): arbitrary-s descents are covered by the PrioritizedReplay goldens
            pending.add(idx)
        elif idx in pending:  # update: only pending leaves
            pending.remove(idx)
            tree.update(cu([idx], dev), cu(np.asarray([np.float32(x)], dtype=np.float64), dev))
    assert np.array_equal(tree.as_tensor().cpu().numpy(), g["cap%d_tree" % cap])
    assert np.array_equal(sorted(pending), g["cap%d_pending" % cap])
    tree.close()


@pytest.mark.gpu
def test_sumtree_large_parallel_equals_ordered(dev):
    """1M-leaf tree: parallel level-by-level update == reference-order incremental update ==
    bottom-up rebuild, for fp32-valued priorities (the exactness regime of SURVEY.md section 7)."""
    from deeprl_amd import ops
    cap = 1_000_003
    rs = np.random.RandomState(1)
    ta, tb = ops.SumTree(cap), ops.SumTree(cap)
    leaves0 = cu(np.arange(cap - 1, 2 * cap - 1, dtype=np.int64), dev)
    view_a, view_b = ta.as_tensor(), tb.as_tensor()
    init = torch.ones(cap, dtype=torch.float64, device=dev)
    view_a[cap - 1:] = init
    view_b[cap - 1:] = init
    ta.rebuild()
    tb.rebuild()
    for _ in range(20):
        leaf = rs.choice(cap, size=32, replace=False).astype(np.int64) + cap - 1
        prio = np.sqrt(np.abs(rs.standard_normal(32)).astype(np.float32) + np.float32(0.01)).astype(np.float64)
        ta.update(cu(leaf, dev), cu(prio, dev), ordered=False)
        tb.update(cu(leaf, dev), cu(prio, dev), ordered=True)
    torch.cuda.synchronize()
    a, b = view_a.cpu().numpy(), view_b.cpu().numpy()
    assert np.array_equal(a, b)
    tb.rebuild()
    assert np.array_equal(view_b.cpu().numpy(), a)
    # stratified sampling agrees with the oracle descent on the same tree
    from oracle.sumtree_oracle import SumTreeOracle
    o = SumTreeOracle(cap)
    o.tree = a.copy()
    u = rs.rand(32)
    idx, p, total = ta.sample(cu(u, dev))
    seg = o.total() / 32
    for i in range(32):
        s = seg * i + (seg * (i + 1) - seg * i) * u[i]
        oi, op_, _ = o.get(s)
        assert oi == int(idx[i]) and op_ == float(p[i])
    assert float(total) == o.total()
    ta.close()
    tb.close()
    del leaves0


# --------------------------------------------------------------------------------------------- losses
@pytest.mark.parametrize("tag", ["b32a4", "b10a2n3", "b32a4dq", "b7a18"])
@pytest.mark.parametrize("act_i64", [True, False])
def test_td_loss_golden(golden, dev, tag, act_i64):
    from deeprl_amd import ops
    g = golden("dqn_loss")
    k = tag + "_"
    gamma, n_step, double_q, eps, alpha, beta = g[k + "cfg"]
    action = cu(g[k + "action"], dev) if act_i64 else f32(g[k + "action"], dev)
    common = dict(q=f32(g[k + "q"], dev), q_next_target=f32(g[k + "q_next_t"], dev), action=action,
                  reward=f32(g[k + "reward"], dev), mask=f32(g[k + "mask"], dev), gamma_n=gamma ** int(n_step),
                  q_next_online=f32(g[k + "q_next_o"], dev) if double_q else None)
    out = ops.td_loss(**common)
    np.testing.assert_allclose(out["delta"].cpu().numpy(), g[k + "loss_vec"], **TOL)
    np.testing.assert_allclose(out["loss"].item(), g[k + "loss"], **TOL)
    np.testing.assert_allclose(out["dq"].cpu().numpy(), g[k + "grad_q"], **TOL)
    out = ops.td_loss(sampling_prob=f32(g[k + "sampling_prob"], dev), beta=beta, replay_eps=eps, replay_alpha=alpha,
                      **common)
    np.testing.assert_allclose(out["prio"].cpu().numpy(), g[k + "prio"], **TOL)
    np.testing.assert_allclose(out["weights"].cpu().numpy(), g[k + "w"], **TOL)
    np.testing.assert_allclose(out["loss"].item(), g[k + "loss_per"], **TOL)
    np.testing.assert_allclose(out["dq"].cpu().numpy(), g[k + "grad_q_per"], **TOL)


@pytest.mark.parametrize("tag", ["b32a4", "b8a3n3dq", "b5a6at21"])
def test_c51_loss_golden(golden, dev, tag):
    from deeprl_amd import ops
    g = golden("c51_loss")
    k = tag + "_"
    gamma, n_step, double_q, vmin, vmax, n_atoms = g[k + "cfg"]
    atoms = f32(np.linspace(vmin, vmax, int(n_atoms)), dev)
    out = ops.c51_loss(f32(g[k + "logits"], dev), f32(g[k + "logits_next_t"], dev), cu(g[k + "action"], dev),
                       f32(g[k + "reward"], dev), f32(g[k + "mask"], dev), gamma ** int(n_step), atoms, vmin, vmax,
                       logits_next_online=f32(g[k + "logits_next_o"], dev) if double_q else None)
    np.testing.assert_allclose(out["kl"].cpu().numpy(), g[k + "kl"], **TOL)
    np.testing.assert_allclose(out["loss"].item(), g[k + "loss"], **TOL)
    np.testing.assert_allclose(out["dlogits"].cpu().numpy(), g[k + "grad_logits"], **TOL)


@pytest.mark.parametrize("tag", ["b32a4", "b6a3q17n3"])
def test_qr_loss_golden(golden, dev, tag):
    from deeprl_amd import ops
    g = golden("qr_loss")
    k = tag + "_"
    gamma, n_step, nq = g[k + "cfg"]
    out = ops.qr_loss(f32(g[k + "theta"], dev), f32(g[k + "theta_next_t"], dev), cu(g[k + "action"], dev),
                      f32(g[k + "reward"], dev), f32(g[k + "mask"], dev), gamma ** int(n_step))
    np.testing.assert_allclose(out["loss_vec"].cpu().numpy(), g[k + "loss_vec"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["loss"].item(), g[k + "loss"], **TOL)
    np.testing.assert_allclose(out["dtheta"].cpu().numpy(), g[k + "grad_theta"], **TOL)


@pytest.mark.parametrize("tag", ["m64", "m256", "m5"])
def test_ppo_loss_golden(golden, dev, tag):
    from deeprl_amd import ops
    g = golden("ppo_loss")
    k = tag + "_"
    out3, gs = ops.ppo_loss(f32(g[k + "lp"], dev), f32(g[k + "ent"], dev), f32(g[k + "v"], dev), f32(g[k + "old_lp"], dev),
                            f32(g[k + "adv"], dev), f32(g[k + "ret"], dev), 0.2, 0.01)
    np.testing.assert_allclose(out3.cpu().numpy(), g[k + "out"], **TOL)
    for got, key in zip(gs, ("g_lp", "g_ent", "g_v")):
        np.testing.assert_allclose(got.cpu().numpy(), g[k + key], **TOL)


def test_a2c_loss_vs_oracle(dev):
    from deeprl_amd import ops
    rs = np.random.RandomState(0)
    m = 80
    arr = [rs.standard_normal((m, 1)).astype(np.float32) for _ in range(5)]
    lp, ent, v = [torch.tensor(a, requires_grad=True) for a in arr[:3]]
    loss = L.a2c_loss(lp, ent, v, torch.tensor(arr[3]), torch.tensor(arr[4]), 0.01, 1.0)
    gs = torch.autograd.grad(loss, [lp, ent, v])
    out4, gg = ops.a2c_loss(*[f32(a, dev) for a in arr], 0.01, 1.0)
    np.testing.assert_allclose(out4[0].item(), loss.item(), **TOL)
    for got, want in zip(gg, gs):
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), **TOL)


# --------------------------------------------------------------------------------------------- scan
@pytest.mark.parametrize("name,tag", [("a2c_step", "t5n16"), ("a2c_step", "t5n16gae"), ("a2c_step", "t20n3gae"),
                                      ("ppo_step", "t64n2"), ("ppo_step", "t32n4")])
def test_gae_golden(golden, dev, name, tag):
    from deeprl_amd import ops
    g = golden(name)
    k = tag + "_"
    cfg = g[k + "cfg"]
    gamma, tau = cfg[0], cfg[1]
    use_gae = bool(cfg[2]) if name == "a2c_step" else True
    adv, ret = ops.gae(f32(g[k + "reward"], dev), f32(g[k + "mask"], dev), f32(g[k + "v"], dev), gamma, tau, use_gae)
    np.testing.assert_allclose(adv.cpu().numpy(), g[k + "adv"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), g[k + "ret"], rtol=1e-5, atol=1e-5)
    if name == "ppo_step":
        flat = adv.reshape(-1, 1).clone()
        ops.adv_normalize_(flat)
        np.testing.assert_allclose(flat.cpu().numpy(), g[k + "ent_adv_normalized"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("t_len,n_env,use_gae", [(2048, 16, True), (2048, 1, True), (128, 8, True), (5, 16, False),
                                                 (700, 5, True)])
def test_gae_vs_oracle_baseline_sizes(dev, t_len, n_env, use_gae):
    from deeprl_amd import ops
    rs = np.random.RandomState(t_len + n_env)
    r = rs.standard_normal((t_len, n_env, 1)).astype(np.float32)
    m = (rs.rand(t_len, n_env, 1) > 0.01).astype(np.float32)
    v = rs.standard_normal((t_len + 1, n_env, 1)).astype(np.float32)
    wa, wr = L.gae_reverse(torch.tensor(r), torch.tensor(m), torch.tensor(v), 0.99, 0.95, use_gae)
    adv, ret = ops.gae(f32(r, dev), f32(m, dev), f32(v, dev), 0.99, 0.95, use_gae)
    np.testing.assert_allclose(adv.cpu().numpy(), wa.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), wr.numpy(), rtol=1e-5, atol=1e-5)


# --------------------------------------------------------------------------------------------- optimiser
@pytest.mark.parametrize("name", ["rmsprop_centered", "rmsprop_plain", "adam", "adam_default"])
@pytest.mark.parametrize("clip", [5.0, 0.5])
def test_optim_golden(golden, dev, name, clip):
    from deeprl_amd import ops
    g = golden("optim")
    shapes = [g["p0_%d" % j].shape for j in range(4)]
    sizes = [int(np.prod(s)) for s in shapes]
    n = sum(sizes)
    flat = lambda arrs: np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in arrs])
    p = f32(flat([g["p0_%d" % j] for j in range(4)]), dev)
    s1, s2 = torch.zeros_like(p), torch.zeros_like(p)
    npart = ops.norm_partials()
    partials = torch.zeros(npart, dtype=torch.float64, device=dev)
    norm = torch.zeros(1, dtype=torch.float32, device=dev)
    norms = []
    for i in range(5):
        gr = f32(flat([g["g%d_%d" % (i, j)] for j in range(4)]), dev)
        ops.grad_sqnorm(gr, partials)
        if name == "rmsprop_centered":
            ops.rmsprop_step(p, gr, s1, s2, partials, npart, clip, 0.00025, 0.95, 0.01, True, norm)
        elif name == "rmsprop_plain":
            ops.rmsprop_step(p, gr, s1, s2, partials, npart, clip, 1e-4, 0.99, 1e-5, False, norm)
        elif name == "adam":
            ops.adam_step(p, gr, s1, s2, partials, npart, clip, 2.5e-4, 0.9, 0.999, 0.01 / 32, i + 1, norm)
        else:
            ops.adam_step(p, gr, s1, s2, partials, npart, clip, 3e-4, 0.9, 0.999, 1e-8, i + 1, norm)
        norms.append(norm.item())
    np.testing.assert_allclose(norms, g["%s_clip%g_norms" % (name, clip)], rtol=1e-6)
    got = p.cpu().numpy()
    off = 0
    for j in range(4):
        np.testing.assert_allclose(got[off:off + sizes[j]].reshape(shapes[j]), g["%s_clip%g_p%d" % (name, clip, j)],
                                   rtol=1e-5, atol=1e-6)
        off += sizes[j]
    assert off == n


def test_grad_sqnorm_folds_slabs(dev):
    from deeprl_amd import ops
    rs = np.random.RandomState(2)
    n, s = 78_563, 7
    stride = (n + 3) // 4 * 4
    slabs = rs.standard_normal((s, stride)).astype(np.float32)
    grad = torch.zeros(n, dtype=torch.float32, device=dev)
    partials = torch.zeros(ops.norm_partials(), dtype=torch.float64, device=dev)
    ops.grad_sqnorm(grad, partials, slabs=f32(slabs, dev), n_slabs=s, slab_stride=stride)
    want = slabs[0, :n].copy()
    for k in range(1, s):
        want = want + slabs[k, :n]
    assert np.array_equal(grad.cpu().numpy(), want)
    np.testing.assert_allclose(partials.sum().item(), (want.astype(np.float64) ** 2).sum(), rtol=1e-6)


# --------------------------------------------------------------------------------------------- contractions
def _scale_close(got, want, rel=1e-5, what=None):
    """Contraction parity bar (north star: fp32 within 1e-5): max |got - want| <= rel * max |want|, `want` an fp64
    contraction of the SAME fp32 operands (round 3 compared with an fp32 CPU result that carries its own summation noise,
    at 1e-4).  The measured error / scale goes to the parity log (profiles/r04*_parity_errors.json)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert want.dtype == np.float64
    scale = np.abs(want).max() + 1e-12
    err = np.abs(got - want).max()
    if what is None:
        import inspect
        fr = inspect.stack()[1]
        what = "%s:%d" % (fr.function, fr.lineno)
    from parity_log import record_parity
    record_parity("contraction " + what, err_over_scale=err / scale, scale=scale)
    assert err <= rel * scale, "max abs err %.3e vs scale %.3e (%.2e of it)" % (err, scale, err / scale)


def t64(a, grad=True):
    """fp32 operand values as an fp64 CPU tensor: the reference contraction then has no summation noise of its own."""
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


CONV = {1: (4, 84, 32, 8, 4), 2: (32, 20, 64, 4, 2), 3: (64, 9, 64, 3, 1)}


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("batch", [32, 1, 5])
def test_conv_fwd_bwd_vs_oracle(dev, layer, batch):
    """Forward (two weight sets in one launch), weight/bias gradient (split-K slabs folded by
    dra_grad_sqnorm) and input gradient against F.conv2d autograd on the CPU."""
    import torch.nn.functional as F
    from deeprl_amd import ops
    c, h, oc, k, s = CONV[layer]
    rs = np.random.RandomState(10 * layer + batch)
    ws = [(rs.standard_normal((oc, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32) for _ in range(2)]
    bs = [(rs.standard_normal(oc) * 0.1).astype(np.float32) for _ in range(2)]
    if layer == 1:
        xs_u8 = [rs.randint(0, 256, size=(batch, c, h, h)).astype(np.uint8) for _ in range(2)]
        xs = [NUM.image_normalize_sync(x) for x in xs_u8]
        ys = ops.conv_fwd(1, [cu(x, dev) for x in xs_u8], [f32(w, dev) for w in ws], [f32(b, dev) for b in bs],
                          act="relu", u8_coef=1.0 / 255)
    else:
        xs = [np.maximum(rs.standard_normal((batch, c, h, h)), 0).astype(np.float32) for _ in range(2)]
        ys = ops.conv_fwd(layer, [f32(x, dev) for x in xs], [f32(w, dev) for w in ws], [f32(b, dev) for b in bs],
                          act="relu")
    refs = []
    for z in range(2):
        xt = t64(xs[z])
        wt, bt = t64(ws[z]), t64(bs[z])
        pre = F.conv2d(xt, wt, bt, stride=s)
        refs.append((xt, wt, bt, pre))
        _scale_close(ys[z].cpu().numpy(), F.relu(pre).detach().numpy())
    # backward of set 0 with a random upstream gradient (w.r.t. the post-ReLU output).  The reference differentiates through
    # the DEVICE's ReLU gate: an fp32 pre-activation within rounding of zero may be gated differently from the fp64 one,
    # which is the gate caveat of DESIGN.md section 2, not a contraction error
    xt, wt, bt, pre = refs[0]
    dy = rs.standard_normal(tuple(pre.shape)).astype(np.float32)
    pre.backward(t64(dy * (ys[0].cpu().numpy() > 0), False))
    dpre = ops.act_bwd(f32(dy, dev), ys[0], "relu")
    x_dev = cu(xs_u8[0], dev) if layer == 1 else f32(xs[0], dev)
    ksplit = 16
    dw_s, db_s = ops.conv_bwd_w(layer, dpre, x_dev, ksplit=ksplit, u8_coef=1.0 / 255 if layer == 1 else None)
    _scale_close(dw_s.sum(0).cpu().numpy().reshape(wt.shape), wt.grad.numpy())
    _scale_close(db_s.sum(0).cpu().numpy(), bt.grad.numpy())
    if layer > 1:
        dx = ops.conv_bwd_x(layer, dpre, f32(ws[0], dev))
        _scale_close(dx.cpu().numpy(), xt.grad.numpy())
        # fused activation-derivative mask of the layer below (xact = this layer's input, post-ReLU)
        dxm = ops.conv_bwd_x(layer, dpre, f32(ws[0], dev), xact=f32(xs[0], dev), act="relu")
        _scale_close(dxm.cpu().numpy(), xt.grad.numpy() * (xs[0] > 0))


@pytest.mark.parametrize("batch,fin,fout,act", [(32, 3136, 512, "relu"), (32, 512, 4, None), (32, 512, 204, None),
                                                (10, 4, 64, "relu"), (64, 17, 64, "tanh"), (1, 3136, 512, "relu"),
                                                (256, 64, 1, None), (32, 512, 800, None), (256, 3136, 512, "relu"),
                                                (80, 3136, 512, "relu"), (5, 1024, 512, None)])
def test_linear_fwd_bwd_vs_oracle(dev, batch, fin, fout, act):
    import torch.nn.functional as F
    from deeprl_amd import ops
    rs = np.random.RandomState(batch + fin + fout)
    xs = [rs.standard_normal((batch, fin)).astype(np.float32) for _ in range(2)]
    ws = [(rs.standard_normal((fout, fin)) / np.sqrt(fin)).astype(np.float32) for _ in range(2)]
    bs = [(rs.standard_normal(fout) * 0.1).astype(np.float32) for _ in range(2)]
    ys = ops.linear_fwd([f32(x, dev) for x in xs], [f32(w, dev) for w in ws], [f32(b, dev) for b in bs], act=act)
    fn = {None: lambda t: t, "relu": F.relu, "tanh": torch.tanh}[act]
    refs = []
    for z in range(2):
        xt = t64(xs[z])
        wt, bt = t64(ws[z]), t64(bs[z])
        pre = F.linear(xt, wt, bt)
        refs.append((xt, wt, bt, pre))
        _scale_close(ys[z].cpu().numpy(), fn(pre).detach().numpy())
    xt, wt, bt, pre = refs[1]
    dy = rs.standard_normal((batch, fout)).astype(np.float32)
    if act == "relu":
        pre.backward(t64(dy * (ys[1].cpu().numpy() > 0), False))   # through the device's ReLU gate
    else:
        fn(pre).backward(t64(dy, False))
    dpre = ops.act_bwd(f32(dy, dev), ys[1], act) if act else f32(dy, dev)
    dw, db = ops.linear_bwd_w(dpre, f32(xs[1], dev))
    _scale_close(dw.cpu().numpy(), wt.grad.numpy())
    _scale_close(db.cpu().numpy(), bt.grad.numpy())
    dx = ops.linear_bwd_x(dpre, f32(ws[1], dev))
    _scale_close(dx.cpu().numpy(), xt.grad.numpy())


def test_mfma_layout_transpose_detecting(dev):
    """A = I with an ASYMMETRIC B: catches a row<->column swap in the MFMA C/D write-out."""
    from deeprl_amd import ops
    n = 64
    x = np.eye(n, dtype=np.float32)
    w = (np.arange(n * n, dtype=np.float32).reshape(n, n) / 7.0)  # w[o][i] asymmetric
    y = ops.linear_fwd([f32(x, dev)], [f32(w, dev)], [None], act=None)[0]
    assert np.array_equal(y.cpu().numpy(), w.T)


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("batch", [32, 1, 5])
def test_conv_koc_fwd_bwd_vs_oracle(dev, layer, batch):
    """One-round-trip forward (conv_v2.hip) and the KOC-layout weight / input gradients."""
    import torch.nn.functional as F
    from deeprl_amd import ops
    c, h, oc, k, s = CONV[layer]
    rs = np.random.RandomState(100 * layer + batch)
    ws = [(rs.standard_normal((oc, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32) for _ in range(3)]
    bs = [(rs.standard_normal(oc) * 0.1).astype(np.float32) for _ in range(3)]
    wts = [ops.to_koc(f32(w, dev)) for w in ws]
    if layer == 1:
        xs_u8 = [rs.randint(0, 256, size=(batch, c, h, h)).astype(np.uint8) for _ in range(3)]
        xs = [NUM.image_normalize_sync(x) for x in xs_u8]
        ys = ops.conv_fwd_koc(1, [cu(x, dev) for x in xs_u8], wts, [f32(b, dev) for b in bs], u8_coef=1.0 / 255)
        ys_f = ops.conv_fwd_koc(1, [f32(x, dev) for x in xs], wts, [f32(b, dev) for b in bs])
        for a_, b_ in zip(ys, ys_f):
            assert torch.equal(a_, b_)  # u8 path normalises bit-exactly, so both inputs agree to the bit
    else:
        xs = [np.maximum(rs.standard_normal((batch, c, h, h)), 0).astype(np.float32) for _ in range(3)]
        ys = ops.conv_fwd_koc(layer, [f32(x, dev) for x in xs], wts, [f32(b, dev) for b in bs])
    refs = []
    for z in range(3):
        xt = t64(xs[z])
        wt, bt = t64(ws[z]), t64(bs[z])
        pre = F.conv2d(xt, wt, bt, stride=s)
        refs.append((xt, wt, bt, pre))
        _scale_close(ys[z].cpu().numpy(), F.relu(pre).detach().numpy())
    xt, wt, bt, pre = refs[2]
    dy = rs.standard_normal(tuple(pre.shape)).astype(np.float32)
    pre.backward(t64(dy * (ys[2].cpu().numpy() > 0), False))      # through the device's ReLU gate (see above)
    dpre = ops.act_bwd(f32(dy, dev), ys[2], "relu")
    x_dev = cu(xs_u8[2], dev) if layer == 1 else f32(xs[2], dev)
    dw_s, db_s = ops.conv_bwd_w_koc(layer, dpre, x_dev, ksplit=16, u8_coef=1.0 / 255 if layer == 1 else None)
    dw = ops.from_koc(dw_s.sum(0).contiguous(), (oc, c, k, k))
    _scale_close(dw.cpu().numpy(), wt.grad.numpy())
    _scale_close(db_s.sum(0).cpu().numpy(), bt.grad.numpy())
    if layer > 1:
        dx = ops.conv_bwd_x_koc(layer, dpre, wts[2])
        _scale_close(dx.cpu().numpy(), xt.grad.numpy())
        dxm = ops.conv_bwd_x_koc(layer, dpre, wts[2], xact=f32(xs[2], dev), act="relu")
        _scale_close(dxm.cpu().numpy(), xt.grad.numpy() * (xs[2] > 0))


@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("batch", [80, 131, 256, 600, 1024])
def test_conv_autograd_function_at_rollout_batches(dev, layer, batch):
    """nets._ConvKocFn -- what NatureConvBody's layers run under autograd in the generic agents (A2C batch 80, PPO minibatch 256,
    and a batch above the former one-slab-per-sample limit of 256) -- forward, weight / bias gradient (one slab per (sample, row
    chunk), folded by the segmented norm kernel) and input gradient against F.conv2d in float64 at 1e-5 of each tensor's scale."""
    import torch.nn.functional as F
    from deeprl_amd import nets, ops
    c, h, oc, k, s = CONV[layer]
    rs = np.random.RandomState(7 * layer + batch)
    w = (rs.standard_normal((oc, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
    b = (rs.standard_normal(oc) * 0.1).astype(np.float32)
    w_dev = ops.to_koc(f32(w, dev)).view(c, k, k, oc).permute(3, 0, 1, 2).requires_grad_(True)   # [OC,C,KH,KW] view of KOC storage
    b_dev = f32(b, dev).requires_grad_(True)
    if layer == 1:
        x_u8 = rs.randint(0, 256, size=(batch, c, h, h)).astype(np.uint8)
        x = NUM.image_normalize_sync(x_u8)
        x_dev, coef = cu(x_u8, dev), 1.0 / 255
    else:
        x = np.maximum(rs.standard_normal((batch, c, h, h)), 0).astype(np.float32)
        x_dev, coef = f32(x, dev).requires_grad_(True), None
    y = nets._ConvKocFn.apply(x_dev, w_dev, b_dev, layer, coef)
    xt, wt, bt = t64(x), t64(w), t64(b)
    pre = F.conv2d(xt, wt, bt, stride=s)
    _scale_close(y.detach().cpu().numpy(), F.relu(pre).detach().numpy())
    dy = rs.standard_normal(tuple(pre.shape)).astype(np.float32)
    y.backward(f32(dy, dev))
    pre.backward(t64(dy * (y.detach().cpu().numpy() > 0), False))          # through the device's ReLU gate
    _scale_close(w_dev.grad.cpu().numpy(), wt.grad.numpy())
    _scale_close(b_dev.grad.cpu().numpy(), bt.grad.numpy())
    if layer > 1:
        _scale_close(x_dev.grad.cpu().numpy(), xt.grad.numpy())


@pytest.mark.parametrize("layer", [2, 3])
@pytest.mark.parametrize("batch", [256, 259, 511, 513, 600, 1024])
def test_conv_input_gradient_scatter_form(dev, layer, batch):
    """DRA_VAR_DGRAD_SCATTER (csrc/dgrad_scatter.h, round 6): conv2 / conv3 input gradient contracted over the OUTPUT positions
    (v_mfma_f32_16x16x4_f32 per tap, col2im by in-order read-add-write into an LDS image of dX) -- against F.conv2d's input
    gradient in float64 at 1e-5 of the tensor's scale, with and without the activation mask; bit-identical on a second run (no
    atomics: every dX element is summed in a fixed order); the weight-gradient slabs of the same call are the gather variant's
    bit for bit (same role); one launch and two.  Batches: the first one the form applies to (PPO's minibatch), an odd one, one either
    side of conv3's samples-per-workgroup switch at 512 (511: one sample per workgroup; 513: two, the last workgroup with one;
    600; 1024).  network_bodies.py:10-33."""
    import torch.nn.functional as F
    from deeprl_amd import ops
    c, h, oc, k, s = CONV[layer]
    rs = np.random.RandomState(31 * layer + batch)
    w = (rs.standard_normal((oc, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
    x = np.maximum(rs.standard_normal((batch, c, h, h)), 0).astype(np.float32)
    oh = (h - k) // s + 1
    dy = rs.standard_normal((batch, oc, oh, oh)).astype(np.float32)
    xt = t64(x)
    F.conv2d(xt, t64(w, False), None, stride=s).backward(t64(dy, False))
    ref = xt.grad.numpy()
    wt_dev, x_dev, dy_dev = ops.to_koc(f32(w, dev)), f32(x, dev), f32(dy, dev)
    base = ops.VAR_FUSED_BWD | ops.VAR_ONESHOT_DGRAD | ops.VAR_ONESHOT_WGRAD
    dw_g, db_g, dx_g, _ = ops.conv_bwd_fused(layer, dy_dev, x_dev, wt=wt_dev, xact=x_dev, ksplit=16, variant=base)
    for var in (base | ops.VAR_DGRAD_SCATTER, (base | ops.VAR_DGRAD_SCATTER) & ~ops.VAR_FUSED_BWD):
        dw_s, db_s, dx_s, _ = ops.conv_bwd_fused(layer, dy_dev, x_dev, wt=wt_dev, xact=x_dev, ksplit=16, variant=var)
        assert not torch.isnan(dx_s).any()          # every element is written
        _scale_close(dx_s.cpu().numpy(), ref * (x > 0))
        assert torch.equal(dw_s, dw_g) and torch.equal(db_s, db_g)
        again = ops.conv_bwd_fused(layer, dy_dev, x_dev, wt=wt_dev, xact=x_dev, ksplit=16, variant=var)[2]
        assert torch.equal(again, dx_s)
    plain = ops.conv_bwd_fused(layer, dy_dev, x_dev, wt=wt_dev, xact=None, ksplit=16, variant=base | ops.VAR_DGRAD_SCATTER)[2]
    _scale_close(plain.cpu().numpy(), ref)
    _scale_close(dx_g.cpu().numpy(), ref * (x > 0))


@pytest.mark.parametrize("layer", [1, 2, 3])
def test_conv_koc_fwd_throughput_shape(dev, layer):
    """conv_v2.hip picks the multi-tile (throughput) workgroup shape from batch 128 up: same arithmetic per
    output position as the one-tile (latency) shape, so the two agree to the bit on a shared prefix of the
    batch, and both match F.conv2d (incl. ragged last tile groups: 400 = 6x64+16, 81 = 96-15, 49 = 64-15)."""
    import torch.nn.functional as F
    from deeprl_amd import ops
    c, h, oc, k, s = CONV[layer]
    batch = 131
    rs = np.random.RandomState(77 + layer)
    w = (rs.standard_normal((oc, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
    b = (rs.standard_normal(oc) * 0.1).astype(np.float32)
    wt = ops.to_koc(f32(w, dev))
    if layer == 1:
        x_u8 = rs.randint(0, 256, size=(batch, c, h, h)).astype(np.uint8)
        x = NUM.image_normalize_sync(x_u8)
        big = ops.conv_fwd_koc(1, [cu(x_u8, dev)], [wt], [f32(b, dev)], u8_coef=1.0 / 255)[0]
        small = ops.conv_fwd_koc(1, [cu(x_u8[:40], dev)], [wt], [f32(b, dev)], u8_coef=1.0 / 255)[0]
    else:
        x = np.maximum(rs.standard_normal((batch, c, h, h)), 0).astype(np.float32)
        big = ops.conv_fwd_koc(layer, [f32(x, dev)], [wt], [f32(b, dev)])[0]
        small = ops.conv_fwd_koc(layer, [f32(x[:40], dev)], [wt], [f32(b, dev)])[0]
    assert torch.equal(big[:40], small)
    ref = F.relu(F.conv2d(t64(x, False), t64(w, False), t64(b, False), stride=s)).numpy()
    _scale_close(big.cpu().numpy(), ref)


@pytest.mark.parametrize("batch", [389, 1025])
def test_conv1_full_k_throughput_kernel(dev, batch):
    """conv1's own throughput kernel (conv_v2.hip conv1_fwd_u8_tp_kernel, uint8 batches >= 384): all of K in a wave's
    registers, the latency shape's four K quarters as four accumulation chains folded (q0 + q1) + (q2 + q3) -- bit-identical
    with the four-wave latency shape (17 ... 127 samples) on a shared prefix / suffix, and F.conv2d within the contraction tolerance.  389: one sample per workgroup
    iteration, an odd batch; 1025: two samples per iteration, a second iteration, a last group of one sample."""
    import torch.nn.functional as F
    from deeprl_amd import ops
    c, h, oc, k, s = CONV[1]
    rs = np.random.RandomState(batch)
    w = (rs.standard_normal((oc, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
    b = (rs.standard_normal(oc) * 0.1).astype(np.float32)
    wt = ops.to_koc(f32(w, dev))
    x_u8 = rs.randint(0, 256, size=(batch, c, h, h)).astype(np.uint8)
    big = ops.conv_fwd_koc(1, [cu(x_u8, dev)], [wt], [f32(b, dev)], u8_coef=1.0 / 255)[0]
    small = ops.conv_fwd_koc(1, [cu(x_u8[:33], dev)], [wt], [f32(b, dev)], u8_coef=1.0 / 255)[0]
    # (19 samples: batches of <= 16 run the EIGHT-wave latency shape from round 5 on -- another summation tree)
    tail = ops.conv_fwd_koc(1, [cu(x_u8[-19:], dev)], [wt], [f32(b, dev)], u8_coef=1.0 / 255)[0]
    assert torch.equal(big[:33], small) and torch.equal(big[-19:], tail)
    keep = np.r_[0:40, batch - 40:batch]
    x = NUM.image_normalize_sync(x_u8[keep])
    ref = F.relu(F.conv2d(t64(x, False), t64(w, False), t64(b, False), stride=s)).numpy()
    _scale_close(big.cpu().numpy()[keep], ref)


# ---------------------------------------------------------------- fused launches + one-pass kernels (fused.hip)
@pytest.mark.parametrize("layer", [1, 2, 3])
@pytest.mark.parametrize("batch", [32, 1, 5])
@pytest.mark.parametrize("variant", [1, 3, 9, 11])
def test_conv_bwd_fused_vs_autograd(dev, layer, batch, variant):
    """dra_conv_bwd_fused: weight/bias gradient slabs and the masked input gradient of one layer in ONE launch,
    for the K-chunked (1), one-pass dgrad (3), one-pass wgrad (9) and all-one-pass (11) variants, against
    F.conv2d autograd; the slab fold goes through dra_grad_sqnorm_segs."""
    import torch.nn.functional as F
    from deeprl_amd import ops
    c, h, oc, k, s = CONV[layer]
    rs = np.random.RandomState(1000 * layer + 10 * batch + variant)
    w = (rs.standard_normal((oc, c, k, k)) / np.sqrt(c * k * k)).astype(np.float32)
    wt_dev = ops.to_koc(f32(w, dev))
    if layer == 1:
        x_u8 = rs.randint(0, 256, size=(batch, c, h, h)).astype(np.uint8)
        x = NUM.image_normalize_sync(x_u8)
        x_dev = cu(x_u8, dev)
    else:
        x = np.maximum(rs.standard_normal((batch, c, h, h)), 0).astype(np.float32)
        x_dev = f32(x, dev)
    xt, wt = t64(x), t64(w)
    bt = torch.zeros(oc, dtype=torch.float64, requires_grad=True)
    yt = F.conv2d(xt, wt, bt, stride=s)
    dy = rs.standard_normal(tuple(yt.shape)).astype(np.float32)
    yt.backward(t64(dy, False))
    dw_s, db_s, dx, slab_buf = ops.conv_bwd_fused(layer, f32(dy, dev), x_dev, wt=wt_dev if layer > 1 else None,
                                                  xact=x_dev if layer > 1 else None, ksplit=16,
                                                  u8_coef=1.0 / 255 if layer == 1 else None, variant=variant)
    assert not torch.isnan(dw_s).any() and not torch.isnan(db_s).any()      # every slab element is written
    # fold the slabs with the segmented norm kernel: grad segment = [w | b]
    kk = c * k * k
    stride = dw_s.stride(0)
    n_slabs = dw_s.shape[0]
    if variant & 8:     # one-pass weight gradient: one slab per (sample, row chunk); conv1 has 5 chunks of 4 output rows
        assert n_slabs == batch * (5 if layer == 1 else 1)
    seg = oc * kk + oc
    grad = torch.zeros(seg + 1000, dtype=torch.float32, device=dev)
    tail = rs.standard_normal(1000).astype(np.float32)
    grad[seg:] = f32(tail, dev)
    partials = torch.zeros(ops.norm_partials_max(), dtype=torch.float64, device=dev)
    n_part = ops.grad_sqnorm_segs(grad, [(0, seg, slab_buf, stride, n_slabs)], partials)
    g = grad.cpu().numpy()
    dw = ops.from_koc(grad[:oc * kk].contiguous(), (oc, c, k, k)).cpu().numpy()
    _scale_close(dw, wt.grad.numpy())
    _scale_close(g[oc * kk:seg], bt.grad.numpy())
    assert np.array_equal(g[seg:], tail)
    np.testing.assert_allclose(partials[:n_part].sum().item(), (g.astype(np.float64) ** 2).sum(), rtol=1e-6)
    if layer > 1:
        assert not torch.isnan(dx).any()
        _scale_close(dx.cpu().numpy(), xt.grad.numpy() * (x > 0))


@pytest.mark.parametrize("batch", [32, 5, 1, 64])
@pytest.mark.parametrize("variant", [0, 2])
def test_fc_bwd_fused_vs_autograd(dev, batch, variant):
    import torch.nn.functional as F
    from deeprl_amd import ops
    rs = np.random.RandomState(batch + variant)
    a, fin = 4, 3136
    x3 = np.maximum(rs.standard_normal((batch, fin)), 0).astype(np.float32)
    w4 = (rs.standard_normal((512, fin)) / np.sqrt(fin)).astype(np.float32)
    b4 = (rs.standard_normal(512) * 0.1).astype(np.float32)
    wh = (rs.standard_normal((a, 512)) / np.sqrt(512)).astype(np.float32)
    dq = rs.standard_normal((batch, a)).astype(np.float32)
    x3t, w4t, b4t = t64(x3), t64(w4), t64(b4)
    wht, bht = t64(wh), torch.zeros(a, dtype=torch.float64, requires_grad=True)
    h4 = F.relu(F.linear(x3t, w4t, b4t))
    q = F.linear(h4, wht, bht)
    q.backward(t64(dq, False))
    h4n = h4.detach().numpy()
    dh4 = (dq @ wh) * (h4n > 0)
    dwh, dbh, dw4, db4, dx3 = ops.fc_bwd_fused(f32(dq, dev), f32(h4n, dev), f32(dh4.astype(np.float32), dev), f32(x3, dev),
                                               f32(w4, dev), variant=variant)
    for t in (dwh, dbh, dw4, db4, dx3):
        assert not torch.isnan(t).any()
    _scale_close(dwh.cpu().numpy(), wht.grad.numpy())
    _scale_close(dbh.cpu().numpy(), bht.grad.numpy())
    _scale_close(dw4.cpu().numpy(), w4t.grad.numpy())
    _scale_close(db4.cpu().numpy(), b4t.grad.numpy())
    _scale_close(dx3.cpu().numpy(), x3t.grad.numpy() * (x3 > 0))


@pytest.mark.parametrize("batch", [32, 1, 7])
def test_linear_fwd_slabs_one_pass(dev, batch):
    from deeprl_amd import ops
    rs = np.random.RandomState(batch)
    xs = [np.maximum(rs.standard_normal((batch, 3136)), 0).astype(np.float32) for _ in range(2)]
    ws = [(rs.standard_normal((512, 3136)) / 56.0).astype(np.float32) for _ in range(2)]
    one = ops.linear_fwd_slabs([f32(x, dev) for x in xs], [f32(w, dev) for w in ws], 8, one_pass=True)
    old = ops.linear_fwd_slabs([f32(x, dev) for x in xs], [f32(w, dev) for w in ws], 8, one_pass=False)
    assert not torch.isnan(one).any()
    for z in range(2):
        want = xs[z].astype(np.float64) @ ws[z].astype(np.float64).T
        _scale_close(one[z].sum(0).cpu().numpy(), want)
        _scale_close(old[z].sum(0).cpu().numpy(), want)
        # (the two kernels partition K differently -- contiguous eighths vs interleaved chunks -- so only the
        #  slab SUM is comparable, which is what head_fused_kernel consumes)


def test_grad_sqnorm_segs_many_slabs(dev):
    """Three contiguous segments with 160 / 32 / 5 slabs and a plain tail, against a numpy fold in the kernel's
    order (16 slab groups of stride 16, then group order)."""
    from deeprl_amd import ops
    rs = np.random.RandomState(4)
    counts, nsl = [8224, 32832, 36928], [160, 32, 5]
    tail = 50_000
    n = sum(counts) + tail
    grad = f32(rs.standard_normal(n).astype(np.float32), dev)
    tail_np = grad[sum(counts):].cpu().numpy().copy()
    segs, want, off = [], [], 0
    for cnt, ns in zip(counts, nsl):
        sl = rs.standard_normal((ns, cnt)).astype(np.float32)
        t = f32(sl.reshape(-1), dev)
        segs.append((off, cnt, t, cnt, ns))
        groups = []
        for g in range(min(16, ns)):
            acc = np.zeros(cnt, dtype=np.float32)
            for s_ in range(g, ns, 16):
                acc = acc + sl[s_]
            groups.append(acc)
        while len(groups) < 16:
            groups.append(np.zeros(cnt, dtype=np.float32))
        r = groups[0].copy()
        for g in range(1, 16):
            r = r + groups[g]
        want.append(r)
        off += cnt
    partials = torch.zeros(ops.norm_partials_max(), dtype=torch.float64, device=dev)
    n_part = ops.grad_sqnorm_segs(grad, segs, partials)
    got = grad.cpu().numpy()
    off = 0
    for cnt, w_ in zip(counts, want):
        assert np.array_equal(got[off:off + cnt], w_)
        off += cnt
    assert np.array_equal(got[off:], tail_np)
    np.testing.assert_allclose(partials[:n_part].sum().item(), (got.astype(np.float64) ** 2).sum(), rtol=1e-6)




@pytest.mark.parametrize("b,a", [(1, 2), (16, 4), (80, 18), (1024, 6), (300, 64)])
def test_categorical_policy_kernels_vs_torch(dev, b, a):
    """K12 (network_heads.py:249-254): log_prob / entropy of Categorical(logits) and their gradient w.r.t. the logits against
    torch.distributions on the CPU in fp32 (absolute 2e-6 on values of magnitude <= log(A)); sampled actions are the inverse
    CDF of the given uniforms."""
    from deeprl_amd import nets, ops
    rs = np.random.RandomState(b * 131 + a)
    logits = (rs.standard_normal((b, a)) * 3).astype(np.float32)
    action = rs.randint(0, a, size=b).astype(np.int64)
    g_lp, g_ent = rs.standard_normal(b).astype(np.float32), rs.standard_normal(b).astype(np.float32)
    x = torch.from_numpy(logits).requires_grad_(True)
    dist = torch.distributions.Categorical(logits=x)
    lp_ref, ent_ref = dist.log_prob(torch.from_numpy(action)), dist.entropy()
    torch.autograd.backward([lp_ref, ent_ref], [torch.from_numpy(g_lp), torch.from_numpy(g_ent)])
    xd = torch.from_numpy(logits).to(dev).requires_grad_(True)
    act_out, lp, ent = nets.categorical_policy(xd, torch.from_numpy(action).to(dev))
    torch.autograd.backward([lp, ent], [torch.from_numpy(g_lp).to(dev).unsqueeze(-1), torch.from_numpy(g_ent).to(dev).unsqueeze(-1)])
    assert np.array_equal(act_out.cpu().numpy(), action)
    np.testing.assert_allclose(lp.detach().cpu().numpy()[:, 0], lp_ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ent.detach().cpu().numpy()[:, 0], ent_ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), x.grad.numpy(), rtol=1e-5, atol=2e-6)
    # sampling: action = first index whose cumulative probability exceeds u
    u = rs.rand(b).astype(np.float32)
    got, lp_s, _ = ops.categorical_fwd(torch.from_numpy(logits).to(dev), uniform=torch.from_numpy(u).to(dev))
    p = torch.softmax(torch.from_numpy(logits).double(), dim=-1).numpy()
    cum = np.cumsum(p, axis=1)
    got = got.cpu().numpy()
    for i in range(b):
        lo = cum[i, got[i] - 1] if got[i] > 0 else 0.0
        assert lo - 1e-6 <= u[i] <= cum[i, got[i]] + 1e-6 or got[i] == a - 1, (i, got[i], u[i])
    np.testing.assert_allclose(lp_s.cpu().numpy(), np.log(p[np.arange(b), got]), rtol=1e-5, atol=3e-6)


@pytest.mark.parametrize("b,k,o0,o1", [(16, 512, 4, 1), (8, 512, 18, 1), (1, 512, 6, 1), (33, 64, 3, 2), (128, 400, 9, 9), (5, 17, 1, 1)])
def test_linear_fwd_pair_equals_two_layers(dev, b, k, o0, o1):
    """dra_linear_fwd_pair (both heads of the actor-critic nets on the shared features in one launch, used in rollout steps)
    == dra_linear_fwd of each head on its own (what the update's forward runs through the two Linear modules): bit for bit."""
    from deeprl_amd import ops
    rs = np.random.RandomState(b + 3 * k + 5 * o0 + 7 * o1)
    x = f32(rs.standard_normal((b, k)).astype(np.float32), dev)
    w0, w1 = f32(rs.standard_normal((o0, k)).astype(np.float32), dev), f32(rs.standard_normal((o1, k)).astype(np.float32), dev)
    b0, b1 = f32(rs.standard_normal(o0).astype(np.float32), dev), f32(rs.standard_normal(o1).astype(np.float32), dev)
    y0, y1 = ops.linear_fwd_pair(x, w0, b0, w1, b1)
    r0 = ops.linear_fwd([x], [w0], [b0])[0]
    r1 = ops.linear_fwd([x], [w1], [b1])[0]
    assert torch.equal(y0, r0) and torch.equal(y1, r1)
    ref = (x.double().cpu() @ w0.double().cpu().T + b0.double().cpu()).numpy()
    _scale_close(y0.cpu().numpy(), ref)


@pytest.mark.parametrize("b,k,a", [(16, 512, 4), (8, 512, 18), (1, 512, 6), (33, 64, 3), (128, 400, 64), (5, 17, 1), (1024, 512, 9)])
def test_policy_heads_sample_equals_heads_then_categorical(dev, b, k, a):
    """dra_policy_heads_sample (a rollout step's two heads + Categorical sample / log-probability / entropy in ONE launch)
    == dra_linear_fwd_pair followed by dra_categorical_fwd on the same uniforms: every output bit for bit, into caller-owned
    rows as well as freshly allocated ones; and the sampled action is the inverse-CDF action of the float64 softmax."""
    from deeprl_amd import ops
    rs = np.random.RandomState(b + 3 * k + 5 * a)
    x = f32(rs.standard_normal((b, k)).astype(np.float32), dev)
    w0, w1 = f32(0.2 * rs.standard_normal((a, k)).astype(np.float32), dev), f32(rs.standard_normal((1, k)).astype(np.float32), dev)
    b0, b1 = f32(rs.standard_normal(a).astype(np.float32), dev), f32(rs.standard_normal(1).astype(np.float32), dev)
    u = f32(rs.uniform(size=b).astype(np.float32), dev)
    logits, v = ops.linear_fwd_pair(x, w0, b0, w1, b1)
    act, lp, ent = ops.categorical_fwd(logits, uniform=u)
    got = ops.policy_heads_sample(x, w0, b0, w1, b1, u)
    rows = (torch.full((3, b), -1, dtype=torch.int64, device=dev), torch.zeros((3, b), device=dev), torch.zeros((3, b), device=dev),
            torch.zeros((3, b), device=dev))
    got2 = ops.policy_heads_sample(x, w0, b0, w1, b1, u, tuple(r[1] for r in rows))
    for g in (got, got2):
        assert torch.equal(g[0], act.reshape(-1).long()) and torch.equal(g[1], lp.reshape(-1))
        assert torch.equal(g[2], ent.reshape(-1)) and torch.equal(g[3], v.reshape(-1))
    assert all(torch.equal(r[1], g) for r, g in zip(rows, got)) and int(rows[0][0].max()) == -1 and float(rows[3][2].abs().max()) == 0
    p = torch.softmax(logits.double(), dim=1).cpu().numpy()
    cum = np.cumsum(p, axis=1)
    a_np, u_np = got[0].cpu().numpy(), u.cpu().numpy()
    for i in range(b):
        lo = cum[i, a_np[i] - 1] if a_np[i] > 0 else 0.0
        assert lo - 1e-6 <= u_np[i] <= cum[i, a_np[i]] + 1e-6 or a_np[i] == a - 1


@pytest.mark.parametrize("b,k,a,relu", [(80, 512, 4, True), (256, 512, 18, True), (1, 512, 6, False), (33, 64, 3, True), (5, 17, 1, False),
                                        (1000, 512, 9, True)])
def test_policy_heads_given_and_backward_equal_the_separate_launches(dev, b, k, a, relu):
    """dra_policy_heads_given == dra_linear_fwd_pair + dra_categorical_fwd(action) and dra_policy_heads_bwd == dra_categorical_bwd +
    dra_linear_bwd_pair (+ dra_act_bwd's ReLU mask on the input gradient): every output bit for bit -- the fused launches keep
    the separate kernels' sums in their order -- also with missing gradients (None = zero) and caller-owned gradient buffers."""
    from deeprl_amd import ops
    rs = np.random.RandomState(7 * b + k + a)
    x = f32(np.maximum(rs.standard_normal((b, k)), 0).astype(np.float32), dev)         # a ReLU output: about half zeros
    w0, w1 = f32(0.2 * rs.standard_normal((a, k)).astype(np.float32), dev), f32(rs.standard_normal((1, k)).astype(np.float32), dev)
    b0, b1 = f32(rs.standard_normal(a).astype(np.float32), dev), f32(rs.standard_normal(1).astype(np.float32), dev)
    act = torch.from_numpy(rs.randint(0, a, size=b)).to(dev)
    logits, v = ops.linear_fwd_pair(x, w0, b0, w1, b1)
    _, lp, ent = ops.categorical_fwd(logits, action=act)
    lp2, ent2, v2, logits2 = ops.policy_heads_given(x, w0, b0, w1, b1, act)
    assert torch.equal(lp2, lp.reshape(-1)) and torch.equal(ent2, ent.reshape(-1)) and torch.equal(v2, v.reshape(-1))
    assert torch.equal(logits2, logits)
    g_lp, g_ent, g_v = [f32(rs.standard_normal(b).astype(np.float32), dev) for _ in range(3)]
    for use in ((True, True, True), (True, True, False), (False, False, True)):
        gl, ge, gv = [g if u else None for g, u in zip((g_lp, g_ent, g_v), use)]
        zero = torch.zeros(b, device=dev)
        dlogits = ops.categorical_bwd(logits, act, gl if gl is not None else zero, ge if ge is not None else zero)
        rdx, rdw0, rdb0, rdw1, rdb1 = ops.linear_bwd_pair(dlogits, (gv if gv is not None else zero).reshape(b, 1), x, w0, w1)
        if relu:
            rdx = ops.act_bwd(rdx, x, "relu")
        own = (torch.full((a, k), 7.0, device=dev), torch.full((a,), 7.0, device=dev), torch.full((1, k), 7.0, device=dev),
               torch.full((1,), 7.0, device=dev))
        dx, dw0, db0, dw1, db1 = ops.policy_heads_bwd(logits2, act, gl, ge, gv, x, w0, w1, *own, relu_mask=relu)
        assert dw0 is own[0] and db1 is own[3]
        assert torch.equal(dx, rdx), float((dx - rdx).abs().max())
        assert torch.equal(dw0, rdw0) and torch.equal(db0, rdb0) and torch.equal(dw1, rdw1) and torch.equal(db1, rdb1)
    assert ops.policy_heads_bwd(logits2, act, g_lp, g_ent, g_v, x, w0, w1, want_dx=False)[0] is None


@pytest.mark.parametrize("b,relu", [(80, True), (256, True), (33, False), (1, True)])
def test_linear_bwd_xw_512_equals_the_two_launches(dev, b, relu):
    """dra_linear_bwd_xw_one512 (fc4's input gradient and weight / bias gradient as two roles of one launch) == dra_linear_bwd_x +
    dra_linear_bwd_w, bit for bit."""
    from deeprl_amd import ops
    rs = np.random.RandomState(b)
    x = f32(np.maximum(rs.standard_normal((b, 3136)), 0).astype(np.float32), dev)
    w = f32(0.05 * rs.standard_normal((512, 3136)).astype(np.float32), dev)
    dy = f32(rs.standard_normal((b, 512)).astype(np.float32), dev)
    rdx = ops.linear_bwd_x(dy, w, xact=x if relu else None, act="relu" if relu else None)
    rdw, rdb = ops.linear_bwd_w(dy, x)
    dx, dw, db = ops.linear_bwd_xw_512(dy, x, w, relu)
    assert torch.equal(dx, rdx) and torch.equal(dw, rdw) and torch.equal(db, rdb)


@pytest.mark.parametrize("b,a", [(8, 4), (16, 18), (1, 6), (32, 9), (5, 64)])
def test_fc4_policy_heads_sample_folds_the_28_slices_like_the_finish_kernel(dev, b, a):
    """ops.fc4_policy_heads_sample (a rollout step's fc4 through the one-pass 28-slice kernel, its finish inside the head launch):
    the head's outputs equal dra_policy_heads_sample on phi = relu(b4 + slab 0 + slab 1 + ... + slab 27) formed in that order in
    fp32 -- bit for bit -- and phi agrees with the float64 product at 1e-5 of its scale."""
    from deeprl_amd import ops
    from deeprl_amd._lib import lib, stream_ptr
    rs = np.random.RandomState(b + 7 * a)
    y3 = f32(np.maximum(rs.standard_normal((b, 3136)), 0).astype(np.float32), dev)
    w4, b4 = f32(0.03 * rs.standard_normal((512, 3136)).astype(np.float32), dev), f32(rs.standard_normal(512).astype(np.float32), dev)
    w0, w1 = f32(0.2 * rs.standard_normal((a, 512)).astype(np.float32), dev), f32(rs.standard_normal((1, 512)).astype(np.float32), dev)
    b0, b1 = f32(rs.standard_normal(a).astype(np.float32), dev), f32(rs.standard_normal(1).astype(np.float32), dev)
    u = f32(rs.uniform(size=b).astype(np.float32), dev)
    got = ops.fc4_policy_heads_sample(y3, w4, b4, w0, b0, w1, b1, u)
    slabs = ops.linear_fwd_slabs([y3], [w4], ksplit=28, one_pass=True)[0]
    phi = slabs[0].clone()
    for s in range(1, 28):
        phi = phi + slabs[s]
    phi = torch.relu(phi + b4)
    ref = ops.policy_heads_sample(phi, w0, b0, w1, b1, u)
    for g, r in zip(got, ref):
        assert torch.equal(g, r)
    _scale_close(phi.cpu().numpy(), np.maximum(y3.double().cpu().numpy() @ w4.double().cpu().numpy().T + b4.double().cpu().numpy(), 0))


@pytest.mark.parametrize("rows,n", [(1024, 256), (80, 16), (7, 7), (2048, 1)])
def test_gather_rows_equals_indexing(dev, rows, n):
    """dra_gather_rows (the five fields of a PPO minibatch -- uint8 frame stacks, int64 actions, three float columns -- by one
    index vector in one launch) == x[idx] field by field, bit for bit; odd row sizes and negative indices included."""
    from deeprl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(rows + n)
    fields = [torch.randint(0, 256, (rows, 4, 84, 84), dtype=torch.uint8, generator=g), torch.randint(0, 18, (rows,), generator=g),
              torch.randn((rows, 1), generator=g), torch.randn((rows, 1), generator=g), torch.randn((rows, 3, 5), generator=g),
              torch.randint(0, 256, (rows, 13), dtype=torch.uint8, generator=g)]
    fields = [f.to(dev) for f in fields]
    idx = torch.randint(-rows, rows, (n,), generator=g).to(dev)
    got = ops.gather_rows(fields, idx)
    for f, o in zip(fields, got):
        assert o.dtype == f.dtype and torch.equal(o, f[idx])


@pytest.mark.parametrize("b,k,o0,o1", [(80, 512, 4, 1), (256, 512, 18, 1), (1, 512, 6, 1), (33, 64, 3, 2), (5, 17, 1, 1)])
def test_linear_bwd_pair_vs_fp64(dev, b, k, o0, o1):
    """dra_linear_bwd_pair (input gradient and both layers' weight / bias gradients of the paired heads in one launch) against
    float64: d x = g0 W0 + g1 W1, dW_h = g_h^T x, db_h = column sums, each at 1e-5 of its scale."""
    from deeprl_amd import ops
    rs = np.random.RandomState(11 * b + k + o0 + o1)
    x, g0, g1 = [rs.standard_normal(sh).astype(np.float32) for sh in ((b, k), (b, o0), (b, o1))]
    w0, w1 = rs.standard_normal((o0, k)).astype(np.float32), rs.standard_normal((o1, k)).astype(np.float32)
    dx, dw0, db0, dw1, db1 = ops.linear_bwd_pair(f32(g0, dev), f32(g1, dev), f32(x, dev), f32(w0, dev), f32(w1, dev))
    X, G0, G1, W0, W1 = [a.astype(np.float64) for a in (x, g0, g1, w0, w1)]
    _scale_close(dx.cpu().numpy(), G0 @ W0 + G1 @ W1)
    _scale_close(dw0.cpu().numpy(), G0.T @ X)
    _scale_close(dw1.cpu().numpy(), G1.T @ X)
    _scale_close(db0.cpu().numpy(), G0.sum(0))
    _scale_close(db1.cpu().numpy(), G1.sum(0))


@pytest.mark.parametrize("b,k,o,act", [(16, 512, 4, None), (16, 512, 1, None), (80, 512, 18, None), (1, 17, 64, "tanh"),
                                      (64, 64, 64, "relu"), (33, 400, 300, "relu"), (128, 512, 204, None), (5, 3, 2, None),
                                      (8, 3136, 512, "relu"), (16, 3136, 512, "relu"), (5, 1024, 37, None), (9, 4096, 16, "tanh"),
                                      (32, 2052, 6, None), (80, 3136, 512, "relu"), (256, 3136, 512, "relu"), (33, 3136, 7, None)])
def test_linear_small_layers_vs_torch(dev, b, k, o, act):
    """The one-pass forwards behind dra_linear_fwd -- in_features <= 512, batch <= 128 (heads and FCBody layers), and the wide
    GEMV for 512 < in_features <= 4096 at batch <= 32 (fc4 of NatureConvBody at rollout batch sizes) -- against F.linear in
    fp32 on the CPU: 1e-5 relative to the operand scale sum|x||w|."""
    from deeprl_amd import ops
    rs = np.random.RandomState(b + 7 * k + 13 * o)
    x = rs.standard_normal((b, k)).astype(np.float32)
    w = (rs.standard_normal((o, k)) / np.sqrt(k)).astype(np.float32)
    bias = rs.standard_normal(o).astype(np.float32)
    want = torch.nn.functional.linear(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(bias))
    if act == "relu":
        want = torch.relu(want)
    elif act == "tanh":
        want = torch.tanh(want)
    got = ops.linear_fwd([torch.from_numpy(x).to(dev)], [torch.from_numpy(w).to(dev)], [torch.from_numpy(bias).to(dev)], act=act)[0]
    scale = float((np.abs(x) @ np.abs(w).T).max())
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5 * scale + 1e-6)


def test_atari_preprocess_kernel_equals_oracle(dev):
    """csrc/preproc.hip (max of two raw frames, RGB2GRAY, INTER_AREA resize 210x160 -> 84x84; envs.py:39-47 via baselines'
    MaxAndSkipEnv / WarpFrame) against oracle/preproc_oracle.py: BIT-EXACT (same fp32 operation order), including the axis
    tables dra_resize_area_tab builds on the host.  Parity unpinned by the reference (cv2 / baselines are not installed)."""
    from deeprl_amd import ops
    from oracle import preproc_oracle as P
    rs = np.random.RandomState(5)
    raw = rs.randint(0, 256, size=(3, 2, 210, 160, 3)).astype(np.uint8)
    raw[2, 0] = 0
    raw[2, 1, ::2] = 255
    pre = ops.AtariPreprocess()
    for (si, al, off), (ssize, dsize) in zip((pre.tabs[0:3], pre.tabs[3:6]), ((160, 84), (210, 84))):
        tab = P.resize_area_tab(ssize, dsize)
        flat = [e for row in tab for e in row]
        assert si.cpu().tolist() == [s for s, _ in flat]
        assert np.array_equal(al.cpu().numpy(), np.asarray([a for _, a in flat], dtype=np.float32))
        assert off.cpu().tolist() == list(np.cumsum([0] + [len(r) for r in tab]))
    got = pre(raw).cpu().numpy()
    assert np.array_equal(got, P.atari_preprocess(raw))
