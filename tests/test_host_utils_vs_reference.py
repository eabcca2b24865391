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
