"""GPU parity at BASELINE.json's FULL sizes through size-independent properties, plus the edge cases of the
replay semantics: the 1M-frame / 7 GB ring cannot be mirrored by the python-list oracle
in seconds, so the checks are (i) every gathered frame is the counter-hash frame of its slot (the oracle's
generator evaluated only for the slots touched), (ii) state / next_state overlap, (iii) the fp64 n-step fold
against a numpy loop, (iv) the sum tree's root equals the exact fp64 sum of its leaves and descents agree with
the numpy restatement of sum_tree.py on the same 1M-leaf heap."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.synth_oracle import synth_transitions  # noqa: E402
from oracle.sumtree_oracle import SumTreeOracle  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need an MI355X")
    from deeprl_amd.support import select_device, Config
    select_device(0)
    return Config.DEVICE


def _frames_of(slots, seed):
    """counter-hash frames of the given ring slots (the ring was filled with counter == slot)."""
    slots = np.asarray(slots, dtype=np.int64)
    out = np.empty((len(slots), 7056), dtype=np.uint8)
    for i, s in enumerate(slots):
        out[i] = synth_transitions(int(s), 1, 7056, seed=seed)[0][0]
    return out


@pytest.mark.parametrize("n_step,b", [(1, 256), (3, 256), (1, 4096), (3, 2048)])
def test_full_size_ring_gather_properties(dev, n_step, b):
    """BASELINE configs[1]/[3]: 1 000 000-frame ring of 84x84 uint8 frames, H = 4, batch 32 x 8 minibatches and
    32 x 128 / 32 x 64 (the many-minibatch launches the gather roofline is quoted on), indices that include the first /
    last valid slots on both sides of the write head."""
    from deeprl_amd import ops
    cap, h, gamma, seed = 1_000_000, 4, 0.99, 11
    ring = ops.Ring(cap, 7056, 8, h, n_step, gamma)
    ring.fill_synthetic(0, cap, 0, seed, n_actions=4, done_period=37)   # frequent terminals: the mask chain matters
    pos = 123_457                                                       # write head: samples must not straddle it
    rs = np.random.RandomState(n_step)
    edge = [h - 1, pos - n_step - 1, pos + h - 1, cap - n_step - 1]     # replay.py:105-110 boundaries (all valid)
    cand = rs.randint(0, cap, size=4 * b)
    ok = ((cand - h + 1 >= 0) & (cand + n_step < pos)) | ((cand - h + 1 >= pos) & (cand + n_step < cap))
    idx = np.concatenate([edge, cand[ok]])[:b].astype(np.int64)
    got = ring.gather(torch.from_numpy(idx).to(dev), (84, 84), torch.uint8, torch.int64, want_f32=True)
    torch.cuda.synchronize()
    st = got["state"].cpu().numpy().reshape(b, h, 7056)
    ns = got["next_state"].cpu().numpy().reshape(b, h, 7056)
    # (i) every frame is the frame of its slot; checked exhaustively for 40 samples incl. the edge ones
    for k in list(range(8)) + list(rs.randint(8, b, size=32)):
        want = _frames_of(np.arange(idx[k] - h + 1, idx[k] + n_step + 1), seed)
        assert np.array_equal(st[k], want[:h]), k
        assert np.array_equal(ns[k], want[n_step:n_step + h]), k
    # (ii) overlap: next_state[j] == state[j + n] for every sample
    if n_step < h:
        assert np.array_equal(ns[:, :h - n_step], st[:, n_step:])
    # (iii) action / n-step reward / mask against the numpy restatement of replay.py:133-139 on the touched slots
    _, act, rew, msk = synth_transitions(0, 1, 7056, seed=seed, n_actions=4, done_period=37)
    want_r, want_m, want_a = np.empty(b), np.empty(b, dtype=np.int32), np.empty(b, dtype=np.int64)
    for k in range(b):
        _, a_k, r_k, m_k = synth_transitions(int(idx[k]), n_step, 7056, seed=seed, n_actions=4, done_period=37)
        cum_r, cum_m = 0.0, 1
        for j in reversed(range(n_step)):
            cum_r = r_k[j] + m_k[j] * gamma * cum_r
            cum_m = cum_m and m_k[j]
        want_r[k], want_m[k], want_a[k] = cum_r, cum_m, a_k[0]
    assert np.array_equal(got["reward"].cpu().numpy(), want_r)
    assert np.array_equal(got["mask"].cpu().numpy(), want_m)
    assert np.array_equal(got["action"].cpu().numpy(), want_a)
    assert np.array_equal(got["reward_f32"].cpu().numpy(), want_r.astype(np.float32))
    ring.close()


@pytest.mark.parametrize("frame_bytes,action_bytes,h,n", [(32, 8, 1, 1), (24, 8, 2, 2), (7056, 8, 4, 1), (40, 48, 1, 3)])
def test_ring_edge_shapes(dev, frame_bytes, action_bytes, h, n):
    """Ragged shapes: frames that are not a multiple of 16 bytes (scalar copy path), multi-word actions, batch 1,
    capacity barely larger than one sample, the first and the last valid index."""
    from deeprl_amd import ops
    cap = h + n + 2
    ring = ops.Ring(cap, frame_bytes, action_bytes, h, n, 0.9)
    rs = np.random.RandomState(frame_bytes)
    frames = rs.randint(0, 256, size=(cap, frame_bytes)).astype(np.uint8)
    actions = rs.randint(0, 256, size=(cap, action_bytes)).astype(np.uint8)
    rew = rs.standard_normal(cap)
    msk = (rs.rand(cap) > 0.3).astype(np.int32)
    for s in range(cap):
        ring.put_host(s, frames[s], actions[s], float(rew[s]), int(msk[s]))
    for i in (h - 1, cap - n - 1):   # ring full, pos == 0: valid <=> i-h+1 >= 0 and i+n < cap
        got = ring.gather(torch.tensor([i], dtype=torch.int64, device=dev), (frame_bytes,), torch.uint8, torch.uint8)
        torch.cuda.synchronize()
        assert np.array_equal(got["state"].cpu().numpy().reshape(h, frame_bytes), frames[i - h + 1:i + 1])
        assert np.array_equal(got["next_state"].cpu().numpy().reshape(h, frame_bytes), frames[i - h + 1 + n:i + n + 1])
        assert np.array_equal(got["action"].cpu().numpy().reshape(-1), actions[i])
        cum_r, cum_m = 0.0, 1
        for j in reversed(range(n)):
            cum_r = rew[i + j] + msk[i + j] * 0.9 * cum_r
            cum_m = cum_m and msk[i + j]
        assert got["reward"].cpu().numpy()[0] == cum_r and got["mask"].cpu().numpy()[0] == cum_m
    ring.close()


def test_full_size_sumtree_properties(dev):
    """1M-leaf tree (BASELINE configs[3]): fp32-valued priorities -> every internal node is exact, so the
    root equals math.fsum of the leaves, a bottom-up rebuild equals the incrementally updated heap, and
    stratified descents (incl. u = 0 and u -> total) return the leaf the numpy restatement returns."""
    import math
    from deeprl_amd import ops
    cap = 1_000_000
    tree = ops.SumTree(cap)
    view = tree.as_tensor()
    rs = np.random.RandomState(3)
    leaves = (rs.rand(cap).astype(np.float32) * 3 + 0.1).astype(np.float64)
    view[cap - 1:] = torch.from_numpy(leaves).to(dev)
    tree.rebuild()
    # 64 rounds of 32 parallel updates (distinct leaves), then compare with a rebuild of the same leaves
    for r in range(64):
        li = rs.choice(cap, 32, replace=False).astype(np.int64)
        pr = (rs.rand(32).astype(np.float32) * 5 + 0.1).astype(np.float64)
        leaves[li] = pr
        tree.update(torch.from_numpy(li + cap - 1).to(dev), torch.from_numpy(pr).to(dev))
    torch.cuda.synchronize()
    heap = view.cpu().numpy().copy()
    assert heap[0] == math.fsum(leaves)
    tree.rebuild()
    torch.cuda.synchronize()
    assert np.array_equal(view.cpu().numpy(), heap)
    orc = SumTreeOracle(cap)
    orc.tree[:] = heap
    total = heap[0]
    u = rs.rand(32)
    u[0], u[31] = 0.0, np.nextafter(1.0, 0.0)          # first leaf of stratum 0, last reachable point of stratum 31
    seg = total / 32
    s_host = [seg * i + (seg * (i + 1) - seg * i) * u[i] for i in range(32)]   # random.uniform(a, b) = a + (b-a)*random()
    idx_d, p_d, tot_d = tree.sample(torch.from_numpy(u).to(dev))
    torch.cuda.synchronize()
    want = [orc.get(float(x)) for x in s_host]
    assert np.array_equal(idx_d.cpu().numpy(), np.asarray([w[0] for w in want], dtype=np.int64))
    assert np.array_equal(p_d.cpu().numpy(), np.asarray([w[1] for w in want]))
    assert tot_d.item() == total
    tree.close()
