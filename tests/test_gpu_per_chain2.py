"""dra_sumtree_per_chain2 (PrioritizedReplay.sample() on the device, csrc/sumtree.hip) through the C ABI against the oracle's
restatement of the reference (oracle/sumtree_oracle.py = sum_tree.py, replay.py:164-196 spelled out below), round after round:
priorities of the last minibatch -> tree (pending_idx gating, first occurrence wins), max_priority, the next agent step's adds,
the stratified draw from python's `random`, valid_index filter, random.choice padding.  Bit-exact: tree (every node), drawn
leaves, leaf priorities, total, ring indices, sampling probabilities, words of `random` consumed; the importance weights (powf)
at 1e-6.  Ring sizes: 300 (leaves on two depths, most draws padded), 4096 (power of two), 100003; a run with the reference's
ordered walk forced."""
import ctypes
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.sumtree_oracle import SumTreeOracle  # noqa: E402


@pytest.fixture(scope="module")
def dra():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need an MI355X")
    import deeprl_amd as d
    d.select_device(0)
    return d


def _valid(di, pos, size, h, n):       # replay.py:122-127
    return (di - h + 1 >= 0 and di + n < pos) or (di - h + 1 >= pos and di + n < size)


@pytest.mark.parametrize("cap,batch,add_n,ordered", [(300, 32, 4, 0), (4096, 32, 4, 0), (100003, 64, 8, 0), (300, 32, 4, 1),
                                                     (1000, 256, 16, 0)])
def test_per_chain2_kernel_equals_oracle(dra, cap, batch, add_n, ordered):
    d = dra
    from deeprl_amd._lib import lib
    ops = d.ops
    dev = d.Config.DEVICE
    h, n_step, eps, alpha, rounds = 4, 1, 0.01, 0.5, 12
    rs = np.random.RandomState(cap + batch)
    # ---- oracle state: a partly filled ring (every leaf at max_priority 1), then one draw on the host
    orc = SumTreeOracle(cap)
    size0 = min(cap, 3 * cap // 4 + 7)
    for _ in range(size0):
        orc.add(1.0)
    pos, size, max_p, min_p = size0 % cap, size0, 1.0, 1.0
    random.seed(cap)

    def draw():
        total = orc.total()
        seg = total / batch
        picked, raw = [], []
        for i in range(batch):
            idx, p, di = orc.get(random.uniform(seg * i, seg * (i + 1)))
            raw.append(idx)
            if _valid(di, pos, size, h, n_step):
                picked.append((idx, p))
        n_valid = len(picked)
        while len(picked) < batch:
            picked.append(random.choice(picked))
        return raw, [t[0] for t in picked], [t[1] for t in picked], total, n_valid

    _, cur_idx, cur_p, cur_total, _ = draw()
    # ---- device state
    tree = ops.SumTree(cap)
    tree_t = tree.as_tensor()
    tree_t.copy_(torch.from_numpy(orc.tree))
    stat = torch.tensor([max_p, min_p], dtype=torch.float64, device=dev)
    sb = ctypes.c_int64()
    lib.dra_sumtree_per_chain2_state_bytes(ctypes.byref(sb))
    state = torch.zeros(sb.value, dtype=torch.uint8, device=dev)
    io_t = torch.zeros(ctypes.sizeof(ops.PerChain2IO), dtype=torch.uint8).pin_memory()
    io = ops.PerChain2IO.from_address(io_t.data_ptr())
    words_t = torch.zeros(ops.PER_RNG_WORDS, dtype=torch.int32).pin_memory()
    words = words_t.numpy().view(np.uint32)
    idx_out = torch.zeros(1024, dtype=torch.int64, device=dev)
    samp = torch.zeros(batch + 1, dtype=torch.float32, device=dev)
    weights = torch.zeros(batch, dtype=torch.float32, device=dev)
    prio_out = torch.zeros(batch, dtype=torch.float32, device=dev)
    lib.dra_sumtree_per_chain2_state_set(ctypes.c_void_p(state.data_ptr()), 0, 0,
                                         np.asarray(cur_idx, dtype=np.int64).ctypes.data_as(ctypes.c_void_p), batch)
    # the word ring: the generator's next outputs, exactly as replay.DeviceDraw produces them
    st0 = random.getstate()
    n_words = 40000
    words[:n_words] = np.frombuffer(random.getrandbits(32 * n_words).to_bytes(4 * n_words, "little"), dtype="<u4")
    random.setstate(st0)
    consumed = 0
    write = orc.write
    beta = 0.4
    stream = torch.cuda.current_stream()
    for r in range(rounds):
        loss = rs.randn(batch).astype(np.float32) * (3.0 if r % 3 else 0.05)
        loss_t = torch.from_numpy(loss).to(dev)
        # ---- reference order: update_priorities(minibatch r), feed x add_n, sample (replay.py:164-196, DQN_agent.py:121-127)
        prio = np.sqrt(np.abs(loss) + np.float32(eps)).astype(np.float32)
        for idx, p in zip(cur_idx, prio):
            max_p = max(max_p, float(p))
            min_p = min(min_p, float(p))
            orc.update(idx, float(p))
        for _ in range(add_n):
            orc.add(max_p)
            if pos >= size:
                size += 1
            pos = (pos + 1) % cap
        before = random.getstate()
        raw, nxt_idx, nxt_p, total, n_valid = draw()
        after = random.getstate()
        # ---- the kernel
        io.add_n, io.batch, io.next_batch, io.force_ordered = add_n, batch, batch, ordered
        io.history, io.n_step, io.add_write0, io.memory_size = h, n_step, write, cap
        io.pos_after, io.size_after, io.rng_produced, io.beta_next = pos, size, n_words, beta
        write = (write + add_n) % cap
        lib.dra_sumtree_per_chain2(tree.h, ctypes.c_void_p(io_t.data_ptr()), ops.ptr(loss_t), eps, alpha, ops.ptr(prio_out), ops.ptr(stat),
                                   ctypes.c_void_p(state.data_ptr()), ctypes.c_void_p(words_t.data_ptr()), ops.ptr(idx_out),
                                   ops.ptr(samp), ops.ptr(weights), batch, ctypes.c_void_p(stream.cuda_stream))
        torch.cuda.synchronize()
        msg = "round %d" % r
        assert io.out_seq == r + 1 and io.out_flags == 0, msg
        assert np.array_equal(prio_out.cpu().numpy(), prio), msg
        assert np.array_equal(tree_t.cpu().numpy(), orc.tree), msg + ": tree"
        assert stat.cpu().tolist() == [max_p, min_p], msg
        assert list(io.out_raw_idx[:batch]) == raw, msg
        assert io.out_n_valid == n_valid, msg
        assert list(io.out_idx[:batch]) == nxt_idx and list(io.out_p[:batch]) == nxt_p and io.out_total == total, msg
        assert np.array_equal(idx_out[:batch].cpu().numpy(), np.asarray(nxt_idx) - (cap - 1)), msg
        want_sp = (np.asarray(nxt_p) / total).astype(np.float32)
        got = samp.cpu().numpy()
        assert np.array_equal(got[:batch], want_sp) and got[batch] == np.float32(beta), msg
        wraw = np.power(want_sp * np.float32(batch) + np.float32(1e-6), np.float32(-beta), dtype=np.float32)
        np.testing.assert_allclose(weights.cpu().numpy(), wraw / wraw.max(), rtol=1e-6, err_msg=msg)
        # words consumed == what python's generator consumed for this draw
        random.setstate(before)
        used = int(io.out_rng_cursor) - consumed
        if used:
            random.getrandbits(32 * used)
        assert random.getstate() == after, msg + ": %d words" % used
        consumed = int(io.out_rng_cursor)
        cur_idx = nxt_idx
        beta = min(1.0, beta + 0.05)
    assert n_valid <= batch
    tree.close()
