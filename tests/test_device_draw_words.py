"""The word protocol between python's `random` and the device-side prioritized draw (csrc/sumtree.hip
sumtree_per_chain2_kernel; host side replay.DeviceDraw), on the CPU: the kernel's arithmetic on raw Mersenne-Twister words,
transcribed here, must reproduce random.uniform / random.choice (replay.py:169-186) value for value, and
DeviceDraw.release() must leave the module-level generator exactly behind the last word consumed."""
import collections
import random

import numpy as np

from deeprl_amd import ops
from deeprl_amd.replay import DeviceDraw


def words_of(n):
    return np.frombuffer(random.getrandbits(32 * n).to_bytes(4 * n, "little"), dtype="<u4")


def kernel_uniform(w0, w1):
    return (float(int(w0) >> 5) * 67108864.0 + float(int(w1) >> 6)) * (1.0 / 9007199254740992.0)


def kernel_randbelow(words, cur, n):
    k = n.bit_length()
    while True:
        r = int(words[cur]) >> (32 - k)
        cur += 1
        if r < n:
            return r, cur


def test_words_reproduce_random_random_and_choice():
    for seed in (0, 3, 12345):
        random.seed(seed)
        w = words_of(4096)
        random.seed(seed)
        cur = 0
        for step in range(40):
            # 32 uniforms of a stratified draw (random.uniform(a, b) = a + (b - a) * random())
            for i in range(32):
                a, b = 0.37 * i, 0.37 * (i + 1)
                want = random.uniform(a, b)
                got = a + (b - a) * kernel_uniform(w[cur], w[cur + 1])
                cur += 2
                assert got == want
            # paddings over lists of every length the kernel can meet
            picked = list(range(1 + (step * 7) % 31))
            while len(picked) < 32:
                want = random.choice(picked)
                r, cur = kernel_randbelow(w, cur, len(picked))
                assert picked[r] == want
                picked.append(want)
        # the generator stands exactly behind word `cur`
        tail = random.getrandbits(32)
        assert tail == int(w[cur])


def _bare_draw():
    dd = object.__new__(DeviceDraw)
    dd.words = np.zeros(ops.PER_RNG_WORDS, dtype=np.uint32)
    dd.produced = dd.consumed = 0
    dd.ckpt = collections.deque()
    dd.fifo = collections.deque()
    return dd


def test_release_rewinds_the_generator_to_the_consumed_word():
    random.seed(11)
    ref = words_of(200_000)
    random.seed(11)
    dd = _bare_draw()
    consumed = 0
    rng = np.random.RandomState(0)
    for k in range(300):
        dd._ensure_words(8 * 32 + 256)
        # the ring holds the reference stream at every position not yet consumed
        for pos in (dd.consumed, dd.produced - 1):
            assert dd.words[pos & (ops.PER_RNG_WORDS - 1)] == ref[pos]
        consumed += 64 + int(rng.randint(0, 40))          # a draw + some rejection / padding words
        assert consumed <= dd.produced
        dd.consumed = consumed
        if k % 37 == 36:
            dd.release()
            assert dd.produced == dd.consumed == consumed and not dd.ckpt
            state = random.getstate()
            assert random.getrandbits(32) == int(ref[consumed])
            random.setstate(state)
    dd.release()
    assert random.getrandbits(32) == int(ref[consumed])


def test_collect_keeps_pending_idx_in_the_devices_order():
    """DeviceDraw.collect() mirrors sum_tree.py's pending_idx one launch late, in the order the kernel worked: commit of the
    launch's own minibatch (every distinct sampled leaf), the next agent step's adds, then every leaf of the new draw (the
    filtered-out ones included).  Driven here with a stub learner and hand-made launch blocks against the oracle's own
    bookkeeping (oracle/sumtree_oracle.py: get() adds, update() / add() remove)."""
    from oracle.sumtree_oracle import SumTreeOracle
    cap, B, n_env, h, n_step = 64, 8, 2, 4, 1
    rs = np.random.RandomState(3)
    orc = SumTreeOracle(cap)
    for _ in range(40):
        orc.add(1.0)
    pos, size = 40, 40

    class _Rp:
        pass

    rp = _Rp()
    rp._pending, rp.batch_size, rp.memory_size = set(), B, cap

    class _L:
        def per_chain2_wait(self, slot, seq):
            pass

    dd = _bare_draw()
    dd.rp, dd.L = rp, _L()
    dd.blocks = [(None, None, dict(raw=np.zeros(1024, np.int64), idx=np.zeros(1024, np.int64), p=np.zeros(1024), total=np.zeros(1),
                                   tail=np.zeros(2, np.int32), cursor=np.zeros(1, np.uint64))) for _ in range(4)]

    def draw():          # replay.py:164-186 on the oracle: every descent is pending, invalid ones are dropped, then padding
        raw, picked = [], []
        for _ in range(B):
            idx, p, di = orc.get(rs.uniform(0, orc.total()))
            raw.append(idx)
            if (di - h + 1 >= 0 and di + n_step < pos) or (di - h + 1 >= pos and di + n_step < size):
                picked.append(idx)
        n_valid = len(picked)
        while len(picked) < B:
            picked.append(picked[rs.randint(len(picked))])
        return raw, picked, n_valid

    raw, cur, _ = draw()                      # the first minibatch: drawn on the host (DeviceDraw.start)
    rp._pending = set(orc.pending)
    dd._collect_leaves = list(cur)
    dd.next_tree_idx = np.asarray(cur)
    write = orc.write
    some_invalid = False
    for t in range(60):
        # what launch t does on the device, replayed on the oracle
        for leaf in cur:
            orc.update(leaf, float(rs.uniform(0.1, 2.0)))
        adds = [(write + i) % cap + cap - 1 for i in range(n_env)]
        for _ in range(n_env):
            orc.add(2.0)
            if pos >= size:
                size += 1
            pos = (pos + 1) % cap
        write = (write + n_env) % cap
        raw, nxt, n_valid = draw()
        some_invalid = some_invalid or n_valid < B
        # its pinned block, then the host's (lagging) collect
        slot = t & 3
        v = dd.blocks[slot][2]
        v["raw"][:B], v["idx"][:B], v["tail"][:] = raw, nxt, (n_valid, 0)
        v["cursor"][0] = 100 * (t + 1)
        dd.fifo.append((slot, t + 1, adds, 0.5))
        dd.collect()
        assert rp._pending == orc.pending, "launch %d" % t
        assert dd.next_tree_idx.tolist() == nxt and dd.consumed == 100 * (t + 1)
        cur = nxt
    assert some_invalid, "the scenario must contain filtered draws"


def test_delta_propagation_commutes_exactly_in_the_exact_regime():
    """sumtree_per_chain2_kernel writes priorities as tree[ancestor] += (new - old) with f64 atomics in whatever order the
    hardware performs them.  That equals the reference's sequential walk (sum_tree.py:46-60) BIT FOR BIT whenever every value
    is a multiple of one quantum and the total stays below 2^53 quanta -- the kernel's `ordered == 0` condition
    (capacity * max <= 2^(53 + ilogb(min) - 23) for f32 priorities) -- and this test replays both on the oracle's tree in a
    shuffled order.  Outside the regime (a priority 2^40 times smaller than the largest) the orders do differ, which is why
    the kernel falls back to the ordered walk there."""
    from oracle.sumtree_oracle import SumTreeOracle
    rs = np.random.RandomState(5)
    cap = 1000
    for spread, expect_exact in ((1.0, True), (2.0 ** -40, False)):
        a, b = SumTreeOracle(cap), SumTreeOracle(cap)
        for t in (a, b):
            for _ in range(cap):
                t.add(1.0)
        differed = False
        lo, hi = 1.0, 1.0
        for _ in range(40):
            leaves = rs.choice(cap, size=32, replace=False) + cap - 1
            prio = np.sqrt(np.abs(rs.standard_normal(32)).astype(np.float32) + np.float32(0.01)).astype(np.float32)
            prio[::7] *= np.float32(spread)
            lo, hi = min(lo, float(prio.min())), max(hi, float(prio.max()))
            for leaf, p in zip(leaves, prio):                        # the reference: one leaf after the other, leaf -> root
                a.pending.add(int(leaf))
                a.update(int(leaf), float(p))
            ops_ = []                                                # the kernel: every (ancestor, delta) pair, any order
            for leaf, p in zip(leaves, prio):
                delta = float(p) - b.tree[leaf]
                b.tree[leaf] = float(p)
                node = int(leaf)
                while node > 0:
                    node = (node - 1) // 2
                    ops_.append((node, delta))
            for k in rs.permutation(len(ops_)):
                b.tree[ops_[k][0]] += ops_[k][1]
            differed = differed or not np.array_equal(a.tree, b.tree)
        in_regime = cap * hi <= 2.0 ** (53 + int(np.floor(np.log2(lo))) - 23)
        assert in_regime == expect_exact
        assert differed != expect_exact
