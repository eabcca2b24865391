"""Host side of a prioritized draw (PrioritizedReplay.draw_end, replay.py:173-186 of the reference) without a GPU: the
vectorised all-valid shortcut and the per-sample loop must agree with a literal restatement of the reference's loop --
same leaves, probabilities, data indices, pending marks, and the same consumption of python's `random` (one
random.choice per padded slot)."""
import random

import numpy as np

import deeprl_amd as d


class _Done:
    def synchronize(self):
        pass


def _reference_loop(rp, tree_idx, p, total, batch_size):
    pending, picked = set(), []
    for i in range(batch_size):
        ti = int(tree_idx[i])
        pending.add(ti)                                   # sum_tree.py:66
        di = ti - rp.memory_size + 1
        if not rp.valid_index(di):                        # replay.py:176-177
            continue
        picked.append((ti, p[i] / total, di))
    while len(picked) < batch_size:
        picked.append(random.choice(picked))              # replay.py:184-186
    return (np.asarray([t[0] for t in picked], dtype=np.int64), np.asarray([t[1] for t in picked], dtype=np.float64),
            np.asarray([t[2] for t in picked], dtype=np.int64)), pending


def test_draw_end_shortcut_and_loop_equal_the_reference_loop():
    rs = np.random.RandomState(3)
    n_fast = n_slow = 0
    for case in range(300):
        cap = int(rs.choice([50, 300, 4096]))
        h, n = int(rs.choice([1, 4])), int(rs.choice([1, 3]))
        b = int(rs.choice([8, 32]))
        rp = d.PrioritizedReplay(memory_size=cap, batch_size=b, n_step=n, discount=0.99, history_length=h)
        full = bool(rs.rand() < 0.6)
        rp.pos = int(rs.randint(0, cap))
        rp._size = cap if full else max(rp.pos, 1)
        # data indices mostly away from the write head, sometimes right at it (invalid draws -> padding)
        if rs.rand() < 0.5:
            di = rs.randint(0, rp._size, size=b)
        else:
            lo, hi = h - 1, rp._size - n - 1
            di = rs.randint(lo, max(lo + 1, hi), size=b)
            di = di[(di - h + 1 >= rp.pos) | (di + n < rp.pos)] if full else di[di + n < rp.pos]
            if len(di) == 0:
                continue
            di = np.resize(di, b)
        tree_idx = (di + cap - 1).astype(np.int64)
        p = rs.rand(b) + 0.1
        total = float(p.sum() * 3)
        rp._draw_np = (tree_idx.copy(), p.copy(), np.asarray([total]))
        if not any(rp.valid_index(int(x)) for x in di):
            continue                                      # (the reference itself would fail on an empty `picked`)
        random.seed(case)
        want, want_pending = _reference_loop(rp, tree_idx, p, total, b)
        state_after = random.getstate()
        random.seed(case)
        rp._pending = set()
        got = rp.draw_end((b, _Done()))
        assert random.getstate() == state_after
        for g, w in zip(got, want):
            assert g.dtype == w.dtype and np.array_equal(g, w)
        assert rp._pending == want_pending
        if all(rp.valid_index(int(x)) for x in di):
            n_fast += 1
        else:
            n_slow += 1
    assert n_fast > 30 and n_slow > 30, (n_fast, n_slow)
