"""Oracle: replay ring (uniform + prioritized).  TEST INFRASTRUCTURE ONLY.

Restates deep_rl/component/replay.py:57-196 over flat numpy arrays (the
reference keeps Python lists of per-slot numpy arrays).  Semantics preserved,
including the quirks listed in SURVEY.md Appendix A:
  * a sample is the contiguous slot run [i-H+1, i+n]  (replay.py:105-110,115-125)
  * frame stacks cross episode boundaries           (replay.py:124-125)
  * next_state is rebuilt from state[i+n]           (replay.py:125)
  * one global-RandomState draw per attempt, rejected draws consumed (replay.py:97-100)
  * n-step return in fp64: cum_r = r + (m*gamma)*cum_r, reversed (replay.py:135-139)
"""
import random

import numpy as np

from .sumtree_oracle import SumTreeOracle


class UniformReplayOracle:
    def __init__(self, memory_size, batch_size, n_step=1, discount=1, history_length=1):
        self.memory_size = int(memory_size)
        self.batch_size = int(batch_size)
        self.n_step = int(n_step)
        self.discount = discount
        self.history_length = int(history_length)
        self.pos = 0
        self._size = 0
        self.state = None
        self.action = None
        self.reward = np.zeros(self.memory_size, dtype=np.float64)
        self.mask = np.zeros(self.memory_size, dtype=np.int32)

    def size(self):
        return self._size

    def feed_one(self, state, action, reward, mask):
        """replay.py:75-90 for one environment (the only well-defined case, see
        SURVEY.md section 7 'quirks'): write slot `pos`, grow to memory_size, wrap."""
        state = np.asarray(state)
        action = np.asarray(action)
        if self.state is None:
            self.state = np.zeros((self.memory_size,) + state.shape, dtype=state.dtype)
            self.action = np.zeros((self.memory_size,) + action.shape, dtype=action.dtype)
        p = self.pos
        self.state[p] = state
        self.action[p] = action
        self.reward[p] = reward
        self.mask[p] = mask
        if p >= self._size:
            self._size += 1
        self.pos = (p + 1) % self.memory_size

    def valid_index(self, i):
        """replay.py:105-110."""
        h, n = self.history_length, self.n_step
        if i - h + 1 >= 0 and i + n < self.pos:
            return True
        if i - h + 1 >= self.pos and i + n < self._size:
            return True
        return False

    def draw_indices(self, batch_size=None):
        """replay.py:92-100: rejection loop over np.random.randint(0, size)."""
        b = self.batch_size if batch_size is None else batch_size
        out = []
        while len(out) < b:
            i = int(np.random.randint(0, self._size))
            if self.valid_index(i):
                out.append(i)
        return np.asarray(out, dtype=np.int64)

    def nstep(self, i):
        """replay.py:135-139 (fp64; python `and` chain for the mask)."""
        cum_r = 0
        cum_mask = np.int32(1)
        for k in range(self.n_step - 1, -1, -1):
            m = self.mask[i + k]
            cum_r = self.reward[i + k] + (m * self.discount) * cum_r
            cum_mask = m if cum_mask else cum_mask
        return np.float64(cum_r), np.int32(cum_mask)

    def gather(self, idx):
        """replay.py:112-140 for a batch of already-validated indices."""
        h, n = self.history_length, self.n_step
        idx = np.asarray(idx, dtype=np.int64)
        if h == 1:
            state = self.state[idx]
            next_state = self.state[idx + n]
        else:
            state = np.stack([self.state[i - h + 1:i + 1] for i in idx])
            next_state = np.stack([self.state[i - h + 1 + n:i + n + 1] for i in idx])
        action = self.action[idx]
        rm = [self.nstep(int(i)) for i in idx]
        reward = np.asarray([r for r, _ in rm], dtype=np.float64)
        mask = np.asarray([m for _, m in rm], dtype=np.int32)
        return state, action, reward, next_state, mask

    def sample(self, batch_size=None):
        idx = self.draw_indices(batch_size)
        return self.gather(idx) + (idx,)


class PrioritizedReplayOracle(UniformReplayOracle):
    """replay.py:152-196."""

    def __init__(self, memory_size, batch_size, n_step=1, discount=1, history_length=1):
        super().__init__(memory_size, batch_size, n_step, discount, history_length)
        self.tree = SumTreeOracle(memory_size)
        self.max_priority = 1

    def feed_one(self, state, action, reward, mask):
        super().feed_one(state, action, reward, mask)
        self.tree.add(self.max_priority)  # replay.py:160-162

    def draw(self, batch_size=None):
        """replay.py:164-186: stratified random.uniform over `total/B` segments;
        invalid leaves are skipped, then the batch is padded by random.choice.
        Returns (tree_idx, sampling_prob, data_idx)."""
        b = self.batch_size if batch_size is None else batch_size
        segment = self.tree.total() / b
        picked = []
        for i in range(b):
            s = random.uniform(segment * i, segment * (i + 1))
            idx, p, data_idx = self.tree.get(s)
            if not self.valid_index(data_idx):
                continue
            picked.append((idx, p / self.tree.total(), data_idx))
        while len(picked) < b:
            picked.append(random.choice(picked))
        tree_idx = np.asarray([t[0] for t in picked], dtype=np.int64)
        prob = np.asarray([t[1] for t in picked], dtype=np.float64)
        data_idx = np.asarray([t[2] for t in picked], dtype=np.int64)
        return tree_idx, prob, data_idx

    def sample(self, batch_size=None):
        tree_idx, prob, data_idx = self.draw(batch_size)
        return self.gather(data_idx) + (prob, tree_idx)

    def update_priorities(self, info):
        """replay.py:193-196: max_priority tracks every offered priority, even
        when the tree update is dropped (idx not pending)."""
        for idx, priority in info:
            self.max_priority = max(self.max_priority, priority)
            self.tree.update(idx, priority)
