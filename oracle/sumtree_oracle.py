"""Oracle: fp64 binary-heap sum tree.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates deep_rl/utils/sum_tree.py:6-66 iteratively (the reference recurses).
Heap layout: 2*cap-1 float64 nodes, root 0, children 2i+1 / 2i+2, leaves at
cap-1 .. 2cap-2 (sum_tree.py:10,23-33).
"""
import numpy as np


class SumTreeOracle:
    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.tree = np.zeros(2 * self.capacity - 1, dtype=np.float64)  # sum_tree.py:10
        self.write = 0
        self.n_entries = 0
        self.pending = set()  # sum_tree.py:13

    def total(self):
        return self.tree[0]  # sum_tree.py:35-36

    def update(self, idx, p):
        """sum_tree.py:54-60 + _propagate :16-20: no-op unless idx is pending;
        change is added to every ancestor in leaf->root order."""
        idx = int(idx)
        if idx not in self.pending:
            return False
        self.pending.remove(idx)
        change = p - self.tree[idx]
        self.tree[idx] = p
        node = idx
        while True:
            node = (node - 1) // 2
            self.tree[node] += change
            if node == 0:
                break
        return True

    def add(self, p):
        """sum_tree.py:39-51: the new leaf is self-marked pending, then updated."""
        idx = self.write + self.capacity - 1
        self.pending.add(idx)
        self.update(idx, p)
        self.write += 1
        if self.write >= self.capacity:
            self.write = 0
        if self.n_entries < self.capacity:
            self.n_entries += 1

    def get(self, s):
        """sum_tree.py:23-33,63-66: descend `s <= left ? left : (right, s-left)`
        until 2i+1 >= len(tree); marks the leaf pending."""
        idx = 0
        n = len(self.tree)
        while True:
            left = 2 * idx + 1
            if left >= n:
                break
            lv = self.tree[left]
            if s <= lv:
                idx = left
            else:
                idx = left + 1
                s = s - lv
        self.pending.add(idx)
        return idx, self.tree[idx], idx - self.capacity + 1

    def rebuilt(self):
        """Bottom-up rebuild of the internal nodes from the leaves (used to check
        the fp64-exactness regime described in SURVEY.md section 7)."""
        t = self.tree.copy()
        for i in range(self.capacity - 2, -1, -1):
            t[i] = t[2 * i + 1] + t[2 * i + 2]
        return t
