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
