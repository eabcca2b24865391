"""Oracle: the synthetic transition generator of SURVEY.md 8(d).  TEST INFRASTRUCTURE ONLY.

numpy restatement of dra_ring_fill_synthetic (deeprl_amd/csrc/ring.hip): frame k is
splitmix64-finalised counter words, so the CPU oracle and the GPU ring hold identical bytes
without shipping a multi-GB host array.  (The reference has no synthetic generator; this pair
only has to agree with each other.)"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_transitions(counter0, count, frame_bytes, seed, n_actions=4, done_period=800):
    """Returns (frames u8 [count, frame_bytes], action i64 [count], reward f64 [count], mask i32 [count])."""
    assert frame_bytes % 8 == 0
    words = frame_bytes // 8
    ctr = np.arange(counter0, counter0 + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * _GOLD + ctr * np.uint64(words)
        w = mix64(base[:, None] + np.arange(words, dtype=np.uint64)[None, :])
        h = mix64((np.uint64(seed) + np.uint64(1)) * _GOLD + ctr)
        h2 = mix64((np.uint64(seed) + np.uint64(2)) * _GOLD + ctr)
    frames = w.astype("<u8").view(np.uint8).reshape(count, frame_bytes)
    action = ((h & np.uint64(0xFFFFFFFF)) % np.uint64(n_actions)).astype(np.int64)
    u = ((h >> np.uint64(32)).astype(np.uint32) % np.uint32(10))
    reward = np.where(u == 0, -1.0, np.where(u == 9, 1.0, 0.0)).astype(np.float64)
    mask = np.where(h2 % np.uint64(done_period) == 0, 0, 1).astype(np.int32)
    return frames, action, reward, mask
