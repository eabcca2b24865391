"""Oracle: normalisers, schedules, epsilon-greedy, minibatch permuter.  TEST INFRASTRUCTURE ONLY.

numpy restatement of the small host-side numerics on the hot path.
"""
import numpy as np


def image_normalize_sync(u8):
    """Sync-replay path numerics: uint8 -> f64 * (1/255) -> f32.
    deep_rl/utils/normalizer.py:58-66 (coef * np.asarray(x): python float times a
    uint8 array is float64) followed by deep_rl/utils/torch_utils.py:23
    (np.asarray(x, dtype=np.float32))."""
    x = np.asarray(u8)
    return np.asarray((1.0 / 255) * x, dtype=np.float32)


def image_lut():
    """The 256 possible outputs of image_normalize_sync, as float32."""
    return image_normalize_sync(np.arange(256, dtype=np.uint8))


def rescale_lut(coef):
    return np.asarray(coef * np.arange(256, dtype=np.uint8), dtype=np.float32)


def sign_normalize(x):
    """deep_rl/utils/normalizer.py:69-71."""
    return np.sign(x)


class RunningMeanStdOracle:
    """baselines.common.running_mean_std.RunningMeanStd (openai/baselines @ 8e56dd,
    third-party, NOT vendored by the reference: call sites normalizer.py:8,39,41).
    Published algorithm: count-weighted parallel merge of (mean, var, count), count
    initialised to 1e-4, population variance.  Parity unpinned by the reference; it
    is checked against a two-pass numpy computation in tests."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, x):
        x = np.asarray(x)
        bm, bv, bc = x.mean(axis=0), x.var(axis=0), x.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + np.square(delta) * self.count * bc / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


class MeanStdNormalizerOracle:
    """deep_rl/utils/normalizer.py:28-51."""

    def __init__(self, read_only=False, clip=10.0, epsilon=1e-8):
        self.read_only, self.clip, self.epsilon = read_only, clip, epsilon
        self.rms = None

    def __call__(self, x):
        x = np.asarray(x)
        if self.rms is None:
            self.rms = RunningMeanStdOracle(shape=(1,) + x.shape[1:])
        if not self.read_only:
            self.rms.update(x)
        return np.clip((x - self.rms.mean) / np.sqrt(self.rms.var + self.epsilon), -self.clip, self.clip)


class LinearScheduleOracle:
    """deep_rl/utils/schedule.py:16-31: returns the current value, then advances."""

    def __init__(self, start, end=None, steps=None):
        if end is None:
            end, steps = start, 1
        self.inc = (end - start) / float(steps)
        self.current, self.end = start, end
        self.bound = min if end > start else max

    def __call__(self, steps=1):
        v = self.current
        self.current = self.bound(self.current + self.inc * steps, self.end)
        return v


def epsilon_greedy(epsilon, q):
    """deep_rl/utils/torch_utils.py:51-58.  RNG-order sensitive: 2-D input draws
    randint(A, size=N) then rand(N); 1-D input draws rand() first and randint only
    when exploring."""
    q = np.asarray(q)
    if q.ndim == 1:
        return np.random.randint(len(q)) if np.random.rand() < epsilon else np.argmax(q)
    rnd = np.random.randint(q.shape[1], size=q.shape[0])
    greedy = np.argmax(q, axis=-1)
    dice = np.random.rand(q.shape[0])
    return np.where(dice < epsilon, rnd, greedy)


def random_sample(indices, batch_size):
    """deep_rl/utils/misc.py:55-62: one np.random.permutation, full minibatches,
    then the remainder."""
    perm = np.asarray(np.random.permutation(indices))
    full = len(perm) // batch_size * batch_size
    for row in perm[:full].reshape(-1, batch_size):
        yield row
    if len(perm) % batch_size:
        yield perm[full:]
