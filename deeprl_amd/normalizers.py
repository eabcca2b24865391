"""State / reward normalisers (deep_rl/utils/normalizer.py:11-71).

numpy inputs (what environments hand over) are processed on the host exactly as the reference
does; a uint8 DEVICE tensor (what the HBM replay returns) goes through the HIP table kernel
whose 256 entries are f32(f64(v) * coef) -- the reference's sync-replay numerics
(normalizer.py:58-61 then torch_utils.py:23), bit for bit.
"""
import numpy as np
import torch


class BaseNormalizer:
    def __init__(self, read_only=False):
        self.read_only = read_only

    def set_read_only(self):
        self.read_only = True

    def unset_read_only(self):
        self.read_only = False

    def state_dict(self):
        return None

    def load_state_dict(self, _):
        return


class RunningMeanStd:
    """Streaming mean / population variance over axis 0 by Chan's pairwise merge: a batch with moments (m_b, v_b, n_b) is
    folded into the running (m, v, n) as  m' = m + d n_b / N,  v' = (v n + v_b n_b + d^2 n n_b / N) / N  with d = m_b - m,
    N = n + n_b.  Starts from mean 0, variance 1 and a pseudo-count of 1e-4 -- the published algorithm of
    baselines.common.running_mean_std (@8e56dd), the third-party class the reference imports at normalizer.py:8 (absent
    here; restated, checked against a two-pass computation in tests/test_oracle_vs_golden.py)."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean, self.var, self.count = np.zeros(shape, 'float64'), np.ones(shape, 'float64'), epsilon

    def update(self, x):
        x = np.asarray(x)
        self.merge(np.mean(x, axis=0), np.var(x, axis=0), x.shape[0])

    def merge(self, b_mean, b_var, b_count):
        n, total = self.count, self.count + b_count
        delta = b_mean - self.mean
        m2 = self.var * n + b_var * b_count + np.square(delta) * n * b_count / total
        self.mean, self.var, self.count = self.mean + delta * b_count / total, m2 / total, total


class MeanStdNormalizer(BaseNormalizer):
    """clip((x - running mean) / sqrt(running var + epsilon), +-clip), the statistics updated by every call unless
    read_only (the interface and state_dict format {'mean', 'var'} of normalizer.py:28-51).  Host-side fp64 numpy, like
    the reference: it sits on the environment side of the boundary (one [num_envs, 17] block per environment step)."""

    def __init__(self, read_only=False, clip=10.0, epsilon=1e-8):
        BaseNormalizer.__init__(self, read_only)
        self.clip, self.epsilon, self.rms = clip, epsilon, None

    def __call__(self, x):
        if isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float64:
            return self._call_device(x)
        x = np.asarray(x)
        self.sync_from_device()
        if self.rms is None:                       # statistics are per feature, shared by the environments
            self.rms = RunningMeanStd(shape=(1,) + x.shape[1:])
        if not self.read_only:
            self.rms.update(x)
            self._push_to_device()                 # (statistics attached to the device: the device copy is the one that lives on)
        z = (x - self.rms.mean) / np.sqrt(self.rms.var + self.epsilon)
        return np.clip(z, -self.clip, self.clip)

    def _push_to_device(self):
        """A host-side update while the statistics live on the device: write mean / var / count back, so that the next
        device rollout (or sync_from_device) continues from them instead of discarding the update."""
        dev = getattr(self, '_dev', None)
        if dev is not None:
            d = self._dim
            h = np.concatenate([np.asarray(self.rms.mean, dtype=np.float64).reshape(-1),
                                np.asarray(self.rms.var, dtype=np.float64).reshape(-1), [float(self.rms.count)]])
            dev[:2 * d + 1].copy_(torch.from_numpy(h))

    def attach_device(self, rms_dev, dim):
        """The statistics moved to the device (device_env.DeviceContinuousVec: f64 [mean (dim) | var (dim) | count]); the host
        copy is refreshed from there whenever it is asked for."""
        self._dev, self._dim = rms_dev, int(dim)

    def sync_from_device(self):
        dev = getattr(self, '_dev', None)
        if dev is not None:
            h, d = dev.cpu().numpy(), self._dim
            shape = self.rms.mean.shape
            self.rms.mean, self.rms.var, self.rms.count = h[:d].reshape(shape).copy(), h[d:2 * d].reshape(shape).copy(), float(h[2 * d])

    def _call_device(self, x):
        """A float64 DEVICE batch [n, d]: one kernel (dra_rms_normalize) updates the statistics -- kept on the device from the
        first such call on -- and returns the float32 tensor tensor() would have uploaded.  Same operation order as the host
        arithmetic above, bit for bit (tests/test_gpu_ppo_mlp.py)."""
        from . import ppo_mlp
        d = x.shape[1]
        if getattr(self, '_dev', None) is None:
            if self.rms is None:
                self.rms = RunningMeanStd(shape=(1, d))
            h = np.concatenate([np.asarray(self.rms.mean, dtype=np.float64).reshape(-1),
                                np.asarray(self.rms.var, dtype=np.float64).reshape(-1), [float(self.rms.count)]])
            self.attach_device(torch.from_numpy(h).to(x.device), d)
        dev = self._dev
        out, _ = ppo_mlp.rms_normalize(x, dev[:d], dev[d:2 * d], dev[2 * d:], update=not self.read_only, epsilon=self.epsilon,
                                       clip=self.clip)
        return out

    def state_dict(self):
        self.sync_from_device()
        return dict(mean=self.rms.mean, var=self.rms.var)

    def load_state_dict(self, saved):
        self.rms.mean, self.rms.var = saved['mean'], saved['var']
        dev = getattr(self, '_dev', None)
        if dev is not None:
            import torch
            d = self._dim
            dev[:d].copy_(torch.from_numpy(np.asarray(self.rms.mean, dtype=np.float64).reshape(-1)))
            dev[d:2 * d].copy_(torch.from_numpy(np.asarray(self.rms.var, dtype=np.float64).reshape(-1)))


class RescaleNormalizer(BaseNormalizer):
    def __init__(self, coef=1.0):
        BaseNormalizer.__init__(self)
        self.coef = coef
        self._lut = {}

    def lut(self, device):
        """f32(f64(v) * coef) for v in 0..255, resident on `device`."""
        key = str(device)
        if key not in self._lut:
            table = np.asarray(self.coef * np.arange(256, dtype=np.uint8), dtype=np.float32)
            self._lut[key] = torch.from_numpy(table).to(device)
        return self._lut[key]

    def __call__(self, x):
        if isinstance(x, torch.Tensor):
            if x.dtype == torch.uint8 and x.is_cuda:
                from . import ops
                return ops.u8_to_f32(x, self.lut(x.device))
            if self.coef == 1.0:
                return x
            return self.coef * x
        return self.coef * np.asarray(x)


class ImageNormalizer(RescaleNormalizer):
    def __init__(self):
        RescaleNormalizer.__init__(self, 1.0 / 255)


class SignNormalizer(BaseNormalizer):
    def __call__(self, x):
        return np.sign(x)
