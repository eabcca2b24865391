"""Exploration noise for the deterministic-policy agents (the interface of deep_rl/component/random_process.py:
`sample()` -> numpy noise of the configured shape, `reset_states()`; `std` is a schedule object called once per
sample, e.g. LinearSchedule(0.2)).  Both processes draw from the GLOBAL np.random stream with one standard-normal
block per sample, so a seeded run consumes the stream exactly like the reference."""
import numpy as np


class RandomProcess:
    def reset_states(self):
        """Called at every episode start (DDPG_agent.py:42,66)."""


class GaussianProcess(RandomProcess):
    """Independent N(0, std()^2) noise (TD3's exploration, examples.py:606-607)."""

    def __init__(self, size, std):
        self.size, self.std = size, std

    def sample(self):
        z = np.random.randn(*self.size)
        return z * self.std()


class OrnsteinUhlenbeckProcess(RandomProcess):
    """Euler-Maruyama discretisation of dx = theta (mu - x) dt + std dW (DDPG's exploration, examples.py:577-578):
    x <- x + theta (mu - x) dt + std() sqrt(dt) N(0, I), restarted from x0 (default 0) at every episode."""

    def __init__(self, size, std, theta=.15, dt=1e-2, x0=None):
        self.size, self.std, self.theta, self.dt, self.x0, self.mu = size, std, theta, dt, x0, 0
        self.reset_states()

    def reset_states(self):
        self.x_prev = np.zeros(self.size) if self.x0 is None else self.x0

    def sample(self):
        drift = self.theta * (self.mu - self.x_prev) * self.dt
        diffusion = self.std() * np.sqrt(self.dt) * np.random.randn(*self.size)
        self.x_prev = self.x_prev + drift + diffusion
        return self.x_prev
