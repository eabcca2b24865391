"""Environment surface (deep_rl/component/envs.py:92-189): `LazyFrames`, the auto-resetting
vector env contract and `Task`.

The emulators themselves (gym 0.10.8 / baselines / atari-py / mujoco, envs.py:8-16,27-55) are
third-party CPU code outside the hot path and absent from this image, so `Task`
builds SYNTHETIC environments with the same surface -- reset() / step(actions) -> (obs, reward,
done, info tuple with 'episodic_return'), state_dim, action_dim, name, action_space -- and the
same observation shapes / dtypes: Atari-like names give uint8 [4,84,84] frame stacks, everything
else float64 vectors.  Frames are a counter hash (identical to dra_ring_fill_synthetic), so runs
are reproducible without an emulator.
"""
import numpy as np


class LazyFrames(object):
    """envs.py:92-113: frame stack that materialises on demand; `s[-1]` is the newest frame."""

    def __init__(self, frames):
        self._frames = frames

    def __array__(self, dtype=None, copy=None):
        out = np.concatenate(self._frames, axis=0)
        if dtype is not None:
            out = out.astype(dtype)
        return out

    def __len__(self):
        return len(self.__array__())

    def __getitem__(self, i):
        return self.__array__()[i]


class Discrete:
    def __init__(self, n):
        self.n = n


class Box:
    def __init__(self, low, high, shape, seed=0):
        self.low, self.high, self.shape = low, high, shape
        self._rs = np.random.RandomState(seed)      # gym's spaces draw from their own generator, not np.random

    def sample(self):
        return self._rs.uniform(np.broadcast_to(self.low, self.shape), np.broadcast_to(self.high, self.shape))


_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synthetic_frame(counter, seed, frame_bytes=7056):
    """Frame `counter` of stream `seed`: the same bytes dra_ring_fill_synthetic writes."""
    words = frame_bytes // 8
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * _GOLD + np.uint64(counter) * np.uint64(words)
        w = _mix64(base + np.arange(words, dtype=np.uint64))
    return w.astype("<u8").view(np.uint8)


def synthetic_reward_done(counter, seed, done_period):
    """(reward, done) hashed from `counter`: what SyntheticAtari.step returns when its frame counter is `counter`, and what
    the device environment stores for dra_dqn_step_params.rcounter (csrc/actor_env.h synth_reward / synth_mask)."""
    with np.errstate(over="ignore"):
        h = int(_mix64(np.uint64(seed + 1) * _GOLD + np.uint64(counter)))
        h2 = int(_mix64(np.uint64(seed + 2) * _GOLD + np.uint64(counter)))
    u = (h >> 32) % 10
    return (-1.0 if u == 0 else (1.0 if u == 9 else 0.0)), (h2 % done_period == 0)


def synthetic_reward_done_vec(counters, seeds, done_periods):
    """synthetic_reward_done for arrays (one environment per element): (rewards f64[N], dones bool[N])."""
    with np.errstate(over="ignore"):
        c = np.asarray(counters).astype(np.uint64)
        sd = np.asarray(seeds).astype(np.uint64)
        h = _mix64((sd + np.uint64(1)) * _GOLD + c)
        h2 = _mix64((sd + np.uint64(2)) * _GOLD + c)
    u = (h >> np.uint64(32)) % np.uint64(10)
    reward = np.where(u == 0, -1.0, np.where(u == 9, 1.0, 0.0))
    return reward, (h2 % np.asarray(done_periods).astype(np.uint64)) == 0


class SyntheticAtari:
    """uint8 [history,84,84] observations as LazyFrames of (1,84,84) frames, `n_actions` discrete
    actions, reward in {-1,0,1}, episode ends w.p. 1/800 per step."""

    def __init__(self, seed=0, history=4, n_actions=4, done_period=800):
        self.seed, self.history, self.n_actions, self.done_period = seed, history, n_actions, done_period
        self.counter = 0
        self.frames = None
        self.ret = 0.0

    # The stand-in emulator must not be what a host-environment benchmark measures: frames and (reward, done) pairs are
    # hashed BLOCK counters at a time (one vectorised numpy pass instead of ~25 us of scalar-array overhead per step; same
    # bytes as synthetic_frame / synthetic_reward_done: tests/test_dropin_surface.py).
    BLOCK = 256

    def _block(self, counter):
        base = getattr(self, "_blk_base", None)
        if base is None or not (base <= counter < base + self.BLOCK):
            base = self._blk_base = counter
            n, words = self.BLOCK, 7056 // 8
            with np.errstate(over="ignore"):
                ctr = np.arange(base, base + n, dtype=np.uint64)
                b0 = np.uint64(self.seed) * _GOLD + ctr * np.uint64(words)
                w = _mix64(b0[:, None] + np.arange(words, dtype=np.uint64)[None, :])
            self._blk_frames = w.astype("<u8").view(np.uint8).reshape(n, 1, 84, 84)
            r, d = synthetic_reward_done_vec(ctr.astype(np.int64), np.full(n, self.seed, dtype=np.int64),
                                             np.full(n, self.done_period, dtype=np.int64))
            self._blk_r, self._blk_d = r.tolist(), d.tolist()
        return counter - base

    def _next_frame(self):
        i = self._block(self.counter)
        self.counter += 1
        return self._blk_frames[i]

    def reset(self):
        f = self._next_frame()
        self.frames = [f] * self.history
        self.ret = 0.0
        return LazyFrames(list(self.frames))

    def step(self, action):
        i = self._block(self.counter)
        reward, done = self._blk_r[i], bool(self._blk_d[i])
        self.frames = self.frames[1:] + [self._next_frame()]
        self.ret += reward
        info = {'episodic_return': self.ret if done else None}
        return LazyFrames(list(self.frames)), reward, done, info


class SyntheticVector:
    def __init__(self, seed=0, state_dim=4, n_actions=2, continuous=False, horizon=200):
        self.rs = np.random.RandomState(seed)
        self.state_dim, self.n_actions, self.continuous, self.horizon = state_dim, n_actions, continuous, horizon
        self.s = None
        self.ret = 0.0

    def reset(self):
        self.s = self.rs.uniform(-0.05, 0.05, size=self.state_dim)
        self.ret = 0.0
        return self.s.copy()

    def step(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        self.s = self.s + 0.01 * (a.mean() - (0.0 if self.continuous else (self.n_actions - 1) / 2.0)) + \
            self.rs.uniform(-0.02, 0.02, size=self.state_dim)
        reward = float(self.rs.randn()) if self.continuous else 1.0
        done = bool(self.rs.rand() < 1.0 / self.horizon)
        self.ret += reward
        info = {'episodic_return': self.ret if done else None}
        return self.s.copy(), reward, done, info


class SyntheticContinuous:
    """Continuous-control stand-in (HalfCheetah shapes by default: 17 observations, 6 actions in [-1, 1]) whose every
    quantity is a hash of (seed, stream, step counter, component) -- the host statement of csrc/cont_env.h, bit for bit, so
    that the same environments can live on the device (device_env.DeviceContinuousVec) and a device rollout can be compared
    with this class stepped from python:
        step(a):  c += 1;  m = (((a0 + a1) + a2) + ...) / A  (fp64);  s = (s + 0.01 m) + (-0.02 + 0.04 U(0, c, j))
                  reward = (((U(1,c,0) + U(1,c,1)) + (U(1,c,2) + U(1,c,3))) - 2) * sqrt(3);  done = hash(2, c) mod horizon == 0
        reset():  s_j = -0.05 + 0.1 U(3, c, j)
    The observation drifts with the mean action, so a rollout keeps the data dependence a real simulator has."""

    def __init__(self, seed=0, state_dim=17, action_dim=6, horizon=1000):
        self.seed, self.state_dim, self.action_dim, self.horizon = int(seed), int(state_dim), int(action_dim), int(horizon)
        self.c = 0
        self.s = None
        self.ret = 0.0
        self._j = np.arange(self.state_dim, dtype=np.uint64)

    def _hash(self, stream, j):
        with np.errstate(over="ignore"):
            base = (np.uint64(self.seed) * np.uint64(8) + np.uint64(stream + 1)) * _GOLD + np.uint64(self.c) * np.uint64(64)
            return _mix64(base + j)

    def _u(self, stream, j):
        return (self._hash(stream, j) >> np.uint64(11)).astype(np.float64) * 1.1102230246251565e-16

    def reset(self):
        self.s = -0.05 + 0.1 * self._u(3, self._j)
        self.ret = 0.0
        return self.s.copy()

    def step(self, action):
        a = np.clip(np.asarray(action, dtype=np.float32).reshape(-1), np.float32(-1.0), np.float32(1.0)).astype(np.float64)
        self.c += 1
        mean_a = float(np.cumsum(a)[-1]) / float(a.size)         # left-to-right sum (np.mean's pairwise order is numpy's business)
        self.s = (self.s + 0.01 * mean_a) + (-0.02 + 0.04 * self._u(0, self._j))
        u = self._u(1, np.arange(4, dtype=np.uint64))
        reward = float((((u[0] + u[1]) + (u[2] + u[3])) - 2.0) * 1.7320508075688772)
        done = bool(int(self._hash(2, np.uint64(0))) % self.horizon == 0)
        self.ret += reward
        info = {'episodic_return': self.ret if done else None}
        return self.s.copy(), reward, done, info


class DummyVecEnv:
    """envs.py:126-150: serial vector env with auto-reset on done."""

    def __init__(self, envs):
        self.envs = envs
        self.num_envs = len(envs)
        self.actions = None

    def step_async(self, actions):
        self.actions = actions

    def step_wait(self):
        data = []
        for i in range(self.num_envs):
            obs, rew, done, info = self.envs[i].step(self.actions[i])
            if done:
                obs = self.envs[i].reset()
            data.append([obs, rew, done, info])
        obs, rew, done, info = zip(*data)
        return obs, np.asarray(rew), np.asarray(done), info

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def reset(self):
        return [env.reset() for env in self.envs]

    def close(self):
        return


_WARNED = set()


class _PassThrough:
    """Wrapper base that forwards everything it does not define to the wrapped environment (so that gym.Wrapper subclasses
    such as baselines' wrap_deepmind stack can sit on top of it: they read action_space / unwrapped / metadata ...)."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name == 'env':
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)


class OriginalReturnWrapper(_PassThrough):
    """envs.py:58-76: the return of the WHOLE game.  It sits directly on the seeded environment, UNDER wrap_deepmind, so
    that EpisodicLifeEnv's per-life resets do not reset it; reset() does not clear the sum (only a real `done` does)."""

    def __init__(self, env):
        _PassThrough.__init__(self, env)
        self.total_rewards = 0

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.total_rewards += reward
        info = dict(info or {})
        if done:
            info['episodic_return'] = self.total_rewards
            self.total_rewards = 0
        else:
            info['episodic_return'] = None
        return obs, reward, done, info


class TransposeImage(_PassThrough):
    """envs.py:79-91: HWC -> CHW observations (and the observation space's shape)."""

    def __init__(self, env):
        _PassThrough.__init__(self, env)
        sp = env.observation_space
        shp = tuple(sp.shape)
        lo = np.asarray(sp.low).reshape(-1)[0] if hasattr(sp, 'low') else 0
        hi = np.asarray(sp.high).reshape(-1)[0] if hasattr(sp, 'high') else 255
        self.observation_space = Box(lo, hi, (shp[2], shp[1], shp[0]))

    def reset(self, **kw):
        return np.asarray(self.env.reset(**kw)).transpose(2, 0, 1)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        return np.asarray(obs).transpose(2, 0, 1), reward, done, info


class FrameStack(_PassThrough):
    """envs.py:116-123 over baselines' FrameStack: the last k observations as LazyFrames concatenated on AXIS 0 (k frames of
    [1, 84, 84] -> [4, 84, 84]); reset() repeats the first observation k times."""

    def __init__(self, env, k):
        _PassThrough.__init__(self, env)
        from collections import deque
        self.k = k
        self.frames = deque([], maxlen=k)
        shp = tuple(env.observation_space.shape)
        sp = env.observation_space
        lo = np.asarray(sp.low).reshape(-1)[0] if hasattr(sp, 'low') else 0
        hi = np.asarray(sp.high).reshape(-1)[0] if hasattr(sp, 'high') else 255
        self.observation_space = Box(lo, hi, (shp[0] * k,) + shp[1:])

    def _get_ob(self):
        assert len(self.frames) == self.k
        return LazyFrames(list(self.frames))

    def reset(self, **kw):
        ob = self.env.reset(**kw)
        for _ in range(self.k):
            self.frames.append(ob)
        return self._get_ob()

    def step(self, action):
        ob, reward, done, info = self.env.step(action)
        self.frames.append(ob)
        return self._get_ob(), reward, done, info


def wrap_like_reference(env, seed, rank, is_atari, wrap_deepmind=None, episode_life=True):
    """make_env's wrapper stack (envs.py:39-53) around an already-constructed environment, in the reference's ORDER:
    seed -> OriginalReturnWrapper -> [wrap_deepmind(episode_life, clip_rewards=False, frame_stack=False, scale=False) ->
    TransposeImage (3-D observations) -> FrameStack(4)]."""
    env.seed(seed + rank)
    env = OriginalReturnWrapper(env)
    if is_atari:
        env = wrap_deepmind(env, episode_life=episode_life, clip_rewards=False, frame_stack=False, scale=False)
        if len(env.observation_space.shape) == 3:
            env = TransposeImage(env)
        env = FrameStack(env, 4)
    return env


def _real_task_envs(name, num_envs, seed, episode_life):
    """The reference's path (envs.py:27-55: gym.make + baselines' Atari wrappers) when those third-party packages are
    installed; None when they are not (this image: no gym, no baselines, no emulators).  The wrapper stack itself is
    wrap_like_reference (tested here against a stand-in emulator: tests/test_dropin_surface.py)."""
    try:
        import gym
    except ImportError:
        return None
    if not getattr(gym, "__file__", None) or not hasattr(gym, "make"):   # a placeholder module, not an installation
        return None
    try:
        from baselines.common.atari_wrappers import make_atari, wrap_deepmind
    except ImportError:
        make_atari = wrap_deepmind = None
    envs = []
    from .support import random_seed
    for i in range(num_envs):
        random_seed(seed)        # the reference's thunk reseeds np.random / torch for EVERY environment it builds (envs.py:28):
        #                          the global generators' state after Task(...) is part of what a seeded run reproduces
        if name.startswith("dm"):
            import dm_control2gym
            _, domain, task = name.split('-')
            env = dm_control2gym.make(domain_name=domain, task_name=task)
        else:
            env = gym.make(name)
        is_atari = hasattr(gym.envs, 'atari') and isinstance(env.unwrapped, gym.envs.atari.atari_env.AtariEnv)
        if is_atari:
            if make_atari is None:
                raise ImportError("gym is installed but baselines.common.atari_wrappers is not: %s needs the reference's "
                                  "Atari preprocessing (envs.py:39-47)" % name)
            env = make_atari(name)
        envs.append(wrap_like_reference(env, seed, i, is_atari, wrap_deepmind, episode_life))
    return envs


class Task:
    """envs.py:153-189.  Real environments when gym (+ baselines for Atari) is installed -- the reference's own path.
    Otherwise, and for explicit `synthetic-*` names, a SYNTHETIC stand-in with the same surface and observation shapes
    (module docstring); falling back for a real environment name warns loudly once per name (DEEPRL_AMD_STRICT_ENVS=1
    makes it an error): the numbers such a run produces are throughput numbers, not learning curves."""

    def __init__(self, name, num_envs=1, single_process=True, log_dir=None, episode_life=True, seed=None,
                 synthetic_done_period=None):
        if seed is None:
            seed = np.random.randint(int(1e9))
        self.name = name
        explicit = name.startswith('synthetic-')
        real = None if explicit else _real_task_envs(name, num_envs, seed, episode_life)
        if real is not None:
            self.env = DummyVecEnv(real)
            self.observation_space, self.action_space = real[0].observation_space, real[0].action_space
            self.state_dim = int(np.prod(self.observation_space.shape))
            self.action_dim = self.action_space.n if hasattr(self.action_space, 'n') else self.action_space.shape[0]
            return
        if not explicit and name not in _WARNED:
            import os
            import warnings
            msg = ("Task(%r): gym / the emulator for this environment is not installed; using a SYNTHETIC environment with "
                   "the same observation and action shapes (counter-hash frames / random-walk vectors).  Throughput is "
                   "meaningful, returns are not.  Use a 'synthetic-*' name to silence this." % name)
            if os.environ.get("DEEPRL_AMD_STRICT_ENVS") == "1":
                raise ImportError(msg)
            warnings.warn(msg, stacklevel=2)
            _WARNED.add(name)
        atari = 'NoFrameskip' in name or name.startswith('synthetic-atari')
        continuous = any(k in name for k in ('HalfCheetah', 'Walker', 'Hopper', 'Reacher', 'Swimmer', 'Ant', 'Humanoid',
                                             'dm-', 'synthetic-continuous'))
        if atari:
            envs = [SyntheticAtari(seed + i, done_period=synthetic_done_period or 800) for i in range(num_envs)]
            self.observation_space = Box(0, 255, (4, 84, 84))
            self.action_space = Discrete(4)
        elif continuous:
            envs = [SyntheticContinuous(seed + i, 17, 6, horizon=synthetic_done_period or 1000) for i in range(num_envs)]
            self.observation_space = Box(-np.inf, np.inf, (17,))
            self.action_space = Box(-1.0, 1.0, (6,))
        else:
            envs = [SyntheticVector(seed + i, 4, 2, horizon=synthetic_done_period or 200) for i in range(num_envs)]
            self.observation_space = Box(-np.inf, np.inf, (4,))
            self.action_space = Discrete(2)
        self.env = DummyVecEnv(envs)
        self.state_dim = int(np.prod(self.observation_space.shape))
        if isinstance(self.action_space, Discrete):
            self.action_dim = self.action_space.n
        else:
            self.action_dim = self.action_space.shape[0]

    def reset(self):
        return self.env.reset()

    def step(self, actions):
        sp = self.action_space      # envs.py:186-189: Box actions are clipped (gym's Box as well as this module's)
        if isinstance(sp, Box) or (hasattr(sp, 'low') and hasattr(sp, 'high') and not hasattr(sp, 'n')):
            actions = np.clip(actions, sp.low, sp.high)
        return self.env.step(actions)
