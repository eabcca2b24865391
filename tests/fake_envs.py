"""Deterministic stand-in environments with the reference `Task` surface
(deep_rl/component/envs.py:153-189: reset(), step(actions) -> (obs, reward, done,
info), state_dim, action_dim, name).  TEST TOOLING: used identically by the golden
generator (driving the reference agents) and by the parity tests (driving ours).
Every stream comes from a private RandomState so the global np.random stream is
left to the code under test, as in the reference."""
import numpy as np


class VectorTask:
    """CartPole-like: f64[state_dim] random-walk state, `action_dim` discrete
    actions, reward 1 per step, done w.p. 1/horizon (auto-reset like
    DummyVecEnv.step_wait, envs.py:136-144)."""

    def __init__(self, seed=0, state_dim=4, action_dim=2, horizon=50, num_envs=1, name="FakeCartPole"):
        self.rs = np.random.RandomState(seed)
        self.state_dim, self.action_dim, self.name = state_dim, action_dim, name
        self.horizon, self.num_envs = horizon, num_envs
        self.s = None
        self.ret = np.zeros(num_envs)

    def reset(self):
        self.s = self.rs.uniform(-0.05, 0.05, size=(self.num_envs, self.state_dim))
        self.ret[:] = 0
        return self.s.copy()

    def step(self, actions):
        actions = np.asarray(actions).reshape(self.num_envs)
        drift = (actions.astype(np.float64) - (self.action_dim - 1) / 2.0)[:, None] * 0.01
        self.s = self.s + drift + self.rs.uniform(-0.02, 0.02, size=self.s.shape)
        rew = np.ones(self.num_envs)
        done = self.rs.rand(self.num_envs) < 1.0 / self.horizon
        self.ret += rew
        infos = []
        for e in range(self.num_envs):
            if done[e]:
                infos.append({"episodic_return": float(self.ret[e])})
                self.ret[e] = 0
                self.s[e] = self.rs.uniform(-0.05, 0.05, size=self.state_dim)
            else:
                infos.append({"episodic_return": None})
        return self.s.copy(), rew, done, tuple(infos)


class PixelVectorTask:
    """`num_envs` synthetic Atari emulators (counter-hash 84x84 uint8 frames, history 4; deeprl_amd.envs.SyntheticAtari is a
    pure numpy host emulator) behind the Task surface with DummyVecEnv's auto-reset: states uint8 [N,4,84,84]."""

    def __init__(self, seed=0, num_envs=4, done_period=9, n_actions=4, name="FakePixels"):
        from deeprl_amd.envs import SyntheticAtari
        self.envs = [SyntheticAtari(seed=seed + 1000 * e, history=4, n_actions=n_actions, done_period=done_period)
                     for e in range(num_envs)]
        self.state_dim, self.action_dim, self.name, self.num_envs = (4, 84, 84), n_actions, name, num_envs

    def reset(self):
        return np.stack([np.asarray(e.reset()) for e in self.envs])

    def step(self, actions):
        actions = np.asarray(actions).reshape(self.num_envs)
        obs, rew, done, infos = [], [], [], []
        for e, a in zip(self.envs, actions):
            o, r, d, i = e.step(int(a))
            if d:
                o = e.reset()
            obs.append(np.asarray(o)); rew.append(r); done.append(d); infos.append(i)
        return np.stack(obs), np.asarray(rew, dtype=np.float64), np.asarray(done), tuple(infos)


class BoxSpace:
    """gym.spaces.Box surface the continuous-control agents use (low / high / sample()); sample() draws from a private
    stream like gym's own space RNG, never from the global np.random."""

    def __init__(self, dim, seed):
        self.low, self.high, self.shape = -np.ones(dim), np.ones(dim), (dim,)
        self.rs = np.random.RandomState(seed + 4242)

    def sample(self):
        return self.rs.uniform(self.low, self.high)


class ContinuousTask:
    """HalfCheetah-like: f64[state_dim] ~ N(0,1) observations, continuous actions,
    reward ~ N(0,1), done w.p. 1/horizon."""

    def __init__(self, seed=0, state_dim=17, action_dim=6, horizon=1000, num_envs=1, name="FakeCheetah"):
        self.rs = np.random.RandomState(seed)
        self.state_dim, self.action_dim, self.name = state_dim, action_dim, name
        self.horizon, self.num_envs = horizon, num_envs
        self.ret = np.zeros(num_envs)
        self.action_space = BoxSpace(action_dim, seed)

    def reset(self):
        self.ret[:] = 0
        return self.rs.randn(self.num_envs, self.state_dim)

    def step(self, actions):
        actions = np.asarray(actions, dtype=np.float64).reshape(self.num_envs, -1)
        obs = self.rs.randn(self.num_envs, self.state_dim) + 0.1 * actions.sum(-1, keepdims=True)
        rew = self.rs.randn(self.num_envs)
        done = self.rs.rand(self.num_envs) < 1.0 / self.horizon
        self.ret += rew
        infos = []
        for e in range(self.num_envs):
            if done[e]:
                infos.append({"episodic_return": float(self.ret[e])})
                self.ret[e] = 0
            else:
                infos.append({"episodic_return": None})
        return obs, rew, done, tuple(infos)


class PixelTask:
    """Atari-like: uint8 [history, side, side] stacked observations whose newest
    frame is pseudo-random, `action_dim` discrete actions, reward in {-2,0,3}
    (so SignNormalizer matters), done w.p. 1/horizon."""

    def __init__(self, seed=0, side=84, history=4, action_dim=4, horizon=200, num_envs=1, name="FakeBreakout"):
        self.rs = np.random.RandomState(seed)
        self.side, self.history, self.action_dim, self.name = side, history, action_dim, name
        self.state_dim = (history, side, side)
        self.horizon, self.num_envs = horizon, num_envs
        self.stack = None
        self.ret = np.zeros(num_envs)

    def _frame(self):
        return self.rs.randint(0, 256, size=(self.num_envs, self.side, self.side)).astype(np.uint8)

    def reset(self):
        f = self._frame()
        self.stack = np.repeat(f[:, None], self.history, axis=1)
        self.ret[:] = 0
        return self.stack.copy()

    def step(self, actions):
        f = self._frame()
        self.stack = np.concatenate([self.stack[:, 1:], f[:, None]], axis=1)
        rew = self.rs.choice([-2.0, 0.0, 3.0], size=self.num_envs, p=[0.1, 0.8, 0.1])
        done = self.rs.rand(self.num_envs) < 1.0 / self.horizon
        self.ret += rew
        infos = []
        for e in range(self.num_envs):
            if done[e]:
                infos.append({"episodic_return": float(self.ret[e])})
                self.ret[e] = 0
                self.stack[e] = np.repeat(self._frame()[e][None], self.history, axis=0)
            else:
                infos.append({"episodic_return": None})
        return self.stack.copy(), rew, done, tuple(infos)


def numpy_params(shapes, seed, scale=0.05):
    """Version-stable parameter generator (legacy RandomState) so fixtures need to
    store only a seed: dict name -> float32 array."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        s = scale if len(shape) == 1 else 1.0 / np.sqrt(fan_in)
        out[name] = (rs.standard_normal(shape) * s).astype(np.float32)
    return out


NATURE_SHAPES = [
    ("body.conv1.weight", (32, 4, 8, 8)), ("body.conv1.bias", (32,)),
    ("body.conv2.weight", (64, 32, 4, 4)), ("body.conv2.bias", (64,)),
    ("body.conv3.weight", (64, 64, 3, 3)), ("body.conv3.bias", (64,)),
    ("body.fc4.weight", (512, 3136)), ("body.fc4.bias", (512,)),
]


def NATURE_SHAPES_PREFIXED(prefix):
    """NatureConvBody's tensors under another module path (e.g. 'network.phi_body.' of the actor-critic nets)."""
    return [(prefix + name[len("body."):], shape) for name, shape in NATURE_SHAPES]


def nature_vanilla_shapes(action_dim):
    return NATURE_SHAPES + [("fc_head.weight", (action_dim, 512)), ("fc_head.bias", (action_dim,))]


def fc_vanilla_shapes(state_dim, action_dim, hidden=(64, 64)):
    dims = (state_dim,) + tuple(hidden)
    out = []
    for i in range(len(hidden)):
        out += [("body.layers.%d.weight" % i, (dims[i + 1], dims[i])), ("body.layers.%d.bias" % i, (dims[i + 1],))]
    return out + [("fc_head.weight", (action_dim, dims[-1])), ("fc_head.bias", (action_dim,))]
