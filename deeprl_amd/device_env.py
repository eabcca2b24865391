"""Device-resident vector environment for the on-policy agents (SURVEY.md section 8 f1 for A2C / PPO).

The synthetic Atari environment is a pure function of its frame counter (envs.SyntheticAtari: counter-hash frames,
rewards and terminals), so N of them need no state on the device at all: the host shadow (learner.SyntheticEpisodeStream,
one per environment, the same class the DQN device pipeline uses) lays out a whole rollout ahead of time -- per step and
environment the counter of the observation, its episode age, the reward and the terminal flag -- uploads that plan with
one copy, and `states(plan, t)` produces the uint8 [N,4,84,84] observations of step t with one kernel
(dra_synth_stacks).  Nothing crosses the host boundary inside a rollout: no D2H of actions, no H2D of frames; the
observations go through the same image normaliser table (f32(f64(v) / 255)) as host frames, so the arithmetic an agent
performs is bit-identical to the host-emulator path.
"""
import ctypes
from collections import namedtuple

import numpy as np
import torch

from ._lib import lib, stream_ptr
from .support import Config

Plan = namedtuple("Plan", ["counters", "ages", "reward", "mask", "infos", "t_len"])


class VecEpisodeShadow:
    """N learner.SyntheticEpisodeStream objects advanced in lockstep with numpy vector arithmetic (one python iteration per
    rollout STEP instead of one per transition: the scalar shadow costs ~20 us per transition, 20 ms of a PPO rollout of
    1024).  Same state per environment: the counter of the current observation (c), its episode age, the next counter to
    hand out, the running return."""

    def __init__(self, seeds, counters0, done_periods, history):
        n = len(seeds)
        self.seeds = np.asarray(seeds, dtype=np.int64)
        self.done_periods = np.asarray(done_periods, dtype=np.int64)
        self.history = int(history)
        self.next_counter = np.asarray(counters0, dtype=np.int64).copy()
        self.c = self.next_counter.copy()           # reset(): one new frame per environment
        self.next_counter += 1
        self.age = np.zeros(n, dtype=np.int32)
        self.ret = np.zeros(n, dtype=np.float64)

    def step(self):
        """One transition of every environment -> (counter, age, reward, done, episodic_return-or-nan) arrays."""
        from .envs import synthetic_reward_done_vec
        counter, age = self.c.copy(), self.age.copy()
        rc = self.next_counter.copy()               # reward / done are hashed from the counter of the frame step() generates
        reward, done = synthetic_reward_done_vec(rc, self.seeds, self.done_periods)
        self.next_counter += 1
        self.ret += reward
        ep_ret = np.where(done, self.ret, np.nan)
        # done: DummyVecEnv drops the post-step frame and resets (a new frame, the stack restarts, the return too)
        self.c = np.where(done, self.next_counter, rc)
        self.next_counter += done.astype(np.int64)
        self.age = np.where(done, 0, np.minimum(age + 1, self.history - 1)).astype(np.int32)
        self.ret = np.where(done, 0.0, self.ret)
        return counter, age, reward, done, ep_ret


class DeviceAtariVec:
    """Task surface (state_dim / action_dim / action_space ...) over the environments of `task`, observations on device."""
    on_device = True

    def __init__(self, task):
        envs = task.env.envs
        self.task = task
        self.name, self.state_dim, self.action_dim = task.name, task.state_dim, task.action_dim
        self.observation_space, self.action_space = task.observation_space, task.action_space
        self.num_envs, self.history = len(envs), envs[0].history
        self.shadow = VecEpisodeShadow([e.seed for e in envs], [e.counter for e in envs], [e.done_period for e in envs],
                                       envs[0].history)
        for e in envs:
            e.frames = "device"         # the host emulators are retired: stepping them too would fork the streams
        dev = Config.DEVICE
        self.seeds = torch.tensor([e.seed for e in envs], dtype=torch.int64, device=dev)
        self._bufs = {}
        self._no_info = tuple({'episodic_return': None} for _ in envs)

    @staticmethod
    def eligible(task, config):
        """`task` is an envs.Task over fresh synthetic Atari emulators and the configuration can consume device frames."""
        from .envs import DummyVecEnv, SyntheticAtari, Task
        from .normalizers import RescaleNormalizer
        if Config.DEVICE.type != 'cuda' or getattr(config, 'device_env', True) is False or type(task) is not Task:
            return False
        env = getattr(task, 'env', None)
        if type(env) is not DummyVecEnv or not env.envs:
            return False
        if not all(type(e) is SyntheticAtari and e.frames is None and e.history == env.envs[0].history for e in env.envs):
            return False
        return isinstance(config.state_normalizer, RescaleNormalizer)

    def reset(self):
        return None     # observations come from states(plan, t); the first one is every stream's initial reset

    def _static(self, t_len):
        """Persistent device buffers + rotating pinned staging of one rollout plan (graph replays read the same addresses)."""
        if t_len not in self._bufs:
            n, dev = self.num_envs, Config.DEVICE
            words = (t_len + 1) * n * 2 + (t_len + 1) * n + 2 * t_len * n      # i64 counters | i32 ages | f32 reward, mask
            self._bufs[t_len] = dict(
                dev=torch.zeros(words * 4 + 16, dtype=torch.uint8, device=dev),
                stage=[torch.zeros(words * 4 + 16, dtype=torch.uint8).pin_memory() for _ in range(4)], events=[None] * 4, k=0)
        return self._bufs[t_len]

    def plan(self, t_len, reward_normalizer):
        """Advances every environment by t_len transitions on the host shadow and uploads what the device needs."""
        n = self.num_envs
        b = self._static(t_len)
        k = b['k']
        b['k'] = (k + 1) % len(b['stage'])
        if b['events'][k] is not None:
            b['events'][k].synchronize()
        raw = b['stage'][k].numpy()
        o1 = (t_len + 1) * n * 8
        o2 = o1 + (t_len + 1) * n * 4
        o3 = o2 + t_len * n * 4
        counters = raw[:o1].view(np.int64).reshape(t_len + 1, n)
        ages = raw[o1:o2].view(np.int32).reshape(t_len + 1, n)
        reward = raw[o2:o3].view(np.float32).reshape(t_len, n)
        mask = raw[o3:o3 + t_len * n * 4].view(np.float32).reshape(t_len, n)
        infos = []
        sh = self.shadow
        for t in range(t_len):
            c, age, rew, done, ep_ret = sh.step()
            counters[t], ages[t], mask[t] = c, age, 1.0 - done
            reward[t] = np.asarray(reward_normalizer(rew), dtype=np.float32)       # what tensor(rewards) uploads
            infos.append(tuple({'episodic_return': float(r) if d else None} for r, d in zip(ep_ret, done)) if done.any()
                         else self._no_info)
        counters[t_len], ages[t_len] = sh.c, sh.age   # the observation the NEXT rollout starts from (bootstrap value)
        d = b['dev']
        d.copy_(b['stage'][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        b['events'][k] = ev
        return Plan(counters=d[:o1].view(torch.int64).view(t_len + 1, n), ages=d[o1:o2].view(torch.int32).view(t_len + 1, n),
                    reward=d[o2:o3].view(torch.float32).view(t_len, n, 1), mask=d[o3:o3 + t_len * n * 4].view(torch.float32).view(t_len, n, 1),
                    infos=infos, t_len=t_len)

    def states(self, plan, t):
        """uint8 [N, history, 84, 84] observations of rollout step t (t = t_len: the bootstrap observation)."""
        out = torch.empty((self.num_envs, self.history, 84, 84), dtype=torch.uint8, device=Config.DEVICE)
        lib.dra_synth_stacks(ctypes.c_void_p(plan.counters[t].data_ptr()), ctypes.c_void_p(plan.ages[t].data_ptr()),
                             ctypes.c_void_p(self.seeds.data_ptr()), self.num_envs, self.history,
                             ctypes.c_void_p(out.data_ptr()), stream_ptr())
        return out

    def states_all(self, plan):
        """uint8 [t_len + 1, N, history, 84, 84]: the observations of every step of the planned rollout (row t_len: the
        bootstrap observation) from ONE launch -- the environment is a pure function of the planned counters, so nothing orders
        observation t + 1 behind action t; states(plan, t) row by row costs a launch per rollout step."""
        rows = (plan.t_len + 1) * self.num_envs
        key = ('seeds', plan.t_len)
        if key not in self._bufs:
            self._bufs[key] = self.seeds.repeat(plan.t_len + 1).contiguous()
        out = torch.empty((plan.t_len + 1, self.num_envs, self.history, 84, 84), dtype=torch.uint8, device=Config.DEVICE)
        lib.dra_synth_stacks(ctypes.c_void_p(plan.counters.data_ptr()), ctypes.c_void_p(plan.ages.data_ptr()),
                             ctypes.c_void_p(self._bufs[key].data_ptr()), rows, self.history, ctypes.c_void_p(out.data_ptr()),
                             stream_ptr())
        return out

    def step(self, actions):
        raise RuntimeError("DeviceAtariVec is driven through plan() / states(); its environments are not steppable one by one")

    def close(self):
        return


class DeviceContinuousVec:
    """The SyntheticContinuous environments of a Task, resident on the device (BASELINE configs[2]: PPO, 16 HalfCheetah-shaped
    workers): raw fp64 observations, step counters and the observation normaliser's running statistics live in HBM and one
    launch (dra_ppo_mlp_rollout) walks a whole rollout -- normalise, both forwards, sample, environment step -- without a host
    round trip.  Rewards and terminals are hashes of (seed, counter) alone, so the host keeps a shadow of them (vectorised,
    one pass per rollout) for the episodic-return log lines; observations depend on the actions and exist only on the device.
    Same arithmetic as envs.SyntheticContinuous + normalizers.MeanStdNormalizer stepped from python (csrc/cont_env.h;
    tests/test_gpu_ppo_mlp.py compares the two paths)."""
    on_device = True

    def __init__(self, task, agent_states, normalizer):
        """task: the envs.Task whose (already reset) environments move to the device; agent_states: the normalised observations
        the agent holds (what config.state_normalizer(task.reset()) returned); normalizer: config.state_normalizer."""
        from .normalizers import MeanStdNormalizer
        envs = task.env.envs
        dev = Config.DEVICE
        self.task = task
        self.name, self.state_dim, self.action_dim = task.name, task.state_dim, task.action_dim
        self.observation_space, self.action_space = task.observation_space, task.action_space
        self.num_envs, self.horizon = len(envs), int(envs[0].horizon)
        self.seeds_host = np.asarray([e.seed for e in envs], dtype=np.int64)
        self.counters_host = np.asarray([e.c for e in envs], dtype=np.int64)
        self.ret_host = np.asarray([e.ret for e in envs], dtype=np.float64)
        self.env_state = torch.from_numpy(np.stack([e.s for e in envs]).astype(np.float64)).to(dev)
        self.env_counter = torch.from_numpy(self.counters_host.copy()).to(dev)
        self.env_seed = torch.from_numpy(self.seeds_host.copy()).to(dev)
        self.cur_state = torch.from_numpy(np.ascontiguousarray(np.asarray(agent_states, dtype=np.float32))).to(dev)
        s = self.state_dim
        rms = np.zeros(2 * s + 1, dtype=np.float64)
        self.normalizer = normalizer
        if isinstance(normalizer, MeanStdNormalizer):
            rms[:s], rms[s:2 * s], rms[2 * s] = normalizer.rms.mean.reshape(-1), normalizer.rms.var.reshape(-1), normalizer.rms.count
            self.rms_epsilon, self.rms_clip, self.rms_kind = float(normalizer.epsilon), float(normalizer.clip), 'meanstd'
        else:       # RescaleNormalizer(1.0): (x - 0) / sqrt(1 + 0) = x exactly, no clipping, no statistics
            rms[s:2 * s] = 1.0
            self.rms_epsilon, self.rms_clip, self.rms_kind = 0.0, float('inf'), 'identity'
        self.rms = torch.from_numpy(rms).to(dev)
        if self.rms_kind == 'meanstd':
            normalizer.attach_device(self.rms, s)
        for e in envs:
            e.s = "device"          # the host objects are retired: stepping them too would fork the streams
        self._bufs = {}

    @staticmethod
    def eligible(task, config):
        from .envs import DummyVecEnv, SyntheticContinuous, Task
        from .normalizers import MeanStdNormalizer, RescaleNormalizer
        if Config.DEVICE.type != 'cuda' or getattr(config, 'device_env', True) is False or type(task) is not Task:
            return False
        env = getattr(task, 'env', None)
        if type(env) is not DummyVecEnv or not env.envs or len(env.envs) > 64:
            return False
        e0 = env.envs[0]
        if not all(type(e) is SyntheticContinuous and e.horizon == e0.horizon and e.state_dim == e0.state_dim and
                   e.action_dim == e0.action_dim for e in env.envs):
            return False
        sn, rn = config.state_normalizer, config.reward_normalizer
        state_ok = type(sn) is MeanStdNormalizer or (type(sn) is RescaleNormalizer and sn.coef == 1.0)
        return state_ok and type(rn) is RescaleNormalizer

    def buffers(self, t_len):
        if t_len not in self._bufs:
            n, s, a, dev = self.num_envs, self.state_dim, self.action_dim, Config.DEVICE
            f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
            self._bufs[t_len] = dict(state=f(t_len, n, s), action=f(t_len, n, a), log_pi_a=f(t_len, n, 1), v=f(t_len + 1, n, 1),
                                     reward=f(t_len, n, 1), mask=f(t_len, n, 1))
        return self._bufs[t_len]

    def shadow(self, t_len):
        """Advances the host shadow by t_len steps of every environment -> list of (t, env, episodic_return) for the episodes
        that end inside the rollout (envs.SyntheticContinuous.step's reward / done / return arithmetic, vectorised)."""
        from .envs import _GOLD, _mix64
        n = self.num_envs
        c = (self.counters_host[None, :] + np.arange(1, t_len + 1, dtype=np.int64)[:, None]).astype(np.uint64)     # [T, N]
        with np.errstate(over="ignore"):
            sd = self.seeds_host.astype(np.uint64)[None, :] * np.uint64(8)
            base_r = (sd + np.uint64(2)) * _GOLD + c * np.uint64(64)
            u = [(_mix64(base_r + np.uint64(j)) >> np.uint64(11)).astype(np.float64) * 1.1102230246251565e-16 for j in range(4)]
            done = (_mix64((sd + np.uint64(3)) * _GOLD + c * np.uint64(64)) % np.uint64(self.horizon)) == 0
        reward = (((u[0] + u[1]) + (u[2] + u[3])) - 2.0) * 1.7320508075688772
        events = []
        for i in range(n):
            ends = np.nonzero(done[:, i])[0]
            start, carry = 0, self.ret_host[i]
            for t in ends:
                carry = float(np.cumsum(np.concatenate([[carry], reward[start:t + 1, i]]))[-1])
                events.append((int(t), i, carry))
                start, carry = int(t) + 1, 0.0
            self.ret_host[i] = float(np.cumsum(np.concatenate([[carry], reward[start:, i]]))[-1])
        self.counters_host += t_len
        events.sort()
        return events

    def reset(self):
        return None

    def step(self, actions):
        raise RuntimeError("DeviceContinuousVec is driven through PPOAgent's device rollout; its environments are not steppable")

    def close(self):
        return
