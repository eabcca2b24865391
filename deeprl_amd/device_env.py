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

    def step(self, actions):
        raise RuntimeError("DeviceAtariVec is driven through plan() / states(); its environments are not steppable one by one")

    def close(self):
        return
