"""The reference's own examples.py against the drop-in surface (CPU, no GPU work): with deeprl_amd installed as
`deep_rl`, the file -- modulo the `async` keyword it cannot legally contain on Python >= 3.7 (examples.py:116,149,180,
214) -- executes, and every global name its entry functions for the in-scope algorithms reference is
provided by the package.  Skipped where /root/reference is absent (the GPU box)."""
import builtins
import dis
import os
import re
import types

import pytest

REF = os.environ.get("DEEPRL_REFERENCE_ROOT", "/root/reference")
IN_SCOPE = ["ddpg_continuous", "td3_continuous", "option_critic_feature", "option_critic_pixel", "dqn_feature", "dqn_pixel", "quantile_regression_dqn_feature", "quantile_regression_dqn_pixel",
            "categorical_dqn_feature", "categorical_dqn_pixel", "rainbow_feature", "rainbow_pixel", "a2c_feature",
            "a2c_pixel", "a2c_continuous", "n_step_dqn_feature", "n_step_dqn_pixel", "ppo_continuous", "ppo_pixel"]
OUT_OF_SCOPE_NAMES = set()      # every name examples.py uses is provided


def _global_names(fn):
    names = set()
    todo = [fn.__code__]
    while todo:
        code = todo.pop()
        for ins in dis.get_instructions(code):
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME"):
                names.add(ins.argval)
        todo += [c for c in code.co_consts if isinstance(c, types.CodeType)]
    return names


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "examples.py")), reason="reference tree not present")
def test_reference_examples_resolve_against_the_package():
    import sys
    import deeprl_amd
    saved = {k: v for k, v in sys.modules.items() if k == "deep_rl" or k.startswith("deep_rl.")}
    try:
        deeprl_amd.install_as_deep_rl()
        _check_examples(deeprl_amd)
    finally:                                   # other tests import the REAL reference under the same name
        for k in [k for k in sys.modules if k == "deep_rl" or k.startswith("deep_rl.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def _check_examples(deeprl_amd):
    src = re.sub(r"\basync\b", "async_", open(os.path.join(REF, "examples.py")).read())
    mod = types.ModuleType("ref_examples_dropin")
    exec(compile(src, "examples.py", "exec"), mod.__dict__)     # star-imports deep_rl == deeprl_amd; defines the entry points
    missing = {}
    for name in IN_SCOPE:
        fn = getattr(mod, name, None)
        assert callable(fn), "examples.py has no %s" % name
        lacking = [g for g in _global_names(fn) if g not in mod.__dict__ and not hasattr(builtins, g)]
        if lacking:
            missing[name] = lacking
    assert not missing, "names examples.py needs that the package does not export: %s" % missing
    # the whole file: only the out-of-scope agents may be absent
    every = set()
    for v in mod.__dict__.values():
        if isinstance(v, types.FunctionType) and v.__module__ == mod.__name__:
            every |= {g for g in _global_names(v) if g not in mod.__dict__ and not hasattr(builtins, g)}
    assert every <= OUT_OF_SCOPE_NAMES, "unexpected missing names: %s" % sorted(every - OUT_OF_SCOPE_NAMES)
    for name in ("OptionCriticAgent", "DDPGAgent", "TD3Agent"):      # real agents (tests/test_gpu_more_agents.py)
        assert issubclass(getattr(deeprl_amd, name), deeprl_amd.BaseAgent)


def test_replay_wrapper_accepts_every_spelling_of_the_async_flag():
    """replay.py:205: the third argument is called `async` in the reference; callers pass it positionally
    (examples.py:41,84), as `async_=` after the rename, or through **{'async': ...}."""
    from deeprl_amd.replay import ReplayWrapper, UniformReplay
    kw = dict(memory_size=16, batch_size=2)
    for w in (ReplayWrapper(UniformReplay, kw, True), ReplayWrapper(UniformReplay, kw, async_=False),
              ReplayWrapper(UniformReplay, kw, **{"async": True})):
        assert isinstance(w.replay, UniformReplay) and w.size() == 0
    with pytest.raises(TypeError):
        ReplayWrapper(UniformReplay, kw, bogus=1)


def test_real_env_wrapper_stack_mirrors_make_env():
    """ADVICE r2 (envs.py real-environment path): deeprl_amd.envs.wrap_like_reference must reproduce the ORDER and the
    observation layout of the reference's make_env (deep_rl/component/envs.py:27-55) -- checked here with a stand-in emulator
    (gym is not installed in this image): whole-game returns survive per-life resets (the return wrapper sits UNDER
    wrap_deepmind), observations are CHW LazyFrames of shape (4, 84, 84), Task.step clips gym-style Box actions."""
    import numpy as np
    from deeprl_amd import envs as E

    class Emu:
        """3 lives, one life lost every 5 steps, reward 1 per step, HWC uint8 frames."""
        class Space:
            shape, low, high = (84, 84, 1), np.zeros((84, 84, 1)), np.full((84, 84, 1), 255.0)
        observation_space, action_space = Space(), E.Discrete(4)

        def __init__(self):
            self.seeded, self.t, self.lives = None, 0, 3
            self.unwrapped = self

        def seed(self, s):
            self.seeded = s

        def _obs(self):
            return np.full((84, 84, 1), self.t % 256, dtype=np.uint8)

        def reset(self):
            self.t, self.lives = 0, 3
            return self._obs()

        def step(self, a):
            self.t += 1
            if self.t % 5 == 0:
                self.lives -= 1
            return self._obs(), 1.0, self.lives == 0, {}

    class EpisodicLife(E._PassThrough):
        """baselines' EpisodicLifeEnv in miniature: a lost life ends the episode for the agent; reset() only restarts the
        game when it is really over."""
        def __init__(self, env):
            E._PassThrough.__init__(self, env)
            self.lives, self.real_done = 0, True

        def step(self, a):
            obs, r, done, info = self.env.step(a)
            self.real_done = done
            lives = self.env.unwrapped.lives
            if 0 < lives < self.lives:
                done = True
            self.lives = lives
            return obs, r, done, info

        def reset(self):
            if self.real_done:
                obs = self.env.reset()
            else:
                obs, _, _, _ = self.env.step(0)
            self.lives = self.env.unwrapped.lives
            return obs

    calls = {}

    def fake_wrap_deepmind(env, episode_life, clip_rewards, frame_stack, scale):
        calls.update(episode_life=episode_life, clip_rewards=clip_rewards, frame_stack=frame_stack, scale=scale,
                     under=type(env).__name__)
        return EpisodicLife(env) if episode_life else env

    emu = Emu()
    env = E.wrap_like_reference(emu, seed=7, rank=2, is_atari=True, wrap_deepmind=fake_wrap_deepmind, episode_life=True)
    assert emu.seeded == 9
    assert calls == dict(episode_life=True, clip_rewards=False, frame_stack=False, scale=False, under="OriginalReturnWrapper")
    assert tuple(env.observation_space.shape) == (4, 84, 84)
    ob = env.reset()
    assert isinstance(ob, E.LazyFrames) and np.asarray(ob).shape == (4, 84, 84)
    returns, steps = [], 0
    while len(returns) < 1 and steps < 100:
        ob, r, done, info = env.step(1)
        steps += 1
        if info['episodic_return'] is not None:
            returns.append(info['episodic_return'])
        if done:
            ob = env.reset()
    # the game ends when the third life is lost: 15 emulator steps of reward 1 (+1 no-op step per life-reset) -- ONE return
    # for the whole game, not one per life
    assert returns and returns[0] >= 15
    a = np.asarray(ob)
    assert a.shape == (4, 84, 84) and a.dtype == np.uint8
    # Task.step clips Box actions of a duck-typed (gym) space too
    class GymBox:
        low, high, shape = np.full(2, -1.0), np.full(2, 1.0), (2,)
    t = E.Task.__new__(E.Task)
    t.action_space = GymBox()
    seen = {}

    class Rec:
        def step(self, actions):
            seen['a'] = actions
            return None
    t.env = Rec()
    t.step(np.array([[3.0, -2.0]]))
    assert np.array_equal(seen['a'], np.array([[1.0, -1.0]]))


def test_real_env_construction_reseeds_like_the_reference_thunk(monkeypatch):
    """make_env's thunk starts with random_seed(seed) (deep_rl/component/envs.py:28) -- np.random and torch are
    reseeded for EVERY environment built, so the global generators' state after Task(...) is that of a fresh
    random_seed(seed), whatever was drawn before.  Checked with a stand-in `gym` (the image has none)."""
    import sys
    import types
    import numpy as np
    import torch
    from deeprl_amd import envs as E
    from deeprl_amd.support import random_seed

    class Space:
        shape, low, high = (3,), np.zeros(3), np.ones(3)

    class Emu:
        observation_space, action_space = Space(), E.Discrete(2)

        def __init__(self):
            self.unwrapped, self.seeded = self, None
            np.random.rand(5)                   # a constructor that consumes the global stream (as emulators do)

        def seed(self, s):
            self.seeded = s

        def reset(self):
            return np.zeros(3)

        def step(self, a):
            return np.zeros(3), 0.0, False, {}

    gym = types.ModuleType("gym")
    gym.__file__ = "/nonexistent/gym/__init__.py"
    gym.make = lambda name: Emu()
    gym.envs = types.SimpleNamespace()
    monkeypatch.setitem(sys.modules, "gym", gym)
    np.random.seed(123)
    np.random.rand(7)
    built = E._real_task_envs("Fake-v0", 3, 11, True)
    assert built is not None and len(built) == 3
    got_np, got_torch = np.random.get_state(), torch.get_rng_state()
    random_seed(11)
    np.random.rand(5)                           # the last environment's constructor ran after the last reseed
    want_np, want_torch = np.random.get_state(), torch.get_rng_state()
    assert got_np[0] == want_np[0] and np.array_equal(got_np[1], want_np[1]) and got_np[2:] == want_np[2:]
    assert torch.equal(got_torch, want_torch)
    e = built[2]
    while hasattr(e, "env"):
        e = e.env
    assert e.seeded == 11 + 2


def test_block_hashed_emulator_equals_the_per_step_definition():
    """SyntheticAtari hashes its frames and (reward, done) pairs 256 counters at a time (so that a host-environment run measures
    the pipeline and not numpy's per-call overhead); every observation, reward and `done` must equal the per-step definition
    (synthetic_frame / synthetic_reward_done: the bytes the device environment writes), across block boundaries and the
    resets DummyVecEnv performs."""
    import numpy as np
    from deeprl_amd.envs import DummyVecEnv, SyntheticAtari, synthetic_frame, synthetic_reward_done
    seed, period = 5, 37
    env = DummyVecEnv([SyntheticAtari(seed=seed, done_period=period)])
    obs = env.reset()
    counter, stack = 1, [0, 0, 0, 0]                 # frame counters of the observation's four frames
    n_done = 0
    for t in range(3 * SyntheticAtari.BLOCK + 11):
        o = np.asarray(obs[0])
        for j in range(4):
            assert np.array_equal(o[j].reshape(-1), synthetic_frame(stack[j], seed)), (t, j)
        obs, rew, done, info = env.step([t % 4])
        want_r, want_d = synthetic_reward_done(counter, seed, period)
        assert rew[0] == want_r and bool(done[0]) == want_d, t
        stack = stack[1:] + [counter]
        counter += 1
        if want_d:                                   # auto-reset (envs.py:136-137): a fresh stack of the next frame
            stack = [counter] * 4
            counter += 1
            n_done += 1
    assert n_done >= 5
