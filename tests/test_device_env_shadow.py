"""Host side of the device-resident environments (CPU): the vectorised episode shadow of device_env.DeviceAtariVec equals
N scalar learner.SyntheticEpisodeStream objects -- which the GPU tests pin, transition for transition, to
envs.SyntheticAtari + DummyVecEnv's auto-reset -- and SyntheticEpisodeStream equals the host emulator it shadows."""
import numpy as np


def test_vec_shadow_equals_scalar_streams():
    from deeprl_amd.device_env import VecEpisodeShadow
    from deeprl_amd.learner import SyntheticEpisodeStream
    seeds, c0, dp = [5, 6, 7, 1005], [0, 0, 3, 0], [7, 9, 800, 5]
    scalar = [SyntheticEpisodeStream(s, c, d, 4) for s, c, d in zip(seeds, c0, dp)]
    vec = VecEpisodeShadow(seeds, c0, dp, 4)
    for t in range(400):
        c, age, rew, done, ep = vec.step()
        for e, s in enumerate(scalar):
            cc, _, aa, r, d, info = s.transition()
            assert (cc, aa, r, bool(d)) == (int(c[e]), int(age[e]), float(rew[e]), bool(done[e])), (t, e)
            assert (info['episodic_return'] is None) == (not done[e])
            if done[e]:
                assert info['episodic_return'] == ep[e]
    for e, s in enumerate(scalar):
        if s.c is None:
            s._reset()
        assert (s.c, s.age) == (int(vec.c[e]), int(vec.age[e]))


def test_scalar_stream_equals_host_emulator():
    """counter / age of every observation reproduce the frame stack SyntheticAtari + DummyVecEnv hand to the agent."""
    from deeprl_amd.envs import DummyVecEnv, SyntheticAtari, synthetic_frame
    from deeprl_amd.learner import SyntheticEpisodeStream
    env = DummyVecEnv([SyntheticAtari(seed=3, done_period=6)])
    s = SyntheticEpisodeStream(3, 0, 6, 4)
    obs = env.reset()
    for t in range(60):
        c, _, age, r, d, info = s.transition()
        stack = np.asarray(obs[0])
        for j in range(4):
            want = synthetic_frame(c - min(3 - j, age), 3).reshape(84, 84)
            assert np.array_equal(stack[j], want), (t, j)
        obs, rew, done, infos = env.step([0])
        assert (float(rew[0]), bool(done[0])) == (r, bool(d))
        assert infos[0]['episodic_return'] == info['episodic_return']
