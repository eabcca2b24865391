"""CPU checks of the ppo_mlp path's test infrastructure and host logic (no GPU): the oracle against the reference-generated
golden run, the three statements of the synthetic continuous environment against each other (python host class, independent
oracle, csrc/cont_env.h compiled for the host), the host shadow of the device environment, the normaliser restatements."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_KEYS_A = ["actor_body.layers.0.weight", "actor_body.layers.0.bias", "actor_body.layers.1.weight", "actor_body.layers.1.bias",
           "fc_action.weight", "fc_action.bias", "std"]
_KEYS_C = ["critic_body.layers.0.weight", "critic_body.layers.0.bias", "critic_body.layers.1.weight", "critic_body.layers.1.bias",
           "fc_critic.weight", "fc_critic.bias"]
_SHORT = ["w1", "b1", "w2", "b2", "w3", "b3", "std"]


@pytest.mark.parametrize("tag", ["t64n2", "t32n4"])
def test_ppo_mlp_oracle_reproduces_reference_run(golden, tag):
    """oracle.ppo_mlp_oracle.ppo_update on the rollout entries of the reference's own PPOAgent.step() (tests/golden/ppo_step.npz,
    written by make_golden.py from /root/reference) ends on the reference's parameters: pins the oracle the GPU tests use."""
    from collections import OrderedDict
    from oracle import ppo_mlp_oracle as O
    g = golden("ppo_step")
    k = tag + "_"
    gamma, tau, ew, clip, target_kl, epochs, mb, t_len, n_env = g[k + "cfg"]
    actor = OrderedDict((s, torch.from_numpy(g[k + "init_" + n].copy()).requires_grad_(True)) for s, n in zip(_SHORT, _KEYS_A))
    critic = OrderedDict((s, torch.from_numpy(g[k + "init_" + n].copy()).requires_grad_(True)) for s, n in zip(_SHORT, _KEYS_C))
    entries = [torch.from_numpy(g[k + n]) for n in ("ent_state", "ent_action", "ent_log_pi_a", "ent_ret", "ent_adv_normalized")]
    n = entries[0].shape[0]
    np.random.seed(21)
    perms = [np.random.permutation(np.arange(n)) for _ in range(int(epochs))]
    O.ppo_update(actor, critic, entries, perms, int(mb), clip, ew, target_kl)
    for s, name in zip(_SHORT, _KEYS_A):
        np.testing.assert_allclose(actor[s].detach().numpy(), g[k + "final_" + name], rtol=1e-6, atol=1e-7, err_msg=name)
    for s, name in zip(_SHORT, _KEYS_C):
        np.testing.assert_allclose(critic[s].detach().numpy(), g[k + "final_" + name], rtol=1e-6, atol=1e-7, err_msg=name)


def test_continuous_env_host_class_equals_oracle():
    """deeprl_amd.envs.SyntheticContinuous (vectorised numpy, what Task builds) and oracle.ContinuousEnvOracle (scalar python)
    produce the same observations, rewards and terminals bit for bit, resets included."""
    from deeprl_amd.envs import DummyVecEnv, SyntheticContinuous
    from oracle.ppo_mlp_oracle import ContinuousEnvOracle
    rs = np.random.RandomState(0)
    host = DummyVecEnv([SyntheticContinuous(5, 17, 6, horizon=13)])
    orc = ContinuousEnvOracle(5, 17, 6, 13)
    a = host.reset()[0]
    b = orc.reset()
    assert np.array_equal(a, b)
    dones = 0
    for t in range(200):
        act = (rs.randn(6) * 1.5).astype(np.float32)
        obs, rew, done, info = host.step([act])
        s, r, d = orc.step(act)
        assert np.array_equal(obs[0], s) and rew[0] == r and bool(done[0]) == d, t
        dones += d
    assert dones >= 5


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_cont_env_header_host_functions_equal_oracle(tmp_path):
    """csrc/cont_env.h's __host__ __device__ functions, compiled into a host program: counters, observations, rewards,
    terminals and resets equal the oracle's, so the device kernels and the python statements share one definition."""
    from oracle.ppo_mlp_oracle import ContinuousEnvOracle
    src = tmp_path / "probe.cpp"
    src.write_text(r'''
#include "deeprl_amd/csrc/cont_env.h"
#include <stdio.h>
int main() {
  const uint64_t seed = 11; const int S = 7, A = 3; const int64_t horizon = 9;
  double s[S]; int64_t c = 0;
  for (int j = 0; j < S; ++j) s[j] = cenv_reset_state(seed, c, j);
  for (int t = 0; t < 60; ++t) {
    float a[A];
    for (int d = 0; d < A; ++d) a[d] = (float)((t * 7 + d * 3) % 11) * 0.3f - 1.4f;
    c += 1;
    double m = 0.0;
    for (int d = 0; d < A; ++d) { float x = a[d]; x = x < -1.f ? -1.f : (x > 1.f ? 1.f : x); m += (double)x; }
    m = m / (double)A;
    const bool done = cenv_done(seed, c, horizon);
    for (int j = 0; j < S; ++j) s[j] = done ? cenv_reset_state(seed, c, j) : (s[j] + 0.01 * m) + (-0.02 + 0.04 * cenv_u(seed, 0, c, j));
    printf("%d %.17g %d", t, cenv_reward(seed, c), done ? 1 : 0);
    for (int j = 0; j < S; ++j) printf(" %.17g", s[j]);
    printf("\n");
  }
  return 0;
}
''')
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "probe")
    subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O1", "-ffp-contract=off", "-I", ROOT, str(src), "-o", exe],
                   check=True, capture_output=True)
    lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    orc = ContinuousEnvOracle(11, 7, 3, 9)
    orc.reset()
    for t, line in enumerate(lines):
        f = line.split()
        act = np.asarray([np.float32((t * 7 + d * 3) % 11) * np.float32(0.3) - np.float32(1.4) for d in range(3)], dtype=np.float32)
        s, r, d = orc.step(act)
        assert int(f[0]) == t and float(f[1]) == r and int(f[2]) == int(d), line
        assert np.array_equal(np.asarray([float(x) for x in f[3:]]), s), (t, line)


def test_device_env_shadow_equals_stepped_host_envs():
    """device_env.DeviceContinuousVec.shadow (rewards / terminals / episodic returns from the counters alone, vectorised) lists
    exactly the finished episodes the host environments report when stepped, with the same returns and positions."""
    from deeprl_amd.device_env import DeviceContinuousVec
    from deeprl_amd.envs import DummyVecEnv, SyntheticContinuous
    n, t_len, horizon = 5, 40, 11
    envs = [SyntheticContinuous(100 + i, 4, 2, horizon=horizon) for i in range(n)]
    vec = DummyVecEnv(envs)
    vec.reset()
    sh = object.__new__(DeviceContinuousVec)
    sh.num_envs, sh.horizon = n, horizon
    sh.seeds_host = np.asarray([e.seed for e in envs], dtype=np.int64)
    sh.counters_host = np.zeros(n, dtype=np.int64)
    sh.ret_host = np.zeros(n, dtype=np.float64)
    rs = np.random.RandomState(1)
    for rollout in range(3):
        want = []
        for t in range(t_len):
            _, _, _, infos = vec.step(rs.randn(n, 2).astype(np.float32))
            want += [(t, i, info['episodic_return']) for i, info in enumerate(infos) if info['episodic_return'] is not None]
        got = sh.shadow(t_len)
        assert got == sorted(want) and len(got) > 0
    assert np.array_equal(sh.counters_host, [e.c for e in envs])
    assert np.array_equal(sh.ret_host, [e.ret for e in envs])


def test_normaliser_restatements_agree():
    """normalizers.MeanStdNormalizer (product host class) == numerics_oracle.MeanStdNormalizerOracle bit for bit, and both equal
    a two-pass computation of the same statistics (the third-party RunningMeanStd is PARITY UNPINNED: restated from its published
    algorithm)."""
    from deeprl_amd.normalizers import MeanStdNormalizer
    from oracle.numerics_oracle import MeanStdNormalizerOracle
    rs = np.random.RandomState(3)
    a, b = MeanStdNormalizer(), MeanStdNormalizerOracle()
    seen = []
    for _ in range(50):
        x = rs.randn(16, 17) * rs.uniform(0.1, 5.0, size=17) + rs.uniform(-3, 3, size=17)
        seen.append(x)
        assert np.array_equal(a(x), b(x))
    allx = np.concatenate(seen)
    # count starts at 1e-4 with mean 0 / var 1: the two-pass moments agree to that prior's weight
    np.testing.assert_allclose(a.rms.mean.reshape(-1), allx.mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a.rms.var.reshape(-1), allx.var(0), rtol=1e-5)


def test_gauss_noise_oracle_is_standard_normal():
    from oracle.ppo_mlp_oracle import gauss_noise
    z = np.concatenate([gauss_noise(7, t, 16, np.arange(16), 6).reshape(-1) for t in range(400)])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and np.isfinite(z).all()
    assert np.array_equal(gauss_noise(7, 3, 16, [5], 6), gauss_noise(7, 3, 16, np.arange(16), 6)[5:6])   # position-only stream
