"""The ctypes mirrors of the C-ABI structs must have the layout the C compiler gives include/deeprl_amd.h: size and the offset
of every field, checked by compiling a probe with gcc (no GPU, no HIP)."""
import ctypes
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_layout(tmp_path, struct, fields):
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "include/deeprl_amd.h"', 'int main(void) {',
           '  printf("%%zu\\n", sizeof(%s));' % struct]
    src += ['  printf("%%zu\\n", offsetof(%s, %s));' % (struct, f) for f in fields]
    src += ['  return 0;', '}']
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "probe")
    subprocess.run(["gcc", "-std=c99", str(c), "-I", ROOT, "-o", exe], check=True)
    return [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
@pytest.mark.parametrize("which", ["dra_per_chain2_io", "dra_dqn_step_params", "dra_fold_seg", "dra_ppo_mlp_net", "dra_ppo_mlp_cfg",
                                   "dra_ppo_mlp_rollout_io"])
def test_ctypes_mirror_matches_the_header(tmp_path, which):
    from deeprl_amd import ops
    from deeprl_amd.learner import StepParams
    from deeprl_amd import ppo_mlp
    mirror = {"dra_per_chain2_io": ops.PerChain2IO, "dra_dqn_step_params": StepParams, "dra_fold_seg": ops.FoldSeg,
              "dra_ppo_mlp_net": ppo_mlp.Net, "dra_ppo_mlp_cfg": ppo_mlp.Cfg, "dra_ppo_mlp_rollout_io": ppo_mlp.RolloutIO}[which]
    names = [f[0] for f in mirror._fields_]
    got = _c_layout(tmp_path, which, names)
    assert got[0] == ctypes.sizeof(mirror)
    assert got[1:] == [getattr(mirror, n).offset for n in names]


def test_step_params_numpy_view_has_the_ctypes_layout():
    from deeprl_amd.learner import _STEP_PARAMS_DTYPE, StepParams
    assert _STEP_PARAMS_DTYPE.itemsize == ctypes.sizeof(StepParams)
    head = [f[0] for f in StepParams._fields_ if f[0] not in ("reserved", "idx")]
    assert list(_STEP_PARAMS_DTYPE.names) == head
    for n in head:
        assert _STEP_PARAMS_DTYPE.fields[n][1] == getattr(StepParams, n).offset
        assert _STEP_PARAMS_DTYPE.fields[n][0].itemsize == getattr(StepParams, n).size


@pytest.mark.parametrize("n_actions", [1, 2, 4, 16, 6])
def test_actor_randomness_block_is_the_scalar_stream(n_actions):
    """DQNLearnerBench._push_blocks draws the actor's epsilon-greedy randomness for 64 env steps in one call: the values and the
    generator state afterwards must be the scalar calls' (randint(n, size=1), rand(1) per env step: torch_utils.py:51-58)."""
    import numpy as np
    from deeprl_amd.learner import actor_randomness_block
    a, b = np.random.RandomState(977), np.random.RandomState(977)
    got = actor_randomness_block(b, n_actions, 64)
    if n_actions & (n_actions - 1):
        assert got is None                    # no vectorised form: the caller keeps the scalar draws
        return
    ra = np.empty(64, dtype=np.int64)
    dice = np.empty(64, dtype=np.float64)
    for i in range(64):
        ra[i] = a.randint(n_actions, size=1)[0]
        dice[i] = a.rand(1)[0]
    assert np.array_equal(got[0], ra) and np.array_equal(got[1], dice)
    assert a.randint(1 << 30) == b.randint(1 << 30)


def test_push_blocks_array_form_equals_the_per_field_form(monkeypatch):
    """The array-filled upload of 16 agent steps is byte for byte the one the per-field loop builds (no GPU: the C entry point
    is replaced by a recorder)."""
    import numpy as np
    from deeprl_amd import learner as lm

    class _Rec:
        def __init__(self):
            self.calls = []

        def dra_dqn_learner_actor_ring_push(self, h, blocks, n, stream):
            raw = ctypes.string_at(ctypes.addressof(blocks.contents) if hasattr(blocks, "contents") else ctypes.addressof(blocks),
                                   n * ctypes.sizeof(lm.StepParams))
            head = lm.StepParams.idx.offset
            self.calls.append(b"".join(raw[i * ctypes.sizeof(lm.StepParams):i * ctypes.sizeof(lm.StepParams) + head] for i in range(n)))
            return 0

    class _L:
        h = None
        actor_stream = None
        set_env_steps = lm.DQNLearner.set_env_steps

        def __init__(self):
            self.params = lm.StepParams()

        def _sp(self, s=None):
            return None

    def make(vectorised):
        b = lm.DQNLearnerBench.__new__(lm.DQNLearnerBench)
        b.learner, b.n_actions, b.capacity, b.epsilon = _L(), 4, 1000, 0.01
        b.pos, b.size, b.counter = 990, 995, 123456
        b.actor_rs = np.random.RandomState(3)
        b._ring_pushed = 0
        if not vectorised:
            monkeypatch.setattr(lm, "actor_randomness_block", lambda *a: None)
        return b

    rec = _Rec()
    monkeypatch.setattr(lm, "lib", rec)
    fast = make(True)
    fast._push_blocks(16)
    fast._push_blocks(16)
    slow = make(False)
    slow._push_blocks(16)
    slow._push_blocks(16)
    assert rec.calls[0] == rec.calls[2] and rec.calls[1] == rec.calls[3]
    assert (fast.pos, fast.size, fast.counter, fast._ring_pushed) == (slow.pos, slow.size, slow.counter, slow._ring_pushed)
    assert bytes(fast.learner.params)[:lm.StepParams.idx.offset] == bytes(slow.learner.params)[:lm.StepParams.idx.offset]
    assert fast.actor_rs.randint(1 << 30) == slow.actor_rs.randint(1 << 30)


def test_device_actor_pipeline_push_array_form_equals_the_per_field_form(monkeypatch):
    """DeviceActorPipeline._push (the agents' async device pipeline): the array-filled upload of 16 agent steps -- episode
    shadow, epsilon schedule, the actor's own random stream -- is byte for byte the per-field one, and reports the same
    (reward, done, info) per transition."""
    import numpy as np
    from deeprl_amd import learner as lm
    from deeprl_amd.support import LinearSchedule
    head = lm.StepParams.idx.offset
    size = ctypes.sizeof(lm.StepParams)

    class _Rec:
        def __init__(self):
            self.calls = []

        def dra_dqn_learner_actor_ring_push(self, h, blocks, n, stream):
            addr = ctypes.addressof(blocks.contents) if hasattr(blocks, "contents") else ctypes.addressof(blocks)
            raw = ctypes.string_at(addr, n * size)
            self.calls.append(b"".join(raw[i * size:i * size + head] for i in range(n)))
            return 0

    class _L:
        h = None
        actor_stream = None
        set_env_steps = lm.DQNLearner.set_env_steps

        def __init__(self):
            self.params = lm.StepParams()

        def _sp(self, s=None):
            return None

    def make():
        p = lm.DeviceActorPipeline.__new__(lm.DeviceActorPipeline)
        p.L, p.A, p.n_env, p.capacity, p.slot = _L(), 4, 4, 5000, 4990
        p.stream = lm.SyntheticEpisodeStream(seed=7, counter0=100, done_period=37)
        p.epsilon_fn = LinearSchedule(1.0, 0.01, 500)
        p.rs = np.random.RandomState(1977)
        p.async_actor, p.pending, p.pushed = True, [], 0
        return p

    rec = _Rec()
    monkeypatch.setattr(lm, "lib", rec)
    fast = make()
    fast._push()
    fast._push(16)
    monkeypatch.setattr(lm, "actor_randomness_block", lambda *a: None)
    slow = make()
    slow._push()
    slow._push(16)
    assert rec.calls[0] == rec.calls[2] and rec.calls[1] == rec.calls[3]
    assert fast.pending == slow.pending and len(fast.pending) == 32
    assert any(done for step in fast.pending for (_, done, _) in step)          # (episode boundaries inside the blocks)
    assert (fast.slot, fast.pushed) == (slow.slot, slow.pushed)
    assert bytes(fast.L.params)[:head] == bytes(slow.L.params)[:head]
    assert fast.rs.randint(1 << 30) == slow.rs.randint(1 << 30)
    assert fast.epsilon_fn() == slow.epsilon_fn() and fast.stream.state_dict() == slow.stream.state_dict()
