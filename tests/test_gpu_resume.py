"""True resume (the reference's BaseAgent.save keeps weights + normaliser only, BaseAgent.py:24-33,
misc.py:24-25): 40 agent steps + save_full + a FRESH process + load_full + 20 agent steps must equal 60 uninterrupted agent
steps bit for bit -- parameters, target network, replay ring contents, stored actions -- for the async two-stream pipeline
(the benchmarked configuration), the in-order pipeline, and C51 + PrioritizedReplay."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "resume_worker.py")


def _run(args):
    r = subprocess.run([sys.executable, WORKER] + [str(a) for a in args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("kind", ["dqn_async", "dqn_sync", "c51_per_async"])
def test_save_full_load_full_continues_bit_for_bit(tmp_path, kind):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    whole, part1, part2 = str(tmp_path / "whole.npz"), str(tmp_path / "p1.npz"), str(tmp_path / "p2.npz")
    ckpt = str(tmp_path / "ckpt")
    _run([kind, whole, "run", 60])
    _run([kind, part1, "save", 40, ckpt])
    for ext in (".model", ".stats", ".resume", ".replay"):
        assert os.path.isfile(ckpt + ext), ext                       # the reference-format files are still written
    _run([kind, part2, "load", 20, ckpt])
    a, b = dict(np.load(whole)), dict(np.load(part2))
    assert int(a["total_steps"]) == int(b["total_steps"]) == 240
    moved = dict(np.load(part1))
    assert any(not np.array_equal(moved[k], a[k]) for k in a if k.startswith("p_")), "the last 20 steps must change something"
    for k in a:
        assert np.array_equal(a[k], b[k]), k
