"""CPU transcription of fold_norm_kernel's work decomposition (csrc/optim.hip) (tools/emulate_oneshot.py emu_clip_step: block ranges, the
narrow / wide fold workgroups, the owner threads, the plain workgroups) against a plain numpy fold: every float4 of the
gradient is owned by exactly ONE (workgroup, thread), slab
segments are folded in the documented order (slabs g, g+16, ... per group, then the 16 groups in order), elements outside
the segments are untouched, and the partials add up to the squared norm.  Shapes the GPU tests do not sweep: segment sizes
that are not multiples of the workgroup's element count, fewer than 16 slabs, 33-160 slabs, more than 160 (two passes)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emu():
    spec = importlib.util.spec_from_file_location("emulate_oneshot", os.path.join(ROOT, "tools", "emulate_oneshot.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _ordered_fold(slabs, ns):
    groups = []
    for g in range(16):
        acc = np.zeros(slabs.shape[1], dtype=np.float32)
        for s_ in range(g, ns, 16):
            acc = acc + slabs[s_]
        groups.append(acc)
    r = groups[0].copy()
    for g in range(1, 16):
        r = r + groups[g]
    return r


@pytest.mark.parametrize("counts,nsl,tail", [([8224, 32832, 36928], [160, 32, 32], 4100), ([260, 68, 1028], [5, 33, 200], 8),
                                             ([64, 4], [16, 1], 0), ([], [], 5000), ([1300], [161], 1024 * 4 + 12)])
def test_clip_step_decomposition(counts, nsl, tail):
    emu = _emu()
    rs = np.random.RandomState(len(counts) * 7 + tail)
    n = sum(counts) + tail
    grad = rs.standard_normal(n).astype(np.float32)
    segs, want, off = [], grad.copy(), 0
    for cnt, ns in zip(counts, nsl):
        sl = rs.standard_normal((ns, cnt)).astype(np.float32)
        segs.append((off, cnt, sl, cnt, ns))
        want[off:off + cnt] = _ordered_fold(sl, ns)
        off += cnt
    got, partials, owners = emu.emu_clip_step(grad, segs)
    assert np.array_equal(got, want)
    assert np.array_equal(owners, np.ones(n // 4, dtype=np.int64)), "every float4 is held by exactly one thread"
    np.testing.assert_allclose(partials.sum(), (want.astype(np.float64) ** 2).sum(), rtol=1e-6)
