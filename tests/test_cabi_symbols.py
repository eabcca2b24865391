"""CPU checks of the drop-in boundary: the shared library builds, loads without a GPU and exports
every symbol include/deeprl_amd.h declares; product code never imports the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from deeprl_amd import _lib
    if not os.path.isfile(_lib.LIBRARY):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "deeprl_amd", "csrc"), "-j8"])
    protos = _lib.parse_header()
    assert len(protos) >= 30
    dll = ctypes.CDLL(_lib.LIBRARY)
    for name in protos:
        assert hasattr(dll, name), "missing export %s" % name
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIBRARY]).decode()
    exported = set(re.findall(r" T (dra_\w+)", exported))
    assert exported == set(protos), "header and library disagree: %s" % sorted(exported ^ set(protos))


def test_calls_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    from deeprl_amd._lib import DraError, lib
    h = ctypes.c_void_p()
    with pytest.raises(DraError):
        lib.dra_sumtree_create(ctypes.byref(h), 16)  # hipErrorNoDevice -> exception, never a CPU fallback


def test_product_never_imports_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "deeprl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "ref_shim" in src:
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_adam_hyper_matches_torch_bias_corrections():
    """dra_adam_hyper (host arithmetic shared by the eager and the graph-replayable Adam step) reproduces
    torch.optim.Adam's step_size = lr / (1 - b1^t) and 1 / sqrt(1 - b2^t) (examples.py:139,204 optimisers)."""
    import math
    from deeprl_amd._lib import lib
    for lr, b1, b2 in ((0.00025, 0.9, 0.999), (5e-5, 0.9, 0.999), (3e-4, 0.5, 0.9)):
        for step in (1, 2, 10, 1000, 123456):
            out = (ctypes.c_float * 2)()
            lib.dra_adam_hyper(lr, b1, b2, step, out)
            f32 = lambda v: ctypes.c_float(v).value
            bc1 = 1.0 - float(f32(b1)) ** step
            bc2 = 1.0 - float(f32(b2)) ** step
            assert out[0] == f32(float(f32(lr)) / bc1)
            assert out[1] == f32(1.0 / math.sqrt(bc2))


def test_vectorised_index_draw_consumes_the_reference_stream():
    """learner.draw_uniform_indices == the scalar rejection loop of replay.py:92-110 (same indices, same
    np.random state afterwards), including a write head in the middle of the ring and a ring that is not full."""
    import numpy as np
    from deeprl_amd.learner import draw_uniform_indices

    def scalar(size, pos, batch, h, n):
        out = []
        while len(out) < batch:
            i = int(np.random.randint(0, size))
            if (i - h + 1 >= 0 and i + n < pos) or (i - h + 1 >= pos and i + n < size):
                out.append(i)
        return np.asarray(out, dtype=np.int64)

    for size, pos, h, n in ((1000, 0, 4, 1), (1000, 517, 4, 3), (50, 50, 4, 1), (12, 7, 2, 2), (1_000_000, 123_456, 4, 1)):
        np.random.seed(size + pos)
        want = scalar(size, pos, 32, h, n)
        tail_want = np.random.randint(0, 1 << 30, size=3)
        np.random.seed(size + pos)
        got = draw_uniform_indices(size, pos, 32, h, n)
        tail_got = np.random.randint(0, 1 << 30, size=3)
        assert np.array_equal(got, want) and np.array_equal(tail_got, tail_want)


def test_clip_step_plan_host_arithmetic():
    """dra_grad_sqnorm_segs_blocks (pure host code: the work decomposition of the fold + norm launch): one fold workgroup per 64 float4 for segments of <= 32 slabs, per 16 float4 above, one plain
    workgroup per 1024 float4 -- 796 for the DQN learner's gradient (conv segments of 160 / 32 / 32 slabs, fc4 + head plain);
    malformed layouts are refused."""
    from deeprl_amd import ops
    from deeprl_amd._lib import lib

    class _T:                                    # a fake 16-byte aligned "tensor": the planner only validates pointers
        def __init__(self, addr):
            self._a = addr

        def data_ptr(self):
            return self._a

    def blocks(n, segs):
        b = ctypes.c_int(0)
        rc = lib.dra_grad_sqnorm_segs_blocks.raw(int(n), ops._fold_seg_array(segs), len(segs), ctypes.byref(b))
        return rc, b.value

    counts, nsl = [8224, 32832, 36928], [160, 32, 32]
    tail = 3136 * 512 + 512 + 4 * 512 + 4
    segs, off = [], 0
    for cnt, ns in zip(counts, nsl):
        segs.append((off, cnt, _T(0x10000), cnt, ns))
        off += cnt
    rc, b = blocks(off + tail, segs)
    want = -(-(8224 // 4) // 16) + -(-(32832 // 4) // 64) + -(-(36928 // 4) // 64) + -(-(tail // 4) // 1024)
    assert rc == 0 and b == want == 796
    assert blocks(off + tail, [])[0] == 0                                           # no slab segments: plain only
    assert blocks(off + tail + 1, segs)[0] == -22                                   # n not a multiple of 4
    assert blocks(off + tail, [(4, 8220, _T(0x10000), 8224, 160)])[0] == -22        # segments must start at 0, contiguously
    assert blocks(off + tail, [(0, 8224, _T(0x10004), 8224, 160)])[0] == -22        # misaligned slab pointer
    rc, b = blocks(64 * 1024 * 1024, [])                                            # more float4s than 4096 workgroups hold at once:
    assert rc == 0 and b <= 4096                                                    # plain workgroups walk several strides
