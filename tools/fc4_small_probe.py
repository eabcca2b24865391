#!/usr/bin/env python
"""fc4 (3136 -> 512) forward at rollout batch sizes: the eight-wave GEMV (dra_linear_fwd) against the one-pass K-slice kernel
(dra_linear_fwd_slabs_one, 8 / 14 / 28 slices; its fold would ride in the next launch) under HIP events, in a captured graph of
32 back-to-back launches (the way a rollout runs them)."""
import ctypes
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprl_amd as d
from deeprl_amd import ops
from deeprl_amd._lib import lib, stream_ptr

d.select_device(0)
dev = torch.device("cuda:0")
w = torch.randn(512, 3136, device=dev) * 0.02
bias = torch.randn(512, device=dev)


def timed(fn, reps=32, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / (iters * reps))
    return best


for b in (8, 16, 32, 80, 256):
    x = torch.randn(b, 3136, device=dev)
    out = {"batch": b}
    y = torch.empty(b, 512, device=dev)
    out["linear_fwd_us"] = timed(lambda: ops.linear_fwd([x], [w], [bias], act="relu"))      # <= 32: eight-wave GEMV; above: 14 slices + finish
    for ks in (8, 14, 28):
        slabs = torch.empty(1, ks, b, 512, device=dev)
        xa, wa = ops.ptr_array([x]), ops.ptr_array([w])
        out["slabs%d_us" % ks] = timed(lambda: lib.dra_linear_fwd_slabs_one(1, xa, wa, b, 3136, 512, ks, ctypes.c_void_p(slabs.data_ptr()), stream_ptr()))
    print(json.dumps(out))
