#!/usr/bin/env python
"""Where the large-batch NatureConv forward (conv_fwd_v2_persist_kernel) spends its time: in-kernel stamps of the trace build.

    make -C deeprl_amd/csrc trace
    DEEPRL_AMD_LIB=deeprl_amd/lib/libdeeprl_amd_trace.so python tools/phase_conv_big.py [batch]

Per layer: workgroups, groups per workgroup, and the mean over workgroups of: prologue (weights + first group staged), MFMA phase
of the first group (thread 0's wave: issue, then the barrier that also waits for the next group's rows), partial sums -> LDS +
staging of the next group, and the whole workgroup."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d  # noqa: E402
from deeprl_amd import ops  # noqa: E402
from deeprl_amd._lib import LIBRARY  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dll = ctypes.CDLL(LIBRARY)
nreg, nwg = ctypes.c_int(), ctypes.c_int()
dll.dra_trace_layout(ctypes.byref(nreg), ctypes.byref(nwg))
nreg, nwg = nreg.value, nwg.value
d.select_device(0)
dev = d.Config.DEVICE
GEOM = {1: (4, 84, 32, 8, 4), 2: (32, 20, 64, 4, 2), 3: (64, 9, 64, 3, 1)}
out = {}
for layer, (c, h, oc, kh, s) in GEOM.items():
    oh = (h - kh) // s + 1
    x = (torch.randint(0, 256, (B, c, h, h), dtype=torch.uint8, device=dev) if layer == 1
         else torch.randn(B, c, h, h, device=dev))
    wt = torch.randn(c * kh * kh, oc, device=dev) * 0.05
    bb = torch.randn(oc, device=dev) * 0.05
    call = lambda: ops.conv_fwd_koc(layer, [x], [wt], [bb], u8_coef=1.0 / 255 if layer == 1 else None)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        call()
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / 20
    buf = torch.zeros(nreg * nwg * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    dll.dra_trace_set.argtypes = [ctypes.c_void_p]
    assert dll.dra_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
    call()
    torch.cuda.synchronize()
    assert dll.dra_trace_set(None) == 0
    raw = buf.cpu().numpy().view(np.uint64).reshape(nreg, nwg, 8)[1 + (layer - 1)]       # TR_CONV1_F = 1
    used = raw[:, 0] > 0
    st = raw[used].astype(np.float64) / 100.0        # us
    t0 = st[:, 0].min()
    flops = 2.0 * B * oh * oh * oc * c * kh * kh
    if layer == 1 and B >= 384:
        # conv1's own throughput kernel (all of K in registers): entry, prologue (weights through LDS), end
        out["conv1"] = {"batch": B, "us_per_call": us, "TFLOPs": flops / us / 1e6, "frac_mfma_f32": flops / us / 1e6 / 157.3,
                        "kernel": "conv1_fwd_u8_tp_kernel", "workgroups": int(used.sum()),
                        "kernel_span_us": float(st[:, 5].max() - t0), "prologue_us": float((st[:, 2] - st[:, 0]).mean()),
                        "whole_workgroup_us": float((st[:, 5] - st[:, 0]).mean()), "start_spread_us": float(st[:, 0].max() - t0)}
        continue
    rec = {"batch": B, "us_per_call": us, "TFLOPs": flops / us / 1e6, "frac_mfma_f32": flops / us / 1e6 / 157.3,
           "workgroups": int(used.sum()), "kernel_span_us": float(st[:, 5].max() - t0),
           "prologue_us": float((st[:, 2] - st[:, 0]).mean()),
           "first_group_mfma_issue_us": float((st[:, 1] - st[:, 2]).mean()),
           "barrier_after_mfma_us": float((st[:, 3] - st[:, 1]).mean()),
           "partials_and_staging_us": float((st[:, 4] - st[:, 3]).mean()),
           "whole_workgroup_us": float((st[:, 5] - st[:, 0]).mean()),
           "start_spread_us": float(st[:, 0].max() - t0)}
    out["conv%d" % layer] = rec
print(json.dumps(out, indent=1))
