#!/usr/bin/env python
"""Where a workgroup of the scatter-form input gradient (csrc/dgrad_scatter.h) spends its time: in-kernel stamps of the trace build.

    make -C deeprl_amd/csrc trace
    DEEPRL_AMD_LIB=deeprl_amd/lib/libdeeprl_amd_trace.so python tools/phase_dgrad_scatter.py [batches ...]

Per layer and batch, means over workgroups (thread 0's wave): operands arrived (weights + first tile's gradient), image zeroed +
barrier, first tile (MFMAs + read-add-write), remaining tiles, barrier, epilogue (image -> dX), whole workgroup; kernel span."""
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

dll = ctypes.CDLL(LIBRARY)
nreg, nwg = ctypes.c_int(), ctypes.c_int()
dll.dra_trace_layout(ctypes.byref(nreg), ctypes.byref(nwg))
nreg, nwg = nreg.value, nwg.value
dll.dra_trace_set.argtypes = [ctypes.c_void_p]
d.select_device(0)
dev = d.Config.DEVICE
GEOM = {2: (32, 20, 64, 4, 2), 3: (64, 9, 64, 3, 1)}
REGION = {3: 7, 2: 8}       # TR_CONV3_B, TR_CONV2_B
var = ops.VAR_FUSED_BWD | ops.VAR_ONESHOT_DGRAD | ops.VAR_ONESHOT_WGRAD | ops.VAR_DGRAD_SCATTER | 2097152    # input-gradient role alone
for B in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    for layer, (c, h, oc, kh, s) in GEOM.items():
        oh = (h - kh) // s + 1
        x = torch.relu(torch.randn(B, c, h, h, device=dev))
        wt = torch.randn(c, kh, kh, oc, device=dev) * 0.05
        dy = torch.randn(B, oc, oh, oh, device=dev)
        call = lambda: ops.conv_bwd_fused_koc(layer, dy, x, wt, ksplit=16, variant=var, xact=x)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        buf = torch.zeros(nreg * nwg * 8, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        assert dll.dra_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
        call()
        torch.cuda.synchronize()
        assert dll.dra_trace_set(None) == 0
        raw = buf.cpu().numpy().view(np.uint64).reshape(nreg, nwg, 8)[REGION[layer]]
        used = raw[:, 0] > 0
        st = raw[used].astype(np.float64) / 100.0        # us
        t0 = st[:, 0].min()
        m = lambda a, b: round(float((st[:, a] - st[:, b]).mean()), 2)
        print(json.dumps({"layer": layer, "batch": B, "workgroups": int(used.sum()), "kernel_span_us": round(float(st[:, 7].max() - t0), 2),
                          "start_spread_us": round(float(st[:, 0].max() - t0), 2), "operands_arrived_us": m(1, 0), "zero_barrier_us": m(2, 1),
                          "first_tile_us": m(3, 2), "other_tiles_us": m(4, 3), "barrier_us": m(5, 4), "epilogue_us": m(7, 5),
                          "whole_workgroup_us": m(7, 0),
                          "whole_workgroup_us_min_max": [round(float((st[:, 7] - st[:, 0]).min()), 2), round(float((st[:, 7] - st[:, 0]).max()), 2)]}), flush=True)
