#!/usr/bin/env python
"""Per-phase timeline of every kernel of one DQN agent step, from in-kernel s_memrealtime stamps.

    make -C deeprl_amd/csrc trace
    DEEPRL_AMD_LIB=deeprl_amd/lib/libdeeprl_amd_trace.so python tools/phase_trace.py [--sync] [--variant V] > out.json

The trace build writes, per workgroup, time stamps at phase boundaries (entry / operands consumed / LDS staged /
MFMAs issued / partials exchanged / results stored / all stores complete) plus the XCD it ran on; the stamps are
passive, so the step traced is the REAL pipelined step (actor chain and update chain overlapped on their CU
partitions, graphs replayed).  The last traced step survives in the buffer.  Output: for every kernel family its
span, number of workgroups, workgroups per XCD, distribution of workgroup start times (rounds), and the mean /
p50 / p90 duration of every phase in microseconds (s_memrealtime ticks at 100 MHz)."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d  # noqa: E402
from deeprl_amd._lib import LIBRARY  # noqa: E402
from deeprl_amd.learner import DQNLearnerBench  # noqa: E402

REGIONS = ["gather", "conv1_fwd", "conv2_fwd", "conv3_fwd", "fc4_fwd", "head_loss", "fc_bwd", "conv3_bwd", "conv2_bwd",
           "conv1_bwd_w", "grad_norm", "rmsprop_step", "actor_conv1", "actor_conv2", "actor_conv3", "actor_fc4", "actor_head_env"]
PHASES = ["issue+wait operands", "barrier (staged)", "mfma issue", "barrier (mfma done)", "fold/epilogue", "store drain"]
TICK_US = 0.01


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sync", action="store_true", help="in-order mode on the whole chip (default: async pipelined)")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=8000)     # (about a second: a cold device runs the first tens of milliseconds at a lower clock)
    ap.add_argument("--ring", type=int, default=200_000)
    args = ap.parse_args()
    if "trace" not in os.path.basename(LIBRARY):
        raise SystemExit("run with DEEPRL_AMD_LIB=<...>/libdeeprl_amd_trace.so (make -C deeprl_amd/csrc trace)")
    dll = ctypes.CDLL(LIBRARY)
    nreg, nwg = ctypes.c_int(), ctypes.c_int()
    dll.dra_trace_layout(ctypes.byref(nreg), ctypes.byref(nwg))
    nreg, nwg = nreg.value, nwg.value
    d.select_device(0)
    bench = DQNLearnerBench(ring_capacity=args.ring, batch=32, seed=0, actor=True, async_actor=not args.sync, variant=args.variant)
    for _ in range(args.steps):
        bench.step()
    torch.cuda.synchronize()
    buf = torch.zeros(nreg * nwg * 8, dtype=torch.int64, device=d.Config.DEVICE)
    torch.cuda.synchronize()
    dll.dra_trace_set.argtypes = [ctypes.c_void_p]
    assert dll.dra_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
    for _ in range(6):
        bench.step()
    torch.cuda.synchronize()
    assert dll.dra_trace_set(None) == 0
    raw = buf.cpu().numpy().view(np.uint64).reshape(nreg, nwg, 8)
    out = {"mode": "sync" if args.sync else "async", "variant": bench.learner.variant, "tick_us": TICK_US,
           "update_cus": bench.learner.update_cus, "actor_cus": bench.learner.actor_cus, "kernels": {}}
    t_ref = None
    spans = {}
    for r in range(nreg):
        rec = raw[r]
        live = rec[:, 0] != 0
        if not live.any():
            continue
        rec = rec[live].astype(np.int64)
        spans[r] = (rec[:, 0].min(), rec[:, 7].max())
    # the update chain's kernels of the LAST step: start the clock at the earliest update kernel
    upd = [spans[r][0] for r in spans if 1 <= r <= 11]
    t_ref = min(upd) if upd else min(v[0] for v in spans.values())
    for r in range(nreg):
        if r not in spans:
            continue
        rec = raw[r]
        rec = rec[rec[:, 0] != 0].astype(np.int64)
        n = len(rec)
        start = (rec[:, 0] - rec[:, 0].min()) * TICK_US
        dur = (rec[:, 7] - rec[:, 0]) * TICK_US
        xcc = ((rec[:, 6] >> 32) & 0xffff)
        se_cu = rec[:, 6] & 0xffffffff
        k = {"workgroups": n, "t_start_us": round((spans[r][0] - t_ref) * TICK_US, 2),
             "span_us": round((spans[r][1] - spans[r][0]) * TICK_US, 2),
             "wg_dur_us": {"mean": round(float(dur.mean()), 2), "p50": round(float(np.median(dur)), 2),
                           "p90": round(float(np.percentile(dur, 90)), 2), "max": round(float(dur.max()), 2)},
             "wg_start_us_hist(0,1,2,4,6,8,12,16+)": np.histogram(start, bins=[0, 1, 2, 4, 6, 8, 12, 16, 1e9])[0].tolist(),
             "wgs_per_xcd": np.bincount(xcc, minlength=8).tolist(),
             "distinct_cus": int(len(set(zip(xcc.tolist(), ((se_cu >> 8) & 0xf).tolist(), ((se_cu >> 13) & 0x7).tolist(),
                                            ((se_cu >> 12) & 0x1).tolist()))))}
        ph = {}
        prev = rec[:, 0]
        for s in (1, 2, 3, 4, 5, 7):
            cur = rec[:, s]
            ok = cur != 0
            if not ok.any():
                continue
            dd = (cur[ok] - prev[ok]) * TICK_US
            name = "->s%d %s" % (s, PHASES[{1: 0, 2: 1, 3: 2, 4: 3, 5: 4, 7: 5}[s]])
            ph[name] = {"mean": round(float(dd.mean()), 2), "p50": round(float(np.median(dd)), 2),
                        "p90": round(float(np.percentile(dd, 90)), 2)}
            prev = np.where(ok, cur, prev)
        k["phases_us"] = ph
        out["kernels"][REGIONS[r]] = k
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
