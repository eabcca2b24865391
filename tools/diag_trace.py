#!/usr/bin/env python
"""Profiler-free timeline of the pipelined async agent step (dra_dqn_learner_trace): microseconds of
gather start/end, actor-graph end (actor stream) and update-graph start/end (update stream)."""
import argparse, ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
from deeprl_amd._lib import lib
from deeprl_amd.learner import DQNLearnerBench

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=255)
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--brief", action="store_true")
a = ap.parse_args()
d.select_device(0)
b = DQNLearnerBench(ring_capacity=200_000, batch=32, seed=0, actor=True, async_actor=True, variant=a.variant)
for _ in range(300):
    b.step()
lib.dra_dqn_learner_trace(b.learner.h, a.steps)
for _ in range(a.steps + 50):
    b.step()
out = (ctypes.c_float * (a.steps * 5))()
n = ctypes.c_int()
lib.dra_dqn_learner_trace_read(b.learner.h, out, a.steps, ctypes.byref(n))
t = 1e3 * np.array(out[: n.value * 5]).reshape(n.value, 5)
if a.brief:
    per = np.diff(t[:, 0])
    print(json.dumps({"variant": a.variant, "actor_cus": os.environ.get("DRA_ACTOR_CUS"), "layout": os.environ.get("DRA_CU_LAYOUT"),
                      "period_us": round(float(per.mean()), 1), "gather_us": round(float((t[:, 1] - t[:, 0]).mean()), 1),
                      "actor_chain_us": round(float((t[:, 2] - t[:, 1]).mean()), 1),
                      "update_chain_us": round(float((t[:, 4] - t[:, 3]).mean()), 1)}))
    sys.exit(0)
print("step   gather_start gather_end actor_end | update_start update_end   (us)   [actor chain, update chain, period]")
for i in range(n.value):
    per = t[i, 0] - t[i - 1, 0] if i else 0.0
    print("%3d   %9.1f %9.1f %9.1f | %9.1f %9.1f      [%6.1f %6.1f %6.1f]" % (i, *t[i], t[i, 2] - t[i, 1], t[i, 4] - t[i, 3], per))
