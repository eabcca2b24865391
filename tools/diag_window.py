#!/usr/bin/env python
"""The driver's 20-step window: per-step host time of the first steps after a device synchronise (the pipeline starts empty), against
the steady state -- python loop and the C call's parts."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
from deeprl_amd.learner import DQNLearnerBench


def main():
    d.select_device(0)
    np.random.seed(0); torch.manual_seed(0)
    b = DQNLearnerBench(ring_capacity=1_000_000, batch=32, seed=0, actor=True, async_actor=True)
    L = b.learner
    for _ in range(305):
        b.step()
    out = []
    for rep in range(3):
        torch.cuda.synchronize()
        if rep == 2:
            t_spin = time.perf_counter()
            while time.perf_counter() - t_spin < 0.002:   # keep the core busy for 2 ms before the window (clock / C-state check)
                pass
        per, parts = [], []
        t0 = time.perf_counter()
        for _ in range(20):
            ta = time.perf_counter()
            s0 = L.lane_stats()["host_us_per_step"]; n0 = L.lane_stats()["steps"]
            tb = time.perf_counter()
            b.step()
            tc = time.perf_counter()
            s1 = L.lane_stats()["host_us_per_step"]; n1 = L.lane_stats()["steps"]
            per.append(round(1e6 * (tc - tb), 1))
            parts.append({k: round(s1[k] * n1 - s0[k] * n0, 1) for k in s1})
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append({"rep": rep, "window_us_per_step_incl_probe": 1e6 * dt / 20, "step_us": per, "first": parts[0], "fifth": parts[4], "last": parts[-1]})
    # plain window, no probes
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            b.step()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append({"plain_window": rep, "us_per_step": 1e6 * dt / 20, "host_loop_us_per_step": 1e6 * th / 20})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
