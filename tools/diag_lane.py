#!/usr/bin/env python
"""Host side of the event-free lane (DRA_VAR_FLAG_SYNC): where the microseconds of one bench step go -- python (index draw, block
push, ctypes) against the C call's parts (pacing wait, index staging, the update's launches, the actor launch)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
from deeprl_amd.learner import DQNLearnerBench, draw_uniform_indices


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    d.select_device(0)
    np.random.seed(0); torch.manual_seed(0)
    b = DQNLearnerBench(ring_capacity=200_000, batch=32, seed=0, actor=True, async_actor=True)
    L = b.learner
    for _ in range(400):
        b.step()
    torch.cuda.synchronize()
    s0 = L.lane_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    s1 = L.lane_stats()
    n = s1["steps"] - s0["steps"]
    parts = {k: (s1["host_us_per_step"][k] * s1["steps"] - s0["host_us_per_step"][k] * s0["steps"]) / max(1, n) for k in s1["host_us_per_step"]}
    # the python side alone: the same draws without the C call
    t1 = time.perf_counter()
    for _ in range(steps):
        draw_uniform_indices(b.size, b.pos, b.batch, b.history, b.n_step)
    t_draw = time.perf_counter() - t1
    print(json.dumps({"variant": L.variant, "steps": steps, "lane_steps": n, "updates_per_s": steps / t_all,
                      "us_per_step": 1e6 * t_all / steps, "python_loop_us_per_step": 1e6 * t_host / steps,
                      "c_call_parts_us": parts, "draw_uniform_indices_us": 1e6 * t_draw / steps}))


if __name__ == "__main__":
    main()
