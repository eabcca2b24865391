#!/usr/bin/env python
"""Where does an async agent step go?  Host-side timing of the bench loop in steady state (no profiler):
per-step python time, C-call time, and the rates of learner-only / actor-only loops."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
from deeprl_amd.learner import DQNLearnerBench, draw_uniform_indices


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=int, default=255)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--ring", type=int, default=200_000)
    a = ap.parse_args()
    d.select_device(0)
    out = {}
    b = DQNLearnerBench(ring_capacity=a.ring, batch=32, seed=0, actor=True, async_actor=True, variant=a.variant)
    L = b.learner
    for _ in range(200):
        b.step()
    torch.cuda.synchronize()
    import ctypes
    from deeprl_amd._lib import lib
    st0 = (ctypes.c_double * 3)()
    lib.dra_dqn_learner_host_stats(L.h, st0, 1)
    # (1) full async loop, host time split
    py = call = 0.0
    calls = []
    t_all = time.perf_counter()
    for _ in range(a.steps):
        t0 = time.perf_counter()
        b.pos, b.size = b._next
        idx = draw_uniform_indices(b.size, b.pos, b.batch, b.history, b.n_step)
        b._next = b._queue_env_steps(4)
        t1 = time.perf_counter()
        L.step(idx, True, True)
        t2 = time.perf_counter()
        py += t1 - t0; call += t2 - t1; calls.append(t2 - t1)
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    c = np.array(calls) * 1e6
    import ctypes
    from deeprl_amd._lib import lib
    st = (ctypes.c_double * 3)()
    lib.dra_dqn_learner_host_stats(L.h, st, 1)
    out["c_call"] = {"calls": st[0], "call_us": 1e6 * st[1] / max(1, st[0]), "blocked_us": 1e6 * st[2] / max(1, st[0])}
    out["async"] = {"us_per_step_total": 1e6 * t_tot / a.steps, "host_loop_us": 1e6 * t_host / a.steps,
                    "python_us": 1e6 * py / a.steps, "call_us_mean": float(c.mean()), "call_us_p10": float(np.percentile(c, 10)),
                    "call_us_p50": float(np.percentile(c, 50)), "call_us_p90": float(np.percentile(c, 90))}
    # (2) learner only (no env transitions): same call, n_env = 0
    L.params.n_env = 0
    for _ in range(100):
        L.step(draw_uniform_indices(b.size, b.pos, b.batch, b.history, b.n_step), True, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        L.step(draw_uniform_indices(b.size, b.pos, b.batch, b.history, b.n_step), True, True)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    out["learner_only"] = {"us_per_step": 1e6 * (time.perf_counter() - t0) / a.steps, "host_us": 1e6 * th / a.steps}
    # (3) actor only (no update)
    for _ in range(50):
        b._queue_env_steps(4); L.step(None, False, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        b._queue_env_steps(4); L.step(None, False, True)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    out["actor_only"] = {"us_per_step": 1e6 * (time.perf_counter() - t0) / a.steps, "host_us": 1e6 * th / a.steps}
    # (4) pure host: python part only
    t0 = time.perf_counter()
    for _ in range(a.steps):
        draw_uniform_indices(b.size, b.pos, b.batch, b.history, b.n_step); b._queue_env_steps(4)
    out["python_only_us"] = 1e6 * (time.perf_counter() - t0) / a.steps
    print(json.dumps(out))


if __name__ == "__main__":
    main()
