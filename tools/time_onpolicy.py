#!/usr/bin/env python
"""Host vs device time of an on-policy agent step (measurement aid): for a bench_agents case, the wall time of agent.step()
with and without a device synchronisation after every step, and a cProfile of the host side.

    python tools/time_onpolicy.py <case> [steps]
"""
import cProfile
import io
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import deeprl_amd as d  # noqa: E402
import deeprl_amd.agents as agents_mod  # noqa: E402
import bench_agents as B  # noqa: E402

agents_mod.get_logger = lambda *x, **k: B._Quiet()
d.select_device(0)
d.random_seed(0)
case = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
agent, meta = B.CASES[case]()
for _ in range(6):
    agent.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    agent.step()
t_host = time.perf_counter() - t0            # time to ENQUEUE n steps (the host may run ahead of the device)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# device time alone: events around the same loop
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    agent.step()
e1.record()
torch.cuda.synchronize()
print(json.dumps({"case": case, "steps": n, "wall_ms_per_step": 1e3 * t_all / n, "host_enqueue_ms_per_step": 1e3 * t_host / n,
                  "device_span_ms_per_step": e0.elapsed_time(e1) / n, "env_steps_per_s": n * meta["env_per_step"] / t_all}))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    agent.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(line[:160] for line in s.getvalue().splitlines()[:40]))
