#!/usr/bin/env python
"""cProfile of agent.step() for a bench_agents case (default: the host-emulator DQN with the async actor)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
import deeprl_amd.agents as agents_mod
import bench_agents as B

agents_mod.get_logger = lambda *x, **k: B._Quiet()
d.select_device(0)
d.random_seed(0)
case = sys.argv[1] if len(sys.argv) > 1 else "dqn_pixel_uniform_host_async"
agent, meta = B.CASES[case]()
for _ in range(300):
    agent.step()
torch.cuda.synchronize()
n = 1500
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    agent.step()
pr.disable()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s: %.1f us per agent.step() under cProfile (%d steps)" % (case, 1e6 * dt / n, n))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
