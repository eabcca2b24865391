#!/usr/bin/env python
"""cProfile of agent.step() on the generic (autograd) path: where does the host time go?"""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprl_amd as d
import deeprl_amd.agents as agents_mod
import bench_agents as B

agents_mod.get_logger = lambda *x, **k: B._Quiet()
d.select_device(0)
d.random_seed(0)
case = sys.argv[1]
n = int(sys.argv[2])
agent, meta = B.CASES[case]()
warm = 80 if ("dqn" in case or "c51" in case) else 2
for _ in range(warm):
    agent.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    agent.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:50]))
