#!/usr/bin/env python
"""Where dra_sumtree_per_chain2's time goes: needs the measurement build (DEEPRL_AMD_LIB=.../libdeeprl_amd_trace.so), whose
kernel leaves s_memrealtime stamps (100 MHz) in the unused tail of out_raw_idx."""
import os
import sys

os.environ.setdefault("DRA_PER_RIDE", "0")     # the draw as its own launch: one set of stamps per launch (riding halves overwrite slot 1)

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
import deeprl_amd.agents as agents_mod
import bench_agents as B

agents_mod.get_logger = lambda *x, **k: B._Quiet()
d.select_device(0)
d.random_seed(0)
agent, meta = B.CASES[sys.argv[1] if len(sys.argv) > 1 else "dqn_pixel_per_device"]()
for _ in range(600):
    agent.step()
dd = agent._pipe._dd
names = ["PCIe inputs, priorities, max / min", "stat + first occurrence", "leaf dedupe, old leaves, atomic deltas issued",
         "atomics acknowledged (barrier)", "top of the tree -> LDS", "descent + valid_index", "filter / padding (lane 0)",
         "hand-over stores + ack"]
acc = np.zeros(8)
n = 0
for _ in range(400):
    agent.step()
    agent.sync_host()
    for io, t, v in dd.blocks:
        st = v["raw"][1024 - 16:1024 - 16 + 9].astype(np.float64)
        if st[0] > 0 and (np.diff(st) >= 0).all():
            acc += np.diff(st) / 100.0          # us
            n += 1
print("dra_sumtree_per_chain2 phases, mean of %d launches (us):" % n)
for k, name in enumerate(names):
    print("  %-46s %6.2f" % (name, acc[k] / n))
print("  %-46s %6.2f" % ("total (each stamp adds a PCIe store: ~+3 us)", acc.sum() / n))
