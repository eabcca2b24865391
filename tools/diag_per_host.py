#!/usr/bin/env python
"""Host time of every call of one PrioritizedReplay agent step in the async device pipeline (perf_counter around the
calls DeviceActorPipeline.step makes), averaged over N steps."""
import os
import sys
import time

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
for _ in range(400):
    agent.step()
torch.cuda.synchronize()
pipe, L, rp = agent._pipe, agent._learner, agent._inner_replay()
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


rp.advance = timed("rp.advance", rp.advance)
rp.draw_begin = timed("rp.draw_begin", rp.draw_begin)
rp.draw_end = timed("rp.draw_end (incl. the wait)", rp.draw_end)
rp.commit_device = timed("rp.commit_device", rp.commit_device)
rp.tree.commit_f32 = timed("  tree.commit_f32 (C call)", rp.tree.commit_f32)
rp.tree.sample_into = timed("  tree.sample_into (C call)", rp.tree.sample_into)
L.upload_sampling_prob = timed("L.upload_sampling_prob", L.upload_sampling_prob)
L.step = timed("L.step (C call)", L.step)
L.wait_loss = timed("L.wait_loss", L.wait_loss)
L.sync_loss = timed("L.sync_loss (host wait)", L.sync_loss)
rp.chain_collect = timed("rp.chain_collect", rp.chain_collect)
rp.commit_select = timed("rp.commit_select", rp.commit_select)
rp.chain_fill = timed("rp.chain_fill", rp.chain_fill)
L.next_slot = timed("L.next_slot", L.next_slot)
L.set_per = timed("L.set_per", L.set_per)
pipe._push = timed("pipe._push (amortised)", pipe._push)
L.step_update = timed("L.step_update (C call)", L.step_update)
L.step_actor = timed("L.step_actor (C call)", L.step_actor)
L.per_chain2_wait = timed("  L.per_chain2_wait (spin)", L.per_chain2_wait)
if getattr(pipe, "_dd", None) is not None:
    dd = pipe._dd
    dd.fill = timed("dd.fill", dd.fill)
    dd.issued_idx = timed("dd.issued_idx (incl. the wait)", dd.issued_idx)
    dd._ensure_words = timed("  dd._ensure_words", dd._ensure_words)
pipe._block = timed("  pipe._block", pipe._block)
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    agent.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%.1f us per agent step (%.0f updates/s)" % (1e6 * dt / n, n / dt))
for k, v in acc.items():
    print("%-32s %7.1f us" % (k, 1e6 * v / n))

# ---- issue -> loss latency with an idle GPU: how long do [forward + loss + chain kernel] take by themselves?
if getattr(pipe, "chain", 0) == 1:
    import ctypes
    from deeprl_amd._lib import lib
    lat = []
    orig_step = L.step.__wrapped__ if hasattr(L.step, "__wrapped__") else None
    real_sync = lib.dra_dqn_learner_sync_loss
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.step()                      # begins with sync_loss of the previous update (complete: ~0), ends after issuing
        t1 = time.perf_counter()
        real_sync(L.h)
        t2 = time.perf_counter()
        lat.append((t1 - t0, t2 - t1))
    a = np.asarray(lat[20:])
    print("idle GPU: agent.step() host time %.1f us, then issue -> loss event %.1f us" % (1e6 * a[:, 0].mean(), 1e6 * a[:, 1].mean()))
