"""Diagnostic (not a test): per-step divergence between the pipelined DQN step on the GPU and the schedule oracle.
    python tests/diag_schedule.py [async|sync] [variant] [init: ortho|normal]"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import deeprl_amd as d  # noqa: E402
import fake_envs  # noqa: E402
from deeprl_amd.learner import DQNLearnerBench  # noqa: E402
from oracle.async_schedule_oracle import AsyncDqnScheduleOracle  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "async"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else -1
init = sys.argv[3] if len(sys.argv) > 3 else "ortho"
d.select_device(0)
cap, b, a, seed, steps = 4000, 32, 4, 3, 10
d.random_seed(11)
torch.manual_seed(5)
bench = DQNLearnerBench(ring_capacity=cap, batch=b, seed=seed, actor=True, async_actor=(mode == "async"), variant=variant)
if init == "normal":
    p0 = fake_envs.numpy_params(fake_envs.nature_vanilla_shapes(a), 21)
    bench.network.load_state_dict({k: torch.from_numpy(v) for k, v in p0.items()})
    bench.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in p0.items()})
p_np = {k: v.detach().cpu().numpy().copy() for k, v in bench.network.state_dict().items()}
orc = AsyncDqnScheduleOracle(p_np, p_np, cap, b, seed, n_actions=a, epsilon=bench.epsilon)
if mode != "async":
    orc.actor_rs = np.random   # in-order mode: one global stream, draw for draw
np.random.seed(5)
gd, gp = [], []
for _ in range(steps):
    bench.step()
    bench.learner.synchronize()
    gd.append(bench.learner.delta.cpu().numpy().copy())
    gp.append({k: v.detach().cpu().numpy().copy() for k, v in bench.network.state_dict().items()})
n_tr = 4 * (steps + 1)
ga = d.ops._wrap_device_pointer(bench.ring.pointers()[1], n_tr, torch.int64).cpu().numpy().copy()
np.random.seed(5)
if mode == "async":
    orc.actor_step(orc._snapshot(), override_actions=ga[0:4])
for k in range(steps):
    if mode == "async":
        idx, batch = orc.sample()
        res = orc.actor_step(orc._snapshot(), override_actions=ga[4 * (k + 1):4 * (k + 2)])
    else:
        res = orc.actor_step(orc._snapshot(), override_actions=ga[4 * k:4 * (k + 1)])
        idx, batch = orc.sample()
    loss, delta, q, norm = orc.update(batch)
    perr = max(float(np.abs(gp[k][n] - orc.p[n].detach().numpy()).max()) for n in gp[k])
    worst = max(gp[k], key=lambda n: float(np.abs(gp[k][n] - orc.p[n].detach().numpy()).max()))
    mism = [(e, r[0], int(ga[(4 * (k + 1) if mode == 'async' else 4 * k) + e]), r[1]) for e, r in enumerate(res)
            if r[0] != ga[(4 * (k + 1) if mode == 'async' else 4 * k) + e]]
    print("step %2d  margin %.1e  delta err %.2e  qmax %.3f  loss %.5f  norm %.4f  param err %.2e (%s)  action mismatches %s" % (
        k, orc.relu_margin, float(np.abs(gd[k] - delta).max()), float(np.abs(q).max()), loss, norm, perr, worst, mism), flush=True)
