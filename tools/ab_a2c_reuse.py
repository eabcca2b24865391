#!/usr/bin/env python
"""a2c_pixel (16 workers x 5 steps) env-steps/s with config.reuse_rollout_activations cleared and set, interleaved on one box."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import deeprl_amd as d
import bench_agents as ba
d.select_device(0)
for rep in range(3):
    for on in (False, True):
        agent, per = ba.a2c_pixel(16, reuse_rollout_activations=on)
        for _ in range(30):
            agent.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 3.0:
            agent.step(); n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"case": "a2c_pixel_16", "reuse_rollout_activations": on, "env_steps_per_s": round(n * per["env_per_step"] / dt, 1), "agent_steps": n}), flush=True)
        agent.close()
