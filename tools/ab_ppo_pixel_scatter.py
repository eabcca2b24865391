import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch
import deeprl_amd as d
from deeprl_amd import ops
import bench_agents as ba
d.select_device(0)
bit = ops.VAR_DGRAD_SCATTER
for rep in range(2):
    for on in (False, True):
        ops.VAR_DGRAD_SCATTER = bit if on else 0
        agent, per = ba.ppo_pixel()
        agent.logger = ba._Quiet() if hasattr(ba, "_Quiet") else agent.logger
        for _ in range(3):
            agent.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 5.0:
            agent.step(); n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"case": "ppo_pixel", "VAR_DGRAD_SCATTER": on, "env_steps_per_s": round(n * per["env_per_step"] / dt, 1), "agent_steps": n}), flush=True)
        try:
            agent.close()
        except Exception:
            pass
