#!/usr/bin/env python
"""Host side of DQNAgent.step() on the device pipeline (the drop-in API's dqn_pixel with a device-resident environment): cProfile of
3000 steady-state steps (where the python microseconds go) next to the plain rate.  usage: python tools/diag_agent_host.py [case]"""
import cProfile, io, json, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_agents as ba
import deeprl_amd as d
import deeprl_amd.agents as agents_mod


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "dqn_pixel_uniform_device"
    agents_mod.get_logger = lambda *x, **k: ba._Quiet()
    d.select_device(0)
    d.random_seed(0)
    agent, meta = ba.CASES[case]()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    for _ in range(max(8, n // 8)):
        agent.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        agent.step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"case": case, "us_per_step": 1e6 * dt / n, "host_loop_us_per_step": 1e6 * th / n,
                      "updates_per_s": n * meta["updates_per_step"] / dt, "env_steps_per_s": n * meta["env_per_step"] / dt}))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        agent.step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
    print(s.getvalue()[:7000])


if __name__ == "__main__":
    main()
