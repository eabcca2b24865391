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
    for _ in range(400):
        agent.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        agent.step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"case": case, "us_per_step": 1e6 * dt / 3000, "host_loop_us_per_step": 1e6 * th / 3000,
                      "updates_per_s": 3000 * meta["updates_per_step"] / dt}))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        agent.step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print(s.getvalue()[:5000])


if __name__ == "__main__":
    main()
