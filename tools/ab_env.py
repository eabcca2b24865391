#!/usr/bin/env python
"""Same-box A/B of ENVIRONMENT switches on the benchmarked pipeline (DQNLearnerBench, async actor, BASELINE configs[1] shapes):
every setting runs in its own process (the switches are read once per process), interleaved over `--rounds` repetitions.
Per setting: median / max updates/s and the eager per-kernel-group event times of the update.

    python tools/ab_env.py --rounds 3 --steps 3000 DRA_BWD_LIN=0 DRA_BWD_LIN=3 "DRA_BWD_LIN=3 DRA_X=1"
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(steps, ring):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import deeprl_amd as d
    from deeprl_amd.learner import DQNLearnerBench
    d.select_device(0)
    np.random.seed(0)
    torch.manual_seed(0)
    b = DQNLearnerBench(ring_capacity=ring, batch=32, seed=0, actor=True, async_actor=True)
    for _ in range(400):
        b.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.step()
    torch.cuda.synchronize()
    rate = steps / (time.perf_counter() - t0)
    b.roofline(150)
    digest = float(b.learner.flat.flat.double().sum().item())
    print(json.dumps({"updates_per_s": rate, "kernel_us": {k: round(1e3 * v, 2) for k, v in b.kernel_ms.items()},
                      "param_sum_after": digest}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--ring", type=int, default=200_000)
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("settings", nargs="*")
    args = ap.parse_args()
    if args.worker:
        return worker(args.steps, args.ring)
    res = {s: [] for s in args.settings}
    for _ in range(args.rounds):
        for s in args.settings:
            env = dict(os.environ)
            for kv in s.split():
                k, v = kv.split("=", 1)
                env[k] = v
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--steps", str(args.steps), "--ring", str(args.ring)],
                                 env=env, capture_output=True, text=True)
            try:
                res[s].append(json.loads(out.stdout.strip().splitlines()[-1]))
            except Exception:
                res[s].append({"error": (out.stderr or out.stdout)[-400:]})
    for s, rs in res.items():
        ok = [r for r in rs if "updates_per_s" in r]
        rates = sorted(r["updates_per_s"] for r in ok)
        rec = {"setting": s, "updates_per_s_median": rates[len(rates) // 2] if rates else None, "updates_per_s_all": [round(r, 1) for r in rates],
               "kernel_us": ok[-1]["kernel_us"] if ok else None, "param_sum_after": [r["param_sum_after"] for r in ok],
               "errors": [r["error"] for r in rs if "error" in r]}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
