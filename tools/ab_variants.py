#!/usr/bin/env python
"""A/B of the learner's kernel variants (DRA_VAR_* masks) in ONE process, interleaved rounds:
for every mask a DQNLearnerBench (BASELINE configs[1] shapes) is built on a shared-size ring and timed
for `--steps` agent steps per round in async and in-order actor mode; per-kernel-group times of the
update (HIP events, eager) are reported once per mask.  Prints one JSON object per line.

    python tools/ab_variants.py --masks 0,1,3,7,15,31,63,127 --rounds 3 --steps 600
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d  # noqa: E402
from deeprl_amd.learner import DQNLearnerBench  # noqa: E402


def run(bench, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bench.step()
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="0,1,3,7,15,31,63,127")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--ring", type=int, default=200_000)
    ap.add_argument("--profile", type=int, default=100, help="eager updates averaged for the per-kernel times")
    args = ap.parse_args()
    d.select_device(0)
    masks = [int(m) for m in args.masks.split(",")]
    benches = {}
    for m in masks:
        for mode in ("async", "sync"):
            np.random.seed(0)
            torch.manual_seed(0)
            try:
                b = DQNLearnerBench(ring_capacity=args.ring, batch=32, seed=0, actor=True, async_actor=(mode == "async"),
                                    variant=m)
                for _ in range(100):
                    b.step()
                torch.cuda.synchronize()
                benches[(m, mode)] = b
            except Exception as e:  # a variant that fails must not hide the others
                print(json.dumps({"mask": m, "mode": mode, "error": repr(e)}), flush=True)
    rates = {k: [] for k in benches}
    for _ in range(args.rounds):
        for k, b in benches.items():
            rates[k].append(run(b, args.steps))
    for (m, mode), b in benches.items():
        rec = {"mask": m, "mode": mode, "updates_per_s_median": float(np.median(rates[(m, mode)])),
               "updates_per_s_max": float(np.max(rates[(m, mode)])), "rounds": [round(r, 1) for r in rates[(m, mode)]]}
        if mode == "sync":
            b.roofline(args.profile)
            rec["kernel_us"] = {k: round(1e3 * v, 2) for k, v in b.kernel_ms.items()}
            rec["kernel_us_sum"] = round(1e3 * sum(b.kernel_ms.values()), 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
