#!/usr/bin/env python
"""Small workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE are collected in separate runs):
  1. calibration: the ring gather at 1024 minibatches per launch on a ring that does not fit the 256 MiB
     Infinity Cache -- a kernel whose HBM bytes are known exactly (read 32*1024*5*7056 B with 16 B/lane
     loads, write 2*32*1024*4*7056 B);
  2. `--steps` in-order agent steps of the DQN learner (BASELINE configs[1] shapes) with `--variant`.
tools/pmc_traffic.py turns the two counter CSVs into per-kernel bytes per launch."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d  # noqa: E402
from deeprl_amd import ops  # noqa: E402
from deeprl_amd.learner import DQNLearnerBench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--ring", type=int, default=400_000)
    ap.add_argument("--no-calibration", action="store_true", help="skip the 1024-minibatch gather launches")
    ap.add_argument("--async-actor", action="store_true",
                    help="the benchmarked two-stream pipeline (the chained launches run only there).  Counter collection serialises "
                         "kernels: pass a --variant without DRA_VAR_FLAG_SYNC / DRA_VAR_ACTOR_PERSIST, whose launches wait for each other "
                         "ACROSS streams on device words")
    args = ap.parse_args()
    d.select_device(0)
    dev = d.Config.DEVICE
    bench = DQNLearnerBench(ring_capacity=args.ring, batch=32, seed=0, actor=True, async_actor=args.async_actor, variant=args.variant)
    if not args.no_calibration:
        rs = np.random.RandomState(0)
        idx = torch.from_numpy(rs.randint(3, args.ring - 2, size=32 * 1024).astype(np.int64)).to(dev)
        bufs = bench.ring.gather(idx, (84, 84), torch.uint8, torch.int64, want_f32=True, block=False)     # the two-tensor form the calibration assumes
        for _ in range(3):
            bench.ring.gather(idx, (84, 84), torch.uint8, torch.int64, want_f32=True, out=bufs)
        torch.cuda.synchronize()
    for _ in range(args.steps):
        bench.step()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
