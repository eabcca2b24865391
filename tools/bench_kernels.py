#!/usr/bin/env python
"""Stand-alone kernel measurements quoted in DESIGN.md (HIP events on the launch stream):
  * ring gather (K1) at K minibatches per launch on the 1M-frame ring -> HBM GB/s vs the roofline;
  * single-minibatch gather latency;
  * clip+RMSprop step (K11) bytes/s;  GAE scan (K9) latency;  sum-tree sample/update latency;
  * NatureConv forward (conv_v2.hip) at batch 32 x K -> fp32 MFMA TFLOP/s.
Prints one JSON object."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d  # noqa: E402
from deeprl_amd import ops  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3  # seconds


def main():
    d.select_device(0)
    dev = d.Config.DEVICE
    out = {}
    cap = int(os.environ.get("RING", 1_000_000))
    ring = ops.Ring(cap, 7056, 8, 4, 1, 0.99)
    ring.fill_synthetic(0, cap, 0, 0)
    torch.cuda.synchronize()
    rs = np.random.RandomState(0)
    for k in (1, 64, 1024):
        b = 32 * k
        idx = torch.from_numpy(rs.randint(3, cap - 2, size=b).astype(np.int64)).to(dev)
        # both output forms: two stacked tensors (the reference's layout: 2 x 4 frames written per sample) and, from round 5 on what
        # sample() returns, one [B, 5, 84, 84] block with state / next_state as views (5 frames written per sample)
        for form, blockf, frames_written in (("", False, 8), ("_block", True, 5)):
            bufs = ring.gather(idx, (84, 84), torch.uint8, torch.int64, want_f32=True, block=blockf)
            t = timeit(lambda: ring.gather(idx, (84, 84), torch.uint8, torch.int64, want_f32=True, out=bufs), n=10 if k > 64 else 50)
            rd, wr = b * 5 * 7056, b * frames_written * 7056
            out["gather%s_k%d" % (form, k)] = {"minibatches": k, "seconds": t, "read_GBps": rd / t / 1e9,
                                             "read_plus_write_GBps": (rd + wr) / t / 1e9, "frac_hbm_read_8TBps": rd / t / 8e12,
                                             "frac_hbm_total_8TBps": (rd + wr) / t / 8e12, "frac_hbm_total_6.3TBps": (rd + wr) / t / 6.3e12}
            del bufs
    ring.close()
    # clip + centered RMSprop over the DQN parameter count
    n = 1_686_180
    p, g, s1, s2 = [torch.randn(n, device=dev) for _ in range(4)]
    s1.abs_()
    partials = torch.zeros(ops.norm_partials(), dtype=torch.float64, device=dev)
    def opt():
        ops.grad_sqnorm(g, partials)
        ops.rmsprop_step(p, g, s1, s2, partials, ops.norm_partials(), 5.0, 2.5e-4, 0.95, 0.01, True)
    t = timeit(opt, n=50)
    out["clip_rmsprop"] = {"seconds": t, "GBps": 32 * n / t / 1e9, "frac_hbm_8TBps": 32 * n / t / 8e12}
    # GAE scan
    for t_len, n_env in ((2048, 16), (128, 8), (5, 16)):
        r = torch.randn(t_len, n_env, 1, device=dev)
        m = (torch.rand(t_len, n_env, 1, device=dev) > 0.01).float()
        v = torch.randn(t_len + 1, n_env, 1, device=dev)
        t = timeit(lambda: ops.gae(r, m, v, 0.99, 0.95, True), n=50)
        out["gae_T%d_N%d" % (t_len, n_env)] = {"microseconds": t * 1e6, "bytes": (5 * t_len + 1) * n_env * 4}
    # sum tree
    tree = ops.SumTree(1_000_000)
    view = tree.as_tensor()
    view[999_999:] = torch.rand(1_000_000, device=dev, dtype=torch.float64).float().double() + 0.1
    tree.rebuild()
    u = torch.rand(32, dtype=torch.float64, device=dev)
    out["sumtree_sample_us"] = timeit(lambda: tree.sample(u), n=100) * 1e6
    leaf = torch.from_numpy(rs.choice(1_000_000, 32, replace=False).astype(np.int64) + 999_999).to(dev)
    pr = torch.rand(32, dtype=torch.float64, device=dev).float().double() + 0.1
    out["sumtree_update_parallel_us"] = timeit(lambda: tree.update(leaf, pr), n=100) * 1e6
    out["sumtree_update_ordered_us"] = timeit(lambda: tree.update(leaf, pr, ordered=True), n=20) * 1e6
    tree.close()
    # conv forward at large batch (MFMA rate once latency is amortised)
    for layer, (c, h, oc, k, s) in ops._CONV_GEOM.items():
        o = (h - k) // s + 1
        for batch in (32, 1024):
            x = (torch.randint(0, 256, (batch, c, h, h), device=dev, dtype=torch.uint8) if layer == 1
                 else torch.rand(batch, c, h, h, device=dev))
            wt = torch.randn(c * k * k, oc, device=dev) * 0.05
            bb = torch.zeros(oc, device=dev)
            t = timeit(lambda: ops.conv_fwd_koc(layer, [x], [wt], [bb], u8_coef=1.0 / 255 if layer == 1 else None), n=30)
            fl = 2.0 * batch * o * o * oc * c * k * k
            out["conv%d_fwd_b%d" % (layer, batch)] = {"microseconds": t * 1e6, "TFLOPs": fl / t / 1e12, "frac_mfma_f32": fl / t / 157.3e12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
