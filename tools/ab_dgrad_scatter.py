#!/usr/bin/env python
"""DRA_VAR_DGRAD_SCATTER against the gather-form input gradient (conv2 / conv3, rollout batch sizes): maximum deviation of dX
from a float64 F.conv2d backward for both forms, run-to-run identity, and HIP-event microseconds of the whole backward launch(es)
and of each role alone (DRA_VAR_MEASURE_*), interleaved on one box.

    python tools/ab_dgrad_scatter.py [batches ...]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deeprl_amd as d  # noqa: E402
from deeprl_amd import ops  # noqa: E402


def timed(call, reps=30):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    import torch.nn.functional as F
    d.select_device(0)
    dev = d.Config.DEVICE
    batches = [int(a) for a in sys.argv[1:]] or [128, 131, 256, 512, 1024, 1025]
    GEOM = {2: (32, 20, 64, 4, 2), 3: (64, 9, 64, 3, 1)}
    base = ops.VAR_FUSED_BWD | ops.VAR_ONESHOT_DGRAD | ops.VAR_ONESHOT_WGRAD
    forms = (("gather", base), ("scatter", base | ops.VAR_DGRAD_SCATTER),
             ("scatter_two_launches", (base | ops.VAR_DGRAD_SCATTER) & ~ops.VAR_FUSED_BWD))
    for B in batches:
        for layer, (c, h, oc, kh, s) in GEOM.items():
            oh = (h - kh) // s + 1
            g = torch.Generator(device="cpu").manual_seed(1000 * layer + B)
            x = torch.relu(torch.randn(B, c, h, h, generator=g))
            w = torch.randn(oc, c, kh, kh, generator=g) / np.sqrt(c * kh * kh)
            dy = torch.randn(B, oc, oh, oh, generator=g)
            xt = x.double().requires_grad_(True)
            F.conv2d(xt, w.double(), None, stride=s).backward(dy.double())
            ref = (xt.grad * (x > 0)).numpy()
            wt = ops.to_koc(w.to(dev))
            xd, dyd = x.to(dev), dy.to(dev)
            wt4 = wt.view(c, kh, kh, oc)
            rec = {"batch": B, "layer": layer}
            outs = {}
            fl = 2.0 * B * oh * oh * oc * c * kh * kh
            for name, var in forms:
                dw_s, db_s, dx, _ = ops.conv_bwd_fused(layer, dyd, xd, wt=wt, xact=xd, ksplit=16, variant=var)
                torch.cuda.synchronize()
                outs[name] = (dx.cpu().numpy().copy(), dw_s.sum(0).cpu().numpy().copy())
                dx2 = ops.conv_bwd_fused(layer, dyd, xd, wt=wt, xact=xd, ksplit=16, variant=var)[2]
                rec[name + "_rerun_identical"] = bool(torch.equal(dx, dx2))
                rec[name + "_dx_err_of_scale"] = float(np.abs(outs[name][0] - ref).max() / np.abs(ref).max())
                rec[name + "_nan"] = bool(np.isnan(outs[name][0]).any())
                us = timed(lambda: ops.conv_bwd_fused_koc(layer, dyd, xd, wt4, ksplit=16, variant=var, xact=xd))
                rec[name + "_us"] = round(us, 1)
                rec[name + "_frac"] = round(2 * fl / us / 1e6 / 157.3, 3)
                if name != "scatter_two_launches":
                    usd = timed(lambda: ops.conv_bwd_fused_koc(layer, dyd, xd, wt4, ksplit=16, variant=var | 2097152, xact=xd))
                    usw = timed(lambda: ops.conv_bwd_fused_koc(layer, dyd, xd, wt4, ksplit=16, variant=var | 4194304, xact=xd))
                    rec[name + "_dgrad_only_us"] = round(usd, 1)
                    rec[name + "_dgrad_only_frac"] = round(fl / usd / 1e6 / 157.3, 3)
                    rec[name + "_wgrad_only_us"] = round(usw, 1)
            rec["weight_gradient_identical"] = bool(np.array_equal(outs["gather"][1], outs["scatter"][1]))
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
