#!/usr/bin/env python
"""NatureConv forward, input gradient and weight gradient at the multi-minibatch shapes the on-policy agents run (ppo_pixel
minibatch 256, and 512 / 1024): per layer and pass, HIP-event microseconds per call and the FLOP-derived fraction of the fp32
MFMA peak (157.3 TFLOP/s).  The backward is the launch nets._ConvKocFn.backward makes (dra_conv_bwd_fused: weight + input
gradient in ONE launch; the one-pass slab kernels up to DRA_ONESHOT_WGRAD_MAX_BATCH = 256, the K-chunked split-K weight
gradient above).  tools/conv_big_counters.sh runs this under rocprofv3 (kernel durations) and under a PMC pass
(SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE: counter-derived MFMA utilisation next to the FLOP-derived one).

    python tools/conv_big_bwd.py <batch> [reps] [roles]      (roles: also the input- / weight-gradient role of conv2 / conv3 alone)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deeprl_amd as d  # noqa: E402
from deeprl_amd import nets, ops  # noqa: E402


def main():
    d.select_device(0)
    dev = d.Config.DEVICE
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    roles = len(sys.argv) > 3 and sys.argv[3] == "roles"     # also time each role of the backward launch alone
    GEOM = {1: (4, 84, 32, 8, 4), 2: (32, 20, 64, 4, 2), 3: (64, 9, 64, 3, 1)}
    out = {}
    for layer, (c, h, oc, kh, s) in GEOM.items():
        oh = (h - kh) // s + 1
        x = (torch.randint(0, 256, (B, c, h, h), dtype=torch.uint8, device=dev) if layer == 1
             else torch.relu(torch.randn(B, c, h, h, device=dev)))
        wt = torch.randn(c * kh * kh, oc, device=dev) * 0.05
        bb = torch.randn(oc, device=dev) * 0.05
        dy = torch.randn(B, oc, oh, oh, device=dev)
        coef = 1.0 / 255 if layer == 1 else None
        one = B <= nets._ONESHOT_WGRAD_MAX_BATCH
        # (the variant nets._ConvKocFn.backward passes: from round 6 on with the scatter-form input gradient, which acts from 256 samples;
        # DRA_CONV_BIG_GATHER=1 measures the gather form instead)
        scatter = 0 if os.environ.get("DRA_CONV_BIG_GATHER") else ops.VAR_DGRAD_SCATTER
        variant = (ops.VAR_FUSED_BWD | ops.VAR_ONESHOT_DGRAD | ((ops.VAR_ONESHOT_WGRAD | scatter) if one else 0))
        calls = {"fwd": lambda: ops.conv_fwd_koc(layer, [x], [wt], [bb], u8_coef=coef),
                 "bwd": lambda: ops.conv_bwd_fused_koc(layer, dy, x, wt.view(c, kh, kh, oc), ksplit=16, u8_coef=coef, variant=variant)}
        fl_pass = 2.0 * B * oh * oh * oc * c * kh * kh
        flops = {"fwd": fl_pass, "bwd": fl_pass * (1 if layer == 1 else 2)}       # conv1 has no input gradient
        if layer > 1 and roles:       # each role of the backward launch alone (DRA_VAR_MEASURE_*: the other role's outputs are not written)
            calls["bwd_x_only"] = lambda: ops.conv_bwd_fused_koc(layer, dy, x, wt.view(c, kh, kh, oc), ksplit=16, u8_coef=coef,
                                                                 variant=variant | 2097152)
            calls["bwd_w_only"] = lambda: ops.conv_bwd_fused_koc(layer, dy, x, wt.view(c, kh, kh, oc), ksplit=16, u8_coef=coef,
                                                                 variant=variant | 4194304)
            flops["bwd_x_only"] = flops["bwd_w_only"] = fl_pass
        for name, call in calls.items():
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
            out["conv%d_%s" % (layer, name)] = {"us": round(best, 1), "flops": flops[name],
                                                "frac_of_157.3_TFLOPs": round(flops[name] / best / 1e6 / 157.3, 3)}
    print(json.dumps({"batch": B, "oneshot_wgrad": B <= nets._ONESHOT_WGRAD_MAX_BATCH, "dgrad_scatter": not os.environ.get("DRA_CONV_BIG_GATHER"), "timing": "HIP events, best of 3 x %d calls" % reps,
                      "kernels": out}))


if __name__ == "__main__":
    main()
