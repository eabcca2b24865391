#!/usr/bin/env python
"""NatureConv forward at a large batch, per layer: best-of-3 microseconds per call and fraction of the fp32 MFMA peak."""
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import deeprl_amd as d
from deeprl_amd import ops
d.select_device(0); dev = d.Config.DEVICE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
GEOM = {1: (4, 84, 32, 8, 4), 2: (32, 20, 64, 4, 2), 3: (64, 9, 64, 3, 1)}
out = {}
for layer, (c, h, oc, kh, s) in GEOM.items():
    oh = (h - kh) // s + 1
    x = (torch.randint(0, 256, (B, c, h, h), dtype=torch.uint8, device=dev) if layer == 1 else torch.randn(B, c, h, h, device=dev))
    wt = torch.randn(c * kh * kh, oc, device=dev) * 0.05; bb = torch.randn(oc, device=dev) * 0.05
    call = lambda: ops.conv_fwd_koc(layer, [x], [wt], [bb], u8_coef=1.0 / 255 if layer == 1 else None)
    for _ in range(10): call()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 30)
    fl = 2.0 * B * oh * oh * oc * c * kh * kh
    out["conv%d" % layer] = (round(best, 1), round(fl / best / 1e6 / 157.3, 3))
print(json.dumps({"DRA_CONV2_MODE": os.environ.get("DRA_CONV2_MODE", "2"), "batch": B, "us_and_frac_of_157.3_TFLOPs": out}))
