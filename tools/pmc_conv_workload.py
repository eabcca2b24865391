#!/usr/bin/env python
"""NatureConv forward kernels at batch 1024 (and 32), a few launches each: workload for rocprofv3 --pmc passes
that explain where the MFMA pipe idles (SQ wait / busy / LDS-conflict counters)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
from deeprl_amd import ops

d.select_device(0)
dev = d.Config.DEVICE
batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1024").split(",")]
for layer, (c, h, oc, k, s) in ops._CONV_GEOM.items():
    for batch in batches:
        x = (torch.randint(0, 256, (batch, c, h, h), device=dev, dtype=torch.uint8) if layer == 1
             else torch.rand(batch, c, h, h, device=dev))
        wt = torch.randn(c * k * k, oc, device=dev) * 0.05
        bb = torch.zeros(oc, device=dev)
        for _ in range(6):
            ops.conv_fwd_koc(layer, [x], [wt], [bb], u8_coef=1.0 / 255 if layer == 1 else None)
        torch.cuda.synchronize()
