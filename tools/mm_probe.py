#!/usr/bin/env python
"""What a library fp32 GEMM does at fc4's update shapes (a yardstick for the hand-written kernels): torch.mm under HIP events."""
import json
import torch
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False


def t_us(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


for m in (80, 256, 1024):
    x = torch.randn(m, 3136, device=dev)
    w = torch.randn(512, 3136, device=dev)
    b = torch.randn(512, device=dev)
    dy = torch.randn(m, 512, device=dev)
    out = {"batch": m}
    out["fwd_addmm_us"] = t_us(lambda: torch.addmm(b, x, w.t()))
    out["dgrad_mm_us"] = t_us(lambda: torch.mm(dy, w))
    out["wgrad_mm_us"] = t_us(lambda: torch.mm(dy.t(), x))
    fl = 2.0 * m * 3136 * 512
    for k in ("fwd_addmm_us", "dgrad_mm_us", "wgrad_mm_us"):
        out[k.replace("_us", "_tflops")] = fl / out[k] / 1e6
    print(json.dumps(out))
