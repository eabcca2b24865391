#!/usr/bin/env python
"""Maps CU-mask bits (hipExtStreamCreateWithCUMask) to XCDs / shader engines / CUs on this GPU: for every
single-bit mask a one-workgroup kernel reports where it ran."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeprl_amd as d
from deeprl_amd._lib import lib

d.select_device(0)
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
words = (n_cu + 31) // 32
out = torch.zeros(2 * 64, dtype=torch.int32, device="cuda")
rows = []
for b in range(n_cu):
    mask = (ctypes.c_uint32 * words)()
    mask[b // 32] = 1 << (b % 32)
    h = ctypes.c_void_p()
    lib.dra_stream_create_masked(ctypes.byref(h), mask, words)
    lib.dra_probe_hw_id(ctypes.c_void_p(out.data_ptr()), 8, h)
    torch.cuda.synchronize()
    v = out[:16].cpu().tolist()
    xcc = sorted(set(x & 0xf for x in v[0::2]))
    cu = sorted(set(((x >> 8) & 0xf, (x >> 12) & 1, (x >> 13) & 7) for x in v[1::2]))
    rows.append((b, xcc, cu))
    lib.dra_stream_destroy(h)
print("n_cu", n_cu)
for b, xcc, cu in rows:
    print(b, "xcc", xcc, "cu/sh/se", cu)
