#!/usr/bin/env python
"""Where do two runs of tests/_switch_probe.py part ways?  diag_ks.py a.npz b.npz"""
import sys
import numpy as np
a, b = dict(np.load(sys.argv[1])), dict(np.load(sys.argv[2]))
neq = np.nonzero(a["act"] != b["act"])[0]
print("actions differing:", len(neq), "first at transition", (int(neq[0]) if len(neq) else None), "of", len(a["act"]))
for k in a:
    if k.startswith("p_"):
        d = np.abs(a[k] - b[k])
        print("%-28s max|diff| %.3e  max|a| %.3e" % (k, d.max(), np.abs(a[k]).max()))
